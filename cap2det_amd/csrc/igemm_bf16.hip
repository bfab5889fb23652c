// bf16 implicit-GEMM convolution with a DEEP LDS-DMA RING (second stage in the bf16 storage mode,
// BASELINE configs[2] / [4]: slim.conv2d of models/utils.py:165-167 and its input gradient).
//
// Same iteration space, tap tables, pixel-major tile skipping, multi-segment / multi-output modes
// and epilogue as round 2's igemm_bf16_kernel (which this kernel replaces), which staged ONE slab ahead: a
// workgroup then waits out the whole global -> LDS latency of a slab (~2,600 cycles under load)
// for 512-1,400 cycles of MFMAs per slab, and issues all of a slab's DMA pieces in front of its
// MFMAs (6-8 wave-instructions of ~100 issue cycles each, both waves of a SIMD at the same time
// behind the barrier).  Here
//   * operand stages go through a ring of D buffers: D - 2 stages stay in flight across the
//     barrier (counted s_waitcnt vmcnt, raw s_barrier — never __syncthreads() with a DMA pending,
//     cdna_hip_programming.md §5 "Pipelining across barriers");
//   * a stage is BKT = 64 or 32 elements of K deep (128- or 64-byte rows): the 32-deep form makes
//     a ring of three small enough for TWO workgroups per CU on the 128x256 tile.
// What the sweeps showed (launch_tile below): deep rings lose, co-resident workgroups win.
// LDS image of a stage: unpadded rows; 16-byte chunk c of row r sits at position c ^ swz(r) with
// swz(r) = (r >> 1) & 7 for 128-byte rows and (r >> 2) & 3 for 64-byte rows — applied to the DMA's
// per-lane GLOBAL address and to the fragment reads (an LDS-DMA image is lane-linear) — which puts
// the 16 lanes of every ds_read_b128 group on 16 distinct 16-byte slots (MI355X_MICROARCH.md §LDS).
#include "igemm_ring.h"

namespace c2d_ig {
namespace {


// ---- host side ---------------------------------------------------------------------------------
// Tuning hooks (read once, only with C2D_TUNE=1): ring_bk=32|64, ring_d=2|3 (more in a
// -DC2D_RING_SWEEP build) force the stage depth / ring depth where an instance exists;
// ring=0 makes every bf16 convolution fail with C2D_ERR_UNSUPPORTED (there is no second bf16
// GEMM: a check that nothing falls back silently).
#ifdef C2D_RING_TRACE
unsigned long long* g_ring_trace = nullptr;
#endif
struct RingTune { int bk, d, off; };
const RingTune& ring_tune() {
  static const RingTune t = [] {
    RingTune r = {0, 0, 0};
    if (c2d_tune_on()) {
      if (const char* e = c2d_tune_get("ring_bk")) r.bk = atoi(e);
      if (const char* e = c2d_tune_get("ring_d")) r.d = atoi(e);
      if (const char* e = c2d_tune_get("ring")) r.off = e[0] == '0';
    }
    return r;
  }();
  return t;
}

// Collector of ring_group_begin() .. ring_group_end(): launches of group-eligible instances are held
// back (every problem with the launchers of ITS instance) and leave as one grouped launch when they
// all chose the same instance, one by one otherwise.
struct RingCollect {
  bool active;
  int num;
  IgemmArgs a[4];
  int (*single[4])(const IgemmArgs&, hipStream_t);
  int (*group[4])(const IgemmRingGroup&, hipStream_t);
};
thread_local RingCollect g_ring_collect = {false, 0, {}, {}, {}};

template <int MODE, int WM, int WN, int MT, int NT, bool PM, int BKT, int D, int ES, bool FUSED>
int ring_launch_single(const IgemmArgs& a, hipStream_t s) {
  // (spelled as rocprofv3 prints the instance: the FUSED flag included)
  dispatch_note_ext(PM ? (FUSED ? "igemm_ring_kernel<%d, %d, %d, %d, %d, true, %d, %d, %d, true, 1>"
                                : "igemm_ring_kernel<%d, %d, %d, %d, %d, true, %d, %d, %d, false, 1>")
                       : (FUSED ? "igemm_ring_kernel<%d, %d, %d, %d, %d, false, %d, %d, %d, true, 1>"
                                : "igemm_ring_kernel<%d, %d, %d, %d, %d, false, %d, %d, %d, false, 1>"),
                    MODE, WM, WN, MT, NT, BKT, D, ES);
  hipLaunchKernelGGL((igemm_ring_kernel<MODE, WM, WN, MT, NT, PM, BKT, D, ES, FUSED>),
                     dim3(a.m_tiles * a.n_tiles), dim3(WM * WN * 64), 0, s, a);
  return c2d_launch_status();
}

template <int MODE, int WM, int WN, int MT, int NT, bool PM, int BKT, int D, int ES, bool FUSED>
int ring_launch_group(const IgemmRingGroup& g, hipStream_t s) {
  dispatch_note_ext(FUSED ? "igemm_ring_group_kernel<%d, %d, %d, %d, %d, false, %d, %d, %d, true, 1>"
                          : "igemm_ring_group_kernel<%d, %d, %d, %d, %d, false, %d, %d, %d, false, 1>",
                    MODE, WM, WN, MT, NT, BKT, D, ES);
  hipLaunchKernelGGL((igemm_ring_group_kernel<MODE, WM, WN, MT, NT, PM, BKT, D, ES, FUSED>),
                     dim3(g.first[g.num]), dim3(WM * WN * 64), 0, s, g);
  return c2d_launch_status();
}

template <int MODE, int WM, int WN, int MT, int NT, bool PM, int BKT, int D, int ES = 2>
int launch_one(IgemmArgs a, hipStream_t s) {
  constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
  a.m_tiles = c2d_ceil_div(a.M, BM);
  a.n_tiles = c2d_ceil_div(a.N, BN);
  if (a.nseg > 1) {
    a.total_slabs = 0;
    for (int i = 0; i < a.nseg; ++i) a.total_slabs += c2d_ceil_div(a.segK[i], BKT);
  }
  static const int dbg_env = (c2d_tune_on() && c2d_tune_get("igemm_dbg")) ? atoi(c2d_tune_get("igemm_dbg")) : 0;
  a.dbg = dbg_env;
#ifdef C2D_RING_TRACE
  a.trace = g_ring_trace;
#endif
  // the row-major input-gradient instances of the stride-2 layers' tiles can leave in a grouped launch
  constexpr bool GROUPABLE = MODE == 1 && ES == 2 && !PM && WM == 2 && MT <= 2 &&
                             ((WN == 4 && MT == 2 && NT == 2) || (WN == 2 && NT == 1));
  if constexpr (MODE == 1 && ES == 2) {
    if (a.fy != nullptr) {      // the producer's BN/ReLU backward rides in the epilogue
      if constexpr (GROUPABLE) {
        RingCollect& c = g_ring_collect;
        if (c.active && c.num < 4) {
          c.a[c.num] = a;
          c.single[c.num] = &ring_launch_single<MODE, WM, WN, MT, NT, PM, BKT, D, ES, true>;
          c.group[c.num] = &ring_launch_group<MODE, WM, WN, MT, NT, PM, BKT, D, ES, true>;
          ++c.num;
          return C2D_OK;
        }
      }
      return ring_launch_single<MODE, WM, WN, MT, NT, PM, BKT, D, ES, true>(a, s);
    }
  } else if (a.fy != nullptr) {
    return C2D_ERR_UNSUPPORTED;   // (fp32 networks fuse in igemm_nt_kernel, conv_gemm.hip)
  }
  if constexpr (GROUPABLE) {
    RingCollect& c = g_ring_collect;
    if (c.active && c.num < 4) {
      c.a[c.num] = a;
      c.single[c.num] = &ring_launch_single<MODE, WM, WN, MT, NT, PM, BKT, D, ES, false>;
      c.group[c.num] = &ring_launch_group<MODE, WM, WN, MT, NT, PM, BKT, D, ES, false>;
      ++c.num;
      return C2D_OK;
    }
  }
  return ring_launch_single<MODE, WM, WN, MT, NT, PM, BKT, D, ES, false>(a, s);
}

// (stage depth, ring depth) of a block tile, measured per GEMM call of the step (tools/
// sweep_step_gemms.sh + tools/sweep_cold.py, N = 2000 ROIs; with the operands warm from the previous
// repetition and, as inside a step, cold: bench_cold=1).  What decides is how many workgroups
// share a CU, not how deep one workgroup prefetches: these GEMMs are short (K = 128 .. 2304: 2 - 36
// stages), so a workgroup spends as long in its ramp (first stages in flight) and its epilogue as
// in its K loop, and only ANOTHER workgroup's K loop fills the matrix pipe meanwhile.  Rings of
// four or more stages (one workgroup per CU) measured 10 - 45 % slower on every call.  So:
//   * 128x256 (8 waves): 32-deep stages in a ring of three (72 KiB: TWO workgroups per CU) —
//     first for launches of more than 1.5 rounds of the chip (block-entry input gradients 109 ->
//     84 us, 192->256 3x3 on 7x7 maps 126 -> 108 us); since the steady-state loop (half the
//     instructions per stage) also for the one-round launches of 4x4 maps (224->224 3x3 forward /
//     input gradient 47 -> 41 / 46 -> 39 us cold, 38.6 -> 34.5 / 37.4 -> 33.2 warm);
//   * 128x192 (8 waves, full width): the same ring for one-round launches (61 KiB: two per CU;
//     160->224 / 192->224 input gradients 41 -> 37.5 us cold); 80 KiB of 64-deep stages otherwise;
//   * 128x64 (4 waves): many-tile launches likewise (36 KiB: four per CU; 576-wide entry gradient
//     142 -> 102 us);
//   * everything else: 64-deep stages, two buffers.
#ifdef C2D_RING_SWEEP
#define K_RING_COMBOS(X) X(64, 2) X(32, 3) X(64, 3) X(32, 4)
#else
#define K_RING_COMBOS(X) X(64, 2) X(32, 3)
#endif
template <int MODE, int WM, int WN, int MT, int NT, bool PM>
int launch_tile(const IgemmArgs& a, hipStream_t s) {
  constexpr int ROWS = (WM * MT + WN * NT) * 32;
  constexpr int LDS = 160 * 1024;
  const RingTune& t = ring_tune();
  const long long blocks = (long long)c2d_ceil_div(a.M, WM * MT * 32) * c2d_ceil_div(a.N, WN * NT * 32);
  int bk = 64, d = 2;
  if (WM == 2 && WN == 4) { bk = 32; d = 3; }
  if (WM == 4 && WN == 2 && NT == 3 && blocks <= 384) { bk = 32; d = 3; }
  if (WM == 2 && WN == 2 && MT == 2 && NT == 1 && blocks > 768) { bk = 32; d = 3; }
  if (t.bk) bk = t.bk;
  if (t.d) d = t.d;
#define K_RING_CASE(BKT_, D_)                                                                \
  if constexpr (D_ * ROWS * BKT_ * 2 <= LDS && (D_ - 2) * (ROWS * BKT_ * 2 / (WM * WN * 1024) + 1) <= 63) \
    if (bk == BKT_ && d == D_) return launch_one<MODE, WM, WN, MT, NT, PM, BKT_, D_>(a, s);
  K_RING_COMBOS(K_RING_CASE)
#undef K_RING_CASE
  // (a forced combination that does not exist for this tile: the default)
  return launch_one<MODE, WM, WN, MT, NT, PM, 64, 2>(a, s);
}

template <int WM, int WN, int MT, int NT>
int launch_shape(const IgemmArgs& a, bool pm, hipStream_t s) {
  if (a.g.mode == 0)
    return pm ? launch_tile<0, WM, WN, MT, NT, true>(a, s) : launch_tile<0, WM, WN, MT, NT, false>(a, s);
  return pm ? launch_tile<1, WM, WN, MT, NT, true>(a, s) : launch_tile<1, WM, WN, MT, NT, false>(a, s);
}

// fp32 operands on the same ring.  Swept on the step's 26 forward / input-gradient calls (tools/
// bench_step_gemms.py fp32; 16- / 32-float stages, rings of 2-4): the register-staged
// igemm_nt_kernel of conv_gemm.hip stays ahead by 3-6 % overall (3.66 + 3.57 ms against 3.77 + 3.79
// with the best ring) — at 64 cycles per fp32 MFMA the staging instructions of a workgroup sit
// under the MFMAs of the three others on its CU — with ONE exception the default takes: the 3x3
// forward convolutions with 257..384 output columns on the per-ROI maps (192->320 of Mixed_5b/5c:
// five 64-column tiles per row block), 284 -> 234 us per call with 16-float stages in two buffers
// (24 KiB: five workgroups per CU).  C2D_TUNE=ring_fp32=1|0 forces the ring on (all tiles,
// sweep build) / off; ringf_bk = 16 | 32, ringf_d = ring depth (sweep build).
struct RingF32Tune { int on, bk, d; };
const RingF32Tune& ringf_tune() {
  static const RingF32Tune t = [] {
    RingF32Tune r = {-1, 0, 0};
    if (c2d_tune_on()) {
      if (const char* e = c2d_tune_get("ring_fp32")) r.on = atoi(e);
      if (const char* e = c2d_tune_get("ringf_bk")) r.bk = atoi(e);
      if (const char* e = c2d_tune_get("ringf_d")) r.d = atoi(e);
    }
    return r;
  }();
  return t;
}

#ifdef C2D_RING_SWEEP
template <int MODE, int WM, int WN, int MT, int NT, bool PM>
int launch_tile_f32(const IgemmArgs& a, hipStream_t s) {
  const RingF32Tune& t = ringf_tune();
  const int bk = t.bk == 32 ? 32 : 16;
  const int d = t.d >= 2 && t.d <= 4 ? t.d : 2;
  if (bk == 16 && d == 3) return launch_one<MODE, WM, WN, MT, NT, PM, 16, 3, 4>(a, s);
  if (bk == 16 && d == 4) return launch_one<MODE, WM, WN, MT, NT, PM, 16, 4, 4>(a, s);
  if (bk == 16 && d == 2) return launch_one<MODE, WM, WN, MT, NT, PM, 16, 2, 4>(a, s);
  if (bk == 32 && d == 3) return launch_one<MODE, WM, WN, MT, NT, PM, 32, 3, 4>(a, s);
  return launch_one<MODE, WM, WN, MT, NT, PM, 32, 2, 4>(a, s);
}

template <int WM, int WN, int MT, int NT>
int launch_shape_f32(const IgemmArgs& a, bool pm, hipStream_t s) {
  if (a.g.mode == 0)
    return pm ? launch_tile_f32<0, WM, WN, MT, NT, true>(a, s) : launch_tile_f32<0, WM, WN, MT, NT, false>(a, s);
  return pm ? launch_tile_f32<1, WM, WN, MT, NT, true>(a, s) : launch_tile_f32<1, WM, WN, MT, NT, false>(a, s);
}
#endif

}  // namespace

int launch_igemm_f32_ring(const IgemmArgs& a, int wm, int wn, int mt, int nt, bool pm,
                          hipStream_t s, int* m_tiles_out, bool query) {
  const int on = ringf_tune().on;
  if (on == 0) return C2D_ERR_UNSUPPORTED;
  const int key = ((wm * 10 + wn) * 10 + mt) * 10 + nt;
  const bool pick = key == 2221 && pm && a.g.mode == 0 && a.N > 256 && a.N <= 384 && !a.fy && !a.mo_n &&
                    a.nseg == 1;
#ifdef C2D_RING_SWEEP
  if (on == 1) {
    if (key != 2221 && key != 2222 && key != 2211) return C2D_ERR_UNSUPPORTED;
    if (m_tiles_out) *m_tiles_out = c2d_ceil_div(a.M, wm * mt * 32);
    if (query) return C2D_OK;
    switch (key) {
      case 2221: return launch_shape_f32<2, 2, 2, 1>(a, pm, s);      // 128 x 64
      case 2222: return launch_shape_f32<2, 2, 2, 2>(a, pm, s);      // 128 x 128
      default: return launch_shape_f32<2, 2, 1, 1>(a, pm, s);        // 64 x 64
    }
  }
#endif
  if (!pick) return C2D_ERR_UNSUPPORTED;
  if (m_tiles_out) *m_tiles_out = c2d_ceil_div(a.M, wm * mt * 32);
  if (query) return C2D_OK;
  return launch_one<0, 2, 2, 2, 1, true, 16, 2, 4>(a, s);
}

namespace {
}  // namespace

void ring_group_begin() {
  g_ring_collect.active = true;
  g_ring_collect.num = 0;
}

int ring_group_end(hipStream_t s) {
  RingCollect& c = g_ring_collect;
  c.active = false;
  const int num = c.num;
  c.num = 0;
  if (num == 0) return C2D_OK;
  bool same = num > 1;
  for (int i = 1; i < num; ++i) same = same && c.group[i] == c.group[0];
  if (!same) {
    for (int i = 0; i < num; ++i) {
      const int rc = c.single[i](c.a[i], s);
      if (rc) return rc;
    }
    return C2D_OK;
  }
  IgemmRingGroup g;
  g.num = num;
  int blocks = 0;
  for (int i = 0; i < 4; ++i) {
    g.first[i] = blocks;
    if (i < num) {
      g.a[i] = c.a[i];
      blocks += c.a[i].m_tiles * c.a[i].n_tiles;
    } else {
      g.a[i] = c.a[0];
    }
  }
  g.first[4] = blocks;
  for (int i = num; i < 4; ++i) g.first[i] = blocks;
  return c.group[0](g, s);
}

int launch_igemm_bf16_ring(const IgemmArgs& a, int wm, int wn, int mt, int nt, bool pm,
                           hipStream_t s, int* m_tiles_out, bool query) {
  if (ring_tune().off) return C2D_ERR_UNSUPPORTED;
  const int bm = wm * mt * 32;
  int rc = C2D_ERR_UNSUPPORTED;
  const int key = ((wm * 10 + wn) * 10 + mt) * 10 + nt;
  switch (key) {
    case 2221: case 2422: case 4212: case 4213: case 4214: case 4215: case 4216: case 2222: case 2211:
      break;
    default:
      return C2D_ERR_UNSUPPORTED;
  }
  if (m_tiles_out) *m_tiles_out = c2d_ceil_div(a.M, bm);
  if (query) return C2D_OK;
  switch (key) {
    case 2221: rc = launch_shape<2, 2, 2, 1>(a, pm, s); break;      // 128 x 64, 4 waves
    case 2222: rc = launch_shape<2, 2, 2, 2>(a, pm, s); break;      // 128 x 128, 4 waves
    case 2211: rc = launch_shape<2, 2, 1, 1>(a, pm, s); break;      // 64 x 64, 4 waves
    case 2422: rc = launch_shape<2, 4, 2, 2>(a, pm, s); break;      // 128 x 256, 8 waves (2 x 4)
    case 4212: rc = launch_shape<4, 2, 1, 2>(a, pm, s); break;      // 128 x full width, 8 waves (4 x 2)
    case 4213: rc = launch_shape<4, 2, 1, 3>(a, pm, s); break;
    case 4214: rc = launch_shape<4, 2, 1, 4>(a, pm, s); break;
    case 4215: rc = launch_shape<4, 2, 1, 5>(a, pm, s); break;
    case 4216: rc = launch_shape<4, 2, 1, 6>(a, pm, s); break;
  }
  return rc;
}


// ---------------------------------------------------------------------------------------------
// Filter gradient of a 1x1 / stride-1 convolution, bf16 operands (Conv2DBackpropFilter of the
// block-entry and branch-closing convolutions of Mixed_5a-c, train/trainer.py:141-146):
//   dW[i][j] += sum_m x[m][i] * dC[m][j]        (K = rows, split over workgroups)
// Round 2's kernel (wgrad_tn_bf16_kernel) staged 32-row slabs through registers with two barriers
// per slab and ran at ~270 TFLOP/s on these calls.  Here the same 128 x (64 | 128) output tile gets
//   * its operand stages global -> LDS by DMA (no staging registers, no ds_write), a ring of D
//     stages of BKT rows, ONE barrier per stage (as igemm_ring_kernel);
//   * 64-row stages in two buffers (64 KiB: two workgroups per CU);
//   * the row-major tiles read transposed with ds_read_b64_tr_b16 (cdna_hip_programming.md T10)
//     from a lane-linear DMA image: the 64-byte granule g of row r sits at position g ^ (r & 3)
//     (256-byte rows) / g ^ ((r >> 1) & 1) (128-byte rows), applied to the DMA's source address
//     and to the reads, so that the four rows a 32-lane half reads hit four different bank groups.
// ---------------------------------------------------------------------------------------------
namespace {
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ bf16x8 tr_frag2(const char* lo_p, const char* hi_p) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)lo_p);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)hi_p);
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int NTJ, int BKT, int D>
__device__ __forceinline__ void wgrad1x1_bf16_ring_body(const WgradArgs& a, const WgradBlock blk) {
  constexpr int BJ = NTJ * 64;                         // output columns per block (64 or 128)
  constexpr int A_RB = 256, G_RB = BJ * 2;             // bytes per staged row
  constexpr int A_BYTES = BKT * A_RB, G_BYTES = BKT * G_RB;
  constexpr int A_LOADS = A_BYTES / 4096;              // a pass = 256 lanes x 16 B = 4 KiB
  constexpr int G_LOADS = G_BYTES / 4096;
  constexpr int PER = A_LOADS + G_LOADS;
  constexpr int KS = BKT / 16;
  static_assert(A_BYTES % 4096 == 0 && G_BYTES % 4096 == 0 && D >= 2 && D <= 4, "stage geometry");
  __shared__ __attribute__((aligned(1024))) char smem[D * (A_BYTES + G_BYTES)];
  char* const smemG = smem + D * A_BYTES;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int itiles = (a.I + 127) / 128;
  const int i0 = (blk.x % itiles) * 128;
  const int j0 = blk.y * BJ;
  const int mbeg = blk.z * a.rows_per_split;
  const int cnt = (min(a.rows_per_split, a.M - mbeg) + BKT - 1) / BKT;

  // DMA lanes: A pass = 16 rows x 16 chunks; G pass = 16 rows x 16 chunks (BJ = 128) or 32 rows x
  // 8 chunks (BJ = 64).  The lane at LDS position p of row r fetches source chunk p ^ swz(r).
  const int arow = tid >> 4, apos = tid & 15;
  const int achunk = apos ^ ((arow & 3) << 2);
  constexpr int GCPR = G_RB / 16;                      // chunks per dC row (16 or 8)
  const int grow = tid / GCPR, gpos = tid % GCPR;
  const int gchunk = GCPR == 16 ? gpos ^ ((grow & 3) << 2) : gpos ^ (((grow >> 1) & 1) << 2);
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc_b((const char*)a.A + (size_t)a.a_off * 2,
                                                 (a.a_rows * a.lda - a.a_off) * 2);
  const __amdgpu_buffer_rsrc_t rsG = make_rsrc_b((const char*)a.G + (size_t)a.g_off * 2,
                                                 ((long long)a.M * a.ldg - a.g_off) * 2);
  // (columns beyond I / J are clamped: their products land in dW rows / columns never stored)
  const unsigned aoff = (unsigned)(arow * a.lda + min(i0 + achunk * 8, a.I - 8)) * 2u;
  const unsigned goff = (unsigned)(grow * a.ldg + min(j0 + gchunk * 8, a.J - 8)) * 2u;
  constexpr int A_PASS_ROWS = 16, G_PASS_ROWS = 256 / GCPR;

  f32x16 acc[2][NTJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NTJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  unsigned tile_bits = 0;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NTJ; ++j)
      if ((i0 + wm * 64 + i * 32 < a.I) && (j0 + (wn * NTJ + j) * 32 < a.J))
        tile_bits |= 1u << (i * NTJ + j);
  tile_bits = __builtin_amdgcn_readfirstlane(tile_bits);

  // stage `st` (rows mbeg + st * BKT ...) into ring slot `slot`; rows >= M lie outside the
  // descriptors and come back as zeros (splits end on stage boundaries: no row is counted twice)
#define K_W_ISSUE(SLOT, ST)                                                                  \
  {                                                                                            \
    const int m0 = mbeg + (ST) * BKT;                                                          \
    _Pragma("unroll") for (int p = 0; p < A_LOADS; ++p)                                        \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                \
          rsA, (lds_void_t*)(smem + (SLOT) * A_BYTES + p * 4096 + wave * 1024), 16, (int)aoff, \
          (m0 + p * A_PASS_ROWS) * a.lda * 2, 0, 0);                                           \
    _Pragma("unroll") for (int p = 0; p < G_LOADS; ++p)                                        \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(                                                \
          rsG, (lds_void_t*)(smemG + (SLOT) * G_BYTES + p * 4096 + wave * 1024), 16, (int)goff, \
          (m0 + p * G_PASS_ROWS) * a.ldg * 2, 0, 0);                                           \
  }
#pragma unroll
  for (int d = 0; d < D - 1; ++d)
    if (d < cnt) K_W_ISSUE(d, d)

  // transposed-read addresses: k rows 16 s + 8 lh + tq (+4), 64-byte granule of the tile's columns
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
  int a_lo[2], g_lo[NTJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
    a_lo[i] = (8 * lh + tq) * A_RB + (((wm * 2 + i) ^ tq) << 6) + 32 * tg + 8 * tp;
#pragma unroll
  for (int j = 0; j < NTJ; ++j)
    g_lo[j] = (8 * lh + tq) * G_RB +
              (((wn * NTJ + j) ^ (GCPR == 16 ? tq : (tq >> 1))) << 6) + 32 * tg + 8 * tp;

  int slot = 0, slot_in = D - 1;
  for (int it = 0; it < cnt; ++it) {
    const int ahead = min(cnt, it + D - 1) - it - 1;
    if (D > 3 && ahead >= 2) wait_vmcnt<(D > 3 ? 2 : 0) * PER>();
    else if (D > 2 && ahead >= 1) wait_vmcnt<(D > 2 ? 1 : 0) * PER>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (it + D - 1 < cnt) K_W_ISSUE(slot_in, it + D - 1)
    const char* const As = smem + slot * A_BYTES;
    const char* const Gs = smemG + slot * G_BYTES;
    bf16x8 af[2][KS], bf[NTJ][KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
        af[i][s] = tr_frag2(As + s * 16 * A_RB + a_lo[i], As + (s * 16 + 4) * A_RB + a_lo[i]);
#pragma unroll
      for (int j = 0; j < NTJ; ++j)
        bf[j][s] = tr_frag2(Gs + s * 16 * G_RB + g_lo[j], Gs + (s * 16 + 4) * G_RB + g_lo[j]);
    }
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NTJ; ++j)
        if ((tile_bits >> (i * NTJ + j)) & 1u) {
#pragma unroll
          for (int s = 0; s < KS; ++s)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
        }
    __builtin_amdgcn_s_setprio(0);
    if (++slot == D) slot = 0;
    if (++slot_in == D) slot_in = 0;
  }
#undef K_W_ISSUE

  // split-K result: fp32 atomics into dW, or (part_stride > 0) plain stores into this split's slab
  float* dw = a.dW + (size_t)blk.z * a.part_stride;
  const bool part = a.part_stride > 0;
#pragma unroll
  for (int j = 0; j < NTJ; ++j) {
    const int jj = j0 + (wn * NTJ + j) * 32 + li;
    if (jj >= a.J) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ii = i0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (ii >= a.I) continue;
        if (part) dw[(size_t)ii * a.J + jj] = acc[i][j][r];
        else atomicAdd(dw + (size_t)ii * a.J + jj, acc[i][j][r]);
      }
  }
}

#define K_WRING_BOUNDS(NTJ, BKT, D)                                                            \
  __launch_bounds__(256, (160 * 1024) / (D * BKT * (128 + NTJ * 64) * 2) >= 3                      \
                             ? 3 : ((160 * 1024) / (D * BKT * (128 + NTJ * 64) * 2) >= 2 ? 2 : 1))
template <int NTJ, int BKT, int D>
__global__ K_WRING_BOUNDS(NTJ, BKT, D) void wgrad1x1_bf16_ring_kernel(WgradArgs a) {
  wgrad1x1_bf16_ring_body<NTJ, BKT, D>(a, wgrad_block(a));
}
// several filter gradients of one input in one launch (see WgradGroupArgs)
template <int NTJ, int BKT, int D>
__global__ K_WRING_BOUNDS(NTJ, BKT, D) void wgrad1x1_bf16_ring_group_kernel(WgradGroupArgs g) {
  int p;
  const WgradBlock blk = wgrad_group_block(g, &p);
  wgrad1x1_bf16_ring_body<NTJ, BKT, D>(g.a[p], blk);
}
#undef K_WRING_BOUNDS

struct WgradTune { int bk, d, slots, off; };
const WgradTune& wgrad_tune() {
  static const WgradTune t = [] {
    WgradTune r = {0, 0, 0, 0};
    if (c2d_tune_on()) {
      if (const char* e = c2d_tune_get("wring_bk")) r.bk = atoi(e);
      if (const char* e = c2d_tune_get("wring_d")) r.d = atoi(e);
      if (const char* e = c2d_tune_get("wring_slots")) r.slots = atoi(e);
      if (const char* e = c2d_tune_get("wring")) r.off = e[0] == '0';
    }
    return r;
  }();
  return t;
}
}  // namespace

int launch_wgrad1x1_bf16_ring(WgradArgs a, hipStream_t s, int* splits_out, bool splits_only) {
  const WgradTune& t = wgrad_tune();
  if (t.off) return C2D_ERR_UNSUPPORTED;
  if (a.I % 8 != 0 || a.J % 8 != 0 || a.I < 8 || a.J < 8 || a.lda % 8 != 0 || a.ldg % 8 != 0 ||
      a.a_off % 8 != 0 || a.g_off % 8 != 0)
    return C2D_ERR_UNSUPPORTED;
  // Measured per call of the step (tools/sweep_wgrad.sh, N = 2000): these launches are bound by
  // their split-K atomics as much as by staging — every workgroup adds a whole 128 x 128 fp32 tile
  // (16.8 MB per 256 workgroups at the chip-wide 1.3 TB/s of float atomics, all at the end of the
  // launch) — so the DMA ring only pays where the tile count keeps the splits low: output widths
  // of 129..256 columns (576->192: 68 -> 49 us, 1024->192: 40 -> 33, 1024->160: 39 -> 30);
  // 128-wide layers stay on the two-K-group kernel of conv_gemm.hip (30 us); the 352-wide ones
  // measured equal with warm operands (49 us either way) and 78 -> 65 us with cold ones
  // (bench_cold=1: what the step sees), so they take the ring as well.
  if (!t.bk && !t.d && !t.slots && !(a.J > 128 && a.J <= 384)) return C2D_ERR_UNSUPPORTED;
  const bool narrow = a.J % 128 != 0 && a.J % 128 <= 64;    // 128 x 64 block tiles
  const int bj = narrow ? 64 : 128;
  const int bk = t.bk == 32 ? 32 : 64;
  const int d = t.d >= 2 && t.d <= 4 ? t.d : 2;
  a.tiles_x = c2d_ceil_div(a.I, 128);
  a.tiles_y = c2d_ceil_div(a.J, bj);
  const int tiles = a.tiles_x * a.tiles_y;
  // two workgroups per CU in one round; at least 4 stages per workgroup
  int splits = (t.slots > 0 ? t.slots : c2d_cu_scaled(512)) / tiles;
  const int max_splits = c2d_ceil_div(a.M, 4 * bk);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  a.rows_per_split = c2d_ceil_div(c2d_ceil_div(a.M, splits), bk) * bk;
  a.nsplits = c2d_ceil_div(a.M, a.rows_per_split);
  if (splits_out) *splits_out = a.nsplits;
  if (splits_only) return C2D_OK;
  const dim3 grid(tiles * a.nsplits), block(256);
  dispatch_note_ext("wgrad1x1_bf16_ring_kernel<%d, %d, %d>", narrow ? 1 : 2, bk, d);
#define K_WR(NTJ_, BK_, D_)                                                                   \
  if ((narrow ? 1 : 2) == NTJ_ && bk == BK_ && d == D_) {                                      \
    hipLaunchKernelGGL((wgrad1x1_bf16_ring_kernel<NTJ_, BK_, D_>), grid, block, 0, s, a);      \
    return c2d_launch_status();                                                                \
  }
  K_WR(1, 64, 2) K_WR(2, 64, 2) K_WR(1, 32, 3) K_WR(2, 32, 3)
#undef K_WR
  return C2D_ERR_UNSUPPORTED;
}


// Grouped form (c2d_conv1x1_wgrad_multi_bf16): `num` 1x1 / stride-1 filter gradients that share
// the activations x; 128-column tiles for every output, ONE split count from the tiles of all.
int launch_wgrad1x1_bf16_ring_group(WgradArgs* a, int num, hipStream_t s) {
  if (num < 1 || num > WGRAD_GROUP_MAX) return C2D_ERR_UNSUPPORTED;
  WgradGroupArgs g;
  int tiles = 0;
  for (int p = 0; p < num; ++p) {
    if (a[p].I % 8 != 0 || a[p].J % 8 != 0 || a[p].I < 8 || a[p].J < 8 || a[p].lda % 8 != 0 ||
        a[p].ldg % 8 != 0 || a[p].a_off % 8 != 0 || a[p].g_off % 8 != 0 || a[p].M != a[0].M)
      return C2D_ERR_UNSUPPORTED;
    a[p].tiles_x = c2d_ceil_div(a[p].I, 128);
    a[p].tiles_y = c2d_ceil_div(a[p].J, 128);
    g.first_tile[p] = tiles;
    tiles += a[p].tiles_x * a[p].tiles_y;
  }
  for (int p = num; p <= WGRAD_GROUP_MAX; ++p) g.first_tile[p] = tiles;
  const WgradTune& t = wgrad_tune();
  int splits = (t.slots > 0 ? t.slots : c2d_cu_scaled(512)) / tiles;       // two workgroups per CU in one round
  const int max_splits = c2d_ceil_div(a[0].M, 4 * 64);      // at least 4 stages per workgroup
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  const int rps = c2d_ceil_div(c2d_ceil_div(a[0].M, splits), 64) * 64;
  g.nsplits = c2d_ceil_div(a[0].M, rps);
  g.num = num;
  for (int p = 0; p < num; ++p) {
    a[p].rows_per_split = rps; a[p].nsplits = g.nsplits; a[p].part_stride = 0;
    a[p].a_rows = a[p].M;
    g.a[p] = a[p];
  }
  dispatch_note_ext("wgrad1x1_bf16_ring_group_kernel<2, 64, 2>");
  hipLaunchKernelGGL((wgrad1x1_bf16_ring_group_kernel<2, 64, 2>), dim3(tiles * g.nsplits), dim3(256), 0, s, g);
  return c2d_launch_status();
}

}  // namespace c2d_ig

#ifdef C2D_RING_TRACE
// Diagnostic build only (make EXTRA_CXXFLAGS=-DC2D_RING_TRACE; tools/trace_ring.py): 16 x u64 per
// block — start, prologue, K loop, epilogue, time at wait + barrier, time issuing DMA (cycles of
// wave 0), stages, s_memrealtime at the end.
namespace c2d_ig { unsigned long long* ring_trace_buffer() { return g_ring_trace; } }
extern "C" int c2d_debug_set_ring_trace(void* buf) {
  c2d_ig::g_ring_trace = (unsigned long long*)buf;
  return C2D_OK;
}
#endif
