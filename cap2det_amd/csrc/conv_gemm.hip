// Implicit-GEMM convolution on the fp32 MFMA pipe of gfx950 (v_mfma_f32_32x32x2_f32).
//
// Replaces slim.conv2d (+ folded inference BatchNorm + ReLU) of the Inception-V2 extractor
// called at models/utils.py:133-136,165-167 and slim.fully_connected at
// models/cap2det_model.py:79-88,191-197 of the reference, plus their TF gradients
// (Conv2DBackpropInput / Conv2DBackpropFilter, train/trainer.py:141-146).
//
// Two kernel families, both NHWC with "rows" = n*h*w and channels contiguous:
//   igemm_nt  C[m][n] (+)= epi( sum_{tap} sum_{k<K} A[src(m,tap)][k] * Bt[tap][n][k] )
//             forward  : A = x,  Bt = W^T per tap ([tap][cout][cin]),  src = input pixel of tap
//             dgrad    : A = dC, Bt = W (HWIO is [tap][cin][cout] = [tap][n][k]), src = the
//                        output pixel that used this input pixel through `tap` (or none)
//   wgrad_tn  dW[tap][i][j] += sum_m A[src(m,tap)][i] * G[m][j]      (split over m, fp32 atomics)
//
// MFMA mapping (cdna_hip_programming.md §3): one wave owns MT x NT tiles of 32x32; lane
// l = (h = l>>5, i = l&31) feeds A[i][k], B[k][i]; a BK = 32 slab is consumed as 16 MFMA steps
// where half-wave h takes k = 16h + s (a permutation of the k order common to A and B, which
// lets each lane fetch its 16 k-values with four ds_read_b128).  fp32 in, fp32 accumulate:
// bit-for-bit an fmaf chain, no reduced precision anywhere.
#include "c2d_common.h"
#include <stdio.h>
#include <stdlib.h>

namespace {

// Dispatch record of the calling thread's LAST convolution entry point (c2d_debug_last_dispatch):
// which kernel template instances it launched, spelled the way rocprofv3 prints them, so a parity
// test can prove which tile path produced the numbers it compared.  A few integer stores per
// launch; formatted only when queried.
struct DispatchRec {
  const char* fmt;
  int p[8];
};
constexpr int DISPATCH_MAX = 8;
thread_local DispatchRec g_dispatch[DISPATCH_MAX];
thread_local int g_ndispatch = 0;
inline void dispatch_reset() { g_ndispatch = 0; }
inline void dispatch_note(const char* fmt, int a = 0, int b = 0, int c = 0, int d = 0, int e = 0,
                          int f = 0, int g = 0, int h = 0) {
  if (g_ndispatch >= DISPATCH_MAX) return;
  DispatchRec& r = g_dispatch[g_ndispatch++];
  r.fmt = fmt;
  r.p[0] = a; r.p[1] = b; r.p[2] = c; r.p[3] = d; r.p[4] = e; r.p[5] = f; r.p[6] = g; r.p[7] = h;
}

}  // namespace
#include "igemm_common.h"
using namespace c2d_ig;
namespace {
#ifdef C2D_TRACE
unsigned long long* g_trace = nullptr;
#endif


struct SkPlan {
  int enabled;
  int period;                       // block rows after which the tile costs repeat
  int cost[SK_MAX_PERIOD];          // slab iterations of a tile in block row r (mod period)
  int prefix[SK_MAX_PERIOD + 1];    // prefix[r] = sum_{q<r} cost[q]; prefix[period] = period sum
  int total;                        // all iterations of the launch
  int share;                        // ceil(total / grid)
  float* partials;                  // [grid][2][BM*BN]: slot 0 = piece that starts inside a
                                    // tile, slot 1 = piece that starts a tile it does not finish
  int* counters;                    // [tiles]: zero on entry, re-zeroed by the reducing piece
};

struct IgemmSkArgs {
  IgemmArgs a = {};
  SkPlan sk;
};

// Second __launch_bounds__ argument = waves per SIMD the register allocator must leave room
// for: the 128x128 config sits right at the 168-register step (3 waves/SIMD); losing it cost
// 25-35 % on the layers with three n-tiles.
// ES = operand element size: 4 = fp32 operands on v_mfma_f32_32x32x2_f32 (exact fp32), 2 = bf16
// operands / bf16 output on v_mfma_f32_32x32x16_bf16 with fp32 accumulation (BASELINE
// configs[2] / [4]: "bf16 storage / fp32 accumulate").  Both stage 128-B row slabs (32 floats or
// 64 bf16) with the same 16-B loads, LDS image and tile masks; only the fragment reads, the MFMA
// and the epilogue's store width differ.
template <int MODE, int WM, int WN, int MT, int NT, int BKT, bool PM, bool SK, int ES = 4>
__device__ __forceinline__ void igemm_body(const IgemmArgs& a, const SkPlan& sk) {
  // BKT = elements of K per slab (one 128-B line per row: 32 floats / 64 bf16).
  static_assert(BKT * ES == 128, "a slab row is one 128-B line");
  constexpr int LDS_STRIDE = 32 + 4;            // floats (144 B) per staged row
  constexpr int EPL = 16 / ES;                  // elements per 16-B lane load
  constexpr int LPR = 8;                        // lanes per row (16 B each)
  constexpr int BM = WM * MT * 32;
  constexpr int BN = WN * NT * 32;
  constexpr int NTHREADS = WM * WN * 64;
  constexpr int ROWS_PER_PASS = NTHREADS / LPR;
  constexpr int A_LOADS = BM / ROWS_PER_PASS;
  constexpr int B_LOADS = BN / ROWS_PER_PASS;
  static_assert(A_LOADS >= 1 && B_LOADS >= 1, "tile too small for the block");
  __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDS_STRIDE];
  float* const As = smem;
  float* const Bs = smem + BM * LDS_STRIDE;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  const int q4 = (tid % LPR) * EPL;  // element offset inside the k slab
  const int q4f = (tid % LPR) * 4;   // float (4-B) offset of the lane's 16 B inside a staged row
#ifdef C2D_TRACE
  const unsigned long long tr_t0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long tr_c0 = __builtin_amdgcn_s_memtime();
  int tr_iters = 0;
  unsigned long long tr_seg[5] = {0, 0, 0, 0, 0}, tr_a, tr_b;
#define K_STAMP(v)                                                                   \
  __builtin_amdgcn_sched_barrier(0);                                                   \
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v)::"memory");            \
  __builtin_amdgcn_sched_barrier(0);
#else
#define K_STAMP(v)
#endif

  // ---- work range of this workgroup ---------------------------------------------------------
  const int lb = xcd_remap(blockIdx.x, gridDim.x);   // logical block: consecutive ones share an XCD
  int mt, nt, s0 = 0, remaining = 1;
  if (SK) {
    const int lo = min(lb * sk.share, sk.total);
    remaining = min(lo + sk.share, sk.total) - lo;
    const int perlen = a.n_tiles * sk.prefix[sk.period];
    const int per = lo / perlen;
    int rem = lo - per * perlen;
    int r = 0;
    while (r + 1 < sk.period && a.n_tiles * sk.prefix[r + 1] <= rem) ++r;
    rem -= a.n_tiles * sk.prefix[r];
    nt = rem / sk.cost[r];
    s0 = rem - nt * sk.cost[r];
    mt = per * sk.period + r;
  } else {
    block_tile(a.g, a.n_tiles, blockIdx.x, gridDim.x, &mt, &nt);
  }

  const int ntaps = a.g.nky * a.g.nkx;
  const int kslabs = (a.K + BKT - 1) / BKT;
  const size_t tap_stride = (size_t)a.N * a.K;

  while (remaining > 0) {
    const int m0 = mt * BM, n0 = nt * BN;

    // loader coordinates
    RowPos apos[A_LOADS];
    int arow_l[A_LOADS];
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
      arow_l[i] = (tid / LPR) + i * ROWS_PER_PASS;
      apos[i] = decompose<PM>(m0 + arow_l[i], a.M, a.g);
    }
    int brow_l[B_LOADS], brow_off[B_LOADS];
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
      brow_l[i] = (tid / LPR) + i * ROWS_PER_PASS;
      brow_off[i] = min(n0 + brow_l[i], a.N - 1);   // columns >= N are never stored: clamp only
    }
    // Operands are read with raw buffer loads: the address is descriptor base (SGPRs) + a per-lane
    // byte offset that changes only with the tap + a scalar byte offset that walks K, so a slab
    // costs NO vector ALU work for addresses, and SAME-padding rows / lanes past the end of K
    // simply carry an out-of-range offset for which the hardware returns 0.0 (no v_cndmask).
    // That matters: the SIMD issues vector ALU instructions of co-resident waves very slowly
    // while fp32 MFMAs (64 cycles each) occupy it — the address + mask work of the flat-load
    // form took as long as the MFMAs themselves (per-phase stamps, tools/trace_igemm.py).
    int lda = a.lda, Kc = a.K, sgi = 0;   // current segment (wave-uniform)
    __amdgpu_buffer_rsrc_t rsA = make_rsrc_b((const char*)a.A + (size_t)a.a_off * ES,
                                             (a.a_rows * a.lda - a.a_off) * ES);
    __amdgpu_buffer_rsrc_t rsB = make_rsrc_b(
        a.Bt, a.mo_n ? a.mo_bbytes
                     : (a.nseg > 1 ? (long long)a.N * a.K : (long long)a.g.kh * a.g.kw * a.N * a.K) * ES);
    int brow_base[B_LOADS];            // element offset of the staged weight row inside a tap's plane
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i)
      brow_base[i] = a.mo_n ? mo_weight_row(a, brow_off[i]) : brow_off[i] * a.K;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // bit i*NT+j: 32x32 tile (i,j) of this wave lies inside M x N (scalar: all inputs uniform)
    unsigned tile_bits = 0;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        if (tile_has_rows<PM>(m0 + (wm * MT + i) * 32, a.M, a.g) && (n0 + (wn * NT + j) * 32 < a.N))
          tile_bits |= 1u << (i * NT + j);
    tile_bits = __builtin_amdgcn_readfirstlane(tile_bits);

    // Taps this tile iterates over (bit t <-> ky = ky0 + kstep*(t / nkx), kx = kx0 + kstep*(t %
    // nkx)).  PM: a tap that is SAME padding for every 32-row tile of the block is not visited at
    // all (no loads, no barriers), and within a visited tap each wave skips the MFMAs of its
    // padding tiles.
    // Tap table, one tap per lane (every wave holds all of it): the row delta of the activation
    // rows (the source row of a tap is row_base + delta wherever the tap reads a real pixel), the
    // offset of the tap's weight plane and (PM) which 32-row tiles of the block are real for the
    // tap are computed ONCE per tile here, so that a tap change in the K loop is three v_readlane
    // and a few VALU per staged row instead of re-deriving every source row.
    int tab_delta = 0, tab_toff = 0;
    unsigned tab_tv = 0;
    if (lane < ntaps) {
      const int ty_ = lane / a.g.nkx;
      const int ky = a.g.ky0 + a.g.kstep * ty_, kx = a.g.kx0 + a.g.kstep * (lane - ty_ * a.g.nkx);
      tab_toff = (ky * a.g.kw + kx) * (int)tap_stride;
      if (MODE == 0) {
        tab_delta = (ky - a.g.pad_t) * a.g.iw + (kx - a.g.pad_l);
      } else {
        const int sh = a.g.stride - 1;   // (stride-2 launches hold the taps of ONE parity class)
        tab_delta = ((a.g.y0 + a.g.pad_t - ky) >> sh) * a.g.ow + ((a.g.x0 + a.g.pad_l - kx) >> sh);
      }
      tab_tv = 0xffu;
      if (PM) {
        tab_tv = 0;
        const int hw = a.g.rh * a.g.rw;
#pragma unroll
        for (int tb = 0; tb < BM / 32; ++tb) {
          const unsigned t = (unsigned)(m0 + tb * 32) >> a.g.pm;
          const unsigned grp = (unsigned)(((unsigned long long)t * a.g.magic_hw) >> 40);
          const unsigned px = t - grp * (unsigned)hw;
          const int y = (int)(((unsigned long long)px * a.g.magic_w) >> 40);
          const int x = (int)px - y * a.g.rw;
          tab_tv |= (tap_ok<MODE>(a.g, y, x, ky, kx) ? 1u : 0u) << tb;
        }
      }
    }
    // Taps this tile iterates over (bit t <-> lane t of the table).  PM: a tap that is SAME
    // padding for every 32-row tile of the block is not visited at all (no loads, no barriers),
    // and within a visited tap each wave skips the MFMAs of its padding tiles.
    const unsigned long long tapmask = __ballot(lane < ntaps && tab_tv != 0);
    // per staged activation row: its base row in the operand and the taps that are real for it
    int row_base[A_LOADS];
    unsigned long long amask[A_LOADS];
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
      row_base[i] = MODE == 0
                        ? (apos[i].img * a.g.ih + apos[i].y * a.g.stride) * a.g.iw + apos[i].x * a.g.stride
                        : (apos[i].img * a.g.oh + apos[i].y) * a.g.ow + apos[i].x;
      amask[i] = 0;
      for (int ty_ = 0, tp = 0; ty_ < a.g.nky; ++ty_)
        for (int tx_ = 0; tx_ < a.g.nkx; ++tx_, ++tp)
          if (src_row<MODE>(a.g, apos[i], a.g.ky0 + a.g.kstep * ty_, a.g.kx0 + a.g.kstep * tx_) >= 0)
            amask[i] |= 1ull << tp;
    }
    int cost = __builtin_popcountll(tapmask) * kslabs;   // slab iterations of the whole tile
    if (a.nseg > 1) cost = a.total_slabs;
    const int cnt = SK ? min(cost - s0, remaining) : cost;   // ... of this piece
    const bool whole = !SK || cnt == cost;

    f32x4 ra[A_LOADS], rb[B_LOADS];
    int tap = 0, kc = 0;   // wave-uniform slab cursor
    unsigned long long taps_left = tapmask;
    // bit i: the tap of the slab being LOADED (tv_load) / being MULTIPLIED (tv_mma, one slab
    // behind) is real for row tile i of this wave
    unsigned tv_load = ~0u, tv_mma = ~0u;
    if (SK && s0 > 0) {       // the piece starts inside the tile: move the cursor to slab s0
      if (a.nseg > 1) {
        int sl = s0;
        while (sl >= (a.segK[sgi] + BKT - 1) / BKT) { sl -= (a.segK[sgi] + BKT - 1) / BKT; ++sgi; }
        kc = sl * BKT;
        lda = a.seg_lda[sgi]; Kc = a.segK[sgi];
        rsA = make_rsrc_b((const char*)a.segA[sgi] + (size_t)a.seg_off[sgi] * ES,
                          (a.a_rows * lda - a.seg_off[sgi]) * ES);
        rsB = make_rsrc_b(a.segB[sgi], (long long)a.N * Kc * ES);
      } else {
        const int tskip = s0 / kslabs;
        kc = (s0 - tskip * kslabs) * BKT;
        for (int q = 0; q < tskip; ++q) taps_left &= taps_left - 1ull;
      }
    }

    // Row pointers of the current tap / segment.  They change only when the cursor moves to the
    // next tap (every K/BKT slabs; never for a 1x1 convolution), so the per-slab address work is
    // one add per load instead of the whole gather arithmetic.
    unsigned aoff[A_LOADS], boff[B_LOADS];   // per-lane BYTE offsets (operands are < 2 GB)
#define K_RETAP()                                                                            \
  {                                                                                            \
    const int delta = __builtin_amdgcn_readlane(tab_delta, tap);                               \
    _Pragma("unroll") for (int i = 0; i < A_LOADS; ++i)                                        \
        aoff[i] = ((amask[i] >> tap) & 1ull)                                                   \
                      ? (unsigned)((row_base[i] + delta) * lda + q4) * (unsigned)ES            \
                      : OOB_OFFSET;                                                            \
    const int toff = __builtin_amdgcn_readlane(tab_toff, tap);                                 \
    _Pragma("unroll") for (int i = 0; i < B_LOADS; ++i)                                        \
        boff[i] = (unsigned)((a.nseg > 1 ? brow_off[i] * Kc : brow_base[i]) + toff + q4) *    \
                  (unsigned)ES;                                                                \
    if (PM)                                                                                    \
      tv_load = ((unsigned)__builtin_amdgcn_readlane((int)tab_tv, tap) >> (wm * MT)) &         \
                ((1u << MT) - 1u);                                                             \
  }
#define K_TAP_FROM_MASK() tap = taps_left ? __builtin_ctzll(taps_left) : 0;
#define K_ISSUE()                                                                            \
  {                                                                                            \
    const int soff = kc * ES;                                                                  \
    if (kc + BKT <= Kc) {      /* whole slab inside K: offsets as they stand */                \
      _Pragma("unroll") for (int i = 0; i < A_LOADS; ++i) ra[i] = buf_load4(rsA, aoff[i], soff); \
      _Pragma("unroll") for (int i = 0; i < B_LOADS; ++i) rb[i] = buf_load4(rsB, boff[i], soff); \
    } else {                   /* last slab of a K that is not a multiple of BKT */            \
      const bool in = kc + q4 < Kc;                                                            \
      _Pragma("unroll") for (int i = 0; i < A_LOADS; ++i)                                      \
          ra[i] = buf_load4(rsA, in ? aoff[i] : OOB_OFFSET, soff);                             \
      _Pragma("unroll") for (int i = 0; i < B_LOADS; ++i)                                      \
          rb[i] = buf_load4(rsB, in ? boff[i] : OOB_OFFSET, soff);                             \
    }                                                                                          \
  }

    // Prologue: loads of the first slab.  Inside the loop the loads of slab it+1 are issued
    // unconditionally right after the barrier (the last iteration harmlessly re-loads the last
    // slab) so that the loop body is straight-line code and the loads fly under the MFMAs.
    // K is a multiple of 16, not necessarily of BKT: lanes past the end of the last slab re-read
    // the row's last float4 (in bounds) and contribute zeros through the A mask.
    if (cnt > 0) {   // (a stride-2 parity class can have no tap at all: it just stores zeros)
      K_TAP_FROM_MASK();
      K_RETAP();
      K_ISSUE();
    }
    for (int it = 0; it < cnt; ++it) {
      if (SK) {
        // The SIMD arbitrates its matrix pipe by priority, then AGE: with static priorities the
        // oldest of the co-resident workgroups runs at full speed and the youngest starves, so
        // equal shares finish up to 70 % apart (per-block timeline).  Rotating the priority
        // with the slab index gives every co-resident workgroup the same share of the pipe.
        switch (((blockIdx.x >> 8) + it) & 3) {
          case 0: __builtin_amdgcn_s_setprio(0); break;
          case 1: __builtin_amdgcn_s_setprio(1); break;
          case 2: __builtin_amdgcn_s_setprio(2); break;
          default: __builtin_amdgcn_s_setprio(3); break;
        }
      }
      K_STAMP(tr_a)
      tv_mma = tv_load;   // validity of the slab now in registers (to be staged + multiplied)
      // tiles beyond M / N or (PM) in the SAME padding of this tap cost no MFMA time
      unsigned onbits = tile_bits;
      if (PM) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
          if (!((tv_mma >> i) & 1u)) onbits &= ~(((1u << NT) - 1u) << (i * NT));
        onbits = __builtin_amdgcn_readfirstlane(onbits);
      }
#pragma unroll
      for (int i = 0; i < A_LOADS; ++i)
        *reinterpret_cast<f32x4*>(&As[arow_l[i] * LDS_STRIDE + q4f]) = ra[i];
#pragma unroll
      for (int i = 0; i < B_LOADS; ++i)
        *reinterpret_cast<f32x4*>(&Bs[brow_l[i] * LDS_STRIDE + q4f]) = rb[i];
#ifdef C2D_TRACE
      K_STAMP(tr_b) tr_seg[0] += tr_b - tr_a;    // wait for the loads + LDS stores
#endif
      __syncthreads();
#ifdef C2D_TRACE
      K_STAMP(tr_a) tr_seg[1] += tr_a - tr_b;    // barrier 1
#endif
      {
        // advance the wave-uniform cursor (saturating at the last slab)
        if (it + 1 < cnt) {
          kc += BKT;
          if (kc >= Kc) {
            kc = 0;
            if (a.nseg > 1) {          // next (A, Bt) segment of a multi-segment 1x1 GEMM
              ++sgi;
              lda = a.seg_lda[sgi]; Kc = a.segK[sgi];
              rsA = make_rsrc_b((const char*)a.segA[sgi] + (size_t)a.seg_off[sgi] * ES,
                                (a.a_rows * lda - a.seg_off[sgi]) * ES);
              rsB = make_rsrc_b(a.segB[sgi], (long long)a.N * Kc * ES);
            } else {
              taps_left &= taps_left - 1ull;   // next tap that is real for some tile of the block
              K_TAP_FROM_MASK();
            }
            K_RETAP();
          }
        }
        K_ISSUE();
      }
      __builtin_amdgcn_sched_barrier(0);
#ifdef C2D_TRACE
      K_STAMP(tr_b) tr_seg[2] += tr_b - tr_a;    // cursor + issue of the next slab's loads
#endif

      if constexpr (ES == 4) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {   // (32 floats per slab row)
        float af[MT][8], bf[NT][8];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const float* p = &As[(wm * MT * 32 + i * 32 + li) * LDS_STRIDE + lh * 16 + half * 8];
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(p);
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(p + 4);
          af[i][0] = v0.x; af[i][1] = v0.y; af[i][2] = v0.z; af[i][3] = v0.w;
          af[i][4] = v1.x; af[i][5] = v1.y; af[i][6] = v1.z; af[i][7] = v1.w;
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const float* p = &Bs[(wn * NT * 32 + j * 32 + li) * LDS_STRIDE + lh * 16 + half * 8];
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(p);
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(p + 4);
          bf[j][0] = v0.x; bf[j][1] = v0.y; bf[j][2] = v0.z; bf[j][3] = v0.w;
          bf[j][4] = v1.x; bf[j][5] = v1.y; bf[j][6] = v1.z; bf[j][7] = v1.w;
        }
        // one scalar branch per 32x32 tile (8 chained MFMAs each: the dependent-accumulator
        // latency of v_mfma_f32_32x32x2_f32 equals its issue interval, so a chain does not stall)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            if ((onbits >> (i * NT + j)) & 1u) {
#pragma unroll
              for (int s = 0; s < 8; ++s)
                acc[i][j] =
                    __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
            }
      }
      } else {
        // bf16: lane (li, lh) feeds k = 16*st + 8*lh .. +8 of row li to each 32x32x16 MFMA:
        // one ds_read_b128 per fragment; all fragments of the slab first, then one scalar
        // branch per 32x32 tile around its 4 chained MFMAs.
        bf16x8 af[MT][4], bf[NT][4];
#pragma unroll
        for (int st = 0; st < 4; ++st) {
#pragma unroll
          for (int i = 0; i < MT; ++i)
            af[i][st] = *reinterpret_cast<const bf16x8*>(
                &As[(wm * MT * 32 + i * 32 + li) * LDS_STRIDE + st * 8 + lh * 4]);
#pragma unroll
          for (int j = 0; j < NT; ++j)
            bf[j][st] = *reinterpret_cast<const bf16x8*>(
                &Bs[(wn * NT * 32 + j * 32 + li) * LDS_STRIDE + st * 8 + lh * 4]);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            if ((onbits >> (i * NT + j)) & 1u) {
#pragma unroll
              for (int st = 0; st < 4; ++st)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][st], bf[j][st], acc[i][j],
                                                                    0, 0, 0);
            }
      }
#ifdef C2D_TRACE
      K_STAMP(tr_a) tr_seg[3] += tr_a - tr_b;    // fragment reads + MFMA issue
#endif
      __syncthreads();
#ifdef C2D_TRACE
      K_STAMP(tr_b) tr_seg[4] += tr_b - tr_a;    // barrier 2
#endif
    }
#undef K_RETAP
#undef K_ISSUE
#undef K_TAP_FROM_MASK
#ifdef C2D_TRACE
    tr_iters += cnt;
#endif

    bool finish = true;   // does this workgroup run the tile's epilogue?
    if (SK && !whole) {
      // ---- a piece of a tile: publish the accumulators, the last arriver reduces --------------
      const int per = mt / sk.period, r = mt - per * sk.period;
      const int tbeg = a.n_tiles * (per * sk.prefix[sk.period] + sk.prefix[r]) + nt * sk.cost[r];
      const int first_b = tbeg / sk.share;
      const int pieces = (tbeg + cost - 1) / sk.share - first_b + 1;
      float* mine = sk.partials + ((size_t)lb * 2 + (s0 > 0 ? 0 : 1)) * (size_t)(BM * BN);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int rq = 0; rq < 4; ++rq) {
            f32x4 v = {acc[i][j][4 * rq], acc[i][j][4 * rq + 1], acc[i][j][4 * rq + 2],
                       acc[i][j][4 * rq + 3]};
            *reinterpret_cast<f32x4*>(mine + ((size_t)(((i * NT + j) * 4 + rq) * NTHREADS + tid)) * 4) = v;
          }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains its stores
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        int* cntp = sk.counters + (mt * a.n_tiles + nt);
        const int old = __hip_atomic_fetch_add(cntp, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old == pieces - 1;
        if (last) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __hip_atomic_store(cntp, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next launch
        }
        reinterpret_cast<volatile int*>(smem)[0] = last;
      }
      __syncthreads();
      finish = reinterpret_cast<volatile int*>(smem)[0] != 0;
      __syncthreads();   // smem is reused below / by the next piece
      if (finish) {
        // sum every piece's slab in piece order (own one included: fixed order, reproducible)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r2 = 0; r2 < 16; ++r2) acc[i][j][r2] = 0.0f;
        for (int pc = 0; pc < pieces; ++pc) {
          const float* src = sk.partials + ((size_t)(first_b + pc) * 2 + (pc == 0 ? 1 : 0)) * (size_t)(BM * BN);
#pragma unroll
          for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
              for (int rq = 0; rq < 4; ++rq) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(
                    src + ((size_t)(((i * NT + j) * 4 + rq) * NTHREADS + tid)) * 4);
                acc[i][j][4 * rq] += v.x; acc[i][j][4 * rq + 1] += v.y;
                acc[i][j][4 * rq + 2] += v.z; acc[i][j][4 * rq + 3] += v.w;
              }
        }
      }
    }

    if (finish) {
      // Epilogue.  C/D map of a 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5),
      // i.e. 4 B per lane per store.  Each wave transposes 32-row strips through its private
      // slice of the (now idle) staging LDS so that the global stores are 16 B per lane on
      // contiguous 128*NT-byte row segments (4x fewer store instructions, T21).
      constexpr int SCOLS = NT * 32;
      constexpr int SSTR = SCOLS + 4;
      static_assert(WM * WN * 32 * SSTR <= (BM + BN) * LDS_STRIDE, "epilogue staging exceeds LDS");
      float* stage = smem + wave * (32 * SSTR);   // (the K loop ended with a block barrier)
      constexpr int C4 = SCOLS / 4;          // float4 per strip row
      constexpr int RPP = 64 / C4;           // rows per pass over the strip
      const int ec4 = lane % C4, er = lane / C4;
      const int ncol = n0 + wn * SCOLS + ec4 * 4;
      f32x4 esc = {1.f, 1.f, 1.f, 1.f}, esh = {0.f, 0.f, 0.f, 0.f};
      const bool ncol_ok = ncol < a.N;       // N is a multiple of 4
      if (ncol_ok && a.scale) esc = *reinterpret_cast<const f32x4*>(a.scale + ncol);
      if (ncol_ok && a.shift) esh = *reinterpret_cast<const f32x4*>(a.shift + ncol);
      // several convolutions in one GEMM: this lane's four columns belong to one of them
      float* oC = a.C; int oldc = a.ldc, ocoff = a.c_off + ncol, orelu = a.relu;
      if (MODE == 0 && a.mo_n && ncol_ok) {
        const MoOut o = mo_output(a, ncol);
        oC = o.C; oldc = o.ldc; ocoff = o.coff + (ncol - o.lo); orelu = o.relu;
        esc = *reinterpret_cast<const f32x4*>(o.scale + (ncol - o.lo));
        esh = *reinterpret_cast<const f32x4*>(o.shift + (ncol - o.lo));
      }
      // fused BN/ReLU backward of the producer layer (see IgemmArgs::fy)
      const bool fused = MODE == 1 && !SK && a.fy != nullptr;
      f32x4 fsc = {0.f, 0.f, 0.f, 0.f}, fbe = fsc, fig = fsc, fsb = fsc, fsg = fsc;
      bool fpass = false;      // columns of a pooling branch: plain gradient
      if (fused && ncol_ok) fpass = fused_bn_params(a, ncol, fsc, fbe, fig);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            stage[((r & 3) + 8 * (r >> 2) + 4 * lh) * SSTR + j * 32 + li] = acc[i][j][r];
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): this wave's LDS writes have landed
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int pass = 0; pass < 32 / RPP; ++pass) {
          const int row = pass * RPP + er;
          const int m = m0 + (wm * MT + i) * 32 + row;
          f32x4 v = *reinterpret_cast<const f32x4*>(&stage[row * SSTR + ec4 * 4]);
          bool row_ok = m < a.M;
          int drow = m;
          if (PM || (MODE == 1 && a.g.sub > 1)) {   // permuted rows / rows of a parity class
            const RowPos p = decompose<PM>(m, a.M, a.g);
            row_ok = p.valid;
            drow = MODE == 0 ? (p.img * a.g.rh + p.y) * a.g.rw + p.x
                             : (p.img * a.g.ih + p.y * a.g.sub + a.g.y0) * a.g.iw +
                                   p.x * a.g.sub + a.g.x0;
          }
          if (row_ok && ncol_ok) {
            v = v * esc + esh;
            if (orelu) {
              v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            }
            if constexpr (ES == 4) {
              f32x4* dst = reinterpret_cast<f32x4*>(oC + (size_t)drow * oldc + ocoff);
              if (a.accumulate) v += *dst;
              if (fused && !fpass)     // (on the complete gradient: after the accumulation)
                v = fused_bn_item(v, load_act4<ES>(a.fy, (size_t)drow * a.fldy + a.fyoff + ncol),
                                  fsc, fbe, fig, fsb, fsg);
              *dst = v;
            } else {                                  // 4 bf16 = 8 B per lane
              bf16x4* dst = reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(oC) +
                                                      (size_t)drow * oldc + ocoff);
              if (a.accumulate) {
                const bf16x4 o = *dst;
                v.x += (float)o[0]; v.y += (float)o[1]; v.z += (float)o[2]; v.w += (float)o[3];
              }
              if (fused && !fpass)
                v = fused_bn_item(v, load_act4<ES>(a.fy, (size_t)drow * a.fldy + a.fyoff + ncol),
                                  fsc, fbe, fig, fsb, fsg);
              bf16x4 o;
              o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
              *dst = o;
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      if (fused)      // (block-uniform)
        fused_bn_finish<WM, WN, SCOLS, RPP>(smem, nullptr, a.fpart, a.fpart_row0, a.N, fsb, fsg, tid,
                                            wave, ec4, er, er < RPP, n0, mt);
    }

    if (!SK) break;
    __syncthreads();      // the epilogue's LDS strips must be drained before the next piece stages
    remaining -= cnt;
    s0 = 0;
    if (++nt == a.n_tiles) { nt = 0; ++mt; }
  }
#ifdef C2D_TRACE
  if (a.trace && tid == 0) {
    unsigned long long* t = a.trace + (size_t)blockIdx.x * 8;
    t[0] = tr_t0; t[1] = __builtin_amdgcn_s_memrealtime();
    t[2] = tr_c0; t[3] = __builtin_amdgcn_s_memtime();
    t[4] = ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (31 << 11)) << 32) |   // HW_REG_XCC_ID
           (unsigned)__builtin_amdgcn_s_getreg(4 | (31 << 11));                         // HW_REG_HW_ID
    t[5] = (unsigned long long)lb; t[6] = (unsigned long long)tr_iters; t[7] = 0;
  }
  if (a.trace && lane == 0) {   // per-wave phase sums (shader cycles) behind the block records
    unsigned long long* t = a.trace + (size_t)(8192 + blockIdx.x * 4 + wave) * 8;
    for (int q = 0; q < 5; ++q) t[q] = tr_seg[q];
    t[5] = (unsigned long long)tr_iters;
  }
#endif
}

template <int MODE, int WM, int WN, int MT, int NT, int BKT, bool PM, int ES = 4>
__global__ __launch_bounds__(WM * WN * 64, WM == 4 ? 2 : (MT * NT == 2 ? 4 : 3)) void igemm_nt_kernel(IgemmArgs a) {
  SkPlan none;
  igemm_body<MODE, WM, WN, MT, NT, BKT, PM, false, ES>(a, none);
}

template <int MODE, int WM, int WN, int MT, int NT, int BKT, bool PM, int ES = 4>
__global__ __launch_bounds__(WM * WN * 64, WM == 4 ? 2 : 3) void igemm_sk_kernel(IgemmSkArgs p) {
  igemm_body<MODE, WM, WN, MT, NT, BKT, PM, true, ES>(p.a, p.sk);
}

// (The bf16-native implicit GEMM — direct-to-LDS operand stages, ring of 2-3 buffers — lives in
// igemm_bf16.hip: igemm_ring_kernel.)

// ---------------------------------------------------------------------------------------------
// Small-problem kernel (first stage: one image, 32x32 .. 125x125 maps => 1k-16k rows).
// A 64x64-tile launch gives only 16-126 workgroups and every wave owns one 32x32 tile for the
// whole K loop: K*taps/2 dependent MFMAs (15 us for a 3x3x128 conv) whatever the chip size.
// Here a block owns ONE 32x32 output tile and its 4 waves split the K loop 4 ways; each lane
// loads its MFMA operands straight from global memory (lane (i,h) needs A[i][16h..16h+15] = four
// 16-B loads of its own row: no LDS, no barrier in the loop), the next slab is prefetched into a
// second register set, and the 4 partial tiles are summed through LDS before the epilogue.
// ---------------------------------------------------------------------------------------------
// ES = 2 (round 3): bf16 operands / output on v_mfma_f32_32x32x16_bf16 — the bf16 step's first
// stage.  Lane (i, h) feeds its MFMA operand straight from memory as well: 16 bytes are the 8
// elements k = 16 st + 8 h .. of its row, a slab is 64 elements of K (four MFMAs), and the MFMAs
// are 1/16 of the fp32 form's time, so the K loop is a chain of load round trips: FOUR register
// sets in flight instead of two.
template <int MODE, int ES = 4>
__device__ __forceinline__ void igemm_small_body(const IgemmArgs& a, const int tile) {
  __shared__ __attribute__((aligned(16))) float red[4 * 32 * 33];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int m0 = (tile / a.n_tiles) * 32, n0 = (tile % a.n_tiles) * 32;
  const RowPos pos = decompose(m0 + li, a.M, a.g);
  const int ncol = min(n0 + li, a.N - 1);
  constexpr int SLAB = ES == 2 ? 64 : 32;        // elements of K per slab
  const int kslabs = (a.K + SLAB - 1) / SLAB;
  const int total = a.g.nky * a.g.nkx * kslabs;
  const int sbeg = (int)((long long)total * wave / 4), send = (int)((long long)total * (wave + 1) / 4);
  const size_t tap_stride = (size_t)a.N * a.K;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;

  if constexpr (ES == 2) {
    constexpr int SETS = 4;
    bf16x8 ga[SETS][4], gb[SETS][4];
    unsigned okb[SETS];
    const __bf16* const A16 = reinterpret_cast<const __bf16*>(a.A);
    const __bf16* const B16 = reinterpret_cast<const __bf16*>(a.Bt);
#define K_SMB_LOAD(SLABI, SET)                                                               \
  {                                                                                            \
    const int tp = (SLABI) / kslabs;                                                           \
    const int kc = ((SLABI) - tp * kslabs) * 64 + lh * 8;                                      \
    const int ty_ = tp / a.g.nkx, tx_ = tp - ty_ * a.g.nkx;                                    \
    const int ky = a.g.ky0 + ty_ * a.g.kstep, kx = a.g.kx0 + tx_ * a.g.kstep;                  \
    const int sr = src_row<MODE>(a.g, pos, ky, kx);                                            \
    const __bf16* ap = A16 + (size_t)max(sr, 0) * a.lda + a.a_off;                             \
    const __bf16* bp = B16 + (size_t)(ky * a.g.kw + kx) * tap_stride + (size_t)ncol * a.K;     \
    okb[SET] = 0;                                                                              \
    _Pragma("unroll") for (int v = 0; v < 4; ++v) {                                            \
      const int k = kc + 16 * v;                                                               \
      const int kk = min(k, a.K - 8);                                                          \
      okb[SET] |= ((sr >= 0 && k < a.K) ? 1u : 0u) << v;                                       \
      ga[SET][v] = *reinterpret_cast<const bf16x8*>(ap + kk);                                  \
      gb[SET][v] = *reinterpret_cast<const bf16x8*>(bp + kk);                                  \
    }                                                                                          \
  }
#define K_SMB_MMA(SET)                                                                       \
  {                                                                                            \
    _Pragma("unroll") for (int v = 0; v < 4; ++v) {                                            \
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};                                                    \
      const bf16x8 av = ((okb[SET] >> v) & 1u) ? ga[SET][v] : __builtin_bit_cast(bf16x8, z);   \
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, gb[SET][v], acc, 0, 0, 0);             \
    }                                                                                          \
  }
#pragma unroll
    for (int q = 0; q < SETS - 1; ++q)
      if (sbeg + q < send) K_SMB_LOAD(sbeg + q, q);
    for (int sl = sbeg; sl < send; sl += SETS) {
#pragma unroll
      for (int q = 0; q < SETS; ++q) {
        if (sl + q < send) {
          if (sl + q + SETS - 1 < send) K_SMB_LOAD(sl + q + SETS - 1, (q + SETS - 1) % SETS);
          K_SMB_MMA(q);
        }
      }
    }
#undef K_SMB_LOAD
#undef K_SMB_MMA
  } else {
  f32x4 fa[2][4], fb[2][4];
  unsigned okm[2] = {0u, 0u};
  // slab index -> (tap, kc)
#define K_SM_LOAD(SLAB, SET)                                                                 \
  {                                                                                            \
    const int tp = (SLAB) / kslabs;                                                            \
    const int kc = ((SLAB) - tp * kslabs) * 32 + lh * 16;                                      \
    const int ty_ = tp / a.g.nkx, tx_ = tp - ty_ * a.g.nkx;                                    \
    const int ky = a.g.ky0 + ty_ * a.g.kstep, kx = a.g.kx0 + tx_ * a.g.kstep;                  \
    const int sr = src_row<MODE>(a.g, pos, ky, kx);                                            \
    const float* ap = a.A + (size_t)max(sr, 0) * a.lda + a.a_off;                              \
    const float* bp = a.Bt + (size_t)(ky * a.g.kw + kx) * tap_stride + (size_t)ncol * a.K;     \
    okm[SET] = 0;                                                                              \
    _Pragma("unroll") for (int v = 0; v < 4; ++v) {                                            \
      const int k = kc + 4 * v;                                                                \
      const int kk = min(k, a.K - 4);                                                          \
      okm[SET] |= ((sr >= 0 && k < a.K) ? 1u : 0u) << v;                                       \
      fa[SET][v] = *reinterpret_cast<const f32x4*>(ap + kk);                                   \
      fb[SET][v] = *reinterpret_cast<const f32x4*>(bp + kk);                                   \
    }                                                                                          \
  }
#define K_SM_MMA(SET)                                                                        \
  {                                                                                            \
    _Pragma("unroll") for (int v = 0; v < 4; ++v) {                                            \
      const f32x4 av = mask4(fa[SET][v], (okm[SET] >> v) & 1u);                                \
      const f32x4 bv = fb[SET][v];                                                             \
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);                    \
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);                    \
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);                    \
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);                    \
    }                                                                                          \
  }

  if (sbeg < send) K_SM_LOAD(sbeg, 0);
  for (int sl = sbeg; sl < send; sl += 2) {
    if (sl + 1 < send) K_SM_LOAD(sl + 1, 1);
    K_SM_MMA(0);
    if (sl + 1 < send) {
      if (sl + 2 < send) K_SM_LOAD(sl + 2, 0);
      K_SM_MMA(1);
    }
  }
#undef K_SM_LOAD
#undef K_SM_MMA
  }

  // sum the 4 partial tiles; C/D map: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int r = 0; r < 16; ++r)
    red[wave * (32 * 33) + ((r & 3) + 8 * (r >> 2) + 4 * lh) * 33 + li] = acc[r];
  __syncthreads();
  // 256 threads: row = tid/8, 4 consecutive columns at (tid%8)*4
  const int er = tid >> 3, ec = (tid & 7) * 4;
  const int m = m0 + er, n = n0 + ec;
  if (m < a.M && n < a.N) {
    f32x4 v;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int o = er * 33 + ec + c;
      v[c] = red[o] + red[32 * 33 + o] + red[2 * 32 * 33 + o] + red[3 * 32 * 33 + o];
    }
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (a.scale) sc = *reinterpret_cast<const f32x4*>(a.scale + n);
    if (a.shift) sh = *reinterpret_cast<const f32x4*>(a.shift + n);
    v = v * sc + sh;
    if (a.relu) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    int drow = m;
    if (MODE == 1 && a.g.sub > 1) {
      const RowPos p = decompose(m, a.M, a.g);
      drow = (p.img * a.g.ih + p.y * a.g.sub + a.g.y0) * a.g.iw + p.x * a.g.sub + a.g.x0;
    }
    if constexpr (ES == 2) {
      bf16x4* dst = reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(a.C) + (size_t)drow * a.ldc + a.c_off + n);
      if (a.accumulate) {
        const bf16x4 o = *dst;
        v.x += (float)o[0]; v.y += (float)o[1]; v.z += (float)o[2]; v.w += (float)o[3];
      }
      bf16x4 o;
      o[0] = (__bf16)v.x; o[1] = (__bf16)v.y; o[2] = (__bf16)v.z; o[3] = (__bf16)v.w;
      *dst = o;
    } else {
      f32x4* dst = reinterpret_cast<f32x4*>(a.C + (size_t)drow * a.ldc + a.c_off + n);
      if (a.accumulate) v += *dst;
      *dst = v;
    }
  }
}

template <int MODE, int ES = 4>
__global__ __launch_bounds__(256) void igemm_small_kernel(IgemmArgs a) {
  igemm_small_body<MODE, ES>(a, blockIdx.x);
}

// Several INDEPENDENT small problems in one launch (the convolutions of one dependency level of
// an Inception block on the single-image first stage: 3-4 launches of 30-250 workgroups each,
// every one bound by its own K-loop latency, become one launch that fills the chip).
constexpr int SMALL_GROUP_MAX = 8;
struct IgemmGroupArgs {
  IgemmArgs a[SMALL_GROUP_MAX];
  int first[SMALL_GROUP_MAX + 1];   // first workgroup of problem p; first[num] = grid size
  int num;
};
template <int MODE, int ES = 4>
__global__ __launch_bounds__(256) void igemm_small_group_kernel(IgemmGroupArgs g) {
  int p = 0;
  for (int i = 1; i < g.num; ++i)
    if ((int)blockIdx.x >= g.first[i]) p = i;
  igemm_small_body<MODE, ES>(g.a[p], (int)blockIdx.x - g.first[p]);
}

constexpr int WBK = 16;             // rows of M per slab (32: the 128x128 form spills, 5 % slower overall)
constexpr int WG_STRIDE = 128 + 4;  // floats per k-row of the [WBK][128] tiles

// Block tile 128(i) x 128(j); 4 waves as 2x2, each 64x64 (2x2 MFMA tiles); 3-D grid
// (tap * i-tiles, j-tiles, row splits).  108 registers: 4 waves per SIMD.  (An XCD-remapped 1-D
// grid with 32-row slabs and a dynamic deal of the valid tiles to the waves measured 10-25 %
// SLOWER on the 1x1 / stride-2 layers this kernel still serves: it cost the fourth wave.)
// NTJ = 32-column MFMA tiles per wave along j: 2 (block tile 128x128) or 1 (128x64, used when
// the last 128-wide j-tile would be at most half full: J = 192, 160, 320 ...).
// PLAIN = 1x1 / stride 1 (source row = output row): the loader's offsets are slab-invariant.
template <int NTJ, bool PLAIN, int ES>
__device__ __forceinline__ void wgrad_tn_body(const WgradArgs& a, const WgradBlock blk) {
  constexpr int BJ = 2 * NTJ * 32;
  __shared__ __attribute__((aligned(16))) float As[WBK * WG_STRIDE];
  __shared__ __attribute__((aligned(16))) float Gs[WBK * WG_STRIDE];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int itiles = (a.I + 127) / 128;
  const int tap = blk.x / itiles;
  const int i0 = (blk.x - tap * itiles) * 128;
  const int j0 = blk.y * BJ;
  const int ky = tap / a.g.kw, kx = tap - ky * a.g.kw;
  const int mbeg = blk.z * a.rows_per_split;
  const int mend = min(a.M, mbeg + a.rows_per_split);

  // loader: thread -> (k row kr and kr+8, float4 column c4); columns beyond I/J are clamped
  // (their products land in dW rows/cols that are never stored).
  const int kr = tid >> 5;   // 0..7
  const int c4 = (tid & 31) * 4;
  const bool gload = c4 < BJ;   // (BJ = 64: the upper half of each 32-lane row group idles)
  // raw buffer loads (see igemm_body): descriptors end at the operands' last row, masked rows
  // carry an out-of-range offset and come back as zeros
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc_b((const char*)a.A + (size_t)a.a_off * ES,
                                                 (a.a_rows * a.lda - a.a_off) * ES);
  const __amdgpu_buffer_rsrc_t rsG = make_rsrc_b((const char*)a.G + (size_t)a.g_off * ES,
                                                 ((long long)a.M * a.ldg - a.g_off) * ES);
  const unsigned acol = (unsigned)min(i0 + c4, a.I - 4) * (unsigned)ES;
  const unsigned gcol = (unsigned)min(j0 + min(c4, BJ - 4), a.J - 4) * (unsigned)ES;

  f32x16 acc[2][NTJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NTJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  unsigned tile_bits = 0;   // bit i*NTJ+j: 32x32 tile inside I x J (others issue no MFMA); scalar
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NTJ; ++j)
      if ((i0 + wm * 64 + i * 32 < a.I) && (j0 + (wn * NTJ + j) * 32 < a.J))
        tile_bits |= 1u << (i * NTJ + j);
  tile_bits = __builtin_amdgcn_readfirstlane(tile_bits);

  constexpr int LU = WBK / 8;      // loads per thread and operand per slab
  f32x4 ra[LU], rg[LU];
  // PLAIN: splits are whole slabs except the last one, whose rows >= M fall outside the
  // descriptors, so the offsets never change and the slab's first row is the scalar offset.
  unsigned aoffs[LU], goffs[LU];
#pragma unroll
  for (int u = 0; u < LU; ++u) {
    aoffs[u] = (unsigned)((kr + u * 8) * a.lda) * (unsigned)ES + acol;
    goffs[u] = (unsigned)((kr + u * 8) * a.ldg) * (unsigned)ES + gcol;
  }
#define K_WG_LOAD(MB)                                                                        \
  {                                                                                            \
    if (PLAIN) {                                                                               \
      const int sa = (MB) * a.lda * ES, sg = (MB) * a.ldg * ES;                                \
      _Pragma("unroll") for (int u = 0; u < LU; ++u) {                                          \
        ra[u] = buf_load_elems4<ES>(rsA, aoffs[u], sa);                                        \
        rg[u] = buf_load_elems4<ES>(rsG, goffs[u], sg);                                        \
      }                                                                                        \
    } else {                                                                                   \
      _Pragma("unroll") for (int u = 0; u < LU; ++u) {                                          \
        const int m = (MB) + kr + u * 8;                                                       \
        const RowPos p = decompose(m, mend, a.g);                                              \
        const int sr = src_row<0>(a.g, p, ky, kx);                                             \
        ra[u] = buf_load_elems4<ES>(                                                           \
            rsA, sr >= 0 ? (unsigned)(sr * a.lda) * (unsigned)ES + acol : OOB_OFFSET, 0);      \
        rg[u] = buf_load_elems4<ES>(                                                           \
            rsG, p.valid ? (unsigned)(m * a.ldg) * (unsigned)ES + gcol : OOB_OFFSET, 0);       \
      }                                                                                        \
    }                                                                                          \
  }
  K_WG_LOAD(mbeg);
  for (int mb = mbeg; mb < mend; mb += WBK) {
#pragma unroll
    for (int u = 0; u < LU; ++u) {
      *reinterpret_cast<f32x4*>(&As[(kr + u * 8) * WG_STRIDE + c4]) = ra[u];
      if (gload) *reinterpret_cast<f32x4*>(&Gs[(kr + u * 8) * WG_STRIDE + c4]) = rg[u];
    }
    __syncthreads();
    K_WG_LOAD(mb + WBK);   // next slab (rows past `mend` come back as zeros / are never used)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kc = 0; kc < WBK / 16; ++kc) {
      // all fragments of a 16-row chunk first, then ONE scalar branch per 32x32 tile around its
      // 8 chained MFMAs (per-MFMA conditions make hipcc shuffle whole accumulators through copies)
      float af[2][8], bf[NTJ][8];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int k = kc * 16 + lh * 8 + s;
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i][s] = As[k * WG_STRIDE + wm * 64 + i * 32 + li];
#pragma unroll
        for (int j = 0; j < NTJ; ++j) bf[j][s] = Gs[k * WG_STRIDE + (wn * NTJ + j) * 32 + li];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NTJ; ++j)
          if ((tile_bits >> (i * NTJ + j)) & 1u) {
#pragma unroll
            for (int s = 0; s < 8; ++s)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
          }
    }
    __syncthreads();
  }

#undef K_WG_LOAD
  // split-K result: atomics into dW, or this split's own slab (see wgrad_tn_bf16_kernel)
  float* dw = a.dW + (size_t)tap * a.I * a.J + (size_t)blk.z * a.part_stride;
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

template <int NTJ, bool PLAIN, int ES>
__global__ __launch_bounds__(256, 4) void wgrad_tn_kernel(WgradArgs a) {
  wgrad_tn_body<NTJ, PLAIN, ES>(a, wgrad_block(a));
}

// Filter gradients of SEVERAL 1x1 / stride-1 convolutions of one input in ONE launch (the entry
// convolutions of an Inception block, c2d_conv1x1_wgrad_multi): the row splits are shared, so the
// launch has as many splits as ALL its tiles leave room for — a third of the split-K atomics of
// the separate launches — and the x rows of a split are fetched into the XCD's L2 once for every
// output.  Block -> (split, problem, tile): all tiles of a split are consecutive on one XCD.
template <int NTJ, int ES>
__global__ __launch_bounds__(256, 4) void wgrad_tn_group_kernel(WgradGroupArgs g) {
  int p;
  const WgradBlock blk = wgrad_group_block(g, &p);
  wgrad_tn_body<NTJ, true, ES>(g.a[p], blk);
}

// ---------------------------------------------------------------------------------------------
// 3x3 / stride-1 filter gradient with ALL NINE TAPS per block.
//
// The per-tap kernel above re-reads the x and dC rows once per (tap, i-tile, j-tile): 36 passes
// over 175 MB for the 192->256 layer, which made it memory-bound (PMC: MFMA busy 59 %, a third
// of it on padding tiles).  Here a block owns 32 input channels x 128 output channels x a row
// range and keeps 9 accumulators (one per tap) per wave: the dC slab is staged once and used by
// all taps, and x is staged once with a (w+1)-row halo because tap (ky,kx) of output row m reads
// input row m + (ky-1)*w + (kx-1) of the same image.  SAME padding = a 9-bit validity mask per
// output row (LDS), applied to the A operand at fragment-read time.
// ---------------------------------------------------------------------------------------------
constexpr int W3_ASTR = 32 + 4;
constexpr int W3_GSTR = 128 + 4;

struct Wgrad3Args {
  const float* A; int lda; int a_off;
  const float* G; int ldg; int g_off;
  float* dW;
  int M, I, J;
  int rows_per_split, itiles, jtiles, tiles, splits;
  int h, w;
  long long part_stride;                // as WgradArgs::part_stride
};

// WC = compile-time (square) map width, IMGS = whole images per slab (even).  The x slab is
// staged in ZERO-PADDED image coordinates (every image gets a one-pixel border in LDS), so tap
// (ky,kx) of the output pixel at padded index P is simply LDS row P + (ky-1)*(WC+2) + (kx-1) and
// because a slab is a whole number of images every LDS row index is a compile-time constant.
// One MFMA k-step (k = 2) pairs the SAME pixel of two images (lower half-wave: image 2i, upper:
// image 2i+1), so whether tap (ky,kx) falls into the SAME padding is a compile-time property of
// the step and those MFMAs are simply not issued: 100 of 144 (pixel, tap) pairs remain on a 4x4
// map, 361 of 441 on 7x7 — no matrix-pipe time is spent multiplying padding zeros.
// Wave v owns output columns [j0+32v, j0+32v+32), all taps.
template <int WC, int IMGS, int ES>
__global__ __launch_bounds__(256, 2) void wgrad3x3_kernel(Wgrad3Args a) {
  constexpr bool PAIR = IMGS % 2 == 0;       // else: a k-step takes rows 2s, 2s+1 (no skipping)
  constexpr int PW = WC + 2;
  constexpr int HW = WC * WC;
  constexpr int PIMG = PW * PW;              // padded pixels per image
  constexpr int R = IMGS * HW;               // output rows per slab
  constexpr int STEPS = (R + 1) / 2;         // MFMA k-steps (PAIR: (image pair, pixel))
  constexpr int RP = STEPS * 2;              // staged dC rows (odd R: one zero row)
  constexpr int AROWS = IMGS * PIMG;
  constexpr int A_LD = (AROWS + 31) / 32;    // x loads per thread (rows ar0 + 32u)
  constexpr int G_LD = (RP + 7) / 8;         // dC loads per thread (rows gkr + 8u)
  __shared__ __attribute__((aligned(16))) float As[AROWS * W3_ASTR];
  __shared__ __attribute__((aligned(16))) float Gs[RP * W3_GSTR];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int logical = xcd_remap(blockIdx.x, a.tiles * a.splits);
  const int split = logical / a.tiles;
  const int t = logical - split * a.tiles;
  const int jt = t % a.jtiles, it_ = t / a.jtiles;
  const int i0 = it_ * 32, j0 = jt * 128;
  const int mbeg = split * a.rows_per_split;           // multiple of R
  const int mend = min(a.M, mbeg + a.rows_per_split);  // M is a multiple of HW
  const bool wave_on = j0 + wave * 32 < a.J;

  f32x16 acc[9];
#pragma unroll
  for (int q = 0; q < 9; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.0f;

  // loaders: everything about WHERE a staged row comes from is slab-invariant
  // Raw buffer loads (see igemm_body): per-lane byte offsets are slab-invariant, the slab's first
  // row goes into the scalar offset, rows of the zero border carry an out-of-range offset, and
  // rows past M (only the last slab can have them: splits are whole slabs) fall outside the
  // descriptors, which end at row M — the loader needs no vector ALU work per slab.
  const int gkr = tid >> 5, gc4 = (tid & 31) * 4;       // dC: rows gkr + 8u, float4 column gc4
  const __amdgpu_buffer_rsrc_t rsG = make_rsrc_b((const char*)a.G + (size_t)a.g_off * ES,
                                                 ((long long)a.M * a.ldg - a.g_off) * ES);
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc_b((const char*)a.A + (size_t)a.a_off * ES,
                                                 ((long long)a.M * a.lda - a.a_off) * ES);
  const int aq4 = (tid & 7) * 4, ar0 = tid >> 3;        // x: LDS rows ar0 + 32u, float4 col aq4
  // dC rows gkr + 8u: one per-lane offset, the 8-row steps ride in the scalar offset (a register
  // per load would cost the two-image 7x7 form eleven spills); only the last load can hold rows
  // >= R (they stay zero)
  unsigned aoffs[A_LD];
  const unsigned goff0 = (unsigned)(gkr * a.ldg + min(j0 + gc4, a.J - 4)) * (unsigned)ES;
  const unsigned goff_last = gkr + (G_LD - 1) * 8 < R ? goff0 : OOB_OFFSET;
  const int gstep = 8 * a.ldg * ES;
#pragma unroll
  for (int u = 0; u < A_LD; ++u) {
    const int r = ar0 + u * 32;
    const int im = r / PIMG, rr = r - im * PIMG;
    const int yp = rr / PW, xp = rr - yp * PW;
    const bool real = r < AROWS && yp >= 1 && yp <= WC && xp >= 1 && xp <= WC;
    const int apix = im * HW + (yp - 1) * WC + (xp - 1);   // pixel offset inside the slab
    aoffs[u] = real ? (unsigned)(apix * a.lda + i0 + aq4) * (unsigned)ES : OOB_OFFSET;
  }
  f32x4 rg[G_LD], ra[A_LD];

#define K_W3_LOAD(MB)                                                                        \
  {                                                                                            \
    const int sg = (MB) * a.ldg * ES, sa = (MB) * a.lda * ES;                                  \
    _Pragma("unroll") for (int u = 0; u < G_LD; ++u)                                           \
        rg[u] = buf_load_elems4<ES>(rsG, u == G_LD - 1 ? goff_last : goff0, sg + u * gstep);   \
    _Pragma("unroll") for (int u = 0; u < A_LD; ++u)                                           \
        ra[u] = buf_load_elems4<ES>(rsA, aoffs[u], sa);                                        \
  }

  // per-lane bases: the upper half-wave reads the odd image of each pair (PAIR) / the odd row
  const float* const apl = &As[(PAIR ? lh * PIMG : 0) * W3_ASTR + li];
  const float* const gpl = &Gs[(PAIR ? lh * HW : lh) * W3_GSTR + wave * 32 + li];

  K_W3_LOAD(mbeg);
  for (int mb = mbeg; mb < mend; mb += R) {
#pragma unroll
    for (int u = 0; u < G_LD; ++u)
      if (gkr + u * 8 < RP)
        *reinterpret_cast<f32x4*>(&Gs[(gkr + u * 8) * W3_GSTR + gc4]) = rg[u];
#pragma unroll
    for (int u = 0; u < A_LD; ++u)
      if (ar0 + u * 32 < AROWS)
        *reinterpret_cast<f32x4*>(&As[(ar0 + u * 32) * W3_ASTR + aq4]) = ra[u];
    __syncthreads();
    K_W3_LOAD(mb + R);
    __builtin_amdgcn_sched_barrier(0);
    if (wave_on) {
      if constexpr (PAIR) {
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
          const int ip = s / HW, p = s % HW;     // image pair, pixel (compile-time after unroll)
          const int y = p / WC, x = p % WC;
          const float* ap = apl + ((2 * ip) * PIMG + (y + 1) * PW + (x + 1)) * W3_ASTR;
          const float bv = gpl[((2 * ip) * HW + p) * W3_GSTR];
#pragma unroll
          for (int q = 0; q < 9; ++q) {
            const int dy = q / 3 - 1, dx = q % 3 - 1;
            if (y + dy < 0 || y + dy >= WC || x + dx < 0 || x + dx >= WC) continue;   // padding
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[(dy * PW + dx) * W3_ASTR], bv,
                                                          acc[q], 0, 0, 0);
          }
        }
      } else {
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
          // rows k = 2s (lower half-wave) and 2s+1 (upper); padded LDS row of pixel k
          const int k0 = 2 * s, k1 = (2 * s + 1 < R) ? 2 * s + 1 : 2 * s;
          const int p0 = (k0 / HW) * PIMG + ((k0 % HW) / WC + 1) * PW + (k0 % WC) + 1;
          const int p1 = (k1 / HW) * PIMG + ((k1 % HW) / WC + 1) * PW + (k1 % WC) + 1;
          const float* ap = apl + (lh ? p1 : p0) * W3_ASTR;
          const float bv = gpl[2 * s * W3_GSTR];
#pragma unroll
          for (int q = 0; q < 9; ++q) {
            const int off = ((q / 3) - 1) * PW + ((q % 3) - 1);
            acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[off * W3_ASTR], bv, acc[q], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();
  }
#undef K_W3_LOAD

  if (wave_on) {
    const int jj = j0 + wave * 32 + li;
    const bool part = a.part_stride > 0;     // (see wgrad_tn_bf16_kernel)
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      float* dw = a.dW + (size_t)q * a.I * a.J + (size_t)split * a.part_stride;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ii = i0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (part) dw[(size_t)ii * a.J + jj] = acc[q][r];
        else atomicAdd(dw + (size_t)ii * a.J + jj, acc[q][r]);
      }
    }
  }
}


// The same plan for the two STRIDE-2 3x3 convolutions of Mixed_5a (7x7 -> 4x4, TF SAME: one pad
// row / column on either side): x is staged per image as a zero-padded 9x9 map, dC as 16 rows; the
// k-step (s = output pixel (oy, ox) of an image pair) reads tap (ky, kx) at padded input
// (2 oy + ky, 2 ox + kx), a compile-time LDS row, and the taps that fall into the padding (oy or ox
// = 3 with ky / kx = 2) are not issued.  The per-tap kernel these layers ran on before re-read x
// and dC once per tap and tile and reached 86-97 TFLOP/s against 135-145 for the nine-tap form.
template <int ES>
__global__ __launch_bounds__(256, 2) void wgrad3x3_s2_kernel(Wgrad3Args a) {
  constexpr int WI = 7, WO = 4, IMGS = 2;
  constexpr int PW = WI + 2;
  constexpr int HWI = WI * WI, HWO = WO * WO;
  constexpr int PIMG = PW * PW;
  constexpr int R = IMGS * HWO;              // dC rows per slab (32)
  constexpr int STEPS = R / 2;               // (image pair, output pixel)
  constexpr int AROWS = IMGS * PIMG;         // staged (padded) x rows per slab (162)
  constexpr int A_LD = (AROWS + 31) / 32;
  constexpr int G_LD = (R + 7) / 8;
  __shared__ __attribute__((aligned(16))) float As[AROWS * W3_ASTR];
  __shared__ __attribute__((aligned(16))) float Gs[R * W3_GSTR];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int logical = xcd_remap(blockIdx.x, a.tiles * a.splits);
  const int split = logical / a.tiles;
  const int t = logical - split * a.tiles;
  const int jt = t % a.jtiles, it_ = t / a.jtiles;
  const int i0 = it_ * 32, j0 = jt * 128;
  const int mbeg = split * a.rows_per_split;           // OUTPUT rows, multiple of R
  const int mend = min(a.M, mbeg + a.rows_per_split);  // M = images * 16
  const bool wave_on = j0 + wave * 32 < a.J;

  f32x16 acc[9];
#pragma unroll
  for (int q = 0; q < 9; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.0f;

  const int gkr = tid >> 5, gc4 = (tid & 31) * 4;
  const long long in_rows = (long long)(a.M / HWO) * HWI;
  const __amdgpu_buffer_rsrc_t rsG = make_rsrc_b((const char*)a.G + (size_t)a.g_off * ES,
                                                 ((long long)a.M * a.ldg - a.g_off) * ES);
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc_b((const char*)a.A + (size_t)a.a_off * ES,
                                                 (in_rows * a.lda - a.a_off) * ES);
  const int aq4 = (tid & 7) * 4, ar0 = tid >> 3;
  unsigned goffs[G_LD], aoffs[A_LD];
#pragma unroll
  for (int u = 0; u < G_LD; ++u) {
    const int k = gkr + u * 8;
    goffs[u] = k < R ? (unsigned)(k * a.ldg + min(j0 + gc4, a.J - 4)) * (unsigned)ES : OOB_OFFSET;
  }
#pragma unroll
  for (int u = 0; u < A_LD; ++u) {
    const int r = ar0 + u * 32;
    const int im = r / PIMG, rr = r - im * PIMG;
    const int yp = rr / PW, xp = rr - yp * PW;
    const bool real = r < AROWS && yp >= 1 && yp <= WI && xp >= 1 && xp <= WI;
    const int apix = im * HWI + (yp - 1) * WI + (xp - 1);
    aoffs[u] = real ? (unsigned)(apix * a.lda + i0 + aq4) * (unsigned)ES : OOB_OFFSET;
  }
  f32x4 rg[G_LD], ra[A_LD];
#define K_W3S_LOAD(MB)                                                                       \
  {                                                                                            \
    const int sg = (MB) * a.ldg * ES, sa = ((MB) / HWO) * HWI * a.lda * ES;                    \
    _Pragma("unroll") for (int u = 0; u < G_LD; ++u)                                           \
        rg[u] = buf_load_elems4<ES>(rsG, goffs[u], sg);                                        \
    _Pragma("unroll") for (int u = 0; u < A_LD; ++u)                                           \
        ra[u] = buf_load_elems4<ES>(rsA, aoffs[u], sa);                                        \
  }
  const float* const apl = &As[(lh * PIMG) * W3_ASTR + li];           // upper half-wave: odd image
  const float* const gpl = &Gs[(lh * HWO) * W3_GSTR + wave * 32 + li];

  K_W3S_LOAD(mbeg);
  for (int mb = mbeg; mb < mend; mb += R) {
#pragma unroll
    for (int u = 0; u < G_LD; ++u)
      if (gkr + u * 8 < R)
        *reinterpret_cast<f32x4*>(&Gs[(gkr + u * 8) * W3_GSTR + gc4]) = rg[u];
#pragma unroll
    for (int u = 0; u < A_LD; ++u)
      if (ar0 + u * 32 < AROWS)
        *reinterpret_cast<f32x4*>(&As[(ar0 + u * 32) * W3_ASTR + aq4]) = ra[u];
    __syncthreads();
    K_W3S_LOAD(mb + R);
    __builtin_amdgcn_sched_barrier(0);
    if (wave_on) {
#pragma unroll
      for (int p = 0; p < HWO; ++p) {        // STEPS = HWO with IMGS = 2: one image pair per slab
        const int oy = p / WO, ox = p % WO;
        const float* ap = apl + ((2 * oy + 1) * PW + (2 * ox + 1)) * W3_ASTR;
        const float bv = gpl[p * W3_GSTR];
#pragma unroll
        for (int q = 0; q < 9; ++q) {
          const int dy = q / 3 - 1, dx = q % 3 - 1;
          if (2 * oy + dy < 0 || 2 * oy + dy >= WI || 2 * ox + dx < 0 || 2 * ox + dx >= WI) continue;
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[(dy * PW + dx) * W3_ASTR], bv, acc[q], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
#undef K_W3S_LOAD
  static_assert(STEPS == HWO, "one image pair per slab");

  if (wave_on) {
    const int jj = j0 + wave * 32 + li;
    const bool part = a.part_stride > 0;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      float* dw = a.dW + (size_t)q * a.I * a.J + (size_t)split * a.part_stride;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ii = i0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (part) dw[(size_t)ii * a.J + jj] = acc[q][r];
        else atomicAdd(dw + (size_t)ii * a.J + jj, acc[q][r]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// bf16 operands on v_mfma_f32_32x32x16_bf16 (fp32 accumulate), BASELINE configs[2] / [4].
//
// A filter gradient contracts over ROWS: both MFMA operands need 8 consecutive k (= rows) of
// ONE channel per lane, i.e. column reads of the row-major [row][channel] tiles.  gfx950's
// ds_read_b64_tr_b16 does that transpose in the LDS read path (cdna_hip_programming.md T10): per
// 16-lane group, lane 4q+p supplies the address of row q / columns 4p..4p+3 of a 4x16 block and
// lane i receives column i of the four rows.  Group g = lane>>4 reads columns 16*(g&1).. of rows
// 8*(g>>1) + 4t.. (t = 0,1: two reads = the 8 k of the lane's half), so lane l ends up with
// channel l&31 and k = 8*(l>>5) + e in element e: the 32x32x16 A/B operand map.  Both operands
// go through the same read, hence share the k order.  The tiles are staged as they lie in HBM
// (16-B chunks of 8 channels): no register transposes, no per-element LDS writes.
// A 32-lane half reads 4 rows x 64 B: conflict-free when the row stride is 64 B mod 256 B.
// ---------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// the two transposed reads of one operand fragment: rows +0..3 at p, rows +4..7 at p + hi_off
__device__ __forceinline__ bf16x8 tr_frag(const char* p, int hi_off) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)p);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + hi_off));
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

constexpr int WB_KB = 32;          // rows of M per slab: two 16-row MFMA k-steps
constexpr int WB_RS = 256 + 64;    // bytes per staged row (128 bf16 + pad: stride = 64 mod 256)

// Per-tap filter gradient, bf16 operands: same tiling / grid as wgrad_tn_kernel.
// KG = K-groups per block: KG x 4 waves work on the SAME 128 x BJ output tile, each group on its
// own rows of the block's split with its own staging buffers; the groups' accumulators are summed
// through LDS before ONE group adds the tile to dW.  The split-K atomics (1.3 TB/s chip-wide,
// a third of these launches at bf16 MFMA rates) scale with the number of BLOCKS, the latency
// hiding with the number of WAVES: KG decouples the two.
template <int NTJ, bool PLAIN, int KG>
__global__ __launch_bounds__(256 * KG, KG == 1 ? 4 : KG == 2 ? 2 : 1)
void wgrad_tn_bf16_kernel(WgradArgs a) {
  constexpr int BJ = 2 * NTJ * 32;
  constexpr int TILE_BYTES = WB_KB * WB_RS;
  __shared__ __attribute__((aligned(16))) char smem[KG * 2 * TILE_BYTES];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = wave >> 2, w4 = wave & 3;
  const int wm = w4 >> 1, wn = w4 & 1;
  const int li = lane & 31, lh = lane >> 5;
  char* const As = smem + kg * 2 * TILE_BYTES;
  char* const Gs = As + TILE_BYTES;
  const int itiles = (a.I + 127) / 128;
  const WgradBlock blk = wgrad_block(a);
  const int tap = blk.x / itiles;
  const int i0 = (blk.x - tap * itiles) * 128;
  const int j0 = blk.y * BJ;
  const int ky = tap / a.g.kw, kx = tap - ky * a.g.kw;
  // rows of this block's split, dealt to the K-groups in equal shares (host: rows_per_split is a
  // multiple of KG * WB_KB); every group runs the same number of slabs (rows past M are zeros)
  const int bbeg = blk.z * a.rows_per_split;
  const int rpg = a.rows_per_split / KG;
  const int mbeg = bbeg + kg * rpg;
  const int mend = min(a.M, mbeg + rpg);
  const int nslabs = (min(rpg, a.M - bbeg) + WB_KB - 1) / WB_KB;

  // loader: thread -> rows kr, kr+16 of the slab, 16-B chunk (8 channels) c8; columns beyond
  // I / J are clamped (their products land in dW rows / columns that are never stored)
  const int tg_ = tid & 255;
  const int kr = tg_ >> 4;
  const int c8 = (tg_ & 15) * 8;
  const bool gload = c8 < BJ;
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc_b((const char*)a.A + (size_t)a.a_off * 2,
                                                 (a.a_rows * a.lda - a.a_off) * 2);
  const __amdgpu_buffer_rsrc_t rsG = make_rsrc_b((const char*)a.G + (size_t)a.g_off * 2,
                                                 ((long long)a.M * a.ldg - a.g_off) * 2);
  const unsigned acol = (unsigned)min(i0 + c8, a.I - 8) * 2u;
  const unsigned gcol = (unsigned)min(j0 + min(c8, BJ - 8), a.J - 8) * 2u;

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

  f32x4 ra[2], rg[2];
  unsigned aoffs[2], goffs[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    aoffs[u] = (unsigned)((kr + u * 16) * a.lda) * 2u + acol;
    goffs[u] = (unsigned)((kr + u * 16) * a.ldg) * 2u + gcol;
  }
  // (PLAIN: rows >= M lie outside the descriptors and come back as zeros; a group's rows end
  // where the next group's begin, so no row is counted twice)
#define K_WB_LOAD(MB)                                                                        \
  {                                                                                            \
    if (PLAIN) {                                                                               \
      const int sa = (MB) * a.lda * 2, sg = (MB) * a.ldg * 2;                                  \
      _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                          \
        ra[u] = buf_load4(rsA, aoffs[u], sa);                                                  \
        rg[u] = buf_load4(rsG, goffs[u], sg);                                                  \
      }                                                                                        \
    } else {                                                                                   \
      _Pragma("unroll") for (int u = 0; u < 2; ++u) {                                          \
        const int m = (MB) + kr + u * 16;                                                      \
        const RowPos p = decompose(m, mend, a.g);                                              \
        const int sr = src_row<0>(a.g, p, ky, kx);                                             \
        ra[u] = buf_load4(rsA, sr >= 0 ? (unsigned)(sr * a.lda) * 2u + acol : OOB_OFFSET, 0);  \
        rg[u] = buf_load4(rsG, p.valid ? (unsigned)(m * a.ldg) * 2u + gcol : OOB_OFFSET, 0);   \
      }                                                                                        \
    }                                                                                          \
  }
  // transposed-read bases: row 8*lh + q, 16-column half (lane>>4)&1, 4-column slot p
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
  const char* const apl = As + (8 * lh + tq) * WB_RS + (wm * 64 + 16 * tg + 4 * tp) * 2;
  const char* const gpl = Gs + (8 * lh + tq) * WB_RS + (wn * NTJ * 32 + 16 * tg + 4 * tp) * 2;

  K_WB_LOAD(mbeg);
  for (int sl = 0, mb = mbeg; sl < nslabs; ++sl, mb += WB_KB) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      *reinterpret_cast<f32x4*>(As + (kr + u * 16) * WB_RS + c8 * 2) = ra[u];
      if (gload) *reinterpret_cast<f32x4*>(Gs + (kr + u * 16) * WB_RS + c8 * 2) = rg[u];
    }
    __syncthreads();
    K_WB_LOAD(mb + WB_KB);
    __builtin_amdgcn_sched_barrier(0);
    {
      bf16x8 af[2][2], bf[NTJ][2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i][s] = tr_frag(apl + s * 16 * WB_RS + i * 64, 4 * WB_RS);
#pragma unroll
        for (int j = 0; j < NTJ; ++j) bf[j][s] = tr_frag(gpl + s * 16 * WB_RS + j * 64, 4 * WB_RS);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NTJ; ++j)
          if ((tile_bits >> (i * NTJ + j)) & 1u) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
          }
    }
    __syncthreads();
  }
#undef K_WB_LOAD

  // K-groups 1 .. KG-1 hand their accumulators to group 0 through LDS, one 32x32 tile per round
  // ((KG - 1) x 16 KiB of the staging buffers); group 0 sums in group order
  if constexpr (KG > 1) {
    float* const red = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NTJ; ++j) {
        if (kg > 0) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            red[((kg - 1) * 4 + w4) * 1024 + r * 64 + lane] = acc[i][j][r];
        }
        __syncthreads();
        if (kg == 0) {
#pragma unroll
          for (int g = 1; g < KG; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] += red[((g - 1) * 4 + w4) * 1024 + r * 64 + lane];
        }
        __syncthreads();
      }
    if (kg > 0) return;
  }

  // Split-K result: fp32 atomics into dW, or (part_stride > 0) plain stores into this split's own
  // slab — global float atomics run at 1.3 TB/s chip-wide against 6 TB/s for stores of the same
  // shape (MI355X_MICROARCH.md); the slabs are summed in split order by wgrad_reduce_kernel.
  float* dw = a.dW + (size_t)tap * a.I * a.J + (size_t)blk.z * a.part_stride;
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

// 3x3 / stride-1 filter gradient, all nine taps per block, bf16 operands.  Same plan as
// wgrad3x3_kernel (x staged once in zero-padded image coordinates so that tap (ky,kx) of the
// output pixel at padded index P is LDS row P + (ky-1)*(WC+2) + (kx-1); dC staged once for all
// taps), but a k-step is 16 CONSECUTIVE output rows (a whole 4x4 map) and the padding taps ride
// along as zeros: skipping them would need 16 images per pixel in a step (a 16-image dC slab does
// not fit beside the accumulators), and at bf16 rates the kernel is bound by staging, not MFMA.
// WI = 32-channel i-groups per block (waves = 4*WI: wave -> (i-group, 32-column j-group)).
constexpr int W3B_RSA = 64;         // 32 bf16 per x row: consecutive rows = 64 B apart
constexpr int W3B_RSG = 256 + 64;   // 128 bf16 per dC row + pad

// KG = K-groups (see wgrad_tn_bf16_kernel): KG x (4 * WI) waves on the same output tile, each
// group on its own images with its own staging buffers, summed through LDS before the atomics.
template <int WC, int IMGS, int WI, int KG>
__device__ __forceinline__ void wgrad3x3_bf16_body(const Wgrad3Args& a, const int split, const int t) {
  constexpr int NT = 256 * WI;                // threads of one K-group
  constexpr int PW = WC + 2;
  constexpr int HW = WC * WC;
  constexpr int PIMG = PW * PW;
  constexpr int R = IMGS * HW;                // output rows per slab
  constexpr int STEPS = (R + 15) / 16;        // MFMA k-steps
  constexpr int RP = STEPS * 16;              // staged dC rows (rows >= R stay zero)
  constexpr int AROWS = IMGS * PIMG;
  constexpr int A_LD = (R * 4 + 255) / 256;   // 16-B x chunks per thread (per i-group: 256 thr)
  constexpr int G_LD = (R * 16 + NT - 1) / NT;
  constexpr int AS_BYTES = WI * AROWS * W3B_RSA, GS_BYTES = RP * W3B_RSG;
  __shared__ __attribute__((aligned(16))) char smem[KG * (AS_BYTES + GS_BYTES)];
  const int kg = __builtin_amdgcn_readfirstlane((int)threadIdx.x / NT);
  const int tid = (int)threadIdx.x - kg * NT;    // thread inside the K-group
  char* const As = smem + kg * (AS_BYTES + GS_BYTES);
  char* const Gs = As + AS_BYTES;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave >> 2, wj = wave & 3;
  const int li = lane & 31, lh = lane >> 5;
  const int jt = t % a.jtiles, it_ = t / a.jtiles;
  const int i0 = it_ * 32 * WI, j0 = jt * 128;
  // the split's rows in equal shares per K-group (host: rows_per_split is a multiple of KG * R);
  // every group runs the same number of slabs (rows past M are zeros)
  const int bbeg = split * a.rows_per_split;
  const int rpg = a.rows_per_split / KG;
  const int mbeg = bbeg + kg * rpg;
  const int nslabs = (min(rpg, a.M - bbeg) + R - 1) / R;
  const bool wave_on = (j0 + wj * 32 < a.J) && (i0 + wi * 32 < a.I);

  f32x16 acc[9];
#pragma unroll
  for (int q = 0; q < 9; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.0f;

  // zero the border rows of x and the tail rows of dC once: the loaders only rewrite real rows
  for (int e = tid; e < AS_BYTES / 16; e += NT)
    reinterpret_cast<f32x4*>(As)[e] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int e = tid; e < GS_BYTES / 16; e += NT)
    reinterpret_cast<f32x4*>(Gs)[e] = f32x4{0.f, 0.f, 0.f, 0.f};

  const __amdgpu_buffer_rsrc_t rsG = make_rsrc_b((const char*)a.G + (size_t)a.g_off * 2,
                                                 ((long long)a.M * a.ldg - a.g_off) * 2);
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc_b((const char*)a.A + (size_t)a.a_off * 2,
                                                 ((long long)a.M * a.lda - a.a_off) * 2);
  // x: i-group ag = tid / 256 stages its own [AROWS][32] image; chunk e -> (row e>>2, 8 channels)
  const int ag = tid >> 8, at = tid & 255;
  unsigned aoffs[A_LD]; int adst[A_LD];
#pragma unroll
  for (int u = 0; u < A_LD; ++u) {
    const int e = at + u * 256, k = e >> 2, c = e & 3;
    const int im = k / HW, pix = k - im * HW, y = pix / WC, x = pix - y * WC;
    const bool on = k < R && i0 + ag * 32 < a.I;
    aoffs[u] = on ? (unsigned)(k * a.lda + i0 + ag * 32 + c * 8) * 2u : OOB_OFFSET;
    adst[u] = on ? (ag * AROWS + im * PIMG + (y + 1) * PW + x + 1) * W3B_RSA + c * 16 : -1;
  }
  unsigned goffs[G_LD]; int gdst[G_LD];
#pragma unroll
  for (int u = 0; u < G_LD; ++u) {
    const int e = tid + u * NT, k = e >> 4, c = e & 15;
    const bool on = k < R;
    goffs[u] = on ? (unsigned)(k * a.ldg + min(j0 + c * 8, a.J - 8)) * 2u : OOB_OFFSET;
    gdst[u] = on ? k * W3B_RSG + c * 16 : -1;
  }
  f32x4 rg[G_LD], ra[A_LD];
#define K_W3B_LOAD(MB)                                                                       \
  {                                                                                            \
    const int sg = (MB) * a.ldg * 2, sa = (MB) * a.lda * 2;                                    \
    _Pragma("unroll") for (int u = 0; u < G_LD; ++u) rg[u] = buf_load4(rsG, goffs[u], sg);     \
    _Pragma("unroll") for (int u = 0; u < A_LD; ++u) ra[u] = buf_load4(rsA, aoffs[u], sa);     \
  }

  // transposed-read addresses: k-step s, read t -> slab row k = 16s + 8*lh + 4t + q
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
  const char* const abase = As + wi * AROWS * W3B_RSA + (16 * tg + 4 * tp) * 2;
  const char* const gbase = Gs + (8 * lh + tq) * W3B_RSG + (wj * 32 + 16 * tg + 4 * tp) * 2;
  int arow[STEPS][2];      // padded x row of the lane's k (bytes); compile-time stride when HW == 16
#pragma unroll
  for (int s = 0; s < STEPS; ++s)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      int k = 16 * s + 8 * lh + 4 * u + tq;
      if (k >= R) k = 0;      // dC rows >= R are zero: any staged x row will do
      const int im = k / HW, pix = k - im * HW, y = pix / WC, x = pix - y * WC;
      arow[s][u] = (im * PIMG + (y + 1) * PW + x + 1) * W3B_RSA;
    }

  K_W3B_LOAD(mbeg);
  __syncthreads();
  for (int sl = 0, mb = mbeg; sl < nslabs; ++sl, mb += R) {
#pragma unroll
    for (int u = 0; u < G_LD; ++u)
      if (gdst[u] >= 0) *reinterpret_cast<f32x4*>(Gs + gdst[u]) = rg[u];
#pragma unroll
    for (int u = 0; u < A_LD; ++u)
      if (adst[u] >= 0) *reinterpret_cast<f32x4*>(As + adst[u]) = ra[u];
    __syncthreads();
    K_W3B_LOAD(mb + R);
    __builtin_amdgcn_sched_barrier(0);
    if (wave_on) {
#pragma unroll
      for (int s = 0; s < STEPS; ++s) {
        const bf16x8 bv = tr_frag(gbase + s * 16 * W3B_RSG, 4 * W3B_RSG);
        const char* const a0 = abase + (HW == 16 ? arow[0][0] + s * PIMG * W3B_RSA : arow[s][0]);
        const char* const a1 = abase + (HW == 16 ? arow[0][1] + s * PIMG * W3B_RSA : arow[s][1]);
#pragma unroll
        for (int q = 0; q < 9; ++q) {
          const int off = ((q / 3 - 1) * PW + (q % 3 - 1)) * W3B_RSA;
          const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + off));
          const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a1 + off));
          const bf16x8 av = __builtin_bit_cast(
              bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[q], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
#undef K_W3B_LOAD

  // K-groups 1 .. KG-1 hand their nine accumulators to group 0 through LDS, three taps per round
  if constexpr (KG > 1) {
    float* const red = reinterpret_cast<float*>(smem);
    static_assert((KG - 1) * 4 * WI * 48 * 64 * 4 <= KG * (AS_BYTES + GS_BYTES), "reduction staging");
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      __syncthreads();
      if (kg > 0) {
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            red[(((kg - 1) * 4 * WI + wave) * 48 + q * 16 + r) * 64 + lane] = acc[3 * c + q][r];
      }
      __syncthreads();
      if (kg == 0) {
#pragma unroll
        for (int g = 1; g < KG; ++g)
#pragma unroll
          for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              acc[3 * c + q][r] += red[(((g - 1) * 4 * WI + wave) * 48 + q * 16 + r) * 64 + lane];
      }
    }
    if (kg > 0) return;
  }
  if (wave_on) {
    const int jj = j0 + wj * 32 + li;
    const bool part = a.part_stride > 0;     // (see wgrad_tn_bf16_kernel)
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      float* dw = a.dW + (size_t)q * a.I * a.J + (size_t)split * a.part_stride;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ii = i0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (part) dw[(size_t)ii * a.J + jj] = acc[q][r];
        else atomicAdd(dw + (size_t)ii * a.J + jj, acc[q][r]);
      }
    }
  }
}

template <int WC, int IMGS, int WI, int KG>
__global__ __launch_bounds__(256 * WI * KG, WI * KG == 1 ? 2 : 1) void wgrad3x3_bf16_kernel(Wgrad3Args a) {
  const int logical = xcd_remap(blockIdx.x, a.tiles * a.splits);
  const int split = logical / a.tiles;
  wgrad3x3_bf16_body<WC, IMGS, WI, KG>(a, split, logical - split * a.tiles);
}

// The nine-tap filter gradients of SEVERAL 3x3 convolutions over the same per-ROI maps — the 3x3
// layers of one Inception block — in ONE launch (round 5).  A nine-tap launch of 4x4 maps is
// dominated by what does not scale with its rows: 252 workgroups x 147 KB of fp32 atomics (37 MB at
// the chip's 1.3 TB/s) and the ramps, 45 of its 55 us (time = 45 us + 0.8 us per 128-row slab,
// profiles/r05_experiments/README.md).  The problems of a group share the row splits, so the launch
// has as many splits as ALL its tiles leave room for in one round of workgroups: a third of the
// atomics and of the ramps of three launches.  Block -> (split, problem, tile); all tiles of a
// split consecutive on one XCD.
constexpr int WGRAD3_GROUP_MAX = 3;
struct Wgrad3GroupArgs {
  Wgrad3Args a[WGRAD3_GROUP_MAX];
  int first[WGRAD3_GROUP_MAX + 1];   // first tile of problem p inside a split; [num] = tiles per split
  int num, splits;
};
template <int WC, int IMGS>
__global__ __launch_bounds__(256, 2) void wgrad3x3_bf16_group_kernel(Wgrad3GroupArgs g) {
  const int total = g.first[g.num];
  const int logical = xcd_remap(blockIdx.x, total * g.splits);
  const int split = logical / total;
  int t = logical - split * total;
  int p = 0;
  for (int i = 1; i < g.num; ++i)
    if (t >= g.first[i]) p = i;
  wgrad3x3_bf16_body<WC, IMGS, 1, 1>(g.a[p], split, t - g.first[p]);
}

// dW[i] += sum over splits (in split order: reproducible) of the slabs written by the bf16
// filter-gradient kernels in their partial mode; one launch for all layers of a backward pass.
struct WgradReduceDesc {
  long long ws_off;      // first float of split 0's slab in the workspace
  long long dw_off;      // first float of the filter gradient in the flat gradient buffer
  int numel;             // taps * cin * cout
  int splits;
  int begin;             // first 1024-element chunk of this layer in the launch
  int pad;
};
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const WgradReduceDesc* __restrict__ desc,
                                                           int num, const float* __restrict__ ws,
                                                           float* __restrict__ grads) {
  int d = 0;
  for (int i = 1; i < num; ++i)
    if ((int)blockIdx.x >= desc[i].begin) d = i;
  const WgradReduceDesc L = desc[d];
  const int e = ((int)blockIdx.x - L.begin) * 1024 + threadIdx.x * 4;
  if (e >= L.numel) return;              // numel is a multiple of 4 (cin, cout multiples of 8)
  const float* src = ws + L.ws_off + e;
  f32x4 sum = *reinterpret_cast<const f32x4*>(src);
  for (int z = 1; z < L.splits; ++z) sum += *reinterpret_cast<const f32x4*>(src + (size_t)z * L.numel);
  f32x4* dst = reinterpret_cast<f32x4*>(grads + L.dw_off + e);
  *dst = *dst + sum;
}

void set_magic(ConvGeom* g);

int fill_geom(ConvGeom* g, int ih, int iw, int kh, int kw, int stride, int mode) {
  if (ih <= 0 || iw <= 0 || kh <= 0 || kw <= 0 || stride <= 0 || stride > 2) return C2D_ERR_INVALID_ARG;
  g->ih = ih; g->iw = iw; g->kh = kh; g->kw = kw; g->stride = stride; g->mode = mode;
  g->oh = (ih + stride - 1) / stride;
  g->ow = (iw + stride - 1) / stride;
  // TF 'SAME': pad_total = max((out-1)*stride + k - in, 0), the extra pixel goes bottom/right.
  const int pth = (g->oh - 1) * stride + kh - ih, ptw = (g->ow - 1) * stride + kw - iw;
  g->pad_t = (pth > 0 ? pth : 0) / 2;
  g->pad_l = (ptw > 0 ? ptw : 0) / 2;
  g->rh = mode == 0 ? g->oh : ih;
  g->rw = mode == 0 ? g->ow : iw;
  g->sub = 1; g->y0 = 0; g->x0 = 0;
  g->ky0 = 0; g->kx0 = 0; g->kstep = 1; g->nky = kh; g->nkx = kw;
  g->nimg = 0; g->pm = 0; g->lpt_ngx = 0;
  set_magic(g);
  return C2D_OK;
}

void set_magic(ConvGeom* g) {
  const unsigned long long one = 1ull << 40;
  g->magic_hw = (one + (unsigned long long)(g->rh * g->rw) - 1) / (unsigned long long)(g->rh * g->rw);
  g->magic_w = (one + (unsigned long long)g->rw - 1) / (unsigned long long)g->rw;
}

struct IgemmWs {
  void* ptr;
  long long bytes;
};

int gcd_int(int a, int b) { return b == 0 ? a : gcd_int(b, a % b); }

// Fills the stream-K plan for tile BM x BN; returns false when the launch should use the
// one-tile-per-block form instead (no workspace, degenerate costs, ...).
template <int MODE, int BM, int BN, int BKT, bool PM>
bool make_sk_plan(const IgemmArgs& a, int slots, const IgemmWs& ws, SkPlan* sk, int* err) {
  *err = C2D_OK;
  if (!ws.ptr || slots <= 0) return false;
  const int kslabs = c2d_ceil_div(a.K, BKT);
  const int ntaps = a.g.nky * a.g.nkx;
  sk->enabled = 1;
  if (PM) {
    const int hw = a.g.rh * a.g.rw, tb = BM / 32;
    sk->period = hw / gcd_int(hw, tb);
    if (sk->period > SK_MAX_PERIOD) return false;
    for (int r = 0; r < sk->period; ++r) {
      unsigned long long mask = 0;
      for (int t = 0; t < tb; ++t) {
        const int px = (r * tb + t) % hw, y = px / a.g.rw, x = px % a.g.rw;
        for (int tp = 0; tp < ntaps; ++tp) {
          const int ty = tp / a.g.nkx;
          if (tap_ok<MODE>(a.g, y, x, a.g.ky0 + a.g.kstep * ty,
                           a.g.kx0 + a.g.kstep * (tp - ty * a.g.nkx)))
            mask |= 1ull << tp;
        }
      }
      sk->cost[r] = __builtin_popcountll(mask) * kslabs;
    }
  } else {
    sk->period = 1;
    sk->cost[0] = a.nseg > 1 ? a.total_slabs : ntaps * kslabs;
  }
  sk->prefix[0] = 0;
  for (int r = 0; r < sk->period; ++r) {
    if (sk->cost[r] <= 0) return false;
    sk->prefix[r + 1] = sk->prefix[r] + sk->cost[r];
  }
  const long long rows = (long long)(a.m_tiles / sk->period) * sk->prefix[sk->period] +
                         sk->prefix[a.m_tiles % sk->period];
  const long long total = rows * a.n_tiles;
  if (total <= 0 || total >= (1ll << 30)) return false;
  sk->total = (int)total;
  int grid = slots < sk->total ? slots : sk->total;
  sk->share = (sk->total + grid - 1) / grid;
  const long long tiles = (long long)a.m_tiles * a.n_tiles;
  const long long part_bytes = (long long)grid * 2 * BM * BN * 4;
  if (part_bytes + tiles * 4 > ws.bytes) { *err = C2D_ERR_WORKSPACE; return false; }
  sk->partials = (float*)ws.ptr;
  sk->counters = (int*)((char*)ws.ptr + part_bytes);
  return true;
}

bool sk_disabled_by_env() {
  static const bool tune = c2d_tune_on();
  if (!tune) return false;
  const char* e = c2d_tune_get("igemm_sk");
  return e && e[0] == '0';
}

// Row blocks (m-tiles) of the launch run_igemm chose: recorded for the fused BN/ReLU backward
// (the caller sizes / offsets the partial-sum rows with it); g_tile_query: record only, no launch.
static thread_local int g_last_m_tiles = 0;
static thread_local bool g_tile_query = false;

template <int MODE, int WM, int WN, int MT, int NT, int BKT, bool PM, int ES>
int launch_igemm_mode(IgemmArgs a, hipStream_t s, const IgemmWs& ws) {
  constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
  a.m_tiles = c2d_ceil_div(a.M, BM);
  a.n_tiles = c2d_ceil_div(a.N, BN);
  g_last_m_tiles = a.m_tiles;
  if (ES == 4 && !ws.ptr) {
    // the DMA-ring kernel of igemm_bf16.hip on fp32 operands, where enabled (round 3)
    IgemmArgs b = a;
    b.g.mode = MODE;
    int mtiles = 0;
    const int rc = launch_igemm_f32_ring(b, WM, WN, MT, NT, PM, s, &mtiles, g_tile_query);
    if (rc != C2D_ERR_UNSUPPORTED) return rc;
  }
  if (g_tile_query) return C2D_OK;
  if (a.nseg > 1) {
    a.total_slabs = 0;
    for (int i = 0; i < a.nseg; ++i) a.total_slabs += c2d_ceil_div(a.segK[i], BKT);
  }
#ifdef C2D_TRACE
  a.trace = g_trace;
#endif
  const dim3 block(WM * WN * 64);
  if (ws.ptr && !sk_disabled_by_env()) {
    // resident workgroups of the persistent kernel = its occupancy x the CU count (queried once)
    static int slots = -1;
    if (slots < 0) {
      int occ = 0, dev = 0;
      hipDeviceProp_t prop;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(
              &occ, igemm_sk_kernel<MODE, WM, WN, MT, NT, BKT, PM, ES>, WM * WN * 64, 0) == hipSuccess &&
          hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
        slots = occ * prop.multiProcessorCount;
      else
        slots = 0;
    }
    IgemmSkArgs p;
    p.a = a;
    int err = C2D_OK;
    if (make_sk_plan<MODE, BM, BN, BKT, PM>(a, slots, ws, &p.sk, &err)) {
      const int grid = c2d_ceil_div(p.sk.total, p.sk.share);
      dispatch_note(PM ? "igemm_sk_kernel<%d, %d, %d, %d, %d, %d, true, %d>"
                       : "igemm_sk_kernel<%d, %d, %d, %d, %d, %d, false, %d>", MODE, WM, WN, MT, NT, BKT, ES);
      hipLaunchKernelGGL((igemm_sk_kernel<MODE, WM, WN, MT, NT, BKT, PM, ES>), dim3(grid), block, 0, s, p);
      return c2d_launch_status();
    }
    if (err) return err;
  }
  const dim3 grid(a.m_tiles * a.n_tiles);
  dispatch_note(PM ? "igemm_nt_kernel<%d, %d, %d, %d, %d, %d, true, %d>"
                   : "igemm_nt_kernel<%d, %d, %d, %d, %d, %d, false, %d>", MODE, WM, WN, MT, NT, BKT, ES);
  hipLaunchKernelGGL((igemm_nt_kernel<MODE, WM, WN, MT, NT, BKT, PM, ES>), grid, block, 0, s, a);
  return c2d_launch_status();
}

template <int MODE, int WM, int WN, int MT, int NT, bool PM>
int launch_igemm_bf16(IgemmArgs a, hipStream_t s) {
  constexpr int BM = WM * MT * 32;
  g_last_m_tiles = c2d_ceil_div(a.M, BM);
  a.g.mode = MODE;
  int mtiles = 0;
  return launch_igemm_bf16_ring(a, WM, WN, MT, NT, PM, s, &mtiles, g_tile_query);
}

// Full-width tiles: 128 rows x every output column (N <= 384, rounded up to 64) in ONE block of
// 8 waves (4 x 2, each 32 rows x N/2 columns), one block per CU: the activations stream from HBM
// exactly once and a launch of 2000 4x4 maps is 250 blocks, one round of the 256 CUs.
template <bool PM>
int launch_igemm_bf16_wide(const IgemmArgs& a, hipStream_t s) {
  const int nt = c2d_ceil_div(a.N, 64);
#define K_WIDE(NT_)                                                                 \
  case NT_:                                                                           \
    if (a.g.mode == 0) return launch_igemm_bf16<0, 4, 2, 1, NT_, PM>(a, s);           \
    return launch_igemm_bf16<1, 4, 2, 1, NT_, PM>(a, s);
  switch (nt) {
    K_WIDE(2) K_WIDE(3) K_WIDE(4) K_WIDE(5) K_WIDE(6)
    default: break;
  }
#undef K_WIDE
  if (a.g.mode == 0) return launch_igemm_bf16<0, 2, 4, 2, 2, PM>(a, s);
  return launch_igemm_bf16<1, 2, 4, 2, 2, PM>(a, s);
}

template <int WM, int WN, int MT, int NT, int BKT, bool PM = false>
int launch_igemm(const IgemmArgs& a, hipStream_t s, const IgemmWs& ws) {
  if (a.es == 2) {
    if (a.g.mode == 0) return launch_igemm_bf16<0, WM, WN, MT, NT, PM>(a, s);
    return launch_igemm_bf16<1, WM, WN, MT, NT, PM>(a, s);
  }
  if (a.g.mode == 0) return launch_igemm_mode<0, WM, WN, MT, NT, BKT, PM, 4>(a, s, ws);
  return launch_igemm_mode<1, WM, WN, MT, NT, BKT, PM, 4>(a, s, ws);
}

// Assembly of a grouped small-problem launch: while g_collect is set (host, one thread: the
// C-ABI is not re-entrant across threads for grouped calls, see the header), run_igemm records
// small problems instead of launching them and refuses everything else.
struct SmallCollect {
  IgemmGroupArgs args;
  int num;
  int mode;
  int es;
};
static thread_local SmallCollect* g_collect = nullptr;

// f32x9 (igemm_x9.hip): does every weight operand of `a` lie in an arena bound with c2d_f32x9_bind?
// Fills the plane pointers when so.
bool x9_resolve(IgemmArgs* a) {
  if (a->es != 4 || a->N % 4 != 0 || a->K % 16 != 0) return false;
  long long stride = 0, st = 0;
  if (a->nseg > 1) {
    for (int i = 0; i < a->nseg; ++i) {
      if (a->segK[i] % 16 != 0) return false;
      a->segBp[i] = x9_planes_of(a->segB[i], (long long)a->N * a->segK[i] * 4, &st);
      if (!a->segBp[i] || (i && st != stride)) return false;
      stride = st;
    }
    a->Bp = a->segBp[0];
  } else {
    const long long bytes = a->mo_n ? a->mo_bbytes : (long long)a->g.kh * a->g.kw * a->N * a->K * 4;
    a->Bp = x9_planes_of(a->Bt, bytes, &stride);
    if (!a->Bp) return false;
  }
  a->bp_stride = stride;
  return true;
}

// f32x9 block tile: 128 rows (four waves of 32 rows: no two waves split the same activation rows) x
// 32 NT columns.  Pixel-major launches (one pixel of 128 images per block, 4 - 9 taps of work):
// 128-column tiles (tools/sweep_x9.sh: 64-column tiles re-stream the activations once more for
// nothing, wider ones leave too few blocks for the heavy-first order to even the CUs out);
// row-major launches: the NT in 2..5 that pads the output width least, the widest on a tie.
int x9_launch(IgemmArgs a, bool pm, hipStream_t s) {
  static const bool tune = c2d_tune_on();
  static const int force_nt = (tune && c2d_tune_get("x9_nt")) ? atoi(c2d_tune_get("x9_nt")) : 0;
  static const int force_nt_pm = (tune && c2d_tune_get("x9_nt_pm")) ? atoi(c2d_tune_get("x9_nt_pm")) : 0;
  // Tile width = NT x 32 columns: the fewest padded columns, the wider tile on a tie.  Pixel-major
  // launches (3x3 over 4x4 / 7x7 maps) stop at 128 columns: measured per call of the step
  // (bench.py --per-call, C2D_TUNE=x9_nt_pm=2..6, N = 2000): 192 columns 3 x 64 or 2 x 96 beat
  // 2 x 128 by 12 - 17 % (input gradient 256 -> 192 on 7x7: 606 -> 509 us), 160 columns 2 x 96
  // beats 2 x 128 and 1 x 160 (165 / 171 -> 138 us), 320 columns 5 x 64 beats 3 x 128 and 2 x 160
  // (206 / 191 -> 182 us), 224 and 256 columns stay on 128-wide tiles.
  int best_nt = 2, best_cols = 1 << 30;
  for (int nt = pm ? 4 : 5; nt >= 2; --nt) {
    const int cols = c2d_ceil_div(a.N, nt * 32) * nt * 32;
    if (cols < best_cols) { best_cols = cols; best_nt = nt; }
  }
  int wm = 4, wn = 1, mt = 1;
  const int f = pm ? force_nt_pm : force_nt;
  if (f >= 2 && f <= 6) best_nt = f;
  if (f == 22 || f == 24) { wm = 2; wn = 2; mt = 2; best_nt = f - 20; }      // (the 2 x 2 wave forms)
  if (f == 42) { wm = 2; wn = 4; mt = 2; best_nt = 2; }
  int mtiles = 0;
  const int rc = launch_igemm_x9_ring(a, wm, wn, mt, best_nt, pm, s, &mtiles, g_tile_query);
  g_last_m_tiles = mtiles;
  return rc;
}

int run_igemm(const IgemmArgs& a_in, hipStream_t s, const IgemmWs& ws_in = IgemmWs{nullptr, 0}) {
  IgemmArgs a = a_in;
  if (a.M <= 0 || a.N <= 0) { g_last_m_tiles = 0; return g_collect ? C2D_ERR_UNSUPPORTED : C2D_OK; }
  // fp32 operands whose weights have bf16 planes bound: nine partial products on the bf16 pipe
  const bool x9 = !g_collect && x9_resolve(&a);
  const IgemmWs ws = x9 ? IgemmWs{nullptr, 0} : ws_in;
  // Pixel-major rows for multi-tap convolutions over small per-ROI maps (see decompose<true>).
  const int hw = a.g.rh * a.g.rw;
  // Tuning hooks for tools/sweep_igemm.py (read only when C2D_TUNE is set at load time):
  // igemm_row_major=1 disables the pixel-major order, igemm_cfg=2|3 forces the 128x64 /
  // 128x128 tile, igemm_sk=0 the one-tile-per-block form.
  static const bool tune = c2d_tune_on();
  bool row_major_only = false;
  int force = 0;   // 2: 128x64, 3: 128x128, 4-6: bf16-only forms (see below)
  if (tune) {
    const char* e = c2d_tune_get("igemm_row_major");
    row_major_only = e && e[0] == '1';
    e = c2d_tune_get("igemm_cfg");
    force = e ? atoi(e) : 0;
  }
  // bf16 (igemm_ring_kernel<..., 2>) block tile by output width, measured per GEMM call of the step
  // (tools/bench_step_gemms.py + tools/sweep_step_gemms.sh, N = 2000 ROIs; round 3):
  //   129..192 and 257..384 columns: ONE 128-row x full-width tile per block (8 waves 4 x 2): the
  //     activations stream from HBM once; with three 64-column tiles the ring is 80 KiB and two
  //     workgroups share a CU (192->256 input gradient on 7x7 maps: 130 -> 107 us, 192->320: 47 -> 46,
  //     128->192 stride 2: 29 -> 26);
  //   193..256 columns: 128x256 (8 waves 2 x 4);
  //   wider (block-entry GEMMs: 736 / 832 columns forward, 576 / 1024 input gradient): 128x256 unless
  //     the last 256-wide tile would be more than 35 % padding (the 576-wide entry gradient, a
  //     third padding, reads its operand three times instead of nine: 122 -> 105 us cold, 87.5 ->
  //     81.5 warm);
  //   up to 128 columns: 128x64 / 64x64 as before.
  if (!force && a.es == 2) {
    if (a.N > 256 && a.N <= 384) force = 7;
    else if (a.N > 192 && a.N <= 256) force = 6;
    else if (a.N > 128 && a.N <= 192) force = 7;
    else if (a.N > 384) {
      const int waste = c2d_ceil_div(a.N, 256) * 256 - a.N;
      force = waste * 20 <= 7 * a.N ? 6 : 2;
    } else {
      force = 2;
    }
  }
  // bf16 operands: 128x64 tiles throughout (tools/bench_conv_bf16.py: 0.96 ms against 1.06 ms on
  // the second-stage shapes; the stride-2 input gradients gain most, 108 -> 66 us)
  // stride-2 input gradients (four parity-class launches of a quarter of the rows each): 128x64
  // as well, the smaller launches fill the chip better (256->256 on 7x7: 310 -> 254 us)
  const bool narrow = force ? force == 2
                            : (a.es == 2 || a.g.sub > 1 || (a.N % 128 != 0 && a.N % 128 <= 64));
  if (!row_major_only && a.nseg == 1 && a.g.kh * a.g.kw > 1 && hw <= 64 && a.g.nimg >= 64 &&
      a.N % 4 == 0 && a.g.sub == 1) {
    if (g_collect) return C2D_ERR_UNSUPPORTED;
    // image groups of 128 (= the rows of every pixel-major block tile: all of a block's MFMA tiles
    // are then ONE pixel, see decompose<true>) unless rounding the image count up to 128 would add
    // more than 10 % of empty rows; C2D_TUNE=pm_group=32 keeps round 3's groups of 32
    static const int pm_force = (tune && c2d_tune_get("pm_group")) ? atoi(c2d_tune_get("pm_group")) : 0;
    const int n128 = c2d_ceil_div(a.g.nimg, 128) * 128;
    // (the stream-K form — a workspace was passed — keeps groups of 32: its cost model,
    // make_sk_plan, walks the pixels of a block's four tiles)
    a.g.pm = (ws.ptr || pm_force == 32 || (pm_force != 128 && n128 * 10 > a.g.nimg * 11)) ? 5 : 7;
    a.M = c2d_ceil_div(a.g.nimg, 1 << a.g.pm) * (1 << a.g.pm) * hw;
    // heavy pixels first (ConvGeom::lpt_ngx): every non-tuning pixel-major tile has 128 rows = one
    // pixel of one image group; the image groups must split evenly over the 8 XCDs
    static const bool lpt_off = tune && c2d_tune_get("pm_lpt") && c2d_tune_get("pm_lpt")[0] == '0';
    const int ngrp = c2d_ceil_div(a.g.nimg, 128);
    a.g.lpt_ngx = 0;
    if (a.g.pm == 7 && ngrp % 8 == 0 && !lpt_off && force != 4 && force != 5 && !ws.ptr) {
      int cnt[64];
      for (int px = 0; px < hw; ++px) {
        const int y = px / a.g.rw, x = px % a.g.rw;
        cnt[px] = 0;
        for (int ty = 0; ty < a.g.nky; ++ty)
          for (int tx = 0; tx < a.g.nkx; ++tx) {
            const int ky = a.g.ky0 + a.g.kstep * ty, kx = a.g.kx0 + a.g.kstep * tx;
            cnt[px] += (a.g.mode == 0 ? tap_ok<0>(a.g, y, x, ky, kx) : tap_ok<1>(a.g, y, x, ky, kx)) ? 1 : 0;
          }
      }
      int n = 0;
      for (int c = a.g.nky * a.g.nkx; c >= 0; --c)          // stable, by falling tap count
        for (int px = 0; px < hw; ++px)
          if (cnt[px] == c) a.g.px_order[n++] = (unsigned char)px;
      a.g.lpt_ngx = ngrp / 8;
    }
    // 128x64 tiles (4 waves per SIMD) measured best or within 3 % of best on every 3x3 layer of
    // the second stage (tools/sweep_igemm.py); 128x128 only when forced by the tuning hook.
    if (force == 4 && a.es == 2) {      // 256x128 block, 128x64 per wave (bf16 only)
      if (a.g.mode == 0) return launch_igemm_bf16<0, 2, 2, 4, 2, true>(a, s);
      return launch_igemm_bf16<1, 2, 2, 4, 2, true>(a, s);
    }
    if (force == 5 && a.es == 2) {      // 256x128 block, 8 waves of 64x64
      if (a.g.mode == 0) return launch_igemm_bf16<0, 4, 2, 2, 2, true>(a, s);
      return launch_igemm_bf16<1, 4, 2, 2, 2, true>(a, s);
    }
    if (force == 6 && a.es == 2) {      // 128x256 block, 8 waves of 64x64
      if (a.g.mode == 0) return launch_igemm_bf16<0, 2, 4, 2, 2, true>(a, s);
      return launch_igemm_bf16<1, 2, 4, 2, 2, true>(a, s);
    }
    if (force == 7 && a.es == 2) return launch_igemm_bf16_wide<true>(a, s);
    // (x9: one-pixel blocks only — its stage loop has no per-tile tap test)
    if (x9 && a.g.pm == 7) return x9_launch(a, true, s);
    if (force != 3) return launch_igemm<2, 2, 2, 1, 32, true>(a, s, ws);
    return launch_igemm<2, 2, 2, 2, 32, true>(a, s, ws);
  }
  // Tile choice: big tiles when the grid still fills 256 CUs, otherwise 64x64 tiles.
  const long long big_blocks = (long long)c2d_ceil_div(a.M, 128) * c2d_ceil_div(a.N, 128);
  // (a deep 1x1 GEMM over a few thousand rows — the 416-column heads of the 80-class configs,
  //  2000 x 416 x 1024: 0.85 GFLOP — is better off on 64x64 tiles through LDS than on the
  //  one-tile-per-block kernel, whose fragments come straight from memory: forward 37.8 -> 29.2 us;
  //  the 112-column heads of the 20-class configs stay: 12.7 against 27.7 us)
  static const bool keep_small = tune && c2d_tune_get("igemm_keep_small") != nullptr;
  const bool deep_1x1 = a.g.kh * a.g.kw == 1 && a.M >= 1024 &&
                        (long long)a.M * a.N * a.K >= 600000000ll && !keep_small;
  // (up to 8192 rows: the 10,584-row maps of two 1000-px images — Mixed_4a-e at the reference's
  //  as-shipped operating point — are better off on the 64x64 LDS tiles: 3.54 -> 3.17 ms per bf16
  //  step of two images, 9.04 -> 8.78 fp32; the benchmark's 500x500 image keeps its 1024- and
  //  3969-row layers here and moves Conv2d_2b/2c, 15,625 rows: 3.05 -> 3.02 / 10.83 -> 10.84 ms.
  //  C2D_TUNE=igemm_small_max_m=<rows>)
  static const int small_max_m = (tune && c2d_tune_get("igemm_small_max_m")) ? atoi(c2d_tune_get("igemm_small_max_m")) : 8192;
  const int cus = c2d_available_cus();
  if (big_blocks < cus && a.nseg == 1 && a.N % 4 == 0 && a.M <= small_max_m && !a.fy && !a.mo_n && !deep_1x1) {
    // small problems (first stage): one 32x32 tile per block, K split over the 4 waves
    IgemmArgs b = a;
    b.m_tiles = c2d_ceil_div(a.M, 32);
    b.n_tiles = c2d_ceil_div(a.N, 32);
    if (g_collect) {          // grouped launch being assembled (run_small_group)
      if (g_collect->num < SMALL_GROUP_MAX &&
          (g_collect->num == 0 || (g_collect->mode == a.g.mode && g_collect->es == a.es))) {
        g_collect->args.a[g_collect->num++] = b;
        g_collect->mode = a.g.mode;
        g_collect->es = a.es;
        return C2D_OK;
      }
      return C2D_ERR_UNSUPPORTED;
    }
    const dim3 grid(b.m_tiles * b.n_tiles), block(256);
    if (a.es == 2) {
      dispatch_note("igemm_small_kernel<%d, 2>", a.g.mode);
      if (a.g.mode == 0) hipLaunchKernelGGL((igemm_small_kernel<0, 2>), grid, block, 0, s, b);
      else hipLaunchKernelGGL((igemm_small_kernel<1, 2>), grid, block, 0, s, b);
      return c2d_launch_status();
    }
    dispatch_note("igemm_small_kernel<%d, 4>", a.g.mode);
    if (a.g.mode == 0) hipLaunchKernelGGL(igemm_small_kernel<0>, grid, block, 0, s, b);
    else hipLaunchKernelGGL(igemm_small_kernel<1>, grid, block, 0, s, b);
    return c2d_launch_status();
  } else if (g_collect) {
    return C2D_ERR_UNSUPPORTED;     // not a small problem: the caller launches it on its own
  } else if (big_blocks < cus) {
    return launch_igemm<2, 2, 1, 1, 32>(a, s, IgemmWs{nullptr, 0});          // 64x64 tiles
  } else if (force == 4 && a.es == 2) {
    if (a.g.mode == 0) return launch_igemm_bf16<0, 2, 2, 4, 2, false>(a, s);
    return launch_igemm_bf16<1, 2, 2, 4, 2, false>(a, s);
  } else if (force == 5 && a.es == 2) {
    if (a.g.mode == 0) return launch_igemm_bf16<0, 4, 2, 2, 2, false>(a, s);
    return launch_igemm_bf16<1, 4, 2, 2, 2, false>(a, s);
  } else if (force == 6 && a.es == 2) {
    if (a.g.mode == 0) return launch_igemm_bf16<0, 2, 4, 2, 2, false>(a, s);
    return launch_igemm_bf16<1, 2, 4, 2, 2, false>(a, s);
  } else if (force == 7 && a.es == 2) {
    return launch_igemm_bf16_wide<false>(a, s);
  } else if (x9) {
    return x9_launch(a, false, s);
  } else if (narrow) {
    // N = 192, 320, 576, 160 ...: 128x64 tiles (waves 2x2, each 64x32) waste at most half a
    // 64-wide tile instead of half a 128-wide one, and the smaller accumulator file leaves room
    // for a fourth wave per SIMD.
    return launch_igemm<2, 2, 2, 1, 32>(a, s, ws);
  }
  return launch_igemm<2, 2, 2, 2, 32>(a, s, ws);          // 128x128, waves 2x2
}

}  // namespace

namespace c2d_ig {
void dispatch_note_ext(const char* fmt, int a, int b, int c, int d, int e, int f, int g, int h) {
  dispatch_note(fmt, a, b, c, d, e, f, g, h);
}
}  // namespace c2d_ig

#ifdef C2D_TRACE
// Diagnostic build only (make trace): per-block timeline buffer, 8 x u64 per block.
extern "C" int c2d_debug_set_trace(void* buf) { g_trace = (unsigned long long*)buf; return C2D_OK; }
#endif

extern "C" int c2d_debug_last_dispatch(char* buf, int len) {
  C2D_CHECK_ARG(buf && len > 0);
  int pos = 0;
  buf[0] = 0;
  for (int i = 0; i < g_ndispatch; ++i) {
    const DispatchRec& r = g_dispatch[i];
    const int n = snprintf(buf + pos, (size_t)(len - pos), r.fmt, r.p[0], r.p[1], r.p[2], r.p[3],
                           r.p[4], r.p[5], r.p[6], r.p[7]);
    if (n < 0 || pos + n + 2 >= len) return C2D_ERR_WORKSPACE;
    pos += n;
    if (i + 1 < g_ndispatch) buf[pos++] = ';';
    buf[pos] = 0;
  }
  return C2D_OK;
}

extern "C" long long c2d_conv_workspace_bytes(void) {
  // <= 1024 resident workgroups x 2 slabs of a 128x128 fp32 tile + the tile counters
  return 1024ll * 2 * 128 * 128 * 4 + (4ll << 20);
}

static int conv_fwd_impl(const float* x, int ldx, int xoff, const float* wt,
                         const float* scale, const float* shift, float* y, int ldy,
                         int yoff, int n, int ih, int iw, int cin, int cout, int kh, int kw,
                         int stride, int relu, IgemmWs ws, void* stream, int es = 4) {
  dispatch_reset();
  C2D_CHECK_ARG(x && wt && y && n > 0 && cin > 0 && cout > 0);
  C2D_CHECK_ARG(cin % 16 == 0 && ldx % 4 == 0 && xoff % 4 == 0);
  IgemmArgs a = {};
  int rc = fill_geom(&a.g, ih, iw, kh, kw, stride, 0);
  if (rc) return rc;
  a.A = x; a.lda = ldx; a.a_off = xoff; a.Bt = wt; a.C = y; a.ldc = ldy; a.c_off = yoff;
  a.scale = scale; a.shift = shift; a.relu = relu; a.accumulate = 0; a.nseg = 1;
  a.M = n * a.g.oh * a.g.ow; a.N = cout; a.K = cin; a.g.nimg = n; a.es = es;
  a.a_rows = (long long)n * ih * iw;
  // (bf16: 16-byte loads and stores = 8 elements)
  C2D_CHECK_ARG(kh * kw <= 64 && (es == 4 || (ldx % 8 == 0 && xoff % 8 == 0 && ldy % 8 == 0 && yoff % 8 == 0 && cout % 8 == 0)));
  C2D_CHECK_ARG(a.a_rows * ldx * 4 < (long long)OOB_OFFSET && (long long)kh * kw * cin * cout * 4 < (long long)OOB_OFFSET);
  return run_igemm(a, (hipStream_t)stream, ws);
}

extern "C" int c2d_conv_fwd(const float* x, int ldx, int xoff, const float* wt,
                            const float* scale, const float* shift, float* y, int ldy,
                            int yoff, int n, int ih, int iw, int cin, int cout, int kh, int kw,
                            int stride, int relu, void* stream) {
  return conv_fwd_impl(x, ldx, xoff, wt, scale, shift, y, ldy, yoff, n, ih, iw, cin, cout, kh, kw,
                       stride, relu, IgemmWs{nullptr, 0}, stream);
}

extern "C" int c2d_conv_fwd_ws(const float* x, int ldx, int xoff, const float* wt,
                               const float* scale, const float* shift, float* y, int ldy,
                               int yoff, int n, int ih, int iw, int cin, int cout, int kh, int kw,
                               int stride, int relu, void* workspace, long long workspace_bytes,
                               void* stream) {
  C2D_CHECK_ARG(workspace && workspace_bytes > 0);
  return conv_fwd_impl(x, ldx, xoff, wt, scale, shift, y, ldy, yoff, n, ih, iw, cin, cout, kh, kw,
                       stride, relu, IgemmWs{workspace, workspace_bytes}, stream);
}

// Several 1x1 / stride-1 convolutions of one input as one forward GEMM (IgemmArgs::mo_n).
static_assert(sizeof(C2dConvOut) == 48, "C2dConvOut layout");
static int conv1x1_fwd_multi_impl(const void* x, int ldx, int xoff, int nout, const C2dConvOut* outs,
                                  int rows, int cin, void* stream, int es) {
  dispatch_reset();
  C2D_CHECK_ARG(x && outs && nout >= 1 && nout <= 4 && rows > 0 && cin > 0);
  C2D_CHECK_ARG(cin % 16 == 0 && ldx % (16 / es) == 0 && xoff % (16 / es) == 0);   // (as c2d_conv_fwd)
  IgemmArgs a = {};
  int rc = fill_geom(&a.g, 1, 1, 1, 1, 1, 0);
  if (rc) return rc;
  const char* base = nullptr; const char* top = nullptr;
  int ntot = 0;
  for (int s = 0; s < nout; ++s) {
    const C2dConvOut& o = outs[s];
    C2D_CHECK_ARG(o.wt && o.scale && o.shift && o.dst && o.cout > 0 && o.cout % 4 == 0);
    C2D_CHECK_ARG(o.ld_dst % 4 == 0 && o.off_dst % 4 == 0);
    C2D_CHECK_ARG(es == 4 || (o.cout % 8 == 0 && o.ld_dst % 8 == 0 && o.off_dst % 8 == 0));
    const char* w = (const char*)o.wt;
    if (!base || w < base) base = w;
    if (!top || w + (long long)o.cout * cin * es > top) top = w + (long long)o.cout * cin * es;
    ntot += o.cout;
  }
  C2D_CHECK_ARG(top - base < (long long)OOB_OFFSET);
  a.mo_n = nout; a.mo_bbytes = top - base;
  int end = 0;
  for (int s = 0; s < nout; ++s) {
    const C2dConvOut& o = outs[s];
    C2D_CHECK_ARG(((const char*)o.wt - base) % es == 0);
    end += o.cout;
    a.mo_end[s] = end; a.mo_relu[s] = o.relu;
    a.mo_boff[s] = ((const char*)o.wt - base) / es;
    a.mo_C[s] = (float*)o.dst; a.mo_ldc[s] = o.ld_dst; a.mo_coff[s] = o.off_dst;
    a.mo_scale[s] = o.scale; a.mo_shift[s] = o.shift;
  }
  a.A = (const float*)x; a.lda = ldx; a.a_off = xoff; a.a_rows = rows;
  a.Bt = (const float*)base;
  a.C = (float*)outs[0].dst; a.ldc = outs[0].ld_dst; a.c_off = outs[0].off_dst;
  a.scale = nullptr; a.shift = nullptr; a.relu = 0; a.accumulate = 0; a.nseg = 1;
  a.M = rows; a.N = ntot; a.K = cin; a.g.nimg = rows; a.es = es;
  C2D_CHECK_ARG((long long)rows * ldx * es < (long long)OOB_OFFSET);
  return run_igemm(a, (hipStream_t)stream, IgemmWs{nullptr, 0});
}

extern "C" int c2d_conv1x1_fwd_multi(const float* x, int ldx, int xoff, int nout,
                                     const C2dConvOut* outs, int rows, int cin, void* stream) {
  return conv1x1_fwd_multi_impl(x, ldx, xoff, nout, outs, rows, cin, stream, 4);
}
extern "C" int c2d_conv1x1_fwd_multi_bf16(const void* x, int ldx, int xoff, int nout,
                                          const C2dConvOut* outs, int rows, int cin, void* stream) {
  return conv1x1_fwd_multi_impl(x, ldx, xoff, nout, outs, rows, cin, stream, 2);
}

// The BN/ReLU backward fused into an input-gradient launch (IgemmArgs::fy ...); blocks_out: the
// number of partial-sum rows the launch(es) wrote.  query: count the rows only, launch nothing.
struct FusedBn {
  const void* y; int ldy, yoff;
  int nprod; int seg_end[4]; int ident[4];
  const float* scale[4]; const float* beta[4]; const float* gamma[4];
  float* part;
  int* blocks_out;
  bool query;
};

static int fused_bn_fill(IgemmArgs* a, const FusedBn* fb, int ncols) {
  const bool query = fb->query;
  C2D_CHECK_ARG(fb->nprod >= 1 && fb->nprod <= 4 && fb->seg_end[fb->nprod - 1] == ncols);
  C2D_CHECK_ARG(query || (fb->y && fb->part && fb->ldy % 4 == 0 && fb->yoff % 4 == 0));
  // (any non-null pointer in a query: only the tile choice is evaluated)
  a->fy = query ? reinterpret_cast<const void*>(16) : fb->y;
  a->fldy = fb->ldy; a->fyoff = fb->yoff; a->fnprod = fb->nprod; a->fpart = fb->part;
  for (int p = 0; p < fb->nprod; ++p) {
    C2D_CHECK_ARG(fb->seg_end[p] % 4 == 0 && fb->seg_end[p] > (p ? fb->seg_end[p - 1] : 0));
    C2D_CHECK_ARG(query || fb->ident[p] || (fb->scale[p] && (!fb->gamma[p] || fb->beta[p])));
    a->fseg_end[p] = fb->seg_end[p]; a->fident[p] = fb->ident[p];
    a->fscale[p] = fb->scale[p]; a->fbeta[p] = fb->beta[p]; a->fgamma[p] = fb->gamma[p];
  }
  return C2D_OK;
}

static int conv_dgrad_impl(const float* dc, int ldc, int coff, const float* w, float* dx,
                           int lddx, int dxoff, int n, int ih, int iw, int cin, int cout,
                           int kh, int kw, int stride, int accumulate, IgemmWs ws, void* stream,
                           int es = 4, const FusedBn* fb = nullptr) {
  dispatch_reset();
  const bool query = fb && fb->query;
  if (query) { dc = w = reinterpret_cast<const float*>(16); dx = reinterpret_cast<float*>(16); }
  C2D_CHECK_ARG(dc && w && dx && n > 0 && cin > 0 && cout > 0);
  C2D_CHECK_ARG(cout % 16 == 0 && ldc % 4 == 0 && coff % 4 == 0);
  IgemmArgs a = {};
  int rc = fill_geom(&a.g, ih, iw, kh, kw, stride, 1);
  if (rc) return rc;
  a.A = dc; a.lda = ldc; a.a_off = coff; a.Bt = w; a.C = dx; a.ldc = lddx; a.c_off = dxoff;
  a.scale = nullptr; a.shift = nullptr; a.relu = 0; a.accumulate = accumulate; a.nseg = 1;
  a.N = cin; a.K = cout; a.g.nimg = n; a.es = es;
  a.a_rows = (long long)n * a.g.oh * a.g.ow;
  C2D_CHECK_ARG(kh * kw <= 64 && (es == 4 || (ldc % 8 == 0 && coff % 8 == 0 && lddx % 8 == 0 && dxoff % 8 == 0 && cin % 8 == 0)));
  C2D_CHECK_ARG(a.a_rows * ldc * 4 < (long long)OOB_OFFSET && (long long)kh * kw * cin * cout * 4 < (long long)OOB_OFFSET);
  int blocks = 0;
  if (fb) {
    C2D_CHECK_ARG(!accumulate && cin % 4 == 0);
    rc = fused_bn_fill(&a, fb, cin);
    if (rc) return rc;
    ws = IgemmWs{nullptr, 0};          // one tile per block: every block owns whole column sums
  }
  struct QueryScope {                  // (run_igemm only records the tile choice while this is set)
    bool on;
    explicit QueryScope(bool q) : on(q) { if (on) g_tile_query = true; }
    ~QueryScope() { if (on) g_tile_query = false; }
  } scope(query);
  if (stride == 1) {
    a.M = n * ih * iw;
    rc = run_igemm(a, (hipStream_t)stream, ws);
    if (fb && fb->blocks_out) *fb->blocks_out = g_last_m_tiles;
    return rc;
  }
  // stride 2: one launch per parity class (py, px) of the input pixel; a pixel of the class
  // only meets the taps with ky = (py + pad_t) mod 2 (+2, ...), i.e. 2.25 taps per pixel on
  // average for a 3x3 kernel instead of 9 mostly-empty ones.
  // bf16: the four classes are 141-250 workgroups each — less than one per CU, most of a launch ramp
  // and epilogue: they leave as ONE grouped launch (igemm_bf16.hip: ring_group_begin / _end) when
  // they pick the same ring instance.  C2D_TUNE=dgrad_s2_group=0: four launches.
  static const bool group_off = c2d_tune_on() && c2d_tune_get("dgrad_s2_group") &&
                                c2d_tune_get("dgrad_s2_group")[0] == '0';
  const bool grouped = es == 2 && !query && !ws.ptr && !group_off;
  if (grouped) ring_group_begin();
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px) {
      IgemmArgs b = a;
      b.g.sub = 2; b.g.y0 = py; b.g.x0 = px;
      b.g.rh = (ih - py + 1) / 2; b.g.rw = (iw - px + 1) / 2;
      if (b.g.rh <= 0 || b.g.rw <= 0) continue;
      b.g.kstep = 2;
      b.g.ky0 = (py + b.g.pad_t) & 1; b.g.kx0 = (px + b.g.pad_l) & 1;
      b.g.nky = b.g.ky0 < kh ? (kh - b.g.ky0 + 1) / 2 : 0;
      b.g.nkx = b.g.kx0 < kw ? (kw - b.g.kx0 + 1) / 2 : 0;
      if (b.g.nky == 0 || b.g.nkx == 0) { b.g.nky = 0; b.g.nkx = 0; }
      set_magic(&b.g);
      b.M = n * b.g.rh * b.g.rw;
      b.fpart_row0 = blocks;                        // (fused: each class owns its rows of partials)
      rc = run_igemm(b, (hipStream_t)stream, ws);   // (launches of one stream run in order: the
      if (rc) {                                      //  four classes share the workspace)
        if (grouped) ring_group_end((hipStream_t)stream);
        return rc;
      }
      blocks += g_last_m_tiles;
    }
  if (grouped) {
    rc = ring_group_end((hipStream_t)stream);
    if (rc) return rc;
  }
  if (fb && fb->blocks_out) *fb->blocks_out = blocks;
  return C2D_OK;
}

// Fused forms: dc_out [n*ih*iw, cin] dense (row stride cin) = the dc of the producer layer.
template <typename T>
static int conv_dgrad_bn_relu_impl(const T* dc, int ldc, int coff, const T* w, const T* y, int ldy,
                                   int yoff, const float* scale, const float* beta,
                                   const float* gamma, T* dc_out, float* partials, int n, int ih,
                                   int iw, int cin, int cout, int kh, int kw, int stride,
                                   void* stream, int* blocks_out, bool query) {
  FusedBn fb = {};
  fb.y = y; fb.ldy = ldy; fb.yoff = yoff; fb.nprod = 1; fb.seg_end[0] = cin;
  fb.scale[0] = scale; fb.beta[0] = beta; fb.gamma[0] = gamma;
  fb.part = partials; fb.blocks_out = blocks_out; fb.query = query;
  return conv_dgrad_impl(reinterpret_cast<const float*>(dc), ldc, coff,
                         reinterpret_cast<const float*>(w), reinterpret_cast<float*>(dc_out), cin, 0,
                         n, ih, iw, cin, cout, kh, kw, stride, 0, IgemmWs{nullptr, 0}, stream,
                         (int)sizeof(T), &fb);
}

extern "C" int c2d_conv_dgrad_bn_relu(const float* dc, int ldc, int coff, const float* w,
                                      const float* y, int ldy, int yoff, const float* scale,
                                      const float* beta, const float* gamma, float* dc_out,
                                      float* partials, int n, int ih, int iw, int cin, int cout,
                                      int kh, int kw, int stride, void* stream) {
  return conv_dgrad_bn_relu_impl<float>(dc, ldc, coff, w, y, ldy, yoff, scale, beta, gamma, dc_out,
                                        partials, n, ih, iw, cin, cout, kh, kw, stride, stream,
                                        nullptr, false);
}

extern "C" int c2d_conv_dgrad_bn_relu_bf16(const void* dc, int ldc, int coff, const void* w,
                                           const void* y, int ldy, int yoff, const float* scale,
                                           const float* beta, const float* gamma, void* dc_out,
                                           float* partials, int n, int ih, int iw, int cin,
                                           int cout, int kh, int kw, int stride, void* stream) {
  // (the fused epilogue reads the producer's y in 16-byte chunks: 8 bf16)
  C2D_CHECK_ARG(ldy % 8 == 0 && yoff % 8 == 0);
  return conv_dgrad_bn_relu_impl<c2d_bf16>((const c2d_bf16*)dc, ldc, coff, (const c2d_bf16*)w,
                                           (const c2d_bf16*)y, ldy, yoff, scale, beta, gamma,
                                           (c2d_bf16*)dc_out, partials, n, ih, iw, cin, cout, kh, kw,
                                           stride, stream, nullptr, false);
}

extern "C" int c2d_conv_dgrad_bn_relu_partial_blocks(int elem_size, int n, int ih, int iw, int cin,
                                                     int cout, int kh, int kw, int stride) {
  int blocks = 0, rc;
  if (elem_size == 2)
    rc = conv_dgrad_bn_relu_impl<c2d_bf16>(nullptr, cout, 0, nullptr, nullptr, cin, 0, nullptr,
                                           nullptr, nullptr, nullptr, nullptr, n, ih, iw, cin, cout,
                                           kh, kw, stride, nullptr, &blocks, true);
  else
    rc = conv_dgrad_bn_relu_impl<float>(nullptr, cout, 0, nullptr, nullptr, cin, 0, nullptr, nullptr,
                                        nullptr, nullptr, nullptr, n, ih, iw, cin, cout, kh, kw,
                                        stride, nullptr, &blocks, true);
  return rc == C2D_OK ? blocks : -1;
}

extern "C" int c2d_conv_dgrad(const float* dc, int ldc, int coff, const float* w, float* dx,
                              int lddx, int dxoff, int n, int ih, int iw, int cin, int cout,
                              int kh, int kw, int stride, int accumulate, void* stream) {
  return conv_dgrad_impl(dc, ldc, coff, w, dx, lddx, dxoff, n, ih, iw, cin, cout, kh, kw, stride,
                         accumulate, IgemmWs{nullptr, 0}, stream);
}

extern "C" int c2d_conv_dgrad_ws(const float* dc, int ldc, int coff, const float* w, float* dx,
                                 int lddx, int dxoff, int n, int ih, int iw, int cin, int cout,
                                 int kh, int kw, int stride, int accumulate, void* workspace,
                                 long long workspace_bytes, void* stream) {
  C2D_CHECK_ARG(workspace && workspace_bytes > 0);
  return conv_dgrad_impl(dc, ldc, coff, w, dx, lddx, dxoff, n, ih, iw, cin, cout, kh, kw, stride,
                         accumulate, IgemmWs{workspace, workspace_bytes}, stream);
}

static int dgrad_multi_impl(int nseg, const float* const* dcs, const int* ldcs,
                            const int* coffs, const float* const* ws,
                            const int* couts, float* dx, int lddx, int dxoff,
                            int rows, int cin, int accumulate, IgemmWs wsp, void* stream,
                            int es = 4, const FusedBn* fb = nullptr) {
  dispatch_reset();
  const float* fake_ptrs[4]; int fake_ints[4] = {0, 0, 0, 0};
  if (fb && fb->query) {       // only the tile choice is evaluated
    for (int i = 0; i < 4; ++i) fake_ptrs[i] = reinterpret_cast<const float*>(16);
    dcs = ws = fake_ptrs; coffs = fake_ints; dx = reinterpret_cast<float*>(16);
  }
  C2D_CHECK_ARG(nseg >= 1 && nseg <= 4 && dcs && ldcs && coffs && ws && couts && dx);
  C2D_CHECK_ARG(rows > 0 && cin > 0 && cin % 4 == 0 && lddx % 4 == 0 && dxoff % 4 == 0);
  IgemmArgs a = {};
  int rc = fill_geom(&a.g, 1, 1, 1, 1, 1, 1);
  if (rc) return rc;
  a.total_slabs = 0;
  for (int i = 0; i < nseg; ++i) {
    C2D_CHECK_ARG(dcs[i] && ws[i] && couts[i] > 0 && couts[i] % 16 == 0);
    C2D_CHECK_ARG(ldcs[i] % 4 == 0 && coffs[i] % 4 == 0);
    a.segA[i] = dcs[i]; a.segB[i] = ws[i]; a.seg_lda[i] = ldcs[i]; a.seg_off[i] = coffs[i];
    a.segK[i] = couts[i];
    a.total_slabs += (couts[i] + BK - 1) / BK;
    C2D_CHECK_ARG((long long)rows * ldcs[i] * 4 < (long long)OOB_OFFSET);
  }
  a.a_rows = rows;
  a.nseg = nseg; a.es = es;
  for (int i = 0; i < nseg && es == 2; ++i) C2D_CHECK_ARG(ldcs[i] % 8 == 0 && coffs[i] % 8 == 0);
  C2D_CHECK_ARG(es == 4 || (cin % 8 == 0 && lddx % 8 == 0 && dxoff % 8 == 0));
  a.A = dcs[0]; a.lda = ldcs[0]; a.a_off = coffs[0]; a.Bt = ws[0]; a.K = couts[0];
  a.C = dx; a.ldc = lddx; a.c_off = dxoff;
  a.scale = nullptr; a.shift = nullptr; a.relu = 0; a.accumulate = accumulate;
  a.M = rows; a.N = cin;
  if (!fb) return run_igemm(a, (hipStream_t)stream, wsp);
  rc = fused_bn_fill(&a, fb, cin);
  if (rc) return rc;
  if (fb->query) g_tile_query = true;
  rc = run_igemm(a, (hipStream_t)stream, IgemmWs{nullptr, 0});
  g_tile_query = false;
  if (fb->blocks_out) *fb->blocks_out = g_last_m_tiles;
  return rc;
}

static_assert(sizeof(C2dBnProducer) == 32, "C2dBnProducer layout");

static int fused_from_producers(FusedBn* fb, const void* y, int ldy, int yoff, int nprod,
                                const C2dBnProducer* prods, float* partials) {
  C2D_CHECK_ARG(nprod >= 1 && nprod <= 4 && prods);
  fb->y = y; fb->ldy = ldy; fb->yoff = yoff; fb->nprod = nprod; fb->part = partials;
  int end = 0;
  for (int p = 0; p < nprod; ++p) {
    C2D_CHECK_ARG(prods[p].width > 0);
    end += prods[p].width;
    fb->seg_end[p] = end; fb->ident[p] = prods[p].identity;
    fb->scale[p] = prods[p].scale; fb->beta[p] = prods[p].beta; fb->gamma[p] = prods[p].gamma;
  }
  return C2D_OK;
}

extern "C" int c2d_conv1x1_dgrad_multi_bn_relu(int nseg, const float* const* dcs, const int* ldcs,
                                               const int* coffs, const float* const* ws,
                                               const int* couts, const float* y, int ldy, int yoff,
                                               int nprod, const C2dBnProducer* prods, float* dx,
                                               int lddx, int dxoff, float* partials, int rows,
                                               int cin, int accumulate, void* stream) {
  FusedBn fb = {};
  int rc = fused_from_producers(&fb, y, ldy, yoff, nprod, prods, partials);
  if (rc) return rc;
  return dgrad_multi_impl(nseg, dcs, ldcs, coffs, ws, couts, dx, lddx, dxoff, rows, cin, accumulate,
                          IgemmWs{nullptr, 0}, stream, 4, &fb);
}

static int dgrad_multi_bn_relu_blocks(int nseg, const int* couts, int rows, int cin, int es) {
  C2D_CHECK_ARG(nseg >= 1 && nseg <= 4 && couts);
  FusedBn fb = {};
  fb.nprod = 1; fb.seg_end[0] = cin; fb.ident[0] = 1; fb.query = true;
  int blocks = 0;
  fb.blocks_out = &blocks;
  int lds[4];
  for (int i = 0; i < nseg; ++i) lds[i] = couts[i];
  const int rc = dgrad_multi_impl(nseg, nullptr, lds, nullptr, nullptr, couts, nullptr, cin, 0, rows,
                                  cin, 0, IgemmWs{nullptr, 0}, nullptr, es, &fb);
  return rc == C2D_OK ? blocks : -1;
}

extern "C" int c2d_conv1x1_dgrad_multi_bn_relu_partial_blocks(int nseg, const int* couts, int rows,
                                                              int cin) {
  return dgrad_multi_bn_relu_blocks(nseg, couts, rows, cin, 4);
}

// bf16 storage (round 5): the ring kernel's fused epilogue instance behind the same interface
extern "C" int c2d_conv1x1_dgrad_multi_bn_relu_partial_blocks_bf16(int nseg, const int* couts,
                                                                   int rows, int cin) {
  return dgrad_multi_bn_relu_blocks(nseg, couts, rows, cin, 2);
}

extern "C" int c2d_conv1x1_dgrad_multi_bn_relu_bf16(int nseg, const void* const* dcs, const int* ldcs,
                                                    const int* coffs, const void* const* ws,
                                                    const int* couts, const void* y, int ldy,
                                                    int yoff, int nprod, const C2dBnProducer* prods,
                                                    void* dx, int lddx, int dxoff, float* partials,
                                                    int rows, int cin, int accumulate, void* stream) {
  FusedBn fb = {};
  int rc = fused_from_producers(&fb, y, ldy, yoff, nprod, prods, partials);
  if (rc) return rc;
  C2D_CHECK_ARG(ldy % 8 == 0 && yoff % 8 == 0);
  for (int p = 0; p < nprod; ++p) C2D_CHECK_ARG(prods[p].width % 8 == 0);
  return dgrad_multi_impl(nseg, (const float* const*)dcs, ldcs, coffs, (const float* const*)ws, couts,
                          (float*)dx, lddx, dxoff, rows, cin, accumulate, IgemmWs{nullptr, 0}, stream,
                          2, &fb);
}

extern "C" int c2d_conv1x1_dgrad_multi(int nseg, const float* const* dcs, const int* ldcs,
                                       const int* coffs, const float* const* ws,
                                       const int* couts, float* dx, int lddx, int dxoff,
                                       int rows, int cin, int accumulate, void* stream) {
  return dgrad_multi_impl(nseg, dcs, ldcs, coffs, ws, couts, dx, lddx, dxoff, rows, cin,
                          accumulate, IgemmWs{nullptr, 0}, stream);
}

extern "C" int c2d_conv1x1_dgrad_multi_ws(int nseg, const float* const* dcs, const int* ldcs,
                                          const int* coffs, const float* const* ws,
                                          const int* couts, float* dx, int lddx, int dxoff,
                                          int rows, int cin, int accumulate, void* workspace,
                                          long long workspace_bytes, void* stream) {
  C2D_CHECK_ARG(workspace && workspace_bytes > 0);
  return dgrad_multi_impl(nseg, dcs, ldcs, coffs, ws, couts, dx, lddx, dxoff, rows, cin,
                          accumulate, IgemmWs{workspace, workspace_bytes}, stream);
}

// ---- grouped launches of independent small convolutions ---------------------------------------
static int launch_small_group(SmallCollect& c, hipStream_t st) {
  if (c.num == 0) return C2D_OK;
  int first = 0;
  for (int p = 0; p < c.num; ++p) {
    c.args.first[p] = first;
    first += c.args.a[p].m_tiles * c.args.a[p].n_tiles;
  }
  for (int p = c.num; p <= SMALL_GROUP_MAX; ++p) c.args.first[p] = first;
  c.args.num = c.num;
  if (c.es == 2) {
    dispatch_note("igemm_small_group_kernel<%d, 2>", c.mode);
    if (c.mode == 0) hipLaunchKernelGGL((igemm_small_group_kernel<0, 2>), dim3(first), dim3(256), 0, st, c.args);
    else hipLaunchKernelGGL((igemm_small_group_kernel<1, 2>), dim3(first), dim3(256), 0, st, c.args);
    return c2d_launch_status();
  }
  dispatch_note("igemm_small_group_kernel<%d, 4>", c.mode);
  if (c.mode == 0) hipLaunchKernelGGL(igemm_small_group_kernel<0>, dim3(first), dim3(256), 0, st, c.args);
  else hipLaunchKernelGGL(igemm_small_group_kernel<1>, dim3(first), dim3(256), 0, st, c.args);
  return c2d_launch_status();
}

static int conv_desc_run(const C2dConvDesc& d, int dgrad, void* stream, int es) {
  if (!dgrad)
    return conv_fwd_impl(d.src, d.ld_src, d.off_src, d.weights, d.scale, d.shift, d.dst, d.ld_dst,
                         d.off_dst, d.n, d.ih, d.iw, d.cin, d.cout, d.kh, d.kw, d.stride, d.flag,
                         IgemmWs{nullptr, 0}, stream, es);
  return conv_dgrad_impl(d.src, d.ld_src, d.off_src, d.weights, d.dst, d.ld_dst, d.off_dst, d.n,
                         d.ih, d.iw, d.cin, d.cout, d.kh, d.kw, d.stride, d.flag,
                         IgemmWs{nullptr, 0}, stream, es);
}

static int conv_grouped_impl(const C2dConvDesc* descs, int num, int dgrad, void* stream, int es = 4) {
  dispatch_reset();
  C2D_CHECK_ARG(descs && num >= 1 && num <= 64);
  // every problem of the group is "small" (run_igemm's one-tile-per-block domain): ONE launch;
  // otherwise (or more than SMALL_GROUP_MAX sub-problems) each convolution is launched on its own
  SmallCollect c;
  c.num = 0; c.mode = 0; c.es = es;
  g_collect = &c;
  int rc = C2D_OK;
  for (int i = 0; i < num && rc == C2D_OK; ++i) rc = conv_desc_run(descs[i], dgrad, stream, es);
  g_collect = nullptr;
  if (rc == C2D_OK) return launch_small_group(c, (hipStream_t)stream);
  if (rc != C2D_ERR_UNSUPPORTED) return rc;
  for (int i = 0; i < num; ++i) {
    rc = conv_desc_run(descs[i], dgrad, stream, es);
    if (rc) return rc;
  }
  return C2D_OK;
}

extern "C" int c2d_conv_fwd_grouped(const C2dConvDesc* descs, int num, void* stream) {
  return conv_grouped_impl(descs, num, 0, stream);
}

extern "C" int c2d_conv_dgrad_grouped(const C2dConvDesc* descs, int num, void* stream) {
  return conv_grouped_impl(descs, num, 1, stream);
}

extern "C" int c2d_conv_fwd_grouped_bf16(const C2dConvDesc* descs, int num, void* stream) {
  return conv_grouped_impl(descs, num, 0, stream, 2);
}

extern "C" int c2d_conv_dgrad_grouped_bf16(const C2dConvDesc* descs, int num, void* stream) {
  return conv_grouped_impl(descs, num, 1, stream, 2);
}

// ---- bf16 storage / fp32 accumulate forms (BASELINE configs[2] / [4]) -------------------------
// Same arguments as the fp32 calls; x / wt / y (dc / w / dx) are bf16 (uint16 storage),
// scale / shift stay fp32.  Strides and offsets are in ELEMENTS; ld % 8 == 0, off % 8 == 0.
extern "C" int c2d_conv_fwd_bf16(const void* x, int ldx, int xoff, const void* wt,
                                 const float* scale, const float* shift, void* y, int ldy,
                                 int yoff, int n, int ih, int iw, int cin, int cout, int kh, int kw,
                                 int stride, int relu, void* stream) {
  return conv_fwd_impl((const float*)x, ldx, xoff, (const float*)wt, scale, shift, (float*)y, ldy,
                       yoff, n, ih, iw, cin, cout, kh, kw, stride, relu, IgemmWs{nullptr, 0}, stream,
                       2);
}

extern "C" int c2d_conv_dgrad_bf16(const void* dc, int ldc, int coff, const void* w, void* dx,
                                   int lddx, int dxoff, int n, int ih, int iw, int cin, int cout,
                                   int kh, int kw, int stride, int accumulate, void* stream) {
  return conv_dgrad_impl((const float*)dc, ldc, coff, (const float*)w, (float*)dx, lddx, dxoff, n,
                         ih, iw, cin, cout, kh, kw, stride, accumulate, IgemmWs{nullptr, 0}, stream,
                         2);
}

extern "C" int c2d_conv1x1_dgrad_multi_bf16(int nseg, const void* const* dcs, const int* ldcs,
                                            const int* coffs, const void* const* ws,
                                            const int* couts, void* dx, int lddx, int dxoff,
                                            int rows, int cin, int accumulate, void* stream) {
  return dgrad_multi_impl(nseg, (const float* const*)dcs, ldcs, coffs, (const float* const*)ws,
                          couts, (float*)dx, lddx, dxoff, rows, cin, accumulate,
                          IgemmWs{nullptr, 0}, stream, 2);
}

// Tuning hooks (read only when C2D_TUNE is set at load time): wgrad_bf16_mfma=0 sends bf16
// operands through the widening fp32 kernels, wgrad3_wi forces the i-groups per block.
static bool wgrad_bf16_mfma_enabled() {
  static const bool tune = c2d_tune_on();
  if (!tune) return true;
  const char* e = c2d_tune_get("wgrad_bf16_mfma");
  return !(e && e[0] == '0');
}
// wgrad_bf16_slots=<percent>: resident-workgroup budget of the bf16 filter-gradient launches
// relative to the default (fewer workgroups = fewer split-K atomics, less latency hiding).
// Round 5: 75 % of a resident round by default.  A filter-gradient workgroup on a CU (59 KB of LDS
// for the nine-tap kernel) leaves room for ONE of the main stream's ring-GEMM workgroups instead
// of two, and those GEMMs live on co-residency (section 3b of DESIGN.md); a quarter fewer
// filter-gradient workgroups measured 2.922-2.926 against 2.945-3.002 ms per bf16 step over three
// alternating runs (85 %: 2.948-2.959, 65 %: 2.934-2.949, 50 %: 2.99).
static int wgrad_bf16_slots(int slots) {
  static const bool tune = c2d_tune_on();
  if (tune) {
    const char* e = c2d_tune_get("wgrad_bf16_slots");
    if (e && atoi(e) > 0) return c2d_cu_scaled(slots) * atoi(e) / 100;
  }
  return c2d_cu_scaled(slots) * 3 / 4;
}
static int wgrad3_bf16_igroups(int cin) {
  static const bool tune = c2d_tune_on();
  if (tune) {
    const char* e = c2d_tune_get("wgrad3_wi");
    if (e && (e[0] == '1' || e[0] == '2')) return e[0] - '0';
  }
  (void)cin;
  return 1;   // measured: two i-groups (8 waves, one block per CU) 0-15 % slower on the 4x4 / 7x7 layers
}

// partial != null: every split stores its own slab at partial +
// split * taps*cin*cout floats; *splits_out receives the number of slabs.  splits_only: just
// report how many splits the launch would use (workspace sizing), launch nothing.
template <int ES>
static int conv_wgrad_impl(const float* x, int ldx, int xoff, const float* dc, int ldc,
                           int coff, float* dw, int n, int ih, int iw, int cin, int cout,
                           int kh, int kw, int stride, void* stream, float* partial = nullptr,
                           int* splits_out = nullptr, bool splits_only = false) {
  dispatch_reset();
  if (splits_only) { x = dc = reinterpret_cast<const float*>(16); dw = reinterpret_cast<float*>(16); }
  if (partial) dw = partial;
  const long long dw_numel = (long long)kh * kw * cin * cout;
  C2D_CHECK_ARG(x && dc && dw && n > 0 && cin > 0 && cout > 0);
  C2D_CHECK_ARG(cin % 4 == 0 && cout % 4 == 0 && ldx % 4 == 0 && xoff % 4 == 0);
  C2D_CHECK_ARG(ldc % 4 == 0 && coff % 4 == 0);
  // bf16 operands with 16-B aligned rows go to the bf16-MFMA kernels; anything else is widened
  // to fp32 as it is staged (same results up to summation order)
  const bool bf16_mfma = ES == 2 && cin % 8 == 0 && cout % 8 == 0 && ldx % 8 == 0 &&
                         xoff % 8 == 0 && ldc % 8 == 0 && coff % 8 == 0 && wgrad_bf16_mfma_enabled();
  if (bf16_mfma && kh == 3 && kw == 3 && stride == 1 && ih == iw && (iw == 4 || iw == 7) &&
      cin % 32 == 0 && cout % 32 == 0 && n >= 256) {
    const int wi = wgrad3_bf16_igroups(cin);
    Wgrad3Args b;
    b.A = x; b.lda = ldx; b.a_off = xoff; b.G = dc; b.ldg = ldc; b.g_off = coff; b.dW = dw;
    b.M = n * ih * iw; b.I = cin; b.J = cout; b.h = ih; b.w = iw;
    b.itiles = c2d_ceil_div(cin, 32 * wi); b.jtiles = c2d_ceil_div(cout, 128);
    b.tiles = b.itiles * b.jtiles;
    C2D_CHECK_ARG((long long)b.M * ldx * 2 < (long long)OOB_OFFSET && (long long)b.M * ldc * 2 < (long long)OOB_OFFSET);
    const int slab = iw == 4 ? 8 * 16 : 2 * 49;            // whole images per slab
    const int nslabs = c2d_ceil_div(b.M, slab);
    // ONE round of resident blocks (see the fp32 form below), and fewer of them than fit: at
    // bf16 rates the split-K atomics (1.3 TB/s chip-wide) are 40 % of the launch, so half the
    // workgroups (half the atomic bytes) win over the extra latency hiding (tools/bench_wgrad_bf16.py)
    // K-groups per block (wgrad3x3_bf16_kernel): two groups of four waves sharing an output tile
    // measured 2-6 % faster alone and 1 % slower inside the step (4.177 against 4.127 ms): off.
    // C2D_TUNE=wgrad3_kg=1|2.
    static const int kg_env = (c2d_tune_on() && c2d_tune_get("wgrad3_kg")) ? atoi(c2d_tune_get("wgrad3_kg")) : 0;
    const int kg = wi != 1 ? 1 : (kg_env == 1 || kg_env == 2) ? kg_env : 1;
    int splits = wgrad_bf16_slots(wi != 1 || kg != 1 ? 256 : iw == 4 ? 256 : 384) / b.tiles;
    if (splits < 1) splits = 1;
    if (splits > nslabs / (4 * kg)) splits = nslabs / (4 * kg) > 0 ? nslabs / (4 * kg) : 1;
    b.rows_per_split = c2d_ceil_div(nslabs, splits * kg) * kg * slab;
    b.splits = c2d_ceil_div(b.M, b.rows_per_split);
    if (splits_out) *splits_out = b.splits;
    if (splits_only) return C2D_OK;
    b.part_stride = partial ? dw_numel : 0;
    if (partial) b.dW = partial;
    const dim3 grid(b.tiles * b.splits);
    hipStream_t st = (hipStream_t)stream;
    dispatch_note("wgrad3x3_bf16_kernel<%d, %d, %d, %d>", iw, iw == 4 ? 8 : 2, wi, kg);
    if (iw == 4 && wi == 1 && kg == 2) hipLaunchKernelGGL((wgrad3x3_bf16_kernel<4, 8, 1, 2>), grid, dim3(512), 0, st, b);
    else if (iw == 4 && wi == 1) hipLaunchKernelGGL((wgrad3x3_bf16_kernel<4, 8, 1, 1>), grid, dim3(256), 0, st, b);
    else if (iw == 4) hipLaunchKernelGGL((wgrad3x3_bf16_kernel<4, 8, 2, 1>), grid, dim3(512), 0, st, b);
    else if (wi == 1 && kg == 2) hipLaunchKernelGGL((wgrad3x3_bf16_kernel<7, 2, 1, 2>), grid, dim3(512), 0, st, b);
    else if (wi == 1) hipLaunchKernelGGL((wgrad3x3_bf16_kernel<7, 2, 1, 1>), grid, dim3(256), 0, st, b);
    else hipLaunchKernelGGL((wgrad3x3_bf16_kernel<7, 2, 2, 1>), grid, dim3(512), 0, st, b);
    return c2d_launch_status();
  }
  if (ES == 4 && !partial && !splits_only && kh == 3 && kw == 3 && ih == iw && n >= 256 &&
      ((stride == 1 && (iw == 4 || iw == 7)) || (stride == 2 && iw == 7)) && x9_active()) {
    // fp32 networks with f32x9 on: one tap per block, pixel-ordered rows, nine bf16 partial products
    // (C2D_TUNE=x9_wgrad3=0 keeps the fp32 nine-tap kernel)
    static const bool off = c2d_tune_on() && c2d_tune_get("x9_wgrad3") && atoi(c2d_tune_get("x9_wgrad3")) == 0;
    if (!off) {
      C2D_CHECK_ARG((long long)n * ih * iw * ldx * 4 < (long long)OOB_OFFSET && (long long)n * ih * iw * ldc * 4 < (long long)OOB_OFFSET);
      WgradArgs a3;
      a3.A = x; a3.lda = ldx; a3.a_off = xoff; a3.G = dc; a3.ldg = ldc; a3.g_off = coff; a3.dW = dw;
      a3.I = cin; a3.J = cout; a3.part_stride = 0; a3.rows_per_split = 0; a3.nsplits = 0;
      const int rc3 = launch_wgrad3x3_x9(a3, n, iw, stride, (hipStream_t)stream);
      if (rc3 != C2D_ERR_UNSUPPORTED) return rc3;
    }
  }
  if (kh == 3 && kw == 3 && stride == 1 && ih == iw && (iw == 4 || iw == 7) && cin % 32 == 0 &&
      cout % 32 == 0 && n >= 256) {
    Wgrad3Args b;
    b.A = x; b.lda = ldx; b.a_off = xoff; b.G = dc; b.ldg = ldc; b.g_off = coff; b.dW = dw;
    b.M = n * ih * iw; b.I = cin; b.J = cout; b.h = ih; b.w = iw;
    b.itiles = cin / 32; b.jtiles = c2d_ceil_div(cout, 128); b.tiles = b.itiles * b.jtiles;
    C2D_CHECK_ARG((long long)b.M * ldx * 4 < (long long)OOB_OFFSET && (long long)b.M * ldc * 4 < (long long)OOB_OFFSET);
    static const int pair7 = (c2d_tune_on() && c2d_tune_get("wgrad3_pair7")) ? atoi(c2d_tune_get("wgrad3_pair7")) : 1;
    const int slab = iw == 4 ? 32 : pair7 ? 98 : 49;       // whole images per slab
    const int nslabs = c2d_ceil_div(b.M, slab);
    // 2 blocks per CU in ONE round: rounding the split count UP put a handful of blocks into a
    // second round that cost half a round again (192->256 on 7x7: 516 blocks, 0.94 -> 0.64 ms)
    int splits = 512 / b.tiles;
    if (splits < 1) splits = 1;
    if (splits > nslabs / 4) splits = nslabs / 4 > 0 ? nslabs / 4 : 1;
    b.rows_per_split = c2d_ceil_div(nslabs, splits) * slab;
    b.splits = c2d_ceil_div(b.M, b.rows_per_split);
    if (splits_out) *splits_out = b.splits;
    if (splits_only) return C2D_OK;
    b.part_stride = partial ? dw_numel : 0;
    if (partial) b.dW = partial;
    const dim3 grid(b.tiles * b.splits);
    hipStream_t st = (hipStream_t)stream;
    dispatch_note("wgrad3x3_kernel<%d, %d, %d>", iw, iw == 4 || pair7 ? 2 : 1, ES);
    if (iw == 4) hipLaunchKernelGGL((wgrad3x3_kernel<4, 2, ES>), grid, dim3(256), 0, st, b);
    else if (pair7) hipLaunchKernelGGL((wgrad3x3_kernel<7, 2, ES>), grid, dim3(256), 0, st, b);
    else hipLaunchKernelGGL((wgrad3x3_kernel<7, 1, ES>), grid, dim3(256), 0, st, b);
    return c2d_launch_status();
  }
  if (!bf16_mfma && kh == 3 && kw == 3 && stride == 2 && ih == 7 && iw == 7 && cin % 32 == 0 &&
      cout % 32 == 0 && n >= 256) {
    // the two stride-2 convolutions of Mixed_5a (7x7 -> 4x4): nine taps per block
    Wgrad3Args b;
    b.A = x; b.lda = ldx; b.a_off = xoff; b.G = dc; b.ldg = ldc; b.g_off = coff; b.dW = dw;
    b.M = n * 16; b.I = cin; b.J = cout; b.h = ih; b.w = iw;
    b.itiles = cin / 32; b.jtiles = c2d_ceil_div(cout, 128); b.tiles = b.itiles * b.jtiles;
    C2D_CHECK_ARG((long long)n * 49 * ldx * 4 < (long long)OOB_OFFSET && (long long)b.M * ldc * 4 < (long long)OOB_OFFSET);
    const int slab = 32;                                   // an image pair
    const int nslabs = c2d_ceil_div(b.M, slab);
    int splits = 512 / b.tiles;                            // 2 blocks per CU in one round
    if (splits < 1) splits = 1;
    if (splits > nslabs / 4) splits = nslabs / 4 > 0 ? nslabs / 4 : 1;
    b.rows_per_split = c2d_ceil_div(nslabs, splits) * slab;
    b.splits = c2d_ceil_div(b.M, b.rows_per_split);
    if (splits_out) *splits_out = b.splits;
    if (splits_only) return C2D_OK;
    b.part_stride = partial ? dw_numel : 0;
    if (partial) b.dW = partial;
    dispatch_note("wgrad3x3_s2_kernel<%d>", ES);
    hipLaunchKernelGGL((wgrad3x3_s2_kernel<ES>), dim3(b.tiles * b.splits), dim3(256), 0,
                       (hipStream_t)stream, b);
    return c2d_launch_status();
  }
  WgradArgs a;
  int rc = fill_geom(&a.g, ih, iw, kh, kw, stride, 0);
  if (rc) return rc;
  a.A = x; a.lda = ldx; a.a_off = xoff; a.G = dc; a.ldg = ldc; a.g_off = coff; a.dW = dw;
  a.M = n * a.g.oh * a.g.ow; a.I = cin; a.J = cout;
  const bool narrow = cout % 128 != 0 && cout % 128 <= 64;   // 128x64 block tiles
  const int bj = narrow ? 64 : 128;
  const int tiles = kh * kw * c2d_ceil_div(cin, 128) * c2d_ceil_div(cout, bj);
  static const int fp32_slots = (c2d_tune_on() && c2d_tune_get("wgrad_slots")) ? atoi(c2d_tune_get("wgrad_slots")) : 1024;
  int splits = (bf16_mfma ? wgrad_bf16_slots(512) : c2d_cu_scaled(fp32_slots)) / tiles;   // 4 (bf16: 2) blocks per CU, one round
  const int max_splits = c2d_ceil_div(a.M, 4 * WBK);      // at least 4 slabs per block
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  a.rows_per_split = c2d_ceil_div(c2d_ceil_div(a.M, splits), WBK) * WBK;
  splits = c2d_ceil_div(a.M, a.rows_per_split);
  a.tiles_x = kh * kw * c2d_ceil_div(cin, 128); a.tiles_y = c2d_ceil_div(cout, bj); a.nsplits = splits;
  dim3 grid(a.tiles_x * a.tiles_y * a.nsplits);
  a.a_rows = (long long)n * ih * iw;
  C2D_CHECK_ARG(a.a_rows * ldx * 4 < (long long)OOB_OFFSET && (long long)a.M * ldc * 4 < (long long)OOB_OFFSET);
  const bool plain = kh == 1 && kw == 1 && stride == 1;
  hipStream_t st = (hipStream_t)stream;
  a.part_stride = 0;
  if (bf16_mfma && plain) {
    // 1x1 / stride 1: the DMA-ring kernel of igemm_bf16.hip (round 3)
    WgradArgs b = a;
    if (partial) { b.part_stride = dw_numel; b.dW = partial; }
    const int rc2 = launch_wgrad1x1_bf16_ring(b, st, splits_out, splits_only);
    if (rc2 != C2D_ERR_UNSUPPORTED) return rc2;
  }
  if (bf16_mfma) {
    // K-groups per block (see wgrad_tn_bf16_kernel): 256 blocks of 2 groups = the waves of 512
    // one-group blocks at half their atomic traffic.  Measured (tools/bench_wgrad_bf16.py, the
    // seven per-tap shapes of the second stage): alone 384 us with one group, 359 with two, 321
    // with four; INSIDE the training step, where these launches share the CUs with the input-
    // gradient GEMMs of the main stream, 4.127 / 4.113 / 4.166 ms per step — the 1024-thread,
    // 80-KiB blocks of four groups no longer fit beside a GEMM block.  C2D_TUNE=wgrad_kg=1|2|4.
    static const int kg_env = (c2d_tune_on() && c2d_tune_get("wgrad_kg")) ? atoi(c2d_tune_get("wgrad_kg")) : 0;
    const int kg = kg_env == 1 || kg_env == 2 || kg_env == 4 ? kg_env : 2;
    if (kg > 1) {
      splits = wgrad_bf16_slots(256) / tiles;
      if (splits > max_splits) splits = max_splits;
      if (splits < 1) splits = 1;
    }
    const int unit = kg * WB_KB;
    a.rows_per_split = c2d_ceil_div(c2d_ceil_div(a.M, splits), unit) * unit;
    a.nsplits = c2d_ceil_div(a.M, a.rows_per_split);
    if (splits_out) *splits_out = a.nsplits;
    if (splits_only) return C2D_OK;
    if (partial) { a.part_stride = dw_numel; a.dW = partial; }
    grid.x = a.tiles_x * a.tiles_y * a.nsplits;
    dispatch_note(plain ? "wgrad_tn_bf16_kernel<%d, true, %d>" : "wgrad_tn_bf16_kernel<%d, false, %d>",
                  narrow ? 1 : 2, kg);
#define K_WB_LAUNCH(KG_)                                                                      \
  {                                                                                            \
    const dim3 blockdim(256 * KG_);                                                            \
    if (narrow && plain) hipLaunchKernelGGL((wgrad_tn_bf16_kernel<1, true, KG_>), grid, blockdim, 0, st, a);   \
    else if (narrow) hipLaunchKernelGGL((wgrad_tn_bf16_kernel<1, false, KG_>), grid, blockdim, 0, st, a);      \
    else if (plain) hipLaunchKernelGGL((wgrad_tn_bf16_kernel<2, true, KG_>), grid, blockdim, 0, st, a);        \
    else hipLaunchKernelGGL((wgrad_tn_bf16_kernel<2, false, KG_>), grid, blockdim, 0, st, a);                  \
  }
    if (kg == 4) K_WB_LAUNCH(4) else if (kg == 2) K_WB_LAUNCH(2) else K_WB_LAUNCH(1)
#undef K_WB_LAUNCH
    return c2d_launch_status();
  }
  if (splits_out) *splits_out = a.nsplits;
  if (splits_only) return C2D_OK;
  if (partial) { a.part_stride = dw_numel; a.dW = partial; }
  if (ES == 4 && plain && !partial && x9_active()) {
    // fp32 networks with f32x9 on: the 1x1 filter gradients as nine bf16 partial products too
    // (C2D_TUNE=x9_wgrad=0 keeps them on the fp32 pipe)
    static const bool off = c2d_tune_on() && c2d_tune_get("x9_wgrad") && atoi(c2d_tune_get("x9_wgrad")) == 0;
    if (!off) {
      const int rc2 = launch_wgrad1x1_x9(a, st);
      if (rc2 != C2D_ERR_UNSUPPORTED) return rc2;
    }
  }
  dispatch_note(plain ? "wgrad_tn_kernel<%d, true, %d>" : "wgrad_tn_kernel<%d, false, %d>", narrow ? 1 : 2, ES);
  if (narrow && plain) hipLaunchKernelGGL((wgrad_tn_kernel<1, true, ES>), grid, dim3(256), 0, st, a);
  else if (narrow) hipLaunchKernelGGL((wgrad_tn_kernel<1, false, ES>), grid, dim3(256), 0, st, a);
  else if (plain) hipLaunchKernelGGL((wgrad_tn_kernel<2, true, ES>), grid, dim3(256), 0, st, a);
  else hipLaunchKernelGGL((wgrad_tn_kernel<2, false, ES>), grid, dim3(256), 0, st, a);
  return c2d_launch_status();
}

extern "C" int c2d_conv_wgrad(const float* x, int ldx, int xoff, const float* dc, int ldc,
                              int coff, float* dw, int n, int ih, int iw, int cin, int cout,
                              int kh, int kw, int stride, void* stream) {
  return conv_wgrad_impl<4>(x, ldx, xoff, dc, ldc, coff, dw, n, ih, iw, cin, cout, kh, kw, stride,
                            stream);
}

// bf16 activations / gradients in, fp32 filter gradient out (the operands are widened to fp32 as
// they are staged: same fp32 MFMA accumulation as the fp32 call).
extern "C" int c2d_conv_wgrad_bf16(const void* x, int ldx, int xoff, const void* dc, int ldc,
                                   int coff, float* dw, int n, int ih, int iw, int cin, int cout,
                                   int kh, int kw, int stride, void* stream) {
  return conv_wgrad_impl<2>((const float*)x, ldx, xoff, (const float*)dc, ldc, coff, dw, n, ih, iw,
                            cin, cout, kh, kw, stride, stream);
}

// ---- several 1x1 filter gradients of one input in one launch -----------------------------------
template <int ES>
static int conv1x1_wgrad_multi_impl(const float* x, int ldx, int xoff, int nseg,
                                    const float* const* dcs, const int* ldcs, const int* coffs,
                                    float* const* dws, const int* couts, int rows, int cin,
                                    void* stream) {
  dispatch_reset();
  C2D_CHECK_ARG(x && dcs && ldcs && coffs && dws && couts && nseg >= 1 && nseg <= WGRAD_GROUP_MAX);
  C2D_CHECK_ARG(rows > 0 && cin > 0 && cin % 4 == 0 && ldx % 4 == 0 && xoff % 4 == 0);
  C2D_CHECK_ARG((long long)rows * ldx * 4 < (long long)OOB_OFFSET);
  WgradArgs a[WGRAD_GROUP_MAX];
  for (int p = 0; p < nseg; ++p) {
    C2D_CHECK_ARG(dcs[p] && dws[p] && couts[p] > 0 && couts[p] % 4 == 0 && ldcs[p] % 4 == 0 &&
                  coffs[p] % 4 == 0 && (long long)rows * ldcs[p] * 4 < (long long)OOB_OFFSET);
    const int rc = fill_geom(&a[p].g, 1, 1, 1, 1, 1, 0);
    if (rc) return rc;
    a[p].A = x; a[p].lda = ldx; a[p].a_off = xoff; a[p].G = dcs[p]; a[p].ldg = ldcs[p];
    a[p].g_off = coffs[p]; a[p].dW = dws[p]; a[p].M = rows; a[p].I = cin; a[p].J = couts[p];
    a[p].a_rows = rows; a[p].part_stride = 0;
  }
  hipStream_t st = (hipStream_t)stream;
  if (ES == 2) {
    C2D_CHECK_ARG(cin % 8 == 0 && ldx % 8 == 0 && xoff % 8 == 0);
    for (int p = 0; p < nseg; ++p)
      C2D_CHECK_ARG(couts[p] % 8 == 0 && ldcs[p] % 8 == 0 && coffs[p] % 8 == 0);
    return launch_wgrad1x1_bf16_ring_group(a, nseg, st);
  }
  // fp32: 128 x 64 block tiles for every output (the 32-column MFMA tiles beyond an output's
  // width issue nothing), four workgroups per CU in one round
  WgradGroupArgs g;
  int tiles = 0;
  for (int p = 0; p < nseg; ++p) {
    a[p].tiles_x = c2d_ceil_div(cin, 128);
    a[p].tiles_y = c2d_ceil_div(couts[p], 64);
    g.first_tile[p] = tiles;
    tiles += a[p].tiles_x * a[p].tiles_y;
  }
  for (int p = nseg; p <= WGRAD_GROUP_MAX; ++p) g.first_tile[p] = tiles;
  int splits = 1024 / tiles;
  const int max_splits = c2d_ceil_div(rows, 4 * WBK);
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  const int rps = c2d_ceil_div(c2d_ceil_div(rows, splits), WBK) * WBK;
  g.nsplits = c2d_ceil_div(rows, rps);
  g.num = nseg;
  for (int p = 0; p < nseg; ++p) {
    a[p].rows_per_split = rps; a[p].nsplits = g.nsplits;
    g.a[p] = a[p];
  }
  dispatch_note("wgrad_tn_group_kernel<1, %d>", ES);
  hipLaunchKernelGGL((wgrad_tn_group_kernel<1, 4>), dim3(tiles * g.nsplits), dim3(256), 0, st, g);
  return c2d_launch_status();
}

extern "C" int c2d_conv1x1_wgrad_multi(const float* x, int ldx, int xoff, int nseg,
                                       const float* const* dcs, const int* ldcs, const int* coffs,
                                       float* const* dws, const int* couts, int rows, int cin,
                                       void* stream) {
  return conv1x1_wgrad_multi_impl<4>(x, ldx, xoff, nseg, dcs, ldcs, coffs, dws, couts, rows, cin,
                                     stream);
}

extern "C" int c2d_conv1x1_wgrad_multi_bf16(const void* x, int ldx, int xoff, int nseg,
                                            const void* const* dcs, const int* ldcs,
                                            const int* coffs, float* const* dws, const int* couts,
                                            int rows, int cin, void* stream) {
  return conv1x1_wgrad_multi_impl<2>((const float*)x, ldx, xoff, nseg, (const float* const*)dcs, ldcs,
                                     coffs, dws, couts, rows, cin, stream);
}

// ---- split-K slabs instead of atomics (bf16 MFMA filter gradients) -----------------------------
// Grouped nine-tap filter gradients (wgrad3x3_bf16_group_kernel): `num` 3x3 / stride-1 / SAME
// convolutions over the same n maps of hw x hw (4 or 7), bf16 operands, fp32 atomics into dws[p].
// C2D_ERR_UNSUPPORTED when a problem is not one the nine-tap kernel takes (the caller launches
// them one by one).
extern "C" int c2d_conv3x3_wgrad_multi_bf16(int num, const void* const* xs, const int* ldxs,
                                            const int* xoffs, const void* const* dcs,
                                            const int* ldcs, const int* coffs, float* const* dws,
                                            const int* cins, const int* couts, int n, int hw,
                                            void* stream) {
  dispatch_reset();
  C2D_CHECK_ARG(num >= 1 && num <= WGRAD3_GROUP_MAX && xs && ldxs && xoffs && dcs && ldcs && coffs &&
                dws && cins && couts && n > 0);
  if (!(hw == 4 || hw == 7) || n < 256 || !wgrad_bf16_mfma_enabled()) return C2D_ERR_UNSUPPORTED;
  Wgrad3GroupArgs g = {};
  g.num = num;
  int total = 0;
  const int M = n * hw * hw;
  for (int p = 0; p < num; ++p) {
    C2D_CHECK_ARG(xs[p] && dcs[p] && dws[p] && cins[p] > 0 && couts[p] > 0);
    if (cins[p] % 32 || couts[p] % 32 || ldxs[p] % 8 || xoffs[p] % 8 || ldcs[p] % 8 || coffs[p] % 8 ||
        wgrad3_bf16_igroups(cins[p]) != 1)
      return C2D_ERR_UNSUPPORTED;
    C2D_CHECK_ARG((long long)M * ldxs[p] * 2 < (long long)OOB_OFFSET &&
                  (long long)M * ldcs[p] * 2 < (long long)OOB_OFFSET);
    Wgrad3Args& b = g.a[p];
    b.A = (const float*)xs[p]; b.lda = ldxs[p]; b.a_off = xoffs[p];
    b.G = (const float*)dcs[p]; b.ldg = ldcs[p]; b.g_off = coffs[p]; b.dW = dws[p];
    b.M = M; b.I = cins[p]; b.J = couts[p]; b.h = hw; b.w = hw;
    b.itiles = c2d_ceil_div(cins[p], 32); b.jtiles = c2d_ceil_div(couts[p], 128);
    b.tiles = b.itiles * b.jtiles;
    b.part_stride = 0;
    g.first[p] = total;
    total += b.tiles;
  }
  g.first[num] = total;
  const int slab = hw == 4 ? 8 * 16 : 2 * 49;            // whole images per slab
  const int nslabs = c2d_ceil_div(M, slab);
  int splits = wgrad_bf16_slots(hw == 4 ? 256 : 384) / total;   // one round of resident workgroups
  if (splits < 1) splits = 1;
  if (splits > nslabs / 4) splits = nslabs / 4 > 0 ? nslabs / 4 : 1;
  const int rows_per_split = c2d_ceil_div(nslabs, splits) * slab;
  g.splits = c2d_ceil_div(M, rows_per_split);
  for (int p = 0; p < num; ++p) { g.a[p].rows_per_split = rows_per_split; g.a[p].splits = g.splits; }
  const dim3 grid(total * g.splits);
  dispatch_note("wgrad3x3_bf16_group_kernel<%d, %d>", hw, hw == 4 ? 8 : 2);
  if (hw == 4) hipLaunchKernelGGL((wgrad3x3_bf16_group_kernel<4, 8>), grid, dim3(256), 0, (hipStream_t)stream, g);
  else hipLaunchKernelGGL((wgrad3x3_bf16_group_kernel<7, 2>), grid, dim3(256), 0, (hipStream_t)stream, g);
  return c2d_launch_status();
}

extern "C" int c2d_conv_wgrad_bf16_splits(int ldx, int xoff, int ldc, int coff, int n, int ih,
                                          int iw, int cin, int cout, int kh, int kw, int stride) {
  int splits = 0;
  const int rc = conv_wgrad_impl<2>(nullptr, ldx, xoff, nullptr, ldc, coff, nullptr, n, ih, iw, cin,
                                    cout, kh, kw, stride, nullptr, nullptr, &splits, true);
  return rc == C2D_OK ? splits : rc;
}

extern "C" int c2d_conv_wgrad_bf16_partial(const void* x, int ldx, int xoff, const void* dc,
                                           int ldc, int coff, float* partials,
                                           long long partial_floats, int n, int ih, int iw,
                                           int cin, int cout, int kh, int kw, int stride,
                                           void* stream) {
  C2D_CHECK_ARG(partials);
  const int splits = c2d_conv_wgrad_bf16_splits(ldx, xoff, ldc, coff, n, ih, iw, cin, cout, kh, kw,
                                                stride);
  if (splits < 0) return splits;
  if ((long long)splits * kh * kw * cin * cout > partial_floats) return C2D_ERR_WORKSPACE;
  return conv_wgrad_impl<2>((const float*)x, ldx, xoff, (const float*)dc, ldc, coff, nullptr, n, ih,
                            iw, cin, cout, kh, kw, stride, stream, partials);
}

extern "C" int c2d_conv_wgrad_splits(int ldx, int xoff, int ldc, int coff, int n, int ih, int iw,
                                     int cin, int cout, int kh, int kw, int stride) {
  int splits = 0;
  const int rc = conv_wgrad_impl<4>(nullptr, ldx, xoff, nullptr, ldc, coff, nullptr, n, ih, iw, cin,
                                    cout, kh, kw, stride, nullptr, nullptr, &splits, true);
  return rc == C2D_OK ? splits : rc;
}

extern "C" int c2d_conv_wgrad_partial(const float* x, int ldx, int xoff, const float* dc, int ldc,
                                      int coff, float* partials, long long partial_floats, int n,
                                      int ih, int iw, int cin, int cout, int kh, int kw, int stride,
                                      void* stream) {
  C2D_CHECK_ARG(partials);
  const int splits = c2d_conv_wgrad_splits(ldx, xoff, ldc, coff, n, ih, iw, cin, cout, kh, kw, stride);
  if (splits < 0) return splits;
  if ((long long)splits * kh * kw * cin * cout > partial_floats) return C2D_ERR_WORKSPACE;
  return conv_wgrad_impl<4>(x, ldx, xoff, dc, ldc, coff, nullptr, n, ih, iw, cin, cout, kh, kw,
                            stride, stream, partials);
}

static_assert(sizeof(WgradReduceDesc) == sizeof(C2dWgradReduceDesc), "C2dWgradReduceDesc layout");
extern "C" int c2d_wgrad_reduce_batched(const C2dWgradReduceDesc* desc, int num, int total_chunks,
                                        const float* workspace, float* grads, void* stream) {
  C2D_CHECK_ARG(desc && workspace && grads && num >= 0 && total_chunks >= 0);
  if (num == 0 || total_chunks == 0) return C2D_OK;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(total_chunks), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const WgradReduceDesc*>(desc), num, workspace, grads);
  return c2d_launch_status();
}
