// fp32 convolutions on the bf16 matrix pipe: every fp32 operand value is the EXACT sum of three
// bf16 terms (truncation split, 8 + 8 + 8 significant bits), every product of two terms is exact in
// fp32, so nine v_mfma_f32_32x32x16_bf16 per 16 k accumulate the fp32 product in fp32 at 288
// matrix-pipe cycles instead of the 512 of eight v_mfma_f32_32x32x2_f32 ("f32x9"; slim.conv2d of
// models/utils.py:165-167 and its input gradient at the reference's own precision).
//
// Round 5 measured the idea on the register-staged fp32 skeleton, whose loads the shorter MFMA
// phase no longer covered (profiles/r05_experiments/README.md).  Here it runs on the DMA ring of
// igemm_ring.h:
//   * WEIGHTS are split once per optimiser step into three bf16 planes in HBM (c2d_split3_bf16,
//     this file) and reach LDS by buffer_load ... lds like any bf16 operand;
//   * ACTIVATIONS (and activation gradients) stay fp32 in HBM and in LDS — no producer kernel
//     changes, 4 instead of 6 bytes per element — and are split by the lane that reads a fragment:
//     44 single-issue vector instructions per 8-element fragment, issued in the gaps of the
//     9 x NT MFMAs the fragment feeds (igemm_ring.h: split3_frag).
// The fp32 entry points of conv_gemm.hip take this path for a weight operand that lies inside an
// arena bound with c2d_f32x9_bind (the one piece of process state behind the C-ABI: a table of at
// most 64 (fp32 arena, plane arena) pairs, written before the step loop, read-only inside it).
#include "igemm_ring.h"
#include <mutex>

namespace c2d_ig {
namespace {

// ---- the split --------------------------------------------------------------------------------
// One thread per four elements; planes[p][i] = plane p of src[i], plane p at planes + p * stride
// elements.  Non-finite values: hi carries them, mid / lo are zero (Inf x w stays Inf x w).
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ src,
                                                     unsigned short* __restrict__ planes,
                                                     long long stride, long long n4) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const f32x4 v = *reinterpret_cast<const f32x4*>(src + i * 4);
  unsigned short h[4], m[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const unsigned u = __float_as_uint(v[e]);
    const float r1 = v[e] - __uint_as_float(u & 0xffff0000u);
    const unsigned u1 = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(u1 & 0xffff0000u);
    const bool finite = (u & 0x7f800000u) != 0x7f800000u;
    h[e] = (unsigned short)(u >> 16);
    m[e] = finite ? (unsigned short)(u1 >> 16) : (unsigned short)0;
    l[e] = finite ? (unsigned short)(__float_as_uint(r2) >> 16) : (unsigned short)0;
  }
  typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
  *reinterpret_cast<u16x4*>(planes + i * 4) = u16x4{h[0], h[1], h[2], h[3]};
  *reinterpret_cast<u16x4*>(planes + stride + i * 4) = u16x4{m[0], m[1], m[2], m[3]};
  *reinterpret_cast<u16x4*>(planes + 2 * stride + i * 4) = u16x4{l[0], l[1], l[2], l[3]};
}

// ---- the bindings -----------------------------------------------------------------------------
struct X9Arena {
  const char* base;          // fp32 arena
  long long bytes;
  const char* planes;        // plane 0 of element 0
  long long stride_bytes;    // plane p + 1 sits this many bytes behind plane p
};
constexpr int X9_MAX_ARENAS = 64;
X9Arena g_arena[X9_MAX_ARENAS];
int g_narena = 0;
int g_enabled = 1;
std::mutex g_arena_mu;

// ---- launchers ----------------------------------------------------------------------------------
template <int MODE, int WM, int WN, int MT, int NT, bool PM, int BKT, int D, bool FUSED>
int x9_launch(const IgemmArgs& a, hipStream_t s) {
  dispatch_note_ext(PM ? (FUSED ? "igemm_ring_kernel<%d, %d, %d, %d, %d, true, %d, %d, 4, true, 3>"
                                : "igemm_ring_kernel<%d, %d, %d, %d, %d, true, %d, %d, 4, false, 3>")
                       : (FUSED ? "igemm_ring_kernel<%d, %d, %d, %d, %d, false, %d, %d, 4, true, 3>"
                                : "igemm_ring_kernel<%d, %d, %d, %d, %d, false, %d, %d, 4, false, 3>"),
                    MODE, WM, WN, MT, NT, BKT, D);
  hipLaunchKernelGGL((igemm_ring_kernel<MODE, WM, WN, MT, NT, PM, BKT, D, 4, FUSED, 3>),
                     dim3(a.m_tiles * a.n_tiles), dim3(WM * WN * 64), 0, s, a);
  return c2d_launch_status();
}

template <int MODE, int WM, int WN, int MT, int NT, bool PM, int BKT, int D>
int x9_one(IgemmArgs a, hipStream_t s) {
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
  a.trace = ring_trace_buffer();
#endif
  if constexpr (MODE == 1) {
    if (a.fy != nullptr) return x9_launch<MODE, WM, WN, MT, NT, PM, BKT, D, true>(a, s);
  } else if (a.fy != nullptr) {
    return C2D_ERR_UNSUPPORTED;
  }
  return x9_launch<MODE, WM, WN, MT, NT, PM, BKT, D, false>(a, s);
}

// Ring geometry per tile (tools/sweep_x9.sh, tools/x9_probe.py; a -DC2D_X9_SWEEP build holds the other
// combinations behind C2D_TUNE=x9_bk / x9_d): 32-deep stages (128-byte activation rows,
// 64-byte weight-plane rows) in two buffers up to 128 columns — 80 KiB for a 128 x 128 tile, two
// workgroups of four waves per CU; 16-deep stages in two buffers beyond.  Every combination measured
// within 10 % of every other: what bounds these kernels is the clock the chip holds under bf16 MFMAs
// (1.70 - 1.84 GHz against 2.39 GHz under the fp32 MFMA kernel, tools/x9_clock.py), not the ring.
struct X9Tune { int bk, d; };
const X9Tune& x9_tune() {
  static const X9Tune t = [] {
    X9Tune r = {0, 0};
    if (c2d_tune_on()) {
      if (const char* e = c2d_tune_get("x9_bk")) r.bk = atoi(e);
      if (const char* e = c2d_tune_get("x9_d")) r.d = atoi(e);
    }
    return r;
  }();
  return t;
}

template <int MODE, int WM, int WN, int MT, int NT, bool PM>
int x9_tile(const IgemmArgs& a, hipStream_t s) {
  constexpr int ROWB = (WM * MT * 4 + WN * NT * 6) * 32;      // bytes per unit of stage depth
  constexpr bool DEEP = WN * NT <= 4 && 2 * ROWB * 32 <= 160 * 1024;
#ifdef C2D_X9_SWEEP
  const X9Tune& t = x9_tune();
  const int bk = t.bk ? t.bk : (DEEP ? 32 : 16), d = t.d ? t.d : 2;
  if (bk == 16 && d == 2) return x9_one<MODE, WM, WN, MT, NT, PM, 16, 2>(a, s);
  if constexpr (3 * ROWB * 16 <= 160 * 1024)
    if (bk == 16 && d == 3) return x9_one<MODE, WM, WN, MT, NT, PM, 16, 3>(a, s);
  if constexpr (4 * ROWB * 16 <= 160 * 1024)
    if (bk == 16 && d == 4) return x9_one<MODE, WM, WN, MT, NT, PM, 16, 4>(a, s);
  if constexpr (2 * ROWB * 32 <= 160 * 1024)
    if (bk == 32 && d == 2) return x9_one<MODE, WM, WN, MT, NT, PM, 32, 2>(a, s);
#endif
  if constexpr (DEEP) return x9_one<MODE, WM, WN, MT, NT, PM, 32, 2>(a, s);
  else return x9_one<MODE, WM, WN, MT, NT, PM, 16, 2>(a, s);
}

template <int WM, int WN, int MT, int NT>
int x9_shape(const IgemmArgs& a, bool pm, hipStream_t s) {
  if (a.g.mode == 0)
    return pm ? x9_tile<0, WM, WN, MT, NT, true>(a, s) : x9_tile<0, WM, WN, MT, NT, false>(a, s);
  return pm ? x9_tile<1, WM, WN, MT, NT, true>(a, s) : x9_tile<1, WM, WN, MT, NT, false>(a, s);
}

}  // namespace

// Plane pointer of the fp32 weight operand [w, w + bytes): null when no bound arena holds it.
const void* x9_planes_of(const void* w, long long bytes, long long* stride_bytes) {
  if (!g_enabled) return nullptr;
  const char* p = (const char*)w;
  for (int i = 0; i < g_narena; ++i) {
    const X9Arena& r = g_arena[i];
    if (p >= r.base && p + bytes <= r.base + r.bytes) {
      *stride_bytes = r.stride_bytes;
      return r.planes + (p - r.base) / 2;
    }
  }
  return nullptr;
}

int launch_igemm_x9_ring(const IgemmArgs& a, int wm, int wn, int mt, int nt, bool pm, hipStream_t s,
                         int* m_tiles_out, bool query) {
  const int key = ((wm * 10 + wn) * 10 + mt) * 10 + nt;
  switch (key) {
    case 4112: case 4113: case 4114: case 4115: case 4116: case 2222: case 2224: case 2422:
      break;
    default:
      return C2D_ERR_UNSUPPORTED;
  }
  if (m_tiles_out) *m_tiles_out = c2d_ceil_div(a.M, wm * mt * 32);
  if (query) return C2D_OK;
  switch (key) {
    // four waves of 32 rows x every column of the tile: a lane splits its activation fragment once
    // for 9 x NT MFMAs and no two waves split the same rows
    case 4112: return x9_shape<4, 1, 1, 2>(a, pm, s);      // 128 x 64
    case 4113: return x9_shape<4, 1, 1, 3>(a, pm, s);      // 128 x 96
    case 4114: return x9_shape<4, 1, 1, 4>(a, pm, s);      // 128 x 128
    case 4115: return x9_shape<4, 1, 1, 5>(a, pm, s);      // 128 x 160
    case 4116: return x9_shape<4, 1, 1, 6>(a, pm, s);      // 128 x 192
    case 2222: return x9_shape<2, 2, 2, 2>(a, pm, s);      // 128 x 128, waves 2 x 2
    case 2422: return x9_shape<2, 4, 2, 2>(a, pm, s);      // 128 x 256, eight waves 2 x 4
    default: return x9_shape<2, 2, 2, 4>(a, pm, s);        // 128 x 256, waves 2 x 2
  }
}

namespace {

// ---- filter gradient of a 1x1 / stride-1 convolution as nine partial products -------------------
// dW[i][j] = sum over rows m of x[m][i] * dC[m][j]: BOTH operands are fp32 activations, the contraction
// runs over rows.  The loader threads split what they fetch — 8 channels of a row = one split3_frag =
// three 16-byte LDS writes, once per element instead of once per fragment use — into three bf16 plane
// tiles per operand, staged as they lie in HBM ([row][channel]); the MFMA operands (8 consecutive rows
// of one channel per lane) come out of ds_read_b64_tr_b16 as in wgrad_tn_bf16_kernel (conv_gemm.hip).
// Block tile 128 (i) x 64 NTJ (j), four waves 2 x 2, 16-row slabs; 30 KiB of LDS: three blocks per CU.
// The fp32 kernel this replaces (wgrad_tn_kernel, 16-row register-staged slabs on v_mfma_f32_32x32x2)
// runs the step's fifteen 1x1 filter gradients at 0.50 - 0.69 of the fp32 matrix peak.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4_t;
__device__ __forceinline__ bf16x8 tr_frag_x9(const char* p, int hi_off) {
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)p);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_t*)(p + hi_off));
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

constexpr int WX_RS = 256 + 64;     // bytes per staged row of a plane tile (128 bf16 + pad: 64 mod 256)

// WX_KB = rows per slab: 32 (two 16-row MFMA k-steps, 60 KiB of LDS: two workgroups per CU) or 16
// (30 KiB: three workgroups per CU at ~150 registers, twice the barriers per row)
template <int NTJ, int WX_KB>
__global__ __launch_bounds__(256, WX_KB == 32 ? 2 : 3) void wgrad1x1_x9_kernel(WgradArgs a) {
  constexpr int BJ = 2 * NTJ * 32;
  constexpr int WX_TILE = WX_KB * WX_RS;
  constexpr int LU = WX_KB / 16;      // rows per loader thread
  __shared__ __attribute__((aligned(16))) char smem[6 * WX_TILE];   // A planes 0..2, G planes 0..2
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  char* const As = smem;
  char* const Gs = smem + 3 * WX_TILE;
  const WgradBlock blk = wgrad_block(a);
  const int i0 = blk.x * 128;
  const int j0 = blk.y * BJ;
  const int mbeg = blk.z * a.rows_per_split;
  const int nslabs = (min(a.rows_per_split, a.M - mbeg) + WX_KB - 1) / WX_KB;

  // loader: thread -> rows kr, kr + 16 of the slab, 8 channels c8 (two 16-byte loads per row);
  // columns beyond I / J are clamped (their products land in dW rows / columns never stored);
  // rows >= M lie outside the descriptors and come back as zeros
  const int kr = tid >> 4;
  const int c8 = (tid & 15) * 8;
  const bool gload = c8 < BJ;
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc_b((const char*)a.A + (size_t)a.a_off * 4,
                                                 (a.a_rows * a.lda - a.a_off) * 4);
  const __amdgpu_buffer_rsrc_t rsG = make_rsrc_b((const char*)a.G + (size_t)a.g_off * 4,
                                                 ((long long)a.M * a.ldg - a.g_off) * 4);
  const unsigned acol = (unsigned)min(i0 + c8, a.I - 8) * 4u;
  const unsigned gcol = (unsigned)min(j0 + min(c8, BJ - 8), a.J - 8) * 4u;
  unsigned aoffs[LU], goffs[LU];
#pragma unroll
  for (int u = 0; u < LU; ++u) {
    aoffs[u] = (unsigned)((kr + u * 16) * a.lda) * 4u + acol;
    goffs[u] = (unsigned)((kr + u * 16) * a.ldg) * 4u + gcol;
  }

  f32x16 acc[2][NTJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NTJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  unsigned tile_bits = 0;   // bit i * NTJ + j: 32 x 32 tile inside I x J (others issue no MFMA)
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NTJ; ++j)
      if ((i0 + wm * 64 + i * 32 < a.I) && (j0 + (wn * NTJ + j) * 32 < a.J))
        tile_bits |= 1u << (i * NTJ + j);
  tile_bits = __builtin_amdgcn_readfirstlane(tile_bits);

  f32x4 ra[LU][2], rg[LU][2];
#define K_WX_LOAD(MB)                                                      \
  {                                                                          \
    const int sa = (MB) * a.lda * 4, sg = (MB) * a.ldg * 4;                  \
    _Pragma("unroll") for (int u = 0; u < LU; ++u) {                         \
      ra[u][0] = buf_load4(rsA, aoffs[u], sa);                               \
      ra[u][1] = buf_load4(rsA, aoffs[u] + 16u, sa);                         \
      rg[u][0] = buf_load4(rsG, goffs[u], sg);                               \
      rg[u][1] = buf_load4(rsG, goffs[u] + 16u, sg);                         \
    }                                                                        \
  }
  // transposed-read bases: row 8 * lh + q, 16-column half (lane >> 4) & 1, 4-column slot p
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
  const char* const apl = As + (8 * lh + tq) * WX_RS + (wm * 64 + 16 * tg + 4 * tp) * 2;
  const char* const gpl = Gs + (8 * lh + tq) * WX_RS + (wn * NTJ * 32 + 16 * tg + 4 * tp) * 2;

  K_WX_LOAD(mbeg);
  for (int sl = 0, mb = mbeg; sl < nslabs; ++sl, mb += WX_KB) {
#pragma unroll
    for (int u = 0; u < LU; ++u) {
      bf16x8 h, m, l;
      char* const wa = As + (kr + u * 16) * WX_RS + c8 * 2;
      split3_frag(ra[u][0], ra[u][1], h, m, l);
      *reinterpret_cast<bf16x8*>(wa) = h;
      *reinterpret_cast<bf16x8*>(wa + WX_TILE) = m;
      *reinterpret_cast<bf16x8*>(wa + 2 * WX_TILE) = l;
      if (gload) {
        char* const wg = Gs + (kr + u * 16) * WX_RS + c8 * 2;
        split3_frag(rg[u][0], rg[u][1], h, m, l);
        *reinterpret_cast<bf16x8*>(wg) = h;
        *reinterpret_cast<bf16x8*>(wg + WX_TILE) = m;
        *reinterpret_cast<bf16x8*>(wg + 2 * WX_TILE) = l;
      }
    }
    __syncthreads();
    K_WX_LOAD(mb + WX_KB);     // next slab (rows past M: zeros)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < LU; ++s) {
      bf16x8 af[2][3], bf[NTJ][3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
          af[i][p] = tr_frag_x9(apl + p * WX_TILE + s * 16 * WX_RS + i * 64, 4 * WX_RS);
#pragma unroll
        for (int j = 0; j < NTJ; ++j)
          bf[j][p] = tr_frag_x9(gpl + p * WX_TILE + s * 16 * WX_RS + j * 64, 4 * WX_RS);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NTJ; ++j)
          if ((tile_bits >> (i * NTJ + j)) & 1u) {
            // smallest products first (planes 0 hi, 1 mid, 2 lo), as in the ring kernel
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][2], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][2], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][2], bf[j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][2], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][1], bf[j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][0], bf[j][0], acc[i][j], 0, 0, 0);
          }
    }
    __syncthreads();
  }
#undef K_WX_LOAD

  // split-K result: fp32 atomics into dW (pre-zeroed or accumulated into), as wgrad_tn_kernel
  float* dw = a.dW;
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
        atomicAdd(dw + (size_t)ii * a.J + jj, acc[i][j][r]);
      }
  }
}


// ---- 3x3 / stride-1 / SAME filter gradient over 4x4 / 7x7 maps as nine partial products ----------
// One TAP per block, rows in pixel order: a 16-row k-step is ONE output pixel of 16 consecutive
// images (dC rows img * HW + p, x rows img * HW + p + the tap's pixel offset: row stride HW in both),
// so the (pixel, tap) pairs that fall into the SAME padding are never staged and never multiplied —
// what the fp32 nine-tap kernel achieves by pairing the same pixel of two images in a k = 2 step, and
// what a 16-consecutive-row k-step (one 4x4 image) cannot.  The price: x and dC are fetched once per
// tap that uses them (6.25 of 9 on 4x4 maps) instead of once — from the L2 of the XCD that runs all
// tiles of a (tap, image range).  Blocks of a tap with fewer valid pixels get more images (equal
// k-steps per block: one round of resident blocks).  Loader / LDS planes / MFMA order as above.
struct Wgrad3X9Args {
  WgradArgs w;        // A, lda, a_off, a_rows, G, ldg, g_off, dW, M, I, J, tiles_x (i), tiles_y (j)
  int hw, wc, nimg;   // output (dC) pixels per image, output map width, images
  int hwx, wcx, stride;   // input (x) pixels per image, input map width; 1, or 2 (7x7 -> 4x4)
  int swapped;        // the 128-wide "i" operand is dC (I = cout) and the "j" operand is x (J = cin):
                      // fewer padded tile columns for cin = 160 / 192 (host); dW stays [tap][cin][cout]
  int first[10];      // first logical block of tap t; first[9] = grid size
  int ipb[9];         // images per block of tap t (a multiple of 16)
};

template <int NTJ, bool SWAP>
__global__ __launch_bounds__(256, 3) void wgrad3x3_x9_kernel(Wgrad3X9Args q) {
  const WgradArgs& a = q.w;
  constexpr int BJ = 2 * NTJ * 32;
  constexpr int WX_TILE = 16 * WX_RS;
  __shared__ __attribute__((aligned(16))) char smem[6 * WX_TILE];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  char* const As = smem;
  char* const Gs = smem + 3 * WX_TILE;
  const int logical = xcd_remap(blockIdx.x, q.first[9]);
  int tap = 0;
#pragma unroll
  for (int t = 1; t < 9; ++t)
    if (logical >= q.first[t]) tap = t;
  const int tiles = a.tiles_x * a.tiles_y;
  const int rblk = logical - q.first[tap];
  const int split = rblk / tiles;
  const int tile = rblk - split * tiles;
  const int by = tile / a.tiles_x, bx = tile - by * a.tiles_x;
  const int i0 = bx * 128, j0 = by * BJ;
  const int ky = tap / 3, kx = tap - 3 * ky;
  const int wc = q.wc, hw = q.hw, hwx = q.hwx, wcx = q.wcx, st = q.stride;
  // output pixels (y, x) whose source pixel (st y + ky - 1, st x + kx - 1) lies inside the input map
  const int y0 = ky == 0 ? 1 : 0, y1 = min(wc, (wcx - ky) / st + 1);
  const int x0 = kx == 0 ? 1 : 0, x1 = min(wc, (wcx - kx) / st + 1);
  const int doff = (ky - 1) * wcx + (kx - 1);
  const int img0 = split * q.ipb[tap];
  const int nsl = (min(q.ipb[tap], q.nimg - img0) + 15) / 16;
  const int total = (y1 - y0) * (x1 - x0) * nsl;

  // loader: thread -> image kr of the slab, 8 channels c8 (two 16-byte loads); images >= nimg lie
  // outside the descriptors (zeros); columns beyond I / J are clamped (never stored)
  const int kr = tid >> 4;
  const int c8 = (tid & 15) * 8;
  const bool gload = c8 < BJ;
  const __amdgpu_buffer_rsrc_t rsA = make_rsrc_b((const char*)a.A + (size_t)a.a_off * 4,
                                                 (a.a_rows * a.lda - a.a_off) * 4);
  const __amdgpu_buffer_rsrc_t rsG = make_rsrc_b((const char*)a.G + (size_t)a.g_off * 4,
                                                 ((long long)a.M * a.ldg - a.g_off) * 4);
  const unsigned aoff = (unsigned)(kr * (SWAP ? hw : hwx) * a.lda + min(i0 + c8, a.I - 8)) * 4u;
  const unsigned goff = (unsigned)(kr * (SWAP ? hwx : hw) * a.ldg + min(j0 + min(c8, BJ - 8), a.J - 8)) * 4u;

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
  int py = y0, px = x0, sl = 0;         // (scalar) the k-step being LOADED
#define K_W3_LOAD()                                                          \
  {                                                                            \
    const int rowc = (img0 + sl * 16) * hw + py * wc + px;                     \
    const int rowx = (img0 + sl * 16) * hwx + st * (py * wcx + px) + doff;     \
    const int sg = (SWAP ? rowx : rowc) * a.ldg * 4;                           \
    const int sa = (SWAP ? rowc : rowx) * a.lda * 4;                           \
    ra[0] = buf_load4(rsA, aoff, sa);                                          \
    ra[1] = buf_load4(rsA, aoff + 16u, sa);                                    \
    rg[0] = buf_load4(rsG, goff, sg);                                          \
    rg[1] = buf_load4(rsG, goff + 16u, sg);                                    \
    if (++sl == nsl) { sl = 0; if (++px == x1) { px = x0; ++py; } }            \
  }
  const int tq = (lane & 15) >> 2, tp = lane & 3, tg = (lane >> 4) & 1;
  const char* const apl = As + (8 * lh + tq) * WX_RS + (wm * 64 + 16 * tg + 4 * tp) * 2;
  const char* const gpl = Gs + (8 * lh + tq) * WX_RS + (wn * NTJ * 32 + 16 * tg + 4 * tp) * 2;

  K_W3_LOAD();
  for (int step = 0; step < total; ++step) {
    {
      bf16x8 h, m, l;
      char* const wa = As + kr * WX_RS + c8 * 2;
      split3_frag(ra[0], ra[1], h, m, l);
      *reinterpret_cast<bf16x8*>(wa) = h;
      *reinterpret_cast<bf16x8*>(wa + WX_TILE) = m;
      *reinterpret_cast<bf16x8*>(wa + 2 * WX_TILE) = l;
      if (gload) {
        char* const wg = Gs + kr * WX_RS + c8 * 2;
        split3_frag(rg[0], rg[1], h, m, l);
        *reinterpret_cast<bf16x8*>(wg) = h;
        *reinterpret_cast<bf16x8*>(wg + WX_TILE) = m;
        *reinterpret_cast<bf16x8*>(wg + 2 * WX_TILE) = l;
      }
    }
    __syncthreads();
    if (step + 1 < total) K_W3_LOAD();
    __builtin_amdgcn_sched_barrier(0);
    {
      bf16x8 af[2][3], bf[NTJ][3];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i][p] = tr_frag_x9(apl + p * WX_TILE + i * 64, 4 * WX_RS);
#pragma unroll
        for (int j = 0; j < NTJ; ++j) bf[j][p] = tr_frag_x9(gpl + p * WX_TILE + j * 64, 4 * WX_RS);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NTJ; ++j)
          if ((tile_bits >> (i * NTJ + j)) & 1u) {
            // (SWAP: the operands change places, so that the lanes of the result run along cout —
            //  the contiguous dimension of dW — in both forms)
#define K_W3_MFMA(PA, PB)                                                                              \
  acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[j][PB], af[i][PA], acc[i][j], 0, 0, 0)  \
                   : __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][PA], bf[j][PB], acc[i][j], 0, 0, 0);
            K_W3_MFMA(2, 2) K_W3_MFMA(2, 1) K_W3_MFMA(1, 2) K_W3_MFMA(1, 1) K_W3_MFMA(2, 0)
            K_W3_MFMA(0, 2) K_W3_MFMA(1, 0) K_W3_MFMA(0, 1) K_W3_MFMA(0, 0)
#undef K_W3_MFMA
          }
    }
    __syncthreads();
  }
#undef K_W3_LOAD

  float* dw = a.dW + (size_t)tap * a.I * a.J;
#pragma unroll
  for (int j = 0; j < NTJ; ++j)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (!((tile_bits >> (i * NTJ + j)) & 1u)) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2) + 4 * lh;
        if constexpr (SWAP) {      // rows of the tile = cin (kernel j), lanes = cout (kernel i)
          const int jj = j0 + (wn * NTJ + j) * 32 + rr, ii = i0 + wm * 64 + i * 32 + li;
          if (jj < a.J && ii < a.I) atomicAdd(dw + (size_t)jj * a.I + ii, acc[i][j][r]);
        } else {
          const int ii = i0 + wm * 64 + i * 32 + rr, jj = j0 + (wn * NTJ + j) * 32 + li;
          if (ii < a.I && jj < a.J) atomicAdd(dw + (size_t)ii * a.J + jj, acc[i][j][r]);
        }
      }
    }
}

}  // namespace

bool x9_active() { return g_enabled && g_narena > 0; }

int launch_wgrad1x1_x9(WgradArgs a, hipStream_t s) {
  if (a.I % 8 || a.J % 8 || a.lda % 4 || a.ldg % 4 || a.a_off % 4 || a.g_off % 4 || a.I < 8 || a.J < 8 ||
      a.part_stride > 0)
    return C2D_ERR_UNSUPPORTED;
  // (the single-image first stage — 1024 rows, a handful of blocks of a few slabs — is quicker on
  //  the fp32 kernel's 16-row slabs: 12.6 - 14.0 against 14.5 - 15.3 us)
  if (a.M < 8192) return C2D_ERR_UNSUPPORTED;
  static const bool tune = c2d_tune_on();
  static const int slots_env = (tune && c2d_tune_get("x9_wgrad_slots")) ? atoi(c2d_tune_get("x9_wgrad_slots")) : 512;
  // Output widths whose last 128-column tile would be at most half full (192, 160 columns) take
  // 128 x 64 tiles, where the loader's split arithmetic is spread over half the MFMAs: alone those
  // launches measure 0.96 - 0.97 x the fp32 kernel (128 and 352 columns: 1.27 - 1.42 x), inside the
  // step — matrix-pipe-bound across its two streams — they still pay: 9.70 -> 9.63 - 9.68 ms with
  // them, 9.84 - 9.87 with every filter gradient on the fp32 pipe.  C2D_TUNE=x9_wgrad_narrow=0
  // leaves them to the fp32 kernel.
  static const bool skip_narrow = tune && c2d_tune_get("x9_wgrad_narrow") && atoi(c2d_tune_get("x9_wgrad_narrow")) == 0;
  const bool narrow = a.J % 128 != 0 && a.J % 128 <= 64;      // 128 x 64 block tiles
  if (narrow && skip_narrow) return C2D_ERR_UNSUPPORTED;
  const int bj = narrow ? 64 : 128;
  a.tiles_x = c2d_ceil_div(a.I, 128);
  a.tiles_y = c2d_ceil_div(a.J, bj);
  const int tiles = a.tiles_x * a.tiles_y;
  int splits = c2d_cu_scaled(slots_env) / tiles;               // two blocks per CU, one round
  // 16-row slabs (30 KiB of LDS, three workgroups per CU) against 32-row slabs (two): the ten launches
  // of the step 1.31 -> 1.14 ms in one stream, the step 10.04 - 10.12 -> 9.81 - 9.88 ms on one box
  static const int kb = (tune && c2d_tune_get("x9_wgrad_kb") && atoi(c2d_tune_get("x9_wgrad_kb")) == 32) ? 32 : 16;
  const int max_splits = c2d_ceil_div(a.M, 4 * 32);            // at least 128 rows per block
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  a.rows_per_split = c2d_ceil_div(c2d_ceil_div(a.M, splits), 32) * 32;
  a.nsplits = c2d_ceil_div(a.M, a.rows_per_split);
  const dim3 grid(tiles * a.nsplits), block(256);
  dispatch_note_ext("wgrad1x1_x9_kernel<%d, %d>", narrow ? 1 : 2, kb);
  if (kb == 16) {
    if (narrow) hipLaunchKernelGGL((wgrad1x1_x9_kernel<1, 16>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((wgrad1x1_x9_kernel<2, 16>), grid, block, 0, s, a);
  } else {
    if (narrow) hipLaunchKernelGGL((wgrad1x1_x9_kernel<1, 32>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((wgrad1x1_x9_kernel<2, 32>), grid, block, 0, s, a);
  }
  return c2d_launch_status();
}


// a: A (x), lda, a_off, G (dC), ldg, g_off, dW, I (cin), J (cout) filled; n images of wcx x wcx input
// pixels, stride 1 (wcx = 4 or 7) or 2 (7 -> 4)
int launch_wgrad3x3_x9(WgradArgs a, int n, int wcx, int stride, hipStream_t s) {
  if (a.I % 8 || a.J % 8 || a.lda % 4 || a.ldg % 4 || a.a_off % 4 || a.g_off % 4 || a.I < 8 || a.J < 8 ||
      a.part_stride > 0 || n < 256 || !((stride == 1 && (wcx == 4 || wcx == 7)) || (stride == 2 && wcx == 7)))
    return C2D_ERR_UNSUPPORTED;
  const int wc = (wcx + stride - 1) / stride;
  static const bool tune = c2d_tune_on();
  static const int slots_env = (tune && c2d_tune_get("x9_wgrad3_slots")) ? atoi(c2d_tune_get("x9_wgrad3_slots")) : 768;
  static const int swap_env = (tune && c2d_tune_get("x9_wgrad3_swap")) ? atoi(c2d_tune_get("x9_wgrad3_swap")) : -1;
  // block tile = 128 columns of one operand x 128 (or 64, when the last 128 would be at most half
  // full) of the other: the operand roles that pad the I x J output least
  auto padded = [](int wide, int other) {
    const bool nar = other % 128 != 0 && other % 128 <= 64;
    return (long long)c2d_ceil_div(wide, 128) * 128 * c2d_ceil_div(other, nar ? 64 : 128) * (nar ? 64 : 128);
  };
  const bool swapped = swap_env >= 0 ? swap_env != 0 : padded(a.J, a.I) < padded(a.I, a.J);
  a.M = n * wc * wc;                 // rows of the "G" operand's descriptor
  a.a_rows = (long long)n * wcx * wcx;   // rows of the "A" operand's descriptor
  if (swapped) {
    WgradArgs b = a;
    b.A = a.G; b.lda = a.ldg; b.a_off = a.g_off; b.I = a.J; b.a_rows = a.M;
    b.G = a.A; b.ldg = a.lda; b.g_off = a.a_off; b.J = a.I; b.M = (int)a.a_rows;
    a = b;
  }
  Wgrad3X9Args q;
  const bool narrow = a.J % 128 != 0 && a.J % 128 <= 64;
  const int bj = narrow ? 64 : 128;
  a.tiles_x = c2d_ceil_div(a.I, 128);
  a.tiles_y = c2d_ceil_div(a.J, bj);
  const int tiles = a.tiles_x * a.tiles_y;
  q.hw = wc * wc; q.wc = wc; q.nimg = n; q.swapped = swapped;
  q.hwx = wcx * wcx; q.wcx = wcx; q.stride = stride;
  int valid[9], sum = 0;
  for (int t = 0; t < 9; ++t) {
    const int ky = t / 3, kx = t % 3;
    const int ny = std::min(wc, (wcx - ky) / stride + 1) - (ky == 0), nx = std::min(wc, (wcx - kx) / stride + 1) - (kx == 0);
    valid[t] = ny * nx;
    sum += valid[t];
  }
  // equal k-steps per block: images per block of tap t = C / valid[t], C from one round of `slots`
  const long long c = (long long)n * tiles * sum / c2d_cu_scaled(slots_env);
  int first = 0;
  for (int t = 0; t < 9; ++t) {
    int ipb = (int)(c / valid[t]) / 16 * 16;
    if (ipb < 64) ipb = 64;
    if (ipb > (n + 15) / 16 * 16) ipb = (n + 15) / 16 * 16;
    q.ipb[t] = ipb;
    q.first[t] = first;
    first += c2d_ceil_div(n, ipb) * tiles;
  }
  q.first[9] = first;
  q.w = a;
  const dim3 grid(first), block(256);
  dispatch_note_ext(swapped ? "wgrad3x3_x9_kernel<%d, true>" : "wgrad3x3_x9_kernel<%d, false>", narrow ? 1 : 2);
  if (swapped) {
    if (narrow) hipLaunchKernelGGL((wgrad3x3_x9_kernel<1, true>), grid, block, 0, s, q);
    else hipLaunchKernelGGL((wgrad3x3_x9_kernel<2, true>), grid, block, 0, s, q);
  } else {
    if (narrow) hipLaunchKernelGGL((wgrad3x3_x9_kernel<1, false>), grid, block, 0, s, q);
    else hipLaunchKernelGGL((wgrad3x3_x9_kernel<2, false>), grid, block, 0, s, q);
  }
  return c2d_launch_status();
}

}  // namespace c2d_ig

using namespace c2d_ig;

extern "C" int c2d_split3_bf16(const float* src, void* planes, long long plane_stride, long long n,
                               void* stream) {
  C2D_CHECK_ARG(src && planes && n > 0 && n % 4 == 0 && plane_stride >= n && plane_stride % 4 == 0);
  C2D_CHECK_ARG(((uintptr_t)src & 15) == 0 && ((uintptr_t)planes & 7) == 0);
  const long long n4 = n / 4;
  hipLaunchKernelGGL(split3_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     src, (unsigned short*)planes, plane_stride, n4);
  return c2d_launch_status();
}

extern "C" int c2d_f32x9_bind(const float* arena, long long numel, const void* planes,
                              long long plane_stride) {
  C2D_CHECK_ARG(arena && planes && numel > 0 && plane_stride >= numel);
  C2D_CHECK_ARG(((uintptr_t)arena & 15) == 0 && ((uintptr_t)planes & 15) == 0 && plane_stride % 8 == 0);
  // (the ring reaches plane 2 through a 32-bit scalar offset of a descriptor over all three planes)
  C2D_CHECK_ARG(plane_stride * 2 * 3 < (long long)OOB_OFFSET);
  std::lock_guard<std::mutex> lock(g_arena_mu);
  int slot = g_narena;
  for (int i = 0; i < g_narena; ++i)
    if (g_arena[i].base == (const char*)arena) slot = i;
  if (slot == X9_MAX_ARENAS) return C2D_ERR_WORKSPACE;
  g_arena[slot] = X9Arena{(const char*)arena, numel * 4, (const char*)planes, plane_stride * 2};
  if (slot == g_narena) ++g_narena;
  return C2D_OK;
}

extern "C" int c2d_f32x9_unbind(const float* arena) {
  std::lock_guard<std::mutex> lock(g_arena_mu);
  for (int i = 0; i < g_narena; ++i) {
    if (arena == nullptr || g_arena[i].base == (const char*)arena) {
      g_arena[i] = g_arena[g_narena - 1];
      --g_narena;
      --i;
    }
  }
  return C2D_OK;
}

extern "C" int c2d_f32x9_enable(int on) {
  const int was = g_enabled;
  g_enabled = on != 0;
  return was;
}
