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
// l = (h = l>>5, i = l&31) feeds A[i][k], B[k][i]; a BK = 16 slab is consumed as 8 MFMA steps
// where half-wave h takes k = 8h + s (a permutation of the k order common to A and B, which
// lets each lane fetch its 8 k-values with two ds_read_b128).  fp32 in, fp32 accumulate:
// bit-for-bit an fmaf chain, no reduced precision anywhere.
#include "c2d_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 16;
constexpr int LDS_STRIDE = BK + 4;  // floats; 80-byte rows: 16-B aligned, b128 reads conflict-free

struct ConvGeom {
  int ih, iw;      // conv input spatial size
  int oh, ow;      // conv output spatial size
  int kh, kw;      // taps
  int stride;
  int pad_t, pad_l;
  int mode;        // 0: rows index conv OUTPUT pixels (forward / wgrad); 1: rows index conv
                   // INPUT pixels and src() yields OUTPUT pixels (dgrad)
};

// Row m of the iteration space -> (image, y, x).
struct RowPos {
  int img, y, x;
  bool valid;
};

__device__ __forceinline__ RowPos decompose(int m, int M, int h, int w) {
  RowPos p;
  p.valid = m < M;
  const int mm = p.valid ? m : 0;
  p.img = mm / (h * w);
  const int r = mm - p.img * h * w;
  p.y = r / w;
  p.x = r - p.y * w;
  return p;
}

// Source row (in the A operand's row space) for iteration row `p` and tap (ky,kx); -1 if none.
__device__ __forceinline__ int src_row(const ConvGeom& g, const RowPos& p, int ky, int kx) {
  if (!p.valid) return -1;
  if (g.mode == 0) {
    const int iy = p.y * g.stride - g.pad_t + ky;
    const int ix = p.x * g.stride - g.pad_l + kx;
    if (iy < 0 || iy >= g.ih || ix < 0 || ix >= g.iw) return -1;
    return (p.img * g.ih + iy) * g.iw + ix;
  } else {
    const int ty = p.y + g.pad_t - ky;
    const int tx = p.x + g.pad_l - kx;
    if (ty < 0 || tx < 0) return -1;
    if (g.stride > 1 && ((ty % g.stride) != 0 || (tx % g.stride) != 0)) return -1;
    const int oy = ty / g.stride, ox = tx / g.stride;
    if (oy >= g.oh || ox >= g.ow) return -1;
    return (p.img * g.oh + oy) * g.ow + ox;
  }
}

struct IgemmArgs {
  const float* A; int lda; int a_off;
  const float* Bt;          // [taps][N][K]
  float* C; int ldc; int c_off;
  const float* scale;       // [N] or null (=1)
  const float* shift;       // [N] or null (=0)
  int relu;
  int accumulate;           // C += result
  int M, N, K;
  ConvGeom g;
};

template <int WM, int WN, int MT, int NT>
__global__ __launch_bounds__(WM * WN * 64) void igemm_nt_kernel(IgemmArgs a) {
  constexpr int BM = WM * MT * 32;
  constexpr int BN = WN * NT * 32;
  constexpr int NTHREADS = WM * WN * 64;
  constexpr int A_LOADS = BM * (BK / 4) / NTHREADS;  // float4 per thread per slab
  constexpr int B_LOADS = BN * (BK / 4) / NTHREADS;
  static_assert(A_LOADS >= 1 && B_LOADS >= 1, "tile too small for the block");
  __shared__ __attribute__((aligned(16))) float As[BM * LDS_STRIDE];
  __shared__ __attribute__((aligned(16))) float Bs[BN * LDS_STRIDE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;

  // loader coordinates
  const int q = tid & 3;  // which float4 of the 16-float k slab
  RowPos apos[A_LOADS];
  int arow_l[A_LOADS];
#pragma unroll
  for (int i = 0; i < A_LOADS; ++i) {
    arow_l[i] = (tid >> 2) + i * (NTHREADS / 4);
    apos[i] = decompose(m0 + arow_l[i], a.M, a.g.mode == 0 ? a.g.oh : a.g.ih,
                        a.g.mode == 0 ? a.g.ow : a.g.iw);
  }
  int brow_l[B_LOADS];
  bool bvalid[B_LOADS];
#pragma unroll
  for (int i = 0; i < B_LOADS; ++i) {
    brow_l[i] = (tid >> 2) + i * (NTHREADS / 4);
    bvalid[i] = (n0 + brow_l[i]) < a.N;
  }

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int ksteps = a.K / BK;
  const int ntaps = a.g.kh * a.g.kw;
  const int total = ntaps * ksteps;

  float4 ra[A_LOADS], rb[B_LOADS];
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);

  auto load_slab = [&](int it) {
    const int tap = it / ksteps;
    const int kc = (it - tap * ksteps) * BK + q * 4;
    const int ky = tap / a.g.kw, kx = tap - ky * a.g.kw;
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i) {
      const int s = src_row(a.g, apos[i], ky, kx);
      ra[i] = s >= 0 ? *reinterpret_cast<const float4*>(a.A + (size_t)s * a.lda + a.a_off + kc)
                     : zero4;
    }
    const float* bt = a.Bt + (size_t)tap * a.N * a.K;
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i) {
      rb[i] = bvalid[i]
                  ? *reinterpret_cast<const float4*>(bt + (size_t)(n0 + brow_l[i]) * a.K + kc)
                  : zero4;
    }
  };

  load_slab(0);
  for (int it = 0; it < total; ++it) {
#pragma unroll
    for (int i = 0; i < A_LOADS; ++i)
      *reinterpret_cast<float4*>(&As[arow_l[i] * LDS_STRIDE + q * 4]) = ra[i];
#pragma unroll
    for (int i = 0; i < B_LOADS; ++i)
      *reinterpret_cast<float4*>(&Bs[brow_l[i] * LDS_STRIDE + q * 4]) = rb[i];
    __syncthreads();
    if (it + 1 < total) load_slab(it + 1);  // in flight during the MFMAs below

    float af[MT][8], bf[NT][8];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const float* p = &As[(wm * MT * 32 + i * 32 + li) * LDS_STRIDE + lh * 8];
      const float4 v0 = *reinterpret_cast<const float4*>(p);
      const float4 v1 = *reinterpret_cast<const float4*>(p + 4);
      af[i][0] = v0.x; af[i][1] = v0.y; af[i][2] = v0.z; af[i][3] = v0.w;
      af[i][4] = v1.x; af[i][5] = v1.y; af[i][6] = v1.z; af[i][7] = v1.w;
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const float* p = &Bs[(wn * NT * 32 + j * 32 + li) * LDS_STRIDE + lh * 8];
      const float4 v0 = *reinterpret_cast<const float4*>(p);
      const float4 v1 = *reinterpret_cast<const float4*>(p + 4);
      bf[j][0] = v0.x; bf[j][1] = v0.y; bf[j][2] = v0.z; bf[j][3] = v0.w;
      bf[j][4] = v1.x; bf[j][5] = v1.y; bf[j][6] = v1.z; bf[j][7] = v1.w;
    }
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], bf[j][s], acc[i][j], 0, 0, 0);
    __syncthreads();
  }

  // Epilogue.  C/D map of the 32x32 tile: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + wn * NT * 32 + j * 32 + li;
    if (n >= a.N) continue;
    const float sc = a.scale ? a.scale[n] : 1.0f;
    const float sh = a.shift ? a.shift[n] : 0.0f;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * MT * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= a.M) continue;
        float v = acc[i][j][r] * sc + sh;
        if (a.relu) v = fmaxf(v, 0.0f);
        float* dst = a.C + (size_t)m * a.ldc + a.c_off + n;
        if (a.accumulate) v += *dst;
        *dst = v;
      }
    }
  }
}

struct WgradArgs {
  const float* A; int lda; int a_off;   // activations x (rows of the conv input)
  const float* G; int ldg; int g_off;   // dC rows (conv output rows)
  float* dW;                            // [taps][I][J], pre-zeroed or accumulated into
  int M;                                // conv output rows (reduction length)
  int I, J;                             // cin, cout
  int rows_per_split;                   // multiple of BK
  ConvGeom g;                           // mode 0
};

constexpr int WG_STRIDE = 128 + 4;  // floats per k-row of the [BK][128] tiles

// Block tile 128(i) x 128(j); 4 waves as 2x2, each 64x64 (2x2 MFMA tiles).
__global__ __launch_bounds__(256) void wgrad_tn_kernel(WgradArgs a) {
  __shared__ __attribute__((aligned(16))) float As[BK * WG_STRIDE];
  __shared__ __attribute__((aligned(16))) float Gs[BK * WG_STRIDE];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int li = lane & 31, lh = lane >> 5;
  const int itiles = (a.I + 127) / 128;
  const int tap = blockIdx.x / itiles;
  const int i0 = (blockIdx.x - tap * itiles) * 128;
  const int j0 = blockIdx.y * 128;
  const int ky = tap / a.g.kw, kx = tap - ky * a.g.kw;
  const int mbeg = blockIdx.z * a.rows_per_split;
  const int mend = min(a.M, mbeg + a.rows_per_split);

  // loader: thread -> (k row kr and kr+8, float4 column c4)
  const int kr = tid >> 5;   // 0..7
  const int c4 = (tid & 31) * 4;
  const bool ivalid = (i0 + c4) < a.I;   // I, J multiples of 4
  const bool jvalid = (j0 + c4) < a.J;

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  float4 ra[2], rg[2];
  auto load_slab = [&](int mb) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int m = mb + kr + u * 8;
      const RowPos p = decompose(m, mend, a.g.oh, a.g.ow);
      const int s = src_row(a.g, p, ky, kx);
      ra[u] = (s >= 0 && ivalid)
                  ? *reinterpret_cast<const float4*>(a.A + (size_t)s * a.lda + a.a_off + i0 + c4)
                  : zero4;
      rg[u] = (p.valid && jvalid)
                  ? *reinterpret_cast<const float4*>(a.G + (size_t)m * a.ldg + a.g_off + j0 + c4)
                  : zero4;
    }
  };

  if (mbeg < mend) load_slab(mbeg);
  for (int mb = mbeg; mb < mend; mb += BK) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      *reinterpret_cast<float4*>(&As[(kr + u * 8) * WG_STRIDE + c4]) = ra[u];
      *reinterpret_cast<float4*>(&Gs[(kr + u * 8) * WG_STRIDE + c4]) = rg[u];
    }
    __syncthreads();
    if (mb + BK < mend) load_slab(mb + BK);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const int k = lh * 8 + s;
      float af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = As[k * WG_STRIDE + wm * 64 + i * 32 + li];
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[j] = Gs[k * WG_STRIDE + wn * 64 + j * 32 + li];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  float* dw = a.dW + (size_t)tap * a.I * a.J;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int jj = j0 + wn * 64 + j * 32 + li;
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

int fill_geom(ConvGeom* g, int ih, int iw, int kh, int kw, int stride, int mode) {
  if (ih <= 0 || iw <= 0 || kh <= 0 || kw <= 0 || stride <= 0) return C2D_ERR_INVALID_ARG;
  g->ih = ih; g->iw = iw; g->kh = kh; g->kw = kw; g->stride = stride; g->mode = mode;
  g->oh = (ih + stride - 1) / stride;
  g->ow = (iw + stride - 1) / stride;
  // TF 'SAME': pad_total = max((out-1)*stride + k - in, 0), the extra pixel goes bottom/right.
  const int pth = (g->oh - 1) * stride + kh - ih, ptw = (g->ow - 1) * stride + kw - iw;
  g->pad_t = (pth > 0 ? pth : 0) / 2;
  g->pad_l = (ptw > 0 ? ptw : 0) / 2;
  return C2D_OK;
}

template <int WM, int WN, int MT, int NT>
void launch_igemm(const IgemmArgs& a, hipStream_t s) {
  constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
  dim3 grid(c2d_ceil_div(a.M, BM), c2d_ceil_div(a.N, BN));
  hipLaunchKernelGGL((igemm_nt_kernel<WM, WN, MT, NT>), grid, dim3(WM * WN * 64), 0, s, a);
}

int run_igemm(const IgemmArgs& a, hipStream_t s) {
  if (a.M <= 0 || a.N <= 0) return C2D_OK;
  // Tile choice: big tiles when the grid still fills 256 CUs, otherwise 64x64 tiles.
  const long long big_blocks = (long long)c2d_ceil_div(a.M, 128) * c2d_ceil_div(a.N, 128);
  if (big_blocks >= 256) launch_igemm<2, 2, 2, 2>(a, s);
  else launch_igemm<2, 2, 1, 1>(a, s);
  return c2d_launch_status();
}

}  // namespace

extern "C" int c2d_conv_fwd(const float* x, int ldx, int xoff, const float* wt,
                            const float* scale, const float* shift, float* y, int ldy,
                            int yoff, int n, int ih, int iw, int cin, int cout, int kh, int kw,
                            int stride, int relu, void* stream) {
  C2D_CHECK_ARG(x && wt && y && n > 0 && cin > 0 && cout > 0);
  C2D_CHECK_ARG(cin % BK == 0 && ldx % 4 == 0 && xoff % 4 == 0);
  IgemmArgs a;
  int rc = fill_geom(&a.g, ih, iw, kh, kw, stride, 0);
  if (rc) return rc;
  a.A = x; a.lda = ldx; a.a_off = xoff; a.Bt = wt; a.C = y; a.ldc = ldy; a.c_off = yoff;
  a.scale = scale; a.shift = shift; a.relu = relu; a.accumulate = 0;
  a.M = n * a.g.oh * a.g.ow; a.N = cout; a.K = cin;
  return run_igemm(a, (hipStream_t)stream);
}

extern "C" int c2d_conv_dgrad(const float* dc, int ldc, int coff, const float* w, float* dx,
                              int lddx, int dxoff, int n, int ih, int iw, int cin, int cout,
                              int kh, int kw, int stride, int accumulate, void* stream) {
  C2D_CHECK_ARG(dc && w && dx && n > 0 && cin > 0 && cout > 0);
  C2D_CHECK_ARG(cout % BK == 0 && ldc % 4 == 0 && coff % 4 == 0);
  IgemmArgs a;
  int rc = fill_geom(&a.g, ih, iw, kh, kw, stride, 1);
  if (rc) return rc;
  a.A = dc; a.lda = ldc; a.a_off = coff; a.Bt = w; a.C = dx; a.ldc = lddx; a.c_off = dxoff;
  a.scale = nullptr; a.shift = nullptr; a.relu = 0; a.accumulate = accumulate;
  a.M = n * ih * iw; a.N = cin; a.K = cout;
  return run_igemm(a, (hipStream_t)stream);
}

extern "C" int c2d_conv_wgrad(const float* x, int ldx, int xoff, const float* dc, int ldc,
                              int coff, float* dw, int n, int ih, int iw, int cin, int cout,
                              int kh, int kw, int stride, void* stream) {
  C2D_CHECK_ARG(x && dc && dw && n > 0 && cin > 0 && cout > 0);
  C2D_CHECK_ARG(cin % 4 == 0 && cout % 4 == 0 && ldx % 4 == 0 && xoff % 4 == 0);
  C2D_CHECK_ARG(ldc % 4 == 0 && coff % 4 == 0);
  WgradArgs a;
  int rc = fill_geom(&a.g, ih, iw, kh, kw, stride, 0);
  if (rc) return rc;
  a.A = x; a.lda = ldx; a.a_off = xoff; a.G = dc; a.ldg = ldc; a.g_off = coff; a.dW = dw;
  a.M = n * a.g.oh * a.g.ow; a.I = cin; a.J = cout;
  const int tiles = kh * kw * c2d_ceil_div(cin, 128) * c2d_ceil_div(cout, 128);
  int splits = c2d_ceil_div(1024, tiles);                 // aim at ~4 blocks per CU
  const int max_splits = c2d_ceil_div(a.M, 4 * BK);       // at least 4 slabs per block
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  a.rows_per_split = c2d_ceil_div(c2d_ceil_div(a.M, splits), BK) * BK;
  splits = c2d_ceil_div(a.M, a.rows_per_split);
  dim3 grid(kh * kw * c2d_ceil_div(cin, 128), c2d_ceil_div(cout, 128), splits);
  hipLaunchKernelGGL(wgrad_tn_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
  return c2d_launch_status();
}
