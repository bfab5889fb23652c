// HBM-bound NHWC kernels around the convolutions: pools, BN/ReLU backward, spatial mean +
// dropout, preprocessing, im2col for the 7x7 stem, weight transposes, Adagrad.
// All are float4-vectorised along channels (16 B per lane, coalesced).
//
// Reference call sites replaced: slim.max_pool2d / slim.avg_pool2d inside the Inception-V2
// extractor [3P] (models/utils.py:133-136,165-167), tf.reduce_mean + slim.dropout
// (models/utils.py:169-174), FasterRCNN preprocess (models/utils.py:127), the BatchNorm/ReLU
// gradients TF derives for train_op (train/trainer.py:141-146) and
// tf.train.AdagradOptimizer (core/training_utils.py:45-50).
#include "c2d_common.h"

namespace {

struct PoolGeom {
  int ih, iw, oh, ow, stride, pad_t, pad_l;
};

__host__ __device__ inline PoolGeom make_pool_geom(int ih, int iw, int stride) {
  PoolGeom g;
  g.ih = ih; g.iw = iw; g.stride = stride;
  g.oh = (ih + stride - 1) / stride;
  g.ow = (iw + stride - 1) / stride;
  const int pth = (g.oh - 1) * stride + 3 - ih, ptw = (g.ow - 1) * stride + 3 - iw;
  g.pad_t = (pth > 0 ? pth : 0) / 2;
  g.pad_l = (ptw > 0 ? ptw : 0) / 2;
  return g;
}

// mode 0: max (writes uint8 argmax = ky*3+kx of the first maximum), mode 1: avg over valid cells.
template <typename T>
__global__ __launch_bounds__(256) void pool3x3_fwd_kernel(
    const T* __restrict__ x, int ldx, int xoff, T* __restrict__ y, int ldy, int yoff,
    uint8_t* __restrict__ arg, int n, int c4n, PoolGeom g, int mode, int relu) {
  const long long total = (long long)n * g.oh * g.ow * c4n;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % c4n);
    const long long row = idx / c4n;
    const int ox = (int)(row % g.ow);
    const int oy = (int)((row / g.ow) % g.oh);
    const int img = (int)(row / ((long long)g.ow * g.oh));
    float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
    uchar4 am = make_uchar4(0, 0, 0, 0);
    int cnt = 0;
    bool first = true;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = oy * g.stride - g.pad_t + ky;
      if (iy < 0 || iy >= g.ih) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = ox * g.stride - g.pad_l + kx;
        if (ix < 0 || ix >= g.iw) continue;
        const float4 v = c2d_ld4(x + ((size_t)(img * g.ih + iy) * g.iw + ix) * ldx + xoff + c4 * 4);
        const unsigned char k = (unsigned char)(ky * 3 + kx);
        if (first) {
          best = v; am = make_uchar4(k, k, k, k); first = false;
        } else {
          if (v.x > best.x) { best.x = v.x; am.x = k; }
          if (v.y > best.y) { best.y = v.y; am.y = k; }
          if (v.z > best.z) { best.z = v.z; am.z = k; }
          if (v.w > best.w) { best.w = v.w; am.w = k; }
        }
        sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
        ++cnt;
      }
    }
    float4 out;
    if (mode == 0) {
      out = best;
      if (arg) *reinterpret_cast<uchar4*>(arg + (size_t)row * c4n * 4 + c4 * 4) = am;
    } else {
      const float d = (float)cnt;
      out = make_float4(sum.x / d, sum.y / d, sum.z / d, sum.w / d);
    }
    if (relu) out = make_float4(fmaxf(out.x, 0.f), fmaxf(out.y, 0.f), fmaxf(out.z, 0.f), fmaxf(out.w, 0.f));
    c2d_st4(y + (size_t)row * ldy + yoff + c4 * 4, out);
  }
}

// Gather form of the pool gradient: one lane per INPUT element loops over the <= 9 outputs
// whose window contains it (no atomics, deterministic).
template <typename T>
__global__ __launch_bounds__(256) void pool3x3_bwd_kernel(
    const T* __restrict__ dy, int lddy, int dyoff, const uint8_t* __restrict__ arg,
    T* __restrict__ dx, int lddx, int dxoff, int n, int c4n, PoolGeom g, int mode,
    int accumulate, const T* __restrict__ ymask, int ldym, int ymoff) {
  const long long total = (long long)n * g.ih * g.iw * c4n;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % c4n);
    const long long row = idx / c4n;
    const int ix = (int)(row % g.iw);
    const int iy = (int)((row / g.iw) % g.ih);
    const int img = (int)(row / ((long long)g.iw * g.ih));
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int ty = iy + g.pad_t - ky;
      if (ty < 0 || (ty % g.stride) != 0) continue;
      const int oy = ty / g.stride;
      if (oy >= g.oh) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int tx = ix + g.pad_l - kx;
        if (tx < 0 || (tx % g.stride) != 0) continue;
        const int ox = tx / g.stride;
        if (ox >= g.ow) continue;
        const size_t orow = (size_t)(img * g.oh + oy) * g.ow + ox;
        float4 gy = c2d_ld4(dy + orow * lddy + dyoff + c4 * 4);
        if (ymask) {       // the pool's output went through a ReLU: its gradient passes where y > 0
          const float4 yv = c2d_ld4(ymask + orow * ldym + ymoff + c4 * 4);
          gy.x = yv.x > 0.f ? gy.x : 0.f; gy.y = yv.y > 0.f ? gy.y : 0.f;
          gy.z = yv.z > 0.f ? gy.z : 0.f; gy.w = yv.w > 0.f ? gy.w : 0.f;
        }
        if (mode == 0) {
          const uchar4 am = *reinterpret_cast<const uchar4*>(arg + orow * c4n * 4 + c4 * 4);
          const unsigned char k = (unsigned char)(ky * 3 + kx);
          if (am.x == k) acc.x += gy.x;
          if (am.y == k) acc.y += gy.y;
          if (am.z == k) acc.z += gy.z;
          if (am.w == k) acc.w += gy.w;
        } else {
          // number of valid cells of output (oy,ox)'s window
          const int y0 = oy * g.stride - g.pad_t, x0 = ox * g.stride - g.pad_l;
          const int ny = min(y0 + 3, g.ih) - max(y0, 0);
          const int nx = min(x0 + 3, g.iw) - max(x0, 0);
          const float d = (float)(ny * nx);
          acc.x += gy.x / d; acc.y += gy.y / d; acc.z += gy.z / d; acc.w += gy.w / d;
        }
      }
    }
    T* dst = dx + (size_t)row * lddx + dxoff + c4 * 4;
    if (accumulate) {
      const float4 o = c2d_ld4(dst);
      acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
    }
    c2d_st4(dst, acc);
  }
}

// ---------------------------------------------------------------------------------------------
// Whole-map forms for the per-ROI maps of the second stage (4x4 stride 1, 7x7 stride 2).
// One lane owns one (ROI, float4 channel group) and walks the map with compile-time pixel
// indices: every input element is fetched from memory ONCE (the generic kernels above fetch it
// up to nine times, and with rows dealt round-robin to the eight XCD L2s the re-reads went to
// the fabric: PMC showed 4.7 GB fetched for 0.9 GB of tensors).  Arithmetic order (scan order of
// the taps, first-maximum tie rule, sum then divide) is that of the generic kernels.
// ---------------------------------------------------------------------------------------------
template <int IH, int STRIDE, int MODE, typename T>
__global__ __launch_bounds__(256) void pool3x3_map_fwd_kernel(
    const T* __restrict__ x, int ldx, int xoff, T* __restrict__ y, int ldy, int yoff,
    uint8_t* __restrict__ arg, int n, int c4n) {
  constexpr int IW = IH;
  constexpr int OH = (IH + STRIDE - 1) / STRIDE, OW = OH;
  constexpr int PT = ((OH - 1) * STRIDE + 3 - IH) > 0 ? ((OH - 1) * STRIDE + 3 - IH) / 2 : 0;
  const long long total = (long long)n * c4n;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % c4n);
    const long long img = idx / c4n;
    const T* xp = x + (size_t)img * (IH * IW) * ldx + xoff + c4 * 4;
    T* yp = y + (size_t)img * (OH * OW) * ldy + yoff + c4 * 4;
    uint8_t* ap = arg ? arg + ((size_t)img * (OH * OW) * c4n + c4) * 4 : nullptr;
    float4 rows[3][IW];   // the three input rows of the current output row
#pragma unroll
    for (int oy = 0; oy < OH; ++oy) {
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * STRIDE - PT + ky;
        if (iy < 0 || iy >= IH) continue;
        if (STRIDE == 2 && ky == 0 && oy > 0) {   // row shared with the previous output row
#pragma unroll
          for (int ix = 0; ix < IW; ++ix) rows[0][ix] = rows[2][ix];
          continue;
        }
        if (STRIDE == 1 && oy > 0 && ky < 2) {    // slide the window down by one row
#pragma unroll
          for (int ix = 0; ix < IW; ++ix) rows[ky][ix] = rows[ky + 1][ix];
          continue;
        }
#pragma unroll
        for (int ix = 0; ix < IW; ++ix)
          rows[ky][ix] = c2d_ld4(xp + (size_t)(iy * IW + ix) * ldx);
      }
#pragma unroll
      for (int ox = 0; ox < OW; ++ox) {
        float4 best = make_float4(0.f, 0.f, 0.f, 0.f), sum = best;
        uchar4 am = make_uchar4(0, 0, 0, 0);
        int cnt = 0;
        bool first = true;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int iy = oy * STRIDE - PT + ky;
          if (iy < 0 || iy >= IH) continue;
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * STRIDE - PT + kx;
            if (ix < 0 || ix >= IW) continue;
            const float4 v = rows[ky][ix];
            const unsigned char k = (unsigned char)(ky * 3 + kx);
            if (MODE == 0) {
              if (first) {
                best = v; am = make_uchar4(k, k, k, k); first = false;
              } else {
                if (v.x > best.x) { best.x = v.x; am.x = k; }
                if (v.y > best.y) { best.y = v.y; am.y = k; }
                if (v.z > best.z) { best.z = v.z; am.z = k; }
                if (v.w > best.w) { best.w = v.w; am.w = k; }
              }
            } else {
              sum.x += v.x; sum.y += v.y; sum.z += v.z; sum.w += v.w;
              ++cnt;
            }
          }
        }
        float4 out;
        if (MODE == 0) {
          out = best;
          if (ap) *reinterpret_cast<uchar4*>(ap + (size_t)(oy * OW + ox) * c4n * 4) = am;
        } else {
          const float d = (float)cnt;
          out = make_float4(sum.x / d, sum.y / d, sum.z / d, sum.w / d);
        }
        c2d_st4(yp + (size_t)(oy * OW + ox) * ldy, out);
      }
    }
  }
}

// Gradient: the lane keeps the whole dy map (and arg-max map) of its ROI in registers and emits
// each input pixel's gradient from the <= 9 outputs whose window holds it, in the generic
// kernel's (ky, kx) order.
template <int IH, int STRIDE, int MODE, typename T, bool YM = false>
__global__ __launch_bounds__(256) void pool3x3_map_bwd_kernel(
    const T* __restrict__ dy, int lddy, int dyoff, const uint8_t* __restrict__ arg,
    T* __restrict__ dx, int lddx, int dxoff, int n, int c4n, int accumulate,
    const T* __restrict__ ymask = nullptr, int ldym = 0, int ymoff = 0) {
  constexpr int IW = IH;
  constexpr int OH = (IH + STRIDE - 1) / STRIDE, OW = OH;
  constexpr int PT = ((OH - 1) * STRIDE + 3 - IH) > 0 ? ((OH - 1) * STRIDE + 3 - IH) / 2 : 0;
  const long long total = (long long)n * c4n;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % c4n);
    const long long img = idx / c4n;
    const T* gp = dy + (size_t)img * (OH * OW) * lddy + dyoff + c4 * 4;
    T* dp = dx + (size_t)img * (IH * IW) * lddx + dxoff + c4 * 4;
    float4 g[OH * OW];
    uchar4 am[OH * OW];
#pragma unroll
    for (int o = 0; o < OH * OW; ++o) {
      g[o] = c2d_ld4(gp + (size_t)o * lddy);
      if (YM) {            // the pool's output went through a ReLU: its gradient passes where y > 0
        const float4 yv = c2d_ld4(ymask + ((size_t)img * (OH * OW) + o) * ldym + ymoff + c4 * 4);
        g[o].x = yv.x > 0.f ? g[o].x : 0.f; g[o].y = yv.y > 0.f ? g[o].y : 0.f;
        g[o].z = yv.z > 0.f ? g[o].z : 0.f; g[o].w = yv.w > 0.f ? g[o].w : 0.f;
      }
      if (MODE == 0)
        am[o] = *reinterpret_cast<const uchar4*>(arg + ((size_t)img * (OH * OW) + o) * c4n * 4 +
                                                 c4 * 4);
    }
#pragma unroll
    for (int iy = 0; iy < IH; ++iy) {
#pragma unroll
      for (int ix = 0; ix < IW; ++ix) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const int ty = iy + PT - ky;
          if (ty < 0 || (ty % STRIDE) != 0 || ty / STRIDE >= OH) continue;
          const int oy = ty / STRIDE;
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const int tx = ix + PT - kx;
            if (tx < 0 || (tx % STRIDE) != 0 || tx / STRIDE >= OW) continue;
            const int ox = tx / STRIDE;
            const float4 gy = g[oy * OW + ox];
            if (MODE == 0) {
              const uchar4 a = am[oy * OW + ox];
              const unsigned char k = (unsigned char)(ky * 3 + kx);
              if (a.x == k) acc.x += gy.x;
              if (a.y == k) acc.y += gy.y;
              if (a.z == k) acc.z += gy.z;
              if (a.w == k) acc.w += gy.w;
            } else {
              const int y0 = oy * STRIDE - PT, x0 = ox * STRIDE - PT;
              const int ny = (y0 + 3 < IH ? y0 + 3 : IH) - (y0 > 0 ? y0 : 0);
              const int nx = (x0 + 3 < IW ? x0 + 3 : IW) - (x0 > 0 ? x0 : 0);
              const float d = (float)(ny * nx);
              acc.x += gy.x / d; acc.y += gy.y / d; acc.z += gy.z / d; acc.w += gy.w / d;
            }
          }
        }
        T* dst = dp + (size_t)(iy * IW + ix) * lddx;
        if (accumulate) {
          const float4 o = c2d_ld4(dst);
          acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
        }
        c2d_st4(dst, acc);
      }
    }
  }
}

// dc = dy * (y > 0) * scale[c];  dbeta[c] += sum dz;  dgamma[c] += sum dz * (y - beta)/gamma
// where dz = dy * (y > 0).  (y = gamma*xhat + beta wherever y > 0, so xhat = (y-beta)/gamma.)
// Block: TX lanes over float4 channel groups x TY row lanes; LDS reduce over TY.
// HEAD: dy is not read but derived from the gradient of the spatially averaged (and dropped-out)
// features, dy[r][c] = dmean[r / spatial][c] * mask[r / spatial][c] * mul — the backward of
// c2d_spatial_mean_dropout_fwd folded into this kernel (c2d_bn_relu_bwd_partial_head).
struct HeadGrad {
  const float* dmean; int ld, off;
  const uint8_t* mask; int mask_ld, mask_off;
  int spatial; float mul;
  int norelu;      // 1: the layer has no ReLU behind its BatchNorm (c2d_bn_bwd_partial): dz = dy
};

template <bool HEAD, typename T>
__device__ __forceinline__ float4 bn_bwd_dy(const T* dy, int lddy, int dyoff, const HeadGrad& hg,
                                            int r, int c) {
  if constexpr (HEAD) {
    const int rr = r / hg.spatial;
    float4 g = *reinterpret_cast<const float4*>(hg.dmean + (size_t)rr * hg.ld + hg.off + c);
    float4 m = make_float4(hg.mul, hg.mul, hg.mul, hg.mul);
    if (hg.mask) {
      const uchar4 k = *reinterpret_cast<const uchar4*>(hg.mask + (size_t)rr * hg.mask_ld + hg.mask_off + c);
      m.x *= (float)k.x; m.y *= (float)k.y; m.z *= (float)k.z; m.w *= (float)k.w;
    }
    return make_float4(g.x * m.x, g.y * m.y, g.z * m.z, g.w * m.w);
  } else {
    return c2d_ld4(dy + (size_t)r * lddy + dyoff + c);
  }
}

template <int TX, typename T, bool HEAD = false>
__global__ __launch_bounds__(256) void bn_relu_bwd_kernel(
    const T* __restrict__ dy, int lddy, int dyoff, const T* __restrict__ y, int ldy,
    int yoff, const float* __restrict__ scale, const float* __restrict__ beta,
    const float* __restrict__ gamma, T* __restrict__ dc, float* __restrict__ dbeta,
    float* __restrict__ dgamma, float* __restrict__ partials, int M, int c4n,
    int rows_per_block, HeadGrad hg) {
  constexpr int TY = 256 / TX;
  __shared__ float4 red[2][TY][TX];
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  for (int base = 0; base < c4n; base += TX) {  // usually a single trip; uniform for barriers
    const int cg = base + tx;
    const bool active = cg < c4n;
    const int c = (active ? cg : 0) * 4;
    const float4 sc = *reinterpret_cast<const float4*>(scale + c);
    float4 sb = make_float4(0.f, 0.f, 0.f, 0.f), sg = sb;
    float4 be = make_float4(0.f, 0.f, 0.f, 0.f), ig = be;
    if (dgamma || (partials && gamma)) {
      be = *reinterpret_cast<const float4*>(beta + c);
      const float4 ga = *reinterpret_cast<const float4*>(gamma + c);
      ig = make_float4(ga.x != 0.f ? 1.f / ga.x : 0.f, ga.y != 0.f ? 1.f / ga.y : 0.f,
                       ga.z != 0.f ? 1.f / ga.z : 0.f, ga.w != 0.f ? 1.f / ga.w : 0.f);
    }
    if (active) {
      // 4 rows per trip: 8 independent 16-B loads in flight per lane
      int r = r0 + ty;
      for (; r + 3 * TY < r1; r += 4 * TY) {
        float4 g[4], v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          g[u] = bn_bwd_dy<HEAD>(dy, lddy, dyoff, hg, r + u * TY, c);
          v[u] = c2d_ld4(y + (size_t)(r + u * TY) * ldy + yoff + c);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          float4 dz;
          dz.x = (v[u].x > 0.f || hg.norelu) ? g[u].x : 0.f; dz.y = (v[u].y > 0.f || hg.norelu) ? g[u].y : 0.f;
          dz.z = (v[u].z > 0.f || hg.norelu) ? g[u].z : 0.f; dz.w = (v[u].w > 0.f || hg.norelu) ? g[u].w : 0.f;
          sb.x += dz.x; sb.y += dz.y; sb.z += dz.z; sb.w += dz.w;
          sg.x += dz.x * (v[u].x - be.x) * ig.x; sg.y += dz.y * (v[u].y - be.y) * ig.y;
          sg.z += dz.z * (v[u].z - be.z) * ig.z; sg.w += dz.w * (v[u].w - be.w) * ig.w;
          c2d_st4(dc + (size_t)(r + u * TY) * c4n * 4 + c,
                  make_float4(dz.x * sc.x, dz.y * sc.y, dz.z * sc.z, dz.w * sc.w));
        }
      }
      for (; r < r1; r += TY) {
        const float4 g = bn_bwd_dy<HEAD>(dy, lddy, dyoff, hg, r, c);
        const float4 v = c2d_ld4(y + (size_t)r * ldy + yoff + c);
        float4 dz;
        dz.x = (v.x > 0.f || hg.norelu) ? g.x : 0.f; dz.y = (v.y > 0.f || hg.norelu) ? g.y : 0.f;
        dz.z = (v.z > 0.f || hg.norelu) ? g.z : 0.f; dz.w = (v.w > 0.f || hg.norelu) ? g.w : 0.f;
        sb.x += dz.x; sb.y += dz.y; sb.z += dz.z; sb.w += dz.w;
        sg.x += dz.x * (v.x - be.x) * ig.x; sg.y += dz.y * (v.y - be.y) * ig.y;
        sg.z += dz.z * (v.z - be.z) * ig.z; sg.w += dz.w * (v.w - be.w) * ig.w;
        c2d_st4(dc + (size_t)r * c4n * 4 + c,
                make_float4(dz.x * sc.x, dz.y * sc.y, dz.z * sc.z, dz.w * sc.w));
      }
    }
    red[0][ty][tx] = sb;
    red[1][ty][tx] = sg;
    __syncthreads();
    if (ty == 0 && active) {
      float4 tb = make_float4(0.f, 0.f, 0.f, 0.f), tg = tb;
      for (int k = 0; k < TY; ++k) {
        const float4 b = red[0][k][tx], gq = red[1][k][tx];
        tb.x += b.x; tb.y += b.y; tb.z += b.z; tb.w += b.w;
        tg.x += gq.x; tg.y += gq.y; tg.z += gq.z; tg.w += gq.w;
      }
      if (partials) {
        // [block][2][C]: summed in block order by bn_partials_reduce_batched (no atomics, so
        // the beta/gamma gradients are bitwise reproducible)
        float* pp = partials + (size_t)blockIdx.x * 2 * (c4n * 4) + c;
        *reinterpret_cast<float4*>(pp) = tb;
        *reinterpret_cast<float4*>(pp + c4n * 4) = tg;
      }
      if (dbeta) {
        atomicAdd(dbeta + c + 0, tb.x); atomicAdd(dbeta + c + 1, tb.y);
        atomicAdd(dbeta + c + 2, tb.z); atomicAdd(dbeta + c + 3, tb.w);
      }
      if (dgamma) {
        atomicAdd(dgamma + c + 0, tg.x); atomicAdd(dgamma + c + 1, tg.y);
        atomicAdd(dgamma + c + 2, tg.z); atomicAdd(dgamma + c + 3, tg.w);
      }
    }
    __syncthreads();
  }
}

// out[j] += sum_m x[m][xoff + j]   (bias gradients).  grid (row blocks, 64-column chunks): a
// [2000][416] operand (COCO heads) is 63 x 7 workgroups instead of 8 that walk the chunks in turn.
__global__ __launch_bounds__(256) void col_sum_kernel(const float* __restrict__ x, int ldx,
                                                      int xoff, float* __restrict__ out, int M,
                                                      int ncols, int rows_per_block) {
  __shared__ float red[256];
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(M, r0 + rows_per_block);
  // 64 column lanes x 4 row lanes
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + tx;
  float s = 0.f;
  if (c < ncols)
    for (int r = r0 + ty; r < r1; r += 4) s += x[(size_t)r * ldx + xoff + c];
  red[threadIdx.x] = s;
  __syncthreads();
  if (ty == 0 && c < ncols)
    atomicAdd(out + c, red[tx] + red[tx + 64] + red[tx + 128] + red[tx + 192]);
}

// y[r][c] = mean_s x[r][s][c] * (mask ? mask[r][c] * inv_keep : 1)
template <typename T>
__global__ __launch_bounds__(256) void spatial_mean_dropout_fwd_kernel(
    const T* __restrict__ x, float* __restrict__ y, const uint8_t* __restrict__ mask,
    int rows, int spatial, int c4n, float inv_keep) {
  const long long total = (long long)rows * c4n;
  const float inv_s = 1.0f / (float)spatial;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % c4n);
    const long long r = idx / c4n;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < spatial; ++k) {
      const float4 v = c2d_ld4(x + ((size_t)r * spatial + k) * c4n * 4 + c4 * 4);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    s.x *= inv_s; s.y *= inv_s; s.z *= inv_s; s.w *= inv_s;
    if (mask) {
      const uchar4 m = *reinterpret_cast<const uchar4*>(mask + (size_t)idx * 4);
      s.x = s.x * inv_keep * (float)m.x; s.y = s.y * inv_keep * (float)m.y;
      s.z = s.z * inv_keep * (float)m.z; s.w = s.w * inv_keep * (float)m.w;
    }
    *reinterpret_cast<float4*>(y + (size_t)idx * 4) = s;
  }
}

template <typename T>
__global__ __launch_bounds__(256) void spatial_mean_dropout_bwd_kernel(
    const float* __restrict__ dy, int lddy, int dyoff, T* __restrict__ dx,
    const uint8_t* __restrict__ mask, int rows, int spatial, int c4n, float inv_keep) {
  const long long total = (long long)rows * spatial * c4n;
  const float inv_s = 1.0f / (float)spatial;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int c4 = (int)(idx % c4n);
    const long long r = idx / ((long long)c4n * spatial);
    float4 g = *reinterpret_cast<const float4*>(dy + (size_t)r * lddy + dyoff + c4 * 4);
    if (mask) {
      const uchar4 m = *reinterpret_cast<const uchar4*>(mask + ((size_t)r * c4n + c4) * 4);
      g.x = g.x * inv_keep * (float)m.x; g.y = g.y * inv_keep * (float)m.y;
      g.z = g.z * inv_keep * (float)m.z; g.w = g.w * inv_keep * (float)m.w;
    }
    g.x *= inv_s; g.y *= inv_s; g.z *= inv_s; g.w *= inv_s;
    c2d_st4(dx + (size_t)idx * 4, g);
  }
}

// Counter-based RNG (splitmix64 finaliser) -> keep mask; reproducible from (seed, offset).
__global__ __launch_bounds__(256) void dropout_mask_kernel(uint8_t* __restrict__ mask,
                                                           long long n, unsigned long long seed,
                                                           const long long* __restrict__ seed_dev,
                                                           float keep_prob) {
  if (seed_dev) seed = (unsigned long long)seed_dev[0];   // graph replays: seed lives in HBM
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(i + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z = z ^ (z >> 31);
    const float u = (float)(z >> 40) * (1.0f / 16777216.0f);  // [0,1)
    mask[i] = u < keep_prob ? 1 : 0;
  }
}

// out[p][0..2] = (2/255)*img[p][0..2] - 1, out[p][3] = 0   (FasterRCNN preprocess + pad to 4ch)
__global__ __launch_bounds__(256) void preprocess_kernel(const float* __restrict__ img,
                                                         float4* __restrict__ out,
                                                         long long pixels) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < pixels;
       i += (long long)gridDim.x * blockDim.x) {
    const float s = 2.0f / 255.0f;
    out[i] = make_float4(s * img[i * 3 + 0] - 1.0f, s * img[i * 3 + 1] - 1.0f,
                         s * img[i * 3 + 2] - 1.0f, 0.0f);
  }
}

// im2col with SAME padding for a 4-channel image: out[row][(ky*kw+kx)*4 + c], zero padded to kpad.
__global__ __launch_bounds__(256) void im2col4_kernel(const float4* __restrict__ x,
                                                      float4* __restrict__ out, int n, int ih,
                                                      int iw, int oh, int ow, int kh, int kw,
                                                      int stride, int pad_t, int pad_l,
                                                      int kpad4) {
  const long long total = (long long)n * oh * ow * kpad4;
  for (long long idx = blockIdx.x * (long long)blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const int t = (int)(idx % kpad4);
    const long long row = idx / kpad4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t < kh * kw) {
      const int ox = (int)(row % ow);
      const int oy = (int)((row / ow) % oh);
      const int img = (int)(row / ((long long)ow * oh));
      const int ky = t / kw, kx = t % kw;
      const int iy = oy * stride - pad_t + ky, ix = ox * stride - pad_l + kx;
      if (iy >= 0 && iy < ih && ix >= 0 && ix < iw) v = x[((size_t)img * ih + iy) * iw + ix];
    }
    out[idx] = v;
  }
}

// wt[t][j][i] = w[t][i][j]
__global__ __launch_bounds__(256) void transpose_taps_kernel(const float* __restrict__ w,
                                                             float* __restrict__ wt, int taps,
                                                             int I, int J) {
  __shared__ float tile[32][33];
  const int t = blockIdx.z;
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const float* src = w + (size_t)t * I * J;
  float* dst = wt + (size_t)t * I * J;
  for (int k = ty; k < 32; k += 8) {
    const int i = i0 + k, j = j0 + tx;
    tile[k][tx] = (i < I && j < J) ? src[(size_t)i * J + j] : 0.f;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int j = j0 + k, i = i0 + tx;
    if (i < I && j < J) dst[(size_t)j * I + i] = tile[tx][k];
  }
}

// g' = mult * (g + l2 * w); acc += g'^2; w -= lr * g' / sqrt(acc)       (TF ApplyAdagrad)
// One operation order for every kernel that applies the rule (the products of g' rounded
// separately, the accumulator update one fma — what adagrad_kernel has always compiled to), so that
// the one-launch forms are bitwise the per-segment launches.
__device__ __forceinline__ void adagrad_update(float& wi, float& ai, float g, float lr, float l2,
                                               float mult, float grad_scale) {
#pragma clang fp contract(off)
  const float t1 = g * grad_scale, t2 = l2 * wi;
  const float gi = mult * (t1 + t2);
  ai = __builtin_fmaf(gi, gi, ai);
  const float u = lr * gi;
  wi = wi - u / sqrtf(ai);
}

__global__ __launch_bounds__(256) void adagrad_kernel(float* __restrict__ w,
                                                      const float* __restrict__ g,
                                                      float* __restrict__ acc, long long n,
                                                      float lr, float l2, float mult,
                                                      float grad_scale) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    float wi = w[i], ai = acc[i];
    adagrad_update(wi, ai, g[i], lr, l2, mult, grad_scale);
    acc[i] = ai;
    w[i] = wi;
  }
}

// The same update over up to eight segments of the flat buffers in ONE launch (grid.y = segment;
// runs of variables that share multiplier and L2 weight: second stage / head weights / head
// biases), optionally leaving the bf16 mirror of the updated values behind (bf16 networks read
// their input-gradient operand from it: no separate cast pass over the store).
struct AdagradSegs {
  long long off[8], end[8];
  float mult[8], l2[8];
};
__global__ __launch_bounds__(256) void adagrad_multi_kernel(float* __restrict__ w,
                                                            const float* __restrict__ g,
                                                            float* __restrict__ acc,
                                                            AdagradSegs segs, float lr,
                                                            float grad_scale,
                                                            c2d_bf16* __restrict__ w16) {
  const int sgi = blockIdx.y;
  const long long end = segs.end[sgi];
  const float mult = segs.mult[sgi], l2 = segs.l2[sgi];
  for (long long i = segs.off[sgi] + blockIdx.x * (long long)blockDim.x + threadIdx.x; i < end;
       i += (long long)gridDim.x * blockDim.x) {
    float wi = w[i], ai = acc[i];
    adagrad_update(wi, ai, g[i], lr, l2, mult, grad_scale);
    acc[i] = ai;
    w[i] = wi;
    if (w16) w16[i] = (c2d_bf16)wi;
  }
}

// The general form of the optimiser step (reference branches no shipped config takes):
//   g' = m * (grad_scale*g + l2*w + l1*sign(w)),  m = mult * (col_mult ? col_mult[i % ld] : 1)
// m <= 0 freezes the element (train/trainer.py:104-125); lr may be read from device memory
// (a hipGraph then replays across a continuous learning-rate decay).
__global__ __launch_bounds__(256) void adagrad_ex_kernel(float* __restrict__ w,
                                                         const float* __restrict__ g,
                                                         float* __restrict__ acc, long long n,
                                                         float lr, const float* __restrict__ lr_dev,
                                                         float l1, float l2, float mult,
                                                         float grad_scale,
                                                         const float* __restrict__ col_mult,
                                                         int ld) {
  if (lr_dev) lr = lr_dev[0];
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const float m = col_mult ? mult * col_mult[i % ld] : mult;
    if (!(m > 0.f)) continue;
    const float wi = w[i];
    const float sg = wi > 0.f ? 1.f : (wi < 0.f ? -1.f : 0.f);
    const float gi = m * (g[i] * grad_scale + l2 * wi + l1 * sg);
    const float a = acc[i] + gi * gi;
    acc[i] = a;
    w[i] = wi - lr * gi / sqrtf(a);
  }
}

// tf.contrib.training.clip_gradient_norms (train/trainer.py:134-136): every variable's FINAL
// gradient g' (scaled, regularised, multiplied) is clipped on its OWN L2 norm:
// g' * max_norm / max(|g'|, max_norm).  One workgroup per variable (a [rows][cols] window of a
// row-major buffer with row stride ld): pass 1 writes g' in place and sums its squares in a fixed
// order (bitwise reproducible), pass 2 scales.
struct ClipDesc {
  long long offset;   // first element in the flat gradient / value buffers
  int rows, cols, ld;
  float l1, l2, mult;
};
__global__ __launch_bounds__(1024) void clip_gradient_norms_kernel(
    float* __restrict__ grads, const float* __restrict__ values,
    const ClipDesc* __restrict__ desc, float grad_scale, float max_norm) {
  const ClipDesc d = desc[blockIdx.x];
  float* g = grads + d.offset;
  const float* w = values + d.offset;
  const long long n = (long long)d.rows * d.cols;
  __shared__ float red[16];
  __shared__ float factor;
  float s = 0.f;
  for (long long i = threadIdx.x; i < n; i += blockDim.x) {
    const long long e = (i / d.cols) * d.ld + (i % d.cols);
    const float wi = w[e];
    const float sg = wi > 0.f ? 1.f : (wi < 0.f ? -1.f : 0.f);
    const float gi = d.mult * (g[e] * grad_scale + d.l2 * wi + d.l1 * sg);
    g[e] = gi;
    s += gi * gi;
  }
  s = c2d_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) t += red[k];
    const float norm = sqrtf(t);
    factor = max_norm / fmaxf(norm, max_norm);
  }
  __syncthreads();
  const float f = factor;
  if (f == 1.f) return;
  for (long long i = threadIdx.x; i < n; i += blockDim.x) {
    const long long e = (i / d.cols) * d.ld + (i % d.cols);
    g[e] *= f;
  }
}

// out[0] += weight * sum |w|   (slim l1_regularizer)
__global__ __launch_bounds__(256) void l1_loss_kernel(const float* __restrict__ w, long long n,
                                                      float weight, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)
    s += fabsf(w[i]);
  s = c2d_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, weight * (red[0] + red[1] + red[2] + red[3]));
}

// out[0] += 0.5 * weight * sum w^2
__global__ __launch_bounds__(256) void l2_loss_kernel(const float* __restrict__ w, long long n,
                                                      float weight, float* __restrict__ out) {
  __shared__ float red[4];
  float s = 0.f;
  const long long n4 = (reinterpret_cast<unsigned long long>(w) & 15) == 0 ? (n >> 2) : 0;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4;
       i += (long long)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(w)[i];
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  for (long long i = (n4 << 2) + blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x)            // (tail, or an unaligned variable)
    s += w[i] * w[i];
  s = c2d_wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, 0.5f * weight * (red[0] + red[1] + red[2] + red[3]));
}

// scale[c] = gamma[c] * rsqrt(var[c] + eps); shift[c] = beta[c] - mean[c] * scale[c]
__global__ void bn_fold_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ mean, const float* __restrict__ var,
                               float eps, float* __restrict__ scale, float* __restrict__ shift,
                               int c) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= c) return;
  const float s = (gamma ? gamma[i] : 1.0f) / sqrtf(var[i] + eps);
  scale[i] = s;
  shift[i] = beta[i] - mean[i] * s;
}

// Batched "refresh" of the per-layer kernel operands after an optimiser step: one launch
// transposes the weights of every trainable layer, one folds every BatchNorm (instead of two
// 4-us launches per layer).  Descriptors live in device memory (built once by the caller):
//   transpose: {src_off, dst_off, taps, rows, cols, tile_begin} in floats / 32x32 tiles
//   bn_fold  : {gamma_off (or -1), beta_off, mean_off, var_off, scale_off, shift_off, c, begin}
struct TransDesc { long long src_off, dst_off; int taps, rows, cols, tile_begin; };
struct FoldDesc { long long gamma, beta, mean, var, scale, shift; int c, begin; };

__global__ __launch_bounds__(256) void transpose_taps_batched_kernel(
    const TransDesc* __restrict__ desc, int num, const float* __restrict__ src_base,
    float* __restrict__ dst_base, c2d_bf16* __restrict__ dst16_base) {
  __shared__ float tile[32][33];
  int lo = 0, hi = num - 1;          // last descriptor with tile_begin <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (desc[mid].tile_begin <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const TransDesc d = desc[lo];
  int t = blockIdx.x - d.tile_begin;
  const int tj = (d.cols + 31) / 32, ti = (d.rows + 31) / 32;
  const int tap = t / (ti * tj);
  t -= tap * ti * tj;
  const int i0 = (t / tj) * 32, j0 = (t % tj) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float* src = src_base + d.src_off + (size_t)tap * d.rows * d.cols;
  float* dst = dst_base + d.dst_off + (size_t)tap * d.rows * d.cols;
  for (int k = ty; k < 32; k += 8) {
    const int i = i0 + k, j = j0 + tx;
    tile[k][tx] = (i < d.rows && j < d.cols) ? src[(size_t)i * d.cols + j] : 0.f;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int j = j0 + k, i = i0 + tx;
    if (i < d.rows && j < d.cols) {
      dst[(size_t)j * d.rows + i] = tile[tx][k];
      if (dst16_base)      // the bf16 mirror of the derived operand (same element offsets)
        dst16_base[d.dst_off + (size_t)tap * d.rows * d.cols + (size_t)j * d.rows + i] =
            (c2d_bf16)tile[tx][k];
    }
  }
}

__global__ __launch_bounds__(256) void bn_fold_batched_kernel(
    const FoldDesc* __restrict__ desc, int num, int total, const float* __restrict__ vars,
    const float* __restrict__ stats, float eps, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  int lo = 0, hi = num - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (desc[mid].begin <= i) lo = mid; else hi = mid - 1;
  }
  const FoldDesc d = desc[lo];
  const int c = i - d.begin;
  const float g = d.gamma >= 0 ? vars[d.gamma + c] : 1.0f;
  const float s = g / sqrtf(stats[d.var + c] + eps);
  out[d.scale + c] = s;
  out[d.shift + c] = vars[d.beta + c] - stats[d.mean + c] * s;
}

// Sums the per-block partial sums of bn_relu_bwd (partial form) into the flat gradient buffer.
// One workgroup per (layer, 64-channel chunk): 16 float4 channel lanes x 16 block lanes.
// wide: 0, or the width W of the partial rows when the layer's c columns are a slice of rows
// [2][W] written for a whole concat buffer (c2d_conv1x1_dgrad_multi_bn_relu): ws_off then points
// at the layer's first column of block 0's beta half.
struct BnPartDesc { long long ws_off, dbeta_off, dgamma_off; int nblocks, c, begin, wide; };

constexpr int BNR_TY = 64;      // row lanes per 64-channel chunk (64 instead of 16: four times the loads in flight per chunk)
__global__ __launch_bounds__(16 * BNR_TY) void bn_partials_reduce_kernel(
    const BnPartDesc* __restrict__ desc, int num, const float* __restrict__ ws,
    float* __restrict__ grads) {
  __shared__ float4 red[2][BNR_TY][16];
  int lo = 0, hi = num - 1;          // last descriptor with begin <= blockIdx.x
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (desc[mid].begin <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const BnPartDesc d = desc[lo];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int c = ((blockIdx.x - d.begin) * 16 + tx) * 4;
  const bool active = c < d.c;
  float4 sb = make_float4(0.f, 0.f, 0.f, 0.f), sg = sb;
  if (active) {
    const float* base = ws + d.ws_off + c;
    const int half = d.wide > 0 ? d.wide : d.c;
    const size_t bstride = (size_t)2 * half;
    for (int b = ty; b < d.nblocks; b += BNR_TY) {
      const float4 vb = *reinterpret_cast<const float4*>(base + b * bstride);
      const float4 vg = *reinterpret_cast<const float4*>(base + b * bstride + half);
      sb.x += vb.x; sb.y += vb.y; sb.z += vb.z; sb.w += vb.w;
      sg.x += vg.x; sg.y += vg.y; sg.z += vg.z; sg.w += vg.w;
    }
  }
  red[0][ty][tx] = sb;
  red[1][ty][tx] = sg;
  __syncthreads();
  if (ty < 2 && active) {           // ty 0: beta, ty 1: gamma
    const long long off = ty == 0 ? d.dbeta_off : d.dgamma_off;
    if (off >= 0) {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int k = 0; k < BNR_TY; ++k) {
        const float4 v = red[ty][k][tx];
        t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
      }
      float* g = grads + off + c;
      g[0] += t.x; g[1] += t.y; g[2] += t.z; g[3] += t.w;
    }
  }
}

inline int bn_rows_per_block(int rows) {
  // ~1024 row blocks (4 per CU) with at least 32 rows each, a multiple of 32 rows
  int rpb = (int)(((long long)rows + 1023) / 1024);
  rpb = (rpb + 31) / 32 * 32;
  return rpb < 32 ? 32 : rpb;
}

inline int grid_for(long long total) {
  long long b = (total + 255) / 256;
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (int)b;
}

template <typename T>
int launch_bn_relu_bwd(const T* dy, int lddy, int dyoff, const T* y, int ldy, int yoff,
                       const float* scale, const float* beta, const float* gamma, T* dc,
                       float* dbeta, float* dgamma, float* partials, int rows, int c,
                       int rows_per_block, hipStream_t s, const HeadGrad* head = nullptr) {
  const int c4n = c / 4;
  const int blocks = c2d_ceil_div(rows, rows_per_block);
  const HeadGrad hg = head ? *head : HeadGrad{nullptr, 0, 0, nullptr, 0, 0, 1, 1.0f, 0};
#define K_BNB(TX)                                                                          \
  {                                                                                          \
    if (head && head->dmean)                                                                 \
      hipLaunchKernelGGL((bn_relu_bwd_kernel<TX, T, true>), dim3(blocks), dim3(256), 0, s, dy, lddy, \
                         dyoff, y, ldy, yoff, scale, beta, gamma, dc, dbeta, dgamma, partials, rows, \
                         c4n, rows_per_block, hg);                                           \
    else                                                                                     \
      hipLaunchKernelGGL((bn_relu_bwd_kernel<TX, T, false>), dim3(blocks), dim3(256), 0, s, dy, lddy, \
                         dyoff, y, ldy, yoff, scale, beta, gamma, dc, dbeta, dgamma, partials, rows, \
                         c4n, rows_per_block, hg);                                           \
  }
  if (c4n <= 16) K_BNB(16)
  else if (c4n <= 32) K_BNB(32)
  else if (c4n <= 64) K_BNB(64)
  else K_BNB(128)
#undef K_BNB
  return c2d_launch_status();
}

}  // namespace

template <typename T>
int pool3x3_fwd_impl(const T* x, int ldx, int xoff, T* y, int ldy, int yoff, uint8_t* argmax, int n,
                     int ih, int iw, int c, int stride, int mode, void* stream, int relu = 0) {
  C2D_CHECK_ARG(x && y && n > 0 && ih > 0 && iw > 0 && c > 0 && c % 4 == 0);
  C2D_CHECK_ARG((stride == 1 || stride == 2) && (mode == 0 || mode == 1));
  C2D_CHECK_ARG(ldx % 4 == 0 && xoff % 4 == 0 && ldy % 4 == 0 && yoff % 4 == 0);
  const PoolGeom g = make_pool_geom(ih, iw, stride);
  const long long total = (long long)n * g.oh * g.ow * (c / 4);
  if (!relu && n >= 64 && ih == iw && ((ih == 4 && stride == 1) || (ih == 7 && stride == 2 && mode == 0))) {
    // per-ROI maps of the second stage: whole-map kernel, every input element fetched once
    const dim3 grid(grid_for((long long)n * (c / 4))), block(256);
    hipStream_t st = (hipStream_t)stream;
#define K_POOL_F(IH, S, MD)                                                                    \
  hipLaunchKernelGGL((pool3x3_map_fwd_kernel<IH, S, MD, T>), grid, block, 0, st, x, ldx, xoff, y, \
                     ldy, yoff, argmax, n, c / 4)
    if (ih == 4 && mode == 0) K_POOL_F(4, 1, 0);
    else if (ih == 4) K_POOL_F(4, 1, 1);
    else K_POOL_F(7, 2, 0);
#undef K_POOL_F
    return c2d_launch_status();
  }
  hipLaunchKernelGGL(pool3x3_fwd_kernel<T>, dim3(grid_for(total)), dim3(256), 0,
                     (hipStream_t)stream, x, ldx, xoff, y, ldy, yoff, argmax, n, c / 4, g, mode, relu);
  return c2d_launch_status();
}

template <typename T>
int pool3x3_bwd_impl(const T* dy, int lddy, int dyoff, const uint8_t* argmax, T* dx, int lddx,
                     int dxoff, int n, int ih, int iw, int c, int stride, int mode, int accumulate,
                     void* stream, const T* ymask = nullptr, int ldym = 0, int ymoff = 0) {
  C2D_CHECK_ARG(dy && dx && n > 0 && ih > 0 && iw > 0 && c > 0 && c % 4 == 0);
  C2D_CHECK_ARG((stride == 1 || stride == 2) && (mode == 0 || mode == 1));
  C2D_CHECK_ARG(mode == 1 || argmax);
  C2D_CHECK_ARG(lddx % 4 == 0 && dxoff % 4 == 0 && lddy % 4 == 0 && dyoff % 4 == 0);
  const PoolGeom g = make_pool_geom(ih, iw, stride);
  const long long total = (long long)n * ih * iw * (c / 4);
  if (n >= 64 && ih == iw && (!ymask || mode == 1) &&
      ((ih == 4 && stride == 1) || (ih == 7 && stride == 2 && mode == 0))) {
    const dim3 grid(grid_for((long long)n * (c / 4))), block(256);
    hipStream_t st = (hipStream_t)stream;
#define K_POOL_B(IH, S, MD, YM)                                                                 \
  hipLaunchKernelGGL((pool3x3_map_bwd_kernel<IH, S, MD, T, YM>), grid, block, 0, st, dy, lddy,     \
                     dyoff, argmax, dx, lddx, dxoff, n, c / 4, accumulate, ymask, ldym, ymoff)
    if (ih == 4 && mode == 0) K_POOL_B(4, 1, 0, false);
    else if (ih == 4 && ymask) K_POOL_B(4, 1, 1, true);
    else if (ih == 4) K_POOL_B(4, 1, 1, false);
    else K_POOL_B(7, 2, 0, false);
#undef K_POOL_B
    return c2d_launch_status();
  }
  hipLaunchKernelGGL(pool3x3_bwd_kernel<T>, dim3(grid_for(total)), dim3(256), 0,
                     (hipStream_t)stream, dy, lddy, dyoff, argmax, dx, lddx, dxoff, n, c / 4, g,
                     mode, accumulate, ymask, ldym, ymoff);
  return c2d_launch_status();
}

extern "C" int c2d_pool3x3_fwd(const float* x, int ldx, int xoff, float* y, int ldy, int yoff,
                               uint8_t* argmax, int n, int ih, int iw, int c, int stride,
                               int mode, void* stream) {
  return pool3x3_fwd_impl<float>(x, ldx, xoff, y, ldy, yoff, argmax, n, ih, iw, c, stride, mode,
                                 stream);
}

extern "C" int c2d_pool3x3_fwd_bf16(const void* x, int ldx, int xoff, void* y, int ldy, int yoff,
                                    uint8_t* argmax, int n, int ih, int iw, int c, int stride,
                                    int mode, void* stream) {
  return pool3x3_fwd_impl<c2d_bf16>((const c2d_bf16*)x, ldx, xoff, (c2d_bf16*)y, ldy, yoff, argmax,
                                    n, ih, iw, c, stride, mode, stream);
}

extern "C" int c2d_pool3x3_bwd(const float* dy, int lddy, int dyoff, const uint8_t* argmax,
                               float* dx, int lddx, int dxoff, int n, int ih, int iw, int c,
                               int stride, int mode, int accumulate, void* stream) {
  return pool3x3_bwd_impl<float>(dy, lddy, dyoff, argmax, dx, lddx, dxoff, n, ih, iw, c, stride,
                                 mode, accumulate, stream);
}

extern "C" int c2d_pool3x3_bwd_bf16(const void* dy, int lddy, int dyoff, const uint8_t* argmax,
                                    void* dx, int lddx, int dxoff, int n, int ih, int iw, int c,
                                    int stride, int mode, int accumulate, void* stream) {
  return pool3x3_bwd_impl<c2d_bf16>((const c2d_bf16*)dy, lddy, dyoff, argmax, (c2d_bf16*)dx, lddx,
                                    dxoff, n, ih, iw, c, stride, mode, accumulate, stream);
}

// relu(avg_pool3x3(x)) and its gradient dx (+)= avg_pool3x3_bwd(dy * (y > 0)): the two halves of
// an average-pooling branch commuted behind its 1x1 convolution (include/cap2det_hip.h).
extern "C" int c2d_avgpool3x3_relu_fwd(const float* x, int ldx, int xoff, float* y, int ldy,
                                       int yoff, int n, int ih, int iw, int c, int stride,
                                       void* stream) {
  return pool3x3_fwd_impl<float>(x, ldx, xoff, y, ldy, yoff, nullptr, n, ih, iw, c, stride, 1, stream, 1);
}
extern "C" int c2d_avgpool3x3_relu_fwd_bf16(const void* x, int ldx, int xoff, void* y, int ldy,
                                            int yoff, int n, int ih, int iw, int c, int stride,
                                            void* stream) {
  return pool3x3_fwd_impl<c2d_bf16>((const c2d_bf16*)x, ldx, xoff, (c2d_bf16*)y, ldy, yoff, nullptr,
                                    n, ih, iw, c, stride, 1, stream, 1);
}
extern "C" int c2d_avgpool3x3_relu_bwd(const float* dy, int lddy, int dyoff, const float* y,
                                       int ldy, int yoff, float* dx, int lddx, int dxoff, int n,
                                       int ih, int iw, int c, int stride, int accumulate,
                                       void* stream) {
  C2D_CHECK_ARG(y && ldy % 4 == 0 && yoff % 4 == 0);
  return pool3x3_bwd_impl<float>(dy, lddy, dyoff, nullptr, dx, lddx, dxoff, n, ih, iw, c, stride, 1,
                                 accumulate, stream, y, ldy, yoff);
}
extern "C" int c2d_avgpool3x3_relu_bwd_bf16(const void* dy, int lddy, int dyoff, const void* y,
                                            int ldy, int yoff, void* dx, int lddx, int dxoff,
                                            int n, int ih, int iw, int c, int stride,
                                            int accumulate, void* stream) {
  C2D_CHECK_ARG(y && ldy % 4 == 0 && yoff % 4 == 0);
  return pool3x3_bwd_impl<c2d_bf16>((const c2d_bf16*)dy, lddy, dyoff, nullptr, (c2d_bf16*)dx, lddx,
                                    dxoff, n, ih, iw, c, stride, 1, accumulate, stream,
                                    (const c2d_bf16*)y, ldy, yoff);
}

extern "C" int c2d_bn_relu_bwd(const float* dy, int lddy, int dyoff, const float* y, int ldy,
                               int yoff, const float* scale, const float* beta,
                               const float* gamma, float* dc, float* dbeta, float* dgamma,
                               int rows, int c, void* stream) {
  C2D_CHECK_ARG(dy && y && scale && dc && rows > 0 && c > 0 && c % 4 == 0);
  C2D_CHECK_ARG(!dgamma || (beta && gamma));
  C2D_CHECK_ARG(lddy % 4 == 0 && dyoff % 4 == 0 && ldy % 4 == 0 && yoff % 4 == 0);
  return launch_bn_relu_bwd(dy, lddy, dyoff, y, ldy, yoff, scale, beta, gamma, dc, dbeta, dgamma,
                            nullptr, rows, c, 128, (hipStream_t)stream);
}

extern "C" int c2d_bn_relu_bwd_partial_blocks(int rows, int c) {
  (void)c;
  return rows > 0 ? c2d_ceil_div(rows, bn_rows_per_block(rows)) : 0;
}

extern "C" int c2d_bn_relu_bwd_partial(const float* dy, int lddy, int dyoff, const float* y,
                                       int ldy, int yoff, const float* scale, const float* beta,
                                       const float* gamma, float* dc, float* partials, int rows,
                                       int c, void* stream) {
  C2D_CHECK_ARG(dy && y && scale && dc && partials && rows > 0 && c > 0 && c % 4 == 0);
  C2D_CHECK_ARG(!gamma || beta);
  C2D_CHECK_ARG(lddy % 4 == 0 && dyoff % 4 == 0 && ldy % 4 == 0 && yoff % 4 == 0);
  return launch_bn_relu_bwd(dy, lddy, dyoff, y, ldy, yoff, scale, beta, gamma, dc, nullptr,
                            nullptr, partials, rows, c, bn_rows_per_block(rows),
                            (hipStream_t)stream);
}

extern "C" int c2d_bn_relu_bwd_partial_bf16(const void* dy, int lddy, int dyoff, const void* y,
                                            int ldy, int yoff, const float* scale,
                                            const float* beta, const float* gamma, void* dc,
                                            float* partials, int rows, int c, void* stream) {
  C2D_CHECK_ARG(dy && y && scale && dc && partials && rows > 0 && c > 0 && c % 4 == 0);
  C2D_CHECK_ARG(!gamma || beta);
  C2D_CHECK_ARG(lddy % 4 == 0 && dyoff % 4 == 0 && ldy % 4 == 0 && yoff % 4 == 0);
  return launch_bn_relu_bwd<c2d_bf16>((const c2d_bf16*)dy, lddy, dyoff, (const c2d_bf16*)y, ldy,
                                      yoff, scale, beta, gamma, (c2d_bf16*)dc, nullptr, nullptr,
                                      partials, rows, c, bn_rows_per_block(rows),
                                      (hipStream_t)stream);
}

// Head form: the gradient of the spatially averaged features instead of a dy buffer (HeadGrad).
template <typename T>
static int bn_relu_bwd_partial_head_impl(const float* dmean, int ldd, int doff, const uint8_t* mask,
                                         int mask_ld, int mask_off, int spatial, float keep_prob,
                                         const T* y, int ldy, int yoff, const float* scale,
                                         const float* beta, const float* gamma, T* dc,
                                         float* partials, int rows, int c, void* stream) {
  C2D_CHECK_ARG(dmean && y && scale && beta && dc && partials && rows > 0 && c > 0 && c % 4 == 0);
  C2D_CHECK_ARG(ldd % 4 == 0 && doff % 4 == 0 && ldy % 4 == 0 && yoff % 4 == 0 && spatial > 0 &&
                rows % spatial == 0 && keep_prob > 0.f && (!mask || (mask_ld % 4 == 0 && mask_off % 4 == 0)));
  const HeadGrad hg = {dmean, ldd, doff, mask, mask_ld, mask_off, spatial,
                       (mask ? 1.0f / keep_prob : 1.0f) / (float)spatial, 0};
  return launch_bn_relu_bwd<T>(nullptr, 0, 0, y, ldy, yoff, scale, beta, gamma, dc, nullptr, nullptr,
                               partials, rows, c, bn_rows_per_block(rows), (hipStream_t)stream, &hg);
}

extern "C" int c2d_bn_relu_bwd_partial_head(const float* dmean, int ldd, int doff,
                                            const uint8_t* mask, int mask_ld, int mask_off,
                                            int spatial, float keep_prob, const float* y, int ldy,
                                            int yoff, const float* scale, const float* beta,
                                            const float* gamma, float* dc, float* partials,
                                            int rows, int c, void* stream) {
  return bn_relu_bwd_partial_head_impl<float>(dmean, ldd, doff, mask, mask_ld, mask_off, spatial,
                                              keep_prob, y, ldy, yoff, scale, beta, gamma, dc,
                                              partials, rows, c, stream);
}

extern "C" int c2d_bn_relu_bwd_partial_head_bf16(const float* dmean, int ldd, int doff,
                                                 const uint8_t* mask, int mask_ld, int mask_off,
                                                 int spatial, float keep_prob, const void* y,
                                                 int ldy, int yoff, const float* scale,
                                                 const float* beta, const float* gamma, void* dc,
                                                 float* partials, int rows, int c, void* stream) {
  return bn_relu_bwd_partial_head_impl<c2d_bf16>(dmean, ldd, doff, mask, mask_ld, mask_off, spatial,
                                                 keep_prob, (const c2d_bf16*)y, ldy, yoff, scale,
                                                 beta, gamma, (c2d_bf16*)dc, partials, rows, c,
                                                 stream);
}

// BatchNorm backward of a layer WITHOUT a ReLU behind it (the 1x1 convolution of a commuted
// average-pooling branch): dc = dy * scale, sums of dy and dy * (y - beta) / gamma.
template <typename T>
static int bn_bwd_partial_impl(const T* dy, int lddy, int dyoff, const T* y, int ldy, int yoff,
                               const float* scale, const float* beta, const float* gamma, T* dc,
                               float* partials, int rows, int c, void* stream) {
  C2D_CHECK_ARG(dy && y && scale && dc && partials && rows > 0 && c > 0 && c % 4 == 0);
  C2D_CHECK_ARG(!gamma || beta);
  C2D_CHECK_ARG(lddy % 4 == 0 && dyoff % 4 == 0 && ldy % 4 == 0 && yoff % 4 == 0);
  const HeadGrad hg = {nullptr, 0, 0, nullptr, 0, 0, 1, 1.0f, 1};
  return launch_bn_relu_bwd<T>(dy, lddy, dyoff, y, ldy, yoff, scale, beta, gamma, dc, nullptr, nullptr,
                               partials, rows, c, bn_rows_per_block(rows), (hipStream_t)stream, &hg);
}
extern "C" int c2d_bn_bwd_partial(const float* dy, int lddy, int dyoff, const float* y, int ldy,
                                  int yoff, const float* scale, const float* beta,
                                  const float* gamma, float* dc, float* partials, int rows, int c,
                                  void* stream) {
  return bn_bwd_partial_impl<float>(dy, lddy, dyoff, y, ldy, yoff, scale, beta, gamma, dc, partials,
                                    rows, c, stream);
}
extern "C" int c2d_bn_bwd_partial_bf16(const void* dy, int lddy, int dyoff, const void* y, int ldy,
                                       int yoff, const float* scale, const float* beta,
                                       const float* gamma, void* dc, float* partials, int rows,
                                       int c, void* stream) {
  return bn_bwd_partial_impl<c2d_bf16>((const c2d_bf16*)dy, lddy, dyoff, (const c2d_bf16*)y, ldy, yoff,
                                       scale, beta, gamma, (c2d_bf16*)dc, partials, rows, c, stream);
}

extern "C" int c2d_bn_partials_reduce_batched(const void* desc, int num, int total_chunks,
                                              const float* ws, float* grads, void* stream) {
  C2D_CHECK_ARG(desc && ws && grads && num > 0 && total_chunks > 0);
  hipLaunchKernelGGL(bn_partials_reduce_kernel, dim3(total_chunks), dim3(16 * BNR_TY), 0,
                     (hipStream_t)stream, (const BnPartDesc*)desc, num, ws, grads);
  return c2d_launch_status();
}

extern "C" int c2d_col_sum(const float* x, int ldx, int xoff, float* out, int rows, int ncols,
                           void* stream) {
  C2D_CHECK_ARG(x && out && rows > 0 && ncols > 0);
  const int rows_per_block = 32;
  hipLaunchKernelGGL(col_sum_kernel,
                     dim3(c2d_ceil_div(rows, rows_per_block), c2d_ceil_div(ncols, 64)), dim3(256),
                     0, (hipStream_t)stream, x, ldx, xoff, out, rows, ncols, rows_per_block);
  return c2d_launch_status();
}

extern "C" int c2d_spatial_mean_dropout_fwd(const float* x, float* y, const uint8_t* mask,
                                            int rows, int spatial, int c, float keep_prob,
                                            void* stream) {
  C2D_CHECK_ARG(x && y && rows > 0 && spatial > 0 && c > 0 && c % 4 == 0 && keep_prob > 0.f);
  hipLaunchKernelGGL(spatial_mean_dropout_fwd_kernel<float>,
                     dim3(grid_for((long long)rows * c / 4)), dim3(256), 0, (hipStream_t)stream, x,
                     y, mask, rows, spatial, c / 4, 1.0f / keep_prob);
  return c2d_launch_status();
}

extern "C" int c2d_spatial_mean_dropout_fwd_bf16(const void* x, float* y, const uint8_t* mask,
                                                 int rows, int spatial, int c, float keep_prob,
                                                 void* stream) {
  C2D_CHECK_ARG(x && y && rows > 0 && spatial > 0 && c > 0 && c % 4 == 0 && keep_prob > 0.f);
  hipLaunchKernelGGL(spatial_mean_dropout_fwd_kernel<c2d_bf16>,
                     dim3(grid_for((long long)rows * c / 4)), dim3(256), 0, (hipStream_t)stream,
                     (const c2d_bf16*)x, y, mask, rows, spatial, c / 4, 1.0f / keep_prob);
  return c2d_launch_status();
}

extern "C" int c2d_spatial_mean_dropout_bwd(const float* dy, int lddy, int dyoff, float* dx,
                                            const uint8_t* mask, int rows, int spatial, int c,
                                            float keep_prob, void* stream) {
  C2D_CHECK_ARG(dy && dx && rows > 0 && spatial > 0 && c > 0 && c % 4 == 0 && keep_prob > 0.f);
  C2D_CHECK_ARG(lddy % 4 == 0 && dyoff % 4 == 0);
  hipLaunchKernelGGL(spatial_mean_dropout_bwd_kernel<float>,
                     dim3(grid_for((long long)rows * spatial * c / 4)), dim3(256), 0,
                     (hipStream_t)stream, dy, lddy, dyoff, dx, mask, rows, spatial, c / 4,
                     1.0f / keep_prob);
  return c2d_launch_status();
}

extern "C" int c2d_spatial_mean_dropout_bwd_bf16(const float* dy, int lddy, int dyoff, void* dx,
                                                 const uint8_t* mask, int rows, int spatial, int c,
                                                 float keep_prob, void* stream) {
  C2D_CHECK_ARG(dy && dx && rows > 0 && spatial > 0 && c > 0 && c % 4 == 0 && keep_prob > 0.f);
  C2D_CHECK_ARG(lddy % 4 == 0 && dyoff % 4 == 0);
  hipLaunchKernelGGL(spatial_mean_dropout_bwd_kernel<c2d_bf16>,
                     dim3(grid_for((long long)rows * spatial * c / 4)), dim3(256), 0,
                     (hipStream_t)stream, dy, lddy, dyoff, (c2d_bf16*)dx, mask, rows, spatial, c / 4,
                     1.0f / keep_prob);
  return c2d_launch_status();
}

extern "C" int c2d_dropout_mask(uint8_t* mask, long long n, unsigned long long seed,
                                float keep_prob, void* stream) {
  C2D_CHECK_ARG(mask && n >= 0);
  if (n == 0) return C2D_OK;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream,
                     mask, n, seed, (const long long*)nullptr, keep_prob);
  return c2d_launch_status();
}

extern "C" int c2d_dropout_mask_dev(uint8_t* mask, long long n, const long long* seed_dev,
                                    float keep_prob, void* stream) {
  C2D_CHECK_ARG(mask && seed_dev && n >= 0);
  if (n == 0) return C2D_OK;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream,
                     mask, n, 0ull, seed_dev, keep_prob);
  return c2d_launch_status();
}

extern "C" int c2d_preprocess_pad4(const float* image, float* out, long long pixels,
                                   void* stream) {
  C2D_CHECK_ARG(image && out && pixels > 0);
  hipLaunchKernelGGL(preprocess_kernel, dim3(grid_for(pixels)), dim3(256), 0,
                     (hipStream_t)stream, image, (float4*)out, pixels);
  return c2d_launch_status();
}

extern "C" int c2d_im2col4(const float* x, float* out, int n, int ih, int iw, int kh, int kw,
                           int stride, int kpad, void* stream) {
  C2D_CHECK_ARG(x && out && n > 0 && ih > 0 && iw > 0 && kh > 0 && kw > 0 && stride > 0);
  C2D_CHECK_ARG(kpad % 4 == 0 && kpad >= kh * kw * 4);
  const int oh = (ih + stride - 1) / stride, ow = (iw + stride - 1) / stride;
  const int pth = (oh - 1) * stride + kh - ih, ptw = (ow - 1) * stride + kw - iw;
  const int pad_t = (pth > 0 ? pth : 0) / 2, pad_l = (ptw > 0 ? ptw : 0) / 2;
  const long long total = (long long)n * oh * ow * (kpad / 4);
  hipLaunchKernelGGL(im2col4_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)x, (float4*)out, n, ih, iw, oh, ow, kh, kw, stride, pad_t,
                     pad_l, kpad / 4);
  return c2d_launch_status();
}

extern "C" int c2d_transpose_taps(const float* w, float* wt, int taps, int rows, int cols,
                                  void* stream) {
  C2D_CHECK_ARG(w && wt && taps > 0 && rows > 0 && cols > 0);
  dim3 grid(c2d_ceil_div(cols, 32), c2d_ceil_div(rows, 32), taps);
  hipLaunchKernelGGL(transpose_taps_kernel, grid, dim3(256), 0, (hipStream_t)stream, w, wt,
                     taps, rows, cols);
  return c2d_launch_status();
}

extern "C" int c2d_adagrad_step(float* w, const float* g, float* acc, long long n, float lr,
                                float l2, float mult, float grad_scale, void* stream) {
  C2D_CHECK_ARG(w && g && acc && n >= 0);
  if (n == 0) return C2D_OK;
  hipLaunchKernelGGL(adagrad_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, w, g,
                     acc, n, lr, l2, mult, grad_scale);
  return c2d_launch_status();
}

extern "C" int c2d_adagrad_step_ex(float* w, const float* g, float* acc, long long n, float lr,
                                   const float* lr_dev, float l1, float l2, float mult,
                                   float grad_scale, const float* col_mult, int ld,
                                   void* stream) {
  C2D_CHECK_ARG(w && g && acc && n >= 0 && (!col_mult || ld > 0));
  if (n == 0) return C2D_OK;
  hipLaunchKernelGGL(adagrad_ex_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, w,
                     g, acc, n, lr, lr_dev, l1, l2, mult, grad_scale, col_mult, ld);
  return c2d_launch_status();
}

extern "C" int c2d_adagrad_step_multi(float* values, const float* grads, float* accum,
                                      int num_segments, const long long* offsets,
                                      const long long* ends, const float* mults,
                                      const float* l2s, float lr, float grad_scale,
                                      void* values_bf16, void* stream) {
  C2D_CHECK_ARG(values && grads && accum && offsets && ends && mults && l2s);
  C2D_CHECK_ARG(num_segments > 0 && num_segments <= 8);
  AdagradSegs segs;
  long long longest = 0;
  for (int i = 0; i < 8; ++i) {
    const bool on = i < num_segments;
    C2D_CHECK_ARG(!on || (offsets[i] >= 0 && ends[i] >= offsets[i]));
    segs.off[i] = on ? offsets[i] : 0;
    segs.end[i] = on ? ends[i] : 0;
    segs.mult[i] = on ? mults[i] : 0.f;
    segs.l2[i] = on ? l2s[i] : 0.f;
    if (on && ends[i] - offsets[i] > longest) longest = ends[i] - offsets[i];
  }
  if (longest == 0) return C2D_OK;
  hipLaunchKernelGGL(adagrad_multi_kernel, dim3(grid_for(longest), num_segments), dim3(256), 0,
                     (hipStream_t)stream, values, grads, accum, segs, lr, grad_scale,
                     (c2d_bf16*)values_bf16);
  return c2d_launch_status();
}

static_assert(sizeof(ClipDesc) == sizeof(C2dClipDesc), "C2dClipDesc layout");
extern "C" int c2d_clip_gradient_norms(float* grads, const float* values,
                                       const C2dClipDesc* desc, int num, float grad_scale,
                                       float max_norm, void* stream) {
  C2D_CHECK_ARG(grads && values && desc && num >= 0 && max_norm > 0.f);
  if (num == 0) return C2D_OK;
  hipLaunchKernelGGL(clip_gradient_norms_kernel, dim3(num), dim3(1024), 0, (hipStream_t)stream,
                     grads, values, reinterpret_cast<const ClipDesc*>(desc), grad_scale,
                     max_norm);
  return c2d_launch_status();
}

extern "C" int c2d_l1_loss(const float* w, long long n, float weight, float* out,
                           void* stream) {
  C2D_CHECK_ARG(w && out && n >= 0);
  if (n == 0) return C2D_OK;
  long long b = (n + 255) / 256;
  if (b > 1024) b = 1024;
  hipLaunchKernelGGL(l1_loss_kernel, dim3((int)b), dim3(256), 0, (hipStream_t)stream, w, n,
                     weight, out);
  return c2d_launch_status();
}

extern "C" int c2d_l2_loss(const float* w, long long n, float weight, float* out,
                           void* stream) {
  C2D_CHECK_ARG(w && out && n >= 0);
  if (n == 0) return C2D_OK;
  // (every block ends in one atomic on out[0]: 128 blocks of 16-byte loads, not 1024 — on the 1.7 MB
  // of head weights)
  long long b = (n + 4095) / 4096;
  if (b > 128) b = 128;
  hipLaunchKernelGGL(l2_loss_kernel, dim3((int)b), dim3(256), 0, (hipStream_t)stream, w, n,
                     weight, out);
  return c2d_launch_status();
}

extern "C" int c2d_bn_fold(const float* gamma, const float* beta, const float* mean,
                           const float* var, float eps, float* scale, float* shift, int c,
                           void* stream) {
  C2D_CHECK_ARG(beta && mean && var && scale && shift && c > 0);
  hipLaunchKernelGGL(bn_fold_kernel, dim3(c2d_ceil_div(c, 256)), dim3(256), 0,
                     (hipStream_t)stream, gamma, beta, mean, var, eps, scale, shift, c);
  return c2d_launch_status();
}

extern "C" int c2d_transpose_taps_batched(const void* desc, int num, int total_tiles,
                                          const float* src_base, float* dst_base,
                                          void* stream) {
  C2D_CHECK_ARG(desc && src_base && dst_base && num > 0 && total_tiles > 0);
  hipLaunchKernelGGL(transpose_taps_batched_kernel, dim3(total_tiles), dim3(256), 0,
                     (hipStream_t)stream, (const TransDesc*)desc, num, src_base, dst_base,
                     (c2d_bf16*)nullptr);
  return c2d_launch_status();
}

extern "C" int c2d_transpose_taps_batched_mirror(const void* desc, int num, int total_tiles,
                                                 const float* src_base, float* dst_base,
                                                 void* dst_bf16, void* stream) {
  C2D_CHECK_ARG(desc && src_base && dst_base && dst_bf16 && num > 0 && total_tiles > 0);
  hipLaunchKernelGGL(transpose_taps_batched_kernel, dim3(total_tiles), dim3(256), 0,
                     (hipStream_t)stream, (const TransDesc*)desc, num, src_base, dst_base,
                     (c2d_bf16*)dst_bf16);
  return c2d_launch_status();
}

extern "C" int c2d_bn_fold_batched(const void* desc, int num, int total_channels,
                                   const float* vars, const float* stats, float eps, float* out,
                                   void* stream) {
  C2D_CHECK_ARG(desc && vars && stats && out && num > 0 && total_channels > 0);
  hipLaunchKernelGGL(bn_fold_batched_kernel, dim3(c2d_ceil_div(total_channels, 256)), dim3(256),
                     0, (hipStream_t)stream, (const FoldDesc*)desc, num, total_channels, vars,
                     stats, eps, out);
  return c2d_launch_status();
}

namespace {
__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src,
                                                        c2d_bf16* __restrict__ dst, long long n4) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4;
       i += (long long)gridDim.x * blockDim.x)
    c2d_st4(dst + i * 4, c2d_ld4(src + i * 4));
}
}  // namespace

namespace {
__global__ __launch_bounds__(256) void cast_f32_kernel(const c2d_bf16* __restrict__ src,
                                                       float* __restrict__ dst, long long n4) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4;
       i += (long long)gridDim.x * blockDim.x)
    c2d_st4(dst + i * 4, c2d_ld4(src + i * 4));
}
}  // namespace

extern "C" int c2d_cast_f32(const void* src, float* dst, long long n, void* stream) {
  C2D_CHECK_ARG(src && dst && n >= 0 && n % 4 == 0);
  if (n == 0) return C2D_OK;
  hipLaunchKernelGGL(cast_f32_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream,
                     (const c2d_bf16*)src, dst, n / 4);
  return c2d_launch_status();
}

extern "C" int c2d_cast_bf16(const float* src, void* dst, long long n, void* stream) {
  C2D_CHECK_ARG(src && dst && n >= 0 && n % 4 == 0);
  if (n == 0) return C2D_OK;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, src,
                     (c2d_bf16*)dst, n / 4);
  return c2d_launch_status();
}

// ---- step-start zeroing and the loss total (round 3: no torch fill / reduce kernels in a step) ----
namespace {
// Zeroes up to C2D_ZERO_MAX byte ranges (each 16-byte aligned, a multiple of 16 bytes long) in ONE
// launch: the flat gradient bucket, the loss vector, the ROI-crop gradient map ... of a training
// step (train/trainer.py:55-61 `zero the gradients` is implicit in tf.gradients; here the
// accumulating kernels need zeroed destinations).  Block b clears 16 KiB: the ranges' chunk
// prefix sums come by value.
__global__ __launch_bounds__(256) void zero_ranges_kernel(C2dZeroRanges r) {
  int k = 0;
#pragma unroll
  for (int q = 1; q < C2D_ZERO_MAX; ++q)
    if (q < r.num && (int)blockIdx.x >= r.first_chunk[q]) k = q;
  const long long chunk = (long long)((int)blockIdx.x - r.first_chunk[k]);
  const long long n16 = r.bytes[k] >> 4;                // 16-byte elements of the range
  float4* p = reinterpret_cast<float4*>(r.ptr[k]);
  const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const long long i = chunk * 1024 + j * 256 + threadIdx.x;
    if (i < n16) p[i] = z;
  }
}

__global__ __launch_bounds__(64) void sum_small_kernel(const float* __restrict__ x, int n,
                                                      float* __restrict__ out) {
  float v = 0.f;
  for (int i = threadIdx.x; i < n; i += 64) v += x[i];
  v = c2d_wave_sum(v);
  if (threadIdx.x == 0) *out = v;
}
}  // namespace

extern "C" int c2d_zero_ranges(const C2dZeroRanges* ranges, void* stream) {
  C2D_CHECK_ARG(ranges && ranges->num >= 0 && ranges->num <= C2D_ZERO_MAX);
  C2dZeroRanges r = *ranges;
  int chunks = 0;
  for (int k = 0; k < r.num; ++k) {
    C2D_CHECK_ARG(r.ptr[k] && r.bytes[k] >= 0 && r.bytes[k] % 16 == 0 &&
                  reinterpret_cast<unsigned long long>(r.ptr[k]) % 16 == 0);
    r.first_chunk[k] = chunks;
    chunks += c2d_ceil_div(r.bytes[k], 16 * 1024);
  }
  if (chunks == 0) return C2D_OK;
  hipLaunchKernelGGL(zero_ranges_kernel, dim3(chunks), dim3(256), 0, (hipStream_t)stream, r);
  return c2d_launch_status();
}

extern "C" int c2d_sum_small(const float* x, int n, float* out, void* stream) {
  C2D_CHECK_ARG(x && out && n >= 0 && n <= 4096);
  hipLaunchKernelGGL(sum_small_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, x, n, out);
  return c2d_launch_status();
}

// ---- the reference's other optimisers (core/training_utils.py:14-71 `build_optimizer`) ----------
namespace {
// TensorFlow 1.x update rules (third party: tensorflow/core/kernels/training_ops.cc), on the
// regularised / multiplied gradient g' of c2d_adagrad_step_ex:
//   SGD       w -= lr g'
//   MOMENTUM  a = mu a + g';  w -= lr a            (Nesterov: w -= lr (g' + mu a))
//   ADAM      m = b1 m + (1-b1) g';  v = b2 v + (1-b2) g'^2;  w -= lr_t m / (sqrt(v) + eps)
//             lr_t = lr sqrt(1 - b2^t) / (1 - b1^t), computed by the caller (p3)
//   RMSPROP   ms = rho ms + (1-rho) g'^2  [centered: mg = rho mg + (1-rho) g']
//             mom = mu mom + lr g' / sqrt(ms [- mg^2] + eps);  w -= mom
__global__ __launch_bounds__(256) void optimizer_kernel(
    int kind, int flags, float* __restrict__ w, const float* __restrict__ g,
    float* __restrict__ s0, float* __restrict__ s1, float* __restrict__ s2, long long n, float lr,
    const float* __restrict__ lr_dev, float p0, float p1, float p2, float p3, float l1, float l2,
    float mult, float grad_scale, const float* __restrict__ col_mult, int ld) {
  float lr_scale = 1.0f;
  if (lr_dev) { lr_scale = lr != 0.f ? lr_dev[0] / lr : 0.f; }
  const float lre = lr * lr_scale;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n;
       i += (long long)gridDim.x * blockDim.x) {
    const float m = col_mult ? mult * col_mult[i % ld] : mult;
    if (!(m > 0.f)) continue;
    const float wi = w[i];
    const float sg = wi > 0.f ? 1.f : (wi < 0.f ? -1.f : 0.f);
    const float gi = m * (g[i] * grad_scale + l2 * wi + l1 * sg);
    if (kind == C2D_OPT_SGD) {
      w[i] = wi - lre * gi;
    } else if (kind == C2D_OPT_MOMENTUM) {
      const float a = p0 * s0[i] + gi;
      s0[i] = a;
      w[i] = (flags & 1) ? wi - lre * (gi + p0 * a) : wi - lre * a;
    } else if (kind == C2D_OPT_ADAM) {
      const float mm = p0 * s0[i] + (1.f - p0) * gi;
      const float vv = p1 * s1[i] + (1.f - p1) * gi * gi;
      s0[i] = mm; s1[i] = vv;
      w[i] = wi - (p3 * lr_scale) * mm / (sqrtf(vv) + p2);
    } else {   // RMSPROP: s0 = ms, s1 = mom, s2 = mg (centered)
      const float ms = p0 * s0[i] + (1.f - p0) * gi * gi;
      s0[i] = ms;
      float denom = ms;
      if (flags & 2) {
        const float mg = p0 * s2[i] + (1.f - p0) * gi;
        s2[i] = mg;
        denom = ms - mg * mg;
      }
      const float mom = p1 * s1[i] + lre * gi / sqrtf(denom + p2);
      s1[i] = mom;
      w[i] = wi - mom;
    }
  }
}
}  // namespace

extern "C" int c2d_optimizer_step(int kind, int flags, float* w, const float* g, float* s0,
                                  float* s1, float* s2, long long n, float lr,
                                  const float* lr_dev, float p0, float p1, float p2, float p3,
                                  float l1, float l2, float mult, float grad_scale,
                                  const float* col_mult, int ld, void* stream) {
  C2D_CHECK_ARG(w && g && n >= 0 && kind >= C2D_OPT_SGD && kind <= C2D_OPT_RMSPROP);
  C2D_CHECK_ARG(kind == C2D_OPT_SGD || s0);
  C2D_CHECK_ARG((kind != C2D_OPT_ADAM && kind != C2D_OPT_RMSPROP) || s1);
  C2D_CHECK_ARG(!(kind == C2D_OPT_RMSPROP && (flags & 2)) || s2);
  C2D_CHECK_ARG(!col_mult || ld > 0);
  if (n == 0) return C2D_OK;
  hipLaunchKernelGGL(optimizer_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, kind,
                     flags, w, g, s0, s1, s2, n, lr, lr_dev, p0, p1, p2, p3, l1, l2, mult,
                     grad_scale, col_mult, ld);
  return c2d_launch_status();
}
