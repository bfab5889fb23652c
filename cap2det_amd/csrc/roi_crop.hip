// ROI crop kernels for gfx950: tf.image.crop_and_resize (+ fused 2x2 max-pool), fwd and bwd.
//
// Semantics restated from TensorFlow 1.15 core/kernels/crop_and_resize_op.cc (third party,
// not vendored in the reference) as called by the reference at models/utils.py:147-160.
// This file is compiled with -ffp-contract=off: the sampling coordinate
// in_y = y1*(H-1) + y*height_scale decides a DISCONTINUOUS test (in_y > H-1 => whole row is
// the extrapolation value 0), so the fp32 operation order must be exactly the unfused one.
//
// Layout: feature map NHWC fp32, channels vectorised as float4 (16 B/lane, coalesced).
// One workgroup per ROI; the 2*crop sampling descriptors of the ROI live in LDS.
#include "c2d_common.h"

namespace {

constexpr int kMaxCrop = 64;

struct SampleAxis {
  int lo;      // floor index (top / left), -1 when the sample is out of range
  int hi;      // ceil index (bottom / right)
  float lerp;  // in - lo
};

// Computes the `crop` sampling descriptors of one box axis (TF: crop_and_resize_op.cc,
// CropAndResizePerBox lambda).  `a1`,`a2` = normalised box min/max on this axis, `n` = map
// extent on this axis.
__device__ __forceinline__ SampleAxis sample_axis(float a1, float a2, int n, int crop, int i) {
  const float nm1 = (float)(n - 1);
  float in;
  if (crop > 1) {
    const float scale = __fdiv_rn(__fmul_rn(__fsub_rn(a2, a1), nm1), (float)(crop - 1));
    in = __fadd_rn(__fmul_rn(a1, nm1), __fmul_rn((float)i, scale));
  } else {
    in = __fmul_rn(__fmul_rn(0.5f, __fadd_rn(a1, a2)), nm1);
  }
  SampleAxis s;
  if (in < 0.0f || in > nm1 || !(in == in)) {
    s.lo = -1;
    s.hi = -1;
    s.lerp = 0.0f;
  } else {
    s.lo = (int)floorf(in);
    s.hi = (int)ceilf(in);
    s.lerp = __fsub_rn(in, (float)s.lo);
  }
  return s;
}

__device__ __forceinline__ float4 lerp4(float4 a, float4 b, float t) {
  // a + (b - a) * t, unfused, per component (TF order).
  float4 r;
  r.x = __fadd_rn(a.x, __fmul_rn(__fsub_rn(b.x, a.x), t));
  r.y = __fadd_rn(a.y, __fmul_rn(__fsub_rn(b.y, a.y), t));
  r.z = __fadd_rn(a.z, __fmul_rn(__fsub_rn(b.z, a.z), t));
  r.w = __fadd_rn(a.w, __fmul_rn(__fsub_rn(b.w, a.w), t));
  return r;
}

// Bilinear sample of float4 channel group `d4` at (sy, sx); zeros when out of range.
__device__ __forceinline__ float4 bilinear4(const float4* __restrict__ img, int wf, int d4n,
                                            int d4, const SampleAxis& sy, const SampleAxis& sx) {
  if (sy.lo < 0 || sx.lo < 0) return make_float4(0.f, 0.f, 0.f, 0.f);
  const float4* rt = img + (size_t)sy.lo * wf * d4n + d4;
  const float4* rb = img + (size_t)sy.hi * wf * d4n + d4;
  const float4 tl = rt[(size_t)sx.lo * d4n];
  const float4 tr = rt[(size_t)sx.hi * d4n];
  const float4 bl = rb[(size_t)sx.lo * d4n];
  const float4 br = rb[(size_t)sx.hi * d4n];
  const float4 top = lerp4(tl, tr, sx.lerp);
  const float4 bot = lerp4(bl, br, sx.lerp);
  return lerp4(top, bot, sy.lerp);
}

__device__ __forceinline__ void load_axes(SampleAxis* ys, SampleAxis* xs, const float* boxes,
                                          int roi, int hf, int wf, int crop) {
  const int t = threadIdx.x;
  const float y1 = boxes[roi * 4 + 0], x1 = boxes[roi * 4 + 1];
  const float y2 = boxes[roi * 4 + 2], x2 = boxes[roi * 4 + 3];
  if (t < crop) ys[t] = sample_axis(y1, y2, hf, crop, t);
  if (t >= 64 && t < 64 + crop) xs[t - 64] = sample_axis(x1, x2, wf, crop, t - 64);
  __syncthreads();
}

__global__ __launch_bounds__(256) void crop_and_resize_fwd_kernel(
    const float4* __restrict__ feat, const float* __restrict__ boxes,
    const int32_t* __restrict__ box_ind, float4* __restrict__ out, int batch, int hf, int wf,
    int d4n, int crop) {
  __shared__ SampleAxis ys[kMaxCrop], xs[kMaxCrop];
  const int roi = blockIdx.x;
  const int b = box_ind[roi];
  if (b < 0 || b >= batch) return;  // uniform per block
  load_axes(ys, xs, boxes, roi, hf, wf, crop);
  const float4* img = feat + (size_t)b * hf * wf * d4n;
  float4* o = out + (size_t)roi * crop * crop * d4n;
  const int total = crop * crop * d4n;
  for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
    const int d4 = idx % d4n;
    const int p = idx / d4n;
    const int cx = p % crop, cy = p / crop;
    o[idx] = bilinear4(img, wf, d4n, d4, ys[cy], xs[cx]);
  }
}

__global__ __launch_bounds__(256) void roi_crop_pool_fwd_kernel(
    const float4* __restrict__ feat, const float* __restrict__ boxes,
    const int32_t* __restrict__ box_ind, float4* __restrict__ out,
    uchar4* __restrict__ argmax, int batch, int hf, int wf, int d4n, int crop, int pk, int ps,
    int pout) {
  __shared__ SampleAxis ys[kMaxCrop], xs[kMaxCrop];
  const int roi = blockIdx.x;
  const int b = box_ind[roi];
  if (b < 0 || b >= batch) return;
  load_axes(ys, xs, boxes, roi, hf, wf, crop);
  const float4* img = feat + (size_t)b * hf * wf * d4n;
  const size_t obase = (size_t)roi * pout * pout * d4n;
  const int total = pout * pout * d4n;
  for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
    const int d4 = idx % d4n;
    const int p = idx / d4n;
    const int px = p % pout, py = p / pout;
    float4 best = make_float4(0.f, 0.f, 0.f, 0.f);
    uchar4 arg = make_uchar4(0, 0, 0, 0);
    for (int dy = 0; dy < pk; ++dy) {
      for (int dx = 0; dx < pk; ++dx) {
        const float4 v = bilinear4(img, wf, d4n, d4, ys[py * ps + dy], xs[px * ps + dx]);
        const unsigned char k = (unsigned char)(dy * pk + dx);
        if (k == 0) {
          best = v;
        } else {  // strict '>' keeps the FIRST maximum (TF MaxPoolGrad tie rule)
          if (v.x > best.x) { best.x = v.x; arg.x = k; }
          if (v.y > best.y) { best.y = v.y; arg.y = k; }
          if (v.z > best.z) { best.z = v.z; arg.z = k; }
          if (v.w > best.w) { best.w = v.w; arg.w = k; }
        }
      }
    }
    out[obase + idx] = best;
    if (argmax) argmax[obase + idx] = arg;
  }
}

__device__ __forceinline__ void scatter1(float* __restrict__ dimg, int wf, int depth, int d,
                                         const SampleAxis& sy, const SampleAxis& sx, float g) {
  if (sy.lo < 0 || sx.lo < 0 || g == 0.0f) return;
  const float dtop = (1.0f - sy.lerp) * g;
  const float dbot = sy.lerp * g;
  float* rt = dimg + (size_t)sy.lo * wf * depth + d;
  float* rb = dimg + (size_t)sy.hi * wf * depth + d;
  atomicAdd(rt + (size_t)sx.lo * depth, (1.0f - sx.lerp) * dtop);
  atomicAdd(rt + (size_t)sx.hi * depth, sx.lerp * dtop);
  atomicAdd(rb + (size_t)sx.lo * depth, (1.0f - sx.lerp) * dbot);
  atomicAdd(rb + (size_t)sx.hi * depth, sx.lerp * dbot);
}

__global__ __launch_bounds__(256) void roi_crop_pool_bwd_kernel(
    const float* __restrict__ dout, const uint8_t* __restrict__ argmax,
    const float* __restrict__ boxes, const int32_t* __restrict__ box_ind,
    float* __restrict__ dfeat, int batch, int hf, int wf, int depth, int crop, int pk, int ps,
    int pout) {
  __shared__ SampleAxis ys[kMaxCrop], xs[kMaxCrop];
  const int roi = blockIdx.x;
  const int b = box_ind[roi];
  if (b < 0 || b >= batch) return;
  load_axes(ys, xs, boxes, roi, hf, wf, crop);
  float* dimg = dfeat + (size_t)b * hf * wf * depth;
  const size_t obase = (size_t)roi * pout * pout * depth;
  const int total = pout * pout * depth;
  // One lane per channel: a wave-instruction's atomics cover 256 contiguous bytes
  // (the full-rate shape of MI355X_MICROARCH 'Global float atomics').
  for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
    const int d = idx % depth;
    const int p = idx / depth;
    const int px = p % pout, py = p / pout;
    const float g = dout[obase + idx];
    const int k = argmax[obase + idx];
    scatter1(dimg, wf, depth, d, ys[py * ps + k / pk], xs[px * ps + k % pk], g);
  }
}

}  // namespace

extern "C" int c2d_crop_and_resize_fwd(const float* feat, const float* boxes,
                                       const int32_t* box_ind, float* out, int batch, int hf,
                                       int wf, int depth, int num_boxes, int crop,
                                       void* stream) {
  C2D_CHECK_ARG(feat && boxes && box_ind && out);
  C2D_CHECK_ARG(batch > 0 && hf > 0 && wf > 0 && depth > 0 && depth % 4 == 0);
  C2D_CHECK_ARG(crop > 0 && crop <= kMaxCrop && num_boxes >= 0);
  if (num_boxes == 0) return C2D_OK;
  hipLaunchKernelGGL(crop_and_resize_fwd_kernel, dim3(num_boxes), dim3(256), 0,
                     (hipStream_t)stream, (const float4*)feat, boxes, box_ind, (float4*)out,
                     batch, hf, wf, depth / 4, crop);
  return c2d_launch_status();
}

extern "C" int c2d_roi_crop_pool_fwd(const float* feat, const float* boxes,
                                     const int32_t* box_ind, float* out, uint8_t* argmax,
                                     int batch, int hf, int wf, int depth, int num_boxes,
                                     int crop, int pool_k, int pool_s, void* stream) {
  C2D_CHECK_ARG(feat && boxes && box_ind && out);
  C2D_CHECK_ARG(batch > 0 && hf > 0 && wf > 0 && depth > 0 && depth % 4 == 0);
  C2D_CHECK_ARG(crop > 0 && crop <= kMaxCrop && num_boxes >= 0);
  C2D_CHECK_ARG(pool_k > 0 && pool_s > 0 && pool_k <= crop && pool_k * pool_k <= 255);
  if (num_boxes == 0) return C2D_OK;
  const int pout = (crop - pool_k) / pool_s + 1;
  hipLaunchKernelGGL(roi_crop_pool_fwd_kernel, dim3(num_boxes), dim3(256), 0,
                     (hipStream_t)stream, (const float4*)feat, boxes, box_ind, (float4*)out,
                     (uchar4*)argmax, batch, hf, wf, depth / 4, crop, pool_k, pool_s, pout);
  return c2d_launch_status();
}

extern "C" int c2d_roi_crop_pool_bwd(const float* dout, const uint8_t* argmax,
                                     const float* boxes, const int32_t* box_ind, float* dfeat,
                                     int batch, int hf, int wf, int depth, int num_boxes,
                                     int crop, int pool_k, int pool_s, void* stream) {
  C2D_CHECK_ARG(dout && argmax && boxes && box_ind && dfeat);
  C2D_CHECK_ARG(batch > 0 && hf > 0 && wf > 0 && depth > 0);
  C2D_CHECK_ARG(crop > 0 && crop <= kMaxCrop && num_boxes >= 0);
  C2D_CHECK_ARG(pool_k > 0 && pool_s > 0 && pool_k <= crop && pool_k * pool_k <= 255);
  if (num_boxes == 0) return C2D_OK;
  const int pout = (crop - pool_k) / pool_s + 1;
  hipLaunchKernelGGL(roi_crop_pool_bwd_kernel, dim3(num_boxes), dim3(256), 0,
                     (hipStream_t)stream, dout, argmax, boxes, box_ind, dfeat, batch, hf, wf,
                     depth, crop, pool_k, pool_s, pout);
  return c2d_launch_status();
}
