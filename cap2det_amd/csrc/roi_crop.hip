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
#include <stdlib.h>
#include <type_traits>
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
  // a + (b - a) * t, unfused, per component (TF order; the file is compiled with
  // -ffp-contract=off).  Written on two-element vectors so that the three operations are the
  // packed v_pk_add_f32 / v_pk_mul_f32 forms: 6 instructions per float4, exactly rounded each.
  typedef float f2 __attribute__((ext_vector_type(2)));
  const f2 alo = {a.x, a.y}, ahi = {a.z, a.w}, blo = {b.x, b.y}, bhi = {b.z, b.w};
  const f2 tt = {t, t};
  const f2 rlo = alo + (blo - alo) * tt;
  const f2 rhi = ahi + (bhi - ahi) * tt;
  return make_float4(rlo.x, rlo.y, rhi.x, rhi.y);
}

// float4 copy as two v_mov_b64 (volatile: stays where it is written)
__device__ __forceinline__ float4 mov4(float4 s) {
  typedef float f2 __attribute__((ext_vector_type(2)));
  const f2 lo = {s.x, s.y}, hi = {s.z, s.w};
  f2 dlo, dhi;
  asm volatile("v_mov_b64 %0, %1" : "=v"(dlo) : "v"(lo));
  asm volatile("v_mov_b64 %0, %1" : "=v"(dhi) : "v"(hi));
  return make_float4(dlo.x, dlo.y, dhi.x, dhi.y);
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

template <typename TO>   // pooled output storage: float or bf16 (rounded once, at the store)
__global__ __launch_bounds__(256) void roi_crop_pool_fwd_kernel(
    const float4* __restrict__ feat, const float* __restrict__ boxes,
    const int32_t* __restrict__ box_ind, TO* __restrict__ out,
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
    c2d_st4(out + (obase + idx) * 4, best);
    if (argmax) argmax[obase + idx] = arg;
  }
}

// Column-streaming form of the fused crop + 2x2/stride-2 max-pool (the shipped configuration).
// The generic kernel above fetches 16 taps per pooled float4 and is bound by the L1 / address
// path (64 B/clk/CU), not by HBM.  Here a thread owns one (pooled row, channel quad) and walks
// the 14 crop columns left to right keeping the four source rows of the CURRENT and NEXT source
// column in registers: a source column is fetched once per pooled row however many crop columns
// interpolate from it (boxes narrower than 14 feature pixels — three quarters of Selective-Search
// style proposals — sample every column 2-8 times).  Which columns are new is a property of the
// box alone, so those branches are uniform over the workgroup.  Same operands, same lerp order,
// same tie rule as the generic kernel: bit-identical outputs.
template <typename TO>
__device__ __forceinline__ void crop_pool2_stream_body(
    const float4* __restrict__ img, const SampleAxis* ys, const SampleAxis* xs,
    TO* __restrict__ out, uchar4* __restrict__ argmax, size_t obase, int hf, int wf, int d4n,
    int pout, int part, int splits) {
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  const int total = pout * d4n;
  // The kernel is vector-ALU-bound (87 % busy), so the walk is written for few
  // instructions: raw buffer loads (descriptor over the image, the lane's four row offsets in
  // VGPRs, the source column's offset as the scalar operand: no address arithmetic per load),
  // two crop columns per trip (the even column's samples stay in their registers until the odd
  // column completes the pooling window), output pointers advanced per pooled cell, and the
  // zeroing of out-of-range crop rows only in waves that have one.
  const unsigned long long ub = (unsigned long long)img;
  const unsigned ub_lo = __builtin_amdgcn_readfirstlane((unsigned)ub);
  const unsigned ub_hi = __builtin_amdgcn_readfirstlane((unsigned)(ub >> 32));
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(((unsigned long long)ub_hi << 32) | ub_lo), (short)0,
      __builtin_amdgcn_readfirstlane(hf * wf * d4n * 16), 0x00020000);
  auto ld = [&](unsigned voff, int soff) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, soff, 0));
  };
  for (int idx = part * blockDim.x + threadIdx.x; idx < total; idx += splits * blockDim.x) {
    const int d4 = idx % d4n;
    const int py = idx / d4n;
    const SampleAxis y0 = ys[2 * py], y1 = ys[2 * py + 1];
    // byte offsets of the four source rows (clamped when the crop row is out of range: its
    // samples are zeroed below, the loads only need a valid address)
    const unsigned r0 = (unsigned)((max(y0.lo, 0) * wf) * d4n + d4) * 16u;
    const unsigned r1 = (unsigned)((max(y0.hi, 0) * wf) * d4n + d4) * 16u;
    const unsigned r2 = (unsigned)((max(y1.lo, 0) * wf) * d4n + d4) * 16u;
    const unsigned r3 = (unsigned)((max(y1.hi, 0) * wf) * d4n + d4) * 16u;
    const bool wave_rows_ok = __ballot(y0.lo < 0 || y1.lo < 0) == 0ull;
    float4 a0 = zero, a1 = zero, a2 = zero, a3 = zero;   // rows at source column acol
    float4 b0 = zero, b1 = zero, b2 = zero, b3 = zero;   // rows at source column bcol
    int acol = -1, bcol = -1;
    // one crop column: the two samples (crop rows 2 py, 2 py + 1) at crop column x
    auto column = [&](int x, float4& v0, float4& v1) {
      const SampleAxis sx = xs[x];                        // uniform over the workgroup
      if (sx.lo < 0) { v0 = zero; v1 = zero; return; }
      if (acol != sx.lo) {
        // (the walk advanced by one source column: the old right column is the new left one;
        // eight 64-bit moves that the compiler may neither split nor hoist above this branch)
        if (bcol == sx.lo) { a0 = mov4(b0); a1 = mov4(b1); a2 = mov4(b2); a3 = mov4(b3); }
        else {
          const int o = sx.lo * d4n * 16;
          a0 = ld(r0, o); a1 = ld(r1, o); a2 = ld(r2, o); a3 = ld(r3, o);
        }
        acol = sx.lo;
      }
      if (bcol != sx.hi) {
        // (right = left on an integer sampling coordinate: fetched again rather than copied — a
        // copy here became sixteen speculative moves in front of EVERY right-column fetch)
        const int o = sx.hi * d4n * 16;
        b0 = ld(r0, o); b1 = ld(r1, o); b2 = ld(r2, o); b3 = ld(r3, o);
        bcol = sx.hi;
      }
      v0 = lerp4(lerp4(a0, b0, sx.lerp), lerp4(a1, b1, sx.lerp), y0.lerp);
      v1 = lerp4(lerp4(a2, b2, sx.lerp), lerp4(a3, b3, sx.lerp), y1.lerp);
      if (!wave_rows_ok) {
        if (y0.lo < 0) v0 = zero;
        if (y1.lo < 0) v1 = zero;
      }
    };
    TO* op = out + (obase + (size_t)py * pout * d4n + d4) * 4;
    uchar4* ap = argmax ? argmax + obase + (size_t)py * pout * d4n + d4 : nullptr;
    for (int px = 0; px < pout; ++px) {
      float4 c0, c2, v0, v1;
      column(2 * px, c0, c2);                             // window samples k = 0 and k = 2
      column(2 * px + 1, v0, v1);                         // k = 1 and k = 3
      // k = 0, 1, 2, 3 in order; strict '>' keeps the FIRST maximum (TF MaxPoolGrad tie rule)
      float4 best = c0;
      uchar4 arg = make_uchar4(0, 0, 0, 0);
      if (v0.x > best.x) { best.x = v0.x; arg.x = 1; }
      if (v0.y > best.y) { best.y = v0.y; arg.y = 1; }
      if (v0.z > best.z) { best.z = v0.z; arg.z = 1; }
      if (v0.w > best.w) { best.w = v0.w; arg.w = 1; }
      if (c2.x > best.x) { best.x = c2.x; arg.x = 2; }
      if (c2.y > best.y) { best.y = c2.y; arg.y = 2; }
      if (c2.z > best.z) { best.z = c2.z; arg.z = 2; }
      if (c2.w > best.w) { best.w = c2.w; arg.w = 2; }
      if (v1.x > best.x) { best.x = v1.x; arg.x = 3; }
      if (v1.y > best.y) { best.y = v1.y; arg.y = 3; }
      if (v1.z > best.z) { best.z = v1.z; arg.z = 3; }
      if (v1.w > best.w) { best.w = v1.w; arg.w = 3; }
      c2d_st4(op, best);
      op += (size_t)d4n * 4;
      if (ap) { *ap = arg; ap += d4n; }
    }
  }
}

// Row-walking form of the same fused crop + 2x2/stride-2 max-pool (round 4).  TensorFlow
// interpolates horizontally first: top = tl + (tr - tl) * lx on source row y.lo, bottom likewise on
// y.hi, then out = top + (bottom - top) * ly.  The horizontally interpolated source row is the same
// number for every crop row that samples it, and consecutive crop rows of a box share source rows
// (boxes narrower than 14 feature pixels sample every row 2-8 times).  Here a lane owns ONE pooled
// column (two crop columns) of a channel quad and walks the 14 crop rows top to bottom keeping the
// two horizontally interpolated source rows of both columns in registers: one vertical lerp per
// crop pixel plus one horizontal lerp per DISTINCT source row — 1.8 lerps per crop pixel on the
// benchmark's boxes against 3 in the column-streaming form, the same operands in the same
// operation order (bit-identical output).  Which source rows are new depends on the box only, so
// the row bookkeeping is scalar.  (Round 3 tried four columns per lane: 132-146 VGPRs, three
// waves per SIMD, slower; two columns need ~80.)
template <typename TO>
__device__ __forceinline__ void crop_pool2_rowwalk_body(
    const float4* __restrict__ img, const SampleAxis* ys, const SampleAxis* xs,
    TO* __restrict__ out, uchar4* __restrict__ argmax, size_t obase, int hf, int wf, int d4n,
    int pout, int part, int splits) {
  const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
  const int total = pout * d4n;
  const unsigned long long ub = (unsigned long long)img;
  const unsigned ub_lo = __builtin_amdgcn_readfirstlane((unsigned)ub);
  const unsigned ub_hi = __builtin_amdgcn_readfirstlane((unsigned)(ub >> 32));
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(((unsigned long long)ub_hi << 32) | ub_lo), (short)0,
      __builtin_amdgcn_readfirstlane(hf * wf * d4n * 16), 0x00020000);
  auto ld = [&](unsigned voff, int soff) {
    return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, soff, 0));
  };
  const int row_bytes = wf * d4n * 16;
  for (int idx = part * blockDim.x + threadIdx.x; idx < total; idx += splits * blockDim.x) {
    const int d4 = idx % d4n;
    const int px = idx / d4n;
    const SampleAxis x0 = xs[2 * px], x1 = xs[2 * px + 1];
    const bool ok0 = x0.lo >= 0, ok1 = x1.lo >= 0;
    // byte offsets of the four column taps inside a source row (a column out of range is zeroed
    // below: its loads only need a valid address)
    const unsigned cl0 = (unsigned)(max(x0.lo, 0) * d4n + d4) * 16u;
    const unsigned ch0 = (unsigned)(max(x0.hi, 0) * d4n + d4) * 16u;
    const unsigned cl1 = (unsigned)(max(x1.lo, 0) * d4n + d4) * 16u;
    const unsigned ch1 = (unsigned)(max(x1.hi, 0) * d4n + d4) * 16u;
    // (The four column taps of the lane's two crop columns overlap on narrow boxes — 2 or 3 distinct
    //  ones; skipping the duplicate loads behind wave-uniform tests measured SLOWER, 93-97 us against
    //  79-82: two more registers cost the sixth wave per SIMD, forced back it spills.)
    float4 hl0 = zero, hl1 = zero, hh0 = zero, hh1 = zero;   // interpolated rows rowl / rowh
    int rowl = -1, rowh = -1;                                // (scalar: uniform over the workgroup)
    TO* op = out + (obase + (size_t)px * d4n + d4) * 4;
    uchar4* ap = argmax ? argmax + obase + (size_t)px * d4n + d4 : nullptr;
    for (int py = 0; py < pout; ++py) {
      float4 v[2][2];
#pragma unroll
      for (int dy = 0; dy < 2; ++dy) {
        const SampleAxis sy = ys[2 * py + dy];
        const int lo = __builtin_amdgcn_readfirstlane(sy.lo);
        const int hi = __builtin_amdgcn_readfirstlane(sy.hi);
        if (lo < 0) { v[dy][0] = zero; v[dy][1] = zero; continue; }
        const bool move_lo = lo != rowl && lo == rowh;          // the walk advanced by one row
        const bool load_lo = lo != rowl && lo != rowh;
        const bool load_hi = hi != lo && hi != rowh;
        // (every load of the step first, then the lerps)
        float4 a0, b0, a1, b1, c0, d0, c1, d1;
        if (load_lo) {
          const int so = lo * row_bytes;
          a0 = ld(cl0, so); b0 = ld(ch0, so); a1 = ld(cl1, so); b1 = ld(ch1, so);
        }
        if (load_hi) {
          const int so = hi * row_bytes;
          c0 = ld(cl0, so); d0 = ld(ch0, so); c1 = ld(cl1, so); d1 = ld(ch1, so);
        }
        if (move_lo) { hl0 = mov4(hh0); hl1 = mov4(hh1); }
        if (load_lo) { hl0 = lerp4(a0, b0, x0.lerp); hl1 = lerp4(a1, b1, x1.lerp); }
        rowl = lo;
        if (load_hi) { hh0 = lerp4(c0, d0, x0.lerp); hh1 = lerp4(c1, d1, x1.lerp); }
        else if (hi == lo && rowh != hi) { hh0 = mov4(hl0); hh1 = mov4(hl1); }   // integer coordinate
        rowh = hi;
        v[dy][0] = lerp4(hl0, hh0, sy.lerp);
        v[dy][1] = lerp4(hl1, hh1, sy.lerp);
        if (!ok0) v[dy][0] = zero;
        if (!ok1) v[dy][1] = zero;
      }
      // k = 0, 1, 2, 3 in order; strict '>' keeps the FIRST maximum (TF MaxPoolGrad tie rule)
      const float4 c0_ = v[0][0], v0 = v[0][1], c2 = v[1][0], v1 = v[1][1];
      float4 best = c0_;
      uchar4 arg = make_uchar4(0, 0, 0, 0);
      if (v0.x > best.x) { best.x = v0.x; arg.x = 1; }
      if (v0.y > best.y) { best.y = v0.y; arg.y = 1; }
      if (v0.z > best.z) { best.z = v0.z; arg.z = 1; }
      if (v0.w > best.w) { best.w = v0.w; arg.w = 1; }
      if (c2.x > best.x) { best.x = c2.x; arg.x = 2; }
      if (c2.y > best.y) { best.y = c2.y; arg.y = 2; }
      if (c2.z > best.z) { best.z = c2.z; arg.z = 2; }
      if (c2.w > best.w) { best.w = c2.w; arg.w = 2; }
      if (v1.x > best.x) { best.x = v1.x; arg.x = 3; }
      if (v1.y > best.y) { best.y = v1.y; arg.y = 3; }
      if (v1.z > best.z) { best.z = v1.z; arg.z = 3; }
      if (v1.w > best.w) { best.w = v1.w; arg.w = 3; }
      c2d_st4(op, best);
      op += (size_t)pout * d4n * 4;
      if (ap) { *ap = arg; ap += (size_t)pout * d4n; }
    }
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


// LDS-privatised backward: a workgroup owns the gradient image of CH channels
// ([hf*wf][CH] fp32 in LDS), sweeps a group of ROIs with LDS float atomics, then flushes the
// non-zero entries with one global atomic each.  Cuts the global atomic traffic from
// 4 * N*49*D adds (903 MB at the BASELINE point, ~0.7 ms at the chip's 1.3 TB/s atomic rate)
// to groups * hf*wf*D adds.  grid = (D/CH, groups, batch), block = 512.
template <int CH>
__global__ __launch_bounds__(512) void roi_crop_pool_bwd_lds_kernel(
    const float* __restrict__ dout, const uint8_t* __restrict__ argmax,
    const float* __restrict__ boxes, const int32_t* __restrict__ box_ind,
    float* __restrict__ dfeat, int hf, int wf, int depth, int num_boxes, int crop, int pk,
    int ps, int pout) {
  extern __shared__ __attribute__((aligned(16))) float acc[];
  const int npix = hf * wf;
  for (int i = threadIdx.x; i < npix * CH; i += 512) acc[i] = 0.0f;
  __syncthreads();
  const int c0 = blockIdx.x * CH;
  const int b = blockIdx.z;
  const int per = (num_boxes + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * per;
  const int r1 = min(num_boxes, r0 + per);
  const int p2 = pout * pout;
  const int ch = threadIdx.x % CH;
  constexpr int CELLS = 512 / CH;
  constexpr int U = 4;   // independent (roi, cell) items per lane per trip: loads issued together
  const long long end = (long long)r1 * p2;
  for (long long base = (long long)r0 * p2 + threadIdx.x / CH; base < end; base += U * CELLS) {
    float g[U];
    int k[U], roi[U];
    float bx[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long long idx = base + (long long)u * CELLS;
      const bool in = idx < end;
      const long long ci = in ? idx : (end - 1);
      roi[u] = (int)(ci / p2);
      const size_t o = (size_t)ci * depth + c0 + ch;
      g[u] = in ? dout[o] : 0.0f;
      k[u] = argmax[o];
      if (box_ind[roi[u]] != b) g[u] = 0.0f;
      const float4 bb = *reinterpret_cast<const float4*>(boxes + (size_t)roi[u] * 4);
      bx[u][0] = bb.x; bx[u][1] = bb.y; bx[u][2] = bb.z; bx[u][3] = bb.w;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (g[u] == 0.0f) continue;
      const long long idx = base + (long long)u * CELLS;
      const int cell = (int)(idx - (long long)roi[u] * p2);
      const int py = cell / pout, px = cell - py * pout;
      const SampleAxis sy = sample_axis(bx[u][0], bx[u][2], hf, crop, py * ps + k[u] / pk);
      const SampleAxis sx = sample_axis(bx[u][1], bx[u][3], wf, crop, px * ps + k[u] % pk);
      if (sy.lo < 0 || sx.lo < 0) continue;
      const float dtop = (1.0f - sy.lerp) * g[u], dbot = sy.lerp * g[u];
      float* top = acc + (size_t)sy.lo * wf * CH + ch;
      float* bot = acc + (size_t)sy.hi * wf * CH + ch;
      atomicAdd(top + sx.lo * CH, (1.0f - sx.lerp) * dtop);
      atomicAdd(top + sx.hi * CH, sx.lerp * dtop);
      atomicAdd(bot + sx.lo * CH, (1.0f - sx.lerp) * dbot);
      atomicAdd(bot + sx.hi * CH, sx.lerp * dbot);
    }
  }
  __syncthreads();
  float* dimg = dfeat + (size_t)b * npix * depth + c0;
  for (int i = threadIdx.x; i < npix * CH; i += 512) {
    const float v = acc[i];
    if (v != 0.0f) atomicAdd(dimg + (size_t)(i / CH) * depth + (i % CH), v);
  }
}


// ---------------------------------------------------------------------------------------------
// Atomic-free, bitwise-reproducible backward ("row owner" form).
//   1. roi_axes_kernel      : the 2*crop sampling descriptors of every box -> workspace tables.
//   2. roi_bin_rows_kernel  : for every STRIP ROW (feature row y x a range of at most 32 columns),
//                             the ORDERED list of pooled cells (roi, py, px) that can touch it (a
//                             stable compaction per cell-range segment, so the summation order
//                             below is fixed).
//   3. roi_plan_strips_kernel : cuts all lists into trips of 16 entries, strings them into ONE
//                             sequence and hands every strip workgroup an equal share of it.
//   4. roi_bwd_strip_kernel : a workgroup (share w, channel chunk) walks its trips.
//                             ONE LANE OWNS ONE CHANNEL of the row strip [wr + 1][chunk] in LDS for
//                             the whole walk, so its read-modify-writes race with nobody (no atomics,
//                             no private copies); a list entry's operands are wave-uniform (scalar
//                             loads) and the dpooled / arg-max reads of an entry are contiguous
//                             chunk-wide segments of the cell's channel row; a partial row per
//                             (workgroup, strip row) pair.
//   5. roi_bwd_sum_parts_kernel : adds a strip row's partial rows in a fixed order into the
//                             gradient map.
// LDS float atomics retire ~1.4 lane-adds per clock per CU and global float atomics are at the
// chip-wide atomic rate already; this form has neither.
// (Round 1 walked the lists with 16 lane groups x 16 channels and 16 private strips per
// workgroup: 64-byte gathers and two dependent global round trips per trip — 365 us, 8 % of HBM.)
// ---------------------------------------------------------------------------------------------
struct AxisRec { int lo, hi; float lerp; int pad; };

__global__ __launch_bounds__(256) void roi_axes_kernel(const float* __restrict__ boxes,
                                                       AxisRec* __restrict__ ys,
                                                       AxisRec* __restrict__ xs, int num_boxes,
                                                       int hf, int wf, int crop) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= num_boxes * crop) return;
  const int roi = i / crop, c = i - roi * crop;
  const float y1 = boxes[roi * 4 + 0], x1 = boxes[roi * 4 + 1];
  const float y2 = boxes[roi * 4 + 2], x2 = boxes[roi * 4 + 3];
  const SampleAxis sy = sample_axis(y1, y2, hf, crop, c);
  const SampleAxis sx = sample_axis(x1, x2, wf, crop, c);
  ys[i] = {sy.lo, sy.hi, sy.lerp, 0};
  xs[i] = {sx.lo, sx.hi, sx.lerp, 0};
}

constexpr int kBinSegs = 16;  // cell-range segments per row list = list parts of the strip kernel

// One list entry = everything about (cell, row y) that does not depend on the channel, in the
// form the strip kernel consumes without a branch: the cell's element offset in dpooled / arg-max,
// the weight of row y for either vertical sample of the 2x2 pooling window, and for either
// horizontal sample the BYTE offsets of its two column taps inside the workgroup's LDS strip
// ([wf + 1][chunk] fp32) with the lerp weight of the right one.  A sample outside the map points
// both taps at the strip's spare column wf (never stored); taps that coincide (integer coordinate)
// are moved apart with weight 0 on the second.  Computed once per cell and row instead of 576 times.
struct RowEntry {
  int cell_off;      // (roi * p*p + py * p + px) * depth
  float wy0, dwy;    // weight of row y if the argmax sample is the upper one; lower minus upper
  int off0;          // lo_bytes | hi_bytes << 16 of the left sample
  float lx0;
  int xoff;          // off0 ^ (the right sample's offsets): off = off0 ^ (xoff & -sx)
  float dlx;         // right sample's lerp weight minus the left one's
  int pad;
};
constexpr int kListPad = 16;   // zero-weight entries behind every list: the strip kernel walks
                               // whole trips without a tail test

__device__ __forceinline__ float row_weight(const AxisRec& r, int y) {
  if (r.lo < 0) return 0.0f;
  float w = 0.0f;
  if (r.lo == y) w += 1.0f - r.lerp;   // TF: dtop = (1-ly)*g goes to row lo
  if (r.hi == y) w += r.lerp;          //     dbottom = ly*g to row hi
  return w;
}

// Column taps of one horizontal sample for the strip of columns [c0, c0 + wr): LDS byte offsets
// (column - c0) * chunk * 4, taps outside the range (and samples outside the map) at the strip's
// spare column wr.
__device__ __forceinline__ void column_taps(const AxisRec& a, int wf, int chunk, int c0, int wr,
                                            int* off, float* lx) {
  int lo = a.lo, hi = a.hi;
  *lx = a.lerp;
  if (lo < 0) { lo = hi = -1; *lx = 0.0f; }                  // outside the map: the spare column
  else if (hi == lo) { hi = lo + 1 < wf ? lo + 1 : lo - 1; *lx = 0.0f; }
  lo = (lo >= c0 && lo < c0 + wr) ? lo - c0 : wr;
  hi = (hi >= c0 && hi < c0 + wr) ? hi - c0 : wr;
  *off = (lo * chunk * 4) | ((hi * chunk * 4) << 16);
}

// does either tap of the sample fall into the columns [c0, c0 + wr)?
__device__ __forceinline__ bool taps_in_range(const AxisRec& a, int c0, int wr) {
  if (a.lo < 0) return false;
  return (a.lo >= c0 && a.lo < c0 + wr) || (a.hi >= c0 && a.hi < c0 + wr);
}

// grid (hf, kBinSegs, batch): stable compaction of the cells of one segment whose 2x2 pooling
// window has a sample on row y (pool_k == 2).
__global__ __launch_bounds__(256) void roi_bin_rows_kernel(
    const AxisRec* __restrict__ ys, const AxisRec* __restrict__ xs,
    const int32_t* __restrict__ box_ind, RowEntry* __restrict__ lists,
    int32_t* __restrict__ counts, int num_boxes, int hf, int wf, int depth, int chunk, int ps,
    int pout, int crop, int cap, int nr, int wr) {
  __shared__ int wave_cnt[4];
  __shared__ int running;
  // blockIdx.x = y * nr + r: "strip row" (feature row y, column range r = [r wr, r wr + wr))
  const int y = blockIdx.x / nr, c0 = (blockIdx.x - y * nr) * wr;
  const int seg = blockIdx.y, b = blockIdx.z;
  const int p2 = pout * pout;
  const int total = num_boxes * p2;
  const int per = ((total + kBinSegs - 1) / kBinSegs + 255) / 256 * 256;
  const int beg = seg * per, end = min(total, beg + per);
  const size_t lrow = ((size_t)b * hf * nr + blockIdx.x) * kBinSegs + seg;
  RowEntry* list = lists + lrow * cap;
  if (threadIdx.x == 0) running = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int base = beg; base < end; base += 256) {
    const int cell = base + threadIdx.x;
    bool hit = false;
    RowEntry e;
    e.pad = 0;
    if (cell < end) {
      const int roi = cell / p2;
      if (box_ind[roi] == b) {
        const int c = cell - roi * p2;
        const int py = c / pout, px = c - py * pout;
        e.cell_off = cell * depth;
        e.wy0 = row_weight(ys[roi * crop + py * ps], y);
        const float wy1 = row_weight(ys[roi * crop + py * ps + 1], y);
        e.dwy = wy1 - e.wy0;
        const AxisRec x0 = xs[roi * crop + px * ps], x1 = xs[roi * crop + px * ps + 1];
        hit = (e.wy0 != 0.0f || wy1 != 0.0f) &&
              (nr == 1 || taps_in_range(x0, c0, wr) || taps_in_range(x1, c0, wr));
        if (hit) {
          int off1;
          float lx1;
          column_taps(x0, wf, chunk, c0, wr, &e.off0, &e.lx0);
          column_taps(x1, wf, chunk, c0, wr, &off1, &lx1);
          e.xoff = e.off0 ^ off1;
          e.dlx = lx1 - e.lx0;
        }
      }
    }
    const unsigned long long m = __ballot(hit);
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int off = running;
    for (int w = 0; w < wave; ++w) off += wave_cnt[w];
    if (hit) list[off + __popcll(m & ((1ull << lane) - 1ull))] = e;
    __syncthreads();
    if (threadIdx.x == 0) running += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
    __syncthreads();
  }
  if (threadIdx.x < kListPad) {          // zero-weight tail (cell 0 is always a valid address)
    RowEntry z;
    z.cell_off = 0; z.wy0 = 0.0f; z.dwy = 0.0f; z.lx0 = 0.0f; z.dlx = 0.0f; z.pad = 0;
    z.off0 = (wr * chunk * 4) | ((wr * chunk * 4) << 16);
    z.xoff = 0;
    list[running + threadIdx.x] = z;
  }
  if (threadIdx.x == 0) counts[lrow] = running;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t strip_rsrc(const void* p, long long bytes) {
  const unsigned long long ub = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)ub);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(ub >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), (short)0,
                                           __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000);
}
template <typename R> __device__ __forceinline__ R strip_load(__amdgpu_buffer_rsrc_t rs, int v, int s);
template <> __device__ __forceinline__ unsigned strip_load<unsigned>(__amdgpu_buffer_rsrc_t rs, int v, int s) {
  return __builtin_amdgcn_raw_buffer_load_b32(rs, v, s, 0);
}
template <> __device__ __forceinline__ unsigned short strip_load<unsigned short>(__amdgpu_buffer_rsrc_t rs, int v, int s) {
  return __builtin_amdgcn_raw_buffer_load_b16(rs, v, s, 0);
}
__device__ __forceinline__ float strip_value(unsigned raw) { return __uint_as_float(raw); }
__device__ __forceinline__ float strip_value(unsigned short raw) {      // bf16 -> fp32: exact
  return __uint_as_float((unsigned)raw << 16);
}

#ifndef C2D_STRIP_DBG
#define C2D_STRIP_DBG 0     // ablation builds of the strip kernel (-DC2D_STRIP_DBG=1: no vector loads, 2: no LDS updates)
#endif

constexpr int kRowParts = kBinSegs;   // strip workgroups per channel chunk: at most strip rows * kRowParts
constexpr int kTrip = 16;             // list entries per trip of the strip kernel (= kListPad)

// The strip kernel's work plan (built once per box set, with the lists).  The lists of one row
// differ in length by 6x between segments (a segment is a range of cells = of boxes) and rows at
// the map's border are half as long as those in the middle, and the strip launch is ONE round of
// workgroups: with a workgroup per (row, segment) the launch lasted as long as its longest list,
// 1.6x the mean.  So the lists are cut into TRIPS of kTrip entries (a segment's last trip runs
// into its zero-weight tail), all trips of all rows form one sequence in (row, segment) order, and
// workgroup w takes trips [w T / W, (w + 1) T / W): equal work, at most one row change per
// workgroup in practice.  Every (workgroup, row) pair owns one partial row ("slot", numbered in
// sequence order, so a row's slots are consecutive and the sum below has a fixed order).
struct StripPlan {
  int trips;          // T
  int slots;          // partial rows in use
  int pad[2];
};
// layout behind the StripPlan header (ints): wslot[W] | rowslot[2 R] (begin, end) | trip[2 maxT]
// (entry index of the trip's first entry, row)

// One workgroup; R = batch * hf rows, W strip workgroups per channel chunk.
// Slots in closed form: the (workgroup, row) pairs are the intervals the CUT POINTS — the first
// trip of every non-empty workgroup and of every non-empty row — divide the trip sequence into, so
// the pair that starts at trip p has slot |{cut points < p}| = starts(p) + rows(p) - both(p):
// workgroup starts below p (workgroup w starts at floor(w T / W); all W are non-empty when
// T >= W, else exactly one starts at every trip), non-empty rows that begin below p, and the
// points that are both.  Every thread evaluates that for its workgroups / rows; the only serial
// part is a prefix scan over the R rows.
__device__ __forceinline__ int strip_starts_below(int p, int T, int W) {   // workgroup starts < p
  if (p <= 0) return 0;
  if (T < W) return min(p, T);
  const long long c = ((long long)p * W + T - 1) / T;                       // ceil(p W / T)
  return (int)(c < W ? c : W);
}

// exclusive prefix of (a, b, c) over the 256 threads of the workgroup (wave shuffles + the four wave
// totals through LDS); *ta / *tb / *tc: the totals
__device__ __forceinline__ void plan_scan3(int& a, int& b, int& c, int* ta, int* tb, int* tc,
                                           int* wsum /* [12] */) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int ia = a, ib = b, ic = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int ua = __shfl_up(ia, o, 64), ub = __shfl_up(ib, o, 64), uc = __shfl_up(ic, o, 64);
    if (lane >= o) { ia += ua; ib += ub; ic += uc; }
  }
  __syncthreads();
  if (lane == 63) { wsum[wave] = ia; wsum[4 + wave] = ib; wsum[8 + wave] = ic; }
  __syncthreads();
  int oa = 0, ob = 0, oc = 0;
  for (int w = 0; w < wave; ++w) { oa += wsum[w]; ob += wsum[4 + w]; oc += wsum[8 + w]; }
  *ta = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  *tb = wsum[4] + wsum[5] + wsum[6] + wsum[7];
  *tc = wsum[8] + wsum[9] + wsum[10] + wsum[11];
  a = oa + ia - a; b = ob + ib - b; c = oc + ic - c;
}

__global__ __launch_bounds__(256) void roi_plan_strips_kernel(const int32_t* __restrict__ counts,
                                                              int32_t* __restrict__ plan, int R,
                                                              int W, int cap) {
  extern __shared__ int rowtrip[];        // [R + 1] first trip of every row | nzb | both | flag
  __shared__ int wsum[12];
  int* nzb = rowtrip + R + 1;             // [R + 1] non-empty rows in front of row y
  int* both = nzb + R + 1;                // [R + 1] of those: rows whose first trip is a workgroup's too
  int* flag = both + R + 1;               // [R]     row y begins where a workgroup begins
  StripPlan* head = reinterpret_cast<StripPlan*>(plan);
  int32_t* wslot = plan + 4;
  int32_t* rowslot = wslot + W;
  int32_t* trip = rowslot + 2 * R;
  // every thread owns a run of consecutive rows: its sums, a scan over the threads, its prefixes
  const int per = (R + 255) / 256;
  const int y0 = min(R, (int)threadIdx.x * per), y1 = min(R, y0 + per);
  int mine = 0, z0 = 0, z1 = 0, tot, t1, t2;
  for (int y = y0; y < y1; ++y) {
    int t = 0;
    for (int sgm = 0; sgm < kBinSegs; ++sgm) t += (counts[y * kBinSegs + sgm] + kTrip - 1) / kTrip;
    flag[y] = t;                          // (the row's trips, until its flag is known)
    mine += t;
  }
  plan_scan3(mine, z0, z1, &tot, &t1, &t2, wsum);
  const int T = tot;
  if (threadIdx.x == 0) rowtrip[R] = T;
  for (int y = y0; y < y1; ++y) {
    rowtrip[y] = mine;
    mine += flag[y];
  }
  __syncthreads();
  int n = 0, m = 0, z2 = 0;
  for (int y = y0; y < y1; ++y) {
    const int c = rowtrip[y];
    int f = 0;
    if (rowtrip[y + 1] > c) {
      if (T < W) {
        f = 1;
      } else {
        const long long w0 = ((long long)c * W + T - 1) / T;        // first workgroup at or behind c
        f = w0 < W && (int)(w0 * T / W) == c;
      }
      ++n;
      m += f;
    }
    flag[y] = f;
  }
  const int n_mine = n, m_mine = m;
  plan_scan3(n, m, z2, &t1, &t2, &tot, wsum);
  {
    int nn = n, mm = m;
    for (int y = y0; y < y1; ++y) {
      nzb[y] = nn;
      both[y] = mm;
      if (rowtrip[y + 1] > rowtrip[y]) { ++nn; mm += flag[y]; }
    }
    (void)n_mine; (void)m_mine;
  }
  if (threadIdx.x == 0) {
    nzb[R] = t1;
    both[R] = t2;
    head->trips = T;
    head->slots = strip_starts_below(T, T, W) + t1 - t2;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < R * kBinSegs; i += blockDim.x) {
    const int y = i / kBinSegs, sgm = i - y * kBinSegs;
    int pos = rowtrip[y];
    for (int q = 0; q < sgm; ++q) pos += (counts[y * kBinSegs + q] + kTrip - 1) / kTrip;
    const int n = (counts[i] + kTrip - 1) / kTrip;
    for (int j = 0; j < n; ++j) {
      trip[2 * (pos + j)] = i * cap + j * kTrip;
      trip[2 * (pos + j) + 1] = y;
    }
  }
  for (int w = threadIdx.x; w < W; w += blockDim.x) {
    const int tb = (int)((long long)w * T / W), te = (int)((long long)(w + 1) * T / W);
    int slot = 0;
    if (te > tb) {
      int lo = 0, hi = R - 1;                         // the row that holds trip tb
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (rowtrip[mid] <= tb) lo = mid; else hi = mid - 1;
      }
      // (rows of no trips share their first trip with the row behind them: take the last)
      const int y = lo;
      const int inside = rowtrip[y] < tb ? 1 : 0;
      slot = (T < W ? tb : w) + nzb[y] + inside - (both[y] + (inside ? flag[y] : 0));
    }
    wslot[w] = slot;
  }
  for (int y = threadIdx.x; y < R; y += blockDim.x) {
    int sb = 0, se = 0;
    if (rowtrip[y + 1] > rowtrip[y]) {
      sb = strip_starts_below(rowtrip[y], T, W) + nzb[y] - both[y];
      se = sb + 1 + strip_starts_below(rowtrip[y + 1], T, W) - strip_starts_below(rowtrip[y] + 1, T, W);
    }
    rowslot[2 * y] = sb;
    rowslot[2 * y + 1] = se;
  }
}

// 1-D grid of W * chunks workgroups (w fastest), block = CHUNK threads (lane <-> channel c0 + tid).
// A lane owns its channel's column of the LDS strip outright — zeroing, accumulation and the
// write-out of a partial row need no barrier.
// U list entries per half-trip: their scalar entry loads, then their 2*U vector loads, are issued
// together before the first read-modify-write.
template <int CHUNK, typename TG>
__global__ __launch_bounds__(CHUNK) void roi_bwd_strip_kernel(
    const TG* __restrict__ dout, const uint8_t* __restrict__ argmax,
    const RowEntry* __restrict__ list, const int32_t* __restrict__ plan,
    float* __restrict__ parts, int R, int W, int wf, int depth,
    long long total_cells_depth) {
  extern __shared__ __attribute__((aligned(16))) float acc[];   // [wf + 1][CHUNK] (+1: spare column)
  const int w = blockIdx.x % W;
  const int c0 = blockIdx.x / W * CHUNK;
  const int T = reinterpret_cast<const StripPlan*>(plan)->trips;
  const int tb = (int)((long long)w * T / W), te = (int)((long long)(w + 1) * T / W);
  if (tb == te) return;
  const int32_t* __restrict__ trip = plan + 4 + W + 2 * R;
  int slot = __builtin_amdgcn_readfirstlane(plan[4 + w]);
  const int tid = threadIdx.x;
  for (int x = 0; x <= wf; ++x) acc[x * CHUNK + tid] = 0.0f;
  char* const mine = reinterpret_cast<char*>(acc + tid);
  const bool on = c0 + tid < depth;                 // (ragged last chunk: depth % CHUNK != 0)
  const int cc = on ? c0 + tid : c0;
  // dpooled / arg-max through raw buffer descriptors: the lane's channel is the vector offset, the
  // entry's cell the SCALAR offset — no per-load address arithmetic in the vector ALU.
  const __amdgpu_buffer_rsrc_t gres = strip_rsrc(dout, total_cells_depth * (long long)sizeof(TG));
  const __amdgpu_buffer_rsrc_t kres = strip_rsrc(argmax, total_cells_depth);
  const int gv = cc * (int)sizeof(TG), kv = cc;
  // U list entries per trip, two register sets in ping-pong: the dpooled / arg-max values of trip
  // t+1 are in flight while trip t is accumulated into the lane's own channel column, in list
  // order (reproducible sum).  An entry's fields are wave-uniform (scalar loads); it is read
  // twice (addresses, then weights) rather than kept: two live sets do not fit the 102 SGPRs.
  // Straight-line code: zero-weight entries pad the list to whole trips, samples outside the map
  // land in the spare column, and a lane whose arg-max sample does not touch this row adds 0.
  // The per-lane choices (which vertical / horizontal sample the arg-max names) are MASKS, not
  // selects between two scalars (gfx9 reads one scalar per vector instruction: a select between
  // two costs two moves): value & -bit picks the lane's term of  w = wy0 + sy * dwy  and
  // lx = lx0 + sx * dlx, and the packed column offsets are off0 ^ (xoff & -sx).  16 vector
  // instructions per entry instead of 22 (the kernel is issue-bound: DESIGN.md section 3).
  constexpr int U = 8;
  float dbg_acc = 0.0f;         // (ablation builds only)
  using Raw = typename std::conditional<sizeof(TG) == 2, unsigned short, unsigned>::type;
  Raw g[U], gn[U];              // raw loaded values: nothing consumes them before their trip
  unsigned k[U], kn[U];
#define K_STRIP_FETCH(G, K, I0)                                                              \
  _Pragma("unroll") for (int u = 0; u < U; ++u) {                                              \
    const int o = list[(I0) + u].cell_off;                                                     \
    if (C2D_STRIP_DBG & 1) { (G)[u] = (Raw)(o + u); (K)[u] = o + u; continue; }                \
    (G)[u] = strip_load<Raw>(gres, gv, o * (int)sizeof(TG));                                   \
    (K)[u] = __builtin_amdgcn_raw_buffer_load_b8(kres, kv, o, 0);                              \
  }
  /* (all scalar entry loads of the trip first: SMEM and LDS share one wait counter, and an  */
  /*  entry load in flight would turn every LDS wait below into a wait for it as well)       */
#define K_STRIP_ADD(G, K, I0)                                                                \
  {                                                                                            \
    RowEntry e[U];                                                                             \
    _Pragma("unroll") for (int u = 0; u < U; ++u) e[u] = list[(I0) + u];                       \
    __builtin_amdgcn_sched_barrier(0);                                                         \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                            \
      const int mx = __builtin_amdgcn_sbfe((int)(K)[u], 0, 1);      /* -sx */                  \
      const int my = __builtin_amdgcn_sbfe((int)(K)[u], 1, 1);      /* -sy */                  \
      const int off = e[u].off0 ^ (e[u].xoff & mx);                                            \
      float* const plo = reinterpret_cast<float*>(mine + (off & 0xffff));                      \
      float* const phi = reinterpret_cast<float*>(mine + ((unsigned)off >> 16));               \
      float alo, ahi;                                                                          \
      if (C2D_STRIP_DBG & 2) { alo = dbg_acc; ahi = __int_as_float(off); }                     \
      else { alo = *plo; ahi = *phi; }                 /* (distinct columns by construction) */ \
      const float gf = strip_value((G)[u]);                                                    \
      const float v = __builtin_fmaf(__int_as_float(__float_as_int(gf) & my), e[u].dwy,        \
                                     gf * e[u].wy0);                                           \
      const float whi = __builtin_fmaf(__int_as_float(__float_as_int(v) & mx), e[u].dlx,       \
                                       v * e[u].lx0);                                          \
      /* TF CropAndResizeGradImage: (1 - lx) * dtop to column lo, lx * dtop to column hi */    \
      if (C2D_STRIP_DBG & 2) { dbg_acc = alo + (v - whi) + (ahi + whi); continue; }            \
      *plo = alo + (v - whi);                                                                  \
      *phi = ahi + whi;                                                                        \
    }                                                                                          \
  }
  int row = __builtin_amdgcn_readfirstlane(trip[2 * tb + 1]);
  auto flush = [&](int sl) {               // the lane's column of partial row `sl`
    float* drow = parts + ((size_t)sl * wf) * depth + c0 + tid;
    if (on)
      for (int x = 0; x < wf; ++x) drow[(size_t)x * depth] = acc[x * CHUNK + tid];
  };
  int i0 = __builtin_amdgcn_readfirstlane(trip[2 * tb]);
  K_STRIP_FETCH(g, k, i0);
  for (int t = tb; t < te; ++t) {                  // fixed order: the sum is reproducible
    const int r = __builtin_amdgcn_readfirstlane(trip[2 * t + 1]);
    if (r != row) {
      flush(slot);
      for (int x = 0; x <= wf; ++x) acc[x * CHUNK + tid] = 0.0f;
      ++slot;
      row = r;
    }
    const int tn = min(t + 1, te - 1);             // (the last trip re-reads itself: harmless)
    const int inext = __builtin_amdgcn_readfirstlane(trip[2 * tn]);
    K_STRIP_FETCH(gn, kn, i0 + U);               // (reads into the zero-weight tail at most)
    __builtin_amdgcn_sched_barrier(0);
    K_STRIP_ADD(g, k, i0);
    __builtin_amdgcn_sched_barrier(0);
    K_STRIP_FETCH(g, k, inext);
    __builtin_amdgcn_sched_barrier(0);
    K_STRIP_ADD(gn, kn, i0 + U);
    i0 = inext;
  }
#undef K_STRIP_ADD
#undef K_STRIP_FETCH
  if (C2D_STRIP_DBG & 2) acc[tid] = dbg_acc;
  flush(slot);
}

// dfeat[strip row] += its partial rows, in slot order (fixed), one float4 per lane.
// grid (x-blocks, R): strip row v = (feature row v / nr, columns [(v % nr) wr, ... + wr)).
__global__ __launch_bounds__(256) void roi_bwd_sum_parts_kernel(const float4* __restrict__ parts,
                                                                const int32_t* __restrict__ plan,
                                                                float4* __restrict__ dfeat, int W,
                                                                int R, int nr, int wr, int wf,
                                                                int depth4) {
  const int v = blockIdx.y;
  const int32_t* rowslot = plan + 4 + W;
  const int sb = rowslot[2 * v], se = rowslot[2 * v + 1];
  const int y = v / nr, c0 = (v - y * nr) * wr;
  const int row4 = wr * depth4;                               // float4 per partial row
  const int live4 = min(wr, wf - c0) * depth4;                // (the last range may be narrower)
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (sb == se || i >= live4) return;
  float4 s = parts[(size_t)sb * row4 + i];
  int q = sb + 1;
  for (; q + 4 <= se; q += 4) {            // four loads in flight, summed in slot order
    const float4 v0 = parts[(size_t)q * row4 + i], v1 = parts[(size_t)(q + 1) * row4 + i];
    const float4 v2 = parts[(size_t)(q + 2) * row4 + i], v3 = parts[(size_t)(q + 3) * row4 + i];
    s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
    s.x += v1.x; s.y += v1.y; s.z += v1.z; s.w += v1.w;
    s.x += v2.x; s.y += v2.y; s.z += v2.z; s.w += v2.w;
    s.x += v3.x; s.y += v3.y; s.z += v3.z; s.w += v3.w;
  }
  for (; q < se; ++q) {
    const float4 u = parts[(size_t)q * row4 + i];
    s.x += u.x; s.y += u.y; s.z += u.z; s.w += u.w;
  }
  float4* out = dfeat + ((size_t)y * wf + c0) * depth4 + i;
  float4 d = *out;
  d.x += s.x; d.y += s.y; d.z += s.z; d.w += s.w;
  *out = d;
}

template <typename TO>
__global__ __launch_bounds__(256) void roi_crop_pool2_fwd_stream_kernel(
    const float4* __restrict__ feat, const float* __restrict__ boxes,
    const int32_t* __restrict__ box_ind, TO* __restrict__ out,
    uchar4* __restrict__ argmax, int batch, int hf, int wf, int d4n, int crop, int pout,
    int splits) {
  __shared__ SampleAxis ys[kMaxCrop], xs[kMaxCrop];
  // `splits` workgroups share a ROI (interleaved (pooled row, channel quad) items): shorter
  // workgroups = a shorter tail when 2000 ROIs meet 256 CUs x 6 resident workgroups
  const int roi = blockIdx.x / splits, part = blockIdx.x - roi * splits;
  const int b = box_ind[roi];
  if (b < 0 || b >= batch) return;
  load_axes(ys, xs, boxes, roi, hf, wf, crop);
  crop_pool2_stream_body<TO>(feat + (size_t)b * hf * wf * d4n, ys, xs, out, argmax,
                             (size_t)roi * pout * pout * d4n, hf, wf, d4n, pout, part, splits);
}

template <typename TO>
__global__ __launch_bounds__(256) void roi_crop_pool2_fwd_rowwalk_kernel(
    const float4* __restrict__ feat, const float* __restrict__ boxes,
    const int32_t* __restrict__ box_ind, TO* __restrict__ out,
    uchar4* __restrict__ argmax, int batch, int hf, int wf, int d4n, int crop, int pout,
    int splits) {
  __shared__ SampleAxis ys[kMaxCrop], xs[kMaxCrop];
  const int roi = blockIdx.x / splits, part = blockIdx.x - roi * splits;
  const int b = box_ind[roi];
  if (b < 0 || b >= batch) return;
  load_axes(ys, xs, boxes, roi, hf, wf, crop);
  crop_pool2_rowwalk_body<TO>(feat + (size_t)b * hf * wf * d4n, ys, xs, out, argmax,
                              (size_t)roi * pout * pout * d4n, hf, wf, d4n, pout, part, splits);
}

// (A prefetching variant — three-slot register ring over the sorted list of needed columns, the
// next column's loads in flight while the current one is interpolated — measured 151 us against
// 108 us: 104 VGPRs cost two of the six waves per SIMD and the slot selects add VALU work.)
}  // namespace

// 2x2 / stride-2 pooling over an even crop: the column-streaming kernel (C2D_TUNE=1
// crop_stream=0 keeps the generic one, for A/B timing).
static int crop_stream_splits() {
  // measured (tools/bench_crop_fwd.py, N = 2000): 1 -> 119.6 us, 2 -> 107.0, 4 -> 104.6
  static const char* e = c2d_tune_on() ? c2d_tune_get("crop_split") : nullptr;
  const int v = e ? atoi(e) : 4;
  return v >= 1 && v <= 8 ? v : 1;
}

static int crop_stream_form(int crop, int pool_k, int pool_s, int pout, int hf, int wf,
                            int depth) {
  static const char* e = c2d_tune_on() ? c2d_tune_get("crop_stream") : nullptr;
  const int want = e ? atoi(e) : 2;       // 2: row-walking form (round 4), 1: column-streaming form
  // the stream kernels address ONE image's map through a raw buffer descriptor with 32-bit byte
  // offsets: maps of 2 GiB and more take the generic kernel (64-bit pointer arithmetic)
  if ((long long)hf * wf * depth * 4 >= (1ll << 31)) return 0;
  return (pool_k == 2 && pool_s == 2 && crop == 2 * pout) ? want : 0;
}

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
  const int form = crop_stream_form(crop, pool_k, pool_s, pout, hf, wf, depth);
  if (form == 2)
    hipLaunchKernelGGL(roi_crop_pool2_fwd_rowwalk_kernel<float>,
                       dim3(num_boxes * crop_stream_splits()), dim3(256), 0,
                       (hipStream_t)stream, (const float4*)feat, boxes, box_ind, out,
                       (uchar4*)argmax, batch, hf, wf, depth / 4, crop, pout, crop_stream_splits());
  else if (form == 1)
    hipLaunchKernelGGL(roi_crop_pool2_fwd_stream_kernel<float>,
                       dim3(num_boxes * crop_stream_splits()), dim3(256), 0,
                       (hipStream_t)stream, (const float4*)feat, boxes, box_ind, out,
                       (uchar4*)argmax, batch, hf, wf, depth / 4, crop, pout, crop_stream_splits());
  else
    hipLaunchKernelGGL(roi_crop_pool_fwd_kernel<float>, dim3(num_boxes), dim3(256), 0,
                       (hipStream_t)stream, (const float4*)feat, boxes, box_ind, out,
                       (uchar4*)argmax, batch, hf, wf, depth / 4, crop, pool_k, pool_s, pout);
  return c2d_launch_status();
}

extern "C" int c2d_roi_crop_pool_fwd_bf16(const float* feat, const float* boxes,
                                          const int32_t* box_ind, void* out, uint8_t* argmax,
                                          int batch, int hf, int wf, int depth, int num_boxes,
                                          int crop, int pool_k, int pool_s, void* stream) {
  C2D_CHECK_ARG(feat && boxes && box_ind && out);
  C2D_CHECK_ARG(batch > 0 && hf > 0 && wf > 0 && depth > 0 && depth % 4 == 0);
  C2D_CHECK_ARG(crop > 0 && crop <= kMaxCrop && num_boxes >= 0);
  C2D_CHECK_ARG(pool_k > 0 && pool_s > 0 && pool_k <= crop && pool_k * pool_k <= 255);
  if (num_boxes == 0) return C2D_OK;
  const int pout = (crop - pool_k) / pool_s + 1;
  const int form = crop_stream_form(crop, pool_k, pool_s, pout, hf, wf, depth);
  if (form == 2)
    hipLaunchKernelGGL(roi_crop_pool2_fwd_rowwalk_kernel<c2d_bf16>,
                       dim3(num_boxes * crop_stream_splits()), dim3(256), 0,
                       (hipStream_t)stream, (const float4*)feat, boxes, box_ind, (c2d_bf16*)out,
                       (uchar4*)argmax, batch, hf, wf, depth / 4, crop, pout, crop_stream_splits());
  else if (form == 1)
    hipLaunchKernelGGL(roi_crop_pool2_fwd_stream_kernel<c2d_bf16>,
                       dim3(num_boxes * crop_stream_splits()), dim3(256), 0,
                       (hipStream_t)stream, (const float4*)feat, boxes, box_ind, (c2d_bf16*)out,
                       (uchar4*)argmax, batch, hf, wf, depth / 4, crop, pout, crop_stream_splits());
  else
    hipLaunchKernelGGL(roi_crop_pool_fwd_kernel<c2d_bf16>, dim3(num_boxes), dim3(256), 0,
                       (hipStream_t)stream, (const float4*)feat, boxes, box_ind, (c2d_bf16*)out,
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
  // LDS-privatised path when a [hf*wf][CH] fp32 image fits 128 KiB of the CU's 160 KiB LDS.
  const size_t pix_bytes = (size_t)hf * wf * sizeof(float);
  int ch = 0;
  for (int c : {16, 8})   // <= 64 KiB per workgroup: two 512-thread workgroups per CU
    if (depth % c == 0 && pix_bytes * c <= 64 * 1024) { ch = c; break; }
  if (ch != 0 && num_boxes >= 64) {
    const int chunks = depth / ch;
    int groups = (4 * 256 + chunks * batch - 1) / (chunks * batch);   // ~4 workgroups per CU
    if (groups > num_boxes / 16) groups = num_boxes / 16 > 0 ? num_boxes / 16 : 1;
    const dim3 grid(chunks, groups, batch);
    const size_t smem = pix_bytes * ch;
#define K_BWD_LDS(CHV)                                                                      \
  static const hipError_t attr_##CHV = hipFuncSetAttribute(                                   \
      (const void*)roi_crop_pool_bwd_lds_kernel<CHV>,                                         \
      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                                \
  (void)attr_##CHV;                                                                           \
  hipLaunchKernelGGL((roi_crop_pool_bwd_lds_kernel<CHV>), grid, dim3(512), smem,              \
                     (hipStream_t)stream, dout, argmax, boxes, box_ind, dfeat, hf, wf, depth, \
                     num_boxes, crop, pool_k, pool_s, pout)
    if (ch == 16) { K_BWD_LDS(16); }
    else { K_BWD_LDS(8); }
#undef K_BWD_LDS
    return c2d_launch_status();
  }
  hipLaunchKernelGGL(roi_crop_pool_bwd_kernel, dim3(num_boxes), dim3(256), 0,
                     (hipStream_t)stream, dout, argmax, boxes, box_ind, dfeat, batch, hf, wf,
                     depth, crop, pool_k, pool_s, pout);
  return c2d_launch_status();
}

// A feature row wider than 32 columns is walked as nr STRIP ROWS of wr <= 32 columns each (round 4):
// the LDS strip of a workgroup is (wr + 1) x chunk floats, and how many of them fit a CU is how many
// waves hide the walk's scalar-load and LDS round trips — 33 columns x 192 channels are 6
// workgroups (18 waves) per CU on the benchmark's 32-column map, 85 columns were 2 (6 waves) on the
// 84-column maps of 1000-px images.  A cell is listed in every range one of its taps falls into (its
// other taps go to that strip's spare column): a few % more list entries.  Measured (tools/
// bench_crop_bwd.py, accumulation + part sum, fp32 / bf16 gradients, us): 84 x 84 map, 1000 boxes:
// whole rows 210 / 206, strips of 42: 129 / 124, 28: 114 / 112, 20: 121 / 119, 14: 127 / 127;
// 63 x 63: 151 / 145 -> 96 / 94 (32 columns); 48 x 48, 2000 boxes: 224 / 202 -> 179 / 156 (24); the
// 32 x 32 map LOSES with two strips of 16 (144 / 131 -> 165 / 141) and keeps whole rows.
// C2D_TUNE=roi_strip_cols=<columns> (0: whole rows).
static void strip_ranges(int wf, int* nr, int* wr) {
  static const int target = (c2d_tune_on() && c2d_tune_get("roi_strip_cols"))
                                ? atoi(c2d_tune_get("roi_strip_cols")) : 32;
  const int t = target > 0 ? target : wf;
  *nr = (wf + t - 1) / t;
  *wr = (wf + *nr - 1) / *nr;
}

extern "C" long long c2d_roi_crop_pool_bwd_workspace_bytes(int batch, int hf, int wf, int depth,
                                                           int num_boxes, int crop, int pool_k,
                                                           int pool_s) {
  if (batch <= 0 || hf <= 0 || wf <= 0 || depth <= 0 || num_boxes < 0 || crop <= 0 ||
      pool_k <= 0 || pool_s <= 0)
    return -1;
  const long long pout = (crop - pool_k) / pool_s + 1;
  const long long cells = (long long)num_boxes * pout * pout;
  const long long seg_cap = ((cells + kBinSegs - 1) / kBinSegs + 255) / 256 * 256 + kListPad;
  int nr, wr;
  strip_ranges(wf, &nr, &wr);
  const long long rows = (long long)batch * hf * nr, wgs = rows * kRowParts;
  const long long plan_ints = 4 + wgs + 2 * rows + 2 * rows * kBinSegs * (seg_cap / kTrip);
  return 2ll * num_boxes * crop * (long long)sizeof(AxisRec) + 256 +
         rows * kBinSegs * 4 + 256 +
         rows * kBinSegs * seg_cap * (long long)sizeof(RowEntry) + 256 +
         plan_ints * 4 + 256 +
         (wgs + rows) * wr * depth * 4;          // one partial row per (workgroup, strip row) pair
}

// Channel chunk of the strip kernel = its workgroup size: the largest of 256 / 192 / 128 that
// divides the depth, else 64 with a ragged last chunk (576 -> 192: 768-byte segments per cell) —
// and small enough that the byte offset of the strip's spare column still fits the 16-bit halves
// of RowEntry::off0 (strips are at most 32 columns wide since round 4: always).
// 0: no chunk fits (the caller falls back to the atomic kernel).
static int strip_chunk(int depth, int wf) {
  for (int c : {256, 192, 128})
    if (depth % c == 0 && (long long)wf * c * 4 <= 65535) return c;
  return (long long)wf * 64 * 4 <= 65535 ? 64 : 0;
}

extern "C" int c2d_roi_crop_pool_bwd_ws_supported(int wf, int depth, int crop, int pool_k,
                                                  int pool_s) {
  if (wf < 2 || depth <= 0 || depth % 16 != 0 || pool_k != 2 || pool_s <= 0 || crop <= 0 ||
      crop > kMaxCrop || (crop - pool_k) / pool_s + 1 > 16)
    return 0;
  int nr, wr;
  strip_ranges(wf, &nr, &wr);
  return wf <= 255 ? strip_chunk(depth, wr) : 0;
}

// Everything c2d_roi_crop_pool_bwd_ws / _prepare / _run check before they touch the workspace, as
// one query a caller can cache per shape (elem_size: 4 = fp32, 2 = bf16 pooled gradients):
// returns the channel chunk (> 0) when the row-owner form covers the shape, 0 when it would return
// C2D_ERR_UNSUPPORTED (the caller then runs c2d_roi_crop_pool_bwd).  Beyond the per-map rules of
// c2d_roi_crop_pool_bwd_ws_supported: the pooled gradient must stay below 2 GiB (32-bit buffer
// offsets: fp32 7x7x576 cells -> fewer than 19,022 boxes), the row lists below 2^31 entries, and
// the plan kernel's row table inside 64 KiB of LDS (batch * hf * strips-per-row <= 4095).
extern "C" int c2d_roi_crop_pool_bwd_ws_shape_supported(int batch, int hf, int wf, int depth,
                                                        int num_boxes, int crop, int pool_k,
                                                        int pool_s, int elem_size) {
  const int chunk = c2d_roi_crop_pool_bwd_ws_supported(wf, depth, crop, pool_k, pool_s);
  if (chunk <= 0 || batch <= 0 || hf <= 0 || num_boxes < 0 || (elem_size != 4 && elem_size != 2))
    return 0;
  const long long pout = (crop - pool_k) / pool_s + 1;
  if (num_boxes >= (1 << 23) || (long long)num_boxes * pout * pout * depth * elem_size >= (1ll << 31))
    return 0;
  int nr, wr;
  strip_ranges(wf, &nr, &wr);
  const long long cap = ((num_boxes * pout * pout + kBinSegs - 1) / kBinSegs + 255) / 256 * 256 + kListPad;
  const long long R = (long long)batch * hf * nr;
  if (R * kBinSegs * cap >= (1ll << 31) || (4 * R + 3) * 4 > 64 * 1024) return 0;
  return chunk;
}

template <typename TG>
static int roi_crop_pool_bwd_ws_impl(const TG* dout, const uint8_t* argmax, const float* boxes,
                                     const int32_t* box_ind, float* dfeat, int batch, int hf, int wf,
                                     int depth, int num_boxes, int crop, int pool_k, int pool_s,
                                     void* workspace, long long workspace_bytes, void* stream,
                                     int phase = 0) {
  // phase 0: everything; 1: only the box-dependent tables and row lists (c2d_roi_crop_pool_bwd_
  // prepare: they depend on the boxes alone, so a caller can build them during the forward pass,
  // off the critical path); 2: only the accumulation (c2d_roi_crop_pool_bwd_run)
  C2D_CHECK_ARG(boxes && box_ind && workspace && (phase == 1 || (dout && argmax && dfeat)));
  C2D_CHECK_ARG(batch > 0 && hf > 0 && wf > 0 && depth > 0 && depth % 16 == 0);
  C2D_CHECK_ARG(crop > 0 && crop <= kMaxCrop && num_boxes >= 0 && pool_s > 0);
  int nr, wr;
  strip_ranges(wf, &nr, &wr);
  const int chunk = strip_chunk(depth, wr);
  // (the strip kernel addresses dpooled through a raw buffer descriptor: 32-bit byte offsets; the
  //  plan kernel keeps its row table in LDS — every limit lives in the shape query)
  if (chunk == 0 || c2d_roi_crop_pool_bwd_ws_shape_supported(batch, hf, wf, depth, num_boxes, crop,
                                                             pool_k, pool_s, (int)sizeof(TG)) <= 0)
    return C2D_ERR_UNSUPPORTED;
  if (num_boxes == 0) return C2D_OK;
  if (workspace_bytes <
      c2d_roi_crop_pool_bwd_workspace_bytes(batch, hf, wf, depth, num_boxes, crop, pool_k,
                                            pool_s))
    return C2D_ERR_WORKSPACE;
  const int pout = (crop - pool_k) / pool_s + 1;
  const int cap = ((num_boxes * pout * pout + kBinSegs - 1) / kBinSegs + 255) / 256 * 256 + kListPad;
  const int R = batch * hf * nr;              // strip rows: (image, feature row, column range)
  char* w = (char*)workspace;
  AxisRec* ys = (AxisRec*)w;
  AxisRec* xs = ys + (size_t)num_boxes * crop;
  size_t off = ((2 * (size_t)num_boxes * crop * sizeof(AxisRec)) + 255) / 256 * 256;
  int32_t* counts = (int32_t*)(w + off);
  off = (off + (size_t)R * kBinSegs * 4 + 255) / 256 * 256;
  RowEntry* lists = (RowEntry*)(w + off);
  off = (off + (size_t)R * kBinSegs * cap * sizeof(RowEntry) + 255) / 256 * 256;
  // Strip workgroups per channel chunk.  A workgroup zeroes and writes out a whole partial row of
  // wr x chunk floats whatever its share, so: no more than give every workgroup ~16 trips of the
  // expected ~3 list entries per cell (on the 63 x 84 maps of two 1000-px images with 500 boxes
  // each, 2016 workgroups of 4 trips wrote 414 MB of partial rows); ONE round of resident
  // workgroups when that is within reach (equal shares: a second round of a few would double the
  // launch), whole rounds otherwise; at most the R * kRowParts the workspace is sized for.
  const int nchunks = (depth + chunk - 1) / chunk;
  const size_t lds = (size_t)chunk * (wr + 1) * sizeof(float);
  int per_cu = (int)((160 * 1024) / lds);
  if (per_cu > 30 * 64 / chunk) per_cu = 30 * 64 / chunk;      // (wave slots: 42 VGPRs)
  if (per_cu < 1) per_cu = 1;
  const int round1 = 256 * per_cu / nchunks > 8 ? 256 * per_cu / nchunks / 8 * 8 : 8;
  const long long trips_est = (long long)num_boxes * pout * pout * 3 / kTrip;
  int W = (int)(trips_est / 16 < 256 ? 256 : (trips_est / 16 + 7) / 8 * 8);
  if (W > round1) W = W <= 3 * round1 / 2 ? round1 : (W + round1 - 1) / round1 * round1;
  if (W > R * kRowParts) W = R * kRowParts;
  int32_t* plan = (int32_t*)(w + off);
  off = (off + (4 + (size_t)W + 2 * (size_t)R + 2 * (size_t)R * kBinSegs * (cap / kTrip)) * 4 + 255) /
        256 * 256;
  float* parts = (float*)(w + off);
  hipStream_t st = (hipStream_t)stream;
  if (phase != 2) {
    hipLaunchKernelGGL(roi_axes_kernel, dim3(c2d_ceil_div((long long)num_boxes * crop, 256)),
                       dim3(256), 0, st, boxes, ys, xs, num_boxes, hf, wf, crop);
    hipLaunchKernelGGL(roi_bin_rows_kernel, dim3(hf * nr, kBinSegs, batch), dim3(256), 0, st, ys,
                       xs, box_ind, lists, counts, num_boxes, hf, wf, depth, chunk, pool_s, pout,
                       crop, cap, nr, wr);
    hipLaunchKernelGGL(roi_plan_strips_kernel, dim3(1), dim3(256), (size_t)(4 * R + 3) * 4, st,
                       counts, plan, R, W, cap);
  }
  if (phase == 1) return c2d_launch_status();
#define K_STRIP(CHV)                                                                          \
  hipLaunchKernelGGL((roi_bwd_strip_kernel<CHV, TG>), dim3(W * nchunks), dim3(CHV), lds, st,    \
                     dout, argmax, lists, plan, parts, R, W, wr, depth,                         \
                     (long long)num_boxes * pout * pout * depth)
  if (chunk == 256) { K_STRIP(256); }
  else if (chunk == 192) { K_STRIP(192); }
  else if (chunk == 128) { K_STRIP(128); }
  else { K_STRIP(64); }
#undef K_STRIP
  const int row4 = wr * depth / 4;
  hipLaunchKernelGGL(roi_bwd_sum_parts_kernel, dim3((row4 + 255) / 256, R), dim3(256), 0, st,
                     (const float4*)parts, plan, (float4*)dfeat, W, R, nr, wr, wf, depth / 4);
  return c2d_launch_status();
}

extern "C" int c2d_roi_crop_pool_bwd_ws(const float* dout, const uint8_t* argmax,
                                        const float* boxes, const int32_t* box_ind, float* dfeat,
                                        int batch, int hf, int wf, int depth, int num_boxes,
                                        int crop, int pool_k, int pool_s, void* workspace,
                                        long long workspace_bytes, void* stream) {
  return roi_crop_pool_bwd_ws_impl<float>(dout, argmax, boxes, box_ind, dfeat, batch, hf, wf, depth,
                                          num_boxes, crop, pool_k, pool_s, workspace,
                                          workspace_bytes, stream);
}

extern "C" int c2d_roi_crop_pool_bwd_ws_bf16(const void* dout, const uint8_t* argmax,
                                             const float* boxes, const int32_t* box_ind,
                                             float* dfeat, int batch, int hf, int wf, int depth,
                                             int num_boxes, int crop, int pool_k, int pool_s,
                                             void* workspace, long long workspace_bytes,
                                             void* stream) {
  return roi_crop_pool_bwd_ws_impl<c2d_bf16>((const c2d_bf16*)dout, argmax, boxes, box_ind, dfeat,
                                             batch, hf, wf, depth, num_boxes, crop, pool_k, pool_s,
                                             workspace, workspace_bytes, stream);
}

// The two halves of c2d_roi_crop_pool_bwd_ws (same workspace): `prepare` builds what depends on
// the boxes alone, `run` accumulates.  prepare may run on another stream during the forward pass.
extern "C" int c2d_roi_crop_pool_bwd_prepare(const float* boxes, const int32_t* box_ind, int batch,
                                             int hf, int wf, int depth, int num_boxes, int crop,
                                             int pool_k, int pool_s, void* workspace,
                                             long long workspace_bytes, void* stream) {
  return roi_crop_pool_bwd_ws_impl<float>(nullptr, nullptr, boxes, box_ind, nullptr, batch, hf, wf,
                                          depth, num_boxes, crop, pool_k, pool_s, workspace,
                                          workspace_bytes, stream, 1);
}

extern "C" int c2d_roi_crop_pool_bwd_run(const float* dout, const uint8_t* argmax,
                                         const float* boxes, const int32_t* box_ind, float* dfeat,
                                         int batch, int hf, int wf, int depth, int num_boxes,
                                         int crop, int pool_k, int pool_s, void* workspace,
                                         long long workspace_bytes, void* stream) {
  return roi_crop_pool_bwd_ws_impl<float>(dout, argmax, boxes, box_ind, dfeat, batch, hf, wf, depth,
                                          num_boxes, crop, pool_k, pool_s, workspace,
                                          workspace_bytes, stream, 2);
}

extern "C" int c2d_roi_crop_pool_bwd_run_bf16(const void* dout, const uint8_t* argmax,
                                              const float* boxes, const int32_t* box_ind,
                                              float* dfeat, int batch, int hf, int wf, int depth,
                                              int num_boxes, int crop, int pool_k, int pool_s,
                                              void* workspace, long long workspace_bytes,
                                              void* stream) {
  return roi_crop_pool_bwd_ws_impl<c2d_bf16>((const c2d_bf16*)dout, argmax, boxes, box_ind, dfeat,
                                             batch, hf, wf, depth, num_boxes, crop, pool_k, pool_s,
                                             workspace, workspace_bytes, stream, 2);
}
