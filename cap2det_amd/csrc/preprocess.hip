// GPU side of the input pipeline: everything at pixel rate after the
// host's entropy decoding.
//
// Reference call sites replaced: tf.image.flip_left_right (core/preprocess.py:48-52), the image
// resizers of core/builder.py:70-128 (tf.image.resize_images, TF1 legacy bilinear on a uint8
// image -> fp32), the zero padding of `padded_batch` (readers/cap2det_reader.py:220-247) and
// the random batch rescale (readers/cap2det_reader.py:143-172, a second legacy-bilinear resize
// of the padded fp32 batch: c2d_resize_bilinear, once per image).
// Compiled with -ffp-contract=off: bit-exact against the restated fp32 formulas.
#include "c2d_common.h"

namespace {

// One decoded uint8 RGB image -> (optionally mirrored) legacy-bilinear resize to oh x ow,
// written to the top-left corner of a zero-initialised ph x pw fp32 canvas.
__global__ __launch_bounds__(256) void resize_pad_u8_kernel(const uint8_t* __restrict__ in, int ih,
                                                            int iw, int flip,
                                                            float* __restrict__ canvas, int oh,
                                                            int ow, int ph, int pw, float hs,
                                                            float ws) {
  const long long total = (long long)ph * pw;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(i % pw), y = (int)(i / pw);
    float r = 0.f, g = 0.f, b = 0.f;
    if (y < oh && x < ow) {
      const float sy = (float)y * hs, sx = (float)x * ws;
      const float fy = floorf(sy), fx = floorf(sx);
      const int y0 = max((int)fy, 0), y1 = min((int)ceilf(sy), ih - 1);
      int x0 = max((int)fx, 0), x1 = min((int)ceilf(sx), iw - 1);
      const float ly = sy - fy, lx = sx - fx;
      if (flip) { x0 = iw - 1 - x0; x1 = iw - 1 - x1; }   // sample the mirrored image
      const uint8_t* tl = in + ((size_t)y0 * iw + x0) * 3;
      const uint8_t* tr = in + ((size_t)y0 * iw + x1) * 3;
      const uint8_t* bl = in + ((size_t)y1 * iw + x0) * 3;
      const uint8_t* br = in + ((size_t)y1 * iw + x1) * 3;
      float o[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float ftl = (float)tl[c], ftr = (float)tr[c], fbl = (float)bl[c], fbr = (float)br[c];
        const float top = ftl + (ftr - ftl) * lx;
        const float bot = fbl + (fbr - fbl) * lx;
        o[c] = top + (bot - top) * ly;
      }
      r = o[0]; g = o[1]; b = o[2];
    }
    float* dst = canvas + (size_t)i * 3;
    dst[0] = r; dst[1] = g; dst[2] = b;
  }
}

inline int grid_for(long long total) {
  long long b = (total + 255) / 256;
  if (b > 256 * 16) b = 256 * 16;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" int c2d_image_resize_pad_u8(const uint8_t* image, int ih, int iw, int flip_left_right,
                                       float* canvas, int oh, int ow, int ph, int pw,
                                       void* stream) {
  C2D_CHECK_ARG(image && canvas && ih > 0 && iw > 0 && oh > 0 && ow > 0 && ph >= oh && pw >= ow);
  const float hs = (float)ih / (float)oh, ws = (float)iw / (float)ow;
  hipLaunchKernelGGL(resize_pad_u8_kernel, dim3(grid_for((long long)ph * pw)), dim3(256), 0,
                     (hipStream_t)stream, image, ih, iw, flip_left_right ? 1 : 0, canvas, oh, ow,
                     ph, pw, hs, ws);
  return c2d_launch_status();
}
