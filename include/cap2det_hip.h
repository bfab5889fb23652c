/*
 * cap2det_hip.h — C-ABI of the MI355X-native Cap2Det/WSOD hot path.
 *
 * The reference (yekeren/Cap2Det) is pure Python on TF1 graph mode and exposes NO FFI for
 * this path; its "operators" are stock TF ops called from Python.  Each entry point below
 * therefore cites the reference call site (file:line under /root/reference) whose TF op(s)
 * it replaces.  A maintainer binds them with ctypes.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / C++ types.
 *   - every buffer is a caller-owned DEVICE pointer (fp32 unless noted), contiguous;
 *     activations are NHWC ("rows" = n*h*w, channels contiguous, a row stride `ld*` in
 *     floats and a channel offset `*off` so that Inception concat outputs are written in
 *     place as channel slices of one buffer).
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued asynchronously.
 *   - return 0 (C2D_OK) or a negative C2D_ERR_*; never throws, never allocates.
 *   - thread-safe per stream; no global mutable state.
 */
#ifndef CAP2DET_HIP_H_
#define CAP2DET_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define C2D_OK 0
#define C2D_ERR_INVALID_ARG (-1)
#define C2D_ERR_UNSUPPORTED (-2)
#define C2D_ERR_LAUNCH (-3)
#define C2D_ERR_WORKSPACE (-4)

/* ABI version: major*10000 + minor*100 + patch. */
int c2d_version(void);
/* Human readable message for a C2D_ERR_* code (static storage). */
const char* c2d_error_string(int code);

/* ---------------------------------------------------------------------------------------
 * ROI crop  (replaces tf.image.crop_and_resize, models/utils.py:151-155, and the fused
 * slim.max_pool2d that follows it, models/utils.py:157-160)
 * ------------------------------------------------------------------------------------- */

/* Literal tf.image.crop_and_resize (bilinear, extrapolation_value 0).
 *   feat    [batch, hf, wf, depth]
 *   boxes   [num_boxes, 4]  normalised (ymin, xmin, ymax, xmax)
 *   box_ind [num_boxes]     int32 image index of each box (out-of-range => box skipped,
 *                           its output is left untouched, as TF does)
 *   out     [num_boxes, crop, crop, depth]
 * depth must be a multiple of 4. */
int c2d_crop_and_resize_fwd(const float* feat, const float* boxes, const int32_t* box_ind,
                            float* out, int batch, int hf, int wf, int depth, int num_boxes,
                            int crop, void* stream);

/* crop_and_resize(crop x crop) followed by max_pool(k=pool_k, stride=pool_s, VALID), fused:
 * the crop tensor is never materialised.
 *   out    [num_boxes, p, p, depth],  p = (crop - pool_k) / pool_s + 1
 *   argmax [num_boxes, p, p, depth] uint8 or NULL: index (dy*pool_k+dx) of the first
 *          maximum inside the pooling window (TF MaxPoolGrad tie rule), kept for backward.
 */
int c2d_roi_crop_pool_fwd(const float* feat, const float* boxes, const int32_t* box_ind,
                          float* out, uint8_t* argmax, int batch, int hf, int wf, int depth,
                          int num_boxes, int crop, int pool_k, int pool_s, void* stream);

/* Backward of c2d_roi_crop_pool_fwd w.r.t. feat (CropAndResizeGradImage o MaxPoolGrad):
 * dfeat [batch,hf,wf,depth] must be zero-filled by the caller (or hold a gradient to be
 * accumulated into); contributions are added with fp32 atomics. */
int c2d_roi_crop_pool_bwd(const float* dout, const uint8_t* argmax, const float* boxes,
                          const int32_t* box_ind, float* dfeat, int batch, int hf, int wf,
                          int depth, int num_boxes, int crop, int pool_k, int pool_s,
                          void* stream);

#ifdef __cplusplus
}
#endif
#endif /* CAP2DET_HIP_H_ */
