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
#define C2D_ERR_DATA (-5)        /* malformed input data (record framing, protobuf, JPEG stream) */

/* ABI version, bumped whenever an entry point is added or a signature changes (round 1: 100 with
 * 28 entry points; round 4: 400; round 5: 500; round 6: 600).  c2d_version() returns the value the library was built with:
 * a host side compiled against another header must refuse to run (cap2det_amd/_lib.py does). */
#define C2D_ABI_VERSION 600
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

/* ---------------------------------------------------------------------------------------
 * Convolution as implicit GEMM on the fp32 MFMA pipe  (replaces slim.conv2d + inference
 * BatchNorm + ReLU of the Inception-V2 extractor [third party] called at
 * models/utils.py:133-136 and :165-167, slim.fully_connected at
 * models/cap2det_model.py:79-88,191-197, and the gradients TF derives for them when
 * train/trainer.py:141-146 builds train_op).  SAME padding (TF rule), NHWC, fp32.
 *   x   rows n*ih*iw, row stride ldx floats, channels [xoff, xoff+cin)
 *   y   rows n*oh*ow (oh = ceil(ih/stride)), row stride ldy, channels [yoff, yoff+cout)
 * ------------------------------------------------------------------------------------- */

/* y = act(scale[co] * conv(x, W) + shift[co]).  wt is W transposed per tap:
 * [kh*kw][cout][cin] (see c2d_transpose_taps).  scale/shift may be NULL (1 / 0); relu != 0
 * applies max(.,0).  cin % 16 == 0. */
int c2d_conv_fwd(const float* x, int ldx, int xoff, const float* wt, const float* scale,
                 const float* shift, float* y, int ldy, int yoff, int n, int ih, int iw,
                 int cin, int cout, int kh, int kw, int stride, int relu, void* stream);

/* dx (+)= Conv2DBackpropInput(dc, W).  w is HWIO [kh*kw][cin][cout]; dc has the conv OUTPUT
 * geometry (rows n*oh*ow, stride ldc, channels [coff, coff+cout)); dx has the input geometry.
 * cout % 16 == 0.  accumulate != 0 adds into dx (sum over Inception branches). */
int c2d_conv_dgrad(const float* dc, int ldc, int coff, const float* w, float* dx, int lddx,
                   int dxoff, int n, int ih, int iw, int cin, int cout, int kh, int kw,
                   int stride, int accumulate, void* stream);

/* Input gradient of an Inception block whose branches all start with a stride-1 1x1
 * convolution of the same input (the sum TF's AddN forms over the branches): one GEMM over the
 * concatenated reduction dx[rows][cin] (+)= sum_s dc_s[rows][cout_s] . W_s^T, with nseg <= 4
 * segments given as HOST arrays of device pointers / leading dims / channel offsets / widths.
 * W_s is HWIO 1x1 = [cin][cout_s]. */
int c2d_conv1x1_dgrad_multi(int nseg, const float* const* dcs, const int* ldcs, const int* coffs,
                            const float* const* ws, const int* couts, float* dx, int lddx,
                            int dxoff, int rows, int cin, int accumulate, void* stream);

/* Several INDEPENDENT convolutions in one call: the convolutions of one dependency level of an
 * Inception block (models/utils.py:133-136 runs the slim Inception graph; its branches do not
 * depend on each other).  No destination of one descriptor may overlap a source or destination
 * of another.  Same results as the single calls, bit for bit: when every problem is in the
 * small-problem domain (fp32, <= 16384 rows, fewer than 256 128x128 tiles) they run as ONE launch
 * (first stage on a single image: 30-250 workgroups each otherwise), else one launch each.
 * HOST array of 1..64 descriptors. */
typedef struct C2dConvDesc {
  const float* src; int ld_src; int off_src;   /* fwd: x          dgrad: dc                   */
  const float* weights;                        /* fwd: wt [taps][cout][cin]   dgrad: w (HWIO)  */
  const float* scale; const float* shift;      /* fwd only, may be NULL                        */
  float* dst; int ld_dst; int off_dst;         /* fwd: y          dgrad: dx                   */
  int n, ih, iw, cin, cout, kh, kw, stride;    /* as in c2d_conv_fwd / c2d_conv_dgrad          */
  int flag;                                    /* fwd: relu       dgrad: accumulate           */
} C2dConvDesc;
int c2d_conv_fwd_grouped(const C2dConvDesc* descs, int num, void* stream);
int c2d_conv_dgrad_grouped(const C2dConvDesc* descs, int num, void* stream);
/* bf16 storage forms (src / weights / dst hold bf16; see c2d_conv_fwd_bf16): the grouped launch
 * is igemm_small_group_kernel<*, 2> — the bf16 step's first stage. */
int c2d_conv_fwd_grouped_bf16(const C2dConvDesc* descs, int num, void* stream);
int c2d_conv_dgrad_grouped_bf16(const C2dConvDesc* descs, int num, void* stream);

/* Debug query: the kernel template instances launched by the calling thread's LAST convolution
 * entry point (c2d_conv_fwd / _dgrad / _wgrad / c2d_conv1x1_dgrad_multi, their _bf16 / _ws /
 * _grouped forms), spelled as rocprofv3 prints them and separated by ';', e.g.
 * "igemm_nt_kernel<0, 2, 2, 2, 1, 32, true, 4>".  Lets a parity test prove WHICH tile path
 * produced the result it compared with the oracle.  C2D_ERR_WORKSPACE if buf is too small. */
int c2d_debug_last_dispatch(char* buf, int len);

/* Balanced ("stream-K") forms of the three calls above.  `workspace` is a caller-owned device
 * buffer of at least c2d_conv_workspace_bytes() bytes that is ZERO when first used (the kernels
 * leave its counters zero again) and is not shared by launches that may run concurrently.
 * The launch then runs as one persistent workgroup per resident slot, each taking an equal share
 * of the (tile, K-slab) iterations; a tile cut by a share boundary is reduced, in a fixed order,
 * by whichever of its pieces finishes last, so results are bitwise reproducible and no CU idles
 * in a partial last round of tiles.  Same results as the plain forms up to fp32 summation order.
 * Returns C2D_ERR_WORKSPACE if the buffer is too small. */
long long c2d_conv_workspace_bytes(void);
int c2d_conv_fwd_ws(const float* x, int ldx, int xoff, const float* wt, const float* scale,
                    const float* shift, float* y, int ldy, int yoff, int n, int ih, int iw,
                    int cin, int cout, int kh, int kw, int stride, int relu, void* workspace,
                    long long workspace_bytes, void* stream);
int c2d_conv_dgrad_ws(const float* dc, int ldc, int coff, const float* w, float* dx, int lddx,
                      int dxoff, int n, int ih, int iw, int cin, int cout, int kh, int kw,
                      int stride, int accumulate, void* workspace, long long workspace_bytes,
                      void* stream);
int c2d_conv1x1_dgrad_multi_ws(int nseg, const float* const* dcs, const int* ldcs,
                               const int* coffs, const float* const* ws, const int* couts,
                               float* dx, int lddx, int dxoff, int rows, int cin, int accumulate,
                               void* workspace, long long workspace_bytes, void* stream);

/* bf16 storage / fp32 accumulate forms of the three convolution calls (BASELINE.json configs[2]
 * and [4]): activations, weights and outputs are bf16 (uint16 storage; leading dimensions and
 * offsets in ELEMENTS, multiples of 8, and so are the channel counts of the written tensor —
 * cout forward, cin for the input gradients: every lane loads and stores 16 bytes), BatchNorm
 * scale / shift stay fp32, products are
 * accumulated in fp32 on v_mfma_f32_32x32x16_bf16 and rounded to bf16 (nearest even) once, in
 * the epilogue.  The reference has no reduced-precision mode; tolerances are stated in
 * tests/test_gpu_bf16.py. */
int c2d_conv_fwd_bf16(const void* x, int ldx, int xoff, const void* wt, const float* scale,
                      const float* shift, void* y, int ldy, int yoff, int n, int ih, int iw,
                      int cin, int cout, int kh, int kw, int stride, int relu, void* stream);
int c2d_conv_dgrad_bf16(const void* dc, int ldc, int coff, const void* w, void* dx, int lddx,
                        int dxoff, int n, int ih, int iw, int cin, int cout, int kh, int kw,
                        int stride, int accumulate, void* stream);
int c2d_conv1x1_dgrad_multi_bf16(int nseg, const void* const* dcs, const int* ldcs,
                                 const int* coffs, const void* const* ws, const int* couts,
                                 void* dx, int lddx, int dxoff, int rows, int cin, int accumulate,
                                 void* stream);

/* The filter gradients of SEVERAL 1x1 / stride-1 convolutions of ONE input (the entry convolutions
 * of an Inception block: nets/inception_v2 via models/utils.py:165-167, gradients taken by
 * train/trainer.py:141-146) in one launch: dws[s][cin][couts[s]] += x^T . dcs[s], nseg <= 4, HOST
 * arrays of device pointers / leading dimensions / column offsets / widths.  The row splits are
 * shared by all outputs (a third of the split-K atomics of nseg separate c2d_conv_wgrad calls,
 * the x rows of a split fetched once); same sums up to the fp32 order of the split additions.
 * _bf16: x and dcs hold bf16, dws stay fp32. */
int c2d_conv1x1_wgrad_multi(const float* x, int ldx, int xoff, int nseg, const float* const* dcs,
                            const int* ldcs, const int* coffs, float* const* dws,
                            const int* couts, int rows, int cin, void* stream);
int c2d_conv1x1_wgrad_multi_bf16(const void* x, int ldx, int xoff, int nseg,
                                 const void* const* dcs, const int* ldcs, const int* coffs,
                                 float* const* dws, const int* couts, int rows, int cin,
                                 void* stream);

/* The filter gradients of `num` (<= 3) 3x3 / stride-1 / SAME convolutions over the same n per-ROI
 * maps of hw x hw (4 or 7) — the 3x3 layers of one Inception block — in ONE launch with shared row
 * splits (round 5: a third of the split-K atomics and of the ramps of three launches).  bf16
 * operands (channel counts multiples of 32; leading dimensions / offsets multiples of 8), fp32
 * atomics into dws[p][9][cin_p][cout_p].  C2D_ERR_UNSUPPORTED when a problem is not one the
 * nine-tap kernel takes: launch them with c2d_conv_wgrad_bf16 then. */
int c2d_conv3x3_wgrad_multi_bf16(int num, const void* const* xs, const int* ldxs, const int* xoffs,
                                 const void* const* dcs, const int* ldcs, const int* coffs,
                                 float* const* dws, const int* cins, const int* couts, int n,
                                 int hw, void* stream);

/* dw[kh*kw][cin][cout] += Conv2DBackpropFilter(x, dc)  (fp32 atomics over row splits; the
 * caller zero-fills dw once per step). */
int c2d_conv_wgrad(const float* x, int ldx, int xoff, const float* dc, int ldc, int coff,
                   float* dw, int n, int ih, int iw, int cin, int cout, int kh, int kw,
                   int stride, void* stream);

/* wt[t][j][i] = w[t][i][j] for t < taps (HWIO <-> per-tap transposed weights). */
int c2d_transpose_taps(const float* w, float* wt, int taps, int rows, int cols, void* stream);

/* ---------------------------------------------------------------------------------------
 * f32x9: the fp32 convolutions on the bf16 matrix pipe  (same reference ops as above — slim.conv2d
 * of models/utils.py:165-167 and its gradients, at the reference's own fp32 precision)
 *
 * Every fp32 value is the exact sum of three bf16 terms (truncation split: 8 + 8 + 8 significant
 * bits), every product of two terms is exact in fp32; nine v_mfma_f32_32x32x16_bf16 per 16 k
 * accumulate the fp32 product in fp32 at 288 instead of 512 matrix-pipe cycles.  The WEIGHT
 * operand is split once per optimiser step into three bf16 planes (c2d_split3_bf16); the
 * activation operand stays fp32 and is split in registers by the kernel.
 *
 * State: c2d_f32x9_bind registers (fp32 arena, plane arena) pairs — at most 64, the one piece of
 * process-wide state behind this header.  c2d_conv_fwd / c2d_conv1x1_fwd_multi / c2d_conv_dgrad* /
 * c2d_conv1x1_dgrad_multi* calls whose weight operand lies inside a bound arena (and whose GEMM
 * has at least 256 tiles of 128 x 128) take the f32x9 kernels; every other call is unchanged.
 * While at least one arena is bound (and the switch is on), c2d_conv_wgrad takes the nine-product
 * form as well for a 1x1 / stride-1 convolution over 8192 rows or more and for a 3x3 convolution
 * over 256 or more maps of 4x4 / 7x7 (stride 1) or 7x7 (stride 2) pixels (both operands are
 * activations there: the kernel splits them as it stages them).
 * The caller keeps the planes current (c2d_split3_bf16 after every update of the arena); bind /
 * unbind / enable are not to be called while GEMM calls are in flight on other threads.
 * ------------------------------------------------------------------------------------- */

/* planes[p * plane_stride + i] (bf16, p = 0 hi, 1 mid, 2 lo) = plane p of src[i], i < n:
 * hi = the top 16 bits of src[i], mid = the top 16 bits of src[i] - hi, lo = src[i] - hi - mid.
 * hi + mid + lo == src[i] exactly for |src[i]| >= 2^-110 (below: to within bf16's subnormal
 * spacing 2^-133); Inf / NaN: hi carries it, mid = lo = 0.  n and plane_stride multiples of 4,
 * src 16-byte and planes 8-byte aligned. */
int c2d_split3_bf16(const float* src, void* planes, long long plane_stride, long long n,
                    void* stream);
/* Binds the plane arena `planes` (3 x plane_stride bf16, plane_stride >= numel, a multiple of 8,
 * 6 * plane_stride < 2^31) to the fp32 arena [arena, arena + numel): element i of the arena has its
 * planes at planes[p * plane_stride + i].  Re-binding an arena replaces its entry.
 * C2D_ERR_WORKSPACE: all 64 slots taken. */
int c2d_f32x9_bind(const float* arena, long long numel, const void* planes, long long plane_stride);
/* Removes the binding of `arena` (NULL: all bindings). */
int c2d_f32x9_unbind(const float* arena);
/* on = 0: bound arenas are ignored (every call takes the fp32-MFMA kernels); returns the previous
 * setting (> 0: on).  For A/B measurements and tests. */
int c2d_f32x9_enable(int on);

/* Batched forms for the per-step refresh of every trainable layer in ONE launch each.  `desc`
 * is a DEVICE array of `num` records sorted by their begin field:
 *   transpose: struct { int64 src_off, dst_off; int32 taps, rows, cols, tile_begin; } — offsets
 *              in floats from src_base / dst_base, tile_begin = running count of 32x32 tiles;
 *   bn_fold  : struct { int64 gamma (or -1), beta, mean, var, scale, shift; int32 c, begin; } —
 *              gamma/beta index `vars`, mean/var index `stats`, scale/shift index `out`. */
int c2d_transpose_taps_batched(const void* desc, int num, int total_tiles, const float* src_base,
                               float* dst_base, void* stream);
/* c2d_transpose_taps_batched that also leaves the bf16 mirror of every transposed operand at the
 * same element offsets of dst_bf16 (bf16 networks read their forward weights from it): replaces a
 * cast pass over the whole derived-operand buffer after every optimiser step (round 5). */
int c2d_transpose_taps_batched_mirror(const void* desc, int num, int total_tiles,
                                      const float* src_base, float* dst_base, void* dst_bf16,
                                      void* stream);
int c2d_bn_fold_batched(const void* desc, int num, int total_channels, const float* vars,
                        const float* stats, float eps, float* out, void* stream);

/* Inference BatchNorm folded to an affine: scale = gamma*rsqrt(var+eps) (gamma NULL => 1),
 * shift = beta - mean*scale. */
int c2d_bn_fold(const float* gamma, const float* beta, const float* mean, const float* var,
                float eps, float* scale, float* shift, int c, void* stream);

/* Backward of y = relu(scale*c + shift): dc[rows][c] (dense) = dy*(y>0)*scale;
 * dbeta[c] += sum dy*(y>0); dgamma[c] += sum dy*(y>0)*(y-beta)/gamma (skipped when NULL). */
int c2d_bn_relu_bwd(const float* dy, int lddy, int dyoff, const float* y, int ldy, int yoff,
                    const float* scale, const float* beta, const float* gamma, float* dc,
                    float* dbeta, float* dgamma, int rows, int c, void* stream);

/* Atomic-free form of c2d_bn_relu_bwd for the training step: the same dc, but every row block
 * stores its partial sums to partials[block][2][c] (beta sums, then gamma sums; the gamma half is
 * zero when gamma is NULL) and ONE c2d_bn_partials_reduce_batched launch at the end of the
 * backward pass adds them, in block order, into the flat gradient buffer — the BatchNorm
 * gradients become bitwise reproducible and the kernel keeps >= 1000 row blocks in flight.
 * c2d_bn_relu_bwd_partial_blocks(rows, c) = number of row blocks (size the workspace with it).
 * desc: DEVICE array of struct { int64 ws_off, dbeta_off, dgamma_off (-1: none); int32 nblocks,
 * c, chunk_begin, wide; } sorted by chunk_begin = running count of ceil(c/64) channel chunks;
 * ws_off indexes `ws`, dbeta_off/dgamma_off index `grads` (floats).  wide = 0, or the width W of
 * the partial rows when the layer's c columns are a slice of rows [2][W] written for a whole
 * concat buffer (c2d_conv1x1_dgrad_multi_bn_relu): ws_off then addresses the layer's first
 * column in block 0's first half. */
int c2d_bn_relu_bwd_partial_blocks(int rows, int c);
int c2d_bn_relu_bwd_partial(const float* dy, int lddy, int dyoff, const float* y, int ldy,
                            int yoff, const float* scale, const float* beta, const float* gamma,
                            float* dc, float* partials, int rows, int c, void* stream);
int c2d_bn_partials_reduce_batched(const void* desc, int num, int total_chunks, const float* ws,
                                   float* grads, void* stream);
/* c2d_bn_relu_bwd_partial for the convolutions that write the network OUTPUT, with the backward of
 * c2d_spatial_mean_dropout_fwd folded in: instead of a dy buffer the kernel takes the gradient of
 * the averaged features, dy[r][c] = dmean[r / spatial][doff + c] * mask[r / spatial][mask_off + c]
 * / keep_prob / spatial (mask NULL: no dropout) — models/utils.py:183-188 (reduce_mean over the
 * map, slim.dropout) differentiated; the per-pixel gradient map is never materialised.
 * rows = ROIs * spatial (rows of y / dc); same row blocks as c2d_bn_relu_bwd_partial_blocks. */
/* An average-pooling branch commuted behind its 1x1 convolution (Inception `Branch_3`:
 * avg_pool2d 3x3 -> conv2d 1x1 -> BN -> ReLU, nets/inception_v2 via models/utils.py:165-167):
 * pooling and the 1x1 convolution are both linear and act on different axes, and the average of
 * a constant over the VALID cells of a SAME window is that constant, so
 *   relu(bn(conv(avgpool(x)))) == relu(avgpool(bn(conv(x))))
 * up to fp rounding — and the pool then runs over cout channels instead of cin (128 instead
 * of 1024).  Forward: c2d_conv_fwd(relu = 0) then c2d_avgpool3x3_relu_fwd; backward:
 * c2d_avgpool3x3_relu_bwd (dx (+)= avgpool_bwd(dy * (y > 0)), y = the pool's output), then
 * c2d_bn_bwd_partial on the convolution's output (BatchNorm backward without a ReLU: dc = dy *
 * scale, partial sums of dy and dy * (y - beta) / gamma as c2d_bn_relu_bwd_partial). */
int c2d_avgpool3x3_relu_fwd(const float* x, int ldx, int xoff, float* y, int ldy, int yoff, int n,
                            int ih, int iw, int c, int stride, void* stream);
int c2d_avgpool3x3_relu_bwd(const float* dy, int lddy, int dyoff, const float* y, int ldy,
                            int yoff, float* dx, int lddx, int dxoff, int n, int ih, int iw, int c,
                            int stride, int accumulate, void* stream);
int c2d_bn_bwd_partial(const float* dy, int lddy, int dyoff, const float* y, int ldy, int yoff,
                       const float* scale, const float* beta, const float* gamma, float* dc,
                       float* partials, int rows, int c, void* stream);
int c2d_bn_relu_bwd_partial_head(const float* dmean, int ldd, int doff, const uint8_t* mask,
                                 int mask_ld, int mask_off, int spatial, float keep_prob,
                                 const float* y, int ldy, int yoff, const float* scale,
                                 const float* beta, const float* gamma, float* dc,
                                 float* partials, int rows, int c, void* stream);

/* c2d_conv_dgrad fused with the c2d_bn_relu_bwd_partial of the layer that PRODUCED the
 * convolution's input (two consecutive slim.conv2d of an Inception branch, reference
 * nets/inception_v2 via models/utils.py:108-188; TF computes the ReLU and FusedBatchNorm gradients
 * between the two Conv2DBackpropInput ops): instead of storing dx and reading it back,
 * the GEMM epilogue applies the producer's ReLU mask and folded BN scale,
 *   dc_out[n*ih*iw][cin] (dense) = dx * (y > 0) * scale[cin],
 * and every row block of the launch stores its column sums of dz = dx * (y > 0) and of
 * dz * (y - beta) / gamma at partials[block][2][cin] (gamma NULL: the second half is zero) for
 * c2d_bn_partials_reduce_batched.  y: the producer's forward output (rows of ldy, offset yoff).
 * c2d_conv_dgrad_bn_relu_partial_blocks(elem_size 4|2, shape) = number of row blocks the launch
 * (stride 2: its four parity-class launches) writes; -1 for an unsupported shape. */
int c2d_conv_dgrad_bn_relu_partial_blocks(int elem_size, int n, int ih, int iw, int cin, int cout,
                                          int kh, int kw, int stride);
/* Several 1x1 / stride-1 convolutions of the SAME input as ONE GEMM (the entry convolutions of
 * the branches of an Inception block, nets/inception_v2 via models/utils.py:165-167): the input
 * is read once and the GEMM is N = sum(cout) wide; output s goes to its own buffer (rows of
 * ld_dst, column offset off_dst) through its own folded-BN scale / shift and ReLU flag.  Every
 * output element is the same K-ordered sum as c2d_conv_fwd computes: bitwise equal results.
 * wt: [cout][cin] rows (the layout c2d_conv_fwd takes for a 1x1 convolution); all wt of a call
 * must lie within 2 GB of each other (one allocation in practice). */
typedef struct C2dConvOut {
  const void* wt;
  const float* scale;
  const float* shift;
  void* dst;
  int ld_dst, off_dst;
  int cout;             /* multiple of 4 (bf16: of 8, as ld_dst and off_dst) */
  int relu;
} C2dConvOut;
int c2d_conv1x1_fwd_multi(const float* x, int ldx, int xoff, int nout, const C2dConvOut* outs,
                          int rows, int cin, void* stream);

/* The same fusion for the summed input gradient of an Inception block (c2d_conv1x1_dgrad_multi)
 * whose input is the concat buffer of the block in front: the columns belong to up to four
 * PRODUCERS (the last op of each branch of that block), in column order.  identity != 0: a
 * pooling branch (no BatchNorm / ReLU: its columns keep the plain gradient, sums stay zero).
 * accumulate != 0: dx already holds the other contributions to the block-input gradient (the
 * pooling branch of THIS block) and the mask / scale / sums apply to the complete sum, which is
 * why this launch has to be the last writer of dx.  partials[block][2][cin]; the producers'
 * halves are the column ranges of each row. */
typedef struct C2dBnProducer {
  const float* scale;   /* [width] folded BN scale (NULL when identity) */
  const float* beta;    /* [width] or NULL (no gamma sums) */
  const float* gamma;   /* [width] or NULL */
  int width;            /* columns of this producer (multiple of 4) */
  int identity;
} C2dBnProducer;
int c2d_conv1x1_dgrad_multi_bn_relu_partial_blocks(int nseg, const int* couts, int rows, int cin);
int c2d_conv1x1_dgrad_multi_bn_relu(int nseg, const float* const* dcs, const int* ldcs,
                                    const int* coffs, const float* const* ws, const int* couts,
                                    const float* y, int ldy, int yoff, int nprod,
                                    const C2dBnProducer* prods, float* dx, int lddx, int dxoff,
                                    float* partials, int rows, int cin, int accumulate,
                                    void* stream);
/* The same for bf16 storage (round 5; dcs / ws / y / dx bf16, partials fp32; every width, leading
 * dimension and offset a multiple of 8): the DMA-ring kernel's fused epilogue instance. */
int c2d_conv1x1_dgrad_multi_bn_relu_partial_blocks_bf16(int nseg, const int* couts, int rows,
                                                        int cin);
int c2d_conv1x1_dgrad_multi_bn_relu_bf16(int nseg, const void* const* dcs, const int* ldcs,
                                         const int* coffs, const void* const* ws, const int* couts,
                                         const void* y, int ldy, int yoff, int nprod,
                                         const C2dBnProducer* prods, void* dx, int lddx, int dxoff,
                                         float* partials, int rows, int cin, int accumulate,
                                         void* stream);
int c2d_conv_dgrad_bn_relu(const float* dc, int ldc, int coff, const float* w, const float* y,
                           int ldy, int yoff, const float* scale, const float* beta,
                           const float* gamma, float* dc_out, float* partials, int n, int ih,
                           int iw, int cin, int cout, int kh, int kw, int stride, void* stream);

/* out[j] += sum_rows x[row][xoff+j]  (bias gradients). */
int c2d_col_sum(const float* x, int ldx, int xoff, float* out, int rows, int ncols,
                void* stream);

/* 3x3 SAME pooling, stride 1 or 2 (slim.max_pool2d / slim.avg_pool2d inside Inception-V2).
 * mode 0 = max (argmax[rows_out][c] uint8 = ky*3+kx of the first maximum, may be NULL in
 * inference), mode 1 = average over the valid cells (TF AvgPool SAME divisor). */
int c2d_pool3x3_fwd(const float* x, int ldx, int xoff, float* y, int ldy, int yoff,
                    uint8_t* argmax, int n, int ih, int iw, int c, int stride, int mode,
                    void* stream);
int c2d_pool3x3_bwd(const float* dy, int lddy, int dyoff, const uint8_t* argmax, float* dx,
                    int lddx, int dxoff, int n, int ih, int iw, int c, int stride, int mode,
                    int accumulate, void* stream);

/* tf.reduce_mean over the spatial axis + slim.dropout (models/utils.py:169-174).
 * x [rows][spatial][c] -> y [rows][c]; mask [rows][c] uint8 {0,1} or NULL (no dropout). */
int c2d_spatial_mean_dropout_fwd(const float* x, float* y, const uint8_t* mask, int rows,
                                 int spatial, int c, float keep_prob, void* stream);
int c2d_spatial_mean_dropout_bwd(const float* dy, int lddy, int dyoff, float* dx,
                                 const uint8_t* mask, int rows, int spatial, int c,
                                 float keep_prob, void* stream);
/* Counter-based keep mask: mask[i] = u(seed, i) < keep_prob. */
int c2d_dropout_mask(uint8_t* mask, long long n, unsigned long long seed, float keep_prob,
                     void* stream);
/* Same with the seed read from device memory (seed_dev[0]) at execution time, so that a
 * captured hipGraph of the step draws a fresh mask on every replay. */
int c2d_dropout_mask_dev(uint8_t* mask, long long n, const long long* seed_dev, float keep_prob,
                         void* stream);

/* FasterRCNN preprocess (2/255)*x - 1 [third party, models/utils.py:127] fused with a pad of
 * the RGB image to 4 channels: image [pixels][3] -> out [pixels][4]. */
int c2d_preprocess_pad4(const float* image, float* out, long long pixels, void* stream);
/* SAME-padded im2col of a 4-channel image for the 7x7/2 stem:
 * out[n*oh*ow][kpad], column (ky*kw+kx)*4 + c, zero filled up to kpad (kpad % 16 == 0). */
int c2d_im2col4(const float* x, float* out, int n, int ih, int iw, int kh, int kw, int stride,
                int kpad, void* stream);

/* ---------------------------------------------------------------------------------------
 * MIDN / OICR / losses  (models/cap2det_model.py:53-109,274-330; models/utils.py:15-105)
 * logits buffers are [batch*n][ld] with one column block per head.
 * ------------------------------------------------------------------------------------- */

/* proba = softmax_r(mask*Lr - 1e10*(1-mask))*mask; class_logits = sum_r mask*Lc*proba;
 * scores = sigmoid(class_logits)*proba.   proba/scores [batch][n][C], class_logits [batch][C]. */
int c2d_midn_fwd(const float* logits, int ld, int off_r, int off_c,
                 const int32_t* num_proposals, float* proba, float* class_logits, float* scores,
                 int batch, int n, int num_classes, void* stream);
/* Given dloss/dclass_logits writes dloss/dLr and dloss/dLc into dlogits[batch*n][lddl] at the
 * same column offsets. */
int c2d_midn_bwd(const float* dclass_logits, const float* logits, int ld, int off_r, int off_c,
                 const int32_t* num_proposals, const float* proba, const float* class_logits,
                 float* dlogits, int lddl, int batch, int n, int num_classes, void* stream);
/* tf.nn.sigmoid_cross_entropy_with_logits, mean over n elements, times weight:
 * *loss += weight*mean; dlogits = weight*(sigmoid(x)-z)/n (either output may be NULL). */
int c2d_sigmoid_ce_fwd_bwd(const float* logits, const float* labels, int n, float weight,
                           float* loss, float* dlogits, void* stream);
/* masked_argmax over proposals for every class column of s0[batch*n][ld] (+off): stores the
 * winning proposal index idx[batch][C] and its box top_boxes[batch][C][4]. */
int c2d_oicr_select(const float* s0, int ld, int off, const int32_t* num_proposals,
                    const float* boxes, int32_t* idx, float* top_boxes, int batch, int n,
                    int num_classes, void* stream);
/* One OICR refinement loss (models/utils.py:64-103): pseudo labels from IoU(box, top box) >=
 * thr gated by the image labels, soft-label softmax CE over the C+1 columns of
 * scores[batch*n][ld] (+off), masked mean over proposals, mean over batch, times weight.
 * *loss += value; dscores[batch*n][lddl] (+doff) = gradient; softmax_out [batch*n][C+1] =
 * softmax(scores) (the next iteration's s0).  Outputs may be NULL. */
int c2d_oicr_loss_fwd_bwd(const float* scores, int ld, int off, const float* top_boxes,
                          const float* boxes, const float* labels,
                          const int32_t* num_proposals, float iou_threshold, float weight,
                          int batch, int n, int num_classes, float* loss, float* dscores,
                          int lddl, int doff, float* softmax_out, void* stream);
/* All `stages` OICR refinement losses of a step (the loop of models/cap2det_model.py:306-330:
 * select on s0, loss 1, select on softmax(scores 1)[..., 1:], loss 2, ...) in three launches
 * instead of two per stage — the selection of stage k needs only the SCORES of stage k - 1, which
 * the forward pass has left behind, not its loss.  Stage k reads / writes the C+1 columns at
 * off + k (C+1) / doff + k (C+1), adds its loss to loss[k], and leaves softmax_out[k][batch*n][C+1],
 * idx[k][batch][C], top_boxes[k][batch][C][4].  Bitwise the results of `stages` pairs of
 * c2d_oicr_select / c2d_oicr_loss_fwd_bwd (the loss scalars up to the order of their atomics). */
int c2d_oicr_refine_fwd_bwd(const float* scores, int ld, int off, int stages, const float* s0,
                            int s0_ld, int s0_off, const float* boxes, const float* labels,
                            const int32_t* num_proposals, float iou_threshold, float weight,
                            int batch, int n, int num_classes, float* loss, float* dscores,
                            int lddl, int doff, float* softmax_out, int32_t* idx,
                            float* top_boxes, void* stream);

/* ---------------------------------------------------------------------------------------
 * Caption -> label branch  (models/label_extractor.py)
 * ------------------------------------------------------------------------------------- */

/* `_match_labels` / ExtendMatch after the host resolved strings to ids: labels[b][c] = 1 iff
 * some ids[b][t] == c; ids >= num_classes are out-of-vocabulary. */
int c2d_labels_from_ids(const int32_t* ids, int batch, int num_tokens, int num_classes,
                        float* labels, void* stream);
/* TextClassifierMatchExtractor: embedding gather ([vocab_size+1][emb_dims], last row = OOV) ->
 * FC emb_dims->hidden (no activation) -> masked_maximum over tokens (mask = id != OOV) ->
 * ReLU -> FC hidden->C = logits[batch][C].  If labels != NULL also
 * labels = any(exact_labels>0) ? exact_labels : (sigmoid(logits) > label_threshold). */
int c2d_text_classifier_fwd(const int32_t* ids, int batch, int num_tokens,
                            const float* embedding, int vocab_size, int emb_dims,
                            const float* w1, const float* b1, int hidden_units, const float* w2,
                            const float* b2, int num_classes, const float* exact_labels,
                            float label_threshold, float* logits, float* labels, void* stream);
/* Same computation with a caller-owned workspace of c2d_text_classifier_workspace_bytes(batch,
 * hidden_units) bytes (the hidden layer): grid over (hidden units / 64, captions) with the
 * workgroup's W1 columns held in LDS — the form the detector's training step uses (the
 * one-workgroup form above needs ~1 ms per 60-token caption, this one ~30 us). */
long long c2d_text_classifier_workspace_bytes(int batch, int hidden_units);
int c2d_text_classifier_fwd_ws(const int32_t* ids, int batch, int num_tokens,
                               const float* embedding, int vocab_size, int emb_dims,
                               const float* w1, const float* b1, int hidden_units,
                               const float* w2, const float* b2, int num_classes,
                               const float* exact_labels, float label_threshold, float* logits,
                               float* labels, void* workspace, long long workspace_bytes,
                               void* stream);

/* WordVectorMatchExtractor (models/label_extractor.py:251-328): cosine similarity of every
 * token with every class name (class_ids = vocabulary ids of the class names) in the embedding
 * table ([vocab_size+1][emb_dims], last row = OOV), masked max-pool over non-OOV tokens, one-hot
 * of the best class (zeros when every token is OOV); rows with an exact match keep
 * exact_labels.  num_tokens*num_classes floats must fit 64 KiB of LDS. */
int c2d_word_vector_match_fwd(const int32_t* ids, int batch, int num_tokens,
                              const float* embedding, int vocab_size, int emb_dims,
                              const int32_t* class_ids, int num_classes,
                              const float* exact_labels, float* labels, void* stream);

/* ---------------------------------------------------------------------------------------
 * Optimiser  (train/trainer.py:55-61,104-146; core/training_utils.py:45-50)
 * ------------------------------------------------------------------------------------- */

/* g' = mult*(grad_scale*g + l2*w); acc += g'^2; w -= lr*g'/sqrt(acc)   (tf.train.Adagrad;
 * grad_scale = 1/world_size after the RCCL sum all-reduce). */
int c2d_adagrad_step(float* w, const float* g, float* acc, long long n, float lr, float l2,
                     float mult, float grad_scale, void* stream);
/* *out += 0.5*weight*sum(w^2)  (slim l2_regularizer). */
int c2d_l2_loss(const float* w, long long n, float weight, float* out, void* stream);
/* *out += weight*sum(|w|)  (slim l1_regularizer, core/training_utils.py:167-168). */
int c2d_l1_loss(const float* w, long long n, float weight, float* out, void* stream);

/* Step-start zeroing: clears up to C2D_ZERO_MAX device byte ranges (16-byte aligned, lengths a
 * multiple of 16) in ONE launch — the flat gradient bucket, the loss scalars, the gradient map
 * the ROI-crop backward accumulates into.  TensorFlow's tf.gradients (train/trainer.py:96-103)
 * has no such step; the accumulating kernels here (`+=` semantics above) need it.  The struct is
 * passed by pointer on the HOST; first_chunk is filled by the callee. */
#define C2D_ZERO_MAX 8
typedef struct C2dZeroRanges {
  void* ptr[C2D_ZERO_MAX];
  long long bytes[C2D_ZERO_MAX];
  int first_chunk[C2D_ZERO_MAX];
  int num;
} C2dZeroRanges;
int c2d_zero_ranges(const C2dZeroRanges* ranges, void* stream);
/* *out = x[0] + ... + x[n-1] (n <= 4096): `total_loss = tf.add_n(losses)` of
 * train/trainer.py:55-61 over the loss scalars kept in one device vector. */
int c2d_sum_small(const float* x, int n, float* out, void* stream);

/* General optimiser step for the reference branches no shipped config takes:
 *   g' = m*(grad_scale*g + l2*w + l1*sign(w)),  m = mult * (col_mult ? col_mult[i % ld] : 1);
 *   elements with m <= 0 are frozen (gradient multipliers <= 0, train/trainer.py:104-125; the
 *   per-column form serves the five heads fused in one [D][ld] buffer when their multipliers
 *   differ).  lr_dev != NULL: the learning rate is read from device memory (a captured hipGraph
 *   then follows a continuous tf.train.exponential_decay, train/trainer.py:76-82). */
int c2d_adagrad_step_ex(float* w, const float* g, float* acc, long long n, float lr,
                        const float* lr_dev, float l1, float l2, float mult, float grad_scale,
                        const float* col_mult, int ld, void* stream);
/* c2d_adagrad_step over up to 8 segments [offsets[i], ends[i]) of the flat value / gradient /
 * accumulator buffers in ONE launch (HOST arrays; segment i with its own multiplier and L2 weight:
 * the runs of variables train/trainer.py:104-125 gives one gradient multiplier), bitwise the
 * separate calls.  values_bf16 != NULL: the bf16 mirror of every updated value is written as well
 * (same element offsets; bf16 networks read their input-gradient weights from it). */
int c2d_adagrad_step_multi(float* values, const float* grads, float* accum, int num_segments,
                           const long long* offsets, const long long* ends, const float* mults,
                           const float* l2s, float lr, float grad_scale, void* values_bf16,
                           void* stream);

/* The reference's other optimisers (core/training_utils.py:14-71 `build_optimizer`: sgd, momentum,
 * adam, rmsprop; every shipped config uses adagrad), TensorFlow 1.x update rules on the gradient
 * g' of c2d_adagrad_step_ex (same grad_scale / l1 / l2 / mult / col_mult / lr_dev meaning):
 *   C2D_OPT_SGD       w -= lr g'
 *   C2D_OPT_MOMENTUM  s0 = p0 s0 + g';  w -= lr s0   (flags & 1, use_nesterov: w -= lr (g' + p0 s0))
 *   C2D_OPT_ADAM      s0 = p0 s0 + (1-p0) g';  s1 = p1 s1 + (1-p1) g'^2;  w -= p3 s0 / (sqrt(s1) + p2)
 *                     with p0 = beta1, p1 = beta2, p2 = epsilon and p3 = lr sqrt(1-beta2^t)/(1-beta1^t)
 *                     computed by the caller for step t = 1, 2, ...
 *   C2D_OPT_RMSPROP   s0 = p0 s0 + (1-p0) g'^2  [flags & 2, centered: s2 = p0 s2 + (1-p0) g'];
 *                     s1 = p1 s1 + lr g' / sqrt(s0 [- s2^2] + p2);  w -= s1
 *                     with p0 = decay, p1 = momentum, p2 = epsilon (TF initialises s0 to ONES).
 * s0 / s1 / s2: caller-owned slot buffers of n floats (NULL where the rule has none). */
#define C2D_OPT_SGD 0
#define C2D_OPT_MOMENTUM 1
#define C2D_OPT_ADAM 2
#define C2D_OPT_RMSPROP 3
int c2d_optimizer_step(int kind, int flags, float* w, const float* g, float* s0, float* s1,
                       float* s2, long long n, float lr, const float* lr_dev, float p0, float p1,
                       float p2, float p3, float l1, float l2, float mult, float grad_scale,
                       const float* col_mult, int ld, void* stream);

/* One variable = a [rows][cols] window (row stride ld) at `offset` of the flat buffers. */
typedef struct C2dClipDesc {
  long long offset;
  int rows, cols, ld;
  float l1, l2, mult;
} C2dClipDesc;
/* tf.contrib.training.clip_gradient_norms (train/trainer.py:132-136): per variable, in place,
 *   g' = mult*(grad_scale*g + l2*w + l1*sign(w));  g' *= max_norm / max(||g'||_2, max_norm).
 * desc: DEVICE array of `num` descriptors.  Afterwards c2d_adagrad_step runs with l2 = 0,
 * mult = 1, grad_scale = 1.  Deterministic (fixed summation order). */
int c2d_clip_gradient_norms(float* grads, const float* values, const C2dClipDesc* desc, int num,
                            float grad_scale, float max_norm, void* stream);

/* Atomic-free, bitwise-reproducible form of c2d_roi_crop_pool_bwd (same semantics: adds into
 * dfeat).  Needs a caller-owned device workspace of at least
 * c2d_roi_crop_pool_bwd_workspace_bytes(...) bytes (sampling tables, per-row cell lists, the
 * strip kernel's work plan and its partial rows; feature rows wider than 32 columns are walked
 * as strips of at most 32 columns, each with lists of its own);
 * returns C2D_ERR_WORKSPACE if it is too small, C2D_ERR_UNSUPPORTED unless
 * c2d_roi_crop_pool_bwd_ws_supported(...) != 0: pool_k == 2, the pooled map at most 16x16,
 * depth % 16 == 0 and 2 <= wf <= 255 (use c2d_roi_crop_pool_bwd then).  Maps of the reference's
 * 1000-px training images (readers/cap2det_reader.py:143-172: up to ~100 feature columns) are
 * inside that range; the value returned is the channel chunk (= workgroup size) the strip kernel
 * will use for (wf, depth). */
int c2d_roi_crop_pool_bwd_ws_supported(int wf, int depth, int crop, int pool_k, int pool_s);
/* The same question for a whole call (round 5): every condition under which
 * c2d_roi_crop_pool_bwd_ws / _prepare / _run return C2D_ERR_UNSUPPORTED, so that a caller decides
 * ONCE per shape which backward it runs instead of meeting the error inside a training step.  On
 * top of the per-map rules: pooled gradient < 2 GiB (elem_size 4 = fp32, 2 = bf16; fp32 7x7x576
 * cells: fewer than 19,022 boxes), row lists < 2^31 entries, batch * hf * (strips per feature
 * row) <= 4095.  Returns the channel chunk (> 0) or 0. */
int c2d_roi_crop_pool_bwd_ws_shape_supported(int batch, int hf, int wf, int depth, int num_boxes,
                                             int crop, int pool_k, int pool_s, int elem_size);
long long c2d_roi_crop_pool_bwd_workspace_bytes(int batch, int hf, int wf, int depth,
                                                int num_boxes, int crop, int pool_k, int pool_s);
int c2d_roi_crop_pool_bwd_ws(const float* dout, const uint8_t* argmax, const float* boxes,
                             const int32_t* box_ind, float* dfeat, int batch, int hf, int wf,
                             int depth, int num_boxes, int crop, int pool_k, int pool_s,
                             void* workspace, long long workspace_bytes, void* stream);
/* The two halves of c2d_roi_crop_pool_bwd_ws over the same workspace: `prepare` builds what
 * depends on the boxes alone (sampling tables, per-row cell lists) and may run on another stream
 * during the forward pass; `run` (after it) accumulates dout into dfeat. */
int c2d_roi_crop_pool_bwd_prepare(const float* boxes, const int32_t* box_ind, int batch, int hf,
                                  int wf, int depth, int num_boxes, int crop, int pool_k,
                                  int pool_s, void* workspace, long long workspace_bytes,
                                  void* stream);
int c2d_roi_crop_pool_bwd_run(const float* dout, const uint8_t* argmax, const float* boxes,
                              const int32_t* box_ind, float* dfeat, int batch, int hf, int wf,
                              int depth, int num_boxes, int crop, int pool_k, int pool_s,
                              void* workspace, long long workspace_bytes, void* stream);
int c2d_roi_crop_pool_bwd_run_bf16(const void* dout, const uint8_t* argmax, const float* boxes,
                                   const int32_t* box_ind, float* dfeat, int batch, int hf, int wf,
                                   int depth, int num_boxes, int crop, int pool_k, int pool_s,
                                   void* workspace, long long workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * bf16 storage forms of the activation-side kernels (BASELINE.json configs[2] / [4]: "bf16
 * storage / fp32 accumulate").  Same arguments and semantics as the fp32 calls of the same
 * name; the tensors marked void* hold bf16 (uint16) elements, strides / offsets stay in
 * ELEMENTS, all arithmetic is fp32 with ONE round-to-nearest-even at the store.  Statistics,
 * BatchNorm vectors, filter gradients, the first-stage feature map and its gradient stay fp32.
 * ------------------------------------------------------------------------------------- */
int c2d_cast_bf16(const float* src, void* dst, long long n, void* stream);   /* n % 4 == 0 */
/* the other way (exact): the bf16 first stage's feature map in front of the fp32 ROI crop */
int c2d_cast_f32(const void* src, float* dst, long long n, void* stream);    /* n % 4 == 0 */
int c2d_roi_crop_pool_fwd_bf16(const float* feat, const float* boxes, const int32_t* box_ind,
                               void* out, uint8_t* argmax, int batch, int hf, int wf, int depth,
                               int num_boxes, int crop, int pool_k, int pool_s, void* stream);
int c2d_roi_crop_pool_bwd_ws_bf16(const void* dout, const uint8_t* argmax, const float* boxes,
                                  const int32_t* box_ind, float* dfeat, int batch, int hf, int wf,
                                  int depth, int num_boxes, int crop, int pool_k, int pool_s,
                                  void* workspace, long long workspace_bytes, void* stream);
int c2d_pool3x3_fwd_bf16(const void* x, int ldx, int xoff, void* y, int ldy, int yoff,
                         uint8_t* argmax, int n, int ih, int iw, int c, int stride, int mode,
                         void* stream);
int c2d_pool3x3_bwd_bf16(const void* dy, int lddy, int dyoff, const uint8_t* argmax, void* dx,
                         int lddx, int dxoff, int n, int ih, int iw, int c, int stride, int mode,
                         int accumulate, void* stream);
int c2d_bn_relu_bwd_partial_bf16(const void* dy, int lddy, int dyoff, const void* y, int ldy,
                                 int yoff, const float* scale, const float* beta,
                                 const float* gamma, void* dc, float* partials, int rows, int c,
                                 void* stream);
int c2d_avgpool3x3_relu_fwd_bf16(const void* x, int ldx, int xoff, void* y, int ldy, int yoff,
                                 int n, int ih, int iw, int c, int stride, void* stream);
int c2d_avgpool3x3_relu_bwd_bf16(const void* dy, int lddy, int dyoff, const void* y, int ldy,
                                 int yoff, void* dx, int lddx, int dxoff, int n, int ih, int iw,
                                 int c, int stride, int accumulate, void* stream);
int c2d_bn_bwd_partial_bf16(const void* dy, int lddy, int dyoff, const void* y, int ldy, int yoff,
                            const float* scale, const float* beta, const float* gamma, void* dc,
                            float* partials, int rows, int c, void* stream);
int c2d_bn_relu_bwd_partial_head_bf16(const float* dmean, int ldd, int doff, const uint8_t* mask,
                                      int mask_ld, int mask_off, int spatial, float keep_prob,
                                      const void* y, int ldy, int yoff, const float* scale,
                                      const float* beta, const float* gamma, void* dc,
                                      float* partials, int rows, int c, void* stream);
int c2d_conv1x1_fwd_multi_bf16(const void* x, int ldx, int xoff, int nout, const C2dConvOut* outs,
                               int rows, int cin, void* stream);
/* (ldy and yoff multiples of 8: the fused epilogue reads the producer's y in 16-byte chunks) */
int c2d_conv_dgrad_bn_relu_bf16(const void* dc, int ldc, int coff, const void* w, const void* y,
                                int ldy, int yoff, const float* scale, const float* beta,
                                const float* gamma, void* dc_out, float* partials, int n, int ih,
                                int iw, int cin, int cout, int kh, int kw, int stride,
                                void* stream);
int c2d_spatial_mean_dropout_fwd_bf16(const void* x, float* y, const uint8_t* mask, int rows,
                                      int spatial, int c, float keep_prob, void* stream);
int c2d_spatial_mean_dropout_bwd_bf16(const float* dy, int lddy, int dyoff, void* dx,
                                      const uint8_t* mask, int rows, int spatial, int c,
                                      float keep_prob, void* stream);
/* x, dc bf16 -> dw fp32 (operands widened as they are staged; fp32 MFMA accumulation). */
int c2d_conv_wgrad_bf16(const void* x, int ldx, int xoff, const void* dc, int ldc, int coff,
                        float* dw, int n, int ih, int iw, int cin, int cout, int kh, int kw,
                        int stride, void* stream);

/* Split-K slabs instead of atomics for the bf16 filter gradients.  Global float atomics run at
 * 1.3 TB/s chip-wide, plain stores at 6 TB/s; at bf16 MFMA rates the atomics of
 * c2d_conv_wgrad_bf16 are 40 % of its time.  c2d_conv_wgrad_bf16_partial makes every K split store
 * its own fp32 slab (partials[split][kh*kw][cin][cout], `partial_floats` floats available;
 * C2D_ERR_WORKSPACE if too few, C2D_ERR_UNSUPPORTED when the operands do not qualify for the bf16
 * MFMA kernels); c2d_conv_wgrad_bf16_splits returns the number of slabs that launch writes (or a
 * negative C2D_ERR_*); c2d_wgrad_reduce_batched adds the slabs of `num` layers, in split order
 * (bitwise reproducible), into their filter gradients: ONE launch per backward pass.
 * desc: DEVICE array; total_chunks = sum over layers of ceil(numel / 1024). */
int c2d_conv_wgrad_bf16_splits(int ldx, int xoff, int ldc, int coff, int n, int ih, int iw,
                               int cin, int cout, int kh, int kw, int stride);
int c2d_conv_wgrad_bf16_partial(const void* x, int ldx, int xoff, const void* dc, int ldc,
                                int coff, float* partials, long long partial_floats, int n,
                                int ih, int iw, int cin, int cout, int kh, int kw, int stride,
                                void* stream);
/* The same for fp32 operands (c2d_conv_wgrad's kernels with slab stores). */
int c2d_conv_wgrad_splits(int ldx, int xoff, int ldc, int coff, int n, int ih, int iw, int cin,
                          int cout, int kh, int kw, int stride);
int c2d_conv_wgrad_partial(const float* x, int ldx, int xoff, const float* dc, int ldc, int coff,
                           float* partials, long long partial_floats, int n, int ih, int iw,
                           int cin, int cout, int kh, int kw, int stride, void* stream);
typedef struct C2dWgradReduceDesc {
  long long ws_off;   /* first float of split 0's slab in `workspace` */
  long long dw_off;   /* first float of the filter gradient in `grads` */
  int numel;          /* kh*kw*cin*cout */
  int splits;
  int begin;          /* first 1024-element chunk (= workgroup) of this layer */
  int pad;
} C2dWgradReduceDesc;
int c2d_wgrad_reduce_batched(const C2dWgradReduceDesc* desc, int num, int total_chunks,
                             const float* workspace, float* grads, void* stream);

/* ---------------------------------------------------------------------------------------
 * Inference post-processing
 * ------------------------------------------------------------------------------------- */

/* Multi-class non-max suppression over SHARED boxes: replaces
 * object_detection.core.post_processing.batch_multiclass_non_max_suppression as called by
 * core/builder.py:57-65 from models/cap2det_model.py:137-140.  boxes [batch][n][4]
 * (ymin,xmin,ymax,xmax), scores [batch][n][ld] with the class columns at [off, off+num_classes).
 * Per class: candidates with score > score_thresh by decreasing score (ties: lower index), a
 * candidate is dropped when its IoU with a kept box is > iou_thresh, at most max_size_per_class
 * kept; all classes merged by decreasing score (ties: class order, then selection order), first
 * max_total_size written, rest zero padded.  out_classes is 1-based (float), as the reference
 * returns `nmsed_classes + 1`.  n <= 8192.  Workspace from the query below. */
long long c2d_multiclass_nms_workspace_bytes(int batch, int n, int num_classes,
                                             int max_size_per_class);
int c2d_multiclass_nms(const float* boxes, const float* scores, int ld, int off, int batch, int n,
                       int num_classes, float score_thresh, float iou_thresh,
                       int max_size_per_class, int max_total_size, int32_t* num_detections,
                       float* out_boxes, float* out_scores, float* out_classes, void* workspace,
                       long long workspace_bytes, void* stream);

/* out[rows][c1-1] = softmax(logits[r][off .. off+c1))[1:]  (tf.nn.softmax(...)[:, :, 1:],
 * models/cap2det_model.py:135). */
int c2d_softmax_drop_background(const float* logits, int ld, int off, int rows,
                                int num_classes_plus_one, float* out, void* stream);

/* Running mean over the evaluation scales (tf.stack + tf.reduce_mean,
 * models/cap2det_model.py:262-267): dst[rows][cols] (dense) = init ? src : dst + src, with src
 * rows of stride ld at column off; then dst /= count. */
int c2d_scores_accumulate(float* dst, const float* src, int ld, int off, int rows, int cols,
                          int init, void* stream);
int c2d_scores_divide(float* x, long long n, float divisor, void* stream);

/* tf.image.resize_images(BILINEAR, align_corners=False) of TF1 (legacy scaler, no half-pixel
 * offset) on one NHWC fp32 image: core/imgproc.py:348-351, and the reader's resizer (f1). */
int c2d_resize_bilinear(const float* in, int ih, int iw, int channels, float* out, int oh, int ow,
                        void* stream);

/* ---------------------------------------------------------------------------------------
 * Input pipeline: host-side record / proto / JPEG decoding
 * (io_native.cpp, plain C++: entropy decoding is serial) and GPU-side pixel work.
 * All pointers in this block are HOST pointers unless stated otherwise.
 * ------------------------------------------------------------------------------------- */

/* CRC-32C and TFRecord's masked form rotr(crc, 15) + 0xa282ead8. */
unsigned int c2d_crc32c(const void* data, long long n);
unsigned int c2d_masked_crc32c(const void* data, long long n);
/* Next record of a TFRecord byte buffer (what tf.data.TFRecordDataset yields,
 * readers/cap2det_reader.py:217-218): returns the position after the record, 0 at a clean end,
 * C2D_ERR_DATA on truncation / CRC mismatch (verify_crc != 0). */
long long c2d_tfrecord_next(const uint8_t* buf, long long size, long long pos,
                            long long* payload_off, long long* payload_len, int verify_crc);
/* Frames one payload (n + 16 bytes written); used to write fixtures. */
long long c2d_tfrecord_frame(const uint8_t* payload, long long n, uint8_t* out);
/* tf.parse_single_example (readers/cap2det_reader.py:40-59) of one serialized tf.Example for
 * `nkeys` feature names: kinds[k] = 0 absent, 1 bytes, 2 float, 3 int64; counts[k] values
 * starting at starts[k] in the arena of that kind; bytes values are (offset, length) pairs into
 * `rec`.  C2D_ERR_WORKSPACE when an arena is too small. */
int c2d_example_parse(const uint8_t* rec, long long len, const char* const* keys, int nkeys,
                      int* kinds, long long* counts, long long* starts, float* floats,
                      long long float_cap, long long* ints, long long int_cap, long long* spans,
                      long long span_cap);
/* TensorFlow's Hash64 (seed 0xDECAFCAFFE): tf.strings.to_hash_bucket(s, k) = hash % k, the
 * shard filter of readers/cap2det_reader.py:201-211. */
unsigned long long c2d_tf_hash64(const void* data, long long n);
/* Huffman-coded 8-bit JPEG (baseline, extended sequential, progressive; one or several scans;
 * restart intervals; 4:4:4 / 4:2:2 / 4:4:0 / 4:2:0 / grayscale) -> RGB u8 [height][width][3]
 * with libjpeg's default decompression choices (islow IDCT, fancy upsampling), i.e.
 * tf.image.decode_jpeg(channels=3) (readers/cap2det_reader.py:91-92).  Truncated or corrupted
 * streams: C2D_ERR_DATA; arithmetic-coded / lossless / 12-bit / CMYK: C2D_ERR_UNSUPPORTED. */
int c2d_jpeg_info(const uint8_t* data, long long n, int* height, int* width, int* components);
long long c2d_jpeg_workspace_bytes(int height, int width);
int c2d_jpeg_decode_rgb(const uint8_t* data, long long n, uint8_t* out, int height, int width,
                        void* workspace, long long workspace_bytes);

/* DEVICE pointers: one decoded RGB u8 image [ih][iw][3] -> optional tf.image.flip_left_right
 * (core/preprocess.py:48-52) -> TF1 legacy-bilinear resize to oh x ow (core/builder.py:70-128
 * resizers) -> written into the top-left corner of a ph x pw fp32 canvas, zeros elsewhere (the
 * zero padding of `padded_batch`, readers/cap2det_reader.py:220-247). */
int c2d_image_resize_pad_u8(const uint8_t* image, int ih, int iw, int flip_left_right,
                            float* canvas, int oh, int ow, int ph, int pw, void* stream);

/* ---------------------------------------------------------------------------------------
 * Text-classifier training (models/text_model.py:31-129 over
 * models/label_extractor.py:353-421 with is_training = True).  The two fully connected layers
 * run on c2d_conv_fwd / c2d_conv_dgrad / c2d_conv_wgrad (1x1 geometry); these do the rest.
 * ------------------------------------------------------------------------------------- */

/* x[rows][ld] = embedding[ids[r]] (ids outside [0, vocab_size) -> the OOV row vocab_size),
 * zero padded from emb_dims to ld columns: tf.nn.embedding_lookup, :384-390. */
int c2d_embedding_gather(const int32_t* ids, long long rows, const float* embedding,
                         int vocab_size, int emb_dims, int ld, float* x, void* stream);
/* hidden[b][h] = dropout(relu(masked_maximum_t(pre[b][t][h]; id_t != OOV))) (:397-410,
 * core/utils.py:63-79; keep_mask NULL = evaluation) and its gradient w.r.t. pre with
 * TensorFlow's tie rules (reduce_max / reduce_min share the gradient among tied extrema). */
int c2d_text_pool_fwd(const float* pre, const int32_t* ids, int batch, int num_tokens,
                      int hidden_units, int vocab_size, const uint8_t* keep_mask, float keep_prob,
                      float* hidden, void* stream);
int c2d_text_pool_bwd(const float* dhidden, const float* pre, const int32_t* ids, int batch,
                      int num_tokens, int hidden_units, int vocab_size, const uint8_t* keep_mask,
                      float keep_prob, float* dpre, void* stream);

/* CUs this process may count on when it sizes "one round of workgroups" launches (split counts of
 * the filter gradients, the small-problem threshold): 256 by default, fewer for a data-parallel rank
 * whose RCCL channel kernels hold CUs under the backward pass (train_wsod.sh:46-88 runs one
 * process per GPU; bench.py --available-cus).  8..256; process-wide, set before the step loop. */
int c2d_set_available_cus(int cus);
int c2d_get_available_cus(void);

/* Diagnostic (tools/cu_withhold.py): occupies `workgroups` CUs (one 160-KiB-LDS workgroup each) on
 * `stream` for `microseconds` — the rehearsal of a collective's channel kernels on a one-GPU box. */
int c2d_debug_hold_cus(int workgroups, long long microseconds, void* stream);

/* ---------------------------------------------------------------------------------------
 * Step plan: the call list of one training step, recorded once, replayed by ONE call
 * (replaces the per-step session.run of slim.learning.train, train/trainer.py:141-146: the
 * reference hands its whole step to the TensorFlow runtime; here the step is ~140 of the entry points
 * above on four streams, which cap2det_amd/step_plan.py records while a step runs eagerly)
 *
 * A plan is an ordered list of nodes: CALL (an entry point of this header, by name, with its
 * argument words), RECORD (plan event e on a stream) and WAIT (a stream waits for plan event e).
 * Argument words are 64-bit: pointers and integers as they are, floats in the low 32 bits.
 * kinds[i]: 0 = the word is a constant; 1 = a pointer into the tensor bound to slots[i] (the word is
 * the byte offset from its base); 2 = the word itself comes from binding slots[i] (a scalar).
 * Host arrays passed by pointer (descriptor tables) must outlive the plan unchanged.
 * Not thread-safe per plan; replays of one plan must not overlap.
 * ------------------------------------------------------------------------------------- */
long long c2d_plan_create(void);                 /* handle (pass as void*), 0 on failure */
/* dst[0..bytes) = src[0..bytes) (device to device, a copy kernel) on `stream`: the hand-over of the
 * look-ahead's first-stage prefix to the step that consumes it.  16-byte aligned, bytes % 16 == 0. */
int c2d_copy_bytes(const void* src, void* dst, long long bytes, void* stream);
int c2d_plan_destroy(void* plan);
/* C2D_ERR_UNSUPPORTED: no entry point `name` returning int; C2D_ERR_INVALID_ARG: wrong nargs. */
int c2d_plan_add_call(void* plan, const char* name, int nargs, const long long* vals,
                      const uint8_t* kinds, const int* slots);
int c2d_plan_add_event_record(void* plan, int event, void* stream);
int c2d_plan_add_stream_wait(void* plan, void* stream, int event);
/* Closes the plan: every other stream of its RECORD / WAIT nodes records a plan event behind its last
 * node, which main_stream waits for at the start of the NEXT replay; a stream that does not begin
 * with a wait for an event of main_stream starts behind main_stream's position at the replay call.
 * The first replay after anything else ran on those streams needs them joined by the caller. */
int c2d_plan_finish(void* plan, void* main_stream);
int c2d_plan_size(void* plan);                   /* nodes, boundary nodes included */
/* Issues the plan: bindings[s] = the base pointer (kind 1) or the word (kind 2) of slot s.
 * Stops at the first failing node: its C2D_ERR_* code is returned, its index stored in
 * *failed_node (may be NULL). */
int c2d_plan_replay(void* plan, const long long* bindings, int num_bindings, int* failed_node);

#ifdef __cplusplus
}
#endif
#endif /* CAP2DET_HIP_H_ */
