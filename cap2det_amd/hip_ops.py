"""Thin typed wrappers: torch CUDA tensors -> raw device pointers -> C-ABI (include/cap2det_hip.h).

PyTorch is plumbing only (allocation, streams); every arithmetic op on the hot path runs in
the hand-written HIP kernels behind these calls.  No fallback: a missing library raises.
"""
import ctypes

import torch

from cap2det_amd import _lib


def _p(t):
  if t is None:
    return None
  assert t.is_cuda and t.is_contiguous(), "C-ABI buffers must be contiguous device tensors"
  return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_cur_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
  """Raw hipStream_t of torch's current stream on the current device.  `torch.cuda.current_stream()`
  builds a Stream object per call and was a third of the host time of a training step (240
  launches); the two private queries it wraps cost a tenth of that."""
  if _raw_stream is None or _cur_device is None:
    return torch.cuda.current_stream().cuda_stream
  return _raw_stream(_cur_device())


def _f32(*ts):
  for t in ts:
    assert t is None or t.dtype == torch.float32


# -- ROI crop ---------------------------------------------------------------------------

def crop_and_resize(feat, boxes, box_ind, crop):
  b, hf, wf, d = feat.shape
  _f32(feat, boxes)
  out = torch.zeros(boxes.shape[0], crop, crop, d, device=feat.device, dtype=torch.float32)
  _lib.call("c2d_crop_and_resize_fwd", _p(feat), _p(boxes), _p(box_ind), _p(out), b, hf, wf, d,
            boxes.shape[0], crop, _stream())
  return out


def roi_crop_pool_fwd(feat, boxes, box_ind, crop, pool_k, pool_s, out=None, argmax=None,
                      want_argmax=True):
  b, hf, wf, d = feat.shape
  r = boxes.shape[0]
  p = (crop - pool_k) // pool_s + 1
  if out is None:
    out = torch.zeros(r, p, p, d, device=feat.device, dtype=torch.float32)
  if argmax is None and want_argmax:
    argmax = torch.zeros(r, p, p, d, device=feat.device, dtype=torch.uint8)
  fn = "c2d_roi_crop_pool_fwd_bf16" if out.dtype == torch.bfloat16 else "c2d_roi_crop_pool_fwd"
  _lib.call(fn, _p(feat), _p(boxes), _p(box_ind), _p(out), _p(argmax), b, hf, wf, d, r, crop,
            pool_k, pool_s, _stream())
  return out, argmax


def roi_crop_pool_bwd(dout, argmax, boxes, box_ind, dfeat, crop, pool_k, pool_s):
  """The atomic form (fp32 gradients only: the bf16 storage mode runs the row-owner form, which
  covers maps up to 255 columns wide)."""
  if dout.dtype != torch.float32:
    raise NotImplementedError("c2d_roi_crop_pool_bwd takes fp32 gradients; bf16 gradients need the "
                              "workspace form (feature maps up to 255 columns wide)")
  b, hf, wf, d = dfeat.shape
  _lib.call("c2d_roi_crop_pool_bwd", _p(dout), _p(argmax), _p(boxes), _p(box_ind), _p(dfeat), b,
            hf, wf, d, boxes.shape[0], crop, pool_k, pool_s, _stream())
  return dfeat


def roi_crop_pool_bwd_ws_supported(wf, depth, crop, pool_k, pool_s):
  """Channel chunk of the atomic-free ROI-crop backward for this map width, 0 = unsupported."""
  return int(_lib.load().c2d_roi_crop_pool_bwd_ws_supported(wf, depth, crop, pool_k, pool_s))


def roi_crop_pool_bwd_ws_shape_supported(batch, hf, wf, depth, num_boxes, crop, pool_k, pool_s,
                                         elem_size):
  """Channel chunk when the atomic-free backward covers the whole CALL (map width, pooled-gradient
  bytes, list and row-table limits), 0 = it would return C2D_ERR_UNSUPPORTED."""
  return int(_lib.load().c2d_roi_crop_pool_bwd_ws_shape_supported(batch, hf, wf, depth, num_boxes,
                                                                  crop, pool_k, pool_s, elem_size))


def roi_crop_pool_bwd_workspace_bytes(batch, hf, wf, depth, num_boxes, crop, pool_k, pool_s):
  return int(_lib.load().c2d_roi_crop_pool_bwd_workspace_bytes(batch, hf, wf, depth, num_boxes,
                                                               crop, pool_k, pool_s))


def roi_crop_pool_bwd_ws(dout, argmax, boxes, box_ind, dfeat, crop, pool_k, pool_s, workspace):
  """Atomic-free, bitwise-reproducible ROI-crop backward (workspace: uint8 device tensor)."""
  b, hf, wf, d = dfeat.shape
  fn = "c2d_roi_crop_pool_bwd_ws_bf16" if dout.dtype == torch.bfloat16 else "c2d_roi_crop_pool_bwd_ws"
  _lib.call(fn, _p(dout), _p(argmax), _p(boxes), _p(box_ind), _p(dfeat), b, hf, wf, d,
            boxes.shape[0], crop, pool_k, pool_s, _p(workspace), workspace.numel(), _stream())
  return dfeat


def roi_crop_pool_bwd_prepare(boxes, box_ind, batch, hf, wf, depth, crop, pool_k, pool_s, workspace):
  """Box-dependent half of roi_crop_pool_bwd_ws (any stream, e.g. during the forward pass)."""
  _lib.call("c2d_roi_crop_pool_bwd_prepare", _p(boxes), _p(box_ind), batch, hf, wf, depth,
            boxes.shape[0], crop, pool_k, pool_s, _p(workspace), workspace.numel(), _stream())


def roi_crop_pool_bwd_run(dout, argmax, boxes, box_ind, dfeat, crop, pool_k, pool_s, workspace):
  """Accumulation half of roi_crop_pool_bwd_ws; needs roi_crop_pool_bwd_prepare on this workspace."""
  b, hf, wf, d = dfeat.shape
  fn = ("c2d_roi_crop_pool_bwd_run_bf16" if dout.dtype == torch.bfloat16
        else "c2d_roi_crop_pool_bwd_run")
  _lib.call(fn, _p(dout), _p(argmax), _p(boxes), _p(box_ind), _p(dfeat), b, hf, wf, d,
            boxes.shape[0], crop, pool_k, pool_s, _p(workspace), workspace.numel(), _stream())
  return dfeat


# -- convolution ------------------------------------------------------------------------

_conv_ws = None   # (tensor, nbytes): workspace of the balanced (stream-K) convolution forms


def conv_workspace_bytes():
  return int(_lib.load().c2d_conv_workspace_bytes())


def set_conv_workspace(ws):
  """ws: ZERO-filled uint8 device tensor of >= conv_workspace_bytes() bytes (or None to go back
  to the one-tile-per-block launches).  One workspace serves all convolutions of one stream."""
  global _conv_ws
  if ws is None:
    _conv_ws = None
    return
  assert ws.is_cuda and ws.dtype == torch.uint8 and ws.numel() >= conv_workspace_bytes()
  _conv_ws = (ws, ws.numel())


def conv_fwd(x, ldx, xoff, wt, scale, shift, y, ldy, yoff, n, ih, iw, cin, cout, kh, kw, stride,
             relu):
  if x.dtype == torch.bfloat16:
    assert wt.dtype == torch.bfloat16 and y.dtype == torch.bfloat16
    _lib.call("c2d_conv_fwd_bf16", _p(x), ldx, xoff, _p(wt), _p(scale), _p(shift), _p(y), ldy,
              yoff, n, ih, iw, cin, cout, kh, kw, stride, int(relu), _stream())
    return
  if _conv_ws is not None:
    _lib.call("c2d_conv_fwd_ws", _p(x), ldx, xoff, _p(wt), _p(scale), _p(shift), _p(y), ldy, yoff,
              n, ih, iw, cin, cout, kh, kw, stride, int(relu), _p(_conv_ws[0]), _conv_ws[1],
              _stream())
    return
  _lib.call("c2d_conv_fwd", _p(x), ldx, xoff, _p(wt), _p(scale), _p(shift), _p(y), ldy, yoff, n,
            ih, iw, cin, cout, kh, kw, stride, int(relu), _stream())


class ConvDesc(ctypes.Structure):
  """C2dConvDesc of include/cap2det_hip.h."""
  _fields_ = [("src", ctypes.c_void_p), ("ld_src", ctypes.c_int), ("off_src", ctypes.c_int),
              ("weights", ctypes.c_void_p), ("scale", ctypes.c_void_p), ("shift", ctypes.c_void_p),
              ("dst", ctypes.c_void_p), ("ld_dst", ctypes.c_int), ("off_dst", ctypes.c_int),
              ("n", ctypes.c_int), ("ih", ctypes.c_int), ("iw", ctypes.c_int),
              ("cin", ctypes.c_int), ("cout", ctypes.c_int), ("kh", ctypes.c_int),
              ("kw", ctypes.c_int), ("stride", ctypes.c_int), ("flag", ctypes.c_int)]


def conv_group(calls):
  """Packs [(x, ldx, xoff, wt, scale, shift, y, ldy, yoff, n, ih, iw, cin, cout, kh, kw, stride,
  relu), ...] (the arguments of conv_fwd) into a reusable descriptor array for conv_fwd_grouped.
  The tensors must stay alive (and in place) for as long as the group is used."""
  arr = (ConvDesc * len(calls))()
  flops = 0.0
  for d, c in zip(arr, calls):
    (x, ldx, xoff, wt, scale, shift, y, ldy, yoff, n, ih, iw, cin, cout, kh, kw, stride, relu) = c
    assert x.dtype == y.dtype == wt.dtype and x.dtype == calls[0][0].dtype
    d.src, d.ld_src, d.off_src, d.weights = _p(x), ldx, xoff, _p(wt)
    d.scale, d.shift = _p(scale), _p(shift)
    d.dst, d.ld_dst, d.off_dst = _p(y), ldy, yoff
    d.n, d.ih, d.iw, d.cin, d.cout, d.kh, d.kw, d.stride, d.flag = (n, ih, iw, cin, cout, kh, kw,
                                                                    stride, int(relu))
    flops += 2.0 * n * (-(-ih // stride)) * (-(-iw // stride)) * cin * cout * kh * kw
  return arr, len(calls), flops, calls[0][0].dtype


def conv_fwd_grouped(group):
  """Independent convolutions of one dependency level in one call (c2d_conv_fwd_grouped)."""
  fn = "c2d_conv_fwd_grouped_bf16" if group[3] == torch.bfloat16 else "c2d_conv_fwd_grouped"
  _lib.call(fn, ctypes.cast(group[0], ctypes.c_void_p), group[1], _stream())


def conv_dgrad_group(calls):
  """Packs [(dc, ldc, coff, w, dx, lddx, dxoff, n, ih, iw, cin, cout, kh, kw, stride, accumulate),
  ...] (the arguments of conv_dgrad) into a descriptor array for conv_dgrad_grouped."""
  arr = (ConvDesc * len(calls))()
  for d, c in zip(arr, calls):
    (dc, ldc, coff, w, dx, lddx, dxoff, n, ih, iw, cin, cout, kh, kw, stride, accumulate) = c
    assert dc.dtype == dx.dtype == w.dtype and dc.dtype == calls[0][0].dtype
    d.src, d.ld_src, d.off_src, d.weights = _p(dc), ldc, coff, _p(w)
    d.scale, d.shift = None, None
    d.dst, d.ld_dst, d.off_dst = _p(dx), lddx, dxoff
    d.n, d.ih, d.iw, d.cin, d.cout, d.kh, d.kw, d.stride, d.flag = (n, ih, iw, cin, cout, kh, kw,
                                                                    stride, int(accumulate))
  return arr, len(calls), 0.0, calls[0][0].dtype


def conv_dgrad_grouped(group):
  """Independent input gradients in one call (c2d_conv_dgrad_grouped / _bf16)."""
  fn = "c2d_conv_dgrad_grouped_bf16" if group[3] == torch.bfloat16 else "c2d_conv_dgrad_grouped"
  _lib.call(fn, ctypes.cast(group[0], ctypes.c_void_p), group[1], _stream())


def conv_dgrad(dc, ldc, coff, w, dx, lddx, dxoff, n, ih, iw, cin, cout, kh, kw, stride,
               accumulate):
  if dc.dtype == torch.bfloat16:
    assert w.dtype == torch.bfloat16 and dx.dtype == torch.bfloat16
    _lib.call("c2d_conv_dgrad_bf16", _p(dc), ldc, coff, _p(w), _p(dx), lddx, dxoff, n, ih, iw, cin,
              cout, kh, kw, stride, int(accumulate), _stream())
    return
  if _conv_ws is not None:
    _lib.call("c2d_conv_dgrad_ws", _p(dc), ldc, coff, _p(w), _p(dx), lddx, dxoff, n, ih, iw, cin,
              cout, kh, kw, stride, int(accumulate), _p(_conv_ws[0]), _conv_ws[1], _stream())
    return
  _lib.call("c2d_conv_dgrad", _p(dc), ldc, coff, _p(w), _p(dx), lddx, dxoff, n, ih, iw, cin, cout,
            kh, kw, stride, int(accumulate), _stream())


def conv1x1_dgrad_multi(dcs, ldcs, coffs, ws, couts, dx, lddx, dxoff, rows, cin, accumulate):
  """dx (+)= sum_s dc_s . W_s^T over up to 4 (dc, W) segments (see include/cap2det_hip.h)."""
  n = len(dcs)
  ptrs = ctypes.c_void_p * n
  ints = ctypes.c_int * n
  if dcs[0].dtype == torch.bfloat16:
    _lib.call("c2d_conv1x1_dgrad_multi_bf16", n, ptrs(*[_p(t) for t in dcs]), ints(*ldcs),
              ints(*coffs), ptrs(*[_p(t) for t in ws]), ints(*couts), _p(dx), lddx, dxoff, rows,
              cin, int(accumulate), _stream())
    return
  if _conv_ws is not None:
    _lib.call("c2d_conv1x1_dgrad_multi_ws", n, ptrs(*[_p(t) for t in dcs]), ints(*ldcs),
              ints(*coffs), ptrs(*[_p(t) for t in ws]), ints(*couts), _p(dx), lddx, dxoff, rows,
              cin, int(accumulate), _p(_conv_ws[0]), _conv_ws[1], _stream())
    return
  _lib.call("c2d_conv1x1_dgrad_multi", n, ptrs(*[_p(t) for t in dcs]), ints(*ldcs), ints(*coffs),
            ptrs(*[_p(t) for t in ws]), ints(*couts), _p(dx), lddx, dxoff, rows, cin,
            int(accumulate), _stream())


def conv1x1_wgrad_multi(x, ldx, xoff, dcs, ldcs, coffs, dws, couts, rows, cin):
  """dws[s] += x^T . dcs[s] for up to 4 1x1 / stride-1 convolutions of ONE input in one launch
  (c2d_conv1x1_wgrad_multi(_bf16), see include/cap2det_hip.h)."""
  n = len(dcs)
  ptrs = ctypes.c_void_p * n
  ints = ctypes.c_int * n
  assert all(t.dtype == x.dtype for t in dcs) and all(t.dtype == torch.float32 for t in dws)
  fn = "c2d_conv1x1_wgrad_multi_bf16" if x.dtype == torch.bfloat16 else "c2d_conv1x1_wgrad_multi"
  _lib.call(fn, _p(x), ldx, xoff, n, ptrs(*[_p(t) for t in dcs]), ints(*ldcs), ints(*coffs),
            ptrs(*[_p(t) for t in dws]), ints(*couts), rows, cin, _stream())


def conv3x3_wgrad_multi(probs, n, hw):
  """Nine-tap filter gradients of several 3x3 / stride-1 convolutions over the same per-ROI maps in
  ONE launch (bf16 operands).  probs: [(x, ldx, xoff, dc, ldc, coff, dw, cin, cout), ...] (<= 3).
  Returns False when the library declines the group (C2D_ERR_UNSUPPORTED): launch them one by one."""
  k = len(probs)
  assert all(p[0].dtype == torch.bfloat16 and p[3].dtype == torch.bfloat16 and p[6].dtype == torch.float32
             for p in probs)
  xs = (ctypes.c_void_p * k)(*[_p(p[0]) for p in probs])
  dcs = (ctypes.c_void_p * k)(*[_p(p[3]) for p in probs])
  dws = (ctypes.c_void_p * k)(*[_p(p[6]) for p in probs])
  ints = lambda i: (ctypes.c_int * k)(*[int(p[i]) for p in probs])
  rc = _lib.load().c2d_conv3x3_wgrad_multi_bf16(k, xs, ints(1), ints(2), dcs, ints(4), ints(5), dws,
                                                ints(7), ints(8), n, hw, _stream())
  if rc == -2:         # C2D_ERR_UNSUPPORTED
    return False
  _lib.check(rc, "c2d_conv3x3_wgrad_multi_bf16")
  return True


def conv_wgrad(x, ldx, xoff, dc, ldc, coff, dw, n, ih, iw, cin, cout, kh, kw, stride):
  fn = "c2d_conv_wgrad_bf16" if x.dtype == torch.bfloat16 else "c2d_conv_wgrad"
  assert x.dtype == dc.dtype and dw.dtype == torch.float32
  _lib.call(fn, _p(x), ldx, xoff, _p(dc), ldc, coff, _p(dw), n, ih, iw, cin, cout, kh, kw, stride,
            _stream())


def transpose_taps(w, wt, taps, rows, cols):
  _lib.call("c2d_transpose_taps", _p(w), _p(wt), taps, rows, cols, _stream())


def transpose_taps_batched(desc, num, total_tiles, src_base, dst_base):
  _lib.call("c2d_transpose_taps_batched", _p(desc), num, total_tiles, _p(src_base), _p(dst_base),
            _stream())


def transpose_taps_batched_mirror(desc, num, total_tiles, src_base, dst_base, dst_bf16):
  """transpose_taps_batched that also writes the bf16 mirror of every transposed operand."""
  assert dst_bf16.dtype == torch.bfloat16 and dst_bf16.numel() == dst_base.numel()
  _lib.call("c2d_transpose_taps_batched_mirror", _p(desc), num, total_tiles, _p(src_base),
            _p(dst_base), _p(dst_bf16), _stream())


def bn_fold_batched(desc, num, total_channels, vars_base, stats_base, eps, out_base):
  _lib.call("c2d_bn_fold_batched", _p(desc), num, total_channels, _p(vars_base), _p(stats_base),
            float(eps), _p(out_base), _stream())


def bn_fold(gamma, beta, mean, var, eps, scale, shift):
  _lib.call("c2d_bn_fold", _p(gamma), _p(beta), _p(mean), _p(var), float(eps), _p(scale),
            _p(shift), beta.numel(), _stream())


def bn_relu_bwd(dy, lddy, dyoff, y, ldy, yoff, scale, beta, gamma, dc, dbeta, dgamma, rows, c):
  _lib.call("c2d_bn_relu_bwd", _p(dy), lddy, dyoff, _p(y), ldy, yoff, _p(scale), _p(beta),
            _p(gamma), _p(dc), _p(dbeta), _p(dgamma), rows, c, _stream())


def conv_dgrad_bn_relu_blocks(dtype, n, ih, iw, cin, cout, kh, kw, stride):
  """Row blocks of partial sums c2d_conv_dgrad_bn_relu writes for this shape (-1: unsupported)."""
  es = 2 if dtype == torch.bfloat16 else 4
  return int(_lib.load().c2d_conv_dgrad_bn_relu_partial_blocks(es, n, ih, iw, cin, cout, kh, kw, stride))


def conv_dgrad_bn_relu(dc, ldc, coff, w, y, ldy, yoff, scale, beta, gamma, dc_out, partials, n, ih,
                       iw, cin, cout, kh, kw, stride):
  """Input gradient of a convolution + BN/ReLU backward of the layer that produced its input."""
  fn = "c2d_conv_dgrad_bn_relu_bf16" if dc.dtype == torch.bfloat16 else "c2d_conv_dgrad_bn_relu"
  assert w.dtype == dc.dtype and y.dtype == dc.dtype and dc_out.dtype == dc.dtype
  _lib.call(fn, _p(dc), ldc, coff, _p(w), _p(y), ldy, yoff, _p(scale), _p(beta),
            _p(gamma) if gamma is not None else None, _p(dc_out), _p(partials), n, ih, iw, cin,
            cout, kh, kw, stride, _stream())


def avgpool3x3_relu_fwd(x, ldx, xoff, y, ldy, yoff, n, ih, iw, c, stride):
  fn = "c2d_avgpool3x3_relu_fwd_bf16" if x.dtype == torch.bfloat16 else "c2d_avgpool3x3_relu_fwd"
  _lib.call(fn, _p(x), ldx, xoff, _p(y), ldy, yoff, n, ih, iw, c, stride, _stream())


def avgpool3x3_relu_bwd(dy, lddy, dyoff, y, ldy, yoff, dx, lddx, dxoff, n, ih, iw, c, stride, accumulate):
  fn = "c2d_avgpool3x3_relu_bwd_bf16" if dy.dtype == torch.bfloat16 else "c2d_avgpool3x3_relu_bwd"
  _lib.call(fn, _p(dy), lddy, dyoff, _p(y), ldy, yoff, _p(dx), lddx, dxoff, n, ih, iw, c, stride,
            int(accumulate), _stream())


def bn_bwd_partial(dy, lddy, dyoff, y, ldy, yoff, scale, beta, gamma, dc, partials, rows, c):
  """BatchNorm backward without a ReLU (see c2d_bn_bwd_partial)."""
  fn = "c2d_bn_bwd_partial_bf16" if dy.dtype == torch.bfloat16 else "c2d_bn_bwd_partial"
  _lib.call(fn, _p(dy), lddy, dyoff, _p(y), ldy, yoff, _p(scale), _p(beta),
            _p(gamma) if gamma is not None else None, _p(dc), _p(partials), rows, c, _stream())


def bn_relu_bwd_partial_head(dmean, ldd, doff, mask, mask_ld, mask_off, spatial, keep_prob, y, ldy,
                             yoff, scale, beta, gamma, dc, partials, rows, c):
  """bn_relu_bwd_partial with dy derived from the gradient of the averaged features."""
  fn = "c2d_bn_relu_bwd_partial_head_bf16" if y.dtype == torch.bfloat16 else "c2d_bn_relu_bwd_partial_head"
  assert dmean.dtype == torch.float32 and dc.dtype == y.dtype
  _lib.call(fn, _p(dmean), ldd, doff, _p(mask) if mask is not None else None, mask_ld, mask_off,
            spatial, float(keep_prob), _p(y), ldy, yoff, _p(scale), _p(beta),
            _p(gamma) if gamma is not None else None, _p(dc), _p(partials), rows, c, _stream())


class ConvOut(ctypes.Structure):
  """C2dConvOut of include/cap2det_hip.h."""
  _fields_ = [("wt", ctypes.c_void_p), ("scale", ctypes.c_void_p), ("shift", ctypes.c_void_p),
              ("dst", ctypes.c_void_p), ("ld_dst", ctypes.c_int), ("off_dst", ctypes.c_int),
              ("cout", ctypes.c_int), ("relu", ctypes.c_int)]


def conv_outs(outs):
  """[(wt, scale, shift, dst, ld_dst, off_dst, cout, relu)] -> (C2dConvOut array, count, dtype)."""
  arr = (ConvOut * len(outs))()
  for d, (wt, scale, shift, dst, ld, off, cout, relu) in zip(arr, outs):
    d.wt, d.scale, d.shift, d.dst = _p(wt), _p(scale), _p(shift), _p(dst)
    d.ld_dst, d.off_dst, d.cout, d.relu = ld, off, cout, int(relu)
  return arr, len(outs), outs[0][0].dtype


def conv1x1_fwd_multi(x, ldx, xoff, outs, rows, cin):
  """Several 1x1 convolutions of one input as one GEMM (c2d_conv1x1_fwd_multi); outs: conv_outs()."""
  arr, n, dtype = outs
  assert x.dtype == dtype
  fn = "c2d_conv1x1_fwd_multi_bf16" if dtype == torch.bfloat16 else "c2d_conv1x1_fwd_multi"
  _lib.call(fn, _p(x), ldx, xoff, n, arr, rows, cin, _stream())


class BnProducer(ctypes.Structure):
  """C2dBnProducer of include/cap2det_hip.h."""
  _fields_ = [("scale", ctypes.c_void_p), ("beta", ctypes.c_void_p), ("gamma", ctypes.c_void_p),
              ("width", ctypes.c_int), ("identity", ctypes.c_int)]


def conv1x1_dgrad_multi_bn_relu_blocks(couts, rows, cin, dtype=torch.float32):
  arr = (ctypes.c_int * len(couts))(*couts)
  lib = _lib.load()
  fn = (lib.c2d_conv1x1_dgrad_multi_bn_relu_partial_blocks_bf16 if dtype == torch.bfloat16
        else lib.c2d_conv1x1_dgrad_multi_bn_relu_partial_blocks)
  return int(fn(len(couts), arr, rows, cin))


def bn_producers(prods):
  """[(scale, beta, gamma, width) or (None, None, None, width) for a pooling branch] -> array."""
  arr = (BnProducer * len(prods))()
  for i, (sc, be, ga, width) in enumerate(prods):
    arr[i].scale = _p(sc) if sc is not None else None
    arr[i].beta = _p(be) if be is not None else None
    arr[i].gamma = _p(ga) if ga is not None else None
    arr[i].width = width
    arr[i].identity = 1 if sc is None else 0
  return arr


def conv1x1_dgrad_multi_bn_relu(dcs, ldcs, coffs, ws, couts, y, ldy, yoff, prods, dx, lddx, dxoff,
                                partials, rows, cin, accumulate):
  """c2d_conv1x1_dgrad_multi as the last writer of a block-input gradient, fused with the
  BN/ReLU backward of the producers of the block input (fp32 or bf16 storage; the partial sums
  are fp32).  prods: bn_producers(...)."""
  n = len(dcs)
  low = dcs[0].dtype == torch.bfloat16
  assert all(t.dtype == dcs[0].dtype for t in list(dcs) + list(ws) + [y, dx]) and partials.dtype == torch.float32
  pa = (ctypes.c_void_p * n)(*[_p(t) for t in dcs])
  pw = (ctypes.c_void_p * n)(*[_p(t) for t in ws])
  il = (ctypes.c_int * n)(*ldcs); io = (ctypes.c_int * n)(*coffs); ic = (ctypes.c_int * n)(*couts)
  _lib.call("c2d_conv1x1_dgrad_multi_bn_relu_bf16" if low else "c2d_conv1x1_dgrad_multi_bn_relu",
            n, pa, il, io, pw, ic, _p(y), ldy, yoff, len(prods),
            prods, _p(dx), lddx, dxoff, _p(partials), rows, cin, int(accumulate), _stream())


def bn_relu_bwd_partial_blocks(rows, c):
  return int(_lib.load().c2d_bn_relu_bwd_partial_blocks(rows, c))


def bn_relu_bwd_partial(dy, lddy, dyoff, y, ldy, yoff, scale, beta, gamma, dc, partials, rows, c):
  fn = "c2d_bn_relu_bwd_partial_bf16" if dy.dtype == torch.bfloat16 else "c2d_bn_relu_bwd_partial"
  assert dy.dtype == y.dtype == dc.dtype
  _lib.call(fn, _p(dy), lddy, dyoff, _p(y), ldy, yoff, _p(scale), _p(beta), _p(gamma), _p(dc),
            _p(partials), rows, c, _stream())


def bn_partials_reduce_batched(desc, num, total_chunks, ws, grads):
  _lib.call("c2d_bn_partials_reduce_batched", _p(desc), num, total_chunks, _p(ws), _p(grads),
            _stream())


def cast_bf16(src, dst):
  assert src.dtype == torch.float32 and dst.dtype == torch.bfloat16 and src.numel() == dst.numel()
  _lib.call("c2d_cast_bf16", _p(src), _p(dst), src.numel(), _stream())


def copy_bytes(src, dst):
  assert src.numel() * src.element_size() == dst.numel() * dst.element_size()
  _lib.call("c2d_copy_bytes", _p(src), _p(dst), src.numel() * src.element_size(), _stream())


def cast_f32(src, dst):
  assert src.dtype == torch.bfloat16 and dst.dtype == torch.float32 and src.numel() == dst.numel()
  _lib.call("c2d_cast_f32", _p(src), _p(dst), src.numel(), _stream())


def col_sum(x, ldx, xoff, out, rows, ncols):
  _lib.call("c2d_col_sum", _p(x), ldx, xoff, _p(out), rows, ncols, _stream())


def pool3x3_fwd(x, ldx, xoff, y, ldy, yoff, argmax, n, ih, iw, c, stride, mode):
  fn = "c2d_pool3x3_fwd_bf16" if x.dtype == torch.bfloat16 else "c2d_pool3x3_fwd"
  assert x.dtype == y.dtype
  _lib.call(fn, _p(x), ldx, xoff, _p(y), ldy, yoff, _p(argmax), n, ih, iw, c, stride, mode,
            _stream())


def pool3x3_bwd(dy, lddy, dyoff, argmax, dx, lddx, dxoff, n, ih, iw, c, stride, mode, accumulate):
  fn = "c2d_pool3x3_bwd_bf16" if dy.dtype == torch.bfloat16 else "c2d_pool3x3_bwd"
  assert dy.dtype == dx.dtype
  _lib.call(fn, _p(dy), lddy, dyoff, _p(argmax), _p(dx), lddx, dxoff, n, ih, iw, c, stride, mode,
            int(accumulate), _stream())


def spatial_mean_dropout_fwd(x, y, mask, rows, spatial, c, keep_prob):
  fn = ("c2d_spatial_mean_dropout_fwd_bf16" if x.dtype == torch.bfloat16
        else "c2d_spatial_mean_dropout_fwd")
  _lib.call(fn, _p(x), _p(y), _p(mask), rows, spatial, c, float(keep_prob), _stream())


def spatial_mean_dropout_bwd(dy, lddy, dyoff, dx, mask, rows, spatial, c, keep_prob):
  fn = ("c2d_spatial_mean_dropout_bwd_bf16" if dx.dtype == torch.bfloat16
        else "c2d_spatial_mean_dropout_bwd")
  _lib.call(fn, _p(dy), lddy, dyoff, _p(dx), _p(mask), rows, spatial, c, float(keep_prob),
            _stream())


def dropout_mask(mask, seed, keep_prob):
  """seed: an int, or a step_plan.Sym (the per-step dropout key of a recorded step)."""
  from cap2det_amd.step_plan import sym_like
  _lib.call_sym("c2d_dropout_mask", _p(mask), mask.numel(), sym_like(seed, int(seed) & 0xFFFFFFFFFFFFFFFF),
                float(keep_prob), _stream())


def dropout_mask_dev(mask, seed_dev, keep_prob):
  assert seed_dev.dtype == torch.int64
  _lib.call("c2d_dropout_mask_dev", _p(mask), mask.numel(), _p(seed_dev), float(keep_prob),
            _stream())


def preprocess_pad4(image, out):
  _lib.call("c2d_preprocess_pad4", _p(image), _p(out), image.numel() // 3, _stream())


def im2col4(x, out, n, ih, iw, kh, kw, stride, kpad):
  _lib.call("c2d_im2col4", _p(x), _p(out), n, ih, iw, kh, kw, stride, kpad, _stream())


class ZeroRanges(ctypes.Structure):
  """C2dZeroRanges of include/cap2det_hip.h."""
  _fields_ = [("ptr", ctypes.c_void_p * 8), ("bytes", ctypes.c_longlong * 8),
              ("first_chunk", ctypes.c_int * 8), ("num", ctypes.c_int)]


def zero_ranges(tensors):
  """Zeroes up to 8 contiguous device tensors (16-byte aligned, byte sizes multiples of 16) in
  one launch (c2d_zero_ranges)."""
  r = ZeroRanges()
  r.num = len(tensors)
  for i, t in enumerate(tensors):
    assert t.is_contiguous()
    r.ptr[i] = _p(t)
    r.bytes[i] = t.numel() * t.element_size()
  _lib.call("c2d_zero_ranges", ctypes.byref(r), _stream())


def sum_small(x, out):
  _lib.call("c2d_sum_small", _p(x), x.numel(), _p(out), _stream())


# -- heads / losses ---------------------------------------------------------------------

def midn_fwd(logits, ld, off_r, off_c, num_proposals, proba, class_logits, scores, batch, n, c):
  _lib.call("c2d_midn_fwd", _p(logits), ld, off_r, off_c, _p(num_proposals), _p(proba),
            _p(class_logits), _p(scores), batch, n, c, _stream())


def midn_bwd(dclass_logits, logits, ld, off_r, off_c, num_proposals, proba, class_logits, dlogits,
             lddl, batch, n, c):
  _lib.call("c2d_midn_bwd", _p(dclass_logits), _p(logits), ld, off_r, off_c, _p(num_proposals),
            _p(proba), _p(class_logits), _p(dlogits), lddl, batch, n, c, _stream())


def sigmoid_ce_fwd_bwd(logits, labels, weight, loss, dlogits):
  _lib.call("c2d_sigmoid_ce_fwd_bwd", _p(logits), _p(labels), logits.numel(), float(weight),
            _p(loss), _p(dlogits), _stream())


def oicr_select(s0, ld, off, num_proposals, boxes, idx, top_boxes, batch, n, c):
  _lib.call("c2d_oicr_select", _p(s0), ld, off, _p(num_proposals), _p(boxes), _p(idx),
            _p(top_boxes), batch, n, c, _stream())


def oicr_loss_fwd_bwd(scores, ld, off, top_boxes, boxes, labels, num_proposals, iou_threshold,
                      weight, batch, n, c, loss, dscores, lddl, doff, softmax_out):
  _lib.call("c2d_oicr_loss_fwd_bwd", _p(scores), ld, off, _p(top_boxes), _p(boxes), _p(labels),
            _p(num_proposals), float(iou_threshold), float(weight), batch, n, c, _p(loss),
            _p(dscores), lddl, doff, _p(softmax_out), _stream())


def oicr_refine_fwd_bwd(scores, ld, off, stages, s0, s0_ld, s0_off, boxes, labels, num_proposals,
                        iou_threshold, weight, batch, n, c, loss, dscores, lddl, doff, softmax_out,
                        idx, top_boxes):
  """All OICR stages in three launches (c2d_oicr_refine_fwd_bwd): `loss` [stages], `softmax_out`
  [stages, batch * n, c + 1], `idx` [stages, batch, c], `top_boxes` [stages, batch, c, 4]."""
  assert softmax_out.is_contiguous() and softmax_out.numel() == stages * batch * n * (c + 1)
  assert idx.numel() == stages * batch * c and top_boxes.numel() == stages * batch * c * 4
  _lib.call("c2d_oicr_refine_fwd_bwd", _p(scores), ld, off, stages, _p(s0), s0_ld, s0_off, _p(boxes),
            _p(labels), _p(num_proposals), float(iou_threshold), float(weight), batch, n, c,
            _p(loss), _p(dscores), lddl, doff, _p(softmax_out), _p(idx), _p(top_boxes), _stream())


def labels_from_ids(ids, num_classes, labels):
  batch, t = ids.shape
  _lib.call("c2d_labels_from_ids", _p(ids) if t > 0 else None, batch, t, num_classes, _p(labels),
            _stream())


def text_classifier_fwd(ids, embedding, w1, b1, w2, b2, exact_labels, label_threshold, logits,
                        labels, workspace=None):
  """workspace: float32 device tensor [batch, hidden] -> the multi-workgroup form
  (c2d_text_classifier_fwd_ws); None -> the one-workgroup-per-caption form."""
  batch, t = ids.shape
  if workspace is not None:
    _lib.call("c2d_text_classifier_fwd_ws", _p(ids), batch, t, _p(embedding),
              embedding.shape[0] - 1, embedding.shape[1], _p(w1), _p(b1), w1.shape[1], _p(w2),
              _p(b2), w2.shape[1], _p(exact_labels), float(label_threshold), _p(logits), _p(labels),
              _p(workspace), workspace.numel() * workspace.element_size(), _stream())
    return
  _lib.call("c2d_text_classifier_fwd", _p(ids), batch, t, _p(embedding), embedding.shape[0] - 1,
            embedding.shape[1], _p(w1), _p(b1), w1.shape[1], _p(w2), _p(b2), w2.shape[1],
            _p(exact_labels), float(label_threshold), _p(logits), _p(labels), _stream())


def word_vector_match_fwd(ids, embedding, class_ids, exact_labels, labels):
  batch, t = ids.shape
  _lib.call("c2d_word_vector_match_fwd", _p(ids), batch, t, _p(embedding), embedding.shape[0] - 1,
            embedding.shape[1], _p(class_ids), class_ids.numel(), _p(exact_labels), _p(labels),
            _stream())


# -- inference post-processing ------------------------------------------------------------

def multiclass_nms(boxes, scores, ld, off, num_classes, score_thresh, iou_thresh,
                   max_size_per_class, max_total_size, workspace=None):
  """boxes [B,N,4]; scores: [B,N,ld] tensor whose class columns start at `off`.  Returns
  (num_detections int32 [B], boxes [B,T,4], scores [B,T], classes [B,T] 1-based)."""
  b, n = boxes.shape[0], boxes.shape[1]
  _f32(boxes, scores)
  dev = boxes.device
  mpc = min(max_size_per_class, n)
  need = int(_lib.load().c2d_multiclass_nms_workspace_bytes(b, n, num_classes, mpc))
  if workspace is None or workspace.numel() < need:
    workspace = torch.empty(need, dtype=torch.uint8, device=dev)
  num = torch.empty(b, dtype=torch.int32, device=dev)
  ob = torch.empty(b, max_total_size, 4, device=dev)
  osc = torch.empty(b, max_total_size, device=dev)
  ocl = torch.empty(b, max_total_size, device=dev)
  _lib.call("c2d_multiclass_nms", _p(boxes), _p(scores), ld, off, b, n, num_classes,
            float(score_thresh), float(iou_thresh), max_size_per_class, max_total_size, _p(num),
            _p(ob), _p(osc), _p(ocl), _p(workspace), workspace.numel(), _stream())
  return num, ob, osc, ocl


def softmax_drop_background(logits, ld, off, rows, c1, out):
  _lib.call("c2d_softmax_drop_background", _p(logits), ld, off, rows, c1, _p(out), _stream())


def scores_accumulate(dst, src, ld, off, rows, cols, init):
  _lib.call("c2d_scores_accumulate", _p(dst), _p(src), ld, off, rows, cols, int(init), _stream())


def scores_divide(x, divisor):
  _lib.call("c2d_scores_divide", _p(x), x.numel(), float(divisor), _stream())


def resize_bilinear(image, oh, ow, out=None):
  """image [H,W,C] fp32 -> [oh,ow,C] (TF1 legacy bilinear, align_corners=False)."""
  ih, iw, c = image.shape
  _f32(image)
  if out is None:
    out = torch.empty(oh, ow, c, device=image.device, dtype=torch.float32)
  _lib.call("c2d_resize_bilinear", _p(image), ih, iw, c, _p(out), oh, ow, _stream())
  return out


# -- input pipeline (GPU side) ----------------------------------------------------------

def image_resize_pad_u8(image_u8, flip, canvas, oh, ow):
  """image_u8 [ih,iw,3] uint8 device tensor -> canvas [ph,pw,3] fp32 (top-left oh x ow)."""
  ih, iw, _ = image_u8.shape
  ph, pw, _ = canvas.shape
  assert image_u8.dtype == torch.uint8 and canvas.dtype == torch.float32
  _lib.call("c2d_image_resize_pad_u8", _p(image_u8), ih, iw, int(bool(flip)), _p(canvas), oh, ow,
            ph, pw, _stream())


# -- text-classifier training -----------------------------------------------------------

def embedding_gather(ids, embedding, ld, x):
  _lib.call("c2d_embedding_gather", _p(ids), ids.numel(), _p(embedding), embedding.shape[0] - 1,
            embedding.shape[1], ld, _p(x), _stream())


def text_pool_fwd(pre, ids, hidden_units, vocab_size, keep_mask, keep_prob, hidden):
  b, t = ids.shape
  _lib.call("c2d_text_pool_fwd", _p(pre), _p(ids), b, t, hidden_units, vocab_size, _p(keep_mask),
            float(keep_prob), _p(hidden), _stream())


def text_pool_bwd(dhidden, pre, ids, hidden_units, vocab_size, keep_mask, keep_prob, dpre):
  b, t = ids.shape
  _lib.call("c2d_text_pool_bwd", _p(dhidden), _p(pre), _p(ids), b, t, hidden_units, vocab_size,
            _p(keep_mask), float(keep_prob), _p(dpre), _stream())


# -- optimiser --------------------------------------------------------------------------

def adagrad_step(w, g, acc, lr, l2, mult, grad_scale=1.0):
  _lib.call("c2d_adagrad_step", _p(w), _p(g), _p(acc), w.numel(), float(lr), float(l2),
            float(mult), float(grad_scale), _stream())


def adagrad_step_multi(values, grads, accum, segments, lr, grad_scale=1.0, values_bf16=None):
  """adagrad_step over `segments` = [(offset, end, mult, l2), ...] (<= 8) of the flat buffers in one
  launch; values_bf16: the bf16 mirror of `values` to refresh along the way (or None)."""
  n = len(segments)
  assert 0 < n <= 8 and values.numel() == grads.numel() == accum.numel()
  assert all(0 <= s[0] <= s[1] <= values.numel() for s in segments)
  assert values_bf16 is None or (values_bf16.dtype == torch.bfloat16 and
                                 values_bf16.numel() == values.numel())
  lls, fls = ctypes.c_longlong * n, ctypes.c_float * n
  from cap2det_amd.step_plan import sym_like
  _lib.call_sym("c2d_adagrad_step_multi", _p(values), _p(grads), _p(accum), n,
                lls(*[int(s[0]) for s in segments]), lls(*[int(s[1]) for s in segments]),
                fls(*[float(s[2]) for s in segments]), fls(*[float(s[3]) for s in segments]),
                sym_like(lr, float(lr)), float(grad_scale),
                None if values_bf16 is None else _p(values_bf16), _stream())


def l2_loss(w, weight, out):
  _lib.call("c2d_l2_loss", _p(w), w.numel(), float(weight), _p(out), _stream())


def l1_loss(w, weight, out):
  _lib.call("c2d_l1_loss", _p(w), w.numel(), float(weight), _p(out), _stream())


def adagrad_step_ex(w, g, acc, lr, l1=0.0, l2=0.0, mult=1.0, grad_scale=1.0, col_mult=None, ld=0,
                    lr_dev=None):
  _lib.call("c2d_adagrad_step_ex", _p(w), _p(g), _p(acc), w.numel(), float(lr), _p(lr_dev),
            float(l1), float(l2), float(mult), float(grad_scale), _p(col_mult), int(ld), _stream())


OPT_KINDS = {"sgd": 0, "momentum": 1, "adam": 2, "rmsprop": 3}


def optimizer_step(kind, w, g, slots, lr, p=(0.0, 0.0, 0.0, 0.0), flags=0, l1=0.0, l2=0.0, mult=1.0,
                   grad_scale=1.0, col_mult=None, ld=0, lr_dev=None):
  """c2d_optimizer_step: sgd / momentum / adam / rmsprop (TF 1.x rules) on a flat segment.
  slots: up to three state tensors of w's size (None where the rule has none)."""
  s = list(slots) + [None] * (3 - len(slots))
  _lib.call("c2d_optimizer_step", OPT_KINDS[kind], int(flags), _p(w), _p(g), _p(s[0]), _p(s[1]),
            _p(s[2]), w.numel(), float(lr), _p(lr_dev), float(p[0]), float(p[1]), float(p[2]),
            float(p[3]), float(l1), float(l2), float(mult), float(grad_scale), _p(col_mult), int(ld),
            _stream())


class ClipDesc(ctypes.Structure):
  """C2dClipDesc of include/cap2det_hip.h."""
  _fields_ = [("offset", ctypes.c_longlong), ("rows", ctypes.c_int), ("cols", ctypes.c_int),
              ("ld", ctypes.c_int), ("l1", ctypes.c_float), ("l2", ctypes.c_float),
              ("mult", ctypes.c_float)]


def clip_descriptors(records, device):
  """records: [(offset, rows, cols, ld, l1, l2, mult)] -> uint8 device tensor of C2dClipDesc."""
  arr = (ClipDesc * len(records))()
  for d, r in zip(arr, records):
    d.offset, d.rows, d.cols, d.ld, d.l1, d.l2, d.mult = r
  return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device), len(records)


def clip_gradient_norms(grads, values, desc, num, grad_scale, max_norm):
  _lib.call("c2d_clip_gradient_norms", _p(grads), _p(values), _p(desc), num, float(grad_scale),
            float(max_norm), _stream())


def last_dispatch():
  """Kernel template instances launched by this thread's last convolution call (debug query)."""
  buf = ctypes.create_string_buffer(1024)
  _lib.call("c2d_debug_last_dispatch", ctypes.cast(buf, ctypes.c_void_p), 1024)
  return [k for k in buf.value.decode().split(";") if k]


def conv_wgrad_splits(dtype, ldx, xoff, ldc, coff, n, ih, iw, cin, cout, kh, kw, stride):
  """Number of split-K slabs conv_wgrad_partial writes for this layer (< 0: the layer does not
  qualify, e.g. bf16 operands that miss the alignment of the bf16 MFMA kernels)."""
  fn = "c2d_conv_wgrad_bf16_splits" if dtype == torch.bfloat16 else "c2d_conv_wgrad_splits"
  return int(getattr(_lib.load(), fn)(ldx, xoff, ldc, coff, n, ih, iw, cin, cout, kh, kw, stride))


def conv_wgrad_partial(x, ldx, xoff, dc, ldc, coff, partials, n, ih, iw, cin, cout, kh, kw,
                       stride):
  """conv_wgrad with every K split storing its own fp32 slab (no atomics); wgrad_reduce_batched
  adds the slabs into the filter gradient."""
  assert x.dtype == dc.dtype and partials.dtype == torch.float32
  fn = "c2d_conv_wgrad_bf16_partial" if x.dtype == torch.bfloat16 else "c2d_conv_wgrad_partial"
  _lib.call(fn, _p(x), ldx, xoff, _p(dc), ldc, coff, _p(partials), partials.numel(), n, ih, iw,
            cin, cout, kh, kw, stride, _stream())


class WgradReduceDesc(ctypes.Structure):
  """C2dWgradReduceDesc of include/cap2det_hip.h."""
  _fields_ = [("ws_off", ctypes.c_longlong), ("dw_off", ctypes.c_longlong),
              ("numel", ctypes.c_int), ("splits", ctypes.c_int), ("begin", ctypes.c_int),
              ("pad", ctypes.c_int)]


def wgrad_reduce_descriptors(records, device):
  """records: [(ws_off, dw_off, numel, splits)] -> (uint8 device tensor, num, total_chunks)."""
  arr = (WgradReduceDesc * len(records))()
  chunks = 0
  for d, (ws_off, dw_off, numel, splits) in zip(arr, records):
    d.ws_off, d.dw_off, d.numel, d.splits, d.begin, d.pad = ws_off, dw_off, numel, splits, chunks, 0
    chunks += -(-numel // 1024)
  return torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device), len(records), chunks


def wgrad_reduce_batched(desc, num, total_chunks, workspace, grads):
  _lib.call("c2d_wgrad_reduce_batched", _p(desc), num, total_chunks, _p(workspace), _p(grads),
            _stream())


# -- f32x9: fp32 convolutions as nine bf16 partial products (csrc/igemm_x9.hip) -------------------

def split3_bf16(src, planes):
  """planes [3, >= src.numel()] bf16 <- the truncation split of the flat fp32 tensor `src`."""
  assert src.dtype == torch.float32 and planes.dtype == torch.bfloat16 and planes.dim() == 2
  assert planes.shape[0] == 3 and planes.shape[1] >= src.numel()
  _lib.call("c2d_split3_bf16", _p(src), _p(planes), planes.shape[1], src.numel(), _stream())
  return planes


def split3_span(src, planes, lo, hi):
  """split3_bf16 of the elements [lo, hi) of the flat tensor `src` into the same columns of `planes`
  (lo, hi multiples of 4)."""
  assert src.dtype == torch.float32 and planes.dtype == torch.bfloat16 and planes.shape[0] == 3
  assert 0 <= lo < hi <= src.numel() <= planes.shape[1] and lo % 4 == 0 and hi % 4 == 0
  _lib.call("c2d_split3_bf16", _p(src) + 4 * lo, _p(planes) + 2 * lo, planes.shape[1], hi - lo, _stream())


def f32x9_bind(arena, planes):
  """GEMM calls whose fp32 weight operand lies inside the flat tensor `arena` run as nine bf16
  partial products on the planes [3, n] (kept current by the caller: split3_bf16)."""
  assert arena.dtype == torch.float32 and planes.dtype == torch.bfloat16 and planes.shape[0] == 3
  _lib.call("c2d_f32x9_bind", _p(arena), arena.numel(), _p(planes), planes.shape[1])


def f32x9_unbind(arena=None):
  _lib.call("c2d_f32x9_unbind", _p(arena))


def f32x9_enable(on):
  """Returns the previous setting."""
  return bool(_lib.load().c2d_f32x9_enable(1 if on else 0))


def x9_planes(t):
  """A fresh plane arena for the flat fp32 tensor t (row length padded to 8 elements), bound."""
  n = -(-t.numel() // 8) * 8
  planes = torch.zeros(3, n, device=t.device, dtype=torch.bfloat16)
  split3_bf16(t.view(-1), planes)
  f32x9_bind(t.view(-1), planes)
  return planes
