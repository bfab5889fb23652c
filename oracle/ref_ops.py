"""ORACLE (test infrastructure, never shipped, never imported by cap2det_amd/).

CPU restatement in numpy of the tensor ops on the Cap2Det hot path.  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this package.

Parity status
  * PINNED by the reference's own known-answer tests (ported as tests/golden/*.json):
    `masked_*` (core/utils_test.py:13-202), `area/intersect/iou/flip_left_right/
    scale_to_new_size` (core/box_utils_test.py:11-107).
  * PARITY UNPINNED (arithmetic lives in un-vendored third-party code, the reference holds
    no test and TensorFlow cannot be installed here): `crop_and_resize` (TensorFlow 1.15.0
    core/kernels/crop_and_resize_op.cc), SAME-padded conv / pools / inference batch-norm
    (TF 1.15 + tf.contrib.slim), i.e. everything below the "third-party TF op semantics"
    marker.  Those are restated from the published TF algorithms and cross-checked in
    tests/ against independent torch-CPU implementations.

All functions take/return numpy arrays; `dtype` follows the inputs (float32 mirrors the
reference, float64 is used as the high-precision arbiter in tolerance tests).
"""
import numpy as np

_BIG_NUMBER = 1e10    # core/utils.py:10
_SMALL_NUMBER = 1e-10  # core/utils.py:11


# ----------------------------------------------------------------------------------------
# core/utils.py
# ----------------------------------------------------------------------------------------

def masked_maximum(data, mask, dim=1):
  """core/utils.py:63-79."""
  axis_minimums = np.min(data, axis=dim, keepdims=True)
  return np.max((data - axis_minimums) * mask, axis=dim, keepdims=True) + axis_minimums


def masked_minimum(data, mask, dim=1):
  """core/utils.py:82-98."""
  axis_maximums = np.max(data, axis=dim, keepdims=True)
  return np.min((data - axis_maximums) * mask, axis=dim, keepdims=True) + axis_maximums


def masked_sum(data, mask, dim=1):
  """core/utils.py:101-113."""
  return np.sum(data * mask, axis=dim, keepdims=True)


def masked_avg(data, mask, dim=1):
  """core/utils.py:116-132."""
  masked_sums = masked_sum(data, mask, dim)
  denom = np.maximum(np.asarray(_SMALL_NUMBER, dtype=data.dtype),
                     np.sum(mask, axis=dim, keepdims=True))
  return masked_sums / denom


def masked_sum_nd(data, mask, dim=1):
  """core/utils.py:135-148."""
  return np.sum(data * mask[..., None], axis=dim, keepdims=True)


def masked_avg_nd(data, mask, dim=1):
  """core/utils.py:151-169."""
  masked_sums = masked_sum_nd(data, mask, dim)
  denom = np.maximum(np.asarray(_SMALL_NUMBER, dtype=data.dtype),
                     np.sum(mask, axis=dim, keepdims=True)[..., None])
  return masked_sums / denom


def softmax(x, axis=-1):
  """tf.nn.softmax: exp(x - max) / sum."""
  m = np.max(x, axis=axis, keepdims=True)
  e = np.exp(x - m)
  return e / np.sum(e, axis=axis, keepdims=True)


def log_softmax(x, axis=-1):
  m = np.max(x, axis=axis, keepdims=True)
  s = x - m
  return s - np.log(np.sum(np.exp(s), axis=axis, keepdims=True))


def masked_softmax(data, mask, dim=-1):
  """core/utils.py:172-184: softmax(data - 1e10 * (1 - mask))."""
  big = np.asarray(_BIG_NUMBER, dtype=data.dtype)
  return softmax(data - big * (1.0 - mask), axis=dim)


def masked_argmax(data, mask, dim=1):
  """core/utils.py:187-199 (tf.argmax: first index among ties)."""
  axis_minimums = np.min(data, axis=dim, keepdims=True)
  return np.argmax((data - axis_minimums) * mask, axis=dim).astype(np.int64)


def masked_argmin(data, mask, dim=1):
  """core/utils.py:202-214."""
  axis_maximums = np.max(data, axis=dim, keepdims=True)
  return np.argmin((data - axis_maximums) * mask, axis=dim).astype(np.int64)


def sequence_mask(lengths, maxlen, dtype=np.float32):
  """tf.sequence_mask as used at models/cap2det_model.py:70-72."""
  return (np.arange(maxlen)[None, :] < np.asarray(lengths)[:, None]).astype(dtype)


# ----------------------------------------------------------------------------------------
# core/box_utils.py
# ----------------------------------------------------------------------------------------

def scale_to_new_size(box, img_shape, pad_shape):
  """core/box_utils.py:9-26."""
  box = np.asarray(box)
  img_h, img_w = np.float32(img_shape[0]), np.float32(img_shape[1])
  pad_h, pad_w = np.float32(pad_shape[0]), np.float32(pad_shape[1])
  ymin, xmin, ymax, xmax = [box[..., i] for i in range(4)]
  return np.stack([ymin * img_h / pad_h, xmin * img_w / pad_w,
                   ymax * img_h / pad_h, xmax * img_w / pad_w], axis=-1)


def flip_left_right(box):
  """core/box_utils.py:29-41."""
  ymin, xmin, ymax, xmax = [box[:, i] for i in range(4)]
  return np.stack([ymin, 1.0 - xmax, ymax, 1.0 - xmin], axis=-1)


def area(box):
  """core/box_utils.py:44-57."""
  ymin, xmin, ymax, xmax = [box[:, i] for i in range(4)]
  return np.maximum(xmax - xmin, 0.0) * np.maximum(ymax - ymin, 0.0)


def intersect(box1, box2):
  """core/box_utils.py:60-80."""
  ymin1, xmin1, ymax1, xmax1 = [box1[:, i] for i in range(4)]
  ymin2, xmin2, ymax2, xmax2 = [box2[:, i] for i in range(4)]
  return np.stack([np.maximum(ymin1, ymin2), np.maximum(xmin1, xmin2),
                   np.minimum(ymax1, ymax2), np.minimum(xmax1, xmax2)], axis=-1)


def iou(box1, box2):
  """core/box_utils.py:83-97.  0/0 yields NaN exactly as the TF graph does."""
  inter = area(intersect(box1, box2))
  union = area(box1) + area(box2) - inter
  with np.errstate(divide="ignore", invalid="ignore"):
    return inter / union


# ----------------------------------------------------------------------------------------
# third-party TF op semantics (PARITY UNPINNED — see module docstring)
# ----------------------------------------------------------------------------------------

def _axis_samples(a1, a2, n, crop, dtype=None):
  """Sampling coordinates of one box axis, TF crop_and_resize_op.cc order of operations.
  Coordinates are ALWAYS computed in float32 as the TF kernel does (also when the arbiter runs
  the rest of the arithmetic in float64), because `in > size-1` is a discontinuous test."""
  f = np.float32
  a1, a2 = f(a1), f(a2)
  nm1 = f(n - 1)
  if crop > 1:
    scale = f(f(f(a2 - a1) * nm1) / f(crop - 1))
    coords = [f(f(a1 * nm1) + f(f(i) * scale)) for i in range(crop)]
  else:
    coords = [f(f(f(0.5) * f(a1 + a2)) * nm1)]
  out = []
  for c in coords:
    if c < 0 or c > nm1 or c != c:
      out.append(None)
    else:
      lo = int(np.floor(c))
      hi = int(np.ceil(c))
      out.append((lo, hi, f(c - f(lo))))
  return out


def crop_and_resize(image, boxes, box_ind, crop_size, extrapolation_value=0.0):
  """tf.image.crop_and_resize (bilinear) as called at models/utils.py:151-155.

  image [B,H,W,D]; boxes [R,4] normalised (y1,x1,y2,x2); box_ind [R]; returns
  [R,crop,crop,D].  Rows/pixels whose sampling coordinate falls outside [0, size-1] take
  `extrapolation_value`.
  """
  image = np.asarray(image)
  f = image.dtype.type
  boxes = np.asarray(boxes, dtype=image.dtype)
  nb, h, w, d = image.shape
  r = boxes.shape[0]
  ch = cw = int(crop_size)
  out = np.full((r, ch, cw, d), extrapolation_value, dtype=image.dtype)
  for b in range(r):
    bi = int(box_ind[b])
    if bi < 0 or bi >= nb:
      continue
    y1, x1, y2, x2 = [f(v) for v in boxes[b]]
    ys = _axis_samples(y1, y2, h, ch, f)
    xs = _axis_samples(x1, x2, w, cw, f)
    xv = [(i, s) for i, s in enumerate(xs) if s is not None]
    if not xv:
      continue
    xi = np.array([i for i, _ in xv])
    xl = np.array([s[0] for _, s in xv])
    xr = np.array([s[1] for _, s in xv])
    lx = np.array([s[2] for _, s in xv], dtype=image.dtype)[:, None]
    img = image[bi]
    for y, sy in enumerate(ys):
      if sy is None:
        continue
      t, bt, ly = sy
      tl, tr = img[t, xl], img[t, xr]
      bl, br = img[bt, xl], img[bt, xr]
      top = tl + (tr - tl) * lx
      bot = bl + (br - bl) * lx
      out[b, y, xi] = top + (bot - top) * ly
  return out


def crop_and_resize_grad_image(grads, boxes, box_ind, image_shape):
  """CropAndResizeGradImage (TF 1.15): scatter-add of the four bilinear taps."""
  grads = np.asarray(grads)
  f = grads.dtype.type
  boxes = np.asarray(boxes, dtype=grads.dtype)
  nb, h, w, d = image_shape
  r, ch, cw, _ = grads.shape
  out = np.zeros(image_shape, dtype=grads.dtype)
  for b in range(r):
    bi = int(box_ind[b])
    if bi < 0 or bi >= nb:
      continue
    y1, x1, y2, x2 = [f(v) for v in boxes[b]]
    ys = _axis_samples(y1, y2, h, ch, f)
    xs = _axis_samples(x1, x2, w, cw, f)
    for y, sy in enumerate(ys):
      if sy is None:
        continue
      t, bt, ly = sy
      for x, sx in enumerate(xs):
        if sx is None:
          continue
        l, rr, lx = sx
        g = grads[b, y, x]
        dtop = (f(1) - ly) * g
        dbot = ly * g
        out[bi, t, l] += (f(1) - lx) * dtop
        out[bi, t, rr] += lx * dtop
        out[bi, bt, l] += (f(1) - lx) * dbot
        out[bi, bt, rr] += lx * dbot
  return out


def same_padding(n, k, s):
  """TF 'SAME': out = ceil(n/s), pad_total = max((out-1)*s + k - n, 0), extra at the end."""
  out = -(-n // s)
  pad = max((out - 1) * s + k - n, 0)
  return out, pad // 2, pad - pad // 2


def _windows(x, kh, kw, s, pad_value, padding):
  """Returns strided window view [N,OH,OW,kh,kw,C] of padded x plus geometry."""
  n, h, w, c = x.shape
  if padding == "SAME":
    oh, pt, pb = same_padding(h, kh, s)
    ow, pl, pr = same_padding(w, kw, s)
  else:
    oh, ow = (h - kh) // s + 1, (w - kw) // s + 1
    pt = pb = pl = pr = 0
  xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)), mode="constant",
              constant_values=pad_value)
  sn, sh, sw, sc = xp.strides
  view = np.lib.stride_tricks.as_strided(
      xp, shape=(n, oh, ow, kh, kw, c), strides=(sn, sh * s, sw * s, sh, sw, sc),
      writeable=False)
  return view, (oh, ow, pt, pl, xp.shape)


_CONV_BACKEND = "numpy"


def set_conv_backend(name):
  """"numpy" (default: im2col + matmul in the input dtype, the arbiter of the parity tests) or
  "torch" (torch-CPU conv2d on all host threads, float32 only: bench.py's measured CPU baseline,
  SURVEY.md §8d; tests/test_oracle_vs_torch.py checks the two backends against each other)."""
  global _CONV_BACKEND
  if name not in ("numpy", "torch"):
    raise ValueError(name)
  _CONV_BACKEND = name


def _torch_conv_operands(x, w, stride, padding):
  """NCHW views of NHWC memory (channels_last: no copy).  A symmetric SAME padding is handed to
  torch's own `padding` argument; an asymmetric one (extra cell at the bottom / right) is
  materialised with F.pad."""
  import torch
  import torch.nn.functional as F
  kh, kw, _, _ = w.shape
  if padding == "SAME":
    _, pt, pb = same_padding(x.shape[1], kh, stride)
    _, pl, pr = same_padding(x.shape[2], kw, stride)
  else:
    pt = pb = pl = pr = 0
  xt = torch.from_numpy(np.ascontiguousarray(x)).permute(0, 3, 1, 2)      # NCHW view, NHWC memory
  wt = torch.from_numpy(np.ascontiguousarray(w)).permute(3, 2, 0, 1)      # OIHW view of HWIO
  if pt == pb and pl == pr:
    return torch, F, xt, wt, (0, 0), (pt, pl)
  return torch, F, F.pad(xt, (pl, pr, pt, pb)), wt, (pt, pl), (0, 0)


def _conv2d_torch(x, w, stride, padding):
  torch, F, xp, wt, _, pad = _torch_conv_operands(x, w, stride, padding)
  with torch.no_grad():
    y = F.conv2d(xp, wt, stride=stride, padding=pad)
  return np.ascontiguousarray(y.permute(0, 2, 3, 1).numpy())


def _conv2d_backward_torch(x, w, dy, stride, padding, need_dx):
  torch, F, xp, wt, (pt, pl), pad = _torch_conv_operands(x, w, stride, padding)
  g = torch.from_numpy(np.ascontiguousarray(dy)).permute(0, 3, 1, 2)
  with torch.no_grad():
    dw = torch.nn.grad.conv2d_weight(xp, wt.shape, g, stride=stride, padding=pad)
    dx = None
    if need_dx:
      dxp = torch.nn.grad.conv2d_input(xp.shape, wt, g, stride=stride, padding=pad)
      dx = np.ascontiguousarray(
          dxp[:, :, pt:pt + x.shape[1], pl:pl + x.shape[2]].permute(0, 2, 3, 1).numpy())
  return dx, np.ascontiguousarray(dw.permute(2, 3, 1, 0).numpy())


def conv2d(x, w, stride=1, padding="SAME"):
  """tf.nn.conv2d NHWC/HWIO (slim.conv2d core), no bias."""
  if _CONV_BACKEND == "torch" and x.dtype == np.float32:
    return _conv2d_torch(x, w.astype(np.float32, copy=False), stride, padding)
  kh, kw, cin, cout = w.shape
  view, (oh, ow, _, _, _) = _windows(x, kh, kw, stride, 0.0, padding)
  cols = view.reshape(x.shape[0] * oh * ow, kh * kw * cin)
  return (cols @ w.reshape(kh * kw * cin, cout)).reshape(x.shape[0], oh, ow, cout)


def conv2d_backward(x, w, dy, stride=1, padding="SAME", need_dx=True):
  """Gradients of conv2d: returns (dx or None, dw)."""
  if _CONV_BACKEND == "torch" and x.dtype == np.float32:
    return _conv2d_backward_torch(x, w.astype(np.float32, copy=False),
                                  dy.astype(np.float32, copy=False), stride, padding, need_dx)
  kh, kw, cin, cout = w.shape
  n = x.shape[0]
  view, (oh, ow, pt, pl, pshape) = _windows(x, kh, kw, stride, 0.0, padding)
  cols = view.reshape(n * oh * ow, kh * kw * cin)
  dy2 = dy.reshape(n * oh * ow, cout)
  dw = (cols.T @ dy2).reshape(kh, kw, cin, cout)
  dx = None
  if need_dx:
    dcols = (dy2 @ w.reshape(kh * kw * cin, cout).T).reshape(n, oh, ow, kh, kw, cin)
    dxp = np.zeros(pshape, dtype=x.dtype)
    for ky in range(kh):
      for kx in range(kw):
        dxp[:, ky:ky + stride * oh:stride, kx:kx + stride * ow:stride, :] += dcols[:, :, :, ky, kx, :]
    dx = dxp[:, pt:pt + x.shape[1], pl:pl + x.shape[2], :]
  return dx, dw


def depthwise_conv2d(x, w, stride=1, padding="SAME"):
  """tf.nn.depthwise_conv2d: w [kh,kw,cin,mult]; out channel = ci*mult + m."""
  kh, kw, cin, mult = w.shape
  view, (oh, ow, _, _, _) = _windows(x, kh, kw, stride, 0.0, padding)
  out = np.einsum("nhwklc,klcm->nhwcm", view, w, optimize=True)
  return out.reshape(x.shape[0], oh, ow, cin * mult)


def batch_norm_inference(x, gamma, beta, mean, var, eps=0.001):
  """FusedBatchNorm(is_training=False): (x - mean) * (rsqrt(var + eps) * gamma) + beta."""
  f = x.dtype.type
  scaling = (f(1) / np.sqrt(var + f(eps))) * (gamma if gamma is not None else f(1))
  return (x - mean) * scaling.astype(x.dtype) + beta


def _pool_geometry(h, w, k, stride, padding):
  if padding == "SAME":
    oh, pt, pb = same_padding(h, k, stride)
    ow, pl, pr = same_padding(w, k, stride)
  else:
    oh, ow = (h - k) // stride + 1, (w - k) // stride + 1
    pt = pb = pl = pr = 0
  return oh, ow, pt, pb, pl, pr


def _max_pool_torch(x, k, stride, padding):
  """torch-CPU form (conv backend "torch"): same values; the returned `arg` is torch's flat
  index into the PADDED plane (int64) instead of the window index — `max_pool_backward`
  recognises it by dtype.  torch keeps the first maximum in scan order, as TF does."""
  import torch
  import torch.nn.functional as F
  n, h, w, c = x.shape
  oh, ow, pt, pb, pl, pr = _pool_geometry(h, w, k, stride, padding)
  xt = torch.from_numpy(np.ascontiguousarray(x)).permute(0, 3, 1, 2)
  xp = F.pad(xt, (pl, pr, pt, pb), value=float("-inf"))
  with torch.no_grad():
    y, idx = F.max_pool2d(xp, k, stride, return_indices=True)
  return (np.ascontiguousarray(y.permute(0, 2, 3, 1).numpy()),
          np.ascontiguousarray(idx.permute(0, 2, 3, 1).numpy()))


def _max_pool_backward_torch(x_shape, arg, dy, k, stride, padding):
  import torch
  n, h, w, c = x_shape
  oh, ow, pt, pb, pl, pr = _pool_geometry(h, w, k, stride, padding)
  hp, wp = h + pt + pb, w + pl + pr
  idx = torch.from_numpy(arg).reshape(n, oh * ow, c)                   # NHWC throughout: no copies
  g = torch.from_numpy(np.ascontiguousarray(dy)).reshape(n, oh * ow, c)
  dxp = torch.zeros(n, hp * wp, c, dtype=g.dtype)
  dxp.scatter_add_(1, idx, g)
  return np.ascontiguousarray(dxp.view(n, hp, wp, c)[:, pt:pt + h, pl:pl + w, :].numpy())


def batch_norm_relu(x, gamma, beta, mean, var, eps=0.001):
  """relu(batch_norm_inference(x)) — the tail of every slim conv2d of the extractor."""
  if _CONV_BACKEND == "torch" and x.dtype == np.float32:
    import torch
    f = np.float32
    scaling = ((f(1) / np.sqrt(var + f(eps))) * (gamma if gamma is not None else f(1))).astype(f)
    xt = torch.from_numpy(x)
    with torch.no_grad():
      y = torch.relu_((xt - torch.from_numpy(mean.astype(f))) * torch.from_numpy(scaling) +
                      torch.from_numpy(beta.astype(f)))
    return y.numpy()
  return np.maximum(batch_norm_inference(x, gamma, beta, mean, var, eps), 0)


def max_pool(x, k, stride, padding):
  """slim.max_pool2d.  Returns (y, argmax) with argmax = first maximum in (ky,kx) scan order."""
  if _CONV_BACKEND == "torch" and x.dtype == np.float32:
    return _max_pool_torch(x, k, stride, padding)
  view, (oh, ow, _, _, _) = _windows(x, k, k, stride, -np.inf, padding)
  flat = np.moveaxis(view.reshape(x.shape[0], oh, ow, k * k, x.shape[3]), 3, -1)
  arg = np.argmax(flat, axis=-1)
  y = np.take_along_axis(flat, arg[..., None], axis=-1)[..., 0]
  return np.ascontiguousarray(y), arg.astype(np.uint8)


def max_pool_backward(x_shape, arg, dy, k, stride, padding):
  """MaxPoolGrad: the whole gradient goes to the first maximum of each window."""
  if arg.dtype == np.int64:
    return _max_pool_backward_torch(x_shape, arg, dy, k, stride, padding)
  n, h, w, c = x_shape
  if padding == "SAME":
    oh, pt, pb = same_padding(h, k, stride)
    ow, pl, pr = same_padding(w, k, stride)
  else:
    oh, ow = (h - k) // stride + 1, (w - k) // stride + 1
    pt = pb = pl = pr = 0
  dxp = np.zeros((n, h + pt + pb, w + pl + pr, c), dtype=dy.dtype)
  for ky in range(k):
    for kx in range(k):
      sel = (arg == ky * k + kx)
      dxp[:, ky:ky + stride * oh:stride, kx:kx + stride * ow:stride, :] += dy * sel
  return dxp[:, pt:pt + h, pl:pl + w, :]


def avg_pool_same(x, k=3):
  """slim.avg_pool2d stride 1 SAME: padded cells are excluded from the divisor (TF AvgPool)."""
  if _CONV_BACKEND == "torch" and x.dtype == np.float32 and k % 2 == 1:
    import torch
    import torch.nn.functional as F
    xt = torch.from_numpy(np.ascontiguousarray(x)).permute(0, 3, 1, 2)
    with torch.no_grad():
      y = F.avg_pool2d(xt, k, 1, k // 2, count_include_pad=False)
    return np.ascontiguousarray(y.permute(0, 2, 3, 1).numpy())
  view, _ = _windows(x, k, k, 1, 0.0, "SAME")
  ones = np.ones((1,) + x.shape[1:3] + (1,), dtype=x.dtype)
  cnt, _ = _windows(ones, k, k, 1, 0.0, "SAME")
  return view.sum(axis=(3, 4)) / cnt.sum(axis=(3, 4))


def avg_pool_same_backward(x_shape, dy, k=3):
  n, h, w, c = x_shape
  if _CONV_BACKEND == "torch" and dy.dtype == np.float32 and k % 2 == 1:
    # the adjoint of "sum over the window / valid count": g = dy / count, summed over windows
    import torch
    import torch.nn.functional as F
    ones = torch.ones(1, 1, h, w)
    cnt = F.avg_pool2d(ones, k, 1, k // 2, count_include_pad=True) * float(k * k)
    g = torch.from_numpy(np.ascontiguousarray(dy)).permute(0, 3, 1, 2) / cnt
    with torch.no_grad():
      dx = F.avg_pool2d(g, k, 1, k // 2, count_include_pad=True) * float(k * k)
    return np.ascontiguousarray(dx.permute(0, 2, 3, 1).numpy())
  ones = np.ones((1, h, w, 1), dtype=dy.dtype)
  cnt, _ = _windows(ones, k, k, 1, 0.0, "SAME")
  g = dy / cnt.sum(axis=(3, 4))
  _, pt, pb = same_padding(h, k, 1)
  _, pl, pr = same_padding(w, k, 1)
  dxp = np.zeros((n, h + pt + pb, w + pl + pr, c), dtype=dy.dtype)
  for ky in range(k):
    for kx in range(k):
      dxp[:, ky:ky + h, kx:kx + w, :] += g
  return dxp[:, pt:pt + h, pl:pl + w, :]


def sigmoid(x):
  return 1.0 / (1.0 + np.exp(-x))


def sigmoid_cross_entropy_with_logits(labels, logits):
  """tf.nn.sigmoid_cross_entropy_with_logits: max(x,0) - x*z + log1p(exp(-|x|))."""
  return np.maximum(logits, 0) - logits * labels + np.log1p(np.exp(-np.abs(logits)))


def softmax_cross_entropy_with_logits(labels, logits):
  """tf.nn.softmax_cross_entropy_with_logits over the last axis (soft labels)."""
  return -np.sum(labels * log_softmax(logits, axis=-1), axis=-1)


def fully_connected(x, w, b):
  """slim.fully_connected with activation_fn=None: x . W + b over the last axis."""
  return x @ w + b
