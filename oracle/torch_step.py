"""ORACLE (test infrastructure, never shipped, never imported by cap2det_amd/).

The timed CPU baseline of bench.py (`cpu_baseline`, kind "port"): the reference's training step
with its heavy part — `extract_frcnn_feature` (models/utils.py:108-188: Inception-V2 towers,
tf.image.crop_and_resize, the 2x2 max-pool, spatial mean, dropout) and its gradient — on
torch-CPU (oneDNN convolutions, autograd) the way the reference runs it on TensorFlow-CPU
(Eigen / MKL-DNN kernels, tf.gradients), all host threads.  The heads, MIDN, the OICR losses
(models/cap2det_model.py:53-109,274-330, models/utils.py:15-105) and the Adagrad tail are the
numpy oracle's own functions (oracle/ref_model.py), which are cheap at any size.

tests/test_oracle_vs_torch.py pins this file against the numpy oracle (same losses, same
gradients, same updated variables on a small step), so the baseline times the same arithmetic
the parity tests arbitrate with.  TensorFlow itself cannot run here."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import ref_model
from oracle.ref_ops import same_padding

BN_EPS = ref_model.BN_EPS


def _nchw(a):
  """numpy NHWC -> torch NCHW view over the same (channels_last) memory."""
  return torch.from_numpy(np.ascontiguousarray(a)).permute(0, 3, 1, 2)


def _pad_same(x, k, s, value=0.0):
  _, pt, pb = same_padding(x.shape[2], k, s)
  _, pl, pr = same_padding(x.shape[3], k, s)
  if pt or pb or pl or pr:
    x = F.pad(x, (pl, pr, pt, pb), value=value)
  return x


def _conv_bn_relu(op, x, T, name):
  if op[0] == "sepconv":
    dw = T[name + "/depthwise_weights"]                     # [k,k,cin,mult] -> [cin*mult,1,k,k]
    k, _, cin, mult = dw.shape
    c = F.conv2d(_pad_same(x, k, op[4]), dw.permute(2, 3, 0, 1).reshape(cin * mult, 1, k, k),
                 stride=op[4], groups=cin)
    c = F.conv2d(c, T[name + "/pointwise_weights"].permute(3, 2, 0, 1))
  else:
    w = T[name + "/weights"]
    c = F.conv2d(_pad_same(x, w.shape[0], op[4]), w.permute(3, 2, 0, 1), stride=op[4])
  bn = name + "/BatchNorm/"
  y = F.batch_norm(c, T[bn + "moving_mean"], T[bn + "moving_variance"], T.get(bn + "gamma"),
                   T[bn + "beta"], False, 0.0, BN_EPS)
  return torch.relu(y)


def _op(op, x, T, prefix):
  kind = op[0]
  if kind in ("conv", "sepconv"):
    return _conv_bn_relu(op, x, T, prefix + op[1])
  if kind == "maxpool":
    return F.max_pool2d(_pad_same(x, op[2], op[3], float("-inf")), op[2], op[3])
  if kind == "avgpool":
    return F.avg_pool2d(x, op[2], 1, op[2] // 2, count_include_pad=False)
  outs = []
  for branch in op[2]:
    h = x
    for bop in branch:
      h = _op(bop, h, T, prefix + op[1] + "/")
    outs.append(h)
  return torch.cat(outs, dim=1)


def crop_and_resize(feat, boxes, box_ind, crop):
  """tf.image.crop_and_resize (bilinear, extrapolation 0) on feat [B,D,H,W] (channels_last) as
  four row gathers + lerps; coordinates in float32 in TF's order (ref_ops._axis_samples)."""
  b, d, h, w = feat.shape
  rows = feat.permute(0, 2, 3, 1).reshape(b * h * w, d)
  boxes = torch.from_numpy(np.ascontiguousarray(boxes, np.float32))
  ind = torch.from_numpy(np.ascontiguousarray(box_ind)).long()
  steps = torch.arange(crop, dtype=torch.float32)

  def axis(a1, a2, n):
    nm1 = torch.tensor(float(n - 1), dtype=torch.float32)
    scale = ((a2 - a1) * nm1) / float(crop - 1)
    c = (a1 * nm1)[:, None] + steps[None, :] * scale[:, None]
    ok = (c >= 0) & (c <= nm1)
    cc = torch.where(ok, c, torch.zeros_like(c))
    lo = torch.floor(cc)
    return ok, lo.long(), torch.ceil(cc).long(), cc - lo

  oky, t, bt, ly = axis(boxes[:, 0], boxes[:, 2], h)
  okx, l, r, lx = axis(boxes[:, 1], boxes[:, 3], w)
  base = (ind * (h * w))[:, None, None]

  def gather(yy, xx):
    idx = base + yy[:, :, None] * w + xx[:, None, :]
    return rows.index_select(0, idx.reshape(-1)).view(idx.shape + (d,))

  # (coordinates and lerp weights are float32 as in TF's kernel; the lerps run in the map's type)
  lx4, ly4 = lx[:, None, :, None].to(feat.dtype), ly[:, :, None, None].to(feat.dtype)
  tl, tr = gather(t, l), gather(t, r)
  top = tl + (tr - tl) * lx4
  del tl, tr
  bl, br = gather(bt, l), gather(bt, r)
  bot = bl + (br - bl) * lx4
  del bl, br
  out = top + (bot - top) * ly4
  out = out * (oky[:, :, None] & okx[:, None, :])[..., None].to(out.dtype)
  return out.permute(0, 3, 1, 2)          # [R, D, crop, crop], channels_last memory


def predict_scores(P, examples, options, oicr_iterations):
  """Forward pass only (evaluation mode: no dropout), for fixtures at sizes the numpy oracle does not
  finish in seconds: `extract_frcnn_feature` on torch-CPU in P's arithmetic type, heads / MIDN in
  the numpy oracle.  Returns {oicr_proposal_scores_at_i, midn_proba_r_given_c, midn_class_logits}
  exactly as oracle.ref_model.build_prediction does (models/cap2det_model.py:152-216)."""
  dt = next(iter(P.values())).dtype.type
  T = {k: torch.from_numpy(v) for k, v in P.items()}
  x = _nchw(examples["image"].astype(dt)) * (2.0 / 255.0) - 1.0
  x = x.contiguous(memory_format=torch.channels_last)
  proposals = examples["proposals"]
  batch, n, _ = proposals.shape
  with torch.no_grad():
    for op in ref_model.FIRST_STAGE:
      x = _op(op, x, T, ref_model.FIRST_SCOPE)
    box_ind = np.repeat(np.arange(batch, dtype=np.int64), n)
    net = F.max_pool2d(crop_and_resize(x, proposals.reshape(-1, 4), box_ind, options.initial_crop_size),
                       options.maxpool_kernel_size, options.maxpool_stride)
    for op in ref_model.SECOND_STAGE:
      net = _op(op, net, T, ref_model.SECOND_SCOPE)
    f_np = net.mean(dim=(2, 3)).reshape(batch, n, -1).numpy()
  class_logits, scores, proba, _ = ref_model.build_midn_network(examples["number_of_proposals"], f_np, P)
  out = {"midn_class_logits": class_logits, "midn_proba_r_given_c": proba,
         "oicr_proposal_scores_at_0": scores}
  for i in range(oicr_iterations):
    out["oicr_proposal_scores_at_%d" % (i + 1)] = (
        f_np @ P["oicr/iter%d/weights" % (i + 1)] + P["oicr/iter%d/biases" % (i + 1)])
  return out


def train_step(P, accum, examples, labels, options, loss_opts, multipliers, learning_rate,
               l2_weight, dropout_mask=None, timings=None):
  """Same contract as oracle.ref_model.train_step (P / accum updated in place).  The arithmetic
  type is P's: float32 for the timed CPU baseline, float64 for the committed mid-size fixtures
  (tests/golden/gen_step_fixture.py).  `timings`: a dict that receives the wall seconds of the
  step's stages (BASELINE.md section 4's per-stage split; the backward pass of the towers is ONE
  autograd call and is reported as one stage)."""
  import time
  marks = [time.perf_counter()]

  def lap(name):
    if timings is not None:
      now = time.perf_counter()
      timings[name] = timings.get(name, 0.0) + (now - marks[0])
      marks[0] = now
  K = loss_opts["oicr_iterations"]
  dt = next(iter(P.values())).dtype.type
  tdt = torch.float64 if dt == np.float64 else torch.float32
  trainable_names = [k for k in P if not (k.endswith("moving_mean") or
                                          k.endswith("moving_variance"))]
  mult = ref_model.resolve_gradient_multipliers(trainable_names, multipliers)
  first_from = ref_model.first_trainable_index(mult)
  T = {}
  leaves = {}
  for k, v in P.items():
    t = torch.from_numpy(v)
    if k in mult and not (k.startswith("midn/") or k.startswith("oicr/")):
      t = t.clone().requires_grad_(True)
      leaves[k] = t
    T[k] = t
  image = examples["image"].astype(dt)
  x = _nchw(image) * (2.0 / 255.0) - 1.0
  x = x.contiguous(memory_format=torch.channels_last)
  spec1 = ref_model.FIRST_STAGE
  split = len(spec1) if first_from is None else first_from
  with torch.no_grad():
    for op in spec1[:split]:
      x = _op(op, x, T, ref_model.FIRST_SCOPE)
  for op in spec1[split:]:
    x = _op(op, x, T, ref_model.FIRST_SCOPE)
  lap("first_stage_forward")
  proposals = examples["proposals"]
  batch, n, _ = proposals.shape
  box_ind = np.repeat(np.arange(batch, dtype=np.int64), n)
  cropped = crop_and_resize(x, proposals.reshape(-1, 4), box_ind, options.initial_crop_size)
  lap("roi_crop_forward")
  pooled = F.max_pool2d(cropped, options.maxpool_kernel_size, options.maxpool_stride)
  lap("roi_maxpool_forward")
  net = pooled
  for op in ref_model.SECOND_STAGE:
    net = _op(op, net, T, ref_model.SECOND_SCOPE)
  avg = net.mean(dim=(2, 3))
  if dropout_mask is not None:
    avg = avg * (1.0 / options.dropout_keep_prob) * torch.from_numpy(
        dropout_mask.astype(dt)).reshape(avg.shape)
  features = avg.reshape(batch, n, -1)

  lap("second_stage_forward")
  f_np = features.detach().numpy()
  num_proposals = examples["number_of_proposals"]
  class_logits, scores, proba, midn_saved = ref_model.build_midn_network(num_proposals, f_np, P)
  predictions = {"num_proposals": num_proposals, "proposal_boxes": proposals,
                 "midn_class_logits": class_logits, "midn_proba_r_given_c": proba,
                 "oicr_proposal_scores_at_0": scores}
  for i in range(K):
    predictions["oicr_proposal_scores_at_%d" % (i + 1)] = (
        f_np @ P["oicr/iter%d/weights" % (i + 1)] + P["oicr/iter%d/biases" % (i + 1)])
  loss_dict, loss_grads = ref_model.build_loss(predictions, labels, loss_opts)
  dfeatures, grads = ref_model.heads_backward(
      loss_grads, dict(features=f_np, midn=midn_saved), P, K)
  lap("heads_losses_and_their_gradients")
  names = list(leaves)
  gs = torch.autograd.grad(features, [leaves[k] for k in names],
                           torch.from_numpy(np.ascontiguousarray(dfeatures, dt)).to(tdt),
                           allow_unused=True)
  for k, g in zip(names, gs):
    if g is not None:
      grads[k] = g.numpy()
  lap("towers_backward(second stage, ROI crop, trainable first stage)")
  out = ref_model.finish_step(P, accum, grads, loss_dict, mult, dt, learning_rate,
                              l2_weight, predictions=predictions)
  lap("regularisers_and_adagrad")
  return out
