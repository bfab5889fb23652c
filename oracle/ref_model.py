"""ORACLE (test infrastructure, never shipped, never imported by cap2det_amd/).

numpy restatement of the Cap2Det training step (forward, losses, backward, Adagrad):
  * `extract_frcnn_feature`  — models/utils.py:108-188  (backbone [3P], crop, pool, dropout)
  * `build_midn_network`     — models/cap2det_model.py:53-109
  * `build_prediction`       — models/cap2det_model.py:152-216
  * `calc_oicr_loss`         — models/utils.py:15-105
  * `build_loss`             — models/cap2det_model.py:274-330
  * `train_step`             — train/trainer.py:55-61,73-146 + core/training_utils.py:45-50

PARITY UNPINNED for the Inception-V2 arithmetic: it lives in the un-vendored
`object_detection` fork (install-env.sh:10-13, no commit pinned; call sites
models/utils.py:127-136,165-167) and in tf.contrib.slim `nets/inception_v2.py`; the layer
table below restates those published definitions.
Backward passes are hand-derived; tests cross-check them against torch autograd on CPU.
"""
import numpy as np

from oracle import ref_ops as ops

BN_EPS = 0.001

# ----------------------------------------------------------------------------------------
# Inception-V2 layer tables.  op tuples:
#   ("sepconv", name, cout, k, stride, depth_multiplier)   separable conv + BN + ReLU
#   ("conv", name, cout, k, stride)                         conv + BN + ReLU
#   ("maxpool", name, k, stride) / ("avgpool", name, k, 1)  SAME padding
#   ("block", name, [branch, ...])                          branches concatenated on channels
# ----------------------------------------------------------------------------------------


def _mixed(name, b0, b1, b2, b3, pool="avgpool"):
  return ("block", name, [
      [("conv", "Branch_0/Conv2d_0a_1x1", b0, 1, 1)],
      [("conv", "Branch_1/Conv2d_0a_1x1", b1[0], 1, 1),
       ("conv", "Branch_1/Conv2d_0b_3x3", b1[1], 3, 1)],
      [("conv", "Branch_2/Conv2d_0a_1x1", b2[0], 1, 1),
       ("conv", "Branch_2/Conv2d_0b_3x3", b2[1], 3, 1),
       ("conv", "Branch_2/Conv2d_0c_3x3", b2[2], 3, 1)],
      [(pool, "Branch_3/%s_0a_3x3" % ("AvgPool" if pool == "avgpool" else "MaxPool"), 3, 1),
       ("conv", "Branch_3/Conv2d_0b_1x1", b3, 1, 1)],
  ])


def _reduction(name, b0, b1):
  return ("block", name, [
      [("conv", "Branch_0/Conv2d_0a_1x1", b0[0], 1, 1),
       ("conv", "Branch_0/Conv2d_1a_3x3", b0[1], 3, 2)],
      [("conv", "Branch_1/Conv2d_0a_1x1", b1[0], 1, 1),
       ("conv", "Branch_1/Conv2d_0b_3x3", b1[1], 3, 1),
       ("conv", "Branch_1/Conv2d_1a_3x3", b1[2], 3, 2)],
      [("maxpool", "Branch_2/MaxPool_1a_3x3", 3, 2)],
  ])


FIRST_STAGE = [
    ("sepconv", "Conv2d_1a_7x7", 64, 7, 2, 8),
    ("maxpool", "MaxPool_2a_3x3", 3, 2),
    ("conv", "Conv2d_2b_1x1", 64, 1, 1),
    ("conv", "Conv2d_2c_3x3", 192, 3, 1),
    ("maxpool", "MaxPool_3a_3x3", 3, 2),
    _mixed("Mixed_3b", 64, (64, 64), (64, 96, 96), 32),
    _mixed("Mixed_3c", 64, (64, 96), (64, 96, 96), 64),
    _reduction("Mixed_4a", (128, 160), (64, 96, 96)),
    _mixed("Mixed_4b", 224, (64, 96), (96, 128, 128), 128),
    _mixed("Mixed_4c", 192, (96, 128), (96, 128, 128), 128),
    _mixed("Mixed_4d", 160, (128, 160), (128, 160, 160), 96),
    _mixed("Mixed_4e", 96, (128, 192), (160, 192, 192), 96),
]

SECOND_STAGE = [
    _reduction("Mixed_5a", (128, 192), (192, 256, 256)),
    _mixed("Mixed_5b", 352, (192, 320), (160, 224, 224), 128),
    _mixed("Mixed_5c", 352, (192, 320), (192, 224, 224), 128, pool="maxpool"),
]

FIRST_SCOPE = "first_stage_feature_extraction/InceptionV2/"
SECOND_SCOPE = "second_stage_feature_extraction/InceptionV2/"


def _depth(d, dm, min_depth=16):
  return max(int(d * dm), min_depth)


def iter_convs(spec, cin, dm=1.0, prefix=""):
  """Yields (full_name, op, cin, cout) for every conv/sepconv in `spec`; returns via
  StopIteration nothing — use `spec_out_channels` for the output width."""
  for op in spec:
    kind = op[0]
    if kind == "block":
      for branch in op[2]:
        c = cin
        for bop in branch:
          if bop[0] in ("conv", "sepconv"):
            cout = _depth(bop[2], dm)
            yield prefix + op[1] + "/" + bop[1], bop, c, cout
            c = cout
      cin = spec_out_channels([op], cin, dm)
    elif kind in ("conv", "sepconv"):
      cout = _depth(op[2], dm)
      yield prefix + op[1], op, cin, cout
      cin = cout


def spec_out_channels(spec, cin, dm=1.0):
  for op in spec:
    kind = op[0]
    if kind in ("conv", "sepconv"):
      cin = _depth(op[2], dm)
    elif kind == "block":
      total = 0
      for branch in op[2]:
        c = cin
        for bop in branch:
          if bop[0] in ("conv", "sepconv"):
            c = _depth(bop[2], dm)
        total += c
      cin = total
  return cin


def init_backbone_params(rng, dm=1.0, bn_scale=True, randomize_bn=True, dtype=np.float32):
  """Seeded synthetic Inception-V2 variables under the reference's variable names
  (SURVEY.md §5 'Checkpoint / resume').  He-normal conv weights."""
  params = {}
  for scope, spec, cin0 in ((FIRST_SCOPE, FIRST_STAGE, 3),
                            (SECOND_SCOPE, SECOND_STAGE,
                             spec_out_channels(FIRST_STAGE, 3, dm))):
    for name, op, cin, cout in iter_convs(spec, cin0, dm, scope):
      k = op[3]
      if op[0] == "sepconv":
        mult = min(int(cout / cin), op[5])
        params[name + "/depthwise_weights"] = (
            rng.standard_normal((k, k, cin, mult)) * np.sqrt(2.0 / (k * k))).astype(dtype)
        params[name + "/pointwise_weights"] = (
            rng.standard_normal((1, 1, cin * mult, cout)) *
            np.sqrt(2.0 / (cin * mult))).astype(dtype)
      else:
        params[name + "/weights"] = (
            rng.standard_normal((k, k, cin, cout)) * np.sqrt(2.0 / (k * k * cin))).astype(dtype)
      bn = name + "/BatchNorm/"
      if randomize_bn:
        params[bn + "beta"] = (0.1 * rng.standard_normal(cout)).astype(dtype)
        params[bn + "moving_mean"] = (0.1 * rng.standard_normal(cout)).astype(dtype)
        params[bn + "moving_variance"] = rng.uniform(0.5, 1.5, cout).astype(dtype)
        if bn_scale:
          params[bn + "gamma"] = rng.uniform(0.5, 1.5, cout).astype(dtype)
      else:
        params[bn + "beta"] = np.zeros(cout, dtype)
        params[bn + "moving_mean"] = np.zeros(cout, dtype)
        params[bn + "moving_variance"] = np.ones(cout, dtype)
        if bn_scale:
          params[bn + "gamma"] = np.ones(cout, dtype)
  return params


def truncated_normal(rng, shape, stddev, dtype=np.float32):
  """tf.truncated_normal_initializer: resample beyond 2 sigma."""
  x = rng.standard_normal(shape)
  bad = np.abs(x) > 2
  while bad.any():
    x[bad] = rng.standard_normal(int(bad.sum()))
    bad = np.abs(x) > 2
  return (x * stddev).astype(dtype)


def init_head_params(rng, feature_dims, num_classes, oicr_iterations, stddev=0.01,
                     dtype=np.float32):
  """midn/* and oicr/iter{k}/* variables (models/cap2det_model.py:78-88,190-197)."""
  p = {}
  for name, cols in ([("midn/proba_r_given_c", num_classes),
                      ("midn/proba_c_given_r", num_classes)] +
                     [("oicr/iter%d" % (i + 1), num_classes + 1)
                      for i in range(oicr_iterations)]):
    p[name + "/weights"] = truncated_normal(rng, (feature_dims, cols), stddev, dtype)
    p[name + "/biases"] = np.zeros(cols, dtype)
  return p


# ----------------------------------------------------------------------------------------
# conv + BN + ReLU and the network executor (forward with tape, backward)
# ----------------------------------------------------------------------------------------

def _bn_terms(P, name, dtype):
  bn = name + "/BatchNorm/"
  gamma = P.get(bn + "gamma")
  f = np.dtype(dtype).type
  rstd = (f(1) / np.sqrt(P[bn + "moving_variance"].astype(dtype) + f(BN_EPS))).astype(dtype)
  return gamma, P[bn + "beta"], P[bn + "moving_mean"], P[bn + "moving_variance"], rstd


def _conv_fwd(op, x, P, name):
  if op[0] == "sepconv":
    c = ops.depthwise_conv2d(x, P[name + "/depthwise_weights"], op[4], "SAME")
    c = ops.conv2d(c, P[name + "/pointwise_weights"], 1, "SAME")
  else:
    c = ops.conv2d(x, P[name + "/weights"], op[4], "SAME")
  gamma, beta, mean, var, _ = _bn_terms(P, name, x.dtype)
  y = ops.batch_norm_relu(c, gamma, beta, mean, var, BN_EPS)
  return y, (x, c, y)


def _conv_bwd(op, saved, dy, P, name, grads, need_dx):
  if op[0] == "sepconv":
    raise NotImplementedError("the separable stem is never trainable in the reference configs")
  x, c, y = saved
  gamma, beta, mean, var, rstd = _bn_terms(P, name, x.dtype)
  dz = dy * (y > 0)
  red = tuple(range(dz.ndim - 1))
  bn = name + "/BatchNorm/"
  grads[bn + "beta"] = dz.sum(axis=red)
  if gamma is not None:
    grads[bn + "gamma"] = (dz * ((c - mean) * rstd)).sum(axis=red)
    dc = dz * (rstd * gamma)
  else:
    dc = dz * rstd
  dx, dw = ops.conv2d_backward(x, P[name + "/weights"], dc, op[4], "SAME", need_dx)
  grads[name + "/weights"] = dw
  return dx


def _op_fwd(op, x, P, prefix):
  kind = op[0]
  if kind in ("conv", "sepconv"):
    return _conv_fwd(op, x, P, prefix + op[1])
  if kind == "maxpool":
    y, arg = ops.max_pool(x, op[2], op[3], "SAME")
    return y, (x.shape, arg)
  if kind == "avgpool":
    return ops.avg_pool_same(x, op[2]).astype(x.dtype), (x.shape,)
  if kind == "block":
    outs, tapes = [], []
    for branch in op[2]:
      h, t = x, []
      for bop in branch:
        h, s = _op_fwd(bop, h, P, prefix + op[1] + "/")
        t.append(s)
      outs.append(h)
      tapes.append(t)
    return np.concatenate(outs, axis=-1), (tapes, [o.shape[-1] for o in outs])
  raise ValueError(kind)


def _op_bwd(op, saved, dy, P, prefix, grads, need_dx):
  kind = op[0]
  if kind in ("conv", "sepconv"):
    return _conv_bwd(op, saved, dy, P, prefix + op[1], grads, need_dx)
  if kind == "maxpool":
    return ops.max_pool_backward(saved[0], saved[1], dy, op[2], op[3], "SAME") if need_dx else None
  if kind == "avgpool":
    return ops.avg_pool_same_backward(saved[0], dy, op[2]) if need_dx else None
  if kind == "block":
    tapes, widths = saved
    dx, off = None, 0
    for branch, t, wdt in zip(op[2], tapes, widths):
      g = dy[..., off:off + wdt]
      off += wdt
      for i in range(len(branch) - 1, -1, -1):
        g = _op_bwd(branch[i], t[i], g, P, prefix + op[1] + "/", grads,
                    need_dx or i > 0)
      if need_dx:
        dx = g if dx is None else dx + g
    return dx
  raise ValueError(kind)


def net_forward(spec, x, P, prefix):
  tape = []
  for op in spec:
    x, s = _op_fwd(op, x, P, prefix)
    tape.append(s)
  return x, tape


def net_backward(spec, tape, dy, P, prefix, first_trainable=0, need_input_grad=False):
  """Backward through spec[first_trainable:]; returns (dx or None, grads)."""
  grads = {}
  g = dy
  for i in range(len(spec) - 1, first_trainable - 1, -1):
    need_dx = need_input_grad or i > first_trainable
    g = _op_bwd(spec[i], tape[i], g, P, prefix, grads, need_dx)
  return g, grads


# ----------------------------------------------------------------------------------------
# models/utils.py:108-188  extract_frcnn_feature
# ----------------------------------------------------------------------------------------

class FrcnnOptions(object):
  """The fields of protos/frcnn.proto used on the path (+ depth multiplier for small tests)."""

  def __init__(self, initial_crop_size=14, maxpool_kernel_size=2, maxpool_stride=2,
               dropout_keep_prob=0.5, dropout_on_feature_map=False, depth_multiplier=1.0):
    self.initial_crop_size = initial_crop_size
    self.maxpool_kernel_size = maxpool_kernel_size
    self.maxpool_stride = maxpool_stride
    self.dropout_keep_prob = dropout_keep_prob
    self.dropout_on_feature_map = dropout_on_feature_map
    self.depth_multiplier = depth_multiplier


def preprocess(image):
  """FasterRCNNInceptionV2FeatureExtractor.preprocess [3P]: (2/255) * x - 1."""
  f = image.dtype.type
  return f(2.0 / 255.0) * image - f(1.0)


def extract_frcnn_feature(image, num_proposals, proposals, P, options, is_training=False,
                          dropout_mask=None, feature_map_dropout_mask=None):
  """models/utils.py:108-188.  `dropout_mask` [B*N, D] of {0,1} injects the RNG of
  slim.dropout (models/utils.py:171-174); None means no dropout (inference, or keep_prob 1).
  Returns (proposal_features [B,N,D], tape)."""
  del num_proposals  # unused by the reference on this path as well
  x = preprocess(image)
  feat, tape1 = net_forward(FIRST_STAGE, x, P, FIRST_SCOPE)
  fmask = None
  feat_pre = feat
  if options.dropout_on_feature_map and is_training and feature_map_dropout_mask is not None:
    # models/utils.py:138-142: slim.dropout on features_to_crop (the mask injects its RNG)
    f = feat.dtype.type
    fmask = feature_map_dropout_mask.reshape(feat.shape).astype(feat.dtype)
    feat = feat * f(1.0 / options.dropout_keep_prob) * fmask
  batch, n, _ = proposals.shape
  box_ind = np.repeat(np.arange(batch, dtype=np.int32), n)          # models/utils.py:147-149
  boxes = proposals.reshape(-1, 4)
  cropped = ops.crop_and_resize(feat, boxes, box_ind, options.initial_crop_size)
  pooled, pool_arg = ops.max_pool(cropped, options.maxpool_kernel_size,
                                  options.maxpool_stride, "VALID")   # :157-160
  net, tape2 = net_forward(SECOND_STAGE, pooled, P, SECOND_SCOPE)     # :165-167
  avg = net.mean(axis=(1, 2))                                         # :169-170
  if is_training and dropout_mask is not None:
    f = avg.dtype.type
    out = avg * f(1.0 / options.dropout_keep_prob) * dropout_mask.astype(avg.dtype)
  else:
    out = avg
  tape = dict(tape1=tape1, feat=feat, feat_pre=feat_pre, fmap_mask=fmask, boxes=boxes, box_ind=box_ind, cropped_shape=cropped.shape,
              pool_arg=pool_arg, pooled=pooled, tape2=tape2, net_shape=net.shape,
              dropout_mask=dropout_mask if is_training else None)
  return out.reshape(batch, n, -1), tape


def extract_frcnn_feature_backward(dfeatures, tape, P, options, first_stage_from=None):
  """Gradients of extract_frcnn_feature w.r.t. the trainable backbone variables.

  first_stage_from: index into FIRST_STAGE of the earliest trainable op (e.g. the index of
  Mixed_4e for configs/voc07_groundtruth.pbtxt:120-123) or None when the whole first stage is
  frozen (configs/voc07_inc2.pbtxt)."""
  nshape = tape["net_shape"]
  g = dfeatures.reshape(nshape[0], nshape[3])
  if tape["dropout_mask"] is not None:
    f = g.dtype.type
    g = g * f(1.0 / options.dropout_keep_prob) * tape["dropout_mask"].astype(g.dtype)
  f = g.dtype.type
  dnet = np.broadcast_to((g / f(nshape[1] * nshape[2]))[:, None, None, :], nshape)
  need_first = first_stage_from is not None
  dpooled, grads = net_backward(SECOND_STAGE, tape["tape2"], dnet, P, SECOND_SCOPE, 0, need_first)
  if need_first:
    dcrop = ops.max_pool_backward(tape["cropped_shape"], tape["pool_arg"], dpooled,
                                  options.maxpool_kernel_size, options.maxpool_stride, "VALID")
    dfeat = ops.crop_and_resize_grad_image(dcrop, tape["boxes"], tape["box_ind"],
                                           tape["feat"].shape)
    if tape.get("fmap_mask") is not None:
      dfeat = dfeat * dfeat.dtype.type(1.0 / options.dropout_keep_prob) * tape["fmap_mask"]
    _, g1 = net_backward(FIRST_STAGE, tape["tape1"], dfeat, P, FIRST_SCOPE, first_stage_from, False)
    grads.update(g1)
    grads["__dfeat__"] = dfeat
  grads["__dpooled__"] = dpooled
  return grads


# ----------------------------------------------------------------------------------------
# models/cap2det_model.py
# ----------------------------------------------------------------------------------------

def build_midn_network(num_proposals, proposal_features, P):
  """models/cap2det_model.py:53-109.  Returns (class_logits [B,C], proposal_scores [B,N,C],
  proba_r_given_c [B,N,C], saved)."""
  batch, n, _ = proposal_features.shape
  dt = proposal_features.dtype
  mask = ops.sequence_mask(num_proposals, n, dt)[..., None]
  logits_r_given_c = ops.fully_connected(proposal_features, P["midn/proba_r_given_c/weights"],
                                         P["midn/proba_r_given_c/biases"])
  logits_c_given_r = ops.fully_connected(proposal_features, P["midn/proba_c_given_r/weights"],
                                         P["midn/proba_c_given_r/biases"])
  proba_r_given_c = ops.masked_softmax(mask * logits_r_given_c, mask, dim=1)
  proba_r_given_c = mask * proba_r_given_c
  class_logits = ops.masked_sum(logits_c_given_r * proba_r_given_c, mask, dim=1)
  proposal_scores = ops.sigmoid(class_logits) * proba_r_given_c
  saved = (mask, logits_c_given_r, proba_r_given_c, class_logits[:, 0, :])
  return class_logits[:, 0, :], proposal_scores, proba_r_given_c, saved


def build_midn_network_backward(dclass_logits, saved):
  """d(loss)/d(logits_r_given_c), d(loss)/d(logits_c_given_r) given d(loss)/d(class_logits).
  (proposal_scores / proba_r_given_c reach the loss only under stop_gradient,
  models/cap2det_model.py:306-321.)"""
  mask, lc, p, cl = saved
  g = dclass_logits[:, None, :]
  dlc = g * p * mask
  dlr = g * p * (lc * mask - cl[:, None, :]) * mask
  return dlr, dlc


def build_prediction(examples, P, options, oicr_iterations, is_training=False,
                     dropout_mask=None, feature_map_dropout_mask=None):
  """models/cap2det_model.py:152-216 (without the NMS post-process, which `train_op` never
  fetches).  examples: dict with 'image' [B,H,W,3] 0..255, 'number_of_proposals' [B],
  'proposals' [B,N,4]."""
  image, num_proposals, proposals = (examples["image"], examples["number_of_proposals"],
                                     examples["proposals"])
  features, tape = extract_frcnn_feature(image, num_proposals, proposals, P, options,
                                         is_training, dropout_mask, feature_map_dropout_mask)
  class_logits, scores, proba, midn_saved = build_midn_network(num_proposals, features, P)
  predictions = {
      "num_proposals": num_proposals,
      "proposal_boxes": proposals,
      "midn_class_logits": class_logits,
      "midn_proba_r_given_c": proba,
      "oicr_proposal_scores_at_0": scores,
  }
  for i in range(oicr_iterations):
    predictions["oicr_proposal_scores_at_%d" % (i + 1)] = ops.fully_connected(
        features, P["oicr/iter%d/weights" % (i + 1)], P["oicr/iter%d/biases" % (i + 1)])
  saved = dict(frcnn=tape, features=features, midn=midn_saved)
  return predictions, saved


def calc_oicr_loss(labels, num_proposals, proposals, scores_0, scores_1, iou_threshold=0.5):
  """models/utils.py:15-105.  Returns (loss scalar, dloss/dscores_1, proposal_labels)."""
  batch, n, c1 = scores_0.shape
  num_classes = c1 - 1
  dt = scores_1.dtype
  proposal_mask = ops.sequence_mask(num_proposals, n, dt)
  proposal_ind = ops.masked_argmax(scores_0[:, :, 1:], proposal_mask[..., None], dim=1)
  targets = []
  for c in range(num_classes):
    confident = proposals[np.arange(batch), proposal_ind[:, c]]          # [B,4]
    tiled = np.repeat(confident[:, None, :], n, axis=1)
    iou = ops.iou(proposals.reshape(-1, 4), tiled.reshape(-1, 4)).reshape(batch, n)
    with np.errstate(invalid="ignore"):
      target = (iou >= dt.type(iou_threshold)).astype(dt)
    target = np.where((labels[:, c] > 0)[:, None], target, np.zeros_like(target))
    targets.append(target)
  proposal_labels = np.stack(targets, axis=-1)
  bkg = np.logical_not(proposal_labels.sum(axis=-1) > 0)
  proposal_labels = np.concatenate([bkg.astype(dt)[..., None], proposal_labels], axis=-1)
  proposal_labels = proposal_labels / proposal_labels.sum(axis=-1, keepdims=True)
  assert np.all(np.abs(proposal_labels.sum(axis=-1) - 1) < 1e-6), "Probabilities not sum to ONE"
  losses = ops.softmax_cross_entropy_with_logits(proposal_labels, scores_1)
  per_image = ops.masked_avg(losses, proposal_mask, dim=1)                # [B,1]
  loss = per_image.mean()
  denom = np.maximum(dt.type(1e-10), proposal_mask.sum(axis=1, keepdims=True))
  dscores = (ops.softmax(scores_1, axis=-1) - proposal_labels) * \
      (proposal_mask / denom)[..., None] / dt.type(batch)
  return loss, dscores, proposal_labels


def build_loss(predictions, labels, options_loss):
  """models/cap2det_model.py:274-330.  options_loss: dict(midn_loss_weight, oicr_loss_weight,
  oicr_iterations, oicr_iou_threshold, oicr_use_proba_r_given_c).
  Returns (loss_dict, grad_dict w.r.t. midn_class_logits and oicr_proposal_scores_at_k)."""
  o = options_loss
  logits = predictions["midn_class_logits"]
  dt = logits.dtype
  losses = ops.sigmoid_cross_entropy_with_logits(labels.astype(dt), logits)
  loss_dict = {"midn_cross_entropy_loss": losses.mean() * dt.type(o["midn_loss_weight"])}
  grads = {"midn_class_logits":
           (ops.sigmoid(logits) - labels) / dt.type(logits.size) * dt.type(o["midn_loss_weight"])}
  num_proposals, proposals = predictions["num_proposals"], predictions["proposal_boxes"]
  batch, n, _ = proposals.shape
  s0 = predictions["oicr_proposal_scores_at_0"]
  if o["oicr_use_proba_r_given_c"]:
    s0 = predictions["midn_proba_r_given_c"]
  s0 = np.concatenate([np.zeros((batch, n, 1), dt), s0], axis=-1)
  for i in range(o["oicr_iterations"]):
    s1 = predictions["oicr_proposal_scores_at_%d" % (i + 1)]
    loss, ds1, _ = calc_oicr_loss(labels, num_proposals, proposals, s0, s1,
                                  o["oicr_iou_threshold"])
    loss_dict["oicr_cross_entropy_loss_at_%d" % (i + 1)] = loss * dt.type(o["oicr_loss_weight"])
    grads["oicr_proposal_scores_at_%d" % (i + 1)] = ds1 * dt.type(o["oicr_loss_weight"])
    s0 = ops.softmax(s1, axis=-1)
  return loss_dict, grads


HEAD_NAMES = ("midn/proba_r_given_c", "midn/proba_c_given_r")


def heads_backward(loss_grads, saved, P, oicr_iterations):
  """Backward through the five fully-connected heads; returns (dfeatures [B,N,D], grads)."""
  x = saved["features"]
  x2 = x.reshape(-1, x.shape[-1])
  dlr, dlc = build_midn_network_backward(loss_grads["midn_class_logits"], saved["midn"])
  grads = {}
  dx = np.zeros_like(x2)
  pairs = [("midn/proba_r_given_c", dlr), ("midn/proba_c_given_r", dlc)]
  for i in range(oicr_iterations):
    pairs.append(("oicr/iter%d" % (i + 1), loss_grads["oicr_proposal_scores_at_%d" % (i + 1)]))
  for name, dl in pairs:
    dl2 = dl.reshape(-1, dl.shape[-1])
    grads[name + "/weights"] = x2.T @ dl2
    grads[name + "/biases"] = dl2.sum(axis=0)
    dx += dl2 @ P[name + "/weights"].T
  return dx.reshape(x.shape), grads


# ----------------------------------------------------------------------------------------
# train/trainer.py step semantics
# ----------------------------------------------------------------------------------------

def resolve_gradient_multipliers(var_names, multipliers):
  """train/trainer.py:104-125: prefix match in config order, later entries override,
  multiplier <= 0 freezes.  multipliers: list of (scope, value).  Returns
  {var_name: multiplier} for the variables that stay trainable (1.0 when unmatched)."""
  out = {}
  for name in var_names:
    trainable, mult = True, 1.0
    for scope, value in multipliers:
      if name.startswith(scope):
        mult = value
        trainable = value > 0
    if trainable:
      out[name] = mult
  return out


def is_regularized(name):
  """slim weights_regularizer applies to FC `weights` only (core/training_utils.py:123-128);
  the Inception extractor is built with weight_decay 0 [3P]."""
  return (name.startswith("midn/") or name.startswith("oicr/")) and name.endswith("/weights")


def first_trainable_index(multipliers_resolved):
  """Index into FIRST_STAGE of the earliest op owning a trainable variable, or None."""
  for i, op in enumerate(FIRST_STAGE):
    pre = FIRST_SCOPE + op[1]
    if any(k.startswith(pre + "/") for k in multipliers_resolved):
      return i
  return None


def train_step(P, accum, examples, labels, options, loss_opts, multipliers, learning_rate,
               l2_weight, dropout_mask=None, feature_map_dropout_mask=None, l1_weight=0.0,
               max_gradient_norm=None, optimizer=None):
  """One step of train/trainer.py:_model_fn in TRAIN mode with Adagrad
  (core/training_utils.py:45-50; tf.train.AdagradOptimizer: acc += g^2; w -= lr*g*rsqrt(acc)).

  Returns dict(losses, total_loss, grads, predictions); P and accum are updated in place."""
  K = loss_opts["oicr_iterations"]
  predictions, saved = build_prediction(examples, P, options, K, True, dropout_mask,
                                        feature_map_dropout_mask)
  loss_dict, loss_grads = build_loss(predictions, labels, loss_opts)
  dfeatures, grads = heads_backward(loss_grads, saved, P, K)
  trainable_names = [k for k in P if not (k.endswith("moving_mean") or
                                          k.endswith("moving_variance"))]
  mult = resolve_gradient_multipliers(trainable_names, multipliers)
  first_from = first_trainable_index(mult)
  grads.update(extract_frcnn_feature_backward(dfeatures, saved["frcnn"], P, options, first_from))
  return finish_step(P, accum, grads, loss_dict, mult, dfeatures.dtype.type, learning_rate,
                     l2_weight, l1_weight, max_gradient_norm, predictions, optimizer)


def optimizer_update(kind, opts, w, g, slots, lr, step, dt):
  """TensorFlow 1.x update rules of core/training_utils.py:14-71's optimisers (third party:
  tensorflow/core/kernels/training_ops.cc; parity unpinned, formulas as documented by TF).
  slots: list of state arrays, updated in place; w updated in place.  step: 1-based step count."""
  if kind == "sgd":
    w -= dt(lr) * g
  elif kind == "momentum":
    a = slots[0]
    a[...] = dt(opts["momentum"]) * a + g
    if opts.get("use_nesterov"):
      w -= dt(lr) * (g + dt(opts["momentum"]) * a)
    else:
      w -= dt(lr) * a
  elif kind == "adagrad":
    slots[0] += g * g
    w -= dt(lr) * g / np.sqrt(slots[0])
  elif kind == "adam":
    b1, b2, eps = dt(opts["beta1"]), dt(opts["beta2"]), dt(opts["epsilon"])
    m, v = slots
    m[...] = b1 * m + (dt(1) - b1) * g
    v[...] = b2 * v + (dt(1) - b2) * g * g
    lr_t = dt(lr) * np.sqrt(dt(1) - b2 ** step) / (dt(1) - b1 ** step)
    w -= lr_t * m / (np.sqrt(v) + eps)
  elif kind == "rmsprop":
    rho, mu, eps = dt(opts["decay"]), dt(opts["momentum"]), dt(opts["epsilon"])
    ms, mom = slots[0], slots[1]
    ms[...] = rho * ms + (dt(1) - rho) * g * g
    denom = ms
    if opts.get("centered"):
      mg = slots[2]
      mg[...] = rho * mg + (dt(1) - rho) * g
      denom = ms - mg * mg
    mom[...] = mu * mom + dt(lr) * g / np.sqrt(denom + eps)
    w -= mom
  else:
    raise ValueError("Invalid optimizer: {}.".format(kind))


def init_optimizer_slots(kind, opts, P):
  """Initial slot values as TensorFlow creates them (Adagrad: initial_accumulator_value,
  RMSProp: rms = 1, everything else 0)."""
  out = {}
  for k, v in P.items():
    if kind == "adagrad":
      out[k] = [np.full(v.shape, opts.get("initial_accumulator_value", 0.1), v.dtype)]
    elif kind == "momentum":
      out[k] = [np.zeros_like(v)]
    elif kind == "adam":
      out[k] = [np.zeros_like(v), np.zeros_like(v)]
    elif kind == "rmsprop":
      out[k] = [np.ones_like(v), np.zeros_like(v)] + ([np.zeros_like(v)] if opts.get("centered") else [])
    else:
      out[k] = []
  return out


def finish_step(P, accum, grads, loss_dict, mult, dt, learning_rate, l2_weight, l1_weight=0.0,
                max_gradient_norm=None, predictions=None, optimizer=None):
  """Tail of `train_step`: regularisers, gradient multipliers, per-variable norm clipping,
  Adagrad (shared with oracle/torch_step.py, bench.py's timed CPU baseline)."""
  reg = {}
  for name in P:
    if is_regularized(name):
      # slim.l2_regularizer: w * sum(x^2)/2; slim.l1_regularizer: w * sum|x|
      # (core/training_utils.py:152-171; one of the two is configured)
      reg[name] = (dt(l2_weight) * dt(0.5) * np.sum(P[name] * P[name]) +
                   dt(l1_weight) * np.sum(np.abs(P[name])))
      grads[name] = grads[name] + dt(l2_weight) * P[name] + dt(l1_weight) * np.sign(P[name])
  total = sum(loss_dict.values()) + sum(reg.values())
  applied = {}
  for name, m in mult.items():
    if name not in grads:
      continue
    g = grads[name].astype(P[name].dtype) * dt(m)
    if max_gradient_norm is not None:
      # tf.contrib.training.clip_gradient_norms (train/trainer.py:132-136): tf.clip_by_norm of
      # every gradient on its own L2 norm, after the multipliers
      norm = np.sqrt(np.sum(g * g))
      g = g * dt(max_gradient_norm) / np.maximum(norm, dt(max_gradient_norm))
    if optimizer is None:
      accum[name] += g * g
      P[name] -= dt(learning_rate) * g / np.sqrt(accum[name])
    else:
      # optimizer = dict(kind, opts, slots {name: [arrays]}, step): accum is unused
      optimizer_update(optimizer["kind"], optimizer["opts"], P[name], g, optimizer["slots"][name],
                       learning_rate, optimizer["step"], dt)
    applied[name] = g
  return dict(losses=loss_dict, reg_losses=reg, total_loss=total, grads=grads, applied=applied,
              predictions=predictions)
