"""Cap2Det / WSOD model on MI355X (reference: models/cap2det_model.py:29-346).

Same plugin surface as the reference (`Model(model_proto, is_training)`, `build_prediction`,
`build_loss`, `build_evaluation`, registered under `Cap2DetModel.ext`) with the TF graph
replaced by a static launch plan over hand-written HIP kernels.  Because there is no autograd
graph, `build_loss` also produces d(loss)/d(head logits) and `backward()` runs the explicit
gradient kernels into the flat gradient buffer consumed by `train/trainer.py`.
"""
import math
import os

import numpy as np
import torch

from cap2det_amd import hip_ops as ops
from cap2det_amd.core import builder as function_builder
from cap2det_amd.core.standard_fields import (Cap2DetPredictions, DetectionResultFields,
                                              InputDataFields)
from cap2det_amd.models.frcnn_engine import FIRST_SCOPE, SECOND_SCOPE, FrcnnEngine, VariableStore
from cap2det_amd.models.label_extractor import build_label_extractor
from cap2det_amd.models.model_base import ModelBase
from cap2det_amd.models.registry import register_model_class
from cap2det_amd.protos import cap2det_model_pb2
from cap2det_amd.protos.message import unwrap

HEADS_W = "heads/fused_weights"
HEADS_B = "heads/fused_biases"


class Model(ModelBase):
  """Cap2Det model."""

  def __init__(self, model_proto, is_training=False, device="cuda:0", depth_multiplier=1.0,
               bn_scale=True, seed=0, compute_dtype="fp32", allow_missing_pretrained=False):
    """compute_dtype: "fp32" (BASELINE configs[0], [1], [3]: exact fp32 everywhere) or "bf16"
    (configs[2], [4]: the convolution towers behind the stem — the single-image first stage, the ROI
    crop output and the second stage — in bf16 storage with fp32 accumulation; the stem, the map
    the ROI crop interpolates, heads, losses, variables and optimiser stay fp32;
    C2D_TUNE=first_stage_fp32=1 keeps the first stage in fp32 as well).
    allow_missing_pretrained: keep the synthetic initial values when
    `frcnn_options.checkpoint_path` names a file that does not exist (benchmarks and tests; the
    reference's tf.train.init_from_checkpoint fails hard, models/utils.py:181-186, and so does
    this class by default)."""
    model_proto = unwrap(model_proto)
    super(Model, self).__init__(model_proto, is_training)
    if not isinstance(model_proto, cap2det_model_pb2.Cap2DetModel):
      raise ValueError('The model_proto has to be an instance of Cap2DetModel.')
    options = model_proto
    fe_type = options.frcnn_options.feature_extractor.type
    if fe_type != 'faster_rcnn_inception_v2':
      raise ValueError('Unknown Faster R-CNN feature_extractor: {}'.format(fe_type))
    self._device = torch.device(device)
    self._midn_postprocess_fn = function_builder.build_post_processor(options.midn_post_processor)
    self._oicr_postprocess_fn = function_builder.build_post_processor(options.oicr_post_processor)
    self._label_extractor = build_label_extractor(options.label_extractor, self._device)
    self._num_classes = self._label_extractor.num_classes
    self._oicr_iterations = options.oicr_iterations

    self.store = VariableStore(self._device)
    if compute_dtype not in ("fp32", "bf16"):
      raise ValueError("compute_dtype must be 'fp32' or 'bf16'")
    self.compute_dtype = compute_dtype
    self.engine = FrcnnEngine(self.store, options.frcnn_options, bn_scale, depth_multiplier,
                              act_dtype=torch.bfloat16 if compute_dtype == "bf16" else torch.float32)
    # Five fully-connected heads fused into one [D, Npad] GEMM operand:
    # columns = [r|c (C), c|r (C), oicr_1 (C+1), ..., oicr_K (C+1)], zero padded to 16.
    c, k = self._num_classes, self._oicr_iterations
    self._head_cols = [("midn/proba_r_given_c", 0, c), ("midn/proba_c_given_r", c, c)]
    for i in range(k):
      self._head_cols.append(("oicr/iter%d" % (i + 1), 2 * c + i * (c + 1), c + 1))
    self._ncols = 2 * c + k * (c + 1)
    self._npad = -(-self._ncols // 16) * 16
    d = self.engine.feature_dims
    self.store.declare(HEADS_W, (d, self._npad))
    self.store.declare(HEADS_B, (self._npad,))
    self.store.finalize()
    self.engine.finalize(extra_transposes=[(HEADS_W, "heads/wt", 1, d, self._npad)])
    self._heads_wt = self.engine.stats.derived["heads/wt"]
    # core/training_utils.py:152-171 `_build_slim_regularizer` (FC weights only)
    self._l2_weight = self._l1_weight = 0.0
    reg = options.fc_hyperparams.regularizer
    if reg.WhichOneof('regularizer_oneof') == 'l2_regularizer':
      self._l2_weight = reg.l2_regularizer.weight
    elif reg.WhichOneof('regularizer_oneof') == 'l1_regularizer':
      self._l1_weight = reg.l1_regularizer.weight
    # midn, oicr_1..K, regularisation | total (padded to a multiple of 4 floats: c2d_zero_ranges)
    self._nloss = 2 + k
    self._losses = torch.zeros(-(-(self._nloss + 1) // 4) * 4, device=self._device)
    self._cache = {}
    self._ctx = None
    self.initialize(seed)
    # the reference initialises both towers from `frcnn_options.checkpoint_path` at graph build
    # (models/utils.py:181-186) and fails when the file is missing; so does this class, unless
    # the caller explicitly asks for the synthetic initial values (benchmarks / tests)
    ckpt = self._model_proto.frcnn_options.checkpoint_path
    self.restored_from = None
    if ckpt:
      from cap2det_amd.train import tf_checkpoint
      if tf_checkpoint.checkpoint_exists(ckpt):
        self.init_from_checkpoint(ckpt)
        self.restored_from = ckpt
      elif not allow_missing_pretrained:
        raise FileNotFoundError(
            "frcnn_options.checkpoint_path %r does not exist (pass allow_missing_pretrained=True "
            "to train from the synthetic initial values)" % ckpt)

  # -- variables ----------------------------------------------------------------------
  @property
  def num_classes(self):
    return self._num_classes

  @property
  def label_extractor(self):
    return self._label_extractor

  @property
  def l2_weight(self):
    return self._l2_weight

  @property
  def l1_weight(self):
    return self._l1_weight

  def head_columns(self):
    """[(reference scope of the head, first column, width)] inside the fused heads buffers."""
    return list(self._head_cols)

  def head_view(self, name):
    """Strided view of one head inside the fused buffers, under the reference variable name
    (`midn/proba_r_given_c/weights`, `oicr/iter2/biases`, ...)."""
    scope, leaf = name.rsplit("/", 1)
    for hname, off, width in self._head_cols:
      if hname == scope:
        if leaf == "weights":
          return self.store.var[HEADS_W][:, off:off + width]
        return self.store.var[HEADS_B][off:off + width]
    raise KeyError(name)

  def variable_names(self):
    """All variables under the reference's names (SURVEY.md §5 'Checkpoint / resume')."""
    names = list(self.engine.stem_vars) + [n for n in self.store.names()
                                           if not n.startswith("heads/")]
    names += list(self.engine.stats)
    for hname, _, _ in self._head_cols:
      names += [hname + "/weights", hname + "/biases"]
    return names

  def get_variables_to_train(self):
    """Trainable-capable variables (tf.trainable_variables() in the reference): everything
    except BatchNorm moving statistics."""
    return [n for n in self.variable_names()
            if not (n.endswith("moving_mean") or n.endswith("moving_variance"))]

  def _var_tensor(self, name):
    if name in self.engine.stem_vars:
      return self.engine.stem_vars[name]
    if name in self.engine.stats:
      return self.engine.stats[name]
    if name in self.store.var:
      return self.store.var[name]
    return self.head_view(name)

  def state_dict(self):
    return {n: self._var_tensor(n).detach().cpu().numpy().copy() for n in self.variable_names()}

  def load_state_dict(self, arrays, strict=True):
    """arrays: {reference variable name: numpy array}."""
    missing = []
    for n in self.variable_names():
      if n not in arrays:
        missing.append(n)
        continue
      t = self._var_tensor(n)
      a = torch.from_numpy(np.ascontiguousarray(arrays[n], dtype=np.float32)).to(self._device)
      if tuple(a.shape) != tuple(t.shape):
        raise ValueError("shape mismatch for %s: %s vs %s" % (n, tuple(a.shape), tuple(t.shape)))
      t.copy_(a)
    if strict == "checkpoint":
      # a checkpoint of the reference: its Inception-V2 BatchNorm has no gamma (slim's
      # inception arg_scope leaves `scale` off), which equals the gamma = 1 kept here; any other
      # absent variable means the file does not belong to this model
      hard = [n for n in missing if not n.endswith("/BatchNorm/gamma")]
      if hard:
        raise KeyError("checkpoint lacks %d model variables: %s ..." % (len(hard), hard[:5]))
    elif strict and missing:
      raise KeyError("missing variables: %s ..." % missing[:5])
    self.refresh()
    return missing

  def _slot_tensor(self, name):
    """Adagrad accumulator of variable `name` (None for BatchNorm statistics / stem variables,
    which the optimiser never touches)."""
    if name in self.store.acc:
      return self.store.acc[name]
    try:
      scope, leaf = name.rsplit("/", 1)
    except ValueError:
      return None
    for hname, off, width in self._head_cols:
      if hname == scope:
        return (self.store.acc[HEADS_W][:, off:off + width] if leaf == "weights"
                else self.store.acc[HEADS_B][off:off + width])
    return None

  def optimizer_slots(self):
    """{`<variable>/Adagrad`: accumulator} — the slot names tf.train.AdagradOptimizer saves."""
    out = {}
    for n in self.variable_names():
      t = self._slot_tensor(n)
      if t is not None:
        out[n + "/Adagrad"] = t.detach().cpu().numpy().copy()
    return out

  def load_optimizer_slots(self, arrays):
    """Restores every `<variable>/Adagrad` tensor present in `arrays`; returns the names."""
    done = []
    for n in self.variable_names():
      t = self._slot_tensor(n)
      a = arrays.get(n + "/Adagrad")
      if t is None or a is None:
        continue
      a = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(self._device)
      if tuple(a.shape) != tuple(t.shape):
        raise ValueError("shape mismatch for %s/Adagrad" % n)
      t.copy_(a)
      done.append(n)
    return done

  def init_from_checkpoint(self, path):
    """models/utils.py:181-186: both towers take their variables from ONE ImageNet checkpoint,
    `tf.train.init_from_checkpoint(path, {"/": "<tower scope>/"})`: model variable
    `<scope>/InceptionV2/X` <- checkpoint tensor `InceptionV2/X` (V1 `inception_v2.ckpt` or a V2
    prefix; cap2det_amd/train/tf_checkpoint.py).  The heads keep their initialiser."""
    from cap2det_amd.train import tf_checkpoint
    arrays = tf_checkpoint.read_checkpoint(path)
    names = [n for n in self.variable_names()]
    state = {}
    for scope in (FIRST_SCOPE.split("/")[0], SECOND_SCOPE.split("/")[0]):
      state.update(tf_checkpoint.assignment(arrays, names, scope))
    self.load_state_dict(state, strict=False)
    return sorted(state)

  def grad_dict(self):
    """Gradients of the last step under the reference variable names (numpy)."""
    out = {}
    for n in self.store.names():
      if n == HEADS_W or n == HEADS_B:
        continue
      out[n] = self.store.grad[n].detach().cpu().numpy().copy()
    for hname, off, width in self._head_cols:
      out[hname + "/weights"] = self.store.grad[HEADS_W][:, off:off + width].cpu().numpy().copy()
      out[hname + "/biases"] = self.store.grad[HEADS_B][off:off + width].cpu().numpy().copy()
    return out

  def initialize(self, seed=0):
    """Synthetic initial values (no checkpoint can be read here): He-normal convolutions,
    identity BatchNorm, heads per `fc_hyperparams.initializer` (truncated normal 0.01 in every
    shipped config, configs/*.pbtxt:58-72), zero biases."""
    gen = torch.Generator(device="cpu").manual_seed(seed)
    eng = self.engine
    # (all variables are drawn into a HOST image of the flat buffer and uploaded once: a handful
    # of copies instead of ~300 per-variable fill / copy launches)
    host = torch.zeros(self.store.values.numel())

    def normal(shape, std):
      return torch.randn(shape, generator=gen) * std

    def put(name, value):
      off, numel = self.store.offset[name]
      host[off:off + numel] = value.reshape(-1)

    sv = eng.stem_vars
    for key, std in ((eng.STEM + "/depthwise_weights", math.sqrt(2.0 / 49)),
                     (eng.STEM + "/pointwise_weights", math.sqrt(2.0 / (3 * eng.stem_mult)))):
      sv[key].copy_(normal(sv[key].shape, std).to(self._device))
    for net in (eng.first, eng.second):
      for L in net.layers.values():
        put(L.name + "/weights", normal((L.k, L.k, L.cin, L.cout), math.sqrt(2.0 / (L.k * L.k * L.cin))))
        if L.bn_scale:
          put(L.name + "/BatchNorm/gamma", torch.ones(L.cout))
        # (beta = 0: the image starts zeroed)
    init = self._model_proto.fc_hyperparams.initializer
    which = init.WhichOneof('initializer_oneof')
    std, mean = 0.01, 0.0
    if which == 'truncated_normal_initializer':
      std, mean = init.truncated_normal_initializer.stddev, init.truncated_normal_initializer.mean
    elif which == 'random_normal_initializer':
      std, mean = init.random_normal_initializer.stddev, init.random_normal_initializer.mean
    w = torch.empty(self.engine.feature_dims, self._ncols)
    torch.nn.init.trunc_normal_(w, mean=mean, std=std, a=mean - 2 * std, b=mean + 2 * std,
                                generator=gen)
    hw = torch.zeros(self.store.var[HEADS_W].shape)
    hw[:, :self._ncols] = w
    put(HEADS_W, hw)                          # (HEADS_B = 0)
    self.store.values.copy_(host.to(self._device))
    self.refresh()

  def refresh(self, only_trainable=False):
    """Re-derives kernel operands (transposed weights, folded BN) from the variables."""
    self.engine.refresh(only_trainable)     # (the fused heads operand rides in the same launch)

  def set_trainable(self, trainable_names):
    self.engine.set_trainable(set(trainable_names))

  # -- buffers ------------------------------------------------------------------------
  def _bufs(self, b, n):
    key = (b, n)
    if key not in self._cache:
      dev, c, k = self._device, self._num_classes, self._oicr_iterations
      self._cache[key] = dict(
          logits=torch.empty(b * n, self._npad, device=dev),
          dlogits=torch.zeros(b * n, self._npad, device=dev),
          proba=torch.empty(b, n, c, device=dev),
          class_logits=torch.empty(b, c, device=dev),
          dclass_logits=torch.empty(b, c, device=dev),
          scores0=torch.empty(b, n, c, device=dev),
          softmax=torch.empty(max(k, 1), b * n, c + 1, device=dev),      # [stage] planes
          idx=torch.empty(max(k, 1), b, c, dtype=torch.int32, device=dev),
          top_boxes=torch.empty(max(k, 1), b, c, 4, device=dev),
          dfeatures=torch.empty(b * n, self.engine.feature_dims, device=dev))
    return self._cache[key]

  # -- reference API ------------------------------------------------------------------
  def _build_prediction(self, examples, dropout_seed=None, dropout_mask=None,
                        feature_map_dropout_mask=None):
    """models/cap2det_model.py:152-216."""
    image = examples[InputDataFields.image]
    num_proposals = examples[InputDataFields.num_proposals]
    proposals = examples[InputDataFields.proposals]
    b, n = proposals.shape[0], proposals.shape[1]
    c, k = self._num_classes, self._oicr_iterations
    features, fctx = self.engine.forward(image, proposals, self._is_training, dropout_seed,
                                         dropout_mask, feature_map_dropout_mask)
    bufs = self._bufs(b, n)
    d = self.engine.feature_dims
    # all five heads in one GEMM: logits = X . W + b  (activation_fn=None, :79-88,:191-197)
    ops.conv_fwd(features, d, 0, self._heads_wt, None, self.store.var[HEADS_B], bufs["logits"],
                 self._npad, 0, b * n, 1, 1, d, self._npad, 1, 1, 1, False)
    ops.midn_fwd(bufs["logits"], self._npad, 0, c, num_proposals, bufs["proba"],
                 bufs["class_logits"], bufs["scores0"], b, n, c)
    predictions = {
        DetectionResultFields.class_labels: list(self._label_extractor.classes),
        DetectionResultFields.num_proposals: num_proposals,
        DetectionResultFields.proposal_boxes: proposals,
        Cap2DetPredictions.midn_class_logits: bufs["class_logits"],
        Cap2DetPredictions.midn_proba_r_given_c: bufs["proba"],
        Cap2DetPredictions.oicr_proposal_scores + '_at_0': bufs["scores0"],
    }
    lg = bufs["logits"].view(b, n, self._npad)
    for i in range(k):
      off = 2 * c + i * (c + 1)
      predictions[Cap2DetPredictions.oicr_proposal_scores + '_at_{}'.format(i + 1)] = \
          lg[:, :, off:off + c + 1]
    self._ctx = dict(fctx=fctx, bufs=bufs, b=b, n=n, features=features,
                     num_proposals=num_proposals, proposals=proposals)
    return predictions

  def _postprocess(self, inputs, predictions):
    """models/cap2det_model.py:111-150: per-class NMS of the MIDN proposal scores (iteration 0)
    and of softmax(OICR scores)[..., 1:] (iterations 1..K)."""
    results = {}
    k, c = self._oicr_iterations, self._num_classes
    proposals = predictions[DetectionResultFields.proposal_boxes]
    b, n = proposals.shape[0], proposals.shape[1]
    for i in range(1 + k):
      scores = predictions[Cap2DetPredictions.oicr_proposal_scores + '_at_{}'.format(i)]
      post_process_fn = self._midn_postprocess_fn
      if i > 0:
        post_process_fn = self._oicr_postprocess_fn
        base = scores._base if scores._base is not None else scores
        ld = base.shape[-1]
        off = scores.storage_offset() - base.storage_offset()   # column slice of a wider buffer
        probs = torch.empty(b, n, c, device=self._device)
        ops.softmax_drop_background(base, ld, off, b * n, c + 1, probs)
        scores = probs
      num, boxes, sc, classes, _ = post_process_fn(proposals, scores.contiguous())
      results[DetectionResultFields.num_detections + '_at_{}'.format(i)] = num
      results[DetectionResultFields.detection_boxes + '_at_{}'.format(i)] = boxes
      results[DetectionResultFields.detection_scores + '_at_{}'.format(i)] = sc
      results[DetectionResultFields.detection_classes + '_at_{}'.format(i)] = classes
    return results

  def build_prediction(self, examples, **kwargs):
    """models/cap2det_model.py:218-272.  Training mode / no eval_min_dimension: one pass (the
    training step skips the NMS outputs, which `train_op` never consumes: pass
    postprocess=True to get them).  Evaluation with `eval_min_dimension`: one forward per
    resolution (TF1 legacy-bilinear resize of the single image to the given minimum side,
    core/imgproc.py:300-353), the proposal scores of every OICR iteration averaged over the
    resolutions, then the post-processing."""
    options = self._model_proto
    if self._is_training or len(options.eval_min_dimension) == 0 or kwargs.get("single_scale"):
      predictions = self._build_prediction(examples, kwargs.get("dropout_seed"),
                                           kwargs.get("dropout_mask"),
                                           kwargs.get("feature_map_dropout_mask"))
      if kwargs.get("postprocess", not self._is_training):
        predictions.update(self._postprocess(examples, predictions))
      return predictions

    inputs = examples[InputDataFields.image]
    assert inputs.shape[0] == 1
    k, c = self._oicr_iterations, self._num_classes
    b, n = examples[InputDataFields.proposals].shape[:2]
    key = ("ms", b, n)
    if key not in self._cache:
      self._cache[key] = [torch.empty(b, n, c if i == 0 else c + 1, device=self._device)
                          for i in range(1 + k)]
    sums = self._cache[key]
    examples = dict(examples)
    dims = list(options.eval_min_dimension)
    predictions = None
    for si, min_dimension in enumerate(dims):
      h, w = int(inputs.shape[1]), int(inputs.shape[2])
      oh, ow = resize_to_min_dimension_size(h, w, min_dimension)
      resized = ops.resize_bilinear(inputs[0].contiguous(), oh, ow)
      examples[InputDataFields.image] = resized.unsqueeze(0)
      predictions = self._build_prediction(examples)
      for i in range(1 + k):
        sc = predictions[Cap2DetPredictions.oicr_proposal_scores + '_at_{}'.format(i)]
        base = sc._base if sc._base is not None else sc
        ops.scores_accumulate(sums[i], base, base.shape[-1],
                              sc.storage_offset() - base.storage_offset(), b * n, sc.shape[-1],
                              si == 0)
    predictions_aggregated = dict(predictions)
    for i in range(1 + k):
      ops.scores_divide(sums[i], float(len(dims)))
      predictions_aggregated[Cap2DetPredictions.oicr_proposal_scores + '_at_{}'.format(i)] = sums[i]
    predictions_aggregated.update(self._postprocess(inputs, predictions_aggregated))
    return predictions_aggregated

  def build_loss(self, predictions, examples=None, **kwargs):
    """models/cap2det_model.py:274-330.  Also leaves d(loss)/d(head logits) in the step
    context for `backward()`."""
    options = self._model_proto
    ctx = self._ctx
    if ctx is None:
      raise RuntimeError("build_prediction must run before build_loss")
    bufs, b, n = ctx["bufs"], ctx["b"], ctx["n"]
    c, k = self._num_classes, self._oicr_iterations
    labels = kwargs.get("labels")
    if labels is None:
      labels = self._label_extractor.extract_labels(examples)
    elif kwargs.get("labels_ready") is not None:     # extracted on another stream (Trainer)
      torch.cuda.current_stream().wait_event(kwargs["labels_ready"])
    ctx["labels"] = labels
    losses = self._losses
    if not kwargs.get("step_zeroed"):        # (the trainer zeroes the step's buffers in one launch)
      losses.zero_()
    num_proposals, proposals = ctx["num_proposals"], ctx["proposals"]
    ops.sigmoid_ce_fwd_bwd(bufs["class_logits"], labels, options.midn_loss_weight, losses[0:1],
                           bufs["dclass_logits"])
    ops.midn_bwd(bufs["dclass_logits"], bufs["logits"], self._npad, 0, c, num_proposals,
                 bufs["proba"], bufs["class_logits"], bufs["dlogits"], self._npad, b, n, c)
    loss_dict = {'midn_cross_entropy_loss': losses[0]}
    # s0 = concat(0, proba or scores): the class columns are searched directly (:306-312)
    s0 = bufs["proba"] if options.oicr_use_proba_r_given_c else bufs["scores0"]
    s0_ld, s0_off = c, 0
    if k > 0:
      # all stages in three launches: stage i + 1 selects on softmax(scores_i)[..., 1:] (:328),
      # which depends on the forward pass only, not on the loss of stage i
      ops.oicr_refine_fwd_bwd(bufs["logits"], self._npad, 2 * c, k, s0, s0_ld, s0_off, proposals,
                              labels, num_proposals, options.oicr_iou_threshold,
                              options.oicr_loss_weight, b, n, c, losses[1:1 + k], bufs["dlogits"],
                              self._npad, 2 * c, bufs["softmax"], bufs["idx"], bufs["top_boxes"])
      for i in range(k):
        loss_dict['oicr_cross_entropy_loss_at_{}'.format(i + 1)] = losses[i + 1]
      return loss_dict
    return loss_dict

  def regularization_loss(self, step_zeroed=False):
    """Sum of the slim L2 / L1 regularisers (FC weights only), as a 0-d tensor."""
    out = self._losses[self._nloss - 1:self._nloss]
    if not step_zeroed:
      out.zero_()
    if self._l2_weight > 0:
      ops.l2_loss(self.store.var[HEADS_W], self._l2_weight, out)
    if self._l1_weight > 0:
      ops.l1_loss(self.store.var[HEADS_W], self._l1_weight, out)
    return self._losses[self._nloss - 1]

  def total_loss(self):
    """total_loss = sum of the model losses + regularisers (train/trainer.py:55-61) as a 0-d
    device tensor (c2d_sum_small over the loss vector)."""
    ops.sum_small(self._losses[:self._nloss], self._losses[self._nloss:self._nloss + 1])
    return self._losses[self._nloss]

  def backward(self, after_second_stage=None, after_block=None):
    """Gradients of sum(losses) w.r.t. every trainable variable, accumulated into
    `self.store.grads` (caller zeroes it once per step).  after_second_stage: callable invoked
    once the head and second-stage gradients are final (the data-parallel reducer starts its
    big all-reduce there, under the ROI-crop / first-stage backward)."""
    ctx = self._ctx
    bufs, b, n = ctx["bufs"], ctx["b"], ctx["n"]
    d = self.engine.feature_dims
    g = self.store.grad
    x = ctx["features"]
    def head_variable_gradients():
      ops.conv_wgrad(x, d, 0, bufs["dlogits"], self._npad, 0, g[HEADS_W], b * n, 1, 1, d, self._npad,
                     1, 1, 1)
      ops.col_sum(bufs["dlogits"], self._npad, 0, g[HEADS_B], b * n, self._npad)

    # the heads' filter / bias gradients only meet the rest of the step at the gradient exchange and
    # the optimiser: like the second stage's filter gradients they go out on the filter-gradient
    # stream, beside the input gradient that the backward pass is waiting for (the 416-column heads
    # of the 80-class configs: 41 us off the main stream)
    side = self.engine.second.side
    heads_done = None
    if side is not None:
      fork = torch.cuda.Event()
      fork.record()
      side.wait_event(fork)
      with torch.cuda.stream(side):
        head_variable_gradients()
        heads_done = torch.cuda.Event()
        heads_done.record()
    else:
      head_variable_gradients()
    ops.conv_dgrad(bufs["dlogits"], self._npad, 0, self.store.var[HEADS_W], bufs["dfeatures"], d, 0,
                   b * n, 1, 1, d, self._npad, 1, 1, 1, False)
    if heads_done is not None:
      # whoever hands the heads' gradients on (the data-parallel reducers' hooks) does so behind them
      user_after2, user_block = after_second_stage, after_block

      if user_after2 is not None:
        def after_second_stage():
          torch.cuda.current_stream().wait_event(heads_done)
          user_after2()

      if user_block is not None:
        def after_block(i):
          torch.cuda.current_stream().wait_event(heads_done)
          user_block(i)
    self.engine.backward(bufs["dfeatures"], d, 0, ctx["fctx"], after_second_stage, after_block)
    if heads_done is not None:
      torch.cuda.current_stream().wait_event(heads_done)

  def build_evaluation(self, predictions, examples=None, **kwargs):
    """models/cap2det_model.py:332-343 returns {} in the reference."""
    return {}


def resize_to_min_dimension_size(height, width, min_dimension):
  """core/imgproc.py:329-343 `_compute_new_dynamic_size`: fp32 scale, tf.round (half to even)."""
  scale = np.float32(min_dimension) / np.float32(min(height, width))
  return (int(np.round(np.float32(height) * scale)), int(np.round(np.float32(width) * scale)))


register_model_class(cap2det_model_pb2.Cap2DetModel.ext, Model)
