"""Text-classifier pre-training model (reference: models/text_model.py:
31-129, models/label_extractor.py:353-421, configs/coco17_text.pbtxt).

caption tokens -> frozen GloVe rows -> FC(300 -> hidden) -> masked max over the non-OOV tokens
-> ReLU -> dropout -> FC(hidden -> classes); loss = mean sigmoid cross entropy against the
ground-truth multi-label vector + slim L2 regularisers (weight * sum(w^2) / 2 on both FC
weights); Adagrad.  The FC layers run on the MFMA GEMM kernels, pooling / dropout / their
gradients on `c2d_text_pool_fwd/bwd`.  The weights it trains are what
`TextClassifierMatchExtractor.load_weights` consumes in the detection model.
"""
import math

import numpy as np
import torch

from cap2det_amd import hip_ops as ops
from cap2det_amd.core.standard_fields import InputDataFields
from cap2det_amd.models import label_extractor
from cap2det_amd.models.model_base import ModelBase
from cap2det_amd.models.registry import register_model_class
from cap2det_amd.protos import cap2det_model_pb2
from cap2det_amd.protos.message import unwrap

FIELD_LOGITS = 'logits'
FIELD_TEXT_LOSS = 'text_cross_entropy_loss'
W1, B1 = "text_classifier/layer1/weights", "text_classifier/layer1/biases"
W2, B2 = "text_classifier/layer2/weights", "text_classifier/layer2/biases"


class Model(ModelBase):
  """Text model."""

  def __init__(self, model_proto, is_training=False, device="cuda:0", seed=0):
    model_proto = unwrap(model_proto)
    super(Model, self).__init__(model_proto, is_training)
    if not isinstance(model_proto, cap2det_model_pb2.TextModel):
      raise ValueError('The model_proto has to be an instance of TextModel.')
    options = model_proto
    self._device = torch.device(device)
    self._label_extractor = label_extractor.GroundtruthExtractor(options.label_extractor, device)
    self._text_classifier = label_extractor.TextClassifierMatchExtractor(options.text_classifier,
                                                                        device)
    tc = options.text_classifier
    self._hidden = tc.hidden_units
    self._keep = tc.dropout_keep_proba
    self._reg = tc.regularizer
    self._classes = self._text_classifier.num_classes
    emb = self._text_classifier._embedding
    self._vocab, self._edims = emb.shape[0] - 1, emb.shape[1]
    self._epad = -(-self._edims // 16) * 16          # GEMM reduction dims are multiples of 16
    self._cpad = -(-self._classes // 16) * 16       # (the class axis is a reduction dim in dgrad)
    dev = self._device
    self.vars = {W1: torch.zeros(self._epad, self._hidden, device=dev),
                 B1: torch.zeros(self._hidden, device=dev),
                 W2: torch.zeros(self._hidden, self._cpad, device=dev),
                 B2: torch.zeros(self._cpad, device=dev)}
    self.grads = {k: torch.zeros_like(v) for k, v in self.vars.items()}
    self._wt1 = torch.zeros(1, self._hidden, self._epad, device=dev)
    self._wt2 = torch.zeros(1, self._cpad, self._hidden, device=dev)
    self._losses = torch.zeros(2, device=dev)
    self._ctx = None
    self.initialize(seed)

  @property
  def num_classes(self):
    return self._classes

  def initialize(self, seed=0):
    """slim.fully_connected defaults: Glorot-uniform weights, zero biases."""
    gen = torch.Generator(device="cpu").manual_seed(seed)
    for name, fan_in, fan_out in ((W1, self._edims, self._hidden), (W2, self._hidden, self._classes)):
      lim = math.sqrt(6.0 / (fan_in + fan_out))
      w = (torch.rand(fan_in, fan_out, generator=gen) * 2 - 1) * lim
      self.vars[name].zero_()
      self.vars[name][:fan_in, :fan_out] = w.to(self._device)
    self.vars[B1].zero_(); self.vars[B2].zero_()
    self.refresh()

  def refresh(self):
    ops.transpose_taps(self.vars[W1], self._wt1, 1, self._epad, self._hidden)
    ops.transpose_taps(self.vars[W2], self._wt2, 1, self._hidden, self._cpad)

  def state_dict(self):
    """Under the reference's variable names (models/label_extractor.py:455-457)."""
    return {W1: self.vars[W1][:self._edims].cpu().numpy().copy(),
            B1: self.vars[B1].cpu().numpy().copy(),
            W2: self.vars[W2][:, :self._classes].cpu().numpy().copy(),
            B2: self.vars[B2][:self._classes].cpu().numpy().copy()}

  def load_state_dict(self, arrays):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(self._device)
    for v in self.vars.values():
      v.zero_()
    self.vars[W1][:self._edims] = t(arrays[W1]); self.vars[B1].copy_(t(arrays[B1]))
    self.vars[W2][:, :self._classes] = t(arrays[W2]); self.vars[B2][:self._classes] = t(arrays[B2])
    self.refresh()

  def build_prediction(self, examples, dropout_mask=None, dropout_seed=0, **kwargs):
    """models/text_model.py:53-66 (+ label_extractor.py:353-421)."""
    ids = self._text_classifier._ids(examples)                   # [B, T] int32 (host lookup)
    b, t = ids.shape
    h, c, dev = self._hidden, self._cpad, self._device
    x = torch.empty(b * t, self._epad, device=dev)
    ops.embedding_gather(ids, self._text_classifier._embedding, self._epad, x)
    pre = torch.empty(b * t, h, device=dev)
    ops.conv_fwd(x, self._epad, 0, self._wt1, None, self.vars[B1], pre, h, 0, b * t, 1, 1,
                 self._epad, h, 1, 1, 1, False)
    mask = None
    if self._is_training and self._keep < 1.0:
      if dropout_mask is not None:
        mask = dropout_mask.to(dev).contiguous()
      else:
        mask = torch.empty(b, h, dtype=torch.uint8, device=dev)
        ops.dropout_mask(mask, dropout_seed, self._keep)
    hidden = torch.empty(b, h, device=dev)
    ops.text_pool_fwd(pre, ids, h, self._vocab, mask, self._keep if mask is not None else 1.0,
                      hidden)
    logits = torch.empty(b, c, device=dev)
    ops.conv_fwd(hidden, h, 0, self._wt2, None, self.vars[B2], logits, c, 0, b, 1, 1, h, c, 1, 1,
                 1, False)
    self._ctx = dict(ids=ids, x=x, pre=pre, mask=mask, hidden=hidden, logits=logits, b=b, t=t)
    return {FIELD_LOGITS: logits[:, :self._classes]}

  def build_loss(self, predictions, examples, **kwargs):
    """models/text_model.py:68-84: reduce_mean of the element-wise sigmoid cross entropy."""
    ctx = self._ctx
    labels = kwargs.get("labels")
    if labels is None:
      labels = self._label_extractor.extract_labels(examples)
    b, c = ctx["b"], self._classes
    lg = ctx["logits"][:, :c].contiguous()
    dl = torch.empty(b, c, device=self._device)
    self._losses.zero_()
    ops.sigmoid_ce_fwd_bwd(lg, labels.contiguous(), 1.0, self._losses[0:1], dl)
    ctx["dlogits"] = torch.zeros(b, self._cpad, device=self._device)
    ctx["dlogits"][:, :c] = dl
    ctx["labels"] = labels
    return {FIELD_TEXT_LOSS: self._losses[0]}

  def regularization_loss(self):
    out = self._losses[1:2]
    out.zero_()
    if self._reg > 0:
      ops.l2_loss(self.vars[W1], self._reg, out)
      ops.l2_loss(self.vars[W2], self._reg, out)
    return self._losses[1]

  def backward(self):
    ctx = self._ctx
    b, t, h, c = ctx["b"], ctx["t"], self._hidden, self._cpad
    for g in self.grads.values():
      g.zero_()
    dl = ctx["dlogits"]
    ops.conv_wgrad(ctx["hidden"], h, 0, dl, c, 0, self.grads[W2].view(1, 1, h, c), b, 1, 1, h, c, 1,
                   1, 1)
    ops.col_sum(dl, c, 0, self.grads[B2], b, c)
    dh = torch.empty(b, h, device=self._device)
    ops.conv_dgrad(dl, c, 0, self.vars[W2].view(1, 1, h, c), dh, h, 0, b, 1, 1, h, c, 1, 1, 1, False)
    dpre = torch.empty(b * t, h, device=self._device)
    ops.text_pool_bwd(dh, ctx["pre"], ctx["ids"], h, self._vocab, ctx["mask"],
                      self._keep if ctx["mask"] is not None else 1.0, dpre)
    ops.conv_wgrad(ctx["x"], self._epad, 0, dpre, h, 0, self.grads[W1].view(1, 1, self._epad, h),
                   b * t, 1, 1, self._epad, h, 1, 1, 1)
    ops.col_sum(dpre, h, 0, self.grads[B1], b * t, h)

  def build_evaluation(self, predictions, examples, **kwargs):
    """models/text_model.py:86-126 for ONE batch: the confusion counts the streaming
    tf.metrics accumulate (precision / recall at sigmoid thresholds, precision@k / recall@k)."""
    logits = predictions[FIELD_LOGITS].detach().cpu().numpy()
    labels = self._label_extractor.extract_labels(examples).cpu().numpy() > 0
    out = {}
    p = 1.0 / (1.0 + np.exp(-logits))
    for thr in (0.3, 0.5, 0.7):
      pred = p > thr
      out['counts/threshold_{}'.format(thr)] = (int((pred & labels).sum()), int(pred.sum()),
                                                int(labels.sum()))
    order = np.argsort(-logits, axis=1, kind="stable")
    for k in (1, 5):
      top = np.zeros_like(labels)
      np.put_along_axis(top, order[:, :k], True, axis=1)
      out['counts/top_{}'.format(k)] = (int((top & labels).sum()), int(top.sum()), int(labels.sum()))
    return out


class MetricAccumulator(object):
  """Streaming precision / recall as tf.metrics.* keeps them (true positives / predicted /
  actual counts summed over batches)."""

  def __init__(self):
    self.counts = {}

  def update(self, batch_counts):
    for k, (tp, pred, act) in batch_counts.items():
      a = self.counts.setdefault(k, [0, 0, 0])
      a[0] += tp; a[1] += pred; a[2] += act

  def result(self):
    out = {}
    for k, (tp, pred, act) in self.counts.items():
      name = k.split('/', 1)[1]
      suffix = 'at_' + name.split('_', 1)[1]
      out['metrics/precision_' + suffix] = tp / pred if pred else 0.0
      out['metrics/recall_' + suffix] = tp / act if act else 0.0
    return out


class TextTrainer(object):
  """train/trainer.py `_model_fn` for the text model: total = loss + L2, Adagrad."""

  def __init__(self, pipeline_proto, device="cuda:0", seed=0):
    from cap2det_amd.models import builder
    from cap2det_amd.train.trainer import exponential_decay
    pipeline_proto = unwrap(pipeline_proto)
    self.train_config = pipeline_proto.train_config
    self.model = builder.build(pipeline_proto.model, is_training=True, device=device, seed=seed)
    if self.train_config.optimizer.WhichOneof('optimizer') != 'adagrad':
      raise ValueError('the text model is trained with adagrad (configs/coco17_text.pbtxt)')
    init = self.train_config.optimizer.adagrad.initial_accumulator_value
    self.accum = {k: torch.full_like(v, init) for k, v in self.model.vars.items()}
    self.global_step = 0
    self._decay = exponential_decay

  def learning_rate(self):
    tc = self.train_config
    lr = tc.learning_rate
    if tc.HasField('learning_rate_decay'):
      d = tc.learning_rate_decay
      lr = self._decay(lr, self.global_step, d.decay_steps, d.decay_rate, d.staircase)
    return lr

  def train_step(self, examples, **kwargs):
    m = self.model
    pred = m.build_prediction(examples, dropout_seed=kwargs.pop("dropout_seed", self.global_step),
                              **kwargs)
    losses = dict(m.build_loss(pred, examples, **kwargs))
    losses['regularization_loss'] = m.regularization_loss()
    m.backward()
    lr = self.learning_rate()
    for k, v in m.vars.items():
      l2 = m._reg if k in (W1, W2) else 0.0
      ops.adagrad_step(v.view(-1), m.grads[k].view(-1), self.accum[k].view(-1), lr, l2, 1.0, 1.0)
    m.refresh()
    self.global_step += 1
    losses['total_loss'] = m._losses.sum()
    return losses


register_model_class(cap2det_model_pb2.TextModel.ext, Model)
