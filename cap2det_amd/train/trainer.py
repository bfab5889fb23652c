"""Training step (reference: train/trainer.py:30-172 `_model_fn` in TRAIN mode,
core/training_utils.py:14-71 `build_optimizer`).

total_loss = sum(model losses) + sum(L2 regularisers); learning rate = exponential_decay;
Adagrad; gradient multipliers by scope prefix (later entries override, <= 0 freezes).
The reference distributes with an asynchronous TF parameter server (train_wsod.sh:46-88);
here every rank owns one MI355X, gradients are summed with ONE RCCL all-reduce over the flat
gradient bucket and divided by the world size (synchronous data parallel, equivalent to the
reference's `SyncReplicasOptimizer` option, train/trainer.py:90-94).
"""
import math

import torch

from cap2det_amd import hip_ops as ops
from cap2det_amd.models import builder
from cap2det_amd.models.cap2det_model import HEADS_B, HEADS_W
from cap2det_amd.protos import pipeline_pb2
from cap2det_amd.protos.message import unwrap
from cap2det_amd.train import data_parallel


def resolve_gradient_multipliers(var_names, gradient_multipliers):
  """train/trainer.py:104-125.  Returns {name: multiplier} for variables that stay trainable
  (1.0 when no scope matches)."""
  out = {}
  for name in var_names:
    trainable, mult = True, 1.0
    for gm in gradient_multipliers:
      if name.startswith(gm.scope):
        mult = gm.multiplier
        trainable = gm.multiplier > 0
    if trainable:
      out[name] = mult
  return out


def exponential_decay(lr, step, decay_steps, decay_rate, staircase):
  """tf.train.exponential_decay (train/trainer.py:76-82)."""
  p = step / float(decay_steps)
  if staircase:
    p = math.floor(p)
  return lr * (decay_rate ** p)


class Trainer(object):
  """Owns the model, the optimiser state and the data-parallel reduction."""

  def __init__(self, pipeline_proto, device="cuda:0", model=None, **model_kwargs):
    pipeline_proto = unwrap(pipeline_proto)
    if not isinstance(pipeline_proto, pipeline_pb2.Pipeline):
      raise ValueError('pipeline_proto has to be an instance of Pipeline.')
    self.pipeline = pipeline_proto
    self.train_config = pipeline_proto.train_config
    self.device = torch.device(device)
    self.model = model if model is not None else builder.build(
        pipeline_proto.model, is_training=True, device=device, **model_kwargs)
    opt = self.train_config.optimizer.WhichOneof('optimizer')
    if opt != 'adagrad':
      raise ValueError('Invalid optimizer: {}.'.format(opt) if opt is None else
                       'optimizer %s is not implemented on the HIP path (configs use adagrad)' % opt)
    if self.train_config.HasField('max_gradient_norm'):
      raise NotImplementedError('max_gradient_norm is unset in every shipped config')
    self.global_step = 0
    store = self.model.store
    store.accum.fill_(self.train_config.optimizer.adagrad.initial_accumulator_value)
    # gradient multipliers on the reference variable names; fused head buffers inherit the
    # multiplier of their first head (all heads share scope-less names midn/*, oicr/*).
    names = self.model.get_variables_to_train()
    mult = resolve_gradient_multipliers(names, self.train_config.gradient_multiplier)
    self.multipliers = mult
    self.model.set_trainable(mult.keys())
    head_names = [n for n in names if n.startswith("midn/") or n.startswith("oicr/")]
    head_mults = set(mult.get(n, 0.0) for n in head_names)
    if len(head_mults) != 1:
      raise NotImplementedError("per-head gradient multipliers are not supported (fused heads)")
    head_mult = head_mults.pop()
    # Adagrad segments over the flat buffers: runs of consecutive variables sharing (mult, l2).
    segs = []
    for name in store.names():
      if name == HEADS_W:
        m, l2 = head_mult, self.model.l2_weight
      elif name == HEADS_B:
        m, l2 = head_mult, 0.0
      else:
        m, l2 = mult.get(name, 0.0), 0.0
      off, numel = store.offset[name]
      end = off + -(-numel // store.ALIGN) * store.ALIGN
      if m <= 0:
        continue
      if segs and segs[-1][1] == off and segs[-1][2] == m and segs[-1][3] == l2:
        segs[-1][1] = end
      else:
        segs.append([off, end, m, l2])
    self.segments = segs
    if not segs:
      raise ValueError("no trainable variables")
    self.bucket = (min(s[0] for s in segs), max(s[1] for s in segs))
    self.rank, self.world_size = data_parallel.world_info()

  def learning_rate(self):
    tc = self.train_config
    lr = tc.learning_rate
    if tc.HasField('learning_rate_decay'):
      d = tc.learning_rate_decay
      lr = exponential_decay(lr, self.global_step, d.decay_steps, d.decay_rate, d.staircase)
    return lr

  def train_step(self, examples, **kwargs):
    """One synchronous step; returns {loss name: 0-d device tensor} (+ 'total_loss',
    'regularization_loss').  No host synchronisation happens inside."""
    model, store = self.model, self.model.store
    lo, hi = self.bucket
    store.grads[lo:hi].zero_()
    predictions = model.build_prediction(examples, **kwargs)
    losses = dict(model.build_loss(predictions, examples=examples, **kwargs))
    reg = model.regularization_loss()
    model.backward()
    scale = data_parallel.allreduce_bucket(store.grads[lo:hi])
    lr = self.learning_rate()
    for off, end, m, l2 in self.segments:
      ops.adagrad_step(store.values[off:end], store.grads[off:end], store.accum[off:end], lr, l2,
                       m, scale)
    model.refresh(only_trainable=True)
    self.global_step += 1
    losses['regularization_loss'] = reg
    losses['total_loss'] = model._losses.sum()
    self.predictions = predictions
    return losses
