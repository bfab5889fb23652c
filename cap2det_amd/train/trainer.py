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
import time

import os
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


def dropout_key(seed, global_step, rank, world_size):
  """Key of the counter-based dropout generator for one step of one worker: a splitmix64 hash of
  (trainer seed, global step, rank), so that runs started with different seeds draw different mask
  sequences and no two (step, rank) pairs of one run share a mask.  Kept below 2^63: the hipGraph
  path stores it in an int64 device word."""
  m = (1 << 64) - 1
  x = (int(seed) * 0x9E3779B97F4A7C15 + int(global_step) * int(world_size) + int(rank)) & m
  x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & m
  x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & m
  return (x ^ (x >> 31)) & ((1 << 63) - 1)


class Trainer(object):
  """Owns the model, the optimiser state and the data-parallel reduction."""

  def __init__(self, pipeline_proto, device="cuda:0", model=None, use_plan=True,
               **model_kwargs):
    """use_plan: record the launch list of a step once per input signature and replay it with one
    native call (cap2det_amd/step_plan.py, csrc/plan.hip) instead of queueing ~140 ctypes calls
    from Python every step; steps a plan cannot express (data-parallel hooks, injected dropout
    masks, optimisers other than the one-launch Adagrad, the first and last steps of a run) run
    eagerly.  Replayed and eager steps launch the same kernels with the same arguments."""
    pipeline_proto = unwrap(pipeline_proto)
    if not isinstance(pipeline_proto, pipeline_pb2.Pipeline):
      raise ValueError('pipeline_proto has to be an instance of Pipeline.')
    self.seed = int(model_kwargs.get("seed", 0) or 0)
    self.pipeline = pipeline_proto
    self.train_config = pipeline_proto.train_config
    self.device = torch.device(device)
    self.model = model if model is not None else builder.build(
        pipeline_proto.model, is_training=True, device=device, **model_kwargs)
    # core/training_utils.py:14-71 `build_optimizer`: sgd, momentum, adagrad (every shipped
    # config), adam, rmsprop — TensorFlow 1.x update rules (c2d_adagrad_step / c2d_optimizer_step)
    opt = self.train_config.optimizer.WhichOneof('optimizer')
    if opt not in ('sgd', 'momentum', 'adagrad', 'adam', 'rmsprop'):
      raise ValueError('Invalid optimizer: {}.'.format(opt))
    self.opt_kind = opt
    self.opt_options = getattr(self.train_config.optimizer, opt)
    self.global_step = 0
    store = self.model.store
    # slot buffers (flat, mirroring the variables): `accum` is slot 0
    store.slots = [store.accum]
    if opt == 'adagrad':
      store.accum.fill_(self.opt_options.initial_accumulator_value)
    elif opt == 'momentum':
      store.accum.zero_()
    elif opt == 'adam':
      store.accum.zero_()
      store.slots.append(torch.zeros_like(store.accum))                 # m, v
    elif opt == 'rmsprop':
      store.accum.fill_(1.0)                                            # TF: "rms" slot = ones
      store.slots.append(torch.zeros_like(store.accum))                 # momentum
      if self.opt_options.centered:
        store.slots.append(torch.zeros_like(store.accum))               # mg
    # gradient multipliers on the reference variable names (train/trainer.py:104-125)
    names = self.model.get_variables_to_train()
    mult = resolve_gradient_multipliers(names, self.train_config.gradient_multiplier)
    self.multipliers = mult
    self.model.set_trainable(mult.keys())
    # The five heads live in ONE fused [D][npad] buffer.  With one multiplier for all of them
    # (every shipped config) the buffer is an ordinary segment; otherwise a per-column multiplier
    # vector drives c2d_adagrad_step_ex (columns of frozen heads and the zero padding get 0).
    head_cols = self.model.head_columns()
    npad = store.var[HEADS_W].shape[1]
    wcol, bcol = [0.0] * npad, [0.0] * npad
    for hname, off, width in head_cols:
      for c in range(off, off + width):
        wcol[c] = mult.get(hname + "/weights", 0.0)
        bcol[c] = mult.get(hname + "/biases", 0.0)
    real = [c for _, off, width in head_cols for c in range(off, off + width)]
    uniform = len(set(wcol[c] for c in real) | set(bcol[c] for c in real)) == 1
    l1, l2 = self.model.l1_weight, self.model.l2_weight
    # Adagrad segments over the flat buffers: runs of consecutive variables sharing
    # (multiplier, l1, l2); [off, end, mult, l1, l2, column multipliers or None, their 0/1 mask]
    segs = []
    for name in store.names():
      cols = None
      if name == HEADS_W:
        m, r1, r2 = (wcol[real[0]] if uniform else 1.0), l1, l2
        cols = None if uniform else wcol
      elif name == HEADS_B:
        m, r1, r2 = (bcol[real[0]] if uniform else 1.0), 0.0, 0.0
        cols = None if uniform else bcol
      else:
        m, r1, r2 = mult.get(name, 0.0), 0.0, 0.0
      off, numel = store.offset[name]
      end = off + -(-numel // store.ALIGN) * store.ALIGN
      if m <= 0 or (cols is not None and max(cols) <= 0):
        continue
      if cols is not None:
        cols = torch.tensor(cols, dtype=torch.float32, device=self.device)
      if (segs and cols is None and segs[-1][5] is None and segs[-1][1] == off and
          segs[-1][2:5] == [m, r1, r2]):
        segs[-1][1] = end
      else:
        segs.append([off, end, m, r1, r2, cols,
                     None if cols is None else (cols > 0).to(torch.float32)])
    self.segments = segs
    if not segs:
      raise ValueError("no trainable variables")
    self.bucket = (min(s[0] for s in segs), max(s[1] for s in segs))
    # tf.contrib.training.clip_gradient_norms (train/trainer.py:132-136): one descriptor per
    # reference variable (each head is its own variable = a column window of the fused buffer)
    self._clip = None
    if self.train_config.HasField('max_gradient_norm'):
      recs = []
      d = store.var[HEADS_W].shape[0]
      for name in store.names():
        off, numel = store.offset[name]
        if name == HEADS_W:
          for hname, coff, width in head_cols:
            m = mult.get(hname + "/weights", 0.0)
            if m > 0:
              recs.append((off + coff, d, width, npad, l1, l2, m))
        elif name == HEADS_B:
          for hname, coff, width in head_cols:
            m = mult.get(hname + "/biases", 0.0)
            if m > 0:
              recs.append((off + coff, 1, width, npad, 0.0, 0.0, m))
        elif mult.get(name, 0.0) > 0:
          recs.append((off, 1, numel, numel, 0.0, 0.0, mult[name]))
      desc, num = ops.clip_descriptors(recs, self.device)
      self._clip = (desc, num, float(self.train_config.max_gradient_norm))
    self._lr_dev = None
    self.rank, self.world_size = data_parallel.world_info()
    # first flat offset of the second stage: everything from there on (second stage + heads) is
    # final before the ROI-crop / first-stage backward starts (data_parallel.OverlappedReducer)
    second = [store.offset[n][0] for n in store.names()
              if n.startswith("second_stage_feature_extraction")]
    lo = self.bucket[0]
    self._tail_split = (min(second) - lo) if second and min(second) > lo else 0
    # per-block ranges of the bucket (data_parallel.BlockReducer): one cut where the variables of
    # each second-stage block begin; the last block's range runs to the end (the heads)
    net2 = self.model.engine.second
    starts = {}
    for i, names in enumerate(net2.order):
      offs = [store.offset[v][0] for n in names for v in net2.layers[n].var_names()]
      if offs and lo <= min(offs) < self.bucket[1]:
        starts[i] = min(offs) - lo
    cuts = sorted(set([0] + list(starts.values()) + [self.bucket[1] - lo]))
    self._block_cuts = cuts
    self._block_range = {i: cuts.index(o) for i, o in starts.items()}
    self.use_plan = bool(use_plan) and self.device.type == "cuda"
    # (planned steps read and write the same buffers every step: the look-ahead's prefix is copied
    #  into place instead of swapped in, FrcnnEngine.forward)
    self.model.engine.static_prefix = self.use_plan
    self._plans = {}            # input signature -> dict(plan, eager steps seen, look-ahead phase)
    self._announced = None      # (image tensor, version) the last step ran its look-ahead for
    self._next_labels = None    # labels of the announced batch, extracted under the last step
    self.plan_replays = 0       # steps issued by c2d_plan_replay so far (tests / bench)
    self.replay_s = 0.0
    self._plan_generation = self.model.engine.generation
    self._last_replayed = False

  # -- checkpoint / resume (reference: tf.estimator saves `model.ckpt-<step>` in model_dir,
  #    train/trainer.py:174-208; here one .npz per step under the reference variable names) ----
  def save_checkpoint(self, model_dir):
    import os
    import numpy as np
    os.makedirs(model_dir, exist_ok=True)
    path = os.path.join(model_dir, "model.ckpt-%d" % self.global_step)
    state = self.model.state_dict()
    store = self.model.store
    # written under a temporary name and renamed: a concurrent evaluator polling model_dir
    # (train/predict.py) or a resume after a crash never sees a half-written file
    tmp = path + ".tmp-%d.npz" % os.getpid()
    with open(tmp, "wb") as f:
      extra = {"__optimizer_slot%d" % i: sl.detach().cpu().numpy()
               for i, sl in enumerate(store.slots) if i > 0}
      np.savez(f, __global_step=np.int64(self.global_step),
               __adagrad_accumulators=store.accum.detach().cpu().numpy(), **extra, **state)
      f.flush()
      os.fsync(f.fileno())
    os.replace(tmp, path + ".npz")
    return path

  def export_tf_checkpoint(self, prefix):
    """Writes the model variables (reference names) + `global_step` as a TensorFlow V2 checkpoint
    `prefix.index` / `prefix.data-00000-of-00001`, the kind tf.estimator leaves in model_dir and
    train/predict.py restores (cap2det_amd/train/tf_checkpoint.py)."""
    import numpy as np
    from cap2det_amd.train import tf_checkpoint
    arrays = dict(self.model.state_dict())
    arrays.update(self.model.optimizer_slots())
    arrays["global_step"] = np.array(self.global_step, dtype=np.int64)
    tf_checkpoint.write_v2(prefix, arrays)
    return prefix

  def load_checkpoint(self, path):
    import numpy as np
    from cap2det_amd.train import tf_checkpoint
    if not path.endswith(".npz") and tf_checkpoint.checkpoint_exists(path):
      # a TensorFlow checkpoint of the reference (or of export_tf_checkpoint): variables and
      # global_step; Adagrad slots (`<var>/Adagrad`) when the file carries them
      arrays = tf_checkpoint.read_checkpoint(path)
      self.global_step = int(arrays.pop("global_step", 0))
      self.model.load_state_dict(arrays, strict="checkpoint")
      self.model.load_optimizer_slots(arrays)
      return
    arrays = dict(np.load(path if path.endswith(".npz") else path + ".npz"))
    self.global_step = int(arrays.pop("__global_step"))
    accum = arrays.pop("__adagrad_accumulators")
    extra = {k: arrays.pop(k) for k in list(arrays) if k.startswith("__optimizer_slot")}
    self.model.load_state_dict(arrays)
    self.model.store.accum.copy_(torch.from_numpy(accum).to(self.device))
    for k, a in extra.items():
      self.model.store.slots[int(k[len("__optimizer_slot"):])].copy_(torch.from_numpy(a).to(self.device))

  def learning_rate(self):
    tc = self.train_config
    lr = tc.learning_rate
    if tc.HasField('learning_rate_decay'):
      d = tc.learning_rate_decay
      lr = exponential_decay(lr, self.global_step, d.decay_steps, d.decay_rate, d.staircase)
    return lr

  def _forward_backward(self, examples, after_second_stage=None, prefetch=None, after_block=None,
                        prefetch_ready=None, **kwargs):
    model, store = self.model, self.model.store
    lo, hi = self.bucket
    # everything the step's accumulating kernels add into, cleared by ONE launch: the gradient
    # bucket, the loss scalars, the gradient map of the ROI-crop backward
    from cap2det_amd.core.standard_fields import InputDataFields as F
    ops.zero_ranges([store.grads[lo:hi], model._losses] +
                    model.engine.step_zero_list(examples[F.image].shape, examples[F.proposals].shape[1]))
    kwargs = dict(kwargs, step_zeroed=True)
    # The labels depend on the examples only (captions -> GloVe / text classifier or string
    # matching): they are extracted on the look-ahead stream under the detector's forward pass
    # instead of between its forward pass and its losses (0.2-0.3 ms per step for the text-
    # classifier extractors); build_loss waits for the event.
    stream = getattr(model.engine, "prefetch_stream", None)
    if (kwargs.get("labels") is None and stream is not None and
        getattr(model.label_extractor, "overlaps_forward", False)):
      main = torch.cuda.current_stream()
      fork = torch.cuda.Event(); fork.record()
      stream.wait_event(fork)
      with torch.cuda.stream(stream):
        labels = model.label_extractor.extract_labels(examples)
        ready = torch.cuda.Event(); ready.record()
      labels.record_stream(main)
      kwargs = dict(kwargs, labels=labels, labels_ready=ready)
    try:
      predictions = model.build_prediction(examples, **kwargs)
      if prefetch is not None:
        # look-ahead: the frozen first-stage layers of the NEXT batch's image run on a side stream
        # under this step's second stage (FrcnnEngine.prefetch_first_stage)
        if prefetch_ready is not None and getattr(model.engine, "prefetch_stream", None) is not None:
          # (the next batch came from an input thread's copy stream: only its READER waits for it)
          model.engine.prefetch_stream.wait_event(prefetch_ready)
        model.engine.prefetch_first_stage(prefetch[F.image], prefetch[F.proposals].shape[1], True)
      losses = dict(model.build_loss(predictions, examples=examples, **kwargs))
      losses['regularization_loss'] = model.regularization_loss(step_zeroed=True)
      model.backward(after_second_stage, after_block)
    finally:
      # "zeroed by the step's one launch" holds for THIS backward pass only: if the forward pass or
      # the losses raise, a later direct model.backward() must zero its gradient map itself
      model.engine._step_zeroed = set()
    return predictions, losses

  def _apply_gradients(self, scale, lr, lr_dev=None):
    """Adagrad over the trainable segments of the flat buffers.  The common case (one multiplier
    per segment, L2 only, no clipping) is one c2d_adagrad_step per segment; l1 regularisers,
    per-head multipliers, `max_gradient_norm` and a device-resident learning rate go through
    c2d_clip_gradient_norms / c2d_adagrad_step_ex."""
    store = self.model.store
    v, g, a = store.values, store.grads, store.accum
    clipped = self._clip is not None
    if clipped:
      desc, num, max_norm = self._clip
      ops.clip_gradient_norms(g, v, desc, num, scale, max_norm)    # g <- final clipped gradient
    if self.opt_kind != 'adagrad':
      self._apply_other_optimizer(scale, lr, lr_dev, clipped)
      self.model.refresh(only_trainable=True)
      return
    if (not clipped and lr_dev is None and self.segments and len(self.segments) <= 8 and
        all(cols is None and l1 == 0.0 for _, _, _, l1, _, cols, _ in self.segments)):
      # the common case in ONE launch; a bf16 network's mirror of the variables is written by the
      # same pass, the mirror of the derived operands by the transposes of refresh()
      mirror = self.model.engine.values_mirror()
      ops.adagrad_step_multi(v, g, a, [(off, end, m, l2) for off, end, m, _, l2, _, _ in self.segments],
                             lr, scale, mirror)
      self.model.engine.refresh(only_trainable=True, values_mirrored=mirror is not None)
      return
    for off, end, m, l1, l2, cols, mask in self.segments:
      if clipped:
        # the descriptors applied scale / regularisers / multipliers; what is left is the
        # frozen-column mask of a fused heads buffer
        if cols is None and lr_dev is None:
          ops.adagrad_step(v[off:end], g[off:end], a[off:end], lr, 0.0, 1.0, 1.0)
        else:
          ops.adagrad_step_ex(v[off:end], g[off:end], a[off:end], lr, 0.0, 0.0, 1.0, 1.0, mask,
                              0 if mask is None else mask.numel(), lr_dev)
      elif cols is None and l1 == 0.0 and lr_dev is None:
        ops.adagrad_step(v[off:end], g[off:end], a[off:end], lr, l2, m, scale)
      else:
        ops.adagrad_step_ex(v[off:end], g[off:end], a[off:end], lr, l1, l2, m, scale, cols,
                            0 if cols is None else cols.numel(), lr_dev)
    self.model.refresh(only_trainable=True)

  def _apply_other_optimizer(self, scale, lr, lr_dev, clipped):
    """sgd / momentum / adam / rmsprop over the trainable segments (c2d_optimizer_step)."""
    store, o, kind = self.model.store, self.opt_options, self.opt_kind
    v, g = store.values, store.grads
    flags, p = 0, (0.0, 0.0, 0.0, 0.0)
    if kind == 'momentum':
      p = (o.momentum, 0.0, 0.0, 0.0)
      flags = 1 if o.use_nesterov else 0
    elif kind == 'adam':
      t = self.global_step + 1
      p = (o.beta1, o.beta2, o.epsilon, lr * math.sqrt(1.0 - o.beta2 ** t) / (1.0 - o.beta1 ** t))
    elif kind == 'rmsprop':
      p = (o.decay, o.momentum, o.epsilon, 0.0)
      flags = 2 if o.centered else 0
    for off, end, m, l1, l2, cols, mask in self.segments:
      slots = [] if kind == 'sgd' else [sl[off:end] for sl in store.slots]
      if clipped:     # (the descriptors applied scale / regularisers / multipliers)
        ops.optimizer_step(kind, v[off:end], g[off:end], slots, lr, p, flags, 0.0, 0.0, 1.0, 1.0,
                           mask, 0 if mask is None else mask.numel(), lr_dev)
      else:
        ops.optimizer_step(kind, v[off:end], g[off:end], slots, lr, p, flags, l1, l2, m, scale,
                           cols, 0 if cols is None else cols.numel(), lr_dev)

  @staticmethod
  def _dp_buckets():
    """C2D_DP_BUCKETS=blocks|two; default: one exchange per second-stage block over RCCL (device-
    side collectives queued behind the filter gradients), the two-bucket form over gloo, whose
    collectives are staged through the host (the same-device rehearsal of bench.py --gpus N on a
    one-GPU box: 46 ms against 200-990 ms per step with four host-staged collectives)."""
    want = os.environ.get("C2D_DP_BUCKETS")
    if want in ("blocks", "two"):
      return want
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_backend() != "nccl":
      return "two"
    return "blocks"

  def train_step(self, examples, prefetch=None, prefetch_ready=None, **kwargs):
    """One synchronous step; returns {loss name: 0-d device tensor} (+ 'total_loss',
    'regularization_loss').  No host synchronisation happens inside.  `prefetch`: the NEXT
    step's examples, when the caller already has them (an input pipeline always does): their
    frozen first-stage layers are computed under this step's kernels.  `prefetch_ready`: the event
    behind the uploads of `prefetch` when they were queued on another stream (Trainer.train)."""
    if kwargs.get("dropout_seed") is None and kwargs.get("dropout_mask") is None:
      # slim.dropout draws a fresh mask every step on every worker (models/utils.py:171-174):
      # the counter-based generator is keyed on (trainer seed, global step, rank)
      kwargs["dropout_seed"] = dropout_key(self.seed, self.global_step, self.rank, self.world_size)
    if self.use_plan and self._plan_eligible(kwargs):
      return self._plan_step(examples, prefetch, prefetch_ready, kwargs)
    self._announced = self._next_labels = None
    self._leave_replay()
    kwargs["prefetch"] = prefetch
    kwargs["prefetch_ready"] = prefetch_ready
    store = self.model.store
    lo, hi = self.bucket
    if self._dp_buckets() == "blocks" and len(self._block_cuts) > 2:
      # one asynchronous all-reduce per second-stage block, launched as the backward pass leaves
      # the block (heads ride with the last block), the Mixed_4e prefix at the end
      reducer = data_parallel.BlockReducer(store.grads[lo:hi], self._block_cuts)
      ranges = self._block_range
      hook = (lambda i: reducer.start(ranges[i]) if i in ranges else None) if reducer.on else None
      predictions, losses = self._forward_backward(examples, None, after_block=hook, **kwargs)
    else:
      reducer = data_parallel.OverlappedReducer(store.grads[lo:hi], self._tail_split)
      predictions, losses = self._forward_backward(examples, reducer.start_tail, **kwargs)
    scale = reducer.finish()
    self._apply_gradients(scale, self.learning_rate())
    self.global_step += 1
    losses['total_loss'] = self.model.total_loss()
    self.predictions = predictions
    return losses

  # -- step plans ----------------------------------------------------------------------
  MAX_PLANS = 16      # recorded plans kept (one per input signature; least recently used dropped)

  def _plan_eligible(self, kwargs):
    return (self.world_size == 1 and not data_parallel.collectives_on() and
            self.opt_kind == 'adagrad' and self._clip is None and len(self.segments) <= 8 and
            all(cols is None and l1 == 0.0 for _, _, _, l1, _, cols, _ in self.segments) and
            kwargs.get("dropout_mask") is None and kwargs.get("labels") is None and
            not self.model.engine.dropout_on_feature_map)

  def _join_streams(self):
    """The current stream waits for everything queued on the engine's other streams."""
    eng = self.model.engine
    for s in (eng.prefetch_stream, eng.second.side, eng.second.alt, eng.first.alt):
      if s is not None:
        torch.cuda.current_stream().wait_stream(s)

  def _leave_replay(self):
    """In front of a Python-driven step that follows a replayed one: the replay's other streams are
    joined (its events are the plan's own, unknown to the engine's bookkeeping) and the engine's
    look-ahead record — the recorded step's, stale by now — is dropped."""
    if self._last_replayed:
      self._join_streams()
      self.model.engine.invalidate_prefetch()
      self._last_replayed = False

  @staticmethod
  def _signature(examples, prefetch):
    def sig(d):
      if d is None:
        return None
      return tuple(sorted((k, tuple(v.shape), str(v.dtype)) for k, v in d.items()
                          if isinstance(v, torch.Tensor) and v.is_cuda))
    return sig(examples), sig(prefetch)

  def _eager_core(self, examples, labels, seed, lr, prefetch, prefetch_ready):
    """The step as train_step queues it, labels given: what a plan records."""
    reducer = data_parallel.OverlappedReducer(self.model.store.grads[self.bucket[0]:self.bucket[1]],
                                              self._tail_split)
    predictions, losses = self._forward_backward(examples, None, labels=labels, dropout_seed=seed,
                                                 prefetch=prefetch, prefetch_ready=prefetch_ready)
    scale = reducer.finish()
    self._apply_gradients(scale, lr)
    losses['total_loss'] = self.model.total_loss()
    return predictions, losses

  def _plan_step(self, examples, prefetch, prefetch_ready, kwargs):
    """One step through a recorded plan when one exists for this input signature and the look-
    ahead of the previous step was for this batch; the third such eager step is the one recorded
    (by then every buffer of the shape exists and the look-ahead alternates in steady state)."""
    from cap2det_amd.core.standard_fields import InputDataFields as F
    from cap2det_amd.step_plan import StepPlan, Sym
    model, eng = self.model, self.model.engine
    if self._plan_generation != eng.generation:
      # the engine dropped its launch plans / buffers (set_trainable): recorded addresses are gone
      self._plans.clear()
      self._plan_generation = eng.generation
      self._leave_replay()
    image = examples[F.image]
    seed = kwargs.get("dropout_seed")
    lr = self.learning_rate()
    # labels: extracted under the previous step for the batch it announced, else now
    nl, self._next_labels = self._next_labels, None
    if nl is not None and nl[0] is image and nl[1] == image._version:
      labels = nl[2]
      torch.cuda.current_stream().wait_event(nl[3])
    else:
      labels = model.label_extractor.extract_labels(examples)
    steady = (self._announced is not None and self._announced[0] is image and
              self._announced[1] == image._version)
    key = self._signature(examples, prefetch)
    st = self._plans.setdefault(key, dict(plan=None, eager=0, failed=False))
    b, h, w, _ = image.shape
    n = examples[F.proposals].shape[1]
    look = prefetch is not None and getattr(eng, "prefetch_stream", None) is not None
    if st["plan"] is not None and steady and look:
      plan = st["plan"]
      tensors = {"ex." + k: v for k, v in examples.items() if isinstance(v, torch.Tensor) and v.is_cuda}
      tensors.update({"next." + k: v for k, v in prefetch.items()
                      if isinstance(v, torch.Tensor) and v.is_cuda})
      tensors["labels"] = labels
      if prefetch_ready is not None:
        eng.prefetch_stream.wait_event(prefetch_ready)
      if not self._last_replayed:
        self._join_streams()     # a plan starts from joined streams (c2d_plan_finish)
      t_r = time.perf_counter()
      plan.replay(tensors, {"seed": int(seed) & 0xFFFFFFFFFFFFFFFF, "lr": lr})
      self.replay_s += time.perf_counter() - t_r     # host time inside c2d_plan_replay (tools/host_time.py)
      self.plan_replays += 1
      self._last_replayed = True
      st["used"] = self.global_step
      predictions, losses = st["result"]
    else:
      record = (st["plan"] is None and not st["failed"] and steady and look and st["eager"] >= 2)
      self._leave_replay()
      if record:
        bufs = eng._buffers(b, h, w, n, True)
        upto = eng._prefix_len(bufs)
        record = upto > 0 and bufs.get("prefetched") is not None and "prefix_alt" in bufs
      if record:
        plan = StepPlan()
        for k, v in examples.items():
          if isinstance(v, torch.Tensor) and v.is_cuda:
            plan.bind_tensor("ex." + k, v)
        for k, v in prefetch.items():
          if isinstance(v, torch.Tensor) and v.is_cuda:
            plan.bind_tensor("next." + k, v)
        plan.bind_tensor("labels", labels)
        try:
          with plan.recording():
            predictions, losses = self._eager_core(examples, labels, Sym("seed", seed), Sym("lr", lr),
                                                   prefetch, prefetch_ready)
        except Exception:
          st["failed"] = True
          raise
        st.update(plan=plan, result=(predictions, losses), used=self.global_step)
        # (a reader with many input shapes: keep the plans of the most recently used signatures)
        live = [(v["used"], k) for k, v in self._plans.items() if v["plan"] is not None]
        for _, k in sorted(live)[:-self.MAX_PLANS]:
          self._plans[k].update(plan=None, result=None, eager=0)
      else:
        predictions, losses = self._eager_core(examples, labels, seed, lr, prefetch, prefetch_ready)
        st["eager"] += 1
    self.global_step += 1
    self.predictions = predictions
    # what the next step may rely on: the look-ahead ran for `prefetch`, whose labels are extracted
    # now, under this step's kernels
    self._announced = None
    if look:
      nxt = prefetch[F.image]
      self._announced = (nxt, nxt._version)
      stream = eng.prefetch_stream
      if getattr(model.label_extractor, "overlaps_forward", False):
        with torch.cuda.stream(stream):
          nlab = model.label_extractor.extract_labels(prefetch)
          ready = torch.cuda.Event()
          ready.record()
        nlab.record_stream(torch.cuda.current_stream())
        self._next_labels = (nxt, nxt._version, nlab, ready)
    return losses

  # -- the training loop -----------------------------------------------------------------
  def train(self, batches, max_steps=None, save_dir=None, save_every=None, log=None,
            prefetch_depth=2):
    """train/trainer.py:210-235 (`tf.estimator.train_and_evaluate`'s training half): consumes an
    iterable of example dicts (e.g. `cap2det_reader.get_input_fn(...)()`) until it ends or
    `train_config.max_steps` / `max_steps` is reached, always holding ONE batch of look-ahead
    for `train_step(prefetch=...)`; optional periodic checkpoints.  Returns the last losses.
    `prefetch_depth`: batches the input thread keeps ready (the reference's
    `prefetch_buffer_size`, readers/cap2det_reader.py:266, counts batches too)."""
    limit = max_steps if max_steps is not None else (self.train_config.max_steps or None)
    from cap2det_amd.readers.prefetch import DevicePrefetcher, adopt
    on_gpu = torch.device(self.device).type == "cuda"
    # The input function runs in a thread of its own under a copy stream (readers/prefetch.py): the
    # uploads and reader kernels of batch k+1 are queued while step k-1 is still on the GPU and
    # NEVER wait for the compute stream.  What waits is the reader of a batch: the compute stream
    # for the batch it is about to step on (long finished by then), the look-ahead stream for the
    # next one (`prefetch_ready`).  Round 4 made the copy stream wait for the compute stream and
    # the compute stream for the copy stream around every pull: the upload then sat between two
    # steps (tests/test_gpu_reader.py::test_train_overlaps_the_uploads).
    src = DevicePrefetcher(batches, self.device, depth=prefetch_depth)
    eng = self.model.engine
    import time
    self.input_wait_s = 0.0        # host time this thread spent waiting for the input thread
    self.enqueue_s = 0.0           # host time inside train_step (queueing the step's launches)

    def pull():
      t0 = time.perf_counter()
      batch = next(src, None)
      self.input_wait_s += time.perf_counter() - t0
      if batch is not None and on_gpu:
        adopt(batch, torch.cuda.current_stream(), getattr(eng, "prefetch_stream", None))
      return batch

    losses = None
    try:
      cur = pull()
      while cur is not None and (limit is None or self.global_step < limit):
        nxt = pull()
        last = limit is not None and self.global_step + 1 >= limit
        if on_gpu and cur.get("_ready") is not None:
          torch.cuda.current_stream().wait_event(cur["_ready"])
        ahead = None if (last or nxt is None) else nxt
        t0 = time.perf_counter()
        losses = self.train_step(cur, prefetch=ahead,
                                 prefetch_ready=None if ahead is None else ahead.get("_ready"))
        self.enqueue_s += time.perf_counter() - t0
        if log is not None:
          log(self.global_step, losses)
        if save_dir and save_every and self.global_step % save_every == 0:
          self.save_checkpoint(save_dir)
        cur = nxt
    finally:
      src.close()
    return losses
