"""Host-side execution engine of the Fast-RCNN feature path on MI355X.

Builds, for fixed tensor shapes, a static launch plan over the HIP kernels of
`include/cap2det_hip.h` that computes what the reference computes in
`models/utils.py:108-188` (`extract_frcnn_feature`: Inception-V2 first stage -> crop_and_resize
-> max-pool -> Inception-V2 second stage -> spatial mean -> dropout) and its gradient.

The Inception-V2 layer tables restate tf.contrib.slim `nets/inception_v2.py` and the
`FasterRCNNInceptionV2FeatureExtractor` of the object_detection fork the reference imports
(models/utils.py:9) — third-party code that is not vendored in the reference.

Memory layout (HBM, fp32): activations NHWC as [rows = n*h*w][channels]; every Inception block
owns ONE concat buffer and its branches write channel slices of it in place; variables live
in one flat parameter buffer with a mirrored flat gradient buffer (the RCCL all-reduce bucket)
and a mirrored Adagrad accumulator.
"""
import math
import os

import torch

from cap2det_amd import hip_ops as ops
from cap2det_amd import tune

BN_EPS = 0.001
FIRST_SCOPE = "first_stage_feature_extraction/InceptionV2/"
SECOND_SCOPE = "second_stage_feature_extraction/InceptionV2/"


# ----------------------------------------------------------------------------------------
# layer tables
# ----------------------------------------------------------------------------------------

def _mixed(name, b0, b1, b2, b3, pool="avg"):
  pool_name = "AvgPool_0a_3x3" if pool == "avg" else "MaxPool_0a_3x3"
  return ("block", name, [
      [("conv", "Branch_0/Conv2d_0a_1x1", b0, 1, 1)],
      [("conv", "Branch_1/Conv2d_0a_1x1", b1[0], 1, 1),
       ("conv", "Branch_1/Conv2d_0b_3x3", b1[1], 3, 1)],
      [("conv", "Branch_2/Conv2d_0a_1x1", b2[0], 1, 1),
       ("conv", "Branch_2/Conv2d_0b_3x3", b2[1], 3, 1),
       ("conv", "Branch_2/Conv2d_0c_3x3", b2[2], 3, 1)],
      [(pool, "Branch_3/" + pool_name, 1), ("conv", "Branch_3/Conv2d_0b_1x1", b3, 1, 1)],
  ])


def _reduction(name, b0, b1):
  return ("block", name, [
      [("conv", "Branch_0/Conv2d_0a_1x1", b0[0], 1, 1),
       ("conv", "Branch_0/Conv2d_1a_3x3", b0[1], 3, 2)],
      [("conv", "Branch_1/Conv2d_0a_1x1", b1[0], 1, 1),
       ("conv", "Branch_1/Conv2d_0b_3x3", b1[1], 3, 1),
       ("conv", "Branch_1/Conv2d_1a_3x3", b1[2], 3, 2)],
      [("max", "Branch_2/MaxPool_1a_3x3", 2)],
  ])


# The 7x7/2 separable stem ("Conv2d_1a_7x7") is handled apart (see FrcnnEngine._stem_*).
FIRST_STAGE_AFTER_STEM = [
    ("max", "MaxPool_2a_3x3", 2),
    ("conv", "Conv2d_2b_1x1", 64, 1, 1),
    ("conv", "Conv2d_2c_3x3", 192, 3, 1),
    ("max", "MaxPool_3a_3x3", 2),
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
    _mixed("Mixed_5c", 352, (192, 320), (192, 224, 224), 128, pool="max"),
]


GROUP_MAX_ROWS = 16384   # block inputs up to this many rows use grouped launches per level
# block inputs from this many rows on take ONE grouped filter-gradient launch for their 1x1 entry
# convolutions (c2d_conv1x1_wgrad_multi); below, the row splits of one output already fill the chip
WGRAD_MULTI_MIN_ROWS = 8192
DC_SLOTS = 2   # dC scratch buffers per stream (see _prepare_backward)


def _out_hw(h, w, stride):
  return -(-h // stride), -(-w // stride)


# ----------------------------------------------------------------------------------------
# variables
# ----------------------------------------------------------------------------------------

class VariableStore(object):
  """Flat fp32 buffers (values / gradients / Adagrad accumulators) with named views.

  Variables are laid out in creation order (network order), so the trainable suffix
  [Mixed_4e?, second stage, heads] is one contiguous range = one all-reduce bucket."""

  ALIGN = 4  # floats (16 B)

  def __init__(self, device):
    self.device = device
    self._specs = []     # (name, shape, offset, numel)
    self._size = 0
    self.values = None
    self.grads = None
    self.accum = None
    self.var = {}
    self.grad = {}
    self.acc = {}
    self.offset = {}

  def declare(self, name, shape):
    if self.values is not None:
      raise RuntimeError("VariableStore already finalized")
    numel = int(math.prod(shape))
    self._specs.append((name, tuple(shape), self._size, numel))
    self._size += -(-numel // self.ALIGN) * self.ALIGN

  def finalize(self, initial_accumulator_value=0.1):
    self.values = torch.zeros(self._size, device=self.device, dtype=torch.float32)
    self.grads = torch.zeros(self._size, device=self.device, dtype=torch.float32)
    self.accum = torch.full((self._size,), float(initial_accumulator_value), device=self.device,
                            dtype=torch.float32)
    for name, shape, off, numel in self._specs:
      self.var[name] = self.values[off:off + numel].view(shape)
      self.grad[name] = self.grads[off:off + numel].view(shape)
      self.acc[name] = self.accum[off:off + numel].view(shape)
      self.offset[name] = (off, numel)

  def names(self):
    return [s[0] for s in self._specs]

  def low_precision_view(self, name):
    """bf16 mirror of a variable (refreshed by FrcnnEngine.refresh for bf16 networks)."""
    if getattr(self, "values_bf16", None) is None:
      self.values_bf16 = torch.zeros(self.values.numel(), device=self.device, dtype=torch.bfloat16)
    off, numel = self.offset[name]
    return self.values_bf16[off:off + numel].view(self.var[name].shape)

  def span(self, names):
    """Smallest [lo, hi) flat range covering `names` (aligned ends)."""
    lo = min(self.offset[n][0] for n in names)
    hi = max(self.offset[n][0] + -(-self.offset[n][1] // self.ALIGN) * self.ALIGN for n in names)
    return lo, hi


class DerivedStore(object):
  """Flat buffers for (a) BatchNorm moving statistics and (b) operands derived from the
  variables; dict-like over the statistics so existing `stats[name]` call sites keep working."""

  def __init__(self, device):
    self.device = device
    self._stat_specs, self._der_specs = [], []
    self._stat_size = self._der_size = 0
    self.stat_flat = self.der_flat = None
    self.der_bf16 = None
    self._stats, self.derived = {}, {}
    self.stat_off, self.der_off = {}, {}

  def declare(self, name, shape, fill):
    n = int(math.prod(shape))
    self._stat_specs.append((name, tuple(shape), self._stat_size, n, fill))
    self._stat_size += -(-n // 4) * 4

  def declare_derived(self, name, shape):
    n = int(math.prod(shape))
    self._der_specs.append((name, tuple(shape), self._der_size, n))
    self._der_size += -(-n // 4) * 4

  def finalize(self):
    # (initial statistics are staged on the host and uploaded ONCE: no per-layer fill launches)
    host = torch.zeros(max(self._stat_size, 4))
    for name, shape, off, n, fill in self._stat_specs:
      if fill != 0.0:
        host[off:off + n] = fill
    self.stat_flat = host.to(self.device)
    self.der_flat = torch.zeros(max(self._der_size, 4), device=self.device)
    for name, shape, off, n, fill in self._stat_specs:
      self._stats[name] = self.stat_flat[off:off + n].view(shape)
      self.stat_off[name] = off
    for name, shape, off, n in self._der_specs:
      self.derived[name] = self.der_flat[off:off + n].view(shape)
      self.der_off[name] = off

  def low_precision_view(self, name):
    if self.der_bf16 is None:
      self.der_bf16 = torch.zeros(self.der_flat.numel(), device=self.device, dtype=torch.bfloat16)
    off = self.der_off[name]
    v = self.derived[name]
    return self.der_bf16[off:off + v.numel()].view(v.shape)

  # dict protocol over the moving statistics
  def __getitem__(self, k):
    return self._stats[k]

  def __contains__(self, k):
    return k in self._stats

  def __iter__(self):
    return iter(self._stats)

  def keys(self):
    return self._stats.keys()


class ConvBN(object):
  """conv (SAME, no bias) + inference BatchNorm + ReLU; owns its variables and folded forms."""

  def __init__(self, store, stats, name, cin, cout, k, stride, bn_scale):
    self.name, self.cin, self.cout, self.k, self.stride = name, cin, cout, k, stride
    self.store, self.stats = store, stats
    self.bn_scale = bn_scale
    store.declare(name + "/weights", (k, k, cin, cout))
    if bn_scale:
      store.declare(name + "/BatchNorm/gamma", (cout,))
    store.declare(name + "/BatchNorm/beta", (cout,))
    # moving statistics and the derived kernel operands (per-tap transposed weights, folded
    # BN affine) are views into flat buffers owned by the engine (see DerivedStore), so that
    # ONE batched launch can refresh all layers.
    stats.declare(name + "/BatchNorm/moving_mean", (cout,), 0.0)
    stats.declare(name + "/BatchNorm/moving_variance", (cout,), 1.0)
    stats.declare_derived(name + "/wt", (k * k, cout, cin))
    stats.declare_derived(name + "/scale", (cout,))
    stats.declare_derived(name + "/shift", (cout,))
    self.trainable = False

  @property
  def wt(self):
    return self.stats.derived[self.name + "/wt"]

  def wt_for(self, dtype):
    """Forward operand [tap][cout][cin] in the network's storage type."""
    return self.wt if dtype == torch.float32 else self.stats.low_precision_view(self.name + "/wt")

  def w_for(self, dtype):
    """HWIO weights (the dgrad operand) in the network's storage type."""
    v = self.store.var[self.name + "/weights"]
    return v if dtype == torch.float32 else self.store.low_precision_view(self.name + "/weights")

  @property
  def scale(self):
    return self.stats.derived[self.name + "/scale"]

  @property
  def shift(self):
    return self.stats.derived[self.name + "/shift"]

  def var_names(self):
    n = [self.name + "/weights", self.name + "/BatchNorm/beta"]
    if self.bn_scale:
      n.insert(1, self.name + "/BatchNorm/gamma")
    return n

  def refresh(self):
    """Re-derives the kernel operands (per-tap transposed weights, folded BN) from the
    variables; called after every optimiser step for trainable layers."""
    v = self.store.var
    ops.transpose_taps(v[self.name + "/weights"], self.wt, self.k * self.k, self.cin, self.cout)
    ops.bn_fold(v.get(self.name + "/BatchNorm/gamma"), v[self.name + "/BatchNorm/beta"],
                self.stats[self.name + "/BatchNorm/moving_mean"],
                self.stats[self.name + "/BatchNorm/moving_variance"], BN_EPS, self.scale,
                self.shift)


class Ref(object):
  """A channel slice [off, off+c) of a [rows][ld] activation buffer."""
  __slots__ = ("t", "ld", "off", "c")

  def __init__(self, t, ld, off, c):
    self.t, self.ld, self.off, self.c = t, ld, off, c


# ----------------------------------------------------------------------------------------
# network executor
# ----------------------------------------------------------------------------------------

class Net(object):
  """A stack of conv / pool / Inception-block ops with a static plan per input shape."""

  def __init__(self, store, stats, spec, scope, cin, bn_scale, depth_multiplier=1.0,
               dtype=torch.float32):
    """dtype: storage type of the activations and their gradients (fp32, or bf16 with fp32
    accumulation: BASELINE.json configs[2] / [4])."""
    self.store, self.stats, self.spec, self.scope = store, stats, spec, scope
    self.dtype = dtype
    self.dm = depth_multiplier
    self.layers = {}      # name -> ConvBN
    self.order = []       # top-level op index -> [conv names]
    c = cin
    for op in spec:
      names = []
      if op[0] == "conv":
        c = self._add_conv(op, c, scope, bn_scale, names)
      elif op[0] == "block":
        total = 0
        for branch in op[2]:
          bc = c
          for bop in branch:
            if bop[0] == "conv":
              bc = self._add_conv(bop, bc, scope + op[1] + "/", bn_scale, names)
          total += bc
        c = total
      self.order.append(names)
    self.cout = c
    self.cin = cin
    self._plans = {}

  side = None   # torch.cuda.Stream for the filter gradients, set by FrcnnEngine (eager mode only)
  alt = None    # torch.cuda.Stream for the short branches of an Inception block (C2D_TUNE=branch_streams=0: off)
  alt_min_n = 64    # ... of blocks over at least this many maps (the single-image first stage: 1 —
                    # its backward pass is a chain of 10-30 us launches, three branches side by side)

  def _split_branches(self, st, skip_first):
    """(branch on the current stream, [branches for the branch stream]): the branch with the most
    ops left keeps the current stream.  skip_first: set of branch indices whose first op is done
    elsewhere (fused entry GEMM) / handled by the caller (backward: all of them)."""
    todo = []
    for bi, bsteps in enumerate(st["branches"]):
      rest = bsteps[1:] if (skip_first is None or bi in skip_first) else bsteps
      if rest:
        todo.append((bi, rest))
    if len(todo) < 2:
      return todo, []
    main = max(todo, key=lambda t: len(t[1]))
    return [main], [t for t in todo if t is not main]
  group_small = False   # bf16 nets: few-row block inputs take the grouped launches (first stage)

  def _depth(self, d):
    return max(int(d * self.dm), 16)

  def _add_conv(self, op, cin, prefix, bn_scale, names):
    cout = self._depth(op[2])
    name = prefix + op[1]
    self.layers[name] = ConvBN(self.store, self.stats, name, cin, cout, op[3], op[4], bn_scale)
    names.append(name)
    return cout

  def refresh(self, only_trainable=False):
    for layer in self.layers.values():
      if layer.trainable or not only_trainable:
        layer.refresh()

  # -- plan -----------------------------------------------------------------------
  def plan(self, n, ih, iw, training):
    key = (n, ih, iw, training)
    if key not in self._plans:
      self._plans[key] = self._build_plan(n, ih, iw, training)
    return self._plans[key]

  def _commutes(self, branch, block, n, training):
    """An average-pooling branch whose pool can run behind its 1x1 convolution (per-ROI maps).
    In a training plan only for a TRAINABLE convolution: the backward pass of a commuted layer
    is the no-ReLU form (c2d_bn_bwd_partial), which exists for trainable layers only — a frozen
    layer inside the backward range keeps the reference order (pool in front)."""
    if not (n >= 64 and len(branch) == 2 and branch[0][0] == "avg" and branch[0][2] == 1 and
            branch[1][0] == "conv" and tune.on("commute_avgpool")):
      return False
    layer = self.layers[self.scope + block + "/" + branch[1][1]]
    return layer.k == 1 and layer.stride == 1 and (layer.trainable or not training)

  def _new(self, rows, c, dtype=None):
    return torch.empty(rows, c, device=self.store.device, dtype=dtype or self.dtype)

  def _build_plan(self, n, ih, iw, training):
    steps = []
    h, w, c = ih, iw, self.cin
    x = None  # Ref of the running activation; None = the net input supplied at run time
    scratch_rows_c = 0
    for op in self.spec:
      kind = op[0]
      if kind == "conv":
        layer = self.layers[self.scope + op[1]]
        oh, ow = _out_hw(h, w, layer.stride)
        y = Ref(self._new(n * oh * ow, layer.cout), layer.cout, 0, layer.cout)
        st = dict(kind="conv", layer=layer, x=x, y=y, n=n, ih=h, iw=w, oh=oh, ow=ow)
        scratch_rows_c = max(scratch_rows_c, n * oh * ow * layer.cout)
        steps.append(st)
        x, h, w, c = y, oh, ow, layer.cout
      elif kind in ("max", "avg"):
        stride = op[2]
        oh, ow = _out_hw(h, w, stride)
        y = Ref(self._new(n * oh * ow, c), c, 0, c)
        arg = self._new(n * oh * ow, c, torch.uint8) if (kind == "max" and training) else None
        steps.append(dict(kind="pool", mode=0 if kind == "max" else 1, stride=stride, x=x, y=y,
                          arg=arg, n=n, ih=h, iw=w, oh=oh, ow=ow, c=c))
        x, h, w = y, oh, ow
      elif kind == "block":
        widths, strides = [], []
        for branch in op[2]:
          bc, bs = c, 1
          for bop in branch:
            if bop[0] == "conv":
              bc = self.layers[self.scope + op[1] + "/" + bop[1]].cout
              bs *= bop[4]
            else:
              bs *= bop[2]
          widths.append(bc)
          strides.append(bs)
        oh, ow = _out_hw(h, w, strides[0])
        ctot = sum(widths)
        ybuf = self._new(n * oh * ow, ctot)
        branches, off = [], 0
        for branch, wdt in zip(op[2], widths):
          bsteps = []
          bx, bh, bw, bc = x, h, w, c
          if self._commutes(branch, op[1], n, training):
            # avg_pool 3x3 -> 1x1 conv -> BN -> ReLU over per-ROI maps, run as 1x1 conv -> BN ->
            # avg_pool -> ReLU (c2d_avgpool3x3_relu_fwd: same function up to rounding): the pool
            # and its gradient touch cout channels instead of cin, and the convolution becomes
            # one more 1x1 entry convolution of the block (its input gradient joins the
            # multi-segment GEMM instead of a pooled pass over the whole block input).
            layer = self.layers[self.scope + op[1] + "/" + branch[1][1]]
            z = Ref(self._new(n * h * w, layer.cout), layer.cout, 0, layer.cout)
            bsteps.append(dict(kind="conv", layer=layer, x=bx, y=z, n=n, ih=h, iw=w, oh=h, ow=w,
                               relu=False, commuted=True))
            bsteps.append(dict(kind="pool", mode=1, stride=1, x=z, y=Ref(ybuf, ctot, off, layer.cout),
                               arg=None, n=n, ih=h, iw=w, oh=h, ow=w, c=layer.cout, relu=True))
            scratch_rows_c = max(scratch_rows_c, n * h * w * layer.cout)
            branches.append(bsteps)
            off += wdt
            continue
          for i, bop in enumerate(branch):
            last = i == len(branch) - 1
            if bop[0] == "conv":
              layer = self.layers[self.scope + op[1] + "/" + bop[1]]
              boh, bow = _out_hw(bh, bw, layer.stride)
              y = (Ref(ybuf, ctot, off, layer.cout) if last else
                   Ref(self._new(n * boh * bow, layer.cout), layer.cout, 0, layer.cout))
              bsteps.append(dict(kind="conv", layer=layer, x=bx, y=y, n=n, ih=bh, iw=bw, oh=boh,
                                 ow=bow))
              scratch_rows_c = max(scratch_rows_c, n * boh * bow * layer.cout)
              bx, bh, bw, bc = y, boh, bow, layer.cout
            else:
              stride = bop[2]
              boh, bow = _out_hw(bh, bw, stride)
              y = (Ref(ybuf, ctot, off, bc) if last else
                   Ref(self._new(n * boh * bow, bc), bc, 0, bc))
              arg = (self._new(n * boh * bow, bc, torch.uint8)
                     if (bop[0] == "max" and training) else None)
              bsteps.append(dict(kind="pool", mode=0 if bop[0] == "max" else 1, stride=stride,
                                 x=bx, y=y, arg=arg, n=n, ih=bh, iw=bw, oh=boh, ow=bow, c=bc))
              bx, bh, bw = y, boh, bow
          branches.append(bsteps)
          off += wdt
        y = Ref(ybuf, ctot, 0, ctot)
        # dependency levels: the i-th op of every branch only needs the (i-1)-th op of its own
        # branch, so the ops of one level are independent of each other
        depth = max(len(b) for b in branches)
        levels = [[b[i] for b in branches if len(b) > i] for i in range(depth)]
        steps.append(dict(kind="block", name=op[1], branches=branches, levels=levels, groups={},
                          x=x, y=y, n=n, ih=h, iw=w, oh=oh, ow=ow, cin=c))
        x, h, w, c = y, oh, ow, ctot
    plan = dict(steps=steps, out=x, oh=h, ow=w, n=n, ih=ih, iw=iw, bwd_ready=False,
                scratch_elems=scratch_rows_c)
    return plan

  # -- forward ----------------------------------------------------------------------
  def forward(self, plan, x_in, first=0, last=None):
    """Runs steps[first:last] (default: all)."""
    for st in plan["steps"][first:last]:
      self._fwd_step(st, x_in)
    return plan["out"]

  def _fwd_step(self, st, x_in):
    kind = st["kind"]
    x = st["x"] if st["x"] is not None else x_in
    if kind == "conv":
      L = st["layer"]
      ops.conv_fwd(x.t, x.ld, x.off, L.wt_for(self.dtype), L.scale, L.shift, st["y"].t, st["y"].ld, st["y"].off,
                   st["n"], st["ih"], st["iw"], L.cin, L.cout, L.k, L.k, L.stride, st.get("relu", True))
    elif kind == "pool" and st.get("relu"):
      ops.avgpool3x3_relu_fwd(x.t, x.ld, x.off, st["y"].t, st["y"].ld, st["y"].off, st["n"],
                              st["ih"], st["iw"], st["c"], st["stride"])
    elif kind == "pool":
      ops.pool3x3_fwd(x.t, x.ld, x.off, st["y"].t, st["y"].ld, st["y"].off, st["arg"], st["n"],
                      st["ih"], st["iw"], st["c"], st["stride"], st["mode"])
    elif st["n"] * st["ih"] * st["iw"] > GROUP_MAX_ROWS or (self.dtype != torch.float32 and
                                                             not self.group_small):
      # The 1x1 / stride-1 convolutions that open the branches all read the block input: ONE GEMM
      # over the sum of their output channels (c2d_conv1x1_fwd_multi: the input streams from HBM
      # once, wider tiles; every output element is the same K-ordered sum, bitwise equal results).
      fused = ()
      entry = [(bi, b[0]) for bi, b in enumerate(st["branches"])
               if b[0]["kind"] == "conv" and b[0]["layer"].k == 1 and b[0]["layer"].stride == 1]
      if 2 <= len(entry) <= 4:
        cache = st.setdefault("entry_fwd", {})
        key = (x.t.data_ptr(), x.ld, x.off)
        if key not in cache:
          keep = [(b["layer"].wt_for(self.dtype), b["layer"].scale, b["layer"].shift, b["y"].t,
                   b["y"].ld, b["y"].off, b["layer"].cout, b.get("relu", True)) for _, b in entry]
          cache[key] = (ops.conv_outs(keep), keep)
        ops.conv1x1_fwd_multi(x.t, x.ld, x.off, cache[key][0], st["n"] * st["ih"] * st["iw"], st["cin"])
        fused = tuple(bi for bi, _ in entry)
      if self.alt is not None and st["n"] >= 64:
        # the long branch on this stream, the others on the branch stream beside it: a 3x3
        # convolution over 2000 4x4 maps is ONE round of 250 workgroups (one per CU) and leaves
        # every CU's second workgroup slot to the kernel of another branch
        mains, others = self._split_branches(st, set(fused))
        if others:
          fork = torch.cuda.Event()
          fork.record()
          self.alt.wait_event(fork)
          with torch.cuda.stream(self.alt):
            for _, rest in others:
              for bst in rest:
                self._fwd_step(bst, x)
            joined = torch.cuda.Event()
            joined.record()
        for _, rest in mains:
          for bst in rest:
            self._fwd_step(bst, x)
        if others:
          torch.cuda.current_stream().wait_event(joined)
      else:
        for bi, bsteps in enumerate(st["branches"]):
          for j, bst in enumerate(bsteps):
            if j == 0 and bi in fused:
              continue
            self._fwd_step(bst, x)
    else:
      # few rows (the single-image first stage): each convolution alone is a launch of 30-250
      # workgroups bound by its own K-loop latency; the convolutions of one level go out as ONE
      # grouped launch (c2d_conv_fwd_grouped), level by level
      for li, level in enumerate(st["levels"]):
        convs = [b for b in level if b["kind"] == "conv"]
        if len(convs) >= 2:
          key = (li, x.t.data_ptr(), x.ld, x.off, st["y"].t.data_ptr())
          group = st["groups"].get(key)
          if group is None:
            calls = []
            for b in convs:
              bx = b["x"] if b["x"] is not None else x
              L = b["layer"]
              calls.append((bx.t, bx.ld, bx.off, L.wt_for(self.dtype), L.scale, L.shift, b["y"].t,
                            b["y"].ld, b["y"].off, b["n"], b["ih"], b["iw"], L.cin, L.cout, L.k,
                            L.k, L.stride, b.get("relu", True)))
            group = st["groups"][key] = ops.conv_group(calls)
          ops.conv_fwd_grouped(group)
        for b in level:
          if b["kind"] != "conv" or len(convs) < 2:
            self._fwd_step(b, x)

  # -- backward ---------------------------------------------------------------------
  def _prepare_backward(self, plan, first_idx):
    """Allocates gradient buffers for steps[first_idx:] (dense, one per activation)."""
    if plan["bwd_ready"]:
      return
    dev = self.store.device
    # dC scratch.  Filter gradients on a side stream (see _conv_bwd): a second scratch so that the
    # next layer's BN/ReLU backward does not overwrite a dC the side stream is still reading.  A
    # second SET for the branches of an Inception block that run on the branch stream (self.alt).
    # DC_SLOTS of them per set, taken in turn: the stream that runs the input gradients waits for the
    # side stream only when it comes back to a slot, so it may run DC_SLOTS - 1 filter gradients
    # ahead and reach the ROI-crop backward while the side stream still works.
    def scratch_set():
      extra = DC_SLOTS - 1 if self.side is not None else 0
      return dict(dc=torch.empty(plan["scratch_elems"], device=dev, dtype=self.dtype),
                  dc_alt=([torch.empty(plan["scratch_elems"], device=dev, dtype=self.dtype)
                           for _ in range(extra)] if extra else None),
                  events=[None] * (extra + 1), slot=0)
    plan["scr"] = scratch_set()
    # (the branch stream's set is made on demand: a plan first prepared while Net.alt was switched
    # off — Trainer._graph_step's warm-up and capture — must not leave every LATER eager step of the
    # shape without its branch streams)
    plan["scratch_set"] = scratch_set
    plan["scr_b"] = scratch_set() if self.alt is not None else None
    plan["on_alt"] = False
    steps = plan["steps"]
    for i in range(first_idx, len(steps)):
      st = steps[i]
      y = st["y"]
      rows = st["n"] * st["oh"] * st["ow"]
      st["gy"] = Ref(torch.empty(rows, y.c, device=dev, dtype=self.dtype), y.c, 0, y.c)   # grad of the op output
      if st["kind"] == "block":
        for bsteps in st["branches"]:
          b0 = bsteps[0]
          if (b0["kind"] == "conv" and b0["layer"].k == 1 and b0["layer"].stride == 1 and
              (i > first_idx or plan.get("need_input_grad", True))):
            b0["dc_entry"] = torch.empty(b0["n"] * b0["oh"] * b0["ow"], b0["layer"].cout,
                                         device=dev, dtype=self.dtype)
          for j, bst in enumerate(bsteps):
            if j == len(bsteps) - 1:
              by = bst["y"]
              bst["gy"] = Ref(st["gy"].t, st["gy"].ld, by.off, by.c)
            else:
              brow = bst["n"] * bst["oh"] * bst["ow"]
              bst["gy"] = Ref(torch.empty(brow, bst["y"].c, device=dev, dtype=self.dtype),
                              bst["y"].c, 0, bst["y"].c)
    # BatchNorm beta/gamma gradients: every trainable conv's bn_relu_bwd stores per-row-block
    # partial sums into one workspace; one batched launch at the end of backward() adds them
    # into the flat gradient buffer (atomic-free, bitwise reproducible).
    import numpy as np
    convs, conv_step = [], {}
    for i in range(first_idx, len(steps)):
      st = steps[i]
      if st["kind"] == "conv":
        convs.append(st)
      elif st["kind"] == "block":
        convs.extend(b for bsteps in st["branches"] for b in bsteps if b["kind"] == "conv")
      for c in convs:
        conv_step.setdefault(id(c), i)
    # Consecutive convolutions of a branch over per-ROI maps: the consumer's input-gradient GEMM
    # applies the producer's BN/ReLU backward in its epilogue (c2d_conv_dgrad_bn_relu) — the
    # producer's bn_relu_bwd launch, the store of its dy and the re-read disappear.  Measured per
    # step: fp32 12.33 -> 12.13 ms.  bf16 networks kept the separate launches through round 4 (4.13
    # -> 4.18 ms, 2.95 -> 2.98: round 2's fused pass form inside the plain ring kernel cost more
    # than the launches it removed); round 5 rebuilt the ring kernel's fused epilogue on LDS tables
    # and 16-byte stores as an instance of its own (igemm_bf16.hip, FUSED): 2.925 -> 2.915 ms per
    # step and 26 -> 12 bn_relu_bwd launches, so both storage modes fuse now.
    # (C2D_TUNE=fuse_bn_bwd=0: the separate launches, for the A/B tests)
    if tune.on("fuse_bn_bwd"):
      for i in range(first_idx, len(steps)):
        if steps[i]["kind"] != "block":
          continue
        for bsteps in steps[i]["branches"]:
          for j in range(1, len(bsteps)):
            prod, cons = bsteps[j - 1], bsteps[j]
            if not (prod["kind"] == "conv" and cons["kind"] == "conv" and prod["n"] >= 64 and
                    prod["layer"].trainable):
              continue
            Lc = cons["layer"]
            nb = ops.conv_dgrad_bn_relu_blocks(self.dtype, cons["n"], cons["ih"], cons["iw"],
                                               Lc.cin, Lc.cout, Lc.k, Lc.k, Lc.stride)
            if nb > 0:
              cons["fuse_prev"] = prod
              prod["fused_blocks"] = nb
              if "dc_entry" in prod:        # (entry convolution of the fused multi-segment dgrad)
                prod["dc_entry"] = prod["gy"].t
      # The same at a block boundary: the summed input gradient of a block whose input is
      # the concat buffer of the block in front is written last by the multi-segment GEMM of its
      # 1x1 entry convolutions (c2d_conv1x1_dgrad_multi_bn_relu) — it applies the BN/ReLU backward
      # of the LAST op of every branch of the block in front, per column range.
      for i in range(first_idx + 1, len(steps)):
        cur, prev = steps[i], steps[i - 1]
        if not (cur["kind"] == "block" and prev["kind"] == "block" and cur["n"] >= 64):
          continue
        entry = [b[0] for b in cur["branches"] if "dc_entry" in b[0]]
        lasts = [b[-1] for b in prev["branches"]]
        if len(entry) < 2 or not all(p["kind"] == "pool" or p["layer"].trainable for p in lasts):
          continue
        rows = cur["n"] * cur["ih"] * cur["iw"]
        nb = ops.conv1x1_dgrad_multi_bn_relu_blocks([b["layer"].cout for b in entry], rows, cur["cin"],
                                                    self.dtype)
        if nb <= 0:
          continue
        owner = dict(nb=nb, ctot=cur["cin"], keep=[])
        prods, off = [], 0
        for p in lasts:
          width = p["y"].c
          if p["kind"] == "conv":
            Lp = p["layer"]
            sc, be = Lp.scale, self.store.var[Lp.name + "/BatchNorm/beta"]
            ga = self.store.var.get(Lp.name + "/BatchNorm/gamma")
            owner["keep"] += [sc, be, ga]
            prods.append((sc, be, ga, width))
            p["fused_blocks"] = nb
            p["fused_wide"] = (owner, off)
          else:
            prods.append((None, None, None, width))
          off += width
        assert off == cur["cin"]
        owner["prod_list"] = prods            # (the C2dBnProducer array is built at first use)
        cur["fuse_out"] = owner
    # The last block's last convolutions write the network output: when every branch ends in a
    # trainable convolution over per-ROI maps, their BN/ReLU backward can take the gradient of
    # the averaged features directly (plan["head_grad"], set by the caller of backward()).
    last = steps[-1]
    plan["head_ok"] = False
    if (tune.on("fuse_bn_bwd") and last["kind"] == "block" and last["n"] >= 64 and
        all(b[-1]["kind"] == "conv" and b[-1]["layer"].trainable for b in last["branches"])):
      plan["head_ok"] = True
      for b in last["branches"]:
        b[-1]["head_producer"] = True
    ddt = np.dtype([("ws", "<i8"), ("dbeta", "<i8"), ("dgamma", "<i8"), ("nblocks", "<i4"),
                    ("c", "<i4"), ("begin", "<i4"), ("wide", "<i4")])
    recs, ws_size, chunks = [], 0, 0
    rec_step = []           # top-level step of every record (per-step reductions: backward(after_step=))
    voff = self.store.offset
    for st in convs:
      L = st["layer"]
      if not L.trainable:
        continue
      rec_step.append(conv_step[id(st)])
      rows = st["n"] * st["oh"] * st["ow"]
      g = voff[L.name + "/BatchNorm/gamma"][0] if L.bn_scale else -1
      if "fused_wide" in st:
        # this layer's sums are a column range of the [block][2][concat width] rows written for
        # the whole block boundary (one region per boundary, shared by its producers)
        owner, coloff = st["fused_wide"]
        if "ws_off" not in owner:
          owner["ws_off"] = ws_size
          ws_size += owner["nb"] * 2 * owner["ctot"]
        st["bn_part"] = (owner["ws_off"], owner["nb"] * 2 * owner["ctot"])
        recs.append((owner["ws_off"] + coloff, voff[L.name + "/BatchNorm/beta"][0], g, owner["nb"],
                     L.cout, chunks, owner["ctot"]))
        chunks += -(-L.cout // 64)
        continue
      nb = st.get("fused_blocks") or ops.bn_relu_bwd_partial_blocks(rows, L.cout)
      st["bn_part"] = (ws_size, nb * 2 * L.cout)
      recs.append((ws_size, voff[L.name + "/BatchNorm/beta"][0], g, nb, L.cout, chunks, 0))
      ws_size += nb * 2 * L.cout
      chunks += -(-L.cout // 64)
    plan["bn_ws"] = torch.empty(max(ws_size, 4), device=dev)
    plan["bn_desc"] = (torch.from_numpy(np.array(recs, dtype=ddt).view(np.uint8).copy()).to(dev)
                       if recs else None)
    plan["bn_num"], plan["bn_chunks"] = len(recs), chunks
    # the same records per top-level step (chunk indices rebased), for a backward pass that hands
    # every step's gradients over as soon as the step is done (data_parallel.BlockReducer)
    plan["bn_desc_steps"] = {}
    for i in sorted(set(rec_step)):
      sub = [list(r) for r, si in zip(recs, rec_step) if si == i]
      base = sub[0][5]
      for r in sub:
        r[5] -= base
      nch = sub[-1][5] + -(-sub[-1][4] // 64)
      plan["bn_desc_steps"][i] = (
          torch.from_numpy(np.array([tuple(r) for r in sub], dtype=ddt).view(np.uint8).copy()).to(dev),
          len(sub), nch)
    plan["bwd_ready"] = True
    plan["first_idx"] = first_idx

  def out_grad(self, plan, first_idx=0):
    """Buffer into which the caller writes d(loss)/d(net output) before `backward`."""
    self._prepare_backward(plan, first_idx)
    return plan["steps"][-1]["gy"]

  def join(self, plan):
    """The calling stream waits for the filter gradients still running on the side stream."""
    if plan.get("side_pending"):
      torch.cuda.current_stream().wait_stream(self.side)      # join: gradients complete below here
      plan["side_pending"] = False
      # (the rotation starts over: every step takes the same scratch slots in the same order, so a
      #  recorded step plan — cap2det_amd/step_plan.py — holds for every later step)
      plan["scr"]["events"] = [None] * len(plan["scr"]["events"])
      plan["scr"]["slot"] = 0
      if plan["scr_b"] is not None:
        plan["scr_b"]["events"] = [None] * len(plan["scr_b"]["events"])
        plan["scr_b"]["slot"] = 0

  def backward(self, plan, x_in, first_idx=0, dx_in=None, after_step=None, join=True):
    """join=False: the caller calls join(plan) itself, later (FrcnnEngine.backward: behind the
    ROI-crop backward and the first stage's backward pass, which do not need these gradients).
    Backpropagates plan['steps'][-1]['gy'] down to steps[first_idx]; gradients of the
    variables are ACCUMULATED into the store's flat gradient buffer (zeroed once per step by
    the trainer).  dx_in: Ref receiving d(loss)/d(net input) (overwritten) or None.
    after_step(i): called when every kernel that writes the gradients of top-level step i's
    variables has been enqueued (its BatchNorm partial sums reduced right there instead of at the
    end); with a filter-gradient side stream the call happens INSIDE that stream, behind an event
    of the main stream, so that work queued by the callee (the data-parallel all-reduce of the
    step's range) is ordered behind both and the main stream does not wait."""
    self._prepare_backward(plan, first_idx)
    steps = plan["steps"]
    for i in range(len(steps) - 1, first_idx - 1, -1):
      st = steps[i]
      if i > first_idx:
        gx = steps[i - 1]["gy"]
      else:
        gx = dx_in
      x = st["x"] if st["x"] is not None else x_in
      self._bwd_step(plan, st, x, gx, False)
      if after_step is not None:
        part = plan["bn_desc_steps"].get(i)
        if part is not None:
          ops.bn_partials_reduce_batched(part[0], part[1], part[2], plan["bn_ws"], self.store.grads)
        if plan.get("side_pending"):
          done = torch.cuda.Event()
          done.record()
          self.side.wait_event(done)
          with torch.cuda.stream(self.side):
            after_step(i)
        else:
          after_step(i)
    if join:
      self.join(plan)
    if plan["bn_num"] and after_step is None:
      ops.bn_partials_reduce_batched(plan["bn_desc"], plan["bn_num"], plan["bn_chunks"],
                                     plan["bn_ws"], self.store.grads)

  def _wgrad(self, plan, st, x, dc, dcld, dcoff):
    L = st["layer"]
    ops.conv_wgrad(x.t, x.ld, x.off, dc, dcld, dcoff, self.store.grad[L.name + "/weights"],
                   st["n"], st["ih"], st["iw"], L.cin, L.cout, L.k, L.k, L.stride)

  def _conv_bwd(self, plan, st, x, gx, accumulate, dc=None, defer=None):
    """BN/ReLU backward -> dc, filter gradient, and (when gx is given) the input gradient.
    defer: list collecting (step, dc, row stride, column offset) of the filter gradients the caller
    launches itself (_entry_wgrads: the entry convolutions of a block in ONE grouped launch)."""
    L = st["layer"]
    gy, y = st["gy"], st["y"]
    rows = st["n"] * st["oh"] * st["ow"]
    scr = plan["scr_b"] if plan["on_alt"] else plan["scr"]
    side = self.side if (L.trainable and scr["dc_alt"] is not None) else None
    slot = None
    dcld, dcoff = L.cout, 0
    if "fused_blocks" in st:
      # the consumer's fused input-gradient launch left this layer's dc in its gradient buffer
      # (an inner layer's own dense buffer, or its columns of a concat gradient)
      dc, dcld, dcoff = gy.t, gy.ld, gy.off
    if dc is None:
      buf = scr["dc"]
      if side is not None:
        slot = scr["slot"]
        scr["slot"] = (slot + 1) % len(scr["events"])
        buf = scr["dc"] if slot == 0 else scr["dc_alt"][slot - 1]
        if scr["events"][slot] is not None:       # the side stream's last reader of this scratch
          torch.cuda.current_stream().wait_event(scr["events"][slot])
          scr["events"][slot] = None
      dc = buf[:rows * L.cout].view(rows, L.cout)
    g = self.store.grad
    gamma = self.store.var.get(L.name + "/BatchNorm/gamma")
    beta = self.store.var[L.name + "/BatchNorm/beta"]
    tr = L.trainable
    head = plan.get("head_grad") if st.get("head_producer") else None
    if "fused_blocks" in st:
      pass            # dc was written by the consumer's fused input-gradient launch
    elif head is not None and tr and "bn_part" in st:
      # a convolution that writes the network output: its dy is a function of the gradient of the
      # averaged features (c2d_bn_relu_bwd_partial_head), the gradient map is never stored
      off, size = st["bn_part"]
      ops.bn_relu_bwd_partial_head(head["dmean"], head["ld"], head["off"] + y.off, head["mask"],
                                   head["mask_ld"], y.off, head["spatial"], head["keep_prob"],
                                   y.t, y.ld, y.off, L.scale, beta, gamma, dc,
                                   plan["bn_ws"][off:off + size], rows, L.cout)
    elif tr and "bn_part" in st and st.get("commuted"):
      off, size = st["bn_part"]          # (the ReLU sits behind the pool: see _build_plan)
      ops.bn_bwd_partial(gy.t, gy.ld, gy.off, y.t, y.ld, y.off, L.scale, beta, gamma, dc,
                         plan["bn_ws"][off:off + size], rows, L.cout)
    elif st.get("commuted"):
      raise RuntimeError("commuted convolution %s without a trainable backward form" % L.name)
    elif tr and "bn_part" in st:
      off, size = st["bn_part"]
      ops.bn_relu_bwd_partial(gy.t, gy.ld, gy.off, y.t, y.ld, y.off, L.scale, beta, gamma, dc,
                              plan["bn_ws"][off:off + size], rows, L.cout)
    else:
      if self.dtype != torch.float32:
        raise NotImplementedError("bf16 networks are trained as a whole (frozen layers inside the "
                                  "backward range are not supported)")
      ops.bn_relu_bwd(gy.t, gy.ld, gy.off, y.t, y.ld, y.off, L.scale, beta, gamma, dc,
                      g[L.name + "/BatchNorm/beta"] if tr else None,
                      g[L.name + "/BatchNorm/gamma"] if (tr and gamma is not None) else None,
                      rows, L.cout)
    if tr and defer is not None and slot is None:
      defer.append((st, dc, dcld, dcoff))
    elif tr and side is not None:
      # dW only meets the rest of the step at the all-reduce / optimiser: it runs on a side stream
      # beside the input-gradient GEMM of the same layer, each filling the other's partial rounds
      ready = torch.cuda.Event()
      ready.record()
      side.wait_event(ready)
      with torch.cuda.stream(side):
        self._wgrad(plan, st, x, dc, dcld, dcoff)
        if slot is not None:
          scr["events"][slot] = torch.cuda.Event()
          scr["events"][slot].record()
      plan["side_pending"] = True
    elif tr:
      self._wgrad(plan, st, x, dc, dcld, dcoff)
    prod = st.get("fuse_prev")
    if prod is not None and gx is not None:
      # gx is the producer's (dense) gradient buffer: it receives the producer's dc directly
      assert not accumulate and gx.off == 0 and gx.ld == L.cin
      Lp, yp = prod["layer"], prod["y"]
      off, size = prod["bn_part"]
      ops.conv_dgrad_bn_relu(dc, dcld, dcoff, L.w_for(self.dtype), yp.t, yp.ld, yp.off, Lp.scale,
                             self.store.var[Lp.name + "/BatchNorm/beta"],
                             self.store.var.get(Lp.name + "/BatchNorm/gamma"), gx.t,
                             plan["bn_ws"][off:off + size], st["n"], st["ih"], st["iw"], L.cin,
                             L.cout, L.k, L.k, L.stride)
    elif gx is not None:
      ops.conv_dgrad(dc, dcld, dcoff, L.w_for(self.dtype), gx.t, gx.ld, gx.off,
                     st["n"], st["ih"], st["iw"], L.cin, L.cout, L.k, L.k, L.stride, accumulate)

  def _entry_wgrads(self, plan, x, deferred):
    """Filter gradients of a block's 1x1 entry convolutions, which all read the block input: ONE
    c2d_conv1x1_wgrad_multi launch (shared row splits, a third of the split atomics, the input
    rows fetched once) on the side stream."""
    if not deferred:
      return
    st0 = deferred[0][0]
    rows = st0["n"] * st0["oh"] * st0["ow"]
    side = self.side if plan["scr"]["dc_alt"] is not None else None

    def launch():
      if len(deferred) == 1 or rows < WGRAD_MULTI_MIN_ROWS or self.dtype == torch.float32:
        # (fp32: the grouped launch's 1024 workgroups crowd the input-gradient GEMM it runs
        # beside — 11.77 -> 11.91 ms per step measured — so the outputs stay separate launches)
        for st, dc, dcld, dcoff in deferred:
          self._wgrad(plan, st, x, dc, dcld, dcoff)
        return
      ops.conv1x1_wgrad_multi(
          x.t, x.ld, x.off, [dc for _, dc, _, _ in deferred], [ld for _, _, ld, _ in deferred],
          [off for _, _, _, off in deferred],
          [self.store.grad[st["layer"].name + "/weights"] for st, _, _, _ in deferred],
          [st["layer"].cout for st, _, _, _ in deferred], rows, st0["layer"].cin)

    if side is not None:
      ready = torch.cuda.Event()
      ready.record()
      side.wait_event(ready)
      with torch.cuda.stream(side):
        launch()
      plan["side_pending"] = True
    else:
      launch()

  @staticmethod
  def _entry_dc(b):
    """(tensor, row stride, column offset) of an entry convolution's dc for the multi-segment
    input-gradient GEMM: its own buffer, or — a one-convolution branch whose BN/ReLU backward the
    NEXT block's boundary launch applied — its columns of the concat gradient."""
    if "fused_wide" in b:
      return b["gy"].t, b["gy"].ld, b["gy"].off
    return b["dc_entry"], b["layer"].cout, 0

  def _bwd_step(self, plan, st, x, gx, accumulate):
    kind = st["kind"]
    if kind == "conv":
      self._conv_bwd(plan, st, x, gx, accumulate)
    elif kind == "pool":
      if gx is not None and st.get("relu"):
        gy, y = st["gy"], st["y"]
        ops.avgpool3x3_relu_bwd(gy.t, gy.ld, gy.off, y.t, y.ld, y.off, gx.t, gx.ld, gx.off, st["n"],
                                st["ih"], st["iw"], st["c"], st["stride"], accumulate)
      elif gx is not None:
        gy = st["gy"]
        ops.pool3x3_bwd(gy.t, gy.ld, gy.off, st["arg"], gx.t, gx.ld, gx.off, st["n"], st["ih"],
                        st["iw"], st["c"], st["stride"], st["mode"], accumulate)
    else:
      # Inception block.  Everything but each branch's first op runs branch by branch; the
      # first ops all produce a gradient w.r.t. the shared block input, which TF sums (AddN):
      # the stride-1 1x1 entry convolutions are fused into ONE multi-segment GEMM that writes
      # the sum once, the remaining first ops (pools) accumulate into it afterwards.
      firsts = [bsteps[0] for bsteps in st["branches"]]

      fused = [b for b in firsts if gx is not None and b["kind"] == "conv" and
               b["layer"].k == 1 and b["layer"].stride == 1 and "dc_entry" in b]
      if len(fused) < 2:
        fused = []
      entry_wg = {}          # branch index -> deferred filter gradient of its fused entry convolution

      def tail(rest_owner):
        """Everything of a branch behind its first op, then — a fused entry convolution — the
        BN/ReLU backward of the first op itself (its dC feeds the block's entry GEMM and the
        grouped entry filter gradient): on whichever stream runs the branch."""
        bsteps = st["branches"][rest_owner]
        for j in range(len(bsteps) - 1, 0, -1):
          self._bwd_step(plan, bsteps[j], bsteps[j]["x"], bsteps[j - 1]["gy"], False)
        b0 = bsteps[0]
        if any(b0 is f for f in fused):
          got = []
          self._conv_bwd(plan, b0, x, None, False, dc=b0["dc_entry"], defer=got)
          entry_wg[rest_owner] = got
      early = None       # a pooling FIRST op whose gradient goes out on the branch stream (below)
      if self.alt is not None and plan["scr_b"] is None and st["n"] >= self.alt_min_n:
        plan["scr_b"] = plan["scratch_set"]()
      if self.alt is not None and plan["scr_b"] is not None and st["n"] >= self.alt_min_n:
        mains, others = self._split_branches(st, None)
        # The block-input gradient is the sum over the branches' first ops.  A pooling branch's share
        # (Mixed_5a: 226 MB of max-pool gradient, Mixed_5c: 131 MB) only needs the block's output
        # gradient: it is written on the branch stream, under the convolution tails of the other
        # branches, as the FIRST writer of the block-input gradient; the entry GEMM accumulates.
        pools = [bi for bi, b in enumerate(firsts) if b["kind"] == "pool" and not b.get("relu")]
        rest_firsts = [b for b in firsts if not any(b is f for f in fused)]
        if (gx is not None and len(pools) == 1 and len(rest_firsts) == 1 and fused and
            all(bi != pools[0] for bi, _ in mains)):
          early = pools[0]
          if all(bi != early for bi, _ in others):
            others = others + [(early, [])]     # (a branch that is ONLY the pool: no tail)
        # (branches that are ONLY a fused entry convolution: their BN/ReLU backward, see tail())
        seen = set(bi for bi, _ in mains + others)
        lone = [(bi, []) for bi, b in enumerate(firsts)
                if bi not in seen and any(b is f for f in fused)]
        if others or (mains and lone):
          others = others + lone
        else:
          mains = mains + lone
        if others:
          fork = torch.cuda.Event()
          fork.record()
          self.alt.wait_event(fork)
          with torch.cuda.stream(self.alt):
            plan["on_alt"] = True
            try:
              for bi, _ in others:
                tail(bi)
                if bi == early:
                  self._bwd_step(plan, firsts[bi], x, gx, False)
            finally:
              plan["on_alt"] = False
            joined = torch.cuda.Event()
            joined.record()
        for bi, _ in mains:
          tail(bi)
        if others:
          torch.cuda.current_stream().wait_event(joined)
      else:
        for bi in range(len(st["branches"])):
          tail(bi)
      written = early is not None
      owner = st.get("fuse_out")
      if owner is not None and gx is not None and len(fused) >= 2:
        # block boundary fusion (see _prepare_backward): the other first ops (the pooling branch)
        # write the block-input gradient first, the multi-segment GEMM accumulates onto it and —
        # last writer — applies the BN/ReLU backward of the producers of the block input
        for bi, b in enumerate(firsts):
          if not any(b is f for f in fused) and bi != early:
            self._bwd_step(plan, b, x, gx, written)
            written = True
        deferred = [d for bi in sorted(entry_wg) for d in entry_wg[bi]]
        self._entry_wgrads(plan, x, deferred)
        rows = st["n"] * st["ih"] * st["iw"]
        ws0 = owner["ws_off"]
        if "prods" not in owner:
          owner["prods"] = ops.bn_producers(owner["prod_list"])
        segs = [self._entry_dc(b) for b in fused]
        ops.conv1x1_dgrad_multi_bn_relu(
            [t for t, _, _ in segs], [ld for _, ld, _ in segs], [off for _, _, off in segs],
            [b["layer"].w_for(self.dtype) for b in fused], [b["layer"].cout for b in fused],
            x.t, x.ld, x.off, owner["prods"], gx.t, gx.ld, gx.off,
            plan["bn_ws"][ws0:ws0 + owner["nb"] * 2 * owner["ctot"]], rows, st["cin"], written)
        return
      if len(fused) >= 2:
        deferred = [d for bi in sorted(entry_wg) for d in entry_wg[bi]]
        self._entry_wgrads(plan, x, deferred)
        rows = st["n"] * st["ih"] * st["iw"]
        segs = [self._entry_dc(b) for b in fused]
        cin = st["cin"]
        ops.conv1x1_dgrad_multi(
            [t for t, _, _ in segs], [ld for _, ld, _ in segs], [off for _, _, off in segs],
            [b["layer"].w_for(self.dtype) for b in fused],
            [b["layer"].cout for b in fused], gx.t, gx.ld, gx.off, rows, cin, written)
        written = True
      for bi, b in enumerate(firsts):
        if any(b is f for f in fused) or bi == early:
          continue
        self._bwd_step(plan, b, x, gx, written)
        if gx is not None:
          written = True


# ----------------------------------------------------------------------------------------
# the Fast-RCNN feature path
# ----------------------------------------------------------------------------------------

def _x9_unbind(*arena_ptrs):
  """weakref.finalize callback of FrcnnEngine.enable_f32x9 (module level: holds no engine)."""
  from cap2det_amd import _lib
  try:
    for ptr in arena_ptrs:
      _lib.call("c2d_f32x9_unbind", ptr)
  except Exception:   # noqa: BLE001 -- interpreter shutdown
    pass


class FrcnnEngine(object):
  """extract_frcnn_feature (models/utils.py:108-188) on MI355X."""

  STEM = FIRST_SCOPE + "Conv2d_1a_7x7"

  def __init__(self, store, options, bn_scale=True, depth_multiplier=1.0,
               act_dtype=torch.float32, first_stage_dtype=None):
    """options: FRCNN proto (protos/frcnn.proto:4-33).  act_dtype: storage of the per-ROI
    tensors (ROI crop output, second stage activations and gradients): fp32, or bf16 with fp32
    accumulation.  first_stage_dtype: storage of the single-image tower behind the stem
    convolution (default: act_dtype; C2D_TUNE=first_stage_fp32=1 keeps it fp32 in a bf16 network, as
    rounds 2-3a ran it).  The stem, the ROI crop's input map and its gradient, the heads, all
    statistics and all variables stay fp32."""
    self.store = store
    self.act_dtype = act_dtype
    if first_stage_dtype is None:
      first_stage_dtype = torch.float32 if tune.get("first_stage_fp32") == "1" else act_dtype
    self.first_dtype = first_stage_dtype
    self.device = store.device
    self.options = options
    self.crop = options.initial_crop_size
    self.pool_k = options.maxpool_kernel_size
    self.pool_s = options.maxpool_stride
    self.keep_prob = options.dropout_keep_prob
    # slim.dropout on features_to_crop (models/utils.py:138-142): true by proto default
    # (protos/frcnn.proto:29), false in every shipped config (configs/*.pbtxt:55)
    self.dropout_on_feature_map = bool(options.dropout_on_feature_map)
    self.stats = DerivedStore(self.device)
    dm = depth_multiplier
    self.stem_cout = max(int(64 * dm), 16)
    self.stem_mult = min(int(self.stem_cout / 3), 8)
    # stem variables (frozen in every config: first_stage multiplier 0.0); kept outside the
    # flat trainable store.
    dev = self.device
    self.stem_vars = {
        self.STEM + "/depthwise_weights": torch.zeros(7, 7, 3, self.stem_mult, device=dev),
        self.STEM + "/pointwise_weights": torch.zeros(1, 1, 3 * self.stem_mult, self.stem_cout,
                                                      device=dev),
        self.STEM + "/BatchNorm/beta": torch.zeros(self.stem_cout, device=dev),
        self.STEM + "/BatchNorm/moving_mean": torch.zeros(self.stem_cout, device=dev),
        self.STEM + "/BatchNorm/moving_variance": torch.ones(self.stem_cout, device=dev),
    }
    if bn_scale:
      self.stem_vars[self.STEM + "/BatchNorm/gamma"] = torch.ones(self.stem_cout, device=dev)
    self.stem_kpad = 208  # 7*7*4 = 196 padded to a multiple of 16
    self.stem_wt = torch.zeros(1, self.stem_cout, self.stem_kpad, device=dev)
    self.stem_scale = torch.empty(self.stem_cout, device=dev)
    self.stem_shift = torch.empty(self.stem_cout, device=dev)
    self.first = Net(store, self.stats, FIRST_STAGE_AFTER_STEM, FIRST_SCOPE, self.stem_cout,
                     bn_scale, dm, dtype=first_stage_dtype)
    self.first.group_small = True
    self.second = Net(store, self.stats, SECOND_STAGE, SECOND_SCOPE, self.first.cout, bn_scale, dm,
                      dtype=act_dtype)
    self.feature_dims = self.second.cout
    # Second-stage filter gradients run on a side stream beside the input-gradient GEMMs (Net.
    # _conv_bwd): -4 % step time.  C2D_TUNE=streams=0 keeps everything on one stream (per-
    # kernel durations are then separable: profiles/README.md); hipGraph capture turns it off too.
    self.prefetch_stream = None
    if tune.on("streams") and torch.device(store.device).type == "cuda":
      self.second.side = torch.cuda.Stream(device=store.device)
      self.prefetch_stream = torch.cuda.Stream(device=store.device)
      # the short branches of a second-stage Inception block on a branch stream beside the long one
      # (Net._fwd_step / _bwd_step): measured fp32 11.40 -> 11.27 ms, bf16 3.37 -> 3.26 ms per step
      if tune.on("branch_streams"):
        self.second.alt = torch.cuda.Stream(device=store.device)
        self.first.alt = self.second.alt
        self.first.alt_min_n = 1
    self._shape_cache = {}
    self.first_trainable_idx = None
    self.last_crop_bwd = None      # which ROI-crop backward the last backward() ran (bench / tests)
    self.force_atomic_crop_bwd = False   # tests: run c2d_roi_crop_pool_bwd where the row-owner form fits

  # -- variables --------------------------------------------------------------------
  def finalize(self, extra_transposes=()):
    """Call after VariableStore.finalize(): allocates the flat statistics / derived buffers and
    prepares the batched-refresh descriptor tables.  extra_transposes: (var_name, derived_name,
    taps, rows, cols) operands that the owner wants refreshed in the same launch."""
    for _, dname, taps, rows, cols in extra_transposes:
      self.stats.declare_derived(dname, (taps, cols, rows))
    self.stats.finalize()
    self._extra_transposes = list(extra_transposes)
    self._tables = {}
    # fp32 networks on a GPU: the big forward / input-gradient GEMMs as nine bf16 partial products
    # (C2D_TUNE=f32x9=0 keeps them on the fp32 matrix pipe)
    if (self.act_dtype == torch.float32 and torch.device(self.device).type == "cuda" and
        self.store.values is not None and tune.on("f32x9")):
      self.enable_f32x9()

  def _layers(self, only_trainable):
    out = []
    for net in (self.first, self.second):
      for L in net.layers.values():
        if L.trainable or not only_trainable:
          out.append(L)
    return out

  def _build_tables(self, only_trainable):
    import numpy as np
    layers = self._layers(only_trainable)
    tdt = np.dtype([("src", "<i8"), ("dst", "<i8"), ("taps", "<i4"), ("rows", "<i4"),
                    ("cols", "<i4"), ("begin", "<i4")])
    fdt = np.dtype([("gamma", "<i8"), ("beta", "<i8"), ("mean", "<i8"), ("var", "<i8"),
                    ("scale", "<i8"), ("shift", "<i8"), ("c", "<i4"), ("begin", "<i4")])
    trans, folds, tiles, chans = [], [], 0, 0
    voff, soff, doff = self.store.offset, self.stats.stat_off, self.stats.der_off
    items = [(voff[L.name + "/weights"][0], doff[L.name + "/wt"], L.k * L.k, L.cin, L.cout)
             for L in layers]
    items += [(voff[v][0], doff[dn], taps, rows, cols)
              for v, dn, taps, rows, cols in self._extra_transposes]
    for src, dst, taps, rows, cols in items:
      trans.append((src, dst, taps, rows, cols, tiles))
      tiles += taps * (-(-rows // 32)) * (-(-cols // 32))
    for L in layers:
      g = voff[L.name + "/BatchNorm/gamma"][0] if L.bn_scale else -1
      folds.append((g, voff[L.name + "/BatchNorm/beta"][0],
                    soff[L.name + "/BatchNorm/moving_mean"],
                    soff[L.name + "/BatchNorm/moving_variance"], doff[L.name + "/scale"],
                    doff[L.name + "/shift"], L.cout, chans))
      chans += L.cout
    def dev(arr):
      return torch.from_numpy(arr.view(np.uint8).copy()).to(self.device)
    return dict(trans=dev(np.array(trans, dtype=tdt)), ntrans=len(trans), tiles=tiles,
                folds=dev(np.array(folds, dtype=fdt)) if folds else None, nfolds=len(folds),
                chans=chans)

  def values_mirror(self):
    """The bf16 mirror of the variable store when this is a bf16 network whose mirrors exist
    (an optimiser kernel that updates it itself saves refresh() a cast pass), else None."""
    if self.act_dtype == torch.float32:
      return None
    return getattr(self.store, "values_bf16", None)

  def refresh(self, only_trainable=False, values_mirrored=False):
    """Re-derives every (trainable) layer's kernel operands with two batched launches.
    values_mirrored: the caller's optimiser kernel has already written the bf16 mirror of the
    updated variables (c2d_adagrad_step_multi)."""
    if not only_trainable:
      self._refresh_stem()
      self.invalidate_prefetch()             # frozen weights may have changed under a look-ahead
    key = (bool(only_trainable), tuple(L.name for L in self._layers(only_trainable)))
    if self._tables.get("key") != key:
      self._tables = dict(key=key, t=self._build_tables(only_trainable))
    t = self._tables["t"]
    low = self.act_dtype != torch.float32
    # bf16 mirrors of the variables (dgrad operand) and of the derived operands (forward).  The
    # per-step refresh of the trainable layers writes the mirror of a transposed operand with the
    # operand (the folded BatchNorm scale / shift are read in fp32 only); the full refresh casts
    # both buffers whole.
    fused_mirror = low and only_trainable
    if low:
      any_layer = next(iter(self.second.layers.values()))
      any_layer.w_for(self.act_dtype); any_layer.wt_for(self.act_dtype)     # (allocate once)
    if t["ntrans"]:
      if fused_mirror:
        ops.transpose_taps_batched_mirror(t["trans"], t["ntrans"], t["tiles"], self.store.values,
                                          self.stats.der_flat, self.stats.der_bf16)
      else:
        ops.transpose_taps_batched(t["trans"], t["ntrans"], t["tiles"], self.store.values,
                                   self.stats.der_flat)
    if t["nfolds"]:
      ops.bn_fold_batched(t["folds"], t["nfolds"], t["chans"], self.store.values,
                          self.stats.stat_flat, BN_EPS, self.stats.der_flat)
    if low:
      if not values_mirrored:
        ops.cast_bf16(self.store.values, self.store.values_bf16)
      if not fused_mirror:
        ops.cast_bf16(self.stats.der_flat, self.stats.der_bf16)
    if getattr(self, "_x9", None) is not None:
      self._x9_split(only_trainable)

  def enable_f32x9(self, on=True):
    """The fp32 network's forward / input-gradient GEMMs as nine bf16 partial products (DESIGN.md
    section 5, profiles/r06_f32x9/): binds bf16 plane arenas to the variable store and to the derived
    operands (c2d_f32x9_bind) and keeps them current in refresh().  The bindings are keyed by
    address: they are removed when this engine (and with it the two arenas) goes away."""
    if self.act_dtype != torch.float32:
      raise ValueError("f32x9 is a form of the fp32 network")
    if getattr(self, "_x9_finalizer", None) is not None:
      self._x9_finalizer()              # unbinds (idempotent)
      self._x9_finalizer = None
    self._x9 = None
    if not on:
      return
    pv = torch.zeros(3, -(-self.store.values.numel() // 8) * 8, device=self.device, dtype=torch.bfloat16)
    pd = torch.zeros(3, -(-self.stats.der_flat.numel() // 8) * 8, device=self.device, dtype=torch.bfloat16)
    ops.f32x9_bind(self.store.values, pv)
    ops.f32x9_bind(self.stats.der_flat, pd)
    self._x9 = (pv, pd)
    import weakref
    self._x9_finalizer = weakref.finalize(self, _x9_unbind, self.store.values.data_ptr(),
                                          self.stats.der_flat.data_ptr())
    self._x9_split()

  def _x9_split(self, only_trainable=False):
    """Re-splits the weight planes: whole arenas, or (the per-step refresh) the two spans the
    optimiser step and the operand refresh of the trainable layers just rewrote."""
    pv, pd = self._x9
    if only_trainable:
      spans = self._x9_spans()
      if spans is not None:
        (vlo, vhi), (dlo, dhi) = spans
        ops.split3_span(self.store.values, pv, vlo, vhi)
        ops.split3_span(self.stats.der_flat, pd, dlo, dhi)
        return
    ops.split3_bf16(self.store.values, pv)
    ops.split3_bf16(self.stats.der_flat, pd)

  def _x9_spans(self):
    """((lo, hi) of the variable store, (lo, hi) of the derived operands) covering every trainable
    convolution's weights / transposed weights and the extra transposes (the heads); None when
    nothing trains.  Multiples of 4 elements."""
    key = tuple(L.name for L in self._layers(True))
    if getattr(self, "_x9_span_key", None) != key:
      names = [L.name + "/weights" for L in self._layers(True)] + [v for v, _, _, _, _ in self._extra_transposes]
      ders = ([(self.stats.der_off[L.name + "/wt"], L.k * L.k * L.cin * L.cout) for L in self._layers(True)] +
              [(self.stats.der_off[dn], taps * rows * cols) for _, dn, taps, rows, cols in self._extra_transposes])
      if not names:
        self._x9_span_val = None
      else:
        vlo, vhi = self.store.span(names)
        dlo = min(o for o, _ in ders) // 4 * 4
        dhi = -(-max(o + n for o, n in ders) // 4) * 4
        self._x9_span_val = ((vlo // 4 * 4, -(-vhi // 4) * 4), (dlo, min(dhi, self.stats.der_flat.numel())))
      self._x9_span_key = key
    return self._x9_span_val

  def _refresh_stem(self):
    """Folds depthwise(7x7, x8) o pointwise(1x1) into one 7x7 kernel over the 4-channel padded
    image (setup-time, frozen layer): W[ky,kx,ci,co] = sum_m dw[ky,kx,ci,m] * pw[ci*8+m, co]."""
    sv = self.stem_vars
    dw = sv[self.STEM + "/depthwise_weights"]
    pw = sv[self.STEM + "/pointwise_weights"].view(3, self.stem_mult, self.stem_cout)
    weff = torch.einsum("klcm,cmo->klco", dw, pw)                     # [7,7,3,cout]
    w4 = torch.zeros(7, 7, 4, self.stem_cout, device=self.device)
    w4[:, :, :3] = weff
    self.stem_wt.zero_()
    self.stem_wt[0, :, :196] = w4.reshape(196, self.stem_cout).t()
    ops.bn_fold(sv.get(self.STEM + "/BatchNorm/gamma"), sv[self.STEM + "/BatchNorm/beta"],
                sv[self.STEM + "/BatchNorm/moving_mean"],
                sv[self.STEM + "/BatchNorm/moving_variance"], BN_EPS, self.stem_scale,
                self.stem_shift)

  generation = 0     # bumped whenever launch plans / buffers are dropped: recorded step plans die with them

  def set_trainable(self, trainable_names):
    """trainable_names: set of variable names with a positive gradient multiplier."""
    self.generation += 1
    idx = None
    for net in (self.first, self.second):
      for layer in net.layers.values():
        layer.trainable = (layer.name + "/weights") in trainable_names
      net._plans.clear()           # (launch plans depend on which layers train)
    self.invalidate_prefetch()
    self._shape_cache.clear()
    for i, names in enumerate(self.first.order):
      if any(self.first.layers[n].trainable for n in names):
        idx = i
        break
    self.first_trainable_idx = idx
    self.invalidate_prefetch()

  # -- shapes -----------------------------------------------------------------------
  def _buffers(self, b, h, w, n, training):
    key = (b, h, w, n, training)
    if key in self._shape_cache:
      return self._shape_cache[key]
    dev = self.device
    sh, sw = _out_hw(h, w, 2)
    bufs = dict(
        x4=torch.empty(b * h * w, 4, device=dev),
        cols=torch.empty(b * sh * sw, self.stem_kpad, device=dev),
        stem=Ref(torch.empty(b * sh * sw, self.stem_cout, device=dev), self.stem_cout, 0,
                 self.stem_cout),
        sh=sh, sw=sw)
    bufs["stem_in"] = bufs["stem"]          # what the first stage reads
    if self.first_dtype != torch.float32:
      bufs["stem_in"] = Ref(torch.empty(b * sh * sw, self.stem_cout, device=dev, dtype=self.first_dtype),
                            self.stem_cout, 0, self.stem_cout)
    bufs["plan1"] = self.first.plan(b, sh, sw, training)
    fh, fw = bufs["plan1"]["oh"], bufs["plan1"]["ow"]
    p = (self.crop - self.pool_k) // self.pool_s + 1
    d = self.first.cout
    bufs.update(fh=fh, fw=fw, p=p, b=b, n=n,
                pooled=Ref(torch.empty(b * n * p * p, d, device=dev, dtype=self.act_dtype), d, 0, d),
                pool_arg=torch.empty(b * n * p * p, d, dtype=torch.uint8, device=dev),
                box_ind=torch.arange(b, device=dev, dtype=torch.int32).repeat_interleave(n)
                .contiguous())
    bufs["plan2"] = self.second.plan(b * n, p, p, training)
    bufs["spatial"] = bufs["plan2"]["oh"] * bufs["plan2"]["ow"]
    bufs["features"] = torch.empty(b * n, self.feature_dims, device=dev)
    bufs["mask"] = torch.empty(b * n, self.feature_dims, dtype=torch.uint8, device=dev)
    self._shape_cache[key] = bufs
    return bufs

  def _crop_bwd_ws_ok(self, bufs, d):
    """The atomic-free row-owner backward covers this CALL — map width (up to 255 columns: the
    reference's 1000-px training images give up to ~100), pooled-gradient bytes, list and row-table
    limits (c2d_roi_crop_pool_bwd_ws_shape_supported); decided once per shape, so a batch the strip
    form cannot take (e.g. more than 21 images of 84-column maps, or >= 19,022 fp32 boxes) runs the
    atomic kernel instead of raising C2D_ERR_UNSUPPORTED inside the step."""
    if "crop_ws_ok" not in bufs:
      pooled = bufs["pooled"].t
      bufs["crop_ws_ok"] = ops.roi_crop_pool_bwd_ws_shape_supported(
          bufs["b"], bufs["fh"], bufs["fw"], d, bufs["b"] * bufs["n"], self.crop, self.pool_k,
          self.pool_s, pooled.element_size()) > 0
    return bufs["crop_ws_ok"]

  def _crop_ws(self, bufs, b, n, d):
    if "crop_ws" not in bufs:
      nbytes = ops.roi_crop_pool_bwd_workspace_bytes(b, bufs["fh"], bufs["fw"], d, b * n,
                                                     self.crop, self.pool_k, self.pool_s)
      bufs["crop_ws"] = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
    return bufs["crop_ws"]

  # -- frozen prefix of the first stage, one image ahead ----------------------------------
  def _prefix_len(self, bufs):
    """Steps of plan1 in front of the first trainable layer (all of them when the whole first
    stage is frozen): they depend on the image only, never on the optimiser."""
    nsteps = len(bufs["plan1"]["steps"])
    return nsteps if self.first_trainable_idx is None else self.first_trainable_idx

  @staticmethod
  def _output_refs(st):
    refs = [st["y"]]
    if st["kind"] == "block":
      refs += [bsteps[-1]["y"] for bsteps in st["branches"]]
    return refs

  def _run_prefix(self, bufs, image, upto):
    b, h, w, _ = image.shape
    ops.preprocess_pad4(image, bufs["x4"])
    ops.im2col4(bufs["x4"], bufs["cols"], b, h, w, 7, 7, 2, self.stem_kpad)
    st = bufs["stem"]
    ops.conv_fwd(bufs["cols"], self.stem_kpad, 0, self.stem_wt, self.stem_scale, self.stem_shift,
                 st.t, st.ld, 0, b * bufs["sh"] * bufs["sw"], 1, 1, self.stem_kpad, self.stem_cout,
                 1, 1, 1, True)
    if bufs["stem_in"] is not st:
      ops.cast_bf16(st.t, bufs["stem_in"].t)
    self.first.forward(bufs["plan1"], bufs["stem_in"], 0, upto)

  def prefetch_first_stage(self, image, num_proposals, is_training=True):
    """Starts the frozen part of the first stage for the NEXT step's image on a side stream, so
    that its ~50 small, launch-latency-bound kernels run under the current step's GEMMs instead
    of in front of the next one's (a data-loader style look-ahead: the same work per step, one
    step earlier).  Its output goes to an alternate buffer that `forward` swaps in when it is
    called with the same image tensor; anything else simply recomputes."""
    if self.prefetch_stream is None:
      return
    b, h, w, _ = image.shape
    bufs = self._shape_cache.get((b, h, w, num_proposals, is_training))
    if bufs is None or "prefix_free" not in bufs:
      return                                   # first step of this shape: nothing to overlap yet
    upto = self._prefix_len(bufs)
    if upto == 0:
      return
    last = bufs["plan1"]["steps"][upto - 1]
    refs = self._output_refs(last)
    if "prefix_alt" not in bufs:
      bufs["prefix_alt"] = torch.empty_like(last["y"].t)
    cur, alt = last["y"].t, bufs["prefix_alt"]
    stream = self.prefetch_stream
    stream.wait_event(bufs["prefix_free"])     # the prefix's internal buffers are idle again
    # ... and not before everything the caller has queued so far (the forward pass): the
    # look-ahead then starts in the loss phase, whose dozen tiny kernels leave the GPU empty
    late = torch.cuda.Event()
    late.record()
    stream.wait_event(late)
    for r in refs:
      r.t = alt
    try:
      with torch.cuda.stream(stream):
        self._run_prefix(bufs, image, upto)
        done = torch.cuda.Event()
        done.record()
    finally:
      for r in refs:
        r.t = cur                              # the current step still reads its own features
    # (the image tensor itself is kept: while it is referenced here its memory cannot be handed
    # to another tensor, so "same data_ptr and version" really means "same pixels")
    bufs["prefetched"] = (image, image._version, done)

  static_prefix = False      # see forward(): set by a Trainer that replays step plans

  def step_zero_list(self, image_shape, num_proposals):
    """Gradient maps the coming backward pass of this input shape accumulates into (the ROI-crop
    backward's destination), for the trainer's one-launch zeroing at the start of a step; a
    buffer that does not exist yet (first step of a shape) is zeroed where it is created."""
    b, h, w = int(image_shape[0]), int(image_shape[1]), int(image_shape[2])
    bufs = self._shape_cache.get((b, h, w, int(num_proposals), True))
    t = bufs.get("gcrop_buf") if bufs is not None else None
    out = [t] if t is not None else []
    self._step_zeroed = set(x.data_ptr() for x in out)
    return out

  def invalidate_prefetch(self):
    for bufs in self._shape_cache.values():
      pre = bufs.pop("prefetched", None)
      if pre is not None:                      # later work must still come after the look-ahead
        torch.cuda.current_stream().wait_event(pre[2])

  # -- forward / backward -------------------------------------------------------------
  FMAP_SEED = 0x9E3779B97F4A7C15      # decorrelates the feature-map mask from the ROI-feature mask

  def forward(self, image, proposals, is_training, dropout_seed=None, dropout_mask=None,
              feature_map_dropout_mask=None):
    """image [B,H,W,3] fp32 0..255; proposals [B,N,4].  Returns (features [B*N, D], ctx).
    dropout_mask / feature_map_dropout_mask inject the two slim.dropout draws (parity tests)."""
    b, h, w, _ = image.shape
    n = proposals.shape[1]
    bufs = self._buffers(b, h, w, n, is_training)
    upto = self._prefix_len(bufs)
    pre = bufs.pop("prefetched", None)
    if (pre is not None and upto > 0 and pre[0].data_ptr() == image.data_ptr() and
        pre[0].shape == image.shape and pre[0]._version == pre[1] == image._version):
      # the look-ahead of the previous step computed this image's prefix ...
      last = bufs["plan1"]["steps"][upto - 1]
      cur = last["y"].t
      if self.static_prefix:
        # ... into the look-ahead buffer, copied into place (2.4 MB): every step then reads and
        # writes the SAME buffers, which is what a recorded step plan needs (the addresses of the
        # prefix output also sit inside host descriptor tables of the grouped launches)
        torch.cuda.current_stream().wait_event(pre[2])
        ops.copy_bytes(bufs["prefix_alt"], cur)
      else:
        # ... swap its buffer in
        for r in self._output_refs(last):
          r.t = bufs["prefix_alt"]
        bufs["prefix_alt"] = cur
        torch.cuda.current_stream().wait_event(pre[2])
    else:
      if pre is not None:
        torch.cuda.current_stream().wait_event(pre[2])   # (unused look-ahead: just order after it)
      self._run_prefix(bufs, image, upto)
    if self.prefetch_stream is not None:
      bufs["prefix_free"] = torch.cuda.Event()
      bufs["prefix_free"].record()
    feat = self.first.forward(bufs["plan1"], bufs["stem_in"], upto, None)
    boxes = proposals.reshape(-1, 4)
    fmask = None
    if feat.t.dtype != torch.float32:
      # the ROI crop interpolates an fp32 map: the bf16 tower's output widened (exact)
      if "feat_f32" not in bufs:
        bufs["feat_f32"] = Ref(torch.empty(feat.t.shape, device=self.device), feat.ld, feat.off, feat.c)
      ops.cast_f32(feat.t, bufs["feat_f32"].t)
      feat = bufs["feat_f32"]
    crop_src = feat.t
    if self.dropout_on_feature_map and is_training and self.keep_prob < 1.0:
      if "fmap_mask" not in bufs:
        bufs["fmap_mask"] = torch.empty(feat.t.shape, dtype=torch.uint8, device=self.device)
        bufs["fmap_dropped"] = torch.empty_like(feat.t)
      fmask = bufs["fmap_mask"]
      if feature_map_dropout_mask is not None:
        fmask.copy_(feature_map_dropout_mask.reshape(fmask.shape))
      elif isinstance(dropout_seed, torch.Tensor):
        raise NotImplementedError("dropout_on_feature_map under hipGraph replay")
      else:
        ops.dropout_mask(fmask, (0 if dropout_seed is None else int(dropout_seed)) ^ self.FMAP_SEED,
                         self.keep_prob)
      # (the Mixed_4e output itself is kept: its ReLU mask is needed by the backward pass)
      ops.spatial_mean_dropout_fwd(feat.t, bufs["fmap_dropped"], fmask, feat.t.shape[0], 1, feat.c,
                                   self.keep_prob)
      crop_src = bufs["fmap_dropped"]
    feat4 = crop_src.view(b, bufs["fh"], bufs["fw"], feat.c)
    ops.roi_crop_pool_fwd(feat4, boxes, bufs["box_ind"], self.crop, self.pool_k, self.pool_s,
                          out=bufs["pooled"].t.view(b * n, bufs["p"], bufs["p"], feat.c),
                          argmax=bufs["pool_arg"].view(b * n, bufs["p"], bufs["p"], feat.c))
    # The row lists of the atomic-free ROI-crop backward depend on the boxes only: they are built
    # now, on the (idle) filter-gradient stream, under the second stage's forward pass.
    crop_ready = None
    if (is_training and self.first_trainable_idx is not None and self._crop_bwd_ws_ok(bufs, feat.c)
        and self.second.side is not None):
      ws = self._crop_ws(bufs, b, n, feat.c)
      fork = torch.cuda.Event()
      fork.record()
      self.second.side.wait_event(fork)
      with torch.cuda.stream(self.second.side):
        ops.roi_crop_pool_bwd_prepare(boxes, bufs["box_ind"], b, bufs["fh"], bufs["fw"], feat.c,
                                      self.crop, self.pool_k, self.pool_s, ws)
        crop_ready = torch.cuda.Event()
        crop_ready.record()
    net = self.second.forward(bufs["plan2"], bufs["pooled"])
    mask = None
    if is_training and self.keep_prob < 1.0:
      mask = bufs["mask"]
      if dropout_mask is not None:
        mask.copy_(dropout_mask.reshape(mask.shape))
      elif isinstance(dropout_seed, torch.Tensor):
        ops.dropout_mask_dev(mask, dropout_seed, self.keep_prob)   # seed read on the device
      else:
        ops.dropout_mask(mask, 0 if dropout_seed is None else dropout_seed, self.keep_prob)
    ops.spatial_mean_dropout_fwd(net.t, bufs["features"], mask, b * n, bufs["spatial"], net.c,
                                 self.keep_prob if mask is not None else 1.0)
    ctx = dict(bufs=bufs, b=b, n=n, boxes=boxes, mask=mask, feat4=feat4, fmask=fmask,
               crop_ready=crop_ready)
    return bufs["features"], ctx

  def backward(self, dfeatures, lddf, dfoff, ctx, after_second_stage=None, after_block=None):
    """dfeatures: [B*N][lddf] buffer holding d(loss)/d(features) at columns [dfoff, dfoff+D).
    after_block(i): Net.backward's `after_step` for the second stage (per-block gradient
    exchange, data_parallel.BlockReducer)."""
    bufs, b, n = ctx["bufs"], ctx["b"], ctx["n"]
    plan2 = bufs["plan2"]
    gnet = self.second.out_grad(plan2, 0)
    if plan2.get("head_ok"):
      # the output convolutions' BN/ReLU backward derives its dy from dfeatures (Net._conv_bwd)
      plan2["head_grad"] = dict(dmean=dfeatures, ld=lddf, off=dfoff, mask=ctx["mask"],
                                mask_ld=self.feature_dims, spatial=bufs["spatial"],
                                keep_prob=self.keep_prob if ctx["mask"] is not None else 1.0)
    else:
      plan2["head_grad"] = None
      ops.spatial_mean_dropout_bwd(dfeatures, lddf, dfoff, gnet.t, ctx["mask"], b * n,
                                   bufs["spatial"], self.feature_dims,
                                   self.keep_prob if ctx["mask"] is not None else 1.0)
    need_first = self.first_trainable_idx is not None
    dpooled = None
    if need_first:
      if "dpooled" not in bufs:
        bufs["dpooled"] = Ref(torch.empty_like(bufs["pooled"].t), bufs["pooled"].ld, 0,
                              bufs["pooled"].c)
      dpooled = bufs["dpooled"]
    # The second stage's filter gradients (side stream) are joined at the END of this function when
    # nobody asks for them earlier: the ROI-crop backward and Mixed_4e's backward pass — 0.3 ms in
    # which the chip is half empty — then run beside the filter gradients the side stream still owes.
    lazy = after_second_stage is None and need_first and self.second.side is not None
    self.second.backward(plan2, bufs["pooled"], 0, dpooled, after_step=after_block, join=not lazy)
    if after_second_stage is not None:
      after_second_stage()
    if need_first:
      plan1 = bufs["plan1"]
      gfeat = self.first.out_grad(plan1, self.first_trainable_idx)
      d = self.first.cout
      g32 = gfeat.t              # fp32 d(loss)/d(first-stage output)
      if g32.dtype != torch.float32:
        # the ROI-crop backward sums in fp32; a bf16 tower takes the rounded sum (cast below)
        if "gfeat_f32" not in bufs:
          bufs["gfeat_f32"] = torch.empty(gfeat.t.shape, device=self.device)
        g32 = bufs["gfeat_f32"]
      gcrop = g32                # d(loss)/d(features_to_crop)
      if ctx.get("fmask") is not None:
        if "gfmap" not in bufs:
          bufs["gfmap"] = torch.empty_like(g32)
        gcrop = bufs["gfmap"]
      if gcrop.data_ptr() not in getattr(self, "_step_zeroed", ()):
        gcrop.zero_()
      bufs["gcrop_buf"] = gcrop
      self._step_zeroed = set()
      dp4 = dpooled.t.view(b * n, bufs["p"], bufs["p"], d)
      arg4 = bufs["pool_arg"].view(b * n, bufs["p"], bufs["p"], d)
      gf4 = gcrop.view(b, bufs["fh"], bufs["fw"], d)
      use_ws = self._crop_bwd_ws_ok(bufs, d) and not self.force_atomic_crop_bwd
      self.last_crop_bwd = ("row-owner strips (atomic-free)" if use_ws
                            else "atomic fallback (c2d_roi_crop_pool_bwd)")
      if use_ws:
        # atomic-free, bitwise reproducible row-owner form (needs a workspace)
        ws = self._crop_ws(bufs, b, n, d)
        if ctx.get("crop_ready") is not None:           # lists built during the forward pass
          torch.cuda.current_stream().wait_event(ctx["crop_ready"])
          ops.roi_crop_pool_bwd_run(dp4, arg4, ctx["boxes"], bufs["box_ind"], gf4, self.crop,
                                    self.pool_k, self.pool_s, ws)
        else:
          ops.roi_crop_pool_bwd_ws(dp4, arg4, ctx["boxes"], bufs["box_ind"], gf4, self.crop,
                                   self.pool_k, self.pool_s, ws)
      else:
        ops.roi_crop_pool_bwd(dp4, arg4, ctx["boxes"], bufs["box_ind"], gf4, self.crop,
                              self.pool_k, self.pool_s)
      if ctx.get("fmask") is not None:          # through the feature-map dropout
        ops.spatial_mean_dropout_bwd(gcrop, d, 0, g32, ctx["fmask"], gcrop.shape[0], 1, d,
                                     self.keep_prob)
      if g32 is not gfeat.t:
        ops.cast_bf16(g32, gfeat.t)
      self.first.backward(plan1, bufs["stem_in"], self.first_trainable_idx, None)
    if lazy:
      self.second.join(plan2)
