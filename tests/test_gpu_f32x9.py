"""f32x9: the fp32 convolutions as nine bf16 partial products (cap2det_amd/csrc/igemm_x9.hip).

  * c2d_split3_bf16 against oracle/ref_split.py, bit for bit, on > 10^7 random fp32 bit patterns
    (subnormals, huge values, Inf / NaN included), and hi + mid + lo == x on the planes the GPU wrote;
  * every second-stage layer of Inception-V2 Mixed_5a-c (models/utils.py:165-167), forward and input
    gradient, at the kernel instance the benchmark-size call (2000 ROIs) dispatches, against the
    float64 oracle at the SAME tolerance as the fp32-MFMA kernels (2e-5 of the tensor's scale), plus
    the fused block-entry forms and the fused BN/ReLU-backward epilogue;
  * the switch: c2d_f32x9_enable(0) / an unbound weight tensor takes the fp32-MFMA kernels.
"""
import numpy as np
import pytest
import torch

from oracle import ref_ops, ref_split
from tests.test_gpu_real_shapes import (DEV, N_BENCH, SECOND_STAGE_LAYERS, FUSED_INNER_LAYERS, _Layer,
                                        _n, _scale_close, _seen, _t)

pytestmark = pytest.mark.gpu
TOL = 2e-5


@pytest.fixture(scope="module")
def ops():
  from cap2det_amd import hip_ops
  hip_ops.set_conv_workspace(None)
  yield hip_ops


_bound = []      # the weight tensors this module bound planes to


def _planes(ops, t):
  _bound.append(t)
  return ops.x9_planes(t)


@pytest.fixture(autouse=True)
def _unbind(ops):
  """Bindings are keyed by address: none may outlive the tensors of the test that made it (and the
  bindings of live engines elsewhere in the session stay: no blanket unbind)."""
  yield
  while _bound:
    ops.f32x9_unbind(_bound.pop().view(-1))


def _is_x9(insts):
  return bool(insts) and all(i.endswith(", 3>") for i in insts)


def test_split3_bits(ops):
  rng = np.random.default_rng(11)
  n = 12_000_000
  u = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
  u[:8] = [0x7f800000, 0xff800000, 0x7fc00000, 0x00000001, 0x807fffff, 0x7f7fffff, 0xff7fffff, 0]
  u[8:4096] &= 0x807fffff                       # a run of subnormals
  u[4096:8192] |= 0x7f000000                    # a run of huge values (and a few Inf / NaN)
  x = torch.from_numpy(u.view(np.float32).copy()).to(DEV)
  planes = torch.zeros(3, n, device=DEV, dtype=torch.bfloat16)
  ops.split3_bf16(x, planes)
  got = planes.view(torch.int16).cpu().numpy().view(np.uint16)
  hi, mid, lo = ref_split.split3(u.view(np.float32))
  assert np.array_equal(got[0], hi) and np.array_equal(got[1], mid) and np.array_equal(got[2], lo)
  xf = u.view(np.float32)
  finite = (u & 0x7f800000) != 0x7f800000
  with np.errstate(invalid="ignore"):
    big = finite & (np.abs(xf.astype(np.float64)) >= 2.0 ** -110)
  assert big.sum() > 10_000_000
  s = ref_split.join3(got[0], got[1], got[2])
  assert np.array_equal(s[big], xf[big].astype(np.float64))
  assert np.all(np.abs(s[finite & ~big] - xf[finite & ~big].astype(np.float64)) < 2.0 ** -133)
  assert np.array_equal(got[0][~finite], (u[~finite] >> 16).astype(np.uint16))
  assert not got[1][~finite].any() and not got[2][~finite].any()


class _X9Layer(_Layer):
  def __init__(self, ops, *args):
    super().__init__(ops, *args, torch.float32)
    self.keep = (_planes(ops, self.w_), _planes(ops, self.wt_))


def _check(ops, name, hw, cin, cout, k, s):
  for what in ("fwd", "dgrad"):
    big = _X9Layer(ops, N_BENCH, hw, cin, cout, k, s, 1)
    big.run(ops, what)
    want_inst = ops.last_dispatch()
    # (launches of fewer than 256 tiles of 128 x 128 — the parity classes of a stride-2 input gradient
    #  over 7x7 maps, the 128-column 1x1 layers over 4x4 maps — keep the fp32 64 x 64 tiles: the
    #  documented exception, include/cap2det_hip.h)
    small = all(i.startswith("igemm_nt_kernel<%d, 2, 2, 1, 1," % (what == "dgrad")) for i in want_inst)
    assert _is_x9(want_inst) or small, (name, what, want_inst)
    del big
    for n in (256, 704, N_BENCH):
      lay = _X9Layer(ops, n, hw, cin, cout, k, s, 7 + len(name))
      got = lay.run(ops, what)
      if ops.last_dispatch() == want_inst:
        break
    else:
      raise AssertionError((name, what, want_inst))
    x64, w64, dc64 = lay.x.astype(np.float64), lay.w.astype(np.float64), lay.dc.astype(np.float64)
    if what == "fwd":
      want = ref_ops.conv2d(x64, w64, s)
    else:
      want, _ = ref_ops.conv2d_backward(x64, w64, dc64, s, need_dx=True)
    _scale_close(_n(got), want, TOL, "%s %s n=%d %s" % (name, what, n, want_inst))
    _seen.update(want_inst)          # (the profile-completeness check of tests/test_gpu_real_shapes.py)
    # ... and the fp32-MFMA kernels with the switch off: same numbers to the same tolerance
    was = ops.f32x9_enable(False)
    try:
      ref = lay.run(ops, what)
      assert not _is_x9(ops.last_dispatch())
    finally:
      ops.f32x9_enable(was)
    _scale_close(_n(got), _n(ref).astype(np.float64), TOL, "%s %s vs fp32 MFMA" % (name, what))
    if small:
      assert torch.equal(got, ref)
    del lay, got, want, ref
  torch.cuda.empty_cache()


@pytest.mark.parametrize("layer", SECOND_STAGE_LAYERS, ids=[l[0] for l in SECOND_STAGE_LAYERS])
def test_second_stage_layer_x9(ops, layer):
  _check(ops, *layer)


ONE_BY_ONE = [l for l in SECOND_STAGE_LAYERS if l[4] == 1 and l[5] == 1]
THREE_BY_THREE = [l for l in SECOND_STAGE_LAYERS if l[4] == 3]          # (stride 1 and the two stride-2 layers)


@pytest.mark.parametrize("layer", ONE_BY_ONE + THREE_BY_THREE,
                         ids=[l[0] for l in ONE_BY_ONE + THREE_BY_THREE])
def test_filter_gradient_x9(ops, layer):
  """c2d_conv_wgrad of the second-stage layers in a process that has planes bound: both fp32 operands
  split by the loader, nine partial products per k16 — wgrad1x1_x9_kernel for the 1x1 layers,
  wgrad3x3_x9_kernel (one tap per block, pixel-ordered rows, padding pairs never staged) for the 3x3
  layers on 4x4 / 7x7 maps — at the benchmark's instance against the float64 oracle at the tolerance
  of the fp32-MFMA kernels, and those kernels with the switch off."""
  name, hw, cin, cout, k, s = layer
  x9_kernel, fp32_kernel = (("wgrad1x1_x9_kernel<", "wgrad_tn_kernel<") if k == 1 else
                            ("wgrad3x3_x9_kernel<", "wgrad3x3_kernel<" if s == 1 else "wgrad3x3_s2_kernel<"))
  big = _X9Layer(ops, N_BENCH, hw, cin, cout, k, s, 1)
  big.run(ops, "wgrad")
  want_inst = ops.last_dispatch()
  assert want_inst and all(i.startswith(x9_kernel) for i in want_inst), want_inst
  del big
  for n in (704, N_BENCH):
    lay = _X9Layer(ops, n, hw, cin, cout, k, s, 7 + len(name))
    got = lay.run(ops, "wgrad")
    if ops.last_dispatch() == want_inst:
      break
  else:
    raise AssertionError((name, want_inst))
  _seen.update(want_inst)
  _, want = ref_ops.conv2d_backward(lay.x.astype(np.float64), lay.w.astype(np.float64),
                                    lay.dc.astype(np.float64), s, need_dx=False)
  _scale_close(_n(got), want, TOL, "%s wgrad n=%d %s" % (name, n, want_inst))
  was = ops.f32x9_enable(False)
  try:
    ref = lay.run(ops, "wgrad")
    assert all(i.startswith(fp32_kernel) for i in ops.last_dispatch()), ops.last_dispatch()
  finally:
    ops.f32x9_enable(was)
  _scale_close(_n(got), _n(ref).astype(np.float64), TOL, "%s wgrad vs fp32 MFMA" % name)
  # odd image counts: the last slab / split ends inside the descriptor's zeros
  odd = max(8192 // (hw * hw) + 3, 261)
  lay = _X9Layer(ops, odd, hw, cin, cout, k, s, 3)
  got = lay.run(ops, "wgrad")
  assert all(i.startswith(x9_kernel) for i in ops.last_dispatch())
  _, want = ref_ops.conv2d_backward(lay.x.astype(np.float64), lay.w.astype(np.float64),
                                    lay.dc.astype(np.float64), s, need_dx=False)
  _scale_close(_n(got), want, TOL, "%s wgrad n=%d" % (name, odd))
  torch.cuda.empty_cache()


@pytest.mark.parametrize("block", ["Mixed_5a", "Mixed_5b", "Mixed_5c"])
def test_block_entry_x9(ops, block):
  """The fused block-entry GEMMs: c2d_conv1x1_fwd_multi (one GEMM over the entry convolutions' output
  columns) and c2d_conv1x1_dgrad_multi (K = the concatenation of the branches), weights in ONE bound
  arena as in the engine."""
  hw, cin, couts = {"Mixed_5a": (7, 576, [128, 192]), "Mixed_5b": (4, 1024, [352, 192, 160, 128]),
                    "Mixed_5c": (4, 1024, [352, 192, 192, 128])}[block]
  rng = np.random.default_rng(41)
  n = 704
  rows = n * hw * hw
  x = rng.standard_normal((rows, cin)).astype(np.float32)
  ws = [(rng.standard_normal((cin, c)) / np.sqrt(cin)).astype(np.float32) for c in couts]
  arena = torch.zeros(2 * sum(w.size for w in ws), device=DEV)
  w_dev, wt_dev, off = [], [], 0
  for w in ws:                      # [cin][cout] (input-gradient operand) and [cout][cin] (forward)
    for store, a in ((w_dev, w), (wt_dev, w.T)):
      v = arena[off:off + a.size].view(a.shape)
      v.copy_(_t(a))
      store.append(v)
      off += a.size
  planes = _planes(ops, arena)
  x_ = _t(x)
  ys = [torch.empty(rows, c, device=DEV) for c in couts]
  ones = [torch.ones(c, device=DEV) for c in couts]
  zeros = [torch.zeros(c, device=DEV) for c in couts]
  outs = ops.conv_outs([(wt_dev[i], ones[i], zeros[i], ys[i], couts[i], 0, couts[i], False)
                        for i in range(len(couts))])
  ops.conv1x1_fwd_multi(x_, cin, 0, outs, rows, cin)
  assert _is_x9(ops.last_dispatch()), ops.last_dispatch()
  _seen.update(ops.last_dispatch())
  for y, w in zip(ys, ws):
    _scale_close(_n(y), x.astype(np.float64) @ w.astype(np.float64), TOL, block + " entry fwd")
  dcs = [rng.standard_normal((rows, c)).astype(np.float32) for c in couts]
  dx = torch.zeros(rows, cin, device=DEV)
  ops.conv1x1_dgrad_multi([_t(a) for a in dcs], couts, [0] * len(couts), w_dev, couts, dx, cin, 0, rows,
                          cin, False)
  assert _is_x9(ops.last_dispatch()), ops.last_dispatch()
  _seen.update(ops.last_dispatch())
  want = sum(a.astype(np.float64) @ w.astype(np.float64).T for a, w in zip(dcs, ws))
  _scale_close(_n(dx), want, TOL, block + " entry dgrad")
  del planes


@pytest.mark.parametrize("layer", FUSED_INNER_LAYERS, ids=[l[0] for l in FUSED_INNER_LAYERS])
def test_fused_dgrad_bn_relu_x9(ops, layer):
  """c2d_conv_dgrad_bn_relu on the f32x9 kernels (the FUSED epilogue of the ring in fp32): dc and the
  two column sums against the float64 oracle."""
  name, hw, cin, cout, k, s = layer
  oh = -(-hw // s)
  n = 704 if s == 1 else N_BENCH      # (a stride-2 parity class needs the benchmark's rows to fill 256 tiles)
  rng = np.random.default_rng(19)
  w = (rng.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
  dc = rng.standard_normal((n, oh, oh, cout)).astype(np.float32)
  y = np.maximum(rng.standard_normal((n, hw, hw, cin)), 0).astype(np.float32)
  scale = rng.uniform(0.5, 1.5, cin).astype(np.float32)
  beta = (0.1 * rng.standard_normal(cin)).astype(np.float32)
  gamma = rng.uniform(0.5, 1.5, cin).astype(np.float32)
  w_ = _t(w).view(k * k, cin, cout)
  planes = _planes(ops, w_)
  nb = ops.conv_dgrad_bn_relu_blocks(torch.float32, n, hw, hw, cin, cout, k, k, s)
  out = torch.full((n * hw * hw, cin), 9.0, device=DEV)
  part = torch.full((nb, 2, cin), 7.0, device=DEV)
  ops.conv_dgrad_bn_relu(_t(dc).view(-1, cout), cout, 0, w_, _t(y).view(-1, cin), cin, 0, _t(scale),
                         _t(beta), _t(gamma), out, part, n, hw, hw, cin, cout, k, k, s)
  inst = ops.last_dispatch()
  if name == "5a/B0/3x3s2":           # (128 columns: its classes stay on the fp32 64 x 64 tiles at any size)
    assert not _is_x9(inst), inst
  else:
    assert _is_x9(inst) and all(", true, 3>" in i for i in inst), inst
  _seen.update(inst)
  dx, _ = ref_ops.conv2d_backward(np.zeros((n, hw, hw, cin)), w.astype(np.float64), dc.astype(np.float64), s)
  dz = dx * (y > 0)
  _scale_close(_n(out).reshape(n, hw, hw, cin), dz * scale, TOL, "%s fused dc %s" % (name, inst))
  sums = part.double().sum(0).cpu().numpy()
  for got, want in ((sums[0], dz.reshape(-1, cin).sum(0)),
                    (sums[1], (dz * (y.astype(np.float64) - beta) / gamma).reshape(-1, cin).sum(0))):
    assert np.abs(got - want).max() <= 1e-4 * max(np.abs(want).max(), 1.0), name
  del planes


def test_unbound_weights_take_fp32_kernels(ops):
  lay = _Layer(ops, 704, 4, 224, 224, 3, 1, 3, torch.float32)
  lay.run(ops, "fwd")
  assert not _is_x9(ops.last_dispatch()), ops.last_dispatch()


def test_accumulating_and_offset_outputs_x9(ops):
  """accumulate = 1 and channel-slice destinations (the Inception concat buffers) through the x9
  epilogue."""
  rng = np.random.default_rng(23)
  n, hw, cin, cout = 704, 4, 160, 224
  w = (rng.standard_normal((3, 3, cin, cout)) / np.sqrt(9 * cin)).astype(np.float32)
  dc = rng.standard_normal((n * hw * hw, cout + 32)).astype(np.float32)
  base = rng.standard_normal((n * hw * hw, cin + 64)).astype(np.float32)
  w_ = _t(w).view(9, cin, cout)
  planes = _planes(ops, w_)
  dx = _t(base).clone()
  ops.conv_dgrad(_t(dc), cout + 32, 16, w_, dx, cin + 64, 32, n, hw, hw, cin, cout, 3, 3, 1, True)
  assert _is_x9(ops.last_dispatch())
  want, _ = ref_ops.conv2d_backward(np.zeros((n, hw, hw, cin)), w.astype(np.float64),
                                    dc[:, 16:16 + cout].reshape(n, hw, hw, cout).astype(np.float64), 1)
  full = base.astype(np.float64).copy()
  full[:, 32:32 + cin] += want.reshape(-1, cin)
  _scale_close(_n(dx), full, TOL, "accumulating dgrad into a channel slice")
  del planes
