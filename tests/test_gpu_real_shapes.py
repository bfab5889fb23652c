"""Oracle parity at the REAL layer shapes and tile paths of the benchmark step.

tests/test_gpu_ops.py checks every kernel on small channel counts; the benchmark (N = 2000
proposals, depth multiplier 1.0) runs other template instances of the same kernels: pixel-major
128x64 tiles with K = 9 x 192 .. 9 x 256, 128x128 / 128x64 row-major tiles with K = 576 / 1024,
the nine-tap filter-gradient kernel, the 4-parity stride-2 input gradient and the multi-segment
1x1 input gradient.  Here every distinct second-stage layer of Inception-V2 Mixed_5a-c
(oracle/ref_model.py SECOND_STAGE = the layer table of models/utils.py:165-167's extractor), the
trainable first-stage block Mixed_4e on its 32x32 map, and the fused heads GEMM go through
conv_fwd / conv_dgrad / conv_wgrad / conv1x1_dgrad_multi against the float64 oracle, and
`c2d_debug_last_dispatch` proves that the kernel instance that produced the numbers is the one
the benchmark-size call dispatches (every instance of the newest committed
profiles/rNN_bench_kernel_stats_c1_serial.csv and of its bf16 twin ..._c2_serial.csv must be hit).

Tolerances: fp32 path 2e-5 relative to the tensor's scale (fp32 MFMA chains of up to 2304 terms
vs float64); bf16 path compared with the oracle evaluated on the same bf16-rounded operands, one
bf16 rounding of the output (2^-8 relative) for bf16 outputs, 1e-4 of scale for the fp32 filter
gradients."""
import numpy as np
import pytest
import torch

from oracle import ref_model, ref_ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
N_BENCH = 2000

# (name, map side, cin, cout, kernel, stride): every distinct convolution of Mixed_5a-c
SECOND_STAGE_LAYERS = [
    ("5a/B0/1x1", 7, 576, 128, 1, 1), ("5a/B0/3x3s2", 7, 128, 192, 3, 2),
    ("5a/B1/1x1", 7, 576, 192, 1, 1), ("5a/B1/3x3", 7, 192, 256, 3, 1),
    ("5a/B1/3x3s2", 7, 256, 256, 3, 2),
    ("5bc/B0/1x1", 4, 1024, 352, 1, 1), ("5bc/B1/1x1", 4, 1024, 192, 1, 1),
    ("5bc/B1/3x3", 4, 192, 320, 3, 1), ("5b/B2/1x1", 4, 1024, 160, 1, 1),
    ("5b/B2/3x3a", 4, 160, 224, 3, 1), ("5bc/B2/3x3b", 4, 224, 224, 3, 1),
    ("5c/B2/3x3a", 4, 192, 224, 3, 1), ("5bc/B3/1x1", 4, 1024, 128, 1, 1),
]
# Mixed_4e on the single 32x32 first-stage map (trainable in voc07_groundtruth) + Mixed_4d's
# widest 3x3 for the small-problem kernels
FIRST_STAGE_LAYERS = [
    ("4e/B0/1x1", 32, 576, 96, 1, 1), ("4e/B1/1x1", 32, 576, 128, 1, 1),
    ("4e/B1/3x3", 32, 128, 192, 3, 1), ("4e/B2/1x1", 32, 576, 160, 1, 1),
    ("4e/B2/3x3a", 32, 160, 192, 3, 1), ("4e/B2/3x3b", 32, 192, 192, 3, 1),
]

def _latest_profile_instances(cfg):
  """igemm / wgrad template instances of the newest committed serial kernel-stats summary of
  BASELINE config `cfg` (profiles/rNN_bench_kernel_stats_<cfg>_serial.csv): the completeness
  assertion at the end of this file follows the profile, not a hand-kept list."""
  import csv, glob, os, re
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  files = sorted(glob.glob(os.path.join(root, "profiles", "r*_bench_kernel_stats_%s_serial.csv" % cfg)))
  assert files, cfg
  out = set()
  with open(files[-1]) as f:
    for row in csv.DictReader(f):
      m = re.search(r"((?:igemm|wgrad)\w*<[^>]*>)", row["Name"])
      if m:
        out.add(m.group(1))
  return files[-1], out


_seen = set()


def _t(a, dtype=None):
  return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(DEV).contiguous()


def _n(t):
  return t.detach().float().cpu().numpy()


@pytest.fixture(scope="module")
def ops():
  from cap2det_amd import hip_ops
  hip_ops.set_conv_workspace(None)
  return hip_ops


def _bf16_round(a):
  return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(torch.bfloat16).float().numpy()


def _scale_close(got, want, tol, what):
  scale = max(float(np.abs(want).max()), 1e-30)
  err = float(np.abs(got.astype(np.float64) - want).max())
  assert err <= tol * scale, "%s: max err %.3e vs scale %.3e (tol %.1e)" % (what, err, scale, tol)


class _Layer(object):
  """Operands of one convolution at `n` images, on the device in fp32 or bf16."""

  def __init__(self, ops, n, hw, cin, cout, k, s, seed, dtype):
    rng = np.random.default_rng(seed)
    self.n, self.hw, self.cin, self.cout, self.k, self.s = n, hw, cin, cout, k, s
    self.oh = -(-hw // s)
    x = rng.standard_normal((n, hw, hw, cin)).astype(np.float32)
    w = (rng.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
    dc = rng.standard_normal((n, self.oh, self.oh, cout)).astype(np.float32)
    self.low = dtype == torch.bfloat16
    if self.low:
      x, w, dc = _bf16_round(x), _bf16_round(w), _bf16_round(dc)
    self.x, self.w, self.dc = x, w, dc
    self.dx_, self.w_, self.dc_ = None, _t(w).to(dtype), _t(dc).to(dtype)
    self.x_ = _t(x).to(dtype)
    self.wt_ = torch.empty(k * k, cout, cin, device=DEV)
    ops.transpose_taps(_t(w), self.wt_, k * k, cin, cout)
    self.wt_ = self.wt_.to(dtype)
    self.dtype = dtype

  def run(self, ops, what):
    n, hw, cin, cout, k, s = self.n, self.hw, self.cin, self.cout, self.k, self.s
    if what == "fwd":
      y = torch.empty(n, self.oh, self.oh, cout, device=DEV, dtype=self.dtype)
      ops.conv_fwd(self.x_, cin, 0, self.wt_, None, None, y, cout, 0, n, hw, hw, cin, cout, k, k, s,
                   False)
      return y
    if what == "dgrad":
      dx = torch.zeros(n, hw, hw, cin, device=DEV, dtype=self.dtype)
      ops.conv_dgrad(self.dc_, cout, 0, self.w_, dx, cin, 0, n, hw, hw, cin, cout, k, k, s, False)
      return dx
    dw = torch.zeros(k, k, cin, cout, device=DEV)
    ops.conv_wgrad(self.x_, cin, 0, self.dc_, cout, 0, dw, n, hw, hw, cin, cout, k, k, s)
    return dw


def _bench_dispatch(ops, hw, cin, cout, k, s, what, dtype):
  """Kernel instances the benchmark-size call (n = 2000 ROIs) dispatches (no oracle needed)."""
  lay = _Layer(ops, N_BENCH if hw <= 7 else 1, hw, cin, cout, k, s, 1, dtype)
  lay.run(ops, what)
  return ops.last_dispatch()


def _check_layer(ops, name, hw, cin, cout, k, s, dtype):
  for what in ("fwd", "dgrad", "wgrad"):
    want_inst = _bench_dispatch(ops, hw, cin, cout, k, s, what, dtype)
    assert want_inst, (name, what)
    # smallest image count whose dispatch equals the benchmark's (the oracle is float64 numpy)
    chosen = None
    for n in ([1] if hw > 7 else [256, 704, N_BENCH]):
      lay = _Layer(ops, n, hw, cin, cout, k, s, 7 + len(name), dtype)
      got = lay.run(ops, what)
      if ops.last_dispatch() == want_inst:
        chosen = n
        break
    assert chosen is not None, (name, what, want_inst)
    _seen.update(want_inst)
    x64, w64, dc64 = lay.x.astype(np.float64), lay.w.astype(np.float64), lay.dc.astype(np.float64)
    low = dtype == torch.bfloat16
    if what == "fwd":
      want = ref_ops.conv2d(x64, w64, s)
      _scale_close(_n(got), want, 1.1 * 2.0 ** -8 if low else 2e-5, "%s fwd n=%d %s" % (name, chosen, want_inst))
    elif what == "dgrad":
      want, _ = ref_ops.conv2d_backward(x64, w64, dc64, s, need_dx=True)
      _scale_close(_n(got), want, 1.1 * 2.0 ** -8 if low else 2e-5, "%s dgrad n=%d %s" % (name, chosen, want_inst))
    else:
      _, want = ref_ops.conv2d_backward(x64, w64, dc64, s, need_dx=False)
      _scale_close(_n(got), want, 1e-4 if low else 2e-5, "%s wgrad n=%d %s" % (name, chosen, want_inst))
    del lay, got, want
  torch.cuda.empty_cache()


@pytest.mark.parametrize("layer", SECOND_STAGE_LAYERS, ids=[l[0] for l in SECOND_STAGE_LAYERS])
def test_second_stage_layer_fp32(ops, layer):
  _check_layer(ops, *layer, dtype=torch.float32)


@pytest.mark.parametrize("layer", FIRST_STAGE_LAYERS, ids=[l[0] for l in FIRST_STAGE_LAYERS])
def test_mixed_4e_layer_fp32(ops, layer):
  _check_layer(ops, *layer, dtype=torch.float32)


@pytest.mark.parametrize("layer", SECOND_STAGE_LAYERS, ids=[l[0] for l in SECOND_STAGE_LAYERS])
def test_second_stage_layer_bf16(ops, layer):
  _check_layer(ops, *layer, dtype=torch.bfloat16)


@pytest.mark.parametrize("layer", FIRST_STAGE_LAYERS, ids=[l[0] for l in FIRST_STAGE_LAYERS])
def test_mixed_4e_layer_bf16(ops, layer):
  """The bf16 step's first stage: igemm_small_kernel<*, 2> forward / input gradient, the per-tap bf16
  filter gradient, on the 32x32 map of one image."""
  _check_layer(ops, *layer, dtype=torch.bfloat16)


@pytest.mark.parametrize("c", [20, 80])
def test_heads_gemm_real_size(ops, c):
  """The five heads fused in one [1024, 2C + 3(C+1)] GEMM on N = 2000 rows (103 -> 112 columns for
  VOC, 403 -> 416 for COCO), forward, input gradient and filter gradient."""
  ncols = 2 * c + 3 * (c + 1)
  npad = -(-ncols // 16) * 16
  lay = _Layer(ops, N_BENCH, 1, 1024, npad, 1, 1, 21 + c, torch.float32)
  x64, w64, dc64 = lay.x.astype(np.float64), lay.w.astype(np.float64), lay.dc.astype(np.float64)
  _scale_close(_n(lay.run(ops, "fwd")), ref_ops.conv2d(x64, w64, 1), 2e-5, "heads fwd")
  _seen.update(ops.last_dispatch())
  dx, dw = ref_ops.conv2d_backward(x64, w64, dc64, 1)
  _scale_close(_n(lay.run(ops, "dgrad")), dx, 2e-5, "heads dgrad")
  _seen.update(ops.last_dispatch())
  _scale_close(_n(lay.run(ops, "wgrad")), dw, 2e-5, "heads wgrad")
  _seen.update(ops.last_dispatch())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("block", ["Mixed_5a", "Mixed_5b", "Mixed_5c"])
def test_block_entry_dgrad_multi(ops, block, dtype):
  """The fused input gradient of an Inception block's 1x1 entry convolutions (one multi-segment
  GEMM, K = sum of the branch widths) at the real widths: 5a 576 <- (128, 192) on 7x7,
  5b 1024 <- (352, 192, 160), 5c 1024 <- (352, 192, 192) on 4x4 [the pooled branch's 1x1 is not an
  entry convolution]."""
  hw, cin, couts = {"Mixed_5a": (7, 576, [128, 192]), "Mixed_5b": (4, 1024, [352, 192, 160]),
                    "Mixed_5c": (4, 1024, [352, 192, 192])}[block]
  low = dtype == torch.bfloat16
  rng = np.random.default_rng(31)

  def run(n):
    rows = n * hw * hw
    dcs = [rng.standard_normal((rows, c)).astype(np.float32) for c in couts]
    ws = [(rng.standard_normal((cin, c)) / np.sqrt(c * len(couts))).astype(np.float32) for c in couts]
    if low:
      dcs, ws = [_bf16_round(a) for a in dcs], [_bf16_round(a) for a in ws]
    dx = torch.zeros(rows, cin, device=DEV, dtype=dtype)
    ops.conv1x1_dgrad_multi([_t(a).to(dtype) for a in dcs], couts, [0] * len(couts),
                            [_t(a).to(dtype) for a in ws], couts, dx, cin, 0, rows, cin, False)
    return dcs, ws, dx, ops.last_dispatch()

  _, _, _, want_inst = run(N_BENCH)
  for n in (256, 704, N_BENCH):
    dcs, ws, dx, inst = run(n)
    if inst == want_inst:
      break
  assert inst == want_inst
  _seen.update(inst)
  want = sum(a.astype(np.float64) @ w.astype(np.float64).T for a, w in zip(dcs, ws))
  _scale_close(_n(dx), want, 1.1 * 2.0 ** -8 if low else 2e-5, "%s entry dgrad %s" % (block, inst))



# consumers whose input gradient carries the BN/ReLU backward of the convolution that produced their
# input (Net._prepare_backward, form 1): (name, map side, cin = producer's cout, cout, kernel, stride)
FUSED_INNER_LAYERS = [
    ("5a/B0/3x3s2", 7, 128, 192, 3, 2), ("5a/B1/3x3", 7, 192, 256, 3, 1), ("5a/B1/3x3s2", 7, 256, 256, 3, 2),
    ("5bc/B1/3x3", 4, 192, 320, 3, 1), ("5b/B2/3x3a", 4, 160, 224, 3, 1), ("5bc/B2/3x3b", 4, 224, 224, 3, 1),
    ("5c/B2/3x3a", 4, 192, 224, 3, 1),
]


@pytest.mark.parametrize("layer", FUSED_INNER_LAYERS, ids=[l[0] for l in FUSED_INNER_LAYERS])
def test_second_stage_fused_dgrad_bf16(ops, layer):
  """Round 5: bf16 networks fuse the producer's BN/ReLU backward into the consumer's input-gradient
  GEMM (igemm_ring_kernel<..., FUSED = true>).  Every fused layer of Mixed_5a-c at the instance the
  benchmark-size call dispatches: dc within one bf16 rounding of the float64 oracle on the same
  bf16 operands, the fp32 column sums to 1e-3 of their scale."""
  name, hw, cin, cout, k, s = layer
  oh = -(-hw // s)

  def run(n, check):
    rng = np.random.default_rng(17 + n)
    w = _bf16_round((rng.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32))
    dc = _bf16_round(rng.standard_normal((n, oh, oh, cout)).astype(np.float32))
    y = _bf16_round(np.maximum(rng.standard_normal((n, hw, hw, cin)), 0).astype(np.float32))
    scale = rng.uniform(0.5, 1.5, cin).astype(np.float32)
    beta = (0.1 * rng.standard_normal(cin)).astype(np.float32)
    gamma = rng.uniform(0.5, 1.5, cin).astype(np.float32)
    nb = ops.conv_dgrad_bn_relu_blocks(torch.bfloat16, n, hw, hw, cin, cout, k, k, s)
    out = torch.full((n * hw * hw, cin), 9.0, device=DEV, dtype=torch.bfloat16)
    part = torch.full((nb, 2, cin), 7.0, device=DEV)
    bf = lambda a: _t(a).to(torch.bfloat16)
    ops.conv_dgrad_bn_relu(bf(dc).view(-1, cout), cout, 0, bf(w).view(k * k, cin, cout), bf(y).view(-1, cin),
                           cin, 0, _t(scale), _t(beta), _t(gamma), out, part, n, hw, hw, cin, cout, k, k, s)
    inst = ops.last_dispatch()
    if check:
      dx, _ = ref_ops.conv2d_backward(np.zeros((n, hw, hw, cin)), w.astype(np.float64), dc.astype(np.float64), s)
      dz = dx * (y > 0)
      _scale_close(_n(out).reshape(n, hw, hw, cin), dz * scale, 1.1 * 2.0 ** -8, "%s fused dc %s" % (name, inst))
      sums = part.double().sum(0).cpu().numpy()
      for got, want in ((sums[0], dz.reshape(-1, cin).sum(0)),
                        (sums[1], (dz * (y.astype(np.float64) - beta) / gamma).reshape(-1, cin).sum(0))):
        assert np.abs(got - want).max() <= 1e-3 * max(np.abs(want).max(), 1.0), name
    return inst

  want_inst = run(N_BENCH, False)
  assert all(i.endswith(", true, 1>") for i in want_inst), want_inst
  for n in (256, 704, N_BENCH):
    if run(n, False) == want_inst:
      assert run(n, True) == want_inst
      break
  else:
    raise AssertionError((name, want_inst))
  _seen.update(want_inst)


@pytest.mark.parametrize("block", ["Mixed_5b", "Mixed_5c"])
def test_block_entry_dgrad_multi_fused_bf16(ops, block):
  """The block-boundary form in bf16 (round 5, c2d_conv1x1_dgrad_multi_bn_relu_bf16): the multi-segment
  entry gradient of Mixed_5b / 5c as last writer of the block-input gradient, applying the BN/ReLU
  backward of the last convolution of every branch of the block in front (Mixed_5a: 192 + 256
  convolution columns + 576 max-pool columns; Mixed_5b: 352 + 320 + 224 + 128), accumulating onto the
  pooling branch's share — at the benchmark's instance."""
  hw, cin = 4, 1024
  couts, widths, ident = {
      "Mixed_5b": ([352, 192, 160, 128], [192, 256, 576], [False, False, True]),
      "Mixed_5c": ([352, 192, 192], [352, 320, 224, 128], [False, False, False, False])}[block]

  def run(n, check):
    rng = np.random.default_rng(5 + n)
    rows = n * hw * hw
    dcs = [_bf16_round(rng.standard_normal((rows, c)).astype(np.float32)) for c in couts]
    ws = [_bf16_round((rng.standard_normal((cin, c)) / np.sqrt(c * len(couts))).astype(np.float32)) for c in couts]
    y = _bf16_round(np.maximum(rng.standard_normal((rows, cin)), 0).astype(np.float32))
    base = _bf16_round(rng.standard_normal((rows, cin)).astype(np.float32))
    prods, keep, off = [], [], 0
    want = want_sums = None
    if check:
      dx = sum(d.astype(np.float64) @ w.astype(np.float64).T for d, w in zip(dcs, ws)) + base
      want, want_sums = np.empty_like(dx), np.zeros((2, cin))
    for width, idn in zip(widths, ident):
      sl = slice(off, off + width)
      if idn:
        prods.append((None, None, None, width))
        if check:
          want[:, sl] = dx[:, sl]
      else:
        scale = rng.uniform(0.5, 1.5, width).astype(np.float32)
        beta = (0.1 * rng.standard_normal(width)).astype(np.float32)
        gamma = rng.uniform(0.5, 1.5, width).astype(np.float32)
        t = (_t(scale), _t(beta), _t(gamma))
        keep.append(t)
        prods.append(t + (width,))
        if check:
          dz = dx[:, sl] * (y[:, sl] > 0)
          want[:, sl] = dz * scale
          want_sums[0, sl] = dz.sum(0)
          want_sums[1, sl] = (dz * (y[:, sl].astype(np.float64) - beta) / gamma).sum(0)
      off += width
    nb = ops.conv1x1_dgrad_multi_bn_relu_blocks(couts, rows, cin, torch.bfloat16)
    bf = lambda a: _t(a).to(torch.bfloat16)
    out = bf(base).clone()
    part = torch.full((nb, 2, cin), 7.0, device=DEV)
    ops.conv1x1_dgrad_multi_bn_relu([bf(d) for d in dcs], couts, [0] * len(couts), [bf(w) for w in ws], couts,
                                    bf(y), cin, 0, ops.bn_producers(prods), out, cin, 0, part, rows, cin, True)
    inst = ops.last_dispatch()
    if check:
      _scale_close(_n(out), want, 1.1 * 2.0 ** -8, "%s fused entry dgrad %s" % (block, inst))
      sums = part.double().sum(0).cpu().numpy()
      for k_ in range(2):
        assert np.abs(sums[k_] - want_sums[k_]).max() <= 1e-3 * max(np.abs(want_sums[k_]).max(), 1.0)
    return inst

  want_inst = run(N_BENCH, False)
  assert all(i.endswith(", true, 1>") for i in want_inst), want_inst
  for n in (256, 704, N_BENCH):
    if run(n, False) == want_inst:
      assert run(n, True) == want_inst
      break
  else:
    raise AssertionError((block, want_inst))
  _seen.update(want_inst)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("block", ["Mixed_5a", "Mixed_5b", "Mixed_5c"])
def test_block_entry_wgrad_multi(ops, block, dtype):
  """The filter gradients of a block's 1x1 entry convolutions in ONE launch (c2d_conv1x1_wgrad_multi:
  the bf16 step's launch; shared row splits) at the real widths and 704 ROIs, the gradients at their
  places in the step: inside the block's own buffers or (352-wide branch 0) the 1024-wide concat
  gradient.  Against x^T . dc in float64."""
  hw, cin, couts = {"Mixed_5a": (7, 576, [128, 192]), "Mixed_5b": (4, 1024, [352, 192, 160]),
                    "Mixed_5c": (4, 1024, [352, 192, 192])}[block]
  low = dtype == torch.bfloat16
  rng = np.random.default_rng(37)
  rows = 704 * hw * hw
  x = rng.standard_normal((rows, cin)).astype(np.float32)
  wide = rng.standard_normal((rows, 1024)).astype(np.float32)       # branch 0 = columns 0..c0 of it
  dcs = [wide] + [rng.standard_normal((rows, c)).astype(np.float32) for c in couts[1:]]
  if low:
    x, dcs = _bf16_round(x), [_bf16_round(a) for a in dcs]
  dws = [torch.zeros(cin, c, device=DEV) for c in couts]
  ops.conv1x1_wgrad_multi(_t(x).to(dtype), cin, 0, [_t(a).to(dtype) for a in dcs],
                          [1024] + couts[1:], [0] * len(couts), dws, couts, rows, cin)
  inst = ops.last_dispatch()
  _seen.update(inst)
  for dw, dc, c in zip(dws, dcs, couts):
    want = x.astype(np.float64).T @ dc[:, :c].astype(np.float64)
    _scale_close(_n(dw), want, 2e-5, "%s entry wgrad %s" % (block, inst))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("block", ["Mixed_5a", "Mixed_5b", "Mixed_5c"])
def test_block_entry_fwd_multi(ops, block, dtype):
  """The 1x1 entry convolutions of an Inception block as ONE GEMM (c2d_conv1x1_fwd_multi) at the
  real widths — 5a: 576 -> (128, 192) on 7x7; 5b: 1024 -> (352, 192, 160, 128: the commuted
  pooling branch's convolution runs without ReLU); 5c: 1024 -> (352, 192, 192) — against the
  float64 oracle (tests/test_gpu_ops.py compares the fused launch with the unfused launches
  bitwise; here it meets `ref_ops.conv2d` + the folded BatchNorm affine + ReLU directly, on the
  instance the N = 2000 call dispatches)."""
  hw, cin, couts = {"Mixed_5a": (7, 576, [128, 192]), "Mixed_5b": (4, 1024, [352, 192, 160, 128]),
                    "Mixed_5c": (4, 1024, [352, 192, 192])}[block]
  low = dtype == torch.bfloat16
  rng = np.random.default_rng(37)

  def run(n):
    rows = n * hw * hw
    x = rng.standard_normal((rows, cin)).astype(np.float32)
    if low:
      x = _bf16_round(x)
    flat = (rng.standard_normal(sum(couts) * cin) / np.sqrt(cin)).astype(np.float32)
    if low:
      flat = _bf16_round(flat)
    flat_ = _t(flat).to(dtype)
    outs, params, off = [], [], 0
    for i, c in enumerate(couts):
      wt = flat_[off:off + c * cin].view(1, c, cin)
      w = flat[off:off + c * cin].reshape(c, cin)
      off += c * cin
      scale = rng.uniform(0.5, 1.5, c).astype(np.float32)
      shift = (0.1 * rng.standard_normal(c)).astype(np.float32)
      relu = not (block == "Mixed_5b" and i == 3)
      y = torch.empty(rows, c, device=DEV, dtype=dtype)
      outs.append((wt, _t(scale), _t(shift), y, c, 0, c, relu))
      params.append((w, scale, shift, relu))
    ops.conv1x1_fwd_multi(_t(x).to(dtype), cin, 0, ops.conv_outs(outs), rows, cin)
    return x, outs, params, ops.last_dispatch()

  _, _, _, want_inst = run(N_BENCH)
  for n in (256, 704, N_BENCH):
    x, outs, params, inst = run(n)
    if inst == want_inst:
      break
  assert inst == want_inst
  _seen.update(inst)
  x64 = x.astype(np.float64)
  for (w, scale, shift, relu), o in zip(params, outs):
    want = (x64 @ w.astype(np.float64).T) * scale + shift
    if relu:
      want = np.maximum(want, 0)
    _scale_close(_n(o[3]), want, 1.1 * 2.0 ** -8 if low else 2e-5, "%s entry fwd %s" % (block, inst))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_mixed_4e_entry_group(ops, dtype):
  """The three 1x1 entry convolutions of Mixed_4e (576 -> 96 / 128 / 160 on the 32x32 map) as ONE
  grouped launch (c2d_conv_fwd_grouped(_bf16) -> igemm_small_group_kernel), each against the
  oracle."""
  rng = np.random.default_rng(41)
  hw, cin = 32, 576
  low = dtype == torch.bfloat16
  x = rng.standard_normal((1, hw, hw, cin)).astype(np.float32)
  if low:
    x = _bf16_round(x)
  x_ = _t(x).view(hw * hw, cin).to(dtype)
  calls, outs, wants = [], [], []
  for cout in (96, 128, 160):
    w = (rng.standard_normal((1, 1, cin, cout)) / np.sqrt(cin)).astype(np.float32)
    if low:
      w = _bf16_round(w)
    wt = torch.empty(1, cout, cin, device=DEV)
    ops.transpose_taps(_t(w), wt, 1, cin, cout)
    y = torch.empty(hw * hw, cout, device=DEV, dtype=dtype)
    calls.append((x_, cin, 0, wt.to(dtype), None, None, y, cout, 0, 1, hw, hw, cin, cout, 1, 1, 1, False))
    outs.append(y)
    wants.append(ref_ops.conv2d(x.astype(np.float64), w.astype(np.float64), 1).reshape(hw * hw, cout))
  group = ops.conv_group(calls)
  ops.conv_fwd_grouped(group)
  inst = ops.last_dispatch()
  assert inst == ["igemm_small_group_kernel<0, %d>" % (2 if low else 4)], inst
  _seen.update(inst)
  for y, want in zip(outs, wants):
    _scale_close(_n(y), want, 1.1 * 2.0 ** -8 if low else 2e-5, "grouped 1x1")


@pytest.mark.parametrize("fuse_bn_bwd,commute", [("1", "1"), ("0", "1"), ("1", "0")],
                         ids=["fused_bn_bwd", "separate_bn_bwd", "fused_pool_not_commuted"])
def test_second_stage_fwd_bwd_dm1_n128(monkeypatch, fuse_bn_bwd, commute):
  """The whole second stage (Mixed_5a-c, depth multiplier 1.0) forward + backward on 128 ROIs
  through the engine's launch plan against ref_model.net_forward / net_backward in float64:
  output map, input gradient and every filter / BatchNorm gradient.  Both backward plans: the
  BN/ReLU backward of the inner convolutions fused into their consumers' input-gradient GEMMs
  (c2d_conv_dgrad_bn_relu, nine producer layers here; c2d_conv1x1_dgrad_multi_bn_relu at the two
  block boundaries, six more) and as separate launches."""
  # (Mixed_5b's average-pooling branch runs as 1x1 conv -> BN -> pool -> ReLU unless switched off)
  monkeypatch.setenv("C2D_TUNE", "fuse_bn_bwd=%s,commute_avgpool=%s" % (fuse_bn_bwd, commute))
  from cap2det_amd import hip_ops
  from cap2det_amd.models.frcnn_engine import SECOND_SCOPE, SECOND_STAGE, DerivedStore, Net, Ref, VariableStore
  n, hw, cin = 128, 7, 576
  rng = np.random.default_rng(3)
  store, stats = VariableStore(torch.device(DEV)), DerivedStore(torch.device(DEV))
  net = Net(store, stats, SECOND_STAGE, SECOND_SCOPE, cin, True, 1.0)
  store.finalize(); stats.finalize()
  P = {}
  for name, L in net.layers.items():
    L.trainable = True
    P[name + "/weights"] = (rng.standard_normal((L.k, L.k, L.cin, L.cout)) /
                            np.sqrt(L.k * L.k * L.cin / 2.0)).astype(np.float32)
    P[name + "/BatchNorm/gamma"] = rng.uniform(0.5, 1.5, L.cout).astype(np.float32)
    P[name + "/BatchNorm/beta"] = (0.1 * rng.standard_normal(L.cout)).astype(np.float32)
    P[name + "/BatchNorm/moving_mean"] = (0.1 * rng.standard_normal(L.cout)).astype(np.float32)
    P[name + "/BatchNorm/moving_variance"] = rng.uniform(0.5, 1.5, L.cout).astype(np.float32)
    for leaf in ("weights", "BatchNorm/gamma", "BatchNorm/beta"):
      store.var[name + "/" + leaf].copy_(_t(P[name + "/" + leaf]))
    for leaf in ("moving_mean", "moving_variance"):
      stats[name + "/BatchNorm/" + leaf].copy_(_t(P[name + "/BatchNorm/" + leaf]))
  net.refresh()
  x = np.maximum(rng.standard_normal((n, hw, hw, cin)), 0).astype(np.float32)
  plan = net.plan(n, hw, hw, True)
  xin = Ref(_t(x).view(n * hw * hw, cin), cin, 0, cin)
  out = net.forward(plan, xin)
  P64 = {k: v.astype(np.float64) for k, v in P.items()}
  want, tape = ref_model.net_forward(ref_model.SECOND_STAGE, x.astype(np.float64), P64,
                                     ref_model.SECOND_SCOPE)
  got = _n(out.t).reshape(want.shape)
  _scale_close(got, want, 2e-5, "second stage output")
  # The backward pass branches on data (ReLU masks y > 0, max-pool arg-max): an activation that
  # fp32 and float64 place on different sides of zero flips a whole gradient path — a property of
  # the precision, not of the kernels (a few of the 10^7 activations here do).  The oracle's
  # activations are therefore written into the engine's buffers (and its pools re-run on them) before the
  # backward pass, so that both sides differentiate the SAME piecewise-linear function and the
  # comparison stays tight.
  block_in = xin
  for st, saved in zip(plan["steps"], tape):
    assert st["kind"] == "block"
    for bsteps, btape in zip(st["branches"], saved[0]):
      if bsteps[0].get("commuted"):
        # engine: conv (BN, no ReLU) -> pool + ReLU; oracle: pool -> conv + BN + ReLU.  The
        # convolution re-runs on the oracle-valued block input, the oracle's branch output (the
        # ReLU whose mask the backward pass branches on) goes into the pool's output
        assert commute == "1" and [b["kind"] for b in bsteps] == ["conv", "pool"]
        net._fwd_step(bsteps[0], block_in)
        ref = bsteps[1]["y"]
        ref.t[:, ref.off:ref.off + ref.c].copy_(_t(btape[1][2].reshape(-1, ref.c).astype(np.float32)))
        continue
      for bst, sv in zip(bsteps, btape):
        ref = bst["y"]
        if bst["kind"] == "conv":
          ref.t[:, ref.off:ref.off + ref.c].copy_(_t(sv[2].reshape(-1, ref.c).astype(np.float32)))
        else:
          # pools open their branch: re-run them on the (now oracle-valued) block input so that
          # their outputs and arg-max indices follow the same values
          net._fwd_step(bst, block_in)
    block_in = st["y"]
  dy = rng.standard_normal(want.shape).astype(np.float32)
  gy = net.out_grad(plan, 0)
  gy.t.copy_(_t(dy).view(gy.t.shape))
  dx = Ref(torch.empty(n * hw * hw, cin, device=DEV), cin, 0, cin)
  store.grads.zero_()
  net.backward(plan, xin, 0, dx)
  torch.cuda.synchronize()
  ops_ = [op for st in plan["steps"] if st["kind"] == "block" for b in st["branches"] for op in b]
  inner = sum(1 for op in ops_ if "fused_blocks" in op and "fused_wide" not in op)
  boundary = sum(1 for op in ops_ if "fused_wide" in op)
  # nine inner producers; at the two block boundaries the last convolutions of Mixed_5a (2, its
  # third branch is the max pool) and of Mixed_5b (4)
  # (5 with the commuted branch: its last op is the pool, whose columns pass through)
  assert (inner, boundary) == ((9, 5 if commute == "1" else 6) if fuse_bn_bwd == "1" else (0, 0))
  assert sum(1 for op in ops_ if op.get("commuted")) == (1 if commute == "1" else 0)
  want_dx, grads = ref_model.net_backward(ref_model.SECOND_STAGE, tape, dy.astype(np.float64), P64,
                                          ref_model.SECOND_SCOPE, 0, True)
  _scale_close(_n(dx.t).reshape(want_dx.shape), want_dx, 5e-5, "second stage input gradient")
  assert len(grads) == 3 * len(net.layers)
  for name, g in grads.items():
    _scale_close(_n(store.grad[name]), g, 1e-4, "grad " + name)
  hip_ops.set_conv_workspace(None)


def test_every_benchmark_kernel_instance_was_compared_with_the_oracle():
  """Runs last in this file: the union of the instances the layer tests dispatched (and compared)
  covers every igemm / wgrad instance of the newest committed benchmark profiles (fp32 configs[1] and
  bf16 configs[2])."""
  for cfg in ("c1", "c2"):
    path, want = _latest_profile_instances(cfg)
    assert len(want) >= 12, path
    missing = want - _seen
    assert not missing, "%s: never dispatched by a parity test: %s" % (path, sorted(missing))
  bf16 = {k for k in _seen if "bf16" in k}
  assert any(k.startswith("wgrad3x3_bf16_kernel<4") for k in bf16), sorted(bf16)
  assert any(k.startswith("wgrad3x3_bf16_kernel<7") for k in bf16), sorted(bf16)
  assert any(k.startswith("wgrad_tn_bf16_kernel") for k in bf16), sorted(bf16)
  # the direct-to-LDS bf16 ring kernel in the block tiles the benchmark shapes dispatch (128x64,
  # 128x256, the full-width 128x192 / 128x320 / 128x384 of the 160..192-, 320- and 352-channel
  # layers), row-major and pixel-major, forward and input gradient, both ring forms:
  # igemm_ring_kernel<MODE, WM, WN, MT, NT, PM, BKT, D, ES = 2, FUSED>
  import re
  inst = [re.match(r"igemm_ring_kernel<(\d), (\d), (\d), (\d), (\d), (true|false), (\d+), (\d), 2, (?:true|false), 1>", k)
          for k in _seen]
  inst = [m.groups() for m in inst if m]
  for mode in ("0", "1"):
    for pm in ("true", "false"):
      assert any(g[0] == mode and g[5] == pm for g in inst), (mode, pm, sorted(bf16))
  for tile in (("2", "4", "2", "2"), ("4", "2", "1", "3"), ("4", "2", "1", "5"), ("4", "2", "1", "6")):
    assert any(g[1:5] == tile for g in inst), (tile, sorted(bf16))
  for ring in (("64", "2"), ("32", "3")):
    assert any(g[6:8] == ring for g in inst), (ring, sorted(bf16))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("n,hw,s", [(1024, 4, 1), (2000, 4, 1), (1024, 7, 1), (1024, 7, 2)])
def test_pixel_major_heavy_first_order(ops, n, hw, s, dtype):
  """Round 4: pixel-major blocks are ONE pixel of a 128-image group (ConvGeom::pm = 7) and, when
  the image groups split evenly over the 8 XCDs (1024 and 2000 -> 2048 images: 8 / 16 groups),
  their ids are dealt heavy pixels first (ConvGeom::lpt_ngx, block_tile()).  The oracle tests above
  pick 256 images where they can (2 groups: tile order); here the heavy-first mapping itself meets
  the float64 oracle — 4x4 and 7x7 maps, 2000 images (48 padding images in the last group: whole
  MFMA tiles without rows), forward and input gradient, and the stride-2 input gradient (not
  pixel-major: the row-major fallback of the same entry point)."""
  cin, cout = 32, 64
  lay = _Layer(ops, n, hw, cin, cout, 3, s, 31 + n + hw, dtype)
  low = dtype == torch.bfloat16
  x64, w64, dc64 = lay.x.astype(np.float64), lay.w.astype(np.float64), lay.dc.astype(np.float64)
  got = lay.run(ops, "fwd")
  inst = ops.last_dispatch()
  _scale_close(_n(got), ref_ops.conv2d(x64, w64, s), 1.1 * 2.0 ** -8 if low else 2e-5,
               "fwd n=%d hw=%d %s" % (n, hw, inst))
  if s == 1:
    assert any(", true," in i for i in inst), inst            # (the pixel-major instance)
  got = lay.run(ops, "dgrad")
  want, _ = ref_ops.conv2d_backward(x64, w64, dc64, s, need_dx=True)
  _scale_close(_n(got), want, 1.1 * 2.0 ** -8 if low else 2e-5,
               "dgrad n=%d hw=%d %s" % (n, hw, ops.last_dispatch()))
