"""GPU parity tests, kernel by kernel: the HIP path (through the C-ABI) against the numpy oracle
on identical seeded inputs.  Tolerances are written per test; index/byte outputs are exact."""
import os
import numpy as np
import pytest
import torch

from oracle import ref_labels, ref_model, ref_ops

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _t(a, dtype=None):
  return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).to(DEV).contiguous()


def _n(t):
  return t.detach().cpu().numpy()


@pytest.fixture(scope="module")
def ops():
  from cap2det_amd import hip_ops
  return hip_ops


def _edge_boxes(rng, n):
  c = rng.uniform(0, 1, (n, 2))
  s = np.exp(rng.uniform(np.log(0.04), 0.0, (n, 2)))
  b = np.concatenate([np.clip(c - s / 2, 0, 1), np.clip(c + s / 2, 0, 1)], 1).astype(np.float32)
  b[0] = [0, 0, 0, 0]               # zero-padded proposal (readers/cap2det_reader.py:237)
  b[1] = [0, 0, 1, 1]               # whole image, touches both borders
  b[2] = [0.5, 0.5, 0.5, 0.5]       # single point
  b[3] = [0.7, 0.2, 0.3, 0.9]       # y2 < y1
  b[4] = [0.25, 0.25, 1.0, 1.0]
  b[5] = [-0.1, -0.2, 0.5, 1.3]     # partly outside: extrapolation rows/cols
  return b


@pytest.mark.parametrize("hf,wf,d,n", [(9, 11, 16, 37), (32, 32, 576, 64)])
def test_crop_and_resize_matches_oracle(ops, hf, wf, d, n):
  rng = np.random.default_rng(1234)
  feat = rng.standard_normal((2, hf, wf, d)).astype(np.float32)
  boxes = _edge_boxes(rng, n)
  ind = rng.integers(0, 2, n).astype(np.int32)
  want = ref_ops.crop_and_resize(feat, boxes, ind, 14)
  got = _n(ops.crop_and_resize(_t(feat), _t(boxes), _t(ind), 14))
  # same fp32 operation order as the oracle: bit-exact
  np.testing.assert_array_equal(got, want)


def test_roi_crop_pool_fwd_bwd_matches_oracle(ops):
  rng = np.random.default_rng(7)
  hf, wf, d, n = 12, 10, 32, 41
  feat = np.maximum(rng.standard_normal((2, hf, wf, d)), 0).astype(np.float32)  # post-ReLU map
  boxes = _edge_boxes(rng, n)
  ind = rng.integers(0, 2, n).astype(np.int32)
  crop = ref_ops.crop_and_resize(feat, boxes, ind, 14)
  want, want_arg = ref_ops.max_pool(crop, 2, 2, "VALID")
  out, arg = ops.roi_crop_pool_fwd(_t(feat), _t(boxes), _t(ind), 14, 2, 2)
  np.testing.assert_array_equal(_n(out), want)
  np.testing.assert_array_equal(_n(arg), want_arg)
  dout = rng.standard_normal(want.shape).astype(np.float32)
  dcrop = ref_ops.max_pool_backward(crop.shape, want_arg, dout, 2, 2, "VALID")
  want_df = ref_ops.crop_and_resize_grad_image(dcrop.astype(np.float64), boxes, ind, feat.shape)
  dfeat = torch.zeros(feat.shape, device=DEV)
  ops.roi_crop_pool_bwd(_t(dout), arg, _t(boxes), _t(ind), dfeat, 14, 2, 2)
  # fp32 atomics in arbitrary order vs a float64 sum
  np.testing.assert_allclose(_n(dfeat), want_df, rtol=1e-4, atol=1e-4)


def test_roi_crop_pool_bwd_workspace_form_is_exact_and_deterministic(ops):
  rng = np.random.default_rng(8)
  hf, wf, d, n = 13, 9, 32, 70
  feat = np.maximum(rng.standard_normal((2, hf, wf, d)), 0).astype(np.float32)
  boxes = _edge_boxes(rng, n)
  ind = rng.integers(0, 2, n).astype(np.int32)
  crop = ref_ops.crop_and_resize(feat, boxes, ind, 14)
  pooled, arg = ref_ops.max_pool(crop, 2, 2, "VALID")
  dout = rng.standard_normal(pooled.shape).astype(np.float32)
  dcrop = ref_ops.max_pool_backward(crop.shape, arg, dout, 2, 2, "VALID")
  want = ref_ops.crop_and_resize_grad_image(dcrop.astype(np.float64), boxes, ind, feat.shape)
  ws = torch.empty(ops.roi_crop_pool_bwd_workspace_bytes(2, hf, wf, d, n, 14, 2, 2), dtype=torch.uint8,
                   device=DEV)
  outs = []
  for _ in range(2):
    dfeat = torch.full(feat.shape, 0.25, device=DEV)       # adds into the existing gradient
    ops.roi_crop_pool_bwd_ws(_t(dout), _t(arg), _t(boxes), _t(ind), dfeat, 14, 2, 2, ws)
    outs.append(_n(dfeat))
  np.testing.assert_allclose(outs[0] - 0.25, want, rtol=1e-4, atol=1e-5)
  np.testing.assert_array_equal(outs[0], outs[1])          # no atomics: bitwise reproducible
  small = torch.empty(16, dtype=torch.uint8, device=DEV)
  from cap2det_amd._lib import Cap2DetHipError
  with pytest.raises(Cap2DetHipError):
    ops.roi_crop_pool_bwd_ws(_t(dout), _t(arg), _t(boxes), _t(ind), dfeat, 14, 2, 2, small)



@pytest.mark.parametrize("hf,wf,d,n", [(32, 32, 576, 500), (13, 9, 32, 70), (63, 84, 64, 120)])
def test_roi_crop_pool_bwd_halves_equal_the_one_call_form(ops, hf, wf, d, n):
  """c2d_roi_crop_pool_bwd_prepare + _run (the halves the training step queues on two streams) are
  bitwise the one-call form, fp32 and bf16 pooled gradients, with rows no box touches (the boxes
  stay in the upper half of image 0 and image 1 gets none) left at their old value."""
  rng = np.random.default_rng(17 + n)
  boxes = _edge_boxes(rng, n)
  boxes[:, 0] *= 0.45; boxes[:, 2] *= 0.45           # ymin, ymax: upper half only
  ind = np.zeros(n, np.int32)
  p = 7
  arg = rng.integers(0, 4, (n, p, p, d)).astype(np.uint8)
  dout = rng.standard_normal((n, p, p, d)).astype(np.float32)
  ws = torch.empty(ops.roi_crop_pool_bwd_workspace_bytes(2, hf, wf, d, n, 14, 2, 2), dtype=torch.uint8,
                   device=DEV)
  tb, ti, ta = _t(boxes), _t(ind), _t(arg)
  for dt in (torch.float32, torch.bfloat16):
    g = _t(dout).to(dt)
    want = torch.full((2, hf, wf, d), 0.125, device=DEV)
    ops.roi_crop_pool_bwd_ws(g, ta, tb, ti, want, 14, 2, 2, ws)
    got = torch.full((2, hf, wf, d), 0.125, device=DEV)
    ops.roi_crop_pool_bwd_prepare(tb, ti, 2, hf, wf, d, 14, 2, 2, ws)
    ops.roi_crop_pool_bwd_run(g, ta, tb, ti, got, 14, 2, 2, ws)
    torch.cuda.synchronize()
    assert torch.equal(got, want)
    assert float((got[0, :hf // 3] - 0.125).abs().max()) > 0 and bool((got[1] == 0.125).all())


@pytest.mark.parametrize("kind,hf,wf,d,n", [("one", 32, 32, 64, 1), ("few", 32, 32, 64, 3),
                                            ("band", 20, 17, 48, 40), ("one_image", 16, 16, 32, 50),
                                            ("padded", 12, 12, 32, 9)])
def test_roi_crop_pool_bwd_equal_shares_plan_edge_cases(ops, kind, hf, wf, d, n):
  """The strip plan (roi_plan_strips_kernel: equal shares of ONE trip sequence over all row lists,
  slots in closed form) on box sets that leave rows without a list entry, fewer trips than
  workgroups, one image of the batch untouched, or nothing but zero-padded boxes
  (readers/cap2det_reader.py:237,252: every tap of a padded box is pixel (0, 0))."""
  rng = np.random.default_rng(len(kind) * 100 + n)
  feat = np.maximum(rng.standard_normal((2, hf, wf, d)), 0).astype(np.float32)
  if kind == "band":          # every box inside rows 5..8 of the map: most rows have no entry
    y0 = rng.uniform(5.0, 6.5, n) / (hf - 1)
    y1 = rng.uniform(6.5, 8.0, n) / (hf - 1)
    x0 = rng.uniform(0.0, 0.5, n)
    x1 = rng.uniform(0.5, 1.0, n)
    boxes = np.stack([y0, x0, y1, x1], 1).astype(np.float32)
  elif kind == "padded":
    boxes = np.zeros((n, 4), np.float32)
  else:
    c = rng.uniform(0.2, 0.8, (n, 2))
    sz = rng.uniform(0.05, 0.4, (n, 2))
    boxes = np.concatenate([c - sz / 2, c + sz / 2], 1).astype(np.float32)
  ind = (np.ones(n) if kind == "one_image" else rng.integers(0, 2, n)).astype(np.int32)
  crop = ref_ops.crop_and_resize(feat, boxes, ind, 14)
  pooled, arg = ref_ops.max_pool(crop, 2, 2, "VALID")
  dout = rng.standard_normal(pooled.shape).astype(np.float32)
  dcrop = ref_ops.max_pool_backward(crop.shape, arg, dout, 2, 2, "VALID")
  want = ref_ops.crop_and_resize_grad_image(dcrop.astype(np.float64), boxes, ind, feat.shape)
  ws = torch.empty(ops.roi_crop_pool_bwd_workspace_bytes(2, hf, wf, d, n, 14, 2, 2), dtype=torch.uint8,
                   device=DEV)
  outs = []
  for _ in range(2):
    dfeat = torch.full(feat.shape, -0.5, device=DEV)
    ops.roi_crop_pool_bwd_ws(_t(dout), _t(arg), _t(boxes), _t(ind), dfeat, 14, 2, 2, ws)
    outs.append(_n(dfeat))
  scale = max(np.abs(want).max(), 1.0)
  assert np.abs(outs[0] + 0.5 - want).max() <= 1e-5 * scale
  np.testing.assert_array_equal(outs[0], outs[1])
  if kind == "one_image":
    np.testing.assert_array_equal(outs[0][0], np.full(feat.shape[1:], -0.5, np.float32))


@pytest.mark.parametrize("hf,wf,d,n,chunk", [(63, 84, 192, 300, 192),    # 1000x1333 image: three column ranges of 28
                                             (75, 100, 576, 100, 192),   # 1200x1600: four ranges of 25, three chunks
                                             (75, 100, 80, 90, 64),      # ragged last chunk
                                             (9, 255, 64, 60, 64),       # widest supported map
                                             (32, 32, 576, 120, 192)])   # the benchmark's map
def test_roi_crop_pool_bwd_workspace_form_on_wide_maps(ops, hf, wf, d, n, chunk):
  """The reference trains on keep-aspect 1000-px images x {1.2, .8, .6, .4} (configs/
  voc07_groundtruth.pbtxt:9-23, readers/cap2det_reader.py:143-172): feature maps up to ~100
  columns wide.  The atomic-free row-owner backward walks a feature row as strips of at most 32
  columns (round 4; before: whole rows with narrower channel chunks)."""
  from cap2det_amd import synthetic
  assert ops.roi_crop_pool_bwd_ws_supported(wf, d, 14, 2, 2) == chunk
  assert ops.roi_crop_pool_bwd_ws_supported(256, d, 14, 2, 2) == 0
  rng = np.random.default_rng(hf * 1000 + wf)
  feat = np.maximum(rng.standard_normal((2, hf, wf, d)), 0).astype(np.float32)
  boxes = np.concatenate([_edge_boxes(rng, n // 3), synthetic.synthetic_boxes(rng, n - n // 3)]
                         ).astype(np.float32)
  ind = rng.integers(0, 2, n).astype(np.int32)
  crop = ref_ops.crop_and_resize(feat, boxes, ind, 14)
  pooled, arg = ref_ops.max_pool(crop, 2, 2, "VALID")
  dout = rng.standard_normal(pooled.shape).astype(np.float32)
  dcrop = ref_ops.max_pool_backward(crop.shape, arg, dout, 2, 2, "VALID")
  want = ref_ops.crop_and_resize_grad_image(dcrop.astype(np.float64), boxes, ind, feat.shape)
  ws = torch.empty(ops.roi_crop_pool_bwd_workspace_bytes(2, hf, wf, d, n, 14, 2, 2), dtype=torch.uint8,
                   device=DEV)
  outs = []
  for _ in range(2):
    dfeat = torch.zeros(feat.shape, device=DEV)
    ops.roi_crop_pool_bwd_ws(_t(dout), _t(arg), _t(boxes), _t(ind), dfeat, 14, 2, 2, ws)
    outs.append(_n(dfeat))
  scale = np.abs(want).max()
  assert np.abs(outs[0] - want).max() <= 1e-5 * scale + 1e-5
  np.testing.assert_array_equal(outs[0], outs[1])
  # the atomic kernel (what maps wider than 255 columns fall back to) on the same inputs
  dfeat = torch.zeros(feat.shape, device=DEV)
  ops.roi_crop_pool_bwd(_t(dout), _t(arg), _t(boxes), _t(ind), dfeat, 14, 2, 2)
  assert np.abs(_n(dfeat) - want).max() <= 2e-5 * scale + 1e-5
  if n <= 120:
    # bf16 gradients (the bf16 storage mode, BASELINE configs[2] / [4]): same strips, fp32 sums;
    # the oracle runs on the bf16-rounded gradient
    dout_b = _t(dout).to(torch.bfloat16)
    dcrop_b = ref_ops.max_pool_backward(crop.shape, arg, dout_b.float().cpu().numpy(), 2, 2, "VALID")
    want_b = ref_ops.crop_and_resize_grad_image(dcrop_b.astype(np.float64), boxes, ind, feat.shape)
    dfeat = torch.zeros(feat.shape, device=DEV)
    ops.roi_crop_pool_bwd_ws(dout_b, _t(arg), _t(boxes), _t(ind), dfeat, 14, 2, 2, ws)
    assert np.abs(_n(dfeat) - want_b).max() <= 1e-5 * np.abs(want_b).max() + 1e-5
    with pytest.raises(NotImplementedError):
      ops.roi_crop_pool_bwd(dout_b, _t(arg), _t(boxes), _t(ind), dfeat, 14, 2, 2)


CONV_CASES = [
    # n, ih, iw, cin, cout, k, stride
    (3, 7, 7, 32, 64, 1, 1),
    (5, 7, 7, 48, 96, 3, 1),
    (5, 7, 7, 32, 64, 3, 2),
    (2, 4, 4, 64, 352, 3, 1),
    (1, 13, 9, 16, 32, 3, 2),
    (300, 7, 7, 64, 160, 3, 1),   # M large enough for the 128x128 tile path; pixel-major rows
    (70, 4, 4, 48, 96, 3, 1),     # pixel-major rows (n >= 64), ragged last group of 32 images
    (100, 7, 7, 32, 64, 3, 2),    # pixel-major stride-2 forward + parity-class dgrad
    (65, 4, 4, 32, 192, 3, 1),    # pixel-major, 128x64 tile variant (N % 128 == 64)
    (700, 4, 4, 96, 224, 3, 1),   # enough tiles for stream-K shares that cut tiles (K = 9 x 3 slabs)
    (900, 7, 7, 64, 96, 1, 1),    # 1x1, row-major stream-K
    (261, 7, 7, 32, 160, 3, 2),   # stride-2 nine-tap filter gradient (n >= 256), odd image count
    (257, 7, 7, 32, 96, 3, 1),    # nine-tap filter gradient on 7x7 maps, two-image slabs, odd image count
    (259, 4, 4, 32, 64, 3, 1),    # ... on 4x4 maps, odd image count
]


def _conv_inputs(rng, n, ih, iw, cin, cout, k):
  x = rng.standard_normal((n, ih, iw, cin)).astype(np.float32)
  w = (rng.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin)).astype(np.float32)
  return x, w


@pytest.fixture(params=["tile_per_block", "stream_k"])
def conv_mode(request, ops):
  """Runs a test with the plain launches and with the balanced (stream-K) workspace forms."""
  if request.param == "stream_k":
    ws = torch.zeros(ops.conv_workspace_bytes(), dtype=torch.uint8, device=DEV)
    ops.set_conv_workspace(ws)
    yield ws
    ops.set_conv_workspace(None)
    torch.cuda.synchronize()
    # the kernels must leave every tile counter zero for the next launch
    part = 1024 * 2 * 128 * 128 * 4
    assert int(ws[part:].to(torch.int32).abs().sum()) == 0 or int(ws[part:].max()) == 0
  else:
    ops.set_conv_workspace(None)
    yield None


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_bn_relu(ops, case, conv_mode):
  n, ih, iw, cin, cout, k, s = case
  rng = np.random.default_rng(11)
  x, w = _conv_inputs(rng, n, ih, iw, cin, cout, k)
  gamma = rng.uniform(0.5, 1.5, cout).astype(np.float32)
  beta = (0.1 * rng.standard_normal(cout)).astype(np.float32)
  mean = (0.1 * rng.standard_normal(cout)).astype(np.float32)
  var = rng.uniform(0.5, 1.5, cout).astype(np.float32)
  c = ref_ops.conv2d(x.astype(np.float64), w.astype(np.float64), s)
  want = np.maximum(ref_ops.batch_norm_inference(c, gamma.astype(np.float64), beta, mean,
                                                 var.astype(np.float64)), 0)
  oh, ow = want.shape[1:3]
  # input lives in a wider concat buffer (channel slice), output too
  ldx, xoff, ldy, yoff = cin + 16, 8, cout + 32, 16
  xb = np.zeros((n, ih, iw, ldx), np.float32); xb[..., xoff:xoff + cin] = x
  yb = torch.full((n, oh, ow, ldy), -7.0, device=DEV)
  wt = torch.empty(k * k, cout, cin, device=DEV)
  ops.transpose_taps(_t(w), wt, k * k, cin, cout)
  scale = torch.empty(cout, device=DEV); shift = torch.empty(cout, device=DEV)
  ops.bn_fold(_t(gamma), _t(beta), _t(mean), _t(var), 0.001, scale, shift)
  ops.conv_fwd(_t(xb), ldx, xoff, wt, scale, shift, yb, ldy, yoff, n, ih, iw, cin, cout, k, k, s, 1)
  got = _n(yb)
  np.testing.assert_allclose(got[..., yoff:yoff + cout], want, rtol=2e-5, atol=2e-5)
  assert np.all(got[..., :yoff] == -7.0) and np.all(got[..., yoff + cout:] == -7.0)


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_dgrad_wgrad(ops, case, conv_mode):
  n, ih, iw, cin, cout, k, s = case
  rng = np.random.default_rng(13)
  x, w = _conv_inputs(rng, n, ih, iw, cin, cout, k)
  oh, ow = -(-ih // s), -(-iw // s)
  dc = rng.standard_normal((n, oh, ow, cout)).astype(np.float32)
  want_dx, want_dw = ref_ops.conv2d_backward(x.astype(np.float64), w.astype(np.float64),
                                             dc.astype(np.float64), s)
  dx = torch.full((n, ih, iw, cin), 0.5, device=DEV)
  ops.conv_dgrad(_t(dc), cout, 0, _t(w), dx, cin, 0, n, ih, iw, cin, cout, k, k, s, 1)
  np.testing.assert_allclose(_n(dx) - 0.5, want_dx, rtol=1e-4, atol=1e-4)
  dw = torch.zeros(k, k, cin, cout, device=DEV)
  ops.conv_wgrad(_t(x), cin, 0, _t(dc), cout, 0, dw, n, ih, iw, cin, cout, k, k, s)
  scale = np.abs(want_dw).max()
  np.testing.assert_allclose(_n(dw), want_dw, rtol=1e-4, atol=1e-5 * scale + 1e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("rows,cin,couts", [(20000, 64, (32, 48)), (17001, 96, (40, 64, 104)),
                                            (32000, 1024, (352, 192, 160, 128)), (19000, 576, (128, 192))])
def test_conv1x1_fwd_multi_equals_separate_convs(ops, rows, cin, couts, dtype):
  """Several 1x1 convolutions of one input as one GEMM: bitwise equal to one c2d_conv_fwd per
  convolution (same K-ordered sums; more than 16384 rows, so that the separate calls do not take
  the split-K kernel of the single-image first stage), each output through its own scale / shift /
  ReLU flag into its own buffer slice; the second convolution has no ReLU."""
  rng = np.random.default_rng(43)
  x = _t(rng.standard_normal((rows, cin + 16)).astype(np.float32)).to(dtype)
  flat = _t((rng.standard_normal(sum(couts) * cin + 64) / np.sqrt(cin)).astype(np.float32)).to(dtype)
  outs, wants, keep, off = [], [], [], 16            # (weights of one flat buffer, as in the engine)
  for i, c in enumerate(couts):
    wt = flat[off:off + c * cin].view(1, c, cin); off += c * cin
    scale = _t(rng.uniform(0.5, 1.5, c).astype(np.float32))
    shift = _t((0.1 * rng.standard_normal(c)).astype(np.float32))
    relu = i != 1
    want = torch.full((rows, c + 16), -3.0, device=DEV, dtype=dtype)
    ops.conv_fwd(x, cin + 16, 8, wt, scale, shift, want, c + 16, 8, rows, 1, 1, cin, c, 1, 1, 1, relu)
    got = torch.full((rows, c + 16), -3.0, device=DEV, dtype=dtype)
    outs.append((wt, scale, shift, got, c + 16, 8, c, relu)); wants.append(want)
  ops.conv1x1_fwd_multi(x, cin + 16, 8, ops.conv_outs(outs), rows, cin)
  torch.cuda.synchronize()
  for o, want in zip(outs, wants):
    assert torch.equal(o[3], want)
  assert float(wants[1].float().min()) < 0.0          # (the ReLU-free output really is one)


FUSED_CASES = [
    (3, 7, 7, 32, 64, 1, 1),      # few rows: 64x64 tiles
    (70, 4, 4, 48, 96, 3, 1),     # pixel-major rows, ragged last group of 32 images
    (900, 7, 7, 64, 96, 1, 1),    # row-major, many row blocks
    (100, 7, 7, 32, 64, 3, 2),    # stride 2: four parity-class launches share the partials
    (261, 7, 7, 32, 160, 3, 2),   # odd image count
    (300, 7, 7, 160, 64, 3, 1),   # N = cin = 160
    (70, 4, 4, 36, 48, 3, 1),     # N = 36: a ragged last 32-column tile, pixel-major
    (300, 7, 7, 100, 64, 1, 1),   # N = 100, row-major
]


def _fused_reference(rng, n, ih, iw, cin, cout, k, s, with_gamma, quantize=None):
  """Inputs + float64 expectations of c2d_conv_dgrad_bn_relu: dx of the convolution, then the
  producer's ReLU mask / BN scale and the beta / gamma column sums."""
  q = quantize or (lambda a: a.astype(np.float32))
  x, w = _conv_inputs(rng, n, ih, iw, cin, cout, k)
  w = q(w)
  oh, ow = -(-ih // s), -(-iw // s)
  dc = q(rng.standard_normal((n, oh, ow, cout)).astype(np.float32))
  y = q(np.maximum(rng.standard_normal((n, ih, iw, cin)), 0).astype(np.float32))   # ties at 0
  scale = rng.uniform(0.5, 1.5, cin).astype(np.float32)
  beta = (0.1 * rng.standard_normal(cin)).astype(np.float32)
  gamma = rng.uniform(0.5, 1.5, cin).astype(np.float32) if with_gamma else None
  dx, _ = ref_ops.conv2d_backward(x.astype(np.float64), w.astype(np.float64), dc.astype(np.float64), s)
  dz = dx * (y > 0)
  want_dc = dz * scale
  want_db = dz.reshape(-1, cin).sum(0)
  want_dg = ((dz * (y.astype(np.float64) - beta) / gamma).reshape(-1, cin).sum(0)
             if with_gamma else np.zeros(cin))
  return w, dc, y, scale, beta, gamma, want_dc, want_db, want_dg


@pytest.mark.parametrize("with_gamma", [True, False], ids=["gamma", "nogamma"])
@pytest.mark.parametrize("case", FUSED_CASES)
def test_conv_dgrad_bn_relu(ops, case, with_gamma):
  """Input gradient fused with the producer layer's BN/ReLU backward against conv2d_backward +
  the bn_relu_bwd formulas of the oracle; the partial sums are reproducible (no atomics)."""
  n, ih, iw, cin, cout, k, s = case
  rng = np.random.default_rng(29)
  w, dc, y, scale, beta, gamma, want_dc, want_db, want_dg = _fused_reference(
      rng, n, ih, iw, cin, cout, k, s, with_gamma)
  nb = ops.conv_dgrad_bn_relu_blocks(torch.float32, n, ih, iw, cin, cout, k, k, s)
  assert nb >= 1
  # y lives in a wider concat-style buffer
  ldy, yoff = cin + 8, 4
  yb = np.zeros((n, ih, iw, ldy), np.float32); yb[..., yoff:yoff + cin] = y
  outs = []
  for _ in range(2):
    out = torch.full((n * ih * iw, cin), 9.0, device=DEV)
    part = torch.full((nb, 2, cin), 7.0, device=DEV)
    ops.conv_dgrad_bn_relu(_t(dc), cout, 0, _t(w), _t(yb), ldy, yoff, _t(scale), _t(beta),
                           _t(gamma) if with_gamma else None, out, part, n, ih, iw, cin, cout, k, k, s)
    outs.append((_n(out), _n(part)))
  got_dc, got_part = outs[0]
  sc = np.abs(want_dc).max()
  np.testing.assert_allclose(got_dc.reshape(want_dc.shape), want_dc, rtol=1e-4, atol=1e-5 * sc + 1e-6)
  sums = got_part.astype(np.float64).sum(0)
  np.testing.assert_allclose(sums[0], want_db, rtol=1e-4, atol=1e-4 * np.abs(want_db).max() + 1e-5)
  np.testing.assert_allclose(sums[1], want_dg, rtol=1e-4, atol=1e-4 * max(np.abs(want_dg).max(), 1.0))
  assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("accumulate", [False, True], ids=["overwrite", "accumulate"])
@pytest.mark.parametrize("rows,cin,couts,widths", [
    (2000, 64, (32, 48), (24, 40)),                    # two producers
    (3333, 256, (96, 64, 32), (96, 32, 128)),          # the middle one a pooling branch
    (5000, 1024, (352, 192, 160), (352, 320, 224, 128)),   # Mixed_5b -> Mixed_5c widths
])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_conv1x1_dgrad_multi_bn_relu(ops, rows, cin, couts, widths, accumulate, dtype):
  """Summed 1x1 input gradient of a block as the last writer of the block-input gradient, with
  the BN/ReLU backward of the producers of the block input per column range; the second of three
  producers is a pooling branch (plain gradient, zero sums).  bf16 storage (round 5: the DMA-ring
  kernel's fused epilogue instance): operands rounded to bf16 first, dc within one bf16 rounding of
  the float64 result on those operands, the fp32 column sums to 1e-3 of their scale."""
  rng = np.random.default_rng(31)
  assert sum(widths) == cin
  low = dtype == torch.bfloat16
  rnd = (lambda a: torch.from_numpy(a).to(torch.bfloat16).float().numpy()) if low else (lambda a: a)
  dcs = [rnd(rng.standard_normal((rows, c)).astype(np.float32)) for c in couts]
  wsn = [rnd((rng.standard_normal((cin, c)) / np.sqrt(c)).astype(np.float32)) for c in couts]
  y = rnd(np.maximum(rng.standard_normal((rows, cin)), 0).astype(np.float32))
  base = rnd(rng.standard_normal((rows, cin)).astype(np.float32))
  dx = sum(d.astype(np.float64) @ w.astype(np.float64).T for d, w in zip(dcs, wsn))
  if accumulate:
    dx = dx + base
  identity = [len(widths) == 3 and i == 1 for i in range(len(widths))]
  want, prods, keep, off = np.empty_like(dx), [], [], 0
  want_sums = np.zeros((2, cin))
  for width, ident in zip(widths, identity):
    sl = slice(off, off + width)
    if ident:
      want[:, sl] = dx[:, sl]
      prods.append((None, None, None, width))
    else:
      scale = rng.uniform(0.5, 1.5, width).astype(np.float32)
      beta = (0.1 * rng.standard_normal(width)).astype(np.float32)
      gamma = rng.uniform(0.5, 1.5, width).astype(np.float32)
      dz = dx[:, sl] * (y[:, sl] > 0)
      want[:, sl] = dz * scale
      want_sums[0, sl] = dz.sum(0)
      want_sums[1, sl] = (dz * (y[:, sl].astype(np.float64) - beta) / gamma).sum(0)
      t = (_t(scale), _t(beta), _t(gamma))
      keep.append(t)
      prods.append(t + (width,))
    off += width
  nb = ops.conv1x1_dgrad_multi_bn_relu_blocks(list(couts), rows, cin, dtype)
  assert nb >= 1
  out = (_t(base).clone() if accumulate else torch.full((rows, cin), 5.0, device=DEV)).to(dtype)
  part = torch.full((nb, 2, cin), 7.0, device=DEV)
  wts = [_t(w).to(dtype) for w in wsn]             # [cin][cout], as c2d_conv1x1_dgrad_multi
  ops.conv1x1_dgrad_multi_bn_relu([_t(d).to(dtype) for d in dcs], list(couts), [0] * len(couts), wts,
                                  list(couts), _t(y).to(dtype), cin, 0, ops.bn_producers(prods), out,
                                  cin, 0, part, rows, cin, accumulate)
  if low:
    assert ops.last_dispatch()[0].endswith(", 2, true, 1>"), ops.last_dispatch()    # the fused ring instance
  sc = np.abs(want).max()
  if low:
    assert np.abs(_n(out.float()) - want).max() <= 1.1 * 2.0 ** -8 * sc
  else:
    np.testing.assert_allclose(_n(out), want, rtol=1e-4, atol=1e-5 * sc + 1e-6)
  sums = _n(part).astype(np.float64).sum(0)
  for k in range(2):
    np.testing.assert_allclose(sums[k], want_sums[k], rtol=1e-3 if low else 1e-4,
                               atol=(1e-3 if low else 1e-4) * max(np.abs(want_sums[k]).max(), 1.0))


@pytest.mark.parametrize("mode,stride", [(0, 1), (0, 2), (1, 1)])
@pytest.mark.parametrize("ih,iw,n", [(7, 7, 6), (4, 4, 6), (9, 5, 6), (7, 7, 70), (4, 4, 70)])
def test_pool3x3(ops, mode, stride, ih, iw, n):
  """n = 70 takes the whole-map kernels of the per-ROI second-stage maps (n >= 64)."""
  rng = np.random.default_rng(17)
  c = 24
  x = np.maximum(rng.standard_normal((n, ih, iw, c)), 0).astype(np.float32)  # ties at 0
  if mode == 0:
    want, want_arg = ref_ops.max_pool(x, 3, stride, "SAME")
  else:
    want, want_arg = ref_ops.avg_pool_same(x, 3), None
  oh, ow = want.shape[1:3]
  y = torch.empty(n, oh, ow, c, device=DEV)
  arg = torch.empty(n, oh, ow, c, dtype=torch.uint8, device=DEV)
  ops.pool3x3_fwd(_t(x), c, 0, y, c, 0, arg, n, ih, iw, c, stride, mode)
  np.testing.assert_allclose(_n(y), want, rtol=1e-6, atol=1e-6)
  dy = rng.standard_normal(want.shape).astype(np.float32)
  if mode == 0:
    # the oracle's argmax numbering is over the padded window; map both to input coordinates
    want_dx = ref_ops.max_pool_backward(x.shape, want_arg, dy, 3, stride, "SAME")
  else:
    want_dx = ref_ops.avg_pool_same_backward(x.shape, dy, 3)
  dx = torch.zeros(n, ih, iw, c, device=DEV)
  ops.pool3x3_bwd(_t(dy), c, 0, arg, dx, c, 0, n, ih, iw, c, stride, mode, 0)
  np.testing.assert_allclose(_n(dx), want_dx, rtol=1e-5, atol=1e-6)
  # channel slices of wider buffers (concat layout) + accumulate into an existing gradient
  xw = torch.zeros(n, ih, iw, c + 8, device=DEV); xw[..., 4:4 + c] = _t(x)
  yw = torch.full((n, oh, ow, c + 12), 7.0, device=DEV)
  ops.pool3x3_fwd(xw, c + 8, 4, yw, c + 12, 8, arg, n, ih, iw, c, stride, mode)
  np.testing.assert_allclose(_n(yw[..., 8:8 + c]), want, rtol=1e-6, atol=1e-6)
  assert float(yw[..., :8].min()) == 7.0 and float(yw[..., 8 + c:].min()) == 7.0
  dxw = torch.ones(n, ih, iw, c + 8, device=DEV)
  dyw = torch.zeros(n, oh, ow, c + 12, device=DEV); dyw[..., 8:8 + c] = _t(dy)
  ops.pool3x3_bwd(dyw, c + 12, 8, arg, dxw, c + 8, 4, n, ih, iw, c, stride, mode, 1)
  np.testing.assert_allclose(_n(dxw[..., 4:4 + c]), want_dx + 1.0, rtol=1e-5, atol=1e-5)
  assert float(dxw[..., :4].max()) == 1.0 and float(dxw[..., 4 + c:].min()) == 1.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("ih,iw,n,accumulate", [(4, 4, 70, False), (4, 4, 130, True), (7, 7, 6, True),
                                                (9, 5, 3, False)])
def test_avgpool3x3_relu_and_bn_bwd_without_relu(ops, ih, iw, n, accumulate, dtype):
  """The three pieces of an average-pooling branch commuted behind its 1x1 convolution:
  y = relu(avg_pool(z)), dz (+)= avg_pool_bwd(dy * (y > 0)), and the BatchNorm backward of a
  layer without a ReLU (dc = dz * scale, sums of dz and dz * (z - beta) / gamma)."""
  rng = np.random.default_rng(41)
  c = 24
  zt = _t(rng.standard_normal((n, ih, iw, c)).astype(np.float32)).to(dtype)
  z = zt.float().cpu().numpy()
  want_y = np.maximum(ref_ops.avg_pool_same(z.astype(np.float64), 3), 0)
  y = torch.full((n, ih, iw, c + 8), 7.0, device=DEV, dtype=dtype)
  ops.avgpool3x3_relu_fwd(zt, c, 0, y, c + 8, 4, n, ih, iw, c, 1)
  tol = 1e-6 if dtype == torch.float32 else 2.0 ** -8
  np.testing.assert_allclose(y[..., 4:4 + c].float().cpu().numpy(), want_y, rtol=tol, atol=tol)
  assert float(y[..., :4].float().min()) == 7.0 and float(y[..., 4 + c:].float().min()) == 7.0
  dyt = _t(rng.standard_normal((n, ih, iw, c)).astype(np.float32)).to(dtype)
  dy = dyt.float().cpu().numpy().astype(np.float64)
  ymask = y[..., 4:4 + c].float().cpu().numpy() > 0
  want_dz = ref_ops.avg_pool_same_backward(z.shape, dy * ymask, 3)
  base = _t(rng.standard_normal((n, ih, iw, c)).astype(np.float32)).to(dtype)
  dz = base.clone()
  ops.avgpool3x3_relu_bwd(dyt, c, 0, y, c + 8, 4, dz, c, 0, n, ih, iw, c, 1, accumulate)
  want = want_dz + (base.float().cpu().numpy() if accumulate else 0.0)
  np.testing.assert_allclose(dz.float().cpu().numpy(), want, rtol=4 * tol, atol=4 * tol)
  # BatchNorm backward without a ReLU on z (rows = n * ih * iw)
  rows = n * ih * iw
  scale = rng.uniform(0.5, 1.5, c).astype(np.float32)
  beta = (0.1 * rng.standard_normal(c)).astype(np.float32)
  gamma = rng.uniform(0.5, 1.5, c).astype(np.float32)
  g = dz.float().cpu().numpy().reshape(rows, c).astype(np.float64)
  nb = ops.bn_relu_bwd_partial_blocks(rows, c)
  dc = torch.empty(rows, c, device=DEV, dtype=dtype)
  part = torch.full((nb, 2, c), 7.0, device=DEV)
  ops.bn_bwd_partial(dz, c, 0, zt, c, 0, _t(scale), _t(beta), _t(gamma), dc, part, rows, c)
  np.testing.assert_allclose(dc.float().cpu().numpy(), g * scale, rtol=tol, atol=tol * np.abs(g).max())
  sums = part.double().sum(0).cpu().numpy()
  np.testing.assert_allclose(sums[0], g.sum(0), rtol=1e-4, atol=1e-4 * np.abs(g.sum(0)).max() + 1e-5)
  want_g = (g * (z.reshape(rows, c).astype(np.float64) - beta) / gamma).sum(0)
  np.testing.assert_allclose(sums[1], want_g, rtol=1e-4, atol=1e-4 * max(np.abs(want_g).max(), 1.0))


def test_bn_relu_bwd(ops):
  rng = np.random.default_rng(19)
  rows, c = 1000, 96
  y = np.maximum(rng.standard_normal((rows, c)), 0).astype(np.float32)
  dy = rng.standard_normal((rows, c)).astype(np.float32)
  gamma = rng.uniform(0.5, 1.5, c).astype(np.float32)
  beta = (0.1 * rng.standard_normal(c)).astype(np.float32)
  scale = rng.uniform(0.5, 1.5, c).astype(np.float32)
  dz = dy * (y > 0)
  dc = torch.empty(rows, c, device=DEV)
  dbeta = torch.zeros(c, device=DEV); dgamma = torch.zeros(c, device=DEV)
  ops.bn_relu_bwd(_t(dy), c, 0, _t(y), c, 0, _t(scale), _t(beta), _t(gamma), dc, dbeta, dgamma,
                  rows, c)
  np.testing.assert_allclose(_n(dc), dz * scale, rtol=1e-6, atol=1e-6)
  np.testing.assert_allclose(_n(dbeta), dz.astype(np.float64).sum(0), rtol=1e-4, atol=1e-3)
  np.testing.assert_allclose(_n(dgamma),
                             (dz.astype(np.float64) * (y - beta) / gamma).sum(0),
                             rtol=1e-4, atol=1e-3)



@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("rois,spatial,c,coff,ctot,with_mask", [
    (70, 16, 96, 32, 160, True), (300, 16, 352, 0, 1024, True), (129, 4, 64, 64, 128, False)])
def test_bn_relu_bwd_partial_head(ops, rois, spatial, c, coff, ctot, with_mask, dtype):
  """BN/ReLU backward of an output convolution with the backward of the spatial mean + dropout
  folded in (dy is never stored) against the two formulas applied one after the other."""
  rng = np.random.default_rng(37)
  rows = rois * spatial
  dmean = rng.standard_normal((rois, ctot + 8)).astype(np.float32)      # columns [4, 4 + ctot)
  mask = (rng.uniform(size=(rois, ctot)) < 0.5).astype(np.uint8) if with_mask else None
  keep = 0.5 if with_mask else 1.0
  yt = _t(np.maximum(rng.standard_normal((rows, ctot)), 0).astype(np.float32)).to(dtype)
  y = yt.float().cpu().numpy()[:, coff:coff + c]
  scale = rng.uniform(0.5, 1.5, c).astype(np.float32)
  beta = (0.1 * rng.standard_normal(c)).astype(np.float32)
  gamma = rng.uniform(0.5, 1.5, c).astype(np.float32)
  g = dmean[:, 4 + coff:4 + coff + c].astype(np.float64)
  if with_mask:
    g = g * mask[:, coff:coff + c] / keep
  dy = np.repeat(g / spatial, spatial, axis=0)
  dz = dy * (y > 0)
  nb = ops.bn_relu_bwd_partial_blocks(rows, c)
  dc = torch.full((rows, c), 3.0, device=DEV, dtype=dtype)
  part = torch.full((nb, 2, c), 7.0, device=DEV)
  ops.bn_relu_bwd_partial_head(_t(dmean), ctot + 8, 4 + coff, _t(mask) if with_mask else None, ctot,
                               coff, spatial, keep, yt, ctot, coff, _t(scale), _t(beta), _t(gamma),
                               dc, part, rows, c)
  tol = 1e-6 if dtype == torch.float32 else 2.0 ** -8
  np.testing.assert_allclose(dc.float().cpu().numpy(), dz * scale, rtol=tol, atol=tol * np.abs(dz).max())
  sums = part.double().sum(0).cpu().numpy()
  np.testing.assert_allclose(sums[0], dz.sum(0), rtol=1e-4, atol=1e-4 * np.abs(dz.sum(0)).max())
  want_g = (dz * (y.astype(np.float64) - beta) / gamma).sum(0)
  np.testing.assert_allclose(sums[1], want_g, rtol=1e-4, atol=1e-4 * max(np.abs(want_g).max(), 1.0))


@pytest.mark.parametrize("rows,c,with_gamma", [(1000, 96, True), (70001, 352, True), (3136, 128, False)])
def test_bn_relu_bwd_partial_form_is_reproducible(ops, rows, c, with_gamma):
  """Atomic-free form used by the training step: per-row-block partial sums + ONE batched
  reduce into the flat gradient buffer (two layers share the launch here)."""
  rng = np.random.default_rng(23)
  y = np.maximum(rng.standard_normal((rows, c)), 0).astype(np.float32)
  dy = rng.standard_normal((rows, c)).astype(np.float32)
  gamma = rng.uniform(0.5, 1.5, c).astype(np.float32)
  beta = (0.1 * rng.standard_normal(c)).astype(np.float32)
  scale = rng.uniform(0.5, 1.5, c).astype(np.float32)
  dz = (dy * (y > 0)).astype(np.float64)
  nb = ops.bn_relu_bwd_partial_blocks(rows, c)
  assert nb >= 1 and (rows < 32 * 1024 or nb >= 512)
  ddt = np.dtype([("ws", "<i8"), ("dbeta", "<i8"), ("dgamma", "<i8"), ("nblocks", "<i4"),
                  ("c", "<i4"), ("begin", "<i4"), ("pad", "<i4")])
  chunks = -(-c // 64)
  # layer 0 and layer 1 are the same tensors: grads = [dbeta0 | dgamma0 | dbeta1 | dgamma1]
  recs = np.array([(0, 0, c if with_gamma else -1, nb, c, 0, 0),
                   (nb * 2 * c, 2 * c, 3 * c if with_gamma else -1, nb, c, chunks, 0)], dtype=ddt)
  desc = torch.from_numpy(recs.view(np.uint8).copy()).to(DEV)
  outs = []
  for _ in range(2):
    ws = torch.full((2 * nb * 2 * c,), float("nan"), device=DEV)
    grads = torch.ones(4 * c, device=DEV)          # the reduce ACCUMULATES into the gradients
    dc = torch.empty(rows, c, device=DEV)
    for layer in range(2):
      ops.bn_relu_bwd_partial(_t(dy), c, 0, _t(y), c, 0, _t(scale), _t(beta),
                              _t(gamma) if with_gamma else None, dc,
                              ws[layer * nb * 2 * c:(layer + 1) * nb * 2 * c], rows, c)
    ops.bn_partials_reduce_batched(desc, 2, 2 * chunks, ws, grads)
    outs.append(_n(grads))
    np.testing.assert_allclose(_n(dc), (dz * scale).astype(np.float32), rtol=1e-6, atol=1e-6)
  np.testing.assert_array_equal(outs[0], outs[1])              # bitwise reproducible
  tol = dict(rtol=1e-4, atol=1e-5 * np.sqrt(rows) + 1e-4)
  for layer in range(2):
    np.testing.assert_allclose(outs[0][2 * layer * c:(2 * layer + 1) * c] - 1.0, dz.sum(0), **tol)
    if with_gamma:
      np.testing.assert_allclose(outs[0][(2 * layer + 1) * c:(2 * layer + 2) * c] - 1.0,
                                 (dz * (y - beta) / gamma).sum(0), **tol)
    else:
      np.testing.assert_array_equal(outs[0][(2 * layer + 1) * c:(2 * layer + 2) * c], 1.0)

def test_spatial_mean_dropout(ops):
  rng = np.random.default_rng(23)
  rows, sp, c = 50, 16, 64
  x = rng.standard_normal((rows, sp, c)).astype(np.float32)
  mask = torch.empty(rows, c, dtype=torch.uint8, device=DEV)
  ops.dropout_mask(mask, 1234, 0.5)
  m = _n(mask)
  assert 0.4 < m.mean() < 0.6 and set(np.unique(m)) <= {0, 1}
  y = torch.empty(rows, c, device=DEV)
  ops.spatial_mean_dropout_fwd(_t(x), y, mask, rows, sp, c, 0.5)
  np.testing.assert_allclose(_n(y), x.mean(1) * 2.0 * m, rtol=1e-5, atol=1e-6)
  dy = rng.standard_normal((rows, c)).astype(np.float32)
  dx = torch.empty(rows, sp, c, device=DEV)
  ops.spatial_mean_dropout_bwd(_t(dy), c, 0, dx, mask, rows, sp, c, 0.5)
  np.testing.assert_allclose(_n(dx), np.broadcast_to((dy * 2.0 * m / sp)[:, None, :], x.shape),
                             rtol=1e-6, atol=1e-7)


def test_stem_preprocess_im2col(ops):
  rng = np.random.default_rng(29)
  img = rng.uniform(0, 255, (1, 21, 17, 3)).astype(np.float32)
  x4 = torch.empty(1, 21, 17, 4, device=DEV)
  ops.preprocess_pad4(_t(img), x4)
  want = ref_model.preprocess(img)
  np.testing.assert_allclose(_n(x4)[..., :3], want, rtol=1e-6, atol=1e-6)
  assert np.all(_n(x4)[..., 3] == 0)
  oh, ow = 11, 9
  cols = torch.empty(oh * ow, 208, device=DEV)
  ops.im2col4(x4, cols, 1, 21, 17, 7, 7, 2, 208)
  w = rng.standard_normal((7, 7, 3, 5)).astype(np.float32)
  want_c = ref_ops.conv2d(want, w, 2)
  w4 = np.zeros((7, 7, 4, 5), np.float32); w4[:, :, :3] = w
  got_c = _n(cols)[:, :196] @ w4.reshape(196, 5)
  np.testing.assert_allclose(got_c.reshape(1, oh, ow, 5), want_c, rtol=1e-4, atol=1e-4)
  assert np.all(_n(cols)[:, 196:] == 0)


@pytest.mark.parametrize("nb", [[37, 20], [37, 0], [1, 5]])
def test_midn_and_bce(ops, nb):
  rng = np.random.default_rng(31)
  b, n, c, d = 2, 37, 5, 32
  x = rng.standard_normal((b, n, d)).astype(np.float32)
  P = ref_model.init_head_params(rng, d, c, 0, stddev=0.5)
  num = np.asarray(nb, np.int32)
  cl, scores, proba, saved = ref_model.build_midn_network(num, x, P)
  ld = 16
  logits = np.zeros((b * n, ld), np.float32)
  logits[:, 0:c] = (x @ P["midn/proba_r_given_c/weights"]).reshape(-1, c)
  logits[:, 8:8 + c] = (x @ P["midn/proba_c_given_r/weights"]).reshape(-1, c)
  tl = _t(logits); tn = _t(num)
  proba_g = torch.empty(b, n, c, device=DEV); cl_g = torch.empty(b, c, device=DEV)
  sc_g = torch.empty(b, n, c, device=DEV)
  ops.midn_fwd(tl, ld, 0, 8, tn, proba_g, cl_g, sc_g, b, n, c)
  np.testing.assert_allclose(_n(proba_g), proba, rtol=1e-5, atol=1e-7)
  np.testing.assert_allclose(_n(cl_g), cl, rtol=1e-5, atol=1e-6)
  np.testing.assert_allclose(_n(sc_g), scores, rtol=1e-5, atol=1e-7)
  labels = (rng.uniform(size=(b, c)) > 0.6).astype(np.float32)
  loss = torch.zeros(1, device=DEV); dcl = torch.empty(b, c, device=DEV)
  ops.sigmoid_ce_fwd_bwd(cl_g, _t(labels), 1.0, loss, dcl)
  want_loss = ref_ops.sigmoid_cross_entropy_with_logits(labels, cl).mean()
  np.testing.assert_allclose(_n(loss)[0], want_loss, rtol=1e-5)
  want_dcl = (ref_ops.sigmoid(cl) - labels) / labels.size
  np.testing.assert_allclose(_n(dcl), want_dcl, rtol=1e-5, atol=1e-7)
  dlr, dlc = ref_model.build_midn_network_backward(want_dcl, saved)
  dl = torch.zeros(b * n, ld, device=DEV)
  ops.midn_bwd(dcl, tl, ld, 0, 8, tn, proba_g, cl_g, dl, ld, b, n, c)
  np.testing.assert_allclose(_n(dl)[:, 0:c].reshape(b, n, c), dlr, rtol=1e-4, atol=1e-7)
  np.testing.assert_allclose(_n(dl)[:, 8:8 + c].reshape(b, n, c), dlc, rtol=1e-4, atol=1e-7)


@pytest.mark.parametrize("c", [5, 80])
def test_oicr_select_and_loss(ops, c):
  rng = np.random.default_rng(37)
  b, n = 2, 61
  num = np.asarray([61, 40], np.int32)
  boxes = np.stack([_edge_boxes(rng, n), _edge_boxes(rng, n)]).astype(np.float32)
  s0 = rng.uniform(0, 1, (b, n, c + 1)).astype(np.float32)
  s0[0, 3, 2] = s0[0, 9, 2] = 5.0        # tied maxima -> first index
  s1 = rng.standard_normal((b, n, c + 1)).astype(np.float32)
  labels = (rng.uniform(size=(b, c)) > 0.5).astype(np.float32)
  labels[1, :] = 0; labels[1, 0] = 1
  want_loss, want_ds, _ = ref_model.calc_oicr_loss(labels, num, boxes, s0, s1, 0.6)
  mask = ref_ops.sequence_mask(num, n)
  want_idx = ref_ops.masked_argmax(s0[:, :, 1:], mask[..., None], dim=1)
  idx = torch.empty(b, c, dtype=torch.int32, device=DEV)
  top = torch.empty(b, c, 4, device=DEV)
  ops.oicr_select(_t(s0), c + 1, 1, _t(num), _t(boxes), idx, top, b, n, c)
  np.testing.assert_array_equal(_n(idx), want_idx)
  loss = torch.zeros(1, device=DEV)
  ds = torch.zeros(b * n, c + 1, device=DEV); q = torch.empty(b * n, c + 1, device=DEV)
  ops.oicr_loss_fwd_bwd(_t(s1), c + 1, 0, top, _t(boxes), _t(labels), _t(num), 0.6, 0.5, b, n, c,
                        loss, ds, c + 1, 0, q)
  np.testing.assert_allclose(_n(loss)[0], 0.5 * want_loss, rtol=2e-5)
  np.testing.assert_allclose(_n(ds).reshape(b, n, c + 1), 0.5 * want_ds, rtol=1e-4, atol=1e-8)
  np.testing.assert_allclose(_n(q).reshape(b, n, c + 1), ref_ops.softmax(s1), rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("b,n,c,stages", [(2, 61, 5, 3), (1, 2000, 20, 3), (2, 300, 80, 2), (1, 77, 7, 1)])
def test_oicr_refine_all_stages_at_once(ops, b, n, c, stages):
  """c2d_oicr_refine_fwd_bwd (three launches for all stages) against the stage-by-stage calls it
  replaces — indices, top boxes, softmax planes and score gradients BITWISE, the loss scalars to the
  order of their atomics — and stage by stage against the oracle (models/cap2det_model.py:306-330:
  stage k + 1 selects on softmax(scores of stage k)[..., 1:]).  Scores live in the column slices of
  a wider head-logits buffer, as in the model."""
  rng = np.random.default_rng(53 + n)
  num = np.minimum(n, rng.integers(max(1, n // 2), n + 1, b)).astype(np.int32); num[0] = n
  boxes = np.stack([_edge_boxes(rng, n) for _ in range(b)]).astype(np.float32)
  ld, off = 2 * c + stages * (c + 1) + 3, 2 * c
  logits = rng.standard_normal((b * n, ld)).astype(np.float32)
  s0 = rng.uniform(0, 1, (b, n, c)).astype(np.float32)
  s0[0, 1, 0] = s0[0, min(9, n - 1), 0] = 5.0        # tied maxima -> first index
  labels = (rng.uniform(size=(b, c)) > 0.5).astype(np.float32); labels[0, 0] = 1
  tl, ts0, tb, tlab, tnum = _t(logits), _t(s0), _t(boxes), _t(labels), _t(num)
  # stage by stage
  idx1 = torch.empty(stages, b, c, dtype=torch.int32, device=DEV)
  top1 = torch.empty(stages, b, c, 4, device=DEV)
  q1 = torch.empty(stages, b * n, c + 1, device=DEV)
  loss1 = torch.zeros(stages, device=DEV); ds1 = torch.zeros(b * n, ld, device=DEV)
  src, sld, soff = ts0, c, 0
  for k in range(stages):
    ops.oicr_select(src, sld, soff, tnum, tb, idx1[k], top1[k], b, n, c)
    ops.oicr_loss_fwd_bwd(tl, ld, off + k * (c + 1), top1[k], tb, tlab, tnum, 0.5, 0.7, b, n, c,
                          loss1[k:k + 1], ds1, ld, off + k * (c + 1), q1[k])
    src, sld, soff = q1[k], c + 1, 1
  # at once
  idx2 = torch.full((stages, b, c), -1, dtype=torch.int32, device=DEV)
  top2 = torch.full((stages, b, c, 4), -1.0, device=DEV)
  q2 = torch.full((stages, b * n, c + 1), -1.0, device=DEV)
  loss2 = torch.zeros(stages, device=DEV); ds2 = torch.zeros(b * n, ld, device=DEV)
  ops.oicr_refine_fwd_bwd(tl, ld, off, stages, ts0, c, 0, tb, tlab, tnum, 0.5, 0.7, b, n, c, loss2, ds2,
                          ld, off, q2, idx2, top2)
  torch.cuda.synchronize()
  assert torch.equal(idx1, idx2) and torch.equal(top1, top2)
  assert torch.equal(q1, q2) and torch.equal(ds1, ds2)
  np.testing.assert_allclose(_n(loss2), _n(loss1), rtol=2e-6)
  assert float(ds2[:, :off].abs().max()) == 0 and float(ds2[:, off + stages * (c + 1):].abs().max()) == 0
  # oracle, stage by stage
  prev = np.concatenate([np.zeros((b, n, 1), np.float32), s0], axis=-1)
  mask = ref_ops.sequence_mask(num, n)
  for k in range(stages):
    sk = logits.reshape(b, n, ld)[:, :, off + k * (c + 1):off + (k + 1) * (c + 1)]
    want_idx = ref_ops.masked_argmax(prev[:, :, 1:], mask[..., None], dim=1)
    np.testing.assert_array_equal(_n(idx2[k]), want_idx)
    want_loss, want_ds, _ = ref_model.calc_oicr_loss(labels, num, boxes, prev, sk, 0.5)
    np.testing.assert_allclose(_n(loss2)[k], 0.7 * want_loss, rtol=3e-5)
    got = _n(ds2).reshape(b, n, ld)[:, :, off + k * (c + 1):off + (k + 1) * (c + 1)]
    np.testing.assert_allclose(got, 0.7 * want_ds, rtol=1e-4, atol=1e-8)
    prev = ref_ops.softmax(sk)
    np.testing.assert_allclose(_n(q2[k]).reshape(b, n, c + 1), prev, rtol=1e-5, atol=1e-8)
    prev = _n(q2[k]).reshape(b, n, c + 1)      # (the next selection sees the device's own bits)


def test_labels_and_text_classifier(ops):
  rng = np.random.default_rng(41)
  b, t, v, e, h, c = 4, 9, 50, 300, 400, 7
  ids = rng.integers(0, v + 1, (b, t)).astype(np.int32)
  ids[2, :] = v                      # all-OOV caption
  ids[3, :3] = [1, 2, 3]
  lab = torch.empty(b, c, device=DEV)
  ops.labels_from_ids(_t(ids), c, lab)
  want = np.zeros((b, c), np.float32)
  for i in range(b):
    for j in ids[i]:
      if j < c:
        want[i, j] = 1
  np.testing.assert_array_equal(_n(lab), want)
  emb = (0.4 * rng.standard_normal((v + 1, e))).astype(np.float32)
  w1 = (rng.standard_normal((e, h)) / np.sqrt(e)).astype(np.float32)
  b1 = (0.1 * rng.standard_normal(h)).astype(np.float32)
  w2 = (rng.standard_normal((h, c)) / np.sqrt(h)).astype(np.float32)
  b2 = (0.1 * rng.standard_normal(c)).astype(np.float32)
  exact = np.zeros((b, c), np.float32); exact[3, 2] = 1
  want_logits = ref_labels.text_classifier_logits(ids, emb, w1, b1, w2, b2)
  want_labels = ref_labels.text_classifier_match_extract(ids, exact, emb, w1, b1, w2, b2, 0.5)
  safe = np.abs(ref_ops.sigmoid(want_logits) - 0.5) > 1e-3
  # one workgroup per caption, and the workspace form (hidden units spread over workgroups)
  for ws in (None, torch.empty(b, h, device=DEV)):
    logits = torch.full((b, c), 7.0, device=DEV); labels = torch.full((b, c), 7.0, device=DEV)
    ops.text_classifier_fwd(_t(ids), _t(emb), _t(w1), _t(b1), _t(w2), _t(b2), _t(exact), 0.5,
                            logits, labels, workspace=ws)
    np.testing.assert_allclose(_n(logits), want_logits, rtol=1e-4, atol=1e-4)
    np.testing.assert_array_equal(_n(labels)[safe], want_labels[safe])


@pytest.mark.parametrize("t,h,c", [(60, 400, 80), (1, 400, 80), (17, 96, 20), (33, 130, 5)])
def test_text_classifier_real_sizes(ops, t, h, c):
  """BASELINE configs[3]/[4] sizes (60 tokens, 300-d GloVe, 400 hidden units, 80 classes) and
  ragged ones (token counts that are not a multiple of the 16-token chunk, hidden widths that
  are not a multiple of 64), both kernel forms, incl. an all-OOV and a one-real-token caption."""
  rng = np.random.default_rng(43 + t)
  b, v, e = 3, 500, 300
  ids = rng.integers(0, v + 1, (b, t)).astype(np.int32)
  ids[1, :] = v
  ids[2, 1:] = v
  emb = (0.4 * rng.standard_normal((v + 1, e))).astype(np.float32)
  w1 = (rng.standard_normal((e, h)) / np.sqrt(e)).astype(np.float32)
  b1 = (0.1 * rng.standard_normal(h)).astype(np.float32)
  w2 = (rng.standard_normal((h, c)) / np.sqrt(h)).astype(np.float32)
  b2 = (0.1 * rng.standard_normal(c)).astype(np.float32)
  want = ref_labels.text_classifier_logits(ids, emb, w1, b1, w2, b2)
  got = []
  for ws in (None, torch.empty(b, h, device=DEV)):
    logits = torch.full((b, c), 7.0, device=DEV)
    ops.text_classifier_fwd(_t(ids), _t(emb), _t(w1), _t(b1), _t(w2), _t(b2), None, 0.0, logits,
                            None, workspace=ws)
    np.testing.assert_allclose(_n(logits), want, rtol=1e-4, atol=1e-4)
    got.append(_n(logits))


def test_adagrad_and_l2(ops):
  rng = np.random.default_rng(43)
  n = 10007
  w = rng.standard_normal(n).astype(np.float32); g = rng.standard_normal(n).astype(np.float32)
  acc = np.full(n, 0.1, np.float32)
  tw, ta = _t(w), _t(acc)
  ops.adagrad_step(tw, _t(g), ta, 0.01, 1e-4, 1.0, 0.5)
  g2 = 0.5 * g + 1e-4 * w
  acc2 = acc + g2 * g2
  np.testing.assert_allclose(_n(ta), acc2, rtol=1e-6)
  np.testing.assert_allclose(_n(tw), w - 0.01 * g2 / np.sqrt(acc2), rtol=1e-6, atol=1e-7)
  out = torch.zeros(1, device=DEV)
  ops.l2_loss(_t(w), 1e-2, out)
  np.testing.assert_allclose(_n(out)[0], 0.5e-2 * (w.astype(np.float64) ** 2).sum(), rtol=1e-5)
  # a variable that starts at an odd float of the flat buffer (no 16-byte loads), accumulating
  tw1 = _t(w)[1:]
  ops.l2_loss(tw1, 1e-2, out)
  np.testing.assert_allclose(_n(out)[0], 0.5e-2 * ((w.astype(np.float64) ** 2).sum() +
                                                   (w[1:].astype(np.float64) ** 2).sum()), rtol=1e-5)



@pytest.mark.parametrize("kind,opts,flags", [
    ("sgd", {}, 0), ("momentum", dict(momentum=0.9), 0), ("momentum", dict(momentum=0.9, use_nesterov=True), 1),
    ("adam", dict(beta1=0.9, beta2=0.999, epsilon=1e-8), 0),
    ("rmsprop", dict(decay=0.9, momentum=0.0, epsilon=1e-10), 0),
    ("rmsprop", dict(decay=0.9, momentum=0.5, epsilon=1e-10, centered=True), 2)])
def test_optimizer_step_kinds(ops, kind, opts, flags):
  """c2d_optimizer_step (core/training_utils.py:14-71's sgd / momentum / adam / rmsprop) against
  the oracle's TensorFlow 1.x rules over three steps, with L2 / L1 regularisers, a gradient
  multiplier, the 1/world gradient scale and per-column multipliers incl. a frozen column."""
  rng = np.random.default_rng(11)
  rows, ld = 37, 12
  n = rows * ld
  w0 = rng.standard_normal(n).astype(np.float32)
  col = np.array([1.0, 0.5, 0.0, 2.0] * 3, np.float32)
  lr, l1, l2, mult, scale = 0.05, 1e-3, 1e-2, 0.5, 0.25
  w = _t(w0)
  nslots = {"sgd": 0, "momentum": 1, "adam": 2, "rmsprop": 3 if opts.get("centered") else 2}[kind]
  slots = [torch.zeros(n, device=DEV) for _ in range(nslots)]
  if kind == "rmsprop":
    slots[0].fill_(1.0)
  w64 = w0.astype(np.float64)
  sl64 = ref_model.init_optimizer_slots(kind, opts, {"w": w64})["w"]
  m_el = (mult * np.tile(col, rows)).astype(np.float64)
  for step in range(1, 4):
    g = rng.standard_normal(n).astype(np.float32)
    p = {"sgd": (0, 0, 0, 0), "momentum": (opts.get("momentum", 0), 0, 0, 0),
         "adam": (opts.get("beta1", 0), opts.get("beta2", 0), opts.get("epsilon", 0),
                  lr * np.sqrt(1 - opts.get("beta2", 0) ** step) / (1 - opts.get("beta1", 0.5) ** step)),
         "rmsprop": (opts.get("decay", 0), opts.get("momentum", 0), opts.get("epsilon", 0), 0)}[kind]
    ops.optimizer_step(kind, w, _t(g), slots, lr, p, flags, l1, l2, mult, scale, _t(col), ld)
    gp = m_el * (scale * g.astype(np.float64) + l2 * w64 + l1 * np.sign(w64))
    live = m_el > 0
    wl, sll = w64[live].copy(), [a[live].copy() for a in sl64]
    ref_model.optimizer_update(kind, opts, wl, gp[live], sll, lr, step, np.float64)
    w64[live] = wl
    for a, b in zip(sl64, sll):
      a[live] = b
    np.testing.assert_allclose(_n(w), w64, rtol=2e-5, atol=2e-6, err_msg="step %d" % step)
  frozen = np.tile(col, rows) == 0
  np.testing.assert_array_equal(_n(w)[frozen], w0[frozen])
  for a, b in zip(slots, sl64):
    np.testing.assert_allclose(_n(a), b, rtol=2e-5, atol=2e-6)


def test_conv_fwd_grouped_matches_single_calls():
  """c2d_conv_fwd_grouped: independent convolutions of one Inception level in ONE launch (all
  small) or one launch each (a large one in the group): bit-identical to c2d_conv_fwd."""
  from cap2det_amd import hip_ops as ops
  rng = np.random.default_rng(11)
  for n, hw, big in [(1, 32, False), (1, 63, False), (3, 9, False), (400, 7, True)]:
    cin = 64
    x = torch.from_numpy(rng.standard_normal((n * hw * hw, cin + 16)).astype(np.float32)).to(DEV)
    calls, singles = [], []
    for (cout, k, stride) in [(96, 1, 1), (32, 3, 1), (48, 3, 2), (20, 1, 1), (64, 1, 2)]:
      oh = -(-hw // stride)
      wt = torch.from_numpy((rng.standard_normal((k * k, cout, cin)) / np.sqrt(k * k * cin)).astype(np.float32)).to(DEV)
      sc = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32)).to(DEV)
      sh = torch.from_numpy((0.1 * rng.standard_normal(cout)).astype(np.float32)).to(DEV)
      y1 = torch.full((n * oh * oh, cout + 8), -3.0, device=DEV)
      y2 = y1.clone()
      args = [x, cin + 16, 16, wt, sc, sh, None, cout + 8, 4, n, hw, hw, cin, cout, k, k, stride, True]
      calls.append(tuple(args[:6] + [y1] + args[7:]))
      singles.append((args, y2))
    group = ops.conv_group(calls)
    ops.conv_fwd_grouped(group)
    ops.conv_fwd_grouped(group)      # descriptors are reusable
    for (args, y2), c in zip(singles, calls):
      ops.conv_fwd(*(args[:6] + [y2] + args[7:]))
      assert torch.equal(c[6], y2), (n, hw, args[13:17])
      assert float(y2[:, :4].max()) == -3.0


@pytest.mark.parametrize("hf,wf,d,n,crop,pk,ps", [(32, 32, 64, 300, 14, 2, 2), (9, 33, 16, 60, 14, 2, 2),
                                                   (32, 32, 576, 48, 14, 2, 2), (63, 84, 576, 24, 14, 2, 2),
                                                   (20, 20, 32, 50, 8, 2, 2), (16, 16, 32, 40, 9, 3, 3),
                                                   (16, 16, 32, 40, 14, 2, 1)])
def test_roi_crop_pool_forms_match_oracle(hf, wf, d, n, crop, pk, ps):
  """The column-streaming 2x2/stride-2 kernel (first five cases; two at the benchmark's depth 576,
  on the 32x32 map of a 500-px image and the 63x84 map of a 1000x1333 one) and the generic kernel
  (other poolings) against crop_and_resize + max_pool of the oracle, bit for bit incl. the arg-max."""
  from cap2det_amd import hip_ops as ops
  rng = np.random.default_rng(hf * 100 + n)
  feat = np.maximum(rng.standard_normal((2, hf, wf, d)), 0).astype(np.float32)
  feat[0, 3:6] = 0.5                                  # ties inside pooling windows
  boxes = np.concatenate([_edge_boxes(rng, n // 2), util_boxes(rng, n - n // 2)])
  ind = rng.integers(0, 2, n).astype(np.int32)
  c = ref_ops.crop_and_resize(feat, boxes, ind, crop)
  want, want_arg = ref_ops.max_pool(c, pk, ps, "VALID")
  out, arg = ops.roi_crop_pool_fwd(_t(feat), _t(boxes), _t(ind), crop, pk, ps)
  np.testing.assert_array_equal(_n(out), want)
  np.testing.assert_array_equal(_n(arg), want_arg)


def util_boxes(rng, n):
  from tests import util_model
  return util_model.synthetic_boxes(rng, n)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("case", [(300, 4, 4, 64, 96, 3, 1), (300, 7, 7, 32, 64, 1, 1),
                                  (260, 7, 7, 32, 64, 3, 2), (257, 7, 7, 64, 160, 3, 1)])
def test_conv_wgrad_partial_slabs(ops, case, dtype):
  """c2d_conv_wgrad(_bf16)_partial + c2d_wgrad_reduce_batched (split-K slabs with plain stores,
  summed in split order) against c2d_conv_wgrad's atomics and the float64 oracle; two runs are
  bitwise equal (the point of the slab form)."""
  n, ih, iw, cin, cout, k, s = case
  rng = np.random.default_rng(77)
  x, w = _conv_inputs(rng, n, ih, iw, cin, cout, k)
  oh, ow = -(-ih // s), -(-iw // s)
  dc = rng.standard_normal((n, oh, ow, cout)).astype(np.float32)
  tx, tdc = _t(x).to(dtype), _t(dc).to(dtype)
  x64, dc64 = _n(tx.float()).astype(np.float64), _n(tdc.float()).astype(np.float64)
  _, want = ref_ops.conv2d_backward(x64, w.astype(np.float64), dc64, s, need_dx=False)
  splits = ops.conv_wgrad_splits(dtype, cin, 0, cout, 0, n, ih, iw, cin, cout, k, k, s)
  assert splits >= 1
  numel = k * k * cin * cout
  outs = []
  for _ in range(2):
    ws = torch.full((splits * numel + 8,), 7.0, device=DEV)
    ops.conv_wgrad_partial(tx, cin, 0, tdc, cout, 0, ws[:splits * numel], n, ih, iw, cin, cout, k, k, s)
    grads = torch.full((numel + 64,), 0.5, device=DEV)
    desc, num, chunks = ops.wgrad_reduce_descriptors([(0, 32, numel, splits)], DEV)
    ops.wgrad_reduce_batched(desc, num, chunks, ws, grads)
    assert float(ws[splits * numel:].min()) == 7.0 and float(grads[:32].max()) == 0.5
    assert float(grads[32 + numel:].min()) == 0.5
    outs.append(grads[32:32 + numel].clone())
  assert torch.equal(outs[0], outs[1])
  got = _n(outs[0]).reshape(want.shape) - 0.5
  scale = np.abs(want).max()
  np.testing.assert_allclose(got, want, rtol=2e-4, atol=(1e-4 if dtype == torch.bfloat16 else 2e-5) * scale)


def test_fused_entry_points_agree_with_unfused_launches_on_random_shapes():
  """tools/fuzz_fused.py (30 random shapes per fused entry point, fp32 and bf16): the fused
  input gradient, the multi-output 1x1 forward and the block-boundary input gradient against the
  unfused launches they replace."""
  import subprocess, sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_fused.py"), "30", "11"],
                     capture_output=True, text=True, timeout=600)
  assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
  assert "0 mismatches" in r.stdout


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
@pytest.mark.parametrize("rows,cin,couts", [(4096, 576, (128, 192)), (3000, 1024, (352, 192, 160, 128)),
                                            (1000, 64, (32, 48, 80)), (2500, 96, (40,))])
def test_conv1x1_wgrad_multi_matches_the_oracle(ops, rows, cin, couts, dtype):
  """c2d_conv1x1_wgrad_multi(_bf16): the filter gradients of several 1x1 convolutions of one input
  in ONE launch (shared row splits) against x^T . dc in float64 on the same operands, inputs and
  gradients inside wider buffers, ACCUMULATING into non-zero dw; the instance is the grouped one."""
  rng = np.random.default_rng(rows + cin)
  low = dtype == torch.bfloat16
  ldx, xoff = cin + 16, 8
  xb = torch.from_numpy(rng.standard_normal((rows, ldx)).astype(np.float32)).to(DEV).to(dtype)
  x64 = xb[:, xoff:xoff + cin].float().cpu().numpy().astype(np.float64)
  dcs, ldcs, coffs, dws, wants = [], [], [], [], []
  for c in couts:
    ld, off = c + 24, 16
    d = torch.from_numpy(rng.standard_normal((rows, ld)).astype(np.float32)).to(DEV).to(dtype)
    base = rng.standard_normal((cin, c)).astype(np.float32)
    dcs.append(d); ldcs.append(ld); coffs.append(off)
    dws.append(torch.from_numpy(base.copy()).to(DEV))
    wants.append(base.astype(np.float64) + x64.T @ d[:, off:off + c].float().cpu().numpy().astype(np.float64))
  ops.conv1x1_wgrad_multi(xb, ldx, xoff, dcs, ldcs, coffs, dws, list(couts), rows, cin)
  inst = ops.last_dispatch()
  assert inst == (["wgrad1x1_bf16_ring_group_kernel<2, 64, 2>"] if low else ["wgrad_tn_group_kernel<1, 4>"]), inst
  for dw, want in zip(dws, wants):
    err = np.abs(dw.cpu().numpy().astype(np.float64) - want).max()
    assert err <= 3e-5 * np.abs(want).max(), (err, np.abs(want).max())
