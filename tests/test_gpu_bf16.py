"""bf16 storage / fp32 accumulate path (BASELINE.json configs[2] and [4]).  The reference has no
reduced-precision mode, so the oracle is the float64 restatement applied to the SAME bf16-rounded
operands; what is left is the fp32 accumulation order (~1e-6 relative) and ONE round-to-nearest
bf16 of the result (relative 2^-9 = 0.2 %).  Tolerance: |got - want| <= 2^-8 |want| + 1e-3 scale."""
import numpy as np
import pytest
import torch

from oracle import ref_ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _bf(a):
  """numpy fp32 -> (bf16 device tensor, the exactly representable fp64 values)."""
  t = torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(DEV).to(torch.bfloat16)
  return t, t.float().cpu().numpy().astype(np.float64)


def _check(got_bf16, want, what):
  got = got_bf16.float().cpu().numpy().astype(np.float64)
  scale = np.abs(want).max()
  err = np.abs(got - want)
  assert (err <= 2.0 ** -8 * np.abs(want) + 1e-3 * scale).all(), \
      "%s: max err %.3e at scale %.3e" % (what, err.max(), scale)


CASES = [  # n, ih, iw, cin, cout, k, stride
    (3, 7, 7, 32, 64, 1, 1), (5, 7, 7, 48, 96, 3, 1), (5, 7, 7, 32, 64, 3, 2),
    (300, 7, 7, 64, 160, 3, 1), (70, 4, 4, 48, 96, 3, 1), (100, 7, 7, 32, 64, 3, 2),
    (65, 4, 4, 32, 192, 3, 1), (900, 7, 7, 64, 96, 1, 1), (40, 4, 4, 80, 32, 1, 1),
    # full-width tiles (output widths 257..384: one 128 x N block of 8 waves), ragged in M and N:
    # pixel-major forward (N = 288) / input gradient (N = cin = 320, accumulating epilogue with
    # idle store lanes), row-major 1x1 forward (N = 352) / input gradient (N = 288)
    (70, 4, 4, 48, 288, 3, 1), (70, 4, 4, 320, 64, 3, 1), (700, 4, 4, 64, 352, 1, 1),
    (700, 4, 4, 288, 64, 1, 1),
    # the single-image first stage (one 32x32 output tile per block, K split over the four waves,
    # operands straight from memory: igemm_small_kernel<*, 2>), K tails of the 64-deep slabs and
    # odd tap counts per wave included
    (1, 32, 32, 96, 128, 3, 1), (1, 33, 31, 64, 96, 3, 2), (1, 32, 32, 576, 224, 1, 1),
    (2, 19, 23, 80, 48, 3, 1), (1, 63, 63, 192, 64, 1, 1)]


@pytest.mark.parametrize("case", CASES)
def test_conv_fwd_dgrad_bf16(case):
  from cap2det_amd import hip_ops as ops
  n, ih, iw, cin, cout, k, s = case
  rng = np.random.default_rng(sum(case))
  x, x64 = _bf(rng.standard_normal((n, ih, iw, cin)))
  w, w64 = _bf(rng.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin))
  scale = rng.uniform(0.5, 1.5, cout).astype(np.float32)
  shift = (0.1 * rng.standard_normal(cout)).astype(np.float32)
  c = ref_ops.conv2d(x64, w64, s)
  want = np.maximum(c * scale + shift, 0)
  oh, ow = want.shape[1:3]
  # input / output live in wider concat buffers (channel slices), as in the engine
  ldx, xoff, ldy, yoff = cin + 16, 8, cout + 32, 16
  xb = torch.zeros(n, ih, iw, ldx, device=DEV, dtype=torch.bfloat16); xb[..., xoff:xoff + cin] = x
  yb = torch.full((n, oh, ow, ldy), -7.0, device=DEV, dtype=torch.bfloat16)
  wt = w.permute(0, 1, 3, 2).contiguous().view(k * k, cout, cin)        # [tap][cout][cin]
  ops.conv_fwd(xb, ldx, xoff, wt, torch.from_numpy(scale).to(DEV), torch.from_numpy(shift).to(DEV), yb,
               ldy, yoff, n, ih, iw, cin, cout, k, k, s, True)
  _check(yb[..., yoff:yoff + cout], want, "fwd")
  assert float(yb[..., :yoff].float().max()) == -7.0 and float(yb[..., yoff + cout:].float().min()) == -7.0
  # dgrad (accumulating into an existing bf16 gradient)
  dc, dc64 = _bf(rng.standard_normal((n, oh, ow, cout)))
  want_dx, _ = ref_ops.conv2d_backward(x64, w64, dc64, s)
  base, base64 = _bf(0.5 * rng.standard_normal((n, ih, iw, cin)))
  dx = base.clone()
  ops.conv_dgrad(dc, cout, 0, w.view(k * k, cin, cout), dx, cin, 0, n, ih, iw, cin, cout, k, k, s, True)
  _check(dx, want_dx + base64, "dgrad+accumulate")
  dx2 = torch.empty_like(dx)
  ops.conv_dgrad(dc, cout, 0, w.view(k * k, cin, cout), dx2, cin, 0, n, ih, iw, cin, cout, k, k, s, False)
  _check(dx2, want_dx, "dgrad")


def test_first_stage_shapes_take_the_small_kernel():
  from cap2det_amd import hip_ops as ops
  x = torch.zeros(32 * 32, 96, device=DEV, dtype=torch.bfloat16)
  wt = torch.zeros(9, 128, 96, device=DEV, dtype=torch.bfloat16)
  y = torch.empty(32 * 32, 128, device=DEV, dtype=torch.bfloat16)
  ops.conv_fwd(x, 96, 0, wt, None, None, y, 128, 0, 1, 32, 32, 96, 128, 3, 3, 1, False)
  assert ops.last_dispatch() == ["igemm_small_kernel<0, 2>"]
  dx = torch.empty_like(x)
  ops.conv_dgrad(y, 128, 0, wt.view(9, 96, 128), dx, 96, 0, 1, 32, 32, 96, 128, 3, 3, 1, False)
  assert ops.last_dispatch() == ["igemm_small_kernel<1, 2>"]


def test_conv_fwd_grouped_bf16_matches_single_calls():
  """c2d_conv_fwd_grouped_bf16: the convolutions of one Inception level of the single-image first
  stage in ONE launch (igemm_small_group_kernel<0, 2>), bit-identical to c2d_conv_fwd_bf16; a
  group with a large member falls back to one launch each."""
  from cap2det_amd import hip_ops as ops
  rng = np.random.default_rng(12)
  for n, hw, big in [(1, 32, False), (1, 63, False), (3, 9, False), (400, 7, True)]:
    cin = 64
    x = torch.from_numpy(rng.standard_normal((n * hw * hw, cin + 16)).astype(np.float32)).to(DEV).to(torch.bfloat16)
    calls, singles = [], []
    for (cout, k, stride) in [(96, 1, 1), (32, 3, 1), (48, 3, 2), (24, 1, 1), (64, 1, 2)]:
      oh = -(-hw // stride)
      wt = torch.from_numpy((rng.standard_normal((k * k, cout, cin)) / np.sqrt(k * k * cin)).astype(np.float32)).to(DEV).to(torch.bfloat16)
      sc = torch.from_numpy(rng.uniform(0.5, 1.5, cout).astype(np.float32)).to(DEV)
      sh = torch.from_numpy((0.1 * rng.standard_normal(cout)).astype(np.float32)).to(DEV)
      y1 = torch.full((n * oh * oh, cout + 8), -3.0, device=DEV, dtype=torch.bfloat16)
      y2 = y1.clone()
      args = [x, cin + 16, 16, wt, sc, sh, None, cout + 8, 8, n, hw, hw, cin, cout, k, k, stride, True]
      calls.append(tuple(args[:6] + [y1] + args[7:]))
      singles.append((args, y2))
    group = ops.conv_group(calls)
    ops.conv_fwd_grouped(group)
    inst = ops.last_dispatch()
    assert (inst == ["igemm_small_group_kernel<0, 2>"]) == (not big), inst
    ops.conv_fwd_grouped(group)      # descriptors are reusable
    for (args, y2), c in zip(singles, calls):
      ops.conv_fwd(*(args[:6] + [y2] + args[7:]))
      assert torch.equal(c[6], y2), (n, hw, args[13:17])
      assert float(y2[:, :8].float().max()) == -3.0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["fp32", "bf16"])
def test_conv_dgrad_grouped_matches_single_calls(dtype):
  """c2d_conv_dgrad_grouped(_bf16): independent input gradients of one Inception level (single
  image: one igemm_small_group_kernel<1, ES> launch), bit-identical to the single calls, with and
  without accumulation into the destination."""
  from cap2det_amd import hip_ops as ops
  rng = np.random.default_rng(13)
  n, hw = 1, 32
  calls, singles = [], []
  for (cin, cout, k, acc) in [(64, 96, 1, False), (96, 32, 3, True), (48, 64, 3, False), (80, 32, 1, True)]:
    dc = torch.from_numpy(rng.standard_normal((n * hw * hw, cout + 8)).astype(np.float32)).to(DEV).to(dtype)
    w = torch.from_numpy((rng.standard_normal((k * k, cin, cout)) / np.sqrt(k * k * cout)).astype(np.float32)).to(DEV).to(dtype)
    base = torch.from_numpy(rng.standard_normal((n * hw * hw, cin + 8)).astype(np.float32)).to(DEV).to(dtype)
    dx1, dx2 = base.clone(), base.clone()
    args = [dc, cout + 8, 8, w, None, cin + 8, 8, n, hw, hw, cin, cout, k, k, 1, acc]
    calls.append(tuple(args[:4] + [dx1] + args[5:]))
    singles.append((args, dx2))
  group = ops.conv_dgrad_group(calls)
  ops.conv_dgrad_grouped(group)
  assert ops.last_dispatch() == ["igemm_small_group_kernel<1, %d>" % (2 if dtype == torch.bfloat16 else 4)]
  for (args, dx2), c in zip(singles, calls):
    ops.conv_dgrad(*(args[:4] + [dx2] + args[5:]))
    assert torch.equal(c[4], dx2), args[10:16]


def test_cast_f32_is_exact():
  from cap2det_amd import hip_ops as ops
  src = torch.randn(1024 * 576, device=DEV).to(torch.bfloat16)
  dst = torch.empty(src.numel(), device=DEV)
  ops.cast_f32(src, dst)
  assert torch.equal(dst, src.float())


FUSED_CASES = [(70, 4, 4, 48, 96, 3, 1), (900, 7, 7, 64, 96, 1, 1), (100, 7, 7, 32, 64, 3, 2),
               (70, 4, 4, 320, 64, 3, 1), (300, 7, 7, 160, 64, 3, 1), (40, 4, 4, 80, 32, 1, 1)]


@pytest.mark.parametrize("case", FUSED_CASES)
def test_conv_dgrad_bn_relu_bf16(case):
  """bf16 form of the fused input gradient + BN/ReLU backward (the full-width 128 x 320 tile of
  the (70, 4, 4, 320, ...) case included): dc within one bf16 rounding of the float64 oracle on
  the same bf16 operands, the fp32 column sums to 1e-3 of their scale."""
  from cap2det_amd import hip_ops as ops
  n, ih, iw, cin, cout, k, s = case
  rng = np.random.default_rng(sum(case) + 1)
  w, w64 = _bf(rng.standard_normal((k, k, cin, cout)) / np.sqrt(k * k * cin))
  oh, ow = -(-ih // s), -(-iw // s)
  dc, dc64 = _bf(rng.standard_normal((n, oh, ow, cout)))
  y, y64 = _bf(np.maximum(rng.standard_normal((n, ih, iw, cin)), 0))
  scale = rng.uniform(0.5, 1.5, cin).astype(np.float32)
  beta = (0.1 * rng.standard_normal(cin)).astype(np.float32)
  gamma = rng.uniform(0.5, 1.5, cin).astype(np.float32)
  x64 = np.zeros((n, ih, iw, cin))
  dx, _ = ref_ops.conv2d_backward(x64, w64, dc64, s)
  dz = dx * (y64 > 0)
  nb = ops.conv_dgrad_bn_relu_blocks(torch.bfloat16, n, ih, iw, cin, cout, k, k, s)
  assert nb >= 1
  out = torch.full((n * ih * iw, cin), 9.0, device=DEV, dtype=torch.bfloat16)
  part = torch.full((nb, 2, cin), 7.0, device=DEV)
  t = lambda a: torch.from_numpy(a).to(DEV)
  ops.conv_dgrad_bn_relu(dc, cout, 0, w.view(k * k, cin, cout), y, cin, 0, t(scale), t(beta), t(gamma),
                         out, part, n, ih, iw, cin, cout, k, k, s)
  _check(out.view(n, ih, iw, cin), dz * scale, "fused dc")
  sums = part.double().sum(0).cpu().numpy()
  want_db = dz.reshape(-1, cin).sum(0)
  want_dg = (dz * (y64 - beta) / gamma).reshape(-1, cin).sum(0)
  for got, want, what in ((sums[0], want_db, "dbeta"), (sums[1], want_dg, "dgamma")):
    assert np.abs(got - want).max() <= 1e-3 * max(np.abs(want).max(), 1.0), what


def test_conv1x1_dgrad_multi_bf16():
  from cap2det_amd import hip_ops as ops
  rng = np.random.default_rng(5)
  rows, cin, couts = 3000, 64, [32, 48, 80]
  dcs, ws, want = [], [], 0
  for co in couts:
    d, d64 = _bf(rng.standard_normal((rows, co)))
    w, w64 = _bf(rng.standard_normal((cin, co)) / np.sqrt(co))
    dcs.append(d); ws.append(w); want = want + d64 @ w64.T
  dx = torch.empty(rows, cin, device=DEV, dtype=torch.bfloat16)
  ops.conv1x1_dgrad_multi(dcs, couts, [0, 0, 0], ws, couts, dx, cin, 0, rows, cin, False)
  _check(dx, want, "multi-segment dgrad")


WGRAD_CASES = [  # n, ih, iw, cin, cout, k, stride     (kernel the bf16 call dispatches to)
    (300, 4, 4, 64, 160, 3, 1),    # nine-tap, 4x4 maps, two i-groups per block
    (261, 4, 4, 96, 32, 3, 1),     # nine-tap, 4x4, one i-group, ragged last slab
    (257, 7, 7, 32, 192, 3, 1),    # nine-tap, 7x7 (k-steps straddle image rows), one i-group
    (256, 7, 7, 128, 96, 3, 1),    # nine-tap, 7x7, two i-groups
    (900, 4, 4, 160, 200, 1, 1),   # per-tap PLAIN, ragged 128x128 tiles
    (123, 7, 7, 72, 64, 1, 1),     # per-tap PLAIN, narrow j tile
    (100, 7, 7, 64, 96, 3, 2),     # per-tap, stride 2 (7x7 -> 4x4)
    (5, 9, 6, 40, 48, 3, 1),       # per-tap 3x3 on a map the nine-tap kernel does not take
    (40, 4, 4, 36, 44, 1, 1)]      # channels not multiples of 8: widening fp32 kernels


@pytest.mark.parametrize("case", WGRAD_CASES)
def test_conv_wgrad_bf16(case):
  """Filter gradient from bf16 activations / gradients (fp32 result): the operands are exact in
  fp64, so only the fp32 accumulation order is left: |err| <= 2e-5 of the largest entry."""
  from cap2det_amd import hip_ops as ops
  n, ih, iw, cin, cout, k, s = case
  rng = np.random.default_rng(sum(case))
  x, x64 = _bf(rng.standard_normal((n, ih, iw, cin)))
  oh, ow = (ih + s - 1) // s, (iw + s - 1) // s
  dc, dc64 = _bf(rng.standard_normal((n, oh, ow, cout)))
  _, want = ref_ops.conv2d_backward(x64, np.zeros((k, k, cin, cout)), dc64, s)
  a8 = 8 if cin % 8 == 0 and cout % 8 == 0 else 4
  ldx, xoff, ldc, coff = cin + 3 * a8, a8, cout + 2 * a8, 2 * a8
  xb = torch.full((n, ih, iw, ldx), 3.0, device=DEV, dtype=torch.bfloat16); xb[..., xoff:xoff + cin] = x
  db = torch.full((n, oh, ow, ldc), -5.0, device=DEV, dtype=torch.bfloat16); db[..., coff:coff + cout] = dc
  dw = torch.zeros(k * k, cin, cout, device=DEV)
  ops.conv_wgrad(xb, ldx, xoff, db, ldc, coff, dw, n, ih, iw, cin, cout, k, k, s)
  got = dw.cpu().numpy().astype(np.float64).reshape(k, k, cin, cout)
  err = np.abs(got - want).max()
  assert err <= 2e-5 * np.abs(want).max(), "wgrad: max err %.3e at scale %.3e" % (err, np.abs(want).max())
  # accumulates into dw (the engine zeroes the gradient buffer once per step)
  ops.conv_wgrad(xb, ldx, xoff, db, ldc, coff, dw, n, ih, iw, cin, cout, k, k, s)
  got2 = dw.cpu().numpy().astype(np.float64).reshape(k, k, cin, cout)
  assert np.abs(got2 - 2 * want).max() <= 4e-5 * np.abs(want).max()


def _rel_l2(got, want):
  got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
  return float(np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30))


@pytest.mark.parametrize("first_stage", ["bf16", "fp32"])
@pytest.mark.parametrize("dm,hw,n,nums", [(1.0, (64, 48), 6, [6, 4]), (0.5, (40, 72), 9, [9, 0])])
def test_train_step_bf16_tracks_the_fp64_oracle(monkeypatch, dm, hw, n, nums, first_stage):
  """first_stage = "fp32": the single-image tower kept in fp32 (C2D_TUNE=first_stage_fp32=1, the bf16
  mode of rounds 2-3a) holds the tighter bounds of TOL_FP32_FIRST."""
  from oracle import ref_labels
  from tests import util_model
  monkeypatch.setenv("C2D_TUNE", "first_stage_fp32=%s" % ("1" if first_stage == "fp32" else "0"))
  _check_train_step_bf16(
      util_model.load_pipeline(), dm, hw, n, nums,
      lambda ex, classes: ref_labels.groundtruth_extract(ex["object_texts"], classes),
      tol=TOL_FP32_FIRST if first_stage == "fp32" else TOL)


def test_train_step_bf16_coco17_extend_match(tmp_path):
  """BASELINE configs[2] as named: coco17_extend_match (caption label extractor + MIL) in the
  bf16 storage mode.  The labels (fp32 / integer work) must equal the oracle's exactly."""
  from oracle import ref_labels
  from tests import util_model
  from tests.test_gpu_model import _captions, _coco_like_classes
  rng = np.random.default_rng(5)
  classes, syn = _coco_like_classes(rng)
  lf = tmp_path / "coco_label_synonyms.txt"
  lf.write_text("\n".join("%s\t%s" % (c, ",".join(s)) for c, s in zip(classes, syn)))
  pipeline = util_model.load_pipeline("coco17_extend_match_hotpath", label_file=str(lf))
  name2id, _ = ref_labels.read_synonym_file(str(lf))
  caps = _captions(rng, classes, syn)
  _check_train_step_bf16(
      pipeline, 0.5, (48, 40), 7, [7, 5],
      lambda ex, cl: ref_labels.extend_match_extract(ex["concat_caption_string"], name2id, len(cl)),
      extra_examples=lambda r, cl: {"concat_caption_string": caps}, seed=5)


def test_train_step_bf16_text_classifier_match(tmp_path):
  """BASELINE configs[4] as named: *_text_classifier_match (Flickr30k-sized vocabulary) in the
  bf16 storage mode; the text classifier itself stays fp32, labels equal the oracle's exactly."""
  from tests.test_gpu_model import _text_classifier_setup
  pipeline, make_labels, extra, check_oov = _text_classifier_setup(tmp_path, 211)
  check_oov(_check_train_step_bf16(pipeline, 0.5, (48, 40), 7, [7, 5], make_labels,
                                   extra_examples=extra, seed=99))


# relative bounds: proposal scores / logits (of the tensor's max), losses, a filter or head gradient
# tensor (L2), a BatchNorm beta / gamma gradient (L2), the whole gradient (L2)
TOL = dict(score=5e-2, loss=2e-2, grad=0.15, bn_grad=0.25, whole=5e-2)
TOL_FP32_FIRST = dict(score=2e-2, loss=2e-2, grad=0.10, bn_grad=0.20, whole=3e-2)


def _check_train_step_bf16(pipeline, dm, hw, n, nums, make_labels, extra_examples=None, seed=99,
                           tol=TOL):
  """Full training step with compute_dtype='bf16' (the convolution towers behind the stem — the
  single-image first stage, the ROI crop output and the second stage — in bf16 storage, fp32
  accumulation) against the float64 oracle of the reference semantics.  There is no
  bf16 reference: the stated tolerance (TOL) is what ~25 bf16 roundings per path (2^-9 relative
  each) allow — proposal scores within 5 % of the tensor's max (observed 3.4 %), losses 2 %
  relative (0.7 %), every filter / head gradient tensor within 15 % relative L2 error (these tiny
  cases sum over < 300 pixels; observed 11 %), the whole gradient within 5 % (4.0 %) and at cosine
  >= 0.999 of the oracle's; with the first stage kept in fp32 (~12 roundings): 2 % / 2 % / 10 % /
  3 % (TOL_FP32_FIRST).  The fp32 path (tests/test_gpu_model.py) holds 1e-4."""
  from cap2det_amd.train.trainer import Trainer
  from oracle import ref_model
  from tests import util_model
  rng = np.random.default_rng(seed)
  trainer = Trainer(pipeline, device=DEV, depth_multiplier=dm, compute_dtype="bf16")
  model = trainer.model
  assert model.engine.second.dtype == torch.bfloat16
  assert model.engine.first.dtype == (torch.float32 if tol is TOL_FP32_FIRST else torch.bfloat16)
  classes = model.label_extractor.classes
  c, k = len(classes), 3
  P32, d = util_model.oracle_state(5, c, k, dm)
  model.load_state_dict(P32)
  ex = util_model.make_examples(rng, 2, hw[0], hw[1], n, nums, classes)
  if extra_examples is not None:
    ex.update(extra_examples(rng, classes))
  mask = (rng.uniform(size=(2 * n, d)) < 0.5).astype(np.uint8)
  P = {kk: v.astype(np.float64) for kk, v in P32.items()}
  acc = {kk: np.full(v.shape, 0.1) for kk, v in P.items()}
  labels = make_labels(ex, classes).astype(np.float64)
  ex64 = dict(image=ex["image"].astype(np.float64), number_of_proposals=ex["number_of_proposals"],
              proposals=ex["proposals"].astype(np.float64))
  opts = ref_model.FrcnnOptions(depth_multiplier=dm)
  loss_opts = dict(midn_loss_weight=1.0, oicr_loss_weight=0.5, oicr_iterations=k,
                   oicr_iou_threshold=0.6, oicr_use_proba_r_given_c=True)
  mults = [(g.scope, g.multiplier) for g in pipeline.train_config.gradient_multiplier]
  P_before = {kk: v.copy() for kk, v in P.items()}
  want = ref_model.train_step(P, acc, ex64, labels, opts, loss_opts, mults, 0.01, 1e-6, mask)

  dev = dict(ex)
  for key in ("image", "proposals"):
    dev[key] = torch.from_numpy(ex[key]).to(DEV).contiguous()
  dev["number_of_proposals"] = torch.from_numpy(ex["number_of_proposals"]).to(DEV)
  losses = trainer.train_step(dev, dropout_mask=torch.from_numpy(mask).to(DEV))
  torch.cuda.synchronize()
  np.testing.assert_array_equal(model._ctx["labels"].cpu().numpy(), labels)   # extractor parity
  pred, wp = trainer.predictions, want["predictions"]
  seen = {}
  for name in ["midn_class_logits", "midn_proba_r_given_c"] + \
      ["oicr_proposal_scores_at_%d" % i for i in range(k + 1)]:
    got = pred[name].detach().float().cpu().numpy()
    seen[name] = np.abs(got - wp[name]).max() / max(np.abs(wp[name]).max(), 1e-6)
  for name, v in want["losses"].items():
    seen["loss " + name] = abs(losses[name].item() - v) / abs(v)
  print("bf16 step vs float64 oracle, relative deviations:", {kk: round(float(v), 4) for kk, v in seen.items()})
  for name, e in seen.items():
    assert e <= (tol["score"] if not name.startswith("loss ") else tol["loss"]), (name, e, seen)
  grads = model.grad_dict()
  worst = 0.0
  gots, wants = [], []
  for name in want["applied"]:
    w = want["grads"][name]
    if ref_model.is_regularized(name):
      w = w - 1e-6 * P_before[name]
    if np.linalg.norm(w) < 1e-12:
      continue
    e = _rel_l2(grads[name], w)
    worst = max(worst, e)
    # BatchNorm beta / gamma gradients are signed sums over only 96-294 pixels in this tiny case
    # (heavy cancellation): their own, wider bound
    bound = tol["bn_grad"] if "/BatchNorm/" in name else tol["grad"]
    assert e <= bound, "grad %s: relative L2 error %.3e" % (name, e)
    gots.append(np.asarray(grads[name], np.float64).ravel()); wants.append(np.asarray(w).ravel())
  assert worst > 1e-5        # (it really ran in reduced precision)
  g, w = np.concatenate(gots), np.concatenate(wants)
  cos = float(g @ w / (np.linalg.norm(g) * np.linalg.norm(w)))
  assert cos >= 0.999, "whole-gradient cosine similarity %.5f" % cos
  assert _rel_l2(g, w) <= tol["whole"], _rel_l2(g, w)
  return trainer


def test_feature_map_dropout_branch_bf16_tracks_fp32():
  """`dropout_on_feature_map: true` (models/utils.py:138-142; off in every shipped config) in the
  bf16 mode: the mask is applied to the fp32 widening of the bf16 tower's output in front of the
  ROI crop, its backward to the fp32 crop gradient before the cast for Mixed_4e.  Same injected
  masks in both modes: losses within 2 %, the whole gradient at cosine >= 0.995 (observed 0.9988 on
  this tiny case: 96-pixel feature maps, half of them dropped)."""
  from cap2det_amd.protos import cap2det_model_pb2
  from cap2det_amd.train.trainer import Trainer
  from tests import util_model
  dm, hw, n, nums = 0.5, (48, 56), 7, [7, 5]
  rng = np.random.default_rng(31)
  out = {}
  ex = mask = fmask = None
  for dtype in ("fp32", "bf16"):
    pipeline = util_model.load_pipeline()
    m = pipeline.model.Extensions[cap2det_model_pb2.Cap2DetModel.ext]
    m.frcnn_options.dropout_on_feature_map = True
    trainer = Trainer(pipeline, device=DEV, depth_multiplier=dm, compute_dtype=dtype)
    model = trainer.model
    classes = model.label_extractor.classes
    P32, d = util_model.oracle_state(6, len(classes), 3, dm)
    model.load_state_dict(P32)
    if ex is None:
      ex = util_model.make_examples(rng, 2, hw[0], hw[1], n, nums, classes)
      mask = (rng.uniform(size=(2 * n, d)) < 0.5).astype(np.uint8)
    dev = dict(ex)
    for key in ("image", "proposals"):
      dev[key] = torch.from_numpy(ex[key]).to(DEV).contiguous()
    dev["number_of_proposals"] = torch.from_numpy(ex["number_of_proposals"]).to(DEV)
    if fmask is None:
      bufs = model.engine._buffers(2, hw[0], hw[1], n, True)
      fmask = (rng.uniform(size=(2 * bufs["fh"] * bufs["fw"], model.engine.first.cout)) < 0.5).astype(np.uint8)
    losses = trainer.train_step(dev, dropout_mask=torch.from_numpy(mask).to(DEV),
                                feature_map_dropout_mask=torch.from_numpy(fmask).to(DEV))
    torch.cuda.synchronize()
    grads = model.grad_dict()
    out[dtype] = ({k: float(v) for k, v in losses.items()},
                  np.concatenate([np.asarray(grads[k], np.float64).ravel() for k in sorted(grads)]))
  (l32, g32), (l16, g16) = out["fp32"], out["bf16"]
  for k in l32:
    assert abs(l32[k] - l16[k]) <= 2e-2 * max(abs(l32[k]), 1e-6), (k, l32[k], l16[k])
  cos = float(g32 @ g16 / (np.linalg.norm(g32) * np.linalg.norm(g16)))
  assert cos >= 0.995, cos
  assert float(np.abs(g32 - g16).max()) > 0.0


@pytest.mark.parametrize("n,hw,layers", [
    (300, 4, [(64, 160), (96, 32), (32, 64)]),          # three 4x4 layers, ragged last slab
    (2000, 4, [(192, 320), (160, 224), (224, 224)]),    # Mixed_5b at the benchmark's size
    (257, 7, [(32, 192), (128, 96)]),                   # 7x7 maps, two layers
])
def test_conv3x3_wgrad_multi_bf16(n, hw, layers):
  """c2d_conv3x3_wgrad_multi_bf16 (round 5): the nine-tap filter gradients of a block's 3x3 layers
  in ONE launch with shared row splits — every output equal to the single-layer launch's up to the
  order of the split-K atomics, and within 1e-4 of scale of the float64 oracle on the same bf16
  operands (small case)."""
  from cap2det_amd import hip_ops as ops
  rng = np.random.default_rng(n + hw)
  probs, singles, keep = [], [], []
  for cin, cout in layers:
    x, x64 = _bf(rng.standard_normal((n, hw, hw, cin)))
    dc, dc64 = _bf(rng.standard_normal((n, hw, hw, cout)))
    dw = torch.zeros(9, cin, cout, device=DEV)
    dw1 = torch.zeros(9, cin, cout, device=DEV)
    probs.append((x.view(-1, cin), cin, 0, dc.view(-1, cout), cout, 0, dw, cin, cout))
    ops.conv_wgrad(x.view(-1, cin), cin, 0, dc.view(-1, cout), cout, 0, dw1, n, hw, hw, cin, cout, 3, 3, 1)
    assert ops.last_dispatch()[0].startswith("wgrad3x3_bf16_kernel<%d" % hw)
    singles.append(dw1)
    keep.append((x64, dc64))
  assert ops.conv3x3_wgrad_multi(probs, n, hw) is True
  assert ops.last_dispatch() == ["wgrad3x3_bf16_group_kernel<%d, %d>" % (hw, 8 if hw == 4 else 2)]
  for (cin, cout), p, dw1, (x64, dc64) in zip(layers, probs, singles, keep):
    scale = float(dw1.abs().max())
    assert float((p[6] - dw1).abs().max()) <= 2e-5 * scale, (cin, cout)
    if n <= 300:
      _, want = ref_ops.conv2d_backward(x64, np.zeros((3, 3, cin, cout)), dc64, 1, need_dx=False)
      assert np.abs(p[6].cpu().numpy().reshape(3, 3, cin, cout) - want).max() <= 1e-4 * np.abs(want).max()
  # a layer the nine-tap kernel does not take (cin not a multiple of 32): the group is declined
  x = torch.zeros(n * hw * hw, 40, device=DEV, dtype=torch.bfloat16)
  dc = torch.zeros(n * hw * hw, 64, device=DEV, dtype=torch.bfloat16)
  assert ops.conv3x3_wgrad_multi([(x, 40, 0, dc, 64, 0, torch.zeros(9, 40, 64, device=DEV), 40, 64)], n, hw) is False


def test_adagrad_step_multi_equals_the_separate_calls():
  """c2d_adagrad_step_multi: several segments (own multiplier / L2 weight, gaps between them) in one
  launch — values and accumulators BITWISE those of one c2d_adagrad_step per segment, elements
  outside the segments untouched, and the bf16 mirror equal to the rounded new values."""
  from cap2det_amd import hip_ops as ops
  g = torch.Generator(device="cpu"); g.manual_seed(5)
  n = 1_300_003
  v0 = torch.randn(n, generator=g).to(DEV); gr = torch.randn(n, generator=g).to(DEV)
  a0 = (torch.rand(n, generator=g) + 0.1).to(DEV)
  segs = [(0, 700_001, 1.0, 0.0), (700_004, 700_004, 3.0, 0.0), (800_000, 1_250_000, 1.0, 1e-4),
          (1_299_000, 1_300_003, 2.0, 0.0)]
  v1, a1 = v0.clone(), a0.clone()
  for off, end, m, l2 in segs:
    if end > off:
      ops.adagrad_step(v1[off:end], gr[off:end], a1[off:end], 0.01, l2, m, 0.5)
  for mirror in (False, True):
    v2, a2 = v0.clone(), a0.clone()
    m16 = torch.full((n,), 7.0, device=DEV, dtype=torch.bfloat16) if mirror else None
    ops.adagrad_step_multi(v2, gr, a2, segs, 0.01, 0.5, m16)
    torch.cuda.synchronize()
    assert torch.equal(v1, v2) and torch.equal(a1, a2)
    assert not torch.equal(v0, v2)
    if mirror:
      inside = torch.zeros(n, dtype=torch.bool, device=DEV)
      for off, end, _, _ in segs:
        inside[off:end] = True
      assert torch.equal(m16[inside], v2[inside].to(torch.bfloat16))
      assert bool((m16[~inside] == 7.0).all())


def test_step_leaves_consistent_bf16_mirrors(monkeypatch):
  """After optimiser steps of a bf16 network the bf16 mirrors the NEXT forward / backward pass reads
  are the rounded fp32 originals: every variable's mirror (written by c2d_adagrad_step_multi) and
  every layer's transposed operand (written by c2d_transpose_taps_batched_mirror)."""
  from cap2det_amd.train.trainer import Trainer
  from tests import util_model
  dm, hw, n, nums = 0.5, (64, 64), 9, [9, 6]
  rng = np.random.default_rng(77)
  for fused in (True,):
    pipeline = util_model.load_pipeline()
    trainer = Trainer(pipeline, device=DEV, depth_multiplier=dm, compute_dtype="bf16")
    model = trainer.model
    classes = model.label_extractor.classes
    P32, _ = util_model.oracle_state(6, len(classes), 3, dm)
    model.load_state_dict(P32)
    ex = util_model.make_examples(rng, 2, hw[0], hw[1], n, nums, classes)
    dev = dict(ex)
    for key in ("image", "proposals"):
      dev[key] = torch.from_numpy(ex[key]).to(DEV).contiguous()
    dev["number_of_proposals"] = torch.from_numpy(ex["number_of_proposals"]).to(DEV)
    before = model.store.values.clone()
    for _ in range(2):
      trainer.train_step(dev)
    torch.cuda.synchronize()
    eng, store = model.engine, model.store
    assert not torch.equal(before, store.values)
    assert torch.equal(store.values_bf16, store.values.to(torch.bfloat16))
    checked = 0
    for net in (eng.first, eng.second):
      for L in net.layers.values():
        wt16 = L.wt_for(torch.bfloat16)
        assert torch.equal(wt16, L.wt.to(torch.bfloat16)), L.name
        w = store.var[L.name + "/weights"]
        assert torch.equal(L.wt.reshape(L.k * L.k, L.cout, L.cin),
                           w.reshape(L.k * L.k, L.cin, L.cout).transpose(1, 2)), L.name
        checked += 1
    assert checked > 20
