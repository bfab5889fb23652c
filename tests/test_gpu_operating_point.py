"""The reference's AS-SHIPPED operating point on the HIP path (full depth, real sizes).

configs/voc07_groundtruth.pbtxt:9-23 trains on keep-aspect `min_dimension: 1000` images rescaled
per batch by {1.2, 0.8, 0.6, 0.4} (readers/cap2det_reader.py:143-172) with `batch_size: 2` and
`max_num_proposals: 500`; it evaluates one 1000-px image at 1200 / 800 / 600 / 400 (`:87-90`).
The float64 oracle does not finish these sizes in seconds, so the step is checked through
size-independent properties (as tests/test_gpu_model.py::test_full_size_caption_configs does for
the BASELINE configs) — plus one cross-check that only these sizes can give: at >= 1040 px the
feature map is wider than 64 columns, and the atomic-free row-owner ROI-crop backward must still
be the kernel that runs and must agree with the atomic kernel it used to fall back to."""
import numpy as np
import pytest
import torch

from oracle import ref_labels, ref_model as rm, ref_postprocess as pp
from tests import util_model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _to_dev(ex):
  out = dict(ex)
  for k in ("image", "proposals"):
    out[k] = torch.from_numpy(ex[k]).to(DEV).contiguous()
  out["number_of_proposals"] = torch.from_numpy(ex["number_of_proposals"]).to(DEV)
  return out


@pytest.mark.parametrize("h,w,dtype", [(1000, 1333, "fp32"),     # min_dimension 1000, 4:3 image
                                       (1200, 1600, "fp32"),     # x 1.2: 75 x 100 feature map
                                       (400, 533, "fp32"),       # x 0.4: 25 x 34
                                       (1200, 1600, "bf16")])
def test_train_step_at_the_reference_operating_point(h, w, dtype):
  from cap2det_amd.train.trainer import Trainer
  pipeline = util_model.load_pipeline()
  trainer = Trainer(pipeline, device=DEV, seed=3, compute_dtype=dtype)
  model = trainer.model
  classes = model.label_extractor.classes
  rng = np.random.default_rng(h + w)
  b, n, reals = 2, 500, [500, 430]
  ex = util_model.make_examples(rng, b, h, w, n, reals, classes)
  dev = _to_dev(ex)
  before = model.state_dict()
  m4e = "first_stage_feature_extraction/InceptionV2/Mixed_4e/"

  def step(force_atomic=False):
    model.load_state_dict(before)
    model.store.accum.fill_(0.1)
    trainer.global_step = 0
    model.engine.force_atomic_crop_bwd = force_atomic
    losses = trainer.train_step(dev, dropout_seed=11)
    torch.cuda.synchronize()
    model.engine.force_atomic_crop_bwd = False
    pred = {k: v.detach().clone() for k, v in trainer.predictions.items() if isinstance(v, torch.Tensor)}
    grads = {k: model.store.grad[k].clone() for k in model.store.names()
             if k.startswith(m4e) or k.startswith("second_stage") or "/" in k and k.split("/")[0] in ("midn", "oicr")}
    return {k: float(v.item()) for k, v in losses.items()}, pred, grads, model.engine.last_crop_bwd

  l1, p1, g1, path1 = step()
  after = model.state_dict()
  l2, p2, g2, _ = step()
  fh, fw = -(-h // 16), -(-w // 16)
  bufs = next(iter(model.engine._shape_cache.values()))
  assert (bufs["fh"], bufs["fw"]) == (fh, fw)
  assert path1.startswith("row-owner"), path1            # also on the maps wider than 64 columns
  # forward: no atomics -> bitwise reproducible; per-class proposal softmax per image
  for k in p1:
    assert torch.equal(p1[k], p2[k]), "step not reproducible: " + k
  proba = p1["midn_proba_r_given_c"]
  for bi, real in enumerate(reals):
    np.testing.assert_allclose(proba[bi, :real].sum(0).cpu().numpy(), np.ones(len(classes)),
                               rtol=0, atol=3e-5)
    if real < n:
      assert float(proba[bi, real:].abs().max()) == 0.0
  for i in range(1, 4):
    assert bool(torch.isfinite(p1["oicr_proposal_scores_at_%d" % i]).all())
  # losses follow from the kernels' own scores by the reference formulas (oracle, float64)
  labels = ref_labels.groundtruth_extract(ex["object_texts"], classes).astype(np.float64)
  pred64 = {k: v.double().cpu().numpy() for k, v in p1.items() if v.is_floating_point()}
  pred64["num_proposals"] = ex["number_of_proposals"]
  pred64["proposal_boxes"] = ex["proposals"].astype(np.float64)
  loss_opts = dict(midn_loss_weight=1.0, oicr_loss_weight=0.5, oicr_iterations=3,
                   oicr_iou_threshold=0.6, oicr_use_proba_r_given_c=True)
  want_losses, _ = rm.build_loss(pred64, labels, loss_opts)
  tol = 2e-4 if dtype == "fp32" else 2e-3
  for k, v in want_losses.items():
    np.testing.assert_allclose(l1[k], v, rtol=tol, err_msg=k)
  # gradients: reproducible up to the order of the filter gradients' fp32 atomics
  for k in g1:
    scale = float(g1[k].abs().max())
    assert np.isfinite(scale)
    assert float((g1[k] - g2[k]).abs().max()) <= 3e-5 * scale + 1e-12, k
  assert any(float(g1[k].abs().max()) > 0 for k in g1 if k.startswith(m4e))
  # ... and the Mixed_4e gradients (everything behind the ROI-crop backward) equal the ones the
  # ATOMIC ROI-crop backward gives: the wide-map strips against an independent kernel, at size
  # (fp32 only: the atomic kernel takes fp32 gradients; the bf16 storage mode has the strips alone)
  if dtype == "fp32":
    _, _, g3, path3 = step(force_atomic=True)
    assert path3.startswith("atomic"), path3
    for k in g1:
      if k.startswith(m4e):
        scale = float(g1[k].abs().max())
        assert float((g1[k] - g3[k]).abs().max()) <= 2e-4 * scale + 1e-12, k
  # which variables moved
  moved = [k for k in before if not np.array_equal(before[k], after[k])]
  assert any(k.startswith(m4e) for k in moved) and any(k.startswith("second_stage") for k in moved)
  for k in moved:
    assert (k.startswith(m4e) or k.startswith("second_stage_feature_extraction/") or
            k.startswith("midn/") or k.startswith("oicr/")), "frozen variable moved: " + k


def test_multiscale_inference_at_the_reference_eval_sizes():
  """eval_reader: one keep-aspect 1000-px image, 500 proposals; eval_min_dimension 1200, 800, 600,
  400 (configs/voc07_groundtruth.pbtxt:25-40,87-90; models/cap2det_model.py:236-272).  The
  aggregated scores must be the mean of the four single-scale passes, and the detections the
  oracle NMS of those scores."""
  from cap2det_amd import hip_ops as ops
  from cap2det_amd.models import builder
  from cap2det_amd.models.cap2det_model import resize_to_min_dimension_size
  pipeline = util_model.load_pipeline()
  model = builder.build(pipeline.model, is_training=False, device=DEV, seed=2)
  assert list(model._model_proto.eval_min_dimension) == [1200, 800, 600, 400]
  classes = model.label_extractor.classes
  # (the sigma = 0.01 initial heads give near-uniform scores: spread them so that the NMS order
  # is not decided by the last bit)
  state = model.state_dict()
  for k in state:
    if k.endswith("/weights") and k.split("/")[0] in ("midn", "oicr"):
      state[k] = state[k] * 30.0
  model.load_state_dict(state)
  rng = np.random.default_rng(77)
  n, real, h, w = 500, 480, 1000, 1333
  ex = util_model.make_examples(rng, 1, h, w, n, [real], classes)
  dev = _to_dev(ex)
  pred = model.build_prediction(dev)
  torch.cuda.synchronize()
  got = [pred["oicr_proposal_scores_at_%d" % i].clone() for i in range(4)]
  again = model.build_prediction(dev)
  for i in range(4):
    assert torch.equal(got[i], again["oicr_proposal_scores_at_%d" % i])
  sums = None
  for md in (1200, 800, 600, 400):
    oh, ow = resize_to_min_dimension_size(h, w, md)
    assert min(oh, ow) == md
    one = dict(dev)
    one["image"] = ops.resize_bilinear(dev["image"][0].contiguous(), oh, ow).unsqueeze(0)
    p = model.build_prediction(one, single_scale=True)
    cur = [p["oicr_proposal_scores_at_%d" % i].double() for i in range(4)]
    sums = cur if sums is None else [a + c for a, c in zip(sums, cur)]
  for i in range(4):
    want = (sums[i] / 4.0).cpu().numpy()
    # (iterations 1..K average logits of magnitude ~10: fp32 sums of four of them)
    assert np.abs(got[i].cpu().numpy() - want).max() <= 1e-6 * max(1.0, float(np.abs(want).max())), i
    assert bool(torch.isfinite(got[i]).all())
    assert float(got[i][0, real:].abs().max()) == 0.0 or i > 0
  mid = dict(score_thresh=1e-5, iou_thresh=0.4, max_size_per_class=100, max_total_size=300)
  oic = dict(mid, iou_thresh=0.3)
  for i in range(4):
    s = got[i].cpu().numpy()
    if i > 0:
      # the kernel's own fp32 probabilities (c2d_softmax_drop_background is oracle-tested in
      # tests/test_gpu_postprocess.py): the NMS decisions are then integer work on equal inputs
      probs = torch.empty(1, n, len(classes), device=DEV)
      ops.softmax_drop_background(got[i].contiguous(), len(classes) + 1, 0, n, len(classes) + 1, probs)
      np.testing.assert_allclose(probs.cpu().numpy(), pp.softmax_drop_background(s.astype(np.float64)),
                                 rtol=2e-5, atol=1e-7)
      s = probs.cpu().numpy()
    num, bx, sc, cl = pp.batch_multiclass_nms(ex["proposals"], s, **(mid if i == 0 else oic))
    np.testing.assert_array_equal(pred["num_detections_at_%d" % i].cpu().numpy(), num)
    np.testing.assert_array_equal(pred["detection_classes_at_%d" % i].cpu().numpy(), cl)
    np.testing.assert_array_equal(pred["detection_boxes_at_%d" % i].cpu().numpy(), bx)
