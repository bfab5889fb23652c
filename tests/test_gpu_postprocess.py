"""GPU parity of the inference post-processing against
oracle/ref_postprocess.py, through the C-ABI: multi-class NMS (integer / index work: exact),
softmax without background, legacy bilinear resize (bit-exact: same fp32 operation order), the
reference-shaped `build_post_processor` callable, and the multi-scale `build_prediction` of the
model in evaluation mode."""
import numpy as np
import pytest
import torch

from oracle import ref_labels, ref_model, ref_postprocess as pp
from tests import util_model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
  from cap2det_amd import hip_ops
  return hip_ops


def _t(a):
  return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _boxes(rng, b, n, degenerate=True):
  bx = np.stack([util_model.synthetic_boxes(rng, n) for _ in range(b)])
  if degenerate and n >= 8:
    bx[:, 1] = bx[:, 0]                       # duplicates (IoU 1)
    bx[:, 2] = [0.2, 0.2, 0.2, 0.7]           # zero area
    bx[:, 3] = 0.0                            # zero-padded proposal
    bx[:, 4] = bx[:, 5][:, [2, 3, 0, 1]]      # flipped corners of box 5
  return bx


@pytest.mark.parametrize("b,n,c,thr,iou,mpc,mtot", [
    (1, 1, 1, 0.0, 0.5, 100, 300), (2, 37, 5, 0.3, 0.4, 100, 300), (1, 130, 20, 1e-5, 0.3, 7, 50),
    (1, 600, 4, 1e-5, 0.4, 100, 300), (2, 64, 3, 0.99, 0.5, 100, 10), (1, 300, 80, 0.5, 0.3, 100, 300)])
def test_multiclass_nms_matches_oracle(ops, b, n, c, thr, iou, mpc, mtot):
  rng = np.random.default_rng(100 + n)
  boxes = _boxes(rng, b, n)
  scores = rng.uniform(0, 1, (b, n, c)).astype(np.float32)
  if n >= 8:
    scores[:, 6] = scores[:, 7]               # exact score ties (broken by index)
    scores[:, :, 0] = np.round(scores[:, :, 0], 1)   # many ties in class 0
  want = pp.batch_multiclass_nms(boxes, scores, score_thresh=thr, iou_thresh=iou,
                                 max_size_per_class=mpc, max_total_size=mtot)
  # class columns inside a wider buffer (the model hands over slices of the fused logits)
  wide = np.full((b, n, c + 5), 7.0, np.float32); wide[:, :, 2:2 + c] = scores
  got = ops.multiclass_nms(_t(boxes), _t(wide), c + 5, 2, c, thr, iou, mpc, mtot)
  torch.cuda.synchronize()
  np.testing.assert_array_equal(got[0].cpu().numpy(), want[0])
  np.testing.assert_array_equal(got[3].cpu().numpy(), want[3])     # classes
  np.testing.assert_array_equal(got[2].cpu().numpy(), want[2])     # scores (copied values)
  np.testing.assert_array_equal(got[1].cpu().numpy(), want[1])     # boxes (copied values)


def test_post_processor_builder_and_softmax(ops):
  from cap2det_amd.core import builder
  from cap2det_amd.protos import pipeline_pb2, post_process_pb2, text_format
  opt = post_process_pb2.PostProcess()
  text_format.Merge("score_thresh: 0.05 iou_thresh: 0.3 max_size_per_class: 4 max_total_size: 9", opt)
  fn = builder.build_post_processor(opt)
  with pytest.raises(ValueError):
    builder.build_post_processor(pipeline_pb2.Pipeline())          # core/builder.py:27-29
  assert post_process_pb2.PostProcess().score_thresh == pytest.approx(1e-6)   # proto defaults
  assert post_process_pb2.PostProcess().max_total_size == 300
  rng = np.random.default_rng(3)
  boxes = _boxes(rng, 2, 50)
  logits = rng.standard_normal((2, 50, 7)).astype(np.float32) * 2
  probs = torch.empty(2, 50, 6, device=DEV)
  ops.softmax_drop_background(_t(logits), 7, 0, 100, 7, probs)
  want_p = pp.softmax_drop_background(logits.astype(np.float64))
  np.testing.assert_allclose(probs.cpu().numpy(), want_p, rtol=2e-6, atol=1e-7)
  num, nb, ns, nc, extra = fn(_t(boxes), probs)
  want = pp.batch_multiclass_nms(boxes, probs.cpu().numpy(), score_thresh=0.05, iou_thresh=0.3,
                                 max_size_per_class=4, max_total_size=9)
  assert extra is None
  np.testing.assert_array_equal(num.cpu().numpy(), want[0])
  np.testing.assert_array_equal(nc.cpu().numpy(), want[3])
  np.testing.assert_array_equal(nb.cpu().numpy(), want[1])


@pytest.mark.parametrize("ih,iw,oh,ow", [(5, 7, 10, 14), (33, 50, 40, 61), (64, 48, 20, 15),
                                         (17, 9, 17, 9), (3, 4, 1, 1)])
def test_resize_bilinear_is_bit_exact(ops, ih, iw, oh, ow):
  rng = np.random.default_rng(ih * 100 + iw)
  img = rng.uniform(0, 255, (ih, iw, 3)).astype(np.float32)
  got = ops.resize_bilinear(_t(img), oh, ow).cpu().numpy()
  np.testing.assert_array_equal(got, pp.resize_bilinear_legacy(img, oh, ow))


def test_scores_mean(ops):
  rng = np.random.default_rng(5)
  parts = [rng.standard_normal((33, 9)).astype(np.float32) for _ in range(3)]
  acc = torch.empty(33, 6, device=DEV)
  for i, p in enumerate(parts):
    ops.scores_accumulate(acc, _t(p), 9, 2, 33, 6, i == 0)
  ops.scores_divide(acc, 3.0)
  want = ((parts[0][:, 2:8] + parts[1][:, 2:8]) + parts[2][:, 2:8]) / np.float32(3.0)
  np.testing.assert_array_equal(acc.cpu().numpy(), want)


def test_multiscale_inference_matches_oracle():
  """Model in evaluation mode (models/cap2det_model.py:236-272): one forward per
  eval_min_dimension on the legacy-bilinear resized image, scores averaged over the resolutions,
  softmax + NMS.  Oracle: float64 restatement of the same pipeline at small sizes."""
  from cap2det_amd.models import builder
  pipeline = util_model.load_pipeline()
  rng = np.random.default_rng(21)
  dm, n = 0.5, 9
  model = builder.build(pipeline.model, is_training=False, device=DEV, depth_multiplier=dm)
  opt = model._model_proto
  del opt.eval_min_dimension[:]
  opt.eval_min_dimension.extend([48, 40, 33])
  classes = model.label_extractor.classes
  c, k = len(classes), 3
  P32, d = util_model.oracle_state(5, c, k, dm, head_std=0.3)
  model.load_state_dict(P32)
  ex = util_model.make_examples(rng, 1, 36, 44, n, [n], classes)
  dev = dict(ex)
  for key in ("image", "proposals"):
    dev[key] = _t(ex[key])
  dev["number_of_proposals"] = _t(ex["number_of_proposals"])
  pred = model.build_prediction(dev)
  torch.cuda.synchronize()

  P = {kk: v.astype(np.float64) for kk, v in P32.items()}
  opts = ref_model.FrcnnOptions(depth_multiplier=dm)
  sums = None
  for md in (48, 40, 33):
    img = pp.resize_image_to_min_dimension(ex["image"][0], md)           # fp32, as the kernel
    oh, ow = pp.min_dimension_size(36, 44, md)
    assert img.shape[:2] == (oh, ow)
    e64 = dict(image=img[None].astype(np.float64), number_of_proposals=ex["number_of_proposals"],
               proposals=ex["proposals"].astype(np.float64))
    wp = ref_model.build_prediction(e64, P, opts, k, is_training=False)[0]
    cur = [wp["oicr_proposal_scores_at_%d" % i] for i in range(k + 1)]
    sums = cur if sums is None else [a + b_ for a, b_ in zip(sums, cur)]
  want_scores = [s / 3.0 for s in sums]
  for i in range(k + 1):
    got = pred["oicr_proposal_scores_at_%d" % i].cpu().numpy()
    assert np.abs(got - want_scores[i]).max() <= 1e-4, i                 # north-star tolerance
  # post-processing of the GPU's own aggregated scores must equal the oracle's NMS of them
  mid = dict(score_thresh=1e-5, iou_thresh=0.4, max_size_per_class=100, max_total_size=300)
  oic = dict(mid, iou_thresh=0.3)
  got_scores = [pred["oicr_proposal_scores_at_%d" % i].cpu().numpy() for i in range(k + 1)]
  for i in range(k + 1):
    s = got_scores[i]
    if i > 0:
      s = pp.softmax_drop_background(s.astype(np.float64)).astype(np.float32)
    num, b, sc, cl = pp.batch_multiclass_nms(ex["proposals"], s, **(mid if i == 0 else oic))
    np.testing.assert_array_equal(pred["num_detections_at_%d" % i].cpu().numpy(), num)
    np.testing.assert_array_equal(pred["detection_classes_at_%d" % i].cpu().numpy(), cl)
    np.testing.assert_array_equal(pred["detection_boxes_at_%d" % i].cpu().numpy(), b)
    np.testing.assert_allclose(pred["detection_scores_at_%d" % i].cpu().numpy(), sc, rtol=1e-5,
                               atol=1e-7)
    assert num[0] > 0


def test_multiscale_inference_bf16_tracks_fp32():
  """Evaluation mode with compute_dtype="bf16" (both towers in bf16 storage): the scores averaged
  over three resolutions stay within 5 % of the tensor's maximum of the fp32 model's on the same
  weights, and the post-processing of the model's OWN scores equals the oracle's NMS of them
  exactly (labels, boxes, order)."""
  from cap2det_amd.models import builder
  pipeline = util_model.load_pipeline()
  rng = np.random.default_rng(22)
  dm, n = 0.5, 9
  ex, preds = None, {}
  for dtype in ("fp32", "bf16"):
    model = builder.build(pipeline.model, is_training=False, device=DEV, depth_multiplier=dm,
                          compute_dtype=dtype)
    opt = model._model_proto
    del opt.eval_min_dimension[:]
    opt.eval_min_dimension.extend([48, 40, 33])
    classes = model.label_extractor.classes
    c, k = len(classes), 3
    P32, d = util_model.oracle_state(5, c, k, dm, head_std=0.3)
    model.load_state_dict(P32)
    if ex is None:
      ex = util_model.make_examples(rng, 1, 36, 44, n, [n], classes)
    dev = dict(ex)
    for key in ("image", "proposals"):
      dev[key] = _t(ex[key])
    dev["number_of_proposals"] = _t(ex["number_of_proposals"])
    pred = model.build_prediction(dev)
    torch.cuda.synchronize()
    preds[dtype] = {kk: v.detach().float().cpu().numpy() if v.dtype.is_floating_point else v.cpu().numpy()
                    for kk, v in pred.items() if torch.is_tensor(v)}
    if dtype == "bf16":
      assert model.engine.first.dtype == torch.bfloat16 and model.engine.second.dtype == torch.bfloat16
  k = 3
  for i in range(k + 1):
    a, b = preds["fp32"]["oicr_proposal_scores_at_%d" % i], preds["bf16"]["oicr_proposal_scores_at_%d" % i]
    assert np.abs(a - b).max() <= 5e-2 * np.abs(a).max(), (i, np.abs(a - b).max(), np.abs(a).max())
  mid = dict(score_thresh=1e-5, iou_thresh=0.4, max_size_per_class=100, max_total_size=300)
  oic = dict(mid, iou_thresh=0.3)
  p16 = preds["bf16"]
  for i in range(k + 1):
    s = p16["oicr_proposal_scores_at_%d" % i]
    if i > 0:
      s = pp.softmax_drop_background(s.astype(np.float64)).astype(np.float32)
    num, b, sc, cl = pp.batch_multiclass_nms(ex["proposals"], s, **(mid if i == 0 else oic))
    np.testing.assert_array_equal(p16["num_detections_at_%d" % i], num)
    np.testing.assert_array_equal(p16["detection_classes_at_%d" % i], cl)
    np.testing.assert_array_equal(p16["detection_boxes_at_%d" % i], b)
    np.testing.assert_allclose(p16["detection_scores_at_%d" % i], sc, rtol=1e-5, atol=1e-7)
