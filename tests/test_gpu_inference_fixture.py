"""Multi-scale inference at the reference's evaluation sizes against the float64 oracle
(tests/golden/inference_375x500.npz, tests/golden/gen_inference_fixture.py): a 375x500 image with
2000 proposals at depth 1.0, one forward pass per eval_min_dimension 1200 / 800 / 600 / 400
(/root/reference/configs/voc07_groundtruth.pbtxt:87-90), the proposal scores of every OICR
iteration averaged over the four resolutions (/root/reference/models/cap2det_model.py:236-272).
The HIP path — legacy-bilinear resize (c2d_resize_bilinear), first stage on 600x800 ... 200x267
stems, ROI crop of 2000 boxes on 75x100 ... 25x34 maps, second stage, heads, c2d_scores_accumulate /
divide — must give the same averaged scores within the north-star 1e-4, and the detections it
reports must be the oracle's NMS of those scores (integer work: exact)."""
import os

import numpy as np
import pytest
import torch

from oracle import ref_postprocess as pp
from tests.golden import gen_inference_fixture as gen
from tests import util_model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_multiscale_inference_replays_the_float64_fixture():
  from cap2det_amd.models import builder
  fix = np.load(os.path.join(ROOT, "tests", "golden", "inference_%dx%d.npz" % (gen.H, gen.W)))
  ex, P32 = gen.inputs()
  np.testing.assert_allclose(gen.checksum(ex, P32), fix["checksum"], rtol=1e-12)
  pipeline = util_model.load_pipeline()
  model = builder.build(pipeline.model, is_training=False, device=DEV, depth_multiplier=gen.DM)
  opt = model._model_proto
  del opt.eval_min_dimension[:]
  opt.eval_min_dimension.extend(gen.EVAL_DIMS)
  model.load_state_dict(P32)
  dev = dict(ex)
  for key in ("image", "proposals", "number_of_proposals"):
    dev[key] = torch.from_numpy(ex[key]).to(DEV).contiguous()
  pred = model.build_prediction(dev)
  torch.cuda.synchronize()
  worst = {}
  for i in range(gen.K + 1):
    got = pred["oicr_proposal_scores_at_%d" % i].cpu().numpy().astype(np.float64)
    want = fix["scores_%d" % i].astype(np.float64)
    err = np.abs(got - want)
    worst[i] = float(err.max())
    assert err.max() <= 1e-4, (i, err.max())                       # north-star tolerance
    colmax = np.abs(want).max(axis=1, keepdims=True)
    assert (err <= 1e-3 * colmax + 1e-9).all(), i                  # and relative to the class maximum
  print("inference fixture, max abs score error per iteration:", worst)
  # the detections are the oracle's NMS of the HIP path's own scores (post-processing options of the
  # pipeline: models/cap2det_model.py:236-272, core/builder.py:15-67)
  mid = dict(score_thresh=1e-5, iou_thresh=0.4, max_size_per_class=100, max_total_size=300)
  oic = dict(mid, iou_thresh=0.3)
  for i in range(gen.K + 1):
    s = pred["oicr_proposal_scores_at_%d" % i].cpu().numpy()
    if i > 0:
      s = pp.softmax_drop_background(s.astype(np.float64)).astype(np.float32)
    num, b, sc, cl = pp.batch_multiclass_nms(ex["proposals"], s, **(mid if i == 0 else oic))
    np.testing.assert_array_equal(pred["num_detections_at_%d" % i].cpu().numpy(), num)
    np.testing.assert_array_equal(pred["detection_classes_at_%d" % i].cpu().numpy(), cl)
    np.testing.assert_array_equal(pred["detection_boxes_at_%d" % i].cpu().numpy(), b)
    assert num[0] > 0
