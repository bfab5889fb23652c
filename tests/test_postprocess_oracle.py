"""CPU tests of the post-processing oracle (oracle/ref_postprocess.py): hand-computed known
answers for the restated NMS / resize semantics (the reference holds no test for them: parity
unpinned, see the oracle header)."""
import numpy as np

from oracle import ref_postprocess as pp


def test_iou_matches_box_utils_known_answers():
  # the reference's own IoU known answers (core/box_utils_test.py:83-107, transcribed in
  # tests/golden/box_utils_known_answers.json) must also hold for the NMS kernel's IoU form
  import json, os
  path = os.path.join(os.path.dirname(__file__), "golden", "box_utils_known_answers.json")
  case = [c for c in json.load(open(path))["cases"] if c["fn"] == "iou"][0]
  b1 = np.asarray(case["box1"], np.float32)
  b2 = np.asarray(case["box2"], np.float32)
  for k, want in enumerate(case["expected"]):
    got = pp.iou_tf(np.stack([b1[k], b2[k]]), 0, 1)
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-7)


def test_nms_hand_case():
  boxes = np.array([[0, 0, 1, 1], [0, 0, 1, 0.9], [0, 0.5, 1, 1.5], [2, 2, 3, 3], [0, 0, 0, 0]],
                   np.float32)
  scores = np.array([0.9, 0.8, 0.7, 0.6, 0.95], np.float32)
  # box 4 is degenerate (area 0): IoU 0 with everything, so it is kept first; box 1 overlaps box
  # 0 with IoU 0.9 > 0.5; box 2 overlaps box 0 with IoU 1/3
  assert pp.non_max_suppression(boxes, scores, 10, np.float32(0.5), np.float32(0.0)) == [4, 0, 2, 3]
  assert pp.non_max_suppression(boxes, scores, 2, np.float32(0.5), np.float32(0.0)) == [4, 0]
  assert pp.non_max_suppression(boxes, scores, 10, np.float32(0.3), np.float32(0.65)) == [4, 0]
  # threshold is strict on both tests: score == thresh is dropped, iou == thresh is kept
  assert pp.non_max_suppression(boxes[:2], np.array([0.5, 0.5], np.float32), 10, np.float32(0.9),
                                np.float32(0.5)) == []
  b = np.array([[0, 0, 1, 1], [0, 0, 1, 0.5]], np.float32)       # IoU exactly 0.5
  assert pp.non_max_suppression(b, np.array([0.9, 0.8], np.float32), 10, np.float32(0.5),
                                np.float32(0.0)) == [0, 1]
  # ties: lower index first; flipped corners are re-ordered
  b = np.array([[1, 1, 0, 0], [0, 0, 1, 1]], np.float32)
  assert pp.non_max_suppression(b, np.array([0.5, 0.5], np.float32), 10, np.float32(0.5),
                                np.float32(0.0)) == [0]


def test_multiclass_nms_merge_and_padding():
  boxes = np.array([[0, 0, 1, 1], [0, 0, 1, 0.9], [2, 2, 3, 3]], np.float32)
  scores = np.array([[0.9, 0.1], [0.8, 0.85], [0.3, 0.85]], np.float32)
  num, b, s, c = pp.multiclass_nms(boxes, scores, 0.2, 0.5, 100, 4)
  # class 0 keeps boxes 0, 2 (box 1 suppressed); class 1 keeps boxes 1, 2 (tie: index order)
  assert num == 4
  np.testing.assert_array_equal(c, [1, 2, 2, 1])
  np.testing.assert_allclose(s, [0.9, 0.85, 0.85, 0.3])
  np.testing.assert_array_equal(b[1], boxes[1]); np.testing.assert_array_equal(b[2], boxes[2])
  num, b, s, c = pp.multiclass_nms(boxes, scores, 0.2, 0.5, 100, 6)
  assert num == 4 and np.all(s[4:] == 0) and np.all(c[4:] == 0) and np.all(b[4:] == 0)
  num, b, s, c = pp.multiclass_nms(boxes, scores, 0.2, 0.5, 1, 6)      # one per class
  assert num == 2 and list(c[:2]) == [1, 2]
  num, _, _, _ = pp.multiclass_nms(boxes, scores, 0.95, 0.5, 100, 6)   # nothing above threshold
  assert num == 0


def test_resize_legacy_bilinear_known_answers():
  img = np.arange(12, dtype=np.float32).reshape(3, 4, 1)
  same = pp.resize_bilinear_legacy(img, 3, 4)
  np.testing.assert_array_equal(same, img)
  up = pp.resize_bilinear_legacy(img, 6, 8)
  # legacy scaler: out (y, x) samples in (y/2, x/2); the last row/column replicate the edge
  assert up[0, 1, 0] == 0.5 and up[1, 0, 0] == 2.0 and up[5, 7, 0] == 11.0 and up[4, 6, 0] == 11.0
  down = pp.resize_bilinear_legacy(img, 2, 2)
  np.testing.assert_allclose(down[..., 0], [[0.0, 2.0], [6.0, 8.0]])
  assert pp.min_dimension_size(375, 500, 600) == (600, 800)
  assert pp.min_dimension_size(333, 500, 400) == (400, 601)    # 500 * (400/333) = 600.6
  assert pp.min_dimension_size(3, 5, 2) == (2, 3)              # 5 * (2/3) = 3.33


def test_softmax_drop_background():
  x = np.log(np.array([[1.0, 2.0, 5.0]]))
  np.testing.assert_allclose(pp.softmax_drop_background(x), [[0.25, 0.625]])
