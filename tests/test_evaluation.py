"""Evaluator of SURVEY.md §8f row f3 (cap2det_amd/train/evaluation.py): hand-computed known
answers for the restated PASCAL protocol, an independent brute-force AP, CorLoc, the COCO->VOC
class remap of train/predict.py:284-325 and the best-checkpoint bookkeeping of
core/training_utils.py:233-308."""
import os

import numpy as np
import pytest

from cap2det_amd.train import evaluation as ev


def _ap_bruteforce(scores, tps, num_gt):
  """AP = integral over recall of max precision at recall >= r (definition)."""
  order = np.argsort(-np.asarray(scores), kind="stable")
  t = np.asarray(tps, float)[order]
  tp, fp = np.cumsum(t), np.cumsum(1 - t)
  prec, rec = tp / (tp + fp), tp / num_gt
  ap, prev = 0.0, 0.0
  for r in sorted(set(rec)):
    if r <= 0:
      continue
    ap += (r - prev) * prec[rec >= r].max()
    prev = r
  return ap


def test_average_precision_known_answers():
  # 3 ground truths; detections by score: TP FP TP FP TP -> P = 1, 1/2, 2/3, 2/4, 3/5
  prec = np.array([1, 0.5, 2 / 3, 0.5, 0.6]); rec = np.array([1, 1, 2, 2, 3]) / 3.0
  ap = ev.compute_average_precision(prec, rec)
  np.testing.assert_allclose(ap, (1 + 2 / 3 + 0.6) / 3)
  np.testing.assert_allclose(ap, _ap_bruteforce([5, 4, 3, 2, 1], [1, 0, 1, 0, 1], 3))
  assert ev.compute_average_precision(np.array([1.0]), np.array([1.0])) == 1.0
  assert ev.compute_average_precision(np.array([0.0, 0.0]), np.array([0.0, 0.0])) == 0.0
  assert np.isnan(ev.compute_average_precision(np.array([]), np.array([])))
  rng = np.random.default_rng(0)
  for _ in range(20):
    n = int(rng.integers(1, 30))
    s, t = rng.uniform(size=n), rng.integers(0, 2, n)
    g = int(t.sum() + rng.integers(0, 3)) or 1
    order = np.argsort(-s, kind="stable")
    tp, fp = np.cumsum(t[order]), np.cumsum(1 - t[order])
    got = ev.compute_average_precision(tp / (tp + fp), tp / g)
    np.testing.assert_allclose(got, _ap_bruteforce(s, t, g), atol=1e-12)


def test_pascal_evaluator_matching_rules():
  cats = [{'id': 1, 'name': 'cat'}, {'id': 2, 'name': 'dog'}, {'id': 3, 'name': 'bird'}]
  e = ev.PascalDetectionEvaluator(cats)
  # image A: two cats, one dog
  e.add_single_ground_truth_image_info('A', {
      'groundtruth_boxes': np.array([[0, 0, 10, 10], [20, 20, 30, 30], [0, 50, 10, 60.]]),
      'groundtruth_classes': np.array([1, 1, 2])})
  e.add_single_detected_image_info('A', {
      'detection_boxes': np.array([[0, 0, 10, 9.], [0, 0, 10, 10], [21, 21, 30, 30], [40, 40, 50, 50],
                                   [0, 50, 10, 60.], [0, 0, 10, 10]]),
      'detection_scores': np.array([0.9, 0.8, 0.7, 0.6, 0.3, 0.95]),
      'detection_classes': np.array([1, 1, 1, 1, 2, 2])})
  # image B: one cat, missed; a dog false alarm on an image without dogs
  e.add_single_ground_truth_image_info('B', {'groundtruth_boxes': np.array([[5, 5, 15, 15.]]),
                                             'groundtruth_classes': np.array([1])})
  e.add_single_detected_image_info('B', {'detection_boxes': np.array([[50, 50, 60, 60.]]),
                                         'detection_scores': np.array([0.5]),
                                         'detection_classes': np.array([2])})
  m = e.evaluate()
  # cat: scores .9 TP (IoU .9), .8 FP (duplicate of the claimed box), .7 TP (IoU .81), .6 FP; 3 gts
  want_cat = _ap_bruteforce([0.9, 0.8, 0.7, 0.6], [1, 0, 1, 0], 3)
  np.testing.assert_allclose(m['PascalBoxes_PerformanceByCategory/AP@0.5IOU/cat'], want_cat)
  np.testing.assert_allclose(want_cat, (1.0 + 2 / 3) / 3)
  # dog: .95 FP (on the cat box, IoU 0 with the dog), .5 FP (image B), .3 TP; 1 gt -> P@R=1 = 1/3
  np.testing.assert_allclose(m['PascalBoxes_PerformanceByCategory/AP@0.5IOU/dog'], 1 / 3)
  assert np.isnan(m['PascalBoxes_PerformanceByCategory/AP@0.5IOU/bird'])     # no ground truth
  np.testing.assert_allclose(m['PascalBoxes_Precision/mAP@0.5IOU'], (want_cat + 1 / 3) / 2)
  # CorLoc: cat hit in A (top detection), missed in B -> 1/2; dog: top detection misses -> 0
  np.testing.assert_allclose(m['PascalBoxes_PerformanceByCategory/CorLoc@0.5IOU/cat'], 0.5)
  np.testing.assert_allclose(m['PascalBoxes_PerformanceByCategory/CorLoc@0.5IOU/dog'], 0.0)
  np.testing.assert_allclose(m['PascalBoxes_Precision/meanCorLoc@0.5IOU'], 0.25)
  # IoU exactly at the threshold counts as a hit (>=); difficult boxes are ignored
  e.clear()
  e.add_single_ground_truth_image_info('C', {
      'groundtruth_boxes': np.array([[0, 0, 10, 10.], [20, 20, 30, 30.]]),
      'groundtruth_classes': np.array([1, 1]), 'groundtruth_difficult': np.array([False, True])})
  e.add_single_detected_image_info('C', {
      'detection_boxes': np.array([[0, 0, 10, 5.], [20, 20, 30, 30.]]),
      'detection_scores': np.array([0.5, 0.9]), 'detection_classes': np.array([1, 1])})
  m = e.evaluate()
  np.testing.assert_allclose(m['PascalBoxes_PerformanceByCategory/AP@0.5IOU/cat'], 1.0)


def test_coco_to_voc_and_coordinates():
  boxes = np.arange(16, dtype=np.float64).reshape(4, 4)
  b, s, c = ev.convert_coco_result_to_voc(boxes, np.array([.9, .8, .7, .6]), np.array([1, 8, 63, 5.]))
  np.testing.assert_array_equal(c, [15, 20, 1])                 # person, tv, aeroplane
  np.testing.assert_array_equal(b, boxes[[0, 2, 3]]); np.testing.assert_array_equal(s, [.9, .7, .6])
  b, s, c = ev.convert_coco_result_to_voc(boxes[:1], np.array([.9]), np.array([8]))
  assert b.shape == (0, 4) and s.shape == (0,) and c.dtype == np.int64
  np.testing.assert_array_equal(ev.py_coord_norm_to_abs(np.array([[0.1, 0.2, 0.5, 1.0]]), 100, 50),
                                [[10, 10, 50, 50]])


def test_best_checkpoint_bookkeeping(tmp_path):
  src, dst = tmp_path / "model_dir", str(tmp_path / "saved")
  src.mkdir()
  for step in (100, 200, 300):
    (src / ("model.ckpt-%d.npz" % step)).write_bytes(b"w%d" % step)
  path = lambda s: str(src / ("model.ckpt-%d" % s))
  assert ev.save_model_if_it_is_better(100, 0.30, path(100), dst) == (100, 0.30)
  assert ev.save_model_if_it_is_better(200, 0.25, path(200), dst) == (100, 0.30)   # worse: kept
  assert os.listdir(dst).count("model.ckpt-100.npz") == 1
  assert ev.save_model_if_it_is_better(300, 0.41, path(300), dst) == (300, 0.41)
  assert sorted(os.listdir(dst)) == ["model.ckpt-300.npz", "saved_info.txt"]
  assert open(os.path.join(dst, "saved_info.txt")).read() == "300\t0.41000000"
  assert ev.get_best_model_checkpoint(dst).endswith("model.ckpt-300")
  assert ev.save_model_if_it_is_better(400, 0.1, path(300), dst, reverse=True) == (400, 0.1)


def test_coco_evaluator_hand_computed_cases():
  """pycocotools' bbox protocol (train/predict.py:570-573 instantiates CocoDetectionEvaluator):
  AP over IoU 0.50:0.05:0.95 at 101 recall points, AR@1/10/100, area ranges."""
  from cap2det_amd.train.evaluation import CocoDetectionEvaluator, build_evaluators
  cats = [{'id': 1, 'name': 'a'}, {'id': 2, 'name': 'b'}, {'id': 3, 'name': 'never'}]

  # (1) a perfect detection of one large box: everything 1, empty area ranges -1
  ev = CocoDetectionEvaluator(cats)
  ev.add_single_ground_truth_image_info('i', {'groundtruth_boxes': [[0, 0, 100, 100]], 'groundtruth_classes': [1]})
  ev.add_single_detected_image_info('i', {'detection_boxes': [[0, 0, 100, 100]], 'detection_scores': [0.9], 'detection_classes': [1]})
  m = ev.evaluate()
  assert m['DetectionBoxes_Precision/mAP'] == pytest.approx(1.0)
  assert m['DetectionBoxes_Precision/mAP (large)'] == pytest.approx(1.0)
  assert m['DetectionBoxes_Precision/mAP (small)'] == -1.0 and m['DetectionBoxes_Precision/mAP (medium)'] == -1.0
  assert m['DetectionBoxes_Recall/AR@1'] == pytest.approx(1.0)

  # (2) IoU 0.78: a hit for the six thresholds 0.50 .. 0.75 only
  ev = CocoDetectionEvaluator(cats)
  ev.add_single_ground_truth_image_info('i', {'groundtruth_boxes': [[0, 0, 100, 100]], 'groundtruth_classes': [2]})
  ev.add_single_detected_image_info('i', {'detection_boxes': [[0, 0, 100, 78]], 'detection_scores': [0.9], 'detection_classes': [2]})
  m = ev.evaluate()
  assert m['DetectionBoxes_Precision/mAP'] == pytest.approx(0.6)
  assert m['DetectionBoxes_Precision/mAP@.50IOU'] == pytest.approx(1.0)
  assert m['DetectionBoxes_Precision/mAP@.75IOU'] == pytest.approx(1.0)
  assert m['DetectionBoxes_Recall/AR@100'] == pytest.approx(0.6)

  # (3) TP, FP, TP over two images (merged by score) + a category that only has detections
  ev = build_evaluators('coco', cats, 1)[0]
  ev.add_single_ground_truth_image_info('i', {'groundtruth_boxes': [[0, 0, 50, 50]], 'groundtruth_classes': [1]})
  ev.add_single_ground_truth_image_info('j', {'groundtruth_boxes': [[10, 10, 90, 60]], 'groundtruth_classes': [1]})
  ev.add_single_detected_image_info('i', {'detection_boxes': [[0, 0, 50, 50], [200, 200, 260, 260]],
                                          'detection_scores': [0.9, 0.8], 'detection_classes': [1, 1]})
  ev.add_single_detected_image_info('j', {'detection_boxes': [[10, 10, 90, 60], [0, 0, 5, 5]],
                                          'detection_scores': [0.7, 0.95], 'detection_classes': [1, 2]})
  m = ev.evaluate()
  want = (51 * 1.0 + 50 * (2.0 / 3.0)) / 101
  assert m['DetectionBoxes_Precision/mAP'] == pytest.approx(want, abs=1e-9)
  assert m['DetectionBoxes_Precision/mAP@.50IOU'] == pytest.approx(want, abs=1e-9)
  assert m['DetectionBoxes_Recall/AR@1'] == pytest.approx(1.0)     # per IMAGE: each image's best detection hits
  assert m['DetectionBoxes_Recall/AR@100'] == pytest.approx(1.0)
  # gt areas 2500 / 4000: medium; the 60x60 false positive is medium too, the 5x5 one has no category gt
  assert m['DetectionBoxes_Precision/mAP (medium)'] == pytest.approx(want, abs=1e-9)
  assert m['DetectionBoxes_Precision/mAP (large)'] == -1.0

  # (4) a detection never trades a non-ignored match for an ignored (out-of-range) one, and one
  # ground truth is claimed once: the second, lower-scoring duplicate is a false positive
  ev = CocoDetectionEvaluator(cats[:1])
  ev.add_single_ground_truth_image_info('i', {'groundtruth_boxes': [[0, 0, 40, 40]], 'groundtruth_classes': [1]})
  ev.add_single_detected_image_info('i', {'detection_boxes': [[0, 0, 40, 40], [0, 0, 40, 38]],
                                          'detection_scores': [0.5, 0.9], 'detection_classes': [1, 1]})
  m = ev.evaluate()
  # at t <= 0.95 the 0.9-score box (IoU 0.95) claims the gt first -> [TP, FP]; AP = 1 at those t
  assert m['DetectionBoxes_Precision/mAP@.50IOU'] == pytest.approx(1.0)
  with pytest.raises(ValueError):
    build_evaluators('nope', cats, 1)
