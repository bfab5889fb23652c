"""Detection evaluation loop (reference: train/predict.py:284-325,328-529,
583-611 and core/training_utils.py:233-308).

The reference feeds the NMS outputs of every OICR iteration into object_detection's
`PascalDetectionEvaluator` (third party, not vendored) and keeps the best checkpoint by
mAP@0.5.  `PascalDetectionEvaluator` below restates that evaluator's published protocol:

  * per image and class, detections are visited by decreasing score; a detection is matched to
    the ground-truth box of the same class with which it has the highest IoU; it is a true
    positive when that IoU >= 0.5 and the box has not been claimed by a higher-scoring
    detection, otherwise a false positive (detections whose best box is flagged `difficult`
    are ignored);
  * AP per class = area under the precision envelope (precision made monotonically
    non-increasing from the right) integrated where recall changes — the VOC2010+ "all points"
    rule of object_detection.utils.metrics.compute_average_precision; classes without ground
    truth are left out of the mean (NaN);
  * CorLoc per class = fraction of the images containing the class whose highest-scoring
    detection of that class hits a ground-truth box with IoU >= 0.5.

PARITY UNPINNED (third-party evaluator, no reference test): checked against hand-computed cases
and an independent brute-force integration in tests/test_evaluation.py.
"""
import os
import shutil

import numpy as np

from cap2det_amd.core.standard_fields import DetectionResultFields, InputDataFields


def py_coord_norm_to_abs(box, height, width):
  """core/box_utils.py:188-200."""
  box = np.asarray(box).reshape(-1, 4)
  return np.stack([box[:, 0] * height, box[:, 1] * width, box[:, 2] * height, box[:, 3] * width],
                  axis=-1)


# COCO category id -> VOC category id for `--eval_coco_on_voc` (train/predict.py:293-314).
COCO_TO_VOC = {5: 1, 2: 2, 15: 3, 9: 4, 40: 5, 6: 6, 3: 7, 16: 8, 57: 9, 20: 10, 61: 11, 17: 12,
               18: 13, 4: 14, 1: 15, 59: 16, 19: 17, 58: 18, 7: 19, 63: 20}


def convert_coco_result_to_voc(boxes, scores, classes):
  """train/predict.py:284-325."""
  keep = [i for i, c in enumerate(classes) if int(c) in COCO_TO_VOC]
  if not keep:
    return np.zeros((0, 4)), np.zeros((0)), np.zeros((0), dtype=np.int64)
  return (np.stack([boxes[i] for i in keep], 0), np.stack([scores[i] for i in keep], 0),
          np.stack([COCO_TO_VOC[int(classes[i])] for i in keep], 0))


def iou_matrix(a, b):
  """[len(a), len(b)] IoU of boxes (ymin, xmin, ymax, xmax) in absolute coordinates."""
  a = np.asarray(a, np.float64).reshape(-1, 4)
  b = np.asarray(b, np.float64).reshape(-1, 4)
  area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
  area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
  ih = np.maximum(np.minimum(a[:, None, 2], b[None, :, 2]) - np.maximum(a[:, None, 0], b[None, :, 0]), 0)
  iw = np.maximum(np.minimum(a[:, None, 3], b[None, :, 3]) - np.maximum(a[:, None, 1], b[None, :, 1]), 0)
  inter = ih * iw
  union = area_a[:, None] + area_b[None, :] - inter
  return np.where(union > 0, inter / np.where(union > 0, union, 1), 0.0)


def compute_average_precision(precision, recall):
  """All-points interpolated AP (object_detection.utils.metrics.compute_average_precision)."""
  if precision is None or len(precision) == 0:
    return np.nan
  recall = np.concatenate([[0], recall, [1]])
  precision = np.concatenate([[0], precision, [0]])
  for i in range(len(precision) - 2, -1, -1):
    precision[i] = max(precision[i], precision[i + 1])
  idx = np.where(recall[1:] != recall[:-1])[0] + 1
  return float(np.sum((recall[idx] - recall[idx - 1]) * precision[idx]))


class PascalDetectionEvaluator(object):
  """Same call surface as the evaluator the reference instantiates (train/predict.py:565-569):
  categories = [{'id': 1-based int, 'name': str}, ...]."""

  def __init__(self, categories, matching_iou_threshold=0.5):
    self._categories = list(categories)
    self._ids = [c['id'] for c in self._categories]
    self._thr = matching_iou_threshold
    self.clear()

  def clear(self):
    self._gt = {}           # image_id -> (boxes, classes, difficult)
    self._seen_det = set()
    self._scores = {c: [] for c in self._ids}     # per class: detection scores
    self._tp = {c: [] for c in self._ids}         # per class: 1 TP / 0 FP
    self._num_gt = {c: 0 for c in self._ids}
    self._gt_images = {c: 0 for c in self._ids}
    self._corloc_hits = {c: 0 for c in self._ids}

  def add_single_ground_truth_image_info(self, image_id, groundtruth_dict):
    if image_id in self._gt:
      return                                          # (the reference adds it once per iteration)
    boxes = np.asarray(groundtruth_dict['groundtruth_boxes'], np.float64).reshape(-1, 4)
    classes = np.asarray(groundtruth_dict['groundtruth_classes']).reshape(-1).astype(np.int64)
    difficult = np.asarray(groundtruth_dict.get('groundtruth_difficult',
                                                np.zeros(len(classes), bool))).astype(bool)
    self._gt[image_id] = (boxes, classes, difficult)
    for c in self._ids:
      sel = (classes == c) & ~difficult
      self._num_gt[c] += int(sel.sum())
      if (classes == c).any():
        self._gt_images[c] += 1

  def add_single_detected_image_info(self, image_id, detections_dict):
    if image_id in self._seen_det:
      return
    self._seen_det.add(image_id)
    gt_boxes, gt_classes, gt_diff = self._gt.get(
        image_id, (np.zeros((0, 4)), np.zeros(0, np.int64), np.zeros(0, bool)))
    boxes = np.asarray(detections_dict['detection_boxes'], np.float64).reshape(-1, 4)
    scores = np.asarray(detections_dict['detection_scores'], np.float64).reshape(-1)
    classes = np.asarray(detections_dict['detection_classes']).reshape(-1).astype(np.int64)
    for c in self._ids:
      d = np.where(classes == c)[0]
      g = np.where(gt_classes == c)[0]
      if len(d) == 0:
        continue
      order = d[np.argsort(-scores[d], kind="stable")]
      if len(g) == 0:
        self._scores[c] += list(scores[order]); self._tp[c] += [0] * len(order)
        continue
      iou = iou_matrix(boxes[order], gt_boxes[g])
      best = iou.argmax(axis=1)
      claimed = np.zeros(len(g), bool)
      for r, k in enumerate(order):
        j = best[r]
        if iou[r, j] >= self._thr:
          if gt_diff[g[j]]:
            continue                                  # ignored: neither TP nor FP
          if not claimed[j]:
            claimed[j] = True
            self._scores[c].append(scores[k]); self._tp[c].append(1)
          else:
            self._scores[c].append(scores[k]); self._tp[c].append(0)
        else:
          self._scores[c].append(scores[k]); self._tp[c].append(0)
      if iou[0].max() >= self._thr:                   # CorLoc: the top-scoring detection hits
        self._corloc_hits[c] += 1

  def evaluate(self):
    metrics, aps, corlocs = {}, [], []
    names = {c['id']: c['name'] for c in self._categories}
    for c in self._ids:
      if self._num_gt[c] == 0:
        ap = np.nan
      else:
        s = np.asarray(self._scores[c], np.float64)
        t = np.asarray(self._tp[c], np.float64)
        order = np.argsort(-s, kind="stable")
        tp = np.cumsum(t[order]); fp = np.cumsum(1 - t[order])
        precision = tp / np.maximum(tp + fp, 1e-300)
        recall = tp / self._num_gt[c]
        ap = compute_average_precision(precision, recall) if len(s) else 0.0
      corloc = (self._corloc_hits[c] / self._gt_images[c]) if self._gt_images[c] else np.nan
      aps.append(ap); corlocs.append(corloc)
      metrics['PascalBoxes_PerformanceByCategory/AP@%.1fIOU/%s' % (self._thr, names[c])] = ap
      metrics['PascalBoxes_PerformanceByCategory/CorLoc@%.1fIOU/%s' % (self._thr, names[c])] = corloc
    metrics['PascalBoxes_Precision/mAP@%.1fIOU' % self._thr] = float(np.nanmean(aps)) if np.any(
        ~np.isnan(aps)) else np.nan
    metrics['PascalBoxes_Precision/meanCorLoc@%.1fIOU' % self._thr] = float(
        np.nanmean(corlocs)) if np.any(~np.isnan(corlocs)) else np.nan
    return metrics


class CocoDetectionEvaluator(object):
  """`coco_evaluation.CocoDetectionEvaluator(categories)` of train/predict.py:570-573 (third
  party: object_detection.metrics.coco_evaluation over pycocotools' COCOeval, iouType 'bbox').
  Restated protocol (pycocotools cocoeval.py `evaluateImg` / `accumulate` / `summarize`):

    * IoU thresholds 0.50:0.05:0.95, recall thresholds 0:0.01:1, area ranges all / small
      (< 32^2) / medium / large (>= 96^2) on the ground-truth BOX area (no masks here), maxDets
      1 / 10 / 100; boxes are [ymin, xmin, ymax, xmax] in pixels, IoU on (w, h) = differences
      of the corners (no +1);
    * per image and category: the maxDet best detections by score (stable sort); ground truth
      sorted non-ignored first (ignored = outside the area range; no crowd boxes in this reader);
      each detection takes the still-unmatched ground truth of highest IoU >= t, never trading
      a non-ignored match for an ignored one; a detection matched to an ignored box, or unmatched
      with its own area outside the range, is ignored;
    * per category: detections of all images merged by score (mergesort), tp / fp cumulated over
      the non-ignored ones, precision made non-increasing from the right and sampled at the 101
      recall thresholds (first index with recall >= r, 0 beyond the last); AP = mean over
      thresholds / categories with ground truth, AR = mean of the final recall.

  PARITY UNPINNED (pycocotools is not installable here): hand-computed cases in
  tests/test_evaluation.py."""

  IOU_THRS = np.linspace(0.5, 0.95, 10)
  REC_THRS = np.linspace(0.0, 1.0, 101)
  AREAS = [("all", 0.0, 1e10), ("small", 0.0, 32.0 ** 2), ("medium", 32.0 ** 2, 96.0 ** 2),
           ("large", 96.0 ** 2, 1e10)]
  MAX_DETS = [1, 10, 100]

  def __init__(self, categories):
    self._categories = list(categories)
    self._ids = [c['id'] for c in self._categories]
    self.clear()

  def clear(self):
    self._gt = {}
    self._det = {}

  def add_single_ground_truth_image_info(self, image_id, groundtruth_dict):
    if image_id in self._gt:
      return
    boxes = np.asarray(groundtruth_dict['groundtruth_boxes'], np.float64).reshape(-1, 4)
    classes = np.asarray(groundtruth_dict['groundtruth_classes']).reshape(-1).astype(np.int64)
    self._gt[image_id] = (boxes, classes)

  def add_single_detected_image_info(self, image_id, detections_dict):
    if image_id in self._det:
      return
    boxes = np.asarray(detections_dict['detection_boxes'], np.float64).reshape(-1, 4)
    scores = np.asarray(detections_dict['detection_scores'], np.float64).reshape(-1)
    classes = np.asarray(detections_dict['detection_classes']).reshape(-1).astype(np.int64)
    self._det[image_id] = (boxes, scores, classes)

  @staticmethod
  def _area(b):
    return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])

  def _evaluate_image(self, image_id, cat, lo, hi, max_det):
    gb, gc = self._gt.get(image_id, (np.zeros((0, 4)), np.zeros(0, np.int64)))
    db, ds, dc = self._det.get(image_id, (np.zeros((0, 4)), np.zeros(0), np.zeros(0, np.int64)))
    gb = gb[gc == cat]
    sel = dc == cat
    db, ds = db[sel], ds[sel]
    if len(gb) == 0 and len(db) == 0:
      return None
    g_ignore = ~((self._area(gb) >= lo) & (self._area(gb) <= hi))
    gorder = np.argsort(g_ignore, kind="mergesort")          # non-ignored first
    gb, g_ignore = gb[gorder], g_ignore[gorder]
    dorder = np.argsort(-ds, kind="mergesort")[:max_det]
    db, ds = db[dorder], ds[dorder]
    ious = iou_matrix(db, gb) if len(db) and len(gb) else np.zeros((len(db), len(gb)))
    T = len(self.IOU_THRS)
    gtm = -np.ones((T, len(gb)), np.int64)
    dtm = -np.ones((T, len(db)), np.int64)
    dt_ignore = np.zeros((T, len(db)), bool)
    for ti, t in enumerate(self.IOU_THRS):
      for d in range(len(db)):
        best, m = min(t, 1 - 1e-10), -1
        for g in range(len(gb)):
          if gtm[ti, g] >= 0:
            continue
          if m > -1 and not g_ignore[m] and g_ignore[g]:
            break
          if ious[d, g] < best:
            continue
          best, m = ious[d, g], g
        if m == -1:
          continue
        dt_ignore[ti, d] = g_ignore[m]
        dtm[ti, d] = m
        gtm[ti, m] = d
    d_out = ~((self._area(db) >= lo) & (self._area(db) <= hi))
    dt_ignore |= (dtm < 0) & d_out[None, :]
    return ds, dtm >= 0, dt_ignore, int((~g_ignore).sum())

  def evaluate(self):
    T, R = len(self.IOU_THRS), len(self.REC_THRS)
    K, A, M = len(self._ids), len(self.AREAS), len(self.MAX_DETS)
    precision = -np.ones((T, R, K, A, M))
    recall = -np.ones((T, K, A, M))
    images = sorted(set(self._gt) | set(self._det), key=str)
    for k, cat in enumerate(self._ids):
      for a, (_, lo, hi) in enumerate(self.AREAS):
        for mi, md in enumerate(self.MAX_DETS):
          res = [r for r in (self._evaluate_image(i, cat, lo, hi, md) for i in images) if r]
          npig = sum(r[3] for r in res)
          if npig == 0:
            continue
          scores = np.concatenate([r[0] for r in res]) if res else np.zeros(0)
          order = np.argsort(-scores, kind="mergesort")
          matched = np.concatenate([r[1] for r in res], axis=1)[:, order]
          ignored = np.concatenate([r[2] for r in res], axis=1)[:, order]
          tps = np.cumsum(matched & ~ignored, axis=1).astype(np.float64)
          fps = np.cumsum(~matched & ~ignored, axis=1).astype(np.float64)
          for ti in range(T):
            tp, fp = tps[ti], fps[ti]
            nd = len(tp)
            rc = tp / npig
            pr = tp / (fp + tp + np.spacing(1))
            recall[ti, k, a, mi] = rc[-1] if nd else 0.0
            q = np.zeros(R)
            pr = pr.tolist()
            for i in range(nd - 1, 0, -1):
              if pr[i] > pr[i - 1]:
                pr[i - 1] = pr[i]
            inds = np.searchsorted(rc, self.REC_THRS, side="left")
            for ri, pi in enumerate(inds):
              if pi < nd:
                q[ri] = pr[pi]
            precision[ti, :, k, a, mi] = q

    def mean_valid(x):
      x = x[x > -1]
      return float(x.mean()) if x.size else -1.0

    def ap(iou=None, area=0, md=2):
      p = precision[:, :, :, area, md]
      if iou is not None:
        p = p[np.isclose(self.IOU_THRS, iou)]
      return mean_valid(p)

    def ar(area=0, md=2):
      return mean_valid(recall[:, :, area, md])

    metrics = {
        'DetectionBoxes_Precision/mAP': ap(),
        'DetectionBoxes_Precision/mAP@.50IOU': ap(0.5),
        'DetectionBoxes_Precision/mAP@.75IOU': ap(0.75),
        'DetectionBoxes_Precision/mAP (small)': ap(area=1),
        'DetectionBoxes_Precision/mAP (medium)': ap(area=2),
        'DetectionBoxes_Precision/mAP (large)': ap(area=3),
        'DetectionBoxes_Recall/AR@1': ar(md=0),
        'DetectionBoxes_Recall/AR@10': ar(md=1),
        'DetectionBoxes_Recall/AR@100': ar(md=2),
        'DetectionBoxes_Recall/AR@100 (small)': ar(area=1),
        'DetectionBoxes_Recall/AR@100 (medium)': ar(area=2),
        'DetectionBoxes_Recall/AR@100 (large)': ar(area=3),
    }
    return metrics


def build_evaluators(name, categories, number_of_evaluators):
  """train/predict.py:565-576: `--evaluator pascal|coco`, one evaluator per OICR iteration."""
  if name.lower() == 'pascal':
    return [PascalDetectionEvaluator(categories) for _ in range(max(1, number_of_evaluators))]
  if name.lower() == 'coco':
    return [CocoDetectionEvaluator(categories) for _ in range(max(1, number_of_evaluators))]
  raise ValueError('Invalid evaluator {}.'.format(name))


def run_evaluation(model, batches, evaluators, category_to_id, eval_coco_on_voc=False):
  """train/predict.py:328-420: feeds every OICR iteration's detections to its evaluator.
  batches: iterable of example dicts from the reader (evaluation mode).  Returns the list of
  metric dicts, one per evaluator / OICR iteration."""
  for examples in batches:
    predictions = model.build_prediction(examples)
    batch_size = len(examples[InputDataFields.image_id])
    for i in range(batch_size):
      image_id = examples[InputDataFields.image_id][i]
      h = int(examples[InputDataFields.image_height][i])
      w = int(examples[InputDataFields.image_width][i])
      n_gt = int(examples[InputDataFields.num_objects][i])
      gt_boxes = np.asarray(examples[InputDataFields.object_boxes][i])[:n_gt]
      gt_texts = examples[InputDataFields.object_texts][i][:n_gt]
      for it, evaluator in enumerate(evaluators):
        def get(name):
          v = predictions[name + '_at_{}'.format(it)][i]
          return v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        n_det = int(get(DetectionResultFields.num_detections))
        boxes = py_coord_norm_to_abs(get(DetectionResultFields.detection_boxes)[:n_det], h, w)
        scores = get(DetectionResultFields.detection_scores)[:n_det]
        classes = get(DetectionResultFields.detection_classes)[:n_det]
        evaluator.add_single_ground_truth_image_info(image_id, {
            'groundtruth_boxes': py_coord_norm_to_abs(gt_boxes, h, w),
            'groundtruth_classes': np.array([category_to_id[t] for t in gt_texts], np.int64),
            'groundtruth_difficult': np.zeros([n_gt], dtype=bool)})
        if eval_coco_on_voc:
          boxes, scores, classes = convert_coco_result_to_voc(boxes, scores, classes)
        evaluator.add_single_detected_image_info(image_id, {
            'detection_boxes': boxes, 'detection_scores': scores, 'detection_classes': classes})
  return [e.evaluate() for e in evaluators]


def save_model_if_it_is_better(global_step, model_metric, model_path, saved_ckpts_dir,
                               reverse=False):
  """core/training_utils.py:256-308: keeps a copy of the best checkpoint files
  (`model_path*`) and the record `saved_info.txt` = 'step<TAB>metric'."""
  os.makedirs(saved_ckpts_dir, exist_ok=True)
  filename = os.path.join(saved_ckpts_dir, 'saved_info.txt')
  step_best, metric_best = None, None
  if os.path.exists(filename):
    with open(filename, 'r') as fp:
      step_best, metric_best = fp.readline().strip().split('\t')
    step_best, metric_best = int(step_best), float(metric_best)
  better = (lambda x, y: x > y) if not reverse else (lambda x, y: x < y)
  if metric_best is None or better(model_metric, metric_best):
    step_best, metric_best = global_step, model_metric
    with open(filename, 'w') as fp:
      fp.write('%d\t%.8lf' % (global_step, model_metric))
    import glob
    for existing in glob.glob(os.path.join(saved_ckpts_dir, 'model.ckpt*')):
      os.remove(existing)
    for source in glob.glob(model_path + '*'):
      shutil.copy(source, os.path.join(saved_ckpts_dir, os.path.split(source)[1]))
  return step_best, metric_best


def get_best_model_checkpoint(saved_ckpts_dir):
  """core/training_utils.py:233-253."""
  with open(os.path.join(saved_ckpts_dir, 'saved_info.txt'), 'r') as fp:
    step_best, _ = fp.readline().strip().split('\t')
  return os.path.join(saved_ckpts_dir, 'model.ckpt-{}'.format(step_best))
