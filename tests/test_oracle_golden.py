"""Pins the oracle against every known-answer test the reference holds for the hot path
(SURVEY.md §8c): core/utils_test.py, core/box_utils_test.py, models/label_extractor_test.py."""
import json
import os

import numpy as np
import pytest

from oracle import ref_labels, ref_ops


def _load(golden_dir, name):
  with open(os.path.join(golden_dir, name)) as f:
    return json.load(f)["cases"]


def test_core_utils_known_answers(golden_dir):
  cases = _load(golden_dir, "core_utils_known_answers.json")
  assert len(cases) == 16
  for c in cases:
    fn = getattr(ref_ops, c["fn"])
    data = np.asarray(c["data"], np.float32)
    mask = np.asarray(c["mask"], np.float32)
    kwargs = {}
    got = fn(data, mask, **kwargs) if c["fn"] != "masked_softmax" else fn(data, mask)
    np.testing.assert_allclose(got, np.asarray(c["expected"], np.float32), rtol=1e-6, atol=1e-6,
                               err_msg="%s %s" % (c["fn"], c["cite"]))


def test_box_utils_known_answers(golden_dir):
  for c in _load(golden_dir, "box_utils_known_answers.json"):
    fn = getattr(ref_ops, c["fn"])
    if c["fn"] == "scale_to_new_size":
      got = fn(np.asarray(c["box"], np.float32), c["img_shape"], c["pad_shape"])
    elif "box1" in c:
      got = fn(np.asarray(c["box1"], np.float32), np.asarray(c["box2"], np.float32))
      got = np.nan_to_num(got, nan=0.0) if c["fn"] == "iou" and False else got
    else:
      got = fn(np.asarray(c["box"], np.float32))
    np.testing.assert_allclose(got, np.asarray(c["expected"], np.float32), rtol=1e-6, atol=1e-7,
                               err_msg="%s %s" % (c["fn"], c["cite"]))


def test_label_extractor_known_answers(golden_dir, tmp_path):
  for c in _load(golden_dir, "label_extractor_known_answers.json"):
    path = tmp_path / "label_file.txt"
    path.write_text("\n".join(c["label_file_lines"]))
    for tokens, expected in zip(c["inputs"], c["expected"]):
      if c["extractor"] == "groundtruth_extractor":
        classes = ref_labels.read_label_file(str(path))
        got = ref_labels.groundtruth_extract(tokens, classes)
      elif c["extractor"] == "exact_match_extractor":
        classes = ref_labels.read_label_file(str(path))
        got = ref_labels.exact_match_extract(tokens, classes)
      else:
        name2id, classes = ref_labels.read_synonym_file(str(path))
        got = ref_labels.extend_match_extract(tokens, name2id, len(classes))
      assert classes == c["classes"]
      np.testing.assert_array_equal(got, np.asarray(expected, np.float32), err_msg=c["cite"])


def test_masked_argmax_quirk_all_equal_min():
  """the axis minimum is taken over padded rows too."""
  data = np.array([[[0.0], [5.0], [5.0]]], np.float32)      # row 0 is padding
  mask = np.array([[[0.0], [1.0], [1.0]]], np.float32)
  assert ref_ops.masked_argmax(data, mask, dim=1)[0, 0] == 1
  data2 = np.array([[[7.0], [3.0], [3.0]]], np.float32)      # valid rows all equal the min
  assert ref_ops.masked_argmax(data2, mask, dim=1)[0, 0] == 0


def test_iou_degenerate_is_nan_and_compares_false():
  b = np.zeros((1, 4), np.float32)
  v = ref_ops.iou(b, b)
  assert np.isnan(v[0]) and not (v[0] >= 0.5)
