"""The committed seeded vectors (tests/golden/*.npz, written by tests/golden/gen_golden.py) pin
the oracle: any change to the restatement shows up here on CPU; the same files are the fixed
inputs/outputs of the GPU parity tests (test_gpu_golden below)."""
import os

import numpy as np
import pytest

from oracle import ref_labels, ref_model, ref_ops


def _load(golden_dir, name):
  return dict(np.load(os.path.join(golden_dir, name)))


def test_oracle_reproduces_roi_crop_case(golden_dir):
  g = _load(golden_dir, "roi_crop_case.npz")
  crop = ref_ops.crop_and_resize(g["feat"], g["boxes"], g["box_ind"], 14)
  assert crop.astype(np.float64).sum() == pytest.approx(float(g["crop_checksum"]), rel=1e-12)
  pooled, arg = ref_ops.max_pool(crop, 2, 2, "VALID")
  np.testing.assert_array_equal(pooled, g["pooled"])
  np.testing.assert_array_equal(arg, g["argmax"])
  assert np.all(pooled[0] == g["feat"][g["box_ind"][0], 0, 0])      # zero box -> pixel (0,0)


def test_oracle_reproduces_heads_case(golden_dir):
  g = _load(golden_dir, "heads_case.npz")
  P = {k[2:]: v for k, v in g.items() if k.startswith("P:")}
  cl, scores, proba, _ = ref_model.build_midn_network(g["num"], g["x"], P)
  np.testing.assert_allclose(cl, g["class_logits"], rtol=1e-6)
  np.testing.assert_allclose(proba, g["proba"], rtol=1e-6, atol=1e-9)
  assert np.all(proba[1, 20:] == 0) and np.allclose(proba.sum(1), 1.0, atol=1e-5)


def test_oracle_reproduces_text_and_conv_cases(golden_dir):
  g = _load(golden_dir, "text_classifier_case.npz")
  f = lambda k: g[k].astype(np.float32)
  logits = ref_labels.text_classifier_logits(g["ids"], f("emb"), f("w1"), g["b1"], f("w2"), g["b2"])
  np.testing.assert_allclose(logits, g["logits"], rtol=1e-5, atol=1e-6)
  c = _load(golden_dir, "conv_case.npz")
  for s in (1, 2):
    np.testing.assert_allclose(ref_ops.conv2d(c["x"], c["w"], s), c["y_s%d" % s], rtol=1e-4,
                               atol=1e-5)


@pytest.mark.gpu
def test_gpu_golden(golden_dir):
  import torch
  from cap2det_amd import hip_ops as ops
  dev = "cuda:0"
  t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
  g = _load(golden_dir, "roi_crop_case.npz")
  out, arg = ops.roi_crop_pool_fwd(t(g["feat"]), t(g["boxes"]), t(g["box_ind"]), 14, 2, 2)
  np.testing.assert_array_equal(out.cpu().numpy(), g["pooled"])
  np.testing.assert_array_equal(arg.cpu().numpy(), g["argmax"])
  dfeat = torch.zeros(g["feat"].shape, device=dev)
  ops.roi_crop_pool_bwd(t(g["dout"]), arg, t(g["boxes"]), t(g["box_ind"]), dfeat, 14, 2, 2)
  np.testing.assert_allclose(dfeat.cpu().numpy(), g["dfeat"], rtol=1e-4, atol=1e-4)
  c = _load(golden_dir, "conv_case.npz")
  x, w = t(c["x"]), t(c["w"])
  wt = torch.empty(9, 48, 32, device=dev)
  ops.transpose_taps(w, wt, 9, 32, 48)
  for s in (1, 2):
    oh = -(-7 // s)
    y = torch.empty(3, oh, oh, 48, device=dev)
    ops.conv_fwd(x, 32, 0, wt, None, None, y, 48, 0, 3, 7, 7, 32, 48, 3, 3, s, False)
    np.testing.assert_allclose(y.cpu().numpy(), c["y_s%d" % s], rtol=1e-4, atol=1e-5)
    dx = torch.empty(3, 7, 7, 32, device=dev)
    ops.conv_dgrad(t(c["dy_s%d" % s]), 48, 0, w, dx, 32, 0, 3, 7, 7, 32, 48, 3, 3, s, False)
    np.testing.assert_allclose(dx.cpu().numpy(), c["dx_s%d" % s], rtol=1e-4, atol=1e-5)
    dw = torch.zeros(3, 3, 32, 48, device=dev)
    ops.conv_wgrad(x, 32, 0, t(c["dy_s%d" % s]), 48, 0, dw, 3, 7, 7, 32, 48, 3, 3, s)
    np.testing.assert_allclose(dw.cpu().numpy(), c["dw_s%d" % s], rtol=1e-4, atol=1e-4)
  h = _load(golden_dir, "text_classifier_case.npz")
  f = lambda k: t(h[k].astype(np.float32))
  logits = torch.empty(4, 7, device=dev); labels = torch.empty(4, 7, device=dev)
  ops.text_classifier_fwd(t(h["ids"]), f("emb"), f("w1"), t(h["b1"]), f("w2"), t(h["b2"]),
                          t(h["exact"]), 0.5, logits, labels)
  np.testing.assert_allclose(logits.cpu().numpy(), h["logits"], rtol=1e-4, atol=1e-4)
