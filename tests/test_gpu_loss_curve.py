"""Multi-step check of the reduced-precision storage modes: the same model trained on the
same batches with the same dropout seeds in fp32, in bf16 with an fp32 first stage and in bf16 with
the bf16 first stage (the default of compute_dtype="bf16") must follow the same loss trajectory —
tools/loss_curve.py, a short run here (the committed full-size curves: profiles/r04_loss_curve.json).
The reference computes in fp32 only (/root/reference/models/utils.py:108-188); the bf16 modes are a
changed numerical operating point, bounded here over a trajectory instead of over one step."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bf16_modes_follow_the_fp32_loss_trajectory():
  spec = importlib.util.spec_from_file_location("loss_curve", os.path.join(ROOT, "tools", "loss_curve.py"))
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  doc = mod.run_curves(steps=120, hw=224, proposals=256, dm=1.0, pool=4, window=20)
  for name, c in doc["curves"].items():
    assert c["all_finite"], name
  assert doc["curves"]["bf16"]["first_stage"] == "bfloat16"
  assert doc["curves"]["bf16_fp32first"]["first_stage"] == "float32"
  assert doc["curves"]["bf16_fp32first"]["second_stage"] == "bfloat16"
  print({k: v["max_relative_window_deviation"] for k, v in doc["deviation_from_fp32"].items()})
  print({k: v["trajectory"] for k, v in doc["curves"].items()})
  for name, d in doc["deviation_from_fp32"].items():
    # windows of 20 steps: the total loss of the bf16 runs within 2 % of the fp32 window means,
    # every single term (the OICR terms are two orders smaller than the total) within 10 %
    for term, v in d["max_relative_window_deviation"].items():
      assert v <= (2e-2 if term == "total_loss" else 1e-1), (name, term, v)
  moved = doc["curves"]["fp32"]["trajectory"]["distance_moved"]
  assert moved > 0
  # the shipped fp32 network (second-stage GEMMs as nine bf16 partial products, f32x9) against the
  # same network with every GEMM on the fp32 matrix pipe: the SAME arithmetic to fp32 rounding — the
  # two runs stay an order of magnitude closer to each other than the bf16 modes stay to them
  t = doc["curves"]["fp32_mfma"]["trajectory"]
  assert t["cosine_with_fp32_displacement"] >= 0.999, t
  assert t["relative_deviation"] <= 0.03, t
  for term, v in doc["deviation_from_fp32"]["fp32_mfma"]["max_relative_window_deviation"].items():
    assert v <= (2e-3 if term == "total_loss" else 2e-2), (term, v)
  for name in ("bf16_fp32first", "bf16"):
    t = doc["curves"][name]["trajectory"]
    # the bf16 runs displace the trainable variables the way the fp32 run does: same direction,
    # and they end closer to the fp32 run than a quarter of the way either has come
    assert t["cosine_with_fp32_displacement"] >= 0.97, (name, t)
    assert t["relative_deviation"] <= 0.25, (name, t)
