"""The first stage against the float64 oracle AT THE BENCHMARK'S IMAGE SIZE (missing #2):
tests/golden/first_stage_{500x500,1000x1333}.npz (tests/golden/gen_first_stage_fixture.py: the
object_detection Inception-V2 `extract_proposal_features` of /root/reference/models/utils.py:127-136
restated on torch-CPU float64 ops) hold, for the stem and every top-level op up to Mixed_4e, the
per-channel L2 norms and 4096 sampled entries of the output map.  The HIP first stage — stem =
c2d_preprocess_pad4 + c2d_im2col4 + 1x1 GEMM, then the `igemm_small*` launches (grouped per
Inception level) and the 3x3 pools on 250^2 / 125^2 / 63^2 / 32^2 maps (500^2 / 334^2 ... 63x84
for the reference's 1000x1333 training shape) — must reproduce them:

  fp32 : samples within 2e-5 of the map's largest magnitude, channel norms within 2e-5 relative;
  bf16 first stage (`compute_dtype="bf16"`: every map behind the fp32 stem stored in bf16, fp32
  accumulation; twelve layers, one rounding of 2^-9 relative each): samples within 1.5e-2 of the
  largest magnitude, channel norms within 8e-3 of the layer's largest channel norm (a channel
  the ReLU leaves nearly dead has no relative accuracy of its own); observed at Mixed_4e: samples
  5e-3 / 6e-3, norms 3.5e-3 / 3.8e-3 at 500x500 / 1000x1333 (printed by the test).

At 500x500 the forward instances of the `igemm_small*` family dispatched here must be the ones of
the newest committed benchmark profiles (fp32: c1, bf16: c2)."""
import csv
import glob
import os
import re

import numpy as np
import pytest
import torch

from tests import util_model
from tests.golden import gen_first_stage_fixture as gen

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = {"fp32": dict(sample=2e-5, norm=2e-5), "bf16": dict(sample=1.5e-2, norm=8e-3)}


def _profile_first_stage_forward_instances(cfg):
  path = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_bench_kernel_stats_%s_serial.csv" % cfg)))[-1]
  out = set()
  with open(path) as f:
    for row in csv.DictReader(f):
      m = re.search(r"(igemm_small\w*<0, \d+>)", row["Name"])
      if m:
        out.add(m.group(1))
  return out


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
@pytest.mark.parametrize("hw", gen.SIZES)
def test_first_stage_matches_the_float64_fixture(monkeypatch, hw, dtype):
  from cap2det_amd import hip_ops
  from cap2det_amd.train.trainer import Trainer
  fix = np.load(os.path.join(ROOT, "tests", "golden", "first_stage_%dx%d.npz" % hw))
  image, P32 = gen.inputs(hw)
  np.testing.assert_allclose(gen.checksum(image, P32), fix["checksum"], rtol=1e-12)
  trainer = Trainer(util_model.load_pipeline(), device=DEV, depth_multiplier=gen.DM,
                    compute_dtype=dtype)
  model = trainer.model
  engine = model.engine
  assert engine.first.dtype == (torch.float32 if dtype == "fp32" else torch.bfloat16)
  state = model.state_dict()
  assert all(k in state for k in P32)
  state.update(P32)
  model.load_state_dict(state)
  seen = set()
  for name in ("conv_fwd", "conv_fwd_grouped"):
    inner = getattr(hip_ops, name)
    def wrapped(*a, _inner=inner, **k):
      r = _inner(*a, **k)
      seen.update(hip_ops.last_dispatch())
      return r
    monkeypatch.setattr(hip_ops, name, wrapped)
  img = torch.from_numpy(image).to(DEV)
  boxes = torch.tensor([[[0.1, 0.1, 0.6, 0.7], [0.0, 0.0, 1.0, 1.0], [0.3, 0.2, 0.9, 0.5],
                         [0.5, 0.5, 0.8, 0.95]]], device=DEV)
  engine.forward(img, boxes, True, dropout_seed=1)
  torch.cuda.synchronize()
  bufs = engine._shape_cache[(1, hw[0], hw[1], 4, True)]
  outputs = [bufs["stem"].t] + [st["y"].t for st in bufs["plan1"]["steps"]]
  names = [str(s) for s in fix["names"]]
  assert len(outputs) == len(names) == 12 and names[-1] == "Mixed_4e"
  worst = {}
  tol = TOL[dtype]
  for name, y in zip(names, outputs):
    shape = tuple(int(v) for v in fix[name + "/shape"])
    got = y.detach().float().cpu().numpy().astype(np.float64)
    assert got.shape == (shape[0] * shape[1], shape[2]), (name, got.shape, shape)
    scale = float(fix[name + "/absmax"])
    flat = got.reshape(-1)
    err = np.abs(flat[gen.sample_indices(name, flat.size)] - fix[name + "/samples"]).max() / scale
    norm_want = fix[name + "/channel_norm"]
    norm_got = np.sqrt((got ** 2).sum(0))
    if dtype == "fp32":    # every channel on its own (floor: a thousandth of the largest norm)
      norm_err = (np.abs(norm_got - norm_want) / np.maximum(norm_want, 1e-3 * norm_want.max())).max()
    else:
      norm_err = np.abs(norm_got - norm_want).max() / norm_want.max()
    worst[name] = (float("%.3g" % err), float("%.3g" % norm_err))
  print("first stage %dx%d %s (sample error / absmax, channel-norm error):" % (hw + (dtype,)), worst)
  for name, (err, norm_err) in worst.items():
    assert err <= tol["sample"], (name, err, worst)
    assert norm_err <= tol["norm"], (name, norm_err, worst)
  if dtype == "bf16":
    assert worst["Mixed_4e"][0] > 1e-5           # (it really ran in reduced precision)
  if hw == (500, 500):
    want = _profile_first_stage_forward_instances("c1" if dtype == "fp32" else "c2")
    # (fp32 `igemm_small*<0, 4>` instances of the bf16 profile are the heads GEMM, not first stage)
    want = {w for w in want if w.endswith(", 4>" if dtype == "fp32" else ", 2>")}
    assert want and want <= seen, (sorted(want), sorted(seen))
