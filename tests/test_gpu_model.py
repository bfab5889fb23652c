"""End-to-end GPU parity: one full training step of the HIP path (through the reference-shaped
plugin API) against the float64 numpy oracle on identical seeded inputs.

Tolerance: proposal-score tensors within 1e-4 absolute (BASELINE.json north_star); losses
1e-4 relative; gradients / updated variables 5e-4 of the tensor's max magnitude (fp32 MFMA
accumulation order and fp32 atomics differ from a float64 sum)."""
import numpy as np
import pytest
import torch

from oracle import ref_labels, ref_model
from tests import util_model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _close(got, want, tol, name):
  got = np.asarray(got, np.float64)
  want = np.asarray(want, np.float64)
  scale = max(np.abs(want).max(), 1e-30)
  err = np.abs(got - want).max()
  # 3e-7 absolute floor: sums that are analytically zero (e.g. the bias gradient of the
  # shift-invariant proposal softmax) only carry fp32 round-off of O(1) summands.
  assert err <= tol * scale + 3e-7, "%s: max err %.3e vs scale %.3e" % (name, err, scale)


def _to_dev(ex):
  out = dict(ex)
  for k in ("image", "proposals"):
    out[k] = torch.from_numpy(ex[k]).to(DEV).contiguous()
  out["number_of_proposals"] = torch.from_numpy(ex["number_of_proposals"]).to(DEV)
  return out


@pytest.mark.parametrize("dm,hw,n,nums", [(1.0, (64, 48), 6, [6, 4]), (0.5, (40, 72), 9, [9, 0])])
def test_train_step_matches_oracle(dm, hw, n, nums):
  from cap2det_amd.train.trainer import Trainer
  pipeline = util_model.load_pipeline()
  rng = np.random.default_rng(99)
  trainer = Trainer(pipeline, device=DEV, depth_multiplier=dm)
  model = trainer.model
  classes = model.label_extractor.classes
  c, k = len(classes), 3
  P32, d = util_model.oracle_state(5, c, k, dm)
  model.load_state_dict(P32)
  ex = util_model.make_examples(rng, 2, hw[0], hw[1], n, nums, classes)
  mask = (rng.uniform(size=(2 * n, d)) < 0.5).astype(np.uint8)

  # ---- oracle, float64 arithmetic on the same fp32 values --------------------------
  P = {kk: v.astype(np.float64) for kk, v in P32.items()}
  acc = {kk: np.full(v.shape, 0.1) for kk, v in P.items()}
  labels = ref_labels.groundtruth_extract(ex["object_texts"], classes).astype(np.float64)
  ex64 = dict(image=ex["image"].astype(np.float64), number_of_proposals=ex["number_of_proposals"],
              proposals=ex["proposals"].astype(np.float64))
  opts = ref_model.FrcnnOptions(depth_multiplier=dm)
  loss_opts = dict(midn_loss_weight=1.0, oicr_loss_weight=0.5, oicr_iterations=k,
                   oicr_iou_threshold=0.6, oicr_use_proba_r_given_c=True)
  mults = [(g.scope, g.multiplier) for g in pipeline.train_config.gradient_multiplier]
  P_before = {kk: v.copy() for kk, v in P.items()}
  want = ref_model.train_step(P, acc, ex64, labels, opts, loss_opts, mults, 0.01, 1e-6, mask)

  # ---- HIP path -------------------------------------------------------------------
  losses = trainer.train_step(_to_dev(ex), dropout_mask=torch.from_numpy(mask).to(DEV))
  torch.cuda.synchronize()
  pred = trainer.predictions
  wp = want["predictions"]
  for name in ["midn_class_logits", "midn_proba_r_given_c"] + \
      ["oicr_proposal_scores_at_%d" % i for i in range(k + 1)]:
    got = pred[name].detach().cpu().numpy()
    assert np.abs(got - wp[name]).max() <= 1e-4, name       # north-star tolerance, absolute
  for name, v in want["losses"].items():
    np.testing.assert_allclose(losses[name].item(), v, rtol=1e-4, err_msg=name)
  np.testing.assert_allclose(losses["regularization_loss"].item(),
                             sum(want["reg_losses"].values()), rtol=1e-4)
  np.testing.assert_allclose(losses["total_loss"].item(), want["total_loss"], rtol=1e-4)

  grads = model.grad_dict()
  checked = 0
  for name, g in want["applied"].items():
    w = want["grads"][name]
    if ref_model.is_regularized(name):
      w = w - 1e-6 * P_before[name]        # the HIP path adds l2*w inside the Adagrad kernel
    _close(grads[name], w, 5e-4, "grad " + name)
    checked += 1
  assert checked > 60
  # variables that must not move: everything in the frozen part of the first stage
  state = model.state_dict()
  for name in P:
    if name in want["applied"]:
      # Adagrad step w -= lr*g/sqrt(acc), acc >= 0.1: a gradient error dg (<= 5e-4 of the
      # gradient scale, checked above) moves the update by at most lr*dg/sqrt(0.1).
      gscale = np.abs(want["applied"][name]).max()
      bound = 0.01 / np.sqrt(0.1) * (5e-4 * gscale + 3e-7) + 1e-6 * np.abs(P[name]).max()
      err = np.abs(state[name].astype(np.float64) - P[name]).max()
      assert err <= bound, "updated %s: err %.3e > bound %.3e" % (name, err, bound)
    else:
      np.testing.assert_array_equal(state[name], P32[name], err_msg="frozen " + name)
  assert "first_stage_feature_extraction/InceptionV2/Mixed_4e/Branch_0/Conv2d_0a_1x1/weights" \
      in want["applied"]
  assert "first_stage_feature_extraction/InceptionV2/Mixed_4d/Branch_0/Conv2d_0a_1x1/weights" \
      not in want["applied"]


def test_builder_and_errors():
  from cap2det_amd.models import builder
  from cap2det_amd.models.cap2det_model import Model
  from cap2det_amd.protos import label_extractor_pb2, model_pb2, pipeline_pb2
  pipeline = util_model.load_pipeline()
  m = builder.build(pipeline.model, is_training=False, device=DEV, depth_multiplier=0.5)
  assert isinstance(m, Model) and m.num_classes == 20
  with pytest.raises(ValueError):
    builder.build(pipeline_pb2.Pipeline())            # wrong proto type (models/builder.py:26-27)
  with pytest.raises(ValueError):
    builder.build(model_pb2.Model())                  # no extension set (:35-37)
  with pytest.raises(ValueError):
    Model(label_extractor_pb2.LabelExtractor())       # models/cap2det_model.py:41-42
