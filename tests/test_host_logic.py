"""Host-side logic that needs no GPU: gradient multipliers, LR schedule, token lookup,
registry / builder error behaviour, product never imports the oracle."""
import os
import re

import numpy as np
import pytest

from cap2det_amd.models import builder, label_extractor
from cap2det_amd.models.registry import get_registered_model_classes
from cap2det_amd.protos import cap2det_model_pb2, label_extractor_pb2, model_pb2, pipeline_pb2
from cap2det_amd.train import data_parallel, trainer
from oracle import ref_labels, ref_model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gradient_multipliers_last_match_wins_and_freeze():
  tc = pipeline_pb2.TrainConfig()
  for scope, m in (("first_stage_feature_extraction", 0.0),
                   ("second_stage_feature_extraction", 1.0),
                   ("first_stage_feature_extraction/InceptionV2/Mixed_4e", 1.0)):
    tc.gradient_multiplier.add(scope=scope, multiplier=m)
  names = ["first_stage_feature_extraction/InceptionV2/Mixed_4d/Branch_0/Conv2d_0a_1x1/weights",
           "first_stage_feature_extraction/InceptionV2/Mixed_4e/Branch_0/Conv2d_0a_1x1/weights",
           "second_stage_feature_extraction/InceptionV2/Mixed_5a/Branch_0/Conv2d_0a_1x1/weights",
           "midn/proba_r_given_c/weights"]
  got = trainer.resolve_gradient_multipliers(names, tc.gradient_multiplier)
  assert names[0] not in got and got[names[1]] == 1.0 and got[names[2]] == 1.0
  assert got[names[3]] == 1.0                       # unmatched -> trainable, multiplier 1
  want = ref_model.resolve_gradient_multipliers(
      names, [(g.scope, g.multiplier) for g in tc.gradient_multiplier])
  assert got == want


def test_exponential_decay():
  assert trainer.exponential_decay(0.01, 500, 1000, 0.5, True) == pytest.approx(0.01)
  assert trainer.exponential_decay(0.01, 2500, 1000, 0.5, True) == pytest.approx(0.0025)
  assert trainer.exponential_decay(0.01, 500, 1000, 0.25, False) == pytest.approx(0.005)


def test_tokens_to_ids_matches_oracle_lookup():
  vocab = ["a", "b", "c"]
  table = {w: i for i, w in enumerate(vocab)}
  texts = [["a", "zzz", "c"], ["b"], []]
  ids = label_extractor.tokens_to_ids(texts, table, len(vocab), "cpu").numpy()
  assert ids.tolist() == [[0, 3, 2], [1, 3, 3], [3, 3, 3]]
  padded = [r + [""] * (3 - len(r)) for r in texts]
  np.testing.assert_array_equal(ids, ref_labels.tokens_to_ids(padded, vocab))
  assert label_extractor._replace_class_names(["dining table", "cat"]) == \
      ref_labels.replace_class_names(["dining table", "cat"]) == ["table", "cat"]
  assert label_extractor.tokens_to_ids([[], []], table, 3, "cpu").shape == (2, 0)


def test_builder_and_extractor_value_errors():
  assert cap2det_model_pb2.Cap2DetModel.ext in get_registered_model_classes()
  with pytest.raises(ValueError):
    builder.build(pipeline_pb2.Pipeline())
  with pytest.raises(ValueError):
    builder.build(model_pb2.Model())
  with pytest.raises(ValueError):
    label_extractor.build_label_extractor(model_pb2.Model())
  with pytest.raises(ValueError):
    label_extractor.build_label_extractor(label_extractor_pb2.LabelExtractor())  # oneof unset
  with pytest.raises(ValueError):
    trainer.Trainer(model_pb2.Model())


def test_shard_range_is_a_partition():
  for n in (0, 1, 7, 5011):
    for world in (1, 2, 3, 8):
      spans = [data_parallel.shard_range(n, r, world) for r in range(world)]
      assert spans[0][0] == 0 and spans[-1][1] == n
      assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
      assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
  with pytest.raises(ValueError):
    data_parallel.shard_range(10, 2, 2)


def test_product_never_imports_the_oracle_or_reference():
  pat = re.compile(r"^\s*(from|import)\s+(oracle|tests)\b|/root/reference", re.M)
  for base, _, files in os.walk(os.path.join(ROOT, "cap2det_amd")):
    for f in files:
      if f.endswith((".py", ".hip", ".h")):
        text = open(os.path.join(base, f)).read()
        assert not pat.search(text), os.path.join(base, f)


@pytest.mark.parametrize("fuse,commute,want", [("1", "1", (9, 5, 1)), ("1", "0", (9, 6, 0)),
                                               ("0", "1", (0, 0, 1))])
def test_second_stage_backward_plan(monkeypatch, fuse, commute, want):
  """The launch plan of the second stage for per-ROI maps (host logic only, no kernel runs):
  which convolutions get their BN/ReLU backward from the consumer's input-gradient GEMM (inner:
  9), from the next block's multi-segment GEMM (boundary: the last convolutions of Mixed_5a and
  5b), which branch runs with its average pool commuted, and that every trainable convolution
  owns exactly one row range of the partial-sum workspace."""
  import torch
  from cap2det_amd.models.frcnn_engine import SECOND_SCOPE, SECOND_STAGE, DerivedStore, Net, VariableStore
  monkeypatch.setenv("C2D_TUNE", "fuse_bn_bwd=%s,commute_avgpool=%s" % (fuse, commute))
  dev = torch.device("cpu")
  store, stats = VariableStore(dev), DerivedStore(dev)
  net = Net(store, stats, SECOND_STAGE, SECOND_SCOPE, 576, True, 1.0)
  store.finalize(); stats.finalize()
  for L in net.layers.values():
    L.trainable = True
  plan = net.plan(128, 7, 7, True)
  net._prepare_backward(plan, 0)
  ops_ = [op for st in plan["steps"] for b in st["branches"] for op in b]
  inner = sum(1 for o in ops_ if "fused_blocks" in o and "fused_wide" not in o)
  boundary = sum(1 for o in ops_ if "fused_wide" in o)
  commuted = sum(1 for o in ops_ if o.get("commuted"))
  assert (inner, boundary, commuted) == want
  assert plan["head_ok"] == (fuse != "0")
  # the commuted branch: 1x1 convolution (no ReLU) first, pool + ReLU last, both on cout channels
  for st in plan["steps"]:
    for b in st["branches"]:
      if b[0].get("commuted"):
        assert [o["kind"] for o in b] == ["conv", "pool"] and b[0]["relu"] is False
        assert b[1]["relu"] is True and b[1]["c"] == b[0]["layer"].cout == 128
  # partial-sum rows: one descriptor per trainable convolution, regions inside the workspace,
  # producers of one boundary share a region and split its columns
  convs = [o for o in ops_ if o["kind"] == "conv"]
  assert plan["bn_num"] == len(convs) == 19
  size = plan["bn_ws"].numel()
  for o in convs:
    off, n = o["bn_part"]
    assert 0 <= off and off + n <= size
  owners = {}
  for o in ops_:
    if "fused_wide" in o:
      owner, col = o["fused_wide"]
      owners.setdefault(id(owner), []).append((col, o["layer"].cout, owner["ctot"]))
  for cols in owners.values():
    cols.sort()
    assert all(c0 + w0 <= c1 for (c0, w0, _), (c1, _, _) in zip(cols, cols[1:]))
    assert cols[-1][0] + cols[-1][1] <= cols[0][2]


def test_dropout_key_mixes_seed_step_and_rank():
  """The default dropout key of Trainer.train_step (slim.dropout draws a fresh mask per step and
  per worker, models/utils.py:171-174): distinct for every (seed, step, rank), int64-safe."""
  keys = set()
  for seed in (0, 1, 1234):
    for step in range(50):
      for rank in range(8):
        k = trainer.dropout_key(seed, step, rank, 8)
        assert 0 <= k < (1 << 63)
        keys.add(k)
  assert len(keys) == 3 * 50 * 8
  assert trainer.dropout_key(7, 3, 1, 2) == trainer.dropout_key(7, 3, 1, 2)


def test_predict_checkpoint_errors_are_typed(tmp_path):
  """train/predict.py:583-611 polls model_dir; only an unreadable CHECKPOINT is retried (bounded),
  any other missing file stays the FileNotFoundError it is."""
  from cap2det_amd.train import predict
  with pytest.raises(predict.CheckpointUnreadable):
    predict.read_checkpoint_arrays(str(tmp_path / "model.ckpt-7"))
  bad = tmp_path / "model.ckpt-8.npz"
  bad.write_bytes(b"PK\x03\x04 truncated")
  with pytest.raises(predict.CheckpointUnreadable):
    predict.read_checkpoint_arrays(str(tmp_path / "model.ckpt-8"))
  good = tmp_path / "model.ckpt-9.npz"
  np.savez(str(good), __global_step=np.int64(9), __adagrad_accumulators=np.zeros(3), w=np.ones(2))
  arrays = predict.read_checkpoint_arrays(str(tmp_path / "model.ckpt-9"))
  assert sorted(arrays) == ["w"]
  assert not issubclass(predict.CheckpointUnreadable, FileNotFoundError)


def test_oracle_optimizer_rules_known_answers():
  """Hand-computed steps of the TensorFlow 1.x rules behind core/training_utils.py:14-71's
  `build_optimizer` (oracle/ref_model.optimizer_update; the GPU kernel is tested against it)."""
  f = np.float64
  g = np.array([0.5, -2.0, 0.0])
  # Adam, step 1: m_hat / (sqrt(v_hat) + eps) = sign(g) up to eps  ->  w -= lr * sign(g)
  w = np.ones(3)
  sl = ref_model.init_optimizer_slots("adam", {}, {"w": w})["w"]
  ref_model.optimizer_update("adam", dict(beta1=0.9, beta2=0.999, epsilon=1e-8), w, g, sl, 0.1, 1, f)
  np.testing.assert_allclose(w, [0.9, 1.1, 1.0], atol=1e-7)
  np.testing.assert_allclose(sl[0], 0.1 * g)
  np.testing.assert_allclose(sl[1], 0.001 * g * g)
  # momentum 0.5, two steps of the same gradient: a = g, then 1.5 g;  w -= lr a each step
  w = np.zeros(3)
  sl = ref_model.init_optimizer_slots("momentum", {}, {"w": w})["w"]
  for _ in range(2):
    ref_model.optimizer_update("momentum", dict(momentum=0.5), w, g, sl, 0.1, 1, f)
  np.testing.assert_allclose(w, -0.1 * (1.0 + 1.5) * g)
  # Nesterov, first step: a = g, w -= lr (g + mu a) = lr * 1.5 g
  w = np.zeros(3)
  sl = ref_model.init_optimizer_slots("momentum", {}, {"w": w})["w"]
  ref_model.optimizer_update("momentum", dict(momentum=0.5, use_nesterov=True), w, g, sl, 0.1, 1, f)
  np.testing.assert_allclose(w, -0.1 * 1.5 * g)
  # RMSProp (rms slot starts at ONE), centered, momentum 0.5: worked by hand for g = 0.5
  w = np.ones(1)
  sl = ref_model.init_optimizer_slots("rmsprop", dict(centered=True), {"w": w})["w"]
  ref_model.optimizer_update("rmsprop", dict(decay=0.9, momentum=0.5, epsilon=1e-10, centered=True),
                             w, np.array([0.5]), sl, 0.1, 1, f)
  ms, mg = 0.9 + 0.1 * 0.25, 0.1 * 0.5
  np.testing.assert_allclose(w, [1.0 - 0.1 * 0.5 / np.sqrt(ms - mg * mg + 1e-10)], rtol=1e-12)
  # SGD and Adagrad (accumulator 0.1)
  w = np.ones(3)
  ref_model.optimizer_update("sgd", {}, w, g, [], 0.1, 1, f)
  np.testing.assert_allclose(w, 1.0 - 0.1 * g)
  w = np.ones(3)
  sl = ref_model.init_optimizer_slots("adagrad", dict(initial_accumulator_value=0.1), {"w": w})["w"]
  ref_model.optimizer_update("adagrad", {}, w, g, sl, 0.1, 1, f)
  np.testing.assert_allclose(w, 1.0 - 0.1 * g / np.sqrt(0.1 + g * g))
  with pytest.raises(ValueError):
    ref_model.optimizer_update("lamb", {}, w, g, [], 0.1, 1, f)
