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


def _write_label_files(tmp_path, classes, vocab, emb):
  lf, vf, ef = tmp_path / "labels.txt", tmp_path / "vocab.txt", tmp_path / "emb.npy"
  lf.write_text("\n".join(classes))
  vf.write_text("\n".join(vocab))
  np.save(str(ef), emb)
  return str(lf), str(vf), str(ef)


def test_label_extractors_match_oracle(tmp_path):
  """All five extractors of models/label_extractor.py through the reference's factory
  (`build_label_extractor`), strings in, device labels out, against the oracle."""
  from cap2det_amd.models import label_extractor as le
  from cap2det_amd.protos import label_extractor_pb2, text_format
  rng = np.random.default_rng(17)
  classes = ["person", "bird", "dining table", "tie"]
  vocab = ["person", "bird", "table", "tie", "man", "goose", "fork", "sky", "boy", "red"]
  emb = (0.4 * rng.standard_normal((len(vocab), 300))).astype(np.float32)
  emb[4] = emb[0] + 0.05 * rng.standard_normal(300)          # man ~ person
  emb[5] = emb[1] + 0.05 * rng.standard_normal(300)          # goose ~ bird
  lf, vf, ef = _write_label_files(tmp_path, classes, vocab, emb)
  captions = [["a", "man", "red", "sky"], ["goose", "", "", ""], ["zzz", "qqq", "", ""],
              ["table", "fork", "", ""], ["", "", "", ""]]

  def build(txt):
    cfg = label_extractor_pb2.LabelExtractor()
    text_format.Merge(txt, cfg)
    return le.build_label_extractor(cfg, DEV)

  # string extractors
  ex = build("exact_match_extractor { label_file: '%s' }" % lf)
  got = ex.extract_labels({"concat_caption_string": captions}).cpu().numpy()
  np.testing.assert_array_equal(got, ref_labels.exact_match_extract(captions, classes))
  # word-vector match
  wv = build("word_vector_match_extractor { label_file: '%s' open_vocabulary_file: '%s' "
             "open_vocabulary_word_embedding_file: '%s' }" % (lf, vf, ef))
  assert isinstance(wv, le.WordVectorMatchExtractor) and wv.num_classes == 4
  oov_row = wv._embedding[-1].cpu().numpy()
  full = np.concatenate([emb, oov_row[None]], 0)
  ids = ref_labels.tokens_to_ids(captions, vocab)
  class_ids = [vocab.index(c) for c in ref_labels.replace_class_names(classes)]
  want = ref_labels.word_vector_match_extract(ids, ref_labels.exact_match_extract(captions, classes),
                                              full, class_ids)
  got = wv.extract_labels({"concat_caption_string": captions}).cpu().numpy()
  np.testing.assert_array_equal(got, want)
  assert got[0].tolist() == [1, 0, 0, 0] and got[1].tolist() == [0, 1, 0, 0]   # man->person, goose->bird
  assert got[2].sum() == 0 and got[4].sum() == 0                                # all-OOV captions
  assert got[3].tolist() == [0, 0, 1, 0]                                        # exact match wins
  with pytest.raises(ValueError):                                               # class without a vector
    bad = tmp_path / "bad.txt"; bad.write_text("person\nunicorn")
    build("word_vector_match_extractor { label_file: '%s' open_vocabulary_file: '%s' "
          "open_vocabulary_word_embedding_file: '%s' }" % (str(bad), vf, ef))
  # text classifier
  tc = build("text_classifier_match_extractor { label_file: '%s' open_vocabulary_file: '%s' "
             "open_vocabulary_word_embedding_file: '%s' hidden_units: 32 label_threshold: 0.5 }"
             % (lf, vf, ef))
  w = {"text_classifier/layer1/weights": (rng.standard_normal((300, 32)) / 17).astype(np.float32),
       "text_classifier/layer1/biases": (0.1 * rng.standard_normal(32)).astype(np.float32),
       "text_classifier/layer2/weights": (rng.standard_normal((32, 4)) / 5).astype(np.float32),
       "text_classifier/layer2/biases": (0.1 * rng.standard_normal(4)).astype(np.float32)}
  with pytest.raises(ValueError):
    tc.extract_labels({"concat_caption_string": captions})      # weights not loaded yet
  tc.load_weights(w)
  full = np.concatenate([emb, tc._embedding[-1].cpu().numpy()[None]], 0)
  exact_raw = ref_labels.match_labels(captions, classes)
  want_logits = ref_labels.text_classifier_logits(
      ids, full, w["text_classifier/layer1/weights"], w["text_classifier/layer1/biases"],
      w["text_classifier/layer2/weights"], w["text_classifier/layer2/biases"])
  got_logits = tc.predict({"concat_caption_string": captions}).cpu().numpy()
  np.testing.assert_allclose(got_logits, want_logits, rtol=1e-4, atol=1e-5)
  want = np.where((exact_raw > 0).any(-1)[:, None], exact_raw,
                  (ref_labels.ops.sigmoid(want_logits) > 0.5).astype(np.float32))
  got = tc.extract_labels({"concat_caption_string": captions}).cpu().numpy()
  safe = np.abs(ref_labels.ops.sigmoid(want_logits) - 0.5) > 1e-3
  np.testing.assert_array_equal(got[safe], want[safe])
