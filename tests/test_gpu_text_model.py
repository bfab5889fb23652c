"""GPU parity of the text-classifier pre-training against
oracle/ref_text.py (float64): pooling kernels incl. TensorFlow's tie rules, one full Adagrad
step (logits within the north-star 1e-4, loss, gradients, updated variables), the streaming
metrics, and the hand-over of the trained weights to the detection model's label extractor."""
import numpy as np
import pytest
import torch

from oracle import ref_labels, ref_text
from tests import util_model

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _t(a):
  return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def test_text_pool_kernels_match_oracle_with_ties():
  from cap2det_amd import hip_ops as ops
  rng = np.random.default_rng(3)
  B, T, H, V = 5, 7, 70, 11
  ids = rng.integers(0, V + 1, (B, T)).astype(np.int32)
  ids[1] = V                                    # all OOV
  ids[2, 1:] = V
  pre = rng.standard_normal((B, T, H)).astype(np.float32)
  pre[3, 2] = pre[3, 4]                         # tied rows (same token twice)
  pre[4] = np.round(pre[4])                     # many exact ties
  keep = (rng.uniform(size=(B, H)) < 0.5).astype(np.uint8)
  mu = (ids != V).astype(np.float64)[..., None]
  p64 = pre.astype(np.float64)
  m = p64.min(1, keepdims=True); z = (p64 - m) * mu; y = z.max(1) + m[:, 0]
  want_h = np.maximum(y, 0) * keep / 0.5
  hidden = torch.empty(B, H, device=DEV)
  ops.text_pool_fwd(_t(pre.reshape(B * T, H)), _t(ids), H, V, _t(keep), 0.5, hidden)
  np.testing.assert_allclose(hidden.cpu().numpy(), want_h, rtol=1e-6, atol=1e-6)
  dh = rng.standard_normal((B, H)).astype(np.float32)
  tape = dict(h=want_h, pre=p64, m=m, z=z, mu=mu, y=y, x=np.zeros((B, T, 1)), keep_mask=keep.astype(np.float64),
              keep_prob=0.5)
  # reuse the oracle's pooling gradient: feed dlogits = dh through an identity second layer
  _, want_dpre = ref_text.backward(dh.astype(np.float64), tape, np.eye(H))
  dpre = torch.empty(B * T, H, device=DEV)
  ops.text_pool_bwd(_t(dh), _t(pre.reshape(B * T, H)), _t(ids), H, V, _t(keep), 0.5, dpre)
  np.testing.assert_allclose(dpre.cpu().numpy().reshape(B, T, H), want_dpre, rtol=1e-6, atol=1e-6)
  ops.text_pool_fwd(_t(pre.reshape(B * T, H)), _t(ids), H, V, None, 1.0, hidden)      # evaluation
  np.testing.assert_allclose(hidden.cpu().numpy(), np.maximum(y, 0), rtol=1e-6, atol=1e-6)


def _setup(tmp_path, rng, vocab_size=150, classes=80):
  names = ["cls%02d" % i for i in range(classes)]
  vocab = names[:40] + ["w%03d" % i for i in range(vocab_size - 40)]
  emb = (0.4 * rng.standard_normal((len(vocab), 300))).astype(np.float32)
  lf, vf, ef = tmp_path / "labels.txt", tmp_path / "vocab.txt", tmp_path / "emb.npy"
  lf.write_text("\n".join(names)); vf.write_text("\n".join(vocab)); np.save(str(ef), emb)
  pipeline = util_model.load_pipeline("coco17_text_hotpath", label_file=str(lf),
                                      open_vocabulary_file=str(vf), open_vocabulary_word_embedding_file=str(ef))
  return pipeline, names, vocab, emb


def test_text_train_step_matches_oracle(tmp_path):
  from cap2det_amd.models import text_model
  rng = np.random.default_rng(9)
  pipeline, names, vocab, emb = _setup(tmp_path, rng)
  np.random.seed(11)                                             # OOV row draw (see label tests)
  tr = text_model.TextTrainer(pipeline, device=DEV, seed=5)
  m = tr.model
  assert isinstance(m, text_model.Model) and m.num_classes == 80
  B, T = 20, 33
  caps = [[vocab[i] for i in rng.integers(0, len(vocab), T - 5)] + ["zzz", "", "", "", ""] for _ in range(B)]
  caps[3] = ["qqq"] * T                                          # all-OOV caption
  objs = [[names[i] for i in rng.choice(80, 2, replace=False)] + [""] for _ in range(B)]
  ex = {"concat_caption_string": caps, "object_texts": objs}
  keep = (rng.uniform(size=(B, 400)) < 0.5).astype(np.uint8)
  st = m.state_dict()
  P = {"w1": st[text_model.W1].astype(np.float64), "b1": st[text_model.B1].astype(np.float64),
       "w2": st[text_model.W2].astype(np.float64), "b2": st[text_model.B2].astype(np.float64)}
  acc = {k: np.full(v.shape, 0.1) for k, v in P.items()}
  full = np.concatenate([emb, m._text_classifier._embedding[-1].cpu().numpy()[None]], 0).astype(np.float64)
  ids = ref_labels.tokens_to_ids(caps, vocab)
  labels = ref_labels.groundtruth_extract(objs, names).astype(np.float64)
  P0 = {k: v.copy() for k, v in P.items()}
  want = ref_text.train_step(P, acc, ids, full, labels, keep.astype(np.float64), 0.5, 1e-5, 0.1)

  losses = tr.train_step(ex, dropout_mask=torch.from_numpy(keep))
  torch.cuda.synchronize()
  got_logits = m._ctx["logits"][:, :80].cpu().numpy()
  assert np.abs(got_logits - want["logits"]).max() <= 1e-4
  np.testing.assert_allclose(losses["text_cross_entropy_loss"].item(), want["loss"], rtol=1e-5)
  np.testing.assert_allclose(losses["regularization_loss"].item(), want["reg_loss"], rtol=1e-4)
  np.testing.assert_allclose(losses["total_loss"].item(), want["loss"] + want["reg_loss"], rtol=1e-5)
  g = {"w1": m.grads[text_model.W1][:300], "b1": m.grads[text_model.B1],
       "w2": m.grads[text_model.W2][:, :80], "b2": m.grads[text_model.B2][:80]}
  for k in g:
    w = want["grads"][k] - (1e-5 * P0[k] if k in ("w1", "w2") else 0)   # l2 is added in the Adagrad kernel
    scale = np.abs(w).max()
    assert np.abs(g[k].cpu().numpy() - w).max() <= 5e-4 * scale + 1e-9, k
  new = m.state_dict()
  for k, name in (("w1", text_model.W1), ("b1", text_model.B1), ("w2", text_model.W2), ("b2", text_model.B2)):
    np.testing.assert_allclose(new[name], P[k], rtol=1e-3, atol=2e-5, err_msg=k)
  assert float(m.vars[text_model.W1][300:].abs().max()) == 0.0       # padding rows stay zero
  assert float(m.vars[text_model.W2][:, 80:].abs().max()) == 0.0 if m._cpad > 80 else True


def test_text_model_learns_and_hands_weights_to_the_detector(tmp_path):
  """A few hundred steps on a separable toy task: the loss drops, precision/recall@k rise, and
  the trained weights drive TextClassifierMatchExtractor in the detection model."""
  from cap2det_amd.models import builder, label_extractor, text_model
  from cap2det_amd.protos import label_extractor_pb2, text_format
  rng = np.random.default_rng(12)
  pipeline, names, vocab, emb = _setup(tmp_path, rng, vocab_size=120, classes=8)
  tr = text_model.TextTrainer(pipeline, device=DEV, seed=1)

  def batch(n):
    caps, objs = [], []
    for _ in range(n):
      c = int(rng.integers(8))
      # class c is signalled by the word w(c): an open-vocabulary synonym, never the class name
      toks = ["w%03d" % c] + ["w%03d" % int(i) for i in rng.integers(20, 80, 6)] + [""]
      rng.shuffle(toks)
      caps.append(list(toks)); objs.append([names[c], ""])
    return {"concat_caption_string": caps, "object_texts": objs}

  first = None
  for step in range(300):
    l = tr.train_step(batch(20))["text_cross_entropy_loss"].item()
    first = l if first is None else first
  assert l < 0.25 * first
  ev_model = builder.build(pipeline.model, is_training=False, device=DEV)
  ev_model.load_state_dict(tr.model.state_dict())
  acc = text_model.MetricAccumulator()
  for _ in range(10):
    ex = batch(1)
    acc.update(ev_model.build_evaluation(ev_model.build_prediction(ex), ex))
  res = acc.result()
  assert res["metrics/precision_at_1"] >= 0.9 and res["metrics/recall_at_1"] >= 0.9
  assert res["metrics/recall_at_5"] >= res["metrics/recall_at_1"]
  assert set(res) >= {"metrics/precision_at_0.3", "metrics/recall_at_0.7", "metrics/precision_at_5"}
  # the detector's text-classifier extractor consumes exactly these variables
  cfg = label_extractor_pb2.LabelExtractor()
  tc = pipeline.model.ListFields()[0][1].text_classifier
  text_format.Merge("text_classifier_match_extractor { label_file: '%s' open_vocabulary_file: '%s' "
                    "open_vocabulary_word_embedding_file: '%s' hidden_units: 400 label_threshold: 0.5 }"
                    % (tc.label_file, tc.open_vocabulary_file, tc.open_vocabulary_word_embedding_file), cfg)
  ext = label_extractor.build_label_extractor(cfg, DEV)
  ext.load_weights(tr.model.state_dict())
  ex = batch(16)
  want = ref_labels.groundtruth_extract(ex["object_texts"], names)
  got = ext.extract_labels(ex).cpu().numpy()
  assert (got == want).mean() > 0.95
