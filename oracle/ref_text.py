"""CPU restatement (TEST INFRASTRUCTURE ONLY) of the text-classifier TRAINING step:
models/text_model.py:31-129, models/label_extractor.py:353-421 (is_training=True),
core/utils.py:63-79 (masked_maximum), slim.dropout, slim l2_regularizer, TF Adagrad.

Gradient conventions are TensorFlow's: reduce_max / reduce_min split the incoming gradient
equally among tied extrema (`_MinOrMaxGrad`), relu' = (y > 0).  Cross-checked against torch
autograd (amax / amin share that tie rule) in tests/test_oracle_vs_torch.py.  Parity unpinned in
the usual sense (TensorFlow cannot run here; the reference's label_extractor_test.py:133-219
needs GloVe / checkpoint files that are not in the checkout)."""
import numpy as np


def forward(ids, emb, w1, b1, w2, b2, keep_mask=None, keep_prob=1.0):
  """ids [B,T]; emb [V+1,E] (row V = OOV).  Returns logits and the tape."""
  oov = emb.shape[0] - 1
  x = emb[ids]                                              # [B,T,E]
  mu = (ids != oov).astype(x.dtype)[..., None]             # [B,T,1]
  pre = x @ w1 + b1                                         # [B,T,H]
  m = pre.min(axis=1, keepdims=True)
  z = (pre - m) * mu
  y = z.max(axis=1) + m[:, 0]                               # masked_maximum
  r = np.maximum(y, 0)
  h = r if keep_mask is None else r * keep_mask / keep_prob
  logits = h @ w2 + b2
  return logits, dict(x=x, mu=mu, pre=pre, m=m, z=z, y=y, h=h, keep_mask=keep_mask,
                      keep_prob=keep_prob)


def sigmoid_ce_mean(logits, labels):
  l = np.maximum(logits, 0) - logits * labels + np.log1p(np.exp(-np.abs(logits)))
  return l.mean(), (1.0 / (1.0 + np.exp(-logits)) - labels) / logits.size


def backward(dlogits, tape, w2):
  h, pre, m, z, mu = tape["h"], tape["pre"], tape["m"], tape["z"], tape["mu"]
  g = {}
  g["w2"] = h.T @ dlogits
  g["b2"] = dlogits.sum(0)
  dh = dlogits @ w2.T
  if tape["keep_mask"] is not None:
    dh = dh * tape["keep_mask"] / tape["keep_prob"]
  dy = dh * (tape["y"] > 0)                                 # [B,H]
  is_max = (z == z.max(axis=1, keepdims=True)).astype(pre.dtype)
  n_max = is_max.sum(axis=1, keepdims=True)
  is_min = (pre == m).astype(pre.dtype)
  n_min = is_min.sum(axis=1, keepdims=True)
  dz = dy[:, None, :] * is_max / n_max                      # d/dz of reduce_max
  dm = dy[:, None, :] - (dz * mu).sum(axis=1, keepdims=True)   # via "+ m" and "- m * mu"
  dpre = dz * mu + dm * is_min / n_min
  x = tape["x"]
  g["w1"] = np.einsum("bte,bth->eh", x, dpre)
  g["b1"] = dpre.sum(axis=(0, 1))
  return g, dpre


def train_step(P, acc, ids, emb, labels, keep_mask, keep_prob, reg, lr):
  """One Adagrad step in place on P / acc (dict of w1,b1,w2,b2)."""
  logits, tape = forward(ids, emb, P["w1"], P["b1"], P["w2"], P["b2"], keep_mask, keep_prob)
  loss, dlogits = sigmoid_ce_mean(logits, labels)
  grads, _ = backward(dlogits, tape, P["w2"])
  reg_loss = reg * 0.5 * ((P["w1"] ** 2).sum() + (P["w2"] ** 2).sum())
  for k in ("w1", "w2"):
    grads[k] = grads[k] + reg * P[k]
  for k in P:
    acc[k] += grads[k] ** 2
    P[k] -= lr * grads[k] / np.sqrt(acc[k])
  return dict(logits=logits, loss=loss, reg_loss=reg_loss, grads=grads)
