"""Generates the seeded golden vectors under tests/golden/*.npz FROM THE ORACLE.

The reference itself cannot be imported here (TensorFlow 1.15 / object_detection / cv2 are
not installable), so these vectors pin the oracle's restatement — they make
oracle drift visible (CPU test) and give the GPU parity tests fixed inputs/outputs.
Run from the repo root:  python tests/golden/gen_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_labels, ref_model, ref_ops  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def edge_boxes(rng, n):
  c = rng.uniform(0, 1, (n, 2))
  s = np.exp(rng.uniform(np.log(0.04), 0.0, (n, 2)))
  b = np.concatenate([np.clip(c - s / 2, 0, 1), np.clip(c + s / 2, 0, 1)], 1).astype(np.float32)
  b[0] = [0, 0, 0, 0]
  b[1] = [0, 0, 1, 1]
  b[2] = [0.5, 0.5, 0.5, 0.5]
  b[3] = [0.7, 0.2, 0.3, 0.9]
  b[4] = [0.25, 0.25, 1.0, 1.0]
  b[5] = [-0.1, -0.2, 0.5, 1.3]
  return b


def crop_case():
  rng = np.random.default_rng(1234)
  feat = np.maximum(rng.standard_normal((2, 9, 11, 16)), 0).astype(np.float32)
  boxes = edge_boxes(rng, 37)
  ind = rng.integers(0, 2, 37).astype(np.int32)
  crop = ref_ops.crop_and_resize(feat, boxes, ind, 14)
  pooled, arg = ref_ops.max_pool(crop, 2, 2, "VALID")
  dout = rng.standard_normal(pooled.shape).astype(np.float32)
  dcrop = ref_ops.max_pool_backward(crop.shape, arg, dout, 2, 2, "VALID")
  dfeat = ref_ops.crop_and_resize_grad_image(dcrop.astype(np.float64), boxes, ind, feat.shape)
  np.savez_compressed(os.path.join(OUT, "roi_crop_case.npz"), feat=feat, boxes=boxes, box_ind=ind,
                      crop_checksum=np.float64(crop.astype(np.float64).sum()), pooled=pooled,
                      argmax=arg, dout=dout, dfeat=dfeat.astype(np.float32))


def heads_case():
  rng = np.random.default_rng(1235)
  b, n, c, d, k = 2, 37, 5, 32, 3
  x = rng.standard_normal((b, n, d)).astype(np.float32)
  P = ref_model.init_head_params(rng, d, c, k, stddev=0.5)
  num = np.array([37, 20], np.int32)
  boxes = np.stack([edge_boxes(rng, n), edge_boxes(rng, n)])
  labels = np.array([[1, 0, 1, 0, 0], [0, 0, 0, 1, 0]], np.float32)
  cl, scores, proba, saved = ref_model.build_midn_network(num, x, P)
  pred = {"num_proposals": num, "proposal_boxes": boxes, "midn_class_logits": cl,
          "midn_proba_r_given_c": proba, "oicr_proposal_scores_at_0": scores}
  for i in range(k):
    pred["oicr_proposal_scores_at_%d" % (i + 1)] = ref_ops.fully_connected(
        x, P["oicr/iter%d/weights" % (i + 1)], P["oicr/iter%d/biases" % (i + 1)])
  opts = dict(midn_loss_weight=1.0, oicr_loss_weight=0.5, oicr_iterations=k,
              oicr_iou_threshold=0.6, oicr_use_proba_r_given_c=True)
  losses, grads = ref_model.build_loss(pred, labels, opts)
  out = dict(x=x, num=num, boxes=boxes, labels=labels, class_logits=cl, scores=scores, proba=proba)
  out.update({"P:" + kk: v for kk, v in P.items()})
  out.update({"pred:" + kk: v for kk, v in pred.items() if kk.startswith("oicr")})
  out.update({"loss:" + kk: np.float32(v) for kk, v in losses.items()})
  out.update({"grad:" + kk: v for kk, v in grads.items()})
  np.savez_compressed(os.path.join(OUT, "heads_case.npz"), **out)


def text_case():
  rng = np.random.default_rng(1236)
  b, t, v, e, h, c = 4, 9, 50, 300, 400, 7
  ids = rng.integers(0, v + 1, (b, t)).astype(np.int32)
  ids[2, :] = v
  q = lambda a: a.astype(np.float16).astype(np.float32)   # stored as fp16: keep values exact
  emb = q(0.4 * rng.standard_normal((v + 1, e)))
  w1 = q(rng.standard_normal((e, h)) / np.sqrt(e))
  b1 = (0.1 * rng.standard_normal(h)).astype(np.float32)
  w2 = q(rng.standard_normal((h, c)) / np.sqrt(h))
  b2 = (0.1 * rng.standard_normal(c)).astype(np.float32)
  exact = np.zeros((b, c), np.float32)
  exact[3, 2] = 1
  logits = ref_labels.text_classifier_logits(ids, emb, w1, b1, w2, b2)
  labels = ref_labels.text_classifier_match_extract(ids, exact, emb, w1, b1, w2, b2, 0.5)
  np.savez_compressed(os.path.join(OUT, "text_classifier_case.npz"), ids=ids,
                      emb=emb.astype(np.float16), w1=w1.astype(np.float16), b1=b1,
                      w2=w2.astype(np.float16), b2=b2, exact=exact, logits=logits, labels=labels)


def conv_case():
  rng = np.random.default_rng(1237)
  x = rng.standard_normal((3, 7, 7, 32)).astype(np.float32)
  w = (rng.standard_normal((3, 3, 32, 48)) / 17.0).astype(np.float32)
  out = {"x": x, "w": w}
  for s in (1, 2):
    y = ref_ops.conv2d(x.astype(np.float64), w.astype(np.float64), s)
    dy = rng.standard_normal(y.shape).astype(np.float32)
    dx, dw = ref_ops.conv2d_backward(x.astype(np.float64), w.astype(np.float64),
                                     dy.astype(np.float64), s)
    out.update({"y_s%d" % s: y.astype(np.float32), "dy_s%d" % s: dy,
                "dx_s%d" % s: dx.astype(np.float32), "dw_s%d" % s: dw.astype(np.float32)})
  np.savez_compressed(os.path.join(OUT, "conv_case.npz"), **out)


if __name__ == "__main__":
  crop_case()
  heads_case()
  text_case()
  conv_case()
  print("written to", OUT)
