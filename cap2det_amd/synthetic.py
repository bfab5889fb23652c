"""Seeded synthetic inputs of SURVEY.md §8d and pipeline loading for the shipped hot-path configs.

Used by bench.py, `__graft_entry__.smoke()` and the tests; nothing here touches the oracle, so a
benchmark process only loads the oracle when its `cpu_baseline` leg runs.  The reference has no
equivalent: it always trains from TFRecords (readers/cap2det_reader.py); the distributions below
mirror what its data tools produce (SelectiveSearch boxes with sides >= 20 px,
dataset-tools/create_pascal_selective_search_data.py:91-101; tokenised captions padded with '').
"""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "cap2det_amd", "data")

# BASELINE.json `configs` -> shipped pipeline + the overrides the config line names
BASELINE_CONFIGS = {
    "c0": dict(pipeline="voc07_groundtruth_hotpath", proposals=300, dtype="fp32", classes=20,
               title="configs[0]: voc07_groundtruth, 300 proposals (plumbing case)"),
    "c1": dict(pipeline="voc07_groundtruth_hotpath", proposals=2000, dtype="fp32", classes=20,
               title="configs[1]: voc07_groundtruth, 2000 proposals, fp32"),
    "c2": dict(pipeline="coco17_extend_match_hotpath", proposals=2000, dtype="bf16", classes=80,
               title="configs[2]: coco17_extend_match (caption label_extractor + MIL), 2000 "
                     "proposals, bf16 storage / fp32 accumulate"),
    "c3": dict(pipeline="coco17_text_classifier_match_hotpath", proposals=2000, dtype="fp32",
               classes=80, vocab="coco_open_vocab.txt",
               title="configs[3]: coco17_text_classifier_match, 2000 proposals, fp32, DP + RCCL "
                     "gradient all-reduce"),
    "c4": dict(pipeline="flickr30k_text_classifier_match_hotpath", proposals=2000, dtype="bf16",
               classes=80, vocab="flickr30k_open_vocab.txt",
               title="configs[4]: flickr30k_text_classifier_match, 2000 proposals, bf16 storage / "
                     "fp32 accumulate, DP + RCCL gradient all-reduce"),
}


def load_pipeline(name="voc07_groundtruth_hotpath", **fields):
  """Parses configs/<name>.pbtxt.  Relative `cap2det_amd/data/...` paths are anchored at the
  repo root; `fields` overrides string fields by NAME wherever they occur, e.g.
  load_pipeline("coco17_extend_match_hotpath", label_file="/tmp/x.txt")."""
  from cap2det_amd.protos import pipeline_pb2, text_format
  text = open(os.path.join(ROOT, "configs", name + ".pbtxt")).read()
  for k, v in fields.items():
    text, count = re.subn(r"(\b%s\s*:\s*)'[^']*'" % re.escape(k),
                          lambda m: "%s'%s'" % (m.group(1), v), text)
    if count == 0:
      raise KeyError("field %s does not occur in %s.pbtxt" % (k, name))
  text = text.replace("'cap2det_amd/data/", "'" + DATA + "/")
  p = pipeline_pb2.Pipeline()
  text_format.Merge(text, p)
  return p


def synthetic_boxes(rng, n, min_side=0.04):
  """centre ~U(0,1), log-size ~U(log 0.04, log 1), clipped to [0,1]."""
  c = rng.uniform(0, 1, (n, 2))
  s = np.exp(rng.uniform(np.log(min_side), 0.0, (n, 2)))
  lo = np.clip(c - s / 2, 0, 1)
  hi = np.clip(c + s / 2, 0, 1)
  hi = np.maximum(hi, np.minimum(lo + min_side, 1.0))
  lo = np.minimum(lo, hi - min_side)
  return np.concatenate([lo, hi], axis=1).astype(np.float32)


def make_examples(rng, batch, h, w, n, num_proposals, classes, labels_per_image=2):
  image = rng.integers(0, 256, (batch, h, w, 3)).astype(np.float32)
  proposals = np.stack([synthetic_boxes(rng, n) for _ in range(batch)])
  num = np.asarray(num_proposals, np.int32)
  for b in range(batch):
    proposals[b, num[b]:] = 0.0          # padded_batch zero pad (readers/cap2det_reader.py:237)
  texts = []
  for b in range(batch):
    picks = rng.choice(len(classes), labels_per_image, replace=False)
    texts.append([classes[i] for i in picks] + ["", "not_a_class"])
  return dict(image=image, number_of_proposals=num, proposals=proposals, object_texts=texts)


def read_lines(path):
  with open(path, "r") as f:
    return [line.strip("\n") for line in f.readlines()]


def synthetic_captions(rng, batch, vocab, tokens=60, oov_fraction=0.15, must_contain=()):
  """T caption tokens ~ Zipf over the open vocabulary, 15 % OOV / '' padding.
  `must_contain`: words placed at random positions of every caption (so that a caption names
  at least one class, as a real COCO caption of a training image does)."""
  v = len(vocab)
  ranks = np.arange(1, v + 1, dtype=np.float64)
  p = (1.0 / ranks) / np.sum(1.0 / ranks)
  caps = []
  for _ in range(batch):
    ids = rng.choice(v, size=tokens, p=p)
    toks = [vocab[i] for i in ids]
    real = int(round(tokens * (1.0 - oov_fraction)))
    for i in range(real, tokens):
      toks[i] = "" if i % 2 else "zzzoov"        # '' padding and genuinely unknown words
    for w in must_contain:
      toks[int(rng.integers(0, real))] = w
    caps.append(toks)
  return caps


def write_text_classifier_assets(out_dir, vocab, num_classes, seed=4, hidden_units=400,
                                 embedding_dims=300):
  """The GloVe table and the text-classifier checkpoint are missing from the reference checkout
  (.MISSING_LARGE_BLOBS): synthetic stand-ins of the real shapes (table
  N(0, 0.4^2) [V, 300]).  Returns (embedding .npy path, weights .npz path)."""
  rng = np.random.default_rng(seed)
  emb = (0.4 * rng.standard_normal((len(vocab), embedding_dims))).astype(np.float32)
  w = {"text_classifier/layer1/weights":
           (rng.standard_normal((embedding_dims, hidden_units)) / np.sqrt(embedding_dims))
           .astype(np.float32),
       "text_classifier/layer1/biases": (0.1 * rng.standard_normal(hidden_units)).astype(np.float32),
       "text_classifier/layer2/weights":
           (rng.standard_normal((hidden_units, num_classes)) / 6.0).astype(np.float32),
       "text_classifier/layer2/biases": (rng.standard_normal(num_classes) - 1.0).astype(np.float32)}
  os.makedirs(out_dir, exist_ok=True)
  ef, wf = os.path.join(out_dir, "open_vocab_300d.npy"), os.path.join(out_dir, "text_classifier.npz")
  np.save(ef, emb)
  np.savez(wf, **w)
  return ef, wf


def baseline_pipeline(config, scratch_dir=None):
  """Pipeline proto of one BASELINE.json config ("c0".."c4") over the shipped label files; the
  text-classifier configs get synthetic embedding / classifier files under `scratch_dir`."""
  spec = BASELINE_CONFIGS[config]
  if "vocab" not in spec:
    return load_pipeline(spec["pipeline"])
  if scratch_dir is None:
    raise ValueError("config %s needs a scratch_dir for its synthetic GloVe / classifier files" % config)
  vocab = read_lines(os.path.join(DATA, spec["vocab"]))
  ef, wf = write_text_classifier_assets(scratch_dir, vocab, spec["classes"])
  return load_pipeline(spec["pipeline"], open_vocabulary_word_embedding_file=ef,
                       text_classifier_checkpoint_file=wf)


def caption_vocabulary(pipeline):
  """Words a synthetic caption for this pipeline is drawn from: the extractor's open vocabulary
  when it has one, else the shipped COCO open vocabulary."""
  from cap2det_amd.protos import cap2det_model_pb2
  m = pipeline.model.Extensions[cap2det_model_pb2.Cap2DetModel.ext]
  which = m.label_extractor.WhichOneof("label_extractor_oneof")
  opts = getattr(m.label_extractor, which)
  path = getattr(opts, "open_vocabulary_file", "") or os.path.join(DATA, "coco_open_vocab.txt")
  return read_lines(path)
