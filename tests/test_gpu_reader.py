"""End-to-end GPU parity of the input pipeline: TFRecord files of
tf.Example protos (written here in the layout of dataset-tools/create_pascal_tf_record.py:
147-196) -> `get_input_fn` (native host decoding + HIP flip / resize / pad / batch rescale) vs
the independent oracle (pure-Python framing + proto decoding, libjpeg-turbo via Pillow, numpy
legacy-bilinear resize).  Images are compared BIT-EXACTLY: same fp32 operation order."""
import io
import os

import numpy as np
import pytest
import torch

from cap2det_amd.readers import tfrecord as T
from oracle import ref_reader as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _example(rng, image_id, h, w, nprop, nobj, caption):
  from PIL import Image
  y, x = np.mgrid[0:h, 0:w]
  img = np.clip(np.stack([128 + 90 * np.sin(x / 9.0 + c) * np.cos(y / 5.0 - c) for c in range(3)], -1)
                + rng.normal(0, 10, (h, w, 3)), 0, 255).astype(np.uint8)
  b = io.BytesIO()
  Image.fromarray(img).save(b, format="JPEG", quality=90)

  def boxes(n):
    lo = rng.uniform(0, 0.6, (n, 2)); sz = rng.uniform(0.1, 0.4, (n, 2))
    return np.concatenate([lo, np.minimum(lo + sz, 1.0)], 1).astype(np.float32)

  pb, ob = boxes(nprop), boxes(nobj)
  texts = [("cls%d" % (i % 3)).encode() for i in range(nobj)]
  f = {
      "image/height": (T.INT64, [h]), "image/width": (T.INT64, [w]),
      "image/source_id": (T.BYTES, [image_id.encode()]),
      "image/encoded": (T.BYTES, [b.getvalue()]), "image/format": (T.BYTES, [b"jpeg"]),
      "image/object/bbox/ymin": (T.FLOAT, ob[:, 0].tolist()), "image/object/bbox/xmin": (T.FLOAT, ob[:, 1].tolist()),
      "image/object/bbox/ymax": (T.FLOAT, ob[:, 2].tolist()), "image/object/bbox/xmax": (T.FLOAT, ob[:, 3].tolist()),
      "image/object/class/text": (T.BYTES, texts), "image/object/class/label": (T.INT64, list(range(nobj))),
      "image/caption/string": (T.BYTES, [t.encode() for t in caption]),
      "image/caption/offset": (T.INT64, [0, 2] if len(caption) > 2 else [0]),
      "image/caption/length": (T.INT64, [2, len(caption) - 2] if len(caption) > 2 else [len(caption)]),
      "image/proposal/bbox/ymin": (T.FLOAT, pb[:, 0].tolist()), "image/proposal/bbox/xmin": (T.FLOAT, pb[:, 1].tolist()),
      "image/proposal/bbox/ymax": (T.FLOAT, pb[:, 2].tolist()), "image/proposal/bbox/xmax": (T.FLOAT, pb[:, 3].tolist()),
  }
  return T.encode_example(f)


def _reader_options(pattern, training, batch, extra=""):
  from cap2det_amd.protos import reader_pb2, text_format
  opt = reader_pb2.Reader()
  text_format.Merge("""
    cap2det_reader {
      input_pattern: "%s"
      interleave_cycle_length: 2
      is_training: %s
      shuffle_buffer_size: 5
      map_num_parallel_calls: 3
      batch_size: %d
      image_resizer { keep_aspect_ratio_resizer { min_dimension: 48 } }
      max_num_proposals: 9
      %s
    }""" % (pattern, "true" if training else "false", batch, extra), opt)
  return opt.cap2det_reader


def test_eval_reader_matches_oracle(tmp_path):
  from cap2det_amd.readers import cap2det_reader
  rng = np.random.default_rng(31)
  sizes = [(40, 60), (55, 36), (33, 33), (64, 50), (37, 70)]
  recs = [_example(rng, "%06d" % i, h, w, 5 + 3 * i, 1 + i % 3, ["a", "cat", "on", "mat"][:2 + i % 3])
          for i, (h, w) in enumerate(sizes)]
  T.write_records(str(tmp_path / "part-0.record"), recs[:3])
  T.write_records(str(tmp_path / "part-1.record"), recs[3:])
  opt = _reader_options(str(tmp_path / "part-*.record"), False, 2)
  batches = list(cap2det_reader.get_input_fn(opt, device=DEV)())
  assert len(batches) == 2                                     # drop_remainder: 5 examples -> 2 x 2
  # interleave(cycle_length=2): one record from each file in turn
  order = [0, 3, 1, 4]
  ids = [i for b in batches for i in b["image_id"]]
  assert ids == ["%06d" % i for i in order]
  for bi, batch in enumerate(batches):
    sel = [recs[i] for i in order[2 * bi:2 * bi + 2]]
    want = R.process_batch(sel, 48, 9, [False, False], None)
    np.testing.assert_array_equal(batch["image"].cpu().numpy(), want["image"])      # bit exact
    np.testing.assert_array_equal(batch["image_shape"], want["image_shape"])
    np.testing.assert_array_equal(batch["proposals"].cpu().numpy(), want["proposals"])
    np.testing.assert_array_equal(batch["number_of_proposals"].cpu().numpy(), want["number_of_proposals"])
    np.testing.assert_array_equal(batch["object_boxes"], want["object_boxes"])
    assert batch["proposals"].shape == (2, 9, 4)
    h, w = want["image_shape"][:, 0], want["image_shape"][:, 1]
    assert batch["image"].shape[1:3] == (h.max(), w.max())
    # zero padding outside every image
    for i in range(2):
      assert float(batch["image"][i, h[i]:].abs().max() if h[i] < h.max() else 0) == 0
  b0 = batches[0]
  assert b0["concat_caption_string"][0][:2] == ["a", "cat"]
  assert b0["caption_strings"][0][0] == ["a", "cat"] + [""] * (len(b0["caption_strings"][0][0]) - 2)
  assert b0["object_texts"][1][0] == "cls0"


def test_training_reader_replays_through_oracle(tmp_path):
  """flip + random batch rescale + shard filter; the decisions the reader took are replayed
  through the oracle."""
  from cap2det_amd.readers import cap2det_reader
  rng = np.random.default_rng(32)
  recs = {"%06d" % i: _example(rng, "%06d" % i, 40 + 3 * i, 52 - 2 * i, 12, 2, ["x", "y"])
          for i in range(8)}
  T.write_records(str(tmp_path / "t.record"), list(recs.values()))
  extra = """preprocess_options { random_flip_left_right_prob: 0.5 }
             batch_resize_scale_value: 1.2 batch_resize_scale_value: 0.6
             shard_indicator: "0/2" """
  opt = _reader_options(str(tmp_path / "t.record"), True, 2, extra)
  it = cap2det_reader.get_input_fn(opt, device=DEV, seed=7)()
  seen_flip, seen_scale = set(), set()
  for _ in range(12):
    batch = next(it)                                            # repeats forever in training
    ids = batch["image_id"]
    assert all(T.to_hash_bucket(i, 2) == 0 for i in ids)        # shard 0/2 only
    flips, scale = batch["_flip_left_right"], batch["_batch_scale"]
    seen_flip.update(flips); seen_scale.add(scale)
    want = R.process_batch([recs[i] for i in ids], 48, 9, flips, scale)
    np.testing.assert_array_equal(batch["image"].cpu().numpy(), want["image"])
    np.testing.assert_array_equal(batch["image_shape"], want["image_shape"])
    np.testing.assert_array_equal(batch["proposals"].cpu().numpy(), want["proposals"])
  assert seen_flip == {True, False} and sorted(round(v, 3) for v in seen_scale) == [0.6, 1.2]


def test_reader_feeds_the_model(tmp_path):
  """A batch from the reader drives one training step of the model (plumbing check)."""
  from cap2det_amd.readers import cap2det_reader
  from cap2det_amd.train.trainer import Trainer
  from tests import util_model
  rng = np.random.default_rng(33)
  recs = [_example(rng, "%06d" % i, 50, 64, 9, 2, ["person", "dog"]) for i in range(2)]
  T.write_records(str(tmp_path / "m.record"), recs)
  opt = _reader_options(str(tmp_path / "m.record"), False, 2)
  batch = next(cap2det_reader.get_input_fn(opt, device=DEV)())
  batch["object_texts"] = [["person", "dog"], ["cat", ""]]
  trainer = Trainer(util_model.load_pipeline(), device=DEV, depth_multiplier=0.5)
  losses = trainer.train_step(batch, dropout_seed=0)
  assert np.isfinite(float(losses["total_loss"].item()))


def test_evaluation_loop_and_checkpoint_round_trip(tmp_path):
  """TFRecords -> reader (eval mode) -> multi-scale inference + NMS -> PASCAL evaluator
  (train/predict.py:328-420), and trainer checkpoints (save, keep-best, resume bit-exactly)."""
  from cap2det_amd.models import builder
  from cap2det_amd.readers import cap2det_reader
  from cap2det_amd.train import evaluation as ev
  from cap2det_amd.train.trainer import Trainer
  from tests import util_model
  rng = np.random.default_rng(41)
  recs = [_example(rng, "%06d" % i, 50 + 4 * i, 64 - 3 * i, 9, 2, ["a", "b"]) for i in range(3)]
  T.write_records(str(tmp_path / "e.record"), recs)
  opt = _reader_options(str(tmp_path / "e.record"), False, 1)
  pipeline = util_model.load_pipeline()
  model = builder.build(pipeline.model, is_training=False, device=DEV, depth_multiplier=0.5)
  m = model._model_proto
  del m.eval_min_dimension[:]
  m.eval_min_dimension.extend([48, 40])
  classes = model.label_extractor.classes
  cats = [{'id': i + 1, 'name': c} for i, c in enumerate(classes)]
  cat2id = {c['name']: c['id'] for c in cats}

  def batches():
    for b in cap2det_reader.get_input_fn(opt, device=DEV)():
      # the fixture's object texts are cls0..2: rename them to real class names
      b["object_texts"] = [[classes[int(t[3:])] if t else "" for t in row] for row in b["object_texts"]]
      yield b

  evaluators = [ev.PascalDetectionEvaluator(cats) for _ in range(4)]
  metrics = ev.run_evaluation(model, batches(), evaluators, cat2id)
  assert len(metrics) == 4
  for mt in metrics:
    v = mt['PascalBoxes_Precision/mAP@0.5IOU']
    assert 0.0 <= v <= 1.0
    assert 0.0 <= mt['PascalBoxes_Precision/meanCorLoc@0.5IOU'] <= 1.0
  # checkpoints: train 2 steps, save, train 1 more; resume from the save and repeat the step
  tr = Trainer(pipeline, device=DEV, depth_multiplier=0.5, seed=3)
  ex = util_model.make_examples(rng, 1, 48, 48, 6, [6], classes)
  dev = dict(ex)
  for k in ("image", "proposals"):
    dev[k] = torch.from_numpy(ex[k]).to(DEV)
  dev["number_of_proposals"] = torch.from_numpy(ex["number_of_proposals"]).to(DEV)
  for i in range(2):
    tr.train_step(dev, dropout_seed=i)
  path = tr.save_checkpoint(str(tmp_path / "model_dir"))
  assert path.endswith("model.ckpt-2")
  l3 = float(tr.train_step(dev, dropout_seed=2)["total_loss"].item())
  w3 = tr.model.store.values.clone()
  tr2 = Trainer(pipeline, device=DEV, depth_multiplier=0.5, seed=99)     # different init
  tr2.load_checkpoint(path)
  assert tr2.global_step == 2
  l3b = float(tr2.train_step(dev, dropout_seed=2)["total_loss"].item())
  np.testing.assert_allclose(l3b, l3, rtol=1e-5)
  np.testing.assert_allclose(tr2.model.store.values.cpu().numpy(), w3.cpu().numpy(), rtol=1e-4, atol=1e-6)
  step_best, metric_best = ev.save_model_if_it_is_better(2, 0.5, path, str(tmp_path / "best"))
  assert (step_best, metric_best) == (2, 0.5)
  assert ev.get_best_model_checkpoint(str(tmp_path / "best")).endswith("model.ckpt-2")


def test_trainer_main_and_predict_cli(tmp_path):
  """The reference's two entry points with their flags (train/trainer_main.py:15-20,
  train/predict.py:36-78): records -> `trainer_main` (2 steps, checkpoint, resume for 1 more)
  -> `predict --run_once` (multi-scale inference, NMS, PASCAL and COCO evaluators)."""
  from cap2det_amd.train import predict, trainer_main
  from tests import util_model
  rng = np.random.default_rng(77)
  classes = open(os.path.join(util_model.ROOT, "cap2det_amd", "data", "voc_label.txt")).read().split("\n")
  classes = [c for c in classes if c]
  recs = []
  for i in range(4):
    rec = _example(rng, "%06d" % i, 52 + 2 * i, 60, 8, 2, ["a", "b"])
    recs.append(rec)
  T.write_records(str(tmp_path / "t.record"), recs)
  text = open(os.path.join(util_model.ROOT, "configs", "voc07_groundtruth_hotpath.pbtxt")).read()
  text = text.replace("cap2det_amd/data/", os.path.join(util_model.ROOT, "cap2det_amd", "data") + "/")
  reader = """
    cap2det_reader {
      input_pattern: "%s"
      interleave_cycle_length: 1
      is_training: %s
      shuffle_buffer_size: 4
      map_num_parallel_calls: 2
      batch_size: 1
      max_num_proposals: 8
      image_resizer { keep_aspect_ratio_resizer { min_dimension: 48 } }
    }"""
  pattern = str(tmp_path / "t.record")
  text += "\ntrain_reader {%s}\neval_reader {%s}\n" % (reader % (pattern, "true"), reader % (pattern, "false"))
  text = text.replace("eval_min_dimension: 1200", "").replace("eval_min_dimension: 800", "") \
             .replace("eval_min_dimension: 600", "eval_min_dimension: 48").replace("eval_min_dimension: 400", "eval_min_dimension: 40")
  cfg = tmp_path / "pipeline.pbtxt"
  cfg.write_text(text)
  model_dir = str(tmp_path / "model_dir")

  # the fixture's object texts are cls0..2: give the records real class names instead
  import cap2det_amd.readers.cap2det_reader as cr
  orig = cr.get_input_fn

  def renamed(options, **kw):
    fn = orig(options, **kw)

    def gen():
      for b in fn():
        b["object_texts"] = [[classes[int(t[3:])] if t else "" for t in row] for row in b["object_texts"]]
        yield b
    return gen
  cr.get_input_fn = renamed
  try:
    tr = trainer_main.main(["--pipeline_proto", str(cfg), "--model_dir", model_dir, "--max_steps", "2",
                            "--depth_multiplier", "0.5"])
    assert tr.global_step == 2 and os.path.exists(os.path.join(model_dir, "model.ckpt-2.npz"))
    tr = trainer_main.main(["--pipeline_proto", str(cfg), "--model_dir", model_dir, "--max_steps", "3",
                            "--depth_multiplier", "0.5"])
    assert tr.global_step == 3 and trainer_main.latest_checkpoint(model_dir).endswith("model.ckpt-3")
    label_file = os.path.join(util_model.ROOT, "cap2det_amd", "data", "voc_label.txt")
    for evaluator, key in (("pascal", "PascalBoxes_Precision/mAP@0.5IOU"), ("coco", "DetectionBoxes_Precision/mAP")):
      metrics, metric = predict.main(["--pipeline_proto", str(cfg), "--model_dir", model_dir,
                                      "--label_file", label_file, "--run_once", "--evaluator", evaluator,
                                      "--results_dir", str(tmp_path / "results"), "--depth_multiplier", "0.5"])
      assert len(metrics) == 4 and key in metrics[-1] and (metric != metric or -1.0 <= metric <= 1.0)
    assert len(os.listdir(str(tmp_path / "results"))) == 1
  finally:
    cr.get_input_fn = orig


def test_train_overlaps_the_uploads():
  """ADVICE r4 / VERDICT r4 weak #8: Trainer.train's input thread uploads batch k+1 on a copy stream
  that never waits for the compute stream, so the upload runs beside step k-1 instead of between two
  steps.  Every batch here drags a 64 MB payload across PCIe from pinned memory; events around that
  copy (on the input thread's stream) and behind every step (compute stream) must show the upload of
  batch k+1 STARTING before step k-1 has ended, and the whole run must hide most of the copy time."""
  import time
  from cap2det_amd.train.trainer import Trainer
  from tests import util_model
  rng = np.random.default_rng(5)
  trainer = Trainer(util_model.load_pipeline(), device=DEV, depth_multiplier=0.5)
  classes = trainer.model.label_extractor.classes
  n, steps = 256, 16
  ex = util_model.make_examples(rng, 1, 192, 192, n, [n], classes)
  host = {k: torch.from_numpy(ex[k]).pin_memory() for k in ("image", "proposals", "number_of_proposals")}
  payload = torch.empty(64 << 20, dtype=torch.uint8).pin_memory()
  from cap2det_amd import hip_ops
  probe_src, probe_dst = torch.ones(64, device=DEV), torch.zeros(64, device=DEV, dtype=torch.bfloat16)
  ups = []

  def batches(count, with_payload):
    for _ in range(count):
      b = dict(ex)
      s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      s.record()
      for k, v in host.items():
        b[k] = v.to(DEV, non_blocking=True)
      if with_payload:
        b["_payload"] = payload.to(DEV, non_blocking=True)
      hip_ops.cast_bf16(probe_src, probe_dst)     # a C-ABI call of the INPUT thread (a reader's resize would be)
      e.record()
      ups.append((s, e))
      yield b

  ends = []

  def log(step, losses):
    ev = torch.cuda.Event(enable_timing=True)
    ev.record()
    ends.append(ev)

  def run(with_payload):
    del ups[:], ends[:]
    trainer.global_step = 0
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    trainer.train(batches(steps, with_payload), max_steps=steps, log=log)
    torch.cuda.synchronize()
    return time.perf_counter() - t0

  run(True)                                   # warm-up: buffers, pinned staging, allocator pools
  t_plain = min(run(False) for _ in range(2))
  t_fed = min(run(True) for _ in range(2))    # (as the plain runs: the first run after a change of
                                              #  batch signature pays one-off allocations)
  assert len(ends) == steps and len(ups) >= steps
  copy_ms = sum(s.elapsed_time(e) for s, e in ups[:steps])
  ahead = sum(1 for k in range(2, steps - 1) if ends[k - 1].elapsed_time(ups[k + 1][0]) < 0.0)
  assert ahead >= 0.8 * (steps - 3), (ahead, steps)          # upload k+1 starts before step k-1 ends
  assert copy_ms > 0.2 * 1e3 * t_plain, (copy_ms, t_plain)    # (the payload is not negligible)
  assert t_fed < t_plain + 0.6 * copy_ms * 1e-3, (t_fed, t_plain, copy_ms)
  # the steps were replayed from step plans, and a plan holds the TRAINING thread's calls only: the
  # input thread's events and C-ABI calls (here: the probe cast, absent from an fp32 step) stay out
  assert trainer.plan_replays >= 3 * (steps - 4)
  plans = [st["plan"] for st in trainer._plans.values() if st["plan"] is not None]
  assert plans
  for plan in plans:
    names = set(node[1] for node in plan.nodes if node[0] == "call")
    assert "c2d_cast_bf16" not in names and "c2d_conv_fwd" in names, sorted(names)[:8]
