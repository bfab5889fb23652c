"""`python -m cap2det_amd.train.trainer_main --pipeline_proto X.pbtxt --model_dir DIR`

The flags of the reference's train/trainer_main.py:15-20 over this package's Trainer: the
unchanged `.pbtxt` is parsed, `--model_dir` overrides `pipeline.model_dir`, training resumes from
the newest checkpoint found there, batches come from `pipeline.train_reader` and a checkpoint is
written every `train_config.save_checkpoints_steps` steps (keeping `keep_checkpoint_max`).
One process per GPU under `python -m torch.distributed.run` (RANK / LOCAL_RANK / WORLD_SIZE):
rank k reads shard `k/G` of the records (the reference's `shard_indicator`,
readers/cap2det_reader.py:201-211) and gradients are averaged over RCCL.  The reference's
evaluation half (`train_and_evaluate`'s EvalSpec) is `cap2det_amd.train.predict`.
"""
import argparse
import glob
import os
import re
import sys
import time


def load_pipeline_proto(filename):
  from cap2det_amd.protos import pipeline_pb2, text_format
  pipeline_proto = pipeline_pb2.Pipeline()
  with open(filename, "r") as fp:
    text_format.Merge(fp.read(), pipeline_proto)
  return pipeline_proto


def latest_checkpoint(model_dir):
  """tf.train.latest_checkpoint over `model.ckpt-<step>` prefixes (.npz of this package or a
  TensorFlow V2 `.index`)."""
  best, best_step = None, -1
  for path in glob.glob(os.path.join(model_dir, "model.ckpt-*")):
    m = re.match(r"^(.*model\.ckpt-(\d+))(\.npz|\.index)$", path)
    if m and int(m.group(2)) > best_step:
      best, best_step = m.group(1), int(m.group(2))
  return best


def _prune(model_dir, keep):
  steps = sorted({int(m.group(1)) for p in glob.glob(os.path.join(model_dir, "model.ckpt-*.npz"))
                  for m in [re.search(r"model\.ckpt-(\d+)\.npz$", p)] if m})
  for step in steps[:-keep] if keep > 0 else []:
    os.remove(os.path.join(model_dir, "model.ckpt-%d.npz" % step))


def main(argv=None):
  ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
  ap.add_argument("--pipeline_proto", required=True, help="Path to the pipeline proto file.")
  ap.add_argument("--model_dir", default="", help="Directory which holds model checkpoints.")
  ap.add_argument("--type", default="", help="A message string passed from command-line.")
  ap.add_argument("--max_steps", type=int, default=None, help="Overrides train_config.max_steps.")
  ap.add_argument("--compute_dtype", choices=["fp32", "bf16"], default="fp32")
  ap.add_argument("--depth_multiplier", type=float, default=1.0, help=argparse.SUPPRESS)
  args = ap.parse_args(argv)

  import torch
  import torch.distributed as dist
  from cap2det_amd.readers import cap2det_reader
  from cap2det_amd.train.trainer import Trainer

  pipeline_proto = load_pipeline_proto(args.pipeline_proto)
  if args.model_dir:
    pipeline_proto.model_dir = args.model_dir
  model_dir = pipeline_proto.model_dir
  world = int(os.environ.get("WORLD_SIZE", "1"))
  rank = int(os.environ.get("RANK", "0"))
  local_rank = int(os.environ.get("LOCAL_RANK", "0"))
  # (before the first GPU call: the runtime reads the IPC mode when it initialises)
  os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
  os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
  torch.cuda.set_device(local_rank)
  if world > 1:
    dist.init_process_group(backend="nccl", rank=rank, world_size=world)
    # a rank shares its CUs with RCCL's channel kernels under the backward pass: size the one-round
    # launch budgets for 224 of the 256 (free when nothing is taken, profiles/r06_cu_withhold.json)
    from cap2det_amd import _lib
    _lib.call("c2d_set_available_cus", 224)
    reader = pipeline_proto.train_reader.cap2det_reader
    if not reader.shard_indicator:
      reader.shard_indicator = "%d/%d" % (rank, world)
  device = "cuda:%d" % local_rank
  trainer = Trainer(pipeline_proto, device=device, compute_dtype=args.compute_dtype,
                    depth_multiplier=args.depth_multiplier)
  if model_dir:
    ckpt = latest_checkpoint(model_dir)
    if ckpt:
      trainer.load_checkpoint(ckpt)
      print("restored %s (global_step %d)" % (ckpt, trainer.global_step), file=sys.stderr)
  tc = pipeline_proto.train_config
  input_fn = cap2det_reader.get_input_fn(pipeline_proto.train_reader.cap2det_reader, device=device,
                                         seed=trainer.global_step + rank)
  t0 = [time.perf_counter(), trainer.global_step]

  def log(step, losses):
    if tc.log_step_count_steps and step % tc.log_step_count_steps == 0 and rank == 0:
      now = time.perf_counter()
      print("step %d: total_loss %.6f (%.2f steps/s)" % (
          step, float(losses["total_loss"]), (step - t0[1]) / max(now - t0[0], 1e-9)), file=sys.stderr)
      t0[0], t0[1] = now, step
    if model_dir and rank == 0 and tc.save_checkpoints_steps and step % tc.save_checkpoints_steps == 0:
      trainer.save_checkpoint(model_dir)
      _prune(model_dir, tc.keep_checkpoint_max)

  def forever():
    while True:                        # the reference's dataset repeats (readers/...:251-253)
      n = 0
      for batch in input_fn():
        n += 1
        yield batch
      if n == 0:
        raise ValueError("train_reader matched no records")

  limit = args.max_steps if args.max_steps is not None else (tc.max_steps or None)
  trainer.train(forever(), max_steps=limit, log=log)
  if model_dir and rank == 0:
    trainer.save_checkpoint(model_dir)
    _prune(model_dir, tc.keep_checkpoint_max)
  if world > 1:
    dist.destroy_process_group()
  return trainer


if __name__ == "__main__":
  main()
