"""Where the HOST time of a training step goes (cProfile over K steps of bench.py's batch).

  python tools/profile_host.py [c1|c2] [steps]
"""
import cProfile
import os
import pstats
import shutil
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cap2det_amd import synthetic  # noqa: E402
from cap2det_amd.train.trainer import Trainer  # noqa: E402


def main():
  cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
  steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
  spec = synthetic.BASELINE_CONFIGS[cfg]
  scratch = tempfile.mkdtemp()
  try:
    pipeline = synthetic.baseline_pipeline(cfg, scratch)
    trainer = Trainer(pipeline, device="cuda:0", seed=1, compute_dtype=spec["dtype"],
                      allow_missing_pretrained=True)
  finally:
    shutil.rmtree(scratch, ignore_errors=True)
  classes = trainer.model.label_extractor.classes
  batch, _ = bench.synthetic_batch(1000, "cuda:0", classes, pipeline)
  for i in range(5):
    trainer.train_step(batch, dropout_seed=i, prefetch=batch)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  for i in range(steps):
    trainer.train_step(batch, dropout_seed=10 + i, prefetch=batch)
  host = time.perf_counter() - t0
  torch.cuda.synchronize()
  total = time.perf_counter() - t0
  print("unprofiled: host enqueue %.3f ms per step, step %.3f ms" % (1e3 * host / steps, 1e3 * total / steps))
  # with an EMPTY queue in front of every step (no back-pressure from a GPU that is behind)
  host = 0.0
  for i in range(steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    trainer.train_step(batch, dropout_seed=50 + i, prefetch=batch)
    host += time.perf_counter() - t0
  torch.cuda.synchronize()
  print("host work per step with an empty queue: %.3f ms" % (1e3 * host / steps))
  prof = cProfile.Profile()
  prof.enable()
  for i in range(steps):
    trainer.train_step(batch, dropout_seed=100 + i, prefetch=batch)
  prof.disable()
  torch.cuda.synchronize()
  st = pstats.Stats(prof)
  st.sort_stats("tottime").print_stats(45)


if __name__ == "__main__":
  main()
