"""Where do the small fill / copy launches of a training step come from?  Runs a few steps of the
benchmark trainer under torch.profiler with Python stacks and prints the callers of aten::fill_,
aten::zero_ and aten::copy_ (GPU box)."""
import os, sys, collections, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from cap2det_amd import synthetic
from cap2det_amd.train.trainer import Trainer

cfg = sys.argv[1] if len(sys.argv) > 1 else "c1"
spec = synthetic.BASELINE_CONFIGS[cfg]
tmp = tempfile.mkdtemp()
pipeline = synthetic.baseline_pipeline(cfg, tmp)
trainer = Trainer(pipeline, device="cuda:0", seed=1234, compute_dtype=spec["dtype"],
                  allow_missing_pretrained=True)
ex, _ = bench.synthetic_batch(1000, "cuda:0", trainer.model.label_extractor.classes, pipeline)
for i in range(3):
  trainer.train_step(ex, dropout_seed=i, prefetch=ex)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
  for i in range(2):
    trainer.train_step(ex, dropout_seed=3 + i, prefetch=ex)
  torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
  if ev.name in ("aten::fill_", "aten::zero_", "aten::copy_", "aten::zeros", "aten::clone"):
    stack = [f for f in (ev.stack or []) if "cap2det_amd" in f or "bench.py" in f]
    cnt[(ev.name, tuple(stack[:2]))] += 1
for (name, stack), c in cnt.most_common(40):
  print("%4.1f/step %-12s %s" % (c / 2.0, name, " <- ".join(s.split("/repo/")[-1] for s in stack)))
