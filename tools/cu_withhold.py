"""SCALE readiness: what a training step costs when a collective's channel kernels hold CUs.

On a one-GPU box RCCL moves no data, so the rehearsal is a stand-in: c2d_debug_hold_cus puts W
workgroups of 1024 threads + the whole LDS of a CU each on a stream of their own for the length of
the timed steps (workgroup count, not CU masks).  Per config and W in {0, 8, 16, 32}: ms per step
(GPU events on the compute stream) with the launch plans sized for all 256 CUs and, beside it, with
c2d_set_available_cus(256 - W).

  python tools/cu_withhold.py [c1 c2] > profiles/r06_cu_withhold.json
"""
import json
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cap2det_amd import _lib, synthetic  # noqa: E402
from cap2det_amd.train.trainer import Trainer  # noqa: E402

STEPS = 40
SKIP = 5          # (the steps during which the hold kernel's workgroups find their CUs)
bench.pin_rank_to_cores(0, 1)
torch.set_num_threads(8)
out = {"tool": "tools/cu_withhold.py", "steps": STEPS, "rows": []}
hold_stream = torch.cuda.Stream()
for cfg in sys.argv[1:] or ["c1", "c2"]:
  spec = synthetic.BASELINE_CONFIGS[cfg]
  pipeline = synthetic.baseline_pipeline(cfg, tempfile.mkdtemp())
  for held in (0, 8, 16, 32):
    for sized in ((256,) if held == 0 else (256, 256 - held)):
      _lib.call("c2d_set_available_cus", sized)
      tr = Trainer(pipeline, device="cuda:0", seed=1, compute_dtype=spec["dtype"], allow_missing_pretrained=True)
      batch, _ = bench.synthetic_batch(1000, "cuda:0", tr.model.label_extractor.classes, pipeline)
      for i in range(8):
        tr.train_step(batch, dropout_seed=i, prefetch=batch)
      torch.cuda.synchronize()
      est_us = int((12000 if spec["dtype"] == "fp32" else 3600) * (STEPS + 2) * 1.25)
      _lib.call("c2d_debug_hold_cus", held, est_us, hold_stream.cuda_stream)
      marks = [torch.cuda.Event(enable_timing=True) for _ in range(STEPS + 1)]
      marks[0].record()
      for i in range(STEPS):
        tr.train_step(batch, dropout_seed=100 + i, prefetch=batch)
        marks[i + 1].record()
      marks[-1].synchronize()
      still_held = held == 0 or not hold_stream.query()       # the hold outlasted the timed steps
      torch.cuda.synchronize()
      per = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(SKIP, STEPS))
      row = {"config": cfg, "cus_withheld": held, "available_cus_setting": sized,
             "ms_per_step_p50": per[len(per) // 2], "ms_per_step_p10": per[len(per) // 10],
             "ms_per_step_p90": per[(9 * len(per)) // 10],
             "hold_covered_the_timed_steps": bool(still_held), "steps_replayed": tr.plan_replays}
      out["rows"].append(row)
      print(json.dumps(row), file=sys.stderr)
      del tr
_lib.call("c2d_set_available_cus", 256)
print(json.dumps(out, indent=1))
