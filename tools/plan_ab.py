"""Where a replayed step differs from a Python-driven one in GPU time: A = Python-driven (use_plan
off), B = the plan trainer's own eager step (look-ahead copied into place, labels of the next batch
extracted under the step; recording disabled), C = replay.  ms per step over 60 back-to-back steps."""
import os, sys, time, tempfile, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cap2det_amd import synthetic
from cap2det_amd.train.trainer import Trainer

for cfg in sys.argv[1:] or ["c1", "c2"]:
    spec = synthetic.BASELINE_CONFIGS[cfg]
    pipeline = synthetic.baseline_pipeline(cfg, tempfile.mkdtemp())
    for rep in range(2):
        for mode in ("A", "B", "C"):
            tr = Trainer(pipeline, device="cuda:0", seed=1, compute_dtype=spec["dtype"],
                         allow_missing_pretrained=True, use_plan=mode != "A")
            batch, _ = bench.synthetic_batch(1000, "cuda:0", tr.model.label_extractor.classes, pipeline)
            if mode == "B":
                real = tr._plans.setdefault
                tr._plans = type("D", (dict,), {"setdefault": lambda self, k, v: dict.setdefault(self, k, dict(v, failed=True))})()
            for i in range(8): tr.train_step(batch, dropout_seed=i, prefetch=batch)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(60): tr.train_step(batch, dropout_seed=i, prefetch=batch)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            print(cfg, mode, "replays %3d" % tr.plan_replays, "%.3f ms/step" % ((t2 - t0) / 60 * 1e3))
            del tr
