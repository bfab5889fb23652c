"""Host time to QUEUE one training step (GPU idle at the start, so nothing blocks on a full queue),
Python-driven against replayed from a step plan, next to the GPU step time."""
import os, sys, time, tempfile, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cap2det_amd import synthetic
from cap2det_amd.train.trainer import Trainer
for cfg in sys.argv[1:] or ["c1", "c2"]:
    spec = synthetic.BASELINE_CONFIGS[cfg]
    pipeline = synthetic.baseline_pipeline(cfg, tempfile.mkdtemp())
    for use_plan in (False, True):
        tr = Trainer(pipeline, device="cuda:0", seed=1, compute_dtype=spec["dtype"], allow_missing_pretrained=True,
                     use_plan=use_plan)
        batch, _ = bench.synthetic_batch(1000, "cuda:0", tr.model.label_extractor.classes, pipeline)
        for i in range(6): tr.train_step(batch, dropout_seed=i, prefetch=batch)
        torch.cuda.synchronize()
        host = []
        r0, n0 = tr.replay_s, tr.plan_replays
        for i in range(20):
            t0 = time.perf_counter()
            tr.train_step(batch, dropout_seed=i, prefetch=batch)
            host.append(time.perf_counter() - t0)
            torch.cuda.synchronize()
        inside = (tr.replay_s - r0) / max(tr.plan_replays - n0, 1) * 1e3
        t0 = time.perf_counter()
        for i in range(20): tr.train_step(batch, dropout_seed=i, prefetch=batch)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        host.sort()
        print(cfg, "plan" if use_plan else "python", "replays %d" % tr.plan_replays,
              "host ms to queue one step (GPU idle): median %.3f min %.3f (inside c2d_plan_replay %.3f);  back to back: host %.2f  total %.2f ms/step"
              % (host[10] * 1e3, host[0] * 1e3, inside, (t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
        del tr
