"""Host enqueue time per training step next to the GPU step time (is the step launch-bound?)."""
import os, sys, time, tempfile, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from cap2det_amd import synthetic
from cap2det_amd.train.trainer import Trainer
for cfg in sys.argv[1:] or ["c1", "c2"]:
    spec = synthetic.BASELINE_CONFIGS[cfg]
    pipeline = synthetic.baseline_pipeline(cfg, tempfile.mkdtemp())
    tr = Trainer(pipeline, device="cuda:0", seed=1, compute_dtype=spec["dtype"], allow_missing_pretrained=True)
    batch, _ = bench.synthetic_batch(1000, "cuda:0", tr.model.label_extractor.classes, pipeline)
    for i in range(3): tr.train_step(batch, dropout_seed=i, prefetch=batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20): tr.train_step(batch, dropout_seed=i, prefetch=batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(cfg, "host enqueue ms/step %.2f  total ms/step %.2f" % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
    del tr
