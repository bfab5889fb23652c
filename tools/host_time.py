"""Host enqueue time per training step next to the GPU step time (is the step launch-bound?).
Measured on the GPU box: fp32 7.3 ms enqueue (the host runs into the full queue) for a 12.75 ms
step, bf16 mode 2.7 ms of host work for a 4.7 ms step."""
import sys, time, torch
sys.path.insert(0, ".")
import bench
from cap2det_amd.train.trainer import Trainer
from tests import util_model
for dtype in ("fp32", "bf16"):
    tr = Trainer(util_model.load_pipeline("voc07_groundtruth_hotpath"), device="cuda:0", seed=1, compute_dtype=dtype)
    batch, _ = bench.synthetic_batch(1000, "cuda:0", tr.model.label_extractor.classes)
    for i in range(3): tr.train_step(batch, dropout_seed=i, prefetch=batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(20): tr.train_step(batch, dropout_seed=i, prefetch=batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(dtype, "host enqueue ms/step %.2f  total ms/step %.2f" % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
