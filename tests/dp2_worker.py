"""Worker of tests/test_gpu_dp2.py: ONE rank of a two-rank data-parallel training step on the HIP
path (or the single-process batch-of-2 step it must reproduce).

Reference: /root/reference/train_wsod.sh:46-88 starts one worker process per GPU, each on its own
shard of the images; train/trainer.py:55-61 reduces every loss with reduce_mean over the batch, so
the mean of the per-image gradients of two one-image workers IS the gradient of the batch-of-2
step (the synchronous form: `SyncReplicasOptimizer`, train/trainer.py:90-94).

A pool box has one GPU: both ranks use cuda:0 and exchange over gloo (the code path of a real
multi-GPU run except for the transport: RCCL itself is rehearsed by tests/test_gpu_rccl.py).
Started fresh by torch.distributed.run (RANK / WORLD_SIZE / MASTER_* in the environment), never
exec'ed from a process that touched the GPU.

  python tests/dp2_worker.py OUT_PREFIX fp32|bf16 eager|graph dp|single [small|full]

`full`: the benchmark's own shape per rank (depth 1.0, one 500x500 image with 2000 proposals per
rank; the single-process reference steps on both images = 4000 ROIs).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)

DM, HW, N, NUMS, K = 0.5, (64, 80), 24, [24, 17], 3
STEPS = 2


def main():
  global DM, HW, N, NUMS
  out_prefix, dtype, launch, mode = sys.argv[1:5]
  if len(sys.argv) > 5 and sys.argv[5] == "full":
    DM, HW, N, NUMS = 1.0, (500, 500), 2000, [2000, 1741]
  import numpy as np
  import torch
  import torch.distributed as dist
  from tests import util_model
  from cap2det_amd.protos import cap2det_model_pb2
  from cap2det_amd.train.trainer import Trainer

  torch.cuda.set_device(0)
  dev = "cuda:0"
  rank, world = 0, 1
  if mode == "dp":
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
  pipeline = util_model.load_pipeline()
  assert launch == "eager"         # (injected dropout masks and data-parallel hooks: Python-driven steps)
  graph = False
  trainer = Trainer(pipeline, device=dev, depth_multiplier=DM, compute_dtype=dtype)
  model = trainer.model
  classes = model.label_extractor.classes
  P32, d = util_model.oracle_state(11, len(classes), K, DM, head_std=0.01 if DM == 1.0 else 0.05)
  model.load_state_dict(P32)
  rng = np.random.default_rng(31)
  losses_log = []
  for step in range(STEPS):
    ex = util_model.make_examples(rng, 2, HW[0], HW[1], N, NUMS, classes)
    mask = (rng.uniform(size=(2 * N, d)) < 0.5).astype(np.uint8)
    lo, hi = (rank, rank + 1) if mode == "dp" else (0, 2)
    sub = {}
    for k, v in ex.items():
      v = v[lo:hi]
      sub[k] = torch.from_numpy(v).to(dev) if isinstance(v, np.ndarray) and v.dtype.kind in "fiu" else v
    kw = {} if graph else dict(dropout_mask=torch.from_numpy(mask[lo * N:hi * N]).to(dev))
    losses = trainer.train_step(sub, **kw)
    torch.cuda.synchronize()
    losses_log.append({k: float(v) for k, v in losses.items()})
    if step == 0:
      blo, bhi = trainer.bucket
      np.savez(out_prefix + "_%s_r%d.npz" % (mode, rank),
               values=model.store.values[blo:bhi].cpu().numpy(),
               accum=model.store.accum[blo:bhi].cpu().numpy(),
               grads=model.store.grads[blo:bhi].cpu().numpy(),
               losses=np.array([losses_log[0][k] for k in sorted(losses_log[0])]),
               loss_names=np.array(sorted(losses_log[0])))
  # the later steps only have to run (fp rounding moves the discrete OICR selections) and to keep
  # the ranks in lock step: every rank holds the same variables after every step
  blo, bhi = trainer.bucket
  final = model.store.values[blo:bhi].clone()
  if mode == "dp":
    other = final.clone()
    dist.broadcast(other, src=0)
    assert torch.equal(other, final), "ranks diverged"
    dist.barrier()
    dist.destroy_process_group()
  assert all(np.isfinite(v) for l in losses_log for v in l.values())
  # (one write: the ranks share the launcher's stdout)
  sys.stdout.write("dp2_worker ok %s %d %r\n" % (mode, rank, losses_log[-1]["total_loss"]))
  sys.stdout.flush()
  with open(out_prefix + "_%s_r%d.done" % (mode, rank), "w") as f:
    f.write("ok\n")


if __name__ == "__main__":
  main()
