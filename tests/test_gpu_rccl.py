"""RCCL on hardware, rehearsed on the one GPU a pool box has: a CHILD process (a fresh HIP
context, as a rank of `torch.distributed.run` is) initialises a one-rank process group over
nccl (= RCCL on ROCm) and runs full-size training steps with the gradient collectives forced on
(C2D_FORCE_ALLREDUCE=1) against the same steps without them, and the per-block exchange
(data_parallel.BlockReducer, the RCCL default) against the two-bucket one —
tools/rccl_rehearsal.py.
Reference: one process per GPU, /root/reference/train_wsod.sh:46-88; the synchronous gradient
mean is its SyncReplicasOptimizer option (train/trainer.py:90-94)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  with socket.socket() as sock:
    sock.bind(("127.0.0.1", 0))
    return str(sock.getsockname()[1])


def test_forced_allreduce_at_world_size_one_reproduces_the_plain_step():
  env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
             MASTER_PORT=_free_port(), HSA_ENABLE_IPC_MODE_LEGACY="0")
  env.pop("C2D_FORCE_ALLREDUCE", None)
  r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_rehearsal.py")], env=env,
                     stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
  lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
  assert lines, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
  rep = json.loads(lines[-1])
  assert r.returncode == 0 and rep["ok"], rep
  assert rep["backend"] == "nccl" and rep["world_size"] == 1
  assert rep["ranks_counted_by_all_reduce"] == 1
  modes = {c["mode"]: c for c in rep["checks"]}
  assert modes["eager"]["collectives_with"] == 12
  assert all(c["first_forward_bitwise_equal"] for c in rep["checks"] if "first_forward_bitwise_equal" in c)
  # the per-block exchange on RCCL (collectives issued inside the filter-gradient stream) reduces
  # what the two-bucket form reduces (to the order of the step's remaining fp32 atomics)
  bv = modes["blocks_vs_two"]
  assert bv["ok"] and bv["gradient_diff_of_scale"] <= 1e-5 and bv["updated_variables_diff_of_scale"] <= 1e-5, bv
  assert bv["collectives_blocks"] == 4 and bv["collectives_two"] == 2


def test_bench_line_reports_the_process_group():
  """`bench.py` under the rehearsal switch: the JSON line carries what the process group saw."""
  env = dict(os.environ, C2D_FORCE_ALLREDUCE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port())
  for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
    env.pop(k, None)
  r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "2",
                      "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                     text=True, timeout=900)
  lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
  assert r.returncode == 0 and lines, (r.returncode, r.stderr[-4000:])
  out = json.loads(lines[-1])
  pg = out["process_group"]
  assert {k: pg[k] for k in ("backend", "world_size", "ranks_counted_by_all_reduce", "forced_at_one_rank")} == {
      "backend": "nccl", "world_size": 1, "ranks_counted_by_all_reduce": 1, "forced_at_one_rank": True}
  # round 6: per-rank step time and the GPU time inside the reducers' finish(), min / max over the ranks
  assert 0 < pg["step_ms_min_over_ranks"] <= pg["step_ms_max_over_ranks"]
  assert 0 <= pg["allreduce_exposed_ms_min_over_ranks"] <= pg["allreduce_exposed_ms_max_over_ranks"]
  assert pg["allreduce_exposed_ms_max_over_ranks"] < pg["step_ms_max_over_ranks"]
  assert out["available_cus"] == 256
  assert out["n_gpus"] == 1 and out["value"] > 0 and "RCCL" in out["config"]["workload"]
