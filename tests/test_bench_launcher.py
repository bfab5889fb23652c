"""`python bench.py --gpus N` as a plain command (the shape of the driver's BENCH / SCALE
commands): the parent starts one rank per GPU through torch.distributed.run as a child process.
On this CPU box the ranks run the stubbed step (C2D_BENCH_STUB=1: launcher, rendezvous on
127.0.0.1, barrier, two-bucket gradient exchange over gloo, max-over-ranks timing, one JSON line
from rank 0) — the one-process-per-GPU layout of the reference's workers (train_wsod.sh:46-88)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *argv):
  env = dict(os.environ, **extra_env)
  env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
  return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=env,
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)


def test_plain_command_launches_two_ranks():
  r = _run({"C2D_BENCH_STUB": "1"}, "--gpus", "2", "--steps", "3", "--warmup", "1")
  assert r.returncode == 0, r.stderr[-2000:]
  lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
  assert len(lines) == 1, r.stdout                      # rank 0 only
  out = json.loads(lines[0])
  assert out["n_gpus"] == 2 and out["world_size"] == 2 and out["stub"] is True
  assert out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
  assert out["value"] > 0 and out["ms_per_step"] > 0


def test_plain_command_launches_eight_ranks():
  """SCALE readiness: the driver's 8-GPU command shape on this CPU box — eight ranks
  through the launcher, both gradient exchanges over the REAL per-block cuts of the bucket (the
  per-block reducer with uneven start() coverage on odd steps), one disjoint core set per rank chosen
  in-process and reported in the line."""
  r = _run({"C2D_BENCH_STUB": "1"}, "--gpus", "8", "--steps", "4", "--warmup", "1",
           "--nccl-max-nchannels", "4", "--available-cus", "224")
  assert r.returncode == 0, r.stderr[-2000:]
  lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
  assert len(lines) == 1, r.stdout
  out = json.loads(lines[0])
  assert out["n_gpus"] == 8 and out["world_size"] == 8 and out["stub"] is True
  # SCALE readiness, part 2: per-rank step time and the time the compute stream spends
  # inside the reducers' finish(), min / max over the ranks; the channel budget reaches the ranks
  pg = out["process_group"]
  assert pg["world_size"] == 8 and pg["nccl_max_nchannels"] == "4" and out["available_cus"] == 224
  assert 0 < pg["step_ms_min_over_ranks"] <= pg["step_ms_max_over_ranks"]
  assert 0 <= pg["allreduce_exposed_ms_min_over_ranks"] <= pg["allreduce_exposed_ms_max_over_ranks"]
  assert pg["allreduce_exposed_ms_max_over_ranks"] <= pg["step_ms_max_over_ranks"] * 1.01
  aff = out["cpu_affinity"]
  ncores = len(os.sched_getaffinity(0))
  if ncores >= 8:
    assert aff["cores_per_rank"] == ncores // 8 and aff["host_cores"] == ncores
    assert aff["this_rank"][1] - aff["this_rank"][0] + 1 >= aff["cores_per_rank"]


def test_rank_core_sets_are_disjoint(monkeypatch):
  sys.path.insert(0, ROOT)
  import bench
  cores = sorted(os.sched_getaffinity(0))
  try:
    seen = []
    for r in range(4):
      os.sched_setaffinity(0, cores)
      got = bench.pin_rank_to_cores(r, 4)
      if len(cores) < 4:
        assert got is None
        continue
      mine = sorted(os.sched_getaffinity(0))
      assert len(mine) == len(cores) // 4 and got["this_rank"] == [mine[0], mine[-1]]
      assert not set(mine) & set(seen)
      seen += mine
    os.sched_setaffinity(0, cores)
    assert bench.pin_rank_to_cores(0, 1) is None                 # one rank keeps every core
    assert sorted(os.sched_getaffinity(0)) == cores
  finally:
    os.sched_setaffinity(0, cores)


def test_single_rank_stub_runs_in_process():
  r = _run({"C2D_BENCH_STUB": "1"}, "--steps", "2", "--warmup", "0")
  assert r.returncode == 0, r.stderr[-2000:]
  out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
  assert out["n_gpus"] == 1 and out["world_size"] == 1


def test_failing_rank_propagates_exit_code():
  # WORLD_SIZE/--gpus mismatch inside the child: the parent must not report success
  r = _run({"C2D_BENCH_STUB": "1", "C2D_BENCH_STUB_FAIL": "1"}, "--gpus", "2", "--steps", "1",
           "--warmup", "0")
  assert r.returncode != 0


def test_without_gpu_and_without_stub_fails_loudly():
  import torch
  if torch.cuda.device_count() > 0:
    return
  r = _run({}, "--gpus", "2", "--steps", "1", "--warmup", "0")
  assert r.returncode != 0 and "no GPU" in r.stderr


def test_parent_counts_gpus_without_torch():
  """The launcher parent must stay GPU-free by construction: the device count comes from the KFD
  topology in sysfs (cap2det_amd/train/gpu_count.py), no torch / HIP import."""
  code = ("import sys; sys.path.insert(0, %r); "
          "from cap2det_amd.train.gpu_count import count_visible_gpus; "
          "n = count_visible_gpus(); assert isinstance(n, int) and n >= 0; "
          "assert 'torch' not in sys.modules; print(n)" % ROOT)
  r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                     text=True, timeout=60)
  assert r.returncode == 0, r.stderr
  env = dict(os.environ, HIP_VISIBLE_DEVICES="")
  r2 = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, text=True, timeout=60)
  assert r2.returncode == 0 and r2.stdout.strip() == "0"      # an empty device list hides every GPU


def test_gpu_count_fallbacks(monkeypatch, tmp_path):
  """without the KFD sysfs tree (a container with /dev/kfd passed through) the count
  falls back to the render nodes of /dev/dri, C2D_NUM_GPUS is an explicit override, and an
  out-of-range index in a *_VISIBLE_DEVICES list ends the list as it does in the HIP runtime."""
  from cap2det_amd.train import gpu_count
  real_listdir = os.listdir

  def fake_listdir(path):
    if path == "/sys/class/kfd/kfd/topology/nodes":
      raise OSError("not mounted")
    if path == "/dev/dri":
      return ["card0", "card1", "renderD128", "renderD129", "renderD130", "renderD131"]
    return real_listdir(path)
  for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "C2D_NUM_GPUS"):
    monkeypatch.delenv(var, raising=False)
  monkeypatch.setattr(gpu_count.os, "listdir", fake_listdir)
  assert gpu_count.count_visible_gpus() == 4
  monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1,7,2")
  assert gpu_count.count_visible_gpus() == 2            # index 7 ends the list
  monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "GPU-abc,GPU-def,GPU-123")
  monkeypatch.setenv("HIP_VISIBLE_DEVICES", "2,0")
  assert gpu_count.count_visible_gpus() == 2
  monkeypatch.setenv("C2D_NUM_GPUS", "8")
  assert gpu_count.count_visible_gpus() == 8
