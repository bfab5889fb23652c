"""Is the step host-bound?  bench.py with a busy-wait of D microseconds in front of every C-ABI call
(the host gets slower, the GPU work stays the same): a GPU-bound step does not move.

  python tools/host_delay.py c2 2.0 [bench args ...]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cap2det_amd import _lib  # noqa: E402

cfg, delay = sys.argv[1], float(sys.argv[2]) * 1e-6
inner = _lib.call


def slow_call(name, *args):
  if delay > 0:
    t = time.perf_counter() + delay
    while time.perf_counter() < t:
      pass
  return inner(name, *args)


_lib.call = slow_call
import bench  # noqa: E402
bench.main(["--config", cfg, "--no-cpu-baseline", "--no-kernel-timing"] + sys.argv[3:])
