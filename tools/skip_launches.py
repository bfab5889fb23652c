"""Upper bound of what removing a kernel family would buy: bench.py with the named hip_ops entry
points turned into no-ops (WRONG results, right timing of everything else).

  python tools/skip_launches.py c2 bn_relu_bwd_partial,bn_relu_bwd [bench args ...]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cap2det_amd import hip_ops  # noqa: E402

cfg, names = sys.argv[1], [n for n in sys.argv[2].split(",") if n]
for n in names:
  assert hasattr(hip_ops, n), n
  setattr(hip_ops, n, lambda *a, **k: None)
import bench  # noqa: E402
bench.main(["--config", cfg, "--no-cpu-baseline", "--no-kernel-timing"] + sys.argv[3:])
