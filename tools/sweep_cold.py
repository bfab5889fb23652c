"""Per-call best (block tile, stage depth, ring depth) of the bf16 ring kernel with COLD operands
(C2D_BENCH_COLD=1: as inside a training step), from the logs tools/sweep_step_gemms.sh writes.

  C2D_BENCH_COLD=1 RING_CFGS="64 2;32 3" bash tools/sweep_step_gemms.sh; python tools/sweep_cold.py"""
import glob
import os
import re
import sys

logs = sorted(glob.glob("gpurun_out/step_gemms/*.log"))
table = {}
for path in logs:
  name = os.path.basename(path)[:-4]
  for line in open(path):
    m = re.match(r"(fwd|dgrad)\s+(.*?)\s+([\d.]+) us\s+([\d.]+) TF\s+(\S.*)$", line)
    if m:
      table.setdefault((m.group(1), m.group(2)), {})[name] = (float(m.group(3)), m.group(5).split(";")[0])
tot_def = tot_best = 0.0
for key, row in table.items():
  d = row.get("default")
  best = min(row.items(), key=lambda kv: kv[1][0])
  tot_def += d[0]; tot_best += best[1][0]
  flag = "" if best[1][0] > 0.97 * d[0] else "   <== %s %s" % (best[0], best[1][1])
  print("%-5s %-40s default %6.1f us (%s)  best %6.1f%s" % (key[0], key[1][:40], d[0], d[1][11:], best[1][0], flag))
print("sum default %.1f us, per-call best %.1f us" % (tot_def, tot_best))
