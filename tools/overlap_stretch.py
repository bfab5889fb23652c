"""What the stream overlap costs, kernel by kernel: the committed rocprofv3 kernel statistics of the
default schedule (filter gradients on the side stream, look-ahead prefix on a third) against the
same step on ONE stream (`C2D_TUNE=streams=0`).  Prints the per-step sums, the filter-gradient
share and the kernels that stretch most when they share the chip.

  python tools/overlap_stretch.py [c1|c2] [tag]      (reads profiles/<tag>_bench_kernel_stats_*.csv)"""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
tag = sys.argv[2] if len(sys.argv) > 2 else "r03"
STEPS = 7.0           # tools/profile_round.sh: 2 warm-up + 5 timed steps


def load(name):
  with open(os.path.join(ROOT, "profiles", name)) as f:
    return {r["Name"]: (int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(f)}


ovl = load("%s_bench_kernel_stats_%s.csv" % (tag, cfg))
ser = load("%s_bench_kernel_stats_%s_serial.csv" % (tag, cfg))
tot_o = sum(t for _, t in ovl.values()) / STEPS / 1e6
tot_s = sum(t for _, t in ser.values()) / STEPS / 1e6
wg_o = sum(t for k, (_, t) in ovl.items() if "wgrad" in k) / STEPS / 1e6
wg_s = sum(t for k, (_, t) in ser.items() if "wgrad" in k) / STEPS / 1e6
print("%s: kernel time per step %.3f ms in the default schedule, %.3f ms on one stream" % (cfg, tot_o, tot_s))
print("    filter gradients %.3f ms (one stream: %.3f); everything else %.3f ms (%.3f)"
      % (wg_o, wg_s, tot_o - wg_o, tot_s - wg_s))
rows = []
for k, (c, t) in ovl.items():
  if k in ser and ser[k][0] > 0:
    rows.append(((t - ser[k][1] * c / ser[k][0]) / STEPS / 1e3, k, c / STEPS, t / c / 1e3, ser[k][1] / ser[k][0] / 1e3))
rows.sort(reverse=True)
print("%-72s %6s %9s %9s %10s" % ("kernel", "/step", "ovl us", "alone us", "+us/step"))
for d, k, c, a, b in rows[:14]:
  print("%-72s %6.1f %9.1f %9.1f %10.1f" % (k.replace("(anonymous namespace)::", "")[:72], c, a, b, d))
