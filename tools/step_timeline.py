"""Kernel sequence of ONE steady-state training step from a rocprofv3 --kernel-trace CSV: per
queue (stream) the launches in order with their duration and the idle gap in front of them.
  python tools/step_timeline.py DIR_WITH_kernel_trace_csv [min_gap_us]"""
import csv, glob, os, re, sys
d = sys.argv[1]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
  rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
rows.sort()
marks = [i for i, r in enumerate(rows) if "midn_fwd_kernel" in r[2]]
lo, hi = marks[-3], marks[-2]
seg = rows[lo:hi]
t0 = seg[0][0]
last_end = {}
short = lambda n: re.sub(r"\(.*", "", n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", ""))[:58]
print("step span %.3f ms, %d launches" % ((seg[-1][1] - t0) / 1e6, len(seg)))
busy_end = t0
for s, e, n, q in seg:
  gap_q = (s - last_end[q]) / 1e3 if q in last_end else 0.0
  gap_all = max(0.0, (s - busy_end) / 1e3)
  if gap_all >= min_gap:
    print("%9.1f us  q%-3s %-58s %7.1f us   gap(queue) %6.1f  idle(GPU) %6.1f" % ((s - t0) / 1e3, q, short(n), (e - s) / 1e3, gap_q, gap_all))
  last_end[q] = e
  busy_end = max(busy_end, e)
