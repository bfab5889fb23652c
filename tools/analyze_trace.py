"""Idle time and per-stream overlap of the steady-state training steps from a rocprofv3
--kernel-trace CSV (run on the GPU box; prints a short summary).
  python tools/analyze_trace.py DIR_WITH_kernel_trace_csv [last_k_steps]"""
import csv, glob, os, sys
d = sys.argv[1]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 3
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
  rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", r.get("Stream_Id", "0"))))
rows.sort()
# steps are delimited by midn_fwd_kernel launches
marks = [i for i, r in enumerate(rows) if "midn_fwd_kernel" in r[2]]
if len(marks) < k + 2:
  raise SystemExit("not enough steps")
lo, hi = marks[-k - 1], marks[-1]
seg = rows[lo:hi]
t0, t1 = seg[0][0], seg[-1][0]
# union of busy intervals
busy, cur_s, cur_e = 0, None, None
conc2 = 0
events = []
for s, e, n, q in seg:
  events.append((s, 1)); events.append((min(e, t1), -1))
events.sort()
depth, last = 0, t0
idle = 0
for t, dlt in events:
  if t > last:
    if depth == 0: idle += t - last
    if depth >= 2: conc2 += t - last
    last = t
  depth += dlt
span = t1 - t0
print("steps %d  span %.3f ms/step  idle %.3f ms/step (%.1f %%)  >=2 kernels concurrently %.3f ms/step"
      % (k, span / k / 1e6, idle / k / 1e6, 100.0 * idle / span, conc2 / k / 1e6))
# biggest idle gaps: which kernel follows
gaps = []
evs = sorted(seg)
end_max = evs[0][1]
for s, e, n, q in evs[1:]:
  if s > end_max: gaps.append((s - end_max, n))
  end_max = max(end_max, e)
agg = {}
for g, n in gaps:
  key = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:60]
  a = agg.setdefault(key, [0, 0]); a[0] += g; a[1] += 1
for key, (g, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
  print("  idle before %-60s %8.1f us/step over %5.1f gaps/step" % (key, g / k / 1e3, c / k))
