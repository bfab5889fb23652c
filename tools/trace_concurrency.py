"""How much of a training step runs kernels CONCURRENTLY, from a rocprofv3 --kernel-trace CSV
(VERDICT r3 "next" #7: is the hipGraph replay's multi-branch graph actually executed in parallel?).

  python tools/trace_concurrency.py OUT.json LABEL=DIR [LABEL=DIR ...]

Per trace, over the steady-state steps between the first and the last `midn_fwd_kernel` launch
(one per step): the span per step, the time at least one kernel runs (busy), the time at least TWO
run (overlapped), the sum of the kernel durations, the number of distinct hardware queues the
kernels were dispatched from, and the overlapped time of the filter-gradient kernels (`wgrad*`,
the side-stream work) with anything else."""
import csv
import glob
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _atomic import write_json  # noqa: E402


def analyse(directory):
  f = glob.glob(os.path.join(directory, "**", "*kernel_trace.csv"), recursive=True)[0]
  rows = []
  for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                 r.get("Queue_Id", "0")))
  rows.sort()
  marks = [i for i, r in enumerate(rows) if "midn_fwd_kernel" in r[2]]
  lo, hi = marks[1], marks[-1]          # (skips the first step: warm-up / capture)
  seg = rows[lo:hi]
  steps = len(marks) - 2
  t0, t1 = seg[0][0], max(r[1] for r in seg)
  ev = []
  for s, e, n, q in seg:
    w = 1 if "wgrad" in n else 0
    ev.append((s, 1, w))
    ev.append((e, -1, -w))
  ev.sort()
  depth = wdepth = 0
  busy = over = wover = 0
  last = ev[0][0]
  for t, d, w in ev:
    dt = t - last
    if depth >= 1:
      busy += dt
    if depth >= 2:
      over += dt
    if wdepth >= 1 and depth > wdepth:
      wover += dt
    depth += d
    wdepth += w
    last = t
  total = sum(e - s for s, e, _, _ in seg)
  return {"steps": steps, "launches_per_step": len(seg) / float(steps),
          "span_ms_per_step": (t1 - t0) / 1e6 / steps, "busy_ms_per_step": busy / 1e6 / steps,
          "overlapped_ms_per_step": over / 1e6 / steps,
          "sum_of_kernel_durations_ms_per_step": total / 1e6 / steps,
          "filter_gradient_overlap_ms_per_step": wover / 1e6 / steps,
          "queues": len(set(q for _, _, _, q in seg))}


def main():
  out = {}
  for arg in sys.argv[2:]:
    label, d = arg.split("=", 1)
    out[label] = analyse(d)
    print(label, json.dumps(out[label]))
  write_json(sys.argv[1], out, indent=1, sort_keys=True)


if __name__ == "__main__":
  main()
