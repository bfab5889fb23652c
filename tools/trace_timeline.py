"""Where a multi-stream training step's time goes, from a rocprofv3 --kernel-trace CSV.

  python tools/trace_timeline.py TRACE_DIR [SERIAL_STATS.csv] [--dump OUT.txt]

Steps are cut at the `midn_fwd_kernel` launches (one per step).  For the MEDIAN-length steady-state
step it prints: the span, the time no kernel at all runs (idle: launch latency, event hand-overs,
a host that is behind), per hardware queue the busy time and launch count, and — against the
per-kernel average of a one-stream profile of the same build (profiles/*_kernel_stats_*_serial.csv)
— how much longer every kernel family runs when it shares the chip (sum of durations in the step /
sum of the one-stream averages of the same launches).  --dump writes the step's launches in start
order (offset, queue, duration, one-stream average, name)."""
import csv
import glob
import os
import re
import sys


def family(n):
  low = n.lower()
  for key in ("wgrad", "igemm_ring", "igemm_small", "igemm_nt", "bn_relu_bwd", "roi_", "pool", "midn",
              "rmsprop", "adadelta", "cast_", "nccl"):
    if key in low:
      return key
  return "other"


def short(n):
  n = re.sub(r"\(anonymous namespace\)::", "", n)
  n = re.sub(r"^void ", "", n)
  return n.split("(")[0][:70]


def main():
  args = [a for a in sys.argv[1:] if not a.startswith("--")]
  dump = None
  if "--dump" in sys.argv:
    dump = sys.argv[sys.argv.index("--dump") + 1]
    args = [a for a in args if a != dump]
  f = glob.glob(os.path.join(args[0], "**", "*kernel_trace.csv"), recursive=True)[0]
  serial = {}
  if len(args) > 1:
    for r in csv.DictReader(open(args[1])):
      serial[r["Name"]] = float(r["AverageNs"])
  rows = []
  for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
  rows.sort()
  marks = [i for i, r in enumerate(rows) if "midn_fwd_kernel" in r[2]]
  steps = []
  for a, b in zip(marks[1:-1], marks[2:]):
    steps.append((rows[b][0] - rows[a][0], a, b))
  steps.sort()
  # (the first timed step of bench.py computes its own first stage — no look-ahead was issued
  #  for it: take the median among the steps with the most common launch count per queue)
  def signature(st):
    per = {}
    for r in rows[st[1]:st[2]]:
      per[r[3]] = per.get(r[3], 0) + 1
    return tuple(sorted(per.items()))
  counts = {}
  for st in steps:
    counts[signature(st)] = counts.get(signature(st), 0) + 1
  modal = max(counts, key=counts.get)
  usual = [st for st in steps if signature(st) == modal]
  span, a, b = usual[len(usual) // 2]
  seg = rows[a:b]
  t0 = seg[0][0]
  print("steps %d, spans ms: min %.3f median %.3f max %.3f" % (len(steps), steps[0][0] / 1e6, span / 1e6, steps[-1][0] / 1e6))
  ev = []
  for s, e, n, q in seg:
    ev.append((s, 1))
    ev.append((min(e, t0 + span), -1))
  ev.sort()
  depth, idle, last = 0, 0, t0
  gaps = []
  for t, d in ev:
    if depth == 0 and t > last:
      idle += t - last
      gaps.append((t - last, last - t0))
    depth += d
    last = t
  print("median step: launches %d, idle (no kernel running) %.3f ms in %d gaps; largest gaps (us @ offset us): %s" % (
      len(seg), idle / 1e6, len(gaps), ", ".join("%.0f@%.0f" % (g / 1e3, o / 1e3) for g, o in sorted(gaps)[-8:][::-1])))
  per_q = {}
  for s, e, n, q in seg:
    d = per_q.setdefault(q, [0, 0])
    d[0] += e - s
    d[1] += 1
  for q, (busy, cnt) in sorted(per_q.items(), key=lambda kv: -kv[1][0]):
    print("  queue %s: busy %.3f ms, %d launches" % (q, busy / 1e6, cnt))
  fam = {}
  for s, e, n, q in seg:
    d = fam.setdefault(family(n), [0.0, 0.0, 0])
    d[0] += e - s
    d[1] += serial.get(n, 0.0)
    d[2] += 1
  print("  family: in-step us / one-stream us (ratio), launches")
  for k, (dur, ser, cnt) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
    print("   %-12s %8.0f / %8.0f  (%.2f)  %d" % (k, dur / 1e3, ser / 1e3, dur / ser if ser else 0.0, cnt))
  if dump:
    qs = {q: i for i, q in enumerate(sorted(per_q, key=lambda q: -per_q[q][0]))}
    with open(dump, "w") as out:
      for s, e, n, q in seg:
        out.write("%9.1f q%d %8.1f %8.1f  %s\n" % ((s - t0) / 1e3, qs[q], (e - s) / 1e3, serial.get(n, 0.0) / 1e3, short(n)))


if __name__ == "__main__":
  main()
