"""Reduces the passes of tools/calib/run_calibration.sh into profiles/rNN_counter_calibration.json:
per known-answer kernel the expected quantity (bytes read / written once from a 1 GiB buffer, or
matrix-pipe cycles = MFMAs x 64 (fp32 32x32x2) / 32 (bf16 32x32x16)), what the counter read, and
the factor a summary has to apply (expected / reading).

  python tools/calib/summarize_calibration.py OUT.json PLAIN.json FETCH_DIR WRITE_DIR MFMA_DIR [STATS_DIR [TCC_DIR]]

FETCH_SIZE / WRITE_SIZE are in KiB (x1024).  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES /
(GRBM_GUI_ACTIVE / 8 * 1024): GRBM_GUI_ACTIVE is summed over the 8 XCDs, 1024 = 256 CUs x 4 SIMDs."""
import collections
import csv
import glob
import json
import os
import re
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from _atomic import write_json  # noqa: E402


def rows(directory):
  files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
  if not files:
    raise SystemExit("no counter_collection.csv under " + directory)
  for path in files:
    with open(path) as f:
      for row in csv.DictReader(f):
        yield row


def short(name):
  m = re.search(r"(calib_\w+(?:<[^>]*>)?)", name)
  return m.group(1) if m else None


def per_kernel(directory):
  """-> {kernel: [ {counter: value} per dispatch, in dispatch order ]}"""
  disp = collections.OrderedDict()
  for row in rows(directory):
    k = short(row["Kernel_Name"])
    if k is None:
      continue
    key = (k, int(row.get("Dispatch_Id", 0)))
    disp.setdefault(key, {})[row["Counter_Name"]] = float(row["Counter_Value"])
  out = collections.OrderedDict()
  for (k, d), c in sorted(disp.items(), key=lambda kv: kv[0][1]):
    out.setdefault(k, []).append(c)
  return out


def main():
  out_path, plain_path, fetch_dir, write_dir, mfma_dir = sys.argv[1:6]
  stats_dir = sys.argv[6] if len(sys.argv) > 6 else None
  tcc_dir = sys.argv[7] if len(sys.argv) > 7 else None
  with open(plain_path) as f:
    plain = json.loads([l for l in f.read().splitlines() if l.startswith("{")][-1])
  fetch, write, mfma = per_kernel(fetch_dir), per_kernel(write_dir), per_kernel(mfma_dir)
  try:
    tcc = per_kernel(tcc_dir) if tcc_dir and os.path.isdir(tcc_dir) else {}
  except SystemExit:
    tcc = {}                     # (optional pass: counter names differ between rocprofv3 builds)
  dur = {}
  if stats_dir:
    for path in glob.glob(os.path.join(stats_dir, "**", "*kernel_stats.csv"), recursive=True):
      with open(path) as f:
        for row in csv.DictReader(f):
          k = short(row["Name"])
          if k:
            dur[k] = float(row["AverageNs"]) * 1e-3
  doc = {"unprofiled_run": plain, "memory": {}, "mfma": {}}
  for name, info in plain.items():
    base = name.split("@")[0]
    if info["kind"] == "mfma_cycles":
      # several launch shapes share one kernel name: dispatch order = program order; the program
      # launches every shape twice (warm-up + measured)
      shapes = [n for n in plain if n.split("@")[0] == base and plain[n]["kind"] == "mfma_cycles"]
      idx = shapes.index(name)
      d = mfma.get(base, [])
      if len(d) < 2 * len(shapes):
        continue
      c = d[2 * idx + 1]
      gui = c.get("GRBM_GUI_ACTIVE", 0.0)
      busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
      denom = gui / 8.0 * 1024.0
      doc["mfma"][name] = {
          "expected_mfma_cycles": info["expected"], "simd_share_issuing": info.get("simd_share"),
          "threads_per_block": info.get("threads"), "TFLOPs_unprofiled": info.get("TFLOPs"),
          "SQ_VALU_MFMA_BUSY_CYCLES": busy, "GRBM_GUI_ACTIVE": gui,
          "SQ_BUSY_CU_CYCLES": c.get("SQ_BUSY_CU_CYCLES"),
          "counter_over_expected_cycles": busy / info["expected"] if info["expected"] else None,
          "mfma_busy_reading": busy / denom if denom else None,
          "true_busy": info.get("simd_share"),
          "factor": (info.get("simd_share") / (busy / denom)) if busy and denom else None,
      }
      continue
    f = fetch.get(base, [{}])[-1].get("FETCH_SIZE")
    w = write.get(base, [{}])[-1].get("WRITE_SIZE")
    e = info["expected"]
    rec = {"kind": info["kind"], "expected_bytes": e, "ms_unprofiled": info["ms"],
           "TBps_unprofiled": info.get("TBps"), "avg_us_profiled": dur.get(base),
           "FETCH_SIZE_bytes": None if f is None else f * 1024.0,
           "WRITE_SIZE_bytes": None if w is None else w * 1024.0}
    if f is not None and "read" in info["kind"]:
      rec["fetch_factor"] = e / (f * 1024.0) if f else None
    if w is not None and "write" in info["kind"]:
      rec["write_factor"] = e / (w * 1024.0) if w else None
    if base in tcc:
      rec["tcc"] = tcc[base][-1]
    doc["memory"][name] = rec
  write_json(out_path, doc, indent=1, sort_keys=True)
  for name, r in doc["mfma"].items():
    print("%-28s reading %.4f  true %.2f  counter/expected cycles %.4f" %
          (name, r["mfma_busy_reading"] or 0, r["true_busy"], r["counter_over_expected_cycles"] or 0))
  for name, r in doc["memory"].items():
    print("%-26s %-10s fetch x%s  write x%s  (%s TB/s)" % (
        name, r["kind"], "%.3f" % r["fetch_factor"] if r.get("fetch_factor") else "-",
        "%.3f" % r["write_factor"] if r.get("write_factor") else "-", r.get("TBps_unprofiled")))


if __name__ == "__main__":
  main()
