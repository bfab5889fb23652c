"""Per-kernel averages of the PMC passes of tools/crop_counters.sh:
  python tools/summarize_crop_counters.py PMC_ROOT OUT.json
One entry per ROI-crop kernel: mean counter value per dispatch (summed over the counter's
instances as rocprofv3 reports them), the mean dispatch duration from the kernel trace, and the
ratios the round-3 analysis quotes (profiles/r03_crop_counters.json)."""
import collections
import csv
import glob
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _atomic import write_json  # noqa: E402

KERNELS = {"roi_crop_pool2_fwd_stream_kernel": "fwd_stream", "roi_crop_pool2_fwd_rowwalk_kernel": "fwd_rowwalk", "roi_bwd_strip_kernel": "bwd_strip",
           "roi_bin_rows_kernel": "bwd_bin_rows", "roi_bwd_sum_parts_kernel": "bwd_sum_parts",
           "roi_crop_pool_bwd_lds_kernel": "bwd_atomic_lds"}


def kernel_of(name):
  for k, v in KERNELS.items():
    if k in name:
      return v + ("_bf16" if ("DF16b" in name or "bf16" in name) else "")
  return None


def main():
  root, out = sys.argv[1:3]
  res = collections.defaultdict(lambda: collections.defaultdict(list))
  for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
    per = collections.defaultdict(float)
    with open(path) as f:
      for row in csv.DictReader(f):
        k = kernel_of(row["Kernel_Name"])
        if k:
          per[(k, row.get("Dispatch_Id", ""), row["Counter_Name"])] += float(row["Counter_Value"])
    for (k, _, c), v in per.items():
      res[k][c].append(v)
  for path in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
    with open(path) as f:
      for row in csv.DictReader(f):
        k = kernel_of(row["Kernel_Name"])
        if k:
          res[k]["duration_ns"].append(float(row["End_Timestamp"]) - float(row["Start_Timestamp"]))
  summary = {}
  for k, cs in sorted(res.items()):
    e = {c: sum(v) / len(v) for c, v in sorted(cs.items())}
    e["dispatches_seen"] = max(len(v) for v in cs.values())
    if "FETCH_SIZE" in e:   # KB, halved on gfx950 for wide streaming reads (MI355X_MICROARCH.md §HBM)
      e["hbm_read_bytes_corrected"] = e["FETCH_SIZE"] * 1024.0 * 2.0
    if "WRITE_SIZE" in e:
      e["hbm_write_bytes"] = e["WRITE_SIZE"] * 1024.0
    if e.get("TCC_HIT_sum") is not None and e.get("TCC_MISS_sum") is not None:
      tot = e["TCC_HIT_sum"] + e["TCC_MISS_sum"]
      e["l2_hit_rate"] = e["TCC_HIT_sum"] / tot if tot else None
    if e.get("SQ_WAVE_CYCLES"):
      for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM",
                "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA"):
        if c in e:
          e[c + "_per_wave_cycle"] = e[c] / e["SQ_WAVE_CYCLES"]
    summary[k] = e
  write_json(out, {"source": "tools/crop_counters.sh (rocprofv3 --pmc, separate passes)", "kernels": summary},
             indent=1, sort_keys=True)


if __name__ == "__main__":
  main()
