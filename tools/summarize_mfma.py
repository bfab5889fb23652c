"""MFMA-pipe utilisation per kernel family from a rocprofv3 PMC pass of bench.py
(profiles/rNN_mfma.json; bench.py quotes it as `roofline.mfma_busy`).

  python tools/summarize_mfma.py OUT.json PMC_DIR

PMC_DIR: output of ONE pass (own run, `--kernel-trace` only beside `--pmc`, the program directly
after `--`):
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE \\
            --kernel-trace --output-format csv -d PMC_DIR -- python3 bench.py ...
Formula (MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe busy cycles summed over
the SIMDs, 32 per v_mfma_f32_32x32x16_bf16, 64 per v_mfma_f32_32x32x2_f32; GRBM_GUI_ACTIVE is
summed over the 8 XCDs):
  mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)
i.e. the fraction of SIMD-cycles of the dispatch in which the matrix pipe was executing.  It is a
per-CLOCK figure of the MFMAs actually ISSUED: the pixel-major kernels skip the MFMAs of taps that
fall into the SAME padding (100 of 144 (pixel, tap) pairs are real on a 4x4 map), while
`roofline.achieved` counts the algorithmic FLOPs of the convolution (all nine taps), so
`mfma_busy` can sit BELOW `roofline.frac` — the difference is matrix-pipe time that was never
spent.  On the other side it is not diluted by the clock: the chip holds ~2.1 GHz under this load,
the nominal peaks assume 2.4.

Calibrated in round 4 (tools/calib/, profiles/r04_counter_calibration.json): pure
v_mfma_f32_32x32x2_f32 / v_mfma_f32_32x32x16_bf16 loops on every SIMD (1 and 4 waves per SIMD) read
0.992-0.995, the same loops on two of a CU's four SIMDs 0.489-0.497, and the counter itself equals
64 / 32 cycles x the MFMAs issued to 1e-4 — the reading IS the busy fraction (no factor; the 0.5 %
are the launch's ramp in GRBM_GUI_ACTIVE).  `per_kernel` lists every template instance."""
import collections
import re
import csv
import glob
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _atomic import write_json  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from summarize_pmc import family_of  # noqa: E402


def family(name):
  if "wgrad1x1_x9_kernel" in name or "wgrad3x3_x9_kernel" in name:
    return "wgrad_x9"
  if re.search(r"igemm_ring(?:_group)?_kernel<[^>]*, 4, (?:true|false), 3>", name):
    return "igemm_x9"         # fp32 operands as nine bf16 partial products (csrc/igemm_x9.hip)
  if "igemm_bf16_kernel" in name or re.search(r"igemm_ring(?:_group)?_kernel<[^>]*, 2(?:, (?:true|false))?(?:, 1)?>", name):
    return "igemm_bf16"
  if "wgrad_reduce_kernel" in name:
    return "wgrad_bf16"
  return family_of(name)


def main():
  out_path, pmc_dir = sys.argv[1:3]
  files = glob.glob(os.path.join(pmc_dir, "**", "*counter_collection.csv"), recursive=True)
  if not files:
    raise SystemExit("no counter_collection.csv under " + pmc_dir)
  agg = collections.defaultdict(lambda: collections.defaultdict(float))
  launches = collections.defaultdict(set)
  kagg = collections.defaultdict(lambda: collections.defaultdict(float))
  klaunches = collections.defaultdict(set)
  for path in files:
    with open(path) as f:
      for row in csv.DictReader(f):
        fam = family(row["Kernel_Name"])
        if not fam:
          continue
        km = re.search(r"(\w+_kernel(?:<[^>]*>)?)", row["Kernel_Name"])
        if km:
          kagg[km.group(1)][row["Counter_Name"]] += float(row["Counter_Value"])
          klaunches[km.group(1)].add(row.get("Dispatch_Id", row.get("Correlation_Id", "")))
        fams_of_row = [fam]
        if fam == "igemm":     # the big-tile kernels apart from the single-image first-stage ones
          fams_of_row.append("igemm_small_only" if "igemm_small" in row["Kernel_Name"]
                             else "igemm_nt_only")
        for fq in fams_of_row:
          agg[fq][row["Counter_Name"]] += float(row["Counter_Value"])
          launches[fq].add(row.get("Dispatch_Id", row.get("Correlation_Id", "")))
  # kernel durations INSIDE this PMC pass (its own --kernel-trace), when the trace is still there:
  # the implied clock GRBM_GUI_ACTIVE / 8 / duration says whether the pass ran the kernels at the
  # speed of the unprofiled run (MI355X_MICROARCH.md: profiled passes clock 3-6 % lower)
  kdur = collections.defaultdict(float)
  for path in glob.glob(os.path.join(pmc_dir, "**", "*kernel_trace.csv"), recursive=True):
    with open(path) as f:
      for row in csv.DictReader(f):
        km = re.search(r"(\w+_kernel(?:<[^>]*>)?)", row["Kernel_Name"])
        if km:
          kdur[km.group(1)] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
  fams = {}
  for fam, c in sorted(agg.items()):
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    fams[fam] = {
        "launches": len(launches[fam]),
        "SQ_VALU_MFMA_BUSY_CYCLES": busy,
        "SQ_BUSY_CU_CYCLES": c.get("SQ_BUSY_CU_CYCLES"),
        "GRBM_GUI_ACTIVE": gui,
        "mfma_busy": (busy / (gui / 8.0 * 256 * 4)) if gui else None,
    }
  per_kernel = {}
  for k, c in sorted(kagg.items()):
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    per_kernel[k] = {"launches": len(klaunches[k]),
                     "mfma_busy": (c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8.0 * 1024)) if gui else None,
                     "avg_us_in_this_pass": (kdur[k] / 1e3 / len(klaunches[k])) if k in kdur else None,
                     "implied_clock_GHz": (gui / 8.0 / kdur[k]) if k in kdur and kdur[k] else None}
  write_json(out_path,
             {"formula": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 * 4)",
              "calibration": "profiles/r04_counter_calibration.json: reads 0.992-0.995 at a true 1.0, "
                             "0.489-0.497 at a true 0.5 (no factor applied)",
              "families": fams, "per_kernel": per_kernel}, indent=1)
  for fam, v in fams.items():
    print("%-22s launches %5d  mfma_busy %s" % (fam, v["launches"], v["mfma_busy"]))


if __name__ == "__main__":
  main()
