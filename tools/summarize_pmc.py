"""Turns rocprofv3 PMC passes of `bench.py` into the per-kernel-family HBM traffic summary that
bench.py reads back as `roofline.traffic` (profiles/rNN_traffic.json).

  python tools/summarize_pmc.py OUT.json FETCH_DIR WRITE_DIR [STATS_DIR]

FETCH_DIR / WRITE_DIR are the output directories of two SEPARATE passes
(`rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d DIR -- python3 bench.py ...`
and the same with WRITE_SIZE: the two counters do not fit one pass on gfx950).  Units and
corrections follow MI355X_MICROARCH.md §HBM: both counters are in KiB; on gfx950 FETCH_SIZE
tallies 128-B requests at 64 B, so it is DOUBLED; WRITE_SIZE is exact for 16-B/lane stores and
float atomics.  The factors are CALIBRATED per access shape by tools/calib/ (round 4,
profiles/r04_counter_calibration.json: kernels that read / write 1 GiB once): FETCH_SIZE reads
exactly 1/2 for global loads of 4, 8 and 16 B per lane in contiguous runs of 64 B ... 1 KiB and
for `buffer_load ... lds` of 4 / 16 B per lane (TCC_EA0_RDREQ counts one request per 128 B, the
FETCH_SIZE expression prices it at 64 B); the one exception measured is LDS-DMA of 64-byte runs
(x1.84: 9 % of its requests are re-fetches).  WRITE_SIZE reads 1.000 for 4-, 8-, 16-B stores and
float atomics.  `FETCH_FACTORS` below maps kernel families to their access shape's factor.
Steps are counted by the launches of `midn_fwd_kernel` (exactly one per training step)."""
import collections
import csv
import glob
import json
import os
import re
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _atomic import write_json  # noqa: E402

FAMILIES = [   # (family key used by bench.py, substring of the kernel name)
    ("igemm", "igemm_nt_kernel"),
    ("igemm", "igemm_small_kernel"),
    ("igemm", "igemm_ring_kernel"),      # (fp32 instances: the bf16 ones are matched first)
    ("igemm", "igemm_ring_group_kernel"),
    ("wgrad", "wgrad_tn_kernel"),
    ("wgrad", "wgrad3x3_kernel"),
    ("roi_crop_pool_fwd", "roi_crop_pool_fwd_kernel"),
    ("roi_crop_pool_fwd", "roi_crop_pool2_fwd_stream_kernel"),
    ("roi_crop_pool_fwd", "roi_crop_pool2_fwd_rowwalk_kernel"),
    ("roi_crop_pool_bwd", "roi_bwd_"),
    ("roi_crop_pool_bwd", "roi_bin_rows_kernel"),
    ("roi_crop_pool_bwd", "roi_axes_kernel"),
    ("bn_relu_bwd", "bn_relu_bwd_kernel"),
    ("pool3x3", "pool3x3_"),
    ("adagrad", "adagrad_kernel"),
    ("adagrad", "adagrad_multi_kernel"),
]


# FETCH_SIZE correction per kernel family (access shape -> profiles/r04_counter_calibration.json).
# Everything the step launches reads in one of the calibrated x2.000 shapes; the bf16 ring kernels
# stage 64-byte rows (32-deep stages) in some launches and 128-byte rows in others: 2.0 for the
# 128-byte launches, 1.84 for the 64-byte ones — the family average is taken as 1.92 and the spread
# (+-4 %) is the stated uncertainty of that family's traffic.
FETCH_FACTORS = {"igemm_bf16": 1.92}
DEFAULT_FETCH_FACTOR = 2.0


def family_of(name):
  # bf16-operand kernels (bench.py --dtype bf16) are separate families: their roofline is the
  # bf16 MFMA peak
  if "wgrad1x1_x9_kernel" in name or "wgrad3x3_x9_kernel" in name:
    return "wgrad_x9"         # 1x1 filter gradients as nine bf16 partial products (csrc/igemm_x9.hip)
  if ("wgrad_tn_bf16_kernel" in name or "wgrad3x3_bf16" in name or "wgrad1x1_bf16_ring" in name or
      "wgrad_reduce_kernel" in name):
    return "wgrad_bf16"
  if re.search(r"igemm_ring(?:_group)?_kernel<[^>]*, 4, (?:true|false), 3>", name):
    return "igemm_x9"         # fp32 operands as nine bf16 partial products (csrc/igemm_x9.hip)
  if "igemm_bf16_kernel" in name or re.search(r"igemm_ring(?:_group)?_kernel<[^>]*, 2(?:, (?:true|false))?(?:, 1)?>", name):
    return "igemm_bf16"
  if re.search(r"igemm_small(_group)?_kernel<\d, 2>", name):
    return "igemm_small_bf16"      # the bf16 step's single-image first stage
  m = re.search(r"igemm_nt_kernel<([^>]*)>", name)
  if m and len(m.group(1).split(",")) == 8 and m.group(1).split(",")[-1].strip() == "2":
    return "igemm_bf16"       # igemm_nt_kernel<MODE, WM, WN, MT, NT, BKT, PM, ES = 2>
  for fam, sub in FAMILIES:
    if sub in name:
      return fam
  if "wgrad" in name:       # any other filter-gradient kernel (wgrad3x3_s2_kernel, wgrad_tn_group_kernel, ...)
    return "wgrad_bf16" if "bf16" in name else "wgrad"
  if "igemm" in name:       # (igemm_small_group_kernel<*, 4>: the fp32 first stage's grouped launches)
    return "igemm"
  return None


def instance_of(name):
  """Kernel name reduced to its template instance (no argument list, no namespaces)."""
  n = re.sub(r"\(anonymous namespace\)::|c2d_ig::|^void ", "", name)
  m = re.match(r"([A-Za-z_0-9]+(?:<[^>]*>)?)", n)
  return m.group(1) if m else n[:80]


PER_KERNEL = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))   # counter -> instance -> [KiB, launches]


def read_counter(directory, counter):
  """-> ({family: [sum_KiB, launches]}, steps)"""
  out = collections.defaultdict(lambda: [0.0, 0])
  steps = 0
  files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
  if not files:
    raise SystemExit("no counter_collection.csv under " + directory)
  for path in files:
    with open(path) as f:
      for row in csv.DictReader(f):
        if row["Counter_Name"] != counter:
          continue
        name = row["Kernel_Name"]
        if "midn_fwd_kernel" in name:
          steps += 1
        fam = family_of(name)
        if fam:
          out[fam][0] += float(row["Counter_Value"])
          out[fam][1] += 1
          rec = PER_KERNEL[counter][instance_of(name)]
          rec[0] += float(row["Counter_Value"])
          rec[1] += 1
  return out, steps


def main():
  out_path, fetch_dir, write_dir = sys.argv[1:4]
  fetch, fsteps = read_counter(fetch_dir, "FETCH_SIZE")
  write, wsteps = read_counter(write_dir, "WRITE_SIZE")
  fams = {}
  for fam in sorted(set(fetch) | set(write)):
    fk, fl = fetch.get(fam, [0.0, 0])
    wk, wl = write.get(fam, [0.0, 0])
    launches = fl / float(fsteps) if fsteps else 0.0
    ff = FETCH_FACTORS.get(fam, DEFAULT_FETCH_FACTOR)   # gfx950: FETCH_SIZE reads 1/2 (calibrated)
    rd = ff * fk * 1024.0 / max(fsteps, 1)
    wr = wk * 1024.0 / max(wsteps, 1)
    fams[fam] = {
        "launches_per_step": launches,
        "hbm_read_bytes_per_step": rd,
        "hbm_write_bytes_per_step": wr,
        "hbm_bytes_per_step": rd + wr,
        "hbm_bytes_per_launch": (rd + wr) / launches if launches else None,
        "fetch_factor": ff,
        "raw_FETCH_SIZE_KiB_per_step": fk / max(fsteps, 1),
        "raw_WRITE_SIZE_KiB_per_step": wk / max(wsteps, 1),
    }
  doc = {
      "source": {"fetch_pass": fetch_dir, "write_pass": write_dir,
                 "steps_in_fetch_pass": fsteps, "steps_in_write_pass": wsteps},
      "corrections": "KiB -> bytes (x1024); FETCH_SIZE x fetch_factor (2.0 on gfx950 for every "
                     "calibrated access shape, 1.92 for the bf16 ring kernels' mix of 64- and "
                     "128-byte LDS-DMA rows: profiles/r04_counter_calibration.json); WRITE_SIZE as "
                     "read (calibrated 1.000)",
      "families": fams,
      # per template instance (what a launch of that instance moves, averaged over its launches):
      # the table behind "which kernel wastes traffic" — algorithmic bytes per launch are a property
      # of the call (bench.py --per-call), not of the instance
      "per_kernel": {
          inst: {"family": family_of(inst) or family_of(inst + "("),
                 "launches_per_step": PER_KERNEL["FETCH_SIZE"][inst][1] / float(max(fsteps, 1)),
                 "hbm_read_bytes_per_launch": (FETCH_FACTORS.get(family_of(inst), DEFAULT_FETCH_FACTOR) *
                                               PER_KERNEL["FETCH_SIZE"][inst][0] * 1024.0 /
                                               max(PER_KERNEL["FETCH_SIZE"][inst][1], 1)),
                 "hbm_write_bytes_per_launch": (PER_KERNEL["WRITE_SIZE"][inst][0] * 1024.0 /
                                                max(PER_KERNEL["WRITE_SIZE"][inst][1], 1))}
          for inst in sorted(set(PER_KERNEL["FETCH_SIZE"]) | set(PER_KERNEL["WRITE_SIZE"]))},
  }
  write_json(out_path, doc, indent=1, sort_keys=True)
  for fam, v in fams.items():
    print("%-20s %6.1f launches/step  read %9.1f MB  write %9.1f MB per step"
          % (fam, v["launches_per_step"], v["hbm_read_bytes_per_step"] / 1e6,
             v["hbm_write_bytes_per_step"] / 1e6))


if __name__ == "__main__":
  main()
