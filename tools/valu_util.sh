#!/bin/bash
# Vector-ALU utilisation of every kernel of a benchmark step: one PMC pass (SQ_INSTS_VALU,
# SQ_WAVES, SQ_INSTS_VMEM_RD/WR, SQ_INSTS_SALU, SQ_INSTS_LDS) + the kernel trace of the same run.
# A wave64 vector instruction occupies its 16-lane SIMD for 4 cycles (8 for the 64-bit / MFMA
# forms, not separated here): util = INSTS_VALU x 4 / (duration x clock x 1024 SIMDs).
#   bash tools/valu_util.sh c2        (GPU box, repo root; program directly after `--`)
CFG=${1:-c1}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/valu_util_$CFG; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
C2D_TUNE=streams=0 timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/pmc -o c -- python3 $R/bench.py --config $CFG --steps 4 --warmup 2 --no-cpu-baseline --no-kernel-timing > $O/pmc.log 2>&1 || echo "pmc pass failed"
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter(); dur = collections.defaultdict(float)
for f in glob.glob("$O/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES":
            n[k] += 1; dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
rows = []
for k, d in acc.items():
    if not n[k]: continue
    us = dur[k] / n[k] / 1e3
    valu = d["SQ_INSTS_VALU"] / n[k]
    util = valu * 4 / (us * 1e-6 * 2.1e9 * 1024) if us > 0 else 0
    rows.append((dur[k], k, n[k], us, valu, util, d["SQ_WAVES"] / n[k], d["SQ_INSTS_VMEM_RD"] / n[k], d["SQ_INSTS_VMEM_WR"] / n[k], d["SQ_INSTS_SALU"] / n[k]))
rows.sort(reverse=True)
print("%-70s %5s %8s %10s %6s %8s %9s %9s %10s" % ("kernel", "calls", "us", "VALU", "util", "waves", "vmem_rd", "vmem_wr", "SALU"))
for _, k, c, us, valu, util, w, rd, wr, sa in rows[:45]:
    print("%-70s %5d %8.1f %10.0f %6.2f %8.0f %9.0f %9.0f %10.0f" % (k.replace("(anonymous namespace)::", "")[:70], c, us, valu, util, w, rd, wr, sa))
PY
