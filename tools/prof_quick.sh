#!/bin/bash
# Quick kernel-stats profile of one bench config on one stream (GPU box, repo root):
#   bash tools/prof_quick.sh c2 tag   ->  gpurun_out/<tag>_<cfg>_serial_stats.csv (top kernels printed)
CFG=${1:-c2}; TAG=${2:-q}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export C2D_WGRAD_SIDE_STREAM=0
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_${CFG}_serial -o s -- python3 $R/bench.py --config $CFG --steps 5 --warmup 2 --no-cpu-baseline > $O/${TAG}_${CFG}_serial.log 2>&1
f=$(find $O/${TAG}_${CFG}_serial -name "*kernel_stats.csv" | head -1)
cp "$f" $O/${TAG}_${CFG}_serial_stats.csv
find $O/${TAG}_${CFG}_serial -name "*kernel_trace.csv" -delete
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/${TAG}_${CFG}_serial_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)/7e3
print("total kernel us/step: %.1f"%tot)
for r in rows[:45]:
    print("%-100s %5d %9.1f %5.1f%%"%(r["Name"][:100], int(r["Calls"])//7, float(r["TotalDurationNs"])/7e3, float(r["Percentage"])))
PY
