#!/bin/bash
# Two PMC passes over the ROI-crop forward alone (tools/bench_crop_fwd.py): instruction counts and
# wave-cycle breakdown of whichever form the library dispatches (C2D_TUNE=crop_stream=1|2).
# Separate runs per counter set, the program directly after `--` (GPU box, repo root).
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/crop_fwd_pmc${1:+_$1}; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
run() { name=$1; shift; timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/$name -o c -- python3 $R/tools/bench_crop_fwd.py > $O/$name.log 2>&1 || echo "pass $name failed" ; }
run sq_time   SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
run sq_insts  SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE
python3 - <<PY
import csv, glob, collections
for name in ("sq_time", "sq_insts"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv" % name, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "roi_crop" not in k: continue
            k = k.split("(")[0][-60:]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for k, d in acc.items():
            print(k)
            for c, v in sorted(d.items()): print("   %-28s %14.0f per dispatch" % (c, v / cnt[(k, c)]))
PY
