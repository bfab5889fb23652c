#!/bin/bash
# L2 -> CU traffic of the bf16 step GEMMs (one rocprofv3 --pmc pass + kernel trace over
# tools/bench_step_gemms.py bf16 dgrad|fwd).  GPU box, repo root.  Output: gpurun_out/gemm_l2/
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/gemm_l2; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; export C2D_BENCH_ITERS=3
for what in fwd dgrad; do
  timeout 300 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O/$what -o c -- python3 $R/tools/bench_step_gemms.py bf16 $what > $O/$what.log 2>&1
done
python3 - <<PY
import csv, glob, collections, re
for what in ("fwd", "dgrad"):
    cnt = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int); dur = collections.defaultdict(float)
    for p in glob.glob("$O/%s/**/*counter_collection.csv" % what, recursive=True):
        seen = set()
        for r in csv.DictReader(open(p)):
            m = re.search(r"(igemm_ring_kernel<[^>]*>)", r["Kernel_Name"])
            if not m: continue
            k = m.group(1); cnt[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if (k, r["Dispatch_Id"]) not in seen: seen.add((k, r["Dispatch_Id"])); n[k] += 1
    for p in glob.glob("$O/%s/**/*kernel_trace.csv" % what, recursive=True):
        for r in csv.DictReader(open(p)):
            m = re.search(r"(igemm_ring_kernel<[^>]*>)", r["Kernel_Name"])
            if m: dur[m.group(1)] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    print("==", what)
    for k in sorted(cnt):
        c = cnt[k]; req = c["TCP_TCC_READ_REQ_sum"]; t = dur[k] * 1e-9
        hit = c["TCC_HIT_sum"] / max(c["TCC_HIT_sum"] + c["TCC_MISS_sum"], 1)
        print("%-52s launches %3d  avg %6.1f us  L2->CU %6.2f TB/s @64B/req (%5.2f @128B)  L2 hit %.2f" % (k, n[k], dur[k] / n[k] / 1e3, req * 64 / t / 1e12, req * 128 / t / 1e12, hit))
PY
