#!/bin/bash
# Where the bf16 step GEMMs' cycles go, from the SQ / LDS counters (four rocprofv3 --pmc passes +
# kernel trace over tools/bench_step_gemms.py bf16 fwd|dgrad).  GPU box, repo root.
# Output: gpurun_out/gemm_sq/summary.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/gemm_sq; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; export C2D_BENCH_ITERS=3
WHAT=${1:-fwd}
P1="SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES"
P2="SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC"
P3="SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT"
P4="SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_IFETCH"
i=0
for P in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $O/p$i -o c -- python3 $R/tools/bench_step_gemms.py bf16 $WHAT > $O/p$i.log 2>&1
done
python3 - <<PY | tee $O/summary_$WHAT.txt
import csv, glob, collections, re
cnt = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int); dur = collections.defaultdict(float); nd = collections.defaultdict(int)
for i in (1, 2, 3, 4):
    seen = set()
    for p in glob.glob("$O/p%d/**/*counter_collection.csv" % i, recursive=True):
        for r in csv.DictReader(open(p)):
            m = re.search(r"(igemm_ring_kernel<[^>]*>)", r["Kernel_Name"])
            if not m: continue
            k = m.group(1); cnt[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if i == 1 and (k, r["Dispatch_Id"]) not in seen: seen.add((k, r["Dispatch_Id"])); n[k] += 1
    if i == 1:
        for p in glob.glob("$O/p1/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(p)):
                m = re.search(r"(igemm_ring_kernel<[^>]*>)", r["Kernel_Name"])
                if m: dur[m.group(1)] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"]); nd[m.group(1)] += 1
for k in sorted(cnt):
    c = cnt[k]; L = max(n[k], 1)
    wc = max(c["SQ_WAVE_CYCLES"], 1); bc = max(c["SQ_BUSY_CU_CYCLES"], 1)
    print("%s  launches %d  avg %.1f us (under the counters)" % (k, L, dur[k] / max(nd[k], 1) / 1e3))
    print("   per wave-cycle: waiting on an instruction %.2f (on LDS %.2f), wait any %.2f, issuing %.2f" % (
        c["SQ_WAIT_INST_ANY"] / wc, c["SQ_WAIT_INST_LDS"] / wc, c["SQ_WAIT_ANY"] / wc, c["SQ_ACTIVE_INST_ANY"] / wc))
    print("   per busy-CU cycle (4 = every SIMD every cycle): MFMA busy %.2f  VALU %.2f  scalar %.2f  LDS %.2f  VMEM %.2f  misc %.2f | SALU inst cycles %.2f  VMEM-read inst cycles %.2f" % (
        c["SQ_VALU_MFMA_BUSY_CYCLES"] / bc, c["SQ_ACTIVE_INST_VALU"] / bc, c["SQ_ACTIVE_INST_SCA"] / bc, c["SQ_ACTIVE_INST_LDS"] / bc,
        c["SQ_ACTIVE_INST_VMEM"] / bc, c["SQ_ACTIVE_INST_MISC"] / bc, c["SQ_INST_CYCLES_SALU"] / bc, c["SQ_INST_CYCLES_VMEM_RD"] / bc))
    print("   LDS: array active %.2f of busy-CU cycles, bank-conflict cycles %.2f, cmd FIFO full %.3f, data FIFO full %.3f; TA addr FIFO full %.3f, TA cmd FIFO full %.3f" % (
        c["SQ_LDS_IDX_ACTIVE"] / bc, c["SQ_LDS_BANK_CONFLICT"] / bc, c["SQ_LDS_CMD_FIFO_FULL"] / bc, c["SQ_LDS_DATA_FIFO_FULL"] / bc,
        c["SQ_VMEM_TA_ADDR_FIFO_FULL"] / bc, c["SQ_VMEM_TA_CMD_FIFO_FULL"] / bc))
    m = max(c["SQ_INSTS_MFMA"], 1)
    print("   instructions per MFMA: SALU %.2f  VALU(other) %.2f  LDS %.2f  VMEM read %.2f  branch %.2f  SMEM %.2f  ifetch %.2f" % (
        c["SQ_INSTS_SALU"] / m, (c["SQ_INSTS_VALU"] - c["SQ_INSTS_MFMA"]) / m, c["SQ_INSTS_LDS"] / m, c["SQ_INSTS_VMEM_RD"] / m,
        c["SQ_INSTS_BRANCH"] / m, c["SQ_INSTS_SMEM"] / m, c["SQ_IFETCH"] / m))
    print("   raw:", {kk: int(v / L) for kk, v in sorted(c.items())})
PY
