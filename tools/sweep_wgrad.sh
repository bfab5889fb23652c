#!/bin/bash
# bf16 1x1 filter-gradient ring kernel: (stage rows, ring depth, grid slots) sweep (GPU box, repo root)
O=gpurun_out/wgrad_sweep; mkdir -p $O
python tools/bench_step_gemms.py bf16 wgrad > $O/default.log 2>&1
C2D_TUNE=wring=0 python tools/bench_step_gemms.py bf16 wgrad > $O/tn_kernel_only.log 2>&1
for cfg in ${WRING_CFGS:-"32 3 768" "32 3 512" "32 3 1024" "32 2 768" "32 4 512" "64 2 512" "64 3 256" "64 2 768"}; do
  set -- $cfg
  C2D_TUNE=wring_bk=$1,wring_d=$2,wring_slots=$3 python tools/bench_step_gemms.py bf16 wgrad > $O/bk$1_d$2_s$3.log 2>&1
done
for f in $O/*.log; do echo $f; tail -n 1 $f; done
