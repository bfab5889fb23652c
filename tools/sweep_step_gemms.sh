#!/bin/bash
# bf16 ring-kernel sweep over (block tile, stage depth, ring depth) on the step's GEMM calls
# (GPU box, repo root).  Output: gpurun_out/step_gemms/<tile>_<bk>_<d>.log
O=gpurun_out/step_gemms; mkdir -p $O
python tools/bench_step_gemms.py bf16 > $O/default.log 2>&1
for cfg in ${TILE_CFGS:-0 2 3 6 7}; do
  IFS=";" read -ra RCS <<< "${RING_CFGS:-64 2;32 3;64 3}"
  for rc in "${RCS[@]}"; do
    set -- $rc
    C2D_TUNE=igemm_cfg=$cfg,ring_bk=$1,ring_d=$2 python tools/bench_step_gemms.py bf16 > $O/t${cfg}_bk$1_d$2.log 2>&1
  done
done
for f in $O/*.log; do echo $f; tail -n 2 $f; done
