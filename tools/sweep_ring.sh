#!/bin/bash
# Sweep of the bf16 ring kernel's (stage depth, ring depth) on the second-stage layer shapes
# (run on the GPU box from the repo root; writes gpurun_out/ring_sweep/*.log).
O=gpurun_out/ring_sweep; mkdir -p $O
python tools/bench_conv_bf16.py igemm > $O/default.log 2>&1
for cfg in ${RING_CFGS:-"64 2" "64 3" "32 2" "32 3" "32 4"}; do
  set -- $cfg
  C2D_TUNE=ring_bk=$1,ring_d=$2 python tools/bench_conv_bf16.py igemm > $O/bk$1_d$2.log 2>&1
done
for f in $O/*.log; do echo $f; tail -n 2 $f; done
