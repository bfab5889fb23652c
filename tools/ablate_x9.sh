#!/bin/bash
# f32x9 ablation: the step's GEMM calls with parts of the stage loop switched off
# (IgemmArgs::dbg: 1 the general loop with nothing off, 4 no MFMA, 32 no split arithmetic,
#  64 no DMA after the prologue, 8 no epilogue).
out=gpurun_out/${1:-x9abl}
mkdir -p $out
for dbg in 0 1 4 32 64 8 36 68 100 108; do
  C2D_TUNE=igemm_dbg=$dbg python tools/bench_step_gemms.py x9 fwd > $out/fwd_dbg$dbg.txt 2>&1
done
