#!/bin/bash
# f32x9 tile / ring sweep (needs a -DC2D_X9_SWEEP build of igemm_x9.o): every GEMM call of the step
# alone, per forced configuration.  Output: gpurun_out/$1/x9_<what>_<cfg>.txt
out=gpurun_out/${1:-x9sweep}
mkdir -p $out
for what in fwd dgrad; do
  C2D_TUNE=on=1 python tools/bench_step_gemms.py fp32 $what > $out/f32_$what.txt 2>&1
  for d in 2 3 4; do
    C2D_TUNE=x9_d=$d python tools/bench_step_gemms.py x9 $what > $out/x9_${what}_d$d.txt 2>&1
  done
  for nt in 3 4 22; do
    C2D_TUNE=x9_nt_pm=$nt python tools/bench_step_gemms.py x9 $what > $out/x9_${what}_pm$nt.txt 2>&1
  done
  for nt in 2 4 6 22 24; do
    C2D_TUNE=x9_nt=$nt python tools/bench_step_gemms.py x9 $what > $out/x9_${what}_rm$nt.txt 2>&1
  done
  C2D_TUNE=x9_bk=32,x9_d=2 python tools/bench_step_gemms.py x9 $what > $out/x9_${what}_bk32.txt 2>&1
done
