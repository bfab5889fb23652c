O=gpurun_out/r2n; mkdir -p $O
export C2D_TUNE=1
timeout 200 python -m pytest tests/test_gpu_bf16.py -x -q 2>&1 | tail -2
for pf in 0 1 2 3 5; do echo "== PF lead $pf"; C2D_BF16_PF=$pf timeout 120 python tools/bench_conv_bf16.py igemm 2>&1 | tee $O/pf$pf.log | tail -2; done
C2D_BF16_PF=2 timeout 200 python -m pytest tests/test_gpu_bf16.py -x -q 2>&1 | tail -2
