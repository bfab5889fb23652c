O=gpurun_out/r2e; mkdir -p $O
python -m pytest tests/test_gpu_bf16.py tests/test_gpu_real_shapes.py -x -q -k "bf16 or Mixed or heads" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest.log
export C2D_TUNE=1
echo "== old (register staged) =="; C2D_BF16_GLDS=0 python tools/bench_conv_bf16.py igemm 2>&1 | tee $O/old.log | tail -16
echo "== glds 128x64 =="; C2D_BF16_GLDS=1 python tools/bench_conv_bf16.py igemm 2>&1 | tee $O/glds_2.log | tail -16
echo "== glds 128x128 =="; C2D_BF16_GLDS=1 C2D_IGEMM_CFG=3 python tools/bench_conv_bf16.py igemm 2>&1 | tee $O/glds_3.log | tail -16
echo "== glds 256x128 =="; C2D_BF16_GLDS=1 C2D_IGEMM_CFG=4 python tools/bench_conv_bf16.py igemm 2>&1 | tee $O/glds_4.log | tail -16
