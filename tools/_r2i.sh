O=gpurun_out/r2i; mkdir -p $O
export C2D_TUNE=1
for cfg in 5 6; do echo "== glds cfg $cfg"; C2D_IGEMM_CFG=$cfg timeout 120 python tools/bench_conv_bf16.py igemm 2>&1 | tee $O/cfg$cfg.log | tail -16; done
