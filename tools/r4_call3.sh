set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_first_stage_fixture.py tests/test_gpu_rccl.py tests/test_gpu_loss_curve.py -q -s 2>&1 | tail -40 > $O/r4_tests3.log
timeout 600 python tools/loss_curve.py $O/r04_loss_curve.json --steps 400 > $O/r04_loss_curve.log 2>&1
cd /tmp; export TMPDIR=/tmp
export C2D_WGRAD_SIDE_STREAM=0
for CFG in c1 c2; do
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/r4a_${CFG}_mfma -o m -- python3 $R/bench.py --config $CFG --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing > $O/r4a_${CFG}_mfma.log 2>&1
done
unset C2D_WGRAD_SIDE_STREAM
cd $R
for CFG in c1 c2; do python3 tools/summarize_mfma.py $O/r4a_mfma_${CFG}.json $O/r4a_${CFG}_mfma > /dev/null; done
find $O -name "*kernel_trace.csv" -path "*r4a_*" -delete
find $O -name "*counter_collection.csv" -path "*r4a_*" -delete
timeout 300 python bench.py --no-cpu-baseline > $O/r4a_bench_c1.json 2> $O/r4a_bench_c1.err
timeout 300 python bench.py --config c2 --no-cpu-baseline > $O/r4a_bench_c2.json 2> $O/r4a_bench_c2.err
tail -15 $O/r4_tests3.log; tail -3 $O/r04_loss_curve.log
