# Round profile set (run on the GPU box from the repo root: `bash tools/profile_round.sh`).
# For the headline config (c1, fp32) and the bf16 config (c2): kernel statistics of the default
# command, the same with every kernel on ONE stream (C2D_TUNE=streams=0: no side-stream filter
# gradients, no first-stage look-ahead, so a kernel's duration is that kernel alone), three PMC
# passes (FETCH_SIZE, WRITE_SIZE, matrix-pipe busy cycles: separate runs, never combined with
# tracing domains other than --kernel-trace, the program directly after `--`), and the plain
# benchmark lines of all four BASELINE configs.  Copy the summaries you quote from gpurun_out/
# into profiles/ (tools/summarize_pmc.py, tools/summarize_mfma.py reduce the PMC passes).
set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
T=${PROFILE_TAG:-r05}
cd /tmp; export TMPDIR=/tmp
for CFG in ${PROFILE_CFGS:-c1 c2}; do
  B="python3 $R/bench.py --config $CFG --steps 5 --warmup 2 --no-cpu-baseline"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_${CFG}_stats -o s -- $B > $O/${T}_${CFG}_stats.log 2>&1
  export C2D_TUNE=streams=0
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_${CFG}_serial_stats -o s -- $B > $O/${T}_${CFG}_serial_stats.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${T}_${CFG}_fetch -o f -- $B --no-kernel-timing > $O/${T}_${CFG}_fetch.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${T}_${CFG}_write -o w -- $B --no-kernel-timing > $O/${T}_${CFG}_write.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/${T}_${CFG}_mfma -o m -- $B --no-kernel-timing > $O/${T}_${CFG}_mfma.log 2>&1
  unset C2D_TUNE
done
cd $R
mkdir -p $O/${T}_summaries
for CFG in ${PROFILE_CFGS:-c1 c2}; do
  python3 tools/summarize_pmc.py $O/${T}_summaries/${T}_traffic_${CFG}.json $O/${T}_${CFG}_fetch $O/${T}_${CFG}_write
  python3 tools/summarize_mfma.py $O/${T}_summaries/${T}_mfma_${CFG}.json $O/${T}_${CFG}_mfma
  for K in stats serial_stats; do
    f=$(find $O/${T}_${CFG}_${K} -name "*kernel_stats.csv" | head -1)
    if [ -n "$f" ]; then cp "$f" $O/${T}_summaries/${T}_bench_kernel_stats_${CFG}$( [ $K = serial_stats ] && echo _serial ).csv; fi
  done
done
# (bench.py quotes the PMC summaries of the SAME build: put them where it reads them)
cp $O/${T}_summaries/${T}_traffic_c*.json $O/${T}_summaries/${T}_mfma_c*.json $R/profiles/
find $O -name "*kernel_trace.csv" -path "*${T}_c*" -delete
find $O -name "*counter_collection.csv" -path "*${T}_c*" -delete
timeout 400 python bench.py > $O/${T}_summaries/${T}_bench_c1.json 2> $O/${T}_bench_c1.err
for CFG in c2 c3 c4; do timeout 300 python bench.py --config $CFG --no-cpu-baseline > $O/${T}_summaries/${T}_bench_${CFG}.json 2> $O/${T}_bench_${CFG}.err; done
C2D_TUNE=streams=0 timeout 300 python bench.py --no-cpu-baseline > $O/${T}_summaries/${T}_bench_c1_serial.json 2> $O/${T}_bench_c1_serial.err
C2D_TUNE=streams=0 timeout 300 python bench.py --config c2 --no-cpu-baseline > $O/${T}_summaries/${T}_bench_c2_serial.json 2> $O/${T}_bench_c2_serial.err
# (gloo prints its connection banner to stdout: keep the JSON line only)
timeout 300 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline 2> $O/${T}_bench_gpus2.err | grep '^{' | tail -1 > $O/${T}_summaries/${T}_bench_gpus2_same_device.json
ls -la $O/${T}_summaries; du -sh $O/${T}_*
