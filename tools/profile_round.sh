# Round profile set (run on the GPU box from the repo root: `bash tools/profile_round.sh`).
# Kernel statistics of the default command, the same with every kernel on ONE stream
# (C2D_WGRAD_SIDE_STREAM=0: no side-stream filter gradients, no first-stage look-ahead, so a
# kernel's duration is that kernel alone), two PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs,
# never combined with tracing domains other than --kernel-trace), the bf16-mode equivalents and
# the plain benchmark lines.  Copy the summaries you quote from gpurun_out/ into profiles/.
set -x
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
T=${PROFILE_TAG:-r01s5}
B="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_stats -o s -- $B > $O/${T}_stats.log 2>&1
export C2D_WGRAD_SIDE_STREAM=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_serial_stats -o s -- $B > $O/${T}_serial_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${T}_fetch -o f -- $B --no-kernel-timing > $O/${T}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${T}_write -o w -- $B --no-kernel-timing > $O/${T}_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}b_serial_stats -o s -- $B --dtype bf16 > $O/${T}b_serial_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${T}b_fetch -o f -- $B --dtype bf16 --no-kernel-timing > $O/${T}b_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${T}b_write -o w -- $B --dtype bf16 --no-kernel-timing > $O/${T}b_write.log 2>&1
unset C2D_WGRAD_SIDE_STREAM
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}b_stats -o s -- $B --dtype bf16 > $O/${T}b_stats.log 2>&1
cd $R
python bench.py > $O/${T}_unprofiled.log 2>&1
python bench.py --dtype bf16 > $O/${T}b_unprofiled.log 2>&1
C2D_WGRAD_SIDE_STREAM=0 python bench.py --no-cpu-baseline > $O/${T}_serial_unprofiled.log 2>&1
find $O -name "*kernel_trace.csv" -path "*${T}*_fetch*" -delete; find $O -name "*kernel_trace.csv" -path "*${T}*_write*" -delete
find $O -name "*kernel_trace.csv" -path "*${T}b*" -delete
du -sh $O/${T}*
