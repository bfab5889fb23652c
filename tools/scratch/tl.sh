set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
for CFG in c1 c2; do
  B="python3 $R/bench.py --config $CFG --steps 8 --warmup 3 --no-cpu-baseline --no-kernel-timing"
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tl_${CFG} -o t -- $B > $O/tl_${CFG}.log 2>&1
  C2D_WGRAD_SIDE_STREAM=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tl_${CFG}_serial -o s -- $B > $O/tl_${CFG}_serial.log 2>&1
  find $O/tl_${CFG}_serial -name "*kernel_trace.csv" -delete
done
cd $R
for CFG in c1 c2; do
  python3 tools/trace_timeline.py $O/tl_$CFG $(find $O/tl_${CFG}_serial -name "*kernel_stats.csv" | head -1) --dump $O/tl_${CFG}_dump.txt
done
