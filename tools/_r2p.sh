O=gpurun_out/r2p; mkdir -p $O
R=$GRAFT_REPO_ROOT
for CFG in c2 c1; do
(cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace_$CFG -o t -- python3 $R/bench.py --config $CFG --steps 6 --warmup 2 --no-cpu-baseline --no-kernel-timing > $R/$O/trace_$CFG.log 2>&1)
echo "== $CFG"; python3 tools/analyze_trace.py $O/trace_$CFG 4
done
find $O -name "*kernel_trace.csv" -delete
