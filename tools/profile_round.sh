set -x
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
B="python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r01s4_stats -o s4 -- $B > $O/r01s4_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/r01s4_fetch -o f -- $B --no-kernel-timing > $O/r01s4_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/r01s4_write -o w -- $B --no-kernel-timing > $O/r01s4_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/r01s4b_stats -o s4b -- $B --dtype bf16 > $O/r01s4b_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/r01s4b_fetch -o f -- $B --dtype bf16 --no-kernel-timing > $O/r01s4b_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/r01s4b_write -o w -- $B --dtype bf16 --no-kernel-timing > $O/r01s4b_write.log 2>&1
cd $R
python bench.py > $O/r01s4_unprofiled.log 2>&1
python bench.py --dtype bf16 > $O/r01s4b_unprofiled.log 2>&1
find $O -name "*kernel_trace.csv" -path "*r01s4*_fetch*" -delete; find $O -name "*kernel_trace.csv" -path "*r01s4*_write*" -delete
du -sh $O/r01s4*
