O=gpurun_out/r2d; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/crop_stats -o s -- python3 $GRAFT_REPO_ROOT/tools/bench_crop.py > $GRAFT_REPO_ROOT/$O/crop_prof.log 2>&1
cd $GRAFT_REPO_ROOT
find $O -name "*kernel_stats.csv" | head; f=$(find $O -name "*kernel_stats.csv" | head -1); cut -d, -f1-4 $f | head -14
find $O -name "*kernel_trace.csv" -delete
