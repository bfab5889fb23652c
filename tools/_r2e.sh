O=gpurun_out/r2e; mkdir -p $O
R=$GRAFT_REPO_ROOT
(cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/crop_stats -o s -- python3 $R/tools/bench_crop.py > $R/$O/crop_prof.log 2>&1)
f=$(find $O -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cut -d, -f1-4 "$f" | head -12; cp "$f" $O/crop_kernel_stats.csv; fi
find $O -name "*kernel_trace.csv" -delete
timeout 600 bash tools/_sweep_bf16.sh
