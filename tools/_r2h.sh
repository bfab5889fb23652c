O=gpurun_out/r2h; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -k "roi or crop" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
(cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/crop_stats -o s -- python3 $R/tools/bench_crop.py > $R/$O/crop_prof.log 2>&1)
f=$(find $O -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $O/crop_kernel_stats.csv; python3 - "$f" <<'PY'
import csv,re,sys
for r in list(csv.reader(open(sys.argv[1])))[1:7]:
    print(re.sub(r'\(.*','',r[0].replace('(anonymous namespace)::',''))[:60], r[1], float(r[3])/1000)
PY
fi
find $O -name "*kernel_trace.csv" -delete
grep -E "bwd_ws|deterministic" $O/crop_prof.log
export C2D_TUNE=1
for pad in 0 40000 100000; do echo "== glds 128x64 pad $pad"; C2D_BF16_LDS_PAD=$pad timeout 120 python tools/bench_conv_bf16.py igemm 2>&1 | tail -2; done
for pad in 0 60000; do echo "== glds 128x128 pad $pad"; C2D_IGEMM_CFG=3 C2D_BF16_LDS_PAD=$pad timeout 120 python tools/bench_conv_bf16.py igemm 2>&1 | tail -2; done
