O=gpurun_out/r2l; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
(cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/crop_stats -o s -- python3 $R/tools/bench_crop.py > $R/$O/crop_prof.log 2>&1)
f=$(find $O -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $O/crop_kernel_stats.csv; python3 - "$f" <<'PY'
import csv,re,sys
for r in list(csv.reader(open(sys.argv[1])))[1:7]:
    print(re.sub(r'\(.*','',r[0].replace('(anonymous namespace)::',''))[:60], r[1], float(r[3])/1000)
PY
fi
find $O -name "*kernel_trace.csv" -delete
grep -E "bwd_ws|deterministic|fwd" $O/crop_prof.log
timeout 200 python bench.py --config c2 --no-cpu-baseline --graph > $O/bench_c2_graph.json 2> $O/bench_c2_graph.err; echo "c2 graph rc=$?"
timeout 200 python bench.py --config c2 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?"
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r2l/bench_*.json")):
  try:
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(os.path.basename(f), round(d["value"], 2), round(d["ms_per_step"], 3), "ms", d.get("step_ms_gpu"))
  except Exception as e:
    print(f, "unparsed", e)
PY
tail -3 $O/bench_c2_graph.err
