O=gpurun_out/r2g; mkdir -p $O
R=$GRAFT_REPO_ROOT
timeout 300 python -m pytest tests/test_gpu_ops.py tests/test_gpu_bf16.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
(cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/crop_stats -o s -- python3 $R/tools/bench_crop.py > $R/$O/crop_prof.log 2>&1)
f=$(find $O -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $O/crop_kernel_stats.csv; python3 - "$f" <<'PY'
import csv,re,sys
for r in list(csv.reader(open(sys.argv[1])))[1:7]:
    print(re.sub(r'\(.*','',r[0].replace('(anonymous namespace)::',''))[:60], r[1], float(r[3])/1000)
PY
fi
find $O -name "*kernel_trace.csv" -delete
timeout 200 python tools/bench_conv_bf16.py wgrad 2>&1 | tail -16
timeout 200 python bench.py --config c2 --no-cpu-baseline > $O/bench_c2.json 2> $O/bench_c2.err; echo "c2 rc=$?"
C2D_WGRAD_PARTIALS=0 timeout 200 python bench.py --config c2 --no-cpu-baseline > $O/bench_c2_atomics.json 2> $O/bench_c2_atomics.err; echo "c2 atomics rc=$?"
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r2g/bench_*.json")):
  try:
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(os.path.basename(f), round(d["value"], 2), round(d["ms_per_step"], 3), "ms", {k: (round(v["frac"], 3), round(v.get("family_ms_per_step", 0), 3)) for k, v in d.items() if isinstance(v, dict) and "frac" in v})
  except Exception as e:
    print(f, "unparsed", e)
PY
