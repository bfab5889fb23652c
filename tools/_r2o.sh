O=gpurun_out/r2o; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for c in c1 c2; do timeout 200 python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc=$?"; done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r2o/bench_*.json")):
  try:
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(os.path.basename(f), round(d["value"], 2), round(d["ms_per_step"], 3), "ms", d.get("step_ms_gpu"))
  except Exception as e:
    print(f, "unparsed", e)
PY
