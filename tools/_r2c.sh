O=gpurun_out/r2c; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $O/pytest.log
python tools/bench_crop.py > $O/crop.log 2>&1; cat $O/crop.log
python bench.py --config c3 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"
python bench.py --config c4 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err; echo "c4 rc=$?"
python bench.py --no-cpu-baseline > $O/bench_c1.json 2> $O/bench_c1.err; echo "c1 rc=$?"
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r2c/bench_*.json")):
  try:
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(os.path.basename(f), round(d["value"], 2), round(d["ms_per_step"], 3), "ms", d.get("step_ms_gpu"))
  except Exception as e:
    print(f, "unparsed", e)
PY
