O=gpurun_out/r2r; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
timeout 200 python bench.py --no-cpu-baseline --per-call > $O/bench_c1.json 2> $O/bench_c1.err; echo "c1 rc=$?"
grep "conv_wgrad" $O/bench_c1.err | grep ", 3, 3, 2)" 
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r2r/bench_*.json")):
  try:
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(os.path.basename(f), round(d["value"], 2), round(d["ms_per_step"], 3), "ms", d.get("step_ms_gpu", {}).get("p50"), {k: (round(v["frac"], 3), round(v.get("family_ms_per_step", 0), 3)) for k, v in d.items() if isinstance(v, dict) and "frac" in v})
  except Exception as e:
    print(f, "unparsed", e)
PY
