O=gpurun_out/r2q; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_bf16.py tests/test_gpu_real_shapes.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for c in c1 c2; do
  timeout 200 python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc=$?"
  C2D_WGRAD_PARTIALS=0 timeout 200 python bench.py --config $c --no-cpu-baseline > $O/bench_${c}_atomics.json 2> $O/bench_${c}_atomics.err; echo "$c atomics rc=$?"
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r2q/bench_*.json")):
  try:
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(os.path.basename(f), round(d["value"], 2), round(d["ms_per_step"], 3), "ms", d.get("step_ms_gpu", {}).get("p50"), {k: (round(v["frac"], 3), round(v.get("family_ms_per_step", 0), 3)) for k, v in d.items() if isinstance(v, dict) and "frac" in v and "wgrad" in k})
  except Exception as e:
    print(f, "unparsed", e)
PY
