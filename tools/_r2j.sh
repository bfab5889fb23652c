O=gpurun_out/r2j; mkdir -p $O
timeout 120 python tools/bench_conv_bf16.py igemm 2>&1 | tee $O/auto.log | tail -3
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
for c in c1 c2 c3 c4; do timeout 200 python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc=$?"; done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r2j/bench_*.json")):
  try:
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(os.path.basename(f), round(d["value"], 2), round(d["ms_per_step"], 3), "ms", {k: (round(v["frac"], 3), round(v.get("family_ms_per_step", v.get("avg_launch_ms", 0)), 3)) for k, v in d.items() if isinstance(v, dict) and "frac" in v})
  except Exception as e:
    print(f, "unparsed", e)
PY
