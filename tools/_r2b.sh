O=gpurun_out/r2b; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $O/pytest.log
python bench.py --config c3 --no-cpu-baseline > $O/bench_c3.json 2> $O/bench_c3.err; echo "c3 rc=$?"
python bench.py --config c4 --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err; echo "c4 rc=$?"
C2D_CPU_BASELINE_THREADS=64 python bench.py --steps 3 --warmup 1 > $O/bench_cpu64.json 2> $O/bench_cpu64.err; echo "cpu64 rc=$?"
python bench.py --steps 3 --warmup 1 > $O/bench_cpuall.json 2> $O/bench_cpuall.err; echo "cpuall rc=$?"
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r2b/bench_*.json")):
  try:
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(os.path.basename(f), round(d["value"], 2), round(d["ms_per_step"], 3), "ms", d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("cores"), d.get("cpu_baseline", {}).get("step_s"))
  except Exception as e:
    print(f, "unparsed", e)
PY
