# First-look GPU check of a build: parity tests, the headline bench, the plain-command multi-rank
# launch (same-device mode on a 1-GPU box) and the other BASELINE configs.
O=gpurun_out/${TAG:-check}
mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
python bench.py > $O/bench_c1.json 2> $O/bench_c1.err; echo "c1 rc=$?"; tail -c 600 $O/bench_c1.json | head -c 300; echo
python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_gpus2.json 2> $O/bench_gpus2.err; echo "gpus2 rc=$?"
for c in c2 c3 c4; do
  python bench.py --config $c --no-cpu-baseline > $O/bench_$c.json 2> $O/bench_$c.err; echo "$c rc=$?"
done
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join("gpurun_out", os.environ.get("TAG", "check"), "bench_*.json"))):
  try:
    d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    print(os.path.basename(f), round(d["value"], 2), d["unit"], round(d["ms_per_step"], 3), "ms", d.get("n_gpus"),
          d.get("backend", ""), {k: round(v["frac"], 3) for k, v in d.items() if isinstance(v, dict) and "frac" in v},
          d.get("cpu_baseline", {}).get("value"), d.get("cpu_baseline", {}).get("cores"))
  except Exception as e:
    print(f, "unparsed", e)
PY
