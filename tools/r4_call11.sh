set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 300 python -m cProfile -o $O/r4h_c2.prof bench.py --config c2 --no-cpu-baseline --no-kernel-timing --steps 60 --warmup 5 > $O/r4h_c2_prof.json 2> /dev/null
python - <<'PY' > $O/r4h_c2_prof.txt
import pstats
p=pstats.Stats('gpurun_out/r4h_c2.prof')
p.sort_stats('tottime').print_stats(45)
p.sort_stats('cumtime').print_stats(40)
PY
head -120 $O/r4h_c2_prof.txt
