set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_first_stage_fixture.py "tests/test_gpu_model.py" -k "first_stage_matches or graph" -q -s 2>&1 | tail -40 > $O/r4_tests4.log
for LR in 0.1 1.0; do
  timeout 300 python tools/loss_curve.py $O/r04_lc_small_$LR.json --steps 120 --hw 224 --proposals 256 --pool 4 --window 20 --lr $LR 2>&1 | tail -1 > $O/r04_lc_small_$LR.log
done
timeout 300 python bench.py --no-cpu-baseline --graph > $O/r4b_bench_c1_graph.json 2> $O/r4b_bench_c1_graph.err
timeout 300 python bench.py --no-cpu-baseline > $O/r4b_bench_c1.json 2> $O/r4b_bench_c1.err
timeout 300 python bench.py --config c2 --no-cpu-baseline --graph > $O/r4b_bench_c2_graph.json 2> $O/r4b_bench_c2_graph.err
timeout 300 python bench.py --config c2 --no-cpu-baseline > $O/r4b_bench_c2.json 2> $O/r4b_bench_c2.err
C2D_GRAPH_STREAMS=0 timeout 300 python bench.py --no-cpu-baseline --graph > $O/r4b_bench_c1_graph1s.json 2> /dev/null
tail -12 $O/r4_tests4.log; cat $O/r04_lc_small_*.log
for f in c1_graph c1 c2_graph c2 c1_graph1s; do python3 -c "
import json,sys
l=[x for x in open('$O/r4b_bench_$f.json') if x.startswith('{')]
print('$f', json.loads(l[-1])['ms_per_step'] if l else 'NO LINE')"; done
tail -5 $O/r4b_bench_c1_graph.err
