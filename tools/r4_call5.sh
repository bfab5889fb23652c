set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
C2D_BRANCH_STREAMS=1 python -m pytest tests/test_gpu_step_fixture.py "tests/test_gpu_model.py" -k "fixture or graph" -q 2>&1 | tail -8 > $O/r4_tests5.log
for CFG in c1 c2; do
  timeout 300 python bench.py --config $CFG --no-cpu-baseline > $O/r4c_bench_${CFG}.json 2> /dev/null
  C2D_BRANCH_STREAMS=1 timeout 300 python bench.py --config $CFG --no-cpu-baseline > $O/r4c_bench_${CFG}_branch.json 2> $O/r4c_bench_${CFG}_branch.err
  C2D_BRANCH_STREAMS=1 timeout 300 python bench.py --config $CFG --no-cpu-baseline --graph > $O/r4c_bench_${CFG}_branch_graph.json 2> /dev/null
done
timeout 300 python tools/loss_curve.py $O/r04_lc_small.json --steps 120 --hw 224 --proposals 256 --pool 4 --window 20 2>&1 | tail -1 > $O/r04_lc_small.log
# trace of the multi-stream graph replay (concurrency analysis)
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/r4c_graph_trace -o g -- python3 $R/bench.py --no-cpu-baseline --graph --steps 4 --warmup 2 --no-kernel-timing > $O/r4c_graph_trace.log 2>&1
C2D_GRAPH_STREAMS=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/r4c_graph1s_trace -o g -- python3 $R/bench.py --no-cpu-baseline --graph --steps 4 --warmup 2 --no-kernel-timing > $O/r4c_graph1s_trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/r4c_eager_trace -o g -- python3 $R/bench.py --no-cpu-baseline --steps 4 --warmup 2 --no-kernel-timing > $O/r4c_eager_trace.log 2>&1
cd $R
tail -5 $O/r4_tests5.log; cat $O/r04_lc_small.log
for f in c1 c1_branch c1_branch_graph c2 c2_branch c2_branch_graph; do python3 -c "
import json
l=[x for x in open('$O/r4c_bench_$f.json') if x.startswith('{')]
print('$f', json.loads(l[-1])['ms_per_step'] if l else 'NO LINE')"; done
ls -la $O/r4c_graph_trace/ | head
