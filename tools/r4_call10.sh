set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_model.py tests/test_gpu_step_fixture.py -x -q 2>&1 | tail -6 > $O/r4_tests10.log
for CFG in c1 c2; do
  timeout 300 python bench.py --config $CFG --no-cpu-baseline > $O/r4g_bench_${CFG}.json 2> /dev/null
  C2D_BRANCH_STREAMS_FIRST=0 timeout 300 python bench.py --config $CFG --no-cpu-baseline > $O/r4g_bench_${CFG}_nofirst.json 2> /dev/null
  timeout 300 python bench.py --config $CFG --no-cpu-baseline > $O/r4g_bench_${CFG}_b.json 2> /dev/null
done
tail -4 $O/r4_tests10.log
for f in c1 c1_nofirst c1_b c2 c2_nofirst c2_b; do python3 -c "
import json
l=[x for x in open('$O/r4g_bench_$f.json') if x.startswith('{')]
d=json.loads(l[-1]) if l else None
print('$f', d['ms_per_step'] if d else 'NO LINE', d.get('host_enqueue_ms_per_step') if d else '')"; done
