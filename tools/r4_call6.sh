set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_real_shapes.py tests/test_gpu_step_fixture.py tests/test_gpu_bf16.py tests/test_gpu_ops.py -x -q 2>&1 | tail -15 > $O/r4_tests6.log
for CFG in c1 c2; do
  timeout 300 python bench.py --config $CFG --no-cpu-baseline > $O/r4d_bench_${CFG}.json 2> /dev/null
  C2D_TUNE=1 C2D_PM_GROUP=32 timeout 300 python bench.py --config $CFG --no-cpu-baseline > $O/r4d_bench_${CFG}_pm32.json 2> /dev/null
  C2D_BRANCH_STREAMS=1 timeout 300 python bench.py --config $CFG --no-cpu-baseline > $O/r4d_bench_${CFG}_branch.json 2> /dev/null
done
C2D_BRANCH_STREAMS=1 python -m pytest tests/test_gpu_step_fixture.py -x -q > $O/r4_tests6b.log 2>&1
C2D_BRANCH_STREAMS=1 python -m pytest tests/test_gpu_model.py -k "graph" -x -q > $O/r4_tests6c.log 2>&1
tail -5 $O/r4_tests6.log; tail -5 $O/r4_tests6b.log; grep -v "^  File\|^Extension" $O/r4_tests6c.log | tail -25
for f in c1 c1_pm32 c1_branch c2 c2_pm32 c2_branch; do python3 -c "
import json
l=[x for x in open('$O/r4d_bench_$f.json') if x.startswith('{')]
d=json.loads(l[-1]) if l else None
print('$f', d['ms_per_step'] if d else 'NO LINE', d.get('roofline',{}).get('frac') if d else '')"; done
