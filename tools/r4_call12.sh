set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_model.py tests/test_gpu_step_fixture.py tests/test_gpu_bf16.py tests/test_gpu_dp2.py tests/test_gpu_rccl.py tests/test_gpu_operating_point.py -x -q 2>&1 | tail -6 > $O/r4_tests12.log
for CFG in c1 c2; do
  for i in a b; do timeout 300 python bench.py --config $CFG --no-cpu-baseline > $O/r4i_bench_${CFG}_$i.json 2> /dev/null; done
done
tail -4 $O/r4_tests12.log
for f in c1_a c1_b c2_a c2_b; do python3 -c "
import json
l=[x for x in open('$O/r4i_bench_$f.json') if x.startswith('{')]
d=json.loads(l[-1]) if l else None
print('$f', d['ms_per_step'] if d else 'NO LINE', d.get('host_enqueue_ms_per_step') if d else '')"; done
