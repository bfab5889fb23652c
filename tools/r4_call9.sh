set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_real_shapes.py tests/test_gpu_step_fixture.py tests/test_gpu_bf16.py tests/test_gpu_model.py tests/test_gpu_dp2.py -x -q 2>&1 | tail -8 > $O/r4_tests9.log
for CFG in c1 c2; do
  timeout 300 python bench.py --config $CFG --no-cpu-baseline > $O/r4f_bench_${CFG}.json 2> /dev/null
  C2D_TUNE=1 C2D_PM_LPT=0 timeout 300 python bench.py --config $CFG --no-cpu-baseline > $O/r4f_bench_${CFG}_nolpt.json 2> /dev/null
  timeout 300 python bench.py --config $CFG --no-cpu-baseline > $O/r4f_bench_${CFG}_b.json 2> /dev/null
done
tail -5 $O/r4_tests9.log
for f in c1 c1_nolpt c1_b c2 c2_nolpt c2_b; do python3 -c "
import json
l=[x for x in open('$O/r4f_bench_$f.json') if x.startswith('{')]
d=json.loads(l[-1]) if l else None
print('$f', d['ms_per_step'] if d else 'NO LINE', d.get('roofline',{}).get('frac') if d else '')"; done
