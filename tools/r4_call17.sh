set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_real_shapes.py tests/test_gpu_bf16.py tests/test_gpu_step_fixture.py -x -q 2>&1 | tail -6 > $O/r4_tests17.log
for i in a b; do
  timeout 300 python bench.py --config c2 --no-cpu-baseline > $O/r4l_bench_c2_$i.json 2> /dev/null
  C2D_TUNE=1 C2D_PM_SPLIT=0 timeout 300 python bench.py --config c2 --no-cpu-baseline > $O/r4l_bench_c2_nosplit_$i.json 2> /dev/null
done
tail -4 $O/r4_tests17.log
for f in c2_a c2_nosplit_a c2_b c2_nosplit_b; do python3 -c "
import json
l=[x for x in open('$O/r4l_bench_$f.json') if x.startswith('{')]
d=json.loads(l[-1]) if l else None
print('$f', d['ms_per_step'] if d else 'NO LINE', d.get('roofline',{}).get('frac'), d.get('roofline',{}).get('family_ms_per_step'))"; done
