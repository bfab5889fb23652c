set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_model.py tests/test_gpu_step_fixture.py tests/test_gpu_bf16.py tests/test_gpu_dp2.py tests/test_gpu_rccl.py tests/test_gpu_operating_point.py tests/test_gpu_text_model.py -x -q 2>&1 | tail -5 > $O/r4_tests18.log
for CFG in c1 c2; do for i in a b; do
  timeout 300 python bench.py --config $CFG --no-cpu-baseline > $O/r4n_bench_${CFG}_$i.json 2> /dev/null
  C2D_EARLY_OPTIMIZER=0 timeout 300 python bench.py --config $CFG --no-cpu-baseline > $O/r4n_bench_${CFG}_noearly_$i.json 2> /dev/null
done; done
tail -4 $O/r4_tests18.log
for f in c1_a c1_noearly_a c1_b c1_noearly_b c2_a c2_noearly_a c2_b c2_noearly_b; do python3 -c "
import json
l=[x for x in open('$O/r4n_bench_$f.json') if x.startswith('{')]
d=json.loads(l[-1]) if l else None
print('$f', d['ms_per_step'] if d else 'NO LINE')"; done
