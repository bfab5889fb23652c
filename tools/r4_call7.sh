set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
(time python -m pytest tests -m gpu -q -x 2>&1 | tail -25) > $O/r4_tests7.log 2>&1
for CFG in c1 c2; do
  timeout 300 python bench.py --config $CFG --no-cpu-baseline > $O/r4e_bench_${CFG}.json 2> /dev/null
  C2D_FUSE_BN_BWD=1 timeout 300 python bench.py --config $CFG --no-cpu-baseline > $O/r4e_bench_${CFG}_fusebn.json 2> /dev/null
done
timeout 600 python tools/loss_curve.py $O/r04_loss_curve.json --steps 400 > $O/r04_loss_curve.log 2>&1
timeout 600 python tools/loss_curve.py $O/r04_loss_curve_lr003.json --steps 400 --lr 0.03 > $O/r04_loss_curve_lr003.log 2>&1
tail -12 $O/r4_tests7.log; tail -1 $O/r04_loss_curve.log; tail -1 $O/r04_loss_curve_lr003.log
for f in c1 c1_fusebn c2 c2_fusebn; do python3 -c "
import json
l=[x for x in open('$O/r4e_bench_$f.json') if x.startswith('{')]
d=json.loads(l[-1]) if l else None
print('$f', d['ms_per_step'] if d else 'NO LINE', d.get('roofline',{}).get('frac') if d else '')"; done
