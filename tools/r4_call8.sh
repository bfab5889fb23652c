set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
python -m pytest tests/test_gpu_loss_curve.py -q -s 2>&1 | tail -12 > $O/r4_tests8.log
timeout 600 python tools/loss_curve.py $O/r04_loss_curve.json --steps 400 > $O/r04_loss_curve.log 2>&1
timeout 600 python tools/loss_curve.py $O/r04_loss_curve_lr003.json --steps 400 --lr 0.03 > $O/r04_loss_curve_lr003.log 2>&1
PROFILE_TAG=r04 bash tools/profile_round.sh > $O/r04_profile_round.log 2>&1
tail -12 $O/r4_tests8.log | cut -c1-400; tail -1 $O/r04_loss_curve.log | cut -c1-600
ls $O/r04_summaries
