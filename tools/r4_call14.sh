set -x
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
(time python -m pytest tests -m gpu -q -x 2>&1 | tail -8) > $O/r4_tests14.log 2>&1
PROFILE_TAG=r04 bash tools/profile_round.sh > $O/r04_profile_round.log 2>&1
bash tools/crop_counters.sh > $O/r04_crop_counters.log 2>&1
tail -6 $O/r4_tests14.log
ls $O/r04_summaries | head -30
