# Same-box A/B of the round's kernel / schedule changes (run on the GPU box from the repo root:
# `bash tools/ab_round.sh > gpurun_out/r05_ab.txt`).  Boxes of the pool differ by 2-3 %, so a round's
# gain is read off alternating runs on ONE box: the default build against the same build with the
# round-5 switches turned back (fused bf16 BN/ReLU epilogues, 75 % filter-gradient slot budget,
# two-image 7x7 nine-tap slabs, three-launch OICR, one-launch Adagrad + mirrored transposes).
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
line() { python3 -c "import sys,json; b=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2 ms_per_step', round(b['ms_per_step'],4), 'p50', round(b['step_ms_gpu']['p50'],4), 'images/s', round(b['value'],1))"; }
# (round 4 already folded the BN/ReLU backward into the fp32 input-gradient GEMMs: only the bf16
#  configuration turns that switch back)
OLD_c1="C2D_TUNE=1 C2D_WGRAD3_PAIR7=0 C2D_OICR_STAGEWISE=1 C2D_ADAGRAD_MULTI=0"
OLD_c2="C2D_TUNE=1 C2D_FUSE_BN_BWD=0 C2D_WGRAD_BF16_SLOTS=100 C2D_OICR_STAGEWISE=1 C2D_ADAGRAD_MULTI=0 C2D_REFRESH_CAST=1"
for i in 1 2 3; do
  for CFG in c1 c2; do
    OLD=OLD_$CFG
    python3 bench.py --config $CFG --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | line $CFG round5
    env ${!OLD} python3 bench.py --config $CFG --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | line $CFG round4_switches
  done
done
