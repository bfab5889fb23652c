#!/bin/bash
cd /root/repo
for v in 0 1 0 1; do
  for c in c2 c1; do
  echo -n "skip fusable bn bwd $v $c: "
  C2D_EXP_SKIP_BN=$v timeout 300 python3 bench.py --config $c --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"
  done
done
