#!/bin/bash
# usage: env_ab.sh "ENV_A" "ENV_B" [reps] [bench args]  -- alternating bench.py runs on one box under two environments, ms/step of the median region
cd $GRAFT_REPO_ROOT
A="$1"; B="$2"; N=${3:-3}; X="$4"
for i in $(seq $N); do
 for v in "$A" "$B"; do
  r=$(env $v timeout 300 python bench.py --no-cpu-baseline --no-edgeconv-stack --graph plan $X 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], 'spread', d['config']['repeat_spread'])")
  echo "[$v] : $r"
 done
done
