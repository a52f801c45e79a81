#!/bin/bash
# Runs ON THE GPU BOX: alternating A/B of environment switches on one box.  usage: ab.sh <reps> NAME=ENVVAR=VALUE ...   (X=0 for "base")
cd $GRAFT_REPO_ROOT
reps=$1; shift
for rep in $(seq $reps); do
  for spec in "$@"; do
    name=${spec%%=*}; kv=${spec#*=}
    ms=$(env $kv python bench.py --no-cpu-baseline --no-edgeconv-stack --graph plan --repeats 9 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], min(d['config']['per_repeat_ms_per_step']))")
    echo "$name $ms"
  done
done
