#!/bin/bash
# usage: ab.sh "ENV_A" "ENV_B" [reps]
A="$1"; B="$2"; N=${3:-3}
for i in $(seq $N); do
 for v in "$A" "$B"; do
  r=$(env $v timeout 300 python bench.py --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['launch_mode_calibration'])")
  echo "$v : $r"
 done
done
