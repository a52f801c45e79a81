#!/bin/bash
# A/B of one environment switch inside bench.py, alternating, same box: bash profiles/r03/ab.sh VAR "0 1" [pairs=4] [extra bench args]
cd /root/repo
VAR=$1; VALS=${2:-"0 1"}; PAIRS=${3:-4}; shift 3
J='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); print(d["ms_per_step"])'
for rep in $(seq $PAIRS); do for v in $VALS; do
  echo -n "$VAR=$v  "; env $VAR=$v python bench.py --steps 40 --warmup 5 --no-cpu-baseline --graph 0 --no-edgeconv-stack "$@" 2>/dev/null | python -c "$J"
done; done
