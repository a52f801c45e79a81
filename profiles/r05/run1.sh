#!/bin/bash
# first contact of the launch plan / deterministic default / FC_LAYERS=0 with the GPU
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run1; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_graph.py tests/test_gpu_ref_gaps.py tests/test_gpu_rccl.py tests/test_gpu_deterministic.py -x -q -m gpu > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
for m in auto 0 plan 1; do
  timeout 300 python bench.py --no-cpu-baseline --graph $m > $O/bench_$m.json 2> $O/bench_$m.err
done
timeout 300 python bench.py --no-cpu-baseline --atomics > $O/bench_atomics.json 2> $O/bench_atomics.err
tail -5 $O/tests.log
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); c=d['config']
    print(sys.argv[1].split('/')[-1], d['ms_per_step'], d['value'], c['launch_mode'], c['launch_mode_calibration'], 'host', c['host_enqueue_ms_per_step'], 'spread', c['repeat_spread'], c['repeats'], c.get('launches_per_step'))
except Exception as e:
    print(sys.argv[1], 'ERR', e)
PY
done
