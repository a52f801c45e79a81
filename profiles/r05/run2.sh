#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run2; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_deterministic.py tests/test_gpu_bf16_edge_mlp.py tests/test_gpu_parity.py tests/test_gpu_graph.py tests/test_gpu_dp.py -x -q -m gpu > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
tail -4 $O/tests.log
for i in 1 2; do
for m in "" "--atomics"; do
  timeout 300 python bench.py --no-cpu-baseline --no-edgeconv-stack $m 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']; print('$m', d['ms_per_step'], d['value'], c['launch_mode'], c['launch_mode_calibration']['eager_ms'], c['launch_mode_calibration']['plan_ms'], 'spread', c['repeat_spread'])"
done
done
