#!/bin/bash
# reproduce "timed region 40 % slower than its own calibration, host_enqueue ~5 ms" (BENCH_r04, gpurun_out/b.json)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_stall; mkdir -p $O
line() { python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config']
    print('%-34s ms/step %.3f host_enqueue %.3f mode %s calib %s spread %s per_repeat %s' % (sys.argv[1], d['ms_per_step'], c['host_enqueue_ms_per_step'], c['launch_mode'], c.get('launch_mode_calibration'), c.get('repeat_spread'), c.get('per_repeat_ms_per_step')))
except Exception as e: print(sys.argv[1], 'ERR', e)
" "$1"; }
echo "== nproc $(nproc), load $(cat /proc/loadavg)" 
echo "== A. the round-4 bench.py (one 20-step region, eager-or-graph by a 2 % edge), 6 runs"
for i in 1 2 3 4 5 6; do DGCNN_DETERMINISTIC=0 timeout 300 python profiles/r05/bench_r04_tmp.py --no-cpu-baseline --no-edgeconv-stack 2>/dev/null | line "r04 bench.py run $i"; done
echo "== B. this round's bench.py, eager vs plan, quiet box"
timeout 300 python bench.py --no-cpu-baseline --no-edgeconv-stack --graph 0 --repeats 25 2>/dev/null | line "eager quiet"
timeout 300 python bench.py --no-cpu-baseline --no-edgeconv-stack --graph plan --repeats 25 2>/dev/null | line "plan quiet"
echo "== C. with rocm-smi polled every 0.2 s beside it (what a monitoring driver does)"
( while true; do rocm-smi --showuse --showmemuse --json > /dev/null 2>&1; sleep 0.2; done ) & SMI=$!
timeout 300 python bench.py --no-cpu-baseline --no-edgeconv-stack --graph 0 --repeats 25 2>/dev/null | line "eager + rocm-smi poll"
timeout 300 python bench.py --no-cpu-baseline --no-edgeconv-stack --graph plan --repeats 25 2>/dev/null | line "plan + rocm-smi poll"
kill $SMI 2>/dev/null; wait $SMI 2>/dev/null
echo "== D. with every host core busy (nproc spinning processes)"
PIDS=""; for i in $(seq $(nproc)); do ( while :; do :; done ) & PIDS="$PIDS $!"; done
timeout 300 python bench.py --no-cpu-baseline --no-edgeconv-stack --graph 0 --repeats 25 2>/dev/null | line "eager + all cores busy"
timeout 300 python bench.py --no-cpu-baseline --no-edgeconv-stack --graph plan --repeats 25 2>/dev/null | line "plan + all cores busy"
kill $PIDS 2>/dev/null; wait 2>/dev/null
echo "== E. with 2x oversubscription pinned to this process' CPUs"
CPUS=$(python -c "import os; print(','.join(map(str,sorted(os.sched_getaffinity(0))[:4])))")
PIDS=""; for i in 1 2 3 4 5 6 7 8; do ( taskset -c $CPUS sh -c 'while :; do :; done' ) & PIDS="$PIDS $!"; done
taskset -c $CPUS timeout 300 python bench.py --no-cpu-baseline --no-edgeconv-stack --graph 0 --repeats 25 2>/dev/null | line "eager, 4 cpus shared with 8 spinners"
taskset -c $CPUS timeout 300 python bench.py --no-cpu-baseline --no-edgeconv-stack --graph plan --repeats 25 2>/dev/null | line "plan, 4 cpus shared with 8 spinners"
kill $PIDS 2>/dev/null; wait 2>/dev/null
