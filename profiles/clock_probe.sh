#!/bin/bash
# Samples sclk / power with rocm-smi while one GEMM shape runs in a loop (GPU box).
# usage: bash profiles/clock_probe.sh MODE transA transB M N K iters
cd $GRAFT_REPO_ROOT
python profiles/gemm_one.py "$@" &
PID=$!
sleep 4
for i in 1 2 3 4 5 6; do
  /opt/rocm/bin/rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | tr '\n' ' '; echo
  sleep 0.3
done
wait $PID
