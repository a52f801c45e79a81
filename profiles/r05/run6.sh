#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_bf16_edge_mlp.py -x -q -m gpu 2>&1 | tail -4
for i in 1 2; do python profiles/config_sweep.py --only "bf16 edge-MLP" 2>&1 | grep configs; done
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r05_c2b; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bf16 -- python $GRAFT_REPO_ROOT/profiles/config_sweep.py --only "bf16 edge-MLP" > $O/bf16.log 2>&1
f=$(ls -t $O/bf16/*/*kernel_trace.csv | head -1); python $GRAFT_REPO_ROOT/profiles/trace_summary.py $f 8 | grep -E "edge_mlp|csr_gather|total"
