#!/bin/bash
# per-kernel split of configs[2] (8,16384,k=40, residual x6): fp32 fold vs the named bf16 edge-MLP mode
O=$GRAFT_REPO_ROOT/gpurun_out/r05_c2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for m in fp32 bf16; do
  sel="configs[2] B=8 N=16384 k=40 residual x6 (fp32"; [ $m = bf16 ] && sel="bf16 edge-MLP"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$m -- python $GRAFT_REPO_ROOT/profiles/config_sweep.py --only "$sel" > $O/$m.log 2>&1
  grep configs $O/$m.log
  f=$(ls -t $O/$m/*/*kernel_trace.csv | head -1)
  python $GRAFT_REPO_ROOT/profiles/trace_summary.py $f 8 > $O/$m.txt
  head -24 $O/$m.txt
done
