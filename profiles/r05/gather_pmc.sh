#!/bin/bash
# counters of the vector-memory path for the gather passes (the request ceiling), two --pmc passes
O=$GRAFT_REPO_ROOT/gpurun_out/r05_gather; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --repeats 1 --no-cpu-baseline --no-edgeconv-stack --steps 2 --warmup 2 --graph 0"
timeout 300 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $O/p1 -- $B > $O/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum --output-format csv -d $O/p2 -- $B > $O/p2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU --output-format csv -d $O/p3 -- $B > $O/p3.log 2>&1
tail -2 $O/p1.log $O/p2.log $O/p3.log
ls $O/p1/*/ $O/p2/*/ $O/p3/*/
