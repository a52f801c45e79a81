#!/bin/bash
# ON THE GPU BOX: SQ counters of the two forward gather passes of an EdgeConv layer (profiles/r03/edge_fwd_bench.py)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ef
timeout 120 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d /tmp/ef -- python /root/repo/profiles/r03/edge_fwd_bench.py > /dev/null 2>&1
python /root/repo/profiles/pmc_summary.py /tmp/ef/*/*counter_collection.csv | grep -i "kernel |\|gather_add\|kreduce" | cut -c1-260
rm -rf /tmp/ef2
timeout 120 rocprofv3 --kernel-trace --pmc TA_BUSY_sum TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d /tmp/ef2 -- python /root/repo/profiles/r03/edge_fwd_bench.py > /dev/null 2>&1
python /root/repo/profiles/pmc_summary.py /tmp/ef2/*/*counter_collection.csv | grep -i "kernel |\|gather_add\|kreduce" | cut -c1-260
