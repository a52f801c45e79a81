#!/bin/bash
# ON THE GPU BOX (via gpurun) from the repo root: round-4 evidence bundle -> gpurun_out/prof_r04/
OUT=/root/repo/gpurun_out/prof_r04
R=/root/repo
rm -rf $OUT; mkdir -p $OUT
cd $R
python bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-edgeconv-stack --graph 0"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B --steps 8 --warmup 2 > $OUT/trace.log 2>&1 </dev/null
timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B --steps 2 --warmup 2 > $OUT/pmc_fetch.log 2>&1 </dev/null
timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B --steps 2 --warmup 2 > $OUT/pmc_write.log 2>&1 </dev/null
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- $B --steps 2 --warmup 2 > $OUT/pmc_sq.log 2>&1 </dev/null
# L2 side of the gather-bound passes: requests the CUs' vector caches send to the XCD L2s, and what the L2s hit / miss
timeout 200 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/pmc_l2 -- $B --steps 2 --warmup 2 > $OUT/pmc_l2.log 2>&1 </dev/null
# configs[2] in its named mode: matrix-pipe busy share of the fused bf16 edge-MLP kernels
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_cfg2 -- python $R/profiles/config_sweep.py --only "bf16 edge-MLP" > $OUT/pmc_cfg2.log 2>&1 </dev/null
python $R/profiles/step_timeline.py $(ls $OUT/trace/*/*kernel_trace.csv | head -1) 1 --all > $OUT/timeline.txt 2>&1
find $OUT -name "*.csv" | wc -l
tail -3 $OUT/pmc_l2.log
cut -c1-300 $OUT/bench.json
