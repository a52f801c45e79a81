#!/bin/bash
# Runs ON THE GPU BOX (via gpurun) from the repo root: bench + rocprofv3 kernel trace + PMC passes.
# Outputs under gpurun_out/prof_$TAG ; summaries are produced locally by profiles/make_summaries.py and committed.
# The traced runs launch EAGERLY (--graph 0): a replayed launch plan issues the same kernels on the same streams, and the per-kernel
# event table of bench.py needs eager launches anyway; the bench line itself (bench.json) is the default command.
TAG=${1:-r05}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
python bench.py > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --repeats 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $B --steps 8 --warmup 2 > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_eager -- $B --steps 8 --warmup 2 --graph 0 > $OUT/trace_eager.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B --steps 2 --warmup 2 --graph 0 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B --steps 2 --warmup 2 --graph 0 > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- $B --steps 2 --warmup 2 --graph 0 > $OUT/pmc_sq.log 2>&1
ls -R $OUT | head -40
