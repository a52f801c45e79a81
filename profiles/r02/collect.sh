#!/bin/bash
# ON THE GPU BOX (via gpurun) from the repo root: round-2 evidence bundle -> gpurun_out/prof_r02/
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_r02
R=$GRAFT_REPO_ROOT
rm -rf $OUT; mkdir -p $OUT
cd $R
python bench.py > $OUT/bench.json 2> $OUT/bench.err
python profiles/config_sweep.py > $OUT/config_sweep.txt 2>&1
python profiles/config_sweep.py --graph > $OUT/config_sweep_graph.txt 2>&1
python profiles/r02/edge_mlp_forms.py > $OUT/edge_mlp_forms.txt 2>&1
python profiles/knn_bench.py > $OUT/knn_bench.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $OUT/trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $OUT/pmc_sq.log 2>&1
# the old split-K block order, for the traffic comparison
DGCNN_GEMM_ZMAJOR=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_zminor -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $OUT/pmc_fetch_zminor.log 2>&1
DGCNN_GEMM_ZMAJOR=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write_zminor -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline > $OUT/pmc_write_zminor.log 2>&1
find $OUT -name "*.csv" | head -30
cat $OUT/bench.json | cut -c1-300; cat $OUT/edge_mlp_forms.txt
