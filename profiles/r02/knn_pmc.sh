#!/bin/bash
# ON THE GPU BOX: SQ counters of the k-NN kernels at (24,2048,64,k20) for the pipelined (DGCNN_KNN_PIPE=1) and the
# phase-alternating (=0) kernel.  Output: gpurun_out/knn_pmc/{pipe1,pipe0}_{a,b}.txt (pmc_summary.py tables)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/knn_pmc
rm -rf $O; mkdir -p $O
for p in 1 0; do
  export DGCNN_KNN_PIPE=$p
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/a$p -- python $R/profiles/knn_one.py > $O/a$p.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA --output-format csv -d $O/b$p -- python $R/profiles/knn_one.py > $O/b$p.log 2>&1
  python $R/profiles/pmc_summary.py $(find $O/a$p -name "*counter_collection.csv") | grep -E "kernel|knn" > $O/pipe${p}_a.txt
  python $R/profiles/pmc_summary.py $(find $O/b$p -name "*counter_collection.csv") | grep -E "kernel|knn" > $O/pipe${p}_b.txt
done
cat $O/pipe*_a.txt $O/pipe*_b.txt
