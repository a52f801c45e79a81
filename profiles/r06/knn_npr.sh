#!/bin/bash
# Runs ON THE GPU BOX: the append-form scan's filter with 3 products (two bf16 terms per operand) against 1 product (plain bf16
# operands, margin 2^-6 t), per form: DGCNN_KNN_APPEND_NPR="<N < 8192>,<N >= 8192>".  profiles/knn_bench.py times dgcnn.ops-level calls
# WITHOUT a seed (list kernels); this script times the seeded call of a 2-layer stack instead (the model's layers >= 1).
cd $GRAFT_REPO_ROOT
for v in 3,3 1,1 3,3 1,1; do
  echo "== DGCNN_KNN_APPEND_NPR=$v"
  DGCNN_KNN_APPEND_NPR=$v python profiles/r06/knn_seeded_bench.py
done
