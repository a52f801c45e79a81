#!/bin/bash
# Runs ON THE GPU BOX: variant libraries of the append-form scan (csrc/variants/lib_<fetch><acc>.so), seeded call timings, alternating.
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for v in "$@"; do
    echo "== $v"
    DGCNN_HIP_LIB=$GRAFT_REPO_ROOT/dynamic-gcnn_amd/csrc/variants/lib_$v.so python profiles/r06/knn_seeded_bench.py 2>&1 | grep seeded
  done
done
