#!/bin/bash
# Runs ON THE GPU BOX: bound tightening of the append-form scan every n-th tile (DGCNN_KNN_TIGHTEN_EVERY), seeded call timings.
cd $GRAFT_REPO_ROOT
for v in 1 4 8 16 1 4 8 16; do
  echo "== DGCNN_KNN_TIGHTEN_EVERY=$v"
  DGCNN_KNN_TIGHTEN_EVERY=$v python profiles/r06/knn_seeded_bench.py 2>&1 | grep seeded
done
