#!/bin/bash
# usage: knn_ab.sh [reps]   -- alternating bench.py runs on one box: feature-space k-NN kernel (fp32 MFMA scan | bf16 filter + exact
# re-check) x seeded by the previous layer's graph (from N = 4096 on | always), ms/step of the median region
cd $GRAFT_REPO_ROOT
N=${1:-2}
for i in $(seq $N); do
 for v in "DGCNN_KNN_BF16F=auto DGCNN_KNN_SEED_MIN_N=4096" "DGCNN_KNN_BF16F=1 DGCNN_KNN_SEED_MIN_N=4096" "DGCNN_KNN_BF16F=auto DGCNN_KNN_SEED_MIN_N=0" "DGCNN_KNN_BF16F=1 DGCNN_KNN_SEED_MIN_N=0"; do
  r=$(env $v timeout 300 python bench.py --no-cpu-baseline --no-edgeconv-stack --graph plan 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], 'spread', d['config']['repeat_spread'])")
  echo "[$v] : $r"
 done
done
