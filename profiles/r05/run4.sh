#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run4; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_knn_grid.py tests/test_gpu_parity.py tests/test_gpu_bf16_edge_mlp.py -x -q -m gpu > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
tail -4 $O/tests.log
for i in 1 2 3; do for v in 0 1; do
  r=$(DGCNN_KNN_SEED=$v timeout 300 python bench.py --no-cpu-baseline --no-edgeconv-stack --graph plan 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=[e for e in d['roofline_extra'] if e['kernel'].startswith('knn_kernel<C64')]; print(d['ms_per_step'], d['value'], 'knn C64 avg_us', k[0]['avg_us'] if k else None)")
  echo "[DGCNN_KNN_SEED=$v] : $r"
done; done
