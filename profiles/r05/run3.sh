#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_run3; mkdir -p $O
timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_graph.py tests/test_gpu_deterministic.py tests/test_gpu_ref_gaps.py tests/test_gpu_loops.py tests/test_gpu_gemm_planes.py tests/test_gpu_head_planes.py -x -q -m gpu > $O/tests.log 2>&1
echo "tests rc=$?" >> $O/tests.log
tail -4 $O/tests.log
bash profiles/r05/ab.sh "" "--atomics" 2
