cd /root/repo
python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_deterministic.py -x -q 2>&1 | grep -v amdgpu.ids | tail -3
for fl in 0 1; do echo "ROWS=$fl"; DGCNN_EDGE_KREDUCE_ROWS=$fl python profiles/r03/edge_fwd_bench.py 2>&1 | grep -v amdgpu.ids; done
bash profiles/r03/ab.sh DGCNN_EDGE_KREDUCE_ROWS "0 1" 4
