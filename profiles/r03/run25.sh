cd /root/repo
python -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q 2>&1 | grep -v amdgpu.ids | tail -3
python profiles/r03/edge_fwd_bench.py 2>&1 | grep -v amdgpu.ids
