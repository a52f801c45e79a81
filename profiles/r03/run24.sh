cd /root/repo
python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_graph.py tests/test_gpu_deterministic.py tests/test_gpu_head_planes.py -x -q 2>&1 | grep -v amdgpu.ids | tail -4
