cd /root/repo
python -m pytest tests/test_gpu_parity.py -x -q -k "dropout_fused" 2>&1 | grep -v amdgpu.ids | grep -B3 -A12 "AssertionError\|assert " | head -50
