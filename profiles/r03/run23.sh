cd /root/repo
python -m pytest tests/test_gpu_parity.py -x -q -k "dgrad or epilogue or gemm" 2>&1 | grep -v amdgpu.ids | tail -15
