cd /root/repo
python -m pytest tests/test_gpu_parity.py -x -q -k "gemm" 2>&1 | grep -v amdgpu.ids | tail -12
for fl in 0 1; do DGCNN_GEMM_RS=$fl python profiles/r03/small_gemm_bench.py 2>&1 | grep -v amdgpu.ids; done
