cd /root/repo
python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_deterministic.py -x -q 2>&1 | grep -v amdgpu.ids | tail -3
python profiles/r03/edge_fwd_bench.py 2>&1 | grep -v amdgpu.ids
for i in 1 2 3; do python bench.py --steps 40 --warmup 5 --no-cpu-baseline --graph 0 --no-edgeconv-stack 2>/dev/null | cut -c150-215; done
