cd /root/repo
python -m pytest tests/test_gpu_deterministic.py tests/test_gpu_parity.py tests/test_golden.py -x -q 2>&1 | grep -v amdgpu.ids | tail -4
for i in 1 2; do python bench.py --steps 20 --warmup 3 --no-cpu-baseline --graph 0 --no-edgeconv-stack --deterministic 2>/dev/null | cut -c150-215; done
for i in 1 2; do python bench.py --steps 40 --warmup 5 --no-cpu-baseline --graph 0 --no-edgeconv-stack 2>/dev/null | cut -c150-215; done
