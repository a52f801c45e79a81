set -x
cd /root/repo
mkdir -p gpurun_out
python -m pytest tests/test_gpu_gemm_planes.py -x -q 2>&1 | tail -30 > gpurun_out/t_planes.txt
python profiles/r03/gemm_planes_bench.py 0 > gpurun_out/planes_bench0.txt 2>&1
python profiles/r03/gemm_planes_bench.py 1 > gpurun_out/planes_bench1.txt 2>&1
python -m pytest tests/test_gpu_graph.py tests/test_gpu_deterministic.py tests/test_gpu_bench_contract.py -x -q 2>&1 | tail -30 > gpurun_out/t_misc.txt
cat gpurun_out/t_planes.txt gpurun_out/planes_bench0.txt gpurun_out/planes_bench1.txt gpurun_out/t_misc.txt
