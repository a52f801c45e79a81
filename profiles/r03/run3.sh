cd /root/repo
mkdir -p gpurun_out
python -m pytest tests/test_gpu_gemm_planes.py -x -q 2>&1 | tail -30 > gpurun_out/t_planes.txt
S="49152 512 1728 20"
for kind in old_nn kc0 kc1; do
  for ab in 0 1 2 3 4; do
    if [ $kind = old_nn ] && [ $ab != 0 ]; then continue; fi
    DGCNN_PL_ABLATE=$ab python profiles/r03/gemm_pl_one.py $kind $S >> gpurun_out/ablate3.txt 2>&1
  done
done
python profiles/r03/gemm_planes_bench.py 0 > gpurun_out/planes_bench0.txt 2>&1
python profiles/r03/gemm_planes_bench.py 1 > gpurun_out/planes_bench1.txt 2>&1
cat gpurun_out/t_planes.txt gpurun_out/ablate3.txt gpurun_out/planes_bench0.txt gpurun_out/planes_bench1.txt | grep -v amdgpu.ids
