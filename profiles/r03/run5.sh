cd /root/repo
mkdir -p gpurun_out
python -m pytest tests/test_gpu_head_planes.py tests/test_gpu_gemm_planes.py tests/test_gpu_graph.py -x -q 2>&1 | grep -v amdgpu.ids | tail -15 > gpurun_out/t_head.txt
rm -f gpurun_out/stageC.txt
for hp in f16 0; do
  echo "== DGCNN_HEAD_PLANES=$hp" >> gpurun_out/stageC.txt
  DGCNN_HEAD_PLANES=$hp python bench.py --steps 20 --warmup 5 --no-cpu-baseline --kernel-table 2>&1 | grep -v amdgpu.ids | grep "planes\|split\|bn1\|bn_bwd\|bn_act\|^{" >> gpurun_out/stageC.txt
done
cat gpurun_out/t_head.txt; cut -c1-330 gpurun_out/stageC.txt
