cd /root/repo
mkdir -p gpurun_out
python -m pytest tests/test_gpu_gemm_planes.py -x -q -s 2>&1 | grep -v amdgpu.ids | tail -30 > gpurun_out/t_planes.txt
for hp in 0 bf16 f16; do
  echo "== DGCNN_HEAD_PLANES=$hp" >> gpurun_out/stageB.txt
  DGCNN_HEAD_PLANES=$hp python bench.py --steps 20 --warmup 5 --no-cpu-baseline --kernel-table >> gpurun_out/stageB.txt 2>&1
done
DGCNN_HEAD_PLANES=f16 python -m pytest tests/test_gpu_baseline_sizes.py -x -q -s -k "config1" 2>&1 | grep -v amdgpu.ids | tail -30 > gpurun_out/t_f16_parity.txt
cat gpurun_out/t_planes.txt gpurun_out/stageB.txt gpurun_out/t_f16_parity.txt | grep -v amdgpu.ids | cut -c1-400
