cd /root/repo
for cap in 4096 2048 1024 768 512; do echo "GRID_CAP=$cap"; DGCNN_BN_GRID_CAP=$cap python profiles/r03/edge_fwd_bench.py 2>&1 | grep -v amdgpu.ids; done
