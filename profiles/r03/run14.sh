cd /root/repo
python -m pytest tests/test_gpu_parity.py -x -q -k "first_layer_fused or model_logits or edge_conv_forward" 2>&1 | grep -v amdgpu.ids | tail -8
J='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); c=d["config"]; print(d["ms_per_step"], "host", c["host_enqueue_ms_per_step"], c["launch_mode"], c["launch_mode_calibration"], d.get("edgeconv_stack",{}).get("ms_per_pass"))'
for rep in 1 2; do for fl in 0 1; do
echo "FUSED_L0=$fl"; DGCNN_EDGE_BWD_FUSED_L0=$fl python bench.py --steps 30 --warmup 5 --no-cpu-baseline --graph 0 2>/dev/null | python -c "$J"
done; done
