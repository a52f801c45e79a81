cd /root/repo
python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_graph.py -x -q 2>&1 | grep -v amdgpu.ids | tail -5
J='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); c=d["config"]; print(d["ms_per_step"], "host", c["host_enqueue_ms_per_step"], c["launch_mode"])'
for rep in 1 2; do for fl in 0 1; do
echo "TWO_SOURCES=$fl"; DGCNN_BN1_BWD_TWO_SOURCES=$fl python bench.py --steps 30 --warmup 5 --no-cpu-baseline --graph 0 --no-edgeconv-stack 2>/dev/null | python -c "$J"
done; done
