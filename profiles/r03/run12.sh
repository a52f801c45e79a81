cd /root/repo
J='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); c=d["config"]; print(d["ms_per_step"], "host", c["host_enqueue_ms_per_step"], c["launch_mode"])'
for ss in 1 0; do for hp in 0 f16; do
echo "SIDE_STREAM=$ss HEAD_PLANES=$hp"; DGCNN_SIDE_STREAM=$ss DGCNN_HEAD_PLANES=$hp python bench.py --steps 30 --warmup 5 --no-cpu-baseline --graph 0 --no-edgeconv-stack 2>/dev/null | python -c "$J"
done; done
