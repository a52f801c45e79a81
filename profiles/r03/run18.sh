cd /root/repo
J='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); c=d["config"]; print(d["ms_per_step"], "host", c["host_enqueue_ms_per_step"], c["launch_mode"])'
for rep in 1 2; do for fl in auto 1; do
echo "KNN_BF16F=$fl"; DGCNN_KNN_BF16F=$fl python bench.py --steps 30 --warmup 5 --no-cpu-baseline --graph 0 --no-edgeconv-stack 2>/dev/null | python -c "$J"
done; done
