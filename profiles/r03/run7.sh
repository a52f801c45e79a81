cd /root/repo
J='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); c=d["config"]; print(d["ms_per_step"], "host", c["host_enqueue_ms_per_step"], c["launch_mode"], c["launch_mode_calibration"], d["roofline"]["achieved"], d["roofline"]["launches"])'
for hp in f16 0; do
export DGCNN_HEAD_PLANES=$hp
echo "HEAD_PLANES=$hp graph 0:";    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --graph 0 2>/dev/null | python -c "$J"
echo "auto:"; python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$J"
done
