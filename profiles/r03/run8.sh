cd /root/repo
J='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); c=d["config"]; print(d["ms_per_step"], "host", c["host_enqueue_ms_per_step"], c["launch_mode"], c["launch_mode_calibration"])'
for hp in 0 f16; do for wa in 0 1; do
echo "HEAD_PLANES=$hp WGRAD_AFTER_DGRAD=$wa"; DGCNN_HEAD_PLANES=$hp DGCNN_WGRAD_AFTER_DGRAD=$wa python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "$J"
done; done
