cd /root/repo
J='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); c=d["config"]; print(d["ms_per_step"], "host", c["host_enqueue_ms_per_step"], c["launch_mode"], c["launch_mode_calibration"])'
for rep in 1 2; do
echo "default env"; python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-edgeconv-stack 2>/dev/null | python -c "$J"
echo "HIP_FORCE_DEV_KERNARG=1"; HIP_FORCE_DEV_KERNARG=1 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-edgeconv-stack 2>/dev/null | python -c "$J"
echo "HIP_FORCE_DEV_KERNARG=1 GPU_MAX_HW_QUEUES=2"; HIP_FORCE_DEV_KERNARG=1 GPU_MAX_HW_QUEUES=2 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-edgeconv-stack 2>/dev/null | python -c "$J"
done
