cd /root/repo
J='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); c=d["config"]; print(d["ms_per_step"], [ (e["kernel"], e["avg_us"]) for e in d["roofline_extra"] if "apply_kernel<edge>" in e["kernel"] or "csr_gather" in e["kernel"]])'
for rep in 1 2; do
echo "default"; python bench.py --steps 30 --warmup 5 --no-cpu-baseline --graph 0 --no-edgeconv-stack 2>/dev/null | python -c "$J"
echo "nontemporal dY"; DGCNN_HIP_LIB=/root/repo/dynamic-gcnn_amd/dgcnn/libdgcnn_hip_nt.so python bench.py --steps 30 --warmup 5 --no-cpu-baseline --graph 0 --no-edgeconv-stack 2>/dev/null | python -c "$J"
done
