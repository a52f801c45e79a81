cd /root/repo
python -m pytest tests/test_gpu_parity.py -x -q -k gemm 2>&1 | grep -v amdgpu.ids | tail -2
for nt in 0 1; do DGCNN_GEMM_NT_STORE=$nt python profiles/r03/nt_store_bench.py 2>&1 | grep -v amdgpu.ids; done
cd /tmp && export TMPDIR=/tmp
for nt in 0 1; do
  rm -rf /tmp/pf$nt
  DGCNN_GEMM_NT_STORE=$nt timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pf$nt -- python /root/repo/profiles/gemm_one.py 6 0 1 49152 1728 512 6 > /dev/null 2>&1
  echo "NT_STORE=$nt FC0 dgrad FETCH_SIZE (KB, x2 = bytes):"; python /root/repo/profiles/pmc_summary.py /tmp/pf$nt/*/*counter_collection.csv | grep -i "gemm_x3w2" | cut -c1-160
  rm -rf /tmp/pg$nt
  DGCNN_GEMM_NT_STORE=$nt timeout 120 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pg$nt -- python /root/repo/profiles/gemm_one.py 6 0 0 49152 1024 192 6 > /dev/null 2>&1
  echo "NT_STORE=$nt Merged fwd FETCH_SIZE:"; python /root/repo/profiles/pmc_summary.py /tmp/pg$nt/*/*counter_collection.csv | grep -i "gemm_x3w2" | cut -c1-160
done
cd /root/repo
J='import sys,json
for l in sys.stdin:
    if l.startswith("{"):
        d=json.loads(l); c=d["config"]; print(d["ms_per_step"], "host", c["host_enqueue_ms_per_step"], c["launch_mode"])'
for rep in 1 2; do for fl in 0 1; do
echo "NT_STORE=$fl"; DGCNN_GEMM_NT_STORE=$fl python bench.py --steps 30 --warmup 5 --no-cpu-baseline --graph 0 --no-edgeconv-stack 2>/dev/null | python -c "$J"
done; done
