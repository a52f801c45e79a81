cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/plpmc2
rm -rf $O; mkdir -p $O
S="49152 512 1728 12"
for kind in old_nn kc0 kc1; do
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_EA0_RDREQ_sum" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_TAG_STALL_sum TCC_BUSY_sum"; do
    i=$((i+1))
    timeout 90 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p${i}_$kind -- python $R/profiles/r03/gemm_pl_one.py $kind $S > $O/p${i}_$kind.log 2>&1 || echo "pass $i $kind failed/timeout: $set"
  done
done
python - <<'PY'
import csv, glob, collections, os
O = "/root/repo/gpurun_out/plpmc2"
for d in sorted(glob.glob(O + "/p*_*")):
    if not os.path.isdir(d): continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:50]
            if "gemm" not in k: continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in acc:
        print(os.path.basename(d), k[28:], " ".join("%s=%.4g" % (c, sum(v[3:]) / max(len(v[3:]), 1)) for c, v in sorted(acc[k].items())))
PY
