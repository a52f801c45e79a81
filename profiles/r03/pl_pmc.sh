# PMC passes + ablations of the plane GEMM at the FC0 forward shape (49152 x 512 x 1728)
cd /tmp && export TMPDIR=/tmp
R=/root/repo
O=$R/gpurun_out/plpmc
rm -rf $O; mkdir -p $O
S="49152 512 1728 20"
for kind in old_nn kc0 kc1; do
  for ab in 0 1 2 3 4; do
    if [ $kind = old_nn ] && [ $ab != 0 ]; then continue; fi
    DGCNN_PL_ABLATE=$ab python $R/profiles/r03/gemm_pl_one.py $kind $S >> $O/ablate.txt 2>&1
  done
  python $R/profiles/r03/gemm_pl_one.py $kind $S zero >> $O/ablate.txt 2>&1
done
python $R/profiles/r03/gemm_pl_one.py tr0 1728 512 49152 20 >> $O/ablate.txt 2>&1
python $R/profiles/r03/gemm_pl_one.py tr1 1728 512 49152 20 >> $O/ablate.txt 2>&1
grep -v amdgpu.ids $O/ablate.txt
for kind in old_nn kc0 kc1; do
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $O/a_$kind -- python $R/profiles/r03/gemm_pl_one.py $kind $S > $O/a_$kind.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA --output-format csv -d $O/b_$kind -- python $R/profiles/r03/gemm_pl_one.py $kind $S > $O/b_$kind.log 2>&1
done
python - <<'PY'
import csv, glob, collections, os
O = "/root/repo/gpurun_out/plpmc"
for d in sorted(glob.glob(O + "/[ab]_*")):
    if not os.path.isdir(d): continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            if "gemm" not in k: continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:60]
            if "gemm" in k: dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    for k in acc:
        n = len(dur[k])
        us = sum(dur[k][3:]) / max(len(dur[k][3:]), 1)
        print(os.path.basename(d), k, "launches", n, "avg_us %.1f" % us, " ".join("%s=%.4g" % (c, sum(v[3:]) / max(len(v[3:]), 1)) for c, v in sorted(acc[k].items())))
PY
