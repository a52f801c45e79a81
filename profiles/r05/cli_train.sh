#!/bin/bash
# 2000 optimizer steps through the run loop (CLI defaults: USE_GRAPH auto = launch plans, deterministic kernels)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05_cli; rm -rf $O; mkdir -p $O
for ug in auto 0; do
python dynamic-gcnn_amd/bin/dgcnn.py train -io synthetic -bs 24 -mbs 24 -np 2048 -ecf 64,64,128 -it 2000 -rs 500 -ss 0 -chks 0 -sd 1 -ld $O/log_$ug -ug $ug > $O/train_$ug.log 2>&1
python - $O/log_$ug <<'PY'
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/train_log-*.csv"))[0]
rows = list(csv.DictReader(open(f)))
t = [float(r["titer"]) for r in rows[100:]]
tt = [float(r["ttrain"]) for r in rows[100:]]
print(sys.argv[1], "iterations", len(rows), "mean titer %.3f ms  ttrain %.3f ms  loss %.3f -> %.3f  acc %.3f" % (1e3*sum(t)/len(t), 1e3*sum(tt)/len(tt), float(rows[0]["loss"]), float(rows[-1]["loss"]), float(rows[-1]["accuracy"])))
import torch
PY
done
head -1 $O/log_auto/train_log-*.csv; sed -n '2p;101p;501p;1001p;2001p' $O/log_auto/train_log-*.csv
