#!/usr/bin/env python
"""Where does the gradient deviation at (24,2048,20,3) come from?  Three-way comparison, per variable, of one training
micro-step (dropout off, the HIP path's neighbour graphs fed to both oracles):
    HIP fp32 path  vs  numpy fp64 twin      (the number the tests bound)
    numpy fp32 oracle  vs  numpy fp64 twin  (what ANY fp32 evaluation of this function is worth)
    HIP  vs  numpy fp32 oracle
Relative Frobenius norm and max-abs relative to max|ref|.  usage: python profiles/r02/grad_error_3way.py [B N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "dynamic-gcnn_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import dgcnn
from dgcnn import _engine as E
from oracle import dgcnn_oracle as O
from gpu_helpers import run_model, host

B, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (24, 2048)
flags = dgcnn.DGCNN_FLAGS(MODEL_NAME="dgcnn", EDGE_CONV_LAYERS=3, EDGE_CONV_FILTERS=[64, 64, 128], FC_LAYERS=2,
                          FC_FILTERS=[512, 256], NUM_CLASS=2, KVALUE=20, NUM_CHANNEL=3, TRAIN=True)
rng = np.random.default_rng(1)
pts = rng.random((B, N, 3), dtype=np.float32)
labels = rng.integers(0, 2, (B, N)).astype(np.int32)
params = O.init_params(flags, 3, seed=1)
E.DROPOUT_KEEP = 1.0
tv, res, cap = run_model(dgcnn, flags, pts, params, train=True, labels=labels)
g_hip = {n: host(tv.gradients[n]).astype(np.float64) for n in params}
idx_list = [cap["EdgeConv%d" % i][1] for i in range(3)]
G32, loss32, _, _ = O.train_step_grads(pts, labels, flags, params, idx_list=idx_list)
G32 = {n: v.astype(np.float64) for n, v in G32.items()}
p64 = {n: v.astype(np.float64) for n, v in params.items()}
G64, loss64, _, _ = O.train_step_grads(pts.astype(np.float64), labels, flags, p64, idx_list=idx_list)
print("loss: hip %.7f  oracle32 %.7f  oracle64 %.7f" % (float(res[2]), float(loss32), float(loss64)))
print("%-34s %10s | %-21s | %-21s | %-21s" % ("variable", "max|ref|", "hip vs f64 (fro, max)", "ora32 vs f64", "hip vs ora32"))
for n in params:
    ref = G64[n]
    sc = max(np.abs(ref).max(), 1e-30)
    fro = lambda a, b: np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
    mx = lambda a, b: np.abs(a - b).max() / sc
    print("%-34s %10.3e | %9.2e %9.2e | %9.2e %9.2e | %9.2e %9.2e" % (
        n, sc, fro(g_hip[n], ref), mx(g_hip[n], ref), fro(G32[n], ref), mx(G32[n], ref), fro(g_hip[n], G32[n]), mx(g_hip[n], G32[n])))
