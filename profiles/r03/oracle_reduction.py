#!/usr/bin/env python
"""How far is the float32 oracle from exact arithmetic at (24,2048,20,3), with the BatchNorm-axis reductions as
(a) numpy's running float32 sum (round 2) and (b) tree_colsum (fold-in-half pairwise, round 3)?  CPU only.
Both float32 evaluations and the float64 twin are fed the SAME neighbour graphs (the float32 C oracle's per layer, taken
from the tree evaluation's own features).  usage: python profiles/r03/oracle_reduction.py [B N]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import dgcnn_oracle as O

B, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (24, 2048)
flags = O.Flags(EDGE_CONV_FILTERS=[64, 64, 128], FC_FILTERS=[512, 256], KVALUE=20, TRAIN=True)
rng = np.random.default_rng(1)
pts = rng.random((B, N, 3), dtype=np.float32)
labels = rng.integers(0, 2, (B, N)).astype(np.int32)
params = O.init_params(flags, 3, seed=1)

_, cache = O.model_forward(pts, flags, params)
idx_list = [cache["layers"][i]["ec"]["idx"] for i in range(3)]
del cache


def run(dtype, naive):
    p = {n: v.astype(dtype) for n, v in params.items()}
    tree_sum, tree_mean = O.tree_colsum, O.tree_colmean
    if naive:                                      # round-2 behaviour: numpy's own reduction over the leading axes
        O.tree_colsum = lambda a: a.sum(axis=tuple(range(a.ndim - 1)), dtype=a.dtype)
        O.tree_colmean = lambda a: a.mean(axis=tuple(range(a.ndim - 1)), dtype=a.dtype)
    try:
        G, loss, _, sm = O.train_step_grads(pts.astype(dtype), labels, flags, p, idx_list=idx_list)
        logits, _ = O.model_forward(pts.astype(dtype), O.Flags(EDGE_CONV_FILTERS=[64, 64, 128], FC_FILTERS=[512, 256], KVALUE=20,
                                                               TRAIN=False), p, idx_list=idx_list)
    finally:
        O.tree_colsum, O.tree_colmean = tree_sum, tree_mean
    return {n: v.astype(np.float64) for n, v in G.items()}, float(loss), logits.astype(np.float64)


G64, l64, z64 = run(np.float64, False)
for name, naive in (("float32, running sums (round 2)", True), ("float32, tree_colsum (round 3)", False)):
    G, l, z = run(np.float32, naive)
    fro = {n: np.linalg.norm(G[n] - G64[n]) / max(np.linalg.norm(G64[n]), 1e-30) for n in G}
    worst = max(fro, key=fro.get)
    print("%-34s logits max|diff| vs fp64 %.3e (mean %.1e) | loss diff %.1e | gradients rel. Frobenius: worst %.2e (%s), median %.2e"
          % (name, np.abs(z - z64).max(), np.abs(z - z64).mean(), abs(l - l64), fro[worst], worst, np.median(list(fro.values()))))
