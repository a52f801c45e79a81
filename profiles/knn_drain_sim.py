#!/usr/bin/env python
"""Monte-Carlo model of the k-NN selection drain (knn.hip, MFMA kernel): a wave = 64 lanes, each lane keeps its own
top-20 list over 512 candidates met in 32 tiles of 16; after every tile the wave runs max-over-lanes(#survivors)
insert rounds.  Prints the rounds per wave for the per-tile drain, for a perfectly balanced schedule (lower bound)
and for register staging of S survivors per lane before a merge."""
import numpy as np

rng = np.random.default_rng(0)
K, T, P, L, W = 20, 32, 16, 64, 200
A, Bv, Cs = [], [], {2: [], 4: [], 8: []}
for w in range(W):
    d = rng.random((L, T * P))
    passes = np.zeros((L, T), int)
    for l in range(L):
        lst, thr = [], np.inf
        for t in range(T):
            c = 0
            for x in d[l, t * P:(t + 1) * P]:
                if x < thr:
                    c += 1
                    lst.append(x); lst.sort(); lst = lst[:K]
                    if len(lst) == K:
                        thr = lst[-1]
            passes[l, t] = c
    A.append(passes.max(0).sum())
    Bv.append(passes.sum(1).max())
    for S in Cs:
        fill, ins = np.zeros(L, int), 0
        for t in range(T):
            rem = passes[:, t].copy()
            while rem.max() > 0:
                take = np.minimum(rem, S - fill)
                fill += take; rem -= take
                if fill.max() == S:
                    ins += S; fill[:] = 0
        Cs[S].append(ins + fill.max())
print("insert rounds per wave: per-tile drain %.0f | balanced lower bound %.0f | staged S=2 %.0f S=4 %.0f S=8 %.0f"
      % (np.mean(A), np.mean(Bv), np.mean(Cs[2]), np.mean(Cs[4]), np.mean(Cs[8])))
print("mean survivors per lane %.1f" % passes.sum(1).mean())
# -> 181 | 101 | 167 | 153 | 134 ; 83.3
