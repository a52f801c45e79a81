#!/usr/bin/env python
"""Monte-Carlo model of the selection drain of knn_mfma_kernel with the SHARED bound: a query row's 2048 candidates are split over
4 sorted lists (k = 20 each, 16 candidates per list and tile, 32 tiles); a wave = 32 rows x 2 lists; after every tile the wave runs
max-over-lanes(#survivors) insert rounds.  Bounds (all exact: a candidate above them can never enter the row's top 20):
  own     d < own list's 20th
  quad    ... and d <= max over the 4 lists of their 5th entry              (knn.hip today; published once per tile)
  pair    ... and d <= min over list pairs of max(10th, 10th)               (two lists with 10 entries <= x hold 20 candidates <= x)
  exact   ... and d <= the 20th smallest entry of the 4 lists together      (the best any bound from the previous tiles can do)
Prints insert rounds per wave-tile."""
import numpy as np

rng = np.random.default_rng(0)
K, T, P, ROWS, W = 20, 32, 16, 32, 60
INF = np.inf


def run(mode):
    tot = 0
    for w in range(W):
        d = rng.random((ROWS, 4, T * P))
        lists = np.full((ROWS, 4, K), INF)
        pub5 = np.full((ROWS, 4), INF)          # as published at the start of the tile (state after the previous tile)
        pub10 = np.full((ROWS, 4), INF)
        for t in range(T):
            pub5[:] = lists[:, :, 4]
            pub10[:] = lists[:, :, 9]
            surv = np.zeros((ROWS, 4), int)
            for r in range(ROWS):
                tau = pub5[r].max()
                if mode == "exact":
                    tau = np.sort(lists[r].ravel())[K - 1]          # the 20th smallest of everything the row has seen (previous tiles)
                if mode == "pair":
                    p = pub10[r]
                    pairs = [max(p[a], p[b]) for a in range(4) for b in range(a + 1, 4)]
                    tau = min(tau, min(pairs))
                for l in range(4):
                    thr = lists[r, l, K - 1]
                    if mode != "own":
                        thr = min(thr, np.nextafter(tau, INF))
                    for x in d[r, l, t * P:(t + 1) * P]:
                        if x < thr:
                            surv[r, l] += 1
                            if x < lists[r, l, K - 1]:
                                lst = np.append(lists[r, l], x)
                                lst.sort()
                                lists[r, l] = lst[:K]
            # two waves per candidate half: lists (0,1) in one wave, (2,3) in the other
            tot += surv[:, :2].max() + surv[:, 2:].max()
    return tot / (W * T * 2)


for m in ("own", "quad", "pair", "exact"):
    print("%-5s %.2f insert rounds per wave-tile" % (m, run(m)))
