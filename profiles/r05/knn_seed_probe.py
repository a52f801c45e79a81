#!/usr/bin/env python
"""Feasibility of seeding the feature-space k-NN's filter with the PREVIOUS layer's graph: tau0[i] = max over the k previous
neighbours j of d(i, j) is a rigorous upper bound of the row's k-th distance (they are k actual candidates).  How many candidates
does it admit (count of d <= tau0 per row; the online threshold inserts ~k (1 + ln(N / k)) = 128 per row at N = 2048, k = 20)?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dynamic-gcnn_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import dgcnn
from gpu_helpers import capture_layers
B, N, K = 24, 2048, 20
flags = dgcnn.DGCNN_FLAGS(MODEL_NAME="dgcnn", EDGE_CONV_LAYERS=3, EDGE_CONV_FILTERS=[64, 64, 128], FC_LAYERS=2, FC_FILTERS=[512, 256],
                          NUM_CLASS=2, KVALUE=K, NUM_CHANNEL=3, TRAIN=True, SEED=1)
tv = dgcnn.trainval(flags).initialize()
rng = np.random.default_rng(0)
pts = torch.from_numpy(rng.random((B, N, 3), dtype=np.float32)).cuda()
lab = torch.from_numpy(rng.integers(0, 2, (B, N)).astype(np.int32)).cuda()
for step in range(3):
    with capture_layers() as cap:
        tv.zero_gradients(None); tv.accum_gradient(None, [pts], [lab]); tv.apply_gradient(None)
    prev = None
    for i in range(3):
        xin, idx = cap.layers["EdgeConv%d" % i]
        if prev is not None:
            x = torch.from_numpy(xin).cuda()                        # (B, N, C)
            sq = (x * x).sum(-1)
            D = sq[:, :, None] + sq[:, None, :] - 2 * torch.bmm(x, x.transpose(1, 2))
            pi = torch.from_numpy(prev.astype(np.int64)).cuda()
            dseed = torch.gather(D, 2, pi)                          # distances to the previous layer's neighbours
            tau0 = dseed.max(-1).values
            cnt = (D <= tau0[:, :, None]).sum(-1).float()
            kth = torch.topk(D, K, dim=-1, largest=False).values[..., -1]
            print("step %d layer %d: candidates with d <= tau0 per row: mean %.1f median %.0f p90 %.0f p99 %.0f max %.0f | tau0 / true k-th distance: median %.2f"
                  % (step, i, cnt.mean(), cnt.median(), cnt.quantile(0.9), cnt.quantile(0.99), cnt.max(), (tau0 / kth.clamp_min(1e-12)).median()))
            # second variant: tau1 = k-th smallest over the UNION of seeds of the row and seeds of its first seed (2 hops) -- skipped
        prev = idx
