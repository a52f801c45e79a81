"""Layer-0 k-NN (raw coordinates): exact cell-grid search (csrc/knn_grid.hip) vs the all-pairs kernel, event-timed.
    python profiles/knn_grid_bench.py            -> table on stdout"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dynamic-gcnn_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch

import dgcnn
from dgcnn import _hip as H

lib = H.load()


def timed(fn, it=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3


print("%-34s %12s %12s %12s" % ("(B, N, C, k) / data", "all pairs us", "cell grid us", "default us"))
rng = np.random.default_rng(0)
for B, N, C, k, kind in [(24, 2048, 3, 20, "uniform"), (24, 2048, 4, 20, "uniform"), (24, 2048, 3, 20, "tracks"), (24, 2048, 3, 20, "lattice"),
                         (2, 512, 3, 10, "uniform"), (24, 1024, 3, 20, "uniform"), (24, 4096, 3, 20, "uniform"), (24, 4096, 3, 20, "tracks"), (8, 16384, 3, 40, "uniform"), (8, 65536, 3, 20, "uniform"), (8, 65536, 3, 20, "tracks")]:
    if kind == "uniform":
        pts = rng.random((B, N, C), dtype=np.float32)
    elif kind == "tracks":
        pts = np.cumsum(rng.normal(0, 0.02, (B, N, C)), axis=1).astype(np.float32)
    else:
        pts = rng.integers(0, 16, (B, N, C)).astype(np.float32)
    x = torch.from_numpy(pts).cuda()
    res = {}
    for on in (0, 2, 1):
        lib.dgcnn_knn_grid(on)
        res[on] = timed(lambda: dgcnn.ops.k_nn(x, k), it=10 if N > 4096 else 30)
    print("%-34s %12.1f %12.1f %12.1f" % ("(%d, %d, %d, %d) %s" % (B, N, C, k, kind), res[0], res[2], res[1]))
