#!/usr/bin/env python
"""Time of the feature-space k-NN scan on random (24,2048,64) rows; run with DGCNN_HIP_LIB = a library built with -DKNN_ABLATE=2
(no re-check, no insert: wrong results) for the floor of the bf16-filter kernel: tile staging + MFMAs + filter only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dynamic-gcnn_amd"))
import torch
import dgcnn
from dgcnn import _engine as E
B, N, K = 24, 2048, 20
x = torch.rand((B * N, 64), device="cuda")


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


print("k-NN (24,2048,64,20) incl. sqnorm: %.1f us" % timeit(lambda: E.knn(x, B, N, K)))
