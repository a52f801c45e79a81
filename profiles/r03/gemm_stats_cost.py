#!/usr/bin/env python
"""What the BatchNorm-statistics epilogue (LDS partial sums + fp64 slot atomics) costs the forward GEMMs of the head."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "dynamic-gcnn_amd"))
import torch
from dgcnn import _engine as E, _hip as H

R = 49152


def timeit(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3


for name, Cin, Cout in (("FC0 1728->512", 1728, 512), ("Merged 192->1024", 192, 1024), ("FC1 512->256", 512, 256), ("conv1 256->64", 256, 64), ("UV 64->256", 64, 256)):
    X = torch.randn(R, Cin, device="cuda").relu_()
    W = torch.randn(Cin, Cout, device="cuda") * 0.05
    Y = torch.empty(R, Cout, device="cuda")
    st = torch.zeros(H.STAT_SLOTS * 2 * Cout, dtype=torch.float64, device="cuda")
    t0 = timeit(lambda: E.gemm(X, W, Y))
    t1 = timeit(lambda: E.gemm(X, W, Y, stats=st))
    print("%-18s plain %7.1f us | with BatchNorm column sums %7.1f us (+%.0f %%)" % (name, t0 * 1e6, t1 * 1e6, 100 * (t1 - t0) / t0))
