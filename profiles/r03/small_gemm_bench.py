#!/usr/bin/env python
"""The short-reduction GEMMs of the EdgeConv blocks (conv1, the point-level [U|V] product, their data gradients) alone on an idle GPU:
time, and HBM GB/s of the algorithmic bytes (A read once, C written once).  R = 49152 rows.  Run with DGCNN_GEMM_X3_BM64=0|1."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "dynamic-gcnn_amd"))
import torch
from dgcnn import _engine as E

R = 49152


def timeit(fn, n=40, warm=5):
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


print("# DGCNN_GEMM_X3_BM64=%s" % os.environ.get("DGCNN_GEMM_X3_BM64", "default(1)"))
tot = 0.0
for name, Cin, Cout in (("conv1 128->64", 128, 64), ("conv1 256->64", 256, 64), ("UV 64->128", 64, 128), ("UV 64->256", 64, 256)):
    X = torch.randn(R, Cin, device="cuda").relu_()
    W = torch.randn(Cin, Cout, device="cuda") * 0.05
    dT = torch.randn(R, Cout, device="cuda") * 1e-3
    Y = torch.empty(R, Cout, device="cuda")
    dX = torch.empty(R, Cin, device="cuda")
    st = torch.zeros(E.H.STAT_SLOTS * 2 * Cout, dtype=torch.float64, device="cuda")
    by = 4.0 * R * (Cin + Cout)
    t = [timeit(lambda: E.gemm(X, W, Y)), timeit(lambda: E.gemm(X, W, Y, stats=st)), timeit(lambda: E.gemm(dT, W, dX, transB=True))]
    tot += t[1] + t[2]
    print("%-14s fwd %6.1f us %5.0f GB/s | fwd + BatchNorm sums %6.1f us | dgrad %6.1f us %5.0f GB/s" % (
        name, t[0] * 1e6, by / t[0] / 1e9, t[1] * 1e6, t[2] * 1e6, by / t[2] / 1e9))
print("sum (fwd with sums + dgrad) %.1f us" % (tot * 1e6))
