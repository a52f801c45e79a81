#!/usr/bin/env python
"""Head GEMMs alone: time with plain vs nontemporal stores of the output tile ($DGCNN_GEMM_NT_STORE=0|1).  R = 49152."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "dynamic-gcnn_amd"))
import torch
from dgcnn import _engine as E

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


print("# DGCNN_GEMM_NT_STORE=%s" % os.environ.get("DGCNN_GEMM_NT_STORE", "default(1)"))
for name, Cin, Cout in (("FC0 1728->512", 1728, 512), ("Merged 192->1024", 192, 1024), ("FC1 512->256", 512, 256)):
    X = torch.randn(R, Cin, device="cuda").relu_()
    W = torch.randn(Cin, Cout, device="cuda") * 0.05
    dT = torch.randn(R, Cout, device="cuda") * 1e-3
    Y = torch.empty(R, Cout, device="cuda")
    dX = torch.empty(R, Cin, device="cuda")
    fl = 2.0 * R * Cin * Cout
    t = [timeit(lambda: E.gemm(X, W, Y)), timeit(lambda: E.gemm(dT, W, dX, transB=True))]
    print("%-18s fwd %7.1f us %6.1f TF/s | dgrad %7.1f us %6.1f TF/s" % (name, t[0] * 1e6, fl / t[0] / 1e12, t[1] * 1e6, fl / t[1] / 1e12))
