#!/usr/bin/env python
"""In-kernel-split GEMM (gemm_x3.hip): 256 x 128 wave-specialised tile (1 workgroup per CU) vs 128 x 128 (2 per CU) per head shape.
Run twice: default and DGCNN_GEMM_X3_BM=128.  R = 49152 rows."""
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


print("# DGCNN_GEMM_X3_BM=%s" % os.environ.get("DGCNN_GEMM_X3_BM", "default"))
for name, Cin, Cout in (("FC0 1728->512", 1728, 512), ("Merged 192->1024", 192, 1024), ("FC1 512->256", 512, 256),
                        ("conv1 256->64", 256, 64), ("conv1 128->64", 128, 64), ("UV 64->256", 64, 256), ("UV 64->128", 64, 128)):
    X = torch.randn(R, Cin, device="cuda").relu_()
    W = torch.randn(Cin, Cout, device="cuda") * 0.05
    dT = torch.randn(R, Cout, device="cuda") * 1e-3
    Y = torch.empty(R, Cout, device="cuda")
    dX = torch.empty(R, Cin, device="cuda")
    dW = torch.zeros(Cin, Cout, device="cuda")
    fl = 2.0 * R * Cin * Cout
    t = [timeit(lambda: E.gemm(X, W, Y)), timeit(lambda: E.gemm(dT, W, dX, transB=True)), timeit(lambda: E.gemm(X, dT, dW, transA=True, beta=1.0))]
    print("%-18s fwd %7.1f us %6.1f TF/s | dgrad %7.1f us %6.1f | wgrad %7.1f us %6.1f" % (
        name, t[0] * 1e6, fl / t[0] / 1e12, t[1] * 1e6, fl / t[1] / 1e12, t[2] * 1e6, fl / t[2] / 1e12))
