#!/usr/bin/env python
"""In-kernel-split GEMM (gemm_x3.hip): 256 x 128 wave-specialised tile (1 workgroup per CU) vs 128 x 128 (2 per CU) per head shape,
forward WITH the BatchNorm-sum epilogue (and the column-max epilogue for MergedEdgeConv), as the model issues them.  R = 49152 rows."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "dynamic-gcnn_amd"))
import torch
from dgcnn import _engine as E, _hip as H

R = 49152
lib = H.load()


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


for bm in (256, 128):
    lib.dgcnn_gemm_x3_tile_override(bm)
    print("# row tile %d" % bm)
    for name, Cin, Cout, colmax in (("FC0 1728->512", 1728, 512, False), ("Merged 192->1024", 192, 1024, True), ("FC1 512->256", 512, 256, False)):
        X = torch.randn(R, Cin, device="cuda").relu_()
        W = torch.randn(Cin, Cout, device="cuda") * 0.05
        dT = torch.randn(R, Cout, device="cuda") * 1e-3
        Y = torch.empty(R, Cout, device="cuda")
        dX = torch.empty(R, Cin, device="cuda")
        dW = torch.zeros(Cin, Cout, device="cuda")
        st = torch.zeros(H.STAT_SLOTS * 2 * Cout, dtype=torch.float64, device="cuda")
        keys = torch.zeros(24 * Cout, dtype=torch.int64, device="cuda") if colmax else None
        fl = 2.0 * R * Cin * Cout
        t = [timeit(lambda: E.gemm(X, W, Y)), timeit(lambda: E.gemm(X, W, Y, stats=st, colmax=keys, colmax_rpg=2048 if colmax else 0)),
             timeit(lambda: E.gemm(dT, W, dX, transB=True)), timeit(lambda: E.gemm(X, dT, dW, transA=True, beta=1.0))]
        print("%-18s fwd %7.1f us %6.1f TF/s | fwd+stats%s %7.1f us %6.1f | dgrad %7.1f us %6.1f | wgrad %7.1f us %6.1f" % (
            name, t[0] * 1e6, fl / t[0] / 1e12, "+colmax" if colmax else "", t[1] * 1e6, fl / t[1] / 1e12, t[2] * 1e6, fl / t[2] / 1e12,
            t[3] * 1e6, fl / t[3] / 1e12))
lib.dgcnn_gemm_x3_tile_override(0)
