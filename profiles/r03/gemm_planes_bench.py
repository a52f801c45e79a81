#!/usr/bin/env python
"""Plane GEMM (gemm_pl.hip) vs the in-kernel-split GEMM (gemm_x3.hip) on the head GEMMs of configs[1] (R = 49152 rows),
each kernel alone on an idle GPU: TFLOP/s of algorithmic fp32 work and fraction of the bf16x6 ceiling (2500 / 6 = 416.7).
usage: python profiles/r03/gemm_planes_bench.py [fmt]   (fmt 0 = bf16x3 planes, 1 = f16x2 planes)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "dynamic-gcnn_amd")):
    sys.path.insert(0, p)
import torch
from dgcnn import _engine as E, _planes as P, _hip as H

FMT = int(sys.argv[1]) if len(sys.argv) > 1 else 0
R = 49152
dev = "cuda"
ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)


def timeit(fn, n=20, warm=3):
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


peak = 2500.0 / (6 if FMT == 0 else 3)
print("# plane format %s; ceiling %.1f TFLOP/s of fp32-equivalent work" % ("bf16x3 (6 products)" if FMT == 0 else "f16x2 (3 products)", peak))
print("%-34s %9s %9s | %9s %8s %6s | %9s" % ("GEMM", "old ms", "old TF/s", "planes ms", "TF/s", "frac", "split ms"))
for name, Cin, Cout in (("FC0 (1728 -> 512)", 1728, 512), ("MergedEdgeConv (192 -> 1024)", 192, 1024), ("FC1 (512 -> 256)", 512, 256),
                        ("conv1 L2 (256 -> 64)", 256, 64), ("UV L2 (64 -> 256)", 64, 256)):
    X = torch.randn(R, Cin, device=dev)
    W = torch.randn(Cin, Cout, device=dev) * 0.05
    dT = torch.randn(R, Cout, device=dev) * 1e-3
    Y = torch.empty(R, Cout, device=dev)
    dX = torch.empty(R, Cin, device=dev)
    dW = torch.zeros(Cin, Cout, device=dev)
    fl = 2.0 * R * Cin * Cout
    Xp = P.PlaneSet(R, Cin, FMT, device=dev)
    dTp = P.PlaneSet(R, Cout, FMT, device=dev)
    Wt = P.from_f32(W, FMT, transpose=True)      # (Cout rows, Cin channels): B of the forward
    Wd = P.from_f32(W, FMT)                      # (Cin rows, Cout channels): B of the dgrad
    t_sx = timeit(lambda: Xp.fill_from(X))
    t_sd = timeit(lambda: dTp.fill_from(dT))
    rows = (("fwd  Y = X W", lambda: E.gemm(X, W, Y), lambda: P.gemm(P.KC, Xp, Wt, Y), t_sx),
            ("dgrad dX = dT W^T", lambda: E.gemm(dT, W, dX, transB=True), lambda: P.gemm(P.KC, dTp, Wd, dX), t_sd),
            ("wgrad dW += X^T dT", lambda: E.gemm(X, dT, dW, transA=True, beta=1.0), lambda: P.gemm(P.TR, Xp, dTp, dW, beta=1.0, ws=ws), 0.0))
    for what, old, new, ts in rows:
        to = timeit(old)
        try:
            tn = timeit(new)
        except Exception as e:
            print("%-34s %9.3f %9.1f | unsupported: %s" % (name[:14] + " " + what, to * 1e3, fl / to / 1e12, str(e)[:60]))
            continue
        print("%-34s %9.3f %9.1f | %9.3f %8.1f %6.3f | %9.3f" % (name[:14] + " " + what, to * 1e3, fl / to / 1e12, tn * 1e3, fl / tn / 1e12,
                                                                 fl / tn / 1e12 / peak, ts * 1e3))
