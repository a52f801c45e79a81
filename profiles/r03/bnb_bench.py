#!/usr/bin/env python
"""FC0 / FC1 data-gradient GEMMs alone: plain, with the BatchNorm-backward sums of the layer below in the epilogue, and the
stand-alone reduce pass those sums replace.  R = 49152."""
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
    return a.elapsed_time(b) / n * 1e3


for name, N, K, c0, F in (("FC0 dgrad (1728 <- 512), MergedEdgeConv sums", 1728, 512, 704, 1024), ("FC1 dgrad (512 <- 256), FC0 sums", 512, 256, 0, 512)):
    dT = torch.randn(R, K, device="cuda") * 1e-3
    W = torch.randn(N, K, device="cuda") * 0.05
    dX = torch.empty(R, N, device="cuda")
    T = torch.randn(R, F, device="cuda")
    mean, rstd, beta = torch.zeros(F, device="cuda"), torch.ones(F, device="cuda"), torch.zeros(F, device="cuda")
    red = torch.zeros(E.H.STAT_SLOTS * 2 * F, dtype=torch.float64, device="cuda")
    t0 = timeit(lambda: E.gemm(dT, W, dX, transB=True))
    t1 = timeit(lambda: E.H.call("dgcnn_gemm_bn_bwd_f32", R, N, K, dT.data_ptr(), K, W.data_ptr(), K, dX.data_ptr(), N, 0.0, T.data_ptr(), F,
                                 mean.data_ptr(), rstd.data_ptr(), beta.data_ptr(), 1, c0, F, red.data_ptr()))
    sl = dX[:, c0:c0 + F]
    t2 = timeit(lambda: E.H.call("dgcnn_bn_bwd_reduce_f32", T.data_ptr(), R, 1, F, mean.data_ptr(), rstd.data_ptr(), beta.data_ptr(), 1,
                                 sl.data_ptr(), N, 0, 0, 0, 0, 0, red.data_ptr()))
    print("%-48s plain %6.1f us | with sums %6.1f us (+%.1f) | stand-alone reduce pass %6.1f us" % (name, t0, t1, t1 - t0, t2))
