#!/usr/bin/env python
"""Micro-benchmark of the k = 1 BatchNorm passes (R = 49152 rows) per layer width: GB/s of algorithmic traffic.
usage: python profiles/bn1_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamic-gcnn_amd"))
import torch
from dgcnn import _hip as H

R = 49152


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3


for F in (64, 256, 512, 1024):
    T = torch.randn((R, F), device="cuda")
    out = torch.empty_like(T)
    dout = torch.randn((R, F), device="cuda")
    mean, rstd, beta = torch.zeros(F, device="cuda"), torch.ones(F, device="cuda"), torch.zeros(F, device="cuda")
    red = torch.zeros((32, 2, F), dtype=torch.float64, device="cuda")
    dbeta = torch.zeros(F, device="cuda")
    dT = torch.empty_like(T)
    pr = (mean.data_ptr(), rstd.data_ptr(), beta.data_ptr(), 1)
    t1 = timeit(lambda: H.call("dgcnn_bn_act_kreduce_f32", T.data_ptr(), R, 1, F, *pr, out.data_ptr(), F, 0, 0, 0, 0, 0))
    t2 = timeit(lambda: H.call("dgcnn_bn_bwd_reduce_f32", T.data_ptr(), R, 1, F, *pr, dout.data_ptr(), F, 0, 0, 0, 0, 0, red.data_ptr()))
    t3 = timeit(lambda: H.call("dgcnn_bn_bwd_apply_f32", T.data_ptr(), R, 1, F, *pr, dout.data_ptr(), F, 0, 0, 0, 0, 0,
                               red.data_ptr(), dT.data_ptr(), 0, 0, dbeta.data_ptr(), 0.0))
    by = 4.0 * R * F
    print("F=%5d  act %6.1f us %5.0f GB/s | bwd_reduce %6.1f us %5.0f GB/s | bwd_apply(+finalize) %6.1f us %5.0f GB/s"
          % (F, t1 * 1e6, 2 * by / t1 / 1e9, t2 * 1e6, 2 * by / t2 / 1e9, t3 * 1e6, 3 * by / t3 / 1e9))
