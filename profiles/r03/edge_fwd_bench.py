#!/usr/bin/env python
"""The two forward gather passes of an EdgeConv layer alone (statistics of y = V[nbr] + U[pt]; BatchNorm + ReLU + max/mean over k):
time and the L2 gather rate (R k F 4 bytes per pass).  configs[1] shapes: R = 49152, k = 20, F = 64 / 128."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "dynamic-gcnn_amd"))
import torch
from dgcnn import _engine as E
H = E.H
B, N, k = 24, 2048, 20
R = B * N


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
    return a.elapsed_time(b) / n * 1e3


for F in (64, 128):
    UV = torch.randn(R, 2 * F, device="cuda")
    idx = torch.randint(0, N, (B, N, k), device="cuda", dtype=torch.int32)
    st = torch.zeros(H.STAT_SLOTS * 2 * F, dtype=torch.float64, device="cuda")
    mean, rstd, beta = torch.zeros(F, device="cuda"), torch.ones(F, device="cuda"), torch.zeros(F, device="cuda")
    mm = torch.empty(R, 2 * F, device="cuda")
    cnt = torch.empty(R, F, device="cuda")
    U, V = UV[:, :F], UV[:, F:]
    t0 = timeit(lambda: H.call("dgcnn_edge_gather_add_f32", V.data_ptr(), 2 * F, U.data_ptr(), 2 * F, idx.data_ptr(), B, N, k, F, 0, st.data_ptr()))
    t1 = timeit(lambda: H.call("dgcnn_edge_bn_act_kreduce_f32", V.data_ptr(), 2 * F, U.data_ptr(), 2 * F, idx.data_ptr(), B, N, k, F,
                               mean.data_ptr(), rstd.data_ptr(), beta.data_ptr(), 1, mm.data_ptr(), 2 * F, mm[:, F:].data_ptr(), 2 * F, cnt.data_ptr()))
    gb = 4.0 * R * k * F / 1e9
    print("F=%3d  statistics pass %6.1f us (%5.0f GB/s gathered) | BatchNorm+ReLU+max/mean pass %6.1f us (%5.0f GB/s)" % (F, t0, gb / t0 * 1e6, t1, gb / t1 * 1e6))
