#!/usr/bin/env python
"""conv0 of an EdgeConv layer at the configs[1] shapes, forward only, three ways (HIP events, 10 repetitions):
  literal  dgcnn_edge_mlp_f32: the (B*N*k) x 2C edge GEMM north_star describes -- E = [x_i, x_j - x_i] gathered straight into
           the LDS A tile, fp32 MFMA (v_mfma_f32_32x32x2_f32), Y (B*N*k, F) written, BatchNorm sums in the epilogue;
  fold     what the model path runs: Wcat = [Wa-Wb | Wb], [U|V] = X Wcat (point-level GEMM, bf16x6 arithmetic), then the
           BatchNorm statistics pass over y = V[nbr] + U[pt] (edge_gather_add, Y never written);
Roofline columns: literal vs the 157.3 TFLOP/s fp32-MFMA peak and vs HBM for the Y write."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "dynamic-gcnn_amd"))
import numpy as np, torch
from dgcnn import _engine as E, _hip as H

B, N, k = 24, 2048, 20
rng = np.random.default_rng(0)


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


print("%-22s %12s %12s %10s %12s | %12s %10s" % ("layer (C -> F)", "literal ms", "TFLOP/s", "of 157.3", "Y write GB/s", "fold ms", "speed-up"))
for C, F in [(3, 64), (64, 64), (64, 128)]:
    R = B * N
    x = torch.from_numpy(np.maximum(rng.normal(size=(R, C)), 0).astype(np.float32) if C > 4 else rng.random((R, C), dtype=np.float32)).cuda()
    W0 = torch.from_numpy(rng.normal(0, 0.2, (2 * C, F)).astype(np.float32)).cuda()
    idx = E.knn(x, B, N, k)
    st = torch.zeros(H.STAT_SLOTS * 2 * F, dtype=torch.float64, device="cuda")
    Y = torch.empty((R * k, F), device="cuda")

    def literal():
        H.call("dgcnn_edge_mlp_f32", x.data_ptr(), H.ld2(x), idx.data_ptr(), W0.data_ptr(), B, N, C, k, F, Y.data_ptr(), st.data_ptr())

    Cp = (C + 3) // 4 * 4
    xg = x
    if Cp != C:
        xg = torch.zeros((R, Cp), device="cuda")
        xg[:, :C] = x
    wcat = torch.zeros((Cp, 2 * F), device="cuda")
    UV = torch.empty((R, 2 * F), device="cuda")

    def fold():
        H.call("dgcnn_edge_weight_split_f32", W0.data_ptr(), C, F, wcat.data_ptr())
        E.gemm(xg, wcat, UV)
        H.call("dgcnn_edge_gather_add_f32", UV[:, F:].data_ptr(), 2 * F, UV.data_ptr(), 2 * F, idx.data_ptr(), B, N, k, F, 0, st.data_ptr())

    tl, tf = timed(literal), timed(fold)
    flops = 2.0 * R * k * 2 * C * F
    print("%-22s %12.3f %12.1f %10.3f %12.0f | %12.3f %9.1fx" % ("%d -> %d" % (C, F), tl * 1e3, flops / tl / 1e12, flops / tl / 1e12 / 157.3,
                                                             R * k * F * 4 / tl / 1e9, tf * 1e3, tl / tf))
