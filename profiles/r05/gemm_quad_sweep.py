#!/usr/bin/env python
"""In-kernel-split GEMM (gemm_x3.hip): the 256 x 256 unspecialised kernel (gemm_x3q_kernel, tile override 512) against the shipped
256 x 128 wave-specialised one per head shape -- forward, forward WITH the BatchNorm-sum (+ column-max) epilogue, data gradient,
weight gradient -- and a bit-for-bit comparison of the outputs (the per-element k order is the same).  R = 49152 rows."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "dynamic-gcnn_amd"))
import torch
from dgcnn import _engine as E, _hip as H

R = 49152
lib = H.load()
E.DETERMINISTIC = False


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


outs = {}
for rnd in range(2):
  for bm in (256, 512, 448):
    lib.dgcnn_gemm_x3_tile_override(bm)
    print("# %s" % {256: "256 x 128 wave-specialised (shipped rule)", 512: "256 x 256 unspecialised", 448: "192 x 256 unspecialised"}[bm])
    for name, Cin, Cout, colmax in (("FC0 1728->512", 1728, 512, False), ("Merged 192->1024", 192, 1024, True), ("FC1 512->256", 512, 256, False)):
        g = torch.Generator(device="cuda").manual_seed(7)
        X = torch.randn(R, Cin, device="cuda", generator=g).relu_()
        W = torch.randn(Cin, Cout, device="cuda", generator=g) * 0.05
        dT = torch.randn(R, Cout, device="cuda", generator=g) * 1e-3
        Y = torch.empty(R, Cout, device="cuda")
        dX = torch.empty(R, Cin, device="cuda")
        dW = torch.zeros(Cin, Cout, device="cuda")
        st = torch.zeros(H.STAT_SLOTS * 2 * Cout, dtype=torch.float64, device="cuda")
        keys = torch.zeros(24 * Cout, dtype=torch.int64, device="cuda") if colmax else None
        fl = 2.0 * R * Cin * Cout
        t = [timeit(lambda: E.gemm(X, W, Y)), timeit(lambda: E.gemm(X, W, Y, stats=st, colmax=keys, colmax_rpg=2048 if colmax else 0)),
             timeit(lambda: E.gemm(dT, W, dX, transB=True)), timeit(lambda: E.gemm(X, dT, dW, transA=True, beta=0.0))]
        print("%-18s fwd %7.1f us %6.1f TF/s | fwd+stats%s %7.1f us %6.1f | dgrad %7.1f us %6.1f | wgrad %7.1f us %6.1f" % (
            name, t[0] * 1e6, fl / t[0] / 1e12, "+colmax" if colmax else "", t[1] * 1e6, fl / t[1] / 1e12, t[2] * 1e6, fl / t[2] / 1e12,
            t[3] * 1e6, fl / t[3] / 1e12))
        st.zero_()
        E.gemm(X, W, Y, stats=st)
        cur = (Y.clone(), dX.clone(), dW.clone(), st.view(H.STAT_SLOTS, 2, Cout).sum(0).clone())
        if (name, 256) in outs and bm != 256:
            ref = outs[(name, 256)]
            print("   vs 256 x 128: Y %s  dX %s  dW %s (split-K partials differ in grouping)  column sums rel %.1e" % (
                "bit-identical" if torch.equal(cur[0], ref[0]) else "DIFFERS %.2e" % float((cur[0] - ref[0]).abs().max()),
                "bit-identical" if torch.equal(cur[1], ref[1]) else "DIFFERS %.2e" % float((cur[1] - ref[1]).abs().max()),
                "bit-identical" if torch.equal(cur[2], ref[2]) else "max rel %.1e" % float(((cur[2] - ref[2]).abs().max() / ref[2].abs().max())),
                float(((cur[3] - ref[3]).abs() / ref[3].abs().clamp_min(1e-30)).max())))
        outs[(name, bm)] = cur
lib.dgcnn_gemm_x3_tile_override(0)
