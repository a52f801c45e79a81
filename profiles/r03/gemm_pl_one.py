#!/usr/bin/env python
"""One GEMM, N launches (for rocprofv3 --pmc passes and ablations).
usage: python profiles/r03/gemm_pl_one.py KIND M N K [iters] [zero]
  KIND: old_nn | old_nt | old_tn | kc0 | kc1 | tr0 | tr1   (plane kernels: form + format 0 = bf16x3 / 1 = f16x2)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "dynamic-gcnn_amd"))
import torch
from dgcnn import _engine as E, _planes as P

kind = sys.argv[1]
M, N, K = [int(a) for a in sys.argv[2:5]]
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
zero = len(sys.argv) > 6 and sys.argv[6] == "zero"
mk = (lambda *s: torch.zeros(*s, device="cuda")) if zero else (lambda *s: torch.randn(*s, device="cuda"))
ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
C = torch.zeros((M, N), device="cuda")
if kind.startswith("old"):
    ta, tb = kind == "old_tn", kind == "old_nt"
    A = mk(K, M) if ta else mk(M, K)
    B = mk(N, K) if tb else mk(K, N)
    fn = lambda: E.gemm(A, B, C, transA=ta, transB=tb)
else:
    fmt = int(kind[2])
    if kind.startswith("kc"):
        Ap, Bp = P.from_f32(mk(M, K), fmt), P.from_f32(mk(N, K), fmt)
        fn = lambda: P.gemm(P.KC, Ap, Bp, C)
    else:
        Ap, Bp = P.from_f32(mk(K, M), fmt), P.from_f32(mk(K, N), fmt)
        fn = lambda: P.gemm(P.TR, Ap, Bp, C, ws=ws)
for _ in range(3):
    fn()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    fn()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print("%s M=%d N=%d K=%d %s ablate=%s: %.3f ms  %.1f TFLOP/s" % (kind, M, N, K, "zero" if zero else "randn", os.environ.get("DGCNN_PL_ABLATE", "0"),
                                                        dt * 1e3, 2.0 * M * N * K / dt / 1e12))
