#!/usr/bin/env python
"""One GEMM shape, one arithmetic mode, N launches (for rocprofv3 --pmc passes).
usage: python profiles/gemm_one.py MODE transA transB M N K [iters]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamic-gcnn_amd"))
import torch
from dgcnn import _engine as E
from dgcnn import _hip as H

mode, ta, tb, M, N, K = [int(a) for a in sys.argv[1:7]]
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 10
H.set_gemm_arith(mode)
A = torch.randn((K, M) if ta else (M, K), device="cuda")
B = torch.randn((N, K) if tb else (K, N), device="cuda")
C = torch.zeros((M, N), device="cuda")
for _ in range(iters):
    E.gemm(A, B, C, transA=bool(ta), transB=bool(tb))
torch.cuda.synchronize()
