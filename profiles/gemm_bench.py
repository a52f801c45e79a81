#!/usr/bin/env python
"""Micro-benchmark of dgcnn_gemm_f32 on the model's main shapes (experiments; DGCNN_GEMM_BM overrides
the row-tile height).  usage: python profiles/gemm_bench.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamic-gcnn_amd"))
import torch
from dgcnn import _engine as E

SHAPES = [  # (name, transA, transB, M, N, K)
    ("FC0 fwd NN", 0, 0, 49152, 512, 1728), ("FC0 dgrad NT", 0, 1, 49152, 1728, 512), ("FC0 wgrad TN", 1, 0, 1728, 512, 49152),
    ("merged fwd NN", 0, 0, 49152, 1024, 192), ("merged dgrad NT", 0, 1, 49152, 192, 1024), ("merged wgrad TN", 1, 0, 192, 1024, 49152),
    ("FC1 fwd NN", 0, 0, 49152, 256, 512), ("conv1 fwd NN", 0, 0, 49152, 64, 128),
]


def main():
  from dgcnn import _hip as H
  for mode in ([int(a) for a in sys.argv[1:]] or [0, 6, 9]):
    H.set_gemm_arith(mode)
    print("# arithmetic: %s" % {0: "native fp32 MFMA", 6: "bf16 split, 6 partial products", 9: "bf16 split, 9 partial products"}[mode])
    for name, ta, tb, M, N, K in SHAPES:
          A = torch.randn((K, M) if ta else (M, K), device="cuda")
          B = torch.randn((N, K) if tb else (K, N), device="cuda")
          C = torch.zeros((M, N), device="cuda")
          for _ in range(3):
              E.gemm(A, B, C, transA=bool(ta), transB=bool(tb))
          torch.cuda.synchronize()
          a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
          a.record()
          for _ in range(10):
              E.gemm(A, B, C, transA=bool(ta), transB=bool(tb))
          b.record()
          torch.cuda.synchronize()
          ms = a.elapsed_time(b) / 10
          print("%-18s M=%6d N=%5d K=%6d  %8.1f us  %6.1f TFLOP/s" % (name, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9))


if __name__ == "__main__":
    main()
