"""The head's nine products at configs[1], one after the other (HIP events, 30 launches each after a warm-up); DGCNN_HIP_LIB selects a variant library."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "dynamic-gcnn_amd"))
import torch
from dgcnn import _engine as E
R = 49152
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
tot = 0.0
for name, K, N in (("Merged", 192, 1024), ("FC0", 1728, 512), ("FC1", 512, 256)):
    x = torch.randn(R, K, device="cuda"); W = torch.randn(K, N, device="cuda") * 0.05
    dT = torch.randn(R, N, device="cuda"); out = torch.empty(R, N, device="cuda")
    dx = torch.empty(R, K, device="cuda"); dW = torch.zeros(K, N, device="cuda")
    f = timeit(lambda: E.gemm(x, W, out))
    d = timeit(lambda: E.gemm(dT, W, dx, transB=True))
    w = timeit(lambda: E.gemm(x, dT, dW, transA=True, beta=1.0))
    tot += f + d + w
    print("%-6s forward %6.1f  data gradient %6.1f  weight gradient %6.1f us" % (name, f, d, w))
print("sum %.1f us" % tot)
