"""k-NN kernel micro-benchmark (HIP events around E.knn, sqnorm included).  DGCNN_HIP_LIB selects a variant library."""
import os
import sys
sys.path.insert(0, "dynamic-gcnn_amd")
import torch, numpy as np
from dgcnn import _engine as E
rng = np.random.default_rng(0)
print("lib:", os.environ.get("DGCNN_HIP_LIB", "default"))
for (B, N, C, k) in [(24, 2048, 64, 20), (24, 2048, 3, 20), (8, 16384, 64, 40), (8, 16384, 3, 40), (8, 65536, 64, 20), (8, 65536, 3, 20)]:
    a = rng.normal(size=(B * N, C))
    x = torch.from_numpy((np.maximum(a, 0) if C > 4 else rng.random((B * N, C))).astype(np.float32)).cuda()
    for _ in range(2): E.knn(x, B, N, k)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    n = 5
    for _ in range(n): idx = E.knn(x, B, N, k)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / n
    print("knn B=%d N=%d C=%d k=%d: %.3f ms  %.1f TFLOP/s (2 B N^2 C)" % (B, N, C, k, ms, 2.0 * B * N * N * C / ms / 1e9))
