"""Raw-coordinate k-NN call over a few shapes below the cell grid's range (HIP events; DGCNN_KNN_HIST selects the histogram bound)."""
import os
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "dynamic-gcnn_amd"))
import numpy as np
import torch
from dgcnn import _engine as E
rng = np.random.default_rng(0)
for (B, N, C, k) in [(2, 512, 3, 10), (24, 512, 3, 20), (24, 1024, 3, 20), (24, 2048, 3, 20), (24, 2048, 4, 20), (8, 4095, 3, 20), (8, 4095, 3, 40), (1, 3000, 3, 40)]:
    x = torch.from_numpy(rng.random((B * N, C), dtype=np.float32)).cuda()
    for _ in range(3):
        E.knn(x, B, N, k)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        idx = E.knn(x, B, N, k)
    b.record()
    torch.cuda.synchronize()
    print("knn B=%d N=%d C=%d k=%d: %.1f us per call" % (B, N, C, k, a.elapsed_time(b) / 20 * 1e3))
