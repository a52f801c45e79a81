import os, sys
sys.path.insert(0, "dynamic-gcnn_amd")
import torch, numpy as np
import dgcnn
from dgcnn import _engine as E
rng = np.random.default_rng(0)
for (B, N, C, k) in [(24, 2048, 64, 20), (24, 2048, 3, 20)]:
    x = torch.from_numpy(np.maximum(rng.normal(size=(B * N, C)), 0).astype(np.float32)).cuda()
    for _ in range(3): E.knn(x, B, N, k)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): idx = E.knn(x, B, N, k)
    b.record(); torch.cuda.synchronize()
    print("knn B=%d N=%d C=%d k=%d: %.1f us  checksum %d" % (B, N, C, k, a.elapsed_time(b) / 20 * 1e3, int(idx.long().sum())))
