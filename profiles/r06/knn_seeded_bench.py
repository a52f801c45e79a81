"""Seeded feature-space k-NN call (dgcnn_knn_seeded_f32: sqnorm + seed bound + append scan + selection) on real layer-1 features:
a one-layer EdgeConv forward gives features and the graph that seeds the search.  HIP events around the call."""
import sys
sys.path.insert(0, "dynamic-gcnn_amd")
import numpy as np
import torch
import dgcnn
from dgcnn import _engine as E

for (B, N, k) in [(24, 2048, 20), (8, 16384, 40), (8, 65536, 20)]:
    dgcnn.reset()
    rng = np.random.default_rng(0)
    pts = torch.from_numpy(rng.random((B, N, 3), dtype=np.float32)).cuda()
    c = dgcnn.ctx()
    c.begin_step()
    with E.variable_scope("bench"):
        tensors = dgcnn.ops.edge_conv(pts, k=k, num_filters=64, trainable=False)
    net = tensors[2]                                     # (B, N, 1, 64)
    x, _, _ = E.as2d(net)
    seed = E.knn(E.as2d(pts)[0], B, N, k)                # layer 0's graph
    for _ in range(3):
        idx = E.knn(x, B, N, k, seed=seed)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10 if N <= 16384 else 4
    a.record()
    for _ in range(n):
        idx = E.knn(x, B, N, k, seed=seed)
    b.record()
    torch.cuda.synchronize()
    print("seeded knn B=%d N=%d C=64 k=%d: %.3f ms per call (checksum %d)" % (B, N, k, a.elapsed_time(b) / n, int(idx.long().sum())))
