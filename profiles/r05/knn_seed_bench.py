#!/usr/bin/env python
"""dgcnn_knn_seeded_f32 against dgcnn_knn_f32 on real layer inputs of the configs[1] model (features of layers 1 and 2 with the
previous layer's graph as seed).  Run under rocprofv3 --kernel-trace for the per-kernel split."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dynamic-gcnn_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import dgcnn
from dgcnn import _engine as E, _hip as H
from gpu_helpers import capture_layers
if os.environ.get("KNN_BENCH_CONFIG") == "2":      # BASELINE configs[2]: (8,16384,k=40), residual x6 -- layers 1 and 5
    B, N, K, LAYERS = 8, 16384, 40, (1, 5)
    flags = dgcnn.DGCNN_FLAGS(MODEL_NAME="residual-dgcnn", EDGE_CONV_LAYERS=6, EDGE_CONV_FILTERS=64, FC_LAYERS=2, FC_FILTERS=[512, 256],
                              NUM_CLASS=2, KVALUE=K, NUM_CHANNEL=3, TRAIN=True, SEED=1)
else:
    B, N, K, LAYERS = 24, 2048, 20, (1, 2)
    flags = dgcnn.DGCNN_FLAGS(MODEL_NAME="dgcnn", EDGE_CONV_LAYERS=3, EDGE_CONV_FILTERS=[64, 64, 128], FC_LAYERS=2, FC_FILTERS=[512, 256],
                              NUM_CLASS=2, KVALUE=K, NUM_CHANNEL=3, TRAIN=True, SEED=1)
tv = dgcnn.trainval(flags).initialize()
rng = np.random.default_rng(0)
pts = torch.from_numpy(rng.random((B, N, 3), dtype=np.float32)).cuda()
lab = torch.from_numpy(rng.integers(0, 2, (B, N)).astype(np.int32)).cuda()
with capture_layers() as cap:
    tv.zero_gradients(None); tv.accum_gradient(None, [pts], [lab])


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for i in LAYERS:
    xin, idx = cap.layers["EdgeConv%d" % i]
    _, prev = cap.layers["EdgeConv%d" % (i - 1)]
    x = torch.from_numpy(xin.reshape(B * N, -1)).cuda()
    seed = torch.from_numpy(prev).cuda()
    ref = E.knn(x, B, N, K)
    got = E.knn(x, B, N, K, seed=seed)
    assert torch.equal(ref, got)
    print("layer %d: unseeded %.1f us | seeded by layer %d's graph %.1f us | seeded by its own result (tightest bound) %.1f us"
          % (i, timeit(lambda: E.knn(x, B, N, K)), i - 1, timeit(lambda: E.knn(x, B, N, K, seed=seed)), timeit(lambda: E.knn(x, B, N, K, seed=ref))))
