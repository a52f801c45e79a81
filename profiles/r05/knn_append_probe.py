#!/usr/bin/env python
"""Candidates per row that the append-form scan keeps (the count array of its workspace), on the real layer inputs of the configs[1]
model: how far above k the bound (seed bound + histogram tightening) leaves the buffers."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dynamic-gcnn_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import dgcnn
from dgcnn import _engine as E, _hip as H
from gpu_helpers import capture_layers
B, N, K = 24, 2048, 20
flags = dgcnn.DGCNN_FLAGS(MODEL_NAME="dgcnn", EDGE_CONV_LAYERS=3, EDGE_CONV_FILTERS=[64, 64, 128], FC_LAYERS=2, FC_FILTERS=[512, 256],
                          NUM_CLASS=2, KVALUE=K, NUM_CHANNEL=3, TRAIN=True, SEED=1)
tv = dgcnn.trainval(flags).initialize()
rng = np.random.default_rng(0)
pts = torch.from_numpy(rng.random((B, N, 3), dtype=np.float32)).cuda()
lab = torch.from_numpy(rng.integers(0, 2, (B, N)).astype(np.int32)).cuda()
with capture_layers() as cap:
    tv.zero_gradients(None); tv.accum_gradient(None, [pts], [lab])
lib = H.load()
for i in (1, 2):
    xin, idx = cap.layers["EdgeConv%d" % i]
    _, prev = cap.layers["EdgeConv%d" % (i - 1)]
    x = torch.from_numpy(xin.reshape(B * N, -1)).cuda()
    for name, sd in (("previous layer's graph", prev), ("own result", idx)):
        seed = torch.from_numpy(np.ascontiguousarray(sd)).cuda()
        nws = int(lib.dgcnn_knn_workspace_bytes(B, N, 64, K))
        ws = torch.zeros((nws,), dtype=torch.uint8, device="cuda")
        out = torch.empty((B, N, K), dtype=torch.int32, device="cuda")
        H.call("dgcnn_knn_seeded_f32", x.data_ptr(), B, N, 64, 64, K, seed.data_ptr(), K, K, out.data_ptr(), ws.data_ptr(), nws)
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy(), idx)
        sq = (B * N * 4 + 255) // 256 * 256
        cnt = ws[2 * sq:2 * sq + 4 * B * N].view(torch.int32).cpu().numpy()
        print("layer %d seeded by %-24s: kept per row mean %.1f median %d p90 %d p99 %d max %d   (k = %d)"
              % (i, name, cnt.mean(), np.median(cnt), np.percentile(cnt, 90), np.percentile(cnt, 99), cnt.max(), K))
