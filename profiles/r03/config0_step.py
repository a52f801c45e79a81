#!/usr/bin/env python
"""BASELINE configs[0] (B=2, N=512, k=10, one EdgeConv layer): a few training micro-steps for rocprofv3 (eager or --graph)."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "dynamic-gcnn_amd"))
import numpy as np
import torch
import dgcnn

flags = dgcnn.DGCNN_FLAGS(NUM_CLASS=2, FC_LAYERS=2, FC_FILTERS=[512, 256], TRAIN=True, NUM_CHANNEL=3, MODEL_NAME="dgcnn",
                          EDGE_CONV_LAYERS=1, EDGE_CONV_FILTERS=64, KVALUE=10)
tv = dgcnn.trainval(flags).initialize().use_graph("--graph" in sys.argv)
rng = np.random.default_rng(0)
pts = torch.from_numpy(rng.random((2, 512, 3), dtype=np.float32)).cuda()
lab = torch.from_numpy(rng.integers(0, 2, (2, 512)).astype(np.int32)).cuda()
for _ in range(12):
    tv.zero_gradients(None)
    tv.accum_gradient(None, [pts], [lab])
    tv.apply_gradient(None)
torch.cuda.synchronize()
