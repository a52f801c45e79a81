"""Raw-coordinate k-NN call at the headline shape only (for rocprofv3 --kernel-trace --stats runs)."""
import sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "dynamic-gcnn_amd"))
import numpy as np
import torch
from dgcnn import _engine as E
rng = np.random.default_rng(0)
B, N, C, k = 24, 2048, 3, 20
x = torch.from_numpy(rng.random((B * N, C), dtype=np.float32)).cuda()
for _ in range(20):
    idx = E.knn(x, B, N, k)
torch.cuda.synchronize()
