"""One k-NN shape, a few launches (for rocprofv3 passes).  usage: knn_one.py [B N C k]"""
import sys
sys.path.insert(0, __file__.rsplit("/", 2)[0] + "/dynamic-gcnn_amd")
import numpy as np, torch
from dgcnn import _engine as E
B, N, C, k = [int(a) for a in sys.argv[1:5]] if len(sys.argv) > 4 else (24, 2048, 64, 20)
rng = np.random.default_rng(0)
x = torch.from_numpy(np.maximum(rng.normal(size=(B * N, C)), 0).astype(np.float32)).cuda()
for _ in range(4):
    E.knn(x, B, N, k)
torch.cuda.synchronize()
