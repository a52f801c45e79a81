import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from oracle import dgcnn_oracle as O
from oracle import torch_twin as T
flags = O.Flags(EDGE_CONV_FILTERS=[64, 64, 128], FC_FILTERS=[512, 256], KVALUE=20, TRAIN=True)
rng = np.random.default_rng(0)
P = {n: torch.tensor(v, requires_grad=True) for n, v in O.init_params(flags, 3, seed=1).items()}
pts = torch.from_numpy(rng.random((4, 2048, 3), dtype=np.float32)); lab = torch.from_numpy(rng.integers(0, 2, (4, 2048)).astype(np.int64))
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    T.train_step(pts[:1], lab[:1], flags, P)
    t0 = time.perf_counter(); T.train_step(pts, lab, flags, P); dt = time.perf_counter() - t0
    print("threads %3d: %.2f s for 4 clouds -> %.3f clouds/s" % (nt, dt, 4 / dt), flush=True)
