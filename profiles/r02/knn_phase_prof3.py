"""Per-phase shader-clock totals of the pipelined k-NN kernel (knn_mfma3_kernel; library built with -DKNN_PROF=1)."""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "dynamic-gcnn_amd"))
import numpy as np, torch
from dgcnn import _engine as E, _hip as H
lib = H.load()
lib.dgcnn_knn_prof_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
out = (ctypes.c_ulonglong * 8)()
rng = np.random.default_rng(0)
for (B, N, C, k) in [(24, 2048, 64, 20), (8, 65536, 64, 20)]:
    x = torch.from_numpy(np.maximum(rng.normal(size=(B * N, C)), 0).astype(np.float32)).cuda()
    E.knn(x, B, N, k)
    lib.dgcnn_knn_prof_read(out, 1)
    E.knn(x, B, N, k)
    lib.dgcnn_knn_prof_read(out, 1)
    v = [int(a) for a in out]
    blocks = B * ((N + 63) // 64)
    tiles = (N + 63) // 64 - 1
    names = ["barrier A", "stash", "barrier B", "chain+4 rounds", "fetch issue", "leftover rounds", "filter"]
    tot = sum(v[:7])
    print("B=%d N=%d C=%d k=%d: cycles per tile: " % (B, N, C, k) + ", ".join("%s %.0f" % (n, c / blocks / tiles) for n, c in zip(names, v[:7])) +
          " | total %.0f | leftover rounds per tile %.2f" % (tot / blocks / tiles, v[7] / blocks / tiles))
