"""Per-phase shader-clock totals of the phase-alternating k-NN MFMA kernel (variant library built with -DKNN_PROF=1;
timers taken by lane 0 of wave 0 of every block).  usage: DGCNN_HIP_LIB=.../libdgcnn_knn_prof.so DGCNN_KNN_PIPE=0 python knn_phase_prof.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "dynamic-gcnn_amd"))
import numpy as np, torch
from dgcnn import _engine as E, _hip as H
lib = H.load()
lib.dgcnn_knn_prof_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
out = (ctypes.c_ulonglong * 8)()
rng = np.random.default_rng(0)
for (B, N, C, k) in [(24, 2048, 64, 20), (8, 16384, 64, 40), (8, 65536, 64, 20)]:
    x = torch.from_numpy(np.maximum(rng.normal(size=(B * N, C)), 0).astype(np.float32)).cuda()
    E.knn(x, B, N, k)
    lib.dgcnn_knn_prof_read(out, 1)
    E.knn(x, B, N, k)
    lib.dgcnn_knn_prof_read(out, 1)
    v = [int(a) for a in out]
    blocks = B * ((N + 63) // 64)
    tiles = (N + 63) // 64
    names = ["fetch issue", "mfma chain", "dist+filter", "drain", "stash", "barrier wait"]
    tot = sum(v[:6])
    print("B=%d N=%d C=%d k=%d: wave 0 of %d blocks, %d tiles each; cycles per tile: " % (B, N, C, k, blocks, tiles) +
          ", ".join("%s %.0f" % (n, c / blocks / tiles) for n, c in zip(names, v[:6])) +
          " | total %.0f | drain rounds per tile %.2f" % (tot / blocks / tiles, v[7] / max(v[6], 1)))
