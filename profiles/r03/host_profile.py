#!/usr/bin/env python
"""Where does the HOST time of a training step go?  cProfile over eager steps at configs[1] (DGCNN_HEAD_PLANES as set)."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "dynamic-gcnn_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
import dgcnn
import bench

flags = bench.make_flags(dgcnn)
tv = dgcnn.trainval(flags).initialize()
rng = np.random.default_rng(0)
pts = torch.from_numpy(rng.random((24, 2048, 3), dtype=np.float32)).cuda()
lab = torch.from_numpy(rng.integers(0, 2, (24, 2048)).astype(np.int32)).cuda()


def step():
    tv.zero_gradients(None)
    tv.accum_gradient(None, [pts], [lab])
    tv.apply_gradient(None)


for _ in range(5):
    step()
torch.cuda.synchronize()
n = 20
t0 = time.perf_counter()
for _ in range(n):
    step()
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("HEAD_PLANES=%s: host enqueue %.3f ms/step, step %.3f ms" % (os.environ.get("DGCNN_HEAD_PLANES", "0"), t_issue / n * 1e3, t_all / n * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(n):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(18)
