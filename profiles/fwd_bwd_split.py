#!/usr/bin/env python
"""Forward-only vs full step time at the bench shape (where does the step go?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "dynamic-gcnn_amd"))
import numpy as np, torch
import dgcnn, bench
flags = bench.make_flags(dgcnn)
tv = dgcnn.trainval(flags).initialize()
rng = np.random.default_rng(0)
pts = torch.from_numpy(rng.random((24, 2048, 3), dtype=np.float32)).cuda()
lab = torch.from_numpy(rng.integers(0, 2, (24, 2048)).astype(np.int32)).cuda()
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def full():
    tv.zero_gradients(None); tv.accum_gradient(None, [pts], [lab]); tv.apply_gradient(None)
def fwd():
    c = dgcnn.ctx(); c.begin_step(); c.recording = True
    dgcnn.model.build(pts, flags)
print("full step %.3f ms   forward (recording, train graph) %.3f ms" % (t(full), t(fwd)))
