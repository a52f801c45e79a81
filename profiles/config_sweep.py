#!/usr/bin/env python
"""Times one training micro-step (fwd+bwd+Adam) on the other BASELINE.json configs (fp32 path) on
one GPU, and checks that the loss is finite and that every layer's k-NN graph contains self.
These are parity-test shapes, not bench lines (bench.py reports configs[1])."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dynamic-gcnn_amd"))
import numpy as np
import torch
import dgcnn

CONFIGS = [
    ("configs[0] B=2 N=512 k=10, 1 EdgeConv", dict(MODEL_NAME="dgcnn", EDGE_CONV_LAYERS=1, EDGE_CONV_FILTERS=64, KVALUE=10), 2, 512, 3),
    ("configs[1] B=24 N=2048 k=20, 3 EdgeConv (64,64,128)", dict(MODEL_NAME="dgcnn", EDGE_CONV_LAYERS=3, EDGE_CONV_FILTERS=[64, 64, 128], KVALUE=20), 24, 2048, 3),
    ("configs[2] B=8 N=16384 k=40 residual x6 (fp32 path)", dict(MODEL_NAME="residual-dgcnn", EDGE_CONV_LAYERS=6, EDGE_CONV_FILTERS=64, KVALUE=40), 8, 16384, 3),
    ("configs[2] B=8 N=16384 k=40 residual x6, bf16 edge-MLP (named mode)", dict(MODEL_NAME="residual-dgcnn", EDGE_CONV_LAYERS=6, EDGE_CONV_FILTERS=64, KVALUE=40, EDGE_MLP_DTYPE="bf16"), 8, 16384, 3),
    ("configs[4] per-GPU B=8 N=65536 k=20, 3 EdgeConv", dict(MODEL_NAME="dgcnn", EDGE_CONV_LAYERS=3, EDGE_CONV_FILTERS=[64, 64, 128], KVALUE=20), 8, 65536, 3),
]
if "--only" in sys.argv:
    CONFIGS = [c for c in CONFIGS if sys.argv[sys.argv.index("--only") + 1] in c[0]]


def main():
    graph = "--graph" in sys.argv                  # replay the tower as a captured HIP graph (trainval.use_graph)
    if "--plan" in sys.argv:
        graph = "plan"                             # ... or from a recorded launch plan
    extra = {"HEAD_PLANES": "f16"} if "--f16-planes" in sys.argv else {}
    print("# graph replay: %s%s" % (graph, "  head GEMMs from fp16 operand planes" if extra else ""))
    for name, cfg, B, N, C in CONFIGS:
        flags = dgcnn.DGCNN_FLAGS(NUM_CLASS=2, FC_LAYERS=2, FC_FILTERS=[512, 256], TRAIN=True, NUM_CHANNEL=C, **cfg, **extra)
        tv = dgcnn.trainval(flags).initialize().use_graph(graph)
        rng = np.random.default_rng(0)
        pts = torch.from_numpy(rng.random((B, N, C), dtype=np.float32)).cuda()
        lab = torch.from_numpy(rng.integers(0, 2, (B, N)).astype(np.int32)).cuda()
        idx = dgcnn.ops.k_nn(pts, int(flags.KVALUE))
        assert bool(((idx == torch.arange(N, device="cuda", dtype=torch.int32)[None, :, None]).sum(-1) == 1).all())

        def step():
            tv.zero_gradients(None)
            r = tv.accum_gradient(None, [pts], [lab])
            tv.apply_gradient(None)
            return r
        for _ in range(3):
            r = step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 50 if B * N <= 4096 else 5
        for _ in range(n):
            r = step()
        t_host = (time.perf_counter() - t0) / n
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        loss = float(r[2])
        assert np.isfinite(loss)
        print("%-58s %9.3f ms/step (host enqueue %.3f)  %9.1f clouds/s  loss %.4f  peak mem %.1f GB" % (
            name, dt * 1e3, t_host * 1e3, B / dt, loss, torch.cuda.max_memory_allocated() / 2**30), flush=True)
        del tv
        torch.cuda.empty_cache()
        torch.cuda.reset_peak_memory_stats()


if __name__ == "__main__":
    main()
