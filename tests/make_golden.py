#!/usr/bin/env python
"""Generates tests/golden/*.npz -- small known-answer vectors for the hot path.

The reference cannot run here (Python 2 + TensorFlow 1.x) and ships no
tests, so these vectors are produced by the oracle (oracle/dgcnn_oracle.py + oracle/knn_oracle.c),
float outputs by its float64 twin.  They pin (a) the oracle itself against regressions and (b) the
HIP path on the GPU box, where neither /root/reference nor this generator's choices are available.
Inputs are stored in the files (not re-derived), so a fixture is pure data: inputs + expected outputs.

    python tests/make_golden.py        # rewrites tests/golden/
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import dgcnn_oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def save(name, **arrays):
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **arrays)
    print("%-28s %s" % (name, {k: v.shape for k, v in arrays.items()}))


def main():
    os.makedirs(OUT, exist_ok=True)
    # G1: k_nn on uniform points + G5: the edge tensor for the same cloud
    rng = np.random.default_rng(0)
    pts = rng.random((2, 64, 3), dtype=np.float32)
    idx = O.k_nn(pts, 5)
    save("g1_knn_uniform", points=pts, k=np.int32(5), idx=idx, edges=O.edges(pts, 5, idx))
    # G2: integer coordinates: every accumulation order gives the same bits; pins tie rule + self inclusion
    pts = rng.integers(0, 6, (2, 100, 3)).astype(np.float32)
    save("g2_knn_integer_ties", points=pts, k=np.int32(7), idx=O.k_nn(pts, 7))
    # G3: 64-d post-ReLU-like features (dynamic graph of layers >= 1), k = 20
    pts = np.maximum(rng.normal(0, 1, (1, 128, 64)), 0).astype(np.float32)
    save("g3_knn_features", points=pts, k=np.int32(20), idx=O.k_nn(pts, 20))
    # G6: edge_conv forward + backward, B=2 N=64 k=5 C=3 F=8, float64 twin
    pts = rng.random((2, 64, 3), dtype=np.float32)
    W0 = rng.normal(0, 0.5, (6, 8)).astype(np.float32)
    b0 = rng.normal(0, 0.3, 8).astype(np.float32)
    W1 = rng.normal(0, 0.3, (16, 64)).astype(np.float32)
    b1 = rng.normal(0, 0.3, 64).astype(np.float32)
    idx = O.k_nn(pts, 5)
    outs, cache = O.edge_conv(pts.astype(np.float64), 5, W0.astype(np.float64), b0.astype(np.float64),
                              W1.astype(np.float64), b1.astype(np.float64), idx=idx)
    d = [rng.normal(size=o.shape) for o in outs]
    dx, g = O.edge_conv_bwd(d[0], d[1], d[2], cache)
    save("g6_edge_conv", points=pts, k=np.int32(5), W0=W0, beta0=b0, W1=W1, beta1=b1, idx=idx,
         net_max=outs[0], net_mean=outs[1], net=outs[2], d_max=d[0], d_mean=d[1], d_net=d[2],
         dx=dx, dW0=g["W0"], dbeta0=g["beta0"], dW1=g["W1"], dbeta1=g["beta1"])
    # G7 / G8: model logits (TRAIN False) for the config-1 shape and a residual stack; parameters are
    # Xavier draws of numpy default_rng(1) in variable-creation order, betas perturbed, all stored.
    for name, cfg, shape in (
            ("g7_model_config1", dict(MODEL_NAME="dgcnn", EDGE_CONV_LAYERS=1, EDGE_CONV_FILTERS=64, KVALUE=10,
                                      FC_FILTERS=[64, 32]), (2, 512, 3)),
            ("g8_model_residual", dict(MODEL_NAME="residual-dgcnn", EDGE_CONV_LAYERS=2, EDGE_CONV_FILTERS=64,
                                       KVALUE=8, FC_FILTERS=[32, 16]), (2, 96, 4))):
        flags = O.Flags(TRAIN=False, NUM_CLASS=2, FC_LAYERS=2, **cfg)
        pts = rng.random(shape, dtype=np.float32)
        labels = rng.integers(0, 2, shape[:2]).astype(np.int32)
        params = O.init_params(flags, shape[2], seed=1)
        for n in params:
            if n.endswith("beta"):
                params[n] = rng.normal(0, 0.2, params[n].shape).astype(np.float32)
        logits32, cache = O.model_forward(pts, flags, params)
        idxs = [l["ec"]["idx"] for l in cache["layers"]]
        p64 = {n: v.astype(np.float64) for n, v in params.items()}
        logits, _ = O.model_forward(pts.astype(np.float64), flags, p64, idx_list=idxs)
        loss, sm, acc, _ = O.softmax_xent(logits, labels)
        arrays = {"param:" + n: v for n, v in params.items()}
        arrays.update({"idx%d" % i: a for i, a in enumerate(idxs)})
        save(name, points=pts, labels=labels, logits=logits, softmax=sm, loss=np.float64(loss),
             accuracy=np.float64(acc), cfg=np.array(repr(cfg)), **arrays)
    # G9: two accumulated micro-steps on two replicas -> mean over replicas, sum over steps -> one Adam step
    flags = O.Flags(EDGE_CONV_LAYERS=1, KVALUE=6, FC_FILTERS=[32, 16], TRAIN=True, EDGE_CONV_FILTERS=64)
    params = {n: v.astype(np.float64) for n, v in O.init_params(flags, 3, seed=1).items()}
    acc = {n: np.zeros_like(v) for n, v in params.items()}
    data = {}
    for step in range(2):
        reps = []
        for r in range(2):
            p = rng.random((2, 64, 3), dtype=np.float32)
            l = rng.integers(0, 2, (2, 64)).astype(np.int32)
            data["points_s%d_r%d" % (step, r)] = p
            data["labels_s%d_r%d" % (step, r)] = l
            G, _, _, _ = O.train_step_grads(p.astype(np.float64), l, flags, params)
            reps.append(G)
        for n in acc:
            acc[n] += (reps[0][n] + reps[1][n]) / 2          # trainval.py:64-73 mean, :79 sum
    new = {n: v.copy() for n, v in params.items()}
    for n in new:
        O.adam_step(new[n], acc[n], np.zeros_like(new[n]), np.zeros_like(new[n]), 1, 1e-3)
    arrays = {"param:" + n: v.astype(np.float32) for n, v in params.items()}
    arrays.update({"accum:" + n: v.astype(np.float32) for n, v in acc.items()})
    arrays.update({"new:" + n: v.astype(np.float32) for n, v in new.items()})
    save("g9_two_replicas_adam", **data, **arrays)


if __name__ == "__main__":
    main()
