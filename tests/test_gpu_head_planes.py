"""Plane mode of the head (dgcnn._engine.HEAD_PLANES; csrc/gemm_pl.hip + planes_bn.hip): MergedEdgeConv, FC0 and FC1 read their
GEMM operands as pre-split 16-bit planes written by the producing BatchNorm pass.  Same model, same tolerances as the default
path: logits within 1e-3 and every gradient within 1e-2 (relative Frobenius) of the float64 twin fed the same neighbour graphs."""
import numpy as np
import pytest
import torch

import dgcnn
from dgcnn import _engine as E, _planes as P
from oracle import dgcnn_oracle as O
from gpu_helpers import host, run_model

pytestmark = pytest.mark.gpu

B, N, C = 4, 2048, 3


def _flags(train, mode=None):
    hp = {None: None, P.F16X2: "f16"}[mode]
    return dgcnn.DGCNN_FLAGS(MODEL_NAME="dgcnn", EDGE_CONV_LAYERS=2, EDGE_CONV_FILTERS=[64, 64], FC_LAYERS=2, FC_FILTERS=[512, 256],
                             NUM_CLASS=3, KVALUE=12, NUM_CHANNEL=C, TRAIN=train, SEED=3, HEAD_PLANES=hp)


@pytest.fixture(scope="module")
def case():
    rng = np.random.default_rng(21)
    pts = rng.random((B, N, C), dtype=np.float32)
    labels = rng.integers(0, 3, (B, N)).astype(np.int32)
    params = O.init_params(_flags(True), C, seed=4)
    for n in params:
        if n.endswith("beta"):
            params[n] = rng.normal(0, 0.2, params[n].shape).astype(np.float32)
    return pts, labels, params


def _run(mode, case, train):
    pts, labels, params = case
    old, keep = E.HEAD_PLANES, E.DROPOUT_KEEP
    E.HEAD_PLANES, E.DROPOUT_KEEP = mode, 1.0
    used = []
    orig = P.gemm

    def spy(form, A, Bm, Cm, **kw):
        used.append((form, A.fmt, tuple(Cm.shape)))
        return orig(form, A, Bm, Cm, **kw)
    P.gemm = spy
    try:
        tv, res, layers = run_model(dgcnn, _flags(train, mode), pts, params, train=train, labels=labels)
        grads = {n: host(tv.gradients[n]).astype(np.float64) for n in params} if train else None
        out = [host(r) if isinstance(r, torch.Tensor) else r for r in res]
    finally:
        E.HEAD_PLANES, E.DROPOUT_KEEP = old, keep
        P.gemm = orig
        dgcnn.reset()
    return out, grads, [layers["EdgeConv%d" % i][1] for i in range(2)], used


@pytest.mark.parametrize("mode", [P.F16X2])
def test_plane_mode_training_step_matches_the_float64_twin(case, mode):
    pts, labels, params = case
    res, grads, idx_list, used = _run(mode, case, train=True)
    # the nine head products ran on the plane GEMM: 3 forward + 3 dgrad (KC) + 3 wgrad (TR)
    assert sum(1 for u in used if u[0] == P.KC) == 6 and sum(1 for u in used if u[0] == P.TR) == 3, used
    assert all(u[1] == mode for u in used)
    p64 = {n: v.astype(np.float64) for n, v in params.items()}
    G64, loss64, _, _ = O.train_step_grads(pts.astype(np.float64), labels, _flags(True), p64, idx_list=idx_list)
    assert abs(float(res[2]) - float(loss64)) < 1e-4, (float(res[2]), float(loss64))
    rel = {n: np.linalg.norm(grads[n] - G64[n]) / max(np.linalg.norm(G64[n]), 1e-30) for n in params}
    worst = max(rel, key=rel.get)
    print("plane mode %d: loss %.7f vs twin %.7f; gradients rel. Frobenius worst %.2e (%s)" % (mode, float(res[2]), float(loss64), rel[worst], worst))
    assert rel[worst] < 1e-2, rel


@pytest.mark.parametrize("mode", [P.F16X2])
def test_plane_mode_inference_logits(case, mode):
    pts, labels, params = case
    res, _, idx_list, used = _run(mode, case, train=False)
    assert len(used) == 3 and all(u[0] == P.KC for u in used)
    f = _flags(False)
    p64 = {n: v.astype(np.float64) for n, v in params.items()}
    ref, _ = O.model_forward(pts.astype(np.float64), f, p64, idx_list=idx_list)
    sm = res[0]                                                    # softmax (B, N, ncls)
    e = np.exp(ref - ref.max(-1, keepdims=True))
    np.testing.assert_allclose(sm, e / e.sum(-1, keepdims=True), rtol=0, atol=1e-3)


def test_plane_mode_is_off_in_deterministic_mode_and_for_small_problems(case):
    old = E.HEAD_PLANES
    try:
        dgcnn.trainval(_flags(True, P.F16X2)).initialize()
        assert E.HEAD_PLANES == P.F16X2
        assert E.planes_ok(8192, 128, 1024) and not E.planes_ok(4096, 128, 1024) and not E.planes_ok(8192, 100, 1024)
        E.DETERMINISTIC = True
        assert not E.planes_ok(8192, 128, 1024)
    finally:
        E.HEAD_PLANES, E.DETERMINISTIC = old, E.DETERMINISTIC_ENV_DEFAULT
        dgcnn.reset()


def test_plane_mode_on_the_residual_architecture(case):
    """configs[2]'s architecture (residual-dgcnn, 6 x 64, k = 40) at B=4, N=2048 in fp16 plane mode: the activation scale must
    cover residual sums of two batch-normalised tensors (factor 2 in dgcnn_param_scales_f32); logits within 1e-3 of the twin."""
    pts, labels, _ = case
    f = dgcnn.DGCNN_FLAGS(MODEL_NAME="residual-dgcnn", EDGE_CONV_LAYERS=6, EDGE_CONV_FILTERS=64, FC_LAYERS=2, FC_FILTERS=[512, 256],
                          NUM_CLASS=2, KVALUE=40, NUM_CHANNEL=C, TRAIN=False, SEED=3, HEAD_PLANES="f16")
    rng = np.random.default_rng(8)
    params = O.init_params(f, C, seed=6)
    for n in params:
        if n.endswith("beta"):
            params[n] = rng.normal(0, 0.2, params[n].shape).astype(np.float32)
    used = []
    orig = P.gemm

    def spy(form, A, Bm, Cm, **kw):
        used.append(form)
        return orig(form, A, Bm, Cm, **kw)
    P.gemm = spy
    try:
        tv, res, layers = run_model(dgcnn, f, pts, params, train=False, labels=labels)
    finally:
        P.gemm = orig
        E.HEAD_PLANES = E.HEAD_PLANES_ENV_DEFAULT
        dgcnn.reset()
    assert used == [P.KC] * 3
    idx_list = [layers["EdgeConv%d" % i][1] for i in range(6)]
    p64 = {n: v.astype(np.float64) for n, v in params.items()}
    ref, _ = O.model_forward(pts.astype(np.float64), f, p64, idx_list=idx_list)
    e = np.exp(ref - ref.max(-1, keepdims=True))
    np.testing.assert_allclose(host(res[0]), e / e.sum(-1, keepdims=True), rtol=0, atol=1e-3)
