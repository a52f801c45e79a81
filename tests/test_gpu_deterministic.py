"""Deterministic mode (dgcnn._engine.DETERMINISTIC / DGCNN_FLAGS(DETERMINISTIC=True) / DGCNN_DETERMINISTIC=1): every sum has a fixed
order inside a workgroup, every statistics slot a single writer workgroup (dgcnn_set_stat_slots), the slots are added in a fixed
order and the transposed adjacency is sorted, so two runs of the same schedule are BIT-identical -- logits, dynamic graphs,
gradients and parameters after several optimizer steps -- where the default configuration (several writers per slot) agrees to
~1e-7 only.  Since round 3 the mode runs the default kernels (no materialised conv0 output, no extra passes)."""
import numpy as np
import pytest
import torch

import dgcnn
from dgcnn import _engine as E, _hip as H
from gpu_helpers import dev, host, capture_layers

pytestmark = pytest.mark.gpu


@pytest.fixture()
def det():
    old = E.DETERMINISTIC
    E.DETERMINISTIC = True
    yield
    E.DETERMINISTIC = old
    dgcnn.reset()


def _flags(**kw):
    base = dict(MODEL_NAME="residual-dgcnn", EDGE_CONV_LAYERS=3, EDGE_CONV_FILTERS=64, KVALUE=16, FC_LAYERS=2, FC_FILTERS=[128, 64],
                NUM_CLASS=3, NUM_CHANNEL=3, TRAIN=True, SEED=5, LEARNING_RATE=1e-3,
                DETERMINISTIC=bool(E.DETERMINISTIC))      # initialize() resolves the mode from the flag on every instance
    base.update(kw)
    return dgcnn.DGCNN_FLAGS(**base)


def _run(steps, pts, lab, graphs=False):
    tv = dgcnn.trainval(_flags()).initialize().use_graph(graphs)
    c = dgcnn.ctx()
    grads, idxs = [], []
    for s in range(steps):
        tv.zero_gradients(None)
        with capture_layers(keep_inputs=False) as cap:
            tv.accum_gradient(None, [pts[s]], [lab[s]])
        idxs.append([cap.layers["EdgeConv%d" % i][1] for i in range(3)])
        grads.append(c.flat_grad.clone())
        tv.apply_gradient(None)
    return c.flat_param.clone(), grads, idxs


def test_two_runs_are_bit_identical(det):
    rng = np.random.default_rng(0)
    pts = torch.from_numpy(rng.random((4, 6, 1024, 3), dtype=np.float32)).cuda()       # 4 steps of 6 clouds x 1024 points
    lab = torch.from_numpy(rng.integers(0, 3, (4, 6, 1024)).astype(np.int32)).cuda()
    p1, g1, i1 = _run(4, pts, lab)
    p2, g2, i2 = _run(4, pts, lab)
    for a, b in zip(g1, g2):
        assert torch.equal(a, b)
    for sa, sb in zip(i1, i2):
        for a, b in zip(sa, sb):
            np.testing.assert_array_equal(a, b)                   # the dynamic graphs themselves are reproduced
    assert torch.equal(p1, p2)
    assert float((p1 - dgcnn.trainval(_flags()).initialize()._ctx.flat_param).abs().max()) > 1e-3


def test_deterministic_mode_agrees_with_default_kernels():
    """Same math: one step in both modes agrees to the default mode's own run-to-run noise (forward bitwise-close, gradients
    to ~1e-5 of their scale)."""
    rng = np.random.default_rng(1)
    pts = torch.from_numpy(rng.random((1, 4, 512, 3), dtype=np.float32)).cuda()
    lab = torch.from_numpy(rng.integers(0, 3, (1, 4, 512)).astype(np.int32)).cuda()
    keep, E.DROPOUT_KEEP = E.DROPOUT_KEEP, 1.0
    old = E.DETERMINISTIC
    try:
        E.DETERMINISTIC = False
        _, g0, i0 = _run(1, pts, lab)
        E.DETERMINISTIC = True
        _, g1, i1 = _run(1, pts, lab)
    finally:
        E.DETERMINISTIC, E.DROPOUT_KEEP = old, keep
        dgcnn.reset()
    np.testing.assert_array_equal(i0[0][0], i1[0][0])             # layer 0: identical inputs
    a, b = g0[0].cpu().numpy(), g1[0].cpu().numpy()
    assert np.linalg.norm(a - b) <= 2e-3 * np.linalg.norm(a), np.linalg.norm(a - b) / np.linalg.norm(a)


def test_det_kernels_against_numpy(det):
    rng = np.random.default_rng(2)
    R, F = 5000, 70
    wide = rng.normal(size=(R, 100)).astype(np.float32)
    T = dev(wide)[:, 10:10 + F]
    st = torch.zeros(H.STAT_SLOTS * 2 * F, dtype=torch.float64, device="cuda")
    E.colstats_det(T, st)
    s = host(st).reshape(H.STAT_SLOTS, 2, F)
    assert np.abs(s[1:]).max() == 0
    np.testing.assert_allclose(s[0, 0], wide[:, 10:10 + F].astype(np.float64).sum(0), rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(s[0, 1], (wide[:, 10:10 + F].astype(np.float64) ** 2).sum(0), rtol=1e-12, atol=1e-9)
    # sorted adjacency
    B, N, k = 3, 200, 7
    idx = dev(rng.integers(0, N, (B, N, k)).astype(np.int32))
    Rn = B * N
    cws = torch.zeros(2 * Rn, dtype=torch.int32, device="cuda")
    off = torch.empty(Rn + 1, dtype=torch.int32, device="cuda")
    rev = torch.empty(Rn * k, dtype=torch.int32, device="cuda")
    H.call("dgcnn_edge_csr_build", idx.data_ptr(), B, N, k, cws.data_ptr(), off.data_ptr(), rev.data_ptr())
    srt = torch.full_like(rev, -1)
    H.call("dgcnn_edge_csr_sort", idx.data_ptr(), B, N, k, off.data_ptr(), rev.data_ptr(), srt.data_ptr())
    o, r, r0 = host(off), host(srt), host(rev)
    hidx = host(idx).reshape(-1)
    for j in range(Rn):
        seg = r[o[j]:o[j + 1]]
        assert (np.diff(seg) > 0).all()
        np.testing.assert_array_equal(seg, np.sort(r0[o[j]:o[j + 1]]))          # the bucket's own entries, reordered
        assert ((seg // (N * k)) * N + hidx[seg] == j).all()                      # every edge in it points at j
    rev = srt


def test_every_trainval_resolves_the_mode_for_itself():
    """An instance created with the default flag (None) must not inherit deterministic mode from an earlier instance."""
    old = E.DETERMINISTIC
    try:
        dgcnn.trainval(_flags(DETERMINISTIC=True)).initialize()
        assert E.DETERMINISTIC is True
        dgcnn.trainval(_flags(DETERMINISTIC=False)).initialize()
        assert E.DETERMINISTIC is False
        dgcnn.trainval(_flags(DETERMINISTIC=None)).initialize()
        assert E.DETERMINISTIC == E.DETERMINISTIC_ENV_DEFAULT and E.DETERMINISTIC_ENV_DEFAULT is True    # the default since round 5
        # a model the deterministic kernels do not take (an EdgeConv filter count that is not a multiple of 4): the DEFAULT falls
        # back to the atomics mode, an explicit request is refused
        dgcnn.trainval(_flags(DETERMINISTIC=None, EDGE_CONV_FILTERS=[30, 64, 64])).initialize()
        assert E.DETERMINISTIC is False
        with pytest.raises(ValueError):
            dgcnn.trainval(_flags(DETERMINISTIC=True, EDGE_CONV_FILTERS=[30, 64, 64])).initialize()
        dgcnn.trainval(_flags(DETERMINISTIC=None, HEAD_PLANES="f16")).initialize()      # the opt-in plane GEMMs sum with atomics
        assert E.DETERMINISTIC is False
    finally:
        E.HEAD_PLANES = E.HEAD_PLANES_ENV_DEFAULT
        E.DETERMINISTIC = old
        dgcnn.reset()


def test_deterministic_mode_runs_the_default_kernels(det):
    """No pass of its own on the model path any more: conv0's output stays virtual, the BatchNorm sums come from the GEMM epilogues /
    the statistics pass into single-writer slots; only the class dimension (F = 3 here) takes the fixed-order twin of det.hip."""
    rng = np.random.default_rng(2)
    pts = torch.from_numpy(rng.random((1, 4, 512, 3), dtype=np.float32)).cuda()
    lab = torch.from_numpy(rng.integers(0, 3, (1, 4, 512)).astype(np.int32)).cuda()
    calls = []
    orig = H.call

    def spy(name, *a, **kw):
        calls.append(name)
        return orig(name, *a, **kw)
    E.H.call = spy
    try:
        _run(1, pts, lab)
    finally:
        E.H.call = orig
    assert H.STAT_SLOTS >= 256
    assert "dgcnn_colstats_det_f32" not in calls
    assert calls.count("dgcnn_bn_bwd_reduce_det_f32") == 1                 # the Final layer (3 classes)
    assert "dgcnn_edge_bn_act_kreduce_f32" in calls and "dgcnn_edge_csr_sort" in calls
    dgcnn.reset()
    E.DETERMINISTIC = False
    dgcnn.reset()
    assert H.STAT_SLOTS == 32


def test_two_full_size_steps_are_bit_identical(det):
    """BASELINE configs[1] (24 clouds x 2048 points, k = 20, filters 64/64/128, FC 512/256): the producer with the most workgroups
    is a GEMM over 49152 rows in 64-row tiles -- 768 writers, 768 slots; the statistics pass and the class-dimension kernel cap
    their grids at that.  Two runs of the same two steps: identical gradients and parameters, bit for bit."""
    rng = np.random.default_rng(7)
    pts = torch.from_numpy(rng.random((2, 24, 2048, 3), dtype=np.float32)).cuda()
    lab = torch.from_numpy(rng.integers(0, 2, (2, 24, 2048)).astype(np.int32)).cuda()

    def run():
        f = dgcnn.DGCNN_FLAGS(MODEL_NAME="dgcnn", EDGE_CONV_LAYERS=3, EDGE_CONV_FILTERS=[64, 64, 128], KVALUE=20, FC_LAYERS=2,
                              FC_FILTERS=[512, 256], NUM_CLASS=2, NUM_CHANNEL=3, TRAIN=True, SEED=3, LEARNING_RATE=1e-3,
                              DETERMINISTIC=True)
        tv = dgcnn.trainval(f).initialize()
        c = dgcnn.ctx()
        grads = []
        for s in range(2):
            tv.zero_gradients(None)
            tv.accum_gradient(None, [pts[s]], [lab[s]])
            grads.append(c.flat_grad.clone())
            tv.apply_gradient(None)
        return c.flat_param.clone(), grads, H.STAT_SLOTS
    p1, g1, slots = run()
    p2, g2, _ = run()
    assert slots == 768
    for a, b in zip(g1, g2):
        assert torch.equal(a, b)
    assert torch.equal(p1, p2)


def test_odd_filter_counts_are_refused_in_deterministic_mode(det):
    """Filter counts that are not multiples of 4 (6, 10) take conv0's edge-level form, whose neighbour gradient is an atomic
    scatter: the mode says so at initialize() instead of returning sums that depend on the order of the atomics."""
    f = dgcnn.DGCNN_FLAGS(MODEL_NAME="dgcnn", EDGE_CONV_LAYERS=2, EDGE_CONV_FILTERS=[6, 10], KVALUE=3, FC_LAYERS=2,
                          FC_FILTERS=[512, 256], NUM_CLASS=2, NUM_CHANNEL=3, TRAIN=True, SEED=2, DETERMINISTIC=True)
    with pytest.raises(ValueError, match="multiples of 4"):
        dgcnn.trainval(f).initialize()
    f.EDGE_CONV_FILTERS = [8, 12]                      # multiples of 4 that are not powers of two: fine, and repeatable
    rng = np.random.default_rng(5)
    pts = torch.from_numpy(rng.random((2, 96, 3), dtype=np.float32)).cuda()
    lab = torch.from_numpy(rng.integers(0, 2, (2, 96)).astype(np.int32)).cuda()

    def run():
        tv = dgcnn.trainval(f).initialize()
        tv.zero_gradients(None)
        tv.accum_gradient(None, [pts], [lab])
        return dgcnn.ctx().flat_grad.clone()
    a, b = run(), run()
    assert torch.equal(a, b) and float(a.abs().sum()) > 0
