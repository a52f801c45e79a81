"""GPU parity tests (run with -m gpu on an MI355X): every HIP entry point, called through the
C ABI via the dgcnn package, against the CPU oracle on identical seeded inputs.

Bars (BASELINE.json north_star): k-NN indices bit-exact; floating point within the tolerance
written at each assert (logits 1e-3 absolute, fp32)."""
import numpy as np
import pytest
import torch

from oracle import dgcnn_oracle as O

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


@pytest.fixture()
def dg():
    import dgcnn
    dgcnn.reset()
    return dgcnn


@pytest.fixture(params=[0, 6, 9], ids=["f32mfma", "bf16x6", "bf16x9"])
def arith(request):
    """GEMM arithmetic under test; the library default is restored afterwards."""
    from dgcnn import _hip as H
    old = H.gemm_arith()
    H.set_gemm_arith(request.param)
    yield request.param
    H.set_gemm_arith(old)


def _assert_idx_equal(idx_gpu, idx_ref, what):
    a, b = host(idx_gpu), idx_ref
    if not np.array_equal(a, b):
        bad = np.argwhere((a != b).any(-1))
        raise AssertionError("%s: %d / %d rows differ, first %s: gpu %s ref %s" % (
            what, len(bad), a.shape[0] * a.shape[1], bad[0], a[tuple(bad[0])], b[tuple(bad[0])]))


# ------------------------------------------------------------------------------------------
# K1 k_nn: bit-exact indices
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,N,C,k,kind", [
    (2, 64, 3, 5, "uniform"),          # golden G1 shape
    (2, 100, 3, 7, "integer"),         # ties + duplicates, N not a multiple of 64 (G2)
    (2, 256, 64, 20, "relu"),          # feature-space graph, many zeros (G3)
    (1, 130, 4, 8, "uniform"),         # C = 4 inputs (iotool.py:82), ragged N
    (2, 512, 3, 10, "uniform"),        # BASELINE config 1
    (1, 300, 64, 40, "relu"),          # production k = 40 (scripts/lsf/train_dgcnn.sh:8)
    (1, 64, 3, 64, "integer"),         # k == N: everything selected
    (1, 200, 20, 33, "uniform"),       # odd C / k, exercises padding and the 64-slot list
    (3, 2048, 3, 20, "uniform"),       # headline N, k
    (2, 2048, 64, 20, "relu"),
])
def test_knn_bit_exact(dg, B, N, C, k, kind):
    rng = np.random.default_rng(B * 1000 + N + C + k)
    if kind == "uniform":
        pts = rng.random((B, N, C), dtype=np.float32)
    elif kind == "integer":
        pts = rng.integers(0, 6, (B, N, C)).astype(np.float32)
    else:
        pts = np.maximum(rng.normal(0, 1, (B, N, C)), 0).astype(np.float32)
    ref = O.k_nn(pts, k)
    idx = dg.ops.k_nn(dev(pts), k)
    assert idx.dtype == torch.int32 and tuple(idx.shape) == (B, N, k)
    _assert_idx_equal(idx, ref, "k_nn %s" % ((B, N, C, k, kind),))


def test_knn_strided_view_and_errors(dg):
    rng = np.random.default_rng(5)
    wide = rng.random((2, 128, 200), dtype=np.float32)
    t = dev(wide)[:, :, 72:136]                       # 64 channels inside a wider buffer (ld = 200)
    ref = O.k_nn(np.ascontiguousarray(wide[:, :, 72:136]), 9)
    _assert_idx_equal(dg.ops.k_nn(t, 9), ref, "strided")
    with pytest.raises(ValueError):
        dg.ops.k_nn(dev(wide[:, :4, :3]), 5)           # N < k  (tf.nn.top_k raises)
    with pytest.raises(Exception):
        dg.ops.k_nn(torch.zeros(1, 8, 3), 2)           # CPU tensor: no fallback


def test_knn_mfma_equals_valu_at_large_n(dg):
    """Feature-space graph at N=8192, k=40 (production k, scripts/lsf/train_dgcnn.sh:8): the MFMA distance
    kernel must reproduce the VALU fmaf-chain kernel (exact by construction) bit for bit; the raw-point
    graph (C=3) at the same N is compared with the C oracle."""
    from dgcnn import _hip as H
    rng = np.random.default_rng(8)
    feats = np.maximum(rng.normal(0, 1, (2, 8192, 64)), 0).astype(np.float32)
    t = dev(feats)
    a = dg.ops.k_nn(t, 40)
    prev = H.load().dgcnn_knn_force_valu(1)
    try:
        b = dg.ops.k_nn(t, 40)
    finally:
        H.load().dgcnn_knn_force_valu(prev)
    assert torch.equal(a, b)
    ia = host(a)
    assert ((ia == np.arange(8192)[None, :, None]).sum(-1) == 1).all()          # self among the k
    pts = rng.random((1, 8192, 3), dtype=np.float32)
    np.testing.assert_array_equal(host(dg.ops.k_nn(dev(pts), 40)), O.k_nn(pts, 40))


@pytest.mark.parametrize("B,N,C,k,kind", [
    (2, 256, 64, 20, "relu"), (1, 300, 64, 40, "relu"), (2, 2048, 64, 20, "relu"), (1, 1000, 48, 8, "uniform"),
    (2, 200, 20, 33, "uniform"), (1, 130, 64, 64, "integer"), (1, 4096, 32, 20, "zeros"), (3, 700, 64, 20, "dupes"),
])
def test_knn_bf16_filter_kernel_bit_exact(dg, B, N, C, k, kind):
    """The large-N kernel (approximate distances from two bf16 terms on the bf16 matrix pipe as a conservative filter, the
    normative fp32 fmaf chain for the survivors), forced on at small N: the same indices as the C oracle, including exact
    ties / duplicate points (integer data), all-zero rows, C < 64 and C % 4 == 0 paddings, ragged N, every list size."""
    from dgcnn import _hip as H
    rng = np.random.default_rng(B * 977 + N + C + k)
    if kind == "uniform":
        pts = rng.random((B, N, C), dtype=np.float32)
    elif kind == "integer":
        pts = rng.integers(0, 4, (B, N, C)).astype(np.float32)
    elif kind == "zeros":
        pts = np.maximum(rng.normal(0, 1, (B, N, C)), 0).astype(np.float32)
        pts[:, ::7] = 0.0                                              # many identical all-zero rows
    elif kind == "dupes":
        pts = np.maximum(rng.normal(0, 1, (B, N, C)), 0).astype(np.float32)
        pts[:, 100:200] = pts[:, 0:100]                                # exact duplicates: ties at d = 0 decided by index
    else:
        pts = np.maximum(rng.normal(0, 1, (B, N, C)), 0).astype(np.float32)
    pts = (pts * np.exp(rng.normal(0, 2, (B, N, 1)))).astype(np.float32) if kind == "relu" else pts     # wide range of norms
    ref = O.k_nn(pts, k)
    prev = H.load().dgcnn_knn_bf16_filter(1)
    try:
        idx = dg.ops.k_nn(dev(pts), k)
    finally:
        H.load().dgcnn_knn_bf16_filter(prev)
    _assert_idx_equal(idx, ref, "k_nn bf16-filter kernel %s" % ((B, N, C, k, kind),))
    assert torch.equal(idx, dg.ops.k_nn(dev(pts), k))                   # == the default kernel at this size


def test_knn_headline_shape_properties(dg):
    """(24,2048,3) k=20: full bit-exact compare + size-independent properties."""
    rng = np.random.default_rng(0)
    pts = rng.random((24, 2048, 3), dtype=np.float32)
    idx = host(dg.ops.k_nn(dev(pts), 20))
    assert idx.min() >= 0 and idx.max() < 2048
    assert ((idx == np.arange(2048)[None, :, None]).sum(-1) == 1).all()     # self is always among the k (ops.py:18: no exclusion)
    for b in (0, 11, 23):
        D = O.dist_matrix_f32(pts[b])
        dsel = np.take_along_axis(D, idx[b].astype(np.int64), 1)
        assert (np.diff(dsel, axis=1) >= 0).all()                    # ascending
        kth = dsel[:, -1:]
        assert ((D < kth).sum(1) <= 19).all()                        # nothing closer was left out
    np.testing.assert_array_equal(idx, O.k_nn(pts, 20))


# ------------------------------------------------------------------------------------------
# K2 edges
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,N,C,k", [(2, 64, 3, 5), (1, 96, 64, 20)])
def test_edges_exact(dg, B, N, C, k):
    rng = np.random.default_rng(1)
    pts = rng.random((B, N, C), dtype=np.float32)
    E = host(dg.ops.edges(dev(pts), k))
    np.testing.assert_array_equal(E, O.edges(pts, k))                # gather + one subtract: exact


# ------------------------------------------------------------------------------------------
# GEMM family
# ------------------------------------------------------------------------------------------
def _gemm(dg, A, B, C, **kw):
    from dgcnn import _engine as E
    E.gemm(A, B, C, **kw)


@pytest.mark.parametrize("M,N,K", [(128, 64, 16), (300, 70, 50), (1024, 512, 192), (257, 2, 256), (640, 128, 6),
                                   (129, 130, 131),
                                   (8200, 64, 128), (9000, 256, 64), (8192, 100, 256)])   # 64-row tiles (short reduction, many rows)
def test_gemm_nn_nt(dg, arith, M, N, K):
    rng = np.random.default_rng(M + N + K)
    A = rng.normal(size=(M, K)).astype(np.float32)
    B = rng.normal(size=(K, N)).astype(np.float32)
    ref = A.astype(np.float64) @ B.astype(np.float64)
    tol = 2e-6 * K ** 0.5 * 4 + 1e-5
    C = torch.empty((M, N), device="cuda")
    _gemm(dg, dev(A), dev(B), C)
    np.testing.assert_allclose(host(C), ref, rtol=1e-5, atol=tol)
    # NT: B given as [N][K]
    C2 = torch.empty((M, N), device="cuda")
    _gemm(dg, dev(A), dev(B.T.copy()), C2, transB=True)
    np.testing.assert_allclose(host(C2), ref, rtol=1e-5, atol=tol)
    # beta = 1 accumulate
    C3 = dev(np.ones((M, N), np.float32))
    _gemm(dg, dev(A), dev(B), C3, beta=1.0)
    np.testing.assert_allclose(host(C3), ref + 1, rtol=1e-5, atol=tol)


@pytest.mark.parametrize("M,N,K", [(192, 1024, 3000), (128, 64, 8192), (6, 64, 5000), (1728, 512, 4096), (70, 3, 1000)])
def test_gemm_tn_splitk(dg, arith, M, N, K):
    rng = np.random.default_rng(M + N + K)
    X = rng.normal(size=(K, M)).astype(np.float32)      # stored [K][M]
    dY = rng.normal(size=(K, N)).astype(np.float32)
    ref = X.astype(np.float64).T @ dY.astype(np.float64)
    C = dev(np.full((M, N), 2.0, np.float32))
    _gemm(dg, dev(X), dev(dY), C, transA=True, beta=1.0)
    np.testing.assert_allclose(host(C), ref + 2, rtol=1e-4, atol=1e-5 * K ** 0.5 * 4)


@pytest.mark.parametrize("M,N,K", [(1000, 2, 256), (49152, 2, 256), (130, 4, 64), (77, 1, 1024), (512, 3, 8)])
def test_gemm_skinny_class_dimension(dg, M, N, K):
    """The Final layer's products (N = NUM_CLASS) take dedicated streaming kernels: NN with BN sums and beta,
    the weight gradient (TN, reduction over the rows) and the data gradient (NT, reduction length N)."""
    from dgcnn import _engine as E
    rng = np.random.default_rng(M + N + K)
    A = rng.normal(size=(M, K)).astype(np.float32)
    W = rng.normal(size=(K, N)).astype(np.float32)
    ref = A.astype(np.float64) @ W
    tol = dict(rtol=1e-5, atol=2e-6 * K ** 0.5 * 4 + 1e-5)
    C = dev(np.full((M, N), 0.5, np.float32))
    st = torch.zeros(E.H.STAT_SLOTS * 2 * N, dtype=torch.float64, device="cuda")
    E.gemm(dev(A), dev(W), C, beta=2.0, stats=st)
    np.testing.assert_allclose(host(C), ref + 1.0, **tol)
    s = host(st).reshape(E.H.STAT_SLOTS, 2, N).sum(0)
    np.testing.assert_allclose(s[0], (ref + 1.0).sum(0), rtol=1e-5, atol=1e-2)
    np.testing.assert_allclose(s[1], ((ref + 1.0) ** 2).sum(0), rtol=1e-5, atol=1e-2)
    # weight gradient: dW (K x N) += A^T dT
    dT = rng.normal(size=(M, N)).astype(np.float32)
    dW = dev(np.ones((K, N), np.float32))
    E.gemm(dev(A), dev(dT), dW, transA=True, beta=1.0)
    np.testing.assert_allclose(host(dW), A.astype(np.float64).T @ dT + 1.0, rtol=1e-4, atol=1e-5 * M ** 0.5 * 4)
    # data gradient: dA (M x K) = dT W^T  (+ beta dA)
    dA = dev(np.full((M, K), 3.0, np.float32))
    E.gemm(dev(dT), dev(W), dA, transB=True, beta=1.0)
    np.testing.assert_allclose(host(dA), dT.astype(np.float64) @ W.T + 3.0, rtol=1e-5, atol=1e-5)
    dA2 = torch.empty((M, K), device="cuda")
    E.gemm(dev(dT), dev(W), dA2, transB=True)
    np.testing.assert_allclose(host(dA2), dT.astype(np.float64) @ W.T, rtol=1e-5, atol=1e-5)


def test_gemm_split_accuracy(dg):
    """The bf16-split kernels (gemm_x3.hip) against the native fp32-MFMA kernel, both measured against an
    fp64 product: error relative to sum_k |a_k b_k| (the natural fp32 scale of a dot product).  All nine
    Measured (profiles/r01_gemm_arith.txt): both splits are slightly MORE accurate than the native fmaf chain (the
    MFMA sums 16 exact products per instruction before rounding into the accumulator); the asserts allow 10 % rms slack."""
    from dgcnn import _hip as H
    rng = np.random.default_rng(5)
    old = H.gemm_arith()
    rows = []
    try:
        for (M, N, K, ta, tb) in [(512, 256, 1728, 0, 0), (384, 512, 512, 0, 1), (256, 128, 16384, 1, 0), (640, 64, 64, 0, 0)]:
            # activations-like (non-negative, wide dynamic range) times weights-like (zero mean)
            A = (np.abs(rng.normal(size=(M, K))) * np.exp(rng.normal(0, 1.5, size=(M, K)))).astype(np.float32)
            B = rng.normal(0, 0.05, size=(K, N)).astype(np.float32)
            ref = A.astype(np.float64) @ B.astype(np.float64)
            scale = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64)
            dA = dev(A.T.copy()) if ta else dev(A)
            dB = dev(B.T.copy()) if tb else dev(B)
            err = {}
            for mode in (0, 6, 9):
                H.set_gemm_arith(mode)
                C = torch.empty((M, N), device="cuda")
                _gemm(dg, dA, dB, C, transA=bool(ta), transB=bool(tb))
                e = np.abs(host(C) - ref) / scale
                err[mode] = (float(e.max()), float(np.sqrt((e ** 2).mean())))
            rows.append(((M, N, K, ta, tb), err))
            print("M=%d N=%d K=%d ta=%d tb=%d  max/rms rel. err  f32mfma %.2e/%.2e  bf16x6 %.2e/%.2e  bf16x9 %.2e/%.2e"
                  % ((M, N, K, ta, tb) + err[0] + err[6] + err[9]))
            assert err[9][0] <= 1.5 * err[0][0] + 1e-9 and err[9][1] <= 1.1 * err[0][1] + 1e-10
            assert err[6][0] <= 1.5 * err[0][0] + 1e-9 and err[6][1] <= 1.1 * err[0][1] + 1e-10
            assert err[6][0] < 2e-6                                  # fp32 class in absolute terms
    finally:
        H.set_gemm_arith(old)


def test_gemm_strided_stats_and_group_bias(dg, arith):
    from dgcnn import _engine as E
    rng = np.random.default_rng(7)
    R, Cin, F, G = 384, 96, 64, 3
    wide = rng.normal(size=(R, 300)).astype(np.float32)
    A = dev(wide)[:, 100:100 + Cin]
    W = rng.normal(size=(Cin, F)).astype(np.float32)
    gb = rng.normal(size=(G, F)).astype(np.float32)
    outw = torch.zeros((R, 200), device="cuda")
    C = outw[:, 40:40 + F]
    st = torch.zeros(E.H.STAT_SLOTS * 2 * F, dtype=torch.float64, device="cuda")
    E.gemm(A, dev(W), C, gbias=dev(gb), rpg=R // G, stats=st)
    ref = wide[:, 100:100 + Cin].astype(np.float64) @ W + np.repeat(gb, R // G, 0)
    np.testing.assert_allclose(host(C), ref, rtol=1e-5, atol=1e-4)
    assert host(outw)[:, :40].max() == 0 and host(outw)[:, 40 + F:].max() == 0     # nothing outside the slice
    s = host(st).reshape(E.H.STAT_SLOTS, 2, F).sum(0)
    np.testing.assert_allclose(s[0], ref.sum(0), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(s[1], (ref ** 2).sum(0), rtol=1e-5, atol=1e-3)
    mean, rstd = E.bn_finalize(st, F, R)
    np.testing.assert_allclose(host(mean), ref.mean(0), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(host(rstd), 1 / np.sqrt(ref.var(0) + 1e-3), rtol=1e-5)


# ------------------------------------------------------------------------------------------
# edge_conv block: forward + backward against the oracle
# ------------------------------------------------------------------------------------------
def _set_vars(dg, params):
    c = dg.ctx()
    for n, v in params.items():
        c.set_variable(n, v)


@pytest.mark.parametrize("B,N,C,k,F", [(2, 64, 3, 5, 8), (2, 128, 64, 20, 64), (1, 96, 4, 7, 128), (2, 64, 64, 10, 32)])
def test_edge_conv_forward_backward(dg, B, N, C, k, F):
    """max / mean / net and gradients w.r.t. X, W0, beta0, W1, beta1."""
    from dgcnn import _engine as E
    rng = np.random.default_rng(C * 10 + F)
    pts = rng.random((B, N, C), dtype=np.float32)
    P = {"conv0/weights": rng.normal(0, 0.5, (2 * C, F)).astype(np.float32),
         "conv0/BatchNorm/beta": rng.normal(0, 0.3, F).astype(np.float32),
         "conv1/weights": rng.normal(0, 0.3, (2 * F, 64)).astype(np.float32),
         "conv1/BatchNorm/beta": rng.normal(0, 0.3, 64).astype(np.float32)}
    c = dg.ctx()
    c.begin_step()
    c.recording = True
    x = c.new_buffer(B * N, C)                 # tracked, so that d(point_cloud) is produced
    x.copy_(dev(pts.reshape(B * N, C)))
    for n, v in P.items():                     # create the variables, then load the oracle's values
        c.get_variable(n, v.shape)
    _set_vars(dg, P)
    outs = dg.ops.edge_conv(x.view(B, N, C), k, F, True)
    idx = host(dg.ops.edge_conv.last_idx)
    np.testing.assert_array_equal(idx, O.k_nn(pts, k))

    ref, cache = O.edge_conv(pts.astype(np.float64), k, *[P[n].astype(np.float64) for n in P], idx=idx)
    for name, a, b in zip(("max", "mean", "net"), outs, ref):
        assert tuple(a.shape) == b.shape                      # (B,N,1,ch): ops.py:73
        np.testing.assert_allclose(host(a), b, rtol=1e-4, atol=1e-4, err_msg=name)   # tol 1e-4 (fp32 vs fp64 twin)

    d = [rng.normal(size=r.shape) for r in ref]
    for t, g in zip(outs, d):
        v, _, _ = E.as2d(t)
        c.grad(v).copy_(dev(g.reshape(B * N, -1).astype(np.float32)))
    c.backward()
    dx_ref, g_ref = O.edge_conv_bwd(d[0], d[1], d[2], cache)
    scale = lambda r: 1e-3 * max(1.0, float(np.abs(r).max()))
    np.testing.assert_allclose(host(c.grad(x)).reshape(B, N, C), dx_ref, rtol=1e-3, atol=scale(dx_ref), err_msg="dx")
    for n, key in (("conv0/weights", "W0"), ("conv0/BatchNorm/beta", "beta0"), ("conv1/weights", "W1"),
                   ("conv1/BatchNorm/beta", "beta1")):
        np.testing.assert_allclose(host(c.var_grads[n]), g_ref[key], rtol=1e-3, atol=scale(g_ref[key]), err_msg=n)


@pytest.mark.parametrize("B,N,C,k,F", [(2, 192, 64, 12, 64), (2, 160, 3, 9, 48), (1, 256, 64, 20, 128)])
def test_edge_mlp_forms_agree(dg, B, N, C, k, F):
    """conv0 is issued as point-level GEMM [U|V] = X [Wa-Wb | Wb] and its (B*N*k, F) output is never written:
    the BatchNorm passes recompute V[neighbour] + U[point] (default).  Materialising it with the gather-add
    kernel, the edge-level factored GEMM and the literal (B*N*k) x 2C GEMM over E = [x_i, x_j - x_i] stay
    behind switches.  Same outputs and gradients from all four."""
    from dgcnn import _engine as E
    rng = np.random.default_rng(21)
    pts = rng.normal(size=(B, N, C)).astype(np.float32)
    W = {"conv0/weights": rng.normal(0, 0.3, (2 * C, F)).astype(np.float32), "conv0/BatchNorm/beta": rng.normal(0, 0.2, F).astype(np.float32),
         "conv1/weights": rng.normal(0, 0.3, (2 * F, 64)).astype(np.float32), "conv1/BatchNorm/beta": rng.normal(0, 0.2, 64).astype(np.float32)}
    res = {}
    for form in ("literal", "nbr_gemm", "gather_y", "gather"):
        dg.reset()
        c = dg.ctx()
        E.EDGE_MLP_LITERAL = form == "literal"
        E.EDGE_MLP_NBR_GEMM = form == "nbr_gemm"
        E.EDGE_MATERIALIZE_Y = form == "gather_y"
        try:
            c.begin_step()
            c.recording = True
            x = c.new_buffer(B * N, C)
            x.copy_(dev(pts.reshape(B * N, C)))
            for n, v in W.items():
                c.get_variable(n, v.shape)
            _set_vars(dg, W)
            outs = dg.ops.edge_conv(x.view(B, N, C), k, F, True)
            for t in outs:
                v, _, _ = E.as2d(t)
                c.grad(v).fill_(0.01)
            c.backward()
            res[form] = [host(t).copy() for t in outs] + [host(c.grad(x)).copy()] + [host(c.var_grads[n]).copy() for n in W]
        finally:
            E.EDGE_MLP_LITERAL = False
            E.EDGE_MLP_NBR_GEMM = False
            E.EDGE_MATERIALIZE_Y = False
    for a, b in zip(res["gather_y"][:3], res["gather"][:3]):      # same y = V + U, written or recomputed: same forward
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-5)       # (up to the fp64-atomic order of the BN sums)
    for form in ("nbr_gemm", "gather_y", "gather"):
        for a, b in zip(res["literal"], res[form]):
            np.testing.assert_allclose(a, b, rtol=2e-3, atol=2e-4 * max(1.0, float(np.abs(a).max())), err_msg=form)


@pytest.mark.parametrize("B,N,k,F", [(2, 100, 7, 64), (3, 64, 20, 128), (1, 130, 5, 48), (2, 50, 1, 8), (9, 40, 6, 16)])
def test_edge_bn_passes_equal_dense_passes(dg, B, N, k, F):
    """dgcnn_edge_bn_* recompute y = V[idx] + U instead of reading Y: against the dense kernels on the
    materialised Y the forward outputs and dY are bit-identical and the reductions agree to fp64 round-off."""
    from dgcnn import _hip as H
    rng = np.random.default_rng(B * 100 + F)
    R = B * N
    UV = dev(rng.normal(size=(R, 2 * F)).astype(np.float32))
    idx = dev(rng.integers(0, N, (B, N, k)).astype(np.int32))
    e = (UV[:, F:].data_ptr(), 2 * F, UV.data_ptr(), 2 * F, idx.data_ptr(), B, N, k, F)
    Y = torch.empty((R * k, F), device="cuda")
    H.call("dgcnn_edge_gather_add_f32", *e, Y.data_ptr(), 0)
    mean, rstd, beta = (dev(rng.normal(0, 0.3, F).astype(np.float32)), dev((0.5 + rng.random(F)).astype(np.float32)),
                        dev(rng.normal(0, 0.3, F).astype(np.float32)))
    outs = []
    for edge in (False, True):
        mm = torch.zeros((R, 2 * F + 4), device="cuda")                  # max | mean inside a wider buffer
        mx, mn = mm[:, :F], mm[:, F:2 * F]
        cnt = torch.zeros((R, F), device="cuda")
        if edge:
            H.call("dgcnn_edge_bn_act_kreduce_f32", *e, mean.data_ptr(), rstd.data_ptr(), beta.data_ptr(), 1,
                   mx.data_ptr(), H.ld2(mx), mn.data_ptr(), H.ld2(mn), cnt.data_ptr())
        else:
            H.call("dgcnn_bn_act_kreduce_f32", Y.data_ptr(), R, k, F, mean.data_ptr(), rstd.data_ptr(), beta.data_ptr(), 1,
                   mx.data_ptr(), H.ld2(mx), mn.data_ptr(), H.ld2(mn), 0, 0, cnt.data_ptr())
        outs.append((host(mm).copy(), host(cnt).copy()))
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    # the edge variant packs (#ties of the max) + 256 * (#rows with z > 0) into cnt_out
    packed = outs[1][1]
    npos, ties = np.floor(packed / 256), packed % 256
    np.testing.assert_array_equal(outs[0][1], ties)
    z = np.maximum((host(Y).reshape(R, k, F) - host(mean)) * host(rstd) + host(beta), 0)
    np.testing.assert_array_equal(npos, (z > 0).sum(1))
    assert outs[0][1].min() >= 1 and outs[0][1].max() <= k
    mx = dev(outs[0][0][:, :F].copy())
    mn = dev(outs[0][0][:, F:2 * F].copy())
    cnt = dev(outs[0][1])
    cnt_packed = dev(packed)
    dmx, dmn = dev(rng.normal(size=(R, F)).astype(np.float32)), dev(rng.normal(size=(R, F)).astype(np.float32))
    reds, dys = [], []
    for edge in (False, True):
        red = torch.zeros((H.STAT_SLOTS, 2, F), dtype=torch.float64, device="cuda")
        args = (mean.data_ptr(), rstd.data_ptr(), beta.data_ptr(), 1, dmx.data_ptr(), F, dmn.data_ptr(), F,
                mx.data_ptr(), F, cnt.data_ptr())
        if edge:
            eargs = args[:-1] + (cnt_packed.data_ptr(),)
            H.call("dgcnn_edge_bn_bwd_reduce_f32", *e, *eargs, red.data_ptr())
        else:
            H.call("dgcnn_bn_bwd_reduce_f32", Y.data_ptr(), R, k, F, *args, red.data_ptr())
        reds.append(host(red).sum(0))
    np.testing.assert_allclose(reds[0], reds[1], rtol=1e-6, atol=1e-4)
    # the same two sums from per-point data only (no pass over the edges)
    red = torch.zeros((H.STAT_SLOTS, 2, F), dtype=torch.float64, device="cuda")
    H.call("dgcnn_edge_bn_bwd_reduce_points_f32", mx.data_ptr(), F, mn.data_ptr(), F, cnt_packed.data_ptr(), dmx.data_ptr(), F,
           dmn.data_ptr(), F, beta.data_ptr(), R, k, F, red.data_ptr())
    rp = host(red).sum(0)
    np.testing.assert_allclose(rp, reds[0], rtol=2e-4, atol=2e-3 * max(1.0, float(np.abs(reds[0]).max()) * 1e-2))
    red0 = torch.zeros((H.STAT_SLOTS, 2, F), dtype=torch.float64, device="cuda")
    red0[0] = dev(reds[0])
    for edge in (False, True):
        red = red0.clone()
        dY = torch.empty((R * k, F), device="cuda")
        dsum = torch.zeros((R, F + 4), device="cuda")
        dbeta = torch.zeros(F, device="cuda")
        tail = (red.data_ptr(), dY.data_ptr(), dsum.data_ptr(), F + 4, dbeta.data_ptr(), 0.0)
        if edge:
            H.call("dgcnn_edge_bn_bwd_apply_f32", *e, *(args[:-1] + (cnt_packed.data_ptr(),)), *tail)
        else:
            H.call("dgcnn_bn_bwd_apply_f32", Y.data_ptr(), R, k, F, *args, *tail)
        dys.append((host(dY).copy(), host(dsum).copy(), host(dbeta).copy()))
    for a, b in zip(dys[0], dys[1]):
        np.testing.assert_array_equal(a, b)
    assert np.abs(dys[0][1][:, F:]).max() == 0
    with pytest.raises(H.HipError):
        H.call("dgcnn_edge_bn_bwd_reduce_f32", UV[:, F:].data_ptr(), 2 * F, UV.data_ptr(), 2 * F, idx.data_ptr(), B, N, k, 6,
               *args, red0.data_ptr())
    with pytest.raises(ValueError):
        H.call("dgcnn_edge_bn_bwd_reduce_points_f32", mx.data_ptr(), F, mn.data_ptr(), F, cnt_packed.data_ptr(), dmx.data_ptr(), F,
               dmn.data_ptr(), F, beta.data_ptr(), R, 256, F, red0.data_ptr())


@pytest.mark.parametrize("B,N,k", [(2, 100, 7), (24, 2048, 20), (1, 40000, 3), (3, 64, 1), (5, 777, 13), (1, 70000, 2)])
def test_edge_csr_build_and_incoming_sum(dg, B, N, k):
    """Transposed adjacency (who points at me): off is the exclusive prefix of the in-degrees, bucket j of rev
    holds exactly the edges whose neighbour is j (any order), and the gather-sum over it equals tf.gather^T."""
    from dgcnn import _hip as H
    rng = np.random.default_rng(N + k)
    idx = rng.integers(0, N, (B, N, k)).astype(np.int32)
    idx[0, :, 0] = 0 if N < 1000 else idx[0, :, 0]                      # a very popular target
    d_idx = dev(idx)
    R = B * N
    cws = torch.zeros(2 * R, dtype=torch.int32, device="cuda")
    off = torch.full((R + 1,), -1, dtype=torch.int32, device="cuda")
    rev = torch.full((R * k,), -1, dtype=torch.int32, device="cuda")
    H.call("dgcnn_edge_csr_build", d_idx.data_ptr(), B, N, k, cws.data_ptr(), off.data_ptr(), rev.data_ptr())
    tgt = (idx.reshape(B, N * k).astype(np.int64) + (np.arange(B) * N)[:, None]).reshape(-1)     # global target of every edge
    deg = np.bincount(tgt, minlength=R)
    want_off = np.concatenate([[0], np.cumsum(deg)])
    np.testing.assert_array_equal(host(off), want_off)
    r = host(rev).astype(np.int64)
    assert np.array_equal(np.sort(r), np.arange(R * k))                 # a permutation of the edges
    np.testing.assert_array_equal(tgt[r], np.repeat(np.arange(R), deg))  # bucket j holds only edges pointing at j
    F = 8
    dY = rng.normal(size=(R * k, F)).astype(np.float32)
    S = torch.empty((R, F), device="cuda")
    H.call("dgcnn_edge_gather_sum_f32", dev(dY).data_ptr(), off.data_ptr(), rev.data_ptr(), R, F, S.data_ptr(), F)
    want = np.zeros((R, F))
    np.add.at(want, tgt, dY.astype(np.float64))
    np.testing.assert_allclose(host(S), want, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("B,N,k,F,ld", [(2, 100, 7, 64, 128), (3, 64, 20, 128, 256), (1, 130, 5, 48, 100), (2, 50, 3, 4, 8)])
def test_edge_gather_add_and_weight_fold(dg, B, N, k, F, ld):
    """Y[b,i,m] = V[b, idx[b,i,m]] + U[b,i] with BatchNorm column sums; Wcat = [Wa-Wb | Wb] and its
    gradient fold (the pieces that replace the edge-level GEMM of ops.py:47-52)."""
    from dgcnn import _hip as H
    rng = np.random.default_rng(F + k)
    UV = rng.normal(size=(B * N, ld)).astype(np.float32)
    idx = rng.integers(0, N, (B, N, k)).astype(np.int32)
    d_uv, d_idx = dev(UV), dev(idx)
    Y = torch.full((B * N * k, F), float("nan"), device="cuda")
    st = torch.zeros((H.STAT_SLOTS if hasattr(H, "STAT_SLOTS") else 32, 2, F), dtype=torch.float64, device="cuda")
    H.call("dgcnn_edge_gather_add_f32", d_uv[:, F:].data_ptr(), ld, d_uv.data_ptr(), ld, d_idx.data_ptr(), B, N, k, F,
           Y.data_ptr(), st.data_ptr())
    U, V = UV[:, :F].reshape(B, N, F), UV[:, F:2 * F].reshape(B, N, F)
    ref = V[np.arange(B)[:, None, None], idx] + U[:, :, None, :]
    got = host(Y).reshape(B, N, k, F)
    np.testing.assert_array_equal(got, ref)                                  # one fp32 add per element: exact
    s = host(st).sum(0)
    r64 = ref.astype(np.float64).reshape(-1, F)
    np.testing.assert_allclose(s[0], r64.sum(0), rtol=1e-5, atol=1e-3)       # fp32 partials per thread, fp64 across
    np.testing.assert_allclose(s[1], (r64 ** 2).sum(0), rtol=1e-5, atol=1e-3)
    # no stats requested
    H.call("dgcnn_edge_gather_add_f32", d_uv[:, F:].data_ptr(), ld, d_uv.data_ptr(), ld, d_idx.data_ptr(), B, N, k, F,
           Y.data_ptr(), 0)
    np.testing.assert_array_equal(host(Y).reshape(B, N, k, F), ref)
    # no Y requested: column sums only
    st.zero_()
    H.call("dgcnn_edge_gather_add_f32", d_uv[:, F:].data_ptr(), ld, d_uv.data_ptr(), ld, d_idx.data_ptr(), B, N, k, F,
           0, st.data_ptr())
    np.testing.assert_allclose(host(st).sum(0)[0], r64.sum(0), rtol=1e-5, atol=1e-3)

    C = 5
    W0 = rng.normal(size=(2 * C, F)).astype(np.float32)
    wcat = torch.empty((C, 2 * F), device="cuda")
    H.call("dgcnn_edge_weight_split_f32", dev(W0).data_ptr(), C, F, wcat.data_ptr())
    np.testing.assert_array_equal(host(wcat), np.concatenate([W0[:C] - W0[C:], W0[C:]], axis=1))
    dwcat = rng.normal(size=(C, 2 * F)).astype(np.float32)
    acc = rng.normal(size=(2 * C, F)).astype(np.float32)
    d_acc = dev(acc)
    H.call("dgcnn_edge_wgrad_combine_f32", dev(dwcat).data_ptr(), C, F, d_acc.data_ptr())
    want = acc.copy()
    want[:C] += dwcat[:, :F]
    want[C:] += dwcat[:, F:] - dwcat[:, :F]
    np.testing.assert_array_equal(host(d_acc), want)
    with pytest.raises(H.HipError):
        H.call("dgcnn_edge_gather_add_f32", d_uv[:, F:].data_ptr(), ld, d_uv.data_ptr(), ld, d_idx.data_ptr(), B, N, k, 6,
               Y.data_ptr(), 0)


# ------------------------------------------------------------------------------------------
# model.build: logits within 1e-3 of the oracle (north_star), all three MODEL_NAMEs
# ------------------------------------------------------------------------------------------
def _run_model(dg, flags, pts, params, train, labels=None):
    from dgcnn import _engine as E
    tv = dg.trainval(flags)
    tv.initialize()
    _set_vars(dg, params)
    cap = {}
    orig = E.edge_conv_block

    def capture(x, B, N, k, F, **kw):
        mm, net, idx = orig(x, B, N, k, F, **kw)
        cap["/".join(E.ctx().scope)] = (host(x).reshape(B, N, -1).copy(), host(idx))
        return mm, net, idx
    E.edge_conv_block = capture
    try:
        if train:
            tv.zero_gradients(None)
            res = tv.accum_gradient(None, [pts], [labels])
        else:
            res = tv.inference(None, [pts], None if labels is None else [labels])
    finally:
        E.edge_conv_block = orig
    return tv, res, cap


CONFIGS = [
    # BASELINE config 1: B=2 N=512 k=10 C=3, 1 EdgeConv layer, FC 512,256, 2 classes
    dict(MODEL_NAME="dgcnn", EDGE_CONV_LAYERS=1, EDGE_CONV_FILTERS=64, KVALUE=10, B=2, N=512, C=3),
    # config-2 architecture at reduced B, N
    dict(MODEL_NAME="dgcnn", EDGE_CONV_LAYERS=3, EDGE_CONV_FILTERS=[64, 64, 128], KVALUE=20, B=3, N=256, C=3),
    dict(MODEL_NAME="residual-dgcnn", EDGE_CONV_LAYERS=3, EDGE_CONV_FILTERS=64, KVALUE=12, B=2, N=192, C=4),   # G8
    dict(MODEL_NAME="residual-dgcnn-nofc", EDGE_CONV_LAYERS=2, EDGE_CONV_FILTERS=64, KVALUE=8, B=2, N=128, C=3),
    # odd shapes: filter counts that are not multiples of 4 (conv0 falls back to the edge-level factored GEMM, scalar
    # BN / GEMM paths), ragged N, one cloud, k = 3
    dict(MODEL_NAME="dgcnn", EDGE_CONV_LAYERS=2, EDGE_CONV_FILTERS=[6, 10], KVALUE=3, B=1, N=50, C=3),
    # filter counts that are multiples of 4 but not powers of two (gather path with 12 / 20 float4 lanes per row), C = 4
    dict(MODEL_NAME="dgcnn", EDGE_CONV_LAYERS=2, EDGE_CONV_FILTERS=[48, 80], KVALUE=5, B=2, N=77, C=4),
    # residual stack whose filter count changes to 64: the `shortcut` 1x1 conv + BN (no activation) of ops.py:124-133 EXISTS
    # and runs (conv1 always emits 64 channels, so 64 is the only changed width whose add at ops.py:134 has matching shapes)
    dict(MODEL_NAME="residual-dgcnn", EDGE_CONV_LAYERS=3, EDGE_CONV_FILTERS=[32, 64, 64], KVALUE=9, B=2, N=160, C=3),
]


# Default mode = the deterministic kernels (one writer per statistics slot, fixed-order sums, sorted adjacency): a run is ONE set of
# numbers, so the gradient bars sit at ~2x what is measured instead of covering the run-to-run spread of the opt-in atomics mode.
# Measured (round 4, three runs, identical): see DET_GRAD_BAR below.
DET_GRAD_FRO = 5e-3
DET_GRAD_ELT = 2e-2


@pytest.mark.parametrize("det", [True, False], ids=["default", "atomics"])      # default = the deterministic kernels (round 5)
@pytest.mark.parametrize("cfg", CONFIGS)
def test_model_logits_and_gradients(dg, cfg, det):
    cfg = dict(cfg)
    B, N, C = cfg.pop("B"), cfg.pop("N"), cfg.pop("C")
    if det and any(f % 4 for f in (cfg["EDGE_CONV_FILTERS"] if isinstance(cfg["EDGE_CONV_FILTERS"], list) else [cfg["EDGE_CONV_FILTERS"]])):
        pytest.skip("deterministic mode refuses EdgeConv filter counts that are not multiples of 4")
    flags = dg.DGCNN_FLAGS(NUM_CLASS=2, FC_LAYERS=2, FC_FILTERS=[512, 256], TRAIN=False, NUM_CHANNEL=C,
                           DETERMINISTIC=None if det else False, **cfg)      # None: the library default; False: the atomics mode
    rng = np.random.default_rng(0)
    pts = rng.random((B, N, C), dtype=np.float32)
    labels = rng.integers(0, 2, (B, N)).astype(np.int32)
    params = O.init_params(flags, C, seed=1)
    for n in params:
        if n.endswith("beta"):
            params[n] = rng.normal(0, 0.2, params[n].shape).astype(np.float32)

    # ---- inference graph (TRAIN False: no dropout), logits via softmax inversion is lossy -> use build()
    tv, res, cap = _run_model(dg, flags, pts, params, train=False, labels=labels)
    L = int(flags.EDGE_CONV_LAYERS)
    idx_list = []
    for i in range(L):
        xin, idx = cap["EdgeConv%d" % i]
        np.testing.assert_array_equal(idx, O.k_nn(xin, int(flags.KVALUE)))       # bit-exact on identical inputs
        idx_list.append(idx)
    logits_ref, _ = O.model_forward(pts, flags, params, idx_list=idx_list)
    loss_ref, sm_ref, acc_ref, _ = O.softmax_xent(logits_ref, labels)
    import dgcnn
    dg.ctx().recording = False
    logits = host(dgcnn.build(dev(pts), flags))
    assert logits.shape == (B, N, 2)
    np.testing.assert_allclose(logits, logits_ref, rtol=0, atol=1e-3)             # north_star: 1e-3 fp32
    np.testing.assert_allclose(host(res[0]), sm_ref, rtol=0, atol=1e-3)
    assert abs(float(res[-1]) - float(loss_ref)) < 1e-3 and abs(float(res[-2]) - float(acc_ref)) < 2.0 / (B * N)

    # ---- training graph without dropout randomness: compare every gradient with the fp64 twin
    flags.TRAIN = True
    import dgcnn._engine as E
    keep = E.DROPOUT_KEEP
    E.DROPOUT_KEEP = 1.0
    try:
        tv, res, cap = _run_model(dg, flags, pts, params, train=True, labels=labels)
    finally:
        E.DROPOUT_KEEP = keep
    idx_list = [cap["EdgeConv%d" % i][1] for i in range(L)]
    p64 = {n: v.astype(np.float64) for n, v in params.items()}
    G, loss64, _, _ = O.train_step_grads(pts.astype(np.float64), labels, flags, p64, idx_list=idx_list)
    assert abs(float(res[2]) - float(loss64)) < 1e-3
    # fp32 path vs the fp64 twin fed the same graphs.  Measured (profiles/r02/grad_error_3way.txt): 6e-5 .. 1.4e-4 relative
    # Frobenius at (3,256), 1e-3 .. 2.2e-3 at (24,2048) -- an order of magnitude CLOSER to the twin than the fp32 numpy
    # restatement is (3e-3 / 3e-2: its float32 BatchNorm reductions).  Isolated elements move more when a ReLU /
    # max-over-k / global-max decision flips in fp32 (one point's whole contribution is rerouted): the elementwise
    # bar is 5e-2 of the tensor's scale (observed up to 2.1e-2), the Frobenius bar 2e-2 (observed 1e-6 .. 1.1e-2 over the seven configurations and
    # run to run: the tiny (3,256) clouds feel a single flipped decision most).  A wrong or missing term is O(1).
    worst, worst_e = (0.0, ""), (0.0, "")
    bar_fro, bar_elt = (DET_GRAD_FRO, DET_GRAD_ELT) if det else (2e-2, 5e-2)
    for n in params:
        g = host(tv.gradients[n]).astype(np.float64)
        ref = G[n]
        scale = max(float(np.abs(ref).max()), 1e-3)
        err = np.abs(g - ref)
        fro = np.linalg.norm(g - ref) / max(np.linalg.norm(ref), 1e-6)
        worst = max(worst, (fro, n))
        worst_e = max(worst_e, (float(err.max() / scale), n))
    print("%s [%s]: worst relative Frobenius gradient error vs the fp64 twin %.2e (%s), worst element / scale %.2e (%s)"
          % (cfg.get("MODEL_NAME"), "default (deterministic)" if det else "atomics", worst[0], worst[1], worst_e[0], worst_e[1]))
    from dgcnn import _engine as E2
    E2.DETERMINISTIC = E2.DETERMINISTIC_ENV_DEFAULT
    assert worst_e[0] <= bar_elt, worst_e
    assert worst[0] <= bar_fro, worst


@pytest.mark.parametrize("ncls", [3, 4, 5, 8])
def test_logits_other_class_counts(dg, ncls):
    """NUM_CLASS other than 2: the class-dimension products switch between the streaming kernels (<= 4 classes)
    and the generic / bf16-split GEMMs; logits within 1e-3 of the oracle, loss finite, one optimizer step runs."""
    flags = dg.DGCNN_FLAGS(EDGE_CONV_LAYERS=2, EDGE_CONV_FILTERS=[32, 64], KVALUE=6, FC_LAYERS=1, FC_FILTERS=[64], NUM_CLASS=ncls,
                           NUM_CHANNEL=3, TRAIN=False)
    rng = np.random.default_rng(ncls)
    B, N = 2, 160
    pts = rng.random((B, N, 3), dtype=np.float32)
    labels = rng.integers(0, ncls, (B, N)).astype(np.int32)
    params = O.init_params(flags, 3, seed=2)
    tv, res, cap = _run_model(dg, flags, pts, params, train=False, labels=labels)
    idx_list = [cap["EdgeConv%d" % i][1] for i in range(2)]
    logits_ref, _ = O.model_forward(pts, flags, params, idx_list=idx_list)
    loss_ref, sm_ref, acc_ref, _ = O.softmax_xent(logits_ref, labels)
    np.testing.assert_allclose(host(res[0]), sm_ref, rtol=0, atol=1e-3)
    assert abs(float(res[-1]) - float(loss_ref)) < 1e-3
    flags.TRAIN = True
    tv2 = dg.trainval(flags).initialize()
    tv2.zero_gradients(None)
    r = tv2.accum_gradient(None, [pts], [labels])
    tv2.apply_gradient(None)
    assert np.isfinite(float(r[2])) and float(dg.ctx().flat_grad.abs().sum()) > 0


def test_side_stream_does_not_change_results(dg):
    """Weight-gradient GEMMs / the transposed adjacency run on a second HIP stream; with it switched off the
    same gradients must come out (up to the run-to-run atomic-order noise of the BN sums)."""
    from dgcnn import _engine as E
    flags = dg.DGCNN_FLAGS(EDGE_CONV_LAYERS=2, EDGE_CONV_FILTERS=[64, 64], KVALUE=10, FC_FILTERS=[64, 32], NUM_CHANNEL=3,
                           TRAIN=True, SEED=3)
    rng = np.random.default_rng(11)
    pts = rng.random((4, 512, 3), dtype=np.float32)
    lab = rng.integers(0, 2, (4, 512)).astype(np.int32)
    keep, E.DROPOUT_KEEP = E.DROPOUT_KEEP, 1.0
    old, old_min = E.WGRAD_SIDE_STREAM, E.SIDE_STREAM_MIN_ROWS
    E.SIDE_STREAM_MIN_ROWS = 0                                   # the test shape is below the production threshold
    got = {}
    try:
        for side in (False, True):
            E.WGRAD_SIDE_STREAM = side
            tv = dg.trainval(flags).initialize()
            for _ in range(2):                                   # two micro-steps: accumulation across joins
                tv.accum_gradient(None, [pts], [lab])
            got[side] = host(dg.ctx().flat_grad).copy()
    finally:
        E.WGRAD_SIDE_STREAM, E.SIDE_STREAM_MIN_ROWS = old, old_min
        E.DROPOUT_KEEP = keep
    d = np.abs(got[True] - got[False])
    scale = np.abs(got[False]).max()
    assert np.linalg.norm(got[True] - got[False]) <= 2e-3 * np.linalg.norm(got[False]), d.max() / scale
    assert np.abs(got[True]).sum() > 0


def test_two_microsteps_and_adam(dg):
    """gradients are SUMMED over micro-steps (trainval.py:79), then one Adam step."""
    flags = dg.DGCNN_FLAGS(EDGE_CONV_LAYERS=1, KVALUE=6, FC_FILTERS=[32, 16], NUM_CHANNEL=3, TRAIN=True)
    rng = np.random.default_rng(3)
    import dgcnn._engine as E
    keep = E.DROPOUT_KEEP
    E.DROPOUT_KEEP = 1.0
    try:
        tv = dg.trainval(flags).initialize()
        params = {n: host(v).copy() for n, v in tv.variables.items()}
        tv.zero_gradients(None)
        gsum = {n: np.zeros_like(v, dtype=np.float64) for n, v in params.items()}
        for step in range(2):
            pts = rng.random((2, 64, 3), dtype=np.float32)
            lab = rng.integers(0, 2, (2, 64)).astype(np.int32)
            w = rng.random((2, 64), dtype=np.float32)
            tv.accum_gradient(None, [pts], [lab], [w])
            G, _, _, _ = O.train_step_grads(pts.astype(np.float64), lab, flags,
                                            {n: v.astype(np.float64) for n, v in params.items()}, weight=w)
            for n in G:
                gsum[n] += G[n]
        for n in params:
            np.testing.assert_allclose(host(tv.gradients[n]), gsum[n], rtol=2e-3,
                                       atol=2e-3 * max(np.abs(gsum[n]).max(), 1e-3), err_msg=n)
        tv.apply_gradient(None)
        for n in params:
            p = params[n].astype(np.float64).copy()
            m = np.zeros_like(p)
            v = np.zeros_like(p)
            O.adam_step(p, host(tv.gradients[n]).astype(np.float64), m, v, 1, flags.LEARNING_RATE)
            np.testing.assert_allclose(host(tv.variables[n]), p, rtol=1e-5, atol=1e-6, err_msg=n)
    finally:
        E.DROPOUT_KEEP = keep


def test_errors_match_reference(dg):
    flags = dg.DGCNN_FLAGS(MODEL_NAME="nope")
    with pytest.raises(NotImplementedError):
        dg.trainval(flags).initialize()
    with pytest.raises(ValueError):
        dg.ops.repeat_edge_conv(torch.zeros(1, 32, 3, device="cuda"), 3, [5, 5], 64, True)   # ops.py:80-82
    f2 = dg.DGCNN_FLAGS(TRAIN=False)
    tv = dg.trainval(f2).initialize()
    for call in (lambda: tv.zero_gradients(None), lambda: tv.apply_gradient(None),
                 lambda: tv.accum_gradient(None, [None], [None])):
        with pytest.raises(NotImplementedError):                                              # trainval.py:111-128
            call()


def test_residual_shortcut_shape_mismatch_is_reproduced(dg):
    """ops.py:124-134: when num_filters[i] != num_filters[i-1] the shortcut is convolved to num_filters[i] channels while conv1
    always emits 64 (ops.py:62-63), so any changed width other than 64 fails at the add -- TF raises a shape error at
    graph-build time; here ValueError at the same call."""
    x = torch.rand(1, 64, 3, device="cuda")
    with pytest.raises(ValueError, match="shortcut"):
        dg.ops.repeat_residual_edge_conv(x, 2, 5, [64, 128], True)
    dg.reset()
    out = dg.ops.repeat_residual_edge_conv(x, 2, 5, [32, 64], True)       # changed width 64: the shortcut conv runs
    assert "EdgeConv1/shortcut/weights" in dg.ctx().vars and tuple(dg.ctx().vars["EdgeConv1/shortcut/weights"].shape) == (64, 64)
    assert tuple(out[-1].shape) == (1, 64, 1, 64) and float(out[-1].min()) >= 0.0      # relu(shortcut + net)
    dg.reset()
    dg.ops.repeat_residual_edge_conv(x, 2, 5, 64, True)
    assert not any("shortcut" in n for n in dg.ctx().vars)               # equal widths: no shortcut variables (production, train_dgcnn.sh:9)


def test_dropout_and_misc_kernels(dg):
    from dgcnn import _engine as E, _hip as H
    x = torch.ones(1 << 20, device="cuda")
    y = torch.empty_like(x)
    H.call("dgcnn_dropout_f32", x.data_ptr(), y.data_ptr(), x.numel(), 0.7, 1234)
    yh = host(y)
    kept = (yh != 0)
    assert abs(kept.mean() - 0.7) < 5e-3 and np.allclose(yh[kept], 1 / 0.7)
    y2 = torch.empty_like(x)
    H.call("dgcnn_dropout_f32", x.data_ptr(), y2.data_ptr(), x.numel(), 0.7, 1234)
    assert torch.equal(y, y2)                                            # same seed -> same mask (backward)
    rng = np.random.default_rng(0)
    a = rng.normal(size=(3, 500, 70)).astype(np.float32)
    a[1, 17] = a[1, 400] = 9.0                                          # tie -> first arg-max
    c = E.ctx()
    out = torch.empty((3, 70), device="cuda")
    arg = torch.empty((3, 70), dtype=torch.int32, device="cuda")
    H.call("dgcnn_global_max_f32", dev(a).data_ptr(), 70, 3, 500, 70, out.data_ptr(), arg.data_ptr())
    np.testing.assert_array_equal(host(out), a.max(1))
    np.testing.assert_array_equal(host(arg), a.argmax(1))
    cs = torch.empty((3, 70), device="cuda")
    H.call("dgcnn_group_colsum_f32", dev(a).data_ptr(), 70, 3, 500, 70, cs.data_ptr())
    np.testing.assert_allclose(host(cs), a.astype(np.float64).sum(1), rtol=1e-5, atol=1e-4)


# ------------------------------------------------------------------------------------------
# per-block gradients of the head: MergedEdgeConv -> global max -> FC0 with the FOLDED global feature
# (model.py:65-88; the tiled 1024 channels are never materialised, see dgcnn/model.py) -> FC1 -> Final, against torch fp64
# autograd of the literal graph (tile + concat).  Block-level bars like test_edge_conv_forward_backward: 1e-3.
# ------------------------------------------------------------------------------------------
def _bn_act64(t, beta, relu=True):
    mu = t.mean(0, keepdim=True)
    var = t.var(0, unbiased=False, keepdim=True)
    z = (t - mu) / torch.sqrt(var + 1e-3) + beta
    return torch.relu(z) if relu else z


@pytest.mark.parametrize("B,N,widths,fcf,ncls", [(2, 96, [64, 64, 128], [512, 256], 2), (3, 50, [8, 12], [64, 32], 3)])
def test_head_block_gradients(dg, B, N, widths, fcf, ncls):
    from dgcnn import _engine as E
    rng = np.random.default_rng(B * N)
    R, L = B * N, len(widths)
    local_w = sum(2 * f + 64 for f in widths)
    ctot = local_w + 1024
    local = rng.normal(size=(R, local_w)).astype(np.float32)
    min_ = np.abs(rng.normal(size=(R, 64 * L))).astype(np.float32)
    P = {"MergedEdgeConv/weights": rng.normal(0, 0.1, (64 * L, 1024)), "MergedEdgeConv/BatchNorm/beta": rng.normal(0, 0.2, 1024),
         "FC0/weights": rng.normal(0, 0.05, (1024 + ctot, fcf[0])), "FC0/BatchNorm/beta": rng.normal(0, 0.2, fcf[0]),
         "FC1/weights": rng.normal(0, 0.1, (fcf[0], fcf[1])), "FC1/BatchNorm/beta": rng.normal(0, 0.2, fcf[1]),
         "Final/weights": rng.normal(0, 0.2, (fcf[1], ncls)), "Final/BatchNorm/beta": rng.normal(0, 0.2, ncls)}
    P = {n: v.astype(np.float32) for n, v in P.items()}
    dfin = rng.normal(size=(R, ncls)).astype(np.float32)

    # ---- HIP path: the statements of dgcnn/model.py:build for the head ----
    c = dg.ctx()
    c.begin_step()
    c.recording = True
    for n, v in P.items():
        c.get_variable(n, v.shape)
    _set_vars(dg, P)
    big = c.new_buffer(R, ctot)
    big[:, :local_w].copy_(dev(local))
    merged_in = c.new_buffer(R, 64 * L)
    merged_in.copy_(dev(min_))
    merged = E.conv_bn_act(merged_in, "MergedEdgeConv", 1024, relu=True, out=big[:, ctot - 1024:])
    g = E.global_max(merged, B, N)
    with E.variable_scope("FC0"):
        wleaf = c.get_variable("weights", (1024 + ctot, fcf[0]))
    gb = E.plain_gemm(g, wleaf, (0, 1024), fcf[0])
    net = E.conv_bn_act(big, "FC0", fcf[0], relu=True, gbias=gb, rpg=N, w_rows=(1024, 1024 + ctot, 1024 + ctot))
    net = E.conv_bn_act(net, "FC1", fcf[1], relu=True)
    fin = E.conv_bn_act(net, "Final", ncls, relu=True)
    c.grad(fin).copy_(dev(dfin))
    c.backward()
    torch.cuda.synchronize()

    # ---- torch fp64 autograd of the literal graph (global feature tiled and concatenated in front: model.py:80-85) ----
    t = {n: torch.tensor(v.astype(np.float64), requires_grad=True) for n, v in P.items()}
    tl = torch.tensor(local.astype(np.float64), requires_grad=True)
    tm = torch.tensor(min_.astype(np.float64), requires_grad=True)
    m64 = _bn_act64(tm @ t["MergedEdgeConv/weights"], t["MergedEdgeConv/BatchNorm/beta"])
    g64 = m64.view(B, N, 1024).max(1).values
    x64 = torch.cat([g64[:, None, :].expand(B, N, 1024).reshape(R, 1024), tl, m64], 1)
    h = _bn_act64(x64 @ t["FC0/weights"], t["FC0/BatchNorm/beta"])
    h = _bn_act64(h @ t["FC1/weights"], t["FC1/BatchNorm/beta"])
    f64 = _bn_act64(h @ t["Final/weights"], t["Final/BatchNorm/beta"])
    (f64 * torch.tensor(dfin.astype(np.float64))).sum().backward()

    np.testing.assert_allclose(host(fin), f64.detach().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(host(g), g64.detach().numpy(), rtol=1e-4, atol=1e-4)
    scale = lambda r: 1e-3 * max(1.0, float(np.abs(r).max()))
    chk = [("d merged_in", host(c.grad(merged_in)), tm.grad.numpy()), ("d local", host(c.grad(big))[:, :local_w], tl.grad.numpy())]
    chk += [(n, host(c.var_grads[n]), t[n].grad.numpy()) for n in P]
    for name, got, ref in chk:
        np.testing.assert_allclose(got, ref, rtol=1e-3, atol=scale(ref), err_msg=name)
        fro = np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-12)
        assert fro <= 1e-3, (name, fro)


@pytest.mark.parametrize("R,F,ld", [(300, 64, 64), (77, 12, 20), (1000, 128, 192)])
def test_residual_add_relu_gradients_exact(dg, R, F, ld):
    """ops.py:134 `relu(net + shortcut)`: forward and both input gradients are exact (elementwise, no reduction)."""
    from dgcnn import _engine as E
    rng = np.random.default_rng(R + F)
    a, b = rng.normal(size=(R, F)).astype(np.float32), rng.normal(size=(R, F)).astype(np.float32)
    d = rng.normal(size=(R, F)).astype(np.float32)
    c = dg.ctx()
    c.begin_step()
    c.recording = True
    ta, tb_full = c.new_buffer(R, F), c.new_buffer(R, ld)
    tb = tb_full[:, :F]                                           # a strided view, as the concat buffer hands out
    ta.copy_(dev(a)); tb.copy_(dev(b))
    out = E.add_relu(ta, tb)
    c.grad(out).copy_(dev(d))
    c.backward()
    ref = np.maximum(a + b, 0)
    np.testing.assert_array_equal(host(out), ref)
    gd = d * (ref > 0)
    np.testing.assert_array_equal(host(c.grad(ta)), gd)
    np.testing.assert_array_equal(host(c.grad(tb_full))[:, :F], gd)


@pytest.mark.parametrize("B,N,C,k,F", [(2, 4096, 3, 20, 64), (3, 1500, 4, 9, 128), (1, 8192, 3, 12, 32)])
def test_first_layer_fused_backward_matches_the_unfused_passes(dg, B, N, C, k, F):
    """Input layer of the model (C <= 4, nobody asks for d(points)): dgcnn_edge_bn_bwd_apply_wgrad_f32 forms dY per edge and
    consumes it in the same pass (dW0 += [x_i, x_j - x_i]^T dY) -- against the general path (dY written, transposed-adjacency
    sum, two point-level GEMMs) on the same upstream gradients, and against the float64 oracle."""
    from dgcnn import _engine as E
    rng = np.random.default_rng(B * N + F)
    pts = rng.random((B, N, C), dtype=np.float32)
    P = {"conv0/weights": rng.normal(0, 0.5, (2 * C, F)).astype(np.float32),
         "conv0/BatchNorm/beta": rng.normal(0, 0.3, F).astype(np.float32),
         "conv1/weights": rng.normal(0, 0.3, (2 * F, 64)).astype(np.float32),
         "conv1/BatchNorm/beta": rng.normal(0, 0.3, 64).astype(np.float32)}
    ups = [rng.normal(size=(B * N, F)).astype(np.float32), rng.normal(size=(B * N, F)).astype(np.float32),
           rng.normal(size=(B * N, 64)).astype(np.float32)]
    got = {}
    old = E.EDGE_BWD_FUSED_L0
    old_det, E.DETERMINISTIC = E.DETERMINISTIC, True      # both variants on identical forward values: only the summation order of dW0 differs
    calls = []
    orig = E.H.call

    def spy(name, *a, **kw):
        calls.append(name)
        return orig(name, *a, **kw)
    try:
        for fused in (True, False):
            E.EDGE_BWD_FUSED_L0 = fused
            dg.reset()
            c = dg.ctx()
            c.begin_step()
            c.recording = True
            for n, v in P.items():
                c.get_variable(n, v.shape)
            _set_vars(dg, P)
            x = dev(pts)                                      # NOT a tracked buffer: no d(point_cloud) is requested
            del calls[:]
            E.H.call = spy
            try:
                outs = dg.ops.edge_conv(x, k, F, True)
                for t, g in zip(outs, ups):
                    v, _, _ = E.as2d(t)
                    c.grad(v).copy_(dev(g))
                c.backward()
            finally:
                E.H.call = orig
            assert ("dgcnn_edge_bn_bwd_apply_wgrad_f32" in calls) == fused
            assert ("dgcnn_edge_bn_bwd_apply_f32" in calls) == (not fused) and ("dgcnn_edge_csr_build" in calls) == (not fused)
            got[fused] = {n: host(c.var_grads[n]).copy() for n in P}
            idx = host(dg.ops.edge_conv.last_idx)
    finally:
        E.EDGE_BWD_FUSED_L0 = old
        E.DETERMINISTIC = old_det
        dg.reset()
    for n in P:
        a, b = got[True][n].astype(np.float64), got[False][n].astype(np.float64)
        # deterministic mode: the two forward passes are bit-identical, so no ReLU / max decision can flip between the runs (in
        # the default mode that cost up to 1e-3 of the norm); what is left is fp32 summation order.  A wrong term is O(1)
        assert np.linalg.norm(a - b) <= 2e-5 * max(np.linalg.norm(b), 1e-9), (n, np.linalg.norm(a - b) / np.linalg.norm(b))
    ref, cache = O.edge_conv(pts.astype(np.float64), k, *[P[n].astype(np.float64) for n in P], idx=idx)
    d = [u.reshape(B, N, 1, -1).astype(np.float64) for u in ups]
    _, g_ref = O.edge_conv_bwd(d[0], d[1], d[2], cache)
    for n, key in (("conv0/weights", "W0"), ("conv0/BatchNorm/beta", "beta0")):
        r = g_ref[key]
        assert np.linalg.norm(got[True][n] - r) <= 1e-2 * np.linalg.norm(r), (n, np.linalg.norm(got[True][n] - r) / np.linalg.norm(r))


def test_dropout_fused_into_the_last_fc_layer_equals_the_separate_passes(dg):
    """model.py:88-91: fc -> tf.nn.dropout(0.7) -> Final.  The last FC layer's BatchNorm passes apply the dropout mask themselves
    (forward: the dropped output is the only one stored; backward: d(dropped output) is read through the regenerated mask).
    Same seed => same mask as the separate dropout kernels: identical loss, gradients equal up to the run-to-run noise of the atomically summed statistics."""
    from dgcnn import _engine as E
    rng = np.random.default_rng(77)
    B, N = 4, 1024
    pts = rng.random((B, N, 3), dtype=np.float32)
    lab = rng.integers(0, 2, (B, N)).astype(np.int32)
    flags = dg.DGCNN_FLAGS(MODEL_NAME="dgcnn", EDGE_CONV_LAYERS=1, EDGE_CONV_FILTERS=64, FC_LAYERS=2, FC_FILTERS=[128, 64],
                           NUM_CLASS=2, KVALUE=8, NUM_CHANNEL=3, TRAIN=True, SEED=9)
    old = E.FUSE_DROPOUT
    flags.DETERMINISTIC = True          # bit-identical forward passes: the comparison below sees only the fusion
    out = {}
    calls = []
    orig = E.H.call

    def spy(name, *a, **kw):
        calls.append(name)
        return orig(name, *a, **kw)
    try:
        for fused in (True, False):
            E.FUSE_DROPOUT = fused
            tv = dg.trainval(flags).initialize()
            del calls[:]
            E.H.call = spy
            try:
                tv.zero_gradients(None)
                res = tv.accum_gradient(None, [pts], [lab])
            finally:
                E.H.call = orig
            assert ("dgcnn_bn1_act_dropout_f32" in calls) == fused and ("dgcnn_bn1_bwd_dropout_f32" in calls) == fused
            assert (calls.count("dgcnn_dropout_dev_f32") == 0) == fused
            out[fused] = (float(res[2]), host(dg.ctx().flat_grad).copy())
    finally:
        E.FUSE_DROPOUT = old
        E.DETERMINISTIC = False
        dg.reset()
    assert abs(out[True][0] - out[False][0]) < 2e-6, (out[True][0], out[False][0])
    ga, gb = out[True][1].astype(np.float64), out[False][1].astype(np.float64)
    # deterministic mode on both sides (in the default mode two runs of the SAME variant differ by ~3e-4 here); a wrong mask in the
    # backward is O(0.3)
    assert np.linalg.norm(ga - gb) <= 2e-5 * np.linalg.norm(gb), np.linalg.norm(ga - gb) / np.linalg.norm(gb)
    # and the mask is really applied: about 30 % of the Final layer's input is zero
    E.FUSE_DROPOUT = True
    try:
        tv = dg.trainval(flags).initialize()
        seen = {}
        orig_cba = E.conv_bn_act

        def hook(x, scope, *a, **kw):
            if scope == "Final":
                seen["frac0"] = float((x == 0).float().mean())
            return orig_cba(x, scope, *a, **kw)
        E.conv_bn_act = hook
        try:
            tv.accum_gradient(None, [pts], [lab])
        finally:
            E.conv_bn_act = orig_cba
    finally:
        E.FUSE_DROPOUT = old
        dg.reset()
    assert 0.3 <= seen["frac0"] <= 0.9, seen          # 30 % dropped + the ReLU's own zeros


@pytest.mark.parametrize("B,N,Cin,F", [(3, 512, 64, 128), (2, 1024, 192, 1024), (5, 256, 32, 96)])
def test_column_maximum_from_the_gemm_epilogue(dg, B, N, Cin, F):
    """model.py:76-77 (max_pool_v2 over the points of a cloud): the per-cloud column maximum and its FIRST row come out of the
    epilogue of the GEMM that produces the tensor (packed keys + dgcnn_colmax_decode_f32) -- against numpy, ties included."""
    from dgcnn import _engine as E
    rng = np.random.default_rng(B * N + F)
    X = rng.normal(size=(B * N, Cin)).astype(np.float32)
    X[rng.integers(0, B * N, 200)] = X[0]                       # duplicate rows: exact ties of whole output rows
    W = rng.normal(0, 0.2, size=(Cin, F)).astype(np.float32)
    W[:, :3] = 0.0                                              # all-equal columns: every row ties, the first must win
    T = torch.empty((B * N, F), device="cuda")
    keys = torch.zeros(B * F, dtype=torch.int64, device="cuda")
    st = torch.zeros(E.H.STAT_SLOTS * 2 * F, dtype=torch.float64, device="cuda")
    E.gemm(dev(X), dev(W), T, stats=st, colmax=keys, colmax_rpg=N)
    vals = torch.empty((B, F), device="cuda")
    arg = torch.empty((B, F), dtype=torch.int32, device="cuda")
    E.H.call("dgcnn_colmax_decode_f32", keys.data_ptr(), B * F, vals.data_ptr(), arg.data_ptr())
    Th = host(T).reshape(B, N, F)
    np.testing.assert_array_equal(host(vals), Th.max(1))
    np.testing.assert_array_equal(host(arg), Th.argmax(1))          # numpy's argmax is the first maximum too
    s = host(st).reshape(E.H.STAT_SLOTS, 2, F).sum(0)               # the BatchNorm column sums of the same epilogue
    np.testing.assert_allclose(s[0], Th.reshape(-1, F).astype(np.float64).sum(0), rtol=1e-5, atol=1e-2)
    np.testing.assert_allclose(s[1], (Th.reshape(-1, F).astype(np.float64) ** 2).sum(0), rtol=1e-5, atol=1e-2)


@pytest.mark.parametrize("M,N,K,transB", [(8200, 64, 128, False), (8200, 64, 256, True), (12345, 128, 64, False), (9000, 256, 64, True),
                                          (4100, 64, 64, False), (49152, 256, 64, False)])
def test_short_reduction_gemms_of_the_edgeconv_blocks(dg, M, N, K, transB):
    """conv1 / the point-level [U|V] product / their data gradients (ops.py:47-70): K <= 256, N <= 256 over many rows (64-row
    tiles) -- product, beta accumulate, the BatchNorm column sums against float64, and output into a column slice."""
    from dgcnn import _engine as E
    rng = np.random.default_rng(M + N + K)
    A = rng.normal(size=(M, K)).astype(np.float32)
    A[rng.integers(0, M, 50)] = 0.0
    W = rng.normal(0, 0.3, size=(N, K) if transB else (K, N)).astype(np.float32)
    ref = A.astype(np.float64) @ (W.T if transB else W).astype(np.float64)
    tol = 2e-6 * K ** 0.5 * 4 + 1e-5
    Ad, Wd = dev(A), dev(W)
    C = torch.empty((M, N), device="cuda")
    st = torch.zeros(E.H.STAT_SLOTS * 2 * N, dtype=torch.float64, device="cuda")
    E.gemm(Ad, Wd, C, transB=transB, stats=st)
    np.testing.assert_allclose(host(C), ref, rtol=1e-5, atol=tol)
    s = host(st).reshape(E.H.STAT_SLOTS, 2, N).sum(0)
    np.testing.assert_allclose(s[0], ref.sum(0), rtol=1e-5, atol=2e-2)
    np.testing.assert_allclose(s[1], (ref ** 2).sum(0), rtol=1e-5, atol=2e-2)
    old = rng.normal(size=(M, N)).astype(np.float32)
    C2 = dev(old)
    E.gemm(Ad, Wd, C2, transB=transB, beta=1.0)
    np.testing.assert_allclose(host(C2), ref + old, rtol=1e-5, atol=tol)
    # into a column slice of a wider tensor (the concat buffer of model.build): leading dimension != N
    wide = torch.zeros((M, N + 40), device="cuda")
    E.gemm(Ad, Wd, wide[:, 8:8 + N], transB=transB)
    np.testing.assert_allclose(host(wide[:, 8:8 + N]), ref, rtol=1e-5, atol=tol)
    assert float(wide[:, :8].abs().max()) == 0.0 and float(wide[:, 8 + N:].abs().max()) == 0.0


def test_training_trajectory_of_eight_adam_steps_follows_the_oracle(dg):
    """trainval.py:75-80 (zero -> accumulate -> apply) eight times on fresh batches, dropout off: at EVERY step the loss the HIP path
    reports agrees with the float64 oracle that is stepped alongside it (same Adam, same neighbour graphs as the HIP path built
    from its own current features), and after the eight steps the accumulated parameter movement agrees where it is not noise.
    (The oracle follows its OWN parameter trajectory: an error in any gradient or in the optimizer would separate the two.)"""
    from gpu_helpers import capture_layers
    import dgcnn._engine as E
    flags = dg.DGCNN_FLAGS(MODEL_NAME="dgcnn", EDGE_CONV_LAYERS=2, EDGE_CONV_FILTERS=[32, 32], KVALUE=8, FC_LAYERS=2,
                           FC_FILTERS=[64, 32], NUM_CLASS=2, NUM_CHANNEL=3, TRAIN=True, SEED=11, LEARNING_RATE=1e-3,
                           DETERMINISTIC=True)
    rng = np.random.default_rng(21)
    keep, E.DROPOUT_KEEP = E.DROPOUT_KEEP, 1.0
    try:
        tv = dg.trainval(flags).initialize()
        p64 = {n: host(v).astype(np.float64) for n, v in tv.variables.items()}
        init = {n: v.copy() for n, v in p64.items()}
        m = {n: np.zeros_like(v) for n, v in p64.items()}
        vv = {n: np.zeros_like(v) for n, v in p64.items()}
        losses = []
        for step in range(1, 9):
            pts = rng.random((3, 256, 3), dtype=np.float32)
            lab = (pts[..., 0] + 0.3 * pts[..., 1] > 0.6).astype(np.int32)          # a learnable rule
            tv.zero_gradients(None)
            with capture_layers(keep_inputs=False) as cap:
                res = tv.accum_gradient(None, [pts], [lab])
            tv.apply_gradient(None)
            idx_list = [cap.layers["EdgeConv%d" % i][1] for i in range(2)]
            G, loss64, _, _ = O.train_step_grads(pts.astype(np.float64), lab, flags, p64, idx_list=idx_list)
            for n in p64:
                O.adam_step(p64[n], G[n], m[n], vv[n], step, flags.LEARNING_RATE)
            losses.append((float(res[2]), float(loss64)))
        for step, (lh, lo) in enumerate(losses, 1):
            assert abs(lh - lo) < 2e-5 * max(1.0, abs(lo)), (step, lh, lo)            # measured 2.8e-6 over the eight steps
        assert losses[-1][1] < losses[0][1]                                          # and it learns
        worst = 0.0
        for n in p64:
            moved_o = p64[n] - init[n]
            moved_h = host(tv.variables[n]).astype(np.float64) - init[n]
            big = np.abs(moved_o) >= 0.25 * np.abs(moved_o).max()                    # Adam's normalised step amplifies noise where the gradient is ~0
            assert big.any()
            err = np.abs(moved_h - moved_o)[big].max() / np.abs(moved_o).max()
            worst = max(worst, err)
            assert err < 1e-1, (n, err)                   # measured 4.7e-2 (Adam's sign-like step where a gradient is small); a wrong gradient is O(1)
        print("8 Adam steps: loss %.5f -> %.5f, max |loss_hip - loss_oracle| %.2e, worst parameter-movement error %.2e of the largest movement"
              % (losses[0][1], losses[-1][1], max(abs(a - b) for a, b in losses), worst))
    finally:
        E.DROPOUT_KEEP = keep
        E.DETERMINISTIC = False
        dg.reset()
