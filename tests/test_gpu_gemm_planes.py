"""The plane GEMM (csrc/gemm_pl.hip): fp32-class products from operands pre-split into two fp16 planes (DGCNN_PLANES_F16X2, the
opt-in HEAD_PLANES='f16' mode), against float64 numpy and against the in-kernel-split GEMM (dgcnn_gemm_f32, arithmetic 6)."""
import numpy as np
import pytest
import torch

from dgcnn import _hip as H, _planes as P, _engine as E

pytestmark = pytest.mark.gpu


def _rand(shape, seed, scale=1.0):
    rng = np.random.default_rng(seed)
    return (rng.standard_normal(shape) * scale).astype(np.float32)


def _planes_to_host(ps):
    """Reassemble the fp32 values a plane set represents (sum of its planes), on the host."""
    npl = P.NPLANES[ps.fmt]
    raw = ps.buf.cpu().numpy().reshape(npl, -1)
    out = np.zeros((ps.rows, ps.cols), np.float64)
    for p in range(npl):
        a = raw[p].view(np.uint16).reshape(-1, ps.ra, 8)
        v = a.view(np.float16).astype(np.float32)
        o0 = ps.c0 // 8
        blk = v[o0:o0 + ps.cols // 8]                         # (noct, ra, 8)
        out += blk.transpose(1, 0, 2).reshape(ps.ra, -1)[:ps.rows].astype(np.float64)
    return out / (1.0 if ps.scale is None else float(ps.scale))


@pytest.mark.parametrize("rows,cols", [(64, 8), (100, 64), (1000, 200), (513, 1728)])
def test_split_pads_with_zeros_and_takes_strided_views(rows, cols):
    x = _rand((rows, cols), 1)
    ps = P.from_f32(torch.from_numpy(x).cuda())
    np.testing.assert_allclose(_planes_to_host(ps), x.astype(np.float64), rtol=2.0 ** -21, atol=2.0 ** -24)    # 22 bits per element
    raw = ps.buf.cpu().numpy().reshape(2, cols // 8, ps.ra, 16)
    assert not raw[:, :, rows:, :].any()                                                 # pad rows are zero
    wide = torch.from_numpy(_rand((rows, cols + 24), 3)).cuda()                          # a strided source view
    ps2 = P.from_f32(wide[:, 8:8 + cols])
    np.testing.assert_allclose(_planes_to_host(ps2), wide[:, 8:8 + cols].cpu().numpy().astype(np.float64), rtol=2.0 ** -21, atol=2.0 ** -24)


def test_split_of_the_transposed_orientation():
    w = _rand((192, 1024), 5, 0.05)
    pt = P.from_f32(torch.from_numpy(w).cuda(), transpose=True)                          # rows = 1024 outputs, cols = 192 inputs
    assert (pt.rows, pt.cols) == (1024, 192)
    np.testing.assert_allclose(_planes_to_host(pt), w.T.astype(np.float64), rtol=2.0 ** -21, atol=2.0 ** -26)


def _err(C, ref, A, B):
    """max |C - ref| relative to sum_k |a||b| (the scale of the rounding errors of the product)."""
    return float((np.abs(C - ref) / np.maximum(np.abs(A) @ np.abs(B), 1e-30)).max())


@pytest.mark.parametrize("M,N,K", [(256, 128, 32), (300, 130, 64), (1024, 512, 1728), (4096, 192, 1024), (777, 1000, 96)])
def test_kc_product_matches_float64(M, N, K):
    """forward / dgrad form: C = X W with X (M, K) as planes and W^T (N, K) as planes."""
    X, W = _rand((M, K), 10), _rand((K, N), 11, 0.1)
    Xp = P.from_f32(torch.from_numpy(X).cuda())
    Wp = P.from_f32(torch.from_numpy(W).cuda(), transpose=True)
    C = torch.full((M, N), float("nan"), device="cuda")
    P.gemm(P.KC, Xp, Wp, C)
    ref = X.astype(np.float64) @ W.astype(np.float64)
    e = _err(C.cpu().numpy().astype(np.float64), ref, X.astype(np.float64), W.astype(np.float64))
    assert e < 6e-7, e                                        # fp32 class (2^-22 per operand: worst case 2 x 2.4e-7 + the dropped h2 h2 term)
    # against the in-kernel-split GEMM (exact 3-way bf16 split, 6 partial products)
    C2 = torch.empty((M, N), device="cuda")
    E.gemm(torch.from_numpy(X).cuda(), torch.from_numpy(W).cuda(), C2)
    assert _err(C.cpu().numpy().astype(np.float64), C2.cpu().numpy().astype(np.float64), X.astype(np.float64), W.astype(np.float64)) < 6e-7


def test_kc_epilogue_options_beta_bias_stats_and_channel_views():
    M, N, K = 2048, 256, 128
    big = _rand((M, 64 + K + 32), 20)
    X = big[:, 64:64 + K]
    W = _rand((K, N), 21, 0.1)
    C0 = _rand((M, N), 22)
    gb = _rand((8, N), 23)
    allp = P.from_f32(torch.from_numpy(big).cuda())
    Xp = allp.cols_view(64, 64 + K)                           # a channel range of a wider plane set
    Wp = P.from_f32(torch.from_numpy(W).cuda(), transpose=True)
    C = torch.from_numpy(C0.copy()).cuda()
    st = torch.zeros(H.STAT_SLOTS * 2 * N, dtype=torch.float64, device="cuda")
    P.gemm(P.KC, Xp, Wp, C, beta=1.0, gbias=torch.from_numpy(gb).cuda(), rpg=M // 8, stats=st)
    ref = C0 + X.astype(np.float64) @ W.astype(np.float64) + np.repeat(gb, M // 8, axis=0)
    got = C.cpu().numpy().astype(np.float64)
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-5)
    s = st.cpu().numpy().reshape(H.STAT_SLOTS, 2, N).sum(0)
    np.testing.assert_allclose(s[0], got.sum(0), rtol=1e-6, atol=1e-3)
    np.testing.assert_allclose(s[1], (got * got).sum(0), rtol=1e-6, atol=1e-3)


@pytest.mark.parametrize("M,N,K", [(256, 128, 64), (192, 1024, 5000), (1728, 512, 49152), (64, 96, 1000), (512, 256, 12345)])
def test_tr_product_matches_float64(M, N, K):
    """weight-gradient form: dW = X^T dY with X (K, M) and dY (K, N) as planes, reduction over the rows (split-K)."""
    X, dY = _rand((K, M), 30), _rand((K, N), 31, 1e-3)
    Xp = P.from_f32(torch.from_numpy(X).cuda())
    Yp = P.from_f32(torch.from_numpy(dY).cuda())
    ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    C0 = _rand((M, N), 32, 1e-2)
    C = torch.from_numpy(C0.copy()).cuda()
    P.gemm(P.TR, Xp, Yp, C, beta=1.0, ws=ws)
    Xd, Yd = X.astype(np.float64), dY.astype(np.float64)
    ref = C0 + Xd.T @ Yd
    e = float((np.abs(C.cpu().numpy() - ref) / np.maximum(np.abs(Xd).T @ np.abs(Yd), 1e-30)).max())
    assert e < 6e-7, e


# ------------------------------------------------------------------------------------------------------
# DGCNN_PLANES_F16X2: x * 2^e = h1 + h2 (two fp16 terms), three partial products
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("mag", [1.0, 1e-6, 3e4])
def test_f16x2_split_keeps_22_bits_relative_to_the_tensor_scale(mag):
    rows, cols = 777, 256
    x = (_rand((rows, cols), 40) * np.exp(_rand((rows, cols), 41) * 3) * mag).astype(np.float32)     # ~7 decades inside the tensor
    ps = P.from_f32(torch.from_numpy(x).cuda(), P.F16X2)
    sc = float(ps.scale)
    mx = np.abs(x).max()
    assert sc == 2.0 ** np.round(np.log2(sc)) and 2.0 ** 14 <= mx * sc < 2.0 ** 15               # a power of two; max lands in [2^14, 2^15)
    err = np.abs(_planes_to_host(ps) - x.astype(np.float64))
    # 22 significant bits per element, and never worse than 2^-25 in scaled units (h2 subnormal) = 2^-40 of the tensor's max
    bound = np.maximum(np.abs(x).astype(np.float64) * 2.0 ** -22, 2.0 ** -25 / sc)
    assert (err <= bound).all(), float((err / bound).max())
    raw = ps.buf.cpu().numpy().reshape(2, cols // 8, ps.ra, 16)
    assert not raw[:, :, rows:, :].any()


@pytest.mark.parametrize("M,N,K,xs,ws", [(1024, 512, 1728, 1.0, 0.05), (4096, 192, 1024, 1e-5, 0.05), (300, 130, 64, 50.0, 1e-3)])
def test_f16x2_kc_product_is_fp32_class(M, N, K, xs, ws):
    X, W = _rand((M, K), 50, xs), _rand((K, N), 51, ws)
    Xd, Wd = X.astype(np.float64), W.astype(np.float64)
    ref = Xd @ Wd
    C = torch.full((M, N), float("nan"), device="cuda")
    P.gemm(P.KC, P.from_f32(torch.from_numpy(X).cuda(), P.F16X2), P.from_f32(torch.from_numpy(W).cuda(), P.F16X2, transpose=True), C)
    e16 = _err(C.cpu().numpy().astype(np.float64), ref, Xd, Wd)
    H.set_gemm_arith(0)                                        # native fp32 MFMA (an fmaf chain) on the same operands
    try:
        C0 = torch.empty((M, N), device="cuda")
        E.gemm(torch.from_numpy(X).cuda(), torch.from_numpy(W).cuda(), C0)
    finally:
        H.set_gemm_arith(6)
    e32 = _err(C0.cpu().numpy().astype(np.float64), ref, Xd, Wd)
    rms = lambda c: float(np.sqrt(np.mean(((c - ref) / np.maximum(np.abs(Xd) @ np.abs(Wd), 1e-30)) ** 2)))
    print("K=%d: max err / sum|a||b|: f16x2 planes %.2e, fp32 MFMA chain %.2e; rms %.2e vs %.2e"
          % (K, e16, e32, rms(C.cpu().numpy().astype(np.float64)), rms(C0.cpu().numpy().astype(np.float64))))
    assert e16 < 6e-7, e16                                     # 2^-22 per operand: worst case 2 x 2.4e-7 (+ the dropped h2 h2 term)


def test_f16x2_tr_product_with_gradient_sized_operands():
    K, M, N = 20000, 512, 256
    X, dY = _rand((K, M), 60), (_rand((K, N), 61) * np.exp(_rand((K, N), 62) * 2) * 1e-6).astype(np.float32)
    ws = torch.empty(256 << 20, dtype=torch.uint8, device="cuda")
    C = torch.zeros((M, N), device="cuda")
    P.gemm(P.TR, P.from_f32(torch.from_numpy(X).cuda(), P.F16X2), P.from_f32(torch.from_numpy(dY).cuda(), P.F16X2), C, ws=ws)
    Xd, Yd = X.astype(np.float64), dY.astype(np.float64)
    e = float((np.abs(C.cpu().numpy() - Xd.T @ Yd) / np.maximum(np.abs(Xd).T @ np.abs(Yd), 1e-30)).max())
    assert e < 6e-7, e
