"""GPU parity of two corners of the reference's surface that the model path itself never takes:
  * FC_LAYERS = 0 -- dgcnn/model.py:88 calls ops.fc with repeat = 0 (ops.py:151: a loop of zero trips), so dropout and `Final`
    act on the 1024 + sum(2F+64) + 1024 channel concat itself (the tiled global feature is materialised only here);
  * ops.repeat_edge_conv with a LIST-valued k (ops.py:77-82): one k per EdgeConv layer (model.py:16 casts KVALUE to int, so only
    a direct caller of the operator reaches this form).
Both against the oracle on identical inputs: graphs bit-exact, floating point within the bar written at the assert."""
import numpy as np
import pytest
import torch

from oracle import dgcnn_oracle as O
from gpu_helpers import capture_layers, dev, host, run_model, set_vars

pytestmark = pytest.mark.gpu


@pytest.fixture()
def dg():
    import dgcnn
    dgcnn.reset()
    return dgcnn


@pytest.mark.parametrize("det", [False, True], ids=["atomics", "deterministic"])
def test_model_with_no_fc_layer(dg, det):
    from dgcnn import _engine as E
    B, N, C, k = 3, 256, 3, 8
    flags = dg.DGCNN_FLAGS(EDGE_CONV_LAYERS=2, EDGE_CONV_FILTERS=[32, 64], KVALUE=k, FC_LAYERS=0, FC_FILTERS=[], NUM_CLASS=2,
                           NUM_CHANNEL=C, TRAIN=False, DETERMINISTIC=det)
    rng = np.random.default_rng(5)
    pts = rng.random((B, N, C), dtype=np.float32)
    labels = rng.integers(0, 2, (B, N)).astype(np.int32)
    params = O.init_params(flags, C, seed=3)
    assert params["Final/weights"].shape == (1024 + (64 + 64) + (128 + 64) + 1024, 2) and not any(n.startswith("FC") for n in params)
    for n in params:
        if n.endswith("beta"):
            params[n] = rng.normal(0, 0.2, params[n].shape).astype(np.float32)
    try:
        # inference graph: every layer's k-NN graph bit-exact on the layer's own input, logits within 1e-3 (north_star)
        tv, res, cap = run_model(dg, flags, pts, params, train=False, labels=labels)
        assert set(tv.variables) == set(params)
        idx_list = []
        for i in range(2):
            xin, idx = cap["EdgeConv%d" % i]
            np.testing.assert_array_equal(idx, O.k_nn(xin, k))
            idx_list.append(idx)
        logits_ref, _ = O.model_forward(pts, flags, params, idx_list=idx_list)
        loss_ref, sm_ref, acc_ref, _ = O.softmax_xent(logits_ref, labels)
        dg.ctx().recording = False
        logits = host(dg.build(dev(pts), flags))
        np.testing.assert_allclose(logits, logits_ref, rtol=0, atol=1e-3)
        np.testing.assert_allclose(host(res[0]), sm_ref, rtol=0, atol=1e-3)
        assert abs(float(res[-1]) - float(loss_ref)) < 1e-3

        # training graph, dropout mask all ones (the reference's mask stream is its own): every gradient against the fp64 twin
        flags.TRAIN = True
        keep, E.DROPOUT_KEEP = E.DROPOUT_KEEP, 1.0
        try:
            tv, res, cap = run_model(dg, flags, pts, params, train=True, labels=labels)
        finally:
            E.DROPOUT_KEEP = keep
        idx_list = [cap["EdgeConv%d" % i][1] for i in range(2)]
        p64 = {n: v.astype(np.float64) for n, v in params.items()}
        G, loss64, _, _ = O.train_step_grads(pts.astype(np.float64), labels, flags, p64, idx_list=idx_list)
        assert abs(float(res[2]) - float(loss64)) < 1e-3
        for n in params:
            g = host(tv.gradients[n]).astype(np.float64)
            fro = np.linalg.norm(g - G[n]) / max(np.linalg.norm(G[n]), 1e-6)
            assert fro <= 5e-3, (n, fro)                                     # (the deterministic-mode bar of test_gpu_parity.py)

        # with the dropout on, the step runs, masks a different 30 % every step and one optimizer step goes through
        tv = dg.trainval(flags).initialize()
        set_vars(dg, params)
        tv.zero_gradients(None)
        r1 = tv.accum_gradient(None, [pts], [labels])
        g1 = host(dg.ctx().flat_grad).copy()
        tv.zero_gradients(None)
        r2 = tv.accum_gradient(None, [pts], [labels])
        g2 = host(dg.ctx().flat_grad).copy()
        assert np.isfinite(float(r1[2])) and np.isfinite(float(r2[2])) and np.abs(g1).sum() > 0
        assert np.linalg.norm(g1 - g2) > 1e-3 * np.linalg.norm(g1)            # another mask: another gradient
        tv.apply_gradient(None)
    finally:
        E.DETERMINISTIC = E.DETERMINISTIC_ENV_DEFAULT


def test_repeat_edge_conv_with_one_k_per_layer(dg):
    """ops.repeat_edge_conv(points, 3, k=[20, 10, 5], num_filters=[64, 64, 128]) called directly, as a user of the operator module
    would (ops.py:75-98): per-layer graphs bit-exact with the oracle's k_nn on that layer's input, the nine outputs within 1e-4 of
    the fp64 oracle fed those graphs, and the gradients of every variable of the stack against the oracle's backward."""
    from dgcnn import _engine as E
    B, N, C = 2, 384, 3
    kl, fl = [20, 10, 5], [64, 64, 128]
    rng = np.random.default_rng(9)
    pts = rng.random((B, N, C), dtype=np.float32)
    P = {}
    cin = C
    for i, f in enumerate(fl):
        s = "EdgeConv%d/" % i
        P[s + "conv0/weights"] = rng.normal(0, 0.4, (2 * cin, f)).astype(np.float32)
        P[s + "conv0/BatchNorm/beta"] = rng.normal(0, 0.2, f).astype(np.float32)
        P[s + "conv1/weights"] = rng.normal(0, 0.2, (2 * f, 64)).astype(np.float32)
        P[s + "conv1/BatchNorm/beta"] = rng.normal(0, 0.2, 64).astype(np.float32)
        cin = 64
    c = dg.ctx()
    c.begin_step()
    c.recording = True
    for n, v in P.items():
        c.get_variable(n, v.shape)
    set_vars(dg, P)
    with capture_layers() as cap:
        tensors = dg.ops.repeat_edge_conv(dev(pts), 3, kl, fl, True)
    assert len(tensors) == 9
    idx_list = []
    for i in range(3):
        xin, idx = cap.layers["EdgeConv%d" % i]
        assert idx.shape == (B, N, kl[i])
        np.testing.assert_array_equal(idx, O.k_nn(xin, kl[i]))                  # bit-exact on identical inputs
        idx_list.append(idx)
    p64 = {n: v.astype(np.float64) for n, v in P.items()}
    ref, layers = O.repeat_edge_conv(pts.astype(np.float64), 3, kl, fl, p64, idx_list=idx_list)
    for j, (a, b) in enumerate(zip(tensors, ref)):
        assert tuple(a.shape) == b.shape                                          # (B,N,1,ch): ops.py:73
        np.testing.assert_allclose(host(a), b, rtol=1e-4, atol=1e-4, err_msg="tensor %d" % j)

    # backward: a seeded upstream gradient on all nine outputs
    d = [rng.normal(size=r.shape) for r in ref]
    for t, g in zip(tensors, d):
        v, _, _ = E.as2d(t)
        c.grad(v).copy_(dev(g.reshape(B * N, -1).astype(np.float32)))
    c.backward()
    d_next = None
    for i in reversed(range(3)):
        d_net = d[3 * i + 2] if d_next is None else d[3 * i + 2] + d_next
        dx, g = O.edge_conv_bwd(d[3 * i], d[3 * i + 1], d_net, layers[i]["ec"])
        d_next = dx[:, :, None, :]
        s = "EdgeConv%d/" % i
        for leaf, key in (("conv0/weights", "W0"), ("conv0/BatchNorm/beta", "beta0"), ("conv1/weights", "W1"),
                          ("conv1/BatchNorm/beta", "beta1")):
            got = host(c.var_grads[s + leaf]).astype(np.float64)
            fro = np.linalg.norm(got - g[key]) / max(np.linalg.norm(g[key]), 1e-9)
            assert fro <= 2e-3, (s + leaf, fro)

    with pytest.raises(ValueError):
        dg.ops.repeat_edge_conv(dev(pts), 3, [20, 10], fl, True)                # ops.py:80-82
    with pytest.raises(ValueError):
        dg.ops.repeat_edge_conv(dev(pts), 3, kl, [64, 64], True)                # ops.py:84-87
