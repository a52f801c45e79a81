"""Parity at the sizes BASELINE.json names (run with -m gpu on an MI355X).

  configs[1]  B=24 N=2048 k=20 C=3, dgcnn 3 x (64,64,128) + FC(512,256): logits vs the fp32 oracle at FULL size,
              per-layer k-NN bit-exact on the HIP path's own layer inputs, and the END-TO-END neighbour-set
              mismatch rate of the dynamic graphs (layers 1-2) against the oracle run on its own features;
  configs[2]  B=8 N=16384 k=40 residual-dgcnn x6: architecture logits at N=2048, k_nn bit-exact at N=16384 k=40
              C=64, full-size training step with size-independent properties;
  configs[4]  per-GPU shard B=8 N=65536 k=20: k_nn bit-exact on a seeded row sample at N=65536 (C=3 and C=64),
              full-size training step with the same properties.
(configs[3] is configs[1] under data parallelism: tests/test_gpu_dp.py.)

The oracle is O(N^2 C) per cloud: full compares where that takes seconds, seeded row samples above.
"""
import os

import numpy as np
import pytest
import torch

from oracle import dgcnn_oracle as O
from gpu_helpers import dev, host, run_model, capture_layers, idx_set_mismatch, set_vars

pytestmark = pytest.mark.gpu


@pytest.fixture()
def dg():
    import dgcnn
    dgcnn.reset()
    yield dgcnn
    dgcnn.reset()
    torch.cuda.empty_cache()


def relu_features(rng, shape):
    """Post-ReLU-like feature rows (many exact zeros), the kind layers >= 1 build their graphs on."""
    return np.maximum(rng.normal(0, 1, shape), 0).astype(np.float32)


def check_knn_properties(cloud, idx, k, rows):
    """Size-independent properties of one cloud's neighbour lists on the query rows `rows` + bit-exactness
    against the C oracle on those rows (dgcnn/ops.py:8-19)."""
    N = cloud.shape[0]
    sel = idx[rows]
    assert sel.min() >= 0 and sel.max() < N
    assert (np.sort(sel, axis=1)[:, 1:] != np.sort(sel, axis=1)[:, :-1]).all()          # k distinct neighbours
    D = O.dist_pairs_f32(cloud, rows, sel)
    assert (np.diff(D, axis=1) >= 0).all()                                               # ascending (top_k of -D)
    ties = np.diff(D, axis=1) == 0
    assert (np.diff(sel, axis=1)[ties] > 0).all()                                        # ties -> lower index first
    np.testing.assert_array_equal(sel, O.k_nn_rows(cloud, k, rows))                      # bit-exact


# ------------------------------------------------------------------------------------------------------
# k_nn at the large N of configs[2] / configs[4]
# ------------------------------------------------------------------------------------------------------
def test_config2_knn_n16384_k40_c64_bit_exact(dg):
    """configs[2] layer >= 1 graph: one cloud, N=16384, k=40, C=64 relu-like features; ALL rows against the C oracle."""
    rng = np.random.default_rng(16384)
    x = relu_features(rng, (1, 16384, 64))
    idx = host(dg.ops.k_nn(dev(x), 40))
    np.testing.assert_array_equal(idx, O.k_nn(x, 40))
    assert ((idx[0] == np.arange(16384)[:, None]).sum(-1) == 1).all()                    # self among the k (ops.py:18)


def test_config2_knn_n16384_k40_c3_bit_exact(dg):
    """configs[2] layer 0 graph (raw coordinates)."""
    rng = np.random.default_rng(3)
    x = rng.random((2, 16384, 3), dtype=np.float32)
    np.testing.assert_array_equal(host(dg.ops.k_nn(dev(x), 40)), O.k_nn(x, 40))


@pytest.mark.parametrize("C,kind", [(3, "uniform"), (64, "relu"), (4, "integer")])
def test_config4_knn_n65536_k20_bit_exact(dg, C, kind):
    """configs[4] (LArTPC scale): one cloud of N=65536, k=20.  C=3: every row; C=64: a seeded sample of rows sized by
    the host's cores (4.3e9 pairs x 64 channels is minutes of scalar work); integer voxel coordinates
    (C=4, many exact ties and duplicate points): the tie rule at scale, every row."""
    N, k = 65536, 20
    rng = np.random.default_rng(65536 + C)
    if kind == "uniform":
        x = rng.random((1, N, C), dtype=np.float32)
    elif kind == "relu":
        x = relu_features(rng, (1, N, C))
    else:
        x = rng.integers(0, 24, (1, N, C)).astype(np.float32)
    idx = host(dg.ops.k_nn(dev(x), k))
    assert idx.min() >= 0 and idx.max() < N
    if C <= 4:
        np.testing.assert_array_equal(idx, O.k_nn(x, k))
    else:
        ncpu = os.cpu_count() or 1
        nrows = 16384 if ncpu >= 32 else 2048
        print("C=%d relu features at N=65536: %d of the 65536 query rows checked against the C oracle (%d host CPUs%s)"
              % (C, nrows, ncpu, "" if ncpu >= 32 else ": fewer than 32, the sample is cut from 16384 to keep the run in minutes"))
        rows = np.sort(rng.permutation(N)[:nrows]).astype(np.int32)
        check_knn_properties(x[0], idx[0], k, rows)
    if kind != "integer":
        assert ((idx[0] == np.arange(N)[:, None]).sum(-1) == 1).all()


# ------------------------------------------------------------------------------------------------------
# configs[1] at full size
# ------------------------------------------------------------------------------------------------------
# Bars of the DEFAULT mode (deterministic kernels: a run is ONE set of numbers) in the end-to-end comparison (measured in round 4,
# identical in three runs: see DESIGN.md 3); the opt-in atomics mode keeps the wider bars that cover its run-to-run spread
DET_LAYER_FACTOR = 1.1
DET_LOGIT_SHARE_SLACK = 0.02


def config1_flags(dg, train):
    return dg.DGCNN_FLAGS(MODEL_NAME="dgcnn", EDGE_CONV_LAYERS=3, EDGE_CONV_FILTERS=[64, 64, 128], FC_LAYERS=2,
                          FC_FILTERS=[512, 256], NUM_CLASS=2, KVALUE=20, NUM_CHANNEL=3, TRAIN=train)


def _forward_with_graphs(dg, flags, pts, params, L):
    """One inference forward of the HIP path; logits and every layer's (input, idx) from the SAME forward (the BatchNorm
    sums are accumulated with atomics: two runs differ in the last bits of the layer >= 1 features and with them in a
    few near-tie neighbour lists), each layer's k-NN checked bit-exact against the C oracle on that input."""
    dg.trainval(flags).initialize()
    set_vars(dg, params)
    dg.ctx().begin_step()
    with capture_layers() as capl:
        logits = host(dg.build(dev(pts), flags))
    idx_list = []
    for i in range(L):
        xin, idx = capl.layers["EdgeConv%d" % i]
        np.testing.assert_array_equal(idx, O.k_nn(xin, int(flags.KVALUE)), err_msg="layer %d" % i)   # bit-exact on identical inputs
        idx_list.append(idx)
    return logits, idx_list


def _oracle_three_way(pts, flags, params, idx_list, logits, what):
    """Logits of the HIP path against the oracle fed the same graphs, in float64 (the exact-arithmetic anchor) and in
    float32 (the op-for-op restatement, whose own accumulation error is what `reference fp32` is worth at this size)."""
    p64 = {n: v.astype(np.float64) for n, v in params.items()}
    ref64, _ = O.model_forward(pts.astype(np.float64), flags, p64, idx_list=idx_list)
    ref32, _ = O.model_forward(pts, flags, params, idx_list=idx_list)
    e_hip, e_o32, e_h32 = np.abs(logits - ref64), np.abs(ref32 - ref64), np.abs(logits - ref32)
    print("%s, same graphs: logits max|diff|  HIP vs fp64 twin %.3e (mean %.1e) | fp32 oracle vs fp64 twin %.3e (mean %.1e) | "
          "HIP vs fp32 oracle %.3e" % (what, e_hip.max(), e_hip.mean(), e_o32.max(), e_o32.mean(), e_h32.max()))
    np.testing.assert_allclose(logits, ref64, rtol=0, atol=1e-3)                     # north_star bar, against exact arithmetic
    # ... and, plainly, against the float32 restatement of the reference (the "reference fp32 CPU path" stand-in): with its
    # BatchNorm-axis reductions in a defined tree order (oracle.tree_colsum) it sits 2e-5 from the twin at this size
    # (profiles/r03/oracle_reduction.txt; round 2's running float32 sums: 1.3e-3 .. 1.9e-3)
    np.testing.assert_allclose(logits, ref32, rtol=0, atol=1e-3)
    assert e_o32.max() <= 2e-4, e_o32.max()
    return ref64


def _end_to_end_rates(pts, flags, params, idx_list, logits, L):
    """The oracle builds its OWN dynamic graphs from its own features (float64 twin and float32): per layer, the
    fraction of rows whose neighbour SET differs from the HIP path's."""
    p64 = {n: v.astype(np.float64) for n, v in params.items()}
    ref64, c64 = O.model_forward(pts.astype(np.float64), flags, p64)
    i64 = [c64["layers"][i]["ec"]["idx"] for i in range(L)]
    del c64
    ref32, c32 = O.model_forward(pts, flags, params)
    i32 = [c32["layers"][i]["ec"]["idx"] for i in range(L)]
    del c32
    hip = [idx_set_mismatch(idx_list[i], i64[i])[0] for i in range(L)]
    o32 = [idx_set_mismatch(i32[i], i64[i])[0] for i in range(L)]
    fmt = lambda r: " ".join("%.4f%%" % (100 * v) for v in r)
    err = np.abs(logits - ref64)
    err32 = np.abs(ref32 - ref64)
    print("end to end (each side builds its own graphs), rows per layer whose neighbour set differs from the fp64 twin's: "
          "HIP [%s] | fp32 oracle [%s]; logits within 1e-3 of the twin: HIP %.3f%% (max %.2e) | fp32 oracle %.3f%% (max %.2e)"
          % (fmt(hip), fmt(o32), 100 * (err <= 1e-3).mean(), err.max(), 100 * (err32 <= 1e-3).mean(), err32.max()))
    # layer 0 sees identical inputs on both fp32 sides: the HIP graph must equal the fp32 oracle's exactly (the fp64
    # twin's layer-0 graph may differ on near-tie rows: different distance arithmetic)
    np.testing.assert_array_equal(idx_list[0], i32[0])
    return hip, o32, float((err <= 1e-3).mean()), float((err32 <= 1e-3).mean())


@pytest.mark.parametrize("det", [True, False], ids=["default", "atomics"])       # default = the deterministic kernels (round 5)
def test_config1_full_size_logits_and_dynamic_graphs(dg, det):
    """(B,N,k,C) = (24,2048,20,3), the headline configuration, inference graph (default kernels, and DETERMINISTIC mode, where the
    run is ONE set of numbers and the end-to-end bars are relative to the fp32 oracle's own figures instead of fixed floors):
      * every layer's k-NN bit-exact against the C oracle on the layer's actual input (all 24 x 2048 rows);
      * logits within 1e-3 (north_star) of the oracle fed the same neighbour graphs;
      * END TO END (the oracle builds its own graphs from its own features): fraction of rows whose neighbour set
        differs in layers 1-2 and the logits deviation that follows -- printed, and bounded by what the fp32 numpy
        restatement of the reference itself does against the float64 twin (a dynamic graph amplifies ANY last-bit
        difference of the features into different neighbour lists for near-tie rows)."""
    B, N, C = 24, 2048, 3
    flags = config1_flags(dg, train=False)
    flags.DETERMINISTIC = None if det else False          # None: the library default (deterministic); False: the atomics mode
    rng = np.random.default_rng(0)
    pts = rng.random((B, N, C), dtype=np.float32)
    params = O.init_params(flags, C, seed=1)
    for n in params:
        if n.endswith("beta"):
            params[n] = rng.normal(0, 0.2, params[n].shape).astype(np.float32)
    try:
        logits, idx_list = _forward_with_graphs(dg, flags, pts, params, 3)
    finally:
        from dgcnn import _engine as E
        E.DETERMINISTIC = E.DETERMINISTIC_ENV_DEFAULT
    assert logits.shape == (B, N, 2)
    _oracle_three_way(pts, flags, params, idx_list, logits, "configs[1] full size" + ("" if det else " [atomics]"))
    hip, o32, w_hip, w_o32 = _end_to_end_rates(pts, flags, params, idx_list, logits, 3)
    if det:
        # per layer: rows whose neighbour set differs from the float64 twin's, against the same figure of the fp32 oracle;
        # end to end: the share of logits within 1e-3 of the twin, against the fp32 oracle's share
        for i in range(3):
            assert hip[i] <= DET_LAYER_FACTOR * o32[i] + 1e-3, (i, hip, o32)
        assert w_hip >= w_o32 - DET_LOGIT_SHARE_SLACK, (w_hip, w_o32)
        return
    assert max(hip) <= 1.5 * max(o32) + 2e-3, (hip, o32)    # no worse than a reference-grade fp32 evaluation
    # end to end no fp32 evaluation stays within 1e-3 of another everywhere: a row whose neighbour set differs changes its
    # own logits at O(0.1) and, through the global max-pool feature and the BatchNorm statistics, nudges its whole cloud
    # (measured: HIP 81 .. 90 % of the logits within 1e-3 of the twin run to run; the fp32 numpy restatement 62 % with
    # running float32 BatchNorm sums (round 2), 91 % with the tree-ordered sums it uses now)
    assert w_hip >= w_o32 - 0.12 and w_hip > 0.6, (w_hip, w_o32)


# relative Frobenius vs the float64 twin.  Measured at full size over seven runs of round 3 (the neighbour graphs, and with them which
# ReLU / max-over-k decisions sit on a last-bit difference, change run to run with the atomically summed BatchNorm statistics):
# HIP worst tensor 1.7e-3, 2.0e-3, 2.3e-3, 2.4e-3, 3.9e-3, 3.9e-3, 6.6e-3; the fp32 oracle on the same graphs 0.8e-3 .. 4.2e-3.
# 5e-3 (round 2) failed one run in seven; a wrong or missing term is O(1).
GRAD_BAR = 1e-2            # atomics mode (opt-in)
# default (deterministic) mode: the HIP run is bit-reproducible, so the figure is ONE number per seed (round 3, this seed, every run: HIP
# worst tensor 1.47e-3, fp32 oracle on the same graphs 1.43e-3, HIP vs fp32 oracle 1.54e-3) and the bar can sit at twice that
GRAD_BAR_DET = 3e-3


@pytest.mark.parametrize("det", [True, False], ids=["default", "atomics"])
def test_config1_full_size_training_step(dg, det):
    """One full training micro-step at (24,2048,20,3) with dropout off, the HIP graphs fed to the oracle: loss within 1e-4
    and EVERY gradient tensor within 1e-2 (relative Frobenius; measured 1.7e-3 .. 6.6e-3) of the float64 twin; so is the float32
    oracle (tree-ordered BatchNorm sums: 0.8e-3 .. 4.2e-3; round 2's running float32 sums over 983040 rows: 2e-2 .. 4e-2), and the
    two float32 evaluations agree within 2e-2 -- then the Adam step."""
    B, N, C = 24, 2048, 3
    flags = config1_flags(dg, train=True)
    flags.DETERMINISTIC = None if det else False          # None: the library default (deterministic); False: the atomics mode
    bar = GRAD_BAR_DET if det else GRAD_BAR
    rng = np.random.default_rng(1)
    pts = rng.random((B, N, C), dtype=np.float32)
    labels = rng.integers(0, 2, (B, N)).astype(np.int32)
    params = O.init_params(flags, C, seed=1)
    from dgcnn import _engine as E
    keep, E.DROPOUT_KEEP = E.DROPOUT_KEEP, 1.0
    try:
        tv, res, cap = run_model(dg, flags, pts, params, train=True, labels=labels)
    finally:
        E.DROPOUT_KEEP = keep
        assert E.DETERMINISTIC is bool(det)                  # (the default resolved to the deterministic kernels)
        E.DETERMINISTIC = E.DETERMINISTIC_ENV_DEFAULT
    idx_list = [cap["EdgeConv%d" % i][1] for i in range(3)]
    p64 = {n: v.astype(np.float64) for n, v in params.items()}
    G64, loss64, _, _ = O.train_step_grads(pts.astype(np.float64), labels, flags, p64, idx_list=idx_list)
    G32, loss32, _, _ = O.train_step_grads(pts, labels, flags, params, idx_list=idx_list)
    assert abs(float(res[2]) - float(loss64)) < 1e-4
    rel = lambda a, b: np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
    r_hip = {n: rel(host(tv.gradients[n]).astype(np.float64), G64[n]) for n in params}
    r_o32 = {n: rel(G32[n].astype(np.float64), G64[n]) for n in params}
    print("configs[1] full size, relative Frobenius gradient error vs the fp64 twin: HIP worst %.2e (%s) | fp32 oracle worst %.2e"
          % (max(r_hip.values()), max(r_hip, key=r_hip.get), max(r_o32.values())))
    assert max(r_hip.values()) < bar, r_hip
    assert max(r_o32.values()) < bar, r_o32               # the fp32 oracle is a credible target itself (round 2: 2e-2 .. 4e-2)
    r_h32 = {n: rel(host(tv.gradients[n]).astype(np.float64), G32[n].astype(np.float64)) for n in params}
    print("HIP vs fp32 oracle: worst %.2e (%s)" % (max(r_h32.values()), max(r_h32, key=r_h32.get)))
    assert max(r_h32.values()) < 2 * bar, r_h32           # two fp32 evaluations, each within the bar of exact arithmetic
    before = host(dg.ctx().flat_param).copy()
    tv.apply_gradient(None)
    after = host(dg.ctx().flat_param)
    step = np.abs(after - before)
    assert np.isfinite(after).all() and step.max() <= 1.001e-3 and np.median(step) > 5e-4   # first Adam step ~= lr


# ------------------------------------------------------------------------------------------------------
# configs[2]: residual-dgcnn x 6, k = 40
# ------------------------------------------------------------------------------------------------------
def config2_flags(dg, train):
    return dg.DGCNN_FLAGS(MODEL_NAME="residual-dgcnn", EDGE_CONV_LAYERS=6, EDGE_CONV_FILTERS=64, FC_LAYERS=2,
                          FC_FILTERS=[512, 256], NUM_CLASS=2, KVALUE=40, NUM_CHANNEL=3, TRAIN=train)


@pytest.mark.parametrize("det", [True, False], ids=["default", "atomics"])       # default = the deterministic kernels (round 5)
def test_config2_architecture_logits_n2048(dg, det):
    """configs[2] architecture (scripts/lsf/train_dgcnn.sh:8-9: residual-dgcnn, 6 layers x 64, k=40) at B=2, N=2048:
    per-layer bit-exact graphs, logits within 1e-3 of the oracle fed the same graphs; end to end the six stacked dynamic
    graphs diverge layer by layer for ANY fp32 evaluation -- the HIP path's rates are printed next to the fp32 numpy
    restatement's and bounded by them."""
    B, N, C, L = 2, 2048, 3, 6
    flags = config2_flags(dg, train=False)
    flags.DETERMINISTIC = None if det else False          # None: the library default (deterministic); False: the atomics mode
    rng = np.random.default_rng(2)
    pts = rng.random((B, N, C), dtype=np.float32)
    params = O.init_params(flags, C, seed=3)
    for n in params:
        if n.endswith("beta"):
            params[n] = rng.normal(0, 0.2, params[n].shape).astype(np.float32)
    try:
        logits, idx_list = _forward_with_graphs(dg, flags, pts, params, L)
    finally:
        from dgcnn import _engine as E
        E.DETERMINISTIC = E.DETERMINISTIC_ENV_DEFAULT
    _oracle_three_way(pts, flags, params, idx_list, logits, "configs[2] architecture at N=2048" + ("" if det else " [atomics]"))
    hip, o32, w_hip, w_o32 = _end_to_end_rates(pts, flags, params, idx_list, logits, L)
    for i in range(1, L):
        if det:
            assert hip[i] <= DET_LAYER_FACTOR * o32[i] + 1e-3, (i, hip, o32)
        else:
            assert hip[i] <= 1.5 * o32[i] + 5e-3, (i, hip, o32)
    if det:
        assert w_hip >= w_o32 - DET_LOGIT_SHARE_SLACK, (w_hip, w_o32)


def _full_size_property_run(dg, flags, B, N, k, L, nrows, seed):
    rng = np.random.default_rng(seed)
    pts = rng.random((B, N, 3), dtype=np.float32)
    labels = rng.integers(0, 2, (B, N)).astype(np.int32)
    tv = dg.trainval(flags).initialize()
    tv.zero_gradients(None)
    with capture_layers() as cap:
        res = tv.accum_gradient(None, [dev(pts)], [dev(labels)])
    loss, acc = float(res[2]), float(res[1])
    assert np.isfinite(loss) and 0.3 < loss < 3.0 and 0.0 <= acc <= 1.0, (loss, acc)   # 2 classes, random init: ~ln 2
    g = host(dg.ctx().flat_grad)
    assert np.isfinite(g).all() and np.abs(g).sum() > 0
    for name in tv.gradients:                                    # every variable received a gradient
        assert float(tv.gradients[name].abs().sum()) > 0 or name.endswith("conv1/BatchNorm/beta"), name
    for i in range(L):
        xin, idx = cap.layers["EdgeConv%d" % i]
        assert idx.shape == (B, N, k)
        assert ((idx == np.arange(N)[None, :, None]).sum(-1) == 1).all(), "layer %d: self not among the k" % i
        for b in (0, B - 1):
            rows = np.sort(np.random.default_rng(seed + 10 * i + b).permutation(N)[:nrows]).astype(np.int32)
            check_knn_properties(xin[b], idx[b], k, rows)
    tv.apply_gradient(None)
    assert np.isfinite(host(dg.ctx().flat_param)).all()
    return loss


def test_config2_full_size_property_run(dg):
    """configs[2] at FULL size on one GPU: B=8, N=16384, k=40, residual-dgcnn x6, one training step.  Size-independent
    properties: finite loss near ln 2, finite non-zero gradients, every layer's lists contain self, are ascending with
    the lower-index tie rule, and are bit-exact against the C oracle on a seeded sample of rows of two clouds."""
    loss = _full_size_property_run(dg, config2_flags(dg, train=True), B=8, N=16384, k=40, L=6, nrows=512, seed=42)
    print("configs[2] full size: loss %.5f, peak HBM %.1f GB" % (loss, torch.cuda.max_memory_allocated() / 2 ** 30))


def test_config4_full_size_property_run(dg):
    """configs[4] per-GPU shard at FULL size: B=8, N=65536, k=20, dgcnn 3 x (64,64,128): the pairwise distances are
    tiled (a 65536^2 matrix would be 17 GB per cloud), same properties."""
    flags = dg.DGCNN_FLAGS(MODEL_NAME="dgcnn", EDGE_CONV_LAYERS=3, EDGE_CONV_FILTERS=[64, 64, 128], FC_LAYERS=2,
                           FC_FILTERS=[512, 256], NUM_CLASS=2, KVALUE=20, NUM_CHANNEL=3, TRAIN=True)
    torch.cuda.reset_peak_memory_stats()
    loss = _full_size_property_run(dg, flags, B=8, N=65536, k=20, L=3, nrows=256, seed=65)
    print("configs[4] shard full size: loss %.5f, peak HBM %.1f GB" % (loss, torch.cuda.max_memory_allocated() / 2 ** 30))
