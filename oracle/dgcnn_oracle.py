"""
oracle/dgcnn_oracle.py -- TEST INFRASTRUCTURE ONLY.

CPU (numpy) restatement of the EdgeConv hot path of DeepLearnPhysics/dynamic-gcnn, used as the
checker for the HIP path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this module; the product package (dynamic-gcnn_amd/dgcnn) never does.

PARITY UNPINNED.  The reference's arithmetic lives in TensorFlow 1.x + tf.contrib.slim
(README.md:6, dgcnn/ops.py:5-7) which is neither vendored under /root/reference nor
installable here, and the reference ships no tests / golden vectors.
Every function below restates the reference's *graph* line by line (citations are
file:line into /root/reference) with the TF1 library semantics of SURVEY.md Appendix A;
the k-NN arithmetic order is made normative in oracle/knn_oracle.c.

All functions are dtype-generic: run them in float32 (the oracle proper) or float64
(the "twin" used to set tolerances).  Backward passes are hand-written and are themselves
checked against torch-CPU autograd in tests/test_oracle.py.
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
BN_EPS = 1e-3  # slim.batch_norm default epsilon [TF1-lib], SURVEY Appendix A.3


# ----------------------------------------------------------------------------------------
# C restatement (exact fp32 k-NN arithmetic)
# ----------------------------------------------------------------------------------------
def build_c(force: bool = False, fma: bool = True) -> str:
    """Compile oracle/knn_oracle.c -> oracle/_build/liboracle.so (gcc, -ffp-contract=off); fma=False: the -DORACLE_NO_FMA variant
    (liboracle_nofma.so: p_ij without fused multiply-add -- the chain SURVEY A.1 did NOT choose)."""
    out_dir = os.path.join(_HERE, "_build")
    so = os.path.join(out_dir, "liboracle.so" if fma else "liboracle_nofma.so")
    src = os.path.join(_HERE, "knn_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        os.makedirs(out_dir, exist_ok=True)
        subprocess.check_call(
            ["gcc", "-O2", "-fopenmp", "-ffp-contract=off", "-fno-fast-math"] + ([] if fma else ["-DORACLE_NO_FMA"])
            + ["-shared", "-fPIC", "-o", so, src, "-lm"]
        )
    return so


_LIB_NOFMA = None
KNN_FMA = True        # the normative chain; knn_chain(fma=False) switches the float32 k_nn to the non-FMA variant (sensitivity tests)


class knn_chain(object):
    """with knn_chain(fma=False): ...  -- every float32 k_nn inside (also those of model_forward) uses the non-FMA p_ij chain."""
    def __init__(self, fma):
        self.fma = bool(fma)

    def __enter__(self):
        global KNN_FMA
        self.prev, KNN_FMA = KNN_FMA, self.fma
        return self

    def __exit__(self, *exc):
        global KNN_FMA
        KNN_FMA = self.prev


def _lib():
    global _LIB, _LIB_NOFMA
    if not KNN_FMA:
        if _LIB_NOFMA is None:
            _LIB_NOFMA = _bind(ctypes.CDLL(build_c(fma=False)))
        return _LIB_NOFMA
    if _LIB is None:
        _LIB = _bind(ctypes.CDLL(build_c()))
    return _LIB


def _bind(lib):
    lib.oracle_knn_f32.restype = ctypes.c_int
    lib.oracle_knn_f32.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                    ctypes.c_long, ctypes.c_int, ctypes.c_void_p]
    lib.oracle_knn_rows_f32.restype = ctypes.c_int
    lib.oracle_knn_rows_f32.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_long, ctypes.c_int,
                                         ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    lib.oracle_dist_pairs_f32.restype = None
    lib.oracle_dist_pairs_f32.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_long, ctypes.c_void_p,
                                           ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    lib.oracle_dist_f32.restype = None
    lib.oracle_dist_f32.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_long,
                                     ctypes.c_void_p]
    lib.oracle_edges_f32.restype = None
    lib.oracle_edges_f32.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    return lib


def dist_matrix_f32(x: np.ndarray) -> np.ndarray:
    """(N,C) float32 -> D (N,N) with the normative arithmetic (dgcnn/ops.py:12-16)."""
    x = np.ascontiguousarray(x, np.float32)
    N, C = x.shape
    D = np.empty((N, N), np.float32)
    _lib().oracle_dist_f32(x.ctypes.data, N, C, C, D.ctypes.data)
    return D


# ----------------------------------------------------------------------------------------
# dgcnn/ops.py:8-19  k_nn
# ----------------------------------------------------------------------------------------
def k_nn(points: np.ndarray, k: int) -> np.ndarray:
    """idx (B,N,k) int32: k smallest D_ij per row incl. self, ascending, ties -> lower index.

    float32 input -> exact C restatement (bit-defined).  Any other dtype -> numpy in that dtype
    with a stable sort (same tie rule; used by the float64 twin)."""
    B, N, C = points.shape
    if k > N:
        raise ValueError("k_nn: k=%d > N=%d (tf.nn.top_k raises InvalidArgument)" % (k, N))
    if points.dtype == np.float32:
        x = np.ascontiguousarray(points)
        idx = np.empty((B, N, k), np.int32)
        rc = _lib().oracle_knn_f32(x.ctypes.data, B, N, C, C, k, idx.ctypes.data)
        if rc != 0:
            raise ValueError("oracle_knn_f32 failed")
        return idx
    M = points
    inner = np.matmul(M, np.transpose(M, (0, 2, 1)))                    # ops.py:12-13
    squared = np.sum(np.square(M), axis=-1, keepdims=True)              # ops.py:14
    nn_dist = squared + np.transpose(squared, (0, 2, 1)) - 2 * inner    # ops.py:15-16
    order = np.argsort(nn_dist, axis=-1, kind="stable")                 # ops.py:18 (top_k of -D)
    return order[..., :k].astype(np.int32)


def k_nn_rows(cloud: np.ndarray, k: int, rows: np.ndarray) -> np.ndarray:
    """k_nn of ONE cloud (N,C) float32 restricted to the query rows `rows` -> (len(rows), k) int32.
    Same C row routine as k_nn (bit-defined); for the N = 16384 / 65536 tests, where a seeded sample of
    rows is checked instead of all 4.3e9 pairs of a cloud."""
    x = np.ascontiguousarray(cloud, np.float32)
    N, C = x.shape
    rows = np.ascontiguousarray(rows, np.int32)
    out = np.empty((len(rows), k), np.int32)
    if _lib().oracle_knn_rows_f32(x.ctypes.data, N, C, C, k, rows.ctypes.data, len(rows), out.ctypes.data) != 0:
        raise ValueError("k_nn: k=%d > N=%d (tf.nn.top_k raises InvalidArgument)" % (k, N))
    return out


def dist_pairs_f32(cloud: np.ndarray, rows: np.ndarray, cand: np.ndarray) -> np.ndarray:
    """D[r][m] between query row rows[r] and candidate cand[r][m] of one cloud, normative arithmetic."""
    x = np.ascontiguousarray(cloud, np.float32)
    N, C = x.shape
    rows = np.ascontiguousarray(rows, np.int32)
    cand = np.ascontiguousarray(cand, np.int32)
    D = np.empty(cand.shape, np.float32)
    _lib().oracle_dist_pairs_f32(x.ctypes.data, N, C, C, rows.ctypes.data, len(rows), cand.ctypes.data, cand.shape[1],
                                 D.ctypes.data)
    return D


# ----------------------------------------------------------------------------------------
# dgcnn/ops.py:21-40  edges
# ----------------------------------------------------------------------------------------
def edges(points: np.ndarray, k: int = 20, idx: np.ndarray | None = None) -> np.ndarray:
    """E (B,N,k,2C) = concat[x_i, x_j - x_i]; first C channels = centre (ops.py:39)."""
    if idx is None:
        idx = k_nn(points, k)                                            # ops.py:23
    B, N, C = points.shape
    off = (np.arange(B) * N).reshape(B, 1, 1)                            # ops.py:30-31
    flat = points.reshape(-1, C)                                         # ops.py:33
    nbr = flat[(idx + off).reshape(-1)].reshape(B, N, k, C)              # ops.py:34
    cen = np.tile(points[:, :, None, :], (1, 1, k, 1))                   # ops.py:35-37
    return np.concatenate([cen, nbr - cen], axis=-1)                     # ops.py:39


def edges_bwd(dE: np.ndarray, idx: np.ndarray, B: int, N: int, C: int) -> np.ndarray:
    """Transpose of edges(): tile^T (sum over k) and gather^T (scatter-add). SURVEY A.5."""
    k = idx.shape[-1]
    d_cen = dE[..., :C] - dE[..., C:]
    dx = d_cen.sum(axis=2)
    off = (np.arange(B) * N).reshape(B, 1, 1)
    flat = np.zeros((B * N, C), dE.dtype)
    np.add.at(flat, (idx + off).reshape(-1), dE[..., C:].reshape(-1, C))
    return dx + flat.reshape(B, N, C)


# ----------------------------------------------------------------------------------------
# slim.conv2d(kernel 1, no bias) + slim.batch_norm(defaults) + activation   [TF1-lib A.2/A.3]
# ----------------------------------------------------------------------------------------
def tree_colsum(a):
    """Column sums of a (..., F) array over all leading axes IN THE ARRAY'S OWN DTYPE, by fold-in-half pairwise
    summation -- the oracle's normative order for every reduction over the BatchNorm axes (SURVEY Appendix A gives TF1
    no order; Eigen's reductions are tree-shaped / vectorised, not a running row-by-row sum):

        rows r_0 .. r_{n-1};  while n > 1:  h = n // 2;  r_i <- r_i + r_{n-h+i} for i < h;  n <- n - h;   result r_0

    (with n odd the middle row waits for the next round).  Every addition is one rounding in dtype; the error grows with
    log2(n) instead of n: at the 983 040 rows of configs[1]'s conv0 a running float32 sum (what numpy's `mean(axis=0)`
    does over a non-contiguous axis) is 1.9e-3 off in the logits and 2-4e-2 in the gradients, this order ~1e-5 (measured
    against the float64 evaluation, profiles/r03/oracle_reduction.txt)."""
    F = a.shape[-1]
    r = np.array(a, copy=True).reshape(-1, F)
    n = r.shape[0]
    while n > 1:
        h = n // 2
        r[:h] += r[n - h:n]
        n -= h
    return r[0].copy()


def tree_colmean(a):
    n = a.size // a.shape[-1]
    return tree_colsum(a) / a.dtype.type(n)


def bf16_round(a):
    """Round every element to the nearest bfloat16 (ties to even) and return it IN THE ARRAY'S OWN DTYPE: the operand
    rounding of the `bf16 edge-MLP` mode (BASELINE.json configs[2]; what v_cvt_pk_bf16_f32 does on the device).  A float64
    input goes through float32 first, as it would on the device, where the operands exist in float32 before they are rounded."""
    f = np.ascontiguousarray(a, np.float32)
    u = f.view(np.uint32)
    r = ((u + np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1))) & np.uint32(0xFFFF0000)).view(np.float32)
    r = np.where(np.isfinite(f), r, f)
    return r.astype(a.dtype)


def conv_bn_act(x, W, beta, relu=True, eps=BN_EPS, operand_round=None):
    """x (..., Cin) @ W (Cin,Cout); BN over all axes but the last with batch statistics
    (biased variance), +beta, no gamma; optional ReLU.  Returns (out, cache).
    Reductions over the BatchNorm axes: tree_colsum (fold-in-half pairwise, in the working dtype).
    operand_round: None, or a function applied to BOTH operands of the product (and, in the backward, to the three operands of
    the two gradient products) before they are multiplied -- products and sums stay in the working dtype.  `bf16_round` makes
    this the bf16 edge-MLP of BASELINE.json configs[2]: bf16 operands, wide accumulate."""
    dt = x.dtype
    if operand_round is not None:
        x, W = operand_round(x), operand_round(W)
    y = np.matmul(x, W)
    mu = tree_colmean(y)
    var = tree_colmean(np.square(y - mu))
    rstd = (1.0 / np.sqrt(var + dt.type(eps))).astype(dt)
    xhat = (y - mu) * rstd
    z = xhat + beta
    out = np.maximum(z, 0) if relu else z
    return out, dict(x=x, W=W, xhat=xhat, rstd=rstd, out=out, relu=relu, operand_round=operand_round)


def conv_bn_act_bwd(dout, cache):
    """-> (dx, dW, dbeta).  BN-train backward of SURVEY A.5; ReLU grad = [out>0]."""
    x, W, xhat, rstd = cache["x"], cache["W"], cache["xhat"], cache["rstd"]
    dz = dout * (cache["out"] > 0) if cache["relu"] else dout
    dbeta = tree_colsum(dz)
    m1 = dbeta / dz.dtype.type(dz.size // dz.shape[-1])
    m2 = tree_colmean(dz * xhat)
    dy = rstd * (dz - m1 - xhat * m2)
    if cache.get("operand_round") is not None:       # (x and W in the cache are the rounded operands of the forward)
        dy = cache["operand_round"](dy.astype(x.dtype))
    Cin, Cout = W.shape
    dW = np.matmul(x.reshape(-1, Cin).T, dy.reshape(-1, Cout))
    dx = np.matmul(dy, W.T)
    return dx, dW, dbeta


# ----------------------------------------------------------------------------------------
# dgcnn/ops.py:42-73  edge_conv
# ----------------------------------------------------------------------------------------
def edge_conv(point_cloud, k, W0, beta0, W1, beta1, relu1=True, idx=None, edge_mlp_dtype="f32"):
    """Returns ([net_max, net_mean, net], cache), each (B,N,1,ch) as ops.py:73.
    `idx` may be forced (tests feed the HIP path's own indices for layers >= 1).
    edge_mlp_dtype = "bf16" (BASELINE.json configs[2] "bf16 edge-MLP MFMA"; the reference itself has no such switch -- TF1
    computes conv0 in float32): the operands of conv0's product, E = [x_i, x_j - x_i] (formed in the working dtype FIRST)
    and W0, are rounded to bfloat16 once; accumulation, BatchNorm and everything else stay in the working dtype; the two
    gradient products of conv0 round their operands (dy, E, W0) the same way.  conv1 is not part of the edge MLP."""
    if idx is None:
        idx = k_nn(point_cloud, k)
    if edge_mlp_dtype not in ("f32", "bf16"):
        raise ValueError("edge_mlp_dtype must be f32 or bf16, got %r" % (edge_mlp_dtype,))
    E = edges(point_cloud, k, idx)                                       # ops.py:45
    y, c0 = conv_bn_act(E, W0, beta0, relu=True,                         # ops.py:47-54
                        operand_round=bf16_round if edge_mlp_dtype == "bf16" else None)
    net_max = y.max(axis=-2, keepdims=True)                              # ops.py:56
    net_mean = y.mean(axis=-2, keepdims=True, dtype=y.dtype)             # ops.py:57
    cat = np.concatenate([net_max, net_mean], axis=-1)                   # ops.py:58
    net, c1 = conv_bn_act(cat, W1, beta1, relu=relu1)                    # ops.py:62-70
    cache = dict(idx=idx, c0=c0, c1=c1, y=y, net_max=net_max, shape=point_cloud.shape)
    return [net_max, net_mean, net], cache


def edge_conv_bwd(d_max, d_mean, d_net, cache):
    """-> (d_point_cloud, dict(W0,beta0,W1,beta1)).  reduce_max ties share equally (A.5)."""
    B, N, C = cache["shape"]
    y = cache["y"]
    k = y.shape[-2]
    F = y.shape[-1]
    dcat, dW1, db1 = conv_bn_act_bwd(d_net, cache["c1"])
    dmx = d_max + dcat[..., :F]
    dmn = d_mean + dcat[..., F:]
    is_max = (y == cache["net_max"])
    cnt = is_max.sum(axis=-2, keepdims=True)
    dy = is_max * (dmx / cnt) + dmn / k
    dE, dW0, db0 = conv_bn_act_bwd(dy.astype(y.dtype), cache["c0"])
    dx = edges_bwd(dE, cache["idx"], B, N, C)
    return dx, dict(W0=dW0, beta0=db0, W1=dW1, beta1=db1)


# ----------------------------------------------------------------------------------------
# Parameters: names / shapes / creation order of tf.trainable_variables()   
# ----------------------------------------------------------------------------------------
def _as_list(v, n, what):
    if isinstance(v, list):
        if len(v) != n:
            raise ValueError("Length of %s != repeat" % what)   # ops.py:80-87,105-112,147-149
        return [int(a) for a in v]
    return [int(v)] * n


def param_specs(flags, num_channel):
    """[(name, shape)] in variable-creation order of dgcnn/model.py:9-106 under scope 'dgcnn/'."""
    L = int(flags.EDGE_CONV_LAYERS)
    ecf = _as_list(flags.EDGE_CONV_FILTERS, L, "num_filters")
    specs = []
    cin = num_channel
    residual = flags.MODEL_NAME in ("residual-dgcnn", "residual-dgcnn-nofc")
    for i in range(L):
        s = "EdgeConv%d" % i
        specs += [(s + "/conv0/weights", (2 * cin, ecf[i])), (s + "/conv0/BatchNorm/beta", (ecf[i],)),
                  (s + "/conv1/weights", (2 * ecf[i], 64)), (s + "/conv1/BatchNorm/beta", (64,))]
        if residual and i > 0 and ecf[i] != ecf[i - 1]:                   # ops.py:124-133
            specs += [(s + "/shortcut/weights", (64, ecf[i])), (s + "/shortcut/BatchNorm/beta", (ecf[i],))]
        cin = 64
    nc = int(flags.NUM_CLASS)
    if flags.MODEL_NAME == "residual-dgcnn-nofc":                          # model.py:45-58
        specs += [("Final/weights", (64, nc)), ("Final/BatchNorm/beta", (nc,))]
        return specs
    specs += [("MergedEdgeConv/weights", (64 * L, 1024)), ("MergedEdgeConv/BatchNorm/beta", (1024,))]
    cat = 1024 + sum(2 * f + 64 for f in ecf) + 1024                       # model.py:83-85
    fcl = int(flags.FC_LAYERS)
    fcf = _as_list(flags.FC_FILTERS, fcl, "num_filters")
    cin = cat
    for i in range(fcl):
        specs += [("FC%d/weights" % i, (cin, fcf[i])), ("FC%d/BatchNorm/beta" % i, (fcf[i],))]
        cin = fcf[i]
    specs += [("Final/weights", (cin, nc)), ("Final/BatchNorm/beta", (nc,))]
    return specs


def init_params(flags, num_channel, seed=1, dtype=np.float32):
    """Xavier-uniform weights (slim default initializer, A.2), beta = 0 (A.3)."""
    rng = np.random.default_rng(seed)
    params = {}
    for name, shape in param_specs(flags, num_channel):
        if name.endswith("weights"):
            lim = np.sqrt(6.0 / (shape[0] + shape[1]))
            params[name] = rng.uniform(-lim, lim, size=shape).astype(dtype)
        else:
            params[name] = np.zeros(shape, dtype)
    return params


# ----------------------------------------------------------------------------------------
# dgcnn/model.py:9-106  build   (+ ops.py:75-98 / :100-140 / :142-163)
# ----------------------------------------------------------------------------------------
def repeat_edge_conv(point_cloud, repeat, k, num_filters, P, residual=False, idx_list=None, edge_mlp_dtype="f32"):
    """dgcnn/ops.py:75-98 (residual=False) / ops.py:100-140 (residual=True): `repeat` EdgeConv layers, layer i+1 building its graph on
    squeeze(tensors[-1]).  k and num_filters are ints or lists of length `repeat` (ops.py:77-87: a list of another length prints and
    raises ValueError).  P: name -> array under the scopes EdgeConv<i>/...  Returns (tensors [3 * repeat], per-layer caches)."""
    repeat = int(repeat)
    kv = _as_list(k, repeat, "k")
    ecf = _as_list(num_filters, repeat, "num_filters")
    net = point_cloud
    tensors, layers = [], []
    shortcut = None
    for i in range(repeat):                                                 # ops.py:91-96 / :116-138
        s = "EdgeConv%d/" % i
        relu1 = not (residual and shortcut is not None)                      # ops.py:123 activation=None
        outs, c = edge_conv(net, kv[i], P[s + "conv0/weights"], P[s + "conv0/BatchNorm/beta"],
                            P[s + "conv1/weights"], P[s + "conv1/BatchNorm/beta"], relu1=relu1,
                            idx=None if idx_list is None else idx_list[i], edge_mlp_dtype=edge_mlp_dtype)
        rec = dict(ec=c, sc=None, pre=None)
        if residual and shortcut is not None:
            sc_in = shortcut
            if ecf[i] != ecf[i - 1]:                                        # ops.py:124-133
                sc_in, csc = conv_bn_act(shortcut, P[s + "shortcut/weights"],
                                         P[s + "shortcut/BatchNorm/beta"], relu=False)
                rec["sc"] = csc
            if sc_in.shape != outs[2].shape:
                raise ValueError("residual shortcut shape mismatch (ops.py:134)")
            pre = sc_in + outs[2]
            outs[2] = np.maximum(pre, 0)                                    # ops.py:134
            rec["pre"] = pre
        tensors += outs
        layers.append(rec)
        net = outs[2][:, :, 0, :]                                           # ops.py:95-96 squeeze
        if residual:
            shortcut = outs[2]                                               # ops.py:137
    return tensors, layers


def model_forward(point_cloud, flags, params, idx_list=None, dropout_mask=None, k=None):
    """logits (B,N,num_class) and a cache for model_backward.

    idx_list: optional per-EdgeConv-layer forced neighbour indices.
    dropout_mask: (B,N,1,Cfc) of {0, 1/0.7} applied when flags.TRAIN (model.py:90-91);
                  None => no dropout (parity runs).
    k: None => int(flags.KVALUE) for every layer, as model.py:16 casts it; a list gives one k per EdgeConv layer -- the form
       ops.repeat_edge_conv itself accepts (ops.py:77-82), which the tests of that operator drive through this function."""
    dt = point_cloud.dtype
    P = {n: v.astype(dt) for n, v in params.items()}
    L = int(flags.EDGE_CONV_LAYERS)
    name = flags.MODEL_NAME
    if name not in ("dgcnn", "residual-dgcnn", "residual-dgcnn-nofc"):
        raise NotImplementedError("Unsupported MODEL_NAME: %s" % name)      # model.py:41-43
    residual = name != "dgcnn"
    tensors, layers = repeat_edge_conv(point_cloud, L, int(flags.KVALUE) if k is None else k, flags.EDGE_CONV_FILTERS, P,
                                       residual=residual, idx_list=idx_list,
                                       edge_mlp_dtype=str(getattr(flags, "EDGE_MLP_DTYPE", "f32")))
    cache = dict(layers=layers, flags=flags, L=L, P=P, residual=residual)
    if name == "residual-dgcnn-nofc":                                        # model.py:45-58
        fin, cf = conv_bn_act(tensors[-1], P["Final/weights"], P["Final/BatchNorm/beta"], relu=True)
        cache.update(final=cf, nofc=True)
        return fin[:, :, 0, :], cache
    cat = np.concatenate([tensors[3 * i + 2] for i in range(L)], axis=-1)    # model.py:60-63
    merged, cm = conv_bn_act(cat, P["MergedEdgeConv/weights"], P["MergedEdgeConv/BatchNorm/beta"], relu=True)
    tensors.append(merged)                                                   # model.py:74
    B, N = merged.shape[0], merged.shape[1]
    gmax = merged.max(axis=1, keepdims=True)                                 # model.py:76-77 max_pool over N
    gtile = np.tile(gmax.reshape(B, -1, 1, 1024), (1, N, 1, 1))              # model.py:80-81
    big = np.concatenate([gtile] + tensors, axis=3)                          # model.py:83-85
    fcl = int(flags.FC_LAYERS)
    fcs = []
    net = big
    for i in range(fcl):                                                     # ops.py:151-160
        net, cfc = conv_bn_act(net, P["FC%d/weights" % i], P["FC%d/BatchNorm/beta" % i], relu=True)
        fcs.append(cfc)
    if bool(flags.TRAIN) and dropout_mask is not None:                       # model.py:90-91
        net = net * dropout_mask.astype(dt)
    fin, cf = conv_bn_act(net, P["Final/weights"], P["Final/BatchNorm/beta"], relu=True)  # model.py:94-101
    cache.update(merged=cm, merged_out=merged, fcs=fcs, final=cf, nofc=False,
                 dropout_mask=dropout_mask if bool(flags.TRAIN) else None,
                 widths=[t.shape[-1] for t in tensors])
    return fin[:, :, 0, :], cache                                            # model.py:104 squeeze


def model_backward(dlogits, cache):
    """-> dict name -> gradient, for every trainable variable."""
    G = {}
    L = cache["L"]
    d = dlogits[:, :, None, :]
    d, G["Final/weights"], G["Final/BatchNorm/beta"] = conv_bn_act_bwd(d, cache["final"])
    layers = cache["layers"]
    if cache["nofc"]:
        d_t = [None] * (3 * L)
        d_t[-1] = d
    else:
        if cache["dropout_mask"] is not None:
            d = d * cache["dropout_mask"].astype(d.dtype)
        for i in reversed(range(len(cache["fcs"]))):
            d, G["FC%d/weights" % i], G["FC%d/BatchNorm/beta" % i] = conv_bn_act_bwd(d, cache["fcs"][i])
        widths = cache["widths"]
        off = 1024
        d_g = d[..., :1024].sum(axis=1, keepdims=True)                      # tile^T
        d_t = []
        for w in widths:
            d_t.append(d[..., off:off + w]); off += w
        d_merged = d_t.pop()                                                 # tensors.append(net) model.py:74
        merged = cache["merged_out"]
        B, N = merged.shape[:2]
        am = merged.argmax(axis=1)                                           # first arg-max (A.5 max_pool grad)
        gm = np.zeros_like(merged)
        bi, _, ci = np.meshgrid(np.arange(B), np.arange(1), np.arange(1024), indexing="ij")
        gm[bi, am, 0 * am, ci] = d_g.reshape(B, 1, 1024)
        d_merged = d_merged + gm
        dcat, G["MergedEdgeConv/weights"], G["MergedEdgeConv/BatchNorm/beta"] = conv_bn_act_bwd(d_merged, cache["merged"])
        for i in range(L):
            d_t[3 * i + 2] = d_t[3 * i + 2] + dcat[..., 64 * i:64 * (i + 1)]
    d_next = None     # gradient flowing into layer i's output from layer i+1's input / shortcut
    for i in reversed(range(L)):
        s = "EdgeConv%d/" % i
        rec = layers[i]
        d_net = d_t[3 * i + 2] if d_t[3 * i + 2] is not None else 0
        if d_next is not None:
            d_net = d_net + d_next
        zero = np.zeros_like(rec["ec"]["net_max"])
        d_max = d_t[3 * i] if d_t[3 * i] is not None else zero
        d_mean = d_t[3 * i + 1] if d_t[3 * i + 1] is not None else zero
        d_short = None
        if rec["pre"] is not None:                                           # relu(shortcut + net) ops.py:134
            d_pre = d_net * (rec["pre"] > 0)
            d_net = d_pre
            d_short = d_pre
            if rec["sc"] is not None:
                d_short, G[s + "shortcut/weights"], G[s + "shortcut/BatchNorm/beta"] = conv_bn_act_bwd(d_pre, rec["sc"])
        dx, g = edge_conv_bwd(d_max, d_mean, d_net, rec["ec"])
        G[s + "conv0/weights"], G[s + "conv0/BatchNorm/beta"] = g["W0"], g["beta0"]
        G[s + "conv1/weights"], G[s + "conv1/BatchNorm/beta"] = g["W1"], g["beta1"]
        d_next = dx[:, :, None, :]
        if d_short is not None:
            d_next = d_next + d_short
    return G


# ----------------------------------------------------------------------------------------
# dgcnn/trainval.py:39-52  softmax / accuracy / loss ; :64-80 gradient handling ; Adam (A.4)
# ----------------------------------------------------------------------------------------
def softmax_xent(logits, labels, weight=None):
    """-> (loss, softmax, accuracy, dlogits).  loss = mean over (MBS,N) of CE * weight."""
    dt = logits.dtype
    z = logits - logits.max(axis=-1, keepdims=True)
    e = np.exp(z)
    sm = e / e.sum(axis=-1, keepdims=True)                                   # trainval.py:39
    B, N, K = logits.shape
    acc = np.mean((np.argmax(logits, axis=2) == labels).astype(np.float32))  # trainval.py:41-42
    lse = np.log(e.sum(axis=-1)) - z[np.arange(B)[:, None], np.arange(N)[None, :], labels]
    w = np.ones((B, N), dt) if weight is None else weight.astype(dt)         # trainval.py:47-51
    loss = np.mean(lse * w, dtype=dt)                                        # trainval.py:52
    onehot = np.zeros_like(sm)
    onehot[np.arange(B)[:, None], np.arange(N)[None, :], labels] = 1
    dlogits = (sm - onehot) * (w / dt.type(B * N))[..., None]
    return loss, sm, acc, dlogits.astype(dt)


def adam_step(param, grad, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer defaults (A.4); t is the 1-based step count. In-place."""
    dt = param.dtype
    lr_t = dt.type(lr * np.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t))
    m += (grad - m) * dt.type(1 - b1)
    v += (np.square(grad) - v) * dt.type(1 - b2)
    param -= lr_t * m / (np.sqrt(v) + dt.type(eps))


def train_step_grads(points, labels, flags, params, weight=None, idx_list=None, dropout_mask=None):
    """One micro-step of trainval.accum_gradient on one tower: returns (grads, loss, acc, softmax)."""
    logits, cache = model_forward(points, flags, params, idx_list=idx_list, dropout_mask=dropout_mask)
    loss, sm, acc, dlogits = softmax_xent(logits, labels, weight)
    return model_backward(dlogits, cache), loss, acc, sm


class Flags(object):
    """The attributes dgcnn/model.py:11-18,27 and dgcnn/trainval.py:17,26,31,47 read."""
    NUM_CLASS = 2
    MODEL_NAME = "dgcnn"
    TRAIN = True
    KVALUE = 20
    DEBUG = False
    EDGE_CONV_LAYERS = 3
    EDGE_CONV_FILTERS = 64
    FC_LAYERS = 2
    FC_FILTERS = [512, 256]
    LEARNING_RATE = 0.001
    GPUS = [0]
    MINIBATCH_SIZE = 1
    NUM_CHANNEL = 3
    WEIGHT_KEY = ""
    EDGE_MLP_DTYPE = "f32"

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)
