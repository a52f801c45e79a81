"""MI355X-native counterpart of the reference's operator module (dgcnn/ops.py:8-163).

Same names, positional order, defaults, return conventions and errors as the reference; tensors
are torch-ROCm tensors (memory holders) instead of tf.Tensor, and a call executes the HIP
kernels at once instead of adding nodes to a TF graph.  `knn` / `build_edge_conv` are the
aliases BASELINE.json's north_star uses for `k_nn` / `edge_conv`.
"""
from __future__ import annotations

import torch

from . import _engine as E
from . import _hip as H


def relu(x):
    """Marker for `activation=` (tf.nn.relu in the reference); never called on tensors."""
    raise TypeError("dgcnn.ops.relu is an activation marker, not a callable op")


def _is_relu(activation):
    if activation is None:
        return False
    if activation is relu or getattr(activation, "__name__", "") == "relu":
        return True
    raise NotImplementedError("only relu / None activations exist in the reference (ops.py:42,123)")


def k_nn(points, k):
    """dgcnn/ops.py:8-19.  points (B,N,C) -> idx (B,N,k) int32: the k nearest (squared L2 via
    (s_i+s_j)-2<x_i,x_j>), self included, ascending, ties -> lower index.  Bit-exact vs the oracle."""
    x, B, N = E.as2d(points)
    k = int(k)
    if k > N or k <= 0:
        raise ValueError("k_nn: k=%d must be in [1, N=%d] (tf.nn.top_k raises otherwise)" % (k, N))
    return E.knn(x, B, N, k)


knn = k_nn


def edges(points, k=20):
    """dgcnn/ops.py:21-40.  (B,N,C) -> edge features (B,N,k,2C) = concat[x_i, x_j - x_i]."""
    x, B, N = E.as2d(points)
    C = x.shape[1]
    idx = k_nn(points, k)
    out = torch.empty((B, N, int(k), 2 * C), dtype=torch.float32, device=x.device)
    H.call("dgcnn_edge_gather_f32", x.data_ptr(), H.ld2(x), idx.data_ptr(), B, N, C, int(k), out.data_ptr())
    return out


def edge_conv(point_cloud, k, num_filters, trainable, activation=relu, debug=False, _outs=None, _net2=None, _seed=None):
    """dgcnn/ops.py:42-73.  Returns the list [net_max, net_mean, net], each (B,N,1,ch).
    _seed: the previous layer's neighbour graph (the stacks pass it): only speeds this layer's k-NN up."""
    x, B, N = E.as2d(point_cloud)
    F = int(num_filters)
    mm, net, idx = E.edge_conv_block(x, B, N, int(k), F, relu1=_is_relu(activation), outs=_outs, net2=_net2, seed=_seed)
    res = [E.rank4(mm[:, :F], B, N), E.rank4(mm[:, F:], B, N), E.rank4(net, B, N)]
    if debug:
        for t in res:
            print("Shape %s ... Name %s" % (tuple(t.shape), E.ctx().full_name("edge_conv")))
    edge_conv.last_idx = idx
    return res


build_edge_conv = edge_conv


def _listify(v, repeat, what):
    if isinstance(v, list):
        if len(v) != repeat:
            print("Length of %s != repeat" % what)
            raise ValueError("Length of %s != repeat" % what)     # ops.py:80-87
        return [int(a) for a in v]
    return [int(v)] * repeat


def repeat_edge_conv(point_cloud, repeat, k, num_filters, trainable, debug=False, _plan=None):
    """dgcnn/ops.py:75-98.  Flat list of 3*repeat tensors; layer i+1 builds its k-NN graph on
    squeeze(tensors[-1]) -- the dynamic graph."""
    repeat = int(repeat)
    k = _listify(k, repeat, "k")
    num_filters = _listify(num_filters, repeat, "num_filters")
    net = point_cloud
    tensors = []
    seed = None
    for i in range(repeat):
        with E.variable_scope("EdgeConv%d" % i):
            outs, net2 = _plan(i) if _plan is not None else (None, None)
            tensors += edge_conv(net, k[i], num_filters[i], trainable, debug=debug, _outs=outs, _net2=net2, _seed=seed)
            seed = edge_conv.last_idx                    # layer i's graph seeds layer i + 1's search (same points)
            net = tensors[-1][:, :, 0, :]
    return tensors


def repeat_residual_edge_conv(point_cloud, repeat, k, num_filters, trainable, debug=False, _plan=None):
    """dgcnn/ops.py:100-140.  Layers >= 1: conv1 without activation, optional shortcut conv when
    num_filters changes, tensors[-1] = relu(shortcut + tensors[-1])."""
    repeat = int(repeat)
    k = _listify(k, repeat, "k")
    num_filters = _listify(num_filters, repeat, "num_filters")
    net = point_cloud
    tensors = []
    shortcut = None
    seed = None
    for i in range(repeat):
        with E.variable_scope("EdgeConv%d" % i):
            outs, net2 = _plan(i) if _plan is not None else (None, None)
            if shortcut is None:
                tensors += edge_conv(net, k[i], num_filters[i], trainable, debug=debug, _outs=outs, _net2=net2, _seed=seed)
            else:
                # conv1 (no activation) goes to a scratch buffer; relu(shortcut + conv1) takes the planned slot
                tensors += edge_conv(net, k[i], num_filters[i], trainable, activation=None, debug=debug,
                                     _outs=None if outs is None else (outs[0], None), _seed=seed)
                sc, B, N = E.as2d(shortcut)
                if not num_filters[i] == num_filters[i - 1]:
                    sc = E.conv_bn_act(sc, "shortcut", num_filters[i], relu=False)       # ops.py:125-133
                pre, _, _ = E.as2d(tensors[-1])
                res = E.add_relu(sc, pre, out=None if outs is None else outs[1])          # ops.py:134
                if net2 is not None:
                    H.call("dgcnn_copy2d_f32", res.data_ptr(), H.ld2(res), net2.data_ptr(), H.ld2(net2),
                           res.shape[0], res.shape[1], 0)
                    E_copy_grad(res, net2)
                tensors[-1] = E.rank4(res, B, N)
            seed = edge_conv.last_idx                    # layer i's graph seeds layer i + 1's search (same points)
            net = tensors[-1]
            shortcut = tensors[-1]
            net = net[:, :, 0, :]
    return tensors


def E_copy_grad(src, dst):
    """Backward of `dst = copy(src)`: d(src) += d(dst)."""
    c = E.ctx()
    if not c.recording:
        return

    def bwd():
        gd, gs = c.grad(dst), c.grad(src)
        if gd is not None and gs is not None:
            H.call("dgcnn_copy2d_f32", gd.data_ptr(), H.ld2(gd), gs.data_ptr(), H.ld2(gs), gd.shape[0], gd.shape[1], 1)
    c.tape.append(bwd)


def fc(net, repeat, num_filters, trainable, debug=False):
    """dgcnn/ops.py:142-163.  repeat x [1x1 conv + BN + ReLU] under scopes FC0, FC1, ..."""
    repeat = int(repeat)
    num_filters = _listify(num_filters, repeat, "num_filters")
    x, B, N = E.as2d(net)
    for i in range(repeat):
        x = E.conv_bn_act(x, "FC%d" % i, num_filters[i], relu=True)
        if debug:
            print("Shape %s ... Name FC%d" % ((B, N, 1, num_filters[i]), i))
    return E.rank4(x, B, N)
