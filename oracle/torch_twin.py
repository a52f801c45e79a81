"""oracle/torch_twin.py -- TEST INFRASTRUCTURE ONLY.

An independent op-for-op restatement of the reference graph (dgcnn/ops.py:8-163,
dgcnn/model.py:9-106, dgcnn/trainval.py:39-52) on torch-CPU tensors, with torch autograd supplying
the backward.  Two uses: (1) tests/test_oracle.py checks the numpy oracle's hand-written backward
against it; (2) bench.py's `cpu_baseline` leg times it on the GPU box's host cores (multi-threaded
MKL), as the stand-in for the un-runnable TF1-CPU path.  PARITY UNPINNED,
same caveat as oracle/dgcnn_oracle.py.  Never imported by the product package.
"""
import torch

BN_EPS = 1e-3


def conv_bn_act(x, W, beta, relu=True):
    y = x @ W
    dims = tuple(range(y.dim() - 1))
    mu = y.mean(dim=dims)
    var = ((y - mu) ** 2).mean(dim=dims)
    z = (y - mu) / torch.sqrt(var + BN_EPS) + beta
    return torch.relu(z) if relu else z


def k_nn(points, k):
    """ops.py:8-19 with the (B,N,N) matrix materialised, as the reference does."""
    inner = points @ points.transpose(1, 2)
    sq = (points ** 2).sum(-1, keepdim=True)
    d = sq + sq.transpose(1, 2) - 2 * inner
    return torch.topk(-d, k, dim=-1).indices


def edges(net, idx):
    B, N, C = net.shape
    k = idx.shape[-1]
    off = (torch.arange(B) * N).view(B, 1, 1)
    nbr = net.reshape(-1, C)[(idx + off).reshape(-1)].reshape(B, N, k, C)
    cen = net[:, :, None, :].expand(-1, -1, k, -1)
    return torch.cat([cen, nbr - cen], dim=-1)


def as_list(v, n):
    return [int(a) for a in v] if isinstance(v, list) else [int(v)] * n


def model(points, flags, P, idx_list=None, k=None):
    """logits (B,N,num_class).  idx_list forces the neighbour graph (tests); None = compute it.  k: None = int(flags.KVALUE) for
    every layer (model.py:16); a list = one k per layer, the form ops.repeat_edge_conv accepts (ops.py:77-82)."""
    L = int(flags.EDGE_CONV_LAYERS)
    residual = flags.MODEL_NAME != "dgcnn"
    ecf = as_list(flags.EDGE_CONV_FILTERS, L)
    kv = as_list(int(flags.KVALUE) if k is None else k, L)
    net = points
    tensors = []
    shortcut = None
    B, N, _ = points.shape
    for i in range(L):
        s = "EdgeConv%d/" % i
        idx = k_nn(net.detach(), kv[i]) if idx_list is None else torch.as_tensor(idx_list[i], dtype=torch.long)
        E = edges(net, idx)
        y = conv_bn_act(E, P[s + "conv0/weights"], P[s + "conv0/BatchNorm/beta"])
        mx = y.amax(dim=-2, keepdim=True)
        mn = y.mean(dim=-2, keepdim=True)
        relu1 = not (residual and shortcut is not None)
        out = conv_bn_act(torch.cat([mx, mn], -1), P[s + "conv1/weights"], P[s + "conv1/BatchNorm/beta"], relu1)
        if residual and shortcut is not None:
            sc = shortcut
            if ecf[i] != ecf[i - 1]:
                sc = conv_bn_act(sc, P[s + "shortcut/weights"], P[s + "shortcut/BatchNorm/beta"], False)
            out = torch.relu(sc + out)
        tensors += [mx, mn, out]
        net = out[:, :, 0, :]
        if residual:
            shortcut = out
    if flags.MODEL_NAME == "residual-dgcnn-nofc":
        return conv_bn_act(tensors[-1], P["Final/weights"], P["Final/BatchNorm/beta"])[:, :, 0, :]
    cat = torch.cat([tensors[3 * i + 2] for i in range(L)], -1)
    merged = conv_bn_act(cat, P["MergedEdgeConv/weights"], P["MergedEdgeConv/BatchNorm/beta"])
    tensors.append(merged)
    g = merged.amax(dim=1, keepdim=True).expand(-1, N, -1, -1)
    net = torch.cat([g] + tensors, dim=3)
    for i in range(int(flags.FC_LAYERS)):
        net = conv_bn_act(net, P["FC%d/weights" % i], P["FC%d/BatchNorm/beta" % i])
    if bool(flags.TRAIN) and getattr(flags, "_DROPOUT", False):
        net = torch.nn.functional.dropout(net, p=0.3, training=True)            # keep_prob 0.7, model.py:91
    return conv_bn_act(net, P["Final/weights"], P["Final/BatchNorm/beta"])[:, :, 0, :]


def train_step(points, labels, flags, P):
    """forward + loss + backward of one tower (trainval.py:38-54); returns the loss value."""
    for v in P.values():
        v.grad = None
    logits = model(points, flags, P)
    loss = torch.nn.functional.cross_entropy(logits.reshape(-1, logits.shape[-1]), labels.reshape(-1))
    loss.backward()
    return float(loss.detach())
