"""Shared helpers of the -m gpu tests (imported by test modules; nothing here runs without a GPU)."""
import numpy as np
import torch


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


def set_vars(dg, params):
    c = dg.ctx()
    for n, v in params.items():
        c.set_variable(n, v)


class capture_layers(object):
    """Context manager: records (input features (B,N,C) on the host, idx (B,N,k)) of every EdgeConv layer the
    HIP path runs, keyed by the layer's variable scope ('EdgeConv0', ...)."""

    def __init__(self, keep_inputs=True):
        self.layers = {}
        self.keep_inputs = keep_inputs

    def __enter__(self):
        from dgcnn import _engine as E
        self._E = E
        self._orig = orig = E.edge_conv_block
        cap = self

        def hook(x, B, N, k, F, **kw):
            mm, net, idx = orig(x, B, N, k, F, **kw)
            xin = host(x).reshape(B, N, -1).copy() if cap.keep_inputs else None
            cap.layers["/".join(E.ctx().scope)] = (xin, host(idx))
            return mm, net, idx
        E.edge_conv_block = hook
        return self

    def __exit__(self, *exc):
        self._E.edge_conv_block = self._orig
        return False


def run_model(dg, flags, pts, params, train, labels=None):
    """trainval on `pts` with the given parameter values; -> (trainval, result list, {scope: (x_in, idx)})."""
    tv = dg.trainval(flags)
    tv.initialize()
    if params is not None:
        set_vars(dg, params)
    with capture_layers() as cap:
        if train:
            tv.zero_gradients(None)
            res = tv.accum_gradient(None, [pts], [labels])
        else:
            res = tv.inference(None, [pts], None if labels is None else [labels])
    return tv, res, cap.layers


def idx_set_mismatch(a, b):
    """Fraction of rows whose neighbour SET differs / whose ordered list differs, for two (B,N,k) index arrays."""
    sa, sb = np.sort(a, axis=-1), np.sort(b, axis=-1)
    return float((sa != sb).any(-1).mean()), float((a != b).any(-1).mean())
