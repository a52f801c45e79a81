"""world_size-2 test of the data-parallel host logic on CPU (gloo): clouds sharded by rank, one flat
all-reduce, mean over replicas, parameters broadcast from rank 0.  Per-rank gradients come from the
oracle (no GPU here); the expected value is the oracle's single-process tower mean
(dgcnn/trainval.py:64-73), i.e. what the reference computes for GPUS=[0,1]."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import dgcnn_oracle as O

WORLD = 2


def _flags():
    return O.Flags(EDGE_CONV_LAYERS=1, KVALUE=4, FC_FILTERS=[16, 8], EDGE_CONV_FILTERS=64, TRAIN=True)


def _data():
    rng = np.random.default_rng(11)
    return rng.random((4, 32, 3), dtype=np.float32), rng.integers(0, 2, (4, 32)).astype(np.int32)


def _flat(G, specs):
    return torch.from_numpy(np.concatenate([G[n].reshape(-1) for n, _ in specs]).astype(np.float32))


def _worker(rank, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=WORLD)
    from dgcnn import parallel
    flags = _flags()
    specs = O.param_specs(flags, 3)
    d, r, w = parallel.dist_state()
    assert (r, w) == (rank, WORLD)
    # parameters: every rank draws different values, rank 0's must win after the broadcast
    params = O.init_params(flags, 3, seed=100 + rank)
    flat_p = _flat(params, specs)
    parallel.broadcast_(flat_p, d, src=0)
    # this rank's shard of the 4-cloud global batch, two accumulated micro-steps of one cloud each
    pts, lab = _data()
    lo, hi = parallel.shard_bounds(4, rank, WORLD)
    p0 = O.init_params(flags, 3, seed=100)
    acc = None
    for i in range(lo, hi):
        G, _, _, _ = O.train_step_grads(pts[i:i + 1], lab[i:i + 1], flags, p0)
        f = _flat(G, specs)
        acc = f if acc is None else acc + f                 # sum over micro-steps (trainval.py:79)
    parallel.allreduce_mean_(acc, d, w)                      # ONE collective per optimizer step
    if rank == 0:
        torch.save({"grad": acc, "param": flat_p}, out)
    else:
        torch.save({"param": flat_p}, out + ".r1")
    dist.destroy_process_group()


def test_two_rank_gradient_mean_matches_reference_tower_mean(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "dp.pt")
    mp.spawn(_worker, args=(port, out), nprocs=WORLD, join=True)
    got = torch.load(out)
    flags = _flags()
    specs = O.param_specs(flags, 3)
    p0 = O.init_params(flags, 3, seed=100)
    pts, lab = _data()
    # reference semantics: micro-step s feeds cloud (lo_r + s) to tower r; mean over towers, sum over steps
    exp = None
    for step in range(2):
        towers = [_flat(O.train_step_grads(pts[r * 2 + step:r * 2 + step + 1], lab[r * 2 + step:r * 2 + step + 1],
                                           flags, p0)[0], specs) for r in range(WORLD)]
        m = (towers[0] + towers[1]) / WORLD
        exp = m if exp is None else exp + m
    np.testing.assert_allclose(got["grad"].numpy(), exp.numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_array_equal(got["param"].numpy(), _flat(p0, specs).numpy())
    np.testing.assert_array_equal(torch.load(out + ".r1")["param"].numpy(), _flat(p0, specs).numpy())


def test_shard_bounds():
    from dgcnn import parallel
    assert [parallel.shard_bounds(192, r, 8) for r in (0, 7)] == [(0, 24), (168, 192)]    # BASELINE config 4
    with pytest.raises(ValueError):
        parallel.shard_bounds(25, 0, 8)
