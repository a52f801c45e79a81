"""Data-parallel host logic (the only parallelism the reference has: dgcnn/trainval.py:26-80).

Reference: in-graph towers, per-variable gradients copied to the host and averaged there
(trainval.py:16,64-73).  Here: one process per GPU, clouds sharded by rank, and ONE all-reduce of
the flat fp32 gradient bucket per optimizer step (RCCL over xGMI through torch.distributed backend
"nccl"; "gloo" on CPU in tests/test_dp_gloo.py).  Because sum_steps(mean_gpu g) == mean_gpu(sum_steps g),
the collective runs once after local accumulation, not once per micro-step.  N (points of one
cloud) is never sharded and BatchNorm statistics stay per replica, exactly as in the reference.
"""
from __future__ import annotations

import torch

from . import _hip as H


def shard_bounds(global_batch, rank, world):
    """Clouds [lo, hi) of rank `rank` (the global batch must divide evenly, main_funcs.py:57-61)."""
    if global_batch % world:
        raise ValueError("--batch_size (%d) must be a multiple of the number of replicas (%d)" % (global_batch, world))
    per = global_batch // world
    return rank * per, (rank + 1) * per


def dist_state():
    """(dist module or None, rank, world)."""
    try:
        import torch.distributed as dist
    except ImportError:
        return None, 0, 1
    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


def broadcast_(flat, dist, src=0):
    """Variables are shared by all towers (tf.AUTO_REUSE, trainval.py:29): rank 0's init wins."""
    if dist is not None:
        dist.broadcast(flat, src=src)
    return flat


def allreduce_mean_(flat, dist, world):
    """flat <- mean over replicas of flat (trainval.py:64-73), in place, one collective."""
    if dist is None:
        return flat
    dist.all_reduce(flat)                                   # SUM over replicas (a one-rank group still issues it)
    if world == 1:
        return flat
    if flat.is_cuda:
        H.call("dgcnn_axpby_f32", flat.data_ptr(), 1.0 / world, flat.data_ptr(), 0.0, flat.numel())
    else:                                                   # host tensors only exist in the gloo tests
        flat.mul_(1.0 / world)
    return flat
