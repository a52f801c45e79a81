"""Data-parallel host logic (the only parallelism the reference has: dgcnn/trainval.py:26-80).

Reference: in-graph towers, per-variable gradients copied to the host and averaged there
(trainval.py:16,64-73).  Here: one process per GPU, clouds sharded by rank, and one all-reduce of
the flat fp32 gradient bucket per optimizer step.  Because sum_steps(mean_gpu g) == mean_gpu(sum_steps g),
the collective runs once after local accumulation, not once per micro-step.  N (points of one
cloud) is never sharded and BatchNorm statistics stay per replica, exactly as in the reference.

Data path: the library's OWN RCCL communicator (dgcnn/rccl.py over csrc/comm.cc: dgcnn_allreduce_f32 /
dgcnn_broadcast_f32, unique id over a TCP socket -- no torch.distributed): `init_rccl()` registers it and
`broadcast_` / `allreduce_mean_` / `allreduce_sum_async` use it for device tensors.  The gradient bucket is reduced in two
pieces: the head's gradients (97 % of the bytes; ready when the head's backward is done) travel on the communicator's stream
while the EdgeConv backward is still running, the rest follows in apply_gradient.
torch.distributed stays as (a) the control plane of the run loops (resume iteration, output gathering: host objects) and
(b) the fallback collective when no RCCL group is registered -- "gloo" in the CPU tests (tests/test_dp_gloo.py) and in the
two-ranks-on-one-GPU test, where RCCL refuses duplicate devices.
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


_RCCL = None          # dgcnn.rccl.Group: the data-path communicator of this process, when registered


def init_rccl(rank=None, world=None, addr=None, port=None):
    """Create and register this process' RCCL communicator (the current HIP device must already be selected)."""
    global _RCCL
    from . import rccl
    if _RCCL is not None:
        return _RCCL
    _RCCL = rccl.Group(rank=rank, world=world, addr=addr, port=port)
    return _RCCL


def rccl_group():
    return _RCCL


def shutdown_rccl():
    global _RCCL
    if _RCCL is not None:
        _RCCL.destroy()
        _RCCL = None


def dist_state():
    """(torch.distributed module or None, rank, world).  With only the RCCL group registered the module is None and
    rank / world come from the group."""
    try:
        import torch.distributed as dist
    except ImportError:
        dist = None
    if dist is not None and dist.is_available() and dist.is_initialized():
        return dist, dist.get_rank(), dist.get_world_size()
    if _RCCL is not None:
        return None, _RCCL.rank, _RCCL.world
    return None, 0, 1


def broadcast_(flat, dist, src=0):
    """Variables are shared by all towers (tf.AUTO_REUSE, trainval.py:29): rank 0's init wins."""
    if _RCCL is not None and flat.is_cuda:
        _RCCL.broadcast_(flat, root=src)
    elif dist is not None:
        dist.broadcast(flat, src=src)
    return flat


def allreduce_sum_async(flat):
    """Start the SUM all-reduce of a device bucket on the RCCL group's stream (ordered after the current stream); True when
    started -- the caller later calls rccl_group().wait().  False: no group (the caller reduces later by allreduce_mean_)."""
    if _RCCL is None or not flat.is_cuda:
        return False
    _RCCL.allreduce_sum_async(flat)
    return True


def scale_(flat, world):
    if world > 1:
        H.call("dgcnn_axpby_f32", flat.data_ptr(), 1.0 / world, flat.data_ptr(), 0.0, flat.numel())
    return flat


def allreduce_mean_(flat, dist, world):
    """flat <- mean over replicas of flat (trainval.py:64-73), in place, one collective."""
    if _RCCL is not None and flat.is_cuda:
        _RCCL.allreduce_sum_(flat)                          # SUM over replicas (a one-rank group still issues it)
        return scale_(flat, world)
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
