"""One data-parallel replica of the real trainer (launched by tests/test_gpu_dp.py through torch.distributed.run).

    python -m torch.distributed.run --nproc-per-node W tests/dp_worker.py --backend nccl|gloo --out DIR --data FILE

Every rank uses cuda:0 (the GPU box has one device): rank r trains on its shard of the global batch
(parallel.shard_bounds), two accumulated micro-steps, then trainval.apply_gradient = ONE all-reduce of the flat
gradient bucket + the mean over replicas + Adam (dgcnn/trainval.py:64-80 semantics).  Each rank writes the
post-all-reduce gradient bucket and the updated parameters to DIR/rank<r>.npz.
Exit code 3: the backend refused this layout (RCCL: two ranks on one device) -- the caller falls back to gloo.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "dynamic-gcnn_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch
import torch.distributed as dist


def flags_for(dgcnn):
    return dgcnn.DGCNN_FLAGS(MODEL_NAME="dgcnn", EDGE_CONV_LAYERS=3, EDGE_CONV_FILTERS=[64, 64, 128], FC_LAYERS=2,
                             FC_FILTERS=[512, 256], NUM_CLASS=2, KVALUE=20, NUM_CHANNEL=3, TRAIN=True, SEED=7,
                             LEARNING_RATE=1e-3, DETERMINISTIC=True)   # bit-reproducible kernels: the comparison between
    #                          separate processes must not depend on which near-tie neighbour an atomically summed BN picked


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--out", required=True)
    ap.add_argument("--data", required=True)
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    torch.cuda.set_device(0)
    try:
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
        else:
            dist.init_process_group(backend=args.backend, rank=rank, world_size=world)
        probe = torch.ones(4, device="cuda") * (rank + 1)
        dist.all_reduce(probe)                      # the first collective creates the communicator
        torch.cuda.synchronize()
        assert float(probe[0]) == world * (world + 1) / 2
    except Exception as e:                          # RCCL: "Duplicate GPU detected" for two ranks on one device
        sys.stderr.write("BACKEND_REFUSED %s: %s\n" % (args.backend, str(e).splitlines()[0] if str(e) else type(e).__name__))
        os._exit(3)

    import dgcnn
    from dgcnn import _engine as E
    from dgcnn import parallel
    E.DROPOUT_KEEP = 1.0                            # the reference result has its own dropout stream: switched off on both sides
    z = np.load(args.data)
    pts, lab = z["points"], z["labels"]             # (steps, global_batch, N, 3), (steps, global_batch, N)
    tv = dgcnn.trainval(flags_for(dgcnn)).initialize()     # parameters broadcast from rank 0 inside
    assert tv._world == world and tv._rank == rank
    init = dgcnn.ctx().flat_param.clone()
    lo, hi = parallel.shard_bounds(pts.shape[1], rank, world)
    tv.zero_gradients(None)
    losses = []
    for s in range(pts.shape[0]):
        res = tv.accum_gradient(None, [pts[s, lo:hi]], [lab[s, lo:hi]])
        losses.append(float(res[2]))
    local = dgcnn.ctx().flat_grad.clone()
    tv.apply_gradient(None)
    torch.cuda.synchronize()
    c = dgcnn.ctx()
    np.savez(os.path.join(args.out, "rank%d.npz" % rank), grad=c.flat_grad.cpu().numpy(), param=c.flat_param.cpu().numpy(),
             init=init.cpu().numpy(), local_grad=local.cpu().numpy(), losses=np.asarray(losses),
             backend=np.array(dist.get_backend()))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
