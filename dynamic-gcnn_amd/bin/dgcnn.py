#!/usr/bin/env python
"""Command-line front end (reference bin/dgcnn.py): `dgcnn.py {train,inference,iotest} [options]`.
For N GPUs launch one process per GPU:
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 bin/dgcnn.py train ...
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main(argv=None):
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # data path (parameter broadcast, gradient all-reduce): the library's own RCCL communicator (dgcnn/rccl.py);
        # control plane of the run loops (resume iteration, output gathering: host objects): a gloo group
        dist.init_process_group(backend="gloo")
        from dgcnn import parallel
        if os.environ.get("DGCNN_COLLECTIVE", "rccl") == "rccl":
            parallel.init_rccl()
    from dgcnn import DGCNN_FLAGS
    DGCNN_FLAGS().parse_args(argv)
    if world > 1:
        parallel.shutdown_rccl()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
