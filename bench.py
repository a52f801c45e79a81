#!/usr/bin/env python
"""bench.py -- headline metric of BASELINE.json on MI355X.

  metric : point-clouds/sec, forward+backward (one training micro-step: zero accumulators, forward,
           loss, backward, gradient all-reduce for N>1, Adam) at (B,N,k,C)=(24,2048,20,3),
           3 EdgeConv layers (64,64,128) + FC (512,256), 2 classes, fp32 (BASELINE.json configs[1]).
  step   : one pass of the hot path over one synthetic batch of 24 clouds per GPU (weak scaling:
           per-GPU work fixed); inputs are resident in HBM before the timed region.
  value  : clouds processed by ALL ranks / max-over-ranks wall time of K steps.

Launch:  python bench.py [--gpus N --steps K --warmup W]
   N>1 : python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
            --master-port P bench.py --gpus N --steps K --warmup W      (one rank per GPU, RCCL)
         or plain `python bench.py --gpus N ...` from a bare shell: with no WORLD_SIZE in the environment the script
         re-launches itself as N ranks through torch.distributed.run (self_launch below) and relays rank 0's JSON line.
Prints ONE JSON line on rank 0 (roofline + cpu_baseline objects: see DESIGN.md "Measurement").
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "dynamic-gcnn_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np
import torch

B, N, K_NN, C = 24, 2048, 20, 3
PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: v_mfma_f32_32x32x16_bf16 dense peak
PEAK_HBM_GBS = 8000.0             # HBM3E spec
PEAK_L2_GBS = 34500.0             # MI355X_MICROARCH.md: aggregate L2 -> CU bandwidth (8 XCDs x 4 MB)


def make_flags(dgcnn, train=True):
    return dgcnn.DGCNN_FLAGS(MODEL_NAME="dgcnn", EDGE_CONV_LAYERS=3, EDGE_CONV_FILTERS=[64, 64, 128], FC_LAYERS=2,
                             FC_FILTERS=[512, 256], NUM_CLASS=2, KVALUE=K_NN, NUM_CHANNEL=C, MINIBATCH_SIZE=B,
                             LEARNING_RATE=1e-3, TRAIN=train, DEBUG=False, SEED=1,
                             STATIC_INPUTS=True)     # the resident batch is one tensor nobody else writes: replays do not re-stage it


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown CPU"


def cpu_baseline(clouds=24, iters=5):
    """The reference graph restated op-for-op on torch-CPU (oracle/torch_twin.py), fwd+bwd, timed on
    this host's cores on the SAME batch size as the GPU step (all 24 clouds: BatchNorm statistics over the whole batch, as in
    the reference; MEDIAN of 5 iterations, ~40 s).
    torch's intra-op pool collapses when oversubscribed (256 threads: 0.14 clouds/s, 16 threads: 2.9 clouds/s on the
    2x EPYC 9575F GPU-box host, profiles/r01_cpu_threads.txt), so a short sweep picks the thread count first and
    `cores` reports the count actually used."""
    from oracle import dgcnn_oracle as O
    from oracle import torch_twin as T
    ncpu = os.cpu_count() or 1
    flags = O.Flags(EDGE_CONV_FILTERS=[64, 64, 128], FC_FILTERS=[512, 256], KVALUE=K_NN, TRAIN=True)
    rng = np.random.default_rng(0)
    P = {n: torch.tensor(v, requires_grad=True) for n, v in O.init_params(flags, C, seed=1).items()}
    pts = torch.from_numpy(rng.random((clouds, N, C), dtype=np.float32))
    lab = torch.from_numpy(rng.integers(0, 2, (clouds, N)).astype(np.int64))
    best_nt, best_rate = 1, 0.0
    for nt in sorted({min(n, ncpu) for n in (8, 16, 32)}):
        torch.set_num_threads(nt)
        T.train_step(pts[:1], lab[:1], flags, P)            # thread-pool / allocator warm-up
        t0 = time.perf_counter()
        T.train_step(pts[:2], lab[:2], flags, P)
        rate = 2 / (time.perf_counter() - t0)
        if rate > best_rate:
            best_nt, best_rate = nt, rate
    torch.set_num_threads(best_nt)
    ts = []
    for _ in range(iters):
        t0 = time.perf_counter()
        T.train_step(pts, lab, flags, P)
        ts.append(time.perf_counter() - t0)
    med = float(np.median(ts))
    return {"value": round(clouds / med, 3), "unit": "clouds/s", "cores": best_nt, "kind": "port", "cpu_model": cpu_model(),
            "sample": "all %d clouds of the batch (N=2048,k=20,C=3, same model) fwd+bwd, median of %d iterations (%.1f s each), %d torch "
                      "threads (best of a {8,16,32} sweep) on a %d-CPU host (%s); torch-CPU op-for-op restatement of the "
                      "TF1 graph (TF1 itself cannot run, BASELINE.md 2)" % (clouds, iters, med, best_nt, ncpu, cpu_model())}


def edgeconv_stack_rate(dgcnn, pts, iters=10, warm=3):
    """the EdgeConv stack ALONE (3 layers: k-NN, conv0, BN+ReLU, max/mean over k, conv1 and all their backward
    passes; no head, no loss, no optimizer), forward + backward on the same resident batch.  Runs after the timed region in a
    fresh engine context (its own variables); the nine outputs [max, mean, net] x 3 get a constant upstream gradient."""
    from dgcnn import _engine as E, ops
    dgcnn.reset()
    c = dgcnn.ctx()

    def once():
        c.begin_step()
        c.recording = True
        tensors = ops.repeat_edge_conv(pts, repeat=3, k=K_NN, num_filters=[64, 64, 128], trainable=True)
        for t in tensors:
            v, _, _ = E.as2d(t)
            c.grad(v).fill_(1e-3)
        c.backward()

    for _ in range(warm):
        once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        once()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    return {"value": round(B / dt, 1), "unit": "clouds/s", "ms_per_pass": round(dt * 1e3, 3),
            "what": "3 EdgeConv layers (64,64,128) forward+backward alone on the same 24 clouds: k-NN, conv0 (folded), BN+ReLU, "
                    "max/mean over k, conv1 and their backward; no head / loss / optimizer"}


def self_launch(ngpus):
    """`python bench.py --gpus N` from a bare shell (no torchrun environment): start the N ranks ourselves -- the same
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <argv>`
    command the driver uses -- and hand its exit code back.  Rank 0's JSON line goes to our stdout unchanged."""
    import socket
    import subprocess
    with socket.socket() as s:                      # a free rendezvous port on the loopback interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL across processes needs it on this driver
    env["DGCNN_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


class stdout_to_stderr(object):
    """RCCL prints a version banner to STDOUT when a communicator comes up; this script's contract is ONE JSON line there.  The
    library's own communicator handles it inside dgcnn_comm_init; this guards the torch.distributed fallback path."""

    def __enter__(self):
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


METRIC = "point-clouds/sec fwd+bwd at (B,N,k,C)=(24,2048,20,3), 1/2/4/8 MI355X"


def make_guard(args, rank, world):
    """Multi-rank runs only: every phase that contains a collective for the first time (communicator bring-up, the first EAGER step
    with the gradient all-reduce, the recording of a launch plan that holds the ncclAllReduce, then each timed region) runs under a
    dgcnn.rccl.Deadline.  A hang inside RCCL raises nothing; when the deadline passes rank 0 prints ONE JSON line
    {"error": "<stage>", "value": null, ...} and every rank exits non-zero -- within $DGCNN_BENCH_DEADLINE seconds (default 120;
    the communicator bring-up gets 2.5 x that: topology detection on a cold 8-GPU node), not at the driver's 1800.  Ranks other than 0 wait 10 s longer so that rank 0's line gets out before the launcher reaps it."""
    import contextlib
    if world <= 1:
        return lambda stage, factor=1.0: contextlib.nullcontext()
    from dgcnn import rccl
    seconds = float(os.environ.get("DGCNN_BENCH_DEADLINE", "120"))

    def expire(stage, secs):
        if rank == 0:
            sys.stdout.write(json.dumps({"metric": METRIC, "value": None, "unit": "clouds/s", "n_gpus": world, "steps": args.steps,
                                         "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
                                         "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                                         "error": "%s did not finish within %.0f s" % (stage, secs),
                                         "config": {"workload": "aborted multi-rank run", "parallelism": "dp%d" % world}}) + "\n")
            sys.stdout.flush()

    def guard(stage, factor=1.0):
        st = (lambda: "%s [%s]" % (stage, rccl.LAST_STAGE[0])) if stage.startswith("communicator") else stage
        return rccl.Deadline(st, seconds * factor + (0 if rank == 0 else 10), on_expire=expire)
    return guard


def comm_fields(group, dist, world, per_rank_elapsed, steps, exposed_ms, bucket_elems):
    """The multi-rank evidence of a bench line (rank 0): how many ranks the gradient all-reduce ran on ACCORDING TO RCCL
    (ncclCommCount / ncclCommUserRank through dgcnn_comm_info -- not what the launcher said), every rank's own ms/step, and the
    part of the all-reduce the step had to wait for."""
    src = None
    if group is not None:
        info = group.info()
        backend, ranks, myrank, src = "rccl (dgcnn_allreduce_f32, own communicator)", info["nranks"], info["rank"], \
            "ncclCommCount / ncclCommUserRank (dgcnn_comm_info)"
    elif dist is not None:
        backend = dist.get_backend()
        ranks, myrank = (dist.get_world_size(), dist.get_rank()) if backend == "nccl" else (0, 0)
        src = "torch.distributed (launcher's world size)"
    else:
        backend, ranks, myrank = None, None, None
    ms = [round(t / steps * 1e3, 3) for t in per_rank_elapsed]
    ar = {"bucket_MB": round(bucket_elems * 4 / 1e6, 2),
          "exposed_ms_per_step": {"median": round(exposed_ms[len(exposed_ms) // 2], 4), "max": round(exposed_ms[-1], 4),
                                  "steps": len(exposed_ms)} if exposed_ms else None,
          "note": "event pair on the step's stream around the all-reduce section of apply_gradient in steps right after the "
                  "timed regions, launched the way the timed steps were: with the own communicator the head bucket (97 % of the bytes) is already travelling when that "
                  "section starts, so this is the EXPOSED part of the collective (+ the 1/world scale); at N = 1 there is no "
                  "collective in the step and it reads ~0"}
    return {"collective_backend": backend, "rccl_ranks": ranks, "rccl_rank": myrank, "rccl_ranks_source": src,
            "per_rank_ms_per_step": {"min": min(ms), "max": max(ms), "all": ms}, "allreduce": ar}


def one_rank_rccl_probe(bucket_elems, iters=20):
    """N = 1 carries no collective in its step; so that the line still shows what RCCL says on this box, a ONE-rank communicator
    is created AFTER the timed region through the product path (dlopen, unique id, ncclCommInitRank), asked for its size / rank
    (dgcnn_comm_info) and timed on the gradient bucket, then destroyed.  None when RCCL is not usable here."""
    from dgcnn import rccl
    try:
        g = rccl.Group(rank=0, world=1)
    except Exception as e:          # noqa: BLE001
        return {"error": str(e)[:200]}
    try:
        info = g.info()
        buf = torch.ones(bucket_elems, dtype=torch.float32, device="cuda")
        g.allreduce_sum_(buf)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            g.allreduce_sum_(buf)
        torch.cuda.synchronize()
        info["bucket_allreduce_us"] = round((time.perf_counter() - t0) / iters * 1e6, 1)
        info["source"] = "ncclCommCount / ncclCommUserRank / ncclCommCuDevice on a one-rank communicator created after the timed region"
        return info
    finally:
        g.destroy()


def dry_run(args, rank, world):
    """No GPU: everything of an N-rank bench run that is NOT the model -- launcher environment, the communicator's rendezvous, the
    barrier + max-over-ranks timing of exactly K steps, the replica checksum guard, comm_fields, ONE JSON line on rank 0 -- with a
    stand-in step (a host-side all-reduce of a bucket-sized buffer through the communicator class $DGCNN_BENCH_GROUP names)."""
    import importlib
    mod, cls = os.environ["DGCNN_BENCH_GROUP"].split(":")
    guard = make_guard(args, rank, world)
    with guard("communicator bring-up"):
        group = getattr(importlib.import_module(mod), cls)(rank=rank, world=world)
    bucket = torch.full((1797186,), float(rank + 1))
    state = {"sum": 0.0}

    def step():
        g = bucket.clone()
        group.allreduce_sum_(g)                      # every rank must issue the same collectives
        state["sum"] += float(g[0])

    with guard("first step with the gradient all-reduce"):
        for _ in range(args.warmup):
            step()
        group.barrier()
    with guard("timed region"):
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        group.barrier()
        elapsed = time.perf_counter() - t0
        allv = group.gather_scalars([elapsed, state["sum"]])
    per_rank = [float(x) for x in allv[:, 0]]
    if not bool((allv[:, 1] == allv[0, 1]).all()):
        raise SystemExit("replicas diverged: checksums differ across ranks")
    want = (args.steps + args.warmup) * world * (world + 1) / 2.0
    if abs(state["sum"] - want) > 1e-3:
        raise SystemExit("stand-in all-reduce returned %r, expected %r" % (state["sum"], want))
    comm = comm_fields_dry(group, per_rank, args.steps)
    if rank == 0:
        print(json.dumps({"metric": METRIC, "dry_run": True,
                          "value": None, "unit": "clouds/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                          "ms_per_step": round(max(per_rank) / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                          "vs_baseline": None, "dtype": "f32", "data": "none (dry run: no model, no GPU)",
                          "config": dict(comm, workload="dry run", global_batch=world * B, parallelism="dp%d" % world)}))
    group.destroy()
    return 0


def comm_fields_dry(group, per_rank, steps):
    f = comm_fields(group, None, group.world, per_rank, steps, [], 1797186)
    f["collective_backend"] = "stand-in (%s)" % type(group).__name__
    return f


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernel-table", action="store_true", help="print per-kernel event timings to stderr")
    ap.add_argument("--backend", default="rccl", choices=["rccl", "nccl", "gloo"],
                    help="N > 1: rccl (default) = the library's own RCCL communicator (dgcnn/rccl.py over csrc/comm.cc, no "
                         "torch.distributed); nccl = RCCL through torch.distributed; gloo = only for the single-GPU-box sanity run "
                         "of the N > 1 logic, with --same-device")
    ap.add_argument("--same-device", action="store_true", help="all ranks on cuda:0 (sanity runs only)")
    ap.add_argument("--deterministic", action="store_true", help="(the default since round 5; kept so that old command lines run)")
    ap.add_argument("--atomics", action="store_true", help="the atomics mode instead of the deterministic kernels (32 statistics slots, "
                    "several writers each, unsorted adjacency): 3-4 %% faster, reproducible to ~1e-7 per step only (DESIGN 2)")
    ap.add_argument("--repeats", type=int, default=0, help="the K-step timed region is run this many times back to back and the MEDIAN "
                    "region is reported (all of them in config.per_repeat_ms_per_step); 0 = as many as fill ~2 s, at least 5")
    ap.add_argument("--no-edgeconv-stack", action="store_true", help="skip the EdgeConv-stack-only passes after the timed region "
                    "(profiling runs)")
    ap.add_argument("--graph", default="auto", choices=["0", "1", "plan", "auto"],
                    help="how the forward+backward tower is launched (the reference replays a static TF graph with sess.run): "
                         "plan = recorded launch plan re-issued from one C loop (csrc/plan.cc); 1 = captured HIP graph; 0 = eager "
                         "launches from Python; auto (default) = a few untimed steps of eager and plan during warm-up, then the plan "
                         "unless eager is more than 5 %% faster AND its host enqueue takes less than half a step (an eager step is "
                         "~140 launches from Python: it is the mode a busy host slows down, BENCH_r04)")
    ap.add_argument("--x3-tile", type=int, default=0, help="tools: dgcnn_gemm_x3_tile_override (256 = the 256 x 128 kernel wherever a "
                    "big tile is legal, 512 / 448 = the 256 x 256 / 192 x 256 kernel, 0 = the shipped rule)")
    ap.add_argument("--dry-run", action="store_true",
                    help="plumbing check WITHOUT a GPU (tests/test_bench_launch.py): the launch / rendezvous / fence / timing / "
                         "gather / JSON path of an N-rank run with a stand-in step and the communicator class named by "
                         "$DGCNN_BENCH_GROUP (module:Class); the line says dry_run and measures nothing")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if os.environ.get("DGCNN_BENCH_SELF_LAUNCHED"):
            raise SystemExit("bench.py: launched as a rank but WORLD_SIZE is not set")
        raise SystemExit(self_launch(args.gpus))
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: start one rank per GPU (`python bench.py --gpus N` does it itself)"
                         % (args.gpus, world))
    if args.dry_run:
        return dry_run(args, rank, world)
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if os.environ.get("DGCNN_BENCH_MAIN_PRIORITY"):          # A/B switch (profiles/r06/side_stream_ab.sh): the step's own stream
        torch.cuda.set_stream(torch.cuda.Stream(priority=int(os.environ["DGCNN_BENCH_MAIN_PRIORITY"])))
    import dgcnn
    from dgcnn import _hip as H
    from dgcnn import parallel
    dist = group = None
    guard = make_guard(args, rank, world)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = args.backend
        if backend == "rccl":
            try:
                with guard("communicator bring-up", 2.5):
                    group = parallel.init_rccl(rank=rank, world=world)
            except Exception as e:          # every rank fails alike (no library / rendezvous): fall back together, loudly
                sys.stderr.write("bench.py: own RCCL communicator unavailable (%s); using torch.distributed nccl\n" % e)
                backend = "nccl"
        if group is None:
            import torch.distributed as dist
            with stdout_to_stderr():
                if backend == "nccl":
                    dist.init_process_group(backend="nccl", rank=rank, world_size=world,
                                            device_id=torch.device("cuda", local_rank))
                    dist.barrier()                     # (RCCL initialises lazily: bring the communicator up inside the guard)
                    torch.cuda.synchronize()
                else:
                    dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if args.x3_tile:
        H.load().dgcnn_gemm_x3_tile_override(args.x3_tile)
    flags = make_flags(dgcnn)
    flags.DETERMINISTIC = not args.atomics
    tv = dgcnn.trainval(flags).initialize()
    tv.use_graph({"1": True, "plan": "plan"}.get(args.graph, False))

    rng = np.random.default_rng(rank)                     # per-rank synthetic shard
    pts = torch.from_numpy(rng.random((B, N, C), dtype=np.float32)).cuda()
    lab = torch.from_numpy(rng.integers(0, 2, (B, N)).astype(np.int32)).cuda()

    def step():
        tv.zero_gradients(None)
        res = tv.accum_gradient(None, [pts], [lab], last=True)        # one micro-step per update: the head bucket may go early
        tv.apply_gradient(None)
        return res

    def fence():
        torch.cuda.synchronize()
        if group is not None:
            group.barrier()
        elif dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- warm-up; the SECOND warm-up step (code objects already loaded) is event-timed per kernel to
    # find the dominant one ----
    # N > 1: the FIRST step is eager whatever --graph says (a recorded plan would hold the ncclAllReduce) and runs, fence included,
    # under a deadline; only after it came back clean may a plan be recorded (make_guard)
    first_mode = tv._use_graph
    if world > 1:
        tv.use_graph(False)
    with guard("first eager step with the gradient all-reduce (broadcast of the parameters, head-bucket all-reduce from inside "
               "the backward, rest piece, Adam)"):
        step()
        fence()
    tv.use_graph(first_mode)
    # kernel table from a step with the side stream switched off (launches serialised): per-kernel event times are
    # then those of each kernel ALONE; in the timed region weight-gradient GEMMs overlap the backward chain
    from dgcnn import _engine as E
    side = E.WGRAD_SIDE_STREAM
    E.WGRAD_SIDE_STREAM = False
    H.TIMER = H.Timer()
    step()
    table = H.TIMER.summary()
    H.TIMER = None
    E.WGRAD_SIDE_STREAM = side
    # the kernel with the largest time in that step.  The three operand layouts of the head GEMM (forward / data gradient / weight
    # gradient: the same flops each) are within a few per cent of each other, so "largest" flipped from run to run; among tags within
    # 5 % of the largest the choice is fixed (first in name order) so that successive bench lines report the same kernel
    tmax = max(v[1] for v in table.values())
    dominant = sorted(t for t in table if table[t][1] >= 0.95 * tmax)[0]
    for _ in range(max(args.warmup - 2, 0)):
        step()
    calib = None
    if args.graph == "auto":                    # untimed calibration: which launch mode is faster on THIS host + GPU
        def rate(n=6):
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                step()
            t_host = time.perf_counter() - t
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / n, t_host / n
        tv.use_graph(False)
        t_eager, h_eager = rate()
        tv.use_graph("plan")
        with guard("recording the launch plan (holds the ncclAllReduce) and its first replay"):
            step()
            step()                              # sighting + recording
            step()
            fence()
        rate(12)                                # (the first second of replay runs 3-4 % slow: profiles/r05/bench_stall.txt B)
        t_plan, h_plan = rate()
        # the plan unless eager wins clearly AND has host headroom: an eager step is ~140 launches from Python, the mode that a
        # busy host stretches (BENCH_r04: calibration 4.75 ms, timed region 6.69 ms with 5.4 ms of host enqueue)
        v = [t_eager, t_plan, h_eager]
        if group is not None:                   # every rank must make the same choice
            v = [float(x) for x in group.gather_scalars(v).max(0).values]
        elif dist is not None:
            vt = torch.tensor(v, dtype=torch.float64, device="cuda")
            dist.all_reduce(vt, op=dist.ReduceOp.MAX)
            v = [float(x) for x in vt]
        eager_wins = v[0] < 0.95 * v[1] and v[2] < 0.5 * v[0]
        tv.use_graph(False if eager_wins else "plan")
        calib = {"eager_ms": round(t_eager * 1e3, 3), "plan_ms": round(t_plan * 1e3, 3),
                 "eager_host_enqueue_ms": round(h_eager * 1e3, 3), "plan_host_enqueue_ms": round(h_plan * 1e3, 3),
                 "rule": "plan unless eager_ms < 0.95 plan_ms and eager_host_enqueue_ms < 0.5 eager_ms (max over ranks)"}
    launch_mode = tv._use_graph
    mode_name = {"plan": "launch-plan replay", True: "hip-graph replay", False: "eager"}[launch_mode]

    # ---- timed regions: R times EXACTLY K steps, each bracketed by fence() on both sides, no instrumentation (an event pair
    # around every launch of the dominant kernel perturbs what it measures: with the plane GEMMs +0.1 .. +1.9 ms per step,
    # profiles/r03/bench_events.txt).  The MEDIAN region is the reported one; all R are in the line. ----
    H.TIMER = None
    step()
    fence()
    t0 = time.perf_counter()
    step()
    fence()
    t_probe = time.perf_counter() - t0
    R = args.repeats if args.repeats > 0 else max(5, min(25, int(2.0 / max(args.steps * t_probe, 1e-4)) + 1))
    if group is not None:                       # (one R for every rank)
        R = int(group.gather_scalars([float(R)]).max(0).values[0])
    elif dist is not None:
        vt = torch.tensor([float(R)], dtype=torch.float64, device="cuda")
        dist.all_reduce(vt, op=dist.ReduceOp.MAX)
        R = int(vt[0])
    regions, issues = [], []
    for _ in range(R):
        with guard("timed region of %d steps" % args.steps):
            fence()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                res = step()
            issues.append(time.perf_counter() - t0)  # host time to enqueue K steps (launch-bound check)
            fence()
            regions.append(time.perf_counter() - t0)
    tail_guard = guard("post-region measurements (event-timed eager steps, exposed all-reduce, gathers over the ranks)", 2.0)
    tail_guard.__enter__()
    # ---- the dominant kernel, live: HIP events (on the launch stream) around its launches in eager steps that follow the
    # timed regions immediately (same process, same buffers, same two-stream schedule) ----
    tv.use_graph(False)
    H.TIMER = H.Timer(watch={dominant})
    for _ in range(max(args.steps // 4, 3)):
        step()
    dom = H.TIMER.summary()[dominant]
    H.TIMER = None
    # ---- the collective, live: event pairs around the part of the all-reduce the step's stream has to WAIT for (the head bucket
    # travels under the EdgeConv backward; what is left is the exposed cost) -- in the launch mode of the timed regions ----
    tv.use_graph(launch_mode)
    tv._ar_events = []
    for _ in range(max(args.steps // 4, 3)):
        step()
    torch.cuda.synchronize()
    exposed_ms = sorted(a.elapsed_time(b) for a, b in tv._ar_events)
    tv._ar_events = None
    loss = float(res[2])
    plan_info = tv.launch_plan_info()

    # replicas must hold identical parameters after identical Adam steps on the all-reduced gradient
    chk = torch.stack([dgcnn.ctx().flat_param.double().sum(), dgcnn.ctx().flat_param.double().abs().sum()])
    per_rank_regions = [regions]
    if group is not None:
        allv = group.gather_scalars(regions + [float(x) for x in chk.float()])      # (world, R + 2)
        per_rank_regions = [[float(x) for x in row[:R]] for row in allv]
        if not bool((allv[:, R:] == allv[0, R:]).all()):
            raise SystemExit("replicas diverged: parameter checksums differ across ranks")
    elif dist is not None:
        box = [torch.zeros(R, dtype=torch.float64, device="cuda") for _ in range(world)]
        dist.all_gather(box, torch.tensor(regions, dtype=torch.float64, device="cuda"))
        per_rank_regions = [[float(v) for v in x] for x in box]
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if not torch.equal(lo, hi):
            raise SystemExit("replicas diverged: parameter checksums differ across ranks")
    tail_guard.__exit__(None, None, None)
    region_max = [max(r[i] for r in per_rank_regions) for i in range(R)]          # per region: the slowest rank
    order = sorted(range(R), key=lambda i: region_max[i])
    mid = order[R // 2]                                                            # the median region
    elapsed = region_max[mid]
    per_rank = [r[mid] for r in per_rank_regions]
    t_issue = issues[mid]

    # the group the gradient all-reduce ran in: backend "nccl" IS RCCL on ROCm (gloo only in the one-device sanity run)
    comm = comm_fields(group, dist, world, per_rank, args.steps, exposed_ms, dgcnn.ctx().flat_grad.numel())
    arith_name = {0: "native fp32 MFMA", 6: "exact 3-way bf16 split, 6 partial products on the bf16 MFMA pipe",
                  9: "exact 3-way bf16 split, 9 partial products on the bf16 MFMA pipe"}[H.gemm_arith()]
    if rank == 0:
        is_gemm = dominant.startswith("gemm") or dominant.startswith("knn")
        launches, secs, work = dom[:3]
        if is_gemm:
            achieved = work / secs / 1e12          # algorithmic fp32 flops (2 M N K per GEMM)
            peak, note = PEAK_F32_MFMA_TFLOPS, "fp32 MFMA dense peak"
            if dominant.startswith("gemm_x3"):      # gemm_x3_kernel / gemm_x3w2_kernel
                # fp32 GEMM computed as NP exact bf16 partial products on the bf16 matrix pipe (gemm_x3.hip):
                # the ceiling for ALGORITHMIC fp32 flops is the bf16 dense peak / NP
                nprod = int(dominant.rstrip(">").split("bf16x")[1])
                peak = round(PEAK_BF16_MFMA_TFLOPS / nprod, 1)
                note = "bf16 MFMA dense peak %.0f / %d partial products per fp32 product" % (PEAK_BF16_MFMA_TFLOPS, nprod)
            elif dominant.startswith("gemm_pl"):    # plane GEMM (gemm_pl.hip): 3 bf16 planes -> 6 products, 2 fp16 planes -> 3
                nprod = 3 if "f16x2" in dominant else 6
                peak = round(PEAK_BF16_MFMA_TFLOPS / nprod, 1)
                note = ("16-bit MFMA dense peak %.0f (bf16 = fp16 rate) / %d partial products per fp32 product"
                        % (PEAK_BF16_MFMA_TFLOPS, nprod))
            roof = {"kernel": dominant, "bound": "mfma" if dominant.startswith("gemm") else "valu",
                    "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(achieved / peak, 4), "traffic": None,
                    "launches": launches, "avg_us": round(secs / launches * 1e6, 1), "peak_note": note}
        else:
            achieved = work / secs / 1e9
            roof = {"kernel": dominant, "bound": "hbm", "achieved": round(achieved, 1), "peak": PEAK_HBM_GBS,
                    "unit": "GB/s", "frac": round(achieved / PEAK_HBM_GBS, 4), "traffic": None,
                    "launches": launches, "avg_us": round(secs / launches * 1e6, 1)}
        if side and is_gemm:
            n1, s1, w1 = table[dominant][:3]
            roof["achieved_serialized"] = round(w1 / s1 / 1e12, 2)
            roof["frac_serialized"] = round(w1 / s1 / 1e12 / roof["peak"], 4)
            roof["note"] = ("achieved/frac: HIP events around every launch of this kernel in eager steps right after the timed "
                            "region, where weight-gradient GEMMs run concurrently on a second stream and stretch it; "
                            "*_serialized: the same kernel timed in a warm-up step with the side stream off")
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                roof["traffic"] = json.load(open(pmc)).get(dominant)
            except Exception:
                pass
        # north_star: "rocprof HBM GB/s on the distance/gather kernels and MFMA utilisation on the edge MLP": the other kernel
        # families of the step, each timed ALONE (the serialised warm-up step above), against the ceiling that bounds it
        extra = []
        for t, (n, sec, work, nbytes) in sorted(table.items(), key=lambda kv: -kv[1][1]):
            if t == dominant or sec <= 0:
                continue
            if t.startswith("knn"):
                # a k-NN CALL (dgcnn_knn_f32 / dgcnn_knn_seeded_f32) is several kernels; the tag lists them and names the pipe the
                # 2 B N^2 C distance flops run on: the raw-coordinate kernel forms them on the fp32 VALU, the list-keeping feature kernel
                # on the fp32 MFMA, the append-form scan as a conservative bf16-MFMA filter with an exact fp32 re-check of the survivors
                c_in = int(t.split("<C")[1].split(",")[0])
                if "knn_bf16a_kernel" in t:
                    pipe, pk = "bf16 MFMA filter (+ exact fp32 VALU re-check of the survivors); staging / barrier bound", PEAK_BF16_MFMA_TFLOPS
                elif c_in <= 4:
                    pipe, pk = "fp32 VALU (distances and sorted inserts share the vector pipe)", PEAK_F32_MFMA_TFLOPS
                else:
                    pipe, pk = "fp32 MFMA (blocks VALU issue: shares the vector pipe with the selection)", PEAK_F32_MFMA_TFLOPS
                extra.append({"kernel": t, "bound": pipe,
                              "achieved": round(work / sec / 1e12, 2), "peak": pk, "unit": "TFLOP/s",
                              "frac": round(work / sec / 1e12 / pk, 4), "launches": n,
                              "avg_us": round(sec / n * 1e6, 1),
                              "materialised_equiv_GBs": round(n * 2.0 * B * N * N * 4 / sec / 1e9, 1),
                              "note": "one CALL = the kernels in the tag; flops = 2 B N^2 C (C = %d padded); materialised_equiv = the "
                                      "2 B N^2 4-byte write+read of the reference's (B,N,N) distance tensor that never touches HBM here" % c_in})
            elif t.startswith("gemm"):
                pk = PEAK_F32_MFMA_TFLOPS
                if "bf16x" in t:
                    pk = PEAK_BF16_MFMA_TFLOPS / int(t.rstrip(">").split("bf16x")[1])
                if t.startswith("gemm_pl"):
                    pk = PEAK_BF16_MFMA_TFLOPS / (3 if "f16x2" in t else 6)
                # which roof: flops per compulsory byte against the ridge of this arithmetic (peak flops / HBM peak)
                if nbytes > 0 and work / nbytes < pk * 1e12 / (PEAK_HBM_GBS * 1e9):
                    extra.append({"kernel": t, "bound": "hbm", "achieved": round(nbytes / sec / 1e9, 1), "peak": PEAK_HBM_GBS,
                                  "unit": "GB/s", "frac": round(nbytes / sec / 1e9 / PEAK_HBM_GBS, 4), "launches": n,
                                  "avg_us": round(sec / n * 1e6, 1), "mfma_frac": round(work / sec / 1e12 / pk, 4),
                                  "note": "short reduction: %.0f flops per operand/result byte, below the ridge of %.0f" % (
                                      work / nbytes, pk * 1e12 / (PEAK_HBM_GBS * 1e9))})
                else:
                    extra.append({"kernel": t, "bound": "mfma", "achieved": round(work / sec / 1e12, 2), "peak": round(pk, 1),
                                  "unit": "TFLOP/s", "frac": round(work / sec / 1e12 / pk, 4), "launches": n,
                                  "avg_us": round(sec / n * 1e6, 1)})
            elif nbytes > 0:
                # passes that re-form y = V[neighbour] + U[point] per edge instead of streaming a materialised (B N k, F) tensor: the
                # bytes that move are rows GATHERED from the point-level [U | V] buffer (25-50 MB: resident in the XCDs' L2s), so the
                # roof is the L2 -> CU bandwidth, not HBM; `hbm_GBs` is what the pass moves through HBM (outputs, indices)
                extra.append({"kernel": t, "bound": "l2-gather", "achieved": round(nbytes / sec / 1e9, 1), "peak": PEAK_L2_GBS,
                              "unit": "GB/s", "frac": round(nbytes / sec / 1e9 / PEAK_L2_GBS, 4), "launches": n,
                              "avg_us": round(sec / n * 1e6, 1), "hbm_GBs": round(work / sec / 1e9, 1),
                              "note": "gathered bytes = 4 (R k F + R F) per launch (one F-float row of V per edge + U once per point)"})
            elif work > 0:
                extra.append({"kernel": t, "bound": "hbm", "achieved": round(work / sec / 1e9, 1), "peak": PEAK_HBM_GBS,
                              "unit": "GB/s", "frac": round(work / sec / 1e9 / PEAK_HBM_GBS, 4), "launches": n,
                              "avg_us": round(sec / n * 1e6, 1)})
        # the step as a whole: executed fp32-equivalent flops (GEMMs 2 M N K as launched -- the fold and the FC0 global-feature split
        # already taken out -- plus the k-NN distance products) and algorithmic HBM bytes (every tagged launch's compulsory operand +
        # result bytes) per step, over the MEDIAN region's ms_per_step, against both peaks
        st_flops = sum(v[2] for t, v in table.items() if t.startswith("gemm") or t.startswith("knn"))
        st_bytes = sum((v[3] if t.startswith("gemm") else (v[2] if not t.startswith("knn") else 0.0)) for t, v in table.items())
        st_ms = elapsed / args.steps * 1e3
        roof["step"] = {"executed_GFLOP": round(st_flops / 1e9, 1), "TFLOP/s": round(st_flops / st_ms / 1e9, 1),
                        "frac_of_mfma_peak": round(st_flops / st_ms / 1e9 / (PEAK_BF16_MFMA_TFLOPS / 6), 4),
                        "mfma_peak": round(PEAK_BF16_MFMA_TFLOPS / 6, 1),
                        "hbm_GB": round(st_bytes / 1e9, 2), "GB/s": round(st_bytes / st_ms / 1e6, 1),
                        "frac_of_hbm_peak": round(st_bytes / st_ms / 1e6 / PEAK_HBM_GBS, 4),
                        "kernel_ms_serialised": round(sum(v[1] for v in table.values()) * 1e3, 3), "ms_per_step": round(st_ms, 3),
                        "note": "sums over the tagged launches of one step (event-timed with the side stream off): fp32-equivalent "
                                "flops of the GEMMs and k-NN products as executed, compulsory operand/result bytes of every launch"}
        if args.kernel_table:
            tot = sum(v[1] for v in table.values())
            for t, (n, s, w, _b) in sorted(table.items(), key=lambda kv: -kv[1][1]):
                sys.stderr.write("%-46s launches %3d  %8.3f ms  %5.1f%%  work/s %.3e\n" % (t, n, s * 1e3, 100 * s / tot, w / s))
        out = {
            "metric": METRIC,
            "value": round(world * B * args.steps / elapsed, 2),
            "unit": "clouds/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: B=24/GPU N=2048 k=20 C=3, 3 EdgeConv (64,64,128) + merged 1024 + "
                                   "FC (512,256) + Final, fp32 in/out/accumulate (GEMM arithmetic: %s), dropout on; step = zero-grad + fwd + loss + bwd + "
                                   "(RCCL all-reduce) + Adam" % arith_name,
                       "global_batch": world * B, "points_per_cloud": N, "parallelism": "dp%d" % world,
                       "collective_backend": comm["collective_backend"], "rccl_ranks": comm["rccl_ranks"],
                       "rccl_rank": comm["rccl_rank"], "rccl_ranks_source": comm["rccl_ranks_source"],
                       "per_rank_ms_per_step": comm["per_rank_ms_per_step"], "allreduce": comm["allreduce"],
                       "final_loss": round(loss, 5), "host_enqueue_ms_per_step": round(t_issue / args.steps * 1e3, 3),
                       "launch_mode": mode_name, "launch_mode_calibration": calib,
                       "deterministic": not args.atomics,
                       "repeats": R, "per_repeat_ms_per_step": [round(t / args.steps * 1e3, 3) for t in region_max],
                       "repeat_spread": round((max(region_max) - min(region_max)) / elapsed, 4),
                       "timing": "the K-step region (fence = device synchronize + barrier on both sides) is run `repeats` times "
                                 "back to back; ms_per_step / value are those of the MEDIAN region (max over ranks per region)",
                       "launches_per_step": (dict(plan_info[0], outside_plan="zero-gradient memset, dropout-seed fill, softmax / "
                                                  "[loss, accuracy] copies, (all-reduce rest piece + scale), Adam")
                                             if plan_info else None)},
            "roofline": roof,
            "roofline_extra": extra[:14],
        }
        if world == 1:
            probe = one_rank_rccl_probe(dgcnn.ctx().flat_grad.numel())
            out["config"]["rccl_probe"] = probe
            if probe and "nranks" in probe:
                out["config"].update(rccl_ranks=probe["nranks"], rccl_rank=probe["rank"], rccl_ranks_source=probe["source"])
        if world == 1 and not args.no_edgeconv_stack:
            out["edgeconv_stack"] = edgeconv_stack_rate(dgcnn, pts)     # after the timed region; resets the engine context
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if group is not None:
        parallel.shutdown_rccl()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
