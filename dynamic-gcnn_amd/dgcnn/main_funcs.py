"""Run loops around the hot path: the caller side of dgcnn/trainval (reference dgcnn/main_funcs.py:9-306).

Same entry points and log formats as the reference -- `train(flags)`, `inference(flags)`,
`iotest(flags)`, `prepare(flags)`, `train_loop`, `inference_loop`, `iteration_from_filename`,
`round_decimals`; `train_log-%07d.csv` / `inference_log-%07d.csv` with the reference's columns;
checkpoints `<WEIGHT_PREFIX>-<iteration>` every CHECKPOINT_STEP with at most CHECKPOINT_NUM kept --
re-thought for one process per GPU:

  * BATCH_SIZE is the GLOBAL batch of one optimizer step.  Every replica draws the same batch from
    its data source (same SEED), keeps clouds [lo,hi) = parallel.shard_bounds(BATCH_SIZE, rank, world)
    and walks them in micro-steps of MINIBATCH_SIZE x len(GPUS) clouds (main_funcs.py:146-164).  One
    all-reduce per optimizer step happens inside trainer.apply_gradient.
  * loss / accuracy in the log are means over micro-steps and, for N>1, over replicas.
  * rank 0 alone writes CSV lines, stdout reports, checkpoints and stored softmax.
  * no tf.Session / TensorBoard: `handlers.sess` stays None and SUMMARY_STEP summaries go to a second
    CSV (`summary-%07d.csv`: iter,accuracy,loss) instead of an event file.
"""
from __future__ import annotations

import datetime
import glob
import os
import sys
import time

import numpy as np
import torch

from . import parallel
from .iotool import io_factory
from .trainval import trainval

TRAIN_COLUMNS = ("iter,epoch,titer,ttrain,tio,tsave,tsummary,"
                 "tsumiter,tsumtrain,tsumio,tsumsave,tsumsummary,loss,accuracy")
INFERENCE_COLUMNS = "iter,epoch,titer,tinference,tio,tsumiter,tsuminference,tsumio,loss,accuracy"


def round_decimals(val, digits):
    scale = 10.0 ** digits
    return int(val * scale + 0.5) / scale


def iteration_from_filename(file_name):
    """main_funcs.py:13-14 (an optional .npz extension is ignored)."""
    stem = file_name[:-4] if file_name.endswith(".npz") else file_name
    return int(stem.split("-")[-1])


class Saver(object):
    """The slice of tf.train.Saver the loops use (main_funcs.py:83-84,91,183): save with a
    global_step suffix, keep the newest `max_to_keep`, restore by name."""

    def __init__(self, max_to_keep=10, keep_checkpoint_every_n_hours=0.0, clock=time.time):
        """keep_checkpoint_every_n_hours (main_funcs.py:82 passes CHECKPOINT_HOUR): as in tf.train.Saver, a checkpoint that
        falls out of the `max_to_keep` window is KEPT for good when at least that many hours have passed since the last one
        kept this way (0 / negative: off)."""
        self._keep = int(max_to_keep)
        self._every = float(keep_checkpoint_every_n_hours or 0.0) * 3600.0
        self._clock = clock
        self._next_keep = self._clock() + self._every
        self._written = []            # [(name, time written)] still inside the retention window
        self.kept_forever = []

    def save(self, trainer, prefix, global_step):
        name = trainer.save(prefix, global_step)
        if trainer._rank == 0:
            self._written.append((name, self._clock()))
            while self._keep > 0 and len(self._written) > self._keep:
                old, t_old = self._written.pop(0)
                # tf.train.Saver._MaybeDeleteOldCheckpoints: `p[1] > self._next_checkpoint_time` and
                # `self._next_checkpoint_time += keep_checkpoint_every_n_hours * 3600` (the deadline advances by whole periods;
                # after a long gap the next few evicted checkpoints are all kept until it has caught up, as in TF)
                if self._every > 0.0 and t_old > self._next_keep:
                    self._next_keep += self._every
                    self.kept_forever.append(old)
                    continue
                if os.path.exists(old + ".npz"):
                    os.remove(old + ".npz")
            with open(os.path.join(os.path.dirname(name) or ".", "checkpoint"), "w") as f:
                f.write('model_checkpoint_path: "%s"\n' % os.path.basename(name))
        return name

    def restore(self, trainer, path):
        trainer.restore(path)
        return iteration_from_filename(path)


class Handlers(object):
    sess = None
    data_io = None
    csv_logger = None
    weight_io = None
    train_logger = None
    trainer = None
    iteration = 0
    rank = 0
    world = 1


def iotest(flags):
    io = io_factory(flags)
    io.initialize()
    seen, total = 0, io.num_entries()
    while seen < total:
        idx, data, label, weight = io.next()
        print("%d/%d ... %s %s%s%s" % (seen, total, idx, data[0].shape,
                                      "" if label is None else " %s" % (label[0].shape,),
                                      "" if weight is None else " %s" % (weight[0].shape,)))
        seen += len(data)
    io.finalize()


def train(flags):
    flags.TRAIN = True
    handlers = prepare(flags)
    train_loop(flags, handlers)
    return handlers


def inference(flags):
    flags.TRAIN = False
    handlers = prepare(flags)
    inference_loop(flags, handlers)
    return handlers


def _per_step(flags, world):
    return int(flags.MINIBATCH_SIZE) * len(flags.GPUS) * world


def prepare(flags):
    h = Handlers()
    dist, h.rank, h.world = parallel.dist_state()
    if dist is not None and h.world > 1:          # one seed for all replicas: same batches, same dropout stream
        box = [int(flags.SEED)]
        dist.broadcast_object_list(box, src=0)
        flags.SEED = box[0]
    if int(flags.BATCH_SIZE) % _per_step(flags, h.world):
        sys.stderr.write("--batch_size (%d) must be a multiple of replicas (%d) * --gpus (%d) * --minibatch_size (%d)\n"
                         % (flags.BATCH_SIZE, h.world, len(flags.GPUS), flags.MINIBATCH_SIZE))
        sys.exit(1)

    h.data_io = io_factory(flags)
    h.data_io.initialize()
    flags.NUM_CHANNEL = h.data_io.num_channels()

    h.trainer = trainval(flags)
    h.trainer.initialize()
    h.weight_io = Saver(max_to_keep=getattr(flags, "CHECKPOINT_NUM", 10),
                        keep_checkpoint_every_n_hours=getattr(flags, "CHECKPOINT_HOUR", 0.0))

    h.iteration = 0
    loaded = 0
    if getattr(flags, "MODEL_PATH", ""):
        loaded = h.weight_io.restore(h.trainer, flags.MODEL_PATH)
        if flags.TRAIN:
            h.iteration = loaded + 1

    if getattr(flags, "LOG_DIR", "") and h.rank == 0:
        if not os.path.exists(flags.LOG_DIR):
            os.makedirs(flags.LOG_DIR)
        stem = "train_log" if flags.TRAIN else "inference_log"
        h.csv_logger = open("%s/%s-%07d.csv" % (flags.LOG_DIR, stem, loaded), "w")
        if flags.TRAIN:
            h.train_logger = open("%s/summary-%07d.csv" % (flags.LOG_DIR, loaded), "w")
            h.train_logger.write("iter,accuracy,loss\n")
    return h


def _tower_array(chunk, what):
    """Rows [at, at+MINIBATCH_SIZE) of a batch as ONE array (MINIBATCH_SIZE, N, ...), the shape the reference's
    placeholders take (trainval.py:31-35: (MINIBATCH_SIZE, None, NUM_CHANNEL)).  Dense sources hand over an array
    slice; variable-N sources (io_larcv-like lists, iotool.py:127-150) a list of per-cloud arrays, which only stacks
    when the clouds agree in N -- the production regime is `-mbs 1` (scripts/lsf/train_dgcnn.sh:28)."""
    if isinstance(chunk, np.ndarray):
        return chunk
    if len({c.shape[0] for c in chunk}) != 1:
        raise ValueError("clouds of %s in one micro-batch have different point counts %s: a variable-N source needs "
                         "--minibatch_size 1" % (what, [c.shape[0] for c in chunk]))
    return np.stack(chunk)


def _micro_batches(flags, h, data, label, weight):
    """Yield (data_v, label_v, weight_v) tower lists covering this replica's share of the batch."""
    lo, hi = parallel.shard_bounds(int(flags.BATCH_SIZE), h.rank, h.world)
    mbs = int(flags.MINIBATCH_SIZE)
    at = lo
    while at < hi:
        dv, lv, wv = [], None if label is None else [], None if weight is None else []
        for _ in flags.GPUS:
            dv.append(_tower_array(data[at:at + mbs], "data"))
            if lv is not None:
                lv.append(_tower_array(label[at:at + mbs], "label"))
            if wv is not None:
                wv.append(_tower_array(weight[at:at + mbs], "weight"))
            at += mbs
        yield dv, lv, wv


def _replica_mean(h, values, local_only=False):
    """Mean over micro-steps (device scalars, one sync) and over replicas (local_only: this replica's mean, no collective --
    the error path, where the other replicas may not be there to answer)."""
    if not values:
        return -1.0
    t = torch.stack([torch.as_tensor(v, dtype=torch.float32).reshape(()) for v in values]).mean()
    if local_only:
        return float(t)
    dist, _, world = parallel.dist_state()
    if world > 1 and parallel.rccl_group() is not None and t.is_cuda:
        t = parallel.rccl_group().allreduce_sum_(t.clone().reshape(1)) / world
    elif dist is not None and world > 1:
        t = t.clone()
        dist.all_reduce(t)
        t = t / world
    return float(t)


def _gather_rows(h, softmax, idx):
    """Softmax rows of the whole batch on rank 0, in batch order.  Every replica holds the towers of its own shard
    [lo, hi) (main_funcs.py:264-268 stores from the one process the reference has).  The entry indices `idx` must be the
    same on every rank (each rank draws them from an identically seeded source): checked against rank 0's.  Equal-shaped
    rows travel as ONE all_gather on the torch.distributed CONTROL-PLANE group (bin/dgcnn.py creates it with gloo next to the
    RCCL communicator: device tensors are staged through the host there -- output rows are not on the training hot path);
    ragged batches (-np -1 -mbs 1) go through a host object gather."""
    if h.world == 1:
        return [row for tower in softmax for row in tower.cpu().numpy()]
    dist = parallel.dist_state()[0]
    if dist is None:
        raise RuntimeError("OUTPUT_FILE with more than one replica needs torch.distributed for the control plane "
                           "(bin/dgcnn.py initialises a gloo group next to the RCCL communicator)")
    dev = softmax[0].device
    mine = torch.as_tensor(np.asarray(idx, dtype=np.int64), device=dev)
    ref = mine.clone()
    dist.broadcast(ref, src=0)
    if not torch.equal(ref, mine):
        raise RuntimeError("inference_loop: rank %d drew different entry indices than rank 0 (data sources must be seeded "
                           "identically on every replica)" % h.rank)
    shapes = {tuple(t.shape[1:]) for t in softmax}
    same = torch.tensor([1 if len(shapes) == 1 else 0], device=dev)
    dist.all_reduce(same, op=dist.ReduceOp.MIN)
    if int(same.item()):
        local = torch.cat([t.reshape((-1,) + t.shape[1:]) for t in softmax]).contiguous()
        box = [torch.empty_like(local) for _ in range(h.world)]
        dist.all_gather(box, local)
        return [row for part in box for row in part.cpu().numpy()] if h.rank == 0 else []
    rows = [row for tower in softmax for row in tower.cpu().numpy()]
    box = [None] * h.world if h.rank == 0 else None
    dist.gather_object(rows, box, dst=0)
    return [r for part in box for r in part] if h.rank == 0 else []


def train_loop(flags, h):
    """main_funcs.py:96-167.  The loss / accuracy of iteration t are read back from the device AFTER iteration t + 1 has been put on
    the queue (`resolve` below): the host never drains the GPU between two steps, which cost 0.35 ms of a 5.07 ms iteration at
    configs[1] (profiles/r03_cli_train).  The CSV rows and report lines are the reference's, one iteration late; `titer` is the
    period of the loop (start of iteration t to start of t + 1 -- in steady state the device time of a step, since each pass
    blocks on the previous iteration's scalars), `ttrain` that period minus IO / save / summary time.  An iteration that writes a
    checkpoint is logged at once (never a snapshot without its rows), and the row of the last completed iteration is written
    even when the loop leaves through an exception."""
    if h.csv_logger:
        h.csv_logger.write(TRAIN_COLUMNS + "\n")
    tsum = dict(iter=0.0, train=0.0, io=0.0, save=0.0, summary=0.0)

    def resolve(p, t_next, local_only=False):
        """Finish iteration p: scalars to the host (syncs on p's kernels only), summary / CSV / report lines."""
        it = p["it"]
        loss, acc = _replica_mean(h, p["losses"], local_only), _replica_mean(h, p["accs"], local_only)
        t0 = time.time()
        if p["summarize"]:
            h.train_logger.write("%d,%g,%g\n" % (it, acc, loss))
        t_summary = time.time() - t0
        t_spent = t_next - p["t_iter"]
        t_train = max(t_spent - p["t_io"] - p["t_save"] - t_summary, 0.0)
        for k, v in (("iter", t_spent), ("train", t_train), ("io", p["t_io"]), ("save", p["t_save"]), ("summary", t_summary)):
            tsum[k] += v
        if h.csv_logger:
            h.csv_logger.write("%d,%g,%g,%g,%g,%g,%g,%g,%g,%g,%g,%g,%g,%g\n" % (
                it, p["epoch"], t_spent, t_train, p["t_io"], p["t_save"], t_summary,
                tsum["iter"], tsum["train"], tsum["io"], tsum["save"], tsum["summary"], loss, acc))
        if p["report"] and h.rank == 0:
            mem = torch.cuda.max_memory_allocated() if torch.cuda.is_available() else 0
            print("Iteration %d (epoch %g) @ %s ... train time fraction %g%% max mem. %g ... loss %g accuracy %g"
                  % (it, round_decimals(p["epoch"], 2), p["stamp"], round_decimals(t_train / max(t_spent, 1e-12) * 100.0, 2), mem,
                     round_decimals(loss, 4), round_decimals(acc, 4)))
            sys.stdout.flush()
            if h.csv_logger:
                h.csv_logger.flush()
            if h.train_logger:
                h.train_logger.flush()

    pending = None
    try:
        while h.iteration < int(flags.ITERATION):
            it = h.iteration
            t_iter = time.time()
            p = dict(it=it, t_iter=t_iter, stamp=datetime.datetime.fromtimestamp(t_iter).strftime("%Y-%m-%d %H:%M:%S"),
                     report=bool(flags.REPORT_STEP) and (it + 1) % flags.REPORT_STEP == 0,
                     summarize=bool(getattr(flags, "SUMMARY_STEP", 0)) and h.train_logger is not None and
                     (it + 1) % flags.SUMMARY_STEP == 0,
                     epoch=it * float(flags.BATCH_SIZE) / h.data_io.num_entries(), t_save=0.0)
            checkpoint = bool(getattr(flags, "CHECKPOINT_STEP", 0)) and bool(getattr(flags, "WEIGHT_PREFIX", "")) and \
                (it + 1) % flags.CHECKPOINT_STEP == 0                           # main_funcs.py:131: no prefix, no snapshot

            t0 = time.time()
            idx, data, label, weight = h.data_io.next()
            p["t_io"] = time.time() - t0

            losses, accs = [], []
            h.trainer.zero_gradients(h.sess)
            micro = list(_micro_batches(flags, h, data, label, weight))
            for mi, (dv, lv, wv) in enumerate(micro):
                # (last: the head's gradient bucket may start its all-reduce under this micro-step's EdgeConv backward)
                res = h.trainer.accum_gradient(h.sess, dv, lv, wv, summary=p["summarize"], last=(mi == len(micro) - 1))
                accs.append(res[1])
                losses.append(res[2])
            h.trainer.apply_gradient(h.sess)
            p["losses"], p["accs"] = losses, accs

            # iteration `it` is on the queue: now read back the iteration before it
            if pending is not None:
                done, pending = pending, None
                resolve(done, t_iter)
            pending = p

            if checkpoint:
                # a snapshot is never on disk ahead of its own log rows: the pending iteration (this one) is written first
                # (the save copies the state to the host and drains the queue anyway)
                t0 = time.time()
                h.weight_io.save(h.trainer, flags.WEIGHT_PREFIX, global_step=it)
                p["t_save"] = time.time() - t0
                pending = None
                resolve(p, time.time())
            h.iteration += 1
        if pending is not None:
            done, pending = pending, None
            resolve(done, time.time())
    finally:
        # a step that raised (or an interrupt) must not lose the row of the last COMPLETED iteration; its replica mean is
        # taken locally -- the other replicas may be gone
        if pending is not None and "losses" in pending:
            try:
                resolve(pending, time.time(), local_only=True)
            except Exception as e:      # noqa: BLE001 -- (the device may be the reason we are here; never mask the original error)
                sys.stderr.write("dgcnn: could not write the log row of iteration %d (%s)\n" % (pending["it"], e))
        for f in (h.train_logger, h.csv_logger):
            if f:
                f.close()
    h.data_io.finalize()


def inference_loop(flags, h):
    if h.csv_logger:
        h.csv_logger.write(INFERENCE_COLUMNS + "\n")
    tsum = dict(iter=0.0, inference=0.0, io=0.0)
    has_label = bool(getattr(flags, "LABEL_KEY", ""))
    lo, _ = parallel.shard_bounds(int(flags.BATCH_SIZE), h.rank, h.world)
    while h.iteration < int(flags.ITERATION):
        it = h.iteration
        stamp = datetime.datetime.fromtimestamp(time.time()).strftime("%Y-%m-%d %H:%M:%S")
        t_iter = time.time()
        report = bool(flags.REPORT_STEP) and (it + 1) % flags.REPORT_STEP == 0

        t0 = time.time()
        idx, data, label, weight = h.data_io.next()
        if not has_label:
            label = weight = None
        t_io = time.time() - t0

        t0 = time.time()
        softmax, losses, accs = [], [], []
        for dv, lv, wv in _micro_batches(flags, h, data, label, weight):
            res = h.trainer.inference(h.sess, dv, lv, wv)
            if has_label:
                softmax += [s.clone() for s in res[:-2]]      # the engine reuses its buffers every step
                accs.append(res[-2])
                losses.append(res[-1])
            else:
                softmax += [s.clone() for s in res]
        loss, acc = _replica_mean(h, losses), _replica_mean(h, accs)
        if not has_label:
            torch.cuda.synchronize()
        t_inf = time.time() - t0

        if getattr(flags, "OUTPUT_FILE", ""):
            # every replica holds the softmax of its own shard [lo, hi) of the batch: gather the shards to rank 0,
            # which owns the output file (main_funcs.py:264-268 stores from the one process the reference has)
            rows = _gather_rows(h, softmax, idx)
            if h.rank == 0:
                for at, row in enumerate(rows):
                    h.data_io.store(idx[at], row)

        epoch = it * float(flags.BATCH_SIZE) / h.data_io.num_entries()
        t_spent = time.time() - t_iter
        for k, v in (("iter", t_spent), ("inference", t_inf), ("io", t_io)):
            tsum[k] += v
        if h.csv_logger:
            h.csv_logger.write("%d,%g,%g,%g,%g,%g,%g,%g,%g,%g\n" % (
                it, epoch, t_spent, t_inf, t_io, tsum["iter"], tsum["inference"], tsum["io"], loss, acc))
        if report and h.rank == 0:
            mem = torch.cuda.max_memory_allocated() if torch.cuda.is_available() else 0
            print("Iteration %d (epoch %g) @ %s ... inference time fraction %g%% max mem. %g ... loss %g accuracy %g"
                  % (it, round_decimals(epoch, 2), stamp, round_decimals(t_inf / t_spent * 100.0, 2), mem,
                     round_decimals(loss, 4), round_decimals(acc, 4)))
            sys.stdout.flush()
            if h.csv_logger:
                h.csv_logger.flush()
        h.iteration += 1

    if h.csv_logger:
        h.csv_logger.close()
    h.data_io.finalize()


def latest_checkpoint(prefix):
    """Newest `<prefix>-<iter>` on disk (by iteration), or '' -- convenience for resuming."""
    names = [p[:-4] for p in glob.glob(prefix + "-*.npz") if p[len(prefix) + 1:-4].isdigit()]
    return max(names, key=iteration_from_filename) if names else ""
