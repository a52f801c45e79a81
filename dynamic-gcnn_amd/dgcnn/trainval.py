"""MI355X-native counterpart of dgcnn/trainval.py:7-129 (same class name and method surface).

Reference: one process, one TF tower per entry of flags.GPUS, gradients copied to the host and
averaged there (trainval.py:16,64-73).  Here: ONE PROCESS PER GPU (rank r <-> HIP device
LOCAL_RANK); each process runs the towers it was given (normally one) on its own device, adds the
tower-mean gradient into a flat fp32 accumulator (sum over micro-steps, trainval.py:75-79), and
`apply_gradient` issues ONE RCCL all-reduce of that 7.2 MB bucket over xGMI (torch.distributed
backend "nccl" == RCCL; "gloo" in the CPU tests of the host logic), scales by 1/world and runs a
fused Adam (tf.train.AdamOptimizer defaults).  BatchNorm statistics stay local to a replica, as
in the reference.  `sess` arguments are accepted and ignored.
"""
from __future__ import annotations

import math
import os
import sys
from collections import OrderedDict

import numpy as np
import torch

from . import _engine as E
from . import _hip as H
from . import model
from . import parallel


TF_SCOPE = "dgcnn/"       # tf.variable_scope('dgcnn', reuse=tf.AUTO_REUSE): trainval.py:29
GRAPH_CACHE_MAX = 8       # captured towers kept per trainval (each pins a private pool with that shape's activations)
GRAPH_CACHE_MAX_FRACTION = 0.25   # ... and at most this share of the device's memory pinned by their pools together (LRU eviction)
GRAPH_CAPTURE_AFTER = 1   # eager sightings of a (shape, mode) before it is captured; "auto" mode uses GRAPH_CAPTURE_AFTER_AUTO
GRAPH_CAPTURE_AFTER_AUTO = 3
GRAPH_SEEN_MAX = 4096     # distinct keys remembered for the sighting count (variable-N sources produce thousands)
PLAN_AUTO_MAX_ROWS = 65536   # "auto": towers up to this many points are replayed from a launch plan (larger steps are tens of
                             # milliseconds of GPU work behind ~2 ms of host work: nothing to hide, and a plan pins the step's memory)


def _pool_bytes(pool):
    """Bytes of device memory a recorded plan's private pool holds (0 for HIP graphs, whose pool torch owns)."""
    if pool is None:
        return 0
    try:
        return int(sum(seg.get("total_size", 0) for seg in pool.snapshot()))
    except Exception:
        return 0


def param_specs(flags, num_channel):
    """[(name, shape)] in the creation order of tf.trainable_variables()."""
    from .ops import _listify
    L = int(flags.EDGE_CONV_LAYERS)
    ecf = _listify(flags.EDGE_CONV_FILTERS, L, "num_filters")
    residual = flags.MODEL_NAME in ("residual-dgcnn", "residual-dgcnn-nofc")
    specs = []
    cin = int(num_channel)
    for i in range(L):
        s = "EdgeConv%d" % i
        specs += [(s + "/conv0/weights", (2 * cin, ecf[i])), (s + "/conv0/BatchNorm/beta", (ecf[i],)),
                  (s + "/conv1/weights", (2 * ecf[i], 64)), (s + "/conv1/BatchNorm/beta", (64,))]
        if residual and i > 0 and ecf[i] != ecf[i - 1]:
            specs += [(s + "/shortcut/weights", (64, ecf[i])), (s + "/shortcut/BatchNorm/beta", (ecf[i],))]
        cin = 64
    nc = int(flags.NUM_CLASS)
    if flags.MODEL_NAME == "residual-dgcnn-nofc":
        return specs + [("Final/weights", (64, nc)), ("Final/BatchNorm/beta", (nc,))]
    specs += [("MergedEdgeConv/weights", (64 * L, 1024)), ("MergedEdgeConv/BatchNorm/beta", (1024,))]
    cin = 1024 + sum(2 * f + 64 for f in ecf) + 1024
    fcl = int(flags.FC_LAYERS)
    fcf = _listify(flags.FC_FILTERS, fcl, "num_filters")
    for i in range(fcl):
        specs += [("FC%d/weights" % i, (cin, fcf[i])), ("FC%d/BatchNorm/beta" % i, (fcf[i],))]
        cin = fcf[i]
    return specs + [("Final/weights", (cin, nc)), ("Final/BatchNorm/beta", (nc,))]


class trainval(object):

    def __init__(self, flags):
        self._flags = flags

    # ------------------------------------------------------------------ trainval.py:12-85
    def initialize(self):
        f = self._flags
        if f.MODEL_NAME not in ("dgcnn", "residual-dgcnn", "residual-dgcnn-nofc"):
            raise NotImplementedError("Unsupported MODEL_NAME: %s" % f.MODEL_NAME)
        self._ctx = E.reset()
        self._ctx.seed = int(getattr(f, "SEED", 1)) if int(getattr(f, "SEED", 1)) >= 0 else 1
        self._ctx.allocate_variables(param_specs(f, int(f.NUM_CHANNEL)), seed=self._ctx.seed)
        self._lr = float(f.LEARNING_RATE)
        self._dist, self._rank, self._world = parallel.dist_state()
        parallel.broadcast_(self._ctx.flat_param, self._dist, src=0)
        # the head's variables (MergedEdgeConv, FC*, Final) sit at the END of the flat bucket (creation order, SURVEY
        # Appendix B): [head_off, end) is the bucket that can be all-reduced while the EdgeConv backward is still running
        self._head_off = 0
        off = 0
        for name, shape in param_specs(f, int(f.NUM_CHANNEL)):
            if name.startswith("MergedEdgeConv/"):
                self._head_off = off
                break
            off += int(np.prod(shape))
        self._head_reduced = False
        self._ar_events = None        # a list: apply_gradient appends one (start, end) event pair around the exposed part of the all-reduce
        # resolved on EVERY initialize (flag, else the DGCNN_DETERMINISTIC environment default): an instance never inherits
        # the mode of an earlier one.  The switch itself is process-wide, like the GEMM arithmetic.
        det = getattr(f, "DETERMINISTIC", None)
        hp = getattr(f, "HEAD_PLANES", None)             # per instance, like DETERMINISTIC: flag, else the environment default
        planes_asked = (str(hp).lower() == "f16") if hp is not None else (E.HEAD_PLANES_ENV_DEFAULT is not None)
        # what the deterministic kernels do not take: csrc/det.hip walks at most 1024 columns per launch and reads its (R*k, F) input
        # densely; EdgeConv filter counts that are not multiples of 4 send the neighbour gradient through an atomic scatter
        # (dgcnn_edge_mlp_dgrad_scatter_f32: no float4 rows for the transposed-adjacency sum); the plane GEMMs' statistics are atomic
        specs = param_specs(f, int(f.NUM_CHANNEL))
        wide = [n for n, shp in specs if n.endswith("weights") and shp[1] > 1024]
        odd = [n for n, shp in specs if n.endswith("conv0/weights") and shp[1] % 4]
        if det is None:
            # the default is the deterministic mode wherever it exists; a model outside it runs the atomics kernels (say so once)
            det = E.DETERMINISTIC_ENV_DEFAULT
            if det and (wide or odd or planes_asked):
                det = False
                if not planes_asked:
                    sys.stderr.write("dgcnn: deterministic kernels do not take %s; using the atomics mode (reproducible to ~1e-7 "
                                     "per step)\n" % ", ".join((wide + odd)[:3]))
        E.DETERMINISTIC = bool(det)
        if E.DETERMINISTIC:            # asked for explicitly: say what is wrong here, not as an EINVAL from the middle of a step
            if wide:
                raise ValueError("DETERMINISTIC mode supports at most 1024 filters per layer (csrc/det.hip); too wide: %s"
                                 % ", ".join(wide))
            if odd:
                raise ValueError("DETERMINISTIC mode needs EdgeConv filter counts that are multiples of 4; got %s" % ", ".join(odd))
        emd = str(getattr(f, "EDGE_MLP_DTYPE", "f32") or "f32").lower()
        if emd not in ("f32", "bf16"):
            raise ValueError("EDGE_MLP_DTYPE must be f32 or bf16, got %r" % (getattr(f, "EDGE_MLP_DTYPE"),))
        E.EDGE_MLP_DTYPE = emd                           # per instance, like DETERMINISTIC
        if hp is None:
            E.HEAD_PLANES = E.HEAD_PLANES_ENV_DEFAULT
        else:
            if str(hp).lower() not in ("0", "", "none", "off", "f16"):
                raise ValueError("HEAD_PLANES must be 0 or f16, got %r" % (hp,))
            E.HEAD_PLANES = {"f16": E.PL.F16X2}.get(str(hp).lower())
        self._graphs, self._graph_seen = OrderedDict(), {}     # captured towers (LRU order) / sightings per key
        self._graph_used, self._sighting_key = {}, None         # key -> doubles of the statistics arena an eager run of it used
        ug = str(getattr(f, "USE_GRAPH", "0")).lower()
        self._use_graph = ug if ug in ("auto", "plan") else ug in ("1", "true", "yes", "on", "graph")
        self._static_inputs = bool(getattr(f, "STATIC_INPUTS", False))
        return self

    def _split_reduce(self):
        """The gradient bucket is reduced as [head piece, rest piece] on the library's RCCL communicator.  A function of the
        model and of the group's existence only -- never of a batch's shape -- so that every rank issues the same collectives."""
        return parallel.rccl_group() is not None and 0 < self._head_off < self._ctx.flat_grad.numel()

    @property
    def variables(self):
        return self._ctx.vars

    @property
    def gradients(self):
        return self._ctx.var_grads

    def feed_dict(self, data, label=None, weight=None):
        """trainval.py:87-95: one entry per tower (GPU) in each list."""
        res = {"data": list(data)}
        if label is not None:
            res["label"] = list(label)
        if weight is not None:
            res["weight"] = list(weight)
        n = len(res["data"])
        for key in ("label", "weight"):
            if key in res and len(res[key]) != n:
                raise ValueError("feed_dict: %d data towers but %d %s towers" % (n, len(res[key]), key))
        return res

    def _to_dev(self, a, dtype):
        if a is None:
            return None
        t = a if isinstance(a, torch.Tensor) else torch.as_tensor(a)
        return t.to(device=self._ctx.device, dtype=dtype, non_blocking=True).contiguous()

    def _tower(self, data, label, weight, train):
        """forward (+ backward when train) of one tower; returns (softmax (MBS,N,ncls), scal[loss, acc])."""
        pts = self._to_dev(data, torch.float32)
        lab = self._to_dev(label, torch.int32)
        wgt = self._to_dev(weight, torch.float32)
        if pts.dim() != 3:
            raise ValueError("points must be (MINIBATCH_SIZE, N, NUM_CHANNEL), got %s" % (tuple(pts.shape),))
        kind = self._wants_graph(pts.shape[0] * pts.shape[1])
        if kind is not None:
            self._sighting_key = None
            out = self._tower_graph(pts, lab, wgt, train, kind)
            if out is not None:
                return out
            out = self._tower_eager(pts, lab, wgt, train)
            if self._sighting_key is not None:          # an eager sighting of a key that will be captured: what the step uses of the
                self._graph_used[self._sighting_key] = self._ctx.stat_off     # statistics arena (the capture zeroes that much)
            return out
        return self._tower_eager(pts, lab, wgt, train)

    def _wants_graph(self, rows):
        """How a tower over `rows` points is launched: None (eager), "plan" (recorded launch plan) or "graph" (HIP graph)."""
        use = self._use_graph
        if H.TIMER is not None or not use:
            return None
        if use == "auto":                 # replay pays where the host matters (profiles/r05/launch_modes.txt)
            return "plan" if rows <= PLAN_AUTO_MAX_ROWS else None
        return "plan" if use == "plan" else "graph"

    def _tower_body(self, pts, lab, wgt, train):
        c = self._ctx
        c.begin_step()
        c.recording = bool(train)
        logits = model.build(pts, self._flags)                       # trainval.py:38
        B, N, ncls = logits.shape
        sm, scal = E.softmax_loss(logits.view(B * N, ncls), None if lab is None else lab.view(-1),
                                  None if wgt is None else wgt.view(-1), want_grad=bool(train))
        if train:
            c.backward()                                             # compute_gradients, trainval.py:54
        c.recording = False
        return sm.view(B, N, ncls), scal

    def _tower_eager(self, pts, lab, wgt, train):
        return self._tower_body(pts, lab, wgt, train)

    # ---- replay of a tower (the reference replays a static TF graph with sess.run: trainval.py:103-119) ----
    def use_graph(self, on=True):
        """How towers are launched from now on.
          False    eager: every kernel launched from Python as the tower is traversed (library default; tests hook into it);
          "plan"   launch plan (csrc/plan.cc): the second time a (shape, mode) is seen the step is RECORDED while it runs -- every
                   launch, memset, cross-stream wait and RCCL call with its arguments -- and afterwards re-issued by one C loop on
                   the same two streams: the GPU sees the eager schedule, the host spends ~2 us per launch instead of 12-30;
          True     HIP graph: the tower captured with torch.cuda.graph and replayed (one graph launch; the side stream becomes graph
                   branches, which this runtime schedules 2-5 % slower than the streams themselves, and no RCCL call inside);
          "auto"   plans for towers up to PLAN_AUTO_MAX_ROWS points that came back GRAPH_CAPTURE_AFTER_AUTO times, else eager.
        Per-step host state is never baked in: the dropout seed lives in device memory and is advanced before every replay, Adam
        and zero_gradients stay outside.  Shapes seen once (variable-N sources) run eagerly."""
        self._use_graph = on if on in ("auto", "plan") else bool(on)
        return self

    def _tower_graph(self, pts, lab, wgt, train, kind):
        c = self._ctx
        hooked = c.head_grads_hook is not None
        key = (kind, tuple(pts.shape), bool(train), lab is not None, wgt is not None, float(E.DROPOUT_KEEP), H.gemm_arith(),
               E.WGRAD_SIDE_STREAM, E.HEAD_PLANES, E.DETERMINISTIC, E.EDGE_MLP_DTYPE, hooked,
               int(torch.cuda.current_stream().cuda_stream))     # (a plan bakes in the streams that were current at record time)
        ent = self._graphs.get(key)
        if ent is not None:
            # DETERMINISTIC mode grows the slot count (and with it the statistics arena) when a larger cloud arrives: a graph
            # captured before that replays launches that zero and write the OLD arena -- memory the allocator may have handed
            # to someone else by now.  Such a graph is dropped and the shape captured again.
            c.ensure_arena(int(pts.shape[0]) * int(pts.shape[1]))
            if ent["arena"] != c.arena_key():
                self._drop_graph(key)
                ent = None
        if ent is None:
            # first sightings run eagerly (the very first also allocates workspaces / arenas).  A variable-N source
            # (-np -1 -mbs 1) shows thousands of distinct point counts: under "auto" a shape must come back a few times
            # before it is worth a capture (a capture costs a device synchronisation and pins that shape's activations)
            need = GRAPH_CAPTURE_AFTER_AUTO if self._use_graph == "auto" else GRAPH_CAPTURE_AFTER
            n = self._graph_seen.get(key, 0)
            if n < need:
                if len(self._graph_seen) >= GRAPH_SEEN_MAX and key not in self._graph_seen:
                    self._graph_seen.clear()
                    self._graph_used.clear()
                self._graph_seen[key] = n + 1
                self._sighting_key = key
                return None
            ent = self._capture(key, pts, lab, wgt, train, kind)
            if ent is None:
                return None
            if kind == "plan":                          # recording a plan RUNS the step: this call is done
                return ent["sm"].clone(), ent["scal"].clone()
        else:
            self._graphs.move_to_end(key)
        # the recorded step reads its OWN input buffers: stage the caller's tensors into them -- unless this very tensor OBJECT (kept
        # referenced here, so its address cannot be handed to another tensor), unmodified since (torch's version counter, shared by
        # all views of it), is what they already hold.  (An eager step does not copy its device-resident inputs either.)
        # That skip is OPT-IN (flag STATIC_INPUTS, set by bench.py for its resident batch): a write that bypasses torch (another
        # library through data_ptr / DLPack) does not move the version counter.
        staged = ent.setdefault("staged", {})
        for name, dst, src in (("pts", ent["pts"], pts), ("lab", ent["lab"], lab), ("wgt", ent["wgt"], wgt)):
            if dst is None or dst.data_ptr() == src.data_ptr():
                continue
            ver = None
            if self._static_inputs:
                try:
                    ver = src._version
                except RuntimeError:                     # (inference-mode tensors have no version counter)
                    ver = None
            prev = staged.get(name)
            if ver is None or prev is None or prev[0] is not src or prev[1] != ver:
                dst.copy_(src, non_blocking=True)
                staged[name] = (src, ver)
        c.advance_seed()
        if kind == "plan":
            c.head_grads_hook = None                     # (the recorded step contains what the hook issued)
            ent["plan"].replay()
            self._head_reduced = ent["head_reduced"]
        else:
            ent["graph"].replay()
        c.stat_off = max(c.stat_off, ent["stat_used"])       # (the next eager step re-zeroes what this replay dirtied)
        # the replay's output buffers are overwritten by the next replay of this key: hand out copies (stream ordered),
        # so that several towers / micro-steps of one shape each keep their own softmax and [loss, accuracy]
        return ent["sm"].clone(), ent["scal"].clone()

    def _capture(self, key, pts, lab, wgt, train, kind):
        c = self._ctx
        ent = {"pts": pts.clone(), "lab": None if lab is None else lab.clone(), "wgt": None if wgt is None else wgt.clone()}
        hook = c.head_grads_hook
        try:
            c.ensure_arena(int(pts.shape[0]) * int(pts.shape[1]))
            ent["arena"] = c.arena_key()
            # The recorded step zeroes only what IT uses of the statistics arena; whatever the steps before it dirtied beyond that
            # (a larger cloud, a train step before an eval capture, a replay) is zeroed HERE, outside the recording, so that "all
            # of the arena behind stat_off is zero" still holds for the eager steps that follow.
            if c.stat_arena is not None and c.stat_off > 0:
                H.memset(c.stat_arena[:min(c.stat_off, c.stat_arena.numel())])
                c.stat_off = 0
            torch.cuda.synchronize()
            c.capturing = True
            c.capture_extent = self._graph_used.get(key)
            if kind == "plan":
                # every allocation of the recorded step comes from a private pool that lives as long as the plan: the addresses in
                # the recorded launches stay this step's own (what torch.cuda.graph does for a captured graph)
                pool = torch.cuda.MemPool()
                c.advance_seed()
                with torch.cuda.use_mem_pool(pool):
                    with H.Plan.record() as plan:
                        ent["sm"], ent["scal"] = self._tower_body(ent["pts"], ent["lab"], ent["wgt"], train)
                ent["plan"], ent["pool"], ent["head_reduced"] = plan, pool, self._head_reduced
                ent["info"] = plan.info()
            else:
                c.head_grads_hook = None                 # (a captured HIP graph cannot carry the RCCL call)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    ent["sm"], ent["scal"] = self._tower_body(ent["pts"], ent["lab"], ent["wgt"], train)
                ent["graph"] = g
            if c.capture_extent is not None and c.stat_off > c.capture_extent:
                # the recorded memset covers less than the step writes (the slot count or the arena changed between the sighting and
                # the recording).  THIS execution was right -- everything behind the recorded extent was zeroed just above -- but a
                # replay would not be: keep the results, drop the recording, and let the next one zero the full extent.
                self._graph_used.pop(key, None)
                ent["discard"] = True
        except Exception as e:                        # capture is an optimisation: fall back to eager launches, loudly
            sys.stderr.write("dgcnn: %s capture failed (%s: %s); running eagerly\n"
                             % ("launch-plan" if kind == "plan" else "HIP graph", type(e).__name__, e))
            self._use_graph = False
            ent = None
            if kind == "plan":
                # the step may have died half way: its gradient contributions are partial.  Nothing was returned to the caller
                # yet, who falls back to the eager tower -- on top of whatever this one added.  Refuse silently wrong sums.
                raise
        finally:
            c.capturing = False
            c.capture_extent = None
            c.recording = False
            c.tape = []
            c.roots = []
            c.side_busy = False
            c.side_hold = []
            c.head_grads_hook = hook if ent is None else None
        if ent is not None:
            ent["stat_used"] = c.stat_off                        # doubles of the statistics arena the captured step writes
            ent["pool_bytes"] = _pool_bytes(ent.get("pool"))
            self._graphs[key] = ent
            if ent.pop("discard", False):                        # results stand, the recording does not (see above)
                self._graph_seen[key] = 0
                out = {"sm": ent["sm"].clone(), "scal": ent["scal"].clone()}
                self._drop_graph(key)
                return out if kind == "plan" else None
            if ent["arena"] != c.arena_key():                    # (cannot happen after ensure_arena; never replay such a graph)
                self._drop_graph(key)
                return None
            # least recently replayed first; frees its private pool.  Bounded by count AND by the bytes the pools pin together
            # (a variable-N source that repeats a handful of large shapes pins GBs per key that eager steps cannot reclaim).
            cap = GRAPH_CACHE_MAX_FRACTION * torch.cuda.get_device_properties(c.device).total_memory
            while len(self._graphs) > 1 and (len(self._graphs) > GRAPH_CACHE_MAX
                                             or sum(e.get("pool_bytes", 0) for e in self._graphs.values()) > cap):
                victim = next(iter(self._graphs))
                if victim == key:
                    break
                self._drop_graph(victim)
        return ent

    def _drop_graph(self, key):
        old = self._graphs.pop(key, None)
        if old is not None:
            torch.cuda.current_stream().synchronize()            # (a replay of it may still be in flight)
            if self._ctx.side is not None:
                self._ctx.side.synchronize()
            if "graph" in old:
                old["graph"].reset()
            if "plan" in old:
                old["plan"].destroy()
            old.clear()

    def launch_plan_info(self):
        """[{kernels, memsets, waits, collectives}] of the recorded plans (bench.py reports the enqueues per step from it)."""
        return [dict(e["info"]) for e in self._graphs.values() if "info" in e]

    def make_summary(self, sess, data, label, weight):
        if not self._flags.TRAIN:
            raise NotImplementedError
        res = self.inference(sess, data, label, weight)
        return {"accuracy": float(res[-2]), "loss": float(res[-1])}

    def inference(self, sess, data, label=None, weight=None):
        """trainval.py:103-108: [softmax_tower0, ..., (accuracy, loss)]."""
        fd = self.feed_dict(data, label, weight)
        outs, scals = [], []
        for i in range(len(fd["data"])):
            sm, scal = self._tower(fd["data"][i], fd["label"][i] if label is not None else None,
                                   fd["weight"][i] if weight is not None else None, train=False)
            outs.append(sm)
            scals.append(scal)
        if label is not None:
            s = torch.stack(scals).mean(0)                           # trainval.py:59-60
            outs += [s[1], s[0]]
        return outs

    def accum_gradient(self, sess, data, label, weight=None, summary=False, last=False):
        """trainval.py:110-119: [accum_results, accuracy, loss(, summary)]; tower-mean gradient is
        ADDED to the accumulators (sum over micro-steps, trainval.py:79).
        last=True (not in the reference: its averaging happens inside apply_gradient's session run): this is the final
        micro-step before apply_gradient, so the head's share of the gradient bucket may be all-reduced as soon as the head's
        backward has produced it, underneath the EdgeConv backward (RCCL group registered, one tower per process)."""
        if not self._flags.TRAIN:
            raise NotImplementedError
        c = self._ctx
        fd = self.feed_dict(data, label, weight)
        T = len(fd["data"])
        c.head_grads_hook = None
        d0 = fd["data"][0]
        # Whether the head bucket starts early is a LOCAL matter (this rank's shape decides between eager launches and a
        # replayed graph, which cannot carry the RCCL call); the SEQUENCE of collectives is not: with an RCCL group the bucket
        # always travels as [head piece, rest piece] (apply_gradient sends whatever the hook did not), so ranks holding clouds
        # of different sizes (-np -1 -mbs 1) issue identical calls.
        # (a replayed HIP graph cannot carry the RCCL call; eager launches and a launch plan can)
        eager = self._wants_graph(int(d0.shape[0]) * int(d0.shape[1])) != "graph"
        if last and T == 1 and eager and self._split_reduce():
            def hook():
                c.join_side()                                   # the head's weight-gradient GEMMs (side stream) have landed
                self._head_reduced = parallel.allreduce_sum_async(c.flat_grad[self._head_off:])
            c.head_grads_hook = hook
        saved = None
        if T > 1:
            saved = c.flat_grad.clone()
            H.memset(c.flat_grad)
        scals = []
        for i in range(T):
            _, scal = self._tower(fd["data"][i], fd["label"][i], fd["weight"][i] if weight is not None else None,
                                  train=True)
            scals.append(scal)
        if T > 1:                                                     # mean over towers, trainval.py:64-73
            H.call("dgcnn_axpby_f32", saved.data_ptr(), 1.0, c.flat_grad.data_ptr(), 1.0 / T, c.flat_grad.numel())
        s = scals[0] if T == 1 else torch.stack(scals).mean(0)
        res = [c.flat_grad, s[1], s[0]]
        if summary:
            res.append({"accuracy": s[1], "loss": s[0]})
        return res

    def zero_gradients(self, sess):
        if not self._flags.TRAIN:
            raise NotImplementedError
        H.memset(self._ctx.flat_grad)                                 # trainval.py:76
        return [self._ctx.flat_grad]

    def apply_gradient(self, sess):
        """trainval.py:80,126-129 + the cross-replica mean of :64-73 (one flat RCCL all-reduce)."""
        if not self._flags.TRAIN:
            raise NotImplementedError
        c = self._ctx
        g = c.flat_grad
        ev = self._ar_events
        if ev is not None:            # measurement only (bench.py): how long the step's stream waits for the collective
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        if self._split_reduce():
            # two pieces, ALWAYS in this order on every rank: head (97 % of the bytes; already travelling when the backward's
            # hook could start it under the EdgeConv backward), then the rest
            if not self._head_reduced:
                parallel.allreduce_sum_async(g[self._head_off:])
            self._head_reduced = False
            parallel.allreduce_sum_async(g[:self._head_off])
            parallel.rccl_group().wait()
            parallel.scale_(g, self._world)
        else:
            parallel.allreduce_mean_(g, self._dist, self._world)      # sum over replicas / world
        if ev is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            ev.append((e0, e1))
        c.adam_t += 1
        b1, b2, eps = 0.9, 0.999, 1e-8                                # tf.train.AdamOptimizer defaults
        lr_t = self._lr * math.sqrt(1.0 - b2 ** c.adam_t) / (1.0 - b1 ** c.adam_t)
        H.call("dgcnn_adam_f32", c.flat_param.data_ptr(), g.data_ptr(), c.flat_m.data_ptr(), c.flat_v.data_ptr(),
               g.numel(), lr_t, b1, b2, eps)
        return None

    # ------------------------------------------------------------------ main_funcs.py:83-94,181-185
    def _slots(self):
        """[(variable name, offset into the flat buckets, view)] in bucket order.  The buckets are laid out by
        allocate_variables(param_specs(...)); a variable created lazily by get_variable outside the specs has no
        slice in them and is not checkpointed."""
        c = self._ctx
        base = c.flat_param.data_ptr()
        out = []
        for name, var in c.vars.items():
            off = (var.data_ptr() - base) // 4
            if 0 <= off and off + var.numel() <= c.flat_param.numel() and var.data_ptr() >= base:
                out.append((name, int(off), var))
        return out

    def state_dict(self):
        """Host copy of everything tf.train.Saver would write for this graph, under the reference's names: every
        variable lives in the outer scope `dgcnn/` (trainval.py:29), slim stores a 1x1 convolution's weights as
        [1, 1, Cin, Cout], Adam keeps two slots per trainable variable (`<name>/Adam`,
        `<name>/Adam_1`) and two beta power accumulators, and every BatchNorm owns `moving_mean` / `moving_variance`,
        which the reference never updates (their update ops sit in UPDATE_OPS and are never run, Appendix A.3): they
        are written with their initial values (0 / 1) so that a reader keyed on the TF names finds them."""
        c = self._ctx
        out = {}
        m, v = c.flat_m.cpu().numpy(), c.flat_v.cpu().numpy()
        as_tf = lambda a: a.reshape((1, 1) + a.shape) if a.ndim == 2 else a
        for name, off, var in self._slots():
            n = var.numel()
            key = TF_SCOPE + name
            out[key] = as_tf(var.detach().cpu().numpy().copy())
            out[key + "/Adam"] = as_tf(m[off:off + n].reshape(tuple(var.shape)).copy())
            out[key + "/Adam_1"] = as_tf(v[off:off + n].reshape(tuple(var.shape)).copy())
            if name.endswith("BatchNorm/beta"):
                stem = key[:-len("beta")]
                out[stem + "moving_mean"] = np.zeros(tuple(var.shape), np.float32)
                out[stem + "moving_variance"] = np.ones(tuple(var.shape), np.float32)
        out["beta1_power"] = np.float32(0.9 ** (c.adam_t + 1))      # TF: beta^(t+1) after t updates
        out["beta2_power"] = np.float32(0.999 ** (c.adam_t + 1))
        out["adam_step"] = np.int64(c.adam_t)
        out["dropout_step"] = np.int64(c.step_seed)           # position in the dropout mask stream
        return out

    def load_state_dict(self, state, strict=True):
        """Accepts the names / shapes state_dict writes and the round-1 form (no `dgcnn/` prefix, 2-D weights)."""
        c = self._ctx
        slots = self._slots()
        find = lambda n: TF_SCOPE + n if TF_SCOPE + n in state else (n if n in state else None)
        missing = [n for n, _, _ in slots if find(n) is None]
        if missing and strict:
            raise KeyError("checkpoint lacks variables: %s" % ", ".join(missing[:4]))
        m = np.zeros(c.flat_param.numel(), np.float32)
        v = np.zeros_like(m)
        p = c.flat_param.cpu().numpy().copy()
        for name, off, var in slots:
            n = var.numel()
            key = find(name)
            if key is None:
                continue
            a = np.asarray(state[key], np.float32)
            if a.ndim == 4 and a.shape[:2] == (1, 1):
                a = a[0, 0]
            if a.shape != tuple(var.shape):
                raise ValueError("checkpoint variable %s has shape %s, graph wants %s" % (key, a.shape, tuple(var.shape)))
            p[off:off + n] = a.reshape(-1)
            if key + "/Adam" in state:
                m[off:off + n] = np.asarray(state[key + "/Adam"], np.float32).reshape(-1)
                v[off:off + n] = np.asarray(state[key + "/Adam_1"], np.float32).reshape(-1)
        c.flat_param.copy_(torch.from_numpy(p))
        c.flat_m.copy_(torch.from_numpy(m))
        c.flat_v.copy_(torch.from_numpy(v))
        c.adam_t = int(state["adam_step"]) if "adam_step" in state else 0
        c.step_seed = int(state["dropout_step"]) if "dropout_step" in state else 0
        return self

    def save(self, prefix, global_step):
        """`weight_io.save(sess, WEIGHT_PREFIX, global_step=iteration)` (main_funcs.py:183): writes
        `<prefix>-<global_step>.npz` (rank 0 only; replicas hold identical state) and returns the
        checkpoint name WITHOUT extension, the form MODEL_PATH / iteration_from_filename expect."""
        name = "%s-%d" % (prefix, int(global_step))
        if self._rank == 0:
            d = os.path.dirname(name)
            if d and not os.path.isdir(d):
                os.makedirs(d)
            tmp = name + ".tmp.npz"
            np.savez(tmp, **self.state_dict())
            os.replace(tmp, name + ".npz")
        return name

    def restore(self, path):
        """`weight_io.restore(sess, MODEL_PATH)` (main_funcs.py:91): every rank reads the file."""
        fn = path if path.endswith(".npz") else path + ".npz"
        with np.load(fn) as z:
            self.load_state_dict({k: z[k] for k in z.files})
        return self
