"""Host-side engine: variable store, backward tape and the composite EdgeConv / conv+BN blocks.

The reference builds a TF1 graph once and replays it with sess.run (dgcnn/trainval.py:12-85).
Here every call runs the HIP kernels immediately on the current stream and records one closure
per block on a tape; `backward()` replays the tape in reverse.  There is no autograd and no
torch math on the path: torch tensors only hold memory.

Layout conventions (DESIGN.md "Data layout"):
  * every activation is a 2-D row-major view (rows = points or edges, cols = channels) with an
    explicit leading dimension, so the outputs of one block can be written straight into column
    slices of a wider buffer (the tf.concat of dgcnn/ops.py:58 and model.py:63,85 never copies);
  * gradients mirror the forward buffers: the gradient of a column slice is the same slice of the
    gradient of its parent buffer (found by address), zero-initialised, always accumulated into;
  * parameters, their gradient accumulators and the Adam moments live in four flat fp32 buckets
    (one RCCL all-reduce, one fused Adam launch per step);
  * conv0 of an EdgeConv layer is a point-level GEMM [U|V] = X [Wa-Wb | Wb]; its (B*N*k, F) output is never
    written: the BatchNorm passes recompute V[neighbour] + U[point] (edge_conv_block);
  * work that only the optimizer needs (weight-gradient GEMMs, the transposed adjacency) runs on a second
    HIP stream (Context.off_critical_path) and joins before the tape is released.
"""
from __future__ import annotations

import contextlib
import os
from collections import OrderedDict

import numpy as np
import torch

from . import _hip as H
from . import _planes as PL

BN_EPS = 1e-3          # slim.batch_norm default epsilon [TF1-lib]
DROPOUT_KEEP = 0.7     # tf.nn.dropout(net, 0.7): dgcnn/model.py:91
WS_BYTES = 256 << 20
EDGE_MLP_LITERAL = False  # True: conv0 as the literal (B*N*k) x 2C GEMM over E = [x_i, x_j - x_i] (A/B switch)
EDGE_MATERIALIZE_Y = False  # True: gather-add writes the (B*N*k, F) conv0 output and BatchNorm streams it (A/B switch);
                            # False: the BatchNorm passes recompute y = V[neighbour] + U[point] (no edge tensor in the forward)
WGRAD_SIDE_STREAM = os.environ.get("DGCNN_SIDE_STREAM", "1") != "0"   # weight-gradient GEMMs on a second HIP stream
# Fusions with a separate-pass twin that the tests compare them against (module constants, not run-time switches):
EDGE_BWD_FUSED_L0 = True   # input layer (C <= 4, no input gradient): conv0's whole backward in one pass
FUSE_DROPOUT = True        # tf.nn.dropout inside the last FC layer's BatchNorm passes
BF16_FUSED_BWD = True      # bf16 edge-MLP: the layer's backward in one pass over the edges (tests switch it off to compare)
EDGE_BWD_REDUCE_POINTS = True   # BN backward sums of conv0 from per-point data (False: a pass over the edges)
SIDE_STREAM_MIN_ROWS = 16384   # below this many points the side stream is not used
# the side stream's work is off the critical path: least priority / a CU mask that leaves some compute units to the main stream alone
# (csrc/plan.cc:dgcnn_stream_create; A/B switches -- neither helps, the mask costs 40 %: profiles/r06/side_stream.txt)
SIDE_STREAM_LOW_PRIORITY = os.environ.get("DGCNN_SIDE_LOW_PRIORITY", "0") not in ("0", "")
SIDE_STREAM_RESERVE_CUS = int(os.environ.get("DGCNN_SIDE_RESERVE_CUS", "0"))
# conv0 of every EdgeConv layer with bf16 OPERANDS (BASELINE.json configs[2] "bf16 edge-MLP MFMA"): E = [x_i, x_j - x_i] is formed
# in fp32, E and W0 are rounded to bf16 once, the literal (B*N*k) x 2C x F product runs on v_mfma_f32_32x32x16_bf16 with fp32
# accumulation; the two gradient products round their operands the same way.  "f32" (default): the fp32-class folded form.
# trainval.initialize() sets it per instance from flags.EDGE_MLP_DTYPE.
EDGE_MLP_DTYPE = "f32"
EDGE_MLP_NBR_GEMM = False  # True: factored conv0 with an edge-level neighbour GEMM instead of point-level GEMM + gather-add
SCATTER_ATOMICS = False  # True: dx_j += dY W^T by fp32 atomics in the GEMM epilogue (A/B switch)
# Deterministic mode (the DEFAULT since round 5): run-to-run bit-reproducible results.  Every statistics / reduction slot has ONE
# writer (dgcnn_set_stat_slots raises the slot count to the widest producer's workgroup count), the finalize kernels add the slots
# in a fixed order, the transposed adjacency's buckets are sorted.  Same kernels as the atomics mode otherwise: +3-4 % at
# configs[1].  The atomics mode (DETERMINISTIC=False / DGCNN_DETERMINISTIC=0: 32 slots, fp64 atomics from several
# workgroups per slot, unsorted buckets) is reproducible to ~1e-7 per step only, which the dynamic graphs amplify into a few
# different neighbour lists per step.  The reported loss / accuracy scalars are summed with atomics in both (they feed nothing back).
DETERMINISTIC_ENV_DEFAULT = os.environ.get("DGCNN_DETERMINISTIC", "1") not in ("0", "")
DETERMINISTIC = DETERMINISTIC_ENV_DEFAULT      # trainval.initialize() sets it per instance (flag, else this default)


# Head GEMMs (MergedEdgeConv, FC0, FC1: 98 % of the step's GEMM flops) from pre-split operand planes (csrc/gemm_pl.hip):
#   "f16"  two fp16 planes per operand, 3 partial products   "0"  off
HEAD_PLANES_ENV_DEFAULT = {"f16": PL.F16X2}.get(os.environ.get("DGCNN_HEAD_PLANES", "0").lower())
HEAD_PLANES = HEAD_PLANES_ENV_DEFAULT          # trainval.initialize() sets it per instance (flag HEAD_PLANES, else this default)
PLANES_MIN_ROWS = 8192


def planes_ok(R, Cin, Cout):
    """The plane GEMM takes this layer: head-sized problems with tile-friendly widths, variables in the flat bucket (the
    step's scales come from it), default (atomically summed) BatchNorm statistics."""
    return (HEAD_PLANES is not None and not DETERMINISTIC and ctx().flat_param is not None and R >= PLANES_MIN_ROWS and
            Cin % 32 == 0 and Cout % 32 == 0 and Cout >= 128 and Cin >= 128 and Cout <= 1024)


class Context(object):
    def __init__(self):
        self.vars = OrderedDict()        # name -> 2-D / 1-D tensor (views of flat_param when allocated)
        self.var_grads = {}              # name -> tensor (views of flat_grad)
        self.flat_param = self.flat_grad = self.flat_m = self.flat_v = None
        self.scope = []
        self.tape = []
        self.roots = []                  # [(buffer, [grad or None])]
        self.recording = False
        self.ws = None
        self.ws_side = None
        self.side = None                 # second HIP stream: weight-gradient GEMMs (nothing on the critical path needs them)
        self.side_busy = False
        self.side_hold = []              # recording a launch plan: tensors the side stream touches, kept until the join
        self._slot_stack = []
        self.step_slots = 32             # slot count the step's arena is sized for (configure_slots); ops may use fewer (op_slots)
        self.capture_extent = None       # recording: doubles of the arena the recorded step zeroes (what an eager run of it used)
        self.seed = 1
        self._rng = None
        self.stat_arena = None
        self.stat_off = 0
        self.arena_gen = 0               # bumped whenever the statistics arena is (re)allocated
        self.step_seed = 0
        self.seed_dev = None             # device uint64: the dropout seed of the running step (read by the kernels)
        self.capturing = False           # inside a HIP-graph capture: no host-side per-step state may be baked in
        self.debug = False
        self.planes = {}                 # (data_ptr, rows, cols) of an fp32 2-D view -> PlaneSet holding it as GEMM operand planes (this step)
        self.pl_scales = None            # device float[2]: power-of-two scales of the step's activation / weight plane sets (fp16 planes)
        self.pl_scales_ready = False
        self.pl_ws = None
        self.wprep = {}                  # weight-only work of the step, issued ahead on the side stream: key -> tensor / PlaneSet
        self.wprep_event = None          # True while the main stream has not yet been made to wait for it (prepared())
        self.head_grads_hook = None      # called by the backward when every head gradient is final (trainval: bucketed all-reduce)

    # ---- device / scratch -------------------------------------------------------------
    @property
    def device(self):
        if not torch.cuda.is_available():
            raise H.HipError("no GPU visible: the dgcnn HIP path has no CPU fallback")
        return torch.device("cuda", torch.cuda.current_device())

    def workspace(self):
        if torch.cuda.current_stream() == self.side and self.side is not None:
            if self.ws_side is None or self.ws_side.device != self.device:
                self.ws_side = torch.empty(WS_BYTES, dtype=torch.uint8, device=self.device)
            return self.ws_side
        if self.ws is None or self.ws.device != self.device:
            self.ws = torch.empty(WS_BYTES, dtype=torch.uint8, device=self.device)
        return self.ws

    @contextlib.contextmanager
    def off_critical_path(self, *temporaries, rows=1 << 30):
        """Run the enclosed launches on the side stream, ordered after everything issued so far on the
        current stream.  The bf16-split GEMMs run at the package power cap while the BatchNorm / gather passes
        of the backward chain are bandwidth bound: a weight-gradient GEMM (needed only by the optimizer)
        issued here overlaps the chain instead of stalling it.  `temporaries`: tensors allocated on the main
        stream that the side work touches and that may be freed before the streams join."""
        if not WGRAD_SIDE_STREAM or rows < SIDE_STREAM_MIN_ROWS:      # tiny problems are host-launch bound: the
            yield                                                       # stream switches only add latency there
            return
        main = torch.cuda.current_stream()
        if self.side is None or self.side.device != self.device:
            self.side = H.make_stream(self.device, low_priority=SIDE_STREAM_LOW_PRIORITY, reserve_cus=SIDE_STREAM_RESERVE_CUS)
        H.stream_wait(self.side, main)
        for t in temporaries:
            t.record_stream(self.side)
        if self.capturing:
            # a recorded step is replayed with the addresses of THIS run: a block the allocator handed out again because it SAW the
            # side stream finish (an event query, not stream order) would be a race on replay -- side-stream temporaries stay
            # alive until the streams have joined
            self.side_hold.extend(temporaries)
        self.side_busy = True
        with torch.cuda.stream(self.side):
            yield

    def join_side(self):
        if self.side_busy:
            H.stream_wait(torch.cuda.current_stream(), self.side)
            self.side_busy = False
        self.side_hold = []

    def stats(self, F):
        """Zeroed double[SLOTS][2][F] carved from one arena that is memset once per step."""
        n = H.STAT_SLOTS * 2 * F
        want = self.arena_want()
        if self.stat_arena is None or self.stat_arena.device != self.device or self.stat_arena.numel() < want:
            # a captured HIP graph has the arena's address (and the slot count) baked into its launches: trainval keys its
            # graphs on arena_key() and drops those recorded against an arena that no longer exists
            self.stat_arena = H.zeros(want, torch.float64, self.device)
            self.stat_off = 0
            self.arena_gen += 1
        if self.stat_off + n > want:
            return H.zeros(n, torch.float64, self.device)
        s = self.stat_arena[self.stat_off:self.stat_off + n]
        self.stat_off += n
        return s

    def arena_want(self):
        """Doubles of the statistics arena a step may use at its slot count: 8 MB at 32 slots, 96 MB at 768.  The arena itself may
        be larger (an earlier, larger cloud grew it): a step carves from -- and a captured step zeroes -- at most this much."""
        n = max(self.step_slots, H.STAT_SLOTS)
        return (1 << 20) if n <= 32 else (1 << 19) * -(-n // 32)

    def arena_key(self):
        """What a captured step depends on besides its inputs: the statistics arena it zeroes / writes and the slot count its
        launches carry as arguments."""
        return (self.arena_gen, H.STAT_SLOTS, 0 if self.stat_arena is None else self.stat_arena.data_ptr())

    def ensure_arena(self, rows):
        """Slot count and arena for a step over `rows` points, settled OUTSIDE a capture (so that the capture neither allocates
        the arena inside its private pool nor changes the slot count half way)."""
        self.configure_slots(rows)
        off = self.stat_off
        self.stats(1)
        self.stat_off = off

    def configure_slots(self, rows=0):
        """DETERMINISTIC: every stats / red slot gets a single writer -- as many slots as the producer with the most workgroups has
        (a GEMM over `rows` rows in 64-row tiles; the reduction kernels cap their grids at the slot count) -- so that no sum
        depends on the order in which workgroups finish.  A pure function of `rows`: the reduction kernels' grids follow the slot
        count, so a count that remembered earlier, larger steps would make a step's rounding depend on the process' history
        (round 5: the same step in a fresh process and after a 65536-point cloud differed by 4e-4 -- both "deterministic").
        Atomics mode: DGCNN_STAT_SLOTS (32) slots, several writers each."""
        if DETERMINISTIC:
            n = max(256, -(-int(rows) // 64))                              # (the finalize kernels walk all of them: no rounding up)
            n = -(-n // 64) * 64
        else:
            n = 32
        self.step_slots = n
        if n != H.STAT_SLOTS:
            H.set_stat_slots(n)

    def push_slots(self, writers=0):
        """op_slots() as a pair of calls (the producer and its finalize sit far apart in the block functions)."""
        self._slot_stack.append(H.STAT_SLOTS)
        if DETERMINISTIC:
            n = max(256, -(-int(writers) // 64) * 64)
            if n != H.STAT_SLOTS:
                H.set_stat_slots(n)

    def pop_slots(self):
        prev = self._slot_stack.pop()
        if H.STAT_SLOTS != prev:
            H.set_stat_slots(prev)

    @contextlib.contextmanager
    def op_slots(self, writers=0):
        """DETERMINISTIC: the slot count of ONE op's statistics buffers -- its producer's writer workgroups (a GEMM's row tiles:
        dgcnn_gemm_stat_writers; 0 for the reduction kernels, whose grids follow the slot count), at least 256 -- instead of the
        step-wide maximum: the head's GEMMs run 192 row tiles where conv1's run 768, and every finalize kernel walks (and every
        step zeroes) all slots of its buffer.  Producer and finalize are issued inside the block: they see the same count."""
        if not DETERMINISTIC:
            yield
            return
        n = max(256, -(-int(writers) // 64) * 64)
        prev = H.STAT_SLOTS
        if n != prev:
            H.set_stat_slots(n)
        try:
            yield
        finally:
            if H.STAT_SLOTS != prev:
                H.set_stat_slots(prev)

    # ---- operand planes of the head GEMMs (csrc/gemm_pl.hip, planes_bn.hip) ------------------------------
    @staticmethod
    def plane_key(v):
        return (v.data_ptr(), int(v.shape[0]), int(v.shape[1]))

    def ensure_plane_scales(self, rows_max):
        """One pass over the parameter bucket per step: the power-of-two scales of every activation plane set
        (|z| <= 2 (sqrt(rows_max) + max |parameter|) bounds any batch-normalised tensor of the model and the residual sums
        of two of them) and of the weight plane sets.  Only the fp16 plane format is scaled."""
        if HEAD_PLANES != PL.F16X2 or self.pl_scales_ready:
            return
        if self.pl_scales is None or self.pl_scales.device != self.device:
            self.pl_scales = torch.ones(2, dtype=torch.float32, device=self.device)
            self.pl_ws = torch.zeros(1, dtype=torch.int32, device=self.device)
        H.call("dgcnn_param_scales_f32", self.flat_param.data_ptr(), self.flat_param.numel(), float(rows_max), 2.0,
               self.pl_scales.data_ptr(), self.pl_ws.data_ptr())
        self.pl_scales_ready = True

    def new_planes(self, rows, cols, kind):
        """An empty plane set in the head format; kind 'act' / 'w' picks the step's preset scale, 'own' leaves it to the producer."""
        ps = PL.PlaneSet(rows, cols, HEAD_PLANES, device=self.device)
        if HEAD_PLANES == PL.F16X2:
            if kind == "own":
                ps.scale = torch.empty(1, dtype=torch.float32, device=self.device)
            else:
                ps.scale = self.pl_scales[0:1] if kind == "act" else self.pl_scales[1:2]
        return ps

    def planes_of(self, x):
        """The operand planes of the fp32 view x: registered by its producer, or split here (one extra pass)."""
        ps = self.planes.get(self.plane_key(x))
        if ps is None:
            ps = self.new_planes(x.shape[0], x.shape[1], "act").fill_from(x)
            self.planes[self.plane_key(x)] = ps
        return ps

    def stats_raw(self, n):
        """n zeroed doubles from the per-step arena."""
        if self.stat_arena is None or self.stat_arena.device != self.device:
            self.stats(1)
            self.stat_off = 0
        if self.stat_off + n > min(self.stat_arena.numel(), self.arena_want()):
            return H.zeros(n, torch.float64, self.device)
        s = self.stat_arena[self.stat_off:self.stat_off + n]
        self.stat_off += n
        return s

    def zeros_f32(self, n):
        """n zeroed floats: small requests come out of the statistics arena (zeroed once per step by ONE memset; every separate
        memset is a ~6-us serial step of its own on the main stream), large ones get their own fill."""
        if n <= 65536 and self.stat_arena is not None:
            nd = ((n + 3) // 4) * 2                       # whole 16-byte units, from a 16-byte aligned start
            off = (self.stat_off + 1) & ~1
            if off + nd <= min(self.stat_arena.numel(), self.arena_want()):
                s = self.stat_arena[off:off + nd].view(torch.float32)[:n]
                self.stat_off = off + nd
                return s
        return H.zeros(n, torch.float32, self.device)

    def zeros_like_small(self, t):
        if t.dtype == torch.float32 and t.is_contiguous():
            return self.zeros_f32(t.numel()).view(t.shape)
        return H.memset(torch.empty_like(t))

    def begin_step(self):
        """Drop the previous tape / gradient roots, re-zero the statistics arena and advance the dropout stream."""
        self.tape = []
        self.roots = []
        self.planes = {}
        self.pl_scales_ready = False
        self.wprep = {}
        self.wprep_event = None
        self._slot_stack = []
        if not DETERMINISTIC and H.STAT_SLOTS != 32:
            H.set_stat_slots(32)
        elif DETERMINISTIC and H.STAT_SLOTS < 256:
            self.configure_slots(0)           # (stand-alone ops; a model step sets its own count in model.build)
        if self.stat_arena is not None:
            if self.capturing:
                # a captured step cannot know what ran before it: everything the step uses -- as measured on an eager run of the
                # same shape (trainval passes it) -- else everything a step at this slot count may use
                ext = min(self.stat_arena.numel(), self.arena_want())
                if self.capture_extent is not None:
                    ext = min(ext, max(int(self.capture_extent), 1))
                H.memset(self.stat_arena[:ext])
            elif self.stat_off > 0:
                H.memset(self.stat_arena[:self.stat_off])
        self.stat_off = 0
        if not self.capturing:                   # (a replayed graph gets its seed from advance_seed(), called by the replayer)
            self.advance_seed()

    def advance_seed(self):
        """Next position of the dropout mask stream.  The seed lives in DEVICE memory (dgcnn_dropout_dev_f32 reads it at
        execution time), written here by a fill kernel that carries the value as a launch argument -- stream ordered, so
        several towers / micro-steps in flight each see their own value, and a captured graph sees a fresh one per replay."""
        self.step_seed += 1
        if self.seed_dev is None or self.seed_dev.device != self.device:
            self.seed_dev = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.seed_dev.fill_((self.seed * 1000003 + self.step_seed) & 0xFFFFFFFFFFFF)

    # ---- variables (tf.variable_scope / slim variables) ---------------------------------
    def full_name(self, leaf):
        return "/".join(self.scope + [leaf])

    def allocate_variables(self, specs, seed=1):
        """Create every variable of `specs` ([(name, shape)] in creation order) inside flat buckets."""
        dev = self.device
        total = sum(int(np.prod(s)) for _, s in specs)
        self.flat_param = torch.empty(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_m = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_v = torch.zeros(total, dtype=torch.float32, device=dev)
        self.vars = OrderedDict()
        self.var_grads = {}
        rng = np.random.default_rng(seed)
        host = np.empty(total, np.float32)
        off = 0
        for name, shape in specs:
            n = int(np.prod(shape))
            if name.endswith("weights"):     # slim default: xavier_initializer (uniform)
                lim = np.sqrt(6.0 / (shape[0] + shape[1]))
                host[off:off + n] = rng.uniform(-lim, lim, size=shape).astype(np.float32).reshape(-1)
            else:                            # BatchNorm beta: zeros
                host[off:off + n] = 0
            self.vars[name] = self.flat_param[off:off + n].view(*shape)
            self.var_grads[name] = self.flat_grad[off:off + n].view(*shape)
            off += n
        self.flat_param.copy_(torch.from_numpy(host))
        self.adam_t = 0

    def get_variable(self, leaf, shape):
        name = self.full_name(leaf)
        v = self.vars.get(name)
        if v is None:        # stand-alone use of an op outside trainval: create on first use
            dev = self.device
            if self._rng is None:
                self._rng = np.random.default_rng(self.seed)
            if leaf.endswith("weights"):
                lim = np.sqrt(6.0 / (shape[0] + shape[1]))
                host = self._rng.uniform(-lim, lim, size=shape).astype(np.float32)
            else:
                host = np.zeros(shape, np.float32)
            v = torch.from_numpy(host).to(dev)
            self.vars[name] = v
            self.var_grads[name] = torch.zeros_like(v)
        if tuple(v.shape) != tuple(shape):
            raise ValueError("variable %s has shape %s, requested %s" % (name, tuple(v.shape), tuple(shape)))
        return name, v

    def set_variable(self, name, array):
        self.vars[name].copy_(torch.as_tensor(np.ascontiguousarray(array, np.float32)).to(self.vars[name].device))

    # ---- gradient buffers ---------------------------------------------------------------
    def new_buffer(self, rows, cols):
        t = torch.empty((rows, cols), dtype=torch.float32, device=self.device)
        if self.recording:
            self.roots.append((t, [None]))
        return t

    def grad(self, v):
        """Gradient view matching the 2-D view `v` (same slice of its root's gradient) or None."""
        if not self.recording:
            return None
        ptr = v.data_ptr()
        for root, slot in self.roots:
            base = root.data_ptr()
            if base <= ptr < base + root.numel() * 4:
                if slot[0] is None:
                    slot[0] = self.zeros_like_small(root)
                off = (ptr - base) // 4
                ld = root.shape[1]
                r0, c0 = off // ld, off % ld
                assert H.ld2(v) == ld or v.shape[0] == 1, (v.shape, v.stride(), ld)
                return slot[0][r0:r0 + v.shape[0], c0:c0 + v.shape[1]]
        return None

    def tracked(self, v):
        """True when `v` lies inside a buffer whose gradient is being tracked (no allocation, unlike grad())."""
        if not self.recording:
            return False
        ptr = v.data_ptr()
        return any(root.data_ptr() <= ptr < root.data_ptr() + root.numel() * 4 for root, _ in self.roots)

    def grad_w(self, v):
        """(gradient view, beta) for a GEMM that adds into d(v): beta = 0.0 -- and the buffer is left
        uninitialised instead of zero-filled -- when this is the first touch and v IS its whole root
        (saves a memset and a read-modify-write pass over e.g. the 340 MB d(big) of FC0's dgrad)."""
        if not self.recording:
            return None, 1.0
        ptr = v.data_ptr()
        for root, slot in self.roots:
            if root.data_ptr() == ptr and tuple(root.shape) == tuple(v.shape) and slot[0] is None:
                slot[0] = torch.empty_like(root)
                return slot[0], 0.0
        return self.grad(v), 1.0

    def backward(self):
        self.join_side()                 # side work issued during the forward (transposed adjacency)
        for fn in reversed(self.tape):
            fn()
        self.join_side()                 # before the tape's tensors are released
        self.tape = []


_CTX = Context()


def ctx():
    return _CTX


def reset():
    """Forget all variables / state (tf.reset_default_graph analogue)."""
    global _CTX
    _CTX = Context()
    if H.STAT_SLOTS != 32:
        H.set_stat_slots(32)                 # (the first step picks the count of its mode and shape)
    return _CTX


@contextlib.contextmanager
def variable_scope(name):
    c = ctx()
    c.scope.append(name)
    try:
        yield
    finally:
        c.scope.pop()


# ----------------------------------------------------------------------------------------------
# shape helpers
# ----------------------------------------------------------------------------------------------
def as2d(t):
    """(B,N,C) / (B,N,1,C) / (R,C) -> ((R,C) row-major view with stride (ld,1), B, N)."""
    H.require_gpu(t)
    H.f32(t)
    if t.dim() == 4:
        if t.shape[2] != 1:
            raise ValueError("rank-4 inputs must have a singleton axis -2, got %s" % (tuple(t.shape),))
        t = t[:, :, 0, :]
    if t.dim() == 2:
        if t.stride(1) != 1 and t.shape[1] != 1:
            t = t.contiguous()
        return t, 1, t.shape[0]
    if t.dim() != 3:
        raise ValueError("expected a rank 2/3/4 tensor, got rank %d" % t.dim())
    B, N, C = t.shape
    ok = (t.stride(2) == 1 or C == 1) and (B == 1 or t.stride(0) == N * t.stride(1)) and t.stride(1) >= C
    if not ok:
        t = t.contiguous()
    v = t.as_strided((B * N, C), (t.stride(1), 1), t.storage_offset())
    return v, B, N


def rank4(v, B, N):
    """(R,C) view -> (B,N,1,C) view (the rank the reference's ops return, ops.py:73)."""
    return v.as_strided((B, N, 1, v.shape[1]), (N * v.stride(0), v.stride(0), v.stride(0), 1), v.storage_offset())


# ----------------------------------------------------------------------------------------------
# thin kernel wrappers (2-D views in, explicit leading dimensions out)
# ----------------------------------------------------------------------------------------------
KNN_SEED = os.environ.get("DGCNN_KNN_SEED", "1") != "0"   # seed a layer's k-NN filter with the previous layer's graph (A/B switch)


def knn(x2d, B, N, k, seed=None):
    """idx (B,N,k) of x2d (B*N, C).  seed: an earlier graph of the same clouds, (B,N,ks) int32 with ks >= k -- ks distinct candidates
    per row whose largest distance bounds the row's k-th distance from above (dgcnn_knn_seeded_f32); the result does not depend on it."""
    C = x2d.shape[1]
    idx = torch.empty((B, N, k), dtype=torch.int32, device=x2d.device)
    nws = int(H.load().dgcnn_knn_workspace_bytes(B, N, C, k))           # s_i, seed bounds (+ the cell grid's scratch for raw coordinates)
    ws = torch.empty((nws,), dtype=torch.uint8, device=x2d.device)
    Cp, kc = (4 if C <= 4 else 16 if C <= 16 else 64 if C <= 64 else 128), (8 if k <= 8 else 20 if k <= 20 else 40 if k <= 40 else 64)
    seeded = (KNN_SEED and seed is not None and seed.dim() == 3 and seed.shape[0] == B and seed.shape[1] == N and seed.shape[2] >= k
              and seed.dtype == torch.int32 and seed.is_contiguous())
    # bench.py's tag of the CALL: the kernels it launches (the library picks the form: knn.hip:knn_dispatch)
    if seeded and 16 < C <= 64:
        tag = "knn_call<C%d,k%d>[sqnorm_kernel+knn_seed_bound_kernel+knn_bf16a_kernel+knn_select_kernel]" % (Cp, kc)
    elif C <= 4:
        tag = "knn_call<C%d,k%d>[%s]" % (Cp, kc, "knn_grid_*" if N >= 4096 else "sqnorm_kernel+knn_hist_bound_kernel+knn_kernel")
    else:
        tag = "knn_call<C%d,k%d>[sqnorm_kernel+%s]" % (Cp, kc, "knn_bf16f_kernel" if N >= 8192 else "knn_mfma_kernel")
    if (KNN_SEED and seed is not None and seed.dim() == 3 and seed.shape[0] == B and seed.shape[1] == N and seed.shape[2] >= k and
            seed.dtype == torch.int32 and seed.is_contiguous()):
        H.call("dgcnn_knn_seeded_f32", x2d.data_ptr(), B, N, C, H.ld2(x2d), k, seed.data_ptr(), int(seed.shape[2]), int(seed.shape[2]),
               idx.data_ptr(), ws.data_ptr(), nws, tag=tag, work=2.0 * B * N * N * C)
    else:
        H.call("dgcnn_knn_f32", x2d.data_ptr(), B, N, C, H.ld2(x2d), k, idx.data_ptr(), ws.data_ptr(), nws, tag=tag,
               work=2.0 * B * N * N * C)
    return idx


def _tile_m(M, N):
    """Mirror of gemm.hip:tile_m (only used to name the kernel instance in bench.py's roofline tags)."""
    return 128


def _gemm_tag(M, N, K, transA, transB, A, Bm):
    """Kernel identity of a dgcnn_gemm_f32 launch for bench.py's table (mirrors the C dispatch); None when nobody is timing."""
    if H.TIMER is None:
        return None
    lda, ldb = H.ld2(A), H.ld2(Bm)
    vec = (lda % 4 == 0 and ldb % 4 == 0 and A.data_ptr() % 16 == 0 and Bm.data_ptr() % 16 == 0 and
           ((M if transA else K) % 4 == 0) and ((K if transB else N) % 4 == 0))
    arith = H.gemm_arith()
    cd = lambda a, b: -(-a // b)
    small = cd(M, 128) * cd(N, 128) * cd(K, 256) < 256       # gemm.hip:tile_n
    bn = 64 if (N <= 64 or small) else 128
    if vec and arith:
        kinds = ("KSTRIDED" if transA else "KCONTIG", "KCONTIG" if transB else "KSTRIDED")
        rows = H.load().dgcnn_gemm_x3_tile_rows(M, N, K)
        if H.load().dgcnn_gemm_x3_tile_cols(M, N, K) == 256:     # dg::x3_tile_n: the 256-column unspecialised kernels
            return "gemm_x3q_kernel<%s,%s,%d,bf16x%d>" % (kinds + (rows, arith))
        if rows == 256:                                           # dg::x3_tile_m: the 256 x 128 wave-specialised kernel
            return "gemm_x3w2_kernel<%s,%s,bf16x%d>" % (kinds + (arith,))
        return "gemm_x3_kernel<%s,%s,%d,bf16x%d>" % (kinds + (bn, arith))
    return "gemm_kernel<%s,%s,STORE,%d,%d>" % ("A_COL" if transA else "A_ROW", "B_COL" if transB else "B_ROW", _tile_m(M, N), bn)


def gemm(A, Bm, C, transA=False, transB=False, beta=0.0, gbias=None, rpg=0, stats=None, arith=None, colmax=None, colmax_rpg=0):
    """C (+)= op(A) op(B); shapes are those of the stored matrices.  arith: arithmetic of THIS product (None = the
    process-wide setting; 1 = plain bf16 operands: measurements only, not fp32 class)."""
    if arith is not None and arith != H.gemm_arith():
        prev = H.gemm_arith()
        H.set_gemm_arith(arith)
        try:
            return gemm(A, Bm, C, transA, transB, beta, gbias, rpg, stats, None, colmax, colmax_rpg)
        finally:
            H.set_gemm_arith(prev)
    M = A.shape[1] if transA else A.shape[0]
    K = A.shape[0] if transA else A.shape[1]
    N = Bm.shape[0] if transB else Bm.shape[1]
    Kb = Bm.shape[1] if transB else Bm.shape[0]
    assert K == Kb and tuple(C.shape) == (M, N), (A.shape, Bm.shape, C.shape, transA, transB)
    ws = ctx().workspace()
    tag = _gemm_tag(M, N, K, transA, transB, A, Bm)
    H.call("dgcnn_gemm_f32", int(transA), int(transB), M, N, K, A.data_ptr(), H.ld2(A), Bm.data_ptr(), H.ld2(Bm),
           C.data_ptr(), H.ld2(C), float(beta), H._p(gbias), 0 if gbias is None else H.ld2(gbias), int(rpg),
           H._p(stats), H._p(colmax), int(colmax_rpg), ws.data_ptr(), ws.numel(), tag=tag, work=2.0 * M * N * K,
           nbytes=4.0 * (M * K + K * N + (2 if beta != 0.0 else 1) * M * N))


def colstats_det(T, st):
    """Fixed-order BatchNorm column sums of the materialised (R,F) tensor T into slot 0 of `st` (deterministic mode)."""
    ws = ctx().workspace()
    H.call("dgcnn_colstats_det_f32", T.data_ptr(), T.shape[0], T.shape[1], H.ld2(T), st.data_ptr(), ws.data_ptr(), ws.numel())


def k1_reduce_fixed_order(Y, k, F, mean, rstd, beta, dmx, dmn, mx):
    """The k = 1 BatchNorm-backward reduce kernel (float4 channels, column-fixed threads) sums in a fixed order by itself."""
    return (k == 1 and F % 4 == 0 and mx is None and Y.data_ptr() % 16 == 0 and dmx.data_ptr() % 16 == 0 and H.ld2(dmx) % 4 == 0 and
            mean.data_ptr() % 16 == 0 and rstd.data_ptr() % 16 == 0 and beta.data_ptr() % 16 == 0 and
            (dmn is None or (dmn.data_ptr() % 16 == 0 and H.ld2(dmn) % 4 == 0)))


def bn_bwd_reduce(Y, R, k, F, mean, rstd, beta, relu, dmx, dmn, mx, cnt, red, tag, work):
    """sum dZ / sum dZ*xhat of a materialised Y.  DETERMINISTIC: the column-fixed k = 1 kernel (float4 channels) reduces in a fixed
    order by itself; every other shape (the class dimension, materialised edge tensors) takes the fixed-order twin of det.hip."""
    k1 = k1_reduce_fixed_order(Y, k, F, mean, rstd, beta, dmx, dmn, mx)
    if DETERMINISTIC and not k1:
        ws = ctx().workspace()
        H.call("dgcnn_bn_bwd_reduce_det_f32", Y.data_ptr(), R, k, F, mean.data_ptr(), rstd.data_ptr(), beta.data_ptr(), int(relu),
               dmx.data_ptr(), H.ld2(dmx), H._p(dmn), 0 if dmn is None else H.ld2(dmn), H._p(mx), 0 if mx is None else H.ld2(mx),
               H._p(cnt), red.data_ptr(), ws.data_ptr(), ws.numel())
    else:
        H.call("dgcnn_bn_bwd_reduce_f32", Y.data_ptr(), R, k, F, mean.data_ptr(), rstd.data_ptr(), beta.data_ptr(), int(relu),
               dmx.data_ptr(), H.ld2(dmx), H._p(dmn), 0 if dmn is None else H.ld2(dmn), H._p(mx), 0 if mx is None else H.ld2(mx),
               H._p(cnt), red.data_ptr(), tag=tag, work=work)


def bn_finalize(stats, F, count):
    dev = stats.device
    mr = torch.empty((2, F), dtype=torch.float32, device=dev)
    H.call("dgcnn_bn_finalize_f32", stats.data_ptr(), F, float(count), BN_EPS, mr[0].data_ptr(), mr[1].data_ptr())
    return mr[0], mr[1]


# ----------------------------------------------------------------------------------------------
# slim.conv2d(1x1, no bias) + slim.batch_norm + activation on per-point tensors (k = 1)
# dgcnn/ops.py:62-70,125-133,153-160 ; dgcnn/model.py:46-53,65-72,94-101
# ----------------------------------------------------------------------------------------------
def conv_bn_act(x, leaf_scope, num_outputs, relu=True, out=None, out2=None, gbias=None, rpg=0, w_rows=None, arith=None,
                plane_out=None, f32_out=True, gmax=None, drop_keep=None):
    """x: (R,Cin) view.  Variables `<scope>/weights` [Cin(+extra), Cout], `<scope>/BatchNorm/beta`.
    w_rows: (lo, hi) row range of the weight that multiplies x (FC0 with the folded global feature).
    Returns the (R,Cout) output (a fresh tracked buffer unless `out` is given).
    Plane mode (HEAD_PLANES, planes_ok): the products run on the plane GEMM; plane_out = PlaneSet view that receives the
    activated output as operand planes of the NEXT product (f32_out = False: the fp32 `out` is then never written -- it only
    names the tensor and carries its gradient); gmax = (B, N): also return the per-cloud max over the points of the output
    (model.py:76-77), taken on the GEMM output and normalised afterwards (BN + ReLU are monotone).
    drop_keep: tf.nn.dropout(out, drop_keep) behind the layer (model.py:90-91), fused into the BatchNorm passes where the
    kernels allow it (the returned tensor is the DROPPED output either way)."""
    c = ctx()
    R, Cin = x.shape
    with variable_scope(leaf_scope):
        if w_rows is None:
            wname, W = c.get_variable("weights", (Cin, num_outputs))
            Wx = W
        else:
            wname, W = c.get_variable("weights", (w_rows[2], num_outputs))
            Wx = W[w_rows[0]:w_rows[1]]
        bname, beta = c.get_variable("BatchNorm/beta", (num_outputs,))
    F = num_outputs
    T = torch.empty((R, F), dtype=torch.float32, device=x.device)
    # slots of this layer's statistics buffer = row tiles of its GEMM (asked from the library; another arithmetic: the step's maximum)
    # (0 = a kernel whose grid FOLLOWS the slot count -- the class dimension's streaming product: the step's maximum, more writers)
    c.push_slots((H.load().dgcnn_gemm_stat_writers(0, R, F, Cin, x.data_ptr(), H.ld2(x), Wx.data_ptr(), H.ld2(Wx)) or c.step_slots)
                 if arith is None else c.step_slots)
    st = c.stats(F)
    use_pl = arith is None and planes_ok(R, Cin, F)
    xp = None
    # the per-cloud column maximum (model.py:76-77), when asked for, comes out of the GEMM's epilogue as packed (value, first row)
    # keys -- a tile of 256 rows must lie inside one cloud; otherwise a separate pass over T below
    keys = None
    if gmax is not None and gmax[1] % 256 == 0 and gmax[0] * gmax[1] == R and F > 4:
        keys = c.stats_raw(gmax[0] * F)                                    # zeroed uint64[B][F]
    if use_pl:
        c.ensure_plane_scales(R)
        xp = c.planes_of(x)                                                # (R, Cin) activations
        wt = prepared(("wT", Wx.data_ptr()))                               # (F rows, Cin channels) = W^T
        if wt is None:
            wt = c.new_planes(F, Cin, "w").fill_from(Wx, transpose=True)
        PL.gemm(PL.KC, xp, wt, T, gbias=gbias, rpg=rpg, stats=st, colmax=keys, colmax_rpg=0 if keys is None else gmax[1])
    else:
        plane_out, f32_out = None, True
        gemm(x, Wx, T, gbias=gbias, rpg=rpg, stats=st, arith=arith, colmax=keys, colmax_rpg=0 if keys is None else gmax[1])
    mean, rstd = bn_finalize(st, F, R)
    c.pop_slots()
    if out is None:
        out = c.new_buffer(R, F)
    fuse_drop = (drop_keep is not None and FUSE_DROPOUT and not use_pl and F % 4 == 0 and out2 is None and
                 gmax is None and plane_out is None)
    if fuse_drop:
        if c.seed_dev is None:
            c.advance_seed()
        seed = c.seed_dev                       # device-resident: the backward regenerates the mask from the same value
        H.call("dgcnn_bn1_act_dropout_f32", T.data_ptr(), R, F, mean.data_ptr(), rstd.data_ptr(), beta.data_ptr(), int(relu),
               float(drop_keep), seed.data_ptr(), out.data_ptr(), H.ld2(out), tag="bn_act_kreduce_kernel<k=1>", work=4.0 * R * F * 2)
    elif plane_out is not None:
        H.call("dgcnn_bn_act_planes_f32", T.data_ptr(), F, R, F, mean.data_ptr(), rstd.data_ptr(), beta.data_ptr(), int(relu),
               plane_out.fmt, H._p(plane_out.scale), plane_out.ptr(), plane_out.plane_stride, plane_out.ra,
               out.data_ptr() if f32_out else 0, H.ld2(out), H._p(out2) if f32_out else 0, 0 if out2 is None else H.ld2(out2),
               tag="bn_act_planes_kernel", work=4.0 * R * F * (2 + (1 if f32_out else 0)))
        c.planes[c.plane_key(out)] = plane_out
    else:
        H.call("dgcnn_bn_act_kreduce_f32", T.data_ptr(), R, 1, F, mean.data_ptr(), rstd.data_ptr(), beta.data_ptr(),
               int(relu), out.data_ptr(), H.ld2(out), 0, 0, H._p(out2), 0 if out2 is None else H.ld2(out2), 0,
               tag="bn_act_kreduce_kernel<k=1>", work=4.0 * R * F * (2 if out2 is None else 3))

    if c.recording:
        def bwd():
            c.push_slots(0)                      # (the reduction kernels' grids follow the slot count: 256 writers)
            try:
                bwd_body()
            finally:
                c.pop_slots()

        def bwd_body():
            dout = c.grad(out)
            if dout is None:
                return
            d2 = c.grad(out2) if out2 is not None else None
            # (DETERMINISTIC: unaligned parameter slices send the reduce to the fixed-order twin, which takes one gradient)
            if d2 is not None and (fuse_drop or use_pl or F % 4 != 0 or
                                   (DETERMINISTIC and not k1_reduce_fixed_order(T, 1, F, mean, rstd, beta, dout, d2, None))):
                # the second copy's gradient joins the first (the default passes below read both instead)
                H.call("dgcnn_copy2d_f32", d2.data_ptr(), H.ld2(d2), dout.data_ptr(), H.ld2(dout), R, F, 1)
                d2 = None
            red = c.stats(F)
            dWx = c.var_grads[wname] if w_rows is None else c.var_grads[wname][w_rows[0]:w_rows[1]]
            dgb = c.grad(gbias) if gbias is not None else None
            if fuse_drop:
                # sums, finalise (+ dbeta) and dT (in place of T), all reading d(dropped output) through the regenerated mask
                H.call("dgcnn_bn1_bwd_dropout_f32", T.data_ptr(), R, F, mean.data_ptr(), rstd.data_ptr(), beta.data_ptr(), int(relu),
                       float(drop_keep), seed.data_ptr(), dout.data_ptr(), H.ld2(dout), red.data_ptr(), T.data_ptr(),
                       c.var_grads[bname].data_ptr(), 1.0, tag="bn_bwd_apply_kernel<k=1>", work=4.0 * R * F * 5)
                dT = T
                dx, bx = c.grad_w(x)
                if dx is not None:
                    gemm(dT, Wx, dx, transB=True, beta=bx, arith=arith)       # data gradient first, on the critical path
                with c.off_critical_path(rows=R):
                    gemm(x, dT, dWx, transA=True, beta=1.0, arith=arith)      # weight gradient behind it, on the side stream
                if dgb is not None:
                    tmp = torch.empty_like(gbias)
                    H.call("dgcnn_group_colsum_f32", dT.data_ptr(), H.ld2(dT), gbias.shape[0], rpg, F, tmp.data_ptr())
                    H.call("dgcnn_axpby_f32", tmp.data_ptr(), 1.0, dgb.data_ptr(), 1.0, tmp.numel())
                return
            if use_pl:
                # sums + column maxima -> bound on |dT| -> the dT plane set's scale -> dT written as planes (never as fp32)
                maxbits = c.stats_raw(F + 1)                              # uint32[2 F + 1]: column maxima + the tensor-wide bound
                H.call("dgcnn_bn1_bwd_reduce_max_f32", T.data_ptr(), R, F, mean.data_ptr(), rstd.data_ptr(), beta.data_ptr(),
                       int(relu), dout.data_ptr(), H.ld2(dout), red.data_ptr(), maxbits.data_ptr(),
                       tag="bn1_bwd_reduce_max_kernel", work=4.0 * R * F * 2)
                dTp = PL.PlaneSet(R, F, HEAD_PLANES, device=x.device)
                sc = torch.empty(1, dtype=torch.float32, device=x.device)
                if HEAD_PLANES == PL.F16X2:
                    dTp.scale = sc
                fused_gsum = dgb is not None and rpg % 64 == 0
                tmp = H.memset(torch.empty_like(gbias)) if fused_gsum else None
                dT32 = T if (dgb is not None and not fused_gsum) else None        # (in place: every element is read before it is written)
                H.call("dgcnn_bn1_bwd_apply_planes_f32", T.data_ptr(), R, F, mean.data_ptr(), rstd.data_ptr(), beta.data_ptr(),
                       int(relu), dout.data_ptr(), H.ld2(dout), red.data_ptr(), maxbits.data_ptr(), HEAD_PLANES, sc.data_ptr(),
                       dTp.ptr(), dTp.plane_stride, dTp.ra, H._p(dT32), H._p(tmp), F, int(rpg), c.var_grads[bname].data_ptr(), 1.0,
                       tag="bn1_bwd_apply_planes_kernel", work=4.0 * R * F * 3)
                # data gradient FIRST, on the critical path; the weight gradient is issued behind it on the side stream, so that it
                # runs under the bandwidth-bound BatchNorm / gather passes that follow on the main stream -- not next to the data
                # gradient, another matrix-pipe kernel at the package power cap (both then just run at half speed:
                # profiles/r03/wgrad_order.txt)
                dx, bx = c.grad_w(x)
                if dx is not None:
                    wd = prepared(("w", Wx.data_ptr()))                                  # (Cin rows, F channels)
                    if wd is None:
                        wd = c.new_planes(Wx.shape[0], F, "w").fill_from(Wx)
                    PL.gemm(PL.KC, dTp, wd, dx, beta=bx)                                 # dx (+)= dT W^T
                with c.off_critical_path(dTp.buf, sc, rows=R):
                    PL.gemm(PL.TR, xp, dTp, dWx, beta=1.0, ws=c.workspace())           # dW += x^T dT
                if dgb is not None:                                                     # tf.tile^T: sum over the cloud
                    if not fused_gsum:
                        tmp = torch.empty_like(gbias)
                        H.call("dgcnn_group_colsum_f32", dT32.data_ptr(), F, gbias.shape[0], rpg, F, tmp.data_ptr())
                    H.call("dgcnn_axpby_f32", tmp.data_ptr(), 1.0, dgb.data_ptr(), 1.0, tmp.numel())
                return
            nsrc = 1 if d2 is None else 2                                  # k = 1: dz = dout + d2 (the kernels' dmax + dmean / k)
            bn_bwd_reduce(T, R, 1, F, mean, rstd, beta, relu, dout, d2, None, None, red,
                          tag="bn_bwd_reduce_kernel<k=1>", work=4.0 * R * F * (1 + nsrc))
            H.call("dgcnn_bn_bwd_apply_f32", T.data_ptr(), R, 1, F, mean.data_ptr(), rstd.data_ptr(), beta.data_ptr(),
                   int(relu), dout.data_ptr(), H.ld2(dout), H._p(d2), 0 if d2 is None else H.ld2(d2), 0, 0, 0, red.data_ptr(),
                   T.data_ptr(), 0, 0, c.var_grads[bname].data_ptr(), 1.0, tag="bn_bwd_apply_kernel<k=1>", work=4.0 * R * F * (2 + nsrc))
            dT = T
            # data gradient FIRST, on the critical path; the weight gradient is issued behind it on the side stream, so that it
            # runs under the bandwidth-bound BatchNorm / gather passes that follow on the main stream -- not next to the data
            # gradient, another matrix-pipe kernel at the package power cap (both then run at half speed: profiles/r03/wgrad_order.txt)
            dx, bx = c.grad_w(x)
            if dx is not None:
                gemm(dT, Wx, dx, transB=True, beta=bx, arith=arith)    # dx (+)= dT W^T
            with c.off_critical_path(rows=R):
                gemm(x, dT, dWx, transA=True, beta=1.0, arith=arith)   # dW += x^T dT
            if dgb is not None:                                         # tf.tile^T: sum over the cloud
                tmp = torch.empty_like(gbias)
                H.call("dgcnn_group_colsum_f32", dT.data_ptr(), H.ld2(dT), gbias.shape[0], rpg, F, tmp.data_ptr())
                H.call("dgcnn_axpby_f32", tmp.data_ptr(), 1.0, dgb.data_ptr(), 1.0, tmp.numel())
        c.tape.append(bwd)
    if gmax is None:
        if drop_keep is not None and not fuse_drop:
            return dropout(out, drop_keep)         # (plane / deterministic modes: the separate pass)
        return out
    # model.py:76-77 max_pool over the points of each cloud, on the GEMM output: z = relu((t - mean) rstd + beta) is
    # non-decreasing in t, so max_n z[n] = z(max_n t[n]) and the first arg-max of t is an arg-max of z
    Bc, Nc = gmax
    graw = torch.empty((Bc, F), dtype=torch.float32, device=x.device)
    arg = torch.empty((Bc, F), dtype=torch.int32, device=x.device)
    if keys is not None:
        H.call("dgcnn_colmax_decode_f32", keys.data_ptr(), Bc * F, graw.data_ptr(), arg.data_ptr())
    else:
        H.call("dgcnn_global_max_f32", T.data_ptr(), F, Bc, Nc, F, graw.data_ptr(), arg.data_ptr())
    g = c.new_buffer(Bc, F)
    H.call("dgcnn_bn_act_kreduce_f32", graw.data_ptr(), Bc, 1, F, mean.data_ptr(), rstd.data_ptr(), beta.data_ptr(),
           int(relu), g.data_ptr(), F, 0, 0, 0, 0, 0)
    if c.recording:
        def bwd_g():
            dg_, dout = c.grad(g), c.grad(out)
            if dg_ is None or dout is None:
                return
            H.call("dgcnn_global_max_bwd_f32", dg_.data_ptr(), arg.data_ptr(), Bc, Nc, F, dout.data_ptr(), H.ld2(dout))
        c.tape.append(bwd_g)
    return out, g


# ----------------------------------------------------------------------------------------------
# dgcnn/ops.py:42-73 edge_conv as one block
# ----------------------------------------------------------------------------------------------
def prepare_step_weights(edge_w0, head_w):
    """Everything that depends on the PARAMETERS only -- conv0's folded weights [Wa - Wb | Wb] of every EdgeConv layer and, in
    plane mode, the head weights as operand planes in both orientations -- issued at the start of the step on the side stream,
    where it runs under the first k-NN instead of as ~10 six-microsecond kernels on the critical path.
    edge_w0: [(W0 (2C, F), C, F)]; head_w: [Wx views] (plane mode, else empty)."""
    c = ctx()
    if not (WGRAD_SIDE_STREAM and c.flat_param is not None) or c.wprep:
        return
    dev = c.device
    items, temps = [], []
    for W0, C, F in edge_w0:
        Cp = (C + 3) // 4 * 4
        wcat = torch.empty((Cp, 2 * F), dtype=torch.float32, device=dev)
        if Cp != C:
            H.memset(wcat)
        items.append((("wcat", W0.data_ptr()), wcat, W0, C, F))
        temps.append(wcat)
    planes = []
    for Wx in head_w:
        wt = c.new_planes(Wx.shape[1], Wx.shape[0], "w")
        wd = c.new_planes(Wx.shape[0], Wx.shape[1], "w")
        planes.append((Wx, wt, wd))
        temps += [wt.buf, wd.buf]
    with c.off_critical_path(*temps):
        for key, wcat, W0, C, F in items:
            H.call("dgcnn_edge_weight_split_f32", W0.data_ptr(), C, F, wcat.data_ptr())
            c.wprep[key] = wcat
        for Wx, wt, wd in planes:
            c.wprep[("wT", Wx.data_ptr())] = wt.fill_from(Wx, transpose=True)
            c.wprep[("w", Wx.data_ptr())] = wd.fill_from(Wx)
        if c.side is not None and torch.cuda.current_stream() == c.side:
            c.wprep_event = True          # the first consumer makes the main stream wait for the side stream (prepared())


def prepared(key):
    """A tensor prepared by prepare_step_weights (the main stream is made to wait for the preparation once), or None."""
    c = ctx()
    v = c.wprep.get(key)
    if v is not None and c.wprep_event:
        # (waits for everything on the side stream so far: between the preparation and its first consumer -- conv0 of the first
        # EdgeConv layer, right behind the first k-NN -- nothing else is issued there)
        H.stream_wait(torch.cuda.current_stream(), c.side)
        c.wprep_event = None
    return v


def _point_gemm(c, x, W0, R, C, F):
    """Wcat = [Wa - Wb | Wb] and [U | V] = X Wcat (conv0 folded to the points).  C = 3 (raw coordinates): the reduction
    dimension is padded to 4 with a zero column / zero weight row so that the GEMM takes the float4 path."""
    Cp = (C + 3) // 4 * 4
    xg = x
    ready = prepared(("wcat", W0.data_ptr()))
    if Cp != C:
        xg = torch.empty((R, Cp), dtype=torch.float32, device=x.device)
        wcat = ready if ready is not None else H.zeros((Cp, 2 * F), torch.float32, x.device)
    else:
        wcat = ready if ready is not None else torch.empty((C, 2 * F), dtype=torch.float32, device=x.device)
    UV = torch.empty((R, 2 * F), dtype=torch.float32, device=x.device)

    if Cp != C:
        H.call("dgcnn_pad_copy_f32", x.data_ptr(), H.ld2(x), C, xg.data_ptr(), Cp, R)
    if ready is None:
        H.call("dgcnn_edge_weight_split_f32", W0.data_ptr(), C, F, wcat.data_ptr())
    gemm(xg, wcat, UV, arith=None)
    return Cp, xg, wcat, UV


def build_csr(idx, B, N, k, side=None):
    """Transposed adjacency of idx (B,N,k): (off (R+1), rev (R*k)) -- for every point the ids e = point * k + m of the edges that
    point at it.  DETERMINISTIC: every bucket in ascending edge order (the build fills buckets through LDS cursors: any order).
    side = (ctx, rows): the kernels run on the side stream (the buffers are allocated on the current one and handed over)."""
    R = B * N
    dev = idx.device
    cws = torch.empty(2 * R, dtype=torch.int32, device=dev)
    off = torch.empty(R + 1, dtype=torch.int32, device=dev)
    rev = torch.empty(R * k, dtype=torch.int32, device=dev)
    srt = torch.empty(R * k, dtype=torch.int32, device=dev) if DETERMINISTIC else None

    def fill():
        H.call("dgcnn_edge_csr_build", idx.data_ptr(), B, N, k, cws.data_ptr(), off.data_ptr(), rev.data_ptr())
        if srt is not None:
            H.call("dgcnn_edge_csr_sort", idx.data_ptr(), B, N, k, off.data_ptr(), rev.data_ptr(), srt.data_ptr())
    if side is None:
        fill()
    else:
        with side[0].off_critical_path(*[t for t in (cws, off, rev, srt) if t is not None], rows=side[1]):
            fill()
    return off, (rev if srt is None else srt)


def edge_conv_block(x, B, N, k, num_filters, relu1=True, outs=None, net2=None, seed=None):
    """x: (B*N, C) view.  Returns (mm, net, idx): mm = (R,2F) [max | mean], net = (R,64).
    outs = (mm_view, net_view) destination slices (model path) or None (fresh buffers).
    seed: the previous EdgeConv layer's graph (B,N,k') or None: seeds this layer's k-NN filter (same result, fewer inserts)."""
    c = ctx()
    R, C = x.shape
    F = int(num_filters)
    if k > N:
        raise ValueError("k_nn: k=%d > N=%d (tf.nn.top_k raises InvalidArgument)" % (k, N))
    with variable_scope("conv0"):
        w0name, W0 = c.get_variable("weights", (2 * C, F))
        b0name, beta0 = c.get_variable("BatchNorm/beta", (F,))
    c.push_slots(c.step_slots)                   # conv0's statistics come from passes whose grids FOLLOW the slot count: the maximum
    st = c.stats(F)
    bf16 = EDGE_MLP_DTYPE == "bf16"
    literal = EDGE_MLP_LITERAL or bf16
    gather = (not literal) and (not EDGE_MLP_NBR_GEMM) and F % 4 == 0 and F <= 1024
    idx = knn(x, B, N, k, seed=seed)                                    # ops.py:8-19
    virtual = gather and not EDGE_MATERIALIZE_Y and k < 256   # conv0 output never written: recomputed from (V, U, idx)
    #                                                           (the edge BN passes pack tie / positive counts: k < 256)
    # the fused bf16 kernels write neither E nor y: decided BEFORE anything of (R*k, F) is allocated (1.3 GB per layer at configs[2])
    fused_bf16 = (bf16 and bool(H.load().dgcnn_edge_mlp_bf16_supported(C, k, F)) and W0.is_contiguous() and
                  (C <= 4 or (H.ld2(x) % 4 == 0 and x.data_ptr() % 16 == 0)))
    Y = None if (virtual or fused_bf16) else torch.empty((R * k, F), dtype=torch.float32, device=x.device)
    wd = wcat = UV = None
    Ee = W0p = None
    fused_bwd = False
    Cp = (C + 3) // 4 * 4

    def bf16_operands():
        """The literal edge tensor in fp32 (the difference x_j - x_i BEFORE any rounding; channels padded to a multiple of 4 so
        that the products take the float4 path) and the matching weight rows -- the operands the bf16 products round."""
        xg, Wp = x, W0
        if Cp != C:
            xg = H.zeros((R, Cp), torch.float32, x.device)
            H.call("dgcnn_copy2d_f32", x.data_ptr(), H.ld2(x), xg.data_ptr(), Cp, R, C, 0)
            Wp = H.zeros((2 * Cp, F), torch.float32, x.device)
            H.call("dgcnn_copy2d_f32", W0[:C].data_ptr(), F, Wp[:C].data_ptr(), F, C, F, 0)
            H.call("dgcnn_copy2d_f32", W0[C:].data_ptr(), F, Wp[Cp:Cp + C].data_ptr(), F, C, F, 0)
        E_ = torch.empty((R * k, 2 * Cp), dtype=torch.float32, device=x.device)
        H.call("dgcnn_edge_gather_f32", xg.data_ptr(), H.ld2(xg), idx.data_ptr(), B, N, Cp, k, E_.data_ptr())   # ops.py:21-40
        return E_, Wp

    if bf16:
        bsrc = (x.data_ptr(), H.ld2(x), idx.data_ptr(), W0.data_ptr(), B, N, C, k, F)
        # the backward in one pass over the edges (8 <= k < 256): the forward then packs (#ties, #positives) as the fp32 edge kernels do
        fused_bwd = fused_bf16 and c.recording and BF16_FUSED_BWD and bool(H.load().dgcnn_edge_mlp_bf16_bwd_supported(C, k, F))
        if fused_bf16:
            # csrc/edge_mlp_bf16.hip: E = [x_i, x_j - x_i] gathered, rounded and multiplied tile by tile on the bf16 MFMA pipe;
            # neither E nor y is written: this pass takes the BatchNorm sums, the next one recomputes y for BN + ReLU + max / mean
            H.call("dgcnn_edge_mlp_bf16_stats", *bsrc, st.data_ptr(), tag="edge_mlp_bf16_kernel<stats>",
                   work=2.0 * R * k * 2 * C * F, nbytes=4.0 * (R * k * C + R * C))
        else:
            # shapes the fused kernels do not take: the edge tensor in fp32, then ONE product whose operands the GEMM rounds to
            # bf16 (arith = 1: v_cvt_pk_bf16_f32, round to nearest even) with fp32 accumulation on the bf16 MFMA pipe
            Ee, W0p = bf16_operands()
            gemm(Ee, W0p, Y, stats=None if DETERMINISTIC else st, arith=1)                                      # ops.py:47-52
            if DETERMINISTIC:
                colstats_det(Y, st)
            if not c.recording:
                Ee = None
    elif literal:
        H.call("dgcnn_edge_mlp_f32", x.data_ptr(), H.ld2(x), idx.data_ptr(), W0.data_ptr(), B, N, C, k, F,
               Y.data_ptr(), 0 if DETERMINISTIC else st.data_ptr(),
               tag="gemm_kernel<A_EDGE,B_ROW,STORE,%d,%d>" % (_tile_m(R * k, F), 64 if F <= 64 else 128),
               work=2.0 * R * k * 2 * C * F)                            # ops.py:21-52 (gather fused)
        if DETERMINISTIC:
            colstats_det(Y, st)
    elif gather:
        # conv0 is linear: E W0 = x_i (Wa-Wb) + x_j Wb = U[i] + V[j] with [U | V] = X [Wa-Wb | Wb] -- ONE
        # point-level GEMM (k times fewer MACs than the edge tensor product), then a per-edge gather-add
        # bound by the HBM write of Y, which also takes the BatchNorm column sums.
        # C = 3 (raw coordinates): pad the reduction dimension to 4 with a zero column / zero weight row so that
        # the point-level GEMMs take the float4 path (the generic scalar kernel costs ~10x more on them)
        Cp, xg, wcat, UV = _point_gemm(c, x, W0, R, C, F)
        H.call("dgcnn_edge_gather_add_f32", UV[:, F:].data_ptr(), 2 * F, UV.data_ptr(), 2 * F, idx.data_ptr(),
               B, N, k, F, H._p(Y), st.data_ptr(),
               tag="edge_gather_add_kernel", work=4.0 * ((0 if virtual else R * k * F) + 2 * R * F) + 4.0 * R * k,
               nbytes=4.0 * (R * k * F + R * F))   # ops.py:21-52   (nbytes: rows gathered from L2 -- bench.py's l2-gather roof)
        if not virtual:
            UV = None
    else:
        # factored conv0, edge-level form (kept for odd F and as a cross-check): centre term once per point
        # (U); the per-edge (B*N*k) x C GEMM gathers the neighbour rows and adds U[point] in its epilogue.
        wd = torch.empty((C, F), dtype=torch.float32, device=x.device)
        H.call("dgcnn_copy2d_f32", W0[:C].data_ptr(), F, wd.data_ptr(), F, C, F, 0)
        H.call("dgcnn_axpby_f32", W0[C:].data_ptr(), -1.0, wd.data_ptr(), 1.0, C * F)
        U = torch.empty((R, F), dtype=torch.float32, device=x.device)
        gemm(x, wd, U)
        H.call("dgcnn_edge_nbr_gemm_f32", x.data_ptr(), H.ld2(x), idx.data_ptr(), W0[C:].data_ptr(), U.data_ptr(), F,
               B, N, C, k, F, Y.data_ptr(), 0 if DETERMINISTIC else st.data_ptr(),
               tag="gemm_kernel<A_EDGE,B_ROW,STORE,%d,%d>" % (_tile_m(R * k, F), 64 if F <= 64 else 128),
               work=2.0 * R * k * C * F)                                # ops.py:21-52 (gather fused)
        if DETERMINISTIC:
            colstats_det(Y, st)
    mean, rstd = bn_finalize(st, F, R * k)                              # ops.py:53
    c.pop_slots()
    if outs is None:
        mm = c.new_buffer(R, 2 * F)
        net_out = None
    else:
        mm, net_out = outs
    mx, mn = mm[:, :F], mm[:, F:]
    cnt = torch.empty((R, F), dtype=torch.float32, device=x.device) if c.recording else None   # ties of the max
    if fused_bf16:
        H.call("dgcnn_edge_mlp_bf16_bn_kreduce", *bsrc, mean.data_ptr(), rstd.data_ptr(), beta0.data_ptr(),
               mx.data_ptr(), H.ld2(mx), mn.data_ptr(), H.ld2(mn), H._p(cnt), 1 if fused_bwd else 0,
               tag="edge_mlp_bf16_kernel<bn_kreduce>", work=2.0 * R * k * 2 * C * F, nbytes=4.0 * (R * k * C + R * C))   # ops.py:53-58
    elif virtual:
        esrc = (UV[:, F:].data_ptr(), 2 * F, UV.data_ptr(), 2 * F, idx.data_ptr(), B, N, k, F)
        H.call("dgcnn_edge_bn_act_kreduce_f32", *esrc, mean.data_ptr(), rstd.data_ptr(), beta0.data_ptr(), 1,
               mx.data_ptr(), H.ld2(mx), mn.data_ptr(), H.ld2(mn), H._p(cnt),
               tag="bn_act_kreduce_kernel<edge>", work=4.0 * (4 * R * F) + 4.0 * R * k, nbytes=4.0 * (R * k * F + R * F))   # ops.py:54-58
    else:
        H.call("dgcnn_bn_act_kreduce_f32", Y.data_ptr(), R, k, F, mean.data_ptr(), rstd.data_ptr(), beta0.data_ptr(), 1,
               mx.data_ptr(), H.ld2(mx), mn.data_ptr(), H.ld2(mn), 0, 0, H._p(cnt),
               tag="bn_act_kreduce_kernel", work=4.0 * (R * k * F + 2 * R * F))   # ops.py:54-58

    fv = F // 4
    fused_l0 = (c.recording and virtual and EDGE_BWD_FUSED_L0 and EDGE_BWD_REDUCE_POINTS and C <= 4 and F % 4 == 0 and
                fv & (fv - 1) == 0 and fv <= 256 and not c.tracked(x))
    csr = None
    if c.recording and gather and WGRAD_SIDE_STREAM and R >= SIDE_STREAM_MIN_ROWS and not fused_l0:
        # the transposed adjacency depends only on idx: build it now, off the critical path, for the backward
        off_t, rev_t = build_csr(idx, B, N, k, side=(c, R))
        csr = (off_t, rev_t)

    if c.recording:
        def bwd():
            c.push_slots(0)
            try:
                bwd_body()
            finally:
                c.pop_slots()

        def bwd_body():
            nonlocal Y, Ee, W0p
            dmm = c.grad(mm)
            if dmm is None:
                return
            dmx, dmn = dmm[:, :F], dmm[:, F:]
            red = c.stats(F)
            if fused_bf16 and fused_bwd:
                # csrc/edge_mlp_bf16.hip: sums of the BatchNorm backward from the forward's per-point outputs, then ONE pass over the
                # edges that recomputes y, forms dY (rounded to bf16), takes dW0 = E^T dY on the matrix pipe and leaves dY as bf16
                # for the transposed-adjacency sum -- E, y and an fp32 dY are never written
                H.call("dgcnn_edge_bn_bwd_reduce_points_f32", mx.data_ptr(), H.ld2(mx), mn.data_ptr(), H.ld2(mn),
                       cnt.data_ptr(), dmx.data_ptr(), H.ld2(dmx), dmn.data_ptr(), H.ld2(dmn), beta0.data_ptr(), R, k, F,
                       red.data_ptr(), tag="edge_bwd_reduce_points_kernel", work=4.0 * 5 * R * F)
                dx = c.grad(x)
                dYb = dysum = None
                if dx is not None:
                    dYb = torch.empty((R * k, F), dtype=torch.bfloat16, device=x.device)
                    dysum = torch.empty((R, F), dtype=torch.float32, device=x.device)
                ws = c.workspace()
                H.call("dgcnn_edge_mlp_bf16_bwd", *bsrc, mean.data_ptr(), rstd.data_ptr(), beta0.data_ptr(), mx.data_ptr(), H.ld2(mx),
                       cnt.data_ptr(), dmx.data_ptr(), H.ld2(dmx), dmn.data_ptr(), H.ld2(dmn), red.data_ptr(), H._p(dYb), H._p(dysum),
                       F, c.var_grads[w0name].data_ptr(), c.var_grads[b0name].data_ptr(), 1.0, ws.data_ptr(), ws.numel(),
                       tag="edge_mlp_bf16_kernel<bwd>", work=2.0 * R * k * 2 * C * F * 2, nbytes=4.0 * (R * k * C + R * C))
                if dx is not None:
                    # dx_i += (sum_m dY) (W0'[:C] - W0'[C:])^T,  dx_j += (sum of the incoming dY rows) W0'[C:]^T,  W0' = bf16(W0)
                    W0r = torch.empty_like(W0)
                    H.call("dgcnn_round_bf16_f32", W0.data_ptr(), W0r.data_ptr(), W0.numel())
                    wdb = torch.empty((C, F), dtype=torch.float32, device=x.device)
                    H.call("dgcnn_copy2d_f32", W0r[:C].data_ptr(), F, wdb.data_ptr(), F, C, F, 0)
                    H.call("dgcnn_axpby_f32", W0r[C:].data_ptr(), -1.0, wdb.data_ptr(), 1.0, C * F)
                    gemm(dysum, wdb, dx, transB=True, beta=1.0)
                    if csr is not None:
                        off, rev = csr
                    else:
                        off, rev = build_csr(idx, B, N, k)
                    S = torch.empty((R, F), dtype=torch.float32, device=x.device)
                    H.call("dgcnn_edge_gather_sum_bf16", dYb.data_ptr(), off.data_ptr(), rev.data_ptr(), R, F, S.data_ptr(), F,
                           tag="csr_gather_sum_kernel<bf16>", work=2.0 * R * k * F + 4.0 * R * F)
                    gemm(S, W0r[C:], dx, transB=True, beta=1.0)
                return
            if fused_bf16:
                # the forward wrote neither E nor y: both are formed again for the backward (y by the SAME instruction sequence,
                # so the recomputed z compares equal to the maxima the forward took), used by the passes below and dropped
                Y = torch.empty((R * k, F), dtype=torch.float32, device=x.device)
                H.call("dgcnn_edge_mlp_bf16", *bsrc, Y.data_ptr(), tag="edge_mlp_bf16_kernel<write>",
                       work=2.0 * R * k * 2 * C * F, nbytes=4.0 * (R * k * C + R * C))
                Ee, W0p = bf16_operands()
            if virtual and EDGE_BWD_REDUCE_POINTS:
                # sum dz, sum dz*xhat from the forward's per-point outputs alone (bn.hip): no pass over the edges
                H.call("dgcnn_edge_bn_bwd_reduce_points_f32", mx.data_ptr(), H.ld2(mx), mn.data_ptr(), H.ld2(mn),
                       cnt.data_ptr(), dmx.data_ptr(), H.ld2(dmx), dmn.data_ptr(), H.ld2(dmn), beta0.data_ptr(), R, k, F,
                       red.data_ptr(), tag="edge_bwd_reduce_points_kernel", work=4.0 * 5 * R * F)
            elif virtual:
                assert UV is not None          # esrc holds raw pointers into UV: this reference keeps it alive
                H.call("dgcnn_edge_bn_bwd_reduce_f32", *esrc, mean.data_ptr(), rstd.data_ptr(),
                       beta0.data_ptr(), 1, dmx.data_ptr(), H.ld2(dmx), dmn.data_ptr(), H.ld2(dmn),
                       mx.data_ptr(), H.ld2(mx), cnt.data_ptr(), red.data_ptr(),
                       tag="bn_bwd_reduce_kernel<edge>", work=4.0 * (6 * R * F) + 4.0 * R * k)
            else:
                bn_bwd_reduce(Y, R, k, F, mean, rstd, beta0, 1, dmx, dmn, mx, cnt, red,
                              tag="bn_bwd_reduce_kernel", work=4.0 * (R * k * F + 2 * R * F))
            dx = c.grad(x)
            if fused_l0 and dx is None:
                # first layer (raw coordinates, nobody wants d(points)): dY is formed and consumed in one pass, straight into dW0
                ws = c.workspace()
                H.call("dgcnn_edge_bn_bwd_apply_wgrad_f32", *esrc, mean.data_ptr(), rstd.data_ptr(), beta0.data_ptr(),
                       dmx.data_ptr(), H.ld2(dmx), dmn.data_ptr(), H.ld2(dmn), mx.data_ptr(), H.ld2(mx), cnt.data_ptr(),
                       red.data_ptr(), x.data_ptr(), H.ld2(x), C, c.var_grads[w0name].data_ptr(), c.var_grads[b0name].data_ptr(), 1.0,
                       ws.data_ptr(), ws.numel(), tag="edge_bwd_apply_wgrad_kernel", work=4.0 * (4 * R * F) + 4.0 * R * k,
                       nbytes=4.0 * (R * k * F + R * F))
                return
            need_sum = (dx is not None) or not literal
            dUV = dysum = None
            if gather:
                dUV = torch.empty((R, 2 * F), dtype=torch.float32, device=x.device)    # [dU | dV]
                dysum = dUV[:, :F]
            elif need_sum:
                dysum = torch.empty((R, F), dtype=torch.float32, device=x.device)
            if virtual:
                assert UV is not None          # esrc holds raw pointers into UV: this reference keeps it alive
                dY = torch.empty((R * k, F), dtype=torch.float32, device=x.device)
                H.call("dgcnn_edge_bn_bwd_apply_f32", *esrc, mean.data_ptr(), rstd.data_ptr(), beta0.data_ptr(),
                       1, dmx.data_ptr(), H.ld2(dmx), dmn.data_ptr(), H.ld2(dmn), mx.data_ptr(), H.ld2(mx), cnt.data_ptr(),
                       red.data_ptr(), dY.data_ptr(),
                       H._p(dysum), 0 if dysum is None else H.ld2(dysum), c.var_grads[b0name].data_ptr(), 1.0,
                       tag="bn_bwd_apply_kernel<edge>", work=4.0 * (R * k * F + 7 * R * F) + 4.0 * R * k, nbytes=4.0 * (R * k * F + R * F))
            else:
                H.call("dgcnn_bn_bwd_apply_f32", Y.data_ptr(), R, k, F, mean.data_ptr(), rstd.data_ptr(), beta0.data_ptr(),
                       3 if bf16 else 1, dmx.data_ptr(), H.ld2(dmx), dmn.data_ptr(), H.ld2(dmn), mx.data_ptr(), H.ld2(mx), cnt.data_ptr(),
                       red.data_ptr(), Y.data_ptr(),
                       H._p(dysum), 0 if dysum is None else H.ld2(dysum), c.var_grads[b0name].data_ptr(), 1.0,
                       tag="bn_bwd_apply_kernel", work=4.0 * (2 * R * k * F + 3 * R * F))
                dY = Y
            ws = c.workspace()
            dW0 = c.var_grads[w0name]

            def incoming_sum(S):
                # tf.gather^T as a gather: bucket edges by target, then every point sums its incoming dY rows
                if csr is not None:
                    off, rev = csr                    # built on the side stream during the forward
                else:
                    off, rev = build_csr(idx, B, N, k)
                H.call("dgcnn_edge_gather_sum_f32", dY.data_ptr(), off.data_ptr(), rev.data_ptr(), R, F, S.data_ptr(),
                       H.ld2(S), tag="csr_gather_sum_kernel", work=4.0 * (R * k * F + R * F))

            if gather:
                # dU = sum_m dY, dV = sum of incoming dY;  dWcat = X^T [dU|dV],  dx += [dU|dV] Wcat^T
                incoming_sum(dUV[:, F:])
                dwcat = torch.empty((Cp, 2 * F), dtype=torch.float32, device=x.device)
                # dUV / dwcat are locals of this closure allocated on the main stream: record them on the side stream,
                # or the caching allocator may hand dUV's block to the next main-stream allocation while the side GEMM reads it
                if dx is not None:
                    gemm(dUV, wcat[:C], dx, transB=True, beta=1.0, arith=None)
                with c.off_critical_path(dwcat, dUV, rows=R):
                    gemm(xg, dUV, dwcat, transA=True, arith=None)
                    H.call("dgcnn_edge_wgrad_combine_f32", dwcat.data_ptr(), C, F, dW0.data_ptr())
                return
            if bf16:
                # dW0 = E^T dY with bf16 operands: E and W0 rounded as in the forward; dY was written on the bf16 grid by the
                # apply pass (relu flag 3), so the product's own rounding (arith = 1) leaves it unchanged
                if W0p is W0:
                    gemm(Ee, dY, dW0, transA=True, beta=1.0, arith=1)
                else:
                    dWp = torch.empty_like(W0p)
                    gemm(Ee, dY, dWp, transA=True, arith=1)
                    H.call("dgcnn_copy2d_f32", dWp[:C].data_ptr(), F, dW0[:C].data_ptr(), F, C, F, 1)
                    H.call("dgcnn_copy2d_f32", dWp[Cp:Cp + C].data_ptr(), F, dW0[C:].data_ptr(), F, C, F, 1)
                if dx is not None:
                    # dE = dY W0'^T (W0' = bf16(W0)) never exists: the transpose of `edges` is linear, so
                    #   dx_i += (sum_m dY) (W0'[:C] - W0'[C:])^T        dx_j += (sum of the dY rows of the edges that point at j) W0'[C:]^T
                    # -- the same sums of the same bf16 x bf16 products as the literal edge-level product + scatter-add, taken at
                    # the points (k times fewer MACs, no atomics); the point-level products run in the exact arithmetic
                    W0r = torch.empty_like(W0)
                    H.call("dgcnn_round_bf16_f32", W0.data_ptr(), W0r.data_ptr(), W0.numel())
                    wdb = torch.empty((C, F), dtype=torch.float32, device=x.device)
                    H.call("dgcnn_copy2d_f32", W0r[:C].data_ptr(), F, wdb.data_ptr(), F, C, F, 0)
                    H.call("dgcnn_axpby_f32", W0r[C:].data_ptr(), -1.0, wdb.data_ptr(), 1.0, C * F)
                    gemm(dysum, wdb, dx, transB=True, beta=1.0)
                    if F % 4 != 0:
                        raise H.HipError("bf16 edge-MLP: the input gradient needs filter counts that are multiples of 4 (F = %d)" % F)
                    S = torch.empty((R, F), dtype=torch.float32, device=x.device)
                    incoming_sum(S)
                    gemm(S, W0r[C:], dx, transB=True, beta=1.0)
                if fused_bf16:
                    Y = Ee = None                   # (recomputed for this backward only)
                return
            if literal:
                H.call("dgcnn_edge_mlp_wgrad_f32", x.data_ptr(), H.ld2(x), idx.data_ptr(), dY.data_ptr(), B, N, C, k, F,
                       dW0.data_ptr(), 1.0, ws.data_ptr(), ws.numel(),
                       tag=("edge_wgrad_smallc_kernel" if C <= 4 else "gemm_kernel<A_EDGE_T,B_ROW,STORE,128,%d>" % (64 if F <= 64 else 128)),
                       work=2.0 * R * k * 2 * C * F)
            else:
                # dW0[C:] += sum_e x_j^T dY  -  X^T dYsum ;  dW0[:C] += X^T dYsum   (W0[:C]-W0[C:] and W0[C:] are the
                # factors of the forward: d(Wa-Wb) = X^T dYsum, dWb = edge part)
                H.call("dgcnn_edge_nbr_wgrad_f32", x.data_ptr(), H.ld2(x), idx.data_ptr(), dY.data_ptr(), B, N, C, k, F,
                       dW0[C:].data_ptr(), 1.0, ws.data_ptr(), ws.numel(),
                       tag=("edge_wgrad_smallc_kernel" if C <= 4 else "gemm_kernel<A_EDGE_T,B_ROW,STORE,128,%d>" % (64 if F <= 64 else 128)),
                       work=2.0 * R * k * C * F)
                dwd = torch.empty((C, F), dtype=torch.float32, device=x.device)
                gemm(x, dysum, dwd, transA=True)
                H.call("dgcnn_axpby_f32", dwd.data_ptr(), 1.0, dW0[:C].data_ptr(), 1.0, C * F)
                H.call("dgcnn_axpby_f32", dwd.data_ptr(), -1.0, dW0[C:].data_ptr(), 1.0, C * F)
            if dx is not None:
                # E = [x_i, x_j - x_i]  =>  dx_i += (sum_m dY) (W0[:C]-W0[C:])^T ; dx_j += dY W0[C:]^T
                wdb = wd
                if wdb is None:
                    wdb = torch.empty((C, F), dtype=torch.float32, device=x.device)
                    H.call("dgcnn_copy2d_f32", W0[:C].data_ptr(), F, wdb.data_ptr(), F, C, F, 0)
                    H.call("dgcnn_axpby_f32", W0[C:].data_ptr(), -1.0, wdb.data_ptr(), 1.0, C * F)
                gemm(dysum, wdb, dx, transB=True, beta=1.0)
                if SCATTER_ATOMICS or F % 4 != 0:
                    H.call("dgcnn_edge_mlp_dgrad_scatter_f32", dY.data_ptr(), W0.data_ptr(), idx.data_ptr(), B, N, C, k,
                           F, dx.data_ptr(), H.ld2(dx),
                           tag="gemm_kernel<A_ROW,B_COL,SCATTER,%d,%d>" % (_tile_m(R * k, C), 64 if C <= 64 else 128),
                           work=2.0 * R * k * C * F)
                else:
                    S = torch.empty((R, F), dtype=torch.float32, device=x.device)
                    incoming_sum(S)
                    gemm(S, W0[C:], dx, transB=True, beta=1.0)
        c.tape.append(bwd)

    net = conv_bn_act(mm, "conv1", 64, relu=relu1, out=net_out, out2=net2, arith=None)   # ops.py:62-70 (64 hard-coded)
    return mm, net, idx


# ----------------------------------------------------------------------------------------------
# residual add (ops.py:134), dropout (model.py:91), global max-pool (model.py:76-81)
# ----------------------------------------------------------------------------------------------
def add_relu(a, b, out=None):
    c = ctx()
    R, F = a.shape
    if tuple(b.shape) != (R, F):
        raise ValueError("residual shortcut shape mismatch %s vs %s (ops.py:134)" % (tuple(a.shape), tuple(b.shape)))
    if out is None:
        out = c.new_buffer(R, F)
    H.call("dgcnn_add_relu_f32", a.data_ptr(), H.ld2(a), b.data_ptr(), H.ld2(b), R, F, out.data_ptr(), H.ld2(out))
    if c.recording:
        def bwd():
            dout = c.grad(out)
            if dout is None:
                return
            d = torch.empty((R, F), dtype=torch.float32, device=a.device)
            H.call("dgcnn_relu_bwd_f32", dout.data_ptr(), H.ld2(dout), out.data_ptr(), H.ld2(out), R, F, d.data_ptr(), F)
            for t in (a, b):
                g = c.grad(t)
                if g is not None:
                    H.call("dgcnn_copy2d_f32", d.data_ptr(), F, g.data_ptr(), H.ld2(g), R, F, 1)
        c.tape.append(bwd)
    return out


def dropout(x, keep=DROPOUT_KEEP):
    c = ctx()
    R, F = x.shape
    assert x.is_contiguous()
    out = c.new_buffer(R, F)
    if c.seed_dev is None:
        c.advance_seed()
    seed = c.seed_dev                       # device-resident: the backward regenerates the mask from the same value
    H.call("dgcnn_dropout_dev_f32", x.data_ptr(), out.data_ptr(), R * F, float(keep), seed.data_ptr())
    if c.recording:
        def bwd():
            dout = c.grad(out)
            if dout is None:
                return
            dx, bx = c.grad_w(x)
            if dx is None:
                return
            if bx == 0.0:                                   # first touch of d(x): write the masked gradient in place
                H.call("dgcnn_dropout_dev_f32", dout.data_ptr(), dx.data_ptr(), R * F, float(keep), seed.data_ptr())
                return
            tmp = torch.empty_like(dout)
            H.call("dgcnn_dropout_dev_f32", dout.data_ptr(), tmp.data_ptr(), R * F, float(keep), seed.data_ptr())
            H.call("dgcnn_copy2d_f32", tmp.data_ptr(), F, dx.data_ptr(), H.ld2(dx), R, F, 1)
        c.tape.append(bwd)
    return out


def tile_rows(g, out, rows):
    """tf.tile of a per-cloud (B,F) tensor over the `rows` points of its cloud into the (B*rows, F) view `out` (model.py:80-81);
    backward: tf.tile^T = the sum over the cloud."""
    c = ctx()
    G, F = g.shape
    H.call("dgcnn_tile_rows_f32", g.data_ptr(), H.ld2(g), G, int(rows), F, out.data_ptr(), H.ld2(out))
    if c.recording:
        def bwd():
            dout, dg_ = c.grad(out), c.grad(g)
            if dout is None or dg_ is None:
                return
            tmp = torch.empty((G, F), dtype=torch.float32, device=g.device)
            H.call("dgcnn_group_colsum_f32", dout.data_ptr(), H.ld2(dout), G, int(rows), F, tmp.data_ptr())
            H.call("dgcnn_copy2d_f32", tmp.data_ptr(), F, dg_.data_ptr(), H.ld2(dg_), G, F, 1)
        c.tape.append(bwd)
    return out


def global_max(x, B, N):
    """(B*N,F) -> (B,F) max over the points of each cloud, first arg-max remembered for the backward."""
    c = ctx()
    F = x.shape[1]
    out = c.new_buffer(B, F)
    arg = torch.empty((B, F), dtype=torch.int32, device=x.device)
    H.call("dgcnn_global_max_f32", x.data_ptr(), H.ld2(x), B, N, F, out.data_ptr(), arg.data_ptr())
    if c.recording:
        def bwd():
            dout = c.grad(out)
            dx = c.grad(x)
            if dout is None or dx is None:
                return
            H.call("dgcnn_global_max_bwd_f32", dout.data_ptr(), arg.data_ptr(), B, N, F, dx.data_ptr(), H.ld2(dx))
        c.tape.append(bwd)
    return out


def plain_gemm(a, leaf_scope_weight, w_rows, Cout):
    """out = a @ W[lo:hi]  (the per-cloud part of FC0: global feature x its 1024 weight rows)."""
    c = ctx()
    wname, W = leaf_scope_weight
    Wx = W[w_rows[0]:w_rows[1]]
    out = c.new_buffer(a.shape[0], Cout)
    gemm(a, Wx, out)
    if c.recording:
        def bwd():
            dout = c.grad(out)
            if dout is None:
                return
            gemm(a, dout, c.var_grads[wname][w_rows[0]:w_rows[1]], transA=True, beta=1.0)
            da = c.grad(a)
            if da is not None:
                gemm(dout, Wx, da, transB=True, beta=1.0)
        c.tape.append(bwd)
    return out


# ----------------------------------------------------------------------------------------------
# dgcnn/trainval.py:39-52 softmax / accuracy / loss (+ the seed of the backward pass)
# ----------------------------------------------------------------------------------------------
def mark_head_gradients_complete():
    """Tape marker placed (in forward order) right after the EdgeConv stack: in the backward it runs when every closure of
    the head (MergedEdgeConv, FC*, Final) has run, i.e. when the head's share of the gradient bucket is final."""
    c = ctx()
    if not c.recording:
        return

    def bwd():
        hook, c.head_grads_hook = c.head_grads_hook, None
        if hook is not None:
            hook()
    c.tape.append(bwd)


def softmax_loss(logits2d, labels, weight, want_grad):
    """-> (softmax (R,ncls), scal tensor [loss, accuracy] on device).  Seeds d(logits) if want_grad."""
    c = ctx()
    R, ncls = logits2d.shape
    assert logits2d.is_contiguous()
    sm = torch.empty((R, ncls), dtype=torch.float32, device=logits2d.device)
    scal = H.zeros(2, torch.float32, logits2d.device)      # (its own tensor: the caller reads it after later towers' begin_step)
    dl = None
    if want_grad:
        dl, _ = c.grad_w(logits2d)          # (written, not accumulated: a first touch of a whole root needs no zero fill)
        assert dl is not None and dl.is_contiguous()
    H.call("dgcnn_softmax_xent_f32", logits2d.data_ptr(), H._p(labels), H._p(weight), R, ncls, sm.data_ptr(),
           H._p(dl), scal.data_ptr())
    return sm, scal
