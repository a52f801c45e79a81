"""ctypes binding of libdgcnn_hip.so (the C ABI declared in include/dgcnn_hip.h).

PyTorch-ROCm tensors are only memory holders here: every wrapper passes `tensor.data_ptr()` and
the current HIP stream to the C entry point.  There is NO fallback: if the shared library is
missing, or a call is made without a visible GPU, this module raises -- the product path never
routes through the oracle or through torch math.
"""
from __future__ import annotations

import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DGCNN_HIP_LIB") or os.path.join(_HERE, "libdgcnn_hip.so")   # env override: experiments
STAT_SLOTS = 32

c_int, c_i64, c_f32, c_f64, c_vp, c_sz, c_u64 = (ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double,
                                                  ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint64)

# name -> argtypes, exactly as include/dgcnn_hip.h declares them (restype is int unless noted)
PROTOTYPES = {
    "dgcnn_version": [],
    "dgcnn_last_error": [],
    "dgcnn_knn_workspace_bytes": [c_int, c_int, c_int, c_int],
    "dgcnn_knn_grid": [c_int],
    "dgcnn_knn_force_valu": [c_int],
    "dgcnn_knn_bf16_filter": [c_int],
    "dgcnn_knn_f32": [c_vp, c_int, c_int, c_int, c_i64, c_int, c_vp, c_vp, c_sz, c_vp],
    "dgcnn_knn_seed_min_n": [c_int],
    "dgcnn_knn_append": [c_int],
    "dgcnn_knn_append_products": [c_int],
    "dgcnn_knn_hist": [c_int],
    "dgcnn_knn_seeded_f32": [c_vp, c_int, c_int, c_int, c_i64, c_int, c_vp, c_i64, c_int, c_vp, c_vp, c_sz, c_vp],
    "dgcnn_edge_gather_f32": [c_vp, c_i64, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp],
    "dgcnn_edge_gather_bwd_f32": [c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp, c_i64, c_vp],
    "dgcnn_edge_mlp_f32": [c_vp, c_i64, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp],
    "dgcnn_edge_mlp_wgrad_f32": [c_vp, c_i64, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_f32,
                                 c_vp, c_sz, c_vp],
    "dgcnn_edge_mlp_dgrad_scatter_f32": [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_i64, c_vp],
    "dgcnn_edge_nbr_gemm_f32": [c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp],
    "dgcnn_edge_nbr_wgrad_f32": [c_vp, c_i64, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_f32,
                                 c_vp, c_sz, c_vp],
    "dgcnn_edge_csr_build": [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp],
    "dgcnn_edge_gather_sum_f32": [c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_i64, c_vp],
    "dgcnn_edge_weight_split_f32": [c_vp, c_int, c_int, c_vp, c_vp],
    "dgcnn_edge_gather_add_f32": [c_vp, c_i64, c_vp, c_i64, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp],
    "dgcnn_edge_wgrad_combine_f32": [c_vp, c_int, c_int, c_vp, c_vp],
    "dgcnn_edge_bn_act_kreduce_f32": [c_vp, c_i64, c_vp, c_i64, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp,
                                      c_int, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp],
    "dgcnn_edge_bn_bwd_reduce_f32": [c_vp, c_i64, c_vp, c_i64, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp,
                                     c_int, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp],
    "dgcnn_edge_bn_bwd_reduce_points_f32": [c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_int,
                                            c_int, c_vp, c_vp],
    "dgcnn_edge_bn_bwd_apply_f32": [c_vp, c_i64, c_vp, c_i64, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp,
                                    c_int, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp,
                                    c_f32, c_vp],
    "dgcnn_edge_bn_bwd_apply_wgrad_f32": [c_vp, c_i64, c_vp, c_i64, c_vp, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp,
                                          c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_f32,
                                          c_vp, c_sz, c_vp],
    "dgcnn_bn1_act_dropout_f32": [c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_int, c_f32, c_vp, c_vp, c_i64, c_vp],
    "dgcnn_bn1_bwd_dropout_f32": [c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_int, c_f32, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_f32, c_vp],
    "dgcnn_set_stat_slots": [c_int],
    "dgcnn_get_stat_slots": [],
    "dgcnn_gemm_set_arith": [c_int],
    "dgcnn_gemm_get_arith": [],
    "dgcnn_gemm_stat_writers": [c_int, c_int, c_int, c_int, c_vp, c_i64, c_vp, c_i64],
    "dgcnn_gemm_x3_tile_rows": [c_int, c_int, c_int],
    "dgcnn_gemm_x3_tile_cols": [c_int, c_int, c_int],
    "dgcnn_gemm_x3_tile_override": [c_int],
    "dgcnn_gemm_f32": [c_int, c_int, c_int, c_int, c_int, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_f32,
                       c_vp, c_i64, c_int, c_vp, c_vp, c_int, c_vp, c_sz, c_vp],
    "dgcnn_colmax_decode_f32": [c_vp, c_i64, c_vp, c_vp, c_vp],
    "dgcnn_split_planes_f32": [c_vp, c_i64, c_i64, c_i64, c_int, c_int, c_vp, c_vp, c_i64, c_i64, c_vp],
    "dgcnn_planes_scale_f32": [c_vp, c_i64, c_i64, c_int, c_f32, c_vp, c_vp, c_vp],
    "dgcnn_gemm_planes_f32": [c_int, c_int, c_int, c_int, c_int, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp, c_i64, c_f32,
                              c_vp, c_i64, c_int, c_vp, c_vp, c_int, c_vp, c_sz, c_vp],
    "dgcnn_param_scales_f32": [c_vp, c_i64, c_f64, c_f32, c_vp, c_vp, c_vp],
    "dgcnn_bn_act_planes_f32": [c_vp, c_i64, c_i64, c_int, c_vp, c_vp, c_vp, c_int, c_int, c_vp, c_vp, c_i64, c_i64, c_vp, c_i64,
                                c_vp, c_i64, c_vp],
    "dgcnn_bn1_bwd_reduce_max_f32": [c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_int, c_vp, c_i64, c_vp, c_vp, c_vp],
    "dgcnn_bn1_bwd_apply_planes_f32": [c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_int, c_vp, c_i64, c_vp, c_vp, c_int, c_vp, c_vp,
                                       c_i64, c_i64, c_vp, c_vp, c_i64, c_int, c_vp, c_f32, c_vp],
    "dgcnn_bn_finalize_f32": [c_vp, c_int, c_f64, c_f32, c_vp, c_vp, c_vp],
    "dgcnn_bn_act_kreduce_f32": [c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_int, c_vp, c_i64, c_vp, c_i64,
                                 c_vp, c_i64, c_vp, c_vp],
    "dgcnn_bn_bwd_reduce_f32": [c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_int, c_vp, c_i64, c_vp, c_i64,
                                c_vp, c_i64, c_vp, c_vp, c_vp],
    "dgcnn_bn_bwd_apply_f32": [c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_int, c_vp, c_i64, c_vp, c_i64,
                               c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_f32, c_vp],
    "dgcnn_det_workspace_bytes": [c_int],
    "dgcnn_colstats_det_f32": [c_vp, c_i64, c_int, c_i64, c_vp, c_vp, c_sz, c_vp],
    "dgcnn_bn_bwd_reduce_det_f32": [c_vp, c_i64, c_int, c_int, c_vp, c_vp, c_vp, c_int, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64,
                                    c_vp, c_vp, c_vp, c_sz, c_vp],
    "dgcnn_edge_csr_sort": [c_vp, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp],
    "dgcnn_global_max_f32": [c_vp, c_i64, c_int, c_int, c_int, c_vp, c_vp, c_vp],
    "dgcnn_global_max_bwd_f32": [c_vp, c_vp, c_int, c_int, c_int, c_vp, c_i64, c_vp],
    "dgcnn_group_colsum_f32": [c_vp, c_i64, c_int, c_int, c_int, c_vp, c_vp],
    "dgcnn_tile_rows_f32": [c_vp, c_i64, c_int, c_int, c_int, c_vp, c_i64, c_vp],
    "dgcnn_dropout_f32": [c_vp, c_vp, c_i64, c_f32, c_u64, c_vp],
    "dgcnn_dropout_dev_f32": [c_vp, c_vp, c_i64, c_f32, c_vp, c_vp],
    "dgcnn_add_relu_f32": [c_vp, c_i64, c_vp, c_i64, c_i64, c_int, c_vp, c_i64, c_vp],
    "dgcnn_relu_bwd_f32": [c_vp, c_i64, c_vp, c_i64, c_i64, c_int, c_vp, c_i64, c_vp],
    "dgcnn_pad_copy_f32": [c_vp, c_i64, c_int, c_vp, c_int, c_i64, c_vp],
    "dgcnn_copy2d_f32": [c_vp, c_i64, c_vp, c_i64, c_i64, c_int, c_int, c_vp],
    "dgcnn_softmax_xent_f32": [c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_vp, c_vp, c_vp],
    "dgcnn_axpby_f32": [c_vp, c_f32, c_vp, c_f32, c_i64, c_vp],
    "dgcnn_adam_f32": [c_vp, c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32, c_vp],
    "dgcnn_edge_mlp_bf16_supported": [c_int, c_int, c_int],
    "dgcnn_round_bf16_f32": [c_vp, c_vp, c_i64, c_vp],
    "dgcnn_edge_mlp_bf16": [c_vp, c_i64, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp],
    "dgcnn_edge_mlp_bf16_stats": [c_vp, c_i64, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp],
    "dgcnn_edge_mlp_bf16_bn_kreduce": [c_vp, c_i64, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_i64,
                                       c_vp, c_i64, c_vp, c_int, c_vp],
    "dgcnn_edge_mlp_bf16_bwd_supported": [c_int, c_int, c_int],
    "dgcnn_edge_mlp_bf16_bwd_workspace_bytes": [c_int, c_int, c_int, c_int, c_int],
    "dgcnn_edge_mlp_bf16_bwd": [c_vp, c_i64, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp,
                                c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_f32, c_vp, ctypes.c_size_t, c_vp],
    "dgcnn_edge_gather_sum_bf16": [c_vp, c_vp, c_vp, c_i64, c_int, c_vp, c_i64, c_vp],
    "dgcnn_plan_begin": [],
    "dgcnn_plan_end": [c_vp],
    "dgcnn_plan_abort": [],
    "dgcnn_plan_replay": [c_vp],
    "dgcnn_plan_info": [c_vp, c_vp, c_vp, c_vp, c_vp],
    "dgcnn_plan_destroy": [c_vp],
    "dgcnn_stream_wait": [c_vp, c_vp],
    "dgcnn_stream_create": [c_int, c_int, c_vp],
    "dgcnn_stream_destroy": [c_vp],
    "dgcnn_stream_priority_range": [c_vp, c_vp],
    "dgcnn_memset_async": [c_vp, c_int, c_sz, c_vp],
    "dgcnn_comm_available": [],
    "dgcnn_comm_unique_id": [c_vp],
    "dgcnn_comm_init": [c_int, c_int, c_vp, c_vp],
    "dgcnn_comm_destroy": [c_vp],
    "dgcnn_comm_info": [c_vp, c_vp, c_vp, c_vp],
    "dgcnn_allreduce_f32": [c_vp, c_i64, c_vp, c_vp],
    "dgcnn_broadcast_f32": [c_vp, c_i64, c_int, c_vp, c_vp],
    "dgcnn_comm_counters": [c_vp, c_vp],
}

INT64_RESULTS = ("dgcnn_knn_workspace_bytes", "dgcnn_edge_mlp_bf16_bwd_workspace_bytes")      # byte counts; every other entry point returns an int status

_lib = None


class HipError(RuntimeError):
    pass


def load():
    """Load the library (no GPU needed for loading / symbol checks)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipError("libdgcnn_hip.so not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "or `make -C dynamic-gcnn_amd/csrc` -- there is no CPU fallback" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, args in PROTOTYPES.items():
            fn = getattr(lib, name)
            fn.argtypes = args
            fn.restype = (ctypes.c_char_p if name == "dgcnn_last_error" else
                          ctypes.c_int64 if name in INT64_RESULTS else ctypes.c_int)
        _lib = lib
    return _lib


def set_gemm_arith(mode):
    """0 = native fp32 MFMA, 6 / 9 = partial products of the exact 3-way bf16 split, 1 = bf16 operands (include/dgcnn_hip.h)."""
    lib = load()
    if lib.dgcnn_gemm_set_arith(int(mode)) != 0:
        raise ValueError(lib.dgcnn_last_error().decode())


def set_stat_slots(n):
    """Slots of every stats / red buffer from now on (include/dgcnn_hip.h: dgcnn_set_stat_slots); mirrored in STAT_SLOTS."""
    global STAT_SLOTS
    lib = load()
    if lib.dgcnn_set_stat_slots(int(n)) != 0:
        raise ValueError(lib.dgcnn_last_error().decode())
    STAT_SLOTS = int(n)


def gemm_arith():
    lib = load()
    return int(lib.dgcnn_gemm_get_arith())


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return 0 if t is None else t.data_ptr()


class Timer(object):
    """Optional per-launch HIP-event timing (bench.py): events are recorded on the stream the
    kernel is launched on (torch's current stream)."""

    def __init__(self, watch=None):
        self.watch = watch          # None = every tagged call, else a set of tags
        self.records = []           # (tag, start_event, stop_event, work, bytes)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for tag, a, b, work, nbytes in self.records:
            d = out.setdefault(tag, [0, 0.0, 0.0, 0.0])
            d[0] += 1
            d[1] += a.elapsed_time(b) * 1e-3
            d[2] += work
            d[3] += nbytes
        return out                  # tag -> [launches, seconds, work, operand + result bytes (GEMMs; 0 where not given)]


TIMER = None


def call(name, *args, tag=None, work=0.0, nbytes=0.0):
    """Invoke an entry point on the current stream; raise ValueError/HipError on a negative code.
    tag/work: kernel identity and algorithmic flops-or-bytes of this launch, for bench.py's roofline; nbytes: the compulsory
    operand + result bytes of a GEMM launch (its `work` is flops), which decide whether HBM or the matrix pipe bounds it."""
    lib = load()
    tm = TIMER
    if tm is not None and tag is not None and (tm.watch is None or tag in tm.watch):
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        rc = getattr(lib, name)(*args, _stream())
        b.record()
        tm.records.append((tag, a, b, work, nbytes))
    else:
        rc = getattr(lib, name)(*args, _stream())
    if rc != 0:
        msg = lib.dgcnn_last_error().decode()
        if rc == -1:
            raise ValueError(msg)
        raise HipError("%s failed (%d): %s" % (name, rc, msg))


def _check(rc, name):
    if rc != 0:
        msg = load().dgcnn_last_error().decode()
        if rc == -1:
            raise ValueError(msg)
        raise HipError("%s failed (%d): %s" % (name, rc, msg))


def memset(t, value=0):
    """hipMemsetAsync of the whole (contiguous) tensor on the current stream -- through the library, so that a launch plan being
    recorded sees it (a torch .zero_() would be missing from the replay)."""
    assert t.is_contiguous()
    n = t.numel() * t.element_size()
    if n:
        _check(load().dgcnn_memset_async(t.data_ptr(), int(value), n, _stream()), "dgcnn_memset_async")
    return t


def zeros(shape, dtype, device):
    return memset(torch.empty(shape, dtype=dtype, device=device))


def make_stream(device, low_priority=False, reserve_cus=0):
    """A torch view (ExternalStream) of a HIP stream the library creates: at the device's LEAST priority and / or on a CU mask that
    keeps `reserve_cus` compute units free for the other streams (dgcnn_stream_create).  Plain torch streams only go from normal
    priority upwards and know no CU masks.  The stream lives as long as the process (a handful per process)."""
    if not low_priority and reserve_cus <= 0:
        return torch.cuda.Stream(device=device)
    h = ctypes.c_void_p()
    with torch.cuda.device(device):
        _check(load().dgcnn_stream_create(int(bool(low_priority)), int(reserve_cus), ctypes.byref(h)), "dgcnn_stream_create")
    return torch.cuda.ExternalStream(h.value, device=device)


def stream_wait(waiter, signaller):
    """`waiter` (a torch stream) waits for everything issued so far on `signaller`; recorded by a launch plan under construction."""
    _check(load().dgcnn_stream_wait(waiter.cuda_stream, signaller.cuda_stream), "dgcnn_stream_wait")


class Plan(object):
    """A recorded launch list (csrc/plan.cc).  `with Plan.record() as p:` runs the enclosed step AND records it; p.replay()
    re-issues it.  The caller keeps the step's device addresses alive and unchanged (trainval records inside a private memory pool)."""

    def __init__(self):
        self.handle = None

    class _Rec(object):
        def __init__(self, plan):
            self.plan = plan

        def __enter__(self):
            _check(load().dgcnn_plan_begin(), "dgcnn_plan_begin")
            return self.plan

        def __exit__(self, et, ev, tb):
            if et is not None:
                load().dgcnn_plan_abort()
                return False
            h = ctypes.c_void_p()
            _check(load().dgcnn_plan_end(ctypes.byref(h)), "dgcnn_plan_end")
            self.plan.handle = h
            return False

    @classmethod
    def record(cls):
        return cls._Rec(cls())

    def replay(self):
        _check(load().dgcnn_plan_replay(self.handle), "dgcnn_plan_replay")

    def info(self):
        v = [ctypes.c_int(0) for _ in range(4)]
        _check(load().dgcnn_plan_info(self.handle, *[ctypes.byref(x) for x in v]), "dgcnn_plan_info")
        return dict(zip(("kernels", "memsets", "waits", "collectives"), [int(x.value) for x in v]))

    def destroy(self):
        if self.handle is not None:
            load().dgcnn_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


def comm_counters():
    """(all-reduce calls, elements) issued through the library so far -- direct calls and replayed launch plans alike."""
    a, b = ctypes.c_int64(0), ctypes.c_int64(0)
    _check(load().dgcnn_comm_counters(ctypes.byref(a), ctypes.byref(b)), "dgcnn_comm_counters")
    return int(a.value), int(b.value)


def require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise HipError("dgcnn ops need device (ROCm) tensors; got a %s tensor -- no CPU fallback" % t.device)


def f32(t):
    if t.dtype != torch.float32:
        raise TypeError("expected float32, got %s" % t.dtype)
    return t


def ld2(t):
    """Leading dimension (row stride, in elements) of a 2-D row-major view."""
    assert t.dim() == 2 and (t.stride(1) == 1 or t.shape[1] == 1), (t.shape, t.stride())
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])
