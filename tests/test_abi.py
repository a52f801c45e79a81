"""CPU checks of the drop-in boundary: libdgcnn_hip.so loads without a GPU and exports exactly the
symbols include/dgcnn_hip.h declares, and the ctypes prototypes mirror the header's arity."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "dgcnn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|int64_t|const char\*)\s+(dgcnn_\w+)\s*\(([^;]*?)\)\s*;", text, flags=re.S):
        args = m.group(2).strip()
        out[m.group(1)] = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
    return out


def test_library_exports_every_declared_symbol():
    from dgcnn import _hip
    lib = ctypes.CDLL(_hip.LIB_PATH)
    decl = header_functions()
    assert len(decl) >= 24
    for name in decl:
        assert hasattr(lib, name), "missing export %s" % name
    assert lib.dgcnn_version() >= 100


def test_ctypes_prototypes_match_header():
    from dgcnn import _hip
    decl = header_functions()
    assert set(decl) == set(_hip.PROTOTYPES), set(decl) ^ set(_hip.PROTOTYPES)
    for name, n in decl.items():
        protos = _hip.PROTOTYPES[name]
        extra = 0 if name in ("dgcnn_version", "dgcnn_last_error", "dgcnn_knn_workspace_bytes") else 0
        assert len(protos) == n + extra, (name, len(protos), n)


def test_ops_fail_loudly_without_gpu_or_library():
    import torch
    import dgcnn
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(Exception):
        dgcnn.ops.k_nn(torch.zeros(1, 8, 3), 2)           # CPU tensor: no fallback path exists
    with pytest.raises(Exception):
        dgcnn.trainval(dgcnn.DGCNN_FLAGS()).initialize()    # no device


def test_host_logic_errors_without_gpu():
    import dgcnn
    with pytest.raises(ValueError):
        dgcnn.ops._listify([1, 2], 3, "k")                  # ops.py:80-82
    with pytest.raises(NotImplementedError):
        dgcnn.trainval(dgcnn.DGCNN_FLAGS(MODEL_NAME="nope")).initialize()   # model.py:41-43
    f = dgcnn.DGCNN_FLAGS(edge_conv_filters="64,64,128", fc_filters="512,256", gpus="0,1")
    assert f.EDGE_CONV_FILTERS == [64, 64, 128] and f.FC_FILTERS == [512, 256] and f.GPUS == [0, 1]   # flags.py:154-164
    from dgcnn.trainval import param_specs
    specs = param_specs(f, 3)
    assert sum(int(__import__("numpy").prod(s)) for _, s in specs) == 1797186


def test_stat_slots_setter_and_gemm_tile_query_work_without_a_gpu():
    """Process-wide modes behind the C ABI are plain host state: the slot count of the statistics buffers (deterministic mode:
    one writer per slot) and the tile the bf16-split GEMM picks for a shape (tools name kernel instances with it)."""
    from dgcnn import _hip as H
    lib = H.load()
    assert lib.dgcnn_get_stat_slots() == H.STAT_SLOTS
    try:
        H.set_stat_slots(768)
        assert lib.dgcnn_get_stat_slots() == 768 and H.STAT_SLOTS == 768
        with pytest.raises(ValueError):
            H.set_stat_slots(8)                                   # fewer than DGCNN_STAT_SLOTS: buffers of the plane kernels assume 32
        assert lib.dgcnn_get_stat_slots() == 768
    finally:
        H.set_stat_slots(32)
    assert lib.dgcnn_gemm_x3_tile_rows(49152, 512, 1728) == 256     # FC0 forward: the wave-specialised 256 x 128 kernel
    assert lib.dgcnn_gemm_x3_tile_rows(49152, 64, 128) == 64        # conv1: short reduction over many rows
    assert lib.dgcnn_gemm_x3_tile_rows(1024, 512, 192) == 128       # configs[0]-sized problems


def test_knn_workspace_sizes_and_switches_are_plain_host_state():
    """dgcnn_knn_workspace_bytes: [s_i | seed bounds] for every shape; + the cell grid's scratch for raw coordinates; + one count and
    256 ... 512 eight-byte candidate entries per row for the shapes the append-form scan takes (16 < C <= 64, C % 4 == 0, k <= 64),
    the capacity following k and the form (tile-by-tile re-check below N = 8192, carried queue above)."""
    from dgcnn import _hip as H
    lib = H.load()
    pad = lambda n: (n + 255) // 256 * 256
    for B, N in ((24, 2048), (2, 300), (8, 16384)):
        sq = pad(4 * B * N)
        assert lib.dgcnn_knn_workspace_bytes(B, N, 128, 20) == 2 * sq                      # C > 64: lists only
        assert lib.dgcnn_knn_workspace_bytes(B, N, 16, 20) == 2 * sq                       # C <= 16: lists only
        assert lib.dgcnn_knn_workspace_bytes(B, N, 3, 20) > 2 * sq                         # raw coordinates: + cell grid
        for k, cap_lx, cap_cq in ((8, 256, 320), (20, 256, 320), (40, 384, 448), (64, 512, 512)):
            cap = cap_lx if N < 8192 else cap_cq
            assert cap >= k + (64 if N < 8192 else 192)
            for C in (32, 64):
                assert lib.dgcnn_knn_workspace_bytes(B, N, C, k) == 2 * sq + pad(4 * B * N) + 8 * cap * B * N, (B, N, C, k)
    prev = lib.dgcnn_knn_append(0)
    try:
        assert lib.dgcnn_knn_append(-1) == 0 and lib.dgcnn_knn_append(1) == 0 and lib.dgcnn_knn_append(-1) == 1
    finally:
        lib.dgcnn_knn_append(prev)
    prev = lib.dgcnn_knn_append_products(0)                                                # query only
    try:
        assert prev in (1, 3)
        assert lib.dgcnn_knn_append_products(3) == prev and lib.dgcnn_knn_append_products(7) == 3 and lib.dgcnn_knn_append_products(1) == 3
    finally:
        lib.dgcnn_knn_append_products(prev)
    prev = lib.dgcnn_knn_seed_min_n(-2)                                                    # query only
    try:
        assert lib.dgcnn_knn_seed_min_n(4096) == prev and lib.dgcnn_knn_seed_min_n(-2) == 4096
        assert lib.dgcnn_knn_seed_min_n(-1) == 4096 and lib.dgcnn_knn_seed_min_n(-2) == -1  # -1: the library's rule
    finally:
        lib.dgcnn_knn_seed_min_n(prev)
