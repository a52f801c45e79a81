"""The exact cell-grid k-NN of the raw-coordinate layer (csrc/knn_grid.hip; C <= 4, k <= 40) against the all-pairs kernel and the
C oracle (dgcnn/ops.py:8-19; oracle/knn_oracle.c): the same indices, bit for bit, on inputs chosen to break a spatial search --
duplicates, exact ties, empty cells, clouds far from the origin, degenerate axes, k == N, ragged N."""
import numpy as np
import pytest
import torch

from oracle import dgcnn_oracle as O
from gpu_helpers import dev, host

pytestmark = pytest.mark.gpu


@pytest.fixture()
def dg():
    import dgcnn
    dgcnn.reset()
    return dgcnn


def both(dg, pts, k, mode=2):
    """(grid result, all-pairs result) of ops.k_nn on the same cloud."""
    from dgcnn import _hip as H
    lib = H.load()
    x = dev(pts)
    prev = lib.dgcnn_knn_grid(mode)
    try:
        a = host(dg.ops.k_nn(x, k))
        lib.dgcnn_knn_grid(0)
        b = host(dg.ops.k_nn(x, k))
    finally:
        lib.dgcnn_knn_grid(prev)
    return a, b


def clouds(rng):
    yield "uniform", rng.random((3, 1000, 3), dtype=np.float32), 20
    yield "uniform, N % 64 != 0, k = 8", rng.random((2, 777, 3), dtype=np.float32), 8
    yield "k = 40", rng.random((2, 2048, 3), dtype=np.float32), 40
    yield "integer lattice (exact ties, duplicates)", rng.integers(0, 12, (2, 1500, 3)).astype(np.float32), 20
    yield "all points identical", np.full((1, 300, 3), 0.25, np.float32), 20
    yield "two values only", rng.integers(0, 2, (2, 400, 3)).astype(np.float32), 20
    line = np.zeros((2, 900, 3), np.float32)
    line[..., 0] = rng.random((2, 900))
    yield "points on a line (two flat axes)", line, 20
    plane = rng.random((1, 1200, 3), dtype=np.float32)
    plane[..., 2] = 0.5
    yield "points in a plane", plane, 20
    far = rng.random((2, 1024, 3), dtype=np.float32) + np.float32(1000.0)
    yield "cloud far from the origin (margins swallow the bound)", far, 20
    cl = rng.normal(0, 0.01, (2, 1500, 3)).astype(np.float32)
    cl[:, :20] += 50.0
    cl[:, 20:40] -= 30.0
    yield "tight cluster + distant outliers", cl, 20
    tr = np.cumsum(rng.normal(0, 0.02, (2, 4096, 3)), axis=1).astype(np.float32)
    yield "random-walk tracks (LArTPC-like, very uneven density)", tr, 20
    yield "C = 4", rng.random((2, 1100, 4), dtype=np.float32), 20
    c4 = rng.random((2, 900, 4), dtype=np.float32)
    c4[..., 3] *= 100.0
    yield "C = 4, the 4th channel dominates the distance", c4, 20
    yield "C = 2", rng.random((2, 640, 2), dtype=np.float32), 20
    yield "C = 1", rng.random((2, 500, 1), dtype=np.float32), 8
    yield "k == N", rng.random((3, 20, 3), dtype=np.float32), 20
    yield "N = 5, k = 3", rng.random((4, 5, 3), dtype=np.float32), 3
    yield "tiny values (denormal squares)", (rng.random((1, 600, 3)) * 1e-20).astype(np.float32), 20
    yield "huge values", (rng.random((1, 600, 3)) * 1e15).astype(np.float32), 20


def test_grid_search_equals_all_pairs_and_the_oracle(dg):
    rng = np.random.default_rng(2024)
    for what, pts, k in clouds(rng):
        a, b = both(dg, pts, k)
        ref = O.k_nn(pts, k)
        bad = np.argwhere((a != ref).any(-1))
        assert len(bad) == 0, "%s: grid differs from the oracle in %d rows, first %s: %s vs %s" % (
            what, len(bad), bad[0], a[tuple(bad[0])], ref[tuple(bad[0])])
        np.testing.assert_array_equal(b, ref, err_msg=what + " (all-pairs kernel)")


def test_grid_search_at_the_headline_shape(dg):
    """(24, 2048, 3, 20) with the grid forced on against the C oracle, every row of four clouds; skewed clouds too."""
    from dgcnn import _hip as H
    prev = H.load().dgcnn_knn_grid(2)
    rng = np.random.default_rng(11)
    pts = rng.random((4, 2048, 3), dtype=np.float32)
    pts[1] = np.cumsum(rng.normal(0, 0.02, (2048, 3)), axis=0).astype(np.float32)
    pts[2, :, 0] *= 100.0                                       # one long axis
    pts[3] = rng.integers(0, 16, (2048, 3)).astype(np.float32)
    try:
        np.testing.assert_array_equal(host(dg.ops.k_nn(dev(pts), 20)), O.k_nn(pts, 20))
    finally:
        H.load().dgcnn_knn_grid(prev)


@pytest.mark.parametrize("N,k,C", [(16384, 40, 3), (65536, 20, 3), (65536, 20, 4)])
def test_grid_search_at_the_large_baseline_sizes(dg, N, k, C):
    """configs[2] / configs[4] point counts on the raw-coordinate layer: a seeded sample of rows against the C oracle (the grid
    looks at ~22 k candidates per row instead of N)."""
    rng = np.random.default_rng(N + C)
    pts = rng.random((2, N, C), dtype=np.float32)
    idx = host(dg.ops.k_nn(dev(pts), k))
    rows = np.sort(rng.permutation(N)[:2048]).astype(np.int32)
    for b in range(2):
        np.testing.assert_array_equal(idx[b][rows], O.k_nn_rows(pts[b], k, rows))


def test_workspace_too_small_for_the_grid_selects_the_all_pairs_kernel(dg):
    from dgcnn import _hip as H
    lib = H.load()
    B, N, C, k = 2, 300, 3, 10
    rng = np.random.default_rng(1)
    pts = rng.random((B, N, C), dtype=np.float32)
    x = dev(pts)
    full = int(lib.dgcnn_knn_workspace_bytes(B, N, C, k))
    small = (B * N * 4 + 255) // 256 * 256
    # [s_i | seed bounds | grid scratch];  feature-space rows: [s_i | seed bounds | counts | 256 candidates per row] (append-form scan)
    assert full > 2 * small and int(lib.dgcnn_knn_workspace_bytes(B, N, 64, k)) == 3 * small + B * N * 256 * 8
    assert int(lib.dgcnn_knn_workspace_bytes(B, N, 128, k)) == 2 * small
    idx = torch.empty((B, N, k), dtype=torch.int32, device="cuda")
    ws = torch.empty(small, dtype=torch.uint8, device="cuda")
    H.call("dgcnn_knn_f32", x.data_ptr(), B, N, C, C, k, idx.data_ptr(), ws.data_ptr(), small)
    np.testing.assert_array_equal(host(idx), O.k_nn(pts, k))
    with pytest.raises(ValueError):          # DGCNN_EINVAL: not even room for the s_i
        H.call("dgcnn_knn_f32", x.data_ptr(), B, N, C, C, k, idx.data_ptr(), ws.data_ptr(), small - 256)


# ------------------------------------------------------------------------------------------------------
# seeded search (dgcnn_knn_seeded_f32): any k distinct candidates per row bound the row's k-th distance from above
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("append", [1, 3, 0], ids=["append-scan", "append-scan-3-products", "lists"])
@pytest.mark.parametrize("B,N,C,k", [(2, 512, 64, 20), (1, 700, 32, 8), (2, 2048, 64, 20), (1, 9000, 64, 40), (1, 333, 48, 64),
                                     (1, 40, 64, 20), (3, 64, 32, 10), (2, 130, 64, 8), (1, 8200, 20, 20), (2, 65, 64, 64)])
def test_seeded_knn_equals_the_unseeded_search_whatever_the_seeds(B, N, C, k, append):
    """The result must not depend on the seeds: true neighbours (a tight bound), random distinct candidates (a loose one), the row's
    own index repeated (not distinct: the row gets no bound), out-of-range indices (no bound), more seeds than k (the first k count).
    All against the oracle, bit for bit -- with the append-form scan (rows without a bound fill and compact their candidate buffers
    over and over: the slow path of that kernel) and with the list-keeping kernels."""
    from dgcnn import _engine as E, _hip as H
    prev = H.load().dgcnn_knn_seed_min_n(0)                                       # (lists: the library seeds from N = 4096 on by default)
    prev_a = H.load().dgcnn_knn_append(1 if append else 0)
    prev_p = H.load().dgcnn_knn_append_products(3 if append == 3 else 1)       # the filter's products: 1 (default) / 3
    try:
        _seeded_cases(E, B, N, C, k)
    finally:
        H.load().dgcnn_knn_seed_min_n(prev)
        H.load().dgcnn_knn_append(prev_a)
        H.load().dgcnn_knn_append_products(prev_p)


def _seeded_cases(E, B, N, C, k):
    rng = np.random.default_rng(N + C)
    x = np.maximum(rng.normal(size=(B, N, C)), 0).astype(np.float32)             # ReLU features: many zeros, near-ties
    x[:, 5] = x[:, 3]                                                             # duplicate rows: exact distance ties
    ref = O.k_nn(x, k)
    xd = dev(x.reshape(B * N, C))
    plain = host(E.knn(xd, B, N, k))
    np.testing.assert_array_equal(plain, ref)
    seeds = {
        "true neighbours": ref,
        "true neighbours + extra columns": np.concatenate([ref, (ref[..., :3] + 1) % N], -1),
        "random distinct": np.stack([np.stack([rng.permutation(N)[:k] for _ in range(N)]) for _ in range(B)]),
        "not distinct": np.broadcast_to(np.arange(N)[None, :, None], (B, N, k)).copy(),
        "out of range": np.full((B, N, k), N + 7),
    }
    if N >= 2048:
        del seeds["random distinct"]                                              # (host-side generation is slow; covered at small N)
    for name, sd in seeds.items():
        got = host(E.knn(xd, B, N, k, seed=dev(np.ascontiguousarray(sd.astype(np.int32)))))
        np.testing.assert_array_equal(got, ref, err_msg=name)
    # mixed: half of the rows seeded well, half badly, in one launch
    mix = ref.copy()
    mix[:, ::2] = N + 1
    np.testing.assert_array_equal(host(E.knn(xd, B, N, k, seed=dev(mix.astype(np.int32)))), ref)


# ------------------------------------------------------------------------------------------------------
# raw-coordinate rows below the cell grid's range: the histogram bound (knn_hist_bound_kernel, round 6)
# ------------------------------------------------------------------------------------------------------
def _raw_cloud(kind, rng, B, N, C):
    if kind == "uniform":
        return rng.random((B, N, C), dtype=np.float32)
    if kind == "lattice":                       # integer coordinates: many exact ties and duplicate points
        return rng.integers(0, 5, (B, N, C)).astype(np.float32)
    if kind == "far":                           # far from the origin: s_i ~ 3e6, distances ~ 1e-2 (cancellation; everything in the lowest bins)
        return (rng.random((B, N, C)) + 1000.0).astype(np.float32)
    if kind == "tiny":                          # a cloud of extent 1e-4
        return (rng.random((B, N, C)) * 1e-4).astype(np.float32)
    if kind == "huge":
        return (rng.random((B, N, C)) * 3e4).astype(np.float32)
    if kind == "track":                         # points along a line + noise
        t = rng.random((B, N, 1))
        return (t * np.array([1.0, 0.5, 0.25, 0.1][:C]) + rng.normal(0, 1e-3, (B, N, C))).astype(np.float32)
    if kind == "same":                          # every point identical: all distances 0, ties decided by index
        return np.broadcast_to(rng.random((B, 1, C), dtype=np.float32), (B, N, C)).copy()
    if kind == "clusters":                      # two tight clusters far apart + one outlier per cloud
        c = np.where(rng.random((B, N, 1)) < 0.5, 0.0, 50.0)
        x = (c + rng.normal(0, 1e-2, (B, N, C))).astype(np.float32)
        x[:, 0] = 1e3
        return x
    raise ValueError(kind)


@pytest.mark.parametrize("stride", [1, 2, 4])
@pytest.mark.parametrize("kind", ["uniform", "lattice", "far", "tiny", "huge", "track", "same", "clusters"])
def test_histogram_bound_never_changes_the_raw_coordinate_graph(dg, kind, stride):
    """The bound is rigorous for any input: with every sample stride, on clouds that stress the bins (scales from 1e-4 to 3e4,
    cancellation far from the origin, lattices with hundreds of equal distances, a cloud of identical points, clusters with an
    outlier that sets the cloud's largest distance), at ragged N and every list size, the graph equals the C oracle's -- and the
    search without the bound."""
    from dgcnn import _hip as H
    lib = H.load()
    rng = np.random.default_rng(sum(map(ord, kind)) + stride)
    prev = lib.dgcnn_knn_hist(stride)
    try:
        for (B, N, C, k) in [(2, 2048, 3, 20), (1, 1000, 3, 40), (2, 300, 4, 8), (1, 257, 3, 64), (1, 3000, 2, 20), (1, 600, 1, 5)]:
            pts = _raw_cloud(kind, rng, B, N, C)
            ref = O.k_nn(pts, k)
            got = host(dg.ops.k_nn(dev(pts), k))
            np.testing.assert_array_equal(got, ref, err_msg="%s stride %d %s" % (kind, stride, (B, N, C, k)))
        assert lib.dgcnn_knn_hist(0) == stride
        pts = _raw_cloud(kind, rng, 2, 1500, 3)
        np.testing.assert_array_equal(host(dg.ops.k_nn(dev(pts), 20)), O.k_nn(pts, 20))          # the search without the bound
    finally:
        lib.dgcnn_knn_hist(prev)
