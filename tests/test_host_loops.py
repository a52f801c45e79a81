"""Host-side logic around the hot path that needs no GPU: batch sources (reference iotool.py),
flags / CLI (flags.py) and the pure helpers of main_funcs.py."""
import os

import numpy as np
import pytest

import dgcnn
from dgcnn import main_funcs as M


def _flags(**kw):
    base = dict(IO_TYPE="synthetic", BATCH_SIZE=4, MINIBATCH_SIZE=2, NUM_POINT=32, NUM_ENTRIES=10, SHUFFLE=0, SEED=3)
    base.update(kw)
    return dgcnn.DGCNN_FLAGS(**base)


def test_csv_columns_are_the_reference_ones():
    # main_funcs.py:119-122 and :213-216 write exactly these headers
    assert M.TRAIN_COLUMNS == ("iter,epoch,titer,ttrain,tio,tsave,tsummary,tsumiter,tsumtrain,tsumio,tsumsave,"
                               "tsumsummary,loss,accuracy")
    assert M.INFERENCE_COLUMNS == "iter,epoch,titer,tinference,tio,tsumiter,tsuminference,tsumio,loss,accuracy"


def test_pure_helpers():
    assert M.iteration_from_filename("./weights/snapshot-499") == 499
    assert M.iteration_from_filename("a-b/c-d-12.npz") == 12
    assert M.round_decimals(0.123456, 4) == 0.1235
    assert M.round_decimals(2.5, 0) == 3.0


def test_sequential_cursor_wraps_like_the_reference():
    io = dgcnn.io_factory(_flags())
    io.initialize()
    assert io.num_entries() == 10 and io.num_channels() == 3 and io.batch_size() == 4
    got = [io.next()[0].tolist() for _ in range(4)]
    assert got == [[0, 1, 2, 3], [4, 5, 6, 7], [8, 9, 0, 1], [2, 3, 4, 5]]          # iotool.py:269-275
    idx, data, label, weight = io.next()
    assert data.shape == (4, 32, 3) and data.dtype == np.float32
    assert label.shape == (4, 32) and label.dtype == np.int32 and set(np.unique(label)) <= {0, 1}
    assert weight is None


def test_shuffle_draws_distinct_entries_and_is_seeded():
    a, b = dgcnn.io_factory(_flags(SHUFFLE=1)), dgcnn.io_factory(_flags(SHUFFLE=1))
    a.initialize(), b.initialize()
    for _ in range(5):
        ia, ib = a.next()[0], b.next()[0]
        assert len(set(ia.tolist())) == 4 and ia.tolist() == ib.tolist()
    c = dgcnn.io_factory(_flags(SHUFFLE=1, SEED=4))
    c.initialize()
    assert any(c.next()[0].tolist() != a.next()[0].tolist() for _ in range(4))


def test_npz_source_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    for i in range(2):
        np.savez(tmp_path / ("in%d.npz" % i), data=rng.random((3, 16, 4), dtype=np.float32),
                 label=rng.integers(0, 3, (3, 16)), w=rng.random((3, 16), dtype=np.float32))
    f = _flags(IO_TYPE="npz", INPUT_FILE="%s,%s" % (tmp_path / "in0.npz", tmp_path / "in1.npz"), WEIGHT_KEY="w",
               OUTPUT_FILE=str(tmp_path / "out.npz"), BATCH_SIZE=4)
    io = dgcnn.io_factory(f)
    io.initialize()
    assert io.num_entries() == 6 and io.num_channels() == 4
    idx, data, label, weight = io.next()
    assert data.shape == (4, 16, 4) and label.shape == (4, 16) and weight.shape == (4, 16)
    first = np.load(tmp_path / "in0.npz")
    assert np.array_equal(data[:3], first["data"]) and np.array_equal(weight[:3], first["w"])
    sm = rng.random((16, 3), dtype=np.float32)
    io.store(idx[1], sm)
    with pytest.raises(ValueError):
        io.store(6, sm)
    io.finalize()
    out = np.load(tmp_path / "out.npz")
    assert out["idx"].tolist() == [1] and np.array_equal(out["softmax"][0], sm)
    assert np.array_equal(out["data"][0], first["data"][1]) and np.array_equal(out["label"][0], first["label"][1])


class _FakeH5File(dict):
    """Just enough of h5py.File (context manager, item access, create_dataset) to drive io_h5 here,
    where h5py is not installed."""
    disk = {}

    def __init__(self, path, mode):
        super().__init__(self.disk.get(path, {}) if mode == "r" else {})
        self._path, self._mode = path, mode

    def __enter__(self):
        return self

    def __exit__(self, *a):
        if self._mode == "w":
            self.disk[self._path] = dict(self)

    def create_dataset(self, name, data, **kw):
        self[name] = np.asarray(data)


def test_h5_source_with_a_stand_in_module(monkeypatch):
    import sys
    import types
    if "h5py" not in sys.modules:
        try:
            import h5py  # noqa: F401
        except ImportError:                      # no h5py: the factory hands out the plain-Python reader / writer (tests/test_h5min.py)
            assert dgcnn.io_factory(_flags(IO_TYPE="h5"))._h5.__name__.endswith("_h5min")
    monkeypatch.setitem(sys.modules, "h5py", types.SimpleNamespace(File=_FakeH5File))
    rng = np.random.default_rng(1)
    _FakeH5File.disk["a.h5"] = {"d": rng.random((5, 8, 4), dtype=np.float32), "l": rng.integers(0, 2, (5, 8))}
    io = dgcnn.io_factory(_flags(IO_TYPE="h5", INPUT_FILE="a.h5", DATA_KEY="d", LABEL_KEY="l", OUTPUT_FILE="o.h5"))
    io.initialize()
    assert (io.num_entries(), io.num_channels()) == (5, 4)
    idx, data, label, weight = io.next()
    assert idx.tolist() == [0, 1, 2, 3] and np.array_equal(data, _FakeH5File.disk["a.h5"]["d"][:4]) and weight is None
    io.store(2, np.ones((8, 2), np.float32))
    io.finalize()
    assert set(_FakeH5File.disk["o.h5"]) == {"idx", "d", "softmax", "l"}       # iotool.py:238-245: data/softmax/label


def test_npz_source_rejects_bad_shapes(tmp_path):
    np.savez(tmp_path / "bad.npz", data=np.zeros((3, 16), np.float32), label=np.zeros((3, 16), np.int32))
    io = dgcnn.io_factory(_flags(IO_TYPE="npz", INPUT_FILE=str(tmp_path / "bad.npz")))
    with pytest.raises(ValueError):
        io.initialize()
    np.savez(tmp_path / "bad2.npz", data=np.zeros((3, 16, 3), np.float32), label=np.zeros((3, 15), np.int32))
    io = dgcnn.io_factory(_flags(IO_TYPE="npz", INPUT_FILE=str(tmp_path / "bad2.npz")))
    with pytest.raises(ValueError):
        io.initialize()


def test_unshipped_readers_say_so():
    for kind in ("larcv", "nope"):
        with pytest.raises(NotImplementedError):
            dgcnn.io_factory(_flags(IO_TYPE=kind))
    io = dgcnn.iotool.io_base(_flags())
    for call in (io.initialize, lambda: io.store(0, None)):
        with pytest.raises(NotImplementedError):
            call()


def test_cli_has_the_reference_options(capsys):
    f = dgcnn.DGCNN_FLAGS()
    script = f.parse_args(["train", "-kv", "10", "-ecl", "2", "-ecf", "32,48", "-fcl", "1", "-fcf", "128", "-nc", "5",
                           "-np", "256", "-it", "7", "-bs", "8", "-mbs", "4", "-rs", "1", "-mn", "residual-dgcnn",
                           "-io", "npz", "-if", "a.npz,b.npz", "-dkey", "d", "-lkey", "l", "-wkey", "w", "-sd", "11",
                           "-wp", "w/s", "-lr", "0.01", "-ss", "2", "-chks", "3", "-chkn", "1", "-chkh", "1.0",
                           "--gpus", "0,1", "-sh", "0", "-db", "0", "-ld", "logs", "-mp", "w/s-3", "-of", "o.npz"],
                          run=False)
    capsys.readouterr()
    assert script == "train"
    assert (f.KVALUE, f.EDGE_CONV_LAYERS, f.EDGE_CONV_FILTERS, f.FC_LAYERS, f.FC_FILTERS) == (10, 2, [32, 48], 1, 128)
    assert (f.NUM_CLASS, f.NUM_POINT, f.ITERATION, f.BATCH_SIZE, f.MINIBATCH_SIZE, f.REPORT_STEP) == (5, 256, 7, 8, 4, 1)
    assert (f.MODEL_NAME, f.IO_TYPE, f.INPUT_FILE, f.DATA_KEY, f.LABEL_KEY, f.WEIGHT_KEY) == \
        ("residual-dgcnn", "npz", ["a.npz", "b.npz"], "d", "l", "w")
    assert (f.SEED, f.WEIGHT_PREFIX, f.LEARNING_RATE, f.SUMMARY_STEP, f.CHECKPOINT_STEP, f.CHECKPOINT_NUM) == \
        (11, "w/s", 0.01, 2, 3, 1)
    assert (f.GPUS, f.SHUFFLE, f.DEBUG, f.LOG_DIR, f.MODEL_PATH, f.OUTPUT_FILE) == ([0, 1], False, False, "logs", "w/s-3", "o.npz")
    # inference has no training-only options (flags.py:127-139)
    with pytest.raises(SystemExit):
        dgcnn.DGCNN_FLAGS().parse_args(["inference", "-lr", "0.1"], run=False)
    capsys.readouterr()
    g = dgcnn.DGCNN_FLAGS()
    assert g.parse_args(["iotest", "-wkey", "w"], run=False) == "iotest" and g.SEED == 1


def test_micro_batches_cover_the_replica_share():
    f = _flags(BATCH_SIZE=8, MINIBATCH_SIZE=2, GPUS="0,1")
    data = np.arange(8)[:, None, None] * np.ones((1, 4, 3), np.float32)
    label = np.arange(8)[:, None] * np.ones((1, 4), np.int32)
    h = M.Handlers()
    h.rank, h.world = 1, 2                                  # second of two replicas: clouds 4..7
    steps = list(M._micro_batches(f, h, data, label, None))
    assert len(steps) == 1
    dv, lv, wv = steps[0]
    assert wv is None and [d[:, 0, 0].tolist() for d in dv] == [[4.0, 5.0], [6.0, 7.0]]
    assert [l[:, 0].tolist() for l in lv] == [[4, 5], [6, 7]]
    h.rank, h.world = 0, 1
    assert [[d[:, 0, 0].tolist() for d in s[0]] for s in M._micro_batches(f, h, data, label, None)] == \
        [[[0.0, 1.0], [2.0, 3.0]], [[4.0, 5.0], [6.0, 7.0]]]


def test_prepare_rejects_indivisible_batch(capsys):
    with pytest.raises(SystemExit):
        M.prepare(_flags(BATCH_SIZE=5, MINIBATCH_SIZE=2))
    assert "must be a multiple" in capsys.readouterr().err


CONDA_PY = "/opt/conda/bin/python3.9"      # the interpreter of this image that has h5py


@pytest.mark.skipif(not os.path.exists(CONDA_PY), reason="no interpreter with h5py in this image")
def test_real_hdf5_roundtrip_through_io_h5(tmp_path):
    """N2: io_h5 against REAL HDF5 files (dense layout of iotool.py:212-231 and the ragged layout), written and re-read
    with h5py by tests/h5_roundtrip.py under the interpreter that has h5py."""
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    p = subprocess.run([CONDA_PY, os.path.join(here, "h5_roundtrip.py"), str(tmp_path)], stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, timeout=300, env={"PATH": os.environ.get("PATH", "")})
    out = p.stdout.decode(errors="replace")
    if p.returncode != 0 and "No module named 'h5py'" in out:
        pytest.skip("h5py missing under %s" % CONDA_PY)
    assert p.returncode == 0 and "H5_ROUNDTRIP_OK" in out, out[-2000:]


def _ragged_file(path, counts, C=4, seed=0, weight=False):
    rng = np.random.default_rng(seed)
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    arrays = dict(data=rng.random((off[-1], C), dtype=np.float32), label=rng.integers(0, 2, off[-1]).astype(np.int32),
                  data_offsets=off)
    if weight:
        arrays["w"] = rng.random(off[-1], dtype=np.float32)
    np.savez(path, **arrays)
    return arrays


def test_ragged_npz_source(tmp_path):
    """N4: clouds with different point counts (what io_larcv yields, iotool.py:76-98): lists out of next(), the
    <256-point cut of iotool.py:81, sequential wrap-around, seeded contiguous shuffle window, CSR output."""
    a = _ragged_file(tmp_path / "r.npz", [300, 255, 700, 256, 1000], weight=True)
    off = a["data_offsets"]
    f = _flags(IO_TYPE="npz", INPUT_FILE=str(tmp_path / "r.npz"), WEIGHT_KEY="w", BATCH_SIZE=3, OUTPUT_FILE=str(tmp_path / "o.npz"))
    io = dgcnn.io_factory(f)
    io.initialize()
    assert io.num_entries() == 4 and io.num_channels() == 4           # 255 points: dropped
    idx, data, label, weight = io.next()
    assert idx.tolist() == [0, 1, 2] and [d.shape for d in data] == [(300, 4), (700, 4), (256, 4)]
    assert np.array_equal(data[1], a["data"][off[2]:off[3]]) and np.array_equal(label[2], a["label"][off[3]:off[4]])
    assert np.array_equal(weight[0], a["w"][:300])
    assert io.next()[0].tolist() == [3, 0, 1]
    io.store(1, np.ones((700, 2), np.float32))
    io.finalize()
    o = np.load(tmp_path / "o.npz")
    assert o["data_offsets"].tolist() == [0, 700] and o["softmax"].shape == (700, 2) and o["label"].shape == (700,)
    g = _flags(IO_TYPE="npz", INPUT_FILE=str(tmp_path / "r.npz"), BATCH_SIZE=2, SHUFFLE=1, MIN_POINTS=1)
    io2 = dgcnn.io_factory(g)
    io2.initialize()
    assert io2.num_entries() == 5
    for _ in range(6):
        i2 = io2.next()[0]
        assert i2[1] == i2[0] + 1 and 0 <= i2[0] <= 3                  # io_larcv.next: a contiguous window
    # dense and ragged files do not mix; malformed offsets are refused
    np.savez(tmp_path / "d.npz", data=np.zeros((2, 300, 4), np.float32), label=np.zeros((2, 300), np.int32))
    with pytest.raises(ValueError):
        dgcnn.io_factory(_flags(IO_TYPE="npz", INPUT_FILE="%s,%s" % (tmp_path / "r.npz", tmp_path / "d.npz"))).initialize()
    np.savez(tmp_path / "bad.npz", data=np.zeros((10, 4), np.float32), label=np.zeros(10, np.int32), data_offsets=np.array([0, 4, 9]))
    with pytest.raises(ValueError):
        dgcnn.io_factory(_flags(IO_TYPE="npz", INPUT_FILE=str(tmp_path / "bad.npz"), MIN_POINTS=1)).initialize()


def test_micro_batches_of_a_ragged_batch():
    """main_funcs.py:139-155 with `-mbs 1`: every micro-step is ONE cloud as a (1, N_i, C) tower array; a micro-batch
    that mixes point counts is refused with the reason."""
    h = M.Handlers()
    h.rank, h.world = 0, 1
    rng = np.random.default_rng(0)
    data = [rng.random((n, 4), dtype=np.float32) for n in (300, 512, 300, 700)]
    label = [np.zeros(len(d), np.int32) for d in data]
    f = _flags(BATCH_SIZE=4, MINIBATCH_SIZE=1)
    steps = list(M._micro_batches(f, h, data, label, None))
    assert [s[0][0].shape for s in steps] == [(1, 300, 4), (1, 512, 4), (1, 300, 4), (1, 700, 4)]
    assert all(s[1][0].shape == (1, s[0][0].shape[1]) and s[2] is None for s in steps)
    with pytest.raises(ValueError, match="minibatch_size 1"):
        list(M._micro_batches(_flags(BATCH_SIZE=4, MINIBATCH_SIZE=2), h, data, label, None))
    same = list(M._micro_batches(_flags(BATCH_SIZE=2, MINIBATCH_SIZE=2), h, [data[0], data[2]], [label[0], label[2]], None))
    assert same[0][0][0].shape == (2, 300, 4)                            # equal N stacks


def test_saver_keeps_a_checkpoint_every_n_hours(tmp_path):
    """tf.train.Saver(max_to_keep, keep_checkpoint_every_n_hours) as main_funcs.py:82-84 builds it: the newest CHECKPOINT_NUM
    stay; one that leaves that window survives when CHECKPOINT_HOUR hours have passed since the last survivor."""
    from dgcnn import main_funcs as M

    class FakeTrainer(object):
        _rank = 0

        def save(self, prefix, step):
            name = "%s-%d" % (prefix, step)
            open(name + ".npz", "w").close()
            return name

    now = [0.0]
    sv = M.Saver(max_to_keep=2, keep_checkpoint_every_n_hours=1.0, clock=lambda: now[0])
    prefix = str(tmp_path / "snap")
    for step in range(8):                       # one checkpoint every 30 minutes
        now[0] = step * 1800.0
        sv.save(FakeTrainer(), prefix, step)
    left = sorted(int(f.name.split("-")[-1][:-4]) for f in tmp_path.iterdir() if f.name.endswith(".npz"))
    # window {6, 7} + one per hour: TF keeps an evicted checkpoint when its time is strictly PAST the deadline (`>`), and the
    # deadline advances by whole periods (`+=`): deadlines 3600, 7200, 10800 -> the checkpoints written at 5400 and 9000
    assert left == [3, 5, 6, 7], left
    assert (tmp_path / "checkpoint").read_text().strip().endswith('snap-7"')
    sv0 = M.Saver(max_to_keep=2, keep_checkpoint_every_n_hours=0.0, clock=lambda: now[0])
    prefix = str(tmp_path / "plain")
    for step in range(5):
        now[0] += 7200.0
        sv0.save(FakeTrainer(), prefix, step)
    assert sorted(f.name for f in tmp_path.iterdir() if f.name.startswith("plain")) == ["plain-3.npz", "plain-4.npz"]
