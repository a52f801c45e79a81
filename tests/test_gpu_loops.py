"""Run loops + checkpoints on the GPU (reference main_funcs.py:41-306): a short training run on the
synthetic source must write the reference's CSV, learn, checkpoint with retention, and resume
on the same trajectory; inference must reproduce the trainer's softmax and store it."""
import csv
import glob
import os

import numpy as np
import pytest
import torch

import dgcnn
from dgcnn import main_funcs as M

pytestmark = pytest.mark.gpu


def _flags(tmp, **kw):
    base = dict(IO_TYPE="synthetic", NUM_ENTRIES=16, NUM_POINT=256, NUM_CHANNEL=3, BATCH_SIZE=8, MINIBATCH_SIZE=4,
                KVALUE=8, EDGE_CONV_LAYERS=2, EDGE_CONV_FILTERS=[32, 64], FC_LAYERS=1, FC_FILTERS=[64], NUM_CLASS=2,
                ITERATION=6, REPORT_STEP=2, SUMMARY_STEP=2, CHECKPOINT_STEP=2, CHECKPOINT_NUM=2, SEED=5, SHUFFLE=1,
                LEARNING_RATE=1e-3, LOG_DIR=str(tmp / "log"), WEIGHT_PREFIX=str(tmp / "w" / "snap"))
    base.update(kw)
    return dgcnn.DGCNN_FLAGS(**base)


def _same_trajectory(a, b):
    """Training is not bit-reproducible run to run (BatchNorm statistics are summed with fp64 atomics,
    and Adam turns a last-bit flip of a ~0 gradient into a +-lr step), so two runs of the same
    schedule agree closely but not exactly; a wrong restore (lost Adam slots, wrong step count, wrong
    dropout position) is off by orders of magnitude more."""
    d = (a - b).abs()
    return float((d <= 1e-5).float().mean()) > 0.98 and float(d.max()) < 5e-3


def _rows(path):
    with open(path) as f:
        return list(csv.DictReader(f))


def test_train_loop_logs_checkpoints_and_resumes(tmp_path, capsys):
    f = _flags(tmp_path)
    h = M.train(f)
    out = capsys.readouterr().out
    assert out.count("Iteration ") == 3 and "train time fraction" in out
    rows = _rows(tmp_path / "log" / "train_log-0000000.csv")
    assert list(rows[0].keys()) == M.TRAIN_COLUMNS.split(",")
    assert [int(r["iter"]) for r in rows] == list(range(6))
    assert [float(r["epoch"]) for r in rows] == [i * 8 / 16 for i in range(6)]
    assert all(np.isfinite(float(r["loss"])) and 0.0 <= float(r["accuracy"]) <= 1.0 for r in rows)
    assert float(rows[-1]["tsumtrain"]) >= float(rows[0]["tsumtrain"]) > 0
    assert [int(r["iter"]) for r in _rows(tmp_path / "log" / "summary-0000000.csv")] == [1, 3, 5]
    # checkpoints at iterations 1,3,5; CHECKPOINT_NUM=2 keeps the newest two
    kept = sorted(os.path.basename(p) for p in glob.glob(str(tmp_path / "w" / "snap-*.npz")))
    assert kept == ["snap-3.npz", "snap-5.npz"]
    assert open(tmp_path / "w" / "checkpoint").read() == 'model_checkpoint_path: "snap-5"\n'
    assert M.latest_checkpoint(f.WEIGHT_PREFIX) == f.WEIGHT_PREFIX + "-5"
    z = np.load(tmp_path / "w" / "snap-5.npz")
    # TF names: outer scope dgcnn/ (trainval.py:29), slim's [1,1,Cin,Cout] weights, Adam slots, the (dead) moving statistics
    assert z["dgcnn/EdgeConv0/conv0/weights"].shape == (1, 1, 6, 32) and "dgcnn/Final/BatchNorm/beta/Adam_1" in z.files
    assert z["dgcnn/EdgeConv0/conv0/weights/Adam"].shape == (1, 1, 6, 32)
    assert float(np.abs(z["dgcnn/FC0/BatchNorm/moving_mean"]).max()) == 0 and float(z["dgcnn/FC0/BatchNorm/moving_variance"].min()) == 1
    assert int(z["adam_step"]) == 6 and np.isclose(float(z["beta1_power"]), 0.9 ** 7)
    final = h.trainer._ctx.flat_param.clone()

    # resume from iteration 3: the run continues at 4 (main_funcs.py:93) and logs to train_log-0000003.csv
    g = _flags(tmp_path, MODEL_PATH=f.WEIGHT_PREFIX + "-3", CHECKPOINT_STEP=0)
    h2 = M.prepare(g)
    assert h2.iteration == 4 and h2.trainer._ctx.adam_t == 4
    for _ in range(4):                       # put the data source where the first run was after 4 batches
        h2.data_io.next()
    M.train_loop(g, h2)
    capsys.readouterr()
    assert [int(r["iter"]) for r in _rows(tmp_path / "log" / "train_log-0000003.csv")] == [4, 5]
    assert _same_trajectory(h2.trainer._ctx.flat_param, final), "resumed run must retrace the original"
    fresh = dgcnn.trainval(_flags(tmp_path, LOG_DIR="")).initialize()._ctx.flat_param
    assert not _same_trajectory(fresh, final)


def test_training_learns_the_synthetic_labels(tmp_path, capsys):
    f = _flags(tmp_path, ITERATION=40, CHECKPOINT_STEP=0, SUMMARY_STEP=0, REPORT_STEP=0, LEARNING_RATE=3e-3)
    M.train(f)
    rows = _rows(tmp_path / "log" / "train_log-0000000.csv")
    first = np.mean([float(r["loss"]) for r in rows[:5]])
    last = np.mean([float(r["loss"]) for r in rows[-5:]])
    assert last < 0.8 * first, (first, last)
    assert np.mean([float(r["accuracy"]) for r in rows[-5:]]) > 0.6


def test_save_restore_roundtrip(tmp_path):
    f = _flags(tmp_path, LOG_DIR="")
    tv = dgcnn.trainval(f).initialize()
    rng = np.random.default_rng(0)
    pts = rng.random((4, 256, 3), dtype=np.float32)
    lab = rng.integers(0, 2, (4, 256)).astype(np.int32)
    for _ in range(2):
        tv.zero_gradients(None), tv.accum_gradient(None, [pts], [lab]), tv.apply_gradient(None)
    name = tv.save(str(tmp_path / "ck" / "s"), 1)
    assert name.endswith("s-1") and os.path.exists(name + ".npz")
    tv.zero_gradients(None), tv.accum_gradient(None, [pts], [lab]), tv.apply_gradient(None)
    want = tv._ctx.flat_param.clone()

    tv2 = dgcnn.trainval(_flags(tmp_path, LOG_DIR="", SEED=5)).initialize().restore(name)
    c = tv2._ctx
    # the round-1 checkpoint form (no dgcnn/ scope, 2-D weights) still loads to the same state
    with np.load(name + ".npz") as z:
        legacy = {(k[len("dgcnn/"):] if k.startswith("dgcnn/") else k): (z[k][0, 0] if z[k].ndim == 4 else z[k]) for k in z.files}
    tv3 = dgcnn.trainval(_flags(tmp_path, LOG_DIR="", SEED=6)).initialize().load_state_dict(legacy)
    assert torch.equal(tv3._ctx.flat_param, c.flat_param) and torch.equal(tv3._ctx.flat_v, c.flat_v)
    # (initialize() resets the process-wide engine context: rebuild tv2 before using it again)
    tv2 = dgcnn.trainval(_flags(tmp_path, LOG_DIR="", SEED=5)).initialize().restore(name)
    c = tv2._ctx
    assert c.adam_t == 2 and float(c.flat_m.abs().sum()) > 0 and float(c.flat_v.abs().sum()) > 0
    tv2.zero_gradients(None), tv2.accum_gradient(None, [pts], [lab]), tv2.apply_gradient(None)
    assert _same_trajectory(c.flat_param, want)
    stale = dgcnn.trainval(_flags(tmp_path, LOG_DIR="", SEED=5)).initialize().restore(name)
    stale._ctx.flat_m.zero_(), stale._ctx.flat_v.zero_()            # what a weights-only restore would do
    stale.zero_gradients(None), stale.accum_gradient(None, [pts], [lab]), stale.apply_gradient(None)
    assert not _same_trajectory(stale._ctx.flat_param, want)
    # a checkpoint of a different graph is refused
    other = dgcnn.trainval(_flags(tmp_path, LOG_DIR="", EDGE_CONV_FILTERS=[32, 48])).initialize()
    with pytest.raises(ValueError):
        other.restore(name)
    with pytest.raises(KeyError):
        dgcnn.trainval(_flags(tmp_path, LOG_DIR="", FC_LAYERS=2, FC_FILTERS=[64, 32])).initialize().restore(name)


def test_inference_loop_logs_and_stores_softmax(tmp_path, capsys):
    f = _flags(tmp_path, ITERATION=2, CHECKPOINT_STEP=2)
    M.train(f)
    data = M.io_factory(f)
    data.initialize()
    np.savez(tmp_path / "in.npz", data=data._data, label=data._label)
    g = _flags(tmp_path, IO_TYPE="npz", INPUT_FILE=str(tmp_path / "in.npz"), OUTPUT_FILE=str(tmp_path / "out.npz"),
               MODEL_PATH=f.WEIGHT_PREFIX + "-1", ITERATION=2, SHUFFLE=0, REPORT_STEP=1)
    h = M.inference(g)
    out = capsys.readouterr().out
    assert out.count("inference time fraction") == 2 and g.TRAIN is False
    rows = _rows(tmp_path / "log" / "inference_log-0000001.csv")
    assert list(rows[0].keys()) == M.INFERENCE_COLUMNS.split(",") and len(rows) == 2
    z = np.load(tmp_path / "out.npz")
    assert z["idx"].tolist() == list(range(16)) and z["softmax"].shape == (16, 256, 2)
    assert np.allclose(z["softmax"].sum(-1), 1.0, atol=1e-5)
    assert np.array_equal(z["data"], data._data) and np.array_equal(z["label"], data._label)
    # the stored rows are what the trainer's inference returns for the same micro-batch
    res = h.trainer.inference(None, [data._data[4:8]], [data._label[4:8]])
    assert np.allclose(res[0].cpu().numpy(), z["softmax"][4:8], atol=2e-6)      # fp64-atomic BN sums: last-bit noise
    acc = (z["softmax"].argmax(-1) == z["label"]).mean()
    assert abs(acc - np.mean([float(r["accuracy"]) for r in rows])) < 1e-6
    # without labels: loss/accuracy columns are -1 (main_funcs.py:275)
    k = _flags(tmp_path, IO_TYPE="npz", INPUT_FILE=str(tmp_path / "in.npz"), LABEL_KEY="", MODEL_PATH=f.WEIGHT_PREFIX + "-1",
               ITERATION=1, SHUFFLE=0, LOG_DIR=str(tmp_path / "log2"))
    M.inference(k)
    r = _rows(tmp_path / "log2" / "inference_log-0000001.csv")
    assert float(r[0]["loss"]) == -1 and float(r[0]["accuracy"]) == -1
    with pytest.raises(NotImplementedError):
        h.trainer.accum_gradient(None, [data._data[:4]], [data._label[:4]])


def test_train_and_inference_on_a_variable_n_source(tmp_path, capsys):
    """N4: the production regime `-np -1 -mbs 1` (scripts/lsf/train_dgcnn.sh:28) -- every cloud has its own point count and
    C = 4 channels (iotool.py:82), one cloud per micro-step, gradients summed over the BATCH_SIZE micro-steps
    (main_funcs.py:139-166).  A ragged .npz with 5 distinct N (one cloud below the 256-point cut of iotool.py:81) goes
    through train_loop, a checkpoint, and inference_loop, whose stored softmax rows must have each cloud's own length
    and reproduce the trainer's inference on that cloud."""
    rng = np.random.default_rng(4)
    counts = [300, 1000, 120, 517, 2048, 300, 777]
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    pts = rng.random((off[-1], 4), dtype=np.float32)
    lab = (pts[:, 0] > 0.5).astype(np.int32)
    np.savez(tmp_path / "ragged.npz", data=pts, label=lab, data_offsets=off)
    common = dict(IO_TYPE="npz", INPUT_FILE=str(tmp_path / "ragged.npz"), NUM_POINT=-1, BATCH_SIZE=6, MINIBATCH_SIZE=1,
                  SHUFFLE=0, KVALUE=8, NUM_CHANNEL=-1)
    f = _flags(tmp_path, ITERATION=3, CHECKPOINT_STEP=3, SUMMARY_STEP=0, REPORT_STEP=0, **common)
    h = M.train(f)
    assert f.NUM_CHANNEL == 4 and h.data_io.num_entries() == 6           # the 120-point cloud is dropped
    rows = _rows(tmp_path / "log" / "train_log-0000000.csv")
    assert len(rows) == 3 and all(np.isfinite(float(r["loss"])) for r in rows)
    assert os.path.exists(f.WEIGHT_PREFIX + "-2.npz")
    g = _flags(tmp_path, ITERATION=1, MODEL_PATH=f.WEIGHT_PREFIX + "-2", OUTPUT_FILE=str(tmp_path / "out.npz"),
               LOG_DIR=str(tmp_path / "ilog"), **common)
    h2 = M.inference(g)
    capsys.readouterr()
    z = np.load(tmp_path / "out.npz")
    kept = [c for c in counts if c >= 256]
    assert z["idx"].tolist() == [0, 1, 2, 3, 4, 5] and np.diff(z["data_offsets"]).tolist() == kept
    assert z["softmax"].shape == (sum(kept), 2) and np.allclose(z["softmax"].sum(1), 1.0, atol=1e-5)
    # the stored rows of the third kept cloud (N = 517) == a direct inference call on that cloud
    lo, hi = int(z["data_offsets"][2]), int(z["data_offsets"][3])
    cloud = z["data"][lo:hi]
    assert cloud.shape == (517, 4)
    sm = h2.trainer.inference(None, [cloud[None]])[0].cpu().numpy()[0]
    np.testing.assert_allclose(z["softmax"][lo:hi], sm, rtol=0, atol=2e-4)
    # a micro-batch that mixes point counts is refused with the reason
    bad = _flags(tmp_path, ITERATION=1, CHECKPOINT_STEP=0, LOG_DIR="", **dict(common, MINIBATCH_SIZE=2))
    with pytest.raises(ValueError, match="minibatch_size 1"):
        M.train(bad)


def test_train_and_inference_from_real_hdf5_files(tmp_path, capsys):
    """N2 on the GPU box: `-io h5` on files the HDF5 library wrote (tests/golden/h5: the reference's dense layout, contiguous and
    chunked + gzip + shuffle), read here by dgcnn/_h5min.py where h5py is absent -- a short training run (CSV, finite losses), then
    inference with an output file (iotool.py:233-245) that is re-read."""
    h5 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "h5")
    files = [os.path.join(h5, "contiguous.h5"), os.path.join(h5, "chunked_gzip_shuffle.h5")]
    exp = np.load(os.path.join(h5, "expected.npz"))
    f = _flags(tmp_path, IO_TYPE="h5", INPUT_FILE=files, DATA_KEY="data", LABEL_KEY="label", WEIGHT_KEY="weight", NUM_CHANNEL=4,
               NUM_CLASS=3, BATCH_SIZE=5, MINIBATCH_SIZE=5, KVALUE=6, ITERATION=4, CHECKPOINT_STEP=4, SUMMARY_STEP=0, REPORT_STEP=0)
    M.train(f)
    rows = _rows(tmp_path / "log" / "train_log-0000000.csv")
    assert len(rows) == 4 and all(np.isfinite(float(r["loss"])) for r in rows)
    out = str(tmp_path / "pred.h5")
    g = _flags(tmp_path, IO_TYPE="h5", INPUT_FILE=files, DATA_KEY="data", LABEL_KEY="label", NUM_CHANNEL=4, NUM_CLASS=3, BATCH_SIZE=5,
               MINIBATCH_SIZE=5, KVALUE=6, ITERATION=2, SHUFFLE=0, MODEL_PATH=f.WEIGHT_PREFIX + "-3", OUTPUT_FILE=out, LOG_DIR=str(tmp_path / "ilog"))
    M.inference(g)
    capsys.readouterr()
    try:
        import h5py
        opener = lambda p: h5py.File(p, "r")
    except ImportError:
        from dgcnn import _h5min
        opener = _h5min.File
    with opener(out) as z:
        assert sorted(z.keys()) == ["data", "idx", "label", "softmax"]
        idx = np.asarray(z["idx"])
        assert idx.tolist() == list(range(10))
        sm = np.asarray(z["softmax"])
        assert sm.shape == (10, 37, 3) and np.allclose(sm.sum(-1), 1.0, atol=1e-5)
        np.testing.assert_array_equal(np.asarray(z["data"])[:5], exp["data"])
