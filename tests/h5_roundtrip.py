#!/usr/bin/env python
"""Round-trips REAL HDF5 files through dgcnn.iotool.io_h5 (run with an interpreter that has h5py; in this image that
is /opt/conda/bin/python3.9, which has numpy + h5py but no torch -- so iotool.py is loaded by path, it imports numpy only).

Writes, with h5py, a file in the reference's dense layout (datasets DATA_KEY (entries,N,C) f32, LABEL_KEY (entries,N),
WEIGHT_KEY (entries,N): dgcnn/iotool.py:212-231, flags.py:42-44) and one in the ragged layout (<DATA_KEY>_offsets),
reads them back through io_h5 (sequential wrap-around and shuffle), stores softmax rows and re-opens the output file.
Then the other direction for dgcnn/_h5min.py (the plain-Python stand-in of h5py on hosts without it): files IT writes are opened with
the real HDF5 library here, and files h5py writes in every storage variant it claims are read by it and compared with h5py's own
reading.  Prints H5_ROUNDTRIP_OK on success.   usage: h5_roundtrip.py <workdir>"""
import importlib.util
import os
import sys

import numpy as np
import h5py

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("dgcnn_iotool", os.path.join(HERE, "..", "dynamic-gcnn_amd", "dgcnn", "iotool.py"))
iotool = importlib.util.module_from_spec(spec)
spec.loader.exec_module(iotool)


class Flags(object):
    IO_TYPE = "h5"; BATCH_SIZE = 4; SHUFFLE = 0; SEED = 3; DATA_KEY = "data"; LABEL_KEY = "label"; WEIGHT_KEY = ""
    INPUT_FILE = ""; OUTPUT_FILE = ""

    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)


def h5min_against_the_library(work):
    M = iotool._h5min_module()
    rng = np.random.default_rng(5)
    arrs = {"data": rng.random((3, 17, 4), dtype=np.float32), "label": rng.integers(-3, 3, (3, 17)).astype(np.int64),
            "softmax": rng.random((3, 17, 2)), "idx": np.arange(3, dtype=np.int64), "u8": np.arange(9, dtype=np.uint8),
            "s": np.float32(1.5), "e": np.zeros((0, 4), np.float32)}
    path = os.path.join(work, "written_by_h5min.h5")
    with M.File(path, "w") as f:
        for k, v in arrs.items():
            f.create_dataset(k, data=v)
    with h5py.File(path, "r") as f:                                   # the HDF5 library is a strict parser
        assert sorted(f.keys()) == sorted(arrs)
        for k, v in arrs.items():
            assert f[k].shape == np.shape(v) and f[k].dtype == np.asarray(v).dtype and np.array_equal(f[k][()], v), k
    # ... and h5py-written storage variants read by _h5min
    variants = dict(contiguous={}, chunked=dict(chunks=(2, 5, 3)), gzip=dict(chunks=(2, 5, 3), compression="gzip"),
                    shuffle_gzip_fletcher=dict(chunks=(3, 17, 2), shuffle=True, compression="gzip", fletcher32=True))
    for libver in ("earliest", "latest"):
        for name, kw in variants.items():
            p = os.path.join(work, "v_%s_%s.h5" % (libver, name))
            a = rng.random((5, 17, 4), dtype=np.float32)
            b = rng.integers(0, 1000, (5, 17)).astype(np.int32)
            with h5py.File(p, "w", libver=libver) as f:
                f.create_dataset("a", data=a, **kw)
                f.create_dataset("b", data=b, **({k: (v[:2] if k == "chunks" else v) for k, v in kw.items()}))
            with M.File(p) as f:
                assert np.array_equal(f["a"], a) and np.array_equal(f["b"], b), (libver, name)
    # an extendable dataset with attributes, the way PyTables' earrays of the reference's output are stored (iotool.py:233-245)
    p = os.path.join(work, "v_earray_like.h5")
    a = rng.random((7, 17, 4), dtype=np.float32)
    with h5py.File(p, "w") as f:
        d = f.create_dataset("softmax", shape=(0, 17, 4), maxshape=(None, 17, 4), dtype="f4", chunks=(2, 17, 4), compression="gzip")
        d.attrs["CLASS"] = "EARRAY"
        d.attrs["TITLE"] = "softmax"
        for i in range(7):
            d.resize(i + 1, axis=0)
            d[i] = a[i]
        f.attrs["PYTABLES_FORMAT_VERSION"] = "2.1"
    with M.File(p) as f:
        assert f["softmax"].shape == (7, 17, 4) and np.array_equal(f["softmax"], a)


def main(work):
    h5min_against_the_library(work)
    rng = np.random.default_rng(0)
    # ---- dense layout, two files (the reference's multi-file concatenate, iotool.py:227-229, with the intent implemented)
    files = []
    for i in range(2):
        path = os.path.join(work, "dense%d.h5" % i)
        with h5py.File(path, "w") as f:
            f.create_dataset("data", data=rng.random((3, 32, 4), dtype=np.float32))
            f.create_dataset("label", data=rng.integers(0, 3, (3, 32)).astype(np.int64))
            f.create_dataset("weight", data=rng.random((3, 32), dtype=np.float32))
        files.append(path)
    out = os.path.join(work, "out_dense.h5")
    io = iotool.io_factory(Flags(INPUT_FILE=files, WEIGHT_KEY="weight", OUTPUT_FILE=out))
    assert type(io).__name__ == "io_h5"
    io.initialize()
    assert (io.num_entries(), io.num_channels()) == (6, 4)
    idx, data, label, weight = io.next()
    with h5py.File(files[0], "r") as f:
        assert idx.tolist() == [0, 1, 2, 3] and np.array_equal(data[:3], f["data"][...]) and np.array_equal(label[:3], f["label"][...])
        assert np.array_equal(weight[:3], f["weight"][...]) and data.dtype == np.float32 and label.dtype == np.int32
    idx2 = io.next()[0]
    assert idx2.tolist() == [4, 5, 0, 1]                               # sequential wrap-around (iotool.py:264-272)
    sm = rng.random((32, 3), dtype=np.float32)
    io.store(idx[2], sm)
    io.finalize()
    with h5py.File(out, "r") as f:
        assert sorted(f.keys()) == ["data", "idx", "label", "softmax"]  # iotool.py:238-245: data / softmax / label
        assert np.array_equal(f["softmax"][0], sm) and f["idx"][...].tolist() == [2] and f["data"].shape == (1, 32, 4)
    sh = iotool.io_factory(Flags(INPUT_FILE=files, SHUFFLE=1))
    sh.initialize()
    assert len(set(sh.next()[0].tolist())) == 4

    # ---- ragged layout: clouds of different N, one below the 256-point cut (iotool.py:81)
    counts = [300, 100, 512, 257, 1024]
    off = np.concatenate([[0], np.cumsum(counts)])
    pts = rng.random((off[-1], 4), dtype=np.float32)
    lab = rng.integers(0, 2, off[-1]).astype(np.int32)
    path = os.path.join(work, "ragged.h5")
    with h5py.File(path, "w") as f:
        f.create_dataset("data", data=pts)
        f.create_dataset("label", data=lab)
        f.create_dataset("data_offsets", data=off.astype(np.int64))
    out = os.path.join(work, "out_ragged.h5")
    io = iotool.io_factory(Flags(INPUT_FILE=[path], BATCH_SIZE=3, OUTPUT_FILE=out))
    io.initialize()
    assert io.num_entries() == 4 and io.num_channels() == 4           # the 100-point cloud is dropped
    idx, data, label, weight = io.next()
    assert isinstance(data, list) and [len(d) for d in data] == [300, 512, 257] and weight is None
    assert np.array_equal(data[1], pts[off[2]:off[3]]) and np.array_equal(label[1], lab[off[2]:off[3]])
    assert [len(d) for d in io.next()[1]] == [1024, 300, 512]
    io.store(0, rng.random((300, 2), dtype=np.float32))
    io.store(3, rng.random((1024, 2), dtype=np.float32))
    io.finalize()
    with h5py.File(out, "r") as f:
        assert f["data_offsets"][...].tolist() == [0, 300, 1324] and f["softmax"].shape == (1324, 2) and f["data"].shape == (1324, 4)
    print("H5_ROUNDTRIP_OK h5py %s" % h5py.__version__)


if __name__ == "__main__":
    main(sys.argv[1])
