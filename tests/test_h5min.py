"""dgcnn/_h5min.py -- the plain-Python HDF5 reader / writer behind `io_h5` on hosts without h5py (N2 of SURVEY 8f;
dgcnn/iotool.py:199-280) -- against files written by the REAL HDF5 library (tests/golden/h5/*.h5, generated with h5py by
tests/make_h5_fixtures.py together with expected.npz), and `io_h5` end to end on them without h5py."""
import os
import sys

import numpy as np
import pytest

import dgcnn
from dgcnn import _h5min as M

H5 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "h5")
FILES = {
    "contiguous.h5": ["data", "label", "weight"],                       # superblock 0, symbol-table group, contiguous storage
    "chunked_gzip_shuffle.h5": ["data", "label", "weight"],            # B-tree v1 chunk index; deflate, shuffle, fletcher32; edge chunks
    "latest.h5": ["data", "label", "weight"],                          # superblock 3, OHDR v2, link messages; single-chunk and fixed-array indexes
    "latest_implicit.h5": ["early"],                                   # implicit chunk index
    "types.h5": ["be_i4", "u1", "i2", "be_f8", "scalar", "compact", "never_written", "many_chunks"],   # user block; number types; 2-level B-tree
}


@pytest.mark.parametrize("fname", sorted(FILES))
def test_reader_against_files_written_by_the_hdf5_library(fname):
    exp = np.load(os.path.join(H5, "expected.npz"))
    with M.File(os.path.join(H5, fname)) as f:
        assert sorted(f.keys()) == sorted(FILES[fname])
        for name in FILES[fname]:
            a = f[name]
            assert a.shape == exp[name].shape and a.dtype.kind == exp[name].dtype.kind and a.dtype.itemsize == exp[name].dtype.itemsize, name
            assert a.dtype.isnative
            np.testing.assert_array_equal(a, exp[name], err_msg="%s:%s" % (fname, name))
        with pytest.raises(KeyError):
            f["no such dataset"]


def test_writer_round_trip_and_limits(tmp_path):
    rng = np.random.default_rng(1)
    arrs = {"data": rng.random((3, 17, 4), dtype=np.float32), "label": rng.integers(-3, 3, (3, 17)).astype(np.int64),
            "softmax": rng.random((3, 17, 2)), "idx": np.arange(3, dtype=np.int64), "s": np.float32(1.5), "e": np.zeros((0, 4), np.float32),
            "flag": np.array([True, False, True])}
    path = str(tmp_path / "w.h5")
    with M.File(path, "w") as f:
        for k, v in arrs.items():
            f.create_dataset(k, data=v, compression="gzip", compression_opts=5)      # (options accepted, storage contiguous)
    with M.File(path) as f:
        assert sorted(f.keys()) == sorted(arrs)
        for k, v in arrs.items():
            want = np.asarray(v).astype(np.uint8) if np.asarray(v).dtype == bool else np.asarray(v)
            assert f[k].shape == want.shape and f[k].dtype == want.dtype, k
            np.testing.assert_array_equal(f[k], want)
    with pytest.raises(M.H5FormatError):
        M.write_file(str(tmp_path / "many.h5"), {"d%d" % i: np.zeros(2) for i in range(9)})
    with pytest.raises(M.H5FormatError):
        M.write_file(str(tmp_path / "c.h5"), {"z": np.zeros(2, np.complex64)})
    open(tmp_path / "junk.h5", "wb").write(b"not an hdf5 file" * 100)
    with pytest.raises(M.H5FormatError):
        M.File(str(tmp_path / "junk.h5"))


def test_unsupported_content_is_refused_not_misread(tmp_path):
    """A dataset of a type the reader does not know (here: a file whose datatype message is patched to the string class) must raise,
    never return numbers."""
    src = open(os.path.join(H5, "contiguous.h5"), "rb").read()
    with M.File(os.path.join(H5, "contiguous.h5")) as f:
        r = f._r
        addr = r.links["label"]
        (s,) = [s for t, _fl, s, _n in r._messages(addr) if t == 0x03]
    patched = bytearray(src)
    patched[s] = (patched[s] & 0xF0) | 3                                # class 3 = string
    p = tmp_path / "patched.h5"
    p.write_bytes(bytes(patched))
    with M.File(str(p)) as f:
        np.testing.assert_array_equal(f["data"], np.load(os.path.join(H5, "expected.npz"))["data"])     # the others still read
        with pytest.raises(M.H5FormatError, match="datatype class 3"):
            f["label"]


class _Flags(dgcnn.DGCNN_FLAGS):
    pass


def test_io_h5_reads_and_writes_real_hdf5_without_h5py(tmp_path, monkeypatch):
    """The reference's dense layout (iotool.py:212-231) from files the HDF5 library wrote -- contiguous and chunked + compressed --
    through dgcnn.io_factory with h5py ABSENT: batches, sequential wrap-around, output file (iotool.py:233-245) re-read."""
    monkeypatch.setitem(sys.modules, "h5py", None)                     # `import h5py` raises ImportError from here on
    exp = np.load(os.path.join(H5, "expected.npz"))
    files = [os.path.join(H5, "contiguous.h5"), os.path.join(H5, "chunked_gzip_shuffle.h5")]
    out = str(tmp_path / "out.h5")
    io = dgcnn.io_factory(dgcnn.DGCNN_FLAGS(IO_TYPE="h5", INPUT_FILE=files, DATA_KEY="data", LABEL_KEY="label", WEIGHT_KEY="weight",
                                            BATCH_SIZE=4, SHUFFLE=0, OUTPUT_FILE=out))
    assert type(io).__name__ == "io_h5" and io._h5.__name__.endswith("_h5min")
    io.initialize()
    assert (io.num_entries(), io.num_channels()) == (10, 4)
    idx, data, label, weight = io.next()
    assert idx.tolist() == [0, 1, 2, 3]
    np.testing.assert_array_equal(data, exp["data"][:4])
    np.testing.assert_array_equal(label, exp["label"][:4].astype(np.int32))
    np.testing.assert_array_equal(weight, exp["weight"][:4].astype(np.float32))
    assert data.dtype == np.float32 and label.dtype == np.int32 and weight.dtype == np.float32
    io.next()
    idx3, data3, _, _ = io.next()
    assert idx3.tolist() == [8, 9, 0, 1]                               # wrap-around; entries 5..9 come from the second (chunked) file
    np.testing.assert_array_equal(data3[:2], exp["data"][3:5])
    sm = np.random.default_rng(0).random((37, 3), dtype=np.float32)
    io.store(7, sm)
    io.finalize()
    with M.File(out) as f:
        assert sorted(f.keys()) == ["data", "idx", "label", "softmax"]
        assert f["idx"].tolist() == [7]
        np.testing.assert_array_equal(f["softmax"][0], sm)
        np.testing.assert_array_equal(f["data"][0], exp["data"][2])     # entry 7 = entry 2 of the second file
