"""Batch sources either side of the hot path (reference dgcnn/iotool.py:6-287).

Same protocol as the reference's `io_base` (iotool.py:6-32): `initialize()`, `next()` ->
`(idx, data, label, weight)` with data (BATCH_SIZE, N, C) float32, `store(idx, softmax)`,
`finalize()`, `num_entries()`, `num_channels()`, `batch_size()`.  The file formats differ on
purpose: LArCV/ROOT and h5py/PyTables are absent from this image (out of scope), so
the sources here are

  IO_TYPE 'npz'        arrays DATA_KEY / LABEL_KEY / WEIGHT_KEY of one or more .npz files, the same
                       dense (entries, N, C) / (entries, N) layout io_h5 reads (iotool.py:212-231);
                       OUTPUT_FILE collects data / softmax / label rows like io_h5.store.
                       RAGGED files (clouds with different point counts -- what io_larcv yields, iotool.py:76-98, and
                       what production trains on with `-np -1 -mbs 1`, scripts/lsf/train_dgcnn.sh:28) additionally
                       hold `<DATA_KEY>_offsets` (entries+1 int64): cloud e = rows offsets[e]:offsets[e+1] of the
                       concatenated DATA_KEY (points, C) / LABEL_KEY (points,) / WEIGHT_KEY (points,) arrays.  Clouds
                       with fewer than MIN_POINTS (256, iotool.py:81) points are dropped at load time; next() returns
                       LISTS of per-cloud arrays, like io_larcv.next (iotool.py:127-150).
  IO_TYPE 'h5'         the same two layouts as HDF5 datasets through h5py when that module is importable (it is
                       optional, as in the reference README); otherwise NotImplementedError.
  IO_TYPE 'synthetic'  seeded clouds of the benchmark's shape; labels are a deterministic function of
                       position so that a training loop has something to learn.

'larcv' raises NotImplementedError with the reason.  Entry order follows iotool.py:262-279:
SHUFFLE draws BATCH_SIZE distinct entries per call, otherwise a cursor wraps around the file.
"""
from __future__ import annotations

import numpy as np


class io_base(object):

    def __init__(self, flags):
        self._flags = flags
        self._batch_size = int(flags.BATCH_SIZE)
        self._num_entries = -1
        self._num_channels = -1
        self._data = self._label = self._weight = None
        self._cursor = 0
        seed = int(getattr(flags, "SEED", -1))
        self._rng = np.random.RandomState(seed if seed >= 0 else None)

    def batch_size(self, size=None):
        if size is None:
            return self._batch_size
        self._batch_size = int(size)

    def num_entries(self):
        return self._num_entries

    def num_channels(self):
        return self._num_channels

    def initialize(self):
        raise NotImplementedError

    def store(self, idx, softmax):
        raise NotImplementedError

    def finalize(self):
        pass

    # entry selection shared by the dense sources (iotool.py:262-279)
    def _draw(self):
        n, bs = self._num_entries, self._batch_size
        if getattr(self._flags, "SHUFFLE", 0):
            if bs <= n:
                return self._rng.permutation(n)[:bs]
            return self._rng.randint(0, n, size=bs)
        idx = (self._cursor + np.arange(bs)) % n
        self._cursor = int(idx[-1] + 1) % n
        return idx

    def next(self):
        idx = self._draw()
        pick = lambda a: None if a is None else a[idx, ...]
        return idx, pick(self._data), pick(self._label), pick(self._weight)


class io_npz(io_base):

    def __init__(self, flags):
        super(io_npz, self).__init__(flags)
        self._out = None
        self._ragged = False

    # container access; io_h5 overrides these two
    def _open(self, path):
        return np.load(path)

    def _write(self, path, arrays):
        np.savez_compressed(path, **arrays)

    def initialize(self):
        f = self._flags
        files = f.INPUT_FILE if isinstance(f.INPUT_FILE, (list, tuple)) else str(f.INPUT_FILE).split(",")
        parts = {"data": [], "label": [], "weight": []}
        okey = f.DATA_KEY + "_offsets"
        ragged = None
        for path in files:
            with self._open(path) as z:
                names = set(z.keys())
                is_ragged = okey in names
                if ragged is None:
                    ragged = is_ragged
                elif ragged != is_ragged:
                    raise ValueError("cannot mix dense and ragged ('%s' present) input files" % okey)
                data = np.asarray(z[f.DATA_KEY], np.float32)
                label = np.asarray(z[f.LABEL_KEY], np.int32) if getattr(f, "LABEL_KEY", "") else None
                weight = np.asarray(z[f.WEIGHT_KEY], np.float32) if getattr(f, "WEIGHT_KEY", "") else None
                if ragged:
                    off = np.asarray(z[okey], np.int64)
                    if data.ndim != 2 or off.ndim != 1 or off[0] != 0 or off[-1] != len(data) or (np.diff(off) < 0).any():
                        raise ValueError("ragged file %s: '%s' must be (points, channels) and '%s' a non-decreasing "
                                         "prefix array ending at %d" % (path, f.DATA_KEY, okey, len(data)))
                    for name, a in (("label", label), ("weight", weight)):
                        if a is not None and a.shape != (len(data),):
                            raise ValueError("%s shape %s does not match the %d points of '%s'" % (name, a.shape, len(data), f.DATA_KEY))
                    for e in range(len(off) - 1):
                        lo, hi = int(off[e]), int(off[e + 1])
                        if hi - lo < int(getattr(f, "MIN_POINTS", 256)):     # iotool.py:81: `if num_point < 256: continue`
                            continue
                        parts["data"].append(data[lo:hi])
                        if label is not None:
                            parts["label"].append(label[lo:hi])
                        if weight is not None:
                            parts["weight"].append(weight[lo:hi])
                else:
                    parts["data"].append(data)
                    if label is not None:
                        parts["label"].append(label)
                    if weight is not None:
                        parts["weight"].append(weight)
        self._ragged = bool(ragged)
        if self._ragged:
            self._data = parts["data"]
            self._label = parts["label"] or None
            self._weight = parts["weight"] or None
            if not self._data:
                raise ValueError("no cloud with at least %d points in %s" % (int(getattr(f, "MIN_POINTS", 256)), files))
            self._num_entries, self._num_channels = len(self._data), self._data[0].shape[1]
        else:
            cat = lambda v: np.concatenate(v, axis=0) if v else None
            self._data, self._label, self._weight = cat(parts["data"]), cat(parts["label"]), cat(parts["weight"])
            if self._data is None or self._data.ndim != 3:
                raise ValueError("'%s' must hold (entries, points, channels), got %s"
                                 % (f.DATA_KEY, None if self._data is None else self._data.shape))
            for name, a in (("label", self._label), ("weight", self._weight)):
                if a is not None and a.shape != self._data.shape[:2]:
                    raise ValueError("%s shape %s does not match data %s" % (name, a.shape, self._data.shape))
            self._num_entries, _, self._num_channels = self._data.shape
        self._cursor = 0
        if getattr(f, "OUTPUT_FILE", ""):
            self._out = {"idx": [], "data": [], "softmax": [], "label": []}

    def next(self):
        if not self._ragged:
            return super(io_npz, self).next()
        n, bs = self._num_entries, self._batch_size
        if getattr(self._flags, "SHUFFLE", 0) and bs <= n:
            start = int(self._rng.random_sample() * (n - bs))          # io_larcv.next: a random contiguous window
            idx = np.arange(start, start + bs)
        else:
            idx = (self._cursor + np.arange(bs)) % n                     # sequential with wrap-around
            self._cursor = int(idx[-1] + 1) % n
        pick = lambda v: None if v is None else [v[i] for i in idx]
        return idx, pick(self._data), pick(self._label), pick(self._weight)

    def store(self, idx, softmax):
        if self._out is None:
            raise NotImplementedError
        idx = int(idx)
        if idx >= self._num_entries:
            raise ValueError
        self._out["idx"].append(idx)
        self._out["data"].append(self._data[idx])
        self._out["softmax"].append(np.asarray(softmax, np.float32))
        if self._label is not None:
            self._out["label"].append(self._label[idx])

    def finalize(self):
        if self._out is not None and self._out["idx"]:
            out = {"idx": "idx", "data": self._flags.DATA_KEY, "softmax": "softmax",
                   "label": getattr(self._flags, "LABEL_KEY", "") or "label"}
            if self._ragged:                                             # same CSR layout as the input
                arrays = {"idx": np.asarray(self._out["idx"], np.int64),
                          self._flags.DATA_KEY + "_offsets": np.concatenate([[0], np.cumsum([len(d) for d in self._out["data"]])]).astype(np.int64)}
                for k in ("data", "softmax", "label"):
                    if self._out[k]:
                        arrays[out[k]] = np.concatenate(self._out[k], axis=0)
                self._write(self._flags.OUTPUT_FILE, arrays)
            else:
                self._write(self._flags.OUTPUT_FILE, {out[k]: np.stack(v) for k, v in self._out.items() if v})
        self._out = None


def _h5min_module():
    """dgcnn/_h5min.py, also when this file was loaded by path outside the package (tests/h5_roundtrip.py does)."""
    try:
        from . import _h5min
        return _h5min
    except ImportError:
        import importlib.util
        import os
        spec = importlib.util.spec_from_file_location("dgcnn_h5min", os.path.join(os.path.dirname(os.path.abspath(__file__)), "_h5min.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod


class io_h5(io_npz):
    """HDF5 flavour of the same layouts (iotool.py:199-280): datasets DATA_KEY (entries, N, C),
    LABEL_KEY / WEIGHT_KEY (entries, N), or the ragged form with `<DATA_KEY>_offsets`.  With h5py where it is installed; otherwise
    through dgcnn/_h5min.py -- a plain-Python reader of what h5py / PyTables / the HDF5 library write by default (contiguous, compact
    and chunked datasets with deflate / shuffle / fletcher32, superblock 0-3) and a writer of contiguous datasets that the HDF5
    library reads back (the reference writes its output with PyTables earrays: the same three datasets, iotool.py:233-245)."""

    def __init__(self, flags):
        super(io_h5, self).__init__(flags)
        try:
            import h5py
            self._h5 = h5py
        except ImportError:
            self._h5 = _h5min_module()

    def _open(self, path):
        return self._h5.File(path, "r")

    def _write(self, path, arrays):
        with self._h5.File(path, "w") as f:
            for name, a in arrays.items():
                f.create_dataset(name, data=a, compression="gzip", compression_opts=5)


class io_synthetic(io_base):
    """NUM_ENTRIES (default 64) clouds of NUM_POINT points, NUM_CHANNEL (default 3) channels, uniform
    in the unit cube; label = which side of a tilted plane a point lies on (NUM_CLASS bands)."""

    def initialize(self):
        f = self._flags
        n = int(getattr(f, "NUM_ENTRIES", 64))
        npt = int(f.NUM_POINT)
        ch = int(f.NUM_CHANNEL) if int(getattr(f, "NUM_CHANNEL", -1)) > 0 else 3
        g = np.random.RandomState(20180801)
        self._data = g.random_sample((n, npt, ch)).astype(np.float32)
        score = self._data[..., :3].mean(-1) if ch >= 3 else self._data[..., 0]
        self._label = np.minimum((score * int(f.NUM_CLASS)).astype(np.int32), int(f.NUM_CLASS) - 1)
        self._weight = np.ones((n, npt), np.float32) if getattr(f, "WEIGHT_KEY", "") else None
        self._num_entries, self._num_channels = n, ch
        self._cursor = 0
        self._stored = {}

    def store(self, idx, softmax):
        if int(idx) >= self._num_entries:
            raise ValueError
        self._stored[int(idx)] = np.asarray(softmax, np.float32)


def io_factory(flags):
    """iotool.py:282-287."""
    kind = getattr(flags, "IO_TYPE", "")
    if kind == "npz":
        return io_npz(flags)
    if kind == "synthetic":
        return io_synthetic(flags)
    if kind == "h5":
        return io_h5(flags)
    if kind == "larcv":
        raise NotImplementedError("IO_TYPE 'larcv' needs LArCV/ROOT, which this build does not bind; export the "
                                  "sparse3d voxels as dense (entries, N, C) arrays and use IO_TYPE 'npz' or 'h5'")
    raise NotImplementedError("unknown IO_TYPE '%s'" % kind)
