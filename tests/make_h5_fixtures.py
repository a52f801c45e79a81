#!/usr/bin/env python
"""Writes, WITH THE REAL HDF5 LIBRARY (h5py; in this image: /opt/conda/bin/python3.9), the small files under tests/golden/h5/ that pin
dgcnn/_h5min.py's reader, and expected.npz with the arrays they hold.  Every storage variant the reader claims: contiguous,
compact, chunked with deflate / shuffle / fletcher32, partial edge chunks, an unallocated chunked dataset, big-endian numbers,
superblock 0 (default) and 2+ (libver='latest': version-2 object headers, compact link messages), a file with a user block.
usage: /opt/conda/bin/python3.9 tests/make_h5_fixtures.py"""
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "golden", "h5")


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20260930)
    data = rng.random((5, 37, 4), dtype=np.float32)
    label = rng.integers(0, 3, (5, 37)).astype(np.int64)
    weight = rng.random((5, 37)).astype(np.float64)
    expected = {"data": data, "label": label, "weight": weight}
    with h5py.File(os.path.join(OUT, "contiguous.h5"), "w") as f:                       # the reference's dense layout, default storage
        f.create_dataset("data", data=data)
        f.create_dataset("label", data=label)
        f.create_dataset("weight", data=weight)
    with h5py.File(os.path.join(OUT, "chunked_gzip_shuffle.h5"), "w") as f:             # edge chunks in every dimension
        f.create_dataset("data", data=data, chunks=(2, 16, 3), compression="gzip", compression_opts=5, shuffle=True)
        f.create_dataset("label", data=label, chunks=(3, 10), compression="gzip")
        f.create_dataset("weight", data=weight, chunks=(5, 37), fletcher32=True)
    with h5py.File(os.path.join(OUT, "latest.h5"), "w", libver="latest") as f:          # superblock 3, OHDR v2, link messages
        f.create_dataset("data", data=data, chunks=(5, 37, 4), shuffle=True, compression="gzip", fletcher32=True)
        f.create_dataset("label", data=label, chunks=(2, 10), compression="gzip")          # fixed-array chunk index, filtered chunks
        f.create_dataset("weight", data=weight, chunks=(3, 20))                            # fixed-array chunk index, plain chunks
    with h5py.File(os.path.join(OUT, "latest_implicit.h5"), "w", libver="latest") as f:
        dcpl = h5py.h5p.create(h5py.h5p.DATASET_CREATE)
        dcpl.set_chunk((4, 4))
        dcpl.set_alloc_time(h5py.h5d.ALLOC_TIME_EARLY)
        sid = h5py.h5s.create_simple((6, 10))
        did = h5py.h5d.create(f.id, b"early", h5py.h5t.NATIVE_INT16, sid, dcpl=dcpl)
        h5py.Dataset(did)[...] = np.arange(60, dtype=np.int16).reshape(6, 10)
    with h5py.File(os.path.join(OUT, "types.h5"), "w", userblock_size=512) as f:        # signature at 512; other number types
        f.create_dataset("be_i4", data=np.arange(-6, 6, dtype=">i4").reshape(3, 4))
        f.create_dataset("u1", data=np.arange(200, 230, dtype=np.uint8))
        f.create_dataset("i2", data=np.arange(-5, 5, dtype=np.int16))
        f.create_dataset("be_f8", data=np.linspace(-1, 1, 7).astype(">f8"))
        f.create_dataset("scalar", data=np.float32(2.5))
        f.create_dataset("compact", data=np.arange(6, dtype=np.int32), dtype="<i4")
        f.create_dataset("never_written", shape=(4, 6), dtype="f4", chunks=(2, 3))
        many = f.create_dataset("many_chunks", shape=(70, 9), dtype="i4", chunks=(2, 2))   # > 64 chunks: a two-level chunk B-tree
        m = np.arange(70 * 9, dtype=np.int32).reshape(70, 9)
        many[...] = m
    expected.update(be_i4=np.arange(-6, 6, dtype=np.int32).reshape(3, 4), u1=np.arange(200, 230, dtype=np.uint8),
                    i2=np.arange(-5, 5, dtype=np.int16), be_f8=np.linspace(-1, 1, 7), scalar=np.float32(2.5),
                    compact=np.arange(6, dtype=np.int32), never_written=np.zeros((4, 6), np.float32), many_chunks=m,
                    early=np.arange(60, dtype=np.int16).reshape(6, 10))
    np.savez(os.path.join(OUT, "expected.npz"), **expected)
    for n in sorted(os.listdir(OUT)):
        print("%-28s %7d bytes" % (n, os.path.getsize(os.path.join(OUT, n))))


if __name__ == "__main__":
    main()
