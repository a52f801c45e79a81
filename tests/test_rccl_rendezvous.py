"""The unique-id rendezvous of dgcnn/rccl.py (plain TCP, no torch.distributed): rank 0 serves the 128-byte id to every other
rank, late joiners retry until rank 0 listens.  CPU only: the id is a stand-in byte string."""
import socket
import threading
import time

from dgcnn import rccl


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_every_rank_receives_rank_zeros_id():
    port = _free_port()
    ident = bytes(range(128))
    got = {}

    def worker(rank, delay):
        time.sleep(delay)
        got[rank] = rccl._exchange_id(rank, 4, "127.0.0.1", port, lambda: ident, timeout=30.0)
    threads = [threading.Thread(target=worker, args=(r, d)) for r, d in ((1, 0.0), (2, 0.0), (0, 0.5), (3, 0.8))]   # rank 0 arrives late
    for t in threads:
        t.start()
    for t in threads:
        t.join(40)
    assert got == {r: ident for r in range(4)}


def test_single_rank_needs_no_socket():
    assert rccl._exchange_id(0, 1, "127.0.0.1", 1, lambda: b"x" * 128) == b"x" * 128
