"""A stand-in for dgcnn.rccl.Group that needs no GPU and no RCCL: same constructor and method surface, collectives over plain TCP
through rank 0 (host tensors).  Test infrastructure only -- `bench.py --dry-run` loads it through $DGCNN_BENCH_GROUP to run the
N-rank plumbing (self-launch, rendezvous with late ranks, barrier-fenced timing, gathers, the JSON line) on a CPU-only box.
The rendezvous is the product's own (`dgcnn.rccl._exchange_id`): rank 0 serves a 128-byte id, the others fetch it with retries."""
import os
import pickle
import socket
import struct
import time

import torch

from dgcnn import rccl


def _send(sock, obj):
    blob = pickle.dumps(obj, protocol=4)
    sock.sendall(struct.pack("<Q", len(blob)) + blob)


def _recv(sock):
    def exact(n):
        buf = bytearray()
        while len(buf) < n:
            chunk = sock.recv(min(1 << 20, n - len(buf)))
            if not chunk:
                raise ConnectionError("peer closed")
            buf += chunk
        return bytes(buf)
    (n,) = struct.unpack("<Q", exact(8))
    return pickle.loads(exact(n))


class StubGroup(object):
    def __init__(self, rank=None, world=None, addr=None, port=None):
        env = os.environ
        self.rank = int(env.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(env.get("WORLD_SIZE", "1")) if world is None else int(world)
        addr = env.get("MASTER_ADDR", "127.0.0.1") if addr is None else addr
        port = int(env.get("MASTER_PORT", "29500")) + 1 if port is None else int(port)
        delay = float(env.get("STUB_RCCL_DELAY_RANK%d" % self.rank, "0"))      # tests make a rank arrive late
        if delay:
            time.sleep(delay)
        # the stage announcements of dgcnn.rccl.Group (DGCNN_RCCL_TRACE), and a failure injected at one of them on one rank
        # (STUB_RCCL_FAIL="<rank>:<stage>"): what profiles/scale_probe.sh diagnoses
        # STUB_RCCL_HANG="<rank>:<n>": that rank's n-th collective (0-based, counted from the constructor's first one) never
        # completes -- the rank sits in it for ever and its peers block waiting for its contribution: what a fabric that does not
        # come up looks like.  Nothing raises; only a deadline outside the communicator can end the run (bench.py's guard).
        hang = env.get("STUB_RCCL_HANG", "")
        self._hang_at = int(hang.split(":", 1)[1]) if hang and int(hang.split(":", 1)[0]) == self.rank else None
        self._ncoll = 0
        fail = env.get("STUB_RCCL_FAIL", "")
        self._fail = fail.split(":", 1)[1] if fail and int(fail.split(":", 1)[0]) == self.rank else None
        self._stage("dlopen")
        self._stage("unique-id TCP")
        ident = rccl._exchange_id(self.rank, self.world, addr, port, lambda: os.urandom(rccl.ID_BYTES),
                                  timeout=float(env.get("DGCNN_RCCL_TIMEOUT", "90")))
        self._stage("ncclCommInitRank")
        self.ident = ident
        # data plane: a star through rank 0 on port + 1 (peers announce their rank; late peers retry like the id fetch)
        self.peers = {}
        self.up = None
        if self.world > 1:
            if self.rank == 0:
                srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
                srv.bind(("127.0.0.1" if addr == "localhost" else addr, port + 1))
                srv.listen(self.world)
                srv.settimeout(90.0)
                while len(self.peers) < self.world - 1:
                    conn, _ = srv.accept()
                    conn.settimeout(120.0)
                    r, peer_id = _recv(conn)
                    assert peer_id == ident, "a peer holds a different id: the rendezvous mixed two jobs"
                    self.peers[r] = conn
                srv.close()
            else:
                deadline = time.time() + 90.0
                while True:
                    try:
                        self.up = socket.create_connection((addr, port + 1), timeout=5.0)
                        break
                    except OSError:
                        if time.time() > deadline:
                            raise
                        time.sleep(0.1)
                self.up.settimeout(120.0)
                _send(self.up, (self.rank, ident))

        self._stage("first collective")
        self._reduce([1.0])
        self._stage("ready")

    def _stage(self, name):
        import sys
        rccl.LAST_STAGE[0] = "RCCL bring-up: " + name
        if os.environ.get("DGCNN_RCCL_TRACE", "0") not in ("0", ""):
            sys.stderr.write("[dgcnn.rccl rank %d/%d] stage: %s (stand-in)\n" % (self.rank, self.world, name))
            sys.stderr.flush()
        if self._fail == name:
            raise RuntimeError("RCCL group, rank %d of %d, stage `%s`: injected failure (stand-in communicator)"
                               % (self.rank, self.world, name))

    # ---- the surface bench.py uses -----------------------------------------------------------
    def info(self):
        """What the communicator ITSELF knows: rank 0 counts its connections, a peer asks rank 0."""
        n = self._reduce([1.0])[0]
        return {"nranks": int(round(n)), "rank": self.rank, "device": -1}

    def _reduce(self, values):
        """SUM of a list of floats / a tensor over the ranks, result on every rank."""
        n, self._ncoll = self._ncoll, self._ncoll + 1
        if self._hang_at is not None and n >= self._hang_at:
            while True:                              # "collective never completes"
                time.sleep(3600)
        if self.world == 1:
            return values
        if self.rank == 0:
            acc = values.clone() if isinstance(values, torch.Tensor) else list(values)
            for r in sorted(self.peers):
                v = _recv(self.peers[r])
                acc = acc + v if isinstance(acc, torch.Tensor) else [a + b for a, b in zip(acc, v)]
            for r in sorted(self.peers):
                _send(self.peers[r], acc)
            return acc
        _send(self.up, values)
        return _recv(self.up)

    def allreduce_sum_(self, t):
        t.copy_(self._reduce(t))
        return t

    def allreduce_sum_async(self, t):
        return self.allreduce_sum_(t)

    def wait(self):
        pass

    def barrier(self):
        self._reduce([0.0])

    def gather_scalars(self, values):
        m = torch.zeros((self.world, len(values)), dtype=torch.float64)
        m[self.rank] = torch.tensor(values, dtype=torch.float64)
        return self._reduce(m)

    def destroy(self):
        for s in list(self.peers.values()) + ([self.up] if self.up else []):
            try:
                s.close()
            except OSError:
                pass
        self.peers, self.up = {}, None
