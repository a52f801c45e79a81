"""The path's one collective on RCCL, without torch.distributed: a ctypes view of the communicator entry points of
libdgcnn_hip.so (csrc/comm.cc dlopens librccl.so) plus the rendezvous that ships rank 0's unique id to the other ranks.

Reference: in-graph towers, gradients copied to the host and averaged there (dgcnn/trainval.py:16,64-73).  Here: one process per
GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the launcher's environment, as torch.distributed.run sets
them); the 128-byte ncclUniqueId travels over a plain TCP socket on MASTER_ADDR:(MASTER_PORT + 1) (or $DGCNN_RCCL_PORT);
collectives run on a dedicated HIP stream so that a bucket can be reduced while the backward pass is still producing the next."""
from __future__ import annotations

import ctypes
import os
import socket
import sys
import time

import torch

from . import _hip as H

ID_BYTES = 128
LAST_STAGE = [None]          # the bring-up stage this process announced last (Deadline reports it when a bring-up hangs)


class Deadline(object):
    """`with Deadline(stage, seconds, on_expire):` -- a watchdog THREAD for code that can hang inside RCCL, where nothing raises:
    ncclCommInitRank waiting for a rank that never arrives, the first multi-rank collective on a fabric that does not come up.  The
    guarded calls are ctypes / torch calls that release the GIL while they block, so the thread gets to run; if the body has not
    left the block `seconds` after entering it, on_expire(stage_text, seconds) is called -- by default one line on stderr and
    os._exit(124): a hung communicator cannot be torn down in-process, and a launcher (torch.distributed.run) that sees one rank die
    stops the others.  `stage` may be a callable (evaluated at expiry: the LAST stage announced)."""

    def __init__(self, stage, seconds, on_expire=None):
        self.stage, self.seconds, self.on_expire = stage, float(seconds), on_expire
        self._done = None

    def _text(self):
        return str(self.stage() if callable(self.stage) else self.stage)

    def _run(self):
        if self._done.wait(self.seconds):
            return
        text = self._text()
        if self.on_expire is not None:
            self.on_expire(text, self.seconds)
        sys.stderr.write("[dgcnn.rccl] `%s` did not finish within %.0f s: giving up (exit 124)\n" % (text, self.seconds))
        sys.stderr.flush()
        os._exit(124)

    def __enter__(self):
        import threading
        self._done = threading.Event()
        self._thread = threading.Thread(target=self._run, name="dgcnn-deadline", daemon=True)
        self._thread.start()
        return self

    def __exit__(self, *exc):
        self._done.set()
        return False


def _exchange_id(rank, world, addr, port, make_id, timeout=90.0):
    """rank 0: draw the id, serve it to world - 1 connections; others: connect (with retries) and read it."""
    if rank == 0:
        ident = make_id()
        if world == 1:
            return ident
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind((addr if addr not in ("localhost",) else "127.0.0.1", port))
        srv.listen(world)
        srv.settimeout(timeout)
        served = 0
        try:
            for _ in range(world - 1):
                conn, _ = srv.accept()
                with conn:
                    conn.sendall(ident)
                served += 1
        except socket.timeout:
            raise H.HipError("RCCL rendezvous: rank 0 served the unique id to %d of %d ranks on %s:%d within %.0f s -- the others "
                             "never connected (not started, another MASTER_ADDR / MASTER_PORT, or a firewall)"
                             % (served, world - 1, addr, port, timeout))
        finally:
            srv.close()
        return ident
    deadline = time.time() + timeout
    last = None
    while time.time() < deadline:
        try:
            with socket.create_connection((addr, port), timeout=5.0) as s:
                buf = b""
                while len(buf) < ID_BYTES:
                    chunk = s.recv(ID_BYTES - len(buf))
                    if not chunk:
                        break
                    buf += chunk
                if len(buf) == ID_BYTES:
                    return buf
        except OSError as e:            # rank 0 not listening yet
            last = e
        time.sleep(0.1)
    raise H.HipError("RCCL rendezvous: rank %d could not fetch the unique id from %s:%d (%s)" % (rank, addr, port, last))


class Group(object):
    """One RCCL communicator for this process' current device."""

    def __init__(self, rank=None, world=None, addr=None, port=None):
        env = os.environ
        self.rank = int(env.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(env.get("WORLD_SIZE", "1")) if world is None else int(world)
        addr = env.get("MASTER_ADDR", "127.0.0.1") if addr is None else addr
        if port is None:
            port = int(env["DGCNN_RCCL_PORT"]) if "DGCNN_RCCL_PORT" in env else int(env.get("MASTER_PORT", "29500")) + 1
        if not torch.cuda.is_available():
            raise H.HipError("RCCL group needs a GPU")
        self.lib = H.load()
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.comm = None
        timeout = float(env.get("DGCNN_RCCL_TIMEOUT", "90"))
        # Four stages, each named in the error it raises (and on stderr with DGCNN_RCCL_TRACE=1: a HANG inside RCCL cannot raise --
        # profiles/scale_probe.sh reads the last stage a rank announced): dlopen -> unique-id TCP -> ncclCommInitRank -> first collective
        try:
            self._stage("dlopen", "librccl via %s" % (env.get("DGCNN_RCCL_LIB") or "the loader's search path"))
            self._check(self.lib.dgcnn_comm_available(), "dgcnn_comm_available")

            def make_id():
                buf = ctypes.create_string_buffer(ID_BYTES)
                self._check(self.lib.dgcnn_comm_unique_id(buf), "dgcnn_comm_unique_id")
                return buf.raw
            self._stage("unique-id TCP", "%s:%d, timeout %.0f s" % (addr, int(port), timeout))
            ident = _exchange_id(self.rank, self.world, addr, int(port), make_id, timeout=timeout)
            self._stage("ncclCommInitRank", "device %d, HSA_ENABLE_IPC_MODE_LEGACY=%s, NCCL_SOCKET_IFNAME=%s"
                        % (self.device.index, env.get("HSA_ENABLE_IPC_MODE_LEGACY"), env.get("NCCL_SOCKET_IFNAME")))
            comm = ctypes.c_void_p()
            self._check(self.lib.dgcnn_comm_init(self.world, self.rank, ident, ctypes.byref(comm)), "dgcnn_comm_init")
            self.comm = comm
            self.stream = torch.cuda.Stream(device=self.device)          # collectives run here
            # what RCCL itself reports for this communicator must match what the launcher said -- checked once, loudly
            info = self.info()
            if (info["nranks"], info["rank"]) != (self.world, self.rank):
                raise H.HipError("RCCL communicator reports rank %d of %d, the launcher said rank %d of %d"
                                 % (info["rank"], info["nranks"], self.rank, self.world))
            self._stage("first collective", "all-reduce of one float over %d ranks" % self.world)
            one = torch.ones(1, dtype=torch.float32, device=self.device)
            self.allreduce_sum_(one)
            torch.cuda.synchronize()
            if float(one) != float(self.world):
                raise H.HipError("the first all-reduce over %d ranks returned %r" % (self.world, float(one)))
            self._stage("ready", "")
        except Exception as e:
            raise H.HipError("RCCL group, rank %d of %d, stage `%s`: %s" % (self.rank, self.world, self.stage, e))

    def _stage(self, name, detail):
        self.stage = name
        LAST_STAGE[0] = "RCCL bring-up: " + name
        if os.environ.get("DGCNN_RCCL_TRACE", "0") not in ("0", ""):
            sys.stderr.write("[dgcnn.rccl rank %d/%d] stage: %s%s\n" % (self.rank, self.world, name, " (%s)" % detail if detail else ""))
            sys.stderr.flush()

    def info(self):
        """{'nranks', 'rank', 'device'} from ncclCommCount / ncclCommUserRank / ncclCommCuDevice (dgcnn_comm_info)."""
        n, r, d = ctypes.c_int(-1), ctypes.c_int(-1), ctypes.c_int(-1)
        self._check(self.lib.dgcnn_comm_info(self.comm, ctypes.byref(n), ctypes.byref(r), ctypes.byref(d)), "dgcnn_comm_info")
        return {"nranks": int(n.value), "rank": int(r.value), "device": int(d.value)}

    def _check(self, rc, what):
        if rc != 0:
            raise H.HipError("%s failed (%d): %s" % (what, rc, self.lib.dgcnn_last_error().decode()))

    # ---- collectives (in place, on self.stream, ordered after everything issued so far on the current stream) ----
    def _on_comm_stream(self, fn, tensor):
        cur = torch.cuda.current_stream()
        H.stream_wait(self.stream, cur)              # (through the library: a launch plan being recorded sees the fork)
        tensor.record_stream(self.stream)
        fn(self.stream.cuda_stream)

    def allreduce_sum_async(self, t):
        """SUM over the ranks, in place; returns immediately.  `wait()` orders the current stream after it."""
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        self._on_comm_stream(lambda st: self._check(self.lib.dgcnn_allreduce_f32(t.data_ptr(), t.numel(), self.comm, st),
                                                    "dgcnn_allreduce_f32"), t)

    def broadcast_async(self, t, root=0):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        self._on_comm_stream(lambda st: self._check(self.lib.dgcnn_broadcast_f32(t.data_ptr(), t.numel(), int(root), self.comm, st),
                                                    "dgcnn_broadcast_f32"), t)

    def wait(self):
        H.stream_wait(torch.cuda.current_stream(), self.stream)

    def allreduce_sum_(self, t):
        self.allreduce_sum_async(t)
        self.wait()
        return t

    def broadcast_(self, t, root=0):
        self.broadcast_async(t, root)
        self.wait()
        return t

    def barrier(self):
        one = torch.ones(1, dtype=torch.float32, device=self.device)
        self.allreduce_sum_(one)
        torch.cuda.synchronize()

    def gather_scalars(self, values):
        """Every rank's list of floats on every rank: (world, len(values)) via one SUM all-reduce of a one-hot-by-rank matrix."""
        m = torch.zeros((self.world, len(values)), dtype=torch.float32, device=self.device)
        m[self.rank] = torch.tensor(values, dtype=torch.float32, device=self.device)
        self.allreduce_sum_(m)
        return m.cpu()

    def destroy(self):
        if getattr(self, "comm", None):
            torch.cuda.synchronize()
            self.lib.dgcnn_comm_destroy(self.comm)
            self.comm = None
