"""Single-node process rendezvous over a Unix-domain socket: barrier, all-gather of small byte strings.

The reference's only parallelism is one chain per worker process, coordinated through `multiprocess`
queues (reference samplers.py:546-565, 668-772).  Here one process per GPU owns a contiguous shard of the
chains and the processes need exactly three host-side exchanges: a barrier around a timed region, the
maximum / sum of a few floats, and the 128-byte RCCL unique id handed from rank 0 to everyone before the
xGMI communicator exists.  This module supplies those with the standard library only (no torch):
rank 0 serves a tiny star topology on an AF_UNIX socket, every collective is one request / one reply.

    rdzv = Rendezvous.from_env()            # RANK / WORLD_SIZE / MICI_AMD_RDZV (or MASTER_PORT) from the env
    rdzv.barrier()
    parts = rdzv.allgather(b"...")          # list of world_size byte strings, rank order
    uid = rdzv.broadcast(uid_bytes if rdzv.rank == 0 else None)
    rdzv.close()
"""

from __future__ import annotations

import os
import pickle
import socket
import struct
import tempfile
import time

import numpy as np

_HDR = struct.Struct("!I")


def _send(sock, payload):
    sock.sendall(_HDR.pack(len(payload)) + payload)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        chunk = sock.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("rendezvous peer closed the connection")
        buf += chunk
    return bytes(buf)


def _recv(sock):
    (n,) = _HDR.unpack(_recv_exact(sock, _HDR.size))
    return _recv_exact(sock, n)


def default_path():
    """Socket path shared by the ranks of one job: MICI_AMD_RDZV if set, else derived from what a
    `torch.distributed.run` launch exports (MASTER_PORT + run id), else from the parent pid."""
    p = os.environ.get("MICI_AMD_RDZV")
    if p:
        return p
    tag = "{}_{}".format(os.environ.get("MASTER_PORT", "p%d" % os.getppid()),
                         os.environ.get("TORCHELASTIC_RUN_ID", "none"))
    return os.path.join(tempfile.gettempdir(), "mici_amd_rdzv_%d_%s.sock" % (os.getuid(), tag))


class Rendezvous:
    """World of ``world_size`` processes on one node.  ``world_size == 1`` needs no socket."""

    def __init__(self, rank, world_size, path=None, timeout=120.0):
        self.rank, self.world = int(rank), int(world_size)
        self.timeout = float(timeout)
        self.path = path or default_path()
        self._server = None
        self._peers = {}       # rank 0: rank -> socket
        self._sock = None      # rank > 0: socket to rank 0
        if self.world <= 1:
            return
        if self.rank == 0:
            try:
                os.unlink(self.path)
            except FileNotFoundError:
                pass
            srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            srv.bind(self.path)
            srv.listen(self.world)
            srv.settimeout(self.timeout)
            self._server = srv
            deadline = time.monotonic() + self.timeout
            while len(self._peers) < self.world - 1:
                if time.monotonic() > deadline:
                    raise TimeoutError(f"rendezvous: only {len(self._peers) + 1} of {self.world} ranks arrived")
                conn, _ = srv.accept()
                conn.settimeout(self.timeout)
                r = pickle.loads(_recv(conn))
                self._peers[int(r)] = conn
        else:
            deadline = time.monotonic() + self.timeout
            while True:
                s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
                try:
                    s.connect(self.path)
                    break
                except (FileNotFoundError, ConnectionRefusedError):
                    s.close()
                    if time.monotonic() > deadline:
                        raise TimeoutError(f"rendezvous: rank 0 never listened on {self.path}")
                    time.sleep(0.02)
            s.settimeout(self.timeout)
            _send(s, pickle.dumps(self.rank))
            self._sock = s

    @classmethod
    def from_env(cls, timeout=120.0):
        return cls(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), None, timeout)

    # -- collectives (every rank must call them in the same order) ---------------------------------------
    def allgather(self, payload: bytes):
        """Every rank contributes a byte string; every rank receives the list in rank order."""
        if self.world <= 1:
            return [payload]
        if self.rank == 0:
            parts = [None] * self.world
            parts[0] = payload
            for r, conn in self._peers.items():
                parts[r] = _recv(conn)
            blob = pickle.dumps(parts)
            for conn in self._peers.values():
                _send(conn, blob)
            return parts
        _send(self._sock, payload)
        return pickle.loads(_recv(self._sock))

    def barrier(self):
        self.allgather(b"")

    def broadcast(self, payload, src=0):
        parts = self.allgather(payload if self.rank == src else b"")
        return parts[src]

    def allgather_array(self, a):
        """Stack equal-shape NumPy arrays of all ranks along a new leading axis."""
        a = np.ascontiguousarray(a)
        parts = self.allgather(pickle.dumps(a, protocol=pickle.HIGHEST_PROTOCOL))
        return np.stack([pickle.loads(p) for p in parts], axis=0)

    def reduce_max(self, x: float) -> float:
        return float(max(struct.unpack("!d", p)[0] for p in self.allgather(struct.pack("!d", float(x)))))

    def reduce_sum(self, x: float) -> float:
        return float(sum(struct.unpack("!d", p)[0] for p in self.allgather(struct.pack("!d", float(x)))))

    def close(self):
        for conn in self._peers.values():
            try:
                conn.close()
            except OSError:
                pass
        self._peers = {}
        if self._sock is not None:
            try:
                self._sock.close()
            except OSError:
                pass
            self._sock = None
        if self._server is not None:
            try:
                self._server.close()
            finally:
                self._server = None
                try:
                    os.unlink(self.path)
                except OSError:
                    pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def spawn_ranks(argv, world_size, env_extra=None, timeout=None):
    """Start ``world_size`` copies of ``argv`` (one per GPU: RANK = LOCAL_RANK = 0..N-1, WORLD_SIZE = N, a
    fresh rendezvous socket in MICI_AMD_RDZV), wait for all of them and return their exit codes.  The
    children inherit stdout / stderr.  If one rank fails the others are terminated."""
    import subprocess

    tmp = tempfile.mkdtemp(prefix="mici_amd_rdzv_")
    path = os.path.join(tmp, "rdzv.sock")
    procs = []
    for r in range(world_size):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(world_size),
                    "MICI_AMD_RDZV": path, "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
        env.update(env_extra or {})
        procs.append(subprocess.Popen(argv, env=env))
    codes = [None] * world_size
    t0 = time.monotonic()
    try:
        while any(c is None for c in codes):
            for i, p in enumerate(procs):
                if codes[i] is None:
                    c = p.poll()
                    if c is not None:
                        codes[i] = c
                        if c != 0:  # one rank died: the others would wait in a collective for ever
                            for q in procs:
                                if q.poll() is None:
                                    q.terminate()
            if timeout is not None and time.monotonic() - t0 > timeout:
                for q in procs:
                    if q.poll() is None:
                        q.terminate()
                raise TimeoutError(f"ranks still running after {timeout} s")
            time.sleep(0.05)
    finally:
        for q in procs:
            if q.poll() is None:
                q.kill()
        try:
            os.unlink(path)
        except OSError:
            pass
        try:
            os.rmdir(tmp)
        except OSError:
            pass
    return codes
