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
import socket
import stat
import struct
import tempfile
import time

import numpy as np

# Wire format (no pickle anywhere: a frame is never more than bytes the receiver asked for).
#   frame      = u64 length (network order) + payload
#   hello      = b"MICI" + u32 rank + u32 world + u32 len(tag) + tag         rank > 0 -> rank 0, once
#   gather     = every rank sends one frame; rank 0 answers with u32 count + count x (u64 length + bytes)
#   array      = u8 len(dtype.str) + dtype.str + u8 ndim + ndim x u64 + the C-contiguous bytes
_HDR = struct.Struct("!Q")
_MAX_FRAME = 1 << 40
_MAGIC = b"MICI"


def _send(sock, payload):
    sock.sendall(_HDR.pack(len(payload)))
    sock.sendall(payload)


def _recv_exact(sock, n):
    buf = bytearray(n)
    view, got = memoryview(buf), 0
    while got < n:
        k = sock.recv_into(view[got:], min(n - got, 1 << 24))
        if not k:
            raise ConnectionError("rendezvous peer closed the connection")
        got += k
    return bytes(buf)


def _recv(sock):
    (n,) = _HDR.unpack(_recv_exact(sock, _HDR.size))
    if n > _MAX_FRAME:
        raise ConnectionError(f"rendezvous: implausible frame length {n}")
    return _recv_exact(sock, n)


def _pack_parts(parts):
    out = [struct.pack("!I", len(parts))]
    for p in parts:
        out.append(_HDR.pack(len(p)))
        out.append(p)
    return b"".join(out)


def _unpack_parts(blob):
    (count,) = struct.unpack_from("!I", blob, 0)
    off, parts = 4, []
    for _ in range(count):
        (n,) = _HDR.unpack_from(blob, off)
        off += _HDR.size
        if off + n > len(blob):
            raise ConnectionError("rendezvous: truncated gather reply")
        parts.append(blob[off:off + n])
        off += n
    return parts


def pack_array(a):
    a = np.asarray(a)
    if not a.flags.c_contiguous:
        a = np.ascontiguousarray(a)
    if a.dtype.hasobject:
        raise TypeError("rendezvous ships plain numeric arrays only")
    ds = a.dtype.str.encode()
    head = struct.pack("!B", len(ds)) + ds + struct.pack("!B", a.ndim) + struct.pack("!%dQ" % a.ndim, *a.shape)
    return head + a.tobytes()


def unpack_array(blob):
    n = blob[0]
    dtype = np.dtype(blob[1:1 + n].decode())
    if dtype.hasobject:
        raise TypeError("rendezvous ships plain numeric arrays only")
    off = 1 + n
    ndim = blob[off]
    shape = struct.unpack_from("!%dQ" % ndim, blob, off + 1)
    off += 1 + 8 * ndim
    return np.frombuffer(blob, dtype=dtype, count=int(np.prod(shape, dtype=np.int64)), offset=off).reshape(shape).copy()


def private_dir():
    """A directory only this user can enter (0700, owned by us) under $XDG_RUNTIME_DIR or the temp dir: the socket of
    a job lives inside it, so another user can neither pre-bind its name nor connect to it."""
    base = os.environ.get("XDG_RUNTIME_DIR") or tempfile.gettempdir()
    d = os.path.join(base, "mici_amd_%d" % os.getuid())
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise PermissionError(f"rendezvous directory {d} is not a private directory of uid {os.getuid()}")
    return d


def job_tag():
    """What the ranks of ONE job agree on without talking: MICI_AMD_RDZV_TAG, else what a `torch.distributed.run`
    launch exports (MASTER_PORT + run id), else the parent pid (ranks started by hand from one shell)."""
    t = os.environ.get("MICI_AMD_RDZV_TAG")
    if t:
        return t
    return "{}_{}".format(os.environ.get("MASTER_PORT", "p%d" % os.getppid()),
                          os.environ.get("TORCHELASTIC_RUN_ID", "none"))


def default_path():
    """Socket path shared by the ranks of one job: MICI_AMD_RDZV if set (spawn_ranks sets it to a fresh private
    directory), else <private dir>/<job tag>.sock."""
    p = os.environ.get("MICI_AMD_RDZV")
    if p:
        return p
    safe = "".join(c if c.isalnum() or c in "-_." else "_" for c in job_tag())
    return os.path.join(private_dir(), "rdzv_%s.sock" % safe)


class Rendezvous:
    """World of ``world_size`` processes on one node.  ``world_size == 1`` needs no socket.

    ``timeout`` bounds how long the ranks wait for each other to ARRIVE (connect / accept / hello); collectives then
    wait ``collective_timeout`` seconds (default: for ever - a rank may legitimately reach a barrier or a trace gather
    long after its peers: load imbalance, a hipRTC compile, a long sampling run).  ``allgather_array`` / ``gather_host``
    ship whole shards as single frames and rank 0 holds every part in memory at once: fine for the per-collection
    trace shards of this path (MBs), not a transport for hundreds of GBs."""

    def __init__(self, rank, world_size, path=None, timeout=120.0, collective_timeout=None, tag=None):
        self.rank, self.world = int(rank), int(world_size)
        self.timeout = float(timeout)
        self.collective_timeout = collective_timeout
        self._server = None
        self._peers = {}       # rank 0: rank -> socket
        self._sock = None      # rank > 0: socket to rank 0
        if self.world <= 1:
            self.path = path
            return
        if not 0 <= self.rank < self.world:
            raise ValueError(f"rank {self.rank} outside world of {self.world}")
        self.path = path or default_path()
        self.tag = (tag if tag is not None else job_tag()).encode()
        if self.rank == 0:
            try:
                os.unlink(self.path)
            except FileNotFoundError:
                pass
            srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            old = os.umask(0o177)  # the socket itself: owner only
            try:
                srv.bind(self.path)
            finally:
                os.umask(old)
            srv.listen(self.world)
            self._server = srv
            deadline = time.monotonic() + self.timeout
            while len(self._peers) < self.world - 1:
                left = deadline - time.monotonic()
                if left <= 0:
                    raise TimeoutError(f"rendezvous: only {len(self._peers) + 1} of {self.world} ranks arrived")
                srv.settimeout(left)
                try:
                    conn, _ = srv.accept()
                except socket.timeout:
                    continue
                conn.settimeout(max(1.0, min(10.0, left)))
                try:
                    r = self._read_hello(conn)
                except (ConnectionError, ValueError, socket.timeout, struct.error):
                    conn.close()  # not one of this job's ranks (stale rank of another job, a stray client)
                    continue
                self._peers[r] = conn
            for conn in self._peers.values():
                conn.settimeout(self.collective_timeout)
            for conn in self._peers.values():  # every rank learns that the world is complete and consistent
                _send(conn, _MAGIC)
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
            s.settimeout(max(1.0, deadline - time.monotonic()) + 5.0)
            _send(s, _MAGIC + struct.pack("!III", self.rank, self.world, len(self.tag)) + self.tag)
            try:
                ack = _recv(s)
            except (ConnectionError, socket.timeout) as e:
                s.close()
                raise ConnectionError(f"rendezvous: rank 0 on {self.path} did not accept rank {self.rank} of job "
                                      f"{self.tag.decode()!r} (another job's socket, a duplicate rank, or not all "
                                      f"ranks arrived): {e}") from e
            if ack != _MAGIC:
                s.close()
                raise ConnectionError("rendezvous: unexpected acknowledgement")
            s.settimeout(self.collective_timeout)
            self._sock = s

    def _read_hello(self, conn):
        msg = _recv(conn)
        if len(msg) < 16 or msg[:4] != _MAGIC:
            raise ValueError("not a rendezvous hello")
        r, w, n = struct.unpack_from("!III", msg, 4)
        if w != self.world or msg[16:16 + n] != self.tag or len(msg) != 16 + n:
            raise ValueError("hello from another job")
        if not 0 < r < self.world or r in self._peers:
            raise ValueError("rank out of range or already present")
        return r

    @classmethod
    def from_env(cls, timeout=120.0, collective_timeout=None):
        return cls(int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), None, timeout,
                   collective_timeout)

    # -- collectives (every rank must call them in the same order) ---------------------------------------
    def allgather(self, payload: bytes):
        """Every rank contributes a byte string; every rank receives the list in rank order."""
        payload = bytes(payload)
        if self.world <= 1:
            return [payload]
        if self.rank == 0:
            parts = [None] * self.world
            parts[0] = payload
            for r, conn in self._peers.items():
                parts[r] = _recv(conn)
            blob = _pack_parts(parts)
            for conn in self._peers.values():
                _send(conn, blob)
            return parts
        _send(self._sock, payload)
        return _unpack_parts(_recv(self._sock))

    def barrier(self):
        self.allgather(b"")

    def broadcast(self, payload, src=0):
        parts = self.allgather(payload if self.rank == src else b"")
        return parts[src]

    def allgather_array(self, a):
        """Stack equal-shape numeric NumPy arrays of all ranks along a new leading axis."""
        parts = self.allgather(pack_array(a))
        return np.stack([unpack_array(p) for p in parts], axis=0)

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

    tmp = tempfile.mkdtemp(prefix="mici_amd_rdzv_")  # 0700, ours: a fresh private socket directory per job
    path = os.path.join(tmp, "rdzv.sock")
    tag = "spawn_%d_%s" % (os.getpid(), os.path.basename(tmp))
    procs = []
    for r in range(world_size):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(world_size),
                    "MICI_AMD_RDZV": path, "MICI_AMD_RDZV_TAG": tag, "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
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
