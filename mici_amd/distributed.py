"""Multi-GPU: shard independent chains across ranks, gather traces once per collection.

The reference's only parallelism is data-parallel over chains: one chain per worker process, work
handed out through queues and traces collected through per-chain ``.npy`` memmaps
(reference samplers.py:546-565, 668-772, 104-138).  Here each rank (one process per GPU) owns a
contiguous shard of the chains; there is no communication during integration; trace collection is
one all-gather of the shard positions - RCCL over xGMI on device buffers (``RcclTraceGather``) or,
for host arrays, the standard-library rendezvous of ``mici_amd.rendezvous`` (``gather_host``).  Any
object with ``world``, ``rank``, ``allgather_array(ndarray) -> stacked ndarray`` and
``broadcast(bytes | None) -> bytes`` serves as the group - the product itself imports no torch (a
``torch.distributed`` adapter lives with the gloo test, tests/test_distributed_cpu.py).

Shards are padded to equal length (the collective needs equal counts); padding rows are dropped
again after the gather."""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _ffi


def shard_bounds(n_chains, world_size):
    """Contiguous balanced shards: returns [(start, stop)] * world_size covering range(n_chains)."""
    base, extra = divmod(int(n_chains), int(world_size))
    bounds, start = [], 0
    for r in range(world_size):
        stop = start + base + (1 if r < extra else 0)
        bounds.append((start, stop))
        start = stop
    return bounds


def padded_shard_len(n_chains, world_size):
    return -(-int(n_chains) // int(world_size))


def take_shard(array, rank, world_size, pad=True):
    """Rows of ``array`` owned by ``rank``; padded (by repeating the last owned row, or zeros for
    an empty shard) to the common length so that every rank runs the same batch shape."""
    array = np.asarray(array)
    n = array.shape[0]
    start, stop = shard_bounds(n, world_size)[rank]
    shard = array[start:stop]
    if pad:
        want = padded_shard_len(n, world_size)
        if shard.shape[0] < want:
            fill = shard[-1:] if shard.shape[0] else np.zeros((1,) + array.shape[1:], array.dtype)
            shard = np.concatenate([shard] + [fill] * (want - shard.shape[0]), axis=0)
    return np.ascontiguousarray(shard)


def unpad_gathered(gathered, n_chains, world_size):
    """Drop the padding rows of a rank-major gather of padded shards -> global chain order."""
    want = padded_shard_len(n_chains, world_size)
    gathered = np.asarray(gathered)
    parts = []
    for r, (start, stop) in enumerate(shard_bounds(n_chains, world_size)):
        parts.append(gathered[r * want: r * want + (stop - start)])
    return np.concatenate(parts, axis=0)


def _check_group(group):
    if group is None or not all(hasattr(group, a) for a in ("world", "rank", "allgather_array", "broadcast")):
        raise TypeError("a rendezvous group is required (mici_amd.rendezvous.Rendezvous, or any object with "
                        "world / rank / allgather_array / broadcast)")
    return group


def gather_host(local, n_chains, group):
    """All-gather padded host shards and return the un-padded global array on every rank.  ``group``: a
    :class:`mici_amd.rendezvous.Rendezvous` or any object with the same four members (module docstring).
    Size limit: the rendezvous ships whole shards through one socket frame each (see its docstring)."""
    group = _check_group(group)
    local = np.ascontiguousarray(local)
    stacked = np.asarray(group.allgather_array(local))
    return unpad_gathered(stacked.reshape((-1,) + local.shape[1:]), n_chains, group.world)


def exchange_unique_id(ctx, group):
    """Rank 0 creates the RCCL unique id; its 128 bytes travel over the host rendezvous."""
    group = _check_group(group)
    blob = None
    if group.rank == 0:
        raw = (C.c_uint8 * _ffi.MM_COMM_ID_BYTES)()
        _ffi.check(ctx._lib.mm_comm_unique_id(raw), None, "mm_comm_unique_id")
        blob = bytes(raw)
    blob = group.broadcast(blob)
    return (C.c_uint8 * _ffi.MM_COMM_ID_BYTES)(*blob)


class RcclTraceGather:
    """One RCCL communicator per rank; ``gather(batch)`` all-gathers the device-resident position
    shards over xGMI and returns the global [world * n_local, D] array on the host."""

    def __init__(self, ctx, rank, world_size, unique_id):
        self.ctx, self.rank, self.world = ctx, int(rank), int(world_size)
        h = C.c_void_p()
        _ffi.check(ctx._lib.mm_comm_create(ctx.handle, self.world, self.rank, unique_id, C.byref(h)),
                   ctx.handle, "mm_comm_create")
        self.handle = h
        self._out = None

    def ranks_seen(self):
        """(n_ranks, rank) as the RCCL communicator itself reports them (ncclCommCount / ncclCommUserRank)."""
        n, r = C.c_int32(0), C.c_int32(0)
        _ffi.check(self.ctx._lib.mm_comm_count(self.handle, C.byref(n), C.byref(r)), self.ctx.handle, "mm_comm_count")
        return int(n.value), int(r.value)

    def gather(self, batch):
        shape = (self.world * batch.n_chains, batch.dim)
        if self._out is None or self._out.shape != shape:
            self._out = np.empty(shape)
        _ffi.check(self.ctx._lib.mm_comm_allgather_pos(
            self.handle, batch.handle, self._out.ctypes.data_as(_ffi.c_double_p)),
            self.ctx.handle, "mm_comm_allgather_pos")
        return self._out

    def gather_async(self, batch, want_host=True):
        """Snapshot the shard and start the all-gather (+ optional copy to pinned host memory) on the
        communicator's own stream; the context stream is free to integrate the next trajectory."""
        self._shape = (self.world * batch.n_chains, batch.dim)
        _ffi.check(self.ctx._lib.mm_comm_allgather_pos_async(self.handle, batch.handle,
                                                             1 if want_host else 0),
                   self.ctx.handle, "mm_comm_allgather_pos_async")

    def wait(self, want_host=True):
        """Block until the last ``gather_async`` has landed; returns the host array if requested."""
        out = None
        if want_host:
            if self._out is None or self._out.shape != self._shape:
                self._out = np.empty(self._shape)
            out = self._out
        _ffi.check(self.ctx._lib.mm_comm_wait(
            self.handle, None if out is None else out.ctypes.data_as(_ffi.c_double_p)),
            self.ctx.handle, "mm_comm_wait")
        return out

    def close(self):
        if getattr(self, "handle", None) and self.ctx.handle:
            self.ctx._lib.mm_comm_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
