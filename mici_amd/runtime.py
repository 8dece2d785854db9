"""Thin Python owners of the C-ABI handles: Context (device + stream), DeviceModel, DeviceBatch.

Host arrays are NumPy-owned and only borrowed for the duration of a call; device buffers are owned
by the handles and released in ``close()`` / ``__del__`` (ownership)."""

from __future__ import annotations

import ctypes as C
import os
import threading
import weakref

import numpy as np

from . import _ffi
from .errors import DeviceError


def _dptr(a):
    return a.ctypes.data_as(_ffi.c_double_p)


class Context:
    """One HIP stream on one device.  Not thread safe; distinct contexts are independent."""

    def __init__(self, device=None, dev=False):
        # dev=True: the developer build of the library (extra test / micro-benchmark entry points, mici_amd/build.py)
        self._lib = _ffi.load(dev=dev)
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        count = C.c_int(0)
        _ffi.check(self._lib.mm_device_count(C.byref(count)), None, "mm_device_count")
        if count.value == 0:
            raise DeviceError("no HIP device visible; mici_amd needs an MI355X (no CPU fallback)")
        device = device % count.value
        h = C.c_void_p()
        _ffi.check(self._lib.mm_ctx_create(device, C.byref(h)), None, "mm_ctx_create")
        self.handle = h
        self.device = device
        self._caches = weakref.WeakSet()  # ContextCache objects holding device objects that live on this context

    def sync(self):
        _ffi.check(self._lib.mm_ctx_sync(self.handle), self.handle, "mm_ctx_sync")

    def record(self, slot):
        _ffi.check(self._lib.mm_ctx_record(self.handle, slot), self.handle, "mm_ctx_record")

    def elapsed_ms(self, a, b):
        ms = C.c_double(0.0)
        _ffi.check(self._lib.mm_ctx_elapsed_ms(self.handle, a, b, C.byref(ms)), self.handle,
                   "mm_ctx_elapsed_ms")
        return ms.value

    def close(self):
        if getattr(self, "handle", None):
            for cache in list(getattr(self, "_caches", ())):  # device objects cached elsewhere die with their context
                cache.evict(self)
            self._lib.mm_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ContextCache:
    """Device objects (models, single-state batches, proposal batches) cached per context by a long-lived host object
    - a system shared by every worker thread's integrator copy, say.  Thread safe, and an entry goes when its context
    is closed (``Context.close`` evicts), so per-thread default contexts of short-lived worker threads do not pile
    up streams, pinned staging buffers and device models behind a shared system."""

    def __init__(self):
        self._d = {}
        self._lock = threading.Lock()

    def get(self, ctx, *extra):
        with self._lock:
            obj = self._d.get((id(ctx),) + extra)
        if obj is None or getattr(obj, "handle", None) is None or obj.ctx is not ctx:
            return None
        return obj

    def put(self, ctx, obj, *extra):
        with self._lock:
            self._d[(id(ctx),) + extra] = obj
        ctx._caches.add(self)
        return obj

    def evict(self, ctx):
        with self._lock:
            gone = [self._d.pop(k) for k in [k for k in self._d if k[0] == id(ctx)]]
        for obj in gone:
            _close_cached(obj)

    def clear(self):
        with self._lock:
            gone, self._d = list(self._d.values()), {}
        for obj in gone:
            _close_cached(obj)

    def __len__(self):
        return len(self._d)

    def __deepcopy__(self, memo):  # device handles never travel: a copy starts empty
        return ContextCache()

    def __reduce__(self):
        return (ContextCache, ())


def _close_cached(obj):
    try:
        if isinstance(obj, DeviceBatch):
            obj.close(force=True)
        else:
            obj.close()
    except Exception:
        pass


class _ThreadContexts(dict):
    """The default contexts of one host thread; closed (and evicted from every cache) when the thread ends."""

    def __del__(self):
        for ctx in list(self.values()):
            try:
                ctx.close()
            except Exception:
                pass


_default = threading.local()


def default_context(device=None):
    """Default context of the CALLING THREAD for a device (created lazily).  A context owns one stream and one
    pinned staging buffer and is not thread safe (include/mici_amd.h), while ctypes releases the GIL during calls:
    the reference's thread-pool chain parallelism (per-chain deep copies of the transitions, each calling
    ``sample``) must therefore never share one - so the default is one context per (host thread, device)."""
    key = device if device is not None else int(os.environ.get("LOCAL_RANK", "0"))
    table = getattr(_default, "ctxs", None)
    if table is None:
        table = _default.ctxs = _ThreadContexts()
    ctx = table.get(key)
    if ctx is None or ctx.handle is None:
        ctx = table[key] = Context(key)
    return ctx


class DeviceModel:
    """Device copy of a built-in model (target + fixed metric + Riemannian metric + constraint)."""

    def __init__(self, ctx, dim, target, metric_kind=0, metric=None, rmetric=0, rmetric_params=None,
                 constr=0, constr_params=None, gaussian_split=False, dens_wrt_ambient=False, n_constr=0,
                 constr_source=None, rmetric_source=None):
        self.ctx = ctx
        self._lib = ctx._lib
        self._keep = []

        def arr(a):
            a = np.ascontiguousarray(np.zeros(0) if a is None else a, dtype=np.float64).ravel()
            self._keep.append(a)
            return (_dptr(a) if a.size else None), a.size

        d = _ffi.ModelDesc()
        d.dim = dim
        d.target = target.tid
        d.target_params, d.n_target_params = arr(target.params)
        d.metric_kind = metric_kind
        d.gaussian_split = int(bool(gaussian_split))
        d.metric, d.n_metric = arr(metric)
        d.rmetric = rmetric
        d.rmetric_params, d.n_rmetric_params = arr(rmetric_params)
        d.constr = constr
        d.n_constr = int(n_constr)
        d.dens_wrt_ambient = int(bool(dens_wrt_ambient))
        d.constr_params, d.n_constr_params = arr(constr_params)
        h = C.c_void_p()
        # user-defined target / metric / constraint: ONE source text, compiled by hipRTC inside the library
        sources = [t for t in (getattr(target, "source", None), rmetric_source, constr_source) if t is not None]
        source = "\n".join(sources) if sources else None
        if source is not None:
            _ffi.check(self._lib.mm_model_create_from_source(ctx.handle, C.byref(d), source.encode(), C.byref(h)),
                       ctx.handle, "mm_model_create_from_source")
        else:
            _ffi.check(self._lib.mm_model_create(ctx.handle, C.byref(d), C.byref(h)), ctx.handle,
                       "mm_model_create")
        self.handle = h
        self.dim = dim
        self._keep = []

    def close(self):
        if getattr(self, "handle", None) and self.ctx.handle:
            self._lib.mm_model_destroy(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceBatch:
    """N chains resident in HBM: pos[N,D], mom[N,D] (fp64, row-major), dir[N] (int8)."""

    def __init__(self, ctx, n_chains, dim, mapped=False):
        """``mapped``: keep the chain state in pinned host memory the kernels access in place (small, long-lived
        batches: the reused single-state buffers behind ``Integrator.step`` / ``System.h``)."""
        self.ctx = ctx
        self._lib = ctx._lib
        h = C.c_void_p()
        alloc = self._lib.mm_state_alloc_mapped if mapped else self._lib.mm_state_alloc
        _ffi.check(alloc(ctx.handle, int(n_chains), int(dim), C.byref(h)), ctx.handle, "mm_state_alloc")
        self.handle = h
        self.n_chains = int(n_chains)
        self.dim = int(dim)

    @staticmethod
    def _mat(a, n, d, name):
        if a is None:
            return None
        a = np.ascontiguousarray(a, dtype=np.float64)
        if a.shape != (n, d):
            raise ValueError(f"{name} must have shape ({n}, {d}), got {a.shape}")
        return a

    def upload(self, pos=None, mom=None, dir=None):  # noqa: A002
        pos = self._mat(pos, self.n_chains, self.dim, "pos")
        mom = self._mat(mom, self.n_chains, self.dim, "mom")
        if dir is not None:
            dir = np.ascontiguousarray(np.broadcast_to(np.asarray(dir), (self.n_chains,)),  # noqa: A001
                                       dtype=np.int8)
            if not np.all(np.abs(dir) == 1):
                raise ValueError("dir entries must be +1 or -1")
        _ffi.check(self._lib.mm_state_upload(
            self.handle, None if pos is None else _dptr(pos), None if mom is None else _dptr(mom),
            None if dir is None else dir.ctypes.data_as(_ffi.c_int8_p)), self.ctx.handle,
            "mm_state_upload")

    def download(self):
        pos = np.empty((self.n_chains, self.dim))
        mom = np.empty((self.n_chains, self.dim))
        dir_ = np.empty(self.n_chains, dtype=np.int8)
        _ffi.check(self._lib.mm_state_download(self.handle, _dptr(pos), _dptr(mom),
                                               dir_.ctypes.data_as(_ffi.c_int8_p)),
                   self.ctx.handle, "mm_state_download")
        return pos, mom, dir_

    def download_all(self):
        """(pos, mom, dir, status, n_done) in one transfer."""
        pos = np.empty((self.n_chains, self.dim))
        mom = np.empty((self.n_chains, self.dim))
        dir_ = np.empty(self.n_chains, dtype=np.int8)
        status = np.zeros(self.n_chains, dtype=np.int32)
        n_done = np.zeros(self.n_chains, dtype=np.int32)
        _ffi.check(self._lib.mm_state_download_all(
            self.handle, _dptr(pos), _dptr(mom), dir_.ctypes.data_as(_ffi.c_int8_p),
            status.ctypes.data_as(_ffi.c_int32_p), n_done.ctypes.data_as(_ffi.c_int32_p)),
            self.ctx.handle, "mm_state_download_all")
        return pos, mom, dir_, status, n_done

    def download_status(self):
        status = np.zeros(self.n_chains, dtype=np.int32)
        n_done = np.zeros(self.n_chains, dtype=np.int32)
        _ffi.check(self._lib.mm_state_download_status(
            self.handle, status.ctypes.data_as(_ffi.c_int32_p),
            n_done.ctypes.data_as(_ffi.c_int32_p)), self.ctx.handle, "mm_state_download_status")
        return status, n_done

    def set_step_scale(self, scale):
        """Per-chain step-size factors [N] (``None`` removes them): every integrator call on this batch uses
        ``step_size * scale[chain]``."""
        if scale is None:
            ptr = None
        else:
            scale = np.ascontiguousarray(np.broadcast_to(np.asarray(scale, dtype=np.float64), (self.n_chains,)))
            ptr = _dptr(scale)
        _ffi.check(self._lib.mm_state_set_step_scale(self.handle, ptr), self.ctx.handle,
                   "mm_state_set_step_scale")

    def set_chain_steps(self, steps):
        """Per-chain trajectory lengths [N] (``None`` removes them): an integrator call with ``n_steps`` advances
        chain i by ``min(n_steps, steps[i])`` steps."""
        if steps is None:
            ptr = None
        else:
            steps = np.ascontiguousarray(np.broadcast_to(np.asarray(steps, dtype=np.int32), (self.n_chains,)))
            ptr = steps.ctypes.data_as(_ffi.c_int32_p)
        _ffi.check(self._lib.mm_state_set_chain_steps(self.handle, ptr), self.ctx.handle,
                   "mm_state_set_chain_steps")

    def mapped_views(self):
        """NumPy views (pos[N, D], mom[N, D], dir[N], status[N], n_done[N]) ONTO the pinned host memory of a mapped
        batch: the single-state path writes its inputs there, launches, synchronises and reads the results in place.
        Only to be touched while no launch on the batch is in flight."""
        v = getattr(self, "_views", None)
        if v is None:
            ptrs = [C.c_void_p() for _ in range(5)]
            _ffi.check(self._lib.mm_state_mapped_ptrs(self.handle, *[C.byref(p) for p in ptrs]), self.ctx.handle,
                       "mm_state_mapped_ptrs")
            n, d = self.n_chains, self.dim

            def view(ptr, ctype, shape):
                return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=shape)

            v = self._views = (view(ptrs[0], C.c_double, (n, d)), view(ptrs[1], C.c_double, (n, d)),
                               view(ptrs[2], C.c_int8, (n,)), view(ptrs[3], C.c_int32, (n,)),
                               view(ptrs[4], C.c_int32, (n,)))
        return v

    def download_errors(self, clear=True):
        """Sticky error word of the device-resident transitions run on this batch: uint32 [N], bit k set when a
        proposal of the chain ended with status k since the last clearing read (include/mici_amd.h)."""
        out = np.zeros(self.n_chains, dtype=np.uint32)
        _ffi.check(self._lib.mm_state_download_errors(self.handle, out.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                      1 if clear else 0), self.ctx.handle, "mm_state_download_errors")
        return out

    def set_rng(self, seed, chain_offset=0):
        """Device-side random draws for this batch (include/mici_amd.h, mm_state_set_rng): chain i draws the stream
        of global chain ``chain_offset + i`` of job ``seed``."""
        _ffi.check(self._lib.mm_state_set_rng(self.handle, int(seed) & (2**64 - 1), int(chain_offset)),
                   self.ctx.handle, "mm_state_set_rng")

    def rng_draws(self, transition, lo=None, hi=None):
        """(z[N, D], u[N], steps[N] or None) of transition number ``transition`` - what the device-draw entry points
        consume; for reproducibility checks."""
        z = np.empty((self.n_chains, self.dim))
        u = np.empty(self.n_chains)
        steps = np.empty(self.n_chains, dtype=np.int32) if lo is not None else None
        _ffi.check(self._lib.mm_rng_draws(self.handle, int(transition), _dptr(z), _dptr(u),
                                          None if steps is None else steps.ctypes.data_as(_ffi.c_int32_p),
                                          0 if lo is None else int(lo), 1 if hi is None else int(hi)),
                   self.ctx.handle, "mm_rng_draws")
        return z, u, steps

    def device_ptrs(self):
        p, m, d = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _ffi.check(self._lib.mm_state_device_ptrs(self.handle, C.byref(p), C.byref(m), C.byref(d)),
                   self.ctx.handle, "mm_state_device_ptrs")
        return p.value, m.value, d.value

    keep = False  # a cached batch ignores close() and is freed with its owner

    def close(self, force=False):
        if self.keep and not force:
            return
        if getattr(self, "handle", None) and self.ctx.handle:
            self._lib.mm_state_free(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.close(force=True)
        except Exception:
            pass
