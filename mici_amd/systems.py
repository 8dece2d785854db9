"""Hamiltonian systems: the reference's ``System`` surface (mici/systems.py) over device models.

Constructor argument names follow the reference (``neg_log_dens``, ``metric``, ``metric_func``,
``constr``, ``softabs_coeff`` ...), but where the reference takes Python callables these classes take
the built-in descriptors of :mod:`mici_amd.models` - derivatives are closed forms on the device, so
``grad_neg_log_dens`` / ``vjp_metric_func`` / ``jacob_constr`` / ``backend`` must be left ``None``.

Methods used by mici.transitions / samplers (``h``, ``dh_dmom``, ``sample_momentum``;
transitions.py:141-195, 281-301, 434-473) accept a single ``ChainState`` (mici's or ours) or, via
the ``*_batch`` variants, arrays of shape [N, D]."""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _ffi, models
from .runtime import ContextCache, DeviceBatch, DeviceModel, default_context


class System:
    """Base: owns the model description; device handles are created lazily per context."""

    _kind = "euclid"

    def __init__(self, neg_log_dens, grad_neg_log_dens=None, backend=None):
        if not isinstance(neg_log_dens, models.Target):
            raise TypeError(
                "neg_log_dens must be a built-in mici_amd.models.Target descriptor: the device "
                "integrators need device-side closed-form derivatives (SURVEY.md H4)")
        if grad_neg_log_dens is not None or backend is not None:
            raise ValueError("derivatives are supplied by the device model; leave "
                             "grad_neg_log_dens / backend as None")
        self.target = neg_log_dens
        self.dim = neg_log_dens.dim
        self._device = ContextCache()
        self._one = ContextCache()  # single-chain DeviceBatch per context (h / dh_dmom / sample_momentum of one state)

    # ---- description -> device model --------------------------------------------------------
    def _model_args(self):
        return {}

    def device_model(self, ctx=None):
        ctx = ctx or default_context()
        m = self._device.get(ctx)
        if m is None:
            m = self._device.put(ctx, DeviceModel(ctx, self.dim, self.target, **self._model_args()))
        return m

    def __getstate__(self):
        d = self.__dict__.copy()
        d["_device"] = ContextCache()  # device handles are re-created lazily after unpickling
        d["_one"] = ContextCache()
        return d

    def __deepcopy__(self, memo):
        import copy
        new = object.__new__(type(self))
        for k, v in self.__dict__.items():
            new.__dict__[k] = ContextCache() if k in ("_device", "_one") else copy.deepcopy(v, memo)
        return new

    # ---- batched quantities ---------------------------------------------------------------------
    def _batch(self, pos, mom, ctx=None):
        ctx = ctx or default_context()
        pos = np.ascontiguousarray(pos, dtype=np.float64)
        if pos.ndim != 2 or pos.shape[1] != self.dim:
            raise ValueError(f"pos must have shape [N, {self.dim}]")
        if pos.shape[0] == 1:  # the single-state calls of mici.transitions: keep the device buffers
            batch = self._one.get(ctx)
            if batch is None:
                batch = self._one.put(ctx, DeviceBatch(ctx, 1, self.dim, mapped=True))
                batch.keep = True  # close() is a no-op; the buffers go when the system does
        else:
            batch = DeviceBatch(ctx, pos.shape[0], self.dim)
        batch.upload(pos, mom, None)
        return ctx, batch

    def h_batch(self, pos, mom, ctx=None):
        """``System.h`` for N chains -> [N] (NaN where the reference would raise LinAlgError)."""
        ctx, batch = self._batch(pos, mom, ctx)
        out = np.empty(batch.n_chains)
        _ffi.check(ctx._lib.mm_hamiltonian(ctx.handle, self.device_model(ctx).handle, batch.handle,
                                           out.ctypes.data_as(_ffi.c_double_p)), ctx.handle,
                   "mm_hamiltonian")
        batch.close()
        return out

    def dh_dmom_batch(self, pos, mom, ctx=None):
        ctx, batch = self._batch(pos, mom, ctx)
        out = np.empty((batch.n_chains, self.dim))
        _ffi.check(ctx._lib.mm_dh_dmom(ctx.handle, self.device_model(ctx).handle, batch.handle,
                                       out.ctypes.data_as(_ffi.c_double_p)), ctx.handle,
                   "mm_dh_dmom")
        batch.close()
        return out

    def sample_momentum_batch(self, pos, z, ctx=None):
        """mom = M^{1/2} z (then cotangent projection for constrained systems); z ~ N(0, I) is
        drawn by the caller's NumPy Generator."""
        ctx, batch = self._batch(pos, np.zeros_like(np.asarray(pos, dtype=np.float64)), ctx)
        z = np.ascontiguousarray(z, dtype=np.float64)
        if z.shape != (batch.n_chains, self.dim):
            raise ValueError("z must have the shape of pos")
        _ffi.check(ctx._lib.mm_sample_momentum(ctx.handle, self.device_model(ctx).handle,
                                               batch.handle, z.ctypes.data_as(_ffi.c_double_p)),
                   ctx.handle, "mm_sample_momentum")
        _, mom, _ = batch.download()
        batch.close()
        return mom

    # ---- single-state surface used by mici.transitions ----------------------------------------------
    # One state at a time goes through a cached buffer of pinned host memory the kernels access in place: inputs are
    # written through NumPy views, no upload call (tools/host_latency.py: the transfer calls were half of a call's time).
    def _one_state(self, pos, mom):
        ctx = default_context()
        batch = self._one.get(ctx)
        if batch is None:
            batch = self._one.put(ctx, DeviceBatch(ctx, 1, self.dim, mapped=True))
            batch.keep = True
        vq, vp, _, _, _ = batch.mapped_views()
        pos = np.asarray(pos, dtype=np.float64)
        if pos.shape != (self.dim,):
            raise ValueError(f"pos must have shape [{self.dim}]")
        if mom is not None:
            mom = np.asarray(mom, dtype=np.float64)
            if mom.shape != (self.dim,):  # (an assignment into the mapped view would broadcast a scalar silently)
                raise ValueError(f"mom must have shape [{self.dim}]")
        ctx.sync()  # nothing may still be reading the buffer
        vq[0] = pos
        if mom is not None:
            vp[0] = mom
        return ctx, batch, vp

    def h(self, state):
        ctx, batch, _ = self._one_state(state.pos, state.mom)
        out = np.empty(1)
        _ffi.check(ctx._lib.mm_hamiltonian(ctx.handle, self.device_model(ctx).handle, batch.handle,
                                           out.ctypes.data_as(_ffi.c_double_p)), ctx.handle, "mm_hamiltonian")
        val = out[0]
        if np.isnan(val) and np.all(np.isfinite(state.pos)) and np.all(np.isfinite(state.mom)):
            from .errors import LinAlgError
            raise LinAlgError("Cholesky factorisation failed.")
        return float(val)

    def dh_dmom(self, state):
        ctx, batch, _ = self._one_state(state.pos, state.mom)
        out = np.empty(self.dim)
        _ffi.check(ctx._lib.mm_dh_dmom(ctx.handle, self.device_model(ctx).handle, batch.handle,
                                       out.ctypes.data_as(_ffi.c_double_p)), ctx.handle, "mm_dh_dmom")
        return out

    def dh2_dmom(self, state):
        return self.dh_dmom(state)

    def sample_momentum(self, state, rng):
        z = np.ascontiguousarray(rng.standard_normal(np.asarray(state.pos).shape))
        ctx, batch, vp = self._one_state(state.pos, None)
        vp[0] = 0.0
        _ffi.check(ctx._lib.mm_sample_momentum(ctx.handle, self.device_model(ctx).handle, batch.handle,
                                               z.ctypes.data_as(_ffi.c_double_p)), ctx.handle, "mm_sample_momentum")
        ctx.sync()
        return vp[0].copy()


class EuclideanMetricSystem(System):
    """Fixed-metric system, h2 = p^T M^-1 p / 2 (reference systems.py:264-366)."""

    _kind = "euclid"

    def __init__(self, neg_log_dens, *, metric=None, grad_neg_log_dens=None, backend=None):
        super().__init__(neg_log_dens, grad_neg_log_dens, backend)
        if metric is None:
            self.metric_kind, self.metric = models.METRIC_IDENTITY, None
        else:
            metric = np.asarray(metric, dtype=np.float64)
            if metric.ndim == 1:
                self.metric_kind = models.METRIC_DIAG
            elif metric.ndim == 2:
                self.metric_kind = models.METRIC_DENSE
            else:
                raise ValueError("If NumPy ndarray value is used for `metric` must be either 1D "
                                 "(diagonal matrix) or 2D (dense positive definite matrix).")
            if metric.shape[0] != self.dim or metric.shape[-1] != self.dim:
                raise ValueError("metric shape does not match the target dimension")
            self.metric = metric

    def _model_args(self):
        return dict(metric_kind=self.metric_kind, metric=self.metric)

    def set_metric(self, metric):
        """Replace the fixed metric (``system.metric = ...`` of the reference's metric adapters,
        adapters.py:506, 633): ``None`` for the identity, a 1-D array for a diagonal metric, a 2-D array for a
        dense one.  Device models and cached single-state buffers are dropped and rebuilt on next use."""
        if metric is None:
            kind, metric = models.METRIC_IDENTITY, None
        else:
            metric = np.array(metric, dtype=np.float64)
            if metric.ndim not in (1, 2) or metric.shape[0] != self.dim or metric.shape[-1] != self.dim:
                raise ValueError("metric must be [D] (diagonal) or [D, D] (dense) for this system's dimension")
            kind = models.METRIC_DIAG if metric.ndim == 1 else models.METRIC_DENSE
        self.metric_kind, self.metric = kind, metric
        self._device.clear()


class GaussianEuclideanMetricSystem(EuclideanMetricSystem):
    """Euclidean-metric system whose target is a density with respect to the standard Gaussian measure
    (reference systems.py:369-474): h1 = neg_log_dens, h2 = q.q/2 + p.M^-1 p/2, and ``h2_flow`` is the
    exact rotation in the metric's eigenbasis (systems.py:464-474), so the splitting integrators only
    discretise the non-Gaussian part.  As in the reference, ``dh_dpos`` is inherited from
    EuclideanMetricSystem (systems.py:359-360) and therefore ImplicitMidpointIntegrator sees dh1_dpos only."""

    _gaussian_split = True

    def _model_args(self):
        return dict(metric_kind=self.metric_kind, metric=self.metric, gaussian_split=True)


class DenseRiemannianMetricSystem(System):
    """Position-dependent dense metric M(q) (reference systems.py:1690-1734, 1187-1402)."""

    _kind = "riemann"

    def __init__(self, neg_log_dens, metric_func, *, vjp_metric_func=None, grad_neg_log_dens=None,
                 backend=None):
        super().__init__(neg_log_dens, grad_neg_log_dens, backend)
        if not isinstance(metric_func, models.RiemannianMetric):
            raise TypeError("metric_func must be a built-in mici_amd.models.RiemannianMetric")
        if vjp_metric_func is not None:
            raise ValueError("vjp_metric_func is supplied by the device model; leave it None")
        if metric_func.dim != self.dim:
            raise ValueError("metric_func dimension does not match the target dimension")
        self.rmetric = metric_func

    def _model_args(self):
        return dict(rmetric=self.rmetric.mid, rmetric_params=self.rmetric.params,
                    rmetric_source=getattr(self.rmetric, "source", None))


class SoftAbsRiemannianMetricSystem(System):
    """SoftAbs-regularised Hessian metric (reference systems.py:1737-1920).  The built-in targets with a device Hessian
    (funnel, poly) bring it themselves; for any other target ``hess_neg_log_dens`` takes a ``models.UserHessian`` - the
    Hessian and its matrix-Tressian product as device code (the reference's ``hess_neg_log_dens`` / ``mtp_neg_log_dens``
    callables), dense, dim <= 256 (beyond 64 on the workspace tiers of csrc/softabs.h)."""

    _kind = "riemann"

    def __init__(self, neg_log_dens, *, grad_neg_log_dens=None, hess_neg_log_dens=None,
                 mtp_neg_log_dens=None, softabs_coeff=1.0, backend=None):
        super().__init__(neg_log_dens, grad_neg_log_dens, backend)
        if mtp_neg_log_dens is not None:
            raise ValueError("the matrix-Tressian product comes with the device Hessian (models.UserHessian); leave it None")
        if hess_neg_log_dens is not None and not isinstance(hess_neg_log_dens, models.UserHessian):
            raise TypeError("hess_neg_log_dens must be a mici_amd.models.UserHessian (or None for a built-in device Hessian)")
        if softabs_coeff <= 0:
            raise ValueError("softabs_coeff must be positive.")
        self.softabs_coeff = float(softabs_coeff)
        self.hessian = hess_neg_log_dens

    def _model_args(self):
        if self.hessian is not None:
            return dict(rmetric=models.RMETRIC_SOFTABS_USER,
                        rmetric_params=np.concatenate([[self.softabs_coeff], self.hessian.params]),
                        rmetric_source=self.hessian.source)
        return dict(rmetric=models.RMETRIC_SOFTABS, rmetric_params=[self.softabs_coeff])


class DenseConstrainedEuclideanMetricSystem(EuclideanMetricSystem):
    """Euclidean-metric system on the manifold {q : constr(q) = 0} (reference systems.py:876-1031,
    619-873).  ``dens_wrt_hausdorff=False`` adds the half log-determinant of the Gram matrix to h1 and its
    gradient (through the device model's constraint Hessian) to dh1_dpos (systems.py:829-862, 1024-1031)."""

    _kind = "constrained"

    def __init__(self, neg_log_dens, constr, *, metric=None, dens_wrt_hausdorff=True,
                 grad_neg_log_dens=None, jacob_constr=None, mhp_constr=None, backend=None):
        super().__init__(neg_log_dens, metric=metric, grad_neg_log_dens=grad_neg_log_dens,
                         backend=backend)
        if not isinstance(constr, models.Constraint):
            raise TypeError("constr must be a built-in mici_amd.models.Constraint")
        if jacob_constr is not None or mhp_constr is not None:
            raise ValueError("constraint derivatives are supplied by the device model")
        self.dens_wrt_hausdorff = bool(dens_wrt_hausdorff)
        self.constraint = constr

    def _model_args(self):
        args = EuclideanMetricSystem._model_args(self)
        args.update(constr=self.constraint.cid, constr_params=self.constraint.params,
                    dens_wrt_ambient=not self.dens_wrt_hausdorff,
                    n_constr=getattr(self.constraint, "n_constr", 0),
                    constr_source=getattr(self.constraint, "source", None))
        return args


class GaussianDenseConstrainedEuclideanMetricSystem(DenseConstrainedEuclideanMetricSystem):
    """Gaussian split on a constrained system (reference systems.py:1034-1184): h2 = q.q/2 + p.M^-1 p/2 with
    the exact rotation as h2_flow, dh2_flow_dmom = (V diag(sin(w|t|) w) V^T, V diag(cos(w|t|)) V^T), symmetric
    (eigendecomposed) Gram-type matrices, and always ``dens_wrt_hausdorff=False``."""

    def __init__(self, neg_log_dens, constr, *, metric=None, grad_neg_log_dens=None, jacob_constr=None,
                 mhp_constr=None, backend=None):
        super().__init__(neg_log_dens, constr, metric=metric, dens_wrt_hausdorff=False,
                         grad_neg_log_dens=grad_neg_log_dens, jacob_constr=jacob_constr,
                         mhp_constr=mhp_constr, backend=backend)

    def _model_args(self):
        args = super()._model_args()
        args.update(gaussian_split=True)
        return args
