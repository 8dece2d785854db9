"""Symplectic integrators with the reference's ``Integrator`` surface (mici/integrators.py) whose
``step`` runs on the MI355X through libmici_amd.so.

Each class offers
  * ``step(state) -> new state``  - exactly the reference contract (integrators.py:63-80): the input
    state is not modified, ``AdaptationError`` if ``step_size is None``, and the reference's
    exception classes (``ConvergenceError``, ``NonReversibleStepError``, ``LinAlgError``) are raised
    for the corresponding per-chain device status, so mici.transitions / adapters / samplers drive
    it unchanged (verified by duck-typing: ``system``, ``step_size`` read/write, ``step``);
  * ``step_batch(pos, mom, dir, n_steps) -> (pos, mom, status, n_done)`` - N chains, host arrays;
  * ``step_device(batch, n_steps)`` - chains already resident in HBM (``DeviceBatch``), no copies.
There is no CPU path: a missing library or device raises ``DeviceError``."""

from __future__ import annotations

import ctypes as C

import numpy as np

from . import _ffi, solvers
from .errors import AdaptationError, raise_for_status
from .runtime import ContextCache, DeviceBatch, default_context


class Integrator:
    """Base class (reference integrators.py:30-89)."""

    _needs = None  # system kind required

    def __init__(self, system, step_size=None):
        if self._needs is not None and getattr(system, "_kind", None) != self._needs:
            raise ValueError(f"{type(self).__name__} needs a {self._needs} system, got "
                             f"{type(system).__name__}")
        self.system = system
        self.step_size = step_size
        self._one = ContextCache()  # single-chain DeviceBatch per context
        self.last_counters = None

    # pickling / deepcopy: device handles are dropped and lazily re-created; every copy has its own
    # step_size (adapters mutate it per chain, adapters.py:373) - SURVEY.md H9
    def __getstate__(self):
        d = self.__dict__.copy()
        d["_one"] = ContextCache()
        return d

    def __deepcopy__(self, memo):
        import copy
        new = object.__new__(type(self))
        for k, v in self.__dict__.items():
            new.__dict__[k] = ContextCache() if k == "_one" else (v if k == "system" else copy.deepcopy(v, memo))
        return new

    def _check_step_size(self):
        if self.step_size is None:
            raise AdaptationError(
                "Integrator `step_size` is `None`. This value should only be used if a step size "
                "adapter is being used to set the step size.")

    # ---- device entry point, implemented by subclasses ------------------------------------------
    def _launch(self, ctx, model, batch, n_steps):
        raise NotImplementedError

    def step_device(self, batch, n_steps=1, ctx=None):
        """Advance a device-resident batch by ``n_steps`` (asynchronous on the context stream)."""
        self._check_step_size()
        ctx = ctx or batch.ctx
        self._launch(ctx, self.system.device_model(ctx), batch, int(n_steps))

    def step_batch(self, pos, mom, dir=1, n_steps=1, ctx=None):  # noqa: A002
        """N chains from host arrays; returns ``(pos, mom, status[N], n_done[N])``.  A failed chain is
        frozen at its last successfully completed step (transitions.py:292-295)."""
        self._check_step_size()
        ctx = ctx or default_context()
        pos = np.ascontiguousarray(pos, dtype=np.float64)
        if pos.ndim != 2:
            raise ValueError("pos must be [N, D]")
        batch = DeviceBatch(ctx, pos.shape[0], pos.shape[1])
        try:
            batch.upload(pos, mom, dir)
            self.step_device(batch, n_steps, ctx)
            q, p, _, status, n_done = batch.download_all()
        finally:
            batch.close()
        return q, p, status, n_done

    def _status(self, batch, n_steps):
        return batch.download_status()

    def step(self, state):
        """Single-chain step with the reference's semantics (integrators.py:63-80).  The chain's state lives in a
        cached buffer of pinned host memory that the kernels access in place: the inputs are written through NumPy
        views, one launch, one stream synchronisation, and the results are read back through the same views - no
        upload / download calls (tools/host_latency.py: launch + synchronisation is 14 us of the call on the MI355X
        box, the two transfer calls used to add 17 us)."""
        self._check_step_size()
        ctx = default_context()
        # (the mapped views below take whatever broadcasts: a scalar or length-1 pos / mom would fill the whole vector)
        pos = np.asarray(state.pos, dtype=np.float64)
        mom = np.asarray(state.mom, dtype=np.float64)
        if pos.ndim != 1 or mom.shape != pos.shape:
            raise ValueError("state.pos and state.mom must be vectors of the same length")
        dim = pos.shape[0]
        batch = self._one.get(ctx)
        if batch is None or batch.dim != dim:
            batch = self._one.put(ctx, DeviceBatch(ctx, 1, dim, mapped=True))
        vq, vp, vd, vs, _ = batch.mapped_views()
        d = int(state.dir)
        if d != 1 and d != -1:
            raise ValueError("dir entries must be +1 or -1")
        vq[0] = pos
        vp[0] = mom
        vd[0] = d
        self.step_device(batch, 1, ctx)
        ctx.sync()
        st = int(vs[0])
        if st:
            raise_for_status(st)
        new = state.copy()
        new.pos = vq[0].copy()
        new.mom = vp[0].copy()
        return new


class LeapfrogIntegrator(Integrator):
    r"""Explicit leapfrog :math:`\Phi_1(t/2)\circ\Phi_2(t)\circ\Phi_1(t/2)` on an
    ``EuclideanMetricSystem`` (reference integrators.py:134-173)."""

    _needs = "euclid"

    def _launch(self, ctx, model, batch, n_steps):
        _ffi.check(ctx._lib.mm_leapfrog_euclid(ctx.handle, model.handle, batch.handle,
                                               float(self.step_size), n_steps),
                   ctx.handle, "mm_leapfrog_euclid")


class SymmetricCompositionIntegrator(Integrator):
    r"""Symmetric composition integrator :math:`\Psi(t) = A(a_S t) \circ B(b_S t) \circ \dots \circ
    A(a_0 t)` for Hamiltonians with tractable flows, on an ``EuclideanMetricSystem`` (reference
    integrators.py:176-274).  ``free_coefficients`` are the :math:`S - 1` free coefficients
    :math:`(a_0, b_1, a_1, \dots)`; consistency and symmetry fix the rest exactly as the reference does
    (:258-268)."""

    _needs = "euclid"
    MAX_COEFFICIENTS = 16  # MM_MAX_COMPOSITION_COEFFS

    def __init__(self, system, free_coefficients, step_size=None, initial_h1_flow_step=True):
        super().__init__(system, step_size)
        self.initial_h1_flow_step = initial_h1_flow_step
        free = [float(c) for c in free_coefficients]
        n_free = len(free)
        coefficients = list(free)
        coefficients.append(0.5 - sum(free[n_free % 2::2]))
        coefficients.append(1 - 2 * sum(free[(n_free + 1) % 2::2]))
        self.coefficients = coefficients + coefficients[-2::-1]
        if len(self.coefficients) > self.MAX_COEFFICIENTS:
            raise ValueError(f"at most {self.MAX_COEFFICIENTS} flow coefficients are supported on the device")

    def _launch(self, ctx, model, batch, n_steps):
        coeffs = (C.c_double * len(self.coefficients))(*self.coefficients)
        _ffi.check(ctx._lib.mm_composition_euclid(ctx.handle, model.handle, batch.handle,
                                                  float(self.step_size), n_steps, len(self.coefficients),
                                                  coeffs, 1 if self.initial_h1_flow_step else 0),
                   ctx.handle, "mm_composition_euclid")


class BCSSTwoStageIntegrator(SymmetricCompositionIntegrator):
    """Two-stage integrator of Blanes, Casas & Sanz-Serna (2014), eq. (6.4) (reference integrators.py:277-307)."""

    def __init__(self, system, step_size=None):
        super().__init__(system, ((3 - 3**0.5) / 6,), step_size=step_size, initial_h1_flow_step=True)


class BCSSThreeStageIntegrator(SymmetricCompositionIntegrator):
    """Three-stage integrator of Blanes, Casas & Sanz-Serna (2014), eq. (6.7) (reference integrators.py:310-344)."""

    def __init__(self, system, step_size=None):
        super().__init__(system, (0.11888010966548, 0.29619504261126), step_size=step_size,
                         initial_h1_flow_step=True)


class BCSSFourStageIntegrator(SymmetricCompositionIntegrator):
    """Four-stage integrator of Blanes, Casas & Sanz-Serna (2014), eq. (6.8) (reference integrators.py:347-378)."""

    def __init__(self, system, step_size=None):
        super().__init__(system, (0.071353913450279725904, 0.191667800000000000000, 0.268548791161230105820),
                         step_size=step_size, initial_h1_flow_step=True)


class ImplicitLeapfrogIntegrator(Integrator):
    """Implicit (generalised) leapfrog on a Riemannian-metric system with fixed-point solves and
    reversibility checks (reference integrators.py:381-544).  NB as in the reference every
    sub-map uses the full ``step_size`` (SURVEY.md hazard H1).  A plain Euclidean-metric system is accepted
    too, as in the reference's own tests (tests/test_integrators.py:435-462): its implicit maps are explicit
    and the step is A(t) C(t) C(t) A(t)."""

    _needs = None  # riemann, or plain euclid

    def __init__(self, system, step_size=None, reverse_check_tol=2e-8,
                 reverse_check_norm=solvers.maximum_norm,
                 fixed_point_solver=solvers.solve_fixed_point_direct,
                 fixed_point_solver_kwargs=None):
        kind = getattr(system, "_kind", None)
        if kind not in ("riemann", "euclid") or getattr(system, "_gaussian_split", False):
            raise ValueError(f"{type(self).__name__} needs a Riemannian- or plain Euclidean-metric system, got "
                             f"{type(system).__name__}")
        super().__init__(system, step_size)
        self.reverse_check_tol = reverse_check_tol
        self.reverse_check_norm = reverse_check_norm
        self.fixed_point_solver = fixed_point_solver
        self.fixed_point_solver_kwargs = dict(fixed_point_solver_kwargs or {})

    def _opts(self):
        kw = dict(solvers.FIXED_POINT_DEFAULTS)
        unknown = set(self.fixed_point_solver_kwargs) - set(kw)
        if unknown:
            raise ValueError(f"unknown fixed_point_solver_kwargs: {sorted(unknown)}")
        kw.update(self.fixed_point_solver_kwargs)
        o = _ffi.FpOpts()
        o.conv_tol = kw["convergence_tol"]
        o.div_tol = kw["divergence_tol"]
        o.max_iters = int(kw["max_iters"])
        o.norm = solvers.norm_code(kw["norm"])
        o.solver = solvers.fp_solver_code(self.fixed_point_solver)
        o.rev_norm = solvers.norm_code(self.reverse_check_norm)
        o.rev_tol = self.reverse_check_tol
        return o

    def _launch(self, ctx, model, batch, n_steps):
        opts = self._opts()
        counters = _ffi.Counters()
        _ffi.check(ctx._lib.mm_implicit_leapfrog(ctx.handle, model.handle, batch.handle,
                                                 float(self.step_size), n_steps, C.byref(opts),
                                                 C.byref(counters)),
                   ctx.handle, "mm_implicit_leapfrog")
        self.last_counters = counters.as_dict()

    def _status(self, batch, n_steps):
        return batch.download_status()


class ImplicitMidpointIntegrator(ImplicitLeapfrogIntegrator):
    """Implicit midpoint integrator for general Hamiltonians (reference integrators.py:547-681): implicit
    Euler half step (fixed point in the concatenated (pos, mom) vector), explicit Euler half step,
    reversibility check.  Device support: Euclidean-metric systems (dim <= 1024), dense-Riemannian systems
    (dim <= 1024, built-in and user metrics alike: beyond 279 on the global-memory tier) and SoftAbs systems (dim <= 256, user Hessians included).  Same constructor arguments and solver options as :py:class:`ImplicitLeapfrogIntegrator`."""

    _needs = None  # euclid or riemann

    def __init__(self, system, step_size=None, reverse_check_tol=2e-8,
                 reverse_check_norm=solvers.maximum_norm,
                 fixed_point_solver=solvers.solve_fixed_point_direct, fixed_point_solver_kwargs=None):
        if getattr(system, "_kind", None) not in ("euclid", "riemann"):
            raise ValueError(f"ImplicitMidpointIntegrator needs a Euclidean- or Riemannian-metric system, got "
                             f"{type(system).__name__}")
        Integrator.__init__(self, system, step_size)
        self.reverse_check_tol = reverse_check_tol
        self.reverse_check_norm = reverse_check_norm
        self.fixed_point_solver = fixed_point_solver
        self.fixed_point_solver_kwargs = dict(fixed_point_solver_kwargs or {})

    def _launch(self, ctx, model, batch, n_steps):
        opts = self._opts()
        counters = _ffi.Counters()
        _ffi.check(ctx._lib.mm_implicit_midpoint(ctx.handle, model.handle, batch.handle,
                                                 float(self.step_size), n_steps, C.byref(opts),
                                                 C.byref(counters)),
                   ctx.handle, "mm_implicit_midpoint")
        self.last_counters = counters.as_dict()


class ConstrainedLeapfrogIntegrator(Integrator):
    """Constrained leapfrog with Newton projection onto the manifold and reversibility check
    (reference integrators.py:684-984)."""

    _needs = "constrained"

    def __init__(self, system, step_size=None, n_inner_step=1, reverse_check_tol=2e-8,
                 reverse_check_norm=solvers.maximum_norm,
                 projection_solver=solvers.solve_projection_onto_manifold_newton,
                 projection_solver_kwargs=None):
        super().__init__(system, step_size)
        self.n_inner_step = n_inner_step
        self.reverse_check_tol = reverse_check_tol
        self.reverse_check_norm = reverse_check_norm
        self.projection_solver = projection_solver
        self.projection_solver_kwargs = dict(projection_solver_kwargs or {})

    def _opts(self):
        kw = dict(solvers.PROJECTION_DEFAULTS)
        unknown = set(self.projection_solver_kwargs) - set(kw)
        if unknown:
            raise ValueError(f"unknown projection_solver_kwargs: {sorted(unknown)}")
        kw.update(self.projection_solver_kwargs)
        if ("max_line_search_iters" in self.projection_solver_kwargs
                and solvers.proj_solver_code(self.projection_solver) != 2):
            raise ValueError("max_line_search_iters only applies to the line-search solver")
        o = _ffi.ProjOpts()
        o.constr_tol = kw["constraint_tol"]
        o.pos_tol = kw["position_tol"]
        o.div_tol = kw["divergence_tol"]
        o.max_iters = int(kw["max_iters"])
        o.norm = solvers.norm_code(kw["norm"])
        o.solver = solvers.proj_solver_code(self.projection_solver)
        o.rev_norm = solvers.norm_code(self.reverse_check_norm)
        o.rev_tol = self.reverse_check_tol
        o.n_inner = int(self.n_inner_step)
        o.max_line_search_iters = int(kw["max_line_search_iters"])
        return o

    def _launch(self, ctx, model, batch, n_steps):
        opts = self._opts()
        counters = _ffi.Counters()
        _ffi.check(ctx._lib.mm_constrained_leapfrog(ctx.handle, model.handle, batch.handle,
                                                    float(self.step_size), n_steps, C.byref(opts),
                                                    C.byref(counters)),
                   ctx.handle, "mm_constrained_leapfrog")
        self.last_counters = counters.as_dict()

    def _status(self, batch, n_steps):
        return batch.download_status()
