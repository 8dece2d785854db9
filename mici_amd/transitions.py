"""Momentum refresh and Metropolis static-integration transitions with the reference's ``Transition``
surface (mici/transitions.py), device resident: the trajectory, both Hamiltonian evaluations and the
accept / select step run on the GPU with no per-step host round trip.

Each class offers
  * ``sample(state, rng) -> (state, stats)``  - the reference contract for ONE chain
    (transitions.py:129-142, 275-352): same draws from ``rng`` in the same order (the accept uniform is
    drawn only when the trajectory raised no integrator error), same statistics dictionary;
  * ``sample_batch(batch, ...)``              - N chains resident in HBM (``DeviceBatch``).
There is no CPU path: a missing library or device raises ``DeviceError``."""

from __future__ import annotations

import numpy as np

from . import _ffi
from .errors import LinAlgError
from .runtime import ContextCache, DeviceBatch, default_context


class IndependentMomentumTransition:
    """Independently resample the momentum from its conditional distribution (transitions.py:129-142)."""

    state_variables = {"mom"}
    statistic_types = None

    def __init__(self, system):
        self.system = system

    def sample(self, state, rng):
        state.mom = self.system.sample_momentum(state, rng)
        return state, None

    def sample_batch(self, batch, z, ctx=None):
        """``z``: standard-normal draws [N, D] (host); the momentum of every chain in ``batch`` is replaced."""
        ctx = ctx or batch.ctx
        z = np.ascontiguousarray(z, dtype=np.float64)
        if z.shape != (batch.n_chains, batch.dim):
            raise ValueError(f"z must have shape ({batch.n_chains}, {batch.dim})")
        model = self.system.device_model(ctx)
        _ffi.check(ctx._lib.mm_sample_momentum(ctx.handle, model.handle, batch.handle,
                                               z.ctypes.data_as(_ffi.c_double_p)),
                   ctx.handle, "mm_sample_momentum")


    def sample_batch_device(self, batch, transition, ctx=None):
        """As ``sample_batch`` with z ~ N(0, I) drawn ON THE DEVICE (``batch.set_rng(seed, chain_offset)`` first):
        transition number ``transition`` of every chain's own counter-based stream - nothing is uploaded and the
        call does not synchronise."""
        ctx = ctx or batch.ctx
        model = self.system.device_model(ctx)
        _ffi.check(ctx._lib.mm_momentum_refresh_rng(ctx.handle, model.handle, batch.handle,
                                                    float(getattr(self, "mom_resample_coeff", 1.0)), int(transition)),
                   ctx.handle, "mm_momentum_refresh_rng")


class CorrelatedMomentumTransition(IndependentMomentumTransition):
    """Partial momentum refresh mom <- sqrt(1 - c^2) mom + c mom_ind (Horowitz 1991; transitions.py:143-198)."""

    def __init__(self, system, mom_resample_coeff=1.0):
        super().__init__(system)
        if not (mom_resample_coeff >= 0 and mom_resample_coeff <= 1):
            raise ValueError("mom_resample_coeff should have a value in the interval [0, 1].")
        self.mom_resample_coeff = mom_resample_coeff

    def sample(self, state, rng):
        if state.mom is None or self.mom_resample_coeff == 1:
            state.mom = self.system.sample_momentum(state, rng)
        elif self.mom_resample_coeff != 0:
            ctx = default_context()
            pos = np.ascontiguousarray(state.pos, dtype=np.float64)
            batch = DeviceBatch(ctx, 1, pos.shape[0])
            try:
                batch.upload(pos[None], np.asarray(state.mom, dtype=np.float64)[None], [int(state.dir)])
                self.sample_batch(batch, rng.standard_normal(pos.shape)[None], ctx)
                state.mom = batch.download()[1][0]
            finally:
                batch.close()
        return state, None

    def sample_batch(self, batch, z, ctx=None):
        ctx = ctx or batch.ctx
        z = np.ascontiguousarray(z, dtype=np.float64)
        if z.shape != (batch.n_chains, batch.dim):
            raise ValueError(f"z must have shape ({batch.n_chains}, {batch.dim})")
        model = self.system.device_model(ctx)
        _ffi.check(ctx._lib.mm_momentum_refresh(ctx.handle, model.handle, batch.handle,
                                                z.ctypes.data_as(_ffi.c_double_p), float(self.mom_resample_coeff)),
                   ctx.handle, "mm_momentum_refresh")


class MetropolisStaticIntegrationTransition:
    """Static-trajectory HMC transition with a Metropolis accept step (transitions.py:236-352): integrate
    ``n_step`` steps in the current direction, accept the end point with probability
    min(1, exp(h_init - h_final)), negate the direction on rejection."""

    state_variables = {"pos", "mom", "dir"}

    def __init__(self, system, integrator, n_step):
        if n_step <= 0:
            raise ValueError("Number of integrator steps must be positive.")
        self.system = system
        self.integrator = integrator
        self.n_step = int(n_step)
        self._statistic_types = {
            "n_step": (np.int64, -1),
            "accept_stat": (np.float64, np.nan),
            "non_reversible_step": (bool, False),
            "convergence_error": (bool, False),
            "step_size": (np.float64, np.nan),
            "metrop_accept_prob": (np.float64, np.nan),
        }
        self._proposals = ContextCache()

    @property
    def statistic_types(self):
        return self._statistic_types

    # Device handles never travel: the reference's sampler deep-copies its transitions per chain and per stage and
    # pickles them for process pools (samplers.py:1124-1129); a copy starts without cached device batches and
    # re-creates them lazily on whatever context its own thread / process uses.
    _DEVICE_CACHES = ("_proposals", "_one")

    def __getstate__(self):
        d = self.__dict__.copy()
        d["_proposals"] = ContextCache()
        d.pop("_one", None)
        return d

    def __deepcopy__(self, memo):
        import copy
        new = object.__new__(type(self))
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k == "_proposals":
                new.__dict__[k] = ContextCache()
            elif k != "_one":
                new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    def _proposal_for(self, batch):
        prop = self._proposals.get(batch.ctx, batch.n_chains, batch.dim)
        if prop is None:
            prop = self._proposals.put(batch.ctx, DeviceBatch(batch.ctx, batch.n_chains, batch.dim), batch.n_chains,
                                       batch.dim)
        return prop

    # ---- N chains, device resident ---------------------------------------------------------------------
    def propose_batch(self, batch, ctx=None):
        """Copy the chains and run the trajectory on the copy; returns the proposal batch and its per-chain
        (status, n_done).  Split from :py:meth:`accept_batch` so that a caller that replays a reference
        random stream can draw the accept uniforms only for the chains without an integration error."""
        ctx = ctx or batch.ctx
        prop = self._proposal_for(batch)
        _ffi.check(ctx._lib.mm_state_copy(prop.handle, batch.handle), ctx.handle, "mm_state_copy")
        self.integrator.step_device(prop, self.n_step, ctx)
        status, n_done = prop.download_status()
        if np.any(status == 5):  # a LinAlgError outside a solver is not an IntegratorError: it propagates
            raise LinAlgError("metric construction failed outside a solver for chain(s) "
                              f"{np.flatnonzero(status == 5).tolist()}")
        return prop, status, n_done

    def accept_batch(self, batch, prop, status, n_done, u, ctx=None):
        ctx = ctx or batch.ctx
        n = batch.n_chains
        u = np.ascontiguousarray(np.broadcast_to(np.asarray(u, dtype=np.float64), (n,)))
        prob = np.zeros(n)
        acc = np.zeros(n, dtype=np.int8)
        model = self.system.device_model(ctx)
        _ffi.check(ctx._lib.mm_metropolis_accept(
            ctx.handle, model.handle, batch.handle, prop.handle, u.ctypes.data_as(_ffi.c_double_p),
            prob.ctypes.data_as(_ffi.c_double_p), acc.ctypes.data_as(_ffi.c_int8_p)),
            ctx.handle, "mm_metropolis_accept")
        error = status != 0
        return {
            "n_step": n_done.astype(np.int64),
            "metrop_accept_prob": prob,
            "accept_stat": np.where(error, 0.0, prob),
            "convergence_error": (status >= 1) & (status <= 3),
            "non_reversible_step": status == 4,
            "step_size": np.full(n, self.integrator.step_size, dtype=np.float64),
            "accepted": acc.astype(bool),
        }

    def sample_batch(self, batch, u, ctx=None):
        """One transition for every chain of ``batch`` with the accept uniforms ``u[N]`` given up front."""
        prop, status, n_done = self.propose_batch(batch, ctx)
        return self.accept_batch(batch, prop, status, n_done, u, ctx)

    def sample_batch_device(self, batch, transition, ctx=None, stats=True):
        """One transition for every chain with the accept uniform drawn ON THE DEVICE (``batch.set_rng`` first):
        no per-transition upload.  ``stats=False`` also skips every download - the transition is then a pure
        sequence of asynchronous launches.  What went wrong with a chain is not lost: the accept step ORs every failed
        proposal's status into the batch's sticky error word on the device, and :py:meth:`check_errors` - to be called
        at the caller's next synchronisation point, e.g. with the trace download - raises the ``LinAlgError`` the
        reference lets propagate out of the sampler (status 5; the other statuses are rejections there too)."""
        ctx = ctx or batch.ctx
        prop = self._proposal_for(batch)
        _ffi.check(ctx._lib.mm_state_copy(prop.handle, batch.handle), ctx.handle, "mm_state_copy")
        self.integrator.step_device(prop, self.n_step, ctx)
        model = self.system.device_model(ctx)
        if not stats:
            _ffi.check(ctx._lib.mm_metropolis_accept_rng(ctx.handle, model.handle, batch.handle, prop.handle,
                                                         int(transition), None, None),
                       ctx.handle, "mm_metropolis_accept_rng")
            return None
        status, n_done = prop.download_status()
        if np.any(status == 5):
            raise LinAlgError("metric construction failed outside a solver for chain(s) "
                              f"{np.flatnonzero(status == 5).tolist()}")
        n = batch.n_chains
        prob = np.zeros(n)
        acc = np.zeros(n, dtype=np.int8)
        _ffi.check(ctx._lib.mm_metropolis_accept_rng(
            ctx.handle, model.handle, batch.handle, prop.handle, int(transition),
            prob.ctypes.data_as(_ffi.c_double_p), acc.ctypes.data_as(_ffi.c_int8_p)),
            ctx.handle, "mm_metropolis_accept_rng")
        error = status != 0
        return {
            "n_step": n_done.astype(np.int64),
            "metrop_accept_prob": prob,
            "accept_stat": np.where(error, 0.0, prob),
            "convergence_error": (status >= 1) & (status <= 3),
            "non_reversible_step": status == 4,
            "step_size": np.full(n, self.integrator.step_size, dtype=np.float64),
            "accepted": acc.astype(bool),
        }

    @staticmethod
    def check_errors(batch, clear=True):
        """Read (and clear) the batch's sticky error word; raise ``LinAlgError`` for chains with a status-5 proposal
        (a LinAlgError outside a solver propagates in the reference, transitions.py:292-295); return the word."""
        errs = batch.download_errors(clear)
        bad = np.flatnonzero(errs & (1 << 5))
        if bad.size:
            raise LinAlgError(f"metric construction failed outside a solver for chain(s) {bad.tolist()}")
        return errs

    # ---- one chain, the reference's contract -----------------------------------------------------------
    def sample(self, state, rng):
        ctx = default_context()
        pos = np.ascontiguousarray(state.pos, dtype=np.float64)
        batch = getattr(self, "_one", None)
        if batch is None or batch.handle is None or batch.dim != pos.shape[0] or batch.ctx is not ctx:
            batch = self._one = DeviceBatch(ctx, 1, pos.shape[0])
        batch.upload(pos[None], np.asarray(state.mom, dtype=np.float64)[None], [int(state.dir)])
        prop, status, n_done = self.propose_batch(batch, ctx)
        # `not integration_error and rng.uniform() < accept_prob`: no draw after an integration error
        u = rng.uniform() if status[0] == 0 else 2.0
        st = self.accept_batch(batch, prop, status, n_done, [u], ctx)
        q, p, d = batch.download()
        state.pos, state.mom, state.dir = q[0], p[0], int(d[0])
        stats = {
            "convergence_error": bool(st["convergence_error"][0]),
            "non_reversible_step": bool(st["non_reversible_step"][0]),
            "step_size": self.integrator.step_size,
            "n_step": int(st["n_step"][0]),
            "metrop_accept_prob": float(st["metrop_accept_prob"][0]),
            "accept_stat": float(st["accept_stat"][0]),
        }
        return state, stats


class MetropolisRandomIntegrationTransition(MetropolisStaticIntegrationTransition):
    """As the static transition with the number of steps drawn per transition from
    ``rng.integers(*n_step_range)`` (transitions.py:355-402).  In the reference every chain draws its own
    ``n_step`` for every transition; ``sample_batch`` does the same in one launch through per-chain trajectory
    lengths on the device (``mm_state_set_chain_steps``): pass ``n_step`` as an [N] array, a scalar for one
    common length, or ``rngs`` (one generator per chain, drawn in chain order exactly like ``sample``)."""

    def __init__(self, system, integrator, n_step_range):
        n_step_lower, n_step_upper = n_step_range
        if not (n_step_lower > 0 and n_step_lower < n_step_upper):
            raise ValueError("Range bounds must be non-negative and first entry less than last.")
        super().__init__(system, integrator, n_step_lower)
        self.n_step_range = tuple(int(v) for v in n_step_range)

    def sample(self, state, rng):
        self.n_step = int(rng.integers(*self.n_step_range))
        return super().sample(state, rng)

    def sample_batch(self, batch, u, n_step=None, rng=None, ctx=None, rngs=None):
        if n_step is None:
            if rngs is not None:
                n_step = np.array([int(r.integers(*self.n_step_range)) for r in rngs], dtype=np.int32)
            elif rng is not None:
                n_step = rng.integers(*self.n_step_range, size=batch.n_chains).astype(np.int32)
            else:
                raise ValueError("pass n_step, an rng or one rng per chain to draw it from")
        n_step = np.asarray(n_step)
        if n_step.ndim == 0:
            self.n_step = int(n_step)
            return super().sample_batch(batch, u, ctx)
        if n_step.shape != (batch.n_chains,) or np.any(n_step < 1):
            raise ValueError("n_step must be a positive integer per chain")
        self.n_step = int(n_step.max())
        batch.set_chain_steps(n_step)
        try:
            return super().sample_batch(batch, u, ctx)
        finally:
            batch.set_chain_steps(None)
            self._proposal_for(batch).set_chain_steps(None)

    def sample_batch_device(self, batch, transition, ctx=None, stats=True):
        """Device draws for BOTH the per-chain trajectory length and the accept uniform (``batch.set_rng`` first)."""
        ctx = ctx or batch.ctx
        lo, hi = self.n_step_range
        _ffi.check(ctx._lib.mm_rng_chain_steps(batch.handle, int(transition), int(lo), int(hi)), ctx.handle,
                   "mm_rng_chain_steps")
        keep = self.n_step
        self.n_step = int(hi) - 1  # the launch runs to the longest possible trajectory; chains stop at their own count
        try:
            return super().sample_batch_device(batch, transition, ctx, stats)
        finally:
            self.n_step = keep
            # switches the per-chain counts off again; the device buffers stay allocated (no free, no synchronisation)
            batch.set_chain_steps(None)
            self._proposal_for(batch).set_chain_steps(None)
