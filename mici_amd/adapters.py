"""Warm-up adapters with the reference's ``Adapter`` surface (mici/adapters.py).

Dual-averaging step-size adaptation (Hoffman & Gelman 2014) with the reference's ``Adapter`` surface
(mici/adapters.py:174-389), batched over device-resident chains: every chain adapts ITS OWN step size, as in
the reference (one integrator copy per chain), through per-chain step-size factors attached to the batch
(``DeviceBatch.set_step_scale`` / ``mm_state_set_step_scale``) while the integrator's own ``step_size`` is
held at 1.  The arithmetic on the N adaptation states is NumPy on the host (N scalars per iteration); the
trajectories, Hamiltonians and accept steps stay on the GPU.

  * ``initialize_batch / update_batch / finalize_batch``  - N chains in a ``DeviceBatch``;
  * ``initialize / update / finalize``                   - the reference's single-chain contract."""

from __future__ import annotations

import numpy as np

from . import _ffi
from .errors import AdaptationError
from .runtime import DeviceBatch, default_context


def arithmetic_mean_log_step_size_reducer(log_step_sizes):
    """adapters.py:126-135"""
    return float(np.mean(np.exp(np.asarray(list(log_step_sizes), dtype=np.float64))))


def geometric_mean_log_step_size_reducer(log_step_sizes):
    """adapters.py:138-147"""
    return float(np.exp(np.mean(np.asarray(list(log_step_sizes), dtype=np.float64))))


def min_log_step_size_reducer(log_step_sizes):
    """adapters.py:150-159"""
    return float(np.exp(np.min(np.asarray(list(log_step_sizes), dtype=np.float64))))


def default_adapt_stat_func(stats):
    return stats["accept_stat"]


class DualAveragingStepSizeAdapter:
    """Controls the integrator step size so that a transition statistic (by default ``accept_stat``) is close
    to a target value (reference adapters.py:174-389; same constructor arguments and defaults)."""

    is_fast = True

    def __init__(self, adapt_stat_target=0.8, adapt_stat_func=None, log_step_size_reg_target=None,
                 log_step_size_reg_coefficient=0.05, iter_decay_coeff=0.75, iter_offset=10,
                 max_init_step_size_iters=100, log_step_size_reducer=None):
        self.adapt_stat_target = adapt_stat_target
        self.adapt_stat_func = default_adapt_stat_func if adapt_stat_func is None else adapt_stat_func
        self.log_step_size_reg_target = log_step_size_reg_target
        self.log_step_size_reg_coefficient = log_step_size_reg_coefficient
        self.iter_decay_coeff = iter_decay_coeff
        self.iter_offset = iter_offset
        self.max_init_step_size_iters = max_init_step_size_iters
        self.log_step_size_reducer = (arithmetic_mean_log_step_size_reducer if log_step_size_reducer is None
                                      else log_step_size_reducer)

    # ---- N chains ------------------------------------------------------------------------------------------
    def _hamiltonians(self, system, batch, ctx):
        h = np.zeros(batch.n_chains)
        _ffi.check(ctx._lib.mm_hamiltonian(ctx.handle, system.device_model(ctx).handle, batch.handle,
                                           h.ctypes.data_as(_ffi.c_double_p)), ctx.handle, "mm_hamiltonian")
        return h

    def find_init_step_sizes(self, batch, system, integrator, ctx=None):
        """Coarse search of adapters.py:271-344 for every chain at once: from 1, halve / double each chain's
        step size until the change in the Hamiltonian over ONE step crosses log 2; a failed step counts as
        "too big" from then on.  Chains that have found their value idle while the others continue."""
        ctx = ctx or batch.ctx
        n = batch.n_chains
        h_init = self._hamiltonians(system, batch, ctx)
        if np.any(np.isnan(h_init)):
            raise AdaptationError("Hamiltonian evaluating to NaN at initial state.")
        trial = DeviceBatch(ctx, n, batch.dim)
        eps = np.ones(n)
        too_big = np.zeros(n, dtype=bool)
        found = np.zeros(n, dtype=bool)
        threshold = np.log(2.0)
        saved_step_size = integrator.step_size
        integrator.step_size = 1.0
        try:
            for s in range(self.max_init_step_size_iters):
                batch.set_step_scale(eps)
                _ffi.check(ctx._lib.mm_state_copy(trial.handle, batch.handle), ctx.handle, "mm_state_copy")
                integrator.step_device(trial, 1, ctx)
                status, _ = integrator._status(trial, 1)
                if np.any(status == 5):
                    # a LinAlgError outside a solver is not an IntegratorError: the reference's search only
                    # catches IntegratorError (adapters.py:326-331), so it propagates (as in propose_batch)
                    from .errors import LinAlgError
                    raise LinAlgError("metric construction failed outside a solver for chain(s) "
                                      f"{np.flatnonzero(status == 5).tolist()}")
                failed = status != 0
                with np.errstate(invalid="ignore"):
                    delta_h = np.abs(h_init - self._hamiltonians(system, trial, ctx))
                    nan = np.isnan(delta_h)
                    decide = ~failed & ((s == 0) | nan)
                    too_big = np.where(decide, nan | (delta_h > threshold), too_big)
                    too_big |= failed
                    crossed = ~failed & ((too_big & (delta_h <= threshold)) | (~too_big & (delta_h > threshold)))
                found |= crossed
                if np.all(found):
                    break
                move = ~found
                eps = np.where(move & too_big, eps / 2, np.where(move, eps * 2, eps))
            else:
                raise AdaptationError(
                    f"Could not find reasonable initial step size in {self.max_init_step_size_iters} iterations "
                    f"(final step sizes {eps[~found].tolist()} for chains {np.flatnonzero(~found).tolist()}).")
        finally:
            integrator.step_size = saved_step_size
            trial.close()
        return eps

    def initialize_batch(self, batch, transition, ctx=None):
        """Adaptation state for every chain of ``batch`` (momenta must be set).  Leaves the per-chain initial
        step sizes attached to the batch and ``transition.integrator.step_size == 1``."""
        ctx = ctx or batch.ctx
        eps = self.find_init_step_sizes(batch, transition.system, transition.integrator, ctx)
        transition.integrator.step_size = 1.0
        batch.set_step_scale(eps)
        n = batch.n_chains
        reg = (np.log(10 * eps) if self.log_step_size_reg_target is None
               else np.full(n, float(self.log_step_size_reg_target)))
        return {"iter": 0, "smoothed_log_step_size": np.zeros(n), "adapt_stat_error": np.zeros(n),
                "log_step_size_reg_target": reg, "step_size": eps.copy()}

    def update_batch(self, adapt_state, batch, trans_stats):
        """adapters.py:346-368 for every chain; attaches the new per-chain step sizes to ``batch``."""
        adapt_state["iter"] += 1
        it = adapt_state["iter"]
        error_weight = 1 / (self.iter_offset + it)
        adapt_state["adapt_stat_error"] *= 1 - error_weight
        adapt_state["adapt_stat_error"] += error_weight * (
            self.adapt_stat_target - np.asarray(self.adapt_stat_func(trans_stats), dtype=np.float64))
        smoothing_weight = (1 / it) ** self.iter_decay_coeff
        log_step_size = adapt_state["log_step_size_reg_target"] - (
            adapt_state["adapt_stat_error"] * it ** 0.5 / self.log_step_size_reg_coefficient)
        adapt_state["smoothed_log_step_size"] *= 1 - smoothing_weight
        adapt_state["smoothed_log_step_size"] += smoothing_weight * log_step_size
        adapt_state["step_size"] = np.exp(log_step_size)
        batch.set_step_scale(adapt_state["step_size"])

    def finalize_batch(self, adapt_state, batch, transition):
        """adapters.py:370-389: reduce the per-chain smoothed estimates to the one step size of the main stage."""
        transition.integrator.step_size = self.log_step_size_reducer(adapt_state["smoothed_log_step_size"])
        batch.set_step_scale(None)

    # ---- one chain: the reference's contract -------------------------------------------------------------------
    def initialize(self, chain_state, transition):
        ctx = default_context()
        pos = np.ascontiguousarray(chain_state.pos, dtype=np.float64)
        batch = DeviceBatch(ctx, 1, pos.shape[0])
        try:
            batch.upload(pos[None], np.asarray(chain_state.mom, dtype=np.float64)[None], [int(chain_state.dir)])
            eps = self.find_init_step_sizes(batch, transition.system, transition.integrator, ctx)
        finally:
            batch.close()
        transition.integrator.step_size = float(eps[0])
        return {"iter": 0, "smoothed_log_step_size": 0.0, "adapt_stat_error": 0.0,
                "log_step_size_reg_target": (float(np.log(10 * eps[0])) if self.log_step_size_reg_target is None
                                             else self.log_step_size_reg_target)}

    def update(self, adapt_state, chain_state, trans_stats, transition):
        adapt_state["iter"] += 1
        error_weight = 1 / (self.iter_offset + adapt_state["iter"])
        adapt_state["adapt_stat_error"] *= 1 - error_weight
        adapt_state["adapt_stat_error"] += error_weight * (self.adapt_stat_target - self.adapt_stat_func(trans_stats))
        smoothing_weight = (1 / adapt_state["iter"]) ** self.iter_decay_coeff
        log_step_size = adapt_state["log_step_size_reg_target"] - (
            adapt_state["adapt_stat_error"] * adapt_state["iter"] ** 0.5 / self.log_step_size_reg_coefficient)
        adapt_state["smoothed_log_step_size"] *= 1 - smoothing_weight
        adapt_state["smoothed_log_step_size"] += smoothing_weight * log_step_size
        transition.integrator.step_size = float(np.exp(log_step_size))

    def finalize(self, adapt_states, chain_states, transition, rngs):
        if isinstance(adapt_states, dict):
            transition.integrator.step_size = float(np.exp(adapt_states["smoothed_log_step_size"]))
        else:
            transition.integrator.step_size = self.log_step_size_reducer(
                [a["smoothed_log_step_size"] for a in adapt_states])


class _OnlineMomentsAdapter:
    """Shared part of the two metric adapters: Welford's running mean / second-moment update per chain
    (adapters.py:444-460, 574-590), the pairwise combination of the per-chain statistics at the end
    (Chan et al. / Schubert & Gertz, adapters.py:486-499, 616-630), the regularisation towards ``reg_scale``
    times the identity, and the momentum refresh under the new metric."""

    is_fast = False
    _outer = False  # full second-moment matrix instead of its diagonal

    def __init__(self, reg_iter_offset=5, reg_scale=1e-3):
        self.reg_iter_offset = reg_iter_offset
        self.reg_scale = reg_scale

    def _zeros(self, dim):
        return np.zeros((dim, dim) if self._outer else (dim,))

    def _accumulate(self, acc, before, after):
        if self._outer:
            acc += before[None, :] * after[:, None]
        else:
            acc += before * after

    # ---- the reference's single-chain contract ------------------------------------------------------------
    def initialize(self, chain_state, transition):
        dim = np.asarray(chain_state.pos).shape[0]
        return {"iter": 0, "mean": np.zeros(dim), self._key: self._zeros(dim)}

    def update(self, adapt_state, chain_state, trans_stats, transition):
        pos = np.asarray(chain_state.pos, dtype=np.float64)
        adapt_state["iter"] += 1
        before = pos - adapt_state["mean"]
        adapt_state["mean"] += before / adapt_state["iter"]
        self._accumulate(adapt_state[self._key], before, pos - adapt_state["mean"])

    def _combine(self, adapt_states):
        """(n_iter, accumulated second moment) over one or several chains."""
        if isinstance(adapt_states, dict):
            return adapt_states["iter"], adapt_states.pop(self._key)
        n_iter = mean = acc = None
        for i, st in enumerate(adapt_states):
            if i == 0:
                n_iter, mean, acc = st["iter"], st.pop("mean"), st.pop(self._key)
                continue
            n_prev = n_iter
            n_iter += st["iter"]
            mean_diff = mean - st["mean"]
            mean *= n_prev
            mean += st["iter"] * st["mean"]
            mean /= n_iter
            acc += st[self._key]
            cross = np.outer(mean_diff, mean_diff) if self._outer else mean_diff**2
            acc += cross * (st["iter"] * n_prev) / n_iter
        return n_iter, acc

    def _estimate(self, adapt_states):
        n_iter, est = self._combine(adapt_states)
        if n_iter < 2:
            raise AdaptationError("At least two chain samples required to compute a variance estimates.")
        est /= n_iter - 1
        if not self._outer and (self.reg_iter_offset is None or self.reg_iter_offset == 0):
            return est
        est *= n_iter / (self.reg_iter_offset + n_iter)
        shrink = self.reg_scale * (self.reg_iter_offset / (self.reg_iter_offset + n_iter))
        if self._outer:
            est[np.diag_indices_from(est)] += shrink
        else:
            est += shrink
        return est

    def finalize(self, adapt_states, chain_states, transition, rngs):
        single = isinstance(adapt_states, dict)
        est = self._estimate(adapt_states)
        self._install(transition.system, est)
        # resample the momenta: their distribution changed with the metric (adapters.py:507-509, 634-636)
        for chain_state, rng in zip([chain_states] if single else chain_states, [rngs] if single else rngs):
            chain_state.mom = transition.system.sample_momentum(chain_state, rng)

    # ---- N device-resident chains ---------------------------------------------------------------------------
    def initialize_batch(self, batch, transition=None, ctx=None):
        n, dim = batch.n_chains, batch.dim
        return {"iter": 0, "mean": np.zeros((n, dim)),
                self._key: np.zeros((n, dim, dim) if self._outer else (n, dim))}

    def update_batch(self, adapt_state, batch, trans_stats=None):
        """One Welford update per chain from the batch's current positions (one download of pos[N, D])."""
        pos, _, _ = batch.download()
        adapt_state["iter"] += 1
        before = pos - adapt_state["mean"]
        adapt_state["mean"] += before / adapt_state["iter"]
        after = pos - adapt_state["mean"]
        if self._outer:
            adapt_state[self._key] += before[:, None, :] * after[:, :, None]
        else:
            adapt_state[self._key] += before * after

    def finalize_batch(self, adapt_state, batch, transition, z):
        """Combine the chains' statistics exactly as ``finalize`` does for a list of per-chain states, install the
        metric and resample every chain's momentum on the device from the standard normal draws ``z[N, D]``."""
        n = batch.n_chains
        states = [{"iter": adapt_state["iter"], "mean": adapt_state["mean"][c].copy(),
                   self._key: adapt_state[self._key][c].copy()} for c in range(n)]
        est = self._estimate(states)
        self._install(transition.system, est)
        ctx = batch.ctx
        z = np.ascontiguousarray(z, dtype=np.float64)
        if z.shape != (n, batch.dim):
            raise ValueError("z must be [N, D] standard normal draws")
        _ffi.check(ctx._lib.mm_sample_momentum(ctx.handle, transition.system.device_model(ctx).handle, batch.handle,
                                               z.ctypes.data_as(_ffi.c_double_p)), ctx.handle, "mm_sample_momentum")
        return est


class OnlineVarianceMetricAdapter(_OnlineMomentsAdapter):
    """Diagonal metric from an online estimate of the posterior variances (reference adapters.py:392-514): the
    metric is the INVERSE of the regularised variance estimate."""

    _key = "sum_diff_sq"

    def _install(self, system, var_est):
        system.set_metric(1.0 / var_est)  # PositiveDiagonalMatrix(var_est).inv


class OnlineCovarianceMetricAdapter(_OnlineMomentsAdapter):
    """Dense metric from an online estimate of the posterior covariance (reference adapters.py:517-644): the
    metric is the inverse of the regularised covariance estimate."""

    _key = "sum_diff_outer"
    _outer = True

    def _install(self, system, covar_est):
        system.set_metric(np.linalg.inv(covar_est))  # DensePositiveDefiniteMatrix(covar_est).inv
