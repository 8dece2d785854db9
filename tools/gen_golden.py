#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the IMPORTED reference (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py [--out tests/golden]

The reference (``/root/reference/src``) is imported here and ONLY here; it never travels.
For every case the same closed-form model (``oracle/models.py``) is pushed through
  (1) the reference's own System + Integrator classes, one chain at a time, and
  (2) the NumPy restatement in ``oracle/integrators.py``,
the two are asserted equal (this is what pins the oracle), and the reference's inputs and
outputs are written as a small fixture: model parameters, (q0, p0, dir, step_size), the state
after each checkpointed number of steps, Hamiltonian values, per-chain status / completed
steps and the reference's call counters.  Fixtures are data only - no reference source.
"""

from __future__ import annotations

import argparse
from pathlib import Path
import collections
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")

import mici  # noqa: E402  (the reference)
from mici import errors as merr  # noqa: E402
from mici.states import ChainState  # noqa: E402

from oracle import integrators as orc  # noqa: E402
from oracle import models as mdl  # noqa: E402

SEED = 3046987125  # reference tests/test_integrators.py:8


def case_rng(name):
    """Per-case random stream: a case's inputs depend on its NAME only, not on how many cases were registered
    before it.  Every case added after round 1 draws from here (the round-1 cases keep the shared stream they
    were recorded with, whose positions are frozen by the registration order below)."""
    import zlib
    return np.random.default_rng([SEED, zlib.crc32(name.encode())])


def status_of(exc):
    msg = str(exc)
    if isinstance(exc, merr.NonReversibleStepError):
        return orc.ST_NON_REVERSIBLE
    if isinstance(exc, merr.ConvergenceError):
        if "diverged" in msg:
            return orc.ST_DIVERGED
        if "did not converge" in msg:
            return orc.ST_MAX_ITERS
        return orc.ST_SOLVER_LINALG
    if isinstance(exc, merr.LinAlgError):
        return orc.ST_LINALG
    raise exc


def safe_h(system, state):
    try:
        with np.errstate(all="ignore"):
            return system.h(state)
    except (merr.LinAlgError, ValueError):
        return np.nan


def run_reference(integrator, system, q0, p0, dirs, checkpoints):
    """Step every chain with the reference; freeze a chain at its last good state on failure."""
    n, d = q0.shape
    n_max = max(checkpoints)
    q_out = np.zeros((len(checkpoints), n, d))
    p_out = np.zeros((len(checkpoints), n, d))
    h_out = np.zeros((len(checkpoints), n))
    h0 = np.zeros(n)
    status = np.zeros(n, dtype=np.int32)
    n_done = np.zeros(n, dtype=np.int32)
    counts = collections.Counter()
    for c in range(n):
        cc = collections.Counter()
        state = ChainState(pos=q0[c].copy(), mom=p0[c].copy(), dir=int(dirs[c]), _call_counts=cc)
        h0[c] = safe_h(system, state)
        cc.clear()
        for s in range(1, n_max + 1):
            if status[c] == 0:
                try:
                    state = integrator.step(state)
                    n_done[c] = s
                except (merr.IntegratorError, merr.LinAlgError) as e:
                    status[c] = status_of(e)
            if s in checkpoints:
                k = checkpoints.index(s)
                q_out[k, c], p_out[k, c] = state.pos, state.mom
                h_out[k, c] = safe_h(system, state.copy())
        for key, val in cc.items():
            key = key[0] if isinstance(key, tuple) else key
            counts[str(key).split(".")[-1]] += val
    return dict(q_out=q_out, p_out=p_out, h0=h0, h_out=h_out, status=status, n_done=n_done), counts


def check_close(name, a, b, rtol, atol=0.0):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert np.array_equal(np.isnan(a), np.isnan(b)), f"{name}: NaN pattern differs"
    a, b = np.nan_to_num(a, nan=0.0), np.nan_to_num(b, nan=0.0)
    err = np.max(np.abs(a - b) / (atol + rtol * np.maximum(1.0, np.abs(b)))) if a.size else 0.0
    if not err <= 1.0:
        raise AssertionError(f"{name}: oracle differs from reference (scaled err {err:.3g})")


# ---------------------------------------------------------------------------------------------
def euclid_case(name, target, metric_kind, metric, q0, p0, dirs, h, checkpoints):
    if metric_kind == mdl.METRIC_IDENTITY:
        ref_metric = None
    else:
        ref_metric = np.array(metric)
    system = mici.systems.EuclideanMetricSystem(
        neg_log_dens=target.neg_log_dens, grad_neg_log_dens=target.grad, metric=ref_metric
    )
    integrator = mici.integrators.LeapfrogIntegrator(system, h)
    ref, counts = run_reference(integrator, system, q0, p0, dirs, checkpoints)
    osys = orc.EuclidSystem(target, metric_kind, metric)
    for k, s in enumerate(checkpoints):
        for c in range(q0.shape[0]):
            q, p = orc.leapfrog_steps(osys, q0[c], p0[c], dirs[c] * h, s)
            check_close(f"{name} q@{s}", q, ref["q_out"][k, c], 1e-13 * max(1, s))
            check_close(f"{name} p@{s}", p, ref["p_out"][k, c], 1e-13 * max(1, s))
            check_close(f"{name} h@{s}", np.array(osys.h(q, p)), ref["h_out"][k, c], 1e-12)
        qb, pb = orc.leapfrog_steps_batch(osys, q0, p0, dirs * h, s)
        check_close(f"{name} batch q@{s}", qb, ref["q_out"][k], 1e-12 * max(1, s))
        check_close(f"{name} batch p@{s}", pb, ref["p_out"][k], 1e-12 * max(1, s))
    return dict(
        kind="euclid", target=target.tid, target_params=target.params(), metric_kind=metric_kind,
        metric=np.zeros(0) if metric is None else np.asarray(metric), q0=q0, p0=p0, dir=dirs,
        step_size=h, checkpoints=np.array(checkpoints), **ref,
    ), counts


def symcomp_case(name, target, metric_kind, metric, q0, p0, dirs, h, checkpoints, free, initial_h1):
    """SymmetricCompositionIntegrator / BCSS integrators (integrators.py:176-378) on a Euclidean system."""
    ref_metric = None if metric_kind == mdl.METRIC_IDENTITY else np.array(metric)
    system = mici.systems.EuclideanMetricSystem(
        neg_log_dens=target.neg_log_dens, grad_neg_log_dens=target.grad, metric=ref_metric
    )
    bcss = {2: mici.integrators.BCSSTwoStageIntegrator, 3: mici.integrators.BCSSThreeStageIntegrator,
            4: mici.integrators.BCSSFourStageIntegrator}
    if free is None or isinstance(free, int):
        stages = free
        integrator = bcss[stages](system, h)
        free = orc.BCSS_FREE_COEFFICIENTS[stages]
        assert initial_h1
    else:
        integrator = mici.integrators.SymmetricCompositionIntegrator(
            system, free, step_size=h, initial_h1_flow_step=initial_h1)
    assert np.allclose(integrator.coefficients, orc.composition_coefficients(free), rtol=0, atol=0)
    ref, counts = run_reference(integrator, system, q0, p0, dirs, checkpoints)
    osys = orc.EuclidSystem(target, metric_kind, metric)
    for k, s in enumerate(checkpoints):
        for c in range(q0.shape[0]):
            q, p = orc.composition_steps(osys, q0[c], p0[c], dirs[c] * h, s, free, initial_h1)
            check_close(f"{name} q@{s}", q, ref["q_out"][k, c], 1e-13 * max(1, s))
            check_close(f"{name} p@{s}", p, ref["p_out"][k, c], 1e-13 * max(1, s))
            check_close(f"{name} h@{s}", np.array(osys.h(q, p)), ref["h_out"][k, c], 1e-12)
    return dict(
        kind="symcomp", target=target.tid, target_params=target.params(), metric_kind=metric_kind,
        metric=np.zeros(0) if metric is None else np.asarray(metric), q0=q0, p0=p0, dir=dirs,
        step_size=h, checkpoints=np.array(checkpoints), free_coefficients=np.array(free, dtype=np.float64),
        initial_h1_flow_step=int(bool(initial_h1)), **ref,
    ), counts


def midpoint_case(name, ref_system, osys, extra, q0, p0, dirs, h, checkpoints, fp_solver=0, norm=0,
                  fp_kwargs=None):
    """ImplicitMidpointIntegrator (integrators.py:547-681) on a Euclidean or dense-Riemannian system."""
    fp_kwargs = fp_kwargs or {}
    ref_solvers = {0: mici.solvers.solve_fixed_point_direct, 1: mici.solvers.solve_fixed_point_steffensen}
    ref_norms = {0: mici.solvers.maximum_norm, 1: mici.solvers.euclidean_norm}
    ref_kwargs = dict(fp_kwargs)
    ref_kwargs["norm"] = ref_norms[norm]
    integrator = mici.integrators.ImplicitMidpointIntegrator(
        ref_system, h, reverse_check_norm=ref_norms[norm], fixed_point_solver=ref_solvers[fp_solver],
        fixed_point_solver_kwargs=ref_kwargs)
    ref, counts = run_reference(integrator, ref_system, q0, p0, dirs, checkpoints)
    okw = dict(fp_solver=orc.FP_SOLVERS[fp_solver], rev_norm=orc.NORMS[norm],
               fp_kwargs=dict(fp_kwargs, norm=orc.NORMS[norm]))
    n = q0.shape[0]
    for k, s in enumerate(checkpoints):
        for c in range(n):
            q, p, st, nd = orc.implicit_midpoint_steps(osys, q0[c], p0[c], dirs[c] * h, s, **okw)
            assert st == (ref["status"][c] if ref["n_done"][c] < s else 0) or ref["n_done"][c] >= s, (name, st)
            assert nd == min(s, ref["n_done"][c]), (name, nd, ref["n_done"][c])
            check_close(f"{name} q@{s}", q, ref["q_out"][k, c], 1e-9)
            check_close(f"{name} p@{s}", p, ref["p_out"][k, c], 1e-9)
    full = dict(convergence_tol=1e-9, divergence_tol=1e10, max_iters=100)
    full.update(fp_kwargs)
    return dict(kind="midpoint", q0=q0, p0=p0, dir=dirs, step_size=h, checkpoints=np.array(checkpoints),
                fp_solver=fp_solver, norm=norm, fp_conv_tol=full["convergence_tol"],
                fp_div_tol=full["divergence_tol"], fp_max_iters=full["max_iters"], **extra, **ref), counts


def adapt_case(name, kind, ref_system, ref_integrator, adapter, q0, n_step, n_iters, seed0, extra):
    """DualAveragingStepSizeAdapter (adapters.py:174-389) driving momentum refresh + Metropolis static
    transitions the way the sampler's warm-up does (samplers.py: initial momentum, adapter.initialize, then
    per iteration momentum transition, integration transition, adapter.update), one chain at a time."""
    from oracle import adapters as oad
    from oracle import transitions as otr
    n, d = q0.shape
    mom_tr = mici.transitions.IndependentMomentumTransition(ref_system)
    int_tr = mici.transitions.MetropolisStaticIntegrationTransition(ref_system, ref_integrator, n_step)
    ref_adapter = mici.adapters.DualAveragingStepSizeAdapter()
    z_init = np.zeros((n, d))
    z = np.zeros((n_iters, n, d))
    u = np.full((n_iters, n), np.nan)
    init_step = np.zeros(n)
    step_sizes = np.zeros((n_iters, n))   # step size set by update() after iteration t
    accept = np.zeros((n_iters, n))
    q_final = np.zeros((n, d))
    adapt_states = []
    for c in range(n):
        rng = RecordingRng(seed0 + c)
        state = ChainState(pos=q0[c].copy(), mom=None, dir=1)
        state.mom = ref_system.sample_momentum(state, rng)
        z_init[c] = rng.log[0][1]
        adapt_state = ref_adapter.initialize(state, int_tr)
        init_step[c] = ref_integrator.step_size
        # oracle replay of the search
        p_init = adapter.sample_momentum(q0[c], z_init[c])
        check_close(f"{name} init step c{c}", oad.find_init_step_size(adapter, q0[c], p_init, 1), init_step[c], 1e-15)
        ost = oad.initial_state(init_step[c])
        oq, odir, oeps = q0[c].copy(), 1, init_step[c]
        for t in range(n_iters):
            rng.log.clear()
            state, _ = mom_tr.sample(state, rng)
            state, stats = int_tr.sample(state, rng)
            ref_adapter.update(adapt_state, state, stats, int_tr)
            z[t, c] = rng.log[0][1]
            if len(rng.log) == 2:
                u[t, c] = rng.log[1][1]
            step_sizes[t, c] = ref_integrator.step_size
            accept[t, c] = stats["accept_stat"]
            op = adapter.sample_momentum(oq, z[t, c])
            oq, op, odir, ostats = otr.metropolis_static_transition(
                adapter, oq, op, odir, oeps, n_step, lambda t=t, c=c: u[t, c])
            oeps = oad.update(ost, ostats["accept_stat"])
            check_close(f"{name} step size t{t} c{c}", oeps, step_sizes[t, c], 1e-9)
            check_close(f"{name} q t{t} c{c}", oq, state.pos, 1e-8)
        q_final[c] = state.pos
        adapt_states.append(adapt_state)
        assert abs(ost["smoothed_log_step_size"] - adapt_state["smoothed_log_step_size"]) < 1e-9
    ref_adapter.finalize(adapt_states, None, int_tr, None)
    final_step = ref_integrator.step_size
    print(f"   {name}: init step sizes {init_step.tolist()}, final {final_step:.4f}, mean accept {accept.mean():.2f}")
    return dict(kind=kind, q0=q0, z_init=z_init, z=z, u=u, init_step_size=init_step, step_sizes=step_sizes,
                accept_stat=accept, q_final=q_final, final_step_size=final_step, n_step=n_step,
                smoothed_log_step_size=np.array([s["smoothed_log_step_size"] for s in adapt_states]),
                status=np.zeros(n, dtype=np.int32), n_done=np.zeros(n, dtype=np.int32), **extra), collections.Counter()


class RecordingRng:
    """Wraps a numpy Generator and logs what the reference draws (transition fixtures)."""

    def __init__(self, seed):
        self._rng = np.random.default_rng(seed)
        self.log = []

    def standard_normal(self, size=None):
        v = self._rng.standard_normal(size)
        self.log.append(("z", np.array(v, dtype=np.float64)))
        return v

    def normal(self, size=None):  # RiemannianMetricSystem.sample_momentum, systems.py:1402
        v = self._rng.normal(size=size)
        self.log.append(("z", np.array(v, dtype=np.float64)))
        return v

    def integers(self, *a, **k):
        v = self._rng.integers(*a, **k)
        self.log.append(("n", int(v)))
        return v

    def uniform(self, *a, **k):
        v = self._rng.uniform(*a, **k)
        self.log.append(("u", float(v)))
        return v


def transition_case(name, kind, ref_system, ref_integrator, adapter, q0, n_step, n_transitions, seed0, extra):
    """IndependentMomentumTransition + MetropolisStaticIntegrationTransition (transitions.py:129-142,
    275-352), several successive transitions per chain, reference vs oracle/transitions.py."""
    from oracle import transitions as otr
    n, d = q0.shape
    mom_tr = mici.transitions.IndependentMomentumTransition(ref_system)
    int_tr = mici.transitions.MetropolisStaticIntegrationTransition(ref_system, ref_integrator, n_step)
    z = np.zeros((n_transitions, n, d))
    u = np.full((n_transitions, n), np.nan)  # NaN = the reference drew no uniform (integration error)
    q_out = np.zeros((n_transitions, n, d))
    p_out = np.zeros((n_transitions, n, d))
    dir_out = np.zeros((n_transitions, n), dtype=np.int8)
    stat = {k: np.zeros((n_transitions, n)) for k in
            ("n_step", "accept_stat", "metrop_accept_prob", "convergence_error", "non_reversible_step")}
    step_size = float(ref_integrator.step_size)
    for c in range(n):
        rng = RecordingRng(seed0 + c)
        state = ChainState(pos=q0[c].copy(), mom=None, dir=1)
        oq, odir = q0[c].copy(), 1
        for t in range(n_transitions):
            rng.log.clear()
            state, _ = mom_tr.sample(state, rng)
            state, stats = int_tr.sample(state, rng)
            kinds = [k for k, _ in rng.log]
            assert kinds in (["z"], ["z", "u"]), kinds
            z[t, c] = rng.log[0][1]
            if len(rng.log) == 2:
                u[t, c] = rng.log[1][1]
            q_out[t, c], p_out[t, c], dir_out[t, c] = state.pos, state.mom, state.dir
            for k in stat:
                stat[k][t, c] = float(stats[k])
            # oracle replay
            op = adapter.sample_momentum(oq, z[t, c])
            drew = []

            def draw(t=t, c=c, drew=drew):
                drew.append(1)
                assert not np.isnan(u[t, c]), "oracle draws a uniform the reference did not"
                return u[t, c]

            oq, op, odir, ost = otr.metropolis_static_transition(adapter, oq, op, odir, step_size, n_step, draw)
            assert len(drew) == (0 if np.isnan(u[t, c]) else 1)
            check_close(f"{name} q t{t} c{c}", oq, state.pos, 1e-11)
            check_close(f"{name} p t{t} c{c}", op, state.mom, 1e-11)
            assert odir == state.dir
            for k in stat:
                check_close(f"{name} {k} t{t} c{c}", float(ost[k]), stat[k][t, c], 1e-10)
    print(f"   {name}: accepted {np.mean(stat['accept_stat']):.2f} mean accept_stat, "
          f"{int(np.sum(np.isnan(u)))} integration errors, mean n_step {np.mean(stat['n_step']):.2f}")
    return dict(kind=kind, q0=q0, z=z, u=u, q_out=q_out, p_out=p_out, dir_out=dir_out, n_step=n_step,
                step_size=step_size, status=np.zeros(n, dtype=np.int32), n_done=np.zeros(n, dtype=np.int32),
                **{f"stat_{k}": v for k, v in stat.items()}, **extra), collections.Counter()


def riemann_case(name, target, rmetric, softabs_coeff, q0, p0, dirs, h, checkpoints,
                 fp_solver=0, norm=0, fp_kwargs=None):
    fp_kwargs = fp_kwargs or {}
    if softabs_coeff is None:
        system = mici.systems.DenseRiemannianMetricSystem(
            neg_log_dens=target.neg_log_dens, grad_neg_log_dens=target.grad,
            metric_func=rmetric.metric_func, vjp_metric_func=rmetric.vjp_metric_func,
        )
    else:
        system = mici.systems.SoftAbsRiemannianMetricSystem(
            neg_log_dens=target.neg_log_dens, grad_neg_log_dens=target.grad,
            hess_neg_log_dens=target.hess, mtp_neg_log_dens=target.mtp,
            softabs_coeff=softabs_coeff,
        )
    ref_solvers = {0: mici.solvers.solve_fixed_point_direct,
                   1: mici.solvers.solve_fixed_point_steffensen}
    ref_norms = {0: mici.solvers.maximum_norm, 1: mici.solvers.euclidean_norm}
    ref_kwargs = dict(fp_kwargs)
    ref_kwargs["norm"] = ref_norms[norm]
    iters = collections.Counter()

    def counting_solver(func, x0, **kw):
        def f(x):
            iters["fp_iters"] += 1
            return func(x)
        iters["fp_solves"] += 1
        return ref_solvers[fp_solver](f, x0, **kw)

    integrator = mici.integrators.ImplicitLeapfrogIntegrator(
        system, h, fixed_point_solver=counting_solver, fixed_point_solver_kwargs=ref_kwargs,
        reverse_check_norm=ref_norms[norm],
    )
    ref, counts = run_reference(integrator, system, q0, p0, dirs, checkpoints)
    counts.update(iters)
    # oracle cross-check
    ocount = orc.Counters()
    osys = orc.RiemannianSystem(target, rmetric, softabs_coeff, ocount)
    okw = dict(fp_solver=orc.FP_SOLVERS[fp_solver], rev_norm=orc.NORMS[norm],
               fp_kwargs=dict(fp_kwargs, norm=orc.NORMS[norm]))
    n_max = max(checkpoints)
    for c in range(q0.shape[0]):
        for k, s in enumerate(checkpoints):
            q, p, st, nd = orc.implicit_leapfrog_steps(osys, q0[c], p0[c], dirs[c] * h, s, **okw)
            tol = 2e-11 if softabs_coeff is None else 1e-9
            check_close(f"{name} q@{s} chain {c}", q, ref["q_out"][k, c], tol)
            check_close(f"{name} p@{s} chain {c}", p, ref["p_out"][k, c], tol)
            if s == n_max:
                assert st == ref["status"][c], (name, c, st, ref["status"][c])
                assert nd == ref["n_done"][c], (name, c, nd, ref["n_done"][c])
    # iteration counts must agree exactly when summed over the max-checkpoint run
    ocount.clear()
    for c in range(q0.shape[0]):
        orc.implicit_leapfrog_steps(osys, q0[c], p0[c], dirs[c] * h, n_max, **okw)
    assert ocount.get("fp_iters", 0) == counts["fp_iters"], (name, dict(ocount), dict(counts))
    n_fact = counts.get("metric", 0)
    # (the reference run has the initial metric cached by the h0 evaluation: one per chain)
    assert ocount.get("metric", 0) == n_fact + q0.shape[0], (name, dict(ocount), dict(counts))
    return dict(
        kind="riemann", target=target.tid, target_params=target.params(),
        rmetric=(mdl.RMETRIC_SOFTABS if softabs_coeff is not None else rmetric.mid),
        rmetric_params=(np.array([softabs_coeff]) if softabs_coeff is not None
                        else rmetric.params()),
        q0=q0, p0=p0, dir=dirs, step_size=h, checkpoints=np.array(checkpoints),
        fp_solver=fp_solver, norm=norm,
        fp_conv_tol=fp_kwargs.get("convergence_tol", 1e-9),
        fp_div_tol=fp_kwargs.get("divergence_tol", 1e10),
        fp_max_iters=fp_kwargs.get("max_iters", 100),
        **ref,
    ), counts


def oracle_constrained_system(target, constraint, metric_kind, metric, variant):
    if variant == "gaussian":
        return orc.GaussianConstrainedSystem(target, constraint, metric_kind, metric)
    return orc.ConstrainedSystem(target, constraint, metric_kind, metric,
                                 dens_wrt_hausdorff=(variant == "hausdorff"))


def constrained_case(name, target, constraint, metric_kind, metric, q0, p0, dirs, h,
                     checkpoints, n_inner=1, proj_solver=0, variant="hausdorff"):
    ref_metric = None if metric_kind == mdl.METRIC_IDENTITY else np.array(metric)
    if variant == "gaussian":
        system = mici.systems.GaussianDenseConstrainedEuclideanMetricSystem(
            neg_log_dens=target.neg_log_dens, grad_neg_log_dens=target.grad,
            constr=constraint.constr, jacob_constr=constraint.jacob_constr,
            mhp_constr=constraint.mhp_constr, metric=ref_metric,
        )
    else:
        system = mici.systems.DenseConstrainedEuclideanMetricSystem(
            neg_log_dens=target.neg_log_dens, grad_neg_log_dens=target.grad,
            constr=constraint.constr, jacob_constr=constraint.jacob_constr, metric=ref_metric,
            **({} if variant == "hausdorff" else dict(dens_wrt_hausdorff=False, mhp_constr=constraint.mhp_constr)),
        )
    ref_proj = {0: mici.solvers.solve_projection_onto_manifold_newton,
                1: mici.solvers.solve_projection_onto_manifold_quasi_newton,
                2: mici.solvers.solve_projection_onto_manifold_newton_with_line_search}[proj_solver]
    integrator = mici.integrators.ConstrainedLeapfrogIntegrator(
        system, h, n_inner_step=n_inner, projection_solver=ref_proj)
    ref, counts = run_reference(integrator, system, q0, p0, dirs, checkpoints)
    osys = oracle_constrained_system(target, constraint, metric_kind, metric, variant)
    n_max = max(checkpoints)
    for c in range(q0.shape[0]):
        for k, s in enumerate(checkpoints):
            q, p, st, nd = orc.constrained_leapfrog_steps(
                osys, q0[c], p0[c], dirs[c] * h, s, n_inner_step=n_inner, proj_solver=proj_solver)
            check_close(f"{name} q@{s} chain {c}", q, ref["q_out"][k, c], 1e-10)
            check_close(f"{name} p@{s} chain {c}", p, ref["p_out"][k, c], 1e-10)
            if variant != "hausdorff" and nd == s:
                check_close(f"{name} h@{s} chain {c}", np.array(osys.h(q, p)), ref["h_out"][k, c], 1e-10)
            if s == n_max:
                assert st == ref["status"][c], (name, c, st, ref["status"][c])
                assert nd == ref["n_done"][c], (name, c, nd, ref["n_done"][c])
    return dict(
        kind="constrained", target=target.tid, target_params=target.params(),
        constr=constraint.cid, constr_params=constraint.params(), metric_kind=metric_kind,
        metric=np.zeros(0) if metric is None else np.asarray(metric), q0=q0, p0=p0, dir=dirs,
        step_size=h, checkpoints=np.array(checkpoints), n_inner=n_inner, proj_solver=proj_solver,
        **({} if variant == "hausdorff" else dict(variant=variant)),
        **ref,
    ), counts


def project_momentum(osys, q, p):
    out = np.empty_like(p)
    for c in range(q.shape[0]):
        out[c] = osys.project_onto_cotangent_space(p[c].copy(), osys.constraint.jacob_constr(q[c]))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--only", default=None)
    ap.add_argument("--check", action="store_true",
                    help="regenerate every case into a temporary directory and compare it with the committed "
                         "fixture in --out (same keys, shapes, dtypes; values to 1e-11 relative); nothing is written")
    args = ap.parse_args()
    committed = args.out
    if args.check:
        import tempfile
        tmpdir = tempfile.TemporaryDirectory()
        args.out = tmpdir.name
    os.makedirs(args.out, exist_ok=True)
    rng = np.random.default_rng(SEED)
    cases = {}

    def dirs_for(n):
        d = np.ones(n, dtype=np.int8)
        d[1::3] = -1
        return d

    # ---- explicit leapfrog, Euclidean metric (BASELINE configs c1 / c2) -------------------------
    def add_euclid(name, target, mk, metric, n, h, cps):
        d = target.dim
        q0 = rng.standard_normal((n, d))
        z = rng.standard_normal((n, d))
        osys = orc.EuclidSystem(target, mk, metric)
        p0 = np.stack([osys.msqrt(zz) for zz in z])
        cases[name] = lambda: euclid_case(name, target, mk, metric, q0, p0, dirs_for(n), h, cps)

    add_euclid("euclid_c1_iso_d32", mdl.GaussIso(32), mdl.METRIC_IDENTITY, None, 4, 0.1, [1, 5, 20])
    prec128 = np.exp(-0.2 * rng.standard_normal(128))
    add_euclid("euclid_c2_diag_d128", mdl.GaussDiag(prec128), mdl.METRIC_IDENTITY, None, 8, 0.05,
               [1, 5, 100])
    add_euclid("euclid_c2_diag_diagmetric_d128", mdl.GaussDiag(prec128), mdl.METRIC_DIAG,
               np.exp(0.1 * rng.standard_normal(128)), 8, 0.05, [1, 5, 100])
    P128 = mdl.make_spd(128, rng)
    add_euclid("euclid_c2_dense_d128", mdl.GaussDense(P128), mdl.METRIC_IDENTITY, None, 8, 0.05,
               [1, 5, 20, 100])
    add_euclid("euclid_c2_dense_densemetric_d128", mdl.GaussDense(P128), mdl.METRIC_DENSE, P128,
               8, 0.05, [1, 5, 100])
    P20 = mdl.make_spd(20, rng)
    add_euclid("euclid_dense_d20_ragged", mdl.GaussDense(P20), mdl.METRIC_DENSE,
               mdl.make_spd(20, rng), 5, 0.1, [1, 7])
    for size in (1, 2, 5):  # the reference's own test sizes (tests/test_integrators.py:20-22)
        eigval = np.exp(0.1 * rng.standard_normal(size))
        eigvec = np.linalg.qr(rng.standard_normal((size, size)))[0]
        dense = (eigvec * eigval) @ eigvec.T
        add_euclid(f"euclid_quartic_dense_d{size}", mdl.Poly(size, 0.0, 1.0), mdl.METRIC_DENSE,
                   dense, 5, 0.05, [1, 5, 20])
        add_euclid(f"euclid_quadratic_diag_d{size}", mdl.Poly(size, 1.0, 0.0), mdl.METRIC_DIAG,
                   eigval, 5, 0.25, [1, 5, 20])
    add_euclid("euclid_banana_d16", mdl.Banana(16), mdl.METRIC_IDENTITY, None, 4, 0.02, [1, 10])

    # ---- implicit leapfrog, dense Riemannian metric (c3a, c4) -----------------------------------------
    def add_riemann(name, target, rmetric, coeff, n, h, cps, qscale=1.0, r=None, **kw):
        d = target.dim
        q0 = qscale * (rng if r is None else r).standard_normal((n, d))
        z = (rng if r is None else r).standard_normal((n, d))
        osys = orc.RiemannianSystem(target, rmetric, coeff)
        p0 = np.stack([osys.sample_momentum(orc._State(q0[c], None), z[c]) for c in range(n)])
        cases[name] = lambda: riemann_case(name, target, rmetric, coeff, q0, p0, dirs_for(n), h,
                                           cps, **kw)

    B64 = mdl.make_spd(64, rng)
    add_riemann("riemann_c3_rank1_banana_d64", mdl.Banana(64), mdl.Rank1Metric(B64), None, 6,
                0.02, [1, 5, 20])
    add_riemann("riemann_rank1_poly_d64", mdl.Poly(64, 1.0, 1.0 / 3.0), mdl.Rank1Metric(B64),
                None, 4, 0.05, [1, 10])
    B8 = mdl.make_spd(8, rng)
    add_riemann("riemann_rank1_poly_d8", mdl.Poly(8, 1.0, 1.0 / 3.0), mdl.Rank1Metric(B8), None,
                6, 0.1, [1, 5, 20])
    add_riemann("riemann_rank1_poly_d8_l2_steffensen", mdl.Poly(8, 1.0, 1.0 / 3.0),
                mdl.Rank1Metric(B8), None, 6, 0.1, [1, 5], fp_solver=1, norm=1)
    for size in (1, 2, 5):  # dense twin of tests/test_integrators.py:492-516
        add_riemann(f"riemann_diagquad_poly_d{size}", mdl.Poly(size, 1.0, 1.0 / 3.0),
                    mdl.DiagQuadMetric(size), None, 5, 0.1, [1, 5, 20])
    B256 = mdl.make_spd(256, rng)
    # round 4: 8 chains x 10 steps instead of 2 x 3.  The base matrix keeps its place in the shared
    # stream; the two (2, 256) draws the round-1 case took from it are still consumed, so every later round-1 case keeps
    # its inputs, and the new chains come from the case's own stream.
    rng.standard_normal((2, 256))
    rng.standard_normal((2, 256))
    add_riemann("riemann_c4_rank1_banana_d256", mdl.Banana(256), mdl.Rank1Metric(B256), None, 8,
                0.01, [1, 3, 10], r=case_rng("riemann_c4_rank1_banana_d256"))
    # failure paths: large steps / tiny iteration budgets
    add_riemann("riemann_fail_bigstep_d8", mdl.Poly(8, 1.0, 1.0 / 3.0), mdl.Rank1Metric(B8), None,
                8, 1.5, [1, 4], qscale=2.0)
    add_riemann("riemann_fail_maxiters_d8", mdl.Poly(8, 1.0, 1.0 / 3.0), mdl.Rank1Metric(B8), None,
                6, 0.3, [1, 3], fp_kwargs=dict(max_iters=3))
    add_riemann("riemann_fail_diagquad_d5", mdl.Poly(5, 1.0, 1.0 / 3.0), mdl.DiagQuadMetric(5), None,
                8, 1.2, [1, 4], qscale=2.0)

    add_riemann("riemann_fail_mixed_rank1_d8", mdl.Poly(8, 1.0, 1.0 / 3.0), mdl.Rank1Metric(B8),
                None, 12, 0.6, [1, 3, 6], qscale=2.0)
    add_riemann("riemann_fail_mixed_diagquad_d5", mdl.Poly(5, 1.0, 1.0 / 3.0),
                mdl.DiagQuadMetric(5), None, 12, 0.3, [1, 3, 6], qscale=2.0)

    def add_nonfinite():
        name = "riemann_fail_nonfinite_d8"
        q0 = rng.standard_normal((4, 8))
        p0 = rng.standard_normal((4, 8))
        q0[1, :] = 1e200      # metric overflows -> "Array is not finite" outside a solver
        q0[2, 3] = np.nan     # NaN position -> same
        cases[name] = lambda: riemann_case(name, mdl.Poly(8, 1.0, 1.0 / 3.0), mdl.Rank1Metric(B8),
                                           None, q0, p0, dirs_for(4), 0.1, [1, 3])
    add_nonfinite()

    # ---- SoftAbs (c3b) ------------------------------------------------------------------------------
    add_riemann("softabs_c3_funnel_d64", mdl.Funnel(np.linspace(0.5, 2.0, 63)), None, 1.0, 3,
                0.02, [1, 4])
    add_riemann("softabs_funnel_d8", mdl.Funnel(np.linspace(0.5, 2.0, 7)), None, 1.0, 6,
                0.05, [1, 5, 20])
    add_riemann("softabs_poly_d5", mdl.Poly(5, 1.0, 1.0 / 3.0), None, 1.0, 5, 0.1, [1, 5, 20])

    # ---- constrained leapfrog (c5 + the reference's own constrained test systems) ------------------
    def add_constrained(name, target, constraint, mk, metric, q0, h, cps, n_inner=1, proj_solver=0,
                        variant="hausdorff", r=None):
        n, d = q0.shape
        z = (rng if r is None else r).standard_normal((n, d))
        osys = orc.ConstrainedSystem(target, constraint, mk, metric)
        p0 = project_momentum(osys, q0, np.stack([osys.msqrt(zz) for zz in z]))
        cases[name] = lambda: constrained_case(name, target, constraint, mk, metric, q0, p0,
                                               dirs_for(n), h, cps, n_inner, proj_solver, variant)

    add_constrained("constrained_c5_torus", mdl.Torus(), mdl.TorusConstr(), mdl.METRIC_IDENTITY,
                    None, mdl.torus_init(16, rng), 0.1, [1, 5, 20, 100])
    add_constrained("constrained_torus_inner2", mdl.Torus(), mdl.TorusConstr(),
                    mdl.METRIC_IDENTITY, None, mdl.torus_init(6, rng), 0.2, [1, 10], n_inner=2)
    add_constrained("constrained_torus_fail_bigstep", mdl.Torus(), mdl.TorusConstr(),
                    mdl.METRIC_IDENTITY, None, mdl.torus_init(16, rng), 1.2, [1, 5])
    for size in (2, 5):  # tests/test_integrators.py:519-565
        eigval = np.exp(0.1 * rng.standard_normal(size))
        eigvec = np.linalg.qr(rng.standard_normal((size, size)))[0]
        dense = (eigvec * eigval) @ eigvec.T
        theta = rng.uniform(size=5) * 2 * np.pi
        qc = np.concatenate([np.cos(theta)[:, None], np.sin(theta)[:, None],
                             rng.standard_normal((5, size - 2))], 1)
        add_constrained(f"constrained_circle_dense_d{size}", mdl.Poly(size, 0.0, 0.5),
                        mdl.CircleConstr(), mdl.METRIC_DENSE, dense, qc, 0.1, [1, 5, 20])
        add_constrained(f"constrained_circle_diag_d{size}", mdl.Poly(size, 0.0, 0.5),
                        mdl.CircleConstr(), mdl.METRIC_DIAG, eigval, qc.copy(), 0.1, [1, 5, 20])
        ql = np.concatenate([np.zeros((5, 1)), rng.standard_normal((5, size - 1))], 1)
        add_constrained(f"constrained_linear_dense_d{size}", mdl.Poly(size, 1.0, 0.0),
                        mdl.FirstCoordConstr(), mdl.METRIC_DENSE, dense, ql, 0.1, [1, 5, 20])

    # ---- the other two projection solvers (added last: earlier fixtures keep their random streams) ----
    for ps, tag in ((1, "quasi"), (2, "linesearch")):
        add_constrained(f"constrained_torus_{tag}", mdl.Torus(), mdl.TorusConstr(),
                        mdl.METRIC_IDENTITY, None, mdl.torus_init(12, rng), 0.1, [1, 5, 20],
                        proj_solver=ps)
        add_constrained(f"constrained_torus_{tag}_fail_bigstep", mdl.Torus(), mdl.TorusConstr(),
                        mdl.METRIC_IDENTITY, None, mdl.torus_init(12, rng), 1.0, [1, 4],
                        proj_solver=ps)
        eigval = np.exp(0.1 * rng.standard_normal(5))
        eigvec = np.linalg.qr(rng.standard_normal((5, 5)))[0]
        dense = (eigvec * eigval) @ eigvec.T
        theta = rng.uniform(size=5) * 2 * np.pi
        qc = np.concatenate([np.cos(theta)[:, None], np.sin(theta)[:, None],
                             rng.standard_normal((5, 3))], 1)
        add_constrained(f"constrained_circle_dense_d5_{tag}", mdl.Poly(5, 0.0, 0.5),
                        mdl.CircleConstr(), mdl.METRIC_DENSE, dense, qc, 0.1, [1, 5, 20],
                        proj_solver=ps)

    # ---- symmetric composition integrators on Euclidean systems ---------------------
    # (registered last: the cases above keep the random stream they were generated with)
    def add_symcomp(name, target, mk, metric, n, h, cps, free, initial_h1=True):
        d = target.dim
        q0 = rng.standard_normal((n, d))
        z = rng.standard_normal((n, d))
        osys = orc.EuclidSystem(target, mk, metric)
        p0 = np.stack([osys.msqrt(zz) for zz in z])
        cases[name] = lambda: symcomp_case(name, target, mk, metric, q0, p0, dirs_for(n), h, cps, free,
                                           initial_h1)

    Pc = mdl.make_spd(24, rng)
    Mc = mdl.make_spd(24, rng)
    for stages in (2, 3, 4):
        add_symcomp(f"symcomp_bcss{stages}_dense_d24", mdl.GaussDense(Pc), mdl.METRIC_DENSE, Mc, 5, 0.2,
                    [1, 5, 20], stages)
        add_symcomp(f"symcomp_bcss{stages}_quartic_diag_d5", mdl.Poly(5, 0.0, 1.0), mdl.METRIC_DIAG,
                    np.exp(0.1 * rng.standard_normal(5)), 4, 0.1, [1, 10], stages)
    add_symcomp("symcomp_bcss3_iso_d128", mdl.GaussIso(128), mdl.METRIC_IDENTITY, None, 4, 0.3, [1, 20], 3)
    add_symcomp("symcomp_leapfrog_as_composition_d16", mdl.Banana(16), mdl.METRIC_IDENTITY, None, 4, 0.02,
                [1, 10], ())
    add_symcomp("symcomp_free2_h2first_d7", mdl.Poly(7, 1.0, 0.5), mdl.METRIC_IDENTITY, None, 4, 0.1,
                [1, 10], (0.2, 0.3), initial_h1=False)
    add_symcomp("symcomp_free1_h2first_dense_d3", mdl.GaussDense(mdl.make_spd(3, rng)), mdl.METRIC_DENSE,
                mdl.make_spd(3, rng), 4, 0.1, [1, 10], (0.21,), initial_h1=False)

    # ---- momentum refresh + Metropolis static-integration transitions ---------------
    from oracle import transitions as otr

    def model_keys(target, mk, metric, **more):
        return dict(target=target.tid, target_params=target.params(), metric_kind=mk,
                    metric=np.zeros(0) if metric is None else np.asarray(metric), **more)

    def add_transition_euclid(name, target, mk, metric, n, h, n_step, n_tr, seed0, free=None):
        q0 = rng.standard_normal((n, target.dim))

        def make():
            rsys = mici.systems.EuclideanMetricSystem(
                neg_log_dens=target.neg_log_dens, grad_neg_log_dens=target.grad,
                metric=None if mk == mdl.METRIC_IDENTITY else np.array(metric))
            if free is None:
                rint = mici.integrators.LeapfrogIntegrator(rsys, h)
            else:
                rint = mici.integrators.SymmetricCompositionIntegrator(rsys, free, step_size=h)
            ad = otr.euclid_adapter(orc.EuclidSystem(target, mk, metric), free)
            return transition_case(name, "transition_euclid", rsys, rint, ad, q0, n_step, n_tr, seed0,
                                   model_keys(target, mk, metric, free_coefficients=np.array(
                                       [] if free is None else free, dtype=np.float64),
                                       composition=int(free is not None)))
        cases[name] = make

    Pt = mdl.make_spd(16, rng)
    add_transition_euclid("transition_euclid_dense_d16", mdl.GaussDense(Pt), mdl.METRIC_DENSE,
                          mdl.make_spd(16, rng), 6, 0.35, 5, 6, 1000)
    add_transition_euclid("transition_euclid_quartic_d5_bigstep", mdl.Poly(5, 0.0, 1.0), mdl.METRIC_IDENTITY, None,
                          6, 0.9, 4, 6, 2000)
    add_transition_euclid("transition_euclid_bcss3_d8", mdl.Poly(8, 1.0, 0.5), mdl.METRIC_DIAG,
                          np.exp(0.2 * rng.standard_normal(8)), 4, 0.5, 3, 5, 3000,
                          free=orc.BCSS_FREE_COEFFICIENTS[3])

    def add_transition_riemann(name, target, rmetric, n, h, n_step, n_tr, seed0):
        q0 = rng.standard_normal((n, target.dim))

        def make():
            rsys = mici.systems.DenseRiemannianMetricSystem(
                neg_log_dens=target.neg_log_dens, grad_neg_log_dens=target.grad,
                metric_func=rmetric.metric_func, vjp_metric_func=rmetric.vjp_metric_func)
            rint = mici.integrators.ImplicitLeapfrogIntegrator(rsys, h)
            ad = otr.riemann_adapter(orc.RiemannianSystem(target, rmetric, None))
            return transition_case(name, "transition_riemann", rsys, rint, ad, q0, n_step, n_tr, seed0,
                                   dict(target=target.tid, target_params=target.params(), rmetric=rmetric.mid,
                                        rmetric_params=rmetric.params()))
        cases[name] = make

    add_transition_riemann("transition_riemann_rank1_banana_d8", mdl.Banana(8),
                           mdl.Rank1Metric(mdl.make_spd(8, rng)), 5, 0.05, 4, 5, 4000)
    add_transition_riemann("transition_riemann_diagquad_poly_d5_bigstep", mdl.Poly(5, 1.0, 1.0 / 3.0),
                           mdl.DiagQuadMetric(5), 6, 0.45, 3, 6, 5000)

    def add_transition_constrained(name, n, h, n_step, n_tr, seed0):
        q0 = mdl.torus_init(n, rng)
        target, constraint = mdl.Torus(), mdl.TorusConstr()

        def make():
            rsys = mici.systems.DenseConstrainedEuclideanMetricSystem(
                neg_log_dens=target.neg_log_dens, grad_neg_log_dens=target.grad,
                constr=constraint.constr, jacob_constr=constraint.jacob_constr)
            rint = mici.integrators.ConstrainedLeapfrogIntegrator(rsys, h)
            ad = otr.constrained_adapter(orc.ConstrainedSystem(target, constraint))
            return transition_case(name, "transition_constrained", rsys, rint, ad, q0, n_step, n_tr, seed0,
                                   dict(target=target.tid, target_params=target.params(),
                                        constr=constraint.cid, constr_params=constraint.params(),
                                        metric_kind=mdl.METRIC_IDENTITY, metric=np.zeros(0)))
        cases[name] = make

    add_transition_constrained("transition_constrained_torus", 6, 0.3, 4, 6, 6000)
    add_transition_constrained("transition_constrained_torus_bigstep", 6, 1.2, 3, 6, 7000)

    # ---- implicit midpoint integrator ------------------------------------------------
    def add_midpoint_euclid(name, target, mk, metric, n, h, cps, **kw):
        q0 = rng.standard_normal((n, target.dim))
        osys = orc.EuclidSystem(target, mk, metric)
        p0 = np.stack([osys.msqrt(zz) for zz in rng.standard_normal((n, target.dim))])

        def make():
            rsys = mici.systems.EuclideanMetricSystem(
                neg_log_dens=target.neg_log_dens, grad_neg_log_dens=target.grad,
                metric=None if mk == mdl.METRIC_IDENTITY else np.array(metric))
            return midpoint_case(name, rsys, osys, model_keys(target, mk, metric, system="euclid"), q0, p0,
                                 dirs_for(n), h, cps, **kw)
        cases[name] = make

    def add_midpoint_riemann(name, target, rmetric, n, h, cps, qscale=1.0, **kw):
        q0 = qscale * rng.standard_normal((n, target.dim))
        osys = orc.RiemannianSystem(target, rmetric, None)
        p0 = np.stack([osys.sample_momentum(orc._State(q0[c], None), zz)
                       for c, zz in enumerate(rng.standard_normal((n, target.dim)))])

        def make():
            rsys = mici.systems.DenseRiemannianMetricSystem(
                neg_log_dens=target.neg_log_dens, grad_neg_log_dens=target.grad,
                metric_func=rmetric.metric_func, vjp_metric_func=rmetric.vjp_metric_func)
            return midpoint_case(name, rsys, osys, dict(system="riemann", target=target.tid,
                                                        target_params=target.params(), rmetric=rmetric.mid,
                                                        rmetric_params=rmetric.params()),
                                 q0, p0, dirs_for(n), h, cps, **kw)
        cases[name] = make

    for size in (1, 2, 5):  # the reference's own test systems (tests/test_integrators.py:236-253, 492-516)
        eigval = np.exp(0.1 * rng.standard_normal(size))
        eigvec = np.linalg.qr(rng.standard_normal((size, size)))[0]
        add_midpoint_euclid(f"midpoint_euclid_quartic_dense_d{size}", mdl.Poly(size, 0.0, 1.0), mdl.METRIC_DENSE,
                            (eigvec * eigval) @ eigvec.T, 4, 0.1, [1, 5, 20])
        add_midpoint_riemann(f"midpoint_riemann_diagquad_poly_d{size}", mdl.Poly(size, 1.0, 1.0 / 3.0),
                             mdl.DiagQuadMetric(size), 4, 0.1, [1, 5, 20])
    add_midpoint_euclid("midpoint_euclid_dense_d24", mdl.GaussDense(mdl.make_spd(24, rng)), mdl.METRIC_DIAG,
                        np.exp(0.2 * rng.standard_normal(24)), 4, 0.3, [1, 10])
    add_midpoint_euclid("midpoint_euclid_quartic_d5_steffensen_l2", mdl.Poly(5, 0.0, 1.0), mdl.METRIC_IDENTITY, None,
                        4, 0.2, [1, 10], fp_solver=1, norm=1)
    add_midpoint_euclid("midpoint_euclid_quartic_d3_fail_bigstep", mdl.Poly(3, 0.0, 1.0), mdl.METRIC_IDENTITY, None,
                        8, 2.5, [1, 4])
    add_midpoint_riemann("midpoint_riemann_rank1_banana_d40", mdl.Banana(40), mdl.Rank1Metric(mdl.make_spd(40, rng)),
                         4, 0.02, [1, 5])
    add_midpoint_riemann("midpoint_riemann_rank1_poly_d12", mdl.Poly(12, 1.0, 1.0 / 3.0),
                         mdl.Rank1Metric(mdl.make_spd(12, rng)), 4, 0.1, [1, 10])
    add_midpoint_riemann("midpoint_riemann_diagquad_d5_fail_bigstep", mdl.Poly(5, 1.0, 1.0 / 3.0),
                         mdl.DiagQuadMetric(5), 8, 0.9, [1, 4], qscale=1.5)

    # ---- GaussianEuclideanMetricSystem: exact h2 flow (systems.py:369-474) ------------
    def add_gauss(name, target, mk, metric, n, h, cps, integ_kind, free=()):
        q0 = rng.standard_normal((n, target.dim))
        osys = orc.GaussianEuclidSystem(target, mk, metric)
        p0 = np.stack([osys.msqrt(zz) for zz in rng.standard_normal((n, target.dim))])

        def make():
            rsys = mici.systems.GaussianEuclideanMetricSystem(
                neg_log_dens=target.neg_log_dens, grad_neg_log_dens=target.grad,
                metric=None if mk == mdl.METRIC_IDENTITY else np.array(metric))
            if integ_kind == 0:
                rint = mici.integrators.LeapfrogIntegrator(rsys, h)
            elif integ_kind == 1:
                rint = mici.integrators.SymmetricCompositionIntegrator(rsys, free, step_size=h)
            else:
                rint = mici.integrators.ImplicitMidpointIntegrator(rsys, h)
            dirs = dirs_for(n)
            ref, counts = run_reference(rint, rsys, q0, p0, dirs, cps)
            for k, s in enumerate(cps):
                for c in range(n):
                    if integ_kind == 0:
                        q, p = orc.leapfrog_steps(osys, q0[c], p0[c], dirs[c] * h, s)
                    elif integ_kind == 1:
                        q, p = orc.composition_steps(osys, q0[c], p0[c], dirs[c] * h, s, free)
                    else:
                        q, p, st, nd = orc.implicit_midpoint_steps(osys, q0[c], p0[c], dirs[c] * h, s)
                        assert st == 0 and nd == s
                    check_close(f"{name} q@{s}", q, ref["q_out"][k, c], 1e-12 * max(1, s) if integ_kind < 2 else 1e-9)
                    check_close(f"{name} p@{s}", p, ref["p_out"][k, c], 1e-12 * max(1, s) if integ_kind < 2 else 1e-9)
                    check_close(f"{name} h@{s}", np.array(osys.h(q, p)), ref["h_out"][k, c], 1e-11)
            return dict(kind="gausseuclid", integrator=integ_kind, free_coefficients=np.array(free, dtype=np.float64),
                        q0=q0, p0=p0, dir=dirs, step_size=h, checkpoints=np.array(cps),
                        **model_keys(target, mk, metric), **ref), counts
        cases[name] = make

    add_gauss("gausseuclid_leapfrog_identity_d8", mdl.Poly(8, 0.3, 0.5), mdl.METRIC_IDENTITY, None, 4, 0.3, [1, 10], 0)
    add_gauss("gausseuclid_leapfrog_diag_d8", mdl.Banana(8), mdl.METRIC_DIAG, np.exp(0.3 * rng.standard_normal(8)),
              4, 0.05, [1, 10], 0)
    add_gauss("gausseuclid_leapfrog_dense_d12", mdl.Poly(12, 0.0, 0.25), mdl.METRIC_DENSE, mdl.make_spd(12, rng),
              4, 0.3, [1, 10, 40], 0)
    add_gauss("gausseuclid_bcss3_dense_d12", mdl.GaussDense(mdl.make_spd(12, rng)), mdl.METRIC_DENSE,
              mdl.make_spd(12, rng), 4, 0.4, [1, 10], 1, orc.BCSS_FREE_COEFFICIENTS[3])
    add_gauss("gausseuclid_midpoint_diag_d5", mdl.Poly(5, 0.0, 1.0), mdl.METRIC_DIAG,
              np.exp(0.2 * rng.standard_normal(5)), 4, 0.1, [1, 10], 2)

    # ---- dens_wrt_hausdorff=False and the Gaussian split on constrained systems (systems.py:846-862,
    #      1024-1031, 1034-1184; tests/test_integrators.py:568-615).  Registered after all older cases. ---------
    for variant, tag in (("ambient", "ambient"), ("gaussian", "gauss")):
        eigval = np.exp(0.1 * rng.standard_normal(5))
        eigvec = np.linalg.qr(rng.standard_normal((5, 5)))[0]
        dense = (eigvec * eigval) @ eigvec.T
        theta = rng.uniform(size=6) * 2 * np.pi
        qc = np.concatenate([np.cos(theta)[:, None], np.sin(theta)[:, None], rng.standard_normal((6, 3))], 1)
        hh = 0.1 if variant == "ambient" else 0.05
        add_constrained(f"constrained_{tag}_torus", mdl.Torus(), mdl.TorusConstr(), mdl.METRIC_IDENTITY, None,
                        mdl.torus_init(12, rng), hh, [1, 5, 20], variant=variant)
        add_constrained(f"constrained_{tag}_torus_diag_inner2", mdl.Torus(), mdl.TorusConstr(), mdl.METRIC_DIAG,
                        np.exp(0.2 * rng.standard_normal(3)), mdl.torus_init(6, rng), hh, [1, 10], n_inner=2,
                        variant=variant)
        add_constrained(f"constrained_{tag}_torus_fail_bigstep", mdl.Torus(), mdl.TorusConstr(),
                        mdl.METRIC_IDENTITY, None, mdl.torus_init(12, rng), 1.2, [1, 5], variant=variant)
        for ps, ptag in ((0, "newton"), (1, "quasi")) + (((2, "linesearch"),) if variant == "ambient" else ()):
            add_constrained(f"constrained_{tag}_circle_dense_d5_{ptag}", mdl.Poly(5, 0.0, 0.5), mdl.CircleConstr(),
                            mdl.METRIC_DENSE, dense, qc.copy(), hh, [1, 5, 20], proj_solver=ps, variant=variant)
        ql = np.concatenate([np.zeros((5, 1)), rng.standard_normal((5, 4))], 1)
        add_constrained(f"constrained_{tag}_linear_dense_d5", mdl.Poly(5, 1.0, 0.0), mdl.FirstCoordConstr(),
                        mdl.METRIC_DENSE, dense, ql, 0.5 if variant == "gaussian" else 0.1, [1, 5, 20],
                        variant=variant)

    # ---- more than one constraint: C x C Cholesky / pivoted LU / symmetric inverses (matrices.py:1161-1188,
    #      1270-1411, 1414-1447).  Registered after all older cases. ----------------------------------------------
    def sphere_plane_init(n, d, normal):
        x = rng.standard_normal((n, d))
        x -= np.outer(x @ normal, normal) / (normal @ normal)
        return x / np.linalg.norm(x, axis=1, keepdims=True)

    def linear_init(n, a, b):
        part = np.linalg.lstsq(a, b, rcond=None)[0]
        null = np.linalg.svd(a)[2][a.shape[0]:].T
        return part + rng.standard_normal((n, null.shape[1])) @ null.T

    nrm = rng.standard_normal(4)
    dense4 = mdl.make_spd(4, rng)
    for ps, ptag in ((0, "newton"), (1, "quasi"), (2, "linesearch")):
        add_constrained(f"constrained_c2_sphereplane_d4_{ptag}", mdl.Poly(4, 0.5, 0.25), mdl.SpherePlaneConstr(nrm),
                        mdl.METRIC_DENSE, dense4, sphere_plane_init(6, 4, nrm), 0.1, [1, 5, 20], proj_solver=ps)
    add_constrained("constrained_c2_sphereplane_d4_inner2_identity", mdl.Poly(4, 0.5, 0.25),
                    mdl.SpherePlaneConstr(nrm), mdl.METRIC_IDENTITY, None, sphere_plane_init(6, 4, nrm), 0.2, [1, 10],
                    n_inner=2)
    add_constrained("constrained_c2_sphereplane_d4_fail_bigstep", mdl.Poly(4, 0.5, 0.25), mdl.SpherePlaneConstr(nrm),
                    mdl.METRIC_DENSE, dense4, sphere_plane_init(8, 4, nrm), 1.5, [1, 4])
    add_constrained("constrained_c2_sphereplane_ambient_d4", mdl.Poly(4, 0.5, 0.25), mdl.SpherePlaneConstr(nrm),
                    mdl.METRIC_DIAG, np.exp(0.3 * rng.standard_normal(4)), sphere_plane_init(6, 4, nrm), 0.1,
                    [1, 5, 20], variant="ambient")
    for ps, ptag in ((0, "newton"), (1, "quasi")):
        add_constrained(f"constrained_c2_sphereplane_gauss_d4_{ptag}", mdl.Poly(4, 0.0, 0.5),
                        mdl.SpherePlaneConstr(nrm), mdl.METRIC_DENSE, dense4, sphere_plane_init(6, 4, nrm), 0.05,
                        [1, 5, 20], proj_solver=ps, variant="gaussian")
    a2, b2 = rng.standard_normal((2, 5)), rng.standard_normal(2)
    add_constrained("constrained_c2_linear_d5", mdl.Poly(5, 1.0, 0.25), mdl.LinearConstr(a2, b2), mdl.METRIC_DENSE,
                    mdl.make_spd(5, rng), linear_init(5, a2, b2), 0.1, [1, 5, 20])
    a3, b3 = rng.standard_normal((3, 6)), rng.standard_normal(3)
    add_constrained("constrained_c3_linear_ambient_d6", mdl.Poly(6, 1.0, 0.25), mdl.LinearConstr(a3, b3),
                    mdl.METRIC_DENSE, mdl.make_spd(6, rng), linear_init(5, a3, b3), 0.1, [1, 5, 20],
                    variant="ambient")
    add_constrained("constrained_c3_linear_gauss_d6", mdl.Poly(6, 0.0, 0.25), mdl.LinearConstr(a3, b3),
                    mdl.METRIC_DENSE, mdl.make_spd(6, rng), linear_init(5, a3, b3), 0.3, [1, 5, 20],
                    variant="gaussian")

    # ---- ImplicitLeapfrogIntegrator on a plain Euclidean-metric system (tests/test_integrators.py:435-462) ---------
    def add_implicit_euclid(name, target, mk, metric, n, h, cps):
        q0 = rng.standard_normal((n, target.dim))
        esys = orc.EuclidSystem(target, mk, metric)
        p0 = np.stack([esys.msqrt(zz) for zz in rng.standard_normal((n, target.dim))])

        def make():
            rsys = mici.systems.EuclideanMetricSystem(
                neg_log_dens=target.neg_log_dens, grad_neg_log_dens=target.grad,
                metric=None if mk == mdl.METRIC_IDENTITY else np.array(metric))
            rint = mici.integrators.ImplicitLeapfrogIntegrator(rsys, h)
            dirs = dirs_for(n)
            ref, counts = run_reference(rint, rsys, q0, p0, dirs, cps)
            osys = orc.EuclidAsGeneralSystem(esys)
            for k, s in enumerate(cps):
                for c in range(n):
                    q, p, st, nd = orc.implicit_leapfrog_steps(osys, q0[c], p0[c], dirs[c] * h, s)
                    assert st == 0 and nd == s
                    check_close(f"{name} q@{s}", q, ref["q_out"][k, c], 1e-12 * max(1, s))
                    check_close(f"{name} p@{s}", p, ref["p_out"][k, c], 1e-12 * max(1, s))
            return dict(kind="impliciteuclid", q0=q0, p0=p0, dir=dirs, step_size=h, checkpoints=np.array(cps),
                        **model_keys(target, mk, metric), **ref), counts
        cases[name] = make

    add_implicit_euclid("impliciteuclid_quartic_dense_d5", mdl.Poly(5, 0.0, 1.0), mdl.METRIC_DENSE,
                        mdl.make_spd(5, rng), 5, 0.1, [1, 5, 20])
    add_implicit_euclid("impliciteuclid_gauss_dense_d24", mdl.GaussDense(mdl.make_spd(24, rng)), mdl.METRIC_IDENTITY,
                        None, 4, 0.2, [1, 10])

    # ---- 8 < D <= 16 constrained systems (capacity-16 kernels); the D = 10 unit sphere with the quasi-Newton
    #      solver is the system of the reference's adapter tests (tests/test_adapters.py:156-188) ---------------------
    def sphere_init(n, d):
        x = rng.standard_normal((n, d))
        return x / np.linalg.norm(x, axis=1, keepdims=True)

    add_constrained("constrained_sphere_d10_quasi", mdl.GaussIso(10), mdl.SphereConstr(), mdl.METRIC_IDENTITY, None,
                    sphere_init(6, 10), 0.2, [1, 5, 20], proj_solver=1)
    add_constrained("constrained_sphere_ambient_dense_d9", mdl.Poly(9, 0.5, 0.25), mdl.SphereConstr(),
                    mdl.METRIC_DENSE, mdl.make_spd(9, rng), sphere_init(5, 9), 0.1, [1, 5, 20], variant="ambient")
    a12, b12 = rng.standard_normal((3, 12)), rng.standard_normal(3)
    add_constrained("constrained_c3_linear_d12_linesearch", mdl.Poly(12, 1.0, 0.25), mdl.LinearConstr(a12, b12),
                    mdl.METRIC_DIAG, np.exp(0.2 * rng.standard_normal(12)), linear_init(5, a12, b12), 0.1, [1, 5, 20],
                    proj_solver=2)
    n16 = rng.standard_normal(16)
    add_constrained("constrained_c2_sphereplane_gauss_d16", mdl.Poly(16, 0.0, 0.5), mdl.SpherePlaneConstr(n16),
                    mdl.METRIC_DENSE, mdl.make_spd(16, rng), sphere_plane_init(5, 16, n16), 0.05, [1, 5, 20],
                    variant="gaussian")

    # ---- implicit midpoint on the team kernels (D > 64) and on the SoftAbs metric -------------------------------------
    add_midpoint_riemann("midpoint_riemann_rank1_poly_d70", mdl.Poly(70, 1.0, 1.0 / 3.0),
                         mdl.Rank1Metric(mdl.make_spd(70, rng)), 3, 0.05, [1, 4])
    add_midpoint_riemann("midpoint_riemann_rank1_banana_d100", mdl.Banana(100), mdl.Rank1Metric(mdl.make_spd(100, rng)),
                         3, 0.01, [1, 3])

    def add_midpoint_softabs(name, target, coeff, n, h, cps, qscale=0.5, r=None, **kw):
        r = rng if r is None else r
        q0 = qscale * r.standard_normal((n, target.dim))
        osys = orc.RiemannianSystem(target, None, coeff)
        p0 = np.stack([osys.sample_momentum(orc._State(q0[c], None), zz)
                       for c, zz in enumerate(r.standard_normal((n, target.dim)))])

        def make():
            rsys = mici.systems.SoftAbsRiemannianMetricSystem(
                neg_log_dens=target.neg_log_dens, grad_neg_log_dens=target.grad,
                hess_neg_log_dens=target.hess, mtp_neg_log_dens=target.mtp, softabs_coeff=coeff)
            return midpoint_case(name, rsys, osys, dict(system="softabs", target=target.tid,
                                                        target_params=target.params(), rmetric=mdl.RMETRIC_SOFTABS,
                                                        rmetric_params=np.array([coeff])),
                                 q0, p0, dirs_for(n), h, cps, **kw)
        cases[name] = make

    add_midpoint_softabs("midpoint_softabs_funnel_d8", mdl.Funnel(np.linspace(0.5, 2.0, 7)), 1.0, 4, 0.05, [1, 5])
    add_midpoint_softabs("midpoint_softabs_poly_d16", mdl.Poly(16, 1.0, 0.3), 2.0, 3, 0.05, [1, 5])

    # ---- metric adapters (adapters.py:392-644): Welford updates per chain, pairwise combination, regularisation,
    #      metric = inverse of the estimate, momenta resampled.  No integrator involved: the adapters only read pos. ---
    def add_metric_adapter(name, which, dim, n_chains, n_updates, seed0, multi=True):
        pos_seq = rng.standard_normal((n_updates, n_chains, dim)) * np.exp(0.5 * rng.standard_normal(dim)) + \
            rng.standard_normal(dim)

        def make():
            import types
            adapter = (mici.adapters.OnlineVarianceMetricAdapter() if which == "variance"
                       else mici.adapters.OnlineCovarianceMetricAdapter())
            rsys = mici.systems.EuclideanMetricSystem(neg_log_dens=lambda q: 0.5 * np.sum(q**2),
                                                      grad_neg_log_dens=lambda q: q)
            transition = types.SimpleNamespace(system=rsys)
            states = [ChainState(pos=pos_seq[0, c].copy(), mom=np.zeros(dim), dir=1) for c in range(n_chains)]
            adapt_states = [adapter.initialize(states[c], transition) for c in range(n_chains)]
            for k in range(n_updates):
                for c in range(n_chains):
                    states[c].pos = pos_seq[k, c].copy()
                    adapter.update(adapt_states[c], states[c], {}, transition)
            rngs = [RecordingRng(seed0 + c) for c in range(n_chains)]
            if multi:
                adapter.finalize(adapt_states, states, transition, rngs)
            else:
                adapter.finalize(adapt_states[0], states[0], transition, rngs[0])
            n_fin = n_chains if multi else 1
            metric = rsys.metric.array
            z = np.stack([rngs[c].log[0][1] for c in range(n_fin)])
            mom = np.stack([states[c].mom for c in range(n_fin)])
            return dict(kind="metricadapt", which=which, multi=multi, pos_seq=pos_seq, metric=np.array(metric),
                        status=np.zeros(0, dtype=np.int32), n_done=np.zeros(0, dtype=np.int32),
                        z=z, mom=mom, reg_iter_offset=adapter.reg_iter_offset, reg_scale=adapter.reg_scale), {}
        cases[name] = make

    add_metric_adapter("metricadapt_variance_d6_3chains", "variance", 6, 3, 12, 4100)
    add_metric_adapter("metricadapt_variance_d4_single", "variance", 4, 1, 9, 4200, multi=False)
    add_metric_adapter("metricadapt_covariance_d5_4chains", "covariance", 5, 4, 15, 4300)
    add_metric_adapter("metricadapt_covariance_d3_single", "covariance", 3, 1, 8, 4400, multi=False)

    # ---- correlated momentum refresh + random trajectory length (transitions.py:143-198, 355-402) ----------
    def make_corr_random():
        name = "corrmom_random_nstep_d10"
        target, d, n, n_tr, coeff, rng_range = mdl.Poly(10, 1.0, 0.5), 10, 5, 8, 0.6, (2, 7)
        metric = np.exp(0.2 * np.random.default_rng(77).standard_normal(d))
        q0 = np.random.default_rng(78).standard_normal((n, d))
        rsys = mici.systems.EuclideanMetricSystem(neg_log_dens=target.neg_log_dens, grad_neg_log_dens=target.grad,
                                                  metric=metric)
        rint = mici.integrators.LeapfrogIntegrator(rsys, 0.45)
        mom_tr = mici.transitions.CorrelatedMomentumTransition(rsys, coeff)
        int_tr = mici.transitions.MetropolisRandomIntegrationTransition(rsys, rint, rng_range)
        ad = otr.euclid_adapter(orc.EuclidSystem(target, mdl.METRIC_DIAG, metric))
        z = np.zeros((n_tr, n, d)); u = np.full((n_tr, n), np.nan); n_steps = np.zeros((n_tr, n), dtype=np.int64)
        q_out = np.zeros((n_tr, n, d)); p_out = np.zeros((n_tr, n, d)); dir_out = np.zeros((n_tr, n), dtype=np.int8)
        acc = np.zeros((n_tr, n))
        for c in range(n):
            r = RecordingRng(11000 + c)
            state = ChainState(pos=q0[c].copy(), mom=None, dir=1)
            oq, op, odir = q0[c].copy(), None, 1
            for t in range(n_tr):
                r.log.clear()
                state, _ = mom_tr.sample(state, r)
                state, stats = int_tr.sample(state, r)
                kinds = [k for k, _ in r.log]
                assert kinds in (["z", "n", "u"], ["z", "n"]), kinds
                z[t, c], n_steps[t, c] = r.log[0][1], r.log[1][1]
                if len(r.log) == 3:
                    u[t, c] = r.log[2][1]
                q_out[t, c], p_out[t, c], dir_out[t, c], acc[t, c] = state.pos, state.mom, state.dir, stats["accept_stat"]
                op = otr.correlated_momentum(ad, oq, op, z[t, c], coeff)
                oq, op, odir, ost = otr.metropolis_static_transition(ad, oq, op, odir, 0.45, int(n_steps[t, c]),
                                                                     lambda t=t, c=c: u[t, c])
                check_close(f"{name} q t{t} c{c}", oq, state.pos, 1e-11)
                check_close(f"{name} p t{t} c{c}", op, state.mom, 1e-11)
                assert odir == state.dir
        print(f"   {name}: mean accept {acc.mean():.2f}, n_step range {n_steps.min()}..{n_steps.max()}")
        return dict(kind="transition2", q0=q0, z=z, u=u, n_steps=n_steps, q_out=q_out, p_out=p_out, dir_out=dir_out,
                    accept_stat=acc, coeff=coeff, n_step_range=np.array(rng_range), step_size=0.45,
                    status=np.zeros(n, dtype=np.int32), n_done=np.zeros(n, dtype=np.int32),
                    **model_keys(target, mdl.METRIC_DIAG, metric)), collections.Counter()
    cases["corrmom_random_nstep_d10"] = make_corr_random

    # ---- dual-averaging step-size adaptation -----------------------------------------
    def add_adapt_euclid(name, target, mk, metric, n, n_step, n_iters, seed0, qscale=1.0):
        q0 = qscale * case_rng(name).standard_normal((n, target.dim))

        def make():
            rsys = mici.systems.EuclideanMetricSystem(
                neg_log_dens=target.neg_log_dens, grad_neg_log_dens=target.grad,
                metric=None if mk == mdl.METRIC_IDENTITY else np.array(metric))
            rint = mici.integrators.LeapfrogIntegrator(rsys)
            ad = otr.euclid_adapter(orc.EuclidSystem(target, mk, metric))
            return adapt_case(name, "adapt_euclid", rsys, rint, ad, q0, n_step, n_iters, seed0,
                              model_keys(target, mk, metric))
        cases[name] = make

    add_adapt_euclid("adapt_euclid_dense_d16", mdl.GaussDense(mdl.make_spd(16, case_rng("adapt_euclid_dense_d16/P"))),
                     mdl.METRIC_IDENTITY, None, 5, 4, 40, 8000)
    add_adapt_euclid("adapt_euclid_quartic_d5_far_start", mdl.Poly(5, 0.0, 1.0), mdl.METRIC_DIAG,
                     np.exp(0.3 * case_rng("adapt_euclid_quartic_d5_far_start/M").standard_normal(5)), 5, 3, 40, 9000,
                     qscale=3.0)

    def add_adapt_riemann(name, target, rmetric, n, n_step, n_iters, seed0):
        q0 = case_rng(name).standard_normal((n, target.dim))

        def make():
            rsys = mici.systems.DenseRiemannianMetricSystem(
                neg_log_dens=target.neg_log_dens, grad_neg_log_dens=target.grad,
                metric_func=rmetric.metric_func, vjp_metric_func=rmetric.vjp_metric_func)
            rint = mici.integrators.ImplicitLeapfrogIntegrator(rsys)
            ad = otr.riemann_adapter(orc.RiemannianSystem(target, rmetric, None))
            return adapt_case(name, "adapt_riemann", rsys, rint, ad, q0, n_step, n_iters, seed0,
                              dict(target=target.tid, target_params=target.params(), rmetric=rmetric.mid,
                                   rmetric_params=rmetric.params()))
        cases[name] = make

    add_adapt_riemann("adapt_riemann_diagquad_poly_d5", mdl.Poly(5, 1.0, 1.0 / 3.0), mdl.DiagQuadMetric(5),
                      4, 3, 30, 10000)

    # ---- trace / statistics file format ---------------------------------------------
    def make_tracefmt():
        import tempfile
        from mici import samplers as msamp
        keys = ["pos", "hamiltonian", "weird key/with:chars*", "a.b-c_d"]
        names = {k: [p.name for p in msamp._generate_memmap_filenames("/tmp/x", "trace", k, range(3))] for k in keys}
        rsys = mici.systems.EuclideanMetricSystem(neg_log_dens=lambda q: 0.5 * q @ q, grad_neg_log_dens=lambda q: q)
        tr = mici.transitions.MetropolisStaticIntegrationTransition(rsys, mici.integrators.LeapfrogIntegrator(rsys, 0.1), 2)
        stat_names = sorted(tr.statistic_types)
        with tempfile.TemporaryDirectory() as tmp:
            stats = msamp._init_stats({"integration_transition": tr}, 2, 5, use_memmap=True, memmap_path=tmp)
            traces = msamp._init_traces([lambda s: {"pos": s.pos, "count": 3, "flag": np.array(True)}],
                                        [ChainState(pos=np.zeros(4), mom=np.zeros(4), dir=1)] * 2, 5,
                                        use_memmap=True, memmap_path=tmp)
            files = sorted(p.name for p in Path(tmp).iterdir())
            stat_dtypes = [str(stats["integration_transition"][k][0].dtype) for k in stat_names]
            stat_defaults = np.array([float(stats["integration_transition"][k][0][0]) for k in stat_names])
            trace_meta = {k: (str(v[0].dtype), list(v[0].shape), float(np.asarray(v[0]).ravel()[0])) for k, v in traces.items()}
        return dict(kind="tracefmt", keys=np.array(keys), status=np.zeros(1, dtype=np.int32), n_done=np.zeros(1, dtype=np.int32),
                    **{f"names_{i}": np.array(names[k]) for i, k in enumerate(keys)},
                    files=np.array(files), stat_names=np.array(stat_names), stat_dtypes=np.array(stat_dtypes),
                    stat_defaults=stat_defaults, trace_keys=np.array(sorted(trace_meta)),
                    trace_dtypes=np.array([trace_meta[k][0] for k in sorted(trace_meta)]),
                    trace_ndim=np.array([len(trace_meta[k][1]) for k in sorted(trace_meta)]),
                    trace_init=np.array([trace_meta[k][2] for k in sorted(trace_meta)])), collections.Counter()
    cases["tracefmt_reference"] = make_tracefmt

    # ---- ArviZ layout: the reference's own stat renaming + stacking (interop.py:31-51), which
    # is everything convert_to_inference_data / convert_to_data_tree do before handing the dictionaries to ArviZ
    def make_interop():
        from mici import interop as mint
        r = case_rng("interop_arviz_layout")
        n_chain, n_draw, d = 3, 7, 4
        traces = {"pos": [r.standard_normal((n_draw, d)) for _ in range(n_chain)],
                  "energy": [r.standard_normal(n_draw) for _ in range(n_chain)],
                  "lp": [r.standard_normal(n_draw) for _ in range(n_chain)]}
        stats = {"n_step": [r.integers(1, 9, n_draw) for _ in range(n_chain)],
                 "accept_stat": [r.uniform(size=n_draw) for _ in range(n_chain)],
                 "convergence_error": [r.uniform(size=n_draw) < 0.1 for _ in range(n_chain)],
                 "step_size": [np.full(n_draw, 0.25) for _ in range(n_chain)]}
        out = {}
        for tag, (ek, lk) in {"default": ("energy", "lp"), "nokeys": (None, None), "absent": ("h", "logp")}.items():
            ss = mint._stack_arrays(mint._preprocess_stats(traces, stats, ek, lk))
            out[f"{tag}_stat_keys"] = np.array(sorted(ss))
            for k, v in ss.items():
                out[f"{tag}_stat_{k}"] = v
        post = mint._stack_arrays(traces)
        return dict(kind="interop", status=np.zeros(1, dtype=np.int32), n_done=np.zeros(1, dtype=np.int32),
                    trace_keys=np.array(sorted(traces)), stat_in_keys=np.array(sorted(stats)),
                    **{f"in_trace_{k}": np.stack(v) for k, v in traces.items()},
                    **{f"in_stat_{k}": np.stack(v) for k, v in stats.items()},
                    **{f"post_{k}": v for k, v in post.items()}, **out), collections.Counter()
    cases["interop_arviz_layout"] = make_interop

    # ---- constrained systems beyond D = 16 / C = 3 (round 2: the capacity-64 kernels, up to 8 constraint
    #      functions; systems.py:876-1031, matrices.py:1270-1411).  Per-case random streams. ---------------------------
    def wide_linear(name, d, c, n, mk, h, cps, **kw):
        r = case_rng(name)
        a, b = r.standard_normal((c, d)), r.standard_normal(c)
        part = np.linalg.lstsq(a, b, rcond=None)[0]
        null = np.linalg.svd(a)[2][c:].T
        q0 = part + 0.5 * r.standard_normal((n, d - c)) @ null.T
        metric = (None if mk == mdl.METRIC_IDENTITY else np.exp(0.2 * r.standard_normal(d)) if mk == mdl.METRIC_DIAG
                  else mdl.make_spd(d, r))
        add_constrained(name, mdl.Poly(d, 1.0, 0.25), mdl.LinearConstr(a, b), mk, metric, q0, h, cps, r=r, **kw)

    # ---- a position-dependent metric that is not built into the device library (user source, hipRTC) ------------------
    for name, d, n, hh, cps, kw in (("riemann_user_softplus_poly_d6", 6, 5, 0.08, [1, 5, 20], {}),
                                    ("riemann_user_softplus_banana_d20", 20, 4, 0.02, [1, 5, 20], {}),
                                    ("riemann_user_softplus_poly_d32_steffensen", 32, 3, 0.05, [1, 5], dict(fp_solver=1)),
                                    ("riemann_user_softplus_poly_d6_fail_bigstep", 6, 5, 1.5, [1, 3], {}),
                                    # round 4: the BASELINE Riemannian sizes (the matrix-core kernels compiled around user
                                    # source), the team kernels' sizes in between, a failing case on the matrix cores
                                    ("riemann_user_softplus_banana_d64", 64, 4, 0.02, [1, 5, 20], {}),
                                    ("riemann_user_softplus_poly_d48_steffensen_l2", 48, 3, 0.04, [1, 5], dict(fp_solver=1, norm=1)),
                                    ("riemann_user_softplus_poly_d40_fail_bigstep", 40, 5, 0.9, [1, 3], {}),
                                    ("riemann_user_softplus_poly_d70", 70, 3, 0.03, [1, 4], {}),
                                    ("riemann_user_softplus_banana_d100", 100, 3, 0.015, [1, 4], {}),
                                    ("riemann_user_softplus_banana_d256", 256, 3, 0.01, [1, 3], {}),
                                    ("riemann_user_softplus_poly_d270", 270, 2, 0.01, [1, 2], {}),
                                    # round 5: beyond what a CU's registers hold - the global-memory tier compiled around
                                    # the user's source (plain form: the aux opt-in is 560 doubles at most, 2 D + 2 here)
                                    ("riemann_user_softplus_banana_d320", 320, 2, 0.01, [1, 2], {})):
        r = case_rng(name)
        tgt = mdl.Banana(d) if "banana" in name else mdl.Poly(d, 1.0, 1.0 / 3.0)
        add_riemann(name, tgt, mdl.SoftPlusRank1Metric(0.5 * r.standard_normal(d)), None, n, hh, cps, r=r, **kw)

    # ---- round 6: a user metric that DECLARES its constant + rank-one structure (csrc/user_metric.h MM_USER_LOWRANK; oracle
    #      SinRank1Metric: u nonlinear in q) - the device takes its Woodbury path, the reference its
    #      Cholesky factorisations: c3's kernel (D = 64), c4's (130, 256), a Steffensen / L2 case, a failing step size
    for name, d, n, hh, cps, kw in (("riemann_sinrank1_banana_d64", 64, 4, 0.02, [1, 5, 20], {}),
                                    ("riemann_sinrank1_poly_d130", 130, 3, 0.02, [1, 4, 10], {}),
                                    ("riemann_sinrank1_banana_d256", 256, 3, 0.01, [1, 3, 8], {}),
                                    ("riemann_sinrank1_poly_d48_steffensen_l2", 48, 3, 0.04, [1, 5], dict(fp_solver=1, norm=1)),
                                    ("riemann_sinrank1_poly_d64_fail_bigstep", 64, 5, 0.9, [1, 3], {})):
        r = case_rng(name)
        tgt = mdl.Banana(d) if "banana" in name else mdl.Poly(d, 1.0, 1.0 / 3.0)
        add_riemann(name, tgt, mdl.SinRank1Metric(mdl.make_spd(d, r)), None, n, hh, cps, r=r, **kw)

    # ---- constraints that are not built into the device library: they reach it as USER SOURCE compiled by hipRTC
    #      (tests/test_gpu_user_target.py); here the reference and the oracle run their NumPy twin ---------------------
    def add_user_constrained(name, d, n, mk, h, cps, **kw):
        r = case_rng(name)
        con = mdl.EllipsoidSaddleConstr(np.exp(0.3 * r.standard_normal(d)), 0.3)
        metric = None if mk == mdl.METRIC_IDENTITY else (np.exp(0.2 * r.standard_normal(d)) if mk == mdl.METRIC_DIAG
                                                         else mdl.make_spd(d, r))
        add_constrained(name, mdl.Poly(d, 0.5, 0.25), con, mk, metric, con.init(n, r), h, cps, r=r, **kw)

    add_user_constrained("constrained_user_ellipsoid_d5", 5, 6, mdl.METRIC_DENSE, 0.1, [1, 5, 20])
    add_user_constrained("constrained_user_ellipsoid_ambient_d6", 6, 5, mdl.METRIC_DIAG, 0.08, [1, 5, 20], variant="ambient")
    add_user_constrained("constrained_user_ellipsoid_d12_quasi", 12, 4, mdl.METRIC_IDENTITY, 0.1, [1, 5, 20], proj_solver=1)
    add_user_constrained("constrained_user_ellipsoid_d5_fail_bigstep", 5, 6, mdl.METRIC_IDENTITY, 1.5, [1, 3])
    # round 5: user constraints beyond D = 64 - the wave-per-chain kernels compiled around the user's source
    add_user_constrained("constrained_user_ellipsoid_dense_d100", 100, 3, mdl.METRIC_DENSE, 0.02, [1, 5, 20])
    add_user_constrained("constrained_user_ellipsoid_ambient_d128_quasi", 128, 3, mdl.METRIC_DIAG, 0.02, [1, 5],
                         variant="ambient", proj_solver=1)
    add_user_constrained("constrained_user_ellipsoid_gauss_d72", 72, 3, mdl.METRIC_DENSE, 0.03, [1, 5, 20], variant="gaussian")
    add_user_constrained("constrained_user_ellipsoid_d256_linesearch", 256, 2, mdl.METRIC_IDENTITY, 0.02, [1, 4], proj_solver=2)

    wide_linear("constrained_c4_linear_dense_d32", 32, 4, 4, mdl.METRIC_DENSE, 0.1, [1, 5, 20])
    wide_linear("constrained_c8_linear_diag_d64_quasi", 64, 8, 3, mdl.METRIC_DIAG, 0.1, [1, 5], proj_solver=1)
    wide_linear("constrained_c5_linear_ambient_d20", 20, 5, 4, mdl.METRIC_DENSE, 0.1, [1, 5, 20], variant="ambient")
    wide_linear("constrained_c4_linear_identity_d8", 8, 4, 5, mdl.METRIC_IDENTITY, 0.1, [1, 5, 20])
    wide_linear("constrained_c6_linear_gauss_d24", 24, 6, 4, mdl.METRIC_DENSE, 0.2, [1, 5, 20], variant="gaussian")

    # ---- SoftAbs systems beyond D = 64 (round 2: the NP = 128 instantiation of k_softabs.hip) ------------------------
    # round 4: SoftAbs on the banana - a tridiagonal Hessian that is NOT built into the device library: it reaches it as
    # user source (mici_amd/user_examples.py BANANA_HESS; hess_neg_log_dens / mtp_neg_log_dens, systems.py:1870-1920)
    for nm, d, n, hh, cps, kw in (("softabs_user_banana_d9", 9, 5, 0.05, [1, 5, 20], {}),
                                  ("softabs_user_banana_d40_steffensen", 40, 3, 0.03, [1, 5], dict(fp_solver=1)),
                                  ("softabs_user_banana_d64", 64, 4, 0.02, [1, 5, 20], {}),
                                  ("softabs_user_banana_d9_fail_bigstep", 9, 6, 0.7, [1, 3], dict(qscale=1.5)),
                                  # round 5: user Hessians on the workspace tiers (64 < D <= 256)
                                  ("softabs_user_banana_d100", 100, 2, 0.02, [1, 3], dict(qscale=0.7)),
                                  ("softabs_user_banana_d200", 200, 2, 0.02, [1, 3], dict(qscale=0.7))):
        add_riemann(nm, mdl.Banana(d), None, 1.0, n, hh, cps, r=case_rng(nm), **kw)
    add_midpoint_softabs("midpoint_softabs_user_banana_d16", mdl.Banana(16), 1.0, 3, 0.03, [1, 4],
                         r=case_rng("midpoint_softabs_user_banana_d16"))
    # round 5: the implicit midpoint rule on the SoftAbs workspace tiers (64 < D <= 256), built-in and user Hessian
    add_midpoint_softabs("midpoint_softabs_funnel_d100", mdl.Funnel(np.linspace(0.5, 2.0, 99)), 1.0, 2, 0.02, [1, 3],
                         r=case_rng("midpoint_softabs_funnel_d100"))
    add_midpoint_softabs("midpoint_softabs_user_banana_d160", mdl.Banana(160), 1.0, 2, 0.02, [1, 3], qscale=0.7,
                         r=case_rng("midpoint_softabs_user_banana_d160"))
    add_riemann("softabs_funnel_d100", mdl.Funnel(np.linspace(0.5, 2.0, 99)), None, 1.0, 3, 0.02, [1, 4],
                r=case_rng("softabs_funnel_d100"))
    add_riemann("softabs_poly_d72", mdl.Poly(72, 1.0, 1.0 / 3.0), None, 1.5, 3, 0.05, [1, 5],
                r=case_rng("softabs_poly_d72"))
    add_riemann("softabs_funnel_d128", mdl.Funnel(np.linspace(0.5, 2.0, 127)), None, 1.0, 2, 0.02, [1, 3],
                qscale=0.7, r=case_rng("softabs_funnel_d128"))
    # round 5: 128 < D <= 256 (the NP = 256 instantiation: Jacobi sweeps over columns streamed from memory)
    add_riemann("softabs_poly_d160", mdl.Poly(160, 1.0, 1.0 / 3.0), None, 1.5, 2, 0.05, [1, 3],
                r=case_rng("softabs_poly_d160"))
    add_riemann("softabs_funnel_d256", mdl.Funnel(np.linspace(0.5, 2.0, 255)), None, 1.0, 2, 0.02, [1, 3],
                qscale=0.7, r=case_rng("softabs_funnel_d256"))

    def wide_sphere_plane(name, d, n, mk, h, cps, **kw):
        r = case_rng(name)
        normal = r.standard_normal(d)
        x = r.standard_normal((n, d))
        x -= np.outer(x @ normal, normal) / (normal @ normal)
        q0 = x / np.linalg.norm(x, axis=1, keepdims=True)
        metric = np.exp(0.2 * r.standard_normal(d)) if mk == mdl.METRIC_DIAG else mdl.make_spd(d, r)
        add_constrained(name, mdl.Poly(d, 0.5, 0.25), mdl.SpherePlaneConstr(normal), mk, metric, q0, h, cps, r=r, **kw)

    wide_sphere_plane("constrained_c2_sphereplane_dense_d40", 40, 4, mdl.METRIC_DENSE, 0.05, [1, 5, 20])
    wide_sphere_plane("constrained_c2_sphereplane_diag_d33_linesearch", 33, 4, mdl.METRIC_DIAG, 0.05, [1, 5, 20],
                      proj_solver=2)

    # ---- round 5: dense Riemannian metrics beyond D = 279 - the global-memory tier (implicit_global.h) ------
    for nm, tgt, rm, n, hh, cps, kw in (
            ("riemann_global_rank1_banana_d300", mdl.Banana(300), "rank1", 3, 0.01, [1, 3], {}),
            ("riemann_global_rank1_banana_d512", mdl.Banana(512), "rank1", 2, 0.008, [1, 2], {}),
            ("riemann_global_diagquad_poly_d320_steffensen", mdl.Poly(320, 1.0, 1.0 / 3.0), "diag", 3, 0.05, [1, 3],
             dict(fp_solver=1)),
            ("riemann_global_rank1_poly_d281_fail_bigstep", mdl.Poly(281, 1.0, 1.0 / 3.0), "rank1", 4, 1.2, [1, 2],
             dict(qscale=2.0))):
        rr_ = case_rng(nm)
        metric_ = mdl.Rank1Metric(mdl.make_spd(tgt.dim, rr_)) if rm == "rank1" else mdl.DiagQuadMetric(tgt.dim)
        add_riemann(nm, tgt, metric_, None, n, hh, cps, r=rr_, **kw)

    # ---- round 5: the Gaussian split beyond D = 64 - k_constrained_wave.hip's GAUSS instantiations -----------
    wide_linear("constrained_c6_linear_gauss_dense_d128", 128, 6, 4, mdl.METRIC_DENSE, 0.2, [1, 5, 20], variant="gaussian")
    wide_linear("constrained_c3_linear_gauss_identity_d200_quasi", 200, 3, 3, mdl.METRIC_IDENTITY, 0.1, [1, 5],
                proj_solver=1, variant="gaussian")
    wide_linear("constrained_c8_linear_gauss_diag_d256_linesearch", 256, 8, 3, mdl.METRIC_DIAG, 0.1, [1, 4],
                proj_solver=2, variant="gaussian")
    wide_sphere_plane("constrained_c2_sphereplane_gauss_diag_d100", 100, 4, mdl.METRIC_DIAG, 0.05, [1, 5, 20],
                      variant="gaussian")
    wide_sphere_plane("constrained_c2_sphereplane_gauss_dense_d72_fail_bigstep", 72, 6, mdl.METRIC_DENSE, 1.3, [1, 3],
                      variant="gaussian")

    # ---- round 5: states scaled by 1e+-150 / 1e+-80.  The kernels replace IEEE division and square root
    # by lean Newton forms on their critical paths (mm_device.h rcp_nr / fdiv / sqrt_rsqrt) and test divergence as
    # `err > 1e10 or NaN` (solvers.py:80-84): at these scales the status a chain ends with - diverged, out of iterations,
    # LinAlgError inside or outside a solver, not reversible - must still be the reference's, whatever became inf or NaN on
    # the way.  Chain c of every case: positions x QS[c], momenta x PS[c].
    QS = np.array([1e150, 1e-150, 1e80, 1.0, 1e150, 1e-150, 1e-80, 1.0])
    PS = np.array([1.0, 1.0, 1e80, 1e150, 1e-150, 1e-150, 1e80, 1e-150])

    def extreme_states(name, d, n=8):
        r = case_rng(name)
        return r.standard_normal((n, d)) * QS[:n, None], r.standard_normal((n, d)) * PS[:n, None]

    def add_extreme_riemann(name, target, rmetric, coeff, h, cps, **kw):
        q0, p0 = extreme_states(name, target.dim)
        cases[name] = lambda: riemann_case(name, target, rmetric, coeff, q0, p0, dirs_for(len(q0)), h, cps, **kw)

    with np.errstate(all="ignore"):
        add_extreme_riemann("extreme_riemann_rank1_poly_d8", mdl.Poly(8, 1.0, 1.0 / 3.0), mdl.Rank1Metric(B8), None, 0.1, [1, 3])
        add_extreme_riemann("extreme_riemann_diagquad_poly_d5_steffensen", mdl.Poly(5, 1.0, 1.0 / 3.0), mdl.DiagQuadMetric(5),
                            None, 0.1, [1, 3], fp_solver=1)
        add_extreme_riemann("extreme_riemann_rank1_banana_d40", mdl.Banana(40),
                            mdl.Rank1Metric(mdl.make_spd(40, case_rng("extreme_base40"))), None, 0.02, [1, 3])
        add_extreme_riemann("extreme_riemann_rank1_banana_d100", mdl.Banana(100),
                            mdl.Rank1Metric(mdl.make_spd(100, case_rng("extreme_base100"))), None, 0.02, [1, 2])
        add_extreme_riemann("extreme_softabs_poly_d9", mdl.Poly(9, 1.0, 1.0 / 3.0), None, 1.0, 0.05, [1, 3])

    def add_extreme_euclid(name, target, mk, metric, h, cps):
        q0, p0 = extreme_states(name, target.dim)
        cases[name] = lambda: euclid_case(name, target, mk, metric, q0, p0, dirs_for(len(q0)), h, cps)

    Px = mdl.make_spd(16, case_rng("extreme_prec16"))
    add_extreme_euclid("extreme_euclid_dense_d16", mdl.GaussDense(Px), mdl.METRIC_DENSE,
                       mdl.make_spd(16, case_rng("extreme_metric16")), 0.1, [1, 5, 20])

    def add_extreme_constrained(name, target, constraint, mk, metric, q0, h, cps, **kw):
        # positions stay ON the manifold; the momenta are scaled and projected onto the cotangent space
        r = case_rng(name)
        n, d = q0.shape
        osys = orc.ConstrainedSystem(target, constraint, mk, metric)
        scale = np.array([1e150, 1e-150, 1e80, 1.0, 1e10, 1e-80, 1e40, 1e-300])[:n, None]
        p0 = project_momentum(osys, q0, np.stack([osys.msqrt(zz) for zz in r.standard_normal((n, d))])) * scale
        cases[name] = lambda: constrained_case(name, target, constraint, mk, metric, q0, p0, dirs_for(n), h, cps, **kw)

    add_extreme_constrained("extreme_constrained_torus", mdl.Torus(), mdl.TorusConstr(), mdl.METRIC_IDENTITY, None,
                            mdl.torus_init(8, case_rng("extreme_torus_q")), 0.1, [1, 3])
    add_extreme_constrained("extreme_constrained_torus_quasi", mdl.Torus(), mdl.TorusConstr(), mdl.METRIC_IDENTITY, None,
                            mdl.torus_init(8, case_rng("extreme_torus_q2")), 0.1, [1, 3], proj_solver=1)
    rr = case_rng("extreme_sphereplane")
    nrm = rr.standard_normal(40)
    xs = rr.standard_normal((8, 40))
    xs -= np.outer(xs @ nrm, nrm) / (nrm @ nrm)
    add_extreme_constrained("extreme_constrained_sphereplane_d40", mdl.Poly(40, 0.5, 0.25), mdl.SpherePlaneConstr(nrm),
                            mdl.METRIC_DIAG, np.exp(0.2 * rr.standard_normal(40)),
                            xs / np.linalg.norm(xs, axis=1, keepdims=True), 0.05, [1, 3])

    all_counts = {}
    n_ok, bad = 0, []
    for name, fn in cases.items():
        if args.only and args.only not in name:
            continue
        data, counts = fn()
        path = os.path.join(args.out, name + ".npz")
        np.savez_compressed(path, **{f"count_{k}": v for k, v in counts.items()}, **data)
        all_counts[name] = dict(counts)
        print(f"{name}: ok  status={data['status'].tolist()} n_done={data['n_done'].tolist()} "
              f"counts={dict(counts)}")
        if args.check:
            why = compare_fixture(path, os.path.join(committed, name + ".npz"))
            if why:
                bad.append((name, why))
                print(f"   CHECK FAILED {name}: {why}")
            else:
                n_ok += 1
    if args.check:
        stale = sorted(set(p.stem for p in Path(committed).glob("*.npz")) - set(cases)) if not args.only else []
        print(f"check: {n_ok}/{n_ok + len(bad)} regenerated cases equal the committed fixtures"
              + (f"; committed fixtures without a generator case: {stale}" if stale else ""))
        if bad or stale:
            raise SystemExit(1)


def compare_fixture(new_path, old_path, rtol=1e-11):
    """Empty string if the regenerated fixture equals the committed one, else the first difference."""
    if not os.path.exists(old_path):
        return "no committed fixture"
    a, b = np.load(new_path, allow_pickle=False), np.load(old_path, allow_pickle=False)
    if sorted(a.files) != sorted(b.files):
        return f"keys differ: {sorted(set(a.files) ^ set(b.files))}"
    for k in a.files:
        x, y = a[k], b[k]
        if x.shape != y.shape or x.dtype != y.dtype:
            return f"{k}: shape/dtype {x.shape}/{x.dtype} vs {y.shape}/{y.dtype}"
        if x.dtype.kind in "fc":
            if not np.array_equal(np.isnan(x), np.isnan(y)):
                return f"{k}: NaN pattern"
            xx, yy = np.nan_to_num(x), np.nan_to_num(y)
            err = np.max(np.abs(xx - yy) / np.maximum(1.0, np.abs(yy))) if x.size else 0.0
            if err > rtol:
                return f"{k}: max scaled difference {err:.3g}"
        elif not np.array_equal(x, y):
            return f"{k}: values differ"
    return ""


if __name__ == "__main__":
    main()
