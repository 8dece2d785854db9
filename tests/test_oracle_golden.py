"""CPU suite: the NumPy oracle replayed against the committed reference fixtures.

The fixtures in tests/golden/ are outputs of the imported reference (tools/gen_golden.py);
this is what pins the oracle wherever /root/reference is absent."""

import numpy as np
import pytest

from conftest import assert_close, golden_names, load_golden
from oracle import integrators as orc
from oracle import models as mdl


@pytest.mark.parametrize("name", golden_names("euclid") + golden_names("extreme_euclid"))
def test_euclid_leapfrog(name):
    g = load_golden(name)
    n, d = g["q0"].shape
    target = mdl.target_from_id(g["target"], g["target_params"], d)
    mk = int(g["metric_kind"])
    metric = None if mk == mdl.METRIC_IDENTITY else g["metric"]
    system = orc.EuclidSystem(target, mk, metric)
    h = float(g["step_size"])
    for k, s in enumerate(g["checkpoints"]):
        for c in range(n):
            q, p = orc.leapfrog_steps(system, g["q0"][c], g["p0"][c], g["dir"][c] * h, int(s))
            assert_close(q, g["q_out"][k, c], 1e-13 * max(1, s), f"{name} q@{s}")
            assert_close(p, g["p_out"][k, c], 1e-13 * max(1, s), f"{name} p@{s}")
            assert_close(system.h(q, p), g["h_out"][k, c], 1e-12, f"{name} h@{s}")
        qb, pb = orc.leapfrog_steps_batch(system, g["q0"], g["p0"], g["dir"] * h, int(s))
        assert_close(qb, g["q_out"][k], 1e-12 * max(1, s), f"{name} batch q@{s}")
        assert_close(pb, g["p_out"][k], 1e-12 * max(1, s), f"{name} batch p@{s}")


@pytest.mark.parametrize("name", golden_names("symcomp"))
def test_symmetric_composition(name):
    """SymmetricCompositionIntegrator / BCSS integrators (integrators.py:176-378)."""
    g = load_golden(name)
    n, d = g["q0"].shape
    target = mdl.target_from_id(g["target"], g["target_params"], d)
    mk = int(g["metric_kind"])
    system = orc.EuclidSystem(target, mk, None if mk == mdl.METRIC_IDENTITY else g["metric"])
    h, free, h1 = float(g["step_size"]), list(g["free_coefficients"]), bool(g["initial_h1_flow_step"])
    coefficients = orc.composition_coefficients(free)
    assert len(coefficients) == 2 * len(free) + 3
    assert abs(sum(coefficients[0::2]) - 1) < 1e-15 and abs(sum(coefficients[1::2]) - 1) < 1e-15
    assert coefficients == coefficients[::-1]
    for k, s in enumerate(g["checkpoints"]):
        for c in range(n):
            q, p = orc.composition_steps(system, g["q0"][c], g["p0"][c], g["dir"][c] * h, int(s), free, h1)
            assert_close(q, g["q_out"][k, c], 1e-13 * max(1, s), f"{name} q@{s}")
            assert_close(p, g["p_out"][k, c], 1e-13 * max(1, s), f"{name} p@{s}")
            assert_close(system.h(q, p), g["h_out"][k, c], 1e-12, f"{name} h@{s}")
    if not free:  # no free coefficients = the leapfrog integrator
        q1, p1 = orc.leapfrog_steps(system, g["q0"][0], g["p0"][0], g["dir"][0] * h, 7)
        q2, p2 = orc.composition_steps(system, g["q0"][0], g["p0"][0], g["dir"][0] * h, 7, ())
        assert np.array_equal(q1, q2) and np.array_equal(p1, p2)


@pytest.mark.parametrize("name", golden_names("transition"))
def test_metropolis_static_transitions(name):
    """oracle/transitions.py replays the reference's recorded draws (transitions.py:129-142, 275-352)."""
    from oracle import transitions as otr
    g = load_golden(name)
    n_tr, n, d = g["z"].shape
    kind = str(g["kind"])
    target = mdl.target_from_id(g["target"], g["target_params"], d)
    if kind == "transition_euclid":
        mk = int(g["metric_kind"])
        system = orc.EuclidSystem(target, mk, None if mk == mdl.METRIC_IDENTITY else g["metric"])
        ad = otr.euclid_adapter(system, list(g["free_coefficients"]) if int(g["composition"]) else None)
    elif kind == "transition_riemann":
        ad = otr.riemann_adapter(orc.RiemannianSystem(
            target, mdl.rmetric_from_id(g["rmetric"], g["rmetric_params"], d), None))
    else:
        ad = otr.constrained_adapter(orc.ConstrainedSystem(target, mdl.constr_from_id(g["constr"], g["constr_params"], g["q0"].shape[1])))
    h, n_step = float(g["step_size"]), int(g["n_step"])
    for c in range(n):
        q, direction = g["q0"][c].copy(), 1
        for t in range(n_tr):
            p = ad.sample_momentum(q, g["z"][t, c])
            drew = []

            def draw(t=t, c=c, drew=drew):
                drew.append(1)
                assert not np.isnan(g["u"][t, c])
                return g["u"][t, c]

            q, p, direction, st = otr.metropolis_static_transition(ad, q, p, direction, h, n_step, draw)
            assert len(drew) == (0 if np.isnan(g["u"][t, c]) else 1)
            assert_close(q, g["q_out"][t, c], 1e-11, f"{name} q t{t} c{c}")
            assert_close(p, g["p_out"][t, c], 1e-11, f"{name} p t{t} c{c}")
            assert direction == g["dir_out"][t, c]
            for k in ("n_step", "accept_stat", "metrop_accept_prob", "convergence_error", "non_reversible_step"):
                assert_close(float(st[k]), g[f"stat_{k}"][t, c], 1e-10, f"{name} {k}")


@pytest.mark.parametrize("name", golden_names("gausseuclid"))
def test_gaussian_euclidean_metric_system(name):
    """GaussianEuclideanMetricSystem (systems.py:369-474): h2 = q.q/2 + p.M^-1 p/2 with the exact
    rotation as h2_flow, under the leapfrog, a BCSS composition and the implicit midpoint rule."""
    g = load_golden(name)
    n, d = g["q0"].shape
    target = mdl.target_from_id(g["target"], g["target_params"], d)
    mk = int(g["metric_kind"])
    system = orc.GaussianEuclidSystem(target, mk, None if mk == mdl.METRIC_IDENTITY else g["metric"])
    h, kind, free = float(g["step_size"]), int(g["integrator"]), list(g["free_coefficients"])
    for k, s in enumerate(int(s) for s in g["checkpoints"]):
        for c in range(n):
            dt = g["dir"][c] * h
            if kind == 0:
                q, p = orc.leapfrog_steps(system, g["q0"][c], g["p0"][c], dt, s)
            elif kind == 1:
                q, p = orc.composition_steps(system, g["q0"][c], g["p0"][c], dt, s, free)
            else:
                q, p, st, nd = orc.implicit_midpoint_steps(system, g["q0"][c], g["p0"][c], dt, s)
                assert st == 0 and nd == s
            tol = 1e-12 * max(1, s) if kind < 2 else 1e-9
            assert_close(q, g["q_out"][k, c], tol, f"{name} q@{s}")
            assert_close(p, g["p_out"][k, c], tol, f"{name} p@{s}")
            assert_close(system.h(q, p), g["h_out"][k, c], 1e-11, f"{name} h@{s}")
    # the rotation is exact: composing +dt and -dt returns to the start, and h2 is conserved by it
    q, p = g["q0"][0], g["p0"][0]
    q1, p1 = system.h2_flow(q, p, 0.7)
    q2, p2 = system.h2_flow(q1, p1, -0.7)
    assert_close(q2, q, 1e-13, "h2_flow inverse q")
    assert_close(p2, p, 1e-13, "h2_flow inverse p")
    assert_close(0.5 * q1 @ q1 + 0.5 * p1 @ system.minv(p1), 0.5 * q @ q + 0.5 * p @ system.minv(p), 1e-13, "h2")


@pytest.mark.parametrize("name", golden_names("impliciteuclid"))
def test_implicit_leapfrog_on_a_euclidean_system(name):
    """The reference runs ImplicitLeapfrogIntegrator on plain Euclidean systems too
    (tests/test_integrators.py:435-462): dh2_dpos = 0, so the implicit maps are explicit."""
    g = load_golden(name)
    n, d = g["q0"].shape
    target = mdl.target_from_id(g["target"], g["target_params"], d)
    mk = int(g["metric_kind"])
    esys = orc.EuclidSystem(target, mk, None if mk == mdl.METRIC_IDENTITY else g["metric"])
    system = orc.EuclidAsGeneralSystem(esys)
    h = float(g["step_size"])
    for k, s in enumerate(int(s) for s in g["checkpoints"]):
        for c in range(n):
            q, p, st, nd = orc.implicit_leapfrog_steps(system, g["q0"][c], g["p0"][c], g["dir"][c] * h, s)
            assert st == 0 and nd == s
            assert_close(q, g["q_out"][k, c], 1e-12 * max(1, s), f"{name} q@{s}")
            assert_close(p, g["p_out"][k, c], 1e-12 * max(1, s), f"{name} p@{s}")
    coefficients = [1.0, 1.0, 0.0, 1.0, 1.0]
    qb, pb = orc.leapfrog_steps_batch(esys, g["q0"], g["p0"], g["dir"] * h, int(g["checkpoints"][-1]),
                                      coefficients=coefficients)
    assert_close(qb, g["q_out"][-1], 1e-11, f"{name} as the (1, 1, 0, 1, 1) composition")


@pytest.mark.parametrize("name", golden_names("midpoint"))
def test_implicit_midpoint(name):
    """ImplicitMidpointIntegrator (integrators.py:547-681) on Euclidean and dense-Riemannian systems."""
    g = load_golden(name)
    n, d = g["q0"].shape
    target = mdl.target_from_id(g["target"], g["target_params"], d)
    if str(g["system"]) == "euclid":
        mk = int(g["metric_kind"])
        system = orc.EuclidSystem(target, mk, None if mk == mdl.METRIC_IDENTITY else g["metric"])
    elif str(g["system"]) == "softabs":
        system = orc.RiemannianSystem(target, None, float(g["rmetric_params"][0]))
    else:
        system = orc.RiemannianSystem(target, mdl.rmetric_from_id(g["rmetric"], g["rmetric_params"], d), None)
    norm = orc.NORMS[int(g["norm"])]
    kw = dict(fp_solver=orc.FP_SOLVERS[int(g["fp_solver"])], rev_norm=norm,
              fp_kwargs=dict(norm=norm, convergence_tol=float(g["fp_conv_tol"]),
                             divergence_tol=float(g["fp_div_tol"]), max_iters=int(g["fp_max_iters"])))
    h = float(g["step_size"])
    for k, s in enumerate(int(s) for s in g["checkpoints"]):
        for c in range(n):
            q, p, st, nd = orc.implicit_midpoint_steps(system, g["q0"][c], g["p0"][c], g["dir"][c] * h, s, **kw)
            assert nd == min(s, g["n_done"][c])
            assert st == (0 if g["n_done"][c] >= s else g["status"][c])
            assert_close(q, g["q_out"][k, c], 1e-9, f"{name} q@{s}")
            assert_close(p, g["p_out"][k, c], 1e-9, f"{name} p@{s}")


@pytest.mark.parametrize("name", golden_names("adapt_"))
def test_dual_averaging_adaptation(name):
    """oracle/adapters.py against the reference's DualAveragingStepSizeAdapter run (adapters.py:174-389)."""
    from oracle import adapters as oad
    from oracle import transitions as otr
    g = load_golden(name)
    n_iters, n, d = g["z"].shape
    target = mdl.target_from_id(g["target"], g["target_params"], d)
    if str(g["kind"]) == "adapt_euclid":
        mk = int(g["metric_kind"])
        ad = otr.euclid_adapter(orc.EuclidSystem(target, mk, None if mk == mdl.METRIC_IDENTITY else g["metric"]))
    else:
        ad = otr.riemann_adapter(orc.RiemannianSystem(
            target, mdl.rmetric_from_id(g["rmetric"], g["rmetric_params"], d), None))
    states = []
    for c in range(n):
        q, direction = g["q0"][c].copy(), 1
        eps = oad.find_init_step_size(ad, q, ad.sample_momentum(q, g["z_init"][c]), 1)
        assert eps == g["init_step_size"][c]
        st = oad.initial_state(eps)
        for t in range(n_iters):
            p = ad.sample_momentum(q, g["z"][t, c])
            q, p, direction, stats = otr.metropolis_static_transition(
                ad, q, p, direction, eps, int(g["n_step"]), lambda t=t, c=c: g["u"][t, c])
            assert_close(stats["accept_stat"], g["accept_stat"][t, c], 1e-9, f"{name} accept t{t}")
            eps = oad.update(st, stats["accept_stat"])
            assert_close(eps, g["step_sizes"][t, c], 1e-9, f"{name} step size t{t} c{c}")
        assert_close(q, g["q_final"][c], 1e-8, f"{name} q final")
        states.append(st)
    assert_close(oad.finalize(states), float(g["final_step_size"]), 1e-9, "final step size")
    assert_close(oad.finalize(states[0]), float(np.exp(g["smoothed_log_step_size"][0])), 1e-9, "single chain")


def test_correlated_momentum_and_random_trajectory_length():
    """CorrelatedMomentumTransition + MetropolisRandomIntegrationTransition (transitions.py:143-198, 355-402)."""
    from oracle import transitions as otr
    g = load_golden("corrmom_random_nstep_d10")
    n_tr, n, d = g["z"].shape
    ad = otr.euclid_adapter(orc.EuclidSystem(mdl.target_from_id(g["target"], g["target_params"], d),
                                             int(g["metric_kind"]), g["metric"]))
    for c in range(n):
        q, p, direction = g["q0"][c].copy(), None, 1
        for t in range(n_tr):
            p = otr.correlated_momentum(ad, q, p, g["z"][t, c], float(g["coeff"]))
            q, p, direction, st = otr.metropolis_static_transition(
                ad, q, p, direction, float(g["step_size"]), int(g["n_steps"][t, c]), lambda t=t, c=c: g["u"][t, c])
            assert_close(q, g["q_out"][t, c], 1e-11, f"q t{t} c{c}")
            assert_close(p, g["p_out"][t, c], 1e-11, f"p t{t} c{c}")
            assert direction == g["dir_out"][t, c]
            assert_close(st["accept_stat"], g["accept_stat"][t, c], 1e-10, "accept_stat")


def _riemann_system(g, counters=None):
    n, d = g["q0"].shape
    target = mdl.target_from_id(g["target"], g["target_params"], d)
    mid = int(g["rmetric"])
    if mid == mdl.RMETRIC_SOFTABS:
        return orc.RiemannianSystem(target, None, float(g["rmetric_params"][0]), counters)
    return orc.RiemannianSystem(target, mdl.rmetric_from_id(mid, g["rmetric_params"], d), None,
                                counters)


def _riemann_kwargs(g):
    norm = orc.NORMS[int(g["norm"])]
    return dict(
        fp_solver=orc.FP_SOLVERS[int(g["fp_solver"])], rev_norm=norm,
        fp_kwargs=dict(norm=norm, convergence_tol=float(g["fp_conv_tol"]),
                       divergence_tol=float(g["fp_div_tol"]), max_iters=int(g["fp_max_iters"])),
    )


RIEMANN = [n for n in golden_names() if n.startswith(("riemann", "softabs", "extreme_riemann", "extreme_softabs"))]


@pytest.mark.parametrize("name", RIEMANN)
def test_implicit_leapfrog(name):
    g = load_golden(name)
    if g["q0"].shape[1] > 64:
        cps = [(0, int(g["checkpoints"][0]))]  # keep the CPU suite fast: first checkpoint only
    else:
        cps = list(enumerate(int(s) for s in g["checkpoints"]))
    counters = orc.Counters()
    system = _riemann_system(g, counters)
    kw = _riemann_kwargs(g)
    h = float(g["step_size"])
    tol = 1e-9 if name.startswith("softabs") else 2e-11
    n = g["q0"].shape[0]
    s_max = int(g["checkpoints"].max())
    with np.errstate(all="ignore"):
        for c in range(n):
            for k, s in cps:
                q, p, st, nd = orc.implicit_leapfrog_steps(
                    system, g["q0"][c], g["p0"][c], g["dir"][c] * h, s, **kw)
                assert_close(q, g["q_out"][k, c], tol, f"{name} q@{s} chain {c}")
                assert_close(p, g["p_out"][k, c], tol, f"{name} p@{s} chain {c}")
                if s == s_max:
                    assert st == g["status"][c]
                    assert nd == g["n_done"][c]
    if len(cps) == len(g["checkpoints"]):
        counters.clear()
        with np.errstate(all="ignore"):
            for c in range(n):
                orc.implicit_leapfrog_steps(system, g["q0"][c], g["p0"][c], g["dir"][c] * h,
                                            s_max, **kw)
        assert counters.get("fp_iters", 0) == int(g["count_fp_iters"])


@pytest.mark.parametrize("name", golden_names("constrained") + golden_names("extreme_constrained"))
def test_constrained_leapfrog(name):
    g = load_golden(name)
    n, d = g["q0"].shape
    target = mdl.target_from_id(g["target"], g["target_params"], d)
    constraint = mdl.constr_from_id(g["constr"], g["constr_params"], g["q0"].shape[1])
    mk = int(g["metric_kind"])
    metric = None if mk == mdl.METRIC_IDENTITY else g["metric"]
    variant = str(g.get("variant", "hausdorff"))  # "ambient": dens_wrt_hausdorff=False; "gaussian": Gaussian split
    if variant == "gaussian":
        system = orc.GaussianConstrainedSystem(target, constraint, mk, metric)
    else:
        system = orc.ConstrainedSystem(target, constraint, mk, metric, dens_wrt_hausdorff=(variant == "hausdorff"))
    h = float(g["step_size"])
    s_max = int(g["checkpoints"].max())
    for c in range(n):
        for k, s in enumerate(int(s) for s in g["checkpoints"]):
            q, p, st, nd = orc.constrained_leapfrog_steps(
                system, g["q0"][c], g["p0"][c], g["dir"][c] * h, s, n_inner_step=int(g["n_inner"]),
                proj_solver=int(g.get("proj_solver", 0)))
            assert_close(q, g["q_out"][k, c], 1e-10, f"{name} q@{s} chain {c}")
            assert_close(p, g["p_out"][k, c], 1e-10, f"{name} p@{s} chain {c}")
            if nd == s:
                assert_close(system.h(q, p), g["h_out"][k, c], 1e-10, f"{name} h@{s} chain {c}")
            if s == s_max:
                assert st == g["status"][c]
                assert nd == g["n_done"][c]


# ---- the reference's own solver known-answer tests (tests/test_solvers.py:36, 65-121) ----------------
@pytest.mark.parametrize("solver", [orc.solve_fixed_point_direct, orc.solve_fixed_point_steffensen])
@pytest.mark.parametrize("norm", [orc.maximum_norm, orc.euclidean_norm])
@pytest.mark.parametrize("tol", [1e-6, 1e-8, 1e-10])
def test_fixed_point_known_answers(solver, norm, tol):
    x = solver(np.cos, np.array([1.0]), convergence_tol=tol, norm=norm)
    assert abs(x[0] - 0.7390851332151607) < 10 * tol  # tests/test_solvers.py:36
    x = solver(lambda y: 0.5 * (y + 2.0 / y), np.array([1.0]), convergence_tol=tol, norm=norm)
    assert abs(x[0] - 2.0**0.5) < 10 * tol  # Babylonian square root


@pytest.mark.parametrize("solver", [orc.solve_fixed_point_direct, orc.solve_fixed_point_steffensen])
def test_fixed_point_failures(solver):
    # the reference's divergent problems (tests/test_solvers.py:47-55, 83-92): only the class of
    # the failure is pinned there; the direct solver must report divergence
    # (with float input Steffensen's extrapolation lands exactly on the fixed point 0 of the
    # doubling map - checked against the imported reference - so only direct sees it diverge)
    funcs = [lambda x: 1 + x**2]
    if solver is orc.solve_fixed_point_direct:
        funcs.append(lambda x: 2 * x)
    for func in funcs:
        with np.errstate(all="ignore"), pytest.raises(orc.ConvergenceError) as e:
            solver(func, np.arange(3, dtype=np.float64), max_iters=10000)
        assert e.value.status == orc.ST_DIVERGED
    with pytest.raises(orc.ConvergenceError) as e:
        solver(np.cos, np.array([1.0]), max_iters=1)
    assert e.value.status == orc.ST_MAX_ITERS

    def bad(x):
        raise ValueError("boom")

    with pytest.raises(orc.ConvergenceError) as e:
        solver(bad, np.array([1.0]))
    assert e.value.status == orc.ST_SOLVER_LINALG

    def bad2(x):
        raise orc.LinAlgError("boom")

    with pytest.raises(orc.ConvergenceError) as e:
        solver(bad2, np.array([1.0]))
    assert e.value.status == orc.ST_SOLVER_LINALG
