"""GPU parity: constrained leapfrog (Newton projection) vs reference fixtures and the oracle.
Tolerance: solves stop at 1e-9 / 1e-8, so with equal iteration counts results agree to 1e-10;
status and completed-step counts must match exactly."""

import numpy as np
import pytest

from conftest import assert_close, golden_names, load_golden
from oracle import integrators as orc
from oracle import models as omdl

from mici_amd import integrators, models, solvers, systems
from mici_amd.errors import ConvergenceError, NonReversibleStepError
from mici_amd.states import ChainState

pytestmark = pytest.mark.gpu


def build(g):
    n, d = g["q0"].shape
    target = models.target_from_id(g["target"], g["target_params"], d)
    if int(g["constr"]) == models.CONSTR_USER:  # not built in: reaches the library as HIP source (hipRTC)
        from user_sources import ELLIPSOID_SADDLE
        constr = models.UserConstraint(2, ELLIPSOID_SADDLE, g["constr_params"])
    else:
        constr = models.constr_from_id(g["constr"], g["constr_params"], g["q0"].shape[1])
    mk = int(g["metric_kind"])
    metric = None if mk == models.METRIC_IDENTITY else g["metric"]
    variant = str(g.get("variant", "hausdorff"))
    if variant == "gaussian":  # GaussianDenseConstrainedEuclideanMetricSystem, systems.py:1034-1184
        system = systems.GaussianDenseConstrainedEuclideanMetricSystem(target, constr, metric=metric)
    else:
        system = systems.DenseConstrainedEuclideanMetricSystem(target, constr, metric=metric,
                                                               dens_wrt_hausdorff=(variant == "hausdorff"))
    proj = {0: solvers.solve_projection_onto_manifold_newton,
            1: solvers.solve_projection_onto_manifold_quasi_newton,
            2: solvers.solve_projection_onto_manifold_newton_with_line_search}[int(g.get("proj_solver", 0))]
    integ = integrators.ConstrainedLeapfrogIntegrator(system, float(g["step_size"]),
                                                      n_inner_step=int(g["n_inner"]),
                                                      projection_solver=proj)
    return system, integ


@pytest.mark.parametrize("name", golden_names("constrained"))
def test_constrained_leapfrog_matches_reference_fixture(name):
    g = load_golden(name)
    system, integ = build(g)
    s_max = int(g["checkpoints"].max())
    for k, s in enumerate(int(s) for s in g["checkpoints"]):
        q, p, status, n_done = integ.step_batch(g["q0"], g["p0"], g["dir"], n_steps=s)
        # 1e-10 while roundoff-level differences have had <= 20 steps to grow; 5e-9 beyond that
        # (<= 5e-9 when the nonlinear dynamics has amplified solver-level noise)
        tol = 1e-10 if s <= 20 else 5e-9
        assert_close(q, g["q_out"][k], tol, f"{name} q@{s}")
        assert_close(p, g["p_out"][k], tol, f"{name} p@{s}")
        if s == s_max:
            assert np.array_equal(status, g["status"]), (status, g["status"])
            assert np.array_equal(n_done, g["n_done"]), (n_done, g["n_done"])
        h = system.h_batch(q, p)
        assert_close(h, g["h_out"][k], 10 * tol, f"{name} h@{s}")


def test_torus_full_size_properties():
    """BASELINE config c5 per-GPU shard (2048 chains), h = 0.1: constraint and cotangent residuals
    (tests/test_integrators.py:159-197), reversibility, oracle on a sample."""
    rng = np.random.default_rng(1234)
    n, h, steps = 2048, 0.1, 50
    target, constr = omdl.Torus(), omdl.TorusConstr()
    osys = orc.ConstrainedSystem(target, constr)
    system = systems.DenseConstrainedEuclideanMetricSystem(models.Torus(), models.TorusConstr())
    integ = integrators.ConstrainedLeapfrogIntegrator(system, h)
    q0 = omdl.torus_init(n, rng)
    p0 = system.sample_momentum_batch(q0, rng.standard_normal((n, 3)))
    jac0 = np.stack([constr.jacob_constr(x)[0] for x in q0])
    assert np.max(np.abs(np.sum(jac0 * p0, 1))) < 1e-12  # momentum starts in the cotangent space
    q, p, status, n_done = integ.step_batch(q0, p0, 1, n_steps=steps)
    ok = status == 0
    assert ok.mean() > 0.95
    c = np.array([constr.constr(x)[0] for x in q[ok]])
    assert np.max(np.abs(c)) < 1e-8
    jac = np.stack([constr.jacob_constr(x)[0] for x in q[ok]])
    assert np.max(np.abs(np.sum(jac * p[ok], 1))) < 1e-8
    for cidx in np.concatenate([np.arange(6), rng.integers(0, n, 6)]):
        qo, po, so, no = orc.constrained_leapfrog_steps(osys, q0[cidx], p0[cidx], h, steps)
        assert so == status[cidx] and no == n_done[cidx]
        assert_close(q[cidx], qo, 1e-9, f"q chain {cidx}")
        assert_close(p[cidx], po, 1e-9, f"p chain {cidx}")
    # reversibility as the reference tests it (tests/test_integrators.py:75-91: n_step <= 20, allclose)
    q5, p5, s5, _ = integ.step_batch(q0, p0, 1, n_steps=5)
    ok5 = s5 == 0
    qb, pb, sb, _ = integ.step_batch(q5[ok5], p5[ok5], -1, n_steps=5)
    back = sb == 0
    assert back.mean() > 0.95
    assert np.allclose(qb[back], q0[ok5][back], rtol=1e-5, atol=1e-6)
    assert np.allclose(pb[back], p0[ok5][back], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("variant", ["ambient", "gaussian"])
@pytest.mark.parametrize("metric_kind", ["identity", "diag", "dense"])
def test_gram_term_and_gaussian_split_match_oracle(variant, metric_kind):
    """dens_wrt_hausdorff=False and the Gaussian split at a c5-sized shard: oracle on a sample, manifold and
    cotangent residuals, and h (with its log-det-sqrt-Gram term) on every chain."""
    rng = np.random.default_rng(99)
    n, h, steps = 1024, 0.08, 12
    target, constr = omdl.Torus(), omdl.TorusConstr()
    if metric_kind == "identity":
        mk, metric = omdl.METRIC_IDENTITY, None
    elif metric_kind == "diag":
        mk, metric = omdl.METRIC_DIAG, np.exp(0.3 * rng.standard_normal(3))
    else:
        mk, metric = omdl.METRIC_DENSE, omdl.make_spd(3, rng)
    if variant == "gaussian":
        osys = orc.GaussianConstrainedSystem(target, constr, mk, metric)
        system = systems.GaussianDenseConstrainedEuclideanMetricSystem(models.Torus(), models.TorusConstr(),
                                                                       metric=metric)
    else:
        osys = orc.ConstrainedSystem(target, constr, mk, metric, dens_wrt_hausdorff=False)
        system = systems.DenseConstrainedEuclideanMetricSystem(models.Torus(), models.TorusConstr(),
                                                               metric=metric, dens_wrt_hausdorff=False)
    integ = integrators.ConstrainedLeapfrogIntegrator(system, h)
    q0 = omdl.torus_init(n, rng)
    p0 = system.sample_momentum_batch(q0, rng.standard_normal((n, 3)))
    dirs = np.where(rng.random(n) < 0.5, 1, -1).astype(np.int8)
    q, p, status, n_done = integ.step_batch(q0, p0, dirs, n_steps=steps)
    ok = status == 0
    assert ok.mean() > 0.9
    c = np.array([constr.constr(x)[0] for x in q[ok]])
    assert np.max(np.abs(c)) < 1e-8
    jac = np.stack([constr.jacob_constr(x)[0] for x in q[ok]])
    minv_p = np.stack([osys.minv(x) for x in p[ok]])
    assert np.max(np.abs(np.sum(jac * minv_p, 1))) < 1e-8
    for cidx in np.concatenate([np.arange(5), rng.integers(0, n, 5)]):
        qo, po, so, no = orc.constrained_leapfrog_steps(osys, q0[cidx], p0[cidx], dirs[cidx] * h, steps)
        assert so == status[cidx] and no == n_done[cidx]
        assert_close(q[cidx], qo, 1e-9, f"q chain {cidx}")
        assert_close(p[cidx], po, 1e-9, f"p chain {cidx}")
    ho = np.array([osys.h(q[i], p[i]) for i in range(64)])
    assert_close(system.h_batch(q[:64], p[:64]), ho, 1e-12, "h with the Gram term")


@pytest.mark.parametrize("variant", ["hausdorff", "ambient", "gaussian"])
@pytest.mark.parametrize("solver", [0, 1, 2])
def test_two_and_three_constraints_match_oracle(variant, solver):
    """C = 2 (sphere and plane, D = 6) and C = 3 (linear, D = 7): C x C Cholesky / symmetric inverse of the Gram
    matrix and pivoted LU of the residual Jacobian per chain, against the oracle on a sample of a 512-chain batch."""
    if variant == "gaussian" and solver == 2:
        pytest.skip("the reference tests the Gaussian split with the Newton and quasi-Newton solvers only")
    rng = np.random.default_rng(5 + solver)
    proj = [solvers.solve_projection_onto_manifold_newton, solvers.solve_projection_onto_manifold_quasi_newton,
            solvers.solve_projection_onto_manifold_newton_with_line_search][solver]
    n, steps = 512, 10
    for dim, make in ((6, "sphere"), (7, "linear")):
        metric = omdl.make_spd(dim, rng)
        if make == "sphere":
            normal = rng.standard_normal(dim)
            oc, pc = omdl.SpherePlaneConstr(normal), models.SpherePlaneConstr(normal)
            q0 = rng.standard_normal((n, dim))
            q0 -= np.outer(q0 @ normal, normal) / (normal @ normal)
            q0 /= np.linalg.norm(q0, axis=1, keepdims=True)
            h = 0.05
        else:
            a, b = rng.standard_normal((3, dim)), rng.standard_normal(3)
            oc, pc = omdl.LinearConstr(a, b), models.LinearConstr(a, b)
            null = np.linalg.svd(a)[2][3:].T
            q0 = np.linalg.lstsq(a, b, rcond=None)[0] + rng.standard_normal((n, dim - 3)) @ null.T
            h = 0.1
        ot, pt = omdl.Poly(dim, 0.5, 0.25), models.Poly(dim, 0.5, 0.25)
        if variant == "gaussian":
            osys = orc.GaussianConstrainedSystem(ot, oc, omdl.METRIC_DENSE, metric)
            system = systems.GaussianDenseConstrainedEuclideanMetricSystem(pt, pc, metric=metric)
        else:
            osys = orc.ConstrainedSystem(ot, oc, omdl.METRIC_DENSE, metric, dens_wrt_hausdorff=(variant == "hausdorff"))
            system = systems.DenseConstrainedEuclideanMetricSystem(pt, pc, metric=metric,
                                                                   dens_wrt_hausdorff=(variant == "hausdorff"))
        integ = integrators.ConstrainedLeapfrogIntegrator(system, h, projection_solver=proj)
        p0 = system.sample_momentum_batch(q0, rng.standard_normal((n, dim)))
        jac0 = np.stack([oc.jacob_constr(x) for x in q0])
        minv_p0 = np.stack([osys.minv(x) for x in p0])
        assert np.max(np.abs(np.einsum("ncd,nd->nc", jac0, minv_p0))) < 1e-11  # cotangent space
        dirs = np.where(rng.random(n) < 0.5, 1, -1).astype(np.int8)
        q, p, status, n_done = integ.step_batch(q0, p0, dirs, n_steps=steps)
        ok = status == 0
        assert ok.mean() > 0.9
        assert np.max(np.abs(np.stack([oc.constr(x) for x in q[ok]]))) < 1e-8
        for cidx in np.concatenate([np.arange(4), rng.integers(0, n, 4)]):
            qo, po, so, no = orc.constrained_leapfrog_steps(osys, q0[cidx], p0[cidx], dirs[cidx] * h, steps,
                                                            proj_solver=solver)
            assert so == status[cidx] and no == n_done[cidx]
            assert_close(q[cidx], qo, 1e-9, f"{make} q chain {cidx}")
            assert_close(p[cidx], po, 1e-9, f"{make} p chain {cidx}")
        ho = np.array([osys.h(q[i], p[i]) for i in range(32)])
        assert_close(system.h_batch(q[:32], p[:32]), ho, 1e-12, f"{make} h")


def test_single_state_step_raises_reference_exceptions():
    g = load_golden("constrained_torus_fail_bigstep")
    system, integ = build(g)
    expect = {2: ConvergenceError, 1: ConvergenceError, 3: ConvergenceError, 4: NonReversibleStepError}
    for c in range(g["q0"].shape[0]):
        state = ChainState(pos=g["q0"][c].copy(), mom=g["p0"][c].copy(), dir=int(g["dir"][c]))
        n_ok = 0
        try:
            for _ in range(int(g["checkpoints"].max())):
                state = integ.step(state)
                n_ok += 1
            assert g["status"][c] == 0
        except (ConvergenceError, NonReversibleStepError) as e:
            assert isinstance(e, expect[int(g["status"][c])])
        assert n_ok == g["n_done"][c]


_WAVE_SCRIPT = r"""
import json, sys
import numpy as np
sys.path.insert(0, {root!r})
from mici_amd import integrators, models, solvers, systems
from oracle import models as omdl
out = {{}}
cases = [("sphereplane_d40_newton", 40, 0, "dense"), ("sphereplane_d33_linesearch", 33, 2, "diag"),
         ("linear_c8_d64_quasi", 64, 1, "dense"), ("sphere_d12_newton_inner2", 12, 0, "identity"),
         ("sphereplane_d24_newton_ambient", 24, 0, "dense"), ("sphere_d48_linesearch_ambient", 48, 2, "diag"),
         ("linear_c8_d20_newton_ambient", 20, 0, "dense")]
for name, d, solver, mk in cases:
    rng = np.random.default_rng(d)
    metric = None if mk == "identity" else (np.exp(0.2 * rng.standard_normal(d)) if mk == "diag" else omdl.make_spd(d, rng))
    if name.startswith("linear"):
        a, b = rng.standard_normal((8, d)), rng.standard_normal(8)
        con = models.LinearConstr(a, b)
        x0 = np.linalg.lstsq(a, b, rcond=None)[0]
        q0 = x0 + 0.5 * rng.standard_normal((24, d)) @ (np.eye(d) - a.T @ np.linalg.solve(a @ a.T, a)).T
    elif name.startswith("sphereplane"):
        nrm = rng.standard_normal(d)
        con = models.SpherePlaneConstr(nrm)
        x = rng.standard_normal((24, d))
        x -= np.outer(x @ nrm, nrm) / (nrm @ nrm)
        q0 = x / np.linalg.norm(x, axis=1, keepdims=True)
    else:
        con = models.SphereConstr()
        x = rng.standard_normal((24, d))
        q0 = x / np.linalg.norm(x, axis=1, keepdims=True)
    system = systems.DenseConstrainedEuclideanMetricSystem(models.Poly(d, 0.5, 0.25), con, metric=metric,
                                                           dens_wrt_hausdorff="ambient" not in name)
    proj = [solvers.solve_projection_onto_manifold_newton, solvers.solve_projection_onto_manifold_quasi_newton,
            solvers.solve_projection_onto_manifold_newton_with_line_search][solver]
    integ = integrators.ConstrainedLeapfrogIntegrator(system, 0.04, projection_solver=proj,
                                                      n_inner_step=2 if "inner2" in name else 1)
    p0 = system.sample_momentum_batch(q0, rng.standard_normal(q0.shape))
    q, p, st, nd = integ.step_batch(q0, p0, np.where(np.arange(24) % 3 == 0, -1, 1), n_steps=15)
    out[name] = dict(q=q.tolist(), p=p.tolist(), status=st.tolist(), n_done=nd.tolist(),
                     newton=int(integ.last_counters["n_newton_iters"]))
print(json.dumps(out))
"""


def test_wave_per_chain_kernel_equals_lane_per_chain_core():
    """k_constrained_wave.hip (8 < D <= 64: one wave per chain, Jacobians and vectors in registers, sums through LDS)
    against the lane-per-chain core (MICI_AMD_CONSTRAINED_KERNEL=lane: the padded instantiations with their arrays in
    scratch) on the same inputs: all three projection solvers, two constraints and eight, identity / diagonal / dense
    metric, n_inner_step = 2, both density conventions (dens_wrt_hausdorff=False adds the Gram log-determinant's
    gradient), both time directions - same statuses, step counts and Newton iteration counts, states to rounding (the
    D-long sums are ordered differently)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    for mode in ("wave", "lane"):
        env = dict(os.environ)
        env.pop("MICI_AMD_CONSTRAINED_KERNEL", None)
        if mode == "lane":
            env["MICI_AMD_CONSTRAINED_KERNEL"] = "lane"
        r = subprocess.run([sys.executable, "-c", _WAVE_SCRIPT.format(root=root)], capture_output=True, text=True, env=env,
                           cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        res[mode] = json.loads(r.stdout.strip().splitlines()[-1])
    differs = False
    for key in res["wave"]:
        a, b = res["wave"][key], res["lane"][key]
        assert a["status"] == b["status"] and a["n_done"] == b["n_done"] and a["newton"] == b["newton"], key
        assert all(s == 0 for s in a["status"]), key
        assert_close(np.array(a["q"]), np.array(b["q"]), 1e-11, f"{key} positions")
        assert_close(np.array(a["p"]), np.array(b["p"]), 1e-11, f"{key} momenta")
        differs = differs or a["q"] != b["q"]
    assert differs  # two different kernels really ran (their sums are ordered differently)


@pytest.mark.parametrize("variant", ["hausdorff", "ambient"])
def test_wave_per_chain_kernel_matches_oracle(variant):
    """The wave-per-chain kernel against the oracle directly (D = 20, sphere and plane, dense metric), both density
    conventions: statuses, step counts, states."""
    rng = np.random.default_rng(41)
    n, d, h, steps = 64, 20, 0.05, 8
    metric = omdl.make_spd(d, rng)
    normal = rng.standard_normal(d)
    q0 = rng.standard_normal((n, d))
    q0 -= np.outer(q0 @ normal, normal) / (normal @ normal)
    q0 /= np.linalg.norm(q0, axis=1, keepdims=True)
    hausdorff = variant == "hausdorff"
    osys = orc.ConstrainedSystem(omdl.Poly(d, 0.5, 0.25), omdl.SpherePlaneConstr(normal), omdl.METRIC_DENSE, metric,
                                 dens_wrt_hausdorff=hausdorff)
    system = systems.DenseConstrainedEuclideanMetricSystem(models.Poly(d, 0.5, 0.25), models.SpherePlaneConstr(normal),
                                                           metric=metric, dens_wrt_hausdorff=hausdorff)
    integ = integrators.ConstrainedLeapfrogIntegrator(system, h)
    p0 = system.sample_momentum_batch(q0, rng.standard_normal((n, d)))
    dirs = np.where(rng.random(n) < 0.5, 1, -1).astype(np.int8)
    q, p, status, n_done = integ.step_batch(q0, p0, dirs, n_steps=steps)
    assert np.all(status == 0)
    for cidx in range(0, n, 8):
        qo, po, so, no = orc.constrained_leapfrog_steps(osys, q0[cidx], p0[cidx], dirs[cidx] * h, steps)
        assert so == status[cidx] and no == n_done[cidx]
        assert_close(q[cidx], qo, 1e-10, f"q chain {cidx}")
        assert_close(p[cidx], po, 1e-10, f"p chain {cidx}")


@pytest.mark.parametrize("case", ["sphereplane_d100_dense", "linear_c8_d200_quasi", "sphere_d700_ambient_diag",
                                  "linear_c2_d1024_linesearch", "circle_d130_identity_inner2"])
def test_wave_per_chain_kernel_beyond_64_dimensions_matches_oracle(case):
    """D > 64: the wave-per-chain kernels with four (D <= 256, C <= 8) or sixteen (D <= 1024, C <= 2) coordinates per
    lane - all three projection solvers, the three metric kinds, both density conventions (h with its Gram
    log-determinant included), n_inner_step = 2, both time directions, the momentum projection of sample_momentum."""
    rng = np.random.default_rng(len(case))
    n, steps, solver, n_inner, hausdorff = 12, 6, 0, 1, True
    if case == "sphereplane_d100_dense":
        d, h = 100, 0.03
        metric, mk = omdl.make_spd(d, rng), omdl.METRIC_DENSE
        normal = rng.standard_normal(d)
        oc, pc = omdl.SpherePlaneConstr(normal), models.SpherePlaneConstr(normal)
        q0 = rng.standard_normal((n, d))
        q0 -= np.outer(q0 @ normal, normal) / (normal @ normal)
        q0 /= np.linalg.norm(q0, axis=1, keepdims=True)
    elif case == "linear_c8_d200_quasi":
        d, h, solver = 200, 0.05, 1
        metric, mk = omdl.make_spd(d, rng), omdl.METRIC_DENSE
        a, b = rng.standard_normal((8, d)), rng.standard_normal(8)
        oc, pc = omdl.LinearConstr(a, b), models.LinearConstr(a, b)
        null = np.linalg.svd(a)[2][8:].T
        q0 = np.linalg.lstsq(a, b, rcond=None)[0] + 0.5 * rng.standard_normal((n, d - 8)) @ null.T
    elif case == "sphere_d700_ambient_diag":
        d, h, hausdorff = 700, 0.02, False
        metric, mk = np.exp(0.2 * rng.standard_normal(d)), omdl.METRIC_DIAG
        oc, pc = omdl.SphereConstr(), models.SphereConstr()
        q0 = rng.standard_normal((n, d))
        q0 /= np.linalg.norm(q0, axis=1, keepdims=True)
    elif case == "linear_c2_d1024_linesearch":
        d, h, solver = 1024, 0.05, 2
        metric, mk = None, omdl.METRIC_IDENTITY
        a, b = rng.standard_normal((2, d)), rng.standard_normal(2)
        oc, pc = omdl.LinearConstr(a, b), models.LinearConstr(a, b)
        null = np.linalg.svd(a)[2][2:].T
        q0 = np.linalg.lstsq(a, b, rcond=None)[0] + 0.3 * rng.standard_normal((n, d - 2)) @ null.T
    else:
        d, h, n_inner = 130, 0.05, 2
        metric, mk = None, omdl.METRIC_IDENTITY
        oc, pc = omdl.CircleConstr(), models.CircleConstr()
        q0 = rng.standard_normal((n, d))
        q0[:, :2] /= np.linalg.norm(q0[:, :2], axis=1, keepdims=True)
    proj = [solvers.solve_projection_onto_manifold_newton, solvers.solve_projection_onto_manifold_quasi_newton,
            solvers.solve_projection_onto_manifold_newton_with_line_search][solver]
    osys = orc.ConstrainedSystem(omdl.Poly(d, 0.5, 0.25), oc, mk, metric, dens_wrt_hausdorff=hausdorff)
    system = systems.DenseConstrainedEuclideanMetricSystem(models.Poly(d, 0.5, 0.25), pc, metric=metric,
                                                           dens_wrt_hausdorff=hausdorff)
    integ = integrators.ConstrainedLeapfrogIntegrator(system, h, projection_solver=proj, n_inner_step=n_inner)
    p0 = system.sample_momentum_batch(q0, rng.standard_normal((n, d)))
    jac0 = np.stack([np.atleast_2d(oc.jacob_constr(x)) for x in q0])
    minv_p0 = np.stack([osys.minv(x) for x in p0])
    assert np.max(np.abs(np.einsum("ncd,nd->nc", jac0, minv_p0))) < 1e-10  # in the cotangent space
    dirs = np.where(np.arange(n) % 3 == 0, -1, 1).astype(np.int8)
    q, p, status, n_done = integ.step_batch(q0, p0, dirs, n_steps=steps)
    assert np.all(status == 0), status
    assert np.max(np.abs(np.stack([np.atleast_1d(oc.constr(x)) for x in q]))) < 1e-8
    for cidx in range(0, n, 4):
        qo, po, so, no = orc.constrained_leapfrog_steps(osys, q0[cidx], p0[cidx], dirs[cidx] * h, steps,
                                                        proj_solver=solver, n_inner_step=n_inner)
        assert so == status[cidx] and no == n_done[cidx]
        assert_close(q[cidx], qo, 1e-9, f"q chain {cidx}")
        assert_close(p[cidx], po, 1e-9, f"p chain {cidx}")
    ho = np.array([osys.h(q[i], p[i]) for i in range(4)])
    assert_close(system.h_batch(q[:4], p[:4]), ho, 1e-11, "h")
