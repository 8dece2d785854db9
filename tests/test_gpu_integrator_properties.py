"""GPU property tests mirroring the reference's own integrator suite (tests/test_integrators.py).

Same grid -- integrator x system x metric x size in {1, 2, 5} x 5 initial states -- same step sizes and the
same per-class tolerances (``h_diff_tol``), run through the HIP kernels:
  * reversibility after n in {1, 5, 20} steps and a direction flip            (reference :75-91)
  * approximate conservation of the Hamiltonian over 200 steps                (:93-108)
  * the input state is not modified                                           (:110-124)
  * phase-space volume preservation for linear systems                        (:128-142)
  * position / momentum constraint residuals below 1e-8                       (:159-197)
The five chains of a test are integrated as one batch; a chain whose step fails (status != 0) is treated as the
reference treats an IntegratorError (the trajectory is abandoned)."""

import numpy as np
import pytest

from mici_amd import integrators, models, solvers, systems
from mici_amd.states import ChainState

pytestmark = pytest.mark.gpu

SEED = 3046987125
N_STEPS = (1, 5, 20)
N_STEPS_HAMILTONIAN = 200
N_STATE = 5
SIZES = (1, 2, 5)
METRICS = ("identity", "diagonal", "dense")


def make_metric(kind, size, rng):
    eigval = np.exp(0.1 * rng.standard_normal(size))
    eigvec = np.linalg.qr(rng.standard_normal((size, size)))[0]
    return {"identity": None, "diagonal": eigval, "dense": (eigvec * eigval) @ eigvec.T}[kind]


def metric_matmul(metric, x):
    if metric is None:
        return x
    return metric * x if metric.ndim == 1 else x @ metric.T


# ---- systems of the reference suite (:236-273, 492-500, 519-615) as device models --------------------------
def linear_euclid(size, metric):          # neg_log_dens = sum(q^2) / 2
    return systems.EuclideanMetricSystem(models.GaussIso(size), metric=metric)


def nonlinear_euclid(size, metric):       # sum(q^4) / 4
    return systems.EuclideanMetricSystem(models.Poly(size, 0.0, 1.0), metric=metric)


def linear_gaussian(size, metric):        # 0 (the Gaussian part lives in h2)
    return systems.GaussianEuclideanMetricSystem(models.Poly(size, 0.0, 0.0), metric=metric)


def nonlinear_gaussian(size, metric):     # sum(q^4) / 8
    return systems.GaussianEuclideanMetricSystem(models.Poly(size, 0.0, 0.5), metric=metric)


def diagonal_riemannian(size, metric):    # sum(q^2)/2 + sum(q^4)/12, metric diag(1 + q^2)
    return systems.DenseRiemannianMetricSystem(models.Poly(size, 1.0, 1.0 / 3.0), models.DiagQuadMetric(size))


def constrained_linear(size, metric):     # sum(q^2)/2 on {q_0 = 0}
    return systems.DenseConstrainedEuclideanMetricSystem(models.GaussIso(size), models.FirstCoordConstr(),
                                                         metric=metric)


def constrained_nonlinear(size, metric):  # sum(q^4)/8 on {q_0^2 + q_1^2 = 1}
    return systems.DenseConstrainedEuclideanMetricSystem(models.Poly(size, 0.0, 0.5), models.CircleConstr(),
                                                         metric=metric)


def gaussian_constrained_linear(size, metric):
    return systems.GaussianDenseConstrainedEuclideanMetricSystem(models.Poly(size, 0.0, 0.0),
                                                                 models.FirstCoordConstr(), metric=metric)


def gaussian_constrained_nonlinear(size, metric):
    return systems.GaussianDenseConstrainedEuclideanMetricSystem(models.Poly(size, 0.0, 0.5),
                                                                 models.CircleConstr(), metric=metric)


LEAPFROG = integrators.LeapfrogIntegrator
BCSS2, BCSS3, BCSS4 = (integrators.BCSSTwoStageIntegrator, integrators.BCSSThreeStageIntegrator,
                       integrators.BCSSFourStageIntegrator)
IMPLICIT, MIDPOINT = integrators.ImplicitLeapfrogIntegrator, integrators.ImplicitMidpointIntegrator
NEWTON, QUASI, LINE = (solvers.solve_projection_onto_manifold_newton,
                       solvers.solve_projection_onto_manifold_quasi_newton,
                       solvers.solve_projection_onto_manifold_newton_with_line_search)


def constrained(solver):
    return lambda system, step_size: integrators.ConstrainedLeapfrogIntegrator(system, step_size,
                                                                               projection_solver=solver)


# (id, integrator factory, system factory, step_size, h_diff_tol, linear, init kind) -- reference :282-311, 400-615
CASES = [
    ("leapfrog-linear-euclid", LEAPFROG, linear_euclid, 0.25, 2e-3, True, "free"),
    ("leapfrog-nonlinear-euclid", LEAPFROG, nonlinear_euclid, 0.05, 1e-3, False, "free"),
    ("leapfrog-linear-gaussian", LEAPFROG, linear_gaussian, 0.5, 1e-10, True, "free"),
    ("leapfrog-nonlinear-gaussian", LEAPFROG, nonlinear_gaussian, 0.1, 2e-3, False, "free"),
    ("bcss2-linear-euclid", BCSS2, linear_euclid, 0.25, 2e-4, True, "free"),
    ("bcss2-nonlinear-euclid", BCSS2, nonlinear_euclid, 0.05, 1e-3, False, "free"),
    ("bcss2-linear-gaussian", BCSS2, linear_gaussian, 0.5, 1e-10, True, "free"),
    ("bcss2-nonlinear-gaussian", BCSS2, nonlinear_gaussian, 0.1, 2e-3, False, "free"),
    ("bcss3-linear-euclid", BCSS3, linear_euclid, 0.25, 5e-5, True, "free"),
    ("bcss3-nonlinear-euclid", BCSS3, nonlinear_euclid, 0.25, 5e-4, False, "free"),
    ("bcss3-linear-gaussian", BCSS3, linear_gaussian, 0.5, 1e-10, True, "free"),
    ("bcss3-nonlinear-gaussian", BCSS3, nonlinear_gaussian, 0.5, 5e-4, False, "free"),
    ("bcss4-linear-euclid", BCSS4, linear_euclid, 1.0, 2e-5, True, "free"),
    ("bcss4-nonlinear-euclid", BCSS4, nonlinear_euclid, 0.25, 1e-3, False, "free"),
    ("bcss4-linear-gaussian", BCSS4, linear_gaussian, 1.0, 1e-10, True, "free"),
    ("bcss4-nonlinear-gaussian", BCSS4, nonlinear_gaussian, 0.5, 5e-4, False, "free"),
    ("implicit-leapfrog-linear-euclid", IMPLICIT, linear_euclid, 0.25, 5e-3, True, "free"),
    ("implicit-midpoint-linear-euclid", MIDPOINT, linear_euclid, 0.25, 1e-7, True, "free"),
    ("implicit-leapfrog-nonlinear-euclid", IMPLICIT, nonlinear_euclid, 0.1, 6e-3, False, "free"),
    ("implicit-midpoint-nonlinear-euclid", MIDPOINT, nonlinear_euclid, 0.1, 5e-3, False, "free"),
    ("implicit-leapfrog-diagonal-riemannian", IMPLICIT, diagonal_riemannian, 0.1, 1e-3, False, "free"),
    ("implicit-midpoint-diagonal-riemannian", MIDPOINT, diagonal_riemannian, 0.1, 2e-4, False, "free"),
    ("constrained-linear", constrained(NEWTON), constrained_linear, 0.1, 1e-2, True, "linear"),
    ("constrained-nonlinear-quasi", constrained(QUASI), constrained_nonlinear, 0.1, 1e-2, False, "circle"),
    ("constrained-nonlinear-newton", constrained(NEWTON), constrained_nonlinear, 0.1, 1e-2, False, "circle"),
    ("constrained-nonlinear-linesearch", constrained(LINE), constrained_nonlinear, 0.1, 1e-2, False, "circle"),
    ("constrained-gaussian-linear", constrained(NEWTON), gaussian_constrained_linear, 0.5, 1e-4, True, "linear"),
    ("constrained-gaussian-nonlinear-quasi", constrained(QUASI), gaussian_constrained_nonlinear, 0.05, 5e-2, False,
     "circle"),
    ("constrained-gaussian-nonlinear-newton", constrained(NEWTON), gaussian_constrained_nonlinear, 0.05, 5e-2,
     False, "circle"),
]


def grid():
    for case in CASES:
        name, _, sysf, _, _, _, init = case
        for size in SIZES:
            if init != "free" and size == 1:
                continue  # the reference's size_more_than_one fixture
            kinds = ("identity",) if sysf is diagonal_riemannian else METRICS
            for mk in kinds:
                yield pytest.param(case, size, mk, id=f"{name}-d{size}-{mk}")


def setup(case, size, metric_kind):
    """System, integrator and the reference's N_STATE initial states as one batch."""
    _, intf, sysf, step_size, h_diff_tol, linear, init = case
    rng = np.random.default_rng(SEED)
    metric = make_metric(metric_kind, size, rng)
    system = sysf(size, metric)
    integ = intf(system, step_size)
    if init == "free":
        qp = rng.standard_normal((N_STATE, 2, size))
        q0, p0 = qp[:, 0].copy(), qp[:, 1].copy()
    elif init == "linear":
        qp = rng.standard_normal((N_STATE, 2, size - 1))
        q0 = np.concatenate([np.zeros((N_STATE, 1)), qp[:, 0]], 1)
        p0 = metric_matmul(metric, np.concatenate([np.zeros((N_STATE, 1)), qp[:, 1]], 1))
    else:
        theta = rng.uniform(size=N_STATE) * 2 * np.pi
        q0 = np.concatenate([np.cos(theta)[:, None], np.sin(theta)[:, None],
                             rng.standard_normal((N_STATE, size - 2))], 1)
        p0 = system.sample_momentum_batch(q0, rng.standard_normal((N_STATE, size)))
    return system, integ, q0, p0, h_diff_tol, linear


def integrate_with_reversal(integ, q, p, dirs, n_step):
    """reference :60-68 for a batch: failed chains keep their initial state and direction."""
    qn, pn, status, _ = integ.step_batch(q, p, dirs, n_steps=n_step)
    ok = status == 0
    q_out, p_out, d_out = np.where(ok[:, None], qn, q), np.where(ok[:, None], pn, p), np.where(ok, -dirs, dirs)
    return q_out, p_out, d_out.astype(np.int8)


@pytest.mark.parametrize("case,size,metric_kind", list(grid()))
def test_reversibility(case, size, metric_kind):
    system, integ, q0, p0, _, _ = setup(case, size, metric_kind)
    dirs = np.ones(N_STATE, dtype=np.int8)
    for n_step in N_STEPS:
        q, p, d = integrate_with_reversal(integ, q0, p0, dirs, n_step)
        q, p, d = integrate_with_reversal(integ, q, p, d, n_step)
        assert np.allclose(q, q0), f"positions do not return on reversal after {n_step} steps"
        assert np.allclose(p, p0), f"momenta do not return on reversal after {n_step} steps"
        assert np.array_equal(d, dirs)


@pytest.mark.parametrize("case,size,metric_kind", list(grid()))
def test_approx_hamiltonian_conservation(case, size, metric_kind):
    system, integ, q0, p0, h_diff_tol, _ = setup(case, size, metric_kind)
    h_vals = [system.h_batch(q0, p0)]
    q, p = q0, p0
    alive = np.ones(N_STATE, dtype=bool)
    for _ in range(N_STEPS_HAMILTONIAN):
        q, p, status, _ = integ.step_batch(q, p, 1, n_steps=1)
        alive &= status == 0  # reference :101-102: an IntegratorError ends the test for that state
        h_vals.append(system.h_batch(q, p))
    h_vals = np.array(h_vals)  # [201, N_STATE]
    n = N_STEPS_HAMILTONIAN
    diff_h = h_vals[: n // 2].mean(0) - h_vals[n // 2:].mean(0)
    assert np.all(np.abs(diff_h[alive]) < h_diff_tol), (diff_h, h_diff_tol)


@pytest.mark.parametrize("case,size,metric_kind", list(grid()))
def test_state_mutation(case, size, metric_kind):
    system, integ, q0, p0, _, _ = setup(case, size, metric_kind)
    init_state = ChainState(pos=q0[0].copy(), mom=p0[0].copy(), dir=1)
    try:
        state = integ.step(init_state)
    except Exception as e:  # an IntegratorError is a legitimate outcome; anything else is not
        from mici_amd.errors import IntegratorError
        assert isinstance(e, IntegratorError)
        return
    assert init_state is not state
    assert np.all(init_state.pos == q0[0]) and np.all(init_state.mom == p0[0]) and init_state.dir == 1


@pytest.mark.parametrize("case,size,metric_kind", [g for g in grid() if g.values[0][5]])
def test_volume_preservation(case, size, metric_kind):
    system, integ, q0, p0, _, _ = setup(case, size, metric_kind)
    for n_step in N_STEPS:
        q, p, status, _ = integ.step_batch(q0, p0, 1, n_steps=n_step)
        assert np.all(status == 0)
        init_zs = np.concatenate([q0, p0], 1).T
        final_zs = np.concatenate([q, p], 1).T
        assert np.allclose(np.linalg.det(init_zs @ init_zs.T), np.linalg.det(final_zs @ final_zs.T))


@pytest.mark.parametrize("case,size,metric_kind", [g for g in grid() if g.values[0][6] != "free"])
def test_position_and_momentum_constraints(case, size, metric_kind):
    system, integ, q0, p0, _, _ = setup(case, size, metric_kind)
    init = case[6]

    def residuals(q, p):
        if init == "linear":
            c, jac = q[:, :1], np.tile(np.eye(1, size, 0), (len(q), 1, 1))
        else:
            c = q[:, 0:1] ** 2 + q[:, 1:2] ** 2 - 1.0
            jac = np.zeros((len(q), 1, size))
            jac[:, 0, 0], jac[:, 0, 1] = 2 * q[:, 0], 2 * q[:, 1]
        v = system.dh_dmom_batch(q, p)
        return np.max(np.abs(c)), np.max(np.abs(np.einsum("ncd,nd->nc", jac, v)))

    c0, m0 = residuals(q0, p0)
    assert c0 < 1e-8 and m0 < 1e-8
    dirs = np.ones(N_STATE, dtype=np.int8)
    for n_step in N_STEPS:
        q, p, _ = integrate_with_reversal(integ, q0, p0, dirs, n_step)
        c1, m1 = residuals(q, p)
        assert c1 < 1e-8, f"position constraint violated after {n_step} steps: {c1:.1e}"
        assert m1 < 1e-8, f"momentum constraint violated after {n_step} steps: {m1:.1e}"
