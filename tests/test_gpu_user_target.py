"""GPU: user-defined targets compiled at run time (include/mici_amd.h, mm_model_create_from_source; csrc/mm_rtc.hip).
The reference takes Python callables for neg_log_dens / grad_neg_log_dens (systems.py:107, 119); here the user
brings HIP device code.  Checked: a built-in target re-expressed as user source reproduces the built-in kernels; a
target that is NOT built in matches the oracle driven by its NumPy twin; compile errors and unsupported system
classes fail loudly."""

import numpy as np
import pytest

from conftest import assert_close
from oracle import integrators as orc
from oracle import models as omdl

from mici_amd import integrators, models, systems, transitions
from mici_amd.errors import DeviceError
from mici_amd.runtime import DeviceBatch, default_context

pytestmark = pytest.mark.gpu

BANANA_SRC = """
__device__ double mm_user_grad(const double* q, int i, int dim, const double* params) {
  double g = -(1.0 - q[i]) / 10.0;
  if (i > 0) g += 2.0 * (q[i] - q[i - 1] * q[i - 1]);
  if (i < dim - 1) g -= 4.0 * q[i] * (q[i + 1] - q[i] * q[i]);
  return g;
}
__device__ double mm_user_nld_term(const double* q, int i, int dim, const double* params) {
  double v = (1.0 - q[i]) * (1.0 - q[i]) / 20.0;
  if (i < dim - 1) {
    const double r = q[i + 1] - q[i] * q[i];
    v += r * r;
  }
  return v;
}
"""

# l(q) = sum_i [ log(1 + exp(-a_i q_i)) + b q_i^2 / 2 ] + c (sum_i q_i)^2 / 2 : not separable, not built in
LOGISTIC_SRC = """
__device__ double mm_user_grad(const double* q, int i, int dim, const double* params) {
  const double a = params[i], b = params[dim], c = params[dim + 1];
  double s = 0.0;
  for (int k = 0; k < dim; ++k) s += q[k];
  return -a / (1.0 + exp(a * q[i])) + b * q[i] + c * s;
}
__device__ double mm_user_nld_term(const double* q, int i, int dim, const double* params) {
  const double a = params[i], b = params[dim], c = params[dim + 1];
  double s = 0.0;
  for (int k = 0; k < dim; ++k) s += q[k];
  return log1p(exp(-a * q[i])) + 0.5 * b * q[i] * q[i] + (i == 0 ? 0.5 * c * s * s : 0.0);
}
"""


class LogisticTwin(omdl.Target):
    """NumPy twin of LOGISTIC_SRC for the oracle."""

    def __init__(self, a, b, c):
        self.a, self.b, self.c, self.dim = np.asarray(a, dtype=np.float64), float(b), float(c), len(a)

    def neg_log_dens(self, q):
        return np.sum(np.log1p(np.exp(-self.a * q))) + 0.5 * self.b * q @ q + 0.5 * self.c * q.sum() ** 2

    def grad(self, q):
        return -self.a / (1.0 + np.exp(self.a * q)) + self.b * q + self.c * q.sum()


@pytest.mark.parametrize("dim,metric_kind", [(16, "identity"), (64, "diag"), (200, "dense")])
def test_builtin_target_as_user_source_reproduces_the_builtin(dim, metric_kind):
    rng = np.random.default_rng(dim)
    metric = {"identity": None, "diag": np.exp(0.2 * rng.standard_normal(dim)), "dense": omdl.make_spd(dim, rng)}[metric_kind]
    built = systems.EuclideanMetricSystem(models.Banana(dim), metric=metric)
    user = systems.EuclideanMetricSystem(models.UserTarget(dim, BANANA_SRC), metric=metric)
    n, steps, h = 9, 20, 0.02
    q0 = rng.standard_normal((n, dim))
    p0 = built.sample_momentum_batch(q0, rng.standard_normal((n, dim)))
    assert np.array_equal(user.sample_momentum_batch(q0, rng.standard_normal((n, dim))).shape, p0.shape)
    dirs = np.where(np.arange(n) % 2 == 0, 1, -1).astype(np.int8)
    qb, pb, _, _ = integrators.LeapfrogIntegrator(built, h).step_batch(q0, p0, dirs, n_steps=steps)
    qu, pu, st, nd = integrators.LeapfrogIntegrator(user, h).step_batch(q0, p0, dirs, n_steps=steps)
    assert np.all(st == 0) and np.all(nd == steps)
    assert_close(qu, qb, 1e-13 * steps, "user-source banana q")
    assert_close(pu, pb, 1e-13 * steps, "user-source banana p")
    assert_close(user.h_batch(qu, pu), built.h_batch(qb, pb), 1e-12, "h")
    qb, pb, _, _ = integrators.BCSSThreeStageIntegrator(built, h).step_batch(q0, p0, dirs, n_steps=5)
    qu, pu, _, _ = integrators.BCSSThreeStageIntegrator(user, h).step_batch(q0, p0, dirs, n_steps=5)
    assert_close(qu, qb, 1e-12, "BCSS q")
    assert_close(pu, pb, 1e-12, "BCSS p")


def test_new_target_matches_oracle_and_samples():
    rng = np.random.default_rng(8)
    dim, n, h, steps = 24, 40, 0.1, 15
    a = rng.uniform(0.5, 2.0, dim)
    b, c = 0.7, 0.05
    twin = LogisticTwin(a, b, c)
    metric = np.exp(0.1 * rng.standard_normal(dim))
    system = systems.EuclideanMetricSystem(models.UserTarget(dim, LOGISTIC_SRC, np.concatenate([a, [b, c]])),
                                           metric=metric)
    osys = orc.EuclidSystem(twin, omdl.METRIC_DIAG, metric)
    q0 = rng.standard_normal((n, dim))
    p0 = np.stack([osys.msqrt(z) for z in rng.standard_normal((n, dim))])
    integ = integrators.LeapfrogIntegrator(system, h)
    q, p, st, nd = integ.step_batch(q0, p0, 1, n_steps=steps)
    for cidx in range(0, n, 7):
        qo, po = orc.leapfrog_steps(osys, q0[cidx], p0[cidx], h, steps)
        assert_close(q[cidx], qo, 1e-12, f"q chain {cidx}")
        assert_close(p[cidx], po, 1e-12, f"p chain {cidx}")
        assert_close(system.h_batch(q[cidx:cidx + 1], p[cidx:cidx + 1])[0], osys.h(q[cidx], p[cidx]), 1e-12, "h")
    # a device-resident HMC transition on the user model (device draws: nothing uploaded per transition)
    ctx = default_context()
    batch = DeviceBatch(ctx, n, dim)
    batch.upload(q0, p0, np.ones(n, dtype=np.int8))
    batch.set_rng(5, 0)
    tr = transitions.MetropolisStaticIntegrationTransition(system, integ, 5)
    mom = transitions.IndependentMomentumTransition(system)
    acc = []
    for t in range(10):
        mom.sample_batch_device(batch, t)
        acc.append(tr.sample_batch_device(batch, t)["accept_stat"].mean())
    assert np.mean(acc) > 0.7
    batch.close()


def test_compile_errors_and_unsupported_systems_fail_loudly():
    with pytest.raises(ValueError):
        models.UserTarget(4, "__device__ double f() { return 0; }")
    bad = models.UserTarget(4, BANANA_SRC.replace("return g;", "return g + undefined_symbol;"))
    with pytest.raises(DeviceError, match="undefined_symbol"):
        systems.EuclideanMetricSystem(bad).h_batch(np.zeros((1, 4)), np.zeros((1, 4)))
    with pytest.raises((DeviceError, TypeError, ValueError)):
        sysr = systems.DenseRiemannianMetricSystem(models.UserTarget(4, BANANA_SRC), models.DiagQuadMetric(4))
        integrators.ImplicitLeapfrogIntegrator(sysr, 0.1).step_batch(np.zeros((1, 4)), np.ones((1, 4)), 1, 1)


# ---- user CONSTRAINTS (reference systems.py:786-792: `constr` / `jacob_constr` are callables) -----------
def test_torus_constraint_as_user_source_reproduces_the_builtin():
    """The built-in torus constraint written as user source (with the library's own sqrt / reciprocal helper, which the
    run-time translation unit can call) goes through the SAME constrained-leapfrog core, compiled at run time: states,
    statuses, step counts and Newton iteration counts equal the built-in kernel's."""
    from oracle import models as omdl
    from user_sources import TORUS_AS_USER

    rng = np.random.default_rng(5)
    n, h, steps = 512, 0.1, 30
    q0 = omdl.torus_init(n, rng)
    builtin = systems.DenseConstrainedEuclideanMetricSystem(models.Torus(), models.TorusConstr())
    user = systems.DenseConstrainedEuclideanMetricSystem(models.Torus(), models.UserConstraint(1, TORUS_AS_USER, [1.0, 0.5]))
    z = rng.standard_normal((n, 3))
    p0 = builtin.sample_momentum_batch(q0, z)
    assert np.array_equal(user.sample_momentum_batch(q0, z), p0)  # the cotangent projection, run-time compiled
    ib = integrators.ConstrainedLeapfrogIntegrator(builtin, h)
    iu = integrators.ConstrainedLeapfrogIntegrator(user, h)
    qb, pb, sb, nb = ib.step_batch(q0, p0, 1, n_steps=steps)
    qu, pu, su, nu = iu.step_batch(q0, p0, 1, n_steps=steps)
    assert np.array_equal(sb, su) and np.array_equal(nb, nu)
    assert ib.last_counters["n_newton_iters"] == iu.last_counters["n_newton_iters"]
    # same arithmetic, but two different instantiations of the core (selectors folded at compile time there, tested
    # at run time here): fused-multiply-add contraction may differ by an ulp per operation
    assert_close(qu, qb, 1e-12, "positions")
    assert_close(pu, pb, 1e-12, "momenta")
    assert_close(user.h_batch(qu, pu), builtin.h_batch(qb, pb), 1e-12, "hamiltonian")


def test_user_constraint_errors_fail_loudly():
    from user_sources import ELLIPSOID_SADDLE

    with pytest.raises(ValueError):
        models.UserConstraint(1, "__device__ void f() {}")
    broken = models.UserConstraint(2, ELLIPSOID_SADDLE.replace("c[0] = s - 1.0;", "c[0] = s - undefined_thing;"),
                                   np.ones(6))
    with pytest.raises(DeviceError, match="undefined_thing"):
        systems.DenseConstrainedEuclideanMetricSystem(models.Poly(5, 0.5, 0.25), broken).device_model()
    with pytest.raises(DeviceError, match="n_constr"):  # as many constraints as dimensions
        systems.DenseConstrainedEuclideanMetricSystem(
            models.Poly(2, 0.5, 0.25), models.UserConstraint(2, ELLIPSOID_SADDLE, np.ones(3))).device_model()
    # the built-in funnel target has no gradient inside the constrained core; a user constraint (whose
    # run-time compiled kernels were reached before the built-in refusal) must be refused too: rc = MM_ERR_UNSUPPORTED
    for con in (models.UserConstraint(2, ELLIPSOID_SADDLE, np.ones(6)), models.SphereConstr()):
        with pytest.raises(DeviceError, match=r"rc=-3.*funnel"):
            systems.DenseConstrainedEuclideanMetricSystem(models.Funnel(np.linspace(0.5, 2.0, 4)), con).device_model()


def test_device_transitions_run_on_a_user_constraint():
    """Static HMC with device draws on the user-constrained system: the chains stay on the manifold."""
    from oracle import models as omdl
    from user_sources import ELLIPSOID_SADDLE

    rng = np.random.default_rng(8)
    d, n = 5, 256
    con = omdl.EllipsoidSaddleConstr(np.exp(0.3 * rng.standard_normal(d)), 0.3)
    system = systems.DenseConstrainedEuclideanMetricSystem(models.Poly(d, 0.5, 0.25),
                                                           models.UserConstraint(2, ELLIPSOID_SADDLE, con.params()))
    integ = integrators.ConstrainedLeapfrogIntegrator(system, 0.1)
    tr = transitions.MetropolisStaticIntegrationTransition(system, integ, 5)
    mom = transitions.IndependentMomentumTransition(system)
    ctx = default_context()
    batch = DeviceBatch(ctx, n, d)
    q0 = con.init(n, rng)
    batch.upload(q0, np.zeros((n, d)), np.ones(n, dtype=np.int8))
    batch.set_rng(12, 0)
    acc = []
    for t in range(10):
        mom.sample_batch_device(batch, t)
        acc.append(tr.sample_batch_device(batch, t)["accept_stat"].mean())
    q, p, _ = batch.download()
    assert np.mean(acc) > 0.6
    assert np.max(np.abs([con.constr(x) for x in q])) < 1e-8
    assert np.max(np.abs([con.jacob_constr(x) @ y for x, y in zip(q, p)])) < 1e-8  # momenta in the cotangent space
    assert np.max(np.abs(q - q0)) > 0.05  # and the chains moved
    batch.close()


# ---- user METRICS (reference systems.py:1322-1358: `metric_func` / `vjp_metric_func` are callables) ------
@pytest.mark.parametrize("dim", [5, 8, 16, 27, 32, 40, 64])
def test_rank1_metric_as_user_source_reproduces_the_builtin(dim):
    """The built-in rank-one-update metric written as user source goes through the same wave-per-chain kernels
    (csrc/implicit_wave.h), compiled at run time: leapfrog and midpoint steps, h, dh_dmom and sample_momentum equal the
    built-in kernels' (the vector-Jacobian products are formed differently - through the user's generic V(i, j) - so the
    bar is rounding level, not bits), with identical statuses, step counts and fixed-point evaluation counts."""
    from user_sources import RANK1_AS_USER

    rng = np.random.default_rng(dim)
    n = 12
    B = omdl.make_spd(dim, rng)
    builtin = systems.DenseRiemannianMetricSystem(models.Banana(dim), models.Rank1Metric(B))
    user = systems.DenseRiemannianMetricSystem(models.Banana(dim), models.UserMetric(dim, RANK1_AS_USER, B))
    q0 = rng.standard_normal((n, dim))
    z = rng.standard_normal((n, dim))
    p0 = builtin.sample_momentum_batch(q0, z)
    assert_close(user.sample_momentum_batch(q0, z), p0, 1e-13, "sample_momentum")
    assert_close(user.h_batch(q0, p0), builtin.h_batch(q0, p0), 1e-13, "h")
    assert_close(user.dh_dmom_batch(q0, p0), builtin.dh_dmom_batch(q0, p0), 1e-13, "dh_dmom")
    for cls, h, steps in ((integrators.ImplicitLeapfrogIntegrator, 0.03, 10), (integrators.ImplicitMidpointIntegrator, 0.03, 5)):
        ib, iu = cls(builtin, h), cls(user, h)
        qb, pb, sb, nb = ib.step_batch(q0, p0, 1, n_steps=steps)
        qu, pu, su, nu = iu.step_batch(q0, p0, 1, n_steps=steps)
        assert np.array_equal(sb, su) and np.array_equal(nb, nu) and np.all(sb == 0)
        assert ib.last_counters["n_fp_evals"] == iu.last_counters["n_fp_evals"]
        assert_close(qu, qb, 1e-11, f"{cls.__name__} positions")
        assert_close(pu, pb, 1e-11, f"{cls.__name__} momenta")


def test_user_metric_errors_fail_loudly():
    from user_sources import RANK1_AS_USER

    with pytest.raises(ValueError):
        models.UserMetric(4, "__device__ double f() { return 0; }")
    bad = models.UserMetric(4, RANK1_AS_USER.replace("return s / (double)dim;", "return s / undefined_dim;"), np.eye(4))
    with pytest.raises(DeviceError, match="undefined_dim"):
        systems.DenseRiemannianMetricSystem(models.Banana(4), bad).device_model()
    with pytest.raises(DeviceError, match="dim <= 64"):
        systems.DenseRiemannianMetricSystem(models.Banana(70), models.UserMetric(70, RANK1_AS_USER, np.eye(70))).device_model()
