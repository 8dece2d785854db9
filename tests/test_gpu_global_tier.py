"""Dense Riemannian metrics beyond D = 279: the global-memory tier (csrc/implicit_global.h).

The reference factorises any D (matrices.py:1117-1216, systems.py:1690-1734); rounds 1-4 stopped at the 279 x 279 a CU's
registers hold.  Here: leapfrog steps, h, dh_dmom and sample_momentum at 281 <= D <= 1024 against the oracle (the four
`riemann_global_*` reference fixtures run through tests/test_gpu_implicit.py), reversibility, run-to-run determinism, and the
refined solve-only constructions against fully factorised ones."""

import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, assert_close
from oracle import integrators as orc
from oracle import models as omdl

from mici_amd import integrators, models, systems

pytestmark = pytest.mark.gpu


def _pair(kind, dim, rng):
    if kind == "rank1":
        B = omdl.make_spd(dim, rng)
        return (systems.DenseRiemannianMetricSystem(models.Banana(dim), models.Rank1Metric(B)),
                orc.RiemannianSystem(omdl.Banana(dim), omdl.Rank1Metric(B)))
    return (systems.DenseRiemannianMetricSystem(models.Poly(dim, 1.0, 1.0 / 3.0), models.DiagQuadMetric(dim)),
            orc.RiemannianSystem(omdl.Poly(dim, 1.0, 1.0 / 3.0), omdl.DiagQuadMetric(dim)))


@pytest.mark.parametrize("kind,dim,n,h,steps", [
    ("rank1", 281, 3, 0.01, 3),     # first size of the tier (padded to 320)
    ("rank1", 400, 3, 0.01, 2),
    ("diagquad", 600, 2, 0.05, 2),
    ("rank1", 1024, 2, 0.005, 1),   # the largest: one flat element per thread of the workgroup
])
def test_global_tier_matches_oracle(kind, dim, n, h, steps):
    rng = np.random.default_rng(7000 + dim)
    system, osys = _pair(kind, dim, rng)
    q0 = rng.standard_normal((n, dim))
    z = rng.standard_normal((n, dim))
    p0 = np.stack([osys.sample_momentum(orc._State(q0[c], None), z[c]) for c in range(n)])
    assert_close(system.sample_momentum_batch(q0, z), p0, 1e-11, "sample_momentum (blocked Cholesky, L z)")
    assert_close(system.h_batch(q0, p0), [osys.h(orc._State(q0[c], p0[c])) for c in range(n)], 1e-11, "h")
    assert_close(system.dh_dmom_batch(q0, p0), [osys.dh2_dmom(orc._State(q0[c], p0[c])) for c in range(n)], 1e-11, "dh_dmom")
    integ = integrators.ImplicitLeapfrogIntegrator(system, h)
    dirs = np.ones(n, dtype=np.int8)
    dirs[1::2] = -1
    q, p, st, nd = integ.step_batch(q0, p0, dirs, n_steps=steps)
    counters = dict(integ.last_counters)
    assert np.all(st == 0) and np.all(nd == steps)
    for c in range(n):
        qo, po, so, no = orc.implicit_leapfrog_steps(osys, q0[c], p0[c], dirs[c] * h, steps)
        assert so == 0 and no == steps
        assert_close(q[c], qo, 1e-10, f"q chain {c}")
        assert_close(p[c], po, 1e-10, f"p chain {c}")
    assert counters["n_fp_solves"] == 4 * n * steps
    if kind == "rank1":  # one sweep per step, the solve-only constructions refined (a rank-two perturbation: two CG steps).
        assert counters["n_factor_full"] <= n * (steps + 1) + 2, counters
    else:  # diag(1 + q^2) (round 6): every construction elementwise - no refinement pairs, no workspace
        assert counters["n_refine"] == 0, counters
    # reversible, and the same bits twice
    qb, pb, sb, nb = integ.step_batch(q, p, -dirs, n_steps=steps)
    assert np.all(sb == 0)
    assert_close(qb, q0, 1e-7, "reversed q")
    q2, p2, _, _ = integ.step_batch(q0, p0, dirs, n_steps=steps)
    assert np.array_equal(q, q2) and np.array_equal(p, p2)


def test_global_tier_refined_solves_equal_factorised_solves(tmp_path):
    """MICI_AMD_REFINE=0 (every construction a full blocked sweep) in a second process: same statuses, counts, states."""
    prog = (
        "import numpy as np, sys; sys.path.insert(0, %r)\n"
        "from mici_amd import integrators, models, systems\n"
        "rng = np.random.default_rng(11); d = 300\n"
        "a = rng.standard_normal((d, d)); B = a @ a.T / d + np.eye(d)\n"
        "s = systems.DenseRiemannianMetricSystem(models.Banana(d), models.Rank1Metric(B))\n"
        "q0 = rng.standard_normal((3, d)); p0 = s.sample_momentum_batch(q0, rng.standard_normal((3, d)))\n"
        "i = integrators.ImplicitLeapfrogIntegrator(s, 0.01)\n"
        "q, p, st, nd = i.step_batch(q0, p0, 1, n_steps=2)\n"
        "np.save(sys.argv[1], np.concatenate([q.ravel(), p.ravel(), st, nd, [i.last_counters['n_fp_evals'], i.last_counters['n_factor_full']]]))\n"
        % ROOT)
    outs = []
    for refine in ("1", "0"):
        path = str(tmp_path / f"r{refine}.npy")
        env = dict(os.environ, MICI_AMD_REFINE=refine)
        r = subprocess.run([sys.executable, "-c", prog, path], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(path))
    a, b = outs
    assert np.array_equal(a[-8:-1], b[-8:-1])  # statuses, completed steps, fixed-point evaluations
    assert b[-1] > 5 * a[-1]                   # the factorised run did a sweep per construction
    assert_close(a[:-8], b[:-8], 1e-11, "refined vs factorised states")


def test_global_tier_reports_the_reference_errors():
    from mici_amd.errors import DeviceError
    rng = np.random.default_rng(3)
    d = 300
    system, _ = _pair("rank1", d, rng)
    integ = integrators.ImplicitLeapfrogIntegrator(system, 0.01)
    q0 = rng.standard_normal((3, d))
    p0 = rng.standard_normal((3, d))
    q0[1, 5] = np.nan  # "Array is not finite." at the initial position: LinAlgError outside a solver (status 5)
    q, p, st, nd = integ.step_batch(q0, p0, 1, n_steps=1)
    assert st[1] == 5 and nd[1] == 0 and st[0] == 0 and st[2] == 0
    big = systems.DenseRiemannianMetricSystem(models.Banana(1025), models.Rank1Metric(np.eye(1025)))
    with pytest.raises(DeviceError):
        integrators.ImplicitLeapfrogIntegrator(big, 0.01).step_batch(np.zeros((1, 1025)), np.ones((1, 1025)), 1, 1)


def test_global_tier_implicit_midpoint_matches_oracle():
    """ImplicitMidpointIntegrator (integrators.py:547-681) beyond D = 279: the same backend, a full sweep per evaluation."""
    rng = np.random.default_rng(77)
    dim, n, h, steps = 300, 2, 0.02, 2
    system, osys = _pair("rank1", dim, rng)
    q0 = rng.standard_normal((n, dim))
    p0 = np.stack([osys.sample_momentum(orc._State(q0[c], None), z) for c, z in enumerate(rng.standard_normal((n, dim)))])
    integ = integrators.ImplicitMidpointIntegrator(system, h)
    q, p, st, nd = integ.step_batch(q0, p0, 1, n_steps=steps)
    assert np.all(st == 0) and np.all(nd == steps)
    for c in range(n):
        qo, po, so, no = orc.implicit_midpoint_steps(osys, q0[c], p0[c], h, steps)
        assert so == 0 and no == steps
        assert_close(q[c], qo, 1e-9, f"q chain {c}")
        assert_close(p[c], po, 1e-9, f"p chain {c}")


# ---- user metrics beyond D = 279: the same tier compiled at run time around the user's source (MM_RTC_FAM_GLOBAL) ------------
@pytest.mark.parametrize("dim,flat", [(300, False), (600, True)])
def test_user_metric_on_the_global_tier_matches_the_builtin_and_the_oracle(dim, flat):
    """The built-in rank-one metric written as user source (plain: entry-wise metric + V(i, j) accessor reading the
    workspace itself; flat: the team-form VJP on the tier's column-walk product) against the built-in kernels of the same
    tier and the oracle: h / dh_dmom / sample_momentum, leapfrog steps with identical statuses and evaluation counts."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from user_sources import RANK1_AS_USER, RANK1_AS_USER_FLAT

    rng = np.random.default_rng(8000 + dim)
    n, h, steps = 2, 0.01, 2
    B = omdl.make_spd(dim, rng)
    builtin = systems.DenseRiemannianMetricSystem(models.Banana(dim), models.Rank1Metric(B))
    user = systems.DenseRiemannianMetricSystem(models.Banana(dim),
                                               models.UserMetric(dim, RANK1_AS_USER_FLAT if flat else RANK1_AS_USER, B))
    q0 = rng.standard_normal((n, dim))
    z = rng.standard_normal((n, dim))
    p0 = builtin.sample_momentum_batch(q0, z)
    assert_close(user.sample_momentum_batch(q0, z), p0, 1e-12, "sample_momentum")
    assert_close(user.h_batch(q0, p0), builtin.h_batch(q0, p0), 1e-12, "h")
    assert_close(user.dh_dmom_batch(q0, p0), builtin.dh_dmom_batch(q0, p0), 1e-12, "dh_dmom")
    ib, iu = integrators.ImplicitLeapfrogIntegrator(builtin, h), integrators.ImplicitLeapfrogIntegrator(user, h)
    qb, pb, sb, nb = ib.step_batch(q0, p0, 1, n_steps=steps)
    qu, pu, su, nu = iu.step_batch(q0, p0, 1, n_steps=steps)
    assert np.array_equal(sb, su) and np.array_equal(nb, nu) and np.all(sb == 0)
    assert ib.last_counters["n_fp_evals"] == iu.last_counters["n_fp_evals"]
    assert_close(qu, qb, 1e-10, "positions")
    assert_close(pu, pb, 1e-10, "momenta")
    osys = orc.RiemannianSystem(omdl.Banana(dim), omdl.Rank1Metric(B))
    qo, po, so, no = orc.implicit_leapfrog_steps(osys, q0[0], p0[0], h, steps)
    assert so == 0 and no == steps
    assert_close(qu[0], qo, 1e-10, "positions vs oracle")
    assert_close(pu[0], po, 1e-10, "momenta vs oracle")
