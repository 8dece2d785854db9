"""GPU: per-chain trajectory lengths (mm_state_set_chain_steps).  One launch with steps[N] set must equal N
separate runs of steps[i] steps each -- through every integrator family -- and MetropolisRandomIntegrationTransition
(transitions.py:355-402) must reproduce chain by chain what its single-chain `sample` does with the same
generators."""

import copy

import numpy as np
import pytest

from conftest import assert_close
from oracle import models as omdl

from mici_amd import integrators, models, solvers, systems, transitions
from mici_amd.runtime import DeviceBatch, default_context
from mici_amd.states import ChainState

pytestmark = pytest.mark.gpu


def _cases():
    rng = np.random.default_rng(7)
    P = omdl.make_spd(128, rng)
    yield "leapfrog-mfma", integrators.LeapfrogIntegrator(
        systems.EuclideanMetricSystem(models.GaussDense(P)), 0.05), 37, 128
    yield "leapfrog-mfma-dense-metric", integrators.LeapfrogIntegrator(
        systems.EuclideanMetricSystem(models.GaussDense(P[:48, :48]), metric=omdl.make_spd(48, rng)), 0.05), 20, 48
    yield "leapfrog-elem", integrators.LeapfrogIntegrator(
        systems.EuclideanMetricSystem(models.Poly(33, 1.0, 0.25), metric=np.exp(0.2 * rng.standard_normal(33))),
        0.1), 50, 33
    yield "leapfrog-generic", integrators.LeapfrogIntegrator(
        systems.EuclideanMetricSystem(models.Banana(20)), 0.01), 9, 20
    yield "bcss3-mfma", integrators.BCSSThreeStageIntegrator(
        systems.EuclideanMetricSystem(models.GaussDense(P[:64, :64])), 0.1), 35, 64
    yield "bcss2-gaussian-generic", integrators.BCSSTwoStageIntegrator(
        systems.GaussianEuclideanMetricSystem(models.Poly(12, 0.0, 0.5), metric=omdl.make_spd(12, rng)), 0.2), 9, 12
    yield "midpoint-euclid", integrators.ImplicitMidpointIntegrator(
        systems.EuclideanMetricSystem(models.Poly(9, 0.5, 0.5)), 0.1), 9, 9
    yield "implicit-wave", integrators.ImplicitLeapfrogIntegrator(
        systems.DenseRiemannianMetricSystem(models.Poly(16, 1.0, 1.0 / 3.0), models.DiagQuadMetric(16)), 0.05), 9, 16
    yield "implicit-mfma-wave", integrators.ImplicitLeapfrogIntegrator(
        systems.DenseRiemannianMetricSystem(models.Banana(40), models.Rank1Metric(omdl.make_spd(40, rng))), 0.01), 6, 40
    yield "implicit-mfma-team", integrators.ImplicitLeapfrogIntegrator(
        systems.DenseRiemannianMetricSystem(models.Banana(100), models.Rank1Metric(omdl.make_spd(100, rng))),
        0.01), 5, 100
    yield "implicit-softabs", integrators.ImplicitLeapfrogIntegrator(
        systems.SoftAbsRiemannianMetricSystem(models.Poly(8, 1.0, 0.3)), 0.05), 5, 8
    yield "constrained", integrators.ConstrainedLeapfrogIntegrator(
        systems.DenseConstrainedEuclideanMetricSystem(models.Torus(), models.TorusConstr()), 0.1), 70, 3


@pytest.mark.parametrize("name,integ,n,dim", [pytest.param(*c, id=c[0]) for c in _cases()])
def test_one_launch_equals_separate_runs(name, integ, n, dim):
    rng = np.random.default_rng(3)
    system = integ.system
    if name == "constrained":
        q0 = omdl.torus_init(n, rng)
    else:
        q0 = 0.7 * rng.standard_normal((n, dim))
    p0 = system.sample_momentum_batch(q0, rng.standard_normal((n, dim)))
    dirs = np.where(rng.random(n) < 0.5, 1, -1).astype(np.int8)
    max_steps = 6
    steps = rng.integers(0, max_steps + 1, size=n).astype(np.int32)
    steps[0], steps[-1] = max_steps, 0
    ctx = default_context()
    batch = DeviceBatch(ctx, n, dim)
    batch.upload(q0, p0, dirs)
    batch.set_chain_steps(steps)
    integ.step_device(batch, max_steps, ctx)
    q, p, _, status, n_done = batch.download_all()
    batch.set_chain_steps(None)
    integ.step_device(batch, 0, ctx)  # the counts are gone: nothing moves, nothing breaks
    batch.close()
    for k in range(max_steps + 1):
        sel = np.flatnonzero(steps == k)
        if not len(sel):
            continue
        if k == 0:
            qk, pk, sk, nk = q0[sel], p0[sel], np.zeros(len(sel), int), np.zeros(len(sel), int)
        else:
            qk, pk, sk, nk = integ.step_batch(q0[sel], p0[sel], dirs[sel], n_steps=k)
        assert np.array_equal(status[sel], sk) and np.array_equal(n_done[sel], nk), (name, k)
        # identical arithmetic per chain: the MFMA paths sum in a fixed order, so this is bit-for-bit
        assert np.array_equal(q[sel], qk) and np.array_equal(p[sel], pk), (name, k)
    assert np.all((n_done == steps) | (status != 0))


def test_random_integration_transition_draws_per_chain():
    rng = np.random.default_rng(11)
    n, dim = 24, 10
    P = omdl.make_spd(dim, rng)
    system = systems.EuclideanMetricSystem(models.GaussDense(P))
    integ = integrators.LeapfrogIntegrator(system, 0.15)
    trans = transitions.MetropolisRandomIntegrationTransition(system, integ, (2, 9))
    q0 = rng.standard_normal((n, dim))
    p0 = rng.standard_normal((n, dim))
    seeds = np.random.SeedSequence(99).spawn(n)
    # reference contract, chain by chain: n_step = rng.integers(...), then u = rng.uniform()
    expect, n_steps = [], []
    for c in range(n):
        r = np.random.default_rng(seeds[c])
        st, stats = trans.sample(ChainState(pos=q0[c].copy(), mom=p0[c].copy(), dir=1), r)
        expect.append((st.pos.copy(), st.mom.copy(), st.dir, stats))
        n_steps.append(stats["n_step"])
    assert len(set(n_steps)) > 2
    # one launch: the same generators, drawn in the same order
    rngs = [np.random.default_rng(s) for s in seeds]
    drawn = np.array([int(r.integers(*trans.n_step_range)) for r in rngs], dtype=np.int32)
    u = np.array([r.uniform() for r in rngs])
    ctx = default_context()
    batch = DeviceBatch(ctx, n, dim)
    batch.upload(q0, p0, np.ones(n, dtype=np.int8))
    stats = trans.sample_batch(batch, u, n_step=drawn)
    q, p, d = batch.download()
    batch.close()
    for c in range(n):
        eq, ep, ed, es = expect[c]
        assert stats["n_step"][c] == es["n_step"] == drawn[c]
        assert_close(stats["metrop_accept_prob"][c], es["metrop_accept_prob"], 1e-12, "accept prob")
        assert_close(q[c], eq, 1e-13, f"pos chain {c}")
        assert_close(p[c], ep, 1e-13, f"mom chain {c}")
        assert d[c] == ed
    with pytest.raises(ValueError):
        trans.sample_batch(batch := DeviceBatch(ctx, 2, dim), [0.5, 0.5], n_step=np.array([0, 3]))
    batch.close()


def test_host_mapped_state_behaves_like_a_device_one():
    """mm_state_alloc_mapped: upload / step / copy / accept / download on pinned host memory the kernels access in
    place must give the same bits as the device-resident state."""
    rng = np.random.default_rng(5)
    n, dim = 6, 24
    P = omdl.make_spd(dim, rng)
    system = systems.EuclideanMetricSystem(models.GaussDense(P))
    integ = integrators.LeapfrogIntegrator(system, 0.1)
    trans = transitions.MetropolisStaticIntegrationTransition(system, integ, n_step=7)
    q0, p0 = rng.standard_normal((n, dim)), rng.standard_normal((n, dim))
    u = rng.uniform(size=n)
    ctx = default_context()
    out = []
    for mapped in (False, True):
        batch = DeviceBatch(ctx, n, dim, mapped=mapped)
        batch.upload(q0, p0, np.ones(n, dtype=np.int8))
        integ.step_device(batch, 3, ctx)
        stats = trans.sample_batch(batch, u)  # copies the state into a (device) proposal and back
        q, p, d, status, n_done = batch.download_all()
        q2, p2, d2 = batch.download()
        st2, nd2 = batch.download_status()
        assert np.array_equal(q, q2) and np.array_equal(p, p2) and np.array_equal(d, d2)
        assert np.array_equal(status, st2) and np.array_equal(n_done, nd2)
        out.append((q, p, d, stats["metrop_accept_prob"]))
        batch.close()
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)
