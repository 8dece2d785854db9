"""GPU: device-side random draws for the transitions (include/mici_amd.h, mm_state_set_rng; csrc/k_rng.hip) against
the NumPy restatement oracle/rng.py - bit-exact for the integer work (uniform bits, step counts), to libm rounding
for the Box-Muller normals - and the device-draw transitions against the host-draw (parity-mode) ones fed the very
same numbers."""

import numpy as np
import pytest

from conftest import assert_close
from oracle import models as omdl
from oracle import rng as orng

from mici_amd import integrators, models, systems, transitions
from mici_amd.runtime import DeviceBatch, default_context

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,dim,offset", [(37, 5, 0), (1024, 128, 4096), (3, 1, 2 ** 40 + 7)])
def test_draws_match_oracle(n, dim, offset):
    ctx = default_context()
    batch = DeviceBatch(ctx, n, dim)
    seed = 0xDEADBEEF12345678
    batch.set_rng(seed, offset)
    for t in (0, 5, 2 ** 33 + 1):
        z, u, steps = batch.rng_draws(t, 3, 11)
        assert np.array_equal(u, orng.uniform(seed, offset, n, t))           # integer work: bit-exact
        assert np.array_equal(steps, orng.steps(seed, offset, n, t, 3, 11))
        zo = orng.normal(seed, offset, n, dim, t)
        assert np.max(np.abs(z - zo)) < 1e-13                                # log / cos / sin of the device libm
    batch.close()


def test_streams_do_not_depend_on_sharding():
    ctx = default_context()
    whole = DeviceBatch(ctx, 10, 7)
    a, b = DeviceBatch(ctx, 6, 7), DeviceBatch(ctx, 4, 7)
    for bt, off in ((whole, 100), (a, 100), (b, 106)):
        bt.set_rng(99, off)
    zw, uw, sw = whole.rng_draws(12, 1, 9)
    za, ua, sa = a.rng_draws(12, 1, 9)
    zb, ub, sb = b.rng_draws(12, 1, 9)
    assert np.array_equal(zw, np.concatenate([za, zb])) and np.array_equal(uw, np.concatenate([ua, ub]))
    assert np.array_equal(sw, np.concatenate([sa, sb]))
    for bt in (whole, a, b):
        bt.close()


@pytest.mark.parametrize("kind", ["static", "random", "correlated"])
def test_device_draw_transition_equals_host_draw_transition(kind):
    """Same numbers, two routes: (a) drawn on the device inside the transition; (b) downloaded with rng_draws and fed
    to the parity-mode (host-draw) entry points.  Bitwise equal chains."""
    rng = np.random.default_rng(3)
    dim, n = 12, 33
    P = omdl.make_spd(dim, rng)
    system = systems.EuclideanMetricSystem(models.GaussDense(P), metric=np.exp(0.2 * rng.standard_normal(dim)))
    integ = integrators.LeapfrogIntegrator(system, 0.3)
    if kind == "random":
        tr = transitions.MetropolisRandomIntegrationTransition(system, integ, (2, 7))
    else:
        tr = transitions.MetropolisStaticIntegrationTransition(system, integ, 4)
    mom = (transitions.CorrelatedMomentumTransition(system, 0.7) if kind == "correlated"
           else transitions.IndependentMomentumTransition(system))
    ctx = default_context()
    q0, p0 = rng.standard_normal((2, n, dim))
    dev, host = DeviceBatch(ctx, n, dim), DeviceBatch(ctx, n, dim)
    for b in (dev, host):
        b.upload(q0, p0, np.ones(n, dtype=np.int8))
        b.set_rng(2024, 500)
    for t in range(6):
        lo, hi = (tr.n_step_range if kind == "random" else (1, 2))
        z, u, steps = host.rng_draws(t, lo, hi)
        mom.sample_batch_device(dev, t)
        sd = tr.sample_batch_device(dev, t)
        mom.sample_batch(host, z)
        sh = tr.sample_batch(host, u, n_step=steps) if kind == "random" else tr.sample_batch(host, u)
        for k in ("n_step", "metrop_accept_prob", "accepted"):
            assert np.array_equal(sd[k], sh[k]), (kind, t, k)
        qd, pd, dd = dev.download()
        qh, ph, dh = host.download()
        assert np.array_equal(qd, qh) and np.array_equal(pd, ph) and np.array_equal(dd, dh)
    assert 0.2 < sd["accepted"].mean() <= 1.0
    dev.close()
    host.close()


def test_zero_upload_sampling_recovers_the_target():
    """Momentum refresh + static HMC with device draws and NO per-transition host traffic at all (stats=False):
    a dense Gaussian target's mean and covariance from 2048 chains after 60 transitions."""
    rng = np.random.default_rng(4)
    dim, n = 8, 2048
    P = omdl.make_spd(dim, rng)
    system = systems.EuclideanMetricSystem(models.GaussDense(P))
    integ = integrators.LeapfrogIntegrator(system, 0.25)
    tr = transitions.MetropolisStaticIntegrationTransition(system, integ, 6)
    mom = transitions.IndependentMomentumTransition(system)
    ctx = default_context()
    batch = DeviceBatch(ctx, n, dim)
    batch.upload(np.zeros((n, dim)), np.zeros((n, dim)), np.ones(n, dtype=np.int8))
    batch.set_rng(7, 0)
    for t in range(60):
        mom.sample_batch_device(batch, t)
        assert tr.sample_batch_device(batch, t, stats=False) is None
    q, _, _ = batch.download()
    cov = np.linalg.inv(P)
    assert np.max(np.abs(q.mean(0))) < 5 * np.sqrt(np.max(np.diag(cov)) / n)
    assert_close(np.cov(q.T), cov, 0.15, "sample covariance")
    batch.close()


def test_sticky_error_word_survives_stats_free_transitions():
    """with ``stats=False`` nothing is downloaded per transition and the proposal's status is overwritten
    by the next transition's copy - the accept step therefore ORs every failed proposal's status into the batch's
    error word on the device.  Same seeds, one batch with statistics and one without: the word equals the OR of the
    per-transition flags; a status-5 proposal (LinAlgError outside a solver, which the reference lets propagate)
    is raised by ``check_errors`` at the caller's synchronisation point."""
    from mici_amd.errors import LinAlgError

    rng = np.random.default_rng(11)
    dim, n, T = 8, 64, 5
    system = systems.DenseRiemannianMetricSystem(models.Banana(dim), models.Rank1Metric(omdl.make_spd(dim, rng)))
    # a solver that gives up early: a good share of the trajectories fails with ConvergenceError
    integ = integrators.ImplicitLeapfrogIntegrator(system, 0.5, fixed_point_solver_kwargs=dict(max_iters=7))
    tr = transitions.MetropolisStaticIntegrationTransition(system, integ, 3)
    mom = transitions.IndependentMomentumTransition(system)
    ctx = default_context()
    q0 = rng.standard_normal((n, dim))
    p0 = system.sample_momentum_batch(q0, rng.standard_normal((n, dim)))
    with_stats, without = DeviceBatch(ctx, n, dim), DeviceBatch(ctx, n, dim)
    scale = np.where(np.arange(n) % 2 == 0, 0.02, 1.0)  # even chains: a step size the early-quitting solver manages
    for b in (with_stats, without):
        b.upload(q0, p0, np.ones(n, dtype=np.int8))
        b.set_rng(99, 0)
        b.set_step_scale(scale)
    assert not without.download_errors().any()  # nothing has run yet
    conv = np.zeros(n, dtype=bool)
    nonrev = np.zeros(n, dtype=bool)
    for t in range(T):
        mom.sample_batch_device(with_stats, t)
        st = tr.sample_batch_device(with_stats, t)
        conv |= st["convergence_error"]
        nonrev |= st["non_reversible_step"]
        mom.sample_batch_device(without, t)
        assert tr.sample_batch_device(without, t, stats=False) is None
    assert conv.any() and not conv.all()  # the scenario exercises both kinds of chains
    errs = tr.check_errors(without, clear=False)
    assert np.array_equal((errs & 0b1110) != 0, conv)
    assert np.array_equal((errs & (1 << 4)) != 0, nonrev)
    assert not (errs & (1 << 5)).any()
    assert np.array_equal(without.download_errors(clear=True), errs)
    assert not without.download_errors().any()  # cleared
    a, b = with_stats.download(), without.download()
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    # status 5: a position whose metric is not finite
    q_bad = q0.copy()
    q_bad[3] = 1e200
    without.upload(q_bad, p0, np.ones(n, dtype=np.int8))
    mom_keep = p0  # keep the momenta: the refresh itself would fail on that chain
    tr.sample_batch_device(without, T, stats=False)
    with pytest.raises(LinAlgError, match=r"\[3\]"):
        tr.check_errors(without)
    assert mom_keep is p0
    with_stats.close()
    without.close()


def test_random_length_device_transition_keeps_its_buffers_and_restores_n_step():
    """the random-length transition no longer frees + re-allocates the per-chain step counts (a stream
    synchronisation) every transition, and leaves ``n_step`` as it found it."""
    rng = np.random.default_rng(3)
    dim, n = 6, 128
    system = systems.EuclideanMetricSystem(models.GaussDense(omdl.make_spd(dim, rng)))
    integ = integrators.LeapfrogIntegrator(system, 0.2)
    tr = transitions.MetropolisRandomIntegrationTransition(system, integ, (2, 9))
    ctx = default_context()
    batch = DeviceBatch(ctx, n, dim)
    batch.upload(rng.standard_normal((n, dim)), rng.standard_normal((n, dim)), np.ones(n, dtype=np.int8))
    batch.set_rng(5, 0)
    before = tr.n_step
    seen = set()
    for t in range(4):
        st = tr.sample_batch_device(batch, t)
        seen |= set(st["n_step"].tolist())
        assert tr.n_step == before
        # the counts are switched off again: a plain integrator call advances every chain by the common length
        q0, p0, _ = batch.download()
        integ.step_device(batch, 3, ctx)
        _, nd = batch.download_status()
        assert np.all(nd == 3)
        batch.upload(q0, p0, np.ones(n, dtype=np.int8))
    assert seen <= set(range(2, 9)) and len(seen) > 3
    batch.close()


def test_worker_thread_contexts_are_closed_and_evicted_when_the_thread_ends():
    """the default context of a worker thread (stream, pinned staging buffer) and the device model /
    buffers a SHARED system cached for it go when the thread ends, instead of accumulating behind the system."""
    import gc
    import threading

    from mici_amd.states import ChainState

    system = systems.EuclideanMetricSystem(models.GaussIso(4))
    integ = integrators.LeapfrogIntegrator(system, 0.1)
    seen = []

    def work():
        s = integ.step(ChainState(pos=np.ones(4), mom=np.ones(4), dir=1))
        seen.append((default_context(), float(s.pos[0])))

    for _ in range(3):
        th = threading.Thread(target=work)
        th.start()
        th.join()
        gc.collect()
    assert len(seen) == 3 and len({id(c) for c, _ in seen}) >= 1
    assert all(c.handle is None for c, _ in seen)  # closed with their threads
    assert len(system._device) == 0 and len(integ._one) == 0  # and evicted from the shared caches
    assert len({v for _, v in seen}) == 1
