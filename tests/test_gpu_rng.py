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
