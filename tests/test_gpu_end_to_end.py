"""End-to-end device-resident HMC: warm-up with per-chain dual-averaging step-size adaptation, main stage with
the reduced step size, traces and statistics written in the reference's on-disk format - every piece of the
hot path and of SURVEY section 8f driven together, checked against the known posterior."""

import numpy as np
import pytest

from mici_amd import adapters, integrators, models, systems, traces, transitions
from mici_amd.runtime import DeviceBatch, default_context

pytestmark = pytest.mark.gpu


def run_chains(system, integ, q0, n_step, n_warm, n_main, rng, tmp_path):
    n, d = q0.shape
    ctx = default_context()
    batch = DeviceBatch(ctx, n, d)
    batch.upload(q0, np.zeros((n, d)), np.ones(n, dtype=np.int8))
    mom_tr = transitions.IndependentMomentumTransition(system)
    int_tr = transitions.MetropolisStaticIntegrationTransition(system, integ, n_step)
    adapter = adapters.DualAveragingStepSizeAdapter()
    mom_tr.sample_batch(batch, rng.standard_normal((n, d)))
    state = adapter.initialize_batch(batch, int_tr)
    for _ in range(n_warm):
        mom_tr.sample_batch(batch, rng.standard_normal((n, d)))
        adapter.update_batch(state, batch, int_tr.sample_batch(batch, rng.uniform(size=n)))
    adapter.finalize_batch(state, batch, int_tr)
    writer = traces.MemmapTraceWriter(tmp_path, n, n_main, {"pos": np.zeros(d), "hamiltonian": 0.0},
                                      {"integration_transition": int_tr})
    for it in range(n_main):
        mom_tr.sample_batch(batch, rng.standard_normal((n, d)))
        stats = int_tr.sample_batch(batch, rng.uniform(size=n))
        q, p, _ = batch.download()
        writer.write(it, {"pos": q, "hamiltonian": system.h_batch(q, p)}, {"integration_transition": stats})
    writer.flush()
    batch.close()
    return writer, integ.step_size


def test_static_hmc_on_a_dense_gaussian(tmp_path):
    rng = np.random.default_rng(11)
    d, n = 12, 512
    a = rng.standard_normal((d, d))
    prec = a @ a.T / d + np.eye(d)
    cov = np.linalg.inv(prec)
    system = systems.EuclideanMetricSystem(models.GaussDense(prec))
    integ = integrators.LeapfrogIntegrator(system)
    writer, step_size = run_chains(system, integ, 2.0 * rng.standard_normal((n, d)), 8, 60, 40, rng, tmp_path)
    tr, st = writer.file_paths()
    pos = np.stack([np.load(f) for f in tr["pos"]])  # [chain, iter, d], the reference's layout per file
    acc = np.stack([np.load(f) for f in st["integration_transition"]["accept_stat"]])
    assert 0.05 < step_size < 2.0
    assert 0.6 < acc.mean() < 0.95  # adapted towards the 0.8 target
    samples = pos[:, 10:].reshape(-1, d)
    assert np.abs(samples.mean(0)).max() < 0.08
    assert np.abs(np.cov(samples.T) - cov).max() < 0.12 * np.abs(cov).max() + 0.03
    n_step = np.stack([np.load(f) for f in st["integration_transition"]["n_step"]])
    assert n_step.dtype == np.int64 and np.all(n_step == 8)


def test_static_hmc_with_the_implicit_integrator(tmp_path):
    rng = np.random.default_rng(12)
    d, n = 6, 256
    system = systems.DenseRiemannianMetricSystem(models.Poly(d, 1.0, 1.0 / 3.0), models.DiagQuadMetric(d))
    integ = integrators.ImplicitLeapfrogIntegrator(system)
    writer, step_size = run_chains(system, integ, rng.standard_normal((n, d)), 4, 40, 20, rng, tmp_path)
    _, st = writer.file_paths()
    acc = np.stack([np.load(f) for f in st["integration_transition"]["accept_stat"]])
    conv = np.stack([np.load(f) for f in st["integration_transition"]["convergence_error"]])
    assert 0.02 < step_size < 2.0 and 0.5 < acc.mean() <= 1.0
    assert conv.dtype == bool and conv.mean() < 0.2
