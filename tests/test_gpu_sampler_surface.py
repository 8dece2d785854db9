"""GPU: the integrators expose exactly the surface mici.transitions / samplers consume
(``integrator.step``, ``integrator.step_size`` read/write, ``system.h``, ``system.dh_dmom``,
``system.sample_momentum``), exercised by a static Metropolis HMC transition written only against
that surface (the logic of reference transitions.py:129-198, 275-315), plus the RCCL trace gather
with a one-rank communicator (the only world size a 1-GPU box allows)."""

import ctypes as C

import numpy as np
import pytest

from mici_amd import _ffi, distributed, integrators, models, systems
from mici_amd.errors import IntegratorError
from mici_amd.runtime import DeviceBatch, default_context
from mici_amd.states import ChainState

pytestmark = pytest.mark.gpu


def metropolis_hmc_transition(system, integrator, state, n_step, rng):
    """Independent momentum refresh + static-length Metropolis-corrected trajectory."""
    state.mom = system.sample_momentum(state, rng)            # transitions.py:136-142
    h_init = system.h(state)                                 # :281
    state_p = state
    n_done = 0
    try:
        for _ in range(n_step):                              # :289-291
            state_p = integrator.step(state_p)
            n_done += 1
    except IntegratorError:                                  # :292-295
        return state, dict(accept_stat=0.0, n_step=n_done, failed=True)
    accept = np.exp(min(0.0, h_init - system.h(state_p)))    # :300-309
    if rng.uniform() < accept:
        state = state_p
    return state, dict(accept_stat=accept, n_step=n_done, failed=False)


def run_chain(system, integrator, pos0, n_sample, n_step, seed):
    rng = np.random.default_rng(seed)
    state = ChainState(pos=np.array(pos0, dtype=np.float64), mom=None, dir=1)
    trace, accepts = [], []
    for _ in range(n_sample):
        state, stats = metropolis_hmc_transition(system, integrator, state, n_step, rng)
        trace.append(state.pos.copy())
        accepts.append(stats["accept_stat"])
    return np.array(trace), np.array(accepts)


def test_static_hmc_on_gaussian_target_recovers_moments():
    # BASELINE config c1: EuclideanMetricSystem, iso-Gaussian D=32, LeapfrogIntegrator h=0.1...
    dim = 32
    system = systems.EuclideanMetricSystem(models.GaussIso(dim))
    integ = integrators.LeapfrogIntegrator(system, 0.25)
    trace, acc = run_chain(system, integ, np.zeros(dim), n_sample=300, n_step=8, seed=1)
    assert acc.mean() > 0.8
    assert abs(trace[50:].mean()) < 0.15
    assert abs(trace[50:].var() - 1.0) < 0.25


def test_adapter_style_step_size_mutation():
    # DualAveragingStepSizeAdapter mutates integrator.step_size between steps (adapters.py:322-340)
    system = systems.EuclideanMetricSystem(models.GaussDiag(np.full(4, 2.0)), metric=np.ones(4))
    integ = integrators.LeapfrogIntegrator(system, 0.1)
    s = ChainState(pos=np.ones(4), mom=np.ones(4), dir=1)
    a = integ.step(s)
    integ.step_size = 0.2
    b = integ.step(s)
    assert not np.allclose(a.pos, b.pos)
    assert system.dh_dmom(s).shape == (4,)


def test_riemannian_and_constrained_transitions_run():
    rng = np.random.default_rng(0)
    dim = 8
    B = np.eye(dim) + 0.1 * np.ones((dim, dim))
    system = systems.DenseRiemannianMetricSystem(models.Poly(dim, 1.0, 1.0 / 3.0), models.Rank1Metric(B))
    integ = integrators.ImplicitLeapfrogIntegrator(system, 0.1)
    trace, acc = run_chain(system, integ, 0.1 * rng.standard_normal(dim), 30, 3, seed=2)
    assert np.all(np.isfinite(trace)) and acc.mean() > 0.5
    system = systems.DenseConstrainedEuclideanMetricSystem(models.Torus(), models.TorusConstr())
    integ = integrators.ConstrainedLeapfrogIntegrator(system, 0.1)
    trace, acc = run_chain(system, integ, np.array([1.5, 0.0, 0.0]), 50, 5, seed=3)
    rho = np.sqrt(trace[:, 0] ** 2 + trace[:, 1] ** 2)
    assert np.max(np.abs((rho - 1.0) ** 2 + trace[:, 2] ** 2 - 0.25)) < 1e-8  # stays on the torus


def test_rccl_trace_gather_single_rank():
    ctx = default_context()
    rng = np.random.default_rng(0)
    q = rng.standard_normal((33, 7))
    batch = DeviceBatch(ctx, 33, 7)
    batch.upload(q, q, 1)
    raw = (C.c_uint8 * _ffi.MM_COMM_ID_BYTES)()
    _ffi.check(ctx._lib.mm_comm_unique_id(raw), None, "mm_comm_unique_id")
    gather = distributed.RcclTraceGather(ctx, 0, 1, raw)
    out = gather.gather(batch)
    assert np.array_equal(out, q)
    # overlapped form: the snapshot is taken before the state moves on
    gather.gather_async(batch, want_host=True)
    batch.upload(q + 1.0, None, None)
    assert np.array_equal(gather.wait(want_host=True), q)
    gather.gather_async(batch, want_host=False)
    gather.wait(want_host=False)
    gather.close()
    batch.close()
