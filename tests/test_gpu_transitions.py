"""GPU parity of the device-resident momentum refresh + Metropolis static-integration transition
(mm_sample_momentum, mm_state_copy, <integrator>, mm_metropolis_accept) against fixtures recorded from the
reference's IndependentMomentumTransition + MetropolisStaticIntegrationTransition (transitions.py:129-142,
275-352) with its own random draws."""

import numpy as np
import pytest

from conftest import assert_close, golden_names, load_golden
from mici_amd import integrators, models, systems, transitions
from mici_amd.runtime import DeviceBatch, default_context
from mici_amd.states import ChainState

pytestmark = pytest.mark.gpu


def build(g):
    d = g["q0"].shape[1]
    kind = str(g["kind"])
    target = models.target_from_id(g["target"], g["target_params"], d)
    h = float(g["step_size"])
    if kind == "transition_euclid":
        mk = int(g["metric_kind"])
        system = systems.EuclideanMetricSystem(target, metric=None if mk == models.METRIC_IDENTITY else g["metric"])
        if int(g["composition"]):
            integ = integrators.SymmetricCompositionIntegrator(system, list(g["free_coefficients"]), step_size=h)
        else:
            integ = integrators.LeapfrogIntegrator(system, h)
    elif kind == "transition_riemann":
        system = systems.DenseRiemannianMetricSystem(
            target, models.rmetric_from_id(g["rmetric"], g["rmetric_params"], d))
        integ = integrators.ImplicitLeapfrogIntegrator(system, h)
    else:
        system = systems.DenseConstrainedEuclideanMetricSystem(
            target, models.constr_from_id(g["constr"], g["constr_params"], g["q0"].shape[1]))
        integ = integrators.ConstrainedLeapfrogIntegrator(system, h)
    return system, integ


class ReplayRng:
    """Feeds the reference's recorded draws back, and checks they are requested in the reference's order."""

    def __init__(self, z, u):
        self.z, self.u, self.t, self.stage = z, u, 0, 0

    def standard_normal(self, size=None):
        assert self.stage == 0, "momentum draw out of order"
        self.stage = 1
        return self.z[self.t].copy()

    normal = standard_normal

    def uniform(self):
        assert self.stage == 1 and not np.isnan(self.u[self.t]), "uniform drawn where the reference drew none"
        self.stage = 2
        return float(self.u[self.t])

    def next_transition(self):
        assert self.stage == (1 if np.isnan(self.u[self.t]) else 2), "a recorded draw was not consumed"
        self.t, self.stage = self.t + 1, 0


STAT_KEYS = ("n_step", "accept_stat", "metrop_accept_prob", "convergence_error", "non_reversible_step")


@pytest.mark.parametrize("name", golden_names("transition"))
def test_batched_transitions_match_reference_fixture(name):
    g = load_golden(name)
    system, integ = build(g)
    n_tr, n, d = g["z"].shape
    ctx = default_context()
    batch = DeviceBatch(ctx, n, d)
    batch.upload(g["q0"], np.zeros((n, d)), np.ones(n, dtype=np.int8))
    mom_tr = transitions.IndependentMomentumTransition(system)
    int_tr = transitions.MetropolisStaticIntegrationTransition(system, integ, int(g["n_step"]))
    for t in range(n_tr):
        mom_tr.sample_batch(batch, g["z"][t])
        prop, status, n_done = int_tr.propose_batch(batch)
        assert np.array_equal(status != 0, np.isnan(g["u"][t])), f"{name} t{t}: integration errors differ"
        stats = int_tr.accept_batch(batch, prop, status, n_done, np.nan_to_num(g["u"][t], nan=0.5))
        q, p, dirs = batch.download()
        assert_close(q, g["q_out"][t], 1e-9, f"{name} q t{t}")
        assert_close(p, g["p_out"][t], 1e-9, f"{name} p t{t}")
        assert np.array_equal(dirs, g["dir_out"][t])
        for k in STAT_KEYS:
            assert_close(np.asarray(stats[k], dtype=np.float64), g[f"stat_{k}"][t], 1e-9, f"{name} {k} t{t}")
    batch.close()


@pytest.mark.parametrize("name", ["transition_euclid_quartic_d5_bigstep", "transition_riemann_diagquad_poly_d5_bigstep",
                                  "transition_constrained_torus"])
def test_single_chain_sample_consumes_the_reference_random_stream(name):
    g = load_golden(name)
    system, integ = build(g)
    n_tr, n, d = g["z"].shape
    mom_tr = transitions.IndependentMomentumTransition(system)
    int_tr = transitions.MetropolisStaticIntegrationTransition(system, integ, int(g["n_step"]))
    assert set(int_tr.statistic_types) >= set(STAT_KEYS) | {"step_size"}
    for c in range(min(n, 3)):
        rng = ReplayRng(g["z"][:, c], g["u"][:, c])
        state = ChainState(pos=g["q0"][c].copy(), mom=None, dir=1)
        for t in range(n_tr):
            state, none = mom_tr.sample(state, rng)
            assert none is None
            state, stats = int_tr.sample(state, rng)
            rng.next_transition()
            assert_close(state.pos, g["q_out"][t, c], 1e-9, f"{name} q t{t} c{c}")
            assert_close(state.mom, g["p_out"][t, c], 1e-9, f"{name} p t{t} c{c}")
            assert state.dir == g["dir_out"][t, c]
            for k in STAT_KEYS:
                assert_close(float(stats[k]), g[f"stat_{k}"][t, c], 1e-9, f"{name} {k}")
            assert stats["step_size"] == integ.step_size


def test_transition_argument_checks_and_invariance():
    system = systems.EuclideanMetricSystem(models.GaussIso(8))
    integ = integrators.LeapfrogIntegrator(system, 0.3)
    with pytest.raises(ValueError):
        transitions.MetropolisStaticIntegrationTransition(system, integ, 0)
    # the chain leaves N(0, I) invariant: 2000 chains x 30 transitions from an over-dispersed start
    rng = np.random.default_rng(5)
    n, d = 2000, 8
    ctx = default_context()
    batch = DeviceBatch(ctx, n, d)
    batch.upload(3.0 * rng.standard_normal((n, d)), np.zeros((n, d)), np.ones(n, dtype=np.int8))
    mom_tr = transitions.IndependentMomentumTransition(system)
    int_tr = transitions.MetropolisStaticIntegrationTransition(system, integ, 5)
    acc = []
    for _ in range(30):
        mom_tr.sample_batch(batch, rng.standard_normal((n, d)))
        acc.append(int_tr.sample_batch(batch, rng.uniform(size=n))["accept_stat"].mean())
    q, _, _ = batch.download()
    assert 0.8 < np.mean(acc[10:]) <= 1.0
    assert abs(q.mean()) < 0.05 and abs(q.var() - 1.0) < 0.06
    batch.close()


def test_correlated_momentum_and_random_trajectory_length_match_reference():
    """CorrelatedMomentumTransition (mm_momentum_refresh) + MetropolisRandomIntegrationTransition, single-chain
    contract with the reference's recorded draws (standard_normal, integers, uniform in that order)."""
    g = load_golden("corrmom_random_nstep_d10")
    n_tr, n, d = g["z"].shape
    system = systems.EuclideanMetricSystem(models.target_from_id(g["target"], g["target_params"], d), metric=g["metric"])
    integ = integrators.LeapfrogIntegrator(system, float(g["step_size"]))
    mom_tr = transitions.CorrelatedMomentumTransition(system, float(g["coeff"]))
    int_tr = transitions.MetropolisRandomIntegrationTransition(system, integ, tuple(int(v) for v in g["n_step_range"]))

    class Replay:
        def __init__(self, c):
            self.c, self.t, self.log = c, 0, []

        def standard_normal(self, size=None):
            self.log.append("z")
            return g["z"][self.t, self.c].copy()

        def integers(self, lo, hi):
            assert (lo, hi) == tuple(int(v) for v in g["n_step_range"])
            self.log.append("n")
            return int(g["n_steps"][self.t, self.c])

        def uniform(self):
            self.log.append("u")
            return float(g["u"][self.t, self.c])

    for c in range(n):
        rng = Replay(c)
        state = ChainState(pos=g["q0"][c].copy(), mom=None, dir=1)
        for t in range(n_tr):
            rng.t, rng.log = t, []
            state, _ = mom_tr.sample(state, rng)
            state, stats = int_tr.sample(state, rng)
            assert rng.log == (["z", "n", "u"] if not np.isnan(g["u"][t, c]) else ["z", "n"])
            assert_close(state.pos, g["q_out"][t, c], 1e-9, f"q t{t} c{c}")
            assert_close(state.mom, g["p_out"][t, c], 1e-9, f"p t{t} c{c}")
            assert state.dir == g["dir_out"][t, c] and stats["n_step"] == g["n_steps"][t, c]
            assert_close(stats["accept_stat"], g["accept_stat"][t, c], 1e-9, "accept_stat")
    # batched partial refresh against the closed form
    ctx = default_context()
    batch = DeviceBatch(ctx, n, d)
    rs = np.random.default_rng(4)
    p0, z = rs.standard_normal((n, d)), rs.standard_normal((n, d))
    batch.upload(g["q0"], p0, 1)
    mom_tr.sample_batch(batch, z)
    coeff = float(g["coeff"])
    expect = np.sqrt(1 - coeff**2) * p0 + coeff * system.sample_momentum_batch(g["q0"], z)
    assert_close(batch.download()[1], expect, 1e-14, "partial refresh")
    with pytest.raises(ValueError):
        transitions.CorrelatedMomentumTransition(system, 1.5)
    with pytest.raises(ValueError):
        transitions.MetropolisRandomIntegrationTransition(system, integ, (3, 3))
    batch.close()
