"""GPU parity of the batched dual-averaging step-size adapter (mici_amd.adapters, per-chain step sizes through
mm_state_set_step_scale) against fixtures recorded from the reference's DualAveragingStepSizeAdapter driving
its own transitions with its own random draws (adapters.py:174-389)."""

import numpy as np
import pytest

from conftest import assert_close, golden_names, load_golden
from mici_amd import adapters, integrators, models, systems, transitions
from mici_amd.runtime import DeviceBatch, default_context
from mici_amd.states import ChainState

pytestmark = pytest.mark.gpu


def build(g):
    d = g["q0"].shape[1]
    target = models.target_from_id(g["target"], g["target_params"], d)
    if str(g["kind"]) == "adapt_euclid":
        mk = int(g["metric_kind"])
        system = systems.EuclideanMetricSystem(target, metric=None if mk == models.METRIC_IDENTITY else g["metric"])
        integ = integrators.LeapfrogIntegrator(system)
    else:
        system = systems.DenseRiemannianMetricSystem(target, models.rmetric_from_id(g["rmetric"], g["rmetric_params"], d))
        integ = integrators.ImplicitLeapfrogIntegrator(system)
    return system, integ


@pytest.mark.parametrize("name", golden_names("adapt_"))
def test_batched_adaptation_matches_reference_fixture(name):
    g = load_golden(name)
    system, integ = build(g)
    n_iters, n, d = g["z"].shape
    ctx = default_context()
    batch = DeviceBatch(ctx, n, d)
    batch.upload(g["q0"], np.zeros((n, d)), np.ones(n, dtype=np.int8))
    mom_tr = transitions.IndependentMomentumTransition(system)
    int_tr = transitions.MetropolisStaticIntegrationTransition(system, integ, int(g["n_step"]))
    adapter = adapters.DualAveragingStepSizeAdapter()
    mom_tr.sample_batch(batch, g["z_init"])
    state = adapter.initialize_batch(batch, int_tr)
    assert np.array_equal(state["step_size"], g["init_step_size"]), f"{name}: initial step-size search differs"
    assert integ.step_size == 1.0
    for t in range(n_iters):
        mom_tr.sample_batch(batch, g["z"][t])
        prop, status, n_done = int_tr.propose_batch(batch)
        assert np.array_equal(status != 0, np.isnan(g["u"][t])), f"{name} t{t}"
        stats = int_tr.accept_batch(batch, prop, status, n_done, np.nan_to_num(g["u"][t], nan=0.5))
        assert_close(stats["accept_stat"], g["accept_stat"][t], 1e-6, f"{name} accept_stat t{t}")
        adapter.update_batch(state, batch, stats)
        # rounding differences (FMA contraction) feed back through the adapted step size: 40 iterations amplify
        # 1e-13 to ~1e-8, hence the looser bound than the single-trajectory tests
        assert_close(state["step_size"], g["step_sizes"][t], 1e-6, f"{name} step sizes t{t}")
    q, _, _ = batch.download()
    assert_close(q, g["q_final"], 1e-5, f"{name} final positions")
    assert_close(state["smoothed_log_step_size"], g["smoothed_log_step_size"], 1e-6, "smoothed log step size")
    adapter.finalize_batch(state, batch, int_tr)
    assert_close(integ.step_size, float(g["final_step_size"]), 1e-6, "final step size")
    # the batch is back to one shared step size: a plain trajectory runs with it
    integ.step_device(batch, 2)
    batch.close()


def test_single_chain_adapter_contract():
    g = load_golden("adapt_euclid_dense_d16")
    system, integ = build(g)
    int_tr = transitions.MetropolisStaticIntegrationTransition(system, integ, int(g["n_step"]))
    mom_tr = transitions.IndependentMomentumTransition(system)
    adapter = adapters.DualAveragingStepSizeAdapter()

    class Replay:
        def __init__(self, zs, us):
            self.zs, self.us = list(zs), list(us)

        def standard_normal(self, size=None):
            return self.zs.pop(0).copy()

        def uniform(self):
            return float(self.us.pop(0))

    c = 1
    rng = Replay([g["z_init"][c]] + [z for z in g["z"][:, c]], [u for u in g["u"][:, c] if not np.isnan(u)])
    state = ChainState(pos=g["q0"][c].copy(), mom=None, dir=1)
    state.mom = system.sample_momentum(state, rng)
    adapt_state = adapter.initialize(state, int_tr)
    assert integ.step_size == g["init_step_size"][c]
    for t in range(10):
        state, _ = mom_tr.sample(state, rng)
        state, stats = int_tr.sample(state, rng)
        adapter.update(adapt_state, state, stats, int_tr)
        assert_close(integ.step_size, g["step_sizes"][t, c], 1e-7, f"step size t{t}")
    adapter.finalize(adapt_state, state, int_tr, rng)
    assert integ.step_size == pytest.approx(np.exp(adapt_state["smoothed_log_step_size"]))


def test_step_scale_is_per_chain_and_validated():
    from mici_amd.errors import DeviceError
    system = systems.EuclideanMetricSystem(models.GaussIso(6))
    integ = integrators.LeapfrogIntegrator(system, 1.0)
    rng = np.random.default_rng(3)
    q0, p0 = rng.standard_normal((4, 6)), rng.standard_normal((4, 6))
    ctx = default_context()
    batch = DeviceBatch(ctx, 4, 6)
    eps = np.array([0.1, 0.2, 0.05, 0.4])
    batch.upload(q0, p0, 1)
    batch.set_step_scale(eps)
    integ.step_device(batch, 3)
    q, p, _ = batch.download()
    for c in range(4):
        qc, pc, _, _ = integrators.LeapfrogIntegrator(system, float(eps[c])).step_batch(q0[c:c + 1], p0[c:c + 1], 1, 3)
        assert np.array_equal(q[c], qc[0]) and np.array_equal(p[c], pc[0])
    with pytest.raises(DeviceError):
        batch.set_step_scale([0.1, -1.0, 0.1, 0.1])
    batch.close()


# ---- metric adapters (reference adapters.py:392-644) -------------------------------------------------------------
@pytest.mark.parametrize("name", golden_names("metricadapt"))
def test_metric_adapter_installs_the_reference_metric_on_the_device(name):
    import types
    from mici_amd.runtime import DeviceBatch, default_context
    g = load_golden(name)
    which, multi = str(g["which"]), bool(g["multi"])
    cls = adapters.OnlineVarianceMetricAdapter if which == "variance" else adapters.OnlineCovarianceMetricAdapter
    n_updates, n_chains, dim = g["pos_seq"].shape
    ref_metric = g["metric"]
    ref_metric = np.diag(ref_metric) if (which == "variance" and ref_metric.ndim == 2) else ref_metric

    class Replay:
        def __init__(self, z):
            self.z = z

        def standard_normal(self, size=None):
            return self.z.copy()

    # the reference's contract: one adapter state per chain, list finalize
    system = systems.EuclideanMetricSystem(models.GaussIso(dim))
    transition = types.SimpleNamespace(system=system)
    adapter = cls()
    states = [ChainState(pos=g["pos_seq"][0, c].copy(), mom=np.zeros(dim), dir=1) for c in range(n_chains)]
    adapt_states = [adapter.initialize(states[c], transition) for c in range(n_chains)]
    for k in range(n_updates):
        for c in range(n_chains):
            states[c].pos = g["pos_seq"][k, c].copy()
            adapter.update(adapt_states[c], states[c], {}, transition)
    rngs = [Replay(z) for z in g["z"]]
    if multi:
        adapter.finalize(adapt_states, states, transition, rngs)
    else:
        adapter.finalize(adapt_states[0], states[0], transition, rngs[0])
    assert_close(system.metric, ref_metric, 1e-11, "installed metric")
    minv = np.linalg.inv(np.diag(system.metric) if system.metric.ndim == 1 else system.metric)
    for c in range(len(g["z"])):
        if which == "variance":
            assert_close(states[c].mom, g["mom"][c], 1e-13, f"resampled momentum {c}")
        assert_close(states[c].mom @ minv @ states[c].mom, g["z"][c] @ g["z"][c], 1e-10, "p.M^-1 p = z.z")
    # the new metric is live on the device: h uses it
    h = system.h_batch(g["pos_seq"][-1], np.stack([s.mom for s in states] + [np.zeros(dim)] * (n_chains - len(g["z"]))))
    expect = np.array([0.5 * q @ q for q in g["pos_seq"][-1]])
    expect[:len(g["z"])] += 0.5 * np.array([z @ z for z in g["z"]])
    assert_close(h, expect, 1e-10, "h under the adapted metric")
    if not multi:
        return
    # N device-resident chains: same estimate, momenta resampled on the device
    system2 = systems.EuclideanMetricSystem(models.GaussIso(dim))
    transition2 = types.SimpleNamespace(system=system2)
    ctx = default_context()
    batch = DeviceBatch(ctx, n_chains, dim)
    astate = cls().initialize_batch(batch)
    ad2 = cls()
    for k in range(n_updates):
        batch.upload(g["pos_seq"][k], np.zeros((n_chains, dim)), np.ones(n_chains, dtype=np.int8))
        ad2.update_batch(astate, batch)
    ad2.finalize_batch(astate, batch, transition2, g["z"])
    assert_close(system2.metric, system.metric, 1e-13, "batched estimate")
    _, mom, _ = batch.download()
    batch.close()
    assert_close(mom, np.stack([s.mom for s in states]), 1e-13, "batched momentum refresh")
