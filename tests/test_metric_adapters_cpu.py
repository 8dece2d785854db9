"""CPU: host logic of the metric adapters (mici_amd/adapters.py) against fixtures recorded from the reference's
OnlineVarianceMetricAdapter / OnlineCovarianceMetricAdapter (adapters.py:392-644).  The system is a stub (no
device): what is checked here is the Welford / pairwise-combination / regularisation arithmetic and the metric the
adapter installs."""

import types

import numpy as np
import pytest

from conftest import assert_close, golden_names, load_golden
from mici_amd import adapters
from mici_amd.errors import AdaptationError
from mici_amd.states import ChainState


class StubSystem:
    def __init__(self):
        self.metric = None

    def set_metric(self, metric):
        self.metric = np.array(metric)

    def sample_momentum(self, state, rng):
        z = rng.standard_normal(np.asarray(state.pos).shape)
        return np.sqrt(self.metric) * z if self.metric.ndim == 1 else np.linalg.cholesky(self.metric) @ z


class ReplayRng:
    def __init__(self, z):
        self.z = z

    def standard_normal(self, size=None):
        return self.z.copy()


@pytest.mark.parametrize("name", golden_names("metricadapt"))
def test_metric_adapter_matches_reference_fixture(name):
    g = load_golden(name)
    which, multi = str(g["which"]), bool(g["multi"])
    adapter = (adapters.OnlineVarianceMetricAdapter if which == "variance" else adapters.OnlineCovarianceMetricAdapter)(
        reg_iter_offset=int(g["reg_iter_offset"]), reg_scale=float(g["reg_scale"]))
    n_updates, n_chains, dim = g["pos_seq"].shape
    transition = types.SimpleNamespace(system=StubSystem())
    states = [ChainState(pos=g["pos_seq"][0, c].copy(), mom=np.zeros(dim), dir=1) for c in range(n_chains)]
    adapt_states = [adapter.initialize(states[c], transition) for c in range(n_chains)]
    for k in range(n_updates):
        for c in range(n_chains):
            states[c].pos = g["pos_seq"][k, c].copy()
            adapter.update(adapt_states[c], states[c], {}, transition)
    rngs = [ReplayRng(z) for z in g["z"]]
    if multi:
        adapter.finalize(adapt_states, states, transition, rngs)
    else:
        adapter.finalize(adapt_states[0], states[0], transition, rngs[0])
    metric = transition.system.metric
    ref_metric = g["metric"]
    if which == "variance":
        assert_close(metric, np.diag(ref_metric) if ref_metric.ndim == 2 else ref_metric, 1e-13, "diagonal metric")
        for c in range(len(g["z"])):  # the square root of a diagonal metric is unique: same momenta as the reference
            assert_close(states[c].mom, g["mom"][c], 1e-13, f"mom {c}")
    else:
        assert_close(metric, ref_metric, 1e-11, "dense metric")
        minv = np.linalg.inv(metric)
        for c in range(len(g["z"])):  # any square root S of M gives p = S z with p.M^-1 p = z.z
            assert_close(states[c].mom @ minv @ states[c].mom, g["z"][c] @ g["z"][c], 1e-10, "quadratic form")
            assert_close(g["mom"][c] @ minv @ g["mom"][c], g["z"][c] @ g["z"][c], 1e-10, "reference's square root")


def test_needs_two_samples():
    adapter = adapters.OnlineVarianceMetricAdapter()
    tr = types.SimpleNamespace(system=StubSystem())
    st = ChainState(pos=np.zeros(3), mom=np.zeros(3), dir=1)
    a = adapter.initialize(st, tr)
    adapter.update(a, st, {}, tr)
    with pytest.raises(AdaptationError):
        adapter.finalize(a, st, tr, np.random.default_rng(0))
    assert adapters.OnlineVarianceMetricAdapter.is_fast is False and adapters.DualAveragingStepSizeAdapter.is_fast
