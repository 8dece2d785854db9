"""GPU parity: explicit leapfrog HIP kernels vs the committed reference fixtures and the oracle.

fp64 tolerance: identical operation order up to FMA contraction / MFMA
summation order => <= 1e-13 relative after one step, growing at most linearly with the number of
steps (<= 1e-10 after 100 steps on the linear models)."""

import numpy as np
import pytest

from conftest import assert_close, golden_names, load_golden
from oracle import integrators as orc
from oracle import models as omdl

import mici_amd
from mici_amd import integrators, models, systems
from mici_amd.states import ChainState

pytestmark = pytest.mark.gpu


def system_from_golden(g):
    n, d = g["q0"].shape
    target = models.target_from_id(g["target"], g["target_params"], d)
    mk = int(g["metric_kind"])
    metric = None if mk == models.METRIC_IDENTITY else g["metric"]
    return systems.EuclideanMetricSystem(target, metric=metric)


def oracle_system_from_golden(g):
    n, d = g["q0"].shape
    target = omdl.target_from_id(g["target"], g["target_params"], d)
    mk = int(g["metric_kind"])
    return orc.EuclidSystem(target, mk, None if mk == omdl.METRIC_IDENTITY else g["metric"])


@pytest.mark.parametrize("name", golden_names("euclid"))
def test_leapfrog_matches_reference_fixture(name):
    g = load_golden(name)
    system = system_from_golden(g)
    integ = integrators.LeapfrogIntegrator(system, float(g["step_size"]))
    for k, s in enumerate(int(s) for s in g["checkpoints"]):
        q, p, status, n_done = integ.step_batch(g["q0"], g["p0"], g["dir"], n_steps=s)
        tol = 2e-13 * max(1, s)
        assert_close(q, g["q_out"][k], tol, f"{name} q@{s}")
        assert_close(p, g["p_out"][k], tol, f"{name} p@{s}")
        assert np.all(status == 0) and np.all(n_done == s)
        h = system.h_batch(q, p)
        assert_close(h, g["h_out"][k], 1e-12, f"{name} h@{s}")


@pytest.mark.parametrize("name", golden_names("symcomp"))
def test_symmetric_composition_matches_reference_fixture(name):
    g = load_golden(name)
    system = system_from_golden(g)
    free, h1 = list(g["free_coefficients"]), bool(g["initial_h1_flow_step"])
    integ = integrators.SymmetricCompositionIntegrator(system, free, step_size=float(g["step_size"]),
                                                       initial_h1_flow_step=h1)
    bcss = {1: integrators.BCSSTwoStageIntegrator, 2: integrators.BCSSThreeStageIntegrator,
            3: integrators.BCSSFourStageIntegrator}
    if "bcss" in name:
        named = bcss[len(free)](system, float(g["step_size"]))
        assert named.coefficients == integ.coefficients
    for k, s in enumerate(int(s) for s in g["checkpoints"]):
        q, p, status, n_done = integ.step_batch(g["q0"], g["p0"], g["dir"], n_steps=s)
        tol = 3e-13 * max(1, s)
        assert_close(q, g["q_out"][k], tol, f"{name} q@{s}")
        assert_close(p, g["p_out"][k], tol, f"{name} p@{s}")
        assert np.all(status == 0) and np.all(n_done == s)
        assert_close(system.h_batch(q, p), g["h_out"][k], 1e-12, f"{name} h@{s}")
    # reversibility: integrate back with flipped directions (tests/test_integrators.py:75-91)
    s = int(g["checkpoints"][-1])
    q, p, _, _ = integ.step_batch(g["q0"], g["p0"], g["dir"], n_steps=s)
    qb, pb, _, _ = integ.step_batch(q, p, -g["dir"], n_steps=s)
    assert_close(qb, g["q0"], 1e-9, "reversed q")
    assert_close(pb, g["p0"], 1e-9, "reversed p")
    # the reference's single-state contract
    from mici_amd.states import ChainState
    st = ChainState(pos=g["q0"][0].copy(), mom=g["p0"][0].copy(), dir=int(g["dir"][0]))
    new = integ.step(st)
    assert_close(new.pos, g["q_out"][0][0] if int(g["checkpoints"][0]) == 1 else new.pos, 3e-13, "step pos")
    assert np.array_equal(st.pos, g["q0"][0])


@pytest.mark.parametrize("dim,n,target_kind,metric_kind", [
    (128, 4096, "dense", "identity"),   # BASELINE config c2(iii) with a BCSS integrator: FP64 MFMA path
    (128, 512, "dense", "dense"),
    (100, 300, "diag", "dense"),
    (128, 2048, "iso", "diag"),         # elementwise path
    (37, 100, "poly", "identity"),
])
@pytest.mark.parametrize("stages", [2, 4])
def test_composition_fast_paths_match_oracle(dim, n, target_kind, metric_kind, stages):
    """BCSS compositions on the kernels of the leapfrog (MFMA / elementwise), oracle on a sample of chains."""
    rng = np.random.default_rng(dim + stages)
    if target_kind == "dense":
        P = omdl.make_spd(dim, rng)
        target, otarget = models.GaussDense(P), omdl.GaussDense(P)
    elif target_kind == "diag":
        prec = np.exp(0.2 * rng.standard_normal(dim))
        target, otarget = models.GaussDiag(prec), omdl.GaussDiag(prec)
    elif target_kind == "iso":
        target, otarget = models.GaussIso(dim), omdl.GaussIso(dim)
    else:
        target, otarget = models.Poly(dim, 1.0, 0.25), omdl.Poly(dim, 1.0, 0.25)
    if metric_kind == "identity":
        mk, metric = omdl.METRIC_IDENTITY, None
    elif metric_kind == "diag":
        mk, metric = omdl.METRIC_DIAG, np.exp(0.2 * rng.standard_normal(dim))
    else:
        mk, metric = omdl.METRIC_DENSE, omdl.make_spd(dim, rng)
    system = systems.EuclideanMetricSystem(target, metric=metric)
    osys = orc.EuclidSystem(otarget, mk, metric)
    cls = {2: integrators.BCSSTwoStageIntegrator, 4: integrators.BCSSFourStageIntegrator}[stages]
    h, n_steps = 0.1, 25
    integ = cls(system, h)
    q0 = rng.standard_normal((n, dim))
    p0 = np.stack([osys.msqrt(z) for z in rng.standard_normal((n, dim))])
    dirs = np.where(rng.random(n) < 0.5, 1, -1).astype(np.int8)
    q, p, status, n_done = integ.step_batch(q0, p0, dirs, n_steps=n_steps)
    assert np.all(status == 0) and np.all(n_done == n_steps)
    free = orc.BCSS_FREE_COEFFICIENTS[stages]
    for c in np.concatenate([[0, 1, 15, 16, n - 1], rng.integers(0, n, 5)]):
        qo, po = orc.composition_steps(osys, q0[c], p0[c], dirs[c] * h, n_steps, free)
        assert_close(q[c], qo, 1e-11, f"q chain {c}")
        assert_close(p[c], po, 1e-11, f"p chain {c}")
    qb, pb, _, _ = integ.step_batch(q, p, -dirs, n_steps=n_steps)
    assert_close(qb, q0, 1e-9, "reversed q")


@pytest.mark.parametrize("name", golden_names("impliciteuclid"))
def test_implicit_leapfrog_on_euclidean_system_matches_reference_fixture(name):
    """ImplicitLeapfrogIntegrator accepts plain Euclidean systems as in the reference
    (tests/test_integrators.py:435-462); on the device it is the composition (1, 1, 0, 1, 1)."""
    g = load_golden(name)
    system = system_from_golden(g)
    integ = integrators.ImplicitLeapfrogIntegrator(system, float(g["step_size"]))
    for k, s in enumerate(int(s) for s in g["checkpoints"]):
        q, p, status, n_done = integ.step_batch(g["q0"], g["p0"], g["dir"], n_steps=s)
        assert np.all(status == 0) and np.all(n_done == s)
        assert_close(q, g["q_out"][k], 3e-13 * max(1, s), f"{name} q@{s}")
        assert_close(p, g["p_out"][k], 3e-13 * max(1, s), f"{name} p@{s}")
        assert_close(system.h_batch(q, p), g["h_out"][k], 1e-12, f"{name} h@{s}")
    with pytest.raises(ValueError):
        integrators.ImplicitLeapfrogIntegrator(systems.GaussianEuclideanMetricSystem(models.GaussIso(3)), 0.1)


def test_composition_argument_checks():
    system = systems.EuclideanMetricSystem(models.GaussIso(4))
    with pytest.raises(ValueError):
        integrators.SymmetricCompositionIntegrator(system, [0.1] * 8, step_size=0.1)  # 19 coefficients
    integ = integrators.BCSSTwoStageIntegrator(system)  # step_size None -> AdaptationError like the reference
    from mici_amd.errors import AdaptationError
    from mici_amd.states import ChainState
    with pytest.raises(AdaptationError):
        integ.step(ChainState(pos=np.zeros(4), mom=np.ones(4), dir=1))


@pytest.mark.parametrize("name", ["euclid_c1_iso_d32", "euclid_quartic_dense_d5",
                                  "euclid_dense_d20_ragged", "euclid_banana_d16"])
def test_single_state_step_matches_reference_semantics(name):
    """Integrator.step(state): new object, input untouched (tests/test_integrators.py:110-124),
    dir respected, and step-by-step stepping equals the fused trajectory."""
    g = load_golden(name)
    system = system_from_golden(g)
    integ = integrators.LeapfrogIntegrator(system, float(g["step_size"]))
    k = list(g["checkpoints"]).index(g["checkpoints"].max())
    s_max = int(g["checkpoints"].max())
    for c in range(min(3, g["q0"].shape[0])):
        state = ChainState(pos=g["q0"][c].copy(), mom=g["p0"][c].copy(), dir=int(g["dir"][c]))
        init = state.copy()
        new = integ.step(state)
        assert new is not state
        assert np.array_equal(state.pos, init.pos) and np.array_equal(state.mom, init.mom)
        assert state.dir == init.dir and new.dir == init.dir
        for _ in range(s_max - 1):
            new = integ.step(new)
        assert_close(new.pos, g["q_out"][k, c], 1e-12 * s_max, f"{name} stepwise q")
        assert_close(new.mom, g["p_out"][k, c], 1e-12 * s_max, f"{name} stepwise p")


@pytest.mark.parametrize("dim,n,target_kind,metric_kind", [
    (128, 4096, "dense", "identity"),   # BASELINE config c2 at full size
    (128, 4096, "dense", "dense"),
    (128, 4096, "diag", "diag"),
    (128, 4096, "iso", "identity"),
    (33, 50, "dense", "diag"),          # ragged: D not a multiple of 16, N not a multiple of 16
    (7, 3, "poly", "dense"),
    (1, 5, "dense", "dense"),
    (200, 9, "dense", "dense"),         # D > 128 -> generic wave-per-chain kernel
    (64, 17, "banana", "diag"),
    (5000, 3, "banana", "diag"),        # three D-vectors of a chain fill most of a CU's LDS
])
def test_leapfrog_matches_oracle_and_is_reversible(dim, n, target_kind, metric_kind):
    rng = np.random.default_rng(1234)
    P = omdl.make_spd(dim, rng)
    prec = np.exp(0.2 * rng.standard_normal(dim))
    ot = {"dense": lambda: omdl.GaussDense(P), "diag": lambda: omdl.GaussDiag(prec),
          "iso": lambda: omdl.GaussIso(dim), "poly": lambda: omdl.Poly(dim, 0.5, 0.7),
          "banana": lambda: omdl.Banana(dim)}[target_kind]()
    metric = {"identity": None, "diag": np.exp(0.1 * rng.standard_normal(dim)),
              "dense": omdl.make_spd(dim, rng)}[metric_kind]
    mk = {"identity": 0, "diag": 1, "dense": 2}[metric_kind]
    osys = orc.EuclidSystem(ot, mk, metric)
    system = systems.EuclideanMetricSystem(models.target_from_id(ot.tid, ot.params(), dim),
                                           metric=metric)
    q0 = rng.standard_normal((n, dim))
    p0 = np.stack([osys.msqrt(z) for z in rng.standard_normal((n, dim))])
    dirs = np.where(rng.uniform(size=n) < 0.5, -1, 1).astype(np.int8)
    h, steps = 0.05, 25
    integ = integrators.LeapfrogIntegrator(system, h)
    q, p, status, n_done = integ.step_batch(q0, p0, dirs, n_steps=steps)
    assert np.all(status == 0) and np.all(n_done == steps)
    # oracle on a bounded sample of chains (the oracle is per-chain NumPy)
    sample = np.unique(np.concatenate([np.arange(min(n, 6)), [n - 1], rng.integers(0, n, 6)]))
    for c in sample:
        qo, po = orc.leapfrog_steps(osys, q0[c], p0[c], dirs[c] * h, steps)
        assert_close(q[c], qo, 3e-12, f"q chain {c}")
        assert_close(p[c], po, 3e-12, f"p chain {c}")
    # size-independent properties at full size: reversibility (flip dir, integrate back) and
    # momentum-space sampling consistency
    qb, pb, _, _ = integ.step_batch(q, p, -dirs, n_steps=steps)
    assert_close(qb, q0, 1e-10, "reversed q")
    assert_close(pb, p0, 1e-10, "reversed p")
    # Hamiltonian from the device equals the oracle's on the sample
    hd = system.h_batch(q, p)
    for c in sample:
        assert_close(hd[c], osys.h(q[c], p[c]), 1e-11, f"h chain {c}")


def test_empty_and_zero_step_batches():
    system = systems.EuclideanMetricSystem(models.GaussIso(8))
    integ = integrators.LeapfrogIntegrator(system, 0.1)
    q, p, st, nd = integ.step_batch(np.zeros((0, 8)), np.zeros((0, 8)), 1, n_steps=3)
    assert q.shape == (0, 8) and st.shape == (0,)
    rng = np.random.default_rng(0)
    q0, p0 = rng.standard_normal((2, 5, 8))
    q, p, st, nd = integ.step_batch(q0, p0, 1, n_steps=0)
    assert np.array_equal(q, q0) and np.array_equal(p, p0)


def test_sample_momentum_and_dh_dmom():
    rng = np.random.default_rng(5)
    dim, n = 20, 7
    M = omdl.make_spd(dim, rng)
    for metric, mk in [(None, 0), (np.exp(rng.standard_normal(dim)), 1), (M, 2)]:
        osys = orc.EuclidSystem(omdl.GaussIso(dim), mk, metric)
        system = systems.EuclideanMetricSystem(models.GaussIso(dim), metric=metric)
        q = rng.standard_normal((n, dim))
        z = rng.standard_normal((n, dim))
        mom = system.sample_momentum_batch(q, z)
        for c in range(n):
            assert_close(mom[c], osys.msqrt(z[c]), 1e-13, "sample_momentum")
        v = system.dh_dmom_batch(q, mom)
        for c in range(n):
            assert_close(v[c], osys.minv(mom[c]), 1e-12, "dh_dmom")


def test_nonfinite_state_propagates_like_reference():
    # the explicit integrator never raises in the reference: NaN/inf simply propagate
    system = systems.EuclideanMetricSystem(models.GaussDense(np.eye(16)))
    integ = integrators.LeapfrogIntegrator(system, 0.1)
    q0 = np.zeros((3, 16)); p0 = np.ones((3, 16))
    q0[1, 2] = np.nan
    q, p, st, nd = integ.step_batch(q0, p0, 1, n_steps=2)
    assert np.all(np.isfinite(q[0])) and np.all(np.isfinite(q[2]))
    assert np.isnan(q[1]).any()


@pytest.mark.parametrize("dim,metric_kind", [(129, "dense"), (300, "diag"), (1024, "identity")])
def test_implicit_midpoint_on_euclidean_systems_beyond_128_dimensions(dim, metric_kind):
    """ImplicitMidpointIntegrator (integrators.py:547-681) on Euclidean systems: rounds 1-4 stopped at D = 128 (the check's
    reference state lived in two registers a lane); round 5 keeps it in LDS: D <= 1024."""
    from mici_amd.errors import DeviceError
    rng = np.random.default_rng(4000 + dim)
    n, h, steps = 5, 0.05, 3
    metric = {"identity": None, "diag": np.exp(0.2 * rng.standard_normal(dim)), "dense": omdl.make_spd(dim, rng)}[metric_kind]
    mk = {"identity": omdl.METRIC_IDENTITY, "diag": omdl.METRIC_DIAG, "dense": omdl.METRIC_DENSE}[metric_kind]
    system = systems.EuclideanMetricSystem(models.Poly(dim, 1.0, 0.3), metric=metric)
    osys = orc.EuclidSystem(omdl.Poly(dim, 1.0, 0.3), mk, metric)
    q0 = 0.5 * rng.standard_normal((n, dim))
    p0 = np.stack([osys.msqrt(z) for z in rng.standard_normal((n, dim))])
    dirs = np.where(np.arange(n) % 2 == 0, 1, -1).astype(np.int8)
    integ = integrators.ImplicitMidpointIntegrator(system, h)
    q, p, st, nd = integ.step_batch(q0, p0, dirs, n_steps=steps)
    for c in range(n):
        qo, po, so, no = orc.implicit_midpoint_steps(osys, q0[c], p0[c], dirs[c] * h, steps)
        assert so == st[c] and no == nd[c]
        assert_close(q[c], qo, 1e-10, f"q chain {c}")
        assert_close(p[c], po, 1e-10, f"p chain {c}")
    assert np.all(st == 0)
    if dim == 1024:
        big = systems.EuclideanMetricSystem(models.Poly(1025, 1.0, 0.3))
        with pytest.raises(DeviceError):
            integrators.ImplicitMidpointIntegrator(big, h).step_batch(np.zeros((1, 1025)), np.ones((1, 1025)), 1, 1)
