"""CPU suite: the C-ABI library loads and exports every symbol include/mici_amd.h declares; host
logic (argument checks, error classes, state container) behaves like the reference's."""

import copy
import os
import pickle
import re

import numpy as np
import pytest

import mici_amd
from mici_amd import _ffi, errors, integrators, models, solvers, systems
from mici_amd.states import ChainState

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mici_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _ffi.load()
    names = declared_symbols()
    assert len(names) >= 25
    for name in names:
        assert hasattr(lib, name), f"libmici_amd.so does not export {name}"
        assert name in _ffi.SIGNATURES, f"{name} has no ctypes signature in _ffi.py"
    assert sorted(_ffi.SIGNATURES) == names
    assert lib.mm_abi_version() == _ffi.ABI_VERSION


def _exported(path):
    """EVERY defined dynamic symbol of the library (functions, objects, weak ones, C++ internals alike) except the
    linker's own section markers."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    names = {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    return sorted(n for n in names if n not in ("_init", "_fini", "_edata", "_end", "__bss_start"))


def test_product_library_exports_exactly_the_header():
    """the dynamic symbol table of libmici_amd.so - all of it, not only the ` T mm_` lines -
    equals include/mici_amd.h (the library is built with -fvisibility=hidden and the header opens a default-visibility
    region); the developer entry points (mm_debug_*) live in libmici_amd_dev.so only, which is the product plus those."""
    names = declared_symbols()
    assert _exported(_ffi.lib_path()) == names
    dev = _exported(_ffi.lib_path(dev=True))
    assert set(names) <= set(dev)
    extra = sorted(set(dev) - set(names))
    assert extra and all(n.startswith("mm_debug_") for n in extra), extra
    _ffi.load(dev=True)


def test_struct_sizes_match_header():
    # layouts as the C compiler sees them (x86-64 SysV)
    import ctypes as C
    assert C.sizeof(_ffi.FpOpts) == 40
    assert C.sizeof(_ffi.ProjOpts) == 56
    assert C.sizeof(_ffi.Counters) == 112
    assert C.sizeof(_ffi.ModelDesc) == 96


def test_error_hierarchy_matches_reference():
    # reference errors.py:6-35
    assert issubclass(errors.ConvergenceError, errors.IntegratorError)
    assert issubclass(errors.NonReversibleStepError, errors.IntegratorError)
    assert issubclass(errors.HamiltonianDivergenceError, errors.IntegratorError)
    assert not issubclass(errors.LinAlgError, errors.IntegratorError)
    assert issubclass(errors.IntegratorError, errors.Error)
    assert issubclass(errors.Error, RuntimeError)
    for code, exc in [(1, errors.ConvergenceError), (2, errors.ConvergenceError),
                      (3, errors.ConvergenceError), (4, errors.NonReversibleStepError),
                      (5, errors.LinAlgError)]:
        with pytest.raises(exc):
            errors.raise_for_status(code)
    errors.raise_for_status(0)


def test_chain_state_semantics():
    # reference tests/test_states.py:130-353 (the parts that do not concern the memo cache)
    s = ChainState(pos=np.arange(3.0), mom=np.ones(3), dir=1)
    c = s.copy()
    c.pos[0] = 10.0
    c.dir = -1
    assert s.pos[0] == 0.0 and s.dir == 1
    assert "pos" in s and "foo" not in s
    with pytest.raises(AttributeError):
        s.foo
    r = s.copy(read_only=True)
    with pytest.raises(errors.ReadOnlyStateError):
        r.pos = np.zeros(3)
    t = pickle.loads(pickle.dumps(s))
    assert np.array_equal(t.pos, s.pos) and t.dir == 1


def test_step_size_none_raises_adaptation_error():
    system = systems.EuclideanMetricSystem(models.GaussIso(4))
    integ = integrators.LeapfrogIntegrator(system, None)
    with pytest.raises(errors.AdaptationError):  # integrators.py:72-77
        integ.step(ChainState(pos=np.zeros(4), mom=np.zeros(4), dir=1))


def test_constructor_validation():
    with pytest.raises(TypeError):
        systems.EuclideanMetricSystem(lambda q: 0.0)
    with pytest.raises(ValueError):
        systems.EuclideanMetricSystem(models.GaussIso(3), metric=np.zeros((3, 3, 3)))
    with pytest.raises(ValueError):
        systems.SoftAbsRiemannianMetricSystem(models.Funnel(np.ones(3)), softabs_coeff=0.0)
    with pytest.raises(ValueError):
        integrators.ImplicitLeapfrogIntegrator(systems.GaussianEuclideanMetricSystem(models.GaussIso(3)), 0.1)
    # plain Euclidean systems are accepted, as in the reference's own tests (tests/test_integrators.py:435-462)
    integrators.ImplicitLeapfrogIntegrator(systems.EuclideanMetricSystem(models.GaussIso(3)), 0.1)
    with pytest.raises(ValueError):
        integrators.ImplicitLeapfrogIntegrator(
            systems.DenseConstrainedEuclideanMetricSystem(models.GaussIso(3), models.FirstCoordConstr()), 0.1)
    with pytest.raises(TypeError):
        solvers.solve_fixed_point_direct(np.cos, np.ones(1))
    assert solvers.norm_code(solvers.maximum_norm) == 0
    assert solvers.norm_code(solvers.euclidean_norm) == 1


def test_integrator_deepcopy_and_pickle_drop_device_handles():
    system = systems.EuclideanMetricSystem(models.GaussDiag(np.ones(4)), metric=np.ones(4))
    integ = integrators.LeapfrogIntegrator(system, 0.1)
    clone = copy.deepcopy(integ)
    clone.step_size = 0.2
    assert integ.step_size == 0.1 and clone.system is integ.system
    again = pickle.loads(pickle.dumps(integ))
    assert again.step_size == 0.1 and again.system.metric_kind == models.METRIC_DIAG


def test_no_gpu_means_loud_failure():
    import ctypes as C
    lib = _ffi.load()
    n = C.c_int(0)
    rc = lib.mm_device_count(C.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is visible")
    system = systems.EuclideanMetricSystem(models.GaussIso(4))
    integ = integrators.LeapfrogIntegrator(system, 0.1)
    with pytest.raises(errors.DeviceError):
        integ.step(ChainState(pos=np.zeros(4), mom=np.zeros(4), dir=1))


def test_transition_deepcopy_and_pickle_drop_device_batches():
    """The reference's sampler deep-copies transitions per chain / stage and pickles them for process pools
    (samplers.py:1124-1129); cached device batches (ctypes handles) must not travel with them."""
    import ctypes as C

    from mici_amd import transitions

    system = systems.EuclideanMetricSystem(models.GaussIso(4))
    integ = integrators.LeapfrogIntegrator(system, 0.1)
    for tr in (transitions.MetropolisStaticIntegrationTransition(system, integ, 3),
               transitions.MetropolisRandomIntegrationTransition(system, integ, (1, 4))):
        # what a transition looks like after it has sampled once: cached handles that cannot be pickled
        class FakeCtx:
            def __init__(self):
                self._caches = set()

        class FakeBatch:
            handle = C.c_void_p(1234)
            closed = False

            def close(self, force=False):
                self.closed = True

        fake_ctx = FakeCtx()
        fb = FakeBatch()
        fb.ctx = fake_ctx
        tr._proposals.put(fake_ctx, fb, 1, 4)
        tr._one = FakeBatch()
        clone = copy.deepcopy(tr)
        assert len(clone._proposals) == 0 and not hasattr(clone, "_one")
        assert clone.integrator is not tr.integrator and clone.integrator.step_size == 0.1
        again = pickle.loads(pickle.dumps(tr))
        assert len(again._proposals) == 0 and not hasattr(again, "_one")
        assert tr._one is not None  # the original keeps its cache
        assert tr._proposals.get(fake_ctx, 1, 4) is fb
        # an entry goes when its context is closed (Context.close evicts from every cache it is in)
        assert tr._proposals in fake_ctx._caches
        tr._proposals.evict(fake_ctx)
        assert fb.closed and len(tr._proposals) == 0
