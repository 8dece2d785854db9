"""GPU: the linear algebra of the block-16 matrix-core team kernel (k_implicit_blk16.hip, 75 < D <= 256: BASELINE
config c4) on its own - explicit inverse by the full sweep, M^-1 b by the trailing sweep (blocked LDL^T) + forward /
backward substitution, M^-1 b by the full sweep + mat-vec - against numpy.linalg on the same metric
M(x) = B + x x^T / D (oracle/models.py Rank1Metric) and on the diagonal metric 1 + x^2.  The step-level parity of
the kernel (status, n_done, fixed-point counts, states vs the oracle and the reference fixtures) is in
test_gpu_implicit.py / test_gpu_full_shards.py; this file localises a defect to a phase."""

import ctypes as C

import numpy as np
import pytest

from conftest import assert_close
from oracle import models as omdl

from mici_amd import _ffi, models, systems
from mici_amd.runtime import Context, DeviceBatch

pytestmark = pytest.mark.gpu


KERNELS = ["blk16"]
_DEV_CTX = []


def dev_context():
    """A context on the developer build of the library (libmici_amd_dev.so: the product plus the test hooks)."""
    if not _DEV_CTX:
        _DEV_CTX.append(Context(dev=True))
    return _DEV_CTX[0]



def _linalg(system, x, b, op, kernel="blk16"):
    ctx = dev_context()
    lib = ctx._lib
    fn = lib.mm_debug_blk16_linalg
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, _ffi.c_double_p, _ffi.c_int32_p, C.c_int,
                   _ffi.c_double_p]
    n, d = x.shape
    batch = DeviceBatch(ctx, n, d)
    batch.upload(x, b, np.ones(n, dtype=np.int8))
    out = np.zeros((n, 256, 256) if op == 0 else (n, 256))
    status = np.zeros(n, dtype=np.int32)
    _ffi.check(fn(ctx.handle, system.device_model(ctx).handle, batch.handle, op, out.ctypes.data_as(_ffi.c_double_p),
                  status.ctypes.data_as(_ffi.c_int32_p), 0, None), ctx.handle, "mm_debug_blk16_linalg")
    batch.close()
    return out, status


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("dim", [256, 255, 200, 129, 77])
def test_rank1_metric_inverse_and_solves(dim, kernel):
    rng = np.random.default_rng(dim)
    n = 5
    om = omdl.Rank1Metric(omdl.make_spd(dim, rng))
    system = systems.DenseRiemannianMetricSystem(models.Banana(dim), models.Rank1Metric(om.base))
    x = rng.standard_normal((n, dim))
    b = rng.standard_normal((n, dim))
    inv, st = _linalg(system, x, b, 0, kernel)
    assert np.all(st == 0)
    for c in range(n):
        M = om.metric_func(x[c])
        want = np.linalg.inv(M)
        assert_close(inv[c, :dim, :dim], want, 1e-11, f"explicit inverse, chain {c}")
    for op, what in ((1, "LDL^T solve"), (2, "inverse mat-vec")):
        u, st = _linalg(system, x, b, op, kernel)
        assert np.all(st == 0)
        for c in range(n):
            want = np.linalg.solve(om.metric_func(x[c]), b[c])
            assert_close(u[c, :dim], want, 1e-11, f"{what}, chain {c}")
            assert np.all(u[c, dim:] == 0.0)


@pytest.mark.parametrize("kernel", KERNELS)
def test_diagquad_metric_and_failure_flags(kernel):
    dim, n = 100, 4
    rng = np.random.default_rng(3)
    system = systems.DenseRiemannianMetricSystem(models.Poly(dim, 1.0, 1.0 / 3.0), models.DiagQuadMetric(dim))
    x = rng.standard_normal((n, dim))
    b = rng.standard_normal((n, dim))
    x[2, 7] = np.inf   # "Array is not finite."   (matrices.py:211-215)
    x[3, 0] = np.nan
    for op in (1, 2):
        u, st = _linalg(system, x, b, op, kernel)
        assert st.tolist() == [0, 0, 5, 5]
        for c in range(2):
            assert_close(u[c, :dim], b[c] / (1.0 + x[c] ** 2), 1e-12, f"op {op} chain {c}")


@pytest.mark.parametrize("kernel", KERNELS)
def test_not_positive_definite_is_flagged(kernel):
    dim, n = 96, 3
    rng = np.random.default_rng(4)
    base = omdl.make_spd(dim, rng)
    base[40, 40] = -5.0  # indefinite: "Cholesky factorisation failed." (matrices.py:1170-1172)
    system = systems.DenseRiemannianMetricSystem(models.Banana(dim), models.Rank1Metric(base))
    x = 0.1 * rng.standard_normal((n, dim))
    for op in (0, 1):
        _, st = _linalg(system, x, x, op, kernel)
        assert np.all(st == 5)


def test_permlane_swap_semantics():
    """v_permlane16_swap / v_permlane32_swap (gfx950) give the xor-16 / xor-32 butterfly sums the kernel's
    column reductions assume - checked on the device against ds_bpermute."""
    dim = 96
    system = systems.DenseRiemannianMetricSystem(models.Banana(dim), models.Rank1Metric(np.eye(dim)))
    x = np.zeros((1, dim))
    out, _ = _linalg(system, x, x, 13)
    assert np.all(out[0, :64] == 0.0), out[0, :64]
