"""Edge cases of the lean FP64 arithmetic.

`rcp_nr`, `fdiv` and `sqrt_rsqrt` (mici_amd/csrc/mm_device.h) replace IEEE division and square root on the critical paths
of every kernel.  What the kernels rely on, checked here on the device against the compiler's IEEE expansions evaluated in
the same launch and against NumPy:
  * normal, non-extreme operands: within 1 ulp (2 ulp for the quotient: one more rounding);
  * zero, infinite and NaN operands: "not finite" exactly where IEEE is not finite (possibly a NaN where IEEE gives inf) -
    the divergence tests of the solvers are `err > div_tol or isnan(err)` (solvers.py:80-84, 446-469), and a pivot test
    `!(d > 0)`: both treat inf and NaN alike;
  * sqrt_rsqrt(0) = (0, huge finite), sqrt_rsqrt(inf) = NaN for both (mm_device.h);
  * huge / tiny operands (1e+-300, subnormals): where the result leaves the normal range the lean form may lose it
    (flush to 0 / inf / NaN) - the test pins WHERE, so that a kernel author knows the range the forms are good for:
    |x| in [1e-290, 1e290] for the reciprocal, quotients whose operands and result stay inside [1e-290, 1e290].
Then the same question end to end: states scaled by 1e+-150 through the integrators - the status codes (diverged /
max-iters / LinAlg / non-reversible) must be the oracle's (tests below)."""

import ctypes as C

import numpy as np
import pytest

from mici_amd import _ffi
from mici_amd.runtime import Context

pytestmark = pytest.mark.gpu

_DEV = []


def _lean(a, b):
    if not _DEV:
        _DEV.append(Context(dev=True))
    ctx = _DEV[0]
    fn = ctx._lib.mm_debug_lean_math
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, _ffi.c_double_p, _ffi.c_double_p, _ffi.c_double_p, C.c_int]
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(np.broadcast_to(b, a.shape), dtype=np.float64)
    out = np.empty((8, a.size))
    _ffi.check(fn(ctx.handle, a.ctypes.data_as(_ffi.c_double_p), b.ctypes.data_as(_ffi.c_double_p),
                  out.ctypes.data_as(_ffi.c_double_p), a.size), ctx.handle, "mm_debug_lean_math")
    return dict(rcp=out[0], div=out[1], sqrt=out[2], rsqrt=out[3], ieee_rcp=out[4], ieee_div=out[5], ieee_sqrt=out[6],
                ieee_rsqrt=out[7])


def _ulps(x, ref):
    return np.abs(x - ref) / np.spacing(np.abs(ref))


def test_normal_range_is_within_an_ulp():
    rng = np.random.default_rng(0)
    a = np.concatenate([np.exp(rng.uniform(-600, 600, 20000)) * rng.choice([-1.0, 1.0], 20000),
                        rng.standard_normal(20000), [1.0, -1.0, 2.0, 0.5, 3.0, 1e-290, 1e290, -1e-290, -1e290]])
    b = np.concatenate([np.exp(rng.uniform(-60, 60, 20000)) * rng.choice([-1.0, 1.0], 20000), rng.standard_normal(20000),
                        [3.0, 7.0, -1.0, 1e-30, 1e30, 1.0, 1.0, 1.0, 1.0]])
    r = _lean(a, b)
    with np.errstate(all="ignore"):
        assert _ulps(r["rcp"], 1.0 / a).max() <= 1.0
        q = a / b
        inside = (np.abs(q) > 1e-290) & (np.abs(q) < 1e290)
        assert inside.sum() > 30000 and _ulps(r["div"][inside], q[inside]).max() <= 2.0
    pos = np.abs(a)
    r = _lean(pos, 1.0)
    assert _ulps(r["sqrt"], np.sqrt(pos)).max() <= 1.0
    assert _ulps(r["rsqrt"], 1.0 / np.sqrt(pos)).max() <= 2.0
    # the device's own IEEE expansions agree with NumPy (they are what the lean forms replaced)
    assert np.array_equal(r["ieee_sqrt"], np.sqrt(pos)) and np.array_equal(r["ieee_rcp"], 1.0 / pos)


def test_special_operands_are_not_finite_where_ieee_is_not_finite():
    a = np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1.0, 0.0, np.inf, np.nan, 1.0, np.inf, 0.0])
    b = np.array([1.0, 1.0, 1.0, 1.0, 1.0, 0.0, 0.0, np.inf, 1.0, np.nan, 0.0, np.inf])
    r = _lean(a, b)
    with np.errstate(all="ignore"):
        ieee_rcp, ieee_div = 1.0 / a, a / b
    # reciprocal: 1/0 = inf, 1/inf = 0, 1/NaN = NaN; the lean form may give NaN for the first two - never a finite
    # non-zero number, never finite where IEEE is not finite
    for lean, ieee in ((r["rcp"], ieee_rcp), (r["div"], ieee_div)):
        for x, y in zip(lean, ieee):
            if np.isnan(y) or np.isinf(y):
                assert not np.isfinite(x), (x, y)
            elif y == 0.0:
                assert x == 0.0 or np.isnan(x), (x, y)
            else:
                assert x == y, (x, y)
    # the kernels' tests on such values: `err > tol || err != err` and `!(pivot > 0)` cannot tell inf from NaN
    # (errors and pivots are magnitudes: a norm is never negative, so -inf is compared as |.|)
    for x, y in zip(np.abs(r["div"]), np.abs(ieee_div)):
        assert (x > 1e10 or x != x) == (y > 1e10 or y != y) or y == 0.0, (x, y)
    assert np.array_equal(np.isnan(r["ieee_rcp"]), np.isnan(ieee_rcp))


def test_square_root_of_zero_infinity_nan():
    r = _lean(np.array([0.0, np.inf, np.nan, 4.0, 1e-320, 5e-324, 1e308]), 1.0)
    assert r["sqrt"][0] == 0.0 and np.isfinite(r["rsqrt"][0]) and r["rsqrt"][0] > 1e150  # finite at 0
    assert np.isnan(r["sqrt"][1]) and np.isnan(r["rsqrt"][1])                            # x = inf: NaN for both
    assert np.isnan(r["sqrt"][2]) and np.isnan(r["rsqrt"][2])
    assert r["sqrt"][3] == 2.0 and r["rsqrt"][3] == 0.5
    # subnormal arguments: the result is a normal number (sqrt(1e-320) = 1e-160) and must still be accurate to a few ulps
    # or be flagged here - sums of squares of quantities below 1e-160 do not occur in the kernels' norms (they compare
    # against tolerances >= 1e-12) but the Gram value of one constraint can be arbitrarily small
    for k in (4, 5):
        ref = np.sqrt(np.array([1e-320, 5e-324])[k - 4])
        assert r["sqrt"][k] == 0.0 or abs(r["sqrt"][k] - ref) <= 1e-3 * ref, (k, r["sqrt"][k], ref)
    assert _ulps(r["sqrt"][6:7], np.sqrt(1e308)).max() <= 1.0


def test_extreme_scales_where_the_lean_forms_stop_being_exact():
    """|x| beyond 1e+-290: v_rcp_f64's Newton correction multiplies x by an estimate of 1 / x - exact in range - but for
    a SUBNORMAL result (|x| > 4.5e307) the hardware estimate is flushed to zero and the refinement cannot recover it: the
    lean reciprocal is 0 where IEEE gives a subnormal.  Both are "below any tolerance" for every caller; pinned here."""
    a = np.array([1e300, -1e300, 1e-300, -1e-300, 1e307, 1e-307, 8e307, 2e-308, 1e-310])
    r = _lean(a, 1.0)
    with np.errstate(all="ignore"):
        ref = 1.0 / a
    for x, y, v in zip(r["rcp"], ref, a):
        if abs(y) >= 2.3e-308 and np.isfinite(y):  # result normal: exact to an ulp
            assert abs(x - y) <= np.spacing(abs(y)), (v, x, y)
        elif np.isfinite(y):                       # result subnormal: the subnormal, or flushed to zero
            assert x == 0.0 or abs(x - y) <= 1e-3 * abs(y) + 5e-324, (v, x, y)
        else:                                      # 1 / subnormal overflows in IEEE
            assert not np.isfinite(x), (v, x, y)
    # quotients of huge by tiny and tiny by huge: overflow to "not finite", underflow to (sub)zero - as IEEE up to the
    # inf / NaN and subnormal / zero ambiguities above
    r = _lean(np.array([1e200, 1e-200, 1e200, -1e-200]), np.array([1e-200, 1e200, -1e-200, 1e200]))
    assert not np.isfinite(r["div"][0]) and not np.isfinite(r["div"][2])
    assert abs(r["div"][1]) < 1e-300 or np.isnan(r["div"][1])
    assert abs(r["div"][3]) < 1e-300 or np.isnan(r["div"][3])
