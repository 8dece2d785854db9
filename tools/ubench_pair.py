"""Developer profile: phase clocks (implicit_core.h PH_*) of whole implicit-leapfrog steps on the c3 workload (h = 0.02,
20 steps, 1024 chains, D = 64, rank-one metric), the one-wave kernel (implicit_mfma.h) beside the two-wave kernel
(implicit_pair.h).  Needs libmici_amd_dev.so."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from mici_amd import _ffi, integrators, models, systems  # noqa: E402
from mici_amd.runtime import Context, DeviceBatch  # noqa: E402
from oracle import models as omdl  # noqa: E402

ctx = Context(dev=True)
lib = ctx._lib
rng = np.random.default_rng(0)
dim = 64
system = systems.DenseRiemannianMetricSystem(models.Banana(dim), models.Rank1Metric(omdl.make_spd(dim, rng)))
model = system.device_model(ctx)
integ = integrators.ImplicitLeapfrogIntegrator(system, 0.02)
n, nsteps = 1024, 20
q0 = rng.standard_normal((n, dim))
p0 = system.sample_momentum_batch(q0, rng.standard_normal((n, dim)))
labels = ["other", "grad", "full sweep", "trailing sweep", "M(x) v", "M0^-1 r", "reductions", "momentum solves"]
for name in ("mm_debug_mfma_step_profile", "mm_debug_pair_step_profile"):
    fn = getattr(lib, name)
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.POINTER(_ffi.FpOpts), _ffi.c_double_p]
    batch = DeviceBatch(ctx, n, dim)
    batch.upload(q0, p0, np.ones(n, dtype=np.int8))
    ph = np.zeros((n, 8))
    opts = integ._opts()
    _ffi.check(fn(ctx.handle, model.handle, batch.handle, 0.02, nsteps, C.byref(opts), ph.ctypes.data_as(_ffi.c_double_p)),
               ctx.handle, name)
    tot = ph.sum(1).mean()
    print(f"{name}: cycle-counter ticks per leapfrog step, mean over {n} chains, {nsteps} steps: total {tot / nsteps:.0f}")
    for k, lab in enumerate(labels):
        print(f"  {lab:16s} {ph[:, k].mean() / nsteps:10.0f}  {100 * ph[:, k].mean() / tot:5.1f} %")
    batch.close()
