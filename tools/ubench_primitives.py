"""Developer micro-benchmark: cycles per primitive of the wave-per-chain Riemannian kernel
(build+sweep inverse, build+LDL^T solve, mat-vec, build only), one wave per chain."""
import ctypes as C
import sys

import numpy as np

sys.path.insert(0, ".")
from mici_amd import models, systems  # noqa: E402
from mici_amd.runtime import Context, DeviceBatch  # noqa: E402
from oracle import models as omdl  # noqa: E402

ctx = Context(dev=True)
lib = ctx._lib
lib.mm_debug_primitive_bench.restype = C.c_int
lib.mm_debug_primitive_bench.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                         C.POINTER(C.c_double)]
rng = np.random.default_rng(0)
dim = 64
system = systems.DenseRiemannianMetricSystem(models.Banana(dim), models.Rank1Metric(omdl.make_spd(dim, rng)))
model = system.device_model(ctx)
for n in (1024, 2048, 4096):
    batch = DeviceBatch(ctx, n, dim)
    batch.upload(rng.standard_normal((n, dim)), rng.standard_normal((n, dim)), 1)
    for variant, name in ((0, "build + sweep inverse"), (1, "build + LDL^T solve"), (2, "mat-vec"),
                          (3, "build only")):
        reps = 200 if variant == 2 else 20
        ms = C.c_double(0)
        rc = lib.mm_debug_primitive_bench(ctx.handle, model.handle, batch.handle, variant, reps, C.byref(ms))
        assert rc == 0, rc
        us = ms.value * 1e3 / reps
        print(f"N={n:5d} {name:24s}: {us:8.2f} us per call per wave-slot  (~{us * 2250:9.0f} cycles @2.25GHz)")
    batch.close()

# matrix-core backend (32 < D <= 64): in-kernel cycle counters per primitive
lib.mm_debug_mfma_profile.restype = C.c_int
lib.mm_debug_mfma_profile.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_double)]
for n in (256, 1024, 4096):
    batch = DeviceBatch(ctx, n, dim)
    batch.upload(rng.standard_normal((n, dim)), rng.standard_normal((n, dim)), 1)
    out = (C.c_double * 8)()
    rc = lib.mm_debug_mfma_profile(ctx.handle, model.handle, batch.handle, 50, out)
    assert rc == 0, rc
    print(f"mfma backend N={n:5d}: build {out[0]:7.0f}  sweep {out[1]:7.0f}  matvec {out[2]:7.0f}  "
          f"grad {out[3]:7.0f}  norm {out[4]:7.0f}  M(x)v {out[5]:7.0f}  cycles")
    batch.close()

# phase clocks of whole steps (implicit_core.h PH_*), c3 workload: h = 0.02, 20 steps, 1024 chains
from mici_amd import _ffi, integrators  # noqa: E402
prof_fn = lib.mm_debug_mfma_step_profile
prof_fn.restype = C.c_int
prof_fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.POINTER(_ffi.FpOpts), _ffi.c_double_p]
integ = integrators.ImplicitLeapfrogIntegrator(system, 0.02)
n, nsteps = 1024, 20
q0 = rng.standard_normal((n, dim))
p0 = system.sample_momentum_batch(q0, rng.standard_normal((n, dim)))
batch = DeviceBatch(ctx, n, dim)
batch.upload(q0, p0, np.ones(n, dtype=np.int8))
ph = np.zeros((n, 8))
opts = integ._opts()
_ffi.check(prof_fn(ctx.handle, model.handle, batch.handle, 0.02, nsteps, C.byref(opts),
                   ph.ctypes.data_as(_ffi.c_double_p)), ctx.handle, "mm_debug_mfma_step_profile")
labels = ["other", "grad", "full sweep", "trailing sweep", "M(x) v", "M0^-1 r", "reductions", "momentum solves"]
tot = ph.sum(1).mean()
print(f"step profile (cycle-counter ticks per leapfrog step, mean over {n} chains, {nsteps} steps): total {tot / nsteps:.0f}")
for k, lab in enumerate(labels):
    print(f"  {lab:16s} {ph[:, k].mean() / nsteps:10.0f}  {100 * ph[:, k].mean() / tot:5.1f} %")

