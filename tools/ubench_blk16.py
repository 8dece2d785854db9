#!/usr/bin/env python3
"""Phase timings of the block-16 team kernel (k_implicit_blk16.hip) on the GPU box: one chain per CU (256 chains),
each phase repeated `reps` times inside one launch -> microseconds and cycles (at 2.4 GHz) per phase per chain."""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mici_amd import _ffi, models, systems  # noqa: E402
from mici_amd.runtime import Context, DeviceBatch  # noqa: E402

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
reps = 20
rng = np.random.default_rng(0)
a = rng.standard_normal((dim, dim))
base = a @ a.T / dim + np.eye(dim)
system = systems.DenseRiemannianMetricSystem(models.Banana(dim), models.Rank1Metric(base))
ctx = Context(dev=True)
fn = ctx._lib.mm_debug_blk16_linalg
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, _ffi.c_double_p, _ffi.c_int32_p, C.c_int, _ffi.c_double_p]
batch = DeviceBatch(ctx, n, dim)
batch.upload(rng.standard_normal((n, dim)), rng.standard_normal((n, dim)), np.ones(n, dtype=np.int8))
out = np.zeros((n, 256))
res = {}
names = {3: "build", 4: "build+full_sweep", 5: "build+trailing_sweep", 6: "matvec", 7: "substitution",
         15: "metric_apply", 16: "sum2"}
for op, name in names.items():
    t = {}
    for r in (0, reps):
        best = 1e9
        for _ in range(3):
            ms = C.c_double(0.0)
            _ffi.check(fn(ctx.handle, system.device_model(ctx).handle, batch.handle, op, out.ctypes.data_as(_ffi.c_double_p),
                          None, r, C.byref(ms)), ctx.handle, "mm_debug_blk16_linalg")
            best = min(best, ms.value)
        t[r] = best
    us = (t[reps] - t[0]) / reps * 1e3
    res[name] = dict(us=us, kcycles=us * 2.4)
    print(f"{name:22s} {us:8.2f} us  {us * 2.4:8.1f} kcycles", flush=True)
for op, name in ((8, "full sweep"), (9, "trailing sweep"), (10, "full sweep WITHOUT pivot-block inverse"),
                 (11, "full sweep WITHOUT tile updates"), (12, "full sweep, tile updates WITHOUT operand loads"),
                 (14, "SIMD isolation: waves 0/4 update nothing, wave 0 repeats the pivot inverse during the updates "
                      "(its time in the -W column)")):
    ms = C.c_double(0.0)
    _ffi.check(fn(ctx.handle, system.device_model(ctx).handle, batch.handle, op, out.ctypes.data_as(_ffi.c_double_p),
                  None, 0, C.byref(ms)), ctx.handle, "mm_debug_blk16_linalg")
    prof = out[:, :64].reshape(n, 8, 8).mean(0)  # [wave][phase]
    nb = prof[0, 5]
    print(f"{name}: cycle-counter ticks per block (mean over chains; {nb:.0f} blocks), per wave:")
    print("  wave   publish  barrier  pivot-inv   -W     updates   | sweep total")
    for w in range(8):
        print(f"  {w}    " + "  ".join(f"{prof[w, k] / nb:8.0f}" for k in range(5)) + f"   | {prof[w, 6]:10.0f}")
    res[name + " profile"] = prof.tolist()
# phase clocks of whole steps (implicit_core.h PH_*), c4 workload: h = 0.01, 10 steps
prof_fn = ctx._lib.mm_debug_blk16_step_profile
prof_fn.restype = C.c_int
prof_fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.POINTER(_ffi.FpOpts), _ffi.c_double_p]
from mici_amd import integrators  # noqa: E402
integ = integrators.ImplicitLeapfrogIntegrator(system, 0.01)
q0 = rng.standard_normal((n, dim))
p0 = system.sample_momentum_batch(q0, rng.standard_normal((n, dim)))
batch.upload(q0, p0, np.ones(n, dtype=np.int8))
nsteps = 10
ph = np.zeros((n, 8))
opts = integ._opts()
_ffi.check(prof_fn(ctx.handle, system.device_model(ctx).handle, batch.handle, 0.01, nsteps, C.byref(opts),
                   ph.ctypes.data_as(_ffi.c_double_p)), ctx.handle, "mm_debug_blk16_step_profile")
labels = ["other", "grad", "full sweep", "trailing sweep", "M(x) v", "M0^-1 r", "reductions", "momentum solves"]
tot = ph.sum(1).mean()
print(f"step profile (cycle-counter ticks per leapfrog step, mean over {n} chains, {nsteps} steps): total {tot / nsteps:.0f}")
for k, lab in enumerate(labels):
    print(f"  {lab:16s} {ph[:, k].mean() / nsteps:10.0f}  {100 * ph[:, k].mean() / tot:5.1f} %")
res["step_profile"] = {lab: ph[:, k].mean() / nsteps for k, lab in enumerate(labels)}
print(json.dumps(res))
