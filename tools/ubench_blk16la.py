#!/usr/bin/env python3
"""Construction timings of the look-ahead block-16 kernel (k_implicit_blk16la.hip) next to k_implicit_blk16.hip: one chain
per CU (256 chains), each construction repeated `reps` times inside one launch -> microseconds and cycles (2.4 GHz)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mici_amd import _ffi, models, systems  # noqa: E402
from mici_amd.runtime import DeviceBatch, default_context  # noqa: E402

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
reps = 20
rng = np.random.default_rng(0)
a = rng.standard_normal((dim, dim))
base = a @ a.T / dim + np.eye(dim)
system = systems.DenseRiemannianMetricSystem(models.Banana(dim), models.Rank1Metric(base))
ctx = default_context()
batch = DeviceBatch(ctx, n, dim)
batch.upload(rng.standard_normal((n, dim)), rng.standard_normal((n, dim)), np.ones(n, dtype=np.int8))
out = np.zeros((n, 256))


def timed(fn, op):
    t = {}
    for r in (0, reps):
        best = 1e9
        for _ in range(3):
            ms = C.c_double(0.0)
            _ffi.check(fn(ctx.handle, system.device_model(ctx).handle, batch.handle, op,
                          out.ctypes.data_as(_ffi.c_double_p), None, r, C.byref(ms)), ctx.handle, "debug hook")
            best = min(best, ms.value)
        t[r] = best
    return (t[reps] - t[0]) / reps * 1e3


for name, sym, ops in (("blk16", "mm_debug_blk16_linalg", {"build+full sweep": 4, "build+trailing sweep": 5}),
                       ("blk16la", "mm_debug_blk16la_linalg", {"build+full sweep": 3, "build+trailing sweep+solve": 4})):
    fn = getattr(ctx._lib, sym)
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, _ffi.c_double_p, _ffi.c_int32_p, C.c_int, _ffi.c_double_p]
    for what, op in ops.items():
        us = timed(fn, op)
        print(f"{name:8s} {what:28s} {us:8.2f} us  {us * 2.4:8.1f} kcycles", flush=True)

fn = ctx._lib.mm_debug_blk16la_linalg
for op, name in ((5, "full sweep"), (6, "trailing sweep")):
    ms = C.c_double(0.0)
    _ffi.check(fn(ctx.handle, system.device_model(ctx).handle, batch.handle, op, out.ctypes.data_as(_ffi.c_double_p),
                  None, 0, C.byref(ms)), ctx.handle, "debug hook")
    prof = out[:, :64].reshape(n, 8, 8).mean(0)  # [wave][phase]
    nb = prof[0, 6]
    print(f"blk16la {name}: cycles per block (mean over chains; {nb:.0f} blocks), per wave:")
    print("  wave     -W    ahead+publish  barrier1   updates   invert   barrier2 | block")
    for w in range(8):
        print(f"  {w}    " + "  ".join(f"{prof[w, k] / nb:8.0f}" for k in range(6)) + f"   | {prof[w, :6].sum() / nb:8.0f}")
