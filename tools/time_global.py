"""Developer timing of the global-memory tier (csrc/implicit_global.h) at the c4_d512 sizes: one construction (build +
sweep + one product: mm_dh_dmom) against whole leapfrog steps, 256 chains, D = 512."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import bench  # noqa: E402
from mici_amd import _ffi  # noqa: E402
from mici_amd.runtime import DeviceBatch, default_context  # noqa: E402

ctx = default_context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
w = bench.make_workload("c4_d512", n, np.random.default_rng(1234))
dim, integ, system = w["dim"], w["integ"], w["system"]
m = system.device_model(ctx)
batch = DeviceBatch(ctx, n, dim)
batch.upload(w["q0"], w["p0"], np.ones(n, dtype=np.int8))
out = np.empty((n, dim))
for rep in range(3):
    ctx.sync()
    t0 = time.perf_counter()
    _ffi.check(ctx._lib.mm_dh_dmom(ctx.handle, m.handle, batch.handle, out.ctypes.data_as(_ffi.c_double_p)), ctx.handle)
    dt = time.perf_counter() - t0
print(f"dh_dmom D={dim} N={n}: {dt * 1e3:.3f} ms (build + sweep + one product + download)")
for traj in (1, 5):
    for rep in range(3):
        batch.upload(w["q0"], w["p0"], np.ones(n, dtype=np.int8))
        ctx.sync()
        t0 = time.perf_counter()
        integ.step_device(batch, traj, ctx)
        ctx.sync()
        dt = time.perf_counter() - t0
    c = integ.last_counters or {}
    print(f"leapfrog x{traj} D={dim} N={n}: {dt * 1e3:.3f} ms  pairs/step {c.get('n_refine', 0) / n / traj:.1f} "
          f"sweeps/step {c.get('n_factor_full', 0) / n / traj:.2f} evals/step {c.get('n_fp_evals', 0) / n / traj:.1f}")
