import sys, time, numpy as np
sys.path.insert(0, '.')
from mici_amd import models, systems, _ffi
from mici_amd.runtime import default_context, DeviceBatch
from oracle import models as omdl
rng = np.random.default_rng(0)
ctx = default_context()
for dim, n in ((256, 1024), (256, 256), (64, 1024), (64, 4096)):
    B = omdl.make_spd(dim, rng)
    system = systems.DenseRiemannianMetricSystem(models.Banana(dim), models.Rank1Metric(B))
    q = rng.standard_normal((n, dim)); p = rng.standard_normal((n, dim))
    batch = DeviceBatch(ctx, n, dim); batch.upload(q, p, None)
    out = np.empty((n, dim))
    m = system.device_model(ctx)
    for rep in range(3):
        ctx.sync(); t0 = time.perf_counter()
        _ffi.check(ctx._lib.mm_dh_dmom(ctx.handle, m.handle, batch.handle, out.ctypes.data_as(_ffi.c_double_p)), ctx.handle)
        dt = time.perf_counter() - t0
    print(f"dh_dmom D={dim} N={n}: {dt*1e3:.3f} ms  -> {dt/n*1e6:.2f} us/chain (1 build + 1 sweep + 1 matvec)")
