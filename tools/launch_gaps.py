import sys, time, numpy as np
sys.path.insert(0, ".")
from mici_amd import integrators, models, systems
from mici_amd.runtime import DeviceBatch, default_context
rng = np.random.default_rng(0)
dim, n = 128, 4096
a = rng.standard_normal((dim, dim)); P = a @ a.T / dim + np.eye(dim)
ctx = default_context()
for metric in (None, P):
    system = systems.EuclideanMetricSystem(models.GaussDense(P), metric=metric)
    integ = integrators.LeapfrogIntegrator(system, 0.05)
    batch = DeviceBatch(ctx, n, dim)
    batch.upload(rng.standard_normal((n, dim)), rng.standard_normal((n, dim)), 1)
    for _ in range(3): integ.step_device(batch, 1000, ctx)
    ctx.sync()
    for K in (1, 5, 20, 50):
        t0 = time.perf_counter()
        for _ in range(K): integ.step_device(batch, 1000, ctx)
        t1 = time.perf_counter()
        ctx.sync()
        t2 = time.perf_counter()
        print(f"metric={'dense' if metric is not None else 'none '} K={K:3d}: enqueue {1e3*(t1-t0):8.2f} ms, total {1e3*(t2-t0):8.2f} ms = {1e3*(t2-t0)/K:6.3f} ms/pass")
    batch.close()
