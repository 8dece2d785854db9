"""Round 5: leapfrog steps / s of the global-memory dense-Riemannian tier (implicit_global.h) at a few sizes."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mici_amd import integrators, models, systems
from mici_amd.runtime import DeviceBatch, default_context
ctx = default_context()
for dim, n, steps in ((320, 512, 5), (512, 512, 3), (768, 256, 2), (1024, 256, 2)):
    rng = np.random.default_rng(dim)
    a = rng.standard_normal((dim, dim))
    system = systems.DenseRiemannianMetricSystem(models.Banana(dim), models.Rank1Metric(a @ a.T / dim + np.eye(dim)))
    integ = integrators.ImplicitLeapfrogIntegrator(system, 0.005)
    q0 = rng.standard_normal((n, dim))
    p0 = system.sample_momentum_batch(q0, rng.standard_normal((n, dim)))
    batch = DeviceBatch(ctx, n, dim)
    batch.upload(q0, p0, np.ones(n, dtype=np.int8))
    integ.step_device(batch, 1, ctx); ctx.sync()
    t0 = time.perf_counter()
    integ.step_device(batch, steps, ctx); ctx.sync()
    dt = time.perf_counter() - t0
    st, nd = batch.download_status()
    c = integ.last_counters
    print(f"D={dim} {n} chains x {steps} steps: {nd.sum() / dt:.3e} steps/s  ({dt * 1e3 / steps:.1f} ms/step; status ok {np.mean(st == 0):.2f}; "
          f"{c['n_refine'] / max(nd.sum(), 1):.1f} CG pairs, {c['n_factor_full'] / max(nd.sum(), 1):.2f} sweeps per chain-step)")
    batch.close()
