"""SoftAbs workspace tiers on a PERSISTENT device batch: consecutive launches carry the eigenvector bases (no cold start)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mici_amd import integrators, models, systems  # noqa: E402
from mici_amd.runtime import DeviceBatch, default_context  # noqa: E402

ctx = default_context()
for dim, n, steps in ((128, 256, 2), (256, 256, 2)):
    rng = np.random.default_rng(dim)
    w = np.linspace(0.5, 2.0, dim - 1)
    system = systems.SoftAbsRiemannianMetricSystem(models.Funnel(w), softabs_coeff=1.0)
    integ = integrators.ImplicitLeapfrogIntegrator(system, 0.02)
    q0 = 0.5 * rng.standard_normal((n, dim))
    p0 = system.sample_momentum_batch(q0, rng.standard_normal((n, dim)))
    batch = DeviceBatch(ctx, n, dim)
    batch.upload(q0, p0, 1)
    for k in range(4):
        ctx.sync()
        t0 = time.perf_counter()
        integ.step_device(batch, steps, ctx)
        st, nd = batch.download_status()
        dt = time.perf_counter() - t0
        cn = integ.last_counters
        tot = float(nd.sum())
        print(f"D={dim} n={n} launch {k}: {tot / dt:.3e} steps/s ({dt * 1e3:.0f} ms); per step: eigh {cn['n_eigh'] / tot:.1f} "
              f"refined {cn['n_refine'] / tot:.1f} sweeps {cn['n_newton_iters'] / tot:.2f}; failed {int((st != 0).sum())}", flush=True)
    batch.close()
