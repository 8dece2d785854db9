"""SoftAbs beyond the LDS tier (64 < D <= 256, matrices in a per-chain HBM workspace): steps/s and Jacobi work."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mici_amd import integrators, models, systems  # noqa: E402

for dim, n, steps in ((128, 256, 2), (200, 32, 2), (200, 256, 2), (256, 32, 2), (256, 256, 2), (256, 512, 2)):
    rng = np.random.default_rng(dim)
    w = np.linspace(0.5, 2.0, dim - 1)
    system = systems.SoftAbsRiemannianMetricSystem(models.Funnel(w), softabs_coeff=1.0)
    integ = integrators.ImplicitLeapfrogIntegrator(system, 0.02)
    q0 = 0.5 * rng.standard_normal((n, dim))
    p0 = system.sample_momentum_batch(q0, rng.standard_normal((n, dim)))
    integ.step_batch(q0[:2], p0[:2], 1, n_steps=1)
    t0 = time.perf_counter()
    q, p, s, nd = integ.step_batch(q0, p0, 1, n_steps=steps)
    dt = time.perf_counter() - t0
    cn = integ.last_counters
    tot = float(nd.sum())
    print(f"D={dim} n={n}: {tot / dt:.3e} steps/s ({dt * 1e3:.0f} ms); per step: eigh {cn['n_eigh'] / tot:.1f} "
          f"sweeps {cn['n_newton_iters'] / tot:.1f}; failed {int((s != 0).sum())}", flush=True)
    t0 = time.perf_counter()
    q, p, s, nd = integ.step_batch(q, p, 1, n_steps=steps)  # the next launch of the same chains: bases carried over
    dt = time.perf_counter() - t0
    cn = integ.last_counters
    tot = float(nd.sum())
    print(f"   next launch: {tot / dt:.3e} steps/s ({dt * 1e3:.0f} ms); per step: eigh {cn['n_eigh'] / tot:.1f} "
          f"sweeps {cn['n_newton_iters'] / tot:.1f}; failed {int((s != 0).sum())}", flush=True)
