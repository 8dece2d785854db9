"""Phase cycles of the SoftAbs kernels beyond the LDS tier (library built with -DMM_SOFTABS_PROF: tools/ab_build.py)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mici_amd import integrators, models, systems  # noqa: E402

for dim, n, steps in ((128, 8, 4), (256, 8, 4)):
    rng = np.random.default_rng(dim)
    w = np.linspace(0.5, 2.0, dim - 1)
    system = systems.SoftAbsRiemannianMetricSystem(models.Funnel(w), softabs_coeff=1.0)
    integ = integrators.ImplicitLeapfrogIntegrator(system, 0.02)
    q0 = 0.5 * rng.standard_normal((n, dim))
    p0 = system.sample_momentum_batch(q0, rng.standard_normal((n, dim)))
    print(f"--- D={dim}", flush=True)
    t0 = time.perf_counter()
    q, p, s, nd = integ.step_batch(q0, p0, 1, n_steps=steps)
    print(f"{(time.perf_counter() - t0) * 1e3:.0f} ms for {steps} steps", integ.last_counters, flush=True)
