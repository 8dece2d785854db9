"""Constrained leapfrog with more than one constraint (the C x C Gram matrix: Cholesky inverse + LU solves per Newton
iteration): steps/s of a few sizes.  MICI_AMD_LIB selects the library."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mici_amd import integrators, models, systems  # noqa: E402

rng = np.random.default_rng(3)
for dim, c, n, steps in ((40, 2, 4096, 50), (64, 4, 4096, 30), (256, 8, 2048, 20)):
    A = rng.standard_normal((c, dim)) / np.sqrt(dim)
    b = 0.1 * rng.standard_normal(c)
    system = systems.DenseConstrainedEuclideanMetricSystem(models.GaussIso(dim), models.LinearConstr(A, b))
    integ = integrators.ConstrainedLeapfrogIntegrator(system, 0.05)
    q0 = rng.standard_normal((n, dim))
    q0 -= (np.linalg.pinv(A) @ (A @ q0.T - b[:, None])).T
    p0 = system.sample_momentum_batch(q0, rng.standard_normal((n, dim)))
    integ.step_batch(q0, p0, 1, n_steps=steps)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); q, p, s, nd = integ.step_batch(q0, p0, 1, n_steps=steps); best = min(best, time.perf_counter() - t0)
    print(f"D={dim} C={c} N={n}: {nd.sum() / best:.3e} steps/s (host-timed) failed {int((s != 0).sum())} checksum {np.abs(q).sum():.12e}")
