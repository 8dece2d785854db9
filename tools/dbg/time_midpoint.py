"""Developer timing: ImplicitMidpointIntegrator steps/s of the built-in rank-one metric, one size per kernel family."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from mici_amd import integrators, models, systems
from oracle import models as omdl
for dim, n, steps, h in ((20, 2048, 20, 0.05), (64, 1024, 10, 0.02), (200, 256, 5, 0.01), (320, 128, 4, 0.008)):
    rng = np.random.default_rng(dim)
    system = systems.DenseRiemannianMetricSystem(models.Banana(dim), models.Rank1Metric(omdl.make_spd(dim, rng)))
    integ = integrators.ImplicitMidpointIntegrator(system, h)
    q0 = rng.standard_normal((n, dim)); p0 = system.sample_momentum_batch(q0, rng.standard_normal((n, dim)))
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        q, p, st, nd = integ.step_batch(q0, p0, 1, n_steps=steps)
        best = min(best, time.perf_counter() - t0)
    c = integ.last_counters
    print(f"midpoint D={dim} N={n}: {nd.sum() / best:.4g} steps/s ({best * 1e3:.1f} ms; {int((st != 0).sum())} stopped early) updates {c['n_inverse_update']} sweeps {c['n_factor_full']} evals {c['n_fp_evals']}")
