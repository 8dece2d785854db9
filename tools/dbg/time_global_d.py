"""Developer timing: leapfrog steps/s of the built-in rank-one metric on the global-memory tier at a given D (256 chains, 20 steps)."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from mici_amd import integrators, models, systems
from oracle import models as omdl
dim = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 256; steps = 20
rng = np.random.default_rng(dim)
system = systems.DenseRiemannianMetricSystem(models.Banana(dim), models.Rank1Metric(omdl.make_spd(dim, rng)))
integ = integrators.ImplicitLeapfrogIntegrator(system, 4.0 / dim)
q0 = rng.standard_normal((n, dim)); p0 = system.sample_momentum_batch(q0, rng.standard_normal((n, dim)))
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    q, p, st, nd = integ.step_batch(q0, p0, 1, n_steps=steps)
    best = min(best, time.perf_counter() - t0)
print(f"D={dim} N={n}: {nd.sum() / best:.4g} steps/s ({best * 1e3:.1f} ms, {int((st != 0).sum())} chains stopped early; incl. upload/download)", integ.last_counters.get("n_inverse_update"))
