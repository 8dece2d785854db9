"""Round 5: the actual device-vs-oracle error of tests/test_gpu_implicit.py::test_softabs_long_trajectory_matches_oracle
(tolerance 1e-7) and of the SoftAbs fixtures (tolerance 2e-9), per step count."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mici_amd import integrators, models, systems
from oracle import integrators as orc, models as omdl
rng = np.random.default_rng(17)
dim, n, h = 16, 3, 0.04
w = np.linspace(0.5, 2.0, dim - 1)
system = systems.SoftAbsRiemannianMetricSystem(models.Funnel(w), softabs_coeff=1.0)
osys = orc.RiemannianSystem(omdl.Funnel(w), None, 1.0)
integ = integrators.ImplicitLeapfrogIntegrator(system, h)
q0 = 0.4 * rng.standard_normal((n, dim))
p0 = system.sample_momentum_batch(q0, rng.standard_normal((n, dim)))
for steps in (1, 5, 10, 20, 40):
    q, p, st, nd = integ.step_batch(q0, p0, 1, n_steps=steps)
    err = 0.0
    for c in range(n):
        qo, po, so, no = orc.implicit_leapfrog_steps(osys, q0[c], p0[c], h, steps)
        err = max(err, np.max(np.abs(q[c] - qo) / np.maximum(1, np.abs(qo))), np.max(np.abs(p[c] - po) / np.maximum(1, np.abs(po))))
    print(f"{steps} steps: max scaled error {err:.2e}")
