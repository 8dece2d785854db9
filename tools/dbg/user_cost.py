"""Where a user metric's time goes on the c3 sizes (D = 64, 1024 chains, 100 steps): the built-in rank-one metric, the same
metric as user source, and the softplus metric of c3_user - steps/s and the executed counts per chain-step."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mici_amd import integrators, models, systems, user_examples  # noqa: E402

dim, n, h, traj = 64, 1024, 0.02, 100
rng = np.random.default_rng(7)
A = rng.standard_normal((dim, dim)); B = A @ A.T / dim + np.eye(dim)
c = 0.5 * rng.standard_normal(dim)
cases = [("builtin rank1", models.Rank1Metric(B)),
         ("user rank1 flat", models.UserMetric(dim, user_examples.RANK1_AS_USER_FLAT, B)),
         ("user softplus fast", models.UserMetric(dim, user_examples.SOFTPLUS_RANK1_FAST, c))]
q0 = rng.standard_normal((n, dim)); z = rng.standard_normal((n, dim))
for name, rm in cases:
    system = systems.DenseRiemannianMetricSystem(models.Banana(dim), rm)
    integ = integrators.ImplicitLeapfrogIntegrator(system, h)
    p0 = system.sample_momentum_batch(q0, z)
    integ.step_batch(q0, p0, 1, n_steps=traj)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter(); q, p, s, nd = integ.step_batch(q0, p0, 1, n_steps=traj); best = min(best, time.perf_counter() - t0)
    cn = integ.last_counters; tot = float(nd.sum())
    print(f"{name:20s} {tot / best:.3e} steps/s (host-timed, incl. transfers)  per step: pairs {cn['n_refine'] / tot:.1f} full {cn['n_factor_full'] / tot:.2f} "
          f"solve-sweeps {cn['n_factor_solve'] / tot:.2f} fp_evals {cn['n_fp_evals'] / tot:.1f} metric {cn['n_metric'] / tot:.1f} failed {int((s != 0).sum())}")
