"""The VALU team kernels (64 < D <= 75, 256 < D <= 279) with / without the refinement of the solve-only constructions
(MICI_AMD_REFINE=0 in a second process): steps/s, executed counts, and the difference of the results."""
import sys, os, time, subprocess, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from mici_amd import integrators, models, systems, user_examples  # noqa: E402

out = sys.argv[1] if len(sys.argv) > 1 else None
res = {}
for dim, n, steps in ((70, 512, 10), (270, 256, 5)):
    rng = np.random.default_rng(dim)
    A = rng.standard_normal((dim, dim)); B = A @ A.T / dim + np.eye(dim)
    for name, rm in (("rank1", models.Rank1Metric(B)), ("diagquad", models.DiagQuadMetric(dim)),
                     ("user rank1 flat", models.UserMetric(dim, user_examples.RANK1_AS_USER_FLAT, B))):
        system = systems.DenseRiemannianMetricSystem(models.Banana(dim), rm)
        integ = integrators.ImplicitLeapfrogIntegrator(system, 0.01)
        q0 = rng.standard_normal((n, dim)); p0 = system.sample_momentum_batch(q0, rng.standard_normal((n, dim)))
        integ.step_batch(q0, p0, 1, n_steps=steps)
        t0 = time.perf_counter(); q, p, s, nd = integ.step_batch(q0, p0, 1, n_steps=steps); dt = time.perf_counter() - t0
        cn = integ.last_counters; tot = float(nd.sum())
        print(f"D={dim} {name:16s} {tot / dt:.3e} steps/s  per step: pairs {cn['n_refine'] / tot:.1f} full {cn['n_factor_full'] / tot:.2f} "
              f"solve-sweeps {cn['n_factor_solve'] / tot:.2f} fp_evals {cn['n_fp_evals'] / tot:.1f} failed {int((s != 0).sum())}")
        res[f"{dim}_{name}"] = np.concatenate([q.ravel(), p.ravel(), s.astype(float), nd.astype(float), [cn['n_fp_evals']]])
if out:
    np.savez(out, **res)
else:
    TMP = tempfile.mkdtemp()  # scratch of this invocation (no fixed /tmp names)
    np.savez(os.path.join(TMP, 'team_refine_on.npz'), **res)
    env = dict(os.environ, MICI_AMD_REFINE='0')
    print("MICI_AMD_REFINE=0:")
    subprocess.run([sys.executable, __file__, os.path.join(TMP, 'team_refine_off.npz')], env=env, check=True)
    a, b = np.load(os.path.join(TMP, 'team_refine_on.npz')), np.load(os.path.join(TMP, 'team_refine_off.npz'))
    for k in a.files:
        x, y = a[k], b[k]
        print(k, 'max |diff| of states / statuses / counts %.2e' % np.max(np.abs(x - y) / np.maximum(1.0, np.abs(y))))
