"""One implicit step of a user metric at the given dims (default 128 270) on the run-time compiled team / block-16 kernels against
the oracle, for each form of the source (plain, aux + dense VJP, aux + flat VJP).  python tools/dbg/team_user.py [dims...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, 'tests')
import numpy as np
from user_sources import SOFTPLUS_RANK1, SOFTPLUS_RANK1_FAST_WIDE
from oracle import integrators as orc, models as omdl
from mici_amd import integrators, models, systems
FAST = SOFTPLUS_RANK1_FAST_WIDE
i0 = FAST.index("template <class Ops>")
AUX_DENSE = FAST[:i0].replace("#define MM_USER_VJP_FLAT\n", "") + r"""
__device__ double mm_user_vjp(const double* q, const MmMat& V, int k, int dim, const double* params, const double* aux) {
  double cvc = 0.0;
  for (int i = 0; i < dim; ++i) {
    double r = 0.0;
    for (int j = 0; j < dim; ++j) r += V(i, j) * params[j];
    cvc += params[i] * r;
  }
  return V(k, k) / (1.0 + exp(-q[k])) + cvc * 2.0 * q[k] / (double)dim;
}
"""
j0 = SOFTPLUS_RANK1.index("__device__ double mm_user_vjp")
NOAUX_FLAT = "#define MM_USER_VJP_FLAT\n" + SOFTPLUS_RANK1[:j0] + FAST[i0:]
for dim in (int(a) for a in sys.argv[1:] or ["128"]):
    rng = np.random.default_rng(1000 + dim)
    n = 3
    c = 0.5 * rng.standard_normal(dim)
    osys = orc.RiemannianSystem(omdl.Banana(dim), omdl.SoftPlusRank1Metric(c))
    q0 = rng.standard_normal((n, dim))
    z = rng.standard_normal((n, dim))
    p0 = np.stack([osys.sample_momentum(orc._State(q0[k], None), z[k]) for k in range(n)])
    for name, src in (("plain", SOFTPLUS_RANK1), ("aux+dense", AUX_DENSE), ("noaux+flat", NOAUX_FLAT), ("fast", FAST)):
        user = systems.DenseRiemannianMetricSystem(models.Banana(dim), models.UserMetric(dim, src, c))
        for cls, ofn, h in ((integrators.ImplicitLeapfrogIntegrator, orc.implicit_leapfrog_steps, 0.015),
                            (integrators.ImplicitMidpointIntegrator, orc.implicit_midpoint_steps, 0.015)):
            q, p, st, nd = cls(user, h).step_batch(q0, p0, 1, n_steps=1)
            err = 0.0
            for k in range(n):
                qo, po, so, no = ofn(osys, q0[k], p0[k], h, 1)
                err = max(err, np.abs(q[k] - qo).max(), np.abs(p[k] - po).max())
            print(dim, name, cls.__name__, "status", st.tolist(), "n_done", nd.tolist(), "max err %.2e" % err, flush=True)
