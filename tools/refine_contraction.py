"""Round-3 study behind DESIGN.md section 4.3c (test infrastructure: imports oracle/): how far the metric of the
solve-only constructions of an implicit leapfrog step is from the explicit inverse the step holds, and how many
residual-correction steps (cold / warm started) it takes to reach 1e-13 - on the c3 and c4 workloads of bench.py.  Output: profiles/r03_refine_contraction.txt."""
import os
import sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import integrators as orc, models as omdl
import bench

def run(config, nchain=3, nsteps=3):
    rng = np.random.default_rng(1234)
    dim, h = (64, 0.02) if config == 'c3' else (256, 0.01)
    base = bench._make_spd(dim, rng)
    sysm = orc.RiemannianSystem(omdl.Banana(dim), omdl.Rank1Metric(base))
    q0 = rng.standard_normal((nchain, dim)); z = rng.standard_normal((nchain, dim))
    # instrument DensePD construction
    log = []
    orig_metric = orc.RiemannianSystem.metric
    state = {'F': None, 'u': None, 'p': None}
    def metric(self, st):
        fresh = 'metric' not in st.cache
        m = orig_metric(self, st)
        if fresh:
            log.append(('build', st.pos.copy(), m))
        return m
    orc.RiemannianSystem.metric = metric
    orig_inv_matvec = orc.DensePD.inv_matvec
    def inv_matvec(self, v):
        log.append(('solve', self, v.copy()))
        return orig_inv_matvec(self, v)
    orig_inv = orc.DensePD.inv
    orc.DensePD.inv_matvec = inv_matvec
    og = orc.DensePD.grad_quadratic_form_inv
    def gq(self, v):
        self.full = True
        return og(self, v)
    orc.DensePD.grad_quadratic_form_inv = gq
    ogl = orc.DensePD.grad_log_abs_det.fget
    def gl(self):
        self.full = True
        return ogl(self)
    orc.DensePD.grad_log_abs_det = property(gl)
    res = []
    for c in range(nchain):
        st = orc._State(q0[c], None)
        st.mom = sysm.sample_momentum(st, z[c])
        for s in range(nsteps):
            log.clear()
            orc.implicit_leapfrog_step(sysm, st, h)
            # walk the log: which builds use explicit inverse (.inv accessed) vs solve only
            builds = [e for e in log if e[0] == 'build']
            F = None; uprev = None
            for (_, pos, m) in builds:
                M = m.array if hasattr(m, 'array') else None
                if M is None: M = sysm.rmetric.metric_func(pos)
                needs_inv = getattr(m, 'full', False)
                solves = [e[2] for e in log if e[0] == 'solve' and e[1] is m]
                if F is not None and solves and not needs_inv:
                    p = solves[0]
                    rho = np.linalg.norm(np.eye(dim) - F @ M, 2)
                    exact = np.linalg.solve(M, p)
                    for label, u in (('cold', F @ p), ('warm', uprev if uprev is not None else F @ p)):
                        u = u.copy(); k = 0
                        while np.linalg.norm(u - exact, np.inf) > 1e-13 * max(1, np.linalg.norm(exact, np.inf)) and k < 60:
                            u = u + F @ (p - M @ u); k += 1
                        res.append((c, s, label, rho, k))
                    uprev = exact
                else:
                    if needs_inv or F is None:
                        F = np.linalg.inv(M); uprev = None
                        res.append((c, s, 'INV', 0, 0))
    return res

for cfg in ('c3', 'c4'):
    r = run(cfg, 2, 2)
    print(cfg)
    for x in r: print('  ', x)
