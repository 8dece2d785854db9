"""Round-3 study behind DESIGN.md section 4.3c (test infrastructure: imports oracle/): how far the metric of the
solve-only constructions of an implicit leapfrog step is from the explicit inverse the step holds, and how many
preconditioned-CG iterations it takes to reach 1e-13 - on the c3 and c4 workloads of bench.py.  Output: profiles/r03_refine_contraction.txt."""
import os
import sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import integrators as orc, models as omdl
import bench

class RefSys(orc.RiemannianSystem):
    tol = 1e-13; safety = 4.0; rho_max = 0.1
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.stats = dict(steps=0, solves=0, fallback=0, hist=[])
        self.anchor = None
    def dh2_dmom(self, st):
        if 'metric' in st.cache:
            m = st.cache['metric']; u = m.inv @ st.mom
            self.anchor = dict(F=m.inv, tracks=[(st.pos.copy(), u.copy())], memo=None)
            return u
        a = self.anchor
        if a['memo'] is not None and np.array_equal(a['memo'][0], st.pos):
            return a['memo'][1]
        M = self.rmetric.metric_func(st.pos); p = st.mom; F = a['F']
        # guess: nearest track
        d = [np.linalg.norm(st.pos - t[0]) for t in a['tracks']]
        u = a['tracks'][int(np.argmin(d))][1].copy()
        k = 0; ok = False
        r = p - M @ u; z = F @ r; d = z.copy(); rz = r @ z
        pu = abs(p @ u)
        if rz <= (self.tol ** 2) * pu: ok = True
        while not ok and k < 12:
            q = M @ d; dq = d @ q
            if not dq > 0: break
            al = rz / dq; u = u + al * d; r = r - al * q; z = F @ r; rz2 = r @ z; k += 1
            if rz2 <= (self.tol ** 2) * abs(p @ u): ok = True; break
            d = z + (rz2 / rz) * d; rz = rz2
        self.stats['solves'] += 1; self.stats['steps'] += k; self.stats['hist'].append(k)
        if not ok:
            self.stats['fallback'] += 1
            u = np.linalg.solve(M, p)
        exact = np.linalg.solve(M, p)
        self.stats.setdefault('err', []).append(np.abs(u - exact).max() / np.abs(exact).max())
        a['memo'] = (st.pos.copy(), u)
        # tracks: keep anchor + up to 2 latest distinct
        a['tracks'].append((st.pos.copy(), u.copy()))
        return u

def run(config, nchain, nsteps, tol):
    rng = np.random.default_rng(1234)
    dim, h = (64, 0.02) if config == 'c3' else (256, 0.01)
    base = bench._make_spd(dim, rng)
    q0 = rng.standard_normal((nchain, dim)); z = rng.standard_normal((nchain, dim))
    out = {}
    for name, cls in (('exact', orc.RiemannianSystem), ('refine', RefSys)):
        sysm = cls(omdl.Banana(dim), omdl.Rank1Metric(base))
        if name == 'refine': sysm.tol = tol
        fin = []
        for c in range(nchain):
            st = orc._State(q0[c], None); st.mom = sysm.sample_momentum(st, z[c])
            for s in range(nsteps): orc.implicit_leapfrog_step(sysm, st, h)
            fin.append(np.concatenate([st.pos, st.mom]))
        out[name] = (np.array(fin), sysm)
    dev = np.abs(out['exact'][0] - out['refine'][0]).max()
    s = out['refine'][1].stats
    ce, cr = out['exact'][1].counters, out['refine'][1].counters
    print(config, 'tol', tol, 'dev', dev, 'refine steps/leapfrog step', s['steps'] / (nchain * nsteps), 'solves/step', s['solves'] / (nchain*nsteps), 'fallback', s['fallback'], 'maxerr', max(s['err']), 'hist', np.bincount(s['hist']))
    print('  counters exact', dict(ce), 'refine', dict(cr))

for cfg, nc, ns in (('c3', 2, 100), ('c4', 1, 20)):
    for tol in (1e-13, 1e-14, 1e-15):
        run(cfg, nc, ns, tol)
