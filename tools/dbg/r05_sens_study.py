"""Round 5: at which size of an input change does the ORACLE reproduce the device's outcome on the stress-fuzz case the two
end differently on (seed 511, case 17: diag(1 + q^2) metric, D = 90, Steffensen, ten times the usual step)?"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import fuzz_parity as fp
fp.H_FACTOR, fp.STEP_FACTOR = 10.0, 1
rng = np.random.default_rng(511)
for i in range(18):
    desc, integ, system, osys, q0, p0, dirs, steps, ref, tol = fp.riemann_case(rng)
    if i != 17:
        rng.integers(0, len(q0))
        continue
    print(desc)
    q, p, st, nd = integ.step_batch(q0, p0, dirs, n_steps=steps)
    c = 3
    print("device", int(st[c]), int(nd[c]), " oracle", tuple(int(x) for x in ref(c)[2:]))
    os.environ["MICI_AMD_REFINE"] = "0"
    r2 = np.random.default_rng(0)
    for scale in (4e-16, 1e-15, 1e-14, 1e-13, 1e-12):
        outs = []
        qk, pk = q0[c].copy(), p0[c].copy()
        for t in range(8):
            q0[c] = qk * (1 + scale * r2.choice([-1.0, 1.0], size=qk.shape))
            p0[c] = pk * (1 + scale * r2.choice([-1.0, 1.0], size=pk.shape))
            outs.append(tuple(int(x) for x in ref(c)[2:]))
        q0[c], p0[c] = qk, pk
        print(f"oracle, inputs moved by {scale:.0e}:", outs)
