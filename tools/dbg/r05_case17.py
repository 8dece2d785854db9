"""Round 5: stress-fuzz seed 511 case 17 (diag(1 + q^2) metric, D = 90, Steffensen, h = 0.123, 3 steps): chain 3 completes two
steps on the device and one in the oracle.  Which part of the device path decides that?  Run with MICI_AMD_REFINE=0 / 1."""
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
    q, p, st, nd = integ.step_batch(q0, p0, dirs, n_steps=steps)
    print("REFINE", os.environ.get("MICI_AMD_REFINE", "1"), "device status", st.tolist(), "n_done", nd.tolist(),
          {k: integ.last_counters[k] for k in ("n_fp_evals", "n_refine", "n_factor_full", "n_factor_solve")})
    print("oracle", [tuple(int(x) for x in ref(c)[2:]) for c in range(len(q0))])
    for s in (1, 2):
        qs, ps, sts, nds = integ.step_batch(q0, p0, dirs, n_steps=s)
        for c in (3,):
            qo, po, so, no = fp.orc.implicit_leapfrog_steps(osys, q0[c], p0[c], dirs[c] * integ.step_size, s,
                                                            fp_solver=fp.orc.FP_SOLVERS[1])
            print(f"  {s} step(s), chain {c}: device {int(sts[c])}/{int(nds[c])} oracle {so}/{no}  |dq| {np.max(np.abs(qs[c]-qo)):.2e}")
