"""Round 5: actual device-vs-reference error on every SoftAbs fixture (test tolerance 2e-9)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import golden_names, load_golden
from test_gpu_implicit import build
worst = 0.0
for name in golden_names("softabs") + golden_names("extreme_softabs"):
    g = load_golden(name)
    system, integ = build(g)
    e = 0.0
    for k, s in enumerate(int(s) for s in g["checkpoints"]):
        q, p, st, nd = integ.step_batch(g["q0"], g["p0"], g["dir"], n_steps=s)
        fin = np.isfinite(g["q_out"][k]) & np.isfinite(q)
        for a, b in ((q, g["q_out"][k]), (p, g["p_out"][k])):
            d = np.abs(np.where(fin, a - b, 0.0)) / np.maximum(1.0, np.abs(np.where(fin, b, 0.0)))
            e = max(e, float(d.max()))
    worst = max(worst, e)
    print(f"{name}: {e:.2e}")
print(f"worst {worst:.2e}")
