"""Prototype of the matrix-product eigenvector refinement considered for the SoftAbs path (k_softabs.hip): replay the
sequence of Hessians one c3(b) chain decomposes (oracle, scaled funnel D = 64, h = 0.02) and refine the previous
eigenvectors X towards the new Hessian with the Ogita-Aishima iteration
    R = I - X^T X, S = X^T A X, lam_i = S_ii / (1 - R_ii), delta = 2 (|S - D|_F + |A| |R|_F),
    E_ij = (S_ij + lam_j R_ij) / (lam_j - lam_i) if |lam_i - lam_j| > delta else R_ij / 2,   X <- X + X E
(SIAM J. Matrix Anal. Appl. / Japan J. Indust. Appl. Math. 2018) - four D^3 products a pass, quadratic convergence,
multiple eigenvalues allowed.  Prints passes per decomposition and the final accuracy against numpy.linalg.eigh.
Runs on the CPU: python tools/refine_eigh_proto.py [n_steps]"""
import sys
import numpy as np

sys.path.insert(0, ".")
from oracle import integrators as orc  # noqa: E402
from oracle import models as omdl  # noqa: E402


MODE = "pair"
START_MAX = 0.35
GUARD = 1e-6


def refine(a, x, max_pass=8, log=None):
    n = a.shape[0]
    prev = np.inf
    for k in range(max_pass):
        g = a @ x
        s = x.T @ g
        r = np.eye(n) - x.T @ x
        lam = np.diag(s) / (1.0 - np.diag(r))
        norm_a = np.max(np.abs(lam))
        off = s - np.diag(np.diag(s))
        delta = 2.0 * (np.linalg.norm(off) + norm_a * np.linalg.norm(r))
        gap = lam[None, :] - lam[:, None]
        far = np.abs(gap) > (delta if MODE == 'delta' else GUARD * norm_a)
        e = np.where(far, (s + lam[None, :] * r) / np.where(far, gap, 1.0), 0.5 * r)
        max_e = np.max(np.abs(e))
        near_s = np.max(np.abs(np.where(far | np.eye(n, dtype=bool), 0.0, s))) / norm_a
        if log is not None:
            log.append((k, max_e, near_s, delta / norm_a))
        if k == 0 and not max_e < START_MAX:
            return None, None, k, "start too far"
        x = x + x @ e
        if max_e < 1e-7:
            return (lam, x, k + 1, "ok") if near_s < 1e-11 else (None, None, k + 1, "split cluster")
        if not max_e < prev:
            return None, None, k + 1, "not contracting"
        prev = max_e
    return None, None, max_pass, "too many passes"


def main():
    n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dim, h = 64, 0.02
    wts = np.ones(dim - 1) if "equal" in sys.argv else np.linspace(0.5, 2.0, dim - 1)
    system = orc.RiemannianSystem(omdl.Funnel(wts), None, 1.0)
    hessians = []
    orig = np.linalg.eigh

    def spy(m):
        hessians.append(np.array(m))
        return orig(m)

    rng = np.random.default_rng(2)
    np.linalg.eigh = spy
    try:
        for chain in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
            q = rng.standard_normal(dim)
            z = rng.standard_normal(dim)
            p = system.sample_momentum(orc._State(q, z), z)
            orc.implicit_leapfrog_steps(system, q, p, h, n_steps)
    finally:
        np.linalg.eigh = orig
    print(f"{len(hessians)} decompositions recorded")
    x = None
    passes, outcomes, worst = [], {}, 0.0
    for idx, a in enumerate(hessians):
        lam_ref, v_ref = orig(a)
        if x is None:
            x = v_ref.copy()
            continue
        log = []
        lam, xn, k, why = refine(a, x, log=log)
        outcomes[why] = outcomes.get(why, 0) + 1
        if lam is None:
            print(idx, why, ["%d: E %.1e nearS %.1e delta %.1e" % t for t in log])
            x = v_ref.copy()
            continue
        passes.append(k)
        x = xn
        # accuracy of what the metric uses: V f(lam) V^T with f = softabs
        f = lambda t: t / np.tanh(t)  # noqa: E731
        m_ref = (v_ref * f(lam_ref)) @ v_ref.T
        m = (x * f(lam)) @ x.T
        worst = max(worst, np.max(np.abs(m - m_ref)) / np.max(np.abs(m_ref)))
        if idx < 30:
            print(idx, k, ["E %.1e nearS %.1e" % (t[1], t[2]) for t in log])
    print("outcomes", outcomes, "mean passes", np.mean(passes), "hist", np.bincount(passes))
    print("worst relative error of V softabs(lam) V^T against numpy eigh: %.2e" % worst)
    print("orthogonality of the last basis: %.2e" % np.max(np.abs(x.T @ x - np.eye(dim))))


if __name__ == "__main__":
    main()
