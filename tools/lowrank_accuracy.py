"""Accuracy of the Woodbury path of the rank-one-update metric (csrc/implicit_core.h lowrank_solve / lowrank_update),
restated in numpy: M(x) = B + x x^T / D, F = M(x0)^-1 held explicitly.

    python tools/lowrank_accuracy.py

(1) one solve M(x)^-1 p from F against a LAPACK solve of M(x), both measured against an extended-precision solution,
    over position scales and distances |x - x0|;
(2) the held inverse carried through 200 position updates in a row by the symmetric rank-two update, against a fresh
    inverse at every twentieth position.
Test infrastructure: nothing in the product imports this."""
import numpy as np
import numpy.linalg as la


def make_spd(dim, rng):
    a = rng.standard_normal((dim, dim))
    return a @ a.T / dim + np.eye(dim)


def refined_solution(M, p):
    u = la.solve(M, p)
    for _ in range(3):
        r = (p.astype(np.longdouble) - M.astype(np.longdouble) @ u.astype(np.longdouble)).astype(float)
        u = u + la.solve(M, r)
    return u


def solve_from_inverse(F, x0, x, p, dim):
    """implicit_core.h lowrank_solve: u = c - (F d) w1 - b w2."""
    d = x - x0
    ad, b, c = F @ d, F @ x0, F @ p
    e3, e4, r2, sbb, sbc = d @ ad, d @ b, d @ c, x0 @ b, x0 @ c
    D = float(dim)
    k11, k12, k21, k22, r1 = D + e3 + e4, sbb + e4, e3, D + e4, sbc + r2
    det = k11 * k22 - k12 * k21
    w1, w2 = (r1 * k22 - k12 * r2) / det, (k11 * r2 - k21 * r1) / det
    return c - (ad * w1 + b * w2), det / (D * D)


def update_inverse(F, x0, d, dim):
    """implicit_core.h lowrank_update: F(x0) -> F(x0 + d)."""
    a, b = F @ d, F @ x0
    e3, e4, sbb = d @ a, d @ b, x0 @ b
    D = float(dim)
    k11, k12, k21, k22 = D + e3 + e4, sbb + e4, e3, D + e4
    det = k11 * k22 - k12 * k21
    al, be, ga = -(k22 - k12) / det, -k22 / det, k21 / det
    return F + np.outer(a, al * a + be * b) + np.outer(b, be * a + ga * b)


def main():
    rng = np.random.default_rng(0)
    print("(1) relative error of M(x)^-1 p: Woodbury from F = M(x0)^-1 | LAPACK solve of M(x)   [det K / D^2]")
    for dim in (64, 256, 512):
        B = make_spd(dim, rng)
        for scale in (1.0, 10.0, 100.0):
            for dl in (1e-1, 1e-2, 1e-4):
                x0 = scale * rng.standard_normal(dim)
                x = x0 + dl * scale * rng.standard_normal(dim)
                p = rng.standard_normal(dim)
                M = B + np.outer(x, x) / dim
                F = la.inv(B + np.outer(x0, x0) / dim)
                F = 0.5 * (F + F.T)
                u, detn = solve_from_inverse(F, x0, x, p, dim)
                ref = refined_solution(M, p)
                e_w = la.norm(u - ref) / la.norm(ref)
                e_l = la.norm(la.solve(M, p) - ref) / la.norm(ref)
                print(f"  D {dim:4d} |x| ~ {scale:5.0f} |d| / |x| {dl:6.0e}: {e_w:.2e} | {e_l:.2e}   [{detn:.3g}]")
    print("(2) the held inverse through 200 updates in a row (D = 64, |d| = 0.05 sqrt(D)): max |F - inv(M(x))| / max |inv|")
    dim = 64
    B = make_spd(dim, rng)
    x = 2.0 * rng.standard_normal(dim)
    F = la.inv(B + np.outer(x, x) / dim)
    for step in range(200):
        d = 0.05 * rng.standard_normal(dim)
        F = update_inverse(F, x, d, dim)
        x = x + d
        if step % 20 == 19:
            Fx = la.inv(B + np.outer(x, x) / dim)
            print(f"  after {step + 1:3d}: {np.abs(F - Fx).max() / np.abs(Fx).max():.2e}   asymmetry {np.abs(F - F.T).max():.1e}")


if __name__ == "__main__":
    main()
