"""Randomised parity sweep: HIP path vs the oracle over random sizes / models / integrators.

    python tools/fuzz_parity.py [--seed S] [--cases N]

Test infrastructure (imports oracle/).  Prints one line per case and a summary; exits non-zero on a mismatch."""
import argparse
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import integrators as orc  # noqa: E402
from oracle import models as omdl  # noqa: E402
from mici_amd import integrators, models, solvers, systems  # noqa: E402


def close(a, b, tol):
    a, b = np.asarray(a), np.asarray(b)
    return np.array_equal(np.isnan(a), np.isnan(b)) and np.all(
        np.abs(np.nan_to_num(a) - np.nan_to_num(b)) <= tol * np.maximum(1.0, np.abs(np.nan_to_num(b))))


def targets(dim, rng, kinds):
    k = rng.choice(kinds)
    if k == "iso":
        return models.GaussIso(dim), omdl.GaussIso(dim)
    if k == "diag":
        pr = np.exp(0.3 * rng.standard_normal(dim))
        return models.GaussDiag(pr), omdl.GaussDiag(pr)
    if k == "dense":
        P = omdl.make_spd(dim, rng)
        return models.GaussDense(P), omdl.GaussDense(P)
    if k == "poly":
        a, b = rng.uniform(0, 1.5), rng.uniform(0, 0.5)
        return models.Poly(dim, a, b), omdl.Poly(dim, a, b)
    return models.Banana(dim), omdl.Banana(dim)


def metric_of(dim, rng):
    k = rng.choice(["identity", "diag", "dense"])
    if k == "identity":
        return omdl.METRIC_IDENTITY, None
    if k == "diag":
        return omdl.METRIC_DIAG, np.exp(0.3 * rng.standard_normal(dim))
    return omdl.METRIC_DENSE, omdl.make_spd(dim, rng)


def euclid_case(rng):
    dim = int(rng.choice([1, 2, 3, 7, 15, 16, 17, 31, 33, 48, 64, 65, 100, 127, 128, 129, 200]))
    n = int(rng.choice([1, 3, 15, 16, 17, 70, 300]))
    gaussian = rng.random() < 0.3
    pt, ot = targets(dim, rng, ["iso", "diag", "dense", "poly", "banana"] if dim >= 2 else ["iso", "poly"])
    mk, metric = metric_of(dim, rng)
    if gaussian:
        system, osys = systems.GaussianEuclideanMetricSystem(pt, metric=metric), orc.GaussianEuclidSystem(ot, mk, metric)
    else:
        system, osys = systems.EuclideanMetricSystem(pt, metric=metric), orc.EuclidSystem(ot, mk, metric)
    kind = rng.choice(["leapfrog", "bcss2", "bcss3", "bcss4", "midpoint"] if dim <= 128 else
                      ["leapfrog", "bcss2", "bcss3", "bcss4"])
    h, steps = float(rng.uniform(0.02, 0.15)), int(rng.integers(1, 12))
    if isinstance(ot, omdl.Banana):
        h *= 0.2
    q0 = rng.standard_normal((n, dim))
    p0 = np.stack([osys.msqrt(z) for z in rng.standard_normal((n, dim))])
    dirs = np.where(rng.random(n) < 0.5, 1, -1).astype(np.int8)
    desc = f"euclid{'-gauss' if gaussian else ''} {kind} D={dim} N={n} {type(ot).__name__} metric={mk} h={h:.3f} steps={steps}"
    if kind == "leapfrog":
        integ = integrators.LeapfrogIntegrator(system, h)
        ref = lambda c: orc.leapfrog_steps(osys, q0[c], p0[c], dirs[c] * h, steps) + (0, steps)  # noqa: E731
    elif kind == "midpoint":
        integ = integrators.ImplicitMidpointIntegrator(system, h)
        ref = lambda c: orc.implicit_midpoint_steps(osys, q0[c], p0[c], dirs[c] * h, steps)  # noqa: E731
    else:
        st = int(kind[-1])
        integ = {2: integrators.BCSSTwoStageIntegrator, 3: integrators.BCSSThreeStageIntegrator,
                 4: integrators.BCSSFourStageIntegrator}[st](system, h)
        ref = lambda c: orc.composition_steps(osys, q0[c], p0[c], dirs[c] * h, steps,  # noqa: E731
                                              orc.BCSS_FREE_COEFFICIENTS[st]) + (0, steps)
    return desc, integ, system, osys, q0, p0, dirs, steps, ref, (1e-9 if kind == "midpoint" else 1e-11)


def riemann_case(rng):
    dim = int(rng.choice([1, 2, 5, 8, 9, 16, 31, 32, 33, 40, 63, 64, 65, 70, 75, 76, 90, 128, 200, 255, 256, 257, 270, 279]))
    n = int(rng.choice([1, 2, 5, 9]))
    which = rng.choice(["rank1", "diagquad"])
    pt, ot = targets(dim, rng, ["poly", "banana"] if dim >= 2 else ["poly"])
    if which == "rank1":
        B = omdl.make_spd(dim, rng)
        pm, om = models.Rank1Metric(B), omdl.Rank1Metric(B)
    else:
        pm, om = models.DiagQuadMetric(dim), omdl.DiagQuadMetric(dim)
    system = systems.DenseRiemannianMetricSystem(pt, pm)
    osys = orc.RiemannianSystem(ot, om, None, orc.Counters())
    h, steps = H_FACTOR * float(rng.uniform(0.01, 0.06)), int(rng.integers(1, 5))
    if isinstance(ot, omdl.Banana):
        h *= 0.4
    solver = int(rng.integers(0, 2))
    integ = integrators.ImplicitLeapfrogIntegrator(
        system, h, fixed_point_solver=[solvers.solve_fixed_point_direct, solvers.solve_fixed_point_steffensen][solver])
    q0 = rng.standard_normal((n, dim))
    p0 = system.sample_momentum_batch(q0, rng.standard_normal((n, dim)))
    dirs = np.where(rng.random(n) < 0.5, 1, -1).astype(np.int8)
    desc = f"riemann {which} D={dim} N={n} {type(ot).__name__} solver={solver} h={h:.3f} steps={steps}"
    ref = lambda c: orc.implicit_leapfrog_steps(  # noqa: E731
        osys, q0[c], p0[c], dirs[c] * h, steps, fp_solver=orc.FP_SOLVERS[solver])
    return desc, integ, system, osys, q0, p0, dirs, steps, ref, 1e-9


H_FACTOR = 1.0  # --stress: step sizes this much larger (fixed points that diverge / run out of iterations: the statuses must agree)
STEP_FACTOR = 1  # --long: longer SoftAbs trajectories (the decompositions are refined from one another, k_softabs.hip)


def softabs_case(rng):
    dim = int(rng.choice([2, 3, 5, 8, 13, 16, 17, 33, 48, 63, 64, 65, 72, 100, 127, 128]))
    n = int(rng.choice([1, 2, 5]))
    if rng.random() < 0.5:
        w = np.linspace(0.5, 2.0, dim - 1)
        pt, ot = models.Funnel(w), omdl.Funnel(w)
    else:
        a, b = rng.uniform(0.5, 1.5), rng.uniform(0.1, 0.5)
        pt, ot = models.Poly(dim, a, b), omdl.Poly(dim, a, b)
    coeff = float(rng.choice([0.5, 1.0, 2.0]))
    system = systems.SoftAbsRiemannianMetricSystem(pt, softabs_coeff=coeff)
    osys = orc.RiemannianSystem(ot, None, coeff, orc.Counters())
    h, steps = H_FACTOR * float(rng.uniform(0.01, 0.05)), int(rng.integers(1, 4)) * STEP_FACTOR
    integ = integrators.ImplicitLeapfrogIntegrator(system, h)
    q0 = 0.5 * rng.standard_normal((n, dim))
    p0 = system.sample_momentum_batch(q0, rng.standard_normal((n, dim)))
    dirs = np.where(rng.random(n) < 0.5, 1, -1).astype(np.int8)
    desc = f"softabs D={dim} N={n} {type(ot).__name__} coeff={coeff} h={h:.3f} steps={steps}"
    ref = lambda c: orc.implicit_leapfrog_steps(osys, q0[c], p0[c], dirs[c] * h, steps)  # noqa: E731
    return desc, integ, system, osys, q0, p0, dirs, steps, ref, 5e-8


def riemann_user_case(rng):
    """A metric the library does not have, as USER source (run-time compiled kernels, csrc/user_metric.h): the softplus +
    rank-one metric of oracle/models.py at ANY dim up to the ceiling."""
    from mici_amd import user_examples
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import user_sources
    dim = int(rng.integers(2, 280))
    n = int(rng.choice([1, 3, 9]))
    form = rng.choice(["plain", "fast"])
    src = user_sources.SOFTPLUS_RANK1 if form == "plain" else user_sources.softplus_fast(dim)
    c = 0.3 * rng.standard_normal(dim)
    pt, ot = targets(dim, rng, ["poly", "banana"])
    system = systems.DenseRiemannianMetricSystem(pt, models.UserMetric(dim, src, c))
    osys = orc.RiemannianSystem(ot, omdl.SoftPlusRank1Metric(c), None, orc.Counters())
    h, steps = H_FACTOR * float(rng.uniform(0.01, 0.05)), int(rng.integers(1, 4))
    if isinstance(ot, omdl.Banana):
        h *= 0.4
    midpoint = rng.random() < 0.3
    integ = (integrators.ImplicitMidpointIntegrator if midpoint else integrators.ImplicitLeapfrogIntegrator)(system, h)
    q0 = 0.7 * rng.standard_normal((n, dim))
    p0 = system.sample_momentum_batch(q0, rng.standard_normal((n, dim)))
    dirs = np.where(rng.random(n) < 0.5, 1, -1).astype(np.int8)
    desc = f"riemann_user softplus:{form} D={dim} N={n} {type(ot).__name__} {'midpoint' if midpoint else 'leapfrog'} h={h:.3f} steps={steps}"
    stepper = orc.implicit_midpoint_steps if midpoint else orc.implicit_leapfrog_steps
    ref = lambda cc: stepper(osys, q0[cc], p0[cc], dirs[cc] * h, steps)  # noqa: E731
    return desc, integ, system, osys, q0, p0, dirs, steps, ref, 1e-9


def softabs_user_case(rng):
    """SoftAbs with the Hessian / MTP as USER source (the dense path of csrc/softabs.h): the banana's tridiagonal Hessian."""
    from mici_amd import user_examples
    dim = int(rng.integers(2, 65))
    n = int(rng.choice([1, 2, 5]))
    coeff = float(rng.choice([0.5, 1.0, 2.0]))
    system = systems.SoftAbsRiemannianMetricSystem(
        models.Banana(dim), softabs_coeff=coeff, hess_neg_log_dens=models.UserHessian(user_examples.BANANA_HESS))
    osys = orc.RiemannianSystem(omdl.Banana(dim), None, coeff, orc.Counters())
    h, steps = H_FACTOR * float(rng.uniform(0.004, 0.012)), int(rng.integers(1, 4)) * STEP_FACTOR
    integ = integrators.ImplicitLeapfrogIntegrator(system, h)
    q0 = 0.5 * rng.standard_normal((n, dim))
    p0 = system.sample_momentum_batch(q0, rng.standard_normal((n, dim)))
    dirs = np.where(rng.random(n) < 0.5, 1, -1).astype(np.int8)
    desc = f"softabs_user banana D={dim} N={n} coeff={coeff} h={h:.4f} steps={steps}"
    ref = lambda c: orc.implicit_leapfrog_steps(osys, q0[c], p0[c], dirs[c] * h, steps)  # noqa: E731
    return desc, integ, system, osys, q0, p0, dirs, steps, ref, 5e-8


def constrained_case(rng):
    """Linear-equality and sphere manifolds up to D = 256 / C = 8 and D = 1024 / C <= 2 (the wave-per-chain kernels with
    one, four and sixteen coordinates per lane), both density conventions."""
    dim = int(rng.choice([2, 3, 5, 8, 9, 12, 16, 17, 24, 33, 48, 64, 65, 100, 130, 200, 256, 300, 700, 1024]))
    n = int(rng.choice([1, 3, 7]))
    mk, metric = metric_of(dim, rng)
    pt, ot = targets(dim, rng, ["poly"])
    if rng.random() < 0.6 and dim >= 3:
        c = int(rng.integers(1, min(8 if dim <= 256 else 2, dim - 1) + 1))
        a, b = rng.standard_normal((c, dim)), rng.standard_normal(c)
        pc, oc = models.LinearConstr(a, b), omdl.LinearConstr(a, b)
        part = np.linalg.lstsq(a, b, rcond=None)[0]
        null = np.linalg.svd(a)[2][c:].T
        q0 = part + 0.5 * rng.standard_normal((n, dim - c)) @ null.T
        what = f"linear C={c}"
    else:
        pc, oc = models.SphereConstr(), omdl.SphereConstr()
        x = rng.standard_normal((n, dim))
        q0 = x / np.linalg.norm(x, axis=1, keepdims=True)
        what = "sphere"
    hausdorff = bool(rng.random() < 0.7)
    system = systems.DenseConstrainedEuclideanMetricSystem(pt, pc, metric=metric, dens_wrt_hausdorff=hausdorff)
    osys = orc.ConstrainedSystem(ot, oc, mk, metric, dens_wrt_hausdorff=hausdorff)
    solver = int(rng.integers(0, 3))
    proj = [solvers.solve_projection_onto_manifold_newton, solvers.solve_projection_onto_manifold_quasi_newton,
            solvers.solve_projection_onto_manifold_newton_with_line_search][solver]
    h, steps = float(rng.uniform(0.03, 0.15)), int(rng.integers(1, 8))
    integ = integrators.ConstrainedLeapfrogIntegrator(system, h, projection_solver=proj)
    p0 = np.stack([osys.project_onto_cotangent_space(osys.msqrt(z), osys.constraint.jacob_constr(q0[c]))
                   for c, z in enumerate(rng.standard_normal((n, dim)))])
    dirs = np.where(rng.random(n) < 0.5, 1, -1).astype(np.int8)
    desc = (f"constrained {what} D={dim} N={n} metric={mk} solver={solver} hausdorff={int(hausdorff)} h={h:.3f} "
            f"steps={steps}")
    ref = lambda c: orc.constrained_leapfrog_steps(osys, q0[c], p0[c], dirs[c] * h, steps, proj_solver=solver)  # noqa: E731
    return desc, integ, system, osys, q0, p0, dirs, steps, ref, 1e-9


MAKERS = None


def oracle_is_sensitive(ref, q0, p0, c, device_outcome, rng):
    """A chain on which the device and the oracle END differently (status / completed steps): is that the oracle's own
    last-bit sensitivity?  The oracle is re-run on the chain with its inputs moved by a few parts in 1e16 (what separates
    a momentum drawn on the device from one drawn by the oracle) up to 1e-13; returns the size of the change at which a run
    ends the way the device did (False if none does).
    Such a chain sits on a non-convergent iteration whose outcome no implementation pins."""
    q_keep, p_keep = q0[c].copy(), p0[c].copy()
    try:
        # three sizes of input change: the last bits (a momentum drawn on the device against one drawn by the oracle), 1e-14
        # (what a refined solve M(x)^-1 p differs from a factorised one by) and 1e-13 (ten of those, accumulated)
        for scale in (4e-16, 1e-14, 1e-13):
            for _ in range(6):
                q0[c] = q_keep * (1.0 + scale * rng.integers(-2, 3, size=q_keep.shape))
                p0[c] = p_keep * (1.0 + scale * rng.integers(-2, 3, size=p_keep.shape))
                _, _, so, no = ref(c)
                if (int(so), int(no)) == device_outcome:
                    return scale
        return False
    finally:
        q0[c], p0[c] = q_keep, p_keep


def run_cases(seed, cases, kinds, stress=1.0, long=False, only=-1, out=print, sensitivity=False):
    """The sweep as a function (tests/test_gpu_fuzz_slice.py runs fixed-seed slices of it under `pytest -m gpu`).
    Returns (list of mismatching case records, number of cases with chains that stopped early).  sensitivity=True: a
    chain whose OUTCOME differs is first put to oracle_is_sensitive(); if the oracle itself ends either way it is
    reported as `sensitive` in the record list (key `sensitive`: True) instead of as a mismatch of the implementations."""
    global STEP_FACTOR, H_FACTOR, MAKERS
    H_FACTOR = stress
    STEP_FACTOR = 4 if long else 1
    rng = np.random.default_rng(seed)
    MAKERS = {"euclid": euclid_case, "riemann": riemann_case, "softabs": softabs_case, "constrained": constrained_case,
              "riemann_user": riemann_user_case, "softabs_user": softabs_user_case}
    makers = [MAKERS[k] for k in kinds.split(",")]
    bad, n_failed = [], 0
    for i in range(cases):
        make = makers[int(rng.integers(0, len(makers)))]
        desc, integ, system, osys, q0, p0, dirs, steps, ref, tol = make(rng)
        if only >= 0 and i != only:
            rng.integers(0, len(q0))  # (the draw of the compared chain below)
            continue
        try:
            q, p, status, n_done = integ.step_batch(q0, p0, dirs, n_steps=steps)
        except Exception as e:  # unsupported sizes must fail loudly, never silently
            out(f"[{i}] {desc}: device refused ({type(e).__name__}: {str(e)[:80]})")
            continue
        ok, detail, sens = True, [], False
        for c in sorted(set([0, len(q0) - 1, int(rng.integers(0, len(q0)))])):
            qo, po, so, no = ref(c)
            if so != status[c] or no != n_done[c] or not close(q[c], qo, tol) or not close(p[c], po, tol):
                flips_at = sensitivity and (so != status[c] or no != n_done[c]) and oracle_is_sensitive(
                    ref, q0, p0, c, (int(status[c]), int(n_done[c])), np.random.default_rng([seed, i, c]))
                if flips_at:
                    sens = True
                    out(f"    chain {c}: status {status[c]} vs {so}, n_done {n_done[c]} vs {no}: the oracle ends either "
                        f"way under a {flips_at:.0e} change of its inputs")
                    continue
                ok = False
                err = np.max(np.abs(np.nan_to_num(q[c]) - np.nan_to_num(qo)))
                detail.append(dict(chain=c, status=(int(status[c]), int(so)), n_done=(int(n_done[c]), int(no)), err=float(err)))
                out(f"    chain {c}: status {status[c]} vs {so}, n_done {n_done[c]} vs {no}, max |dq| = {err:.2e}")
        failed = int(np.count_nonzero(status))
        n_failed += failed > 0
        out(f"[{i}] {desc}: {'ok' if ok else 'MISMATCH'}" + (f" ({failed} of {len(status)} chains stopped early: {sorted(set(status[status != 0].tolist()))})" if failed else ""))
        if not ok:
            bad.append(dict(case=i, desc=desc, chains=detail))
        elif sens:
            bad.append(dict(case=i, desc=desc, chains=[], sensitive=True))
    return bad, n_failed


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--long", action="store_true", help="SoftAbs cases integrate four times as many steps")
    ap.add_argument("--stress", type=float, default=1.0, help="multiply the Riemannian step sizes (failing solves)")
    ap.add_argument("--only", type=int, default=-1, help="run this case of the sequence only (the draws of the others are still made)")
    ap.add_argument("--kinds", default="euclid,riemann,softabs,constrained",
                    help="comma-separated case families to draw from (uniformly)")
    a = ap.parse_args()
    bad, n_failed = run_cases(a.seed, a.cases, a.kinds, a.stress, a.long, a.only)
    bad = [b for b in bad if not b.get("sensitive")]
    print(f"{a.cases} cases, {len(bad)} mismatches" + (f" ({n_failed} cases with chains that stopped early, statuses equal)" if n_failed else ""))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
