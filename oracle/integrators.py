"""NumPy/SciPy restatement of the reference's integrator hot path.  TEST INFRASTRUCTURE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; ``mici_amd`` (the product) never does.

What is restated (citations into /root/reference/src/mici):
  * ``LeapfrogIntegrator._step``                         integrators.py:170-173
  * ``ImplicitLeapfrogIntegrator`` sub-steps a/b/c + adjoints  integrators.py:493-544
  * ``ConstrainedLeapfrogIntegrator`` a/b + retraction      integrators.py:929-984
  * ``solve_fixed_point_direct`` / ``_steffensen``         solvers.py:47-94, 97-154
  * ``solve_projection_onto_manifold_newton``              solvers.py:429-469
  * ``maximum_norm`` / ``euclidean_norm``                  solvers.py:20-27
  * Euclidean / Riemannian / constrained system derivatives  systems.py:143-152, 348-366,
                                                           786-873, 1010-1022, 1378-1402
  * dense positive-definite matrix: Cholesky, explicit inverse, log-det, gradient terms
                                                           matrices.py:1161-1188, 897-938, 1060-1061
  * SoftAbs eigen-regularised matrix                       matrices.py:1631-1685
  * state cache semantics (what is recomputed when)          states.py:37-157, 248-279

Status codes (shared with include/mici_amd.h): 0 ok; 1 ConvergenceError (diverged);
2 ConvergenceError (max iterations); 3 ConvergenceError (ValueError/LinAlgError inside a
solver); 4 NonReversibleStepError; 5 LinAlgError raised outside a solver.

Parity is PINNED: tools/gen_golden.py checks every function here against the imported
reference (same inputs, same iteration counts, <=1e-12 relative) and writes the fixtures
that tests/test_oracle_golden.py replays where the reference is absent.
"""

from __future__ import annotations

import numpy as np
import scipy.linalg as sla

from . import models as mdl

ST_OK = 0
ST_DIVERGED = 1
ST_MAX_ITERS = 2
ST_SOLVER_LINALG = 3
ST_NON_REVERSIBLE = 4
ST_LINALG = 5


# ---- error types (hierarchy of reference errors.py:6-35) --------------------------------------
class IntegratorError(RuntimeError):
    status = -1


class ConvergenceError(IntegratorError):
    def __init__(self, msg, status=ST_MAX_ITERS, iteration=-1):
        super().__init__(msg)
        self.status = status
        self.iteration = iteration


class NonReversibleStepError(IntegratorError):
    status = ST_NON_REVERSIBLE


class LinAlgError(RuntimeError):
    status = ST_LINALG


# ---- norms (solvers.py:20-27) ----------------------------------------------------------------
def maximum_norm(v):
    return np.abs(v).max()


def euclidean_norm(v):
    return (v**2).sum() ** 0.5


NORMS = {0: maximum_norm, 1: euclidean_norm, "linf": maximum_norm, "l2": euclidean_norm}


# ---- fixed point solvers (solvers.py:47-94, 97-154) ----------------------------------------------
class Counters(dict):
    def bump(self, key, n=1):
        self[key] = self.get(key, 0) + n


def solve_fixed_point_direct(
    func, x0, convergence_tol=1e-9, divergence_tol=1e10, max_iters=100, norm=maximum_norm,
    counters=None,
):
    error = np.nan
    i = -1
    try:
        for i in range(max_iters):
            if counters is not None:
                counters.bump("fp_iters")  # counts function evaluations
            x = func(x0)
            error = norm(x - x0)
            if error > divergence_tol or np.isnan(error):
                raise ConvergenceError(
                    f"Fixed point iteration diverged on iteration {i}.", ST_DIVERGED, i
                )
            if error < convergence_tol:
                return x
            x0 = x
    except (ValueError, LinAlgError) as e:
        raise ConvergenceError(
            f"{type(e)} at iteration {i} of fixed point solver ({e}).", ST_SOLVER_LINALG, i
        ) from e
    raise ConvergenceError("Fixed point iteration did not converge.", ST_MAX_ITERS, i)


def solve_fixed_point_steffensen(
    func, x0, convergence_tol=1e-9, divergence_tol=1e10, max_iters=100, norm=maximum_norm,
    counters=None,
):
    error = np.nan
    i = -1
    try:
        for i in range(max_iters):
            if counters is not None:
                counters.bump("fp_iters", 2)  # two function evaluations per iteration
            x1 = func(x0)
            x2 = func(x1)
            denom = x2 - 2 * x1 + x0
            denom[abs(denom) == 0.0] = np.finfo(x0.dtype).eps
            x = x0 - (x1 - x0) ** 2 / denom
            error = norm(x - x0)
            if error > divergence_tol or np.isnan(error):
                raise ConvergenceError(
                    f"Fixed point iteration diverged on iteration {i}.", ST_DIVERGED, i
                )
            if error < convergence_tol:
                return x
            x0 = x
    except (ValueError, LinAlgError) as e:
        raise ConvergenceError(
            f"{type(e)} at iteration {i} of fixed point solver ({e}).", ST_SOLVER_LINALG, i
        ) from e
    raise ConvergenceError("Fixed point iteration did not converge.", ST_MAX_ITERS, i)


FP_SOLVERS = {0: solve_fixed_point_direct, 1: solve_fixed_point_steffensen}


# ---- dense matrices (matrices.py) -------------------------------------------------------------
def _chk_finite(a):
    # ExplicitArrayMatrix.__init__, matrices.py:211-215
    if not np.all(np.isfinite(a)):
        raise LinAlgError("Array is not finite.")
    return a


class DensePD:
    """DensePositiveDefiniteMatrix as used on the path (matrices.py:1117-1216)."""

    def __init__(self, array, counters=None):
        self.array = _chk_finite(np.asarray(array, dtype=np.float64))
        self._chol = None
        self._inv = None
        self.counters = counters

    @property
    def chol(self):
        if self._chol is None:
            if self.counters is not None:
                self.counters.bump("chol")
            try:
                self._chol = np.linalg.cholesky(self.array)  # matrices.py:1166
            except np.linalg.LinAlgError as e:
                raise LinAlgError("Cholesky factorisation failed.") from e
        return self._chol

    @property
    def inv(self):
        """Explicit inverse L^-T L^-1 from two triangular solves against the identity
        (matrices.py:1183-1188 -> 976-980 -> 1060-1061 -> 932-938, 897-903)."""
        if self._inv is None:
            if self.counters is not None:
                self.counters.bump("inv")
            lt = self.chol.T
            inv_lt = sla.solve_triangular(lt, np.identity(lt.shape[0]), lower=False,
                                          check_finite=False)
            self._inv = _chk_finite(
                sla.solve_triangular(lt, inv_lt.T, lower=False, check_finite=False)
            )
        return self._inv

    def inv_matvec(self, v):
        return self.inv @ v  # explicit-inverse mat-vec, matrices.py:222-223

    @property
    def log_abs_det(self):
        return 2.0 * np.log(np.abs(np.diagonal(self.chol))).sum()  # :982-984, 850-852

    @property
    def grad_log_abs_det(self):
        return self.inv  # :1175-1177

    def grad_quadratic_form_inv(self, v):
        u = self.inv_matvec(v)
        return -np.outer(u, u)  # :1179-1181

    def sqrt_matvec(self, z):
        return self.chol @ z  # :1215-1216


class SoftAbsPD:
    """SoftAbsRegularizedPositiveDefiniteMatrix (matrices.py:1631-1685)."""

    def __init__(self, sym, coeff, counters=None):
        if counters is not None:
            counters.bump("eigh")
        self.coeff = coeff
        self.lam, self.vec = np.linalg.eigh(sym)  # :1658 (np LinAlgError is a ValueError)
        self.eigval = self.softabs(self.lam)
        if not np.all(self.eigval > 0):  # :1599-1602
            raise ValueError("Eigenvalues must all be positive.")

    def softabs(self, x):
        return x / np.tanh(x * self.coeff)

    def grad_softabs(self, x):
        return 1.0 / np.tanh(self.coeff * x) - self.coeff * x / np.sinh(self.coeff * x) ** 2

    def inv_matvec(self, v):
        return self.vec @ ((1.0 / self.eigval) * (self.vec.T @ v))  # :1568-1575

    @property
    def log_abs_det(self):
        return np.log(np.abs(self.eigval)).sum()

    @property
    def grad_log_abs_det(self):
        g = self.grad_softabs(self.lam) / self.eigval
        return self.vec @ (g[:, None] * (self.vec.T @ np.identity(self.vec.shape[0])))

    def grad_quadratic_form_inv(self, v):
        num = self.eigval[:, None] - self.eigval[None, :]
        num = num + np.diag(self.grad_softabs(self.lam))
        den = self.lam[:, None] - self.lam[None, :]
        np.fill_diagonal(den, 1)
        j = num / den
        e = (self.vec.T @ v) / self.eigval
        return -((self.vec @ (np.outer(e, e) * j)) @ self.vec.T)

    def sqrt_matvec(self, z):
        return self.vec @ (self.eigval**0.5 * (self.vec.T @ z))


# ---- systems ------------------------------------------------------------------------------------
class EuclidSystem:
    """EuclideanMetricSystem with a fixed identity / diagonal / dense metric
    (systems.py:264-366)."""

    def __init__(self, target, metric_kind=mdl.METRIC_IDENTITY, metric=None):
        self.target = target
        self.metric_kind = metric_kind
        self.metric = None if metric is None else np.asarray(metric, dtype=np.float64)
        if metric_kind == mdl.METRIC_DIAG:
            self._inv_diag = 1.0 / self.metric  # DiagonalMatrix inverse, matrices.py:729-733
        elif metric_kind == mdl.METRIC_DENSE:
            self._pd = DensePD(self.metric)
            self._inv = self._pd.inv

    def grad(self, q):
        return self.target.grad(q)

    def minv(self, p):
        if self.metric_kind == mdl.METRIC_IDENTITY:
            return p
        if self.metric_kind == mdl.METRIC_DIAG:
            return self._inv_diag * p
        return self._inv @ p

    def msqrt(self, z):
        if self.metric_kind == mdl.METRIC_IDENTITY:
            return z
        if self.metric_kind == mdl.METRIC_DIAG:
            return self.metric**0.5 * z
        return self._pd.chol @ z

    def h(self, q, p):
        return self.target.neg_log_dens(q) + 0.5 * p @ self.minv(p)

    def h2_flow(self, q, p, dt):
        """pos += dt M^-1 mom (systems.py:362-363); returns the new (q, p)."""
        return q + dt * self.minv(p), p

    def dh_dpos(self, q):
        return self.grad(q)


class GaussianEuclidSystem(EuclidSystem):
    """GaussianEuclideanMetricSystem (systems.py:369-474): the target density is given with respect to the
    standard Gaussian measure, h2 = q^T q / 2 + p^T M^-1 p / 2 and its flow is an exact rotation in the
    metric's eigenbasis (split HMC, Shahbaba et al. 2014)."""

    def __init__(self, target, metric_kind=mdl.METRIC_IDENTITY, metric=None):
        super().__init__(target, metric_kind, metric)
        d = target.dim
        if metric_kind == mdl.METRIC_IDENTITY:
            self.eigval, self.eigvec = np.ones(d), None
        elif metric_kind == mdl.METRIC_DIAG:
            self.eigval, self.eigvec = self.metric, None
        else:
            self.eigval, self.eigvec = np.linalg.eigh(self.metric)  # matrices.py:1207-1213

    def h(self, q, p):
        return self.target.neg_log_dens(q) + 0.5 * q @ q + 0.5 * p @ self.minv(p)

    def h2_flow(self, q, p, dt):
        omega = 1.0 / self.eigval**0.5
        s, c = np.sin(omega * dt), np.cos(omega * dt)
        if self.eigvec is None:
            return c * q + (s * omega) * p, c * p - (s / omega) * q
        a, b = self.eigvec.T @ q, self.eigvec.T @ p
        return self.eigvec @ (c * a + (s * omega) * b), self.eigvec @ (c * b - (s / omega) * a)

    # dh_dpos is deliberately NOT overridden: the reference's EuclideanMetricSystem.dh_dpos returns
    # dh1_dpos alone (systems.py:359-360) and the Gaussian subclass inherits it, so its dh2_dpos = pos
    # (systems.py:461-462) never reaches ImplicitMidpointIntegrator.  Parity follows the reference.


def leapfrog_steps(system, q, p, dt, n_steps):
    """n_steps of LeapfrogIntegrator._step (integrators.py:170-173) for one chain, with
    the end-of-step gradient reused by the next step's first half-kick exactly as the
    state cache does.  dt already includes ``state.dir``."""
    q = np.array(q, dtype=np.float64)
    p = np.array(p, dtype=np.float64)
    g = system.grad(q)
    for _ in range(n_steps):
        p -= (0.5 * dt) * g          # h1_flow(t/2), systems.py:143-152
        q, p = system.h2_flow(q, p, dt)  # h2_flow(t), systems.py:362-363 (exact rotation for the Gaussian split)
        g = system.grad(q)
        p -= (0.5 * dt) * g          # h1_flow(t/2)
    return q, p


def composition_coefficients(free_coefficients):
    """Full coefficient sequence (a_0, b_1, a_1, ..., a_S) of a symmetric composition from its S - 1 free
    coefficients (SymmetricCompositionIntegrator.__init__, integrators.py:258-268): consistency fixes
    the last a and b, symmetry mirrors the rest."""
    free = list(free_coefficients)
    n = len(free)
    coefficients = list(free)
    coefficients.append(0.5 - sum(free[n % 2::2]))
    coefficients.append(1 - 2 * sum(free[(n + 1) % 2::2]))
    return coefficients + coefficients[-2::-1]


BCSS_FREE_COEFFICIENTS = {  # integrators.py:277-378 (Blanes, Casas & Sanz-Serna 2014, eqs 6.4, 6.7, 6.8)
    2: ((3 - 3**0.5) / 6,),
    3: (0.11888010966548, 0.29619504261126),
    4: (0.071353913450279725904, 0.191667800000000000000, 0.268548791161230105820),
}


def composition_steps(system, q, p, dt, n_steps, free_coefficients, initial_h1_flow_step=True):
    """n_steps of SymmetricCompositionIntegrator._step (integrators.py:272-274) for one chain on a
    Euclidean-metric system: alternate h1_flow (mom -= c t grad, systems.py:143-152) and h2_flow
    (pos += c t M^-1 mom, systems.py:362-363) with the coefficient sequence above; the gradient is the
    state cache's, i.e. recomputed only after the position moved."""
    coefficients = composition_coefficients(free_coefficients)
    q = np.array(q, dtype=np.float64)
    p = np.array(p, dtype=np.float64)
    g = system.grad(q)
    for _ in range(n_steps):
        for i, c in enumerate(coefficients):
            if (i % 2 == 0) == bool(initial_h1_flow_step):
                p -= (c * dt) * g
            else:
                q, p = system.h2_flow(q, p, c * dt)
                g = system.grad(q)
    return q, p


def leapfrog_steps_batch(system, q, p, dt, n_steps, coefficients=None, initial_h1_flow_step=True):
    """Vectorised-over-chains variant (rows of q, p are chains) for the cheap targets;
    identical arithmetic per chain.  Used as the multi-chain CPU baseline.  With ``coefficients`` (the full
    sequence from composition_coefficients) it runs that symmetric composition instead of the leapfrog."""
    q = np.array(q, dtype=np.float64)
    p = np.array(p, dtype=np.float64)
    dt = np.broadcast_to(np.asarray(dt, dtype=np.float64), (q.shape[0],))[:, None]
    t = system.target

    def grad(x):
        if t.tid == mdl.TARGET_GAUSS_ISO:
            return x.copy()
        if t.tid == mdl.TARGET_GAUSS_DIAG:
            return t.prec * x
        if t.tid == mdl.TARGET_GAUSS_DENSE:
            return x @ t.prec.T
        if t.tid == mdl.TARGET_POLY:
            return t.a * x + t.b * x**3
        return np.stack([t.grad(r) for r in x])

    def minv(x):
        if system.metric_kind == mdl.METRIC_IDENTITY:
            return x
        if system.metric_kind == mdl.METRIC_DIAG:
            return system._inv_diag * x
        return x @ system._inv.T

    g = grad(q)
    if coefficients is None:
        for _ in range(n_steps):
            p -= (0.5 * dt) * g
            q += dt * minv(p)
            g = grad(q)
            p -= (0.5 * dt) * g
        return q, p
    for _ in range(n_steps):  # SymmetricCompositionIntegrator._step, integrators.py:272-274
        for k, c in enumerate(coefficients):
            if (k % 2 == 0) == initial_h1_flow_step:
                p -= (c * dt) * g
            else:
                q += (c * dt) * minv(p)
                g = grad(q)
    return q, p


class _State:
    """pos/mom plus the pos-keyed cache of the reference ChainState (states.py:160-305).
    Assigning ``pos`` drops pos-dependent entries; ``copy`` shares current entries."""

    __slots__ = ("_pos", "mom", "cache")

    def __init__(self, pos, mom, cache=None):
        self._pos = pos
        self.mom = mom
        self.cache = {} if cache is None else cache

    @property
    def pos(self):
        return self._pos

    @pos.setter
    def pos(self, value):
        self._pos = value
        self.cache = {}

    def copy(self):
        return _State(self._pos.copy(), self.mom.copy(), dict(self.cache))


class RiemannianSystem:
    """DenseRiemannianMetricSystem / SoftAbsRiemannianMetricSystem
    (systems.py:1187-1402, 1690-1734, 1737-1920)."""

    def __init__(self, target, rmetric=None, softabs_coeff=None, counters=None):
        self.target = target
        self.rmetric = rmetric
        self.softabs_coeff = softabs_coeff
        self.counters = counters if counters is not None else Counters()

    # cached-on-pos quantities ------------------------------------------------------------
    def grad(self, st):
        if "grad" not in st.cache:
            self.counters.bump("grad")
            st.cache["grad"] = self.target.grad(st.pos)
        return st.cache["grad"]

    def vjp(self, st):
        if "vjp" not in st.cache:
            self.counters.bump("vjp")
            if self.softabs_coeff is None:
                st.cache["vjp"] = self.rmetric.vjp_metric_func(st.pos)
            else:
                st.cache["vjp"] = self.target.mtp(st.pos)
        return st.cache["vjp"]

    def metric(self, st):
        if "metric" not in st.cache:
            self.counters.bump("metric")
            if self.softabs_coeff is None:
                st.cache["metric"] = DensePD(self.rmetric.metric_func(st.pos), self.counters)
            else:
                st.cache["metric"] = SoftAbsPD(
                    self.target.hess(st.pos), self.softabs_coeff, self.counters
                )
        return st.cache["metric"]

    # Hamiltonian pieces (systems.py:1375-1402) ----------------------------------------------
    def h(self, st):
        m = self.metric(st)
        return (
            self.target.neg_log_dens(st.pos)
            + 0.5 * m.log_abs_det
            + 0.5 * st.mom @ m.inv_matvec(st.mom)
        )

    def dh1_dpos(self, st):
        vjp = self.vjp(st)  # evaluated before the metric, systems.py:1382-1384
        return self.grad(st) + 0.5 * vjp(self.metric(st).grad_log_abs_det)

    def dh2_dpos(self, st):
        vjp = self.vjp(st)
        return 0.5 * vjp(self.metric(st).grad_quadratic_form_inv(st.mom))

    def dh2_dmom(self, st):
        return self.metric(st).inv_matvec(st.mom)

    def sample_momentum(self, st, z):
        return self.metric(st).sqrt_matvec(z)


def implicit_leapfrog_step(
    system, st, dt, fp_solver=solve_fixed_point_direct, rev_tol=2e-8, rev_norm=maximum_norm,
    fp_kwargs=None,
):
    """One ImplicitLeapfrogIntegrator._step on ``st`` in place (integrators.py:493-544).
    NB every sub-map uses the FULL time step (SURVEY.md hazard H1).  Raises the mirrored
    exception types; the caller keeps the last good state."""
    fp_kwargs = dict(fp_kwargs or {})
    fp_kwargs.setdefault("counters", system.counters)

    def step_a(s):
        s.mom = s.mom - dt * system.dh1_dpos(s)  # :493-494, systems.py:143-152

    def step_b_fwd(s, t):  # :496-502
        mom_init = s.mom

        def f(mom):
            s.mom = mom
            return mom_init - t * system.dh2_dpos(s)

        s.mom = fp_solver(f, mom_init, **fp_kwargs)

    def step_c_adj(s, t):  # :530-536
        pos_init = s.pos

        def f(pos):
            s.pos = pos
            return pos_init + t * system.dh2_dmom(s)

        s.pos = fp_solver(f, pos_init, **fp_kwargs)

    def step_c_fwd(s, t):  # :517-528
        pos_init = s.pos.copy()
        s.pos = s.pos + t * system.dh2_dmom(s)
        back = s.copy()
        step_c_adj(back, -t)
        if rev_norm(back.pos - pos_init) > rev_tol:
            raise NonReversibleStepError("Non-reversible step (positions).")

    def step_b_adj(s, t):  # :504-515
        mom_init = s.mom.copy()
        s.mom = s.mom - t * system.dh2_dpos(s)
        back = s.copy()
        step_b_fwd(back, -t)
        if rev_norm(back.mom - mom_init) > rev_tol:
            raise NonReversibleStepError("Non-reversible step (momentums).")

    step_a(st)
    step_b_fwd(st, dt)
    step_c_fwd(st, dt)
    step_c_adj(st, dt)
    step_b_adj(st, dt)
    step_a(st)


class EuclidAsGeneralSystem:
    """A plain EuclideanMetricSystem seen through the generic System interface the implicit leapfrog uses
    (systems.py:132-141, 352-360): dh1_dpos = grad, dh2_dpos = 0, dh2_dmom = M^-1 mom.  The reference runs
    ImplicitLeapfrogIntegrator on such systems in its own tests (tests/test_integrators.py:435-462)."""

    def __init__(self, euclid_system):
        self.inner = euclid_system
        self.counters = Counters()

    def dh1_dpos(self, s):
        return self.inner.grad(s.pos)

    def dh2_dpos(self, s):
        return np.zeros_like(s.pos)

    def dh2_dmom(self, s):
        return self.inner.minv(s.mom)

    def h(self, q, p):
        return self.inner.h(q, p)


def implicit_leapfrog_steps(system, q, p, dt, n_steps, **kw):
    """Run up to n_steps; returns (q, p, status, n_done).  On failure the chain is frozen at
    its last successfully completed step (transitions.py:292-295)."""
    st = _State(np.array(q, dtype=np.float64), np.array(p, dtype=np.float64))
    status, n_done = ST_OK, 0
    for _ in range(n_steps):
        trial = st.copy()
        try:
            implicit_leapfrog_step(system, trial, dt, **kw)
        except IntegratorError as e:
            status = e.status
            break
        except LinAlgError:
            status = ST_LINALG
            break
        st = trial
        n_done += 1
    return st.pos, st.mom, status, n_done


# ---- implicit midpoint integrator (integrators.py:547-681) ---------------------------------------------------
def _midpoint_dh(system, q, p):
    """(dh_dmom, dh_dpos) at (q, p) for a Euclidean-metric or Riemannian-metric oracle system
    (System.dh_dmom / dh_dpos, systems.py:198-224: dh1_dpos + dh2_dpos)."""
    if isinstance(system, RiemannianSystem):
        st = _State(q, p)
        return system.dh2_dmom(st), system.dh1_dpos(st) + system.dh2_dpos(st)
    return system.minv(p), system.dh_dpos(q)


def implicit_midpoint_step(system, q, p, dt, fp_solver=solve_fixed_point_direct, rev_tol=2e-8,
                           rev_norm=maximum_norm, fp_kwargs=None):
    """One ImplicitMidpointIntegrator._step (integrators.py:634-681): implicit Euler half step A(t/2)
    solved as a fixed point in the concatenated (pos, mom) vector, explicit Euler half step A*(t/2), then
    the reversibility check (another implicit half step backwards from the new state)."""
    fp_kwargs = dict(fp_kwargs or {})
    counters = getattr(system, "counters", None)
    if counters is not None:
        fp_kwargs.setdefault("counters", counters)
    d = q.shape[0]

    def a_fwd(q0, p0, t):
        x_init = np.concatenate([q0, p0])

        def func(x):
            dq, dp = _midpoint_dh(system, x[:d], x[d:])
            return x_init + np.concatenate([t * dq, -t * dp])

        if counters is not None:
            counters.bump("fp_solves")
        x = fp_solver(func, x_init, **fp_kwargs)
        return x[:d], x[d:]

    half = dt / 2
    q1, p1 = a_fwd(q, p, half)
    dq, dp = _midpoint_dh(system, q1, p1)
    q2, p2 = q1 + half * dq, p1 - half * dp
    qb, pb = a_fwd(q2, p2, -half)
    rev_diff = rev_norm(np.concatenate([qb - q1, pb - p1]))
    if rev_diff > rev_tol:
        raise NonReversibleStepError(f"Non-reversible step. Distance between initial and forward-backward "
                                     f"integrated (pos, mom) pairs = {rev_diff:.1e}.")
    return q2, p2


def implicit_midpoint_steps(system, q, p, dt, n_steps, **kw):
    """Run up to n_steps; returns (q, p, status, n_done), frozen at the last completed step on failure."""
    q = np.array(q, dtype=np.float64)
    p = np.array(p, dtype=np.float64)
    status, n_done = ST_OK, 0
    for _ in range(n_steps):
        try:
            q2, p2 = implicit_midpoint_step(system, q, p, dt, **kw)
        except IntegratorError as e:
            status = e.status
            break
        except LinAlgError:
            status = ST_LINALG
            break
        q, p = q2, p2
        n_done += 1
    return q, p, status, n_done


# ---- constrained system (systems.py:619-873, 876-1031) ----------------------------------------------
class DenseSymmetric:
    """DenseSymmetricMatrix as used on the path (matrices.py:1414-1447, 428-459): eigendecomposition-based
    inverse and log|det|, no definiteness required."""

    def __init__(self, array):
        self.array = _chk_finite(np.asarray(array, dtype=np.float64))
        self._eig = None

    @property
    def eig(self):
        if self._eig is None:
            try:
                self._eig = np.linalg.eigh(self.array)
            except np.linalg.LinAlgError as e:
                raise LinAlgError(str(e)) from e
        return self._eig

    @property
    def inv(self):
        w, v = self.eig
        return v @ ((1 / w)[:, None] * v.T)

    @property
    def log_abs_det(self):
        return np.log(np.abs(self.eig[0])).sum()


class ConstrainedSystem(EuclidSystem):
    """DenseConstrainedEuclideanMetricSystem (systems.py:876-1031, 619-873).  ``dens_wrt_hausdorff=False``
    adds the half log-determinant of the Gram matrix to h1 (systems.py:846-862, 1024-1031)."""

    def __init__(self, target, constraint, metric_kind=mdl.METRIC_IDENTITY, metric=None,
                 counters=None, dens_wrt_hausdorff=True):
        super().__init__(target, metric_kind, metric)
        self.constraint = constraint
        self.counters = counters if counters is not None else Counters()
        self.dens_wrt_hausdorff = dens_wrt_hausdorff

    def constr(self, q):
        self.counters.bump("constr")
        return self.constraint.constr(q)

    def jacob(self, q):
        self.counters.bump("jacob")
        return self.constraint.jacob_constr(q)

    def minv_mat(self, a):
        """metric.inv @ a for a (D, C) array."""
        if self.metric_kind == mdl.METRIC_IDENTITY:
            return a
        if self.metric_kind == mdl.METRIC_DIAG:
            return self._inv_diag[:, None] * a
        return self._inv @ a

    # ---- the pieces the Gaussian split replaces -------------------------------------------------
    def inner_product(self, a):
        """jacob_constr_inner_product with a single Jacobian (systems.py:1016-1019): Cholesky-factored."""
        return DensePD(a)

    def flow_pos_dmom_mat(self, abs_t, a):
        """dh2_flow_dmom(|t|)[0] @ a for a (D, C) array: |t| M^-1 (systems.py:794-799)."""
        return abs_t * self.minv_mat(a)

    def flow_pos_dmom_vec(self, abs_t, v):
        return abs_t * self.minv(v)

    def flow_mom_dmom_vec(self, abs_t, v):
        """dh2_flow_dmom(|t|)[1] @ v: the identity for a plain Euclidean metric."""
        return v

    # ---- gram-matrix terms ---------------------------------------------------------------------------
    def gram(self, jac):
        return self.inner_product(jac @ self.minv_mat(jac.T))  # systems.py:800-818

    def project_onto_cotangent_space(self, mom, jac):
        # systems.py:863-873;  gram = J M^-1 J^T
        return mom - jac.T @ (self.gram(jac).inv @ (jac @ self.minv(mom)))

    def log_det_sqrt_gram(self, q):
        return 0.5 * self.gram(self.constraint.jacob_constr(q)).log_abs_det  # systems.py:829-831

    def grad_log_det_sqrt_gram(self, q):
        # systems.py:1024-1031: mhp_constr(inv_gram @ jacob_constr @ metric.inv)
        mhp = self.constraint.mhp_constr(q)
        jac = self.constraint.jacob_constr(q)
        return mhp(self.minv_mat((self.gram(jac).inv @ jac).T).T)

    def dh1_dpos(self, q):
        if self.dens_wrt_hausdorff:
            return self.grad(q)
        return self.grad(q) + self.grad_log_det_sqrt_gram(q)  # systems.py:858-862

    def h(self, q, p):
        h = super().h(q, p)
        return h if self.dens_wrt_hausdorff else h + self.log_det_sqrt_gram(q)  # systems.py:853-856


class GaussianConstrainedSystem(ConstrainedSystem):
    """GaussianDenseConstrainedEuclideanMetricSystem (systems.py:1034-1184): the Gaussian split on a
    constrained system.  Always dens_wrt_hausdorff=False; Gram-type matrices are DenseSymmetricMatrix
    (:1148-1161) and dh2_flow_dmom = (V diag(sin(w|t|) w) V^T, V diag(cos(w|t|)) V^T) (:1163-1176)."""

    def __init__(self, target, constraint, metric_kind=mdl.METRIC_IDENTITY, metric=None, counters=None):
        super().__init__(target, constraint, metric_kind, metric, counters, dens_wrt_hausdorff=False)
        d = target.dim
        if metric_kind == mdl.METRIC_IDENTITY:
            self.eigval, self.eigvec = np.ones(d), None
        elif metric_kind == mdl.METRIC_DIAG:
            self.eigval, self.eigvec = self.metric, None
        else:
            self.eigval, self.eigvec = np.linalg.eigh(self.metric)

    h2_flow = GaussianEuclidSystem.h2_flow

    def inner_product(self, a):
        return DenseSymmetric(a)

    def _eig_apply(self, coef, a):
        if self.eigvec is None:
            return (coef * a.T).T
        return self.eigvec @ ((coef * (self.eigvec.T @ a).T).T)

    def flow_pos_dmom_mat(self, abs_t, a):
        omega = 1.0 / self.eigval**0.5
        return self._eig_apply(np.sin(omega * abs_t) * omega, a)

    flow_pos_dmom_vec = flow_pos_dmom_mat

    def flow_mom_dmom_vec(self, abs_t, v):
        omega = 1.0 / self.eigval**0.5
        return self._eig_apply(np.cos(omega * abs_t), v)

    def h(self, q, p):
        return super().h(q, p) + 0.5 * q @ q  # systems.py:451-454


def solve_projection_newton(
    system, q, p, q_prev, jac_prev, t, constraint_tol=1e-9, position_tol=1e-8,
    divergence_tol=1e10, max_iters=50, norm=maximum_norm,
):
    """solve_projection_onto_manifold_newton (solvers.py:429-469) for a Euclidean metric:
    dh2_flow_dmom = (|t| M^-1, I) (systems.py:794-799); residual Jacobian LU-solved
    (matrices.py:1307-1330, 1370-1376).  Returns (q, p)."""
    mu = np.zeros_like(q)
    abs_t = abs(t)
    error = np.nan
    i = -1
    try:
        for i in range(max_iters):
            system.counters.bump("newton_iters")
            jac = system.jacob(q)
            c = system.constr(q)
            error = norm(c)
            a = _chk_finite(jac @ system.flow_pos_dmom_mat(abs_t, jac_prev.T))
            lu, piv = sla.lu_factor(a, check_finite=False)
            delta_mu = jac_prev.T @ sla.lu_solve((lu, piv), c, check_finite=False)
            delta_pos = system.flow_pos_dmom_vec(abs_t, delta_mu)
            if error > divergence_tol or np.isnan(error):
                raise ConvergenceError(f"Newton solver diverged at iteration {i}.",
                                       ST_DIVERGED, i)
            if error < constraint_tol and norm(delta_pos) < position_tol:
                p = p - np.sign(t) * system.flow_mom_dmom_vec(abs_t, mu)
                return q, p
            mu = mu + delta_mu
            q = q - delta_pos
    except (ValueError, LinAlgError) as e:
        raise ConvergenceError(f"{type(e)} at iteration {i} of Newton solver ({e}).",
                               ST_SOLVER_LINALG, i) from e
    raise ConvergenceError(f"Newton solver did not converge in {max_iters} iterations.",
                           ST_MAX_ITERS, i)


def solve_projection_quasi_newton(
    system, q, p, q_prev, jac_prev, t, constraint_tol=1e-9, position_tol=1e-8,
    divergence_tol=1e10, max_iters=50, norm=maximum_norm,
):
    """solve_projection_onto_manifold_quasi_newton (solvers.py:303-343): the Gram matrix
    J_prev (|t| M^-1) J_prev^T is Cholesky-factored ONCE, outside the try block (a failure there is a
    LinAlgError outside the solver); only constr is re-evaluated in the loop."""
    mu = np.zeros_like(q)
    abs_t = abs(t)
    inv_gram = system.inner_product(jac_prev @ system.flow_pos_dmom_mat(abs_t, jac_prev.T)).inv
    error = np.nan
    i = -1
    try:
        for i in range(max_iters):
            system.counters.bump("newton_iters")
            c = system.constr(q)
            error = norm(c)
            delta_mu = jac_prev.T @ (inv_gram @ c)
            delta_pos = system.flow_pos_dmom_vec(abs_t, delta_mu)
            if error > divergence_tol or np.isnan(error):
                raise ConvergenceError(f"Quasi-Newton solver diverged on iteration {i}.",
                                       ST_DIVERGED, i)
            if error < constraint_tol and norm(delta_pos) < position_tol:
                p = p - np.sign(t) * system.flow_mom_dmom_vec(abs_t, mu)
                return q, p, system.jacob(q)
            mu = mu + delta_mu
            q = q - delta_pos
    except (ValueError, LinAlgError) as e:
        raise ConvergenceError(f"{type(e)} at iteration {i} of quasi-Newton solver ({e}).",
                               ST_SOLVER_LINALG, i) from e
    raise ConvergenceError(f"Quasi-Newton solver did not converge with {max_iters} iterations.",
                           ST_MAX_ITERS, i)


def solve_projection_newton_line_search(
    system, q, p, q_prev, jac_prev, t, constraint_tol=1e-9, position_tol=1e-8,
    divergence_tol=1e10, max_iters=50, max_line_search_iters=10, norm=maximum_norm,
):
    """solve_projection_onto_manifold_newton_with_line_search (solvers.py:561-614)."""
    mu = np.zeros_like(q)
    abs_t = abs(t)
    delta_pos, step_size = None, None
    error = np.nan
    for i in range(max_iters):
        try:
            system.counters.bump("newton_iters")
            jac = system.jacob(q)
            c = system.constr(q)
            error = norm(c)
            if i > 0 and (error > divergence_tol or np.isnan(error)):
                raise ConvergenceError(f"Newton solver diverged at iteration {i}.", ST_DIVERGED, i)
            if error < constraint_tol and (i == 0 or norm(step_size * delta_pos) < position_tol):
                p = p - np.sign(t) * system.flow_mom_dmom_vec(abs_t, mu)
                return q, p, jac
            a = _chk_finite(jac @ system.flow_pos_dmom_mat(abs_t, jac_prev.T))
            lu, piv = sla.lu_factor(a, check_finite=False)
            delta_mu = jac_prev.T @ sla.lu_solve((lu, piv), c, check_finite=False)
            delta_pos = -system.flow_pos_dmom_vec(abs_t, delta_mu)
            pos_curr = q.copy()
            step_size = 1.0
            for _ in range(max_line_search_iters):
                q = pos_curr + step_size * delta_pos
                new_error = norm(system.constr(q))
                if new_error < error:
                    break
                step_size *= 0.5
            mu = mu + step_size * delta_mu
        except (ValueError, LinAlgError) as e:
            raise ConvergenceError(f"{type(e)} at iteration {i} of Newton solver ({e}).",
                                   ST_SOLVER_LINALG, i) from e
    raise ConvergenceError(f"Newton solver did not converge in {max_iters} iterations.",
                           ST_MAX_ITERS, i)


def _newton3(*a, **k):
    q, p = solve_projection_newton(*a, **k)
    return q, p, None


PROJ_SOLVERS = {0: _newton3, 1: solve_projection_quasi_newton, 2: solve_projection_newton_line_search}


def constrained_leapfrog_step(system, q, p, dt, n_inner_step=1, rev_tol=2e-8,
                              rev_norm=maximum_norm, proj_kwargs=None, grad_jac=None, proj_solver=0):
    """One ConstrainedLeapfrogIntegrator._step (integrators.py:929-984).  ``grad_jac`` carries
    the cached (gradient, Jacobian) at q from the previous step.  Returns
    (q, p, (grad, jac)) or raises."""
    proj_kwargs = proj_kwargs or {}
    solve = PROJ_SOLVERS[proj_solver]
    if grad_jac is None:
        grad_jac = (system.dh1_dpos(q), system.jacob(q))
    g, jac = grad_jac

    # A(t/2): h1_flow then cotangent projection (:947-949)
    p = p - (0.5 * dt) * g
    p = system.project_onto_cotangent_space(p, jac)
    # B(t): n_inner_step retractions (:951-979)
    t_in = dt / n_inner_step
    for i in range(n_inner_step):
        q_prev, jac_prev = q, jac
        q_new, p = system.h2_flow(q, p, t_in)  # systems.py:362-363 (exact rotation for the Gaussian split)
        q_new, p, _ = solve(system, q_new, p, q_prev, jac_prev, t_in, **proj_kwargs)
        jac_new = system.jacob(q_new)
        if i == n_inner_step - 1:
            g = system.dh1_dpos(q_new)  # pre-evaluated dh1_dpos, :956-969
        p = system.project_onto_cotangent_space(p, jac_new)
        # reversibility check (:971-979)
        q_back, p_back = system.h2_flow(q_new, p, -t_in)
        q_back, _, _ = solve(system, q_back, p_back.copy(), q_new, jac_new, -t_in, **proj_kwargs)
        if rev_norm(q_back - q_prev) > rev_tol:
            raise NonReversibleStepError("Non-reversible step (positions).")
        q, jac = q_new, jac_new
    # A(t/2)
    p = p - (0.5 * dt) * g
    p = system.project_onto_cotangent_space(p, jac)
    return q, p, (g, jac)


def constrained_leapfrog_steps(system, q, p, dt, n_steps, **kw):
    q = np.array(q, dtype=np.float64)
    p = np.array(p, dtype=np.float64)
    status, n_done, gj = ST_OK, 0, None
    for _ in range(n_steps):
        try:
            q2, p2, gj2 = constrained_leapfrog_step(system, q, p, dt, grad_jac=gj, **kw)
        except IntegratorError as e:
            status = e.status
            break
        except LinAlgError:
            status = ST_LINALG
            break
        q, p, gj = q2, p2, gj2
        n_done += 1
    return q, p, status, n_done
