"""Closed-form synthetic models (NumPy twins of the device built-ins).  TEST INFRASTRUCTURE.

Every model here has a device-side twin in ``mici_amd/csrc`` selected by the
same integer id (``include/mici_amd.h``).  The callables have exactly the shape
the reference's ``System`` constructors expect (``neg_log_dens(q)``,
``grad_neg_log_dens(q)``, ``metric_func(q)``, ``vjp_metric_func(q) -> (v -> vjp)``,
``constr(q)``, ``jacob_constr(q)``; reference ``systems.py:264-366, 1187-1402,
1737-1920, 876-1031``) so that ``tools/gen_golden.py`` can push the *same* model
through the imported reference.  Closed forms follow SURVEY.md Appendix A.
"""

from __future__ import annotations

import numpy as np

# ---- ids shared with include/mici_amd.h -------------------------------------------------
TARGET_GAUSS_ISO = 0
TARGET_GAUSS_DIAG = 1
TARGET_GAUSS_DENSE = 2
TARGET_POLY = 3
TARGET_BANANA = 4
TARGET_FUNNEL = 5
TARGET_TORUS = 6

METRIC_IDENTITY = 0
METRIC_DIAG = 1
METRIC_DENSE = 2

RMETRIC_NONE = 0
RMETRIC_RANK1 = 1
RMETRIC_DIAGQUAD = 2
RMETRIC_SOFTABS = 3

CONSTR_NONE = 0
CONSTR_TORUS = 1
CONSTR_FIRST = 2
CONSTR_CIRCLE = 3
CONSTR_LINEAR = 4
CONSTR_SPHERE_PLANE = 5
CONSTR_SPHERE = 6


# ---- targets ---------------------------------------------------------------------------
class Target:
    """A target density: negative log density and its derivatives."""

    tid = -1

    def __init__(self, dim):
        self.dim = int(dim)

    def params(self):
        return np.zeros(0)

    def neg_log_dens(self, q):
        raise NotImplementedError

    def grad(self, q):
        raise NotImplementedError

    # second / third derivatives are only required for the SoftAbs metric
    def hess(self, q):
        raise NotImplementedError

    def mtp(self, q):
        """Return ``m -> sum_ij m_ij T_ijk`` (reference systems.py:1810-1826)."""
        raise NotImplementedError


class GaussIso(Target):
    tid = TARGET_GAUSS_ISO

    def neg_log_dens(self, q):
        return 0.5 * np.sum(q**2)

    def grad(self, q):
        return q.copy()


class GaussDiag(Target):
    tid = TARGET_GAUSS_DIAG

    def __init__(self, prec):
        prec = np.asarray(prec, dtype=np.float64)
        super().__init__(prec.shape[0])
        self.prec = prec

    def params(self):
        return self.prec

    def neg_log_dens(self, q):
        return 0.5 * np.sum(self.prec * q**2)

    def grad(self, q):
        return self.prec * q


class GaussDense(Target):
    tid = TARGET_GAUSS_DENSE

    def __init__(self, prec):
        prec = np.ascontiguousarray(prec, dtype=np.float64)
        super().__init__(prec.shape[0])
        self.prec = prec

    def params(self):
        return self.prec.ravel()

    def neg_log_dens(self, q):
        return 0.5 * q @ (self.prec @ q)

    def grad(self, q):
        return self.prec @ q


class Poly(Target):
    """l(q) = a*sum(q^2)/2 + b*sum(q^4)/4 (covers the reference's quadratic / quartic
    test targets, ``tests/test_integrators.py:236-253, 492-500``)."""

    tid = TARGET_POLY

    def __init__(self, dim, a, b):
        super().__init__(dim)
        self.a, self.b = float(a), float(b)

    def params(self):
        return np.array([self.a, self.b])

    def neg_log_dens(self, q):
        return 0.5 * self.a * np.sum(q**2) + 0.25 * self.b * np.sum(q**4)

    def grad(self, q):
        return self.a * q + self.b * q**3

    def hess(self, q):
        return np.diag(self.a + 3.0 * self.b * q**2)

    def mtp(self, q):
        return lambda m: 6.0 * self.b * q * np.diagonal(m)


class Banana(Target):
    tid = TARGET_BANANA

    def neg_log_dens(self, q):
        r = q[1:] - q[:-1] ** 2
        return np.sum((1.0 - q) ** 2) / 20.0 + np.sum(r**2)

    def grad(self, q):
        r = q[1:] - q[:-1] ** 2
        g = -(1.0 - q) / 10.0
        g[1:] += 2.0 * r
        g[:-1] -= 4.0 * q[:-1] * r
        return g

    # SoftAbs system on the banana (BASELINE c3 "banana/funnel"; systems.py:1870-1920): a TRIDIAGONAL Hessian, i.e. a
    # dense eigenproblem with none of the arrowhead / diagonal structure of the other built-in Hessians - it reaches
    # the device as user source (tests/user_sources.py BANANA_HESS).
    #   H_ii = 1/10 + 2 [i > 0] + (12 q_i^2 - 4 q_{i+1}) [i < D-1],   H_{i,i+1} = -4 q_i
    #   T_iii = 24 q_i [i < D-1],   T_{i,i,i+1} (and permutations) = -4
    def hess(self, q):
        d = q.shape[0]
        h = np.zeros((d, d))
        idx = np.arange(d)
        h[idx, idx] = 0.1
        h[idx[1:], idx[1:]] += 2.0
        h[idx[:-1], idx[:-1]] += 12.0 * q[:-1] ** 2 - 4.0 * q[1:]
        h[idx[:-1], idx[1:]] = -4.0 * q[:-1]
        h[idx[1:], idx[:-1]] = -4.0 * q[:-1]
        return h

    def mtp(self, q):
        def vjp(m):
            dg = np.diagonal(m)
            out = np.zeros_like(q)
            out[:-1] += 24.0 * q[:-1] * dg[:-1] - 4.0 * (np.diagonal(m, 1) + np.diagonal(m, -1))
            out[1:] += -4.0 * dg[:-1]
            return out
        return vjp


class Funnel(Target):
    """Scaled funnel, q = (v, x_1..x_n): l = v^2/18 + n v/2 + exp(-v)/2 * sum(w x^2)."""

    tid = TARGET_FUNNEL

    def __init__(self, w):
        w = np.asarray(w, dtype=np.float64)
        super().__init__(w.shape[0] + 1)
        self.w = w

    def params(self):
        return self.w

    def neg_log_dens(self, q):
        v, x = q[0], q[1:]
        n = x.shape[0]
        return v * v / 18.0 + 0.5 * n * v + 0.5 * np.exp(-v) * np.sum(self.w * x * x)

    def grad(self, q):
        v, x = q[0], q[1:]
        n = x.shape[0]
        e = np.exp(-v)
        s = np.sum(self.w * x * x)
        g = np.empty_like(q)
        g[0] = v / 9.0 + 0.5 * n - 0.5 * e * s
        g[1:] = e * self.w * x
        return g

    def hess(self, q):
        v, x = q[0], q[1:]
        e = np.exp(-v)
        s = np.sum(self.w * x * x)
        d = q.shape[0]
        h = np.zeros((d, d))
        h[0, 0] = 1.0 / 9.0 + 0.5 * e * s
        h[0, 1:] = h[1:, 0] = -e * self.w * x
        h[np.arange(1, d), np.arange(1, d)] = e * self.w
        return h

    def mtp(self, q):
        v, x = q[0], q[1:]
        e = np.exp(-v)
        s = np.sum(self.w * x * x)
        w = self.w

        def apply(m):
            out = np.empty_like(q)
            off = m[0, 1:] + m[1:, 0]
            out[0] = -0.5 * e * s * m[0, 0] + e * np.sum(off * w * x) - e * np.sum(
                np.diagonal(m)[1:] * w
            )
            out[1:] = e * w * x * m[0, 0] - e * w * off
            return out

        return apply


class Torus(Target):
    """README torus density (reference README.md:315-337), D=3."""

    tid = TARGET_TORUS

    def __init__(self, R=1.0, r=0.5, alpha=0.9):
        super().__init__(3)
        self.R, self.r, self.alpha = float(R), float(r), float(alpha)

    def params(self):
        return np.array([self.R, self.r, self.alpha])

    def _angles(self, q):
        x, y, z = q
        rho = np.sqrt(x * x + y * y)
        theta = np.arctan2(y, x)
        phi = np.arctan2(z, rho - self.R)
        return rho, theta, phi

    def neg_log_dens(self, q):
        _, theta, phi = self._angles(q)
        return np.log1p(self.r * np.cos(phi) / self.R) - np.log1p(
            np.sin(4 * theta) * np.cos(phi) * self.alpha
        )

    def grad(self, q):
        x, y, z = q
        R, r, a = self.R, self.r, self.alpha
        rho, theta, phi = self._angles(q)
        s4, c4 = np.sin(4 * theta), np.cos(4 * theta)
        sp, cp = np.sin(phi), np.cos(phi)
        d1 = 1.0 + r * cp / R
        d2 = 1.0 + a * s4 * cp
        dl_dphi = -(r / R) * sp / d1 + a * s4 * sp / d2
        dl_dtheta = -4.0 * a * c4 * cp / d2
        s2 = (rho - R) ** 2 + z * z
        dphi_drho = -z / s2
        dphi_dz = (rho - R) / s2
        g = np.empty(3)
        g[0] = dl_dtheta * (-y / rho**2) + dl_dphi * dphi_drho * (x / rho)
        g[1] = dl_dtheta * (x / rho**2) + dl_dphi * dphi_drho * (y / rho)
        g[2] = dl_dphi * dphi_dz
        return g


# ---- position-dependent (Riemannian) metric functions ---------------------------------------
class Rank1Metric:
    """M(q) = B + q q^T / D;  vjp(V) = (V + V^T) q / D  (SURVEY.md Appendix A)."""

    mid = RMETRIC_RANK1

    def __init__(self, base):
        self.base = np.ascontiguousarray(base, dtype=np.float64)
        self.dim = self.base.shape[0]

    def params(self):
        return self.base.ravel()

    def metric_func(self, q):
        return self.base + np.outer(q, q) / self.dim

    def vjp_metric_func(self, q):
        return lambda v: (v + v.T) @ q / self.dim


RMETRIC_USER = 100


class SoftPlusRank1Metric:
    """A metric that is NOT built into the device library (it reaches it as user HIP source, tests/user_sources.py):
    M(q) = diag(1 + softplus(q_i)) + c c^T (1 + |q|^2 / D);
    vjp(V)_k = V_kk sigmoid(q_k) + (c^T V c) 2 q_k / D."""

    mid = RMETRIC_USER

    def __init__(self, c):
        self.c = np.asarray(c, dtype=np.float64)
        self.dim = self.c.shape[0]

    def params(self):
        return self.c.copy()

    def metric_func(self, q):
        return np.diag(1.0 + np.log1p(np.exp(q))) + np.outer(self.c, self.c) * (1.0 + q @ q / self.dim)

    def vjp_metric_func(self, q):
        return lambda v: np.diagonal(v) / (1.0 + np.exp(-q)) + (self.c @ v @ self.c) * 2.0 * q / self.dim


RMETRIC_USER_SIN = 102  # (fixture id of SinRank1Metric below: the device library knows it as user source only)


class SinRank1Metric:
    """A constant matrix plus a rank-one term in a NONLINEAR vector function of the position - the structure a user metric
    declares with MM_USER_LOWRANK (csrc/user_metric.h): M(q) = B + s u(q) u(q)^T, u_i(q) = q_i + sin(q_i) / 2, s = 2 / D;
    vjp(V)_k = s ((V + V^T) u)_k (1 + cos(q_k) / 2).  params: B row-major."""

    mid = RMETRIC_USER_SIN

    def __init__(self, base):
        self.base = np.ascontiguousarray(base, dtype=np.float64)
        self.dim = self.base.shape[0]

    def params(self):
        return self.base.ravel()

    def u(self, q):
        return q + 0.5 * np.sin(q)

    def metric_func(self, q):
        u = self.u(q)
        return self.base + np.outer(u, u) * (2.0 / self.dim)

    def vjp_metric_func(self, q):
        u = self.u(q)
        return lambda v: (2.0 / self.dim) * ((v + v.T) @ u) * (1.0 + 0.5 * np.cos(q))


class DiagQuadMetric:
    """M(q) = diag(1 + q^2) held as a dense matrix; vjp(V)_i = 2 q_i V_ii
    (dense twin of the reference's DiagonalRiemannian test system,
    ``tests/test_integrators.py:492-500``)."""

    mid = RMETRIC_DIAGQUAD

    def __init__(self, dim):
        self.dim = int(dim)

    def params(self):
        return np.zeros(0)

    def metric_func(self, q):
        return np.diag(1.0 + q**2)

    def vjp_metric_func(self, q):
        return lambda v: 2.0 * q * np.diagonal(v)


# ---- constraints (C = 1) -----------------------------------------------------------------
class TorusConstr:
    cid = CONSTR_TORUS

    def __init__(self, R=1.0, r=0.5):
        self.R, self.r = float(R), float(r)

    def params(self):
        return np.array([self.R, self.r])

    def constr(self, q):
        x, y, z = q
        return np.array([(np.sqrt(x * x + y * y) - self.R) ** 2 + z * z - self.r**2])

    def jacob_constr(self, q):
        x, y, z = q
        rho = np.sqrt(x * x + y * y)
        f = 2.0 * (rho - self.R) / rho
        return np.array([[f * x, f * y, 2.0 * z]])

    def mhp_constr(self, q):
        """m[C, D] -> sum_{c,i} m[c, i] d2 constr_c / dq_i dq_k (the MatrixHessianProduct of systems.py:1006-1008)."""
        x, y, z = q
        rho = np.sqrt(x * x + y * y)
        dr = rho - self.R
        hess = np.array([[2 * x * x / rho**2 + 2 * dr * y * y / rho**3, 2 * x * y * self.R / rho**3, 0.0],
                         [2 * x * y * self.R / rho**3, 2 * y * y / rho**2 + 2 * dr * x * x / rho**3, 0.0],
                         [0.0, 0.0, 2.0]])
        return lambda m: hess @ m[0]


class FirstCoordConstr:
    cid = CONSTR_FIRST

    def params(self):
        return np.zeros(0)

    def constr(self, q):
        return q[:1].copy()

    def jacob_constr(self, q):
        return np.eye(1, q.shape[0], 0)

    def mhp_constr(self, q):
        return lambda m: np.zeros_like(q)


class CircleConstr:
    cid = CONSTR_CIRCLE

    def params(self):
        return np.zeros(0)

    def constr(self, q):
        return q[0:1] ** 2 + q[1:2] ** 2 - 1.0

    def jacob_constr(self, q):
        j = np.zeros((1, q.shape[0]))
        j[0, 0] = 2.0 * q[0]
        j[0, 1] = 2.0 * q[1]
        return j

    def mhp_constr(self, q):
        def mhp(m):
            out = np.zeros_like(q)
            out[0], out[1] = 2.0 * m[0, 0], 2.0 * m[0, 1]
            return out
        return mhp


class LinearConstr:
    """c(q) = A q - b, A of shape [C, D]."""
    cid = CONSTR_LINEAR

    def __init__(self, a, b=None):
        self.a = np.atleast_2d(np.asarray(a, dtype=np.float64))
        self.b = np.zeros(self.a.shape[0]) if b is None else np.asarray(b, dtype=np.float64)

    def params(self):
        return np.concatenate([self.a.ravel(), self.b])

    def constr(self, q):
        return self.a @ q - self.b

    def jacob_constr(self, q):
        return self.a.copy()

    def mhp_constr(self, q):
        return lambda m: np.zeros_like(q)


class SphereConstr:
    """c(q) = |q|^2 - 1 (tests/test_adapters.py:174-181)."""
    cid = CONSTR_SPHERE

    def params(self):
        return np.zeros(0)

    def constr(self, q):
        return np.array([q @ q - 1.0])

    def jacob_constr(self, q):
        return 2.0 * q[None]

    def mhp_constr(self, q):
        return lambda m: 2.0 * m[0]


class SpherePlaneConstr:
    """c_0 = |q|^2 - 1, c_1 = n . q."""
    cid = CONSTR_SPHERE_PLANE

    def __init__(self, normal):
        self.normal = np.asarray(normal, dtype=np.float64)

    def params(self):
        return self.normal.copy()

    def constr(self, q):
        return np.array([q @ q - 1.0, self.normal @ q])

    def jacob_constr(self, q):
        return np.stack([2.0 * q, self.normal])

    def mhp_constr(self, q):
        return lambda m: 2.0 * m[0]


# ---- synthetic parameter generators -----------------------------------------
def make_spd(dim, rng):
    """P = A A^T / D + I with A ~ N(0,1)^{DxD}."""
    a = rng.standard_normal((dim, dim))
    return a @ a.T / dim + np.eye(dim)


CONSTR_USER = 100


class EllipsoidSaddleConstr:
    """Two constraints that are NOT built into the device library (they reach it as user HIP source, tests/
    test_gpu_user_target.py):  c_0 = sum_i a_i q_i^2 - 1,  c_1 = q_0 q_1 - q_2 + kappa q_3^3   (dim >= 4)."""
    cid = CONSTR_USER
    n_constr = 2

    def __init__(self, a, kappa=0.3):
        self.a = np.asarray(a, dtype=np.float64)
        self.kappa = float(kappa)

    def params(self):
        return np.concatenate([self.a, [self.kappa]])

    def constr(self, q):
        return np.array([self.a @ (q * q) - 1.0, q[0] * q[1] - q[2] + self.kappa * q[3] ** 3])

    def jacob_constr(self, q):
        j = np.zeros((2, q.shape[0]))
        j[0] = 2.0 * self.a * q
        j[1, 0], j[1, 1], j[1, 2], j[1, 3] = q[1], q[0], -1.0, 3.0 * self.kappa * q[3] ** 2
        return j

    def mhp_constr(self, q):
        def mhp(m):
            out = 2.0 * self.a * m[0]
            out[0] += m[1, 1]
            out[1] += m[1, 0]
            out[3] += 6.0 * self.kappa * q[3] * m[1, 3]
            return out
        return mhp

    def init(self, n, rng):
        """Points on the manifold: Gauss-Newton projection of random points."""
        d = self.a.shape[0]
        out = np.empty((n, d))
        for c in range(n):
            q = rng.standard_normal(d) / np.sqrt(self.a.sum())
            for _ in range(100):
                j, r = self.jacob_constr(q), self.constr(q)
                if np.max(np.abs(r)) < 1e-14:
                    break
                q = q - j.T @ np.linalg.solve(j @ j.T, r)
            out[c] = q
        return out


def torus_init(n, rng, R=1.0, r=0.5):
    """Initial positions on the torus from (theta, phi) ~ U(0, 2pi) (README.md:344-355)."""
    theta, phi = rng.uniform(0, 2 * np.pi, size=(2, n))
    return np.stack(
        [
            (R + r * np.cos(phi)) * np.cos(theta),
            (R + r * np.cos(phi)) * np.sin(theta),
            r * np.sin(phi),
        ],
        -1,
    )


# ---- rebuild a model from (id, params) as stored in the golden fixtures ------------------------------
def target_from_id(tid, params, dim):
    tid = int(tid)
    params = np.asarray(params, dtype=np.float64)
    if tid == TARGET_GAUSS_ISO:
        return GaussIso(dim)
    if tid == TARGET_GAUSS_DIAG:
        return GaussDiag(params)
    if tid == TARGET_GAUSS_DENSE:
        return GaussDense(params.reshape(dim, dim))
    if tid == TARGET_POLY:
        return Poly(dim, params[0], params[1])
    if tid == TARGET_BANANA:
        return Banana(dim)
    if tid == TARGET_FUNNEL:
        return Funnel(params)
    if tid == TARGET_TORUS:
        return Torus(*params)
    raise ValueError(f"unknown target id {tid}")


def rmetric_from_id(mid, params, dim):
    mid = int(mid)
    if mid == RMETRIC_RANK1:
        return Rank1Metric(np.asarray(params).reshape(dim, dim))
    if mid == RMETRIC_DIAGQUAD:
        return DiagQuadMetric(dim)
    if mid == RMETRIC_SOFTABS:
        return None
    if mid == RMETRIC_USER:  # a metric of the fixtures that the device library only knows as user source
        return SoftPlusRank1Metric(params)
    if mid == RMETRIC_USER_SIN:  # ... and one that declares its constant + rank-one structure there (MM_USER_LOWRANK)
        return SinRank1Metric(np.asarray(params).reshape(dim, dim))
    raise ValueError(f"unknown Riemannian metric id {mid}")


def constr_from_id(cid, params, dim=None):
    cid = int(cid)
    if cid == CONSTR_USER:  # the one constraint of the fixtures that the device library only knows as user source
        params = np.asarray(params, dtype=np.float64)
        return EllipsoidSaddleConstr(params[:-1], params[-1])
    if cid == CONSTR_LINEAR:
        params = np.asarray(params, dtype=np.float64)
        c = params.size // (dim + 1)
        return LinearConstr(params[:c * dim].reshape(c, dim), params[c * dim:])
    if cid == CONSTR_SPHERE_PLANE:
        return SpherePlaneConstr(params)
    if cid == CONSTR_SPHERE:
        return SphereConstr()
    if cid == CONSTR_TORUS:
        return TorusConstr(*params)
    if cid == CONSTR_FIRST:
        return FirstCoordConstr()
    if cid == CONSTR_CIRCLE:
        return CircleConstr()
    raise ValueError(f"unknown constraint id {cid}")
