"""Built-in synthetic models: host-side *descriptors* only (ids + packed fp64 parameters).

The reference takes arbitrary Python callables for ``neg_log_dens`` / ``metric_func`` / ``constr``
(systems.py:107,119,788,792,1332,1358); a HIP kernel needs device-side closed forms, so the targets,
position-dependent metrics and constraints of SURVEY.md section 8d are an enum implemented in
``csrc/mm_device.h`` and selected by the ids below (``include/mici_amd.h``).  No arithmetic on chain
state happens in this module."""

from __future__ import annotations

import numpy as np

TARGET_GAUSS_ISO, TARGET_GAUSS_DIAG, TARGET_GAUSS_DENSE, TARGET_POLY = 0, 1, 2, 3
TARGET_BANANA, TARGET_FUNNEL, TARGET_TORUS = 4, 5, 6
TARGET_USER = 100
METRIC_IDENTITY, METRIC_DIAG, METRIC_DENSE = 0, 1, 2
RMETRIC_NONE, RMETRIC_RANK1, RMETRIC_DIAGQUAD, RMETRIC_SOFTABS = 0, 1, 2, 3
CONSTR_NONE, CONSTR_TORUS, CONSTR_FIRST, CONSTR_CIRCLE, CONSTR_LINEAR, CONSTR_SPHERE_PLANE, CONSTR_SPHERE = 0, 1, 2, 3, 4, 5, 6
CONSTR_USER = 100


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class Target:
    """Descriptor of a built-in target density (negative log density + device derivatives)."""

    def __init__(self, tid, dim, params=()):
        self.tid = int(tid)
        self.dim = int(dim)
        self.params = _f64(params).ravel()

    def __repr__(self):
        return f"{type(self).__name__}(dim={self.dim})"


class UserTarget(Target):
    """A target defined by the USER as HIP device code - the device-side form of the reference's ``neg_log_dens`` /
    ``grad_neg_log_dens`` constructor arguments (systems.py:107, 119).  ``source`` must define

        __device__ double mm_user_grad(const double* q, int i, int dim, const double* params);      // d nld / d q_i
        __device__ double mm_user_nld_term(const double* q, int i, int dim, const double* params);  // nld = sum_i

    and is compiled for gfx950 (hipRTC) when the system's device model is created; ``params`` are handed to both
    functions.  Works with ``EuclideanMetricSystem`` (identity / diagonal / dense metric) and the explicit
    integrators (leapfrog, symmetric compositions)."""

    def __init__(self, dim, source, params=()):
        super().__init__(TARGET_USER, dim, params)
        if not isinstance(source, str) or "mm_user_grad" not in source or "mm_user_nld_term" not in source:
            raise ValueError("source must define mm_user_grad and mm_user_nld_term (see the class docstring)")
        self.source = source


class GaussIso(Target):
    """l(q) = |q|^2 / 2."""

    def __init__(self, dim):
        super().__init__(TARGET_GAUSS_ISO, dim)


class GaussDiag(Target):
    """l(q) = sum_i prec_i q_i^2 / 2."""

    def __init__(self, prec):
        prec = _f64(prec)
        if prec.ndim != 1:
            raise ValueError("prec must be 1D")
        super().__init__(TARGET_GAUSS_DIAG, prec.shape[0], prec)


class GaussDense(Target):
    """l(q) = q^T P q / 2 with a dense symmetric precision P."""

    def __init__(self, prec):
        prec = _f64(prec)
        if prec.ndim != 2 or prec.shape[0] != prec.shape[1]:
            raise ValueError("prec must be a square 2D array")
        super().__init__(TARGET_GAUSS_DENSE, prec.shape[0], prec)


class Poly(Target):
    """l(q) = a sum(q^2)/2 + b sum(q^4)/4."""

    def __init__(self, dim, a, b):
        super().__init__(TARGET_POLY, dim, [a, b])


class Banana(Target):
    """l(q) = sum (1-q_i)^2/20 + sum (q_{i+1} - q_i^2)^2."""

    def __init__(self, dim):
        super().__init__(TARGET_BANANA, dim)


class Funnel(Target):
    """Scaled funnel over q = (v, x): v^2/18 + n v/2 + exp(-v) sum(w x^2)/2, w distinct."""

    def __init__(self, w):
        w = _f64(w)
        super().__init__(TARGET_FUNNEL, w.shape[0] + 1, w)


class Torus(Target):
    """Density on the README torus (reference README.md:315-337); dim 3."""

    def __init__(self, R=1.0, r=0.5, alpha=0.9):
        super().__init__(TARGET_TORUS, 3, [R, r, alpha])


class RiemannianMetric:
    def __init__(self, mid, dim, params=()):
        self.mid = int(mid)
        self.dim = int(dim)
        self.params = _f64(params).ravel()


RMETRIC_USER = 100


class UserMetric(RiemannianMetric):
    """A position-dependent metric defined by the USER as HIP device code - the device-side form of the reference's
    ``metric_func`` / ``vjp_metric_func`` constructor arguments (systems.py:1322-1358).  ``source`` must define

        __device__ double mm_user_metric(const double* q, int i, int j, int dim, const double* params);  // M(q)_ij
        __device__ double mm_user_vjp(const double* q, const MmMat& V, int k, int dim, const double* params);
        // element k of vjp_metric_func(q)(V) = sum_ij V(i, j) d M_ij / d q_k for a symmetric V read as V(i, j)

    and is compiled for gfx950 (hipRTC) with the library's dense-Riemannian kernels (the auxiliary kernels when the
    system's device model is created, the matrix-core leapfrog kernels on their first launch); ``params`` are handed to
    both.  ``dim`` <= 1024 (beyond 279 on the global-memory tier; the aux opt-in is 560 doubles at most, and ``MM_USER_AUX`` must be a plain decimal literal - the library reads it from the text, expressions are rejected).  The text may opt into per-point precomputation (``#define MM_USER_AUX n`` +
    ``mm_user_prepare``) and the team-form vector-Jacobian product (``#define MM_USER_VJP_FLAT`` +
    ``mm_user_vjp_flat``): csrc/user_metric.h - both decide how fast the system runs, neither changes results.  A metric of
    the form C + s u(q) u(q)^T with a constant matrix C may DECLARE it (``#define MM_USER_LOWRANK`` + ``mm_user_lowrank_u`` /
    ``mm_user_lowrank_inv_s``, round 6): at 32 < dim <= 1024 the implicit leapfrog then takes the position solves' M(x)^-1 p from
    the explicit inverse at the step's start by the Woodbury identity and carries that inverse from step to step by a rank-two
    update - the path of the built-in ``Rank1Metric`` (mici_amd/user_examples.py RANK1_AS_USER_LOWRANK,
    SIN_RANK1_LOWRANK)."""

    def __init__(self, dim, source, params=()):
        super().__init__(RMETRIC_USER, dim, params)
        if not isinstance(source, str) or "mm_user_metric" not in source or "mm_user_vjp" not in source:
            raise ValueError("source must define mm_user_metric and mm_user_vjp (see the class docstring)")
        self.source = source


RMETRIC_SOFTABS_USER = 101


class UserHessian:
    """The Hessian of the target and its matrix-Tressian product defined by the USER as HIP device code - the device-side
    form of the reference's ``hess_neg_log_dens`` / ``mtp_neg_log_dens`` constructor arguments of
    ``SoftAbsRiemannianMetricSystem`` (systems.py:1737-1920).  ``source`` must define

        __device__ double mm_user_hess(const double* q, int i, int j, int dim, const double* params);   // H(q)_ij
        __device__ double mm_user_mtp(const double* q, const MmMat& M, int k, int dim, const double* params);
        // element k of mtp_neg_log_dens(q)(M) = sum_ij M(i, j) d3 nld / dq_i dq_j dq_k for a symmetric M read as M(i, j)

    and is compiled for gfx950 (hipRTC) with the library's SoftAbs kernels when the system's device model is created
    (csrc/user_hessian.h).  Nothing is assumed about the Hessian's structure.  ``dim`` <= 256 (the kernels are compiled
    for the padded size of the system's dimension: 64, 128 or 256)."""

    def __init__(self, source, params=()):
        if not isinstance(source, str) or "mm_user_hess" not in source or "mm_user_mtp" not in source:
            raise ValueError("source must define mm_user_hess and mm_user_mtp (see the class docstring)")
        self.source = source
        self.params = _f64(params).ravel()


class Rank1Metric(RiemannianMetric):
    """M(q) = B + q q^T / D."""

    def __init__(self, base):
        base = _f64(base)
        if base.ndim != 2 or base.shape[0] != base.shape[1]:
            raise ValueError("base must be a square 2D array")
        super().__init__(RMETRIC_RANK1, base.shape[0], base)


class DiagQuadMetric(RiemannianMetric):
    """M(q) = diag(1 + q^2), held as a dense matrix."""

    def __init__(self, dim):
        super().__init__(RMETRIC_DIAGQUAD, dim)


class Constraint:
    def __init__(self, cid, params=()):
        self.cid = int(cid)
        self.params = _f64(params).ravel()


class UserConstraint(Constraint):
    """A constraint defined by the USER as HIP device code - the device-side form of the reference's ``constr`` /
    ``jacob_constr`` (and, for ``dens_wrt_hausdorff=False`` systems, ``mhp_constr``) constructor arguments
    (systems.py:786-792, 1006-1008).  ``source`` must define

        __device__ void mm_user_constr(const double* q, int dim, const double* params, double* c);    // c[n_constr]
        __device__ void mm_user_jacob(const double* q, int dim, const double* params, double* jac);   // jac[k*dim + i]

    and for ``dens_wrt_hausdorff=False`` / Gaussian-split systems also

        __device__ void mm_user_mhp_constr(const double* q, int dim, const double* params, const double* m, double* out);
        // out[i] = sum_{k,j} m[k*dim + j] d2 c_k / dq_j dq_i

    It is compiled for gfx950 (hipRTC) together with the library's constrained-leapfrog core when the system's
    device model is created; ``params`` are handed to all three.  1 <= n_constr <= 8, n_constr < dim <= 256 (up to 64
    on the lane-per-chain core, beyond on the wave-per-chain kernels: ``jacob_constr`` / ``mhp_constr`` are then run by one
    lane of the chain's wave on arrays in LDS)."""

    def __init__(self, n_constr, source, params=()):
        super().__init__(CONSTR_USER, params)
        if not isinstance(source, str) or "mm_user_constr" not in source or "mm_user_jacob" not in source:
            raise ValueError("source must define mm_user_constr and mm_user_jacob (see the class docstring)")
        self.n_constr = int(n_constr)
        self.source = source


class TorusConstr(Constraint):
    """c(q) = (sqrt(x^2+y^2) - R)^2 + z^2 - r^2."""

    def __init__(self, R=1.0, r=0.5):
        super().__init__(CONSTR_TORUS, [R, r])


class FirstCoordConstr(Constraint):
    """c(q) = q_0."""

    def __init__(self):
        super().__init__(CONSTR_FIRST)


class CircleConstr(Constraint):
    """c(q) = q_0^2 + q_1^2 - 1."""

    def __init__(self):
        super().__init__(CONSTR_CIRCLE)


class LinearConstr(Constraint):
    """c(q) = A q - b with A of shape [C, D], 1 <= C <= 8 rows (C < D)."""

    def __init__(self, a, b=None):
        a = np.atleast_2d(_f64(a))
        b = np.zeros(a.shape[0]) if b is None else _f64(b).ravel()
        if b.shape[0] != a.shape[0]:
            raise ValueError("b must have one entry per row of A")
        super().__init__(CONSTR_LINEAR, np.concatenate([a.ravel(), b]))
        self.n_constr = a.shape[0]


class SphereConstr(Constraint):
    """c(q) = |q|^2 - 1."""

    def __init__(self):
        super().__init__(CONSTR_SPHERE)


class SpherePlaneConstr(Constraint):
    """Two constraints: c_0(q) = |q|^2 - 1 and c_1(q) = n . q (a great circle / sphere of the unit sphere)."""

    def __init__(self, normal):
        super().__init__(CONSTR_SPHERE_PLANE, normal)
        self.n_constr = 2


def target_from_id(tid, params, dim):
    tid = int(tid)
    params = _f64(params)
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
        return Rank1Metric(_f64(params).reshape(dim, dim))
    if mid == RMETRIC_DIAGQUAD:
        return DiagQuadMetric(dim)
    raise ValueError(f"unknown Riemannian metric id {mid}")


def constr_from_id(cid, params, dim=None):
    cid = int(cid)
    if cid == CONSTR_LINEAR:
        params = _f64(params)
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
