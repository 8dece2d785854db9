"""Examples of USER device code for ``models.UserMetric`` (the device-side form of the reference's ``metric_func`` /
``vjp_metric_func`` constructor arguments, systems.py:1322-1358): the two texts ``bench.py`` measures as ``c3_user`` /
``c4_general`` and the GPU tests compile (tests/user_sources.py re-exports them).  Both use the two opt-ins of
csrc/user_metric.h where they apply - per-point precomputation (``MM_USER_AUX``) and the team-form vector-Jacobian
product (``MM_USER_VJP_FLAT``) - and are written without control flow in ``mm_user_metric``."""

# M(q) = B + q q^T / D, params = B[dim*dim] row-major - the library's built-in rank-one-update metric
# with the library's own arithmetic in every hook: the entry as fma(q_i, q_j / D, B_ij), the vector-Jacobian product
# vjp(V) = (V + V^T) q / D = 2 V q / D on the backend's mat-vec (of which the kernels take one half: V q / D).
RANK1_AS_USER_FLAT = r"""
#define MM_USER_VJP_FLAT
__device__ double mm_user_metric(const double* q, int i, int j, int dim, const double* params) {
  return __builtin_fma(q[i], q[j] * (1.0 / (double)dim), params[i * dim + j]);
}
template <class Ops>
__device__ double mm_user_vjp_flat(Ops& V, const double* q, int k, int dim, const double* params, const double* aux) {
  return 2.0 * (V.matvec(V.active() ? q[k] : 0.0) * (1.0 / (double)dim));
}
"""

# The same metric DECLARING its structure (csrc/user_metric.h MM_USER_LOWRANK, round 6): M(q) = C + s u(q) u(q)^T with C = B,
# u(q) = q, 1 / s = D.  The kernels with a Woodbury path (32 < D <= 1024) then take the position solves' M(x)^-1 p from the held
# inverse and carry the inverse from step to step by a rank-two update - bench.py c4_user_lowrank.
RANK1_AS_USER_LOWRANK = "#define MM_USER_LOWRANK\n" + RANK1_AS_USER_FLAT + r"""
__device__ double mm_user_lowrank_u(const double* q, int i, int dim, const double* params) { return q[i]; }
__device__ double mm_user_lowrank_inv_s(int dim, const double* params) { return (double)dim; }
"""

# M(q) = B + (2 / D) u(q) u(q)^T with u_i(q) = q_i + sin(q_i) / 2 (oracle/models.py SinRank1Metric): constant + rank one in a
# NONLINEAR function of the position, declared.  Per point: u_i and 1 + cos(q_i) / 2 (aux[i], aux[dim + i]);
# vjp(V)_k = (2 / D) 2 (V u)_k (1 + cos(q_k) / 2) on the backend's mat-vec.
SIN_RANK1_LOWRANK = r"""
#define MM_USER_AUX 560
#define MM_USER_VJP_FLAT
#define MM_USER_LOWRANK
template <class Team>
__device__ void mm_user_prepare(Team& tm, const double* q, int dim, const double* params, double* aux) {
  for (int i = tm.rank(); i < dim; i += tm.size()) {
    aux[i] = q[i] + 0.5 * sin(q[i]);
    aux[dim + i] = 1.0 + 0.5 * cos(q[i]);
  }
}
__device__ double mm_user_metric(const double* q, int i, int j, int dim, const double* params, const double* aux) {
  return __builtin_fma(aux[i], aux[j] * (2.0 / (double)dim), params[i * dim + j]);
}
template <class Ops>
__device__ double mm_user_vjp_flat(Ops& V, const double* q, int k, int dim, const double* params, const double* aux) {
  const bool on = V.active();
  const double uk = on ? aux[k] : 0.0, dk = on ? aux[dim + k] : 0.0;
  return (2.0 / (double)dim) * 2.0 * V.matvec(uk) * dk;
}
__device__ double mm_user_lowrank_u(const double* q, int i, int dim, const double* params, const double* aux) { return aux[i]; }
__device__ double mm_user_lowrank_inv_s(int dim, const double* params) { return 0.5 * (double)dim; }
"""

def sin_rank1_lowrank(dim):
    """SIN_RANK1_LOWRANK with the aux block it needs at this size (2 D doubles of at most 560: D <= 280)."""
    return SIN_RANK1_LOWRANK.replace("#define MM_USER_AUX 560", f"#define MM_USER_AUX {2 * int(dim)}")


# M(q) = diag(1 + softplus(q_i)) + c c^T (1 + |q|^2 / D), params = c[dim] (oracle/models.py SoftPlusRank1Metric - not built
# into the device library).  Per point: |q|^2 by one team reduction (aux[0]), softplus(q_i) (aux[1 + i]) and the parameters
# themselves (aux[1 + dim + i]: an entry then reads LDS only - at D <= 64 a lone wave cannot hide a global load); the
# vector-Jacobian product V_kk sigmoid(q_k) + (c^T V c) 2 q_k / D through one mat-vec and one team sum.
SOFTPLUS_RANK1_FAST = r"""
#define MM_USER_AUX 130
#define MM_USER_VJP_FLAT
template <class Team>
__device__ void mm_user_prepare(Team& tm, const double* q, int dim, const double* params, double* aux) {
  double s2 = 0.0;
  for (int i = tm.rank(); i < dim; i += tm.size()) {
    s2 += q[i] * q[i];
    aux[1 + i] = 1.0 + log1p(exp(q[i]));
    aux[1 + dim + i] = params[i];
  }
  s2 = tm.sum(s2);
  if (tm.rank() == 0) aux[0] = 1.0 + s2 / (double)dim;
}
__device__ double mm_user_metric(const double* q, int i, int j, int dim, const double* params, const double* aux) {
  const double v = aux[1 + dim + i] * aux[1 + dim + j] * aux[0], d = aux[1 + i];  // (d unconditionally: no branch)
  return i == j ? v + d : v;
}
template <class Ops>
__device__ double mm_user_vjp_flat(Ops& V, const double* q, int k, int dim, const double* params, const double* aux) {
  const bool on = V.active();
  const double ck = on ? aux[1 + dim + k] : 0.0, qk = on ? q[k] : 0.0;
  const double cvc = V.sum(ck * V.matvec(ck));
  return V.diag() / (1.0 + exp(-qk)) + cvc * 2.0 * qk / (double)dim;
}
"""

# the same metric for every size its aux block (2 D + 2 of at most 560 doubles) allows: dim <= 279
SOFTPLUS_RANK1_FAST_WIDE = SOFTPLUS_RANK1_FAST.replace("#define MM_USER_AUX 130", "#define MM_USER_AUX 560")

# Hessian and matrix-Tressian product of the built-in banana target (oracle/models.py Banana.hess / .mtp) for
# SoftAbsRiemannianMetricSystem(models.Banana(D), hess_neg_log_dens=models.UserHessian(BANANA_HESS)): a TRIDIAGONAL Hessian,
# i.e. a dense eigenproblem with none of the structure of the built-in device Hessians (bench.py c3b_dense).
#   H_ii = 1/10 + 2 [i > 0] + (12 q_i^2 - 4 q_{i+1}) [i < D-1],   H_{i,i+1} = -4 q_i
#   mtp(M)_k = 24 q_k M_kk [k < D-1] - 4 M_{k-1,k-1} [k > 0] - 4 (M_{k,k+1} + M_{k+1,k}) [k < D-1]
BANANA_HESS = r"""
__device__ double mm_user_hess(const double* q, int i, int j, int dim, const double* params) {
  const int lo = i < j ? i : j, hi = i < j ? j : i, nx = lo + 1 < dim ? lo + 1 : lo;   // clamped: reads stay inside q
  const double ql = q[lo], qn = q[nx];
  const double diag = 0.1 + (lo > 0 ? 2.0 : 0.0) + (lo < dim - 1 ? 12.0 * ql * ql - 4.0 * qn : 0.0);
  const double off = hi == lo + 1 ? -4.0 * ql : 0.0;
  return hi == lo ? diag : off;
}
__device__ double mm_user_mtp(const double* q, const MmMat& M, int k, int dim, const double* params) {
  const int kp = k + 1 < dim ? k + 1 : k, km = k > 0 ? k - 1 : 0;   // clamped: every read is inside the matrix
  const double a = 24.0 * q[k] * M(k, k) - 4.0 * (M(k, kp) + M(kp, k));
  const double b = -4.0 * M(km, km);
  return (k < dim - 1 ? a : 0.0) + (k > 0 ? b : 0.0);
}
"""
