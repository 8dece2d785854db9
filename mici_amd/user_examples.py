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
  return 2.0 * (V.matvec(V.active() ? q[k] : 0.0) / (double)dim);
}
"""

# M(q) = diag(1 + softplus(q_i)) + c c^T (1 + |q|^2 / D), params = c[dim] (oracle/models.py SoftPlusRank1Metric - not built
# into the device library).  |q|^2 and softplus(q_i) once per point (aux[0], aux[1 + i]); the vector-Jacobian product
# V_kk sigmoid(q_k) + (c^T V c) 2 q_k / D through one mat-vec and one team sum.
SOFTPLUS_RANK1_FAST = r"""
#define MM_USER_AUX 66
#define MM_USER_VJP_FLAT
__device__ void mm_user_prepare(const double* q, int dim, const double* params, double* aux, int t, int nt) {
  if (t == 0) {
    double s2 = 0.0;
    for (int k = 0; k < dim; ++k) s2 += q[k] * q[k];
    aux[0] = 1.0 + s2 / (double)dim;
  }
  for (int i = t; i < dim; i += nt) aux[1 + i] = 1.0 + log1p(exp(q[i]));
}
__device__ double mm_user_metric(const double* q, int i, int j, int dim, const double* params, const double* aux) {
  const double v = params[i] * params[j] * aux[0], d = aux[1 + i];  // (d loaded unconditionally: no branch, user_metric.h)
  return i == j ? v + d : v;
}
template <class Ops>
__device__ double mm_user_vjp_flat(Ops& V, const double* q, int k, int dim, const double* params, const double* aux) {
  const bool on = V.active();
  const double ck = on ? params[k] : 0.0, qk = on ? q[k] : 0.0;
  const double cvc = V.sum(ck * V.matvec(ck));
  return V.diag() / (1.0 + exp(-qk)) + cvc * 2.0 * qk / (double)dim;
}
"""

# the same metric for every size the library takes (dim <= 279: aux sized for it)
SOFTPLUS_RANK1_FAST_WIDE = SOFTPLUS_RANK1_FAST.replace("#define MM_USER_AUX 66", "#define MM_USER_AUX 288")
