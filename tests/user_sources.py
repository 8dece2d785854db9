"""HIP source of the user-defined model parts the GPU tests hand to the library (compiled there with hipRTC).
Each text is the device-side form of a NumPy twin in oracle/models.py, which the reference itself ran when the
fixtures were recorded (tools/gen_golden.py)."""

# oracle/models.py EllipsoidSaddleConstr: c_0 = sum_i a_i q_i^2 - 1, c_1 = q_0 q_1 - q_2 + kappa q_3^3; params = a[dim], kappa
ELLIPSOID_SADDLE = r"""
__device__ void mm_user_constr(const double* q, int dim, const double* params, double* c) {
  double s = 0.0;
  for (int i = 0; i < dim; ++i) s += params[i] * (q[i] * q[i]);
  c[0] = s - 1.0;
  c[1] = q[0] * q[1] - q[2] + params[dim] * (q[3] * q[3] * q[3]);
}
__device__ void mm_user_jacob(const double* q, int dim, const double* params, double* jac) {
  for (int i = 0; i < dim; ++i) {
    jac[i] = 2.0 * params[i] * q[i];
    jac[dim + i] = 0.0;
  }
  jac[dim + 0] = q[1];
  jac[dim + 1] = q[0];
  jac[dim + 2] = -1.0;
  jac[dim + 3] = 3.0 * params[dim] * (q[3] * q[3]);
}
#ifdef MM_USER_HAS_MHP
__device__ void mm_user_mhp_constr(const double* q, int dim, const double* params, const double* m, double* out) {
  for (int i = 0; i < dim; ++i) out[i] = 2.0 * params[i] * m[i];
  out[0] += m[dim + 1];
  out[1] += m[dim + 0];
  out[3] += 6.0 * params[dim] * q[3] * m[dim + 3];
}
#endif
"""

# the built-in torus constraint (csrc/constrained_core.h constr_value / constr_jacob) written as user source with the
# library's own arithmetic: must reproduce the built-in kernels
TORUS_AS_USER = r"""
__device__ void mm_user_constr(const double* q, int dim, const double* params, double* c) {
  double rho, irho;
  mmdev::sqrt_rsqrt(q[0] * q[0] + q[1] * q[1], &rho, &irho);
  const double dr = rho - params[0];
  c[0] = dr * dr + q[2] * q[2] - params[1] * params[1];
}
__device__ void mm_user_jacob(const double* q, int dim, const double* params, double* jac) {
  double rho, irho;
  mmdev::sqrt_rsqrt(q[0] * q[0] + q[1] * q[1], &rho, &irho);
  const double f = 2.0 * (rho - params[0]) * irho;
  jac[0] = f * q[0];
  jac[1] = f * q[1];
  jac[2] = 2.0 * q[2];
}
"""

# the built-in rank-one metric M(q) = B + q q^T / D (csrc/implicit_wave.h build_metric / half_vjp_*) as user source;
# params = B[dim*dim] row-major.  vjp(V)_k = sum_ij V_ij (delta_ik q_j + q_i delta_jk) / D = ((V + V^T) q)_k / D
RANK1_AS_USER = r"""
__device__ double mm_user_metric(const double* q, int i, int j, int dim, const double* params) {
  return params[i * dim + j] + (q[i] * q[j]) * (1.0 / (double)dim);
}
__device__ double mm_user_vjp(const double* q, const MmMat& V, int k, int dim, const double* params) {
  double s = 0.0;
  for (int j = 0; j < dim; ++j) s += (V(k, j) + V(j, k)) * q[j];
  return s / (double)dim;
}
"""

# oracle/models.py SoftPlusDiagPlusRank1Metric (not built in): M(q) = diag(1 + log(1 + exp(q_i))) + c c^T * (1 + |q|^2 / D)
# params = c[dim]
SOFTPLUS_RANK1 = r"""
__device__ double mm_user_metric(const double* q, int i, int j, int dim, const double* params) {
  double s2 = 0.0;
  for (int k = 0; k < dim; ++k) s2 += q[k] * q[k];
  double v = params[i] * params[j] * (1.0 + s2 / (double)dim);
  if (i == j) v += 1.0 + log1p(exp(q[i]));
  return v;
}
__device__ double mm_user_vjp(const double* q, const MmMat& V, int k, int dim, const double* params) {
  // d M_ij / d q_k = delta_ij delta_ik sigmoid(q_k) + c_i c_j 2 q_k / D
  double cvc = 0.0;
  for (int i = 0; i < dim; ++i) {
    double r = 0.0;
    for (int j = 0; j < dim; ++j) r += V(i, j) * params[j];
    cvc += params[i] * r;
  }
  return V(k, k) / (1.0 + exp(-q[k])) + cvc * 2.0 * q[k] / (double)dim;
}
"""

# ---- round 4: the two opt-ins of csrc/user_metric.h (MM_USER_AUX, MM_USER_VJP_FLAT).  The texts live with the package
# (mici_amd/user_examples.py: bench.py measures them as c3_user / c4_general): RANK1_AS_USER_FLAT must reproduce the built-in
# kernels bit for bit; SOFTPLUS_RANK1_FAST is oracle/models.py SoftPlusRank1Metric with both opt-ins.
from mici_amd.user_examples import (BANANA_HESS, RANK1_AS_USER_FLAT, RANK1_AS_USER_LOWRANK,  # noqa: E402,F401
                                    SIN_RANK1_LOWRANK, SOFTPLUS_RANK1_FAST, SOFTPLUS_RANK1_FAST_WIDE)

SOFTPLUS_RANK1_FAST_256 = SOFTPLUS_RANK1_FAST_WIDE


def softplus_fast(dim):
    return SOFTPLUS_RANK1_FAST if dim <= 64 else SOFTPLUS_RANK1_FAST_WIDE

# the built-in scaled-funnel Hessian / matrix-Tressian product (csrc/softabs.h build_hessian / mtp_lds)
# as user source: goes through the DENSE SoftAbs path and must agree with the built-in arrowhead path.  params = w[dim - 1]
FUNNEL_HESS = r"""
__device__ double mm_user_hess(const double* q, int i, int j, int dim, const double* params) {
  const double e = exp(-q[0]);
  double S = 0.0;
  for (int k = 1; k < dim; ++k) S += params[k - 1] * q[k] * q[k];
  const int lo = i < j ? i : j, hi = i < j ? j : i, h1 = hi > 0 ? hi : 1;
  const double wh = params[h1 - 1], xh = q[h1];
  const double vv = 1.0 / 9.0 + 0.5 * e * S, vi = -e * wh * xh, ii = e * wh;
  return hi == 0 ? vv : (lo == 0 ? vi : (lo == hi ? ii : 0.0));
}
__device__ double mm_user_mtp(const double* q, const MmMat& M, int k, int dim, const double* params) {
  const double e = exp(-q[0]), mvv = M(0, 0);
  if (k == 0) {
    double S = 0.0, s2 = 0.0, s3 = 0.0;
    for (int i = 1; i < dim; ++i) {
      const double w = params[i - 1];
      S += w * q[i] * q[i];
      s2 += (M(0, i) + M(i, 0)) * w * q[i];
      s3 += M(i, i) * w;
    }
    return -0.5 * e * S * mvv + e * s2 - e * s3;
  }
  const double w = params[k - 1];
  return e * w * q[k] * mvv - e * w * (M(0, k) + M(k, 0));
}
"""


# the built-in banana target (mm_device.h MM_TARGET_BANANA) as user source: tests/test_gpu_user_target.py
BANANA_SRC = """
__device__ double mm_user_grad(const double* q, int i, int dim, const double* params) {
  double g = -(1.0 - q[i]) / 10.0;
  if (i > 0) g += 2.0 * (q[i] - q[i - 1] * q[i - 1]);
  if (i < dim - 1) g -= 4.0 * q[i] * (q[i + 1] - q[i] * q[i]);
  return g;
}
__device__ double mm_user_nld_term(const double* q, int i, int dim, const double* params) {
  double v = (1.0 - q[i]) * (1.0 - q[i]) / 20.0;
  if (i < dim - 1) {
    const double r = q[i + 1] - q[i] * q[i];
    v += r * r;
  }
  return v;
}
"""

