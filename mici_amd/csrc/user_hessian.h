// Hooks of a user-supplied Hessian for the SoftAbs system: hess_neg_log_dens / mtp_neg_log_dens of
// SoftAbsRiemannianMetricSystem (systems.py:1737-1920) as device code.  mm_rtc.hip compiles softabs.h around the user's
// source at run time (MM_RMETRIC_SOFTABS_USER); this header is what the backend and the user's text agree on.
//
// The user's source defines (q: the chain's whole position vector in natural order, zero beyond dim; params: what follows
// the SoftAbs coefficient in desc->rmetric_params):
//   double mm_user_hess(q, i, j, dim, params)     entry (i, j) of the Hessian of neg_log_dens at q (symmetric, DENSE:
//                                                 nothing is assumed about its structure - the eigendecomposition and
//                                                 the V f(lambda) V^T forms below run as 64^3 products on the matrix cores)
//   double mm_user_mtp(q, M, k, dim, params)      element k of the matrix-Tressian product mtp_neg_log_dens(q)(M) =
//                                                 sum_ij M(i, j) d^3 nld / dq_i dq_j dq_k, M(i, j) an accessor of the
//                                                 symmetric argument, which the backend forms IN FULL in LDS:
//                                                 grad_log_abs_det = V diag(softabs'(lam) / softabs(lam)) V^T and
//                                                 grad_quadratic_form_inv = -(A J A^T)  (matrices.py:1671-1685)
// The target itself is a built-in one or user code as well (mm_user_grad / mm_user_nld_term, mm_device.h).
// dim <= 256 (round 5; beyond 64 the matrices - M included - live in the chain's workspace in HBM, not in LDS).  As for user
// metrics: no control flow in mm_user_hess where a select will do.
#pragma once
#include "mm_device.h"

#if defined(MM_RTC_BUILD) && !defined(MM_MMMAT_DEFINED)
#define MM_MMMAT_DEFINED 1
// A symmetric D x D matrix handed to a user's vector-Jacobian / matrix-Tressian product: M(i, j).  An explicit dense
// matrix, or the rank-one -u u^T.
struct MmMat {
  const double* a;  // explicit: a[i * ld + j]; nullptr for the rank-one form
  const double* u;
  int ld;
  __device__ __forceinline__ double operator()(int i, int j) const { return a ? a[i * ld + j] : -(u[i] * u[j]); }
};
#endif

#if defined(MM_RTC_BUILD) && defined(MM_RTC_USER_HESSIAN)
__device__ double mm_user_hess(const double* q, int i, int j, int dim, const double* params);
__device__ double mm_user_mtp(const double* q, const MmMat& M, int k, int dim, const double* params);
#endif

namespace mmuserh {
#if defined(MM_RTC_BUILD) && defined(MM_RTC_USER_HESSIAN)
// entry (i, j) zero-padded beyond dim, evaluated at clamped (always valid) indices: no branch around the user's code
__device__ __forceinline__ double hess_padded(const double* q, int i, int j, int dim, const double* params) {
  const int ic = i < dim ? i : dim - 1, jc = j < dim ? j : dim - 1;
  const double v = ::mm_user_hess(q, ic, jc, dim, params);
  return (i < dim && j < dim) ? v : 0.0;
}
__device__ __forceinline__ double mtp(const double* q, const double* m, int ld, int k, int dim, const double* params) {
  const MmMat mm{m, nullptr, ld};
  return ::mm_user_mtp(q, mm, k, dim, params);
}
#else  // the in-tree instantiations never reach a user hook
__device__ __forceinline__ double hess_padded(const double*, int, int, int, const double*) { return 0.0; }
__device__ __forceinline__ double mtp(const double*, const double*, int, int, int, const double*) { return 0.0; }
#endif
}  // namespace mmuserh
