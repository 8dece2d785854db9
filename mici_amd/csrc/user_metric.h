// Hooks of a user-supplied position-dependent metric (MM_RMETRIC_USER): metric_func / vjp_metric_func of
// DenseRiemannianMetricSystem (systems.py:1322-1358, 1690-1734) as device code.  mm_rtc.hip compiles the dense-Riemannian
// backends (implicit_wave.h, implicit_mfma.h, implicit_blk16.h, implicit_team.h) around the user's source at run time; this
// header is what those backends and the user's text agree on.
//
// The user's source defines (q: the chain's whole position vector in natural order, zero beyond dim; params:
// desc->rmetric_params on the device):
//   double mm_user_metric(q, i, j, dim, params [, aux])      entry (i, j) of metric_func(q); must be symmetric
//   double mm_user_vjp(q, V, k, dim, params [, aux])         element k of vjp_metric_func(q)(V) = sum_ij V(i, j) dM_ij/dq_k,
//                                                            V(i, j) an accessor of the symmetric argument (MmMat)
// and may opt into two things that decide how fast its system runs (each by a #define in the text, which the library finds
// and hoists in front of its own headers):
//   #define MM_USER_AUX n                                    n doubles (<= 560) of per-POINT precomputation shared by all
//     template <class Team>                                  entries, in LDS: called once per evaluation point by EVERY
//     void mm_user_prepare(Team& tm, q, dim, params, aux)    thread of the chain's team - tm.rank() of tm.size(); a thread
//                                                            handles i = rank, rank + size, ... and may write any aux[.]
//                                                            (identical values from several threads are fine); tm.sum(x) is
//                                                            a team collective (every thread calls it, idle ones with 0):
//                                                            |q|^2 is one wave reduction, not a 64-step loop.  The metric /
//                                                            vjp hooks then take `aux`.  Put there what every entry needs -
//                                                            sums over q, transcendental functions of q_i, and (D <= 64,
//                                                            where a lone wave cannot hide a global load) the parameters
//                                                            themselves.  Without it an entry that needs sum_k q_k^2
//                                                            recomputes it per entry: D^3 work per construction, and per
//                                                            M(x) v product of the refinement solves.
//   #define MM_USER_VJP_FLAT                                 the vector-Jacobian product in TEAM form:
//     template <class Ops> double mm_user_vjp_flat(Ops& V, q, k, dim, params, aux)
//                                                            called by EVERY thread of the team with its own k (k >= dim:
//                                                            idle threads - they must make the same V.* calls, with zeros;
//                                                            V.active() says which), V offering team collectives on the
//                                                            symmetric argument:  V.matvec(a_k) -> (V a)_k,  V.diag() ->
//                                                            V_kk,  V.sum(x_k) -> sum over k < dim.  The argument is the
//                                                            explicit inverse the backend holds in its register tiles, or
//                                                            the rank-one -u u^T: no dense copy of it is made, the
//                                                            products run on the backend's own mat-vec.
//                                                            Without it the backends dump the inverse to a dense array
//                                                            (LDS on the wave kernels, global memory beyond) for V(i, j).
//   #define MM_USER_LOWRANK                                  (round 6) the metric is a CONSTANT matrix plus a rank-one term in a
//     double mm_user_lowrank_u(q, i, dim, params [, aux])    vector function of the position: M(q) = C + s u(q) u(q)^T with
//     double mm_user_lowrank_inv_s(dim, params)              s > 0 constant - element i of u(q), and 1 / s.  mm_user_metric and
//                                                            the vector-Jacobian product stay as they are (they define the
//                                                            metric; this only DECLARES its structure).  The kernels with a
//                                                            Woodbury path (32 < D <= 1024: DESIGN section 4.3f) then take the
//                                                            position solves' M(x)^-1 p from the explicit inverse at the step's
//                                                            start - one product each instead of a CG refinement - and carry
//                                                            that inverse from step to step by a rank-two update.
// Write mm_user_metric WITHOUT control flow - selects on values that are loaded unconditionally (`d = aux[1 + i]; return
// i == j ? v + d : v;`), not conditional loads (`i == j ? v + aux[1 + i] : v` is a branch): the hook is inlined into loops
// over the register-resident metric tiles, and a branch per entry there made the register allocator keep tiles in scratch
// for the whole step on the block-16 kernel (308 spilled registers against none, measured with tools/rtc_compile_check.py).
#pragma once
#include "mm_device.h"

#if defined(MM_RTC_BUILD) && defined(MM_RTC_USER_METRIC)

// A symmetric D x D matrix handed to the user's vector-Jacobian product: V(i, j).  Either an explicit dense matrix
// (the inverse metric: grad_log_abs_det, matrices.py:1175-1177) or the rank-one -u u^T (grad_quadratic_form_inv,
// matrices.py:1179-1181).
#ifndef MM_MMMAT_DEFINED
#define MM_MMMAT_DEFINED 1
struct MmMat {
  const double* a;  // explicit: a[i * ld + j]; nullptr for the rank-one form
  const double* u;
  int ld;
  __device__ __forceinline__ double operator()(int i, int j) const { return a ? a[i * ld + j] : -(u[i] * u[j]); }
};
#endif

#ifdef MM_USER_AUX
static_assert(MM_USER_AUX >= 1 && MM_USER_AUX <= 560, "MM_USER_AUX: 1 .. 560 doubles");
template <class Team>
__device__ void mm_user_prepare(Team& tm, const double* q, int dim, const double* params, double* aux);
__device__ double mm_user_metric(const double* q, int i, int j, int dim, const double* params, const double* aux);
#else
__device__ double mm_user_metric(const double* q, int i, int j, int dim, const double* params);
#endif
#ifdef MM_USER_VJP_FLAT
template <class Ops>
__device__ double mm_user_vjp_flat(Ops& V, const double* q, int k, int dim, const double* params, const double* aux);
#elif defined(MM_USER_AUX)
__device__ double mm_user_vjp(const double* q, const MmMat& V, int k, int dim, const double* params, const double* aux);
#else
__device__ double mm_user_vjp(const double* q, const MmMat& V, int k, int dim, const double* params);
#endif

#ifdef MM_USER_LOWRANK
#ifdef MM_USER_AUX
__device__ double mm_user_lowrank_u(const double* q, int i, int dim, const double* params, const double* aux);
#else
__device__ double mm_user_lowrank_u(const double* q, int i, int dim, const double* params);
#endif
__device__ double mm_user_lowrank_inv_s(int dim, const double* params);
#endif

#endif  // MM_RTC_BUILD && MM_RTC_USER_METRIC

namespace mmuser {

#if defined(MM_RTC_BUILD) && defined(MM_RTC_USER_METRIC)
#ifdef MM_USER_AUX
constexpr int kAux = MM_USER_AUX;
#else
constexpr int kAux = 0;
#endif
#ifdef MM_USER_VJP_FLAT
constexpr bool kFlatVjp = true;
#else
constexpr bool kFlatVjp = false;
#endif
#ifdef MM_USER_LOWRANK
constexpr bool kLowRank = true;
__device__ __forceinline__ double lowrank_u(const double* q, int i, int dim, const double* params, const double* aux) {
  const int ic = i < dim ? i : dim - 1;  // (idle threads evaluate a valid element; callers mask the result)
#ifdef MM_USER_AUX
  return ::mm_user_lowrank_u(q, ic, dim, params, aux);
#else
  return ::mm_user_lowrank_u(q, ic, dim, params);
#endif
}
__device__ __forceinline__ double lowrank_inv_s(int dim, const double* params) { return ::mm_user_lowrank_inv_s(dim, params); }
#else
constexpr bool kLowRank = false;
__device__ __forceinline__ double lowrank_u(const double*, int, int, const double*, const double*) { return 0.0; }
__device__ __forceinline__ double lowrank_inv_s(int, const double*) { return 1.0; }
#endif
// doubles of LDS a backend sets aside per chain for a user metric: the point q of the held inverse and the point x of the
// refinement products in natural order (zero padded to `np`), and the two aux blocks that belong to them
__host__ __device__ constexpr int lds_doubles(int np) { return 2 * np + 2 * ((kAux + 1) & ~1); }

__device__ __forceinline__ double entry(const double* q, int i, int j, int dim, const double* params, const double* aux) {
#ifdef MM_USER_AUX
  return ::mm_user_metric(q, i, j, dim, params, aux);
#else
  return ::mm_user_metric(q, i, j, dim, params);
#endif
}
// entry (i, j) of the metric zero-padded beyond dim, WITHOUT control flow around the user's code: it is evaluated at clamped
// (always valid) indices and the padding is a select - a branch per entry makes every metric tile register a PHI, which the
// register-starved backends pay for with copies and spills
__device__ __forceinline__ double entry_padded(const double* q, int i, int j, int dim, const double* params,
                                               const double* aux) {
  const int ic = i < dim ? i : dim - 1, jc = j < dim ? j : dim - 1;
  const double v = entry(q, ic, jc, dim, params, aux);
  return (i < dim && j < dim) ? v : 0.0;
}
// every thread of the team calls this BETWEEN two team synchronisations of the caller: q is published before, aux is read
// after.  Team: rank(), size(), sum(x) (a collective of the whole team)
template <class Team>
__device__ __forceinline__ void prepare(Team tm, const double* q, int dim, const double* params, double* aux) {
#ifdef MM_USER_AUX
  ::mm_user_prepare(tm, q, dim, params, aux);
#endif
}
// element k of vjp_metric_func(q)(V) through the dense accessor
__device__ __forceinline__ double vjp_dense(const double* q, const MmMat& V, int k, int dim, const double* params,
                                            const double* aux) {
#ifdef MM_USER_VJP_FLAT
  return 0.0;
#elif defined(MM_USER_AUX)
  return ::mm_user_vjp(q, V, k, dim, params, aux);
#else
  return ::mm_user_vjp(q, V, k, dim, params);
#endif
}
template <class Ops>
__device__ __forceinline__ double vjp_flat(Ops& V, const double* q, int k, int dim, const double* params,
                                           const double* aux) {
#ifdef MM_USER_VJP_FLAT
  return ::mm_user_vjp_flat(V, q, k, dim, params, aux);
#else
  return 0.0;
#endif
}
#else  // the in-tree instantiations never reach a user hook
constexpr int kAux = 0;
constexpr bool kFlatVjp = false;
constexpr bool kLowRank = false;
__device__ __forceinline__ double lowrank_u(const double*, int, int, const double*, const double*) { return 0.0; }
__device__ __forceinline__ double lowrank_inv_s(int, const double*) { return 1.0; }
__host__ __device__ constexpr int lds_doubles(int) { return 0; }
__device__ __forceinline__ double entry(const double*, int, int, int, const double*, const double*) { return 0.0; }
__device__ __forceinline__ double entry_padded(const double*, int, int, int, const double*, const double*) { return 0.0; }
template <class Team>
__device__ __forceinline__ void prepare(Team, const double*, int, const double*, double*) {}
template <class M>
__device__ __forceinline__ double vjp_dense(const double*, const M&, int, int, const double*, const double*) { return 0.0; }
template <class Ops>
__device__ __forceinline__ double vjp_flat(Ops&, const double*, int, int, const double*, const double*) { return 0.0; }
#endif

// one wave as a team (the wave-per-chain backends)
struct WaveTeam {
  int lane;
  __device__ __forceinline__ int rank() const { return lane; }
  __device__ __forceinline__ int size() const { return 64; }
  __device__ __forceinline__ double sum(double x) const { return mmdev::wave_sum(x); }
};

// The team form of the symmetric argument of a vector-Jacobian product.  BK: a backend of implicit_core.h that also has
// diag() (flat diagonal of the held explicit inverse), sum1() and flat_active().
//   Inv:   V = the explicit inverse M(q)^-1 in the backend's register tiles           (grad_log_abs_det)
//   Outer: V = -u u^T with u = M^-1 p flat (element k on thread k, zero beyond dim)    (grad_quadratic_form_inv)
template <class BK>
struct VjpOpsInv {
  BK& bk;
  __device__ __forceinline__ bool active() const { return bk.flat_active(); }
  __device__ __forceinline__ double matvec(double a) { return bk.matvec(bk.flat_active() ? a : 0.0); }
  __device__ __forceinline__ double diag() { return bk.diag(); }
  __device__ __forceinline__ double sum(double x) { return bk.sum1(x); }
};
template <class BK>
struct VjpOpsOuter {
  BK& bk;
  double u;
  __device__ __forceinline__ bool active() const { return bk.flat_active(); }
  __device__ __forceinline__ double matvec(double a) { return -(u * bk.sum1(u * a)); }
  __device__ __forceinline__ double diag() { return -(u * u); }
  __device__ __forceinline__ double sum(double x) { return bk.sum1(x); }
};

}  // namespace mmuser
