// Constrained leapfrog for 8 < D <= 64 with 1 <= C <= 8 constraints: ONE WAVE PER CHAIN (round 3; VERDICT r02 "missing" #4).
//
// The lane-per-chain core of constrained_core.h keeps every per-chain array in registers up to D = 8; beyond, its padded
// instantiations (k_constrained_wide*.hip) run the same code with the arrays - a C x D Jacobian is up to 4 KB, several
// copies live at once - in 5-26 KB of scratch per lane.  Here a chain belongs to a wave instead: lane i holds
// coordinate i of every D-vector and column i of every C x D Jacobian in registers, D-long sums are wave reductions,
// and the C x C systems (Gram matrix + Cholesky, Newton residual Jacobian + pivoted LU: constrained_core.h's own
// routines) are solved redundantly by every lane.  Control flow is wave-uniform - one chain, one wave - so a failed or
// converged chain simply leaves its loops: no SIMT masking.
//
// Reductions.  A Newton iteration needs the C x C matrix J diag-scaled M^-1 J_prev^T: up to 64 D-long sums at once.  They
// go through LDS as a transposition: lane i writes its K products to prod[k][i] (row stride 65 doubles: conflict
// free), lane k sums row k, and the K results are read back as broadcasts - ~3 LDS instructions per sum and lane where K
// wave reductions would be ~25 VALU / DPP instructions each.
//
// Replaces, per chain and per step (reference /root/reference/src/mici): ConstrainedLeapfrogIntegrator._step*
// integrators.py:929-984; solve_projection_onto_manifold_newton / _quasi_newton / _newton_with_line_search
// solvers.py:429-469, 303-343, 561-614; ConstrainedEuclideanMetricSystem.* systems.py:786-873, both density conventions
// (systems.py:829-862, 1024-1031).  The Gaussian split, user constraints and the Gram log-determinant kernel stay on the
// lane-per-chain path.
#include "constrained_core.h"

using namespace mmcon;
using namespace mmdev;

namespace {

constexpr int kWavesPerBlock = 4;
constexpr int kRowStride = 65;                      // doubles per row of the transposition buffer
constexpr int kWaveLds = 64 * kRowStride + 64 + 64 + 64;  // prod[64][65], sums[64], nat[64], vec[64]

struct WaveCtx {
  double* prod;  // [64][65]
  double* sums;  // [64]
  double* nat;   // [64] a D-vector in natural order (target gradient, dense-metric products)
  double* vec;   // [64] second natural-order vector
  int lane, dim;
};

// K D-long sums at once: in  v[k] = this lane's term of sum k;  out v[k] = sum k over the wave, in every lane.
template <int K>
__device__ __forceinline__ void reduce_many(const WaveCtx& w, double (&v)[K]) {
  static_assert(K >= 1 && K <= 64, "one result per lane");
  if constexpr (K <= 2) {  // one or two sums: DPP reductions are cheaper than the LDS round trip
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = wave_sum(v[k]);
    return;
  }
#pragma unroll
  for (int k = 0; k < K; ++k) w.prod[k * kRowStride + w.lane] = v[k];
  wave_sync();
  if (w.lane < K) {
    const double* row = w.prod + w.lane * kRowStride;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
    for (int j = 0; j < 64; j += 4) {
      a0 += row[j];
      a1 += row[j + 1];
      a2 += row[j + 2];
      a3 += row[j + 3];
    }
    w.sums[w.lane] = (a0 + a1) + (a2 + a3);
  }
  wave_sync();
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = w.sums[k];
  wave_sync();
}

__device__ __forceinline__ double wnorm(double x, int kind) {
  return kind == MM_NORM_LINF ? wave_max(fabs(x)) : sqrt(wave_sum(x * x));
}

// y = M^-1 x, one coordinate per lane.  Dense: M^-1 is symmetric, so lane i walks COLUMN i (coalesced over the lanes
// for every j) with x_j broadcast from LDS.
__device__ __forceinline__ double minv1(const ConArgs& A, const WaveCtx& w, double x) {
  if (A.metric_kind == MM_METRIC_IDENTITY) return x;
  if (A.metric_kind == MM_METRIC_DIAG) return w.lane < w.dim ? A.minv[w.lane] * x : 0.0;
  w.vec[w.lane] = x;
  wave_sync();
  double y = 0.0;
  if (w.lane < w.dim) {
    const double* col = A.minv + w.lane;
    for (int j = 0; j < w.dim; ++j) y = __builtin_fma(col[(int64_t)j * w.dim], w.vec[j], y);
  }
  wave_sync();
  return y;
}

template <int C>
struct Col {  // column i of a C x D matrix: this lane's entry of every row
  double v[C];
};

template <int C>
__device__ __forceinline__ Col<C> minv_rows_w(const ConArgs& A, const WaveCtx& w, const Col<C>& j) {
  Col<C> o;
#pragma unroll
  for (int k = 0; k < C; ++k) o.v[k] = minv1(A, w, j.v[k]);
  return o;
}

// column `lane` of jacob_constr(q)
template <int C>
__device__ __forceinline__ Col<C> jacob_w(const ConArgs& A, const WaveCtx& w, double q) {
  Col<C> j;
  const int i = w.lane;
  const bool in = i < w.dim;
#pragma unroll
  for (int k = 0; k < C; ++k) j.v[k] = 0.0;
  if (A.constr == MM_CONSTR_LINEAR) {
#pragma unroll
    for (int k = 0; k < C; ++k) j.v[k] = in ? A.cparams[k * w.dim + i] : 0.0;
  } else if (A.constr == MM_CONSTR_SPHERE) {
    j.v[0] = 2.0 * q;
  } else if (A.constr == MM_CONSTR_SPHERE_PLANE) {
    j.v[0] = 2.0 * q;
    if constexpr (C > 1) j.v[1] = in ? A.cparams[i] : 0.0;
  } else if (A.constr == MM_CONSTR_CIRCLE) {
    j.v[0] = i < 2 ? 2.0 * q : 0.0;
  } else {  // MM_CONSTR_FIRST
    j.v[0] = i == 0 ? 1.0 : 0.0;
  }
  return j;
}

// constr(q)
template <int C>
__device__ __forceinline__ CVec<C> constr_w(const ConArgs& A, const WaveCtx& w, double q) {
  CVec<C> c;
  const int i = w.lane;
  const bool in = i < w.dim;
  double t[C];
#pragma unroll
  for (int k = 0; k < C; ++k) t[k] = 0.0;
  if (A.constr == MM_CONSTR_LINEAR) {
#pragma unroll
    for (int k = 0; k < C; ++k) t[k] = in ? A.cparams[k * w.dim + i] * q : 0.0;
    reduce_many<C>(w, t);
#pragma unroll
    for (int k = 0; k < C; ++k) c.v[k] = t[k] - A.cparams[C * w.dim + k];
    return c;
  }
  if (A.constr == MM_CONSTR_SPHERE_PLANE) {
    t[0] = q * q;
    if constexpr (C > 1) t[1] = in ? A.cparams[i] * q : 0.0;
    reduce_many<C>(w, t);
    c.v[0] = t[0] - 1.0;
    if constexpr (C > 1) c.v[1] = t[1];
#pragma unroll
    for (int k = 2; k < C; ++k) c.v[k] = 0.0;
    return c;
  }
#pragma unroll
  for (int k = 0; k < C; ++k) c.v[k] = 0.0;
  if (A.constr == MM_CONSTR_SPHERE) c.v[0] = wave_sum(q * q) - 1.0;
  else if (A.constr == MM_CONSTR_CIRCLE) c.v[0] = wave_sum(i < 2 ? q * q : 0.0) - 1.0;
  else c.v[0] = readlane_f64(q, 0);  // MM_CONSTR_FIRST
  return c;
}

// g[a][b] = scale * sum_i x[a]_i y[b]_i
template <int C>
__device__ __forceinline__ CMat<C> rows_inner_w(const WaveCtx& w, const Col<C>& x, const Col<C>& y, double scale) {
  double t[C * C];
#pragma unroll
  for (int a = 0; a < C; ++a)
#pragma unroll
    for (int b = 0; b < C; ++b) t[a * C + b] = x.v[a] * y.v[b];
  reduce_many<C * C>(w, t);
  CMat<C> g;
#pragma unroll
  for (int a = 0; a < C; ++a)
#pragma unroll
    for (int b = 0; b < C; ++b) g.m[a][b] = t[a * C + b] * scale;
  return g;
}

template <int C>
__device__ __forceinline__ double combine(const Col<C>& rows, const CVec<C>& x) {
  double s = rows.v[0] * x.v[0];
#pragma unroll
  for (int b = 1; b < C; ++b) s += rows.v[b] * x.v[b];
  return s;
}

// mom - J^T (J M^-1 J^T)^-1 J M^-1 mom     (systems.py:863-873)
template <int C>
__device__ __forceinline__ bool project_cotangent_w(const ConArgs& A, const WaveCtx& w, double& p, const Col<C>& jac) {
  const CMat<C> gram = rows_inner_w<C>(w, jac, minv_rows_w<C>(A, w, jac), 1.0);
  CMat<C> inv;
  double ld;
  if (!all_finite<C>(gram) || !chol_inverse<C>(gram, &inv, &ld)) return false;
  const double mp = minv1(A, w, p);
  double t[C];
#pragma unroll
  for (int a = 0; a < C; ++a) t[a] = jac.v[a] * mp;
  reduce_many<C>(w, t);
  CVec<C> jm;
#pragma unroll
  for (int a = 0; a < C; ++a) jm.v[a] = t[a];
  p -= combine<C>(jac, cmat_vec<C>(inv, jm));
  return true;
}

// grad_neg_log_dens, one coordinate per lane (the position goes through LDS for targets that couple coordinates)
__device__ __forceinline__ double grad_w(const ConArgs& A, const WaveCtx& w, double q) {
  w.nat[w.lane] = w.lane < w.dim ? q : 0.0;
  wave_sync();
  const TargetAux aux;  // no wave-collective targets here (the funnel is rejected on the host)
  const double g = w.lane < w.dim ? target_grad_elem(A.target, aux, w.nat, w.lane, w.dim, A.tparams) : 0.0;
  wave_sync();
  return g;
}

// dh1_dpos (systems.py:858-862): grad_neg_log_dens, plus for dens_wrt_hausdorff=False
// grad_log_det_sqrt_gram = mhp_constr(inv_gram J M^-1) (systems.py:1024-1031).  false = LinAlgError.
// The built-in constraints' Hessians are constant multiples of (part of) the identity: sum_k m[k] * H_k picks
// 2 m[0] on the coordinates the quadratic constraint involves, nothing for the linear ones.
template <int C>
__device__ __forceinline__ bool dh1_dpos_w(const ConArgs& A, const WaveCtx& w, double q, double* out) {
  double g = grad_w(A, w, q);
  if (A.ambient) {
    const Col<C> jac = jacob_w<C>(A, w, q);
    const CMat<C> gram = rows_inner_w<C>(w, jac, minv_rows_w<C>(A, w, jac), 1.0);
    CMat<C> inv;
    double ld;
    if (!all_finite<C>(gram) || !chol_inverse<C>(gram, &inv, &ld)) return false;
    Col<C> m;  // column `lane` of inv_gram @ J
#pragma unroll
    for (int a = 0; a < C; ++a) {
      double s = inv.m[a][0] * jac.v[0];
#pragma unroll
      for (int b = 1; b < C; ++b) s += inv.m[a][b] * jac.v[b];
      m.v[a] = s;
    }
    const double m0 = minv1(A, w, m.v[0]);  // only row 0 meets a non-zero constraint Hessian
    if (A.constr == MM_CONSTR_SPHERE || A.constr == MM_CONSTR_SPHERE_PLANE) g += 2.0 * m0;
    else if (A.constr == MM_CONSTR_CIRCLE) g += w.lane < 2 ? 2.0 * m0 : 0.0;
  }
  *out = g;
  return true;
}

// The three projection solvers (solvers.py:429-469, 303-343, 561-614) on the lane-distributed state.
template <int C>
__device__ __forceinline__ int project_w(const ConArgs& A, const WaveCtx& w, double& q, double& p, const Col<C>& jac_prev,
                                         double t, Col<C>* jac_out, long long* n_iters) {
  const mm_proj_opts& o = A.opts;
  const double abs_t = fabs(t);
  const double sgn = (t > 0.0) ? 1.0 : ((t < 0.0) ? -1.0 : 0.0);
  const Col<C> mjp = minv_rows_w<C>(A, w, jac_prev);
  double mu = 0.0;
  if (o.solver == MM_PROJ_QUASI_NEWTON) {
    CMat<C> inv;
    double ld;
    const CMat<C> g0 = rows_inner_w<C>(w, jac_prev, mjp, abs_t);
    if (!all_finite<C>(g0) || !chol_inverse<C>(g0, &inv, &ld)) return MM_ST_LINALG;
    for (int it = 0; it < o.max_iters; ++it) {
      *n_iters += 1;
      const CVec<C> c = constr_w<C>(A, w, q);
      const double err = cnorm<C>(c, o.norm);
      const CVec<C> x = cmat_vec<C>(inv, c);
      const double dmu = combine<C>(jac_prev, x);
      const double dpos = abs_t * combine<C>(mjp, x);
      if (err > o.div_tol || err != err) return MM_ST_DIVERGED;
      if (err < o.constr_tol && wnorm(dpos, o.norm) < o.pos_tol) {
        p -= sgn * mu;
        *jac_out = jacob_w<C>(A, w, q);
        return MM_ST_OK;
      }
      mu += dmu;
      q -= dpos;
    }
    return MM_ST_MAX_ITERS;
  }
  if (o.solver == MM_PROJ_NEWTON_LINE_SEARCH) {
    double dpos = 0.0, step = 0.0;
    for (int it = 0; it < o.max_iters; ++it) {
      *n_iters += 1;
      const Col<C> jac = jacob_w<C>(A, w, q);
      const CVec<C> c = constr_w<C>(A, w, q);
      const double err = cnorm<C>(c, o.norm);
      if (it > 0 && (err > o.div_tol || err != err)) return MM_ST_DIVERGED;
      const bool small_step = (it == 0) || wnorm(step * dpos, o.norm) < o.pos_tol;
      if (err < o.constr_tol && small_step) {
        p -= sgn * mu;
        *jac_out = jac;
        return MM_ST_OK;
      }
      const CMat<C> a = rows_inner_w<C>(w, jac, mjp, abs_t);
      if (!all_finite<C>(a)) return MM_ST_SOLVER_LINALG;
      const CVec<C> x = lu_solve<C>(a, c);
      const double dmu = combine<C>(jac_prev, x);
      dpos = -(abs_t * combine<C>(mjp, x));
      const double q_curr = q;
      step = 1.0;
      for (int ls = 0; ls < o.max_line_search_iters; ++ls) {
        q = q_curr + step * dpos;
        const double new_err = cnorm<C>(constr_w<C>(A, w, q), o.norm);
        if (new_err < err) break;
        step *= 0.5;
      }
      mu += step * dmu;
    }
    return MM_ST_MAX_ITERS;
  }
  for (int it = 0; it < o.max_iters; ++it) {  // Newton
    *n_iters += 1;
    const Col<C> jac = jacob_w<C>(A, w, q);
    const CVec<C> c = constr_w<C>(A, w, q);
    const double err = cnorm<C>(c, o.norm);
    const CMat<C> a = rows_inner_w<C>(w, jac, mjp, abs_t);
    if (!all_finite<C>(a)) return MM_ST_SOLVER_LINALG;  // "Array is not finite." inside the solver
    const CVec<C> x = lu_solve<C>(a, c);
    const double dmu = combine<C>(jac_prev, x);
    const double dpos = abs_t * combine<C>(mjp, x);
    if (err > o.div_tol || err != err) return MM_ST_DIVERGED;
    if (err < o.constr_tol && wnorm(dpos, o.norm) < o.pos_tol) {
      p -= sgn * mu;
      *jac_out = jac;
      return MM_ST_OK;
    }
    mu += dmu;
    q -= dpos;
  }
  return MM_ST_MAX_ITERS;
}

template <int C>
__global__ __launch_bounds__(64 * kWavesPerBlock) void constrained_wave_kernel(ConArgs A) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t chain = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (chain >= A.n_chains) return;  // no block-level barrier in this kernel
  double* wl = lds + wave * kWaveLds;
  const WaveCtx w{wl, wl + 64 * kRowStride, wl + 64 * kRowStride + 64, wl + 64 * kRowStride + 128, lane, A.dim};
  const int dim = A.dim;
  const bool in = lane < dim;
  double q = in ? A.pos[chain * dim + lane] : 0.0;
  double p = in ? A.mom[chain * dim + lane] : 0.0;
  const double t = signed_step(A.dir, A.step_scale, chain, A.step_size);
  const int n_inner = A.opts.n_inner;
  const double t_in = t / n_inner;
  long long n_newton = 0, n_grad = 0;
  int status = MM_ST_OK, done = 0;

  double g;  // cached dh1_dpos at the current position
  if (!dh1_dpos_w<C>(A, w, q, &g)) status = MM_ST_LINALG;
  Col<C> jac = jacob_w<C>(A, w, q);
  ++n_grad;
  const int my_steps = chain_steps(A.chain_steps, chain, A.n_steps);
  for (int s = 0; s < my_steps && status == MM_ST_OK; ++s) {
    double qs = q, ps = p, gs = g;
    Col<C> js = jac;
    // ---- A(t/2): h1_flow then cotangent projection                    integrators.py:947-949
    ps -= (0.5 * t) * g;
    if (!project_cotangent_w<C>(A, w, ps, js)) { status = MM_ST_LINALG; break; }
    // ---- B(t): n_inner retractions + reversibility checks              integrators.py:951-979
    for (int inn = 0; inn < n_inner && status == MM_ST_OK; ++inn) {
      const double q_prev = qs;
      const Col<C> j_prev = js;
      qs += t_in * minv1(A, w, ps);
      Col<C> j_new;
      status = project_w<C>(A, w, qs, ps, j_prev, t_in, &j_new, &n_newton);
      if (status != MM_ST_OK) break;
      if (inn == n_inner - 1) {  // pre-evaluated dh1_dpos, integrators.py:956-969
        if (!dh1_dpos_w<C>(A, w, qs, &gs)) { status = MM_ST_LINALG; break; }
        ++n_grad;
      }
      if (!project_cotangent_w<C>(A, w, ps, j_new)) { status = MM_ST_LINALG; break; }
      // reversibility check on a copy                                    integrators.py:971-979
      double qb = qs, pb = ps;
      Col<C> j_tmp;
      qb += -t_in * minv1(A, w, pb);
      status = project_w<C>(A, w, qb, pb, j_new, -t_in, &j_tmp, &n_newton);
      if (status != MM_ST_OK) break;
      if (wnorm(qb - q_prev, A.opts.rev_norm) > A.opts.rev_tol) { status = MM_ST_NON_REVERSIBLE; break; }
      js = j_new;
    }
    if (status != MM_ST_OK) break;
    // ---- A(t/2)
    ps -= (0.5 * t) * gs;
    if (!project_cotangent_w<C>(A, w, ps, js)) { status = MM_ST_LINALG; break; }
    q = qs; p = ps; jac = js; g = gs;
    ++done;
  }
  if (in) {
    A.pos[chain * dim + lane] = q;
    A.mom[chain * dim + lane] = p;
  }
  if (lane == 0) {
    A.status[chain] = status;
    A.n_done[chain] = done;
    if (A.counters) {
      atomicAdd((unsigned long long*)&A.counters->n_newton_iters, (unsigned long long)n_newton);
      atomicAdd((unsigned long long*)&A.counters->n_constr, (unsigned long long)n_newton);
      atomicAdd((unsigned long long*)&A.counters->n_grad, (unsigned long long)n_grad);
    }
  }
}

// project_onto_cotangent_space of the momenta (systems.py:863-873) - what sample_momentum calls after the draw
template <int C>
__global__ __launch_bounds__(64 * kWavesPerBlock) void project_momentum_wave_kernel(ConArgs A) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t chain = (int64_t)blockIdx.x * kWavesPerBlock + wave;
  if (chain >= A.n_chains) return;
  double* wl = lds + wave * kWaveLds;
  const WaveCtx w{wl, wl + 64 * kRowStride, wl + 64 * kRowStride + 64, wl + 64 * kRowStride + 128, lane, A.dim};
  const int dim = A.dim;
  const bool in = lane < dim;
  const double q = in ? A.pos[chain * dim + lane] : 0.0;
  double p = in ? A.mom[chain * dim + lane] : 0.0;
  const Col<C> jac = jacob_w<C>(A, w, q);
  const bool ok = project_cotangent_w<C>(A, w, p, jac);
  if (in) A.mom[chain * dim + lane] = ok ? p : __longlong_as_double(0x7ff8000000000000LL);
}

template <int C>
int launch_wave(mm_ctx* ctx, const ConArgs& a, bool project_only) {
  const size_t lds = (size_t)kWavesPerBlock * kWaveLds * sizeof(double);
  const unsigned blocks = (unsigned)((a.n_chains + kWavesPerBlock - 1) / kWavesPerBlock);
  if (project_only) {
    MM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(project_momentum_wave_kernel<C>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((project_momentum_wave_kernel<C>), dim3(blocks), dim3(64 * kWavesPerBlock), lds, ctx->stream, a);
    MM_HIP_CHECK(ctx, hipGetLastError());
    return MM_OK;
  }
  MM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(constrained_wave_kernel<C>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((constrained_wave_kernel<C>), dim3(blocks), dim3(64 * kWavesPerBlock), lds, ctx->stream, a);
  MM_HIP_CHECK(ctx, hipGetLastError());
  return MM_OK;
}

}  // namespace

// true if the wave-per-chain kernel covers this model (the caller falls back to the lane-per-chain path otherwise)
bool mm_constrained_wave_supports(const mmcon::ConArgs& a, int n_constr) {
  if (a.dim <= 8 || a.dim > 64 || n_constr < 1 || n_constr > 8 || n_constr >= a.dim) return false;
  if (a.gaussian) return false;  // the Gaussian split's rotations stay on the lane-per-chain path
  switch (a.constr) {
    case MM_CONSTR_LINEAR: case MM_CONSTR_SPHERE: case MM_CONSTR_CIRCLE: case MM_CONSTR_FIRST: return true;
    case MM_CONSTR_SPHERE_PLANE: return n_constr == 2;
    default: return false;
  }
}

int mm_launch_constrained_wave(mm_ctx* ctx, int n_constr, const mmcon::ConArgs& a, bool project_only) {
  switch (n_constr) {
    case 1: return launch_wave<1>(ctx, a, project_only);
    case 2: return launch_wave<2>(ctx, a, project_only);
    case 3: return launch_wave<3>(ctx, a, project_only);
    case 4: return launch_wave<4>(ctx, a, project_only);
    case 5: return launch_wave<5>(ctx, a, project_only);
    case 6: return launch_wave<6>(ctx, a, project_only);
    case 7: return launch_wave<7>(ctx, a, project_only);
    default: return launch_wave<8>(ctx, a, project_only);
  }
}
