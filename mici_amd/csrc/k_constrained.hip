// Constrained leapfrog on DenseConstrainedEuclideanMetricSystem, one lane per chain, everything in
// registers (D <= 8, C = 1).  gfx950 / CDNA4.
//
// Replaces, per chain and per step (reference /root/reference/src/mici):
//   ConstrainedLeapfrogIntegrator._step / _step_a / _step_b / _h2_flow_retraction_onto_manifold /
//       _project_onto_cotangent_space                       integrators.py:929-984
//   solve_projection_onto_manifold_newton                   solvers.py:429-469
//   ConstrainedEuclideanMetricSystem.constr / jacob_constr / dh2_flow_dmom / gram / inv_gram /
//       project_onto_cotangent_space                        systems.py:786-873
//   DenseConstrainedEuclideanMetricSystem.jacob_constr_inner_product  systems.py:1010-1022
//   DensePositiveDefiniteMatrix (1x1 Gram), DenseSquareMatrix / InverseLUFactoredSquareMatrix (1x1
//       residual Jacobian)                                  matrices.py:1161-1188, 1270-1411
// The state of a chain is 2*D doubles (48 B for the torus): there is no HBM roofline to speak of, the
// kernel is FP64-VALU / transcendental bound; data-dependent Newton iteration counts are handled by
// SIMT masking (lanes of a wave wait for their slowest chain).
#include "mm_device.h"

namespace {

struct ConArgs {
  double* pos;
  double* mom;
  const int8_t* dir;
  const double* step_scale;
  int32_t* status;
  int32_t* n_done;
  int64_t n_chains;
  double step_size;
  int n_steps;
  int target;
  const double* tparams;
  int metric_kind;
  const double* minv;  // diag: 1/diag[D]; dense: explicit inverse [D*D]
  int constr;
  double cp0, cp1;  // constraint params (torus: R, r)
  mm_proj_opts opts;
  mm_counters* counters;
};

template <int D>
struct Vec {
  double v[D];
};

template <int D>
__device__ __forceinline__ Vec<D> target_grad(const ConArgs& A, const Vec<D>& q) {
  Vec<D> g;
  const mmdev::TargetAux aux;  // no wave-collective targets here (funnel is rejected on the host)
#pragma unroll
  for (int i = 0; i < D; ++i) g.v[i] = mmdev::target_grad_elem(A.target, aux, q.v, i, D, A.tparams);
  return g;
}

template <int D>
__device__ __forceinline__ Vec<D> minv_apply(const ConArgs& A, const Vec<D>& x) {
  Vec<D> y;
  if (A.metric_kind == MM_METRIC_IDENTITY) {
    y = x;
  } else if (A.metric_kind == MM_METRIC_DIAG) {
#pragma unroll
    for (int i = 0; i < D; ++i) y.v[i] = A.minv[i] * x.v[i];
  } else {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      double s = 0.0;
#pragma unroll
      for (int j = 0; j < D; ++j) s += A.minv[i * D + j] * x.v[j];
      y.v[i] = s;
    }
  }
  return y;
}

template <int D>
__device__ __forceinline__ double constr_value(const ConArgs& A, const Vec<D>& q) {
  if (A.constr == MM_CONSTR_TORUS) {
    constexpr int I1 = D > 1 ? 1 : 0;
    const double rho = sqrt(q.v[0] * q.v[0] + q.v[I1] * q.v[I1]);
    const double dr = rho - A.cp0;
    return dr * dr + q.v[D > 2 ? 2 : 0] * q.v[D > 2 ? 2 : 0] - A.cp1 * A.cp1;
  }
  if (A.constr == MM_CONSTR_FIRST) return q.v[0];
  return q.v[0] * q.v[0] + q.v[D > 1 ? 1 : 0] * q.v[D > 1 ? 1 : 0] - 1.0;  // circle
}

template <int D>
__device__ __forceinline__ Vec<D> constr_jacob(const ConArgs& A, const Vec<D>& q) {
  Vec<D> j;
#pragma unroll
  for (int i = 0; i < D; ++i) j.v[i] = 0.0;
  if (A.constr == MM_CONSTR_TORUS) {
    constexpr int I1 = D > 1 ? 1 : 0;
    const double rho = sqrt(q.v[0] * q.v[0] + q.v[I1] * q.v[I1]);
    const double f = 2.0 * (rho - A.cp0) / rho;
    j.v[0] = f * q.v[0];
    if constexpr (D > 1) j.v[1] = f * q.v[1];
    if constexpr (D > 2) j.v[2] = 2.0 * q.v[2];
  } else if (A.constr == MM_CONSTR_FIRST) {
    j.v[0] = 1.0;
  } else {
    j.v[0] = 2.0 * q.v[0];
    if constexpr (D > 1) j.v[1] = 2.0 * q.v[1];
  }
  return j;
}

template <int D>
__device__ __forceinline__ double dot(const Vec<D>& a, const Vec<D>& b) {
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < D; ++i) s += a.v[i] * b.v[i];
  return s;
}

template <int D>
__device__ __forceinline__ double vnorm(const Vec<D>& a, int kind) {
  double acc = 0.0;
#pragma unroll
  for (int i = 0; i < D; ++i) acc = mmdev::wave_norm_accum(acc, a.v[i], kind);
  return kind == MM_NORM_LINF ? acc : sqrt(acc);
}

__device__ __forceinline__ bool finite(double x) { return fabs(x) <= 1.79769313486231570815e308; }

// mom - J^T (J M^-1 J^T)^-1 J M^-1 mom     (systems.py:863-873; Gram matrix Cholesky-factored, 1x1)
template <int D>
__device__ __forceinline__ bool project_cotangent(const ConArgs& A, Vec<D>& p, const Vec<D>& jac) {
  const Vec<D> mj = minv_apply<D>(A, jac);
  const double gram = dot<D>(jac, mj);
  if (!(gram > 0.0) || !finite(gram)) return false;  // "Cholesky factorisation failed." / not finite
  const double l = sqrt(gram);
  const double inv = (1.0 / l) / l;  // explicit inverse L^-T L^-1 of the 1x1 factor
  const Vec<D> mp = minv_apply<D>(A, p);
  const double lam = inv * dot<D>(jac, mp);
#pragma unroll
  for (int i = 0; i < D; ++i) p.v[i] -= jac.v[i] * lam;
  return true;
}

// solve_projection_onto_manifold_newton (solvers.py:429-469) for C = 1 and a fixed metric:
// dh2_flow_dmom = (|t| M^-1, I) (systems.py:794-799).
template <int D>
__device__ __forceinline__ int newton_project(const ConArgs& A, Vec<D>& q, Vec<D>& p,
                                              const Vec<D>& jac_prev, double t, Vec<D>* jac_out,
                                              long long* n_iters) {
  const mm_proj_opts& o = A.opts;
  const double abs_t = fabs(t);
  const Vec<D> mjp = minv_apply<D>(A, jac_prev);  // M^-1 J_prev^T
  Vec<D> mu;
#pragma unroll
  for (int i = 0; i < D; ++i) mu.v[i] = 0.0;
  for (int it = 0; it < o.max_iters; ++it) {
    *n_iters += 1;
    const Vec<D> jac = constr_jacob<D>(A, q);
    const double c = constr_value<D>(A, q);
    const double err = fabs(c);  // both norms of a 1-vector
    const double a = dot<D>(jac, mjp) * abs_t;  // residual Jacobian J (|t| M^-1) J_prev^T
    if (!finite(a)) return MM_ST_SOLVER_LINALG;  // "Array is not finite." inside the solver
    const double dmu = c / a;                    // 1x1 LU solve
    Vec<D> dpos;
#pragma unroll
    for (int i = 0; i < D; ++i) dpos.v[i] = abs_t * (mjp.v[i] * dmu);
    if (err > o.div_tol || err != err) return MM_ST_DIVERGED;
    if (err < o.constr_tol && vnorm<D>(dpos, o.norm) < o.pos_tol) {
      const double sgn = (t > 0.0) ? 1.0 : ((t < 0.0) ? -1.0 : 0.0);
#pragma unroll
      for (int i = 0; i < D; ++i) p.v[i] -= sgn * mu.v[i];
      *jac_out = jac;
      return MM_ST_OK;
    }
#pragma unroll
    for (int i = 0; i < D; ++i) {
      mu.v[i] += jac_prev.v[i] * dmu;
      q.v[i] -= dpos.v[i];
    }
  }
  return MM_ST_MAX_ITERS;
}

// solve_projection_onto_manifold_quasi_newton (solvers.py:303-343): Gram matrix J_prev (|t| M^-1) J_prev^T
// Cholesky-factored once before the loop (failure = LinAlgError OUTSIDE the solver), only constr in it.
template <int D>
__device__ __forceinline__ int quasi_newton_project(const ConArgs& A, Vec<D>& q, Vec<D>& p,
                                                    const Vec<D>& jac_prev, double t, Vec<D>* jac_out,
                                                    long long* n_iters) {
  const mm_proj_opts& o = A.opts;
  const double abs_t = fabs(t);
  const Vec<D> mjp = minv_apply<D>(A, jac_prev);
  const double gram = abs_t * dot<D>(jac_prev, mjp);
  if (!(gram > 0.0) || !finite(gram)) return MM_ST_LINALG;
  const double l = sqrt(gram);
  const double inv = (1.0 / l) / l;
  Vec<D> mu;
#pragma unroll
  for (int i = 0; i < D; ++i) mu.v[i] = 0.0;
  for (int it = 0; it < o.max_iters; ++it) {
    *n_iters += 1;
    const double c = constr_value<D>(A, q);
    const double err = fabs(c);
    const double dmu = inv * c;
    Vec<D> dpos;
#pragma unroll
    for (int i = 0; i < D; ++i) dpos.v[i] = abs_t * (mjp.v[i] * dmu);
    if (err > o.div_tol || err != err) return MM_ST_DIVERGED;
    if (err < o.constr_tol && vnorm<D>(dpos, o.norm) < o.pos_tol) {
      const double sgn = (t > 0.0) ? 1.0 : ((t < 0.0) ? -1.0 : 0.0);
#pragma unroll
      for (int i = 0; i < D; ++i) p.v[i] -= sgn * mu.v[i];
      *jac_out = constr_jacob<D>(A, q);
      return MM_ST_OK;
    }
#pragma unroll
    for (int i = 0; i < D; ++i) {
      mu.v[i] += jac_prev.v[i] * dmu;
      q.v[i] -= dpos.v[i];
    }
  }
  return MM_ST_MAX_ITERS;
}

// solve_projection_onto_manifold_newton_with_line_search (solvers.py:561-614)
template <int D>
__device__ __forceinline__ int line_search_project(const ConArgs& A, Vec<D>& q, Vec<D>& p,
                                                   const Vec<D>& jac_prev, double t, Vec<D>* jac_out,
                                                   long long* n_iters) {
  const mm_proj_opts& o = A.opts;
  const double abs_t = fabs(t);
  const Vec<D> mjp = minv_apply<D>(A, jac_prev);
  Vec<D> mu, dpos;
#pragma unroll
  for (int i = 0; i < D; ++i) { mu.v[i] = 0.0; dpos.v[i] = 0.0; }
  double step = 0.0;
  for (int it = 0; it < o.max_iters; ++it) {
    *n_iters += 1;
    const Vec<D> jac = constr_jacob<D>(A, q);
    const double c = constr_value<D>(A, q);
    const double err = fabs(c);
    if (it > 0 && (err > o.div_tol || err != err)) return MM_ST_DIVERGED;
    bool small_step = (it == 0);
    if (!small_step) {
      Vec<D> sd;
#pragma unroll
      for (int i = 0; i < D; ++i) sd.v[i] = step * dpos.v[i];
      small_step = vnorm<D>(sd, o.norm) < o.pos_tol;
    }
    if (err < o.constr_tol && small_step) {
      const double sgn = (t > 0.0) ? 1.0 : ((t < 0.0) ? -1.0 : 0.0);
#pragma unroll
      for (int i = 0; i < D; ++i) p.v[i] -= sgn * mu.v[i];
      *jac_out = jac;
      return MM_ST_OK;
    }
    const double a = dot<D>(jac, mjp) * abs_t;
    if (!finite(a)) return MM_ST_SOLVER_LINALG;
    const double dmu = c / a;
#pragma unroll
    for (int i = 0; i < D; ++i) dpos.v[i] = -(abs_t * (mjp.v[i] * dmu));
    const Vec<D> q_curr = q;
    step = 1.0;
    for (int ls = 0; ls < o.max_line_search_iters; ++ls) {
#pragma unroll
      for (int i = 0; i < D; ++i) q.v[i] = q_curr.v[i] + step * dpos.v[i];
      const double new_err = fabs(constr_value<D>(A, q));
      if (new_err < err) break;
      step *= 0.5;
    }
#pragma unroll
    for (int i = 0; i < D; ++i) mu.v[i] += step * (jac_prev.v[i] * dmu);
  }
  return MM_ST_MAX_ITERS;
}

template <int D>
__device__ __forceinline__ int project(const ConArgs& A, Vec<D>& q, Vec<D>& p, const Vec<D>& jac_prev,
                                       double t, Vec<D>* jac_out, long long* n_iters) {
  if (A.opts.solver == MM_PROJ_QUASI_NEWTON)
    return quasi_newton_project<D>(A, q, p, jac_prev, t, jac_out, n_iters);
  if (A.opts.solver == MM_PROJ_NEWTON_LINE_SEARCH)
    return line_search_project<D>(A, q, p, jac_prev, t, jac_out, n_iters);
  return newton_project<D>(A, q, p, jac_prev, t, jac_out, n_iters);
}

template <int D>
__global__ __launch_bounds__(256) void constrained_leapfrog_kernel(ConArgs A) {
  const int64_t chain = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (chain >= A.n_chains) return;
  Vec<D> q, p;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    q.v[i] = A.pos[chain * D + i];
    p.v[i] = A.mom[chain * D + i];
  }
  const double t = mmdev::signed_step(A.dir, A.step_scale, chain, A.step_size);
  const int n_inner = A.opts.n_inner;
  const double t_in = t / n_inner;
  long long n_newton = 0, n_grad = 0;
  int status = MM_ST_OK, done = 0;

  Vec<D> g = target_grad<D>(A, q);  // cached dh1_dpos at the current position
  Vec<D> jac = constr_jacob<D>(A, q);
  ++n_grad;
  for (int s = 0; s < A.n_steps && status == MM_ST_OK; ++s) {
    Vec<D> qs = q, ps = p, js = jac;
    // ---- A(t/2): h1_flow then cotangent projection                    integrators.py:947-949
#pragma unroll
    for (int i = 0; i < D; ++i) ps.v[i] -= (0.5 * t) * g.v[i];
    if (!project_cotangent<D>(A, ps, js)) { status = MM_ST_LINALG; break; }
    // ---- B(t): n_inner retractions + reversibility checks              integrators.py:951-979
    Vec<D> gs = g;
    for (int in = 0; in < n_inner && status == MM_ST_OK; ++in) {
      const Vec<D> q_prev = qs, j_prev = js;
      const Vec<D> v = minv_apply<D>(A, ps);
#pragma unroll
      for (int i = 0; i < D; ++i) qs.v[i] += t_in * v.v[i];  // h2_flow, systems.py:362-363
      Vec<D> j_new;
      status = project<D>(A, qs, ps, j_prev, t_in, &j_new, &n_newton);
      if (status != MM_ST_OK) break;
      if (in == n_inner - 1) {  // pre-evaluated dh1_dpos, integrators.py:956-969
        gs = target_grad<D>(A, qs);
        ++n_grad;
      }
      if (!project_cotangent<D>(A, ps, j_new)) { status = MM_ST_LINALG; break; }
      // reversibility check on a copy                                    integrators.py:971-979
      Vec<D> qb = qs, pb = ps, j_tmp;
      const Vec<D> vb = minv_apply<D>(A, pb);
#pragma unroll
      for (int i = 0; i < D; ++i) qb.v[i] -= t_in * vb.v[i];
      status = project<D>(A, qb, pb, j_new, -t_in, &j_tmp, &n_newton);
      if (status != MM_ST_OK) break;
      Vec<D> diff;
#pragma unroll
      for (int i = 0; i < D; ++i) diff.v[i] = qb.v[i] - q_prev.v[i];
      if (vnorm<D>(diff, A.opts.rev_norm) > A.opts.rev_tol) { status = MM_ST_NON_REVERSIBLE; break; }
      js = j_new;
    }
    if (status != MM_ST_OK) break;
    // ---- A(t/2)
#pragma unroll
    for (int i = 0; i < D; ++i) ps.v[i] -= (0.5 * t) * gs.v[i];
    if (!project_cotangent<D>(A, ps, js)) { status = MM_ST_LINALG; break; }
    q = qs; p = ps; jac = js; g = gs;
    ++done;
  }
#pragma unroll
  for (int i = 0; i < D; ++i) {
    A.pos[chain * D + i] = q.v[i];
    A.mom[chain * D + i] = p.v[i];
  }
  A.status[chain] = status;
  A.n_done[chain] = done;
  if (A.counters) {
    // one atomic per wave (the compiler coalesces uniform-address atomics of active lanes)
    atomicAdd((unsigned long long*)&A.counters->n_newton_iters, (unsigned long long)n_newton);
    atomicAdd((unsigned long long*)&A.counters->n_constr, (unsigned long long)n_newton);
    atomicAdd((unsigned long long*)&A.counters->n_grad, (unsigned long long)n_grad);
  }
}

template <int D>
__global__ __launch_bounds__(256) void project_momentum_kernel(ConArgs A) {
  const int64_t chain = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (chain >= A.n_chains) return;
  Vec<D> q, p;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    q.v[i] = A.pos[chain * D + i];
    p.v[i] = A.mom[chain * D + i];
  }
  const Vec<D> jac = constr_jacob<D>(A, q);
  const bool ok = project_cotangent<D>(A, p, jac);
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
#pragma unroll
  for (int i = 0; i < D; ++i) A.mom[chain * D + i] = ok ? p.v[i] : nan;
}

ConArgs make_args(const mm_model* m, mm_state* s) {
  ConArgs a{};
  a.pos = s->d_pos;
  a.mom = s->d_mom;
  a.dir = s->d_dir;
  a.step_scale = s->d_step_scale;
  a.status = s->d_status;
  a.n_done = s->d_n_done;
  a.n_chains = s->n;
  a.target = m->target;
  a.tparams = m->d_target_params;
  a.metric_kind = m->metric_kind;
  a.minv = m->d_metric_inv;
  a.constr = m->constr;
  a.cp0 = m->h_constr_params[0];
  a.cp1 = m->h_constr_params[1];
  return a;
}

template <int D>
int launch_d(mm_ctx* ctx, const ConArgs& a, bool project_only) {
  const unsigned blocks = (unsigned)((a.n_chains + 255) / 256);
  if (project_only)
    hipLaunchKernelGGL((project_momentum_kernel<D>), dim3(blocks), dim3(256), 0, ctx->stream, a);
  else
    hipLaunchKernelGGL((constrained_leapfrog_kernel<D>), dim3(blocks), dim3(256), 0, ctx->stream, a);
  MM_HIP_CHECK(ctx, hipGetLastError());
  return MM_OK;
}

int launch(mm_ctx* ctx, const mm_model* m, const ConArgs& a, bool project_only) {
  if (m->target == MM_TARGET_FUNNEL) {
    mm_set_error(ctx, "constrained kernels: the funnel target needs a wave-collective gradient");
    return MM_ERR_UNSUPPORTED;
  }
  switch (m->dim) {
    case 1: return launch_d<1>(ctx, a, project_only);
    case 2: return launch_d<2>(ctx, a, project_only);
    case 3: return launch_d<3>(ctx, a, project_only);
    case 4: return launch_d<4>(ctx, a, project_only);
    case 5: return launch_d<5>(ctx, a, project_only);
    case 6: return launch_d<6>(ctx, a, project_only);
    case 7: return launch_d<7>(ctx, a, project_only);
    case 8: return launch_d<8>(ctx, a, project_only);
    default:
      mm_set_error(ctx, "constrained leapfrog kernels support dim <= 8 (register-resident chains)");
      return MM_ERR_UNSUPPORTED;
  }
}

}  // namespace

int mm_launch_constrained_leapfrog(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int n_steps,
                                   const mm_proj_opts& opts, mm_counters* d_counters) {
  ConArgs a = make_args(m, s);
  a.step_size = h;
  a.n_steps = n_steps;
  a.opts = opts;
  a.counters = d_counters;
  return launch(ctx, m, a, false);
}

int mm_launch_constrained_project_momentum(mm_ctx* ctx, const mm_model* m, mm_state* s) {
  ConArgs a = make_args(m, s);
  return launch(ctx, m, a, true);
}
