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
//   dens_wrt_hausdorff=False: h1 += log det sqrt Gram, dh1_dpos += mhp_constr(inv_gram J M^-1)
//                                                           systems.py:829-831, 846-862, 1024-1031
//   GaussianDenseConstrainedEuclideanMetricSystem (exact h2 rotation, DenseSymmetricMatrix Gram-type
//       matrices, dh2_flow_dmom = (V diag(sin(w|t|) w) V^T, V diag(cos(w|t|)) V^T))   systems.py:1034-1184
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
  int ambient;          // dens_wrt_hausdorff=False: the Gram log-determinant term is part of h1
  int gaussian;         // Gaussian split: exact h2 rotation, symmetric (not Cholesky-factored) Gram matrices
  const double* omega;  // gaussian: 1/sqrt(eigval)[D] (nullptr for the identity metric)
  const double* eigvec; // gaussian + dense metric: V [D*D] then V^T [D*D]
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

// Per-chain constants of the Gaussian split for the chain's |inner time step|: sin(w|t|) w, sin(w|t|)/w, cos(w|t|).
template <int D>
struct Rot {
  double sw[D], sow[D], cw[D];
};

template <int D>
__device__ __forceinline__ Rot<D> make_rot(const ConArgs& A, double abs_t) {
  Rot<D> r;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    const double om = A.omega ? A.omega[i] : 1.0;
    double sn, cs;
    sincos(om * abs_t, &sn, &cs);
    r.sw[i] = sn * om;
    r.sow[i] = sn / om;
    r.cw[i] = cs;
  }
  return r;
}

// V^T x (dense metric) or x itself (identity / diagonal metric: V = I)
template <int D>
__device__ __forceinline__ Vec<D> to_eigenbasis(const ConArgs& A, const Vec<D>& x) {
  if (A.metric_kind != MM_METRIC_DENSE) return x;
  Vec<D> y;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < D; ++j) s += A.eigvec[j * D + i] * x.v[j];
    y.v[i] = s;
  }
  return y;
}

template <int D>
__device__ __forceinline__ Vec<D> from_eigenbasis(const ConArgs& A, const Vec<D>& x) {
  if (A.metric_kind != MM_METRIC_DENSE) return x;
  Vec<D> y;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    double s = 0.0;
#pragma unroll
    for (int j = 0; j < D; ++j) s += A.eigvec[i * D + j] * x.v[j];
    y.v[i] = s;
  }
  return y;
}

// V diag(coef) V^T x: EigendecomposedSymmetricMatrix @ x (matrices.py:1572-1573)
template <int D>
__device__ __forceinline__ Vec<D> eig_apply(const ConArgs& A, const double (&coef)[D], const Vec<D>& x) {
  Vec<D> y = to_eigenbasis<D>(A, x);
#pragma unroll
  for (int i = 0; i < D; ++i) y.v[i] *= coef[i];
  return from_eigenbasis<D>(A, y);
}

// h2_flow over sgn * |t|: pos += t M^-1 mom (systems.py:362-363) or the exact rotation of the Gaussian
// split (systems.py:464-474)
template <int D>
__device__ __forceinline__ void h2_flow(const ConArgs& A, const Rot<D>& rot, Vec<D>& q, Vec<D>& p, double t,
                                        double sgn) {
  if (!A.gaussian) {
    const Vec<D> v = minv_apply<D>(A, p);
#pragma unroll
    for (int i = 0; i < D; ++i) q.v[i] += t * v.v[i];
    return;
  }
  const Vec<D> a = to_eigenbasis<D>(A, q), b = to_eigenbasis<D>(A, p);
  Vec<D> na, nb;
#pragma unroll
  for (int i = 0; i < D; ++i) {
    na.v[i] = rot.cw[i] * a.v[i] + (sgn * rot.sw[i]) * b.v[i];
    nb.v[i] = rot.cw[i] * b.v[i] - (sgn * rot.sow[i]) * a.v[i];
  }
  q = from_eigenbasis<D>(A, na);
  p = from_eigenbasis<D>(A, nb);
}

// dh2_flow_dmom(|t|)[0] @ x without its scalar factor: returns (y, scale) with the product = scale * y.
// Euclidean: |t| M^-1 (systems.py:794-799); Gaussian: V diag(sin(w|t|) w) V^T (systems.py:1163-1176).
template <int D>
__device__ __forceinline__ Vec<D> flow_pos_dmom(const ConArgs& A, const Rot<D>& rot, const Vec<D>& x) {
  return A.gaussian ? eig_apply<D>(A, rot.sw, x) : minv_apply<D>(A, x);
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

// mhp_constr(state)(m) for C = 1: the constraint Hessian applied to the row m (systems.py:1006-1008)
template <int D>
__device__ __forceinline__ Vec<D> constr_hess_apply(const ConArgs& A, const Vec<D>& q, const Vec<D>& m) {
  Vec<D> out;
#pragma unroll
  for (int i = 0; i < D; ++i) out.v[i] = 0.0;
  if (A.constr == MM_CONSTR_TORUS) {
    if constexpr (D > 2) {
      const double x = q.v[0], y = q.v[1];
      const double rho = sqrt(x * x + y * y), rho2 = rho * rho, rho3 = rho2 * rho;
      const double dr = rho - A.cp0;
      const double hxx = 2.0 * x * x / rho2 + 2.0 * dr * y * y / rho3;
      const double hyy = 2.0 * y * y / rho2 + 2.0 * dr * x * x / rho3;
      const double hxy = 2.0 * x * y * A.cp0 / rho3;
      out.v[0] = hxx * m.v[0] + hxy * m.v[1];
      out.v[1] = hxy * m.v[0] + hyy * m.v[1];
      out.v[2] = 2.0 * m.v[2];
    }
  } else if (A.constr == MM_CONSTR_CIRCLE) {
    out.v[0] = 2.0 * m.v[0];
    if constexpr (D > 1) out.v[1] = 2.0 * m.v[1];
  }
  return out;
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

// Inverse of a 1x1 Gram-type matrix.  DensePositiveDefiniteMatrix: Cholesky factor then the explicit
// inverse L^-T L^-1 (fails unless positive, matrices.py:1161-1188); DenseSymmetricMatrix of the Gaussian
// split: eigendecomposition, 1 / eigval (matrices.py:1446-1447).  false = LinAlgError.
__device__ __forceinline__ bool gram_inverse(const ConArgs& A, double gram, double* inv) {
  if (!finite(gram)) return false;
  if (A.gaussian) {
    *inv = 1.0 / gram;
    return true;
  }
  if (!(gram > 0.0)) return false;
  const double l = sqrt(gram);
  *inv = (1.0 / l) / l;
  return true;
}

// mom - J^T (J M^-1 J^T)^-1 J M^-1 mom     (systems.py:863-873; Gram matrix Cholesky-factored, 1x1)
template <int D>
__device__ __forceinline__ bool project_cotangent(const ConArgs& A, Vec<D>& p, const Vec<D>& jac) {
  const Vec<D> mj = minv_apply<D>(A, jac);
  const double gram = dot<D>(jac, mj);
  double inv;
  if (!gram_inverse(A, gram, &inv)) return false;  // "Cholesky factorisation failed." / not finite
  const Vec<D> mp = minv_apply<D>(A, p);
  const double lam = inv * dot<D>(jac, mp);
#pragma unroll
  for (int i = 0; i < D; ++i) p.v[i] -= jac.v[i] * lam;
  return true;
}

// dh1_dpos (systems.py:858-862): grad_neg_log_dens, plus for dens_wrt_hausdorff=False
// grad_log_det_sqrt_gram = mhp_constr(inv_gram J M^-1) (systems.py:1024-1031).  false = LinAlgError.
template <int D>
__device__ __forceinline__ bool dh1_dpos(const ConArgs& A, const Vec<D>& q, Vec<D>* out) {
  Vec<D> g = target_grad<D>(A, q);
  if (A.ambient) {
    const Vec<D> jac = constr_jacob<D>(A, q);
    const Vec<D> mj = minv_apply<D>(A, jac);
    double inv;
    if (!gram_inverse(A, dot<D>(jac, mj), &inv)) return false;
    Vec<D> m;
#pragma unroll
    for (int i = 0; i < D; ++i) m.v[i] = inv * jac.v[i];
    const Vec<D> hm = constr_hess_apply<D>(A, q, minv_apply<D>(A, m));
#pragma unroll
    for (int i = 0; i < D; ++i) g.v[i] += hm.v[i];
  }
  *out = g;
  return true;
}

// solve_projection_onto_manifold_newton (solvers.py:429-469) for C = 1 and a fixed metric:
// dh2_flow_dmom = (|t| M^-1, I) (systems.py:794-799).
template <int D>
__device__ __forceinline__ int newton_project(const ConArgs& A, const Rot<D>& rot, Vec<D>& q, Vec<D>& p,
                                              const Vec<D>& jac_prev, double t, Vec<D>* jac_out,
                                              long long* n_iters) {
  const mm_proj_opts& o = A.opts;
  const double abs_t = A.gaussian ? 1.0 : fabs(t);  // the Gaussian flow matrices carry |t| themselves
  const Vec<D> mjp = flow_pos_dmom<D>(A, rot, jac_prev);  // M^-1 J_prev^T, or V diag(sin(w|t|) w) V^T J_prev^T
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
      const Vec<D> cmu = A.gaussian ? eig_apply<D>(A, rot.cw, mu) : mu;  // dh2_flow_mom_dmom @ mu
#pragma unroll
      for (int i = 0; i < D; ++i) p.v[i] -= sgn * cmu.v[i];
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
__device__ __forceinline__ int quasi_newton_project(const ConArgs& A, const Rot<D>& rot, Vec<D>& q, Vec<D>& p,
                                                    const Vec<D>& jac_prev, double t, Vec<D>* jac_out,
                                                    long long* n_iters) {
  const mm_proj_opts& o = A.opts;
  const double abs_t = A.gaussian ? 1.0 : fabs(t);
  const Vec<D> mjp = flow_pos_dmom<D>(A, rot, jac_prev);
  const double gram = abs_t * dot<D>(jac_prev, mjp);
  double inv;
  if (!gram_inverse(A, gram, &inv)) return MM_ST_LINALG;
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
      const Vec<D> cmu = A.gaussian ? eig_apply<D>(A, rot.cw, mu) : mu;  // dh2_flow_mom_dmom @ mu
#pragma unroll
      for (int i = 0; i < D; ++i) p.v[i] -= sgn * cmu.v[i];
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
__device__ __forceinline__ int line_search_project(const ConArgs& A, const Rot<D>& rot, Vec<D>& q, Vec<D>& p,
                                                   const Vec<D>& jac_prev, double t, Vec<D>* jac_out,
                                                   long long* n_iters) {
  const mm_proj_opts& o = A.opts;
  const double abs_t = A.gaussian ? 1.0 : fabs(t);
  const Vec<D> mjp = flow_pos_dmom<D>(A, rot, jac_prev);
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
      const Vec<D> cmu = A.gaussian ? eig_apply<D>(A, rot.cw, mu) : mu;  // dh2_flow_mom_dmom @ mu
#pragma unroll
      for (int i = 0; i < D; ++i) p.v[i] -= sgn * cmu.v[i];
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
__device__ __forceinline__ int project(const ConArgs& A, const Rot<D>& rot, Vec<D>& q, Vec<D>& p,
                                       const Vec<D>& jac_prev, double t, Vec<D>* jac_out, long long* n_iters) {
  if (A.opts.solver == MM_PROJ_QUASI_NEWTON)
    return quasi_newton_project<D>(A, rot, q, p, jac_prev, t, jac_out, n_iters);
  if (A.opts.solver == MM_PROJ_NEWTON_LINE_SEARCH)
    return line_search_project<D>(A, rot, q, p, jac_prev, t, jac_out, n_iters);
  return newton_project<D>(A, rot, q, p, jac_prev, t, jac_out, n_iters);
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

  Rot<D> rot{};
  if (A.gaussian) rot = make_rot<D>(A, fabs(t_in));
  Vec<D> g;  // cached dh1_dpos at the current position
  if (!dh1_dpos<D>(A, q, &g)) status = MM_ST_LINALG;
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
      h2_flow<D>(A, rot, qs, ps, t_in, t_in < 0.0 ? -1.0 : 1.0);
      Vec<D> j_new;
      status = project<D>(A, rot, qs, ps, j_prev, t_in, &j_new, &n_newton);
      if (status != MM_ST_OK) break;
      if (in == n_inner - 1) {  // pre-evaluated dh1_dpos, integrators.py:956-969
        if (!dh1_dpos<D>(A, qs, &gs)) { status = MM_ST_LINALG; break; }
        ++n_grad;
      }
      if (!project_cotangent<D>(A, ps, j_new)) { status = MM_ST_LINALG; break; }
      // reversibility check on a copy                                    integrators.py:971-979
      Vec<D> qb = qs, pb = ps, j_tmp;
      h2_flow<D>(A, rot, qb, pb, -t_in, t_in < 0.0 ? 1.0 : -1.0);
      status = project<D>(A, rot, qb, pb, j_new, -t_in, &j_tmp, &n_newton);
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

// h1's Gram term for dens_wrt_hausdorff=False: out[chain] += log_det_sqrt_gram = log|det gram| / 2
// (systems.py:829-831, 853-856); NaN where the reference raises LinAlgError.
template <int D>
__global__ __launch_bounds__(256) void add_log_det_sqrt_gram_kernel(ConArgs A, double* __restrict__ out) {
  const int64_t chain = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (chain >= A.n_chains) return;
  Vec<D> q;
#pragma unroll
  for (int i = 0; i < D; ++i) q.v[i] = A.pos[chain * D + i];
  const Vec<D> jac = constr_jacob<D>(A, q);
  const double gram = dot<D>(jac, minv_apply<D>(A, jac));
  double inv;
  double v = __longlong_as_double(0x7ff8000000000000LL);
  if (gram_inverse(A, gram, &inv))
    v = A.gaussian ? 0.5 * log(fabs(gram)) : 0.5 * (2.0 * log(fabs(sqrt(gram))));
  out[chain] += v;
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
  a.ambient = m->dens_wrt_ambient;
  a.gaussian = m->gaussian_split;
  a.omega = m->d_metric_omega;
  a.eigvec = m->d_metric_eigvec;
  return a;
}

template <int D>
int launch_d(mm_ctx* ctx, const ConArgs& a, bool project_only, double* h_out) {
  const unsigned blocks = (unsigned)((a.n_chains + 255) / 256);
  if (h_out)
    hipLaunchKernelGGL((add_log_det_sqrt_gram_kernel<D>), dim3(blocks), dim3(256), 0, ctx->stream, a, h_out);
  else if (project_only)
    hipLaunchKernelGGL((project_momentum_kernel<D>), dim3(blocks), dim3(256), 0, ctx->stream, a);
  else
    hipLaunchKernelGGL((constrained_leapfrog_kernel<D>), dim3(blocks), dim3(256), 0, ctx->stream, a);
  MM_HIP_CHECK(ctx, hipGetLastError());
  return MM_OK;
}

int launch(mm_ctx* ctx, const mm_model* m, const ConArgs& a, bool project_only, double* h_out = nullptr) {
  if (m->target == MM_TARGET_FUNNEL) {
    mm_set_error(ctx, "constrained kernels: the funnel target needs a wave-collective gradient");
    return MM_ERR_UNSUPPORTED;
  }
  switch (m->dim) {
    case 1: return launch_d<1>(ctx, a, project_only, h_out);
    case 2: return launch_d<2>(ctx, a, project_only, h_out);
    case 3: return launch_d<3>(ctx, a, project_only, h_out);
    case 4: return launch_d<4>(ctx, a, project_only, h_out);
    case 5: return launch_d<5>(ctx, a, project_only, h_out);
    case 6: return launch_d<6>(ctx, a, project_only, h_out);
    case 7: return launch_d<7>(ctx, a, project_only, h_out);
    case 8: return launch_d<8>(ctx, a, project_only, h_out);
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

int mm_launch_constrained_add_log_det_sqrt_gram(mm_ctx* ctx, const mm_model* m, mm_state* s, double* d_out) {
  ConArgs a = make_args(m, s);
  return launch(ctx, m, a, false, d_out);
}
