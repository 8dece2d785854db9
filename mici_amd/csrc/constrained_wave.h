// Device code of the wave-per-chain constrained kernels (k_constrained_wave.hip instantiates them for the built-in
// constraints; mm_rtc.hip compiles them at run time around a USER constraint, 64 < D <= 256 - round 5).
//
// Constrained leapfrog for D > 8 with 1 <= C <= 8 constraints: ONE WAVE PER CHAIN (round 3; VERDICT r02 "missing" #3, #4).
//
// The lane-per-chain core of constrained_core.h keeps every per-chain array in registers up to D = 8; beyond, its padded
// instantiations (k_constrained_wide*.hip) run the same code with the arrays - a C x D Jacobian is up to 4 KB, several
// copies live at once - in 5-26 KB of scratch per lane, and stop at D = 64.  Here a chain belongs to a wave instead: lane
// i holds coordinates i, i + 64, ... (NE of them: D <= 64 NE) of every D-vector and those columns of every C x D
// Jacobian in registers, D-long sums are wave reductions,
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
// (systems.py:829-862, 1024-1031) and - round 5 - GaussianDenseConstrainedEuclideanMetricSystem (systems.py:1034-1184): the
// exact rotation as h2_flow, dh2_flow_dmom = (V diag(sin(w|t|) w) V^T, V diag(cos(w|t|)) V^T) in the Newton matrices,
// eigendecomposed Gram inverses (the GAUSS instantiations, D <= 256), and USER constraints (the run-time compiled translation
// unit of mm_rtc.hip, 64 < D <= 256): their hooks work on whole arrays - constr on the position in LDS, jacob_constr /
// mhp_constr run by one lane into / from a C x D array in LDS of which every lane takes its columns.
#pragma once
#include "constrained_core.h"

namespace mmconw {

using namespace mmcon;
using namespace mmdev;


constexpr int kRowStride = 65;  // doubles per row of the transposition buffer
// LDS of one wave: prod[64][65], sums[64], nat[64 NE], vec[64 NE]
template <int NE>
constexpr int wave_lds() { return 64 * kRowStride + 64 + 2 * 64 * NE; }
// ... + for a USER constraint (run-time compiled: mm_rtc.hip) the scratch its hooks work on: a whole C x D array
// (jacob_constr's output; mhp_constr's operand) and one D-vector (mhp_constr's output)
template <int C, int NE>
constexpr int wave_lds_user() { return wave_lds<NE>() + (C + 1) * 64 * NE; }
// waves (chains) per workgroup: four while their LDS fits, two for the 16-coordinates-per-lane instantiations
template <int NE>
constexpr int waves_per_block() { return NE <= 4 ? 4 : 2; }

struct WaveCtx {
  double* prod;  // [64][65]
  double* sums;  // [64]
  double* nat;   // [64 NE] a D-vector in natural order (target gradient, dense-metric products)
  double* vec;   // [64 NE] second natural-order vector
  int lane, dim;
  double* ujac;  // user constraints: [C][dim] (nullptr otherwise)
  double* uout;  // user constraints: [dim]
};

// a D-vector spread over the wave: element e of lane i is coordinate i + 64 e (zero beyond dim)
template <int NE>
struct Vec {
  double v[NE];
};
template <int NE>
__device__ __forceinline__ Vec<NE> vzero() {
  Vec<NE> o;
#pragma unroll
  for (int e = 0; e < NE; ++e) o.v[e] = 0.0;
  return o;
}
// x + a y
template <int NE>
__device__ __forceinline__ Vec<NE> axpy(double a, const Vec<NE>& y, const Vec<NE>& x) {
  Vec<NE> o;
#pragma unroll
  for (int e = 0; e < NE; ++e) o.v[e] = __builtin_fma(a, y.v[e], x.v[e]);
  return o;
}
template <int NE>
__device__ __forceinline__ Vec<NE> scaled(double a, const Vec<NE>& x) {
  Vec<NE> o;
#pragma unroll
  for (int e = 0; e < NE; ++e) o.v[e] = a * x.v[e];
  return o;
}
template <int NE>
__device__ __forceinline__ Vec<NE> vsub(const Vec<NE>& x, const Vec<NE>& y) {
  Vec<NE> o;
#pragma unroll
  for (int e = 0; e < NE; ++e) o.v[e] = x.v[e] - y.v[e];
  return o;
}
template <int NE>
__device__ __forceinline__ Vec<NE> load_vec(const double* src, int lane, int dim) {
  Vec<NE> o;
#pragma unroll
  for (int e = 0; e < NE; ++e) o.v[e] = lane + 64 * e < dim ? src[lane + 64 * e] : 0.0;
  return o;
}
template <int NE>
__device__ __forceinline__ void store_vec(double* dst, int lane, int dim, const Vec<NE>& x) {
#pragma unroll
  for (int e = 0; e < NE; ++e)
    if (lane + 64 * e < dim) dst[lane + 64 * e] = x.v[e];
}
// the whole vector into a natural-order LDS array (zero beyond dim), visible to the wave
template <int NE>
__device__ __forceinline__ void publish(double* arr, int lane, const Vec<NE>& x) {
#pragma unroll
  for (int e = 0; e < NE; ++e) arr[lane + 64 * e] = x.v[e];
  wave_sync();
}

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

template <int NE>
__device__ __forceinline__ double wnorm(const Vec<NE>& x, int kind) {
  if (kind == MM_NORM_LINF) {
    double m = fabs(x.v[0]);
#pragma unroll
    for (int e = 1; e < NE; ++e) m = nanmax(m, fabs(x.v[e]));
    return wave_max(m);
  }
  double s = x.v[0] * x.v[0];
#pragma unroll
  for (int e = 1; e < NE; ++e) s = __builtin_fma(x.v[e], x.v[e], s);
  return sqrt(wave_sum(s));
}

// y_i = sum_j M[j][i] x_j for a dense row-major D x D matrix: a lane walks the COLUMNS of its coordinates (coalesced over the
// lanes for every j) with x_j broadcast from LDS.  (M symmetric: M x.  The Gaussian split hands it V for V^T x and the stored
// V^T for V x.)
template <int NE>
__device__ __forceinline__ Vec<NE> dense_walk(const double* __restrict__ M, const WaveCtx& w, const Vec<NE>& x) {
  Vec<NE> y = vzero<NE>();
  publish<NE>(w.vec, w.lane, x);
  // lanes beyond dim walk the last column (no predicate inside the loop) and drop their sums at the end
  const double* col[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) col[e] = M + (w.lane + 64 * e < w.dim ? w.lane + 64 * e : w.dim - 1);
  for (int j = 0; j < w.dim; ++j) {
    const double xj = w.vec[j];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      y.v[e] = __builtin_fma(*col[e], xj, y.v[e]);
      col[e] += w.dim;  // row j + 1 = column j + 1
    }
  }
#pragma unroll
  for (int e = 0; e < NE; ++e)
    if (w.lane + 64 * e >= w.dim) y.v[e] = 0.0;
  wave_sync();
  return y;
}

// y = M^-1 x.  Dense: M^-1 is symmetric.
template <int NE>
__device__ __forceinline__ Vec<NE> minv1(const ConArgs& A, const WaveCtx& w, const Vec<NE>& x) {
  if (A.metric_kind == MM_METRIC_IDENTITY) return x;
  Vec<NE> y = vzero<NE>();
  if (A.metric_kind == MM_METRIC_DIAG) {
#pragma unroll
    for (int e = 0; e < NE; ++e) y.v[e] = w.lane + 64 * e < w.dim ? A.minv[w.lane + 64 * e] * x.v[e] : 0.0;
    return y;
  }
  return dense_walk<NE>(A.minv, w, x);
}

template <int C, int NE>
struct Col {  // this lane's columns of a C x D matrix: row k, element e
  Vec<NE> r[C];
};

template <int C, int NE>
__device__ __forceinline__ Col<C, NE> minv_rows_w(const ConArgs& A, const WaveCtx& w, const Col<C, NE>& j) {
  Col<C, NE> o;
#pragma unroll
  for (int k = 0; k < C; ++k) o.r[k] = minv1<NE>(A, w, j.r[k]);
  return o;
}

// ---- Gaussian split (systems.py:1034-1184; constrained_core.h make_rot / h2_flow / flow_pos_dmom_rows / apply_mu) ----------
// this lane's coordinates of sin(w|t|) w, sin(w|t|) / w, cos(w|t|) in the metric's eigenbasis (w = eigval^-1/2)
template <int NE>
struct RotW {
  double sw[NE], sow[NE], cw[NE];
};
template <int NE>
__device__ __forceinline__ RotW<NE> make_rot_w(const ConArgs& A, const WaveCtx& w, double abs_t) {
  RotW<NE> r;
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int i = w.lane + 64 * e;
    const double om = (A.omega && i < w.dim) ? A.omega[i] : 1.0;
    double sn, cs;
    sincos(om * abs_t, &sn, &cs);
    r.sw[e] = sn * om;
    r.sow[e] = sn / om;
    r.cw[e] = cs;
  }
  return r;
}
// V^T x / V x (dense metric: A.eigvec holds V then V^T, both row-major) or x itself (identity / diagonal metric: V = I)
template <int NE>
__device__ __forceinline__ Vec<NE> to_eig_w(const ConArgs& A, const WaveCtx& w, const Vec<NE>& x) {
  return A.metric_kind == MM_METRIC_DENSE ? dense_walk<NE>(A.eigvec, w, x) : x;
}
template <int NE>
__device__ __forceinline__ Vec<NE> from_eig_w(const ConArgs& A, const WaveCtx& w, const Vec<NE>& x) {
  return A.metric_kind == MM_METRIC_DENSE ? dense_walk<NE>(A.eigvec + (size_t)w.dim * w.dim, w, x) : x;
}
// V diag(coef) V^T x   (EigendecomposedSymmetricMatrix @ x, matrices.py:1572-1573)
template <int NE>
__device__ __forceinline__ Vec<NE> eig_apply_w(const ConArgs& A, const WaveCtx& w, const double (&coef)[NE], const Vec<NE>& x) {
  Vec<NE> y = to_eig_w<NE>(A, w, x);
#pragma unroll
  for (int e = 0; e < NE; ++e) y.v[e] *= coef[e];
  return from_eig_w<NE>(A, w, y);
}
// h2_flow over sgn * |t|: pos += t M^-1 mom (systems.py:362-363), or the exact rotation (systems.py:464-474)
template <int NE, bool GAUSS>
__device__ __forceinline__ void h2_flow_w(const ConArgs& A, const WaveCtx& w, const RotW<NE>& rot, Vec<NE>& q, Vec<NE>& p,
                                          double t, double sgn) {
  if constexpr (!GAUSS) {
    q = axpy<NE>(t, minv1<NE>(A, w, p), q);
  } else {
    const Vec<NE> a = to_eig_w<NE>(A, w, q), b = to_eig_w<NE>(A, w, p);
    Vec<NE> na, nb;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      na.v[e] = rot.cw[e] * a.v[e] + (sgn * rot.sw[e]) * b.v[e];
      nb.v[e] = rot.cw[e] * b.v[e] - (sgn * rot.sow[e]) * a.v[e];
    }
    q = from_eig_w<NE>(A, w, na);
    p = from_eig_w<NE>(A, w, nb);
  }
}
// dh2_flow_dmom(|t|)[0] applied to every row of J, without its scalar factor: M^-1 J_b^T (the caller multiplies by |t|,
// systems.py:794-799), or V diag(sin(w|t|) w) V^T J_b^T (systems.py:1163-1176)
template <int C, int NE, bool GAUSS>
__device__ __forceinline__ Col<C, NE> flow_rows_w(const ConArgs& A, const WaveCtx& w, const RotW<NE>& rot, const Col<C, NE>& j) {
  Col<C, NE> o;
#pragma unroll
  for (int k = 0; k < C; ++k) {
    if constexpr (GAUSS) o.r[k] = eig_apply_w<NE>(A, w, rot.sw, j.r[k]);
    else o.r[k] = minv1<NE>(A, w, j.r[k]);
  }
  return o;
}
// Gram-type inverse: Cholesky (matrices.py:1161-1188), or - Gaussian split - the eigendecomposed symmetric inverse the
// reference uses there (constrained_core.h gram_inverse; its all_finite test is the reference's "Array is not finite.")
template <int C, bool GAUSS>
__device__ __forceinline__ bool gram_inverse_w(const CMat<C>& g, CMat<C>* inv, double* ld) {
  if (!all_finite<C>(g)) return false;
  if constexpr (GAUSS) {
    sym_inverse<C>(g, inv, ld);
    return true;
  } else {
    return chol_inverse<C>(g, inv, ld);
  }
}

// this lane's columns of jacob_constr(q)
template <int C, int NE>
__device__ __forceinline__ Col<C, NE> jacob_w(const ConArgs& A, const WaveCtx& w, const Vec<NE>& q) {
  Col<C, NE> j;
#pragma unroll
  for (int k = 0; k < C; ++k) j.r[k] = vzero<NE>();
#ifdef MM_RTC_BUILD
  if (A.constr == MM_CONSTR_USER) {
    // the user's jacob_constr fills a whole C x D array (systems.py:786-792 hands back a dense matrix): ONE lane runs it on
    // the position in LDS, every lane then takes its columns
    publish<NE>(w.nat, w.lane, q);
    if (w.lane == 0) ::mm_user_jacob(w.nat, w.dim, A.cparams, w.ujac);
    wave_sync();
#pragma unroll
    for (int k = 0; k < C; ++k)
#pragma unroll
      for (int e = 0; e < NE; ++e) j.r[k].v[e] = w.lane + 64 * e < w.dim ? w.ujac[k * w.dim + w.lane + 64 * e] : 0.0;
    wave_sync();
    return j;
  }
#endif
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int i = w.lane + 64 * e;
    const bool in = i < w.dim;
    if (A.constr == MM_CONSTR_LINEAR) {
#pragma unroll
      for (int k = 0; k < C; ++k) j.r[k].v[e] = in ? A.cparams[k * w.dim + i] : 0.0;
    } else if (A.constr == MM_CONSTR_SPHERE) {
      j.r[0].v[e] = 2.0 * q.v[e];
    } else if (A.constr == MM_CONSTR_SPHERE_PLANE) {
      j.r[0].v[e] = 2.0 * q.v[e];
      if constexpr (C > 1) j.r[1].v[e] = in ? A.cparams[i] : 0.0;
    } else if (A.constr == MM_CONSTR_CIRCLE) {
      j.r[0].v[e] = i < 2 ? 2.0 * q.v[e] : 0.0;
    } else {  // MM_CONSTR_FIRST
      j.r[0].v[e] = i == 0 ? 1.0 : 0.0;
    }
  }
  return j;
}

// constr(q)
template <int C, int NE>
__device__ __forceinline__ CVec<C> constr_w(const ConArgs& A, const WaveCtx& w, const Vec<NE>& q) {
  CVec<C> c;
#ifdef MM_RTC_BUILD
  if (A.constr == MM_CONSTR_USER) {  // every lane evaluates the user's constr on the position in LDS (C values, O(D) reads)
    publish<NE>(w.nat, w.lane, q);
    ::mm_user_constr(w.nat, w.dim, A.cparams, c.v);
    wave_sync();
    return c;
  }
#endif
  double t[C];
#pragma unroll
  for (int k = 0; k < C; ++k) t[k] = 0.0;
  if (A.constr == MM_CONSTR_LINEAR) {
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int i = w.lane + 64 * e;
      if (i < w.dim) {
#pragma unroll
        for (int k = 0; k < C; ++k) t[k] = __builtin_fma(A.cparams[k * w.dim + i], q.v[e], t[k]);
      }
    }
    reduce_many<C>(w, t);
#pragma unroll
    for (int k = 0; k < C; ++k) c.v[k] = t[k] - A.cparams[C * w.dim + k];
    return c;
  }
  if (A.constr == MM_CONSTR_SPHERE_PLANE) {
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int i = w.lane + 64 * e;
      t[0] = __builtin_fma(q.v[e], q.v[e], t[0]);
      if constexpr (C > 1) t[1] = i < w.dim ? __builtin_fma(A.cparams[i], q.v[e], t[1]) : t[1];
    }
    reduce_many<C>(w, t);
    c.v[0] = t[0] - 1.0;
    if constexpr (C > 1) c.v[1] = t[1];
#pragma unroll
    for (int k = 2; k < C; ++k) c.v[k] = 0.0;
    return c;
  }
#pragma unroll
  for (int k = 0; k < C; ++k) c.v[k] = 0.0;
  if (A.constr == MM_CONSTR_SPHERE) {
    double s = 0.0;
#pragma unroll
    for (int e = 0; e < NE; ++e) s = __builtin_fma(q.v[e], q.v[e], s);
    c.v[0] = wave_sum(s) - 1.0;
  } else if (A.constr == MM_CONSTR_CIRCLE) {
    c.v[0] = wave_sum(w.lane < 2 ? q.v[0] * q.v[0] : 0.0) - 1.0;
  } else {
    c.v[0] = readlane_f64(q.v[0], 0);  // MM_CONSTR_FIRST
  }
  return c;
}

// g[a][b] = scale * sum_i x[a]_i y[b]_i
template <int C, int NE>
__device__ __forceinline__ CMat<C> rows_inner_w(const WaveCtx& w, const Col<C, NE>& x, const Col<C, NE>& y, double scale) {
  double t[C * C];
#pragma unroll
  for (int a = 0; a < C; ++a)
#pragma unroll
    for (int b = 0; b < C; ++b) {
      double s = x.r[a].v[0] * y.r[b].v[0];
#pragma unroll
      for (int e = 1; e < NE; ++e) s = __builtin_fma(x.r[a].v[e], y.r[b].v[e], s);
      t[a * C + b] = s;
    }
  reduce_many<C * C>(w, t);
  CMat<C> g;
#pragma unroll
  for (int a = 0; a < C; ++a)
#pragma unroll
    for (int b = 0; b < C; ++b) g.m[a][b] = t[a * C + b] * scale;
  return g;
}

// rows^T x: this lane's coordinates of sum_b x_b rows[b]
template <int C, int NE>
__device__ __forceinline__ Vec<NE> combine(const Col<C, NE>& rows, const CVec<C>& x) {
  Vec<NE> o;
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    double s = rows.r[0].v[e] * x.v[0];
#pragma unroll
    for (int b = 1; b < C; ++b) s += rows.r[b].v[e] * x.v[b];
    o.v[e] = s;
  }
  return o;
}

// rows x: the C sums sum_i rows[a]_i x_i
template <int C, int NE>
__device__ __forceinline__ CVec<C> rows_times(const WaveCtx& w, const Col<C, NE>& rows, const Vec<NE>& x) {
  double t[C];
#pragma unroll
  for (int a = 0; a < C; ++a) {
    double s = rows.r[a].v[0] * x.v[0];
#pragma unroll
    for (int e = 1; e < NE; ++e) s = __builtin_fma(rows.r[a].v[e], x.v[e], s);
    t[a] = s;
  }
  reduce_many<C>(w, t);
  CVec<C> o;
#pragma unroll
  for (int a = 0; a < C; ++a) o.v[a] = t[a];
  return o;
}

// mom - J^T (J M^-1 J^T)^-1 J M^-1 mom     (systems.py:863-873)
template <int C, int NE, bool GAUSS = false>
__device__ __forceinline__ bool project_cotangent_w(const ConArgs& A, const WaveCtx& w, Vec<NE>& p, const Col<C, NE>& jac) {
  const CMat<C> gram = rows_inner_w<C, NE>(w, jac, minv_rows_w<C, NE>(A, w, jac), 1.0);
  CMat<C> inv;
  double ld;
  if (!gram_inverse_w<C, GAUSS>(gram, &inv, &ld)) return false;
  const CVec<C> jm = rows_times<C, NE>(w, jac, minv1<NE>(A, w, p));
  p = vsub<NE>(p, combine<C, NE>(jac, cmat_vec<C>(inv, jm)));
  return true;
}

// grad_neg_log_dens (the position goes through LDS for targets that couple coordinates)
template <int NE>
__device__ __forceinline__ Vec<NE> grad_w(const ConArgs& A, const WaveCtx& w, const Vec<NE>& q) {
  publish<NE>(w.nat, w.lane, q);
  const TargetAux aux;  // no wave-collective targets here (the funnel is rejected on the host)
  Vec<NE> g;
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int i = w.lane + 64 * e;
    g.v[e] = i < w.dim ? target_grad_elem(A.target, aux, w.nat, i, w.dim, A.tparams) : 0.0;
  }
  wave_sync();
  return g;
}

// dh1_dpos (systems.py:858-862): grad_neg_log_dens, plus for dens_wrt_hausdorff=False
// grad_log_det_sqrt_gram = mhp_constr(inv_gram J M^-1) (systems.py:1024-1031).  false = LinAlgError.
// The built-in constraints' Hessians are constant multiples of (part of) the identity: sum_k m[k] * H_k picks
// 2 m[0] on the coordinates the quadratic constraint involves, nothing for the linear ones.
template <int C, int NE, bool GAUSS = false>
__device__ __forceinline__ bool dh1_dpos_w(const ConArgs& A, const WaveCtx& w, const Vec<NE>& q, Vec<NE>* out) {
  Vec<NE> g = grad_w<NE>(A, w, q);
  if (A.ambient) {
    const Col<C, NE> jac = jacob_w<C, NE>(A, w, q);
    const CMat<C> gram = rows_inner_w<C, NE>(w, jac, minv_rows_w<C, NE>(A, w, jac), 1.0);
    CMat<C> inv;
    double ld;
    if (!gram_inverse_w<C, GAUSS>(gram, &inv, &ld)) return false;
#if defined(MM_RTC_BUILD) && defined(MM_USER_HAS_MHP)
    if (A.constr == MM_CONSTR_USER) {
      // mhp_constr(q)(inv_gram J M^-1) (systems.py:1006-1008, 1024-1031): the C x D operand into LDS (every lane its
      // columns), one lane runs the user's hook, every lane takes its coordinates of the result
#pragma unroll
      for (int a = 0; a < C; ++a) {
        Vec<NE> ma;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
          double s = inv.m[a][0] * jac.r[0].v[e];
#pragma unroll
          for (int b = 1; b < C; ++b) s += inv.m[a][b] * jac.r[b].v[e];
          ma.v[e] = s;
        }
        ma = minv1<NE>(A, w, ma);
#pragma unroll
        for (int e = 0; e < NE; ++e)
          if (w.lane + 64 * e < w.dim) w.ujac[a * w.dim + w.lane + 64 * e] = ma.v[e];
      }
      publish<NE>(w.nat, w.lane, q);
      if (w.lane == 0) ::mm_user_mhp_constr(w.nat, w.dim, A.cparams, w.ujac, w.uout);
      wave_sync();
#pragma unroll
      for (int e = 0; e < NE; ++e)
        if (w.lane + 64 * e < w.dim) g.v[e] += w.uout[w.lane + 64 * e];
      wave_sync();
      *out = g;
      return true;
    }
#endif
    Vec<NE> m0;  // row 0 of inv_gram @ J: only it meets a non-zero constraint Hessian
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      double s = inv.m[0][0] * jac.r[0].v[e];
#pragma unroll
      for (int b = 1; b < C; ++b) s += inv.m[0][b] * jac.r[b].v[e];
      m0.v[e] = s;
    }
    m0 = minv1<NE>(A, w, m0);
    if (A.constr == MM_CONSTR_SPHERE || A.constr == MM_CONSTR_SPHERE_PLANE) {
      g = axpy<NE>(2.0, m0, g);
    } else if (A.constr == MM_CONSTR_CIRCLE) {
      g.v[0] += w.lane < 2 ? 2.0 * m0.v[0] : 0.0;
    }
  }
  *out = g;
  return true;
}

// The three projection solvers (solvers.py:429-469, 303-343, 561-614) on the lane-distributed state.
template <int C, int NE, bool GAUSS = false>
__device__ __forceinline__ int project_w(const ConArgs& A, const WaveCtx& w, const RotW<NE>& rot, Vec<NE>& q, Vec<NE>& p,
                                         const Col<C, NE>& jac_prev, double t, Col<C, NE>* jac_out, long long* n_iters) {
  const mm_proj_opts& o = A.opts;
  const double abs_t = GAUSS ? 1.0 : fabs(t);  // the Gaussian flow matrices carry |t| themselves
  const double sgn = (t > 0.0) ? 1.0 : ((t < 0.0) ? -1.0 : 0.0);
  const Col<C, NE> mjp = flow_rows_w<C, NE, GAUSS>(A, w, rot, jac_prev);
  // momentum update at convergence: mom -= sign(t) dh2_flow_mom_dmom @ mu (the identity, or V diag(cos(w|t|)) V^T)
  auto apply_mu = [&](const Vec<NE>& m) {
    if constexpr (GAUSS) p = axpy<NE>(-sgn, eig_apply_w<NE>(A, w, rot.cw, m), p);
    else p = axpy<NE>(-sgn, m, p);
  };
  Vec<NE> mu = vzero<NE>();
  if (o.solver == MM_PROJ_QUASI_NEWTON) {
    CMat<C> inv;
    double ld;
    const CMat<C> g0 = rows_inner_w<C, NE>(w, jac_prev, mjp, abs_t);
    if (!gram_inverse_w<C, GAUSS>(g0, &inv, &ld)) return MM_ST_LINALG;
    for (int it = 0; it < o.max_iters; ++it) {
      *n_iters += 1;
      const CVec<C> c = constr_w<C, NE>(A, w, q);
      const double err = cnorm<C>(c, o.norm);
      const CVec<C> x = cmat_vec<C>(inv, c);
      const Vec<NE> dmu = combine<C, NE>(jac_prev, x);
      const Vec<NE> dpos = scaled<NE>(abs_t, combine<C, NE>(mjp, x));
      if (err > o.div_tol || err != err) return MM_ST_DIVERGED;
      if (err < o.constr_tol && wnorm<NE>(dpos, o.norm) < o.pos_tol) {
        apply_mu(mu);
        *jac_out = jacob_w<C, NE>(A, w, q);
        return MM_ST_OK;
      }
      mu = axpy<NE>(1.0, dmu, mu);
      q = vsub<NE>(q, dpos);
    }
    return MM_ST_MAX_ITERS;
  }
  if (o.solver == MM_PROJ_NEWTON_LINE_SEARCH) {
    Vec<NE> dpos = vzero<NE>();
    double step = 0.0;
    for (int it = 0; it < o.max_iters; ++it) {
      *n_iters += 1;
      const Col<C, NE> jac = jacob_w<C, NE>(A, w, q);
      const CVec<C> c = constr_w<C, NE>(A, w, q);
      const double err = cnorm<C>(c, o.norm);
      if (it > 0 && (err > o.div_tol || err != err)) return MM_ST_DIVERGED;
      const bool small_step = (it == 0) || wnorm<NE>(scaled<NE>(step, dpos), o.norm) < o.pos_tol;
      if (err < o.constr_tol && small_step) {
        apply_mu(mu);
        *jac_out = jac;
        return MM_ST_OK;
      }
      const CMat<C> a = rows_inner_w<C, NE>(w, jac, mjp, abs_t);
      if (!all_finite<C>(a)) return MM_ST_SOLVER_LINALG;
      const CVec<C> x = lu_solve<C>(a, c);
      const Vec<NE> dmu = combine<C, NE>(jac_prev, x);
      dpos = scaled<NE>(-1.0, scaled<NE>(abs_t, combine<C, NE>(mjp, x)));
      const Vec<NE> q_curr = q;
      step = 1.0;
      for (int ls = 0; ls < o.max_line_search_iters; ++ls) {
        q = axpy<NE>(step, dpos, q_curr);
        const double new_err = cnorm<C>(constr_w<C, NE>(A, w, q), o.norm);
        if (new_err < err) break;
        step *= 0.5;
      }
      mu = axpy<NE>(step, dmu, mu);
    }
    return MM_ST_MAX_ITERS;
  }
  for (int it = 0; it < o.max_iters; ++it) {  // Newton
    *n_iters += 1;
    const Col<C, NE> jac = jacob_w<C, NE>(A, w, q);
    const CVec<C> c = constr_w<C, NE>(A, w, q);
    const double err = cnorm<C>(c, o.norm);
    const CMat<C> a = rows_inner_w<C, NE>(w, jac, mjp, abs_t);
    if (!all_finite<C>(a)) return MM_ST_SOLVER_LINALG;  // "Array is not finite." inside the solver
    const CVec<C> x = lu_solve<C>(a, c);
    const Vec<NE> dmu = combine<C, NE>(jac_prev, x);
    const Vec<NE> dpos = scaled<NE>(abs_t, combine<C, NE>(mjp, x));
    if (err > o.div_tol || err != err) return MM_ST_DIVERGED;
    if (err < o.constr_tol && wnorm<NE>(dpos, o.norm) < o.pos_tol) {
      apply_mu(mu);
      *jac_out = jac;
      return MM_ST_OK;
    }
    mu = axpy<NE>(1.0, dmu, mu);
    q = vsub<NE>(q, dpos);
  }
  return MM_ST_MAX_ITERS;
}

// UC: constraints of a USER constraint (0: built in - no scratch for the hooks)
template <int NE, int UC = 0>
__device__ __forceinline__ WaveCtx make_ctx(double* lds, int wave, int lane, int dim) {
  constexpr int per_wave = UC ? wave_lds_user<UC, NE>() : wave_lds<NE>();
  double* wl = lds + wave * per_wave;
  double* ujac = UC ? wl + wave_lds<NE>() : nullptr;
  return WaveCtx{wl, wl + 64 * kRowStride, wl + 64 * kRowStride + 64, wl + 64 * kRowStride + 64 + 64 * NE, lane, dim,
                 ujac, UC ? ujac + UC * 64 * NE : nullptr};
}
// chains per workgroup of a run-time compiled translation unit around a user constraint: what fits 150 KB of LDS
template <int C, int NE>
constexpr int waves_per_block_user() {
  return 4 * wave_lds_user<C, NE>() * 8 <= 150 * 1024 ? 4 : (2 * wave_lds_user<C, NE>() * 8 <= 150 * 1024 ? 2 : 1);
}

// W: chains (waves) per workgroup; UC: see make_ctx
template <int C, int NE, bool GAUSS, int W, int UC>
__device__ __forceinline__ void constrained_wave_body(const ConArgs& A, double* lds) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t chain = (int64_t)blockIdx.x * W + wave;
  if (chain >= A.n_chains) return;  // no block-level barrier in this kernel
  const int dim = A.dim;
  const WaveCtx w = make_ctx<NE, UC>(lds, wave, lane, dim);
  Vec<NE> q = load_vec<NE>(A.pos + chain * dim, lane, dim);
  Vec<NE> p = load_vec<NE>(A.mom + chain * dim, lane, dim);
  const double t = signed_step(A.dir, A.step_scale, chain, A.step_size);
  const int n_inner = A.opts.n_inner;
  const double t_in = t / n_inner;
  long long n_newton = 0, n_grad = 0;
  int status = MM_ST_OK, done = 0;

  RotW<NE> rot{};
  if constexpr (GAUSS) rot = make_rot_w<NE>(A, w, fabs(t_in));
  Vec<NE> g;  // cached dh1_dpos at the current position
  if (!dh1_dpos_w<C, NE, GAUSS>(A, w, q, &g)) status = MM_ST_LINALG;
  Col<C, NE> jac = jacob_w<C, NE>(A, w, q);
  ++n_grad;
  const int my_steps = chain_steps(A.chain_steps, chain, A.n_steps);
  for (int s = 0; s < my_steps && status == MM_ST_OK; ++s) {
    Vec<NE> qs = q, ps = p, gs = g;
    Col<C, NE> js = jac;
    // ---- A(t/2): h1_flow then cotangent projection                    integrators.py:947-949
    ps = axpy<NE>(-0.5 * t, g, ps);
    if (!project_cotangent_w<C, NE, GAUSS>(A, w, ps, js)) { status = MM_ST_LINALG; break; }
    // ---- B(t): n_inner retractions + reversibility checks              integrators.py:951-979
    for (int inn = 0; inn < n_inner && status == MM_ST_OK; ++inn) {
      const Vec<NE> q_prev = qs;
      const Col<C, NE> j_prev = js;
      h2_flow_w<NE, GAUSS>(A, w, rot, qs, ps, t_in, t_in < 0.0 ? -1.0 : 1.0);
      Col<C, NE> j_new;
      status = project_w<C, NE, GAUSS>(A, w, rot, qs, ps, j_prev, t_in, &j_new, &n_newton);
      if (status != MM_ST_OK) break;
      if (inn == n_inner - 1) {  // pre-evaluated dh1_dpos, integrators.py:956-969
        if (!dh1_dpos_w<C, NE, GAUSS>(A, w, qs, &gs)) { status = MM_ST_LINALG; break; }
        ++n_grad;
      }
      if (!project_cotangent_w<C, NE, GAUSS>(A, w, ps, j_new)) { status = MM_ST_LINALG; break; }
      // reversibility check on a copy                                    integrators.py:971-979
      Vec<NE> qb = qs, pb = ps;
      Col<C, NE> j_tmp;
      h2_flow_w<NE, GAUSS>(A, w, rot, qb, pb, -t_in, t_in < 0.0 ? 1.0 : -1.0);
      status = project_w<C, NE, GAUSS>(A, w, rot, qb, pb, j_new, -t_in, &j_tmp, &n_newton);
      if (status != MM_ST_OK) break;
      if (wnorm<NE>(vsub<NE>(qb, q_prev), A.opts.rev_norm) > A.opts.rev_tol) { status = MM_ST_NON_REVERSIBLE; break; }
      js = j_new;
    }
    if (status != MM_ST_OK) break;
    // ---- A(t/2)
    ps = axpy<NE>(-0.5 * t, gs, ps);
    if (!project_cotangent_w<C, NE, GAUSS>(A, w, ps, js)) { status = MM_ST_LINALG; break; }
    q = qs; p = ps; jac = js; g = gs;
    ++done;
  }
  store_vec<NE>(A.pos + chain * dim, lane, dim, q);
  store_vec<NE>(A.mom + chain * dim, lane, dim, p);
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

// which == 1: project_onto_cotangent_space of the momenta (systems.py:863-873) - what sample_momentum calls after the
// draw;  which == 2: out[chain] += log_det_sqrt_gram(pos) (systems.py:829-856), NaN where the Gram matrix is not
// positive definite
template <int C, int NE, bool GAUSS, int W, int UC>
__device__ __forceinline__ void constrained_aux_wave_body(const ConArgs& A, int which, double* __restrict__ out, double* lds) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t chain = (int64_t)blockIdx.x * W + wave;
  if (chain >= A.n_chains) return;
  const int dim = A.dim;
  const WaveCtx w = make_ctx<NE, UC>(lds, wave, lane, dim);
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
  const Vec<NE> q = load_vec<NE>(A.pos + chain * dim, lane, dim);
  const Col<C, NE> jac = jacob_w<C, NE>(A, w, q);
  if (which == 1) {
    Vec<NE> p = load_vec<NE>(A.mom + chain * dim, lane, dim);
    const bool ok = project_cotangent_w<C, NE, GAUSS>(A, w, p, jac);
    if (!ok) {
#pragma unroll
      for (int e = 0; e < NE; ++e) p.v[e] = nan;
    }
    store_vec<NE>(A.mom + chain * dim, lane, dim, p);
    return;
  }
  const CMat<C> gram = rows_inner_w<C, NE>(w, jac, minv_rows_w<C, NE>(A, w, jac), 1.0);
  CMat<C> inv;
  double ld;
  const bool ok = gram_inverse_w<C, GAUSS>(gram, &inv, &ld);
  if (lane == 0) out[chain] += ok ? 0.5 * ld : nan;
}

#ifndef MM_RTC_BUILD  // the in-tree instantiations (a run-time translation unit defines extern "C" wrappers instead)
template <int C, int NE, bool GAUSS = false>
__global__ __launch_bounds__(64 * waves_per_block<NE>()) void constrained_wave_kernel(ConArgs A) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constrained_wave_body<C, NE, GAUSS, waves_per_block<NE>(), 0>(A, lds);
}
template <int C, int NE, bool GAUSS = false>
__global__ __launch_bounds__(64 * waves_per_block<NE>()) void constrained_aux_wave_kernel(ConArgs A, int which,
                                                                                         double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  constrained_aux_wave_body<C, NE, GAUSS, waves_per_block<NE>(), 0>(A, which, out, lds);
}
#endif

}  // namespace mmconw
