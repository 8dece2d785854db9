// Core of the constrained leapfrog kernels: shared by k_constrained.hip (exact D <= 8 instantiations) and
// k_constrained_wide.hip (capacity-16 instantiations for 8 < D <= 16).  See k_constrained.hip for the list of
// reference functions replaced.
#pragma once
#include "mm_device.h"

#ifdef MM_RTC_BUILD
// c[k] = constraint k at q;  jac[k * dim + i] = d c_k / d q_i;  out[i] = sum_{k,j} m[k * dim + j] d2 c_k / dq_j dq_i
// (mhp_constr, systems.py:1006-1008: only systems with dens_wrt_hausdorff=False need it)
__device__ void mm_user_constr(const double* q, int dim, const double* params, double* c);
__device__ void mm_user_jacob(const double* q, int dim, const double* params, double* jac);
#ifdef MM_USER_HAS_MHP
__device__ void mm_user_mhp_constr(const double* q, int dim, const double* params, const double* m, double* out);
#endif
#endif

namespace mmcon {

// Loops over the D coordinates are fully unrolled for the exact kernels (D <= 8: every vector lives in
// registers) and left rolled for the capacity-16 ones, whose per-chain arrays then sit in scratch: those exist
// for coverage, and unrolling them costs half an hour of compile time.
template <int D>
constexpr int kUnrollD = D > 8 ? 1 : D;

struct ConArgs {
  double* pos;
  double* mom;
  const int8_t* dir;
  const double* step_scale;
  const int32_t* chain_steps;
  int32_t* status;
  int32_t* n_done;
  int64_t n_chains;
  int dim;  // the system's dimension: equals the template capacity D in the exact kernels, <= D in the padded ones
  double step_size;
  int n_steps;
  int target;
  const double* tparams;
  int metric_kind;
  const double* minv;  // diag: 1/diag[D]; dense: explicit inverse [D*D]
  int constr;
  double cp0, cp1;        // constraint params (torus: R, r)
  const double* cparams;  // all constraint params on the device (linear: A[C*D] then b[C]; sphere-plane: n[D])
  int ambient;            // dens_wrt_hausdorff=False: the Gram log-determinant term is part of h1
  int gaussian;           // Gaussian split: exact h2 rotation, symmetric (not Cholesky-factored) Gram matrices
  const double* omega;    // gaussian: 1/sqrt(eigval)[D] (nullptr for the identity metric)
  const double* eigvec;   // gaussian + dense metric: V [D*D] then V^T [D*D]
  mm_proj_opts opts;
  mm_counters* counters;
};

template <int D>
struct Vec {
  double v[D];
};

template <int C, int D>
struct Jac {  // C rows of D: a constraint Jacobian, or any C x D block
  Vec<D> r[C];
};

template <int C>
struct CVec {
  double v[C];
};

template <int C>
struct CMat {
  double m[C][C];
};

__device__ __forceinline__ bool finite(double x) { return fabs(x) <= 1.79769313486231570815e308; }

template <int D>
__device__ __forceinline__ Vec<D> target_grad(const ConArgs& A, const Vec<D>& q) {
  Vec<D> g;
  const mmdev::TargetAux aux;  // no wave-collective targets here (funnel is rejected on the host)
#pragma unroll kUnrollD<D>
  for (int i = 0; i < D; ++i)
    g.v[i] = (i < A.dim) ? mmdev::target_grad_elem(A.target, aux, q.v, i, A.dim, A.tparams) : 0.0;
  return g;
}

template <int D>
__device__ __forceinline__ Vec<D> minv_apply(const ConArgs& A, const Vec<D>& x) {
  Vec<D> y;
  if (A.metric_kind == MM_METRIC_IDENTITY) {
    y = x;
  } else if (A.metric_kind == MM_METRIC_DIAG) {
#pragma unroll kUnrollD<D>
    for (int i = 0; i < D; ++i) y.v[i] = (i < A.dim) ? A.minv[i] * x.v[i] : 0.0;
  } else {
#pragma unroll kUnrollD<D>
    for (int i = 0; i < D; ++i) {
      double s = 0.0;
#pragma unroll kUnrollD<D>
      for (int j = 0; j < D; ++j)
        if (i < A.dim && j < A.dim) s += A.minv[i * A.dim + j] * x.v[j];
      y.v[i] = s;
    }
  }
  return y;
}

template <int D>
__device__ __forceinline__ double dot(const Vec<D>& a, const Vec<D>& b) {
  double s = 0.0;
#pragma unroll kUnrollD<D>
  for (int i = 0; i < D; ++i) s += a.v[i] * b.v[i];
  return s;
}

template <int D>
__device__ __forceinline__ double vnorm(const Vec<D>& a, int kind) {
  double acc = 0.0;
#pragma unroll kUnrollD<D>
  for (int i = 0; i < D; ++i) acc = mmdev::wave_norm_accum(acc, a.v[i], kind);
  return kind == MM_NORM_LINF ? acc : sqrt(acc);
}

template <int C>
__device__ __forceinline__ double cnorm(const CVec<C>& a, int kind) {
  double acc = 0.0;
#pragma unroll
  for (int i = 0; i < C; ++i) acc = mmdev::wave_norm_accum(acc, a.v[i], kind);
  return kind == MM_NORM_LINF ? acc : sqrt(acc);
}

// rows^T x: sum_b x_b rows[b]   (J^T lambda)
template <int C, int D>
__device__ __forceinline__ Vec<D> rows_combine(const Jac<C, D>& rows, const CVec<C>& x) {
  Vec<D> y;
#pragma unroll kUnrollD<D>
  for (int i = 0; i < D; ++i) {
    double s = rows.r[0].v[i] * x.v[0];
#pragma unroll
    for (int b = 1; b < C; ++b) s += rows.r[b].v[i] * x.v[b];
    y.v[i] = s;
  }
  return y;
}

// ---- Gaussian split: per-chain constants for the chain's |inner time step| -----------------------------
template <int D>
struct Rot {
  double sw[D], sow[D], cw[D];  // sin(w|t|) w, sin(w|t|)/w, cos(w|t|)
};

template <int D>
__device__ __forceinline__ Rot<D> make_rot(const ConArgs& A, double abs_t) {
  Rot<D> r;
#pragma unroll kUnrollD<D>
  for (int i = 0; i < D; ++i) {
    const double om = (A.omega && i < A.dim) ? A.omega[i] : 1.0;
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
#pragma unroll kUnrollD<D>
  for (int i = 0; i < D; ++i) {
    double s = 0.0;
#pragma unroll kUnrollD<D>
    for (int j = 0; j < D; ++j)
      if (i < A.dim && j < A.dim) s += A.eigvec[j * A.dim + i] * x.v[j];
    y.v[i] = s;
  }
  return y;
}

template <int D>
__device__ __forceinline__ Vec<D> from_eigenbasis(const ConArgs& A, const Vec<D>& x) {
  if (A.metric_kind != MM_METRIC_DENSE) return x;
  Vec<D> y;
#pragma unroll kUnrollD<D>
  for (int i = 0; i < D; ++i) {
    double s = 0.0;
#pragma unroll kUnrollD<D>
    for (int j = 0; j < D; ++j)
      if (i < A.dim && j < A.dim) s += A.eigvec[i * A.dim + j] * x.v[j];
    y.v[i] = s;
  }
  return y;
}

// V diag(coef) V^T x: EigendecomposedSymmetricMatrix @ x (matrices.py:1572-1573)
template <int D>
__device__ __forceinline__ Vec<D> eig_apply(const ConArgs& A, const double (&coef)[D], const Vec<D>& x) {
  Vec<D> y = to_eigenbasis<D>(A, x);
#pragma unroll kUnrollD<D>
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
#pragma unroll kUnrollD<D>
    for (int i = 0; i < D; ++i) q.v[i] += t * v.v[i];
    return;
  }
  const Vec<D> a = to_eigenbasis<D>(A, q), b = to_eigenbasis<D>(A, p);
  Vec<D> na, nb;
#pragma unroll kUnrollD<D>
  for (int i = 0; i < D; ++i) {
    na.v[i] = rot.cw[i] * a.v[i] + (sgn * rot.sw[i]) * b.v[i];
    nb.v[i] = rot.cw[i] * b.v[i] - (sgn * rot.sow[i]) * a.v[i];
  }
  q = from_eigenbasis<D>(A, na);
  p = from_eigenbasis<D>(A, nb);
}

// dh2_flow_dmom(|t|)[0] applied to every row of J, without its scalar factor: Euclidean M^-1 J_b^T (the
// caller multiplies by |t|, systems.py:794-799); Gaussian V diag(sin(w|t|) w) V^T J_b^T (systems.py:1163-1176).
template <int C, int D>
__device__ __forceinline__ Jac<C, D> flow_pos_dmom_rows(const ConArgs& A, const Rot<D>& rot, const Jac<C, D>& j) {
  Jac<C, D> out;
#pragma unroll
  for (int b = 0; b < C; ++b) out.r[b] = A.gaussian ? eig_apply<D>(A, rot.sw, j.r[b]) : minv_apply<D>(A, j.r[b]);
  return out;
}

template <int C, int D>
__device__ __forceinline__ Jac<C, D> minv_rows(const ConArgs& A, const Jac<C, D>& j) {
  Jac<C, D> out;
#pragma unroll
  for (int b = 0; b < C; ++b) out.r[b] = minv_apply<D>(A, j.r[b]);
  return out;
}

// ---- constraint functions: built in, or - in a translation unit compiled at run time around the user's source
// (mm_rtc.hip, MM_RTC_BUILD) - the user's `constr` / `jacob_constr` (systems.py:786-792) --------------------------
template <int C, int D>
__device__ __forceinline__ CVec<C> constr_value(const ConArgs& A, const Vec<D>& q) {
  CVec<C> c;
#pragma unroll
  for (int k = 0; k < C; ++k) c.v[k] = 0.0;
#ifdef MM_RTC_BUILD
  if (A.constr == MM_CONSTR_USER) {
    ::mm_user_constr(q.v, A.dim, A.cparams, c.v);
    return c;
  }
#endif
  if (A.constr == MM_CONSTR_LINEAR) {  // A q - b
#pragma unroll
    for (int k = 0; k < C; ++k) {
      double s = 0.0;
#pragma unroll kUnrollD<D>
      for (int j = 0; j < D; ++j)
        if (j < A.dim) s += A.cparams[k * A.dim + j] * q.v[j];
      c.v[k] = s - A.cparams[C * A.dim + k];
    }
    return c;
  }
  if constexpr (C == 2) {  // sphere-plane: |q|^2 - 1, n . q
    c.v[0] = dot<D>(q, q) - 1.0;
    double s = 0.0;
#pragma unroll kUnrollD<D>
    for (int j = 0; j < D; ++j)
      if (j < A.dim) s += A.cparams[j] * q.v[j];
    c.v[1] = s;
  }
  if constexpr (C == 1) {
    if (A.constr == MM_CONSTR_TORUS) {
      constexpr int I1 = D > 1 ? 1 : 0;
      double rho, irho;
      mmdev::sqrt_rsqrt(q.v[0] * q.v[0] + q.v[I1] * q.v[I1], &rho, &irho);
      const double dr = rho - A.cp0;
      c.v[0] = dr * dr + q.v[D > 2 ? 2 : 0] * q.v[D > 2 ? 2 : 0] - A.cp1 * A.cp1;
    } else if (A.constr == MM_CONSTR_FIRST) {
      c.v[0] = q.v[0];
    } else if (A.constr == MM_CONSTR_SPHERE) {
      c.v[0] = dot<D>(q, q) - 1.0;
    } else {
      c.v[0] = q.v[0] * q.v[0] + q.v[D > 1 ? 1 : 0] * q.v[D > 1 ? 1 : 0] - 1.0;  // circle
    }
  }
  return c;
}

template <int C, int D>
__device__ __forceinline__ Jac<C, D> constr_jacob(const ConArgs& A, const Vec<D>& q) {
  Jac<C, D> j;
#pragma unroll
  for (int k = 0; k < C; ++k)
#pragma unroll kUnrollD<D>
    for (int i = 0; i < D; ++i) j.r[k].v[i] = 0.0;
#ifdef MM_RTC_BUILD
  if (A.constr == MM_CONSTR_USER) {
    double jj[C * D];
    ::mm_user_jacob(q.v, A.dim, A.cparams, jj);
#pragma unroll
    for (int k = 0; k < C; ++k)
#pragma unroll kUnrollD<D>
      for (int i = 0; i < D; ++i) j.r[k].v[i] = (i < A.dim) ? jj[k * A.dim + i] : 0.0;
    return j;
  }
#endif
  if (A.constr == MM_CONSTR_LINEAR) {
#pragma unroll
    for (int k = 0; k < C; ++k)
#pragma unroll kUnrollD<D>
      for (int i = 0; i < D; ++i) j.r[k].v[i] = (i < A.dim) ? A.cparams[k * A.dim + i] : 0.0;
    return j;
  }
  if constexpr (C == 2) {
#pragma unroll kUnrollD<D>
    for (int i = 0; i < D; ++i) {
      j.r[0].v[i] = 2.0 * q.v[i];
      j.r[1].v[i] = (i < A.dim) ? A.cparams[i] : 0.0;
    }
  }
  if constexpr (C == 1) {
    if (A.constr == MM_CONSTR_TORUS) {
      constexpr int I1 = D > 1 ? 1 : 0;
      double rho, irho;  // the same call as in constr_value: one evaluation serves both after inlining
      mmdev::sqrt_rsqrt(q.v[0] * q.v[0] + q.v[I1] * q.v[I1], &rho, &irho);
      const double f = 2.0 * (rho - A.cp0) * irho;
      j.r[0].v[0] = f * q.v[0];
      if constexpr (D > 1) j.r[0].v[1] = f * q.v[1];
      if constexpr (D > 2) j.r[0].v[2] = 2.0 * q.v[2];
    } else if (A.constr == MM_CONSTR_FIRST) {
      j.r[0].v[0] = 1.0;
    } else if (A.constr == MM_CONSTR_SPHERE) {
#pragma unroll kUnrollD<D>
      for (int i = 0; i < D; ++i) j.r[0].v[i] = 2.0 * q.v[i];
    } else {
      j.r[0].v[0] = 2.0 * q.v[0];
      if constexpr (D > 1) j.r[0].v[1] = 2.0 * q.v[1];
    }
  }
  return j;
}

// mhp_constr(state)(m): sum_{c,i} m[c][i] d2 constr_c / dq_i dq_k (systems.py:1006-1008)
template <int C, int D>
__device__ __forceinline__ Vec<D> constr_hess_apply(const ConArgs& A, const Vec<D>& q, const Jac<C, D>& m) {
  Vec<D> out;
#pragma unroll kUnrollD<D>
  for (int i = 0; i < D; ++i) out.v[i] = 0.0;
#if defined(MM_RTC_BUILD) && defined(MM_USER_HAS_MHP)
  if (A.constr == MM_CONSTR_USER) {
    double mm[C * D], oo[D];
#pragma unroll
    for (int k = 0; k < C; ++k)
#pragma unroll kUnrollD<D>
      for (int i = 0; i < D; ++i)
        if (i < A.dim) mm[k * A.dim + i] = m.r[k].v[i];
    ::mm_user_mhp_constr(q.v, A.dim, A.cparams, mm, oo);
#pragma unroll kUnrollD<D>
    for (int i = 0; i < D; ++i) out.v[i] = (i < A.dim) ? oo[i] : 0.0;
    return out;
  }
#endif
  if (A.constr == MM_CONSTR_LINEAR) return out;
  if constexpr (C == 2) {
#pragma unroll kUnrollD<D>
    for (int i = 0; i < D; ++i) out.v[i] = 2.0 * m.r[0].v[i];
  }
  if constexpr (C == 1) {
    if (A.constr == MM_CONSTR_TORUS) {
      if constexpr (D > 2) {
        const double x = q.v[0], y = q.v[1];
        const double rho = sqrt(x * x + y * y), rho2 = rho * rho, rho3 = rho2 * rho;
        const double dr = rho - A.cp0;
        const double hxx = 2.0 * x * x / rho2 + 2.0 * dr * y * y / rho3;
        const double hyy = 2.0 * y * y / rho2 + 2.0 * dr * x * x / rho3;
        const double hxy = 2.0 * x * y * A.cp0 / rho3;
        out.v[0] = hxx * m.r[0].v[0] + hxy * m.r[0].v[1];
        out.v[1] = hxy * m.r[0].v[0] + hyy * m.r[0].v[1];
        out.v[2] = 2.0 * m.r[0].v[2];
      }
    } else if (A.constr == MM_CONSTR_CIRCLE) {
      out.v[0] = 2.0 * m.r[0].v[0];
      if constexpr (D > 1) out.v[1] = 2.0 * m.r[0].v[1];
    } else if (A.constr == MM_CONSTR_SPHERE) {
#pragma unroll kUnrollD<D>
      for (int i = 0; i < D; ++i) out.v[i] = 2.0 * m.r[0].v[i];
    }
  }
  return out;
}

// ---- C x C linear algebra ----------------------------------------------------------------------------------
// g[a][b] = (x[a] . y[b]) * scale
template <int C, int D>
__device__ __forceinline__ CMat<C> rows_inner(const Jac<C, D>& x, const Jac<C, D>& y, double scale) {
  CMat<C> g;
#pragma unroll
  for (int a = 0; a < C; ++a)
#pragma unroll
    for (int b = 0; b < C; ++b) g.m[a][b] = dot<D>(x.r[a], y.r[b]) * scale;
  return g;
}

template <int C>
__device__ __forceinline__ bool all_finite(const CMat<C>& g) {
  bool ok = true;
#pragma unroll
  for (int a = 0; a < C; ++a)
#pragma unroll
    for (int b = 0; b < C; ++b) ok = ok && finite(g.m[a][b]);
  return ok;
}

// DensePositiveDefiniteMatrix: lower Cholesky factor (matrices.py:1161-1173), explicit inverse as the two
// triangular solves U Y = I, U X = Y^T with U = L^T (matrices.py:1183-1188), log|det| = 2 sum log|L_ii|.
template <int C>
__device__ __forceinline__ bool chol_inverse(const CMat<C>& g, CMat<C>* inv, double* log_abs_det) {
  if constexpr (C == 1) {  // one constraint: L = sqrt(g), inverse (1 / L) / L, log|det| = 2 log L
    const double d = g.m[0][0];
    if (!(d > 0.0)) return false;
    double l, il;
    mmdev::sqrt_rsqrt(d, &l, &il);
    inv->m[0][0] = il * il;
    *log_abs_det = 2.0 * log(l);
    return true;
  }
  CMat<C> l;
  double ril[C];
  bool ok = true;
#pragma unroll
  for (int j = 0; j < C; ++j) {
    double d = g.m[j][j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= l.m[j][k] * l.m[j][k];
    ok = ok && (d > 0.0);
    // L_jj and 1 / L_jj from one v_rsq_f64 (mm_device.h): every division by a diagonal entry below is a multiplication -
    // the IEEE expansion is ~30 dependent instructions, and a lane runs C (C + 1) of them per Gram matrix
    double ljj;
    mmdev::sqrt_rsqrt(d > 0.0 ? d : 1.0, &ljj, &ril[j]);
    l.m[j][j] = ljj;
#pragma unroll
    for (int i = j + 1; i < C; ++i) {
      double s = g.m[i][j];
#pragma unroll
      for (int k = 0; k < j; ++k) s -= l.m[i][k] * l.m[j][k];
      l.m[i][j] = s * ril[j];
    }
  }
  if (!ok) return false;
  // Y = U^-1 (upper), U[i][j] = l[j][i]; back substitution column by column
  CMat<C> y;
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int i = C - 1; i >= 0; --i) {
      double s = (i == c) ? 1.0 : 0.0;
#pragma unroll
      for (int k = i + 1; k < C; ++k) s -= l.m[k][i] * y.m[k][c];
      y.m[i][c] = s * ril[i];
    }
  // X = U^-1 Y^T
#pragma unroll
  for (int c = 0; c < C; ++c)
#pragma unroll
    for (int i = C - 1; i >= 0; --i) {
      double s = y.m[c][i];
#pragma unroll
      for (int k = i + 1; k < C; ++k) s -= l.m[k][i] * inv->m[k][c];
      inv->m[i][c] = s * ril[i];
    }
  double ld = 0.0;
#pragma unroll
  for (int i = 0; i < C; ++i) ld += log(fabs(l.m[i][i]));
  *log_abs_det = 2.0 * ld;
  return true;
}

// DenseSymmetricMatrix (Gaussian split): eigendecomposition by Jacobi rotations, inverse V diag(1/w) V^T
// (matrices.py:1446-1447), log|det| = sum log|w| (matrices.py:458-459).  No definiteness required.
template <int C>
__device__ __forceinline__ void sym_inverse(const CMat<C>& g, CMat<C>* inv, double* log_abs_det) {
  CMat<C> a = g, v;
#pragma unroll
  for (int i = 0; i < C; ++i)
#pragma unroll
    for (int j = 0; j < C; ++j) v.m[i][j] = (i == j) ? 1.0 : 0.0;
  if constexpr (C > 1) {
    for (int sweep = 0; sweep < 8; ++sweep) {
#pragma unroll
      for (int p = 0; p < C - 1; ++p)
#pragma unroll
        for (int q = p + 1; q < C; ++q) {
          const double apq = a.m[p][q];
          if (apq != 0.0) {
            const double theta = (a.m[q][q] - a.m[p][p]) / (2.0 * apq);
            const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
            const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
            for (int k = 0; k < C; ++k) {
              const double akp = a.m[k][p], akq = a.m[k][q];
              a.m[k][p] = c * akp - s * akq;
              a.m[k][q] = s * akp + c * akq;
            }
#pragma unroll
            for (int k = 0; k < C; ++k) {
              const double apk = a.m[p][k], aqk = a.m[q][k];
              a.m[p][k] = c * apk - s * aqk;
              a.m[q][k] = s * apk + c * aqk;
            }
#pragma unroll
            for (int k = 0; k < C; ++k) {
              const double vkp = v.m[k][p], vkq = v.m[k][q];
              v.m[k][p] = c * vkp - s * vkq;
              v.m[k][q] = s * vkp + c * vkq;
            }
          }
        }
    }
  }
  double ld = 0.0;
#pragma unroll
  for (int i = 0; i < C; ++i) ld += log(fabs(a.m[i][i]));
  *log_abs_det = ld;
#pragma unroll
  for (int i = 0; i < C; ++i)
#pragma unroll
    for (int j = 0; j < C; ++j) {
      double s = 0.0;
#pragma unroll
      for (int k = 0; k < C; ++k) s += v.m[i][k] * (1.0 / a.m[k][k]) * v.m[j][k];
      inv->m[i][j] = s;
    }
}

// Inverse (and log|det|) of a Gram-type matrix; false = LinAlgError ("Array is not finite." /
// "Cholesky factorisation failed.").
template <int C>
__device__ __forceinline__ bool gram_inverse(const ConArgs& A, const CMat<C>& g, CMat<C>* inv, double* log_abs_det) {
  if (!all_finite<C>(g)) return false;
  if (A.gaussian) {
    sym_inverse<C>(g, inv, log_abs_det);
    return true;
  }
  return chol_inverse<C>(g, inv, log_abs_det);
}

template <int C>
__device__ __forceinline__ CVec<C> cmat_vec(const CMat<C>& m, const CVec<C>& x) {
  CVec<C> y;
#pragma unroll
  for (int i = 0; i < C; ++i) {
    double s = m.m[i][0] * x.v[0];
#pragma unroll
    for (int j = 1; j < C; ++j) s += m.m[i][j] * x.v[j];
    y.v[i] = s;
  }
  return y;
}

// DenseSquareMatrix.inv @ c: LU with partial pivoting (getrf: first row of largest |a_ik|) then the two
// triangular solves (matrices.py:1307-1330, 1370-1376).  A zero pivot divides by zero exactly like LAPACK's
// solve after lu_factor's warning; the NaN/inf then trips the solver's divergence test.
template <int C>
__device__ __forceinline__ CVec<C> lu_solve(CMat<C> a, CVec<C> b) {
  if constexpr (C == 1) {
    CVec<C> x1;
    x1.v[0] = mmdev::fdiv(b.v[0], a.m[0][0]);
    return x1;
  }
  double rp[C];  // reciprocals of the pivots: the divisions of the elimination and of the back substitution as products
#pragma unroll
  for (int k = 0; k < C; ++k) {
    if constexpr (C > 1) {
      int piv = k;
      double best = fabs(a.m[k][k]);
#pragma unroll
      for (int i = k + 1; i < C; ++i) {
        const double v = fabs(a.m[i][k]);
        if (v > best) {
          best = v;
          piv = i;
        }
      }
#pragma unroll
      for (int i = k + 1; i < C; ++i) {
        if (piv == i) {
#pragma unroll
          for (int j = 0; j < C; ++j) {
            const double tmp = a.m[k][j];
            a.m[k][j] = a.m[i][j];
            a.m[i][j] = tmp;
          }
          const double tb = b.v[k];
          b.v[k] = b.v[i];
          b.v[i] = tb;
        }
      }
    }
    rp[k] = mmdev::rcp_nr(a.m[k][k]);  // (a zero pivot: NaN where the division gives inf - "not finite" either way)
#pragma unroll
    for (int i = k + 1; i < C; ++i) {
      const double f = a.m[i][k] * rp[k];
#pragma unroll
      for (int j = k + 1; j < C; ++j) a.m[i][j] -= f * a.m[k][j];
      b.v[i] -= f * b.v[k];
    }
  }
  CVec<C> x;
#pragma unroll
  for (int i = C - 1; i >= 0; --i) {
    double s = b.v[i];
#pragma unroll
    for (int j = i + 1; j < C; ++j) s -= a.m[i][j] * x.v[j];
    x.v[i] = s * rp[i];
  }
  return x;
}

// mom - J^T (J M^-1 J^T)^-1 J M^-1 mom     (systems.py:863-873)
template <int C, int D>
__device__ __forceinline__ bool project_cotangent(const ConArgs& A, Vec<D>& p, const Jac<C, D>& jac) {
  const CMat<C> gram = rows_inner<C, D>(jac, minv_rows<C, D>(A, jac), 1.0);
  CMat<C> inv;
  double ld;
  if (!gram_inverse<C>(A, gram, &inv, &ld)) return false;
  const Vec<D> mp = minv_apply<D>(A, p);
  CVec<C> jm;
#pragma unroll
  for (int a = 0; a < C; ++a) jm.v[a] = dot<D>(jac.r[a], mp);
  const Vec<D> corr = rows_combine<C, D>(jac, cmat_vec<C>(inv, jm));
#pragma unroll kUnrollD<D>
  for (int i = 0; i < D; ++i) p.v[i] -= corr.v[i];
  return true;
}

// dh1_dpos (systems.py:858-862): grad_neg_log_dens, plus for dens_wrt_hausdorff=False
// grad_log_det_sqrt_gram = mhp_constr(inv_gram J M^-1) (systems.py:1024-1031).  false = LinAlgError.
template <int C, int D>
__device__ __forceinline__ bool dh1_dpos(const ConArgs& A, const Vec<D>& q, Vec<D>* out) {
  Vec<D> g = target_grad<D>(A, q);
  if (A.ambient) {
    const Jac<C, D> jac = constr_jacob<C, D>(A, q);
    const CMat<C> gram = rows_inner<C, D>(jac, minv_rows<C, D>(A, jac), 1.0);
    CMat<C> inv;
    double ld;
    if (!gram_inverse<C>(A, gram, &inv, &ld)) return false;
    Jac<C, D> m;  // inv_gram @ J
#pragma unroll
    for (int a = 0; a < C; ++a)
#pragma unroll kUnrollD<D>
      for (int i = 0; i < D; ++i) {
        double s = inv.m[a][0] * jac.r[0].v[i];
#pragma unroll
        for (int b = 1; b < C; ++b) s += inv.m[a][b] * jac.r[b].v[i];
        m.r[a].v[i] = s;
      }
    const Vec<D> hm = constr_hess_apply<C, D>(A, q, minv_rows<C, D>(A, m));
#pragma unroll kUnrollD<D>
    for (int i = 0; i < D; ++i) g.v[i] += hm.v[i];
  }
  *out = g;
  return true;
}

// momentum update at convergence: mom -= sign(t) dh2_flow_mom_dmom @ mu
template <int D>
__device__ __forceinline__ void apply_mu(const ConArgs& A, const Rot<D>& rot, Vec<D>& p, const Vec<D>& mu, double t) {
  const double sgn = (t > 0.0) ? 1.0 : ((t < 0.0) ? -1.0 : 0.0);
  const Vec<D> cmu = A.gaussian ? eig_apply<D>(A, rot.cw, mu) : mu;
#pragma unroll kUnrollD<D>
  for (int i = 0; i < D; ++i) p.v[i] -= sgn * cmu.v[i];
}

// solve_projection_onto_manifold_newton (solvers.py:429-469): residual Jacobian
// J dh2_flow_pos_dmom J_prev^T LU-solved every iteration.
template <int C, int D>
__device__ __forceinline__ int newton_project(const ConArgs& A, const Rot<D>& rot, Vec<D>& q, Vec<D>& p,
                                              const Jac<C, D>& jac_prev, double t, Jac<C, D>* jac_out,
                                              long long* n_iters) {
  const mm_proj_opts& o = A.opts;
  const double abs_t = A.gaussian ? 1.0 : fabs(t);  // the Gaussian flow matrices carry |t| themselves
  const Jac<C, D> mjp = flow_pos_dmom_rows<C, D>(A, rot, jac_prev);
  Vec<D> mu;
#pragma unroll kUnrollD<D>
  for (int i = 0; i < D; ++i) mu.v[i] = 0.0;
  for (int it = 0; it < o.max_iters; ++it) {
    *n_iters += 1;
    const Jac<C, D> jac = constr_jacob<C, D>(A, q);
    const CVec<C> c = constr_value<C, D>(A, q);
    const double err = cnorm<C>(c, o.norm);
    const CMat<C> a = rows_inner<C, D>(jac, mjp, abs_t);
    const bool fin = all_finite<C>(a);  // else "Array is not finite." inside the solver
    const CVec<C> x = lu_solve<C>(a, c);
    const Vec<D> dmu = rows_combine<C, D>(jac_prev, x);
    Vec<D> dpos = rows_combine<C, D>(mjp, x);
#pragma unroll kUnrollD<D>
    for (int i = 0; i < D; ++i) dpos.v[i] = abs_t * dpos.v[i];
    // the iteration's arithmetic is straight-line; ONE data-dependent exit decides among the reference's three
    // (in its order: not finite -> LinAlgError, diverged, converged) - every exit is an exec-mask region in a
    // lane-per-chain kernel
    const bool diverged = err > o.div_tol || err != err;
    const bool converged = err < o.constr_tol && vnorm<D>(dpos, o.norm) < o.pos_tol;
    const int code = !fin ? MM_ST_SOLVER_LINALG : (diverged ? MM_ST_DIVERGED : (converged ? MM_ST_OK : -1));
    if (code >= 0) {
      if (code == MM_ST_OK) {
        apply_mu<D>(A, rot, p, mu, t);
        *jac_out = jac;
      }
      return code;
    }
#pragma unroll kUnrollD<D>
    for (int i = 0; i < D; ++i) {
      mu.v[i] += dmu.v[i];
      q.v[i] -= dpos.v[i];
    }
  }
  return MM_ST_MAX_ITERS;
}

// solve_projection_onto_manifold_quasi_newton (solvers.py:303-343): J_prev dh2_flow_pos_dmom J_prev^T
// factored once before the loop (failure = LinAlgError OUTSIDE the solver), only constr in it.
template <int C, int D>
__device__ __forceinline__ int quasi_newton_project(const ConArgs& A, const Rot<D>& rot, Vec<D>& q, Vec<D>& p,
                                                    const Jac<C, D>& jac_prev, double t, Jac<C, D>* jac_out,
                                                    long long* n_iters) {
  const mm_proj_opts& o = A.opts;
  const double abs_t = A.gaussian ? 1.0 : fabs(t);
  const Jac<C, D> mjp = flow_pos_dmom_rows<C, D>(A, rot, jac_prev);
  CMat<C> inv;
  double ld;
  if (!gram_inverse<C>(A, rows_inner<C, D>(jac_prev, mjp, abs_t), &inv, &ld)) return MM_ST_LINALG;
  Vec<D> mu;
#pragma unroll kUnrollD<D>
  for (int i = 0; i < D; ++i) mu.v[i] = 0.0;
  for (int it = 0; it < o.max_iters; ++it) {
    *n_iters += 1;
    const CVec<C> c = constr_value<C, D>(A, q);
    const double err = cnorm<C>(c, o.norm);
    const CVec<C> x = cmat_vec<C>(inv, c);
    const Vec<D> dmu = rows_combine<C, D>(jac_prev, x);
    Vec<D> dpos = rows_combine<C, D>(mjp, x);
#pragma unroll kUnrollD<D>
    for (int i = 0; i < D; ++i) dpos.v[i] = abs_t * dpos.v[i];
    if (err > o.div_tol || err != err) return MM_ST_DIVERGED;
    if (err < o.constr_tol && vnorm<D>(dpos, o.norm) < o.pos_tol) {
      apply_mu<D>(A, rot, p, mu, t);
      *jac_out = constr_jacob<C, D>(A, q);
      return MM_ST_OK;
    }
#pragma unroll kUnrollD<D>
    for (int i = 0; i < D; ++i) {
      mu.v[i] += dmu.v[i];
      q.v[i] -= dpos.v[i];
    }
  }
  return MM_ST_MAX_ITERS;
}

// solve_projection_onto_manifold_newton_with_line_search (solvers.py:561-614)
template <int C, int D>
__device__ __forceinline__ int line_search_project(const ConArgs& A, const Rot<D>& rot, Vec<D>& q, Vec<D>& p,
                                                   const Jac<C, D>& jac_prev, double t, Jac<C, D>* jac_out,
                                                   long long* n_iters) {
  const mm_proj_opts& o = A.opts;
  const double abs_t = A.gaussian ? 1.0 : fabs(t);
  const Jac<C, D> mjp = flow_pos_dmom_rows<C, D>(A, rot, jac_prev);
  Vec<D> mu, dpos;
#pragma unroll kUnrollD<D>
  for (int i = 0; i < D; ++i) { mu.v[i] = 0.0; dpos.v[i] = 0.0; }
  double step = 0.0;
  for (int it = 0; it < o.max_iters; ++it) {
    *n_iters += 1;
    const Jac<C, D> jac = constr_jacob<C, D>(A, q);
    const CVec<C> c = constr_value<C, D>(A, q);
    const double err = cnorm<C>(c, o.norm);
    if (it > 0 && (err > o.div_tol || err != err)) return MM_ST_DIVERGED;
    bool small_step = (it == 0);
    if (!small_step) {
      Vec<D> sd;
#pragma unroll kUnrollD<D>
      for (int i = 0; i < D; ++i) sd.v[i] = step * dpos.v[i];
      small_step = vnorm<D>(sd, o.norm) < o.pos_tol;
    }
    if (err < o.constr_tol && small_step) {
      apply_mu<D>(A, rot, p, mu, t);
      *jac_out = jac;
      return MM_ST_OK;
    }
    const CMat<C> a = rows_inner<C, D>(jac, mjp, abs_t);
    if (!all_finite<C>(a)) return MM_ST_SOLVER_LINALG;
    const CVec<C> x = lu_solve<C>(a, c);
    const Vec<D> dmu = rows_combine<C, D>(jac_prev, x);
    const Vec<D> raw = rows_combine<C, D>(mjp, x);
#pragma unroll kUnrollD<D>
    for (int i = 0; i < D; ++i) dpos.v[i] = -(abs_t * raw.v[i]);
    const Vec<D> q_curr = q;
    step = 1.0;
    for (int ls = 0; ls < o.max_line_search_iters; ++ls) {
#pragma unroll kUnrollD<D>
      for (int i = 0; i < D; ++i) q.v[i] = q_curr.v[i] + step * dpos.v[i];
      const double new_err = cnorm<C>(constr_value<C, D>(A, q), o.norm);
      if (new_err < err) break;
      step *= 0.5;
    }
#pragma unroll kUnrollD<D>
    for (int i = 0; i < D; ++i) mu.v[i] += step * dmu.v[i];
  }
  return MM_ST_MAX_ITERS;
}

template <int C, int D>
__device__ __forceinline__ int project(const ConArgs& A, const Rot<D>& rot, Vec<D>& q, Vec<D>& p,
                                       const Jac<C, D>& jac_prev, double t, Jac<C, D>* jac_out,
                                       long long* n_iters) {
  if (A.opts.solver == MM_PROJ_QUASI_NEWTON)
    return quasi_newton_project<C, D>(A, rot, q, p, jac_prev, t, jac_out, n_iters);
  if (A.opts.solver == MM_PROJ_NEWTON_LINE_SEARCH)
    return line_search_project<C, D>(A, rot, q, p, jac_prev, t, jac_out, n_iters);
  return newton_project<C, D>(A, rot, q, p, jac_prev, t, jac_out, n_iters);
}

// A kernel instantiation may pin run-time selectors of the model / solver to constants (-1 = left to run time): the
// fields of the kernel's private ConArgs copy are overwritten before everything is inlined, so every `switch` on them
// folds away.  The general core tests ~20 wave-uniform selectors per step (target, metric kind, constraint, solver,
// norms ...); each is a scalar compare + branch in a single wave's dependent instruction stream.
struct SpecNone {
  static constexpr int target = -1, metric_kind = -1, constr = -1, solver = -1, norm = -1, n_inner = -1;
  static constexpr bool plain_steps = false;  // true: no per-chain step sizes / step counts
};
// BASELINE config c5 / the reference's README example: torus density on the torus, identity metric, Newton
// projection, maximum norm, one inner step.
struct SpecTorus {
  static constexpr int target = MM_TARGET_TORUS, metric_kind = MM_METRIC_IDENTITY, constr = MM_CONSTR_TORUS,
                       solver = MM_PROJ_NEWTON, norm = MM_NORM_LINF, n_inner = 1;
  static constexpr bool plain_steps = false;
};
template <class SPEC>
__device__ __forceinline__ void apply_spec(ConArgs& A) {
  if constexpr (SPEC::target >= 0) A.target = SPEC::target;
  if constexpr (SPEC::metric_kind >= 0) A.metric_kind = SPEC::metric_kind;
  if constexpr (SPEC::constr >= 0) A.constr = SPEC::constr;
  if constexpr (SPEC::solver >= 0) A.opts.solver = SPEC::solver;
  if constexpr (SPEC::norm >= 0) {
    A.opts.norm = SPEC::norm;
    A.opts.rev_norm = SPEC::norm;
  }
  if constexpr (SPEC::n_inner >= 0) A.opts.n_inner = SPEC::n_inner;
}
template <class SPEC>
inline bool spec_matches(const ConArgs& a) {
  return (SPEC::target < 0 || a.target == SPEC::target) && (SPEC::metric_kind < 0 || a.metric_kind == SPEC::metric_kind) &&
         (SPEC::constr < 0 || a.constr == SPEC::constr) && (SPEC::solver < 0 || a.opts.solver == SPEC::solver) &&
         (SPEC::norm < 0 || (a.opts.norm == SPEC::norm && a.opts.rev_norm == SPEC::norm)) &&
         (SPEC::n_inner < 0 || a.opts.n_inner == SPEC::n_inner);
}

// EXT = false is the plain dens_wrt_hausdorff=True Euclidean system (BASELINE config c5): the Gram-term and
// Gaussian-split branches are compiled out (the flags are forced to zero before everything is inlined).
// PAD = false: the system's dimension is the template capacity D (A.dim is forced to D so every stride and guard
// folds away); PAD = true: A.dim <= D at run time, the extra coordinates are held at zero.
template <int C, int D, bool EXT, bool PAD, class SPEC = SpecNone>
__device__ __forceinline__ void constrained_leapfrog_body(ConArgs& A) {
  const int64_t chain = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (chain >= A.n_chains) return;
  apply_spec<SPEC>(A);
  if constexpr (!PAD) A.dim = D;
  if constexpr (!EXT) {
    A.ambient = 0;
    A.gaussian = 0;
  }
  Vec<D> q, p;
#pragma unroll kUnrollD<D>
  for (int i = 0; i < D; ++i) {
    q.v[i] = (i < A.dim) ? A.pos[chain * A.dim + i] : 0.0;
    p.v[i] = (i < A.dim) ? A.mom[chain * A.dim + i] : 0.0;
  }
  const double t = mmdev::signed_step(A.dir, A.step_scale, chain, A.step_size);
  const int n_inner = A.opts.n_inner;
  const double t_in = t / n_inner;
  long long n_newton = 0, n_grad = 0;
  int status = MM_ST_OK, done = 0;

  Rot<D> rot{};
  if (A.gaussian) rot = make_rot<D>(A, fabs(t_in));
  Vec<D> g;  // cached dh1_dpos at the current position
  if (!dh1_dpos<C, D>(A, q, &g)) status = MM_ST_LINALG;
  Jac<C, D> jac = constr_jacob<C, D>(A, q);
  ++n_grad;
  const int my_steps = mmdev::chain_steps(A.chain_steps, chain, A.n_steps);
  for (int s = 0; s < my_steps && status == MM_ST_OK; ++s) {
    Vec<D> qs = q, ps = p;
    Jac<C, D> js = jac;
    // ---- A(t/2): h1_flow then cotangent projection                    integrators.py:947-949
#pragma unroll kUnrollD<D>
    for (int i = 0; i < D; ++i) ps.v[i] -= (0.5 * t) * g.v[i];
    if (!project_cotangent<C, D>(A, ps, js)) { status = MM_ST_LINALG; break; }
    // ---- B(t): n_inner retractions + reversibility checks              integrators.py:951-979
    Vec<D> gs = g;
    for (int in = 0; in < n_inner && status == MM_ST_OK; ++in) {
      const Vec<D> q_prev = qs;
      const Jac<C, D> j_prev = js;
      h2_flow<D>(A, rot, qs, ps, t_in, t_in < 0.0 ? -1.0 : 1.0);
      Jac<C, D> j_new;
      status = project<C, D>(A, rot, qs, ps, j_prev, t_in, &j_new, &n_newton);
      if (status != MM_ST_OK) break;
      if (in == n_inner - 1) {  // pre-evaluated dh1_dpos, integrators.py:956-969
        if (!dh1_dpos<C, D>(A, qs, &gs)) { status = MM_ST_LINALG; break; }
        ++n_grad;
      }
      if (!project_cotangent<C, D>(A, ps, j_new)) { status = MM_ST_LINALG; break; }
      // reversibility check on a copy                                    integrators.py:971-979
      Vec<D> qb = qs, pb = ps;
      Jac<C, D> j_tmp;
      h2_flow<D>(A, rot, qb, pb, -t_in, t_in < 0.0 ? 1.0 : -1.0);
      status = project<C, D>(A, rot, qb, pb, j_new, -t_in, &j_tmp, &n_newton);
      if (status != MM_ST_OK) break;
      Vec<D> diff;
#pragma unroll kUnrollD<D>
      for (int i = 0; i < D; ++i) diff.v[i] = qb.v[i] - q_prev.v[i];
      if (vnorm<D>(diff, A.opts.rev_norm) > A.opts.rev_tol) { status = MM_ST_NON_REVERSIBLE; break; }
      js = j_new;
    }
    if (status != MM_ST_OK) break;
    // ---- A(t/2)
#pragma unroll kUnrollD<D>
    for (int i = 0; i < D; ++i) ps.v[i] -= (0.5 * t) * gs.v[i];
    if (!project_cotangent<C, D>(A, ps, js)) { status = MM_ST_LINALG; break; }
    q = qs; p = ps; jac = js; g = gs;
    ++done;
  }
#pragma unroll kUnrollD<D>
  for (int i = 0; i < D; ++i)
    if (i < A.dim) {
      A.pos[chain * A.dim + i] = q.v[i];
      A.mom[chain * A.dim + i] = p.v[i];
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

template <int C, int D, bool PAD>
__device__ __forceinline__ void project_momentum_body(ConArgs& A) {
  const int64_t chain = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (chain >= A.n_chains) return;
  if constexpr (!PAD) A.dim = D;
  Vec<D> q, p;
#pragma unroll kUnrollD<D>
  for (int i = 0; i < D; ++i) {
    q.v[i] = (i < A.dim) ? A.pos[chain * A.dim + i] : 0.0;
    p.v[i] = (i < A.dim) ? A.mom[chain * A.dim + i] : 0.0;
  }
  const Jac<C, D> jac = constr_jacob<C, D>(A, q);
  const bool ok = project_cotangent<C, D>(A, p, jac);
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
#pragma unroll kUnrollD<D>
  for (int i = 0; i < D; ++i)
    if (i < A.dim) A.mom[chain * A.dim + i] = ok ? p.v[i] : nan;
}

// h1's Gram term for dens_wrt_hausdorff=False: out[chain] += log_det_sqrt_gram = log|det gram| / 2
// (systems.py:829-831, 853-856); NaN where the reference raises LinAlgError.
template <int C, int D, bool PAD>
__device__ __forceinline__ void add_log_det_sqrt_gram_body(ConArgs& A, double* __restrict__ out) {
  const int64_t chain = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (chain >= A.n_chains) return;
  if constexpr (!PAD) A.dim = D;
  Vec<D> q;
#pragma unroll kUnrollD<D>
  for (int i = 0; i < D; ++i) q.v[i] = (i < A.dim) ? A.pos[chain * A.dim + i] : 0.0;
  const Jac<C, D> jac = constr_jacob<C, D>(A, q);
  const CMat<C> gram = rows_inner<C, D>(jac, minv_rows<C, D>(A, jac), 1.0);
  CMat<C> inv;
  double ld;
  double v = __longlong_as_double(0x7ff8000000000000LL);
  if (gram_inverse<C>(A, gram, &inv, &ld)) v = 0.5 * ld;
  out[chain] += v;
}

#ifndef MM_RTC_BUILD  // ---- host side and the in-tree kernel instantiations --------------------------------------
template <int C, int D, bool EXT, bool PAD, class SPEC = SpecNone>
__global__ __launch_bounds__(256) void constrained_leapfrog_kernel(ConArgs A) {
  constrained_leapfrog_body<C, D, EXT, PAD, SPEC>(A);
}
template <int C, int D, bool PAD>
__global__ __launch_bounds__(256) void project_momentum_kernel(ConArgs A) {
  project_momentum_body<C, D, PAD>(A);
}
template <int C, int D, bool PAD>
__global__ __launch_bounds__(256) void add_log_det_sqrt_gram_kernel(ConArgs A, double* __restrict__ out) {
  add_log_det_sqrt_gram_body<C, D, PAD>(A, out);
}

inline ConArgs make_args(const mm_model* m, mm_state* s) {
  ConArgs a{};
  a.pos = s->d_pos;
  a.mom = s->d_mom;
  a.dir = s->d_dir;
  a.step_scale = s->d_step_scale;
  a.chain_steps = s->d_chain_steps;
  a.status = s->d_status;
  a.n_done = s->d_n_done;
  a.n_chains = s->n;
  a.dim = m->dim;
  a.target = m->target;
  a.tparams = m->d_target_params;
  a.metric_kind = m->metric_kind;
  a.minv = m->d_metric_inv;
  a.constr = m->constr;
  a.cp0 = m->h_constr_params[0];
  a.cp1 = m->h_constr_params[1];
  a.cparams = m->d_constr_params;
  a.ambient = m->dens_wrt_ambient;
  a.gaussian = m->gaussian_split;
  a.omega = m->d_metric_omega;
  a.eigvec = m->d_metric_eigvec;
  return a;
}

enum { K_STEP = 0, K_PROJECT = 1, K_LOGDET = 2 };

template <int C, int D, bool PAD>
int launch_cd(mm_ctx* ctx, const ConArgs& a, int which, double* h_out) {
  const unsigned blocks = (unsigned)((a.n_chains + 255) / 256);
  if (which == K_LOGDET)
    hipLaunchKernelGGL((add_log_det_sqrt_gram_kernel<C, D, PAD>), dim3(blocks), dim3(256), 0, ctx->stream, a, h_out);
  else if (which == K_PROJECT)
    hipLaunchKernelGGL((project_momentum_kernel<C, D, PAD>), dim3(blocks), dim3(256), 0, ctx->stream, a);
  else if (a.ambient || a.gaussian)
    hipLaunchKernelGGL((constrained_leapfrog_kernel<C, D, true, PAD>), dim3(blocks), dim3(256), 0, ctx->stream, a);
  else if (C == 1 && D == 3 && !PAD && spec_matches<SpecTorus>(a)) {
    if constexpr (C == 1 && D == 3 && !PAD)
      hipLaunchKernelGGL((constrained_leapfrog_kernel<1, 3, false, false, SpecTorus>), dim3(blocks), dim3(256), 0,
                         ctx->stream, a);
  } else
    hipLaunchKernelGGL((constrained_leapfrog_kernel<C, D, false, PAD>), dim3(blocks), dim3(256), 0, ctx->stream, a);
  MM_HIP_CHECK(ctx, hipGetLastError());
  return MM_OK;
}

#endif  // !MM_RTC_BUILD

}  // namespace mmcon
