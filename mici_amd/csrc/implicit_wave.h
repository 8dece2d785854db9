// Device code of the wave-per-chain dense-Riemannian kernels (k_implicit.hip instantiates it for the built-in metrics;
// mm_rtc.hip compiles it at run time around a USER metric: metric_func / vjp_metric_func of
// DenseRiemannianMetricSystem, systems.py:1322-1358).  Layout and algorithms: see the head of k_implicit.hip.
#pragma once
#include "implicit_core.h"

#include "user_metric.h"

namespace mmwave {

using namespace mmdev;
using namespace mmimp;

#ifndef MM_WAVES_PER_BLOCK
#define MM_WAVES_PER_BLOCK 4
#endif
constexpr int kWaves = MM_WAVES_PER_BLOCK;  // chains per workgroup (a run-time translation unit may lower it to fit 64 KB of LDS)
constexpr int kWaveLdsDoubles = 5 * 64 + mmimp::SL_COUNT_REFINE * 64 + 8 * 64;  // scratch vectors + step state + column block

// Per-wave LDS scratch (doubles): 5 vectors of 64.
struct WaveLds {
  double* col;  // published sweep column (permuted order)
  double* vin;  // mat-vec input  (permuted order)
  double* vout; // mat-vec output (permuted order)
  double* nat;  // natural-order vector for target derivatives
  double* aux;  // natural-order spare (z for sample_momentum)
  double* mat;  // user metrics with the dense-accessor VJP only: a DP x DP matrix (the inverse as the user's VJP reads it)
  // user metrics (user_metric.h): the point of the held inverse / of the refinement products in natural order, their aux
  double* uq;
  double* ux;
  double* uaq;
  double* uax;
};

template <int TS>
struct Geo {
  static constexpr int DP = 8 * TS;                    // padded dimension
  static constexpr int TSTRIDE = TS * TS + 2;          // per-lane stride of the shared base matrix
  // position of flat element i in a permuted LDS vector: group (i & 7) holds TS consecutive values
  __device__ static __forceinline__ int pos(int i) { return (i & 7) * TS + (i >> 3); }
};

// per-wave LDS: 5 vectors of 64, the step's slots, the column block of the back substitution; a user metric adds a
// dense DP x DP matrix
template <int TS, int RMETRIC>
__host__ __device__ constexpr int wave_mat_doubles() {
  return (RMETRIC == MM_RMETRIC_USER && !mmuser::kFlatVjp) ? Geo<TS>::DP * Geo<TS>::DP : 0;
}
template <int TS, int RMETRIC>
__host__ __device__ constexpr int wave_lds_doubles() {
  return kWaveLdsDoubles + wave_mat_doubles<TS, RMETRIC>() + (RMETRIC == MM_RMETRIC_USER ? mmuser::lds_doubles(64) : 0);
}
template <int TS, int RMETRIC>
__device__ __forceinline__ WaveLds make_wave_lds(double* wl) {
  WaveLds w{wl, wl + 64, wl + 128, wl + 192, wl + 256, nullptr, nullptr, nullptr, nullptr, nullptr};
  if constexpr (RMETRIC == MM_RMETRIC_USER) {
    double* p = wl + kWaveLdsDoubles;
    w.mat = wave_mat_doubles<TS, RMETRIC>() ? p : nullptr;
    p += wave_mat_doubles<TS, RMETRIC>();
    constexpr int kA = (mmuser::kAux + 1) & ~1;
    w.uq = p;
    w.ux = p + 64;
    w.uaq = p + 128;
    w.uax = p + 128 + kA;
  }
  return w;
}

// ---- symmetric sweep: T <- M^-1, returns false if a pivot is not > 0 (== Cholesky would fail) ------
template <int TS, bool LOGDET, bool CHOLVEC>
__device__ __forceinline__ bool sweep_inverse(double (&T)[TS][TS], int dim, int lane,
                                              const WaveLds& w, double* logdet, double* chol_y) {
  const int ti = lane >> 3, tj = lane & 7;
  bool ok = true;
  double ld = 0.0, y = 0.0;
#pragma unroll
  for (int kb = 0; kb < TS; ++kb) {
#pragma unroll 1
    for (int kt = 0; kt < 8; ++kt) {
      const int k = kb * 8 + kt;  // padded columns (k >= dim) are identity: sweeping them is a no-op
      // owners of column k publish it (rows ti + 8a live at [a][kb] of lanes with tj == kt)
      if (tj == kt) {
#pragma unroll
        for (int a = 0; a < TS; ++a) w.col[ti * TS + a] = T[a][kb];
      }
      wave_sync();
      const double piv = w.col[kt * TS + kb];
      ok = ok && (piv > 0.0);
      const double d = fast_rcp(piv);
      if constexpr (LOGDET) ld += log(piv);
      if constexpr (CHOLVEC) {
        // y += L[:, k] z_k with L[i, k] = a_ik / sqrt(a_kk) for i >= k (Cholesky column from the sweep)
        const double rs = 1.0 / sqrt(piv);
        const double zk = w.aux[k];
        if (lane >= k && lane < dim) y += (w.col[Geo<TS>::pos(lane)] * rs) * zk;
      }
      double ar[TS], ac[TS];
#pragma unroll
      for (int a = 0; a < TS; ++a) {
        ar[a] = w.col[ti * TS + a];
        ac[a] = w.col[tj * TS + a];
      }
      // special entries that make one uniform FMA do the whole sweep step:
      //   col factor  c_k   = a_kk - 1,   row multiplier m_k = 1 - d,   m_i = a_i d otherwise
#pragma unroll
      for (int a = 0; a < TS; ++a) ar[a] *= d;
      if (ti == kt) ar[kb] = 1.0 - d;
      if (tj == kt) ac[kb] = piv - 1.0;
#pragma unroll
      for (int a = 0; a < TS; ++a)
#pragma unroll
        for (int b = 0; b < TS; ++b) T[a][b] = __builtin_fma(-ar[a], ac[b], T[a][b]);
      if (ti == kt && tj == kt) T[kb][kb] -= 2.0;
      wave_sync();
    }
  }
  // T now holds -M^-1 on the leading dim x dim block (padding rows/cols untouched)
#pragma unroll
  for (int a = 0; a < TS; ++a)
#pragma unroll
    for (int b = 0; b < TS; ++b) T[a][b] = -T[a][b];
  if constexpr (LOGDET) *logdet = ld;
  if constexpr (CHOLVEC) *chol_y = y;
  // every lane saw the same pivots, so `ok` is wave-uniform
  return ok;
}

// ---- solve M u = rhs WITHOUT forming the inverse: LDL^T elimination + substitutions -------------------
// The position-space fixed-point iterations (C reversibility check, C adjoint: ~9 of the ~10 metric
// constructions per step) use each metric for exactly one solve.  Symmetric elimination needs only the
// shrinking trailing window: in block kb a lane updates its (TS-kb)^2 entries with a, b >= kb, i.e.
// sum_kb 8 (TS-kb)^2 = 1632 FMAs at D = 64 instead of the 4096 of a full sweep.
//   forward substitution rides along (the published column IS the column of L, read flat);
//   the rows of U = D L^T stay frozen in the tiles (row multipliers of eliminated rows are zeroed) and
//   are re-published 8 columns at a time for a column-oriented back substitution.
// T is destroyed.  Returns false if a pivot is not > 0.
template <int TS>
__device__ __forceinline__ bool eliminate_solve(double (&T)[TS][TS], double rhs, int lane,
                                                const WaveLds& w, double* blk, double* u_out) {
  const int ti = lane >> 3, tj = lane & 7;
  bool ok = true;
  double y = rhs;    // flat: element `lane`
  double invd = 1.0; // 1 / pivot of row `lane`
#pragma unroll
  for (int kb = 0; kb < TS; ++kb) {
#pragma unroll 1
    for (int kt = 0; kt < 8; ++kt) {
      const int k = kb * 8 + kt;
      if (tj == kt) {
#pragma unroll
        for (int a = kb; a < TS; ++a) w.col[ti * TS + a] = T[a][kb];
      }
      wave_sync();
      const double piv = w.col[kt * TS + kb];
      ok = ok & (piv > 0.0);
      const double d = fast_rcp(piv);
      // forward substitution with column k of L (flat): y_i -= (a_ik / piv) y_k for i > k
      const double yk = wave_bcast(y, k);
      const double ci = (lane < Geo<TS>::DP) ? w.col[Geo<TS>::pos(lane)] : 0.0;
      if (lane > k) y = __builtin_fma(-(ci * d), yk, y);
      if (lane == k) invd = d;
      // rank-1 update of the trailing window; rows <= k of the current block row are frozen (U rows)
      double ar[TS], ac[TS];
#pragma unroll
      for (int a = kb; a < TS; ++a) {
        ar[a] = w.col[ti * TS + a] * d;
        ac[a] = w.col[tj * TS + a];
      }
      if (ti <= kt) ar[kb] = 0.0;
#pragma unroll
      for (int a = kb; a < TS; ++a)
#pragma unroll
        for (int b = kb; b < TS; ++b) T[a][b] = __builtin_fma(-ar[a], ac[b], T[a][b]);
      wave_sync();
    }
  }
  // back substitution: z = D^-1 y, then for k = DP-1 .. 0: u_k = z_k, z_i -= (U_ik / d_i) u_k for i < k
  double z = y * invd;
#pragma unroll
  for (int kb = TS - 1; kb >= 0; --kb) {
    // publish the 8 columns of block kb (frozen rows above the diagonal): blk[kt][pos(i)] = entry (i, kb*8+kt)
#pragma unroll
    for (int a = 0; a <= kb; ++a) blk[tj * 64 + ti * TS + a] = T[a][kb];
    wave_sync();
#pragma unroll 1
    for (int kt = 7; kt >= 0; --kt) {
      const int k = kb * 8 + kt;
      const double uk = wave_bcast(z, k);
      const double uik = (lane < k) ? blk[kt * 64 + Geo<TS>::pos(lane)] : 0.0;
      z = __builtin_fma(-(uik * invd), uk, z);
    }
    wave_sync();
  }
  *u_out = z;
  return ok;
}

// ---- y = T x with x, y "flat" (element i on lane i) ----------------------------------------------------
template <int TS>
__device__ __forceinline__ double matvec_flat(const double (&T)[TS][TS], double x, int lane,
                                              const WaveLds& w) {
  const int ti = lane >> 3, tj = lane & 7;
  if (lane < Geo<TS>::DP) w.vin[Geo<TS>::pos(lane)] = x;
  wave_sync();
  double xc[TS], part[TS];
#pragma unroll
  for (int b = 0; b < TS; ++b) xc[b] = w.vin[tj * TS + b];
#pragma unroll
  for (int a = 0; a < TS; ++a) {
    double s = 0.0;
#pragma unroll
    for (int b = 0; b < TS; ++b) s = __builtin_fma(T[a][b], xc[b], s);
    part[a] = s;
  }
#pragma unroll
  for (int a = 0; a < TS; ++a) part[a] = group8_sum(part[a]);  // over the 8 lanes of a row group (DPP)
  if (tj == 0) {
#pragma unroll
    for (int a = 0; a < TS; ++a) w.vout[ti * TS + a] = part[a];
  }
  wave_sync();
  const double y = (lane < Geo<TS>::DP) ? w.vout[Geo<TS>::pos(lane)] : 0.0;
  wave_sync();
  return y;
}

// diagonal of T in flat form
template <int TS>
__device__ __forceinline__ double diag_flat(const double (&T)[TS][TS], int lane, const WaveLds& w) {
  const int ti = lane >> 3, tj = lane & 7;
  if (ti == tj) {
#pragma unroll
    for (int a = 0; a < TS; ++a) w.vout[ti * TS + a] = T[a][a];
  }
  wave_sync();
  const double y = (lane < Geo<TS>::DP) ? w.vout[Geo<TS>::pos(lane)] : 0.0;
  wave_sync();
  return y;
}

// ---- metric_func(q) into the register tiles (padding: identity) ---------------------------------------
// returns false if an entry is not finite ("Array is not finite.", matrices.py:211-215)
template <int TS, int RMETRIC>
__device__ __forceinline__ bool build_metric(double (&T)[TS][TS], double q, int dim, int lane,
                                             const WaveLds& w, const double* base_lds) {
  const int ti = lane >> 3, tj = lane & 7;
  if (lane < Geo<TS>::DP) w.vin[Geo<TS>::pos(lane)] = q;
  if constexpr (RMETRIC == MM_RMETRIC_USER) {  // the point in natural order for the user's hooks, then its aux block
    w.uq[lane] = (lane < dim) ? q : 0.0;
    wave_sync();
    mmuser::prepare(mmuser::WaveTeam{lane}, w.uq, dim, base_lds, w.uaq);
  }
  wave_sync();
  double qr[TS], qc[TS];
#pragma unroll
  for (int a = 0; a < TS; ++a) {
    qr[a] = w.vin[ti * TS + a];
    qc[a] = w.vin[tj * TS + a];
  }
  // Padded rows / columns: q is 0 there and the staged base matrix is 0, so the closed form gives 0
  // off the diagonal; the diagonal of the padding is set to 1 afterwards (identity block).
  bool finite = true;
  const double inv_d = 1.0 / (double)dim;
#pragma unroll
  for (int a = 0; a < TS; ++a)
#pragma unroll
    for (int b = 0; b < TS; ++b) {
      double v;
      if constexpr (RMETRIC == MM_RMETRIC_RANK1) {
        v = base_lds[lane * Geo<TS>::TSTRIDE + a * TS + b] + (qr[a] * qc[b]) * inv_d;
      } else if constexpr (RMETRIC == MM_RMETRIC_USER) {
        // the user's metric_func, entry by entry (w.uq holds q in natural order; base_lds is the params pointer)
        const int i = ti + 8 * a, j = tj + 8 * b;
        v = mmuser::entry_padded(w.uq, i, j, dim, base_lds, w.uaq);
      } else {  // MM_RMETRIC_DIAGQUAD: only diagonal lanes / diagonal tile entries are non-zero
        v = 0.0;
      }
      T[a][b] = v;
    }
  if (ti == tj) {
#pragma unroll
    for (int a = 0; a < TS; ++a) {
      if constexpr (RMETRIC == MM_RMETRIC_DIAGQUAD) T[a][a] = __builtin_fma(qr[a], qr[a], 1.0);
      if (ti + 8 * a >= dim) T[a][a] = 1.0;
    }
  }
  // "Array is not finite." (matrices.py:211-215): x * 0 is NaN exactly for inf / NaN entries
  double chk = 0.0;
#pragma unroll
  for (int a = 0; a < TS; ++a)
#pragma unroll
    for (int b = 0; b < TS; ++b) chk = __builtin_fma(T[a][b], 0.0, chk);
  finite = (chk == 0.0);
  wave_sync();
  return __all(finite);
}

// 0.5 * vjp_metric(V) for the symmetric explicit matrix V held in tiles (general-VJP path):
//   rank-one metric: (V + V^T) q / (2D) = V q / D ;  diag-quad metric: q_i V_ii
template <int TS, int RMETRIC>
__device__ __forceinline__ double half_vjp_tiles(const double (&V)[TS][TS], double q, int dim,
                                                 int lane, const WaveLds& w, const double* uparams = nullptr) {
  if constexpr (RMETRIC == MM_RMETRIC_RANK1) {
    return matvec_flat<TS>(V, q, lane, w) * (1.0 / (double)dim);
  } else {
    return q * diag_flat<TS>(V, lane, w);
  }
}

// 0.5 * vjp_metric(-u u^T)  (dh2_dpos, systems.py:1392-1396 with matrices.py:1179-1181)
template <int RMETRIC>
__device__ __forceinline__ double half_vjp_neg_outer(double u, double q, int dim, int lane, const WaveLds& w,
                                                     const double* uparams = nullptr) {
  if constexpr (RMETRIC == MM_RMETRIC_RANK1) {
    const double uq = wave_sum(lane < dim ? u * q : 0.0);
    return -(u * uq) * (1.0 / (double)dim);
  } else {
    return -q * (u * u);
  }
}

__device__ __forceinline__ double flat_norm(double x, int dim, int lane, int kind) {
  const double acc = wave_norm_accum(0.0, lane < dim ? x : 0.0, kind);
  return wave_norm_finish(acc, kind);
}

template <int TS>
__device__ __forceinline__ double grad_flat(int target, double q, int dim, int lane,
                                            const WaveLds& w, const double* tparams) {
  if (lane < 64) w.nat[lane] = (lane < dim) ? q : 0.0;
  wave_sync();
  const TargetAux aux = target_prepare<false>(target, w.nat, dim, tparams, lane);
  const double g = (lane < dim) ? target_grad_elem<false>(target, aux, w.nat, lane, dim, tparams) : 0.0;
  wave_sync();
  return g;
}

template <int TS, int RMETRIC>
__device__ __forceinline__ void stage_base(double* base_lds, const double* rparams, int dim) {
  if constexpr (RMETRIC == MM_RMETRIC_RANK1) {
    // base_lds[lane][a][b] = B[ti + 8a][tj + 8b], lane stride padded by 16 B against bank conflicts
    for (int idx = threadIdx.x; idx < 64 * TS * TS; idx += blockDim.x) {
      const int l = idx / (TS * TS), r = idx - l * (TS * TS), a = r / TS, b = r - a * TS;
      const int i = (l >> 3) + 8 * a, j = (l & 7) + 8 * b;
      base_lds[l * Geo<TS>::TSTRIDE + r] = (i < dim && j < dim) ? rparams[(int64_t)i * dim + j] : 0.0;
    }
  }
  __syncthreads();
}

// Backend of implicit_core.h for one wave per chain.
template <int TS, int RMETRIC>
struct WaveBackend {
  static constexpr bool kSolveByInverse = false;  // implicit_core.h: solve = invert + mat-vec, one construction site
  static constexpr bool kUnifiedConstruct = false;
  static constexpr bool kCountersInLds = false;
  // implicit_core.h: solve-only constructions refined from the held inverse.  Built-in metrics: M(x) v has a closed form
  // that needs no tiles; a user metric evaluates its TS x TS entries per product (cheap with MM_USER_AUX, user_metric.h)
  static constexpr bool kRefine = true;
  bool refine_on;
  double T[TS][TS];
  int dim, lane, target;
  WaveLds w;
  double* stash;  // [SL_COUNT_REFINE][64] flat state of the step, in LDS to keep VGPRs for the tiles
  double* blk;    // [8][64] one block of re-published columns for the back substitution

  __device__ __forceinline__ double& slot(int i) { return stash[i * 64 + lane]; }
  const double* base_lds;
  const double* tparams;

  __device__ __forceinline__ bool build_and_invert(double x) {
    bool ok = build_metric<TS, RMETRIC>(T, x, dim, lane, w, base_lds);
    ok = sweep_inverse<TS, false, false>(T, dim, lane, w, nullptr, nullptr) && ok;
    return ok;
  }
  // metric at x used for ONE solve u = M(x)^-1 rhs (position-space fixed-point iterations)
  __device__ __forceinline__ bool build_and_solve(double x, double rhs, double* u) {
    bool ok = build_metric<TS, RMETRIC>(T, x, dim, lane, w, base_lds);
    ok = eliminate_solve<TS>(T, rhs, lane, w, blk, u) && ok;
    if (lane >= dim) *u = 0.0;
    return ok;
  }
  __device__ __forceinline__ double matvec(double v) { return matvec_flat<TS>(T, v, lane, w); }
  // ---- implicit_core.h lowrank_solve / lowrank_update (round 6): the built-in rank-one-update metric's
  // solve-only constructions by the Woodbury identity from the held inverse, the inverse carried from step to step by the
  // symmetric rank-two update; decided at run time (MICI_AMD_LOWRANK=0: the CG refinement) -------------------------------
  static constexpr bool kLowRankBuiltin = RMETRIC == MM_RMETRIC_RANK1;
  static constexpr bool kLowRank = kLowRankBuiltin;
  bool lowrank_on_;
  int lr_refresh_;
  __device__ __forceinline__ bool lowrank_on() const { return lowrank_on_; }
  __device__ __forceinline__ int lowrank_refresh() const { return lr_refresh_; }
  __device__ __forceinline__ double lowrank_scale() const { return (double)dim; }
  __device__ __forceinline__ double lowrank_vec(double x) const { return x; }
  __device__ __forceinline__ double& lowrank_u0() { return slot(SL_Q); }
  __device__ __forceinline__ void sum3(double a, double b, double c, double* sa, double* sb, double* sc) {
    *sa = wave_sum(lane < dim ? a : 0.0);
    *sb = wave_sum(lane < dim ? b : 0.0);
    *sc = wave_sum(lane < dim ? c : 0.0);
  }
  // T += a u^T + b v^T with u = al a + be b, v = be a + ga b: lane (ti, tj) holds the entries (ti + 8 a', tj + 8 b') - its row
  // operands are group ti of the permuted vectors a, b, its column operands group tj of u, v
  __device__ __forceinline__ void inverse_update(double al, double be, double ga, double a, double b) {
    const int ti = lane >> 3, tj = lane & 7;
    if (lane < Geo<TS>::DP) {
      const bool act = lane < dim;
      const double am = act ? a : 0.0, bm = act ? b : 0.0;
      const int pp = Geo<TS>::pos(lane);
      w.col[pp] = am;
      w.aux[pp] = bm;
      w.vin[pp] = __builtin_fma(al, am, be * bm);
      w.vout[pp] = __builtin_fma(be, am, ga * bm);
    }
    wave_sync();
    double ar[TS], br[TS], uc[TS], vc[TS];
#pragma unroll
    for (int k = 0; k < TS; ++k) {
      ar[k] = w.col[ti * TS + k];
      br[k] = w.aux[ti * TS + k];
      uc[k] = w.vin[tj * TS + k];
      vc[k] = w.vout[tj * TS + k];
    }
#pragma unroll
    for (int x = 0; x < TS; ++x)
#pragma unroll
      for (int y = 0; y < TS; ++y) T[x][y] = __builtin_fma(ar[x], uc[y], __builtin_fma(br[x], vc[y], T[x][y]));
    wave_sync();
  }
  // ---- refinement solves (implicit_core.h refine_solve): M(x) v in the form that suits the metric ----------------------
  double rs_[RS_COUNT], xpt_;
  __device__ __forceinline__ double& rslot(int i) { return rs_[i]; }
  __device__ __forceinline__ void sum2(double a, double b, double* sa, double* sb) {
    *sa = wave_sum(lane < dim ? a : 0.0);
    *sb = wave_sum(lane < dim ? b : 0.0);
  }
  __device__ __forceinline__ double sum1(double a) { return wave_sum(lane < dim ? a : 0.0); }
  __device__ __forceinline__ bool flat_active() const { return lane < dim; }
  __device__ __forceinline__ double diag() { return diag_flat<TS>(T, lane, w); }
  __device__ __forceinline__ void metric_point(double x) {
    xpt_ = (lane < dim) ? x : 0.0;
    if constexpr (RMETRIC == MM_RMETRIC_USER) {  // the products' point in natural order and its aux block
      w.ux[lane] = xpt_;
      wave_sync();
      mmuser::prepare(mmuser::WaveTeam{lane}, w.ux, dim, base_lds, w.uax);
      wave_sync();
    }
  }
  __device__ __forceinline__ double metric_apply(double v) {
    if constexpr (RMETRIC == MM_RMETRIC_USER) {
      // the user's metric_func at the point, entry by entry, contracted on the fly (matvec_flat's tile order)
      const int ti = lane >> 3, tj = lane & 7;
      if (lane < Geo<TS>::DP) w.vin[Geo<TS>::pos(lane)] = (lane < dim) ? v : 0.0;
      wave_sync();
      double part[TS];
#pragma unroll
      for (int a = 0; a < TS; ++a) {
        double s = 0.0;
#pragma unroll
        for (int b = 0; b < TS; ++b) {
          const int i = ti + 8 * a, j = tj + 8 * b;
          const double m = mmuser::entry_padded(w.ux, i, j, dim, base_lds, w.uax);
          s = __builtin_fma(m, w.vin[tj * TS + b], s);
        }
        part[a] = group8_sum(s);
      }
      if (tj == 0) {
#pragma unroll
        for (int a = 0; a < TS; ++a) w.vout[ti * TS + a] = part[a];
      }
      wave_sync();
      const double y = (lane < Geo<TS>::DP) ? w.vout[Geo<TS>::pos(lane)] : 0.0;
      wave_sync();
      return lane < dim ? y : 0.0;
    } else if constexpr (RMETRIC == MM_RMETRIC_RANK1) {
      // B v (the staged base matrix has this lane's TS x TS entries in matvec_flat's tile order) + x (x . v) / D
      const int ti = lane >> 3, tj = lane & 7;
      if (lane < Geo<TS>::DP) w.vin[Geo<TS>::pos(lane)] = (lane < dim) ? v : 0.0;
      wave_sync();
      double part[TS];
      const double* bt = base_lds + lane * Geo<TS>::TSTRIDE;
#pragma unroll
      for (int a = 0; a < TS; ++a) {
        double s = 0.0;
#pragma unroll
        for (int b = 0; b < TS; ++b) s = __builtin_fma(bt[a * TS + b], w.vin[tj * TS + b], s);
        part[a] = group8_sum(s);
      }
      if (tj == 0) {
#pragma unroll
        for (int a = 0; a < TS; ++a) w.vout[ti * TS + a] = part[a];
      }
      const double dot = wave_sum(lane < dim ? xpt_ * v : 0.0);
      wave_sync();
      const double y = (lane < Geo<TS>::DP) ? w.vout[Geo<TS>::pos(lane)] : 0.0;
      wave_sync();
      return lane < dim ? __builtin_fma(xpt_, dot * (1.0 / (double)dim), y) : 0.0;
    } else {  // diag(1 + x^2)
      return lane < dim ? __builtin_fma(xpt_ * xpt_, v, v) : 0.0;
    }
  }
  // 0.5 * vjp_metric_func(q)(V) of a user metric.  q is the point of the held inverse: build_metric() left it in w.uq
  // (natural order) with its aux block in w.uaq.  OUTER: V = -u u^T, else the explicit inverse in the tiles.
  template <bool OUTER>
  __device__ __forceinline__ double user_half_vjp(double u) {
    double r;
    if constexpr (mmuser::kFlatVjp) {
      if constexpr (OUTER) {
        mmuser::VjpOpsOuter<WaveBackend> ops{*this, lane < dim ? u : 0.0};
        r = mmuser::vjp_flat(ops, w.uq, lane, dim, base_lds, w.uaq);
      } else {
        mmuser::VjpOpsInv<WaveBackend> ops{*this};
        r = mmuser::vjp_flat(ops, w.uq, lane, dim, base_lds, w.uaq);
      }
    } else {
#if defined(MM_RTC_BUILD) && defined(MM_RTC_USER_METRIC)
      if constexpr (OUTER) {
        w.aux[lane] = (lane < dim) ? u : 0.0;
        wave_sync();
        const MmMat vm{nullptr, w.aux, 0};
        r = (lane < dim) ? mmuser::vjp_dense(w.uq, vm, lane, dim, base_lds, w.uaq) : 0.0;
      } else {
        // the tiles go to LDS as a dense DP x DP array, lane k evaluates element k
        const int ti = lane >> 3, tj = lane & 7;
#pragma unroll
        for (int a = 0; a < TS; ++a)
#pragma unroll
          for (int b = 0; b < TS; ++b) w.mat[(ti + 8 * a) * Geo<TS>::DP + tj + 8 * b] = T[a][b];
        wave_sync();
        const MmMat vm{w.mat, nullptr, Geo<TS>::DP};
        r = (lane < dim) ? mmuser::vjp_dense(w.uq, vm, lane, dim, base_lds, w.uaq) : 0.0;
      }
      wave_sync();
#else
      r = 0.0;
#endif
    }
    return lane < dim ? 0.5 * r : 0.0;
  }
  __device__ __forceinline__ double half_vjp_inv(double q) {
    if constexpr (RMETRIC == MM_RMETRIC_USER) return user_half_vjp<false>(0.0);
    else return half_vjp_tiles<TS, RMETRIC>(T, q, dim, lane, w, base_lds);
  }
  // dense metric: grad_quadratic_form_inv(p) = -(M^-1 p)(M^-1 p)^T   (matrices.py:1179-1181)
  __device__ __forceinline__ double dh2_dpos(double p, double q) {
    const double u = matvec_flat<TS>(T, p, lane, w);
    if constexpr (RMETRIC == MM_RMETRIC_USER) return user_half_vjp<true>(u);
    else return half_vjp_neg_outer<RMETRIC>(u, q, dim, lane, w, base_lds);
  }
  __device__ __forceinline__ double norm(double x, int kind) { return flat_norm(x, dim, lane, kind); }
  __device__ __forceinline__ double grad(double q) {
    return grad_flat<TS>(target, q, dim, lane, w, tparams);
  }
};

template <int TS, int RMETRIC>
__device__ __forceinline__ void implicit_leapfrog_body(const ImplicitArgs& A, double* lds) {
  double* base_lds = lds;
  const int base_elems = (RMETRIC == MM_RMETRIC_RANK1) ? 64 * Geo<TS>::TSTRIDE : 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* wl = lds + base_elems + wave * wave_lds_doubles<TS, RMETRIC>();
  stage_base<TS, RMETRIC>(base_lds, A.rparams, A.dim);

  const int64_t chain = (int64_t)blockIdx.x * kWaves + wave;
  if (chain >= A.n_chains) return;  // no block-level barrier below this point
  const int dim = A.dim;
  const bool act = lane < dim;
  double q = act ? A.pos[chain * dim + lane] : 0.0;
  double p = act ? A.mom[chain * dim + lane] : 0.0;
  const double t = signed_step(A.dir, A.step_scale, chain, A.step_size);

  WaveBackend<TS, RMETRIC> bk;
  bk.dim = dim;
  bk.lane = lane;
  bk.target = A.target;
  bk.w = make_wave_lds<TS, RMETRIC>(wl);
  bk.stash = wl + 320;
  bk.blk = wl + 320 + SL_COUNT_REFINE * 64;
  bk.base_lds = (RMETRIC == MM_RMETRIC_USER) ? A.rparams : base_lds;  // a user metric reads its params directly
  bk.refine_on = A.no_refine == 0;
  bk.lowrank_on_ = A.no_lowrank == 0 && A.no_refine == 0;
  bk.lr_refresh_ = A.lowrank_refresh;
  bk.tparams = A.tparams;
  bk.slot(SL_Q) = q;
  bk.slot(SL_P) = p;
  const ChainResult r = implicit_leapfrog_chain(bk, t, mmdev::chain_steps(A.chain_steps, chain, A.n_steps), A.opts);
  q = bk.slot(SL_Q);
  p = bk.slot(SL_P);

  // a failed step leaves q, p at the last completed step (they are only overwritten on success)
  if (act) {
    A.pos[chain * dim + lane] = q;
    A.mom[chain * dim + lane] = p;
  }
  if (lane == 0) {
    A.status[chain] = r.status;
    A.n_done[chain] = r.done;
    add_counters(A.counters, r);
  }
}

// ---- ImplicitMidpointIntegrator (integrators.py:547-681) on the wave-per-chain backend, D <= 64 -----------
template <int TS, int RMETRIC>
__device__ __forceinline__ void implicit_midpoint_body(const ImplicitArgs& A, double* lds) {
  double* base_lds = lds;
  const int base_elems = (RMETRIC == MM_RMETRIC_RANK1) ? 64 * Geo<TS>::TSTRIDE : 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* wl = lds + base_elems + wave * wave_lds_doubles<TS, RMETRIC>();
  stage_base<TS, RMETRIC>(base_lds, A.rparams, A.dim);
  const int64_t chain = (int64_t)blockIdx.x * kWaves + wave;
  if (chain >= A.n_chains) return;  // no block-level barrier below this point
  const int dim = A.dim;
  const bool act = lane < dim;
  WaveBackend<TS, RMETRIC> bk;
  bk.dim = dim;
  bk.lane = lane;
  bk.target = A.target;
  bk.w = make_wave_lds<TS, RMETRIC>(wl);
  bk.stash = wl + 320;
  bk.blk = wl + 320 + SL_COUNT_REFINE * 64;
  bk.base_lds = (RMETRIC == MM_RMETRIC_USER) ? A.rparams : base_lds;  // a user metric reads its params directly
  bk.refine_on = A.no_refine == 0;
  bk.lowrank_on_ = A.no_lowrank == 0 && A.no_refine == 0;
  bk.lr_refresh_ = A.lowrank_refresh;
  bk.tparams = A.tparams;
  bk.slot(MP_Q) = act ? A.pos[chain * dim + lane] : 0.0;
  bk.slot(MP_P) = act ? A.mom[chain * dim + lane] : 0.0;
  const double t = signed_step(A.dir, A.step_scale, chain, A.step_size);
  const ChainResult r = implicit_midpoint_chain(bk, t, mmdev::chain_steps(A.chain_steps, chain, A.n_steps), A.opts);
  if (act) {  // a failed step leaves the last completed state
    A.pos[chain * dim + lane] = bk.slot(MP_Q);
    A.mom[chain * dim + lane] = bk.slot(MP_P);
  }
  if (lane == 0) {
    A.status[chain] = r.status;
    A.n_done[chain] = r.done;
    add_counters(A.counters, r);
  }
}

// ---- System-level quantities for Riemannian systems: op 0 = h, 1 = dh_dmom, 2 = sample_momentum ------
template <int TS, int RMETRIC, int OP>
__device__ __forceinline__ void riemann_aux_body(const ImplicitArgs& A, double* lds) {
  double* base_lds = lds;
  const int base_elems = (RMETRIC == MM_RMETRIC_RANK1) ? 64 * Geo<TS>::TSTRIDE : 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* wl = lds + base_elems + wave * wave_lds_doubles<TS, RMETRIC>();
  const WaveLds w = make_wave_lds<TS, RMETRIC>(wl);
  stage_base<TS, RMETRIC>(base_lds, A.rparams, A.dim);
  const int64_t chain = (int64_t)blockIdx.x * kWaves + wave;
  if (chain >= A.n_chains) return;
  const int dim = A.dim;
  const bool act = lane < dim;
  const double q = act ? A.pos[chain * dim + lane] : 0.0;
  const double p = act ? A.mom[chain * dim + lane] : 0.0;
  double T[TS][TS];
  bool ok = build_metric<TS, RMETRIC>(T, q, dim, lane, w, (RMETRIC == MM_RMETRIC_USER) ? A.rparams : base_lds);
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
  if constexpr (OP == 0) {
    double logdet;
    ok = sweep_inverse<TS, true, false>(T, dim, lane, w, &logdet, nullptr) && ok;
    const double u = matvec_flat<TS>(T, p, lane, w);
    w.nat[lane] = act ? q : 0.0;
    wave_sync();
    const TargetAux aux = target_prepare<false>(A.target, w.nat, dim, A.tparams, lane);
    double e = act ? target_nld_elem<false>(A.target, aux, w.nat, lane, dim, A.tparams) + 0.5 * p * u : 0.0;
    e = wave_sum(e) + 0.5 * logdet;
    if (lane == 0) A.out[chain] = ok ? e : nan;
  } else if constexpr (OP == 1) {
    ok = sweep_inverse<TS, false, false>(T, dim, lane, w, nullptr, nullptr) && ok;
    const double u = matvec_flat<TS>(T, p, lane, w);
    if (act) A.out[chain * dim + lane] = ok ? u : nan;
  } else {
    w.aux[lane] = act ? A.z[chain * dim + lane] : 0.0;
    wave_sync();
    double y;
    ok = sweep_inverse<TS, false, true>(T, dim, lane, w, nullptr, &y) && ok;
    if (act) A.mom[chain * dim + lane] = ok ? y : nan;
  }
}

#ifndef MM_RTC_BUILD  // the in-tree instantiations (a run-time translation unit defines extern "C" wrappers instead)
template <int TS, int RMETRIC>
__global__ __launch_bounds__(64 * kWaves) void implicit_leapfrog_kernel(ImplicitArgs A) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  implicit_leapfrog_body<TS, RMETRIC>(A, lds);
}
template <int TS, int RMETRIC>
__global__ __launch_bounds__(64 * kWaves) void implicit_midpoint_kernel(ImplicitArgs A) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  implicit_midpoint_body<TS, RMETRIC>(A, lds);
}
template <int TS, int RMETRIC, int OP>
__global__ __launch_bounds__(64 * kWaves) void riemann_aux_kernel(ImplicitArgs A) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  riemann_aux_body<TS, RMETRIC, OP>(A, lds);
}
#endif

}  // namespace mmwave
