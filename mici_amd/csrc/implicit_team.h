// Device code of the VALU team kernels (k_implicit_large.hip instantiates it for the built-in metrics; mm_rtc.hip compiles
// it at run time around a USER metric, user_metric.h).
//
// Implicit leapfrog on dense-metric Riemannian systems with the metric held in the REGISTERS of a team
// of waves: one workgroup per chain, 32 < D <= 279 (BASELINE configs c3: D = 64, c4: D = 256).
// gfx950 / CDNA4.
//
// Same reference arithmetic as k_implicit.hip (the step itself is implicit_core.h); what changes is
// where the D x D metric lives.  Only the symmetric half is stored:
//   * the threads form the lower triangle of a PG x PG grid; thread (ti >= tj) owns the TS x TS
//     block-cyclic tile {(ti + PG a, tj + PG b)}.  PG * TS >= D.
//   * the symmetric sweep operator keeps the matrix symmetric, so the mirrored tiles are never needed:
//     step k publishes column k (from the tiles of grid column k % PG and, transposed, of grid row
//     k % PG) into LDS; every thread then applies at(a, b) -= m[a] * c[b] (TS^2 v_fma_f64) from 2 TS
//     LDS operands.
//   * M^-1 v: each tile contributes to TS "row" and (off-diagonal tiles) TS "column" partial sums, laid
//     out in LDS so that every output element has exactly PG private slots -> deterministic reduction.
// Two geometries are instantiated (TeamCfg below): 31 x 31 grid of 9 x 9 tiles on 512 threads for
// D <= 279 (one chain per CU half), and 15 x 15 grid of 5 x 5 tiles on 128 threads for D <= 75.
#pragma once
#include "implicit_core.h"
#include "user_metric.h"

namespace mmteam {

using namespace mmdev;
using namespace mmimp;

// Geometry of a team: PG x PG thread grid (lower triangle owns tiles), TS x TS block-cyclic tiles.
//   TeamCfg<31, 9, 512, true>  64 < D <= 279: a whole CU per chain, last tile row parked in LDS
//   TeamCfg<15, 5, 128, false> 32 < D <= 75 : two waves per chain (25 doubles of metric per thread), eight
//                              chains per CU, so that BASELINE c3 (1024 chains) puts two waves on every
//                              SIMD - a lone wave issues FP64 VALU at only ~half rate on gfx950 (measured)
template <int PG_, int TS_, int NT_, bool PARK_, int MINW_, int NB_>
struct TeamCfg {
  static constexpr int NB = NB_;                    // pivot columns per block of the blocked sweep
  static constexpr int MINW = MINW_;                // waves per SIMD the register allocation targets
  static constexpr int PG = PG_;                    // process-grid side
  static constexpr int TS = TS_;                    // tile side
  static constexpr int DP = PG_ * TS_;              // padded dimension
  static constexpr int NT = NT_;                    // threads per workgroup
  static constexpr int NTILE = PG_ * (PG_ + 1) / 2; // tile-owning threads
  static constexpr int GS = TS_ + (TS_ & 1);        // doubles per grid group in a permuted LDS vector (even)
  static constexpr int PV = PG_ * GS;               // permuted vector length
  static constexpr int SLOTS = PG_ + 2;             // partial-sum slots per output element (+2 pad)
  static constexpr int VL = ((DP + 7) / 8) * 8 + 8; // natural-order vector length (>= DP + 1)
  static constexpr bool PARK = PARK_;               // park the last tile row in LDS (register relief)
  static constexpr int TREG = PARK_ ? TS_ - 1 : TS_;
  // mat-vec partial sums [PG][TS][SLOTS]; the blocked sweep's panel buffers [2][2][NB][PV] alias them
  static constexpr int PART = PG_ * TS_ * SLOTS > 4 * NB_ * PV ? PG_ * TS_ * SLOTS : 4 * NB_ * PV;
  static constexpr int BASE_DOUBLES =
      3 * PV + PART + 2 * VL + 16 + mmimp::SL_COUNT * VL + (PARK_ ? TS_ * NT_ : 0);
  // Refinement of the solve-only constructions (implicit_core.h refine_solve, round 4: VERDICT r03 "missing" #5): two more
  // step slots (the solves' starting guesses) and one scratch vector of the iteration - its other two live in the sweep's
  // column buffers, idle while a solve iterates - plus, for the built-in metrics, the products' point in permuted order.
  // A user metric (user_metric.h) keeps the point of the held inverse in natural order and its aux block; the PRODUCTS'
  // point and aux block alias the two natural-order scratch vectors (nat / aux: only written and read inside one call of
  // grad / the VJPs), so a source whose aux block is larger than one of those - or that no longer fits the CU's LDS -
  // runs without the refinement (every construction factorised, as in round 3).
  static constexpr int kLdsMax = 160 * 1024 / 8;
  static constexpr int KA = (mmuser::kAux + 1) & ~1;
  static constexpr int REFINE_EXTRA = (mmimp::SL_COUNT_REFINE - mmimp::SL_COUNT) * VL + VL;
  static constexpr bool REFINE = BASE_DOUBLES + REFINE_EXTRA + PV <= kLdsMax;
  static constexpr int LDS_DOUBLES = BASE_DOUBLES + (REFINE ? REFINE_EXTRA + PV : 0);
  static constexpr bool USER_REFINE = KA <= VL && BASE_DOUBLES + REFINE_EXTRA + VL + KA <= kLdsMax;
  static constexpr int USER_LDS_DOUBLES = BASE_DOUBLES + VL + KA + (USER_REFINE ? REFINE_EXTRA : 0);
  static_assert(NTILE <= NT_, "not enough threads for the tile triangle");
  __device__ static __forceinline__ int ppos(int i) { return (i % PG_) * GS + i / PG_; }
};
using CfgLarge = TeamCfg<31, 9, 512, true, 2, 4>;
using CfgSmall = TeamCfg<15, 5, 128, false, 3, 1>;
using CfgMid = TeamCfg<22, 3, 256, false, 4, 1>;

// Launder a lane-varying index so that address arithmetic derived from it is recomputed where it is
// used instead of being hoisted out of the step loop into long-lived VGPRs (the register file is full
// of metric tiles; a few integer ops per use are free).
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

struct BlockLds {
  double* col0;
  double* col1;
  double* vin;
  double* part;  // [PG][TS][SLOTS]
  double* nat;   // natural order [DP + pad]
  double* aux;   // natural order [DP + pad]
  double* red;   // [16]
  double* stash; // [SL_COUNT][VL] per-thread flat state of the step (keeps it out of VGPRs)
  double* trow;  // [TS][NT] the last tile row of every thread (register relief, see BlockBackend::at)
  double* uq;    // user metrics: [VL] the point of the held inverse in natural order, [kAux] its aux block
  double* uaq;
  double* rsd;   // refinement: [VL] the third scratch vector of refine_solve (the other two: col0 / col1)
  double* xq;    // refinement, built-in metrics: [PV] the products' point, permuted
  double* ux;    // refinement, user metrics: the products' point in natural order and its aux block (= nat / aux)
  double* uax;
};

template <class C>
__device__ __forceinline__ double block_reduce(double v, int kind_max, double* red) {
  // kind_max: 0 sum, 1 NaN-propagating max.  Uniform result; two barriers.
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = kind_max ? wave_max(v) : wave_sum(v);
  if (lane == 0) red[wave] = v;
  __syncthreads();
  double r = red[0];
#pragma unroll
  for (int w = 1; w < C::NT / 64; ++w) r = kind_max ? nanmax(r, red[w]) : r + red[w];
  __syncthreads();
  return r;
}

// the workgroup as a team (user_metric.h mm_user_prepare)
template <class C>
struct TeamOf {
  double* red;
  int tid;
  __device__ __forceinline__ int rank() const { return tid; }
  __device__ __forceinline__ int size() const { return C::NT; }
  __device__ __forceinline__ double sum(double x) const { return block_reduce<C>(x, 0, red); }
};

template <class C, int RMETRIC, bool LOWRANK = false>
struct BlockBackend {
  // implicit_core.h lowrank_solve / lowrank_update (round 6): the built-in rank-one-update metric's
  // solve-only constructions by the Woodbury identity from the held inverse, the inverse carried from step to step by the
  // symmetric rank-two update (the launcher picks the instantiation: MICI_AMD_LOWRANK=0 is the CG refinement)
  static constexpr bool kLowRankBuiltin = RMETRIC == MM_RMETRIC_RANK1;
  static constexpr bool kLowRank = LOWRANK && kLowRankBuiltin;
  __device__ static constexpr bool lowrank_on() { return true; }
  int lr_refresh_;
  __device__ __forceinline__ int lowrank_refresh() const { return lr_refresh_; }
  __device__ __forceinline__ double lowrank_scale() const { return (double)dim; }
  __device__ __forceinline__ double lowrank_vec(double x) const { return x; }
  __device__ __forceinline__ double& lowrank_u0() { return slot(mmimp::SL_Q); }
  __device__ __forceinline__ void sum3(double a, double b, double c, double* sa, double* sb, double* sc) {
    *sa = sum1(a);
    *sb = sum1(b);
    *sc = sum1(c);
  }
  static constexpr bool kSolveByInverse = true;  // implicit_core.h: solve = invert + mat-vec, one construction site
  static constexpr bool kUnifiedConstruct = false;
  static constexpr bool kCountersInLds = false;
  static constexpr int PG = C::PG, TS = C::TS, DP = C::DP, NT = C::NT, GS = C::GS, SLOTS = C::SLOTS, VL = C::VL;
  __device__ static __forceinline__ int ppos(int i) { return C::ppos(i); }
  // Tile storage: rows 0..TS-2 in registers (144 VGPRs), row TS-1 in LDS.  The compiler could not
  // keep all 81 doubles plus the sweep operands inside the 256-VGPR budget of a wave here and spilled
  // tile entries to scratch (L2-bound: measured 8x slowdown of the sweep); parking one row in LDS by
  // hand removes the spills at the price of 18 LDS accesses per sweep step.
  double Treg[C::TREG][TS];
  __device__ __forceinline__ double& at(int a, int b) {
    if constexpr (C::PARK) {
      return a < TS - 1 ? Treg[a][b] : w.trow[b * NT + tid];
    } else {
      return Treg[a][b];
    }
  }
  int dim, tid, ti, tj, target;
  double inv_dim_;  // 1 / D of the rank-one metric (multiplied with: an IEEE division is ~30 dependent instructions)
  bool tile;  // this thread owns a tile
  BlockLds w;
  const double* base;  // global (L2-resident) base matrix of the rank-one metric, zero-padded DP x DP; user metric: its params
  const double* tparams;
  double* work;        // user metric with the dense-accessor VJP: this chain's DP x DP doubles of global memory
  // implicit_core.h: solve-only constructions refined from the held inverse (when the LDS layout has room: TeamCfg)
  static constexpr bool kRefine = RMETRIC == MM_RMETRIC_USER ? C::USER_REFINE : C::REFINE;
  static constexpr int kSlots = kRefine ? mmimp::SL_COUNT_REFINE : mmimp::SL_COUNT;
  bool refine_on;  // false: MICI_AMD_REFINE=0, every construction is factorised
  __device__ __forceinline__ bool flat_active() const { return tid < dim; }
  __device__ __forceinline__ double sum1(double a) { return block_reduce<C>(tid < dim ? a : 0.0, 0, w.red); }
  __device__ __forceinline__ void sum2(double a, double b, double* sa, double* sb) {
    *sa = sum1(a);
    *sb = sum1(b);
  }
  // scratch of refine_solve: no sweep runs while a solve iterates, so two of its vectors live in the sweep's column buffers
  __device__ __forceinline__ double& rslot(int i) {
    double* const v = i == mmimp::RS_U ? w.col0 : (i == mmimp::RS_R ? w.col1 : w.rsd);
    return v[tid < DP ? tid : DP];
  }

  // flat state only exists for tid < DP; the other threads share one dummy cell per slot
  __device__ __forceinline__ double& slot(int i) { return w.stash[i * VL + (tid < DP ? tid : VL - 1)]; }

  __device__ __forceinline__ double norm(double x, int kind) {
    const double a = tid < dim ? x : 0.0;
    if (kind == MM_NORM_LINF) return block_reduce<C>(fabs(a), 1, w.red);
    return sqrt(block_reduce<C>(a * a, 0, w.red));
  }

  // metric_func(x) into the tiles; returns false if any entry is not finite
  __device__ __forceinline__ bool build(double x) {
    if (tid < DP) w.vin[ppos(tid)] = (tid < dim) ? x : 0.0;
    if constexpr (RMETRIC == MM_RMETRIC_USER) {  // the point in natural order for the user's hooks, then its aux block
      if (tid < VL) w.uq[tid] = (tid < dim) ? x : 0.0;
      __syncthreads();
      mmuser::prepare(TeamOf<C>{w.red, tid}, w.uq, dim, base, w.uaq);
    }
    __syncthreads();
    double chk = 0.0;
    const int ti = opaque(this->ti), tj = opaque(this->tj);
    {  // every thread builds a tile (threads >= 496 duplicate tile (30,30)): the tiles are then fully
       // re-defined here, i.e. dead before this point, which frees their registers for scalar work
      // `base` is the rank-one metric's base matrix zero-padded to DP x DP on the host, and x is 0 on the
      // padding, so the closed form is exactly 0 on padded entries; the padded diagonal is set to 1 below.
      double qc[TS];
#pragma unroll
      for (int b = 0; b < TS; ++b) qc[b] = w.vin[tj * GS + b];
      const double inv_d = 1.0 / (double)dim;
#pragma unroll
      for (int a = 0; a < TS; ++a) {
        // one tile row at a time: 9 loads with immediate offsets from one row pointer
        const double qa = w.vin[ti * GS + a] * inv_d;
        if constexpr (RMETRIC == MM_RMETRIC_RANK1) {
          const double* brow = base + (int64_t)(ti + PG * a) * DP + tj;
#pragma unroll
          for (int b = 0; b < TS; ++b) at(a, b) = __builtin_fma(qa, qc[b], brow[PG * b]);
        } else if constexpr (RMETRIC == MM_RMETRIC_USER) {
          // the user's metric_func, entry by entry (zero on the padding; its diagonal is set to 1 below)
#pragma unroll
          for (int b = 0; b < TS; ++b) {
            const int i = ti + PG * a, j = tj + PG * b;
            at(a, b) = mmuser::entry_padded(w.uq, i, j, dim, base, w.uaq);
          }
        } else {
#pragma unroll
          for (int b = 0; b < TS; ++b) at(a, b) = 0.0;
        }
        if constexpr (C::PARK) __builtin_amdgcn_sched_barrier(0);
      }
      if (ti == tj) {
#pragma unroll
        for (int a = 0; a < TS; ++a) {
          if constexpr (RMETRIC == MM_RMETRIC_DIAGQUAD) at(a, a) = __builtin_fma(qc[a], qc[a], 1.0);
          if (ti + PG * a >= dim) at(a, a) = 1.0;
        }
      }
#pragma unroll
      for (int a = 0; a < TS; ++a)
#pragma unroll
        for (int b = 0; b < TS; ++b) chk = __builtin_fma(at(a, b), 0.0, chk);
    }
    // NaN in chk <=> some entry is inf/NaN ("Array is not finite.", matrices.py:211-215)
    const double bad = block_reduce<C>(chk == 0.0 ? 0.0 : 1.0, 0, w.red);
    return bad == 0.0;
  }

  // symmetric sweep: tiles <- M^-1 (lower-triangular tile set); false if a pivot is not > 0
  template <bool LOGDET, bool CHOLVEC>
  __device__ __forceinline__ bool sweep(double* logdet, double* chol_y) {
    bool ok = true;
    double ld = 0.0, y = 0.0;
#pragma unroll
    for (int kb = 0; kb < TS; ++kb) {
#pragma unroll 1
      for (int kt = 0; kt < PG; ++kt) {
        const int k = kb * PG + kt;
        const int ti = opaque(this->ti), tj = opaque(this->tj);
        double* col = (k & 1) ? w.col1 : w.col0;
        // publish column k: grid column kt holds rows ti + PG a of it; grid row kt (tj < kt) holds, by
        // symmetry, entries (k, tj + PG b)
        // (the small geometry uses one store sequence with selected operands: with two branches the
        // compiler spilt its tile to scratch to merge them; the large one is the other way round)
        if constexpr (C::PARK) {
          if (tile) {
            if (tj == kt) {
#pragma unroll
              for (int a = 0; a < TS; ++a) col[ti * GS + a] = at(a, kb);
            } else if (ti == kt) {
#pragma unroll
              for (int b = 0; b < TS; ++b) col[tj * GS + b] = at(kb, b);
            }
          }
        } else {
          const bool pc = (tj == kt), pr = (ti == kt);
          if (tile && (pc || pr)) {
            double* dst = col + (pc ? ti : tj) * GS;
#pragma unroll
            for (int a = 0; a < TS; ++a) dst[a] = pc ? at(a, kb) : at(kb, a);
          }
        }
        __syncthreads();
        const double piv = col[kt * GS + kb];
        ok = ok && (piv > 0.0);
        const double d = fast_rcp(piv);
        if constexpr (LOGDET) ld += log(piv);
        if constexpr (CHOLVEC) {
          const double rs = 1.0 / sqrt(piv);
          if (tid >= k && tid < dim) y += (col[ppos(tid)] * rs) * w.aux[k];
        }
        {
          double ac[TS];
#pragma unroll
          for (int b = 0; b < TS; ++b) ac[b] = col[tj * GS + b];
          if (tj == kt) ac[kb] = piv - 1.0;
          // one tile row at a time: only one row multiplier is live (register budget: 256 / wave)
          // (PARK: rows are kept apart by scheduling barriers so that the register-starved large geometry
          // does not hoist all row multipliers; the next row's multiplier is prefetched by hand instead)
          double mnext = col[ti * GS];
#pragma unroll
          for (int a = 0; a < TS; ++a) {
            double m = mnext * d;
            if (a + 1 < TS) mnext = col[ti * GS + a + 1];
            if (a == kb && ti == kt) m = 1.0 - d;
#pragma unroll
            for (int b = 0; b < TS; ++b) at(a, b) = __builtin_fma(-m, ac[b], at(a, b));
            if constexpr (C::PARK) __builtin_amdgcn_sched_barrier(0);
          }
          if (ti == kt && tj == kt) at(kb, kb) -= 2.0;
        }
        // the next step publishes into the other buffer; the barrier of that step orders reuse
      }
    }
#pragma unroll
    for (int a = 0; a < TS; ++a)
#pragma unroll
      for (int b = 0; b < TS; ++b) at(a, b) = -at(a, b);
    __syncthreads();
    if constexpr (LOGDET) *logdet = ld;
    if constexpr (CHOLVEC) *chol_y = y;
    return ok;
  }


  // ---- blocked symmetric sweep -------------------------------------------------------------------
  // Sweeping the NBK columns K = {kb PG + kt0 + s} at once (P = A_KK, Q = A_:K):
  //     A_RR -= Q_R P^-1 Q_R^T,   A_RK = Q_R P^-1,   A_KK = -P^-1
  // is, for ALL entries uniformly, the rank-NBK update  A -= W X^T  with
  //     W_i = Q_i P^-1 (i not in K),  W_K = I - P^-1;     X_j = Q_j (j not in K),  X_K = P - I,
  // followed by  A_KK -= 2 I  (the scalar sweep above is the NBK = 1 case).  One block costs two
  // barriers instead of NBK: (1) publish the panel Q; (2) threads tid < DP turn "their" row of Q
  // into a row of W (every thread inverts the tiny P redundantly from broadcast LDS reads - that also
  // gives every thread the pivots for the positive-definiteness check and logdet); (3) NBK
  // back-to-back rank-one tile updates with no synchronisation in between, so their LDS operand
  // loads pipeline.  Panel/W buffers are double-buffered by block parity and alias the mat-vec's
  // partial-sum area (idle during a sweep).
  template <int NBK, bool LOGDET>
  __device__ __forceinline__ void block_step(int kb, int kt0, int par, bool& ok, double& ld) {
    constexpr int PV = C::PV;
    const int ti = opaque(this->ti), tj = opaque(this->tj), tid = opaque(this->tid);
    double* Qp = w.part + par * (2 * C::NB * PV);
    double* Wp = Qp + C::NB * PV;
    // (1) publish: grid column kt0+s holds rows ti + PG a of column s; grid row kt0+s holds (tj < ti),
    // by symmetry, its rows tj + PG b
    if (tile) {
      const int sc = tj - kt0, sr = ti - kt0;
      if (sc >= 0 && sc < NBK) {
#pragma unroll
        for (int a = 0; a < TS; ++a) Qp[sc * PV + ti * GS + a] = at(a, kb);
      }
      if (sr >= 0 && sr < NBK && tj != ti) {
#pragma unroll
        for (int b = 0; b < TS; ++b) Qp[sr * PV + tj * GS + b] = at(kb, b);
      }
    }
    __syncthreads();
    // (2) P^-1 by Gauss-Jordan in registers (uniform across the workgroup)
    {
      double Pm[NBK][NBK];
#pragma unroll
      for (int s = 0; s < NBK; ++s)
#pragma unroll
        for (int t = 0; t < NBK; ++t) {
          // a block hanging over the end of the grid row (PG % NBK != 0) sees identity columns there
          const bool in = (kt0 + s < PG) && (kt0 + t < PG);
          Pm[s][t] = in ? Qp[t * PV + (kt0 + (in ? s : 0)) * GS + kb] : (s == t ? 1.0 : 0.0);
        }
#pragma unroll
      for (int k = 0; k < NBK; ++k) {
        const double piv = Pm[k][k];
        ok = ok && (piv > 0.0);
        if constexpr (LOGDET) ld += log(piv);
        const double d = fast_rcp(piv);
        double rk[NBK];
#pragma unroll
        for (int t = 0; t < NBK; ++t) rk[t] = Pm[k][t] * d;
#pragma unroll
        for (int s = 0; s < NBK; ++s) {
          if (s == k) continue;
          const double f = Pm[s][k];
#pragma unroll
          for (int t = 0; t < NBK; ++t)
            if (t != k) Pm[s][t] = __builtin_fma(-f, rk[t], Pm[s][t]);
          Pm[s][k] = -f * d;
        }
#pragma unroll
        for (int t = 0; t < NBK; ++t) Pm[k][t] = rk[t];
        Pm[k][k] = d;
      }
      // Pm = P^-1 (plain Gauss-Jordan with the pivot row scaled: no sign convention needed here)
      if (tid < DP) {
        const int g = tid % PG, a = tid / PG, pp = g * GS + a;
        double qi[NBK];
#pragma unroll
        for (int t = 0; t < NBK; ++t) qi[t] = (kt0 + t < PG) ? Qp[t * PV + pp] : 0.0;
        const int r = g - kt0;
        const bool in_k = (a == kb) && r >= 0 && r < NBK;
#pragma unroll
        for (int s = 0; s < NBK; ++s) {
          double wv = 0.0;
#pragma unroll
          for (int t = 0; t < NBK; ++t) wv = __builtin_fma(qi[t], Pm[t][s], wv);
          if (in_k) {
            double y = 0.0;
#pragma unroll
            for (int t = 0; t < NBK; ++t) y = (r == t) ? Pm[t][s] : y;
            wv = ((r == s) ? 1.0 : 0.0) - y;
          }
          Wp[s * PV + pp] = wv;
        }
      }
    }
    __syncthreads();
    // (3) rank-NBK update of every tile
#pragma unroll
    for (int s = 0; s < NBK; ++s) {
      if (kt0 + s >= PG) break;  // uniform: only the overhanging tail of the last block
      double ac[TS];
#pragma unroll
      for (int b = 0; b < TS; ++b) ac[b] = Qp[s * PV + tj * GS + b];
      if (tj == kt0 + s) ac[kb] -= 1.0;
      double mnext = Wp[s * PV + ti * GS];
#pragma unroll
      for (int a = 0; a < TS; ++a) {
        const double m = mnext;
        if (a + 1 < TS) mnext = Wp[s * PV + ti * GS + a + 1];
#pragma unroll
        for (int b = 0; b < TS; ++b) at(a, b) = __builtin_fma(-m, ac[b], at(a, b));
        if constexpr (C::PARK) __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (ti == tj && ti >= kt0 && ti < kt0 + NBK) at(kb, kb) -= 2.0;
  }

  template <bool LOGDET>
  __device__ __forceinline__ bool sweep_blocked(double* logdet) {
    constexpr int NB = C::NB;
    bool ok = true;
    double ld = 0.0;
    int par = 0;
#pragma unroll
    for (int kb = 0; kb < TS; ++kb) {
#pragma unroll 1
      for (int kt0 = 0; kt0 < PG; kt0 += NB) {
        block_step<NB, LOGDET>(kb, kt0, par, ok, ld);
        par ^= 1;
      }
    }
#pragma unroll
    for (int a = 0; a < TS; ++a)
#pragma unroll
      for (int b = 0; b < TS; ++b) at(a, b) = -at(a, b);
    __syncthreads();
    if constexpr (LOGDET) *logdet = ld;
    return ok;
  }

  // blocked where it pays (measured: +11% on c4; the small geometries are instruction-issue bound and the
  // redundant P^-1 of a block costs them more than the saved barriers)
  template <bool LOGDET>
  __device__ __forceinline__ bool invert(double* logdet) {
    if constexpr (C::NB > 1) return sweep_blocked<LOGDET>(logdet);
    else return sweep<LOGDET, false>(logdet, nullptr);
  }

  __device__ __forceinline__ bool build_and_invert(double x) {
    bool ok = build(x);
    ok = invert<false>(nullptr) && ok;
    return ok;
  }

  __device__ __forceinline__ bool build_and_solve(double x, double rhs, double* u) {
    const bool ok = build_and_invert(x);
    *u = matvec(rhs);
    return ok;
  }

  // ---- M(x) v of the refinement solves, from the entries of metric_func(x) evaluated where they are used ----------------
  __device__ __forceinline__ void metric_point(double x) {
    if constexpr (RMETRIC == MM_RMETRIC_USER) {
      if (tid < VL) w.ux[tid] = (tid < dim) ? x : 0.0;
      __syncthreads();
      mmuser::prepare(TeamOf<C>{w.red, tid}, w.ux, dim, base, w.uax);
      __syncthreads();
    } else {
      if (tid < DP) w.xq[ppos(tid)] = (tid < dim) ? x : 0.0;  // (read back by the other threads after metric_apply's barrier)
    }
  }
  __device__ __forceinline__ double metric_apply(double v) {
    if constexpr (RMETRIC == MM_RMETRIC_DIAGQUAD) {
      const double x = tid < DP ? w.xq[ppos(tid)] : 0.0;
      return tid < dim ? __builtin_fma(x * x, v, v) : 0.0;
    } else {
      // as matvec() below, with the tile entries generated instead of read: row partials of tile (ti, tj) to slot tj of
      // its rows, column partials (the mirrored tile) to slot ti of its columns, summed in slot order
      const int tid = opaque(this->tid), ti = opaque(this->ti), tj = opaque(this->tj);
      if (tid < DP) w.vin[ppos(tid)] = (tid < dim) ? v : 0.0;
      __syncthreads();
      {
        // CB columns of the tile at a time: next to the held inverse (128 - 144 registers of a thread) the whole tile's
        // operands and column sums do not fit, and what spills is the inverse.  A row's partial sum continues through
        // LDS from one column chunk to the next (same thread, same order of additions as one pass).
        constexpr int CB = C::PARK ? 3 : TS;
        static_assert(TS % CB == 0, "column chunks");
        const double inv_d = 1.0 / (double)dim;
#pragma unroll 1
        for (int b0 = 0; b0 < TS; b0 += CB) {
          double xc[CB], cs[CB];
#pragma unroll
          for (int b = 0; b < CB; ++b) {
            xc[b] = w.vin[tj * GS + b0 + b];
            cs[b] = 0.0;
          }
#pragma unroll
          for (int a = 0; a < TS; ++a) {
            const double xra = w.vin[ti * GS + a];
            double* const ps = w.part + (ti * TS + a) * SLOTS + tj;
            double s = (b0 == 0 || !tile) ? 0.0 : *ps;
            if constexpr (RMETRIC == MM_RMETRIC_RANK1) {  // B_ij + x_i x_j / D (base zero-padded, x zero on the padding)
              const double qa = w.xq[ti * GS + a] * inv_d;
              const double* brow = base + (int64_t)(ti + PG * a) * DP + tj + PG * b0;
#pragma unroll
              for (int b = 0; b < CB; ++b) {
                const double e = __builtin_fma(qa, w.xq[tj * GS + b0 + b], brow[PG * b]);
                s = __builtin_fma(e, xc[b], s);
                cs[b] = __builtin_fma(e, xra, cs[b]);
              }
            } else {
#pragma unroll
              for (int b = 0; b < CB; ++b) {
                const double e = mmuser::entry_padded(w.ux, ti + PG * a, tj + PG * (b0 + b), dim, base, w.uax);
                s = __builtin_fma(e, xc[b], s);
                cs[b] = __builtin_fma(e, xra, cs[b]);
              }
            }
            if (tile) *ps = s;
          }
          if (tile && ti != tj) {
#pragma unroll
            for (int b = 0; b < CB; ++b) w.part[(tj * TS + b0 + b) * SLOTS + ti] = cs[b];
          }
        }
      }
      __syncthreads();
      double y = 0.0;
      if (tid < DP) {
        const double* src = w.part + ((tid % PG) * TS + tid / PG) * SLOTS;
#pragma unroll
        for (int sl = 0; sl < PG; ++sl) y += src[sl];
      }
      __syncthreads();
      return tid < dim ? y : 0.0;
    }
  }

  __device__ __forceinline__ double matvec(double v) {
    const int tid = opaque(this->tid), ti = opaque(this->ti), tj = opaque(this->tj);
    if (tid < DP) w.vin[ppos(tid)] = (tid < dim) ? v : 0.0;
    __syncthreads();
    {
      double xr[TS], xc[TS];
#pragma unroll
      for (int a = 0; a < TS; ++a) {
        xr[a] = w.vin[ti * GS + a];
        xc[a] = w.vin[tj * GS + a];
      }
      // row partials: y[ti + 31 a] += sum_b at(a, b) x[tj + 31 b]  -> slot tj of group ti
#pragma unroll
      for (int a = 0; a < TS; ++a) {
        double s = 0.0;
#pragma unroll
        for (int b = 0; b < TS; ++b) s = __builtin_fma(at(a, b), xc[b], s);
        if (tile) w.part[(ti * TS + a) * SLOTS + tj] = s;
      }
      // column partials of off-diagonal tiles (the mirrored tile): y[tj + 31 b] += sum_a at(a, b) x[ti + 31 a]
#pragma unroll
      for (int b = 0; b < TS; ++b) {
        double s = 0.0;
#pragma unroll
        for (int a = 0; a < TS; ++a) s = __builtin_fma(at(a, b), xr[a], s);
        if (tile && ti != tj) w.part[(tj * TS + b) * SLOTS + ti] = s;
      }
    }
    __syncthreads();
    double y = 0.0;
    if (tid < DP) {
      const double* src = w.part + ((tid % PG) * TS + tid / PG) * SLOTS;
#pragma unroll
      for (int sl = 0; sl < PG; ++sl) y += src[sl];
    }
    __syncthreads();
    return tid < dim ? y : 0.0;
  }

  // implicit_core.h lowrank_update: the tiles += a u^T + b v^T with u = al a + be b, v = be a + ga b.  Thread (ti >= tj) owns
  // the entries (ti + PG a', tj + PG b'): its row operands are group ti of the permuted vectors a, b, its column operands
  // group tj of u, v (three of the four vectors in the idle partial-sum buffer)
  __device__ __forceinline__ void inverse_update(double al, double be, double ga, double a, double b) {
    const int tid = opaque(this->tid), ti = opaque(this->ti), tj = opaque(this->tj);
    constexpr int PVL = C::PV;
    if (tid < DP) {
      const bool act = tid < dim;
      const double am = act ? a : 0.0, bm = act ? b : 0.0;
      const int pp = ppos(tid);
      w.vin[pp] = am;
      w.part[pp] = bm;
      w.part[PVL + pp] = __builtin_fma(al, am, be * bm);
      w.part[2 * PVL + pp] = __builtin_fma(be, am, ga * bm);
    }
    __syncthreads();
    if (tile) {
      double ar[TS], br[TS], uc[TS], vc[TS];
#pragma unroll
      for (int k = 0; k < TS; ++k) {
        ar[k] = w.vin[ti * GS + k];
        br[k] = w.part[ti * GS + k];
        uc[k] = w.part[PVL + tj * GS + k];
        vc[k] = w.part[2 * PVL + tj * GS + k];
      }
#pragma unroll
      for (int x = 0; x < TS; ++x)
#pragma unroll
        for (int y = 0; y < TS; ++y) at(x, y) = __builtin_fma(ar[x], uc[y], __builtin_fma(br[x], vc[y], at(x, y)));
    }
    __syncthreads();
  }

  __device__ __forceinline__ double diag() {
    if (tile && ti == tj) {
#pragma unroll
      for (int a = 0; a < TS; ++a) w.vin[ti * GS + a] = at(a, a);
    }
    __syncthreads();
    const double y = (tid < dim) ? w.vin[ppos(tid)] : 0.0;
    __syncthreads();
    return y;
  }

  // 0.5 * vjp_metric_func(q)(V) of a user metric.  q is the point of the held inverse: build() left it in w.uq with its
  // aux block in w.uaq.  OUTER: V = -u u^T, else the explicit inverse in the tiles - handed to the user's team-form hook as
  // it is (user_metric.h MM_USER_VJP_FLAT), or dumped to the chain's dense global array for V(i, j).
  template <bool OUTER>
  __device__ __forceinline__ double user_half_vjp(double u) {
    double r;
    if constexpr (mmuser::kFlatVjp) {
      if constexpr (OUTER) {
        mmuser::VjpOpsOuter<BlockBackend> ops{*this, tid < dim ? u : 0.0};
        r = mmuser::vjp_flat(ops, w.uq, tid, dim, base, w.uaq);
      } else {
        mmuser::VjpOpsInv<BlockBackend> ops{*this};
        r = mmuser::vjp_flat(ops, w.uq, tid, dim, base, w.uaq);
      }
    } else {
#if defined(MM_RTC_BUILD) && defined(MM_RTC_USER_METRIC)
      if constexpr (OUTER) {
        if (tid < VL) w.aux[tid] = (tid < dim) ? u : 0.0;
        __syncthreads();
        const MmMat vm{nullptr, w.aux, 0};
        r = (tid < dim) ? mmuser::vjp_dense(w.uq, vm, tid, dim, base, w.uaq) : 0.0;
        __syncthreads();
      } else {
        if (tile) {
          const int ti = opaque(this->ti), tj = opaque(this->tj);
#pragma unroll
          for (int a = 0; a < TS; ++a)
#pragma unroll
            for (int b = 0; b < TS; ++b) {
              const double v = at(a, b);
              work[(ti + PG * a) * DP + tj + PG * b] = v;
              work[(tj + PG * b) * DP + ti + PG * a] = v;
            }
        }
        __syncthreads();  // (workgroup-scope release / acquire of the global stores)
        const MmMat vm{work, nullptr, DP};
        r = (tid < dim) ? mmuser::vjp_dense(w.uq, vm, tid, dim, base, w.uaq) : 0.0;
        __syncthreads();
      }
#else
      r = 0.0;
#endif
    }
    return tid < dim ? 0.5 * r : 0.0;
  }

  __device__ __forceinline__ double half_vjp_inv(double q) {
    if constexpr (RMETRIC == MM_RMETRIC_USER) return user_half_vjp<false>(0.0);
    else if constexpr (RMETRIC == MM_RMETRIC_RANK1) return matvec(q) * inv_dim_;
    else return q * diag();
  }

  // dense metric: grad_quadratic_form_inv(p) = -(M^-1 p)(M^-1 p)^T   (matrices.py:1179-1181)
  __device__ __forceinline__ double dh2_dpos(double p, double q) {
    const double u = matvec(p);
    if constexpr (RMETRIC == MM_RMETRIC_USER) {
      return user_half_vjp<true>(u);
    } else if constexpr (RMETRIC == MM_RMETRIC_RANK1) {
      const double uq = block_reduce<C>(tid < dim ? u * q : 0.0, 0, w.red);
      return -(u * uq) * inv_dim_;
    } else {
      return -q * (u * u);
    }
  }

  __device__ __forceinline__ double grad(double q) {
    if (tid < VL) w.nat[tid] = (tid < dim) ? q : 0.0;
    __syncthreads();
    const TargetAux aux = target_prepare<false>(target, w.nat, dim, tparams, threadIdx.x & 63);
    const double g = (tid < dim) ? target_grad_elem<false>(target, aux, w.nat, tid, dim, tparams) : 0.0;
    __syncthreads();
    return g;
  }

  __device__ __forceinline__ double neg_log_dens_elem(double q) {
    if (tid < VL) w.nat[tid] = (tid < dim) ? q : 0.0;
    __syncthreads();
    const TargetAux aux = target_prepare<false>(target, w.nat, dim, tparams, threadIdx.x & 63);
    const double e = (tid < dim) ? target_nld_elem<false>(target, aux, w.nat, tid, dim, tparams) : 0.0;
    __syncthreads();
    return e;
  }
};

template <class C, int RMETRIC, bool LOWRANK = false>
__device__ __forceinline__ void init_backend(BlockBackend<C, RMETRIC, LOWRANK>& bk, const ImplicitArgs& A,
                                             double* lds) {
  constexpr int PG = C::PG, TS = C::TS, PV = C::PV, VL = C::VL, NTILE = C::NTILE;
  const int tid = threadIdx.x;
  bk.dim = A.dim;
  bk.inv_dim_ = 1.0 / (double)A.dim;
  bk.tid = tid;
  bk.target = A.target;
  bk.tile = tid < NTILE;
  int ti = (int)((sqrtf(8.0f * (float)tid + 1.0f) - 1.0f) * 0.5f);
  while (ti * (ti + 1) / 2 > tid) --ti;
  while ((ti + 1) * (ti + 2) / 2 <= tid) ++ti;
  bk.ti = bk.tile ? ti : PG - 1;
  bk.tj = bk.tile ? tid - ti * (ti + 1) / 2 : PG - 1;
  bk.w.col0 = lds;
  bk.w.col1 = lds + PV;
  bk.w.vin = lds + 2 * PV;
  bk.w.part = lds + 3 * PV;
  bk.w.nat = bk.w.part + C::PART;
  bk.w.aux = bk.w.nat + VL;
  bk.w.red = bk.w.aux + VL;
  using BK = BlockBackend<C, RMETRIC, LOWRANK>;
  bk.w.stash = bk.w.red + 16;
  bk.w.trow = bk.w.stash + BK::kSlots * VL;
  double* nxt = bk.w.trow + (C::PARK ? TS * C::NT : 0);
  bk.w.rsd = nxt;
  if constexpr (BK::kRefine) nxt += VL;
  bk.w.xq = nxt;  // (built-in metrics only)
  bk.w.uq = nxt;  // (user metrics only)
  bk.w.uaq = bk.w.uq + VL;
  bk.w.ux = bk.w.nat;
  bk.w.uax = bk.w.aux;
  bk.refine_on = A.no_refine == 0;
  bk.lr_refresh_ = A.lowrank_refresh;
  bk.base = A.rparams;
  bk.tparams = A.tparams;
  bk.work = A.work ? A.work + (int64_t)blockIdx.x * (C::DP * C::DP) : nullptr;
}

// MIDPOINT: ImplicitMidpointIntegrator (integrators.py:547-681) on the same backend and slot storage
template <class C, int RMETRIC, bool MIDPOINT, bool LOWRANK = false>
__device__ __forceinline__ void implicit_team_body(const ImplicitArgs& A, double* lds) {
  const int64_t chain = blockIdx.x;
  BlockBackend<C, RMETRIC, LOWRANK> bk;
  init_backend<C, RMETRIC, LOWRANK>(bk, A, lds);
  const int dim = A.dim, tid = threadIdx.x;
  const bool act = tid < dim;
  double q = act ? A.pos[chain * dim + tid] : 0.0;
  double p = act ? A.mom[chain * dim + tid] : 0.0;
  const double t = signed_step(A.dir, A.step_scale, chain, A.step_size);
  bk.slot(SL_Q) = q;
  bk.slot(SL_P) = p;
  const int my_steps = mmdev::chain_steps(A.chain_steps, chain, A.n_steps);
  const ChainResult r = MIDPOINT ? implicit_midpoint_chain(bk, t, my_steps, A.opts)
                                 : implicit_leapfrog_chain(bk, t, my_steps, A.opts);
  q = bk.slot(SL_Q);
  p = bk.slot(SL_P);
  if (act) {
    A.pos[chain * dim + tid] = q;
    A.mom[chain * dim + tid] = p;
  }
  if (tid == 0) {
    A.status[chain] = r.status;
    A.n_done[chain] = r.done;
    add_counters(A.counters, r);
  }
}

template <class C, int RMETRIC, int OP>
__device__ __forceinline__ void riemann_aux_team_body(const ImplicitArgs& A, double* lds) {
  const int64_t chain = blockIdx.x;
  BlockBackend<C, RMETRIC> bk;
  init_backend<C, RMETRIC>(bk, A, lds);
  const int dim = A.dim, tid = threadIdx.x;
  const bool act = tid < dim;
  const double q = act ? A.pos[chain * dim + tid] : 0.0;
  const double p = act ? A.mom[chain * dim + tid] : 0.0;
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
  bool ok = bk.build(q);
  if constexpr (OP == 0) {
    double logdet;
    ok = bk.template invert<true>(&logdet) && ok;
    const double u = bk.matvec(p);
    const double e = bk.neg_log_dens_elem(q) + (act ? 0.5 * p * u : 0.0);
    const double h = block_reduce<C>(e, 0, bk.w.red) + 0.5 * logdet;
    if (tid == 0) A.out[chain] = ok ? h : nan;
  } else if constexpr (OP == 1) {
    ok = bk.template invert<false>(nullptr) && ok;
    const double u = bk.matvec(p);
    if (act) A.out[chain * dim + tid] = ok ? u : nan;
  } else {
    if (tid < C::VL) bk.w.aux[tid] = act ? A.z[chain * dim + tid] : 0.0;
    __syncthreads();
    double y;
    ok = bk.template sweep<false, true>(nullptr, &y) && ok;
    if (act) A.mom[chain * dim + tid] = ok ? y : nan;
  }
}

#ifndef MM_RTC_BUILD  // the in-tree instantiations (a run-time translation unit defines extern "C" wrappers instead)
template <class C, int RMETRIC, bool MIDPOINT, bool LOWRANK = false>
__global__ __launch_bounds__(C::NT, C::MINW) void implicit_team_kernel(ImplicitArgs A) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  implicit_team_body<C, RMETRIC, MIDPOINT, LOWRANK>(A, lds);
}
template <class C, int RMETRIC, int OP>
__global__ __launch_bounds__(C::NT, C::MINW) void riemann_aux_team_kernel(ImplicitArgs A) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  riemann_aux_team_body<C, RMETRIC, OP>(A, lds);
}
#endif

}  // namespace mmteam
