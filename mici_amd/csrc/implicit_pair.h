// Device code of the TWO-WAVES-PER-CHAIN matrix-core dense-Riemannian kernel (round 6; k_implicit_pair.hip instantiates it
// for the built-in metrics, mm_rtc.hip compiles it at run time around a USER metric).
//
// Implicit leapfrog on dense-metric Riemannian systems, 32 < D <= 64: the kernel of implicit_mfma.h with a chain owned
// by a WORKGROUP OF TWO WAVES instead of one wave.  gfx950 / CDNA4.
//
// Why: at BASELINE c3 (1024 chains, D = 64) the one-wave kernel has exactly one wave per SIMD,
// issues in 53 % of its cycles and keeps 128 registers of inverse row next to everything else in exactly 256 architected
// registers - a second instruction stream (the lock step of round 5) pushes the row into accumulation registers and
// loses.  Here a chain is ROW-SPLIT over two waves: wave w owns the flat elements / matrix rows [32 w, 32 w + 32); lane l
// of it holds columns [32 h, 32 h + 32), h = l >> 5, of row 32 w + (l & 31) of the held inverse (64 registers) AND of the
// base matrix of the rank-one metric (64 registers - it no longer lives in LDS, so a product streams nothing but the
// operand vector).  2048 waves = two per SIMD, from different chains, every register architected, and each wave issues
// about HALF the one-wave kernel's stream: nothing is computed twice (the first form tried - a column split with the flat
// vectors redundant on both waves - issued 85 % of it per wave and lost 9 %, profiles/r06_ab_c3_pair.txt).
//
// Flat vectors: element i = 32 w + (l & 31) on lanes l and l + 32 of wave w (the two halves of a wave carry the same
// value - same instruction, no extra issue).  Team collectives (implicit_core.h) meet through a double-buffered exchange
// slot in LDS behind ONE two-wave s_barrier each:
//   product   y = T v:  both waves publish their 32 operand elements, barrier, a lane contracts its 32 columns against
//             the broadcast operand, the two column halves of a row add up inside the wave (v_permlane32_swap);
//   sum/norm: half-wave DPP reduction, the two partial results exchanged, added in a fixed order (bit-identical on both).
// M(x) v of the rank-one metric sends its x . v partial along with the operand: one barrier.  All control flow of the
// step is team-uniform (it depends on collectives only), so both waves execute the same sequence of barriers.
// The blocked sweep (once per step) runs on wave 0 alone, as in implicit_mfma.h - the partner waits at the barrier and
// costs no issue slots - and hands the inverse over in row form sixteen columns at a time.
//
// Reference arithmetic replaced: as implicit_mfma.h (matrices.py:1161-1188 inside integrators.py:493-544, the products of
// systems.py:1381-1399); the step logic is implicit_core.h.
#pragma once
#include "implicit_mfma.h"

namespace mmpair {

using namespace mmdev;
using namespace mmimp;
using mmmfma::d2;
using mmmfma::d4;
using mmmfma::kRowPitch;
using mmmfma::kTiles;
using mmmfma::tix;

// LDS of a chain (doubles): wave 0's private block is the one-wave kernel's (the sweep's panels, the row-conversion
// buffer, its slots); wave 1 only needs what the flat arithmetic touches; then the shared part
constexpr int kW0Doubles = mmmfma::kMfmaWaveDoubles;
constexpr int kW1Doubles = 64 + 64 + 64 + SL_COUNT_REFINE * 64 + 16;  // qt (point), nat, aux, slots, (profile builds: clocks)
constexpr int kXSlot = 64 + 8;                                         // an exchange slot: operand vector, 2 x 4 scalars
constexpr int kSharedDoubles = 2 * kXSlot + 64 + 8;                    // two slots, factorised-solve result, flags
template <int RMETRIC>
__host__ __device__ constexpr int pair_chain_doubles() {
  return kW0Doubles + kW1Doubles + kSharedDoubles + (RMETRIC == MM_RMETRIC_USER ? 2 * mmuser::lds_doubles(64) : 0);
}

__device__ __forceinline__ double swap_sum32(double m) {  // m[l] + m[l ^ 32], the same bits on both lanes
  const long long b = __double_as_longlong(m);
  const unsigned lo = (unsigned)(b & 0xffffffffLL), hi = (unsigned)(b >> 32);
  const auto l2 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto h2 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  const double x0 = __longlong_as_double(((long long)h2[0] << 32) | (unsigned)l2[0]);
  const double x1 = __longlong_as_double(((long long)h2[1] << 32) | (unsigned)l2[1]);
  return x0 + x1;
}

template <int RMETRIC, bool PROFILE = false>
struct PairBackend : mmmfma::MfmaBackend<RMETRIC, PROFILE> {
  using Base = mmmfma::MfmaBackend<RMETRIC, PROFILE>;
  using Base::acc;
  using Base::dim;
  using Base::lane;
  using Base::w;
  using Base::uparams;
  using Base::inv_dim_;
  using Base::fd_;
  using Base::work;
  using Base::target;
  using Base::tparams;
  static constexpr bool kDual = false;
  int wave;        // 0: owns the sweep; 1: partner
  int idx;         // this lane's flat element / matrix row: 32 wave + (lane & 31)
  int half;        // this lane's column half: lane >> 5
  int xbuf;        // which exchange slot the next collective uses
  double* xs;      // [2][kXSlot]
  double* usol;    // [64] result of a factorised solve (wave 0 -> both)
  double* flag;    // [8]
  double* rowbuf;  // wave 0's row-conversion buffer (read by both)
  double fh_[32];  // columns [32 half, 32 half + 32) of row idx of M(x0)^-1
  double bh_[RMETRIC == MM_RMETRIC_RANK1 ? 32 : 1];  // the same columns of the base matrix' row (rank-one metric)
  double mh_[RMETRIC == MM_RMETRIC_USER ? 32 : 1];   // ... of the user's M(x) at the products' point

  __device__ __forceinline__ bool flat_active() const { return idx < dim; }
  __device__ __forceinline__ void pair_sync() { __syncthreads(); }

  // ---- collectives -----------------------------------------------------------------------------------------------------
  // sum over this wave's 32 elements (lanes 0 .. 31; the upper half carries copies), wave-uniform
  __device__ static __forceinline__ double local_sum(double v) {
    v = group8_sum(v);
    v += dpp_move<kDppMirror>(v);
    return readlane_f64(v, 0) + readlane_f64(v, 16);
  }
  __device__ __forceinline__ double* xslot() { return xs + xbuf * kXSlot; }
  __device__ __forceinline__ double sum1(double a) {
    const double loc = local_sum(idx < dim ? a : 0.0);
    double* b = xslot() + 64;
    if (lane == 0) b[wave * 4] = loc;
    pair_sync();
    const double t = b[0] + b[4];
    xbuf ^= 1;
    return t;
  }
  __device__ __forceinline__ void sum2(double a, double c, double* sa, double* sc) {
    const double la = local_sum(idx < dim ? a : 0.0), lc = local_sum(idx < dim ? c : 0.0);
    double* b = xslot() + 64;
    if (lane == 0) *reinterpret_cast<d2*>(b + wave * 4) = d2{la, lc};
    pair_sync();
    const d2 x0 = *reinterpret_cast<const d2*>(b), x1 = *reinterpret_cast<const d2*>(b + 4);
    xbuf ^= 1;
    *sa = x0[0] + x1[0];
    *sc = x0[1] + x1[1];
  }
  __device__ __forceinline__ double norm(double x, int kind) {
    const double a = wave_norm_accum(0.0, idx < dim ? x : 0.0, kind);
    const double loc = kind == MM_NORM_LINF ? wave_max(a) : local_sum(a);  // (the copies do not change a maximum)
    double* b = xslot() + 64;
    if (lane == 0) b[wave * 4] = loc;
    pair_sync();
    const double b0 = b[0], b1 = b[4];
    xbuf ^= 1;
    return kind == MM_NORM_LINF ? nanmax(b0, b1) : sqrt(b0 + b1);
  }
  // both waves' 32 elements of a flat vector, natural order (element i at [i], zero beyond dim), in this collective's
  // slot; valid until the collective after the next one overwrites it
  __device__ __forceinline__ const double* gather(double v) {
    double* b = xslot();
    if (lane < 32) b[idx] = (idx < dim) ? v : 0.0;
    pair_sync();
    xbuf ^= 1;
    return b;
  }

  static constexpr int kAcc = 4;
  // row idx of T times the operand vector at vb (natural order): this lane's 32 columns, then the two halves of the row
  __device__ __forceinline__ double row_dot(const double (&row)[32], const double* vb) {
    const double* src = vb + 32 * half;
    double y[kAcc];
#pragma unroll
    for (int a = 0; a < kAcc; ++a) y[a] = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const d4 vv = *reinterpret_cast<const d4*>(src + 4 * k);
#pragma unroll
      for (int e = 0; e < 4; ++e) y[(4 * k + e) % kAcc] = __builtin_fma(row[4 * k + e], vv[e], y[(4 * k + e) % kAcc]);
    }
#pragma unroll
    for (int h = kAcc / 2; h >= 1; h >>= 1)
#pragma unroll
      for (int a = 0; a < h; ++a) y[a] += y[a + h];
    return swap_sum32(y[0]);
  }

  // ---- y = M(x0)^-1 v ------------------------------------------------------------------------------------------------
  __device__ __forceinline__ double matvec(double v) {
    const double y = row_dot(fh_, gather(v));
    return idx < dim ? y : 0.0;
  }
  __device__ __forceinline__ double diag() { return idx < dim ? fd_ : 0.0; }

  // ---- refinement products -------------------------------------------------------------------------------------------
  __device__ __forceinline__ void metric_point(double x) {
    w.qt[lane] = (idx < dim) ? x : 0.0;
    if constexpr (RMETRIC == MM_RMETRIC_USER) {
      // the products' point in natural order and its aux block (each wave its own copy: the hooks are wave collectives),
      // then this lane's 32 columns of row idx of the user's metric_func there - once per refinement solve
      const double* b = gather(x);
      w.ux[lane] = b[lane];
      wave_sync();
      mmuser::prepare(mmuser::WaveTeam{lane}, w.ux, dim, uparams, w.uax);
      wave_sync();
      int oi = idx, oc = 32 * half;
      asm volatile("" : "+v"(oi), "+v"(oc));
#pragma unroll
      for (int k = 0; k < 32; ++k) mh_[k] = mmuser::entry_padded(w.ux, oi, oc + k, dim, uparams, w.uax);
    }
  }
  __device__ __forceinline__ double metric_apply(double v) {
    const double x = w.qt[lane];
    if constexpr (RMETRIC == MM_RMETRIC_DIAGQUAD) {
      return idx < dim ? __builtin_fma(x * x, v, v) : 0.0;
    } else if constexpr (RMETRIC == MM_RMETRIC_USER) {
      const double y = row_dot(mh_, gather(v));
      return idx < dim ? y : 0.0;
    } else {
      // rank-one update  B v + x (x . v) / D:  the partial of x . v travels with the operand
      const double loc = local_sum(idx < dim ? x * v : 0.0);
      double* b = xslot();
      if (lane < 32) b[idx] = (idx < dim) ? v : 0.0;
      if (lane == 0) b[64 + wave * 4] = loc;
      pair_sync();
      xbuf ^= 1;
      const double dot = b[64] + b[68];
      const double bv = row_dot(bh_, b);
      const double y = __builtin_fma(x, dot * inv_dim_, bv);
      return idx < dim ? y : 0.0;
    }
  }

  // this lane's 32 columns of row idx of the rank-one metric's base matrix, zero outside dim x dim.  Dead while a sweep
  // runs (construct() says so and reloads them: 32 L2 hits a lane, once per step) - live, their 64 registers next to wave
  // 0's tiles spill
  __device__ __forceinline__ void load_base() {
    if constexpr (RMETRIC == MM_RMETRIC_RANK1) {
      int oi = idx, oc = 32 * half;
      asm volatile("" : "+v"(oi), "+v"(oc));
      const int rc = oi < dim ? oi : dim - 1;
      const double* brow = uparams + (int64_t)rc * dim;
#pragma unroll
      for (int k = 0; k < 32; ++k) {
        const int col = oc + k, cc = col < dim ? col : dim - 1;
        const double b = brow[cc];
        bh_[k] = (oi < dim && col < dim) ? b : 0.0;
      }
    }
  }

  // ---- metric_func(x) into wave 0's tiles; x: the point in LANE order (element `lane`) ---------------------------------
  // (MfmaBackend::build with the rank-one metric's base matrix read from global memory: forty L2 hits a lane and step)
  __device__ __forceinline__ bool build_w0(double x) {
    if constexpr (RMETRIC != MM_RMETRIC_RANK1) {
      return Base::build(x);
    } else {
      // (the lane index laundered: the forty entry addresses derived from it are loop invariant, and hoisted out of the
      // step loop they lived in scratch)
      int ol = lane;
      asm volatile("" : "+v"(ol));
      const int g = ol >> 4, j = ol & 15;
      const double xm = (lane < dim) ? x : 0.0;
      w.nat[lane] = xm;
      w.vperm[(((lane >> 4) * 4 + (lane & 3)) << 2) + ((lane >> 2) & 3)] = xm;
      wave_sync();
      double qc[4];
      d4 qr[4];
#pragma unroll
      for (int X = 0; X < 4; ++X) {
        qc[X] = w.nat[16 * X + j];
        qr[X] = *reinterpret_cast<const d4*>(w.vperm + ((X * 4 + g) << 2));
      }
      const int dm1 = dim - 1;
#pragma unroll
      for (int I = 0; I < 4; ++I)
#pragma unroll
        for (int J = 0; J <= I; ++J) {
          const int t = tix(I, J);
          const int col = 16 * J + j, cc = col < dim ? col : dm1;
          const double qs = qc[J] * inv_dim_;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = 16 * I + 4 * r + g, rc = row < dim ? row : dm1;
            const double b = uparams[rc * dim + cc];
            acc[t][r] = __builtin_fma(qr[I][r], qs, (row < dim && col < dim) ? b : 0.0);
          }
        }
      double chk = 0.0;
#pragma unroll
      for (int I = 0; I < 4; ++I) {
        const int t = tix(I, I);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool on_diag = (j == 4 * r + g);
          if (on_diag && 16 * I + 4 * r + g >= dim) acc[t][r] = 1.0;  // identity on the padding
          chk = __builtin_fma(acc[t][r], 0.0, chk);  // "Array is not finite." (see MfmaBackend::build)
        }
      }
      wave_sync();
      return __all(chk == 0.0);
    }
  }

  // the inverse from wave 0's tiles to half rows on both waves, sixteen columns at a time through wave 0's buffer
  __device__ __forceinline__ void tiles_to_rows_pair() {
    const int g = lane >> 4, j = lane & 15;
    double* buf = rowbuf;  // [64][kRowPitch]
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (wave == 0) {
#pragma unroll
        for (int I = c; I < 4; ++I)
#pragma unroll
          for (int r = 0; r < 4; ++r) buf[(16 * I + 4 * r + g) * kRowPitch + j] = acc[tix(I, c)][r];
#pragma unroll
        for (int J = 0; J < c; ++J)
#pragma unroll
          for (int r = 0; r < 4; ++r) buf[(16 * J + j) * kRowPitch + 4 * r + g] = acc[tix(c, J)][r];
      }
      pair_sync();
      {
        const bool mine = half == (c >> 1);  // these sixteen columns belong to this lane's half
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const d2 x = *reinterpret_cast<const d2*>(buf + idx * kRowPitch + 2 * k);
          if (mine) {
            fh_[16 * (c & 1) + 2 * k] = x[0];
            fh_[16 * (c & 1) + 2 * k + 1] = x[1];
          }
        }
      }
      if ((idx >> 4) == c) fd_ = buf[idx * kRowPitch + (idx & 15)];
      pair_sync();
    }
  }

  // implicit_core.h, kUnifiedConstruct
  __device__ __forceinline__ bool construct(double x, bool need_inverse, double rhs, double* u) {
    // (every construction ends the life of the inverse held so far: see MfmaBackend::construct)
#pragma unroll
    for (int k = 0; k < 32; ++k) fh_[k] = 0.0;
    fd_ = 0.0;
    if constexpr (RMETRIC == MM_RMETRIC_USER) {
#pragma unroll
      for (int k = 0; k < 32; ++k) mh_[k] = 0.0;
    }
    if constexpr (RMETRIC == MM_RMETRIC_RANK1) {
#pragma unroll
      for (int k = 0; k < 32; ++k) bh_[k] = 0.0;
    }
    const double* xb = gather(x);
    const double xl = xb[lane];  // the point in lane order
    double rl = 0.0;
    if (!need_inverse) rl = gather(rhs)[lane];  // (team-uniform)
    if (wave == 0) {
      bool ok = build_w0(xl);
      if (need_inverse) {
        ok = this->template sweep<false>() && ok;
      } else {
        ok = this->template sweep<true>() && ok;
        usol[lane] = this->solve_factored(rl);
      }
      if (lane == 0) flag[0] = ok ? 1.0 : 0.0;
    } else {
      if constexpr (RMETRIC == MM_RMETRIC_USER) {  // the point of the held inverse for this wave's copy of the VJP hooks
        w.uq[lane] = xl;
        wave_sync();
        mmuser::prepare(mmuser::WaveTeam{lane}, w.uq, dim, uparams, w.uaq);
        wave_sync();
      }
    }
    if (need_inverse) {
      tiles_to_rows_pair();  // (its first barrier publishes the flag)
    } else {
      pair_sync();
      *u = (idx < dim) ? usol[idx] : 0.0;
    }
    const bool ok = flag[0] != 0.0;
    load_base();
    pair_sync();  // (the flag / usol are rewritten by the next construction)
    return ok;
  }
  __device__ __forceinline__ bool build_and_invert(double x) {
    double dummy;
    return construct(x, true, 0.0, &dummy);
  }
  __device__ __forceinline__ bool build_and_solve(double x, double rhs, double* u) { return construct(x, false, rhs, u); }

  template <bool OUTER>
  __device__ __forceinline__ double user_half_vjp(double u) {
    double r;
    if constexpr (mmuser::kFlatVjp) {
      if constexpr (OUTER) {
        mmuser::VjpOpsOuter<PairBackend> ops{*this, idx < dim ? u : 0.0};
        r = mmuser::vjp_flat(ops, w.uq, idx, dim, uparams, w.uaq);
      } else {
        mmuser::VjpOpsInv<PairBackend> ops{*this};
        r = mmuser::vjp_flat(ops, w.uq, idx, dim, uparams, w.uaq);
      }
    } else {
#if defined(MM_RTC_BUILD) && defined(MM_RTC_USER_METRIC)
      if constexpr (OUTER) {
        const double* ub = gather(u);
        w.aux[lane] = ub[lane];
        wave_sync();
        const MmMat vm{nullptr, w.aux, 0};
        r = (idx < dim) ? mmuser::vjp_dense(w.uq, vm, idx, dim, uparams, w.uaq) : 0.0;
        wave_sync();
      } else {
#pragma unroll
        for (int jj = 0; jj < 32; ++jj) work[idx * 64 + 32 * half + jj] = fh_[jj];  // this lane's half of row idx
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        pair_sync();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const MmMat vm{work, nullptr, 64};
        r = (idx < dim) ? mmuser::vjp_dense(w.uq, vm, idx, dim, uparams, w.uaq) : 0.0;
      }
#else
      r = 0.0;
#endif
    }
    return idx < dim ? 0.5 * r : 0.0;
  }

  __device__ __forceinline__ double half_vjp_inv(double q) {
    if constexpr (RMETRIC == MM_RMETRIC_USER) return user_half_vjp<false>(0.0);
    else if constexpr (RMETRIC == MM_RMETRIC_RANK1) return matvec(q) * inv_dim_;
    else return q * diag();
  }
  __device__ __forceinline__ double dh2_dpos(double p, double q) {
    const double u = matvec(p);
    if constexpr (RMETRIC == MM_RMETRIC_USER) {
      return user_half_vjp<true>(u);
    } else if constexpr (RMETRIC == MM_RMETRIC_RANK1) {
      const double uq = sum1(u * q);
      return -(u * uq) * inv_dim_;
    } else {
      return -q * (u * u);
    }
  }
  // grad_neg_log_dens: the targets are wave collectives over a position in lane order - each wave evaluates the whole
  // gradient (once per step) and keeps its own elements
  __device__ __forceinline__ double grad(double q) {
    const double* qb = gather(q);
    w.nat[lane] = qb[lane];
    wave_sync();
    const TargetAux aux = target_prepare<false>(target, w.nat, dim, tparams, lane);
    const double gr = (lane < dim) ? target_grad_elem<false>(target, aux, w.nat, lane, dim, tparams) : 0.0;
    w.aux[lane] = gr;
    wave_sync();
    const double mine = (idx < dim) ? w.aux[idx] : 0.0;
    wave_sync();
    return mine;
  }
};

template <int RMETRIC, bool PROFILE = false>
__device__ __forceinline__ void implicit_pair_body(const ImplicitArgs& A, double* lds) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int dim = A.dim;
  const int64_t chain = blockIdx.x;
  if (chain >= A.n_chains) return;
  const int idx = 32 * wave + (lane & 31);
  const bool act = idx < dim;
  double q = act ? A.pos[chain * dim + idx] : 0.0;
  double p = act ? A.mom[chain * dim + idx] : 0.0;
  const double t = signed_step(A.dir, A.step_scale, chain, A.step_size);

  PairBackend<RMETRIC, PROFILE> bk;
  bk.dim = dim;
  bk.inv_dim_ = 1.0 / (double)dim;
  bk.lane = lane;
  bk.wave = wave;
  bk.idx = idx;
  bk.half = lane >> 5;
  bk.xbuf = 0;
  bk.target = A.target;
  double* w0 = lds;
  double* w1 = lds + kW0Doubles;
  double* sh = w1 + kW1Doubles;
  bk.rowbuf = w0 + 704;
  if (wave == 0) {
    bk.w.qt = w0;
    bk.w.wt = w0 + 256;
    bk.w.nat = w0 + 512;
    bk.w.vperm = w0 + 576;
    bk.w.aux = w0 + 640;
    bk.w.part = w0 + 704;
    bk.w.mpart = bk.w.part + 64 * kRowPitch;
    bk.w.stash = bk.w.mpart + 192;
    bk.w.prof = bk.w.stash + SL_COUNT_REFINE * 64;
  } else {
    bk.w.qt = w1;
    bk.w.nat = w1 + 64;
    bk.w.aux = w1 + 128;
    bk.w.stash = w1 + 192;
    bk.w.wt = nullptr;  // (the sweep's buffers: wave 0 only)
    bk.w.vperm = nullptr;
    bk.w.part = nullptr;
    bk.w.mpart = nullptr;
    bk.w.prof = bk.w.stash + SL_COUNT_REFINE * 64;  // (profile builds: only wave 0's clocks are reported)
  }
  bk.xs = sh;
  bk.usol = sh + 2 * kXSlot;
  bk.flag = sh + 2 * kXSlot + 64;
  bk.refine_on = A.no_refine == 0;
  bk.dual_off = true;
  bk.base_lds = nullptr;
  bk.tparams = A.tparams;
  bk.uparams = A.rparams;
  bk.work = nullptr;
  if constexpr (RMETRIC == MM_RMETRIC_USER) {
    constexpr int kA = (mmuser::kAux + 1) & ~1;
    double* up = sh + kSharedDoubles + wave * mmuser::lds_doubles(64);
    bk.w.uq = up;
    bk.w.ux = up + 64;
    bk.w.uaq = up + 128;
    bk.w.uax = up + 128 + kA;
    bk.work = A.work ? A.work + chain * (int64_t)(64 * 64) : nullptr;
  }
  bk.load_base();
  if constexpr (PROFILE) {
    if (lane < PH_COUNT + 2) bk.w.prof[lane] = lane == PH_COUNT + 1 ? (double)__builtin_readcyclecounter() : 0.0;
    wave_sync();
  }
  bk.slot(SL_Q) = q;
  bk.slot(SL_P) = p;
  const ChainResult r = implicit_leapfrog_chain(bk, t, mmdev::chain_steps(A.chain_steps, chain, A.n_steps), A.opts);
  q = bk.slot(SL_Q);
  p = bk.slot(SL_P);
  if (act && lane < 32) {
    A.pos[chain * dim + idx] = q;
    A.mom[chain * dim + idx] = p;
  }
  if (wave == 0) {
    if (lane == 0) {
      A.status[chain] = r.status;
      A.n_done[chain] = r.done;
      add_counters(A.counters, r);
    }
    if constexpr (PROFILE) {
      bk.prof_switch(PH_OTHER);
      wave_sync();
      if (lane < PH_COUNT) A.out[chain * PH_COUNT + lane] = bk.w.prof[lane];
    }
  }
}

#ifndef MM_RTC_BUILD
template <int RMETRIC, bool PROFILE = false>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2))) void implicit_pair_kernel(ImplicitArgs A) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  implicit_pair_body<RMETRIC, PROFILE>(A, lds);
}
#endif

}  // namespace mmpair
