// Device code of the GLOBAL-MEMORY tier of the dense-Riemannian kernels: 279 < D <= 1024 (round 5).
//
// Every other backend keeps a chain's D x D metric on chip - in the registers of a wave (D <= 64), of a CU (D <= 279) - and
// that is where the size limit of rounds 1-4 came from.  The reference factorises any D (DensePositiveDefiniteMatrix,
// matrices.py:1117-1216; DenseRiemannianMetricSystem, systems.py:1690-1734).  Here the matrix lives in HBM: one
// 1024-thread workgroup per chain (one flat vector element per thread, as implicit_core.h wants it), a DP x DP row-major
// workspace per chain (DP = D rounded up to 64), and
//   * the explicit inverse by a BLOCKED symmetric sweep, sixteen (D <= 512) or eight pivots per pass over the matrix (the algebra of
//     implicit_mfma.h's block step, any block size: panel Q = A[K, :], X = Q - E, W = P^-1 X, A -= W^T X, A_KK -= 2 I; after
//     the last block A = -M^-1 - negated in one more pass - the pivots of the in-block eliminations are the Cholesky pivots
//     squared: positive definiteness and log det) - D / NB passes of 2 x 8 DP^2 bytes of HBM traffic;
//   * every product (M^-1 v of the held inverse, M(x) v of the refinement solves - the solve-only constructions are
//     refined from the held inverse as on every other backend, implicit_core.h refine_solve) as a COLUMN walk: thread i
//     accumulates sum_j A[j][i] v_j, the loads of a wave are 512 consecutive bytes for every j, v_j is an LDS broadcast -
//     one pass over the matrix per product (the rank-one metric's base matrix is shared by all chains: L2 / MALL resident);
//   * sample_momentum's Cholesky factor (the reference's metric.sqrt @ z, matrices.py:1161-1178) by a blocked
//     right-looking factorisation on the same panel machinery, stored transposed so that L z is a column walk too.
// A step costs one sweep and ~60 products, all HBM-bound: this tier is about REACH (any D the reference takes, up to the
// 1024 threads of a workgroup), not about the roofline; DESIGN.md section 4.4b has the measured rates.
// Built-in metrics (rank-one update, diag(1 + q^2)) and - compiled at run time around the user's source, mm_rtc.hip - user
// metrics; the leapfrog and implicit-midpoint steps and the three auxiliary operations.
#pragma once
#include "implicit_core.h"
#include "user_metric.h"

namespace mmglob {

using namespace mmdev;
using namespace mmimp;

typedef double d4s __attribute__((ext_vector_type(4)));  // (sym_walk: a lane's four adjacent entries of a tile row)

constexpr int NT = 1024;      // threads per chain = largest D
constexpr int DPMAX = 1024;
// Pivots per block of the sweep: the panel X (NB x DP doubles) must fit a CU's LDS - 32 pivots up to DP = 512, 16 beyond
// (round 6: the second panel W = P^-1 X is no longer stored - a wave forms the W operands of its tile row on the matrix
// cores from X and the inverted pivot block - so the one panel that is left can hold twice the pivots: half the passes over
// the matrix).  LDS (doubles): one natural-order vector, the pivot block and its inverse, flags, two sets of reduction
// partials, then the panel with a pitch of DP.
constexpr int kPanelDoubles = 16 * DPMAX;      // = 32 * 512
constexpr int kOffNat = 0;                     // [DPMAX + 8]
constexpr int kOffPb = kOffNat + DPMAX + 8;    // [32 * 32] pivot block, inverted in place
constexpr int kOffFlag = kOffPb + 1024;        // [8] flags / log det of the block
constexpr int kOffRed = kOffFlag + 8;          // [2][2][16]  (two sets, two values a reduction)
constexpr int kOffRC = kOffRed + 64;           // [2][2][32] pivot row / column of the pivot-block inversion, double-buffered
constexpr int kOffX = kOffRC + 128;            // [NB][pitch]  X = Q - E
constexpr int kLdsDoubles = kOffX + kPanelDoubles;
static_assert(kLdsDoubles * 8 <= 160 * 1024, "LDS budget of a CU");
// a user metric (user_metric.h) adds the point of the HELD inverse in natural order and its aux block (they outlive the
// sweep: the vector-Jacobian products read them); the refinement products' point and aux block sit in the X panel, which is
// idle whenever a product runs
constexpr int kUserAux = (mmuser::kAux + 1) & ~1;
constexpr int kOffUq = kLdsDoubles;            // [DPMAX]
constexpr int kOffUaq = kOffUq + DPMAX;        // [kUserAux]
constexpr int kUserLdsDoubles = kOffUaq + kUserAux;
constexpr int kOffUx = kOffX;                  // [DPMAX]   (aliases the X panel)
constexpr int kOffUax = kOffX + DPMAX;         // [kUserAux]
static_assert(kUserLdsDoubles * 8 <= 160 * 1024, "LDS budget of a CU (user metric)");
static_assert(DPMAX + kUserAux <= kPanelDoubles / 2, "the products' point and aux block must fit the X panel");
template <int RMETRIC>
__host__ __device__ constexpr int global_lds_doubles() { return RMETRIC == MM_RMETRIC_USER ? kUserLdsDoubles : kLdsDoubles; }

__host__ __device__ constexpr int padded_dim(int dim) { return (dim + 63) & ~63; }

template <class BK>
struct TeamOfGlobal {  // the workgroup as a team (user_metric.h mm_user_prepare)
  BK& bk;
  __device__ __forceinline__ int rank() const { return bk.tid; }
  __device__ __forceinline__ int size() const { return NT; }
  __device__ __forceinline__ double sum(double x) const { return bk.reduce(x, false); }
};

template <int RMETRIC, int NB>
struct GlobalBackend {
  static constexpr int PITCH = kPanelDoubles / NB;  // panel row pitch: 1024 (NB = 16), 512 (NB = 32)
  static constexpr int kOffPart = kOffX + kPanelDoubles / 2;  // [2][NT] partial sums of the products (the panel is idle then)
  static constexpr int kOffNat2 = kOffPart + 2 * NT;          // [DPMAX] the second operand vector of a lock-step pair
  static constexpr int CB = NB > 16 ? 16 : NB;               // pivots per block of the Cholesky factorisation (sample_momentum)
  static constexpr bool kSolveByInverse = true;   // implicit_core.h: a factorised solve = invert + product
  static constexpr bool kUnifiedConstruct = false;
  static constexpr bool kCountersInLds = false;
  static constexpr bool kRefine = true;           // solve-only constructions refined from the held inverse
  bool refine_on;
  // implicit_core.h refine_solve2 (round 6): the reversibility-check solve and the C-adjoint solve of a step in lock step - here
  // a product IS a pass over 1 - 2 MB of HBM (or L2, for the shared base matrix), and two right-hand sides share it.  Built-in
  // metrics (a user metric's M(x) v evaluates its entries per product and point: nothing to share).
  static constexpr bool kDual = RMETRIC != MM_RMETRIC_USER;
  bool dual_off;                                  // MICI_AMD_DUAL=0: one solve after the other
  // implicit_core.h lowrank_solve (round 6): the rank-one-update metric's solve-only constructions by the Woodbury identity
  // from the held inverse - ONE pass over the FP64 inverse (F d) each instead of ~2.5 CG pairs; decided at run time here
  // (MICI_AMD_LOWRANK=0: the lock-step CG refinement)
  // (the built-in rank-one-update metric, or a user metric that declares the structure: user_metric.h MM_USER_LOWRANK)
  static constexpr bool kLowRankBuiltin = RMETRIC == MM_RMETRIC_RANK1;
  static constexpr bool kLowRank = kLowRankBuiltin || (RMETRIC == MM_RMETRIC_USER && mmuser::kLowRank);
  bool lowrank_on_;
  // (decided at compile time - the CG refinement's code dead - the kernel's allocation hardly changes: 800 B of scratch a lane
  // against 864)
  __device__ __forceinline__ bool lowrank_on() const { return lowrank_on_; }
  __device__ __forceinline__ double lowrank_scale() const {
    if constexpr (kLowRankBuiltin) return (double)dim;
    else return mmuser::lowrank_inv_s(dim, base);
  }
  // u(x), this thread's element (user metric: the point published for the hook, its aux block prepared - a team collective)
  __device__ __forceinline__ double lowrank_vec(double x) {
    if constexpr (kLowRankBuiltin) {
      return x;
    } else {
      metric_point(x);
      const double u = mmuser::lowrank_u(lds + kOffUx, tid, dim, base, lds + kOffUax);
      __syncthreads();  // (the next point overwrites kOffUx / kOffUax)
      return tid < dim ? u : 0.0;
    }
  }
  __device__ __forceinline__ double& lowrank_u0() {
    if constexpr (kLowRankBuiltin) return st_[SL_Q];
    else return rs_[LR_U0];
  }
  // a user metric's hooks evaluate its vector-Jacobian products at "the point of the held inverse" (kOffUq, kOffUaq - build()
  // sets them): an inverse carried to x by lowrank_update takes the point with it
  __device__ __forceinline__ void held_point(double x) {
    if constexpr (RMETRIC == MM_RMETRIC_USER) {
      __syncthreads();
      lds[kOffUq + tid] = tid < dim ? x : 0.0;
      __syncthreads();
      mmuser::prepare(TeamOfGlobal<GlobalBackend>{*this}, lds + kOffUq, dim, base, lds + kOffUaq);
      __syncthreads();
    }
  }
  int lr_refresh_;
  __device__ __forceinline__ int lowrank_refresh() const { return lr_refresh_; }
  int dim, dp, tid, target, flip;
  double inv_dim_;
  double* lds;
  double* A;            // this chain's DP x DP workspace (row-major, leading dimension dp)
  // Round 6: the held inverse a second time in FP32, scaled by a power of two so that its largest (a diagonal) entry is ~1:
  // the PRECONDITIONER of the refinement solves (implicit_core.h refine_solve: z = F r) - half the bytes of the pass that a
  // CG pair spends most of its time in.  CG converges to the same 1e-14 with any symmetric positive-definite F; everything
  // that needs M(x0)^-1 itself (the momentum solves, the A / C sub-steps, the vector-Jacobian products) reads the FP64 matrix.
  float* Af;
  double pscale_;       // F = pscale_ * Af
  const double* base;   // rank-one metric: base matrix [dim][dim]; user metric: its params
  const double* tparams;
  double st_[SL_COUNT_REFINE];  // the step's flat per-thread state: registers (every index is a compile-time constant)
  double rs_[2 * RS_COUNT];     // (the second system of a lock-step pair in [RS_COUNT ..])
  double xpt_, xpt2_;           // this thread's coordinate of the refinement products' point(s)
  __device__ __forceinline__ double& slot(int i) { return st_[i]; }
  __device__ __forceinline__ double& rslot(int i) { return rs_[i]; }
  __device__ __forceinline__ bool flat_active() const { return tid < dim; }

  // ---- workgroup reductions: one barrier each (two sets of partials used alternately, as softabs.h block_reduce4) ------
  __device__ __forceinline__ double reduce(double v, bool use_max) {
    const int lane = tid & 63, wave = tid >> 6;
    double* const set = lds + kOffRed + 32 * flip;
    flip ^= 1;
    v = use_max ? wave_max(v) : wave_sum(v);
    if (lane == 0) set[wave] = v;
    __syncthreads();
    double r = set[0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) r = use_max ? nanmax(r, set[w]) : r + set[w];
    return r;
  }
  // two values through ONE barrier
  __device__ __forceinline__ void reduce2(double a, double b, bool use_max, double* ra, double* rb) {
    const int lane = tid & 63, wave = tid >> 6;
    double* const set = lds + kOffRed + 32 * flip;
    flip ^= 1;
    a = use_max ? wave_max(a) : wave_sum(a);
    b = use_max ? wave_max(b) : wave_sum(b);
    if (lane == 0) {
      set[wave] = a;
      set[16 + wave] = b;
    }
    __syncthreads();
    double x = set[0], y = set[16];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) {
      x = use_max ? nanmax(x, set[w]) : x + set[w];
      y = use_max ? nanmax(y, set[16 + w]) : y + set[16 + w];
    }
    *ra = x;
    *rb = y;
  }
  __device__ __forceinline__ double sum1(double a) { return reduce(tid < dim ? a : 0.0, false); }
  __device__ __forceinline__ void sum2(double a, double b, double* sa, double* sb) {
    reduce2(tid < dim ? a : 0.0, tid < dim ? b : 0.0, false, sa, sb);
  }
  __device__ __forceinline__ void sum2x(double a, double b, double* sa, double* sb) { sum2(a, b, sa, sb); }
  __device__ __forceinline__ void sum4(double a, double b, double c, double d, double* sa, double* sb, double* sc, double* sd) {
    sum2(a, b, sa, sb);
    sum2(c, d, sc, sd);
  }
  __device__ __forceinline__ void sum3(double a, double b, double c, double* sa, double* sb, double* sc) {
    sum2(a, b, sa, sb);
    *sc = sum1(c);
  }
  __device__ __forceinline__ void norm2(double a, double b, int kind, double* na, double* nb) {
    const double xa = tid < dim ? a : 0.0, xb = tid < dim ? b : 0.0;
    if (kind == MM_NORM_LINF) {
      reduce2(fabs(xa), fabs(xb), true, na, nb);
    } else {
      double sa, sb;
      reduce2(xa * xa, xb * xb, false, &sa, &sb);
      *na = sqrt(sa);
      *nb = sqrt(sb);
    }
  }
  __device__ __forceinline__ double norm(double x, int kind) {
    const double a = tid < dim ? x : 0.0;
    if (kind == MM_NORM_LINF) return reduce(fabs(a), true);
    return sqrt(reduce(a * a, false));
  }

  // natural-order copy of a flat vector (zero beyond dim), visible to the workgroup
  __device__ __forceinline__ void publish(double v) {
    __syncthreads();  // (readers of the previous contents are done)
    lds[kOffNat + tid] = tid < dim ? v : 0.0;
    __syncthreads();
  }

  __device__ __forceinline__ void publish2(double v0, double v1) {
    __syncthreads();
    lds[kOffNat + tid] = tid < dim ? v0 : 0.0;
    lds[kOffNat2 + tid] = tid < dim ? v1 : 0.0;
    __syncthreads();
  }
  // two products in one pass over the matrix: every load feeds both accumulators (implicit_core.h refine_solve2)
  template <class T>
  __device__ __forceinline__ void column_walk2(const T* __restrict__ mat, int ld, int n, double* r0, double* r1) {
    const bool two = 2 * dim <= NT;
    const int h = (two && tid >= NT / 2) ? 1 : 0;
    const int c = tid - h * (NT / 2);
    const int nh = two ? ((n + 1) >> 1) : n;
    const int j0 = h * nh, j1 = (j0 + nh < n) ? j0 + nh : n;
    constexpr int W2 = kWalk / 2;
    double y[W2], z[W2];
#pragma unroll
    for (int e = 0; e < W2; ++e) y[e] = z[e] = 0.0;
    if (c < dim) {
      const T* col = mat + c;
      const double* nat = lds + kOffNat;
      const double* nat2 = lds + kOffNat2;
      int j = j0;
      for (; j + kWalk <= j1; j += kWalk) {
        T a[kWalk];
#pragma unroll
        for (int e = 0; e < kWalk; ++e) a[e] = col[(size_t)(j + e) * ld];
#pragma unroll
        for (int e = 0; e < kWalk; ++e) {
          y[e % W2] = __builtin_fma((double)a[e], nat[j + e], y[e % W2]);
          z[e % W2] = __builtin_fma((double)a[e], nat2[j + e], z[e % W2]);
        }
      }
      for (; j < j1; ++j) {
        const double a = (double)col[(size_t)j * ld];
        y[0] = __builtin_fma(a, nat[j], y[0]);
        z[0] = __builtin_fma(a, nat2[j], z[0]);
      }
    }
#pragma unroll
    for (int hh = W2 / 2; hh >= 1; hh >>= 1)
#pragma unroll
      for (int e = 0; e < hh; ++e) {
        y[e] += y[e + hh];
        z[e] += z[e + hh];
      }
    if (!two) {
      *r0 = c < dim ? y[0] : 0.0;
      *r1 = c < dim ? z[0] : 0.0;
      return;
    }
    double* part = lds + kOffPart;  // [2][NT]
    part[tid] = y[0];
    part[NT + tid] = z[0];
    __syncthreads();
    *r0 = tid < dim ? part[tid] + part[NT / 2 + tid] : 0.0;
    *r1 = tid < dim ? part[NT + tid] + part[NT + NT / 2 + tid] : 0.0;
    __syncthreads();
  }

  // y_i = sum_{j < n} Mat[j * ld + i] nat_j : a column walk (coalesced over the threads for every j, nat_j broadcast).
  // Round 6: a product is a chain of dependent HBM round trips per thread - with four loads in flight on 512 of the 1024
  // threads a 2 MB pass took 180 us (11 GB/s a CU).  Now kWalk = 8 loads in flight a thread, and while the workgroup has two
  // threads a column (2 D <= 1024) the rows are split between them, the two partial sums meeting in the idle W panel.
  static constexpr int kWalk = 8;
  template <class T>
  __device__ __forceinline__ double column_walk(const T* __restrict__ mat, int ld, int n) {
    const bool two = 2 * dim <= NT;  // (team-uniform)
    const int h = (two && tid >= NT / 2) ? 1 : 0;
    const int c = tid - h * (NT / 2);
    const int nh = two ? ((n + 1) >> 1) : n;
    const int j0 = h * nh, j1 = (j0 + nh < n) ? j0 + nh : n;
    double y[kWalk];
#pragma unroll
    for (int e = 0; e < kWalk; ++e) y[e] = 0.0;
    if (c < dim) {
      const T* col = mat + c;
      const double* nat = lds + kOffNat;
      int j = j0;
      for (; j + kWalk <= j1; j += kWalk) {
        T a[kWalk];
#pragma unroll
        for (int e = 0; e < kWalk; ++e) a[e] = col[(size_t)(j + e) * ld];
#pragma unroll
        for (int e = 0; e < kWalk; ++e) y[e] = __builtin_fma((double)a[e], nat[j + e], y[e]);
      }
      for (; j < j1; ++j) y[0] = __builtin_fma((double)col[(size_t)j * ld], nat[j], y[0]);
    }
#pragma unroll
    for (int hh = kWalk / 2; hh >= 1; hh >>= 1)
#pragma unroll
      for (int e = 0; e < hh; ++e) y[e] += y[e + hh];
    if (!two) return c < dim ? y[0] : 0.0;
    double* part = lds + kOffPart;  // [2][NT / 2]  (no sweep is in flight while a product runs)
    part[tid] = y[0];
    __syncthreads();
    const double r = tid < dim ? part[tid] + part[NT / 2 + tid] : 0.0;
    __syncthreads();  // (the next product rewrites the partials)
    return r;
  }

  // ---- y = A v from the LOWER tiles of the symmetric held inverse (round 6) -----------------------------------------------
  // The Woodbury path (implicit_core.h lowrank_solve / lowrank_update) makes a step ~24 products with the held inverse and
  // one sweep per launch: the products ARE the tier's HBM traffic, and a column walk reads both triangles of a symmetric
  // matrix.  Here every 16 x 16 tile on or below the diagonal is read ONCE and used twice - directly (row sums, accumulated
  // in-lane along a tile row) and mirrored (its columns' sums: a transposing butterfly over the sixteen lanes of a DPP row,
  // added into the wave's own partial vector) - half the bytes of the pass.  Wave w owns the tile rows w and nt - 1 - w
  // (nt + 1 tiles for every wave at nt = 32), four adjacent tiles - 512 contiguous bytes of each of the sixteen rows - per
  // trip; the sixteen waves' partial vectors ([16][dp] doubles in the idle panel) are summed in a fixed order at the end:
  // the result does not depend on timing.  The partial vectors fill the panel at the tier's largest padded size (1024).
  bool sym_on_;
  __device__ __forceinline__ bool sym_ok() const { return sym_on_; }  // ([16][dp] partial vectors = the panel at dp = 1024)
  __device__ static __forceinline__ double row_reduce16(const d4s rs, const int j) {
    const bool h8 = (j & 8) != 0, h4 = (j & 4) != 0;
    double k0v = h8 ? rs[2] : rs[0], k1v = h8 ? rs[3] : rs[1];
    const double s0v = h8 ? rs[0] : rs[2], s1v = h8 ? rs[1] : rs[3];
    k0v += dpp_move<kDppMirror>(s0v);
    k1v += dpp_move<kDppMirror>(s1v);
    double kk = h4 ? k1v : k0v;
    const double ss = h4 ? k0v : k1v;
    kk += dpp_move<kDppHalfMirror>(ss);
    kk += dpp_move<kDppXor2>(kk);
    kk += dpp_move<kDppXor1>(kk);
    return kk;
  }
  __device__ __forceinline__ double sym_walk(const double* __restrict__ mat) {  // the operand is published (kOffNat)
    double* buf = lds + kOffX;  // [16][dp]
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int e = tid; e < 16 * dp; e += NT) buf[e] = 0.0;
    __syncthreads();
    const int nt = (dim + 15) >> 4;
    const int row = lane & 15, cg = lane >> 4;
    const double* nat = lds + kOffNat;
    double* mine = buf + wave * dp;
#pragma unroll 1
    for (int p = wave; 2 * p < nt; p += 16) {
#pragma unroll 1
      for (int half = 0; half < 2; ++half) {
        const int I = half == 0 ? p : nt - 1 - p;
        if (half == 1 && I == p) break;  // (the middle tile row of an odd count)
        const double vI = nat[16 * I + row];
        const double* rowp = mat + (size_t)(16 * I + row) * dp + 4 * cg;
        double accd = 0.0;
#pragma unroll 1
        for (int J = 0; J <= I; J += 4) {
          d4s a[4];
#pragma unroll
          for (int u = 0; u < 4; ++u)
            a[u] = (J + u <= I) ? *reinterpret_cast<const d4s*>(rowp + 16 * (J + u)) : d4s{0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            if (J + u > I) break;  // (wave-uniform)
            const d4s vJ = *reinterpret_cast<const d4s*>(nat + 16 * (J + u) + 4 * cg);
#pragma unroll
            for (int k = 0; k < 4; ++k) accd = __builtin_fma(a[u][k], vJ[k], accd);
            if (J + u < I) {  // below the diagonal: the mirrored tile's rows are this tile's columns
              d4s m;
#pragma unroll
              for (int k = 0; k < 4; ++k) m[k] = a[u][k] * vI;
              const double val = row_reduce16(m, row);  // lane `row`: the sum of column 4 cg + (row >> 2) over the sixteen rows
              if ((row & 3) == 0) mine[16 * (J + u) + 4 * cg + (row >> 2)] += val;
            }
          }
        }
        accd += __shfl_xor(accd, 16);
        accd += __shfl_xor(accd, 32);
        if (cg == 0) mine[16 * I + row] += accd;
      }
    }
    __syncthreads();
    double y = 0.0;
    if (tid < dp) {
#pragma unroll
      for (int w = 0; w < 16; ++w) y += buf[w * dp + tid];
    }
    __syncthreads();  // (the next product zeroes the partial vectors)
    return tid < dim ? y : 0.0;
  }

  // ---- metric_func(x) into the workspace (identity on the padding); false: an entry is not finite -------------------
  __device__ __forceinline__ bool build(double x) {
    publish(x);
    const double* nat = lds + kOffNat;
    if constexpr (RMETRIC == MM_RMETRIC_USER) {  // the point in natural order for the user's hooks, then its aux block
      lds[kOffUq + tid] = nat[tid];
      __syncthreads();
      mmuser::prepare(TeamOfGlobal<GlobalBackend>{*this}, lds + kOffUq, dim, base, lds + kOffUaq);
      __syncthreads();
    }
    const int tx = tid & 31, ty = tid >> 5;
    double chk = 0.0;
    for (int i = ty; i < dp; i += 32) {
      const double xi = nat[i] * inv_dim_;
      for (int j = tx; j < dp; j += 32) {
        double v = (i == j) ? 1.0 : 0.0;
        if (i < dim && j < dim) {
          if constexpr (RMETRIC == MM_RMETRIC_RANK1) v = __builtin_fma(xi, nat[j], base[(size_t)i * dim + j]);
          else if constexpr (RMETRIC == MM_RMETRIC_USER) v = mmuser::entry(lds + kOffUq, i, j, dim, base, lds + kOffUaq);
          else v = (i == j) ? __builtin_fma(nat[i], nat[i], 1.0) : 0.0;
        }
        A[(size_t)i * dp + j] = v;
        chk = __builtin_fma(v, 0.0, chk);  // "Array is not finite." (matrices.py:211-215): every entry is looked at
      }
    }
    const double bad = reduce(chk != 0.0 ? 1.0 : 0.0, false);  // (chk is NaN for a non-finite entry; its barrier publishes A)
    return bad == 0.0;
  }

  // A[i][j] -= sum_k Wp[k][i] Xp[k][j] over the whole matrix: thread (tx, ty) owns the elements (ty + 32 a, tx + 32 b),
  // 2 x 2 of them at a time (the panels' sixteen + sixteen operands of a 2 x 2 tile come from LDS once)
  template <int KB>
  __device__ __forceinline__ void rank_update(const double* __restrict__ Wp, const double* __restrict__ Xp) {
    const int tx = tid & 31, ty = tid >> 5;
    for (int a = 0; a < dp / 32; a += 2) {
      const int i0 = ty + 32 * a, i1 = i0 + 32;
      double w0[KB], w1[KB];
#pragma unroll
      for (int k = 0; k < KB; ++k) {
        w0[k] = Wp[k * PITCH + i0];
        w1[k] = Wp[k * PITCH + i1];
      }
      for (int b = 0; b < dp / 32; b += 2) {
        const int j0 = tx + 32 * b, j1 = j0 + 32;
        double* p00 = A + (size_t)i0 * dp + j0;
        double* p10 = A + (size_t)i1 * dp + j0;
        double a00 = p00[0], a01 = p00[32], a10 = p10[0], a11 = p10[32];
#pragma unroll
        for (int k = 0; k < KB; ++k) {
          const double x0 = Xp[k * PITCH + j0], x1 = Xp[k * PITCH + j1];
          a00 = __builtin_fma(-w0[k], x0, a00);
          a01 = __builtin_fma(-w0[k], x1, a01);
          a10 = __builtin_fma(-w1[k], x0, a10);
          a11 = __builtin_fma(-w1[k], x1, a11);
        }
        p00[0] = a00;
        p00[32] = a01;
        p10[0] = a10;
        p10[32] = a11;
      }
    }
  }

  // ---- the sweep's block update on the MATRIX CORES, lower triangle only (round 6) ---------------------------------------------
  // A is symmetric and the sweep keeps it so: only the 16 x 16 tiles (I, J) with I >= J are read and written - half the HBM
  // traffic of a pass - as  tile += (-W)[:, tile I]^T X[:, tile J],  NB / 4 v_mfma_f64_16x16x4_f64 a tile.  Operand layouts of
  // the instruction: A operand lane (i = l % 16, k = l / 16); B operand lane (k = l / 16, j = l % 16); accumulator lane
  // 16 g + j, register r <-> tile entry (4 r + g, j), i.e. four 128-byte row segments a global load.
  // A wave owns tile rows I and nt - 1 - I (nt + 1 tiles: every wave the same work).  For a tile row it first forms its W
  // operands ITSELF:  W[:, tile I] = P^-1 X[:, tile I]  is NB / 16 accumulator tiles of (NB / 4) instructions (A operand: the
  // symmetric P^-1 from LDS, B operand: X from LDS) - and register r of accumulator tile t, negated, IS the A operand of
  // k-step 4 t + r of the update (entry (4 r + g, j) of W's tile = W[16 t + 4 r + g][16 I + j] on lane 16 g + j: the layout the
  // instruction wants for W^T).  No W panel in LDS, no pass of the workgroup to compute it, no barrier for it.
  // Then the row's tiles stream kTJ at a time (kTJ x 2 KB in flight a wave), B operands from the X panel.
  // The NEXT block's panel is rows K' of the matrix, all columns: their part right of the diagonal block is the transpose of
  // the tile columns [jn0, jn1) below the diagonal, so those tiles are ALSO stored transposed into the upper triangle (NB / 16
  // tile columns a block).  Everything else of the upper triangle is stale until invert()'s last pass mirrors the finished
  // lower triangle.
  static constexpr int kTJ = 2;
  typedef double d4 __attribute__((ext_vector_type(4)));
  __device__ __forceinline__ void rank_update_mfma(const double* __restrict__ Xp, const double* __restrict__ pinv, const int nt,
                                                   const int jn0, const int jn1) {
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (scalar: the tile loops branch on it)
    const int g = lane >> 4, j = lane & 15;
    constexpr int KS = NB / 4, KT = NB / 16;
    const int npair = (nt + 1) / 2;
    for (int pr = wave; pr < npair; pr += NT / 64) {
#pragma unroll 1
      for (int side = 0; side < 2; ++side) {
        const int I = side == 0 ? pr : nt - 1 - pr;
        if (side == 1 && I == pr) break;  // (odd nt: the middle row once)
        double a[KS];
        {
          d4 wt[KT];
#pragma unroll
          for (int t = 0; t < KT; ++t) wt[t] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            const double xb = Xp[(4 * s + g) * PITCH + 16 * I + j];
#pragma unroll
            for (int t = 0; t < KT; ++t)
              wt[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(pinv[(16 * t + j) * NB + 4 * s + g], xb, wt[t], 0, 0, 0);
          }
#pragma unroll
          for (int t = 0; t < KT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) a[4 * t + r] = -wt[t][r];
        }
        double* const arow = A + (size_t)(16 * I + g) * dp + j;  // entry (16 I + 4 r + g, 16 J + j) at arow[4 r dp + 16 J]
        // the row's tiles kTJ at a time, the NEXT chunk's loads issued before this chunk's instructions and stores
        d4 c[kTJ], cn[kTJ];
#pragma unroll
        for (int u = 0; u < kTJ; ++u)
          if (u <= I) {
#pragma unroll
            for (int r = 0; r < 4; ++r) c[u][r] = arow[(size_t)(4 * r) * dp + 16 * u];
          }
#pragma unroll 1
        for (int J0 = 0; J0 <= I; J0 += kTJ) {
#pragma unroll
          for (int u = 0; u < kTJ; ++u)
            if (J0 + kTJ + u <= I) {
#pragma unroll
              for (int r = 0; r < 4; ++r) cn[u][r] = arow[(size_t)(4 * r) * dp + 16 * (J0 + kTJ + u)];
            }
#pragma unroll
          for (int u = 0; u < kTJ; ++u)
            if (J0 + u <= I) {
#pragma unroll
              for (int s = 0; s < KS; ++s) {
                const double b = Xp[(4 * s + g) * PITCH + 16 * (J0 + u) + j];
                c[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], b, c[u], 0, 0, 0);
              }
            }
#pragma unroll
          for (int u = 0; u < kTJ; ++u)
            if (J0 + u <= I) {
              const int J = J0 + u;
#pragma unroll
              for (int r = 0; r < 4; ++r) arow[(size_t)(4 * r) * dp + 16 * J] = c[u][r];
              if (J >= jn0 && J < jn1 && I > J) {  // (wave-uniform) the next panel's rows, right of their diagonal tile
#pragma unroll
                for (int r = 0; r < 4; ++r) A[(size_t)(16 * J + j) * dp + 16 * I + 4 * r + g] = c[u][r];
              }
            }
#pragma unroll
          for (int u = 0; u < kTJ; ++u) c[u] = cn[u];
        }
      }
    }
  }

  // the NB x NB pivot block in LDS (pb): in-place Gauss-Jordan inverse (no pivoting: the block is a Schur complement of a
  // positive-definite matrix), its pivots = the Cholesky pivots squared.  flag[0] = 1 unless all pivots are positive and
  // finite, flag[1] = sum of their logarithms.  Round 6: by the WHOLE workgroup, one element a thread held in a register across
  // the NB elimination steps - a step publishes only the current pivot row and column (64 doubles, double-buffered: one
  // barrier a step); one wave walking 16 elements a lane through LDS took 27 % of a sweep.  Called by every thread.
  __device__ __forceinline__ void invert_pivot_block() {
    double* pb = lds + kOffPb;
    double* flag = lds + kOffFlag;
    double* rc = lds + kOffRC;
    const bool act = tid < NB * NB;
    const int r = tid / NB, c = tid % NB;
    double p = act ? pb[tid] : 0.0;
    double mypiv = 1.0;
    bool bad = false;
#pragma unroll 1
    for (int k = 0; k < NB; ++k) {
      double* b = rc + (k & 1) * 64;
      if (act && r == k) b[c] = p;
      if (act && c == k) b[32 + r] = p;
      __syncthreads();
      const double piv = b[k];
      if (!(piv > 0.0) || !(piv < 1.7e308)) bad = true;
      if (tid == k) mypiv = piv;
      const double d = 1.0 / piv;
      if (act) {
        const double pkc = b[c], prk = b[32 + r];
        double v;
        if (r == k && c == k) v = d;
        else if (r == k) v = p * d;
        else if (c == k) v = -p * d;
        else v = __builtin_fma(-prk * d, pkc, p);
        p = v;
      }
    }
    if (act) pb[tid] = p;
    if (tid < 64) {  // (threads 0 .. NB - 1 hold the pivots)
      const double l = wave_sum(tid < NB ? log(mypiv) : 0.0);
      if (tid == 0) {
        flag[0] = bad ? 1.0 : 0.0;
        flag[1] = l;
      }
    }
  }

  // ---- explicit inverse of the matrix build() left in the workspace: A <- -M^-1 ------------------------------------------
  __device__ __forceinline__ bool invert(double* logdet) {
    double* Xp = lds + kOffX;
    double* pb = lds + kOffPb;
    const double* flag = lds + kOffFlag;
    bool ok = true;
    double ld = 0.0;
    static_assert(NB == 16 || NB == 32, "pivot block: whole 16 x 16 tiles");
    const int nblk = (dim + NB - 1) / NB;  // (the padding beyond is the identity, decoupled from the rest)
    const int nt = (dim + 15) >> 4;        // 16 x 16 tiles a side that hold anything but that identity
#ifdef MM_GLOB_PROF
    long long pc[5] = {0, 0, 0, 0, 0}, pt = __builtin_readcyclecounter();
#define MM_GP(i) { const long long now_ = __builtin_readcyclecounter(); pc[i] += now_ - pt; pt = now_; }
#else
#define MM_GP(i)
#endif
    for (int blk = 0; blk < nblk; ++blk) {
      const int k0 = blk * NB;
      // (1) the panel: rows k0 .. k0 + NB - 1 of A (coalesced), X = Q - E, and the pivot block
      {
        const int nrep = NT / dp;  // threads a column: 1 (DP > 512), 2, 3 (DP = 320) - each takes every nrep-th row
        if (tid < nrep * dp) {
          const int h = tid / dp, col = tid - h * dp;
#pragma unroll 8
          for (int k = h; k < NB; k += nrep) {
            const double v = A[(size_t)(k0 + k) * dp + col];
            Xp[k * PITCH + col] = (col == k0 + k) ? v - 1.0 : v;
            if (col >= k0 && col < k0 + NB) pb[k * NB + (col - k0)] = v;
          }
        }
      }
      __syncthreads();
      MM_GP(0)
      // (2) P^-1 (first wave)
      invert_pivot_block();
      __syncthreads();
      MM_GP(1)
      if (flag[0] != 0.0) ok = false;  // (uniform: every thread reads the same cell)
      ld += flag[1];
      // (3) A -= (P^-1 X)^T X on the lower-triangle tiles (matrix cores; the W operands formed per tile row), then A_KK -= 2 I
      rank_update_mfma(Xp, pb, nt, (k0 + NB) >> 4, (k0 + 2 * NB) >> 4);
      __syncthreads();
      MM_GP(2)
      if (tid < NB) A[(size_t)(k0 + tid) * dp + k0 + tid] -= 2.0;
      __syncthreads();
      MM_GP(3)
    }
    // A = -M^-1 in the lower triangle: one more pass turns the sign and mirrors it into the upper triangle (1 / (D / NB) of the
    // sweep's traffic), so that the workspace IS the explicit inverse - what the products' column walks and a user's
    // vector-Jacobian product (through its dense accessor V(i, j)) read - and writes the scaled FP32 copy next to it
    {
      // the largest entry of a positive-definite matrix is on its diagonal: its binary exponent scales the FP32 copy
      const double dmax = reduce(tid < dim ? fabs(A[(size_t)tid * dp + tid]) : 0.0, true);
      int ex = 0;
      if (dmax > 0.0 && dmax < 1.7e308) (void)frexp(dmax, &ex);
      const double sdown = ldexp(1.0, -ex);
      pscale_ = ldexp(1.0, ex);
      const int lane = tid & 63, wave = tid >> 6;
      const int g = lane >> 4, j = lane & 15;
      const int ntiles = nt * (nt + 1) / 2;
      for (int t = wave; t < ntiles; t += NT / 64) {
        // tile number t -> (I, J), I >= J: I = floor((sqrt(8 t + 1) - 1) / 2), fixed up in integers
        int I = (int)((sqrt(8.0 * t + 1.0) - 1.0) * 0.5);
        while ((I + 1) * (I + 2) / 2 <= t) ++I;
        while (I * (I + 1) / 2 > t) --I;
        const int J = t - I * (I + 1) / 2;
        double v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = -A[(size_t)(16 * I + 4 * r + g) * dp + 16 * J + j];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          A[(size_t)(16 * I + 4 * r + g) * dp + 16 * J + j] = v[r];
          Af[(size_t)(16 * I + 4 * r + g) * dp + 16 * J + j] = (float)(v[r] * sdown);
        }
        if (I != J) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            A[(size_t)(16 * J + j) * dp + 16 * I + 4 * r + g] = v[r];
            Af[(size_t)(16 * J + j) * dp + 16 * I + 4 * r + g] = (float)(v[r] * sdown);
          }
        }
      }
    }
    __syncthreads();
    MM_GP(4)
#ifdef MM_GLOB_PROF
    if (tid == 0 && blockIdx.x == 0)
      printf("invert: panel %lld  pivot %lld  update %lld  diag %lld  mirror %lld cycles (%d blocks)\n", pc[0], pc[1], pc[2], pc[3],
             pc[4], nblk);
#endif
    if (logdet) *logdet = ld;
    // a NaN pivot poisons W and with it the whole matrix; a non-positive one is caught by the flag
    return ok && (ld == ld);
  }

  // diag(1 + q^2): the metric is diagonal - its inverse, log det and square root are elementwise, one register a
  // thread, no workspace and no sweep; every construction is "factorised" that way (init_backend switches the refinement off)
  double dinv_;  // MM_RMETRIC_DIAGQUAD: this thread's diagonal entry of the held inverse
  __device__ __forceinline__ bool build_diag(double x, double* logdet) {
    const double d = __builtin_fma(x, x, 1.0);
    dinv_ = tid < dim ? 1.0 / d : 0.0;
    const double bad = reduce((tid < dim && !(d < 1.7e308)) ? 1.0 : 0.0, false);  // "Array is not finite." (NaN / overflow)
    if (logdet) *logdet = reduce(tid < dim ? log(d) : 0.0, false);
    return bad == 0.0;
  }
  __device__ __forceinline__ bool build_and_invert(double x) {
    if constexpr (RMETRIC == MM_RMETRIC_DIAGQUAD) return build_diag(x, nullptr);
    const bool fin = build(x);
    return invert(nullptr) && fin;
  }

  // ---- y = M(x0)^-1 v -------------------------------------------------------------------------------------------------------
  __device__ __forceinline__ double matvec(double v) {
    if constexpr (RMETRIC == MM_RMETRIC_DIAGQUAD) return tid < dim ? dinv_ * v : 0.0;
    publish(v);
    if constexpr (kLowRankBuiltin) {
      if (sym_ok()) return sym_walk(A);
    }
    return column_walk(A, dp, dim);
  }
  // z = F r of the refinement solves (implicit_core.h precond_trait): the scaled FP32 copy
  __device__ __forceinline__ double precond(double v) {
    if constexpr (RMETRIC == MM_RMETRIC_DIAGQUAD) return tid < dim ? dinv_ * v : 0.0;
    publish(v);
    return pscale_ * column_walk(Af, dp, dim);
  }
  // the lock-step pair's preconditioner products z = F r (refine_solve2 calls nothing else through matvec2): one pass
  __device__ __forceinline__ void matvec2(double v0, double v1, double* y0, double* y1) {
    if constexpr (RMETRIC == MM_RMETRIC_DIAGQUAD) {
      *y0 = tid < dim ? dinv_ * v0 : 0.0;
      *y1 = tid < dim ? dinv_ * v1 : 0.0;
      return;
    }
    publish2(v0, v1);
    double a, b;
    column_walk2(Af, dp, dim, &a, &b);
    *y0 = pscale_ * a;
    *y1 = pscale_ * b;
  }
  // implicit_core.h lowrank_update: F += al a a^T + be (a b^T + b a^T) + ga b b^T - ONE read-modify-write pass over the FP64
  // workspace (4 MB at D = 512) where the sweep makes ceil(D / NB) + 3.  Entry (j, c) takes a_j u_c + b_j v_c, u = al a + be b,
  // v = be a + ga b: a, b broadcast from LDS per row, u_c, v_c in registers per column (coalesced over the threads, rows split
  // between the two threads of a column as in column_walk).  The FP32 copy is the CG refinement's preconditioner: not read
  // while the Woodbury path is on, rewritten by the next factorisation.
  __device__ __forceinline__ void inverse_update(double al, double be, double ga, double a, double b) {
    if constexpr (RMETRIC == MM_RMETRIC_DIAGQUAD) return;
    publish2(a, b);
    double* part = lds + kOffPart;  // [2][NT]
    part[tid] = tid < dim ? __builtin_fma(al, a, be * b) : 0.0;
    part[NT + tid] = tid < dim ? __builtin_fma(be, a, ga * b) : 0.0;
    __syncthreads();
    const bool two = 2 * dim <= NT;
    const int h = (two && tid >= NT / 2) ? 1 : 0;
    const int c = tid - h * (NT / 2);
    const int nh = two ? ((dim + 1) >> 1) : dim;
    int j0 = h * nh;
    const int j1 = (j0 + nh < dim) ? j0 + nh : dim;
    // (while the products read the lower tiles only - sym_walk - so does the update: the diagonal tiles whole, nothing above
    // them; the next factorisation's mirror pass rewrites the upper triangle)
    if constexpr (kLowRankBuiltin) {
      if (sym_ok() && j0 < (c & ~15)) j0 = c & ~15;
    }
    if (c < dim) {
      const double uc = part[c], vc = part[NT + c];
      const double* nat = lds + kOffNat;
      const double* nat2 = lds + kOffNat2;
      double* col = A + c;
      int j = j0;
      for (; j + kWalk <= j1; j += kWalk) {
        double x[kWalk];
#pragma unroll
        for (int e = 0; e < kWalk; ++e) x[e] = col[(size_t)(j + e) * dp];
#pragma unroll
        for (int e = 0; e < kWalk; ++e)
          col[(size_t)(j + e) * dp] = __builtin_fma(nat[j + e], uc, __builtin_fma(nat2[j + e], vc, x[e]));
      }
      for (; j < j1; ++j) col[(size_t)j * dp] = __builtin_fma(nat[j], uc, __builtin_fma(nat2[j], vc, col[(size_t)j * dp]));
    }
    __syncthreads();  // (workgroup-scope release / acquire of the global stores: the next product reads them)
  }
  // two products with the held inverse ITSELF in one pass (implicit_core.h lowrank_solve2)
  __device__ __forceinline__ void matvec2_exact(double v0, double v1, double* y0, double* y1) {
    if constexpr (RMETRIC == MM_RMETRIC_DIAGQUAD) {
      *y0 = tid < dim ? dinv_ * v0 : 0.0;
      *y1 = tid < dim ? dinv_ * v1 : 0.0;
      return;
    }
    publish2(v0, v1);
    column_walk2(A, dp, dim, y0, y1);
  }
  __device__ __forceinline__ void metric_point2(double x0, double x1) {
    xpt_ = tid < dim ? x0 : 0.0;
    xpt2_ = tid < dim ? x1 : 0.0;
  }
  __device__ __forceinline__ void metric_apply2(double v0, double v1, double* y0, double* y1) {
    if constexpr (RMETRIC == MM_RMETRIC_DIAGQUAD) {
      *y0 = tid < dim ? __builtin_fma(xpt_ * xpt_, v0, v0) : 0.0;
      *y1 = tid < dim ? __builtin_fma(xpt2_ * xpt2_, v1, v1) : 0.0;
    } else if constexpr (RMETRIC == MM_RMETRIC_RANK1) {
      publish2(v0, v1);
      double a, b, d0, d1;
      column_walk2(base, dim, dim, &a, &b);  // B v0, B v1: one pass over the base matrix
      sum2(xpt_ * v0, xpt2_ * v1, &d0, &d1);
      *y0 = tid < dim ? __builtin_fma(xpt_, d0 * inv_dim_, a) : 0.0;
      *y1 = tid < dim ? __builtin_fma(xpt2_, d1 * inv_dim_, b) : 0.0;
    } else {
      *y0 = *y1 = 0.0;  // (user metrics: kDual is false)
    }
  }
  __device__ __forceinline__ double diag() const {
    if constexpr (RMETRIC == MM_RMETRIC_DIAGQUAD) return tid < dim ? dinv_ : 0.0;
    return tid < dim ? A[(size_t)tid * dp + tid] : 0.0;
  }
  // ---- refinement products: M(x) v matrix-free ---------------------------------------------------------------------------
  __device__ __forceinline__ void metric_point(double x) {
    xpt_ = tid < dim ? x : 0.0;
    if constexpr (RMETRIC == MM_RMETRIC_USER) {  // the products' point in natural order + its aux block, in the idle X panel
      __syncthreads();
      lds[kOffUx + tid] = xpt_;
      __syncthreads();
      mmuser::prepare(TeamOfGlobal<GlobalBackend>{*this}, lds + kOffUx, dim, base, lds + kOffUax);
      __syncthreads();
    }
  }
  __device__ __forceinline__ double metric_apply(double v) {
    if constexpr (RMETRIC == MM_RMETRIC_DIAGQUAD) {
      return tid < dim ? __builtin_fma(xpt_ * xpt_, v, v) : 0.0;
    } else if constexpr (RMETRIC == MM_RMETRIC_USER) {
      // (M(x) v)_i = sum_j M_ij(x) v_j with the user's entries evaluated on the fly (M symmetric: entry (j, i))
      publish(v);
      if (tid >= dim) return 0.0;
      const double* nat = lds + kOffNat;
      double y0 = 0.0, y1 = 0.0;
      int j = 0;
      for (; j + 2 <= dim; j += 2) {
        y0 = __builtin_fma(mmuser::entry(lds + kOffUx, j, tid, dim, base, lds + kOffUax), nat[j], y0);
        y1 = __builtin_fma(mmuser::entry(lds + kOffUx, j + 1, tid, dim, base, lds + kOffUax), nat[j + 1], y1);
      }
      if (j < dim) y0 = __builtin_fma(mmuser::entry(lds + kOffUx, j, tid, dim, base, lds + kOffUax), nat[j], y0);
      return y0 + y1;
    } else {
      publish(v);
      const double y = column_walk(base, dim, dim);  // B v (B symmetric)
      const double dot = sum1(xpt_ * v);
      return tid < dim ? __builtin_fma(xpt_, dot * inv_dim_, y) : 0.0;
    }
  }

  // 0.5 * vjp_metric_func(q)(V) of a user metric; q is the point of the held inverse (build() left it at kOffUq with its aux
  // block).  OUTER: V = -u u^T, else the explicit inverse - handed to the user's team-form hook (MM_USER_VJP_FLAT) or read
  // through the dense accessor straight from the workspace.
  template <bool OUTER>
  __device__ __forceinline__ double user_half_vjp(double u) {
    double r = 0.0;
    const double* uq = lds + kOffUq;
    const double* uaq = lds + kOffUaq;
    if constexpr (mmuser::kFlatVjp) {
      if constexpr (OUTER) {
        mmuser::VjpOpsOuter<GlobalBackend> ops{*this, tid < dim ? u : 0.0};
        r = mmuser::vjp_flat(ops, uq, tid, dim, base, uaq);
      } else {
        mmuser::VjpOpsInv<GlobalBackend> ops{*this};
        r = mmuser::vjp_flat(ops, uq, tid, dim, base, uaq);
      }
    } else {
#if defined(MM_RTC_BUILD) && defined(MM_RTC_USER_METRIC)
      if constexpr (OUTER) {
        publish(u);
        const MmMat vm{nullptr, lds + kOffNat, 0};
        r = (tid < dim) ? mmuser::vjp_dense(uq, vm, tid, dim, base, uaq) : 0.0;
      } else {
        const MmMat vm{A, nullptr, dp};
        r = (tid < dim) ? mmuser::vjp_dense(uq, vm, tid, dim, base, uaq) : 0.0;
      }
#endif
    }
    return tid < dim ? 0.5 * r : 0.0;
  }

  // 0.5 * vjp_metric(M^-1): rank-one metric M^-1 q / D; diag-quad metric q_i (M^-1)_ii
  __device__ __forceinline__ double half_vjp_inv(double q) {
    if constexpr (RMETRIC == MM_RMETRIC_USER) return user_half_vjp<false>(0.0);
    else if constexpr (RMETRIC == MM_RMETRIC_RANK1) return matvec(q) * inv_dim_;
    else return tid < dim ? q * dinv_ : 0.0;
  }
  // dense metric: grad_quadratic_form_inv(p) = -(M^-1 p)(M^-1 p)^T   (matrices.py:1179-1181)
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
  __device__ __forceinline__ double grad(double q) {
    publish(q);
    const TargetAux aux = target_prepare<false>(target, lds + kOffNat, dim, tparams, tid & 63);
    return tid < dim ? target_grad_elem<false>(target, aux, lds + kOffNat, tid, dim, tparams) : 0.0;
  }
  __device__ __forceinline__ double neg_log_dens_elem(double q) {
    publish(q);
    const TargetAux aux = target_prepare<false>(target, lds + kOffNat, dim, tparams, tid & 63);
    return tid < dim ? target_nld_elem<false>(target, aux, lds + kOffNat, tid, dim, tparams) : 0.0;
  }

  // ---- Cholesky factor of the matrix build() left in the workspace, stored TRANSPOSED: row k of A holds L[:, k] from the
  // diagonal on (blocked, right-looking: the trailing update is rank_update with W = X = the panel of L) --------------------
  __device__ __forceinline__ bool cholesky_transposed() {
    double* Xp = lds + kOffX;
    double* pb = lds + kOffPb;
    const double* flag = lds + kOffFlag;
    bool ok = true;
    const int nblk = (dim + CB - 1) / CB;
    for (int blk = 0; blk < nblk; ++blk) {
      const int k0 = blk * CB;
      if (tid >= k0 && tid < k0 + CB) {
#pragma unroll
        for (int k = 0; k < CB; ++k) pb[k * CB + (tid - k0)] = A[(size_t)(k0 + k) * dp + tid];
      }
      __syncthreads();
      if (tid == 0) {  // the CB x CB block's own Cholesky factor, in place (lower triangle of pb), sequentially
        double bad = 0.0;
        for (int j = 0; j < CB; ++j) {
          double d = pb[j * CB + j];
          for (int m = 0; m < j; ++m) d -= pb[j * CB + m] * pb[j * CB + m];
          if (!(d > 0.0) || !(d < 1.7e308)) bad = 1.0;
          const double l = sqrt(d);
          pb[j * CB + j] = l;
          for (int i = j + 1; i < CB; ++i) {
            double s = pb[i * CB + j];
            for (int m = 0; m < j; ++m) s -= pb[i * CB + m] * pb[j * CB + m];
            pb[i * CB + j] = s / l;
          }
        }
        lds[kOffFlag] = bad;
      }
      __syncthreads();
      if (flag[0] != 0.0) ok = false;
      // column j of the panel: y = Lkk^-1 A[K, j] (forward substitution) = L[j][K]^T for j beyond the block; inside the
      // block the factor itself.  Zero left of the block (so that the full-range trailing update leaves those parts alone).
      if (tid < dp) {
        double y[CB];
        if (tid >= k0 + CB) {
#pragma unroll
          for (int k = 0; k < CB; ++k) {
            double s = A[(size_t)(k0 + k) * dp + tid];
#pragma unroll
            for (int m = 0; m < CB; ++m)
              if (m < k) s -= pb[k * CB + m] * y[m];
            y[k] = s / pb[k * CB + k];
          }
        } else {
#pragma unroll
          for (int k = 0; k < CB; ++k) y[k] = (tid >= k0 && tid - k0 >= k) ? pb[(tid - k0) * CB + k] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < CB; ++k) {
          Xp[k * PITCH + tid] = tid >= k0 + CB ? y[k] : 0.0;
          if (tid >= k0) A[(size_t)(k0 + k) * dp + tid] = y[k];  // row k0 + k of A <- L[:, k0 + k]
        }
      }
      __syncthreads();
      rank_update<CB>(Xp, Xp);  // trailing A[i][j] -= sum_k L[i][k] L[j][k] (zero panel entries elsewhere)
      __syncthreads();
    }
    return ok;
  }
};

template <int RMETRIC, int NB>
__device__ __forceinline__ void init_backend(GlobalBackend<RMETRIC, NB>& bk, const ImplicitArgs& A, double* lds) {
  bk.dim = A.dim;
  bk.dp = padded_dim(A.dim);
  bk.tid = threadIdx.x;
  bk.target = A.target;
  bk.flip = 0;
  bk.inv_dim_ = 1.0 / (double)A.dim;
  bk.lds = lds;
  bk.A = A.work + (size_t)blockIdx.x * bk.dp * bk.dp;
  // (the FP32 copies behind the n_chains FP64 matrices: the hosts size the workspace at 12 bytes an entry)
  bk.Af = reinterpret_cast<float*>(A.work + (size_t)A.n_chains * bk.dp * bk.dp) + (size_t)blockIdx.x * bk.dp * bk.dp;
  bk.pscale_ = 1.0;
  bk.base = A.rparams;
  bk.tparams = A.tparams;
  bk.refine_on = A.no_refine == 0 && RMETRIC != MM_RMETRIC_DIAGQUAD;  // (a diagonal metric: every construction elementwise)
  bk.dinv_ = 0.0;
  // (sym_walk reads half the matrix per product: the lock step's shared pass - the same saving for two of a step's ~24
  // products only - is off while it is on)
  bk.sym_on_ = A.no_sym == 0 && A.no_lowrank == 0 && A.no_refine == 0;
  bk.dual_off = A.no_dual != 0 || (RMETRIC == MM_RMETRIC_RANK1 && bk.sym_on_);
  bk.lowrank_on_ = A.no_lowrank == 0 && A.no_refine == 0;
  bk.lr_refresh_ = A.lowrank_refresh;
  bk.xpt_ = 0.0;
  bk.xpt2_ = 0.0;
}

// MIDPOINT: ImplicitMidpointIntegrator (integrators.py:547-681) on the same backend (a full sweep per function evaluation)
template <int RMETRIC, int NB, bool MIDPOINT>
__device__ __forceinline__ void implicit_global_body(const ImplicitArgs& A, double* lds) {
  GlobalBackend<RMETRIC, NB> bk;
  init_backend(bk, A, lds);
  const int64_t chain = blockIdx.x;
  const int tid = threadIdx.x, dim = A.dim;
  const bool act = tid < dim;
  bk.slot(SL_Q) = act ? A.pos[chain * dim + tid] : 0.0;
  bk.slot(SL_P) = act ? A.mom[chain * dim + tid] : 0.0;
  const double t = signed_step(A.dir, A.step_scale, chain, A.step_size);
  const int my_steps = mmdev::chain_steps(A.chain_steps, chain, A.n_steps);
  const ChainResult r = MIDPOINT ? implicit_midpoint_chain(bk, t, my_steps, A.opts)
                                 : implicit_leapfrog_chain(bk, t, my_steps, A.opts);
  if (act) {
    A.pos[chain * dim + tid] = bk.slot(SL_Q);
    A.mom[chain * dim + tid] = bk.slot(SL_P);
  }
  if (tid == 0) {
    A.status[chain] = r.status;
    A.n_done[chain] = r.done;
    add_counters(A.counters, r);
  }
}

// OP 0: h = nld + log det M / 2 + p^T M^-1 p / 2;  1: dh_dmom = M^-1 p;  2: sample_momentum: mom <- L z   (systems.py:1375-1402)
template <int RMETRIC, int NB, int OP>
__device__ __forceinline__ void riemann_aux_global_body(const ImplicitArgs& A, double* lds) {
  GlobalBackend<RMETRIC, NB> bk;
  init_backend(bk, A, lds);
  const int64_t chain = blockIdx.x;
  const int tid = threadIdx.x, dim = A.dim;
  const bool act = tid < dim;
  const double q = act ? A.pos[chain * dim + tid] : 0.0;
  const double p = act ? A.mom[chain * dim + tid] : 0.0;
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
  if constexpr (RMETRIC == MM_RMETRIC_DIAGQUAD) {  // elementwise (see build_diag)
    double logdet;
    const bool okd = bk.build_diag(q, &logdet);
    if constexpr (OP == 0) {
      const double e = bk.neg_log_dens_elem(q) + (act ? 0.5 * p * bk.matvec(p) : 0.0);
      const double h = bk.reduce(e, false) + 0.5 * logdet;
      if (tid == 0) A.out[chain] = okd ? h : nan;
    } else if constexpr (OP == 1) {
      const double u = bk.matvec(p);
      if (act) A.out[chain * dim + tid] = okd ? u : nan;
    } else {
      if (act) A.mom[chain * dim + tid] = okd ? sqrt(__builtin_fma(q, q, 1.0)) * A.z[chain * dim + tid] : nan;
    }
    return;
  }
  bool ok = bk.build(q);
  if constexpr (OP == 0) {
    double logdet;
    ok = bk.invert(&logdet) && ok;
    const double u = bk.matvec(p);
    const double e = bk.neg_log_dens_elem(q) + (act ? 0.5 * p * u : 0.0);
    const double h = bk.reduce(e, false) + 0.5 * logdet;
    if (tid == 0) A.out[chain] = ok ? h : nan;
  } else if constexpr (OP == 1) {
    ok = bk.invert(nullptr) && ok;
    const double u = bk.matvec(p);
    if (act) A.out[chain * dim + tid] = ok ? u : nan;
  } else {
    ok = bk.cholesky_transposed() && ok;
    bk.publish(act ? A.z[chain * dim + tid] : 0.0);
    // (L z)_i = sum_{k <= i} L[i][k] z_k = sum_{k <= i} A[k][i] z_k: the column walk stops at the diagonal
    double y = 0.0;
    if (act) {
      const double* nat = lds + kOffNat;
      for (int k = 0; k <= tid; ++k) y = __builtin_fma(bk.A[(size_t)k * bk.dp + tid], nat[k], y);
    }
    if (act) A.mom[chain * dim + tid] = ok ? y : nan;
  }
}

#ifndef MM_RTC_BUILD  // the in-tree instantiations (a run-time translation unit defines extern "C" wrappers instead)
template <int RMETRIC, int NB, bool MIDPOINT>
__global__ __launch_bounds__(NT) void implicit_global_kernel(ImplicitArgs A) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  implicit_global_body<RMETRIC, NB, MIDPOINT>(A, lds);
}
template <int RMETRIC, int NB, int OP>
__global__ __launch_bounds__(NT) void riemann_aux_global_kernel(ImplicitArgs A) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  riemann_aux_global_body<RMETRIC, NB, OP>(A, lds);
}

#endif

}  // namespace mmglob
