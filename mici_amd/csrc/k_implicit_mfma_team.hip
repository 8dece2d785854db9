// Implicit leapfrog on dense-metric Riemannian systems, 64 < D <= 256 (BASELINE config c4: D = 256):
// one 512-thread workgroup (8 waves, a whole CU) per chain, metric inverted by the blocked symmetric
// sweep of k_implicit_mfma.hip with its rank-4 updates on the FP64 matrix cores.  gfx950 / CDNA4.
//
// Layout.  D is padded to 256 = 16 x 16 tiles of 16 x 16, of which the 136 on or below the diagonal are
// kept, each in the MFMA accumulator layout (lane l = 16 g + j, register r <-> entry
// (16 I + 4 r + g, 16 J + j)).  Wave w owns the two tile ROWS Ia = w and Ib = 15 - w: (w + 1) + (16 - w) =
// 17 tiles per wave, 272 KB per chain in the CU's register file.  Within a row the slot index k is the
// distance from the diagonal, tile (I, I - k): the diagonal tile is always slot 0 and every LDS address of
// slot k is a compile-time offset from a per-row base, so the only run-time (wave-uniform) quantities are
// Ia, Ib; slots k > I of the two fixed-size register arrays (8 + 16 slots) are unused.
//
// Per block of four pivot columns K = 16 I0 + 4 R0 + {0..3} (64 blocks per inversion):
//   (1) publish the panel Q = A[K, :] to LDS: the wave owning tile row I0 writes register R0 of that row's
//       tiles; every wave with a row below writes, transposed, its tile in column I0; barrier;
//   (2) waves 0-3: invert the 4 x 4 pivot block (uniform, closed form via 2 x 2 Schur complements); thread
//       c < 256 turns column c of the panel into column c of -W = -P^-1 (Q - E); barrier;
//   (3) 17 MFMAs per wave: tile (I, J) += (-W)[:, tile I]^T (Q - E)[:, tile J];  A_KK -= 2 I.
// Panel buffers are double-buffered by block parity, so two barriers per block suffice.  With two waves
// per SIMD the matrix-core time (2 x 17 x 64 cycles per block) is the floor this kernel aims at.
//
// M^-1 v: a tile row first accumulates its four row sums over its tiles in-lane, then reduces them over the
// 16 lanes of a DPP row with ONE transposing butterfly; tiles below the diagonal also give sixteen column
// sums each (reduced over the four DPP rows).  They land in part[n][slot] - slot 16 for the row sum, slot
// I for the column sums of tile row I - so the final sum has a fixed order (bitwise reproducible).
//
// Reference arithmetic replaced: DensePositiveDefiniteMatrix factorisation + explicit inverse
// (matrices.py:1161-1188) inside ImplicitLeapfrogIntegrator._step (integrators.py:493-544); the step
// logic is implicit_core.h.
#include "implicit_core.h"

namespace {

using namespace mmdev;
using namespace mmimp;

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int NT16 = 16;              // tile rows
constexpr int DPM = 16 * NT16;        // padded dimension
constexpr int NWAVE = 8;
constexpr int NTHR = 64 * NWAVE;
constexpr int NSA = 8, NSB = 16;      // slots of the two tile rows of a wave (rows w <= 7 and 15 - w >= 8)
constexpr int PSTR = 17;              // row stride of the partial-sum array: 16 column-sum slots + the row sum
constexpr int VLM = DPM + 8;          // flat vectors: DPM elements + a dummy cell for threads >= DPM

// LDS (doubles): panel/W double buffers, flat vectors, partial sums, reduction scratch, step state
constexpr int kOffQt = 0;                          // [2][DPM][4]
constexpr int kOffWt = kOffQt + 2 * DPM * 4;       // [2][DPM][4]
constexpr int kOffXt = kOffWt + 2 * DPM * 4;       // [2][DPM][4] X = Q - E.  NOT written over the panel: while thread
                                                   // c stores its column, other waves may still be reading the
                                                   // pivot block P = Q[:, K] from the panel
constexpr int kOffNat = kOffXt + 2 * DPM * 4;      // [VLM]
constexpr int kOffVperm = kOffNat + VLM;           // [DPM]
constexpr int kOffAux = kOffVperm + DPM;           // [VLM]
constexpr int kOffRed = kOffAux + VLM;             // [16]
constexpr int kOffPart = kOffRed + 16;             // [DPM][PSTR]
constexpr int kOffStash = kOffPart + DPM * PSTR;   // [SL_COUNT][VLM]
constexpr int kTeamLdsDoubles = kOffStash + SL_COUNT * VLM;

// Launder a value so that address arithmetic derived from it is recomputed at the use instead of being
// hoisted out of the step loop into long-lived registers (the register file is full of metric tiles).
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}
// The lane index straight from the hardware (two VALU instructions).  The sweep needs lane-derived LDS
// addresses right after every barrier; held in a member they were spilled (the register file is full of tiles)
// and every block waited on a scratch reload at its most latency-critical point.
__device__ __forceinline__ int fresh_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
__device__ __forceinline__ int opaque_s(int v) {
  v = __builtin_amdgcn_readfirstlane(v);
  asm volatile("" : "+s"(v));
  return v;
}
__device__ __forceinline__ double team_reduce(double v, int kind_max, double* red) {
  // kind_max: 0 sum, 1 NaN-propagating max.  Uniform result; two barriers.
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = kind_max ? wave_max(v) : wave_sum(v);
  if (lane == 0) red[wave] = v;
  __syncthreads();
  double r = red[0];
#pragma unroll
  for (int w = 1; w < NWAVE; ++w) r = kind_max ? nanmax(r, red[w]) : r + red[w];
  __syncthreads();
  return r;
}

template <int RMETRIC, int W>
struct TeamMfma {
  static constexpr bool kSolveByInverse = true;  // implicit_core.h: solve = invert + mat-vec, one construction site
  static constexpr bool kUnifiedConstruct = false;
  static constexpr bool kCountersInLds = false;
  d4 accA[NSA];  // tile row Ia = wave:      slot k <-> tile (Ia, Ia - k), valid for k <= Ia
  d4 accB[NSB];  // tile row Ib = 15 - wave: slot k <-> tile (Ib, Ib - k), valid for k <= Ib
  // The wave index is a TEMPLATE parameter: the kernel switches on it once and each wave runs its own
  // instance, in which the tile rows, the validity of every register slot and most LDS addresses are
  // compile-time (no per-slot branches around the MFMAs, operand loads hoisted by the compiler).
  static constexpr int Ia = W, Ib = NT16 - 1 - W;
  int dim, tid, lane, gq, jq, target;
  double* lds;
  const double* base;  // rank-one metric: base matrix zero-padded, leading dimension base_ld
  int base_ld;
  const double* tparams;

  __device__ __forceinline__ double& slot(int i) { return lds[kOffStash + i * VLM + (tid < DPM ? tid : DPM)]; }

  __device__ __forceinline__ double norm(double x, int kind) {
    const double a = tid < dim ? x : 0.0;
    if (kind == MM_NORM_LINF) return team_reduce(fabs(a), 1, lds + kOffRed);
    return sqrt(team_reduce(a * a, 0, lds + kOffRed));
  }

  // natural-order copy + the [I][g][r] permuted copy that feeds row operands as one 32-byte read
  __device__ __forceinline__ void publish_vector(double x) {
    if (tid < DPM) {
      const double xm = tid < dim ? x : 0.0;
      lds[kOffNat + tid] = xm;
      lds[kOffVperm + ((((tid >> 4) << 2) + (tid & 3)) << 2) + ((tid >> 2) & 3)] = xm;
    }
    __syncthreads();
  }

  // ---- metric_func(x) into one tile row -------------------------------------------------------------
  template <int NS>
  __device__ __forceinline__ void build_row(d4 (&acc)[NS], const int I, double& chk) {
    const int ln = fresh_lane(), g = ln >> 4, j = ln & 15;
    const double inv_d = 1.0 / (double)dim;
    const d4 qr = *reinterpret_cast<const d4*>(lds + kOffVperm + ((I * 4 + g) << 2));
    const double* nat_row = lds + kOffNat + 16 * I + j;                          // - 16 k
    const double* brow = base + (int64_t)(16 * I + g) * base_ld + 16 * I + j;    // - 16 k, + 4 r base_ld
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      if (k <= I) {  // wave-uniform
        if constexpr (RMETRIC == MM_RMETRIC_RANK1) {
          const double qs = nat_row[-16 * k] * inv_d;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            acc[k][r] = __builtin_fma(qr[r], qs, brow[(int64_t)(4 * r) * base_ld - 16 * k]);
        } else {
          acc[k] = d4{0.0, 0.0, 0.0, 0.0};
        }
      }
    }
    // diagonal tile = slot 0: entries (16 I + 4 r + g, same) sit on lanes with j == 4 r + g
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool on_diag = (j == 4 * r + g);
      if constexpr (RMETRIC == MM_RMETRIC_DIAGQUAD) {
        if (on_diag) acc[0][r] = __builtin_fma(qr[r], qr[r], 1.0);
      }
      if (on_diag && 16 * I + 4 * r + g >= dim) acc[0][r] = 1.0;  // identity on the padding
      // "Array is not finite." (matrices.py:211-215): both built-in metrics have their largest entries
      // on the diagonal, so a non-finite entry implies a non-finite diagonal-tile entry
      chk = __builtin_fma(acc[0][r], 0.0, chk);
    }
  }

  __device__ __forceinline__ bool build(double x) {
    publish_vector(x);
    double chk = 0.0;
    build_row<NSA>(accA, Ia, chk);
    build_row<NSB>(accB, Ib, chk);
    const double bad = team_reduce(chk == 0.0 ? 0.0 : 1.0, 0, lds + kOffRed);
    return bad == 0.0;
  }

  // ---- sweep, phase (1): this row's share of the panel  qt[c][s] = A[k0 + s][c] ------------------------
  template <int NS, int R0>
  __device__ __forceinline__ void publish_row(const d4 (&acc)[NS], const int I, const int I0, double* qt) {
    const int ln = fresh_lane(), g = ln >> 4, j = ln & 15;
    if (I == I0) {
      double* dst = qt + ((16 * I + j) << 2) + g;  // - 64 k
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        if (k <= I) dst[-64 * k] = acc[k][R0];
      }
    } else if (I > I0) {
      // tile (I, I0) = slot I - I0: its columns K live on the 16 lanes with (j >> 2) == R0
      double* dst = qt + ((16 * I + g) << 2) + (j & 3);  // + 16 r
      const bool mine = (j >> 2) == R0;
      const int kk = I - I0;
#pragma unroll
      for (int k = 1; k < NS; ++k) {
        if (kk == k) {  // wave-uniform
          if (mine) {
#pragma unroll
            for (int r = 0; r < 4; ++r) dst[16 * r] = acc[k][r];
          }
        }
      }
    }
  }

  // ---- sweep, phase (3): rank-4 update of one tile row on the matrix cores -----------------------------
  template <int NS, int R0>
  __device__ __forceinline__ void update_row(d4 (&acc)[NS], const int I, const int I0, const double* qt,
                                             const double* wt) {
    const int ln = fresh_lane(), g = ln >> 4, j = ln & 15;
    const double av = wt[((16 * I + j) << 2) + g];
    const double* src = qt + ((16 * I + j) << 2) + g;  // - 64 k
#pragma unroll
    for (int k = 0; k < NS; ++k) {
      if (k <= I) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, src[-64 * k], acc[k], 0, 0, 0);
    }
    if (I == I0) {
      if (j == 4 * R0 + g) acc[0][R0] -= 2.0;
    }
  }

  // ---- one block of the sweep; R0 compile-time, I0 run-time (wave-uniform) ----------------------------
  template <int R0>
  __device__ __forceinline__ void block_step(const int I0, const int par, bool& ok) {
    double* qt = lds + kOffQt + par * (DPM * 4);
    double* wt = lds + kOffWt + par * (DPM * 4);
    double* xt = lds + kOffXt + par * (DPM * 4);
    const int k0 = 16 * I0 + 4 * R0;
    const int tid = W * 64 + fresh_lane();
    const int ia = Ia, ib = Ib;
    publish_row<NSA, R0>(accA, ia, I0, qt);
    publish_row<NSB, R0>(accB, ib, I0, qt);
    __syncthreads();
    // (2) P^-1 (uniform) and column tid of -W: the 256 threads of waves 0-3
    if (tid < DPM) {
      const d4 c0 = *reinterpret_cast<const d4*>(qt + ((k0 + 0) << 2));
      const d4 c1 = *reinterpret_cast<const d4*>(qt + ((k0 + 1) << 2));
      const d4 c2 = *reinterpret_cast<const d4*>(qt + ((k0 + 2) << 2));
      const d4 c3 = *reinterpret_cast<const d4*>(qt + ((k0 + 3) << 2));
      // P = [A B; B^T C] with 2 x 2 blocks; c_s is column s of P
      const double a = c0[0], b = c0[1], e = c1[1];
      const double b00 = c0[2], b01 = c0[3], b10 = c1[2], b11 = c1[3];
      const double h = c2[2], i2 = c2[3], jj = c3[3];
      const double det_a = __builtin_fma(a, e, -b * b);
      const double ida = fast_rcp(det_a);
      const double ia00 = e * ida, ia01 = -b * ida, ia11 = a * ida;
      const double t00 = __builtin_fma(ia00, b00, ia01 * b10), t01 = __builtin_fma(ia00, b01, ia01 * b11);
      const double t10 = __builtin_fma(ia01, b00, ia11 * b10), t11 = __builtin_fma(ia01, b01, ia11 * b11);
      const double s00 = h - __builtin_fma(b00, t00, b10 * t10);
      const double s01 = i2 - __builtin_fma(b00, t01, b10 * t11);
      const double s11 = jj - __builtin_fma(b01, t01, b11 * t11);
      const double det_s = __builtin_fma(s00, s11, -s01 * s01);
      const double ids = fast_rcp(det_s);
      const double is00 = s11 * ids, is01 = -s01 * ids, is11 = s00 * ids;
      // pivots of the sequential elimination: a, det_a / a, s00, det_s / s00 (all must be > 0)
      ok = ok && (a > 0.0) && (det_a > 0.0) && (s00 > 0.0) && (det_s > 0.0);
      const double u00 = __builtin_fma(t00, is00, t01 * is01), u01 = __builtin_fma(t00, is01, t01 * is11);
      const double u10 = __builtin_fma(t10, is00, t11 * is01), u11 = __builtin_fma(t10, is01, t11 * is11);
      const double p00 = ia00 + __builtin_fma(u00, t00, u01 * t01);
      const double p01 = ia01 + __builtin_fma(u00, t10, u01 * t11);
      const double p11 = ia11 + __builtin_fma(u10, t10, u11 * t11);
      d4 q = *reinterpret_cast<const d4*>(qt + (tid << 2));
      const int s = tid - k0;
      q[0] -= (s == 0) ? 1.0 : 0.0;
      q[1] -= (s == 1) ? 1.0 : 0.0;
      q[2] -= (s == 2) ? 1.0 : 0.0;
      q[3] -= (s == 3) ? 1.0 : 0.0;
      d4 wv;
      wv[0] = __builtin_fma(-p00, q[0], __builtin_fma(-p01, q[1], __builtin_fma(u00, q[2], u01 * q[3])));
      wv[1] = __builtin_fma(-p01, q[0], __builtin_fma(-p11, q[1], __builtin_fma(u10, q[2], u11 * q[3])));
      wv[2] = __builtin_fma(u00, q[0], __builtin_fma(u10, q[1], __builtin_fma(-is00, q[2], -is01 * q[3])));
      wv[3] = __builtin_fma(u01, q[0], __builtin_fma(u11, q[1], __builtin_fma(-is01, q[2], -is11 * q[3])));
      *reinterpret_cast<d4*>(xt + (tid << 2)) = q;
      *reinterpret_cast<d4*>(wt + (tid << 2)) = wv;
    }
    __syncthreads();
    update_row<NSA, R0>(accA, ia, I0, xt, wt);
    update_row<NSB, R0>(accB, ib, I0, xt, wt);
    // no barrier: the next block uses the other panel / W buffers
  }

  __device__ __forceinline__ bool sweep() {
    bool ok = true;
#pragma unroll 1
    for (int I0 = 0; I0 < NT16; ++I0) {
      block_step<0>(I0, 0, ok);
      block_step<1>(I0, 1, ok);
      block_step<2>(I0, 0, ok);
      block_step<3>(I0, 1, ok);
    }
    const int ia = Ia, ib = Ib;
#pragma unroll
    for (int k = 0; k < NSA; ++k)
      if (k <= ia) accA[k] = -accA[k];
#pragma unroll
    for (int k = 0; k < NSB; ++k)
      if (k <= ib) accB[k] = -accB[k];
    // the pivot checks ran on waves 0-3 only
    const double bad = team_reduce(ok ? 0.0 : 1.0, 0, lds + kOffRed);
    return bad == 0.0;
  }

  __device__ __forceinline__ bool build_and_invert(double x) {
    bool ok = build(x);
    ok = sweep() && ok;
    return ok;
  }
  __device__ __forceinline__ bool build_and_solve(double x, double rhs, double* u) {
    const bool ok = build_and_invert(x);
    *u = matvec(rhs);
    return ok;
  }

  // ---- y = T v, one tile row's contributions ----------------------------------------------------------
  template <int NS>
  __device__ __forceinline__ void matvec_row(const d4 (&acc)[NS], const int I) {
    const int ln = fresh_lane(), g = ln >> 4, j = ln & 15;
    double* part = lds + kOffPart;
    const double* vcol = lds + kOffNat + 16 * I + j;  // - 16 k
    const d4 vr = *reinterpret_cast<const d4*>(lds + kOffVperm + ((I * 4 + g) << 2));
    // rows of the tile row: accumulate over its tiles in-lane, then sum over the 16 lanes of a DPP row
    // (transposing butterfly: 4 values -> 1)
    d4 rs = acc[0] * vcol[0];
#pragma unroll
    for (int k = 1; k < NS; ++k) {
      if (k <= I) {
        const double vc = vcol[-16 * k];
#pragma unroll
        for (int r = 0; r < 4; ++r) rs[r] = __builtin_fma(acc[k][r], vc, rs[r]);
      }
    }
    {
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
      // all four lanes of a quad hold the sum for register r = j >> 2
      part[(16 * I + 4 * (j >> 2) + g) * PSTR + 16] = kk;
    }
    // columns of the tiles below the diagonal (the mirrored tiles' rows): sum over r in-lane, over g across
    // the four DPP rows
    double* pcol = part + (16 * I + j) * PSTR + I;  // - 16 k PSTR
#pragma unroll
    for (int k = 1; k < NS; ++k) {
      if (k <= I) {
        double m = acc[k][0] * vr[0];
        m = __builtin_fma(acc[k][1], vr[1], m);
        m = __builtin_fma(acc[k][2], vr[2], m);
        m = __builtin_fma(acc[k][3], vr[3], m);
        m += __shfl_xor(m, 16);
        m += __shfl_xor(m, 32);
        pcol[-16 * k * PSTR] = m;
      }
    }
  }

  __device__ __forceinline__ double matvec(double v) {
    publish_vector(v);
    matvec_row<NSA>(accA, Ia);
    matvec_row<NSB>(accB, Ib);
    __syncthreads();
    double y = 0.0;
    if (tid < DPM) {
      // slots I <= (row's own tile index) are never written and stay zero from kernel start
      const double* src = lds + kOffPart + tid * PSTR;
#pragma unroll
      for (int k = 0; k < PSTR; ++k) y += src[k];
    }
    __syncthreads();
    return tid < dim ? y : 0.0;
  }

  __device__ __forceinline__ double diag() {
    const int ln = fresh_lane(), g = ln >> 4, j = ln & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (j == 4 * r + g) {
        lds[kOffNat + 16 * Ia + 4 * r + g] = accA[0][r];
        lds[kOffNat + 16 * Ib + 4 * r + g] = accB[0][r];
      }
    }
    __syncthreads();
    const double y = (tid < dim) ? lds[kOffNat + tid] : 0.0;
    __syncthreads();
    return y;
  }

  __device__ __forceinline__ double half_vjp_inv(double q) {
    if constexpr (RMETRIC == MM_RMETRIC_RANK1) return matvec(q) / (double)dim;
    else return q * diag();
  }
  // dense metric: grad_quadratic_form_inv(p) = -(M^-1 p)(M^-1 p)^T   (matrices.py:1179-1181)
  __device__ __forceinline__ double dh2_dpos(double p, double q) {
    const double u = matvec(p);
    if constexpr (RMETRIC == MM_RMETRIC_RANK1) {
      const double uq = team_reduce(tid < dim ? u * q : 0.0, 0, lds + kOffRed);
      return -(u * uq) / (double)dim;
    } else {
      return -q * (u * u);
    }
  }
  __device__ __forceinline__ double grad(double q) {
    double* nat = lds + kOffNat;
    if (tid < VLM) nat[tid] = (tid < dim) ? q : 0.0;
    __syncthreads();
    const TargetAux aux = target_prepare<false>(target, nat, dim, tparams, lane);
    const double gr = (tid < dim) ? target_grad_elem<false>(target, aux, nat, tid, dim, tparams) : 0.0;
    __syncthreads();
    return gr;
  }
};

template <int RMETRIC, int W>
__device__ __forceinline__ void run_team_chain(const ImplicitArgs& A, int base_ld, double* lds) {
  const int64_t chain = blockIdx.x;
  const int tid = threadIdx.x, dim = A.dim;
  TeamMfma<RMETRIC, W> bk;
  bk.dim = dim;
  bk.tid = tid;
  bk.lane = tid & 63;
  bk.gq = (tid & 63) >> 4;
  bk.jq = tid & 15;
  bk.target = A.target;
  bk.lds = lds;
  bk.base = A.rparams;
  bk.base_ld = base_ld;
  bk.tparams = A.tparams;
  const bool act = tid < dim;
  double q = act ? A.pos[chain * dim + tid] : 0.0;
  double p = act ? A.mom[chain * dim + tid] : 0.0;
  const double t = signed_step(A.dir, A.step_scale, chain, A.step_size);
  bk.slot(SL_Q) = q;
  bk.slot(SL_P) = p;
  const ChainResult r = implicit_leapfrog_chain(bk, t, mmdev::chain_steps(A.chain_steps, chain, A.n_steps), A.opts);
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

template <int RMETRIC>
__global__ __launch_bounds__(NTHR, 2) void implicit_mfma_team_kernel(ImplicitArgs A, int base_ld) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  for (int i = threadIdx.x; i < DPM * PSTR; i += NTHR) lds[kOffPart + i] = 0.0;  // unused partial-sum slots stay 0
  __syncthreads();
  // every wave executes the same sequence of barriers; only the tile bookkeeping differs
  switch (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) {
    case 0: run_team_chain<RMETRIC, 0>(A, base_ld, lds); break;
    case 1: run_team_chain<RMETRIC, 1>(A, base_ld, lds); break;
    case 2: run_team_chain<RMETRIC, 2>(A, base_ld, lds); break;
    case 3: run_team_chain<RMETRIC, 3>(A, base_ld, lds); break;
    case 4: run_team_chain<RMETRIC, 4>(A, base_ld, lds); break;
    case 5: run_team_chain<RMETRIC, 5>(A, base_ld, lds); break;
    case 6: run_team_chain<RMETRIC, 6>(A, base_ld, lds); break;
    default: run_team_chain<RMETRIC, 7>(A, base_ld, lds); break;
  }
}

template <class K>
int launch_team(mm_ctx* ctx, K kernel, const ImplicitArgs& a, int base_ld) {
  const size_t lds = kTeamLdsDoubles * sizeof(double);
  MM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kernel, dim3((unsigned)a.n_chains), dim3(NTHR), lds, ctx->stream, a, base_ld);
  MM_HIP_CHECK(ctx, hipGetLastError());
  return MM_OK;
}

}  // namespace

int mm_mfma_team_max_dim() { return DPM; }

int mm_launch_implicit_mfma_team(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int n_steps,
                                 const mm_fp_opts& opts, mm_counters* d_counters) {
  if (m->dim > DPM) {
    mm_set_error(ctx, "matrix-core team kernel supports dim <= 256");
    return MM_ERR_UNSUPPORTED;
  }
  if (m->rmetric == MM_RMETRIC_RANK1 && (m->d_rmetric_padded == nullptr || m->rmetric_pad_dim < DPM)) {
    mm_set_error(ctx, "internal: rank-one base matrix was not padded for the team kernels");
    return MM_ERR_UNSUPPORTED;
  }
  ImplicitArgs a{};
  a.pos = s->d_pos;
  a.mom = s->d_mom;
  a.dir = s->d_dir;
  a.step_scale = s->d_step_scale;
  a.chain_steps = s->d_chain_steps;
  a.status = s->d_status;
  a.n_done = s->d_n_done;
  a.n_chains = s->n;
  a.dim = s->dim;
  a.target = m->target;
  a.tparams = m->d_target_params;
  a.rparams = m->d_rmetric_padded;
  a.step_size = h;
  a.n_steps = n_steps;
  a.opts = opts;
  a.counters = d_counters;
  if (m->rmetric == MM_RMETRIC_RANK1)
    return launch_team(ctx, implicit_mfma_team_kernel<MM_RMETRIC_RANK1>, a, m->rmetric_pad_dim);
  return launch_team(ctx, implicit_mfma_team_kernel<MM_RMETRIC_DIAGQUAD>, a, 0);
}
