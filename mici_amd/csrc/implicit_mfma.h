// Device code of the matrix-core wave-per-chain dense-Riemannian kernel (k_implicit_mfma.hip instantiates it for the
// built-in metrics; mm_rtc.hip compiles it at run time around a USER metric, user_metric.h).
//
// Implicit leapfrog on dense-metric Riemannian systems, 32 < D <= 64, one wave per chain, with the
// metric inverted by a BLOCKED symmetric sweep whose rank-4 updates run on the FP64 matrix cores
// (v_mfma_f64_16x16x4_f64).  gfx950 / CDNA4.
//
// Why: at BASELINE c3 (1024 chains, D = 64) there is exactly one chain per SIMD.  A lone wave issues
// about one VALU instruction per 6 cycles (tools/ubench_latency.hip), so the per-column cost of the
// rank-1 sweep of k_implicit.hip (~105 instructions, 64 of them v_fma_f64) is an instruction-issue
// bound, not a flop bound.  One v_mfma_f64_16x16x4 is ONE issue slot for 1024 fused multiply-adds,
// so a rank-4 update of the whole matrix is 10 instructions instead of 256.
//
// Layout.  D is padded to 64 = 4 x 4 tiles of 16 x 16; only the 10 tiles on or below the diagonal are
// stored, each in the MFMA accumulator layout: lane l = 16 g + j, register r holds entry
// (16 I + 4 r + g, 16 J + j) of tile (I, J).  Consequently the four consecutive matrix rows
// K = 16 I0 + 4 r0 + {0,1,2,3} are, for every tile of tile-row I0, exactly register r0 of all 64 lanes
// -- which is precisely the B-operand layout (lane (k = g, n = j)) of the instruction.
//
// Blocked sweep (same algebra as BlockBackend::block_step in k_implicit_large.hip), per block K:
//   (1) publish the panel Q = A[K, :] (4 x 64) to LDS as Qt[c][g]: tile-row I0 directly, the part right
//       of the diagonal from the transposed tiles (I, I0), I > I0, by symmetry;
//   (2) every lane inverts the 4 x 4 pivot block P redundantly (closed form through 2 x 2 Schur
//       complements: two reciprocals instead of four), then lane c turns column c of the panel into
//       column c of  W = P^-1 (Q - E),  E = identity on the K columns (this one modification yields both
//       W_K = I - P^-1 and X_K = P - I of the uniform rank-4 form  A -= W^T X,  X = Q - E);
//   (3) ten MFMAs: tile (I, J) += (-W)[:, tile I]^T  X[:, tile J];  then A_KK -= 2 I.
// After 16 blocks the tiles hold -M^-1.
//
// Reference arithmetic replaced: DensePositiveDefiniteMatrix factorisation + explicit inverse
// (matrices.py:1161-1188) inside ImplicitLeapfrogIntegrator._step (integrators.py:493-544); the step
// logic itself is implicit_core.h (shared with the other backends).
#pragma once
#include "implicit_core.h"
#include "user_metric.h"

namespace mmmfma {

using namespace mmdev;
using namespace mmimp;

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

// Lock-step position solves (implicit_core.h refine_solve2) on this kernel: OFF.  Measured in round 5
// (profiles/r05_ab_c3_dual.txt): the state machine and the paired products are correct (tests/test_gpu_implicit.py green with
// them on), but with 128 registers of inverse row live across the solve most forms of the paired products make the
// allocator move the row into accumulation registers (1.2-1.7 k v_accvgpr_read in the kernel against 120: 1.35e7 -> 1.0e7
// steps/s), and the one form whose allocation survives (-DMM_MFMA_DUAL=1 -DMM_DUAL_UNROLL=4: the second system's vectors in
// LDS, F r as two passes) runs at 1.257e7 with the lock step on against 1.325e7 off.  DESIGN.md section 4.3d.
#ifndef MM_MFMA_DUAL
#define MM_MFMA_DUAL 0
#endif
#ifndef MM_MFMA_WAVES_PER_BLOCK
#define MM_MFMA_WAVES_PER_BLOCK 4
#endif
constexpr int kWaves = MM_MFMA_WAVES_PER_BLOCK;  // chains per workgroup
constexpr int kTiles = 10;     // lower-triangular 16 x 16 tiles of a 64 x 64 matrix
constexpr int kPartStride = 17;
constexpr int kRowPitch = 18;   // doubles per row of the tile -> row conversion buffer (16 columns + 2: 16-byte aligned rows,
                                // conflict-free for the row owners' 16-byte reads)
constexpr int kBasePitch = 66;  // doubles per row of the staged base matrix (64 + 2: the same two properties)
// per-wave LDS (doubles): Qt[64][4], Wt[64][4], nat[64], vperm[64], aux[64], part[64][17], mpart[3][64],
// stash[SL_COUNT_REFINE][64].  The refinement solves (implicit_core.h refine_solve) keep their scratch in Qt / Wt: no
// sweep runs while one is in flight.
constexpr int kMfmaWaveDoubles = 256 + 256 + 64 + 64 + 64 + 64 * kRowPitch + 192 + SL_COUNT_REFINE * 64 + 16;
static_assert((1 + RS_COUNT) * 64 <= 512, "refinement scratch must fit Qt + Wt");
constexpr int kBaseDoubles = 64 * kBasePitch;  // staged base matrix of the rank-one metric: full rows, zero padded

__host__ __device__ constexpr int tix(int I, int J) { return I * (I + 1) / 2 + J; }

struct MLds {
  double* qt;     // [64][4] panel, column-major in the block index
  double* wt;     // [64][4] -W
  double* nat;    // [64] natural-order vector
  double* vperm;  // [4][4][4] = [I][g][r] copy of a vector for row operands
  double* aux;    // [64]
  double* part;   // [64][18] the tile -> row conversion buffer; [64][17] partial sums of the factored solve
  double* mpart;  // [3][4][16] mirrored partial sums
  double* stash;  // [SL_COUNT_REFINE][64]
  double* prof;   // [16] developer builds: phase clocks (implicit_core.h PH_*)
  // user metrics (user_metric.h): the point of the held inverse / of the refinement products in natural order, their aux
  double* uq;
  double* ux;
  double* uaq;
  double* uax;
};
template <int RMETRIC>
__host__ __device__ constexpr int mfma_wave_doubles() {
  return kMfmaWaveDoubles + (RMETRIC == MM_RMETRIC_USER ? mmuser::lds_doubles(64) : 0);
}

template <int RMETRIC, bool PROFILE = false, bool LOWRANK = false>
struct MfmaBackend {
  static constexpr bool kProf = PROFILE;  // developer builds: cycles per phase of the step
  // implicit_core.h lowrank_solve (round 6): the rank-one-update metric's solve-only constructions by the Woodbury identity
  // from the held inverse - one row product F d each instead of ~3 CG pairs
  // (the built-in rank-one-update metric, or a user metric that declares the structure: user_metric.h MM_USER_LOWRANK)
  static constexpr bool kLowRankBuiltin = RMETRIC == MM_RMETRIC_RANK1;
  static constexpr bool kLowRank = LOWRANK && (kLowRankBuiltin || (RMETRIC == MM_RMETRIC_USER && mmuser::kLowRank));
  __device__ __forceinline__ double lowrank_scale() const {
    if constexpr (kLowRankBuiltin) return (double)dim;
    else return mmuser::lowrank_inv_s(dim, uparams);
  }
  // u(x), this lane's element (user metric: the point published for the hook, its aux block prepared - a wave collective)
  __device__ __forceinline__ double lowrank_vec(double x) {
    if constexpr (kLowRankBuiltin) {
      return x;
    } else {
      w.ux[lane] = (lane < dim) ? x : 0.0;
      wave_sync();
      mmuser::prepare(mmuser::WaveTeam{lane}, w.ux, dim, uparams, w.uax);
      wave_sync();
      const double u = mmuser::lowrank_u(w.ux, lane, dim, uparams, w.uax);
      wave_sync();  // (the next point overwrites w.ux / w.uax)
      return lane < dim ? u : 0.0;
    }
  }
  __device__ __forceinline__ double& lowrank_u0() {
    if constexpr (kLowRankBuiltin) return slot(SL_Q);
    else return rslot(LR_U0);
  }
  // a user metric's hooks evaluate its vector-Jacobian products at "the point of the held inverse" (w.uq, w.uaq - build() sets
  // them): an inverse carried to x by lowrank_update takes the point with it
  __device__ __forceinline__ void held_point(double x) {
    if constexpr (RMETRIC == MM_RMETRIC_USER) {
      w.uq[lane] = (lane < dim) ? x : 0.0;
      wave_sync();
      mmuser::prepare(mmuser::WaveTeam{lane}, w.uq, dim, uparams, w.uaq);
      wave_sync();
    }
  }
  __device__ static constexpr bool lowrank_on() { return true; }  // (compile-time: the launcher picks the instantiation)
  int lr_refresh_;
  __device__ __forceinline__ int lowrank_refresh() const { return lr_refresh_; }
  __device__ __forceinline__ int prof_switch(int phase) {
    int old = 0;
    if (lane == 0) {
      const double now = (double)__builtin_readcyclecounter();
      old = (int)w.prof[PH_COUNT];
      w.prof[old] += now - w.prof[PH_COUNT + 1];
      w.prof[PH_COUNT] = (double)phase;
      w.prof[PH_COUNT + 1] = now;
    }
    return __builtin_amdgcn_readfirstlane(old);
  }
  static constexpr bool kSolveByInverse = false;
  static constexpr bool kUnifiedConstruct = true;  // implicit_core.h: one construction site, the mode at run time
  static constexpr bool kCountersInLds = false;
  static constexpr bool kRefine = true;  // implicit_core.h: solve-only constructions refined from the held inverse
  bool refine_on;                        // false: MICI_AMD_REFINE=0, every construction is factorised
  // implicit_core.h refine_solve2: the two position solves of a step in lock step (round 5).  Built-in metrics; a user
  // metric's M(x) is forty registers of entries per point, two points do not fit next to the inverse's row.
  // (round 6: ON in the Woodbury kernel of the built-in metric - lowrank_solve2 advances the reversibility-check solve and the
  // C-adjoint solve together on interleaved products, matvec2_exact; MICI_AMD_DUAL=0: one after the other)
  static constexpr bool kDual = MM_MFMA_DUAL ? RMETRIC != MM_RMETRIC_USER : (LOWRANK && RMETRIC == MM_RMETRIC_RANK1);
  bool dual_off;                         // true: MICI_AMD_DUAL=0, one solve after the other
  d4 acc[kTiles];
  int dim, lane, target;
  MLds w;
  const double* base_lds;
  const double* tparams;
  const double* uparams;  // user metric: its params (global memory)
  double* work;           // user metric with the dense-accessor VJP: this chain's 64 x 64 doubles of global memory

  // (the step's slots in registers instead of LDS - there is room since round 4 - were measured and lose: 1.17e7 against
  // 1.24e7 steps/s on c3)
#ifndef MM_MFMA_LR_SLOT_REGS
#define MM_MFMA_LR_SLOT_REGS 0
#endif
  // (round 6, the Woodbury kernel: without the CG vectors and the M(x) v operands there is room again - A/B macro)
  static constexpr bool kSlotRegs = kLowRank && MM_MFMA_LR_SLOT_REGS;
  double st_[kSlotRegs ? SL_COUNT_REFINE : 1];
  __device__ __forceinline__ double& slot(int i) {
    if constexpr (kSlotRegs) return st_[i];
    else return w.stash[i * 64 + lane];
  }
  __device__ __forceinline__ bool flat_active() const { return lane < dim; }

  // ---- metric_func(x) into the tiles; false if an entry is not finite ----------------------------
  __device__ __forceinline__ bool build(double x) {
    const int g = lane >> 4, j = lane & 15;
    const double xm = (lane < dim) ? x : 0.0;
    w.nat[lane] = xm;
    w.vperm[(((lane >> 4) * 4 + (lane & 3)) << 2) + ((lane >> 2) & 3)] = xm;
    if constexpr (RMETRIC == MM_RMETRIC_USER) {  // the point in natural order for the user's hooks, then its aux block
      w.uq[lane] = xm;
      wave_sync();
      mmuser::prepare(mmuser::WaveTeam{lane}, w.uq, dim, uparams, w.uaq);
    }
    wave_sync();
    const double inv_d = 1.0 / (double)dim;
    double qc[4];
    d4 qr[4];
#pragma unroll
    for (int X = 0; X < 4; ++X) {
      qc[X] = w.nat[16 * X + j];
      qr[X] = *reinterpret_cast<const d4*>(w.vperm + ((X * 4 + g) << 2));
    }
#pragma unroll
    for (int I = 0; I < 4; ++I)
#pragma unroll
      for (int J = 0; J <= I; ++J) {
        const int t = tix(I, J);
        if constexpr (RMETRIC == MM_RMETRIC_RANK1) {
          const double* bt = base_lds + (16 * I + g) * kBasePitch + 16 * J + j;  // B[16 I + 4 r + g][16 J + j]
          const double qs = qc[J] * inv_d;
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[t][r] = __builtin_fma(qr[I][r], qs, bt[4 * r * kBasePitch]);
        } else if constexpr (RMETRIC == MM_RMETRIC_USER) {
          // the user's metric_func, entry by entry (zero on the padding; its diagonal is set to 1 below); the lane index
          // laundered so that the entries' addresses are computed here, not hoisted out of the step loop
          int og = g, oj = j;
          asm volatile("" : "+v"(og), "+v"(oj));
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = 16 * I + 4 * r + og, jj = 16 * J + oj;
            acc[t][r] = mmuser::entry_padded(w.uq, i, jj, dim, uparams, w.uaq);
          }
        } else {
          acc[t] = d4{0.0, 0.0, 0.0, 0.0};
        }
      }
    // diagonal entries: (16 I + 4 r + g, same) <-> tile (I, I), register r, lanes with j == 4 r + g
    double chk = 0.0;
    if constexpr (RMETRIC == MM_RMETRIC_USER) {  // "Array is not finite.": nothing is known about where a user metric is largest
#pragma unroll
      for (int t = 0; t < kTiles; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) chk = __builtin_fma(acc[t][r], 0.0, chk);
    }
#pragma unroll
    for (int I = 0; I < 4; ++I) {
      const int t = tix(I, I);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool on_diag = (j == 4 * r + g);
        if constexpr (RMETRIC == MM_RMETRIC_DIAGQUAD) {
          const double qi = qr[I][r];
          if (on_diag) acc[t][r] = __builtin_fma(qi, qi, 1.0);
        }
        if (on_diag && 16 * I + 4 * r + g >= dim) acc[t][r] = 1.0;  // identity on the padding
        // "Array is not finite." (matrices.py:211-215).  Both built-in metrics have their largest
        // entries on the diagonal (B_ii + q_i^2 / D with B_ii > 0; 1 + q_i^2), so a non-finite entry
        // anywhere implies a non-finite diagonal-tile entry.
        chk = __builtin_fma(acc[t][r], 0.0, chk);
      }
    }
    wave_sync();
    return __all(chk == 0.0);
  }

  // ---- refinement solves (implicit_core.h): M(x) v formed matrix-free, the tiles keep M(x0)^-1 ---------------------
  // the solve's flat vectors (u, r, d) stay in registers: nothing else of the step is live during a refinement solve
  // (the sweeps' operands are dead), and a lone wave pays ~100 cycles for every dependent LDS access
#ifndef MM_DUAL_RS_LDS
#define MM_DUAL_RS_LDS 1
#endif
#if MM_DUAL_RS_LDS
  // the second system of a lock-step pair keeps its three vectors in LDS (the dead W buffer of the sweeps): with them in
  // registers next to the inverse's row the allocator moves the row to accumulation registers
  double rs_[RS_COUNT];
  __device__ __forceinline__ double& rslot(int i) { return i < RS_COUNT ? rs_[i] : w.wt[(i - RS_COUNT) * 64 + lane]; }
#else
  double rs_[2 * RS_COUNT];  // (the second system of a lock-step pair in [RS_COUNT ..])
  __device__ __forceinline__ double& rslot(int i) { return rs_[i]; }
#endif
  __device__ __forceinline__ void sum2(double a, double b, double* sa, double* sb) {
    *sa = wave_sum(lane < dim ? a : 0.0);
    *sb = wave_sum(lane < dim ? b : 0.0);
  }
  // sums of the lock-step pair: independent dependent chains, interleaved by the scheduler
  __device__ __forceinline__ void sum2x(double a, double b, double* sa, double* sb) { sum2(a, b, sa, sb); }
  __device__ __forceinline__ void sum4(double a, double b, double c, double d, double* sa, double* sb, double* sc,
                                       double* sd) {
    sum2(a, b, sa, sb);
    sum2(c, d, sc, sd);
  }
  // (two independent DPP chains: the scheduler interleaves them)
  __device__ __forceinline__ void norm_dot(double x, int kind, double y, double* err, double* s) {
    const double a = wave_norm_accum(0.0, lane < dim ? x : 0.0, kind);
    *s = wave_sum(lane < dim ? y : 0.0);
    *err = wave_norm_finish(a, kind);
  }
  __device__ __forceinline__ void sum3(double a, double b, double c, double* sa, double* sb, double* sc) {
    sum2(a, b, sa, sb);
    *sc = wave_sum(lane < dim ? c : 0.0);
  }
  __device__ __forceinline__ void norm2(double a, double b, int kind, double* na, double* nb) {
    const double xa = wave_norm_accum(0.0, lane < dim ? a : 0.0, kind), xb = wave_norm_accum(0.0, lane < dim ? b : 0.0, kind);
    *na = wave_norm_finish(xa, kind);
    *nb = wave_norm_finish(xb, kind);
  }
  __device__ __forceinline__ double sum1(double a) { return wave_sum(lane < dim ? a : 0.0); }
  // M(x) v in the form that suits the metric:  rank-one update  B v + x (x . v) / D  (B's tiles from LDS, contracted
  // like matvec() contracts the register tiles);  diag(1 + x^2): per lane
  // ---- products in ROW form (round 4) -------------------------------------------------------------------------------
  // A step applies the held inverse ~40 times (29 refinement pairs, the momentum solves, the half steps) and sweeps it
  // once.  In the tile layout of the sweep a mat-vec is 64 multiply-adds a lane wrapped in two LDS round trips - operands
  // out, 19 partial sums per lane back through LDS and a summation tree (1.4 k cycles, a fifth of them arithmetic).  After
  // the sweep the inverse is therefore re-laid out ONCE, lane i taking row i (tiles_to_rows: 64 LDS stores and 32 16-byte
  // loads a lane); a product is then the vector broadcast from LDS (sixteen 16-byte reads of one address) against 64
  // registers: no partial sums, no reduction, the result already flat.  M(x) v of the refinement solves takes row i of
  // the staged base matrix the same way (rank-one metric), or of the user's metric evaluated once per solve.
  // 1 / D of the rank-one metric's q q^T / D: multiplied with, not divided by - the IEEE division's ~30 dependent
  // instructions sat on the critical path of every M(x) v and every momentum iteration
  double inv_dim_;
  double fr_[64];  // row `lane` of M(x0)^-1 (the padding is the identity: zero off the diagonal)
  double fd_;      // its diagonal entry

  // user metric: the ten lower tiles of M(x) at the products' point (tile layout of the sweep: 40 entries a lane, the
  // symmetric half - a row would be 64, and its evaluation, once per refinement solve, is what a user metric pays for)
  d4 mx_[RMETRIC == MM_RMETRIC_USER ? kTiles : 1];
  // (converted to a row per refinement solve - tiles_to_row, then the reduction-free row product - it loses: 80 + 128
  // more registers next to the inverse's row spill; c3_user 4.16e6 against 4.57e6 steps/s)

  // lower tiles of a symmetric matrix (the sweep's layout) -> row `lane` of the full matrix, sixteen columns at a time
  __device__ __forceinline__ void tiles_to_row(const d4 (&t)[kTiles], double (&row)[64], double* diag_out) {
    const int g = lane >> 4, j = lane & 15;
    double* buf = w.part;  // [64][kRowPitch]
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int I = c; I < 4; ++I)  // tile (I, c): rows 16 I + 4 r + g, column 16 c + j
#pragma unroll
        for (int r = 0; r < 4; ++r) buf[(16 * I + 4 * r + g) * kRowPitch + j] = t[tix(I, c)][r];
#pragma unroll
      for (int J = 0; J < c; ++J)  // tile (c, J) by symmetry: row 16 J + j, column 16 c + 4 r + g
#pragma unroll
        for (int r = 0; r < 4; ++r) buf[(16 * J + j) * kRowPitch + 4 * r + g] = t[tix(c, J)][r];
      wave_sync();
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const d2 x = *reinterpret_cast<const d2*>(buf + lane * kRowPitch + 2 * k);
        row[16 * c + 2 * k] = x[0];
        row[16 * c + 2 * k + 1] = x[1];
      }
      if (diag_out && (lane >> 4) == c) *diag_out = buf[lane * kRowPitch + (lane & 15)];
      wave_sync();
    }
  }
  __device__ __forceinline__ void tiles_to_rows() { tiles_to_row(acc, fr_, &fd_); }

  // independent accumulator chains of a row product: four for the held inverse (registers x broadcast vector), two for
  // M(x) v (whose 32 row loads, not the multiply-adds, pace it) - measured on c3 (tools/ab_build.py grid, steps/s):
  // F4/M4 1.25e7, F4/M2 1.316e7, F4/M1 1.314e7, F8/M2 1.308e7, F2/M2 1.23e7, F16/M4 1.23e7
  static constexpr int kAcc = 4;
  static constexpr int kAccM = 2;
  static_assert(kAccM == 1 || kAccM == 2 || kAccM == 4, "metric_apply2 assigns column 4 k + e to chain e % kAccM");
  // y_lane = sum_j row[j] v_j with v broadcast from LDS (w.nat, zero beyond dim)
  __device__ __forceinline__ double row_dot(const double (&row)[64], double v) {
    w.nat[lane] = (lane < dim) ? v : 0.0;
    wave_sync();
    double y[kAcc];
#pragma unroll
    for (int a = 0; a < kAcc; ++a) y[a] = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const d4 vv = *reinterpret_cast<const d4*>(w.nat + 4 * k);
#pragma unroll
      for (int e = 0; e < 4; ++e) y[(4 * k + e) % kAcc] = __builtin_fma(row[4 * k + e], vv[e], y[(4 * k + e) % kAcc]);
    }
    wave_sync();  // (the next product overwrites w.nat)
#pragma unroll
    for (int h = kAcc / 2; h >= 1; h >>= 1)
#pragma unroll
      for (int a = 0; a < h; ++a) y[a] += y[a + h];
    return y[0];
  }

  __device__ __forceinline__ void matvec2(double v0, double v1, double* y0, double* y1) {
    // the held inverse's row is in registers: nothing to share between the two products but the exchange points, and a
    // row product already runs kAcc independent chains.  ONE inlined product in a two-trip loop, operands and results
    // through LDS: a second inlined copy is what tips the allocator (see refine_solve2)
    w.wt[192 + lane] = v0;
    w.vperm[lane] = v1;
    double a = 0.0, b = 0.0;
#pragma unroll 1
    for (int sidx = 0; sidx < 2; ++sidx) {
      const double v = sidx == 0 ? w.wt[192 + lane] : w.vperm[lane];
      const double y = row_dot(fr_, v);
      if (sidx == 0) a = y;
      else b = y;
    }
    *y0 = lane < dim ? a : 0.0;
    *y1 = lane < dim ? b : 0.0;
  }

  // the points of a lock-step pair: system 0's where metric_point() puts it, system 1's behind it
  // implicit_core.h lowrank_solve2 (round 6): the two position solves' products F d_C, F d_A INTERLEAVED - every register of the
  // inverse's row feeds two multiply-adds with independent accumulators, the two broadcast vectors are read side by side: a
  // lone wave's dependent chains get a second stream to fill their latency with (the Woodbury kernel has no CG vectors and no
  // M(x) v operands next to the row: the registers the lock step of section 4.3d lacked)
  __device__ __forceinline__ void matvec2_exact(double v0, double v1, double* y0, double* y1) {
    w.nat[lane] = (lane < dim) ? v0 : 0.0;
    w.aux[lane] = (lane < dim) ? v1 : 0.0;
    wave_sync();
    double a[kAcc], b[kAcc];
#pragma unroll
    for (int e = 0; e < kAcc; ++e) a[e] = b[e] = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const d4 va = *reinterpret_cast<const d4*>(w.nat + 4 * k), vb = *reinterpret_cast<const d4*>(w.aux + 4 * k);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        a[(4 * k + e) % kAcc] = __builtin_fma(fr_[4 * k + e], va[e], a[(4 * k + e) % kAcc]);
        b[(4 * k + e) % kAcc] = __builtin_fma(fr_[4 * k + e], vb[e], b[(4 * k + e) % kAcc]);
      }
    }
    wave_sync();  // (the next product overwrites w.nat / w.aux)
#pragma unroll
    for (int h = kAcc / 2; h >= 1; h >>= 1)
#pragma unroll
      for (int e = 0; e < h; ++e) {
        a[e] += a[e + h];
        b[e] += b[e + h];
      }
    *y0 = lane < dim ? a[0] : 0.0;
    *y1 = lane < dim ? b[0] : 0.0;
  }
  __device__ __forceinline__ void metric_point2(double x0, double x1) {
    w.qt[lane] = (lane < dim) ? x0 : 0.0;
    w.qt[64 + lane] = (lane < dim) ? x1 : 0.0;
  }
  // M(x0) v0 and M(x1) v1: ONE pass over the staged base matrix' row (the LDS volume that paces metric_apply) for both
  __device__ __forceinline__ void metric_apply2(double v0, double v1, double* y0, double* y1) {
    const double x0 = w.qt[lane], x1 = w.qt[64 + lane];
    if constexpr (RMETRIC == MM_RMETRIC_DIAGQUAD) {
      *y0 = lane < dim ? __builtin_fma(x0 * x0, v0, v0) : 0.0;
      *y1 = lane < dim ? __builtin_fma(x1 * x1, v1, v1) : 0.0;
    } else if constexpr (RMETRIC == MM_RMETRIC_RANK1) {
      w.nat[lane] = (lane < dim) ? v0 : 0.0;
      w.aux[lane] = (lane < dim) ? v1 : 0.0;
      wave_sync();
      const double* brow = base_lds + lane * kBasePitch;
      double ya[kAccM], yb[kAccM];
#pragma unroll
      for (int a = 0; a < kAccM; ++a) ya[a] = yb[a] = 0.0;
      // A REAL loop (kDualUnroll columns-of-four a trip): fully unrolled, the scheduler hoists all 64 loads above the
      // arithmetic - next to the inverse's row (128 registers, live across this product) they land in accumulation
      // registers and come back through 240 v_accvgpr_read.  Nothing here indexes a register array, so nothing needs
      // the unrolling.
#ifndef MM_DUAL_UNROLL
#define MM_DUAL_UNROLL 4
#endif
#pragma unroll MM_DUAL_UNROLL
      for (int k = 0; k < 16; ++k) {
        const d4 xa = *reinterpret_cast<const d4*>(w.nat + 4 * k);
        const d4 xb = *reinterpret_cast<const d4*>(w.aux + 4 * k);
        const d2 b01 = *reinterpret_cast<const d2*>(brow + 4 * k);
        const d2 b23 = *reinterpret_cast<const d2*>(brow + 4 * k + 2);
        // (4 k, 4 k + 2 -> chain 0, 4 k + 1, 4 k + 3 -> chain 1: metric_apply()'s assignment for kAccM = 2)
        ya[0] = __builtin_fma(b01[0], xa[0], ya[0]);
        yb[0] = __builtin_fma(b01[0], xb[0], yb[0]);
        ya[1 % kAccM] = __builtin_fma(b01[1], xa[1], ya[1 % kAccM]);
        yb[1 % kAccM] = __builtin_fma(b01[1], xb[1], yb[1 % kAccM]);
        ya[2 % kAccM] = __builtin_fma(b23[0], xa[2], ya[2 % kAccM]);
        yb[2 % kAccM] = __builtin_fma(b23[0], xb[2], yb[2 % kAccM]);
        ya[3 % kAccM] = __builtin_fma(b23[1], xa[3], ya[3 % kAccM]);
        yb[3 % kAccM] = __builtin_fma(b23[1], xb[3], yb[3 % kAccM]);
      }
      const double dot0 = wave_sum(lane < dim ? x0 * v0 : 0.0);
      const double dot1 = wave_sum(lane < dim ? x1 * v1 : 0.0);
      wave_sync();
#pragma unroll
      for (int h = kAccM / 2; h >= 1; h >>= 1)
#pragma unroll
        for (int a = 0; a < h; ++a) {
          ya[a] += ya[a + h];
          yb[a] += yb[a + h];
        }
      const double r0 = __builtin_fma(x0, dot0 * inv_dim_, ya[0]), r1 = __builtin_fma(x1, dot1 * inv_dim_, yb[0]);
      *y0 = lane < dim ? r0 : 0.0;
      *y1 = lane < dim ? r1 : 0.0;
    } else {
      *y0 = *y1 = 0.0;  // (user metrics: kDual is false)
    }
  }

  __device__ __forceinline__ void metric_point(double x) {
    w.qt[lane] = (lane < dim) ? x : 0.0;
    if constexpr (RMETRIC == MM_RMETRIC_USER) {
      // the products' point in natural order, its aux block, then the tiles of the user's metric_func there - ONCE per
      // refinement solve (its 2 - 8 products are all at that point), dead again before anything else of the step runs
      w.ux[lane] = (lane < dim) ? x : 0.0;
      wave_sync();
      mmuser::prepare(mmuser::WaveTeam{lane}, w.ux, dim, uparams, w.uax);
      wave_sync();
      // (the lane index laundered: everything the entries' addresses derive from is loop invariant, and hoisted out of the
      // step loop they lived in scratch for the whole kernel)
      int og = lane >> 4, oj = lane & 15;
      asm volatile("" : "+v"(og), "+v"(oj));
#pragma unroll
      for (int I = 0; I < 4; ++I)
#pragma unroll
        for (int J = 0; J <= I; ++J)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            mx_[tix(I, J)][r] = mmuser::entry_padded(w.ux, 16 * I + 4 * r + og, 16 * J + oj, dim, uparams, w.uax);
    }
  }

  // sixteen partial sums of a row, pairwise: a lone wave pays every dependent add in full (a serial chain is 16 deep)
  __device__ static __forceinline__ double sum16(const double* src) {
    double a[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = src[k];
#pragma unroll
    for (int h = 8; h >= 1; h >>= 1)
#pragma unroll
      for (int k = 0; k < h; ++k) a[k] += a[k + h];
    return a[0];
  }

  // y = T v for a symmetric matrix held as lower tiles in the sweep's layout (the user metric's M(x)): operands out
  // through LDS, 19 partial sums a lane back through LDS
  __device__ __forceinline__ double tile_matvec(const d4 (&m)[kTiles], double v) {
    const int g = lane >> 4, j = lane & 15;
    w.nat[lane] = (lane < dim) ? v : 0.0;
    w.vperm[(((lane >> 4) * 4 + (lane & 3)) << 2) + ((lane >> 2) & 3)] = (lane < dim) ? v : 0.0;
    wave_sync();
    double vc[4];
    d4 vr[4];
#pragma unroll
    for (int X = 0; X < 4; ++X) {
      vc[X] = w.nat[16 * X + j];
      vr[X] = *reinterpret_cast<const d4*>(w.vperm + ((X * 4 + g) << 2));
    }
    double mir[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int I = 0; I < 4; ++I) {
      d4 sr = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int J = 0; J <= I; ++J) {
        const int t = tix(I, J);
        if (I != J) {
#pragma unroll
          for (int r = 0; r < 4; ++r) mir[J] = __builtin_fma(m[t][r], vr[I][r], mir[J]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) sr[r] = __builtin_fma(m[t][r], vc[J], sr[r]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) w.part[(16 * I + 4 * r + g) * kPartStride + j] = sr[r];
    }
#pragma unroll
    for (int J = 0; J < 3; ++J) w.mpart[(J * 4 + g) * 16 + j] = mir[J];
    wave_sync();
    double y = sum16(w.part + lane * kPartStride);
    if (lane < 48) {
      const double* mp = w.mpart + (lane >> 4) * 64 + (lane & 15);
      y += (mp[0] + mp[16]) + (mp[32] + mp[48]);
    }
    wave_sync();
    return lane < dim ? y : 0.0;
  }

  __device__ __forceinline__ double metric_apply(double v) {
    const double x = w.qt[lane];
    if constexpr (RMETRIC == MM_RMETRIC_DIAGQUAD) {
      return lane < dim ? __builtin_fma(x * x, v, v) : 0.0;
    } else if constexpr (RMETRIC == MM_RMETRIC_USER) {
      return tile_matvec(mx_, v);
    } else {
      // rank-one update: B v from row `lane` of the staged base matrix (16-byte reads, conflict-free at a pitch of 66
      // doubles; zero on the padding) + x (x . v) / D.  (The row in registers between sweeps - 128 more next to the
      // inverse's 128 - was measured and loses: beyond 256 architected registers the allocator parks operands in
      // accumulation registers, and every multiply-add then pays two v_accvgpr_read: c3 1.09e7 against 1.19e7.  Its leading
      // 16 / 32 / 48 columns only: 1.29 / 1.30 / 1.12e7 against 1.32e7.)
      w.nat[lane] = (lane < dim) ? v : 0.0;
      wave_sync();
      const double* brow = base_lds + lane * kBasePitch;
      double ya[kAccM];
#pragma unroll
      for (int a = 0; a < kAccM; ++a) ya[a] = 0.0;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const d4 vv = *reinterpret_cast<const d4*>(w.nat + 4 * k);
        const d2 b01 = *reinterpret_cast<const d2*>(brow + 4 * k);
        const d2 b23 = *reinterpret_cast<const d2*>(brow + 4 * k + 2);
        ya[(4 * k) % kAccM] = __builtin_fma(b01[0], vv[0], ya[(4 * k) % kAccM]);
        ya[(4 * k + 1) % kAccM] = __builtin_fma(b01[1], vv[1], ya[(4 * k + 1) % kAccM]);
        ya[(4 * k + 2) % kAccM] = __builtin_fma(b23[0], vv[2], ya[(4 * k + 2) % kAccM]);
        ya[(4 * k + 3) % kAccM] = __builtin_fma(b23[1], vv[3], ya[(4 * k + 3) % kAccM]);
      }
      const double dot = wave_sum(lane < dim ? x * v : 0.0);
      wave_sync();
#pragma unroll
      for (int h = kAccM / 2; h >= 1; h >>= 1)
#pragma unroll
        for (int a = 0; a < h; ++a) ya[a] += ya[a + h];
      const double y = __builtin_fma(x, dot * inv_dim_, ya[0]);
      return lane < dim ? y : 0.0;
    }
  }

  // operands of one block's rank-4 update.  They stay in registers after the block so that the six
  // "cold" tiles (those the NEXT block's panel does not read) are updated while the next block's scalar
  // work (P^-1, W) is being issued: the matrix core runs asynchronously to the VALU.
  struct Ops {
    double av[4], bv[4];
  };

  // update the tiles of tile-row/column `in` (hot == true) or all the others (hot == false); in < 0: all.
  // jmin: the trailing (LDL^T) sweep leaves the tile columns left of the pivot's alone.
  __device__ __forceinline__ void apply(const Ops& o, const int in, const bool hot, const int jmin = 0) {
#pragma unroll
    for (int I = 0; I < 4; ++I)
#pragma unroll
      for (int J = 0; J <= I; ++J) {
        const bool is_hot = (I == in) || (J == in);
        if (J >= jmin && (in < 0 || is_hot == hot))
          acc[tix(I, J)] = __builtin_amdgcn_mfma_f64_16x16x4f64(o.av[I], o.bv[J], acc[tix(I, J)], 0, 0, 0);
      }
  }

  // ---- one block of the sweep (B compile-time after unrolling) ----------------------------------------
  // TRAILING: only the tiles (I, J) with J >= I0 are updated - a blocked LDL^T (D^3/3 flops instead of D^3) that
  // ends with tile (K, K) = -P_K^-1 and tile (I, K) = A_IK P_K^-1, the factors solve_factored() substitutes with
  template <bool TRAILING>
  __device__ __forceinline__ void block_step(const int B, Ops& ops, double& pmin) {
    const int I0 = B >> 2, R0 = B & 3;
    // (the lane index laundered per block: the LDS addresses derived from it are a few instructions to recompute, and
    // kept live across the whole kernel - with the row-form inverse next to them - they ended up in scratch)
    int lane = this->lane;
    asm volatile("" : "+v"(lane));
    const int g = lane >> 4, j = lane & 15;
    const int k0 = 16 * I0 + 4 * R0;
    // (1) publish rows K of the matrix as Qt[c][s] = A[k0 + s][c].  The transposed part comes from the 16
    // lanes holding columns K of the tiles below the diagonal tile; the other lanes store into the (dead)
    // W buffer instead of branching, which keeps the whole sweep one basic block for the scheduler.
#pragma unroll
    for (int J = TRAILING ? I0 : 0; J <= I0; ++J) w.qt[((16 * J + j) << 2) + g] = acc[tix(I0, J)][R0];
    {
      double* dst = ((j >> 2) == R0) ? w.qt + (g << 2) + (j & 3) : w.wt + j;
#pragma unroll
      for (int I = I0 + 1; I < 4; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[(16 * I + 4 * r) << 2] = acc[tix(I, I0)][r];
    }
    wave_sync();
    // (2) P^-1 (uniform) and this lane's column of W
    const d4 c0 = *reinterpret_cast<const d4*>(w.qt + ((k0 + 0) << 2));
    const d4 c1 = *reinterpret_cast<const d4*>(w.qt + ((k0 + 1) << 2));
    const d4 c2 = *reinterpret_cast<const d4*>(w.qt + ((k0 + 2) << 2));
    const d4 c3 = *reinterpret_cast<const d4*>(w.qt + ((k0 + 3) << 2));
    d4 q = *reinterpret_cast<const d4*>(w.qt + (lane << 2));
    // the previous block's cold tiles: independent of everything below until this block's own update
    if (B > 0) apply(ops, I0, false, TRAILING ? (B - 1) >> 2 : 0);
    d4 wv;
    {
      // P = [A B; B^T C] with 2 x 2 blocks; c_s is column s of P
      const double a = c0[0], b = c0[1], e = c1[1];                    // A = [a b; b e]
      const double b00 = c0[2], b01 = c0[3], b10 = c1[2], b11 = c1[3];  // B = P[0:2, 2:4]
      const double h = c2[2], i2 = c2[3], jj = c3[3];                  // C = [h i2; i2 jj]
      const double det_a = __builtin_fma(a, e, -b * b);
      const double ida = fast_rcp(det_a);
      const double ia00 = e * ida, ia01 = -b * ida, ia11 = a * ida;  // A^-1
      // T = A^-1 B
      const double t00 = __builtin_fma(ia00, b00, ia01 * b10), t01 = __builtin_fma(ia00, b01, ia01 * b11);
      const double t10 = __builtin_fma(ia01, b00, ia11 * b10), t11 = __builtin_fma(ia01, b01, ia11 * b11);
      // S = C - B^T T
      const double s00 = h - __builtin_fma(b00, t00, b10 * t10);
      const double s01 = i2 - __builtin_fma(b00, t01, b10 * t11);
      const double s11 = jj - __builtin_fma(b01, t01, b11 * t11);
      const double det_s = __builtin_fma(s00, s11, -s01 * s01);
      const double ids = fast_rcp(det_s);
      const double is00 = s11 * ids, is01 = -s01 * ids, is11 = s00 * ids;  // S^-1
      // pivots of the sequential elimination: a, det_a / a, s00, det_s / s00  (all must be > 0)
      // v_min_f64 drops NaNs, so a NaN pivot is caught separately at the end of the sweep: it poisons
      // every entry of W and with it every tile
      pmin = __builtin_fmin(__builtin_fmin(pmin, a), __builtin_fmin(det_a, __builtin_fmin(s00, det_s)));
      // U = T S^-1;  P^-1 = [A^-1 + U T^T, -U; -U^T, S^-1]
      const double u00 = __builtin_fma(t00, is00, t01 * is01), u01 = __builtin_fma(t00, is01, t01 * is11);
      const double u10 = __builtin_fma(t10, is00, t11 * is01), u11 = __builtin_fma(t10, is01, t11 * is11);
      const double p00 = ia00 + __builtin_fma(u00, t00, u01 * t01);
      const double p01 = ia01 + __builtin_fma(u00, t10, u01 * t11);
      const double p11 = ia11 + __builtin_fma(u10, t10, u11 * t11);
      // this lane's column of the panel, minus the identity on the block's own columns
      const int s = lane - k0;
      q[0] -= (s == 0) ? 1.0 : 0.0;
      q[1] -= (s == 1) ? 1.0 : 0.0;
      q[2] -= (s == 2) ? 1.0 : 0.0;
      q[3] -= (s == 3) ? 1.0 : 0.0;
      // -W[:, c] = -P^-1 q
      // (one multiply and three fused multiply-adds per component; the signs ride on operand modifiers)
      wv[0] = __builtin_fma(-p00, q[0], __builtin_fma(-p01, q[1], __builtin_fma(u00, q[2], u01 * q[3])));
      wv[1] = __builtin_fma(-p01, q[0], __builtin_fma(-p11, q[1], __builtin_fma(u10, q[2], u11 * q[3])));
      wv[2] = __builtin_fma(u00, q[0], __builtin_fma(u10, q[1], __builtin_fma(-is00, q[2], -is01 * q[3])));
      wv[3] = __builtin_fma(u01, q[0], __builtin_fma(u11, q[1], __builtin_fma(-is01, q[2], -is11 * q[3])));
    }
    *reinterpret_cast<d4*>(w.qt + (lane << 2)) = q;  // only the block's own four columns changed
    *reinterpret_cast<d4*>(w.wt + (lane << 2)) = wv;
    if (B > 0) {
      // scheduling pipeline for this region: one cold MFMA per ~15 VALU instructions (an MFMA occupies the
      // matrix core for 64 cycles, a lone wave issues a VALU instruction about every 6)
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 15, 0);
      }
    }
    wave_sync();
    // (3) rank-4 update on the matrix cores: now only the tiles the next block's panel reads
#pragma unroll
    for (int X = 0; X < 4; ++X) {
      ops.av[X] = w.wt[((16 * X + j) << 2) + g];
      ops.bv[X] = w.qt[((16 * X + j) << 2) + g];
    }
    if (B < 15) apply(ops, (B + 1) >> 2, true, TRAILING ? I0 : 0);
    else apply(ops, -1, true, TRAILING ? I0 : 0);
    if (j == 4 * R0 + g) acc[tix(I0, I0)][R0] -= 2.0;
    wave_sync();  // the next block overwrites Qt / Wt
  }

  template <bool TRAILING>
  __device__ __forceinline__ bool sweep() {
    double pmin = 1.0;  // smallest pivot seen
    Ops ops;
#pragma unroll
    for (int B = 0; B < 16; ++B) block_step<TRAILING>(B, ops, pmin);
    if constexpr (!TRAILING) {
#pragma unroll
      for (int t = 0; t < kTiles; ++t) acc[t] = -acc[t];
    }
    // a NaN pivot poisons every entry of W and with it every tile that is still being updated: the last diagonal
    // tile is updated by every block of both sweeps
    const double probe = acc[tix(3, 3)][0];
    return (pmin > 0.0) && __all(probe == probe);
  }

  // ---- u = M^-1 b from the trailing sweep's factors (tile (K, K) = -P_K^-1, tile (I, K) = T_IK = A_IK P_K^-1) ------
  // forward, right-looking: y_K is final when step K starts; z_K = P_K^-1 y_K and b_I -= T_IK y_K for I > K come
  // from the tiles of tile column K through the mat-vec's row partial sums.  Backward, right-looking: u_I is final
  // when step I starts and z_K -= T_IK^T u_I for K < I comes through the mirrored partial sums.
  __device__ __forceinline__ double solve_factored(double b) {
    const int g = lane >> 4, j = lane & 15;
    w.nat[lane] = (lane < dim) ? b : 0.0;
    wave_sync();
#pragma unroll
    for (int K = 0; K < 4; ++K) {
      const double yk = w.nat[16 * K + j];
#pragma unroll
      for (int I = K; I < 4; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) w.part[(16 * I + 4 * r + g) * kPartStride + j] = acc[tix(I, K)][r] * yk;
      wave_sync();
      if (lane >= 16 * K) {
        const double* src = w.part + lane * kPartStride;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
#pragma unroll
        for (int k = 0; k < 16; k += 4) { a0 += src[k]; a1 += src[k + 1]; a2 += src[k + 2]; a3 += src[k + 3]; }
        const double sum = (a0 + a1) + (a2 + a3);
        w.nat[lane] = (lane < 16 * K + 16) ? -sum : w.nat[lane] - sum;  // z_K (the tile is -P^-1) | b_I - T_IK y_K
      }
      wave_sync();
    }
#pragma unroll
    for (int I = 3; I >= 1; --I) {
      const d4 ur = d4{w.nat[16 * I + g], w.nat[16 * I + 4 + g], w.nat[16 * I + 8 + g], w.nat[16 * I + 12 + g]};
#pragma unroll
      for (int K = 0; K < I; ++K) {
        const d4 a = acc[tix(I, K)];
        double m = a[0] * ur[0];
        m = __builtin_fma(a[1], ur[1], m);
        m = __builtin_fma(a[2], ur[2], m);
        m = __builtin_fma(a[3], ur[3], m);
        w.mpart[(K * 4 + g) * 16 + j] = m;
      }
      wave_sync();
      if (lane < 16 * I) {
        const double* m = w.mpart + (lane >> 4) * 64 + (lane & 15);
        w.nat[lane] -= (m[0] + m[16]) + (m[32] + m[48]);
      }
      wave_sync();
    }
    const double u = (lane < dim) ? w.nat[lane] : 0.0;
    wave_sync();
    return u;
  }

  // implicit_core.h, kUnifiedConstruct: metric_func(x), then the explicit inverse (kept in the tiles for matvec /
  // half_vjp_inv / dh2_dpos) or the single solve u = M(x)^-1 rhs (systems.py:1381-1399)
  __device__ __forceinline__ bool construct(double x, bool need_inverse, double rhs, double* u) {
    // every construction ends the life of the inverse held so far (a solve-only construction overwrites the tiles it would
    // be rebuilt from; nothing applies it before the next construction with need_inverse - implicit_core.h drops the
    // anchor).  Saying so keeps its 128 registers from being carried across the sweeps.
#pragma unroll
    for (int k = 0; k < 64; ++k) fr_[k] = 0.0;
    fd_ = 0.0;
    if constexpr (RMETRIC == MM_RMETRIC_USER) {  // (and a refinement solve's tiles of M(x) are dead between solves)
#pragma unroll
      for (int t = 0; t < kTiles; ++t) mx_[t] = d4{0.0, 0.0, 0.0, 0.0};
    }
    bool ok = build(x);
    if (need_inverse) {  // wave-uniform
      ok = sweep<false>() && ok;
      tiles_to_rows();  // the inverse is APPLIED in row form (above); the tiles are dead until the next construction
    } else {
      ok = sweep<true>() && ok;
      *u = solve_factored(rhs);
    }
    return ok;
  }
  __device__ __forceinline__ bool build_and_invert(double x) {
    double dummy;
    return construct(x, true, 0.0, &dummy);
  }
  __device__ __forceinline__ bool build_and_solve(double x, double rhs, double* u) { return construct(x, false, rhs, u); }

  // ---- y = M(x0)^-1 v with the inverse in row form ------------------------------------------------------------------
  __device__ __forceinline__ double matvec(double v) {
    const double y = row_dot(fr_, v);
    return lane < dim ? y : 0.0;
  }

  __device__ __forceinline__ double diag() { return lane < dim ? fd_ : 0.0; }

  // implicit_core.h lowrank_update: F += al a a^T + be (a b^T + b a^T) + ga b b^T on the row form - lane i adds
  // a_i u_j + b_i v_j to its row, u = al a + be b and v = be a + ga b broadcast from LDS.  (The tiles of the sweep go stale:
  // the next factorisation rebuilds them; every product of the step reads the rows.)
  __device__ __forceinline__ void inverse_update(double al, double be, double ga, double a, double b) {
    const double u = lane < dim ? __builtin_fma(al, a, be * b) : 0.0, v = lane < dim ? __builtin_fma(be, a, ga * b) : 0.0;
    w.nat[lane] = u;
    w.aux[lane] = v;
    wave_sync();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const d4 uu = *reinterpret_cast<const d4*>(w.nat + 4 * k), vv = *reinterpret_cast<const d4*>(w.aux + 4 * k);
#pragma unroll
      for (int e = 0; e < 4; ++e) fr_[4 * k + e] = __builtin_fma(a, uu[e], __builtin_fma(b, vv[e], fr_[4 * k + e]));
    }
    fd_ = __builtin_fma(a, u, __builtin_fma(b, v, fd_));
    wave_sync();
  }

  // 0.5 * vjp_metric_func(q)(V) of a user metric.  q is the point of the held inverse: build() left it in w.uq (natural
  // order) with its aux block in w.uaq.  OUTER: V = -u u^T, else the explicit inverse in the tiles - handed to the user's
  // team-form hook as it is (user_metric.h MM_USER_VJP_FLAT), or dumped to the chain's dense global array for V(i, j).
  template <bool OUTER>
  __device__ __forceinline__ double user_half_vjp(double u) {
    double r;
    if constexpr (mmuser::kFlatVjp) {
      if constexpr (OUTER) {
        mmuser::VjpOpsOuter<MfmaBackend> ops{*this, lane < dim ? u : 0.0};
        r = mmuser::vjp_flat(ops, w.uq, lane, dim, uparams, w.uaq);
      } else {
        mmuser::VjpOpsInv<MfmaBackend> ops{*this};
        r = mmuser::vjp_flat(ops, w.uq, lane, dim, uparams, w.uaq);
      }
    } else {
#if defined(MM_RTC_BUILD) && defined(MM_RTC_USER_METRIC)
      if constexpr (OUTER) {
        w.aux[lane] = (lane < dim) ? u : 0.0;
        wave_sync();
        const MmMat vm{nullptr, w.aux, 0};
        r = (lane < dim) ? mmuser::vjp_dense(w.uq, vm, lane, dim, uparams, w.uaq) : 0.0;
        wave_sync();
      } else {
#pragma unroll
        for (int jj = 0; jj < 64; ++jj) work[lane * 64 + jj] = fr_[jj];  // lane i owns row i of the inverse
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the wave's own stores, read back by its other lanes
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const MmMat vm{work, nullptr, 64};
        r = (lane < dim) ? mmuser::vjp_dense(w.uq, vm, lane, dim, uparams, w.uaq) : 0.0;
      }
#else
      r = 0.0;
#endif
    }
    return lane < dim ? 0.5 * r : 0.0;
  }

  // 0.5 * vjp_metric(M^-1): rank-one metric M^-1 q / D; diag-quad metric q_i (M^-1)_ii
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
      const double uq = wave_sum(lane < dim ? u * q : 0.0);
      return -(u * uq) * inv_dim_;
    } else {
      return -q * (u * u);
    }
  }
  __device__ __forceinline__ double norm(double x, int kind) {
    const double a = wave_norm_accum(0.0, lane < dim ? x : 0.0, kind);
    return wave_norm_finish(a, kind);
  }
  __device__ __forceinline__ double grad(double q) {
    w.nat[lane] = (lane < dim) ? q : 0.0;
    wave_sync();
    const TargetAux aux = target_prepare<false>(target, w.nat, dim, tparams, lane);
    const double gr = (lane < dim) ? target_grad_elem<false>(target, aux, w.nat, lane, dim, tparams) : 0.0;
    wave_sync();
    return gr;
  }
};

template <int RMETRIC, bool PROFILE = false, bool LOWRANK = false>
__device__ __forceinline__ void implicit_mfma_body(const ImplicitArgs& A, double* lds) {
  double* base_lds = lds;
  const int base_elems = (RMETRIC == MM_RMETRIC_RANK1) ? kBaseDoubles : 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int dim = A.dim;
  if constexpr (RMETRIC == MM_RMETRIC_RANK1) {
    // base_lds[row * kBasePitch + col] = B[row][col], zero outside dim x dim
    for (int idx = threadIdx.x; idx < kBaseDoubles; idx += blockDim.x) {
      const int row = idx / kBasePitch, col = idx - row * kBasePitch;
      base_lds[idx] = (row < dim && col < dim) ? A.rparams[(int64_t)row * dim + col] : 0.0;
    }
  }
  __syncthreads();
  const int64_t chain = (int64_t)blockIdx.x * kWaves + wave;
  if (chain >= A.n_chains) return;  // no block-level barrier below this point
  double* wl = lds + base_elems + wave * mfma_wave_doubles<RMETRIC>();
  const bool act = lane < dim;
  double q = act ? A.pos[chain * dim + lane] : 0.0;
  double p = act ? A.mom[chain * dim + lane] : 0.0;
  const double t = signed_step(A.dir, A.step_scale, chain, A.step_size);

  MfmaBackend<RMETRIC, PROFILE, LOWRANK> bk;
  bk.dim = dim;
  bk.inv_dim_ = 1.0 / (double)dim;
  bk.lane = lane;
  bk.target = A.target;
  bk.w.qt = wl;
  bk.w.wt = wl + 256;
  bk.w.nat = wl + 512;
  bk.w.vperm = wl + 576;
  bk.w.aux = wl + 640;
  bk.w.part = wl + 704;
  bk.w.mpart = bk.w.part + 64 * kRowPitch;
  bk.w.stash = bk.w.mpart + 192;
  bk.w.prof = bk.w.stash + SL_COUNT_REFINE * 64;
  bk.refine_on = A.no_refine == 0;
  bk.dual_off = A.no_dual != 0;
  bk.lr_refresh_ = A.lowrank_refresh;
  bk.base_lds = base_lds;
  bk.tparams = A.tparams;
  bk.uparams = A.rparams;
  bk.work = nullptr;
  if constexpr (RMETRIC == MM_RMETRIC_USER) {
    constexpr int kA = (mmuser::kAux + 1) & ~1;
    double* up = wl + kMfmaWaveDoubles;
    bk.w.uq = up;
    bk.w.ux = up + 64;
    bk.w.uaq = up + 128;
    bk.w.uax = up + 128 + kA;
    bk.work = A.work ? A.work + chain * (int64_t)(64 * 64) : nullptr;
  }
  if constexpr (PROFILE) {
    if (lane < PH_COUNT + 2) bk.w.prof[lane] = lane == PH_COUNT + 1 ? (double)__builtin_readcyclecounter() : 0.0;
    wave_sync();
  }
  bk.slot(SL_Q) = q;
  bk.slot(SL_P) = p;
  const ChainResult r = implicit_leapfrog_chain(bk, t, mmdev::chain_steps(A.chain_steps, chain, A.n_steps), A.opts);
  q = bk.slot(SL_Q);
  p = bk.slot(SL_P);
  if (act) {
    A.pos[chain * dim + lane] = q;
    A.mom[chain * dim + lane] = p;
  }
  if (lane == 0) {
    A.status[chain] = r.status;
    A.n_done[chain] = r.done;
    add_counters(A.counters, r);
  }
  if constexpr (PROFILE) {  // out[chain][PH_COUNT]: cycles per phase of this chain's launch
    bk.prof_switch(PH_OTHER);
    wave_sync();
    if (lane < PH_COUNT) A.out[chain * PH_COUNT + lane] = bk.w.prof[lane];
  }
}


#ifndef MM_RTC_BUILD  // the in-tree instantiations (a run-time translation unit defines an extern "C" wrapper instead)
template <int RMETRIC, bool PROFILE = false, bool LOWRANK = false>
__global__ __launch_bounds__(64 * kWaves) void implicit_mfma_kernel(ImplicitArgs A) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  implicit_mfma_body<RMETRIC, PROFILE, LOWRANK>(A, lds);
}
#endif

}  // namespace mmmfma
