// Device code of the block-16 matrix-core team kernel (k_implicit_blk16.hip instantiates it for the built-in metrics;
// mm_rtc.hip compiles it at run time around a USER metric, user_metric.h).
//
// Implicit leapfrog on dense-metric Riemannian systems, 75 < D <= 256 (BASELINE config c4: D = 256, the 8-GPU
// headline): one 256-thread workgroup per chain = ONE WAVE PER SIMD of a CU, each wave with the full 512-register
// budget (256 architected VGPRs + 256 accumulation VGPRs), the chain's metric resident in the accumulation registers
// and factorised sixteen pivots at a time on the FP64 matrix cores.  gfx950 / CDNA4.
//
// What is different from k_implicit_mfma_team.hip (the round-1 kernel this replaces, still reachable with
// MICI_AMD_IMPLICIT_KERNEL=team4):
//   * 4 waves x 34 tiles instead of 8 waves x 17: the tiles (272 registers per wave) are MFMA accumulators in
//     AGPRs, which leaves the architected VGPR file to the operands and to the step's state - the round-1 kernel
//     squeezed 17 tiles + everything else into 256 VGPRs and spilled 744 bytes per lane (36 GB of HBM traffic per
//     launch);
//   * pivot blocks are 16 wide = one tile: ONE workgroup barrier per 16 pivots instead of two per 4, and four
//     back-to-back MFMAs per tile per block, so the per-tile scalar work is amortised fourfold;
//   * the 16 x 16 pivot block is inverted by every wave redundantly (an in-tile 4-wide sweep whose rank-4
//     updates are single MFMAs), so -W = -P^-1 (Q - E) never travels through LDS: each wave forms the four
//     16 x 16 blocks of -W it needs as MFMA outputs, which ARE the A-operand registers of its tile updates;
//   * solve-only constructions (the position fixed-point iterations need ONE product M^-1 p each; only the
//     construction at a new position needs the explicit inverse, systems.py:1381-1399) run the sweep on the
//     trailing tiles only - a blocked LDL^T, D^3/3 flops instead of D^3 - followed by a forward / diagonal /
//     backward substitution over the 16 tile rows;
//   * all waves run the same code: a wave's 34 tiles sit in fixed register slots whose tile coordinates are
//     wave-uniform run-time values (see slot map below), so nothing is specialised per wave.
// The index arithmetic of every phase is restated lane for lane in tools/sim_blk16.py and checked there against
// numpy.linalg.
//
// Layouts.  D is padded to 256 = 16 x 16 tiles of 16 x 16; the 136 tiles on or below the diagonal are kept, each in
// the MFMA accumulator layout: lane l = 16 g + j, register r <-> entry (16 I + 4 r + g, 16 J + j).  Wave w owns the
// tile rows 7-w, w, 15-w, 8+w (8-w, w+1, 16-w, 9+w tiles: 34 for every wave).  Slots 0..8 ("group X"): row 7-w from
// its diagonal leftwards, then row w ENDING on its diagonal (slot 8); slots 9..33 ("group Y", u = s - 9): row 15-w
// from its diagonal leftwards, then row 8+w ending on its diagonal (u = 24).  The panel of a block,
// X[k][c] = A[16 I0 + k][c], lives in LDS as Xl[c][g][kk] (k = 4 kk + g, 18 doubles per column): the four B-operand
// values of a lane are one 32-byte read, bank-conflict free across the sixteen lanes of a row.
//
// Reference arithmetic replaced: DensePositiveDefiniteMatrix factorisation, explicit inverse and solves
// (matrices.py:1161-1188, 932-938) inside ImplicitLeapfrogIntegrator._step (integrators.py:493-544); the step
// logic is implicit_core.h.
#pragma once
#include "implicit_core.h"
#include "user_metric.h"

namespace mmblk16 {

using namespace mmdev;
using namespace mmimp;

typedef double d4 __attribute__((ext_vector_type(4)));

// Every case of the "which slot holds the tile of column K" switches ENDS with a distinct marker: with identical
// tails the optimiser sinks the cases into one block that indexes the tile array dynamically - which moves the
// array from registers to scratch memory (code sinking works from the end of the blocks upwards).
#define MM_CASE_MARK(N) asm volatile("; tile slot case " #N)

#ifndef MM_BLK16_UPD_BARRIER
#define MM_BLK16_UPD_BARRIER 0  // (inverse_update: a scheduling barrier every two tiles - compile record: 760 B of scratch against 500)
#endif
#ifndef MM_BLK16_PERMLANE
#define MM_BLK16_PERMLANE 1  // semantics verified on the MI355X by tests/test_gpu_blk16.py::test_permlane_swap_semantics
#endif

constexpr int NT16 = 16;            // tile rows
constexpr int DPM = 16 * NT16;      // padded dimension
constexpr int NWAVE = 8;            // two per SIMD
constexpr int NTHR = 64 * NWAVE;
constexpr int NSLOT = 17;           // tiles per wave
constexpr int NCLASS = 2;           // tile rows per wave
constexpr int CS = 18;              // doubles per panel column in LDS: 16 + 2 (keeps 16-byte alignment, spreads banks)
constexpr int PSTR = 17;            // partial sums per output element: 16 column-sum slots + the row sum
constexpr int VLM = DPM + 8;        // flat vectors: DPM elements + a dummy cell for threads >= DPM
__host__ __device__ constexpr int tix(int I, int J) { return I * (I + 1) / 2 + J; }  // lower tile (I, J) of the base image
constexpr int kInvWave = 7;         // the wave that inverts the pivot blocks: tile rows 8 and 7, with wave 3 (rows 12, 3)
                                    // the SIMD with the least tile work in a trailing sweep

// LDS (doubles).  The flat per-thread state comes first: its thirteen slots are then ONE address register plus a 16-bit
// immediate offset each (DS instructions carry offsets < 64 KiB), not thirteen address registers.
constexpr int kOffStash = 0;                               // [SL_COUNT_REFINE][VLM] flat per-thread state of the step
constexpr int kOffNat = kOffStash + SL_COUNT_REFINE * VLM; // [VLM] natural-order vector
constexpr int kOffVperm = kOffNat + VLM;                   // [DPM] the same vector as [I][g][r]: row operands as one 32-byte read
constexpr int kOffAux = kOffVperm + DPM;                   // [VLM] second natural-order vector (z of the substitution)
constexpr int kOffRed = kOffAux + VLM;                     // [24]  team reductions / flags
constexpr int kOffScr = kOffRed + 24;                      // [NWAVE][64]: [0] scratch of the in-tile sweep, [1..4] T = -P^-1
                                                           // in lane order, [5][0] its positive-definite flag
constexpr int kOffX = kOffScr + NWAVE * 64;                // [2][DPM][CS]  panel, double-buffered by block parity
constexpr int kOffPart = kOffX + 2 * DPM * CS;             // [DPM][PSTR]
constexpr int kOffB = kOffPart + DPM * PSTR;               // [DPM] right-hand side of a solve-only construction: the forward
                                                           // substitution runs inside the trailing sweep, in place
constexpr int kLdsDoubles = kOffB + DPM;
// a user metric (user_metric.h) adds: the point of the held inverse in natural order, and the aux blocks of that point and of
// the refinement products' point (which itself sits in the panel buffers: kOffXnat)
constexpr int kOffUq = kLdsDoubles;                        // [DPM]
constexpr int kOffUaq = kOffUq + DPM;                      // [kAux, rounded up to even]
constexpr int kOffUax = kOffUaq + ((mmuser::kAux + 1) & ~1);
constexpr int kUserLdsDoubles = kOffUax + ((mmuser::kAux + 1) & ~1);
template <int RMETRIC>
__host__ __device__ constexpr int blk16_lds_doubles() { return RMETRIC == MM_RMETRIC_USER ? kUserLdsDoubles : kLdsDoubles; }
static_assert(kUserLdsDoubles * 8 <= 160 * 1024, "LDS budget of a CU (user metric)");
constexpr int kOffCnt = kOffScr + 6 * 64 + 24;             // [16] work counters (free part of Scr: a user metric's LDS is full to the last 64 bytes)
constexpr int kOffProf = kOffScr + 6 * 64 + 8;             // developer builds: [PH_COUNT + 2] phase clocks (free part of Scr)
static_assert(PH_COUNT + 2 <= 48, "phase clocks must fit the unused part of the Scr block");
// Scratch of the refinement solves (implicit_core.h refine_solve) lives in the panel buffers: no sweep runs while one
// is in flight.  The point x of metric_apply(), then RS_COUNT flat per-thread vectors.
constexpr int kOffXnat = kOffX;                            // [VLM]
constexpr int kOffRs = kOffXnat + VLM;                     // [RS_COUNT][VLM]
static_assert(kOffRs + RS_COUNT * VLM <= kOffPart, "refinement scratch must fit the panel buffers");
static_assert((kOffVperm % 2) == 0 && (kOffScr % 2) == 0 && (kOffX % 2) == 0, "16-byte alignment of the d4 accesses");
static_assert(kLdsDoubles * 8 <= 160 * 1024, "LDS budget of a CU");
static_assert((kLdsDoubles + DPM + 2 * 560) * 8 <= 160 * 1024, "LDS budget of a CU with the largest user aux block (MM_USER_AUX = 560)");

__device__ __forceinline__ int fresh_lane() {
  // the lane index straight from the hardware (two VALU instructions): never worth a long-lived register
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

// Pull a value the matrix core produced (an accumulation-register tuple) into architected VGPRs.  The metric tiles
// fill 240 of a wave's 256 AGPRs; what is transient (T, the -W blocks) or needs VALU access all the time (the four
// diagonal tiles) must not compete for the remaining 16.
__device__ __forceinline__ void to_vgpr(d4& v) { asm volatile("" : "+v"(v)); }

// The wave index re-materialised as an opaque scalar: everything derived from it (the 17 slots' tile coordinates,
// their LDS / global offsets) is then recomputed where it is used - a few SALU instructions - instead of being
// hoisted out of the step loop into dozens of long-lived SGPRs that get spilled to VGPR lanes.
__device__ __forceinline__ int opaque_wave(int v) {
  v = __builtin_amdgcn_readfirstlane(v);
  asm volatile("" : "+s"(v));
  __builtin_assume(v >= 0 && v < NWAVE);
  return v;
}

// A value every lane of the wave agrees on, moved to scalar registers: the step's control flow (implicit_core.h)
// depends only on team-uniform norms and flags; telling the compiler so keeps the whole state machine (mode, solver
// iteration counts, work counters, the time step) in SGPRs instead of VGPRs the tiles need.
__device__ __forceinline__ double uniform_f64(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffLL));
  const int hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ bool uniform_flag(bool v) { return __builtin_amdgcn_readfirstlane(v ? 1 : 0) != 0; }

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

// sum over the four DPP rows (lanes l, l ^ 16, l ^ 32, l ^ 48): result in all four.
// gfx950 has VALU row swaps: v_permlane16_swap exchanges the odd rows of its first operand with the even rows of
// its second, v_permlane32_swap the upper half of the first with the lower half of the second; fed the same value
// twice, the two results are (x[l], x[l ^ 16]) in some order, so their sum is the xor-16 (xor-32) butterfly step
// without the LDS crossbar round trip of ds_bpermute (two of them per step for a double).
constexpr bool kUsePermlaneSwap = MM_BLK16_PERMLANE;
__device__ __forceinline__ double swap_sum16(double m) {
  const long long b = __double_as_longlong(m);
  const unsigned lo = (unsigned)(b & 0xffffffffLL), hi = (unsigned)(b >> 32);
  const auto l2 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto h2 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  const double x0 = __longlong_as_double(((long long)h2[0] << 32) | (unsigned)l2[0]);
  const double x1 = __longlong_as_double(((long long)h2[1] << 32) | (unsigned)l2[1]);
  return x0 + x1;
}
__device__ __forceinline__ double swap_sum32(double m) {
  const long long b = __double_as_longlong(m);
  const unsigned lo = (unsigned)(b & 0xffffffffLL), hi = (unsigned)(b >> 32);
  const auto l2 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto h2 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  const double x0 = __longlong_as_double(((long long)h2[0] << 32) | (unsigned)l2[0]);
  const double x1 = __longlong_as_double(((long long)h2[1] << 32) | (unsigned)l2[1]);
  return x0 + x1;
}
__device__ __forceinline__ double sum_over_g(double m) {
  if constexpr (kUsePermlaneSwap) {
    return swap_sum32(swap_sum16(m));
  } else {
    m += __shfl_xor(m, 16);
    m += __shfl_xor(m, 32);
    return m;
  }
}

// Four lane-partials reduced over the four 16-lane rows of the wave AT ONCE: on return the lanes of row g hold the
// complete sum of value number g.  One v_permlane16_swap pairs the values (the swapped registers are both results: no
// copies), one v_permlane32_swap pairs the pairs - 9 instructions for four sums where four sum_over_g() are 48.
__device__ __forceinline__ double swap_pair(double a, double b, bool by32) {
  const long long ba = __double_as_longlong(a), bb = __double_as_longlong(b);
  const unsigned la = (unsigned)(ba & 0xffffffffLL), ha = (unsigned)(ba >> 32);
  const unsigned lb = (unsigned)(bb & 0xffffffffLL), hb = (unsigned)(bb >> 32);
  if (by32) {
    const auto l2 = __builtin_amdgcn_permlane32_swap(la, lb, false, false);
    const auto h2 = __builtin_amdgcn_permlane32_swap(ha, hb, false, false);
    return __longlong_as_double(((long long)h2[0] << 32) | (unsigned)l2[0]) +
           __longlong_as_double(((long long)h2[1] << 32) | (unsigned)l2[1]);
  }
  const auto l2 = __builtin_amdgcn_permlane16_swap(la, lb, false, false);
  const auto h2 = __builtin_amdgcn_permlane16_swap(ha, hb, false, false);
  return __longlong_as_double(((long long)h2[0] << 32) | (unsigned)l2[0]) +
         __longlong_as_double(((long long)h2[1] << 32) | (unsigned)l2[1]);
}
__device__ __forceinline__ double sum4_over_g(double a, double b, double c, double d) {
  if constexpr (kUsePermlaneSwap) {
    // rows 0, 2 of ab: a summed over the row pairs (0, 1), (2, 3); rows 1, 3: b likewise
    const double ab = swap_pair(a, b, false), cd = swap_pair(c, d, false);
    return swap_pair(ab, cd, true);  // row 0: a, row 1: b, row 2: c, row 3: d
  } else {
    const int g = (int)(threadIdx.x & 63) >> 4;
    const double sa = sum_over_g(a), sb = sum_over_g(b), sc = sum_over_g(c), sd = sum_over_g(d);
    return g == 0 ? sa : (g == 1 ? sb : (g == 2 ? sc : sd));
  }
}

// rs[r] = this lane's partial of row element 4 r + g: sum over the 16 lanes of a DPP row with one transposing
// butterfly (4 values -> 1).  All four lanes of a quad end up with the sum for register r = j >> 2, i.e. for row
// element 4 (j >> 2) + g.
__device__ __forceinline__ double row_reduce16(const d4 rs, const int j) {
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

// the workgroup as a team (user_metric.h mm_user_prepare)
struct Team16 {
  double* red;
  int tid;
  __device__ __forceinline__ int rank() const { return tid; }
  __device__ __forceinline__ int size() const { return NTHR; }
  __device__ __forceinline__ double sum(double x) const { return uniform_f64(team_reduce(x, 0, red)); }
};

template <int RMETRIC, bool PROFILE = false, bool LOWRANK = false>
struct TeamBlk16 {
  // implicit_core.h lowrank_solve (round 6): the rank-one-update metric's solve-only constructions by the Woodbury identity
  // from the held inverse - one product F d (matvec) each instead of ~2.5 CG pairs of M(x) v + F r
  // (the built-in rank-one-update metric, or a user metric that declares the structure: user_metric.h MM_USER_LOWRANK)
  static constexpr bool kLowRankBuiltin = RMETRIC == MM_RMETRIC_RANK1;
  static constexpr bool kLowRank = LOWRANK && (kLowRankBuiltin || (RMETRIC == MM_RMETRIC_USER && mmuser::kLowRank));
  __device__ __forceinline__ double lowrank_scale() const {
    if constexpr (kLowRankBuiltin) return (double)dim;
    else return uniform_f64(mmuser::lowrank_inv_s(dim, base));
  }
  // u(x), this thread's element (user metric: the point published for the hook, its aux block prepared - a team collective)
  __device__ __forceinline__ double lowrank_vec(double x) {
    if constexpr (kLowRankBuiltin) {
      return x;
    } else {
      metric_point(x);
      const double u = mmuser::lowrank_u(lds + kOffXnat, tid, dim, base, lds + kOffUax);
      __syncthreads();  // (the next point overwrites kOffXnat / kOffUax)
      return tid < dim ? u : 0.0;
    }
  }
  __device__ __forceinline__ double& lowrank_u0() {
    if constexpr (kLowRankBuiltin) return slot(SL_Q);
    else return rslot(LR_U0);
  }
  // a user metric's hooks evaluate its vector-Jacobian products at "the point of the held inverse" (kOffUq, kOffUaq - build()
  // sets them): an inverse carried to x by lowrank_update takes the point with it
  __device__ __forceinline__ void held_point(double x) {
    if constexpr (RMETRIC == MM_RMETRIC_USER) {
      if (tid < DPM) lds[kOffUq + tid] = tid < dim ? x : 0.0;
      __syncthreads();
      mmuser::prepare(Team16{lds + kOffRed, (int)tid}, lds + kOffUq, dim, base, lds + kOffUaq);
      __syncthreads();
    }
  }
  __device__ static constexpr bool lowrank_on() { return true; }  // (compile-time: the launcher picks the instantiation)
  int lr_refresh_;
  __device__ __forceinline__ int lowrank_refresh() const { return lr_refresh_; }
  // implicit_core.h lowrank_solve2: the reversibility-check solve and the C-adjoint solve of a step in lock step on the
  // Woodbury path of the built-in metric - one pass over the register tiles serves both products (matvec2_exact), one
  // barrier both norms / all six inner products (MICI_AMD_DUAL=0: one solve after the other).  A/B macro.
#ifndef MM_BLK16_LR_DUAL
#define MM_BLK16_LR_DUAL 1
#endif
  static constexpr bool kDual = MM_BLK16_LR_DUAL && LOWRANK && RMETRIC == MM_RMETRIC_RANK1;
  static constexpr bool kDualLowRankOnly = true;  // (implicit_core.h: no paired CG products on this backend)
  bool dual_off;
  static constexpr bool kSolveByInverse = false;
  static constexpr bool kUnifiedConstruct = true;  // implicit_core.h: one construction site, mode at run time
  static constexpr bool kCountersInLds = true;     // implicit_core.h: work counters in LDS, bumped by thread 0
  static constexpr bool kRefine = true;            // implicit_core.h: solve-only constructions refined from the held inverse
  static constexpr bool kProf = PROFILE;           // developer builds: cycles per phase of the step (prof_switch)
  bool refine_on;                                  // false: MICI_AMD_REFINE=0, every construction is factorised
  d4 acc[NSLOT];
  int wave; // wave index, wave-uniform (phases re-materialise it through opaque_wave)
  int nblk; // number of 16-pivot blocks that contain real rows: ceil(dim / 16)
  int dim, target;
  double inv_dim_;  // 1 / D of the rank-one metric (multiplied with: an IEEE division is ~30 dependent instructions)
  // the thread index, re-materialised opaquely at every use: per-thread addresses derived from it are computed where
  // needed instead of being hoisted out of the step loop into long-lived VGPRs
  struct OpaqueTid {
    int v;
    __device__ __forceinline__ operator int() const {
      int x = v;
      asm volatile("" : "+v"(x));
      return x;
    }
  } tid;
  double* lds;
  const double* base;  // rank-one metric: the base matrix tile by tile in lane order (mm_model::d_rmetric_tiled);
                       // user metric: its params
  const double* tparams;
  double* work;        // user metric with the dense-accessor VJP: this chain's DPM x DPM doubles of global memory
  __device__ __forceinline__ bool flat_active() const { return tid < dim; }

  // ---- slot map (wave-uniform; see the file header).  Row class of a slot: 0 = tile row 15-w (slots 0..15-w, from
  // its diagonal leftwards), 1 = tile row w (slots 16-w..16, ENDING on its diagonal).  Slots 0..8 and 16 have a
  // compile-time class.
  __device__ static __forceinline__ int row_class(const int s, const int w) {
    return (s <= 8 || (s < NSLOT - 1 && s <= 15 - w)) ? 0 : 1;
  }
  __device__ static __forceinline__ int row_of_class(const int c, const int w) { return c == 0 ? 15 - w : w; }
  __device__ static __forceinline__ int tile_i(const int s, const int w) { return row_of_class(row_class(s, w), w); }
  __device__ static __forceinline__ int tile_j(const int s, const int w) {
    return row_class(s, w) == 0 ? 15 - w - s : w + s - 16;
  }
  __device__ static constexpr bool is_diag_slot(const int s) { return s == 0 || s == NSLOT - 1; }
  __device__ static constexpr bool class_known(const int s) { return s <= 8 || s == NSLOT - 1; }
  // per-row quantity of slot s out of the row classes' values (a select, not a branch: a branch around MFMAs makes
  // the accumulator a PHI that the compiler resolves with copies behind a full MFMA drain)
  __device__ static __forceinline__ d4 pick_row(const int s, const int w, const d4 (&v)[NCLASS]) {
    if (s <= 8) return v[0];
    if (s == NSLOT - 1) return v[1];
    const bool lo = s <= 15 - w;
    d4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = lo ? v[0][k] : v[1][k];
    return r;
  }
  // accumulate v * a into the in-lane sums of slot s's row class (masked operands for the run-time slots)
  __device__ static __forceinline__ void add_row(const int s, const int w, d4 (&rs)[NCLASS], const d4 a, const double v) {
    if (class_known(s)) {
      const int c = s <= 8 ? 0 : 1;
#pragma unroll
      for (int r = 0; r < 4; ++r) rs[c][r] = __builtin_fma(a[r], v, rs[c][r]);
    } else {
      const bool lo = s <= 15 - w;
      const double va = lo ? v : 0.0, vb = lo ? 0.0 : v;
#pragma unroll
      for (int r = 0; r < 4; ++r) rs[0][r] = __builtin_fma(a[r], va, rs[0][r]);
#pragma unroll
      for (int r = 0; r < 4; ++r) rs[1][r] = __builtin_fma(a[r], vb, rs[1][r]);
    }
  }

  // The mirrored (column) partials of four consecutive below-diagonal slots s0 .. s0 + 3, reduced over the wave's four
  // rows together (sum4_over_g): the lanes of row g then own slot s0 + g and store its sum.
  __device__ __forceinline__ void store_mirrored4(double* part, const int s0, const int w, const int g, const int j,
                                                  const double a, const double b, const double c, const double d) {
    const double m = sum4_over_g(a, b, c, d);
    const int o0 = 16 * tile_j(s0, w) * PSTR + tile_i(s0, w);
    const int o1 = 16 * tile_j(s0 + 1, w) * PSTR + tile_i(s0 + 1, w);
    const int o2 = 16 * tile_j(s0 + 2, w) * PSTR + tile_i(s0 + 2, w);
    const int o3 = s0 + 3 < NSLOT - 1 ? 16 * tile_j(s0 + 3, w) * PSTR + tile_i(s0 + 3, w) : 0;
    const int off = g == 0 ? o0 : (g == 1 ? o1 : (g == 2 ? o2 : o3));
    if (s0 + 3 < NSLOT - 1 || g < 3) part[off + j * PSTR] = m;
  }

  __device__ __forceinline__ void count(const int which, const int n) {
    if (tid == 0) lds[kOffCnt + which] += (double)n;  // exact in a double far beyond any launch's counts
  }
  __device__ __forceinline__ void read_counts(ChainResult& r) const {  // only thread 0's copy is used
    r.n_evals = (long long)lds[kOffCnt + CNT_EVALS];
    r.n_solves = (long long)lds[kOffCnt + CNT_SOLVES];
    r.n_metric = (long long)lds[kOffCnt + CNT_METRIC];
    r.n_grad = (long long)lds[kOffCnt + CNT_GRAD];
    r.n_refine = (long long)lds[kOffCnt + CNT_REFINE];
    r.n_full = (long long)lds[kOffCnt + CNT_FULL];
    r.n_trail = (long long)lds[kOffCnt + CNT_TRAIL];
    r.n_lowrank = (long long)lds[kOffCnt + CNT_LOWRANK];
    r.n_inv_update = (long long)lds[kOffCnt + CNT_INVUPD];
  }
  static_assert(CNT_COUNT <= 16, "work counters occupy lds[kOffCnt .. + 15]");
  // developer builds: the clock since the last call goes to the phase announced then; [kOffProf + PH_COUNT] = that
  // phase, [+ PH_COUNT + 1] = the time of the call.  Thread 0 only.
  __device__ __forceinline__ int prof_switch(int phase) {
    int old = 0;
    if (tid == 0) {
      double* P = lds + kOffProf;
      const double now = (double)__builtin_readcyclecounter();
      old = (int)P[PH_COUNT];
      P[old] += now - P[PH_COUNT + 1];
      P[PH_COUNT] = (double)phase;
      P[PH_COUNT + 1] = now;
    }
    return __builtin_amdgcn_readfirstlane(old);
  }
  __device__ __forceinline__ double& rslot(int i) { return lds[kOffRs + i * VLM + (tid < DPM ? tid : DPM)]; }

  // two team-uniform sums over the chain's elements in one pass (two barriers)
  // Round 6, the Woodbury kernels (kLowRank): a step there is ~21 products and ~45 team reductions, and its ~130 workgroup
  // barriers are a sixth of its time.  Their reductions alternate between TWO sets of partials (as softabs.h block_reduce4
  // and implicit_global.h reduce do): the writer of a set is always behind a barrier every reader of its previous contents
  // has passed, so ONE barrier per reduction is enough.  rflip_ is team-uniform (every wave runs the same reductions).
  int rflip_;
  __device__ __forceinline__ double* flip_set() {
    double* p = lds + kOffScr + 7 * 64 + 32 * rflip_;  // [2][up to 3 values][8 waves]  (the inverting wave's sweep scratch)
    rflip_ ^= 1;
    return p;
  }
  // up to four values through one barrier: v0 (and v1 with MAX01) by `max0 ? NaN-propagating max : sum`, the others sums
  template <int NV, bool MAX01 = false>
  __device__ __forceinline__ void flip_reduce(const bool max0, double v0, double v1, double v2, double* r0, double* r1,
                                              double* r2, double v3 = 0.0, double* r3 = nullptr) {
    const int lane = fresh_lane(), wv = opaque_wave(wave);
    v0 = max0 ? wave_max(v0) : wave_sum(v0);
    if constexpr (NV > 1) v1 = (MAX01 && max0) ? wave_max(v1) : wave_sum(v1);
    if constexpr (NV > 2) v2 = wave_sum(v2);
    if constexpr (NV > 3) v3 = wave_sum(v3);
    double* red = flip_set();
    if (lane == 0) {
      red[wv] = v0;
      if constexpr (NV > 1) red[8 + wv] = v1;
      if constexpr (NV > 2) red[16 + wv] = v2;
      if constexpr (NV > 3) red[24 + wv] = v3;
    }
    __syncthreads();
    double a = red[0], b = NV > 1 ? red[8] : 0.0, c = NV > 2 ? red[16] : 0.0, d = NV > 3 ? red[24] : 0.0;
#pragma unroll
    for (int k = 1; k < NWAVE; ++k) {
      a = max0 ? nanmax(a, red[k]) : a + red[k];
      if constexpr (NV > 1) b = (MAX01 && max0) ? nanmax(b, red[8 + k]) : b + red[8 + k];
      if constexpr (NV > 2) c += red[16 + k];
      if constexpr (NV > 3) d += red[24 + k];
    }
    *r0 = uniform_f64(a);
    if constexpr (NV > 1) *r1 = uniform_f64(b);
    if constexpr (NV > 2) *r2 = uniform_f64(c);
    if constexpr (NV > 3) *r3 = uniform_f64(d);
  }
  __device__ __forceinline__ void sum4(double a, double b, double c, double d, double* sa, double* sb, double* sc,
                                       double* sd) {
    const bool act = tid < dim;
    flip_reduce<4>(false, act ? a : 0.0, act ? b : 0.0, act ? c : 0.0, sa, sb, sc, act ? d : 0.0, sd);
  }
  __device__ __forceinline__ void norm2(double a, double b, int kind, double* na, double* nb) {
    const bool act = tid < dim, linf = kind == MM_NORM_LINF;
    const double xa = act ? a : 0.0, xb = act ? b : 0.0;
    double ra, rb, unused;
    flip_reduce<2, true>(linf, linf ? fabs(xa) : xa * xa, linf ? fabs(xb) : xb * xb, 0.0, &ra, &rb, &unused);
    *na = linf ? ra : sqrt(ra);
    *nb = linf ? rb : sqrt(rb);
  }
  __device__ __forceinline__ void sum2(double a, double b, double* sa, double* sb) {
    const bool act = tid < dim;
    if constexpr (kLowRank) {
      double unused;
      flip_reduce<2>(false, act ? a : 0.0, act ? b : 0.0, 0.0, sa, sb, &unused);
      return;
    }
    const int lane = fresh_lane(), wv = opaque_wave(wave);
    a = wave_sum(act ? a : 0.0);
    b = wave_sum(act ? b : 0.0);
    double* red = lds + kOffRed;
    if (lane == 0) {
      red[wv] = a;
      red[8 + wv] = b;
    }
    __syncthreads();
    double ra = red[0], rb = red[8];
#pragma unroll
    for (int k = 1; k < NWAVE; ++k) {
      ra += red[k];
      rb += red[8 + k];
    }
    __syncthreads();
    *sa = uniform_f64(ra);
    *sb = uniform_f64(rb);
  }
  // a norm (kind as norm()) and a team sum through one barrier (implicit_core.h momentum_solve_lowrank)
  __device__ __forceinline__ void norm_dot(double x, int kind, double y, double* err, double* s) {
    const bool act = tid < dim;
    const double a = act ? x : 0.0;
    const bool linf = kind == MM_NORM_LINF;  // team-uniform
    double r, unused;
    flip_reduce<2>(linf, linf ? fabs(a) : a * a, act ? y : 0.0, 0.0, &r, s, &unused);
    *err = linf ? r : sqrt(r);
  }
  __device__ __forceinline__ void sum3(double a, double b, double c, double* sa, double* sb, double* sc) {
    const bool act = tid < dim;
    flip_reduce<3>(false, act ? a : 0.0, act ? b : 0.0, act ? c : 0.0, sa, sb, sc);
  }
  __device__ __forceinline__ double& slot(int i) { return lds[kOffStash + i * VLM + (tid < DPM ? tid : DPM)]; }

  __device__ __forceinline__ double norm(double x, int kind) {
    const double a = tid < dim ? x : 0.0;
    if constexpr (kLowRank) {
      const bool linf = kind == MM_NORM_LINF;  // team-uniform
      double r, u1, u2;
      flip_reduce<1>(linf, linf ? fabs(a) : a * a, 0.0, 0.0, &r, &u1, &u2);
      return linf ? r : sqrt(r);
    }
    if (kind == MM_NORM_LINF) return uniform_f64(team_reduce(fabs(a), 1, lds + kOffRed));
    return uniform_f64(sqrt(team_reduce(a * a, 0, lds + kOffRed)));
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

  // ---- metric_func(x) into the tiles ---------------------------------------------------------------------
  // returns this wave's "a diagonal entry is not finite" flag (matrices.py:211-215); combined over waves in sweep()
  __device__ __forceinline__ bool build(double x) {
    publish_vector(x);
    const int w = opaque_wave(wave);
    const int ln = fresh_lane(), g = ln >> 4, j = ln & 15;
    const double inv_d = 1.0 / (double)dim;
    double chk = 0.0;
    if constexpr (RMETRIC == MM_RMETRIC_RANK1) {
      // base matrix (L2-resident) + q q^T / D.  The base matrix is stored tile by tile in lane order: a lane's four
      // entries of a tile are one 32-byte load, a tile is 2 KB contiguous (wave-uniform tile origin + one lane offset).
      const d4* lane_base = reinterpret_cast<const d4*>(base) + ln;
#pragma unroll
      for (int s = 0; s < NSLOT; ++s) {
        const int I = tile_i(s, w), J = tile_j(s, w);
        const d4 b = lane_base[(unsigned)(tix(I, J) * 64)];
        const d4 qr = *reinterpret_cast<const d4*>(lds + kOffVperm + ((I * 4 + g) << 2));
        const double qs = lds[kOffNat + 16 * J + j] * inv_d;
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[s][r] = __builtin_fma(qr[r], qs, b[r]);
      }
    } else if constexpr (RMETRIC == MM_RMETRIC_USER) {
      // the point in natural order for the user's hooks (kOffNat is every mat-vec's scratch), its aux block, then the
      // user's metric_func entry by entry (zero on the padding; its diagonal is set to 1 below)
      if (tid < DPM) lds[kOffUq + tid] = lds[kOffNat + tid];
      __syncthreads();
      mmuser::prepare(Team16{lds + kOffRed, (int)tid}, lds + kOffUq, dim, base, lds + kOffUaq);
      __syncthreads();
#pragma unroll
      for (int s = 0; s < NSLOT; ++s) {
        const int I = tile_i(s, w), J = tile_j(s, w);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * I + 4 * r + g, jj = 16 * J + j;
          acc[s][r] = mmuser::entry_padded(lds + kOffUq, i, jj, dim, base, lds + kOffUaq);
          chk = __builtin_fma(acc[s][r], 0.0, chk);  // "Array is not finite.": every entry of a user metric is looked at
        }
        // one tile at a time: left alone the scheduler hoists every slot's parameter loads to the top - 17 tiles' worth
        // of operands in flight on top of the 136 tile registers being defined - and the allocator answers by keeping
        // tiles in scratch for the whole step
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int s = 0; s < NSLOT; ++s) acc[s] = d4{0.0, 0.0, 0.0, 0.0};
    }
    // the four diagonal tiles: entries (16 I + 4 r + g, same) on lanes j == 4 r + g
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      if (!is_diag_slot(s)) continue;
      const int I = tile_i(s, w);
      const d4 qr = *reinterpret_cast<const d4*>(lds + kOffVperm + ((I * 4 + g) << 2));
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool on_diag = (j == 4 * r + g);
        if constexpr (RMETRIC == MM_RMETRIC_DIAGQUAD) {
          if (on_diag) acc[s][r] = __builtin_fma(qr[r], qr[r], 1.0);
        }
        if (on_diag && 16 * I + 4 * r + g >= dim) acc[s][r] = 1.0;  // identity on the padding
        // both built-in metrics have their largest entries on the diagonal: a non-finite entry anywhere
        // implies a non-finite diagonal entry (matrices.py:211-215, "Array is not finite.")
        chk = __builtin_fma(acc[s][r], 0.0, chk);
      }
    }
    return __builtin_amdgcn_ballot_w64(chk != 0.0) != 0;
  }

  __device__ __forceinline__ double sum1(double a) {  // the second sum rides on the same two barriers for free
    double sa, sb;
    sum2(a, 0.0, &sa, &sb);
    return sa;
  }

  // ---- M(x) v without touching the tiles (they hold -M(x0)^-1): refine_solve's matrix-free product ---------------
  // Each metric in the form that suits it (as half_vjp_inv / dh2_dpos below do for the vector-Jacobian products):
  //   rank-one update  M(x) v = B v + x (x . v) / D   B streamed tile by tile from L2, contracted like matvec() contracts
  //                                                    the register tiles; the dot product rides on the same barrier
  //   diag(1 + x^2)    M(x) v = (1 + x_i^2) v_i       per thread
  __device__ __forceinline__ void metric_point(double x) {
    if (tid < DPM) lds[kOffXnat + tid] = tid < dim ? x : 0.0;  // built-in metrics: read back by the same thread only
    if constexpr (RMETRIC == MM_RMETRIC_USER) {  // the user's hooks read the whole point: publish it, then its aux block
      __syncthreads();
      mmuser::prepare(Team16{lds + kOffRed, (int)tid}, lds + kOffXnat, dim, base, lds + kOffUax);
      __syncthreads();
    }
  }
  __device__ __forceinline__ double metric_apply(double v) {
    const double x = tid < DPM ? lds[kOffXnat + tid] : 0.0;
    if constexpr (RMETRIC == MM_RMETRIC_DIAGQUAD) {
      return tid < dim ? __builtin_fma(x * x, v, v) : 0.0;
    } else {
      publish_vector(v);
      const int w = opaque_wave(wave);
      const int ln = fresh_lane(), g = ln >> 4, j = ln & 15;
      double* part = lds + kOffPart;
      if constexpr (RMETRIC == MM_RMETRIC_RANK1) {
        const double xv = wave_sum(tid < dim ? x * v : 0.0);
        if (ln == 0) lds[kOffRed + w] = xv;
      }
      d4 rs[NCLASS];
#pragma unroll
      for (int c = 0; c < NCLASS; ++c) rs[c] = d4{0.0, 0.0, 0.0, 0.0};
      const d4* lane_base = reinterpret_cast<const d4*>(base) + ln;
      // the base-matrix tiles kAhead slots ahead of their use: a lone load costs an L2 round trip (~1 us x 17 when each
      // waits for the previous slot's arithmetic).  A user metric's entries are evaluated where they are used.
      constexpr int kAhead = 4;
      d4 bq[kAhead];
      if constexpr (RMETRIC == MM_RMETRIC_RANK1) {
#pragma unroll
        for (int a = 0; a < kAhead; ++a) bq[a] = lane_base[(unsigned)(tix(tile_i(a, w), tile_j(a, w)) * 64)];
      }
      double vc_next = lds[kOffNat + 16 * tile_j(0, w) + j];  // (the vector operands one slot ahead, as matvec())
      d4 vr_next = *reinterpret_cast<const d4*>(lds + kOffVperm + ((tile_i(0, w) * 4 + g) << 2));
#pragma unroll
      for (int s = 0; s < NSLOT; ++s) {
        const int I = tile_i(s, w), J = tile_j(s, w);
        d4 m;
        if constexpr (RMETRIC == MM_RMETRIC_USER) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = 16 * I + 4 * r + g, jj = 16 * J + j;
            m[r] = mmuser::entry_padded(lds + kOffXnat, i, jj, dim, base, lds + kOffUax);
          }
        } else {
          m = bq[s % kAhead];
          if (s + kAhead < NSLOT)
            bq[s % kAhead] = lane_base[(unsigned)(tix(tile_i(s + kAhead, w), tile_j(s + kAhead, w)) * 64)];
        }
        const double vc = vc_next;
        const d4 vr = vr_next;
        if (s + 1 < NSLOT) {
          vc_next = lds[kOffNat + 16 * tile_j(s + 1, w) + j];
          vr_next = *reinterpret_cast<const d4*>(lds + kOffVperm + ((tile_i(s + 1, w) * 4 + g) << 2));
        }
        add_row(s, w, rs, m, vc);
        if (!is_diag_slot(s)) {
          double mm = m[0] * vr[0];
          mm = __builtin_fma(m[1], vr[1], mm);
          mm = __builtin_fma(m[2], vr[2], mm);
          mm = __builtin_fma(m[3], vr[3], mm);
          // (round 5: the four-slots-at-a-time reduction matvec() uses - store_mirrored4, 9 instructions for four sums
          // instead of 48 - was tried here: next to the four prefetched base tiles its four pending partials push the
          // kernel over its 256 registers, 392 spilled values, c4 7.0e5 -> 3.0e5 steps/s)
          mm = sum_over_g(mm);
          part[(16 * J + j) * PSTR + I] = mm;
        }
        // keep the prefetch distance: without it the scheduler sinks every load to just before its use
        __builtin_amdgcn_sched_barrier(0x206);  // arithmetic and LDS stores may cross, loads may not
      }
#pragma unroll
      for (int c = 0; c < NCLASS; ++c) {
        const double k = row_reduce16(rs[c], j);
        part[(16 * row_of_class(c, w) + 4 * (j >> 2) + g) * PSTR + 16] = k;
      }
      __syncthreads();
      double y = 0.0;
      if (tid < DPM) {
        const double* src = lds + kOffPart + tid * PSTR;
#pragma unroll
        for (int k = 0; k < PSTR; ++k) y += src[k];
        if constexpr (RMETRIC == MM_RMETRIC_RANK1) {
          double dot = lds[kOffRed];
#pragma unroll
          for (int k = 1; k < NWAVE; ++k) dot += lds[kOffRed + k];
          y = __builtin_fma(x, dot * inv_dim_, y);
        }
      }
      __syncthreads();
      return tid < dim ? y : 0.0;
    }
  }

  // B operands of column tile J: bx[kk] = X[4 kk + g][16 J + j] (the panel is published as Q - E already)
  __device__ static __forceinline__ d4 load_b(const double* X, const int J, const int g, const int j) {
    return *reinterpret_cast<const d4*>(X + (16 * J + j) * CS + 4 * g);
  }

  // ---- the 16 x 16 pivot block in accumulator layout -> T = -P^-1 (same layout), by a 4-wide symmetric sweep whose
  // rank-4 updates are single MFMAs.  Every wave does this redundantly on its own 512 bytes of LDS scratch.
  __device__ __forceinline__ void tile_sweep(d4& t, bool& ok, const int w, const int g, const int j) {
    double* scr = lds + kOffScr + w * 64;
#pragma unroll
    for (int R0 = 0; R0 < 4; ++R0) {
      scr[j * 4 + g] = t[R0];  // scr[c][s] = T[4 R0 + s][c]
      wave_sync();
      const d4 qv = *reinterpret_cast<const d4*>(scr + j * 4);  // my column's four pivot-row entries
      const d4 c0 = *reinterpret_cast<const d4*>(scr + (4 * R0 + 0) * 4);  // uniform: columns of the 4 x 4 pivot block
      const d4 c1 = *reinterpret_cast<const d4*>(scr + (4 * R0 + 1) * 4);
      const d4 c2 = *reinterpret_cast<const d4*>(scr + (4 * R0 + 2) * 4);
      const d4 c3 = *reinterpret_cast<const d4*>(scr + (4 * R0 + 3) * 4);
      wave_sync();
      // -W4 = -P4^-1 X4 column by column through the LDL^T factors of the 4 x 4 pivot block (uniform, ~36 FP64
      // instructions) and a 16-instruction substitution per lane: the in-tile sweep is bound by FP64 issue, and the
      // explicit closed-form inverse + product it replaces took ~75
      const double pa = c0[0], pb = c0[1], pc = c0[2], pd = c0[3];
      const double pe = c1[1], pf = c1[2], pg = c1[3], ph = c2[2], pi = c2[3], pj = c3[3];
      const double r1 = fast_rcp(pa);
      const double l21 = pb * r1, l31 = pc * r1, l41 = pd * r1;
      const double d2 = __builtin_fma(-l21, pb, pe);
      const double t32 = __builtin_fma(-l21, pc, pf), t42 = __builtin_fma(-l21, pd, pg);
      const double r2 = fast_rcp(d2);
      const double l32 = t32 * r2, l42 = t42 * r2;
      const double d3 = __builtin_fma(-l32, t32, __builtin_fma(-l31, pc, ph));
      const double t43 = __builtin_fma(-l32, t42, __builtin_fma(-l31, pd, pi));
      const double r3 = fast_rcp(d3);
      const double l43 = t43 * r3;
      const double d4v = __builtin_fma(-l43, t43, __builtin_fma(-l42, t42, __builtin_fma(-l41, pd, pj)));
      const double r4 = fast_rcp(d4v);
      // the pivots of the sequential elimination (all must be > 0: "Cholesky factorisation failed",
      // matrices.py:1170-1172; a NaN fails every comparison)
      ok = ok && (pa > 0.0) && (d2 > 0.0) && (d3 > 0.0) && (d4v > 0.0);
      d4 q = qv;  // X4 = Q4 - E4
      const int sdx = j - 4 * R0;
      q[0] -= (sdx == 0) ? 1.0 : 0.0;
      q[1] -= (sdx == 1) ? 1.0 : 0.0;
      q[2] -= (sdx == 2) ? 1.0 : 0.0;
      q[3] -= (sdx == 3) ? 1.0 : 0.0;
      const double y2 = __builtin_fma(-l21, q[0], q[1]);
      const double y3 = __builtin_fma(-l32, y2, __builtin_fma(-l31, q[0], q[2]));
      const double y4 = __builtin_fma(-l43, y3, __builtin_fma(-l42, y2, __builtin_fma(-l41, q[0], q[3])));
      const double w3 = -(y4 * r4);
      const double w2 = __builtin_fma(-l43, w3, -(y3 * r3));
      const double w1 = __builtin_fma(-l42, w3, __builtin_fma(-l32, w2, -(y2 * r2)));
      const double w0 = __builtin_fma(-l41, w3, __builtin_fma(-l31, w2, __builtin_fma(-l21, w1, -(q[0] * r1))));
      const double a_op = (g == 0) ? w0 : (g == 1) ? w1 : (g == 2) ? w2 : w3;
      const double b_op = (g == 0) ? q[0] : (g == 1) ? q[1] : (g == 2) ? q[2] : q[3];
      t = __builtin_amdgcn_mfma_f64_16x16x4f64(a_op, b_op, t, 0, 0, 0);
      if (sdx == g) t[R0] -= 2.0;
    }
  }

  // tiles S0 .. S0+N-1: acc += (-W rows)^T X, the N dependent chains interleaved
  template <int S0, int N, bool NOLOAD = false>
  __device__ __forceinline__ void update_group(const double* X, const int w, const int g, const int j,
                                               const d4 (&nw)[NCLASS]) {
    d4 bx[N], a[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      if constexpr (NOLOAD) bx[i] = nw[i & 1];
      else bx[i] = load_b(X, tile_j(S0 + i, w), g, j);
      a[i] = pick_row(S0 + i, w, nw);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
      for (int i = 0; i < N; ++i)
        acc[S0 + i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i][kk], bx[i][kk], acc[S0 + i], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);  // bound how far the next groups' operand loads are hoisted
  }
  // Forward substitution, one step, run inside the trailing sweeps after block I0's updates: y_K = b_K is final
  // once block K starts (every contribution to it came from this row's own tiles in earlier blocks); the tiles
  // (I, K) just became the finished factor tiles T_IK, so b_I -= T_IK y_K for this wave's rows below K, and
  // z_K = P_K^-1 y_K from the pivot row's own tile (-P_K^-1).
  __device__ __forceinline__ void forward_substitution_step(const int I0, const int w, const int g, const int j) {
    const int ib = 15 - w, ia = w;
    double* bv = lds + kOffB;
    const double yk = bv[16 * I0 + j];
    auto sub_row = [&](const int I, const d4 a) {
      d4 c;
#pragma unroll
      for (int k = 0; k < 4; ++k) c[k] = a[k] * yk;
      const int e = 16 * I + 4 * (j >> 2) + g;
      bv[e] = bv[e] - row_reduce16(c, j);  // the four lanes of a quad write the same value
    };
    auto diag_solve = [&](const d4 a) {
      d4 c;
#pragma unroll
      for (int k = 0; k < 4; ++k) c[k] = a[k] * yk;
      lds[kOffAux + 16 * I0 + 4 * (j >> 2) + g] = -row_reduce16(c, j);
    };
    if (I0 == ib) diag_solve(acc[0]);
    if (I0 == ia) diag_solve(acc[NSLOT - 1]);
    if (I0 < ib) {
      switch (ib - I0) {
#define MM_ROW(S) case S: sub_row(ib, acc[S]); MM_CASE_MARK(S); break;
        MM_ROW(1) MM_ROW(2) MM_ROW(3) MM_ROW(4) MM_ROW(5) MM_ROW(6) MM_ROW(7) MM_ROW(8) MM_ROW(9) MM_ROW(10)
        MM_ROW(11) MM_ROW(12) MM_ROW(13) MM_ROW(14) MM_ROW(15)
#undef MM_ROW
        default: break;
      }
    }
    if (I0 < ia) {
      switch (ia - I0) {
#define MM_ROW(D) case D: sub_row(ia, acc[16 - D]); MM_CASE_MARK(D); break;
        MM_ROW(1) MM_ROW(2) MM_ROW(3) MM_ROW(4) MM_ROW(5) MM_ROW(6) MM_ROW(7)
#undef MM_ROW
        default: break;
      }
    }
  }

  // ---- block-16 symmetric sweep.  TRAILING = false: every tile is updated by every block, tiles end as -M^-1.
  // TRAILING = true: only tiles (I, J) with J >= I0 - the blocked LDL^T: tile (K, K) = -P_K^-1, tile (I, K) =
  // A_IK P_K^-1.  `bad` = this wave's non-finite flag from build().  Returns "positive definite and finite" (uniform).
  // EXPER (timing experiments of tools/ubench_blk16.py only; results are wrong): 1 = no pivot-block inverse,
  // 2 = no tile updates, 3 = tile updates without their LDS operand loads
  template <bool TRAILING, bool PROF = false, int EXPER = 0>
  __device__ __forceinline__ bool sweep(const bool bad) {
    bool ok = true;
    long long pc[6] = {0, 0, 0, 0, 0, 0};  // PROF: cycles per phase, summed over the blocks
    if (fresh_lane() == 0) lds[kOffRed + 8 + wave] = bad ? 1.0 : 0.0;  // read by everyone after the barriers below
    // a do-while: the kernel is only launched for dim > 0, and a zero-trip bypass edge around this loop makes the
    // register allocator keep a second, untouched copy of all 136 tile registers alive across it
    int I0 = 0;
#pragma unroll 1
    do {
      const int w = opaque_wave(wave);
      const int ln = fresh_lane(), g = ln >> 4, j = ln & 15;
      const bool on_inv_simd = w == kInvWave || w == kInvWave - 4;  // waves w and w + 4 share a SIMD
      double* X = lds + kOffX + (I0 & 1) * (DPM * CS);
      long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0;
      if constexpr (PROF) c0 = __builtin_readcyclecounter();
      // (1) publish the panel X = Q - E: the owner of tile row I0 writes that row's tiles (the pivot block itself
      // minus the identity); every wave with a tile in tile column I0 (rows below) writes it transposed.  The slot of
      // that tile is a run-time value: a switch (binary search) instead of seventeen compare-and-branch pairs.
      {
        const int ib = 15 - w, ia = w;
        auto put_rowtile = [&](const int J, const d4 v) { *reinterpret_cast<d4*>(X + (16 * J + j) * CS + 4 * g) = v; };
        auto put_coltile = [&](const int I, const d4 v) {
          double* dst = X + (16 * I + g) * CS + (j & 3) * 4 + (j >> 2);  // + 4 r columns
#pragma unroll
          for (int r = 0; r < 4; ++r) dst[4 * r * CS] = v[r];
        };
        auto minus_identity = [&](d4 v) {
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] -= (j == 4 * r + g) ? 1.0 : 0.0;
          return v;
        };
        if (I0 == ib) {  // tile row 15-w is the pivot row
          put_rowtile(I0, minus_identity(acc[0]));
          if constexpr (!TRAILING) {
#pragma unroll
            for (int s = 1; s < NSLOT - 1; ++s)
              if (s <= 15 - w) put_rowtile(15 - w - s, acc[s]);
          }
        } else if (I0 < ib) {  // its tile in column I0: slot ib - I0 (1..15)
          switch (ib - I0) {
#define MM_PUT(S) case S: put_coltile(ib, acc[S]); MM_CASE_MARK(S); break;
            MM_PUT(1) MM_PUT(2) MM_PUT(3) MM_PUT(4) MM_PUT(5) MM_PUT(6) MM_PUT(7) MM_PUT(8) MM_PUT(9) MM_PUT(10)
            MM_PUT(11) MM_PUT(12) MM_PUT(13) MM_PUT(14) MM_PUT(15)
#undef MM_PUT
            default: break;
          }
        }
        if (I0 == ia) {  // tile row w is the pivot row
          put_rowtile(I0, minus_identity(acc[NSLOT - 1]));
          if constexpr (!TRAILING) {
#pragma unroll
            for (int s = 9; s < NSLOT - 1; ++s)
              if (s > 15 - w) put_rowtile(w + s - 16, acc[s]);
          }
        } else if (I0 < ia) {  // its tile in column I0: slot 16 - (ia - I0) (9..15)
          switch (ia - I0) {
#define MM_PUT(K) case K: put_coltile(ia, acc[16 - K]); MM_CASE_MARK(K); break;
            MM_PUT(1) MM_PUT(2) MM_PUT(3) MM_PUT(4) MM_PUT(5) MM_PUT(6) MM_PUT(7)
#undef MM_PUT
            default: break;
          }
        }
      }
      if constexpr (PROF) c1 = __builtin_readcyclecounter();
      __syncthreads();
      if constexpr (PROF) c2 = __builtin_readcyclecounter();
      // (2) T = -P^-1 by ONE wave, shared through LDS.  FP64 vector instructions and FP64 MFMAs run on the same units
      // of a SIMD and the in-tile sweep is bound by FP64 issue (~75 FP64 instructions per 4 x 4 sub-block): measured
      // with every wave inverting the block redundantly, the two waves of a SIMD took 5.7 k cycles against 3.3 k for
      // one wave on its own.
      d4 t;
      {
        double* tbuf = lds + kOffScr + 64;  // [64 lanes][4]
        if (w == kInvWave) {
          t = load_b(X, I0, g, j);
#pragma unroll
          for (int r = 0; r < 4; ++r) t[r] += (j == 4 * r + g) ? 1.0 : 0.0;
          bool okb = true;
          if constexpr (EXPER != 1) tile_sweep(t, okb, w, g, j);
          *reinterpret_cast<d4*>(tbuf + 4 * ln) = t;
          if (ln == 0) lds[kOffScr + 5 * 64] = okb ? 0.0 : 1.0;
        }
        // The forward substitution step of the PREVIOUS block runs here, in the window in which the other waves would
        // otherwise wait for the inverse (3.7 k cycles): its inputs - y_(K-1) and the factor tiles of column K-1 - are
        // final since the previous block's updates, and the next step's y_K is needed one block later.  (At the end of
        // the block that produced them it cost ~1.5 k cycles of every wave's critical path.)  Not on the inverting
        // wave's SIMD: FP64 vector work there would slow the inverse down; those two waves - the ones with the least
        // tile work in a trailing sweep - keep their step at the end of the block.
        if constexpr (TRAILING) {
          if (!on_inv_simd && I0 > 0) forward_substitution_step(I0 - 1, w, g, j);
        }
        __syncthreads();
        t = *reinterpret_cast<const d4*>(tbuf + 4 * ln);
        ok = ok && (lds[kOffScr + 5 * 64] == 0.0);
      }
      if constexpr (PROF) {
        asm volatile("" : "+v"(t));
        c3 = __builtin_readcyclecounter();
      }
      // (3) the four 16 x 16 blocks of -W = T X this wave's tile rows need, straight into A-operand registers:
      // lane (g, i), register kk <-> (-W)[4 kk + g][16 I + i]
      d4 nw[NCLASS];
#pragma unroll
      for (int c = 0; c < NCLASS; ++c) {
        const d4 bb = load_b(X, row_of_class(c, w), g, j);
        nw[c] = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) nw[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(t[kk], bb[kk], nw[c], 0, 0, 0);
      }
      if constexpr (PROF) {
        asm volatile("" : "+v"(nw[0]), "+v"(nw[1]));
        c4 = __builtin_readcyclecounter();
      }
      // (4) rank-16 update of the tiles: four MFMAs each.  The four MFMAs of a tile form a dependent chain
      // (measured: ~100 cycles per MFMA when a wave issues one tile after the other, against 64 for the matrix core),
      // so tiles are processed in groups whose chains are interleaved.
      if constexpr (EXPER == 2) {
      } else if constexpr (EXPER == 4) {
        // SIMD-isolation experiment: waves 0 and 4 (one SIMD) update nothing; wave 0 repeats the pivot-block inverse
        // WHILE the six other waves update their tiles, timed into the "-W" column
        if (w == 0) {
          d4 t2 = load_b(X, I0, g, j);
          bool ok2 = true;
          const long long a0 = __builtin_readcyclecounter();
          tile_sweep(t2, ok2, w, g, j);
          asm volatile("" : "+v"(t2));
          c4 = c3 + (__builtin_readcyclecounter() - a0);
          if (!ok2) lds[kOffScr + 6 * 64] = 1.0;
        } else if (w != 4) {
          update_group<0, 4>(X, w, g, j, nw);
          update_group<4, 4>(X, w, g, j, nw);
          update_group<8, 4>(X, w, g, j, nw);
          update_group<12, 5>(X, w, g, j, nw);
        }
      } else if constexpr (!TRAILING) {
        update_group<0, 4, EXPER == 3>(X, w, g, j, nw);
        update_group<4, 4, EXPER == 3>(X, w, g, j, nw);
        update_group<8, 4, EXPER == 3>(X, w, g, j, nw);
        update_group<12, 5, EXPER == 3>(X, w, g, j, nw);
      } else {
        // the active tiles (J >= I0) are a prefix of tile row 15-w's slots and a suffix of tile row w's: one
        // wave-uniform conditional arm per tile.  The four MFMAs of an arm are a dependent chain; the other wave of
        // the SIMD fills the matrix core in between.
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
          const int J = tile_j(s, w);
          if (J >= I0) {
            const d4 bx = load_b(X, J, g, j);
            const d4 a = pick_row(s, w, nw);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], bx[kk], acc[s], 0, 0, 0);
          }
        }
      }
      // A_KK -= 2 I on the pivot block's own tile
#pragma unroll
      for (int s = 0; s < NSLOT; ++s) {
        if (!is_diag_slot(s)) continue;
        if (tile_i(s, w) == I0) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (j == 4 * r + g) acc[s][r] -= 2.0;
        }
      }
      if constexpr (TRAILING) {
        if (on_inv_simd) forward_substitution_step(I0, w, g, j);
      }
      if constexpr (PROF) {
        const long long c5 = __builtin_readcyclecounter();
        pc[0] += c1 - c0;  // publish (includes waiting for the previous block's MFMA results)
        pc[1] += c2 - c1;  // barrier
        pc[2] += c3 - c2;  // pivot-block inverse by wave 0 + the barrier that hands T over
        pc[3] += c4 - c3;  // -W blocks
        pc[4] += c5 - c4;  // tile updates (issue)
        pc[5] += 1;
      }
      // no barrier here: the next block publishes into the other panel buffer
    } while (++I0 < nblk);
    if constexpr (TRAILING) {  // the last block's forward substitution step (solve() starts with a barrier)
      const int w = opaque_wave(wave);
      const int ln = fresh_lane(), g = ln >> 4, j = ln & 15;
      if (!(w == kInvWave || w == kInvWave - 4)) forward_substitution_step(nblk - 1, w, g, j);
    }
    if constexpr (PROF) {
      if (fresh_lane() == 0) {
        long long* dst = reinterpret_cast<long long*>(lds + kOffPart) + wave * 8;
#pragma unroll
        for (int k = 0; k < 6; ++k) dst[k] = pc[k];
      }
    }
    double flags = 0.0;
#pragma unroll
    for (int k = 0; k < NWAVE; ++k) flags += lds[kOffRed + 8 + k];
    return ok && flags == 0.0;
  }

  // (A look-ahead variant of the trailing sweep - block I0 first updates and publishes only the tiles of column I0+1,
  // then one wave inverts the next pivot block while the others finish block I0 - was built and measured: 287 k cycles
  // per sweep against 192 k.  FP64 vector instructions and FP64 MFMAs share the units of a SIMD, so every instruction
  // of the inverting wave's dependent chain queues behind an MFMA of the wave it shares its SIMD with.  Removed.)

  // ---- y = M^-1 v with the explicit inverse in the tiles (they hold -M^-1 after the full sweep) ----------------
  __device__ __forceinline__ double matvec(double v) {
    publish_vector(v);
    const int w = opaque_wave(wave);
    const int ln = fresh_lane(), g = ln >> 4, j = ln & 15;
    double* part = lds + kOffPart;
    d4 rs[NCLASS];
#pragma unroll
    for (int c = 0; c < NCLASS; ++c) rs[c] = d4{0.0, 0.0, 0.0, 0.0};
    // the vector operands of a slot are loaded one slot ahead: issued where they are used, each LDS round trip
    // (~130 cycles for the two waves of a SIMD) stood in front of the slot's eight multiply-adds
    double vc_next = lds[kOffNat + 16 * tile_j(0, w) + j];
    d4 vr_next = *reinterpret_cast<const d4*>(lds + kOffVperm + ((tile_i(0, w) * 4 + g) << 2));
    double mir[4] = {0.0, 0.0, 0.0, 0.0};  // mirrored partials of up to four slots, reduced together
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      const double vc = vc_next;
      const d4 vr = vr_next;
      if (s + 1 < NSLOT) {
        vc_next = lds[kOffNat + 16 * tile_j(s + 1, w) + j];
        vr_next = *reinterpret_cast<const d4*>(lds + kOffVperm + ((tile_i(s + 1, w) * 4 + g) << 2));
      }
      const d4 a = acc[s];
      add_row(s, w, rs, a, vc);
      if (!is_diag_slot(s)) {  // below the diagonal: the mirrored tile's rows are this tile's columns
        double m = a[0] * vr[0];
        m = __builtin_fma(a[1], vr[1], m);
        m = __builtin_fma(a[2], vr[2], m);
        m = __builtin_fma(a[3], vr[3], m);
        mir[(s - 1) & 3] = m;
        if (((s - 1) & 3) == 3 || s == NSLOT - 2)
          store_mirrored4(part, s - ((s - 1) & 3), w, g, j, mir[0], mir[1], mir[2], ((s - 1) & 3) == 3 ? mir[3] : 0.0);
      }
      __builtin_amdgcn_sched_barrier(0x206);  // arithmetic and LDS stores may cross, loads may not
    }
#pragma unroll
    for (int c = 0; c < NCLASS; ++c) {
      const double k = row_reduce16(rs[c], j);
      part[(16 * row_of_class(c, w) + 4 * (j >> 2) + g) * PSTR + 16] = k;
    }
    __syncthreads();
    double y = 0.0;
    if (tid < DPM) {
      // column-sum slots I <= (the element's own tile row) are never written and stay zero from kernel start
      const double* src = lds + kOffPart + tid * PSTR;
#pragma unroll
      for (int k = 0; k < PSTR; ++k) y += src[k];
    }
    // (the partial sums are next written behind the NEXT product's publish barrier, which every reader here reaches first:
    // the Woodbury kernels - whose only other users of LDS between two products are the flip reductions - do without this one)
    if constexpr (!kLowRank) __syncthreads();
    return tid < dim ? -y : 0.0;
  }

  // ---- two products with the held inverse in ONE pass over the register tiles (implicit_core.h lowrank_solve2) ---------------
  // matvec() with every tile register feeding both vectors.  The second vector's operand copies and partial sums live in the
  // idle panel (behind the Woodbury path's three flat vectors); its never-written partial slots are garbage after a sweep,
  // so both final sums mask them (slot k <= the element's own tile row) instead of relying on zeros.
  static constexpr int kOffNat2 = kOffRs + RS_COUNT * VLM;  // [VLM]
  static constexpr int kOffVperm2 = kOffNat2 + VLM;          // [DPM]
  static constexpr int kOffPart2 = kOffVperm2 + DPM;         // [DPM][PSTR]
  static_assert(kOffPart2 + DPM * PSTR <= kOffPart, "the lock step's second operand / partial sums must fit the panel buffers");
  static_assert((kOffVperm2 % 2) == 0, "16-byte alignment of the d4 accesses");
  __device__ __forceinline__ void matvec2_exact(double v0, double v1, double* y0, double* y1) {
    if (tid < DPM) {
      const double a = tid < dim ? v0 : 0.0, b = tid < dim ? v1 : 0.0;
      const int pi = ((((tid >> 4) << 2) + (tid & 3)) << 2) + ((tid >> 2) & 3);
      lds[kOffNat + tid] = a;
      lds[kOffVperm + pi] = a;
      lds[kOffNat2 + tid] = b;
      lds[kOffVperm2 + pi] = b;
    }
    __syncthreads();
    const int w = opaque_wave(wave);
    const int ln = fresh_lane(), g = ln >> 4, j = ln & 15;
    double* part = lds + kOffPart;
    double* part2 = lds + kOffPart2;
    d4 rs0[NCLASS], rs1[NCLASS];
#pragma unroll
    for (int c = 0; c < NCLASS; ++c) rs0[c] = rs1[c] = d4{0.0, 0.0, 0.0, 0.0};
    double mir0[4] = {0.0, 0.0, 0.0, 0.0}, mir1[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      const int I = tile_i(s, w), J = tile_j(s, w);
      const double vc0 = lds[kOffNat + 16 * J + j], vc1 = lds[kOffNat2 + 16 * J + j];
      const d4 a = acc[s];
      add_row(s, w, rs0, a, vc0);
      add_row(s, w, rs1, a, vc1);
      if (!is_diag_slot(s)) {
        const d4 vr0 = *reinterpret_cast<const d4*>(lds + kOffVperm + ((I * 4 + g) << 2));
        const d4 vr1 = *reinterpret_cast<const d4*>(lds + kOffVperm2 + ((I * 4 + g) << 2));
        double m0 = a[0] * vr0[0], m1 = a[0] * vr1[0];
#pragma unroll
        for (int r = 1; r < 4; ++r) {
          m0 = __builtin_fma(a[r], vr0[r], m0);
          m1 = __builtin_fma(a[r], vr1[r], m1);
        }
        mir0[(s - 1) & 3] = m0;
        mir1[(s - 1) & 3] = m1;
        if (((s - 1) & 3) == 3 || s == NSLOT - 2) {
          const bool full = ((s - 1) & 3) == 3;
          store_mirrored4(part, s - ((s - 1) & 3), w, g, j, mir0[0], mir0[1], mir0[2], full ? mir0[3] : 0.0);
          store_mirrored4(part2, s - ((s - 1) & 3), w, g, j, mir1[0], mir1[1], mir1[2], full ? mir1[3] : 0.0);
        }
      }
      __builtin_amdgcn_sched_barrier(0x206);  // arithmetic and LDS stores may cross, loads may not
    }
#pragma unroll
    for (int c = 0; c < NCLASS; ++c) {
      const int e = (16 * row_of_class(c, w) + 4 * (j >> 2) + g) * PSTR + 16;
      part[e] = row_reduce16(rs0[c], j);
      part2[e] = row_reduce16(rs1[c], j);
    }
    __syncthreads();
    double a = 0.0, b = 0.0;
    if (tid < DPM) {
      const double* s0 = lds + kOffPart + tid * PSTR;
      const double* s1 = lds + kOffPart2 + tid * PSTR;
      const int own = tid >> 4;  // column-sum slots k <= the element's own tile row are never written
#pragma unroll
      for (int k = 0; k < PSTR - 1; ++k) {
        const double p0 = s0[k], p1 = s1[k];
        a += (k > own) ? p0 : 0.0;
        b += (k > own) ? p1 : 0.0;
      }
      a += s0[PSTR - 1];
      b += s1[PSTR - 1];
    }
    *y0 = tid < dim ? -a : 0.0;
    *y1 = tid < dim ? -b : 0.0;
  }

  // ---- implicit_core.h lowrank_update: F += al a a^T + be (a b^T + b a^T) + ga b b^T on the tiles (they hold -F) ------------
  // entry (i, j) takes a_i u_j + b_i v_j with u = al a + be b, v = be a + ga b: the row operands a, b in the [I][g][r] order
  // (one 32-byte read a tile), the column operands u, v in natural order.  Eight multiply-adds a tile and lane.
  __device__ __forceinline__ void inverse_update(double al, double be, double ga, double a, double b) {
    if (tid < DPM) {
      const bool act = tid < dim;
      const int pi = (((tid >> 4) << 2) + (tid & 3)) << 2 | ((tid >> 2) & 3);
      lds[kOffVperm + pi] = act ? a : 0.0;
      lds[kOffB + pi] = act ? b : 0.0;
      lds[kOffNat + tid] = act ? __builtin_fma(al, a, be * b) : 0.0;
      lds[kOffAux + tid] = act ? __builtin_fma(be, a, ga * b) : 0.0;
    }
    __syncthreads();
    const int w = opaque_wave(wave);
    const int ln = fresh_lane(), g = ln >> 4, j = ln & 15;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      const int I = tile_i(s, w), J = tile_j(s, w);
      const d4 ar = *reinterpret_cast<const d4*>(lds + kOffVperm + ((I * 4 + g) << 2));
      const d4 br = *reinterpret_cast<const d4*>(lds + kOffB + ((I * 4 + g) << 2));
      const double uj = lds[kOffNat + 16 * J + j], vj = lds[kOffAux + 16 * J + j];
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[s][r] = __builtin_fma(-ar[r], uj, __builtin_fma(-br[r], vj, acc[s][r]));
      // (left alone the scheduler hoists the seventeen tiles' operand loads to the top, and some fifty values are spilled and
      // reloaded around this loop every step - 12.7 GB of scratch traffic a c4 launch, at L2 speed; a scheduling barrier every
      // two tiles makes the allocation worse, not better: A/B macro)
      if (MM_BLK16_UPD_BARRIER && (s & 1)) __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }

  // ---- u = M^-1 b from the trailing-sweep (LDL^T) factors: the backward substitution over the tile rows, one workgroup
  // barrier per tile row --------------------------------------------------------------------------------------------
  // (the forward and diagonal passes ran inside sweep<true>: aux holds z = D^-1 L^-1 b)
  __device__ __forceinline__ double solve() {
    const int w = opaque_wave(wave);
    const int ln = fresh_lane(), g = ln >> 4, j = ln & 15;
    double* nat = lds + kOffNat;
    double* aux = lds + kOffAux;
    double* part = lds + kOffPart;
    const int ib = 15 - w, ia = w;
    __syncthreads();
    // backward: u_K = z_K - sum_{I > K} T_IK^T u_I.  Step K: the owner of tile row K sums the column partials that
    // rows I > K left in part[16 K + j][I], publishes u_K, and leaves its own row's partials for the columns J < K.
#pragma unroll 1
    for (int K = nblk - 1; K >= 0; --K) {
      d4 ur = d4{0.0, 0.0, 0.0, 0.0};
      auto col_partial = [&](const d4 a, const int J) {
        double m = a[0] * ur[0];
        m = __builtin_fma(a[1], ur[1], m);
        m = __builtin_fma(a[2], ur[2], m);
        m = __builtin_fma(a[3], ur[3], m);
        part[(16 * J + j) * PSTR + K] = sum_over_g(m);
      };
      if (K == ib || K == ia) {
        double u = aux[16 * K + j];
        const double* src = part + (16 * K + j) * PSTR;
        // slots I <= K are never written (zero from kernel start), slots I >= nblk hold the zeros of padding tiles:
        // sixteen independent reads instead of a dependent loop over I = K+1 .. nblk-1
        double acc_u[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int I = 0; I < NT16; ++I) acc_u[I & 3] += src[I];
        u -= (acc_u[0] + acc_u[1]) + (acc_u[2] + acc_u[3]);
        // all four DPP rows computed the same u: natural order (the result) and the [I][g][r] copy
        nat[16 * K + j] = u;
        lds[kOffVperm + (((K << 2) + (j & 3)) << 2) + (j >> 2)] = u;
        wave_sync();
        ur = *reinterpret_cast<const d4*>(lds + kOffVperm + ((K * 4 + g) << 2));
        // before the barrier only what the NEXT step needs: the partial of this row's tile in column K - 1 (slot 1 of
        // tile row 15-w, slot 15 of tile row w); the rest of the row follows after the barrier, while the owner of
        // tile row K - 1 (another wave, except at K = 8 -> 7) is already at work
        if (K == ib) {
          if (ib >= 1) col_partial(acc[1], ib - 1);
        } else {
          if (ia >= 1) col_partial(acc[15], ia - 1);
        }
      }
      __syncthreads();
      if (K == ib) {  // tile row 15-w: slots 2..15-w, column J = ib - s
#pragma unroll
        for (int s = 2; s < NSLOT - 1; ++s)
          if (s <= 8 || s <= 15 - w) {
            if (s <= ib) col_partial(acc[s], ib - s);
          }
      } else if (K == ia) {  // tile row w: slots 16-w..14, column J = w + s - 16
#pragma unroll
        for (int s = 9; s < NSLOT - 2; ++s)
          if (s > 15 - w) col_partial(acc[s], w + s - 16);
      }
    }
    __syncthreads();
    const double u = (tid < dim) ? nat[tid] : 0.0;
    __syncthreads();
    return u;
  }

  // implicit_core.h, kUnifiedConstruct: metric_func(x) then either the explicit inverse (kept in the tiles for
  // matvec / half_vjp_inv / dh2_dpos) or the single solve u = M(x)^-1 rhs
  __device__ __forceinline__ bool construct(double x, bool need_inverse, double rhs, double* u) {
    // build() is instantiated inside each arm on purpose: with one shared copy in front of the branch the register
    // allocator gives the tiles one home for the full-sweep arm and spills the whole set to scratch for the other.
    // The distinct asm markers keep the optimiser from hoisting the common code back out.
    bool ok;
    if (need_inverse) {  // team-uniform
      asm volatile("; construct: explicit inverse");
      const bool bad = build(x);
      ok = sweep<false>(bad);
    } else {
      asm volatile("; construct: factor and solve");
      if (tid < DPM) lds[kOffB + tid] = tid < dim ? rhs : 0.0;  // visible after build()'s barrier
      const bool bad = build(x);
      ok = sweep<true>(bad);
      *u = solve();
    }
    return uniform_flag(ok);
  }
  __device__ __forceinline__ bool build_and_invert(double x) {
    double dummy;
    return construct(x, true, 0.0, &dummy);
  }
  __device__ __forceinline__ bool build_and_solve(double x, double rhs, double* u) { return construct(x, false, rhs, u); }

  __device__ __forceinline__ double diag() {  // diagonal of M^-1 (tiles hold -M^-1)
    const int w = opaque_wave(wave);
    const int ln = fresh_lane(), g = ln >> 4, j = ln & 15;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      if (!is_diag_slot(s)) continue;
      const int I = tile_i(s, w);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (j == 4 * r + g) lds[kOffNat + 16 * I + 4 * r + g] = -acc[s][r];
    }
    __syncthreads();
    const double y = (tid < dim) ? lds[kOffNat + tid] : 0.0;
    __syncthreads();
    return y;
  }

  // 0.5 * vjp_metric_func(q)(V) of a user metric.  q is the point of the held inverse: build() left it at kOffUq with its
  // aux block at kOffUaq.  OUTER: V = -u u^T, else the explicit inverse in the tiles - handed to the user's team-form hook
  // as it is (user_metric.h MM_USER_VJP_FLAT), or dumped to the chain's dense global array for V(i, j).
  template <bool OUTER>
  __device__ __forceinline__ double user_half_vjp(double u) {
    double r;
    const double* uq = lds + kOffUq;
    const double* uaq = lds + kOffUaq;
    if constexpr (mmuser::kFlatVjp) {
      if constexpr (OUTER) {
        mmuser::VjpOpsOuter<TeamBlk16> ops{*this, tid < dim ? u : 0.0};
        r = mmuser::vjp_flat(ops, uq, tid, dim, base, uaq);
      } else {
        mmuser::VjpOpsInv<TeamBlk16> ops{*this};
        r = mmuser::vjp_flat(ops, uq, tid, dim, base, uaq);
      }
    } else {
#if defined(MM_RTC_BUILD) && defined(MM_RTC_USER_METRIC)
      if constexpr (OUTER) {
        if (tid < DPM) lds[kOffAux + tid] = tid < dim ? u : 0.0;
        __syncthreads();
        const MmMat vm{nullptr, lds + kOffAux, 0};
        r = (tid < dim) ? mmuser::vjp_dense(uq, vm, tid, dim, base, uaq) : 0.0;
        __syncthreads();
      } else {
        const int w = opaque_wave(wave);
        const int ln = fresh_lane(), g = ln >> 4, j = ln & 15;
#pragma unroll
        for (int s = 0; s < NSLOT; ++s) {
          const int I = tile_i(s, w), J = tile_j(s, w);
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const double v = -acc[s][rr];  // (the tiles hold -M^-1)
            work[(16 * I + 4 * rr + g) * DPM + 16 * J + j] = v;
            work[(16 * J + j) * DPM + 16 * I + 4 * rr + g] = v;
          }
        }
        __syncthreads();  // (workgroup-scope release / acquire of the global stores)
        const MmMat vm{work, nullptr, DPM};
        r = (tid < dim) ? mmuser::vjp_dense(uq, vm, tid, dim, base, uaq) : 0.0;
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
      const double uq = team_reduce(tid < dim ? u * q : 0.0, 0, lds + kOffRed);
      return -(u * uq) * inv_dim_;
    } else {
      return -q * (u * u);
    }
  }
  __device__ __forceinline__ double grad(double q) {
    double* nat = lds + kOffNat;
    if (tid < VLM) nat[tid] = (tid < dim) ? q : 0.0;
    __syncthreads();
    int i = tid;
    asm volatile("" : "+v"(i));  // opaque: keeps the per-thread global addresses of the dense-Gaussian target's row
                                 // (tparams + i * dim, ...) from being hoisted out of the step loop into VGPRs
    const TargetAux aux = target_prepare<false>(target, nat, dim, tparams, i & 63);
    const double gr = (i < dim) ? target_grad_elem<false>(target, aux, nat, i, dim, tparams) : 0.0;
    __syncthreads();
    return gr;
  }
};

template <int RMETRIC, bool PROFILE, bool LOWRANK>
__device__ __forceinline__ void init_backend(TeamBlk16<RMETRIC, PROFILE, LOWRANK>& bk, const ImplicitArgs& A, double* lds) {
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __builtin_assume(wv >= 0 && wv < NWAVE);
  bk.wave = wv;
  bk.dim = A.dim;
  bk.inv_dim_ = 1.0 / (double)A.dim;
  bk.nblk = (A.dim + 15) >> 4;
  bk.tid.v = threadIdx.x;
  bk.target = A.target;
  bk.lds = lds;
  bk.base = A.rparams;
  bk.tparams = A.tparams;
  bk.work = A.work ? A.work + (int64_t)blockIdx.x * (DPM * DPM) : nullptr;
  bk.refine_on = A.no_refine == 0;
  bk.lr_refresh_ = A.lowrank_refresh;
  bk.rflip_ = 0;
  bk.dual_off = A.no_dual != 0;
  for (int i = threadIdx.x; i < DPM * PSTR; i += NTHR) lds[kOffPart + i] = 0.0;  // unused partial-sum slots stay 0
  if (threadIdx.x < 16) lds[kOffCnt + threadIdx.x] = 0.0;                    // work counters
  if constexpr (PROFILE) {
    if (threadIdx.x < PH_COUNT + 2)
      lds[kOffProf + threadIdx.x] = threadIdx.x == PH_COUNT + 1 ? (double)__builtin_readcyclecounter() : 0.0;
  }
  __syncthreads();
}

template <int RMETRIC, bool PROFILE = false, bool LOWRANK = false>
__device__ __forceinline__ void implicit_blk16_body(const ImplicitArgs& A, double* lds) {
  TeamBlk16<RMETRIC, PROFILE, LOWRANK> bk;
  init_backend(bk, A, lds);
  const int64_t chain = blockIdx.x;
  const int tid = threadIdx.x, dim = A.dim;
  const bool act = tid < dim;
  double q = act ? A.pos[chain * dim + tid] : 0.0;
  double p = act ? A.mom[chain * dim + tid] : 0.0;
  const double t = uniform_f64(signed_step(A.dir, A.step_scale, chain, A.step_size));
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
  if constexpr (PROFILE) {  // out[chain][PH_COUNT]: cycles per phase of this chain's launch
    bk.prof_switch(PH_OTHER);
    if (tid < PH_COUNT) A.out[chain * PH_COUNT + tid] = lds[kOffProf + tid];
  }
}

#ifndef MM_RTC_BUILD  // the in-tree instantiations (a run-time translation unit defines an extern "C" wrapper instead)
template <int RMETRIC, bool PROFILE = false, bool LOWRANK = false>
__global__ __launch_bounds__(NTHR, 2) void implicit_blk16_kernel(ImplicitArgs A) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  implicit_blk16_body<RMETRIC, PROFILE, LOWRANK>(A, lds);
}
#endif

}  // namespace mmblk16
