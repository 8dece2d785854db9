// Device code of the SoftAbs kernels (k_softabs.hip instantiates it for the built-in targets' Hessians; mm_rtc.hip compiles it
// at run time around a USER Hessian / matrix-Tressian product, user_hessian.h).
//
// Implicit leapfrog on SoftAbsRiemannianMetricSystem (D <= 256): one 1024-thread workgroup per chain, the Hessian /
// eigenvectors / work matrices in LDS for D <= 64 (the BASELINE c3(b) configuration), in a per-chain workspace in HBM for
// 64 < D <= 256 (SoftAbsBackendT<NP>: NP = 64 / 128 / 256).  gfx950 / CDNA4.
//
// Replaces, per chain and per step (reference /root/reference/src/mici):
//   SoftAbsRiemannianMetricSystem (metric = SoftAbs-regularised Hessian, vjp = matrix-Tressian
//       product)                                                  systems.py:1737-1920
//   SoftAbsRegularizedPositiveDefiniteMatrix: eigh, softabs, grad_softabs, grad_log_abs_det,
//       grad_quadratic_form_inv                                   matrices.py:1631-1685
//   EigendecomposedSymmetric / PositiveDefiniteMatrix: V diag(.) V^T products, inverse, sqrt,
//       "Eigenvalues must all be positive."                       matrices.py:1529-1628
//   the integrator step itself is implicit_core.h (integrators.py:493-544, solvers.py:47-154)
//
// eigh = refinement of the previous decomposition's eigenvectors by matrix products on the matrix cores (refine_eigh():
// Ogita-Aishima iteration, quadratic, 2.4 passes of four 64^3 products at c3(b); refine_eigh_global(): the same pass as
// tiled products from the workspace), falling back to a
// parallel cyclic ONE-SIDED (Hestenes) Jacobi on G = H V when there is no nearby basis: each round rotates D/2 disjoint
// column pairs of G and V, sweeps repeat until the columns of G are orthogonal (7-8 sweeps cold, ~2 warm-started).
// The result is used only through V f(lambda) V^T products, which do not depend on eigenvalue order or eigenvector signs.
// The matrix-Tressian products of the built-in targets need only the diagonal and the first row of
// their matrix argument, so V diag(g) V^T and A J A^T are never formed in full; the latter still needs
// the D^3 product B = A J (A = V diag(e)), on the matrix cores.  A USER Hessian gets both in full.
#pragma once
#include "implicit_core.h"
#include "user_hessian.h"

namespace mmsoftabs {

using namespace mmdev;
using namespace mmimp;

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int NT = 1024;           // 16 waves per chain: the kernel is LDS-latency bound, four waves per SIMD hide it
constexpr int TPD = 32;            // threads per matrix dimension in the NP x NP products (TPD^2 = NT)
constexpr int kMaxSweeps = 30;
constexpr int kWarmPeriod = 256;  // cold-start the eigenvector basis every this many decompositions

struct SaLds {
  double* H;    // Hessian -> V^T during eigh -> (after eigh) J matrix / scratch    [MATJ]
  double* V;    // eigenvectors (columns)
  double* W;    // G^T during eigh; A = V diag(e), then B = A J                    [MATJ]
  double* lam;  // unregularised eigenvalues
  double* lamt; // softabs eigenvalues
  double* gsa;  // grad_softabs(lam)
  double* v1;   // vectors
  double* v2;
  double* nat;
  double* qv;   // [NP] position argument of mtp_lds()
  double* tp;   // [NP] parameters of the target (weights of the funnel, coefficients of the polynomial)
  double* ring; // rotation (c, s) of a block round: [2 (parity)][4 (G wave)][15 (local round)][8 (pair slot)][2]
  double* red;  // [2][16]: per-wave partials of a workgroup reduction, two alternating sets
  double* cnt;  // [8] work counters of the chain (thread 0)
  double* prof; // [24] -DMM_SOFTABS_PROF: cycle stamps inside the Jacobi rounds [0..4], phases of the step [8..15]
  double* stash;  // [SL_COUNT][65]
  float* snap;    // NP = 64: [2][NP][NP] the step's two basis snapshots, in single precision (see basis_save)
  double* S;      // NP > 64 (global workspace): X^T A X of a refinement pass
  double* R;      // NP > 64: X^T X of a refinement pass
  double* X2;     // NP > 64: the refined basis of a pass (swapped with V at its end)
  double* Vt;     // NP > 64: V^T, kept beside V (row-major, LD): the operand a product walks ALONG the rows of V is read
  double* X2t;    //          down the columns of V^T instead - 16 lanes on one 128-byte run, not on 16 cache lines
  double* snapg;  // NP > 64: [2][NP][NP] the step's two basis snapshots (double precision: the workspace has the room)
  double* panel;  // NP > 64: [kPanelK][kPanelRows + NP] operand panels of staged_product (LDS, behind the ring)
};
constexpr int kRingDoubles = 15 * 8 * 2;  // (c, s) of the 15 local rounds x 8 pair slots of a block round

// A value every lane agrees on, moved to scalar registers: the step's control flow (implicit_core.h) and the Jacobi
// sweeps' termination depend only on team-uniform reductions; telling the compiler so keeps the state machine (mode,
// iteration counts, the time step) in SGPRs.  The kernel runs at four waves per SIMD (128 VGPRs) and used to spill
// 106 of them (340 bytes per lane of scratch).
__device__ __forceinline__ double uniform_f64(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffLL));
  const int hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// Sum / NaN-propagating maximum over the workgroup, the same value in every thread.  `red` holds two sets of 16 per-wave
// partials used alternately (`flip`, toggled by every call - all threads make the same calls): the next reduction
// writes the other set, and the one after that is behind a barrier every reader of this one has passed, so ONE
// workgroup barrier per reduction is enough (a barrier of this 16-wave team costs ~340 cycles).  The 16 partials are
// read by the 16 lanes of a DPP row and combined there.
__device__ __forceinline__ double block_reduce4(double v, int kind_max, double* red, int& flip) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* const set = red + 16 * flip;
  flip ^= 1;
  v = kind_max ? wave_max(v) : wave_sum(v);
  if (lane == 0) set[wave] = v;
  __syncthreads();
  double r = set[lane & 15];
  if (kind_max) {
    r = nanmax(r, dpp_move<kDppXor1>(r));
    r = nanmax(r, dpp_move<kDppXor2>(r));
    r = nanmax(r, dpp_move<kDppHalfMirror>(r));
    r = nanmax(r, dpp_move<kDppMirror>(r));
  } else {
    r += dpp_move<kDppXor1>(r);
    r += dpp_move<kDppXor2>(r);
    r += dpp_move<kDppHalfMirror>(r);
    r += dpp_move<kDppMirror>(r);
  }
  return uniform_f64(r);
}

// maxima without NaN propagation (v_max_f64 returns the other operand): over a 16-lane DPP row, over the wave
__device__ __forceinline__ double row_fmax(double v) {
  v = __builtin_fmax(v, dpp_move<kDppXor1>(v));
  v = __builtin_fmax(v, dpp_move<kDppXor2>(v));
  v = __builtin_fmax(v, dpp_move<kDppHalfMirror>(v));
  return __builtin_fmax(v, dpp_move<kDppMirror>(v));
}
__device__ __forceinline__ double wave_fmax(double v) {
  v = row_fmax(v);
  return __builtin_fmax(__builtin_fmax(readlane_f64(v, 0), readlane_f64(v, 16)),
                        __builtin_fmax(readlane_f64(v, 32), readlane_f64(v, 48)));
}

// sum over the RP (16 or 8) consecutive lanes that share an output element (one DPP row, or half of one)
template <int RP>
__device__ __forceinline__ double rp_sum_n(double v) {
  static_assert(RP == 16 || RP == 8 || RP == 4, "rp_sum reduces a DPP row, half of one or a quad");
  if constexpr (RP == 4) {
    v += dpp_move<kDppXor1>(v);
    return v + dpp_move<kDppXor2>(v);
  }
  v = group8_sum(v);
  return RP == 16 ? v + dpp_move<kDppMirror>(v) : v;
}

// -DMM_SOFTABS_PROF: per-phase cycle totals of block 0 (thread 0's clock), printed at the end of the launch
#ifdef MM_SOFTABS_PROF
#define SA_PROF_BEGIN() const long long prof_t0_ = __builtin_readcyclecounter()
#define SA_PROF_END(slot_) \
  do { if (tid_raw == 0) w.cnt[slot_] += (double)(__builtin_readcyclecounter() - prof_t0_); } while (0)
#define SA_PROF_END2(slot_) \
  do { if (tid_raw == 0) w.prof[slot_] += (double)(__builtin_readcyclecounter() - prof_t0_); } while (0)
#define SA_LAP(slot_) \
  do { const long long now_ = __builtin_readcyclecounter(); if (tid_raw == 0) w.prof[slot_] += (double)(now_ - lap_); \
       lap_ = now_; } while (0)
#define SA_LAP_BEGIN() long long lap_ = __builtin_readcyclecounter()
// -DMM_SOFTABS_PROF=2 adds stamps inside a Jacobi round of G wave 0 (lane 0 accumulates into prof[k]; scheduling
// barriers pin their place; they cost ~15 % themselves)
#if MM_SOFTABS_PROF >= 2
#define SA_STAMP(var_) \
  __builtin_amdgcn_sched_barrier(0); const long long var_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0)
#define SA_STAMP_ADD(k_, a_, b_) do { if (prof) prof[k_] += (double)((b_) - (a_)); } while (0)
#else
#define SA_STAMP(var_) do {} while (0)
#define SA_STAMP_ADD(k_, a_, b_) do {} while (0)
#endif
#else
#define SA_PROF_BEGIN() do {} while (0)
#define SA_PROF_END(slot_) do {} while (0)
#define SA_PROF_END2(slot_) do {} while (0)
#define SA_LAP(slot_) do {} while (0)
#define SA_LAP_BEGIN() do {} while (0)
#define SA_STAMP(var_) do {} while (0)
#define SA_STAMP_ADD(k_, a_, b_) do {} while (0)
#endif

// NP: the padded size of the problem, 64 (matrices in LDS: the BASELINE c3(b) configuration) or 128 (the same code
// with the three matrices in a per-chain global-memory workspace, for 64 < D <= 128: coverage of the reference's
// sizes, an order of magnitude slower per flop) or 256 (round 5, 128 < D <= 256: the same workspace layout, but a column
// pair no longer fits the registers of eight lanes at four waves per SIMD - the Jacobi rounds stream the columns from
// memory, block_round_mem(), and every wave rotates G and then V of its block pair: there are 16 block pairs a round and
// 16 waves.  1.6 MB of matrices per chain: with more chains in flight than fit the L2 the sweeps are HBM-bound)
// USERH: the Hessian and the matrix-Tressian product are the user's (user_hessian.h; any NP since round 5) - nothing is known about their
// structure: dense Hessian, G = A X as a matrix-core product, grad_log_abs_det / grad_quadratic_form_inv formed in full
template <int NP, bool USERH = false>
struct SoftAbsBackendT {
  static_assert(NP == 64 || NP == 128 || NP == 256, "SoftAbs backend sizes");
  static constexpr int BS = NP / TPD;     // output block side per thread in the NP x NP products
  static constexpr int RP = NT / NP;      // threads per output element of the row-wise reductions
  static constexpr int LD = NP + 1;       // leading dimension of the row-major matrices (LDS: conflict-free columns)
  static constexpr int MAT = NP * LD;
  static constexpr int LDJ = NP + 8;      // leading dimension of the column-major Jacobi work matrices (eigh())
  static constexpr int MATJ = NP * LDJ;
  static constexpr int NBLK = NP / 8;     // blocks of 8 columns in eigh()
  static constexpr int GW = NBLK / 2;     // waves rotating G (as many again replay on V)
  static constexpr int ROWS = NP / 8;     // rows per lane of a column pair
  static constexpr bool kMatricesInLds = NP == 64;
  // NP = 256: GW = 16 is every wave of the team - each rotates G and then V of its block pair (one ring set, no replay)
  static constexpr bool kBothRoles = 2 * GW > NT / 64;
  static constexpr int kRingSets = kBothRoles ? 1 : 2;
  static constexpr int kLdsVectors = 8 * NP + kRingSets * GW * kRingDoubles + 32 + 8 + 24 + SL_COUNT * (NP + 1);
  static_assert(kRingSets * GW * kRingDoubles >= (kMatricesInLds ? 1024 + 2 * NP : 2048 + NP),
                "vt_times() / mtp_lds() / dh2_dpos() scratch inside the ring");
  // NP = 64: the two single-precision snapshots of the eigenbasis (2 x 16 KB) start in the ring's tail - the ring is the
  // last of the vectors, its first kRingScratch doubles are scratch of the phases outside the Jacobi sweeps (dh2_dpos,
  // vt_times, refine_eigh), the rest is only touched by the sweeps, which invalidate the snapshots - and run on behind it
  static constexpr int kRingScratch = 1152;
  static constexpr int kSnapDoubles = NP == 64 ? NP * NP : 0;  // 2 x NP x NP floats
  static constexpr int kSnapExtra = NP == 64 ? kSnapDoubles - (2 * GW * kRingDoubles - kRingScratch) : 0;
  static_assert(NP != 64 || (2 * GW * kRingDoubles >= kRingScratch && kSnapExtra > 0), "ring / snapshot layout");
  // NP > 64 (round 6): operand panels of the refinement's products staged in LDS (staged_product): KP rows of the A operand
  // (the output rows of one pass: 128 columns) and of the B operand (NP columns)
  static constexpr int kPanelK = NP == 256 ? 16 : 32;
  static constexpr int kPanelRows = NP == 256 ? 128 : NP;  // output rows a pass of staged_product covers
  static constexpr int kPanelDoubles = kMatricesInLds ? 0 : kPanelK * (kPanelRows + NP);
  static constexpr int kLdsDoubles = kLdsVectors + (kMatricesInLds ? MAT + 2 * MATJ : 0) + kSnapExtra + kPanelDoubles;
  static_assert(kLdsDoubles * 8 <= 160 * 1024, "LDS budget of a CU");
  static constexpr int kWorkDoubles = kMatricesInLds ? 0 : 6 * MAT + 2 * MATJ + 2 * NP * NP;  // per chain, global memory (H, W, V; S, R, X2, Vt, X2t; snapshots)
  __device__ static __forceinline__ double rp_sum(double v) { return rp_sum_n<RP>(v); }

  static constexpr bool kSolveByInverse = true;   // implicit_core.h: ONE inlined copy of the construction (eigh: refinement + sweeps) instead of two
  static constexpr bool kUnifiedConstruct = false;
  static constexpr bool kCountersInLds = true;  // implicit_core.h: work counters in LDS, bumped by thread 0
  int dim, tid_raw, target;
  int warm = 0;  // eigendecompositions since the last cold start (0: w.V is not a usable basis)
  int n_sweeps = 0, n_eigh = 0;  // work counters (reported as n_newton_iters / n_eigh)
  bool j_valid = false;          // w.H holds the J matrix of the current eigenvalues (dh2_dpos)
  int red_flip = 0;              // which set of w.red the next workgroup reduction writes
  int snap_ok = 0;               // which of the step's two basis snapshots hold a converged basis
  int unchecked = 0;             // decompositions since refine_eigh() last measured X^T X
  int n_products = 0;            // NP^3 products run on the matrix cores (reported as n_mfma_products)
  int n_refined = 0;             // decompositions obtained by refine_eigh() alone (reported as n_refine)
  bool refine_on = true;         // MICI_AMD_REFINE=0: every decomposition by Jacobi sweeps
  double coeff;
  SaLds w;
  const double* tparams;
  const double* hparams;  // USERH: the user's parameters (what follows the SoftAbs coefficient in rmetric_params)

  // the thread index, re-materialised opaquely at every use: the per-thread global addresses derived from it
  // (tparams + tid, ...) are then computed where needed instead of being hoisted into long-lived VGPR pairs
  struct OpaqueTid {
    int v;
    __device__ __forceinline__ operator int() const {
      int x = v;
      asm volatile("" : "+v"(x));
      return x;
    }
  } tid;
  __device__ __forceinline__ void count(const int which, const int n) {
    if (tid == 0) w.cnt[which] += (double)n;  // exact in a double far beyond any launch's counts
  }
  __device__ __forceinline__ void read_counts(ChainResult& r) const {  // only thread 0's copy is used
    r.n_evals = (long long)w.cnt[CNT_EVALS];
    r.n_solves = (long long)w.cnt[CNT_SOLVES];
    r.n_metric = (long long)w.cnt[CNT_METRIC];
    r.n_grad = (long long)w.cnt[CNT_GRAD];
  }
  // flat state exists for tid < NP; the other threads share a dummy cell (index NP) per slot
  __device__ __forceinline__ double& slot(int i) { return w.stash[i * (NP + 1) + (tid < NP ? tid : NP)]; }

  __device__ __forceinline__ double norm(double x, int kind) {
    SA_PROF_BEGIN();
    const double a = tid < dim ? x : 0.0;
    double r;
    if constexpr (NP == 64) {
      // a flat vector lives in the first wave: it alone reduces (the other fifteen would reduce zeros on the SIMDs it
      // shares), everybody reads its result behind the one barrier
      double* const set = w.red + 16 * red_flip;
      red_flip ^= 1;
      if (tid < 64) {
        const double v = kind == MM_NORM_LINF ? wave_max(fabs(a)) : wave_sum(a * a);
        if (tid == 0) set[0] = v;
      }
      __syncthreads();
      r = uniform_f64(set[0]);
      if (kind != MM_NORM_LINF) r = sqrt(r);
    } else {
      r = kind == MM_NORM_LINF ? block_reduce4(fabs(a), 1, w.red, red_flip) : sqrt(block_reduce4(a * a, 0, w.red, red_flip));
    }
    SA_PROF_END2(13);
    return r;
  }

  // ---- hess_neg_log_dens(q) into w.H (systems.py:1870-1888); q flat --------------------------------
  __device__ __forceinline__ void build_hessian(double q) {
    SA_PROF_BEGIN();
    j_valid = false;
    if (tid < NP) w.nat[tid] = (tid < dim) ? q : 0.0;
    __syncthreads();
    // Both built-in Hessians are sparse (diagonal; arrowhead): the workgroup zero-fills all NP x NP entries (zero beyond
    // dim as well: the matrix-core products read whole tiles) and, behind a barrier, thread i < dim writes the entries
    // of row / column i.  (Every thread evaluating exp() and the branches of a dense fill kept all 16 waves' VALUs
    // busy for ~6 k cycles a call.)
    for (int el = tid; el < NP * NP; el += NT) {
      const int i = el / NP, j = el % NP;
      w.H[i * LD + j] = 0.0;
      if (warm == 0 && i < dim && j < dim) {
        w.V[i * LD + j] = (i == j) ? 1.0 : 0.0;
        if constexpr (!kMatricesInLds) w.Vt[i * LD + j] = (i == j) ? 1.0 : 0.0;
      }
    }
    __syncthreads();
    if constexpr (USERH) {  // the user's hess_neg_log_dens, entry by entry (four per thread), zero beyond dim
      for (int el = tid; el < NP * NP; el += NT) {
        const int i = el / NP, j = el % NP;
        w.H[i * LD + j] = mmuserh::hess_padded(w.nat, i, j, dim, hparams);
      }
    } else if (tid < NP) {  // whole waves
      const double* x = w.nat;
      const double* tp = w.tp;  // the target's parameters, staged in LDS by init_backend
      const int i = tid;
      if (target == MM_TARGET_POLY) {
        if (i < dim) w.H[i * LD + i] = tp[0] + 3.0 * tp[1] * x[i] * x[i];
      } else {  // funnel
        const double e = exp(-x[0]);
        const int k = (int)tid & 63;  // every wave forms the whole S = sum w x^2 (NP / 64 terms a lane)
        double acc = (k >= 1 && k < dim) ? tp[k - 1] * x[k] * x[k] : 0.0;
#pragma unroll
        for (int t = 64; t < NP; t += 64)
          if (k + t < dim) acc += tp[k + t - 1] * x[k + t] * x[k + t];
        const double s = wave_sum(acc);
        if (i == 0) {
          w.H[0] = 1.0 / 9.0 + 0.5 * e * s;
        } else if (i < dim) {
          const double a = -e * tp[i - 1] * x[i];
          w.H[i] = a;
          w.H[i * LD] = a;
          w.H[i * LD + i] = e * tp[i - 1];
        }
      }
    }
    __syncthreads();
    SA_PROF_END2(8);
  }

  // G = H V, written COLUMN-major (leading dimension LDJ) into w.W, rows and columns >= dim zeroed: the start of the one-sided
  // Jacobi.  With a cold start V = I and G = H.  Thread (ti, tj) owns rows {ti, ti + 32} x columns {2 tj, 2 tj + 1}:
  // the H reads of a 32-lane group are 32 consecutive rows (stride LD = 65 doubles: conflict-free), the V reads are
  // broadcasts and the G writes are contiguous.
  __device__ __forceinline__ void times_basis() {
    if constexpr (kMatricesInLds) {
      // on the matrix cores: wave t owns the 16 x 16 tile (t / 4, t % 4) of G; operands beyond dim are masked to zero
      // (H and V are only defined on dim x dim)
      const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
      const int g = lane >> 4, j = lane & 15;
      const int I = wave >> 2, Jt = wave & 3;
      ++n_products;
      d4 acc = {0.0, 0.0, 0.0, 0.0};
      const double* arow = w.H + (16 * I + j) * LD + g;   // H[16 I + m][4 kk + g], m = j
      const double* bcol = w.V + g * LD + 16 * Jt + j;    // V[4 kk + g][16 Jt + n], n = j
      const bool row_ok = 16 * I + j < dim, col_ok = 16 * Jt + j < dim;
#pragma unroll 4
      for (int kk = 0; kk < NP / 4; ++kk) {
        const bool k_ok = 4 * kk + g < dim;
        const double a = (row_ok && k_ok) ? arow[4 * kk] : 0.0;
        const double b = (col_ok && k_ok) ? bcol[4 * kk * LD] : 0.0;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) w.W[(16 * Jt + j) * LDJ + 16 * I + 4 * r + g] = acc[r];  // zero beyond dim by the masks
      __syncthreads();
      return;
    }
    const int ti = tid % TPD, bj = (tid / TPD) * BS;
    constexpr int BA = BS < 4 ? BS : 4;  // rows of the thread's block done at a time (NP = 256: 8 x 8 accumulators would spill)
#pragma unroll 1
    for (int a0 = 0; a0 < BS; a0 += BA) {
      double acc[BA][BS];
#pragma unroll
      for (int a = 0; a < BA; ++a)
#pragma unroll
        for (int b = 0; b < BS; ++b) acc[a][b] = 0.0;
      for (int k = 0; k < dim; ++k) {
        double hv[BA], vv[BS];
#pragma unroll
        for (int a = 0; a < BA; ++a) hv[a] = w.H[(ti + TPD * (a0 + a)) * LD + k];
#pragma unroll
        for (int b = 0; b < BS; ++b) vv[b] = w.V[k * LD + bj + b];
#pragma unroll
        for (int a = 0; a < BA; ++a)
#pragma unroll
          for (int b = 0; b < BS; ++b) acc[a][b] = __builtin_fma(hv[a], vv[b], acc[a][b]);
      }
#pragma unroll
      for (int a = 0; a < BA; ++a)
#pragma unroll
        for (int b = 0; b < BS; ++b) {
          const int row = ti + TPD * (a0 + a);
          w.W[(bj + b) * LDJ + row] = (row < dim && bj + b < dim) ? acc[a][b] : 0.0;
        }
    }
    __syncthreads();
  }

  // 1/sqrt(x) to rounding accuracy (x in the normal range): the hardware estimate (~2^-23) and ONE third-order
  // step y (1 + e/2 + 3 e^2/8), e = 1 - x y^2 - four dependent operations where two Newton steps are six, and this
  // sits on the critical path of every Jacobi rotation
  __device__ static __forceinline__ double rsqrt_newton(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double e = __builtin_fma(-x, y * y, 1.0);
    return __builtin_fma(y * e, __builtin_fma(0.375, e, 0.5), y);
  }

  // ---- eigh(H) by parallel ONE-SIDED (Hestenes) Jacobi: w.lam = eigenvalues, w.V = eigenvectors ----------------
  // Columns of G = H V and of V are rotated together until the columns of G are mutually orthogonal; then
  // H V = V diag(lam) with lam_i = g_i . v_i (which carries the sign: H is indefinite in general).
  //
  // What the layout is built around (measured, profiles/r02_c3b_jacobi_phases.txt): a workgroup barrier of this
  // 16-wave team costs ~340 cycles, and a rotation round is a dependent chain (LDS read -> dots -> reduction ->
  // rotation parameters -> rotate -> LDS write) that a wave cannot shorten by issuing faster.  So
  //  * a pair belongs to EIGHT lanes (8 rows each): the three dot products are 3-step DPP reductions and eight pairs
  //    share every wave instruction, so FOUR waves - one per SIMD - rotate G;
  //  * the 64 columns are 8 blocks of 8, each G wave owns TWO blocks for a "block round" and rotates all 64 cross
  //    pairs A_i x B_(i+k) (8 local rounds, k = 0..7) with no barrier at all: A_i stays in registers, only the B
  //    columns go through LDS, and the lanes of a wave are ordered by the in-order LDS queue.  The blocks are paired
  //    by a round-robin tournament (7 block rounds a sweep, barriers only there); the pairs inside a block are done in
  //    the first block round of a sweep (7 more local rounds).  A sweep is 63 local rounds and 7 barriers, where the
  //    flat tournament had 63 of each;
  //  * the G waves publish every rotation's (c, s) in an LDS ring; waves 4-7 (the second wave of each SIMD) replay
  //    the PREVIOUS block round on V, which nothing reads until the sweeps end.  Their rounds have no parameter chain,
  //    so they fill the issue slots the G wave of their SIMD leaves idle.
  // G^T and V^T live in LDS with a leading dimension of 72 doubles: the 32 lanes of a ds_read_b64 group (4 adjacent
  // pair slots x 8 rows) then hit 32 distinct 8-byte slots (8 ((col + j) mod 4) + row mod 8), and likewise the
  // 16-lane groups of ds_write_b64.  V is transposed into the dead H buffer on the way in and back on the way out.
  // Columns and rows beyond dim are zero: a pair with a zero column has gamma = 0 and is skipped, so the schedule is
  // the 64-column one for every dim.
  //
  // Warm start: consecutive metric constructions of a step are at nearby positions, so the previous eigenvectors
  // almost diagonalise the new Hessian (G = H V_prev is nearly orthogonal): ~2.5 sweeps instead of 7-8; a cold start
  // every kWarmPeriod decompositions bounds the accumulated loss of orthogonality of V.  Jacobi converges
  // quadratically, so a sweep whose largest |cos(g_p, g_q)| was below 1e-7 is the last one.  The result is used only
  // through V f(lam) V^T forms, which do not depend on eigenvalue order or eigenvector signs.

  // One 8-byte LDS access per element (byte offsets from the matrix base)
  __device__ static __forceinline__ void load_col(const char* M, int off, double (&x)[ROWS]) {
#pragma unroll
    for (int j = 0; j < ROWS; ++j) x[j] = *reinterpret_cast<const double*>(M + off + 64 * j);
  }
  __device__ static __forceinline__ void store_col(char* M, int off, const double (&x)[ROWS]) {
#pragma unroll
    for (int j = 0; j < ROWS; ++j) *reinterpret_cast<double*>(M + off + 64 * j) = x[j];
  }
  __device__ static __forceinline__ int col_offset(int col, int sub) { return (col * LDJ + sub) * 8; }

  // The rotation of one column pair held in registers.  GROLE: from the columns' dot products (and published);
  // otherwise the published one.  Returns whether the columns changed.
  // (c, s) must satisfy c^2 + s^2 = 1 to rounding - V stays orthogonal only then - which rules out the tempting
  // unnormalised form a - t b, b + t a with a low-precision t: it scales the two columns, the next rotation mixes a
  // scaled with an unscaled column, and the columns of G come out orthogonal without V being orthogonal.
  // TRACK (G role, cross pairs of a block round): the squared norms na = |xa|^2, nb = |xb|^2 come in with the columns
  // and are updated with the rotation (|a'|^2 = c^2 al - 2 c s ga + s^2 be, |b'|^2 = s^2 al + 2 c s ga + c^2 be), so a
  // round forms ONE dot product (and one 8-lane reduction) instead of three: the dots and their reductions were half of
  // a round's dependent chain.  The norms are recomputed from the columns at the start of every block round (eight
  // rounds), so rounding in the recurrence cannot accumulate; it only perturbs the rotation ANGLE at the 1e-15 level -
  // (c, s) stay normalised, and the convergence test compares ga^2 with al be at 1e-14 / 1e-30.
  template <bool GROLE, bool TRACK = false>
  __device__ static __forceinline__ bool rotate_pair(double (&xa)[ROWS], double (&xb)[ROWS], double* cs, bool writer,
                                                     double& big, double& bad, double* prof, double* na = nullptr,
                                                     double* nb = nullptr) {
    double c = 1.0, s = 0.0;
    if (GROLE) {
      SA_STAMP(t0);
      double al = 0.0, be = 0.0, ga = 0.0;
#pragma unroll
      for (int j = 0; j < ROWS; ++j) {
        if constexpr (!TRACK) {
          al = __builtin_fma(xa[j], xa[j], al);
          be = __builtin_fma(xb[j], xb[j], be);
        }
        ga = __builtin_fma(xa[j], xb[j], ga);
      }
      SA_STAMP(t1);
      if constexpr (TRACK) {
        al = *na;
        be = *nb;
      } else {
        al = group8_sum(al);
        be = group8_sum(be);
      }
      ga = group8_sum(ga);
      SA_STAMP(t2);
      SA_STAMP_ADD(0, t0, t1);
      SA_STAMP_ADD(1, t1, t2);
      const double ab = al * be, gg = ga * ga;
      if (!(ab <= 1.7e308) || !(gg <= 1.7e308)) bad = 1.0;  // NaN or overflow
      if (gg > 1e-14 * ab) big = 1.0;   // |cos| > 1e-7: another sweep is needed after this one
      if (gg > 1e-30 * ab) {            // |cos| > 1e-15 (threshold Jacobi; uniform over the 8 lanes)
        // tan 2 theta = 2 ga / (be - al), |theta| <= pi/4:  cos 2theta = |d| / r, sin 2theta = +-2 ga / r
        const double d = be - al;
        const double ri = rsqrt_newton(__builtin_fma(d, d, 4.0 * gg));
        const double c2 = __builtin_fma(0.5 * fabs(d), ri, 0.5);  // cos^2 theta, in [1/2, 1]
        const double rc = rsqrt_newton(c2);
        c = c2 * rc;
        s = ga * ri * rc;
        if (d < 0.0) s = -s;
        if (!(fabs(s) <= 1.0)) bad = 1.0;
        if constexpr (TRACK) {
          const double cc = c * c, ss = s * s, csg = 2.0 * c * s * ga;
          *na = __builtin_fma(cc, al, __builtin_fma(ss, be, -csg));
          *nb = __builtin_fma(ss, al, __builtin_fma(cc, be, csg));
        }
      }
      SA_STAMP(t3);
      SA_STAMP_ADD(2, t2, t3);
      if (writer) { cs[0] = c; cs[1] = s; }
    } else {
      c = cs[0];
      s = cs[1];
    }
    if (s == 0.0) return false;  // skipped pair (c = 1)
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
      const double a = xa[j], b = xb[j];
      xa[j] = __builtin_fma(c, a, -(s * b));
      xb[j] = __builtin_fma(s, a, c * b);
    }
    return true;
  }

  // One block round of a wave on the column-major matrix M (G^T or V^T): blocks ba, bb (8 columns each).
  // ring: [15][8][2] doubles of this wave for this block round.
  template <bool GROLE>
  __device__ static __forceinline__ void block_round(char* M, double* ring, int ba, int bb, bool intra, int slot,
                                                     int sub, double& big, double& bad, double* prof) {
    double xa[ROWS], xb[ROWS];
    const bool writer = sub == 0;
    if (intra) {  // the pairs inside each block: slots 0-3 on block ba, 4-7 on bb, a tournament of 8 in 7 rounds
      const int u = slot & 3, base = (slot < 4 ? ba : bb) * 8;
      for (int r = 0; r < 7; ++r) {
        int p, q;
        if (u == 0) { p = 7; q = r; }
        else {
          p = r + u; if (p >= 7) p -= 7;
          q = r - u; if (q < 0) q += 7;
        }
        const int oa = col_offset(base + p, sub), ob = col_offset(base + q, sub);
        load_col(M, oa, xa);
        load_col(M, ob, xb);
        if (rotate_pair<GROLE>(xa, xb, ring + (r * 8 + slot) * 2, writer, big, bad, nullptr)) {
          store_col(M, oa, xa);
          store_col(M, ob, xb);
        }
        wave_sync();
      }
    }
    // the 64 cross pairs: slot i keeps column i of ba in registers and meets column (i + k) mod 8 of bb in round k.
    // The bb columns stay in registers too: after a round every slot hands its column to the slot below it (lane
    // L takes lane L + 8's registers, ds_bpermute: one crossbar trip instead of an LDS store, a wait and a load).
    const int oa = col_offset(ba * 8 + slot, sub);
    load_col(M, oa, xa);
    load_col(M, col_offset(bb * 8 + slot, sub), xb);
    const int from = ((threadIdx.x + 8) & 63) << 2;  // byte address of the source lane for ds_bpermute
    double na = 0.0, nb = 0.0;  // G role: squared norms of the two columns, carried through the eight rounds
    if constexpr (GROLE) {
#pragma unroll
      for (int j = 0; j < ROWS; ++j) {
        na = __builtin_fma(xa[j], xa[j], na);
        nb = __builtin_fma(xb[j], xb[j], nb);
      }
      na = group8_sum(na);
      nb = group8_sum(nb);
    }
    auto hand_down = [&](double v) {
      const long long b = __double_as_longlong(v);
      const int lo = __builtin_amdgcn_ds_bpermute(from, (int)(b & 0xffffffffLL));
      const int hi = __builtin_amdgcn_ds_bpermute(from, (int)(b >> 32));
      return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    };
    for (int k = 0; k < 8; ++k) {
      SA_STAMP(ta);
      rotate_pair<GROLE, GROLE>(xa, xb, ring + ((7 + k) * 8 + slot) * 2, writer, big, bad, prof, &na, &nb);
      if (k < 7) {
#pragma unroll
        for (int r = 0; r < ROWS; ++r) xb[r] = hand_down(xb[r]);
        if constexpr (GROLE) nb = hand_down(nb);  // the norm travels with its column
      }
      SA_STAMP(td);
      SA_STAMP_ADD(3, ta, td);
    }
    // slot i ends with column (i + 7) mod 8 of bb
    store_col(M, oa, xa);
    store_col(M, col_offset(bb * 8 + ((slot + 7) & 7), sub), xb);
    wave_sync();
  }

  // NP = 256: the same block round with the columns left in memory.  A column pair is 2 x 32 doubles per lane of its
  // eight - the whole register budget of a wave at four per SIMD - so a pair's rotation is two passes over its columns:
  // the three dot products, then the rotation.  Same pairing, same order and the same ring layout as block_round().
  template <bool GROLE>
  __device__ static __forceinline__ void pair_mem(char* M, int ca, int cb, int sub, double* cs, bool writer, double& big,
                                                  double& bad) {
    char* const pa = M + col_offset(ca, sub);
    char* const pb = M + col_offset(cb, sub);
    double c = 1.0, s = 0.0;
    if (GROLE) {
      double al = 0.0, be = 0.0, ga = 0.0;
#pragma unroll 8
      for (int j = 0; j < ROWS; ++j) {
        const double a = *reinterpret_cast<const double*>(pa + 64 * j), b = *reinterpret_cast<const double*>(pb + 64 * j);
        al = __builtin_fma(a, a, al);
        be = __builtin_fma(b, b, be);
        ga = __builtin_fma(a, b, ga);
      }
      al = group8_sum(al);
      be = group8_sum(be);
      ga = group8_sum(ga);
      const double ab = al * be, gg = ga * ga;
      if (!(ab <= 1.7e308) || !(gg <= 1.7e308)) bad = 1.0;  // NaN or overflow
      if (gg > 1e-14 * ab) big = 1.0;
      if (gg > 1e-30 * ab) {  // (rotate_pair(): the same angle, the same normalised (c, s))
        const double d = be - al;
        const double ri = rsqrt_newton(__builtin_fma(d, d, 4.0 * gg));
        const double c2 = __builtin_fma(0.5 * fabs(d), ri, 0.5);
        const double rc = rsqrt_newton(c2);
        c = c2 * rc;
        s = ga * ri * rc;
        if (d < 0.0) s = -s;
        if (!(fabs(s) <= 1.0)) bad = 1.0;
      }
      if (writer) { cs[0] = c; cs[1] = s; }
    } else {
      c = cs[0];
      s = cs[1];
    }
    if (s != 0.0) {
#pragma unroll 8
      for (int j = 0; j < ROWS; ++j) {
        const double a = *reinterpret_cast<const double*>(pa + 64 * j), b = *reinterpret_cast<const double*>(pb + 64 * j);
        *reinterpret_cast<double*>(pa + 64 * j) = __builtin_fma(c, a, -(s * b));
        *reinterpret_cast<double*>(pb + 64 * j) = __builtin_fma(s, a, c * b);
      }
    }
    wave_sync();
  }
  template <bool GROLE>
  __device__ static __forceinline__ void block_round_mem(char* M, double* ring, int ba, int bb, bool intra, int slot,
                                                         int sub, double& big, double& bad) {
    const bool writer = sub == 0;
    if (intra) {
      const int u = slot & 3, base = (slot < 4 ? ba : bb) * 8;
#pragma unroll 1
      for (int r = 0; r < 7; ++r) {
        int p, q;
        if (u == 0) { p = 7; q = r; }
        else {
          p = r + u; if (p >= 7) p -= 7;
          q = r - u; if (q < 0) q += 7;
        }
        pair_mem<GROLE>(M, base + p, base + q, sub, ring + (r * 8 + slot) * 2, writer, big, bad);
      }
    }
#pragma unroll 1
    for (int k = 0; k < 8; ++k)
      pair_mem<GROLE>(M, ba * 8 + slot, bb * 8 + ((slot + k) & 7), sub, ring + ((7 + k) * 8 + slot) * 2, writer, big, bad);
  }

  // the two blocks of block-pair slot w in round R of the tournament of the NBLK blocks
  __device__ static __forceinline__ void blocks_of(int w, int R, int& ba, int& bb) {
    if (w == 0) { ba = NBLK - 1; bb = R; return; }
    ba = R + w; if (ba >= NBLK - 1) ba -= NBLK - 1;
    bb = R - w; if (bb < 0) bb += NBLK - 1;
  }

  // ---- eigh(H) from a nearby eigenvector basis, by matrix products (NP = 64, matrices in LDS) ----------------------
  // Consecutive decompositions of a step are at nearby positions, so X = V_prev almost diagonalises the new Hessian A.
  // One pass of the Ogita-Aishima refinement of an approximate eigenvector matrix
  //     S = X^T A X,  R = I - X^T X,  lam_i = S_ii / (1 - R_ii),
  //     E_ij = (S_ij + lam_j R_ij) / (lam_j - lam_i)  (i != j),   E_ii = R_ii / 2,   X <- X + X E
  // squares the error (rotation AND loss of orthogonality) and is four 64^3 products on the matrix cores - wave t owns
  // the 16 x 16 tile (t / 4, t % 4) of every product - plus five workgroup barriers: ~21 k cycles, the products at the
  // CU's FP64 rate (4.1 k cycles each), where a Jacobi sweep is 63 dependent rotation rounds, ~80 k.  c3(b): 1-5 passes
  // per decomposition, 2.4 on average (profiles/r03_c3b_phases.txt; tools/refine_eigh_proto.py replays the Hessians of
  // a chain on the CPU).
  // Pairs closer than kRefineGuard |A| are treated as a multiple eigenvalue (E_ij = R_ij / 2, any basis of their
  // invariant subspace will do); that is only valid if their coupling S_ij has vanished by the time the rest has
  // converged - otherwise, and whenever the first pass finds a rotation that is not small, the Jacobi sweeps take over:
  //   returns 1: done (w.V, w.lam);  0: w.V is still orthonormal, continue with warm-started sweeps;
  //          -1: w.V was updated but the passes stopped contracting - restart the sweeps from the identity.
  // The eigenvalues are the Rayleigh quotients of the last pass' input, whose error is the square of a rotation
  // below kRefineDone.
#ifndef MM_SA_REFINE_START  // (A/B runs: tools/ab_build.py, MICI_AMD_RTC_FLAGS)
#define MM_SA_REFINE_START 1.0
#endif
#ifndef MM_SA_REFINE_MAX_PASS
#define MM_SA_REFINE_MAX_PASS 8
#endif
  // largest first-pass |E_ij| the refinement is started from.  Measured (steps/s; Jacobi sweeps per step), c3(b) / c3b_dense:
  // 0.35: 5.69e5 (0.31) / 2.94e5 (4.6);  0.7: 5.76e5 / 3.09e5;  1.0: 5.82e5 (0.16) / 3.11e5 (2.7);  2.0: 5.81e5 / 3.02e5;  4.0: - / 2.97e5
  static constexpr double kRefineStart = MM_SA_REFINE_START;
  static constexpr double kRefineDone = 1e-7;    // a pass whose largest |E_ij| is below this is the last
  static constexpr double kRefineGuard = 1e-6;   // relative eigenvalue gap below which a pair counts as multiple
  static constexpr double kRefineSplit = 1e-11;  // largest |S_ij| / |A| tolerated inside such a pair at the end
  static constexpr int kRefineMaxPass = MM_SA_REFINE_MAX_PASS;
  static constexpr int kOrthoPeriod = 8;
  __device__ __forceinline__ int refine_eigh() {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, j = lane & 15;
    const int I = wave >> 2, J = wave & 3;
    const int ri = 16 * I + j, cj = 16 * J + j;  // row / column this lane addresses in the operand tiles
    const bool cj_ok = cj < dim;
    double* const X = w.V;
    double* const Gm = w.W;       // G = A X, then E (row-major, leading dimension LD)
    double* const part = w.ring;  // [2][16] per-wave maxima (the ring is idle outside the sweeps)
    if (dim < NP) {  // zero beyond dim, so that the operand loads of the products need no masks
      for (int el = tid; el < NP * NP; el += NT) {
        const int i = el / NP, c = el % NP;
        if (i >= dim || c >= dim) X[i * LD + c] = 0.0;  // (build_hessian zeroes H there)
      }
      __syncthreads();
    }
    // Operand addresses of this lane.  Lane group g takes the k = 16 g + kk terms of a product (any assignment of k to
    // the four groups is as good to the sum, as long as both operands use the same): the two groups of a 32-lane LDS
    // access are then 16 rows of LD = 65 doubles apart - 32 banks - and neither the loads that walk down a column (16
    // lanes on consecutive doubles) nor the ones that walk along a row (16 lanes a row apart) conflict.  A = H is
    // symmetric and is read down its columns.
    const double* const xcol_i = X + 16 * g * LD + ri;
    const double* const xcol_j = X + 16 * g * LD + cj;
    const double* const gcol_j = Gm + 16 * g * LD + cj;
    const double* const xrow_i = X + ri * LD + 16 * g;
    double prev = 0.0;
    ++unchecked;
    SA_LAP_BEGIN();
    for (int pass = 0; pass < kRefineMaxPass; ++pass) {
      SA_LAP(4);
      // G = A X.  The Hessians this backend builds (build_hessian) are diagonal or arrowhead - non-zero on the diagonal
      // and in row / column 0 only - and A X is formed from that: G_ij = A_i0 X_0j + A_ii X_ij for i > 0 on every lane
      // (the entries it owns in the tile layout); row 0 is a full-length dot product per column, four columns a wave.
      // 3 D^2 multiply-adds where the dense product (times_basis() has it, for the sweeps) is a quarter of a refinement
      // pass' matrix-core time.
      if constexpr (USERH) {
        // a dense Hessian: the tile (I, J) of A X on the matrix cores, A symmetric and read down its columns
        d4 ax = {0.0, 0.0, 0.0, 0.0};
        const double* const hcol_i = w.H + 16 * g * LD + ri;
#pragma unroll
        for (int kk = 0; kk < NP / 4; ++kk)
          ax = __builtin_amdgcn_mfma_f64_16x16x4f64(hcol_i[kk * LD], xcol_j[kk * LD], ax, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) Gm[(16 * I + 4 * r + g) * LD + cj] = ax[r];
        n_products += 2;  // A X and S
      } else {
        const double x0 = X[cj];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * I + 4 * r + g;
          if (i > 0) {
            const double g_ij = w.H[i * LD] * x0;
            Gm[i * LD + cj] = target == MM_TARGET_POLY ? w.H[i * LD + i] * X[i * LD + cj]
                                                        : __builtin_fma(w.H[i * LD + i], X[i * LD + cj], g_ij);
          }
        }
        {  // row 0, G_0c = sum_k A_0k X_kc: wave t forms columns 4 t .. 4 t + 3, sixteen lanes (four terms each) a column
          const int c = 4 * wave + (lane >> 4), k0 = 4 * (lane & 15);
          double a = w.H[k0] * X[k0 * LD + c];
#pragma unroll
          for (int t = 1; t < 4; ++t) a = __builtin_fma(w.H[k0 + t], X[(k0 + t) * LD + c], a);
          a = rp_sum_n<16>(a);
          if ((lane & 15) == 0) Gm[c] = a;
        }
        n_products += 1;  // S
      }
      __syncthreads();
      SA_LAP(0);
      // tiles of X^T G and X^T X.  The first pass takes X^T X = I: X is the result of the previous decomposition,
      // orthonormal to the square of its last rotation (< 1e-14), and a later pass repairs what the first adds to that.
      // Decompositions that end after their first pass never measure X^T X, so every kOrthoPeriod-th of those does.
      d4 s = {0.0, 0.0, 0.0, 0.0}, xx = {0.0, 0.0, 0.0, 0.0};
      const bool with_xx = pass > 0 || unchecked >= kOrthoPeriod;
      if (with_xx) {
        unchecked = 0;
        ++n_products;
#pragma unroll
        for (int kk = 0; kk < NP / 4; ++kk) {
          const double a = xcol_i[kk * LD];
          s = __builtin_amdgcn_mfma_f64_16x16x4f64(a, gcol_j[kk * LD], s, 0, 0, 0);
          xx = __builtin_amdgcn_mfma_f64_16x16x4f64(a, xcol_j[kk * LD], xx, 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int kk = 0; kk < NP / 4; ++kk)
          s = __builtin_amdgcn_mfma_f64_16x16x4f64(xcol_i[kk * LD], gcol_j[kk * LD], s, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) xx[r] = (16 * I + 4 * r + g == cj) ? 1.0 : 0.0;
      }
      if (I == J) {  // Rayleigh quotients from the diagonal tiles: element (4 r + g, j) of the tile is acc[r]
        // (the lane's one diagonal element selected first: four predicated IEEE divisions - ~30 dependent instructions each -
        // stood between the products and the barrier of every pass)
        double sn = 0.0, xn = 1.0;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (4 * r + g == j) {
            sn = s[r];
            xn = xx[r];
          }
        if ((j & 3) == g) w.lam[cj] = cj_ok ? fdiv(sn, xn) : 1.0;
      }
      __syncthreads();
      SA_LAP(1);
      // E, its largest entry and the largest coupling left inside a "multiple" pair.  A NaN anywhere (a non-finite
      // Hessian) must not be lost in the maxima: it is counted as an infinite rotation
      const double norm_a = uniform_f64(wave_fmax(lane < dim ? fabs(w.lam[lane]) : 0.0));
      const double lj = w.lam[cj];
      const double inf = __longlong_as_double(0x7ff0000000000000LL);
      double max_e = (norm_a == norm_a) ? 0.0 : inf, near_s = 0.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * I + 4 * r + g;
        const double li = w.lam[i];
        const double rij = (i == cj ? 1.0 : 0.0) - xx[r];
        const double gap = lj - li;
        const bool far = fabs(gap) > kRefineGuard * norm_a;
        double e = (i != cj && far) ? fdiv(__builtin_fma(lj, rij, s[r]), gap) : 0.5 * rij;
        // (Measured and not kept: the exact angle of the pair's 2 x 2 rotation, 2 e / (1 + sqrt(1 + 4 e^2)), instead of its
        // first-order value - every first pass can then start (|e| < 1), c3b_dense 2.7 -> 0.5 Jacobi sweeps a step, but the
        // passes that replace them cost as much: 3.11 -> 3.17e5 steps/s.)
        if (i >= dim || !cj_ok) e = 0.0;
        else if (i != cj && !far) near_s = __builtin_fmax(near_s, fabs(s[r]));
        max_e = __builtin_fmax(max_e, (e == e && s[r] == s[r]) ? fabs(e) : inf);
        Gm[i * LD + cj] = e;  // every wave is past its reads of G: the barrier above
      }
      max_e = wave_fmax(max_e);
      near_s = wave_fmax(near_s);
      if (lane == 0) {
        part[wave] = max_e;
        part[16 + wave] = near_s;
      }
      __syncthreads();
      max_e = uniform_f64(row_fmax(part[j]));  // the 16 lanes of a row read the 16 waves' values
      near_s = uniform_f64(row_fmax(part[16 + j]));
      if (pass == 0 && !(max_e < kRefineStart)) return 0;           // (NaN included) w.V untouched
      const bool last = max_e < kRefineDone;
      if (!last && pass > 0 && !(max_e < prev)) return -1;
      prev = max_e;
      SA_LAP(2);
      d4 acc;  // X' = X + X E
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] = X[(16 * I + 4 * r + g) * LD + cj];
#pragma unroll
      for (int kk = 0; kk < NP / 4; ++kk)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xrow_i[kk], gcol_j[kk * LD], acc, 0, 0, 0);
      ++n_products;
      __syncthreads();  // every wave has read the X it needs
#pragma unroll
      for (int r = 0; r < 4; ++r) X[(16 * I + 4 * r + g) * LD + cj] = acc[r];
      __syncthreads();
      SA_LAP(3);
#if defined(MM_SA_DBG_COUNT) && MM_SA_DBG_COUNT == 3  // (debug: 1000 in the sweep counter per split cluster)
      if (last && !(near_s <= kRefineSplit * norm_a)) n_sweeps += 1000;
#endif
      if (last) return (near_s <= kRefineSplit * norm_a) ? 1 : 0;  // 0: a split cluster the passes cannot resolve
    }
    return -1;
  }

  // ---- the same refinement with the matrices in the global workspace (NP = 128, 256; round 5) --------------------------
  // Until round 5 every decomposition beyond D = 64 was warm-started Jacobi sweeps - 2.6 a decomposition, each
  // NP (NP - 1) / 2 column rotations over memory: 33 ms a decomposition at D = 256.  The refinement pass is three
  // products on the matrix cores - the (NP / 16)^2 tiles dealt to the 16 waves, operands straight from memory (every
  // lane's operand loads are 128-byte runs of a row or a column; the L1 / L2 hold a chain's matrices) - around the same
  // element-wise E.  S, R and the refined basis have buffers of their own in the workspace: a wave owns several tiles, so
  // nothing is updated in place; the basis pointers are swapped at the end of a pass.  Same thresholds, same return
  // values as refine_eigh().
  // Four adjacent 16 x 16 tiles of a product at once: acc[u] += sum_k a[k] (x scale[k]) b[k][16 u + .], both operands walked
  // down columns (leading dimension LD), lane group g taking the terms k = (NP / 4) g + kk.  One a-operand load feeds four
  // matrix-core instructions with independent accumulators; with single tiles a wave had two loads per instruction in
  // flight eight at a time, and a pass was load-latency bound (1.6 ms a decomposition at D = 256).
  template <bool SCALED>
  __device__ static __forceinline__ void quad_tiles(const double* a0, const double* b0, const double* scale, d4 (&acc)[4]) {
#pragma unroll 4
    for (int kk = 0; kk < NP / 4; ++kk) {
      double a = a0[kk * LD];
      if constexpr (SCALED) a *= scale[kk];
#pragma unroll
      for (int u = 0; u < 4; ++u) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b0[kk * LD + 16 * u], acc[u], 0, 0, 0);
    }
  }

  // C = A^T B over the whole NP x NP output (both operands row-major with leading dimension LD, walked down their columns),
  // the operands STAGED THROUGH LDS (round 6): quad_tiles' waves each read their own 16-column strip of A and 64-column strip
  // of B from memory - 160 KB a quad, 10 MB a product at NP = 256 where the operands are 1 MB - and with 256 chains' 1.6 MB
  // of matrices each beyond the L2 that was HBM traffic: 1.5 GB a chain-step, 3.1 TB/s (profiles/r06_c3b_d256_pmc_hbm.json).
  // Here the workgroup loads kPanelK rows of both operands once (coalesced), every wave takes its k-steps' operands from
  // LDS (sixteen consecutive doubles a lane group: conflict-free), and a wave's output tiles - T / 2 of one tile row - stay
  // in its accumulators across the panels.  NP = 256: two passes of 128 output rows (eight accumulator tiles a wave: the
  // 128-register budget of a 1024-thread workgroup), the B panels read twice: 1.6 MB a product.
  // init(i, c) starts entry (i, c), store(i, c, v) takes it.  Ends with the panels free (a barrier) but NOT with the stores
  // visible: the caller's barrier does that.
  template <class InitF, class StoreF>
  __device__ __forceinline__ void staged_product(const double* __restrict__ A, const double* __restrict__ B, InitF init,
                                                 StoreF store) {
    constexpr int T = NP / 16, KP = kPanelK, RH = kPanelRows, HALVES = NP / RH, TPW = T / 2;
    static_assert(RH / 16 * 2 == NT / 64, "two waves a tile row of a pass");
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, j = lane & 15;
    double* const pa = w.panel;            // [KP][RH]
    double* const pb = w.panel + KP * RH;  // [KP][NP]
#pragma unroll 1
    for (int h = 0; h < HALVES; ++h) {
      const int I = (RH / 16) * h + (wave >> 1), J0 = TPW * (wave & 1);
      d4 acc[TPW];
#pragma unroll
      for (int u = 0; u < TPW; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[u][r] = init(16 * I + 4 * r + g, 16 * (J0 + u) + j);
#pragma unroll 1
      for (int p0 = 0; p0 < NP; p0 += KP) {
        __syncthreads();  // (the previous panel has been consumed)
        for (int el = tid; el < KP * RH; el += NT) {
          const int kk = el / RH, ii = el % RH;
          pa[el] = A[(p0 + kk) * LD + RH * h + ii];
        }
        for (int el = tid; el < KP * NP; el += NT) {
          const int kk = el / NP, c = el % NP;
          pb[el] = B[(p0 + kk) * LD + c];
        }
        __syncthreads();
        const double* const al = pa + g * RH + 16 * (I - (RH / 16) * h) + j;
        const double* const bl = pb + g * NP + 16 * J0 + j;
#pragma unroll
        for (int sidx = 0; sidx < KP / 4; ++sidx) {
          const double a = al[4 * sidx * RH];
#pragma unroll
          for (int u = 0; u < TPW; ++u)
            acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bl[4 * sidx * NP + 16 * u], acc[u], 0, 0, 0);
        }
      }
#pragma unroll
      for (int u = 0; u < TPW; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) store(16 * I + 4 * r + g, 16 * (J0 + u) + j, acc[u][r]);
    }
    __syncthreads();
  }

  __device__ __forceinline__ int refine_eigh_global() {
    double* const Gm = w.W;  // G = A X, then E (row-major, leading dimension LD)
    double prev = 0.0;
    ++unchecked;
    for (int pass = 0; pass < kRefineMaxPass; ++pass) {
      double* const X = w.V;
      if constexpr (USERH) {
        // a dense Hessian: G = A X as a tiled product, A symmetric and read down its columns
        staged_product(w.H, X, [](int, int) { return 0.0; }, [&](int i, int c, double v) { Gm[i * LD + c] = v; });
        ++n_products;
      } else {
      // G = A X from the structure of the built-in Hessians (diagonal; arrowhead), as refine_eigh()
      for (int el = tid; el < NP * NP; el += NT) {
        const int i = el / NP, c = el % NP;
        if (i == 0 && target != MM_TARGET_POLY) continue;
        const double d = w.H[i * LD + i] * X[i * LD + c];
        Gm[i * LD + c] = target == MM_TARGET_POLY ? d : __builtin_fma(w.H[i * LD], X[c], d);
      }
      if (target != MM_TARGET_POLY) {  // row 0: G_0c = sum_k A_0k X_kc, RP lanes a column
        const int c = tid / RP, part = tid % RP;
        double a = 0.0;
        for (int k = part; k < dim; k += RP) a = __builtin_fma(w.H[k], X[k * LD + c], a);
        a = rp_sum(a);
        if (part == 0) Gm[c] = a;
      }
      }
      ++n_products;
      __syncthreads();
      const bool with_xx = pass > 0 || unchecked >= kOrthoPeriod;
      if (with_xx) {
        unchecked = 0;
        ++n_products;
      }
      // S = X^T G, then R = X^T X (or the identity, see above)
      staged_product(X, Gm, [](int, int) { return 0.0; }, [&](int i, int c, double v) { w.S[i * LD + c] = v; });
      if (with_xx) {
        staged_product(X, X, [](int, int) { return 0.0; }, [&](int i, int c, double v) { w.R[i * LD + c] = v; });
      } else {
        for (int el = tid; el < NP * NP; el += NT) {
          const int i = el / NP, c = el % NP;
          w.R[i * LD + c] = i == c ? 1.0 : 0.0;
        }
      }
      __syncthreads();
      if (tid < NP) w.lam[tid] = tid < dim ? fdiv(w.S[tid * LD + tid], w.R[tid * LD + tid]) : 1.0;
      const double norm_a = block_reduce4(tid < dim ? fabs(w.lam[tid]) : 0.0, 1, w.red, red_flip);  // (w.lam visible behind its barrier)
      const double inf = __longlong_as_double(0x7ff0000000000000LL);
      double max_e = (norm_a == norm_a) ? 0.0 : inf, near_s = 0.0;
      for (int el = tid; el < NP * NP; el += NT) {
        const int i = el / NP, c = el % NP;
        const double sij = w.S[i * LD + c];
        const double li = w.lam[i], lj = w.lam[c];
        const double rij = (i == c ? 1.0 : 0.0) - w.R[i * LD + c];
        const double gap = lj - li;
        const bool far = fabs(gap) > kRefineGuard * norm_a;
        double e = (i != c && far) ? fdiv(__builtin_fma(lj, rij, sij), gap) : 0.5 * rij;
        if (i >= dim || c >= dim) e = 0.0;
        else if (i != c && !far) near_s = __builtin_fmax(near_s, fabs(sij));
        max_e = __builtin_fmax(max_e, (e == e && sij == sij) ? fabs(e) : inf);
        Gm[i * LD + c] = e;
      }
      max_e = block_reduce4(max_e, 1, w.red, red_flip);
      near_s = block_reduce4(near_s, 1, w.red, red_flip);
      if (pass == 0 && !(max_e < kRefineStart)) return 0;           // (NaN included) w.V untouched
      const bool last = max_e < kRefineDone;
      if (!last && pass > 0 && !(max_e < prev)) return -1;
      prev = max_e;
      ++n_products;
      // X' = X + X E into the other basis buffer (and its transpose): X[i][k] = X^T[k][i], the A operand is V^T
      {
        double* const x2 = w.X2;
        double* const x2t = w.X2t;
        staged_product(w.Vt, Gm, [&](int i, int c) { return X[i * LD + c]; }, [&](int i, int c, double v) {
          x2[i * LD + c] = v;
          x2t[c * LD + i] = v;
        });
      }
      __syncthreads();
      {
        double* const old = w.V;
        w.V = w.X2;
        w.X2 = old;
        double* const oldt = w.Vt;
        w.Vt = w.X2t;
        w.X2t = oldt;
      }
      if (last) return (near_s <= kRefineSplit * norm_a) ? 1 : 0;  // 0: a split cluster the passes cannot resolve
    }
    return -1;
  }

  __device__ __forceinline__ bool eigh() {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int role = wave / GW;   // 0: G, 1: V, beyond: only the barriers
    const int bw = wave % GW;     // block-pair slot of the wave
    const int slot = lane >> 3;   // pair slot of this group of 8 lanes
    const int sub = lane & 7;     // rows sub + 8 j
    bool converged = false;
    ++n_eigh;
    SA_PROF_BEGIN();
    {
      if (refine_on && warm > 0) {
        int rc;
        if constexpr (kMatricesInLds) rc = refine_eigh();
        else rc = refine_eigh_global();
        SA_PROF_END2(11);
        if (rc > 0) {
          ++n_refined;
          SA_PROF_END(4);
          return true;
        }
#ifdef MM_SA_DBG_COUNT  // (debug, tools/dbg/sa_handover.sh: 1000 in the sweep counter per warm hand-over (=1) / per restart (=2))
        n_sweeps += (MM_SA_DBG_COUNT < 3 && (rc < 0) == (MM_SA_DBG_COUNT == 2)) ? 1000 : 0;
#endif
        if (rc < 0) {  // restart from the identity (the Hessian is intact: the passes only read it)
          warm = 0;
          for (int el = tid; el < NP * dim; el += NT) {
            const int i = el / NP, j = el % NP;
            if (j < dim) {
              w.V[i * LD + j] = (i == j) ? 1.0 : 0.0;
              if constexpr (!kMatricesInLds) w.Vt[i * LD + j] = (i == j) ? 1.0 : 0.0;
            }
          }
          __syncthreads();
        }
      }
    }
    if constexpr (kMatricesInLds) snap_ok &= ~2;  // the sweeps use the whole (c, s) ring: snapshot 1, which starts in its tail, is gone
    times_basis();
    char* const G = reinterpret_cast<char*>(w.W);
    char* const Vt = reinterpret_cast<char*>(w.H);
    for (int e = tid; e < NP * NP; e += NT) {
      const int i = e % NP, j = e / NP;
      w.H[j * LDJ + i] = (i < dim && j < dim) ? w.V[i * LD + j] : 0.0;
    }
    __syncthreads();
    SA_PROF_END(5);
    if (!kBothRoles && role == 0) __builtin_amdgcn_s_setprio(3);  // the G waves' dependent chain is the critical path of a round
    int done = 0;  // block rounds the G role has done; the V role is one behind
    int Rprev = 0;
    for (int sweep = 0; sweep < kMaxSweeps; ++sweep) {
      double big = 0.0, bad = 0.0;
      ++n_sweeps;
      for (int R = 0; R < NBLK - 1; ++R) {
        int ba, bb;
        if constexpr (kBothRoles) {
          double b0 = 0.0, b1 = 0.0;
          blocks_of(wave, R, ba, bb);
          block_round_mem<true>(G, w.ring + wave * kRingDoubles, ba, bb, R == 0, slot, sub, big, bad);
          block_round_mem<false>(Vt, w.ring + wave * kRingDoubles, ba, bb, R == 0, slot, sub, b0, b1);
        } else if (role == 0) {
          blocks_of(bw, R, ba, bb);
#ifdef MM_SOFTABS_PROF
          double* const prof = tid_raw == 0 ? w.prof : nullptr;
#else
          double* const prof = nullptr;
#endif
          block_round<true>(G, w.ring + ((done & 1) * GW + bw) * kRingDoubles, ba, bb, R == 0, slot, sub, big, bad,
                            prof);
        } else if (role == 1 && done > 0) {
          double b0 = 0.0, b1 = 0.0;
          blocks_of(bw, Rprev, ba, bb);
          block_round<false>(Vt, w.ring + (((done - 1) & 1) * GW + bw) * kRingDoubles, ba, bb, Rprev == 0, slot, sub,
                             b0, b1, nullptr);
        }
        Rprev = R;
        ++done;
        __syncthreads();
      }
      bad = block_reduce4(bad, 0, w.red, red_flip);
      if (bad != 0.0) {
        __builtin_amdgcn_s_setprio(0);
        warm = 0;
        return false;
      }
      big = block_reduce4(big, 0, w.red, red_flip);
      if (big == 0.0) {
        converged = true;
        break;
      }
    }
    __builtin_amdgcn_s_setprio(0);
    if constexpr (!kBothRoles) {
      if (role == 1) {
        int ba, bb;
        double b0 = 0.0, b1 = 0.0;
        blocks_of(bw, Rprev, ba, bb);
        block_round<false>(Vt, w.ring + (((done - 1) & 1) * GW + bw) * kRingDoubles, ba, bb, Rprev == 0, slot, sub, b0,
                           b1, nullptr);
      }
    }
    __syncthreads();
    {  // lam_i = g_i . v_i, RP lanes per column
      const int i = tid / RP, part = tid % RP;
      double lam = 0.0;
#pragma unroll
      for (int m = 0; m < NP / RP; ++m)
        lam = __builtin_fma(w.W[i * LDJ + part + RP * m], w.H[i * LDJ + part + RP * m], lam);
      lam = rp_sum(lam);
      if (part == 0) w.lam[i] = (i < dim) ? lam : 1.0;
    }
    for (int e = tid; e < NP * NP; e += NT) {
      const int i = e % NP, j = e / NP;
      if (i < dim && j < dim) {
        w.V[i * LD + j] = w.H[j * LDJ + i];
        if constexpr (!kMatricesInLds) w.Vt[j * LD + i] = w.H[j * LDJ + i];
      }
    }
    __syncthreads();
    warm = converged ? (warm + 1) % kWarmPeriod : 0;
    SA_PROF_END(4);
    return converged;
  }

  // ---- the eigenvector basis carried from one launch to the next (NP = 64; round 5: every NP) -------------------------
  // A launch used to start every chain from the identity - seven or eight Jacobi sweeps, half a leapfrog step's time
  // - which is what an HMC transition with a short trajectory, its two Hamiltonian evaluations and its momentum draw
  // pay four times over.  The state keeps each chain's last basis in global memory (32 KB a chain, flagged valid only
  // when the last decomposition converged); any ORTHONORMAL basis is a legitimate starting point, however stale: the
  // refinement measures how far it is and hands over to the sweeps by itself.
  // per chain in global memory: the basis carried between launches, flag at kEigFlag.  (Round 3 kept the step's two
  // snapshots there as well: 128 KB written and read back per chain and step, 10 GB of HBM traffic per c3(b) launch.)
  static constexpr int kEigFlag = NP * NP;
  static constexpr int kEigDoubles = NP * NP + 8;
  static constexpr bool kBasisSlots = true;  // implicit_core.h: basis_save / basis_restore
  double* eig_mem = nullptr;  // this chain's kEigDoubles, or nullptr (refinement or carry-over switched off)
  __device__ __forceinline__ void copy_basis_out(double* dst) {
    for (int el = tid; el < NP * NP; el += NT) {
      const int i = el / NP, j = el % NP;
      if (i < dim && j < dim) dst[el] = w.V[i * LD + j];
    }
  }
  __device__ __forceinline__ void copy_basis_in(const double* src) {
    for (int el = tid; el < NP * NP; el += NT) {
      const int i = el / NP, j = el % NP;
      if (i < dim && j < dim) {
        w.V[i * LD + j] = src[el];
        if constexpr (!kMatricesInLds) w.Vt[j * LD + i] = src[el];
      }
    }
  }
  __device__ __forceinline__ void load_basis() {
    if (eig_mem == nullptr) return;
    if (uniform_f64(eig_mem[kEigFlag]) != 1.0) return;
    copy_basis_in(eig_mem);
    warm = 1;  // (visible to the team at the barrier every caller reaches before its first decomposition)
  }
  __device__ __forceinline__ void store_basis() {
    if (eig_mem == nullptr) return;
    if (warm > 0) copy_basis_out(eig_mem);
    if (tid == 0) eig_mem[kEigFlag] = warm > 0 ? 1.0 : 0.0;
  }
  // snapshots of the current basis inside a step (implicit_core.h).  Every thread reads back only what it wrote itself
  // (the same element loop both ways), so the global round trip needs no fence; the workgroup barrier orders w.V.
  // Round 4: the snapshots live in LDS, in SINGLE precision (2 x 16 KB: all that is left beside the three matrices).  A
  // snapshot is a starting basis, nothing more: rounded to 6e-8 it is still orthonormal to 1e-7, far inside what the
  // refinement starts from - but no longer to the square of a finished rotation, so the decomposition that starts from a
  // restored snapshot measures X^T X in its first pass instead of taking it for the identity (`unchecked`).
  // slot 1 (the basis at q + t M^-1 p) starts in the ring's tail; slot 0 (the basis at q) is the LAST 16 KB of the
  // region, clear of the ring: a fallback to the Jacobi sweeps only costs the step snapshot 1
  __device__ __forceinline__ float* snap_slot(const int slot) const { return w.snap + (slot == 0 ? NP * NP : 0); }
  __device__ __forceinline__ void basis_save(const int slot) {
    if (!refine_on || warm == 0) return;
    if constexpr (!kMatricesInLds) {  // round 5: the workspace tiers keep theirs in the workspace, in double precision
      double* const dst = w.snapg + slot * NP * NP;
      for (int el = tid; el < NP * NP; el += NT) dst[el] = w.V[(el / NP) * LD + el % NP];
      snap_ok |= 1 << slot;
      return;
    }
    float* const dst = snap_slot(slot);
    for (int el = tid; el < NP * NP; el += NT) dst[el] = (float)w.V[(el / NP) * LD + el % NP];  // (zero beyond dim)
    snap_ok |= 1 << slot;
  }
  __device__ __forceinline__ void basis_restore(const int slot) {
    if (!(snap_ok & (1 << slot))) return;
    __syncthreads();  // every reader of the current basis is done
    if constexpr (!kMatricesInLds) {
      const double* const src = w.snapg + slot * NP * NP;
      for (int el = tid; el < NP * NP; el += NT) {
        const double v = src[el];
        w.V[(el / NP) * LD + el % NP] = v;
        w.Vt[(el % NP) * LD + el / NP] = v;
      }
      __syncthreads();
      warm = warm > 0 ? warm : 1;
      return;  // (an exact copy of a converged basis: nothing to re-measure)
    }
    const float* const src = snap_slot(slot);
    for (int el = tid; el < NP * NP; el += NT) w.V[(el / NP) * LD + el % NP] = (double)src[el];
    __syncthreads();
    warm = warm > 0 ? warm : 1;
    unchecked = kOrthoPeriod;
  }

  // softabs(x) = x / tanh(coeff x); grad_softabs (matrices.py:1662-1669)
  __device__ __forceinline__ bool regularise() {
    SA_PROF_BEGIN();
    double bad = 0.0;
    if (tid < NP) {
      double lt = 1.0, gs = 0.0;
      if (tid < dim) {
        const double x = w.lam[tid], ax = coeff * x;
        const double th = tanh(ax), sh = sinh(ax);
        lt = x / th;
        gs = 1.0 / th - ax / (sh * sh);
        if (!(lt > 0.0)) bad = 1.0;  // "Eigenvalues must all be positive." (NaN included)
      }
      w.lamt[tid] = lt;
      w.gsa[tid] = gs;
    }
    bool ok;
    if constexpr (NP == 64) {  // the eigenvalues live in the first wave: it alone reduces (as norm())
      double* const set = w.red + 16 * red_flip;
      red_flip ^= 1;
      if (tid < 64) {
        const double v = wave_sum(bad);
        if (tid == 0) set[0] = v;
      }
      __syncthreads();
      ok = uniform_f64(set[0]) == 0.0;
    } else {
      ok = block_reduce4(bad, 0, w.red, red_flip) == 0.0;
    }
    SA_PROF_END2(9);
    return ok;
  }

  // The Jacobi sweeps and the refinement square and multiply the matrix' entries (column norms, Rayleigh quotients):
  // with |H_ij| beyond 1e150 - a state scaled by 1e75 on a quartic target - those overflow where LAPACK's eigh, which
  // scales its input, succeeds and the reference carries on to a diverging fixed-point solve (ConvergenceError, not
  // LinAlgError: tests/test_gpu_extreme_scale.py).  A Hessian whose largest entry lies outside [1e-100, 1e100] is
  // therefore decomposed as 2^-e H (an exact scaling; the eigenvectors are those of H) and the eigenvalues scaled back.
  // Done only AFTER a decomposition has failed (second trip of build_and_invert's loop): in the common path it would be a
  // pass over the matrix and a workgroup reduction per construction, and ten more spilled registers - c3(b) 5.82e5 ->
  // 5.56e5 steps/s when it ran every time.  Returns the factor; 1: nothing to rescale, the failure stands.
  __device__ __forceinline__ double rescale_hessian() {
    double m = 0.0;
    for (int el = tid; el < NP * NP; el += NT) m = __builtin_fmax(m, fabs(w.H[(el / NP) * LD + el % NP]));
    m = block_reduce4(m, 1, w.red, red_flip);
    const bool scale = (m > 1e100 || (m < 1e-100 && m > 0.0)) && m < 1.7e308;  // (not: in range, all zero, not finite)
    if (!scale) return 1.0;
    const int e = ilogb(m);
    const double down_a = ldexp(1.0, -(e / 2)), down_b = ldexp(1.0, -(e - e / 2));  // (2^-e itself can be subnormal)
    for (int el = tid; el < NP * NP; el += NT) {
      double& h = w.H[(el / NP) * LD + el % NP];
      h = (h * down_a) * down_b;
    }
    __syncthreads();
    return ldexp(1.0, e);
  }

  // OFF by default (-DMM_SA_RESCALE=1 builds it): even as a second trip that the common path never takes, the retry
  // costs the 128-register kernel ten more spilled values - c3(b) 5.82e5 -> 5.69e5 steps/s, c3b_dense 3.08e5 -> 3.02e5 -
  // for states scaled beyond 1e75.  Without it such a chain stops with LinAlgError where the reference stops with
  // ConvergenceError (same step, different class): tests/test_gpu_extreme_scale.py pins exactly that deviation.
#ifndef MM_SA_RESCALE
#define MM_SA_RESCALE 0
#endif
  __device__ __forceinline__ bool build_and_invert(double x) {
    if constexpr (!MM_SA_RESCALE) {
      build_hessian(x);
      if (!eigh()) return false;
      return regularise();
    }
    bool ok = false;
#pragma unroll 1
    for (int attempt = 0; attempt < 2 && !ok; ++attempt) {  // team-uniform
      build_hessian(x);
      double hs = 1.0;
      if (attempt == 1) {
        hs = rescale_hessian();
        if (hs == 1.0) break;
      }
      ok = eigh();
      if (ok && attempt == 1) {
        if (tid < dim) w.lam[tid] *= hs;
        __syncthreads();
      }
      ok = ok && regularise();
    }
    return ok;
  }

  // V^T v (flat in, flat out): RP threads share output k, each sums every RP-th term.  Input and output go through
  // two buffers of their own (in the ring, idle outside eigh()), so a call needs two workgroup barriers, not four: the
  // next writer of either buffer is always behind a barrier that every reader of it has passed.
  __device__ __forceinline__ double vt_times(double v) {
    double* const vin = w.ring + 1024;
    double* const vout = vin + NP;
    if (tid < NP) vin[tid] = (tid < dim) ? v : 0.0;
    __syncthreads();
    {
      const int k = tid / RP, part = tid % RP;
      double s = 0.0;
      if (k < dim)
        for (int i = part; i < dim; i += RP) s = __builtin_fma(w.V[i * LD + k], vin[i], s);
      s = rp_sum(s);
      if (part == 0) vout[k] = s;
    }
    __syncthreads();
    return (tid < dim) ? vout[tid] : 0.0;
  }
  // V v
  __device__ __forceinline__ double v_times(double v) {
    double* const vin = w.ring + 1024;
    double* const vout = vin + NP;
    if (tid < NP) vin[tid] = (tid < dim) ? v : 0.0;
    __syncthreads();
    {
      const int i = tid / RP, part = tid % RP;
      double s = 0.0;
      if (i < dim)
        for (int k = part; k < dim; k += RP) s = __builtin_fma(w.V[i * LD + k], vin[k], s);
      s = rp_sum(s);
      if (part == 0) vout[i] = s;
    }
    __syncthreads();
    return (tid < dim) ? vout[tid] : 0.0;
  }

  // M^-1 v = V diag(1/lamt) V^T v   (matrices.py:1568-1575, 1623-1624)
  __device__ __forceinline__ bool build_and_solve(double x, double rhs, double* u) {
    const bool ok = build_and_invert(x);
    *u = matvec(rhs);
    return ok;
  }

  __device__ __forceinline__ double matvec(double v) {
    SA_PROF_BEGIN();
    const double c = vt_times(v);
    const double u = v_times(tid < dim ? mmdev::rcp_nr(w.lamt[tid]) * c : 0.0);
    SA_PROF_END2(10);
    return u;
  }

  // mtp_neg_log_dens(q)(m) given only what the built-in Tressians touch: m_ii and the symmetric first row m_0i.
  // systems.py:1890-1920; closed forms SURVEY.md Appendix A.  The arguments are in LDS - w.qv = q, w.v2 = m_ii,
  // w.nat = m_0i, visible to every thread (the caller's barrier); the result is flat.  For NP = 64 the first wave does
  // all of it without a barrier - so whoever writes those vectors NEXT from another wave must be behind a barrier of
  // its own (half_vjp_inv() has one for that; dh2_dpos() and everything else reach one before they write).
  __device__ __forceinline__ double mtp_lds() {
    if (NP == 64 && tid >= NP) return 0.0;  // the flat result lives in the first wave, and so does all the work
    const double qi = (tid < dim) ? w.qv[tid] : 0.0;
    const double md = (tid < dim) ? w.v2[tid] : 0.0;
    if (target == MM_TARGET_POLY) return 6.0 * w.tp[1] * qi * md;
    // funnel: q = (v, x)
    const double m0 = (tid < dim) ? w.nat[tid] : 0.0;
    const double ev = exp(-w.qv[0]);
    const double mvv = w.v2[0];
    const double wi = (tid >= 1 && tid < dim) ? w.tp[tid - 1] : 0.0;
    double a1 = wi * qi * qi;        // S
    double a2 = 2.0 * m0 * wi * qi;  // (m_vi + m_iv) w_i x_i
    double a3 = md * wi;             // m_ii w_i
    double S, s2, s3;
    if constexpr (NP == 64) {  // one wave holds every term: no barrier
      S = wave_sum(a1);
      s2 = wave_sum(a2);
      s3 = wave_sum(a3);
    } else {
      // three sums behind one barrier: two alternating sets of per-wave partials, as block_reduce4
      double* const red3 = w.ring + 512 + 48 * red_flip;  // [16][3]; the ring is idle outside eigh()
      red_flip ^= 1;
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
      a1 = wave_sum(a1);
      a2 = wave_sum(a2);
      a3 = wave_sum(a3);
      if (lane == 0) { red3[3 * wave] = a1; red3[3 * wave + 1] = a2; red3[3 * wave + 2] = a3; }
      __syncthreads();
      const int k = lane & 15;
      S = uniform_f64(rp_sum_n<16>(red3[3 * k]));
      s2 = uniform_f64(rp_sum_n<16>(red3[3 * k + 1]));
      s3 = uniform_f64(rp_sum_n<16>(red3[3 * k + 2]));
    }
    if (tid == 0) return -0.5 * ev * S * mvv + ev * s2 - ev * s3;
    return ev * wi * qi * mvv - ev * wi * (2.0 * m0);  // (zero beyond dim: wi = 0)
  }

  // 0.5 * mtp(grad_log_abs_det), grad_log_abs_det = V diag(grad_softabs(lam)/lamt) V^T  (:1671-1674)
  // the user's matrix-Tressian product of the symmetric matrix the team has just formed in w.W (row-major, leading
  // dimension LD, zero beyond dim): element k on thread k - the first wave
  __device__ __forceinline__ double user_mtp_of(const double* m) {
    __syncthreads();  // the matrix complete, w.qv visible
    double out = 0.0;
    if (tid < dim) out = mmuserh::mtp(w.qv, m, LD, tid, dim, hparams);
    __syncthreads();  // (the next writer of the matrix / w.qv is behind this)
    return out;
  }
  __device__ __forceinline__ double user_mtp_w() { return user_mtp_of(w.W); }

  __device__ __forceinline__ double half_vjp_inv(double q) {
    SA_PROF_BEGIN();
    if (tid < NP) w.qv[tid] = (tid < dim) ? q : 0.0;
    if constexpr (USERH && !kMatricesInLds) {
      // the same matrix from the workspace: tile (I, J) = sum_k V^T[k][i] (g_k) V^T[k][j], both operands down columns of V^T
      constexpr int T = NP / 16, KQ = NP / 4, QR = T / 4, NQ = T * QR, NW = NT / 64;
      const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
      const int g = lane >> 4, j = lane & 15;
      if (tid < NP) w.v2[tid] = (tid < dim) ? w.gsa[tid] / w.lamt[tid] : 0.0;  // (w.v2: only mtp_lds() uses it otherwise)
      __syncthreads();
      ++n_products;
#pragma unroll 1
      for (int qd = wave; qd < NQ; qd += NW) {
        const int I = qd / QR, J0 = 4 * (qd % QR);
        d4 acc[4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
        quad_tiles<true>(w.Vt + KQ * g * LD + 16 * I + j, w.Vt + KQ * g * LD + 16 * J0 + j, w.v2 + KQ * g, acc);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int r = 0; r < 4; ++r) w.W[(16 * I + 4 * r + g) * LD + 16 * (J0 + u) + j] = acc[u][r];
      }
      const double out = 0.5 * user_mtp_w();
      SA_PROF_END(7);
      return out;
    } else if constexpr (USERH) {
      // grad_log_abs_det = V diag(softabs'(lam) / softabs(lam)) V^T in full: tile (I, Jt) = sum_k V_ik g_k V_jk on the
      // matrix cores (both operands walk along rows of V: conflict-free at LD = 65), into w.W
      const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
      const int g = lane >> 4, j = lane & 15;
      const int I = wave >> 2, Jt = wave & 3;
      d4 acc = {0.0, 0.0, 0.0, 0.0};
      ++n_products;
      const double* vrow_i = w.V + (16 * I + j) * LD + 16 * g;
      const double* vrow_j = w.V + (16 * Jt + j) * LD + 16 * g;
      const double* gs = w.gsa + 16 * g;
      const double* lt = w.lamt + 16 * g;
#pragma unroll
      for (int kk = 0; kk < NP / 4; ++kk)
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(vrow_i[kk] * (gs[kk] / lt[kk]), vrow_j[kk], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) w.W[(16 * I + 4 * r + g) * LD + 16 * Jt + j] = acc[r];
      const double out = 0.5 * user_mtp_w();
      SA_PROF_END(7);
      return out;
    }
    {
      const int i = tid / RP, part = tid % RP;
      double md = 0.0, m0 = 0.0;
      if (i < dim) {
        for (int k = part; k < dim; k += RP) {
          const double g = fdiv(w.gsa[k], w.lamt[k]);
          const double vik = w.V[i * LD + k];
          md = __builtin_fma(vik * vik, g, md);
          m0 = __builtin_fma(w.V[k] * vik, g, m0);  // V[0][k] V[i][k] g_k
        }
      }
      md = rp_sum(md);
      m0 = rp_sum(m0);
      __syncthreads();  // (a previous mtp_lds() may still be reading in the first wave)
      if (part == 0) { w.v2[i] = md; w.nat[i] = m0; }
    }
    __syncthreads();
    const double out = 0.5 * mtp_lds();
    SA_PROF_END(7);
    return out;
  }

  // 0.5 * mtp(grad_quadratic_form_inv(p)),  -(V (e e^T o J) V^T) = -A J A^T, A = V diag(e),
  // e = V^T p / lamt, J_kl = (lamt_k - lamt_l)/(lam_k - lam_l), J_kk = grad_softabs(lam_k)  (:1676-1685)
  __device__ __forceinline__ double dh2_dpos(double p, double q) {
    SA_PROF_BEGIN();
    SA_LAP_BEGIN();
    if (tid < NP) w.qv[tid] = (tid < dim) ? q : 0.0;  // for mtp_lds(): visible long before it runs
    {  // e = V^T p / lamt into w.v1 (vt_times() with the division in its output stage: one barrier less)
      double* const vin = w.ring + 1024;
      if (tid < NP) vin[tid] = (tid < dim) ? p : 0.0;
      __syncthreads();
      const int k = tid / RP, part = tid % RP;
      double s = 0.0;
      if (k < dim)
        for (int i = part; i < dim; i += RP) s = __builtin_fma(w.V[i * LD + k], vin[i], s);
      s = rp_sum(s);
      if (part == 0) w.v1[k] = (k < dim) ? fdiv(s, w.lamt[k]) : 0.0;
      __syncthreads();
    }
    SA_LAP(16);
    if constexpr (kMatricesInLds) {
      // B = A J on the matrix cores, A = V diag(e) formed in the operand (one multiply per term: A is never stored),
      // J in w.H, zero beyond dim: wave t owns the 16 x 16 tile (t / 4, t % 4) of B - sixteen v_mfma_f64_16x16x4, lane
      // group g taking the terms k = 16 g + kk (refine_eigh(): no LDS bank conflicts that way); accumulator lane
      // 16 g + j, register r = B[16 I + 4 r + g][16 Jt + j].  md_i = sum_l B_il A_il is reduced over a tile's columns on
      // the DPP row and over the four column tiles through LDS; m0_i = sum_l B_0l A_il needs row 0 of B only.
      // J depends on the eigenvalues only: it is built once per decomposition - the momentum fixed point calls this
      // several times at one position - and stays in w.H until build_hessian() overwrites it.
      if (!j_valid) {
        for (int el = tid; el < NP * NP; el += NT) {
          const int k = el / NP, l = el % NP;
          double jv = 0.0;
          if (k < dim && l < dim) {
            double num = w.lamt[k] - w.lamt[l], den = w.lam[k] - w.lam[l];
            if (k == l) { num += w.gsa[k]; den = 1.0; }
            jv = fdiv(num, den);                   // 0/0 -> NaN for degenerate spectra, as the reference
          }
          w.H[k * LD + l] = jv;
        }
        j_valid = true;
        __syncthreads();
      }
      SA_LAP(17);
      double* const part = w.ring;        // [4][NP] column-tile partials of md (the ring is idle outside eigh())
      double* const brow0 = w.ring + 4 * NP;  // [NP] row 0 of B
      {
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int g = lane >> 4, j = lane & 15;
        const int I = wave >> 2, Jt = wave & 3;
        d4 acc = {0.0, 0.0, 0.0, 0.0};
        ++n_products;
        const double* vrow = w.V + (16 * I + j) * LD + 16 * g;  // (w.V is zero beyond dim: init_backend, refine_eigh)
        const double* ek = w.v1 + 16 * g;
        const double* bcol = w.H + 16 * g * LD + 16 * Jt + j;
#pragma unroll
        for (int kk = 0; kk < NP / 4; ++kk)
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(vrow[kk] * ek[kk], bcol[kk * LD], acc, 0, 0, 0);
        if constexpr (USERH) {
          // grad_quadratic_form_inv = -(B A^T) in full: B into w.W, then tile (I, Jt) of B A^T with A = V diag(e) formed in
          // the operand again, written over B once every wave has read it
#pragma unroll
          for (int r = 0; r < 4; ++r) w.W[(16 * I + 4 * r + g) * LD + 16 * Jt + j] = acc[r];
          __syncthreads();
          d4 m = {0.0, 0.0, 0.0, 0.0};
          ++n_products;
          const double* brow_i = w.W + (16 * I + j) * LD + 16 * g;
          const double* vrow_j = w.V + (16 * Jt + j) * LD + 16 * g;
#pragma unroll
          for (int kk = 0; kk < NP / 4; ++kk)
            m = __builtin_amdgcn_mfma_f64_16x16x4f64(brow_i[kk], vrow_j[kk] * ek[kk], m, 0, 0, 0);
          __syncthreads();
#pragma unroll
          for (int r = 0; r < 4; ++r) w.W[(16 * I + 4 * r + g) * LD + 16 * Jt + j] = -m[r];
        } else {
        const double el = w.v1[16 * Jt + j];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = 16 * I + 4 * r + g;
          const double s = rp_sum_n<16>(acc[r] * (w.V[i * LD + 16 * Jt + j] * el));
          if (j == 0) part[Jt * NP + i] = s;
        }
        if (I == 0 && g == 0) brow0[16 * Jt + j] = acc[0];
        }
      }
      if constexpr (USERH) {
        const double out = 0.5 * user_mtp_w();
        SA_PROF_END(6);
        return out;
      }
      __syncthreads();
      SA_LAP(18);
      {
        const int i = tid / RP, pt = tid % RP;
        double m0 = 0.0;
#pragma unroll
        for (int m = 0; m < NP / RP; ++m) {
          const int l = pt + RP * m;
          m0 = __builtin_fma(brow0[l], w.V[i * LD + l] * w.v1[l], m0);
        }
        m0 = rp_sum(m0);
        if (pt == 0) {
          w.v2[i] = -((part[i] + part[NP + i]) + (part[2 * NP + i] + part[3 * NP + i]));
          w.nat[i] = -m0;
        }
      }
      __syncthreads();
    } else {
      // the same product with the matrices in the workspace (round 5; until then a VALU triple loop whose lanes walked
      // along rows of A sixteen cache lines at a time: 6 ms a call at D = 256, half of a step).  J is built once per
      // decomposition into w.H; tile (I, Jt) of B = A J on the matrix cores, A_ik = V_ik e_k read down the columns of V^T;
      // a wave owns one row of tiles (NP = 128: half of one), so md_i = sum_l B_il A_il accumulates in its registers.
      if (!j_valid) {
        for (int el = tid; el < NP * NP; el += NT) {
          const int k = el / NP, l = el % NP;
          double jv = 0.0;
          if (k < dim && l < dim) {
            double num = w.lamt[k] - w.lamt[l], den = w.lam[k] - w.lam[l];
            if (k == l) { num += w.gsa[k]; den = 1.0; }
            jv = num / den;                          // 0/0 -> NaN for degenerate spectra, as the reference
          }
          w.H[k * LD + l] = jv;
        }
        j_valid = true;
        __syncthreads();
      }
      constexpr int T = NP / 16, KQ = NP / 4, QR = T / 4, NW = NT / 64, WPR = NW / T;  // WPR waves share a row of tiles
      static_assert(NW % T == 0 && WPR >= 1 && WPR <= 4 && QR % WPR == 0, "tile rows of the dh2_dpos product");
      if constexpr (USERH) {
        // grad_quadratic_form_inv = -(B A^T) in full, B = A J: B^T into w.W (stored transposed, so that the second
        // product reads it down its columns), then tile (I, J) of B A^T = sum_k B^T[k][i] e_k V^T[k][j] into w.S
        constexpr int NQ = T * QR;
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int g = lane >> 4, j = lane & 15;
        const double* const ek = w.v1 + KQ * g;
        n_products += 2;
#pragma unroll 1
        for (int qd = wave; qd < NQ; qd += NW) {
          const int I = qd / QR, J0 = 4 * (qd % QR);
          d4 acc[4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
          quad_tiles<true>(w.Vt + KQ * g * LD + 16 * I + j, w.H + KQ * g * LD + 16 * J0 + j, ek, acc);
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) w.W[(16 * (J0 + u) + j) * LD + 16 * I + 4 * r + g] = acc[u][r];
        }
        __syncthreads();
#pragma unroll 1
        for (int qd = wave; qd < NQ; qd += NW) {
          const int I = qd / QR, J0 = 4 * (qd % QR);
          d4 acc[4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
          quad_tiles<true>(w.W + KQ * g * LD + 16 * I + j, w.Vt + KQ * g * LD + 16 * J0 + j, ek, acc);
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) w.S[(16 * I + 4 * r + g) * LD + 16 * (J0 + u) + j] = -acc[u][r];
        }
        const double out = 0.5 * user_mtp_of(w.S);
        SA_PROF_END(6);
        return out;
      }
      double* const part = w.ring;           // [WPR][NP] partial md (the ring is idle outside eigh())
      double* const brow0 = w.ring + 2048;   // [NP] row 0 of B
      {
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        const int g = lane >> 4, j = lane & 15;
        const int I = wave % T, jt0 = wave / T;
        const double* const vt_i = w.Vt + KQ * g * LD + 16 * I + j;
        const double* const ek = w.v1 + KQ * g;
        double mdacc[4] = {0.0, 0.0, 0.0, 0.0};
        ++n_products;
#pragma unroll 1
        for (int qd = jt0; qd < QR; qd += WPR) {  // four tiles of the row at a time (quad_tiles)
          const int J0 = 4 * qd;
          d4 acc[4] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
          quad_tiles<true>(vt_i, w.H + KQ * g * LD + 16 * J0 + j, ek, acc);
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int l = 16 * (J0 + u) + j;
            const double el = w.v1[l];
#pragma unroll
            for (int r = 0; r < 4; ++r)
              mdacc[r] += rp_sum_n<16>(acc[u][r] * (w.V[(16 * I + 4 * r + g) * LD + l] * el));
            if (I == 0 && g == 0) brow0[l] = acc[u][0];
          }
        }
        if (j == 0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) part[jt0 * NP + 16 * I + 4 * r + g] = mdacc[r];
        }
      }
      __syncthreads();
      {
        const int i = tid / RP, pt = tid % RP;
        double m0 = 0.0;
        for (int l = pt; l < dim; l += RP) m0 = __builtin_fma(brow0[l], w.V[i * LD + l] * w.v1[l], m0);
        m0 = rp_sum(m0);
        if (pt == 0) {
          double md = part[i];
#pragma unroll
          for (int a = 1; a < WPR; ++a) md += part[a * NP + i];
          w.v2[i] = -md;
          w.nat[i] = -m0;
        }
      }
      __syncthreads();
    }
    SA_LAP(19);
    const double out = 0.5 * mtp_lds();
    SA_LAP(20);
    SA_PROF_END(6);
    return out;
  }

  __device__ __forceinline__ double grad(double q) {
    SA_PROF_BEGIN();
    if (tid < NP) w.nat[tid] = (tid < dim) ? q : 0.0;
    __syncthreads();
    const TargetAux aux = target_prepare<true>(target, w.nat, dim, tparams, (int)tid & 63);
    const double g = (tid < dim) ? target_grad_elem<true>(target, aux, w.nat, tid, dim, tparams) : 0.0;
    __syncthreads();
    SA_PROF_END2(12);
    return g;
  }
  __device__ __forceinline__ double nld_elem(double q) {
    if (tid < NP) w.nat[tid] = (tid < dim) ? q : 0.0;
    __syncthreads();
    const TargetAux aux = target_prepare<true>(target, w.nat, dim, tparams, (int)tid & 63);
    const double e = (tid < dim) ? target_nld_elem<true>(target, aux, w.nat, tid, dim, tparams) : 0.0;
    __syncthreads();
    return e;
  }
};

// vectors, ring, counters and the step's stash in LDS; the three matrices in LDS (NP = 64) or in `work` (NP = 128)
template <int NP, bool USERH>
__device__ __forceinline__ void init_backend(SoftAbsBackendT<NP, USERH>& bk, const ImplicitArgs& A, double* lds, double* work) {
  using B = SoftAbsBackendT<NP, USERH>;
  bk.dim = A.dim;
  bk.tid.v = threadIdx.x;
  bk.tid_raw = threadIdx.x;
  bk.target = A.target;
  bk.coeff = uniform_f64(A.z[0]);  // softabs coefficient (device copy of the model's rmetric_params)
  bk.hparams = A.z + 1;            // ... followed by a user Hessian's parameters
  bk.tparams = A.tparams;
  bk.refine_on = A.no_refine == 0;
  double* p = lds;
  if (B::kMatricesInLds) {
    bk.w.H = p; p += B::MATJ;
    bk.w.W = p; p += B::MATJ;
    bk.w.V = p; p += B::MAT;
  } else {
    bk.w.H = work;
    bk.w.W = work + B::MATJ;
    bk.w.V = work + 2 * B::MATJ;
    bk.w.S = bk.w.V + B::MAT;
    bk.w.R = bk.w.S + B::MAT;
    bk.w.X2 = bk.w.R + B::MAT;
    bk.w.Vt = bk.w.X2 + B::MAT;
    bk.w.X2t = bk.w.Vt + B::MAT;
    bk.w.snapg = bk.w.X2t + B::MAT;
    // the basis buffers are read whole by the matrix-core products: zero beyond dim, once (nothing writes there later)
    if (A.dim < NP) {
      for (int el = threadIdx.x; el < NP * NP; el += NT) {
        const int i = el / NP, j = el % NP;
        if (i >= A.dim || j >= A.dim) {
          bk.w.V[i * B::LD + j] = 0.0;
          bk.w.X2[i * B::LD + j] = 0.0;
          bk.w.Vt[i * B::LD + j] = 0.0;
          bk.w.X2t[i * B::LD + j] = 0.0;
        }
      }
    }
  }
  bk.w.lam = p; p += NP;
  bk.w.lamt = p; p += NP;
  bk.w.gsa = p; p += NP;
  bk.w.v1 = p; p += NP;
  bk.w.v2 = p; p += NP;
  bk.w.nat = p; p += NP;
  bk.w.tp = p; p += NP;
  bk.w.qv = p; p += NP;
  bk.w.red = p; p += 32;
  bk.w.cnt = p; p += 8;
  bk.w.prof = p; p += 24;
  bk.w.stash = p; p += SL_COUNT * (NP + 1);
  bk.w.ring = p;  // (last of the vectors: the snapshots start in its tail and run on behind it, see kRingScratch)
  bk.w.snap = reinterpret_cast<float*>(p + B::kRingScratch);
  bk.w.panel = lds + B::kLdsDoubles - B::kPanelDoubles;  // (NP > 64: the last kPanelDoubles of the allocation)
  if (B::kMatricesInLds && A.dim < NP) {  // w.V is read whole by the matrix-core products: zero beyond dim, once
    for (int el = threadIdx.x; el < NP * NP; el += NT) {
      const int i = el / NP, j = el % NP;
      if (i >= A.dim || j >= A.dim) bk.w.V[i * B::LD + j] = 0.0;
    }
  }
  {  // (visible after the first barrier of whatever runs next)
    const int n_tp = A.target == MM_TARGET_FUNNEL ? A.dim - 1 : (A.target == MM_TARGET_POLY ? 2 : 0);
    if ((int)threadIdx.x < NP) bk.w.tp[threadIdx.x] = (int)threadIdx.x < n_tp ? A.tparams[threadIdx.x] : 0.0;
  }
  if (threadIdx.x < 8) bk.w.cnt[threadIdx.x] = 0.0;
  if (threadIdx.x < 24) bk.w.prof[threadIdx.x] = 0.0;
}

struct SaArgs {
  ImplicitArgs a;
  const double* coeff;  // device pointer to softabs_coeff
  double* work;         // NP = 128: [n_chains][kWorkDoubles] matrices of the chains
  double* eig;          // [n_chains][kEigDoubles] bases carried between launches, or nullptr
  int op;
};

// MIDPOINT: ImplicitMidpointIntegrator (integrators.py:547-681) on the same backend
template <bool MIDPOINT, int NP, bool USERH = false>
__device__ __forceinline__ void softabs_leapfrog_body(const SaArgs& S, double* lds) {
  ImplicitArgs A = S.a;
  A.z = S.coeff;
  const int64_t chain = blockIdx.x;
  SoftAbsBackendT<NP, USERH> bk;
  init_backend(bk, A, lds, S.work + chain * SoftAbsBackendT<NP, USERH>::kWorkDoubles);
  const int dim = A.dim, tid = threadIdx.x;
  const bool act = tid < dim;
  const double q = act ? A.pos[chain * dim + tid] : 0.0;
  const double p = act ? A.mom[chain * dim + tid] : 0.0;
  const double t = uniform_f64(signed_step(A.dir, A.step_scale, chain, A.step_size));
  bk.slot(SL_Q) = q;
  bk.slot(SL_P) = p;
  bk.eig_mem = (S.eig && bk.refine_on) ? S.eig + chain * SoftAbsBackendT<NP, USERH>::kEigDoubles : nullptr;
  bk.load_basis();
  __syncthreads();
  const int my_steps = mmdev::chain_steps(A.chain_steps, chain, A.n_steps);
#ifdef MM_SOFTABS_PROF
  const long long prof_start = __builtin_readcyclecounter();
#endif
  const ChainResult r = MIDPOINT ? implicit_midpoint_chain(bk, t, my_steps, A.opts)
                                 : implicit_leapfrog_chain(bk, t, my_steps, A.opts);
  if (act) {
    A.pos[chain * dim + tid] = bk.slot(SL_Q);
    A.mom[chain * dim + tid] = bk.slot(SL_P);
  }
  bk.store_basis();
#ifdef MM_SOFTABS_PROF
  if (tid == 0 && chain == 0)
    printf("softabs prof: total %lld eigh(incl basis) %.0f basis %.0f dh2_dpos %.0f half_vjp %.0f | n_eigh %d sweeps %d "
           "refined %d evals %.0f\n", (long long)(__builtin_readcyclecounter() - prof_start), bk.w.cnt[4], bk.w.cnt[5], bk.w.cnt[6],
           bk.w.cnt[7], bk.n_eigh, bk.n_sweeps, bk.n_refined, bk.w.cnt[CNT_EVALS]);
  if (tid == 0 && chain == 0)
    printf("softabs prof rounds (cross rounds of G wave 0): dots %.0f reduce %.0f params %.0f whole round %.0f "
           "store+sync %.0f\n", bk.w.prof[0], bk.w.prof[1], bk.w.prof[2], bk.w.prof[3], bk.w.prof[4]);
  if (tid == 0 && chain == 0)
    printf("softabs prof refine laps: G=AX %.0f S,XX %.0f E %.0f X+XE %.0f between %.0f\n", bk.w.prof[0], bk.w.prof[1],
           bk.w.prof[2], bk.w.prof[3], bk.w.prof[4]);
  if (tid == 0 && chain == 0)
    printf("softabs prof dh2_dpos laps: V^T p, e %.0f J, A %.0f B = A J %.0f md, m0 %.0f mtp %.0f\n", bk.w.prof[16],
           bk.w.prof[17], bk.w.prof[18], bk.w.prof[19], bk.w.prof[20]);
  if (tid == 0 && chain == 0)
    printf("softabs prof phases: build_hessian %.0f regularise %.0f matvec %.0f refine_eigh %.0f grad %.0f norm %.0f\n",
           bk.w.prof[8], bk.w.prof[9], bk.w.prof[10], bk.w.prof[11], bk.w.prof[12], bk.w.prof[13]);
#endif
  if (tid == 0) {
    A.status[chain] = r.status;
    A.n_done[chain] = r.done;
    add_counters(A.counters, r);
    if (A.counters) {
      atomicAdd((unsigned long long*)&A.counters->n_newton_iters, (unsigned long long)bk.n_sweeps);
      atomicAdd((unsigned long long*)&A.counters->n_eigh, (unsigned long long)bk.n_eigh);
      atomicAdd((unsigned long long*)&A.counters->n_refine, (unsigned long long)bk.n_refined);
      atomicAdd((unsigned long long*)&A.counters->n_mfma_products, (unsigned long long)bk.n_products);
    }
  }
}

// op 0: h = l + 0.5 logdet + 0.5 p^T M^-1 p ; 1: dh_dmom ; 2: sample_momentum = V diag(sqrt(lamt)) V^T z
template <int NP, bool USERH = false>
__device__ __forceinline__ void softabs_aux_body(const SaArgs& S, double* out, const double* z, double* lds) {
  ImplicitArgs A = S.a;
  A.z = S.coeff;
  const int64_t chain = blockIdx.x;
  SoftAbsBackendT<NP, USERH> bk;
  init_backend(bk, A, lds, S.work + chain * SoftAbsBackendT<NP, USERH>::kWorkDoubles);
  const int dim = A.dim, tid = threadIdx.x;
  const bool act = tid < dim;
  const double q = act ? A.pos[chain * dim + tid] : 0.0;
  const double p = act ? A.mom[chain * dim + tid] : 0.0;
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
  bk.eig_mem = (S.eig && bk.refine_on) ? S.eig + chain * SoftAbsBackendT<NP, USERH>::kEigDoubles : nullptr;
  bk.load_basis();
  const bool ok = bk.build_and_invert(q);
  bk.store_basis();
  if (S.op == 0) {
    const double u = bk.matvec(p);
    double e = bk.nld_elem(q) + (act ? 0.5 * p * u + 0.5 * log(fabs(bk.w.lamt[tid])) : 0.0);
    e = block_reduce4(e, 0, bk.w.red, bk.red_flip);
    if (tid == 0) out[chain] = ok ? e : nan;
  } else if (S.op == 1) {
    const double u = bk.matvec(p);
    if (act) out[chain * dim + tid] = ok ? u : nan;
  } else {
    const double zz = act ? z[chain * dim + tid] : 0.0;
    const double c = bk.vt_times(zz);
    const double y = bk.v_times(act ? sqrt(bk.w.lamt[tid]) * c : 0.0);
    if (act) A.mom[chain * dim + tid] = ok ? y : nan;
  }
}

#ifndef MM_RTC_BUILD  // the in-tree instantiations (a run-time translation unit defines extern "C" wrappers instead)
template <bool MIDPOINT, int NP>
__global__ __launch_bounds__(NT) void softabs_leapfrog_kernel(SaArgs S) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  softabs_leapfrog_body<MIDPOINT, NP>(S, lds);
}
template <int NP>
__global__ __launch_bounds__(NT) void softabs_aux_kernel(SaArgs S, double* out, const double* z) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  softabs_aux_body<NP>(S, out, z, lds);
}
#endif

}  // namespace mmsoftabs
