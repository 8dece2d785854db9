// Implicit leapfrog on dense-metric Riemannian systems, 75 < D <= 256 (BASELINE config c4), LOOK-AHEAD variant of
// k_implicit_blk16.hip: the same block-16 symmetric sweep / blocked LDL^T on the FP64 matrix cores, with the
// 16 x 16 pivot-block inverse taken off the critical path.  gfx950 / CDNA4.
//
// What k_implicit_blk16.hip measures (profiles/r02_c4_ubench_blk16.txt): of a block's 12-16 k cycles only 4-9 k are
// matrix-core work; 3.7 k are the pivot-block inverse (a latency chain on ONE wave, everybody else waiting at a
// barrier) and 3 k are barriers and panel publishing around it.  The inverse of block K+1 could run while the other
// waves still apply block K - but FP64 vector instructions and FP64 MFMAs share the units of a SIMD, so the
// inverting wave's chain crawls if ANY wave of its SIMD issues MFMAs (measured: a look-ahead sweep with the present
// ownership map is slower, 287 k against 192 k cycles), and runs at full speed if none does (op 14 of the debug hook:
// 3.46 k cycles during the other SIMDs' updates against 3.7 k alone).  Hence the ownership map of this kernel:
//   * waves 1,2,3,5,6,7 ("tile waves", t = 0..5, two per SIMD on three SIMDs) own the tile rows 15-t and 4+t:
//     21 tiles each, the slot algebra of k_implicit_blk16.hip with the second row shifted by four;
//   * wave 0 ("pivot wave") inverts every pivot block and owns the three tiles of the tile rows 1 and 0; wave 4, which
//     shares its SIMD, owns the seven tiles of the rows 3 and 2 - that SIMD issues next to no MFMAs.
//   (every wave runs the same code on a wave-uniform slot map: role branches around code that touches the tile
//   registers cost 1.4 KB of scratch per lane; the first version of this kernel had them)
// and the order of a block K:
//   T_K (ready) -> -W blocks -> the tiles of tile column K+1 WITH block K's update, into temporaries -> published as
//   panel K+1 -> barrier -> all tile updates, forward substitution; wave 0 then: T_(K+1) = -P_(K+1)^-1 -> barrier.
// The panel and T are double-buffered by block parity.  Everything else (layouts, substitution, mat-vec, the step
// state machine of implicit_core.h) is k_implicit_blk16.hip's; see there for the layouts.
//
// STATUS (round 2): correct (same tests as k_implicit_blk16.hip, tests/test_gpu_blk16.py runs both) but SLOWER -
// 2.04e5 against 2.65e5 steps/s at c4 - and therefore not the default (MICI_AMD_IMPLICIT_KERNEL=blk16la selects it).
// The pivot-block inverse is hidden as intended (3.4 k cycles on wave 0 under the other waves' updates,
// profiles/r02_c4_ubench_blk16la.txt), but the tile updates of a trailing block take 7.5-9 k cycles for ~9.4 tiles a
// wave - 45 % of the matrix core's rate - and with the tiles on three SIMDs instead of four that outweighs the gain.
// DESIGN.md section 8 has the numbers and what was tried on the update pass.
//
// Reference arithmetic replaced: DensePositiveDefiniteMatrix factorisation, explicit inverse and solves
// (matrices.py:1161-1188, 932-938) inside ImplicitLeapfrogIntegrator._step (integrators.py:493-544).
#include <utility>

#include "implicit_core.h"

namespace {

using namespace mmdev;
using namespace mmimp;

typedef double d4 __attribute__((ext_vector_type(4)));

// every conditional arm that touches a tile register ENDS with a distinct marker: identical tails get merged by the
// optimiser into one block that indexes the tile array dynamically - which moves it from registers to scratch
#define MM_ARM_MARK(N, S) asm volatile("; tile arm " #N " slot %c0" ::"i"(S))

// a loop over compile-time indices 0..N-1 (the slot number of a tile register must be a constant expression for the
// markers above, and no pass may turn the slot loops back into run-time indexed ones)
template <class F, int... S>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, S...>) {
  (f(std::integral_constant<int, S>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

constexpr int NT16 = 16;            // tile rows
constexpr int DPM = 16 * NT16;      // padded dimension
constexpr int NWAVE = 8;            // two per SIMD
constexpr int NTHR = 64 * NWAVE;
constexpr int NSLOT = 21;           // tiles of a tile wave
constexpr int NCLASS = 2;           // tile rows of a tile wave
constexpr int CS = 18;              // doubles per panel column in LDS: 16 + 2 (keeps 16-byte alignment, spreads banks)
constexpr int PSTR = 17;            // partial sums per output element: 16 column-sum slots + the row sum
constexpr int VLM = DPM + 8;        // flat vectors: DPM elements + a dummy cell for threads >= DPM

constexpr int kOffStash = 0;                               // [SL_COUNT][VLM] flat per-thread state of the step
constexpr int kOffNat = kOffStash + SL_COUNT * VLM;        // [VLM] natural-order vector
constexpr int kOffVperm = kOffNat + VLM;                   // [DPM] the same vector as [I][g][r]
constexpr int kOffAux = kOffVperm + DPM;                   // [VLM] second natural-order vector (z of the substitution)
constexpr int kOffRed = kOffAux + VLM;                     // [24]  team reductions / flags / work counters
constexpr int kOffScr = kOffRed + 24;                      // [64] scratch of the in-tile sweep, [2][256] T = -P^-1 in lane
                                                           // order by block parity, [2] positive-definite flags
constexpr int kScrDoubles = 64 + 2 * 256 + 8;
constexpr int kOffX = kOffScr + kScrDoubles;               // [2][DPM][CS]  panel, double-buffered by block parity
constexpr int kOffPart = kOffX + 2 * DPM * CS;             // [DPM][PSTR]
constexpr int kOffB = kOffPart + DPM * PSTR;               // [DPM] right-hand side of a solve-only construction
constexpr int kLdsDoubles = kOffB + DPM;
static_assert((kOffVperm % 2) == 0 && (kOffScr % 2) == 0 && (kOffX % 2) == 0, "16-byte alignment of the d4 accesses");
static_assert(kLdsDoubles * 8 <= 160 * 1024, "LDS budget of a CU");

__device__ __forceinline__ int fresh_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

__device__ __forceinline__ int opaque_scalar(int v) {
  v = __builtin_amdgcn_readfirstlane(v);
  asm volatile("" : "+s"(v));
  return v;
}

__device__ __forceinline__ double uniform_f64(double v) {
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readfirstlane((int)(b & 0xffffffffLL));
  const int hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ bool uniform_flag(bool v) { return __builtin_amdgcn_readfirstlane(v ? 1 : 0) != 0; }

__device__ __forceinline__ double team_reduce(double v, int kind_max, double* red) {
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

// sum over the four DPP rows (lanes l, l ^ 16, l ^ 32, l ^ 48) with gfx950's VALU row swaps (semantics verified by
// tests/test_gpu_blk16.py::test_permlane_swap_semantics)
__device__ __forceinline__ double swap_sum16(double m) {
  const long long b = __double_as_longlong(m);
  const unsigned lo = (unsigned)(b & 0xffffffffLL), hi = (unsigned)(b >> 32);
  const auto l2 = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  const auto h2 = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  return __longlong_as_double(((long long)h2[0] << 32) | (unsigned)l2[0]) +
         __longlong_as_double(((long long)h2[1] << 32) | (unsigned)l2[1]);
}
__device__ __forceinline__ double swap_sum32(double m) {
  const long long b = __double_as_longlong(m);
  const unsigned lo = (unsigned)(b & 0xffffffffLL), hi = (unsigned)(b >> 32);
  const auto l2 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto h2 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __longlong_as_double(((long long)h2[0] << 32) | (unsigned)l2[0]) +
         __longlong_as_double(((long long)h2[1] << 32) | (unsigned)l2[1]);
}
__device__ __forceinline__ double sum_over_g(double m) { return swap_sum32(swap_sum16(m)); }

// rs[r] = this lane's partial of row element 4 r + g: sum over the 16 lanes of a DPP row with one transposing
// butterfly (4 values -> 1).  All four lanes of a quad end up with the sum for register r = j >> 2.
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

template <int RMETRIC>
struct TeamLa {
  static constexpr bool kSolveByInverse = false;
  static constexpr bool kUnifiedConstruct = true;
  static constexpr bool kCountersInLds = true;
  d4 acc[NSLOT];
  int wave;       // hardware wave index 0..7
  int nblk;       // number of 16-pivot blocks that contain real rows: ceil(dim / 16)
  int dim, target;
  struct OpaqueTid {
    int v;
    __device__ __forceinline__ operator int() const {
      int x = v;
      asm volatile("" : "+v"(x));
      return x;
    }
  } tid;
  double* lds;
  const double* base;
  int base_ld;
  const double* tparams;

  // ---- the slot map (wave-uniform; ALL waves run the same code).  A wave owns two tile rows: r0 in the slots
  // 0..r0 (from its diagonal leftwards: tile (r0, r0 - s)) and r1 in the slots c1..20, c1 = 20 - r1 (ENDING on its
  // diagonal: tile (r1, s - c1)); the slots in between are empty.  Tile waves (1,2,3,5,6,7 -> t = 0..5): rows 15-t and
  // 4+t, all 21 slots in use.  Wave 0, which inverts the pivot blocks, owns rows 1 and 0 (3 tiles); wave 4, on the same
  // SIMD, rows 3 and 2 (7 tiles) - so that SIMD issues next to no MFMAs (none at all after block 3 of a trailing sweep).
  // Slots 0, 1 and 20 have a compile-time class and are in use on every wave.
  struct Map {
    int r0, r1, c1;
  };
  __device__ __forceinline__ Map map() const {
    const int w = opaque_scalar(wave);
    const int t = w < 4 ? w - 1 : w - 2;
    Map m;
    m.r0 = w == 0 ? 1 : w == 4 ? 3 : 15 - t;
    m.r1 = w == 0 ? 0 : w == 4 ? 2 : 4 + t;
    m.c1 = NSLOT - 1 - m.r1;
    return m;
  }
  __device__ static __forceinline__ bool in_use(const int s, const Map& m) { return s <= m.r0 || s >= m.c1; }
  __device__ static __forceinline__ int row_class(const int s, const Map& m) { return s <= m.r0 ? 0 : 1; }
  __device__ static __forceinline__ int row_of_class(const int c, const Map& m) { return c == 0 ? m.r0 : m.r1; }
  // tile coordinates; an empty slot gets coordinates no condition on a block index ever matches
  __device__ static __forceinline__ int tile_i(const int s, const Map& m) {
    return s <= m.r0 ? m.r0 : s >= m.c1 ? m.r1 : -64;
  }
  __device__ static __forceinline__ int tile_j(const int s, const Map& m) {
    return s <= m.r0 ? m.r0 - s : s >= m.c1 ? s - m.c1 : -64;
  }
  __device__ static constexpr bool is_diag_slot(const int s) { return s == 0 || s == NSLOT - 1; }
  __device__ static constexpr bool class_known(const int s) { return s <= 1 || s == NSLOT - 1; }
  // per-row quantity of slot s out of the row classes' values (a select, not a branch)
  __device__ static __forceinline__ d4 pick_row(const int s, const Map& m, const d4 (&v)[NCLASS]) {
    if (s <= 1) return v[0];
    if (s == NSLOT - 1) return v[1];
    const bool lo = s <= m.r0;
    d4 r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = lo ? v[0][k] : v[1][k];
    return r;
  }
  // accumulate v * a into the in-lane sums of slot s's row class (masked operands for the run-time slots)
  __device__ static __forceinline__ void add_row(const int s, const Map& m, d4 (&rs)[NCLASS], const d4 a, const double v) {
    if (class_known(s)) {
      const int c = s <= 1 ? 0 : 1;
#pragma unroll
      for (int r = 0; r < 4; ++r) rs[c][r] = __builtin_fma(a[r], v, rs[c][r]);
    } else {
      const bool lo = s <= m.r0;
      const double va = lo ? v : 0.0, vb = lo ? 0.0 : v;
#pragma unroll
      for (int r = 0; r < 4; ++r) rs[0][r] = __builtin_fma(a[r], va, rs[0][r]);
#pragma unroll
      for (int r = 0; r < 4; ++r) rs[1][r] = __builtin_fma(a[r], vb, rs[1][r]);
    }
  }

  __device__ __forceinline__ void count(const int which, const int n) {
    if (tid == 0) lds[kOffRed + 16 + which] += (double)n;
  }
  __device__ __forceinline__ void read_counts(ChainResult& r) const {
    r.n_evals = (long long)lds[kOffRed + 16 + CNT_EVALS];
    r.n_solves = (long long)lds[kOffRed + 16 + CNT_SOLVES];
    r.n_metric = (long long)lds[kOffRed + 16 + CNT_METRIC];
    r.n_grad = (long long)lds[kOffRed + 16 + CNT_GRAD];
  }
  __device__ __forceinline__ double& slot(int i) { return lds[kOffStash + i * VLM + (tid < DPM ? tid : DPM)]; }

  __device__ __forceinline__ double norm(double x, int kind) {
    const double a = tid < dim ? x : 0.0;
    if (kind == MM_NORM_LINF) return uniform_f64(team_reduce(fabs(a), 1, lds + kOffRed));
    return uniform_f64(sqrt(team_reduce(a * a, 0, lds + kOffRed)));
  }

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
    const Map m = map();
    const int ln = fresh_lane(), g = ln >> 4, j = ln & 15;
    const double inv_d = 1.0 / (double)dim;
    double chk = 0.0;
    // the row operands of the wave's two tile rows, once (a slot's row is one of the two)
    d4 qrow[NCLASS];
#pragma unroll
    for (int c = 0; c < NCLASS; ++c)
      qrow[c] = *reinterpret_cast<const d4*>(lds + kOffVperm + ((row_of_class(c, m) * 4 + g) << 2));
    static_for<NSLOT>([&](auto sc_) {
      constexpr int s = decltype(sc_)::value;
      {
        // every slot is (re)defined here, the empty ones with entries of the wave's first tile row: a slot defined only
        // under a condition would keep its previous value alive across the whole build
        const bool used = class_known(s) || in_use(s, m);
        const int I = used ? tile_i(s, m) : m.r0, J = used ? tile_j(s, m) : 0;
        const d4 qr = pick_row(s, m, qrow);
        if constexpr (RMETRIC == MM_RMETRIC_RANK1) {
          // base matrix (L2-resident) + q q^T / D: wave-uniform tile origin (scalar base) + a lane offset shared by all
          const double* tile0 = base + (unsigned)((16 * I) * base_ld + 16 * J);
          const unsigned lane_off = (unsigned)(g * base_ld + j);
          const double qs = lds[kOffNat + 16 * J + j] * inv_d;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            acc[s][r] = __builtin_fma(qr[r], qs, tile0[lane_off + (unsigned)(4 * r * base_ld)]);
        } else {
          acc[s] = d4{0.0, 0.0, 0.0, 0.0};
        }
        if constexpr (is_diag_slot(s)) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const bool on_diag = (j == 4 * r + g);
            if constexpr (RMETRIC == MM_RMETRIC_DIAGQUAD) {
              if (on_diag) acc[s][r] = __builtin_fma(qr[r], qr[r], 1.0);
            }
            if (on_diag && 16 * I + 4 * r + g >= dim) acc[s][r] = 1.0;  // identity on the padding
            // both built-in metrics have their largest entries on the diagonal: a non-finite entry anywhere implies a
            // non-finite diagonal entry (matrices.py:211-215, "Array is not finite.")
            chk = __builtin_fma(acc[s][r], 0.0, chk);
          }
        }
        MM_ARM_MARK(0, s);
      }
    });
    return __builtin_amdgcn_ballot_w64(chk != 0.0) != 0;
  }

  // B operands of column tile J: bx[kk] = X[4 kk + g][16 J + j] (the panel is published as Q - E already)
  __device__ static __forceinline__ d4 load_b(const double* X, const int J, const int g, const int j) {
    return *reinterpret_cast<const d4*>(X + (16 * J + j) * CS + 4 * g);
  }

  // ---- the 16 x 16 pivot block in accumulator layout -> T = -P^-1 (same layout), by a 4-wide symmetric sweep whose
  // rank-4 updates are single MFMAs (k_implicit_blk16.hip)
  __device__ __forceinline__ void tile_sweep(d4& t, bool& ok, const int g, const int j) {
    double* scr = lds + kOffScr;
#pragma unroll
    for (int R0 = 0; R0 < 4; ++R0) {
      scr[j * 4 + g] = t[R0];  // scr[c][s] = T[4 R0 + s][c]
      wave_sync();
      const d4 qv = *reinterpret_cast<const d4*>(scr + j * 4);
      const d4 c0 = *reinterpret_cast<const d4*>(scr + (4 * R0 + 0) * 4);
      const d4 c1 = *reinterpret_cast<const d4*>(scr + (4 * R0 + 1) * 4);
      const d4 c2 = *reinterpret_cast<const d4*>(scr + (4 * R0 + 2) * 4);
      const d4 c3 = *reinterpret_cast<const d4*>(scr + (4 * R0 + 3) * 4);
      wave_sync();
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

  // acc += (-W rows)^T X for one tile: four dependent MFMAs (the other wave of the SIMD fills the matrix core)
  __device__ static __forceinline__ void update_tile(d4& a, const d4 nwr, const double* X, const int J, const int g,
                                                     const int j) {
    const d4 bx = load_b(X, J, g, j);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) a = __builtin_amdgcn_mfma_f64_16x16x4f64(nwr[kk], bx[kk], a, 0, 0, 0);
  }
  // panel K1 into Xn: a tile of the pivot row goes in as it is (the pivot block itself minus the identity), a tile
  // of tile column K1 below the pivot block transposed.  TRAILING panels hold the pivot block and the column only.
  template <bool TRAILING>
  __device__ static __forceinline__ void publish_tile(const d4 v, const int I, const int J, const int K1, double* Xn,
                                                      const int g, const int j) {
    if (I == K1) {
      if (J == K1) {
        d4 m = v;
#pragma unroll
        for (int r = 0; r < 4; ++r) m[r] -= (j == 4 * r + g) ? 1.0 : 0.0;
        *reinterpret_cast<d4*>(Xn + (16 * J + j) * CS + 4 * g) = m;
      } else if (!TRAILING) {
        *reinterpret_cast<d4*>(Xn + (16 * J + j) * CS + 4 * g) = v;
      }
    } else if (J == K1) {
      double* dst = Xn + (16 * I + g) * CS + (j & 3) * 4 + (j >> 2);  // + 4 r columns
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[4 * r * CS] = v[r];
    }
  }

  // Forward substitution, one step, run inside the trailing sweeps after block K's updates: y_K = b_K is final
  // once block K starts; the tiles (I, K) just became the finished factor tiles T_IK, so b_I -= T_IK y_K for the rows
  // below K, and z_K = P_K^-1 y_K from the pivot row's own tile (-P_K^-1).
  __device__ __forceinline__ void forward_tile(const d4 a, const int I, const int K, const int g, const int j) {
    const double yk = lds[kOffB + 16 * K + j];
    d4 c;
#pragma unroll
    for (int k = 0; k < 4; ++k) c[k] = a[k] * yk;
    const double s = row_reduce16(c, j);  // the four lanes of a quad hold the same value
    const int e = 16 * I + 4 * (j >> 2) + g;
    if (I == K) lds[kOffAux + e] = -s;
    else lds[kOffB + e] = lds[kOffB + e] - s;
  }

  // ---- block-16 symmetric sweep with look-ahead.  TRAILING = false: every tile is updated by every block, tiles end
  // as -M^-1.  TRAILING = true: only tiles (I, J) with J >= K - the blocked LDL^T: tile (K, K) = -P_K^-1, tile (I, K) =
  // A_IK P_K^-1.  `bad` = this wave's non-finite flag from build().  Returns "positive definite and finite" (uniform).
  template <bool TRAILING, bool PROF = false>
  __device__ __forceinline__ bool sweep(const bool bad) {
    bool ok = true;
    long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // PROF: cycles per phase, summed over the blocks
    const bool pivot_wave = opaque_scalar(wave) == 0;
    if (fresh_lane() == 0) lds[kOffRed + 8 + wave] = bad ? 1.0 : 0.0;  // read by everyone after the barriers below
    double* const tbuf = lds + kOffScr + 64;        // [2][64 lanes][4]
    double* const okf = lds + kOffScr + 64 + 512;   // [2]

    // T = -P^-1 of panel K1's pivot block, by the pivot wave, into the T buffer of K1's parity
    auto invert = [&](const int K1) {
      const int ln = fresh_lane(), g = ln >> 4, j = ln & 15;
      const double* Xn = lds + kOffX + (K1 & 1) * (DPM * CS);
      d4 t = load_b(Xn, K1, g, j);
#pragma unroll
      for (int r = 0; r < 4; ++r) t[r] += (j == 4 * r + g) ? 1.0 : 0.0;
      bool okb = true;
      tile_sweep(t, okb, g, j);
      *reinterpret_cast<d4*>(tbuf + (K1 & 1) * 256 + 4 * ln) = t;
      if (ln == 0) okf[K1 & 1] = okb ? 0.0 : 1.0;
    };

    {  // prologue: panel 0 straight from the tiles of tile column 0, then T_0
      const Map m = map();
      const int ln = fresh_lane(), g = ln >> 4, j = ln & 15;
      double* Xn = lds + kOffX;
      static_for<NSLOT>([&](auto sc_) {
        constexpr int s = decltype(sc_)::value;
        const int I = tile_i(s, m), J = tile_j(s, m);
        if (J == 0) {
          publish_tile<TRAILING>(acc[s], I, J, 0, Xn, g, j);
          MM_ARM_MARK(1, s);
        }
      });
    }
    __syncthreads();
    if (pivot_wave) invert(0);
    __syncthreads();

    int K = 0;
#pragma unroll 1
    do {
      const Map m = map();
      const int ln = fresh_lane(), g = ln >> 4, j = ln & 15;
      const double* X = lds + kOffX + (K & 1) * (DPM * CS);
      const int K1 = K + 1;
      const bool more = K1 < nblk;
      long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0;
      if constexpr (PROF) c0 = __builtin_readcyclecounter();
      const d4 t = *reinterpret_cast<const d4*>(tbuf + (K & 1) * 256 + 4 * ln);
      ok = ok && (okf[K & 1] == 0.0);
      // the two 16 x 16 blocks of -W = T X this wave's tile rows need, straight into A-operand registers:
      // lane (g, i), register kk <-> (-W)[4 kk + g][16 I + i]  (a finished row of a trailing sweep needs none)
      d4 nw[NCLASS];
#pragma unroll
      for (int c = 0; c < NCLASS; ++c) {
        nw[c] = d4{0.0, 0.0, 0.0, 0.0};
        if (!TRAILING || row_of_class(c, m) >= K) {
          const d4 bb = load_b(X, row_of_class(c, m), g, j);
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) nw[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(t[kk], bb[kk], nw[c], 0, 0, 0);
        }
      }
      if constexpr (PROF) {
        asm volatile("" : "+v"(nw[0]), "+v"(nw[1]));
        c1 = __builtin_readcyclecounter();
      }
      // look-ahead: this wave's share of panel K+1 = its tiles of tile row / column K+1 WITH block K's update, computed
      // into a temporary and published; the tiles themselves get the update in the pass below (a conditional write of
      // a tile register is a PHI the register allocator resolves with copies; conditional reads cost nothing)
      if (more) {
        double* Xn = lds + kOffX + (K1 & 1) * (DPM * CS);
        if constexpr (TRAILING) {
          // at most one tile per row class: the one in tile column K+1 (slot r0 - K1 / c1 + K1).  Gathered by
          // conditional READS, updated as two interleaved MFMA chains (a dependent chain alone runs at ~2/3 of the
          // matrix core's rate, and the other wave of the SIMD does not fill the gaps), published.
          const bool h0 = m.r0 >= K1, h1 = m.r1 >= K1;
          d4 tmp0 = d4{0.0, 0.0, 0.0, 0.0}, tmp1 = d4{0.0, 0.0, 0.0, 0.0};
          static_for<NSLOT>([&](auto sc_) {
            constexpr int s = decltype(sc_)::value;
            if constexpr (s <= 15) {
              if (s == m.r0 - K1) {
                tmp0 = acc[s];
                MM_ARM_MARK(2, s);
              }
            }
            if constexpr (s >= 5) {
              if (s == m.c1 + K1) {
                tmp1 = acc[s];
                MM_ARM_MARK(7, s);
              }
            }
          });
          if (h0 || h1) {
            const d4 bx = load_b(X, K1, g, j);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
              tmp0 = __builtin_amdgcn_mfma_f64_16x16x4f64(nw[0][kk], bx[kk], tmp0, 0, 0, 0);
              tmp1 = __builtin_amdgcn_mfma_f64_16x16x4f64(nw[1][kk], bx[kk], tmp1, 0, 0, 0);
            }
            if (h0) publish_tile<TRAILING>(tmp0, m.r0, K1, K1, Xn, g, j);
            if (h1) publish_tile<TRAILING>(tmp1, m.r1, K1, K1, Xn, g, j);
          }
        } else {
          static_for<NSLOT>([&](auto sc_) {
            constexpr int s = decltype(sc_)::value;
            const int I = tile_i(s, m), J = tile_j(s, m);
            if ((I == K1 && J >= 0) || (J == K1 && I > K1)) {
              d4 tmp = acc[s];
              update_tile(tmp, pick_row(s, m, nw), X, J, g, j);
              publish_tile<TRAILING>(tmp, I, J, K1, Xn, g, j);
              MM_ARM_MARK(2, s);
            }
          });
        }
      }
      if constexpr (PROF) c2 = __builtin_readcyclecounter();
      __syncthreads();
      if constexpr (PROF) c3 = __builtin_readcyclecounter();
      // rank-16 update of the active tiles: one wave-uniform conditional arm per tile, four dependent MFMAs each.
      // Measured alternatives
      // (tools/ubench_blk16la.py), none faster: pairs of tiles with interleaved chains behind three-way conditions;
      // the same with the MFMAs as in-place inline assembly (no copies at the joins); the next tile's B operands read
      // ahead of each arm; a switch on the number of active tiles into straight-line interleaved code (450 bytes of
      // scratch per lane).  The update pass runs at ~45 % of the matrix core's rate in every one of them.
      static_for<NSLOT>([&](auto sc_) {
        constexpr int s = decltype(sc_)::value;
        const int J = tile_j(s, m);
        if (TRAILING ? J >= K : (class_known(s) || in_use(s, m))) {
          update_tile(acc[s], pick_row(s, m, nw), X, J, g, j);
          MM_ARM_MARK(3, s);
        }
      });
      // A_KK -= 2 I on the pivot block's own tile
#pragma unroll
      for (int s = 0; s < NSLOT; ++s) {
        if (!is_diag_slot(s)) continue;
        if (tile_i(s, m) == K) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (j == 4 * r + g) acc[s][r] -= 2.0;
        }
      }
      if constexpr (TRAILING) {
        static_for<NSLOT>([&](auto sc_) {
          constexpr int s = decltype(sc_)::value;
          if (tile_j(s, m) == K) {
            forward_tile(acc[s], tile_i(s, m), K, g, j);
            MM_ARM_MARK(4, s);
          }
        });
      }
      if constexpr (PROF) {
        asm volatile("" : "+v"(acc[0]), "+v"(acc[NSLOT - 1]));
        c4 = __builtin_readcyclecounter();
      }
      // the next pivot block's inverse, while the tile waves are still updating (the pivot wave's own three tiles come
      // first: with the -W blocks dead the in-tile sweep's temporaries fit next to the 21 tile slots)
      if (pivot_wave && more) invert(K1);
      if constexpr (PROF) c5 = __builtin_readcyclecounter();
      __syncthreads();
      if constexpr (PROF) {
        c6 = __builtin_readcyclecounter();
        pc[0] += c1 - c0;  // -W blocks (includes waiting for T)
        pc[1] += c2 - c1;  // look-ahead tiles + publish
        pc[2] += c3 - c2;  // barrier 1
        pc[3] += c4 - c3;  // tile updates, forward substitution
        pc[4] += c5 - c4;  // pivot-block inverse (wave 0)
        pc[5] += c6 - c5;  // barrier 2
        pc[6] += 1;
      }
    } while (++K < nblk);
    if constexpr (PROF) {
      if (fresh_lane() == 0) {
        long long* dst = reinterpret_cast<long long*>(lds + kOffPart) + wave * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) dst[k] = pc[k];
      }
    }
    double flags = 0.0;
#pragma unroll
    for (int k = 0; k < NWAVE; ++k) flags += lds[kOffRed + 8 + k];
    return ok && flags == 0.0;
  }

  // ---- y = M^-1 v with the explicit inverse in the tiles (they hold -M^-1 after the full sweep) ----------------
  __device__ __forceinline__ double matvec(double v) {
    publish_vector(v);
    const Map m = map();
    const int ln = fresh_lane(), g = ln >> 4, j = ln & 15;
    double* part = lds + kOffPart;
    d4 rs[NCLASS];
#pragma unroll
    for (int c = 0; c < NCLASS; ++c) rs[c] = d4{0.0, 0.0, 0.0, 0.0};
    static_for<NSLOT>([&](auto sc_) {
      constexpr int s = decltype(sc_)::value;
      if (class_known(s) || in_use(s, m)) {
        const int I = tile_i(s, m), J = tile_j(s, m);
        const double vc = lds[kOffNat + 16 * J + j];
        const d4 a = acc[s];
        add_row(s, m, rs, a, vc);
        if constexpr (!is_diag_slot(s)) {  // below the diagonal: the mirrored tile's rows are this tile's columns
          const d4 vr = *reinterpret_cast<const d4*>(lds + kOffVperm + ((I * 4 + g) << 2));
          double mm = a[0] * vr[0];
          mm = __builtin_fma(a[1], vr[1], mm);
          mm = __builtin_fma(a[2], vr[2], mm);
          mm = __builtin_fma(a[3], vr[3], mm);
          mm = sum_over_g(mm);
          part[(16 * J + j) * PSTR + I] = mm;
        }
        MM_ARM_MARK(5, s);
      }
    });
#pragma unroll
    for (int c = 0; c < NCLASS; ++c) {
      const double k = row_reduce16(rs[c], j);
      part[(16 * row_of_class(c, m) + 4 * (j >> 2) + g) * PSTR + 16] = k;
    }
    __syncthreads();
    double y = 0.0;
    if (tid < DPM) {
      // column-sum slots I <= (the element's own tile row) are never written and stay zero from kernel start
      const double* src = lds + kOffPart + tid * PSTR;
#pragma unroll
      for (int k = 0; k < PSTR; ++k) y += src[k];
    }
    __syncthreads();
    return tid < dim ? -y : 0.0;
  }

  // ---- u = M^-1 b from the trailing-sweep (LDL^T) factors: the backward substitution over the tile rows, one workgroup
  // barrier per tile row (the forward and diagonal passes ran inside sweep<true>: aux holds z = D^-1 L^-1 b)
  __device__ __forceinline__ double solve() {
    const Map m = map();
    const int ln = fresh_lane(), g = ln >> 4, j = ln & 15;
    double* nat = lds + kOffNat;
    double* part = lds + kOffPart;
    __syncthreads();
    // backward: u_K = z_K - sum_{I > K} T_IK^T u_I.  Step K: the owner of tile row K sums the column partials that
    // rows I > K left in part[16 K + j][I], publishes u_K, and leaves its own row's partials for the columns J < K.
#pragma unroll 1
    for (int K = nblk - 1; K >= 0; --K) {
      if (K == m.r0 || K == m.r1) {
        double u = lds[kOffAux + 16 * K + j];
        const double* src = part + (16 * K + j) * PSTR;
        // slots I <= K are never written (zero from kernel start), slots I >= nblk hold the zeros of padding tiles
        double acc_u[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int I = 0; I < NT16; ++I) acc_u[I & 3] += src[I];
        u -= (acc_u[0] + acc_u[1]) + (acc_u[2] + acc_u[3]);
        // all four DPP rows computed the same u: natural order (the result) and the [I][g][r] copy
        nat[16 * K + j] = u;
        lds[kOffVperm + (((K << 2) + (j & 3)) << 2) + (j >> 2)] = u;
        wave_sync();
        const d4 ur = *reinterpret_cast<const d4*>(lds + kOffVperm + ((K * 4 + g) << 2));
        static_for<NSLOT - 1>([&](auto sc_) {  // the off-diagonal tiles of tile row K
          constexpr int s = decltype(sc_)::value;
          if constexpr (s >= 1) {
            if (tile_i(s, m) == K) {
              const d4 a = acc[s];
              double mm = a[0] * ur[0];
              mm = __builtin_fma(a[1], ur[1], mm);
              mm = __builtin_fma(a[2], ur[2], mm);
              mm = __builtin_fma(a[3], ur[3], mm);
              part[(16 * tile_j(s, m) + j) * PSTR + K] = sum_over_g(mm);
              MM_ARM_MARK(6, s);
            }
          }
        });
      }
      __syncthreads();
    }
    const double u = (tid < dim) ? nat[tid] : 0.0;
    __syncthreads();
    return u;
  }

  // implicit_core.h, kUnifiedConstruct: metric_func(x) then either the explicit inverse (kept in the tiles for
  // matvec / half_vjp_inv / dh2_dpos) or the single solve u = M(x)^-1 rhs
  __device__ __forceinline__ bool construct(double x, bool need_inverse, double rhs, double* u) {
    // build() is instantiated inside each arm on purpose (k_implicit_blk16.hip: one shared copy in front of the branch
    // makes the register allocator spill the whole tile set for one of the arms)
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
    const Map m = map();
    const int ln = fresh_lane(), g = ln >> 4, j = ln & 15;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      if (!is_diag_slot(s)) continue;
      const int I = tile_i(s, m);
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (j == 4 * r + g) lds[kOffNat + 16 * I + 4 * r + g] = -acc[s][r];
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
    int i = tid;
    asm volatile("" : "+v"(i));
    const TargetAux aux = target_prepare<false>(target, nat, dim, tparams, i & 63);
    const double gr = (i < dim) ? target_grad_elem<false>(target, aux, nat, i, dim, tparams) : 0.0;
    __syncthreads();
    return gr;
  }
};

template <int RMETRIC>
__device__ __forceinline__ void init_backend(TeamLa<RMETRIC>& bk, const ImplicitArgs& A, int base_ld, double* lds) {
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  __builtin_assume(wv >= 0 && wv < NWAVE);
  bk.wave = wv;
  bk.dim = A.dim;
  bk.nblk = (A.dim + 15) >> 4;
  bk.tid.v = threadIdx.x;
  bk.target = A.target;
  bk.lds = lds;
  bk.base = A.rparams;
  bk.base_ld = base_ld;
  bk.tparams = A.tparams;
  for (int i = threadIdx.x; i < DPM * PSTR; i += NTHR) lds[kOffPart + i] = 0.0;  // unused partial-sum slots stay 0
  if (threadIdx.x < 8) lds[kOffRed + 16 + threadIdx.x] = 0.0;                     // work counters
  __syncthreads();
}

template <int RMETRIC>
__global__ __launch_bounds__(NTHR, 2) void implicit_blk16la_kernel(ImplicitArgs A, int base_ld) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  TeamLa<RMETRIC> bk;
  init_backend(bk, A, base_ld, lds);
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
}

// Developer / test hook: the linear algebra of the backend on its own (tests/test_gpu_blk16.py).  Per chain, with
// x = pos and b = mom:  op 0: out[chain][256][256] = M(x)^-1 from the full sweep;  op 1: out[chain][256] = M(x)^-1 b by the
// trailing sweep + substitution;  op 2: the same by the full sweep + mat-vec;  op 3 / 4 (timing, tools/ubench_blk16.py):
// build + full / trailing sweep, `reps` times.  status[chain] = 0 if the metric was positive definite and finite, else 5.
template <int RMETRIC>
__global__ __launch_bounds__(NTHR, 2) void blk16la_debug_kernel(ImplicitArgs A, int base_ld, int op, int reps) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  TeamLa<RMETRIC> bk;
  init_backend(bk, A, base_ld, lds);
  const int64_t chain = blockIdx.x;
  const int tid = threadIdx.x, dim = A.dim;
  const bool act = tid < dim;
  const double q = act ? A.pos[chain * dim + tid] : 0.0;
  const double p = act ? A.mom[chain * dim + tid] : 0.0;
  double u = 0.0;
  bool ok = true;
  if (op >= 5) {
    if (tid < DPM) bk.lds[kOffB + tid] = act ? p : 0.0;
    const bool bad = bk.build(q);
    ok = op == 5 ? bk.template sweep<false, true>(bad) : bk.template sweep<true, true>(bad);
    __syncthreads();
    if (tid < 64) u = (double)reinterpret_cast<const long long*>(bk.lds + kOffPart)[tid];
    if (tid < DPM) A.out[chain * (int64_t)DPM + tid] = u;
  } else if (op >= 3) {
    for (int rep = 0; rep < reps; ++rep) ok = bk.construct(q + u * 1e-300, op == 3, p, &u) && ok;
    if (tid < DPM) A.out[chain * (int64_t)DPM + tid] = u + bk.acc[0][0];
  } else {
    ok = bk.construct(q, op != 1, p, &u);
    if (op == 0) {
      double* out = A.out + chain * (int64_t)(DPM * DPM);
      const int ln = fresh_lane(), g = ln >> 4, j = ln & 15;
      const auto m = bk.map();
      static_for<NSLOT>([&](auto sc_) {
        constexpr int s = decltype(sc_)::value;
        if (bk.class_known(s) || bk.in_use(s, m)) {
          const int I = bk.tile_i(s, m), J = bk.tile_j(s, m);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const double v = -bk.acc[s][r];
            out[(int64_t)(16 * I + 4 * r + g) * DPM + 16 * J + j] = v;
            out[(int64_t)(16 * J + j) * DPM + 16 * I + 4 * r + g] = v;
          }
        }
      });
    } else {
      if (op == 2) u = bk.matvec(p);
      if (tid < DPM) A.out[chain * (int64_t)DPM + tid] = u;
    }
  }
  if (tid == 0) A.status[chain] = ok ? 0 : MM_ST_LINALG;
}

template <class K, class... Extra>
int launch_la(mm_ctx* ctx, K kernel, const ImplicitArgs& a, int base_ld, Extra... extra) {
  const size_t lds = kLdsDoubles * sizeof(double);
  MM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kernel, dim3((unsigned)a.n_chains), dim3(NTHR), lds, ctx->stream, a, base_ld, extra...);
  MM_HIP_CHECK(ctx, hipGetLastError());
  return MM_OK;
}

int fill_args(mm_ctx* ctx, const mm_model* m, mm_state* s, ImplicitArgs& a) {
  if (m->dim > DPM) {
    mm_set_error(ctx, "block-16 matrix-core team kernel supports dim <= 256");
    return MM_ERR_UNSUPPORTED;
  }
  if (m->rmetric == MM_RMETRIC_RANK1 && (m->d_rmetric_padded == nullptr || m->rmetric_pad_dim < DPM)) {
    mm_set_error(ctx, "internal: rank-one base matrix was not padded for the team kernels");
    return MM_ERR_UNSUPPORTED;
  }
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
  return MM_OK;
}

}  // namespace

int mm_launch_implicit_blk16la(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int n_steps,
                               const mm_fp_opts& opts, mm_counters* d_counters) {
  ImplicitArgs a{};
  const int rc = fill_args(ctx, m, s, a);
  if (rc != MM_OK) return rc;
  a.step_size = h;
  a.n_steps = n_steps;
  a.opts = opts;
  a.counters = d_counters;
  if (m->rmetric == MM_RMETRIC_RANK1)
    return launch_la(ctx, implicit_blk16la_kernel<MM_RMETRIC_RANK1>, a, m->rmetric_pad_dim);
  return launch_la(ctx, implicit_blk16la_kernel<MM_RMETRIC_DIAGQUAD>, a, 0);
}

// developer / test hook of the look-ahead kernel, same contract as mm_debug_blk16_linalg (ops 0..4)
extern "C" int mm_debug_blk16la_linalg(mm_ctx* ctx, const mm_model* m, mm_state* s, int op, double* out,
                                       int32_t* status, int reps, double* ms) {
  if (!ctx || !m || !s || !out || op < 0 || op > 6 || m->rmetric == MM_RMETRIC_NONE ||
      m->rmetric == MM_RMETRIC_SOFTABS)
    return MM_ERR_INVALID;
  MM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  ImplicitArgs a{};
  const int rc = fill_args(ctx, m, s, a);
  if (rc != MM_OK) return rc;
  const size_t bytes = (size_t)s->n * (op == 0 ? (size_t)DPM * DPM : (size_t)DPM) * sizeof(double);
  double* d_out = nullptr;
  MM_HIP_CHECK(ctx, hipMalloc(&d_out, bytes));
  a.out = d_out;
  int lrc;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, ctx->stream);
  if (m->rmetric == MM_RMETRIC_RANK1)
    lrc = launch_la(ctx, blk16la_debug_kernel<MM_RMETRIC_RANK1>, a, m->rmetric_pad_dim, op, reps);
  else
    lrc = launch_la(ctx, blk16la_debug_kernel<MM_RMETRIC_DIAGQUAD>, a, 0, op, reps);
  (void)hipEventRecord(e1, ctx->stream);
  if (lrc == MM_OK && ms) {
    float f = 0.f;
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&f, e0, e1);
    *ms = f;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (lrc == MM_OK) {
    hipError_t e = hipMemcpyAsync(out, d_out, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && status)
      e = hipMemcpyAsync(status, s->d_status, (size_t)s->n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
      mm_set_error(ctx, std::string("mm_debug_blk16la_linalg: ") + hipGetErrorString(e));
      lrc = MM_ERR_HIP;
    }
  }
  (void)hipFree(d_out);
  return lrc;
}
