// Implicit leapfrog on SoftAbsRiemannianMetricSystem (D <= 64): one 1024-thread workgroup per chain,
// the Hessian / eigenvectors / work matrices in LDS.  gfx950 / CDNA4.
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
// eigh = parallel cyclic ONE-SIDED (Hestenes) Jacobi on G = H V: each round rotates D/2 disjoint column pairs
// (round-robin schedule) of G and V, a pair per half wave, one barrier per round; sweeps repeat until the columns
// of G are orthogonal (quadratic convergence, 7-8 sweeps cold, ~2 warm-started).  The result is used only
// through V f(lambda) V^T products, which do not depend on eigenvalue order or eigenvector signs.
// The matrix-Tressian products of the built-in targets need only the diagonal and the first row of
// their matrix argument, so V diag(g) V^T and A J A^T are never formed in full; the latter still needs
// the D^3 product B = A J (A = V diag(e)), done as an LDS-tiled FMA GEMM.
#include "implicit_core.h"

namespace {

using namespace mmdev;
using namespace mmimp;

constexpr int NT = 1024;           // 16 waves per chain: the kernel is LDS-latency bound, four waves per SIMD hide it
constexpr int TPD = 32;            // threads per matrix dimension in the 64 x 64 products (TPD^2 = NT)
constexpr int BS = 64 / TPD;       // output block side per thread
constexpr int RP = NT / 64;        // threads per output element of the row-wise reductions
constexpr int LD = 65;            // LDS leading dimension of the 64 x 64 matrices
constexpr int MAT = 64 * LD;
constexpr int LDJ = 72;           // leading dimension of the column-major Jacobi work matrices (eigh())
constexpr int MATJ = 64 * LDJ;
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
  double* rc;   // rotation (c, s) per pair slot, two rounds: [2][32][2]
  double* red;  // [16]
  double* cnt;  // [8] work counters of the chain (thread 0)
  double* stash;  // [SL_COUNT][65]
};
constexpr int kLdsDoubles = MAT + 2 * MATJ + 6 * 64 + 4 * 32 + 16 + 8 + SL_COUNT * 65;

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

__device__ __forceinline__ double block_reduce4(double v, int kind_max, double* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = kind_max ? wave_max(v) : wave_sum(v);
  if (lane == 0) red[wave] = v;
  __syncthreads();
  double r = red[0];
#pragma unroll
  for (int w = 1; w < NT / 64; ++w) r = kind_max ? nanmax(r, red[w]) : r + red[w];
  __syncthreads();
  return uniform_f64(r);
}

// sum over the RP = 16 consecutive lanes (one DPP row) that share an output element
__device__ __forceinline__ double rp_sum(double v) {
  v = group8_sum(v);
  return v + dpp_move<kDppMirror>(v);
}
static_assert(RP == 16, "rp_sum reduces a 16-lane DPP row");

// -DMM_SOFTABS_PROF: per-phase cycle totals of block 0 (thread 0's clock), printed at the end of the launch
#ifdef MM_SOFTABS_PROF
#define SA_PROF_BEGIN() const long long prof_t0_ = __builtin_readcyclecounter()
#define SA_PROF_END(slot_) \
  do { if (tid_raw == 0) w.cnt[slot_] += (double)(__builtin_readcyclecounter() - prof_t0_); } while (0)
#else
#define SA_PROF_BEGIN() do {} while (0)
#define SA_PROF_END(slot_) do {} while (0)
#endif

struct SoftAbsBackend {
  static constexpr bool kSolveByInverse = false;  // implicit_core.h
  static constexpr bool kUnifiedConstruct = false;
  static constexpr bool kCountersInLds = true;  // implicit_core.h: work counters in LDS, bumped by thread 0
  int dim, tid_raw, target;
  int warm = 0;  // eigendecompositions since the last cold start (0: w.V is not a usable basis)
  int n_sweeps = 0, n_eigh = 0;  // work counters (reported as n_newton_iters / n_inverse)
  double coeff;
  SaLds w;
  const double* tparams;

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
  // flat state exists for tid < 64; the other threads share a dummy cell (index 64) per slot
  __device__ __forceinline__ double& slot(int i) { return w.stash[i * 65 + (tid < 64 ? tid : 64)]; }

  __device__ __forceinline__ double norm(double x, int kind) {
    const double a = tid < dim ? x : 0.0;
    if (kind == MM_NORM_LINF) return block_reduce4(fabs(a), 1, w.red);
    return sqrt(block_reduce4(a * a, 0, w.red));
  }

  // ---- hess_neg_log_dens(q) into w.H (systems.py:1870-1888); q flat --------------------------------
  __device__ __forceinline__ void build_hessian(double q) {
    if (tid < 64) w.nat[tid] = (tid < dim) ? q : 0.0;
    __syncthreads();
    const double* x = w.nat;
    double e = 0.0, s = 0.0;
    if (target == MM_TARGET_FUNNEL) {
      e = exp(-x[0]);
      double acc = 0.0;
      for (int i = 1 + (threadIdx.x & 63); i < dim; i += 64) acc += tparams[i - 1] * x[i] * x[i];
      s = wave_sum(acc);  // every wave computes the same S = sum w x^2
    }
    for (int i = tid >> 6; i < dim; i += NT / 64) {
      const int j = tid & 63;
      if (j >= dim) continue;
      double h = 0.0;
      if (target == MM_TARGET_POLY) {
        if (i == j) h = tparams[0] + 3.0 * tparams[1] * x[i] * x[i];
      } else {  // funnel: arrowhead
        if (i == 0 && j == 0) h = 1.0 / 9.0 + 0.5 * e * s;
        else if (i == 0) h = -e * tparams[j - 1] * x[j];
        else if (j == 0) h = -e * tparams[i - 1] * x[i];
        else if (i == j) h = e * tparams[i - 1];
      }
      w.H[i * LD + j] = h;
      if (warm == 0) w.V[i * LD + j] = (i == j) ? 1.0 : 0.0;
    }
    __syncthreads();
  }

  // G = H V, written COLUMN-major (leading dimension LDJ) into w.W, rows >= dim zeroed: the start of the one-sided
  // Jacobi.  With a cold start V = I and G = H.  Thread (ti, tj) owns rows {ti, ti + 32} x columns {2 tj, 2 tj + 1}:
  // the H reads of a 32-lane group are 32 consecutive rows (stride LD = 65 doubles: conflict-free), the V reads are
  // broadcasts and the G writes are contiguous.
  __device__ __forceinline__ void times_basis() {
    const int ti = tid % TPD, bj = (tid / TPD) * BS;
    double acc[BS][BS];
#pragma unroll
    for (int a = 0; a < BS; ++a)
#pragma unroll
      for (int b = 0; b < BS; ++b) acc[a][b] = 0.0;
    for (int k = 0; k < dim; ++k) {
      double hv[BS], vv[BS];
#pragma unroll
      for (int a = 0; a < BS; ++a) hv[a] = w.H[(ti + TPD * a) * LD + k];
#pragma unroll
      for (int b = 0; b < BS; ++b) vv[b] = w.V[k * LD + bj + b];
#pragma unroll
      for (int a = 0; a < BS; ++a)
#pragma unroll
        for (int b = 0; b < BS; ++b) acc[a][b] = __builtin_fma(hv[a], vv[b], acc[a][b]);
    }
#pragma unroll
    for (int a = 0; a < BS; ++a)
#pragma unroll
      for (int b = 0; b < BS; ++b) w.W[(bj + b) * LDJ + ti + TPD * a] = (ti + TPD * a < dim) ? acc[a][b] : 0.0;
    __syncthreads();
  }

  // 1/sqrt(x) to rounding accuracy: the hardware estimate and two Newton steps (x in the normal range)
  __device__ static __forceinline__ double rsqrt_newton(double x) {
    double y = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    y = y * __builtin_fma(-hx * y, y, 1.5);
    y = y * __builtin_fma(-hx * y, y, 1.5);
    return y;
  }

  // ---- eigh(H) by parallel ONE-SIDED (Hestenes) Jacobi: w.lam = eigenvalues, w.V = eigenvectors ----------------
  // Columns of G = H V and of V are rotated together until the columns of G are mutually orthogonal; then
  // H V = V diag(lam) with lam_i = g_i . v_i (which carries the sign: H is indefinite in general).  A round rotates
  // D/2 disjoint column pairs (round-robin schedule) and costs ONE workgroup barrier; the two-sided form this
  // replaces needed three (parameters, columns, rows) and every wave recomputed every rotation's parameters.
  //
  // A round is bound by VALU issue (a wave64 FP64 instruction holds its SIMD for 4 cycles whatever the number of
  // lanes that matter) and by the LDS write port, so the work is laid out to issue as few wave instructions as
  // possible: a pair belongs to EIGHT lanes (8 rows each), eight pairs to a wave, so four waves - one per SIMD -
  // cover the 32 pairs and the three dot products of a pair are 3-step DPP reductions.  Waves 0-3 own G: dots,
  // rotation parameters, rotation of G, and publish (c, s); waves 4-7 (the second wave of each SIMD) apply the
  // rotations of the PREVIOUS round to V, which nothing reads until the sweeps end - the two roles hide each
  // other's LDS and dependent-issue latencies.  G^T and V^T live in LDS with a leading dimension of 72 doubles: the
  // 32 lanes of a ds_read_b64 group (4 adjacent pair slots x 8 rows) then hit 32 distinct 8-byte slots
  // (8 ((col + j) mod 4) + row mod 8), and likewise the 16-lane groups of ds_write_b64.  V is transposed into the
  // dead H buffer on the way in and back on the way out (2 x 32 KB of LDS traffic per decomposition, ~1 round's
  // worth).
  //
  // Warm start: consecutive metric constructions of a step are at nearby positions, so the previous eigenvectors
  // almost diagonalise the new Hessian (G = H V_prev is nearly orthogonal): ~2.5 sweeps instead of 7-8; a cold start
  // every kWarmPeriod decompositions bounds the accumulated loss of orthogonality of V.  Jacobi converges
  // quadratically, so a sweep whose largest |cos(g_p, g_q)| was below 1e-7 is the last one.  The result is used only
  // through V f(lam) V^T forms, which do not depend on eigenvalue order or eigenvector signs.
  __device__ __forceinline__ bool eigh() {
    const int n2 = dim + (dim & 1);
    const int half = n2 >> 1;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int role = wave >> 2;                   // 0: G, 1: V, 2-3: only the barriers
    const int slot = 8 * (wave & 3) + (lane >> 3);  // the pair slot of this group of 8 lanes
    const int sub = lane & 7;                     // rows sub + 8 j
    bool converged = false;
    ++n_eigh;
    SA_PROF_BEGIN();
    times_basis();
    double* const G = w.W;
    double* const Vt = w.H;
    double* const prm = w.rc;  // (c, s) per pair slot, double buffered by round parity: [2][32][2]
    for (int e = tid; e < 64 * 64; e += NT) {
      const int i = e & 63, j = e >> 6;
      Vt[j * LDJ + i] = (i < dim && j < dim) ? w.V[i * LD + j] : 0.0;
    }
    __syncthreads();
    SA_PROF_END(5);
    // The tournament as byte offsets of the two columns of this pair slot, advanced from round to round without
    // multiplications: slot 0 keeps column n2 - 1 and meets r; slot t > 0 has (r + t, r - t) mod (n2 - 1).
    const int span = (n2 - 1) * LDJ * 8;  // one lap of the moving columns
    const int step_a = slot == 0 ? 0 : LDJ * 8;  // (both moving columns advance by one column a round)
    const int a_first = (slot == 0 ? n2 - 1 : slot) * LDJ * 8 + sub * 8;
    const int b_first = (slot == 0 ? 0 : n2 - 1 - slot) * LDJ * 8 + sub * 8;
    const int lap_a = slot == 0 ? 0x7fffffff : span;  // (slot 0's resident column n2 - 1 is beyond the lap)
    const int dummy = dim * LDJ * 8 + sub * 8;  // odd dim: the column that does not exist
    const bool odd = (dim & 1) != 0;
    const bool mine = slot < half;
    int oa = a_first, ob = b_first;    // the round this role works on next (G: round g; V: round g - 1)
    int g = 0;                         // rounds done by the G role
    for (int sweep = 0; sweep < kMaxSweeps; ++sweep) {
      double big = 0.0, bad = 0.0;
      ++n_sweeps;
      for (int r = 0; r < n2 - 1; ++r) {
#ifndef MM_SA_EXP
#define MM_SA_EXP 0
#endif
        if (MM_SA_EXP == 4) {
        } else if (role == 0) {
          if (mine) {
            double c = 1.0, s = 0.0;
            if (!odd || (oa != dummy && ob != dummy)) {  // (odd dim: the dummy column's partner sits this round out)
              char* const ca = reinterpret_cast<char*>(G) + oa;
              char* const cb = reinterpret_cast<char*>(G) + ob;
              double xa[8], xb[8];
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                xa[j] = lds_read(ca + 64 * j);
                xb[j] = lds_read(cb + 64 * j);
              }
              double al = 0.0, be = 0.0, ga = 0.0;
#pragma unroll
              for (int j = 0; j < 8; ++j) {
                al = __builtin_fma(xa[j], xa[j], al);
                be = __builtin_fma(xb[j], xb[j], be);
                ga = __builtin_fma(xa[j], xb[j], ga);
              }
              al = group8_sum(al);
              be = group8_sum(be);
              ga = group8_sum(ga);
              const double ab = al * be, gg = ga * ga;
              if (!(ab <= 1.7e308) || !(gg <= 1.7e308)) bad = 1.0;  // NaN or overflow
              if (gg > 1e-14 * ab) big = 1.0;   // |cos| > 1e-7: another sweep is needed after this one
              if (MM_SA_EXP == 3) {
                if (gg == 1.2345) big = 2.0;
              } else if (MM_SA_EXP == 2) {
                c = 0.8; s = gg == 1.2345 ? 0.5 : 0.6;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                  lds_write(ca + 64 * j, __builtin_fma(c, xa[j], -(s * xb[j])));
                  lds_write(cb + 64 * j, __builtin_fma(s, xa[j], c * xb[j]));
                }
              } else
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
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                  lds_write(ca + 64 * j, __builtin_fma(c, xa[j], -(s * xb[j])));
                  lds_write(cb + 64 * j, __builtin_fma(s, xa[j], c * xb[j]));
                }
              }
            }
            if (sub == 0) {
              prm[(g & 1) * 64 + 2 * slot] = c;
              prm[(g & 1) * 64 + 2 * slot + 1] = s;
            }
          }
        } else if (role == 1 && g > 0 && MM_SA_EXP != 1) {
          if (mine) rotate_basis(Vt, prm + ((g - 1) & 1) * 64 + 2 * slot, oa, ob);
        }
        if (role == 0 || g > 0) {  // next round's columns
          oa += step_a; if (oa >= lap_a) oa -= span;
          ob += LDJ * 8; if (ob >= span) ob -= span;
        }
        ++g;
        __syncthreads();
      }
      bad = block_reduce4(bad, 0, w.red);
      if (bad != 0.0) {
        warm = 0;
        return false;
      }
      big = block_reduce4(big, 0, w.red);
      if (MM_SA_EXP != 0) big = sweep < 3 ? 1.0 : 0.0;
      if (big == 0.0) {
        converged = true;
        break;
      }
    }
    if (role == 1 && mine) rotate_basis(Vt, prm + ((g - 1) & 1) * 64 + 2 * slot, oa, ob);
    __syncthreads();
    {  // lam_i = g_i . v_i, 16 lanes per column
      const int i = tid / RP, part = tid % RP;
      double lam = 0.0;
#pragma unroll
      for (int m = 0; m < 64 / RP; ++m) lam = __builtin_fma(G[i * LDJ + part + RP * m], Vt[i * LDJ + part + RP * m], lam);
      lam = rp_sum(lam);
      if (part == 0) w.lam[i] = (i < dim) ? lam : 1.0;
    }
    for (int e = tid; e < 64 * 64; e += NT) {
      const int i = e & 63, j = e >> 6;
      if (i < dim && j < dim) w.V[i * LD + j] = Vt[j * LDJ + i];
    }
    __syncthreads();
    warm = converged ? (warm + 1) % kWarmPeriod : 0;
    SA_PROF_END(4);
    return converged;
  }

  // One 8-byte LDS access per instruction: ds_read_b64 runs at 256 B/clk, the ds_read2_b64 the compiler would merge
  // two of these into at 128 B/clk (MI355X_MICROARCH.md, LDS), and the Jacobi rounds are bound by the LDS array.
  __device__ static __forceinline__ double lds_read(const char* p) {
    return *reinterpret_cast<const double*>(p);
  }
  __device__ static __forceinline__ void lds_write(char* p, double v) { *reinterpret_cast<double*>(p) = v; }

  // the V role of a round: the two columns of V^T at byte offsets oa, ob rotated by the (c, s) the G role published
  __device__ static __forceinline__ void rotate_basis(double* Vt, const double* cs, int oa, int ob) {
    const double c = cs[0], s = cs[1];
    if (s == 0.0) return;  // skipped pair (c = 1)
    char* const ca = reinterpret_cast<char*>(Vt) + oa;
    char* const cb = reinterpret_cast<char*>(Vt) + ob;
    double xa[8], xb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      xa[j] = lds_read(ca + 64 * j);
      xb[j] = lds_read(cb + 64 * j);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      lds_write(ca + 64 * j, __builtin_fma(c, xa[j], -(s * xb[j])));
      lds_write(cb + 64 * j, __builtin_fma(s, xa[j], c * xb[j]));
    }
  }

  // softabs(x) = x / tanh(coeff x); grad_softabs (matrices.py:1662-1669)
  __device__ __forceinline__ bool regularise() {
    double bad = 0.0;
    if (tid < 64) {
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
    return block_reduce4(bad, 0, w.red) == 0.0;
  }

  __device__ __forceinline__ bool build_and_invert(double x) {
    build_hessian(x);
    if (!eigh()) return false;
    return regularise();
  }

  // V^T v (flat in, flat out): RP threads share output k, each sums every RP-th term
  __device__ __forceinline__ double vt_times(double v) {
    if (tid < 64) w.v1[tid] = (tid < dim) ? v : 0.0;
    __syncthreads();
    {
      const int k = tid / RP, part = tid % RP;
      double s = 0.0;
      if (k < dim)
        for (int i = part; i < dim; i += RP) s = __builtin_fma(w.V[i * LD + k], w.v1[i], s);
      s = rp_sum(s);
      __syncthreads();  // every thread has read v1
      if (part == 0) w.v1[k] = s;
    }
    __syncthreads();
    const double out = (tid < dim) ? w.v1[tid] : 0.0;
    __syncthreads();
    return out;
  }
  // V v
  __device__ __forceinline__ double v_times(double v) {
    if (tid < 64) w.v2[tid] = (tid < dim) ? v : 0.0;
    __syncthreads();
    {
      const int i = tid / RP, part = tid % RP;
      double s = 0.0;
      if (i < dim)
        for (int k = part; k < dim; k += RP) s = __builtin_fma(w.V[i * LD + k], w.v2[k], s);
      s = rp_sum(s);
      __syncthreads();
      if (part == 0) w.v2[i] = s;
    }
    __syncthreads();
    const double out = (tid < dim) ? w.v2[tid] : 0.0;
    __syncthreads();
    return out;
  }

  // M^-1 v = V diag(1/lamt) V^T v   (matrices.py:1568-1575, 1623-1624)
  __device__ __forceinline__ bool build_and_solve(double x, double rhs, double* u) {
    const bool ok = build_and_invert(x);
    *u = matvec(rhs);
    return ok;
  }

  __device__ __forceinline__ double matvec(double v) {
    const double c = vt_times(v);
    return v_times(tid < dim ? (1.0 / w.lamt[tid]) * c : 0.0);
  }

  // mtp_neg_log_dens(q)(m) given only what the built-in Tressians touch: m_ii (md) and the symmetric
  // first row m_0i (m0), flat.  systems.py:1890-1920; closed forms SURVEY.md Appendix A.
  __device__ __forceinline__ double mtp(double q, double md, double m0) {
    if (target == MM_TARGET_POLY) return 6.0 * tparams[1] * q * md;
    // funnel: q = (v, x)
    if (tid < 64) {
      w.v1[tid] = (tid < dim) ? q : 0.0;
      w.v2[tid] = (tid < dim) ? md : 0.0;
      w.nat[tid] = (tid < dim) ? m0 : 0.0;
    }
    __syncthreads();
    const double ev = exp(-w.v1[0]);
    const double mvv = w.v2[0];
    double a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (tid >= 1 && tid < dim) {
      const double wi = tparams[tid - 1], xi = w.v1[tid];
      a1 = wi * xi * xi;                    // S
      a2 = 2.0 * w.nat[tid] * wi * xi;      // (m_vi + m_iv) w_i x_i
      a3 = w.v2[tid] * wi;                  // m_ii w_i
    }
    const double S = block_reduce4(a1, 0, w.red);
    const double s2 = block_reduce4(a2, 0, w.red);
    const double s3 = block_reduce4(a3, 0, w.red);
    double out = 0.0;
    if (tid == 0) out = -0.5 * ev * S * mvv + ev * s2 - ev * s3;
    else if (tid < dim) {
      const double wk = tparams[tid - 1];
      out = ev * wk * w.v1[tid] * mvv - ev * wk * (2.0 * w.nat[tid]);
    }
    __syncthreads();
    return out;
  }

  // 0.5 * mtp(grad_log_abs_det), grad_log_abs_det = V diag(grad_softabs(lam)/lamt) V^T  (:1671-1674)
  __device__ __forceinline__ double half_vjp_inv(double q) {
    SA_PROF_BEGIN();
    {
      const int i = tid / RP, part = tid % RP;
      double md = 0.0, m0 = 0.0;
      if (i < dim) {
        for (int k = part; k < dim; k += RP) {
          const double g = w.gsa[k] / w.lamt[k];
          const double vik = w.V[i * LD + k];
          md = __builtin_fma(vik * vik, g, md);
          m0 = __builtin_fma(w.V[k] * vik, g, m0);  // V[0][k] V[i][k] g_k
        }
      }
      md = rp_sum(md);
      m0 = rp_sum(m0);
      if (part == 0) { w.v2[i] = md; w.v1[i] = m0; }
    }
    __syncthreads();
    const double mdf = (tid < dim) ? w.v2[tid] : 0.0;
    const double m0f = (tid < dim) ? w.v1[tid] : 0.0;
    __syncthreads();
    const double out = 0.5 * mtp(q, mdf, m0f);
    SA_PROF_END(7);
    return out;
  }

  // 0.5 * mtp(grad_quadratic_form_inv(p)),  -(V (e e^T o J) V^T) = -A J A^T, A = V diag(e),
  // e = V^T p / lamt, J_kl = (lamt_k - lamt_l)/(lam_k - lam_l), J_kk = grad_softabs(lam_k)  (:1676-1685)
  __device__ __forceinline__ double dh2_dpos(double p, double q) {
    SA_PROF_BEGIN();
    const double c = vt_times(p);
    if (tid < 64) w.v1[tid] = (tid < dim) ? c / w.lamt[tid] : 0.0;  // e
    __syncthreads();
    // J into w.H, A into w.W
    for (int k = tid >> 6; k < dim; k += NT / 64) {
      const int l = tid & 63;
      if (l >= dim) continue;
      double num = w.lamt[k] - w.lamt[l], den = w.lam[k] - w.lam[l];
      if (k == l) { num += w.gsa[k]; den = 1.0; }
      w.H[k * LD + l] = num / den;                 // 0/0 -> NaN for degenerate spectra, as the reference
      w.W[k * LD + l] = w.V[k * LD + l] * w.v1[l]; // A[i=k][k=l]
    }
    __syncthreads();
    // md_i = sum_kl A_ik J_kl A_il ; m0_i = sum_kl A_0k J_kl A_il : thread i accumulates over l of
    // (sum_k A_ik J_kl) A_il.  Work split: RP threads per row i (each every RP-th l).
    double md = 0.0, m0 = 0.0;
    {
      const int i = tid / RP, part = tid % RP;
      if (i < dim) {
        for (int l = part; l < dim; l += RP) {
          double bi = 0.0, b0 = 0.0;
          for (int k = 0; k < dim; ++k) {
            const double jkl = w.H[k * LD + l];
            bi = __builtin_fma(w.W[i * LD + k], jkl, bi);
            b0 = __builtin_fma(w.W[k], jkl, b0);  // A[0][k]
          }
          const double ail = w.W[i * LD + l];
          md = __builtin_fma(bi, ail, md);
          m0 = __builtin_fma(b0, ail, m0);
        }
      }
      md = rp_sum(md);
      m0 = rp_sum(m0);
      __syncthreads();
      if (part == 0 && i < 64) { w.v2[i] = -md; w.nat[i] = -m0; }
      __syncthreads();
    }
    const double mdf = (tid < dim) ? w.v2[tid] : 0.0;
    const double m0f = (tid < dim) ? w.nat[tid] : 0.0;
    __syncthreads();
    const double out = 0.5 * mtp(q, mdf, m0f);
    SA_PROF_END(6);
    return out;
  }

  __device__ __forceinline__ double grad(double q) {
    if (tid < 64) w.nat[tid] = (tid < dim) ? q : 0.0;
    __syncthreads();
    const TargetAux aux = target_prepare<true>(target, w.nat, dim, tparams, threadIdx.x & 63);
    const double g = (tid < dim) ? target_grad_elem<true>(target, aux, w.nat, tid, dim, tparams) : 0.0;
    __syncthreads();
    return g;
  }
  __device__ __forceinline__ double nld_elem(double q) {
    if (tid < 64) w.nat[tid] = (tid < dim) ? q : 0.0;
    __syncthreads();
    const TargetAux aux = target_prepare<true>(target, w.nat, dim, tparams, threadIdx.x & 63);
    const double e = (tid < dim) ? target_nld_elem<true>(target, aux, w.nat, tid, dim, tparams) : 0.0;
    __syncthreads();
    return e;
  }
};

__device__ __forceinline__ void init_backend(SoftAbsBackend& bk, const ImplicitArgs& A, double* lds) {
  bk.dim = A.dim;
  bk.tid.v = threadIdx.x;
  bk.tid_raw = threadIdx.x;
  bk.target = A.target;
  bk.coeff = uniform_f64(A.z[0]);  // softabs coefficient (device copy of the model's rmetric_params)
  bk.tparams = A.tparams;
  double* p = lds;
  bk.w.H = p; p += MATJ;
  bk.w.W = p; p += MATJ;
  bk.w.V = p; p += MAT;
  bk.w.lam = p; p += 64;
  bk.w.lamt = p; p += 64;
  bk.w.gsa = p; p += 64;
  bk.w.v1 = p; p += 64;
  bk.w.v2 = p; p += 64;
  bk.w.nat = p; p += 64;
  bk.w.rc = p; p += 4 * 32;
  bk.w.red = p; p += 16;
  bk.w.cnt = p; p += 8;
  bk.w.stash = p;
  if (threadIdx.x < 8) bk.w.cnt[threadIdx.x] = 0.0;
}

struct SaArgs {
  ImplicitArgs a;
  const double* coeff;  // device pointer to softabs_coeff
  int op;
};

// MIDPOINT: ImplicitMidpointIntegrator (integrators.py:547-681) on the same backend
template <bool MIDPOINT>
__global__ __launch_bounds__(NT) void softabs_leapfrog_kernel(SaArgs S) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  ImplicitArgs A = S.a;
  A.z = S.coeff;
  const int64_t chain = blockIdx.x;
  SoftAbsBackend bk;
  init_backend(bk, A, lds);
  const int dim = A.dim, tid = threadIdx.x;
  const bool act = tid < dim;
  const double q = act ? A.pos[chain * dim + tid] : 0.0;
  const double p = act ? A.mom[chain * dim + tid] : 0.0;
  const double t = uniform_f64(signed_step(A.dir, A.step_scale, chain, A.step_size));
  bk.slot(SL_Q) = q;
  bk.slot(SL_P) = p;
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
#ifdef MM_SOFTABS_PROF
  if (tid == 0 && chain == 0)
    printf("softabs prof: total %lld eigh(incl basis) %.0f basis %.0f dh2_dpos %.0f half_vjp %.0f | n_eigh %d sweeps %d "
           "evals %.0f\n", (long long)(__builtin_readcyclecounter() - prof_start), bk.w.cnt[4], bk.w.cnt[5], bk.w.cnt[6],
           bk.w.cnt[7], bk.n_eigh, bk.n_sweeps, bk.w.cnt[CNT_EVALS]);
#endif
  if (tid == 0) {
    A.status[chain] = r.status;
    A.n_done[chain] = r.done;
    add_counters(A.counters, r);
    if (A.counters) {
      atomicAdd((unsigned long long*)&A.counters->n_newton_iters, (unsigned long long)bk.n_sweeps);
      atomicAdd((unsigned long long*)&A.counters->reserved, (unsigned long long)bk.n_eigh);
    }
  }
}

// op 0: h = l + 0.5 logdet + 0.5 p^T M^-1 p ; 1: dh_dmom ; 2: sample_momentum = V diag(sqrt(lamt)) V^T z
__global__ __launch_bounds__(NT) void softabs_aux_kernel(SaArgs S, double* out, const double* z) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  ImplicitArgs A = S.a;
  A.z = S.coeff;
  const int64_t chain = blockIdx.x;
  SoftAbsBackend bk;
  init_backend(bk, A, lds);
  const int dim = A.dim, tid = threadIdx.x;
  const bool act = tid < dim;
  const double q = act ? A.pos[chain * dim + tid] : 0.0;
  const double p = act ? A.mom[chain * dim + tid] : 0.0;
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
  const bool ok = bk.build_and_invert(q);
  if (S.op == 0) {
    const double u = bk.matvec(p);
    double e = bk.nld_elem(q) + (act ? 0.5 * p * u + 0.5 * log(fabs(bk.w.lamt[tid])) : 0.0);
    e = block_reduce4(e, 0, bk.w.red);
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

SaArgs make_args(const mm_model* m, mm_state* s) {
  SaArgs S{};
  S.a.pos = s->d_pos;
  S.a.mom = s->d_mom;
  S.a.dir = s->d_dir;
  S.a.step_scale = s->d_step_scale;
  S.a.chain_steps = s->d_chain_steps;
  S.a.status = s->d_status;
  S.a.n_done = s->d_n_done;
  S.a.n_chains = s->n;
  S.a.dim = s->dim;
  S.a.target = m->target;
  S.a.tparams = m->d_target_params;
  S.coeff = m->d_rmetric_params;
  return S;
}

int check_dim(mm_ctx* ctx, const mm_model* m) {
  if (m->dim > 64) {
    mm_set_error(ctx, "SoftAbs kernels support dim <= 64 (LDS-resident eigendecomposition)");
    return MM_ERR_UNSUPPORTED;
  }
  return MM_OK;
}

}  // namespace

template <bool MIDPOINT>
static int launch_softabs(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int n_steps,
                          const mm_fp_opts& opts, mm_counters* d_counters) {
  int rc = check_dim(ctx, m);
  if (rc != MM_OK) return rc;
  SaArgs S = make_args(m, s);
  S.a.step_size = h;
  S.a.n_steps = n_steps;
  S.a.opts = opts;
  S.a.counters = d_counters;
  const size_t lds = kLdsDoubles * sizeof(double);
  MM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(softabs_leapfrog_kernel<MIDPOINT>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(softabs_leapfrog_kernel<MIDPOINT>, dim3((unsigned)s->n), dim3(NT), lds, ctx->stream, S);
  MM_HIP_CHECK(ctx, hipGetLastError());
  return MM_OK;
}

int mm_launch_softabs_leapfrog(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int n_steps,
                               const mm_fp_opts& opts, mm_counters* d_counters) {
  return launch_softabs<false>(ctx, m, s, h, n_steps, opts, d_counters);
}

int mm_launch_softabs_midpoint(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int n_steps,
                               const mm_fp_opts& opts, mm_counters* d_counters) {
  return launch_softabs<true>(ctx, m, s, h, n_steps, opts, d_counters);
}

int mm_launch_softabs_aux(mm_ctx* ctx, const mm_model* m, mm_state* s, int op, double* d_out,
                          const double* d_z) {
  int rc = check_dim(ctx, m);
  if (rc != MM_OK) return rc;
  SaArgs S = make_args(m, s);
  S.op = op;
  const size_t lds = kLdsDoubles * sizeof(double);
  MM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(softabs_aux_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(softabs_aux_kernel, dim3((unsigned)s->n), dim3(NT), lds, ctx->stream, S, d_out, d_z);
  MM_HIP_CHECK(ctx, hipGetLastError());
  return MM_OK;
}
