// Device code of the FORKED matrix-core wave kernel (round 6; k_implicit_fork.hip instantiates it for the built-in metrics,
// mm_rtc.hip compiles it at run time around a USER metric).
//
// Implicit leapfrog on dense-metric Riemannian systems, 32 < D <= 64: the one-wave-per-chain kernel of implicit_mfma.h with a
// SECOND wave per chain that takes the reversibility-check solve of C off the first wave's hands.  gfx950 / CDNA4.
//
// Why.  At BASELINE c3 there is one chain per SIMD and the chain's step is one dependent
// instruction stream; the lone wave issues in half its cycles.  Splitting the chain's linear algebra over two waves was
// built and measured this round (implicit_pair.h): every product / reduction becomes an LDS exchange behind a barrier and
// the halved stream does not pay for them.  What a step does offer is TASK parallelism: after C's forward map the
// reference solves  x = q' - t M(x)^-1 p  (the reversibility check, integrators.py:521-528) and then
// x = q' + t M(x)^-1 p  (the C-adjoint map, :530-536) - two fixed-point solves from the same point that do not depend on
// each other; together they are half of a step's work (10 of its 11 metric constructions, 29 refinement pairs).  Here wave
// A of a chain - the one-wave kernel as it is - hands the check's solve to wave B and runs the C-adjoint solve itself; they
// meet again before the metric at the new position is built.  Between fork and join the two waves share NOTHING but the
// read-only staged base matrix: no barrier, no exchange - one mailbox hand-shake each way.
//
// Wave B needs the explicit inverse M(q)^-1 the refinement solves are preconditioned with: when wave A converts its tiles
// to rows (sixteen columns at a time through LDS, tiles_to_rows) wave B reads the same buffer into its own 128 registers -
// four flag hand-shakes a step.  Registers: both waves of a chain must fit one SIMD (2048 waves on 1024 SIMDs), i.e. 256
// registers a lane, all architected - and that cap is what the kernel pays for: with wave B idle it runs at 1.05e7 steps/s
// on c3 against the one-wave kernel's 1.36e7 (320 registers: the allocator now keeps four entries of the row in scratch and
// the scheduler hoists fewer LDS reads; a row tail in LDS and throttled product groups were measured and do not pay,
// profiles/r06_ab_c3_fork.txt); with the fork it reaches 1.48e7.
//
// Results: each solve runs exactly the arithmetic of the sequential kernel (the same refine_solve / fp_feed calls on the
// same inputs), so positions, momenta, status and the fixed-point counters are bit for bit those of implicit_mfma.h.  A
// failing check reports its own status whatever the adjoint solve did meanwhile; a refinement failure on wave B hands the
// check back to wave A, which factorises as before.
#pragma once
#include "implicit_mfma.h"

namespace mmfork {

using namespace mmdev;
using namespace mmimp;
using mmmfma::d2;
using mmmfma::d4;
using mmmfma::kBaseDoubles;
using mmmfma::kBasePitch;
using mmmfma::kMfmaWaveDoubles;
using mmmfma::kRowPitch;
using mmmfma::kTiles;
using mmmfma::tix;

constexpr int kChains = 4;             // chains per workgroup (they share the staged base matrix): 8 waves
constexpr int kHelperDoubles = 256;    // wave B's own LDS: qt, nat, aux (64 each), mailbox (64)
enum { CMD_ROWS = 1, CMD_CHK = 2, CMD_EXIT = 3 };
// mailbox (in wave B's block): int flags [0] cmd_seq [1] cmd_code [2] ack_seq [3] chunk_ready [4] chunk_ack; doubles from [8]
enum { MB_ITER = 8, MB_STAGE, MB_T, MB_OUTCOME, MB_STATUS, MB_RITER, MB_RSTAGE, MB_NEVALS, MB_NPAIRS };

template <int RMETRIC>
__host__ __device__ constexpr int fork_lds_doubles() {
  // (a user metric: both waves carry their own copy of the hooks' LDS blocks, user_metric.h)
  return (RMETRIC == MM_RMETRIC_RANK1 ? kBaseDoubles : 0) +
         kChains * (kMfmaWaveDoubles + kHelperDoubles + (RMETRIC == MM_RMETRIC_USER ? 2 * mmuser::lds_doubles(64) : 0));
}

template <int RMETRIC>
struct ForkBackend : mmmfma::MfmaBackend<RMETRIC, false> {
  using Base = mmmfma::MfmaBackend<RMETRIC, false>;
  using Base::acc;
  using Base::dim;
  using Base::lane;
  using Base::w;
  using Base::fr_;
  using Base::fd_;
  static constexpr bool kDual = false;
  static constexpr bool kFork = true;
  bool fork_off;   // MICI_AMD_FORK=0: wave B idles (A/B runs)
  int mseq;        // commands sent (wave A) / served (wave B)
  int rseq;        // row chunks sent / received
  double* mb;      // the chain's mailbox
  double* mstash;  // wave A's slots (wave B reads the check's state there and writes it back)
  double* rowbuf;  // wave A's row-conversion buffer (wave B reads its copy of the inverse's rows there)

  __device__ __forceinline__ int* flags() const { return reinterpret_cast<int*>(mb); }
  __device__ __forceinline__ void set_flag(int idx, int v) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_store(flags() + idx, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __device__ __forceinline__ void wait_flag(int idx, int want) {
    while (__hip_atomic_load(flags() + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < want)
      __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
  __device__ __forceinline__ bool fork_on() const { return !fork_off; }

  // ---- wave A ----------------------------------------------------------------------------------------------------------
  __device__ __forceinline__ void send(int code) {
    ++mseq;
    if (lane == 0) flags()[1] = code;
    set_flag(0, mseq);
  }
  __device__ __forceinline__ void fork_chk(int iter, int stage, double t) {
    if (lane == 0) {
      mb[MB_ITER] = (double)iter;
      mb[MB_STAGE] = (double)stage;
      mb[MB_T] = t;
    }
    send(CMD_CHK);
  }
  __device__ __forceinline__ ForkOutcome join_chk() {
    wait_flag(2, mseq);
    ForkOutcome fo;
    fo.outcome = (int)mb[MB_OUTCOME];
    fo.status = (int)mb[MB_STATUS];
    fo.iter = (int)mb[MB_RITER];
    fo.stage = (int)mb[MB_RSTAGE];
    fo.n_evals = (int)mb[MB_NEVALS];
    fo.n_pairs = (int)mb[MB_NPAIRS];
    fo.outcome = __builtin_amdgcn_readfirstlane(fo.outcome);
    fo.status = __builtin_amdgcn_readfirstlane(fo.status);
    fo.iter = __builtin_amdgcn_readfirstlane(fo.iter);
    fo.stage = __builtin_amdgcn_readfirstlane(fo.stage);
    fo.n_evals = __builtin_amdgcn_readfirstlane(fo.n_evals);
    fo.n_pairs = __builtin_amdgcn_readfirstlane(fo.n_pairs);
    return fo;
  }
  // columns 16 c .. 16 c + 15 of row `lane` from the row-conversion buffer
  __device__ __forceinline__ void take_chunk(const double* buf, const int c) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const d2 x = *reinterpret_cast<const d2*>(buf + lane * kRowPitch + 2 * k);
      fr_[16 * c + 2 * k] = x[0];
      fr_[16 * c + 2 * k + 1] = x[1];
    }
    if ((lane >> 4) == c) fd_ = buf[lane * kRowPitch + (lane & 15)];
  }
  // MfmaBackend::tiles_to_row for the held inverse, every sixteen-column chunk also offered to wave B
  __device__ __forceinline__ void tiles_to_rows_fork() {
    const int g = lane >> 4, j = lane & 15;
    double* buf = w.part;  // [64][kRowPitch]
    const bool share = !fork_off;
    if (share) send(CMD_ROWS);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
      for (int I = c; I < 4; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) buf[(16 * I + 4 * r + g) * kRowPitch + j] = acc[tix(I, c)][r];
#pragma unroll
      for (int J = 0; J < c; ++J)
#pragma unroll
        for (int r = 0; r < 4; ++r) buf[(16 * J + j) * kRowPitch + 4 * r + g] = acc[tix(c, J)][r];
      ++rseq;
      if (share) set_flag(3, rseq);
      else wave_sync();
      take_chunk(buf, c);
      if (share) wait_flag(4, rseq);  // wave B has its copy: the buffer may be rewritten
      else wave_sync();
    }
    if (share) wait_flag(2, mseq);
  }
  __device__ __forceinline__ bool construct(double x, bool need_inverse, double rhs, double* u) {
#pragma unroll
    for (int k = 0; k < 64; ++k) fr_[k] = 0.0;
    fd_ = 0.0;
    if constexpr (RMETRIC == MM_RMETRIC_USER) {  // (a refinement solve's tiles of M(x) are dead between solves)
#pragma unroll
      for (int t = 0; t < kTiles; ++t) this->mx_[t] = d4{0.0, 0.0, 0.0, 0.0};
    }
    bool ok = Base::build(x);
    if (need_inverse) {  // wave-uniform
      ok = this->template sweep<false>() && ok;
      tiles_to_rows_fork();
    } else {
      ok = this->template sweep<true>() && ok;
      *u = this->solve_factored(rhs);
    }
    return ok;
  }
  __device__ __forceinline__ bool build_and_invert(double x) {
    double dummy;
    return construct(x, true, 0.0, &dummy);
  }
  __device__ __forceinline__ bool build_and_solve(double x, double rhs, double* u) { return construct(x, false, rhs, u); }

  // ---- wave B ----------------------------------------------------------------------------------------------------------
  __device__ __forceinline__ void recv_rows() {
    const double* buf = rowbuf;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      ++rseq;
      wait_flag(3, rseq);
      take_chunk(buf, c);
      set_flag(4, rseq);
    }
  }
  // the reversibility-check solve from the state wave A parked in its slots (implicit_leapfrog_chain's MODE_CHK evaluations)
  __device__ __forceinline__ void run_chk(const mm_fp_opts& o) {
    const int iter = __builtin_amdgcn_readfirstlane((int)mb[MB_ITER]);
    const int stage = __builtin_amdgcn_readfirstlane((int)mb[MB_STAGE]);
    const double t = mb[MB_T];
    double x = mstash[SL_XQ * 64 + lane], sx0 = mstash[SL_SX0 * 64 + lane], sx1 = mstash[SL_SX1 * 64 + lane];
    double uc = mstash[SL_UC * 64 + lane];
    const double pw = mstash[SL_PW * 64 + lane], qw = mstash[SL_QW * 64 + lane];
    FpCtl c{iter, stage};
    ChainResult rr{MM_ST_OK, 0, 0, 0, 0, 0, 0, 0, 0};
    int outcome = FK_FALLBACK, status = MM_ST_OK, n_evals = 0;
#pragma unroll 1
    for (;;) {
      double u;
      if (!refine_solve(*this, x, pw, uc, &u, rr)) break;  // FK_FALLBACK: wave A factorises at x
      uc = u;
      ++n_evals;
      double pt = x;
      int st = MM_ST_OK;
      const int a = fp_feed(*this, c, sx0, sx1, qw - t * u, o, &pt, &st);
      if (a == FP_FAIL) {
        outcome = FK_FAIL;
        status = st;
        break;
      }
      x = pt;
      if (a == FP_DONE) {
        outcome = FK_DONE;
        break;
      }
    }
    mstash[SL_XQ * 64 + lane] = x;
    mstash[SL_SX0 * 64 + lane] = sx0;
    mstash[SL_SX1 * 64 + lane] = sx1;
    mstash[SL_UC * 64 + lane] = uc;
    if (lane == 0) {
      mb[MB_OUTCOME] = (double)outcome;
      mb[MB_STATUS] = (double)status;
      mb[MB_RITER] = (double)c.iter;
      mb[MB_RSTAGE] = (double)c.stage;
      mb[MB_NEVALS] = (double)n_evals;
      mb[MB_NPAIRS] = (double)rr.n_refine;
    }
  }
  __device__ __forceinline__ void helper_loop(const mm_fp_opts& o) {
#pragma unroll 1
    for (;;) {
      ++mseq;
      wait_flag(0, mseq);
      const int code = __builtin_amdgcn_readfirstlane(flags()[1]);
      if (code == CMD_EXIT) break;
      if (code == CMD_ROWS) recv_rows();
      else run_chk(o);
      set_flag(2, mseq);
    }
  }
};

template <int RMETRIC>
__device__ __forceinline__ void implicit_fork_body(const ImplicitArgs& A, double* lds) {
  double* base_lds = lds;
  const int base_elems = (RMETRIC == MM_RMETRIC_RANK1) ? kBaseDoubles : 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int dim = A.dim;
  if constexpr (RMETRIC == MM_RMETRIC_RANK1) {
    for (int idx = threadIdx.x; idx < kBaseDoubles; idx += blockDim.x) {
      const int row = idx / kBasePitch, col = idx - row * kBasePitch;
      base_lds[idx] = (row < dim && col < dim) ? A.rparams[(int64_t)row * dim + col] : 0.0;
    }
  }
  const int cslot = wave & (kChains - 1), role = wave >> 2;
  constexpr int kUser = RMETRIC == MM_RMETRIC_USER ? mmuser::lds_doubles(64) : 0;
  constexpr int kUA = (mmuser::kAux + 1) & ~1;
  double* wa = lds + base_elems + cslot * (kMfmaWaveDoubles + kUser);
  double* wb = lds + base_elems + kChains * (kMfmaWaveDoubles + kUser) + cslot * (kHelperDoubles + kUser);
  if (role == 1 && lane < 16) reinterpret_cast<int*>(wb + 192)[lane] = 0;  // the mailbox flags
  __syncthreads();
  const int64_t chain = (int64_t)blockIdx.x * kChains + cslot;
  if (chain >= A.n_chains) return;  // (both waves of the slot; no block-level barrier below this point)
  ForkBackend<RMETRIC> bk;
  bk.dim = dim;
  bk.inv_dim_ = 1.0 / (double)dim;
  bk.lane = lane;
  bk.target = A.target;
  bk.refine_on = A.no_refine == 0;
  bk.dual_off = true;
  bk.fork_off = A.no_dual != 0;
  bk.base_lds = base_lds;
  bk.tparams = A.tparams;
  bk.uparams = A.rparams;
  bk.work = nullptr;
  bk.mseq = 0;
  bk.rseq = 0;
  bk.mb = wb + 192;
  bk.mstash = wa + 704 + 64 * kRowPitch + 192;
  bk.rowbuf = wa + 704;
  if (role == 1) {
    bk.w.qt = wb;
    bk.w.nat = wb + 64;
    bk.w.aux = wb + 128;
    bk.w.wt = nullptr;  // (the sweep's buffers and the slots: wave A only)
    bk.w.vperm = nullptr;
    bk.w.part = nullptr;
    bk.w.mpart = nullptr;
    bk.w.stash = nullptr;
    bk.w.prof = nullptr;
    if constexpr (RMETRIC == MM_RMETRIC_USER) {  // the products' point and its aux block (the held inverse's: wave A only)
      double* up = wb + kHelperDoubles;
      bk.w.uq = up;
      bk.w.ux = up + 64;
      bk.w.uaq = up + 128;
      bk.w.uax = up + 128 + kUA;
    }
    if (!bk.fork_off) bk.helper_loop(A.opts);
    return;
  }
  bk.w.qt = wa;
  bk.w.wt = wa + 256;
  bk.w.nat = wa + 512;
  bk.w.vperm = wa + 576;
  bk.w.aux = wa + 640;
  bk.w.part = wa + 704;
  bk.w.mpart = bk.w.part + 64 * kRowPitch;
  bk.w.stash = bk.w.mpart + 192;
  bk.w.prof = bk.w.stash + SL_COUNT_REFINE * 64;
  if constexpr (RMETRIC == MM_RMETRIC_USER) {
    double* up = wa + kMfmaWaveDoubles;
    bk.w.uq = up;
    bk.w.ux = up + 64;
    bk.w.uaq = up + 128;
    bk.w.uax = up + 128 + kUA;
    bk.work = A.work ? A.work + chain * (int64_t)(64 * 64) : nullptr;
  }
  const bool act = lane < dim;
  double q = act ? A.pos[chain * dim + lane] : 0.0;
  double p = act ? A.mom[chain * dim + lane] : 0.0;
  const double t = signed_step(A.dir, A.step_scale, chain, A.step_size);
  bk.slot(SL_Q) = q;
  bk.slot(SL_P) = p;
  const ChainResult r = implicit_leapfrog_chain(bk, t, mmdev::chain_steps(A.chain_steps, chain, A.n_steps), A.opts);
  if (!bk.fork_off) bk.send(CMD_EXIT);
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
}

#ifndef MM_RTC_BUILD
template <int RMETRIC>
__global__ __launch_bounds__(128 * kChains) void implicit_fork_kernel(ImplicitArgs A) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  implicit_fork_body<RMETRIC>(A, lds);
}
#endif

}  // namespace mmfork
