// Host side of the block-16 matrix-core team kernel (75 < D <= 256; device code: implicit_blk16.h) and its developer /
// test kernels.
#include "implicit_blk16.h"

namespace {

using namespace mmdev;
using namespace mmimp;
using namespace mmblk16;

#ifdef MM_DEV_KERNELS
// Developer / test hook: the linear algebra of the backend on its own.  Per chain, with x = pos and b = mom:
//   op 0: out[chain][256][256] = the explicit inverse M(x)^-1 (dense, symmetric) from the full sweep
//   op 1: out[chain][256]      = M(x)^-1 b by the trailing sweep + substitution
//   op 2: out[chain][256]      = M(x)^-1 b by the full sweep + mat-vec
//   op 3..7 (timing only, tools/ubench_blk16.py): 3 = build, 4 = build + full sweep, 5 = build + trailing sweep,
//        6 = mat-vec, 7 = substitution - each repeated `reps` times
// status[chain] = 0 if the metric was found positive definite and finite, else 5.
template <int RMETRIC>
__global__ __launch_bounds__(NTHR, 2) void blk16_debug_kernel(ImplicitArgs A, int op, int reps) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  TeamBlk16<RMETRIC> bk;
  init_backend(bk, A, lds);
  const int64_t chain = blockIdx.x;
  const int tid = threadIdx.x, dim = A.dim;
  const bool act = tid < dim;
  const double q = act ? A.pos[chain * dim + tid] : 0.0;
  const double p = act ? A.mom[chain * dim + tid] : 0.0;
  double u = 0.0;
  const bool ok = bk.construct(q, op != 1, p, &u);
  if (op == 0) {
    double* out = A.out + chain * (int64_t)(DPM * DPM);
    const int ln = fresh_lane(), g = ln >> 4, j = ln & 15;
#pragma unroll
    for (int s = 0; s < NSLOT; ++s) {
      const int I = bk.tile_i(s, bk.wave), J = bk.tile_j(s, bk.wave);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double v = -bk.acc[s][r];
        out[(int64_t)(16 * I + 4 * r + g) * DPM + 16 * J + j] = v;
        out[(int64_t)(16 * J + j) * DPM + 16 * I + 4 * r + g] = v;
      }
    }
  } else {
    if (op == 2) u = bk.matvec(p);
    if (tid < DPM) A.out[chain * (int64_t)DPM + tid] = u;
  }
  if (tid == 0) A.status[chain] = ok ? 0 : MM_ST_LINALG;
}

// Timing kernels (tools/ubench_blk16.py), one lean instantiation per OP, rank-one metric only:
//   3 = build, 4 = build + full sweep, 5 = build + trailing sweep, 6 = mat-vec, 7 = substitution   (x reps)
//   8 / 9 = per-phase cycle counts of one full / trailing sweep (summed over its blocks), per wave -> out[chain][64];
//   10..12 = as 8 with a phase removed (sweep<>'s EXPER: results are wrong, only the clock is read)
template <int OP>
__global__ __launch_bounds__(NTHR, 2) void blk16_bench_kernel(ImplicitArgs A, int reps) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  TeamBlk16<MM_RMETRIC_RANK1> bk;
  init_backend(bk, A, lds);
  const int64_t chain = blockIdx.x;
  const int tid = threadIdx.x, dim = A.dim;
  const bool act = tid < dim;
  const double q = act ? A.pos[chain * dim + tid] : 0.0;
  const double p = act ? A.mom[chain * dim + tid] : 0.0;
  double u = 0.0;
  bool ok = true;
  if constexpr (OP == 15 || OP == 16) {  // 15 = metric_apply (M(x) v matrix-free), 16 = sum2 (two team sums)
    bk.metric_point(q);
    for (int rep = 0; rep < reps; ++rep) {
      if constexpr (OP == 15) u = bk.metric_apply(p + u * 1e-300);
      else {
        double sa, sb;
        bk.sum2(p + u * 1e-300, q, &sa, &sb);
        u = sa + sb;
      }
    }
  } else if constexpr (OP <= 7) {
    if constexpr (OP == 6) ok = bk.construct(q, true, p, &u);
    if constexpr (OP == 7) ok = bk.construct(q, false, p, &u);
    for (int rep = 0; rep < reps; ++rep) {
      if constexpr (OP == 3) ok = !bk.build(q + u * 1e-300) && ok;
      if constexpr (OP == 4) ok = bk.template sweep<false>(bk.build(q + u * 1e-300)) && ok;
      if constexpr (OP == 5) ok = bk.template sweep<true>(bk.build(q + u * 1e-300)) && ok;
      if constexpr (OP == 6) u = bk.matvec(p + u * 1e-300);
      if constexpr (OP == 7) u = bk.solve() + u * 1e-300;
    }
    if constexpr (OP <= 5) u = bk.acc[0][0] + bk.acc[NSLOT - 1][1];
  } else {
    const bool bad = bk.build(q);
    const long long t0 = __builtin_readcyclecounter();
    if constexpr (OP == 8) ok = bk.template sweep<false, true>(bad);
    if constexpr (OP == 9) ok = bk.template sweep<true, true>(bad);
    if constexpr (OP == 10) ok = bk.template sweep<false, true, 1>(bad);
    if constexpr (OP == 11) ok = bk.template sweep<false, true, 2>(bad);
    if constexpr (OP == 12) ok = bk.template sweep<false, true, 3>(bad);
    if constexpr (OP == 14) ok = bk.template sweep<false, true, 4>(bad);
    const long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    if (tid < 64) {
      const long long* src = reinterpret_cast<const long long*>(lds + kOffPart);
      u = (tid & 7) == 6 ? (double)(t1 - t0) : (tid & 7) == 7 ? 0.0 : (double)src[tid];
    }
  }
  if (tid < DPM) A.out[chain * (int64_t)DPM + tid] = u;
  if (tid == 0) A.status[chain] = ok ? 0 : MM_ST_LINALG;
}

// out[lane] = 1 if the permlane-swap row sums differ from the ds_bpermute ones for this lane (test hook)
__global__ void blk16_permlane_check_kernel(double* out) {
  const int l = threadIdx.x;
  const double x = 1.0 + 0.37 * l + 1e-3 * l * l;  // distinct per lane, exactly representable sums are not needed
  const double a16 = swap_sum16(x), b16 = x + __shfl_xor(x, 16);
  const double a32 = swap_sum32(x), b32 = x + __shfl_xor(x, 32);
  double y = x + __shfl_xor(x, 16);
  y += __shfl_xor(y, 32);
  const double z = swap_sum32(swap_sum16(x));
  out[l] = (a16 != b16 ? 1.0 : 0.0) + (a32 != b32 ? 2.0 : 0.0) + (y != z ? 4.0 : 0.0);
}

#endif  // MM_DEV_KERNELS

template <class K, class... Extra>
int launch_blk16(mm_ctx* ctx, K kernel, const ImplicitArgs& a, Extra... extra) {
  const size_t lds = kLdsDoubles * sizeof(double);
  MM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kernel, dim3((unsigned)a.n_chains), dim3(NTHR), lds, ctx->stream, a, extra...);
  MM_HIP_CHECK(ctx, hipGetLastError());
  return MM_OK;
}

int fill_args(mm_ctx* ctx, const mm_model* m, mm_state* s, ImplicitArgs& a) {
  if (m->dim > DPM) {
    mm_set_error(ctx, "block-16 matrix-core team kernel supports dim <= 256");
    return MM_ERR_UNSUPPORTED;
  }
  if (m->rmetric == MM_RMETRIC_RANK1 && m->d_rmetric_tiled == nullptr) {
    mm_set_error(ctx, "internal: rank-one base matrix was not tiled for the block-16 kernel");
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
  a.rparams = m->d_rmetric_tiled;
  return MM_OK;
}

}  // namespace

int mm_launch_implicit_blk16(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int n_steps,
                             const mm_fp_opts& opts, mm_counters* d_counters) {
  ImplicitArgs a{};
  const int rc = fill_args(ctx, m, s, a);
  if (rc != MM_OK) return rc;
  a.step_size = h;
  a.n_steps = n_steps;
  a.opts = opts;
  a.no_refine = mm_refine_disabled();
  a.no_dual = mm_dual_disabled();
  a.lowrank_refresh = mm_lowrank_refresh();
  a.counters = d_counters;
  if (m->rmetric == MM_RMETRIC_RANK1) {
    // round 6: solve-only constructions by the Woodbury identity from the held inverse (implicit_core.h lowrank_solve);
    // MICI_AMD_LOWRANK=0: the CG refinement
    if (!mm_lowrank_disabled() && a.no_refine == 0)
      return launch_blk16(ctx, implicit_blk16_kernel<MM_RMETRIC_RANK1, false, true>, a);
    return launch_blk16(ctx, implicit_blk16_kernel<MM_RMETRIC_RANK1>, a);
  }
  return launch_blk16(ctx, implicit_blk16_kernel<MM_RMETRIC_DIAGQUAD>, a);
}

#ifdef MM_DEV_KERNELS
// developer / test hook (tests/test_gpu_blk16.py): see blk16_debug_kernel.  out is a HOST buffer of
// N * 256 * 256 (op 0) or N * 256 (op 1, 2) doubles; status[N] (host, may be NULL) receives 0 / 5 per chain.
extern "C" __attribute__((visibility("default"))) int mm_debug_blk16_linalg(mm_ctx* ctx, const mm_model* m, mm_state* s, int op, double* out,
                                     int32_t* status, int reps, double* ms) {
  if (!ctx || !m || !s || !out || op < 0 || op > 16 || m->rmetric == MM_RMETRIC_NONE ||
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
  if (op == 13) {
    hipLaunchKernelGGL(blk16_permlane_check_kernel, dim3(1), dim3(64), 0, ctx->stream, d_out);
    lrc = hipGetLastError() == hipSuccess ? MM_OK : MM_ERR_HIP;
  } else if (op >= 3) {
    if (m->rmetric != MM_RMETRIC_RANK1) {
      (void)hipFree(d_out);
      return MM_ERR_INVALID;
    }
    switch (op) {
#define MM_BENCH(OP) case OP: lrc = launch_blk16(ctx, blk16_bench_kernel<OP>, a, reps); break;
      MM_BENCH(3) MM_BENCH(4) MM_BENCH(5) MM_BENCH(6) MM_BENCH(7) MM_BENCH(8) MM_BENCH(9) MM_BENCH(10) MM_BENCH(11)
      MM_BENCH(12) MM_BENCH(14) MM_BENCH(15) MM_BENCH(16)
#undef MM_BENCH
      default: lrc = MM_ERR_INVALID; break;
    }
  } else if (m->rmetric == MM_RMETRIC_RANK1)
    lrc = launch_blk16(ctx, blk16_debug_kernel<MM_RMETRIC_RANK1>, a, op, reps);
  else
    lrc = launch_blk16(ctx, blk16_debug_kernel<MM_RMETRIC_DIAGQUAD>, a, op, reps);
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
      mm_set_error(ctx, std::string("mm_debug_blk16_linalg: ") + hipGetErrorString(e));
      lrc = MM_ERR_HIP;
    }
  }
  (void)hipFree(d_out);
  return lrc;
}

// developer hook (tools/ubench_blk16.py): mm_implicit_leapfrog on the block-16 kernel with the phase clocks on;
// out is a HOST buffer of N * 8 doubles: cycles of chain i spent in the phases PH_* of implicit_core.h
extern "C" __attribute__((visibility("default"))) int mm_debug_blk16_step_profile(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int n_steps,
                                           const mm_fp_opts* opts, double* out) {
  if (!ctx || !m || !s || !opts || !out || m->rmetric != MM_RMETRIC_RANK1) return MM_ERR_INVALID;
  MM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  ImplicitArgs a{};
  const int rc = fill_args(ctx, m, s, a);
  if (rc != MM_OK) return rc;
  a.step_size = h;
  a.n_steps = n_steps;
  a.opts = *opts;
  a.lowrank_refresh = mm_lowrank_refresh();
  const size_t bytes = (size_t)s->n * PH_COUNT * sizeof(double);
  double* d_out = nullptr;
  MM_HIP_CHECK(ctx, hipMalloc(&d_out, bytes));
  a.out = d_out;
  int lrc = mm_lowrank_disabled() ? launch_blk16(ctx, implicit_blk16_kernel<MM_RMETRIC_RANK1, true>, a)
                                  : launch_blk16(ctx, implicit_blk16_kernel<MM_RMETRIC_RANK1, true, true>, a);
  if (lrc == MM_OK) {
    hipError_t e = hipMemcpyAsync(out, d_out, bytes, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
      mm_set_error(ctx, std::string("mm_debug_blk16_step_profile: ") + hipGetErrorString(e));
      lrc = MM_ERR_HIP;
    }
  }
  (void)hipFree(d_out);
  return lrc;
}
#endif  // MM_DEV_KERNELS
