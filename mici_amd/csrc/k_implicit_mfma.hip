// Host side of the matrix-core wave-per-chain dense-Riemannian kernel (32 < D <= 64; device code: implicit_mfma.h) and its
// developer kernels.
#include "implicit_mfma.h"

namespace {

using namespace mmdev;
using namespace mmimp;
using namespace mmmfma;

#ifdef MM_DEV_KERNELS
// ---- developer profile (not part of the ABI header; tools/ubench_primitives.py): shader-clock cycles of
// the backend's primitives, measured in-kernel with s_memtime on every wave, chain 0 reported
__global__ __launch_bounds__(64 * kWaves) void mfma_profile_kernel(ImplicitArgs A, int repeats, double* out) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* base_lds = lds;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int dim = A.dim;
  for (int idx = threadIdx.x; idx < kBaseDoubles; idx += blockDim.x) {
    const int row = idx / kBasePitch, col = idx - row * kBasePitch;
    base_lds[idx] = (row < dim && col < dim) ? A.rparams[(int64_t)row * dim + col] : 0.0;
  }
  __syncthreads();
  const int64_t chain = (int64_t)blockIdx.x * kWaves + wave;
  if (chain >= A.n_chains) return;
  double* wl = lds + kBaseDoubles + wave * kMfmaWaveDoubles;
  MfmaBackend<MM_RMETRIC_RANK1> bk;
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
  bk.base_lds = base_lds;
  bk.tparams = A.tparams;
  double q = lane < dim ? A.pos[chain * dim + lane] : 0.0;
  double p = lane < dim ? A.mom[chain * dim + lane] : 0.0;
  long long c[6] = {0, 0, 0, 0, 0, 0};
  double sink = 0.0;
  for (int r = 0; r < repeats; ++r) {
    const long long t0 = __builtin_readcyclecounter();
    bool ok = bk.build(q);
    const long long t1 = __builtin_readcyclecounter();
    ok = bk.template sweep<false>() && ok;
    bk.tiles_to_rows();  // (timed with the sweep: the inverse is applied in row form)
    const long long t2 = __builtin_readcyclecounter();
    const double u = bk.matvec(p);
    const long long t3 = __builtin_readcyclecounter();
    const double gq = bk.grad(q);
    const long long t4 = __builtin_readcyclecounter();
    const double nn = bk.norm(u, MM_NORM_LINF);
    const long long t5 = __builtin_readcyclecounter();
    bk.metric_point(q);
    const double mv = bk.metric_apply(p);
    const long long t6 = __builtin_readcyclecounter();
    c[5] += t6 - t5;
    sink += mv;
    c[0] += t1 - t0;
    c[1] += t2 - t1;
    c[2] += t3 - t2;
    c[3] += t4 - t3;
    c[4] += t5 - t4;
    sink += u + gq + nn + (ok ? 0.0 : 1.0);
    q += 1e-12 * sink;
  }
  if (chain == 0 && lane == 0)
    for (int i = 0; i < 6; ++i) out[i] = (double)c[i] / repeats;
  if (lane == 0) out[8 + chain] = sink;
}

#endif  // MM_DEV_KERNELS

}  // namespace

// k_implicit_fork.hip
int mm_launch_implicit_fork(mm_ctx* ctx, const mm_model* m, mm_state* s, const mmimp::ImplicitArgs& a);
// k_implicit_pair.hip
int mm_launch_implicit_pair(mm_ctx* ctx, const mm_model* m, mm_state* s, const mmimp::ImplicitArgs& a);

int mm_launch_implicit_mfma(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int n_steps,
                            const mm_fp_opts& opts, mm_counters* d_counters) {
  if (m->dim > 64) {
    mm_set_error(ctx, "matrix-core dense-Riemannian kernel supports dim <= 64");
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
  a.rparams = m->d_rmetric_params;
  a.step_size = h;
  a.n_steps = n_steps;
  a.opts = opts;
  a.no_refine = mm_refine_disabled();
  a.no_dual = mm_dual_disabled();
  a.lowrank_refresh = mm_lowrank_refresh();
  a.counters = d_counters;
  const bool r1 = m->rmetric == MM_RMETRIC_RANK1;
  // round 6: MICI_AMD_PAIR=1 selects the two-waves-per-chain kernel (implicit_pair.h: built, parity-green, and measured -
  // it loses 21 % on c3, profiles/r06_ab_c3_pair.txt - so the one-wave kernel stays the default)
  // round 6: the rank-one-update metric's solve-only constructions by the Woodbury identity from the held inverse
  // (implicit_core.h lowrank_solve; one-wave kernel); MICI_AMD_LOWRANK=0: the CG refinement (forked kernel by default)
  if (r1 && !mm_lowrank_disabled() && a.no_refine == 0) {
    const unsigned blocks = (unsigned)((s->n + kWaves - 1) / kWaves);
    const size_t lds = (kBaseDoubles + kWaves * kMfmaWaveDoubles) * sizeof(double);
    MM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(implicit_mfma_kernel<MM_RMETRIC_RANK1, false, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((implicit_mfma_kernel<MM_RMETRIC_RANK1, false, true>), dim3(blocks), dim3(64 * kWaves), lds,
                       ctx->stream, a);
    MM_HIP_CHECK(ctx, hipGetLastError());
    return MM_OK;
  }
  static const bool pair_on = [] { const char* e = getenv("MICI_AMD_PAIR"); return e && e[0] == '1'; }();
  if (pair_on) return mm_launch_implicit_pair(ctx, m, s, a);
  // round 6: the forked kernel (implicit_fork.h: a second wave per chain runs the reversibility-check solve while the first runs
  // the C-adjoint solve - c3 1.36e7 -> 1.48e7 steps/s, profiles/r06_ab_c3_fork.txt) is the default; MICI_AMD_FORK=0 selects the
  // one-wave kernel
  static const bool fork_on = [] { const char* e = getenv("MICI_AMD_FORK"); return !(e && e[0] == '0'); }();
  if (fork_on) return mm_launch_implicit_fork(ctx, m, s, a);
  const unsigned blocks = (unsigned)((s->n + kWaves - 1) / kWaves);
  const size_t lds = ((r1 ? kBaseDoubles : 0) + kWaves * kMfmaWaveDoubles) * sizeof(double);
  if (r1) {
    MM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(implicit_mfma_kernel<MM_RMETRIC_RANK1>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((implicit_mfma_kernel<MM_RMETRIC_RANK1>), dim3(blocks), dim3(64 * kWaves), lds,
                       ctx->stream, a);
  } else {
    MM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(implicit_mfma_kernel<MM_RMETRIC_DIAGQUAD>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((implicit_mfma_kernel<MM_RMETRIC_DIAGQUAD>), dim3(blocks), dim3(64 * kWaves), lds,
                       ctx->stream, a);
  }
  MM_HIP_CHECK(ctx, hipGetLastError());
  return MM_OK;
}

#ifdef MM_DEV_KERNELS
// developer hook: cycles of {build, sweep, mat-vec, grad, norm, M(x) v} into out[0..5]
extern "C" __attribute__((visibility("default"))) int mm_debug_mfma_profile(mm_ctx* ctx, const mm_model* m, mm_state* s, int repeats, double* out) {
  if (!ctx || !m || !s || m->dim > 64 || m->rmetric != MM_RMETRIC_RANK1) return MM_ERR_INVALID;
  ImplicitArgs a{};
  a.pos = s->d_pos;
  a.mom = s->d_mom;
  a.n_chains = s->n;
  a.dim = s->dim;
  a.target = m->target;
  a.tparams = m->d_target_params;
  a.rparams = m->d_rmetric_params;
  double* d_out = nullptr;
  MM_HIP_CHECK(ctx, hipMalloc(&d_out, (8 + s->n) * sizeof(double)));
  const unsigned blocks = (unsigned)((s->n + kWaves - 1) / kWaves);
  const size_t lds = (kBaseDoubles + kWaves * kMfmaWaveDoubles) * sizeof(double);
  MM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(mfma_profile_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(mfma_profile_kernel, dim3(blocks), dim3(64 * kWaves), lds, ctx->stream, a, repeats, d_out);
  MM_HIP_CHECK(ctx, hipMemcpyAsync(out, d_out, 6 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  MM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  (void)hipFree(d_out);
  return MM_OK;
}

// developer hook (tools/ubench_primitives.py): mm_implicit_leapfrog on this kernel with the phase clocks on; out is a
// HOST buffer of N * 8 doubles: cycles of chain i spent in the phases PH_* of implicit_core.h
extern "C" __attribute__((visibility("default"))) int mm_debug_mfma_step_profile(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int n_steps,
                                          const mm_fp_opts* opts, double* out) {
  if (!ctx || !m || !s || !opts || !out || m->dim > 64 || m->rmetric != MM_RMETRIC_RANK1) return MM_ERR_INVALID;
  MM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
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
  a.rparams = m->d_rmetric_params;
  a.step_size = h;
  a.n_steps = n_steps;
  a.opts = *opts;
  a.lowrank_refresh = mm_lowrank_refresh();
  const size_t bytes = (size_t)s->n * PH_COUNT * sizeof(double);
  double* d_out = nullptr;
  MM_HIP_CHECK(ctx, hipMalloc(&d_out, bytes));
  a.out = d_out;
  const unsigned blocks = (unsigned)((s->n + kWaves - 1) / kWaves);
  const size_t lds = (kBaseDoubles + kWaves * kMfmaWaveDoubles) * sizeof(double);
  if (mm_lowrank_disabled()) {
    MM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(implicit_mfma_kernel<MM_RMETRIC_RANK1, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((implicit_mfma_kernel<MM_RMETRIC_RANK1, true>), dim3(blocks), dim3(64 * kWaves), lds, ctx->stream,
                       a);
  } else {
    MM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(implicit_mfma_kernel<MM_RMETRIC_RANK1, true, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((implicit_mfma_kernel<MM_RMETRIC_RANK1, true, true>), dim3(blocks), dim3(64 * kWaves), lds,
                       ctx->stream, a);
  }
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, bytes, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(d_out);
  if (e != hipSuccess) {
    mm_set_error(ctx, std::string("mm_debug_mfma_step_profile: ") + hipGetErrorString(e));
    return MM_ERR_HIP;
  }
  return MM_OK;
}
#endif  // MM_DEV_KERNELS
