// Host side of the two-waves-per-chain matrix-core dense-Riemannian kernel (32 < D <= 64; device code: implicit_pair.h).
// Its own translation unit: the kernel is several thousand instructions per instantiation.
#include "implicit_pair.h"

using namespace mmimp;

int mm_launch_implicit_pair(mm_ctx* ctx, const mm_model* m, mm_state* s, const ImplicitArgs& a) {
  const unsigned blocks = (unsigned)s->n;  // one workgroup of two waves per chain
  if (m->rmetric == MM_RMETRIC_RANK1) {
    const size_t lds = mmpair::pair_chain_doubles<MM_RMETRIC_RANK1>() * sizeof(double);
    MM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(mmpair::implicit_pair_kernel<MM_RMETRIC_RANK1>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((mmpair::implicit_pair_kernel<MM_RMETRIC_RANK1>), dim3(blocks), dim3(128), lds, ctx->stream, a);
  } else {
#ifndef MM_PAIR_RANK1_ONLY
    const size_t lds = mmpair::pair_chain_doubles<MM_RMETRIC_DIAGQUAD>() * sizeof(double);
    MM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(mmpair::implicit_pair_kernel<MM_RMETRIC_DIAGQUAD>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((mmpair::implicit_pair_kernel<MM_RMETRIC_DIAGQUAD>), dim3(blocks), dim3(128), lds, ctx->stream, a);
#endif
  }
  MM_HIP_CHECK(ctx, hipGetLastError());
  return MM_OK;
}

#ifdef MM_DEV_KERNELS
// developer hook (tools/ubench_pair.py): mm_implicit_leapfrog on this kernel with the phase clocks on; out is a HOST buffer
// of N * 8 doubles: cycles of chain i (wave 0's clock) spent in the phases PH_* of implicit_core.h
extern "C" __attribute__((visibility("default"))) int mm_debug_pair_step_profile(mm_ctx* ctx, const mm_model* m, mm_state* s, double h,
                                                                                 int n_steps, const mm_fp_opts* opts, double* out) {
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
  const size_t bytes = (size_t)s->n * PH_COUNT * sizeof(double);
  double* d_out = nullptr;
  MM_HIP_CHECK(ctx, hipMalloc(&d_out, bytes));
  a.out = d_out;
  const size_t lds = mmpair::pair_chain_doubles<MM_RMETRIC_RANK1>() * sizeof(double);
  MM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(mmpair::implicit_pair_kernel<MM_RMETRIC_RANK1, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((mmpair::implicit_pair_kernel<MM_RMETRIC_RANK1, true>), dim3((unsigned)s->n), dim3(128), lds, ctx->stream, a);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, bytes, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(d_out);
  if (e != hipSuccess) {
    mm_set_error(ctx, std::string("mm_debug_pair_step_profile: ") + hipGetErrorString(e));
    return MM_ERR_HIP;
  }
  return MM_OK;
}
#endif  // MM_DEV_KERNELS
