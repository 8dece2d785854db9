// Host side of the forked matrix-core wave kernel (32 < D <= 64, built-in metrics; device code: implicit_fork.h).
#include "implicit_fork.h"

using namespace mmimp;

int mm_launch_implicit_fork(mm_ctx* ctx, const mm_model* m, mm_state* s, const ImplicitArgs& a) {
  const unsigned blocks = (unsigned)((s->n + mmfork::kChains - 1) / mmfork::kChains);
  if (m->rmetric == MM_RMETRIC_RANK1) {
    const size_t lds = mmfork::fork_lds_doubles<MM_RMETRIC_RANK1>() * sizeof(double);
    MM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(mmfork::implicit_fork_kernel<MM_RMETRIC_RANK1>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((mmfork::implicit_fork_kernel<MM_RMETRIC_RANK1>), dim3(blocks), dim3(128 * mmfork::kChains), lds,
                       ctx->stream, a);
  } else {
    const size_t lds = mmfork::fork_lds_doubles<MM_RMETRIC_DIAGQUAD>() * sizeof(double);
    MM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(mmfork::implicit_fork_kernel<MM_RMETRIC_DIAGQUAD>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((mmfork::implicit_fork_kernel<MM_RMETRIC_DIAGQUAD>), dim3(blocks), dim3(128 * mmfork::kChains), lds,
                       ctx->stream, a);
  }
  MM_HIP_CHECK(ctx, hipGetLastError());
  return MM_OK;
}
