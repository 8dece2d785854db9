// Host side of the global-memory tier of the dense-Riemannian kernels, 279 < D <= 1024 (device code: implicit_global.h).
#include "implicit_global.h"

namespace {

using namespace mmimp;
using namespace mmglob;

int fill_args(mm_ctx* ctx, const mm_model* m, mm_state* s, ImplicitArgs& a) {
  if (m->dim > DPMAX) {
    mm_set_error(ctx, "dense-Riemannian kernels support dim <= 1024 (one flat vector element per thread of a workgroup)");
    return MM_ERR_UNSUPPORTED;
  }
  if (m->rmetric != MM_RMETRIC_RANK1 && m->rmetric != MM_RMETRIC_DIAGQUAD) {
    mm_set_error(ctx, "internal: this entry point runs the built-in metrics of the global-memory tier (a user metric beyond dim 279 runs its own run-time compiled kernels, mm_rtc.hip MM_RTC_FAM_GLOBAL)");
    return MM_ERR_UNSUPPORTED;
  }
  a = ImplicitArgs{};
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
  a.rparams = m->d_rmetric_params;  // rank-one metric: the base matrix [dim][dim], read as it is
  const size_t dp = (size_t)padded_dim(m->dim);
  // the chains' matrices, in HBM: FP64, and the FP32 copy of the held inverse behind them (implicit_global.h precond)
  const int rc = mm_state_ensure_work(ctx, s, (size_t)s->n * dp * dp * (sizeof(double) + sizeof(float)));
  if (rc != MM_OK) return rc;
  a.work = static_cast<double*>(s->d_work);
  return MM_OK;
}

template <class K>
int launch(mm_ctx* ctx, K kernel, const ImplicitArgs& a) {
  if (a.n_chains == 0) return MM_OK;
  const size_t lds = kLdsDoubles * sizeof(double);
  MM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds));
  hipLaunchKernelGGL(kernel, dim3((unsigned)a.n_chains), dim3(NT), lds, ctx->stream, a);
  MM_HIP_CHECK(ctx, hipGetLastError());
  return MM_OK;
}

}  // namespace

// NB = 32 pivots per pass while the panel fits the LDS (padded dimension <= 512), 16 beyond
#define MM_GLOB_DISPATCH(KERNEL, ...)                                                              \
  (padded_dim(m->dim) <= 512                                                                       \
       ? (m->rmetric == MM_RMETRIC_RANK1 ? launch(ctx, KERNEL<MM_RMETRIC_RANK1, 32, __VA_ARGS__>, a)    \
                                         : launch(ctx, KERNEL<MM_RMETRIC_DIAGQUAD, 32, __VA_ARGS__>, a)) \
       : (m->rmetric == MM_RMETRIC_RANK1 ? launch(ctx, KERNEL<MM_RMETRIC_RANK1, 16, __VA_ARGS__>, a)     \
                                         : launch(ctx, KERNEL<MM_RMETRIC_DIAGQUAD, 16, __VA_ARGS__>, a)))

static int launch_step(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int n_steps, const mm_fp_opts& opts,
                       mm_counters* d_counters, bool midpoint) {
  ImplicitArgs a{};
  const int rc = fill_args(ctx, m, s, a);
  if (rc != MM_OK) return rc;
  a.step_size = h;
  a.n_steps = n_steps;
  a.opts = opts;
  a.counters = d_counters;
  a.no_refine = mm_refine_disabled();
  a.no_dual = mm_dual_disabled();
  a.lowrank_refresh = mm_lowrank_refresh();
  a.no_lowrank = mm_lowrank_disabled();
  {  // MICI_AMD_GLOBAL_SYM=0: products with the held inverse by the full column walk (A/B runs against sym_walk)
    static const int off = [] { const char* e = getenv("MICI_AMD_GLOBAL_SYM"); return (e && e[0] == '0') ? 1 : 0; }();
    a.no_sym = off;
  }
  if (midpoint) return MM_GLOB_DISPATCH(implicit_global_kernel, true);
  return MM_GLOB_DISPATCH(implicit_global_kernel, false);
}

int mm_launch_implicit_global(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int n_steps, const mm_fp_opts& opts,
                              mm_counters* d_counters) {
  return launch_step(ctx, m, s, h, n_steps, opts, d_counters, false);
}
int mm_launch_implicit_midpoint_global(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int n_steps,
                                       const mm_fp_opts& opts, mm_counters* d_counters) {
  return launch_step(ctx, m, s, h, n_steps, opts, d_counters, true);
}

int mm_launch_riemann_aux_global(mm_ctx* ctx, const mm_model* m, mm_state* s, int op, double* d_out, const double* d_z) {
  ImplicitArgs a{};
  const int rc = fill_args(ctx, m, s, a);
  if (rc != MM_OK) return rc;
  a.out = d_out;
  a.z = d_z;
  if (op == 0) return MM_GLOB_DISPATCH(riemann_aux_global_kernel, 0);
  if (op == 1) return MM_GLOB_DISPATCH(riemann_aux_global_kernel, 1);
  return MM_GLOB_DISPATCH(riemann_aux_global_kernel, 2);
}
