// Host side of the SoftAbs kernels (device code: softabs.h).
#include "softabs.h"

namespace {

using namespace mmdev;
using namespace mmimp;
using namespace mmsoftabs;

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
  S.a.no_refine = mm_refine_disabled();
  S.coeff = m->d_rmetric_params;
  return S;
}

int check_dim(mm_ctx* ctx, const mm_model* m) {
  if (m->dim > 256) {
    mm_set_error(ctx, "SoftAbs kernels support dim <= 256 (one workgroup per chain; matrices in LDS up to 64)");
    return MM_ERR_UNSUPPORTED;
  }
  return MM_OK;
}

// the global-memory matrices of the NP = 128 / 256 kernels: grown on demand, kept with the state
int ensure_work(mm_ctx* ctx, mm_state* s, size_t doubles_per_chain, double** out) {
  const size_t need = (size_t)s->n * doubles_per_chain * sizeof(double);
  if (need > s->work_bytes) {
    MM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (s->d_work) (void)hipFree(s->d_work);
    s->d_work = nullptr;
    s->work_bytes = 0;
    MM_HIP_CHECK(ctx, hipMalloc(&s->d_work, need));
    s->work_bytes = need;
  }
  *out = static_cast<double*>(s->d_work);
  return MM_OK;
}

// MICI_AMD_EIG_CACHE=0: every launch starts its chains from the identity (A/B runs)
bool eig_cache_disabled() {
  static const bool off = [] {
    const char* e = getenv("MICI_AMD_EIG_CACHE");
    return e && e[0] == '0';
  }();
  return off;
}

// the bases the chains of a state carry between launches, zeroed - "no basis yet" - when (re)allocated
int ensure_eig(mm_ctx* ctx, mm_state* s, size_t doubles_per_chain, double** out) {
  const size_t need = (size_t)s->n * doubles_per_chain * sizeof(double);
  if (need != s->eig_bytes) {
    MM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (s->d_eig) (void)hipFree(s->d_eig);
    s->d_eig = nullptr;
    s->eig_bytes = 0;
    MM_HIP_CHECK(ctx, hipMalloc(&s->d_eig, need));
    s->eig_bytes = need;
    MM_HIP_CHECK(ctx, hipMemsetAsync(s->d_eig, 0, need, ctx->stream));
  }
  *out = s->d_eig;
  return MM_OK;
}

template <bool MIDPOINT, int NP>
int launch_softabs_np(mm_ctx* ctx, mm_state* s, SaArgs S) {
  using B = SoftAbsBackendT<NP>;
  if (B::kWorkDoubles) {
    const int rc = ensure_work(ctx, s, B::kWorkDoubles, &S.work);
    if (rc != MM_OK) return rc;
  }
  if (!S.a.no_refine && !eig_cache_disabled()) {
    const int rc = ensure_eig(ctx, s, B::kEigDoubles, &S.eig);
    if (rc != MM_OK) return rc;
  }
  const size_t lds = B::kLdsDoubles * sizeof(double);
  MM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(softabs_leapfrog_kernel<MIDPOINT, NP>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((softabs_leapfrog_kernel<MIDPOINT, NP>), dim3((unsigned)s->n), dim3(NT), lds, ctx->stream, S);
  MM_HIP_CHECK(ctx, hipGetLastError());
  return MM_OK;
}

template <int NP>
int launch_aux_np(mm_ctx* ctx, mm_state* s, SaArgs S, double* d_out, const double* d_z) {
  using B = SoftAbsBackendT<NP>;
  if (B::kWorkDoubles) {
    const int rc = ensure_work(ctx, s, B::kWorkDoubles, &S.work);
    if (rc != MM_OK) return rc;
  }
  if (!S.a.no_refine && !eig_cache_disabled()) {
    const int rc = ensure_eig(ctx, s, B::kEigDoubles, &S.eig);
    if (rc != MM_OK) return rc;
  }
  const size_t lds = B::kLdsDoubles * sizeof(double);
  MM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(softabs_aux_kernel<NP>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(softabs_aux_kernel<NP>, dim3((unsigned)s->n), dim3(NT), lds, ctx->stream, S, d_out, d_z);
  MM_HIP_CHECK(ctx, hipGetLastError());
  return MM_OK;
}

// a user Hessian: the same kernels, compiled at run time around the user's source (mm_rtc.hip) for the padded size of the
// model's dim (softabs_source(): MM_SA_NP), USERH
template <int NP>
int launch_user_np(mm_ctx* ctx, const mm_model* m, mm_state* s, SaArgs S, int which, double* d_out, const double* d_z) {
  using B = SoftAbsBackendT<NP, true>;
  if (B::kWorkDoubles) {
    const int rc = ensure_work(ctx, s, B::kWorkDoubles, &S.work);
    if (rc != MM_OK) return rc;
  }
  if (!S.a.no_refine && !eig_cache_disabled()) {
    const int rc = ensure_eig(ctx, s, B::kEigDoubles, &S.eig);
    if (rc != MM_OK) return rc;
  }
  return mm_rtc_launch_softabs(ctx, m, which, &S, s->n, d_out, d_z);
}
int launch_user(mm_ctx* ctx, const mm_model* m, mm_state* s, SaArgs S, int which, double* d_out, const double* d_z) {
  if (m->dim <= 64) return launch_user_np<64>(ctx, m, s, S, which, d_out, d_z);
  return m->dim <= 128 ? launch_user_np<128>(ctx, m, s, S, which, d_out, d_z)
                       : launch_user_np<256>(ctx, m, s, S, which, d_out, d_z);
}

}  // namespace

template <bool MIDPOINT>
static int launch_softabs(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int n_steps,
                          const mm_fp_opts& opts, mm_counters* d_counters) {
  int rc = check_dim(ctx, m);
  if (rc != MM_OK) return rc;
  if (s->n == 0) return MM_OK;
  SaArgs S = make_args(m, s);
  S.a.step_size = h;
  S.a.n_steps = n_steps;
  S.a.opts = opts;
  S.a.counters = d_counters;
  if (m->rmetric == MM_RMETRIC_SOFTABS_USER) return launch_user(ctx, m, s, S, MIDPOINT ? 1 : 0, nullptr, nullptr);
  if (m->dim <= 64) return launch_softabs_np<MIDPOINT, 64>(ctx, s, S);
  return m->dim <= 128 ? launch_softabs_np<MIDPOINT, 128>(ctx, s, S) : launch_softabs_np<MIDPOINT, 256>(ctx, s, S);
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
  if (s->n == 0) return MM_OK;
  SaArgs S = make_args(m, s);
  S.op = op;
  if (m->rmetric == MM_RMETRIC_SOFTABS_USER) return launch_user(ctx, m, s, S, 2, d_out, d_z);
  if (m->dim <= 64) return launch_aux_np<64>(ctx, s, S, d_out, d_z);
  return m->dim <= 128 ? launch_aux_np<128>(ctx, s, S, d_out, d_z) : launch_aux_np<256>(ctx, s, S, d_out, d_z);
}
