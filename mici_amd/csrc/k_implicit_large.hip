// Host side of the VALU team kernels (device code: implicit_team.h).
#include "implicit_team.h"

namespace {

using namespace mmdev;
using namespace mmimp;
using namespace mmteam;

ImplicitArgs make_args(const mm_model* m, mm_state* s) {
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
  a.rparams = m->d_rmetric_padded;  // rank-one base matrix zero-padded to C::DP x C::DP
  return a;
}

template <class C, class K>
int launch(mm_ctx* ctx, K kernel, const ImplicitArgs& a) {
  const size_t lds = C::LDS_DOUBLES * sizeof(double);
  MM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kernel, dim3((unsigned)a.n_chains), dim3(C::NT), lds, ctx->stream, a);
  MM_HIP_CHECK(ctx, hipGetLastError());
  return MM_OK;
}

template <class C>
int launch_step(mm_ctx* ctx, const mm_model* m, const ImplicitArgs& a, bool midpoint) {
  const bool r1 = m->rmetric == MM_RMETRIC_RANK1;
  if (midpoint) {
    // (round 6: the rank-one-update metric's held inverse carried between the midpoint rule's evaluations by rank-two updates)
    if (r1 && a.no_lowrank == 0 && a.no_refine == 0)
      return launch<C>(ctx, implicit_team_kernel<C, MM_RMETRIC_RANK1, true, true>, a);
    return r1 ? launch<C>(ctx, implicit_team_kernel<C, MM_RMETRIC_RANK1, true>, a)
              : launch<C>(ctx, implicit_team_kernel<C, MM_RMETRIC_DIAGQUAD, true>, a);
  }
  // round 6: the rank-one-update metric's Woodbury path (implicit_core.h lowrank_solve / lowrank_update); MICI_AMD_LOWRANK=0:
  // the CG refinement
  if (r1 && a.no_lowrank == 0 && a.no_refine == 0)
    return launch<C>(ctx, implicit_team_kernel<C, MM_RMETRIC_RANK1, false, true>, a);
  return r1 ? launch<C>(ctx, implicit_team_kernel<C, MM_RMETRIC_RANK1, false>, a)
            : launch<C>(ctx, implicit_team_kernel<C, MM_RMETRIC_DIAGQUAD, false>, a);
}

template <class C>
int launch_aux(mm_ctx* ctx, const mm_model* m, const ImplicitArgs& a, int op) {
  const bool r1 = m->rmetric == MM_RMETRIC_RANK1;
  if (op == 0)
    return r1 ? launch<C>(ctx, riemann_aux_team_kernel<C, MM_RMETRIC_RANK1, 0>, a)
              : launch<C>(ctx, riemann_aux_team_kernel<C, MM_RMETRIC_DIAGQUAD, 0>, a);
  if (op == 1)
    return r1 ? launch<C>(ctx, riemann_aux_team_kernel<C, MM_RMETRIC_RANK1, 1>, a)
              : launch<C>(ctx, riemann_aux_team_kernel<C, MM_RMETRIC_DIAGQUAD, 1>, a);
  return r1 ? launch<C>(ctx, riemann_aux_team_kernel<C, MM_RMETRIC_RANK1, 2>, a)
            : launch<C>(ctx, riemann_aux_team_kernel<C, MM_RMETRIC_DIAGQUAD, 2>, a);
}

bool check_team_model(mm_ctx* ctx, const mm_model* m) {
  if (m->dim > CfgLarge::DP) {
    mm_set_error(ctx, "dense-Riemannian kernels support dim <= 279 (register-resident metric)");
    return false;
  }
  const int want = mm_team_padded_dim(m->dim);
  if (m->rmetric == MM_RMETRIC_RANK1 && (m->d_rmetric_padded == nullptr || m->rmetric_pad_dim != want)) {
    mm_set_error(ctx, "internal: rank-one base matrix was not padded for the team kernels");
    return false;
  }
  return true;
}

}  // namespace

namespace {
// 0: CfgMid, 1: CfgSmall, 2: CfgLarge.  MICI_AMD_TEAM=22 selects the 4-wave geometry where it applies
// (D <= 66); it and the 2-wave one lose to the wave-per-chain kernel at D <= 64 (tools/sweep_chains.sh).
int team_variant(int dim) {
  static const bool want_mid = [] {
    const char* e = getenv("MICI_AMD_TEAM");
    return e && atoi(e) == 22;
  }();
  if (dim <= CfgMid::DP && want_mid) return 0;
  return dim <= CfgSmall::DP ? 1 : 2;
}
}  // namespace

// leading dimension of the zero-padded base matrix the team kernels read (mm_model_create pads to it)
int mm_team_padded_dim(int dim) {
  const int v = team_variant(dim);
  return v == 0 ? CfgMid::DP : v == 1 ? CfgSmall::DP : CfgLarge::DP;
}

static int launch_large(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int n_steps, const mm_fp_opts& opts,
                        mm_counters* d_counters, bool midpoint) {
  if (!check_team_model(ctx, m)) return MM_ERR_UNSUPPORTED;
  ImplicitArgs a = make_args(m, s);
  a.step_size = h;
  a.n_steps = n_steps;
  a.opts = opts;
  a.counters = d_counters;
  a.no_refine = mm_refine_disabled();
  a.no_dual = mm_dual_disabled();
  a.no_lowrank = mm_lowrank_disabled();
  a.lowrank_refresh = mm_lowrank_refresh();
  const int v = team_variant(m->dim);
  return v == 0   ? launch_step<CfgMid>(ctx, m, a, midpoint)
         : v == 1 ? launch_step<CfgSmall>(ctx, m, a, midpoint)
                  : launch_step<CfgLarge>(ctx, m, a, midpoint);
}

int mm_launch_implicit_large(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int n_steps,
                             const mm_fp_opts& opts, mm_counters* d_counters) {
  return launch_large(ctx, m, s, h, n_steps, opts, d_counters, false);
}

int mm_launch_implicit_midpoint_large(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int n_steps,
                                      const mm_fp_opts& opts, mm_counters* d_counters) {
  return launch_large(ctx, m, s, h, n_steps, opts, d_counters, true);
}

int mm_launch_riemann_aux_large(mm_ctx* ctx, const mm_model* m, mm_state* s, int op, double* d_out,
                                const double* d_z) {
  if (!check_team_model(ctx, m)) return MM_ERR_UNSUPPORTED;
  ImplicitArgs a = make_args(m, s);
  a.out = d_out;
  a.z = d_z;
  const int v = team_variant(m->dim);
  return v == 0 ? launch_aux<CfgMid>(ctx, m, a, op) : v == 1 ? launch_aux<CfgSmall>(ctx, m, a, op) : launch_aux<CfgLarge>(ctx, m, a, op);
}
