// Constrained leapfrog on DenseConstrainedEuclideanMetricSystem, one lane per chain, everything in
// registers (D <= 8 exact, 8 < D <= 16 through the zero-padded capacity-16 instantiations of
// k_constrained_wide.hip; C <= 3 constraints, C < D).  gfx950 / CDNA4.
//
// Replaces, per chain and per step (reference /root/reference/src/mici):
//   ConstrainedLeapfrogIntegrator._step / _step_a / _step_b / _h2_flow_retraction_onto_manifold /
//       _project_onto_cotangent_space                       integrators.py:929-984
//   solve_projection_onto_manifold_newton / _quasi_newton / _newton_with_line_search
//                                                           solvers.py:429-469, 303-343, 561-614
//   ConstrainedEuclideanMetricSystem.constr / jacob_constr / dh2_flow_dmom / gram / inv_gram /
//       project_onto_cotangent_space                        systems.py:786-873
//   DenseConstrainedEuclideanMetricSystem.jacob_constr_inner_product  systems.py:1010-1022
//   DensePositiveDefiniteMatrix (C x C Gram: Cholesky + explicit inverse), DenseSquareMatrix /
//       InverseLUFactoredSquareMatrix (C x C residual Jacobian: partially pivoted LU)
//                                                           matrices.py:1161-1188, 1270-1411
//   dens_wrt_hausdorff=False: h1 += log det sqrt Gram, dh1_dpos += mhp_constr(inv_gram J M^-1)
//                                                           systems.py:829-831, 846-862, 1024-1031
//   GaussianDenseConstrainedEuclideanMetricSystem (exact h2 rotation, DenseSymmetricMatrix Gram-type
//       matrices, dh2_flow_dmom = (V diag(sin(w|t|) w) V^T, V diag(cos(w|t|)) V^T))   systems.py:1034-1184
// The state of a chain is 2*D doubles (48 B for the torus): there is no HBM roofline to speak of, the
// kernel is FP64-VALU / transcendental bound; data-dependent Newton iteration counts are handled by
// SIMT masking (lanes of a wave wait for their slowest chain).
#include <cstdlib>
#include <cstring>

#include "constrained_core.h"

using namespace mmcon;

// capacity-16 instantiations (k_constrained_wide.hip)
int mm_launch_constrained_wide(mm_ctx* ctx, int n_constr, const mmcon::ConArgs& a, int which, double* h_out);
// one wave per chain, 8 < D <= 64 (k_constrained_wave.hip)
bool mm_constrained_wave_supports(const mmcon::ConArgs& a, int n_constr);
int mm_launch_constrained_wave(mm_ctx* ctx, int n_constr, const mmcon::ConArgs& a, int which, double* d_out);

namespace {

template <int C>
int launch_c(mm_ctx* ctx, int dim, const ConArgs& a, int which, double* h_out) {
  switch (dim) {
#define MM_CASE(DD)                                                                      \
  case DD:                                                                               \
    if constexpr (C < DD || DD == 1) return launch_cd<C, DD, false>(ctx, a, which, h_out); \
    break;
    MM_CASE(1)
    MM_CASE(2)
    MM_CASE(3)
    MM_CASE(4)
    MM_CASE(5)
    MM_CASE(6)
    MM_CASE(7)
    MM_CASE(8)
#undef MM_CASE
    default: break;
  }
  mm_set_error(ctx, "constrained leapfrog kernels need fewer constraints than dimensions");
  return MM_ERR_UNSUPPORTED;
}

int launch(mm_ctx* ctx, const mm_model* m, const ConArgs& a, int which, double* h_out = nullptr) {
  // (before the run-time compiled branch: a user constraint on the built-in funnel target would run the core with an
  // empty TargetAux - a silently wrong gradient; mm_model_create refuses the combination as well)
  if (m->target == MM_TARGET_FUNNEL) {
    mm_set_error(ctx, "constrained kernels: the funnel target needs a wave-collective gradient");
    return MM_ERR_UNSUPPORTED;
  }
  if (m->rtc_con_module) {  // user constraint / target: the core compiled around the user's source (mm_rtc.hip)
    ConArgs copy = a;
    return mm_rtc_launch_constrained(ctx, m, which, &copy, a.n_chains, h_out);
  }
  // exact register-resident kernels: D <= 8 with C <= 3.  Beyond: the step of a plain (dens_wrt_hausdorff, no Gaussian
  // split) system with a built-in constraint runs one wave per chain (k_constrained_wave.hip); everything else up to
  // D = 64, C = 8 - the other system variants, the auxiliary kernels, and every size with
  // MICI_AMD_CONSTRAINED_KERNEL=lane (A/B runs) - runs the padded (capacity 16 or 64) instantiations of the
  // lane-per-chain core, whose per-chain arrays live in scratch
  static const bool force_lane = [] {
    const char* e = getenv("MICI_AMD_CONSTRAINED_KERNEL");
    return e && strcmp(e, "lane") == 0;
  }();
  if ((which == K_STEP || which == K_PROJECT || m->dim > 64) && !force_lane && mm_constrained_wave_supports(a, m->n_constr))
    return mm_launch_constrained_wave(ctx, m->n_constr, a, which == K_STEP ? 0 : (which == K_PROJECT ? 1 : 2), h_out);
  if (m->dim > 8 || m->n_constr > 3) return mm_launch_constrained_wide(ctx, m->n_constr, a, which, h_out);
  switch (m->n_constr) {
    case 1: return launch_c<1>(ctx, m->dim, a, which, h_out);
    case 2: return launch_c<2>(ctx, m->dim, a, which, h_out);
    default: return launch_c<3>(ctx, m->dim, a, which, h_out);
  }
}

}  // namespace

int mm_launch_constrained_leapfrog(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int n_steps,
                                   const mm_proj_opts& opts, mm_counters* d_counters) {
  ConArgs a = make_args(m, s);
  a.step_size = h;
  a.n_steps = n_steps;
  a.opts = opts;
  a.counters = d_counters;
  return launch(ctx, m, a, K_STEP);
}

int mm_launch_constrained_project_momentum(mm_ctx* ctx, const mm_model* m, mm_state* s) {
  ConArgs a = make_args(m, s);
  return launch(ctx, m, a, K_PROJECT);
}

int mm_launch_constrained_add_log_det_sqrt_gram(mm_ctx* ctx, const mm_model* m, mm_state* s, double* d_out) {
  ConArgs a = make_args(m, s);
  return launch(ctx, m, a, K_LOGDET, d_out);
}
