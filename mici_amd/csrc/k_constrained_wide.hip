// Constrained leapfrog beyond the register-resident sizes (D <= 8, C <= 3): the lane-per-chain kernels of
// constrained_core.h instantiated at a CAPACITY (16 here, 64 in k_constrained_wide64*.hip) with the real dimension
// as a run-time argument (extra coordinates held at zero), for up to 8 constraint functions.  Their per-chain arrays
// (a 64-vector is 512 B, a C x D Jacobian up to 4 KB) live in scratch, 5-26 KB per lane: these kernels exist for
// coverage of the reference's sizes (its own adapter tests use a D = 10 sphere, tests/test_adapters.py:156-188;
// SURVEY section 8f asks for D <= 64, C <= 8), not for the BASELINE configurations.
#include "constrained_core.h"

using namespace mmcon;

int mm_launch_constrained_wide64_lo(mm_ctx* ctx, int n_constr, const mmcon::ConArgs& a, int which, double* h_out);
int mm_launch_constrained_wide64_hi(mm_ctx* ctx, int n_constr, const mmcon::ConArgs& a, int which, double* h_out);

int mm_launch_constrained_wide(mm_ctx* ctx, int n_constr, const mmcon::ConArgs& a, int which, double* h_out) {
  if (a.dim > 64) {
    mm_set_error(ctx, "constrained leapfrog kernels support dim <= 64 (lane-per-chain, per-chain arrays in scratch)");
    return MM_ERR_UNSUPPORTED;
  }
  if (n_constr < 1 || n_constr > 8) {
    mm_set_error(ctx, "constrained leapfrog kernels support at most 8 constraints");
    return MM_ERR_UNSUPPORTED;
  }
  if (a.dim > 16) {
    return n_constr <= 4 ? mm_launch_constrained_wide64_lo(ctx, n_constr, a, which, h_out)
                         : mm_launch_constrained_wide64_hi(ctx, n_constr, a, which, h_out);
  }
  switch (n_constr) {
    case 1: return launch_cd<1, 16, true>(ctx, a, which, h_out);
    case 2: return launch_cd<2, 16, true>(ctx, a, which, h_out);
    case 3: return launch_cd<3, 16, true>(ctx, a, which, h_out);
    case 4: return launch_cd<4, 16, true>(ctx, a, which, h_out);
    case 5: return launch_cd<5, 16, true>(ctx, a, which, h_out);
    case 6: return launch_cd<6, 16, true>(ctx, a, which, h_out);
    case 7: return launch_cd<7, 16, true>(ctx, a, which, h_out);
    default: return launch_cd<8, 16, true>(ctx, a, which, h_out);
  }
}
