// Capacity-64 instantiations of the constrained leapfrog core for 16 < D <= 64, C = 1..4 (see
// k_constrained_wide.hip; split over two files so that they compile in parallel).
#include "constrained_core.h"

using namespace mmcon;

int mm_launch_constrained_wide64_lo(mm_ctx* ctx, int n_constr, const mmcon::ConArgs& a, int which, double* h_out) {
  switch (n_constr) {
    case 1: return launch_cd<1, 64, true>(ctx, a, which, h_out);
    case 2: return launch_cd<2, 64, true>(ctx, a, which, h_out);
    case 3: return launch_cd<3, 64, true>(ctx, a, which, h_out);
    case 4: return launch_cd<4, 64, true>(ctx, a, which, h_out);
    default:
      mm_set_error(ctx, "constrained leapfrog kernels: unsupported number of constraints");
      return MM_ERR_UNSUPPORTED;
  }
}
