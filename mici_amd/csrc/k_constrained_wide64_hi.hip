// Capacity-64 instantiations of the constrained leapfrog core for 16 < D <= 64, C = 5..8 (see
// k_constrained_wide.hip; split over two files so that they compile in parallel).
#include "constrained_core.h"

using namespace mmcon;

int mm_launch_constrained_wide64_hi(mm_ctx* ctx, int n_constr, const mmcon::ConArgs& a, int which, double* h_out) {
  switch (n_constr) {
    case 5: return launch_cd<5, 64, true>(ctx, a, which, h_out);
    case 6: return launch_cd<6, 64, true>(ctx, a, which, h_out);
    case 7: return launch_cd<7, 64, true>(ctx, a, which, h_out);
    case 8: return launch_cd<8, 64, true>(ctx, a, which, h_out);
    default:
      mm_set_error(ctx, "constrained leapfrog kernels: unsupported number of constraints");
      return MM_ERR_UNSUPPORTED;
  }
}
