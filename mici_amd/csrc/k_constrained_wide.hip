// Constrained leapfrog for 8 < D <= 16: the lane-per-chain kernels of constrained_core.h instantiated once at
// capacity 16 with the real dimension as a run-time argument (extra coordinates held at zero).  These spill
// part of the per-chain state to scratch; they exist for coverage (the reference's own adapter tests use a
// D = 10 sphere, tests/test_adapters.py:156-188), not for the BASELINE configurations.
#include "constrained_core.h"

using namespace mmcon;

int mm_launch_constrained_wide(mm_ctx* ctx, int n_constr, const mmcon::ConArgs& a, int which, double* h_out) {
  if (a.dim > 16) {
    mm_set_error(ctx, "constrained leapfrog kernels support dim <= 16 (register-resident chains)");
    return MM_ERR_UNSUPPORTED;
  }
  switch (n_constr) {
    case 1: return launch_cd<1, 16, true>(ctx, a, which, h_out);
    case 2: return launch_cd<2, 16, true>(ctx, a, which, h_out);
    case 3: return launch_cd<3, 16, true>(ctx, a, which, h_out);
    default:
      mm_set_error(ctx, "constrained leapfrog kernels support at most 3 constraints");
      return MM_ERR_UNSUPPORTED;
  }
}
