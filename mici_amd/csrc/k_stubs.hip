// Temporary launch stubs (replaced as kernels land).
#include "mm_internal.h"
int mm_launch_softabs_leapfrog(mm_ctx* ctx, const mm_model*, mm_state*, double, int, const mm_fp_opts&, mm_counters*) {
  mm_set_error(ctx, "SoftAbs kernel not built yet");
  return MM_ERR_UNSUPPORTED;
}
int mm_launch_softabs_aux(mm_ctx* ctx, const mm_model*, mm_state*, int, double*, const double*) {
  mm_set_error(ctx, "SoftAbs kernel not built yet");
  return MM_ERR_UNSUPPORTED;
}
