// Host side of the wave-per-chain constrained kernels (device code: constrained_wave.h).
#include "constrained_wave.h"

using namespace mmcon;
using namespace mmdev;
using namespace mmconw;

namespace {

template <int C, int NE, bool GAUSS = false>
int launch_wave(mm_ctx* ctx, const ConArgs& a, int which, double* d_out) {
  constexpr int W = waves_per_block<NE>();
  const size_t lds = (size_t)W * wave_lds<NE>() * sizeof(double);
  const unsigned blocks = (unsigned)((a.n_chains + W - 1) / W);
  if (which != 0) {
    MM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(constrained_aux_wave_kernel<C, NE, GAUSS>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((constrained_aux_wave_kernel<C, NE, GAUSS>), dim3(blocks), dim3(64 * W), lds, ctx->stream, a, which,
                       d_out);
    MM_HIP_CHECK(ctx, hipGetLastError());
    return MM_OK;
  }
  MM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(constrained_wave_kernel<C, NE, GAUSS>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((constrained_wave_kernel<C, NE, GAUSS>), dim3(blocks), dim3(64 * W), lds, ctx->stream, a);
  MM_HIP_CHECK(ctx, hipGetLastError());
  return MM_OK;
}

// coordinates per lane for a model: 1 (D <= 64), 4 (D <= 256), 16 (D <= 1024, at most two constraints: the Jacobian
// copies of a step must fit a lone wave's 512 registers), 0: not covered
int elements_per_lane(int dim, int n_constr) {
  if (dim <= 64) return 1;
  if (dim <= 256) return 4;
  if (dim <= 1024 && n_constr <= 2) return 16;
  return 0;
}

template <int C>
int launch_wave_c(mm_ctx* ctx, const ConArgs& a, int which, double* d_out) {
  const int ne = elements_per_lane(a.dim, C);
  if (a.gaussian) {  // (64 < D <= 256: mm_constrained_wave_supports)
    if (ne == 4) return launch_wave<C, 4, true>(ctx, a, which, d_out);
    mm_set_error(ctx, "constrained kernels: the Gaussian split on the wave-per-chain kernel covers 64 < dim <= 256");
    return MM_ERR_UNSUPPORTED;
  }
  if (ne == 1) return launch_wave<C, 1>(ctx, a, which, d_out);
  if (ne == 4) return launch_wave<C, 4>(ctx, a, which, d_out);
  if constexpr (C <= 2) {
    if (ne == 16) return launch_wave<C, 16>(ctx, a, which, d_out);
  }
  mm_set_error(ctx, "constrained kernels: dim <= 256 with up to 8 constraints, dim <= 1024 with up to 2");
  return MM_ERR_UNSUPPORTED;
}

}  // namespace

// true if the wave-per-chain kernels cover this model (the caller falls back to the lane-per-chain path otherwise)
bool mm_constrained_wave_supports(const mmcon::ConArgs& a, int n_constr) {
  if (a.dim <= 8 || n_constr < 1 || n_constr > 8 || n_constr >= a.dim) return false;
  if (elements_per_lane(a.dim, n_constr) == 0) return false;
  // Gaussian split (round 5): beyond the lane-per-chain core's D = 64 only - below, that core is the path the existing
  // fixtures pin - and up to 256 (three more coordinate arrays per lane do not fit the D <= 1024 instantiation)
  if (a.gaussian && (a.dim <= 64 || a.dim > 256)) return false;
  switch (a.constr) {
    case MM_CONSTR_LINEAR: case MM_CONSTR_SPHERE: case MM_CONSTR_CIRCLE: case MM_CONSTR_FIRST: return true;
    case MM_CONSTR_SPHERE_PLANE: return n_constr == 2;
    default: return false;
  }
}

// which: 0 the leapfrog step, 1 cotangent projection of the momenta, 2 out += log_det_sqrt_gram
int mm_launch_constrained_wave(mm_ctx* ctx, int n_constr, const mmcon::ConArgs& a, int which, double* d_out) {
  switch (n_constr) {
    case 1: return launch_wave_c<1>(ctx, a, which, d_out);
    case 2: return launch_wave_c<2>(ctx, a, which, d_out);
    case 3: return launch_wave_c<3>(ctx, a, which, d_out);
    case 4: return launch_wave_c<4>(ctx, a, which, d_out);
    case 5: return launch_wave_c<5>(ctx, a, which, d_out);
    case 6: return launch_wave_c<6>(ctx, a, which, d_out);
    case 7: return launch_wave_c<7>(ctx, a, which, d_out);
    default: return launch_wave_c<8>(ctx, a, which, d_out);
  }
}
