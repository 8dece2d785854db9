// Developer hook: the lean FP64 reciprocal / division / square root of mm_device.h (rcp_nr, fdiv, sqrt_rsqrt) evaluated
// element-wise on the device, next to the compiler's IEEE expansions of the same operations - the kernels replace IEEE
// division and sqrt by these on their critical paths, and tests/test_gpu_lean_math.py checks them over subnormal, huge,
// zero, infinite and NaN operands.  Nothing in here is part of the product library.
#include "mm_device.h"

#ifdef MM_DEV_KERNELS
namespace {

// out: [8][n] = rcp_nr(a), fdiv(a, b), sqrt_rsqrt(a).s, sqrt_rsqrt(a).rs, 1 / a, a / b, sqrt(a), 1 / sqrt(a)
__global__ void lean_math_kernel(const double* a, const double* b, double* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = a[i], y = b[i];
  double s, rs;
  mmdev::sqrt_rsqrt(x, &s, &rs);
  out[0 * n + i] = mmdev::rcp_nr(x);
  out[1 * n + i] = mmdev::fdiv(x, y);
  out[2 * n + i] = s;
  out[3 * n + i] = rs;
  out[4 * n + i] = 1.0 / x;
  out[5 * n + i] = x / y;
  out[6 * n + i] = sqrt(x);
  out[7 * n + i] = 1.0 / sqrt(x);
}

}  // namespace

extern "C" __attribute__((visibility("default"))) int mm_debug_lean_math(mm_ctx* ctx, const double* a, const double* b, double* out,
                                                                          int n) {
  if (!ctx || !a || !b || !out || n <= 0) return MM_ERR_INVALID;
  double *da = nullptr, *db = nullptr, *dout = nullptr;
  MM_HIP_CHECK(ctx, hipMalloc(&da, sizeof(double) * n));
  MM_HIP_CHECK(ctx, hipMalloc(&db, sizeof(double) * n));
  MM_HIP_CHECK(ctx, hipMalloc(&dout, sizeof(double) * 8 * n));
  MM_HIP_CHECK(ctx, hipMemcpy(da, a, sizeof(double) * n, hipMemcpyHostToDevice));
  MM_HIP_CHECK(ctx, hipMemcpy(db, b, sizeof(double) * n, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(lean_math_kernel, dim3((n + 255) / 256), dim3(256), 0, ctx->stream, da, db, dout, n);
  MM_HIP_CHECK(ctx, hipGetLastError());
  MM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  MM_HIP_CHECK(ctx, hipMemcpy(out, dout, sizeof(double) * 8 * n, hipMemcpyDeviceToHost));
  (void)hipFree(da);
  (void)hipFree(db);
  (void)hipFree(dout);
  return MM_OK;
}
#endif
