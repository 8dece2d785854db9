// Micro-benchmark: sustained FP64 rate of v_mfma_f64_16x16x4_f64 and v_fma_f64 on gfx950, to anchor
// the "peak" of the roofline (the microarch guide lists no FP64 number).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_fp64.hip -o gpurun_out/ubench_fp64 && ./ubench_fp64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(double* out, int iters, double a0, double b0) {
  double4_t acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = double4_t{0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void fma_loop(double* out, int iters, double a0, double b0) {
  double acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = i;
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_fma(a, acc[i], b);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// both pipes in one wave: NACC mfma + NV fma per iteration
template <int NACC, int NV>
__global__ __launch_bounds__(256) void mixed_loop(double* out, int iters, double a0, double b0) {
  double4_t acc[NACC];
  double v[NV];
  for (int i = 0; i < NACC; ++i) acc[i] = double4_t{0, 0, 0, 0};
  for (int i = 0; i < NV; ++i) v[i] = i;
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = __builtin_fma(a, v[i], b);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  for (int i = 0; i < NV; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
double time_ms(F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("device %s, %d CUs, clock %d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
  const int cus = p.multiProcessorCount;
  double* out; hipMalloc(&out, sizeof(double) * 256 * cus * 8);
  const int iters = 20000;
  for (int wg_per_cu : {1, 2}) {
    const int blocks = cus * wg_per_cu;
    {
      double ms = time_ms([&] { hipLaunchKernelGGL(mfma_loop<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1e-3); });
      double flops = 2048.0 * 4 * iters * 4.0 * blocks;
      printf("mfma_f64_16x16x4 x4acc, %d WG/CU (256 thr): %.3f ms  %.1f TFLOP/s  %.1f cyc/mfma/SIMD @2.4GHz\n",
             wg_per_cu, ms, flops / ms / 1e9, ms * 1e-3 * 2.4e9 / (4.0 * iters * wg_per_cu));
    }
    {
      double ms = time_ms([&] { hipLaunchKernelGGL(mfma_loop<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0, 1e-3); });
      double flops = 2048.0 * 1 * iters * 4.0 * blocks;
      printf("mfma_f64_16x16x4 x1acc (dependent), %d WG/CU: %.3f ms  %.1f TFLOP/s  %.1f cyc/mfma @2.4GHz\n",
             wg_per_cu, ms, flops / ms / 1e9, ms * 1e-3 * 2.4e9 / (1.0 * iters * wg_per_cu));
    }
    {
      double ms = time_ms([&] { hipLaunchKernelGGL(fma_loop<8>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001, 1e-3); });
      double flops = 2.0 * 8 * iters * 256.0 * blocks;
      printf("v_fma_f64 x8acc, %d WG/CU: %.3f ms  %.1f TFLOP/s\n", wg_per_cu, ms, flops / ms / 1e9);
    }
    {
      double ms = time_ms([&] { hipLaunchKernelGGL((mixed_loop<4, 16>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001, 1e-3); });
      double flops = (2048.0 * 4 * 4.0 + 2.0 * 16 * 256.0) * iters * blocks;
      printf("mixed 4 mfma + 16 fma per iter, %d WG/CU: %.3f ms  %.1f TFLOP/s total\n", wg_per_cu, ms, flops / ms / 1e9);
    }
  }
  hipFree(out);
  return 0;
}
