// How fast do dependent chains of v_mfma_f64_16x16x4_f64 run on an MI355X SIMD?  One workgroup per CU; WAVES waves per
// SIMD; each wave runs `iters` rounds of NACC interleaved accumulator chains (NACC = 1: every MFMA depends on the
// previous one).  Prints shader-clock cycles per MFMA per wave and per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_mfma_chain.hip -o /tmp/ubench_mfma_chain && /tmp/ubench_mfma_chain
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void chain_kernel(double* out, long long* cycles, int iters) {
  d4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int NACC>
void run(int waves_per_simd, int iters) {
  const int threads = 64 * 4 * waves_per_simd, blocks = 256;
  double* out;
  long long* cyc;
  hipMalloc(&out, sizeof(double) * threads * blocks);
  hipMalloc(&cyc, sizeof(long long) * blocks * threads / 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(chain_kernel<NACC>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL(chain_kernel<NACC>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  hipEventRecord(e1, 0);
  hipDeviceSynchronize();
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks * threads / 64);
  hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
  double mean = 0.0;
  for (auto v : h) mean += (double)v;
  mean /= h.size();
  const double per_wave = mean / (iters * 8.0 * NACC);
  const double flops = 2048.0 * iters * 8.0 * NACC * (threads / 64.0) * blocks;
  printf("waves/SIMD %d  interleaved chains %d : %7.1f cycles per MFMA per wave, %7.1f per SIMD | launch %.3f ms = %.1f "
         "TFLOP/s, counter %.2f GHz\n", waves_per_simd, NACC, per_wave, per_wave / waves_per_simd, ms,
         flops / (ms * 1e-3) / 1e12, mean / (ms * 1e-3) / 1e9);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  for (int w = 1; w <= 2; ++w) {
    run<1>(w, 20000);
    run<4>(w, 20000);
  }
  return 0;
}
