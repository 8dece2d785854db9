// Micro-benchmark: the matrix-core phase of the team sweep in isolation.  One 512-thread workgroup per CU; per
// iteration every wave loads its 17 + 2 operands from LDS and issues 17 independent v_mfma_f64_16x16x4_f64,
// then the workgroup hits a barrier (variant 1: two barriers, variant 2: none).  Reports cycles per iteration
// (s_memtime) against the matrix-core floor of 2 waves/SIMD x 17 x 64 = 2176.
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 tools/ubench_mfma_phase.hip -o /tmp/ub && /tmp/ub
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NMFMA>
__global__ __launch_bounds__(512, 2) void phase_kernel(int iters, int variant, long long* ticks, double* sink) {
  __shared__ double qt[2048], wt[2048];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 2048; i += 512) { qt[i] = 1e-3 * i; wt[i] = 1e-4 * (i + 1); }
  __syncthreads();
  d4 acc[NMFMA];
  for (int s = 0; s < NMFMA; ++s) acc[s] = d4{0, 0, 0, 0};
  const int g = lane >> 4, j = lane & 15;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const double av = wt[((16 * ((wave + it) & 15) + j) << 2) + g];
    double bv[NMFMA];
#pragma unroll
    for (int s = 0; s < NMFMA; ++s) bv[s] = qt[((16 * ((s + it) & 15) + j) << 2) + g];
#pragma unroll
    for (int s = 0; s < NMFMA; ++s) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv[s], acc[s], 0, 0, 0);
    if (variant == 1) {
      __syncthreads();
      if (tid < 256) wt[tid * 4] = acc[0][0] * 1e-30 + wt[tid * 4];  // a dependent publish, like the sweep
      __syncthreads();
    } else if (variant == 3) {
      __syncthreads();
    }
  }
  double r = 0;
  for (int s = 0; s < NMFMA; ++s) r += acc[s][0] + acc[s][3];
  const long long t1 = __builtin_readcyclecounter();
  if (tid == 0) ticks[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * 512 + tid] = r;
}

int main() {
  long long* d_t; double* d_s;
  hipMalloc(&d_t, 1024 * sizeof(long long)); hipMalloc(&d_s, 1024 * 512 * sizeof(double));
  const int iters = 2000;
  for (int blocks : {1, 256}) for (int variant : {2, 3, 1}) {
    hipLaunchKernelGGL(phase_kernel<17>, dim3(blocks), dim3(512), 0, 0, 10, variant, d_t, d_s);
    hipLaunchKernelGGL(phase_kernel<17>, dim3(blocks), dim3(512), 0, 0, iters, variant, d_t, d_s);
    hipDeviceSynchronize();
    long long t; hipMemcpy(&t, d_t, sizeof(t), hipMemcpyDeviceToHost);
    printf("blocks %3d  %-38s %7.0f cycles / iteration (floor 2176)\n", blocks,
           variant == 2 ? "17 MFMA/wave, no barrier" : variant == 3 ? "17 MFMA/wave + 1 barrier" : "17 MFMA/wave + publish + 2 barriers",
           (double)t / iters);
  }
  return 0;
}
