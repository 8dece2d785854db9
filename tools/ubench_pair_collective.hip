// Micro-benchmark (round 6): what a two-wave team collective costs under the occupancy of the c3 kernel - 128-thread
// workgroups, 36 KB of LDS each (four per CU, two waves per SIMD), every workgroup looping over
//   [K dependent v_fma_f64]  ->  publish (ds_write_b64, lanes < 32)  ->  s_waitcnt + s_barrier  ->  8 x ds_read_b128 + 32 fma
// against the same loop without the barrier, and against one-wave workgroups.  Shader-clock cycles per iteration (s_memtime
// on wave 0 of workgroup 0 and the mean over workgroups).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_pair_collective.hip -o gpurun_out/ubench_pair_collective
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0: publish + barrier + product; 1: no barrier (own data only); 2: scalar exchange (sum) only
__global__ void coll_kernel(int iters, int kdep, double seed, long long* ticks, double* sink) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int idx = 32 * wave + (lane & 31), half = lane >> 5;
  double row[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) row[k] = seed * (k + 1) * 1e-3 + 1e-6 * lane;
  double x = seed + 1e-9 * threadIdx.x;
  double* xs = lds;
  int xbuf = 0;
  for (int i = threadIdx.x; i < 256; i += blockDim.x) lds[i] = 0.0;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    for (int j = 0; j < kdep; ++j) x = __builtin_fma(x, 0.999999, 1e-9);
    double* b = xs + xbuf * 72;
    if (MODE == 2) {
      if (lane == 0) b[64 + wave * 4] = x;
      __syncthreads();
      x = b[64] + b[68];
      xbuf ^= 1;
      x = x * 0.5;
      continue;
    }
    if (lane < 32) b[idx] = x;
    if (MODE == 0) __syncthreads();
    else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
    xbuf ^= 1;
    const double* src = b + (MODE == 0 ? 32 * half : 32 * wave);
    double y[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const d4 vv = *reinterpret_cast<const d4*>(src + 4 * k);
#pragma unroll
      for (int e = 0; e < 4; ++e) y[e] = __builtin_fma(row[4 * k + e], vv[e], y[e]);
    }
    x = ((y[0] + y[1]) + (y[2] + y[3])) * 1e-3 + seed;
  }
  const long long t1 = __builtin_readcyclecounter();
  if (lane == 0 && wave == 0) ticks[blockIdx.x] = t1 - t0;
  if (x == 12345.678) sink[0] = x;
}

template <int MODE>
void run(const char* name, int blocks, int threads, int lds_bytes, int iters, int kdep) {
  long long* d_t;
  double* d_s;
  hipMalloc(&d_t, blocks * sizeof(long long));
  hipMalloc(&d_s, 8);
  hipFuncSetAttribute(reinterpret_cast<const void*>(coll_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(coll_kernel<MODE>, dim3(blocks), dim3(threads), lds_bytes, 0, iters, kdep, 1.0, d_t, d_s);
    hipEventRecord(e1);
    hipDeviceSynchronize();
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(blocks);
  hipMemcpy(h.data(), d_t, blocks * sizeof(long long), hipMemcpyDeviceToHost);
  double mean = 0, mx = 0;
  for (auto v : h) { mean += (double)v; if ((double)v > mx) mx = (double)v; }
  mean /= blocks;
  printf("%-44s blocks %5d x %3d thr  kdep %3d: %8.1f ticks/iter (mean)  %8.1f (max)  %8.3f us/iter by event\n", name, blocks, threads,
         kdep, mean / iters, mx / iters, ms * 1e3 / iters);
  hipFree(d_t);
  hipFree(d_s);
}

int main() {
  const int iters = 2000, lds = 36 * 1024;
  for (int kdep : {0, 16, 64}) {
    for (int blocks : {256, 512, 1024, 2048}) {
      run<0>("pair: publish + barrier + product", blocks, 128, lds, iters, kdep);
    }
    for (int blocks : {256, 1024}) run<2>("pair: scalar exchange (sum)", blocks, 128, lds, iters, kdep);
    for (int blocks : {256, 1024, 2048}) run<1>("one wave: publish + product, no barrier", blocks, 64, lds / 2, iters, kdep);
  }
  return 0;
}
