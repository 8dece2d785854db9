// The tile-update pass of k_implicit_blk16.hip in isolation: NT tiles in registers, each updated by four dependent
// v_mfma_f64_16x16x4_f64 whose B operands (32 bytes per lane) come from LDS.  Variants of the operand reads:
//   0: two ds_read_b128 per tile, each waited for right before the two MFMAs that use it (what the kernel compiles to)
//   1: both reads of a tile issued first, then its four MFMAs
//   2: the NEXT tile's operands are read before the current tile's MFMAs (software pipelining)
//   3: no LDS reads at all (operands in registers)
// Prints shader-clock cycles per MFMA per wave with 1 and 2 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 tools/ubench_mfma_tiles.hip -o /tmp/u && /tmp/u
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int NT = 17;
constexpr int CS = 18;

template <int VAR>
__global__ __launch_bounds__(512, 2) void tiles_kernel(double* out, long long* cycles, int iters) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  d4 acc[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  for (int i = threadIdx.x; i < 256 * CS; i += blockDim.x) lds[i] = 1e-3 * i;
  const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15;
  const d4 a = d4{1.0 + lane * 1e-9, 1.0, 0.5, 0.25};
  const double* xl = lds + j * CS + 4 * g;
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    if constexpr (VAR == 0 || VAR == 1) {
#pragma unroll
      for (int s = 0; s < NT; ++s) {
        const d4 bx = *reinterpret_cast<const volatile d4*>(xl + (s & 15) * (16 * CS));
        if constexpr (VAR == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], bx[kk], acc[s], 0, 0, 0);
      }
    } else if constexpr (VAR == 2) {
      d4 bx = *reinterpret_cast<const d4*>(xl);
#pragma unroll
      for (int s = 0; s < NT; ++s) {
        d4 bxn = bx;
        if (s + 1 < NT) bxn = *reinterpret_cast<const d4*>(xl + ((s + 1) & 15) * (16 * CS));
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], bx[kk], acc[s], 0, 0, 0);
        bx = bxn;
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
#pragma unroll
      for (int s = 0; s < NT; ++s)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], a[3 - kk], acc[s], 0, 0, 0);
    }
  }
  double sum = 0.0;
#pragma unroll
  for (int i = 0; i < NT; ++i) sum += acc[i][0] + acc[i][3];
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  if (lane == 0) cycles[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int VAR>
void run(int waves_per_simd, int iters) {
  const int threads = 64 * 4 * waves_per_simd, blocks = 256;
  double* out;
  long long* cyc;
  hipMalloc(&out, sizeof(double) * threads * blocks);
  hipMalloc(&cyc, sizeof(long long) * blocks * threads / 64);
  const size_t lds = 256 * CS * sizeof(double);
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(tiles_kernel<VAR>, dim3(blocks), dim3(threads), lds, 0, out, cyc, iters);
  hipDeviceSynchronize();
  std::vector<long long> h(blocks * threads / 64);
  hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
  double mean = 0.0;
  for (auto v : h) mean += (double)v;
  mean /= h.size();
  const double per = mean / (iters * 4.0 * NT);
  printf("variant %d  waves/SIMD %d : %7.1f cycles per MFMA per wave, %7.1f per SIMD\n", VAR, waves_per_simd, per,
         per / waves_per_simd);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  for (int w = 1; w <= 2; ++w) {
    run<0>(w, 100);
    run<1>(w, 100);
    run<2>(w, 100);
    run<3>(w, 100);
  }
  return 0;
}
