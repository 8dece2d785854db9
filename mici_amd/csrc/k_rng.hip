// Device-side random draws for the transitions: the reference draws a standard-normal
// vector per momentum refresh (transitions.py:136-142, systems.py:365-366), a uniform per Metropolis accept step
// (transitions.py:300-309) and, for the random-length transition, an integer number of steps (transitions.py:383-386)
// from a NumPy Generator on the host.  Here they come from a counter-based generator evaluated on the device, so a
// transition uploads nothing: Philox4x32-10 (Salmon et al., SC'11) keyed by the job's seed, with the counter
//     (chain index: 64 bits | transition index: 40 bits | purpose: 2 bits | block index: 22 bits)
// A draw is a pure function of (seed, global chain index, transition, purpose, position): independent of how the
// chains are sharded over GPUs and of launch geometry, hence bit-reproducible.  oracle/rng.py restates it in NumPy.
//   uniform double  u = ((x0 >> 5) * 2^26 + (x1 >> 6)) * 2^-53   in [0, 1)   (53 random bits)
//   normal pair     Box-Muller: r = sqrt(-2 log(1 - u_a)), (r cos(2 pi u_b), r sin(2 pi u_b)) for dims 2 k, 2 k + 1
//   step count      lo + ((uint64) x0 * (hi - lo) >> 32)
#include "mm_internal.h"

namespace {

struct Philox {
  uint32_t c[4];
};

__device__ __forceinline__ Philox philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                                uint32_t k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
    const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0;
    c1 = n1;
    c2 = n2;
    c3 = n3;
    k0 += W0;
    k1 += W1;
  }
  return Philox{{c0, c1, c2, c3}};
}

__device__ __forceinline__ Philox draw_block(uint64_t seed, uint64_t chain, uint64_t transition, uint32_t purpose,
                                             uint32_t block) {
  const uint32_t c3 = ((uint32_t)(transition >> 32) & 0xFFu) | ((purpose & 3u) << 8) | ((block & 0x3FFFFFu) << 10);
  return philox4x32_10((uint32_t)chain, (uint32_t)(chain >> 32), (uint32_t)transition, c3, (uint32_t)seed,
                       (uint32_t)(seed >> 32));
}

__device__ __forceinline__ double u53(uint32_t a, uint32_t b) {
  return ((double)(a >> 5) * 67108864.0 + (double)(b >> 6)) * (1.0 / 9007199254740992.0);
}

enum { PURPOSE_NORMAL = 0, PURPOSE_UNIFORM = 1, PURPOSE_STEPS = 2 };

// z[chain][d]: one thread per PAIR of dimensions
__global__ void rng_normal_kernel(double* __restrict__ z, int64_t n_chains, int dim, uint64_t seed, uint64_t chain_offset,
                                  uint64_t transition) {
  const int pairs = (dim + 1) >> 1;
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_chains * pairs) return;
  const int64_t chain = idx / pairs;
  const int k = (int)(idx - chain * pairs);
  const Philox x = draw_block(seed, chain_offset + (uint64_t)chain, transition, PURPOSE_NORMAL, (uint32_t)k);
  const double ua = u53(x.c[0], x.c[1]), ub = u53(x.c[2], x.c[3]);
  const double r = sqrt(-2.0 * log(1.0 - ua));
  const double th = 6.283185307179586476925286766559 * ub;
  double* row = z + chain * dim;
  row[2 * k] = r * cos(th);
  if (2 * k + 1 < dim) row[2 * k + 1] = r * sin(th);
}

__global__ void rng_uniform_kernel(double* __restrict__ u, int64_t n_chains, uint64_t seed, uint64_t chain_offset,
                                   uint64_t transition) {
  const int64_t chain = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (chain >= n_chains) return;
  const Philox x = draw_block(seed, chain_offset + (uint64_t)chain, transition, PURPOSE_UNIFORM, 0);
  u[chain] = u53(x.c[0], x.c[1]);
}

__global__ void rng_steps_kernel(int32_t* __restrict__ steps, int64_t n_chains, uint64_t seed, uint64_t chain_offset,
                                 uint64_t transition, int32_t lo, int32_t hi) {
  const int64_t chain = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (chain >= n_chains) return;
  const Philox x = draw_block(seed, chain_offset + (uint64_t)chain, transition, PURPOSE_STEPS, 0);
  steps[chain] = lo + (int32_t)(((uint64_t)x.c[0] * (uint64_t)(uint32_t)(hi - lo)) >> 32);
}

}  // namespace

int mm_launch_rng_normal(mm_ctx* ctx, double* d_z, int64_t n, int dim, uint64_t seed, uint64_t chain_offset,
                         uint64_t transition) {
  const int64_t work = n * ((dim + 1) >> 1);
  if (work == 0) return MM_OK;
  hipLaunchKernelGGL(rng_normal_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, ctx->stream, d_z, n, dim,
                     seed, chain_offset, transition);
  MM_HIP_CHECK(ctx, hipGetLastError());
  return MM_OK;
}

int mm_launch_rng_uniform(mm_ctx* ctx, double* d_u, int64_t n, uint64_t seed, uint64_t chain_offset,
                          uint64_t transition) {
  if (n == 0) return MM_OK;
  hipLaunchKernelGGL(rng_uniform_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_u, n, seed,
                     chain_offset, transition);
  MM_HIP_CHECK(ctx, hipGetLastError());
  return MM_OK;
}

int mm_launch_rng_steps(mm_ctx* ctx, int32_t* d_steps, int64_t n, uint64_t seed, uint64_t chain_offset,
                        uint64_t transition, int32_t lo, int32_t hi) {
  if (n == 0) return MM_OK;
  hipLaunchKernelGGL(rng_steps_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d_steps, n, seed,
                     chain_offset, transition, lo, hi);
  MM_HIP_CHECK(ctx, hipGetLastError());
  return MM_OK;
}
