// Micro-benchmark: instruction LATENCIES that bound the per-column critical path of the dense-metric
// sweep kernels on gfx950 (one chain's factorisation is a chain of dependent publish -> barrier ->
// read -> reciprocal -> fma steps).  Everything is timed with s_memtime around long dependent chains
// in ONE workgroup per CU; results in shader-clock cycles per operation (s_memtime ticks are
// converted with a calibration loop of dependent v_add_u32 = 1 issue slot each... reported raw too).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_latency.hip -o gpurun_out/ubench_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ long long now() { return __builtin_readcyclecounter(); }

// 0 dependent v_fma_f64; 1 dependent v_rcp_f64; 2 dependent full-precision reciprocal (rcp + 4 fma);
// 3 LDS pointer chase ds_read_b64; 4 publish round trip: ds_write, waitcnt, barrier, ds_read (dependent);
// 5 independent v_fma_f64 x8 (issue interval); 6 dependent v_add_u32; 7 dependent v_mul_f64;
// 8 barrier only; 9 DPP mov dependent; 10 dependent v_cndmask+fma pair
__global__ void lat_kernel(int op, int iters, double seed, long long* ticks, double* sink) {
  __shared__ double buf[1024];
  const int tid = threadIdx.x;
  for (int i = tid; i < 1024; i += blockDim.x) buf[i] = (double)((i * 37 + 11) & 1023);
  __syncthreads();
  double x = seed + 1e-12 * tid, acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = seed * (i + 1);
  unsigned u = tid;
  const long long t0 = now();
  if (op == 0) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 16; ++j) x = __builtin_fma(x, 0.999999, 1e-9);
    }
  } else if (op == 1) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 16; ++j) x = __builtin_amdgcn_rcp(x);
    }
  } else if (op == 2) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        double r = __builtin_amdgcn_rcp(x);
        double e = __builtin_fma(-x, r, 1.0);
        r = __builtin_fma(r, e, r);
        e = __builtin_fma(-x, r, 1.0);
        x = __builtin_fma(r, e, r);
      }
    }
  } else if (op == 3) {
    int idx = tid & 1023;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 16; ++j) idx = (int)buf[idx];
    }
    x = idx;
  } else if (op == 4) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (tid == ((i + j) & 63)) buf[j] = x;
        __syncthreads();
        x = buf[j] + 1e-9;
      }
    }
  } else if (op == 5) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j & 7] = __builtin_fma(acc[j & 7], 0.999999, 1e-9);
    }
    for (int i = 0; i < 8; ++i) x += acc[i];
  } else if (op == 6) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        u = u * 3u + 1u;
        asm volatile("" : "+v"(u));
      }
    }
    x = u;
  } else if (op == 7) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 16; ++j) x = x * 0.9999999;
    }
  } else if (op == 8) {
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 16; ++j) __syncthreads();
    }
  } else if (op == 9) {
    int v = tid;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 16; ++j) v = __builtin_amdgcn_update_dpp(v, v, 0x121 /*row_ror:1*/, 0xf, 0xf, false) + 1;
    }
    x = v;
  } else if (op == 10) {
    // 8 independent LDS reads issued together, then consumed (pipelined LDS)
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int b = ((int)x) & 511;
        double s = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += buf[b + k * 17];
        x = s * 1e-3;
      }
    }
  }
  const long long t1 = now();
  if (tid == 0) ticks[blockIdx.x] = t1 - t0;
  sink[blockIdx.x * blockDim.x + tid] = x + u;
}

int main() {
  const char* names[] = {"dep v_fma_f64", "dep v_rcp_f64", "dep full rcp (rcp+4fma)", "LDS pointer chase b64",
                         "write+barrier+read round trip", "8 indep v_fma_f64 (per fma)", "dep v_mul_lo+add u32 (2 ops)",
                         "dep v_mul_f64", "barrier only", "dep DPP mov+add", "8 LDS reads + 8 adds + mul (per group)"};
  const int per_iter[] = {16, 16, 16, 16, 16, 16, 16, 16, 16, 16, 2};
  long long* d_t;
  double* d_s;
  hipMalloc(&d_t, 1024 * sizeof(long long));
  hipMalloc(&d_s, 1024 * 1024 * sizeof(double));
  const int iters = 2000;
  // s_memtime rate against wall clock: time a long kernel with events
  for (int threads : {64, 128, 256, 512}) {
    printf("--- %d threads (%d waves) per workgroup, 1 workgroup\n", threads, threads / 64);
    for (int op = 0; op <= 10; ++op) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      hipLaunchKernelGGL(lat_kernel, dim3(1), dim3(threads), 0, 0, op, 10, 1.37, d_t, d_s);
      hipEventRecord(e0);
      hipLaunchKernelGGL(lat_kernel, dim3(1), dim3(threads), 0, 0, op, iters, 1.37, d_t, d_s);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      long long t;
      hipMemcpy(&t, d_t, sizeof(t), hipMemcpyDeviceToHost);
      const double n = (double)iters * per_iter[op];
      printf("%-44s %8.1f ticks/op   %8.1f ns/op (event)\n", names[op], t / n, ms * 1e6 / n);
    }
  }
  return 0;
}
