// Explicit leapfrog on Euclidean-metric systems, batched over chains (gfx950 / CDNA4).
//
// Replaces, per chain and per step (reference /root/reference/src/mici):
//   LeapfrogIntegrator._step          integrators.py:170-173   A(t/2) B(t) A(t/2)
//   System.h1_flow                    systems.py:143-152       p -= dt * grad_neg_log_dens(q)
//   EuclideanMetricSystem.h2_flow     systems.py:362-363       q += dt * M^-1 p
//   EuclideanMetricSystem.dh2_dmom    systems.py:352-354       identity / diag^-1 / explicit-inverse mat-vec
// The gradient evaluated at the end of step k is carried in registers into step k+1, which is what
// the reference's state cache does (states.py:136-153).
//
// Two kernels:
//   leapfrog_elem_kernel  - separable targets x identity/diagonal metric: one lane per (chain, dim)
//                           element pair, n_steps fused, HBM-bound at n_steps = 1 (32*D B/chain-step).
//   leapfrog_mfma_kernel  - dense-precision Gaussian target and/or dense metric: 16 chains per
//                           workgroup, the [16 x D] x [D x D] products on v_mfma_f64_16x16x4_f64
//                           with the D x D operand register-resident (one 16-column slab per wave)
//                           and the chain tile exchanged through LDS once per product.
#include <cstdlib>

#include "mm_internal.h"
#include "mm_device.h"

namespace {

typedef double double4_t __attribute__((ext_vector_type(4)));

enum { T_ISO = MM_TARGET_GAUSS_ISO, T_DIAG = MM_TARGET_GAUSS_DIAG, T_DENSE = MM_TARGET_GAUSS_DENSE,
       T_POLY = MM_TARGET_POLY };
enum { M_ID = MM_METRIC_IDENTITY, M_DIAG = MM_METRIC_DIAG, M_DENSE = MM_METRIC_DENSE };

template <int TARGET>
__device__ __forceinline__ double elem_grad(double q, double tp0, double tp1) {
  if constexpr (TARGET == T_ISO) return q;
  if constexpr (TARGET == T_DIAG) return tp0 * q;           // tp0 = prec[dim]
  if constexpr (TARGET == T_POLY) return tp0 * q + tp1 * (q * q * q);  // a q + b q^3
  return 0.0;
}

// ---------------------------------------------------------------------------------------------------
// Separable targets: each lane owns VEC consecutive dims of one chain.
// COMP: a symmetric composition (SymmetricCompositionIntegrator._step, integrators.py:272-274) instead of the
// leapfrog: cf.c[k] t alternately kicks (h1_flow) and drifts (h2_flow), starting with h1 iff cf.initial_h1.
template <int TARGET, int METRIC, int VEC, bool COMP>
__device__ __forceinline__ void leapfrog_elem_body(
    double* __restrict__ pos, double* __restrict__ mom, const int8_t* __restrict__ dir,
    const double* __restrict__ step_scale, const int32_t* __restrict__ chain_steps, int32_t* __restrict__ status,
    int32_t* __restrict__ n_done, int64_t n_chains, int dim, double step_size, int n_steps,
    const double* __restrict__ tparams,
    const double* __restrict__ minv_diag, const mm_comp_coefs& cf) {
  const int vec_per_chain = dim / VEC;
  const int64_t total = n_chains * vec_per_chain;
  for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    const int64_t chain = idx / vec_per_chain;
    const int d0 = (int)(idx - chain * vec_per_chain) * VEC;
    const double t = mmdev::signed_step(dir, step_scale, chain, step_size);
    const double ht = 0.5 * t;
    const int my_steps = mmdev::chain_steps(chain_steps, chain, n_steps);
    double q[VEC], p[VEC], g[VEC], tp0[VEC], tp1[VEC], mi[VEC];
    const int64_t off = chain * dim + d0;
    if constexpr (VEC == 2) {
      const double2 qv = *reinterpret_cast<const double2*>(pos + off);
      const double2 pv = *reinterpret_cast<const double2*>(mom + off);
      q[0] = qv.x; q[1] = qv.y; p[0] = pv.x; p[1] = pv.y;
    } else {
      q[0] = pos[off]; p[0] = mom[off];
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      if constexpr (TARGET == T_DIAG) { tp0[v] = tparams[d0 + v]; tp1[v] = 0.0; }
      else if constexpr (TARGET == T_POLY) { tp0[v] = tparams[0]; tp1[v] = tparams[1]; }
      else { tp0[v] = 0.0; tp1[v] = 0.0; }
      mi[v] = (METRIC == M_DIAG) ? minv_diag[d0 + v] : 1.0;
      g[v] = elem_grad<TARGET>(q[v], tp0[v], tp1[v]);
    }
    auto run = [&](const int steps) {
      if constexpr (!COMP) {
        for (int s = 0; s < steps; ++s) {
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            p[v] -= ht * g[v];
            if constexpr (METRIC == M_DIAG) q[v] += t * (mi[v] * p[v]);
            else q[v] += t * p[v];
            g[v] = elem_grad<TARGET>(q[v], tp0[v], tp1[v]);
            p[v] -= ht * g[v];
          }
        }
      } else {
        for (int s = 0; s < steps; ++s)
          for (int k = 0; k < cf.m; ++k) {
            const double ct = cf.c[k] * t;
            const bool kick = ((k & 1) == 0) == (cf.initial_h1 != 0);
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
              if (kick) {
                p[v] -= ct * g[v];
              } else {
                if constexpr (METRIC == M_DIAG) q[v] += ct * (mi[v] * p[v]);
                else q[v] += ct * p[v];
                g[v] = elem_grad<TARGET>(q[v], tp0[v], tp1[v]);
              }
            }
          }
      }
    };
    // two instances on purpose: with a uniform trip count the loop runs on scalar control flow (a per-lane
    // count cost 30 % on the c2(i) workload)
    if (chain_steps == nullptr) run(n_steps);
    else run(my_steps);
    if constexpr (VEC == 2) {
      *reinterpret_cast<double2*>(pos + off) = make_double2(q[0], q[1]);
      *reinterpret_cast<double2*>(mom + off) = make_double2(p[0], p[1]);
    } else {
      pos[off] = q[0]; mom[off] = p[0];
    }
    if (d0 == 0) {  // explicit steps cannot fail: status 0, n_done = the chain's step count
      status[chain] = 0;
      n_done[chain] = my_steps;
    }
  }
}

template <int TARGET, int METRIC, int VEC>
__global__ __launch_bounds__(256) void leapfrog_elem_kernel(
    double* __restrict__ pos, double* __restrict__ mom, const int8_t* __restrict__ dir,
    const double* __restrict__ step_scale, const int32_t* __restrict__ chain_steps, int32_t* __restrict__ status,
    int32_t* __restrict__ n_done, int64_t n_chains, int dim, double step_size, int n_steps,
    const double* __restrict__ tparams, const double* __restrict__ minv_diag) {
  leapfrog_elem_body<TARGET, METRIC, VEC, false>(pos, mom, dir, step_scale, chain_steps, status, n_done, n_chains, dim, step_size,
      n_steps,
                                                 tparams, minv_diag, mm_comp_coefs{});
}

// The same leapfrog for SHORT launches (n_steps <= kStreamSteps, even dim): there the kernel is HBM-bound - a launch is one read
// and one write of the (q, p) state, 32 D bytes a chain-step at n_steps = 1 (the regime BASELINE.json's north_star calls
// "coalesced HBM loads of the per-chain (q,p) state"; bench.py c2i_stream) - and what limits the loop above is not arithmetic
// but bytes in flight and index arithmetic: one 16-byte load per array per trip of a grid-stride loop with a 64-bit division
// in it.  Here a thread owns kStreamItems consecutive 16-byte element pairs of the flat [n_chains x dim] arrays, all its loads are
// issued before the first use (2 x kStreamItems x 16 B in flight a lane), the chain index comes from a 32-bit division (the
// launcher falls back to the general kernel beyond 2^31 element pairs), and the stores are non-temporal (they are not read
// again by this launch).  Arithmetic identical to leapfrog_elem_body's.
constexpr int kStreamSteps = 8;
constexpr int kStreamItems = 4;
template <int TARGET, int METRIC>
__global__ __launch_bounds__(256) void leapfrog_stream_kernel(
    double* __restrict__ pos, double* __restrict__ mom, const int8_t* __restrict__ dir,
    const double* __restrict__ step_scale, const int32_t* __restrict__ chain_steps, int32_t* __restrict__ status,
    int32_t* __restrict__ n_done, unsigned total_pairs, unsigned pairs_per_chain, int dim, double step_size, int n_steps,
    const double* __restrict__ tparams, const double* __restrict__ minv_diag) {
  // item k of this thread: pair index base + k * 256 (a wave's 64 lanes read 1 KB contiguous per item and array)
  const unsigned base = (blockIdx.x * kStreamItems) * 256u + threadIdx.x;
  typedef double d2v __attribute__((ext_vector_type(2)));
  d2v qv[kStreamItems], pv[kStreamItems];
  unsigned idx[kStreamItems];
#pragma unroll
  for (int k = 0; k < kStreamItems; ++k) {
    idx[k] = base + k * 256u;
    if (idx[k] < total_pairs) {
      qv[k] = __builtin_nontemporal_load(reinterpret_cast<const d2v*>(pos) + idx[k]);
      pv[k] = __builtin_nontemporal_load(reinterpret_cast<const d2v*>(mom) + idx[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < kStreamItems; ++k) {
    if (idx[k] >= total_pairs) continue;
    const unsigned chain = idx[k] / pairs_per_chain;
    const int d0 = (int)(idx[k] - chain * pairs_per_chain) * 2;
    const double t = mmdev::signed_step(dir, step_scale, chain, step_size);
    const double ht = 0.5 * t;
    const int my_steps = mmdev::chain_steps(chain_steps, chain, n_steps);
    double q[2] = {qv[k].x, qv[k].y}, p[2] = {pv[k].x, pv[k].y}, g[2], tp0[2], tp1[2], mi[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
      if constexpr (TARGET == T_DIAG) { tp0[v] = tparams[d0 + v]; tp1[v] = 0.0; }
      else if constexpr (TARGET == T_POLY) { tp0[v] = tparams[0]; tp1[v] = tparams[1]; }
      else { tp0[v] = 0.0; tp1[v] = 0.0; }
      mi[v] = (METRIC == M_DIAG) ? minv_diag[d0 + v] : 1.0;
      g[v] = elem_grad<TARGET>(q[v], tp0[v], tp1[v]);
    }
    for (int s = 0; s < my_steps; ++s) {
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        p[v] -= ht * g[v];
        if constexpr (METRIC == M_DIAG) q[v] += t * (mi[v] * p[v]);
        else q[v] += t * p[v];
        g[v] = elem_grad<TARGET>(q[v], tp0[v], tp1[v]);
        p[v] -= ht * g[v];
      }
    }
    __builtin_nontemporal_store(d2v{q[0], q[1]}, reinterpret_cast<d2v*>(pos) + idx[k]);
    __builtin_nontemporal_store(d2v{p[0], p[1]}, reinterpret_cast<d2v*>(mom) + idx[k]);
    if (d0 == 0) {
      status[chain] = 0;
      n_done[chain] = my_steps;
    }
  }
}

template <int TARGET, int METRIC, int VEC>
__global__ __launch_bounds__(256) void composition_elem_kernel(
    double* __restrict__ pos, double* __restrict__ mom, const int8_t* __restrict__ dir,
    const double* __restrict__ step_scale, const int32_t* __restrict__ chain_steps, int32_t* __restrict__ status,
    int32_t* __restrict__ n_done, int64_t n_chains, int dim, double step_size, int n_steps,
    const double* __restrict__ tparams, const double* __restrict__ minv_diag, mm_comp_coefs cf) {
  leapfrog_elem_body<TARGET, METRIC, VEC, true>(pos, mom, dir, step_scale, chain_steps, status, n_done, n_chains, dim, step_size,
      n_steps,
                                                tparams, minv_diag, cf);
}

// ---------------------------------------------------------------------------------------------------
// Dense target and/or dense metric on FP64 MFMA.
//
// Workgroup = DP/16 waves, 16 chains.  Wave w owns output columns [16w, 16w+16).
//   C/D layout of v_mfma_f64_16x16x4_f64: lane l, reg r -> row (chain) = (l>>4) + 4r, col = l&15.
//   A operand (chain tile):  lane l supplies X[chain = l&15][k],  k = (l>>4)*(DP/4) + kk
//   B operand (matrix slab): lane l supplies W[col0 + (l&15)][k], same k          (kk = MFMA index)
// so MFMA kk accumulates the 4 k-values {kq*(DP/4)+kk : kq=0..3}; over kk = 0..DP/4-1 every k is
// covered once.  With this k-permutation a lane's A fragments are DP/4 consecutive doubles of one
// LDS row -> ds_read_b128.  LDS rows are padded by 2 doubles (one b128 access width) so the 16 rows
// read by a lane group fall in distinct 16-byte bank slots.
template <int DP, int CT>
struct MfmaCfg {
  static constexpr int NW = DP / (16 * CT);  // waves per workgroup (each owns CT 16-column tiles)
  static constexpr int KK = DP / 4;          // MFMAs per product per 16-column tile
  static constexpr int LDW = DP + 2;         // LDS row stride in doubles
  static constexpr int TILE = 16 * LDW;      // doubles per LDS chain tile
};

template <int DP>
__device__ __forceinline__ void load_slab(double (&frag)[DP / 4], const double* __restrict__ w,
                                          int dim, int col, int kq) {
#pragma unroll
  for (int kk = 0; kk < DP / 4; ++kk) {
    const int k = kq * (DP / 4) + kk;
    frag[kk] = (col < dim && k < dim) ? w[(int64_t)col * dim + k] : 0.0;
  }
}

// acc[c] = tile x slab c for the CT column tiles of this wave; each A fragment is read from LDS once
template <int DP, int CT>
__device__ __forceinline__ void tile_times_slabs(const double* __restrict__ tile,
                                                 const double (&frag)[CT][DP / 4], int lane,
                                                 double4_t (&out)[CT]) {
  const double* row = tile + (lane & 15) * (DP + 2) + (lane >> 4) * (DP / 4);
  double4_t acc0[CT], acc1[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    acc0[c] = double4_t{0.0, 0.0, 0.0, 0.0};
    acc1[c] = double4_t{0.0, 0.0, 0.0, 0.0};
  }
#pragma unroll
  for (int kk = 0; kk < DP / 4; kk += 2) {
    const double2 a = *reinterpret_cast<const double2*>(row + kk);
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      acc0[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, frag[c][kk], acc0[c], 0, 0, 0);
      acc1[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, frag[c][kk + 1], acc1[c], 0, 0, 0);
    }
  }
#pragma unroll
  for (int c = 0; c < CT; ++c) out[c] = acc0[c] + acc1[c];
}

template <int DP, int CT, int TARGET, int METRIC, bool COMP>
__device__ __forceinline__ void leapfrog_mfma_body(
    double* __restrict__ pos, double* __restrict__ mom, const int8_t* __restrict__ dir,
    const double* __restrict__ step_scale, const int32_t* __restrict__ chain_steps, int32_t* __restrict__ status,
    int32_t* __restrict__ n_done, int64_t n_chains, int dim, double step_size, int n_steps,
    const double* __restrict__ tparams,
    const double* __restrict__ minv, const mm_comp_coefs& cf) {
  using Cfg = MfmaCfg<DP, CT>;
  __shared__ __attribute__((aligned(16))) double lds[(METRIC == M_DENSE ? 3 : 2) * Cfg::TILE];
  double* qbuf = lds;                    // two q tiles (double buffered)
  double* pbuf = lds + 2 * Cfg::TILE;    // one p tile (dense metric only): with a dense target the q-tile barrier of
                                         // the gradient already separates its readers from the next publish

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int kq = lane >> 4;
  const int64_t chain0 = (int64_t)blockIdx.x * 16;
  int col[CT];  // my columns in the C layout
#pragma unroll
  for (int c = 0; c < CT; ++c) col[c] = (wave * CT + c) * 16 + (lane & 15);

  // register-resident matrix slabs (B operands)
  double pfrag[TARGET == T_DENSE ? CT : 1][TARGET == T_DENSE ? DP / 4 : 1];
  double mfrag[METRIC == M_DENSE ? CT : 1][METRIC == M_DENSE ? DP / 4 : 1];
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    if constexpr (TARGET == T_DENSE) load_slab<DP>(pfrag[c], tparams, dim, col[c], kq);
    if constexpr (METRIC == M_DENSE) load_slab<DP>(mfrag[c], minv, dim, col[c], kq);
  }

  double tp0[CT], tp1[CT], mi[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    tp0[c] = 0.0; tp1[c] = 0.0; mi[c] = 1.0;
    if constexpr (TARGET == T_DIAG) tp0[c] = (col[c] < dim) ? tparams[col[c]] : 0.0;
    if constexpr (TARGET == T_POLY) { tp0[c] = tparams[0]; tp1[c] = tparams[1]; }
    if constexpr (METRIC == M_DIAG) mi[c] = (col[c] < dim) ? minv[col[c]] : 0.0;
  }

  // chain state in the C layout: q[c][r], p[c][r] <-> chain (lane>>4)+4r, column col[c]
  double q[CT][4], p[CT][4], g[CT][4], t[4], ht[4];
  bool live[CT][4];
  int my_steps[4];  // per-chain step counts: a chain that is done has its time step zeroed, so the rest of the
                    // workgroup's steps leave it exactly where it is
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int64_t chain = chain0 + (lane >> 4) + 4 * r;
    t[r] = (chain < n_chains) ? mmdev::signed_step(dir, step_scale, chain, step_size) : 0.0;
    ht[r] = 0.5 * t[r];
    my_steps[r] = (chain < n_chains) ? mmdev::chain_steps(chain_steps, chain, n_steps) : 0;
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      live[c][r] = chain < n_chains && col[c] < dim;
      q[c][r] = live[c][r] ? pos[chain * dim + col[c]] : 0.0;
      p[c][r] = live[c][r] ? mom[chain * dim + col[c]] : 0.0;
    }
  }

  auto publish = [&](double* tile, const double (&x)[CT][4]) {
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) tile[((lane >> 4) + 4 * r) * Cfg::LDW + col[c]] = x[c][r];
  };
  auto gradient = [&](int s) {
    if constexpr (TARGET == T_DENSE) {
      double* tile = qbuf + (s & 1) * Cfg::TILE;
      publish(tile, q);
      __syncthreads();
      double4_t acc[CT];
      tile_times_slabs<DP, CT>(tile, pfrag, lane, acc);
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) g[c][r] = acc[c][r];
    } else {
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) g[c][r] = elem_grad<TARGET>(q[c][r], tp0[c], tp1[c]);
    }
  };

  // q += tt M^-1 p with the product on buffer parity s (dense metric) or elementwise
  auto drift = [&](const double (&tt)[4], int s) {
    if constexpr (METRIC == M_DENSE) {
      double* tile = pbuf;
      (void)s;
      if constexpr (TARGET != T_DENSE) __syncthreads();  // no gradient barrier since the last read of this tile
      publish(tile, p);
      __syncthreads();
      double4_t v[CT];
      tile_times_slabs<DP, CT>(tile, mfrag, lane, v);
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) q[c][r] += tt[r] * v[c][r];
    } else {
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if constexpr (METRIC == M_DIAG) q[c][r] += tt[r] * (mi[c] * p[c][r]);
          else q[c][r] += tt[r] * p[c][r];
        }
    }
  };

  gradient(1);  // g(q0); uses buffer 1 so that step 0 starts on buffer 0
  auto retire = [&](int s) {  // uniform branch: free when the state has no per-chain counts
    if (chain_steps) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (s >= my_steps[r]) {
          t[r] = 0.0;
          ht[r] = 0.0;
        }
    }
  };
  if constexpr (!COMP) {
    for (int s = 0; s < n_steps; ++s) {
      retire(s);
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) p[c][r] -= ht[r] * g[c][r];
      drift(t, s);
      gradient(s);
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) p[c][r] -= ht[r] * g[c][r];
    }
  } else {
    // SymmetricCompositionIntegrator._step (integrators.py:272-274); nd counts the h2 flows done so far and
    // selects the LDS buffer, so consecutive products never share a tile
    int nd = 0;
    for (int s = 0; s < n_steps; ++s) {
      retire(s);
      for (int k = 0; k < cf.m; ++k) {
        double ct[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) ct[r] = cf.c[k] * t[r];
        if (((k & 1) == 0) == (cf.initial_h1 != 0)) {
#pragma unroll
          for (int c = 0; c < CT; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) p[c][r] -= ct[r] * g[c][r];
        } else {
          drift(ct, nd);
          gradient(nd);
          ++nd;
        }
      }
    }
  }

#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (live[c][r]) {
        const int64_t chain = chain0 + (lane >> 4) + 4 * r;
        pos[chain * dim + col[c]] = q[c][r];
        mom[chain * dim + col[c]] = p[c][r];
        if (col[c] == 0) {  // explicit steps cannot fail
          status[chain] = 0;
          n_done[chain] = my_steps[r];
        }
      }
    }
}

template <int DP, int CT, int TARGET, int METRIC>
__global__ __launch_bounds__(DP * 4 / CT) void leapfrog_mfma_kernel(
    double* __restrict__ pos, double* __restrict__ mom, const int8_t* __restrict__ dir,
    const double* __restrict__ step_scale, const int32_t* __restrict__ chain_steps, int32_t* __restrict__ status,
    int32_t* __restrict__ n_done, int64_t n_chains, int dim, double step_size, int n_steps,
    const double* __restrict__ tparams, const double* __restrict__ minv) {
  leapfrog_mfma_body<DP, CT, TARGET, METRIC, false>(pos, mom, dir, step_scale, chain_steps, status, n_done, n_chains, dim, step_size,
      n_steps,
                                                    tparams, minv, mm_comp_coefs{});
}

template <int DP, int CT, int TARGET, int METRIC>
__global__ __launch_bounds__(DP * 4 / CT) void composition_mfma_kernel(
    double* __restrict__ pos, double* __restrict__ mom, const int8_t* __restrict__ dir,
    const double* __restrict__ step_scale, const int32_t* __restrict__ chain_steps, int32_t* __restrict__ status,
    int32_t* __restrict__ n_done, int64_t n_chains, int dim, double step_size, int n_steps,
    const double* __restrict__ tparams, const double* __restrict__ minv, mm_comp_coefs cf) {
  leapfrog_mfma_body<DP, CT, TARGET, METRIC, true>(pos, mom, dir, step_scale, chain_steps, status, n_done, n_chains, dim, step_size,
      n_steps,
                                                   tparams, minv, cf);
}

template <int TARGET, int METRIC>
int launch_elem(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int n_steps, const mm_comp_coefs* cf) {
  const int dim = s->dim;
  const bool vec2 = (dim % 2 == 0);
  const int64_t total = s->n * (vec2 ? dim / 2 : dim);
  int64_t blocks = (total + 255) / 256;
  const int64_t cap = (int64_t)ctx->n_cu * 8;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  if (!cf && vec2 && n_steps <= kStreamSteps && total < (int64_t)1 << 31) {
    const unsigned sblocks = (unsigned)((total + 256 * kStreamItems - 1) / (256 * kStreamItems));
    hipLaunchKernelGGL((leapfrog_stream_kernel<TARGET, METRIC>), dim3(sblocks), dim3(256), 0, ctx->stream, s->d_pos, s->d_mom,
                       s->d_dir, s->d_step_scale, s->d_chain_steps, s->d_status, s->d_n_done, (unsigned)total,
                       (unsigned)(dim / 2), dim, h, n_steps, m->d_target_params, m->d_metric_inv);
    MM_HIP_CHECK(ctx, hipGetLastError());
    return MM_OK;
  }
  if (cf && vec2)
    hipLaunchKernelGGL((composition_elem_kernel<TARGET, METRIC, 2>), dim3((unsigned)blocks), dim3(256),
                       0, ctx->stream, s->d_pos, s->d_mom, s->d_dir, s->d_step_scale, s->d_chain_steps, s->d_status, s->d_n_done, s->n, dim, h, n_steps,
                       m->d_target_params, m->d_metric_inv, *cf);
  else if (cf)
    hipLaunchKernelGGL((composition_elem_kernel<TARGET, METRIC, 1>), dim3((unsigned)blocks), dim3(256),
                       0, ctx->stream, s->d_pos, s->d_mom, s->d_dir, s->d_step_scale, s->d_chain_steps, s->d_status, s->d_n_done, s->n, dim, h, n_steps,
                       m->d_target_params, m->d_metric_inv, *cf);
  else if (vec2)
    hipLaunchKernelGGL((leapfrog_elem_kernel<TARGET, METRIC, 2>), dim3((unsigned)blocks), dim3(256),
                       0, ctx->stream, s->d_pos, s->d_mom, s->d_dir, s->d_step_scale, s->d_chain_steps, s->d_status, s->d_n_done, s->n, dim, h, n_steps,
                       m->d_target_params, m->d_metric_inv);
  else
    hipLaunchKernelGGL((leapfrog_elem_kernel<TARGET, METRIC, 1>), dim3((unsigned)blocks), dim3(256),
                       0, ctx->stream, s->d_pos, s->d_mom, s->d_dir, s->d_step_scale, s->d_chain_steps, s->d_status, s->d_n_done, s->n, dim, h, n_steps,
                       m->d_target_params, m->d_metric_inv);
  MM_HIP_CHECK(ctx, hipGetLastError());
  return MM_OK;
}

template <int DP, int CT, int TARGET, int METRIC>
int launch_mfma_dp(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int n_steps, const mm_comp_coefs* cf) {
  const unsigned blocks = (unsigned)((s->n + 15) / 16);
  if (cf)
    hipLaunchKernelGGL((composition_mfma_kernel<DP, CT, TARGET, METRIC>), dim3(blocks),
                       dim3(DP * 4 / CT), 0, ctx->stream, s->d_pos, s->d_mom, s->d_dir, s->d_step_scale, s->d_chain_steps, s->d_status, s->d_n_done, s->n, s->dim,
                       h, n_steps, m->d_target_params, m->d_metric_inv, *cf);
  else
  hipLaunchKernelGGL((leapfrog_mfma_kernel<DP, CT, TARGET, METRIC>), dim3(blocks),
                     dim3(DP * 4 / CT), 0, ctx->stream, s->d_pos, s->d_mom, s->d_dir, s->d_step_scale, s->d_chain_steps, s->d_status, s->d_n_done, s->n, s->dim, h,
                     n_steps, m->d_target_params, m->d_metric_inv);
  MM_HIP_CHECK(ctx, hipGetLastError());
  return MM_OK;
}

// columns per wave: 16 (CT = 1, two waves per SIMD at D = 128) or 32 (CT = 2, one wave per SIMD, each
// A fragment read once for two column tiles, half the waves at the per-product barrier)
int mfma_ct() {
  static const int ct = [] {
    const char* e = getenv("MICI_AMD_MFMA_CT");
    return (e && e[0] == '2') ? 2 : 1;  // measured equal on c2(iii), CT = 1 faster with a dense metric
  }();
  return ct;
}

template <int TARGET, int METRIC>
int launch_mfma(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int n_steps, const mm_comp_coefs* cf) {
  const int dim = s->dim;
  if (dim <= 16) return launch_mfma_dp<16, 1, TARGET, METRIC>(ctx, m, s, h, n_steps, cf);
  if (dim <= 32) return launch_mfma_dp<32, 1, TARGET, METRIC>(ctx, m, s, h, n_steps, cf);
  if (dim <= 64) return launch_mfma_dp<64, 1, TARGET, METRIC>(ctx, m, s, h, n_steps, cf);
  if (dim <= 128) {
    if (mfma_ct() == 2) return launch_mfma_dp<128, 2, TARGET, METRIC>(ctx, m, s, h, n_steps, cf);
    return launch_mfma_dp<128, 1, TARGET, METRIC>(ctx, m, s, h, n_steps, cf);
  }
  mm_set_error(ctx, "mm_leapfrog_euclid: dense target/metric kernels support dim <= 128");
  return MM_ERR_UNSUPPORTED;
}

}  // namespace

// cf == nullptr: leapfrog; otherwise the symmetric composition with those coefficients
static int launch_euclid(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int n_steps,
                         const mm_comp_coefs* cf) {
  const int T = m->target, M = m->metric_kind;
#define MM_CASE_ELEM(TT, MM_) \
  if (T == TT && M == MM_) return launch_elem<TT, MM_>(ctx, m, s, h, n_steps, cf);
#define MM_CASE_MFMA(TT, MM_) \
  if (T == TT && M == MM_) return launch_mfma<TT, MM_>(ctx, m, s, h, n_steps, cf);
  MM_CASE_ELEM(T_ISO, M_ID)
  MM_CASE_ELEM(T_ISO, M_DIAG)
  MM_CASE_ELEM(T_DIAG, M_ID)
  MM_CASE_ELEM(T_DIAG, M_DIAG)
  MM_CASE_ELEM(T_POLY, M_ID)
  MM_CASE_ELEM(T_POLY, M_DIAG)
  MM_CASE_MFMA(T_DENSE, M_ID)
  MM_CASE_MFMA(T_DENSE, M_DIAG)
  MM_CASE_MFMA(T_DENSE, M_DENSE)
  MM_CASE_MFMA(T_ISO, M_DENSE)
  MM_CASE_MFMA(T_DIAG, M_DENSE)
  MM_CASE_MFMA(T_POLY, M_DENSE)
#undef MM_CASE_ELEM
#undef MM_CASE_MFMA
  return -100;  // not a separable / dense-Gaussian target: caller falls through to the generic kernel
}

int mm_launch_leapfrog_euclid(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int n_steps) {
  return launch_euclid(ctx, m, s, h, n_steps, nullptr);
}

int mm_launch_composition_euclid(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int n_steps,
                                 const mm_comp_coefs& cf) {
  return launch_euclid(ctx, m, s, h, n_steps, &cf);
}
