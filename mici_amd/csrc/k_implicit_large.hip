// Implicit leapfrog on dense-metric Riemannian systems for 64 < D <= 264 (BASELINE config c4:
// D = 256): one 1024-thread workgroup (a whole CU) per chain.  gfx950 / CDNA4.
//
// Same reference arithmetic as k_implicit.hip (the step itself is implicit_core.h); what changes is
// where the D x D metric lives.  512 KB of fp64 does not fit one CU's LDS (160 KB), but the symmetric
// half (D(D+1)/2 * 8 B = 263 KB at D = 256) fits its 512 KB register file:
//   * the 1024 threads (16 waves, 4 per SIMD, 128 VGPRs each) form the lower triangle of a 44 x 44
//     grid (990 tile owners); thread (ti >= tj) owns the 6 x 6 block-cyclic tile
//     {(ti + 44 a, tj + 44 b)} = 72 VGPRs.  44 * 6 = 264 >= 256.
//   * the symmetric sweep operator keeps the matrix symmetric, so the mirrored tiles are never needed:
//     step k publishes column k (from the tiles of grid column k%44 and, transposed, of grid row k%44)
//     into LDS; every thread then applies at(a, b) -= m[a] * c[b] (36 v_fma_f64) from 12 LDS operands.
//   * M^-1 v: each tile contributes to 6 "row" and (off-diagonal tiles) 6 "column" partial sums, laid
//     out in LDS so that every output element has exactly 44 private slots -> deterministic reduction.
// Throughput is one chain per CU, 256 chains in flight per GPU.
#include "implicit_core.h"

namespace {

using namespace mmdev;
using namespace mmimp;

constexpr int PG = 31;             // process-grid side
constexpr int TS = 9;              // tile side
constexpr int DP = PG * TS;        // 279: padded dimension
constexpr int NT = 512;            // threads per workgroup (8 waves, 2 per SIMD -> 256 VGPRs each)
constexpr int NTILE = PG * (PG + 1) / 2;  // 496 tile-owning threads
constexpr int GS = 10;             // doubles reserved per grid group in a permuted LDS vector
constexpr int PV = PG * GS;        // permuted vector length
constexpr int SLOTS = 33;          // 31 partial-sum slots per output element (+2 pad vs bank conflicts)

__device__ __forceinline__ int ppos(int i) { return (i % PG) * GS + i / PG; }

// Launder a lane-varying index so that address arithmetic derived from it is recomputed where it is
// used instead of being hoisted out of the step loop into long-lived VGPRs (the register file is full
// of metric tiles; a few integer ops per use are free).
__device__ __forceinline__ int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}

struct BlockLds {
  double* col0;
  double* col1;
  double* vin;
  double* part;  // [PG][TS][SLOTS]
  double* nat;   // natural order [DP + pad]
  double* aux;   // natural order [DP + pad]
  double* red;   // [16]
  double* stash; // [SL_COUNT][288] per-thread flat state of the step (keeps it out of VGPRs)
  double* trow;  // [TS][NT] the last tile row of every thread (register relief, see BlockBackend::at)
};
constexpr int kLdsDoubles = 3 * PV + PG * TS * SLOTS + 2 * 288 + 16 + SL_COUNT * 288 + TS * NT;

__device__ __forceinline__ double block_reduce(double v, int kind_max, double* red) {
  // kind_max: 0 sum, 1 NaN-propagating max.  Uniform result; two barriers.
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  v = kind_max ? wave_max(v) : wave_sum(v);
  if (lane == 0) red[wave] = v;
  __syncthreads();
  double r = red[0];
#pragma unroll
  for (int w = 1; w < NT / 64; ++w) r = kind_max ? nanmax(r, red[w]) : r + red[w];
  __syncthreads();
  return r;
}

template <int RMETRIC>
struct BlockBackend {
  // Tile storage: rows 0..TS-2 in registers (144 VGPRs), row TS-1 in LDS.  The compiler could not
  // keep all 81 doubles plus the sweep operands inside the 256-VGPR budget of a wave here and spilled
  // tile entries to scratch (L2-bound: measured 8x slowdown of the sweep); parking one row in LDS by
  // hand removes the spills at the price of 18 LDS accesses per sweep step.
  double Treg[TS - 1][TS];
  __device__ __forceinline__ double& at(int a, int b) {
    return a < TS - 1 ? Treg[a][b] : w.trow[b * NT + tid];
  }
  int dim, tid, ti, tj, target;
  bool tile;  // this thread owns a tile
  BlockLds w;
  const double* base;  // global (L2-resident) base matrix of the rank-one metric, zero-padded DP x DP
  const double* tparams;

  // flat state only exists for tid < DP; the other threads share one dummy cell per slot
  __device__ __forceinline__ double& slot(int i) { return w.stash[i * 288 + (tid < DP ? tid : 287)]; }

  __device__ __forceinline__ double norm(double x, int kind) {
    const double a = tid < dim ? x : 0.0;
    if (kind == MM_NORM_LINF) return block_reduce(fabs(a), 1, w.red);
    return sqrt(block_reduce(a * a, 0, w.red));
  }

  // metric_func(x) into the tiles; returns false if any entry is not finite
  __device__ __forceinline__ bool build(double x) {
    if (tid < DP) w.vin[ppos(tid)] = (tid < dim) ? x : 0.0;
    __syncthreads();
    double chk = 0.0;
    const int ti = opaque(this->ti), tj = opaque(this->tj);
    {  // every thread builds a tile (threads >= 496 duplicate tile (30,30)): the tiles are then fully
       // re-defined here, i.e. dead before this point, which frees their registers for scalar work
      // `base` is the rank-one metric's base matrix zero-padded to DP x DP on the host, and x is 0 on the
      // padding, so the closed form is exactly 0 on padded entries; the padded diagonal is set to 1 below.
      double qc[TS];
#pragma unroll
      for (int b = 0; b < TS; ++b) qc[b] = w.vin[tj * GS + b];
      const double inv_d = 1.0 / (double)dim;
#pragma unroll
      for (int a = 0; a < TS; ++a) {
        // one tile row at a time: 9 loads with immediate offsets from one row pointer
        const double qa = w.vin[ti * GS + a] * inv_d;
        if constexpr (RMETRIC == MM_RMETRIC_RANK1) {
          const double* brow = base + (int64_t)(ti + PG * a) * DP + tj;
#pragma unroll
          for (int b = 0; b < TS; ++b) at(a, b) = __builtin_fma(qa, qc[b], brow[PG * b]);
        } else {
#pragma unroll
          for (int b = 0; b < TS; ++b) at(a, b) = 0.0;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (ti == tj) {
#pragma unroll
        for (int a = 0; a < TS; ++a) {
          if constexpr (RMETRIC == MM_RMETRIC_DIAGQUAD) at(a, a) = __builtin_fma(qc[a], qc[a], 1.0);
          if (ti + PG * a >= dim) at(a, a) = 1.0;
        }
      }
#pragma unroll
      for (int a = 0; a < TS; ++a)
#pragma unroll
        for (int b = 0; b < TS; ++b) chk = __builtin_fma(at(a, b), 0.0, chk);
    }
    // NaN in chk <=> some entry is inf/NaN ("Array is not finite.", matrices.py:211-215)
    const double bad = block_reduce(chk == 0.0 ? 0.0 : 1.0, 0, w.red);
    return bad == 0.0;
  }

  // symmetric sweep: tiles <- M^-1 (lower-triangular tile set); false if a pivot is not > 0
  template <bool LOGDET, bool CHOLVEC>
  __device__ __forceinline__ bool sweep(double* logdet, double* chol_y) {
    bool ok = true;
    double ld = 0.0, y = 0.0;
#pragma unroll
    for (int kb = 0; kb < TS; ++kb) {
#pragma unroll 1
      for (int kt = 0; kt < PG; ++kt) {
        const int k = kb * PG + kt;
        const int ti = opaque(this->ti), tj = opaque(this->tj);
        double* col = (k & 1) ? w.col1 : w.col0;
        if (tile) {
          if (tj == kt) {  // grid column kt: rows ti + 31 a
#pragma unroll
            for (int a = 0; a < TS; ++a) col[ti * GS + a] = at(a, kb);
          } else if (ti == kt) {  // grid row kt (tj < kt): entries (k, tj + 31 b) = column k by symmetry
#pragma unroll
            for (int b = 0; b < TS; ++b) col[tj * GS + b] = at(kb, b);
          }
        }
        __syncthreads();
        const double piv = col[kt * GS + kb];
        ok = ok && (piv > 0.0);
        const double d = fast_rcp(piv);
        if constexpr (LOGDET) ld += log(piv);
        if constexpr (CHOLVEC) {
          const double rs = 1.0 / sqrt(piv);
          if (tid >= k && tid < dim) y += (col[ppos(tid)] * rs) * w.aux[k];
        }
        {
          double ac[TS];
#pragma unroll
          for (int b = 0; b < TS; ++b) ac[b] = col[tj * GS + b];
          if (tj == kt) ac[kb] = piv - 1.0;
          // one tile row at a time: only one row multiplier is live (register budget: 256 / wave)
#pragma unroll
          for (int a = 0; a < TS; ++a) {
            double m = col[ti * GS + a] * d;
            if (a == kb && ti == kt) m = 1.0 - d;
#pragma unroll
            for (int b = 0; b < TS; ++b) at(a, b) = __builtin_fma(-m, ac[b], at(a, b));
            __builtin_amdgcn_sched_barrier(0);
          }
          if (ti == kt && tj == kt) at(kb, kb) -= 2.0;
        }
        // the next step publishes into the other buffer; the barrier of that step orders reuse
      }
    }
#pragma unroll
    for (int a = 0; a < TS; ++a)
#pragma unroll
      for (int b = 0; b < TS; ++b) at(a, b) = -at(a, b);
    __syncthreads();
    if constexpr (LOGDET) *logdet = ld;
    if constexpr (CHOLVEC) *chol_y = y;
    return ok;
  }

  __device__ __forceinline__ bool build_and_invert(double x) {
    bool ok = build(x);
    ok = sweep<false, false>(nullptr, nullptr) && ok;
    return ok;
  }

  __device__ __forceinline__ double matvec(double v) {
    const int tid = opaque(this->tid), ti = opaque(this->ti), tj = opaque(this->tj);
    if (tid < DP) w.vin[ppos(tid)] = (tid < dim) ? v : 0.0;
    __syncthreads();
    {
      double xr[TS], xc[TS];
#pragma unroll
      for (int a = 0; a < TS; ++a) {
        xr[a] = w.vin[ti * GS + a];
        xc[a] = w.vin[tj * GS + a];
      }
      // row partials: y[ti + 31 a] += sum_b at(a, b) x[tj + 31 b]  -> slot tj of group ti
#pragma unroll
      for (int a = 0; a < TS; ++a) {
        double s = 0.0;
#pragma unroll
        for (int b = 0; b < TS; ++b) s = __builtin_fma(at(a, b), xc[b], s);
        if (tile) w.part[(ti * TS + a) * SLOTS + tj] = s;
      }
      // column partials of off-diagonal tiles (the mirrored tile): y[tj + 31 b] += sum_a at(a, b) x[ti + 31 a]
#pragma unroll
      for (int b = 0; b < TS; ++b) {
        double s = 0.0;
#pragma unroll
        for (int a = 0; a < TS; ++a) s = __builtin_fma(at(a, b), xr[a], s);
        if (tile && ti != tj) w.part[(tj * TS + b) * SLOTS + ti] = s;
      }
    }
    __syncthreads();
    double y = 0.0;
    if (tid < DP) {
      const double* src = w.part + ((tid % PG) * TS + tid / PG) * SLOTS;
#pragma unroll
      for (int sl = 0; sl < PG; ++sl) y += src[sl];
    }
    __syncthreads();
    return tid < dim ? y : 0.0;
  }

  __device__ __forceinline__ double diag() {
    if (tile && ti == tj) {
#pragma unroll
      for (int a = 0; a < TS; ++a) w.vin[ti * GS + a] = at(a, a);
    }
    __syncthreads();
    const double y = (tid < dim) ? w.vin[ppos(tid)] : 0.0;
    __syncthreads();
    return y;
  }

  __device__ __forceinline__ double half_vjp_inv(double q) {
    if constexpr (RMETRIC == MM_RMETRIC_RANK1) return matvec(q) / (double)dim;
    else return q * diag();
  }

  // dense metric: grad_quadratic_form_inv(p) = -(M^-1 p)(M^-1 p)^T   (matrices.py:1179-1181)
  __device__ __forceinline__ double dh2_dpos(double p, double q) {
    const double u = matvec(p);
    if constexpr (RMETRIC == MM_RMETRIC_RANK1) {
      const double uq = block_reduce(tid < dim ? u * q : 0.0, 0, w.red);
      return -(u * uq) / (double)dim;
    } else {
      return -q * (u * u);
    }
  }

  __device__ __forceinline__ double grad(double q) {
    if (tid < 288) w.nat[tid] = (tid < dim) ? q : 0.0;
    __syncthreads();
    const TargetAux aux = target_prepare<false>(target, w.nat, dim, tparams, threadIdx.x & 63);
    const double g = (tid < dim) ? target_grad_elem<false>(target, aux, w.nat, tid, dim, tparams) : 0.0;
    __syncthreads();
    return g;
  }

  __device__ __forceinline__ double neg_log_dens_elem(double q) {
    if (tid < 288) w.nat[tid] = (tid < dim) ? q : 0.0;
    __syncthreads();
    const TargetAux aux = target_prepare<false>(target, w.nat, dim, tparams, threadIdx.x & 63);
    const double e = (tid < dim) ? target_nld_elem<false>(target, aux, w.nat, tid, dim, tparams) : 0.0;
    __syncthreads();
    return e;
  }
};

template <int RMETRIC>
__device__ __forceinline__ void init_backend(BlockBackend<RMETRIC>& bk, const ImplicitArgs& A,
                                             double* lds) {
  const int tid = threadIdx.x;
  bk.dim = A.dim;
  bk.tid = tid;
  bk.target = A.target;
  bk.tile = tid < NTILE;
  int ti = (int)((sqrtf(8.0f * (float)tid + 1.0f) - 1.0f) * 0.5f);
  while (ti * (ti + 1) / 2 > tid) --ti;
  while ((ti + 1) * (ti + 2) / 2 <= tid) ++ti;
  bk.ti = bk.tile ? ti : PG - 1;
  bk.tj = bk.tile ? tid - ti * (ti + 1) / 2 : PG - 1;
  bk.w.col0 = lds;
  bk.w.col1 = lds + PV;
  bk.w.vin = lds + 2 * PV;
  bk.w.part = lds + 3 * PV;
  bk.w.nat = bk.w.part + PG * TS * SLOTS;
  bk.w.aux = bk.w.nat + 288;
  bk.w.red = bk.w.aux + 288;
  bk.w.stash = bk.w.red + 16;
  bk.w.trow = bk.w.stash + SL_COUNT * 288;
  bk.base = A.rparams;
  bk.tparams = A.tparams;
}

template <int RMETRIC>
__global__ __launch_bounds__(NT, 2) void implicit_large_kernel(ImplicitArgs A) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int64_t chain = blockIdx.x;
  BlockBackend<RMETRIC> bk;
  init_backend(bk, A, lds);
  const int dim = A.dim, tid = threadIdx.x;
  const bool act = tid < dim;
  double q = act ? A.pos[chain * dim + tid] : 0.0;
  double p = act ? A.mom[chain * dim + tid] : 0.0;
  const double t = (double)A.dir[chain] * A.step_size;
  bk.slot(SL_Q) = q;
  bk.slot(SL_P) = p;
  const ChainResult r = implicit_leapfrog_chain(bk, t, A.n_steps, A.opts);
  q = bk.slot(SL_Q);
  p = bk.slot(SL_P);
  if (act) {
    A.pos[chain * dim + tid] = q;
    A.mom[chain * dim + tid] = p;
  }
  if (tid == 0) {
    A.status[chain] = r.status;
    A.n_done[chain] = r.done;
    add_counters(A.counters, r);
  }
}

template <int RMETRIC, int OP>
__global__ __launch_bounds__(NT, 2) void riemann_aux_large_kernel(ImplicitArgs A) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int64_t chain = blockIdx.x;
  BlockBackend<RMETRIC> bk;
  init_backend(bk, A, lds);
  const int dim = A.dim, tid = threadIdx.x;
  const bool act = tid < dim;
  const double q = act ? A.pos[chain * dim + tid] : 0.0;
  const double p = act ? A.mom[chain * dim + tid] : 0.0;
  const double nan = __longlong_as_double(0x7ff8000000000000LL);
  bool ok = bk.build(q);
  if constexpr (OP == 0) {
    double logdet;
    ok = bk.template sweep<true, false>(&logdet, nullptr) && ok;
    const double u = bk.matvec(p);
    const double e = bk.neg_log_dens_elem(q) + (act ? 0.5 * p * u : 0.0);
    const double h = block_reduce(e, 0, bk.w.red) + 0.5 * logdet;
    if (tid == 0) A.out[chain] = ok ? h : nan;
  } else if constexpr (OP == 1) {
    ok = bk.template sweep<false, false>(nullptr, nullptr) && ok;
    const double u = bk.matvec(p);
    if (act) A.out[chain * dim + tid] = ok ? u : nan;
  } else {
    if (tid < 288) bk.w.aux[tid] = act ? A.z[chain * dim + tid] : 0.0;
    __syncthreads();
    double y;
    ok = bk.template sweep<false, true>(nullptr, &y) && ok;
    if (act) A.mom[chain * dim + tid] = ok ? y : nan;
  }
}

ImplicitArgs make_args(const mm_model* m, mm_state* s) {
  ImplicitArgs a{};
  a.pos = s->d_pos;
  a.mom = s->d_mom;
  a.dir = s->d_dir;
  a.status = s->d_status;
  a.n_done = s->d_n_done;
  a.n_chains = s->n;
  a.dim = s->dim;
  a.target = m->target;
  a.tparams = m->d_target_params;
  a.rparams = m->d_rmetric_padded;  // zero-padded to 279 x 279 for the rank-one metric
  return a;
}

template <class K>
int launch(mm_ctx* ctx, K kernel, const ImplicitArgs& a) {
  const size_t lds = kLdsDoubles * sizeof(double);
  MM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kernel, dim3((unsigned)a.n_chains), dim3(NT), lds, ctx->stream, a);
  MM_HIP_CHECK(ctx, hipGetLastError());
  return MM_OK;
}

}  // namespace

int mm_launch_implicit_large(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int n_steps,
                             const mm_fp_opts& opts, mm_counters* d_counters) {
  if (m->dim > DP) {
    mm_set_error(ctx, "dense-Riemannian kernels support dim <= 279 (register-resident metric)");
    return MM_ERR_UNSUPPORTED;
  }
  ImplicitArgs a = make_args(m, s);
  a.step_size = h;
  a.n_steps = n_steps;
  a.opts = opts;
  a.counters = d_counters;
  if (m->rmetric == MM_RMETRIC_RANK1) return launch(ctx, implicit_large_kernel<MM_RMETRIC_RANK1>, a);
  return launch(ctx, implicit_large_kernel<MM_RMETRIC_DIAGQUAD>, a);
}

int mm_launch_riemann_aux_large(mm_ctx* ctx, const mm_model* m, mm_state* s, int op, double* d_out,
                                const double* d_z) {
  if (m->dim > DP) {
    mm_set_error(ctx, "dense-Riemannian kernels support dim <= 279 (register-resident metric)");
    return MM_ERR_UNSUPPORTED;
  }
  ImplicitArgs a = make_args(m, s);
  a.out = d_out;
  a.z = d_z;
  const bool r1 = m->rmetric == MM_RMETRIC_RANK1;
  if (op == 0)
    return r1 ? launch(ctx, riemann_aux_large_kernel<MM_RMETRIC_RANK1, 0>, a)
              : launch(ctx, riemann_aux_large_kernel<MM_RMETRIC_DIAGQUAD, 0>, a);
  if (op == 1)
    return r1 ? launch(ctx, riemann_aux_large_kernel<MM_RMETRIC_RANK1, 1>, a)
              : launch(ctx, riemann_aux_large_kernel<MM_RMETRIC_DIAGQUAD, 1>, a);
  return r1 ? launch(ctx, riemann_aux_large_kernel<MM_RMETRIC_RANK1, 2>, a)
            : launch(ctx, riemann_aux_large_kernel<MM_RMETRIC_DIAGQUAD, 2>, a);
}
