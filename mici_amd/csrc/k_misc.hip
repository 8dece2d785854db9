// Wave-per-chain kernels for Euclidean-metric systems that are not on the fast paths of
// k_euclid.hip: the generic explicit leapfrog (any built-in target, any fixed metric, any D) and the
// System-level quantities mici.transitions needs around the integrator:
//   EuclideanMetricSystem.h / h2     systems.py:187-196, 348-350     h = l(q) + p^T M^-1 p / 2
//   EuclideanMetricSystem.dh2_dmom   systems.py:352-354              M^-1 p
//   EuclideanMetricSystem.sample_momentum  systems.py:365-366        M^{1/2} z  (z from the host RNG)
// One 64-lane wave owns a chain; its q, p vectors live in LDS, element i on lane i mod 64.
#include "mm_device.h"

namespace {

using namespace mmdev;

struct EuclidModelView {
  int target, metric_kind, dim;
  const double* tparams;
  const double* minv;   // diag: 1/diag[D]; dense: explicit inverse [D*D]
  const double* mchol;  // diag: sqrt(diag)[D]; dense: lower Cholesky factor [D*D]
};

__device__ __forceinline__ double minv_elem(const EuclidModelView& m, const double* p, int i) {
  if (m.metric_kind == MM_METRIC_IDENTITY) return p[i];
  if (m.metric_kind == MM_METRIC_DIAG) return m.minv[i] * p[i];
  double s = 0.0;
  const double* row = m.minv + (int64_t)i * m.dim;
  for (int j = 0; j < m.dim; ++j) s += row[j] * p[j];
  return s;
}

// dynamic LDS: per wave 3*dim doubles (q, p, scratch)
__global__ void leapfrog_generic_kernel(EuclidModelView m, double* __restrict__ pos,
                                        double* __restrict__ mom, const int8_t* __restrict__ dir,
                                        int64_t n_chains, double step_size, int n_steps) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int dim = m.dim;
  const int64_t chain = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
  if (chain >= n_chains) return;  // whole wave exits together; no block-level barriers below
  double* q = lds + (size_t)wave * 3 * dim;
  double* p = q + dim;
  double* g = p + dim;
  for (int i = lane; i < dim; i += 64) {
    q[i] = pos[chain * dim + i];
    p[i] = mom[chain * dim + i];
  }
  const double t = (double)dir[chain] * step_size, ht = 0.5 * t;
  wave_sync();
  TargetAux aux = target_prepare(m.target, q, dim, m.tparams, lane);
  for (int i = lane; i < dim; i += 64) g[i] = target_grad_elem(m.target, aux, q, i, dim, m.tparams);
  for (int s = 0; s < n_steps; ++s) {
    for (int i = lane; i < dim; i += 64) p[i] -= ht * g[i];
    wave_sync();
    // q += t * M^-1 p  (dense rows read the whole p, so stage the update through g)
    for (int i = lane; i < dim; i += 64) g[i] = minv_elem(m, p, i);
    wave_sync();
    for (int i = lane; i < dim; i += 64) q[i] += t * g[i];
    wave_sync();
    aux = target_prepare(m.target, q, dim, m.tparams, lane);
    for (int i = lane; i < dim; i += 64) g[i] = target_grad_elem(m.target, aux, q, i, dim, m.tparams);
    wave_sync();
    for (int i = lane; i < dim; i += 64) p[i] -= ht * g[i];
  }
  wave_sync();
  for (int i = lane; i < dim; i += 64) {
    pos[chain * dim + i] = q[i];
    mom[chain * dim + i] = p[i];
  }
}

// SymmetricCompositionIntegrator._step (integrators.py:272-274) on a Euclidean-metric system: the
// coefficient sequence c[0..m) alternates h1_flow (mom -= c t grad) and h2_flow (pos += c t M^-1 mom),
// starting with h1 iff initial_h1.  Same wave-per-chain layout as the generic leapfrog above; the
// gradient is recomputed only after the position moved (the reference's state cache).
struct CompCoefs {
  int m, initial_h1;
  double c[MM_MAX_COMPOSITION_COEFFS];
};

__global__ void composition_generic_kernel(EuclidModelView m, double* __restrict__ pos,
                                           double* __restrict__ mom, const int8_t* __restrict__ dir,
                                           int64_t n_chains, double step_size, int n_steps, CompCoefs cf) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int dim = m.dim;
  const int64_t chain = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
  if (chain >= n_chains) return;  // whole wave exits together; no block-level barriers below
  double* q = lds + (size_t)wave * 3 * dim;
  double* p = q + dim;
  double* g = p + dim;  // the cached gradient; doubles as scratch for M^-1 p during an h2 flow
  for (int i = lane; i < dim; i += 64) {
    q[i] = pos[chain * dim + i];
    p[i] = mom[chain * dim + i];
  }
  const double t = (double)dir[chain] * step_size;
  wave_sync();
  TargetAux aux = target_prepare(m.target, q, dim, m.tparams, lane);
  for (int i = lane; i < dim; i += 64) g[i] = target_grad_elem(m.target, aux, q, i, dim, m.tparams);
  wave_sync();
  for (int s = 0; s < n_steps; ++s) {
    for (int k = 0; k < cf.m; ++k) {
      const double ct = cf.c[k] * t;
      if (((k & 1) == 0) == (cf.initial_h1 != 0)) {
        for (int i = lane; i < dim; i += 64) p[i] -= ct * g[i];
        wave_sync();
      } else {
        for (int i = lane; i < dim; i += 64) g[i] = minv_elem(m, p, i);
        wave_sync();
        for (int i = lane; i < dim; i += 64) q[i] += ct * g[i];
        wave_sync();
        aux = target_prepare(m.target, q, dim, m.tparams, lane);
        for (int i = lane; i < dim; i += 64) g[i] = target_grad_elem(m.target, aux, q, i, dim, m.tparams);
        wave_sync();
      }
    }
  }
  wave_sync();
  for (int i = lane; i < dim; i += 64) {
    pos[chain * dim + i] = q[i];
    mom[chain * dim + i] = p[i];
  }
}

enum { OP_H = 0, OP_DH_DMOM = 1, OP_SAMPLE_MOM = 2 };

template <int OP>
__global__ void euclid_aux_kernel(EuclidModelView m, const double* __restrict__ pos,
                                  double* __restrict__ mom, int64_t n_chains,
                                  double* __restrict__ out, const double* __restrict__ z) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int dim = m.dim;
  const int64_t chain = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
  if (chain >= n_chains) return;
  double* q = lds + (size_t)wave * 3 * dim;
  double* p = q + dim;
  for (int i = lane; i < dim; i += 64) {
    q[i] = pos[chain * dim + i];
    p[i] = (OP == OP_SAMPLE_MOM) ? z[chain * dim + i] : mom[chain * dim + i];
  }
  wave_sync();
  if constexpr (OP == OP_H) {
    const TargetAux aux = target_prepare(m.target, q, dim, m.tparams, lane);
    double acc = 0.0;
    for (int i = lane; i < dim; i += 64)
      acc += target_nld_elem(m.target, aux, q, i, dim, m.tparams) + 0.5 * p[i] * minv_elem(m, p, i);
    acc = wave_sum(acc);
    if (lane == 0) out[chain] = acc;
  } else if constexpr (OP == OP_DH_DMOM) {
    for (int i = lane; i < dim; i += 64) out[chain * dim + i] = minv_elem(m, p, i);
  } else {
    for (int i = lane; i < dim; i += 64) {
      double v;
      if (m.metric_kind == MM_METRIC_IDENTITY) v = p[i];
      else if (m.metric_kind == MM_METRIC_DIAG) v = m.mchol[i] * p[i];
      else {
        v = 0.0;
        const double* row = m.mchol + (int64_t)i * dim;
        for (int j = 0; j <= i; ++j) v += row[j] * p[j];
      }
      mom[chain * dim + i] = v;
    }
  }
}

EuclidModelView view_of(const mm_model* m) {
  return EuclidModelView{m->target, m->metric_kind, m->dim, m->d_target_params, m->d_metric_inv,
                         m->d_metric_chol};
}

int waves_per_block(int dim, size_t* lds_bytes) {
  int w = 4;
  while (w > 1 && (size_t)w * 3 * dim * sizeof(double) > 60 * 1024) w >>= 1;
  *lds_bytes = (size_t)w * 3 * dim * sizeof(double);
  return w;
}

}  // namespace

int mm_launch_leapfrog_generic(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int n_steps) {
  size_t lds;
  const int w = waves_per_block(s->dim, &lds);
  if (lds > 64 * 1024) {
    mm_set_error(ctx, "mm_leapfrog_euclid: dim too large for the generic kernel's LDS tile");
    return MM_ERR_UNSUPPORTED;
  }
  const unsigned blocks = (unsigned)((s->n + w - 1) / w);
  hipLaunchKernelGGL(leapfrog_generic_kernel, dim3(blocks), dim3(64 * w), lds, ctx->stream,
                     view_of(m), s->d_pos, s->d_mom, s->d_dir, s->n, h, n_steps);
  MM_HIP_CHECK(ctx, hipGetLastError());
  return MM_OK;
}

int mm_launch_composition_generic(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int n_steps,
                                  int n_coeffs, const double* coeffs, int initial_h1) {
  size_t lds;
  const int w = waves_per_block(s->dim, &lds);
  if (lds > 64 * 1024) {
    mm_set_error(ctx, "mm_composition_euclid: dim too large for the generic kernel's LDS tile");
    return MM_ERR_UNSUPPORTED;
  }
  CompCoefs cf{};
  cf.m = n_coeffs;
  cf.initial_h1 = initial_h1;
  for (int i = 0; i < n_coeffs; ++i) cf.c[i] = coeffs[i];
  const unsigned blocks = (unsigned)((s->n + w - 1) / w);
  hipLaunchKernelGGL(composition_generic_kernel, dim3(blocks), dim3(64 * w), lds, ctx->stream, view_of(m),
                     s->d_pos, s->d_mom, s->d_dir, s->n, h, n_steps, cf);
  MM_HIP_CHECK(ctx, hipGetLastError());
  return MM_OK;
}

// Metropolis accept / select (transitions.py:296-314): one wave per chain; lane 0 decides, all lanes copy.
__global__ void metropolis_select_kernel(double* __restrict__ pos, double* __restrict__ mom,
                                         int8_t* __restrict__ dir, const double* __restrict__ ppos,
                                         const double* __restrict__ pmom, const int32_t* __restrict__ pstatus,
                                         const int32_t* __restrict__ pn_done, const double* __restrict__ h0,
                                         const double* __restrict__ h1, const double* __restrict__ u,
                                         double* __restrict__ accept_prob, int8_t* __restrict__ accepted,
                                         int64_t n_chains, int dim) {
  const int lane = threadIdx.x & 63;
  const int64_t chain = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (chain >= n_chains) return;
  const bool error = pstatus[chain] != 0;
  const bool moved = pn_done[chain] > 0;
  const double h_diff = h0[chain] - h1[chain];
  double prob = 0.0;
  if (moved && h_diff == h_diff) prob = exp(h_diff < 0.0 ? h_diff : 0.0);
  const bool acc = !error && (u[chain] < prob);
  if (acc) {
    for (int i = lane; i < dim; i += 64) {
      pos[chain * dim + i] = ppos[chain * dim + i];
      mom[chain * dim + i] = pmom[chain * dim + i];
    }
  }
  if (lane == 0) {
    if (!acc) dir[chain] = (int8_t)(-dir[chain]);
    accept_prob[chain] = prob;
    accepted[chain] = acc ? 1 : 0;
  }
}

int mm_launch_metropolis_select(mm_ctx* ctx, mm_state* s, mm_state* prop, const double* d_h0, const double* d_h1,
                                const double* d_u, double* d_prob, int8_t* d_acc) {
  const int w = 4;
  const unsigned blocks = (unsigned)((s->n + w - 1) / w);
  hipLaunchKernelGGL(metropolis_select_kernel, dim3(blocks), dim3(64 * w), 0, ctx->stream, s->d_pos, s->d_mom,
                     s->d_dir, prop->d_pos, prop->d_mom, prop->d_status, prop->d_n_done, d_h0, d_h1, d_u, d_prob,
                     d_acc, s->n, s->dim);
  MM_HIP_CHECK(ctx, hipGetLastError());
  return MM_OK;
}

template <int OP>
static int launch_aux(mm_ctx* ctx, const mm_model* m, mm_state* s, double* out, const double* z) {
  size_t lds;
  const int w = waves_per_block(s->dim, &lds);
  if (lds > 64 * 1024) {
    mm_set_error(ctx, "dim too large for the wave-per-chain auxiliary kernel");
    return MM_ERR_UNSUPPORTED;
  }
  const unsigned blocks = (unsigned)((s->n + w - 1) / w);
  hipLaunchKernelGGL((euclid_aux_kernel<OP>), dim3(blocks), dim3(64 * w), lds, ctx->stream,
                     view_of(m), s->d_pos, s->d_mom, s->n, out, z);
  MM_HIP_CHECK(ctx, hipGetLastError());
  return MM_OK;
}

int mm_launch_euclid_hamiltonian(mm_ctx* ctx, const mm_model* m, mm_state* s, double* d_h) {
  return launch_aux<OP_H>(ctx, m, s, d_h, nullptr);
}
int mm_launch_euclid_dh_dmom(mm_ctx* ctx, const mm_model* m, mm_state* s, double* d_out) {
  return launch_aux<OP_DH_DMOM>(ctx, m, s, d_out, nullptr);
}
int mm_launch_euclid_sample_momentum(mm_ctx* ctx, const mm_model* m, mm_state* s, const double* d_z) {
  return launch_aux<OP_SAMPLE_MOM>(ctx, m, s, nullptr, d_z);
}
