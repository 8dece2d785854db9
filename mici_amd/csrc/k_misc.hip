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

__device__ __forceinline__ int chain_steps_of(const int32_t* cs, int64_t chain, int n_steps) {
  return mmdev::chain_steps(cs, chain, n_steps);  // (a kernel parameter of the same name shadows the function)
}

struct EuclidModelView {
  int target, metric_kind, dim;
  const double* tparams;
  const double* minv;   // diag: 1/diag[D]; dense: explicit inverse [D*D]
  const double* mchol;  // diag: sqrt(diag)[D]; dense: lower Cholesky factor [D*D]
  int gaussian;         // GaussianEuclideanMetricSystem: h2 = q.q/2 + p.M^-1 p/2, exact h2 flow
  const double* omega;  // gaussian: 1/sqrt(eigval)[D] (nullptr for the identity metric)
  const double* eigvec; // gaussian + dense: V [D*D] followed by V^T [D*D], row-major
};

__device__ __forceinline__ double minv_elem(const EuclidModelView& m, const double* p, int i) {
  if (m.metric_kind == MM_METRIC_IDENTITY) return p[i];
  if (m.metric_kind == MM_METRIC_DIAG) return m.minv[i] * p[i];
  double s = 0.0;
  const double* row = m.minv + (int64_t)i * m.dim;
  for (int j = 0; j < m.dim; ++j) s += row[j] * p[j];
  return s;
}

// GaussianEuclideanMetricSystem.h2_flow (systems.py:464-474): with M = V diag(e) V^T and w = 1/sqrt(e),
//   pos <- V (cos(w dt) V^T pos + sin(w dt) w V^T mom),  mom <- V (cos(w dt) V^T mom - sin(w dt)/w V^T pos)
// an exact rotation, so the split integrator only discretises the non-Gaussian part of the target.
// a, b are two scratch vectors; V and V^T are both stored so lane i always walks a contiguous column.
__device__ __forceinline__ void gaussian_h2_flow(const EuclidModelView& m, double* q, double* p, double* a,
                                                 double* b, double dt, int lane) {
  const int dim = m.dim;
  const bool dense = m.metric_kind == MM_METRIC_DENSE;
  if (dense) {
    for (int i = lane; i < dim; i += 64) {
      double sa = 0.0, sb = 0.0;
      for (int j = 0; j < dim; ++j) {
        const double v = m.eigvec[(int64_t)j * dim + i];
        sa += v * q[j];
        sb += v * p[j];
      }
      a[i] = sa;
      b[i] = sb;
    }
    wave_sync();
  }
  for (int i = lane; i < dim; i += 64) {
    const double om = m.omega ? m.omega[i] : 1.0;
    double sn, cs;
    sincos(om * dt, &sn, &cs);
    const double ai = dense ? a[i] : q[i], bi = dense ? b[i] : p[i];
    const double na = cs * ai + (sn * om) * bi, nb = cs * bi - (sn / om) * ai;
    if (dense) {
      a[i] = na;
      b[i] = nb;
    } else {
      q[i] = na;
      p[i] = nb;
    }
  }
  wave_sync();
  if (dense) {
    const double* vt = m.eigvec + (int64_t)dim * dim;
    for (int i = lane; i < dim; i += 64) {
      double sq = 0.0, sp = 0.0;
      for (int j = 0; j < dim; ++j) {
        const double v = vt[(int64_t)j * dim + i];
        sq += v * a[j];
        sp += v * b[j];
      }
      q[i] = sq;
      p[i] = sp;
    }
    wave_sync();
  }
}

// dynamic LDS: per wave 3*dim doubles (q, p, scratch), 4*dim for the Gaussian split
__global__ void leapfrog_generic_kernel(EuclidModelView m, double* __restrict__ pos,
                                        double* __restrict__ mom, const int8_t* __restrict__ dir,
                                        const double* __restrict__ step_scale,
                                        const int32_t* __restrict__ chain_steps, int32_t* __restrict__ status,
                                        int32_t* __restrict__ n_done, int64_t n_chains, double step_size,
                                        int n_steps) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int dim = m.dim;
  const int64_t chain = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
  if (chain >= n_chains) return;  // whole wave exits together; no block-level barriers below
  const int nvec = m.gaussian ? 4 : 3;
  double* q = lds + (size_t)wave * nvec * dim;
  double* p = q + dim;
  double* g = p + dim;
  for (int i = lane; i < dim; i += 64) {
    q[i] = pos[chain * dim + i];
    p[i] = mom[chain * dim + i];
  }
  const double t = signed_step(dir, step_scale, chain, step_size), ht = 0.5 * t;
  n_steps = chain_steps_of(chain_steps, chain, n_steps);
  wave_sync();
  TargetAux aux = target_prepare(m.target, q, dim, m.tparams, lane);
  for (int i = lane; i < dim; i += 64) g[i] = target_grad_elem(m.target, aux, q, i, dim, m.tparams);
  for (int s = 0; s < n_steps; ++s) {
    for (int i = lane; i < dim; i += 64) p[i] -= ht * g[i];
    wave_sync();
    if (m.gaussian) {
      gaussian_h2_flow(m, q, p, g, g + dim, t, lane);
    } else {
      // q += t * M^-1 p  (dense rows read the whole p, so stage the update through g)
      for (int i = lane; i < dim; i += 64) g[i] = minv_elem(m, p, i);
      wave_sync();
      for (int i = lane; i < dim; i += 64) q[i] += t * g[i];
      wave_sync();
    }
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
  if (lane == 0) {  // explicit steps cannot fail
    status[chain] = 0;
    n_done[chain] = n_steps;
  }
}

// SymmetricCompositionIntegrator._step (integrators.py:272-274) on a Euclidean-metric system: the
// coefficient sequence c[0..m) alternates h1_flow (mom -= c t grad) and h2_flow (pos += c t M^-1 mom),
// starting with h1 iff initial_h1.  Same wave-per-chain layout as the generic leapfrog above; the
// gradient is recomputed only after the position moved (the reference's state cache).
using CompCoefs = mm_comp_coefs;

__global__ void composition_generic_kernel(EuclidModelView m, double* __restrict__ pos,
                                           double* __restrict__ mom, const int8_t* __restrict__ dir,
                                           const double* __restrict__ step_scale,
                                           const int32_t* __restrict__ chain_steps, int32_t* __restrict__ status,
                                           int32_t* __restrict__ n_done, int64_t n_chains, double step_size,
                                           int n_steps, CompCoefs cf) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int dim = m.dim;
  const int64_t chain = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
  if (chain >= n_chains) return;  // whole wave exits together; no block-level barriers below
  const int nvec = m.gaussian ? 4 : 3;
  double* q = lds + (size_t)wave * nvec * dim;
  double* p = q + dim;
  double* g = p + dim;  // the cached gradient; doubles as scratch for M^-1 p during an h2 flow
  for (int i = lane; i < dim; i += 64) {
    q[i] = pos[chain * dim + i];
    p[i] = mom[chain * dim + i];
  }
  const double t = signed_step(dir, step_scale, chain, step_size);
  n_steps = chain_steps_of(chain_steps, chain, n_steps);
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
        if (m.gaussian) {
          gaussian_h2_flow(m, q, p, g, g + dim, ct, lane);
        } else {
          for (int i = lane; i < dim; i += 64) g[i] = minv_elem(m, p, i);
          wave_sync();
          for (int i = lane; i < dim; i += 64) q[i] += ct * g[i];
          wave_sync();
        }
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
  if (lane == 0) {  // explicit steps cannot fail
    status[chain] = 0;
    n_done[chain] = n_steps;
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
      acc += target_nld_elem(m.target, aux, q, i, dim, m.tparams) + 0.5 * p[i] * minv_elem(m, p, i) +
             (m.gaussian ? 0.5 * q[i] * q[i] : 0.0);  // GaussianEuclideanMetricSystem.h2, systems.py:451-454
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
                         m->d_metric_chol, m->gaussian_split, m->d_metric_omega, m->d_metric_eigvec};
}

constexpr size_t kMaxGenericLds = 160 * 1024;  // a CU's LDS: D <= 6826 (5120 with the Gaussian split's fourth vector)

int waves_per_block(int dim, size_t* lds_bytes, int nvec = 3) {
  int w = 4;
  while (w > 1 && (size_t)w * nvec * dim * sizeof(double) > 60 * 1024) w >>= 1;
  *lds_bytes = (size_t)w * nvec * dim * sizeof(double);
  return w;
}

}  // namespace

int mm_launch_leapfrog_generic(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int n_steps) {
  size_t lds;
  const int w = waves_per_block(s->dim, &lds, m->gaussian_split ? 4 : 3);
  if (lds > kMaxGenericLds) {
    mm_set_error(ctx, "mm_leapfrog_euclid: dim too large for the generic kernel (three D-vectors of a chain in 160 KB of LDS)");
    return MM_ERR_UNSUPPORTED;
  }
  MM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(leapfrog_generic_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const unsigned blocks = (unsigned)((s->n + w - 1) / w);
  hipLaunchKernelGGL(leapfrog_generic_kernel, dim3(blocks), dim3(64 * w), lds, ctx->stream,
                     view_of(m), s->d_pos, s->d_mom, s->d_dir, s->d_step_scale, s->d_chain_steps, s->d_status, s->d_n_done,
                     s->n, h, n_steps);
  MM_HIP_CHECK(ctx, hipGetLastError());
  return MM_OK;
}

int mm_launch_composition_generic(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int n_steps,
                                  int n_coeffs, const double* coeffs, int initial_h1) {
  size_t lds;
  const int w = waves_per_block(s->dim, &lds, m->gaussian_split ? 4 : 3);
  if (lds > kMaxGenericLds) {
    mm_set_error(ctx, "mm_composition_euclid: dim too large for the generic kernel (three D-vectors of a chain in 160 KB of LDS)");
    return MM_ERR_UNSUPPORTED;
  }
  MM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(composition_generic_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CompCoefs cf{};
  cf.m = n_coeffs;
  cf.initial_h1 = initial_h1;
  for (int i = 0; i < n_coeffs; ++i) cf.c[i] = coeffs[i];
  const unsigned blocks = (unsigned)((s->n + w - 1) / w);
  hipLaunchKernelGGL(composition_generic_kernel, dim3(blocks), dim3(64 * w), lds, ctx->stream, view_of(m),
                     s->d_pos, s->d_mom, s->d_dir, s->d_step_scale, s->d_chain_steps, s->d_status, s->d_n_done, s->n, h, n_steps,
                     cf);
  MM_HIP_CHECK(ctx, hipGetLastError());
  return MM_OK;
}

// ImplicitMidpointIntegrator (integrators.py:547-681) on a Euclidean-metric system, one wave per chain.
// LDS per wave: 13 vectors of dim doubles (dim <= 1024: 104 KB).  dh_dmom = M^-1 p, dh_dpos = grad(q).
struct MidpointLds {
  double *q, *p, *xiq, *xip, *x0q, *x0p, *x1q, *x1p, *ptq, *ptp, *tmp, *q1, *p1;
};

__device__ __forceinline__ double pair_norm_lds(const double* aq, const double* bq, const double* ap,
                                                const double* bp, int dim, int lane, int kind) {
  double acc = 0.0;
  for (int i = lane; i < dim; i += 64) {
    acc = wave_norm_accum(acc, aq[i] - bq[i], kind);
    acc = wave_norm_accum(acc, ap[i] - bp[i], kind);
  }
  return wave_norm_finish(acc, kind);
}

// x = fixed point of x_init + tt * (dh_dmom(x), -dh_dpos(x)) started at x_init (L.xiq, L.xip); the
// solution is left in (L.ptq, L.ptp).  solve_fixed_point_direct / _steffensen (solvers.py:47-154).
__device__ __forceinline__ int midpoint_solve(const EuclidModelView& m, const MidpointLds& L, double tt,
                                              const mm_fp_opts& o, int lane, long long* n_evals) {
  const int dim = m.dim;
  for (int i = lane; i < dim; i += 64) {
    L.x0q[i] = L.xiq[i];
    L.x0p[i] = L.xip[i];
    L.ptq[i] = L.xiq[i];
    L.ptp[i] = L.xip[i];
  }
  wave_sync();
  int stage = 0;
  for (int iter = 0; iter < o.max_iters;) {
    // f(pt) -> (tmp holds M^-1 p first), result overwrites pt
    const TargetAux aux = target_prepare(m.target, L.ptq, dim, m.tparams, lane);
    for (int i = lane; i < dim; i += 64) L.tmp[i] = minv_elem(m, L.ptp, i);
    wave_sync();
    for (int i = lane; i < dim; i += 64) {
      const double g = target_grad_elem(m.target, aux, L.ptq, i, dim, m.tparams);
      L.ptp[i] = L.xip[i] - tt * g;
    }
    wave_sync();  // every lane has read ptq before it is overwritten
    for (int i = lane; i < dim; i += 64) L.ptq[i] = L.xiq[i] + tt * L.tmp[i];
    wave_sync();
    *n_evals += 1;
    if (o.solver != MM_FP_DIRECT) {
      if (stage == 0) {  // x1 = f(x0); evaluate f(x1) next
        for (int i = lane; i < dim; i += 64) {
          L.x1q[i] = L.ptq[i];
          L.x1p[i] = L.ptp[i];
        }
        stage = 1;
        wave_sync();
        continue;
      }
      const double eps = 2.220446049250313e-16;
      for (int i = lane; i < dim; i += 64) {
        double dq = L.ptq[i] - 2.0 * L.x1q[i] + L.x0q[i], dp = L.ptp[i] - 2.0 * L.x1p[i] + L.x0p[i];
        if (fabs(dq) == 0.0) dq = eps;
        if (fabs(dp) == 0.0) dp = eps;
        const double eq = L.x1q[i] - L.x0q[i], ep = L.x1p[i] - L.x0p[i];
        L.ptq[i] = L.x0q[i] - eq * eq / dq;
        L.ptp[i] = L.x0p[i] - ep * ep / dp;
      }
      stage = 0;
      wave_sync();
    }
    const double err = pair_norm_lds(L.ptq, L.x0q, L.ptp, L.x0p, dim, lane, o.norm);
    if (err > o.div_tol || err != err) return MM_ST_DIVERGED;
    if (err < o.conv_tol) return MM_ST_OK;
    for (int i = lane; i < dim; i += 64) {
      L.x0q[i] = L.ptq[i];
      L.x0p[i] = L.ptp[i];
    }
    wave_sync();
    ++iter;
  }
  return MM_ST_MAX_ITERS;
}

__global__ void midpoint_euclid_kernel(EuclidModelView m, double* __restrict__ pos, double* __restrict__ mom,
                                       const int8_t* __restrict__ dir, const double* __restrict__ step_scale,
                                       const int32_t* __restrict__ chain_steps,
                                       int32_t* __restrict__ status, int32_t* __restrict__ n_done,
                                       int64_t n_chains, double step_size,
                                       int n_steps, mm_fp_opts o, mm_counters* counters) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int dim = m.dim;
  const int64_t chain = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
  if (chain >= n_chains) return;
  double* b = lds + (size_t)wave * 13 * dim;
  const MidpointLds L{b,           b + dim,     b + 2 * dim, b + 3 * dim, b + 4 * dim,  b + 5 * dim, b + 6 * dim,
                      b + 7 * dim, b + 8 * dim, b + 9 * dim, b + 10 * dim, b + 11 * dim, b + 12 * dim};
  for (int i = lane; i < dim; i += 64) {
    L.q[i] = pos[chain * dim + i];
    L.p[i] = mom[chain * dim + i];
  }
  wave_sync();
  const double half = 0.5 * signed_step(dir, step_scale, chain, step_size);
  n_steps = chain_steps_of(chain_steps, chain, n_steps);
  int st = MM_ST_OK, done = 0;
  long long n_evals = 0, n_solves = 0, n_grad = 0;
  for (int s = 0; s < n_steps; ++s) {
    // A(t/2): implicit Euler half step
    for (int i = lane; i < dim; i += 64) {
      L.xiq[i] = L.q[i];
      L.xip[i] = L.p[i];
    }
    wave_sync();
    ++n_solves;
    st = midpoint_solve(m, L, half, o, lane, &n_evals);
    if (st != MM_ST_OK) break;
    // A*(t/2): explicit Euler half step from (q1, p1) = pt; keep (q1, p1) in x1 for the check
    {
      const TargetAux aux = target_prepare(m.target, L.ptq, dim, m.tparams, lane);
      for (int i = lane; i < dim; i += 64) L.tmp[i] = minv_elem(m, L.ptp, i);
      wave_sync();
      ++n_grad;
      for (int i = lane; i < dim; i += 64) {
        const double g = target_grad_elem(m.target, aux, L.ptq, i, dim, m.tparams);
        L.xiq[i] = L.ptq[i] + half * L.tmp[i];
        L.xip[i] = L.ptp[i] - half * g;
      }
      wave_sync();
    }
    // reversibility check: A(-t/2) from the new state must return to (q1, p1), kept across the solve in two LDS vectors
    // of their own (round 5: they were two registers a lane, which is where the dim <= 128 limit came from)
    for (int i = lane; i < dim; i += 64) {
      L.q1[i] = L.ptq[i];
      L.p1[i] = L.ptp[i];
    }
    ++n_solves;
    st = midpoint_solve(m, L, -half, o, lane, &n_evals);
    if (st != MM_ST_OK) break;
    {
      double acc = 0.0;
      for (int i = lane; i < dim; i += 64) {
        acc = wave_norm_accum(acc, L.ptq[i] - L.q1[i], o.rev_norm);
        acc = wave_norm_accum(acc, L.ptp[i] - L.p1[i], o.rev_norm);
      }
      const double rev = wave_norm_finish(acc, o.rev_norm);
      if (rev > o.rev_tol) {
        st = MM_ST_NON_REVERSIBLE;
        break;
      }
    }
    for (int i = lane; i < dim; i += 64) {
      L.q[i] = L.xiq[i];
      L.p[i] = L.xip[i];
    }
    wave_sync();
    ++done;
  }
  wave_sync();
  for (int i = lane; i < dim; i += 64) {
    pos[chain * dim + i] = L.q[i];
    mom[chain * dim + i] = L.p[i];
  }
  if (lane == 0) {
    status[chain] = st;
    n_done[chain] = done;
    if (counters) {
      atomicAdd((unsigned long long*)&counters->n_fp_evals, (unsigned long long)n_evals);
      atomicAdd((unsigned long long*)&counters->n_fp_solves, (unsigned long long)n_solves);
      atomicAdd((unsigned long long*)&counters->n_grad, (unsigned long long)(n_evals + n_grad));
    }
  }
}

int mm_launch_implicit_midpoint_euclid(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int n_steps,
                                       const mm_fp_opts& opts, mm_counters* d_counters) {
  if (s->dim > 1024) {
    mm_set_error(ctx, "mm_implicit_midpoint: Euclidean systems are supported for dim <= 1024 (13 vectors of a chain in LDS)");
    return MM_ERR_UNSUPPORTED;
  }
  const size_t per_wave = (size_t)13 * s->dim * sizeof(double);
  int w = (int)(150 * 1024 / per_wave);
  w = w > 4 ? 4 : (w < 1 ? 1 : w);
  const unsigned blocks = (unsigned)((s->n + w - 1) / w);
  MM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(midpoint_euclid_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)(w * per_wave)));
  hipLaunchKernelGGL(midpoint_euclid_kernel, dim3(blocks), dim3(64 * w), w * per_wave, ctx->stream, view_of(m),
                     s->d_pos, s->d_mom, s->d_dir, s->d_step_scale, s->d_chain_steps, s->d_status, s->d_n_done, s->n, h,
                     n_steps, opts,
                     d_counters);
  MM_HIP_CHECK(ctx, hipGetLastError());
  return MM_OK;
}

__global__ void fill_done_kernel(int32_t* __restrict__ status, int32_t* __restrict__ n_done,
                                 const int32_t* __restrict__ chain_steps, int64_t n, int32_t n_steps) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i < n) {
    status[i] = 0;
    n_done[i] = chain_steps_of(chain_steps, i, n_steps);
  }
}

int mm_launch_fill_done(mm_ctx* ctx, mm_state* s, int32_t n_steps) {
  if (s->n == 0) return MM_OK;
  hipLaunchKernelGGL(fill_done_kernel, dim3((unsigned)((s->n + 255) / 256)), dim3(256), 0, ctx->stream, s->d_status,
                     s->d_n_done, s->d_chain_steps, s->n, n_steps);
  MM_HIP_CHECK(ctx, hipGetLastError());
  return MM_OK;
}

// y <- a x + b y  (CorrelatedMomentumTransition: mom *= sqrt(1 - c^2); mom += c mom_ind, transitions.py:194-196)
__global__ void axpby_kernel(double* __restrict__ y, const double* __restrict__ x, double a, double b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double scaled = a * x[i];
    y[i] = scaled + b * y[i];
  }
}

int mm_launch_axpby(mm_ctx* ctx, double* y, const double* x, double a, double b, size_t n) {
  const unsigned blocks = (unsigned)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  hipLaunchKernelGGL(axpby_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, ctx->stream, y, x, a, b, n);
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
                                         uint32_t* __restrict__ errors, int64_t n_chains, int dim) {
  const int lane = threadIdx.x & 63;
  const int64_t chain = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (chain >= n_chains) return;
  const int pst = pstatus[chain];
  const bool error = pst != 0;
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
    // the proposal's status is overwritten by the next transition's copy: keep what went wrong with the CHAIN
    // (bit k = some proposal ended with status k; read and cleared by mm_state_download_errors)
    if (error) errors[chain] |= 1u << (pst & 31);
  }
}

int mm_launch_metropolis_select(mm_ctx* ctx, mm_state* s, mm_state* prop, const double* d_h0, const double* d_h1,
                                const double* d_u, double* d_prob, int8_t* d_acc) {
  const int w = 4;
  const unsigned blocks = (unsigned)((s->n + w - 1) / w);
  hipLaunchKernelGGL(metropolis_select_kernel, dim3(blocks), dim3(64 * w), 0, ctx->stream, s->d_pos, s->d_mom,
                     s->d_dir, prop->d_pos, prop->d_mom, prop->d_status, prop->d_n_done, d_h0, d_h1, d_u, d_prob,
                     d_acc, s->d_errors, s->n, s->dim);
  MM_HIP_CHECK(ctx, hipGetLastError());
  return MM_OK;
}

template <int OP>
static int launch_aux(mm_ctx* ctx, const mm_model* m, mm_state* s, double* out, const double* z) {
  size_t lds;
  const int w = waves_per_block(s->dim, &lds);
  if (lds > kMaxGenericLds) {
    mm_set_error(ctx, "dim too large for the wave-per-chain auxiliary kernel (three D-vectors of a chain in 160 KB of LDS)");
    return MM_ERR_UNSUPPORTED;
  }
  MM_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(euclid_aux_kernel<OP>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
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
