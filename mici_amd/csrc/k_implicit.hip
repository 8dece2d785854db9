// Implicit (generalised) leapfrog on dense-metric Riemannian systems, one 64-lane wave per chain,
// D <= 64 (gfx950 / CDNA4).
//
// Replaces, per chain and per step (reference /root/reference/src/mici):
//   ImplicitLeapfrogIntegrator._step / _step_a / _step_b_fwd / _step_b_adj / _step_c_fwd / _step_c_adj
//                                        integrators.py:493-544  (all six sub-maps use the FULL t: H1)
//   solve_fixed_point_direct / _steffensen solvers.py:47-94, 97-154 (+ maximum/euclidean norm :20-27)
//   RiemannianMetricSystem.dh1_dpos / dh2_dpos / dh2_dmom / h / sample_momentum  systems.py:1375-1402
//   DensePositiveDefiniteMatrix: factorisation, explicit inverse, log|det|, grad_log_abs_det,
//       grad_quadratic_form_inv            matrices.py:1161-1188, 1175-1181, 982-984
//
// Data layout.  The D x D metric of a chain is held ENTIRELY IN REGISTERS of its wave, distributed
// 2-D block-cyclically: lane (ti, tj) = (lane>>3, lane&7) owns the TS x TS entries
// {(ti + 8a, tj + 8b)}, TS = ceil(D/8) <= 8 (128 VGPRs at D = 64).  With this layout
//   * one step of the symmetric sweep operator (Gauss-Jordan on an SPD matrix, which keeps the matrix
//     symmetric so "column k" and "row k" are the same vector) is a rank-1 update in which every lane
//     does TS*TS independent v_fma_f64 from 2*TS operands -> VALU bound, no triangular structure, no
//     sqrt, no back-substitution; 64 sweeps give -M^-1 and the pivots give log det M;
//   * M^-1 v is a TS x TS register mat-vec plus a 3-stage xor-shuffle reduction over the 8 lanes of a
//     row group.
// The published pivot column and all D-vectors move through a few hundred bytes of per-wave LDS;
// the shared base matrix of the rank-one metric sits in LDS once per workgroup.  Each wave iterates
// its own fixed-point solves, so data-dependent iteration counts and failures need no masking: a
// failed chain's wave simply stops (status / n_done, chain frozen at its last good state).
#include "implicit_wave.h"
#include <cstring>

namespace {

using namespace mmdev;
using namespace mmimp;
using namespace mmwave;

#ifdef MM_DEV_KERNELS
// ---- developer micro-benchmark (not part of the ABI header): R repetitions of one primitive per wave
template <int TS>
__global__ __launch_bounds__(64 * kWaves) void debug_primitive_kernel(ImplicitArgs A, int variant,
                                                                      int repeats, double* sink) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double* base_lds = lds;
  const int base_elems = 64 * Geo<TS>::TSTRIDE;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double* wl = lds + base_elems + wave * kWaveLdsDoubles;
  const WaveLds w{wl, wl + 64, wl + 128, wl + 192, wl + 256};
  double* blk = wl + 320 + SL_COUNT_REFINE * 64;
  stage_base<TS, MM_RMETRIC_RANK1>(base_lds, A.rparams, A.dim);
  const int64_t chain = (int64_t)blockIdx.x * kWaves + wave;
  if (chain >= A.n_chains) return;
  const int dim = A.dim;
  double q = lane < dim ? A.pos[chain * dim + lane] : 0.0;
  double p = lane < dim ? A.mom[chain * dim + lane] : 0.0;
  double T[TS][TS];
  double acc = 0.0;
  build_metric<TS, MM_RMETRIC_RANK1>(T, q, dim, lane, w, base_lds);
  if (variant == 2) sweep_inverse<TS, false, false>(T, dim, lane, w, nullptr, nullptr);
  for (int r = 0; r < repeats; ++r) {
    if (variant == 0) {
      build_metric<TS, MM_RMETRIC_RANK1>(T, q, dim, lane, w, base_lds);
      sweep_inverse<TS, false, false>(T, dim, lane, w, nullptr, nullptr);
      acc += T[0][0];
    } else if (variant == 1) {
      build_metric<TS, MM_RMETRIC_RANK1>(T, q, dim, lane, w, base_lds);
      double u;
      eliminate_solve<TS>(T, p, lane, w, blk, &u);
      acc += u;
    } else if (variant == 2) {
      p = matvec_flat<TS>(T, p, lane, w) * 0.5 + 0.1;
      acc += p;
    } else {
      build_metric<TS, MM_RMETRIC_RANK1>(T, q, dim, lane, w, base_lds);
      acc += T[0][0];
    }
    q += 1e-9 * acc;
  }
  if (lane == 0) sink[chain] = acc;
}
#endif  // MM_DEV_KERNELS

template <int TS, int RMETRIC>
size_t lds_bytes() {
  const size_t base = (RMETRIC == MM_RMETRIC_RANK1) ? 64 * Geo<TS>::TSTRIDE : 0;
  return (base + kWaves * kWaveLdsDoubles) * sizeof(double);
}

template <int TS, int RMETRIC>
int launch_step(mm_ctx* ctx, const ImplicitArgs& a, int64_t n) {
  const unsigned blocks = (unsigned)((n + kWaves - 1) / kWaves);
  const size_t lds = lds_bytes<TS, RMETRIC>();
  hipLaunchKernelGGL((implicit_leapfrog_kernel<TS, RMETRIC>), dim3(blocks), dim3(64 * kWaves), lds,
                     ctx->stream, a);
  MM_HIP_CHECK(ctx, hipGetLastError());
  return MM_OK;
}

template <int TS, int RMETRIC>
int launch_aux(mm_ctx* ctx, const ImplicitArgs& a, int64_t n, int op) {
  const unsigned blocks = (unsigned)((n + kWaves - 1) / kWaves);
  const size_t lds = lds_bytes<TS, RMETRIC>();
  if (op == 0)
    hipLaunchKernelGGL((riemann_aux_kernel<TS, RMETRIC, 0>), dim3(blocks), dim3(64 * kWaves), lds, ctx->stream, a);
  else if (op == 1)
    hipLaunchKernelGGL((riemann_aux_kernel<TS, RMETRIC, 1>), dim3(blocks), dim3(64 * kWaves), lds, ctx->stream, a);
  else
    hipLaunchKernelGGL((riemann_aux_kernel<TS, RMETRIC, 2>), dim3(blocks), dim3(64 * kWaves), lds, ctx->stream, a);
  MM_HIP_CHECK(ctx, hipGetLastError());
  return MM_OK;
}

template <class Fn>
int dispatch(mm_ctx* ctx, const mm_model* m, Fn&& fn) {
  const int ts = (m->dim + 7) / 8;
#define MM_TS_CASE(TSV)                                                              \
  if (ts <= TSV) {                                                                   \
    if (m->rmetric == MM_RMETRIC_RANK1) return fn.template operator()<TSV, MM_RMETRIC_RANK1>(); \
    return fn.template operator()<TSV, MM_RMETRIC_DIAGQUAD>();                       \
  }
  MM_TS_CASE(1)
  MM_TS_CASE(2)
  MM_TS_CASE(4)
  MM_TS_CASE(8)
#undef MM_TS_CASE
  mm_set_error(ctx, "dense-Riemannian wave-per-chain kernels support dim <= 64");
  return MM_ERR_UNSUPPORTED;
}

ImplicitArgs make_args(const mm_model* m, mm_state* s) {
  ImplicitArgs a{};
  a.pos = s->d_pos;
  a.mom = s->d_mom;
  a.dir = s->d_dir;
  a.step_scale = s->d_step_scale;
  a.chain_steps = s->d_chain_steps;
  a.status = s->d_status;
  a.n_done = s->d_n_done;
  a.n_chains = s->n;
  a.dim = s->dim;
  a.target = m->target;
  a.tparams = m->d_target_params;
  a.rparams = m->d_rmetric_params;
  return a;
}

struct StepFn {
  mm_ctx* ctx;
  ImplicitArgs a;
  int64_t n;
  template <int TS, int RM>
  int operator()() { return launch_step<TS, RM>(ctx, a, n); }
};
template <int TS, int RMETRIC>
int launch_midpoint(mm_ctx* ctx, const ImplicitArgs& a, int64_t n) {
  const unsigned blocks = (unsigned)((n + kWaves - 1) / kWaves);
  const size_t lds = lds_bytes<TS, RMETRIC>();
  hipLaunchKernelGGL((implicit_midpoint_kernel<TS, RMETRIC>), dim3(blocks), dim3(64 * kWaves), lds,
                     ctx->stream, a);
  MM_HIP_CHECK(ctx, hipGetLastError());
  return MM_OK;
}
struct MidpointFn {
  mm_ctx* ctx;
  ImplicitArgs a;
  int64_t n;
  template <int TS, int RM>
  int operator()() { return launch_midpoint<TS, RM>(ctx, a, n); }
};
struct AuxFn {
  mm_ctx* ctx;
  ImplicitArgs a;
  int64_t n;
  int op;
  template <int TS, int RM>
  int operator()() { return launch_aux<TS, RM>(ctx, a, n, op); }
};

}  // namespace

int mm_launch_softabs_leapfrog(mm_ctx*, const mm_model*, mm_state*, double, int, const mm_fp_opts&,
                               mm_counters*);
int mm_launch_softabs_aux(mm_ctx*, const mm_model*, mm_state*, int, double*, const double*);
int mm_launch_softabs_midpoint(mm_ctx*, const mm_model*, mm_state*, double, int, const mm_fp_opts&, mm_counters*);
int mm_launch_implicit_midpoint_large(mm_ctx*, const mm_model*, mm_state*, double, int, const mm_fp_opts&,
                                      mm_counters*);
int mm_launch_implicit_large(mm_ctx*, const mm_model*, mm_state*, double, int, const mm_fp_opts&,
                             mm_counters*);
int mm_launch_riemann_aux_large(mm_ctx*, const mm_model*, mm_state*, int, double*, const double*);
int mm_launch_implicit_mfma(mm_ctx*, const mm_model*, mm_state*, double, int, const mm_fp_opts&, mm_counters*);
int mm_launch_implicit_blk16(mm_ctx*, const mm_model*, mm_state*, double, int, const mm_fp_opts&, mm_counters*);
// 279 < D <= 1024: the chain's metric in HBM (k_implicit_global.hip)
int mm_launch_implicit_global(mm_ctx*, const mm_model*, mm_state*, double, int, const mm_fp_opts&, mm_counters*);
int mm_launch_riemann_aux_global(mm_ctx*, const mm_model*, mm_state*, int, double*, const double*);
int mm_launch_implicit_midpoint_global(mm_ctx*, const mm_model*, mm_state*, double, int, const mm_fp_opts&, mm_counters*);

// a user metric: the same kernels, compiled at run time around the user's source (mm_rtc.hip)
static int launch_user_metric(mm_ctx* ctx, const mm_model* m, mm_state* s, int which, double h, int n_steps,
                              const mm_fp_opts* opts, mm_counters* d_counters, double* d_out, const double* d_z) {
  ImplicitArgs a = make_args(m, s);
  a.step_size = h;
  a.n_steps = n_steps;
  if (opts) a.opts = *opts;
  a.counters = d_counters;
  a.out = d_out;
  a.z = d_z;
  a.no_refine = mm_refine_disabled();
  a.no_dual = mm_dual_disabled();
  a.no_lowrank = mm_lowrank_disabled();  // (a user metric that declares MM_USER_LOWRANK: user_metric.h)
  a.lowrank_refresh = mm_lowrank_refresh();
  return mm_rtc_launch_riemann(ctx, m, s, which, &a);
}

int mm_launch_implicit_leapfrog(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int n_steps,
                                const mm_fp_opts& opts, mm_counters* d_counters) {
  if (m->rmetric == MM_RMETRIC_USER)
    return launch_user_metric(ctx, m, s, 0, h, n_steps, &opts, d_counters, nullptr, nullptr);
  if (m->rmetric == MM_RMETRIC_SOFTABS || m->rmetric == MM_RMETRIC_SOFTABS_USER)
    return mm_launch_softabs_leapfrog(ctx, m, s, h, n_steps, opts, d_counters);
  // D <= 32: one wave per chain, rank-1 sweep on the VALU (this file).  32 < D <= 64: one wave per chain,
  // blocked sweep on the matrix cores (k_implicit_mfma.hip).  D > 64: a team of waves shares the chain's
  // metric (k_implicit_large.hip).  MICI_AMD_IMPLICIT_KERNEL=wave|team overrides the 32 < D <= 64 choice
  // (A/B measurements, tools/sweep_chains.sh).
  static const int force = [] {
    const char* e = getenv("MICI_AMD_IMPLICIT_KERNEL");
    if (!e) return 0;
    return strcmp(e, "wave") == 0 ? 1 : strcmp(e, "team") == 0 ? 2 : 0;
  }();
  // 75 < D <= 256: team of 8 waves, metric in the CU's register file, 16-pivot blocks on the matrix cores with
  // solve-only constructions as a blocked LDL^T (k_implicit_blk16.hip).  MICI_AMD_IMPLICIT_KERNEL=team selects the
  // VALU team kernel.  (The round-1 4-pivot kernel and the look-ahead variant of round 2 lost their A/B runs and were
  // removed in round 3; they are in the history: k_implicit_mfma_team.hip, k_implicit_blk16la.hip.)
  if (m->dim > 279) return mm_launch_implicit_global(ctx, m, s, h, n_steps, opts, d_counters);
  if (m->dim > 75 && m->dim <= 256 && force != 2)
    return mm_launch_implicit_blk16(ctx, m, s, h, n_steps, opts, d_counters);
  if (m->dim > 64 || (m->dim > 32 && force == 2))
    return mm_launch_implicit_large(ctx, m, s, h, n_steps, opts, d_counters);
  if (m->dim > 32 && force == 0) return mm_launch_implicit_mfma(ctx, m, s, h, n_steps, opts, d_counters);
  ImplicitArgs a = make_args(m, s);
  a.step_size = h;
  a.n_steps = n_steps;
  a.opts = opts;
  a.counters = d_counters;
  a.no_refine = mm_refine_disabled();
  a.no_dual = mm_dual_disabled();
  a.no_lowrank = mm_lowrank_disabled();  // (round 6: the Woodbury path of the built-in rank-one-update metric)
  a.lowrank_refresh = mm_lowrank_refresh();
  return dispatch(ctx, m, StepFn{ctx, a, s->n});
}

int mm_launch_implicit_midpoint_riemann(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int n_steps,
                                        const mm_fp_opts& opts, mm_counters* d_counters) {
  if (m->rmetric == MM_RMETRIC_USER)
    return launch_user_metric(ctx, m, s, 1, h, n_steps, &opts, d_counters, nullptr, nullptr);
  if (m->rmetric == MM_RMETRIC_SOFTABS || m->rmetric == MM_RMETRIC_SOFTABS_USER)
    return mm_launch_softabs_midpoint(ctx, m, s, h, n_steps, opts, d_counters);
  if (m->dim > 279) return mm_launch_implicit_midpoint_global(ctx, m, s, h, n_steps, opts, d_counters);
  if (m->dim > 64) return mm_launch_implicit_midpoint_large(ctx, m, s, h, n_steps, opts, d_counters);
  ImplicitArgs a = make_args(m, s);
  a.step_size = h;
  a.n_steps = n_steps;
  a.opts = opts;
  a.counters = d_counters;
  a.no_refine = mm_refine_disabled();
  a.no_lowrank = mm_lowrank_disabled();  // (round 6: the held inverse carried between the midpoint rule's evaluations)
  a.lowrank_refresh = mm_lowrank_refresh();
  return dispatch(ctx, m, MidpointFn{ctx, a, s->n});
}

int mm_launch_riemann_aux(mm_ctx* ctx, const mm_model* m, mm_state* s, int op, double* d_out,
                          const double* d_z) {
  if (m->rmetric == MM_RMETRIC_USER) return launch_user_metric(ctx, m, s, 2 + op, 0.0, 0, nullptr, nullptr, d_out, d_z);
  if (m->rmetric == MM_RMETRIC_SOFTABS || m->rmetric == MM_RMETRIC_SOFTABS_USER)
    return mm_launch_softabs_aux(ctx, m, s, op, d_out, d_z);
  if (m->dim > 279) return mm_launch_riemann_aux_global(ctx, m, s, op, d_out, d_z);
  if (m->dim > 64) return mm_launch_riemann_aux_large(ctx, m, s, op, d_out, d_z);
  ImplicitArgs a = make_args(m, s);
  a.out = d_out;
  a.z = d_z;
  return dispatch(ctx, m, AuxFn{ctx, a, s->n, op});
}

#ifdef MM_DEV_KERNELS
// developer hook (tools/ubench_primitives.py): time `repeats` repetitions of one primitive per chain
extern "C" __attribute__((visibility("default"))) int mm_debug_primitive_bench(mm_ctx* ctx, const mm_model* m, mm_state* s, int variant,
                                        int repeats, double* ms) {
  if (!ctx || !m || !s || m->dim > 64 || m->dim <= 32 || m->rmetric != MM_RMETRIC_RANK1) return MM_ERR_INVALID;
  ImplicitArgs a = make_args(m, s);
  const unsigned blocks = (unsigned)((s->n + kWaves - 1) / kWaves);
  const size_t lds = lds_bytes<8, MM_RMETRIC_RANK1>();
  hipEvent_t e0, e1;
  MM_HIP_CHECK(ctx, hipEventCreate(&e0));
  MM_HIP_CHECK(ctx, hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    MM_HIP_CHECK(ctx, hipEventRecord(e0, ctx->stream));
    hipLaunchKernelGGL((debug_primitive_kernel<8>), dim3(blocks), dim3(64 * kWaves), lds, ctx->stream, a,
                       variant, repeats, s->d_scratch);
    MM_HIP_CHECK(ctx, hipEventRecord(e1, ctx->stream));
    MM_HIP_CHECK(ctx, hipEventSynchronize(e1));
  }
  float f = 0.f;
  MM_HIP_CHECK(ctx, hipEventElapsedTime(&f, e0, e1));
  *ms = f;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return MM_OK;
}
#endif  // MM_DEV_KERNELS
