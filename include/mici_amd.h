/* mici_amd.h - C ABI of libmici_amd.so: the MI355X (gfx950) symplectic-integrator hot path.
 *
 * The reference (matt-graham/mici) is pure Python and has no FFI; the interface this library sits
 * behind is the duck-type `Integrator.step(state) -> state` / `System.h / dh_dmom / sample_momentum`
 * consumed by mici.transitions / mici.adapters / mici.samplers.  Each entry point below cites the
 * reference symbol (file:line under /root/reference/src/mici) whose arithmetic it replaces, batched
 * over N independent chains.  Plain pointers and sizes only; no callbacks; all numerics fp64.
 *
 * Conventions
 *  - every function returns 0 on success or a negative mm_rc; mm_last_error() describes the failure.
 *  - host pointers are borrowed for the duration of the call only.
 *  - a mm_ctx owns one HIP stream on one device; launches are asynchronous on that stream, the
 *    download / scalar-returning calls synchronise it.  A ctx is not thread safe; distinct ctxs are
 *    independent (one host thread or process per GPU).
 *  - chain state is row-major pos[N][D], mom[N][D] (fp64) and dir[N] (int8, +1/-1): the batched form
 *    of the reference ChainState(pos, mom, dir) (states.py:160-305).
 *  - per-chain failures do not fail the call: status[N] carries the mm_status of the first failed
 *    step and n_done[N] the number of completed steps; a failed chain is frozen at its last good
 *    (pos, mom) exactly as mici.transitions leaves it (transitions.py:292-295).
 */
#ifndef MICI_AMD_H
#define MICI_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* libmici_amd.so is built with -fvisibility=hidden: what this header declares is the whole export list */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define MM_ABI_VERSION 4

typedef struct mm_ctx mm_ctx;
typedef struct mm_model mm_model;
typedef struct mm_state mm_state;
typedef struct mm_comm mm_comm;

typedef enum mm_rc {
  MM_OK = 0,
  MM_ERR_INVALID = -1,     /* bad argument                                  */
  MM_ERR_HIP = -2,         /* HIP runtime error (message in mm_last_error)  */
  MM_ERR_UNSUPPORTED = -3, /* model / size combination has no device kernel */
  MM_ERR_NOMEM = -4,
  MM_ERR_RCCL = -5
} mm_rc;

/* Built-in synthetic models (closed forms: SURVEY.md section 8d / Appendix A).  The reference takes
 * arbitrary Python callables (systems.py:107,119,788,792,1332,1358,1888,1920); a device kernel needs
 * device-side derivatives, so targets / metrics / constraints are an enum with packed fp64 params. */
typedef enum mm_target {
  MM_TARGET_GAUSS_ISO = 0,   /* l = |q|^2/2                                       params: -        */
  MM_TARGET_GAUSS_DIAG = 1,  /* l = sum prec_i q_i^2 / 2                          params: prec[D]  */
  MM_TARGET_GAUSS_DENSE = 2, /* l = q^T P q / 2                                   params: P[D*D]   */
  MM_TARGET_POLY = 3,        /* l = a sum q^2/2 + b sum q^4/4                     params: a, b     */
  MM_TARGET_BANANA = 4,      /* l = sum (1-q_i)^2/20 + sum (q_{i+1}-q_i^2)^2      params: -        */
  MM_TARGET_FUNNEL = 5,      /* scaled funnel, q=(v,x): v^2/18 + n v/2 + e^-v sum w x^2 / 2  params: w[D-1] */
  MM_TARGET_TORUS = 6,       /* README torus density, D=3                         params: R, r, alpha */
  MM_TARGET_USER = 100       /* user-supplied device code: mm_model_create_from_source    params: any      */
} mm_target;

typedef enum mm_metric_kind { /* fixed (Euclidean) metric, systems.py:327-345 */
  MM_METRIC_IDENTITY = 0,
  MM_METRIC_DIAG = 1,  /* metric[D]   (matrices.py:771-792)  */
  MM_METRIC_DENSE = 2  /* metric[D*D] (matrices.py:1191-1216) */
} mm_metric_kind;

typedef enum mm_rmetric { /* position-dependent metric of a RiemannianMetricSystem */
  MM_RMETRIC_NONE = 0,
  MM_RMETRIC_RANK1 = 1,    /* M(q) = B + q q^T / D, params B[D*D]; DenseRiemannianMetricSystem     */
  MM_RMETRIC_DIAGQUAD = 2, /* M(q) = diag(1 + q^2) held dense;     DenseRiemannianMetricSystem     */
  MM_RMETRIC_SOFTABS = 3,  /* SoftAbs of the target Hessian, params coeff; SoftAbsRiemannianMetricSystem */
  MM_RMETRIC_USER = 100,   /* user-supplied device code: mm_model_create_from_source, dim <= 279, params: any */
  MM_RMETRIC_SOFTABS_USER = 101 /* SoftAbs of a USER Hessian (hess_neg_log_dens / mtp_neg_log_dens as device code, dense):
                                   mm_model_create_from_source, dim <= 256, params coeff then the user's own */
} mm_rmetric;

typedef enum mm_constr { /* holonomic constraint, C = 1 */
  MM_CONSTR_NONE = 0,
  MM_CONSTR_TORUS = 1,  /* (sqrt(x^2+y^2)-R)^2 + z^2 - r^2, params R, r */
  MM_CONSTR_FIRST = 2,  /* q_0                                          */
  MM_CONSTR_CIRCLE = 3, /* q_0^2 + q_1^2 - 1                            */
  MM_CONSTR_LINEAR = 4, /* A q - b, C rows: params A[C*D] (row-major) then b[C]; C = n_constr_params / (D + 1) <= 8 */
  MM_CONSTR_SPHERE_PLANE = 5, /* two constraints: |q|^2 - 1 and n . q; params n[D], D >= 3 */
  MM_CONSTR_SPHERE = 6, /* |q|^2 - 1 (the constrained system of the reference's adapter tests), D >= 2 */
  MM_CONSTR_USER = 100  /* user-supplied device code: mm_model_create_from_source, desc->n_constr rows, params: any */
} mm_constr;

/* Per-chain status: which reference exception the failed step would have raised (errors.py:6-35). */
typedef enum mm_status {
  MM_ST_OK = 0,
  MM_ST_DIVERGED = 1,       /* ConvergenceError: solver diverged            (solvers.py:80-84, 446-451) */
  MM_ST_MAX_ITERS = 2,      /* ConvergenceError: max_iters exhausted        (solvers.py:93-94, 465-469) */
  MM_ST_SOLVER_LINALG = 3,  /* ConvergenceError: LinAlg/ValueError in solver (solvers.py:89-92, 461-464) */
  MM_ST_NON_REVERSIBLE = 4, /* NonReversibleStepError    (integrators.py:510-515, 523-528, 974-979) */
  MM_ST_LINALG = 5          /* LinAlgError outside a solver (matrices.py:211-215, 1170-1172)        */
} mm_status;

typedef enum mm_norm { MM_NORM_LINF = 0, MM_NORM_L2 = 1 } mm_norm;           /* solvers.py:20-27 */
typedef enum mm_fp_solver { MM_FP_DIRECT = 0, MM_FP_STEFFENSEN = 1 } mm_fp_solver; /* solvers.py:47-154 */
typedef enum mm_proj_solver {      /* solvers.py:195-343, 346-469, 472-614 */
  MM_PROJ_NEWTON = 0,
  MM_PROJ_QUASI_NEWTON = 1,
  MM_PROJ_NEWTON_LINE_SEARCH = 2
} mm_proj_solver;

typedef struct mm_model_desc {
  int32_t dim;
  int32_t target;
  const double* target_params;
  size_t n_target_params;
  int32_t metric_kind; /* fixed metric (Euclidean / constrained systems) */
  int32_t gaussian_split; /* 1 = GaussianEuclideanMetricSystem (systems.py:369-474): the target is a density
                           * with respect to the standard Gaussian, h2 = q.q/2 + p.M^-1 p/2 and h2_flow is the
                           * exact rotation in the metric's eigenbasis (systems.py:464-474).  With a constraint
                           * this is GaussianDenseConstrainedEuclideanMetricSystem (systems.py:1034-1184), which
                           * implies dens_wrt_ambient.  Not for Riemannian systems. */
  const double* metric;
  size_t n_metric;
  int32_t rmetric; /* position-dependent metric (Riemannian systems) */
  int32_t n_constr; /* MM_CONSTR_USER: the number of constraint functions (1..8, < dim); ignored (may be 0) for the
                     * built-in constraints, whose count follows from their kind.  Sits in what used to be padding. */
  const double* rmetric_params;
  size_t n_rmetric_params;
  int32_t constr;
  int32_t dens_wrt_ambient; /* constrained systems: 1 = dens_wrt_hausdorff=False, h1 includes the half
                             * log-determinant of the Gram matrix (systems.py:829-831, 846-862, 1024-1031);
                             * 0 = the reference default dens_wrt_hausdorff=True */
  const double* constr_params;
  size_t n_constr_params;
} mm_model_desc;

/* ImplicitLeapfrogIntegrator ctor defaults (integrators.py:438-446) + solve_fixed_point_* kwargs
 * (solvers.py:47-54). */
typedef struct mm_fp_opts {
  double conv_tol;   /* 1e-9  */
  double div_tol;    /* 1e10  */
  int32_t max_iters; /* 100   */
  int32_t norm;      /* MM_NORM_LINF */
  int32_t solver;    /* MM_FP_DIRECT */
  int32_t rev_norm;  /* MM_NORM_LINF */
  double rev_tol;    /* 2e-8  */
} mm_fp_opts;

/* ConstrainedLeapfrogIntegrator ctor defaults (integrators.py:855-864) + Newton kwargs
 * (solvers.py:346-357). */
typedef struct mm_proj_opts {
  double constr_tol; /* 1e-9 */
  double pos_tol;    /* 1e-8 */
  double div_tol;    /* 1e10 */
  int32_t max_iters; /* 50   */
  int32_t norm;      /* MM_NORM_LINF */
  int32_t solver;    /* MM_PROJ_NEWTON */
  int32_t rev_norm;  /* MM_NORM_LINF */
  double rev_tol;    /* 2e-8 */
  int32_t n_inner;   /* 1    */
  int32_t max_line_search_iters; /* 10 (line-search solver only, solvers.py:482) */
} mm_proj_opts;

/* Work counters summed over chains (the reference's ChainState._call_counts, states.py:204-212;
 * they are the n_M / n_B denominators of SURVEY.md section 8d). */
typedef struct mm_counters {
  int64_t n_grad;         /* grad_neg_log_dens evaluations              */
  int64_t n_metric;       /* metric constructions (factorisations)      */
  int64_t n_inverse;      /* explicit inverses                          */
  int64_t n_fp_evals;     /* fixed-point function evaluations           */
  int64_t n_fp_solves;    /* fixed-point solves                         */
  int64_t n_newton_iters; /* projection-solver iterations               */
  int64_t n_constr;       /* constraint + Jacobian evaluations          */
  int64_t n_eigh;         /* SoftAbs: eigendecompositions executed      */
  /* Work the device actually executed for the n_metric constructions (round 3): the solve-only constructions of the
   * position fixed points are refined from the explicit inverse at the step's start (implicit_core.h) instead of
   * being factorised.  n_factor_full + n_factor_solve + (refined constructions) = constructions executed. */
  int64_t n_refine;       /* preconditioned-CG product pairs (M(x) v, M(x0)^-1 v); SoftAbs: decompositions obtained
                           * by refining the previous eigenvectors (no Jacobi sweep) */
  int64_t n_factor_full;  /* full sweeps: factorisation + explicit inverse */
  int64_t n_factor_solve; /* trailing sweeps: LDL^T factorisation + one substitution */
  int64_t n_mfma_products; /* SoftAbs, D <= 64: 64^3 products run on the matrix cores (refinement passes, G = H V,
                            * B = A J of grad_quadratic_form_inv) */
  int64_t n_lowrank;      /* round 6: solve-only constructions of the built-in rank-one-update metric obtained from the
                           * explicit inverse at the step's start by the Woodbury identity (one product each; no CG pair,
                           * no factorisation).  MICI_AMD_LOWRANK=0 routes them through the CG refinement instead. */
  int64_t n_inverse_update; /* round 6: explicit inverses of that metric carried to the step's new position by the symmetric
                             * rank-two update of the held inverse instead of a full sweep (n_factor_full counts the sweeps
                             * that did run: one per launch + one every MICI_AMD_LOWRANK_REFRESH steps + the fallbacks) */
} mm_counters;

/* ---- library / context ------------------------------------------------------------------------- */
int mm_abi_version(void);
int mm_device_count(int* count);
int mm_ctx_create(int device, mm_ctx** out);
int mm_ctx_destroy(mm_ctx* ctx);
int mm_ctx_sync(mm_ctx* ctx);
/* ctx may be NULL: returns the last error of a failed mm_ctx_create / argument check on this thread */
const char* mm_last_error(const mm_ctx* ctx);
/* HIP-event timing on the ctx stream (bench.py's kernel timing): record slot 0..15, then query. */
int mm_ctx_record(mm_ctx* ctx, int slot);
int mm_ctx_elapsed_ms(mm_ctx* ctx, int slot_begin, int slot_end, double* ms);

/* ---- model ------------------------------------------------------------------------------------------ */
int mm_model_create(mm_ctx* ctx, const mm_model_desc* desc, mm_model** out);
/* A model with USER CODE in it - the reference's callable constructor arguments for a device.  `hip_source` is HIP C++
 * text, compiled for gfx950 with hipRTC when the model is created (compile errors come back through mm_last_error):
 *  * desc->target == MM_TARGET_USER - `neg_log_dens` / `grad_neg_log_dens` (systems.py:107, 119): the source defines
 *        __device__ double mm_user_grad(const double* q, int i, int dim, const double* params);     // d nld / d q_i
 *        __device__ double mm_user_nld_term(const double* q, int i, int dim, const double* params); // nld = sum_i of it
 *    (q: the chain's whole position vector; params: desc->target_params).  On an EuclideanMetricSystem (identity /
 *    diagonal / dense fixed metric) the wave-per-chain kernels are compiled around it: mm_leapfrog_euclid,
 *    mm_composition_euclid, mm_hamiltonian, mm_dh_dmom, mm_sample_momentum, mm_momentum_refresh*, mm_metropolis_accept*.
 *  * desc->constr == MM_CONSTR_USER - `constr` / `jacob_constr` of a ConstrainedEuclideanMetricSystem
 *    (systems.py:786-792), desc->n_constr functions of the position, 1 <= n_constr <= 8, n_constr < dim <= 256:
 *        __device__ void mm_user_constr(const double* q, int dim, const double* params, double* c);   // c[n_constr]
 *        __device__ void mm_user_jacob(const double* q, int dim, const double* params, double* jac);  // jac[k*dim+i]
 *    (params: desc->constr_params) and, only for dens_wrt_ambient / Gaussian-split systems (mhp_constr,
 *    systems.py:1006-1008: out[i] = sum_{k,j} m[k*dim+j] d2 c_k / dq_j dq_i),
 *        __device__ void mm_user_mhp_constr(const double* q, int dim, const double* params, const double* m, double* out);
 *    The library compiles its constrained-leapfrog core (csrc/constrained_core.h, lane per chain, dim <= 64;
 *    csrc/constrained_wave.h, wave per chain, beyond: all three projection solvers, n_inner, both density
 *    conventions) around them; mm_constrained_leapfrog, mm_hamiltonian, mm_sample_momentum,
 *    mm_momentum_refresh* and mm_metropolis_accept* work on such a model.  Target and constraint may both be user code
 *    (one source text defining all the functions).
 *  * desc->rmetric == MM_RMETRIC_USER - `metric_func` / `vjp_metric_func` of a DenseRiemannianMetricSystem
 *    (systems.py:1322-1358, 1690-1734), dim <= 1024:
 *        __device__ double mm_user_metric(const double* q, int i, int j, int dim, const double* params);  // M(q)_ij
 *        __device__ double mm_user_vjp(const double* q, const MmMat& V, int k, int dim, const double* params);
 *        // element k of vjp_metric_func(q)(V) = sum_ij V(i, j) d M_ij / d q_k;  V(i, j) reads the symmetric argument
 *    (params: desc->rmetric_params).  The text may opt into `#define MM_USER_AUX n` + mm_user_prepare (per-point
 *    precomputation shared by all entries) and `#define MM_USER_VJP_FLAT` + mm_user_vjp_flat (the vector-Jacobian
 *    product in team form, running on the backend's own mat-vec): csrc/user_metric.h states both protocols.  The
 *    library compiles its dense-Riemannian backends around the text: the leapfrog step on the matrix-core kernels
 *    (32 < dim <= 64 one wave per chain, 75 < dim <= 256 one workgroup per chain; compiled on first use) with the
 *    solve-only metric constructions refined through M(x) v products of the user's entries, everything else on the
 *    wave (dim <= 64) / team (dim <= 279) kernels; 279 < dim <= 1024 on the global-memory tier
 *    (csrc/implicit_global.h: the chain's metric in HBM).  mm_implicit_leapfrog, mm_implicit_midpoint, mm_hamiltonian,
 *    mm_dh_dmom, mm_sample_momentum, mm_momentum_refresh* and mm_metropolis_accept* work on such a model; the target
 *    may be built in or user code as well.  Compiled code objects are cached under MICI_AMD_RTC_CACHE (default
 *    ~/.cache/mici_amd/rtc; "off" disables); `python -m mici_amd.precompile` produces them ahead of time, without a GPU.
 *  * desc->rmetric == MM_RMETRIC_SOFTABS_USER - `hess_neg_log_dens` / `mtp_neg_log_dens` of a
 *    SoftAbsRiemannianMetricSystem (systems.py:1737-1920), dim <= 256; rmetric_params = softabs_coeff, then the user's:
 *        __device__ double mm_user_hess(const double* q, int i, int j, int dim, const double* params);   // H(q)_ij
 *        __device__ double mm_user_mtp(const double* q, const MmMat& M, int k, int dim, const double* params);
 *        // element k of mtp_neg_log_dens(q)(M) = sum_ij M(i, j) d3 nld / dq_i dq_j dq_k;  M(i, j) reads the symmetric argument
 *    Nothing is assumed about the Hessian's structure (csrc/user_hessian.h): the eigendecompositions and the
 *    V f(lambda) V^T arguments of the matrix-Tressian product are formed in full on the matrix cores.  The target is
 *    built in or user code; the same entry points as for the built-in SoftAbs system work on such a model. */
int mm_model_create_from_source(mm_ctx* ctx, const mm_model_desc* desc, const char* hip_source, mm_model** out);
int mm_model_destroy(mm_model* model);

/* ---- chain state: batched ChainState(pos, mom, dir) (states.py:160-305) ---------------------------- */
int mm_state_alloc(mm_ctx* ctx, int64_t n_chains, int32_t dim, mm_state** out);
/* As mm_state_alloc, with pos / mom / dir / status / n_done in pinned host memory that the kernels access in
 * place: upload and download become plain host copies around a stream synchronisation.  For small, long-lived
 * batches (n * dim <= 65536) -- the reused single-state buffer behind Integrator.step (integrators.py:63-80);
 * allocation itself is slow. */
int mm_state_alloc_mapped(mm_ctx* ctx, int64_t n_chains, int32_t dim, mm_state** out);
int mm_state_free(mm_state* state);
int mm_state_upload(mm_state* state, const double* pos, const double* mom, const int8_t* dir);
int mm_state_download(mm_state* state, double* pos, double* mom, int8_t* dir);
/* status[N] / n_done[N] of the last implicit / constrained call on this state */
int mm_state_download_status(mm_state* state, int32_t* status, int32_t* n_done);
/* pos, mom, dir, status and n_done in one transfer (any pointer may be NULL).  A small batch (<= 1 MiB of
 * state) moves through a pinned mirror in a single copy each way: the transfer pattern of the reference's
 * single-state Integrator.step (integrators.py:63-80), which is latency- not bandwidth-bound. */
int mm_state_download_all(mm_state* s, double* pos, double* mom, int8_t* dir, int32_t* status, int32_t* n_done);
/* raw device pointers (zero-copy interop / RCCL): any of the outputs may be NULL */
int mm_state_device_ptrs(mm_state* state, double** pos, double** mom, int8_t** dir);
/* A state from mm_state_alloc_mapped lives in pinned host memory: these are HOST-addressable pointers to its pos[N*D],
 * mom[N*D], dir[N], status[N], n_done[N] (any output may be NULL).  A caller that steps one state at a time
 * (Integrator.step, integrators.py:63-80) writes its inputs there, launches, calls mm_ctx_sync and reads the results
 * in place - no upload / download calls.  The memory may only be touched while no launch on it is in flight.
 * MM_ERR_INVALID for a state that is not mapped. */
int mm_state_mapped_ptrs(mm_state* state, double** pos, double** mom, int8_t** dir, int32_t** status, int32_t** n_done);

/* Per-chain step sizes (step-size adaptation runs every chain at its own step size; the reference keeps one
 * integrator copy per chain, adapters.py:322-340, samplers.py:1124-1129): scale[N] (host) multiplies the
 * step_size argument of every integrator entry point for this state, i.e. pass step_size = 1 and the step
 * sizes as scale.  NULL removes the factors.  mm_state_copy propagates them. */
int mm_state_set_step_scale(mm_state* state, const double* scale);
/* Per-chain trajectory lengths: with steps[N] set, every integrator call on this state advances chain i by
 * min(n_steps, steps[i]) steps (n_done reports it); NULL removes them.  This is how one launch serves
 * MetropolisRandomIntegrationTransition, which draws n_step per chain and per transition
 * (transitions.py:355-402). */
int mm_state_set_chain_steps(mm_state* s, const int32_t* steps);

/* Sticky per-chain error word of device-resident transitions (mm_metropolis_accept / _rng): bit k of errors[i] is set
 * when a proposal of chain i ended with status k (1..5, section "status codes") since the word was last cleared.  The
 * reference raises LinAlgError out of the sampler for status 5 (transitions.py:292-295 catches IntegratorError only) and
 * records the others as rejections; a transition that downloads nothing per step (stats off) reads this at its next
 * synchronisation point instead.  errors: host [N]; clear != 0 resets the word after reading. */
int mm_state_download_errors(mm_state* s, uint32_t* errors, int32_t clear);

/* device-to-device copy of pos, mom, dir, status, n_done (same n_chains and dim): the proposal copy of
 * Integrator.step / state.copy() (integrators.py:78, states.py:263-279) without a host round trip */
int mm_state_copy(mm_state* dst, const mm_state* src);

/* ---- the hot path ------------------------------------------------------------------------------------ */
/* LeapfrogIntegrator.step x n_steps on an EuclideanMetricSystem (integrators.py:63-80, 170-173;
 * systems.py:143-152, 352-363): p -= t/2 grad(q); q += t M^-1 p; p -= t/2 grad(q), t = dir*step_size,
 * with the end-of-step gradient reused by the next step (state cache, states.py:136-153). */
int mm_leapfrog_euclid(mm_ctx* ctx, const mm_model* model, mm_state* state, double step_size,
                       int32_t n_steps);

/* SymmetricCompositionIntegrator.step x n_steps on an EuclideanMetricSystem (integrators.py:176-274; the
 * BCSS two/three/four-stage integrators :277-378 are instances): the FULL coefficient sequence
 * coeffs[0..n_coeffs) (a_0, b_1, a_1, ..., a_S as built by integrators.py:258-268, n_coeffs odd,
 * <= MM_MAX_COMPOSITION_COEFFS) alternates h1_flow(c t) (systems.py:143-152) and h2_flow(c t)
 * (systems.py:362-363), starting with h1 iff initial_h1_flow_step != 0; t = dir * step_size. */
#define MM_MAX_COMPOSITION_COEFFS 16
int mm_composition_euclid(mm_ctx* ctx, const mm_model* model, mm_state* state, double step_size,
                          int32_t n_steps, int32_t n_coeffs, const double* coeffs,
                          int32_t initial_h1_flow_step);

/* ImplicitLeapfrogIntegrator.step x n_steps on a Dense / SoftAbs RiemannianMetricSystem
 * (integrators.py:493-544; solvers.py:47-154; systems.py:1360-1402; matrices.py:1161-1188, 1631-1685).
 * opts == NULL selects the reference defaults. counters may be NULL.
 * On a plain EuclideanMetricSystem (the reference runs this integrator on any System, tests/test_integrators.py:435-462)
 * the step reduces to the explicit composition A(t) C(t) C(t) A(t) and is run as such: every chain reports status 0, the
 * solver options other than max_iters >= 2 are not consulted, and a state whose drift |t M^-1 p| exceeds divergence_tol
 * or is not finite - where the reference raises ConvergenceError - is integrated on instead of being frozen. */
int mm_implicit_leapfrog(mm_ctx* ctx, const mm_model* model, mm_state* state, double step_size,
                         int32_t n_steps, const mm_fp_opts* opts, mm_counters* counters);

/* ImplicitMidpointIntegrator.step x n_steps (integrators.py:547-681): implicit Euler half step solved as a
 * fixed point in the concatenated (pos, mom) vector (solvers.py:47-154), explicit Euler half step, and the
 * reversibility check.  Euclidean-metric systems and dense-Riemannian systems (dim <= 1024), SoftAbs systems
 * (dim <= 256, user Hessians included); opts / counters as for mm_implicit_leapfrog. */
int mm_implicit_midpoint(mm_ctx* ctx, const mm_model* model, mm_state* state, double step_size,
                         int32_t n_steps, const mm_fp_opts* opts, mm_counters* counters);

/* ConstrainedLeapfrogIntegrator.step x n_steps on a DenseConstrainedEuclideanMetricSystem
 * (integrators.py:929-984; solvers.py:429-469; systems.py:786-873, 1010-1022). */
int mm_constrained_leapfrog(mm_ctx* ctx, const mm_model* model, mm_state* state, double step_size,
                            int32_t n_steps, const mm_proj_opts* opts, mm_counters* counters);

/* ---- what mici.transitions needs around the path ---------------------------------------------------- */
/* System.h(state) per chain (systems.py:187-196, 348-350, 1375-1379, 850-853); NaN where the
 * reference would raise LinAlgError. h is a host array of N doubles. */
int mm_hamiltonian(mm_ctx* ctx, const mm_model* model, mm_state* state, double* h);
/* System.dh_dmom(state) = M^-1 p per chain (systems.py:352-354, 1398-1399); out is host [N][D]. */
int mm_dh_dmom(mm_ctx* ctx, const mm_model* model, mm_state* state, double* out);
/* System.sample_momentum with the standard-normal draw z supplied by the host RNG:
 * mom = M^{1/2} z (systems.py:365-366, 1401-1402), then projected onto the cotangent space for a
 * constrained system (systems.py:614-616). z is host [N][D]. */
int mm_sample_momentum(mm_ctx* ctx, const mm_model* model, mm_state* state, const double* z);

/* ---- momentum transitions (transitions.py:129-198) ---------------------------------------------------------
 * IndependentMomentumTransition (coeff == 1): mom = sample_momentum(state, z), i.e. mm_sample_momentum.
 * CorrelatedMomentumTransition (0 <= coeff < 1; Horowitz 1991): mom = sqrt(1 - coeff^2) mom + coeff mom_ind
 * with mom_ind = sample_momentum(state, z) (transitions.py:190-196).  z[N*D]: standard-normal draws (host). */
int mm_momentum_refresh(mm_ctx* ctx, const mm_model* model, mm_state* state, const double* z, double coeff);

/* ---- Metropolis accept step of an integration transition, device resident ------
 * MetropolisIntegrationTransition._sample_n_step after the trajectory (transitions.py:275-315): `state` is
 * the chain state BEFORE the trajectory, `proposal` a copy of it (mm_state_copy) advanced by one of the
 * integrator entry points above (its status / n_done are read here).  Per chain:
 *   integration_error = status != 0;  moved = n_done > 0 (else the proposal IS the state, accept_prob 0);
 *   h_diff = h(state) - h(proposal);  accept_prob = isnan(h_diff) ? 0 : exp(min(0, h_diff));
 *   accepted = !integration_error && u < accept_prob;
 *   accepted: state <- proposal's pos, mom (dir unchanged: negated for the involution, negated again);
 *   rejected: state.dir = -state.dir.
 * u[N]: uniform(0,1) draws (host).  accept_prob[N] / accepted[N] (host) may be NULL.  The statistics
 * "metrop_accept_prob" = accept_prob, "accept_stat" = integration_error ? 0 : accept_prob, "n_step" = n_done,
 * "convergence_error" / "non_reversible_step" follow from the proposal's status (mm_state_download_status). */
int mm_metropolis_accept(mm_ctx* ctx, const mm_model* model, mm_state* state, mm_state* proposal,
                         const double* u, double* accept_prob, int8_t* accepted);

/* ---- device-side random draws ("removes the host round-trip per trajectory") ------------
 * The reference draws from a NumPy Generator on the host: a standard-normal vector per momentum refresh
 * (transitions.py:136-142 -> systems.py:365-366, 1401-1402), a uniform per Metropolis accept step
 * (transitions.py:300-309) and an integer per trajectory of MetropolisRandomIntegrationTransition
 * (transitions.py:383-386).  The entry points above take those draws from the host (parity mode: the caller's
 * Generator decides every number).  With mm_state_set_rng the draws of chain i of a state become a pure function of
 * (seed, chain_offset + i, transition, purpose, position) - Philox4x32-10, counter-based - evaluated on the device:
 * nothing is uploaded per transition, and the stream of a chain does not depend on how chains are sharded over GPUs
 * or batched.  `transition` is the caller's transition counter (< 2^40).  oracle/rng.py restates the generator. */
int mm_state_set_rng(mm_state* state, uint64_t seed, uint64_t chain_offset);
/* mm_momentum_refresh with z ~ N(0, I) drawn on the device; asynchronous on the ctx stream. */
int mm_momentum_refresh_rng(mm_ctx* ctx, const mm_model* model, mm_state* state, double coeff, uint64_t transition);
/* mm_metropolis_accept with u ~ U[0, 1) drawn on the device; accept_prob / accepted may be NULL, in which case the
 * call does not synchronise.  (The reference draws no uniform after an integration error; here the draw of such a
 * chain is simply unused - streams are per chain and per transition, so nothing shifts.) */
int mm_metropolis_accept_rng(mm_ctx* ctx, const mm_model* model, mm_state* state, mm_state* proposal,
                             uint64_t transition, double* accept_prob, int8_t* accepted);
/* Per-chain trajectory lengths n_step ~ U{lo, ..., hi - 1} drawn on the device into the state's chain-step counts
 * (see mm_state_set_chain_steps): MetropolisRandomIntegrationTransition's rng.integers(*n_step_range). */
int mm_rng_chain_steps(mm_state* state, uint64_t transition, int32_t lo, int32_t hi);
/* The raw draws of `transition` to host arrays z[N][D], u[N], steps[N] (any may be NULL): reproducibility checks. */
int mm_rng_draws(mm_state* state, uint64_t transition, double* z, double* u, int32_t* steps, int32_t lo, int32_t hi);

/* ---- multi-GPU: chains are sharded, no collective inside integration; one RCCL all-gather over
 * xGMI per trace collection (the role of the reference's process pool + memmaps,
 * samplers.py:668-772, 104-138). ------------------------------------------------------------------- */
#define MM_COMM_ID_BYTES 128
int mm_comm_unique_id(uint8_t id[MM_COMM_ID_BYTES]); /* rank 0 creates, host side distributes */
int mm_comm_create(mm_ctx* ctx, int32_t n_ranks, int32_t rank, const uint8_t id[MM_COMM_ID_BYTES],
                   mm_comm** out);
int mm_comm_destroy(mm_comm* comm);
/* Ranks the communicator spans and this process' rank in it as RCCL reports them (ncclCommCount / ncclCommUserRank);
 * either output may be NULL.  The reference's counterpart is the number of worker processes its pool really started
 * (samplers.py:668-772); bench.py prints it as `config.n_ranks_seen`. */
int mm_comm_count(mm_comm* comm, int32_t* n_ranks, int32_t* rank);
/* Gather every rank's pos shard ([n_local][D], equal n_local on all ranks) into host buffer
 * pos_all[n_ranks*n_local][D] on every rank (rank-major = global chain order). */
int mm_comm_allgather_pos(mm_comm* comm, mm_state* state, double* pos_all);
/* Overlapped form: snapshot the shard (device-to-device) on the ctx stream, then all-gather it on the
 * communicator's own HIP stream - and, if want_host != 0, copy the gathered [n_ranks*n_local][D] array
 * to a pinned host buffer - while the ctx stream goes on integrating the next trajectory.
 * mm_comm_wait blocks until the last enqueued gather has landed; pos_all may be NULL (device-side
 * gather only, e.g. on ranks that do not write traces). */
int mm_comm_allgather_pos_async(mm_comm* comm, mm_state* state, int want_host);
int mm_comm_wait(mm_comm* comm, double* pos_all);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif
#endif /* MICI_AMD_H */
