// Internal definitions shared by the C-ABI translation unit and the HIP kernels (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <mutex>
#include <string>

#include "../../include/mici_amd.h"

struct mm_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t events[16] = {};
  int n_cu = 0;
  std::string last_error;
  mm_counters* d_counters = nullptr;  // device scratch for work counters
  char* h_stage = nullptr;  // pinned staging buffer (kStageLimit bytes) shared by the small batches of this context
};

// Coefficient sequence of a SymmetricCompositionIntegrator (integrators.py:176-274), alternating h1 / h2 flows.
struct mm_comp_coefs {
  int m;
  int initial_h1;
  double c[MM_MAX_COMPOSITION_COEFFS];
};

struct mm_model {
  mm_ctx* ctx = nullptr;
  int dim = 0;
  int target = 0;
  int metric_kind = 0;
  int rmetric = 0;
  int constr = 0;
  int n_constr = 0;  // number of constraint functions C (rows of the constraint Jacobian)
  // device copies (nullptr when absent)
  double* d_target_params = nullptr;
  size_t n_target_params = 0;
  double* d_metric = nullptr;       // as given: diag[D] or dense[D*D]
  double* d_metric_inv = nullptr;   // diag: 1/diag [D]; dense: explicit inverse [D*D]
  double* d_metric_chol = nullptr;  // diag: sqrt(diag) [D]; dense: lower Cholesky factor [D*D]
  int gaussian_split = 0;             // GaussianEuclideanMetricSystem (systems.py:369-474)
  int dens_wrt_ambient = 0;           // constrained systems with dens_wrt_hausdorff=False (systems.py:846-862)
  double* d_metric_omega = nullptr;   // gaussian_split: 1/sqrt(eigval) [D] (nullptr for the identity metric)
  double* d_metric_eigvec = nullptr;  // gaussian_split + dense: V [D*D] then V^T [D*D]
  double* d_rmetric_params = nullptr;
  double* d_rmetric_padded = nullptr;  // rank-one base matrix zero-padded for the team kernels (dim > 32)
  int rmetric_pad_dim = 0;             // its leading dimension (mm_team_padded_dim)
  double* d_rmetric_tiled = nullptr;   // the same matrix as the block-16 kernel's lanes read it (75 < dim <= 256):
                                       // [lower tile I (I + 1) / 2 + J][lane 16 g + j][r] = B[16 I + 4 r + g][16 J + j]
  size_t n_rmetric_params = 0;
  double* d_constr_params = nullptr;
  size_t n_constr_params = 0;
  // user-defined target (MM_TARGET_USER): module compiled at model creation by hipRTC (mm_rtc.hip)
  void* rtc_module = nullptr;
  void* rtc_integrate = nullptr;
  void* rtc_hamiltonian = nullptr;
  void* rtc_con_module = nullptr;  // constrained system with user code: constrained_core.h compiled around it
  void* rtc_con_step = nullptr;
  void* rtc_con_project = nullptr;
  void* rtc_con_logdet = nullptr;
  int rtc_con_waves = 0;  // 0: the lane-per-chain core (256 chains a workgroup); else the wave-per-chain kernels, chains a workgroup
  // dense-Riemannian system with a user metric (user_metric.h): the backends compiled around the user's source, one
  // module per kernel family - MM_RTC_FAM_WAVE (implicit_wave.h, dim <= 64), _MFMA (implicit_mfma.h, 32 < dim <= 64,
  // leapfrog step), _TEAM (implicit_team.h, 64 < dim <= 279), _BLK16 (implicit_blk16.h, 75 < dim <= 256, leapfrog step).
  // The family with the auxiliary kernels is compiled when the model is created (compile errors surface there), the
  // matrix-core step kernels on their first launch.  fn: step, midpoint, h, dh_dmom, sample_momentum.
  void* rtc_riem_module[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  void* rtc_riem_fn[5][5] = {};
  void* rtc_softabs_module = nullptr;  // SoftAbs system with a user Hessian: softabs.h compiled around it
  void* rtc_softabs_fn[3] = {nullptr, nullptr, nullptr};  // leapfrog step, midpoint step, aux (h / dh_dmom / sample_momentum)
  std::mutex rtc_mu;           // serialises the attachment of further kernel families (mm_rtc.hip riem_compile_family)
  std::string user_src;        // the user's text (kept for the families compiled later)
  int user_aux = 0;            // MM_USER_AUX of the user's text (0: none)
  bool user_flat_vjp = false;  // MM_USER_VJP_FLAT
  bool user_lowrank = false;   // MM_USER_LOWRANK (the metric is C + s u(q) u(q)^T: Woodbury path)
  double h_target_params[4] = {0, 0, 0, 0};  // first few params host-side (scalars)
  double h_rmetric_params[4] = {0, 0, 0, 0};
  double h_constr_params[4] = {0, 0, 0, 0};
};

struct mm_state {
  mm_ctx* ctx = nullptr;
  int64_t n = 0;
  int dim = 0;
  double* d_pos = nullptr;
  double* d_mom = nullptr;
  int8_t* d_dir = nullptr;
  int32_t* d_status = nullptr;
  int32_t* d_n_done = nullptr;
  // pos | mom | dir | status | n_done live in ONE device allocation (256-byte aligned sections) mirrored by a
  // pinned host buffer: a small batch goes up in one copy and comes back in one copy (the single-state
  // Integrator.step of the reference's calling pattern is transfer-latency bound)
  char* d_block = nullptr;
  char* h_stage = nullptr;  // the context's pinned staging buffer, nullptr for batches above kStageLimit
  size_t block_bytes = 0, off_mom = 0, off_dir = 0, off_status = 0, off_n_done = 0;
  bool mapped = false;  // d_block is pinned host memory the kernels access in place (mm_state_alloc_mapped)
  double* d_scratch = nullptr;  // [N] or [N*D] doubles for h / dh_dmom / z
  size_t scratch_elems = 0;
  void* d_work = nullptr;  // per-chain workspace of the large-D implicit path
  size_t work_bytes = 0;
  double* d_eig = nullptr;  // SoftAbs, D <= 64: the eigenvectors each chain's last launch ended with (k_softabs.hip)
  size_t eig_bytes = 0;
  double* d_mom_save = nullptr;  // previous momentum during a correlated refresh (mm_momentum_refresh)
  size_t mom_save_elems = 0;
  double* d_step_scale = nullptr;  // optional per-chain step-size factors (mm_state_set_step_scale)
  int32_t* d_chain_steps = nullptr;  // optional per-chain step counts (mm_state_set_chain_steps): the ACTIVE pointer the
                                     // kernels see - nullptr or d_chain_steps_buf
  int32_t* d_chain_steps_buf = nullptr;  // its allocation, kept when the counts are switched off (no free + stream
                                         // synchronisation per transition of a random-length sampler)
  uint32_t* d_errors = nullptr;  // sticky per-chain error word of device-resident transitions: bit k set when a
                                 // proposal of this chain ended with status k (mm_metropolis_accept*)
  bool rng_on = false;  // device-side random draws (mm_state_set_rng): Philox keyed by rng_seed, chain = offset + i
  uint64_t rng_seed = 0, rng_chain_offset = 0;
  double* d_tr = nullptr;  // transition scratch: u[N], accept_prob[N], accepted[N] (mm_metropolis_accept)
  size_t tr_elems = 0;
};

// ---- error plumbing -------------------------------------------------------------------------------
void mm_set_error(const mm_ctx* ctx, const std::string& msg);

#define MM_HIP_CHECK(ctx, expr)                                                              \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess) {                                                                  \
      mm_set_error((ctx), std::string(#expr) + ": " + hipGetErrorString(_e));                \
      return MM_ERR_HIP;                                                                     \
    }                                                                                        \
  } while (0)

#define MM_REQUIRE(ctx, cond, msg)       \
  do {                                   \
    if (!(cond)) {                       \
      mm_set_error((ctx), (msg));        \
      return MM_ERR_INVALID;             \
    }                                    \
  } while (0)

// ---- kernel launchers (defined in the .hip files) -----------------------------------------------------
int mm_launch_leapfrog_euclid(mm_ctx* ctx, const mm_model* m, mm_state* s, double step_size,
                              int n_steps);
int mm_launch_implicit_leapfrog(mm_ctx* ctx, const mm_model* m, mm_state* s, double step_size,
                                int n_steps, const mm_fp_opts& opts, mm_counters* d_counters);
int mm_launch_constrained_leapfrog(mm_ctx* ctx, const mm_model* m, mm_state* s, double step_size,
                                   int n_steps, const mm_proj_opts& opts, mm_counters* d_counters);
int mm_launch_hamiltonian(mm_ctx* ctx, const mm_model* m, mm_state* s, double* d_h);
int mm_launch_dh_dmom(mm_ctx* ctx, const mm_model* m, mm_state* s, double* d_out);
int mm_launch_sample_momentum(mm_ctx* ctx, const mm_model* m, mm_state* s, const double* d_z);
int mm_rtc_attach(mm_ctx* ctx, mm_model* m, const char* user_src);
void mm_rtc_detach(mm_model* m);
int mm_rtc_attach_constrained(mm_ctx* ctx, mm_model* m, const char* user_src);
int mm_rtc_attach_riemann(mm_ctx* ctx, mm_model* m, const char* user_src);
enum { MM_RTC_FAM_WAVE = 0, MM_RTC_FAM_MFMA = 1, MM_RTC_FAM_TEAM = 2, MM_RTC_FAM_BLK16 = 3, MM_RTC_FAM_GLOBAL = 4,
       MM_RTC_FAM_COUNT = 5 };  // _GLOBAL (implicit_global.h, 279 < dim <= 1024: every kernel of the model)
int mm_rtc_launch_riemann(mm_ctx* ctx, const mm_model* m, mm_state* s, int which, void* implicit_args);
int mm_rtc_attach_softabs(mm_ctx* ctx, mm_model* m, const char* user_src);
// which: 0 = leapfrog step, 1 = midpoint step (args: mmsoftabs::SaArgs), 2 = aux (SaArgs, double* out, const double* z)
int mm_rtc_launch_softabs(mm_ctx* ctx, const mm_model* m, int which, void* sa_args, int64_t n_chains, double* d_out,
                          const double* d_z);
int mm_state_ensure_work(mm_ctx* ctx, mm_state* s, size_t bytes);  // grows s->d_work (per-chain global workspace)
int mm_rtc_launch_constrained(mm_ctx* ctx, const mm_model* m, int which, void* con_args, int64_t n_chains, double* d_out);
int mm_rtc_launch_integrate(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int n_steps, const mm_comp_coefs* cf);
int mm_rtc_launch_hamiltonian(mm_ctx* ctx, const mm_model* m, mm_state* s, double* d_h);
int mm_team_padded_dim(int dim);  // k_implicit_large.hip: leading dimension of the padded rank-one base matrix
