// C ABI of libmici_amd.so (see include/mici_amd.h).  Host-side plumbing only: contexts, device
// buffers, model parameter staging (including the one-off host factorisation of a fixed dense
// metric), argument checks, kernel dispatch and the RCCL trace gather.
#include <dlfcn.h>

#include <cmath>
#include <cstring>
#include <limits>
#include <mutex>
#include <vector>

#include "mm_internal.h"

// launchers living in other translation units
int mm_launch_leapfrog_generic(mm_ctx*, const mm_model*, mm_state*, double, int);
int mm_launch_composition_generic(mm_ctx*, const mm_model*, mm_state*, double, int, int, const double*, int);
int mm_launch_composition_euclid(mm_ctx*, const mm_model*, mm_state*, double, int, const mm_comp_coefs&);
int mm_launch_implicit_midpoint_euclid(mm_ctx*, const mm_model*, mm_state*, double, int, const mm_fp_opts&,
                                       mm_counters*);
int mm_launch_implicit_midpoint_riemann(mm_ctx*, const mm_model*, mm_state*, double, int, const mm_fp_opts&,
                                        mm_counters*);
int mm_launch_metropolis_select(mm_ctx*, mm_state*, mm_state*, const double*, const double*, const double*, double*,
                                int8_t*);
int mm_launch_axpby(mm_ctx*, double* y, const double* x, double a, double b, size_t n);
int mm_launch_fill_done(mm_ctx*, mm_state*, int32_t n_steps);
int mm_launch_rng_normal(mm_ctx*, double* d_z, int64_t n, int dim, uint64_t seed, uint64_t chain_offset, uint64_t transition);
int mm_launch_rng_uniform(mm_ctx*, double* d_u, int64_t n, uint64_t seed, uint64_t chain_offset, uint64_t transition);
int mm_launch_rng_steps(mm_ctx*, int32_t* d_steps, int64_t n, uint64_t seed, uint64_t chain_offset, uint64_t transition,
                        int32_t lo, int32_t hi);
int mm_launch_euclid_hamiltonian(mm_ctx*, const mm_model*, mm_state*, double*);
int mm_launch_euclid_dh_dmom(mm_ctx*, const mm_model*, mm_state*, double*);
int mm_launch_euclid_sample_momentum(mm_ctx*, const mm_model*, mm_state*, const double*);
int mm_launch_riemann_aux(mm_ctx*, const mm_model*, mm_state*, int op, double* d_out,
                          const double* d_z);
int mm_launch_constrained_project_momentum(mm_ctx*, const mm_model*, mm_state*);
int mm_launch_constrained_add_log_det_sqrt_gram(mm_ctx*, const mm_model*, mm_state*, double*);

namespace {
thread_local std::string g_last_error;
}

void mm_set_error(const mm_ctx* ctx, const std::string& msg) {
  g_last_error = msg;
  if (ctx) const_cast<mm_ctx*>(ctx)->last_error = msg;
}

// the per-chain global-memory workspace of a state (SoftAbs matrices beyond 64 dimensions, the dense copy of the inverse
// metric a user's accessor-form vector-Jacobian product reads): grown on demand, kept with the state
int mm_state_ensure_work(mm_ctx* ctx, mm_state* s, size_t bytes) {
  if (bytes > s->work_bytes) {
    MM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (s->d_work) (void)hipFree(s->d_work);
    s->d_work = nullptr;
    s->work_bytes = 0;
    //  a batch whose per-chain workspaces do not fit the device fails HERE, with the chain count that would fit,
    // not as a raw allocation error: the global-memory tiers take 12 D^2 bytes (dense Riemannian) / ~5 MB (SoftAbs, D = 256) a chain
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && bytes > free_b) {
      const size_t per_chain = s->n > 0 ? bytes / (size_t)s->n : bytes;
      mm_set_error(ctx, "per-chain device workspace does not fit: " + std::to_string(bytes >> 20) + " MiB needed for " +
                            std::to_string((long long)s->n) + " chains (" + std::to_string(per_chain >> 10) +
                            " KiB a chain), " + std::to_string(free_b >> 20) + " MiB free - at most " +
                            std::to_string((long long)(per_chain ? free_b / per_chain : 0)) +
                            " chains of this size fit one batch; split the batch");
      return MM_ERR_UNSUPPORTED;
    }
    MM_HIP_CHECK(ctx, hipMalloc(&s->d_work, bytes));
    s->work_bytes = bytes;
  }
  return MM_OK;
}

extern "C" {

int mm_abi_version(void) { return MM_ABI_VERSION; }

const char* mm_last_error(const mm_ctx* ctx) {
  return ctx ? ctx->last_error.c_str() : g_last_error.c_str();
}

int mm_device_count(int* count) {
  MM_REQUIRE(nullptr, count != nullptr, "mm_device_count: count is NULL");
  MM_HIP_CHECK(nullptr, hipGetDeviceCount(count));
  return MM_OK;
}

// batches whose pos | mom | dir | status | n_done block is at most this big travel through the context's
// pinned staging buffer (one copy each way)
static constexpr size_t kStageLimit = (size_t)1 << 20;

int mm_ctx_create(int device, mm_ctx** out) {
  MM_REQUIRE(nullptr, out != nullptr, "mm_ctx_create: out is NULL");
  *out = nullptr;
  int count = 0;
  MM_HIP_CHECK(nullptr, hipGetDeviceCount(&count));
  MM_REQUIRE(nullptr, device >= 0 && device < count, "mm_ctx_create: no such device");
  MM_HIP_CHECK(nullptr, hipSetDevice(device));
  hipDeviceProp_t prop;
  MM_HIP_CHECK(nullptr, hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    mm_set_error(nullptr, std::string("mm_ctx_create: device is ") + prop.gcnArchName +
                              ", this library is built for gfx950 (MI355X) only");
    return MM_ERR_UNSUPPORTED;
  }
  mm_ctx* ctx = new mm_ctx();
  ctx->device = device;
  ctx->n_cu = prop.multiProcessorCount;
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
    delete ctx;
    mm_set_error(nullptr, "mm_ctx_create: hipStreamCreate failed");
    return MM_ERR_HIP;
  }
  for (auto& e : ctx->events) {
    if (hipEventCreate(&e) != hipSuccess) {
      mm_set_error(nullptr, "mm_ctx_create: hipEventCreate failed");
      return MM_ERR_HIP;
    }
  }
  if (hipMalloc(&ctx->d_counters, sizeof(mm_counters)) != hipSuccess ||
      hipHostMalloc(reinterpret_cast<void**>(&ctx->h_stage), kStageLimit, hipHostMallocDefault) != hipSuccess) {
    mm_set_error(nullptr, "mm_ctx_create: hipMalloc failed");
    return MM_ERR_NOMEM;
  }
  *out = ctx;
  return MM_OK;
}

int mm_ctx_destroy(mm_ctx* ctx) {
  if (!ctx) return MM_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  for (auto& e : ctx->events) (void)hipEventDestroy(e);
  (void)hipFree(ctx->d_counters);
  if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
  (void)hipStreamDestroy(ctx->stream);
  delete ctx;
  return MM_OK;
}

int mm_ctx_sync(mm_ctx* ctx) {
  MM_REQUIRE(nullptr, ctx != nullptr, "mm_ctx_sync: ctx is NULL");
  MM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return MM_OK;
}

int mm_ctx_record(mm_ctx* ctx, int slot) {
  MM_REQUIRE(nullptr, ctx != nullptr, "mm_ctx_record: ctx is NULL");
  MM_REQUIRE(ctx, slot >= 0 && slot < 16, "mm_ctx_record: slot out of range");
  MM_HIP_CHECK(ctx, hipEventRecord(ctx->events[slot], ctx->stream));
  return MM_OK;
}

int mm_ctx_elapsed_ms(mm_ctx* ctx, int a, int b, double* ms) {
  MM_REQUIRE(nullptr, ctx != nullptr, "mm_ctx_elapsed_ms: ctx is NULL");
  MM_REQUIRE(ctx, a >= 0 && a < 16 && b >= 0 && b < 16 && ms, "mm_ctx_elapsed_ms: bad argument");
  MM_HIP_CHECK(ctx, hipEventSynchronize(ctx->events[b]));
  float f = 0.f;
  MM_HIP_CHECK(ctx, hipEventElapsedTime(&f, ctx->events[a], ctx->events[b]));
  *ms = f;
  return MM_OK;
}

// ---- model --------------------------------------------------------------------------------------------
static int upload(mm_ctx* ctx, const double* src, size_t n, double** dst) {
  *dst = nullptr;
  if (n == 0) return MM_OK;
  if (hipMalloc(dst, n * sizeof(double)) != hipSuccess) {
    mm_set_error(ctx, "hipMalloc failed while staging model parameters");
    return MM_ERR_NOMEM;
  }
  MM_HIP_CHECK(ctx, hipMemcpy(*dst, src, n * sizeof(double), hipMemcpyHostToDevice));
  return MM_OK;
}

// One-off host factorisation of a FIXED dense metric (matrices.py:1161-1188): lower Cholesky factor
// and the explicit inverse L^-T L^-1 the reference multiplies momenta by (matrices.py:222-223).
static bool host_chol_inverse(const double* a, int n, std::vector<double>& l, std::vector<double>& inv) {
  l.assign((size_t)n * n, 0.0);
  for (int j = 0; j < n; ++j) {
    double d = a[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) d -= l[(size_t)j * n + k] * l[(size_t)j * n + k];
    if (!(d > 0.0) || !std::isfinite(d)) return false;
    const double ljj = std::sqrt(d);
    l[(size_t)j * n + j] = ljj;
    for (int i = j + 1; i < n; ++i) {
      double s = a[(size_t)i * n + j];
      for (int k = 0; k < j; ++k) s -= l[(size_t)i * n + k] * l[(size_t)j * n + k];
      l[(size_t)i * n + j] = s / ljj;
    }
  }
  // W = L^-1 (lower), column by column
  std::vector<double> w((size_t)n * n, 0.0);
  for (int c = 0; c < n; ++c) {
    for (int i = c; i < n; ++i) {
      double s = (i == c) ? 1.0 : 0.0;
      for (int k = c; k < i; ++k) s -= l[(size_t)i * n + k] * w[(size_t)k * n + c];
      w[(size_t)i * n + c] = s / l[(size_t)i * n + i];
    }
  }
  inv.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = 0.0;
      for (int k = i; k < n; ++k) s += w[(size_t)k * n + i] * w[(size_t)k * n + j];
      inv[(size_t)i * n + j] = inv[(size_t)j * n + i] = s;
    }
  return true;
}

// Symmetric eigendecomposition by cyclic Jacobi rotations (host side, once per model): a = V diag(w) V^T
// with V row-major, eigenvector k in column k.  Replaces DensePositiveDefiniteMatrix.eigval / eigvec
// (matrices.py:1203-1219, numpy.linalg.eigh); only V f(w) V^T is ever formed from it, which does not
// depend on the ordering or sign conventions of the decomposition.
static bool host_jacobi_eigh(const double* a_in, int n, std::vector<double>& w, std::vector<double>& v) {
  std::vector<double> a(a_in, a_in + (size_t)n * n);
  v.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) v[(size_t)i * n + i] = 1.0;
  double total = 0.0;
  for (double x : a) total += x * x;
  if (!std::isfinite(total)) return false;
  for (int sweep = 0; sweep < 64; ++sweep) {
    double off = 0.0;
    for (int i = 0; i < n; ++i)
      for (int j = i + 1; j < n; ++j) off += a[(size_t)i * n + j] * a[(size_t)i * n + j];
    if (off <= 1e-34 * total) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = a[(size_t)p * n + q];
        if (apq == 0.0) continue;
        const double theta = (a[(size_t)q * n + q] - a[(size_t)p * n + p]) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; ++k) {  // columns p, q of a
          const double akp = a[(size_t)k * n + p], akq = a[(size_t)k * n + q];
          a[(size_t)k * n + p] = c * akp - s * akq;
          a[(size_t)k * n + q] = s * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {  // rows p, q of a
          const double apk = a[(size_t)p * n + k], aqk = a[(size_t)q * n + k];
          a[(size_t)p * n + k] = c * apk - s * aqk;
          a[(size_t)q * n + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          const double vkp = v[(size_t)k * n + p], vkq = v[(size_t)k * n + q];
          v[(size_t)k * n + p] = c * vkp - s * vkq;
          v[(size_t)k * n + q] = s * vkp + c * vkq;
        }
      }
  }
  w.resize(n);
  for (int i = 0; i < n; ++i) w[i] = a[(size_t)i * n + i];
  return true;
}

static int model_create(mm_ctx* ctx, const mm_model_desc* d, const char* user_src, mm_model** out);

int mm_model_create(mm_ctx* ctx, const mm_model_desc* d, mm_model** out) {
  MM_REQUIRE(nullptr, ctx != nullptr, "mm_model_create: ctx is NULL");
  MM_REQUIRE(ctx, d != nullptr && out != nullptr, "mm_model_create: NULL argument");
  MM_REQUIRE(ctx, d->target != MM_TARGET_USER && d->constr != MM_CONSTR_USER && d->rmetric != MM_RMETRIC_USER &&
                      d->rmetric != MM_RMETRIC_SOFTABS_USER,
             "mm_model_create: user code (MM_TARGET_USER / MM_CONSTR_USER / MM_RMETRIC_USER) needs "
             "mm_model_create_from_source");
  return model_create(ctx, d, nullptr, out);
}

int mm_model_create_from_source(mm_ctx* ctx, const mm_model_desc* d, const char* hip_source, mm_model** out) {
  MM_REQUIRE(nullptr, ctx != nullptr, "mm_model_create_from_source: ctx is NULL");
  MM_REQUIRE(ctx, d != nullptr && out != nullptr && hip_source != nullptr, "mm_model_create_from_source: NULL argument");
  MM_REQUIRE(ctx, d->target == MM_TARGET_USER || d->constr == MM_CONSTR_USER || d->rmetric == MM_RMETRIC_USER ||
                      d->rmetric == MM_RMETRIC_SOFTABS_USER,
             "mm_model_create_from_source: one of desc->target / constr / rmetric must be the _USER id");
  MM_REQUIRE(ctx, d->rmetric != MM_RMETRIC_SOFTABS_USER || d->dim <= 256,
             "mm_model_create_from_source: a SoftAbs system with a user Hessian runs one workgroup per chain, dim <= 256");
  MM_REQUIRE(ctx, d->rmetric == MM_RMETRIC_NONE || d->rmetric == MM_RMETRIC_USER || d->rmetric == MM_RMETRIC_SOFTABS_USER,
             "mm_model_create_from_source: a user target on a Riemannian system needs a user metric too (the built-in "
             "metrics' kernels are compiled ahead of time around the built-in targets)");
  MM_REQUIRE(ctx, d->rmetric != MM_RMETRIC_USER || d->dim <= 1024,
             "mm_model_create_from_source: user metrics run on the dense-Riemannian kernels, dim <= 1024 (register-resident to "
             "279, the global-memory tier beyond)");
  MM_REQUIRE(ctx, d->target != MM_TARGET_USER || !d->gaussian_split,
             "mm_model_create_from_source: a user target is a density with respect to the Lebesgue measure (identity / "
             "diagonal / dense fixed metric); the Gaussian-split system classes take built-in targets");
  return model_create(ctx, d, hip_source, out);
}

static int model_create(mm_ctx* ctx, const mm_model_desc* d, const char* user_src, mm_model** out) {
  *out = nullptr;
  const int D = d->dim;
  MM_REQUIRE(ctx, D >= 1, "mm_model_create: dim must be >= 1");
  size_t need_t = 0;
  switch (d->target) {
    case MM_TARGET_GAUSS_ISO: case MM_TARGET_BANANA: need_t = 0; break;
    case MM_TARGET_GAUSS_DIAG: need_t = D; break;
    case MM_TARGET_GAUSS_DENSE: need_t = (size_t)D * D; break;
    case MM_TARGET_POLY: need_t = 2; break;
    case MM_TARGET_FUNNEL: need_t = D - 1; MM_REQUIRE(ctx, D >= 2, "funnel target needs dim >= 2"); break;
    case MM_TARGET_TORUS: need_t = 3; MM_REQUIRE(ctx, D == 3, "torus target needs dim == 3"); break;
    case MM_TARGET_USER: need_t = d->n_target_params; break;
    default: MM_REQUIRE(ctx, false, "mm_model_create: unknown target id");
  }
  MM_REQUIRE(ctx, d->n_target_params == need_t && (need_t == 0 || d->target_params),
             "mm_model_create: wrong number of target params");
  size_t need_m = d->metric_kind == MM_METRIC_IDENTITY ? 0
                  : d->metric_kind == MM_METRIC_DIAG   ? (size_t)D
                  : d->metric_kind == MM_METRIC_DENSE  ? (size_t)D * D
                                                       : (size_t)-1;
  MM_REQUIRE(ctx, need_m != (size_t)-1, "mm_model_create: unknown metric kind");
  MM_REQUIRE(ctx, d->n_metric == need_m && (need_m == 0 || d->metric),
             "mm_model_create: wrong number of metric entries");
  size_t need_r = d->rmetric == MM_RMETRIC_NONE       ? 0
                  : d->rmetric == MM_RMETRIC_RANK1    ? (size_t)D * D
                  : d->rmetric == MM_RMETRIC_DIAGQUAD ? 0
                  : d->rmetric == MM_RMETRIC_SOFTABS  ? 1
                  : d->rmetric == MM_RMETRIC_USER     ? d->n_rmetric_params
                  : d->rmetric == MM_RMETRIC_SOFTABS_USER ? (d->n_rmetric_params >= 1 ? d->n_rmetric_params : (size_t)-1)
                                                      : (size_t)-1;
  MM_REQUIRE(ctx, need_r != (size_t)-1, "mm_model_create: unknown Riemannian metric id");
  MM_REQUIRE(ctx, d->n_rmetric_params == need_r && (need_r == 0 || d->rmetric_params),
             "mm_model_create: wrong number of Riemannian metric params");
  MM_REQUIRE(ctx, d->rmetric == MM_RMETRIC_NONE || d->metric_kind == MM_METRIC_IDENTITY,
             "mm_model_create: a Riemannian system has no fixed metric");
  size_t need_c = d->constr == MM_CONSTR_NONE           ? 0
                  : d->constr == MM_CONSTR_TORUS        ? 2
                  : d->constr == MM_CONSTR_FIRST        ? 0
                  : d->constr == MM_CONSTR_CIRCLE       ? 0
                  : d->constr == MM_CONSTR_LINEAR       ? d->n_constr_params
                  : d->constr == MM_CONSTR_SPHERE_PLANE ? (size_t)D
                  : d->constr == MM_CONSTR_SPHERE       ? 0
                  : d->constr == MM_CONSTR_USER         ? d->n_constr_params
                                                        : (size_t)-1;
  MM_REQUIRE(ctx, need_c != (size_t)-1, "mm_model_create: unknown constraint id");
  int n_constr = d->constr == MM_CONSTR_NONE ? 0 : 1;
  if (d->constr == MM_CONSTR_LINEAR) {
    MM_REQUIRE(ctx, need_c > 0 && need_c % (size_t)(D + 1) == 0,
               "mm_model_create: linear constraint needs C*(D+1) params (A[C*D] then b[C])");
    n_constr = (int)(need_c / (size_t)(D + 1));
    MM_REQUIRE(ctx, n_constr >= 1 && n_constr <= 8 && (n_constr < D || D == 1),
               "mm_model_create: linear constraint supports 1 <= C <= 8 rows, C < dim");
  }
  if (d->constr == MM_CONSTR_USER) {
    n_constr = d->n_constr;
    MM_REQUIRE(ctx, n_constr >= 1 && n_constr <= 8 && n_constr < D && D <= 256,
               "mm_model_create_from_source: a user constraint needs 1 <= n_constr <= 8, n_constr < dim <= 256");
  }
  MM_REQUIRE(ctx, d->constr != MM_CONSTR_SPHERE || D >= 2, "sphere constraint needs dim >= 2");
  if (d->constr == MM_CONSTR_SPHERE_PLANE) {
    MM_REQUIRE(ctx, D >= 3, "sphere-plane constraint needs dim >= 3");
    n_constr = 2;
  }
  MM_REQUIRE(ctx, d->n_constr_params == need_c && (need_c == 0 || d->constr_params),
             "mm_model_create: wrong number of constraint params");
  MM_REQUIRE(ctx, d->constr != MM_CONSTR_TORUS || D == 3, "torus constraint needs dim == 3");
  MM_REQUIRE(ctx, d->constr != MM_CONSTR_CIRCLE || D >= 2, "circle constraint needs dim >= 2");
  MM_REQUIRE(ctx, d->constr == MM_CONSTR_NONE || d->rmetric == MM_RMETRIC_NONE,
             "mm_model_create: constrained Riemannian systems are not part of the path");
  MM_REQUIRE(ctx, d->rmetric == MM_RMETRIC_NONE || d->target != MM_TARGET_TORUS,
             "mm_model_create: the torus target is only defined for constrained systems");
  MM_REQUIRE(ctx, d->target != MM_TARGET_FUNNEL || d->rmetric == MM_RMETRIC_NONE ||
                      d->rmetric == MM_RMETRIC_SOFTABS || d->rmetric == MM_RMETRIC_SOFTABS_USER,
             "mm_model_create: the funnel target pairs with a fixed metric or the SoftAbs metric");
  if (d->target == MM_TARGET_FUNNEL && d->constr != MM_CONSTR_NONE) {  // (built-in and user constraints alike)
    mm_set_error(ctx, "mm_model_create: the funnel target is not available on constrained systems (its gradient is a "
                      "wave-collective the lane-per-chain constrained core does not form)");
    return MM_ERR_UNSUPPORTED;
  }
  MM_REQUIRE(ctx, d->gaussian_split == 0 || d->gaussian_split == 1, "mm_model_create: gaussian_split must be 0 or 1");
  MM_REQUIRE(ctx, !d->gaussian_split || d->rmetric == MM_RMETRIC_NONE,
             "mm_model_create: the Gaussian split is defined for fixed-metric systems only");
  MM_REQUIRE(ctx, d->dens_wrt_ambient == 0 || d->dens_wrt_ambient == 1,
             "mm_model_create: dens_wrt_ambient must be 0 or 1");
  MM_REQUIRE(ctx, !d->dens_wrt_ambient || d->constr != MM_CONSTR_NONE,
             "mm_model_create: dens_wrt_ambient needs a constraint");
  if (d->rmetric == MM_RMETRIC_SOFTABS_USER)
    MM_REQUIRE(ctx, d->rmetric_params[0] > 0.0, "softabs_coeff must be positive");  // matrices.py:1652-1654
  if (d->rmetric == MM_RMETRIC_SOFTABS) {
    MM_REQUIRE(ctx, d->rmetric_params[0] > 0.0, "softabs_coeff must be positive");  // matrices.py:1652-1654
    MM_REQUIRE(ctx, d->target == MM_TARGET_FUNNEL || d->target == MM_TARGET_POLY,
               "SoftAbs metric needs a target with device Hessian/MTP (funnel, poly)");
  }

  MM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  mm_model* m = new mm_model();
  m->ctx = ctx;
  m->dim = D;
  m->target = d->target;
  m->metric_kind = d->metric_kind;
  m->gaussian_split = d->gaussian_split;
  // GaussianDenseConstrainedEuclideanMetricSystem always passes dens_wrt_hausdorff=False (systems.py:1114-1124)
  m->dens_wrt_ambient = (d->dens_wrt_ambient || (d->gaussian_split && d->constr != MM_CONSTR_NONE)) ? 1 : 0;
  m->rmetric = d->rmetric;
  m->constr = d->constr;
  m->n_constr = n_constr;
  m->n_target_params = need_t;
  m->n_rmetric_params = need_r;
  m->n_constr_params = need_c;
  for (size_t i = 0; i < need_t && i < 4; ++i) m->h_target_params[i] = d->target_params[i];
  for (size_t i = 0; i < need_r && i < 4; ++i) m->h_rmetric_params[i] = d->rmetric_params[i];
  for (size_t i = 0; i < need_c && i < 4; ++i) m->h_constr_params[i] = d->constr_params[i];
  int rc = upload(ctx, d->target_params, need_t, &m->d_target_params);
  if (rc == MM_OK) rc = upload(ctx, d->rmetric_params, need_r, &m->d_rmetric_params);
  if (rc == MM_OK) rc = upload(ctx, d->constr_params, need_c, &m->d_constr_params);
  if (rc == MM_OK && d->rmetric == MM_RMETRIC_RANK1 && D > 32 && D <= 279) {
    // team-per-chain kernels read the base matrix through a zero-padded image of their tile geometry
    const int DP = mm_team_padded_dim(D);
    m->rmetric_pad_dim = DP;
    std::vector<double> pad((size_t)DP * DP, 0.0);
    for (int i = 0; i < D; ++i)
      for (int j = 0; j < D; ++j) pad[(size_t)i * DP + j] = d->rmetric_params[(size_t)i * D + j];
    rc = upload(ctx, pad.data(), pad.size(), &m->d_rmetric_padded);
    if (rc == MM_OK && D > 75 && D <= 256) {
      // k_implicit_blk16.hip: every lane's four entries of a 16 x 16 tile are 32 consecutive bytes, a tile 2 KB
      std::vector<double> tl((size_t)136 * 256, 0.0);
      for (int I = 0; I < 16; ++I)
        for (int J = 0; J <= I; ++J)
          for (int lane = 0; lane < 64; ++lane)
            for (int r = 0; r < 4; ++r) {
              const int row = 16 * I + 4 * r + (lane >> 4), col = 16 * J + (lane & 15);
              if (row < D && col < D)
                tl[((size_t)(I * (I + 1) / 2 + J) * 64 + lane) * 4 + r] = d->rmetric_params[(size_t)row * D + col];
            }
      rc = upload(ctx, tl.data(), tl.size(), &m->d_rmetric_tiled);
    }
  }
  if (rc == MM_OK) rc = upload(ctx, d->metric, need_m, &m->d_metric);
  if (rc == MM_OK && d->metric_kind == MM_METRIC_DIAG) {
    std::vector<double> inv(D), sq(D);
    for (int i = 0; i < D; ++i) {
      if (!(d->metric[i] > 0.0) || !std::isfinite(d->metric[i])) {
        mm_set_error(ctx, "mm_model_create: diagonal metric must be positive and finite");
        rc = MM_ERR_INVALID;
        break;
      }
      inv[i] = 1.0 / d->metric[i];
      sq[i] = std::sqrt(d->metric[i]);
    }
    if (rc == MM_OK) rc = upload(ctx, inv.data(), D, &m->d_metric_inv);
    if (rc == MM_OK) rc = upload(ctx, sq.data(), D, &m->d_metric_chol);
    if (rc == MM_OK && d->gaussian_split) {  // omega = 1 / eigval**0.5 (systems.py:465), eigval = diagonal
      std::vector<double> om(D);
      for (int i = 0; i < D; ++i) om[i] = 1.0 / sq[i];
      rc = upload(ctx, om.data(), D, &m->d_metric_omega);
    }
  }
  if (rc == MM_OK && d->metric_kind == MM_METRIC_DENSE) {
    std::vector<double> l, inv;
    if (!host_chol_inverse(d->metric, D, l, inv)) {
      mm_set_error(ctx, "mm_model_create: Cholesky factorisation failed.");  // matrices.py:1170-1172
      rc = MM_ERR_INVALID;
    }
    if (rc == MM_OK) rc = upload(ctx, inv.data(), (size_t)D * D, &m->d_metric_inv);
    if (rc == MM_OK) rc = upload(ctx, l.data(), (size_t)D * D, &m->d_metric_chol);
    if (rc == MM_OK && d->gaussian_split) {
      std::vector<double> w, v;
      if (!host_jacobi_eigh(d->metric, D, w, v)) {
        mm_set_error(ctx, "mm_model_create: eigendecomposition of the metric failed");
        rc = MM_ERR_INVALID;
      }
      if (rc == MM_OK) {
        std::vector<double> om(D), vv((size_t)2 * D * D);
        for (int i = 0; i < D; ++i) om[i] = 1.0 / std::sqrt(w[i]);
        for (int i = 0; i < D; ++i)
          for (int j = 0; j < D; ++j) {
            vv[(size_t)i * D + j] = v[(size_t)i * D + j];
            vv[(size_t)D * D + (size_t)j * D + i] = v[(size_t)i * D + j];
          }
        rc = upload(ctx, om.data(), D, &m->d_metric_omega);
        if (rc == MM_OK) rc = upload(ctx, vv.data(), vv.size(), &m->d_metric_eigvec);
      }
    }
  }
  // user code: the Euclidean wave-per-chain kernels around a user target (h, unconstrained integrators), and for a
  // constrained system the constrained-leapfrog core around the user constraint and / or target
  if (rc == MM_OK && user_src && d->rmetric == MM_RMETRIC_USER) rc = mm_rtc_attach_riemann(ctx, m, user_src);
  if (rc == MM_OK && user_src && d->rmetric == MM_RMETRIC_SOFTABS_USER) rc = mm_rtc_attach_softabs(ctx, m, user_src);
  if (rc == MM_OK && user_src && d->target == MM_TARGET_USER && d->rmetric == MM_RMETRIC_NONE)
    rc = mm_rtc_attach(ctx, m, user_src);
  if (rc == MM_OK && user_src && d->constr != MM_CONSTR_NONE) rc = mm_rtc_attach_constrained(ctx, m, user_src);
  if (rc != MM_OK) {
    mm_model_destroy(m);
    return rc;
  }
  *out = m;
  return MM_OK;
}

int mm_model_destroy(mm_model* m) {
  if (!m) return MM_OK;
  (void)hipSetDevice(m->ctx->device);
  (void)hipStreamSynchronize(m->ctx->stream);  // nothing of this model's run-time compiled modules may still be in flight
  mm_rtc_detach(m);
  (void)hipFree(m->d_target_params);
  (void)hipFree(m->d_metric);
  (void)hipFree(m->d_metric_inv);
  (void)hipFree(m->d_metric_chol);
  (void)hipFree(m->d_metric_omega);
  (void)hipFree(m->d_metric_eigvec);
  (void)hipFree(m->d_rmetric_params);
  (void)hipFree(m->d_rmetric_padded);
  (void)hipFree(m->d_rmetric_tiled);
  (void)hipFree(m->d_constr_params);
  delete m;
  return MM_OK;
}

// ---- state ---------------------------------------------------------------------------------------------
static int state_alloc(mm_ctx* ctx, int64_t n, int32_t dim, bool mapped, mm_state** out);

int mm_state_alloc(mm_ctx* ctx, int64_t n, int32_t dim, mm_state** out) {
  return state_alloc(ctx, n, dim, false, out);
}

int mm_state_alloc_mapped(mm_ctx* ctx, int64_t n, int32_t dim, mm_state** out) {
  MM_REQUIRE(ctx, ctx == nullptr || (n >= 0 && (size_t)n * (size_t)(dim > 0 ? dim : 1) <= 65536),
             "mm_state_alloc_mapped: meant for small, long-lived batches (n * dim <= 65536)");
  return state_alloc(ctx, n, dim, true, out);
}

static int state_alloc(mm_ctx* ctx, int64_t n, int32_t dim, bool mapped, mm_state** out) {
  MM_REQUIRE(nullptr, ctx != nullptr, "mm_state_alloc: ctx is NULL");
  MM_REQUIRE(ctx, out != nullptr, "mm_state_alloc: out is NULL");
  *out = nullptr;
  MM_REQUIRE(ctx, n >= 0 && dim >= 1, "mm_state_alloc: need n_chains >= 0 and dim >= 1");
  MM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  mm_state* s = new mm_state();
  s->ctx = ctx;
  s->n = n;
  s->dim = dim;
  const size_t nd = (size_t)(n > 0 ? n : 1) * dim, n1 = (size_t)(n > 0 ? n : 1);
  s->scratch_elems = nd;
  auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
  s->off_mom = up(nd * sizeof(double));
  s->off_dir = s->off_mom + up(nd * sizeof(double));
  s->off_status = s->off_dir + up(n1);
  s->off_n_done = s->off_status + up(n1 * sizeof(int32_t));
  s->block_bytes = s->off_n_done + up(n1 * sizeof(int32_t));
  s->mapped = mapped;
  bool ok = (mapped ? hipHostMalloc(reinterpret_cast<void**>(&s->d_block), s->block_bytes, hipHostMallocDefault)
                    : hipMalloc(&s->d_block, s->block_bytes)) == hipSuccess &&
            hipMalloc(&s->d_scratch, nd * sizeof(double)) == hipSuccess;
  if (ok) {
    s->d_pos = reinterpret_cast<double*>(s->d_block);
    s->d_mom = reinterpret_cast<double*>(s->d_block + s->off_mom);
    s->d_dir = reinterpret_cast<int8_t*>(s->d_block + s->off_dir);
    s->d_status = reinterpret_cast<int32_t*>(s->d_block + s->off_status);
    s->d_n_done = reinterpret_cast<int32_t*>(s->d_block + s->off_n_done);
    if (!mapped && s->block_bytes <= kStageLimit) s->h_stage = ctx->h_stage;  // every use drains the stream first
  }
  if (!ok) {
    mm_state_free(s);
    mm_set_error(ctx, "mm_state_alloc: hipMalloc failed");
    return MM_ERR_NOMEM;
  }
  if (mapped) {
    std::memset(s->d_status, 0, n1 * sizeof(int32_t));
    std::memset(s->d_n_done, 0, n1 * sizeof(int32_t));
    std::memset(s->d_dir, 1, n1);
  } else {
    (void)hipMemsetAsync(s->d_status, 0, n1 * sizeof(int32_t), ctx->stream);
    (void)hipMemsetAsync(s->d_n_done, 0, n1 * sizeof(int32_t), ctx->stream);
    (void)hipMemsetAsync(s->d_dir, 1, n1, ctx->stream);
  }
  *out = s;
  return MM_OK;
}

int mm_state_free(mm_state* s) {
  if (!s) return MM_OK;
  (void)hipSetDevice(s->ctx->device);
  (void)hipStreamSynchronize(s->ctx->stream);
  if (s->mapped) (void)hipHostFree(s->d_block);
  else (void)hipFree(s->d_block);
  (void)hipFree(s->d_scratch);
  (void)hipFree(s->d_work);
  (void)hipFree(s->d_eig);
  (void)hipFree(s->d_tr);
  (void)hipFree(s->d_mom_save);
  (void)hipFree(s->d_step_scale);
  (void)hipFree(s->d_chain_steps_buf);
  (void)hipFree(s->d_errors);
  delete s;
  return MM_OK;
}

int mm_state_upload(mm_state* s, const double* pos, const double* mom, const int8_t* dir) {
  MM_REQUIRE(nullptr, s != nullptr, "mm_state_upload: state is NULL");
  mm_ctx* ctx = s->ctx;
  MM_HIP_CHECK(ctx, hipSetDevice(ctx->device));  // a process may own contexts on several GPUs
  if (s->n == 0) return MM_OK;
  const size_t nd = (size_t)s->n * s->dim;
  if (dir)
    for (int64_t i = 0; i < s->n; ++i)
      MM_REQUIRE(ctx, dir[i] == 1 || dir[i] == -1, "mm_state_upload: dir entries must be +1 or -1");
  if (s->mapped) {
    // the kernels read this memory in place: wait for whatever still uses it, then plain stores
    MM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (pos) std::memcpy(s->d_pos, pos, nd * sizeof(double));
    if (mom) std::memcpy(s->d_mom, mom, nd * sizeof(double));
    if (dir) std::memcpy(s->d_dir, dir, (size_t)s->n);
    return MM_OK;
  }
  if (s->h_stage && pos && mom && dir) {
    // one copy: the caller's buffers are consumed here (into the pinned mirror), so nothing has to be waited
    // for; the stream is drained first because an earlier transfer may still be reading the mirror
    MM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    std::memcpy(s->h_stage, pos, nd * sizeof(double));
    std::memcpy(s->h_stage + s->off_mom, mom, nd * sizeof(double));
    std::memcpy(s->h_stage + s->off_dir, dir, (size_t)s->n);
    MM_HIP_CHECK(ctx, hipMemcpyAsync(s->d_block, s->h_stage, s->off_dir + (size_t)s->n, hipMemcpyHostToDevice,
                                     ctx->stream));
    return MM_OK;
  }
  if (pos) MM_HIP_CHECK(ctx, hipMemcpyAsync(s->d_pos, pos, nd * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  if (mom) MM_HIP_CHECK(ctx, hipMemcpyAsync(s->d_mom, mom, nd * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  if (dir) MM_HIP_CHECK(ctx, hipMemcpyAsync(s->d_dir, dir, (size_t)s->n, hipMemcpyHostToDevice, ctx->stream));
  MM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // host buffers are only borrowed
  return MM_OK;
}

int mm_state_download_all(mm_state* s, double* pos, double* mom, int8_t* dir, int32_t* status, int32_t* n_done);

int mm_state_download(mm_state* s, double* pos, double* mom, int8_t* dir) {
  MM_REQUIRE(nullptr, s != nullptr, "mm_state_download: state is NULL");
  mm_ctx* ctx = s->ctx;
  MM_HIP_CHECK(ctx, hipSetDevice(ctx->device));  // a process may own contexts on several GPUs
  if (s->n == 0) return MM_OK;
  const size_t nd = (size_t)s->n * s->dim;
  if (s->mapped) return mm_state_download_all(s, pos, mom, dir, nullptr, nullptr);
  if (pos) MM_HIP_CHECK(ctx, hipMemcpyAsync(pos, s->d_pos, nd * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  if (mom) MM_HIP_CHECK(ctx, hipMemcpyAsync(mom, s->d_mom, nd * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  if (dir) MM_HIP_CHECK(ctx, hipMemcpyAsync(dir, s->d_dir, (size_t)s->n, hipMemcpyDeviceToHost, ctx->stream));
  MM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return MM_OK;
}

int mm_state_download_all(mm_state* s, double* pos, double* mom, int8_t* dir, int32_t* status, int32_t* n_done) {
  MM_REQUIRE(nullptr, s != nullptr, "mm_state_download_all: state is NULL");
  mm_ctx* ctx = s->ctx;
  MM_HIP_CHECK(ctx, hipSetDevice(ctx->device));  // a process may own contexts on several GPUs
  if (s->n == 0) return MM_OK;
  const size_t nd = (size_t)s->n * s->dim;
  if (s->mapped) {
    MM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (pos) std::memcpy(pos, s->d_pos, nd * sizeof(double));
    if (mom) std::memcpy(mom, s->d_mom, nd * sizeof(double));
    if (dir) std::memcpy(dir, s->d_dir, (size_t)s->n);
    if (status) std::memcpy(status, s->d_status, (size_t)s->n * sizeof(int32_t));
    if (n_done) std::memcpy(n_done, s->d_n_done, (size_t)s->n * sizeof(int32_t));
    return MM_OK;
  }
  if (!s->h_stage) {
    int rc = mm_state_download(s, pos, mom, dir);
    return rc != MM_OK ? rc : mm_state_download_status(s, status, n_done);
  }
  MM_HIP_CHECK(ctx, hipMemcpyAsync(s->h_stage, s->d_block, s->block_bytes, hipMemcpyDeviceToHost, ctx->stream));
  MM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  if (pos) std::memcpy(pos, s->h_stage, nd * sizeof(double));
  if (mom) std::memcpy(mom, s->h_stage + s->off_mom, nd * sizeof(double));
  if (dir) std::memcpy(dir, s->h_stage + s->off_dir, (size_t)s->n);
  if (status) std::memcpy(status, s->h_stage + s->off_status, (size_t)s->n * sizeof(int32_t));
  if (n_done) std::memcpy(n_done, s->h_stage + s->off_n_done, (size_t)s->n * sizeof(int32_t));
  return MM_OK;
}

int mm_state_download_status(mm_state* s, int32_t* status, int32_t* n_done) {
  MM_REQUIRE(nullptr, s != nullptr, "mm_state_download_status: state is NULL");
  mm_ctx* ctx = s->ctx;
  MM_HIP_CHECK(ctx, hipSetDevice(ctx->device));  // a process may own contexts on several GPUs
  if (s->n == 0) return MM_OK;
  if (s->mapped) return mm_state_download_all(s, nullptr, nullptr, nullptr, status, n_done);
  if (status) MM_HIP_CHECK(ctx, hipMemcpyAsync(status, s->d_status, (size_t)s->n * 4, hipMemcpyDeviceToHost, ctx->stream));
  if (n_done) MM_HIP_CHECK(ctx, hipMemcpyAsync(n_done, s->d_n_done, (size_t)s->n * 4, hipMemcpyDeviceToHost, ctx->stream));
  MM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return MM_OK;
}

int mm_state_copy(mm_state* dst, const mm_state* src) {
  MM_REQUIRE(nullptr, dst != nullptr && src != nullptr, "mm_state_copy: NULL state");
  mm_ctx* ctx = dst->ctx;
  MM_REQUIRE(ctx, src->ctx == ctx && src->n == dst->n && src->dim == dst->dim,
             "mm_state_copy: states differ in ctx / n_chains / dim");
  if (dst == src || dst->n == 0) return MM_OK;
  MM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const size_t nd = (size_t)src->n * src->dim * sizeof(double), n = (size_t)src->n;
  MM_HIP_CHECK(ctx, hipMemcpyAsync(dst->d_pos, src->d_pos, nd, hipMemcpyDefault, ctx->stream));
  MM_HIP_CHECK(ctx, hipMemcpyAsync(dst->d_mom, src->d_mom, nd, hipMemcpyDefault, ctx->stream));
  MM_HIP_CHECK(ctx, hipMemcpyAsync(dst->d_dir, src->d_dir, n, hipMemcpyDefault, ctx->stream));
  MM_HIP_CHECK(ctx, hipMemcpyAsync(dst->d_status, src->d_status, n * 4, hipMemcpyDefault, ctx->stream));
  MM_HIP_CHECK(ctx, hipMemcpyAsync(dst->d_n_done, src->d_n_done, n * 4, hipMemcpyDefault, ctx->stream));
  if (src->d_chain_steps) {  // ... and the same per-chain trajectory lengths
    if (!dst->d_chain_steps_buf) MM_HIP_CHECK(ctx, hipMalloc(&dst->d_chain_steps_buf, n * sizeof(int32_t)));
    dst->d_chain_steps = dst->d_chain_steps_buf;
    MM_HIP_CHECK(ctx, hipMemcpyAsync(dst->d_chain_steps, src->d_chain_steps, n * sizeof(int32_t), hipMemcpyDeviceToDevice, ctx->stream));
  } else {
    dst->d_chain_steps = nullptr;  // switched off; the buffer stays for the next transition
  }
  if (src->d_eig) {  // ... and the SoftAbs eigenvector bases the chains carry from launch to launch
    if (dst->eig_bytes != src->eig_bytes) {
      MM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
      (void)hipFree(dst->d_eig);
      dst->d_eig = nullptr;
      dst->eig_bytes = 0;
      MM_HIP_CHECK(ctx, hipMalloc(&dst->d_eig, src->eig_bytes));
      dst->eig_bytes = src->eig_bytes;
    }
    MM_HIP_CHECK(ctx, hipMemcpyAsync(dst->d_eig, src->d_eig, src->eig_bytes, hipMemcpyDeviceToDevice, ctx->stream));
  } else if (dst->d_eig) {
    // the source carries no bases: the copy must not keep the ones of whatever positions it held before (zero = "no
    // basis yet", as ensure_eig leaves a fresh buffer)
    MM_HIP_CHECK(ctx, hipMemsetAsync(dst->d_eig, 0, dst->eig_bytes, ctx->stream));
  }
  if (src->d_step_scale) {  // the copy integrates with the same per-chain step sizes
    if (!dst->d_step_scale) MM_HIP_CHECK(ctx, hipMalloc(&dst->d_step_scale, n * sizeof(double)));
    MM_HIP_CHECK(ctx, hipMemcpyAsync(dst->d_step_scale, src->d_step_scale, n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  } else if (dst->d_step_scale) {
    MM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(dst->d_step_scale);
    dst->d_step_scale = nullptr;
  }
  return MM_OK;
}

int mm_state_set_chain_steps(mm_state* s, const int32_t* steps) {
  MM_REQUIRE(nullptr, s != nullptr, "mm_state_set_chain_steps: state is NULL");
  mm_ctx* ctx = s->ctx;
  if (!steps) {  // back to one common trajectory length: nothing to wait for, the buffer is kept for the next use
    s->d_chain_steps = nullptr;
    return MM_OK;
  }
  MM_HIP_CHECK(ctx, hipSetDevice(ctx->device));  // a process may own contexts on several GPUs
  if (s->n == 0) return MM_OK;
  for (int64_t i = 0; i < s->n; ++i)
    MM_REQUIRE(ctx, steps[i] >= 0, "mm_state_set_chain_steps: counts must be non-negative");
  if (!s->d_chain_steps_buf) MM_HIP_CHECK(ctx, hipMalloc(&s->d_chain_steps_buf, (size_t)s->n * sizeof(int32_t)));
  s->d_chain_steps = s->d_chain_steps_buf;
  MM_HIP_CHECK(ctx, hipMemcpyAsync(s->d_chain_steps, steps, (size_t)s->n * sizeof(int32_t), hipMemcpyHostToDevice,
                                   ctx->stream));
  MM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // the host buffer is only borrowed
  return MM_OK;
}

int mm_state_download_errors(mm_state* s, uint32_t* errors, int32_t clear) {
  MM_REQUIRE(nullptr, s != nullptr && errors != nullptr, "mm_state_download_errors: NULL argument");
  mm_ctx* ctx = s->ctx;
  MM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (s->n == 0) return MM_OK;
  if (!s->d_errors) {  // no device-resident transition has run on this state yet
    for (int64_t i = 0; i < s->n; ++i) errors[i] = 0;
    return MM_OK;
  }
  MM_HIP_CHECK(ctx, hipMemcpyAsync(errors, s->d_errors, (size_t)s->n * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  if (clear) MM_HIP_CHECK(ctx, hipMemsetAsync(s->d_errors, 0, (size_t)s->n * sizeof(uint32_t), ctx->stream));
  MM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return MM_OK;
}

int mm_state_set_step_scale(mm_state* s, const double* scale) {
  MM_REQUIRE(nullptr, s != nullptr, "mm_state_set_step_scale: state is NULL");
  mm_ctx* ctx = s->ctx;
  MM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (!scale) {  // back to one shared step size
    MM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(s->d_step_scale);
    s->d_step_scale = nullptr;
    return MM_OK;
  }
  if (s->n == 0) return MM_OK;
  for (int64_t i = 0; i < s->n; ++i)
    MM_REQUIRE(ctx, std::isfinite(scale[i]) && scale[i] > 0.0, "mm_state_set_step_scale: factors must be positive and finite");
  if (!s->d_step_scale) MM_HIP_CHECK(ctx, hipMalloc(&s->d_step_scale, (size_t)s->n * sizeof(double)));
  MM_HIP_CHECK(ctx, hipMemcpyAsync(s->d_step_scale, scale, (size_t)s->n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  MM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // host buffer is only borrowed
  return MM_OK;
}

int mm_state_device_ptrs(mm_state* s, double** pos, double** mom, int8_t** dir) {
  MM_REQUIRE(nullptr, s != nullptr, "mm_state_device_ptrs: state is NULL");
  if (pos) *pos = s->d_pos;
  if (mom) *mom = s->d_mom;
  if (dir) *dir = s->d_dir;
  return MM_OK;
}

int mm_state_mapped_ptrs(mm_state* s, double** pos, double** mom, int8_t** dir, int32_t** status, int32_t** n_done) {
  MM_REQUIRE(nullptr, s != nullptr, "mm_state_mapped_ptrs: state is NULL");
  MM_REQUIRE(s->ctx, s->mapped, "mm_state_mapped_ptrs: the state was not allocated with mm_state_alloc_mapped");
  if (pos) *pos = s->d_pos;
  if (mom) *mom = s->d_mom;
  if (dir) *dir = s->d_dir;
  if (status) *status = s->d_status;
  if (n_done) *n_done = s->d_n_done;
  return MM_OK;
}

// ---- hot path dispatch ----------------------------------------------------------------------------------
static int check_pair(mm_ctx* ctx, const mm_model* m, mm_state* s, const char* who) {
  MM_REQUIRE(nullptr, ctx != nullptr, std::string(who) + ": ctx is NULL");
  MM_REQUIRE(ctx, m != nullptr && s != nullptr, std::string(who) + ": NULL model/state");
  MM_REQUIRE(ctx, m->ctx == ctx && s->ctx == ctx, std::string(who) + ": model/state belong to another ctx");
  MM_REQUIRE(ctx, m->dim == s->dim, std::string(who) + ": model dim != state dim");
  MM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  return MM_OK;
}

// explicit integrators cannot fail: status 0, n_done = the chain's step count (read by mm_metropolis_accept).
// Their kernels write both themselves; this is only for the call that has nothing to launch.
// One tiny kernel in the stream rather than hipMemset*Async: those were measured to stall the host between
// launches (a 3.9 ms trajectory kernel became a 6-7.5 ms pass).
static int mark_explicit_done(mm_ctx* ctx, mm_state* s, int32_t n_steps) {
  return mm_launch_fill_done(ctx, s, n_steps);
}

int mm_leapfrog_euclid(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int32_t n_steps) {
  int rc = check_pair(ctx, m, s, "mm_leapfrog_euclid");
  if (rc != MM_OK) return rc;
  MM_REQUIRE(ctx, m->rmetric == MM_RMETRIC_NONE && m->constr == MM_CONSTR_NONE,
             "mm_leapfrog_euclid: model is not a plain EuclideanMetricSystem");
  MM_REQUIRE(ctx, n_steps >= 0, "mm_leapfrog_euclid: n_steps < 0");
  if (s->n == 0) return MM_OK;
  if (n_steps == 0) return mark_explicit_done(ctx, s, 0);
  if (m->target == MM_TARGET_USER) return mm_rtc_launch_integrate(ctx, m, s, h, n_steps, nullptr);
  // the Gaussian split's exact h2 flow lives in the generic kernel only
  rc = m->gaussian_split ? -100 : mm_launch_leapfrog_euclid(ctx, m, s, h, n_steps);
  if (rc == -100 || (rc == MM_ERR_UNSUPPORTED && m->dim > 128))
    rc = mm_launch_leapfrog_generic(ctx, m, s, h, n_steps);
  return rc;
}

int mm_composition_euclid(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int32_t n_steps,
                          int32_t n_coeffs, const double* coeffs, int32_t initial_h1) {
  int rc = check_pair(ctx, m, s, "mm_composition_euclid");
  if (rc != MM_OK) return rc;
  MM_REQUIRE(ctx, m->rmetric == MM_RMETRIC_NONE && m->constr == MM_CONSTR_NONE,
             "mm_composition_euclid: model is not a plain EuclideanMetricSystem");
  MM_REQUIRE(ctx, n_steps >= 0, "mm_composition_euclid: n_steps < 0");
  MM_REQUIRE(ctx, coeffs != nullptr && n_coeffs >= 3 && n_coeffs <= MM_MAX_COMPOSITION_COEFFS && (n_coeffs & 1),
             "mm_composition_euclid: need an odd number of coefficients in [3, 16]");
  for (int i = 0; i < n_coeffs; ++i)
    MM_REQUIRE(ctx, std::isfinite(coeffs[i]), "mm_composition_euclid: non-finite coefficient");
  if (s->n == 0) return MM_OK;
  if (n_steps == 0) return mark_explicit_done(ctx, s, 0);
  // separable / dense-Gaussian models run on the kernels of the leapfrog (elementwise or FP64 MFMA); anything
  // else, and the Gaussian split's exact h2 flow, on the generic wave-per-chain kernel
  mm_comp_coefs cf{};
  cf.m = n_coeffs;
  cf.initial_h1 = initial_h1 != 0;
  for (int i = 0; i < n_coeffs; ++i) cf.c[i] = coeffs[i];
  if (m->target == MM_TARGET_USER) return mm_rtc_launch_integrate(ctx, m, s, h, n_steps, &cf);
  rc = m->gaussian_split ? -100 : mm_launch_composition_euclid(ctx, m, s, h, n_steps, cf);
  if (rc == -100 || (rc == MM_ERR_UNSUPPORTED && m->dim > 128))
    rc = mm_launch_composition_generic(ctx, m, s, h, n_steps, n_coeffs, coeffs, initial_h1 != 0);
  return rc;
}

static int finish_counters(mm_ctx* ctx, mm_counters* counters) {
  if (!counters) return MM_OK;
  MM_HIP_CHECK(ctx, hipMemcpyAsync(counters, ctx->d_counters, sizeof(mm_counters), hipMemcpyDeviceToHost, ctx->stream));
  MM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return MM_OK;
}

int mm_implicit_leapfrog(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int32_t n_steps,
                         const mm_fp_opts* opts, mm_counters* counters) {
  int rc = check_pair(ctx, m, s, "mm_implicit_leapfrog");
  if (rc != MM_OK) return rc;
  MM_REQUIRE(ctx, n_steps >= 0, "mm_implicit_leapfrog: n_steps < 0");
  mm_fp_opts o = {1e-9, 1e10, 100, MM_NORM_LINF, MM_FP_DIRECT, MM_NORM_LINF, 2e-8};
  if (opts) o = *opts;
  MM_REQUIRE(ctx, o.max_iters >= 0 && (o.norm == 0 || o.norm == 1) && (o.rev_norm == 0 || o.rev_norm == 1) &&
                      (o.solver == MM_FP_DIRECT || o.solver == MM_FP_STEFFENSEN),
             "mm_implicit_leapfrog: bad solver options");
  if (m->rmetric == MM_RMETRIC_NONE) {
    // The reference runs this integrator on any System (tests/test_integrators.py:435-462).  On a plain
    // Euclidean-metric system dh2_dpos = 0 and dh2_dmom does not depend on pos, so B and B* are the identity,
    // C and C* are both pos += t M^-1 mom (their fixed-point solves reproduce the explicit value on the second
    // evaluation) and every reversibility check passes exactly: the step is A(t) C(t) C(t) A(t), i.e. the
    // composition with coefficients (1, 1, 0, 1, 1) -- a leapfrog step of size 2t taken as two half drifts.
    MM_REQUIRE(ctx, m->constr == MM_CONSTR_NONE && !m->gaussian_split,
               "mm_implicit_leapfrog: needs a Riemannian-metric or a plain Euclidean-metric system");
    MM_REQUIRE(ctx, o.max_iters >= 2, "mm_implicit_leapfrog: the position solve needs max_iters >= 2");
    const double coeffs[5] = {1.0, 1.0, 0.0, 1.0, 1.0};
    rc = mm_composition_euclid(ctx, m, s, h, n_steps, 5, coeffs, 1);
    if (rc != MM_OK) return rc;
    if (counters) {
      *counters = mm_counters{};
      counters->n_grad = (int64_t)s->n * (2 * (int64_t)n_steps + 1);
    }
    return MM_OK;
  }
  MM_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_counters, 0, sizeof(mm_counters), ctx->stream));
  if (s->n > 0) {
    rc = mm_launch_implicit_leapfrog(ctx, m, s, h, n_steps, o, ctx->d_counters);
    if (rc != MM_OK) return rc;
  }
  return finish_counters(ctx, counters);
}

int mm_implicit_midpoint(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int32_t n_steps,
                         const mm_fp_opts* opts, mm_counters* counters) {
  int rc = check_pair(ctx, m, s, "mm_implicit_midpoint");
  if (rc != MM_OK) return rc;
  MM_REQUIRE(ctx, m->constr == MM_CONSTR_NONE, "mm_implicit_midpoint: constrained systems are not supported");
  MM_REQUIRE(ctx, n_steps >= 0, "mm_implicit_midpoint: n_steps < 0");
  mm_fp_opts o = {1e-9, 1e10, 100, MM_NORM_LINF, MM_FP_DIRECT, MM_NORM_LINF, 2e-8};
  if (opts) o = *opts;
  if (m->target == MM_TARGET_USER && m->rmetric != MM_RMETRIC_USER) {
    mm_set_error(ctx, "mm_implicit_midpoint: on Euclidean-metric systems user-defined targets run on the explicit "
                      "integrators only");
    return MM_ERR_UNSUPPORTED;
  }
  MM_REQUIRE(ctx, o.max_iters >= 0 && (o.norm == 0 || o.norm == 1) && (o.rev_norm == 0 || o.rev_norm == 1) &&
                      (o.solver == MM_FP_DIRECT || o.solver == MM_FP_STEFFENSEN),
             "mm_implicit_midpoint: bad solver options");
  MM_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_counters, 0, sizeof(mm_counters), ctx->stream));
  if (s->n > 0) {
    rc = (m->rmetric != MM_RMETRIC_NONE)
             ? mm_launch_implicit_midpoint_riemann(ctx, m, s, h, n_steps, o, ctx->d_counters)
             : mm_launch_implicit_midpoint_euclid(ctx, m, s, h, n_steps, o, ctx->d_counters);
    if (rc != MM_OK) return rc;
  }
  return finish_counters(ctx, counters);
}

int mm_constrained_leapfrog(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int32_t n_steps,
                            const mm_proj_opts* opts, mm_counters* counters) {
  int rc = check_pair(ctx, m, s, "mm_constrained_leapfrog");
  if (rc != MM_OK) return rc;
  MM_REQUIRE(ctx, m->constr != MM_CONSTR_NONE, "mm_constrained_leapfrog: model has no constraint");
  MM_REQUIRE(ctx, n_steps >= 0, "mm_constrained_leapfrog: n_steps < 0");
  mm_proj_opts o = {1e-9, 1e-8, 1e10, 50, MM_NORM_LINF, MM_PROJ_NEWTON, MM_NORM_LINF, 2e-8, 1, 10};
  if (opts) o = *opts;
  MM_REQUIRE(ctx, o.max_iters >= 0 && o.n_inner >= 1 && (o.norm == 0 || o.norm == 1) &&
                      (o.rev_norm == 0 || o.rev_norm == 1) && o.solver >= MM_PROJ_NEWTON &&
                      o.solver <= MM_PROJ_NEWTON_LINE_SEARCH && o.max_line_search_iters >= 0,
             "mm_constrained_leapfrog: bad solver options");
  MM_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_counters, 0, sizeof(mm_counters), ctx->stream));
  if (s->n > 0) {
    rc = mm_launch_constrained_leapfrog(ctx, m, s, h, n_steps, o, ctx->d_counters);
    if (rc != MM_OK) return rc;
  }
  return finish_counters(ctx, counters);
}

// System.h for every chain of s into d_out[N] (device): h1 + h2 of the model's system class.
static int launch_hamiltonian(mm_ctx* ctx, const mm_model* m, mm_state* s, double* d_out) {
  int rc = (m->rmetric != MM_RMETRIC_NONE)   ? mm_launch_riemann_aux(ctx, m, s, 0, d_out, nullptr)
           : (m->target == MM_TARGET_USER)   ? mm_rtc_launch_hamiltonian(ctx, m, s, d_out)
                                             : mm_launch_euclid_hamiltonian(ctx, m, s, d_out);
  if (rc == MM_OK && m->dens_wrt_ambient) rc = mm_launch_constrained_add_log_det_sqrt_gram(ctx, m, s, d_out);
  return rc;
}

int mm_hamiltonian(mm_ctx* ctx, const mm_model* m, mm_state* s, double* h) {
  int rc = check_pair(ctx, m, s, "mm_hamiltonian");
  if (rc != MM_OK) return rc;
  MM_REQUIRE(ctx, h != nullptr, "mm_hamiltonian: h is NULL");
  if (s->n == 0) return MM_OK;
  rc = launch_hamiltonian(ctx, m, s, s->d_scratch);
  if (rc != MM_OK) return rc;
  MM_HIP_CHECK(ctx, hipMemcpyAsync(h, s->d_scratch, (size_t)s->n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  MM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return MM_OK;
}

int mm_dh_dmom(mm_ctx* ctx, const mm_model* m, mm_state* s, double* out) {
  int rc = check_pair(ctx, m, s, "mm_dh_dmom");
  if (rc != MM_OK) return rc;
  MM_REQUIRE(ctx, out != nullptr, "mm_dh_dmom: out is NULL");
  if (s->n == 0) return MM_OK;
  rc = (m->rmetric != MM_RMETRIC_NONE) ? mm_launch_riemann_aux(ctx, m, s, 1, s->d_scratch, nullptr)
                                       : mm_launch_euclid_dh_dmom(ctx, m, s, s->d_scratch);
  if (rc != MM_OK) return rc;
  MM_HIP_CHECK(ctx, hipMemcpyAsync(out, s->d_scratch, (size_t)s->n * s->dim * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  MM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return MM_OK;
}

// mom = M^{1/2} z with z already in s->d_scratch (device), then the cotangent projection of constrained systems
static int sample_momentum_from_scratch(mm_ctx* ctx, const mm_model* m, mm_state* s) {
  int rc = (m->rmetric != MM_RMETRIC_NONE) ? mm_launch_riemann_aux(ctx, m, s, 2, nullptr, s->d_scratch)
                                           : mm_launch_euclid_sample_momentum(ctx, m, s, s->d_scratch);
  if (rc != MM_OK) return rc;
  if (m->constr != MM_CONSTR_NONE) rc = mm_launch_constrained_project_momentum(ctx, m, s);  // systems.py:614-616
  return rc;
}

int mm_sample_momentum(mm_ctx* ctx, const mm_model* m, mm_state* s, const double* z) {
  int rc = check_pair(ctx, m, s, "mm_sample_momentum");
  if (rc != MM_OK) return rc;
  MM_REQUIRE(ctx, z != nullptr, "mm_sample_momentum: z is NULL");
  if (s->n == 0) return MM_OK;
  MM_HIP_CHECK(ctx, hipMemcpyAsync(s->d_scratch, z, (size_t)s->n * s->dim * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  rc = sample_momentum_from_scratch(ctx, m, s);
  if (rc != MM_OK) return rc;
  MM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return MM_OK;
}

// ---- momentum transitions (transitions.py:129-198) -----------------------------------------------------------
// z: host draws, or NULL = device draws of `transition` (the state must have an RNG: mm_state_set_rng)
static int momentum_refresh(mm_ctx* ctx, const mm_model* m, mm_state* s, const double* z, double coeff,
                            uint64_t transition, const char* who) {
  int rc = check_pair(ctx, m, s, who);
  if (rc != MM_OK) return rc;
  MM_REQUIRE(ctx, coeff >= 0.0 && coeff <= 1.0, "mom_resample_coeff should have a value in the interval [0, 1].");
  MM_REQUIRE(ctx, z != nullptr || s->rng_on, std::string(who) + ": the state has no device RNG (mm_state_set_rng)");
  if (coeff == 0.0 || s->n == 0) return MM_OK;  // transitions.py:193: the momentum is left alone
  const size_t nd = (size_t)s->n * s->dim;
  if (coeff != 1.0) {
    if (s->mom_save_elems < nd) {
      (void)hipFree(s->d_mom_save);
      s->d_mom_save = nullptr;
      s->mom_save_elems = 0;
      MM_HIP_CHECK(ctx, hipMalloc(&s->d_mom_save, nd * sizeof(double)));
      s->mom_save_elems = nd;
    }
    MM_HIP_CHECK(ctx, hipMemcpyAsync(s->d_mom_save, s->d_mom, nd * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  }
  if (z) {
    MM_HIP_CHECK(ctx, hipMemcpyAsync(s->d_scratch, z, nd * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  } else {
    rc = mm_launch_rng_normal(ctx, s->d_scratch, s->n, s->dim, s->rng_seed, s->rng_chain_offset, transition);
    if (rc != MM_OK) return rc;
  }
  rc = sample_momentum_from_scratch(ctx, m, s);  // mom <- independent draw (projected for constrained systems)
  if (rc != MM_OK) return rc;
  if (coeff != 1.0) {
    rc = mm_launch_axpby(ctx, s->d_mom, s->d_mom_save, std::sqrt(1.0 - coeff * coeff), coeff, nd);
    if (rc != MM_OK) return rc;
  }
  if (z) MM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // z is only borrowed; device draws stay asynchronous
  return MM_OK;
}

int mm_momentum_refresh(mm_ctx* ctx, const mm_model* m, mm_state* s, const double* z, double coeff) {
  MM_REQUIRE(ctx, z != nullptr || coeff == 0.0, "mm_momentum_refresh: z is NULL");
  return momentum_refresh(ctx, m, s, z, coeff, 0, "mm_momentum_refresh");
}

int mm_momentum_refresh_rng(mm_ctx* ctx, const mm_model* m, mm_state* s, double coeff, uint64_t transition) {
  return momentum_refresh(ctx, m, s, nullptr, coeff, transition, "mm_momentum_refresh_rng");
}

int mm_state_set_rng(mm_state* s, uint64_t seed, uint64_t chain_offset) {
  MM_REQUIRE(nullptr, s != nullptr, "mm_state_set_rng: state is NULL");
  s->rng_on = true;
  s->rng_seed = seed;
  s->rng_chain_offset = chain_offset;
  return MM_OK;
}

// ---- Metropolis accept (transitions.py:275-315) ------------------------------------------------------------
static int metropolis_accept(mm_ctx* ctx, const mm_model* m, mm_state* s, mm_state* prop, const double* u,
                             uint64_t transition, double* accept_prob, int8_t* accepted, const char* who) {
  int rc = check_pair(ctx, m, s, who);
  if (rc != MM_OK) return rc;
  MM_REQUIRE(ctx, prop != nullptr && prop != s && prop->ctx == ctx && prop->n == s->n && prop->dim == s->dim,
             std::string(who) + ": proposal must be a distinct state of the same shape");
  MM_REQUIRE(ctx, u != nullptr || s->rng_on, std::string(who) + ": the state has no device RNG (mm_state_set_rng)");
  if (s->n == 0) return MM_OK;
  const size_t n = (size_t)s->n;
  // h(state) -> state scratch[0..N), h(proposal) -> proposal scratch[0..N)
  rc = launch_hamiltonian(ctx, m, s, s->d_scratch);
  if (rc != MM_OK) return rc;
  rc = launch_hamiltonian(ctx, m, prop, prop->d_scratch);
  if (rc != MM_OK) return rc;
  // u -> device (the proposal's momentum buffer is dead after the select; use a small dedicated buffer)
  if (s->tr_elems < 2 * n) {
    (void)hipFree(s->d_tr);
    s->d_tr = nullptr;
    s->tr_elems = 0;
    MM_HIP_CHECK(ctx, hipMalloc(&s->d_tr, 2 * n * sizeof(double) + n));
    s->tr_elems = 2 * n;
  }
  if (!s->d_errors) {  // the sticky error word of this state's device-resident transitions
    MM_HIP_CHECK(ctx, hipMalloc(&s->d_errors, n * sizeof(uint32_t)));
    MM_HIP_CHECK(ctx, hipMemsetAsync(s->d_errors, 0, n * sizeof(uint32_t), ctx->stream));
  }
  double* d_u = s->d_tr;
  double* d_prob = s->d_tr + n;
  int8_t* d_acc = reinterpret_cast<int8_t*>(s->d_tr + 2 * n);
  if (u) {
    MM_HIP_CHECK(ctx, hipMemcpyAsync(d_u, u, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  } else {
    rc = mm_launch_rng_uniform(ctx, d_u, s->n, s->rng_seed, s->rng_chain_offset, transition);
    if (rc != MM_OK) return rc;
  }
  rc = mm_launch_metropolis_select(ctx, s, prop, s->d_scratch, prop->d_scratch, d_u, d_prob, d_acc);
  if (rc != MM_OK) return rc;
  if (accept_prob)
    MM_HIP_CHECK(ctx, hipMemcpyAsync(accept_prob, d_prob, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  if (accepted) MM_HIP_CHECK(ctx, hipMemcpyAsync(accepted, d_acc, n, hipMemcpyDeviceToHost, ctx->stream));
  if (u || accept_prob || accepted) MM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // u is only borrowed
  return MM_OK;
}

int mm_metropolis_accept(mm_ctx* ctx, const mm_model* m, mm_state* s, mm_state* prop, const double* u,
                         double* accept_prob, int8_t* accepted) {
  MM_REQUIRE(ctx, u != nullptr, "mm_metropolis_accept: u is NULL");
  return metropolis_accept(ctx, m, s, prop, u, 0, accept_prob, accepted, "mm_metropolis_accept");
}

int mm_metropolis_accept_rng(mm_ctx* ctx, const mm_model* m, mm_state* s, mm_state* prop, uint64_t transition,
                             double* accept_prob, int8_t* accepted) {
  return metropolis_accept(ctx, m, s, prop, nullptr, transition, accept_prob, accepted, "mm_metropolis_accept_rng");
}

int mm_rng_chain_steps(mm_state* s, uint64_t transition, int32_t lo, int32_t hi) {
  MM_REQUIRE(nullptr, s != nullptr, "mm_rng_chain_steps: state is NULL");
  mm_ctx* ctx = s->ctx;
  MM_REQUIRE(ctx, s->rng_on, "mm_rng_chain_steps: the state has no device RNG (mm_state_set_rng)");
  MM_REQUIRE(ctx, lo >= 0 && hi > lo, "mm_rng_chain_steps: need 0 <= lo < hi");
  MM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (s->n == 0) return MM_OK;
  if (!s->d_chain_steps_buf) MM_HIP_CHECK(ctx, hipMalloc(&s->d_chain_steps_buf, (size_t)s->n * sizeof(int32_t)));
  s->d_chain_steps = s->d_chain_steps_buf;
  return mm_launch_rng_steps(ctx, s->d_chain_steps, s->n, s->rng_seed, s->rng_chain_offset, transition, lo, hi);
}

int mm_rng_draws(mm_state* s, uint64_t transition, double* z, double* u, int32_t* steps, int32_t lo, int32_t hi) {
  MM_REQUIRE(nullptr, s != nullptr, "mm_rng_draws: state is NULL");
  mm_ctx* ctx = s->ctx;
  MM_REQUIRE(ctx, s->rng_on, "mm_rng_draws: the state has no device RNG (mm_state_set_rng)");
  MM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (s->n == 0) return MM_OK;
  const size_t n = (size_t)s->n, nd = n * s->dim;
  int rc = MM_OK;
  if (z) {
    rc = mm_launch_rng_normal(ctx, s->d_scratch, s->n, s->dim, s->rng_seed, s->rng_chain_offset, transition);
    if (rc != MM_OK) return rc;
    MM_HIP_CHECK(ctx, hipMemcpyAsync(z, s->d_scratch, nd * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    MM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  }
  if (u) {
    rc = mm_launch_rng_uniform(ctx, s->d_scratch, s->n, s->rng_seed, s->rng_chain_offset, transition);
    if (rc != MM_OK) return rc;
    MM_HIP_CHECK(ctx, hipMemcpyAsync(u, s->d_scratch, n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    MM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  }
  if (steps) {
    MM_REQUIRE(ctx, lo >= 0 && hi > lo, "mm_rng_draws: need 0 <= lo < hi");
    rc = mm_launch_rng_steps(ctx, reinterpret_cast<int32_t*>(s->d_scratch), s->n, s->rng_seed, s->rng_chain_offset,
                             transition, lo, hi);
    if (rc != MM_OK) return rc;
    MM_HIP_CHECK(ctx, hipMemcpyAsync(steps, s->d_scratch, n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    MM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  }
  return MM_OK;
}

// ---- RCCL trace gather (librccl is opened lazily: single-GPU users never load it) ---------------------
typedef struct { char internal[MM_COMM_ID_BYTES]; } mm_nccl_id;  // ncclUniqueId is 128 opaque bytes
typedef void* mm_nccl_comm;
struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(mm_nccl_id*) = nullptr;
  int (*CommInitRank)(mm_nccl_comm*, int, mm_nccl_id, int) = nullptr;
  int (*CommDestroy)(mm_nccl_comm) = nullptr;
  int (*AllGather)(const void*, void*, size_t, int, mm_nccl_comm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  int (*CommCount)(const mm_nccl_comm, int*) = nullptr;     // (optional: mm_comm_count falls back to what it was created with)
  int (*CommUserRank)(const mm_nccl_comm, int*) = nullptr;
};
static RcclApi g_rccl;
static std::mutex g_rccl_mu;

static int rccl_load(const mm_ctx* ctx) {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.lib) return MM_OK;
  void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) {
    mm_set_error(ctx, std::string("cannot load librccl: ") + dlerror());
    return MM_ERR_RCCL;
  }
  g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
  g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(lib, "ncclCommInitRank");
  g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(lib, "ncclCommDestroy");
  g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(lib, "ncclAllGather");
  g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(lib, "ncclGetErrorString");
  g_rccl.CommCount = (decltype(g_rccl.CommCount))dlsym(lib, "ncclCommCount");
  g_rccl.CommUserRank = (decltype(g_rccl.CommUserRank))dlsym(lib, "ncclCommUserRank");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllGather) {
    mm_set_error(ctx, "librccl is missing a required ncclXxx symbol");
    return MM_ERR_RCCL;
  }
  g_rccl.lib = lib;
  return MM_OK;
}

static int rccl_fail(const mm_ctx* ctx, const char* what, int code) {
  mm_set_error(ctx, std::string(what) + ": " +
                        (g_rccl.GetErrorString ? g_rccl.GetErrorString(code) : "RCCL error"));
  return MM_ERR_RCCL;
}

struct mm_comm {
  mm_ctx* ctx = nullptr;
  mm_nccl_comm comm = nullptr;
  int n_ranks = 1, rank = 0;
  double* d_gather = nullptr;
  size_t gather_elems = 0;
  // overlapped gather: own stream, snapshot of the shard, pinned host landing buffer
  hipStream_t stream = nullptr;
  hipEvent_t snap_ready = nullptr, gather_done = nullptr;
  double* d_snap = nullptr;
  size_t snap_elems = 0;
  double* h_pinned = nullptr;
  size_t pinned_elems = 0;
  size_t last_total = 0;
  bool last_host = false, pending = false;
};

int mm_comm_unique_id(uint8_t id[MM_COMM_ID_BYTES]) {
  MM_REQUIRE(nullptr, id != nullptr, "mm_comm_unique_id: id is NULL");
  int rc = rccl_load(nullptr);
  if (rc != MM_OK) return rc;
  mm_nccl_id uid;
  const int e = g_rccl.GetUniqueId(&uid);
  if (e != 0) return rccl_fail(nullptr, "ncclGetUniqueId", e);
  std::memcpy(id, uid.internal, MM_COMM_ID_BYTES);
  return MM_OK;
}

int mm_comm_create(mm_ctx* ctx, int32_t n_ranks, int32_t rank, const uint8_t id[MM_COMM_ID_BYTES],
                   mm_comm** out) {
  MM_REQUIRE(nullptr, ctx != nullptr, "mm_comm_create: ctx is NULL");
  MM_REQUIRE(ctx, out != nullptr && id != nullptr, "mm_comm_create: NULL argument");
  *out = nullptr;
  MM_REQUIRE(ctx, n_ranks >= 1 && rank >= 0 && rank < n_ranks, "mm_comm_create: bad rank / n_ranks");
  int rc = rccl_load(ctx);
  if (rc != MM_OK) return rc;
  MM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  mm_nccl_id uid;
  std::memcpy(uid.internal, id, MM_COMM_ID_BYTES);
  mm_comm* c = new mm_comm();
  c->ctx = ctx;
  c->n_ranks = n_ranks;
  c->rank = rank;
  const int e = g_rccl.CommInitRank(&c->comm, n_ranks, uid, rank);
  if (e != 0) {
    delete c;
    return rccl_fail(ctx, "ncclCommInitRank", e);
  }
  if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&c->snap_ready, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->gather_done, hipEventDisableTiming) != hipSuccess) {
    mm_set_error(ctx, "mm_comm_create: stream / event creation failed");
    mm_comm_destroy(c);
    return MM_ERR_HIP;
  }
  *out = c;
  return MM_OK;
}

// Ranks the communicator spans and this process' rank in it, as RCCL itself reports them (ncclCommCount /
// ncclCommUserRank): what a launcher prints as "ranks seen" - not what it asked for.
int mm_comm_count(mm_comm* c, int32_t* n_ranks, int32_t* rank) {
  MM_REQUIRE(nullptr, c != nullptr, "mm_comm_count: comm is NULL");
  mm_ctx* ctx = c->ctx;
  int n = c->n_ranks, r = c->rank;
  if (g_rccl.CommCount && g_rccl.CommUserRank) {
    int e = g_rccl.CommCount(c->comm, &n);
    if (e != 0) return rccl_fail(ctx, "ncclCommCount", e);
    e = g_rccl.CommUserRank(c->comm, &r);
    if (e != 0) return rccl_fail(ctx, "ncclCommUserRank", e);
  }
  if (n_ranks) *n_ranks = n;
  if (rank) *rank = r;
  return MM_OK;
}

int mm_comm_destroy(mm_comm* c) {
  if (!c) return MM_OK;
  (void)hipSetDevice(c->ctx->device);
  (void)hipStreamSynchronize(c->ctx->stream);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->comm) (void)g_rccl.CommDestroy(c->comm);
  (void)hipFree(c->d_gather);
  (void)hipFree(c->d_snap);
  if (c->h_pinned) (void)hipHostFree(c->h_pinned);
  if (c->snap_ready) (void)hipEventDestroy(c->snap_ready);
  if (c->gather_done) (void)hipEventDestroy(c->gather_done);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  return MM_OK;
}

int mm_comm_allgather_pos(mm_comm* c, mm_state* s, double* pos_all) {
  MM_REQUIRE(nullptr, c != nullptr, "mm_comm_allgather_pos: comm is NULL");
  mm_ctx* ctx = c->ctx;
  MM_REQUIRE(ctx, s != nullptr && pos_all != nullptr && s->ctx == ctx, "mm_comm_allgather_pos: bad argument");
  MM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const size_t local = (size_t)s->n * s->dim, total = local * c->n_ranks;
  if (local == 0) return MM_OK;
  if (c->gather_elems < total) {
    (void)hipFree(c->d_gather);
    c->d_gather = nullptr;
    c->gather_elems = 0;
    if (hipMalloc(&c->d_gather, total * sizeof(double)) != hipSuccess) {
      mm_set_error(ctx, "mm_comm_allgather_pos: hipMalloc failed");
      return MM_ERR_NOMEM;
    }
    c->gather_elems = total;
  }
  const int kNcclFloat64 = 8;  // ncclDataType_t::ncclFloat64 / ncclDouble
  const int e = g_rccl.AllGather(s->d_pos, c->d_gather, local, kNcclFloat64, c->comm, ctx->stream);
  if (e != 0) return rccl_fail(ctx, "ncclAllGather", e);
  MM_HIP_CHECK(ctx, hipMemcpyAsync(pos_all, c->d_gather, total * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  MM_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return MM_OK;
}

int mm_comm_allgather_pos_async(mm_comm* c, mm_state* s, int want_host) {
  MM_REQUIRE(nullptr, c != nullptr, "mm_comm_allgather_pos_async: comm is NULL");
  mm_ctx* ctx = c->ctx;
  MM_REQUIRE(ctx, s != nullptr && s->ctx == ctx, "mm_comm_allgather_pos_async: bad argument");
  MM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const size_t local = (size_t)s->n * s->dim, total = local * c->n_ranks;
  if (local == 0) return MM_OK;
  if (c->snap_elems < local || c->gather_elems < total) {
    MM_HIP_CHECK(ctx, hipStreamSynchronize(c->stream));
    (void)hipFree(c->d_snap);
    (void)hipFree(c->d_gather);
    c->d_snap = c->d_gather = nullptr;
    c->snap_elems = c->gather_elems = 0;
    if (hipMalloc(&c->d_snap, local * sizeof(double)) != hipSuccess ||
        hipMalloc(&c->d_gather, total * sizeof(double)) != hipSuccess) {
      mm_set_error(ctx, "mm_comm_allgather_pos_async: hipMalloc failed");
      return MM_ERR_NOMEM;
    }
    c->snap_elems = local;
    c->gather_elems = total;
  }
  if (want_host && c->pinned_elems < total) {
    MM_HIP_CHECK(ctx, hipStreamSynchronize(c->stream));
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    c->h_pinned = nullptr;
    c->pinned_elems = 0;
    if (hipHostMalloc(reinterpret_cast<void**>(&c->h_pinned), total * sizeof(double), hipHostMallocDefault) != hipSuccess) {
      mm_set_error(ctx, "mm_comm_allgather_pos_async: hipHostMalloc failed");
      return MM_ERR_NOMEM;
    }
    c->pinned_elems = total;
  }
  // the previous gather must have finished reading the snapshot before it is overwritten
  if (c->pending) MM_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->stream, c->gather_done, 0));
  MM_HIP_CHECK(ctx, hipMemcpyAsync(c->d_snap, s->d_pos, local * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
  MM_HIP_CHECK(ctx, hipEventRecord(c->snap_ready, ctx->stream));
  MM_HIP_CHECK(ctx, hipStreamWaitEvent(c->stream, c->snap_ready, 0));
  const int kNcclFloat64 = 8;
  const int e = g_rccl.AllGather(c->d_snap, c->d_gather, local, kNcclFloat64, c->comm, c->stream);
  if (e != 0) return rccl_fail(ctx, "ncclAllGather", e);
  MM_HIP_CHECK(ctx, hipEventRecord(c->gather_done, c->stream));
  if (want_host)
    MM_HIP_CHECK(ctx, hipMemcpyAsync(c->h_pinned, c->d_gather, total * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  c->last_total = total;
  c->last_host = want_host != 0;
  c->pending = true;
  return MM_OK;
}

int mm_comm_wait(mm_comm* c, double* pos_all) {
  MM_REQUIRE(nullptr, c != nullptr, "mm_comm_wait: comm is NULL");
  mm_ctx* ctx = c->ctx;
  MM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  MM_HIP_CHECK(ctx, hipStreamSynchronize(c->stream));
  if (pos_all) {
    MM_REQUIRE(ctx, c->pending && c->last_host, "mm_comm_wait: no host gather is pending");
    std::memcpy(pos_all, c->h_pinned, c->last_total * sizeof(double));
  }
  c->pending = false;
  return MM_OK;
}

}  // extern "C"
