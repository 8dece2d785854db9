// User-defined targets through run-time compilation (hipRTC): the reference takes arbitrary Python callables
// (systems.py:107, 119: neg_log_dens / grad_neg_log_dens); a device kernel needs device code, so a user brings HIP
// source for two device functions and the library compiles the wave-per-chain Euclidean kernels around them
// (csrc/rtc_euclid_src.inc) for gfx950 when the model is created.  libhiprtc is opened lazily: users of the built-in
// models never load it.
#include <dlfcn.h>

#include <mutex>
#include <vector>

#include "mm_internal.h"

namespace {

const char* kRtcPreamble =
#include "rtc_euclid_src.inc"
    ;

typedef struct _mm_hiprtcProgram* mm_hiprtcProgram;
struct RtcApi {
  void* lib = nullptr;
  int (*CreateProgram)(mm_hiprtcProgram*, const char*, const char*, int, const char**, const char**) = nullptr;
  int (*CompileProgram)(mm_hiprtcProgram, int, const char**) = nullptr;
  int (*GetProgramLogSize)(mm_hiprtcProgram, size_t*) = nullptr;
  int (*GetProgramLog)(mm_hiprtcProgram, char*) = nullptr;
  int (*GetCodeSize)(mm_hiprtcProgram, size_t*) = nullptr;
  int (*GetCode)(mm_hiprtcProgram, char*) = nullptr;
  int (*DestroyProgram)(mm_hiprtcProgram*) = nullptr;
};
RtcApi g_rtc;
std::mutex g_rtc_mu;

int rtc_load(const mm_ctx* ctx) {
  std::lock_guard<std::mutex> lk(g_rtc_mu);
  if (g_rtc.lib) return MM_OK;
  void* lib = dlopen("libhiprtc.so", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) lib = dlopen("libhiprtc.so.7", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) lib = dlopen("/opt/rocm/lib/libhiprtc.so", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) {
    mm_set_error(ctx, std::string("cannot load libhiprtc (needed for user-defined targets): ") + dlerror());
    return MM_ERR_UNSUPPORTED;
  }
#define MM_SYM(field, name)                                            \
  g_rtc.field = (decltype(g_rtc.field))dlsym(lib, name);               \
  if (!g_rtc.field) {                                                  \
    mm_set_error(ctx, std::string("libhiprtc is missing ") + name);    \
    return MM_ERR_UNSUPPORTED;                                         \
  }
  MM_SYM(CreateProgram, "hiprtcCreateProgram")
  MM_SYM(CompileProgram, "hiprtcCompileProgram")
  MM_SYM(GetProgramLogSize, "hiprtcGetProgramLogSize")
  MM_SYM(GetProgramLog, "hiprtcGetProgramLog")
  MM_SYM(GetCodeSize, "hiprtcGetCodeSize")
  MM_SYM(GetCode, "hiprtcGetCode")
  MM_SYM(DestroyProgram, "hiprtcDestroyProgram")
#undef MM_SYM
  g_rtc.lib = lib;
  return MM_OK;
}

// must match the structs of rtc_euclid_src.inc
struct RtcModel {
  int metric_kind, dim;
  const double* params;
  const double* minv;
};
struct RtcCoefs {
  int m;
  int initial_h1;
  double c[16];
};

int waves_per_block(int dim, size_t* lds_bytes) {
  int w = 4;
  while (w > 1 && (size_t)w * 3 * dim * sizeof(double) > 60 * 1024) w >>= 1;
  *lds_bytes = (size_t)w * 3 * dim * sizeof(double);
  return w;
}

}  // namespace

// Compile `user_src` behind the kernel preamble and attach the module to the model.
int mm_rtc_attach(mm_ctx* ctx, mm_model* m, const char* user_src) {
  int rc = rtc_load(ctx);
  if (rc != MM_OK) return rc;
  const std::string src = std::string(kRtcPreamble) + "\n#line 1 \"user_target.hip\"\n" + user_src + "\n";
  mm_hiprtcProgram prog = nullptr;
  if (g_rtc.CreateProgram(&prog, src.c_str(), "mici_amd_user_target.hip", 0, nullptr, nullptr) != 0) {
    mm_set_error(ctx, "hiprtcCreateProgram failed");
    return MM_ERR_HIP;
  }
  const char* opts[] = {"--offload-arch=gfx950", "-O3", "-ffp-contract=on", "-std=c++17"};
  const int crc = g_rtc.CompileProgram(prog, 4, opts);
  if (crc != 0) {
    size_t n = 0;
    std::string log;
    if (g_rtc.GetProgramLogSize(prog, &n) == 0 && n > 1) {
      log.resize(n);
      (void)g_rtc.GetProgramLog(prog, &log[0]);
    }
    (void)g_rtc.DestroyProgram(&prog);
    mm_set_error(ctx, "user target does not compile (hipRTC, gfx950):\n" + log.substr(0, 4000));
    return MM_ERR_INVALID;
  }
  size_t code_size = 0;
  std::vector<char> code;
  if (g_rtc.GetCodeSize(prog, &code_size) != 0 || code_size == 0) {
    (void)g_rtc.DestroyProgram(&prog);
    mm_set_error(ctx, "hiprtcGetCodeSize failed");
    return MM_ERR_HIP;
  }
  code.resize(code_size);
  const int grc = g_rtc.GetCode(prog, code.data());
  (void)g_rtc.DestroyProgram(&prog);
  if (grc != 0) {
    mm_set_error(ctx, "hiprtcGetCode failed");
    return MM_ERR_HIP;
  }
  hipModule_t mod = nullptr;
  MM_HIP_CHECK(ctx, hipModuleLoadData(&mod, code.data()));
  hipFunction_t f_int = nullptr, f_h = nullptr;
  if (hipModuleGetFunction(&f_int, mod, "mm_rtc_integrate") != hipSuccess ||
      hipModuleGetFunction(&f_h, mod, "mm_rtc_hamiltonian") != hipSuccess) {
    (void)hipModuleUnload(mod);
    mm_set_error(ctx, "compiled user module lacks the expected kernels");
    return MM_ERR_HIP;
  }
  m->rtc_module = mod;
  m->rtc_integrate = f_int;
  m->rtc_hamiltonian = f_h;
  return MM_OK;
}

void mm_rtc_detach(mm_model* m) {
  if (m->rtc_module) (void)hipModuleUnload(reinterpret_cast<hipModule_t>(m->rtc_module));
  m->rtc_module = nullptr;
}

int mm_rtc_launch_integrate(mm_ctx* ctx, const mm_model* m, mm_state* s, double h, int n_steps,
                            const mm_comp_coefs* cf) {
  size_t lds;
  const int w = waves_per_block(s->dim, &lds);
  if (lds > 64 * 1024) {
    mm_set_error(ctx, "user-target kernels: dim too large for the LDS tile of a wave");
    return MM_ERR_UNSUPPORTED;
  }
  RtcModel mv{m->metric_kind, m->dim, m->d_target_params, m->d_metric_inv};
  RtcCoefs c{};
  int leapfrog = cf ? 0 : 1;
  if (cf) {
    c.m = cf->m;
    c.initial_h1 = cf->initial_h1;
    for (int i = 0; i < cf->m; ++i) c.c[i] = cf->c[i];
  }
  int64_t n = s->n;
  void* args[] = {&mv,          &s->d_pos, &s->d_mom, &s->d_dir, &s->d_step_scale, &s->d_chain_steps,
                  &s->d_status, &s->d_n_done, &n,      &h,        &n_steps,         &leapfrog,
                  &c};
  const unsigned blocks = (unsigned)((s->n + w - 1) / w);
  MM_HIP_CHECK(ctx, hipModuleLaunchKernel(reinterpret_cast<hipFunction_t>(m->rtc_integrate), blocks, 1, 1, 64 * w, 1, 1,
                                          (unsigned)lds, ctx->stream, args, nullptr));
  return MM_OK;
}

int mm_rtc_launch_hamiltonian(mm_ctx* ctx, const mm_model* m, mm_state* s, double* d_h) {
  size_t lds;
  const int w = waves_per_block(s->dim, &lds);
  if (lds > 64 * 1024) {
    mm_set_error(ctx, "user-target kernels: dim too large for the LDS tile of a wave");
    return MM_ERR_UNSUPPORTED;
  }
  RtcModel mv{m->metric_kind, m->dim, m->d_target_params, m->d_metric_inv};
  int64_t n = s->n;
  void* args[] = {&mv, &s->d_pos, &s->d_mom, &n, &d_h};
  const unsigned blocks = (unsigned)((s->n + w - 1) / w);
  MM_HIP_CHECK(ctx, hipModuleLaunchKernel(reinterpret_cast<hipFunction_t>(m->rtc_hamiltonian), blocks, 1, 1, 64 * w, 1,
                                          1, (unsigned)lds, ctx->stream, args, nullptr));
  return MM_OK;
}
