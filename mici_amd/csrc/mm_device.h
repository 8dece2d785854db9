// Device-side helpers shared by the wave-per-chain kernels: wavefront reductions and the built-in
// targets / constraints evaluated cooperatively by one 64-lane wave with the position vector in LDS.
// Closed forms: SURVEY.md section 8d / Appendix A (NumPy twins in oracle/models.py).
#pragma once

#ifdef MM_RTC_BUILD
// compiled at run time (hipRTC) around user source: no host headers, the ABI header comes from memory (mm_rtc.hip)
#include "mici_amd.h"
#ifdef MM_RTC_USER_TARGET
__device__ double mm_user_grad(const double* q, int i, int dim, const double* params);
__device__ double mm_user_nld_term(const double* q, int i, int dim, const double* params);
#endif
#else
#include "mm_internal.h"
#endif

namespace mmdev {

// signed time step of a chain: dir * step_size, times the chain's own scale when the state carries per-chain
// step sizes (mm_state_set_step_scale: step-size adaptation runs every chain at its own step size)
__device__ __forceinline__ double signed_step(const int8_t* __restrict__ dir, const double* __restrict__ scale,
                                              int64_t chain, double step_size) {
  return (double)dir[chain] * (scale ? step_size * scale[chain] : step_size);
}

// number of steps a chain takes in this call: n_steps, capped by the chain's own count when the state carries
// per-chain trajectory lengths (mm_state_set_chain_steps: MetropolisRandomIntegrationTransition draws one
// n_step per chain and transition, transitions.py:355-402)
__device__ __forceinline__ int chain_steps(const int32_t* __restrict__ cs, int64_t chain, int n_steps) {
  if (!cs) return n_steps;
  const int c = cs[chain];
  return c < 0 ? 0 : (c < n_steps ? c : n_steps);
}

// Order LDS traffic between the lanes of ONE wave (the wave executes DS instructions in order; this
// only stops the compiler from moving loads/stores across the exchange point).
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Cross-lane data movement on the VALU (DPP) instead of ds_bpermute: a bpermute is an LDS-crossbar
// round trip (~100+ cycles) that a lone wave cannot hide, and a 6-stage butterfly chains six of them;
// DPP moves cost one VALU slot each.  quad_perm / row_mirror / row_half_mirror act inside rows of 16.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  const long long b = __double_as_longlong(v);
  int lo = (int)(b & 0xffffffffLL), hi = (int)(b >> 32);
  // bound_ctrl with a dead `old`: the controls used here are full permutations of a row, so no lane falls back to
  // `old`, and the compiler need not copy the source into the destination first (3 instead of 5 VALU slots a step)
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
constexpr int kDppXor1 = 0xB1;        // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;        // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141; // lane i <-> 7 - i inside each 8-lane half row
constexpr int kDppMirror = 0x140;     // lane i <-> 15 - i inside each 16-lane row

__device__ __forceinline__ double readlane_f64(double v, int lane) {  // lane must be wave-uniform
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), lane);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// sum over the 8 lanes sharing lane>>3 (result in all 8)
__device__ __forceinline__ double group8_sum(double v) {
  v += dpp_move<kDppXor1>(v);
  v += dpp_move<kDppXor2>(v);
  v += dpp_move<kDppHalfMirror>(v);
  return v;
}

// (Round 4 measured the cross-row part on gfx950's v_permlane16_swap / v_permlane32_swap - two swaps and one add per
// butterfly step, the result read back with v_readfirstlane to keep the callers' control flow scalar - against the four
// v_readlane pairs below: c3(a) 1.250e7 against 1.263e7 steps/s.  The readlane form stays.)
__device__ __forceinline__ double wave_sum(double v) {
  v = group8_sum(v);
  v += dpp_move<kDppMirror>(v);  // every lane of a 16-lane row now holds the row sum
  return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

// ---- lean FP64 reciprocal / division / square root for well-scaled arguments ---------------------------------------
// The compiler's IEEE expansions carry range scaling and special-case fix-ups (v_div_scale / v_div_fmas / v_div_fixup,
// ldexp + class tests around v_rsq): 12-18 dependent instructions each, and the lane-per-chain kernels are bound by
// exactly that - the length of one wave's dependent FP64 stream.  These versions are Newton iterations on v_rcp_f64 /
// v_rsq_f64 with a final residual correction: results within 1 ulp for normal, non-extreme arguments; zero, infinite
// and NaN arguments of rcp_nr / fdiv give inf / NaN as the expansions would (possibly a NaN where IEEE gives inf or 0 -
// every caller treats both as "not finite").  sqrt_rsqrt takes x >= 0 (sums of squares; the Gram value of one
// constraint behind its positivity check) or NaN: sqrt(0) = 0 exactly (the torus constraint at x = y = 0 is
// finite in the reference), with 1 / sqrt(0) a huge finite number instead of inf; x = inf gives NaN for both.
__device__ __forceinline__ double rcp_nr(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0);
  return __builtin_fma(r, e, r);
}
__device__ __forceinline__ double fdiv(double a, double b) {
  const double r = rcp_nr(b);
  const double q = a * r;
  return __builtin_fma(__builtin_fma(-b, q, a), r, q);
}
// s = sqrt(x) and rs = 1 / sqrt(x) from one v_rsq_f64 (Goldschmidt, two steps + a residual correction of s)
__device__ __forceinline__ void sqrt_rsqrt(double x, double* s, double* rs) {
  // rsq(0) = inf would turn g = x y into NaN: clamped (one v_min_f64; every finite x > 0 has rsq(x) < 1e162), x = 0
  // runs through the iteration as g = 0 exactly
  const double y = __builtin_fmin(__builtin_amdgcn_rsq(x), 1e300);
  double g = x * y, h = 0.5 * y;
  double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  const double d = __builtin_fma(-g, g, x);
  *s = __builtin_fma(d, h, g);
  *rs = h + h;
}
__device__ __forceinline__ double fsqrt(double x) {
  double s, rs;
  sqrt_rsqrt(x, &s, &rs);
  return s;
}

// max that propagates NaN (np.abs(x).max() semantics, solvers.py:25-27)
__device__ __forceinline__ double nanmax(double a, double b) {
  // branch-free: v_max_f64 returns the other operand when one is a NaN, so the NaN case is patched in with one
  // unordered compare (a + b is a NaN whenever either operand is).  The nested-ternary form compiled to three
  // exec-mask regions per call - a norm of a D-vector sits in every solver iteration of every kernel.
  const double m = __builtin_fmax(a, b);
  return __builtin_isunordered(a, b) ? a + b : m;
}

__device__ __forceinline__ double wave_max(double v) {
  v = nanmax(v, dpp_move<kDppXor1>(v));
  v = nanmax(v, dpp_move<kDppXor2>(v));
  v = nanmax(v, dpp_move<kDppHalfMirror>(v));
  v = nanmax(v, dpp_move<kDppMirror>(v));
  return nanmax(nanmax(readlane_f64(v, 0), readlane_f64(v, 16)),
                nanmax(readlane_f64(v, 32), readlane_f64(v, 48)));
}

// broadcast from a wave-uniform source lane (v_readlane, no LDS crossbar)
__device__ __forceinline__ double wave_bcast(double v, int src_lane) { return readlane_f64(v, src_lane); }

// norm over the first `dim` entries held one-per-lane-slot: elements i = lane, lane+64, ...
// kind 0: max |x| (maximum_norm, solvers.py:25-27); kind 1: sqrt(sum x^2) (euclidean_norm, :20-22)
__device__ __forceinline__ double wave_norm_accum(double acc, double x, int kind) {
  return kind == MM_NORM_LINF ? nanmax(acc, fabs(x)) : acc + x * x;
}
__device__ __forceinline__ double wave_norm_finish(double acc, int kind) {
  return kind == MM_NORM_LINF ? wave_max(acc) : sqrt(wave_sum(acc));
}

// ---- targets ------------------------------------------------------------------------------------------
struct TargetAux {
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
};

// Wave-collective: reductions a target needs before its per-element gradient can be formed.
// TRIG = false compiles the transcendental targets (torus, funnel) out: the constant tables of
// exp / atan2 / sin / cos get hoisted into long-lived VGPRs, which the register-resident-metric
// kernels cannot spare (measured: 18 VGPRs for exp alone).
template <bool TRIG = true>
__device__ __forceinline__ TargetAux target_prepare(int target, const double* q, int dim,
                                                    const double* __restrict__ tp, int lane) {
  TargetAux a;
  if constexpr (!TRIG) return a;
  if (target == MM_TARGET_FUNNEL) {
    double s = 0.0;
    for (int i = 1 + lane; i < dim; i += 64) s += tp[i - 1] * q[i] * q[i];
    a.s0 = wave_sum(s);   // S = sum w x^2
    a.s1 = exp(-q[0]);    // e = exp(-v)
  }
  return a;
}

// grad_neg_log_dens element i.  For MM_TARGET_GAUSS_DENSE the row dot product reads P from global.
// TRIG = false compiles the torus target out: its atan2/sin/cos/log1p expansions need ~80 VGPRs,
// which the register-resident-metric kernels (Riemannian systems, never on a torus) cannot spare.
template <bool TRIG = true>
__device__ __forceinline__ double target_grad_elem(int target, const TargetAux& a, const double* q,
                                                   int i, int dim, const double* __restrict__ tp) {
  if constexpr (!TRIG) {
    if (target == MM_TARGET_TORUS || target == MM_TARGET_FUNNEL) return 0.0;
  }
  switch (target) {
    case MM_TARGET_GAUSS_ISO:
      return q[i];
    case MM_TARGET_GAUSS_DIAG:
      return tp[i] * q[i];
    case MM_TARGET_GAUSS_DENSE: {
      double s = 0.0;
      const double* row = tp + (int64_t)i * dim;
      for (int j = 0; j < dim; ++j) s += row[j] * q[j];
      return s;
    }
    case MM_TARGET_POLY:
      return tp[0] * q[i] + tp[1] * (q[i] * q[i] * q[i]);
    case MM_TARGET_BANANA: {
      double g = -(1.0 - q[i]) / 10.0;
      if (i > 0) g += 2.0 * (q[i] - q[i - 1] * q[i - 1]);
      if (i < dim - 1) g -= 4.0 * q[i] * (q[i + 1] - q[i] * q[i]);
      return g;
    }
#if defined(MM_RTC_BUILD) && defined(MM_RTC_USER_TARGET)
    case MM_TARGET_USER:
      return mm_user_grad(q, i, dim, tp);
#endif
    case MM_TARGET_FUNNEL: if constexpr (TRIG) {
      if (i == 0) return q[0] / 9.0 + 0.5 * (dim - 1) - 0.5 * a.s1 * a.s0;
      return a.s1 * tp[i - 1] * q[i];
    } else { return 0.0; }
    case MM_TARGET_TORUS: if constexpr (TRIG) {
      // The density is written in the angles theta = atan2(y, x), phi = atan2(z, rho - R); its gradient only
      // needs sin / cos of phi and of 4 theta, which are algebraic in (x, y, z): no atan2 / sin / cos in the
      // constrained kernel's per-step critical path (they were a third of its instructions).
      const double R = tp[0], r = tp[1], al = tp[2];
      const double x = q[0], y = q[1], z = q[2];
      const double rho2 = x * x + y * y;
      double rho, irho;
      sqrt_rsqrt(rho2, &rho, &irho);
      const double ct = x * irho, st = y * irho;            // cos theta, sin theta
      const double ct2 = ct * ct, st2 = st * st;
      const double s4 = 4.0 * st * ct * (ct2 - st2);        // sin 4 theta
      const double c4 = 1.0 - 8.0 * ct2 * st2;              // cos 4 theta
      const double u = rho - R;
      double w_, iw;
      sqrt_rsqrt(u * u + z * z, &w_, &iw);
      const double sp = z * iw, cp = u * iw;                // sin phi, cos phi
      const double r_over_R = r * rcp_nr(R);               // (two parameters: not worth an IEEE division per gradient)
      const double d1 = 1.0 + r_over_R * cp, d2 = 1.0 + al * s4 * cp;
      const double id2 = rcp_nr(d2);
      const double dl_dphi = -r_over_R * sp * rcp_nr(d1) + al * s4 * sp * id2;
      const double dl_dth = -4.0 * al * c4 * cp * id2;
      const double iw2 = iw * iw, irho2 = irho * irho;
      const double dphi_drho = -z * iw2, dphi_dz = u * iw2;
      if (i == 0) return dl_dth * (-y * irho2) + dl_dphi * dphi_drho * ct;
      if (i == 1) return dl_dth * (x * irho2) + dl_dphi * dphi_drho * st;
      return dl_dphi * dphi_dz;
    } else { return 0.0; }
    default:
      return 0.0;
  }
}

// neg_log_dens = wave_sum over i of this per-element term.
template <bool TRIG = true>
__device__ __forceinline__ double target_nld_elem(int target, const TargetAux& a, const double* q,
                                                  int i, int dim, const double* __restrict__ tp) {
  if constexpr (!TRIG) {
    if (target == MM_TARGET_TORUS || target == MM_TARGET_FUNNEL) return 0.0;
  }
  switch (target) {
    case MM_TARGET_GAUSS_ISO:
      return 0.5 * q[i] * q[i];
    case MM_TARGET_GAUSS_DIAG:
      return 0.5 * tp[i] * q[i] * q[i];
    case MM_TARGET_GAUSS_DENSE:
      return 0.5 * q[i] * target_grad_elem<TRIG>(target, a, q, i, dim, tp);
    case MM_TARGET_POLY: {
      const double q2 = q[i] * q[i];
      return 0.5 * tp[0] * q2 + 0.25 * tp[1] * q2 * q2;
    }
    case MM_TARGET_BANANA: {
      double v = (1.0 - q[i]) * (1.0 - q[i]) / 20.0;
      if (i < dim - 1) {
        const double r = q[i + 1] - q[i] * q[i];
        v += r * r;
      }
      return v;
    }
#if defined(MM_RTC_BUILD) && defined(MM_RTC_USER_TARGET)
    case MM_TARGET_USER:
      return mm_user_nld_term(q, i, dim, tp);
#endif
    case MM_TARGET_FUNNEL:
      return i == 0 ? q[0] * q[0] / 18.0 + 0.5 * (dim - 1) * q[0] + 0.5 * a.s1 * a.s0 : 0.0;
    case MM_TARGET_TORUS: if constexpr (TRIG) {
      if (i != 0) return 0.0;
      const double R = tp[0], r = tp[1], al = tp[2];
      const double rho = sqrt(q[0] * q[0] + q[1] * q[1]);
      const double theta = atan2(q[1], q[0]), phi = atan2(q[2], rho - R);
      return log1p(r * cos(phi) / R) - log1p(sin(4.0 * theta) * cos(phi) * al);
    } else { return 0.0; }
    default:
      return 0.0;
  }
}

}  // namespace mmdev
