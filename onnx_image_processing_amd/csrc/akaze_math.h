// Exact-rounding arithmetic helpers of the fused AKAZE scale kernel (csrc/akaze.hip); also included by the debug
// library's exhaustive check (csrc/hooks.hip: mi_debug_akaze_math_check).
#pragma once
#include <hip/hip_runtime.h>

// Correctly rounded fp32 sqrt and division for the operand ranges of the diffusion step (normal, finite, far from
// overflow: |g| <= ~255, kappa and 1 + q^2 >= 1e-4), i.e. the compiler's IEEE expansions without their denormal
// scaling and special-case fix-ups: v_sqrt_f32 / v_rcp_f32 (1 ulp) + exact fma residuals.  Same results as
// sqrtf() and operator/ on these ranges (asserted against the per-step kernels, which use those, bit for bit).
__device__ __forceinline__ float ak_sqrt(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
  const float lo = __uint_as_float(__float_as_uint(s) - 1u), hi = __uint_as_float(__float_as_uint(s) + 1u);
  const float rl = __builtin_fmaf(-lo, s, x), rh = __builtin_fmaf(-hi, s, x);
  float r = (rl <= 0.0f) ? lo : s;
  r = (rh > 0.0f) ? hi : r;
  return r;
}
// The same square root on the fp32 pipe only.  fp32 multiply / add / fma issue a wave in 2 cycles on gfx950, integer
// adds, compares and selects in 4, transcendentals in 16: ak_sqrt's fix-up (two integer adds, two compares, two selects)
// costs more than its v_sqrt_f32.  Here: y = v_rsq_f32(x) (1 ulp), g = x y, h = y / 2, then two residual corrections
// g <- g + (x - g g) h with the residual exact in one fma.  Candidates `steps` = 1, 2 are compared with sqrtf() over every
// float of the operand range by the exhaustive test; the kernel uses the cheapest one that is exact.
template <int STEPS>
__device__ __forceinline__ float ak_sqrt_fp(float x) {
  const float y = __builtin_amdgcn_rsqf(x);
  float g = x * y;
  const float h = 0.5f * y;
#pragma unroll
  for (int i = 0; i < STEPS; ++i) g = __builtin_fmaf(__builtin_fmaf(-g, g, x), h, g);
  return g;
}
__device__ __forceinline__ float ak_div(float a, float b) {
  float y = __builtin_amdgcn_rcpf(b);
  const float e = __builtin_fmaf(-b, y, 1.0f);
  y = __builtin_fmaf(e, y, y);                       // reciprocal to < 1 ulp
  float q = a * y;
  float r = __builtin_fmaf(-b, q, a);                // exact residual
  q = __builtin_fmaf(r, y, q);
  r = __builtin_fmaf(-b, q, a);                      // second correction: the compiler's own sequence
  return __builtin_fmaf(r, y, q);                    // (v_div_fmas without the scaling)
}
// a / b for a divisor that does not change (kappa): rb = the CORRECTLY ROUNDED reciprocal of b, computed once per
// thread by the IEEE division 1.0f / b.  Then q0 = RN(a rb), r = a - b q0 (exact in one fma), q = RN(q0 + r rb) is
// the correctly rounded quotient (Markstein 1990; Cornea / Harrison / Tang, "Scientific Computing on Itanium-based
// Systems", thm. 8.3: no exceptional cases when rb = RN(1 / b)) -- three instructions instead of eight.
__device__ __forceinline__ float ak_div_by(float a, float b, float rb) {
  const float q0 = a * rb;
  const float r = __builtin_fmaf(-b, q0, a);
  return __builtin_fmaf(r, rb, q0);
}
// 1 / d, correctly rounded, for d in the diffusion step's range (1 <= d = 1 + q^2 < 2^40): v_rcp_f32 (1 ulp), one
// Newton step (y1 within 1 ulp of 1 / d) and Markstein's final correction y = RN(y1 + (1 - d y1) y1), exact unless
// d's significand is all ones -- and 1 + q^2 with q^2 >= 0 rounded to 24 bits never is for q^2 < 2^24, while beyond
// that d = q^2 and the test below covers it.  tests/test_gpu_parity.py::test_akaze_fast_division_is_exact compares both
// helpers with the IEEE operators over EVERY float of the operand ranges.  Five instructions instead of eight.
__device__ __forceinline__ float ak_rcp(float d) {
  const float y0 = __builtin_amdgcn_rcpf(d);
  const float y1 = __builtin_fmaf(__builtin_fmaf(-d, y0, 1.0f), y0, y0);
  return __builtin_fmaf(__builtin_fmaf(-d, y1, 1.0f), y1, y1);
}

