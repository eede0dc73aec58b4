/* var_math.h — scalar math shared, source-identical, by the CPU oracle (gcc) and the HIP kernels (hipcc).
 *
 * Why it exists: the sampling loop feeds its own token choices back into itself (reference
 * models/var.py:160-187), so the GPU path and the CPU oracle can only be compared token-for-token if
 * every floating-point value is produced by the same sequence of correctly-rounded IEEE-754 binary32
 * operations on both sides.  +,-,*,/,sqrt and fma are correctly rounded on x86 (SSE/FMA3) and on gfx950
 * (hipcc default: IEEE divide/sqrt, f32 denormals on); libm/ocml transcendentals are not bit-compatible,
 * so exp (and what is built on it) is defined here from those primitive operations only.
 *
 * Both compilers must be run with -ffp-contract=off: every fused multiply-add below is explicit.
 */
#ifndef VAR_MATH_H
#define VAR_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define VM_FN __host__ __device__ static inline
#else
#define VM_FN static inline
#endif

VM_FN float vm_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
VM_FN float vm_sqrt(float a) { return __builtin_sqrtf(a); }
VM_FN float vm_max(float a, float b) { return a > b ? a : b; }
VM_FN float vm_min(float a, float b) { return a < b ? a : b; }

/* e^x, relative error ~1 ulp (Cephes expf polynomial), exactly 0 for x <= -87 (so no subnormal results), NaN -> NaN,
 * clamped at x = 88 (finite).  Used by: attention softmax, sampler softmax, SiLU, GELU(tanh), scale_mul.exp(). */
VM_FN float vm_exp(float x) {
    if (x != x) return x;
    if (!(x > -87.0f)) return 0.0f;
    if (x > 88.0f) x = 88.0f;
    float n = __builtin_rintf(x * 1.44269504088896341f);
    float r = vm_fma(n, -0.693145751953125f, x);
    r = vm_fma(n, -1.42860682030941723212e-6f, r);
    float p = 1.9875691500e-4f;
    p = vm_fma(p, r, 1.3981999507e-3f);
    p = vm_fma(p, r, 8.3334519073e-3f);
    p = vm_fma(p, r, 4.1665795894e-2f);
    p = vm_fma(p, r, 1.6666665459e-1f);
    p = vm_fma(p, r, 5.0000001201e-1f);
    float y = vm_fma(p, r * r, r) + 1.0f;
    union { float f; uint32_t u; } s;
    s.u = (uint32_t)((int32_t)n + 127) << 23;
    return y * s.f;
}

/* e^x for x <= 0 — the softmax numerators of the attention kernel, which are taken relative to the running maximum.
 * Same reduction and polynomial as vm_exp, shaped so that a 64-lane wave spends as few vector instructions on it as possible
 * (they are paid in matrix-pipe time next to fp32 MFMAs):
 *   - x is clamped below at -87: the result is never 0 or subnormal (>= 1.6e-38), so the final scaling by 2^n is an integer
 *     add into the exponent field (no ldexp, no select);
 *   - n = round(x * log2 e) comes out of ONE fma against the 1.5*2^23 constant (its low mantissa bits are n), instead of
 *     multiply + rint + float->int conversion.
 * NaN -> the clamp returns -87's value; the callers never produce NaN (scores are finite or -inf). */
VM_FN float vm_exp_le0(float x) {
    x = vm_max(x, -87.0f);
    const float t = vm_fma(x, 1.44269504088896341f, 12582912.0f);
    const float n = t - 12582912.0f;
    float r = vm_fma(n, -0.693145751953125f, x);
    r = vm_fma(n, -1.42860682030941723212e-6f, r);
    float p = 1.9875691500e-4f;
    p = vm_fma(p, r, 1.3981999507e-3f);
    p = vm_fma(p, r, 8.3334519073e-3f);
    p = vm_fma(p, r, 4.1665795894e-2f);
    p = vm_fma(p, r, 1.6666665459e-1f);
    p = vm_fma(p, r, 5.0000001201e-1f);
    union { float f; uint32_t u; } y, tt;
    y.f = vm_fma(p, r * r, r) + 1.0f;
    tt.f = t;
    y.u = y.u + (tt.u << 23);          /* 12582912.0f == 0x4B400000: (bits << 23) keeps exactly n << 23 (mod 2^32) */
    return y.f;
}

/* ln(x) for x > 0, relative error ~1 ulp (Cephes logf), from integer exponent extraction + fma polynomial only; x <= 0 or
 * subnormal -> -inf, NaN -> NaN, +inf -> +inf.  Used by the gumbel noise -log(Exp(1)) of the more_smooth path (helpers.py:26). */
VM_FN float vm_log(float x) {
    union { float f; uint32_t u; } c;
    c.f = x;
    if (x != x) return x;
    if (!(x >= 1.17549435e-38f)) { c.u = 0xff800000u; return c.f; }
    if (c.u == 0x7f800000u) return x;
    float fe = (float)((int32_t)(c.u >> 23) - 126);
    c.u = (c.u & 0x007fffffu) | 0x3f000000u;            /* mantissa in [0.5, 1) */
    float m = c.f;
    if (m < 0.707106781186547524f) { fe = fe - 1.0f; m = (m + m) - 1.0f; } else { m = m - 1.0f; }
    const float z = m * m;
    float p = 7.0376836292e-2f;
    p = vm_fma(p, m, -1.1514610310e-1f);
    p = vm_fma(p, m, 1.1676998740e-1f);
    p = vm_fma(p, m, -1.2420140846e-1f);
    p = vm_fma(p, m, 1.4249322787e-1f);
    p = vm_fma(p, m, -1.6668057665e-1f);
    p = vm_fma(p, m, 2.0000714765e-1f);
    p = vm_fma(p, m, -2.4999993993e-1f);
    p = vm_fma(p, m, 3.3333331174e-1f);
    float y = (p * m) * z;
    y = vm_fma(-2.12194440e-4f, fe, y);
    y = vm_fma(-0.5f, z, y);
    return vm_fma(0.693359375f, fe, m + y);
}

/* x * sigmoid(x)  (reference: nn.SiLU, basic_var.py:147,170; basic_vae.py:14-15) */
VM_FN float vm_silu(float x) { return x / (1.0f + vm_exp(-x)); }

/* GELU, tanh approximation (reference: nn.GELU(approximate='tanh'), basic_var.py:40):
 * 0.5 x (1 + tanh(u)) == x * sigmoid(2u),  u = sqrt(2/pi) (x + 0.044715 x^3). */
VM_FN float vm_gelu_tanh(float x) {
    float x3 = (x * x) * x;
    float u = 0.7978845608028654f * (x + 0.044715f * x3);
    return x / (1.0f + vm_exp(-2.0f * u));
}

/* monotone map float -> uint32 (ascending order preserved, -0 == +0), used by the sampler's top-k select and sort */
VM_FN uint32_t vm_float_key(float f) {
    union { float f; uint32_t u; } c;
    c.f = f;
    if ((c.u << 1) == 0u) c.u = 0u;                 /* -0 sorts equal to +0, as a float compare would have it */
    return (c.u & 0x80000000u) ? ~c.u : (c.u | 0x80000000u);
}

#endif /* VAR_MATH_H */
