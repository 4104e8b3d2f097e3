/*
 * rt_math.h — the fp32 arithmetic contract of the ray-trace path.
 *
 * GLSL leaves the precision of normalize/sin/cos/acos/pow implementation-defined, and a
 * 1-ulp difference in a ray direction flips which voxel a grazing ray enters (an O(1) pixel
 * error).  "Same result as the reference on the same inputs" is therefore only well-defined
 * once the elementary functions are pinned.  This header pins them: every function below is
 * built from IEEE-754 binary32 +, -, *, /, sqrt, floor, EXPLICIT fused multiply-add (rtm_fma)
 * and integer bit operations only — all of which are correctly rounded both on x86-64 (SSE/FMA3)
 * and on gfx950 — so any conforming compilation (g++ -mfma or hipcc, with IMPLICIT contraction
 * OFF: -ffp-contract=off) yields the same bits.  (GLSL permits a*b+c to be fused unless marked
 * `precise`; where this contract fuses, it says so explicitly.)
 *
 * It is part of the ABI (like rt_abi.h), not of the oracle: the CPU oracle under oracle/
 * restates the shader's ALGORITHM independently and only shares these definitions of the
 * elementary operations.  tests/test_math_contract.py checks each function against libm.
 *
 * Rules for users: compile with -ffp-contract=off (and -mfma on x86 so rtm_fma is one instruction;
 * without it libm's fmaf gives the same result, slowly); do not use -ffast-math; on hipcc keep the
 * default -fhip-fp32-correctly-rounded-divide-sqrt and do not flush denormals.
 *
 * Polynomial coefficients are the classic single-precision minimax sets (Cephes sinf/cosf/
 * asinf/logf/exp2f, S. Moshier, public domain).
 */
#ifndef RT_MATH_H
#define RT_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define RTM_HD __host__ __device__ static inline __attribute__((always_inline))
#else
#define RTM_HD static inline __attribute__((always_inline))
#endif

typedef struct rtm_vec3 { float x, y, z; } rtm_vec3;

#define RTM_PI      3.14159265358979323846f   /* raytrace.comp:60 (rounded to binary32) */
#define RTM_PIO2    1.57079632679489661923f

RTM_HD uint32_t rtm_bits(float f) { return __builtin_bit_cast(uint32_t, f); }
RTM_HD float rtm_from_bits(uint32_t u) { return __builtin_bit_cast(float, u); }

RTM_HD float rtm_floor(float x) { return __builtin_floorf(x); }
/* fused multiply-add, one rounding: a*b + c */
RTM_HD float rtm_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
RTM_HD float rtm_sqrt(float x) { return __builtin_sqrtf(x); }
RTM_HD float rtm_abs(float x) { return __builtin_fabsf(x); }
/* GLSL min/max: min(x,y) = y < x ? y : x ; max(x,y) = x < y ? y : x */
RTM_HD float rtm_min(float x, float y) { return y < x ? y : x; }
RTM_HD float rtm_max(float x, float y) { return x < y ? y : x; }
RTM_HD float rtm_clamp(float x, float lo, float hi) { return rtm_min(rtm_max(x, lo), hi); }
/* GLSL mix(x,y,a) = x*(1-a) + y*a (second product fused into the sum) */
RTM_HD float rtm_mix(float x, float y, float a) { return rtm_fma(y, a, x * (1.0f - a)); }
/* GLSL mod(x,y) = x - y*floor(x/y) */
RTM_HD float rtm_mod(float x, float y) { return x - y * rtm_floor(x / y); }

RTM_HD float rtm_dot3(rtm_vec3 a, rtm_vec3 b) { return rtm_fma(a.z, b.z, rtm_fma(a.y, b.y, a.x * b.x)); }
RTM_HD float rtm_length3(rtm_vec3 v) { return rtm_sqrt(rtm_fma(v.z, v.z, rtm_fma(v.y, v.y, v.x * v.x))); }
RTM_HD float rtm_length2(float x, float y) { return rtm_sqrt(rtm_fma(y, y, x * x)); }
/* normalize(v) = v * (1/length(v)) */
RTM_HD rtm_vec3 rtm_normalize3(rtm_vec3 v) {
    float r = 1.0f / rtm_length3(v);
    rtm_vec3 o = { v.x * r, v.y * r, v.z * r };
    return o;
}

/* ---- sin / cos ------------------------------------------------------------------------- */
/* Valid for |x| < 1e5 (the shader's arguments are 2*pi*[0,1] and the sun angle); outside that
 * range the result is defined as sin = 0, cos = 1. */
RTM_HD void rtm_sincos(float x, float* s_out, float* c_out) {
    if (!(rtm_abs(x) < 1.0e5f)) { *s_out = 0.0f; *c_out = 1.0f; return; }
    float k = rtm_floor(rtm_fma(x, 0.63661977236758134308f, 0.5f));
    /* three-part pi/2 (Cody-Waite) */
    float r = rtm_fma(-k, 7.54978995489188216e-8f, rtm_fma(-k, 4.837512969970703125e-4f, rtm_fma(-k, 1.5703125f, x)));
    int q = ((int)k) & 3;
    float z = r * r;
    float s = rtm_fma(rtm_fma(rtm_fma(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
    float c = rtm_fma(rtm_fma(rtm_fma(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f) * z, z,
                      rtm_fma(-0.5f, z, 1.0f));
    float ss, cc;
    if (q == 0)      { ss = s;  cc = c;  }
    else if (q == 1) { ss = c;  cc = -s; }
    else if (q == 2) { ss = -s; cc = -c; }
    else             { ss = -c; cc = s;  }
    *s_out = ss;
    *c_out = cc;
}
RTM_HD float rtm_sin(float x) { float s, c; rtm_sincos(x, &s, &c); return s; }
RTM_HD float rtm_cos(float x) { float s, c; rtm_sincos(x, &s, &c); return c; }

/* ---- acos ------------------------------------------------------------------------------ */
RTM_HD float rtm_asin_poly(float a) { /* |a| <= 0.5 */
    float z = a * a;
    float p = rtm_fma(rtm_fma(rtm_fma(rtm_fma(4.2163199048e-2f, z, 2.4181311049e-2f), z, 4.5470025998e-2f), z,
                              7.4953002686e-2f), z, 1.6666752422e-1f);
    return rtm_fma(a * z, p, a);
}
RTM_HD float rtm_acos(float x) {
    x = rtm_clamp(x, -1.0f, 1.0f);
    if (x < -0.5f) return rtm_fma(-2.0f, rtm_asin_poly(rtm_sqrt(0.5f * (1.0f + x))), RTM_PI);
    if (x > 0.5f)  return 2.0f * rtm_asin_poly(rtm_sqrt(0.5f * (1.0f - x)));
    return RTM_PIO2 - rtm_asin_poly(x);
}

/* ---- pow(x, y) for the sky model (raytrace.comp:278,280,283) ----------------------------- */
/* log2 of a positive normal float */
RTM_HD float rtm_log2_pos(float x) {
    uint32_t u = rtm_bits(x);
    int e = (int)(u >> 23) - 126;                          /* x = m * 2^e, m in [0.5, 1) */
    float m = rtm_from_bits((u & 0x007FFFFFu) | 0x3F000000u);
    if (m < 0.70710678118654752440f) { e -= 1; m = (m + m) - 1.0f; }
    else { m = m - 1.0f; }
    float z = m * m;
    float y = rtm_fma(7.0376836292e-2f, m, -1.1514610310e-1f);
    y = rtm_fma(y, m, 1.1676998740e-1f);
    y = rtm_fma(y, m, -1.2420140846e-1f);
    y = rtm_fma(y, m, 1.4249322787e-1f);
    y = rtm_fma(y, m, -1.6668057665e-1f);
    y = rtm_fma(y, m, 2.0000714765e-1f);
    y = rtm_fma(y, m, -2.4999993993e-1f);
    y = rtm_fma(y, m, 3.3333331174e-1f);
    y = rtm_fma(y * m, z, -0.5f * z);
    return rtm_fma(m + y, 1.44269504088896340736f, (float)e);
}
RTM_HD float rtm_exp2(float x) {
    if (!(x > -126.0f)) return 0.0f;      /* underflow (and NaN) -> 0 */
    if (x > 127.0f) x = 127.0f;
    float i = rtm_floor(x);
    float f = x - i;
    if (f > 0.5f) { i += 1.0f; f -= 1.0f; }
    float p = rtm_fma(1.535336188319500e-4f, f, 1.339887440266574e-3f);
    p = rtm_fma(p, f, 9.618437357674640e-3f);
    p = rtm_fma(p, f, 5.550332471162809e-2f);
    p = rtm_fma(p, f, 2.402264791363012e-1f);
    p = rtm_fma(p, f, 6.931472028550421e-1f);
    p = rtm_fma(p, f, 1.0f);
    int ii = (int)i;
    if (ii < -126) return 0.0f;
    if (ii > 127) ii = 127;
    return p * rtm_from_bits((uint32_t)(ii + 127) << 23);
}
/* pow(x,y), y > 0.  GLSL leaves x < 0 undefined; this contract defines pow(x,y) = 0 for
 * x < FLT_MIN (zero, denormal, negative or NaN) — documented as quirk Q11 in DESIGN.md. */
RTM_HD float rtm_pow(float x, float y) {
    if (!(x >= 1.17549435e-38f)) return 0.0f;
    return rtm_exp2(y * rtm_log2_pos(x));
}

/* ---- conversions used by the G-buffer stores (raytrace.comp:352-385) ---------------------- */
/* UNORM store: NaN -> 0, clamp to [0,1], round half up. */
RTM_HD uint32_t rtm_unorm(float x, float maxv) {
    if (!(x > 0.0f)) return 0u;
    if (x > 1.0f) x = 1.0f;
    return (uint32_t)rtm_floor(x * maxv + 0.5f);
}
/* uint(f) stored to R16_UINT: NaN/negative -> 0, saturate at 0xFFFF, truncate. */
RTM_HD uint32_t rtm_f2u16(float x) {
    if (!(x > 0.0f)) return 0u;
    if (x >= 65535.0f) return 65535u;
    return (uint32_t)x;
}

#endif /* RT_MATH_H */
