/*
 * Portable single-precision log / cos / exp2 / pow built only from IEEE-754
 * binary32 +, -, *, /, floor and integer bit operations.  Compiled with FMA
 * contraction disabled, every function returns the same bits on the host
 * (gcc/clang, x86-64 SSE) and on gfx950, which is what the renderer's
 * "exact math" mode relies on: with it, whole radiance buffers can be compared
 * bit for bit between the HIP kernels and a CPU restatement that includes this
 * same header.
 *
 * Domain notes (the shader only needs these ranges, shader.wgsl:361-366,393-395):
 *   pm_logf : x in [0, +inf]; 0 -> -inf, negative/NaN -> NaN.  < 1 ulp typical.
 *   pm_cosf : |x| < 2^15 (three-constant Cody-Waite reduction); ~1 ulp for |x| < 100.
 *   pm_exp2f: any x; results below 2^-126 are rounded once into the subnormals.
 *   pm_powf : x >= 0 only, computed as exp2(y*log2(x)) like GPU shader compilers do.
 *
 * Plain C99 so that it can be included from C, C++ and HIP sources alike.
 */
#ifndef PT_PORTABLE_MATH_H
#define PT_PORTABLE_MATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define PM_FN static __host__ __device__ __forceinline__
#else
#define PM_FN static inline
#endif

PM_FN uint32_t pm_bits(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }
PM_FN float pm_float(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }

/* Splits a positive finite x into m in [sqrt(1/2), sqrt(2)) and exponent e with
 * x = m * 2^e.  Returns f = m - 1 and writes e. */
PM_FN float pm_split_log(float x, int32_t *e_out) {
    uint32_t ix = pm_bits(x);
    int32_t e = 0;
    if (ix < 0x00800000u) {            /* subnormal: scale up by 2^25 (exact) */
        x = x * 33554432.0f;
        ix = pm_bits(x);
        e = -25;
    }
    ix += 0x3f800000u - 0x3f3504f3u;   /* put the mantissa split at sqrt(2)/2 */
    e += (int32_t)(ix >> 23) - 127;
    ix = (ix & 0x007fffffu) + 0x3f3504f3u;
    *e_out = e;
    return pm_float(ix) - 1.0f;
}

/* log(1+f) - f + f*f/2 ... returned as the pair used by both logf and log2f:
 * result = f - (hfsq - s*(hfsq+R)) */
PM_FN float pm_log1p_core(float f, float *hfsq_out, float *s_out, float *R_out) {
    const float Lg1 = 0.66666662693f, Lg2 = 0.40000972152f, Lg3 = 0.28498786688f, Lg4 = 0.24279078841f;
    float s = f / (2.0f + f);
    float z = s * s;
    float w = z * z;
    float t1 = w * (Lg2 + w * Lg4);
    float t2 = z * (Lg1 + w * Lg3);
    float R = t2 + t1;
    float hfsq = 0.5f * f * f;
    *hfsq_out = hfsq; *s_out = s; *R_out = R;
    return f - (hfsq - s * (hfsq + R));
}

PM_FN float pm_logf(float x) {
    uint32_t ix = pm_bits(x);
    if ((ix << 1) == 0u) return pm_float(0xff800000u);          /* log(+-0) = -inf */
    if (ix >> 31) return pm_float(0x7fc00000u);                 /* log(<0) = NaN */
    if (ix >= 0x7f800000u) return x;                            /* +inf or NaN */
    if (ix == 0x3f800000u) return 0.0f;
    const float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f;
    int32_t e;
    float f = pm_split_log(x, &e);
    float hfsq, s, R;
    (void)pm_log1p_core(f, &hfsq, &s, &R);
    float dk = (float)e;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

PM_FN float pm_log2f(float x) {
    uint32_t ix = pm_bits(x);
    if ((ix << 1) == 0u) return pm_float(0xff800000u);
    if (ix >> 31) return pm_float(0x7fc00000u);
    if (ix >= 0x7f800000u) return x;
    if (ix == 0x3f800000u) return 0.0f;
    int32_t e;
    float f = pm_split_log(x, &e);
    float hfsq, s, R;
    float lnm = pm_log1p_core(f, &hfsq, &s, &R);
    /* 1/ln2 split into a short head (exact product with the head of lnm is not
     * needed at this accuracy: |lnm| <= 0.35 so the absolute error stays ~1e-7) */
    const float invln2_hi = 1.4426879883f;   /* 0x3fb8a000: 11 significant bits */
    const float invln2_lo = 7.0525690e-06f;  /* 1/ln2 - invln2_hi */
    return (float)e + (lnm * invln2_hi + lnm * invln2_lo);
}

PM_FN float pm_floorf(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_floorf(x);
#else
    return __builtin_floorf(x);
#endif
}

PM_FN float pm_cosf(float x) {
    uint32_t ax = pm_bits(x) & 0x7fffffffu;
    if (ax >= 0x7f800000u) return pm_float(0x7fc00000u);  /* inf / NaN */
    float a = pm_float(ax);                               /* cos is even */
    /* k = nearest integer to a * 2/pi ; r = a - k*pi/2 in three exact-ish steps */
    float kf = pm_floorf(a * 0.63661977237f + 0.5f);
    const float P1 = 1.5703125f;                 /* 0x3fc90000 */
    const float P2 = 4.837512969970703125e-4f;   /* next bits of pi/2 */
    const float P3 = 7.54978995489188216e-8f;
    float r = ((a - kf * P1) - kf * P2) - kf * P3;
    int32_t k = (int32_t)kf;
    float z = r * r;
    float sp = r + r * z * (-1.6666654611e-1f + z * (8.3321608736e-3f + z * (-1.9515295891e-4f)));
    float cp = (1.0f - 0.5f * z) + z * z * (4.166664568298827e-2f + z * (-1.388731625493765e-3f + z * 2.443315711809948e-5f));
    switch (k & 3) {
        case 0: return cp;
        case 1: return -sp;
        case 2: return -cp;
        default: return sp;
    }
}

PM_FN float pm_exp2f(float x) {
    uint32_t ix = pm_bits(x);
    if ((ix & 0x7fffffffu) > 0x7f800000u) return x;       /* NaN */
    if (x >= 128.0f) return pm_float(0x7f800000u);
    if (x < -150.0f) return 0.0f;
    float nf = pm_floorf(x + 0.5f);
    float r = x - nf;                                     /* exact, |r| <= 0.5 */
    /* 2^r = exp(r ln2), degree-6 polynomial */
    float t = r * 0.69314718056f;
    float p = 1.0f + t * (1.0f + t * (0.5f + t * (1.6666667163e-1f + t * (4.1666667908e-2f +
              t * (8.3333337680e-3f + t * 1.3888889225e-3f)))));
    int32_t n = (int32_t)nf;
    if (n >= -126) {
        if (n > 127) { p = p * 2.0f; n -= 1; }
        return p * pm_float((uint32_t)(n + 127) << 23);
    }
    /* subnormal result: scale in two exact steps, one rounding at the end */
    float big = p * pm_float((uint32_t)(n + 64 + 127) << 23);
    return big * pm_float((uint32_t)(127 - 64) << 23);
}

PM_FN float pm_powf(float x, float y) {
    return pm_exp2f(y * pm_log2f(x));
}

#endif /* PT_PORTABLE_MATH_H */
