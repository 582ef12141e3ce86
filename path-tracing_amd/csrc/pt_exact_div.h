/*
 * Correctly rounded binary32 division a / b from a precomputed reciprocal.
 *
 * ray_box (shader.wgsl:241-242) divides six plane offsets by the three ray
 * direction components for every box, and the closest-hit contract of this
 * backend is bit-exactness with that IEEE division.  A ray tests ~150 boxes per
 * cast against the SAME three denominators, so the reciprocal y = RN(1/b) is
 * computed once per ray (one true division) and each quotient is then formed
 * with one multiply and four fused multiply-adds (pt_div_by_reciprocal; the kernels
 * use the shorter pt_div_by_reciprocal2 further down, which needs a second word per
 * reciprocal):
 *
 *     q0 = RN(a*y)            r0 = a - b*q0   (exact in an FMA)
 *     q1 = RN(q0 + r0*y)      r1 = a - b*q1   (exact)
 *     q  = RN(q1 + r1*y)
 *
 * Why q == RN(a/b) (x = a/b, b scaled into [1,2), x in [2^e, 2^(e+1))):
 *   y = (1/b)(1+e1) with |e1| <= b*2^-25, because 1/b lies in (1/2,1] where the
 *   half-ulp is 2^-25.  q1 = RN(x + (x-q0)*e1) is x rounded after a perturbation
 *   of at most a few 2^-47*2^e, so |x - q1| <= (1/2 + 2^-21) ulp.  Then r1 is a
 *   multiple of ulp(b)*ulp(q1) below 2^24 such units, hence exact, and
 *   q = RN(x + (x-q1)*e1) with |(x-q1)*e1| <= b*2^-49*2^e*(1+2^-21).  A quotient
 *   of two 24-bit significands differs from every rounding boundary m (a 25-bit
 *   midpoint) by |a - m*b|/b >= 2^-47*2^e/b.  The perturbation is smaller than
 *   that distance whenever b*b*(1+2^-21) < 4, i.e. for every significand except
 *   the topmost few, which pt_div_safe() excludes together with everything that
 *   could overflow, underflow or produce NaN along the way.  Rays that fail the
 *   predicate use the plain division, so results never depend on this file.
 *
 * The equivalence is also checked by brute force on the CPU
 * (tests/test_exact_division.py: 10^8 random cases plus adversarial near-midpoint
 * cases) and, implicitly, by every bit-exact GPU-vs-oracle test.
 */
#ifndef PT_EXACT_DIV_H
#define PT_EXACT_DIV_H

#include <stdint.h>

#if defined(__HIPCC__)
#define PTD_FN static __host__ __device__ __forceinline__
#else
#define PTD_FN static inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define PTD_FMA(a, b, c) __builtin_fmaf((a), (b), (c))
#else
#include <math.h>
#define PTD_FMA(a, b, c) fmaf((a), (b), (c))
#endif

PTD_FN uint32_t ptd_bits(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }

/* a / b for b that satisfies pt_div_safe_denominator and a that is 0 or has
 * 2^-100 <= |a| <= 2^100;  y must be 1.0f / b (correctly rounded). */
PTD_FN float pt_div_by_reciprocal(float a, float b, float y) {
    float q = a * y;
    float r = PTD_FMA(-b, q, a);
    q = PTD_FMA(r, y, q);
    r = PTD_FMA(-b, q, a);
    return PTD_FMA(r, y, q);
}

/*
 * The same quotient with one operation less, from a reciprocal kept in two parts.
 *
 *     yl = RN((1 - b*y) * y)      with 1 - b*y exact in one FMA: y + yl = (1/b)(1 + e2), |e2| <= 2^-47
 *     t  = RN(a*yl)               q0 = RN(a*y + t)   (one FMA)
 *     r  = a - b*q0   (exact)     q  = RN(q0 + r*y)
 *
 * Why q == RN(a/b) (same scaling as above): a*y + t = x(1 + h) with |h| <= 2^-47 + 2^-48 (the error of y + yl plus
 * the rounding of t, |t| <= |x|*2^-24*(1 + 2^-23)), so |x - q0| <= (1/2 + 2^-22) ulp; r is then a multiple of
 * ulp(b)*ulp(q0) below 2^24 such units, hence exact, and q = RN(x + (x - q0)*e1) with the same bound on the
 * perturbation as in the five-operation form: b*2^-49*2^e*(1 + 2^-21), below the distance 2^-47*2^e/b of a
 * quotient to the nearest rounding boundary whenever b*b*(1 + 2^-21) < 4.  t must not lose more than 2^-48 |x| to
 * underflow: |a| >= 2^-80 (then |x| >= 2^-100 and 2^-150 <= 2^-48 |x|), which pt_div_tame_coordinate grants.
 * Checked by the same brute force (tests/native/div_check.c).
 */
PTD_FN float pt_reciprocal_low(float b, float y) {
    float e = PTD_FMA(-b, y, 1.0f);
    return e * y;
}
PTD_FN float pt_div_by_reciprocal2(float a, float b, float y, float yl) {
    float t = a * yl;
    float q = PTD_FMA(a, y, t);
    float r = PTD_FMA(-b, q, a);
    return PTD_FMA(r, y, q);
}

/* denominator: exponent in [-20, 20], significand below 0x7fff00 */
PTD_FN int pt_div_safe_denominator(float b) {
    uint32_t u = ptd_bits(b);
    uint32_t e = (u >> 23) & 0xffu, m = u & 0x7fffffu;
    return e >= 127u - 20u && e <= 127u + 20u && m < 0x7fff00u;
}

/* A coordinate c (ray origin component or box bound) is "tame" when it is 0 or
 * 2^-56 <= |c| <= 2^90.  The difference of two tame coordinates is 0 or at least
 * 2^-80 in magnitude (it is a multiple of the smaller operand's ulp, 2^-79 at
 * worst) and below 2^91, which is what pt_div_by_reciprocal2 needs (pt_div_by_reciprocal
 * makes do with 2^-100). */
PTD_FN int pt_div_tame_coordinate(float c) {
    uint32_t u = ptd_bits(c) & 0x7fffffffu;
    return u == 0u || (u >= ((127u - 56u) << 23) && u <= ((127u + 90u) << 23));
}

#endif /* PT_EXACT_DIV_H */
