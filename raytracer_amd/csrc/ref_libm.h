// atan2f and asinf exactly as the reference's C library computes them.
//
// The reference maps a sphere hit to texture coordinates with atan2 and asin on floats (sphere.cpp:150-153), i.e.
// with glibc's single-precision routines, which are not correctly rounded: another libm (the device's) returns a
// neighbouring float for a fraction of the arguments, and when u * (width - 1) sits on a texel boundary the pixel
// takes its colour from the neighbouring texel.  These are restatements of the algorithms glibc 2.35 uses
// (sysdeps/ieee754/flt-32/e_atan2f.c, s_atanf.c, e_asinf.c: the fdlibm argument reduction and polynomials, asinf
// with glibc's own five-term polynomial, atanf with its 2^25 large-argument cut-off) in plain float arithmetic,
// which is deterministic without contraction.  tests/test_ref_libm.py compiles them for the host and compares them
// with the C library bit for bit (asinf over every float in [-1, 1], atanf over every non-negative float, atan2f
// over hundreds of millions of pairs when run exhaustively; a sample in the regular suite).
#ifndef RT_REF_LIBM_H
#define RT_REF_LIBM_H

#include <stdint.h>

#ifndef RL_FN
#define RL_FN static inline
#endif

RL_FN int32_t rl_bits(float f) { return __builtin_bit_cast(int32_t, f); }
RL_FN float rl_float(int32_t i) { return __builtin_bit_cast(float, i); }
RL_FN float rl_fabsf(float x) { return rl_float(rl_bits(x) & 0x7fffffff); }

RL_FN float rl_atanf(float x) {
    const float atanhi[4] = { 4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f };
    const float atanlo[4] = { 5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f };
    const float aT[11] = { 3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f, 9.0908870101e-02f, -7.6918758452e-02f,
                           6.6610731184e-02f, -5.8335702866e-02f, 4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f };
    const float one = 1.0f, huge = 1.0e30f;
    float w, s1, s2, z;
    int32_t ix, hx, id;
    hx = rl_bits(x); ix = hx & 0x7fffffff;
    if (ix >= 0x4c000000) {                                 // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;
        return (hx > 0) ? atanhi[3] + atanlo[3] : -atanhi[3] - atanlo[3];
    }
    if (ix < 0x3ee00000) {                                  // |x| < 0.4375
        if (ix < 0x31000000) { if (huge + x > one) return x; }
        id = -1;
    } else {
        x = rl_fabsf(x);
        if (ix < 0x3f980000) {
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - one) / (2.0f + x); }
            else { id = 1; x = (x - one) / (x + one); }
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (one + 1.5f * x); }
            else { id = 3; x = -1.0f / x; }
        }
    }
    z = x * x; w = z * z;
    s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return x - x * (s1 + s2);
    z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
    return (hx < 0) ? -z : z;
}

RL_FN float rl_atan2f(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = 7.8539818525e-01f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    float z;
    int32_t k, m, hx, hy, ix, iy;
    hx = rl_bits(x); ix = hx & 0x7fffffff; hy = rl_bits(y); iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return rl_atanf(y);
    m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
    if (iy == 0) {
        if (m < 2) return y;
        return (m == 2) ? pi + tiny : -pi - tiny;
    }
    if (ix == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            if (m == 0) return pi_o_4 + tiny;
            if (m == 1) return -pi_o_4 - tiny;
            return (m == 2) ? 3.0f * pi_o_4 + tiny : -3.0f * pi_o_4 - tiny;
        }
        if (m == 0) return 0.0f;
        if (m == 1) return -0.0f;
        return (m == 2) ? pi + tiny : -pi - tiny;
    }
    if (iy == 0x7f800000) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
    k = (iy - ix) >> 23;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = rl_atanf(rl_fabsf(y / x));
    if (m == 0) return z;
    if (m == 1) return rl_float(rl_bits(z) ^ (int32_t)0x80000000);
    return (m == 2) ? pi - (z - pi_lo) : (z - pi_lo) - pi;
}

// sqrt_rn = a correctly rounded float square root (sqrtf on the host; the device's IEEE sqrt)
template <class Sqrt>
RL_FN float rl_asinf(float x, Sqrt sqrt_rn) {
    const float one = 1.0f, huge = 1.0e30f, pio2_hi = 1.57079637050628662109375f, pio2_lo = -4.37113900018624283e-8f,
                pio4_hi = 0.785398185253143310546875f,
                p0 = 1.666675248e-1f, p1 = 7.495297643e-2f, p2 = 4.547037598e-2f, p3 = 2.417951451e-2f, p4 = 4.216630880e-2f;
    float t, w, p, q, c, r, s;
    int32_t hx, ix;
    hx = rl_bits(x); ix = hx & 0x7fffffff;
    if (ix == 0x3f800000) return x * pio2_hi + x * pio2_lo;
    if (ix > 0x3f800000) return (x - x) / (x - x);
    if (ix < 0x3f000000) {                                  // |x| < 0.5
        if (ix < 0x32000000) { if (huge + x > one) return x; }
        else { t = x * x; w = t * (p0 + t * (p1 + t * (p2 + t * (p3 + t * p4)))); return x + x * w; }
    }
    w = one - rl_fabsf(x); t = w * 0.5f;
    p = t * (p0 + t * (p1 + t * (p2 + t * (p3 + t * p4))));
    s = sqrt_rn(t);
    if (ix >= 0x3F79999A) {                                 // |x| > 0.975
        t = pio2_hi - (2.0f * (s + s * p) - pio2_lo);
    } else {
        w = rl_float(rl_bits(s) & (int32_t)0xfffff000);
        c = (t - w * w) / (s + w); r = p; p = 2.0f * s * r - (pio2_lo - 2.0f * c); q = pio4_hi - 2.0f * w; t = pio4_hi - (p - q);
    }
    return (hx > 0) ? t : -t;
}

#endif
