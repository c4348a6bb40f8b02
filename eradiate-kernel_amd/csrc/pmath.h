// pmath.h -- bit-reproducible fp32 math shared by host (gcc) and device (hipcc, gfx950).
//
// Why this exists: the reference's scalar_rgb variant calls libm (`enoki::log/exp/sincos/cbrt`
// reduce to std:: functions for scalar floats, e.g. /root/reference/src/librender/medium.cpp:65,
// src/phase/hg.cpp:70, src/phase/rayleigh.cpp:52-53).  glibc's and ROCm's device libm differ by
// >= 1 ulp, which flips comparisons such as `sampled_t <= maxt` (medium.cpp:66) and desynchronises
// the per-pixel PCG32 stream.  Every transcendental below is therefore written with IEEE
// add/mul/fma and integer operations only, so the SAME source gives the SAME bits when compiled
// by gcc for x86-64 (-mfma -ffp-contract=off) and by hipcc for gfx950 (-ffp-contract=off).
// Division and sqrt are IEEE correctly rounded on both (hipcc default
// -fhip-fp32-correctly-rounded-divide-sqrt).  Accuracy is ~1 ulp; tests/test_pmath.py checks it
// against libm.  Denormals: both sides run flush-to-zero (reference worker threads do,
// /root/reference/src/librender/integrator.cpp:117); the routines never rely on denormal
// intermediates.
#pragma once
#include <stdint.h>
#include <math.h>
#include <string.h>

#if defined(__HIPCC__)
#  define PM_HD __host__ __device__ inline
#else
#  define PM_HD static inline
#endif

PM_HD uint32_t pm_bits(float x) { return __builtin_bit_cast(uint32_t, x); }
PM_HD float pm_from_bits(uint32_t u) { return __builtin_bit_cast(float, u); }
PM_HD uint64_t pm_bits_d(double x) { return __builtin_bit_cast(uint64_t, x); }
PM_HD double pm_from_bits_d(uint64_t u) { return __builtin_bit_cast(double, u); }

PM_HD float pm_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
PM_HD float pm_sqrt(float x) { return __builtin_sqrtf(x); }
PM_HD float pm_rcp(float x) { return 1.0f / x; }
PM_HD float pm_rsqrt(float x) { return 1.0f / __builtin_sqrtf(x); }
PM_HD float pm_abs(float x) { return pm_from_bits(pm_bits(x) & 0x7fffffffu); }
PM_HD float pm_min(float a, float b) { return b < a ? b : a; }   // std::min semantics
PM_HD float pm_max(float a, float b) { return a < b ? b : a; }   // std::max semantics
PM_HD float pm_safe_sqrt(float x) { return __builtin_sqrtf(pm_max(x, 0.0f)); }
PM_HD float pm_floor(float x) { return __builtin_floorf(x); }
PM_HD float pm_ceil(float x) { return __builtin_ceilf(x); }
PM_HD float pm_inf() { return pm_from_bits(0x7f800000u); }
PM_HD float pm_nan() { return pm_from_bits(0x7fc00000u); }
PM_HD int pm_isfinite(float x) { return (pm_bits(x) & 0x7f800000u) != 0x7f800000u; }

// enoki::sign / mulsign / mulsign_neg (used by coordinate_system,
// /root/reference/include/mitsuba/core/vector.h:116-136)
PM_HD float pm_sign(float x) { return pm_from_bits((pm_bits(x) & 0x80000000u) | 0x3f800000u); }
PM_HD float pm_mulsign(float a, float b) { return pm_from_bits(pm_bits(a) ^ (pm_bits(b) & 0x80000000u)); }
PM_HD float pm_mulsign_neg(float a, float b) { return pm_from_bits(pm_bits(a) ^ (~pm_bits(b) & 0x80000000u)); }

// Natural logarithm (Cephes-style reduction to [sqrt(1/2), sqrt(2)) + degree-8 polynomial).
PM_HD float pm_log(float x) {
#if defined(PM_USE_LIBM) && !defined(__HIPCC__)
    return logf(x);   // oracle/Makefile, liboracle_libm.so: sensitivity measurement only
#endif
#if defined(EXP_FASTMATH) && defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_logf(x) * 0.6931471805599453f;   // measurement only: breaks parity
#endif
    uint32_t ix = pm_bits(x);
    if (ix >= 0x7f800000u) {            // negative, inf or NaN
        if (ix == 0x7f800000u) return x;                  // +inf
        if ((ix << 1) == 0u) return -pm_inf();            // -0
        return pm_nan();                                  // negative or NaN
    }
    if (ix < 0x00800000u) return -pm_inf();               // +0 and denormals (DAZ)
    int e = (int) (ix >> 23) - 126;
    float m = pm_from_bits((ix & 0x007fffffu) | 0x3f000000u);   // [0.5, 1)
    if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; }
    else                           { m = m - 1.0f; }
    float z = m * m;
    float y = 7.0376836292E-2f;
    y = pm_fma(y, m, -1.1514610310E-1f);
    y = pm_fma(y, m,  1.1676998740E-1f);
    y = pm_fma(y, m, -1.2420140846E-1f);
    y = pm_fma(y, m,  1.4249322787E-1f);
    y = pm_fma(y, m, -1.6668057665E-1f);
    y = pm_fma(y, m,  2.0000714765E-1f);
    y = pm_fma(y, m, -2.4999993993E-1f);
    y = pm_fma(y, m,  3.3333331174E-1f);
    y = y * m * z;
    float fe = (float) e;
    y = pm_fma(fe, -2.12194440e-4f, y);
    y = pm_fma(-0.5f, z, y);
    float r = m + y;
    r = pm_fma(fe, 0.693359375f, r);
    return r;
}

// Exponential. Underflows to +0 below ln(FLT_MIN) (flush-to-zero semantics), overflows to +inf.
PM_HD float pm_exp(float x) {
#if defined(PM_USE_LIBM) && !defined(__HIPCC__)
    return expf(x);   // oracle/Makefile, liboracle_libm.so: sensitivity measurement only
#endif
#if defined(EXP_FASTMATH) && defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_exp2f(x * 1.4426950408889634f);   // measurement only: breaks parity
#endif
    if (!(x == x)) return x;
    if (x > 88.7228317f) return pm_inf();
    if (x < -87.3365402f) return 0.0f;
    float fn = pm_floor(pm_fma(x, 1.44269504088896341f, 0.5f));
    float r = pm_fma(fn, -0.693359375f, x);
    r = pm_fma(fn, 2.12194440e-4f, r);
    float p = 1.9875691500E-4f;
    p = pm_fma(p, r, 1.3981999507E-3f);
    p = pm_fma(p, r, 8.3334519073E-3f);
    p = pm_fma(p, r, 4.1665795894E-2f);
    p = pm_fma(p, r, 1.6666665459E-1f);
    p = pm_fma(p, r, 5.0000001201E-1f);
    p = pm_fma(p, r * r, r);
    p = p + 1.0f;
    int n = (int) fn;
    if (n > 127) { p = p + p; n -= 1; }
    if (n < -126) n = -126;              // cannot happen for x >= -87.34 (kept as a guard)
    return p * pm_from_bits((uint32_t) (n + 127) << 23);
}

// Simultaneous sine / cosine, |x| <~ 1e4 (all call sites pass 2*pi*u or a concentric-disk angle).
PM_HD void pm_sincos(float x, float *s_out, float *c_out) {
#if defined(PM_USE_LIBM) && !defined(__HIPCC__)
    *s_out = sinf(x); *c_out = cosf(x); return;   // oracle/Makefile, liboracle_libm.so: sensitivity measurement only
#endif
    float fj = pm_floor(pm_fma(x, 0.636619772367581343f, 0.5f));
    // Cody-Waite: pi/2 = 1.5703125 + 4.837512969970703125e-4 + 7.54978995489188e-8
    float r = pm_fma(fj, -1.5703125f, x);
    r = pm_fma(fj, -4.837512969970703125e-4f, r);
    r = pm_fma(fj, -7.54978995489188e-8f, r);
    float z = r * r;
    float sp = -1.9515295891E-4f;
    sp = pm_fma(sp, z, 8.3321608736E-3f);
    sp = pm_fma(sp, z, -1.6666654611E-1f);
    float s = pm_fma(sp * z, r, r);
    float cp = 2.443315711809948E-5f;
    cp = pm_fma(cp, z, -1.388731625493765E-3f);
    cp = pm_fma(cp, z, 4.166664568298827E-2f);
    float c = pm_fma(cp * z, z, pm_fma(-0.5f, z, 1.0f));
    int j = (int) fj;
    float ss = (j & 1) ? c : s;
    float cc = (j & 1) ? s : c;
    if (j & 2) ss = -ss;
    if ((j + 1) & 2) cc = -cc;
    *s_out = ss;
    *c_out = cc;
}

// Cube root (sign-preserving), used by the Rayleigh phase function
// (/root/reference/src/phase/rayleigh.cpp:51-55).
PM_HD float pm_cbrt(float x) {
#if defined(PM_USE_LIBM) && !defined(__HIPCC__)
    return cbrtf(x);   // oracle/Makefile, liboracle_libm.so: sensitivity measurement only
#endif
    uint32_t ix = pm_bits(x);
    uint32_t sign = ix & 0x80000000u;
    uint32_t ax = ix & 0x7fffffffu;
    if (ax >= 0x7f800000u) return x;              // inf / NaN
    if (ax < 0x00800000u) return pm_from_bits(sign);   // +-0, denormals (DAZ)
    float a = pm_from_bits(ax);
    float y = pm_from_bits(ax / 3u + 0x2a5137a0u);     // ~5 % initial guess
    // three Newton steps: y <- y - (y^3 - a) / (3 y^2)
    for (int i = 0; i < 3; ++i) {
        float y2 = y * y;
        float num = pm_fma(y2, y, -a);
        y = y - num / (3.0f * y2);
    }
    return pm_from_bits(pm_bits(y) | sign);
}

// ---- double-precision helpers for pm_pow (RPV BRDF, /root/reference/src/bsdfs/rpv.cpp:85-167) ----
PM_HD double pm_fma_d(double a, double b, double c) { return __builtin_fma(a, b, c); }

PM_HD double pm_log_d(double x) {          // x > 0, normal; ~1e-12 relative accuracy
    uint64_t ix = pm_bits_d(x);
    int e = (int) (ix >> 52) - 1022;
    uint64_t mb = (ix & 0x000fffffffffffffull) | 0x3fe0000000000000ull;   // [0.5,1)
    double m = pm_from_bits_d(mb);
    if (m < 0.70710678118654752440) { e -= 1; m = m + m; }
    // log(m) = 2 atanh(s), s = (m-1)/(m+1), |s| <= 0.1716
    double s = (m - 1.0) / (m + 1.0);
    double s2 = s * s;
    double p = 1.0 / 23.0;
    p = pm_fma_d(p, s2, 1.0 / 21.0);
    p = pm_fma_d(p, s2, 1.0 / 19.0);
    p = pm_fma_d(p, s2, 1.0 / 17.0);
    p = pm_fma_d(p, s2, 1.0 / 15.0);
    p = pm_fma_d(p, s2, 1.0 / 13.0);
    p = pm_fma_d(p, s2, 1.0 / 11.0);
    p = pm_fma_d(p, s2, 1.0 / 9.0);
    p = pm_fma_d(p, s2, 1.0 / 7.0);
    p = pm_fma_d(p, s2, 1.0 / 5.0);
    p = pm_fma_d(p, s2, 1.0 / 3.0);
    p = pm_fma_d(p, s2, 1.0);
    return pm_fma_d((double) e, 0.69314718055994530942, 2.0 * s * p);
}

PM_HD double pm_exp_d(double x) {          // |x| < 700
    double fn = __builtin_floor(pm_fma_d(x, 1.44269504088896340736, 0.5));
    double r = pm_fma_d(fn, -0.693147180369123816490, x);
    r = pm_fma_d(fn, -1.90821492927058770002e-10, r);
    double p = 1.0 / 479001600.0;
    p = pm_fma_d(p, r, 1.0 / 39916800.0);
    p = pm_fma_d(p, r, 1.0 / 3628800.0);
    p = pm_fma_d(p, r, 1.0 / 362880.0);
    p = pm_fma_d(p, r, 1.0 / 40320.0);
    p = pm_fma_d(p, r, 1.0 / 5040.0);
    p = pm_fma_d(p, r, 1.0 / 720.0);
    p = pm_fma_d(p, r, 1.0 / 120.0);
    p = pm_fma_d(p, r, 1.0 / 24.0);
    p = pm_fma_d(p, r, 1.0 / 6.0);
    p = pm_fma_d(p, r, 0.5);
    p = pm_fma_d(p, r, 1.0);
    p = pm_fma_d(p, r, 1.0);
    int n = (int) fn;
    uint64_t sb = (uint64_t) (n + 1023) << 52;
    double sc = pm_from_bits_d(sb);
    return p * sc;
}

// x^y for x >= 0 (std::pow semantics for the cases the RPV model can produce).
PM_HD float pm_pow(float x, float y) {
#if defined(PM_USE_LIBM) && !defined(__HIPCC__)
    return powf(x, y);   // oracle/Makefile, liboracle_libm.so: sensitivity measurement only
#endif
    if (y == 0.0f) return 1.0f;
    if (!(x == x) || !(y == y)) return pm_nan();
    if (x < 0.0f) return pm_nan();
    if (pm_bits(x) < 0x00800000u)                       // +0 / denormal
        return y > 0.0f ? 0.0f : pm_inf();
    if (pm_bits(x) == 0x7f800000u) return y > 0.0f ? pm_inf() : 0.0f;
    double t = (double) y * pm_log_d((double) x);
    if (t > 88.8) return pm_inf();
    if (t < -87.3365402) return 0.0f;           // below FLT_MIN: flush-to-zero semantics
    float r = (float) pm_exp_d(t);
    return r < 1.17549435e-38f ? 0.0f : r;
}
