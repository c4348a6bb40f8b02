// pmath.h -- bit-reproducible fp32 math shared by host (gcc) and device (hipcc, gfx950).
//
// Why this exists: the reference's scalar_rgb variant calls libm (`enoki::log/exp/sincos/cbrt/pow` reduce to the std::
// functions for scalar floats, e.g. /root/reference/src/librender/medium.cpp:65, src/phase/hg.cpp:70,
// src/phase/rayleigh.cpp:52-53, src/bsdfs/rpv.cpp:85-167).  libm is a third-party dependency of the reference that is not
// in /root/reference: on the platform Eradiate ships for it is glibc, pinned here to the version of this image, 2.35.
// Any 1-ulp deviation from it is a chance to flip a comparison such as `sampled_t <= maxt` (medium.cpp:66) and to
// desynchronise the pixel's PCG32 stream, so the transcendentals below RESTATE GLIBC'S PUBLISHED ALGORITHMS with IEEE add /
// mul / fma / div and integer operations only:
//   logf, expf, powf, sinf / cosf : the fp64-evaluated routines glibc has shipped since 2.28 (from Arm's Optimized Routines;
//       sysdeps/ieee754/flt-32/{e_logf,e_expf,e_powf,s_sincosf}.c with the tables of {e_logf,e_exp2f,e_powf_log2}_data.c and
//       s_sincosf_data.c), with the fused multiply-adds of the x86-64 FMA build (sysdeps/x86_64/fpu/multiarch/*-fma.c), which is
//       also what aarch64 builds give;
//   cbrtf : the classic flt-32 routine (s_cbrtf.c, glibc <= 2.40; compiled without FMA: it has no multiarch variant).
// The SAME source gives the SAME bits when compiled by gcc for x86-64 (-mfma -ffp-contract=off) and by hipcc for gfx950
// (-ffp-contract=off; fp32 / fp64 division and sqrt are IEEE correctly rounded on both), and -- measured, tests/test_pmath.py
// and tools/pmath_vs_glibc.cpp -- the same bits as glibc 2.35 itself for EVERY fp32 argument of logf, expf (|x| < 88.73),
// sinf / cosf (|x| < 120; beyond that the correctly rounded routine below), cbrtf, and for 10^8 random arguments of powf.  So a
// build of the restatement on glibc (oracle/liboracle_libm.so) renders the same film, bit for bit, as the build on this header.
// Tables: 16 x {1/c, log c}, 16 x log2 c, 32 x 2^(i/32): 640 bytes, read per lane (constant address space on the device).
//
// The correctly rounded routines of the first half of round 4 stay below as pm_*_cr (fp64 evaluation to < 2^-47, one rounding;
// tools/pmath_coeffs.py): table-free, the least distance to ANY good libm (they differ from glibc in 0.06 % (exp, pow) to 1.3 %
// (sin, cos) of the calls, 11 % for cbrtf -- glibc's own distance from correct rounding), selected by -DPM_CORRECTLY_ROUNDED.
// Denormals: both sides run fp32 flush-to-zero (reference worker threads do, /root/reference/src/librender/integrator.cpp:117):
// denormal arguments count as zero, results below FLT_MIN are returned as zero; no fp64 intermediate comes near the fp64
// denormal range.
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
// 1 / x, the IEEE quotient.  On the device: one Newton step on v_rcp_f32, then v_div_fixup_f32 for zeros / infinities / NaNs -- four
// instructions, 35 cycles of latency, where the compiler's expansion of 1.0f / x in this build mode (two v_div_scale, v_rcp, two
// denormal-mode switches, five fma, v_div_fmas, v_div_fixup) takes 92.  The two agree on every one of the 2^32 arguments
// (tests/micro/rcp_exhaustive.hip, run by tests/test_gpu_parity.py::test_pm_rcp_is_the_division_for_every_argument).
PM_HD float pm_rcp(float x) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(PM_RCP_BY_DIVISION)
    const float y0 = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, y0, 1.0f);
    return __builtin_amdgcn_div_fixupf(__builtin_fmaf(e, y0, y0), x, 1.0f);
#else
    return 1.0f / x;
#endif
}
PM_HD float pm_rsqrt(float x) { return pm_rcp(__builtin_sqrtf(x)); }
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

PM_HD double pm_fma_d(double a, double b, double c) { return __builtin_fma(a, b, c); }

// log(m) for a normal fp32 m given by its bits, as e * ln2 + 2 atanh(s), s = (m' - 1) / (m' + 1), m' in [sqrt(1/2), sqrt(2)).
// The quotient comes from a Newton reciprocal (fp32 seed 2^-9.6 -> 2^-19, two fp64 steps -> 2^-76): no division sequence.
PM_HD double pm_log_core_cr(uint32_t ix) {
    int e = (int) (ix >> 23) - 126;
    float mf = pm_from_bits((ix & 0x007fffffu) | 0x3f000000u);   // [0.5, 1)
    if (mf < 0.707106781186547524f) { e -= 1; mf = mf + mf; }      // [sqrt(1/2), sqrt(2)): exact
    float tf = mf + 1.0f;
    float yf = pm_fma(pm_fma(0x1.deab9f5bac12fp-4f, tf, -0x1.71e431e019c40p-1f), tf, 0x1.7a4e362d364a6p+0f);
    yf = pm_fma(yf, pm_fma(-tf, yf, 1.0f), yf);
    double m = (double) mf, t = m + 1.0, y = (double) yf;
    y = pm_fma_d(y, pm_fma_d(-t, y, 1.0), y);
    y = pm_fma_d(y, pm_fma_d(-t, y, 1.0), y);
    double s = (m - 1.0) * y;
    double z = s * s;
    double p = 0x1.58dd5ccd0ae5bp-3;                  // 2 * (atanh(sqrt z) / sqrt z), degree 6, error 2^-52.4
    p = pm_fma_d(p, z, 0x1.7322766efc727p-3);
    p = pm_fma_d(p, z, 0x1.c722d2919c869p-3);
    p = pm_fma_d(p, z, 0x1.249240ad6fd9fp-2);
    p = pm_fma_d(p, z, 0x1.999999a448740p-2);
    p = pm_fma_d(p, z, 0x1.5555555552cebp-1);
    p = pm_fma_d(p, z, 0x1.0000000000001p+1);
    return pm_fma_d((double) e, 0.69314718055994530942, s * p);
}

// Natural logarithm, correctly rounded.
PM_HD float pm_log_cr(float x) {
    uint32_t ix = pm_bits(x);
    if (ix >= 0x7f800000u) {            // negative, inf or NaN
        if (ix == 0x7f800000u) return x;                  // +inf
        if ((ix << 1) == 0u) return -pm_inf();            // -0
        return pm_nan();                                  // negative or NaN
    }
    if (ix < 0x00800000u) return -pm_inf();               // +0 and denormals (DAZ)
    return (float) pm_log_core_cr(ix);
}

// exp(x) in fp64 for |x| < 700: x = n ln2 + r, |r| <= ln2 / 2, degree-10 polynomial (relative error 2^-52.1), exponent added in place.
PM_HD double pm_exp_d_cr(double x) {
    double fn = __builtin_floor(pm_fma_d(x, 1.44269504088896340736, 0.5));
    double r = pm_fma_d(fn, -0.693147180369123816490, x);          // 32 significant bits: fn * hi is exact
    r = pm_fma_d(fn, -1.90821492927058770002e-10, r);
    double p = 0x1.2727a88044faap-22;
    p = pm_fma_d(p, r, 0x1.72f98543deccbp-19);
    p = pm_fma_d(p, r, 0x1.a01b6ce393267p-16);
    p = pm_fma_d(p, r, 0x1.a0197a4e9bd9dp-13);
    p = pm_fma_d(p, r, 0x1.6c16c0c0acbd4p-10);
    p = pm_fma_d(p, r, 0x1.1111112d4977ep-7);
    p = pm_fma_d(p, r, 0x1.55555555933c2p-5);
    p = pm_fma_d(p, r, 0x1.555555554bc09p-3);
    p = pm_fma_d(p, r, 0x1.ffffffffffe17p-2);
    p = pm_fma_d(p, r, 0x1.000000000001dp+0);
    p = pm_fma_d(p, r, 1.0);
    int64_t n = (int64_t) (int) fn;
    return pm_from_bits_d(pm_bits_d(p) + ((uint64_t) n << 52));   // p in [0.70, 1.42], |n| <= 1010: stays normal
}

// Exponential, correctly rounded. Underflows to +0 below ln(FLT_MIN) (flush-to-zero semantics), overflows to +inf.
PM_HD float pm_exp_cr(float x) {
    if (!(x == x)) return x;
    if (x > 88.7228317f) return pm_inf();
    if (x < -87.3365402f) return 0.0f;
    double r = pm_exp_d_cr((double) x);
    return r < 1.17549435082228750797e-38 ? 0.0f : (float) r;
}

// Simultaneous sine / cosine, correctly rounded for |x| < 1.6e6 (j * pio2_hi is exact below 2^20 quadrants; all call sites pass
// 2*pi*u or a concentric-disk angle); beyond that the reduction loses accuracy, deterministically.
PM_HD void pm_sincos_cr(float x, float *s_out, float *c_out) {
    double xd = (double) x;
    double fj = __builtin_floor(pm_fma_d(xd, 0.63661977236758134308, 0.5));
    double r = pm_fma_d(fj, -1.57079632673412561417, xd);           // first 33 bits of pi / 2
    r = pm_fma_d(fj, -6.07710050650619224932e-11, r);
    double z = r * r;
    double sp = -0x1.a950a7cb0bcc1p-26;               // sin(sqrt z) / sqrt z, degree 5, error 2^-47.7
    sp = pm_fma_d(sp, z, 0x1.71d731f8b7cd4p-19);
    sp = pm_fma_d(sp, z, -0x1.a019f8a316107p-13);
    sp = pm_fma_d(sp, z, 0x1.1111110bdf4c5p-7);
    sp = pm_fma_d(sp, z, -0x1.5555555550f10p-3);
    sp = pm_fma_d(sp, z, 0x1.fffffffffffd9p-1);
    float s = (float) (r * sp);
    double cp = 0x1.1b8060fff7aafp-29;                // cos(sqrt z), degree 6, error 2^-54
    cp = pm_fma_d(cp, z, -0x1.27df18b382dc0p-22);
    cp = pm_fma_d(cp, z, 0x1.a019f78fb533dp-16);
    cp = pm_fma_d(cp, z, -0x1.6c16c16338251p-10);
    cp = pm_fma_d(cp, z, 0x1.555555554dd98p-5);
    cp = pm_fma_d(cp, z, -0x1.fffffffffff67p-2);
    cp = pm_fma_d(cp, z, 0x1.fffffffffffffp-1);
    float c = (float) cp;
    int j = (int) fj;
    float ss = (j & 1) ? c : s;
    float cc = (j & 1) ? s : c;
    if (j & 2) ss = -ss;
    if ((j + 1) & 2) cc = -cc;
    *s_out = ss;
    *c_out = cc;
}

// Cube root (sign-preserving), correctly rounded; used by the Rayleigh phase function
// (/root/reference/src/phase/rayleigh.cpp:51-55).  Division-free: r -> a^(-1/3) by Newton (r <- r (4 - a r^3) / 3, error e -> 2 e^2:
// integer seed 3.4e-2, two fp32 steps -> 1e-5, two fp64 steps -> 1e-19), then a r^2.
PM_HD float pm_cbrt_cr(float x) {
    uint32_t ix = pm_bits(x);
    uint32_t sign = ix & 0x80000000u;
    uint32_t ax = ix & 0x7fffffffu;
    if (ax >= 0x7f800000u) return x;              // inf / NaN
    if (ax < 0x00800000u) return pm_from_bits(sign);   // +-0, denormals (DAZ)
    float af = pm_from_bits(ax);
    float rf = pm_from_bits(0x54a232a0u - ax / 3u);
    for (int i = 0; i < 2; ++i) {                 // a r^3 as ((a r) r) r: every intermediate stays normal for every normal a
        float ar3 = af * rf * rf * rf;
        rf = rf * pm_fma(ar3, -0.333333333f, 1.33333333f);
    }
    double a = (double) af, r = (double) rf;
    for (int i = 0; i < 2; ++i) {
        double ar3 = a * r * r * r;
        r = r * pm_fma_d(ar3, -0.33333333333333333333, 1.33333333333333333333);
    }
    float y = (float) (a * r * r);
    return pm_from_bits(pm_bits(y) | sign);
}

// x^y for x >= 0 (std::pow semantics for the cases the RPV model can produce; /root/reference/src/bsdfs/rpv.cpp:85-167):
// exp(y log x) with both in fp64.
PM_HD float pm_pow_cr(float x, float y) {
    if (y == 0.0f) return 1.0f;
    if (!(x == x) || !(y == y)) return pm_nan();
    if (x < 0.0f) return pm_nan();
    if (pm_bits(x) < 0x00800000u)                       // +0 / denormal
        return y > 0.0f ? 0.0f : pm_inf();
    if (pm_bits(x) == 0x7f800000u) return y > 0.0f ? pm_inf() : 0.0f;
    double t = (double) y * pm_log_core_cr(pm_bits(x));
    if (t > 88.8) return pm_inf();
    if (t < -87.3365402) return 0.0f;           // below FLT_MIN: flush-to-zero semantics
    float r = (float) pm_exp_d_cr(t);
    return r < 1.17549435e-38f ? 0.0f : r;
}

// ------------------------------------------------------------------------------------------------------------------------------
// glibc's routines (see the header).  Tables: one initialiser, a host copy and a device copy in the constant address space.
#define PM_LOGF_TAB_INIT { /* {1 / c, log c} of the 16 sub-intervals of [0x1.66p-1, 0x1.66p0) (e_logf_data.c) */ \
    0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2, 0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2, \
    0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2, 0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3, \
    0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3, 0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3, \
    0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4, 0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4, \
    0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5, 0x1p+0, 0x0p+0, \
    0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5, 0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4, \
    0x1.b2036576afce6p-1, 0x1.526e57720db08p-3, 0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3, \
    0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2, 0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2 }
#define PM_LOG2F_TAB_INIT { /* log2 c of the same sub-intervals (e_powf_log2_data.c; its 1 / c column is the one above) */ \
    -0x1.efec65b963019p-2, -0x1.b0b6832d4fca4p-2, -0x1.7418b0a1fb77bp-2, -0x1.39de91a6dcf7bp-2, \
    -0x1.01d9bf3f2b631p-2, -0x1.97c1d1b3b7afp-3, -0x1.2f9e393af3c9fp-3, -0x1.960cbbf788d5cp-4, \
    -0x1.a6f9db6475fcep-5, 0x0p+0, 0x1.338ca9f24f53dp-4, 0x1.476a9543891bap-3, \
    0x1.e840b4ac4e4d2p-3, 0x1.40645f0c6651cp-2, 0x1.88e9c2c1b9ff8p-2, 0x1.ce0a44eb17bccp-2 }
#define PM_EXP2F_TAB_INIT { /* bits(2^(i/32)) - (i << 47), i = 0..31 (e_exp2f_data.c; tools/pmath_coeffs.py recomputes them) */ \
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull, \
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull, \
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull, \
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull, \
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull, \
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull, \
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull, \
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull }
static const double pm_host_logf_tab[32] = PM_LOGF_TAB_INIT;
static const double pm_host_log2f_tab[16] = PM_LOG2F_TAB_INIT;
static const uint64_t pm_host_exp2f_tab[32] = PM_EXP2F_TAB_INIT;
#if defined(__HIPCC__)
static __constant__ const double pm_dev_logf_tab[32] = PM_LOGF_TAB_INIT;
static __constant__ const double pm_dev_log2f_tab[16] = PM_LOG2F_TAB_INIT;
static __constant__ const uint64_t pm_dev_exp2f_tab[32] = PM_EXP2F_TAB_INIT;
#endif
#if defined(__HIP_DEVICE_COMPILE__) && defined(PM_TABLES_IN_LDS)
// The two tables of the hot functions (log: every free-flight distance, exp: every transmittance) as LDS copies: a per-lane lookup
// is one ds_read_b128 / ds_read_b64 (an LDS round trip) instead of a vector-memory load on the critical path of a tracking step --
// measured on C3: 538 (constant address space) -> see profiles/r04_ab_experiments.log.  16 entries x 16 bytes and 32 x 8 bytes are
// exactly 64 banks each: distinct entries never share a bank, equal entries broadcast -- conflict-free for any index pattern.
// Every kernel that can reach pm_log / pm_exp calls pm_tables_to_lds() and passes a workgroup barrier before its first use.
__shared__ __attribute__((aligned(16))) double pm_lds_logf_tab[32];
__shared__ __attribute__((aligned(16))) uint64_t pm_lds_exp2f_tab[32];
__device__ inline void pm_tables_to_lds(uint32_t tid) {
    if (tid < 32u) { pm_lds_logf_tab[tid] = pm_dev_logf_tab[tid]; pm_lds_exp2f_tab[tid] = pm_dev_exp2f_tab[tid]; }
}
#  define PM_LOGF_TAB pm_lds_logf_tab
#  define PM_LOG2F_TAB pm_dev_log2f_tab
#  define PM_EXP2F_TAB pm_lds_exp2f_tab
#elif defined(__HIP_DEVICE_COMPILE__)
__device__ inline void pm_tables_to_lds(uint32_t) {}
#  define PM_LOGF_TAB pm_dev_logf_tab
#  define PM_LOG2F_TAB pm_dev_log2f_tab
#  define PM_EXP2F_TAB pm_dev_exp2f_tab
#else
#  if defined(__HIPCC__)
__device__ inline void pm_tables_to_lds(uint32_t) {}      // host pass of a HIP translation unit: declaration only
#  endif
#  define PM_LOGF_TAB pm_host_logf_tab
#  define PM_LOG2F_TAB pm_host_log2f_tab
#  define PM_EXP2F_TAB pm_host_exp2f_tab
#endif

// e_logf.c: x = 2^k z, z in [0x1.66p-1, 0x1.66p0) (OFF = 0x3f330000), c near the centre of z's sub-interval;
// log x = log1p(z / c - 1) + log c + k ln2 with a degree-3 polynomial for log1p(r) - r.
PM_HD float pm_log(float x) {
#if defined(PM_USE_LIBM) && !defined(__HIPCC__)
    return logf(x);   // oracle/Makefile, liboracle_libm.so: glibc itself, the pin of this header
#endif
#if defined(EXP_FASTMATH) && defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_logf(x) * 0.6931471805599453f;   // measurement only: breaks parity
#endif
#if defined(PM_CORRECTLY_ROUNDED)
    return pm_log_cr(x);
#endif
    uint32_t ix = pm_bits(x);
    if (__builtin_expect(ix - 0x00800000u >= 0x7f000000u, 0)) {   // everything but a positive normal number: ONE test on the common path
        if (ix == 0x7f800000u) return x;                  // +inf
        if ((ix << 1) == 0u) return -pm_inf();            // +-0
        if (ix < 0x00800000u) return -pm_inf();           // denormals (DAZ)
        return pm_nan();                                  // negative or NaN
    }
    // (glibc tests x == 1 first; the arithmetic below returns +0 for it by itself: r = 0, y0 = 0)
    uint32_t tmp = ix - 0x3f330000u;
    uint32_t i = (tmp >> 19) & 15u;
    int k = (int32_t) tmp >> 23;
    double z = (double) pm_from_bits(ix - (tmp & 0xff800000u));
    double invc = PM_LOGF_TAB[2 * i], logc = PM_LOGF_TAB[2 * i + 1];
    double r = pm_fma_d(z, invc, -1.0);
    double y0 = pm_fma_d((double) k, 0x1.62e42fefa39efp-1, logc);
    double r2 = r * r;
    double y = pm_fma_d(0x1.5575b0be00b6ap-2, r, -0x1.ffffef20a4123p-2);
    y = pm_fma_d(-0x1.00ea348b88334p-2, r2, y);
    y = pm_fma_d(y, r2, y0 + r);
    return (float) y;
}

// The tail both expf and powf end in (e_expf.c, e_powf.c: exp2_inline): 2^(k/32) from the table, exponent added in place, degree-3
// polynomial in the remainder r with the coefficients c0..c2 scaled for the caller's unit of r.
PM_HD double pm_exp2_tail(uint64_t ki, double r, double c0, double c1, double c2) {
    double s = pm_from_bits_d(PM_EXP2F_TAB[ki & 31u] + (ki << 47));
    double z = pm_fma_d(c0, r, c1);
    double r2 = r * r;
    double y = pm_fma_d(c2, r, 1.0);
    y = pm_fma_d(z, r2, y);
    return y * s;
}

// e_expf.c: x * 32 / ln2 = k + r (round to nearest through the 0x1.8p52 shift), exp x = 2^(k/32) * 2^(r/32).
// Underflows to +0 below ln(FLT_MIN) (flush-to-zero semantics), overflows to +inf.
PM_HD float pm_exp(float x) {
#if defined(PM_USE_LIBM) && !defined(__HIPCC__)
    return expf(x);   // oracle/Makefile, liboracle_libm.so: glibc itself, the pin of this header
#endif
#if defined(EXP_FASTMATH) && defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_exp2f(x * 1.4426950408889634f);   // measurement only: breaks parity
#endif
#if defined(PM_CORRECTLY_ROUNDED)
    return pm_exp_cr(x);
#endif
    if (__builtin_expect((pm_bits(x) & 0x7fffffffu) > 0x42aeac4fu, 0)) {     // |x| > 87.3365402 or NaN: ONE test on the common path
        if (!(x == x)) return x;
        if (x > 88.7228317f) return pm_inf();             // 0x1.62e42ep6f, e_expf.c's overflow bound
        if (x < 0.0f) return 0.0f;                        // below ln(FLT_MIN): glibc returns a denormal here, zero under flush-to-zero
    }
    double xd = (double) x;
    double z = 0x1.71547652b82fep+5 * xd;
    double kd = z + 0x1.8p+52;
    uint64_t ki = pm_bits_d(kd);
    kd = kd - 0x1.8p+52;
    double r = pm_fma_d(0x1.71547652b82fep+5, xd, -kd);   // the FMA build contracts z - kd over the product
    double y = pm_exp2_tail(ki, r, 0x1.c6af84b912394p-20, 0x1.ebfce50fac4f3p-13, 0x1.62e42ff0c52d6p-6);
    return y < 1.17549435082228750797e-38 ? 0.0f : (float) y;
}

// s_sincosf.h: sinf_poly -- the sine (n even) or cosine (n odd) polynomial of glibc's sincos_t; `neg` selects the second table entry
// (the cosine coefficients negated), which is how the routine applies the quadrant's sign to a cosine.
PM_HD float pm_sinf_poly(double x, double x2, bool neg, int n) {
    if ((n & 1) == 0) {
        double x3 = x * x2;
        double s1 = pm_fma_d(x2, -0x1.994eb3774cf24p-13, 0x1.1107605230bc4p-7);
        double x7 = x3 * x2;
        double s = pm_fma_d(x3, -0x1.555545995a603p-3, x);
        return (float) pm_fma_d(x7, s1, s);
    }
    double sg = neg ? -1.0 : 1.0;
    double x4 = x2 * x2;
    double c2 = pm_fma_d(x2, sg * 0x1.99343027bf8c3p-16, sg * -0x1.6c087e89a359dp-10);
    double c1 = pm_fma_d(x2, sg * -0x1.ffffffd0c621cp-2, sg);
    double x6 = x4 * x2;
    double c = pm_fma_d(x4, sg * 0x1.55553e1068f19p-5, c1);
    return (float) pm_fma_d(x6, c2, c);
}

// s_sinf.c + s_cosf.c for the same argument (the reference's scalar sincos is std::sin and std::cos; glibc's sincosf returns the
// same two values).  |x| < 0.75: no reduction; |x| < 120: reduce_fast (n = round(x * 2/pi) through a 2^24 scaling, x - n * pi/2 in
// one fma); beyond that glibc reduces with 192 bits of 4/pi -- no call site comes near (they pass 2*pi*u or a concentric-disk
// angle) and the correctly rounded routine answers instead.
PM_HD void pm_sincos(float x, float *s_out, float *c_out) {
#if defined(PM_USE_LIBM) && !defined(__HIPCC__)
    *s_out = sinf(x); *c_out = cosf(x); return;   // oracle/Makefile, liboracle_libm.so: glibc itself, the pin of this header
#endif
#if defined(PM_CORRECTLY_ROUNDED)
    pm_sincos_cr(x, s_out, c_out); return;
#endif
    uint32_t top = (pm_bits(x) >> 20) & 0x7ffu;
    double xd = (double) x;
    if (top < 0x3f4u) {                                   // |x| < 0.75 (abstop12(pio4f))
        if (top < 0x398u) { *s_out = x; *c_out = 1.0f; return; }     // |x| < 2^-12
        double x2 = xd * xd;
        *s_out = pm_sinf_poly(xd, x2, false, 0);
        *c_out = pm_sinf_poly(xd, x2, false, 1);
        return;
    }
    if (top >= 0x42fu) { pm_sincos_cr(x, s_out, c_out); return; }    // |x| >= 120 (inf / NaN give NaN there too)
    double r = xd * 0x1.45f306dc9c883p+23;
    int n = ((int32_t) r + 0x800000) >> 24;
    xd = pm_fma_d(-(double) n, 0x1.921fb54442d18p+0, xd);
    double sg = ((n + 1) & 2) ? -1.0 : 1.0;               // sign[n & 3] = {1, -1, -1, 1}
    double x2 = xd * xd;
    *s_out = pm_sinf_poly(xd * sg, x2, (n & 2) != 0, n);
    *c_out = pm_sinf_poly(xd * sg, x2, (n & 2) != 0, n ^ 1);
}

// s_cbrtf.c (glibc <= 2.40): frexp, a quadratic seed, one Halley step in double, the factor 2^((e mod 3) / 3), ldexp.
// Sign-preserving; used by the Rayleigh phase function (/root/reference/src/phase/rayleigh.cpp:51-55).
PM_HD float pm_cbrt(float x) {
#if defined(PM_USE_LIBM) && !defined(__HIPCC__)
    return cbrtf(x);   // oracle/Makefile, liboracle_libm.so: glibc itself, the pin of this header
#endif
#if defined(PM_CORRECTLY_ROUNDED)
    return pm_cbrt_cr(x);
#endif
    uint32_t ix = pm_bits(x);
    uint32_t sign = ix & 0x80000000u;
    uint32_t ax = ix & 0x7fffffffu;
    if (ax >= 0x7f800000u) return x;              // inf / NaN
    if (ax < 0x00800000u) return pm_from_bits(sign);   // +-0, denormals (DAZ)
    int xe = (int) (ax >> 23) - 126;              // frexpf: |x| = xm 2^xe, xm in [0.5, 1)
    float xm = pm_from_bits((ax & 0x007fffffu) | 0x3f000000u);
    float u = (float) (0.492659620528969547 + (0.697570460207922770 - 0.191502161678719066 * (double) xm) * (double) xm);
    float t2 = u * u * u;
    int rem = xe % 3, q = xe / 3;                 // C semantics: truncation towards zero, rem in -2..2
    double f = rem == 0 ? 1.0 : rem == 1 ? 1.2599210498948731648 : rem == 2 ? 1.5874010519681994748
             : rem == -1 ? 1.0 / 1.2599210498948731648 : 1.0 / 1.5874010519681994748;
    float ym = (float) ((double) u * ((double) t2 + 2.0 * (double) xm) / (2.0 * (double) t2 + (double) xm) * f);
    float y = ym * pm_from_bits((uint32_t) (q + 127) << 23);     // ldexpf: exact, the result is normal
    return pm_from_bits(pm_bits(y) | sign);
}

// x^y for x >= 0 (std::pow semantics for the cases the RPV model can produce; /root/reference/src/bsdfs/rpv.cpp:85-167).
// e_powf.c: log2_inline (the table of pm_log with log2 c, a degree-5 polynomial) and exp2_inline in fp64.
PM_HD float pm_pow(float x, float y) {
#if defined(PM_USE_LIBM) && !defined(__HIPCC__)
    return powf(x, y);   // oracle/Makefile, liboracle_libm.so: glibc itself, the pin of this header
#endif
#if defined(PM_CORRECTLY_ROUNDED)
    return pm_pow_cr(x, y);
#endif
    if (y == 0.0f) return 1.0f;
    if (!(x == x) || !(y == y)) return pm_nan();
    if (x < 0.0f) return pm_nan();
    uint32_t ix = pm_bits(x);
    if (ix < 0x00800000u)                               // +0 / denormal
        return y > 0.0f ? 0.0f : pm_inf();
    if (ix == 0x7f800000u) return y > 0.0f ? pm_inf() : 0.0f;
    if (ix == 0x3f800000u) return 1.0f;
    uint32_t tmp = ix - 0x3f330000u;
    uint32_t i = (tmp >> 19) & 15u;
    uint32_t top = tmp & 0xff800000u;
    int k = (int32_t) top >> 23;
    double z = (double) pm_from_bits(ix - top);
    double r = pm_fma_d(z, PM_LOGF_TAB[2 * i], -1.0);
    double y0 = PM_LOG2F_TAB[i] + (double) k;
    double r2 = r * r;
    double l = pm_fma_d(0x1.27616c9496e0bp-2, r, -0x1.71969a075c67ap-2);
    double p = pm_fma_d(0x1.ec70a6ca7baddp-2, r, -0x1.7154748bef6c8p-1);
    double r4 = r2 * r2;
    double q = pm_fma_d(0x1.71547652ab82bp0, r, y0);
    q = pm_fma_d(p, r2, q);
    l = pm_fma_d(l, r4, q);
    double t = (double) y * l;                          // y log2 x
    if (t > 0x1.fffffffd1d571p+6) return pm_inf();      // e_powf.c's overflow bound (|y| = inf ends here or below as well)
    if (!(t > -150.0)) return 0.0f;                     // e_powf.c's underflow bound; denormal results are flushed below
    double kd = t + 0x1.8p+47;
    uint64_t ki = pm_bits_d(kd);
    kd = kd - 0x1.8p+47;
    double w = pm_exp2_tail(ki, t - kd, 0x1.c6af84b912394p-5, 0x1.ebfce50fac4f3p-3, 0x1.62e42ff0c52d6p-1);
    return w < 1.17549435082228750797e-38 ? 0.0f : (float) w;
}
