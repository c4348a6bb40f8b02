// dmath.h -- small fp32 vector / transform toolkit used by the host-side plugin constructors and the
// HIP kernels.  Every operation is an explicit IEEE add / mul / fma / div / sqrt (see pmath.h), so host
// and device evaluate identical bits.  Conventions for operations the reference leaves to enoki
// (absent): dot = fma chain, cross = fmsub form, rcp = 1/x, normalize = v * (1/sqrt(|v|^2)).
// Citations are relative to /root/reference.
#pragma once
#include "pmath.h"

#if defined(__HIPCC__)
#  define DM_HD __host__ __device__ __forceinline__
#else
#  define DM_HD static inline
#endif

namespace mtsamd {

struct F3 { float x, y, z; };
struct F2 { float x, y; };

DM_HD F3 f3(float x, float y, float z) { F3 r; r.x = x; r.y = y; r.z = z; return r; }
DM_HD F3 f3(const float *p) { F3 r; r.x = p[0]; r.y = p[1]; r.z = p[2]; return r; }
DM_HD F3 f3(const __attribute__((address_space(1))) float *p) { F3 r; r.x = p[0]; r.y = p[1]; r.z = p[2]; return r; }
DM_HD F3 f3s(float s) { F3 r; r.x = s; r.y = s; r.z = s; return r; }
DM_HD F3 operator+(F3 a, F3 b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
DM_HD F3 operator-(F3 a, F3 b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
DM_HD F3 operator-(F3 a) { return f3(-a.x, -a.y, -a.z); }
DM_HD F3 operator*(F3 a, float s) { return f3(a.x * s, a.y * s, a.z * s); }
DM_HD F3 operator*(float s, F3 a) { return f3(a.x * s, a.y * s, a.z * s); }
DM_HD F3 operator*(F3 a, F3 b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
DM_HD F3 operator/(F3 a, float s) { const float r = pm_rcp(s); return f3(a.x * r, a.y * r, a.z * r); }   // enoki: array / scalar = reciprocal, then multiply
DM_HD F3 operator/(F3 a, F3 b) { return f3(a.x / b.x, a.y / b.y, a.z / b.z); }
DM_HD float dot(F3 a, F3 b) { return pm_fma(a.z, b.z, pm_fma(a.y, b.y, a.x * b.x)); }
DM_HD float squared_norm(F3 a) { return dot(a, a); }
DM_HD float norm(F3 a) { return pm_sqrt(squared_norm(a)); }
DM_HD F3 normalize(F3 a) { return a * pm_rsqrt(squared_norm(a)); }
DM_HD F3 cross(F3 a, F3 b) {
    return f3(pm_fma(a.y, b.z, -(a.z * b.y)), pm_fma(a.z, b.x, -(a.x * b.z)), pm_fma(a.x, b.y, -(a.y * b.x)));
}
DM_HD F3 fmadd(F3 a, float s, F3 c) { return f3(pm_fma(a.x, s, c.x), pm_fma(a.y, s, c.y), pm_fma(a.z, s, c.z)); }
DM_HD F3 fnmadd(F3 a, float s, F3 c) { return f3(pm_fma(-a.x, s, c.x), pm_fma(-a.y, s, c.y), pm_fma(-a.z, s, c.z)); }
DM_HD float hmax(F3 a) { return pm_max(pm_max(a.x, a.y), a.z); }
DM_HD float hmin(F3 a) { return pm_min(pm_min(a.x, a.y), a.z); }
DM_HD float hmax_abs(F3 a) { return pm_max(pm_max(pm_abs(a.x), pm_abs(a.y)), pm_abs(a.z)); }
DM_HD F3 vrcp(F3 a) { return f3(pm_rcp(a.x), pm_rcp(a.y), pm_rcp(a.z)); }
DM_HD float pick(F3 a, uint32_t i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
DM_HD bool any_nonzero(F3 a) { return a.x != 0.f || a.y != 0.f || a.z != 0.f; }

// ---- The spectrum type of the variant this translation unit is compiled for.  rgb / mono: three channels (Color3f) -- a typedef of
// F3, so the rgb kernels are compiled from the very same expressions as before.  Spectral (MTS_SPEC_N == 4): four wavelengths
// (Spectrum<Float, 4>, core/spectrum.h:57-73); .x is the one the free-flight sampling follows (volpath.cpp:26-36: index_spectrum
// returns spec[0] outside the rgb variants).  The inline namespace keeps the two builds' symbols apart inside one library.
#ifndef MTS_SPEC_N
#define MTS_SPEC_N 3
#endif
#if MTS_SPEC_N == 3
inline namespace v_rgb {
typedef F3 Spec;
DM_HD Spec spec_s(float s) { return f3s(s); }
DM_HD float spec_hmean(Spec a) { return ((a.x + a.y) + a.z) * (1.f / 3.f); }
}
#else
inline namespace v_spectral {
struct Spec { float x, y, z, w; };
DM_HD Spec spec4(float x, float y, float z, float w) { Spec r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
DM_HD Spec spec_s(float s) { return spec4(s, s, s, s); }
DM_HD Spec operator+(Spec a, Spec b) { return spec4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
DM_HD Spec operator-(Spec a, Spec b) { return spec4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
DM_HD Spec operator*(Spec a, float s) { return spec4(a.x * s, a.y * s, a.z * s, a.w * s); }
DM_HD Spec operator*(float s, Spec a) { return spec4(a.x * s, a.y * s, a.z * s, a.w * s); }
DM_HD Spec operator*(Spec a, Spec b) { return spec4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
DM_HD Spec operator/(Spec a, float s) { const float r = pm_rcp(s); return spec4(a.x * r, a.y * r, a.z * r, a.w * r); }   // as for F3: reciprocal, then multiply
DM_HD Spec operator/(Spec a, Spec b) { return spec4(a.x / b.x, a.y / b.y, a.z / b.z, a.w / b.w); }
DM_HD float hmax(Spec a) { return pm_max(pm_max(a.x, a.y), pm_max(a.z, a.w)); }
DM_HD float pick(Spec a, uint32_t) { return a.x; }                       // index_spectrum, volpath.cpp:26-36
DM_HD bool any_nonzero(Spec a) { return a.x != 0.f || a.y != 0.f || a.z != 0.f || a.w != 0.f; }
DM_HD float spec_hmean(Spec a) { return ((a.x + a.y) + (a.z + a.w)) * 0.25f; }      // enoki hmean of a 4-array: pairwise sum (enoki absent: decision)
}
#endif

// core/math.h:13-38
#define MTS_PI 3.14159265358979323846f
#define MTS_INV_PI 0.31830988618379067154f
#define MTS_INV_TWO_PI 0.15915494309189533577f
#define MTS_INV_FOUR_PI 0.07957747154594766788f
#define MTS_TWO_PI 6.28318530717958647692f
#define MTS_EPSILON (1.1920929e-07f / 2)
#define MTS_RAY_EPSILON (MTS_EPSILON * 1500)
#define MTS_SHADOW_EPSILON (MTS_RAY_EPSILON * 10)

// core/vector.h:116-136
DM_HD void coordinate_system(F3 n, F3 &s, F3 &t) {
    float sign = pm_sign(n.z), a = -pm_rcp(sign + n.z), b = n.x * n.y * a;
    s = f3(pm_mulsign(n.x * n.x * a, n.z) + 1.f, pm_mulsign(b, n.z), pm_mulsign_neg(n.x, n.z));
    t = f3(b, sign + n.y * n.y * a, -n.y);
}

// core/frame.h:17-37
struct Frame3 { F3 s, t, n; };
DM_HD Frame3 make_frame(F3 n) { Frame3 f; f.n = n; coordinate_system(n, f.s, f.t); return f; }
DM_HD F3 to_local(const Frame3 &f, F3 v) { return f3(dot(v, f.s), dot(v, f.t), dot(v, f.n)); }
DM_HD F3 to_world(const Frame3 &f, F3 v) { return f.s * v.x + f.t * v.y + f.n * v.z; }

// core/transform.h:90-141; matrices are row-major float[16]
DM_HD F3 mat_point_affine(const float *m, F3 p) {
    return f3(pm_fma(m[2], p.z, pm_fma(m[1], p.y, pm_fma(m[0], p.x, m[3]))),
              pm_fma(m[6], p.z, pm_fma(m[5], p.y, pm_fma(m[4], p.x, m[7]))),
              pm_fma(m[10], p.z, pm_fma(m[9], p.y, pm_fma(m[8], p.x, m[11]))));
}
DM_HD F3 mat_point(const float *m, F3 p) {
    float w = pm_fma(m[14], p.z, pm_fma(m[13], p.y, pm_fma(m[12], p.x, m[15])));
    F3 r = mat_point_affine(m, p);
    return f3(r.x / w, r.y / w, r.z / w);
}
DM_HD F3 mat_vector(const float *m, F3 v) {
    return f3(pm_fma(m[2], v.z, pm_fma(m[1], v.y, m[0] * v.x)),
              pm_fma(m[6], v.z, pm_fma(m[5], v.y, m[4] * v.x)),
              pm_fma(m[10], v.z, pm_fma(m[9], v.y, m[8] * v.x)));
}

// core/warp.h:23,54-90,255-260,287-301,325-333
DM_HD float circ(float x) { return pm_safe_sqrt(pm_fma(-x, x, 1.f)); }
DM_HD F2 square_to_uniform_disk_concentric(F2 sample) {
    float x = pm_fma(2.f, sample.x, -1.f), y = pm_fma(2.f, sample.y, -1.f);
    bool is_zero = x == 0.f && y == 0.f, quadrant_1_or_3 = pm_abs(x) < pm_abs(y);
    float r = quadrant_1_or_3 ? y : x, rp = quadrant_1_or_3 ? x : y;
    float phi = .25f * MTS_PI * rp / r;
    if (quadrant_1_or_3) phi = .5f * MTS_PI - phi;
    if (is_zero) phi = 0.f;
    float s, c; pm_sincos(phi, &s, &c);
    F2 p; p.x = r * c; p.y = r * s;
    return p;
}
DM_HD F3 square_to_uniform_sphere(F2 sample) {
    float z = pm_fma(-2.f, sample.y, 1.f), r = circ(z);
    float s, c; pm_sincos(2.f * MTS_PI * sample.x, &s, &c);
    return f3(r * c, r * s, z);
}
DM_HD F3 square_to_uniform_hemisphere(F2 sample) {
    F2 p = square_to_uniform_disk_concentric(sample);
    float z = 1.f - pm_fma(p.y, p.y, p.x * p.x);
    float k = pm_sqrt(z + 1.f);
    return f3(p.x * k, p.y * k, z);
}
DM_HD F3 square_to_cosine_hemisphere(F2 sample) {
    F2 p = square_to_uniform_disk_concentric(sample);
    float z = pm_safe_sqrt(1.f - pm_fma(p.y, p.y, p.x * p.x));
    return f3(p.x, p.y, z);
}

// PCG32 (public algorithm; enoki::PCG32 via core/random.h:52-54) with the default stream
#define PCG32_DEFAULT_STATE 0x853c49e6748fea9bULL
#define PCG32_DEFAULT_STREAM 0xda3e39cb94b95bdbULL
#define PCG32_MULT 0x5851f42d4c957f2dULL
struct Pcg32 {
    uint64_t state, inc;
    DM_HD uint32_t next_uint32() {
        uint64_t old = state;
        state = old * PCG32_MULT + inc;
        uint32_t xorshifted = (uint32_t) (((old >> 18u) ^ old) >> 27u);
        uint32_t rot = (uint32_t) (old >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31u));
    }
    DM_HD void seed(uint64_t initstate, uint64_t initseq) {
        state = 0u; inc = (initseq << 1u) | 1u;
        next_uint32(); state += initstate; next_uint32();
    }
    DM_HD float next_1d() { return pm_from_bits((next_uint32() >> 9) | 0x3f800000u) - 1.0f; }
    DM_HD F2 next_2d() { F2 p; p.x = next_1d(); p.y = next_1d(); return p; }
};

// Tiny Encryption Algorithm, core/random.h:75-85 (sample_tea_32), :106-116 (sample_tea_64), :137-140 (sample_tea_float32): the
// reference's wavefront variants seed one PCG32 per lane with it (librender/sampler.cpp:89-92)
DM_HD void tea_rounds(uint32_t &v0, uint32_t &v1, int rounds) {
    uint32_t sum = 0;
    for (int i = 0; i < rounds; ++i) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eu);
    }
}
DM_HD uint32_t sample_tea_32(uint32_t v0, uint32_t v1, int rounds = 4) { tea_rounds(v0, v1, rounds); return v1; }
DM_HD uint64_t sample_tea_64(uint32_t v0, uint32_t v1, int rounds = 4) { tea_rounds(v0, v1, rounds); return (uint64_t) v0 + ((uint64_t) v1 << 32); }
DM_HD float sample_tea_float32(uint32_t v0, uint32_t v1, int rounds = 4) { return pm_from_bits((sample_tea_32(v0, v1, rounds) >> 9) | 0x3f800000u) - 1.0f; }
// The same template instantiated with 64-bit arrays, as PCG32Sampler::seed of the wavefront variants calls it (librender/sampler.cpp:
// 89-92: sample_tea_64(UInt64(seed_value), idx) with idx = arange<UInt64>): every operation of core/random.h:106-116 then runs in
// 64-bit arithmetic -- no wrap at 2^32 inside the rounds -- and the result is v0 + (v1 << 32) modulo 2^64.
DM_HD uint64_t sample_tea_64_u64(uint64_t v0, uint64_t v1, int rounds = 4) {
    uint64_t sum = 0;
    for (int i = 0; i < rounds; ++i) {
        sum += 0x9e3779b9ull;
        v0 += ((v1 << 4) + 0xa341316cull) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4ull);
        v1 += ((v0 << 4) + 0xad90777dull) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eull);
    }
    return v0 + (v1 << 32);
}

// enoki::morton_decode (librender/integrator.cpp:200)
DM_HD uint32_t compact_bits(uint32_t x) {
    x &= 0x55555555u; x = (x ^ (x >> 1)) & 0x33333333u; x = (x ^ (x >> 2)) & 0x0f0f0f0fu;
    x = (x ^ (x >> 4)) & 0x00ff00ffu; x = (x ^ (x >> 8)) & 0x0000ffffu; return x;
}

} // namespace mtsamd
