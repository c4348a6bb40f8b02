// oracle_math.h -- TEST INFRASTRUCTURE, not product code.
//
// Scalar fp32 helpers of the CPU restatement (the "oracle") of the reference's scalar_rgb
// semantics.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
// anything under oracle/.  Each function cites the reference file:line it follows
// (paths relative to /root/reference).
//
// Decisions where the reference is ambiguous at the bit level (enoki is absent):
//   * rcp(x) = 1/x, rsqrt(x) = 1/sqrt(x) (enoki's SSE paths use rcpps/rsqrtps + Newton steps whose bits are hardware
//     dependent); array / scalar = array * (1 / scalar), as enoki's operator/ routes a division by a lower-depth operand
//     ("reciprocal, then multiply", enoki array_router.h -- restated from the published source, the submodule is absent);
//     array / array = component-wise true division;
//   * dot(a,b) = fma chain a.x*b.x -> fma(a.y,b.y,.) -> fma(a.z,b.z,.) (enoki generic dot_);
//   * fmadd/fmsub/fnmadd in the reference source are fused; plain `a*b+c` is not;
//   * transcendental functions come from csrc/pmath.h (shared, bit-identical host/device).
#pragma once
#include <stdint.h>
#include "../eradiate-kernel_amd/csrc/pmath.h"

namespace orc {

struct V3 { float x, y, z; };
struct P2 { float x, y; };

static inline V3 v3(float x, float y, float z) { V3 r = { x, y, z }; return r; }
static inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
static inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline V3 operator*(float s, V3 a) { return v3(a.x * s, a.y * s, a.z * s); }
static inline V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline V3 operator/(V3 a, float s) { float r = 1.0f / s; return v3(a.x * r, a.y * r, a.z * r); }   // enoki: reciprocal, then multiply
static inline V3 operator/(V3 a, V3 b) { return v3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline float dot(V3 a, V3 b) { return pm_fma(a.z, b.z, pm_fma(a.y, b.y, a.x * b.x)); }
static inline float squared_norm(V3 a) { return dot(a, a); }
static inline float norm(V3 a) { return pm_sqrt(squared_norm(a)); }
static inline V3 normalize(V3 a) { return a * pm_rsqrt(squared_norm(a)); }
// enoki cross(): fmsub(a.yzx, b.zxy, a.zxy * b.yzx)
static inline V3 cross(V3 a, V3 b) {
    return v3(pm_fma(a.y, b.z, -(a.z * b.y)),
              pm_fma(a.z, b.x, -(a.x * b.z)),
              pm_fma(a.x, b.y, -(a.y * b.x)));
}
static inline V3 fmadd(V3 a, float s, V3 c) { return v3(pm_fma(a.x, s, c.x), pm_fma(a.y, s, c.y), pm_fma(a.z, s, c.z)); }
static inline V3 fnmadd(V3 a, float s, V3 c) { return v3(pm_fma(-a.x, s, c.x), pm_fma(-a.y, s, c.y), pm_fma(-a.z, s, c.z)); }
static inline float hmax(V3 a) { return pm_max(pm_max(a.x, a.y), a.z); }
static inline float hmin(V3 a) { return pm_min(pm_min(a.x, a.y), a.z); }
static inline float hmax_abs(V3 a) { return pm_max(pm_max(pm_abs(a.x), pm_abs(a.y)), pm_abs(a.z)); }
static inline V3 vrcp(V3 a) { return v3(1.0f / a.x, 1.0f / a.y, 1.0f / a.z); }
static inline float idx(V3 a, uint32_t i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }
static inline bool any_nonzero(V3 a) { return a.x != 0.f || a.y != 0.f || a.z != 0.f; }

// ---- The spectrum type of the variant this library is compiled for (oracle/Makefile): liboracle.so = rgb / mono, three channels
// (Color3f; Spec is V3 itself), liboracle_spectral.so (-DMTS_SPEC_N=4) = the semantics of scalar_spectral, Spectrum<Float, 4>
// (core/spectrum.h:57-73): four wavelengths, .x the one index_spectrum follows outside the rgb variants (volpath.cpp:26-36).
// The wavelengths of the sample in flight (ray.wavelengths / si.wavelengths in the reference) live in a thread-local: every
// oracle thread renders one sample at a time, and it spares each texture evaluation an argument.
#ifndef MTS_SPEC_N
#define MTS_SPEC_N 3
#endif
#if MTS_SPEC_N == 3
typedef V3 Spec;
static inline Spec spec_s(float v) { return v3(v, v, v); }
static inline float spec_hmean(Spec a) { return ((a.x + a.y) + a.z) * (1.f / 3.f); }
#else
struct Spec { float x, y, z, w; };
static inline Spec spec4(float x, float y, float z, float w) { Spec r = { x, y, z, w }; return r; }
static inline Spec spec_s(float v) { return spec4(v, v, v, v); }
static inline Spec operator+(Spec a, Spec b) { return spec4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
static inline Spec operator-(Spec a, Spec b) { return spec4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
static inline Spec operator*(Spec a, float s) { return spec4(a.x * s, a.y * s, a.z * s, a.w * s); }
static inline Spec operator*(float s, Spec a) { return spec4(a.x * s, a.y * s, a.z * s, a.w * s); }
static inline Spec operator*(Spec a, Spec b) { return spec4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
static inline Spec operator/(Spec a, float s) { float r = 1.0f / s; return spec4(a.x * r, a.y * r, a.z * r, a.w * r); }   // as for V3: reciprocal, then multiply
static inline Spec operator/(Spec a, Spec b) { return spec4(a.x / b.x, a.y / b.y, a.z / b.z, a.w / b.w); }
static inline float hmax(Spec a) { return pm_max(pm_max(a.x, a.y), pm_max(a.z, a.w)); }
static inline float idx(Spec a, uint32_t) { return a.x; }
static inline bool any_nonzero(Spec a) { return a.x != 0.f || a.y != 0.f || a.z != 0.f || a.w != 0.f; }
static inline float spec_hmean(Spec a) { return ((a.x + a.y) + (a.z + a.w)) * 0.25f; }    // enoki hmean of a 4-array: pairwise sum (enoki absent: decision)
static thread_local Spec tls_wavelengths = { 0.f, 0.f, 0.f, 0.f };
#endif

// math constants, include/mitsuba/core/math.h:13-38
static const float Pi = 3.14159265358979323846f;
static const float InvPi = 0.31830988618379067154f;
static const float InvTwoPi = 0.15915494309189533577f;
static const float InvFourPi = 0.07957747154594766788f;
static const float TwoPi = 6.28318530717958647692f;
static const float Epsilon = 1.1920929e-07f / 2;          // std::numeric_limits<float>::epsilon()/2
static const float RayEpsilon = Epsilon * 1500;
static const float ShadowEpsilon = RayEpsilon * 10;

// include/mitsuba/core/vector.h:116-136 (Duff et al. orthonormal basis)
static inline void coordinate_system(V3 n, V3 *s, V3 *t) {
    float sign = pm_sign(n.z), a = -pm_rcp(sign + n.z), b = n.x * n.y * a;
    *s = v3(pm_mulsign(n.x * n.x * a, n.z) + 1.f, pm_mulsign(b, n.z), pm_mulsign_neg(n.x, n.z));
    *t = v3(b, sign + n.y * n.y * a, -n.y);
}

// include/mitsuba/core/frame.h:17-37
struct Frame {
    V3 s, t, n;
    V3 to_local(V3 v) const { return v3(dot(v, s), dot(v, t), dot(v, n)); }
    V3 to_world(V3 v) const { return s * v.x + t * v.y + n * v.z; }
};
static inline Frame frame_from_normal(V3 n) { Frame f; f.n = n; coordinate_system(n, &f.s, &f.t); return f; }

// include/mitsuba/core/ray.h:30-65
struct Ray {
    V3 o, d, d_rcp;
    float mint, maxt;
    V3 operator()(float t) const { return fmadd(d, t, o); }
};
static inline Ray make_ray(V3 o, V3 d, float mint, float maxt) { Ray r; r.o = o; r.d = d; r.d_rcp = vrcp(d); r.mint = mint; r.maxt = maxt; return r; }

// include/mitsuba/core/transform.h:36-160. Row-major storage: m[r*4+c]; enoki's coeff(i) = column i.
struct Xf { float m[16]; float it[16]; };
static inline Xf xf_identity() { Xf x = {}; for (int i = 0; i < 4; ++i) x.m[i * 5] = x.it[i * 5] = 1.f; return x; }
static inline void mat_transpose(const float *a, float *o) { for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) o[r * 4 + c] = a[c * 4 + r]; }
// transform.h:59-61: inverse = (transpose(inverse_transpose), transpose(matrix))
static inline Xf xf_inverse(const Xf &x) { Xf r; mat_transpose(x.it, r.m); mat_transpose(x.m, r.it); return r; }
// enoki matrix product: column j of the result = sum_k A.col(k) * B(k, j), fmadd chain
static inline void mat_mul(const float *a, const float *b, float *o) {
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 4; ++r) {
            float acc = a[r * 4 + 0] * b[0 * 4 + j];
            for (int k = 1; k < 4; ++k) acc = pm_fma(a[r * 4 + k], b[k * 4 + j], acc);
            o[r * 4 + j] = acc;
        }
}
static inline Xf xf_mul(const Xf &a, const Xf &b) { Xf r; mat_mul(a.m, b.m, r.m); mat_mul(a.it, b.it, r.it); return r; }   // transform.h:53-56
// transform.h:90-97 transform_affine(Point)
static inline V3 xf_point_affine(const Xf &x, V3 p) {
    const float *m = x.m;
    return v3(pm_fma(m[2], p.z, pm_fma(m[1], p.y, pm_fma(m[0], p.x, m[3]))),
              pm_fma(m[6], p.z, pm_fma(m[5], p.y, pm_fma(m[4], p.x, m[7]))),
              pm_fma(m[10], p.z, pm_fma(m[9], p.y, pm_fma(m[8], p.x, m[11]))));
}
// transform.h:103-111 operator*(Point): homogeneous divide
static inline V3 xf_point(const Xf &x, V3 p) {
    const float *m = x.m;
    float w = pm_fma(m[14], p.z, pm_fma(m[13], p.y, pm_fma(m[12], p.x, m[15])));
    V3 r = xf_point_affine(x, p);
    return v3(r.x / w, r.y / w, r.z / w);
}
// transform.h:117-126 operator*(Vector)
static inline V3 mat_vector(const float *m, V3 v) {
    return v3(pm_fma(m[2], v.z, pm_fma(m[1], v.y, m[0] * v.x)),
              pm_fma(m[6], v.z, pm_fma(m[5], v.y, m[4] * v.x)),
              pm_fma(m[10], v.z, pm_fma(m[9], v.y, m[8] * v.x)));
}
static inline V3 xf_vector(const Xf &x, V3 v) { return mat_vector(x.m, v); }
static inline V3 xf_normal(const Xf &x, V3 n) { return mat_vector(x.it, n); }   // transform.h:132-141
static inline V3 xf_translation(const Xf &x) { return v3(x.m[3], x.m[7], x.m[11]); }   // transform.h:64-66
static inline Ray xf_ray_affine(const Xf &x, const Ray &r) { return make_ray(xf_point_affine(x, r.o), xf_vector(x, r.d), r.mint, r.maxt); }   // transform.h:152-157
static inline Xf xf_scale(V3 s) { Xf x = xf_identity(); x.m[0] = s.x; x.m[5] = s.y; x.m[10] = s.z; x.it[0] = 1.f / s.x; x.it[5] = 1.f / s.y; x.it[10] = 1.f / s.z; return x; }   // transform.h:166-170
static inline Xf xf_translate(V3 t) { Xf x = xf_identity(); x.m[3] = t.x; x.m[7] = t.y; x.m[11] = t.z; x.it[12] = -t.x; x.it[13] = -t.y; x.it[14] = -t.z; return x; }   // transform.h:160-163

// include/mitsuba/core/bbox.h:302-325 (ray interval ignored)
struct BBox { V3 min, max; };
static inline BBox bbox_empty() { BBox b; b.min = v3(pm_inf(), pm_inf(), pm_inf()); b.max = v3(-pm_inf(), -pm_inf(), -pm_inf()); return b; }
static inline void bbox_expand(BBox &b, V3 p) {
    b.min = v3(pm_min(b.min.x, p.x), pm_min(b.min.y, p.y), pm_min(b.min.z, p.z));
    b.max = v3(pm_max(b.max.x, p.x), pm_max(b.max.y, p.y), pm_max(b.max.z, p.z));
}
static inline void bbox_expand(BBox &b, const BBox &o) { bbox_expand(b, o.min); bbox_expand(b, o.max); }
static inline bool bbox_valid(const BBox &b) { return b.max.x >= b.min.x && b.max.y >= b.min.y && b.max.z >= b.min.z; }
static inline bool bbox_ray_intersect(const BBox &b, const Ray &ray, float *mint, float *maxt) {
    bool active = (ray.d.x != 0.f || (ray.o.x > b.min.x || ray.o.x < b.max.x)) &&
                  (ray.d.y != 0.f || (ray.o.y > b.min.y || ray.o.y < b.max.y)) &&
                  (ray.d.z != 0.f || (ray.o.z > b.min.z || ray.o.z < b.max.z));
    V3 t1 = (b.min - ray.o) * ray.d_rcp, t2 = (b.max - ray.o) * ray.d_rcp;
    V3 t1p = v3(pm_min(t1.x, t2.x), pm_min(t1.y, t2.y), pm_min(t1.z, t2.z));
    V3 t2p = v3(pm_max(t1.x, t2.x), pm_max(t1.y, t2.y), pm_max(t1.z, t2.z));
    *mint = hmax(t1p);
    *maxt = hmin(t2p);
    return active && *maxt >= *mint;
}

// enoki::PCG32 (absent source, public algorithm pcg32 by M. O'Neill; constants quoted in
// SURVEY.md 8(a) row a5; used via include/mitsuba/core/random.h:52-54)
static const uint64_t PCG32_DEFAULT_STATE = 0x853c49e6748fea9bULL;
static const uint64_t PCG32_DEFAULT_STREAM = 0xda3e39cb94b95bdbULL;
static const uint64_t PCG32_MULT = 0x5851f42d4c957f2dULL;
struct PCG32 {
    uint64_t state, inc;
    uint32_t next_uint32() {
        uint64_t old = state;
        state = old * PCG32_MULT + inc;
        uint32_t xorshifted = (uint32_t) (((old >> 18u) ^ old) >> 27u);
        uint32_t rot = (uint32_t) (old >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31u));
    }
    void seed(uint64_t initstate, uint64_t initseq) {
        state = 0u; inc = (initseq << 1u) | 1u;
        next_uint32(); state += initstate; next_uint32();
    }
    float next_float32() { return pm_from_bits((next_uint32() >> 9) | 0x3f800000u) - 1.0f; }
};

// include/mitsuba/core/random.h:75-85,106-116,137-140
static inline void tea_rounds(uint32_t &v0, uint32_t &v1, int rounds) {
    uint32_t sum = 0;
    for (int i = 0; i < rounds; ++i) {
        sum += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eu);
    }
}
static inline uint32_t sample_tea_32(uint32_t v0, uint32_t v1, int rounds = 4) { tea_rounds(v0, v1, rounds); return v1; }
static inline uint64_t sample_tea_64(uint32_t v0, uint32_t v1, int rounds = 4) { tea_rounds(v0, v1, rounds); return (uint64_t) v0 + ((uint64_t) v1 << 32); }
static inline float sample_tea_float32(uint32_t v0, uint32_t v1, int rounds = 4) { return pm_from_bits((sample_tea_32(v0, v1, rounds) >> 9) | 0x3f800000u) - 1.0f; }
// The same template instantiated with 64-bit arrays, as PCG32Sampler::seed of the wavefront variants calls it (librender/sampler.cpp:
// 89-92: sample_tea_64(UInt64(seed_value), idx) with idx = arange<UInt64>): every operation of core/random.h:106-116 then runs in
// 64-bit arithmetic -- no wrap at 2^32 inside the rounds -- and the result is v0 + (v1 << 32) modulo 2^64.
static inline uint64_t sample_tea_64_u64(uint64_t v0, uint64_t v1, int rounds = 4) {
    uint64_t sum = 0;
    for (int i = 0; i < rounds; ++i) {
        sum += 0x9e3779b9ull;
        v0 += ((v1 << 4) + 0xa341316cull) ^ (v1 + sum) ^ ((v1 >> 5) + 0xc8013ea4ull);
        v1 += ((v0 << 4) + 0xad90777dull) ^ (v0 + sum) ^ ((v0 >> 5) + 0x7e95761eull);
    }
    return v0 + (v1 << 32);
}

// include/mitsuba/core/warp.h:23,54-90,255-260,287-301,325-333
static inline float circ(float x) { return pm_safe_sqrt(pm_fma(-x, x, 1.f)); }
static inline P2 square_to_uniform_disk_concentric(P2 sample) {
    float x = pm_fma(2.f, sample.x, -1.f), y = pm_fma(2.f, sample.y, -1.f);
    bool is_zero = x == 0.f && y == 0.f, quadrant_1_or_3 = pm_abs(x) < pm_abs(y);
    float r = quadrant_1_or_3 ? y : x, rp = quadrant_1_or_3 ? x : y;
    float phi = .25f * Pi * rp / r;
    if (quadrant_1_or_3) phi = .5f * Pi - phi;
    if (is_zero) phi = 0.f;
    float s, c; pm_sincos(phi, &s, &c);
    P2 p = { r * c, r * s };
    return p;
}
static inline V3 square_to_uniform_sphere(P2 sample) {
    float z = pm_fma(-2.f, sample.y, 1.f), r = circ(z);
    float s, c; pm_sincos(2.f * Pi * sample.x, &s, &c);
    return v3(r * c, r * s, z);
}
static inline V3 square_to_uniform_hemisphere(P2 sample) {
    P2 p = square_to_uniform_disk_concentric(sample);
    float z = 1.f - pm_fma(p.y, p.y, p.x * p.x);
    float k = pm_sqrt(z + 1.f);
    return v3(p.x * k, p.y * k, z);
}
static inline V3 square_to_cosine_hemisphere(P2 sample) {
    P2 p = square_to_uniform_disk_concentric(sample);
    float z = pm_safe_sqrt(1.f - pm_fma(p.y, p.y, p.x * p.x));
    return v3(p.x, p.y, z);
}

// enoki::morton_decode<Point2u>(i) (absent source): de-interleave even bits -> x, odd bits -> y
// (used at src/librender/integrator.cpp:200)
static inline uint32_t compact_bits(uint32_t x) {
    x &= 0x55555555u; x = (x ^ (x >> 1)) & 0x33333333u; x = (x ^ (x >> 2)) & 0x0f0f0f0fu;
    x = (x ^ (x >> 4)) & 0x00ff00ffu; x = (x ^ (x >> 8)) & 0x0000ffffu; return x;
}
static inline void morton_decode(uint32_t i, uint32_t *x, uint32_t *y) { *x = compact_bits(i); *y = compact_bits(i >> 1); }

} // namespace orc
