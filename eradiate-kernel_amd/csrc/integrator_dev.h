// integrator_dev.h -- device functions of the path / volpath hot path (gfx950).
//
// One lane owns one pixel and runs that pixel's samples sequentially with the reference's scalar
// PCG32 stream (librender/integrator.cpp:198), so every random draw below happens in exactly the
// order of the scalar_rgb variant (SURVEY.md section 8(a')).  All scene records are read through
// wave-uniform addresses (scalar loads); the only per-lane memory traffic is the volume gathers.
// Citations are relative to /root/reference.
#pragma once
#include <hip/hip_runtime.h>
#include "dscene.h"
#include "dmath.h"
#include "../../include/mtsamd.h"

// Scene traits (round 4).  The render kernels are written for every scene the loader accepts; what a given scene cannot contain still
// costs it: the out-of-line functions behind rarely-used features (a BVH, spheres, mesh emitters, nested blendphase trees, RPV, generic
// grid lookups) impose the calling convention on the whole kernel -- values that live across a call site sit in callee-saved registers or
// in scratch -- and every per-medium case distinction is a chain of scalar branches in the tracking step.  A translation unit compiled
// with MTS_TRAITS != 0 (kernels_lean_*.hip) PROMISES the properties below; the host checks them per scene (scene_host.cpp:
// scene_traits) and mts_render picks the leanest kernel whose promises the scene keeps.  Same source, same arithmetic: the promised-away
// branches are simply never compiled (C3 562 -> 629, C4 397 -> 469 Msamples/s on the 256-spp probes, profiles/r04_ab_experiments.log).
// (the MT_* bits themselves live in dscene.h: the host computes them per scene)
#ifndef MTS_TRAITS
#define MTS_TRAITS 0
#endif
#if MTS_SPEC_N == 3
#if !defined(MTS_VARIANT_NS)
#define MTS_VARIANT_NS v_rgb
#endif
#elif !defined(MTS_VARIANT_NS)
#define MTS_VARIANT_NS v_spectral
#endif
namespace mtsamd {
inline namespace MTS_VARIANT_NS {

#define DEV __device__ __forceinline__
#define DEV_NOINLINE __device__ __noinline__
// functions that are calls in the general kernels so that scenes which never reach them do not carry their registers, and inline in the
// lean translation units that keep them: a unit has few enough of them left, and a kernel WITHOUT any call does not pay the calling
// convention at all (EXP_LEAN_CALLS restores the calls: measurement)
#if defined(MTS_LEAN) && !defined(EXP_LEAN_CALLS)
#define DEV_CALL_UNLESS_LEAN DEV
#else
#define DEV_CALL_UNLESS_LEAN DEV_NOINLINE
#endif

// ---- scalar-load helpers -------------------------------------------------------------------------------
// Scene records live in global memory and are addressed with wave-uniform indices, but the compiler only
// issues scalar loads (s_load, results in SGPRs) for memory it knows to be invariant.  cload() reads a
// record through the constant address space -- same bytes, but invariant by definition -- so uniform
// records cost no VGPRs and no vector-memory instructions.  The scene is never written while a kernel runs.
#define MTS_CONST_AS __attribute__((address_space(4)))
#define MTS_GLOBAL_AS __attribute__((address_space(1)))
// Pointers read out of scene records are generic as far as the compiler can tell, and per-lane accesses through
// them become FLAT instructions (which also tick lgkmcnt and so serialise with LDS / scalar traffic).  Every such
// pointer is device memory: the round trip through the global address space lets InferAddressSpaces emit GLOBAL ones.
template <typename T> __device__ __forceinline__ MTS_GLOBAL_AS T *as_global(T *p) { return (MTS_GLOBAL_AS T *) p; }
template <typename T> DEV T cload(const T *p) {
    static_assert(sizeof(T) % 4 == 0, "records are dword multiples");
    uint32_t tmp[sizeof(T) / 4];
    const MTS_CONST_AS uint32_t *src = (const MTS_CONST_AS uint32_t *) (uintptr_t) p;
#pragma unroll
    for (int k = 0; k < (int) (sizeof(T) / 4); ++k) tmp[k] = src[k];
    T r; __builtin_memcpy(&r, tmp, sizeof(T));
    return r;
}
template <typename T> DEV T cload_k(const MTS_CONST_AS void *p) {      // same, from a constant-address-space pointer (kernel arguments)
    static_assert(sizeof(T) % 4 == 0, "records are dword multiples");
    uint32_t tmp[sizeof(T) / 4];
    const MTS_CONST_AS uint32_t *src = (const MTS_CONST_AS uint32_t *) p;
#pragma unroll
    for (int k = 0; k < (int) (sizeof(T) / 4); ++k) tmp[k] = src[k];
    T r; __builtin_memcpy(&r, tmp, sizeof(T));
    return r;
}
// Waterfall: `idx` may differ between lanes (a lane's current medium / shape) but rarely does.  Peel one
// distinct value per trip so that the record can be addressed with an SGPR.  The comparison goes through
// an opaque copy: otherwise the optimiser learns `idx == uni` inside the branch, substitutes the per-lane
// `idx` for the uniform `uni`, and every record load degrades to a vector load again.
#define WATERFALL_BEGIN(idx, uni)                                           \
    for (bool wf_pending_ = true; wf_pending_;) {                           \
        const int uni = __builtin_amdgcn_readfirstlane(idx);                \
        int wf_cmp_ = uni; asm volatile("" : "+s"(wf_cmp_));                \
        if ((idx) == wf_cmp_) {
#define WATERFALL_END                                                       \
            wf_pending_ = false;                                            \
        }                                                                   \
    }

struct DRay { F3 o, d, d_rcp; float mint, maxt; };
DEV DRay make_ray(F3 o, F3 d, float mint, float maxt) { DRay r; r.o = o; r.d = d; r.d_rcp = vrcp(d); r.mint = mint; r.maxt = maxt; return r; }
DEV F3 ray_at(const DRay &r, float t) { return fmadd(r.d, t, r.o); }                       // core/ray.h:65
DEV DRay spawn_ray(F3 p, F3 d) { return make_ray(p, d, (1.f + hmax_abs(p)) * MTS_RAY_EPSILON, pm_inf()); }   // render/interaction.h:58-61

#if defined(MTSAMD_BLOCKSTATS)
// diagnostic build: per-segment cycle accumulators of the MEDIUM block (clock64 = s_memtime, shader cycles)
struct Counters { uint32_t n_iter, n_lookup, n_nee_step; unsigned long long seg[8]; long long tmark; };
#define MTS_SEG_BEGIN(c) do { (c).tmark = clock64(); } while (0)
#define MTS_SEG(c, k) do { long long t_ = clock64(); (c).seg[k] += (unsigned long long) (t_ - (c).tmark); (c).tmark = t_; } while (0)
#else
struct Counters { uint32_t n_iter, n_lookup, n_nee_step; };
#define MTS_SEG_BEGIN(c) do { } while (0)
#define MTS_SEG(c, k) do { } while (0)
#endif

// Integrator::should_stop() (include/mitsuba/render/integrator.h:143-146) on the device: one lane of the wave reads the host-visible
// stop word (pinned host memory written by mts_cancel / the timeout watchdog), the answer is wave-uniform.
__device__ __forceinline__ bool stop_requested(const uint32_t *stop_flag) {
    uint32_t v = 0;
    const unsigned long long active = __builtin_amdgcn_ballot_w64(true);
    if ((threadIdx.x & 63u) == (uint32_t) __builtin_ctzll(active))                     // first active lane
        v = __hip_atomic_load(stop_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return __builtin_amdgcn_ballot_w64(v != 0u) != 0ull;
}

// The stream of sample j (counted from blk.sample_base) of pixel (lx, ly) in the wavefront (gpu_*) variants: PCG32Sampler::seed seeds lane
// L of the wavefront with (sample_tea_64(seed, L), sample_tea_64(L, seed)) in 64-bit arithmetic (librender/sampler.cpp:89-92), and
// SamplingIntegrator::render lays the lanes out as L = pixel * spp + sample, pixel = y * width + x inside the crop window
// (integrator.cpp:143-163)
DEV void seed_wavefront_sample(Pcg32 &rng, const DSensor &se, const DBlock &blk, uint32_t lx, uint32_t ly, uint32_t j) {
    const uint64_t pixel = (uint64_t) (uint32_t) (blk.oy + (int) ly - se.crop_y) * (uint64_t) (uint32_t) se.crop_w + (uint64_t) (uint32_t) (blk.ox + (int) lx - se.crop_x);
    const uint64_t L = pixel * (uint64_t) (uint32_t) se.sample_count + (uint64_t) blk.sample_base + (uint64_t) j;
    rng.seed(sample_tea_64_u64(se.seed, L), sample_tea_64_u64(L, se.seed));
}

// the increment of that generator alone (Pcg32::seed: inc = (initseq << 1) | 1): what the regrouping kernels recompute on every load
DEV uint64_t wavefront_increment(const DSensor &se, const DBlock &blk, uint32_t lx, uint32_t ly, uint32_t j) {
    const uint64_t pixel = (uint64_t) (uint32_t) (blk.oy + (int) ly - se.crop_y) * (uint64_t) (uint32_t) se.crop_w + (uint64_t) (uint32_t) (blk.ox + (int) lx - se.crop_x);
    const uint64_t L = pixel * (uint64_t) (uint32_t) se.sample_count + (uint64_t) blk.sample_base + (uint64_t) j;
    return (sample_tea_64_u64(L, se.seed) << 1u) | 1u;
}

// What survives of a SurfaceInteraction between loop iterations: the hit distance, the hit point
// (computed from the ray that found it), and the primitive; normals / frames / wi are rebuilt on
// demand by complete_surface() with the same arithmetic.
struct Hit { float t; F3 p; F2 uv; int32_t shape, prim; };
struct Surf { F3 n; Frame3 sh; F3 wi; };
DEV bool hit_valid(const Hit &h) { return h.t != pm_inf(); }

// core/bbox.h:302-325
DEV bool bbox_ray_intersect(const DBBox &b, const DRay &ray, float &mint, float &maxt) {
    F3 bmin = f3(b.min), bmax = f3(b.max);
    bool active = (ray.d.x != 0.f || (ray.o.x > bmin.x || ray.o.x < bmax.x)) &&
                  (ray.d.y != 0.f || (ray.o.y > bmin.y || ray.o.y < bmax.y)) &&
                  (ray.d.z != 0.f || (ray.o.z > bmin.z || ray.o.z < bmax.z));
    F3 t1 = (bmin - ray.o) * ray.d_rcp, t2 = (bmax - ray.o) * ray.d_rcp;
    F3 t1p = f3(pm_min(t1.x, t2.x), pm_min(t1.y, t2.y), pm_min(t1.z, t2.z));
    F3 t2p = f3(pm_max(t1.x, t2.x), pm_max(t1.y, t2.y), pm_max(t1.z, t2.z));
    mint = hmax(t1p);
    maxt = hmin(t2p);
    return active && maxt >= mint;
}

// ---------------------------------------------------------------- primitives
// shapes/rectangle.cpp:139-155
DEV float rectangle_intersect(const float *to_object /* rows 0..2 suffice */, const DRay &ray, F2 &uv) {
    F3 o = mat_point_affine(to_object, ray.o), d = mat_vector(to_object, ray.d);
    float t = -o.z * pm_rcp(d.z);
    float lx = pm_fma(d.x, t, o.x), ly = pm_fma(d.y, t, o.y);
    bool active = t >= ray.mint && t <= ray.maxt && pm_abs(lx) <= 1.f && pm_abs(ly) <= 1.f;
    uv.x = lx; uv.y = ly;
    return active ? t : pm_inf();
}
// shapes/disk.cpp:136-153
DEV float disk_intersect(const float *to_object, const DRay &ray, F2 &uv) {
    F3 o = mat_point_affine(to_object, ray.o), d = mat_vector(to_object, ray.d);
    float t = -o.z * pm_rcp(d.z);
    float lx = pm_fma(d.x, t, o.x), ly = pm_fma(d.y, t, o.y);
    bool active = t >= ray.mint && t <= ray.maxt && lx * lx + ly * ly <= 1.f;
    uv.x = lx; uv.y = ly;
    return active ? t : pm_inf();
}
// render/mesh.h:195-226 with p0 / e1 / e2 precomputed per primitive (same subtraction, done once on the host).
// The two `__ballot(...) == 0` exits skip the rest of the test when NO lane of the wave can still hit this
// triangle; a lane's own result never depends on them.
struct TriRec { float v[9]; };
DEV float triangle_intersect(const TriRec &T, const DRay &ray, F2 &uv) {
    F3 p0 = f3(T.v), e1 = f3(T.v + 3), e2 = f3(T.v + 6);
    F3 pvec = cross(ray.d, e2);
    float inv_det = pm_rcp(dot(e1, pvec));
    F3 tvec = ray.o - p0;
    float u = dot(tvec, pvec) * inv_det;
    bool active = u >= 0.f && u <= 1.f;
    if (__ballot(active) == 0) return pm_inf();
    F3 qvec = cross(tvec, e1);
    float v = dot(ray.d, qvec) * inv_det;
    active = active && v >= 0.f && u + v <= 1.f;
    if (__ballot(active) == 0) return pm_inf();
    float t = dot(e2, qvec) * inv_det;
    active = active && t >= ray.mint && t <= ray.maxt;
    uv.x = u; uv.y = v;
    return active ? t : pm_inf();
}
// shapes/sphere.cpp:272-306 in double precision (the reference's CPU path, sphere.cpp:276) + core/math.h:371-411
// (a real function: double-precision code that only scenes with spheres execute)
DEV_NOINLINE float sphere_intersect_v(float cx, float cy, float cz, float radius, const DRay ray);
#if MTS_TRAITS & MT_NO_SPHERE
DEV float sphere_intersect(const float *, float, const DRay &) { return pm_inf(); }
#else
DEV float sphere_intersect(const float *center, float radius, const DRay &ray) { return sphere_intersect_v(center[0], center[1], center[2], radius, ray); }
#endif
DEV_NOINLINE float sphere_intersect_v(float cx, float cy, float cz, float radius, const DRay ray) {
    const float center[3] = { cx, cy, cz };
    double mint = ray.mint, maxt = ray.maxt;
    double ox = (double) ray.o.x - (double) center[0], oy = (double) ray.o.y - (double) center[1], oz = (double) ray.o.z - (double) center[2];
    double dx = ray.d.x, dy = ray.d.y, dz = ray.d.z;
    double A = pm_fma_d(dz, dz, pm_fma_d(dy, dy, dx * dx));
    double B = 2.0 * pm_fma_d(oz, dz, pm_fma_d(oy, dy, ox * dx));
    double Cc = pm_fma_d(oz, oz, pm_fma_d(oy, oy, ox * ox)) - (double) radius * (double) radius;
    bool linear_case = A == 0.0, valid_linear = linear_case && B != 0.0;
    double x0 = -Cc / B, x1 = x0;
    double discrim = pm_fma_d(B, B, -(4.0 * A * Cc));
    bool valid_quadratic = !linear_case && discrim >= 0.0;
    if (valid_quadratic) {
        double sqrt_discrim = __builtin_sqrt(discrim);
        double temp = -0.5 * (B + __builtin_copysign(sqrt_discrim, B));
        double x0p = temp / A, x1p = Cc / temp;
        x0 = x1p < x0p ? x1p : x0p; x1 = x0p < x1p ? x1p : x0p;
    }
    bool found = valid_linear || valid_quadratic;
    bool out_bounds = !(x0 <= maxt && x1 >= mint);
    bool in_bounds = x0 < mint && x1 > maxt;
    bool active = found && !out_bounds && !in_bounds;
    return active ? (x0 < mint ? (float) x1 : (float) x0) : pm_inf();
}

// ---------------------------------------------------------------- scene traversal
// Closest hit with the semantics of ShapeKDTree::ray_intersect_scalar (render/kdtree.h:2078-2171):
// clip against the scene bounding box, accept t in [mint, maxt], shrink maxt on every accepted hit, a
// later primitive at the same t replaces the earlier one (`t <= maxt`).  For the handful of
// primitives of the atmosphere scenes the "acceleration structure" is the primitive list itself,
// walked with wave-uniform (scalar) loads -- no per-lane memory traffic at all.
// One primitive addressed per lane (BVH leaves): the same tests as the scalar walk, records fetched with vector loads.
struct BvhArgs { const float *nodes; const int32_t *leaf_prims; int32_t node_count; const DWalkPrim *walk;
                 const float *lds_nodes; int32_t lds_count; };
// One primitive addressed per lane (BVH leaves): the same tests as the scalar walk on the same 64-byte record, fetched with
// four 16-byte vector loads (no prim -> shape -> geometry chain of dependent loads).
DEV float prim_intersect_lane(const BvhArgs &a, int pi, const DRay &ray, F2 &uv, int &shape, int &index) {
    typedef int mts_int4 __attribute__((ext_vector_type(4)));
    const MTS_GLOBAL_AS mts_int4 *rec = (const MTS_GLOBAL_AS mts_int4 *) as_global(a.walk + pi);
    const mts_int4 hd = rec[0], g0 = rec[1], g1 = rec[2], g2 = rec[3];
    shape = hd.y; index = hd.z;
    float f[12] = { __int_as_float(g0.x), __int_as_float(g0.y), __int_as_float(g0.z), __int_as_float(g0.w),
                    __int_as_float(g1.x), __int_as_float(g1.y), __int_as_float(g1.z), __int_as_float(g1.w),
                    __int_as_float(g2.x), __int_as_float(g2.y), __int_as_float(g2.z), __int_as_float(g2.w) };
    uv.x = uv.y = 0.f;
    if (hd.x == MTS_SHAPE_RECTANGLE) return rectangle_intersect(f, ray, uv);
    if (hd.x == MTS_SHAPE_DISK) return disk_intersect(f, ray, uv);
    if (hd.x == MTS_SHAPE_SPHERE) return sphere_intersect(f, f[3], ray);
    TriRec T;
    for (int k = 0; k < 9; ++k) T.v[k] = f[k];
    return triangle_intersect(T, ray, uv);
}
// Stack-free traversal of the host-built BVH (dscene.h).  Node boxes are conservative, the exact primitive tests decide;
// "closest t, ties to the later primitive" is the order-independent form of the sequential rule of kdtree.h:2152-2154.
// A real function (arguments by value): the traversal loop keeps its registers out of the callers' hot paths, and its
// callee-saved spills are only paid by scenes that have a BVH.
template <bool ShadowRay>
DEV_NOINLINE Hit bvh_intersect(const BvhArgs a, DRay ray) {
    Hit h; h.t = pm_inf(); h.p = f3s(0.f); h.uv.x = h.uv.y = 0.f; h.shape = -1; h.prim = 0;
    int best = -1;
    const MTS_GLOBAL_AS float *nodes = as_global(a.nodes);
    const MTS_GLOBAL_AS int32_t *leaf_prims = as_global(a.leaf_prims);
    const int n = a.node_count;
    int i = 0;
    while (i < n) {
        float nd[8];                                          // the top levels come from LDS when the kernel staged them
        if (i < a.lds_count) { for (int k = 0; k < 8; ++k) nd[k] = a.lds_nodes[8 * i + k]; }
        else { for (int k = 0; k < 8; ++k) nd[k] = nodes[8 * i + k]; }
        const float t1x = (nd[0] - ray.o.x) * ray.d_rcp.x, t2x = (nd[3] - ray.o.x) * ray.d_rcp.x;
        const float t1y = (nd[1] - ray.o.y) * ray.d_rcp.y, t2y = (nd[4] - ray.o.y) * ray.d_rcp.y;
        const float t1z = (nd[2] - ray.o.z) * ray.d_rcp.z, t2z = (nd[5] - ray.o.z) * ray.d_rcp.z;
        // fmin / fmax drop a NaN operand (0 * inf when the origin lies on a slab plane of a ray parallel to it): that axis
        // then simply does not constrain the interval, which keeps the test conservative
        const float tnear = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t1x, t2x), __builtin_fminf(t1y, t2y)), __builtin_fminf(t1z, t2z));
        const float tfar = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t1x, t2x), __builtin_fmaxf(t1y, t2y)), __builtin_fmaxf(t1z, t2z));
        const int skip = __float_as_int(nd[6]), link = __float_as_int(nd[7]);
        if (!(tnear <= tfar && tfar >= ray.mint && tnear <= ray.maxt)) { i = skip; continue; }
        if (link < 0) { i = -link; continue; }                // inner node: descend into the left child
        const int count = link & 7, first = link >> 3;
        for (int k = 0; k < count; ++k) {
            const int pi = leaf_prims[first + k];
            F2 uv; int shape, index;
            const float t = prim_intersect_lane(a, pi, ray, uv, shape, index);
            if (t != pm_inf() && (t < h.t || pi > best)) {       // the tests accept t <= ray.maxt == h.t: equal t goes to the later primitive
                h.t = t; h.uv = uv; h.shape = shape; h.prim = index; best = pi;
                if (ShadowRay) return h;
                ray.maxt = t;
            }
        }
        i = skip;
    }
    return h;
}

template <bool ShadowRay>
DEV Hit ray_intersect_preliminary(const DScene &sc, DRay ray) {
    Hit h; h.t = pm_inf(); h.p = f3s(0.f); h.uv.x = h.uv.y = 0.f; h.shape = -1; h.prim = 0;
    float bmint, bmaxt;
    bbox_ray_intersect(sc.bbox, ray, bmint, bmaxt);
    float mint = pm_max(ray.mint, bmint), maxt = pm_min(ray.maxt, bmaxt);
    if (!(mint <= maxt)) return h;
#if !(MTS_TRAITS & MT_NO_BVH)
    if (sc.bvh_node_count > 0) {
        BvhArgs a; a.nodes = sc.bvh_nodes; a.leaf_prims = sc.bvh_prims; a.node_count = sc.bvh_node_count; a.walk = sc.walk; a.lds_nodes = sc.bvh_lds; a.lds_count = sc.bvh_lds_count;
        return bvh_intersect<ShadowRay>(a, ray);
    }
#endif
    for (int i = 0; i < sc.prim_count; ++i) {
        const DWalkPrim w = cload(sc.walk + i);               // one 64-byte scalar load per primitive (fetching one ahead measured slower)
        F2 uv; uv.x = uv.y = 0.f; float t;
        if (w.type == MTS_SHAPE_RECTANGLE) t = rectangle_intersect(w.f, ray, uv);
        else if (w.type == MTS_SHAPE_DISK) t = disk_intersect(w.f, ray, uv);
        else if (w.type == MTS_SHAPE_SPHERE) t = sphere_intersect(w.f, w.f[3], ray);
        else { TriRec T; for (int k = 0; k < 9; ++k) T.v[k] = w.f[k]; t = triangle_intersect(T, ray, uv); }
        if (t != pm_inf()) {
            h.t = t; h.uv = uv; h.shape = w.shape; h.prim = w.index;
            if (ShadowRay) return h;
            ray.maxt = t;
        }
    }
    return h;
}

// Hit point: rectangle.cpp:181-185, mesh.cpp:470-483, sphere.cpp:325-327
DEV void hit_point(const DScene &sc, const DShape &s, const DRay &ray, Hit &h) {
    if (s.type == MTS_SHAPE_RECTANGLE || s.type == MTS_SHAPE_DISK) {                          // rectangle.cpp:181-185 == disk.cpp:186-188
        F3 p = ray_at(ray, h.t), n = f3(s.frame_n);
        float dist = dot(f3(s.to_world.m[3], s.to_world.m[7], s.to_world.m[11]) - p, n);
        h.p = fmadd(n, dist, p);
    } else if (s.type == MTS_SHAPE_SPHERE) {
        F3 n = normalize(ray_at(ray, h.t) - f3(s.center));
        h.p = fmadd(n, s.radius, f3(s.center));
        h.uv.x = n.x; h.uv.y = n.y; h.prim = (int32_t) pm_bits(n.z);     // keep the exact normal for complete_surface()
    } else {
        const MTS_GLOBAL_AS float *A = as_global(sc.tri_attr) + 24 * (s.prim_offset + h.prim);
        F3 p0 = f3(A), p1 = f3(A + 3), p2 = f3(A + 6);
        float b1 = h.uv.x, b2 = h.uv.y, b0 = 1.f - b1 - b2;
        h.p = p0 * b0 + p1 * b1 + p2 * b2;
    }
}

// librender/scene_native.inl:23-41
DEV Hit ray_intersect(const DScene &sc, const DRay &ray) {
    Hit h = ray_intersect_preliminary<false>(sc, ray);
    if (hit_valid(h)) {
        WATERFALL_BEGIN(h.shape, su)
            hit_point(sc, cload(sc.shapes + su), ray, h);
        WATERFALL_END
    }
    return h;
}
DEV bool ray_test(const DScene &sc, const DRay &ray) { return hit_valid(ray_intersect_preliminary<true>(sc, ray)); }

// Rebuild n, shading frame and wi: rectangle.cpp:186-193, mesh.cpp:485-545, sphere.cpp:328-371,
// interaction.h:153-156,571-596.  `d` is the direction of the ray that produced the hit.
DEV void complete_surface(const DScene &sc, const DShape &s, const Hit &h, F3 d, Surf &sf) {
    F3 dp_du, dp_dv, shn;
    if (s.type == MTS_SHAPE_RECTANGLE) {
        sf.n = f3(s.frame_n); shn = sf.n; dp_du = f3(s.frame_s);
    } else if (s.type == MTS_SHAPE_DISK) {                                                    // disk.cpp:190-207, h.uv = prim_uv
        sf.n = f3(s.frame_n); shn = sf.n;
        float r = pm_sqrt(pm_fma(h.uv.y, h.uv.y, h.uv.x * h.uv.x)), inv_r = pm_rcp(r);
        float cos_phi = r != 0.f ? h.uv.x * inv_r : 1.f, sin_phi = r != 0.f ? h.uv.y * inv_r : 0.f;
        dp_du = mat_vector(s.to_world.m, f3(cos_phi, sin_phi, 0.f));
    } else if (s.type == MTS_SHAPE_SPHERE) {
        // hit_point() parked the unit normal normalize(ray(t) - center) in (uv.x, uv.y, bits(prim))
        shn = f3(h.uv.x, h.uv.y, pm_from_bits((uint32_t) h.prim));
        F3 local = mat_point_affine(s.to_object.m, h.p);
        dp_du = mat_vector(s.to_world.m, f3(-local.y, local.x, 0.f)) * (2.f * MTS_PI);
        if (s.flip_normals) shn = -shn;
        sf.n = shn;
    } else {
        const MTS_GLOBAL_AS float *A = as_global(sc.tri_attr) + 24 * (s.prim_offset + h.prim);
        F3 p0 = f3(A), p1 = f3(A + 3), p2 = f3(A + 6);
        float b1 = h.uv.x, b2 = h.uv.y, b0 = 1.f - b1 - b2;
        F3 dp0 = p1 - p0, dp1 = p2 - p0;
        sf.n = normalize(cross(dp0, dp1));
        coordinate_system(sf.n, dp_du, dp_dv);
        if (s.has_texcoords) {
            float u0x = A[18], u0y = A[19], u1x = A[20], u1y = A[21], u2x = A[22], u2y = A[23];
            float d0x = u1x - u0x, d0y = u1y - u0y, d1x = u2x - u0x, d1y = u2y - u0y;
            float det = pm_fma(d0x, d1y, -(d0y * d1x)), inv_det = pm_rcp(det);
            if (det != 0.f)
                dp_du = f3(pm_fma(d1y, dp0.x, -(d0y * dp1.x)), pm_fma(d1y, dp0.y, -(d0y * dp1.y)), pm_fma(d1y, dp0.z, -(d0y * dp1.z))) * inv_det;
        }
        if (s.has_normals) {
            F3 n0 = f3(A + 9), n1 = f3(A + 12), n2 = f3(A + 15);
            shn = normalize(n0 * b0 + n1 * b1 + n2 * b2);
        } else shn = sf.n;
    }
    sf.sh.n = shn;
    sf.sh.s = normalize(fnmadd(shn, dot(shn, dp_du), dp_du));
    sf.sh.t = cross(shn, sf.sh.s);
    sf.wi = to_local(sf.sh, -d);
}

DEV void complete_surface(const DScene &sc, const Hit &h, F3 d, Surf &sf) {
    WATERFALL_BEGIN(h.shape, su)
        complete_surface(sc, cload(sc.shapes + su), h, d, sf);
    WATERFALL_END
}

// ---------------------------------------------------------------- spectra
// What a colour parameter of a plugin evaluates to for the sample at hand.  rgb / mono: the record's own three floats.  Spectral:
// the parameter's spectrum (DScene::spectra through the per-plugin index arrays) at the sample's four wavelengths, which travel
// in SpecCtx.  Functions take `cx` (and the plugin's index `id`) as trailing arguments with defaults, so the rgb call sites read
// as they always did and SpecCtx is an empty struct there.
#if MTS_SPEC_N == 3
struct SpecCtx { };
DEV SpecCtx make_ctx(const DScene &) { return SpecCtx(); }
#define BSDF_COLOR(b, field, which, cx, id) f3((b).field)
#define EMITTER_COLOR(e, cx, id) f3((e).radiance)
#else
struct SpecCtx { Spec wl; const DSpectrum *spectra; const int32_t *bsdf_sp, *emitter_sp; const DVolumeSp *volume_sp; };
DEV SpecCtx make_ctx(const DScene &sc) {
    SpecCtx cx; cx.wl = spec_s(0.f); cx.spectra = sc.spectra; cx.bsdf_sp = sc.bsdf_sp; cx.emitter_sp = sc.emitter_sp; cx.volume_sp = sc.volume_sp;
    return cx;
}
// spectra/uniform.cpp:47-57 ; spectra/regular.cpp:71-78 -> ContinuousDistribution::eval_pdf (core/distr_1d.h:378-400)
DEV float spectrum_eval_1(const DSpectrum &s, float lambda) {
    const bool active = lambda >= s.lambda_min && lambda <= s.lambda_max;
    if (s.type == MTS_SPECTRUM_UNIFORM) return active ? s.value : 0.f;
    if (s.type == MTS_SPECTRUM_DISCRETE) return 0.f;                       // discrete.cpp:112-116: a sampling-only response
    if (s.type == MTS_SPECTRUM_IRREGULAR) {                                // irregular.cpp:75-84 -> IrregularContinuousDistribution::eval_pdf (distr_1d.h:655-677)
        const MTS_GLOBAL_AS float *nodes = as_global(s.wavelengths), *vals = as_global(s.values);
        const uint32_t size = (uint32_t) s.count;
        uint32_t start = 0, end = size, iterations = 1;                    // enoki::binary_search(0, size, nodes[i] < x)
        { uint32_t diff = end - start; while (diff >>= 1) iterations++; }
        for (uint32_t i = 0; i < iterations; ++i) {
            const uint32_t middle = (start + end) >> 1;
            if (nodes[min(middle, size - 1u)] < lambda) start = min(middle + 1u, end); else end = middle;
        }
        const uint32_t index = max(min(start, size - 1u), 1u) - 1u;
        const float x0 = nodes[index], x1 = nodes[index + 1], y0 = vals[index], y1 = vals[index + 1];
        const float x = (lambda - x0) / (x1 - x0);
        return active ? pm_fma(x, y1 - y0, y0) : 0.f;
    }
    float x = (lambda - s.lambda_min) * s.inv_interval_size;
    long long xi = (long long) x;
    uint32_t index = (uint32_t) (xi < 0 ? 0 : (xi > (long long) s.count - 2 ? (long long) s.count - 2 : xi));
    const MTS_GLOBAL_AS float *v = as_global(s.values);
    float y0 = active ? v[index] : 0.f, y1 = active ? v[index + 1] : 0.f;
    float w1 = x - (float) index, w0 = 1.f - w1;
    return pm_fma(w0, y0, w1 * y1);
}
DEV Spec spectrum_eval(const DSpectrum *spectra, int idx, Spec wl) {
    const DSpectrum s = spectra[idx];
    return spec4(spectrum_eval_1(s, wl.x), spectrum_eval_1(s, wl.y), spectrum_eval_1(s, wl.z), spectrum_eval_1(s, wl.w));
}
#define BSDF_COLOR(b, field, which, cx, id) spectrum_eval((cx).spectra, (cx).bsdf_sp[(id) * MTS_BSDF_SP_COUNT + (which)], (cx).wl)
#define EMITTER_COLOR(e, cx, id) spectrum_eval((cx).spectra, (cx).emitter_sp[id], (cx).wl)
#endif

// ---------------------------------------------------------------- volumes
// textures/grid3d.cpp:234-250
DEV int wrap_coord(int wrap, int value, int res) {
    if (wrap == MTS_WRAP_CLAMP) return min(max(value, 0), res - 1);
    int div = value / res;
    int mod = value - div * res;
    if (mod < 0) mod += res;
    if (wrap == MTS_WRAP_MIRROR) mod = (((div & 1) == 0) ^ (value < 0)) ? mod : res - 1 - mod;
    return mod;
}
DEV float trilerp(float d000, float d100, float d010, float d110, float d001, float d101, float d011, float d111, F3 w0, F3 w1) {
    float v00 = pm_fma(w0.x, d000, w1.x * d100), v01 = pm_fma(w0.x, d001, w1.x * d101),
          v10 = pm_fma(w0.x, d010, w1.x * d110), v11 = pm_fma(w0.x, d011, w1.x * d111);
    float v0 = pm_fma(w0.y, v00, w1.y * v10), v1 = pm_fma(w0.y, v01, w1.y * v11);
    return pm_fma(w0.z, v0, w1.z * v1);
}
// textures/grid3d.cpp:220-232,259-360 ; textures/constant3d.cpp
// The grid lookup with its wrap modes, channel counts and filters is a real function: it is off the hot paths (the metric scene
// reads its grids through the pair-grid fast path, constant volumes return above), and inlining its ~25 copies of the repeat /
// mirror index arithmetic into every block was costing instruction-cache space.
// It receives the 23 dwords of the volume record it reads, which travel in argument registers; the whole record (by value) would
// be copied through scratch memory at every call.
struct GridRef { float w2l[16]; const float *data; int32_t nx, ny, nz; uint32_t channels_affine_filter_wrap; };
DEV_CALL_UNLESS_LEAN F3 volume_eval_grid(const GridRef g, F3 p_world);
#if MTS_SPEC_N != 3
DEV Spec volume_eval_grid_spectral(const GridRef g, F3 p_world, Spec wl, float lambda_min, float lambda_max);
#endif
DEV Spec volume_eval(const DVolume &v, F3 p_world, const SpecCtx &cx = SpecCtx(), int vid = 0) {
#if MTS_SPEC_N == 3
    if (v.type == MTS_VOLUME_CONST) return f3(v.value);
#else
    if (v.type == MTS_VOLUME_CONST) return spectrum_eval(cx.spectra, cx.volume_sp[vid].value_sp, cx.wl);    // constant3d.cpp: m_color->eval(si)
#endif
    GridRef g;
    for (int k = 0; k < 16; ++k) g.w2l[k] = v.w2l[k];
    g.data = v.data; g.nx = v.nx; g.ny = v.ny; g.nz = v.nz;
    g.channels_affine_filter_wrap = (uint32_t) v.channels | ((uint32_t) (v.affine != 0) << 8) | ((uint32_t) (v.columns_equal != 0) << 9) | ((uint32_t) v.filter << 16) | ((uint32_t) v.wrap << 24);
#if MTS_TRAITS & MT_NO_GRID_EVAL
    return spec_s(0.f);
#elif MTS_SPEC_N == 3
    return volume_eval_grid(g, p_world);
#else
    const DVolumeSp vs = cx.volume_sp[vid];
    if (vs.spectral_grid) return volume_eval_grid_spectral(g, p_world, cx.wl, vs.lambda_min, vs.lambda_max);
    return spec_s(volume_eval_grid(g, p_world).x);          // single-channel grid (the loader refuses 3-channel grids: they need the sRGB model)
#endif
}
#if MTS_SPEC_N != 3
// textures/gridvolume_spectral.cpp:226-388: trilinear in space (cell-centred values, wrapped indices), linear in the spectral
// dimension (values at nodes over [lambda_min, lambda_max], clamped indices), zero outside the interval
// NG grids of identical geometry and spectral interval (a medium's sigma_t and albedo) share the cell, the weights and the spectral nodes
struct SpecPair { Spec a, b; };
// COLUMNS_EQUAL: every grid of the call is a profile in z only (DVolume::columns_equal) -- the four corners of a z-level hold one value:
// two gathers instead of eight, the same values into the same interpolation.  A template parameter, not a branch inside the loops: the
// gathers of all wavelengths must stay one straight-line batch (a uniform branch around them measured 13 % slower on C5S than no fast path).
template <int NG, bool COLUMNS_EQUAL>
DEV void volume_eval_grid_spectral_n(const GridRef &g, const float *data_b, F3 p_world, Spec wl, float lambda_min, float lambda_max, Spec out_s[NG]) {
    const int channels = (int) (g.channels_affine_filter_wrap & 0xffu), affine = (int) ((g.channels_affine_filter_wrap >> 8) & 1u),
              wrap = (int) (g.channels_affine_filter_wrap >> 24);
    F3 p = affine ? mat_point_affine(g.w2l, p_world) : mat_point(g.w2l, p_world);
    const MTS_GLOBAL_AS float *DD[2] = { as_global(g.data), as_global(data_b) }; const int nx = g.nx, ny = g.ny, nz = g.nz, ch = channels;
    p = f3(pm_fma(p.x, (float) nx, -.5f), pm_fma(p.y, (float) ny, -.5f), pm_fma(p.z, (float) nz, -.5f));
    int ix = (int) pm_floor(p.x), iy = (int) pm_floor(p.y), iz = (int) pm_floor(p.z);
    F3 w1 = p - f3((float) ix, (float) iy, (float) iz), w0 = f3(1.f - w1.x, 1.f - w1.y, 1.f - w1.z);
    int x0 = wrap_coord(wrap, ix, nx), x1 = wrap_coord(wrap, ix + 1, nx), y0 = wrap_coord(wrap, iy, ny), y1 = wrap_coord(wrap, iy + 1, ny),
        z0 = wrap_coord(wrap, iz, nz), z1 = wrap_coord(wrap, iz + 1, nz);
    int r00 = ((z0 * ny + y0) * nx), r10 = ((z0 * ny + y1) * nx), r01 = ((z1 * ny + y0) * nx), r11 = ((z1 * ny + y1) * nx);
    const float inv_dlambda = pm_rcp(lambda_max - lambda_min), lambda_scale = (float) (ch - 1);          // array / scalar = array * (1 / scalar)
    float out[NG][4]; const float lam[4] = { wl.x, wl.y, wl.z, wl.w };
    const int corner[8] = { (r00 + x0) * ch, (r00 + x1) * ch, (r10 + x0) * ch, (r10 + x1) * ch, (r01 + x0) * ch, (r01 + x1) * ch, (r11 + x0) * ch, (r11 + x1) * ch };
    for (int k = 0; k < 4; ++k) {
        const float wn = (lam[k] - lambda_min) * inv_dlambda;               // :232-233
        const float ws = wn * lambda_scale;
        const int wi = (int) pm_floor(ws);
        const int c0 = min(max(wi, 0), ch - 1), c1 = min(max(wi + 1, 0), ch - 1);      // wrap_wavelengths: clamp (:262-265)
        const float s1 = ws - (float) wi, s0 = 1.f - s1;
        // the two spectral nodes of a corner are neighbours in memory (or one and the same at the ends): one 8-byte gather per corner
        // from the pair that starts at cb, then pick (same values, same arithmetic as two dword gathers)
        const int cb = min(c0, ch - 2);
        const bool h0 = c0 != cb, h1 = c1 != cb;
        typedef float mts_float2 __attribute__((ext_vector_type(2)));
        typedef mts_float2 __attribute__((aligned(4))) mts_float2_a4;
        for (int gi = 0; gi < NG; ++gi) {
            mts_float2 q[8];
            if (COLUMNS_EQUAL) {
                const mts_float2 lo = *(const MTS_GLOBAL_AS mts_float2_a4 *) (DD[gi] + z0 * ny * nx * ch + cb),
                                 hi = *(const MTS_GLOBAL_AS mts_float2_a4 *) (DD[gi] + z1 * ny * nx * ch + cb);
                q[0] = q[1] = q[2] = q[3] = lo; q[4] = q[5] = q[6] = q[7] = hi;
            } else
            for (int j = 0; j < 8; ++j) q[j] = *(const MTS_GLOBAL_AS mts_float2_a4 *) (DD[gi] + corner[j] + cb);
            float d[2];
            d[0] = trilerp(h0 ? q[0].y : q[0].x, h0 ? q[1].y : q[1].x, h0 ? q[2].y : q[2].x, h0 ? q[3].y : q[3].x,
                           h0 ? q[4].y : q[4].x, h0 ? q[5].y : q[5].x, h0 ? q[6].y : q[6].x, h0 ? q[7].y : q[7].x, w0, w1);
            d[1] = trilerp(h1 ? q[0].y : q[0].x, h1 ? q[1].y : q[1].x, h1 ? q[2].y : q[2].x, h1 ? q[3].y : q[3].x,
                           h1 ? q[4].y : q[4].x, h1 ? q[5].y : q[5].x, h1 ? q[6].y : q[6].x, h1 ? q[7].y : q[7].x, w0, w1);
            const float r = pm_fma(s0, d[0], s1 * d[1]);
            // :381-385: the mask compares the NORMALISED wavelength with lambda_min / lambda_max, exactly as the source does
            out[gi][k] = (wn >= lambda_min && wn <= lambda_max) ? r : 0.f;
        }
    }
    for (int gi = 0; gi < NG; ++gi) out_s[gi] = spec4(out[gi][0], out[gi][1], out[gi][2], out[gi][3]);
}
// Four real functions (eight corners / z profile, one grid / two): their register need counts towards the kernel's, and with both
// instantiations in one function the allocator took 180 VGPRs where the regrouping kernel has 168 (three 256-path workgroups per CU).
template <int NG, bool COLUMNS_EQUAL>
DEV_CALL_UNLESS_LEAN SpecPair volume_eval_grid_spectral_f(const GridRef g, const float *data_b, F3 p_world, Spec wl, float lambda_min, float lambda_max) {
    Spec o[2]; o[1] = spec_s(0.f);
    volume_eval_grid_spectral_n<NG, COLUMNS_EQUAL>(g, data_b, p_world, wl, lambda_min, lambda_max, o);
    SpecPair r; r.a = o[0]; r.b = o[1];
    return r;
}
DEV Spec volume_eval_grid_spectral(const GridRef g, F3 p_world, Spec wl, float lambda_min, float lambda_max) {
    if ((g.channels_affine_filter_wrap >> 9) & 1u) return volume_eval_grid_spectral_f<1, true>(g, nullptr, p_world, wl, lambda_min, lambda_max).a;
    return volume_eval_grid_spectral_f<1, false>(g, nullptr, p_world, wl, lambda_min, lambda_max).a;
}
DEV SpecPair volume_eval_grid_spectral_pair(const GridRef g, const float *data_b, F3 p_world, Spec wl, float lambda_min, float lambda_max) {
    if ((g.channels_affine_filter_wrap >> 9) & 1u) return volume_eval_grid_spectral_f<2, true>(g, data_b, p_world, wl, lambda_min, lambda_max);
    return volume_eval_grid_spectral_f<2, false>(g, data_b, p_world, wl, lambda_min, lambda_max);
}
#endif
DEV_CALL_UNLESS_LEAN F3 volume_eval_grid(const GridRef g, F3 p_world) {
    struct { const float *w2l; const float *data; int nx, ny, nz, channels, affine, filter, wrap; } v;
    v.w2l = g.w2l; v.data = g.data; v.nx = g.nx; v.ny = g.ny; v.nz = g.nz;
    v.channels = (int) (g.channels_affine_filter_wrap & 0xffu); v.affine = (int) ((g.channels_affine_filter_wrap >> 8) & 1u);
    v.filter = (int) ((g.channels_affine_filter_wrap >> 16) & 0xffu); v.wrap = (int) (g.channels_affine_filter_wrap >> 24);
    F3 p = v.affine ? mat_point_affine(v.w2l, p_world) : mat_point(v.w2l, p_world);    // x / 1 == x
    const MTS_GLOBAL_AS float *D = as_global(v.data); const int nx = v.nx, ny = v.ny, nz = v.nz, ch = v.channels;
    if (v.filter == MTS_FILTER_TRILINEAR) {
        p = f3(pm_fma(p.x, (float) nx, -.5f), pm_fma(p.y, (float) ny, -.5f), pm_fma(p.z, (float) nz, -.5f));
        int ix = (int) pm_floor(p.x), iy = (int) pm_floor(p.y), iz = (int) pm_floor(p.z);
        F3 w1 = p - f3((float) ix, (float) iy, (float) iz), w0 = f3(1.f - w1.x, 1.f - w1.y, 1.f - w1.z);
        int x0 = wrap_coord(v.wrap, ix, nx), x1 = wrap_coord(v.wrap, ix + 1, nx), y0 = wrap_coord(v.wrap, iy, ny), y1 = wrap_coord(v.wrap, iy + 1, ny),
            z0 = wrap_coord(v.wrap, iz, nz), z1 = wrap_coord(v.wrap, iz + 1, nz);
        int r00 = (z0 * ny + y0) * nx, r10 = (z0 * ny + y1) * nx, r01 = (z1 * ny + y0) * nx, r11 = (z1 * ny + y1) * nx;
        if (ch == 1) {
            float r = trilerp(D[r00 + x0], D[r00 + x1], D[r10 + x0], D[r10 + x1], D[r01 + x0], D[r01 + x1], D[r11 + x0], D[r11 + x1], w0, w1);
            return f3s(r);
        }
        float out[3];
        for (int c = 0; c < 3; ++c)
            out[c] = trilerp(D[(r00 + x0) * 3 + c], D[(r00 + x1) * 3 + c], D[(r10 + x0) * 3 + c], D[(r10 + x1) * 3 + c],
                             D[(r01 + x0) * 3 + c], D[(r01 + x1) * 3 + c], D[(r11 + x0) * 3 + c], D[(r11 + x1) * 3 + c], w0, w1);
        return f3(out[0], out[1], out[2]);
    }
    p = f3(p.x * (float) nx, p.y * (float) ny, p.z * (float) nz);
    int x = wrap_coord(v.wrap, (int) pm_floor(p.x), nx), y = wrap_coord(v.wrap, (int) pm_floor(p.y), ny), z = wrap_coord(v.wrap, (int) pm_floor(p.z), nz);
    int index = ((z * ny + y) * nx + x) * ch;
    if (ch == 1) return f3s(D[index]);
    return f3(D[index], D[index + 1], D[index + 2]);
}
// eval_1: grid3d.cpp:187-202, constant3d.cpp
DEV float volume_eval_1(const DVolume &v, F3 p_world, const SpecCtx &cx = SpecCtx(), int vid = 0) {
#if MTS_SPEC_N == 3
    F3 r = volume_eval(v, p_world);
    if (v.type == MTS_VOLUME_CONST) return (r.x + r.y + r.z) * (1.f / 3.f);
    if (v.channels == 1) return r.x;
    return r.x * 0.212671f + r.y * 0.715160f + r.z * 0.072169f;
#else
    if (v.type == MTS_VOLUME_CONST) return cx.spectra[cx.volume_sp[vid].value_sp].value;          // uniform.cpp:64-68 eval_1 (the loader admits uniform spectra here)
    return volume_eval(v, p_world, cx, vid).x;                                                    // single-channel grid (grid3d.cpp:187-202)
#endif
}

// ---------------------------------------------------------------- media
struct MediumSample { float t, mint; F3 p; Spec sigma_s, sigma_n, sigma_t, combined; };
DEV bool ms_valid(const MediumSample &m) { return m.t != pm_inf(); }

// librender/medium.cpp:34-75 ; media/homogeneous.cpp:33-54 ; media/heterogeneous.cpp:33-54
template <bool COUNT>
DEV MediumSample medium_sample_interaction(const DScene &sc, int medium, const DRay &ray, float sample, uint32_t channel, Counters &cnt, const SpecCtx &cx = SpecCtx()) {
    const DMedium &m = sc.media[medium];
    MediumSample mi;
    bool active = true; float mint = 0.f, maxt = pm_inf();
    if (!m.is_homogeneous) {
        active = bbox_ray_intersect(m.aabb, ray, mint, maxt);
        active = active && (pm_isfinite(mint) || pm_isfinite(maxt));
        if (!active) { mint = 0.f; maxt = pm_inf(); }
    }
    mint = pm_max(ray.mint, mint);
    maxt = pm_min(ray.maxt, maxt);
    Spec combined = m.is_homogeneous ? volume_eval(sc.volumes[m.sigma_t], ray.o, cx, m.sigma_t) * m.scale : spec_s(m.max_density);
    float mext = pick(combined, channel);
    float sampled_t = mint + (-pm_log(1.f - sample) / mext);
    bool valid_mi = active && (sampled_t <= maxt);
    mi.t = valid_mi ? sampled_t : pm_inf();
    mi.p = ray_at(ray, sampled_t);
    mi.mint = mint;
    mi.sigma_s = mi.sigma_n = mi.sigma_t = spec_s(0.f);
    if (m.is_homogeneous) {
        Spec st = volume_eval(sc.volumes[m.sigma_t], mi.p, cx, m.sigma_t) * m.scale;
        mi.sigma_t = st; mi.sigma_s = st * volume_eval(sc.volumes[m.albedo], mi.p, cx, m.albedo);
    } else if (valid_mi) {
        Spec st = m.scale * volume_eval(sc.volumes[m.sigma_t], mi.p, cx, m.sigma_t);
        mi.sigma_t = st; mi.sigma_s = st * volume_eval(sc.volumes[m.albedo], mi.p, cx, m.albedo);
        mi.sigma_n = spec_s(m.max_density) - st;
        if (COUNT) cnt.n_lookup++;
    }
    mi.combined = combined;
    return mi;
}

// ---------------------------------------------------------------- phase functions
DEV float eval_hg(float g, float cos_theta) {                                               // phase/hg.cpp:52-55
    float temp = 1.0f + g * g + 2.0f * g * cos_theta;
    return MTS_INV_FOUR_PI * (1 - g * g) / (temp * pm_sqrt(temp));
}
DEV float eval_rayleigh(float cos_theta) { return (3.f / 16.f) * MTS_INV_PI * (1.f + cos_theta * cos_theta); }   // phase/rayleigh.cpp:42-45
// core/distr_1d.h:378-400
DEV float distr_eval_pdf(const DPhase &d, float x) {
    bool active = x >= d.range_x && x <= d.range_y;
    x = (x - d.range_x) * d.inv_interval_size;
    long long xi = (long long) x;
    uint32_t index = (uint32_t) (xi < 0 ? 0 : (xi > (long long) d.size - 2 ? (long long) d.size - 2 : xi));
    const MTS_GLOBAL_AS float *pdf = as_global(d.pdf);
    float y0 = active ? pdf[index] : 0.f, y1 = active ? pdf[index + 1] : 0.f;
    float w1 = x - (float) index, w0 = 1.f - w1;
    return pm_fma(w0, y0, w1 * y1);
}
// core/distr_1d.h:438-461
DEV float distr_sample(const DPhase &d, float value) {
    value *= d.integral;
    const MTS_GLOBAL_AS float *pdf = as_global(d.pdf), *cdf = as_global(d.cdf);
    uint32_t start = d.valid_x, end = d.valid_y, iterations = 0;
    if (start < end) { uint32_t diff = end - start; iterations = 1; while (diff >>= 1) iterations++; }
    for (uint32_t i = 0; i < iterations; ++i) {
        uint32_t middle = (start + end) >> 1;
        bool cond = cdf[middle] < value;
        if (cond) start = min(middle + 1, end); else end = middle;
    }
    uint32_t index = start;
    float y0 = pdf[index], y1 = pdf[index + 1], c0 = index > 0 ? cdf[index - 1] : 0.f;
    value = (value - c0) * d.inv_interval_size;
    float t_linear = (y0 - pm_safe_sqrt(y0 * y0 + 2.f * value * (y1 - y0))) / (y0 - y1), t_const = value / y0;
    float t = (y0 == y1) ? t_const : t_linear;
    return pm_fma((float) index + t, d.interval_size, d.range_x);
}

// Record load: U = the index is wave-uniform (scalar loads), otherwise an ordinary per-lane read
template <bool U, typename T> DEV T rload(const T *base, int i) { if (U) return cload(base + i); return base[i]; }

// Leaf phase functions (blendphase recursion is resolved by the two callers below; nesting depth 1)
DEV float phase_eval_leaf(const DPhase &ph, F3 wi, F3 wo) {
    switch (ph.type) {
        case MTS_PHASE_HG: return eval_hg(ph.g, dot(wo, wi));                                // hg.cpp:81-84
        case MTS_PHASE_RAYLEIGH: return eval_rayleigh(dot(wo, wi));                          // rayleigh.cpp:69-73
        case MTS_PHASE_TABULATED: return distr_eval_pdf(ph, -dot(wo, wi)) * ph.normalization * MTS_INV_TWO_PI;   // tabphase.cpp:72-78
        default: return MTS_INV_FOUR_PI;                                                      // isotropic.cpp:43-47
    }
}
// A blendphase whose children are blendphases (blendphase.cpp:42-66 holds two arbitrary PhaseFunction children; multi-species
// atmospheres are built as such trees).  DPhase::size of a blend node is the depth of the tree below it (1: two leaves, the common
// case, handled in line by the callers); deeper trees are walked here, out of line and per lane.  The host admits MTS_BLEND_MAX_DEPTH
// levels and children that precede their parent (no cycles).  Arguments by value: a reference to the scene record would pin it in scratch.
#define MTS_BLEND_MAX_DEPTH 8
DEV float phase_eval_leaf(const DPhase &ph, F3 wi, F3 wo);
DEV_NOINLINE float phase_eval_tree(const DPhase *phases, const DVolume *volumes, int root, F3 wi, F3 p, F3 wo, const SpecCtx cx) {
    // post-order walk: value(node) = value(child 0) * (1 - weight) + value(child 1) * weight, blendphase.cpp:137-138
    float w_st[MTS_BLEND_MAX_DEPTH], v0_st[MTS_BLEND_MAX_DEPTH]; int c1_st[MTS_BLEND_MAX_DEPTH]; bool second[MTS_BLEND_MAX_DEPTH];
    int sp = 0, cur = root;
    for (int guard = 0; guard < (2 << MTS_BLEND_MAX_DEPTH); ++guard) {
        const DPhase ph = phases[cur];
        if (ph.type == MTS_PHASE_BLEND && sp < MTS_BLEND_MAX_DEPTH) {
            const float w = volume_eval_1(volumes[ph.weight_volume], p, cx, ph.weight_volume);
            w_st[sp] = pm_min(pm_max(w, 0.f), 1.f); c1_st[sp] = ph.child[1]; second[sp] = false; ++sp;
            cur = ph.child[0];
            continue;
        }
        float value = phase_eval_leaf(ph, wi, wo);
        for (;;) {                                            // return to the parents
            if (sp == 0) return value;
            if (!second[sp - 1]) { v0_st[sp - 1] = value; second[sp - 1] = true; cur = c1_st[sp - 1]; break; }
            value = v0_st[sp - 1] * (1 - w_st[sp - 1]) + value * w_st[sp - 1];
            --sp;
        }
    }
    return 0.f;
}
// The leaf a nested blendphase samples (blendphase.cpp:92-108: the child below `weight` gets sample1 / weight, the other one
// (sample1 - weight) / (1 - weight)); the leaves themselves only read sample2
DEV_NOINLINE int phase_pick_leaf(const DPhase *phases, const DVolume *volumes, int root, F3 p, float sample1, const SpecCtx cx) {
    int node = root;
    for (int level = 0; level <= MTS_BLEND_MAX_DEPTH; ++level) {
        if (phases[node].type != MTS_PHASE_BLEND) break;
        const int wv = phases[node].weight_volume;
        const float w = volume_eval_1(volumes[wv], p, cx, wv);
        const float weight = pm_min(pm_max(w, 0.f), 1.f);
        if (sample1 > weight) { sample1 = (sample1 - weight) / (1 - weight); node = phases[node].child[0]; }
        else { sample1 = sample1 / weight; node = phases[node].child[1]; }
    }
    return node;
}
template <bool U = false>
DEV float phase_eval(const DScene &sc, int phase, F3 wi, F3 p, F3 wo, const SpecCtx &cx = SpecCtx()) {
    const DPhase ph = rload<U>(sc.phases, phase);
    if (ph.type != MTS_PHASE_BLEND) return phase_eval_leaf(ph, wi, wo);
#if !(MTS_TRAITS & MT_NO_PHASE_TREE)
    if (ph.size > 1) return phase_eval_tree(sc.phases, sc.volumes, phase, wi, p, wo, cx);
#endif
    float w = volume_eval_1(rload<U>(sc.volumes, ph.weight_volume), p, cx, ph.weight_volume);  // blendphase.cpp:113-139
    float weight = pm_min(pm_max(w, 0.f), 1.f);
    return phase_eval_leaf(rload<U>(sc.phases, ph.child[0]), wi, wo) * (1 - weight) + phase_eval_leaf(rload<U>(sc.phases, ph.child[1]), wi, wo) * weight;
}
DEV F3 phase_sample_leaf(const DPhase &ph, const Frame3 &frame, F2 sample2) {
    float cos_theta;
    switch (ph.type) {
        case MTS_PHASE_HG:                                                                    // hg.cpp:57-79
            if (pm_abs(ph.g) < MTS_EPSILON) cos_theta = 1 - 2 * sample2.x;
            else { float sqr_term = (1 - ph.g * ph.g) / (1 - ph.g + 2 * ph.g * sample2.x); cos_theta = (1 + ph.g * ph.g - sqr_term * sqr_term) / (2 * ph.g); }
            break;
        case MTS_PHASE_RAYLEIGH: {                                                            // rayleigh.cpp:47-67
            float z = 2.f * (2.f * sample2.x - 1.f), tmp = pm_sqrt(z * z + 1.f);
            cos_theta = pm_cbrt(z + tmp) + pm_cbrt(z - tmp);
            break;
        }
        case MTS_PHASE_TABULATED: cos_theta = distr_sample(ph, sample2.x); break;             // tabphase.cpp:53-70
        default: return square_to_uniform_sphere(sample2);                                    // isotropic.cpp:31-41
    }
    float sin_theta = pm_safe_sqrt(1.0f - cos_theta * cos_theta);
    float sin_phi, cos_phi; pm_sincos(2.f * MTS_PI * sample2.y, &sin_phi, &cos_phi);
    return to_world(frame, f3(sin_theta * cos_phi, sin_theta * sin_phi, cos_theta));
}
// Returns wo; the pdf is never used by the integrators (volpath.cpp:171 discards it).
template <bool U = false>
DEV F3 phase_sample(const DScene &sc, int phase, const Frame3 &frame, F3 p, float sample1, F2 sample2, const SpecCtx &cx = SpecCtx()) {
    const DPhase ph = rload<U>(sc.phases, phase);
    if (ph.type != MTS_PHASE_BLEND) return phase_sample_leaf(ph, frame, sample2);
#if !(MTS_TRAITS & MT_NO_PHASE_TREE)
    if (ph.size > 1) return phase_sample_leaf(sc.phases[phase_pick_leaf(sc.phases, sc.volumes, phase, p, sample1, cx)], frame, sample2);
#endif
    float w = volume_eval_1(rload<U>(sc.volumes, ph.weight_volume), p, cx, ph.weight_volume);  // blendphase.cpp:68-111
    float weight = pm_min(pm_max(w, 0.f), 1.f);
    if (sample1 > weight) return phase_sample_leaf(rload<U>(sc.phases, ph.child[0]), frame, sample2);
    return phase_sample_leaf(rload<U>(sc.phases, ph.child[1]), frame, sample2);
}

// The same samples with the pdf the phase functions return (volpathmis keeps it; hg.cpp:76, rayleigh.cpp:64, tabphase.cpp:68,
// isotropic.cpp:39; blendphase.cpp:93-108 returns the pdf of the component it sampled)
DEV F3 phase_sample_leaf_pdf(const DPhase &ph, const Frame3 &frame, F2 sample2, float &pdf) {
    float cos_theta;
    switch (ph.type) {
        case MTS_PHASE_HG:
            if (pm_abs(ph.g) < MTS_EPSILON) cos_theta = 1 - 2 * sample2.x;
            else { float sqr_term = (1 - ph.g * ph.g) / (1 - ph.g + 2 * ph.g * sample2.x); cos_theta = (1 + ph.g * ph.g - sqr_term * sqr_term) / (2 * ph.g); }
            pdf = eval_hg(ph.g, -cos_theta);
            break;
        case MTS_PHASE_RAYLEIGH: {
            float z = 2.f * (2.f * sample2.x - 1.f), tmp = pm_sqrt(z * z + 1.f);
            cos_theta = pm_cbrt(z + tmp) + pm_cbrt(z - tmp);
            pdf = eval_rayleigh(-cos_theta);
            break;
        }
        case MTS_PHASE_TABULATED:
            cos_theta = distr_sample(ph, sample2.x);
            pdf = distr_eval_pdf(ph, -cos_theta) * ph.normalization * MTS_INV_TWO_PI;
            break;
        default: pdf = MTS_INV_FOUR_PI; return square_to_uniform_sphere(sample2);
    }
    float sin_theta = pm_safe_sqrt(1.0f - cos_theta * cos_theta);
    float sin_phi, cos_phi; pm_sincos(2.f * MTS_PI * sample2.y, &sin_phi, &cos_phi);
    return to_world(frame, f3(sin_theta * cos_phi, sin_theta * sin_phi, cos_theta));
}
DEV F3 phase_sample_pdf(const DScene &sc, int phase, const Frame3 &frame, F3 p, float sample1, F2 sample2, float &pdf, const SpecCtx &cx = SpecCtx()) {
    const DPhase &ph = sc.phases[phase];
    if (ph.type != MTS_PHASE_BLEND) return phase_sample_leaf_pdf(ph, frame, sample2, pdf);
#if !(MTS_TRAITS & MT_NO_PHASE_TREE)
    if (ph.size > 1) return phase_sample_leaf_pdf(sc.phases[phase_pick_leaf(sc.phases, sc.volumes, phase, p, sample1, cx)], frame, sample2, pdf);
#endif
    float w = volume_eval_1(sc.volumes[ph.weight_volume], p, cx, ph.weight_volume);
    float weight = pm_min(pm_max(w, 0.f), 1.f);
    if (sample1 > weight) return phase_sample_leaf_pdf(sc.phases[ph.child[0]], frame, sample2, pdf);
    return phase_sample_leaf_pdf(sc.phases[ph.child[1]], frame, sample2, pdf);
}

// ---------------------------------------------------------------- BSDFs
struct BSDFSample { F3 wo; float pdf, eta; uint32_t sampled_type; };
DEV float frame_tan_theta(F3 v) { float temp = pm_fma(-v.z, v.z, 1.f); return pm_safe_sqrt(temp) / v.z; }   // core/frame.h:67-70
DEV float frame_sin_theta(F3 v) { return pm_safe_sqrt(pm_fma(v.x, v.x, v.y * v.y)); }
DEV void frame_sincos_phi(F3 v, float &s, float &c) {                                        // core/frame.h:107-118
    float sin_theta_2 = pm_fma(v.x, v.x, v.y * v.y), inv_sin_theta = pm_rsqrt(sin_theta_2);
    float rx = v.x * inv_sin_theta, ry = v.y * inv_sin_theta;
    if (pm_abs(sin_theta_2) <= 4.f * MTS_EPSILON) { rx = 1.f; ry = 0.f; }
    else { rx = pm_min(pm_max(rx, -1.f), 1.f); ry = pm_min(pm_max(ry, -1.f), 1.f); }
    s = ry; c = rx;
}
// bsdfs/rpv.cpp:85-131
struct RpvParams { float rho_0[MTS_SPEC_N], k[MTS_SPEC_N], g[MTS_SPEC_N], rho_c[MTS_SPEC_N]; };     // passed by value: a reference would pin the caller's whole DBsdf copy in scratch
DEV_CALL_UNLESS_LEAN Spec eval_rpv_p(const RpvParams b, F3 wi, F3 wo) {
    float sin_phi1, cos_phi1, sin_phi2, cos_phi2;
    frame_sincos_phi(wi, sin_phi1, cos_phi1); frame_sincos_phi(wo, sin_phi2, cos_phi2);
    float cos_phi1_minus_phi2 = cos_phi1 * cos_phi2 + sin_phi1 * sin_phi2;
    float sin_theta1 = frame_sin_theta(wi), cos_theta1 = wi.z, tan_theta1 = frame_tan_theta(wi);
    float sin_theta2 = frame_sin_theta(wo), cos_theta2 = wo.z, tan_theta2 = frame_tan_theta(wo);
    float G = pm_safe_sqrt(tan_theta1 * tan_theta1 + tan_theta2 * tan_theta2 - 2.f * tan_theta1 * tan_theta2 * cos_phi1_minus_phi2);
    float cos_g = cos_theta1 * cos_theta2 + sin_theta1 * sin_theta2 * cos_phi1_minus_phi2;
    // The two powers are most of the work, and a grey parameter has the same value in every channel: equal inputs give equal
    // results, so a channel whose g (k) equals channel 0's reuses that channel's factor instead of computing it again.
    float out[MTS_SPEC_N], F[MTS_SPEC_N], P[MTS_SPEC_N];
    for (int c = 0; c < MTS_SPEC_N; ++c) {
        const float g = b.g[c];
        if (c > 0 && g == b.g[0]) F[c] = F[0];
        else F[c] = (1.f - g * g) / pm_pow((1.f + g * g + 2.f * g * cos_g), 1.5f);
        if (c > 0 && b.k[c] == b.k[0]) P[c] = P[0];
        else P[c] = pm_pow(cos_theta1 * cos_theta2 * (cos_theta1 + cos_theta2), b.k[c] - 1.f);
        out[c] = b.rho_0[c] * (P[c] * F[c] * (1.f + (1.f - b.rho_c[c]) / (1 + G))) * MTS_INV_PI;
    }
#if MTS_SPEC_N == 3
    return f3(out[0], out[1], out[2]);
#else
    return spec4(out[0], out[1], out[2], out[3]);
#endif
}
DEV Spec eval_rpv(const DBsdf &b, F3 wi, F3 wo, const SpecCtx &cx = SpecCtx(), int id = 0) {
    RpvParams q;
#if MTS_SPEC_N == 3
    for (int c = 0; c < 3; ++c) { q.rho_0[c] = b.rho_0[c]; q.k[c] = b.k[c]; q.g[c] = b.g[c]; q.rho_c[c] = b.rho_c[c]; }
#else
    const Spec r0 = BSDF_COLOR(b, rho_0, MTS_BSDF_SP_RHO_0, cx, id), k = BSDF_COLOR(b, k, MTS_BSDF_SP_K, cx, id),
               g = BSDF_COLOR(b, g, MTS_BSDF_SP_G, cx, id), rc = BSDF_COLOR(b, rho_c, MTS_BSDF_SP_RHO_C, cx, id);
    q.rho_0[0] = r0.x; q.rho_0[1] = r0.y; q.rho_0[2] = r0.z; q.rho_0[3] = r0.w; q.k[0] = k.x; q.k[1] = k.y; q.k[2] = k.z; q.k[3] = k.w;
    q.g[0] = g.x; q.g[1] = g.y; q.g[2] = g.z; q.g[3] = g.w; q.rho_c[0] = rc.x; q.rho_c[1] = rc.y; q.rho_c[2] = rc.z; q.rho_c[3] = rc.w;
#endif
#if MTS_TRAITS & MT_NO_RPV
    return spec_s(0.f);
#else
    return eval_rpv_p(q, wi, wo);
#endif
}
// bsdfs/bilambertian.cpp:62-190
DEV float bilambertian_reflection_weight(const DBsdf &b, const SpecCtx &cx = SpecCtx(), int id = 0) {
    Spec r = BSDF_COLOR(b, reflectance, MTS_BSDF_SP_REFLECTANCE, cx, id), t = BSDF_COLOR(b, transmittance, MTS_BSDF_SP_TRANSMITTANCE, cx, id);
    Spec q = r / (r + t);
    return spec_hmean(q);                                    // hmean; NaN when r + t == 0: masked by the callers
}
DEV bool same_side(float a, float b) { return (pm_bits(a) >> 31) == (pm_bits(b) >> 31); }      // eq(sign(a), sign(b))
DEV Spec bsdf_eval(const DBsdf &b, F3 wi, F3 wo, const SpecCtx &cx = SpecCtx(), int id = 0) {
    if (b.type == MTS_BSDF_BILAMBERTIAN) return (same_side(wi.z, wo.z) ? BSDF_COLOR(b, reflectance, MTS_BSDF_SP_REFLECTANCE, cx, id) : BSDF_COLOR(b, transmittance, MTS_BSDF_SP_TRANSMITTANCE, cx, id)) * (MTS_INV_PI * pm_abs(wo.z));
    bool active = wi.z > 0.f && wo.z > 0.f;
    if (b.type == MTS_BSDF_DIFFUSE) return active ? BSDF_COLOR(b, reflectance, MTS_BSDF_SP_REFLECTANCE, cx, id) * MTS_INV_PI * wo.z : spec_s(0.f);     // diffuse.cpp:106-120
    if (b.type == MTS_BSDF_RPV) return active ? eval_rpv(b, wi, wo, cx, id) * pm_abs(wo.z) : spec_s(0.f);   // rpv.cpp:133-142
    return spec_s(0.f);                                                                                    // null.cpp:60-63
}
DEV float bsdf_pdf(const DBsdf &b, F3 wi, F3 wo, const SpecCtx &cx = SpecCtx(), int id = 0) {
    if (b.type == MTS_BSDF_NULL) return 0.f;                                                               // null.cpp:65-68
    if (b.type == MTS_BSDF_BILAMBERTIAN) {
        float result = MTS_INV_PI * pm_abs(wo.z);
        float rw = bilambertian_reflection_weight(b, cx, id), tw = 1.f - rw;
        if (rw != rw) rw = 0.f;
        if (tw != tw) tw = 0.f;
        return result * (same_side(wi.z, wo.z) ? rw : tw);
    }
    float pdf = MTS_INV_PI * wo.z;                                                                          // warp.h:343-350
    return (wi.z > 0.f && wo.z > 0.f) ? pdf : 0.f;                                                         // diffuse.cpp:122-135, rpv.cpp:144-153
}
DEV Spec bsdf_sample(const DBsdf &b, F3 wi, float sample1, F2 sample2, BSDFSample &bs, const SpecCtx &cx = SpecCtx(), int id = 0) {
    bs.wo = f3s(0.f); bs.pdf = 0.f; bs.eta = 0.f; bs.sampled_type = 0;
    if (b.type == MTS_BSDF_BILAMBERTIAN) {                                                                 // bilambertian.cpp:62-116
        F3 wo = square_to_cosine_hemisphere(sample2);
        float rw = bilambertian_reflection_weight(b, cx, id), tw = 1.f - rw;
        if (rw != rw) rw = 0.f;
        if (tw != tw) tw = 0.f;
        bool selected_r = sample1 < rw;
        Spec value = selected_r ? spec_s(1.f) * (BSDF_COLOR(b, reflectance, MTS_BSDF_SP_REFLECTANCE, cx, id) / rw) : spec_s(1.f) * (BSDF_COLOR(b, transmittance, MTS_BSDF_SP_TRANSMITTANCE, cx, id) / tw);
        bs.pdf = MTS_INV_PI * wo.z;
        bs.pdf = selected_r ? bs.pdf * rw : bs.pdf * tw;
        bs.eta = 1.f;
        bs.sampled_type = selected_r ? F_DiffuseReflection : F_DiffuseTransmission;
        if (!(wi.z > 0.f)) wo.z = -wo.z;
        bs.wo = selected_r ? wo : f3(wo.x, wo.y, -wo.z);
        return bs.pdf > 0.f ? value : spec_s(0.f);
    }
    if (b.type == MTS_BSDF_NULL) {                                                                         // null.cpp:41-58
        bs.wo = -wi; bs.sampled_type = F_Null; bs.eta = 1.f; bs.pdf = 1.f;
        return spec_s(1.f);
    }
    bool active = wi.z > 0.f;
    if (b.type == MTS_BSDF_DIFFUSE) {                                                                      // diffuse.cpp:78-104
        if (!active) return spec_s(0.f);
        bs.wo = square_to_cosine_hemisphere(sample2);
        bs.pdf = MTS_INV_PI * bs.wo.z; bs.eta = 1.f; bs.sampled_type = F_DiffuseReflection;
        return (bs.pdf > 0.f) ? BSDF_COLOR(b, reflectance, MTS_BSDF_SP_REFLECTANCE, cx, id) : spec_s(0.f);
    }
    bs.wo = square_to_cosine_hemisphere(sample2);                                                          // rpv.cpp:85-102
    bs.pdf = MTS_INV_PI * bs.wo.z; bs.eta = 1.f; bs.sampled_type = F_GlossyReflection;
    Spec value = eval_rpv(b, wi, bs.wo, cx, id);
    return (active && bs.pdf > 0.f) ? value : spec_s(0.f);
}

// ---------------------------------------------------------------- emitters
struct DirSample { F3 p, n, d; float pdf, dist; bool delta; int32_t emitter; };

// shapes/rectangle.cpp:111-124 ; shapes/sphere.cpp (sample_position)
// The mesh tables travel as three pointers, not as the scene record: shape_sample_direction() is a real function, and a reference
// to the kernel-argument record would force a copy of all of it into scratch memory.
struct MeshTables { const float *area_pmf, *area_cdf, *tri_attr; };
DEV MeshTables mesh_tables(const DScene &sc) { MeshTables m; m.area_pmf = sc.area_pmf; m.area_cdf = sc.area_cdf; m.tri_attr = sc.tri_attr; return m; }
DEV void shape_sample_position(const MeshTables sc, const DShape &s, F2 sample, F3 &p, F3 &n, float &pdf) {
    if (s.type == MTS_SHAPE_CUBE || s.type == MTS_SHAPE_MESH) {                               // mesh.cpp:352-397
        // DiscreteDistribution::sample_reuse (distr_1d.h:141-151,187-197): first face whose running area reaches sample.y * sum
        const MTS_GLOBAL_AS float *cdf = as_global(sc.area_cdf) + s.prim_offset;
        const float value = sample.y * s.surface_area;
        int lo = s.area_lo, hi = s.area_hi;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (cdf[mid] < value) lo = mid + 1; else hi = mid; }
        const float pmf = as_global(sc.area_pmf)[s.prim_offset + lo] * s.inv_surface_area, c = lo > 0 ? cdf[lo - 1] * s.inv_surface_area : 0.f;
        sample.y = (sample.y - c) / pmf;
        const MTS_GLOBAL_AS float *A = as_global(sc.tri_attr) + 24 * (s.prim_offset + lo);
        F3 p0 = f3(A), p1 = f3(A + 3), p2 = f3(A + 6);
        F3 e0 = p1 - p0, e1 = p2 - p0;
        float t = pm_safe_sqrt(1.f - sample.x), bx = 1.f - t, by = t * sample.y;             // warp.h:153-156
        p = p0 + e0 * bx + e1 * by;
        if (s.has_normals) {
            F3 n0 = f3(A + 9), n1 = f3(A + 12), n2 = f3(A + 15);
            n = normalize(n0 * (1.f - bx - by) + n1 * bx + n2 * by);
        } else n = normalize(cross(e0, e1));
    } else if (s.type == MTS_SHAPE_RECTANGLE) {
        p = mat_point_affine(s.to_world.m, f3(sample.x * 2.f - 1.f, sample.y * 2.f - 1.f, 0.f));
        n = f3(s.frame_n);
    } else if (s.type == MTS_SHAPE_DISK) {                                                    // disk.cpp:114-128
        F2 q = square_to_uniform_disk_concentric(sample);
        p = mat_point_affine(s.to_world.m, f3(q.x, q.y, 0.f));
        n = f3(s.frame_n);
    } else {
        F3 local = square_to_uniform_sphere(sample);
        p = fmadd(local, s.radius, f3(s.center));
        n = s.flip_normals ? -local : local;
    }
    pdf = s.inv_surface_area;
}
// librender/shape.cpp:293-310 ; shapes/sphere.cpp (sample_direction)
DEV_NOINLINE DirSample shape_sample_direction(const MeshTables sc, const DShape &s, F3 ref_p, F2 sample) {
    DirSample ds; ds.delta = false; ds.emitter = -1;
    if (s.type != MTS_SHAPE_SPHERE) {
        shape_sample_position(sc, s, sample, ds.p, ds.n, ds.pdf);
        ds.d = ds.p - ref_p;
        float dist_squared = squared_norm(ds.d);
        ds.dist = pm_sqrt(dist_squared);
        ds.d = ds.d / ds.dist;
        float dp = pm_abs(dot(ds.d, ds.n));
        ds.pdf *= (dp != 0.f) ? dist_squared / dp : 0.f;
        return ds;
    }
    F3 center = f3(s.center);
    F3 dc_v = center - ref_p;
    float dc_2 = squared_norm(dc_v);
    float radius_adj = s.radius * (s.flip_normals ? (1.f + MTS_RAY_EPSILON) : (1.f - MTS_RAY_EPSILON));
    if (dc_2 > radius_adj * radius_adj) {
        float inv_dc = pm_rsqrt(dc_2), sin_theta_max = s.radius * inv_dc, sin_theta_max_2 = sin_theta_max * sin_theta_max,
              inv_sin_theta_max = pm_rcp(sin_theta_max), cos_theta_max = pm_safe_sqrt(1.f - sin_theta_max_2);
        float q = pm_fma(cos_theta_max - 1.f, sample.x, 1.f);
        float sin_theta_2 = sin_theta_max_2 > 0.00068523f ? 1.f - q * q : sin_theta_max_2 * sample.x;
        float cos_theta = pm_safe_sqrt(1.f - sin_theta_2);
        float cos_alpha = sin_theta_2 * inv_sin_theta_max + cos_theta * pm_safe_sqrt(pm_fma(-sin_theta_2, inv_sin_theta_max * inv_sin_theta_max, 1.f)),
              sin_alpha = pm_safe_sqrt(pm_fma(-cos_alpha, cos_alpha, 1.f));
        float sin_phi, cos_phi; pm_sincos(sample.y * (2.f * MTS_PI), &sin_phi, &cos_phi);
        F3 d = to_world(make_frame(dc_v * -inv_dc), f3(cos_phi * sin_alpha, sin_phi * sin_alpha, cos_alpha));
        ds.p = fmadd(d, s.radius, center); ds.n = d; ds.d = ds.p - ref_p;
        float dist2 = squared_norm(ds.d);
        ds.dist = pm_sqrt(dist2); ds.d = ds.d / ds.dist;
        ds.pdf = MTS_INV_TWO_PI / (1.f - cos_theta_max);
        if (ds.dist == 0.f) ds.pdf = 0.f;
    } else {
        F3 d = square_to_uniform_sphere(sample);
        ds.p = fmadd(d, s.radius, center); ds.n = d; ds.d = ds.p - ref_p;
        float dist2 = squared_norm(ds.d);
        ds.dist = pm_sqrt(dist2); ds.d = ds.d / ds.dist;
        ds.pdf = s.inv_surface_area * dist2 / pm_abs(dot(ds.d, ds.n));
    }
    ds.delta = s.radius == 0.f;
    if (s.flip_normals) ds.n = -ds.n;
    return ds;
}
DEV float shape_pdf_direction(const DShape &s, F3 ref_p, const DirSample &ds) {
    if (s.type != MTS_SHAPE_SPHERE) {                                                         // shape.cpp:312-323 (mesh.cpp:417-419)
        float pdf = s.inv_surface_area, dp = pm_abs(dot(ds.d, ds.n));
        pdf *= (dp != 0.f) ? (ds.dist * ds.dist) / dp : 0.f;
        return pdf;
    }
    float sin_alpha = s.radius * pm_rcp(norm(f3(s.center) - ref_p)), cos_alpha = pm_safe_sqrt(1.f - sin_alpha * sin_alpha);
    return sin_alpha < 0x1.fffffep-1f ? MTS_INV_TWO_PI / (1.f - cos_alpha) : s.inv_surface_area * (ds.dist * ds.dist) / pm_abs(dot(ds.d, ds.n));
}
// emitters/directional.cpp:109-141, emitters/area.cpp:122-165, emitters/constant.cpp:81-111
template <bool U = false>
DEV DirSample emitter_sample_direction(const DScene &sc, int ei, F3 ref_p, F2 sample, Spec &spec, const SpecCtx &cx = SpecCtx()) {
    const DEmitter e = rload<U>(sc.emitters, ei);
    DirSample ds;
    if (e.type == MTS_EMITTER_DIRECTIONAL) {
        F3 d = mat_vector(e.to_world.m, f3(0.f, 0.f, 1.f));
        float dist = 2.f * e.bsphere_radius;
        ds.p = ref_p - d * dist; ds.n = d; ds.pdf = 1.f; ds.delta = true; ds.d = -d; ds.dist = dist;
        spec = EMITTER_COLOR(e, cx, ei);
    } else if (e.type == MTS_EMITTER_CONSTANT) {
        F3 d = square_to_uniform_sphere(sample);
        float dist = 2.f * e.bsphere_radius;
        ds.p = ref_p + d * dist; ds.n = -d; ds.pdf = MTS_INV_FOUR_PI; ds.delta = false; ds.d = d; ds.dist = dist;
        spec = EMITTER_COLOR(e, cx, ei) / ds.pdf;
    } else if (e.type == MTS_EMITTER_POINT) {                                                    // point.cpp:80-107
        ds.p = f3(e.to_world.m[3], e.to_world.m[7], e.to_world.m[11]); ds.n = f3s(0.f); ds.pdf = 1.f; ds.delta = true;
        ds.d = ds.p - ref_p; ds.dist = norm(ds.d);
        float inv_dist = pm_rcp(ds.dist);
        ds.d = ds.d * inv_dist;
        spec = EMITTER_COLOR(e, cx, ei) * (inv_dist * inv_dist);
    } else {
#if !(MTS_TRAITS & MT_NO_SHAPE_EMITTER)
        ds = shape_sample_direction(mesh_tables(sc), sc.shapes[e.shape], ref_p, sample);
#else
        ds = DirSample();
#endif
        bool active = dot(ds.d, ds.n) < 0.f && ds.pdf != 0.f;
        spec = active ? EMITTER_COLOR(e, cx, ei) / ds.pdf : spec_s(0.f);
    }
    ds.emitter = ei;
    return ds;
}
// librender/scene.cpp:168-218
DEV DirSample sample_emitter_direction(const DScene &sc, F3 ref_p, F2 sample, bool test_visibility, Spec &spec, const SpecCtx &cx = SpecCtx()) {
    DirSample ds; ds.pdf = 0.f; ds.dist = 0.f; ds.delta = false; ds.emitter = -1; ds.p = ds.n = ds.d = f3s(0.f);
    if (sc.emitter_count == 0) { spec = spec_s(0.f); return ds; }
    if (sc.emitter_count == 1) ds = emitter_sample_direction<true>(sc, 0, ref_p, sample, spec, cx);
    else {
        float n = (float) sc.emitter_count, emitter_pdf = pm_rcp(n);
        uint32_t index = min((uint32_t) (sample.x * n), (uint32_t) sc.emitter_count - 1);
        sample.x = (sample.x - index * emitter_pdf) * n;
        ds = emitter_sample_direction(sc, (int) index, ref_p, sample, spec, cx);
        ds.pdf *= emitter_pdf;
        spec = spec * pm_rcp(emitter_pdf);
    }
    if (test_visibility && ds.pdf != 0.f) {
        DRay ray = make_ray(ref_p, ds.d, MTS_RAY_EPSILON * (1.f + hmax_abs(ref_p)), ds.dist * (1.f - MTS_SHADOW_EPSILON));
        if (ray_test(sc, ray)) spec = spec_s(0.f);
    }
    return ds;
}
// librender/scene.cpp:220-235
DEV float pdf_emitter_direction(const DScene &sc, F3 ref_p, const DirSample &ds) {
    const DEmitter &e = sc.emitters[ds.emitter];
    float value;
    if (e.type == MTS_EMITTER_DIRECTIONAL || e.type == MTS_EMITTER_POINT) value = 0.f;
    else if (e.type == MTS_EMITTER_CONSTANT) value = MTS_INV_FOUR_PI;
    else { float dp = dot(ds.d, ds.n); value = dp < 0.f ? shape_pdf_direction(sc.shapes[e.shape], ref_p, ds) : 0.f; }
    if (sc.emitter_count == 1) return value;
    return value * pm_rcp((float) sc.emitter_count);
}
// si.emitter(scene), render/scene.h:243-253 ; emitter->eval: area.cpp:63-71, constant.cpp:41-44, directional.cpp:75-78
DEV int hit_emitter(const DScene &sc, const Hit &h) { return hit_valid(h) ? sc.shapes[h.shape].emitter : sc.environment; }
DEV Spec emitter_eval(const DScene &sc, int ei, float wi_z, const SpecCtx &cx = SpecCtx()) {
    const DEmitter &e = sc.emitters[ei];
    if (e.type == MTS_EMITTER_AREA) return wi_z > 0.f ? EMITTER_COLOR(e, cx, ei) : spec_s(0.f);
    if (e.type == MTS_EMITTER_CONSTANT) return EMITTER_COLOR(e, cx, ei);
    return spec_s(0.f);
}
// render/interaction.h:178-200
DEV int target_medium(const DShape &s, F3 n, F3 d) { return dot(d, n) > 0 ? s.exterior : s.interior; }
DEV Spec null_transmission(const DScene &sc, const DShape &s) { return sc.bsdfs[s.bsdf].type == MTS_BSDF_NULL ? spec_s(1.f) : spec_s(0.f); }   // null.cpp:70-73, bsdf.cpp:11-14
DEV float mis_weight(float pdf_a, float pdf_b) { pdf_a *= pdf_a; pdf_b *= pdf_b; return pdf_a > 0.0f ? pdf_a / (pdf_a + pdf_b) : 0.0f; }   // volpath.cpp:479-483

// Geometric normal of a hit without the full shading frame (only needed for medium transitions)
DEV F3 hit_geo_normal(const DScene &sc, const DShape &s, const Hit &h) {
    if (s.type == MTS_SHAPE_RECTANGLE || s.type == MTS_SHAPE_DISK) return f3(s.frame_n);
    Surf sf; complete_surface(sc, s, h, f3(0.f, 0.f, 1.f), sf);
    return sf.n;
}
DEV F3 hit_geo_normal(const DScene &sc, const Hit &h) {
    F3 n = f3s(0.f);
    WATERFALL_BEGIN(h.shape, su)
        n = hit_geo_normal(sc, cload(sc.shapes + su), h);
    WATERFALL_END
    return n;
}

#if MTS_SPEC_N == 3
DEV F3 transmittance_exp(float t, F3 combined) { return f3(pm_exp(-t * combined.x), pm_exp(-t * combined.y), pm_exp(-t * combined.z)); }
#else
DEV Spec transmittance_exp(float t, Spec combined) { return spec4(pm_exp(-t * combined.x), pm_exp(-t * combined.y), pm_exp(-t * combined.z), pm_exp(-t * combined.w)); }
#endif

// ---------------------------------------------------------------- volpath
// integrators/volpath.cpp:261-367: NEE with ratio tracking through media and null surfaces
template <bool COUNT>
DEV Spec volpath_sample_emitter(const DScene &sc, F3 ref_p, bool is_medium_interaction, Pcg32 &rng, int medium, uint32_t channel, DirSample &ds, Counters &cnt, const SpecCtx &cx = SpecCtx()) {
    Spec transmittance = spec_s(1.f), emitter_val;
    ds = sample_emitter_direction(sc, ref_p, rng.next_2d(), false, emitter_val, cx);
    if (ds.pdf == 0.f) return spec_s(0.f);
    bool active = true;
    DRay ray = spawn_ray(ref_p, ds.d);
    if (is_medium_interaction) ray.mint = 0.f;
    float total_dist = 0.f;
    Hit si; si.t = pm_inf(); si.shape = -1; si.prim = 0; si.p = f3s(0.f); si.uv.x = si.uv.y = 0.f;
    bool needs_intersection = true;
    while (active) {
        float remaining_dist = ds.dist * (1.f - MTS_SHADOW_EPSILON) - total_dist;
        ray.maxt = remaining_dist;
        active = active && remaining_dist > 0.f;
        if (!active) break;
        if (COUNT) cnt.n_nee_step++;
        bool escaped_medium = false, active_medium = medium >= 0, active_surface = !active_medium;
        if (active_medium) {
            const DMedium &m = sc.media[medium];
            MediumSample mi = medium_sample_interaction<COUNT>(sc, medium, ray, rng.next_1d(), channel, cnt, cx);
            if (m.is_homogeneous && ms_valid(mi)) ray.maxt = pm_min(mi.t, remaining_dist);
            if (needs_intersection) si = ray_intersect(sc, ray);
            if (si.t < mi.t) mi.t = pm_inf();
            needs_intersection = false;
            bool is_spectral = m.has_spectral_extinction != 0, not_spectral = !is_spectral;
            if (is_spectral) {
                float t = pm_min(remaining_dist, pm_min(mi.t, si.t)) - mi.mint;
                Spec tr = transmittance_exp(t, mi.combined);
                Spec free_flight_pdf = (si.t < mi.t || mi.t > remaining_dist) ? tr : tr * mi.combined;
                float tr_pdf = pick(free_flight_pdf, channel);
                transmittance = transmittance * (tr_pdf > 0.f ? tr / tr_pdf : spec_s(0.f));
            }
            if (mi.t > remaining_dist && ms_valid(mi)) total_dist = ds.dist;
            if (mi.t > remaining_dist) mi.t = pm_inf();
            escaped_medium = !ms_valid(mi);
            active_medium = ms_valid(mi);
            if (active_medium) {
                total_dist += mi.t;
                ray.o = mi.p; ray.mint = 0.f; si.t = si.t - mi.t;
                if (is_spectral) transmittance = transmittance * mi.sigma_n;
                if (not_spectral) transmittance = transmittance * (mi.sigma_n / mi.combined);
            }
        }
        bool intersect = active_surface && needs_intersection;
        if (intersect) si = ray_intersect(sc, ray);
        needs_intersection = needs_intersection && !intersect;
        active_surface = active_surface || escaped_medium;
        if (active_surface) total_dist += si.t;
        active_surface = active_surface && hit_valid(si) && active && !active_medium;
        if (active_surface) {
            const DShape &s = sc.shapes[si.shape];
            transmittance = transmittance * null_transmission(sc, s);
            ray = spawn_ray(si.p, ray.d);
        }
        ray.maxt = remaining_dist;
        needs_intersection = needs_intersection || active_surface;
        active = active && (active_medium || active_surface) && any_nonzero(transmittance);
        if (active_surface && sc.shapes[si.shape].is_medium_transition) medium = target_medium(sc.shapes[si.shape], hit_geo_normal(sc, si), ray.d);
    }
    return transmittance * emitter_val;
}

// integrators/volpath.cpp:370-465
template <bool COUNT>
DEV Spec volpath_evaluate_direct_light(const DScene &sc, F3 ref_p, Pcg32 &rng, int medium, DRay ray, Hit si, uint32_t channel, bool active, float &emitter_pdf, Counters &cnt, const SpecCtx &cx = SpecCtx()) {
    Spec emitter_val = spec_s(0.f), transmittance = spec_s(1.f);
    bool needs_intersection = false;
    emitter_pdf = 0.f;
    while (active) {
        if (COUNT) cnt.n_nee_step++;
        bool escaped_medium = false, active_medium = medium >= 0, active_surface = !active_medium;
        if (active_medium) {
            const DMedium &m = sc.media[medium];
            MediumSample mi = medium_sample_interaction<COUNT>(sc, medium, ray, rng.next_1d(), channel, cnt, cx);
            if (m.is_homogeneous && ms_valid(mi)) ray.maxt = mi.t;
            if (needs_intersection) si = ray_intersect(sc, ray);
            if (si.t < mi.t) mi.t = pm_inf();
            bool is_spectral = m.has_spectral_extinction != 0, not_spectral = !is_spectral;
            if (is_spectral) {
                float t = pm_min(mi.t, si.t) - mi.mint;                                       // medium.cpp:77-89
                Spec tr = transmittance_exp(t, mi.combined);
                Spec free_flight_pdf = si.t < mi.t ? tr : tr * mi.combined;
                float tr_pdf = pick(free_flight_pdf, channel);
                transmittance = transmittance * (tr_pdf > 0.f ? tr / tr_pdf : spec_s(0.f));
            }
            needs_intersection = false;
            escaped_medium = !ms_valid(mi);
            active_medium = ms_valid(mi);
            if (active_medium) {
                ray.o = mi.p; ray.mint = 0.f; si.t = si.t - mi.t;
                if (is_spectral) transmittance = transmittance * mi.sigma_n;
                if (not_spectral) transmittance = transmittance * (mi.sigma_n / mi.combined);
            }
        }
        bool intersect = active_surface && needs_intersection;
        if (intersect) si = ray_intersect(sc, ray);
        needs_intersection = needs_intersection && !intersect;
        active_surface = active_surface || escaped_medium;
        int emitter = active_surface ? hit_emitter(sc, si) : -1;
        if (emitter >= 0) {
            Surf sf; sf.wi = -ray.d; sf.sh.n = f3s(0.f);
            if (hit_valid(si)) complete_surface(sc, si, ray.d, sf);
            DirSample ds;                                                                    // render/records.h:168-174
            ds.p = si.p; ds.n = sf.sh.n; ds.d = si.p - ref_p; ds.dist = norm(ds.d); ds.d = ds.d / ds.dist;
            if (!hit_valid(si)) ds.d = -sf.wi;
            ds.emitter = emitter; ds.pdf = 0.f; ds.delta = false;
            emitter_val = emitter_eval(sc, emitter, sf.wi.z, cx);
            emitter_pdf = pdf_emitter_direction(sc, ref_p, ds);
            active = false; active_surface = false; active_medium = false;
        }
        active_surface = active_surface && hit_valid(si) && !active_medium;
        if (active_surface) {
            const DShape &s = sc.shapes[si.shape];
            transmittance = transmittance * null_transmission(sc, s);
            ray = spawn_ray(si.p, ray.d);
        }
        needs_intersection = needs_intersection || active_surface;
        active = active && (active_medium || active_surface) && any_nonzero(transmittance);
        if (active_surface && sc.shapes[si.shape].is_medium_transition) medium = target_medium(sc.shapes[si.shape], hit_geo_normal(sc, si), ray.d);
    }
    return transmittance * emitter_val;
}

// integrators/volpath.cpp:38-257
template <bool COUNT>
DEV Spec volpath_sample(const DScene &sc, Pcg32 &rng, DRay ray, int medium, bool &valid_out, Counters &cnt, const SpecCtx &cx = SpecCtx()) {
    const uint32_t max_depth = (uint32_t) sc.integrator.max_depth, rr_depth = (uint32_t) sc.integrator.rr_depth;
    const bool hide_emitters = sc.integrator.hide_emitters != 0;
    bool valid_ray = !hide_emitters && sc.environment >= 0;
    float eta = 1.f;
    Spec throughput = spec_s(1.f), result = spec_s(0.f);
    bool active = true, specular_chain = !hide_emitters;
    uint32_t depth = 0;
#if MTS_SPEC_N == 3
    uint32_t channel = sc.integrator.monochrome ? 0u : (uint32_t) pm_min(rng.next_1d() * 3.f, 2.f);   // volpath.cpp:63-67 (rgb variants only)
#else
    const uint32_t channel = 0;                                                                        // volpath.cpp:63-67: no draw outside the rgb variants
#endif
    Hit si; si.t = pm_inf(); si.shape = -1; si.prim = 0; si.p = f3s(0.f); si.uv.x = si.uv.y = 0.f;
    bool needs_intersection = true;
    for (;;) {
        active = active && any_nonzero(throughput);
        float q = pm_min(hmax(throughput) * (eta * eta), .95f);
        bool perform_rr = depth > rr_depth;
        active = active && (rng.next_1d() < q || !perform_rr);
        if (perform_rr) throughput = throughput * pm_rcp(q);
        bool exceeded_max_depth = depth >= max_depth;
        if (!active || exceeded_max_depth) break;
        if (COUNT) cnt.n_iter++;
        bool active_medium = medium >= 0, active_surface = !active_medium;
        bool act_null_scatter = false, act_medium_scatter = false, escaped_medium = false;
        MediumSample mi; mi.t = pm_inf();
        bool is_spectral = false, not_spectral = false;
        if (active_medium) {
            const DMedium &m = sc.media[medium];
            is_spectral = m.has_spectral_extinction != 0; not_spectral = !is_spectral;
            mi = medium_sample_interaction<COUNT>(sc, medium, ray, rng.next_1d(), channel, cnt, cx);
            if (m.is_homogeneous && ms_valid(mi)) ray.maxt = mi.t;
            if (needs_intersection) si = ray_intersect(sc, ray);
            needs_intersection = false;
            if (si.t < mi.t) mi.t = pm_inf();
            if (is_spectral) {
                float t = pm_min(mi.t, si.t) - mi.mint;                                       // medium.cpp:77-89
                Spec tr = transmittance_exp(t, mi.combined);
                Spec free_flight_pdf = si.t < mi.t ? tr : tr * mi.combined;
                float tr_pdf = pick(free_flight_pdf, channel);
                throughput = throughput * (tr_pdf > 0.f ? tr / tr_pdf : spec_s(0.f));
            }
            escaped_medium = !ms_valid(mi);
            active_medium = ms_valid(mi);
            bool null_scatter = rng.next_1d() >= pick(mi.sigma_t, channel) / pick(mi.combined, channel);
            act_null_scatter = null_scatter && active_medium;
            act_medium_scatter = !act_null_scatter && active_medium;
            if (is_spectral && act_null_scatter)
                throughput = throughput * (mi.sigma_n * pick(mi.combined, channel) / pick(mi.sigma_n, channel));
            if (act_medium_scatter) depth += 1;
        }
        active = active && depth < max_depth;
        act_medium_scatter = act_medium_scatter && active;
        if (act_null_scatter) { ray.o = mi.p; ray.mint = 0.f; si.t = si.t - mi.t; }
        if (act_medium_scatter) {
            if (is_spectral) throughput = throughput * (mi.sigma_s * pick(mi.combined, channel) / pick(mi.sigma_t, channel));
            if (not_spectral) throughput = throughput * (mi.sigma_s / mi.sigma_t);
            const DMedium &m = sc.media[medium];
            bool sample_emitters = m.sample_emitters != 0;
            valid_ray = true;
            specular_chain = !sample_emitters;
            F3 wi = -ray.d;
            if (sample_emitters) {
                DirSample ds;
                Spec emitted = volpath_sample_emitter<COUNT>(sc, mi.p, true, rng, medium, channel, ds, cnt, cx);
                float phase_val = phase_eval(sc, m.phase, wi, mi.p, ds.d, cx);
                result = result + throughput * phase_val * emitted;
            }
            float s1 = rng.next_1d(); F2 s2 = rng.next_2d();                                  // left-to-right (SURVEY.md 8(a'))
            F3 wo = phase_sample(sc, m.phase, make_frame(ray.d), mi.p, s1, s2, cx);          // mi.sh_frame = Frame3f(ray.d), medium.cpp:42
            ray = spawn_ray(mi.p, wo); ray.mint = 0.0f;
            needs_intersection = true;
        }
        active_surface = active_surface || escaped_medium;
        bool intersect = active_surface && needs_intersection;
        if (intersect) si = ray_intersect(sc, ray);
        Surf sf; sf.wi = -ray.d;
        if (active_surface && hit_valid(si)) complete_surface(sc, si, ray.d, sf);
        if (active_surface) {
            int emitter = hit_emitter(sc, si);
            if (specular_chain && emitter >= 0) result = result + throughput * emitter_eval(sc, emitter, sf.wi.z, cx);
        }
        active_surface = active_surface && hit_valid(si);
        if (active_surface) {
            const DShape &shape = sc.shapes[si.shape];
            const DBsdf &bsdf = sc.bsdfs[shape.bsdf];
            bool active_e = (bsdf.flags & F_Smooth) != 0 && (depth + 1 < max_depth);
            if (active_e) {
                DirSample ds;
                Spec emitted = volpath_sample_emitter<COUNT>(sc, si.p, false, rng, medium, channel, ds, cnt, cx);
                F3 wo = to_local(sf.sh, ds.d);
                Spec bsdf_val = bsdf_eval(bsdf, sf.wi, wo, cx, shape.bsdf);
                float bpdf = bsdf_pdf(bsdf, sf.wi, wo, cx, shape.bsdf);
                result = result + throughput * bsdf_val * mis_weight(ds.pdf, ds.delta ? 0.f : bpdf) * emitted;
            }
            float s1 = rng.next_1d(); F2 s2 = rng.next_2d();
            BSDFSample bs;
            Spec bsdf_val = bsdf_sample(bsdf, sf.wi, s1, s2, bs, cx, shape.bsdf);
            throughput = throughput * bsdf_val;
            eta *= bs.eta;
            ray = spawn_ray(si.p, to_world(sf.sh, bs.wo));
            needs_intersection = true;
            bool non_null_bsdf = !(bs.sampled_type & F_Null);
            if (non_null_bsdf) depth += 1;
            valid_ray = valid_ray || non_null_bsdf;
            specular_chain = specular_chain || (non_null_bsdf && (bs.sampled_type & F_Delta));
            specular_chain = specular_chain && !(bs.sampled_type & F_Smooth);
            bool add_emitter = !(bs.sampled_type & F_Delta) && any_nonzero(throughput) && (depth < max_depth);
            bool intersect2 = needs_intersection && add_emitter;
            Hit si_new = si;
            if (intersect2) si_new = ray_intersect(sc, ray);
            needs_intersection = needs_intersection && !intersect2;
            float emitter_pdf;
            Spec emitted = volpath_evaluate_direct_light<COUNT>(sc, si.p, rng, medium, ray, si_new, channel, add_emitter, emitter_pdf, cnt, cx);
            if (add_emitter && emitter_pdf != 0) result = result + mis_weight(bs.pdf, emitter_pdf) * throughput * emitted;
            if (shape.is_medium_transition) medium = target_medium(shape, sf.n, ray.d);
            if (intersect2) si = si_new;
        }
        active = active && (active_surface || active_medium);
    }
    valid_out = valid_ray;
    return result;
}

// ---------------------------------------------------------------- volpathmis
// integrators/volpathmis.cpp (nested formulation, statement for statement the CPU restatement in oracle/oracle.cpp).
// WeightMatrix (:66-69): n rows of n probability ratios with `use_spectral_mis` (default), n = array_size_v<UnpolarizedSpectrum> = 3 in
// the rgb variants and 4 in the spectral one; one row without (:38-46).  Outside the rgb variants index_spectrum is spec[0] (:74-84,
// `pick`) and `channel` stays 0 (:118-124).
DEV float sget(const Spec &a, int i) {
#if MTS_SPEC_N == 3
    return i == 0 ? a.x : (i == 1 ? a.y : a.z);
#else
    return i == 0 ? a.x : (i == 1 ? a.y : (i == 2 ? a.z : a.w));
#endif
}
DEV float mw_fin(float x) { return pm_isfinite(x) ? x : 0.f; }
DEV float mw_nan0(float x) { return x != x ? 0.f : x; }
#if MTS_SPEC_N == 3
DEV Spec spec_map_fin(Spec a) { return f3(mw_fin(a.x), mw_fin(a.y), mw_fin(a.z)); }
DEV Spec spec_map_nan0(Spec a) { return f3(mw_nan0(a.x), mw_nan0(a.y), mw_nan0(a.z)); }
DEV Spec spec_div_s(Spec p, float f) { float r = pm_rcp(f); return f3(p.x * r, p.y * r, p.z * r); } // spectrum / coefficient (:456): reciprocal, then multiply (enoki array / scalar)
DEV Spec spec_s_div(float p, Spec f) { return f3(p / f.x, p / f.y, p / f.z); }
DEV Spec spec_rcp(Spec f) { return f3(pm_rcp(f.x), pm_rcp(f.y), pm_rcp(f.z)); }                  // 1.f / spectrum: the quotients, through pm_rcp
DEV Spec spec_of(float a, float b, float c, float) { return f3(a, b, c); }
DEV float spec_hsum(Spec a) { return (a.x + a.y) + a.z; }
DEV float spec_hmin_abs(Spec a) { return pm_min(pm_min(pm_abs(a.x), pm_abs(a.y)), pm_abs(a.z)); }
#else
DEV Spec spec_map_fin(Spec a) { return spec4(mw_fin(a.x), mw_fin(a.y), mw_fin(a.z), mw_fin(a.w)); }
DEV Spec spec_map_nan0(Spec a) { return spec4(mw_nan0(a.x), mw_nan0(a.y), mw_nan0(a.z), mw_nan0(a.w)); }
DEV Spec spec_div_s(Spec p, float f) { float r = pm_rcp(f); return spec4(p.x * r, p.y * r, p.z * r, p.w * r); }
DEV Spec spec_s_div(float p, Spec f) { return spec4(p / f.x, p / f.y, p / f.z, p / f.w); }
DEV Spec spec_rcp(Spec f) { return spec4(pm_rcp(f.x), pm_rcp(f.y), pm_rcp(f.z), pm_rcp(f.w)); }
DEV Spec spec_of(float a, float b, float c, float d) { return spec4(a, b, c, d); }
DEV float spec_hsum(Spec a) { return (a.x + a.y) + (a.z + a.w); }                             // hsum of a 4-array: pairwise, as spec_hmean (dmath.h)
DEV float spec_hmin_abs(Spec a) { return pm_min(pm_min(pm_abs(a.x), pm_abs(a.y)), pm_min(pm_abs(a.z), pm_abs(a.w))); }
#endif
template <bool SPEC> struct MisWeights { Spec r[SPEC ? MTS_SPEC_N : 1]; };
template <bool SPEC> DEV MisWeights<SPEC> mw_full(float v) { MisWeights<SPEC> w; for (int i = 0; i < (SPEC ? MTS_SPEC_N : 1); ++i) w.r[i] = spec_s(v); return w; }
template <bool SPEC>
DEV void update_weights(MisWeights<SPEC> &w, Spec p, Spec f, uint32_t channel, bool active) {       // volpathmis.cpp:447-466
    if (SPEC) {
#pragma unroll
        for (int i = 0; i < MTS_SPEC_N; ++i) {
            Spec ratio = spec_map_fin(spec_div_s(p, sget(f, i)));
            ratio = ratio * w.r[i];
            if (active) w.r[i] = spec_map_nan0(ratio);
        }
    } else {
        float pdf = pick(p, channel);
        Spec ratio = w.r[0] * spec_s_div(pdf, f);
        if (active) w.r[0] = spec_map_fin(ratio);
    }
}
// update_weights(w, spec_s(p), spec_s(f)) when both operands carry one value in every channel (the majorant's transmittance, a grey medium):
// every entry of the ratio matrix is the same number, so it is formed once -- the same products, entry by entry, as the general form above.
template <bool SPEC>
DEV void update_weights_uniform(MisWeights<SPEC> &w, float p, float f) {
    if (SPEC) {
        const float ratio = mw_fin(p * pm_rcp(f));
#pragma unroll
        for (int i = 0; i < MTS_SPEC_N; ++i) w.r[i] = spec_map_nan0(spec_s(ratio) * w.r[i]);
    } else {
        w.r[0] = spec_map_fin(w.r[0] * spec_s(p / f));
    }
}
template <bool SPEC> DEV void update_weights(MisWeights<SPEC> &w, float p, Spec f, uint32_t c, bool a) { update_weights(w, spec_s(p), f, c, a); }
template <bool SPEC> DEV void update_weights(MisWeights<SPEC> &w, Spec p, float f, uint32_t c, bool a) { update_weights(w, p, spec_s(f), c, a); }
template <bool SPEC> DEV void update_weights(MisWeights<SPEC> &w, float p, float f, uint32_t c, bool a) { update_weights(w, spec_s(p), spec_s(f), c, a); }
template <bool SPEC>
DEV Spec mis_weight_w(const MisWeights<SPEC> &w) {                                               // volpathmis.cpp:468-481
    if (SPEC) {
        float o[4] = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
        for (int i = 0; i < MTS_SPEC_N; ++i) { float sum = spec_hsum(w.r[i]); o[i] = sum == 0.f ? 0.f : (float) MTS_SPEC_N / sum; }
        return spec_of(o[0], o[1], o[2], o[3]);
    }
    Spec a = w.r[0];
    return spec_hmin_abs(a) == 0.f ? spec_s(0.f) : spec_rcp(a);
}
template <bool SPEC>
DEV Spec mis_weight_w(const MisWeights<SPEC> &a, const MisWeights<SPEC> &b) {                    // volpathmis.cpp:484-498
    if (SPEC) {
        float o[4] = { 0.f, 0.f, 0.f, 0.f };
#pragma unroll
        for (int i = 0; i < MTS_SPEC_N; ++i) { float sum = spec_hsum(a.r[i] + b.r[i]); o[i] = sum == 0.f ? 0.f : (float) MTS_SPEC_N / sum; }
        return spec_of(o[0], o[1], o[2], o[3]);
    }
    Spec sum = a.r[0] + b.r[0];
    return spec_hmin_abs(sum) == 0.f ? spec_s(0.f) : spec_rcp(sum);
}

// volpathmis.cpp:330-445
template <bool COUNT, bool SPEC>
DEV Spec volpathmis_sample_emitter(const DScene &sc, F3 ref_p, bool is_medium_interaction, Pcg32 &rng, int medium, const MisWeights<SPEC> &p_over_f,
                                          uint32_t channel, MisWeights<SPEC> &nee_out, MisWeights<SPEC> &uni_out, DirSample &ds, Counters &cnt, const SpecCtx &cx = SpecCtx()) {
    MisWeights<SPEC> p_over_f_nee = p_over_f, p_over_f_uni = p_over_f;
    Spec emitter_sample_weight;
    ds = sample_emitter_direction(sc, ref_p, rng.next_2d(), false, emitter_sample_weight, cx);
    Spec emitter_val = emitter_sample_weight * ds.pdf;
    if (ds.pdf == 0.f) emitter_val = spec_s(0.f);
    bool active = ds.pdf != 0.f;
    update_weights(p_over_f_nee, ds.pdf, 1.0f, channel, active);
    if (!active) { nee_out = p_over_f_nee; uni_out = p_over_f_uni; return emitter_val; }
    DRay ray = spawn_ray(ref_p, ds.d);
    if (is_medium_interaction) ray.mint = 0.f;
    float total_dist = 0.f;
    Hit si; si.t = pm_inf(); si.shape = -1; si.prim = 0; si.p = f3s(0.f); si.uv.x = si.uv.y = 0.f;
    bool needs_intersection = true;
    while (active) {
        float remaining_dist = ds.dist * (1.f - MTS_SHADOW_EPSILON) - total_dist;
        ray.maxt = remaining_dist;
        active = active && remaining_dist > 0.f;
        if (!active) break;
        if (COUNT) cnt.n_nee_step++;
        bool escaped_medium = false, active_medium = medium >= 0, active_surface = !active_medium;
        if (active_medium) {
            const DMedium &m = sc.media[medium];
            MediumSample mi = medium_sample_interaction<COUNT>(sc, medium, ray, rng.next_1d(), channel, cnt, cx);
            if (m.is_homogeneous && ms_valid(mi)) ray.maxt = pm_min(mi.t, remaining_dist);
            if (needs_intersection) si = ray_intersect(sc, ray);
            if (si.t < mi.t) mi.t = pm_inf();
            needs_intersection = false;
            bool is_spectral = m.has_spectral_extinction != 0, not_spectral = !is_spectral;
            if (is_spectral) {
                float t = pm_min(remaining_dist, pm_min(mi.t, si.t)) - mi.mint;
                Spec tr = transmittance_exp(t, mi.combined);
                Spec free_flight_pdf = (si.t < mi.t || mi.t > remaining_dist) ? tr : tr * mi.combined;
                update_weights(p_over_f_nee, free_flight_pdf, tr, channel, true);
                update_weights(p_over_f_uni, free_flight_pdf, tr, channel, true);
            }
            if (mi.t > remaining_dist && ms_valid(mi)) total_dist = ds.dist;
            if (mi.t > remaining_dist) mi.t = pm_inf();
            escaped_medium = !ms_valid(mi);
            active_medium = ms_valid(mi);
            is_spectral = is_spectral && active_medium; not_spectral = not_spectral && active_medium;
            if (active_medium) {
                total_dist += mi.t;
                ray.o = mi.p; ray.mint = 0.f; si.t = si.t - mi.t;
                if (is_spectral) {
                    update_weights(p_over_f_nee, 1.f, mi.sigma_n, channel, true);
                    update_weights(p_over_f_uni, mi.sigma_n / mi.combined, mi.sigma_n, channel, true);
                }
                if (not_spectral) {
                    update_weights(p_over_f_nee, 1.f, mi.sigma_n / mi.combined, channel, true);
                    update_weights(p_over_f_uni, mi.sigma_n, mi.sigma_n, channel, true);
                }
            }
        }
        bool intersect = active_surface && needs_intersection;
        if (intersect) si = ray_intersect(sc, ray);
        active_surface = active_surface || escaped_medium;
        if (active_surface) total_dist += si.t;
        active_surface = active_surface && hit_valid(si) && active && !active_medium;
        if (active_surface) {
            Spec bsdf_val = null_transmission(sc, sc.shapes[si.shape]);
            update_weights(p_over_f_nee, 1.0f, bsdf_val, channel, true);
            update_weights(p_over_f_uni, 1.0f, bsdf_val, channel, true);
        }
        if (active_surface) ray = spawn_ray(si.p, ray.d);
        ray.maxt = remaining_dist;
        needs_intersection = needs_intersection || active_surface;
        if (SPEC) active = active && (active_medium || active_surface) && any_nonzero(mis_weight_w(p_over_f_uni));
        else active = active && (active_medium || active_surface) && (any_nonzero(p_over_f_uni.r[0]) || any_nonzero(p_over_f_nee.r[0]));
        if (active_surface && sc.shapes[si.shape].is_medium_transition) medium = target_medium(sc.shapes[si.shape], hit_geo_normal(sc, si), ray.d);
    }
    nee_out = p_over_f_nee; uni_out = p_over_f_uni;
    return emitter_val;
}

// volpathmis.cpp:86-328
template <bool COUNT, bool SPEC>
DEV Spec volpathmis_sample(const DScene &sc, Pcg32 &rng, DRay ray, int medium, bool &valid_out, Counters &cnt, const SpecCtx &cx = SpecCtx()) {
    const uint32_t max_depth = (uint32_t) sc.integrator.max_depth, rr_depth = (uint32_t) sc.integrator.rr_depth;
    const bool hide_emitters = sc.integrator.hide_emitters != 0;
    bool valid_ray = !hide_emitters && sc.environment >= 0;
    float eta = 1.f;
    Spec result = spec_s(0.f);
    bool active = true, specular_chain = !hide_emitters;
    uint32_t depth = 0;
    MisWeights<SPEC> p_over_f = mw_full<SPEC>(1.f), p_over_f_nee = mw_full<SPEC>(1.f);
#if MTS_SPEC_N == 3
    uint32_t channel = sc.integrator.monochrome ? 0u : (uint32_t) pm_min(rng.next_1d() * 3.f, 2.f);   // volpathmis.cpp:120-124
#else
    const uint32_t channel = 0;                                                               // :120-124: a draw in the rgb variants only
#endif
    Hit si; si.t = pm_inf(); si.shape = -1; si.prim = 0; si.p = f3s(0.f); si.uv.x = si.uv.y = 0.f;
    bool needs_intersection = true, last_event_was_null = false;
    F3 last_scatter_p = f3s(0.f);                                                             // last_scatter_event: only .p is read
    for (;;) {
        Spec mis_throughput = mis_weight_w(p_over_f);
        float q = pm_min(hmax(mis_throughput) * (eta * eta), .95f);
        bool perform_rr = active && !last_event_was_null && (depth > rr_depth);
        active = active && !(rng.next_1d() >= q && perform_rr);
        update_weights(p_over_f, q, 1.0f, channel, perform_rr);
        last_event_was_null = false;
        bool exceeded_max_depth = depth >= max_depth;
        active = active && !exceeded_max_depth;
        active = active && any_nonzero(mis_weight_w(p_over_f));
        if (!active) break;
        if (COUNT) cnt.n_iter++;
        bool active_medium = active && medium >= 0, active_surface = active && !active_medium;
        bool act_null_scatter = false, act_medium_scatter = false, escaped_medium = false;
        MediumSample mi; mi.t = pm_inf();
        bool is_spectral = active_medium, not_spectral = false;
        if (active_medium) { is_spectral = is_spectral && sc.media[medium].has_spectral_extinction != 0; not_spectral = !is_spectral && active_medium; }
        if (active_medium) {
            const DMedium &m = sc.media[medium];
            mi = medium_sample_interaction<COUNT>(sc, medium, ray, rng.next_1d(), channel, cnt, cx);
            if (m.is_homogeneous && ms_valid(mi)) ray.maxt = mi.t;
            if (needs_intersection) si = ray_intersect(sc, ray);
            needs_intersection = false;
            if (si.t < mi.t) mi.t = pm_inf();
            if (is_spectral) {
                float t = pm_min(mi.t, si.t) - mi.mint;                                       // medium.cpp:77-89
                Spec tr = transmittance_exp(t, mi.combined);
                Spec free_flight_pdf = si.t < mi.t ? tr : tr * mi.combined;
                update_weights(p_over_f, free_flight_pdf, tr, channel, true);
                update_weights(p_over_f_nee, free_flight_pdf, tr, channel, true);
            }
            escaped_medium = !ms_valid(mi);
            active_medium = ms_valid(mi);
            is_spectral = is_spectral && active_medium; not_spectral = not_spectral && active_medium;
        }
        if (active_medium) {
            const int mi_medium = medium;
            bool null_scatter = rng.next_1d() >= pick(mi.sigma_t, channel) / pick(mi.combined, channel);
            act_null_scatter = null_scatter;
            act_medium_scatter = !act_null_scatter;
            if (act_medium_scatter) { depth += 1; last_scatter_p = mi.p; }
            const DMedium &m = sc.media[mi_medium];
            bool sample_emitters = m.sample_emitters != 0;
            active = active && depth < max_depth;
            act_medium_scatter = act_medium_scatter && active;
            specular_chain = specular_chain && !(act_medium_scatter && sample_emitters);
            if (act_null_scatter) {
                if (is_spectral) {
                    update_weights(p_over_f, mi.sigma_n / mi.combined, mi.sigma_n, channel, true);
                    update_weights(p_over_f_nee, 1.0f, mi.sigma_n, channel, true);
                }
                if (not_spectral) {
                    update_weights(p_over_f, mi.sigma_n, mi.sigma_n, channel, true);
                    update_weights(p_over_f_nee, 1.0f, mi.sigma_n / mi.combined, channel, true);
                }
                ray.o = mi.p; ray.mint = 0.f; si.t = si.t - mi.t;
            }
            if (act_medium_scatter) {
                if (is_spectral) update_weights(p_over_f, mi.sigma_t / mi.combined, mi.sigma_s, channel, true);
                if (not_spectral) update_weights(p_over_f, mi.sigma_t, mi.sigma_s, channel, true);
                valid_ray = true;
                F3 wi = -ray.d;
                if (sample_emitters) {
                    MisWeights<SPEC> nee_end, uni_end; DirSample ds;
                    Spec emitted = volpathmis_sample_emitter<COUNT, SPEC>(sc, mi.p, true, rng, medium, p_over_f, channel, nee_end, uni_end, ds, cnt, cx);
                    float phase_val = phase_eval(sc, m.phase, wi, mi.p, ds.d, cx);
                    update_weights(nee_end, 1.0f, phase_val, channel, true);
                    update_weights(uni_end, ds.delta ? 0.f : phase_val, phase_val, channel, true);
                    result = result + mis_weight_w(nee_end, uni_end) * emitted;
                }
                p_over_f_nee = p_over_f;
                float s1 = rng.next_1d(); F2 s2 = rng.next_2d();                              // left-to-right (SURVEY.md 8(a'))
                float phase_pdf;
                F3 wo = phase_sample_pdf(sc, m.phase, make_frame(ray.d), mi.p, s1, s2, phase_pdf, cx);
                ray = spawn_ray(mi.p, wo); ray.mint = 0.0f;
                needs_intersection = true;
                update_weights(p_over_f, phase_pdf, phase_pdf, channel, true);
                update_weights(p_over_f_nee, 1.f, phase_pdf, channel, true);
            }
        }
        active_surface = active_surface || escaped_medium;
        bool intersect = active_surface && needs_intersection;
        if (intersect) si = ray_intersect(sc, ray);
        Surf sf; sf.wi = -ray.d; sf.sh.n = f3s(0.f);
        if (active_surface && hit_valid(si)) complete_surface(sc, si, ray.d, sf);
        if (active_surface) {
            bool ray_from_camera = depth == 0;
            bool count_direct = ray_from_camera || specular_chain;
            int emitter = hit_emitter(sc, si);
            bool active_e = emitter >= 0 && !(depth == 0 && hide_emitters);
            if (active_e) {
                if (!count_direct) {
                    DirSample ds;                                                             // records.h:168-174
                    ds.p = si.p; ds.n = sf.sh.n; ds.d = si.p - last_scatter_p; ds.dist = norm(ds.d); ds.d = ds.d / ds.dist;
                    if (!hit_valid(si)) ds.d = -sf.wi;
                    ds.emitter = emitter; ds.pdf = 0.f; ds.delta = false;
                    float emitter_pdf = pdf_emitter_direction(sc, last_scatter_p, ds);
                    update_weights(p_over_f_nee, emitter_pdf, 1.f, channel, true);
                }
                Spec emitted = emitter_eval(sc, emitter, sf.wi.z, cx);
                Spec contrib = count_direct ? mis_weight_w(p_over_f) * emitted : mis_weight_w(p_over_f, p_over_f_nee) * emitted;
                result = result + contrib;
            }
        }
        active_surface = active_surface && hit_valid(si);
        if (active_surface) {
            const DShape &shape = sc.shapes[si.shape];
            const DBsdf &bsdf = sc.bsdfs[shape.bsdf];
            bool active_e = (bsdf.flags & F_Smooth) != 0 && (depth + 1 < max_depth);
            if (active_e) {
                MisWeights<SPEC> nee_end, uni_end; DirSample ds;
                Spec emitted = volpathmis_sample_emitter<COUNT, SPEC>(sc, si.p, false, rng, medium, p_over_f, channel, nee_end, uni_end, ds, cnt, cx);
                F3 wo_local = to_local(sf.sh, ds.d);
                Spec bsdf_val = bsdf_eval(bsdf, sf.wi, wo_local, cx, shape.bsdf);
                float bpdf = bsdf_pdf(bsdf, sf.wi, wo_local, cx, shape.bsdf);
                update_weights(nee_end, 1.0f, bsdf_val, channel, true);
                update_weights(uni_end, ds.delta ? 0.f : bpdf, bsdf_val, channel, true);
                result = result + mis_weight_w(nee_end, uni_end) * emitted;
            }
            float s1 = rng.next_1d(); F2 s2 = rng.next_2d();
            BSDFSample bs;
            Spec bsdf_weight = bsdf_sample(bsdf, sf.wi, s1, s2, bs, cx, shape.bsdf);
            bool invalid_bsdf_sample = bs.pdf == 0.f;
            active_surface = active_surface && bs.pdf > 0.f;
            if (active_surface) eta *= bs.eta;
            DRay bsdf_ray = spawn_ray(si.p, to_world(sf.sh, bs.wo));
            if (active_surface) { ray = bsdf_ray; needs_intersection = true; }
            bool non_null_bsdf = active_surface && !(bs.sampled_type & F_Null);
            valid_ray = valid_ray || non_null_bsdf || invalid_bsdf_sample;
            specular_chain = specular_chain || (non_null_bsdf && (bs.sampled_type & F_Delta));
            specular_chain = specular_chain && !(active_surface && (bs.sampled_type & F_Smooth));
            if (non_null_bsdf) { depth += 1; last_scatter_p = si.p; }
            if (non_null_bsdf) p_over_f_nee = p_over_f;
            update_weights(p_over_f, bs.pdf, bsdf_weight * bs.pdf, channel, active_surface);
            update_weights(p_over_f_nee, 1.f, bsdf_weight * bs.pdf, channel, non_null_bsdf);
            if (active_surface && shape.is_medium_transition) medium = target_medium(shape, sf.n, ray.d);
        }
        active = active && (active_surface || active_medium);
    }
    valid_out = valid_ray;
    return result;
}

// ---------------------------------------------------------------- path
// integrators/path.cpp:100-211
template <bool COUNT>
DEV Spec path_sample(const DScene &sc, Pcg32 &rng, DRay ray, bool &valid_out, Counters &cnt, const SpecCtx &cx = SpecCtx()) {
    const int max_depth = sc.integrator.max_depth, rr_depth = sc.integrator.rr_depth;
    float eta = 1.f, emission_weight = 1.f;
    Spec throughput = spec_s(1.f), result = spec_s(0.f);
    bool active = true;
    Hit si = ray_intersect(sc, ray);
    bool valid_ray = hit_valid(si);
    int emitter = hit_emitter(sc, si);
    for (int depth = 1;; ++depth) {
        if (COUNT) cnt.n_iter++;
        Surf sf; sf.wi = -ray.d;
        if (hit_valid(si)) complete_surface(sc, si, ray.d, sf);
        if (emitter >= 0 && active) result = result + emission_weight * throughput * emitter_eval(sc, emitter, sf.wi.z, cx);
        active = active && hit_valid(si);
        if (depth > rr_depth) {
            float q = pm_min(hmax(throughput) * (eta * eta), .95f);
            active = active && rng.next_1d() < q;
            throughput = throughput * pm_rcp(q);
        }
        if ((uint32_t) depth >= (uint32_t) max_depth || !active) break;
        const int bsdf_id = sc.shapes[si.shape].bsdf;
        const DBsdf &bsdf = sc.bsdfs[bsdf_id];
        bool active_e = (bsdf.flags & F_Smooth) != 0;
        if (active_e) {
            Spec emitter_val;
            DirSample ds = sample_emitter_direction(sc, si.p, rng.next_2d(), true, emitter_val, cx);
            active_e = active_e && ds.pdf != 0.f;
            F3 wo = to_local(sf.sh, ds.d);
            Spec bsdf_val = bsdf_eval(bsdf, sf.wi, wo, cx, bsdf_id);
            float bpdf = bsdf_pdf(bsdf, sf.wi, wo, cx, bsdf_id);
            float mis = ds.delta ? 1.f : mis_weight(ds.pdf, bpdf);
            if (active_e) result = result + mis * throughput * bsdf_val * emitter_val;
        }
        float s1 = rng.next_1d(); F2 s2 = rng.next_2d();
        BSDFSample bs;
        Spec bsdf_val = bsdf_sample(bsdf, sf.wi, s1, s2, bs, cx, bsdf_id);
        throughput = throughput * bsdf_val;
        active = active && any_nonzero(throughput);
        if (!active) break;
        eta *= bs.eta;
        F3 ref_p = si.p;
        ray = spawn_ray(si.p, to_world(sf.sh, bs.wo));
        Hit si_bsdf = ray_intersect(sc, ray);
        emitter = hit_emitter(sc, si_bsdf);
        if (emitter >= 0) {
            Surf sb; sb.wi = -ray.d; sb.sh.n = f3s(0.f);
            if (hit_valid(si_bsdf)) complete_surface(sc, si_bsdf, ray.d, sb);
            DirSample ds;                                                                    // render/records.h:168-174
            ds.p = si_bsdf.p; ds.n = sb.sh.n; ds.d = si_bsdf.p - ref_p; ds.dist = norm(ds.d); ds.d = ds.d / ds.dist;
            if (!hit_valid(si_bsdf)) ds.d = -sb.wi;
            ds.emitter = emitter; ds.pdf = 0.f; ds.delta = false;
            float emitter_pdf = !(bs.sampled_type & F_Delta) ? pdf_emitter_direction(sc, ref_p, ds) : 0.f;
            emission_weight = mis_weight(bs.pdf, emitter_pdf);
        }
        si = si_bsdf;
    }
    valid_out = valid_ray;
    return result;
}

// SamplingIntegrator::sample of the configured integrator (nested formulations)
// INTEG: -1 = decided at run time from the scene record (the probe kernel), otherwise the integrator is fixed at compile time
// (NI_*: one render-kernel instantiation each, so that `path` does not carry the registers of the volumetric integrators)
enum { NI_ANY = -1, NI_PATH = 0, NI_VOLPATH = 1, NI_VOLPATHMIS = 2, NI_VOLPATHMIS_NOSPEC = 3 };
template <bool COUNT, int INTEG = NI_ANY>
DEV Spec integrator_sample(const DScene &sc, Pcg32 &rng, DRay ray, int medium, bool &valid, Counters &cnt, const SpecCtx &cx = SpecCtx()) {
    if (INTEG == NI_PATH) return path_sample<COUNT>(sc, rng, ray, valid, cnt, cx);
    if (INTEG == NI_VOLPATH) return volpath_sample<COUNT>(sc, rng, ray, medium, valid, cnt, cx);
    if (INTEG == NI_VOLPATHMIS) return volpathmis_sample<COUNT, true>(sc, rng, ray, medium, valid, cnt, cx);
    if (INTEG == NI_VOLPATHMIS_NOSPEC) return volpathmis_sample<COUNT, false>(sc, rng, ray, medium, valid, cnt, cx);
    if (sc.integrator.type == MTS_INTEGRATOR_VOLPATH) return volpath_sample<COUNT>(sc, rng, ray, medium, valid, cnt, cx);
    if (sc.integrator.type == MTS_INTEGRATOR_VOLPATHMIS)
        return sc.integrator.use_spectral_mis ? volpathmis_sample<COUNT, true>(sc, rng, ray, medium, valid, cnt, cx)
                                              : volpathmis_sample<COUNT, false>(sc, rng, ray, medium, valid, cnt, cx);
    return path_sample<COUNT>(sc, rng, ray, valid, cnt, cx);
}

// ---------------------------------------------------------------- film
#if MTS_SPEC_N != 3         // rgb: splat_sample_t (volpath_flat.h)
// ImageBlock::put of the five film values X, Y, Z, alpha, weight of one finished sample (librender/imageblock.cpp:79-172).
// `own` receives the samples that land in the lane's own pixel: either register accumulators (per-lane kernels) or the pixel's film
// entry itself, updated with float atomics in sample order.
// With AOV channels (nbins / bins) a sample carries NA more values; they go to the film entry's channels 5 .. 5 + NA with the same filter
// weights, by atomics (one path writes them in sample order, so the sum is the block's), and the block only warns about non-finite
// values then (integrator.cpp:114-116: warn_negative = !has_aovs).
template <bool OWN_ATOMIC>
DEV void splat_values_t(const DScene &sc, const DBlock &blk, uint32_t lx, uint32_t ly, F2 position_sample, const float v[5],
                        MTS_GLOBAL_AS float *film, float *own, const float *aov = nullptr, int NA = 0) {
    const DSensor &se = sc.sensor;
    const int C = sc.film_channels;
    bool ok = true;                                             // imageblock.cpp:85-109: invalid samples are dropped
    for (int k = 0; k < 5; ++k) ok = ok && (NA > 0 || v[k] >= -1e-5f) && pm_isfinite(v[k]);
    for (int k = 0; k < NA; ++k) ok = ok && pm_isfinite(aov[k]);
    if (!ok) return;
    const DRFilter &rf = se.rfilter;
    const int border = rf.border_size;
    const int sx = blk.sx + 2 * border, sy = blk.sy + 2 * border;
    float posx = position_sample.x - ((float) (blk.ox - border) + .5f), posy = position_sample.y - ((float) (blk.oy - border) + .5f);
    if (rf.radius > 0.5f + MTS_RAY_EPSILON) {
        int lox = max((int) pm_ceil(posx - rf.radius), 0), loy = max((int) pm_ceil(posy - rf.radius), 0);
        int hix = min((int) pm_floor(posx + rf.radius), sx - 1), hiy = min((int) pm_floor(posy + rf.radius), sy - 1);
        uint32_t n = (uint32_t) pm_ceil((rf.radius - 2.f * MTS_RAY_EPSILON) * 2.f);
        float basex = (float) lox - posx, basey = (float) loy - posy;
        for (uint32_t yr = 0; yr < n; ++yr) {
            int y = loy + (int) yr;
            if (y > hiy) break;
            float wy = as_global(rf.values)[min((int) pm_abs((basey + (float) yr) * rf.scale_factor), 31)];     // eval_discretized, core/rfilter.h:62-65
            int fy = blk.oy - border + y - se.crop_y;
            for (uint32_t xr = 0; xr < n; ++xr) {
                int x = lox + (int) xr;
                if (x > hix) break;
                float wx = as_global(rf.values)[min((int) pm_abs((basex + (float) xr) * rf.scale_factor), 31)];
                float weight = wy * wx;
                int fx = blk.ox - border + x - se.crop_x;
                if (fx >= 0 && fy >= 0 && fx < se.crop_w && fy < se.crop_h) {                         // film clipping, imageblock.cpp:49-77
                    float *dst = (float *) (film + (size_t) C * ((size_t) fy * se.crop_w + fx));
                    for (int k = 0; k < 5; ++k) atomicAdd(dst + k, v[k] * weight);
                    for (int k = 0; k < NA; ++k) atomicAdd(dst + 5 + k, aov[k] * weight);
                }
            }
        }
    } else {
        int lox = (int) pm_ceil(posx - .5f), loy = (int) pm_ceil(posy - .5f);
        const bool inside = lox >= 0 && loy >= 0 && lox < sx && loy < sy;
        if (lox == (int) lx && loy == (int) ly) {
            if (OWN_ATOMIC) { for (int k = 0; k < 5; ++k) atomicAdd(own + k, v[k]); }
            else { for (int k = 0; k < 5; ++k) own[k] += v[k]; }
        } else if (inside) {
            float *dst = (float *) (film + (size_t) C * ((size_t) (blk.oy + loy - se.crop_y) * se.crop_w + (blk.ox + lox - se.crop_x)));
            for (int k = 0; k < 5; ++k) atomicAdd(dst + k, v[k]);
        }
        if (NA > 0 && inside) {
            float *dst = (float *) (film + (size_t) C * ((size_t) (blk.oy + loy - se.crop_y) * se.crop_w + (blk.ox + lox - se.crop_x)));
            for (int k = 0; k < NA; ++k) atomicAdd(dst + 5 + k, aov[k]);
        }
    }
}

#endif
// ---------------------------------------------------------------- wavelengths / colour (spectral variants)
#if MTS_SPEC_N != 3
// math::sample_shifted (core/math.h:419-442) + sample_wavelength -> sample_rgb_spectrum -> sample_uniform_spectrum (core/spectrum.h:248-252,
// 266-285,305-314): with MTS_WAVELENGTH_MIN / MAX = 280 / 2400 the "rgb" importance sampling falls back to the uniform one, which the
// reference writes over the CIE range, 360 .. 830 nm; the weight (inverse pdf) is 470 for every wavelength.
#define MTS_CIE_MIN 360.f
#define MTS_CIE_MAX 830.f
#define MTS_CIE_SAMPLES 95
DEV Spec sample_wavelengths(float sample, float &weight) {
    float v[4];
    for (int k = 0; k < 4; ++k) { float x = sample + (float) k / 4.f; if (x > 1.f) x -= 1.f; v[k] = x * (MTS_CIE_MAX - MTS_CIE_MIN) + MTS_CIE_MIN; }
    weight = MTS_CIE_MAX - MTS_CIE_MIN;
    return spec4(v[0], v[1], v[2], v[3]);
}
// The sensor's "srf" draws the wavelengths instead (perspective.cpp:173-182, radiancemeter.cpp:116-124): Texture::sample_spectrum of a
// uniform spectrum (uniform.cpp:92-100) or of a discrete one (discrete.cpp:124-133 -> DiscreteDistribution::sample, distr_1d.h:141-151)
DEV Spec sample_wavelengths_srf(const DScene &sc, float sample, Spec &weight) {
    const DSpectrum r = sc.spectra[sc.srf];
    float v[4], wgt[4];
    for (int k = 0; k < 4; ++k) {
        float x = sample + (float) k / 4.f; if (x > 1.f) x -= 1.f;
        if (r.type == MTS_SPECTRUM_UNIFORM) { v[k] = r.lambda_min + (r.lambda_max - r.lambda_min) * x; wgt[k] = r.value * (r.lambda_max - r.lambda_min); }
        else {
            const MTS_GLOBAL_AS float *cdf = as_global(r.cdf);
            const float value = x * r.cdf_sum;
            uint32_t start = r.valid_x, end = r.valid_y, iterations = 0;                       // enoki::binary_search: first index with !(cdf[i] < value)
            if (start < end) { uint32_t diff = end - start; iterations = 1; while (diff >>= 1) iterations++; }
            for (uint32_t i = 0; i < iterations; ++i) {
                const uint32_t middle = (start + end) >> 1;
                if (cdf[middle] < value) start = min(middle + 1u, end); else end = middle;
            }
            v[k] = as_global(r.wavelengths)[start]; wgt[k] = as_global(r.values)[start];
        }
    }
    weight = spec4(wgt[0], wgt[1], wgt[2], wgt[3]);
    return spec4(v[0], v[1], v[2], v[3]);
}
// The weights sample_wavelengths_srf returned, recovered from the wavelengths themselves (the regrouping machine keeps a sample's
// wavelengths in its hot state, not its weights): a uniform response has one weight; a discrete one is looked up by wavelength (the
// host sends discrete response functions with repeated wavelengths to the per-lane kernel).
DEV Spec srf_weights_of(const DScene &sc, Spec wl) {
    const DSpectrum r = sc.spectra[sc.srf];
    if (r.type == MTS_SPECTRUM_UNIFORM) return spec_s(r.value * (r.lambda_max - r.lambda_min));
    const MTS_GLOBAL_AS float *wavelengths = as_global(r.wavelengths), *values = as_global(r.values);
    const float w4[4] = { wl.x, wl.y, wl.z, wl.w }; float wgt[4];
    for (int k = 0; k < 4; ++k) {
        int lo = 0, hi = r.count - 1;                           // first index with wavelengths[i] >= w4[k]
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (wavelengths[mid] < w4[k]) lo = mid + 1; else hi = mid; }
        wgt[k] = values[lo];
    }
    return spec4(wgt[0], wgt[1], wgt[2], wgt[3]);
}
// nbins.cpp:100-125 / bins.cpp:88-110: per bin the sum of the wrapped integrator's result over the sample's wavelengths inside the bin,
// and their number; hsum of a 4-array: (x + y) + (z + w), as spec_hmean
DEV void bin_aovs(const DScene &sc, Spec L, Spec wl, int i, float &value, float &population) {
    const float lo = as_global(sc.bin_lo)[i], hi = as_global(sc.bin_hi)[i];
    const float w4[4] = { wl.x, wl.y, wl.z, wl.w }, l4[4] = { L.x, L.y, L.z, L.w };
    float val[4], pop[4];
    for (int k = 0; k < 4; ++k) {
        if (sc.bin_mode == 1) { const bool in = pm_abs(w4[k] - lo) <= hi; val[k] = in ? l4[k] : 0.f; pop[k] = in ? 1.f : 0.f; }
        else { const float w = (w4[k] >= lo && w4[k] <= hi) ? 1.f : 0.f; val[k] = w * l4[k]; pop[k] = w; }
    }
    value = (val[0] + val[1]) + (val[2] + val[3]);
    population = (pop[0] + pop[1]) + (pop[2] + pop[3]);
}
// The same splat for a scene with wavelength bins, without staging the AOV values in an array (the regrouping machine's NEW block):
// the bins are walked twice, once to test the values the reference tests (imageblock.cpp:85-109), once to add them.  Box filter: the
// AOV channels of the pixel the sample lands in; wider filters: every tap.  Same arithmetic and order as splat_values_t(aov, NA).
DEV void splat_values_bins(const DScene &sc, const DBlock &blk, uint32_t lx, uint32_t ly, F2 position_sample, const float v[5], Spec L_raw, Spec wl,
                           MTS_GLOBAL_AS float *film, float *own) {
    const DSensor &se = sc.sensor;
    const int C = sc.film_channels, nb = sc.bin_count;
    bool ok = true;
    for (int k = 0; k < 5; ++k) ok = ok && pm_isfinite(v[k]);
    for (int i = 0; i < nb; ++i) { float a, b; bin_aovs(sc, L_raw, wl, i, a, b); ok = ok && pm_isfinite(a) && pm_isfinite(b); }
    if (!ok) return;
    const DRFilter &rf = se.rfilter;
    const int border = rf.border_size;
    const int sx = blk.sx + 2 * border, sy = blk.sy + 2 * border;
    float posx = position_sample.x - ((float) (blk.ox - border) + .5f), posy = position_sample.y - ((float) (blk.oy - border) + .5f);
    if (rf.radius > 0.5f + MTS_RAY_EPSILON) {
        int lox = max((int) pm_ceil(posx - rf.radius), 0), loy = max((int) pm_ceil(posy - rf.radius), 0);
        int hix = min((int) pm_floor(posx + rf.radius), sx - 1), hiy = min((int) pm_floor(posy + rf.radius), sy - 1);
        uint32_t n = (uint32_t) pm_ceil((rf.radius - 2.f * MTS_RAY_EPSILON) * 2.f);
        float basex = (float) lox - posx, basey = (float) loy - posy;
        for (uint32_t yr = 0; yr < n; ++yr) {
            int y = loy + (int) yr;
            if (y > hiy) break;
            float wy = as_global(rf.values)[min((int) pm_abs((basey + (float) yr) * rf.scale_factor), 31)];
            int fy = blk.oy - border + y - se.crop_y;
            for (uint32_t xr = 0; xr < n; ++xr) {
                int x = lox + (int) xr;
                if (x > hix) break;
                float wx = as_global(rf.values)[min((int) pm_abs((basex + (float) xr) * rf.scale_factor), 31)];
                float weight = wy * wx;
                int fx = blk.ox - border + x - se.crop_x;
                if (fx >= 0 && fy >= 0 && fx < se.crop_w && fy < se.crop_h) {
                    float *dst = (float *) (film + (size_t) C * ((size_t) fy * se.crop_w + fx));
                    for (int k = 0; k < 5; ++k) atomicAdd(dst + k, v[k] * weight);
                    for (int i = 0; i < nb; ++i) { float a, b; bin_aovs(sc, L_raw, wl, i, a, b); atomicAdd(dst + 5 + 2 * i, a * weight); atomicAdd(dst + 6 + 2 * i, b * weight); }
                }
            }
        }
    } else {
        int lox = (int) pm_ceil(posx - .5f), loy = (int) pm_ceil(posy - .5f);
        const bool inside = lox >= 0 && loy >= 0 && lox < sx && loy < sy;
        float *dst = (float *) (film + (size_t) C * ((size_t) (blk.oy + loy - se.crop_y) * se.crop_w + (blk.ox + lox - se.crop_x)));
        if (lox == (int) lx && loy == (int) ly) { for (int k = 0; k < 5; ++k) own[k] += v[k]; }
        else if (inside) { for (int k = 0; k < 5; ++k) atomicAdd(dst + k, v[k]); }
        if (inside) for (int i = 0; i < nb; ++i) { float a, b; bin_aovs(sc, L_raw, wl, i, a, b); atomicAdd(dst + 5 + 2 * i, a); atomicAdd(dst + 6 + 2 * i, b); }
    }
}
// cie1931_xyz + spectrum_to_xyz (core/spectrum.h:148-178,210-217): XYZ = hmean(cmf(lambda) * value)
DEV void spectrum_to_xyz(const float *cie, Spec value, Spec wl, float xyz[3]) {
    const MTS_GLOBAL_AS float *T = as_global(cie);
    const float lam[4] = { wl.x, wl.y, wl.z, wl.w }, val[4] = { value.x, value.y, value.z, value.w };
    float cx[4], cy[4], cz[4];
    for (int k = 0; k < 4; ++k) {
        const float t = (lam[k] - MTS_CIE_MIN) * ((MTS_CIE_SAMPLES - 1) / (MTS_CIE_MAX - MTS_CIE_MIN));
        const bool active = lam[k] >= MTS_CIE_MIN && lam[k] <= MTS_CIE_MAX;
        const int i0 = min(max((int) t, 0), MTS_CIE_SAMPLES - 2), i1 = i0 + 1;
        const float w1 = t - (float) i0, w0 = 1.f - w1;
        cx[k] = active ? pm_fma(w0, T[i0], w1 * T[i1]) * val[k] : 0.f * val[k];
        cy[k] = active ? pm_fma(w0, T[MTS_CIE_SAMPLES + i0], w1 * T[MTS_CIE_SAMPLES + i1]) * val[k] : 0.f * val[k];
        cz[k] = active ? pm_fma(w0, T[2 * MTS_CIE_SAMPLES + i0], w1 * T[2 * MTS_CIE_SAMPLES + i1]) * val[k] : 0.f * val[k];
    }
    xyz[0] = spec_hmean(spec4(cx[0], cx[1], cx[2], cx[3])); xyz[1] = spec_hmean(spec4(cy[0], cy[1], cy[2], cy[3])); xyz[2] = spec_hmean(spec4(cz[0], cz[1], cz[2], cz[3]));
}
#endif

// ---------------------------------------------------------------- sensors
// Shape::ray_intersect (shape.cpp:344-352) of a stand-alone analytic shape, reduced to the hit point the distant sensors read
DEV bool shape_hit_point(const DScene &sc, const DShape &s, const DRay &ray, F3 &p) {
    F2 uv; Hit h; h.uv.x = h.uv.y = 0.f; h.shape = -1; h.prim = 0;
    if (s.type == MTS_SHAPE_RECTANGLE) h.t = rectangle_intersect(s.to_object.m, ray, uv);
    else if (s.type == MTS_SHAPE_DISK) h.t = disk_intersect(s.to_object.m, ray, uv);
    else h.t = sphere_intersect(s.center, s.radius, ray);
    if (h.t == pm_inf()) return false;
    hit_point(sc, s, ray, h);
    p = h.p;
    return true;
}
// sensors/perspective.cpp:210-252 ; sensors/distant.cpp:299-386
DEV DRay sensor_sample_ray(const DScene &sc, F2 position_sample, F2 aperture_sample, F3 &weight) {
    const DSensor &se = sc.sensor;
    if (se.type == MTS_SENSOR_PERSPECTIVE) {
        F3 near_p = mat_point(se.s2c, f3(position_sample.x + se.ppo[0], position_sample.y + se.ppo[1], 0.f));
        F3 d = normalize(near_p);
        float inv_z = pm_rcp(d.z);
        weight = f3s(1.f);
        return make_ray(mat_point_affine(se.to_world.m, f3s(0.f)), mat_vector(se.to_world.m, d), se.near_clip * inv_z, se.far_clip * inv_z);
    }
    if (se.type == MTS_SENSOR_MRADIANCEMETER || se.type == MTS_SENSOR_MDISTANT) {
        // Int32 sensor_index(position_sample.x() * m_sensor_count), mradiancemeter.cpp:146 / mdistant.cpp:231; the reference
        // gathers without a bounds check, here the index is clamped (position_sample.x can round up to 1)
        int index = (int) (position_sample.x * (float) se.multi_count);
        index = min(max(index, 0), se.multi_count - 1);
        const MTS_GLOBAL_AS float *gm = as_global(se.multi) + 16 * index;
        float m[16];
        for (int k = 0; k < 16; ++k) m[k] = gm[k];
        F3 d = mat_vector(m, f3(0.f, 0.f, 1.f));
        if (se.type == MTS_SENSOR_MRADIANCEMETER) {                                               // mradiancemeter.cpp:134-157
            weight = f3s(1.f);
            return make_ray(mat_point_affine(m, f3s(0.f)), d, MTS_RAY_EPSILON, pm_inf());
        }
        F3 o; float w = 1.f;                                                                      // mdistant.cpp:212-262
        if (se.target_type == MTS_DISTANT_TARGET_POINT) o = f3(se.target_point) - 2.f * d * se.bsphere_radius;
        else if (se.target_type == MTS_DISTANT_TARGET_SHAPE) {
            F3 tp, n; float pdf;
            shape_sample_position(mesh_tables(sc), se.target_shape, aperture_sample, tp, n, pdf);
            o = tp - 2.f * d * se.bsphere_radius;
            w = pm_rcp(pdf * se.target_area);
        } else {
            F2 offset = square_to_uniform_disk_concentric(aperture_sample);
            F3 perp_offset = mat_vector(m, f3(offset.x, offset.y, 0.f));
            o = f3(se.bsphere_center) + perp_offset * se.bsphere_radius - d * se.bsphere_radius;
        }
        weight = f3s(w);
        return make_ray(o, d, MTS_RAY_EPSILON, pm_inf());
    }
    if (se.type == MTS_SENSOR_DISTANTFLUX) {                                                      // distantflux.cpp:189-240
        F3 d = -mat_vector(se.to_world.m, square_to_uniform_hemisphere(position_sample));
        const F3 reference_normal = mat_vector(se.to_world.m, f3(0.f, 0.f, 1.f));                 // distantflux.cpp:185-186
        float w = dot(-d, reference_normal) / (MTS_INV_TWO_PI * (float) ((uint32_t) se.width * (uint32_t) se.height));
        F3 ray_target = f3(se.target_point);
        if (se.target_type == MTS_DISTANT_TARGET_SHAPE) {
            F3 n; float pdf;
            shape_sample_position(mesh_tables(sc), se.target_shape, aperture_sample, ray_target, n, pdf);
            w *= pm_rcp(pdf * se.target_area);
        } else if (se.target_type == MTS_DISTANT_TARGET_NONE) {
            F2 offset = square_to_uniform_disk_concentric(aperture_sample);
            F3 perp_offset = mat_vector(se.to_world.m, f3(offset.x, offset.y, 0.f));
            ray_target = f3(se.bsphere_center) + perp_offset * se.bsphere_radius;
        }
        F3 o = ray_target - d * 2.f * se.bsphere_radius;
        if (se.origin_type != 0) {                                                                // distantflux.cpp:244-252
            if (!shape_hit_point(sc, se.origin_shape, make_ray(ray_target, -d, MTS_RAY_EPSILON, pm_inf()), o)) { o = f3s(pm_nan()); w = 0.f; }
        }
        weight = f3s(w);
        return make_ray(o, d, MTS_RAY_EPSILON, pm_inf());
    }
    F3 v0 = f3(0.f, 0.f, 1.f);
    if (se.direction_type == 2) v0 = square_to_uniform_hemisphere(position_sample);
    else if (se.direction_type == 1) { float s, c; pm_sincos(MTS_PI * position_sample.x, &s, &c); v0.x = c; v0.z = s; }
    F3 d = se.flip_directions ? mat_vector(se.to_world.m, v0) : mat_vector(se.to_world.m, -v0);
    F3 ray_target = f3(se.target_point), o;
    float w = 1.f;
    if (se.target_type == MTS_DISTANT_TARGET_SHAPE) {
        F3 n; float pdf;
        shape_sample_position(mesh_tables(sc), se.target_shape, aperture_sample, ray_target, n, pdf);
        w = pm_rcp(pdf) * pm_rcp(se.target_area);                                                 // Spectrum / Float / Float: each a reciprocal-multiply
    } else if (se.target_type == MTS_DISTANT_TARGET_NONE) {
        F2 offset = square_to_uniform_disk_concentric(aperture_sample);
        F3 perp_offset = mat_vector(se.to_world.m, f3(offset.x, offset.y, 0.f));
        ray_target = f3(se.bsphere_center) + perp_offset * se.bsphere_radius;
        w = pm_rcp(dot(-d, f3(0.f, 0.f, 1.f)));
    }
    if (se.origin_type != 0) {                                                                    // distant.cpp:368-375
        if (!shape_hit_point(sc, se.origin_shape, make_ray(ray_target, -d, MTS_RAY_EPSILON, pm_inf()), o)) { o = f3s(pm_nan()); w = 0.f; }
    } else if (se.target_type == MTS_DISTANT_TARGET_NONE) o = ray_target - d * se.bsphere_radius;
    else o = ray_target - d * 2.f * se.bsphere_radius;
    weight = f3s(w);
    return make_ray(o, d, MTS_RAY_EPSILON, pm_inf());
}

} // inline namespace
} // namespace mtsamd
