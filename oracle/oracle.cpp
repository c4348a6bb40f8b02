// oracle.cpp -- TEST INFRASTRUCTURE, not product code.
//
// CPU restatement ("oracle") of the reference's scalar_rgb path / volpath render loop.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load liboracle.so;
// the product path (libmtsamd.so) never links, loads or calls anything in this directory.
//
// Pinning status: the reference can be neither compiled nor imported here
// (enoki and every other submodule are absent), so this restatement is pinned by the reference's
// own in-tree known answers -- TEA literals, phase-function closed forms, warp corner cases,
// spiral order, rectangle hit counts, the directional-emitter matrix, the distant-sensor analytic
// render -- all checked in tests/test_oracle_*.py.  The reference holds NO numeric pin for volpath
// itself (its Z-test references live in the absent resources/data submodule), so for volpath the
// status is "parity unpinned by in-tree data"; closed-form radiative-transfer cases
// (tests/test_oracle_transport.py) stand in.
//
// Every function cites the reference file:line it follows (relative to /root/reference).
// Semantics restated: scalar_rgb (Float = float, Spectrum = Color3f): `any_or<true>(m)` == m,
// `none_or<false>(m)` == !m, sampler draws happen regardless of the mask argument.
#include <cstdio>
#include <cstring>
#include <cmath>
#include <thread>
#include <atomic>
#include <mutex>
#include <chrono>
#include <xmmintrin.h>
#include "oracle_scene.h"
#include "oracle.h"

namespace orc {

struct Counters { uint64_t n_iter = 0, n_lookup = 0, n_nee_step = 0; };

// ---------------------------------------------------------------- sampler
// src/librender/sampler.cpp:83-96 (scalar branch), src/samplers/independent.cpp:73-82
struct Sampler {
    PCG32 rng; uint64_t base_seed;
    void seed(uint64_t seed_offset) { rng.seed(base_seed + seed_offset, PCG32_DEFAULT_STREAM); }
    float next_1d() { return rng.next_float32(); }
    P2 next_2d() { P2 p; p.x = next_1d(); p.y = next_1d(); return p; }
};

// ---------------------------------------------------------------- interactions
// include/mitsuba/render/interaction.h
struct SurfaceInteraction {
    float t; V3 p, n; P2 uv; Frame sh_frame; V3 dp_du, dp_dv, wi;
    int shape, prim_index;
    bool is_valid() const { return t != pm_inf(); }
    V3 to_world(V3 v) const { return sh_frame.to_world(v); }
    V3 to_local(V3 v) const { return sh_frame.to_local(v); }
};
struct MediumInteraction {
    float t; V3 p; Frame sh_frame; V3 wi;
    Spec sigma_s, sigma_n, sigma_t, combined_extinction; float mint; int medium;
    bool is_valid() const { return t != pm_inf(); }
};
// interaction.h:58-61
static inline Ray spawn_ray(V3 p, V3 d) { return make_ray(p, d, (1.f + hmax_abs(p)) * RayEpsilon, pm_inf()); }

struct PreliminaryIntersection { float t; P2 prim_uv; int prim_index, shape; };

// ---------------------------------------------------------------- primitives
// rectangle.cpp:139-155
static inline float rectangle_intersect(const Shape &s, const Ray &ray_, P2 *uv) {
    Ray ray = xf_ray_affine(s.to_object, ray_);
    float t = -ray.o.z * ray.d_rcp.z;
    V3 local = ray(t);
    bool active = t >= ray.mint && t <= ray.maxt && pm_abs(local.x) <= 1.f && pm_abs(local.y) <= 1.f;
    uv->x = local.x; uv->y = local.y;
    return active ? t : pm_inf();
}
// disk.cpp:136-153
static inline float disk_intersect(const Shape &s, const Ray &ray_, P2 *uv) {
    Ray ray = xf_ray_affine(s.to_object, ray_);
    float t = -ray.o.z * ray.d_rcp.z;
    V3 local = ray(t);
    bool active = t >= ray.mint && t <= ray.maxt && local.x * local.x + local.y * local.y <= 1.f;
    uv->x = local.x; uv->y = local.y;
    return active ? t : pm_inf();
}
// mesh.h:195-226 (Moeller-Trumbore)
static inline float triangle_intersect(const Shape &s, int index, const Ray &ray, P2 *uv) {
    const uint32_t *fi = &s.faces[3 * index];
    const float *P = s.positions.data();
    V3 p0 = v3(P[3 * fi[0]], P[3 * fi[0] + 1], P[3 * fi[0] + 2]), p1 = v3(P[3 * fi[1]], P[3 * fi[1] + 1], P[3 * fi[1] + 2]),
       p2 = v3(P[3 * fi[2]], P[3 * fi[2] + 1], P[3 * fi[2] + 2]);
    V3 e1 = p1 - p0, e2 = p2 - p0;
    V3 pvec = cross(ray.d, e2);
    float inv_det = pm_rcp(dot(e1, pvec));
    V3 tvec = ray.o - p0;
    float u = dot(tvec, pvec) * inv_det;
    bool active = u >= 0.f && u <= 1.f;
    V3 qvec = cross(tvec, e1);
    float v = dot(ray.d, qvec) * inv_det;
    active = active && v >= 0.f && u + v <= 1.f;
    float t = dot(e2, qvec) * inv_det;
    active = active && t >= ray.mint && t <= ray.maxt;
    uv->x = u; uv->y = v;
    return active ? t : pm_inf();
}
// math.h:371-411 in double precision
static inline bool solve_quadratic_d(double a, double b, double c, double *x0, double *x1) {
    bool linear_case = a == 0.0, valid_linear = linear_case && b != 0.0;
    *x0 = *x1 = -c / b;
    double discrim = std::fma(b, b, -(4.0 * a * c));
    bool valid_quadratic = !linear_case && discrim >= 0.0;
    if (valid_quadratic) {
        double sqrt_discrim = std::sqrt(discrim);
        double temp = -0.5 * (b + std::copysign(sqrt_discrim, b));
        double x0p = temp / a, x1p = c / temp;
        double x0m = std::min(x0p, x1p), x1m = std::max(x0p, x1p);
        *x0 = x0m; *x1 = x1m;
    }
    return valid_linear || valid_quadratic;
}
// sphere.cpp:272-306 (double precision on the CPU, sphere.cpp:276)
static inline float sphere_intersect(const Shape &s, const Ray &ray) {
    double mint = ray.mint, maxt = ray.maxt;
    double ox = (double) ray.o.x - (double) s.center.x, oy = (double) ray.o.y - (double) s.center.y, oz = (double) ray.o.z - (double) s.center.z;
    double dx = ray.d.x, dy = ray.d.y, dz = ray.d.z;
    double A = std::fma(dz, dz, std::fma(dy, dy, dx * dx));
    double B = 2.0 * std::fma(oz, dz, std::fma(oy, dy, ox * dx));
    double C = std::fma(oz, oz, std::fma(oy, oy, ox * ox)) - (double) s.radius * (double) s.radius;
    double near_t, far_t;
    bool found = solve_quadratic_d(A, B, C, &near_t, &far_t);
    bool out_bounds = !(near_t <= maxt && far_t >= mint);
    bool in_bounds = near_t < mint && far_t > maxt;
    bool active = found && !out_bounds && !in_bounds;
    return active ? (near_t < mint ? (float) far_t : (float) near_t) : pm_inf();
}

// ---------------------------------------------------------------- scene traversal
// Closest hit with the semantics of ShapeKDTree::ray_intersect_scalar (kdtree.h:2078-2171):
// the ray is first clipped against the scene bounding box (:2095-2098), every primitive whose
// t lies in [ray.mint, ray.maxt] is a candidate, ray.maxt shrinks to the accepted t (:2152-2154),
// and a later primitive at exactly the same t replaces the earlier one (the test is `t <= maxt`).
// The kd-tree's visiting order is replaced by shape / primitive declaration order.
static inline PreliminaryIntersection ray_intersect_preliminary(const Scene &sc, Ray ray, bool shadow_ray) {
    PreliminaryIntersection pi; pi.t = pm_inf(); pi.prim_uv.x = pi.prim_uv.y = 0.f; pi.prim_index = 0; pi.shape = -1;
    float bmint, bmaxt;
    bbox_ray_intersect(sc.bbox, ray, &bmint, &bmaxt);
    float mint = std::max(ray.mint, bmint), maxt = std::min(ray.maxt, bmaxt);
    if (!(mint <= maxt)) return pi;
    for (size_t i = 0; i < sc.prims.size(); ++i) {
        const Shape &s = sc.shapes[sc.prims[i].shape];
        P2 uv = { 0.f, 0.f }; float t;
        if (s.type == MTS_SHAPE_RECTANGLE) t = rectangle_intersect(s, ray, &uv);
        else if (s.type == MTS_SHAPE_DISK) t = disk_intersect(s, ray, &uv);
        else if (s.type == MTS_SHAPE_SPHERE) t = sphere_intersect(s, ray);
        else t = triangle_intersect(s, sc.prims[i].index, ray, &uv);
        if (t != pm_inf()) {
            pi.t = t; pi.prim_uv = uv; pi.prim_index = sc.prims[i].index; pi.shape = sc.prims[i].shape;
            if (shadow_ray) return pi;
            ray.maxt = t;
        }
    }
    return pi;
}

// rectangle.cpp:167-199
static inline void rectangle_fill(const Shape &s, const Ray &ray, const PreliminaryIntersection &pi, SurfaceInteraction &si) {
    V3 p = ray(pi.t);
    float dist = dot(xf_translation(s.to_world) - p, s.frame.n);
    si.p = fmadd(s.frame.n, dist, p);
    si.n = s.frame.n; si.sh_frame.n = s.frame.n;
    si.dp_du = s.frame.s; si.dp_dv = s.frame.t;
    si.uv.x = pm_fma(pi.prim_uv.x, .5f, .5f); si.uv.y = pm_fma(pi.prim_uv.y, .5f, .5f);
}
// disk.cpp:168-211 (si.uv = (r, phi / 2 pi) is not produced: nothing on this path reads texture coordinates of a disk)
static inline void disk_fill(const Shape &s, const Ray &ray, const PreliminaryIntersection &pi, SurfaceInteraction &si) {
    V3 p = ray(pi.t);
    float dist = dot(xf_translation(s.to_world) - p, s.frame.n);
    si.p = fmadd(s.frame.n, dist, p);
    float r = pm_sqrt(pm_fma(pi.prim_uv.y, pi.prim_uv.y, pi.prim_uv.x * pi.prim_uv.x)), inv_r = pm_rcp(r);
    float cos_phi = r != 0.f ? pi.prim_uv.x * inv_r : 1.f, sin_phi = r != 0.f ? pi.prim_uv.y * inv_r : 0.f;
    si.dp_du = xf_vector(s.to_world, v3(cos_phi, sin_phi, 0.f));
    si.dp_dv = xf_vector(s.to_world, v3(-sin_phi, cos_phi, 0.f));
    si.uv.x = r; si.uv.y = 0.f;
    si.n = s.frame.n; si.sh_frame.n = s.frame.n;
}
// mesh.cpp:448-545
static inline void mesh_fill(const Shape &s, const PreliminaryIntersection &pi, SurfaceInteraction &si) {
    float b1 = pi.prim_uv.x, b2 = pi.prim_uv.y, b0 = 1.f - b1 - b2;
    const uint32_t *fi = &s.faces[3 * pi.prim_index];
    const float *P = s.positions.data();
    V3 p0 = v3(P[3 * fi[0]], P[3 * fi[0] + 1], P[3 * fi[0] + 2]), p1 = v3(P[3 * fi[1]], P[3 * fi[1] + 1], P[3 * fi[1] + 2]),
       p2 = v3(P[3 * fi[2]], P[3 * fi[2] + 1], P[3 * fi[2] + 2]);
    V3 dp0 = p1 - p0, dp1 = p2 - p0;
    si.p = p0 * b0 + p1 * b1 + p2 * b2;
    si.n = normalize(cross(dp0, dp1));
    si.uv.x = b1; si.uv.y = b2;
    coordinate_system(si.n, &si.dp_du, &si.dp_dv);
    if (!s.texcoords.empty()) {
        const float *T = s.texcoords.data();
        P2 uv0 = { T[2 * fi[0]], T[2 * fi[0] + 1] }, uv1 = { T[2 * fi[1]], T[2 * fi[1] + 1] }, uv2 = { T[2 * fi[2]], T[2 * fi[2] + 1] };
        si.uv.x = uv0.x * b0 + uv1.x * b1 + uv2.x * b2;
        si.uv.y = uv0.y * b0 + uv1.y * b1 + uv2.y * b2;
        P2 duv0 = { uv1.x - uv0.x, uv1.y - uv0.y }, duv1 = { uv2.x - uv0.x, uv2.y - uv0.y };
        float det = pm_fma(duv0.x, duv1.y, -(duv0.y * duv1.x)), inv_det = pm_rcp(det);
        if (det != 0.f) {
            // fmsub(duv1.y, dp0, duv0.y * dp1) * inv_det ; fnmadd(duv1.x, dp0, duv0.x * dp1) * inv_det
            si.dp_du = v3(pm_fma(duv1.y, dp0.x, -(duv0.y * dp1.x)), pm_fma(duv1.y, dp0.y, -(duv0.y * dp1.y)), pm_fma(duv1.y, dp0.z, -(duv0.y * dp1.z))) * inv_det;
            si.dp_dv = v3(pm_fma(-duv1.x, dp0.x, duv0.x * dp1.x), pm_fma(-duv1.x, dp0.y, duv0.x * dp1.y), pm_fma(-duv1.x, dp0.z, duv0.x * dp1.z)) * inv_det;
        }
    }
    if (!s.normals.empty()) {
        const float *N = s.normals.data();
        V3 n0 = v3(N[3 * fi[0]], N[3 * fi[0] + 1], N[3 * fi[0] + 2]), n1 = v3(N[3 * fi[1]], N[3 * fi[1] + 1], N[3 * fi[1] + 2]),
           n2 = v3(N[3 * fi[2]], N[3 * fi[2] + 1], N[3 * fi[2] + 2]);
        si.sh_frame.n = normalize(n0 * b0 + n1 * b1 + n2 * b2);
    } else si.sh_frame.n = si.n;
}
// Shape::ray_intersect (shape.cpp:344-352) of a stand-alone analytic shape, reduced to what the distant sensors read: the hit
// point (rectangle.cpp:181-185, disk.cpp:186-188, sphere.cpp:325-327)
static inline bool shape_hit_point(const Shape &s, const Ray &ray, V3 *p) {
    P2 uv; float t;
    if (s.type == MTS_SHAPE_RECTANGLE) t = rectangle_intersect(s, ray, &uv);
    else if (s.type == MTS_SHAPE_DISK) t = disk_intersect(s, ray, &uv);
    else t = sphere_intersect(s, ray);
    if (t == pm_inf()) return false;
    if (s.type == MTS_SHAPE_SPHERE) { V3 n = normalize(ray(t) - s.center); *p = fmadd(n, s.radius, s.center); }
    else { V3 q = ray(t); float dist = dot(xf_translation(s.to_world) - q, s.frame.n); *p = fmadd(s.frame.n, dist, q); }
    return true;
}
// sphere.cpp:308-380 (uv is not needed by any supported BSDF / emitter and is left at zero)
static inline void sphere_fill(const Shape &s, const Ray &ray, const PreliminaryIntersection &pi, SurfaceInteraction &si) {
    si.sh_frame.n = normalize(ray(pi.t) - s.center);
    si.p = fmadd(si.sh_frame.n, s.radius, s.center);
    V3 local = xf_point_affine(s.to_object, si.p);
    float rd_2 = local.x * local.x + local.y * local.y;
    si.uv.x = si.uv.y = 0.f;
    si.dp_du = v3(-local.y, local.x, 0.f);
    float rd = pm_sqrt(rd_2), inv_rd = pm_rcp(rd), cos_phi = local.x * inv_rd, sin_phi = local.y * inv_rd;
    si.dp_dv = v3(local.z * cos_phi, local.z * sin_phi, -rd);
    if (rd == 0.f) si.dp_dv = v3(1.f, 0.f, 0.f);
    si.dp_du = xf_vector(s.to_world, si.dp_du) * (2.f * Pi);
    si.dp_dv = xf_vector(s.to_world, si.dp_dv) * Pi;
    if (s.flip_normals) si.sh_frame.n = -si.sh_frame.n;
    si.n = si.sh_frame.n;
}

// scene_native.inl:23-41 + interaction.h:571-596 (HitComputeFlags::All)
static inline SurfaceInteraction ray_intersect(const Scene &sc, const Ray &ray) {
    PreliminaryIntersection pi = ray_intersect_preliminary(sc, ray, false);
    SurfaceInteraction si; memset(&si, 0, sizeof(si));
    si.shape = -1;
    if (pi.t == pm_inf()) { si.wi = -ray.d; si.t = pm_inf(); return si; }
    const Shape &s = sc.shapes[pi.shape];
    si.t = pi.t;
    if (s.type == MTS_SHAPE_RECTANGLE) rectangle_fill(s, ray, pi, si);
    else if (s.type == MTS_SHAPE_DISK) disk_fill(s, ray, pi, si);
    else if (s.type == MTS_SHAPE_SPHERE) sphere_fill(s, ray, pi, si);
    else mesh_fill(s, pi, si);
    si.prim_index = pi.prim_index; si.shape = pi.shape;
    // initialize_sh_frame, interaction.h:153-156
    si.sh_frame.s = normalize(fnmadd(si.sh_frame.n, dot(si.sh_frame.n, si.dp_du), si.dp_du));
    si.sh_frame.t = cross(si.sh_frame.n, si.sh_frame.s);
    si.wi = si.to_local(-ray.d);
    return si;
}
static inline bool ray_test(const Scene &sc, const Ray &ray) { return ray_intersect_preliminary(sc, ray, true).t != pm_inf(); }   // scene_native.inl:63-67

// ---------------------------------------------------------------- volumes
// grid3d.cpp:234-250
static inline int wrap_coord(const Volume &v, int value, int res) {
    if (v.wrap == MTS_WRAP_CLAMP) return std::min(std::max(value, 0), res - 1);
    int div = value / res;                  // enoki::divisor<int32_t>: truncating division, fixed up below
    int mod = value - div * res;
    if (mod < 0) mod += res;
    if (v.wrap == MTS_WRAP_MIRROR) mod = (((div & 1) == 0) ^ (value < 0)) ? mod : res - 1 - mod;
    return mod;
}
// grid3d.cpp:220-232,259-360 ; constant3d.cpp
#if MTS_SPEC_N != 3
// gridvolume_spectral.cpp:226-388: trilinear in space (cell-centred values, wrapped indices), linear in the spectral dimension
// (nodes over [lambda_min, lambda_max], clamped indices), zero outside the interval
static inline Spec volume_eval_spectral(const Volume &v, V3 p_world) {
    V3 p = xf_point(v.world_to_local, p_world);                                           // :232
    const float *D = v.data; const int nx = v.nx, ny = v.ny, nz = v.nz, ch = v.channels;
    const float inv_dlambda = 1.0f / (v.lambda_max - v.lambda_min), lambda_scale = (float) (ch - 1);      // :186-190; array / scalar = array * (1 / scalar)
    p = v3(pm_fma(p.x, (float) nx, -.5f), pm_fma(p.y, (float) ny, -.5f), pm_fma(p.z, (float) nz, -.5f));
    int ix = (int) pm_floor(p.x), iy = (int) pm_floor(p.y), iz = (int) pm_floor(p.z);
    V3 w1 = p - v3((float) ix, (float) iy, (float) iz), w0 = v3(1.f - w1.x, 1.f - w1.y, 1.f - w1.z);
    int x0 = wrap_coord(v, ix, nx), x1 = wrap_coord(v, ix + 1, nx), y0 = wrap_coord(v, iy, ny), y1 = wrap_coord(v, iy + 1, ny),
        z0 = wrap_coord(v, iz, nz), z1 = wrap_coord(v, iz + 1, nz);
    const float lam[4] = { tls_wavelengths.x, tls_wavelengths.y, tls_wavelengths.z, tls_wavelengths.w };
    float out[4];
    for (int k = 0; k < 4; ++k) {
        const float wn = (lam[k] - v.lambda_min) * inv_dlambda;                            // :233-234 wavelengths_normalized
        const float ws = wn * lambda_scale;                                                // :302
        const int wi = (int) pm_floor(ws);
        const int c0 = std::min(std::max(wi, 0), ch - 1), c1 = std::min(std::max(wi + 1, 0), ch - 1);      // wrap_wavelengths :262-265
        const float s1 = ws - (float) wi, s0 = 1.f - s1;
        float dd[2];
        for (int j = 0; j < 2; ++j) {
            const int c = j ? c1 : c0;
            #define G(X, Y, Z) D[(size_t) (((Z) * ny + (Y)) * nx + (X)) * ch + c]
            float d000 = G(x0, y0, z0), d100 = G(x1, y0, z0), d010 = G(x0, y1, z0), d110 = G(x1, y1, z0),
                  d001 = G(x0, y0, z1), d101 = G(x1, y0, z1), d011 = G(x0, y1, z1), d111 = G(x1, y1, z1);
            #undef G
            float v00 = pm_fma(w0.x, d000, w1.x * d100), v01 = pm_fma(w0.x, d001, w1.x * d101),
                  v10 = pm_fma(w0.x, d010, w1.x * d110), v11 = pm_fma(w0.x, d011, w1.x * d111);
            float v0 = pm_fma(w0.y, v00, w1.y * v10), v1 = pm_fma(w0.y, v01, w1.y * v11);
            dd[j] = pm_fma(w0.z, v0, w1.z * v1);
        }
        const float r = pm_fma(s0, dd[0], s1 * dd[1]);                                     // :373
        // :380-385: the mask compares the NORMALISED wavelength (the argument the function received) with lambda_min / lambda_max
        out[k] = (wn >= v.lambda_min && wn <= v.lambda_max) ? r : 0.f;
    }
    return spec4(out[0], out[1], out[2], out[3]);
}
#endif
#if MTS_SPEC_N == 3
static inline V3 volume_eval(const Volume &v, V3 p_world) {
    if (v.type == MTS_VOLUME_CONST) return v.value;
#else
static inline V3 volume_eval_rgb(const Volume &v, V3 p_world);
static inline Spec volume_eval(const Volume &v, V3 p_world) {
    if (v.type == MTS_VOLUME_CONST) return color_eval(v.value);                           // constant3d.cpp: m_color->eval(si)
    if (v.spectral_grid) return volume_eval_spectral(v, p_world);
    return spec_s(volume_eval_rgb(v, p_world).x);                                         // single-channel grid
}
static inline V3 volume_eval_rgb(const Volume &v, V3 p_world) {
#endif
    V3 p = xf_point(v.world_to_local, p_world);                                           // grid3d.cpp:227
    const float *D = v.data; const int nx = v.nx, ny = v.ny, nz = v.nz, ch = v.channels;
    if (v.filter == MTS_FILTER_TRILINEAR) {
        p = v3(pm_fma(p.x, (float) nx, -.5f), pm_fma(p.y, (float) ny, -.5f), pm_fma(p.z, (float) nz, -.5f));
        int ix = (int) pm_floor(p.x), iy = (int) pm_floor(p.y), iz = (int) pm_floor(p.z);
        V3 w1 = p - v3((float) ix, (float) iy, (float) iz), w0 = v3(1.f - w1.x, 1.f - w1.y, 1.f - w1.z);
        int x0 = wrap_coord(v, ix, nx), x1 = wrap_coord(v, ix + 1, nx), y0 = wrap_coord(v, iy, ny), y1 = wrap_coord(v, iy + 1, ny),
            z0 = wrap_coord(v, iz, nz), z1 = wrap_coord(v, iz + 1, nz);
        V3 r;
        float *out[3] = { &r.x, &r.y, &r.z };
        for (int c = 0; c < ch; ++c) {
            #define G(X, Y, Z) D[(size_t) (((Z) * ny + (Y)) * nx + (X)) * ch + c]
            float d000 = G(x0, y0, z0), d100 = G(x1, y0, z0), d010 = G(x0, y1, z0), d110 = G(x1, y1, z0),
                  d001 = G(x0, y0, z1), d101 = G(x1, y0, z1), d011 = G(x0, y1, z1), d111 = G(x1, y1, z1);
            #undef G
            float v00 = pm_fma(w0.x, d000, w1.x * d100), v01 = pm_fma(w0.x, d001, w1.x * d101),
                  v10 = pm_fma(w0.x, d010, w1.x * d110), v11 = pm_fma(w0.x, d011, w1.x * d111);
            float v0 = pm_fma(w0.y, v00, w1.y * v10), v1 = pm_fma(w0.y, v01, w1.y * v11);
            *out[c] = pm_fma(w0.z, v0, w1.z * v1);
        }
        if (ch == 1) { r.y = r.x; r.z = r.x; }                                             // grid3d.cpp:180-181
        return r;
    } else {
        p = v3(p.x * (float) nx, p.y * (float) ny, p.z * (float) nz);
        int x = wrap_coord(v, (int) pm_floor(p.x), nx), y = wrap_coord(v, (int) pm_floor(p.y), ny), z = wrap_coord(v, (int) pm_floor(p.z), nz);
        size_t index = (size_t) ((z * ny + y) * nx + x) * ch;
        if (ch == 1) return v3(D[index], D[index], D[index]);
        return v3(D[index], D[index + 1], D[index + 2]);
    }
}
// eval_1: grid3d.cpp:187-202 (1 channel: hmean of a 1-vector; 3 channels: luminance), constant3d.cpp (mean)
static inline float volume_eval_1(const Volume &v, V3 p_world) {
#if MTS_SPEC_N == 3
    V3 r = volume_eval(v, p_world);
    if (v.type == MTS_VOLUME_CONST) return (r.x + r.y + r.z) * (1.f / 3.f);
    if (v.channels == 1) return r.x;
    return r.x * 0.212671f + r.y * 0.715160f + r.z * 0.072169f;
#else
    if (v.type == MTS_VOLUME_CONST) return v.value.s->value;                              // uniform.cpp:64-68 eval_1
    return volume_eval(v, p_world).x;
#endif
}

// ---------------------------------------------------------------- media
// homogeneous.cpp:33-54, heterogeneous.cpp:33-54
static inline Spec medium_combined_extinction(const Scene &sc, const Medium &m, V3 p) {
    if (m.is_homogeneous) return volume_eval(sc.volumes[m.sigma_t], p) * m.scale;
    return spec_s(m.max_density);
}
static inline void medium_scattering_coefficients(const Scene &sc, const Medium &m, V3 p, Spec *sigma_s, Spec *sigma_n, Spec *sigma_t, Counters *cnt) {
    if (m.is_homogeneous) {
        Spec st = volume_eval(sc.volumes[m.sigma_t], p) * m.scale;
        *sigma_t = st; *sigma_s = st * volume_eval(sc.volumes[m.albedo], p); *sigma_n = spec_s(0.f);
    } else {
        Spec st = m.scale * volume_eval(sc.volumes[m.sigma_t], p);
        *sigma_t = st; *sigma_s = st * volume_eval(sc.volumes[m.albedo], p);
        *sigma_n = spec_s(m.max_density) - st;
        if (cnt) cnt->n_lookup++;
    }
}
// medium.cpp:34-75
static inline MediumInteraction medium_sample_interaction(const Scene &sc, int medium, const Ray &ray, float sample, uint32_t channel, Counters *cnt) {
    const Medium &m = sc.media[medium];
    MediumInteraction mi;
    mi.sh_frame = frame_from_normal(ray.d);
    mi.wi = -ray.d;
    bool active = true; float mint = 0.f, maxt = pm_inf();
    if (!m.is_homogeneous) {
        active = bbox_ray_intersect(m.aabb, ray, &mint, &maxt);
        active = active && (pm_isfinite(mint) || pm_isfinite(maxt));
        if (!active) { mint = 0.f; maxt = pm_inf(); }
    }
    mint = pm_max(ray.mint, mint);
    maxt = pm_min(ray.maxt, maxt);
    // get_combined_extinction(mi): mi.p is not initialised yet in the reference; the supported
    // homogeneous media read a constvolume, which ignores the position.
    Spec combined = medium_combined_extinction(sc, m, ray.o);
    float mext = idx(combined, channel);
    float sampled_t = mint + (-pm_log(1.f - sample) / mext);
    bool valid_mi = active && (sampled_t <= maxt);
    mi.t = valid_mi ? sampled_t : pm_inf();
    mi.p = ray(sampled_t);
    mi.medium = medium;
    mi.mint = mint;
    if (valid_mi) medium_scattering_coefficients(sc, m, mi.p, &mi.sigma_s, &mi.sigma_n, &mi.sigma_t, cnt);
    else mi.sigma_s = mi.sigma_n = mi.sigma_t = spec_s(0.f);            // eval_impl: `if (none(active)) return zero` (grid3d.cpp:228-229)
    if (!valid_mi && m.is_homogeneous) medium_scattering_coefficients(sc, m, mi.p, &mi.sigma_s, &mi.sigma_n, &mi.sigma_t, cnt);   // constvolume ignores the mask
    mi.combined_extinction = combined;
    return mi;
}

// ---------------------------------------------------------------- phase functions
static inline float eval_hg(float g, float cos_theta) {                                    // hg.cpp:52-55
    float temp = 1.0f + g * g + 2.0f * g * cos_theta;
    return InvFourPi * (1 - g * g) / (temp * pm_sqrt(temp));
}
static inline float eval_rayleigh(float cos_theta) { return (3.f / 16.f) * InvPi * (1.f + cos_theta * cos_theta); }   // rayleigh.cpp:42-45
// distr_1d.h:378-400
static inline float distr_eval_pdf(const ContinuousDistribution &d, float x) {
    bool active = x >= d.range_x && x <= d.range_y;
    x = (x - d.range_x) * d.inv_interval_size;
    uint32_t index = (uint32_t) std::min(std::max((int64_t) x, (int64_t) 0), (int64_t) d.pdf.size() - 2);
    float y0 = active ? d.pdf[index] : 0.f, y1 = active ? d.pdf[index + 1] : 0.f;
    float w1 = x - (float) index, w0 = 1.f - w1;
    return pm_fma(w0, y0, w1 * y1);
}
// distr_1d.h:438-461 with enoki::binary_search (absent source; standard bisection on [valid.x, valid.y])
static inline float distr_sample(const ContinuousDistribution &d, float value) {
    value *= d.integral;
    uint32_t index = distr_binary_search(d.cdf, d.valid_x, d.valid_y, value);
    float y0 = d.pdf[index], y1 = d.pdf[index + 1], c0 = index > 0 ? d.cdf[index - 1] : 0.f;
    value = (value - c0) * d.inv_interval_size;
    float t_linear = (y0 - pm_safe_sqrt(y0 * y0 + 2.f * value * (y1 - y0))) / (y0 - y1), t_const = value / y0;
    float t = (y0 == y1) ? t_const : t_linear;
    return pm_fma((float) index + t, d.interval_size, d.range_x);
}

static float phase_eval(const Scene &sc, int phase, const MediumInteraction &mi, V3 wo) {
    const Phase &ph = sc.phases[phase];
    switch (ph.type) {
        case MTS_PHASE_ISOTROPIC: return InvFourPi;                                       // isotropic.cpp:43-47
        case MTS_PHASE_HG: return eval_hg(ph.g, dot(wo, mi.wi));                          // hg.cpp:81-84
        case MTS_PHASE_RAYLEIGH: return eval_rayleigh(dot(wo, mi.wi));                    // rayleigh.cpp:69-73
        case MTS_PHASE_TABULATED: return distr_eval_pdf(ph.distr, -dot(wo, mi.wi)) * ph.distr.normalization * InvTwoPi;   // tabphase.cpp:72-78
        case MTS_PHASE_BLEND: {                                                           // blendphase.cpp:113-139
            float weight = std::min(std::max(volume_eval_1(sc.volumes[ph.weight_volume], mi.p), 0.f), 1.f);
            return phase_eval(sc, ph.child[0], mi, wo) * (1 - weight) + phase_eval(sc, ph.child[1], mi, wo) * weight;
        }
    }
    return 0.f;
}
static void phase_sample(const Scene &sc, int phase, const MediumInteraction &mi, float sample1, P2 sample2, V3 *wo, float *pdf) {
    const Phase &ph = sc.phases[phase];
    switch (ph.type) {
        case MTS_PHASE_ISOTROPIC: *wo = square_to_uniform_sphere(sample2); *pdf = InvFourPi; return;   // isotropic.cpp:31-41
        case MTS_PHASE_HG: {                                                              // hg.cpp:57-79
            float cos_theta;
            if (std::abs(ph.g) < Epsilon) cos_theta = 1 - 2 * sample2.x;
            else { float sqr_term = (1 - ph.g * ph.g) / (1 - ph.g + 2 * ph.g * sample2.x); cos_theta = (1 + ph.g * ph.g - sqr_term * sqr_term) / (2 * ph.g); }
            float sin_theta = pm_safe_sqrt(1.0f - cos_theta * cos_theta);
            float sin_phi, cos_phi; pm_sincos(2 * Pi * sample2.y, &sin_phi, &cos_phi);
            *wo = mi.sh_frame.to_world(v3(sin_theta * cos_phi, sin_theta * sin_phi, cos_theta));
            *pdf = eval_hg(ph.g, -cos_theta);
            return;
        }
        case MTS_PHASE_RAYLEIGH: {                                                        // rayleigh.cpp:47-67
            float z = 2.f * (2.f * sample2.x - 1.f), tmp = pm_sqrt(z * z + 1.f);
            float A = pm_cbrt(z + tmp), B = pm_cbrt(z - tmp), cos_theta = A + B;
            float sin_theta = pm_safe_sqrt(1.0f - cos_theta * cos_theta);
            float sin_phi, cos_phi; pm_sincos(TwoPi * sample2.y, &sin_phi, &cos_phi);
            *wo = mi.sh_frame.to_world(v3(sin_theta * cos_phi, sin_theta * sin_phi, cos_theta));
            *pdf = eval_rayleigh(-cos_theta);
            return;
        }
        case MTS_PHASE_TABULATED: {                                                       // tabphase.cpp:53-70
            float cos_theta = distr_sample(ph.distr, sample2.x);
            float sin_theta = pm_safe_sqrt(1.0f - cos_theta * cos_theta);
            float sin_phi, cos_phi; pm_sincos(2.f * Pi * sample2.y, &sin_phi, &cos_phi);
            *wo = mi.sh_frame.to_world(v3(sin_theta * cos_phi, sin_theta * sin_phi, cos_theta));
            *pdf = distr_eval_pdf(ph.distr, -cos_theta) * ph.distr.normalization * InvTwoPi;
            return;
        }
        case MTS_PHASE_BLEND: {                                                           // blendphase.cpp:68-111
            float weight = std::min(std::max(volume_eval_1(sc.volumes[ph.weight_volume], mi.p), 0.f), 1.f);
            if (sample1 > weight) phase_sample(sc, ph.child[0], mi, (sample1 - weight) / (1 - weight), sample2, wo, pdf);
            else phase_sample(sc, ph.child[1], mi, sample1 / weight, sample2, wo, pdf);
            return;
        }
    }
}

// PhaseFunctionContext::component (include/mitsuba/render/phase.h): blendphase.cpp:75-87 (sample), :119-134 (eval) address one
// component of the tree -- a blendphase lists the components of its first child, then those of its second (:58-61).  The
// integrators never select a component (ctx.component = -1); these two serve the reference's unit tests (test_blendphase.py:111-200).
static int phase_component_count(const Scene &sc, int phase) {
    const Phase &ph = sc.phases[phase];
    return ph.type == MTS_PHASE_BLEND ? phase_component_count(sc, ph.child[0]) + phase_component_count(sc, ph.child[1]) : 1;
}
static float phase_eval_component(const Scene &sc, int phase, const MediumInteraction &mi, V3 wo, int component) {
    const Phase &ph = sc.phases[phase];
    if (component < 0 || ph.type != MTS_PHASE_BLEND) return phase_eval(sc, phase, mi, wo);
    float weight = std::min(std::max(volume_eval_1(sc.volumes[ph.weight_volume], mi.p), 0.f), 1.f);
    const int n0 = phase_component_count(sc, ph.child[0]);
    const bool sample_first = component < n0;
    if (!sample_first) component -= n0; else weight = 1.f - weight;
    return weight * phase_eval_component(sc, ph.child[sample_first ? 0 : 1], mi, wo, component);
}
static void phase_sample_component(const Scene &sc, int phase, const MediumInteraction &mi, float sample1, P2 sample2, int component, V3 *wo, float *pdf) {
    const Phase &ph = sc.phases[phase];
    if (component < 0 || ph.type != MTS_PHASE_BLEND) { phase_sample(sc, phase, mi, sample1, sample2, wo, pdf); return; }
    float weight = std::min(std::max(volume_eval_1(sc.volumes[ph.weight_volume], mi.p), 0.f), 1.f);
    const int n0 = phase_component_count(sc, ph.child[0]);
    const bool sample_first = component < n0;
    if (!sample_first) component -= n0; else weight = 1.f - weight;
    phase_sample_component(sc, ph.child[sample_first ? 0 : 1], mi, sample1, sample2, component, wo, pdf);
    *pdf *= weight;
}

// ---------------------------------------------------------------- BSDFs
struct BSDFSample { V3 wo; float pdf, eta; uint32_t sampled_type; };
// frame.h:67-70,107-118
static inline float frame_tan_theta(V3 v) { float temp = pm_fma(-v.z, v.z, 1.f); return pm_safe_sqrt(temp) / v.z; }
static inline float frame_sin_theta(V3 v) { return pm_safe_sqrt(pm_fma(v.x, v.x, v.y * v.y)); }
static inline void frame_sincos_phi(V3 v, float *s, float *c) {
    float sin_theta_2 = pm_fma(v.x, v.x, v.y * v.y), inv_sin_theta = pm_rsqrt(sin_theta_2);
    float rx = v.x * inv_sin_theta, ry = v.y * inv_sin_theta;
    if (pm_abs(sin_theta_2) <= 4.f * Epsilon) { rx = 1.f; ry = 0.f; }
    else { rx = std::min(std::max(rx, -1.f), 1.f); ry = std::min(std::max(ry, -1.f), 1.f); }
    *s = ry; *c = rx;
}
// rpv.cpp:85-131
static inline Spec eval_rpv(const Bsdf &b, V3 wi, V3 wo) {
    float sin_phi1, cos_phi1, sin_phi2, cos_phi2;
    frame_sincos_phi(wi, &sin_phi1, &cos_phi1); frame_sincos_phi(wo, &sin_phi2, &cos_phi2);
    float cos_phi1_minus_phi2 = cos_phi1 * cos_phi2 + sin_phi1 * sin_phi2;
    float sin_theta1 = frame_sin_theta(wi), cos_theta1 = wi.z, tan_theta1 = frame_tan_theta(wi);
    float sin_theta2 = frame_sin_theta(wo), cos_theta2 = wo.z, tan_theta2 = frame_tan_theta(wo);
    float G = pm_safe_sqrt(tan_theta1 * tan_theta1 + tan_theta2 * tan_theta2 - 2.f * tan_theta1 * tan_theta2 * cos_phi1_minus_phi2);
    float cos_g = cos_theta1 * cos_theta2 + sin_theta1 * sin_theta2 * cos_phi1_minus_phi2;
    const Spec r0 = color_eval(b.rho_0), rc = color_eval(b.rho_c), gg = color_eval(b.g), kk = color_eval(b.k);
#if MTS_SPEC_N == 3
    float rho_0[3] = { r0.x, r0.y, r0.z }, rho_c[3] = { rc.x, rc.y, rc.z }, g[3] = { gg.x, gg.y, gg.z }, k[3] = { kk.x, kk.y, kk.z }, out[3];
#else
    float rho_0[4] = { r0.x, r0.y, r0.z, r0.w }, rho_c[4] = { rc.x, rc.y, rc.z, rc.w }, g[4] = { gg.x, gg.y, gg.z, gg.w }, k[4] = { kk.x, kk.y, kk.z, kk.w }, out[4];
#endif
    for (int c = 0; c < MTS_SPEC_N; ++c) {
        float F = (1.f - g[c] * g[c]) / pm_pow((1.f + g[c] * g[c] + 2.f * g[c] * cos_g), 1.5f);
        out[c] = rho_0[c] * (pm_pow(cos_theta1 * cos_theta2 * (cos_theta1 + cos_theta2), k[c] - 1.f) * F * (1.f + (1.f - rho_c[c]) / (1 + G))) * InvPi;
    }
#if MTS_SPEC_N == 3
    return v3(out[0], out[1], out[2]);
#else
    return spec4(out[0], out[1], out[2], out[3]);
#endif
}
// bilambertian.cpp:62-190: reflection and transmission lobes, both Lambertian, on both sides
static inline float bilambertian_reflection_weight(const Bsdf &b) {
    Spec r = color_eval(b.reflectance), t = color_eval(b.transmittance);
    Spec q = r / (r + t);
    return spec_hmean(q);                                                                             // hmean; NaN when r + t == 0: masked by the callers
}
static Spec bsdf_eval(const Bsdf &b, const SurfaceInteraction &si, V3 wo) {
    if (b.type == MTS_BSDF_BILAMBERTIAN) {                                                            // bilambertian.cpp:118-146
        bool same = std::signbit(si.wi.z) == std::signbit(wo.z);
        return (same ? color_eval(b.reflectance) : color_eval(b.transmittance)) * (InvPi * pm_abs(wo.z));
    }
    float cos_theta_i = si.wi.z, cos_theta_o = wo.z;
    bool active = cos_theta_i > 0.f && cos_theta_o > 0.f;
    switch (b.type) {
        case MTS_BSDF_DIFFUSE: return active ? color_eval(b.reflectance) * InvPi * cos_theta_o : spec_s(0.f);     // diffuse.cpp:106-120
        case MTS_BSDF_RPV: return active ? eval_rpv(b, si.wi, wo) * pm_abs(cos_theta_o) : spec_s(0.f);   // rpv.cpp:133-142
        default: return spec_s(0.f);                                                                  // null.cpp:60-63
    }
}
static float bsdf_pdf(const Bsdf &b, const SurfaceInteraction &si, V3 wo) {
    if (b.type == MTS_BSDF_NULL) return 0.f;                                                          // null.cpp:65-68
    if (b.type == MTS_BSDF_BILAMBERTIAN) {                                                            // bilambertian.cpp:148-190
        float result = InvPi * pm_abs(wo.z);
        float rw = bilambertian_reflection_weight(b), tw = 1.f - rw;
        if (rw != rw) rw = 0.f;
        if (tw != tw) tw = 0.f;
        bool same = std::signbit(si.wi.z) == std::signbit(wo.z);
        return result * (same ? rw : tw);
    }
    float cos_theta_i = si.wi.z, cos_theta_o = wo.z;
    float pdf = InvPi * wo.z;                                                                          // warp.h:343-350
    return (cos_theta_i > 0.f && cos_theta_o > 0.f) ? pdf : 0.f;                                      // diffuse.cpp:122-135, rpv.cpp:144-153
}
static Spec bsdf_sample(const Bsdf &b, const SurfaceInteraction &si, float sample1, P2 sample2, BSDFSample *bs) {
    bs->wo = v3(0, 0, 0); bs->pdf = 0.f; bs->eta = 0.f; bs->sampled_type = 0;
    if (b.type == MTS_BSDF_BILAMBERTIAN) {                                                            // bilambertian.cpp:62-116
        V3 wo = square_to_cosine_hemisphere(sample2);
        float rw = bilambertian_reflection_weight(b), tw = 1.f - rw;
        if (rw != rw) rw = 0.f;
        if (tw != tw) tw = 0.f;
        bool selected_r = sample1 < rw;
        Spec value = selected_r ? spec_s(1.f) * (color_eval(b.reflectance) / rw) : spec_s(1.f) * (color_eval(b.transmittance) / tw);
        bs->pdf = InvPi * wo.z;
        bs->pdf = selected_r ? bs->pdf * rw : bs->pdf * tw;
        bs->eta = 1.f;
        bs->sampled_type = selected_r ? F_DiffuseReflection : F_DiffuseTransmission;
        if (!(si.wi.z > 0.f)) wo.z = -wo.z;
        bs->wo = selected_r ? wo : v3(wo.x, wo.y, -wo.z);
        return bs->pdf > 0.f ? value : spec_s(0.f);
    }
    if (b.type == MTS_BSDF_NULL) {                                                                    // null.cpp:41-58
        bs->wo = -si.wi; bs->sampled_type = F_Null; bs->eta = 1.f; bs->pdf = 1.f;
        return spec_s(1.f);
    }
    float cos_theta_i = si.wi.z;
    bool active = cos_theta_i > 0.f;
    if (b.type == MTS_BSDF_DIFFUSE) {                                                                 // diffuse.cpp:78-104
        if (!active) return spec_s(0.f);
        bs->wo = square_to_cosine_hemisphere(sample2);
        bs->pdf = InvPi * bs->wo.z; bs->eta = 1.f; bs->sampled_type = F_DiffuseReflection;
        return (bs->pdf > 0.f) ? color_eval(b.reflectance) : spec_s(0.f);
    }
    // rpv.cpp:85-102 (fields are filled even when the lane is inactive)
    bs->wo = square_to_cosine_hemisphere(sample2);
    bs->pdf = InvPi * bs->wo.z; bs->eta = 1.f; bs->sampled_type = F_GlossyReflection;
    Spec value = eval_rpv(b, si.wi, bs->wo);
    return (active && bs->pdf > 0.f) ? value : spec_s(0.f);
}
static inline Spec bsdf_eval_null_transmission(const Bsdf &b) { return b.type == MTS_BSDF_NULL ? spec_s(1.f) : spec_s(0.f); }   // null.cpp:70-73, bsdf.cpp:11-14

// ---------------------------------------------------------------- emitters
struct DirectionSample { V3 p, n, d; float pdf, dist; bool delta; int emitter; };

// rectangle.cpp:111-124 / sphere.cpp sample_position
static inline void shape_sample_position(const Shape &s, P2 sample, V3 *p, V3 *n, float *pdf) {
    if (s.type == MTS_SHAPE_RECTANGLE) {
        *p = xf_point_affine(s.to_world, v3(sample.x * 2.f - 1.f, sample.y * 2.f - 1.f, 0.f));
        *n = s.frame.n; *pdf = s.inv_surface_area;
    } else if (s.type == MTS_SHAPE_DISK) {                                                // disk.cpp:114-128
        P2 q = square_to_uniform_disk_concentric(sample);
        *p = xf_point_affine(s.to_world, v3(q.x, q.y, 0.f));
        *n = s.frame.n; *pdf = s.inv_surface_area;
    } else if (s.type == MTS_SHAPE_CUBE || s.type == MTS_SHAPE_MESH) {                    // mesh.cpp:352-397
        // DiscreteDistribution::sample_reuse (distr_1d.h:141-151,187-197): first face whose running area reaches sample.y * sum
        float reuse, pmf;
        const int lo = (int) discrete_sample_reuse(s.area_distr, sample.y, &reuse, &pmf);
        sample.y = reuse;
        const float *P = s.positions.data(); const uint32_t *f = &s.faces[3 * lo];
        V3 p0 = v3(P[3 * f[0]], P[3 * f[0] + 1], P[3 * f[0] + 2]), p1 = v3(P[3 * f[1]], P[3 * f[1] + 1], P[3 * f[1] + 2]),
           p2 = v3(P[3 * f[2]], P[3 * f[2] + 1], P[3 * f[2] + 2]);
        V3 e0 = p1 - p0, e1 = p2 - p0;
        float t = pm_safe_sqrt(1.f - sample.x), bx = 1.f - t, by = t * sample.y;           // warp.h:153-156
        *p = p0 + e0 * bx + e1 * by;
        if (!s.normals.empty()) {
            const float *N = s.normals.data();
            V3 n0 = v3(N[3 * f[0]], N[3 * f[0] + 1], N[3 * f[0] + 2]), n1 = v3(N[3 * f[1]], N[3 * f[1] + 1], N[3 * f[1] + 2]),
               n2 = v3(N[3 * f[2]], N[3 * f[2] + 1], N[3 * f[2] + 2]);
            *n = normalize(n0 * (1.f - bx - by) + n1 * bx + n2 * by);
        } else *n = normalize(cross(e0, e1));
        *pdf = s.inv_surface_area;
    } else {
        V3 local = square_to_uniform_sphere(sample);
        *p = fmadd(local, s.radius, s.center);
        *n = s.flip_normals ? -local : local; *pdf = s.inv_surface_area;
    }
}
// shape.cpp:293-310 (generic), sphere.cpp sample_direction
static inline DirectionSample shape_sample_direction(const Shape &s, V3 ref_p, P2 sample) {
    DirectionSample ds; memset(&ds, 0, sizeof(ds));
    if (s.type != MTS_SHAPE_SPHERE) {
        shape_sample_position(s, sample, &ds.p, &ds.n, &ds.pdf);
        ds.d = ds.p - ref_p;
        float dist_squared = squared_norm(ds.d);
        ds.dist = pm_sqrt(dist_squared);
        ds.d = ds.d / ds.dist;
        float dp = pm_abs(dot(ds.d, ds.n));
        ds.pdf *= (dp != 0.f) ? dist_squared / dp : 0.f;
        ds.delta = false;
        return ds;
    }
    V3 dc_v = s.center - ref_p;
    float dc_2 = squared_norm(dc_v);
    float radius_adj = s.radius * (s.flip_normals ? (1.f + RayEpsilon) : (1.f - RayEpsilon));
    if (dc_2 > radius_adj * radius_adj) {
        float inv_dc = pm_rsqrt(dc_2), sin_theta_max = s.radius * inv_dc, sin_theta_max_2 = sin_theta_max * sin_theta_max,
              inv_sin_theta_max = pm_rcp(sin_theta_max), cos_theta_max = pm_safe_sqrt(1.f - sin_theta_max_2);
        float sin_theta_2 = sin_theta_max_2 > 0.00068523f ? 1.f - (pm_fma(cos_theta_max - 1.f, sample.x, 1.f)) * (pm_fma(cos_theta_max - 1.f, sample.x, 1.f))
                                                         : sin_theta_max_2 * sample.x;
        float cos_theta = pm_safe_sqrt(1.f - sin_theta_2);
        float cos_alpha = sin_theta_2 * inv_sin_theta_max + cos_theta * pm_safe_sqrt(pm_fma(-sin_theta_2, inv_sin_theta_max * inv_sin_theta_max, 1.f)),
              sin_alpha = pm_safe_sqrt(pm_fma(-cos_alpha, cos_alpha, 1.f));
        float sin_phi, cos_phi; pm_sincos(sample.y * (2.f * Pi), &sin_phi, &cos_phi);
        V3 d = frame_from_normal(dc_v * -inv_dc).to_world(v3(cos_phi * sin_alpha, sin_phi * sin_alpha, cos_alpha));
        ds.p = fmadd(d, s.radius, s.center); ds.n = d; ds.d = ds.p - ref_p;
        float dist2 = squared_norm(ds.d);
        ds.dist = pm_sqrt(dist2); ds.d = ds.d / ds.dist;
        ds.pdf = InvTwoPi / (1.f - cos_theta_max);                                          // warp::square_to_uniform_cone_pdf
        if (ds.dist == 0.f) ds.pdf = 0.f;
    } else {
        V3 d = square_to_uniform_sphere(sample);
        ds.p = fmadd(d, s.radius, s.center); ds.n = d; ds.d = ds.p - ref_p;
        float dist2 = squared_norm(ds.d);
        ds.dist = pm_sqrt(dist2); ds.d = ds.d / ds.dist;
        ds.pdf = s.inv_surface_area * dist2 / pm_abs(dot(ds.d, ds.n));
    }
    ds.delta = s.radius == 0.f;
    if (s.flip_normals) ds.n = -ds.n;
    return ds;
}
static inline float shape_pdf_direction(const Shape &s, V3 ref_p, const DirectionSample &ds) {
    if (s.type != MTS_SHAPE_SPHERE) {                                                       // shape.cpp:312-323 (mesh.cpp:417-419)
        float pdf = s.inv_surface_area, dp = pm_abs(dot(ds.d, ds.n));
        pdf *= (dp != 0.f) ? (ds.dist * ds.dist) / dp : 0.f;
        return pdf;
    }
    float sin_alpha = s.radius * pm_rcp(norm(s.center - ref_p)), cos_alpha = pm_safe_sqrt(1.f - sin_alpha * sin_alpha);   // sphere.cpp pdf_direction
    return sin_alpha < 0x1.fffffep-1f ? InvTwoPi / (1.f - cos_alpha) : s.inv_surface_area * (ds.dist * ds.dist) / pm_abs(dot(ds.d, ds.n));
}

// directional.cpp:109-141, area.cpp:122-165, constant.cpp:81-111
static inline DirectionSample emitter_sample_direction(const Scene &sc, int ei, V3 ref_p, P2 sample, Spec *spec) {
    const Emitter &e = sc.emitters[ei];
    DirectionSample ds; memset(&ds, 0, sizeof(ds));
    if (e.type == MTS_EMITTER_DIRECTIONAL) {
        V3 d = xf_vector(e.to_world, v3(0.f, 0.f, 1.f));
        float dist = 2.f * e.bsphere_radius;
        ds.p = ref_p - d * dist; ds.n = d; ds.pdf = 1.f; ds.delta = true; ds.d = -d; ds.dist = dist;
        *spec = color_eval(e.radiance);
    } else if (e.type == MTS_EMITTER_CONSTANT) {
        V3 d = square_to_uniform_sphere(sample);
        float dist = 2.f * e.bsphere_radius;
        ds.p = ref_p + d * dist; ds.n = -d; ds.pdf = InvFourPi; ds.delta = false; ds.d = d; ds.dist = dist;
        *spec = color_eval(e.radiance) / ds.pdf;
    } else if (e.type == MTS_EMITTER_POINT) {                                                // point.cpp:80-107
        ds.p = xf_translation(e.to_world); ds.n = v3(0, 0, 0); ds.pdf = 1.f; ds.delta = true;
        ds.d = ds.p - ref_p; ds.dist = norm(ds.d);
        float inv_dist = pm_rcp(ds.dist);
        ds.d = ds.d * inv_dist;
        *spec = color_eval(e.radiance) * (inv_dist * inv_dist);
    } else {
        ds = shape_sample_direction(sc.shapes[e.shape], ref_p, sample);
        bool active = dot(ds.d, ds.n) < 0.f && ds.pdf != 0.f;
        *spec = active ? color_eval(e.radiance) / ds.pdf : spec_s(0.f);
    }
    ds.emitter = ei;
    return ds;
}
// scene.cpp:168-218
static inline DirectionSample sample_emitter_direction(const Scene &sc, V3 ref_p, P2 sample, bool test_visibility, Spec *spec) {
    DirectionSample ds; memset(&ds, 0, sizeof(ds)); ds.emitter = -1;
    if (sc.emitters.empty()) { *spec = spec_s(0.f); return ds; }
    bool active = true;
    if (sc.emitters.size() == 1) ds = emitter_sample_direction(sc, 0, ref_p, sample, spec);
    else {
        float n = (float) sc.emitters.size(), emitter_pdf = 1.f / n;
        uint32_t index = std::min((uint32_t) (sample.x * n), (uint32_t) sc.emitters.size() - 1);
        sample.x = (sample.x - index * emitter_pdf) * n;
        ds = emitter_sample_direction(sc, (int) index, ref_p, sample, spec);
        ds.pdf *= emitter_pdf;
        *spec = *spec * pm_rcp(emitter_pdf);
    }
    active = active && ds.pdf != 0.f;
    if (test_visibility && active) {
        Ray ray = make_ray(ref_p, ds.d, RayEpsilon * (1.f + hmax_abs(ref_p)), ds.dist * (1.f - ShadowEpsilon));
        if (ray_test(sc, ray)) *spec = spec_s(0.f);
    }
    return ds;
}
// scene.cpp:220-235
static inline float pdf_emitter_direction(const Scene &sc, V3 ref_p, const DirectionSample &ds) {
    const Emitter &e = sc.emitters[ds.emitter];
    float value;
    if (e.type == MTS_EMITTER_DIRECTIONAL || e.type == MTS_EMITTER_POINT) value = 0.f;     // directional.cpp:143-147, point.cpp:109-112
    else if (e.type == MTS_EMITTER_CONSTANT) value = InvFourPi;                            // constant.cpp:113-117
    else { float dp = dot(ds.d, ds.n); value = dp < 0.f ? shape_pdf_direction(sc.shapes[e.shape], ref_p, ds) : 0.f; }   // area.cpp:168-186
    if (sc.emitters.size() == 1) return value;
    return value * (1.f / sc.emitters.size());
}
// area.cpp:63-71, constant.cpp:41-44, directional.cpp:75-78 ; si.emitter(scene), scene.h:243-253
static inline int si_emitter(const Scene &sc, const SurfaceInteraction &si) { return si.is_valid() ? sc.shapes[si.shape].emitter : sc.environment; }
static inline Spec emitter_eval(const Scene &sc, int ei, const SurfaceInteraction &si) {
    const Emitter &e = sc.emitters[ei];
    if (e.type == MTS_EMITTER_AREA) return si.wi.z > 0.f ? color_eval(e.radiance) : spec_s(0.f);
    if (e.type == MTS_EMITTER_CONSTANT) return color_eval(e.radiance);
    return spec_s(0.f);
}
// interaction.h:178-200
static inline int target_medium(const Scene &sc, const SurfaceInteraction &si, V3 d) {
    const Shape &s = sc.shapes[si.shape];
    return dot(d, si.n) > 0 ? s.exterior : s.interior;
}

// ---------------------------------------------------------------- integrators
static inline float mis_weight(float pdf_a, float pdf_b) { pdf_a *= pdf_a; pdf_b *= pdf_b; return pdf_a > 0.0f ? pdf_a / (pdf_a + pdf_b) : 0.0f; }   // volpath.cpp:479-483

// exp(-t * sigma) per channel (medium.cpp:84)
#if MTS_SPEC_N == 3
static inline Spec transmittance_exp(float t, Spec c) { return v3(pm_exp(-t * c.x), pm_exp(-t * c.y), pm_exp(-t * c.z)); }
#else
static inline Spec transmittance_exp(float t, Spec c) { return spec4(pm_exp(-t * c.x), pm_exp(-t * c.y), pm_exp(-t * c.z), pm_exp(-t * c.w)); }
#endif
// volpath.cpp:261-367
static Spec volpath_sample_emitter(const Scene &sc, V3 ref_p, bool is_medium_interaction, Sampler &sampler, int medium, uint32_t channel, DirectionSample *ds_out, Counters *cnt) {
    Spec transmittance = spec_s(1.f), emitter_val;
    DirectionSample ds = sample_emitter_direction(sc, ref_p, sampler.next_2d(), false, &emitter_val);
    *ds_out = ds;
    if (ds.pdf == 0.f) return spec_s(0.f);
    bool active = true;
    Ray ray = spawn_ray(ref_p, ds.d);
    if (is_medium_interaction) ray.mint = 0.f;
    float total_dist = 0.f;
    SurfaceInteraction si; memset(&si, 0, sizeof(si)); si.t = pm_inf(); si.shape = -1;
    bool needs_intersection = true;
    while (active) {
        float remaining_dist = ds.dist * (1.f - ShadowEpsilon) - total_dist;
        ray.maxt = remaining_dist;
        active = active && remaining_dist > 0.f;
        if (!active) break;
        if (cnt) cnt->n_nee_step++;
        bool escaped_medium = false, active_medium = medium >= 0, active_surface = !active_medium;
        if (active_medium) {
            const Medium &m = sc.media[medium];
            MediumInteraction mi = medium_sample_interaction(sc, medium, ray, sampler.next_1d(), channel, cnt);
            if (m.is_homogeneous && mi.is_valid()) ray.maxt = pm_min(mi.t, remaining_dist);
            if (needs_intersection) si = ray_intersect(sc, ray);
            if (si.t < mi.t) mi.t = pm_inf();
            needs_intersection = false;
            bool is_spectral = m.has_spectral_extinction, not_spectral = !is_spectral;
            if (is_spectral) {
                float t = pm_min(remaining_dist, pm_min(mi.t, si.t)) - mi.mint;
                Spec tr = transmittance_exp(t, mi.combined_extinction);
                Spec free_flight_pdf = (si.t < mi.t || mi.t > remaining_dist) ? tr : tr * mi.combined_extinction;
                float tr_pdf = idx(free_flight_pdf, channel);
                transmittance = transmittance * (tr_pdf > 0.f ? tr / tr_pdf : spec_s(0.f));
            }
            if (mi.t > remaining_dist && mi.is_valid()) total_dist = ds.dist;
            if (mi.t > remaining_dist) mi.t = pm_inf();
            escaped_medium = !mi.is_valid();
            active_medium = mi.is_valid();
            is_spectral = is_spectral && active_medium; not_spectral = not_spectral && active_medium;
            if (active_medium) {
                total_dist += mi.t;
                ray.o = mi.p; ray.mint = 0.f; si.t = si.t - mi.t;
                if (is_spectral) transmittance = transmittance * mi.sigma_n;
                if (not_spectral) transmittance = transmittance * (mi.sigma_n / mi.combined_extinction);
            }
        }
        bool intersect = active_surface && needs_intersection;
        if (intersect) si = ray_intersect(sc, ray);
        needs_intersection = needs_intersection && !intersect;
        active_surface = active_surface || escaped_medium;
        if (active_surface) total_dist += si.t;
        active_surface = active_surface && si.is_valid() && active && !active_medium;
        if (active_surface) transmittance = transmittance * bsdf_eval_null_transmission(sc.bsdf_of(sc.shapes[si.shape]));
        if (active_surface) ray = spawn_ray(si.p, ray.d);
        ray.maxt = remaining_dist;
        needs_intersection = needs_intersection || active_surface;
        active = active && (active_medium || active_surface) && any_nonzero(transmittance);
        if (active_surface && sc.shapes[si.shape].is_medium_transition()) medium = target_medium(sc, si, ray.d);
    }
    return transmittance * emitter_val;
}

// volpath.cpp:370-465
static Spec volpath_evaluate_direct_light(const Scene &sc, V3 ref_p, Sampler &sampler, int medium, Ray ray, const SurfaceInteraction &si_ray,
                                        uint32_t channel, bool active, float *emitter_pdf_out, Counters *cnt) {
    Spec emitter_val = spec_s(0.f), transmittance = spec_s(1.f);
    bool needs_intersection = false;
    float emitter_pdf = 0.f;
    SurfaceInteraction si = si_ray;
    while (active) {
        if (cnt) cnt->n_nee_step++;
        bool escaped_medium = false, active_medium = medium >= 0, active_surface = !active_medium;
        if (active_medium) {
            const Medium &m = sc.media[medium];
            MediumInteraction mi = medium_sample_interaction(sc, medium, ray, sampler.next_1d(), channel, cnt);
            if (m.is_homogeneous && mi.is_valid()) ray.maxt = mi.t;
            if (needs_intersection) si = ray_intersect(sc, ray);
            if (si.t < mi.t) mi.t = pm_inf();
            bool is_spectral = m.has_spectral_extinction, not_spectral = !is_spectral;
            if (is_spectral) {
                float t = pm_min(mi.t, si.t) - mi.mint;                                       // medium.cpp:77-89
                Spec tr = transmittance_exp(t, mi.combined_extinction);
                Spec free_flight_pdf = si.t < mi.t ? tr : tr * mi.combined_extinction;
                float tr_pdf = idx(free_flight_pdf, channel);
                transmittance = transmittance * (tr_pdf > 0.f ? tr / tr_pdf : spec_s(0.f));
            }
            needs_intersection = false;
            escaped_medium = !mi.is_valid();
            active_medium = mi.is_valid();
            if (active_medium) {
                ray.o = mi.p; ray.mint = 0.f; si.t = si.t - mi.t;
                if (is_spectral) transmittance = transmittance * mi.sigma_n;
                if (not_spectral) transmittance = transmittance * (mi.sigma_n / mi.combined_extinction);
            }
        }
        bool intersect = active_surface && needs_intersection;
        if (intersect) si = ray_intersect(sc, ray);
        needs_intersection = needs_intersection && !intersect;
        active_surface = active_surface || escaped_medium;
        int emitter = active_surface ? si_emitter(sc, si) : -1;
        bool emitter_hit = emitter >= 0 && active_surface;
        if (emitter_hit) {
            DirectionSample ds; memset(&ds, 0, sizeof(ds));                                  // records.h:168-174
            ds.p = si.p; ds.n = si.sh_frame.n; ds.d = si.p - ref_p; ds.dist = norm(ds.d); ds.d = ds.d / ds.dist;
            if (!si.is_valid()) ds.d = -si.wi;
            ds.emitter = emitter;
            emitter_val = emitter_eval(sc, emitter, si);
            emitter_pdf = pdf_emitter_direction(sc, ref_p, ds);
            active = false; active_surface = false; active_medium = false;
        }
        active_surface = active_surface && si.is_valid() && !active_medium;
        if (active_surface) transmittance = transmittance * bsdf_eval_null_transmission(sc.bsdf_of(sc.shapes[si.shape]));
        if (active_surface) ray = spawn_ray(si.p, ray.d);
        needs_intersection = needs_intersection || active_surface;
        active = active && (active_medium || active_surface) && any_nonzero(transmittance);
        if (active_surface && sc.shapes[si.shape].is_medium_transition()) medium = target_medium(sc, si, ray.d);
    }
    *emitter_pdf_out = emitter_pdf;
    return transmittance * emitter_val;
}

// volpath.cpp:38-257
static Spec volpath_sample(const Scene &sc, Sampler &sampler, Ray ray, int medium, bool *valid_out, Counters *cnt) {
    const uint32_t max_depth = (uint32_t) sc.integrator.max_depth, rr_depth = (uint32_t) sc.integrator.rr_depth;
    const bool hide_emitters = sc.integrator.hide_emitters != 0;
    bool valid_ray = !hide_emitters && sc.environment >= 0;
    float eta = 1.f;
    Spec throughput = spec_s(1.f), result = spec_s(0.f);
    MediumInteraction mi; memset(&mi, 0, sizeof(mi)); mi.t = pm_inf();
    bool active = true, specular_chain = !hide_emitters;
    uint32_t depth = 0;
#if MTS_SPEC_N == 3
    uint32_t channel = sc.integrator.monochrome ? 0u : (uint32_t) pm_min(sampler.next_1d() * 3.f, 2.f);   // volpath.cpp:63-67 (rgb variants only)
#else
    const uint32_t channel = 0;                                                                            // no draw outside the rgb variants
#endif
    SurfaceInteraction si; memset(&si, 0, sizeof(si)); si.t = pm_inf(); si.shape = -1;
    bool needs_intersection = true;
    for (;;) {
        active = active && any_nonzero(throughput);
        float q = pm_min(hmax(throughput) * (eta * eta), .95f);
        bool perform_rr = depth > rr_depth;
        active = active && (sampler.next_1d() < q || !perform_rr);
        if (perform_rr) throughput = throughput * pm_rcp(q);
        bool exceeded_max_depth = depth >= max_depth;
        if (!active || exceeded_max_depth) break;
        if (cnt) cnt->n_iter++;
        bool active_medium = active && medium >= 0, active_surface = active && !active_medium;
        bool act_null_scatter = false, act_medium_scatter = false, escaped_medium = false;
        bool is_spectral = active_medium, not_spectral = false;
        if (active_medium) { is_spectral = is_spectral && sc.media[medium].has_spectral_extinction; not_spectral = !is_spectral && active_medium; }
        if (active_medium) {
            const Medium &m = sc.media[medium];
            mi = medium_sample_interaction(sc, medium, ray, sampler.next_1d(), channel, cnt);
            if (m.is_homogeneous && mi.is_valid()) ray.maxt = mi.t;
            if (needs_intersection) si = ray_intersect(sc, ray);
            needs_intersection = false;
            if (si.t < mi.t) mi.t = pm_inf();
            if (is_spectral) {
                float t = pm_min(mi.t, si.t) - mi.mint;                                       // medium.cpp:77-89
                Spec tr = transmittance_exp(t, mi.combined_extinction);
                Spec free_flight_pdf = si.t < mi.t ? tr : tr * mi.combined_extinction;
                float tr_pdf = idx(free_flight_pdf, channel);
                throughput = throughput * (tr_pdf > 0.f ? tr / tr_pdf : spec_s(0.f));
            }
            escaped_medium = !mi.is_valid();
            active_medium = mi.is_valid();
            bool null_scatter = sampler.next_1d() >= idx(mi.sigma_t, channel) / idx(mi.combined_extinction, channel);
            act_null_scatter = null_scatter && active_medium;
            act_medium_scatter = !act_null_scatter && active_medium;
            if (is_spectral && act_null_scatter)
                throughput = throughput * (mi.sigma_n * idx(mi.combined_extinction, channel) / idx(mi.sigma_n, channel));
            if (act_medium_scatter) depth += 1;
        }
        active = active && depth < max_depth;
        act_medium_scatter = act_medium_scatter && active;
        if (act_null_scatter) { ray.o = mi.p; ray.mint = 0.f; si.t = si.t - mi.t; }
        if (act_medium_scatter) {
            if (is_spectral) throughput = throughput * (mi.sigma_s * idx(mi.combined_extinction, channel) / idx(mi.sigma_t, channel));
            if (not_spectral) throughput = throughput * (mi.sigma_s / mi.sigma_t);
            const Medium &m = sc.media[mi.medium];
            bool sample_emitters = m.sample_emitters;
            valid_ray = true;
            specular_chain = !sample_emitters;
            if (sample_emitters) {
                DirectionSample ds;
                Spec emitted = volpath_sample_emitter(sc, mi.p, true, sampler, medium, channel, &ds, cnt);
                float phase_val = phase_eval(sc, m.phase, mi, ds.d);
                result = result + throughput * phase_val * emitted;
            }
            float s1 = sampler.next_1d(); P2 s2 = sampler.next_2d();                          // left-to-right, SURVEY.md 8(a')
            V3 wo; float phase_pdf;
            phase_sample(sc, m.phase, mi, s1, s2, &wo, &phase_pdf);
            ray = spawn_ray(mi.p, wo); ray.mint = 0.0f;
            needs_intersection = true;
        }
        active_surface = active_surface || escaped_medium;
        bool intersect = active_surface && needs_intersection;
        if (intersect) si = ray_intersect(sc, ray);
        if (active_surface) {
            int emitter = si_emitter(sc, si);
            if (specular_chain && emitter >= 0) result = result + throughput * emitter_eval(sc, emitter, si);
        }
        active_surface = active_surface && si.is_valid();
        if (active_surface) {
            const Shape &shape = sc.shapes[si.shape];
            const Bsdf &bsdf = sc.bsdf_of(shape);
            bool active_e = (bsdf.flags & F_Smooth) != 0 && (depth + 1 < max_depth);
            if (active_e) {
                DirectionSample ds;
                Spec emitted = volpath_sample_emitter(sc, si.p, false, sampler, medium, channel, &ds, cnt);
                V3 wo = si.to_local(ds.d);
                Spec bsdf_val = bsdf_eval(bsdf, si, wo);
                float bpdf = bsdf_pdf(bsdf, si, wo);
                result = result + throughput * bsdf_val * mis_weight(ds.pdf, ds.delta ? 0.f : bpdf) * emitted;
            }
            float s1 = sampler.next_1d(); P2 s2 = sampler.next_2d();
            BSDFSample bs;
            Spec bsdf_val = bsdf_sample(bsdf, si, s1, s2, &bs);
            throughput = throughput * bsdf_val;
            eta *= bs.eta;
            ray = spawn_ray(si.p, si.to_world(bs.wo));
            needs_intersection = true;
            bool non_null_bsdf = !(bs.sampled_type & F_Null);
            if (non_null_bsdf) depth += 1;
            valid_ray = valid_ray || non_null_bsdf;
            specular_chain = specular_chain || (non_null_bsdf && (bs.sampled_type & F_Delta));
            specular_chain = specular_chain && !(bs.sampled_type & F_Smooth);
            bool add_emitter = !(bs.sampled_type & F_Delta) && any_nonzero(throughput) && (depth < max_depth);
            act_null_scatter = act_null_scatter || (bs.sampled_type & F_Null);
            bool intersect2 = needs_intersection && add_emitter;
            SurfaceInteraction si_new = si;
            if (intersect2) si_new = ray_intersect(sc, ray);
            needs_intersection = needs_intersection && !intersect2;
            float emitter_pdf;
            Spec emitted = volpath_evaluate_direct_light(sc, si.p, sampler, medium, ray, si_new, channel, add_emitter, &emitter_pdf, cnt);
            if (add_emitter && emitter_pdf != 0) result = result + mis_weight(bs.pdf, emitter_pdf) * throughput * emitted;
            if (shape.is_medium_transition()) medium = target_medium(sc, si, ray.d);
            if (intersect2) si = si_new;
        }
        active = active && (active_surface || active_medium);
    }
    *valid_out = valid_ray;
    return result;
}

// ---------------------------------------------------------------- volpathmis
// src/integrators/volpathmis.cpp: the volumetric path tracer with spectral multiple importance sampling.  WeightMatrix (:66-69) is
// an n x n matrix with n = array_size_v<UnpolarizedSpectrum> -- 3 x 3 in the rgb variants, 4 x 4 in the spectral one (one row of
// probability ratios per colour channel / wavelength) -- with `use_spectral_mis` (the default, :29,38-46), a single spectrum without.
// Outside the rgb variants index_spectrum returns spec[0] (:74-84) and `channel` stays 0 without a draw (:118-124).
// hsum of a 3-array is evaluated left to right, of a 4-array pairwise (as spec_hmean, oracle_math.h).
static inline float sget(const Spec &a, int i) { return (&a.x)[i]; }
static inline void sset(Spec &a, int i, float v) { (&a.x)[i] = v; }
static inline float spec_hsum(Spec a) {
#if MTS_SPEC_N == 3
    return (a.x + a.y) + a.z;
#else
    return (a.x + a.y) + (a.z + a.w);
#endif
}
static inline float spec_hmin_abs(Spec a) {
    float m = pm_min(pm_min(pm_abs(a.x), pm_abs(a.y)), pm_abs(a.z));
#if MTS_SPEC_N != 3
    m = pm_min(pm_min(pm_abs(a.x), pm_abs(a.y)), pm_min(pm_abs(a.z), pm_abs(a.w)));
#endif
    return m;
}
template <bool SPEC> struct MisWeights { Spec r[SPEC ? MTS_SPEC_N : 1]; };
template <bool SPEC> static inline MisWeights<SPEC> mw_full(float v) { MisWeights<SPEC> w; for (auto &x : w.r) x = spec_s(v); return w; }
static inline bool finite3(float x) { return pm_isfinite(x); }
// volpathmis.cpp:447-466
template <bool SPEC>
static inline void update_weights(MisWeights<SPEC> &w, Spec p, Spec f, uint32_t channel, bool active) {
    if (SPEC) {
        for (int i = 0; i < MTS_SPEC_N; ++i) {
            float rfi = 1.0f / sget(f, i);                       // :456 spectrum / coefficient: enoki multiplies by the reciprocal (oracle_math.h)
            Spec ratio;
            for (int j = 0; j < MTS_SPEC_N; ++j) { float r = sget(p, j) * rfi; sset(ratio, j, finite3(r) ? r : 0.f); }
            ratio = ratio * w.r[i];
            if (active) for (int j = 0; j < MTS_SPEC_N; ++j) { float r = sget(ratio, j); sset(w.r[i], j, r != r ? 0.f : r); }
        }
    } else {
        float pdf = idx(p, channel);
        Spec ratio;
        for (int j = 0; j < MTS_SPEC_N; ++j) sset(ratio, j, pdf / sget(f, j));
        ratio = w.r[0] * ratio;
        if (active) for (int j = 0; j < MTS_SPEC_N; ++j) { float r = sget(ratio, j); sset(w.r[0], j, finite3(r) ? r : 0.f); }
    }
}
template <bool SPEC> static inline void update_weights(MisWeights<SPEC> &w, float p, Spec f, uint32_t c, bool a) { update_weights(w, spec_s(p), f, c, a); }
template <bool SPEC> static inline void update_weights(MisWeights<SPEC> &w, Spec p, float f, uint32_t c, bool a) { update_weights(w, p, spec_s(f), c, a); }
template <bool SPEC> static inline void update_weights(MisWeights<SPEC> &w, float p, float f, uint32_t c, bool a) { update_weights(w, spec_s(p), spec_s(f), c, a); }
// volpathmis.cpp:468-481
template <bool SPEC>
static inline Spec mis_weight_w(const MisWeights<SPEC> &w) {
    if (SPEC) {
        Spec o;
        for (int i = 0; i < MTS_SPEC_N; ++i) { float sum = spec_hsum(w.r[i]); sset(o, i, sum == 0.f ? 0.f : (float) MTS_SPEC_N / sum); }
        return o;
    }
    Spec a = w.r[0];
    if (spec_hmin_abs(a) == 0.f) return spec_s(0.f);
    Spec o;
    for (int j = 0; j < MTS_SPEC_N; ++j) sset(o, j, 1.f / sget(a, j));
    return o;
}
// volpathmis.cpp:484-498
template <bool SPEC>
static inline Spec mis_weight_w(const MisWeights<SPEC> &a, const MisWeights<SPEC> &b) {
    if (SPEC) {
        Spec o;
        for (int i = 0; i < MTS_SPEC_N; ++i) { float sum = spec_hsum(a.r[i] + b.r[i]); sset(o, i, sum == 0.f ? 0.f : (float) MTS_SPEC_N / sum); }
        return o;
    }
    Spec sum = a.r[0] + b.r[0];
    if (spec_hmin_abs(sum) == 0.f) return spec_s(0.f);
    Spec o;
    for (int j = 0; j < MTS_SPEC_N; ++j) sset(o, j, 1.f / sget(sum, j));
    return o;
}
// volpathmis.cpp:330-445
template <bool SPEC>
static Spec volpathmis_sample_emitter(const Scene &sc, V3 ref_p, bool is_medium_interaction, Sampler &sampler, int medium, const MisWeights<SPEC> &p_over_f,
                                    uint32_t channel, MisWeights<SPEC> *nee_out, MisWeights<SPEC> *uni_out, DirectionSample *ds_out, Counters *cnt) {
    MisWeights<SPEC> p_over_f_nee = p_over_f, p_over_f_uni = p_over_f;
    Spec emitter_sample_weight;
    DirectionSample ds = sample_emitter_direction(sc, ref_p, sampler.next_2d(), false, &emitter_sample_weight);
    Spec emitter_val = emitter_sample_weight * ds.pdf;
    if (ds.pdf == 0.f) emitter_val = spec_s(0.f);
    bool active = ds.pdf != 0.f;
    update_weights(p_over_f_nee, ds.pdf, 1.0f, channel, active);
    *ds_out = ds;
    if (!active) { *nee_out = p_over_f_nee; *uni_out = p_over_f_uni; return emitter_val; }
    Ray ray = spawn_ray(ref_p, ds.d);
    if (is_medium_interaction) ray.mint = 0.f;
    float total_dist = 0.f;
    SurfaceInteraction si; memset(&si, 0, sizeof(si)); si.t = pm_inf(); si.shape = -1;
    bool needs_intersection = true;
    while (active) {
        float remaining_dist = ds.dist * (1.f - ShadowEpsilon) - total_dist;
        ray.maxt = remaining_dist;
        active = active && remaining_dist > 0.f;
        if (!active) break;
        if (cnt) cnt->n_nee_step++;
        bool escaped_medium = false, active_medium = medium >= 0, active_surface = !active_medium;
        if (active_medium) {
            const Medium &m = sc.media[medium];
            MediumInteraction mi = medium_sample_interaction(sc, medium, ray, sampler.next_1d(), channel, cnt);
            if (m.is_homogeneous && mi.is_valid()) ray.maxt = pm_min(mi.t, remaining_dist);
            if (needs_intersection) si = ray_intersect(sc, ray);
            if (si.t < mi.t) mi.t = pm_inf();
            needs_intersection = false;
            bool is_spectral = m.has_spectral_extinction, not_spectral = !is_spectral;
            if (is_spectral) {
                float t = pm_min(remaining_dist, pm_min(mi.t, si.t)) - mi.mint;
                Spec tr = transmittance_exp(t, mi.combined_extinction);
                Spec free_flight_pdf = (si.t < mi.t || mi.t > remaining_dist) ? tr : tr * mi.combined_extinction;
                update_weights(p_over_f_nee, free_flight_pdf, tr, channel, true);
                update_weights(p_over_f_uni, free_flight_pdf, tr, channel, true);
            }
            if (mi.t > remaining_dist && mi.is_valid()) total_dist = ds.dist;
            if (mi.t > remaining_dist) mi.t = pm_inf();
            escaped_medium = !mi.is_valid();
            active_medium = mi.is_valid();
            is_spectral = is_spectral && active_medium; not_spectral = not_spectral && active_medium;
            if (active_medium) {
                total_dist += mi.t;
                ray.o = mi.p; ray.mint = 0.f; si.t = si.t - mi.t;
                if (is_spectral) {
                    update_weights(p_over_f_nee, 1.f, mi.sigma_n, channel, true);
                    update_weights(p_over_f_uni, mi.sigma_n / mi.combined_extinction, mi.sigma_n, channel, true);
                }
                if (not_spectral) {
                    update_weights(p_over_f_nee, 1.f, mi.sigma_n / mi.combined_extinction, channel, true);
                    update_weights(p_over_f_uni, mi.sigma_n, mi.sigma_n, channel, true);
                }
            }
        }
        bool intersect = active_surface && needs_intersection;
        if (intersect) si = ray_intersect(sc, ray);
        active_surface = active_surface || escaped_medium;
        if (active_surface) total_dist += si.t;
        active_surface = active_surface && si.is_valid() && active && !active_medium;
        if (active_surface) {
            Spec bsdf_val = bsdf_eval_null_transmission(sc.bsdf_of(sc.shapes[si.shape]));
            update_weights(p_over_f_nee, 1.0f, bsdf_val, channel, true);
            update_weights(p_over_f_uni, 1.0f, bsdf_val, channel, true);
        }
        if (active_surface) ray = spawn_ray(si.p, ray.d);
        ray.maxt = remaining_dist;
        needs_intersection = needs_intersection || active_surface;
        if (SPEC) active = active && (active_medium || active_surface) && any_nonzero(mis_weight_w(p_over_f_uni));
        else active = active && (active_medium || active_surface) && (any_nonzero(p_over_f_uni.r[0]) || any_nonzero(p_over_f_nee.r[0]));
        if (active_surface && sc.shapes[si.shape].is_medium_transition()) medium = target_medium(sc, si, ray.d);
    }
    *nee_out = p_over_f_nee; *uni_out = p_over_f_uni;
    return emitter_val;
}

// volpathmis.cpp:86-328
template <bool SPEC>
static Spec volpathmis_sample(const Scene &sc, Sampler &sampler, Ray ray, int medium, bool *valid_out, Counters *cnt) {
    const uint32_t max_depth = (uint32_t) sc.integrator.max_depth, rr_depth = (uint32_t) sc.integrator.rr_depth;
    const bool hide_emitters = sc.integrator.hide_emitters != 0;
    bool valid_ray = !hide_emitters && sc.environment >= 0;
    float eta = 1.f;
    Spec result = spec_s(0.f);
    MediumInteraction mi; memset(&mi, 0, sizeof(mi)); mi.t = pm_inf();
    bool active = true, specular_chain = !hide_emitters;
    uint32_t depth = 0;
    MisWeights<SPEC> p_over_f = mw_full<SPEC>(1.f), p_over_f_nee = mw_full<SPEC>(1.f);
#if MTS_SPEC_N == 3
    uint32_t channel = sc.integrator.monochrome ? 0u : (uint32_t) pm_min(sampler.next_1d() * 3.f, 2.f);   // volpathmis.cpp:120-124
#else
    uint32_t channel = 0;                                                                     // :120-124: a draw in the rgb variants only
#endif
    SurfaceInteraction si; memset(&si, 0, sizeof(si)); si.t = pm_inf(); si.shape = -1;
    bool needs_intersection = true, last_event_was_null = false;
    V3 last_scatter_p = v3(0.f, 0.f, 0.f);                                                    // last_scatter_event: only .p is read
    for (;;) {
        Spec mis_throughput = mis_weight_w(p_over_f);
        float q = pm_min(hmax(mis_throughput) * (eta * eta), .95f);
        bool perform_rr = active && !last_event_was_null && (depth > rr_depth);
        active = active && !(sampler.next_1d() >= q && perform_rr);
        update_weights(p_over_f, q, 1.0f, channel, perform_rr);
        last_event_was_null = false;
        bool exceeded_max_depth = depth >= max_depth;
        active = active && !exceeded_max_depth;
        active = active && any_nonzero(mis_weight_w(p_over_f));
        if (!active) break;
        if (cnt) cnt->n_iter++;
        bool active_medium = active && medium >= 0, active_surface = active && !active_medium;
        bool act_null_scatter = false, act_medium_scatter = false, escaped_medium = false;
        bool is_spectral = active_medium, not_spectral = false;
        if (active_medium) { is_spectral = is_spectral && sc.media[medium].has_spectral_extinction; not_spectral = !is_spectral && active_medium; }
        if (active_medium) {
            const Medium &m = sc.media[medium];
            mi = medium_sample_interaction(sc, medium, ray, sampler.next_1d(), channel, cnt);
            if (m.is_homogeneous && mi.is_valid()) ray.maxt = mi.t;
            if (needs_intersection) si = ray_intersect(sc, ray);
            needs_intersection = false;
            if (si.t < mi.t) mi.t = pm_inf();
            if (is_spectral) {
                float t = pm_min(mi.t, si.t) - mi.mint;                                       // medium.cpp:77-89
                Spec tr = transmittance_exp(t, mi.combined_extinction);
                Spec free_flight_pdf = si.t < mi.t ? tr : tr * mi.combined_extinction;
                update_weights(p_over_f, free_flight_pdf, tr, channel, true);
                update_weights(p_over_f_nee, free_flight_pdf, tr, channel, true);
            }
            escaped_medium = !mi.is_valid();
            active_medium = mi.is_valid();
            is_spectral = is_spectral && active_medium; not_spectral = not_spectral && active_medium;
        }
        if (active_medium) {
            bool null_scatter = sampler.next_1d() >= idx(mi.sigma_t, channel) / idx(mi.combined_extinction, channel);
            act_null_scatter = null_scatter;
            act_medium_scatter = !act_null_scatter;
            if (act_medium_scatter) { depth += 1; last_scatter_p = mi.p; }
            const Medium &m = sc.media[mi.medium];
            bool sample_emitters = m.sample_emitters;
            active = active && depth < max_depth;
            act_medium_scatter = act_medium_scatter && active;
            specular_chain = specular_chain && !(act_medium_scatter && sample_emitters);
            if (act_null_scatter) {
                if (is_spectral) {
                    update_weights(p_over_f, mi.sigma_n / mi.combined_extinction, mi.sigma_n, channel, true);
                    update_weights(p_over_f_nee, 1.0f, mi.sigma_n, channel, true);
                }
                if (not_spectral) {
                    update_weights(p_over_f, mi.sigma_n, mi.sigma_n, channel, true);
                    update_weights(p_over_f_nee, 1.0f, mi.sigma_n / mi.combined_extinction, channel, true);
                }
                ray.o = mi.p; ray.mint = 0.f; si.t = si.t - mi.t;
            }
            if (act_medium_scatter) {
                if (is_spectral) update_weights(p_over_f, mi.sigma_t / mi.combined_extinction, mi.sigma_s, channel, true);
                if (not_spectral) update_weights(p_over_f, mi.sigma_t, mi.sigma_s, channel, true);
                valid_ray = true;
                if (sample_emitters) {
                    MisWeights<SPEC> nee_end, uni_end; DirectionSample ds;
                    Spec emitted = volpathmis_sample_emitter<SPEC>(sc, mi.p, true, sampler, medium, p_over_f, channel, &nee_end, &uni_end, &ds, cnt);
                    bool active_e = true;
                    float phase_val = phase_eval(sc, m.phase, mi, ds.d);
                    update_weights(nee_end, 1.0f, phase_val, channel, active_e);
                    update_weights(uni_end, ds.delta ? 0.f : phase_val, phase_val, channel, active_e);
                    result = result + mis_weight_w(nee_end, uni_end) * emitted;
                }
                p_over_f_nee = p_over_f;
                float s1 = sampler.next_1d(); P2 s2 = sampler.next_2d();                      // left-to-right, SURVEY.md 8(a')
                V3 wo; float phase_pdf;
                phase_sample(sc, m.phase, mi, s1, s2, &wo, &phase_pdf);
                ray = spawn_ray(mi.p, wo); ray.mint = 0.0f;
                needs_intersection = true;
                update_weights(p_over_f, phase_pdf, phase_pdf, channel, true);
                update_weights(p_over_f_nee, 1.f, phase_pdf, channel, true);
            }
        }
        active_surface = active_surface || escaped_medium;
        bool intersect = active_surface && needs_intersection;
        if (intersect) si = ray_intersect(sc, ray);
        if (active_surface) {
            bool ray_from_camera = depth == 0;
            bool count_direct = ray_from_camera || specular_chain;
            int emitter = si_emitter(sc, si);
            bool active_e = emitter >= 0 && !(depth == 0 && hide_emitters);
            if (active_e) {
                if (!count_direct) {
                    DirectionSample ds; memset(&ds, 0, sizeof(ds));                          // records.h:168-174
                    ds.p = si.p; ds.n = si.sh_frame.n; ds.d = si.p - last_scatter_p; ds.dist = norm(ds.d); ds.d = ds.d / ds.dist;
                    if (!si.is_valid()) ds.d = -si.wi;
                    ds.emitter = emitter;
                    float emitter_pdf = pdf_emitter_direction(sc, last_scatter_p, ds);
                    update_weights(p_over_f_nee, emitter_pdf, 1.f, channel, true);
                }
                Spec emitted = emitter_eval(sc, emitter, si);
                Spec contrib = count_direct ? mis_weight_w(p_over_f) * emitted : mis_weight_w(p_over_f, p_over_f_nee) * emitted;
                result = result + contrib;
            }
        }
        active_surface = active_surface && si.is_valid();
        if (active_surface) {
            const Shape &shape = sc.shapes[si.shape];
            const Bsdf &bsdf = sc.bsdf_of(shape);
            bool active_e = (bsdf.flags & F_Smooth) != 0 && (depth + 1 < max_depth);
            if (active_e) {
                MisWeights<SPEC> nee_end, uni_end; DirectionSample ds;
                Spec emitted = volpathmis_sample_emitter<SPEC>(sc, si.p, false, sampler, medium, p_over_f, channel, &nee_end, &uni_end, &ds, cnt);
                V3 wo_local = si.to_local(ds.d);
                Spec bsdf_val = bsdf_eval(bsdf, si, wo_local);
                float bpdf = bsdf_pdf(bsdf, si, wo_local);
                update_weights(nee_end, 1.0f, bsdf_val, channel, true);
                update_weights(uni_end, ds.delta ? 0.f : bpdf, bsdf_val, channel, true);
                result = result + mis_weight_w(nee_end, uni_end) * emitted;
            }
            float s1 = sampler.next_1d(); P2 s2 = sampler.next_2d();
            BSDFSample bs;
            Spec bsdf_weight = bsdf_sample(bsdf, si, s1, s2, &bs);
            bool invalid_bsdf_sample = bs.pdf == 0.f;
            active_surface = active_surface && bs.pdf > 0.f;
            if (active_surface) eta *= bs.eta;
            Ray bsdf_ray = spawn_ray(si.p, si.to_world(bs.wo));
            if (active_surface) { ray = bsdf_ray; needs_intersection = true; }
            bool non_null_bsdf = active_surface && !(bs.sampled_type & F_Null);
            valid_ray = valid_ray || non_null_bsdf || invalid_bsdf_sample;
            specular_chain = specular_chain || (non_null_bsdf && (bs.sampled_type & F_Delta));
            specular_chain = specular_chain && !(active_surface && (bs.sampled_type & F_Smooth));
            if (non_null_bsdf) { depth += 1; last_scatter_p = si.p; }
            if (non_null_bsdf) p_over_f_nee = p_over_f;
            update_weights(p_over_f, bs.pdf, bsdf_weight * bs.pdf, channel, active_surface);
            update_weights(p_over_f_nee, 1.f, bsdf_weight * bs.pdf, channel, non_null_bsdf);
            if (active_surface && shape.is_medium_transition()) medium = target_medium(sc, si, ray.d);
        }
        active = active && (active_surface || active_medium);
    }
    *valid_out = valid_ray;
    return result;
}

// path.cpp:100-211
static Spec path_sample(const Scene &sc, Sampler &sampler, Ray ray, bool *valid_out, Counters *cnt) {
    const int max_depth = sc.integrator.max_depth, rr_depth = sc.integrator.rr_depth;
    float eta = 1.f, emission_weight = 1.f;
    Spec throughput = spec_s(1.f), result = spec_s(0.f);
    bool active = true;
    SurfaceInteraction si = ray_intersect(sc, ray);
    bool valid_ray = si.is_valid();
    int emitter = si_emitter(sc, si);
    for (int depth = 1;; ++depth) {
        if (cnt) cnt->n_iter++;
        if (emitter >= 0 && active) result = result + emission_weight * throughput * emitter_eval(sc, emitter, si);
        active = active && si.is_valid();
        if (depth > rr_depth) {
            float q = pm_min(hmax(throughput) * (eta * eta), .95f);
            active = active && sampler.next_1d() < q;
            throughput = throughput * pm_rcp(q);
        }
        if ((uint32_t) depth >= (uint32_t) max_depth || !active) break;
        const Bsdf &bsdf = sc.bsdf_of(sc.shapes[si.shape]);
        bool active_e = active && (bsdf.flags & F_Smooth);
        if (active_e) {
            Spec emitter_val;
            DirectionSample ds = sample_emitter_direction(sc, si.p, sampler.next_2d(), true, &emitter_val);
            active_e = active_e && ds.pdf != 0.f;
            V3 wo = si.to_local(ds.d);
            Spec bsdf_val = bsdf_eval(bsdf, si, wo);
            float bpdf = bsdf_pdf(bsdf, si, wo);
            float mis = ds.delta ? 1.f : mis_weight(ds.pdf, bpdf);
            if (active_e) result = result + mis * throughput * bsdf_val * emitter_val;
        }
        float s1 = sampler.next_1d(); P2 s2 = sampler.next_2d();
        BSDFSample bs;
        Spec bsdf_val = bsdf_sample(bsdf, si, s1, s2, &bs);
        throughput = throughput * bsdf_val;
        active = active && any_nonzero(throughput);
        if (!active) break;
        eta *= bs.eta;
        ray = spawn_ray(si.p, si.to_world(bs.wo));
        SurfaceInteraction si_bsdf = ray_intersect(sc, ray);
        emitter = si_emitter(sc, si_bsdf);
        if (emitter >= 0) {
            DirectionSample ds; memset(&ds, 0, sizeof(ds));                                  // records.h:168-174
            ds.p = si_bsdf.p; ds.n = si_bsdf.sh_frame.n; ds.d = si_bsdf.p - si.p; ds.dist = norm(ds.d); ds.d = ds.d / ds.dist;
            if (!si_bsdf.is_valid()) ds.d = -si_bsdf.wi;
            ds.emitter = emitter;
            float emitter_pdf = !(bs.sampled_type & F_Delta) ? pdf_emitter_direction(sc, si.p, ds) : 0.f;
            emission_weight = mis_weight(bs.pdf, emitter_pdf);
        }
        si = si_bsdf;
    }
    *valid_out = valid_ray;
    return result;
}

static Spec integrator_sample(const Scene &sc, Sampler &sampler, Ray ray, int medium, bool *valid, Counters *cnt) {
    switch (sc.integrator.type) {
        case MTS_INTEGRATOR_VOLPATH: return volpath_sample(sc, sampler, ray, medium, valid, cnt);
        case MTS_INTEGRATOR_VOLPATHMIS: return sc.integrator.use_spectral_mis ? volpathmis_sample<true>(sc, sampler, ray, medium, valid, cnt)
                                                                             : volpathmis_sample<false>(sc, sampler, ray, medium, valid, cnt);
        default: return path_sample(sc, sampler, ray, valid, cnt);
    }
}

// ---------------------------------------------------------------- sensors
// perspective.cpp:210-252 (ray differentials are unused by the supported plugins)
static Ray sensor_sample_ray(const Scene &sc, P2 position_sample, P2 aperture_sample, V3 *weight) {
    const Sensor &se = sc.sensor;
    if (se.type == MTS_SENSOR_PERSPECTIVE) {
        V3 near_p = xf_point(se.sample_to_camera, v3(position_sample.x + se.principal_point_offset.x, position_sample.y + se.principal_point_offset.y, 0.f));
        V3 d = normalize(near_p);
        float inv_z = pm_rcp(d.z);
        Ray ray = make_ray(xf_point_affine(se.to_world, v3(0.f, 0.f, 0.f)), xf_vector(se.to_world, d), se.near_clip * inv_z, se.far_clip * inv_z);
        *weight = v3(1.f, 1.f, 1.f);
        return ray;
    }
    if (se.type == MTS_SENSOR_MRADIANCEMETER || se.type == MTS_SENSOR_MDISTANT) {
        // Int32 sensor_index(position_sample.x() * m_sensor_count): mradiancemeter.cpp:146, mdistant.cpp:231 (clamped: the
        // reference gathers without a bounds check and position_sample.x can round up to 1)
        int index = (int) (position_sample.x * (float) se.multi_count);
        index = std::min(std::max(index, 0), se.multi_count - 1);
        Xf trafo = {}; memcpy(trafo.m, &se.multi[16 * (size_t) index], 64);
        V3 d = xf_vector(trafo, v3(0.f, 0.f, 1.f));                                       // transform_affine(Vector3f{0, 0, 1})
        if (se.type == MTS_SENSOR_MRADIANCEMETER) {                                       // mradiancemeter.cpp:134-157
            *weight = v3(1.f, 1.f, 1.f);
            return make_ray(xf_point_affine(trafo, v3(0.f, 0.f, 0.f)), d, RayEpsilon, pm_inf());
        }
        V3 o; float w = 1.f;                                                              // mdistant.cpp:212-262
        if (se.target_type == MTS_DISTANT_TARGET_POINT) o = se.target_point - 2.f * d * se.bsphere_radius;
        else if (se.target_type == MTS_DISTANT_TARGET_SHAPE) {
            V3 tp, n; float pdf;
            shape_sample_position(se.target_shape, aperture_sample, &tp, &n, &pdf);
            float area = se.target_shape.type == MTS_SHAPE_DISK ? se.target_shape.surface_area : se.target_shape.type == MTS_SHAPE_RECTANGLE ? norm(cross(se.target_shape.frame.s, se.target_shape.frame.t))
                                                                     : 4.f * Pi * se.target_shape.radius * se.target_shape.radius;
            o = tp - 2.f * d * se.bsphere_radius;
            w = 1.f / (pdf * area);
        } else {
            P2 offset = square_to_uniform_disk_concentric(aperture_sample);
            V3 perp_offset = xf_vector(trafo, v3(offset.x, offset.y, 0.f));
            o = se.bsphere_center + perp_offset * se.bsphere_radius - d * se.bsphere_radius;
        }
        *weight = v3(w, w, w);
        return make_ray(o, d, RayEpsilon, pm_inf());
    }
    if (se.type == MTS_SENSOR_DISTANTFLUX) {                                              // distantflux.cpp:189-240
        V3 d = -xf_vector(se.to_world, square_to_uniform_hemisphere(position_sample));
        V3 reference_normal = xf_vector(se.to_world, v3(0.f, 0.f, 1.f));                  // distantflux.cpp:185-186
        // dot(-ray.d, n) / (square_to_uniform_hemisphere_pdf(ray.d) * m_npixels), warp.h:312-320 (no domain test: 1 / 2 pi)
        float w = dot(-d, reference_normal) / (InvTwoPi * (float) ((uint32_t) se.width * (uint32_t) se.height));
        V3 ray_target = se.target_point;
        if (se.target_type == MTS_DISTANT_TARGET_SHAPE) {
            V3 n; float pdf;
            shape_sample_position(se.target_shape, aperture_sample, &ray_target, &n, &pdf);
            float area = se.target_shape.type == MTS_SHAPE_DISK ? se.target_shape.surface_area : se.target_shape.type == MTS_SHAPE_RECTANGLE ? norm(cross(se.target_shape.frame.s, se.target_shape.frame.t))
                                                                     : 4.f * Pi * se.target_shape.radius * se.target_shape.radius;
            w *= 1.f / (pdf * area);
        } else if (se.target_type == MTS_DISTANT_TARGET_NONE) {
            P2 offset = square_to_uniform_disk_concentric(aperture_sample);
            V3 perp_offset = xf_vector(se.to_world, v3(offset.x, offset.y, 0.f));
            ray_target = se.bsphere_center + perp_offset * se.bsphere_radius;
        }
        V3 o = ray_target - d * 2.f * se.bsphere_radius;
        if (se.origin_is_shape) {                                                         // distantflux.cpp:244-252
            if (!shape_hit_point(se.origin_shape, make_ray(ray_target, -d, RayEpsilon, pm_inf()), &o)) { o = v3(pm_nan(), pm_nan(), pm_nan()); w = 0.f; }
        }
        *weight = v3(w, w, w);
        return make_ray(o, d, RayEpsilon, pm_inf());
    }
    // distant.cpp:299-386
    V3 v0 = v3(0.f, 0.f, 1.f);
    if (se.direction_type == 2) v0 = square_to_uniform_hemisphere(position_sample);
    else if (se.direction_type == 1) { float s, c; pm_sincos(Pi * position_sample.x, &s, &c); v0.x = c; v0.z = s; }
    V3 d = se.flip_directions ? xf_vector(se.to_world, v0) : xf_vector(se.to_world, -v0);
    V3 ray_weight, ray_target = se.target_point, o;
    if (se.target_type == MTS_DISTANT_TARGET_POINT) ray_weight = v3(1.f, 1.f, 1.f);
    else if (se.target_type == MTS_DISTANT_TARGET_SHAPE) {
        V3 n; float pdf;
        shape_sample_position(se.target_shape, aperture_sample, &ray_target, &n, &pdf);
        float area = se.target_shape.type == MTS_SHAPE_DISK ? se.target_shape.surface_area : se.target_shape.type == MTS_SHAPE_RECTANGLE ? norm(cross(se.target_shape.frame.s, se.target_shape.frame.t))
                                                                 : 4.f * Pi * se.target_shape.radius * se.target_shape.radius;
        float w = (1.f / pdf) * (1.f / area);                                               // Spectrum / Float / Float: each a reciprocal-multiply
        ray_weight = v3(w, w, w);
    } else {
        P2 offset = square_to_uniform_disk_concentric(aperture_sample);
        V3 perp_offset = xf_vector(se.to_world, v3(offset.x, offset.y, 0.f));
        ray_target = se.bsphere_center + perp_offset * se.bsphere_radius;
        float w = 1.f / dot(-d, v3(0.f, 0.f, 1.f));
        ray_weight = v3(w, w, w);
    }
    if (se.origin_is_shape) {                                                             // distant.cpp:368-375
        if (!shape_hit_point(se.origin_shape, make_ray(ray_target, -d, RayEpsilon, pm_inf()), &o)) { o = v3(pm_nan(), pm_nan(), pm_nan()); ray_weight = v3(0.f, 0.f, 0.f); }
    } else if (se.target_type == MTS_DISTANT_TARGET_NONE) o = ray_target - d * se.bsphere_radius;
    else o = ray_target - d * 2.f * se.bsphere_radius;
    *weight = ray_weight;
    Ray ray; ray.o = o; ray.d = d; ray.d_rcp = vrcp(d); ray.mint = RayEpsilon; ray.maxt = pm_inf();   // ray.h:33-34 defaults
    return ray;
}

// ---------------------------------------------------------------- image block / film
// imageblock.cpp:10-172
struct ImageBlock {
    int ox, oy, w, h, border, channels;
    const RFilter *filter;
    std::vector<float> data;
    void init(int w_, int h_, int channels_, const RFilter *f, bool use_border) {
        filter = f; channels = channels_; border = (f && use_border) ? f->border_size : 0; w = w_; h = h_; ox = oy = 0;
        data.assign((size_t) channels * (w + 2 * border) * (h + 2 * border), 0.f);
    }
    void set_size(int w_, int h_) { if (w_ == w && h_ == h) return; w = w_; h = h_; data.assign((size_t) channels * (w + 2 * border) * (h + 2 * border), 0.f); }
    void clear() { std::fill(data.begin(), data.end(), 0.f); }
    bool warn_negative = true;                                  // integrator.cpp:114-116: !has_aovs
    // imageblock.cpp:79-172 (scalar branch: discretised filter weights)
    bool put(P2 pos_, const float *value) {
        bool active = true;
        for (int k = 0; k < channels; ++k) active = active && (!warn_negative || value[k] >= -1e-5f) && pm_isfinite(value[k]);   // :85-109 (warn + drop)
        if (!active) return false;
        float filter_radius = filter->radius;
        int sx = w + 2 * border, sy = h + 2 * border;
        P2 pos = { pos_.x - ((float) (ox - border) + .5f), pos_.y - ((float) (oy - border) + .5f) };
        if (filter_radius > 0.5f + RayEpsilon) {
            int lox = std::max((int) pm_ceil(pos.x - filter_radius), 0), loy = std::max((int) pm_ceil(pos.y - filter_radius), 0);
            int hix = std::min((int) pm_floor(pos.x + filter_radius), sx - 1), hiy = std::min((int) pm_floor(pos.y + filter_radius), sy - 1);
            uint32_t n = (uint32_t) pm_ceil((filter->radius - 2.f * RayEpsilon) * 2.f);
            float wx[64], wy[64];
            float basex = (float) lox - pos.x, basey = (float) loy - pos.y;
            for (uint32_t i = 0; i < n; ++i) { wx[i] = filter->eval_discretized(basex + (float) i); wy[i] = filter->eval_discretized(basey + (float) i); }
            for (uint32_t yr = 0; yr < n; ++yr) {
                int y = loy + (int) yr;
                bool enabled = y <= hiy;
                for (uint32_t xr = 0; xr < n; ++xr) {
                    int x = lox + (int) xr;
                    float weight = wy[yr] * wx[xr];
                    enabled = enabled && x <= hix;
                    if (enabled) { size_t offset = (size_t) channels * ((size_t) y * sx + x); for (int k = 0; k < channels; ++k) data[offset + k] += value[k] * weight; }
                }
            }
        } else {
            int lox = (int) pm_ceil(pos.x - .5f), loy = (int) pm_ceil(pos.y - .5f);
            bool enabled = lox >= 0 && loy >= 0 && lox < sx && loy < sy;
            if (enabled) { size_t offset = (size_t) channels * ((size_t) loy * sx + lox); for (int k = 0; k < channels; ++k) data[offset + k] += value[k]; }
        }
        return true;
    }
    // imageblock.cpp:49-77 (accumulate_2d with clipping)
    void put_block(const ImageBlock &b) {
        int ssx = b.w + 2 * b.border, ssy = b.h + 2 * b.border, tsx = w + 2 * border, tsy = h + 2 * border;
        int offx = (b.ox - b.border) - (ox - border), offy = (b.oy - b.border) - (oy - border);
        for (int y = 0; y < ssy; ++y) {
            int ty = y + offy; if (ty < 0 || ty >= tsy) continue;
            for (int x = 0; x < ssx; ++x) {
                int tx = x + offx; if (tx < 0 || tx >= tsx) continue;
                for (int k = 0; k < channels; ++k) data[(size_t) channels * ((size_t) ty * tsx + tx) + k] += b.data[(size_t) channels * ((size_t) y * ssx + x) + k];
            }
        }
    }
};

// spiral.cpp:11-72
struct Spiral {
    int size_x, size_y, off_x, off_y, block_size, blocks_x, blocks_y;
    size_t block_count, block_counter, remaining_passes;
    int dir, pos_x, pos_y, steps_left, steps;
    void init(int sx, int sy, int ox, int oy, int bs, size_t passes) {
        size_x = sx; size_y = sy; off_x = ox; off_y = oy; block_size = bs; remaining_passes = passes;
        blocks_x = (int) std::ceil((float) sx / bs); blocks_y = (int) std::ceil((float) sy / bs);
        block_count = (size_t) blocks_x * blocks_y;
        reset();
    }
    void reset() { block_counter = 0; dir = 0; pos_x = blocks_x / 2; pos_y = blocks_y / 2; steps_left = 1; steps = 1; }
    bool next_block(int *ox, int *oy, int *sx, int *sy, size_t *block_id) {
        if (block_count == block_counter) {
            if (remaining_passes > 1) { --remaining_passes; reset(); }
            else { *ox = *oy = *sx = *sy = 0; *block_id = (size_t) -1; return false; }
        }
        *block_id = block_counter + (remaining_passes - 1) * block_count;
        int offx = pos_x * block_size, offy = pos_y * block_size;
        *sx = std::min(block_size, size_x - offx); *sy = std::min(block_size, size_y - offy);
        *ox = offx + off_x; *oy = offy + off_y;
        ++block_counter;
        if (block_counter != block_count) {
            do {
                switch (dir) { case 0: ++pos_x; break; case 1: ++pos_y; break; case 2: --pos_x; break; case 3: --pos_y; break; }
                if (--steps_left == 0) { dir = (dir + 1) % 4; if (dir == 2 || dir == 0) ++steps; steps_left = steps; }
            } while (pos_x < 0 || pos_y < 0 || pos_x >= blocks_x || pos_y >= blocks_y);
        }
        return true;
    }
};

#if MTS_SPEC_N != 3
// cie1931_xyz + spectrum_to_xyz (core/spectrum.h:148-178,210-217): XYZ = hmean(cmf(lambda) * value); the 95-sample tables of
// libcore/spectrum.cpp:110-189 (the published CIE 1931 standard observer, shared as data with the product: csrc/cie_tables.h)
#include "../eradiate-kernel_amd/csrc/cie_tables.h"
static inline void spectrum_to_xyz(Spec value, Spec wl, float xyz[3]) {
    const float lam[4] = { wl.x, wl.y, wl.z, wl.w }, val[4] = { value.x, value.y, value.z, value.w };
    float c[3][4];
    for (int k = 0; k < 4; ++k) {
        const float t = (lam[k] - 360.f) * ((95 - 1) / (830.f - 360.f));
        const bool active = lam[k] >= 360.f && lam[k] <= 830.f;
        const int i0 = std::min(std::max((int) t, 0), 95 - 2), i1 = i0 + 1;
        const float w1 = t - (float) i0, w0 = 1.f - w1;
        for (int a = 0; a < 3; ++a) {
            const float cmf = active ? pm_fma(w0, MTS_CIE1931_XYZ[95 * a + i0], w1 * MTS_CIE1931_XYZ[95 * a + i1]) : 0.f;
            c[a][k] = cmf * val[k];
        }
    }
    for (int a = 0; a < 3; ++a) xyz[a] = spec_hmean(spec4(c[a][0], c[a][1], c[a][2], c[a][3]));
}
#endif

#if MTS_SPEC_N != 3
// Texture::sample_spectrum of the two response functions a sensor's "srf" may be: uniform.cpp:92-100, discrete.cpp:124-133
// (DiscreteDistribution::sample, distr_1d.h:141-151)
static inline void spectrum_sample(const SpectrumRec &r, float x, float *wavelength, float *weight) {
    if (r.type == MTS_SPECTRUM_UNIFORM) { *wavelength = r.lambda_min + (r.lambda_max - r.lambda_min) * x; *weight = r.value * (r.lambda_max - r.lambda_min); }
    else {
        uint32_t index = distr_binary_search(r.cdf, r.valid_x, r.valid_y, x * r.cdf_sum);
        *wavelength = r.wavelengths[index]; *weight = r.values[index];
    }
}
#endif

// ---------------------------------------------------------------- render driver
// integrator.cpp:233-288
static void render_sample(const Scene &sc, Sampler &sampler, ImageBlock &block, float px, float py, Counters *cnt) {
    const Sensor &se = sc.sensor;
    P2 u = sampler.next_2d();
    P2 position_sample = { px + u.x, py + u.y };
    P2 aperture_sample = { .5f, .5f };
    if (se.needs_aperture_sample) aperture_sample = sampler.next_2d();
    if (se.shutter_open_time > 0.f) (void) sampler.next_1d();      // integrator.cpp:248-250: the time sample (nothing is animated)
    float wavelength_sample = sampler.next_1d(); (void) wavelength_sample;
#if MTS_SPEC_N != 3
    // Sensor::sample_ray -> sample_wavelength<Float, Spectrum> (perspective.cpp:169-172, distant.cpp:311-313; core/spectrum.h:305-314):
    // math::sample_shifted (core/math.h:419-442), then sample_rgb_spectrum, which for MTS_WAVELENGTH_MIN / MAX = 280 / 2400 falls back
    // to sample_uniform_spectrum -- written over the CIE range: lambda = s (830 - 360) + 360, weight 830 - 360 (:248-252,266-285)
    Spec wav_weight = spec_s(830.f - 360.f);
    {
        float v[4], wgt[4];
        for (int k = 0; k < 4; ++k) {
            float x = wavelength_sample + (float) k / 4.f; if (x > 1.f) x -= 1.f;
            if (se.srf < 0) { v[k] = x * (830.f - 360.f) + 360.f; wgt[k] = 830.f - 360.f; }
            else spectrum_sample(sc.spectra[(size_t) se.srf], x, &v[k], &wgt[k]);   // perspective.cpp:173-182, radiancemeter.cpp:116-124: the response function draws the wavelengths
        }
        tls_wavelengths = spec4(v[0], v[1], v[2], v[3]);
        wav_weight = spec4(wgt[0], wgt[1], wgt[2], wgt[3]);
    }
#endif
    P2 adjusted = { (position_sample.x - (float) se.crop_x) / (float) se.crop_w, (position_sample.y - (float) se.crop_y) / (float) se.crop_h };
    V3 ray_weight;
    Ray ray = sensor_sample_ray(sc, adjusted, aperture_sample, &ray_weight);
    bool valid;
    float aovs[5 + 2 * 64];
#if MTS_SPEC_N == 3
    V3 L = integrator_sample(sc, sampler, ray, se.medium, &valid, cnt);
    L = ray_weight * L;
    // srgb_to_xyz, spectrum.h:221-227 (matrix * vector = fmadd chain over columns)
    aovs[0] = pm_fma(0.180423f, L.z, pm_fma(0.357580f, L.y, 0.412453f * L.x));
    aovs[1] = pm_fma(0.072169f, L.z, pm_fma(0.715160f, L.y, 0.212671f * L.x));
    aovs[2] = pm_fma(0.950227f, L.z, pm_fma(0.119193f, L.y, 0.019334f * L.x));
    if (sc.integrator.monochrome)                              // integrator.cpp:270-271: xyz = spec_u.x()
        aovs[0] = aovs[1] = aovs[2] = L.x;
#else
    Spec L = integrator_sample(sc, sampler, ray, se.medium, &valid, cnt);
    // nbins.cpp:100-125 / bins.cpp:88-110: the wrapped integrator's own result (before the ray weight), per bin the sum over the
    // sample's wavelengths inside the bin and their number (hsum of a 4-array: (x + y) + (z + w), as spec_hmean)
    for (int i = 0; i < sc.integrator.bin_count; ++i) {
        const float wl[4] = { tls_wavelengths.x, tls_wavelengths.y, tls_wavelengths.z, tls_wavelengths.w }, lv[4] = { L.x, L.y, L.z, L.w };
        float val[4], pop[4];
        for (int k = 0; k < 4; ++k) {
            if (sc.integrator.bin_mode == 1) {
                const bool in = pm_abs(wl[k] - sc.bin_lo[i]) <= sc.bin_hi[i];
                val[k] = in ? lv[k] : 0.f; pop[k] = in ? 1.f : 0.f;
            } else {
                const float w = (wl[k] >= sc.bin_lo[i] && wl[k] <= sc.bin_hi[i]) ? 1.f : 0.f;      // UniformSpectrum(lower, upper, 1).eval
                val[k] = w * lv[k]; pop[k] = w;
            }
        }
        aovs[5 + 2 * i] = (val[0] + val[1]) + (val[2] + val[3]);
        aovs[5 + 2 * i + 1] = (pop[0] + pop[1]) + (pop[2] + pop[3]);
    }
    L = (wav_weight * ray_weight.x) * L;                       // ray_weight = wav_weight (x the sensor's grey weight), integrator.cpp:265
    spectrum_to_xyz(L, tls_wavelengths, aovs);                 // integrator.cpp:266-269
#endif
    aovs[3] = valid ? 1.f : 0.f;
    aovs[4] = 1.f;
    block.put(position_sample, aovs);
}

// integrator.cpp:181-209 (scalar branch)
static void render_block(const Scene &sc, Sampler &sampler, ImageBlock &block, uint32_t block_size, size_t sample_count, size_t block_id, Counters *cnt,
                         const std::atomic<int> *stop) {
    block.clear();
    uint32_t pixel_count = block_size * block_size;
    for (uint32_t i = 0; i < pixel_count && !(stop && stop->load(std::memory_order_relaxed)); ++i) {
        sampler.seed(block_id * pixel_count + i);
        uint32_t x, y; morton_decode(i, &x, &y);
        if (x >= (uint32_t) block.w || y >= (uint32_t) block.h) continue;
        float px = (float) (x + block.ox), py = (float) (y + block.oy);
        for (size_t j = 0; j < sample_count; ++j) {
            if (sc.sensor.wavefront) {
                // The streams of the wavefront (gpu_*) variants: lane L = pixel * spp + sample is seeded with (sample_tea_64(seed, L),
                // sample_tea_64(L, seed)) in 64-bit arithmetic (librender/sampler.cpp:89-92), pixel = y * width + x inside the crop
                // window (integrator.cpp:143-163).  One stream per (pixel, sample) instead of one per pixel.
                const uint64_t pixel = (uint64_t) (y + (uint32_t) block.oy - (uint32_t) sc.sensor.crop_y) * (uint64_t) sc.sensor.crop_w +
                                       (uint64_t) (x + (uint32_t) block.ox - (uint32_t) sc.sensor.crop_x);
                const uint64_t L = pixel * (uint64_t) sc.sensor.sample_count + (uint64_t) j;
                sampler.rng.seed(sample_tea_64_u64(sampler.base_seed, L), sample_tea_64_u64(L, sampler.base_seed));
            }
            render_sample(sc, sampler, block, px, py, cnt);
        }
    }
}

struct OracleScene { Scene *scene; std::atomic<int> stop; };

// integrator.cpp:51-179 (CPU branch; TBB replaced by std::thread workers pulling blocks)
static int render(OracleScene *os, int n_threads, int shard_index, int shard_count, float *film_out, mts_stats *stats) {
    const Scene &sc = *os->scene;
    const Sensor &se = sc.sensor;
    auto t0 = std::chrono::steady_clock::now();
    os->stop = 0;
    size_t total_spp = (size_t) se.sample_count;
    size_t samples_per_pass = sc.integrator.samples_per_pass < 0 ? total_spp : std::min((size_t) sc.integrator.samples_per_pass, total_spp);
    if (samples_per_pass == 0 || (total_spp % samples_per_pass) != 0) throw std::runtime_error("sample_count must be a multiple of samples_per_pass");
    size_t n_passes = (total_spp + samples_per_pass - 1) / samples_per_pass;
    // block size: pinned to MTS_BLOCK_SIZE = 32 when unspecified (SURVEY.md section 7 "hard parts": the
    // reference's heuristic depends on the thread count, integrator.cpp:89-97)
    uint32_t block_size = sc.integrator.block_size > 0 ? (uint32_t) sc.integrator.block_size : 32u;
    { uint32_t p = 1; while (p < block_size) p <<= 1; block_size = p; }                       // integrator.cpp:26-32
    const int channels = 5 + 2 * sc.integrator.bin_count;           // integrator.cpp:67-76: X, Y, Z, A, W + aov_names()
    if (sc.integrator.bin_count > 64) throw std::runtime_error("at most 64 bins");
    ImageBlock film; film.init(se.crop_w, se.crop_h, channels, nullptr, false); film.ox = se.crop_x; film.oy = se.crop_y;   // hdrfilm.cpp:201-203
    Spiral spiral; spiral.init(se.crop_w, se.crop_h, se.crop_x, se.crop_y, (int) block_size, n_passes);
    size_t total_blocks = spiral.block_count * n_passes;
    std::mutex spiral_mutex, film_mutex;
    Counters total; uint64_t samples = 0;
    size_t next = 0;
    auto worker = [&]() {
        const unsigned saved_csr = _mm_getcsr();
        _mm_setcsr(saved_csr | 0x8040);                                                        // scoped_flush_denormals, integrator.cpp:117
        Sampler sampler; sampler.base_seed = se.seed;
        ImageBlock block; block.init((int) block_size, (int) block_size, channels, &se.rfilter, true);
        block.warn_negative = sc.integrator.bin_count == 0;                                     // integrator.cpp:114-116
        Counters cnt; uint64_t my_samples = 0;
        for (;;) {
            int ox, oy, sx, sy; size_t block_id;
            {
                std::lock_guard<std::mutex> lock(spiral_mutex);
                if (next >= total_blocks || os->stop.load()) break;
                ++next;
                if (!spiral.next_block(&ox, &oy, &sx, &sy, &block_id)) break;
            }
            if (shard_count > 1 && (int) (block_id % (size_t) shard_count) != shard_index) continue;
            block.set_size(sx, sy); block.ox = ox; block.oy = oy;
            render_block(sc, sampler, block, block_size, samples_per_pass, block_id, stats ? &cnt : nullptr, &os->stop);
            my_samples += (uint64_t) sx * sy * samples_per_pass;
            std::lock_guard<std::mutex> lock(film_mutex);
            film.put_block(block);
        }
        _mm_setcsr(saved_csr);                                                                 // the caller's thread may run this inline
        std::lock_guard<std::mutex> lock(film_mutex);
        total.n_iter += cnt.n_iter; total.n_lookup += cnt.n_lookup; total.n_nee_step += cnt.n_nee_step; samples += my_samples;
    };
    if (n_threads <= 1) worker();
    else { std::vector<std::thread> th; for (int i = 0; i < n_threads; ++i) th.emplace_back(worker); for (auto &t : th) t.join(); }
    memcpy(film_out, film.data.data(), film.data.size() * sizeof(float));
    if (stats) {
        memset(stats, 0, sizeof(*stats));
        stats->samples = samples; stats->n_iter = total.n_iter; stats->n_lookup = total.n_lookup; stats->n_nee_step = total.n_nee_step;
        stats->wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        stats->cancelled = os->stop.load();
    }
    return 0;
}

} // namespace orc

// ================================================================== C interface (ctypes)
using namespace orc;
static thread_local std::string g_error;
#define ORC_TRY try {
#define ORC_CATCH } catch (const std::exception &e) { g_error = e.what(); return 1; } catch (...) { g_error = "unknown error"; return 1; } return 0;

extern "C" {

const char *oracle_last_error(void) { return g_error.c_str(); }

int oracle_scene_create(const mts_scene_desc *desc, oracle_scene **out) {
    ORC_TRY
    OracleScene *os = new OracleScene(); os->scene = make_scene(desc); os->stop = 0;
    *out = (oracle_scene *) os;
    ORC_CATCH
}
int oracle_scene_destroy(oracle_scene *s) { OracleScene *os = (OracleScene *) s; if (os) { delete os->scene; delete os; } return 0; }
int oracle_cancel(oracle_scene *s) { ((OracleScene *) s)->stop = 1; return 0; }

int oracle_render(oracle_scene *s, int n_threads, int shard_index, int shard_count, float *film, mts_stats *stats) {
    ORC_TRY
    render((OracleScene *) s, n_threads, shard_index, shard_count < 1 ? 1 : shard_count, film, stats);
    ORC_CATCH
}

int oracle_sample(oracle_scene *s, int32_t n, uint64_t seed_offset, const float *ox, const float *oy, const float *oz,
                  const float *dx, const float *dy, const float *dz, float *out_rgb, uint8_t *out_valid) {
    ORC_TRY
    const Scene &sc = *((OracleScene *) s)->scene;
    unsigned csr = _mm_getcsr(); _mm_setcsr(csr | 0x8040);
    for (int i = 0; i < n; ++i) {
        Sampler sampler; sampler.base_seed = sc.sensor.seed; sampler.seed(seed_offset + (uint64_t) i);
        Ray ray = make_ray(v3(ox[i], oy[i], oz[i]), v3(dx[i], dy[i], dz[i]), RayEpsilon, pm_inf());
        bool valid;
        Spec L = integrator_sample(sc, sampler, ray, sc.sensor.medium, &valid, nullptr);      // spectral build: at the wavelengths of oracle_set_wavelengths, first three entries
        out_rgb[3 * i] = L.x; out_rgb[3 * i + 1] = L.y; out_rgb[3 * i + 2] = L.z; out_valid[i] = valid;
    }
    _mm_setcsr(csr);
    ORC_CATCH
}

int oracle_ray_intersect(oracle_scene *s, int32_t n, const float *o, const float *d, const float *mint, const float *maxt,
                         float *out_t, int32_t *out_shape, int32_t *out_prim, float *out_p, float *out_n) {
    ORC_TRY
    const Scene &sc = *((OracleScene *) s)->scene;
    for (int i = 0; i < n; ++i) {
        Ray ray = make_ray(v3(o[3 * i], o[3 * i + 1], o[3 * i + 2]), v3(d[3 * i], d[3 * i + 1], d[3 * i + 2]), mint[i], maxt[i]);
        SurfaceInteraction si = ray_intersect(sc, ray);
        out_t[i] = si.t; out_shape[i] = si.shape; out_prim[i] = si.is_valid() ? si.prim_index : -1;
        out_p[3 * i] = si.p.x; out_p[3 * i + 1] = si.p.y; out_p[3 * i + 2] = si.p.z;
        out_n[3 * i] = si.n.x; out_n[3 * i + 1] = si.n.y; out_n[3 * i + 2] = si.n.z;
    }
    ORC_CATCH
}

// ---- unit-level entry points for the known-answer tests ----
uint32_t oracle_tea32(uint32_t v0, uint32_t v1, int rounds) { return sample_tea_32(v0, v1, rounds); }
uint64_t oracle_tea64(uint32_t v0, uint32_t v1, int rounds) { return sample_tea_64(v0, v1, rounds); }
float oracle_tea_float32(uint32_t v0, uint32_t v1, int rounds) { return sample_tea_float32(v0, v1, rounds); }
void oracle_pcg32(uint64_t initstate, uint64_t initseq, int n, uint32_t *out_u32, float *out_f32) {
    PCG32 a, b; a.seed(initstate, initseq); b = a;
    for (int i = 0; i < n; ++i) { if (out_u32) out_u32[i] = a.next_uint32(); if (out_f32) out_f32[i] = b.next_float32(); }
}
void oracle_sampler_stream(uint64_t base_seed, uint64_t seed_offset, int n, float *out) {
    Sampler s; s.base_seed = base_seed; s.seed(seed_offset); for (int i = 0; i < n; ++i) out[i] = s.next_1d();
}
// PCG32Sampler::seed of the wavefront variants (sampler.cpp:83-92): lane idx seeded with (tea64(seed_value, idx), tea64(idx, seed_value))
void oracle_wavefront_sampler(int lanes, uint64_t seed_value, int count, float *out) {
    for (int i = 0; i < lanes; ++i) {
        PCG32 rng; rng.seed(sample_tea_64_u64(seed_value, (uint64_t) i), sample_tea_64_u64((uint64_t) i, seed_value));
        for (int k = 0; k < count; ++k) out[(size_t) i * count + k] = rng.next_float32();
    }
}
void oracle_warp(int kind, float u, float v, float *out) {
    P2 s = { u, v };
    if (kind == 0) { P2 p = square_to_uniform_disk_concentric(s); out[0] = p.x; out[1] = p.y; out[2] = 0.f; }
    else { V3 r = kind == 1 ? square_to_uniform_sphere(s) : (kind == 2 ? square_to_uniform_hemisphere(s) : square_to_cosine_hemisphere(s)); out[0] = r.x; out[1] = r.y; out[2] = r.z; }
}
void oracle_coordinate_system(const float *n, float *s, float *t) { V3 a, b; coordinate_system(v3(n[0], n[1], n[2]), &a, &b); s[0] = a.x; s[1] = a.y; s[2] = a.z; t[0] = b.x; t[1] = b.y; t[2] = b.z; }
void oracle_morton_decode(uint32_t i, uint32_t *x, uint32_t *y) { morton_decode(i, x, y); }
int oracle_spiral(int sx, int sy, int ox, int oy, int block_size, int passes, int max_blocks, int32_t *out /* 5 per block: ox, oy, sx, sy, id */) {
    Spiral sp; sp.init(sx, sy, ox, oy, block_size, (size_t) passes);
    int n = 0;
    for (; n < max_blocks; ++n) { int a, b, c, d; size_t id; if (!sp.next_block(&a, &b, &c, &d, &id)) break; out[5 * n] = a; out[5 * n + 1] = b; out[5 * n + 2] = c; out[5 * n + 3] = d; out[5 * n + 4] = (int32_t) id; }
    return n;
}
// ImageBlock::put for a list of samples; returns the block storage including borders
int oracle_imageblock_put(int w, int h, int ox, int oy, int channels, int rfilter_type, float radius, float stddev, int use_border,
                          int n, const float *pos /* 2n */, const float *values /* channels*n */, float *out, int *out_border) {
    ORC_TRY
    RFilter f = make_rfilter(rfilter_type, radius, stddev);
    ImageBlock b; b.init(w, h, channels, &f, use_border != 0); b.ox = ox; b.oy = oy;
    for (int i = 0; i < n; ++i) { P2 p = { pos[2 * i], pos[2 * i + 1] }; b.put(p, values + (size_t) channels * i); }
    memcpy(out, b.data.data(), b.data.size() * sizeof(float));
    *out_border = b.border;
    ORC_CATCH
}
float oracle_rfilter_eval(int rfilter_type, float radius, float stddev, float x, int discretized) {
    RFilter f = make_rfilter(rfilter_type, radius, stddev); return discretized ? f.eval_discretized(x) : f.eval(x);
}
// phase function unit access: wi is the incident direction stored in mi.wi (= -ray.d), p the position
int oracle_phase_eval(oracle_scene *s, int phase, const float *wi, const float *p, const float *wo, float *out) {
    ORC_TRY
    const Scene &sc = *((OracleScene *) s)->scene;
    MediumInteraction mi; memset(&mi, 0, sizeof(mi));
    mi.wi = v3(wi[0], wi[1], wi[2]); mi.sh_frame = frame_from_normal(-mi.wi); mi.p = v3(p[0], p[1], p[2]);
    *out = phase_eval(sc, phase, mi, v3(wo[0], wo[1], wo[2]));
    ORC_CATCH
}
int oracle_phase_sample(oracle_scene *s, int phase, const float *wi, const float *p, float s1, float s2x, float s2y, float *wo, float *pdf) {
    ORC_TRY
    const Scene &sc = *((OracleScene *) s)->scene;
    MediumInteraction mi; memset(&mi, 0, sizeof(mi));
    mi.wi = v3(wi[0], wi[1], wi[2]); mi.sh_frame = frame_from_normal(-mi.wi); mi.p = v3(p[0], p[1], p[2]);
    P2 s2 = { s2x, s2y }; V3 w;
    phase_sample(sc, phase, mi, s1, s2, &w, pdf);
    wo[0] = w.x; wo[1] = w.y; wo[2] = w.z;
    ORC_CATCH
}
int oracle_phase_eval_component(oracle_scene *s, int phase, int component, const float *wi, const float *p, const float *wo, float *out, int *component_count) {
    ORC_TRY
    const Scene &sc = *((OracleScene *) s)->scene;
    MediumInteraction mi; memset(&mi, 0, sizeof(mi));
    mi.wi = v3(wi[0], wi[1], wi[2]); mi.sh_frame = frame_from_normal(-mi.wi); mi.p = v3(p[0], p[1], p[2]);
    *out = phase_eval_component(sc, phase, mi, v3(wo[0], wo[1], wo[2]), component);
    *component_count = phase_component_count(sc, phase);
    ORC_CATCH
}
int oracle_phase_sample_component(oracle_scene *s, int phase, int component, const float *wi, const float *p, float s1, float s2x, float s2y, float *wo, float *pdf) {
    ORC_TRY
    const Scene &sc = *((OracleScene *) s)->scene;
    MediumInteraction mi; memset(&mi, 0, sizeof(mi));
    mi.wi = v3(wi[0], wi[1], wi[2]); mi.sh_frame = frame_from_normal(-mi.wi); mi.p = v3(p[0], p[1], p[2]);
    P2 s2 = { s2x, s2y }; V3 w;
    phase_sample_component(sc, phase, mi, s1, s2, component, &w, pdf);
    wo[0] = w.x; wo[1] = w.y; wo[2] = w.z;
    ORC_CATCH
}
// BSDF unit access in the local shading frame
int oracle_bsdf_eval(oracle_scene *s, int bsdf, const float *wi, const float *wo, float *value, float *pdf) {
    ORC_TRY
    const Scene &sc = *((OracleScene *) s)->scene;
    SurfaceInteraction si; memset(&si, 0, sizeof(si)); si.wi = v3(wi[0], wi[1], wi[2]);
    Spec v = bsdf_eval(sc.bsdfs[bsdf], si, v3(wo[0], wo[1], wo[2]));
    value[0] = v.x; value[1] = v.y; value[2] = v.z; *pdf = bsdf_pdf(sc.bsdfs[bsdf], si, v3(wo[0], wo[1], wo[2]));
    ORC_CATCH
}
int oracle_bsdf_sample(oracle_scene *s, int bsdf, const float *wi, float s1, float s2x, float s2y, float *wo, float *pdf, float *weight, uint32_t *sampled_type) {
    ORC_TRY
    const Scene &sc = *((OracleScene *) s)->scene;
    SurfaceInteraction si; memset(&si, 0, sizeof(si)); si.wi = v3(wi[0], wi[1], wi[2]);
    BSDFSample bs; P2 s2 = { s2x, s2y };
    Spec w = bsdf_sample(sc.bsdfs[bsdf], si, s1, s2, &bs);
    wo[0] = bs.wo.x; wo[1] = bs.wo.y; wo[2] = bs.wo.z; *pdf = bs.pdf; weight[0] = w.x; weight[1] = w.y; weight[2] = w.z; *sampled_type = bs.sampled_type;
    ORC_CATCH
}
int oracle_volume_eval(oracle_scene *s, int volume, int n, const float *p, float *out) {
    ORC_TRY
    const Scene &sc = *((OracleScene *) s)->scene;
    for (int i = 0; i < n; ++i) { Spec r = volume_eval(sc.volumes[volume], v3(p[3 * i], p[3 * i + 1], p[3 * i + 2])); out[3 * i] = r.x; out[3 * i + 1] = r.y; out[3 * i + 2] = r.z; }
    ORC_CATCH
}
// sensor rays for given film positions (normalised film coordinates) and aperture samples
int oracle_sensor_sample_ray(oracle_scene *s, int n, const float *film_sample, const float *aperture_sample, float *o, float *d, float *weight) {
    ORC_TRY
    const Scene &sc = *((OracleScene *) s)->scene;
    for (int i = 0; i < n; ++i) {
        P2 f = { film_sample[2 * i], film_sample[2 * i + 1] }, a = { aperture_sample[2 * i], aperture_sample[2 * i + 1] };
        V3 w; Ray r = sensor_sample_ray(sc, f, a, &w);
        o[3 * i] = r.o.x; o[3 * i + 1] = r.o.y; o[3 * i + 2] = r.o.z; d[3 * i] = r.d.x; d[3 * i + 1] = r.d.y; d[3 * i + 2] = r.d.z;
        weight[3 * i] = w.x; weight[3 * i + 1] = w.y; weight[3 * i + 2] = w.z;
    }
    ORC_CATCH
}
int oracle_emitter_sample_direction(oracle_scene *s, const float *ref_p, float u, float v, float *d, float *dist, float *pdf, float *spec) {
    ORC_TRY
    const Scene &sc = *((OracleScene *) s)->scene;
    P2 smp = { u, v }; Spec sp;
    DirectionSample ds = sample_emitter_direction(sc, v3(ref_p[0], ref_p[1], ref_p[2]), smp, false, &sp);
    d[0] = ds.d.x; d[1] = ds.d.y; d[2] = ds.d.z; *dist = ds.dist; *pdf = ds.pdf; spec[0] = sp.x; spec[1] = sp.y; spec[2] = sp.z;
    ORC_CATCH
}
// Spectral build: the wavelengths the hooks above evaluate at, and four-wide evaluations for the reference's spectral unit tests
// (src/spectra/tests/test_uniform.py, test_regular.py, test_d65.py; src/textures/tests/test_gridvolume_spectral.py)
int oracle_spec_n(void) { return MTS_SPEC_N; }
#if MTS_SPEC_N != 3
int oracle_set_wavelengths(const float *w) { tls_wavelengths = spec4(w[0], w[1], w[2], w[3]); return 0; }
// SamplingIntegrator::sample for rays that carry their own wavelengths (Ray::wavelengths, core/ray.h:36); four-wide result
int oracle_sample_spectral(oracle_scene *s, int32_t n, uint64_t seed_offset, const float *ox, const float *oy, const float *oz,
                           const float *dx, const float *dy, const float *dz, const float *wavelengths, float *out_spec, uint8_t *out_valid) {
    ORC_TRY
    const Scene &sc = *((OracleScene *) s)->scene;
    unsigned csr = _mm_getcsr(); _mm_setcsr(csr | 0x8040);
    const Spec saved = tls_wavelengths;
    for (int i = 0; i < n; ++i) {
        Sampler sampler; sampler.base_seed = sc.sensor.seed; sampler.seed(seed_offset + (uint64_t) i);
        Ray ray = make_ray(v3(ox[i], oy[i], oz[i]), v3(dx[i], dy[i], dz[i]), RayEpsilon, pm_inf());
        tls_wavelengths = spec4(wavelengths[4 * i], wavelengths[4 * i + 1], wavelengths[4 * i + 2], wavelengths[4 * i + 3]);
        bool valid;
        Spec L = integrator_sample(sc, sampler, ray, sc.sensor.medium, &valid, nullptr);
        out_spec[4 * i] = L.x; out_spec[4 * i + 1] = L.y; out_spec[4 * i + 2] = L.z; out_spec[4 * i + 3] = L.w; out_valid[i] = valid;
    }
    tls_wavelengths = saved;
    _mm_setcsr(csr);
    ORC_CATCH
}
int oracle_spectrum_eval(oracle_scene *s, int spectrum, const float *w, float *out) {
    ORC_TRY
    const Scene &sc = *((OracleScene *) s)->scene;
    if (spectrum < 0 || spectrum >= (int) sc.spectra.size()) throw std::runtime_error("spectrum index out of range");
    tls_wavelengths = spec4(w[0], w[1], w[2], w[3]);
    Spec r = color_eval(Color{ &sc.spectra[(size_t) spectrum] });
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
    ORC_CATCH
}
int oracle_volume_eval_spectral(oracle_scene *s, int volume, const float *p, const float *w, float *out) {
    ORC_TRY
    const Scene &sc = *((OracleScene *) s)->scene;
    tls_wavelengths = spec4(w[0], w[1], w[2], w[3]);
    Spec r = volume_eval(sc.volumes[volume], v3(p[0], p[1], p[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
    ORC_CATCH
}
int oracle_spectrum_sample(oracle_scene *s, int spectrum, const float *samples, int n, float *wavelengths, float *weights) {
    ORC_TRY
    const Scene &sc = *((OracleScene *) s)->scene;
    if (spectrum < 0 || spectrum >= (int) sc.spectra.size()) throw std::runtime_error("spectrum index out of range");
    const SpectrumRec &r = sc.spectra[(size_t) spectrum];
    if (r.type != MTS_SPECTRUM_UNIFORM && r.type != MTS_SPECTRUM_DISCRETE) throw std::runtime_error("sample_spectrum is restated for uniform and discrete spectra");
    for (int k = 0; k < n; ++k) spectrum_sample(r, samples[k], wavelengths + k, weights + k);
    ORC_CATCH
}
int oracle_spectrum_to_xyz(const float *value, const float *w, float *xyz) {
    spectrum_to_xyz(spec4(value[0], value[1], value[2], value[3]), spec4(w[0], w[1], w[2], w[3]), xyz);
    return 0;
}
#endif
// DiscreteDistribution / ContinuousDistribution hooks for the literals of src/libcore/tests/test_distr_1d.py
int oracle_discrete_distribution(const float *pmf, int n, const float *samples, int m, int32_t *index, float *reuse, float *pmf_norm,
                                 float *cdf_out /* n */, float *sum_norm /* 2 */) {
    ORC_TRY
    DiscreteDistribution d; d.pmf.assign(pmf, pmf + n);
    discrete_update(d);
    for (int i = 0; i < n; ++i) cdf_out[i] = d.cdf[i];
    sum_norm[0] = d.sum; sum_norm[1] = d.normalization;
    for (int i = 0; i < m; ++i) index[i] = (int32_t) discrete_sample_reuse(d, samples[i], &reuse[i], &pmf_norm[i]);
    ORC_CATCH
}
int oracle_continuous_distribution(float r0, float r1, const float *pdf, int n, const float *x, int m, float *eval_pdf_norm,
                                   float *eval_cdf_norm, float *sample, float *sample_pdf_norm, float *integral_norm /* 2 */) {
    ORC_TRY
    if (!(r0 < r1)) throw std::runtime_error("ContinuousDistribution: invalid range!");          // distr_1d.h:318-320
    ContinuousDistribution d; d.pdf.assign(pdf, pdf + n); d.range_x = r0; d.range_y = r1;
    distr_update(d);
    integral_norm[0] = d.integral; integral_norm[1] = d.normalization;
    for (int i = 0; i < m; ++i) {
        eval_pdf_norm[i] = distr_eval_pdf(d, x[i]) * d.normalization;
        {   // eval_cdf, distr_1d.h:400-417 (not on the render path; restated for the pin only)
            float xs = (x[i] - d.range_x) * d.inv_interval_size;
            uint32_t idx = (uint32_t) std::min(std::max((int64_t) xs, (int64_t) 0), (int64_t) d.pdf.size() - 2);
            float y0 = d.pdf[idx], y1 = d.pdf[idx + 1], c0 = idx > 0 ? d.cdf[idx - 1] : 0.f;
            float t = std::min(std::max(xs - (float) idx, 0.f), 1.f);
            eval_cdf_norm[i] = (c0 + t * (y0 + .5f * t * (y1 - y0)) * d.interval_size) * d.normalization;
        }
        sample[i] = distr_sample(d, x[i]);                    // x doubles as the uniform variate
        sample_pdf_norm[i] = distr_eval_pdf(d, sample[i]) * d.normalization;
    }
    ORC_CATCH
}
// math::solve_quadratic (src/libcore/tests/test_math.py:32-35) and BoundingBox3f (test_bbox.py:6-53) hooks
int oracle_solve_quadratic(double a, double b, double c, double *x0, double *x1) { return solve_quadratic_d(a, b, c, x0, x1) ? 1 : 0; }
int oracle_bbox_ops(int n_points, const float *points, int n_boxes, const float *boxes /* 6 each */, float *out_min, float *out_max,
                    int *valid, float *bsphere /* centre 3, radius */) {
    BBox b = bbox_empty();
    for (int i = 0; i < n_points; ++i) bbox_expand(b, v3(points[3 * i], points[3 * i + 1], points[3 * i + 2]));
    for (int i = 0; i < n_boxes; ++i) {
        BBox o; o.min = v3(boxes[6 * i], boxes[6 * i + 1], boxes[6 * i + 2]); o.max = v3(boxes[6 * i + 3], boxes[6 * i + 4], boxes[6 * i + 5]);
        bbox_expand(b, o);
    }
    out_min[0] = b.min.x; out_min[1] = b.min.y; out_min[2] = b.min.z; out_max[0] = b.max.x; out_max[1] = b.max.y; out_max[2] = b.max.z;
    *valid = bbox_valid(b) ? 1 : 0;
    V3 c = (b.min + b.max) * 0.5f;                            // BoundingBox::bounding_sphere, bbox.h:327-331
    bsphere[0] = c.x; bsphere[1] = c.y; bsphere[2] = c.z; bsphere[3] = norm(c - b.max);
    return 0;
}
float oracle_math(int fn, float x, float y) {
    switch (fn) { case 0: return pm_log(x); case 1: return pm_exp(x); case 2: { float s, c; pm_sincos(x, &s, &c); return s; }
                  case 3: { float s, c; pm_sincos(x, &s, &c); return c; } case 4: return pm_cbrt(x); case 5: return pm_pow(x, y);
                  // the correctly rounded routines (pmath.h: -DPM_CORRECTLY_ROUNDED, and the |x| >= 120 tail of pm_sincos)
                  case 6: return pm_log_cr(x); case 7: return pm_exp_cr(x); case 8: { float s, c; pm_sincos_cr(x, &s, &c); return s; }
                  case 9: { float s, c; pm_sincos_cr(x, &s, &c); return c; } case 10: return pm_cbrt_cr(x); case 11: return pm_pow_cr(x, y); }
    return 0.f;
}
void oracle_math_n(int fn, int64_t n, const float *x, const float *y, float *out) {
    for (int64_t i = 0; i < n; ++i) out[i] = oracle_math(fn, x[i], y ? y[i] : 0.f);
}

} // extern "C"
