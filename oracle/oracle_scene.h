// oracle_scene.h -- TEST INFRASTRUCTURE, not product code (see oracle_math.h).
//
// "Plugin constructors" of the CPU restatement: turns the C-ABI scene description
// (include/mtsamd.h, one POD record per reference plugin instance) into the state the reference
// plugins hold after construction.  Citations are relative to /root/reference.
#pragma once
#include <vector>
#include <deque>
#include <string>
#include <stdexcept>
#include <algorithm>
#include "oracle_math.h"
#include "../include/mtsamd.h"

namespace orc {

static inline Xf xf_from_abi(const mts_transform &t) { Xf x; memcpy(x.m, t.matrix, 64); memcpy(x.it, t.inverse_transpose, 64); return x; }

// ---------------------------------------------------------------- colour parameters of plugins
// rgb / mono: the colour itself.  Spectral: the plugin's spectrum texture (spectra/uniform.cpp, spectra/regular.cpp; d65 arrives
// expanded to regular), evaluated at the wavelengths of the sample in flight.
#if MTS_SPEC_N == 3
typedef V3 Color;
static inline Spec color_eval(const Color &c) { return c; }
#else
struct SpectrumRec { int type; float value, lambda_min, lambda_max; std::vector<float> values; float inv_interval_size;
                     std::vector<float> wavelengths, cdf; float cdf_sum = 0.f; uint32_t valid_x = 0, valid_y = 0; };   // irregular nodes / discrete: DiscreteDistribution of the pmf
struct Color { const SpectrumRec *s; };
// uniform.cpp:47-57 ; regular.cpp:71-78 -> ContinuousDistribution::eval_pdf (distr_1d.h:378-400)
static inline float spectrum_eval_1(const SpectrumRec &s, float lambda) {
    const bool active = lambda >= s.lambda_min && lambda <= s.lambda_max;
    if (s.type == MTS_SPECTRUM_UNIFORM) return active ? s.value : 0.f;
    if (s.type == MTS_SPECTRUM_DISCRETE) return 0.f;                       // discrete.cpp:112-116: a sampling-only response
    if (s.type == MTS_SPECTRUM_IRREGULAR) {                                // irregular.cpp:75-84 -> IrregularContinuousDistribution::eval_pdf (distr_1d.h:655-677)
        const uint32_t size = (uint32_t) s.wavelengths.size();
        uint32_t start = 0, end = size, iterations = 0;                    // enoki::binary_search(0, size, nodes[i] < x)
        { uint32_t diff = end - start; iterations = 1; while (diff >>= 1) iterations++; }
        for (uint32_t i = 0; i < iterations; ++i) {
            uint32_t middle = (start + end) >> 1;
            if (s.wavelengths[std::min(middle, size - 1)] < lambda) start = std::min(middle + 1, end); else end = middle;
        }
        uint32_t index = std::max(std::min(start, size - 1u), 1u) - 1u;
        float x0 = s.wavelengths[index], x1 = s.wavelengths[index + 1], y0 = s.values[index], y1 = s.values[index + 1];
        float x = (lambda - x0) / (x1 - x0);
        return active ? pm_fma(x, y1 - y0, y0) : 0.f;
    }
    float x = (lambda - s.lambda_min) * s.inv_interval_size;
    uint32_t index = (uint32_t) std::min(std::max((int64_t) x, (int64_t) 0), (int64_t) s.values.size() - 2);
    float y0 = active ? s.values[index] : 0.f, y1 = active ? s.values[index + 1] : 0.f;
    float w1 = x - (float) index, w0 = 1.f - w1;
    return pm_fma(w0, y0, w1 * y1);
}
static inline Spec color_eval(const Color &c) {
    const Spec wl = tls_wavelengths;
    return spec4(spectrum_eval_1(*c.s, wl.x), spectrum_eval_1(*c.s, wl.y), spectrum_eval_1(*c.s, wl.z), spectrum_eval_1(*c.s, wl.w));
}
#endif

// ---------------------------------------------------------------- Volume
// include/mitsuba/render/texture.h:210-279, src/librender/texture.cpp:89-92,
// src/textures/constant3d.cpp, src/textures/grid3d.cpp:131-161,362
struct Volume {
    int type;
    Color value;
    bool spectral_grid = false; float lambda_min = 0.f, lambda_max = 0.f;      // gridvolume_spectral.cpp:186-190
    Xf world_to_local;
    BBox bbox;
    const float *data;
    int nx, ny, nz, channels, filter, wrap;
    float max;
    bool has_max;
};

static inline Volume make_volume(const mts_volume &d) {
    Volume v = {};
    v.type = d.type;
#if MTS_SPEC_N == 3
    v.value = v3(d.value[0], d.value[1], d.value[2]);
    if (d.type == MTS_VOLUME_GRID_SPECTRAL) throw std::runtime_error("This volume data source can only be used with a spectral variant!");   // gridvolume_spectral.cpp:86-88
#else
    v.value.s = nullptr;                                                 // set by make_scene (needs the scene's spectra)
    if (d.type == MTS_VOLUME_GRID_SPECTRAL) {
        if (d.filter_type != MTS_FILTER_TRILINEAR) throw std::runtime_error("Invalid filter type, must be \"trilinear\"!");
        v.type = MTS_VOLUME_GRID; v.spectral_grid = true; v.lambda_min = d.lambda_min; v.lambda_max = d.lambda_max;
    }
#endif
    v.world_to_local = xf_inverse(xf_from_abi(d.to_world));           // texture.cpp:90
    v.has_max = false;
    if (v.type == MTS_VOLUME_GRID) {
        if (!d.data) throw std::runtime_error("gridvolume: missing data");
        if ((long) d.nx * d.ny * d.nz < 8)                                // volume_data.h:70-73
            throw std::runtime_error("Invalid grid dimensions (must have at least one value at each corner)");
        if (!v.spectral_grid && d.channels != 1 && d.channels != 3)       // grid3d.cpp:115-116
            throw std::runtime_error("Unsupported channel count (expected 1 or 3)");
#if MTS_SPEC_N != 3
        if (!v.spectral_grid && d.channels != 1)
            throw std::runtime_error("spectral variant: 3-channel grids need the sRGB upsampling model (ext/rgb2spec data, absent); use gridvolume_spectral");
#endif
        v.data = d.data; v.nx = d.nx; v.ny = d.ny; v.nz = d.nz; v.channels = d.channels;
        v.filter = d.filter_type; v.wrap = d.wrap_mode;
        float mx = -pm_inf();                                             // volume_data.h:86-98
        size_t n = (size_t) d.nx * d.ny * d.nz * d.channels;
        for (size_t i = 0; i < n; ++i) mx = std::max(mx, d.data[i]);
        v.max = mx; v.has_max = true;
        if (d.use_grid_bbox) {                                            // grid3d.cpp:152-155, volume_data.h:24-33
            V3 bmin = v3(d.file_bbox_min[0], d.file_bbox_min[1], d.file_bbox_min[2]);
            V3 bmax = v3(d.file_bbox_max[0], d.file_bbox_max[1], d.file_bbox_max[2]);
            Xf bt = xf_mul(xf_scale(vrcp(bmax - bmin)), xf_translate(-1.f * bmin));
            v.world_to_local = xf_mul(bt, v.world_to_local);
        }
        if (d.has_max_value) v.max = d.max_value;                         // grid3d.cpp:157-160
    }
    // update_bbox(), texture.h:262-269
    Xf inv = xf_inverse(v.world_to_local);
    V3 a = xf_point(inv, v3(0, 0, 0)), b = xf_point(inv, v3(1, 1, 1));
    v.bbox.min = a; v.bbox.max = a; bbox_expand(v.bbox, b);
    return v;
}

// ---------------------------------------------------------------- 1D distributions
// DiscreteDistribution, include/mitsuba/core/distr_1d.h:27-197 (pinned by src/libcore/tests/test_distr_1d.py:35-132)
struct DiscreteDistribution {
    std::vector<float> pmf, cdf;
    float sum = 0.f, normalization = 0.f;
    uint32_t valid_x = (uint32_t) -1, valid_y = (uint32_t) -1;
};
static inline void discrete_update(DiscreteDistribution &d) {                              // distr_1d.h:49-83
    size_t size = d.pmf.size();
    if (size == 0) throw std::runtime_error("DiscreteDistribution: empty distribution!");
    d.cdf.resize(size);
    d.valid_x = d.valid_y = (uint32_t) -1;
    double sum = 0.0;
    for (uint32_t i = 0; i < size; ++i) {
        double value = (double) d.pmf[i];
        sum += value;
        d.cdf[i] = (float) sum;
        if (value < 0.0) throw std::runtime_error("DiscreteDistribution: entries must be non-negative!");
        else if (value > 0.0) { if (d.valid_x == (uint32_t) -1) d.valid_x = i; d.valid_y = i; }
    }
    if (d.valid_x == (uint32_t) -1) throw std::runtime_error("DiscreteDistribution: no probability mass found!");
    d.sum = (float) sum; d.normalization = (float) (1.0 / sum);
}
// enoki::binary_search (absent source): the first index in [start, end] for which the predicate `cdf[i] < value` is false
static inline uint32_t distr_binary_search(const std::vector<float> &cdf, uint32_t start, uint32_t end, float value) {
    uint32_t iterations = 0;
    if (start < end) { uint32_t diff = end - start; iterations = 1; while (diff >>= 1) iterations++; }
    for (uint32_t i = 0; i < iterations; ++i) {
        uint32_t middle = (start + end) >> 1;
        if (cdf[middle] < value) start = std::min(middle + 1, end); else end = middle;
    }
    return start;
}
static inline uint32_t discrete_sample(const DiscreteDistribution &d, float value) {        // distr_1d.h:141-151
    return distr_binary_search(d.cdf, d.valid_x, d.valid_y, value * d.sum);
}
static inline uint32_t discrete_sample_reuse(const DiscreteDistribution &d, float value, float *reuse, float *pmf_out) {   // distr_1d.h:187-221
    uint32_t index = discrete_sample(d, value);
    float pmf = d.pmf[index] * d.normalization, cdf = index > 0 ? d.cdf[index - 1] * d.normalization : 0.f;
    *reuse = (value - cdf) / pmf; *pmf_out = pmf;
    return index;
}

// ---------------------------------------------------------------- Phase functions
// ContinuousDistribution, include/mitsuba/core/distr_1d.h:293-345
struct ContinuousDistribution {
    std::vector<float> pdf, cdf;
    float range_x, range_y, integral, normalization, interval_size, inv_interval_size;
    uint32_t valid_x, valid_y;
};
static inline void distr_update(ContinuousDistribution &d) {
    size_t size = d.pdf.size();
    if (size < 2) throw std::runtime_error("ContinuousDistribution: needs at least two entries!");
    d.cdf.resize(size - 1);
    d.valid_x = d.valid_y = (uint32_t) -1;
    double range = double(d.range_y) - double(d.range_x), interval_size = range / (size - 1), integral = 0.;
    for (size_t i = 0; i < size - 1; ++i) {
        double y0 = (double) d.pdf[i], y1 = (double) d.pdf[i + 1];
        double value = 0.5 * interval_size * (y0 + y1);
        integral += value;
        d.cdf[i] = (float) integral;
        if (y0 < 0. || y1 < 0.) throw std::runtime_error("ContinuousDistribution: entries must be non-negative!");
        else if (value > 0.) { if (d.valid_x == (uint32_t) -1) d.valid_x = (uint32_t) i; d.valid_y = (uint32_t) i; }
    }
    if (d.valid_x == (uint32_t) -1) throw std::runtime_error("ContinuousDistribution: no probability mass found!");
    d.integral = (float) integral; d.normalization = (float) (1. / integral);
    d.interval_size = (float) interval_size; d.inv_interval_size = (float) (1. / interval_size);
}

struct Phase {
    int type;
    float g;
    int child[2];
    int weight_volume;
    ContinuousDistribution distr;
};
static inline Phase make_phase(const mts_phase &d) {
    Phase p = {};
    p.type = d.type; p.g = d.g; p.child[0] = d.child[0]; p.child[1] = d.child[1]; p.weight_volume = d.weight_volume;
    if (d.type == MTS_PHASE_HG && (d.g >= 1 || d.g <= -1))              // hg.cpp:45-46
        throw std::runtime_error("The asymmetry parameter must lie in the interval (-1, 1)!");
    if (d.type == MTS_PHASE_TABULATED) {                                  // tabphase.cpp:33-51
        p.distr.pdf.assign(d.tab_values, d.tab_values + d.tab_count);
        p.distr.range_x = -1.f; p.distr.range_y = 1.f;
        distr_update(p.distr);
    }
    return p;
}

// ---------------------------------------------------------------- Medium
struct Medium {
    int type, sigma_t, albedo, phase;
    float scale;
    bool sample_emitters, has_spectral_extinction, is_homogeneous;
    float max_density;   // heterogeneous.cpp:29
    BBox aabb;           // heterogeneous.cpp:30
};

// ---------------------------------------------------------------- BSDF
struct Bsdf { int type; Color reflectance, rho_0, k, g, rho_c; uint32_t flags; Color transmittance; };
// bsdf.h:38-124
enum : uint32_t { F_Null = 0x1, F_DiffuseReflection = 0x2, F_DiffuseTransmission = 0x4, F_GlossyReflection = 0x8,
                  F_GlossyTransmission = 0x10, F_DeltaReflection = 0x20, F_DeltaTransmission = 0x40,
                  F_FrontSide = 0x8000, F_BackSide = 0x10000,
                  F_Smooth = 0x2 | 0x4 | 0x8 | 0x10, F_Delta = 0x1 | 0x20 | 0x40 };

// ---------------------------------------------------------------- Shapes
struct Shape {
    int type;
    Xf to_world, to_object;
    int bsdf, interior, exterior, emitter;
    BBox bbox;
    // rectangle (rectangle.cpp:66-74)
    Frame frame;
    float inv_surface_area;
    float surface_area = 0.f;                    // disk (disk.cpp:108-111)
    // sphere (sphere.cpp:80-105)
    V3 center; float radius; bool flip_normals;
    // mesh (mesh.cpp, cube.cpp:54-112): world-space vertex data
    std::vector<float> positions, normals, texcoords;
    std::vector<uint32_t> faces;
    int prim_count;
    // mesh area distribution (mesh.cpp:285-312, distr_1d.h:49-83): unnormalised pmf / cdf over the faces, first / last non-empty face
    DiscreteDistribution area_distr;             // Mesh::m_area_distr (mesh.cpp:285-312)
    bool is_medium_transition() const { return interior >= 0 || exterior >= 0; }   // shape.h:341
};

// cube.cpp:43-67
static const float CUBE_VERTICES[24][3] = {
    { 1, -1, -1 }, { 1, -1, 1 }, { -1, -1, 1 }, { -1, -1, -1 }, { 1, 1, -1 }, { -1, 1, -1 }, { -1, 1, 1 }, { 1, 1, 1 },
    { 1, -1, -1 }, { 1, 1, -1 }, { 1, 1, 1 }, { 1, -1, 1 }, { 1, -1, 1 }, { 1, 1, 1 }, { -1, 1, 1 }, { -1, -1, 1 },
    { -1, -1, 1 }, { -1, 1, 1 }, { -1, 1, -1 }, { -1, -1, -1 }, { 1, 1, -1 }, { 1, -1, -1 }, { -1, -1, -1 }, { -1, 1, -1 } };
static const float CUBE_NORMALS[24][3] = {
    { 0, -1, 0 }, { 0, -1, 0 }, { 0, -1, 0 }, { 0, -1, 0 }, { 0, 1, 0 }, { 0, 1, 0 }, { 0, 1, 0 }, { 0, 1, 0 },
    { 1, 0, 0 }, { 1, 0, 0 }, { 1, 0, 0 }, { 1, 0, 0 }, { 0, 0, 1 }, { 0, 0, 1 }, { 0, 0, 1 }, { 0, 0, 1 },
    { -1, 0, 0 }, { -1, 0, 0 }, { -1, 0, 0 }, { -1, 0, 0 }, { 0, 0, -1 }, { 0, 0, -1 }, { 0, 0, -1 }, { 0, 0, -1 } };
static const float CUBE_TEXCOORDS[24][2] = {
    { 0, 1 }, { 1, 1 }, { 1, 0 }, { 0, 0 }, { 0, 1 }, { 1, 1 }, { 1, 0 }, { 0, 0 }, { 0, 1 }, { 1, 1 }, { 1, 0 }, { 0, 0 },
    { 0, 1 }, { 1, 1 }, { 1, 0 }, { 0, 0 }, { 0, 1 }, { 1, 1 }, { 1, 0 }, { 0, 0 }, { 0, 1 }, { 1, 1 }, { 1, 0 }, { 0, 0 } };
static const uint32_t CUBE_TRIANGLES[12][3] = {
    { 0, 1, 2 }, { 3, 0, 2 }, { 4, 5, 6 }, { 7, 4, 6 }, { 8, 9, 10 }, { 11, 8, 10 },
    { 12, 13, 14 }, { 15, 12, 14 }, { 16, 17, 18 }, { 19, 16, 18 }, { 20, 21, 22 }, { 23, 20, 22 } };

static inline Shape make_shape(const mts_shape &d) {
    Shape s = {};
    s.type = d.type;
    s.to_world = xf_from_abi(d.to_world);
    s.bsdf = d.bsdf; s.interior = d.interior_medium; s.exterior = d.exterior_medium; s.emitter = d.emitter;
    s.flip_normals = d.flip_normals != 0;
    s.bbox = bbox_empty();
    if (d.type == MTS_SHAPE_RECTANGLE) {
        if (d.flip_normals) s.to_world = xf_mul(s.to_world, xf_scale(v3(1.f, 1.f, -1.f)));   // rectangle.cpp:61-62
        s.to_object = xf_inverse(s.to_world);
        V3 dp_du = xf_vector(s.to_world, v3(2.f, 0.f, 0.f)), dp_dv = xf_vector(s.to_world, v3(0.f, 2.f, 0.f));
        V3 n = normalize(xf_normal(s.to_world, v3(0.f, 0.f, 1.f)));
        s.frame.s = dp_du; s.frame.t = dp_dv; s.frame.n = n;                                 // rectangle.cpp:68-72
        s.inv_surface_area = pm_rcp(norm(cross(s.frame.s, s.frame.t)));                     // rectangle.cpp:73,86
        bbox_expand(s.bbox, xf_point_affine(s.to_world, v3(-1.f, -1.f, 0.f)));               // rectangle.cpp:77-83
        bbox_expand(s.bbox, xf_point_affine(s.to_world, v3(1.f, -1.f, 0.f)));
        bbox_expand(s.bbox, xf_point_affine(s.to_world, v3(1.f, 1.f, 0.f)));
        bbox_expand(s.bbox, xf_point_affine(s.to_world, v3(-1.f, 1.f, 0.f)));
        s.prim_count = 1;
    } else if (d.type == MTS_SHAPE_DISK) {                                                   // disk.cpp:74-111
        if (d.flip_normals) s.to_world = xf_mul(s.to_world, xf_scale(v3(1.f, 1.f, -1.f)));
        s.to_object = xf_inverse(s.to_world);
        V3 dp_du = xf_vector(s.to_world, v3(1.f, 0.f, 0.f)), dp_dv = xf_vector(s.to_world, v3(0.f, 1.f, 0.f));
        float du = norm(dp_du), dv = norm(dp_dv);
        V3 n = normalize(xf_normal(s.to_world, v3(0.f, 0.f, 1.f)));
        s.frame.s = dp_du / du; s.frame.t = dp_dv / dv; s.frame.n = n;
        float dts = dot(s.frame.t * dv, s.frame.s);
        float h = pm_sqrt(dv * dv - dts * dts);
        s.surface_area = Pi * du * h;
        s.inv_surface_area = 1.f / s.surface_area;
        bbox_expand(s.bbox, xf_point_affine(s.to_world, v3(-1.f, -1.f, 0.f)));
        bbox_expand(s.bbox, xf_point_affine(s.to_world, v3(-1.f, 1.f, 0.f)));
        bbox_expand(s.bbox, xf_point_affine(s.to_world, v3(1.f, -1.f, 0.f)));
        bbox_expand(s.bbox, xf_point_affine(s.to_world, v3(1.f, 1.f, 0.f)));
        s.prim_count = 1;
    } else if (d.type == MTS_SHAPE_CUBE || d.type == MTS_SHAPE_MESH) {
        s.to_object = xf_inverse(s.to_world);
        int nv, nf; const float *pos, *nor, *uv; const uint32_t *fc;
        if (d.type == MTS_SHAPE_CUBE) { nv = 24; nf = 12; pos = &CUBE_VERTICES[0][0]; nor = &CUBE_NORMALS[0][0]; uv = &CUBE_TEXCOORDS[0][0]; fc = &CUBE_TRIANGLES[0][0]; }
        else { nv = d.vertex_count; nf = d.face_count; pos = d.vertex_positions; nor = d.vertex_normals; uv = d.vertex_texcoords; fc = d.faces;
               if (!pos || !fc || nv <= 0 || nf <= 0) throw std::runtime_error("mesh: missing vertex / face data"); }
        s.positions.resize(3 * nv);
        if (nor) s.normals.resize(3 * nv);
        if (uv) s.texcoords.assign(uv, uv + 2 * nv);
        for (int i = 0; i < nv; ++i) {                                                         // cube.cpp:88-104
            V3 p = xf_point_affine(s.to_world, v3(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]));
            bbox_expand(s.bbox, p);
            s.positions[3 * i] = p.x; s.positions[3 * i + 1] = p.y; s.positions[3 * i + 2] = p.z;
            if (nor) {
                V3 n = normalize(xf_normal(s.to_world, v3(nor[3 * i], nor[3 * i + 1], nor[3 * i + 2])));
                s.normals[3 * i] = n.x; s.normals[3 * i + 1] = n.y; s.normals[3 * i + 2] = n.z;
            }
        }
        s.faces.assign(fc, fc + 3 * nf);
        for (int i = 0; i < 3 * nf; ++i) if (s.faces[i] >= (uint32_t) nv) throw std::runtime_error("mesh: face index out of range");
        s.prim_count = nf;
        // Mesh::build_pmf (mesh.cpp:285-312) + DiscreteDistribution::update (distr_1d.h:49-83): face areas (mesh.h:108-116),
        // running sum in double precision
        s.area_distr.pmf.resize(nf);
        for (int i = 0; i < nf; ++i) {
            const float *P = s.positions.data(); const uint32_t *f = &s.faces[3 * i];
            V3 p0 = v3(P[3 * f[0]], P[3 * f[0] + 1], P[3 * f[0] + 2]), p1 = v3(P[3 * f[1]], P[3 * f[1] + 1], P[3 * f[1] + 2]),
               p2 = v3(P[3 * f[2]], P[3 * f[2] + 1], P[3 * f[2] + 2]);
            s.area_distr.pmf[i] = 0.5f * norm(cross(p1 - p0, p2 - p0));
        }
        bool has_area = false;
        for (int i = 0; i < nf; ++i) has_area = has_area || s.area_distr.pmf[i] > 0.f;
        double sum = 0.0;
        if (has_area) { discrete_update(s.area_distr); for (int i = 0; i < nf; ++i) sum += (double) s.area_distr.pmf[i]; }
        s.surface_area = (float) sum;                                                          // mesh.cpp:346-350
        s.inv_surface_area = (float) (1.0 / sum);
    } else if (d.type == MTS_SHAPE_SPHERE) {
        // sphere.cpp:80-105: to_world * translate(center) * scale(radius); must be a uniform scale without shear.
        Xf tw = xf_mul(s.to_world, xf_mul(xf_translate(v3(d.center[0], d.center[1], d.center[2])), xf_scale(v3(d.radius, d.radius, d.radius))));
        V3 c0 = v3(tw.m[0], tw.m[4], tw.m[8]), c1 = v3(tw.m[1], tw.m[5], tw.m[9]), c2 = v3(tw.m[2], tw.m[6], tw.m[10]);
        float r0 = norm(c0), r1 = norm(c1), r2 = norm(c2);
        if (pm_abs(r0 - r1) > 1e-6f * r0 || pm_abs(r0 - r2) > 1e-6f * r0 ||
            pm_abs(dot(c0, c1)) > 1e-6f * r0 * r0 || pm_abs(dot(c0, c2)) > 1e-6f * r0 * r0 || pm_abs(dot(c1, c2)) > 1e-6f * r0 * r0)
            throw std::runtime_error("'to_world' transform shouldn't contain any scale or shear along the sphere axes");
        s.center = xf_translation(tw);
        s.radius = r0;
        // Reconstructed transform: rotation Q = columns / radius, uniform scale, translation
        Xf rec = xf_identity();
        V3 q0 = c0 / r0, q1 = c1 / r0, q2 = c2 / r0;
        float R[9] = { q0.x, q1.x, q2.x, q0.y, q1.y, q2.y, q0.z, q1.z, q2.z };
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) {
            rec.m[r * 4 + c] = R[r * 3 + c] * s.radius;
            rec.it[r * 4 + c] = R[r * 3 + c] / s.radius;
        }
        rec.m[3] = s.center.x; rec.m[7] = s.center.y; rec.m[11] = s.center.z;
        // inverse transpose translation row: -(R/r)^T c
        Xf tmp = rec;
        float invm[16] = {};
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) invm[r * 4 + c] = R[c * 3 + r] / s.radius;
        for (int r = 0; r < 3; ++r) invm[r * 4 + 3] = -(invm[r * 4] * s.center.x + invm[r * 4 + 1] * s.center.y + invm[r * 4 + 2] * s.center.z);
        invm[15] = 1.f;
        mat_transpose(invm, tmp.it);
        s.to_world = tmp;
        s.to_object = xf_inverse(s.to_world);
        s.inv_surface_area = pm_rcp(4.f * Pi * s.radius * s.radius);
        s.bbox.min = s.center - v3(s.radius, s.radius, s.radius);
        s.bbox.max = s.center + v3(s.radius, s.radius, s.radius);
        s.prim_count = 1;
    } else throw std::runtime_error("unknown shape type");
    return s;
}

// ---------------------------------------------------------------- Emitter
struct Emitter {
    int type; Xf to_world; Color radiance; int shape;
    V3 bsphere_center; float bsphere_radius;    // directional.cpp:68-73, constant.cpp:35-39
    bool is_environment() const { return type == MTS_EMITTER_CONSTANT; }
};

// ---------------------------------------------------------------- Sensor / film / filter
struct RFilter {
    int type; float radius, stddev, alpha, bias;
    float values[32]; float scale_factor; int border_size;
    float eval(float x) const {
        if (type == MTS_RFILTER_BOX) return pm_abs(x) <= radius ? 1.f : 0.f;              // box.cpp:35-37
        return pm_max(0.f, pm_exp(alpha * (x * x)) - bias);                                // gaussian.cpp:45-47
    }
    float eval_discretized(float x) const {                                                // rfilter.h:62-65
        int index = std::min((int) pm_abs(x * scale_factor), 31);
        return values[index];
    }
};
static inline RFilter make_rfilter(int type, float radius, float stddev) {
    RFilter f = {};
    f.type = type;
    if (type == MTS_RFILTER_BOX) f.radius = radius + RayEpsilon;                           // box.cpp:31
    else { f.stddev = stddev; f.radius = 4 * stddev; f.alpha = -1.f / (2.f * stddev * stddev); f.bias = pm_exp(f.alpha * (f.radius * f.radius)); }  // gaussian.cpp:33-42
    for (int i = 0; i < 31; ++i) f.values[i] = f.eval((f.radius * i) / 31);               // rfilter.cpp:9-20, MTS_FILTER_RESOLUTION = 31
    f.values[31] = 0;
    f.scale_factor = 31 / f.radius;
    f.border_size = (int) std::ceil(f.radius - .5f - 2.f * RayEpsilon);
    return f;
}

struct Sensor {
    int type; Xf to_world;
    // perspective (perspective.cpp:101-124)
    Xf camera_to_sample, sample_to_camera;
    float near_clip, far_clip; P2 principal_point_offset;
    // distant (distant.cpp:225-297)
    int direction_type;      // 0 single, 1 sample width, 2 sample all
    bool flip_directions;
    int target_type; V3 target_point; Shape target_shape;
    V3 bsphere_center; float bsphere_radius;
    bool needs_aperture_sample;
    float shutter_open_time = 0.f;
    bool origin_is_shape = false; Shape origin_shape;      // distant.cpp:280-289, distantflux.cpp:172-184
    int medium;
    // film
    int width, height, crop_x, crop_y, crop_w, crop_h;
    RFilter rfilter;
    int sample_count; uint64_t seed; bool wavefront = false;      // wavefront: one TEA-seeded stream per (pixel, sample), the gpu_* variants' seeding
    std::vector<float> multi; int multi_count = 0;   // mradiancemeter / mdistant: m_transforms (mradiancemeter.cpp:95-113, mdistant.cpp:160-175)
    int srf = -1;                                    // spectral variant: index of the "srf" spectrum (perspective.cpp:113-116, radiancemeter.cpp:62-66)
};

// Transform::perspective, transform.h:203-220
static inline Xf xf_perspective(float fov, float near_, float far_) {
    float recip = 1.f / (far_ - near_);
    float tan_ = std::tan(fov * .5f * (Pi / 180.f)), cot = 1.f / tan_;
    Xf x = {};
    x.m[0] = cot; x.m[5] = cot; x.m[10] = far_ * recip; x.m[11] = -near_ * far_ * recip; x.m[14] = 1.f;
    float inv[16] = {};
    inv[0] = tan_; inv[5] = tan_; inv[15] = 1.f / near_; inv[11] = 1.f; inv[14] = (near_ - far_) / (far_ * near_);
    mat_transpose(inv, x.it);
    return x;
}
// sensor.h:196-231
static inline Xf perspective_projection(int fw, int fh, int cw, int ch, int cx, int cy, float fov_x, float near_clip, float far_clip) {
    float fsx = (float) fw, fsy = (float) fh;
    float rel_size_x = (float) cw / fsx, rel_size_y = (float) ch / fsy, rel_off_x = (float) cx / fsx, rel_off_y = (float) cy / fsy;
    float aspect = fsx / fsy;
    return xf_mul(xf_scale(v3(1.f / rel_size_x, 1.f / rel_size_y, 1.f)),
           xf_mul(xf_translate(v3(-rel_off_x, -rel_off_y, 0.f)),
           xf_mul(xf_scale(v3(-0.5f, -0.5f * aspect, 1.f)),
           xf_mul(xf_translate(v3(-1.f, -1.f / aspect, 0.f)),
                  xf_perspective(fov_x, near_clip, far_clip)))));
}

// ---------------------------------------------------------------- Scene
struct Prim { int shape; int index; };

struct Scene {
    std::vector<Volume> volumes;
    std::vector<Phase> phases;
    std::vector<Medium> media;
    std::vector<Bsdf> bsdfs;
    std::vector<Shape> shapes;
    std::vector<Emitter> emitters;
    int environment;
    Sensor sensor;
    mts_integrator integrator;
    BBox bbox;
    std::vector<Prim> prims;
#if MTS_SPEC_N != 3
    std::deque<SpectrumRec> spectra;                                   // stable addresses: the colour parameters point at them
    std::vector<float> bin_lo, bin_hi;                                  // nbins: wavelength, tolerance; bins: interval (integrator.bin_mode / bin_count)
#endif
    Bsdf default_bsdf, default_emitter_bsdf;
    const Bsdf &bsdf_of(const Shape &s) const {
        if (s.bsdf >= 0) return bsdfs[s.bsdf];
        return s.emitter >= 0 ? default_emitter_bsdf : default_bsdf;    // shape.cpp:74-80
    }
};

static inline uint32_t bsdf_flags(int type) {
    if (type == MTS_BSDF_DIFFUSE) return F_DiffuseReflection | F_FrontSide;              // diffuse.cpp:55
    if (type == MTS_BSDF_NULL) return F_Null | F_FrontSide | F_BackSide;                 // null.cpp:26
    if (type == MTS_BSDF_BILAMBERTIAN) return F_DiffuseReflection | F_DiffuseTransmission | F_FrontSide | F_BackSide;   // bilambertian.cpp:55-60
    return F_GlossyReflection | F_FrontSide;                                             // rpv.cpp:66
}

static inline void check_index(int i, size_t n, const char *what, bool allow_none) {
    if (i < 0 && allow_none) return;
    if (i < 0 || (size_t) i >= n) throw std::runtime_error(std::string("index out of range: ") + what);
}

static inline Scene *make_scene(const mts_scene_desc *d) {
    if (!d || d->abi_version != MTS_ABI_VERSION) throw std::runtime_error("scene description: ABI version mismatch");
    Scene *sc = new Scene();
    try {
#if MTS_SPEC_N == 3
        if (d->integrator.spectral) throw std::runtime_error("liboracle.so is the rgb / mono build; spectral scenes need liboracle_spectral.so");
#else
        if (!d->integrator.spectral) throw std::runtime_error("liboracle_spectral.so renders scenes of the spectral variant only");
        for (int i = 0; i < d->spectrum_count; ++i) {                      // uniform.cpp:34-52, regular.cpp:27-58 + distr_1d.h:318-345
            const mts_spectrum &sp = d->spectra[i];
            SpectrumRec r; r.type = sp.type; r.value = sp.value; r.lambda_min = sp.lambda_min; r.lambda_max = sp.lambda_max; r.inv_interval_size = 0.f;
            if (sp.type == MTS_SPECTRUM_UNIFORM) {
                r.lambda_min = std::max(sp.lambda_min, 280.f); r.lambda_max = std::min(sp.lambda_max, 2400.f);      // MTS_WAVELENGTH_MIN / MAX
                if (!(r.lambda_min < r.lambda_max)) throw std::runtime_error("UniformSpectrum: 'lambda_min' must be less than 'lambda_max'");
            } else if (sp.type == MTS_SPECTRUM_REGULAR) {
                if (!(sp.lambda_min < sp.lambda_max)) throw std::runtime_error("ContinuousDistribution: invalid range!");
                if (sp.count < 2 || !sp.values) throw std::runtime_error("ContinuousDistribution: needs at least two entries!");
                bool mass = false;
                for (int k = 0; k < sp.count; ++k) { if (sp.values[k] < 0.f) throw std::runtime_error("ContinuousDistribution: entries must be non-negative!"); mass = mass || sp.values[k] > 0.f; }
                if (!mass) throw std::runtime_error("ContinuousDistribution: no probability mass found!");
                r.values.assign(sp.values, sp.values + sp.count);
                r.inv_interval_size = (float) (1. / ((double(sp.lambda_max) - double(sp.lambda_min)) / (sp.count - 1)));
            } else if (sp.type == MTS_SPECTRUM_IRREGULAR) {                   // irregular.cpp:33-63 + distr_1d.h:560-600
                if (sp.count < 2 || !sp.values || !sp.wavelengths) throw std::runtime_error("IrregularContinuousDistribution: needs at least two entries!");
                bool mass = false;
                for (int k = 0; k < sp.count; ++k) {
                    if (sp.values[k] < 0.f) throw std::runtime_error("IrregularContinuousDistribution: entries must be non-negative!");
                    if (k > 0 && !(sp.wavelengths[k] > sp.wavelengths[k - 1])) throw std::runtime_error("IrregularContinuousDistribution: node positions must be strictly increasing!");
                    mass = mass || sp.values[k] > 0.f;
                }
                if (!mass) throw std::runtime_error("IrregularContinuousDistribution: no probability mass found!");
                r.values.assign(sp.values, sp.values + sp.count); r.wavelengths.assign(sp.wavelengths, sp.wavelengths + sp.count);
                r.lambda_min = sp.wavelengths[0]; r.lambda_max = sp.wavelengths[sp.count - 1];
            } else if (sp.type == MTS_SPECTRUM_DISCRETE) {                    // discrete.cpp:45-100 + DiscreteDistribution (distr_1d.h:49-83)
                if (sp.count < 1 || !sp.values || !sp.wavelengths || !sp.pmf) throw std::runtime_error("DiscreteDistribution: empty distribution!");
                r.values.assign(sp.values, sp.values + sp.count); r.wavelengths.assign(sp.wavelengths, sp.wavelengths + sp.count);
                r.cdf.resize(sp.count); r.valid_x = r.valid_y = (uint32_t) -1;
                double sum = 0.0;
                for (int k = 0; k < sp.count; ++k) {
                    double value = (double) sp.pmf[k];
                    sum += value; r.cdf[k] = (float) sum;
                    if (value < 0.0) throw std::runtime_error("DiscreteDistribution: entries must be non-negative!");
                    else if (value > 0.0) { if (r.valid_x == (uint32_t) -1) r.valid_x = (uint32_t) k; r.valid_y = (uint32_t) k; }
                }
                if (r.valid_x == (uint32_t) -1) throw std::runtime_error("DiscreteDistribution: no probability mass found!");
                r.cdf_sum = (float) sum;
            } else throw std::runtime_error("unknown spectrum type");
            sc->spectra.push_back(r);
        }
        sc->spectra.push_back(SpectrumRec{ MTS_SPECTRUM_UNIFORM, .5f, 280.f, 2400.f, {}, 0.f });      // default BSDF reflectance (shape.cpp:74-80)
        sc->spectra.push_back(SpectrumRec{ MTS_SPECTRUM_UNIFORM, 0.f, 280.f, 2400.f, {}, 0.f });
#endif
        auto color3 = [&](const float *rgb, int sp, bool used) -> Color {
#if MTS_SPEC_N == 3
            (void) sp; (void) used; return v3(rgb[0], rgb[1], rgb[2]);
#else
            (void) rgb;
            if (!used) return Color{ &sc->spectra.back() };
            if (sp < 0 || sp >= d->spectrum_count) throw std::runtime_error("spectral variant: missing spectrum for a colour parameter");
            return Color{ &sc->spectra[(size_t) sp] };
#endif
        };
        for (int i = 0; i < d->volume_count; ++i) {
            sc->volumes.push_back(make_volume(d->volumes[i]));
#if MTS_SPEC_N != 3
            if (d->volumes[i].type == MTS_VOLUME_CONST) sc->volumes.back().value = color3(nullptr, d->volumes[i].value_spectrum, true);
#endif
        }
        for (int i = 0; i < d->phase_count; ++i) {
            sc->phases.push_back(make_phase(d->phases[i]));
            if (d->phases[i].type == MTS_PHASE_BLEND) {
                check_index(d->phases[i].child[0], d->phase_count, "blendphase child", false);
                check_index(d->phases[i].child[1], d->phase_count, "blendphase child", false);
                check_index(d->phases[i].weight_volume, d->volume_count, "blendphase weight", false);
                // nested blendphase plugins recurse (oracle.cpp, phase_eval / phase_sample): children first, so that no cycle can form
                if (d->phases[i].child[0] >= i || d->phases[i].child[1] >= i) throw std::runtime_error("blendphase: a nested phase function must precede the blendphase that holds it");
            }
        }
        for (int i = 0; i < d->medium_count; ++i) {
            const mts_medium &m = d->media[i];
            check_index(m.sigma_t_volume, d->volume_count, "medium sigma_t", false);
            check_index(m.albedo_volume, d->volume_count, "medium albedo", false);
            check_index(m.phase, d->phase_count, "medium phase", false);
            Medium me = {};
            me.type = m.type; me.sigma_t = m.sigma_t_volume; me.albedo = m.albedo_volume; me.phase = m.phase;
            me.scale = m.scale; me.sample_emitters = m.sample_emitters != 0;
            me.has_spectral_extinction = m.has_spectral_extinction != 0;
            me.is_homogeneous = m.type == MTS_MEDIUM_HOMOGENEOUS;
            if (!me.is_homogeneous) {                                                      // heterogeneous.cpp:29-30
                const Volume &st = sc->volumes[m.sigma_t_volume];
                if (!st.has_max) throw std::runtime_error("max() not implemented (constvolume sigma_t in heterogeneous medium)");   // constant3d.cpp:37
                me.max_density = me.scale * st.max;
                me.aabb = st.bbox;
            }
            sc->media.push_back(me);
        }
        for (int i = 0; i < d->bsdf_count; ++i) {
            const mts_bsdf &b = d->bsdfs[i];
            Bsdf bs = {};
            bs.type = b.type;
            static const bool used[4][6] = { { 1, 0, 0, 0, 0, 0 } /* diffuse */, { 0, 0, 0, 0, 0, 0 } /* null */, { 0, 1, 1, 1, 1, 0 } /* rpv */, { 1, 0, 0, 0, 0, 1 } /* bilambertian */ };
            if (b.type < MTS_BSDF_DIFFUSE || b.type > MTS_BSDF_BILAMBERTIAN) throw std::runtime_error("unknown BSDF type");
            bs.reflectance = color3(b.reflectance, b.spectrum[0], used[b.type][0]);
            bs.rho_0 = color3(b.rho_0, b.spectrum[1], used[b.type][1]); bs.k = color3(b.k, b.spectrum[2], used[b.type][2]);
            bs.g = color3(b.g, b.spectrum[3], used[b.type][3]); bs.rho_c = color3(b.rho_c, b.spectrum[4], used[b.type][4]);
            bs.flags = bsdf_flags(b.type);
            bs.transmittance = color3(b.transmittance, b.spectrum[5], used[b.type][5]);
            sc->bsdfs.push_back(bs);
        }
#if MTS_SPEC_N == 3
        sc->default_bsdf = Bsdf{ MTS_BSDF_DIFFUSE, v3(.5f, .5f, .5f), {}, {}, {}, {}, bsdf_flags(MTS_BSDF_DIFFUSE) };
        sc->default_emitter_bsdf = Bsdf{ MTS_BSDF_DIFFUSE, v3(0.f, 0.f, 0.f), {}, {}, {}, {}, bsdf_flags(MTS_BSDF_DIFFUSE) };
#else
        {
            const SpectrumRec *half = &sc->spectra[sc->spectra.size() - 2], *zero = &sc->spectra[sc->spectra.size() - 1];
            sc->default_bsdf = Bsdf{ MTS_BSDF_DIFFUSE, Color{ half }, Color{ half }, Color{ half }, Color{ half }, Color{ half }, bsdf_flags(MTS_BSDF_DIFFUSE), Color{ half } };
            sc->default_emitter_bsdf = Bsdf{ MTS_BSDF_DIFFUSE, Color{ zero }, Color{ zero }, Color{ zero }, Color{ zero }, Color{ zero }, bsdf_flags(MTS_BSDF_DIFFUSE), Color{ zero } };
        }
#endif
        sc->bbox = bbox_empty();
        for (int i = 0; i < d->shape_count; ++i) {
            const mts_shape &s = d->shapes[i];
            check_index(s.bsdf, d->bsdf_count, "shape bsdf", true);
            check_index(s.interior_medium, d->medium_count, "shape interior", true);
            check_index(s.exterior_medium, d->medium_count, "shape exterior", true);
            check_index(s.emitter, d->emitter_count, "shape emitter", true);
            sc->shapes.push_back(make_shape(s));
            bbox_expand(sc->bbox, sc->shapes.back().bbox);                                 // scene.cpp:38
            for (int k = 0; k < sc->shapes.back().prim_count; ++k) sc->prims.push_back(Prim{ i, k });
        }
        sc->environment = -1;
        for (int i = 0; i < d->emitter_count; ++i) {
            const mts_emitter &e = d->emitters[i];
            Emitter em = {};
            em.type = e.type; em.to_world = xf_from_abi(e.to_world);
            em.radiance = color3(e.radiance, e.radiance_spectrum, true); em.shape = e.shape;
            if (e.type == MTS_EMITTER_AREA) {
                check_index(e.shape, d->shape_count, "area emitter shape", false);
                if ((sc->shapes[e.shape].type == MTS_SHAPE_CUBE || sc->shapes[e.shape].type == MTS_SHAPE_MESH) && sc->shapes[e.shape].area_distr.valid_x == (uint32_t) -1)
                    throw std::runtime_error("DiscreteDistribution: no probability mass found!");   // distr_1d.h:78-79
            }
            if (em.is_environment()) {
                if (sc->environment >= 0) throw std::runtime_error("Only one environment emitter can be specified per scene.");   // scene.cpp:48-50
                sc->environment = i;
            }
            // set_scene: directional.cpp:68-73, constant.cpp:35-39; bbox.h:329-332
            V3 c = (sc->bbox.max + sc->bbox.min) * .5f;
            em.bsphere_center = c;
            em.bsphere_radius = pm_max(RayEpsilon, norm(c - sc->bbox.max) * (1.f + RayEpsilon));
            sc->emitters.push_back(em);
        }
        // Sensor
        const mts_sensor &s = d->sensor;
        Sensor &se = sc->sensor;
        se = Sensor();
        se.type = s.type; se.to_world = xf_from_abi(s.to_world);
        se.width = s.film_width; se.height = s.film_height;
        se.crop_x = s.crop_offset[0]; se.crop_y = s.crop_offset[1]; se.crop_w = s.crop_size[0]; se.crop_h = s.crop_size[1];
        if (se.width <= 0 || se.height <= 0 || se.crop_w <= 0 || se.crop_h <= 0) throw std::runtime_error("film: invalid size");
        se.rfilter = make_rfilter(s.rfilter_type, s.rfilter_radius, s.rfilter_stddev);
        se.sample_count = s.sample_count; se.seed = s.sampler_seed; se.medium = s.medium; se.shutter_open_time = s.shutter_open_time;
        se.wavefront = s.sampler_wavefront != 0;
        if (se.wavefront && d->integrator.samples_per_pass >= 0 && d->integrator.samples_per_pass != s.sample_count)
            throw std::runtime_error("wavefront streams: samples_per_pass must cover the whole sample_count (one pass)");
        if ((s.type == MTS_SENSOR_DISTANT || s.type == MTS_SENSOR_DISTANTFLUX) && s.distant_origin_type != 0) {
            se.origin_is_shape = true;
            se.origin_shape = make_shape(s.distant_origin_shape);
            if (se.origin_shape.type != MTS_SHAPE_RECTANGLE && se.origin_shape.type != MTS_SHAPE_SPHERE && se.origin_shape.type != MTS_SHAPE_DISK)
                throw std::runtime_error("distant sensor: the ray origin shape must be a rectangle, a disk or a sphere in this backend");
        }
        check_index(s.medium, d->medium_count, "sensor medium", true);
        if (s.type == MTS_SENSOR_PERSPECTIVE) {
            se.near_clip = s.near_clip; se.far_clip = s.far_clip;
            se.camera_to_sample = perspective_projection(se.width, se.height, se.crop_w, se.crop_h, se.crop_x, se.crop_y, s.fov_x, s.near_clip, s.far_clip);
            se.sample_to_camera = xf_inverse(se.camera_to_sample);
            // perspective.cpp:101-106
            se.principal_point_offset.x = s.principal_point_offset[0] * ((float) se.width / (float) se.crop_w);
            se.principal_point_offset.y = s.principal_point_offset[1] * ((float) se.height / (float) se.crop_h);
            se.needs_aperture_sample = false;                                              // perspective.cpp:122
        } else if (s.type == MTS_SENSOR_MRADIANCEMETER || s.type == MTS_SENSOR_MDISTANT) {
            if (s.multi_count <= 0 || !s.multi_transforms) throw std::runtime_error("multi-sensor: no sub-sensors given");
            if (se.width != s.multi_count || se.height != 1) throw std::runtime_error("Film size must be [sensor_count, 1].");   // mradiancemeter.cpp:115-118
            se.multi.assign(s.multi_transforms, s.multi_transforms + 16 * (size_t) s.multi_count);
            se.multi_count = s.multi_count;
            se.needs_aperture_sample = s.type == MTS_SENSOR_MDISTANT;                      // m_needs_sample_3: mradiancemeter.cpp:126, mdistant.cpp:202
            se.target_type = MTS_DISTANT_TARGET_NONE;
            if (s.type == MTS_SENSOR_MDISTANT) {
                se.target_type = s.distant_target_type;
                se.target_point = v3(s.distant_target_point[0], s.distant_target_point[1], s.distant_target_point[2]);
                if (se.target_type == MTS_DISTANT_TARGET_SHAPE) {
                    se.target_shape = make_shape(s.distant_target_shape);
                    if (se.target_shape.type != MTS_SHAPE_RECTANGLE && se.target_shape.type != MTS_SHAPE_SPHERE && se.target_shape.type != MTS_SHAPE_DISK)
                        throw std::runtime_error("mdistant target shape must be a rectangle or a sphere in this backend");
                }
                V3 c = (sc->bbox.max + sc->bbox.min) * .5f;                                // mdistant.cpp:205-210
                se.bsphere_center = c;
                se.bsphere_radius = pm_max(RayEpsilon, norm(c - sc->bbox.max) * (1.f + RayEpsilon));
            }
        } else {
            // distant.cpp:228-238
            se.direction_type = (se.width == 1 && se.height == 1) ? 0 : (se.height == 1 ? 1 : 2);
            se.flip_directions = s.distant_flip_directions != 0;
            se.target_type = s.distant_target_type;
            se.target_point = v3(s.distant_target_point[0], s.distant_target_point[1], s.distant_target_point[2]);
            if (se.target_type == MTS_DISTANT_TARGET_SHAPE) {
                se.target_shape = make_shape(s.distant_target_shape);
                if (se.target_shape.type != MTS_SHAPE_RECTANGLE && se.target_shape.type != MTS_SHAPE_SPHERE && se.target_shape.type != MTS_SHAPE_DISK)
                    throw std::runtime_error("distant ray_target shape must be a rectangle or a sphere in this backend");
            }
            V3 c = (sc->bbox.max + sc->bbox.min) * .5f;                                    // distant.cpp:292-297
            se.bsphere_center = c;
            se.bsphere_radius = pm_max(RayEpsilon, norm(c - sc->bbox.max) * (1.f + RayEpsilon));
            se.needs_aperture_sample = true;                                               // endpoint.h:244
        }
        sc->integrator = d->integrator;
        sc->integrator.bin_lo = sc->integrator.bin_hi = nullptr;                             // copied below: the caller's arrays may go away
        if (d->integrator.bin_mode != 0) {
#if MTS_SPEC_N == 3
            throw std::runtime_error("This integrator can only be used with a spectral variant!");   // nbins.cpp:57-58, bins.cpp:25-26
#else
            if (d->integrator.bin_mode != 1 && d->integrator.bin_mode != 2) throw std::runtime_error("unknown bin mode");
            if (d->integrator.bin_count < 0 || (d->integrator.bin_count > 0 && (!d->integrator.bin_lo || !d->integrator.bin_hi))) throw std::runtime_error("bins: missing bounds");
            sc->bin_lo.assign(d->integrator.bin_lo, d->integrator.bin_lo + d->integrator.bin_count);
            sc->bin_hi.assign(d->integrator.bin_hi, d->integrator.bin_hi + d->integrator.bin_count);
            if (d->integrator.bin_mode == 2)                                                 // bins.cpp:79-84: a uniform spectrum per bin (bounds clamped, uniform.cpp:41-46)
                for (int i = 0; i < d->integrator.bin_count; ++i) {
                    sc->bin_lo[i] = std::max(sc->bin_lo[i], 280.f); sc->bin_hi[i] = std::min(sc->bin_hi[i], 2400.f);
                    if (!(sc->bin_lo[i] < sc->bin_hi[i])) throw std::runtime_error("UniformSpectrum: 'lambda_min' must be less than 'lambda_max'");
                }
#endif
        } else sc->integrator.bin_count = 0;
#if MTS_SPEC_N != 3
        sc->sensor.srf = -1;
        if (d->sensor.srf != 0) {
            if (d->sensor.srf < 0 || d->sensor.srf > d->spectrum_count) throw std::runtime_error("index out of range: sensor srf");
            if (d->sensor.type != MTS_SENSOR_PERSPECTIVE && !(d->sensor.type == MTS_SENSOR_MRADIANCEMETER && d->sensor.multi_count == 1))
                throw std::runtime_error("srf: only perspective and radiancemeter sensors sample their wavelengths from a response function");
            const SpectrumRec &r = sc->spectra[(size_t) d->sensor.srf - 1];
            if (r.type != MTS_SPECTRUM_UNIFORM && r.type != MTS_SPECTRUM_DISCRETE) throw std::runtime_error("srf: sample_spectrum is restated for uniform and discrete spectra");
            sc->sensor.srf = d->sensor.srf - 1;
        }
#endif
        if (sc->integrator.rr_depth <= 0) throw std::runtime_error("\"rr_depth\" must be set to a value greater than zero!");   // integrator.cpp:306-307
        if (sc->integrator.max_depth < 0 && sc->integrator.max_depth != -1)
            throw std::runtime_error("\"max_depth\" must be set to -1 (infinite) or a value >= 0");   // integrator.cpp:313-314
    } catch (...) { delete sc; throw; }
    return sc;
}

} // namespace orc
