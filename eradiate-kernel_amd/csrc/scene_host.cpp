// scene_host.cpp -- host side of libmtsamd.so: the "plugin constructors" that turn the C-ABI scene
// description into the flattened device scene (dscene.h), and its upload to HBM.
//
// Each block follows the constructor of the reference plugin it stands for (citations relative to
// /root/reference) with the same float arithmetic (explicit fma where the reference uses fmadd), so the
// constants the kernels read are the ones a scalar_rgb build would hold.
#include <cmath>
#include <cstring>
#include <algorithm>
#include "scene_host.h"
#include "cie_tables.h"

namespace mtsamd {

static void mat_transpose(const float *a, float *o) { for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) o[r * 4 + c] = a[c * 4 + r]; }
// enoki matrix product: result(r, j) = fma chain over k of a(r, k) * b(k, j)
static void mat_mul(const float *a, const float *b, float *o) {
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 4; ++r) {
            float acc = a[r * 4 + 0] * b[0 * 4 + j];
            for (int k = 1; k < 4; ++k) acc = pm_fma(a[r * 4 + k], b[k * 4 + j], acc);
            o[r * 4 + j] = acc;
        }
}
static DXf xf_from_abi(const mts_transform &t) { DXf x; memcpy(x.m, t.matrix, 64); memcpy(x.it, t.inverse_transpose, 64); return x; }
static DXf xf_identity() { DXf x; memset(&x, 0, sizeof(x)); for (int i = 0; i < 4; ++i) x.m[i * 5] = x.it[i * 5] = 1.f; return x; }
static DXf xf_inverse(const DXf &x) { DXf r; mat_transpose(x.it, r.m); mat_transpose(x.m, r.it); return r; }          // transform.h:59-61
static DXf xf_mul(const DXf &a, const DXf &b) { DXf r; mat_mul(a.m, b.m, r.m); mat_mul(a.it, b.it, r.it); return r; }  // transform.h:53-56
static DXf xf_scale(F3 s) { DXf x = xf_identity(); x.m[0] = s.x; x.m[5] = s.y; x.m[10] = s.z; x.it[0] = 1.f / s.x; x.it[5] = 1.f / s.y; x.it[10] = 1.f / s.z; return x; }
static DXf xf_translate(F3 t) { DXf x = xf_identity(); x.m[3] = t.x; x.m[7] = t.y; x.m[11] = t.z; x.it[12] = -t.x; x.it[13] = -t.y; x.it[14] = -t.z; return x; }

static void bbox_reset(DBBox &b) { for (int i = 0; i < 3; ++i) { b.min[i] = INFINITY; b.max[i] = -INFINITY; } }
static void bbox_expand(DBBox &b, F3 p) {
    float v[3] = { p.x, p.y, p.z };
    for (int i = 0; i < 3; ++i) { b.min[i] = pm_min(b.min[i], v[i]); b.max[i] = pm_max(b.max[i], v[i]); }
}
static void bbox_expand(DBBox &b, const DBBox &o) { bbox_expand(b, f3(o.min)); bbox_expand(b, f3(o.max)); }
static void store3(float *d, F3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }

static void check_index(int i, int n, const char *what, bool allow_none) {
    if (i < 0 && allow_none) return;
    if (i < 0 || i >= n) throw std::runtime_error(std::string("index out of range: ") + what);
}

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

static uint32_t bsdf_flags(int type) {
    if (type == MTS_BSDF_DIFFUSE) return F_DiffuseReflection | F_FrontSide;                // diffuse.cpp:55
    if (type == MTS_BSDF_NULL) return F_Null | F_FrontSide | F_BackSide;                   // null.cpp:26
    if (type == MTS_BSDF_BILAMBERTIAN) return F_DiffuseReflection | F_DiffuseTransmission | F_FrontSide | F_BackSide;   // bilambertian.cpp:55-60
    return F_GlossyReflection | F_FrontSide;                                               // rpv.cpp:66
}

// Shape constructors: rectangle.cpp:59-74, cube.cpp:70-112 (+ mesh.cpp), sphere.cpp:80-105
static DShape build_shape(const mts_shape &d, HostScene &hs, DBBox &shape_bbox, int &prim_count) {
    DShape s; memset(&s, 0, sizeof(s));
    s.type = d.type;
    s.to_world = xf_from_abi(d.to_world);
    s.bsdf = d.bsdf; s.interior = d.interior_medium; s.exterior = d.exterior_medium; s.emitter = d.emitter;
    s.is_medium_transition = (d.interior_medium >= 0 || d.exterior_medium >= 0) ? 1 : 0;   // shape.h:341
    s.flip_normals = d.flip_normals != 0;
    bbox_reset(shape_bbox);
    if (d.type == MTS_SHAPE_RECTANGLE) {
        if (d.flip_normals) s.to_world = xf_mul(s.to_world, xf_scale(f3(1.f, 1.f, -1.f)));
        s.to_object = xf_inverse(s.to_world);
        F3 dp_du = mat_vector(s.to_world.m, f3(2.f, 0.f, 0.f)), dp_dv = mat_vector(s.to_world.m, f3(0.f, 2.f, 0.f));
        F3 n = normalize(mat_vector(s.to_world.it, f3(0.f, 0.f, 1.f)));
        store3(s.frame_s, dp_du); store3(s.frame_t, dp_dv); store3(s.frame_n, n);
        s.inv_surface_area = pm_rcp(norm(cross(dp_du, dp_dv)));
        bbox_expand(shape_bbox, mat_point_affine(s.to_world.m, f3(-1.f, -1.f, 0.f)));
        bbox_expand(shape_bbox, mat_point_affine(s.to_world.m, f3(1.f, -1.f, 0.f)));
        bbox_expand(shape_bbox, mat_point_affine(s.to_world.m, f3(1.f, 1.f, 0.f)));
        bbox_expand(shape_bbox, mat_point_affine(s.to_world.m, f3(-1.f, 1.f, 0.f)));
        prim_count = 1;
    } else if (d.type == MTS_SHAPE_DISK) {                                                     // disk.cpp:74-111
        if (d.flip_normals) s.to_world = xf_mul(s.to_world, xf_scale(f3(1.f, 1.f, -1.f)));
        s.to_object = xf_inverse(s.to_world);
        F3 dp_du = mat_vector(s.to_world.m, f3(1.f, 0.f, 0.f)), dp_dv = mat_vector(s.to_world.m, f3(0.f, 1.f, 0.f));
        float du = norm(dp_du), dv = norm(dp_dv);
        F3 n = normalize(mat_vector(s.to_world.it, f3(0.f, 0.f, 1.f)));
        F3 fs = dp_du / du, ft = dp_dv / dv;
        store3(s.frame_s, fs); store3(s.frame_t, ft); store3(s.frame_n, n);
        float dts = dot(ft * dv, fs);
        float h = pm_sqrt(dv * dv - dts * dts);
        s.surface_area = MTS_PI * du * h;
        s.inv_surface_area = 1.f / s.surface_area;
        bbox_expand(shape_bbox, mat_point_affine(s.to_world.m, f3(-1.f, -1.f, 0.f)));
        bbox_expand(shape_bbox, mat_point_affine(s.to_world.m, f3(-1.f, 1.f, 0.f)));
        bbox_expand(shape_bbox, mat_point_affine(s.to_world.m, f3(1.f, -1.f, 0.f)));
        bbox_expand(shape_bbox, mat_point_affine(s.to_world.m, f3(1.f, 1.f, 0.f)));
        prim_count = 1;
    } else if (d.type == MTS_SHAPE_CUBE || d.type == MTS_SHAPE_MESH) {
        s.to_object = xf_inverse(s.to_world);
        int nv, nf; const float *pos, *nor, *uv; const uint32_t *fc;
        if (d.type == MTS_SHAPE_CUBE) { nv = 24; nf = 12; pos = &CUBE_VERTICES[0][0]; nor = &CUBE_NORMALS[0][0]; uv = &CUBE_TEXCOORDS[0][0]; fc = &CUBE_TRIANGLES[0][0]; }
        else { nv = d.vertex_count; nf = d.face_count; pos = d.vertex_positions; nor = d.vertex_normals; uv = d.vertex_texcoords; fc = d.faces;
               if (!pos || !fc || nv <= 0 || nf <= 0) throw std::runtime_error("mesh: missing vertex / face data"); }
        // all meshes share one vertex index space; normals / texcoords arrays stay aligned with positions
        s.vertex_offset = (int32_t) (hs.positions.size() / 3);
        s.face_offset = (int32_t) (hs.faces.size() / 3);
        s.has_normals = nor ? 1 : 0; s.has_texcoords = uv ? 1 : 0;
        for (int i = 0; i < nv; ++i) {
            F3 p = mat_point_affine(s.to_world.m, f3(pos + 3 * i));
            bbox_expand(shape_bbox, p);
            hs.positions.push_back(p.x); hs.positions.push_back(p.y); hs.positions.push_back(p.z);
            F3 n = f3s(0.f);
            if (nor) n = normalize(mat_vector(s.to_world.it, f3(nor + 3 * i)));
            hs.normals.push_back(n.x); hs.normals.push_back(n.y); hs.normals.push_back(n.z);
            hs.texcoords.push_back(uv ? uv[2 * i] : 0.f); hs.texcoords.push_back(uv ? uv[2 * i + 1] : 0.f);
        }
        for (int i = 0; i < 3 * nf; ++i) {
            if (fc[i] >= (uint32_t) nv) throw std::runtime_error("mesh: face index out of range");
            hs.faces.push_back(fc[i]);
        }
        prim_count = nf;
    } else if (d.type == MTS_SHAPE_SPHERE) {
        // to_world * translate(center) * scale(radius); radius / rotation recovered from the columns
        // (the reference uses enoki's polar decomposition, absent; identical for rotation * uniform scale)
        DXf tw = xf_mul(s.to_world, xf_mul(xf_translate(f3(d.center)), xf_scale(f3s(d.radius))));
        F3 c0 = f3(tw.m[0], tw.m[4], tw.m[8]), c1 = f3(tw.m[1], tw.m[5], tw.m[9]), c2 = f3(tw.m[2], tw.m[6], tw.m[10]);
        float r0 = norm(c0), r1 = norm(c1), r2 = norm(c2);
        if (pm_abs(r0 - r1) > 1e-6f * r0 || pm_abs(r0 - r2) > 1e-6f * r0 ||
            pm_abs(dot(c0, c1)) > 1e-6f * r0 * r0 || pm_abs(dot(c0, c2)) > 1e-6f * r0 * r0 || pm_abs(dot(c1, c2)) > 1e-6f * r0 * r0)
            throw std::runtime_error("'to_world' transform shouldn't contain any scale or shear along the sphere axes");
        F3 center = f3(tw.m[3], tw.m[7], tw.m[11]);
        store3(s.center, center); s.radius = r0;
        F3 q0 = c0 / r0, q1 = c1 / r0, q2 = c2 / r0;
        float R[9] = { q0.x, q1.x, q2.x, q0.y, q1.y, q2.y, q0.z, q1.z, q2.z };
        DXf rec = xf_identity();
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) rec.m[r * 4 + c] = R[r * 3 + c] * s.radius;
        rec.m[3] = center.x; rec.m[7] = center.y; rec.m[11] = center.z;
        float invm[16]; memset(invm, 0, sizeof(invm));
        for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) invm[r * 4 + c] = R[c * 3 + r] / s.radius;
        for (int r = 0; r < 3; ++r) invm[r * 4 + 3] = -(invm[r * 4] * center.x + invm[r * 4 + 1] * center.y + invm[r * 4 + 2] * center.z);
        invm[15] = 1.f;
        mat_transpose(invm, rec.it);
        s.to_world = rec; s.to_object = xf_inverse(rec);
        s.inv_surface_area = pm_rcp(4.f * MTS_PI * s.radius * s.radius);
        store3(shape_bbox.min, center - f3s(s.radius)); store3(shape_bbox.max, center + f3s(s.radius));
        prim_count = 1;
    } else throw std::runtime_error("unknown shape type");
    return s;
}

static DRFilter build_rfilter(int type, float radius, float stddev, std::vector<float> &values) {
    DRFilter f; memset(&f, 0, sizeof(f));
    f.type = type;
    if (type == MTS_RFILTER_BOX) f.radius = radius + MTS_RAY_EPSILON;                         // box.cpp:31
    else if (type == MTS_RFILTER_GAUSSIAN) {                                                    // gaussian.cpp:33-42
        f.stddev = stddev; f.radius = 4 * stddev; f.alpha = -1.f / (2.f * stddev * stddev); f.bias = pm_exp(f.alpha * (f.radius * f.radius));
    } else throw std::runtime_error("unknown reconstruction filter");
    values.assign(32, 0.f);
    for (int i = 0; i < 31; ++i) {                                                              // rfilter.cpp:9-20 (MTS_FILTER_RESOLUTION = 31)
        float x = (f.radius * i) / 31;
        values[i] = type == MTS_RFILTER_BOX ? (pm_abs(x) <= f.radius ? 1.f : 0.f) : pm_max(0.f, pm_exp(f.alpha * (x * x)) - f.bias);
    }
    values[31] = 0;
    f.scale_factor = 31 / f.radius;
    f.border_size = (int) std::ceil(f.radius - .5f - 2.f * MTS_RAY_EPSILON);
    return f;
}

// Transform::perspective (transform.h:203-220) and perspective_projection (sensor.h:196-231)
static DXf xf_perspective(float fov, float near_, float far_) {
    float recip = 1.f / (far_ - near_);
    float tan_ = std::tan(fov * .5f * (MTS_PI / 180.f)), cot = 1.f / tan_;
    DXf x; memset(&x, 0, sizeof(x));
    x.m[0] = cot; x.m[5] = cot; x.m[10] = far_ * recip; x.m[11] = -near_ * far_ * recip; x.m[14] = 1.f;
    float inv[16]; memset(inv, 0, sizeof(inv));
    inv[0] = tan_; inv[5] = tan_; inv[15] = 1.f / near_; inv[11] = 1.f; inv[14] = (near_ - far_) / (far_ * near_);
    mat_transpose(inv, x.it);
    return x;
}

// Which promises of integrator_dev.h's scene traits (MT_*: what the lean translation units were compiled without) this scene keeps.
// mts_render launches the leanest kernel whose promises are all kept; a scene that keeps none runs on the general kernels.
static int scene_traits(const HostScene &hs, bool spectral) {
    int tr = 0;
    bool media = !hs.media.empty();
    for (size_t i = 0; i < hs.media.size(); ++i) {
        const DMedium &m = hs.media[i];
        if (spectral) media = media && !m.is_homogeneous && m.shared_grid == 2 && m.has_spectral_extinction;      // MT_MEDIA of the spectral variant
        else media = media && !m.is_homogeneous && i < hs.pair_data.size() && !hs.pair_data[i].empty() && m.grey && m.has_spectral_extinction;
    }
    if (media) tr |= MT_MEDIA;
    bool homog = !hs.media.empty();
    for (const DMedium &m : hs.media) homog = homog && m.is_homogeneous;
    if (homog) tr |= MT_HOMOG;
    if (hs.bvh_nodes.empty()) tr |= MT_NO_BVH;
    bool sphere = false, rpv = false, shape_emitter = false, tree = false, grid_eval = false;
    for (const DShape &sh : hs.shapes) sphere = sphere || sh.type == MTS_SHAPE_SPHERE;
    sphere = sphere || hs.scene.sensor.target_shape.type == MTS_SHAPE_SPHERE || hs.scene.sensor.origin_shape.type == MTS_SHAPE_SPHERE;   // the distant sensors' own shapes
    for (const DBsdf &b : hs.bsdfs) rpv = rpv || b.type == MTS_BSDF_RPV;
    for (const DEmitter &e : hs.emitters) shape_emitter = shape_emitter || e.shape >= 0;
    if (!media)                                                         // a medium that is not on a pair grid reads its grids through volume_eval()
        for (const DMedium &m : hs.media)
            grid_eval = grid_eval || hs.volumes[(size_t) m.sigma_t].type == MTS_VOLUME_GRID || hs.volumes[(size_t) m.albedo].type == MTS_VOLUME_GRID;
    for (const DPhase &ph : hs.phases)
        if (ph.type == MTS_PHASE_BLEND) {
            tree = tree || ph.size > 1;
            grid_eval = grid_eval || (ph.weight_volume >= 0 && hs.volumes[(size_t) ph.weight_volume].type == MTS_VOLUME_GRID);
        }
    if (!sphere) tr |= MT_NO_SPHERE;
    if (!grid_eval) tr |= MT_NO_GRID_EVAL;                               // no grid reaches volume_eval() (pair-grid media do not)
    if (!shape_emitter) tr |= MT_NO_SHAPE_EMITTER;
    if (!tree) tr |= MT_NO_PHASE_TREE;
    if (!rpv) tr |= MT_NO_RPV;
    return tr;
}

HostScene *build_host_scene(const mts_scene_desc *d) {
    if (!d) throw std::runtime_error("scene description is NULL");
    if (d->abi_version != MTS_ABI_VERSION) throw std::runtime_error("scene description: ABI version mismatch");
    std::unique_ptr<HostScene> hsp(new HostScene());
    HostScene &hs = *hsp;
    DScene &sc = hs.scene; memset(&sc, 0, sizeof(sc));
    const bool spectral = d->integrator.spectral != 0;
    // ---- spectra (spectral variant; uniform.cpp:34-52, regular.cpp:27-58 + distr_1d.h:318-345)
    if (spectral) {
        if (d->integrator.monochrome) throw std::runtime_error("a scene is either monochromatic or spectral");
        for (int i = 0; i < d->spectrum_count; ++i) {
            const mts_spectrum &sp = d->spectra[i];
            DSpectrum ds; memset(&ds, 0, sizeof(ds));
            ds.type = sp.type; ds.value = sp.value; ds.lambda_min = sp.lambda_min; ds.lambda_max = sp.lambda_max;
            hs.spectrum_values.emplace_back(); hs.spectrum_wavelengths.emplace_back(); hs.spectrum_cdf.emplace_back();
            if (sp.type == MTS_SPECTRUM_UNIFORM) {
                ds.lambda_min = std::max(sp.lambda_min, 280.f); ds.lambda_max = std::min(sp.lambda_max, 2400.f);       // MTS_WAVELENGTH_MIN / MAX
                if (!(ds.lambda_min < ds.lambda_max)) throw std::runtime_error("UniformSpectrum: 'lambda_min' must be less than 'lambda_max'");
            } else if (sp.type == MTS_SPECTRUM_REGULAR) {
                if (!(sp.lambda_min < sp.lambda_max)) throw std::runtime_error("ContinuousDistribution: invalid range!");
                if (sp.count < 2 || !sp.values) throw std::runtime_error("ContinuousDistribution: needs at least two entries!");
                bool mass = false;
                for (int k = 0; k < sp.count; ++k) { if (sp.values[k] < 0.f) throw std::runtime_error("ContinuousDistribution: entries must be non-negative!"); mass = mass || sp.values[k] > 0.f; }
                if (!mass) throw std::runtime_error("ContinuousDistribution: no probability mass found!");
                hs.spectrum_values.back().assign(sp.values, sp.values + sp.count);
                ds.count = sp.count;
                const double interval_size = (double(sp.lambda_max) - double(sp.lambda_min)) / (sp.count - 1);
                ds.inv_interval_size = (float) (1. / interval_size);
            } else if (sp.type == MTS_SPECTRUM_IRREGULAR) {                  // irregular.cpp:33-63 + IrregularContinuousDistribution (distr_1d.h:560-600)
                if (sp.count < 2 || !sp.values || !sp.wavelengths) throw std::runtime_error("IrregularContinuousDistribution: needs at least two entries!");
                bool mass = false;
                for (int k = 0; k < sp.count; ++k) {
                    if (sp.values[k] < 0.f) throw std::runtime_error("IrregularContinuousDistribution: entries must be non-negative!");
                    if (k > 0 && !(sp.wavelengths[k] > sp.wavelengths[k - 1])) throw std::runtime_error("IrregularContinuousDistribution: node positions must be strictly increasing!");
                    mass = mass || sp.values[k] > 0.f;
                }
                if (!mass) throw std::runtime_error("IrregularContinuousDistribution: no probability mass found!");
                hs.spectrum_values.back().assign(sp.values, sp.values + sp.count);
                hs.spectrum_wavelengths.back().assign(sp.wavelengths, sp.wavelengths + sp.count);
                ds.count = sp.count; ds.lambda_min = sp.wavelengths[0]; ds.lambda_max = sp.wavelengths[sp.count - 1];
            } else if (sp.type == MTS_SPECTRUM_DISCRETE) {                   // discrete.cpp:45-100 + DiscreteDistribution::update (distr_1d.h:49-83)
                if (sp.count < 1 || !sp.values || !sp.wavelengths || !sp.pmf) throw std::runtime_error("DiscreteDistribution: empty distribution!");
                hs.spectrum_values.back().assign(sp.values, sp.values + sp.count);
                hs.spectrum_wavelengths.back().assign(sp.wavelengths, sp.wavelengths + sp.count);
                std::vector<float> &cdf = hs.spectrum_cdf.back(); cdf.resize((size_t) sp.count);
                ds.count = sp.count; ds.valid_x = ds.valid_y = (uint32_t) -1;
                double sum = 0.0;
                for (int k = 0; k < sp.count; ++k) {
                    const double value = (double) sp.pmf[k];
                    sum += value; cdf[(size_t) k] = (float) sum;
                    if (value < 0.0) throw std::runtime_error("DiscreteDistribution: entries must be non-negative!");
                    else if (value > 0.0) { if (ds.valid_x == (uint32_t) -1) ds.valid_x = (uint32_t) k; ds.valid_y = (uint32_t) k; }
                }
                if (ds.valid_x == (uint32_t) -1) throw std::runtime_error("DiscreteDistribution: no probability mass found!");
                ds.cdf_sum = (float) sum;
            } else throw std::runtime_error("unknown spectrum type");
            hs.spectra.push_back(ds);
        }
    }
    auto spectrum_index = [&](int idx, const char *what) {
        if (idx < 0 || idx >= d->spectrum_count) throw std::runtime_error(std::string("spectral variant: missing spectrum for ") + what);
        return idx;
    };
    auto add_uniform_spectrum = [&](float value) {
        DSpectrum ds; memset(&ds, 0, sizeof(ds)); ds.type = MTS_SPECTRUM_UNIFORM; ds.value = value; ds.lambda_min = 280.f; ds.lambda_max = 2400.f;
        hs.spectra.push_back(ds); hs.spectrum_values.emplace_back(); hs.spectrum_wavelengths.emplace_back(); hs.spectrum_cdf.emplace_back();
        return (int32_t) hs.spectra.size() - 1;
    };

    // ---- volumes (texture.cpp:89-92, texture.h:262-269, grid3d.cpp:137-161, volume_data.h:24-33,86-98)
    for (int i = 0; i < d->volume_count; ++i) {
        const mts_volume &v = d->volumes[i];
        DVolume dv; memset(&dv, 0, sizeof(dv));
        dv.type = v.type == MTS_VOLUME_GRID_SPECTRAL ? MTS_VOLUME_GRID : v.type; memcpy(dv.value, v.value, 12);
        DVolumeSp vsp; memset(&vsp, 0, sizeof(vsp)); vsp.value_sp = -1;
        if (spectral && v.type == MTS_VOLUME_CONST) vsp.value_sp = spectrum_index(v.value_spectrum, "a constvolume");
        if (v.type == MTS_VOLUME_GRID_SPECTRAL) {
            if (!spectral) throw std::runtime_error("This volume data source can only be used with a spectral variant!");     // gridvolume_spectral.cpp:86-88
            if (v.filter_type != MTS_FILTER_TRILINEAR) throw std::runtime_error("Invalid filter type, must be \"trilinear\"!");
            if (v.channels < 2 || v.channels > 255) throw std::runtime_error("gridvolume_spectral: between 2 and 255 spectral nodes are supported");
            vsp.spectral_grid = 1; vsp.lambda_min = v.lambda_min; vsp.lambda_max = v.lambda_max;
        }
        hs.volume_sp.push_back(vsp);
        DXf w2l = xf_inverse(xf_from_abi(v.to_world));
        if (v.type == MTS_VOLUME_GRID || v.type == MTS_VOLUME_GRID_SPECTRAL) {
            if (!v.data) throw std::runtime_error("gridvolume: missing data");
            if ((long) v.nx * v.ny * v.nz < 8) throw std::runtime_error("Invalid grid dimensions (must have at least one value at each corner)");
            if (v.type == MTS_VOLUME_GRID && v.channels != 1 && v.channels != 3) throw std::runtime_error("Unsupported channel count (expected 1 or 3)");
            if (spectral && v.type == MTS_VOLUME_GRID && v.channels != 1)
                throw std::runtime_error("spectral variant: 3-channel grids need the sRGB upsampling model (ext/rgb2spec data, absent); use gridvolume_spectral");
            if ((int64_t) v.nx * v.ny * v.nz * v.channels >= (int64_t) 1 << 31) throw std::runtime_error("gridvolume: more than 2^31 values");
            dv.nx = v.nx; dv.ny = v.ny; dv.nz = v.nz; dv.channels = v.channels; dv.filter = v.filter_type; dv.wrap = v.wrap_mode;
            size_t n = (size_t) v.nx * v.ny * v.nz * v.channels;
            float mx = -INFINITY;
            for (size_t k = 0; k < n; ++k) mx = std::max(mx, v.data[k]);
            dv.max = mx; dv.has_max = 1;
            {   // a profile that only varies with z (bitwise comparison: the lookups then read column (0, 0) for every corner)
                const size_t row = (size_t) v.nx * v.channels, col = (size_t) v.channels;
                bool equal = true;
                for (size_t z = 0; z < (size_t) v.nz && equal; ++z) {
                    const float *base = v.data + z * (size_t) v.ny * row;
                    for (size_t c = 1; c < (size_t) v.ny * v.nx && equal; ++c) equal = memcmp(base, base + c * col, col * sizeof(float)) == 0;
                }
                dv.columns_equal = equal ? 1 : 0;
            }
            hs.grid_data.emplace_back(v.data, v.data + n);
            if (v.use_grid_bbox) {
                F3 bmin = f3(v.file_bbox_min), bmax = f3(v.file_bbox_max);
                w2l = xf_mul(xf_mul(xf_scale(vrcp(bmax - bmin)), xf_translate(-1.f * bmin)), w2l);
            }
            if (v.has_max_value) dv.max = v.max_value;
        } else if (v.type != MTS_VOLUME_CONST) throw std::runtime_error("unknown volume type");
        else hs.grid_data.emplace_back();
        memcpy(dv.w2l, w2l.m, 64);
        dv.affine = (w2l.m[12] == 0.f && w2l.m[13] == 0.f && w2l.m[14] == 0.f && w2l.m[15] == 1.f) ? 1 : 0;
        DXf inv = xf_inverse(w2l);
        F3 a = mat_point(inv.m, f3s(0.f)), b = mat_point(inv.m, f3s(1.f));
        store3(dv.bbox.min, a); store3(dv.bbox.max, a); bbox_expand(dv.bbox, b);
        hs.volumes.push_back(dv);
    }
    // ---- phase functions (hg.cpp:43-49, tabphase.cpp:33-51, distr_1d.h:293-345, blendphase.cpp:33-56)
    for (int i = 0; i < d->phase_count; ++i) {
        const mts_phase &p = d->phases[i];
        DPhase dp; memset(&dp, 0, sizeof(dp));
        dp.type = p.type; dp.g = p.g; dp.child[0] = p.child[0]; dp.child[1] = p.child[1]; dp.weight_volume = p.weight_volume;
        hs.tab_pdf.emplace_back(); hs.tab_cdf.emplace_back();
        if (p.type == MTS_PHASE_HG && (p.g >= 1 || p.g <= -1)) throw std::runtime_error("The asymmetry parameter must lie in the interval (-1, 1)!");
        if (p.type == MTS_PHASE_BLEND) {
            check_index(p.child[0], d->phase_count, "blendphase child", false);
            check_index(p.child[1], d->phase_count, "blendphase child", false);
            check_index(p.weight_volume, d->volume_count, "blendphase weight", false);
            // nested blendphase plugins (blendphase.cpp:42-66: two arbitrary PhaseFunction children): children come before their parent
            // in the description (the order a loader constructs them in; rules out cycles), DPhase::size = depth of the tree below
            if (p.child[0] >= i || p.child[1] >= i) throw std::runtime_error("blendphase: a nested phase function must precede the blendphase that holds it");
            int depth = 1;
            for (int c = 0; c < 2; ++c)
                if (d->phases[p.child[c]].type == MTS_PHASE_BLEND) depth = std::max(depth, 1 + hs.phases[(size_t) p.child[c]].size);
            if (depth > 8) throw std::runtime_error("blendphase: more than 8 nested levels are not supported by this backend");
            dp.size = depth;
        } else if (p.type == MTS_PHASE_TABULATED) {
            size_t size = (size_t) p.tab_count;
            if (size < 2 || !p.tab_values) throw std::runtime_error("ContinuousDistribution: needs at least two entries!");
            std::vector<float> &pdf = hs.tab_pdf.back(), &cdf = hs.tab_cdf.back();
            pdf.assign(p.tab_values, p.tab_values + size); cdf.resize(size - 1);
            dp.size = (int32_t) size; dp.range_x = -1.f; dp.range_y = 1.f;
            dp.valid_x = dp.valid_y = (uint32_t) -1;
            double range = double(dp.range_y) - double(dp.range_x), interval_size = range / (size - 1), integral = 0.;
            for (size_t k = 0; k < size - 1; ++k) {
                double y0 = (double) pdf[k], y1 = (double) pdf[k + 1];
                double value = 0.5 * interval_size * (y0 + y1);
                integral += value;
                cdf[k] = (float) integral;
                if (y0 < 0. || y1 < 0.) throw std::runtime_error("ContinuousDistribution: entries must be non-negative!");
                else if (value > 0.) { if (dp.valid_x == (uint32_t) -1) dp.valid_x = (uint32_t) k; dp.valid_y = (uint32_t) k; }
            }
            if (dp.valid_x == (uint32_t) -1) throw std::runtime_error("ContinuousDistribution: no probability mass found!");
            dp.integral = (float) integral; dp.normalization = (float) (1. / integral);
            dp.interval_size = (float) interval_size; dp.inv_interval_size = (float) (1. / interval_size);
        } else if (p.type < MTS_PHASE_ISOTROPIC || p.type > MTS_PHASE_TABULATED) throw std::runtime_error("unknown phase function type");
        hs.phases.push_back(dp);
    }
    // ---- media (medium.cpp:12-29, homogeneous.cpp:21-28, heterogeneous.cpp:21-31)
    for (int i = 0; i < d->medium_count; ++i) {
        const mts_medium &m = d->media[i];
        check_index(m.sigma_t_volume, d->volume_count, "medium sigma_t", false);
        check_index(m.albedo_volume, d->volume_count, "medium albedo", false);
        check_index(m.phase, d->phase_count, "medium phase", false);
        DMedium dm; memset(&dm, 0, sizeof(dm));
        dm.type = m.type; dm.sigma_t = m.sigma_t_volume; dm.albedo = m.albedo_volume; dm.phase = m.phase; dm.scale = m.scale;
        dm.sample_emitters = m.sample_emitters != 0; dm.has_spectral_extinction = m.has_spectral_extinction != 0;
        dm.is_homogeneous = m.type == MTS_MEDIUM_HOMOGENEOUS;
        if (m.type == MTS_MEDIUM_HETEROGENEOUS) {
            const DVolume &st = hs.volumes[m.sigma_t_volume];
            if (!st.has_max) throw std::runtime_error("max() not implemented (constvolume sigma_t in heterogeneous medium)");
            dm.max_density = dm.scale * st.max;
            {   // the kernels divide by the majorant two or three times per tracking step: with the correctly rounded reciprocal at hand an
                // IEEE-exact quotient is five multiply-adds instead of the ~11-instruction division sequence (div_by_invariant,
                // volpath_flat.h).  Usable when both are normal numbers well inside the exponent range and the significand of the
                // divisor is not all ones (the one case Markstein's correction step does not cover).
                const uint32_t b = pm_bits(dm.max_density), e = (b >> 23) & 0xffu;
                const float rd = 1.0f / dm.max_density;
                const uint32_t er = (pm_bits(rd) >> 23) & 0xffu;
                dm.inv_max_density = (dm.max_density > 0.f && e >= 67u && e <= 187u && er >= 67u && er <= 187u && (b & 0x7fffffu) != 0x7fffffu) ? rd : 0.f;
            }
            dm.aabb = st.bbox;
        } else if (m.type != MTS_MEDIUM_HOMOGENEOUS) throw std::runtime_error("unknown medium type");
        {   // kernel fast paths that do not change a single bit of the result
            const DVolume &a = hs.volumes[m.sigma_t_volume], &b = hs.volumes[m.albedo_volume];
            dm.shared_grid = (a.type == MTS_VOLUME_GRID && b.type == MTS_VOLUME_GRID && a.nx == b.nx && a.ny == b.ny && a.nz == b.nz &&
                              a.filter == b.filter && a.wrap == b.wrap && memcmp(a.w2l, b.w2l, 64) == 0) ? 1 : 0;
            auto grey = [](const DVolume &v) { return v.type == MTS_VOLUME_GRID ? v.channels == 1 : (v.value[0] == v.value[1] && v.value[1] == v.value[2]); };
            dm.grey = (grey(a) && grey(b) && !spectral) ? 1 : 0;       // spectral variant: values depend on the wavelength
            if (spectral && dm.shared_grid) {                          // two gridvolume_spectral grids over one spectral interval: 2
                const DVolumeSp &sa = hs.volume_sp[m.sigma_t_volume], &sb = hs.volume_sp[m.albedo_volume];
                if (sa.spectral_grid && sb.spectral_grid && a.channels == b.channels && sa.lambda_min == sb.lambda_min && sa.lambda_max == sb.lambda_max)
                    dm.shared_grid = 2;
            }
            std::vector<float> pair;
            if (dm.shared_grid && a.channels == 1 && b.channels == 1 && a.filter == MTS_FILTER_TRILINEAR && a.wrap == MTS_WRAP_CLAMP) {
                const std::vector<float> &ga = hs.grid_data[m.sigma_t_volume], &gb = hs.grid_data[m.albedo_volume];
                // rows of at least two voxels (one 16-byte gather = both x-neighbours of both grids): a one-column grid -- the
                // nz x 1 x 1 grids of 1-D atmospheres -- is stored with its column twice; both x-neighbours clamp to voxel 0 anyway
                const size_t sx = a.nx < 2 ? 2 : (size_t) a.nx, rows = (size_t) a.ny * a.nz;
                pair.assign(2 * (sx * rows + 1), 0.f);
                for (size_t r = 0; r < rows; ++r)
                    for (size_t x = 0; x < sx; ++x) {
                        const size_t src = r * (size_t) a.nx + (x < (size_t) a.nx ? x : (size_t) a.nx - 1);
                        pair[2 * (r * sx + x)] = ga[src]; pair[2 * (r * sx + x) + 1] = gb[src];
                    }
                memcpy(dm.pair_w2l, a.w2l, 64); dm.pair_nx = a.nx; dm.pair_ny = a.ny; dm.pair_nz = a.nz; dm.pair_affine = (a.affine ? 1 : 0) | ((a.columns_equal && b.columns_equal) ? 2 : 0);
            }
            hs.pair_data.push_back(std::move(pair));
        }
        hs.media.push_back(dm);
    }
    // ---- BSDFs
    for (int i = 0; i < d->bsdf_count; ++i) {
        const mts_bsdf &b = d->bsdfs[i];
        if (b.type < MTS_BSDF_DIFFUSE || b.type > MTS_BSDF_BILAMBERTIAN) throw std::runtime_error("unknown BSDF type");
        DBsdf db; memset(&db, 0, sizeof(db));
        db.type = b.type; memcpy(db.reflectance, b.reflectance, 12); memcpy(db.rho_0, b.rho_0, 12); memcpy(db.k, b.k, 12);
        memcpy(db.g, b.g, 12); memcpy(db.rho_c, b.rho_c, 12); memcpy(db.transmittance, b.transmittance, 12); db.flags = bsdf_flags(b.type);
        hs.bsdfs.push_back(db);
        if (spectral) {                                                  // which of the six colour parameters each plugin reads
            static const bool used[4][6] = { { 1, 0, 0, 0, 0, 0 } /* diffuse */, { 0, 0, 0, 0, 0, 0 } /* null */, { 0, 1, 1, 1, 1, 0 } /* rpv */, { 1, 0, 0, 0, 0, 1 } /* bilambertian */ };
            for (int k = 0; k < MTS_BSDF_SP_COUNT; ++k) hs.bsdf_sp.push_back(used[b.type][k] ? spectrum_index(b.spectrum[k], "a BSDF parameter") : 0);
        }
    }
    // default BSDFs appended after the user's (shape.cpp:74-80): diffuse 0.5, and diffuse 0 for emitters
    DBsdf def; memset(&def, 0, sizeof(def)); def.type = MTS_BSDF_DIFFUSE; def.flags = bsdf_flags(MTS_BSDF_DIFFUSE);
    def.reflectance[0] = def.reflectance[1] = def.reflectance[2] = .5f;
    const int default_bsdf = (int) hs.bsdfs.size(); hs.bsdfs.push_back(def);
    def.reflectance[0] = def.reflectance[1] = def.reflectance[2] = 0.f;
    const int default_emitter_bsdf = (int) hs.bsdfs.size(); hs.bsdfs.push_back(def);
    if (spectral) {
        const int32_t half = add_uniform_spectrum(.5f), zero = add_uniform_spectrum(0.f);
        for (int k = 0; k < MTS_BSDF_SP_COUNT; ++k) hs.bsdf_sp.push_back(half);
        for (int k = 0; k < MTS_BSDF_SP_COUNT; ++k) hs.bsdf_sp.push_back(zero);
    }
    // ---- shapes + scene bounding box (scene.cpp:31-40)
    bbox_reset(sc.bbox);
    for (int i = 0; i < d->shape_count; ++i) {
        const mts_shape &s = d->shapes[i];
        check_index(s.bsdf, d->bsdf_count, "shape bsdf", true);
        check_index(s.interior_medium, d->medium_count, "shape interior", true);
        check_index(s.exterior_medium, d->medium_count, "shape exterior", true);
        check_index(s.emitter, d->emitter_count, "shape emitter", true);
        DBBox sb; int prim_count;
        DShape ds = build_shape(s, hs, sb, prim_count);
        if (ds.bsdf < 0) ds.bsdf = ds.emitter >= 0 ? default_emitter_bsdf : default_bsdf;
        ds.bsdf_type = hs.bsdfs[(size_t) ds.bsdf].type; ds.bsdf_flags = hs.bsdfs[(size_t) ds.bsdf].flags;
        ds.prim_offset = (int32_t) hs.prims.size();
        hs.shapes.push_back(ds);
        bbox_expand(sc.bbox, sb);
        for (int k = 0; k < prim_count; ++k) {
            DPrim p; p.shape = i; p.index = k; hs.prims.push_back(p);
            float rec[9] = { 0, 0, 0, 0, 0, 0, 0, 0, 0 };
            if (ds.type == MTS_SHAPE_CUBE || ds.type == MTS_SHAPE_MESH) {        // mesh.h:201-207: p0, e1 = p1 - p0, e2 = p2 - p0
                const uint32_t *fi = &hs.faces[3 * (ds.face_offset + k)];
                const float *P = &hs.positions[3 * ds.vertex_offset];
                F3 p0 = f3(P + 3 * fi[0]), e1 = f3(P + 3 * fi[1]) - p0, e2 = f3(P + 3 * fi[2]) - p0;
                store3(rec, p0); store3(rec + 3, e1); store3(rec + 6, e2);
            }
            hs.tri.insert(hs.tri.end(), rec, rec + 9);
            float attr[24] = { 0 };
            if (ds.type == MTS_SHAPE_CUBE || ds.type == MTS_SHAPE_MESH) {
                const uint32_t *fi = &hs.faces[3 * (ds.face_offset + k)];
                for (int c = 0; c < 3; ++c) {
                    memcpy(attr + 3 * c, &hs.positions[3 * (ds.vertex_offset + fi[c])], 12);
                    memcpy(attr + 9 + 3 * c, &hs.normals[3 * (ds.vertex_offset + fi[c])], 12);
                    memcpy(attr + 18 + 2 * c, &hs.texcoords[2 * (ds.vertex_offset + fi[c])], 8);
                }
            }
            hs.tri_attr.insert(hs.tri_attr.end(), attr, attr + 24);
            DWalkPrim w; memset(&w, 0, sizeof(w));
            w.type = ds.type; w.shape = i; w.index = k;
            if (ds.type == MTS_SHAPE_RECTANGLE || ds.type == MTS_SHAPE_DISK) memcpy(w.f, ds.to_object.m, 48);
            else if (ds.type == MTS_SHAPE_SPHERE) { memcpy(w.f, ds.center, 12); w.f[3] = ds.radius; }
            else memcpy(w.f, rec, 36);
            hs.walk.push_back(w);
            // mesh.h:108-116: face area
            hs.area_pmf.push_back((ds.type == MTS_SHAPE_CUBE || ds.type == MTS_SHAPE_MESH) ? 0.5f * norm(cross(f3(rec + 3), f3(rec + 6))) : 0.f);
        }
        if (ds.type == MTS_SHAPE_CUBE || ds.type == MTS_SHAPE_MESH) {
            // Mesh::build_pmf (mesh.cpp:285-312) + DiscreteDistribution::update (distr_1d.h:49-83): running sum in double precision
            DShape &sh = hs.shapes.back();
            double sum = 0.0; sh.area_lo = sh.area_hi = -1;
            for (int k = 0; k < prim_count; ++k) {
                float a = hs.area_pmf[(size_t) sh.prim_offset + k];
                sum += (double) a; hs.area_cdf.push_back((float) sum);
                if (a > 0.f) { if (sh.area_lo < 0) sh.area_lo = k; sh.area_hi = k; }
            }
            sh.surface_area = (float) sum; sh.inv_surface_area = (float) (1.0 / sum);         // mesh.cpp:346-350,373
        } else for (int k = 0; k < prim_count; ++k) hs.area_cdf.push_back(0.f);
    }
    // ---- emitters + set_scene (scene.cpp:41-52,95-97; directional.cpp:68-73; constant.cpp:35-39; bbox.h:329-332)
    sc.environment = -1;
    build_bvh(hs);
    F3 center = (f3(sc.bbox.max) + f3(sc.bbox.min)) * .5f;
    float bsphere_radius = pm_max(MTS_RAY_EPSILON, norm(center - f3(sc.bbox.max)) * (1.f + MTS_RAY_EPSILON));
    for (int i = 0; i < d->emitter_count; ++i) {
        const mts_emitter &e = d->emitters[i];
        DEmitter de; memset(&de, 0, sizeof(de));
        de.type = e.type; de.to_world = xf_from_abi(e.to_world); memcpy(de.radiance, e.radiance, 12); de.shape = e.shape;
        if (e.type == MTS_EMITTER_AREA) {
            check_index(e.shape, d->shape_count, "area emitter shape", false);
            if ((hs.shapes[e.shape].type == MTS_SHAPE_CUBE || hs.shapes[e.shape].type == MTS_SHAPE_MESH) && hs.shapes[e.shape].area_lo < 0)
                throw std::runtime_error("DiscreteDistribution: no probability mass found!");   // distr_1d.h:78-79
        } else if (e.type == MTS_EMITTER_CONSTANT) {
            if (sc.environment >= 0) throw std::runtime_error("Only one environment emitter can be specified per scene.");
            sc.environment = i;
        } else if (e.type != MTS_EMITTER_DIRECTIONAL && e.type != MTS_EMITTER_POINT) throw std::runtime_error("unknown emitter type");
        store3(de.bsphere_center, center); de.bsphere_radius = bsphere_radius;
        hs.emitters.push_back(de);
        if (spectral) hs.emitter_sp.push_back(spectrum_index(e.radiance_spectrum, "an emitter"));
    }
    // ---- sensor, film, sampler
    const mts_sensor &s = d->sensor;
    DSensor &se = sc.sensor;
    se.type = s.type; se.to_world = xf_from_abi(s.to_world);
    se.width = s.film_width; se.height = s.film_height;
    se.crop_x = s.crop_offset[0]; se.crop_y = s.crop_offset[1]; se.crop_w = s.crop_size[0]; se.crop_h = s.crop_size[1];
    if (se.width <= 0 || se.height <= 0 || se.crop_w <= 0 || se.crop_h <= 0 || se.crop_x < 0 || se.crop_y < 0 ||
        se.crop_x + se.crop_w > se.width || se.crop_y + se.crop_h > se.height) throw std::runtime_error("film: invalid size / crop window");
    se.rfilter = build_rfilter(s.rfilter_type, s.rfilter_radius, s.rfilter_stddev, hs.rfilter_values);
    if (se.rfilter.radius > 16.f) throw std::runtime_error("reconstruction filter radius too large");
    if (s.sample_count <= 0) throw std::runtime_error("sampler: sample_count must be positive");
    se.sample_count = s.sample_count; se.seed = s.sampler_seed; se.medium = s.medium; se.shutter_open_time = s.shutter_open_time;
    se.wavefront = s.sampler_wavefront != 0;
    // (mts_render clamps samples_per_pass to sample_count as integrator.cpp:58-65 does: a larger value is one pass as well)
    if (se.wavefront && d->integrator.samples_per_pass >= 0 && d->integrator.samples_per_pass < s.sample_count)
        throw std::runtime_error("wavefront streams (mts_sensor.sampler_wavefront): samples_per_pass must cover the whole sample_count (one pass)");
    check_index(s.medium, d->medium_count, "sensor medium", true);
    if (s.type == MTS_SENSOR_PERSPECTIVE) {
        se.near_clip = s.near_clip; se.far_clip = s.far_clip;
        float fsx = (float) se.width, fsy = (float) se.height;
        float rel_size_x = (float) se.crop_w / fsx, rel_size_y = (float) se.crop_h / fsy, rel_off_x = (float) se.crop_x / fsx, rel_off_y = (float) se.crop_y / fsy;
        float aspect = fsx / fsy;
        DXf c2s = xf_mul(xf_scale(f3(1.f / rel_size_x, 1.f / rel_size_y, 1.f)),
                  xf_mul(xf_translate(f3(-rel_off_x, -rel_off_y, 0.f)),
                  xf_mul(xf_scale(f3(-0.5f, -0.5f * aspect, 1.f)),
                  xf_mul(xf_translate(f3(-1.f, -1.f / aspect, 0.f)), xf_perspective(s.fov_x, s.near_clip, s.far_clip)))));
        DXf s2c = xf_inverse(c2s);                                                             // perspective.cpp:107-111
        memcpy(se.s2c, s2c.m, 64);
        se.ppo[0] = s.principal_point_offset[0] * ((float) se.width / (float) se.crop_w);      // perspective.cpp:101-106
        se.ppo[1] = s.principal_point_offset[1] * ((float) se.height / (float) se.crop_h);
        se.needs_aperture_sample = 0;                                                          // perspective.cpp:122
    } else if (s.type == MTS_SENSOR_DISTANT || s.type == MTS_SENSOR_DISTANTFLUX) {             // distant.cpp:225-297, distantflux.cpp:141-187
        se.direction_type = (se.width == 1 && se.height == 1) ? 0 : (se.height == 1 ? 1 : 2);
        se.flip_directions = s.distant_flip_directions != 0;
        se.target_type = s.distant_target_type;
        memcpy(se.target_point, s.distant_target_point, 12);
        if (se.target_type == MTS_DISTANT_TARGET_SHAPE) {
            HostScene scratch; DBBox sb; int pc;
            if (s.distant_target_shape.type != MTS_SHAPE_RECTANGLE && s.distant_target_shape.type != MTS_SHAPE_SPHERE && s.distant_target_shape.type != MTS_SHAPE_DISK)
                throw std::runtime_error("distant ray_target shape must be a rectangle, a disk or a sphere in this backend");
            se.target_shape = build_shape(s.distant_target_shape, scratch, sb, pc);
            se.target_area = se.target_shape.type == MTS_SHAPE_DISK ? se.target_shape.surface_area : se.target_shape.type == MTS_SHAPE_RECTANGLE
                ? norm(cross(f3(se.target_shape.frame_s), f3(se.target_shape.frame_t)))
                : 4.f * MTS_PI * se.target_shape.radius * se.target_shape.radius;
        } else if (se.target_type != MTS_DISTANT_TARGET_NONE && se.target_type != MTS_DISTANT_TARGET_POINT)
            throw std::runtime_error("distant sensor: unknown ray_target type");
        if (s.distant_origin_type != 0) {                                                      // distant.cpp:280-289, distantflux.cpp:172-184
            HostScene scratch; DBBox sb; int pc;
            if (s.distant_origin_shape.type != MTS_SHAPE_RECTANGLE && s.distant_origin_shape.type != MTS_SHAPE_SPHERE && s.distant_origin_shape.type != MTS_SHAPE_DISK)
                throw std::runtime_error("distant sensor: the ray origin shape must be a rectangle, a disk or a sphere in this backend");
            se.origin_type = 1;
            se.origin_shape = build_shape(s.distant_origin_shape, scratch, sb, pc);
        }
        store3(se.bsphere_center, center); se.bsphere_radius = bsphere_radius;
        se.needs_aperture_sample = 1;                                                          // endpoint.h:244
    } else if (s.type == MTS_SENSOR_MRADIANCEMETER || s.type == MTS_SENSOR_MDISTANT) {        // mradiancemeter.cpp:72-132, mdistant.cpp:147-203
        if (s.multi_count <= 0 || s.multi_transforms == nullptr) throw std::runtime_error("multi-sensor: no sub-sensors given");
        if (se.width != s.multi_count || se.height != 1) throw std::runtime_error("Film size must be [sensor_count, 1].");
        hs.multi_transforms.assign(s.multi_transforms, s.multi_transforms + 16 * (size_t) s.multi_count);
        se.multi_count = s.multi_count;
        se.needs_aperture_sample = s.type == MTS_SENSOR_MDISTANT ? 1 : 0;                        // m_needs_sample_3
        se.target_type = MTS_DISTANT_TARGET_NONE;
        if (s.type == MTS_SENSOR_MDISTANT) {
            se.target_type = s.distant_target_type;
            memcpy(se.target_point, s.distant_target_point, 12);
            if (se.target_type == MTS_DISTANT_TARGET_SHAPE) {
                HostScene scratch; DBBox sb; int pc;
                if (s.distant_target_shape.type != MTS_SHAPE_RECTANGLE && s.distant_target_shape.type != MTS_SHAPE_SPHERE && s.distant_target_shape.type != MTS_SHAPE_DISK)
                    throw std::runtime_error("mdistant target shape must be a rectangle, a disk or a sphere in this backend");
                se.target_shape = build_shape(s.distant_target_shape, scratch, sb, pc);
                se.target_area = se.target_shape.type == MTS_SHAPE_DISK ? se.target_shape.surface_area : se.target_shape.type == MTS_SHAPE_RECTANGLE
                    ? norm(cross(f3(se.target_shape.frame_s), f3(se.target_shape.frame_t)))
                    : 4.f * MTS_PI * se.target_shape.radius * se.target_shape.radius;
            } else if (se.target_type != MTS_DISTANT_TARGET_NONE && se.target_type != MTS_DISTANT_TARGET_POINT)
                throw std::runtime_error("mdistant sensor: unknown target type");
            store3(se.bsphere_center, center); se.bsphere_radius = bsphere_radius;               // mdistant.cpp:205-210
        }
    } else throw std::runtime_error("unknown sensor type");
    // ---- integrator (integrator.cpp:23-39,302-315)
    const mts_integrator &it = d->integrator;
    if (it.type != MTS_INTEGRATOR_PATH && it.type != MTS_INTEGRATOR_VOLPATH && it.type != MTS_INTEGRATOR_VOLPATHMIS) throw std::runtime_error("unknown integrator type");
    if (it.rr_depth <= 0) throw std::runtime_error("\"rr_depth\" must be set to a value greater than zero!");
    if (it.max_depth < 0 && it.max_depth != -1) throw std::runtime_error("\"max_depth\" must be set to -1 (infinite) or a value >= 0");
    // the regrouping kernel keeps a path's depth in 15 bits of its packed state dword (volpath_flat.h, HotStore::pack): a finite
    // bound beyond that could never trigger, so it is refused instead of being silently treated as infinite
    if (it.max_depth > 32767) throw std::runtime_error("\"max_depth\" must be -1 (infinite) or at most 32767 on this backend");
    sc.integrator.type = it.type; sc.integrator.max_depth = it.max_depth; sc.integrator.rr_depth = it.rr_depth; sc.integrator.hide_emitters = it.hide_emitters != 0;
    sc.integrator.use_spectral_mis = it.use_spectral_mis != 0;
    sc.integrator.monochrome = it.monochrome != 0;
    hs.integrator = it;
    hs.integrator.bin_lo = hs.integrator.bin_hi = nullptr;                                     // copied: the caller's arrays may go away
    // ---- nbins / bins (nbins.cpp:55-98, bins.cpp:23-85) and the sensor's srf (perspective.cpp:113-121, radiancemeter.cpp:62-68)
    sc.srf = -1; sc.bin_mode = 0; sc.bin_count = 0; sc.bin_lo = sc.bin_hi = nullptr; sc.film_channels = 5;
    if (it.bin_mode != 0) {
        if (!spectral) throw std::runtime_error("This integrator can only be used with a spectral variant!");
        if (it.bin_mode != 1 && it.bin_mode != 2) throw std::runtime_error("unknown bin mode");
        if (it.bin_count < 0 || it.bin_count > 64 || (it.bin_count > 0 && (!it.bin_lo || !it.bin_hi))) throw std::runtime_error("bins: between 0 and 64 bins with their bounds");
        hs.bin_lo.assign(it.bin_lo, it.bin_lo + it.bin_count); hs.bin_hi.assign(it.bin_hi, it.bin_hi + it.bin_count);
        if (it.bin_mode == 2)                                                                  // a uniform spectrum per bin (bins.cpp:79-84, uniform.cpp:41-46)
            for (int i = 0; i < it.bin_count; ++i) {
                hs.bin_lo[(size_t) i] = std::max(hs.bin_lo[(size_t) i], 280.f); hs.bin_hi[(size_t) i] = std::min(hs.bin_hi[(size_t) i], 2400.f);
                if (!(hs.bin_lo[(size_t) i] < hs.bin_hi[(size_t) i])) throw std::runtime_error("UniformSpectrum: 'lambda_min' must be less than 'lambda_max'");
            }
        sc.bin_mode = it.bin_mode; sc.bin_count = it.bin_count; sc.film_channels = 5 + 2 * it.bin_count;
    } else hs.integrator.bin_count = 0;
    if (spectral && d->sensor.srf != 0) {
        if (d->sensor.srf < 0 || d->sensor.srf > d->spectrum_count) throw std::runtime_error("index out of range: sensor srf");
        if (d->sensor.type != MTS_SENSOR_PERSPECTIVE && !(d->sensor.type == MTS_SENSOR_MRADIANCEMETER && d->sensor.multi_count == 1))
            throw std::runtime_error("srf: only perspective and radiancemeter sensors sample their wavelengths from a response function");
        const int t = hs.spectra[(size_t) d->sensor.srf - 1].type;
        if (t != MTS_SPECTRUM_UNIFORM && t != MTS_SPECTRUM_DISCRETE) throw std::runtime_error("srf: sample_spectrum is available for uniform and discrete spectra");
        sc.srf = d->sensor.srf - 1;
        if (t == MTS_SPECTRUM_DISCRETE) {                                                      // srf_weights_of (integrator_dev.h) looks a weight up by its wavelength
            const std::vector<float> &w = hs.spectrum_wavelengths[(size_t) sc.srf];
            for (size_t k = 1; k < w.size(); ++k) if (!(w[k] > w[k - 1])) hs.srf_lookup_by_wavelength = false;
        }
    }
    sc.volume_count = (int) hs.volumes.size(); sc.phase_count = (int) hs.phases.size(); sc.medium_count = (int) hs.media.size();
    sc.bsdf_count = (int) hs.bsdfs.size(); sc.shape_count = (int) hs.shapes.size(); sc.prim_count = (int) hs.prims.size();
    sc.emitter_count = (int) hs.emitters.size();
    hs.traits = scene_traits(hs, spectral);
    return hsp.release();
}

// ---------------------------------------------------------------- BVH
// A conservative acceleration structure: every node box is enlarged by a margin far above the rounding error of the slab
// test, so a primitive the exact tests would accept is never culled; the hit that wins is decided by the exact tests and
// the order-independent form of the reference's rule (closest t, ties to the later primitive), so results equal the walk
// over the primitive list bit for bit (which is what the CPU restatement does).
namespace {
struct PrimBox { float lo[3], hi[3], c[3]; int32_t prim; };
struct TmpNode { float lo[3], hi[3]; int left = -1, right = -1, first = 0, count = 0, depth = 0, final_index = -1; };
struct BvhBuilder {
    std::vector<PrimBox> &pb; std::vector<TmpNode> &tmp; std::vector<int32_t> &prims;
    int build(int begin, int end, int depth) {
        const int node = (int) tmp.size();
        tmp.emplace_back();
        float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY }, clo[3], chi[3];
        for (int a = 0; a < 3; ++a) { clo[a] = INFINITY; chi[a] = -INFINITY; }
        for (int i = begin; i < end; ++i) for (int a = 0; a < 3; ++a) {
            lo[a] = std::min(lo[a], pb[i].lo[a]); hi[a] = std::max(hi[a], pb[i].hi[a]);
            clo[a] = std::min(clo[a], pb[i].c[a]); chi[a] = std::max(chi[a], pb[i].c[a]);
        }
        const int count = end - begin;
        int axis = 0;
        for (int a = 1; a < 3; ++a) if (chi[a] - clo[a] > chi[axis] - clo[axis]) axis = a;
        int left = -1, right = -1, first = 0, leaf_count = 0;
        const bool splittable = chi[axis] > clo[axis];
        if (count <= 4 || (!splittable && count <= 7)) {
            first = (int) prims.size(); leaf_count = count;
            for (int i = begin; i < end; ++i) prims.push_back(pb[i].prim);
        } else {
            const int mid = begin + count / 2;
            if (splittable)
                std::nth_element(pb.begin() + begin, pb.begin() + mid, pb.begin() + end,
                                 [axis](const PrimBox &x, const PrimBox &y) { return x.c[axis] < y.c[axis]; });
            left = build(begin, mid, depth + 1); right = build(mid, end, depth + 1);       // coincident centroids: split by index
        }
        TmpNode &n = tmp[(size_t) node];
        for (int a = 0; a < 3; ++a) { n.lo[a] = lo[a]; n.hi[a] = hi[a]; }
        n.left = left; n.right = right; n.first = first; n.count = leaf_count; n.depth = depth;
        return node;
    }
};
// Final layout: the top levels in breadth-first order (they are what the kernels stage in LDS), every subtree below them in
// depth-first order.  A node stores `skip` (where to go when its subtree is missed or done) and `link`: for a leaf
// (first << 3 | count) into bvh_prims, for an inner node minus the index of its left child; the right child is left.skip.
struct BvhLayout {
    std::vector<TmpNode> &tmp; int next = 0;
    void dfs(int n) { tmp[(size_t) n].final_index = next++; if (tmp[(size_t) n].left >= 0) { dfs(tmp[(size_t) n].left); dfs(tmp[(size_t) n].right); } }
    void assign(int root, int top_depth) {
        std::vector<int> level{ root }, below;
        while (!level.empty()) {
            std::vector<int> nxt;
            for (int n : level) {
                tmp[(size_t) n].final_index = next++;
                if (tmp[(size_t) n].left < 0) continue;
                if (tmp[(size_t) n].depth < top_depth) { nxt.push_back(tmp[(size_t) n].left); nxt.push_back(tmp[(size_t) n].right); }
                else { below.push_back(tmp[(size_t) n].left); below.push_back(tmp[(size_t) n].right); }
            }
            level.swap(nxt);
        }
        for (int n : below) dfs(n);
    }
    void emit(int n, int skip, float margin, std::vector<float> &out) {
        const TmpNode &t = tmp[(size_t) n];
        float *o = &out[8 * (size_t) t.final_index];
        for (int a = 0; a < 3; ++a) {
            o[a] = t.lo[a] - margin - 1e-6f * std::fabs(t.lo[a]);
            o[3 + a] = t.hi[a] + margin + 1e-6f * std::fabs(t.hi[a]);
        }
        const int32_t sk = skip;
        const int32_t link = t.left < 0 ? (int32_t) (((uint32_t) t.first << 3) | (uint32_t) t.count) : -(int32_t) tmp[(size_t) t.left].final_index;
        memcpy(&o[6], &sk, 4); memcpy(&o[7], &link, 4);
        if (t.left >= 0) { emit(t.left, tmp[(size_t) t.right].final_index, margin, out); emit(t.right, skip, margin, out); }
    }
};
}

void build_bvh(HostScene &hs) {
    hs.bvh_nodes.clear(); hs.bvh_prims.clear();
    int threshold = 40;                                                    // at or below this the scalar walk over the list is faster (measured: 31 primitives 584 vs 562 Msamples/s)
    if (const char *e = getenv("MTSAMD_BVH_THRESHOLD")) threshold = atoi(e);
    const int n = (int) hs.prims.size();
    if (n <= threshold) return;
    std::vector<PrimBox> pb((size_t) n);
    for (int i = 0; i < n; ++i) {
        const DPrim &pr = hs.prims[(size_t) i]; const DShape &s = hs.shapes[(size_t) pr.shape];
        PrimBox b; b.prim = i;
        for (int a = 0; a < 3; ++a) { b.lo[a] = INFINITY; b.hi[a] = -INFINITY; }
        auto add = [&b](F3 p) { const float v[3] = { p.x, p.y, p.z }; for (int a = 0; a < 3; ++a) { b.lo[a] = std::min(b.lo[a], v[a]); b.hi[a] = std::max(b.hi[a], v[a]); } };
        if (s.type == MTS_SHAPE_RECTANGLE || s.type == MTS_SHAPE_DISK) {
            for (int k = 0; k < 4; ++k) add(mat_point_affine(s.to_world.m, f3((k & 1) ? 1.f : -1.f, (k & 2) ? 1.f : -1.f, 0.f)));
        } else if (s.type == MTS_SHAPE_SPHERE) {
            add(f3(s.center) - f3s(s.radius)); add(f3(s.center) + f3s(s.radius));
        } else {
            const float *t = &hs.tri[9 * (size_t) i];
            F3 p0 = f3(t), e1 = f3(t + 3), e2 = f3(t + 6);
            add(p0); add(p0 + e1); add(p0 + e2);
        }
        for (int a = 0; a < 3; ++a) b.c[a] = .5f * (b.lo[a] + b.hi[a]);
        pb[(size_t) i] = b;
    }
    const F3 diag = f3(hs.scene.bbox.max) - f3(hs.scene.bbox.min);
    std::vector<TmpNode> tmp;
    BvhBuilder bb{ pb, tmp, hs.bvh_prims };
    const int root = bb.build(0, n, 0);
    BvhLayout layout{ tmp };
    layout.assign(root, MTS_BVH_TOP_DEPTH);
    hs.bvh_nodes.assign(8 * tmp.size(), 0.f);
    layout.emit(root, (int) tmp.size(), 1e-5f * norm(diag) + 1e-7f, hs.bvh_nodes);
}

// ---------------------------------------------------------------- upload
#define HIP_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw std::runtime_error(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

template <typename T> static const T *upload(HostScene &hs, const std::vector<T> &v) {
    void *p = nullptr;
    size_t bytes = std::max<size_t>(v.size() * sizeof(T), 16);
    HIP_CHECK(hipMalloc(&p, bytes));
    hs.device_allocs.push_back(p);
    if (!v.empty()) HIP_CHECK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    return (const T *) p;
}

void upload_host_scene(HostScene &hs, int device) {
    HIP_CHECK(hipSetDevice(device));
    hs.device = device;
    for (size_t i = 0; i < hs.volumes.size(); ++i)
        if (hs.volumes[i].type == MTS_VOLUME_GRID) hs.volumes[i].data = upload(hs, hs.grid_data[i]);
    for (size_t i = 0; i < hs.phases.size(); ++i)
        if (hs.phases[i].type == MTS_PHASE_TABULATED) { hs.phases[i].pdf = upload(hs, hs.tab_pdf[i]); hs.phases[i].cdf = upload(hs, hs.tab_cdf[i]); }
    for (size_t i = 0; i < hs.media.size(); ++i)
        if (!hs.pair_data[i].empty()) hs.media[i].pair_grid = upload(hs, hs.pair_data[i]);
    DScene &sc = hs.scene;
    sc.volumes = upload(hs, hs.volumes); sc.phases = upload(hs, hs.phases); sc.media = upload(hs, hs.media);
    sc.bsdfs = upload(hs, hs.bsdfs); sc.shapes = upload(hs, hs.shapes); sc.prims = upload(hs, hs.prims); sc.walk = upload(hs, hs.walk);
    sc.emitters = upload(hs, hs.emitters);
    sc.positions = upload(hs, hs.positions); sc.normals = upload(hs, hs.normals); sc.texcoords = upload(hs, hs.texcoords);
    sc.faces = upload(hs, hs.faces);
    sc.tri = upload(hs, hs.tri);
    sc.tri_attr = upload(hs, hs.tri_attr);
    sc.area_pmf = upload(hs, hs.area_pmf); sc.area_cdf = upload(hs, hs.area_cdf);
    sc.bvh_nodes = nullptr; sc.bvh_prims = nullptr; sc.bvh_node_count = (int32_t) (hs.bvh_nodes.size() / 8);
    sc.bvh_lds = nullptr; sc.bvh_lds_count = 0;
    if (sc.bvh_node_count > 0) { sc.bvh_nodes = upload(hs, hs.bvh_nodes); sc.bvh_prims = upload(hs, hs.bvh_prims); }
    sc.sensor.rfilter.values = upload(hs, hs.rfilter_values);
    sc.sensor.multi = upload(hs, hs.multi_transforms);
    sc.spectra = nullptr; sc.bsdf_sp = nullptr; sc.emitter_sp = nullptr; sc.volume_sp = nullptr; sc.cie = nullptr;
    if (hs.integrator.spectral) {
        for (size_t i = 0; i < hs.spectra.size(); ++i)
            if (hs.spectra[i].type != MTS_SPECTRUM_UNIFORM) {
                hs.spectra[i].values = upload(hs, hs.spectrum_values[i]);
                if (!hs.spectrum_wavelengths[i].empty()) hs.spectra[i].wavelengths = upload(hs, hs.spectrum_wavelengths[i]);
                if (!hs.spectrum_cdf[i].empty()) hs.spectra[i].cdf = upload(hs, hs.spectrum_cdf[i]);
            }
        sc.spectra = upload(hs, hs.spectra); sc.bsdf_sp = upload(hs, hs.bsdf_sp); sc.emitter_sp = upload(hs, hs.emitter_sp);
        sc.volume_sp = upload(hs, hs.volume_sp);
        sc.cie = upload(hs, std::vector<float>(MTS_CIE1931_XYZ, MTS_CIE1931_XYZ + 285));
        if (sc.bin_count > 0) { sc.bin_lo = upload(hs, hs.bin_lo); sc.bin_hi = upload(hs, hs.bin_hi); }
    }
    HIP_CHECK(hipDeviceSynchronize());
    hs.uploaded = true;
}

void free_host_scene(HostScene *hs) {
    if (!hs) return;
    if (hs->uploaded) { (void) hipSetDevice(hs->device); for (void *p : hs->device_allocs) (void) hipFree(p); }
    delete hs;
}

} // namespace mtsamd
