// integration/volpath_amd.cpp -- the reference-side binding of libmtsamd.so: an integrator plugin for the Mitsuba 2 / Eradiate tree.
//
// NOT BUILT IN THIS REPOSITORY: it needs the reference's headers and enoki (absent here).  It is the file a
// maintainer adds as src/integrators/volpath_amd.cpp (+ `add_plugin(volpath_amd volpath_amd.cpp)` and a link against libmtsamd.so in
// src/integrators/CMakeLists.txt).  Written against the interfaces cited line by line below; INTEGRATION.md section 2 explains it.
//
// How the live scene becomes ABI records.  Two sources, used for what each is exact for:
//   (1) TOPOLOGY and everything the plugins expose: typed getters (Scene::shapes / emitters / sensors, scene.h:141-156;
//       Shape::bsdf / emitter / interior_medium / exterior_medium, shape.h:341-348; Medium::phase_function /
//       use_emitter_sampling / has_spectral_extinction, medium.h:72-85; Endpoint::world_transform / shape / medium,
//       endpoint.h:167-202; Sensor::film / sampler / shutter_open_time, sensor.h:73-107; Film::size / crop_size / crop_offset /
//       reconstruction_filter, film.h:70-85; Sampler::sample_count, sampler.h:89) and Object::traverse(TraversalCallback *) with the
//       keys each plugin registers (listed at every visitor).
//   (2) CONSTRUCTOR PARAMETERS the plugins keep private (grid filter / wrap mode, flip_normals, the distant sensor's target, the
//       sampler's seed, ...): the Properties each object was built from.  The reference does not keep them, so the binding adds a
//       recorder to the one place every plugin instance comes from:
//
//         --- src/libcore/plugin.cpp:163-185  PluginManager::create_object
//              ref<Object> object = plugin_class->construct(props);
//         +    record_properties(object.get(), props);          // and the same after class_->construct(props) for "Scene" (:165-166)
//         --- include/mitsuba/core/plugin.h:50
//         +    /// Properties an object was constructed from (nullptr: not created through the plugin manager)
//         +    const Properties *properties_of(const Object *o) const;
//         +    void record_properties(const Object *o, const Properties &p);     // std::unordered_map<const Object *, Properties> + mutex
//         --- src/libcore/xml.cpp:1014,1107 and src/libcore/python/xml_v.cpp:87   (Object::expand(): gridvolume -> GridVolumeImpl, ...)
//         +    for (auto &c : children) PluginManager::instance()->record_properties(c.get(), props);   // an expansion keeps its parent's
//
//       (an entry is overwritten when an address is reused; the table is only read during render()).  Every record of
//       include/mtsamd.h "holds the parameters the plugin reads from its Properties -- same names, defaults and meaning", so each
//       visitor below is the plugin's own constructor read once more.
// Objects reference each other by INDEX in the ABI: every visitor returns the index of the record it appended, one map per record
// array keeps an object from being flattened twice.  Transform4f travels as { matrix, inverse_transpose }, row-major
// (transform.h:36-50).  Colours: in the rgb / mono variants a Texture without spatial variation is evaluated once
// (Texture::eval(si), texture.h:63) -- exactly the Color3f (or luminance) the plugin itself would use; in the spectral variant the
// spectrum plugins' traverse() entries give the representation (uniform.cpp:108-111, regular.cpp:59-61, irregular.cpp:67-69).
// Anything outside the path's plugin list makes flatten() throw; render() then falls back to the stock integrator ON THE REFERENCE
// SIDE (libmtsamd.so itself never falls back).

#include <mitsuba/core/plugin.h>
#include <mitsuba/core/properties.h>
#include <mitsuba/core/rfilter.h>
#include <mitsuba/render/bsdf.h>
#include <mitsuba/render/emitter.h>
#include <mitsuba/render/film.h>
#include <mitsuba/render/imageblock.h>
#include <mitsuba/render/integrator.h>
#include <mitsuba/render/medium.h>
#include <mitsuba/render/mesh.h>
#include <mitsuba/render/phase.h>
#include <mitsuba/render/sampler.h>
#include <mitsuba/render/scene.h>
#include <mitsuba/render/sensor.h>
#include <mitsuba/render/texture.h>
#include <csignal>
#include <deque>
#include <unordered_map>
#include "mtsamd.h"

NAMESPACE_BEGIN(mitsuba)

// ---------------------------------------------------------------------------------------------------------------------------------
// traverse() as a key -> value lookup: the callback interface of include/mitsuba/core/object.h:271-288
class KeyCollector : public TraversalCallback {
public:
    std::unordered_map<std::string, std::pair<void *, const std::type_info *>> params;
    std::unordered_map<std::string, Object *> objects;
    void put_parameter_impl(const std::string &name, const std::type_info &type, void *ptr) override { params[name] = { ptr, &type }; }
    void put_object(const std::string &name, Object *obj) override { objects[name] = obj; }
    template <typename T> const T &get(const std::string &name) const {
        auto it = params.find(name);
        if (it == params.end() || *it->second.second != typeid(T)) Throw("volpath_amd: traverse() entry \"%s\" missing or of another type", name);
        return *(const T *) it->second.first;
    }
    Object *object(const std::string &name) const { auto it = objects.find(name); return it == objects.end() ? nullptr : it->second; }
};

template <typename Float, typename Spectrum>
class SceneFlattener {
public:
    MTS_IMPORT_TYPES(Scene, Sensor, Film, Sampler, ReconstructionFilter, Medium, PhaseFunction, Shape, Mesh, Emitter, BSDF, Texture, Volume)

    // the record arrays (std::deque: records hold pointers into `floats`, which must not move either)
    std::vector<mts_volume> volumes; std::vector<mts_phase> phases; std::vector<mts_medium> media; std::vector<mts_bsdf> bsdfs;
    std::vector<mts_shape> shapes; std::vector<mts_emitter> emitters; std::vector<mts_spectrum> spectra;
    std::deque<std::vector<float>> floats; std::deque<std::vector<uint32_t>> indices;
    mts_sensor sensor_rec;

    mts_scene_desc flatten(const Scene *scene, const Sensor *sensor, const mts_integrator &integrator) {
        if constexpr (is_polarized_v<Spectrum> || !std::is_same_v<scalar_t<Float>, float> || is_array_v<Float>)
            Throw("volpath_amd: scalar single-precision unpolarized variants only");
        for (const auto &s : scene->shapes()) visit_shape(s.get());                 // scene.h:154: declaration order
        for (const auto &e : scene->emitters()) visit_emitter(e.get());             // scene.cpp:31-45: the order the integrators index
        visit_sensor(sensor, scene);
        mts_scene_desc d; std::memset(&d, 0, sizeof(d));
        d.abi_version = MTS_ABI_VERSION;
        d.volumes = volumes.data(); d.volume_count = (int32_t) volumes.size();
        d.phases = phases.data(); d.phase_count = (int32_t) phases.size();
        d.media = media.data(); d.medium_count = (int32_t) media.size();
        d.bsdfs = bsdfs.data(); d.bsdf_count = (int32_t) bsdfs.size();
        d.shapes = shapes.data(); d.shape_count = (int32_t) shapes.size();
        d.emitters = emitters.data(); d.emitter_count = (int32_t) emitters.size();
        d.spectra = spectra.data(); d.spectrum_count = (int32_t) spectra.size();
        d.sensor = sensor_rec; d.integrator = integrator;
        return d;
    }

private:
    std::unordered_map<const Object *, int32_t> m_volume_of, m_phase_of, m_medium_of, m_bsdf_of, m_shape_of, m_emitter_of;

    static const Properties &props_of(const Object *o) {
        const Properties *p = PluginManager::instance()->properties_of(o);
        if (!p) Throw("volpath_amd: %s was not created through the plugin manager", o->class_()->name());
        return *p;
    }
    static mts_transform xf(const ScalarTransform4f &t) {                           // transform.h:36-50: Matrix4f is indexed (row, column)
        mts_transform r;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { r.matrix[4 * i + j] = t.matrix(i, j); r.inverse_transpose[4 * i + j] = t.inverse_transpose(i, j); }
        return r;
    }
    const float *keep(std::vector<float> v) { floats.push_back(std::move(v)); return floats.back().data(); }

    // ---- colours.  rgb / mono: the value the plugin evaluates to (srgb.cpp:28-38: rgb or luminance; uniform.cpp: value; d65 /
    // srgb_d65: the rgb the variant's constructor folded).  spectral: a record of mts_scene_desc.spectra.
    void colour(const Texture *tex, float rgb[3], int32_t &spectrum_index) {
        spectrum_index = -1;
        if constexpr (!is_spectral_v<Spectrum>) {
            SurfaceInteraction3f si = zero<SurfaceInteraction3f>();
            auto v = tex->eval(si, true);                                           // texture.h:63; no plugin of this path varies over si
            if constexpr (is_monochromatic_v<Spectrum>) rgb[0] = rgb[1] = rgb[2] = (float) v[0];
            else { rgb[0] = (float) v[0]; rgb[1] = (float) v[1]; rgb[2] = (float) v[2]; }
        } else {
            KeyCollector k; const_cast<Texture *>(tex)->traverse(&k);
            const std::string cls = props_of(tex).plugin_name();
            mts_spectrum s; std::memset(&s, 0, sizeof(s));
            if (cls == "uniform") {                                                 // uniform.cpp:108-111
                s.type = MTS_SPECTRUM_UNIFORM; s.value = k.get<ScalarFloat>("value");
                s.lambda_min = k.get<ScalarFloat>("lambda_min"); s.lambda_max = k.get<ScalarFloat>("lambda_max");
            } else if (cls == "regular") {                                          // regular.cpp:59-61; a `d65` has expanded into one (d65.cpp:52-65)
                const auto &range = k.get<ScalarVector2f>("range"); const auto &vals = k.get<DynamicBuffer<Float>>("values");
                s.type = MTS_SPECTRUM_REGULAR; s.lambda_min = range.x(); s.lambda_max = range.y();
                s.values = keep(std::vector<float>(vals.data(), vals.data() + vals.size())); s.count = (int32_t) vals.size();
            } else if (cls == "irregular") {                                        // irregular.cpp:67-69
                const auto &wl = k.get<DynamicBuffer<Float>>("wavelengths"); const auto &vals = k.get<DynamicBuffer<Float>>("values");
                s.type = MTS_SPECTRUM_IRREGULAR; s.count = (int32_t) vals.size();
                s.wavelengths = keep(std::vector<float>(wl.data(), wl.data() + wl.size()));
                s.values = keep(std::vector<float>(vals.data(), vals.data() + vals.size()));
            } else
                Throw("volpath_amd: spectrum plugin \"%s\" (rgb colours need the rgb2spec model, which this backend does not carry)", cls);
            spectra.push_back(s); spectrum_index = (int32_t) spectra.size() - 1;
        }
    }

    // ---- volumes: constvolume (constant3d.cpp:39-41: traverse "color" -> Texture), gridvolume (grid3d.cpp:366-370: "data", "size";
    // filter_type / wrap_mode / use_grid_bbox / max_value / to_world from its Properties, :43-61,152-160; texture.cpp:89-92),
    // gridvolume_spectral (gridvolume_spectral.cpp:388-392 + "lambda_min" / "lambda_max", :186-190)
    int32_t visit_volume(const Volume *v) {
        if (auto it = m_volume_of.find(v); it != m_volume_of.end()) return it->second;
        const Properties &p = props_of(v);
        mts_volume r; std::memset(&r, 0, sizeof(r)); r.value_spectrum = -1;
        r.to_world = xf(p.transform("to_world", ScalarTransform4f()));
        KeyCollector k; const_cast<Volume *>(v)->traverse(&k);
        if (p.plugin_name() == "constvolume") {
            r.type = MTS_VOLUME_CONST;
            colour((const Texture *) k.object("color"), r.value, r.value_spectrum);
        } else if (p.plugin_name() == "gridvolume" || p.plugin_name() == "gridvolume_spectral") {
            const bool spectral_grid = p.plugin_name() == "gridvolume_spectral";
            r.type = spectral_grid ? MTS_VOLUME_GRID_SPECTRAL : MTS_VOLUME_GRID;
            const auto &data = k.get<DynamicBuffer<Float>>("data");
            const ScalarVector3i res = v->resolution();                             // texture.h:244: (nx, ny, nz)
            r.nx = res.x(); r.ny = res.y(); r.nz = res.z();
            r.channels = (int32_t) (data.size() / ((size_t) r.nx * r.ny * r.nz));
            r.data = keep(std::vector<float>(data.data(), data.data() + data.size()));
            const std::string ft = p.string("filter_type", "trilinear"), wm = p.string("wrap_mode", "clamp");
            r.filter_type = ft == "nearest" ? MTS_FILTER_NEAREST : MTS_FILTER_TRILINEAR;
            r.wrap_mode = wm == "repeat" ? MTS_WRAP_REPEAT : wm == "mirror" ? MTS_WRAP_MIRROR : MTS_WRAP_CLAMP;
            // use_grid_bbox (grid3d.cpp:152-155) composes the file's bounding box into world_to_local; the live object has done
            // that already and publishes the result as its bbox(): hand the ABI the final placement instead of the two pieces
            r.use_grid_bbox = 0;
            if (p.bool_("use_grid_bbox", false)) {
                const ScalarBoundingBox3f b = v->bbox();                            // texture.h:236 (update_bbox, :262-269)
                r.to_world = xf(ScalarTransform4f::translate(ScalarVector3f(b.min)) * ScalarTransform4f::scale(b.extents()));
            }
            if (p.has_property("max_value")) { r.has_max_value = 1; r.max_value = p.float_("max_value"); }
            if (spectral_grid) { r.lambda_min = p.float_("lambda_min"); r.lambda_max = p.float_("lambda_max"); }
        } else
            Throw("volpath_amd: volume plugin \"%s\"", p.plugin_name());
        volumes.push_back(r); return m_volume_of[v] = (int32_t) volumes.size() - 1;
    }

    // ---- phase functions: hg (hg.cpp:86-88 "g"), rayleigh / isotropic (no parameters), tabphase (tabphase.cpp:97-99 "values": the
    // pdf on a regular cos(theta) grid over [-1, 1]), blendphase (blendphase.cpp:141-145: "weight" volume, "phase_0", "phase_1")
    int32_t visit_phase(const PhaseFunction *ph) {
        if (auto it = m_phase_of.find(ph); it != m_phase_of.end()) return it->second;
        const std::string cls = props_of(ph).plugin_name();
        mts_phase r; std::memset(&r, 0, sizeof(r)); r.child[0] = r.child[1] = r.weight_volume = -1;
        KeyCollector k; const_cast<PhaseFunction *>(ph)->traverse(&k);
        if (cls == "isotropic") r.type = MTS_PHASE_ISOTROPIC;
        else if (cls == "rayleigh") r.type = MTS_PHASE_RAYLEIGH;
        else if (cls == "hg") { r.type = MTS_PHASE_HG; r.g = k.get<ScalarFloat>("g"); }
        else if (cls == "tabphase") {
            const auto &vals = k.get<DynamicBuffer<Float>>("values");
            r.type = MTS_PHASE_TABULATED; r.tab_count = (int32_t) vals.size();
            r.tab_values = keep(std::vector<float>(vals.data(), vals.data() + vals.size()));
        } else if (cls == "blendphase") {                                           // children first: the ABI wants them before the parent
            r.type = MTS_PHASE_BLEND;
            r.child[0] = visit_phase((const PhaseFunction *) k.object("phase_0"));
            r.child[1] = visit_phase((const PhaseFunction *) k.object("phase_1"));
            r.weight_volume = visit_volume((const Volume *) k.object("weight"));
        } else
            Throw("volpath_amd: phase function plugin \"%s\"", cls);
        phases.push_back(r); return m_phase_of[ph] = (int32_t) phases.size() - 1;
    }

    // ---- media: homogeneous / heterogeneous (homogeneous.cpp:56-61, heterogeneous.cpp:56-61: "scale", "albedo", "sigma_t";
    // medium.h:72-85: phase_function(), use_emitter_sampling(), has_spectral_extinction(), is_homogeneous())
    int32_t visit_medium(const Medium *m) {
        if (!m) return -1;
        if (auto it = m_medium_of.find(m); it != m_medium_of.end()) return it->second;
        KeyCollector k; const_cast<Medium *>(m)->traverse(&k);
        mts_medium r; std::memset(&r, 0, sizeof(r));
        r.type = m->is_homogeneous() ? MTS_MEDIUM_HOMOGENEOUS : MTS_MEDIUM_HETEROGENEOUS;
        r.sigma_t_volume = visit_volume((const Volume *) k.object("sigma_t"));
        r.albedo_volume = visit_volume((const Volume *) k.object("albedo"));
        r.scale = k.get<ScalarFloat>("scale");
        r.phase = visit_phase(m->phase_function());
        r.sample_emitters = m->use_emitter_sampling() ? 1 : 0;
        r.has_spectral_extinction = m->has_spectral_extinction() ? 1 : 0;
        media.push_back(r); return m_medium_of[m] = (int32_t) media.size() - 1;
    }

    // ---- BSDFs: diffuse (diffuse.cpp:137-139 "reflectance"), null, rpv (rpv.cpp:169-174 "rho_0", "g", "k", "rho_c"), bilambertian
    // (bilambertian.cpp:195-198 "reflectance", "transmittance")
    int32_t visit_bsdf(const BSDF *b) {
        if (auto it = m_bsdf_of.find(b); it != m_bsdf_of.end()) return it->second;
        const std::string cls = props_of(b).plugin_name();
        KeyCollector k; const_cast<BSDF *>(b)->traverse(&k);
        mts_bsdf r; std::memset(&r, 0, sizeof(r)); for (int32_t &s : r.spectrum) s = -1;
        auto tex = [&](const char *name) { return (const Texture *) k.object(name); };
        if (cls == "diffuse") { r.type = MTS_BSDF_DIFFUSE; colour(tex("reflectance"), r.reflectance, r.spectrum[0]); }
        else if (cls == "null") r.type = MTS_BSDF_NULL;
        else if (cls == "rpv") {
            r.type = MTS_BSDF_RPV;
            colour(tex("rho_0"), r.rho_0, r.spectrum[1]); colour(tex("k"), r.k, r.spectrum[2]);
            colour(tex("g"), r.g, r.spectrum[3]); colour(tex("rho_c"), r.rho_c, r.spectrum[4]);
        } else if (cls == "bilambertian") {
            r.type = MTS_BSDF_BILAMBERTIAN;
            colour(tex("reflectance"), r.reflectance, r.spectrum[0]); colour(tex("transmittance"), r.transmittance, r.spectrum[5]);
        } else
            Throw("volpath_amd: BSDF plugin \"%s\"", cls);
        bsdfs.push_back(r); return m_bsdf_of[b] = (int32_t) bsdfs.size() - 1;
    }

    // ---- shapes: Shape::traverse (shape.cpp:401-413: "to_world" + the child objects) and the getters of shape.h:341-348;
    // rectangle / disk ("flip_normals", rectangle.cpp:60, disk.cpp:60), sphere ("center", "radius", "flip_normals", sphere.cpp:95-98:
    // the constructor folds centre and radius into to_world -- the ABI takes the three as given), cube and every other Mesh
    // (mesh.cpp:835-847: "vertex_count", "face_count", "faces_buf", "vertex_positions_buf", "vertex_normals_buf",
    // "vertex_texcoords_buf": WORLD-space vertices, mesh.h:195-226 -> to_world = identity)
    mts_shape shape_record(const Shape *s) {
        const Properties &p = props_of(s);
        const std::string cls = p.plugin_name();
        mts_shape r; std::memset(&r, 0, sizeof(r));
        r.bsdf = r.interior_medium = r.exterior_medium = r.emitter = -1;
        r.to_world = xf(p.transform("to_world", ScalarTransform4f()));
        if (cls == "rectangle") { r.type = MTS_SHAPE_RECTANGLE; r.flip_normals = p.bool_("flip_normals", false); }
        else if (cls == "disk") { r.type = MTS_SHAPE_DISK; r.flip_normals = p.bool_("flip_normals", false); }
        else if (cls == "sphere") {
            r.type = MTS_SHAPE_SPHERE; r.flip_normals = p.bool_("flip_normals", false);
            const ScalarPoint3f c = p.point3f("center", ScalarPoint3f(0.f)); r.center[0] = c.x(); r.center[1] = c.y(); r.center[2] = c.z();
            r.radius = p.float_("radius", 1.f);
        } else if (s->is_mesh()) {                                                  // cube, obj, ply, serialized
            const Mesh *mesh = (const Mesh *) s;
            KeyCollector k; const_cast<Mesh *>(mesh)->traverse(&k);
            r.type = cls == "cube" ? MTS_SHAPE_CUBE : MTS_SHAPE_MESH;
            if (r.type == MTS_SHAPE_MESH) {
                const auto &pos = k.get<DynamicBuffer<Float>>("vertex_positions_buf"); const auto &nor = k.get<DynamicBuffer<Float>>("vertex_normals_buf");
                const auto &uv = k.get<DynamicBuffer<Float>>("vertex_texcoords_buf"); const auto &fc = k.get<DynamicBuffer<UInt32>>("faces_buf");
                r.vertex_count = (int32_t) mesh->vertex_count(); r.face_count = (int32_t) mesh->face_count();
                r.vertex_positions = keep(std::vector<float>(pos.data(), pos.data() + 3 * r.vertex_count));
                r.vertex_normals = mesh->has_vertex_normals() ? keep(std::vector<float>(nor.data(), nor.data() + 3 * r.vertex_count)) : nullptr;
                r.vertex_texcoords = mesh->has_vertex_texcoords() ? keep(std::vector<float>(uv.data(), uv.data() + 2 * r.vertex_count)) : nullptr;
                indices.emplace_back(fc.data(), fc.data() + 3 * r.face_count); r.faces = indices.back().data();
                r.to_world = xf(ScalarTransform4f());                               // the buffers are in world space already
            }
        } else
            Throw("volpath_amd: shape plugin \"%s\"", cls);
        return r;
    }
    int32_t visit_shape(const Shape *s) {
        if (auto it = m_shape_of.find(s); it != m_shape_of.end()) return it->second;
        mts_shape r = shape_record(s);
        r.bsdf = visit_bsdf(s->bsdf());
        r.interior_medium = visit_medium(s->interior_medium());
        r.exterior_medium = visit_medium(s->exterior_medium());
        shapes.push_back(r);
        const int32_t index = m_shape_of[s] = (int32_t) shapes.size() - 1;
        // r.emitter is patched by visit_emitter (the emitter array follows Scene::emitters(), not the shapes)
        return index;
    }

    // ---- emitters: directional (directional.cpp:149-151 "irradiance"; direction = to_world * +z, :47-63 folds a "direction" property
    // into to_world), area (area.cpp:191-193 "radiance"; Endpoint::shape()), constant (constant.cpp:119-121 "radiance"), point
    // (point.cpp:123-125 "intensity"; position in to_world).  Endpoint::world_transform(), endpoint.h:167.
    int32_t visit_emitter(const Emitter *e) {
        if (auto it = m_emitter_of.find(e); it != m_emitter_of.end()) return it->second;
        const std::string cls = props_of(e).plugin_name();
        KeyCollector k; const_cast<Emitter *>(e)->traverse(&k);
        mts_emitter r; std::memset(&r, 0, sizeof(r)); r.shape = -1; r.radiance_spectrum = -1;
        r.to_world = xf(e->world_transform()->eval(0.f));
        const char *key = nullptr;
        if (cls == "directional") { r.type = MTS_EMITTER_DIRECTIONAL; key = "irradiance"; }
        else if (cls == "area") { r.type = MTS_EMITTER_AREA; key = "radiance"; r.shape = visit_shape(e->shape()); }
        else if (cls == "constant") { r.type = MTS_EMITTER_CONSTANT; key = "radiance"; }
        else if (cls == "point") { r.type = MTS_EMITTER_POINT; key = "intensity"; }
        else Throw("volpath_amd: emitter plugin \"%s\"", cls);
        colour((const Texture *) k.object(key), r.radiance, r.radiance_spectrum);
        emitters.push_back(r);
        const int32_t index = m_emitter_of[e] = (int32_t) emitters.size() - 1;
        if (r.shape >= 0) shapes[(size_t) r.shape].emitter = index;
        return index;
    }

    // ---- sensor + film + sampler: perspective (perspective.cpp:325-328 "x_fov"; near_clip / far_clip / principal_point_offset_*,
    // sensor.cpp:95-96, perspective.cpp:101-104), distant (distant.cpp:225-290: nothing exposed -> Properties), the Eradiate
    // multi-sensors likewise; hdrfilm + rfilter (film.h:70-85, rfilter.h:53), independent (sampler.h:89; "seed" from Properties)
    void visit_sensor(const Sensor *sensor, const Scene *scene) {
        const Properties &p = props_of(sensor);
        const std::string cls = p.plugin_name();
        mts_sensor &r = sensor_rec; std::memset(&r, 0, sizeof(r));
        r.to_world = xf(sensor->world_transform()->eval(0.f));
        r.medium = visit_medium(sensor->medium());
        r.shutter_open_time = sensor->shutter_open_time();
        if (cls == "perspective") {
            KeyCollector k; const_cast<Sensor *>(sensor)->traverse(&k);
            r.type = MTS_SENSOR_PERSPECTIVE; r.fov_x = k.get<ScalarFloat>("x_fov");
            r.near_clip = p.float_("near_clip", 1e-2f); r.far_clip = p.float_("far_clip", 1e4f);
            r.principal_point_offset[0] = p.float_("principal_point_offset_x", 0.f); r.principal_point_offset[1] = p.float_("principal_point_offset_y", 0.f);
        } else if (cls == "distant") {
            r.type = MTS_SENSOR_DISTANT; r.distant_flip_directions = p.bool_("flip_directions", false);
            if (p.has_property("ray_target")) {                                     // distant.cpp:246-272: a point or a nested shape
                if (p.type("ray_target") == Properties::Type::Array3f) {
                    const ScalarPoint3f t = p.point3f("ray_target"); r.distant_target_type = MTS_DISTANT_TARGET_POINT;
                    r.distant_target_point[0] = t.x(); r.distant_target_point[1] = t.y(); r.distant_target_point[2] = t.z();
                } else {
                    r.distant_target_type = MTS_DISTANT_TARGET_SHAPE;
                    r.distant_target_shape = shape_record((const Shape *) p.object("ray_target").get());
                }
            }
            if (p.has_property("ray_origin")) {                                     // distant.cpp:126-130,280-289
                r.distant_origin_type = 1; r.distant_origin_shape = shape_record((const Shape *) p.object("ray_origin").get());
            }
        } else
            Throw("volpath_amd: sensor plugin \"%s\" (mradiancemeter / mdistant / distantflux: multi_transforms from \"origins\" / \"directions\" as "
                  "mradiancemeter.cpp:95-113 and mdistant.cpp:160-175 build them)", cls);
        const Film *film = sensor->film();
        r.film_width = film->size().x(); r.film_height = film->size().y();
        r.crop_offset[0] = film->crop_offset().x(); r.crop_offset[1] = film->crop_offset().y();
        r.crop_size[0] = film->crop_size().x(); r.crop_size[1] = film->crop_size().y();
        const ReconstructionFilter *rf = film->reconstruction_filter();
        const Properties &rp = props_of(rf);
        if (rp.plugin_name() == "box") { r.rfilter_type = MTS_RFILTER_BOX; r.rfilter_radius = rf->radius(); }
        else if (rp.plugin_name() == "gaussian") { r.rfilter_type = MTS_RFILTER_GAUSSIAN; r.rfilter_stddev = rp.float_("stddev", .5f); }
        else Throw("volpath_amd: reconstruction filter \"%s\"", rp.plugin_name());
        const Sampler *sampler = sensor->sampler();
        const Properties &sp = props_of(sampler);
        if (sp.plugin_name() != "independent") Throw("volpath_amd: sampler \"%s\"", sp.plugin_name());
        r.sample_count = (int32_t) sampler->sample_count();
        r.sampler_seed = (uint64_t) sp.size_("seed", 0);                            // sampler.cpp:14-18
        r.sampler_wavefront = 0;                                                    // the scalar variants' streams (integrator.cpp:198)
        (void) scene;
    }
};

// ---------------------------------------------------------------------------------------------------------------------------------
template <typename Float, typename Spectrum>
class VolpathAmdIntegrator final : public Integrator<Float, Spectrum> {
public:
    MTS_IMPORT_BASE(Integrator)
    MTS_IMPORT_TYPES(Scene, Sensor, Film, ImageBlock)

    VolpathAmdIntegrator(const Properties &props) : Base(props), m_fallback_props(props) {
        std::memset(&m_desc, 0, sizeof(m_desc));
        const std::string wrapped = props.string("integrator", "volpath");         // which of the three this instance stands in for
        m_desc.type             = wrapped == "path" ? MTS_INTEGRATOR_PATH : wrapped == "volpathmis" ? MTS_INTEGRATOR_VOLPATHMIS : MTS_INTEGRATOR_VOLPATH;
        m_desc.max_depth        = props.int_("max_depth", -1);                      // integrator.cpp:302-315
        m_desc.rr_depth         = props.int_("rr_depth", 5);
        m_desc.hide_emitters    = props.bool_("hide_emitters", false);
        m_desc.block_size       = (int32_t) props.size_("block_size", 0);           // integrator.cpp:23-39; 0 -> 32 in the backend
        m_desc.samples_per_pass = (int32_t) props.size_("samples_per_pass", (size_t) -1);
        m_desc.timeout          = props.float_("timeout", -1.f);
        m_desc.use_spectral_mis = props.bool_("use_spectral_mis", true);            // volpathmis.cpp:29,38
        m_desc.monochrome       = is_monochromatic_v<Spectrum> ? 1 : 0;
        m_desc.spectral         = is_spectral_v<Spectrum> ? 1 : 0;
        m_fallback_props.set_plugin_name(wrapped);
    }

    bool render(Scene *scene, Sensor *sensor) override {
        SceneFlattener<Float, Spectrum> flat;
        mts_scene_desc d;
        try {
            d = flat.flatten(scene, sensor, m_desc);
        } catch (const std::exception &e) {                                         // a plugin outside the path: the stock integrator renders
            Log(Warn, "%s -- rendering with the stock \"%s\" integrator", e.what(), m_fallback_props.plugin_name());
            ref<Base> stock = PluginManager::instance()->create_object<Base>(m_fallback_props);
            return stock->render(scene, sensor);
        }
        mts_scene *h = nullptr;
        if (mts_scene_create(&d, /*device*/ 0, &h)) Throw("volpath_amd: %s", mts_last_error());
        m_scene = h;

        ref<Film> film = sensor->film();
        const ScalarVector2i size = film->crop_size();
        std::vector<float> xyzaw((size_t) size.x() * size.y() * 5);
        mts_render_opts opts; std::memset(&opts, 0, sizeof(opts));
        opts.shard_count = 1; opts.film_capacity = (int64_t) xyzaw.size();
        mts_stats stats;
        // Ctrl-C: the scope of integrator_v.cpp:129-151 lives behind the ABI (async-signal-safe handler, previous handler restored and
        // re-raised); the Python binding of the reference keeps its own around this call, which then finds the render already over
        const bool scoped = mts_sigint_scope_enter(h) == 0;
        const int rc = mts_render(h, &opts, xyzaw.data(), &stats);
        if (scoped) mts_sigint_scope_exit();
        m_scene = nullptr;
        mts_scene_destroy(h);
        if (rc) Throw("volpath_amd: %s", mts_last_error());

        // hand the raw XYZAW storage to the film exactly as render_block's film->put(block) does (integrator.cpp:134,
        // hdrfilm.cpp:190-199): one block covering the crop window, no border
        film->prepare({ "X", "Y", "Z", "A", "W" });
        ref<ImageBlock> block = new ImageBlock(size, 5, nullptr, false, false, false, false);
        block->set_offset(film->crop_offset());
        std::memcpy(block->data().data(), xyzaw.data(), xyzaw.size() * sizeof(float));
        film->put(block);
        return !stats.cancelled;                                                    // a timeout alone is not a cancellation (integrator.cpp:178)
    }

    void cancel() override { if (mts_scene *h = m_scene.load()) mts_cancel(h); }   // integrator.cpp:43-45

    MTS_DECLARE_CLASS()
private:
    mts_integrator m_desc;
    Properties m_fallback_props;
    std::atomic<mts_scene *> m_scene { nullptr };
};

MTS_IMPLEMENT_CLASS_VARIANT(VolpathAmdIntegrator, Integrator)
MTS_EXPORT_PLUGIN(VolpathAmdIntegrator, "Volumetric path tracer (MI355X backend)");
NAMESPACE_END(mitsuba)
