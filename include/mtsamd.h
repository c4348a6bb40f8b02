/* mtsamd.h -- C ABI of libmtsamd.so, the MI355X-native path / volpath render backend.
 *
 * This is the drop-in boundary.  The reference's interface for this hot
 * path is C++ virtual dispatch on enoki-typed plugins plus a pybind11 surface:
 *
 *   plugin construction  T(const Properties&)            include/mitsuba/core/class.h:195-211,
 *                                                         src/libcore/plugin.cpp:163-185
 *   Scene(const Properties&)                              src/librender/scene.cpp:22-104
 *   bool Integrator::render(Scene*, Sensor*)              include/mitsuba/render/integrator.h:42
 *   void Integrator::cancel()                             include/mitsuba/render/integrator.h:51
 *   SamplingIntegrator::sample(scene, sampler, ray, ...)  include/mitsuba/render/integrator.h:114-119
 *   Film::bitmap(raw)  -> XYZAW float image               src/films/hdrfilm.cpp:251-259
 *   pybind: Integrator.render / load_dict                 src/librender/python/integrator_v.cpp:124-156,
 *                                                         src/libcore/python/xml_v.cpp:100-272
 *
 * An enoki-typed vtable cannot be satisfied without enoki, so the replacement is a plain C ABI:
 * every plugin instance becomes one POD record holding exactly the parameters the reference plugin
 * reads from its Properties (same names, same defaults, same meaning), objects reference each other
 * by index instead of by ref<Object>, and a Transform4f travels the way the reference stores it:
 * matrix + inverse transpose (include/mitsuba/core/transform.h:36-50).
 *
 * Conventions: all functions return 0 on success, non-zero on error (mts_last_error() gives a
 * thread-local message; the reference throws std::runtime_error via Throw(), e.g.
 * src/librender/integrator.cpp:62-63).  Caller owns every buffer it passes in; the library copies
 * what it needs in mts_scene_create.  Handles are opaque.  mts_render is blocking; mts_cancel may be
 * called from another thread (mirrors m_stop, src/librender/integrator.cpp:43-45).
 */
#ifndef MTSAMD_H
#define MTSAMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MTS_ABI_VERSION 9

/* Transform4f: row-major 4x4 matrix and its inverse transpose (transform.h:36-50). */
typedef struct mts_transform {
    float matrix[16];
    float inverse_transpose[16];
} mts_transform;

/* ---- Spectra (spectral variant only: mts_integrator.spectral): src/spectra/uniform.cpp (value inside [lambda_min, lambda_max], 0
 *      outside; the bounds default to, and are clamped to, MTS_WAVELENGTH_MIN / MAX = 280 / 2400 nm, core/spectrum.h:15-21) and
 *      src/spectra/regular.cpp (values at regularly spaced wavelengths over [lambda_min, lambda_max], linearly interpolated, 0 outside).
 *      `d65` (src/spectra/d65.cpp) expands to `regular` on the caller's side, as the plugin itself does. ---- */
/*      `irregular` (src/spectra/irregular.cpp: values at arbitrary increasing wavelengths, linearly interpolated, 0 outside) and
 *      `discrete` (src/spectra/discrete.cpp: a sensor response that SAMPLES one of its wavelengths with probability ~ pmf and
 *      weights it with the matching value; it evaluates to 0). */
enum { MTS_SPECTRUM_UNIFORM = 0, MTS_SPECTRUM_REGULAR = 1, MTS_SPECTRUM_IRREGULAR = 2, MTS_SPECTRUM_DISCRETE = 3 };
typedef struct mts_spectrum {
    int32_t type;
    float value;              /* uniform "value" */
    float lambda_min, lambda_max;
    const float *values;      /* regular / irregular / discrete "values" */
    int32_t count;
    const float *wavelengths; /* irregular / discrete "wavelengths" (count entries)              */
    const float *pmf;         /* discrete "pmf" (count entries)                                   */
} mts_spectrum;

/* ---- Volume (3-D texture): constvolume (src/textures/constant3d.cpp) / gridvolume (grid3d.cpp) /
 *      gridvolume_spectral (src/textures/gridvolume_spectral.cpp, spectral variant only) ---- */
enum { MTS_VOLUME_CONST = 0, MTS_VOLUME_GRID = 1, MTS_VOLUME_GRID_SPECTRAL = 2 };
enum { MTS_FILTER_NEAREST = 0, MTS_FILTER_TRILINEAR = 1 };            /* grid3d.cpp:43-50 */
enum { MTS_WRAP_REPEAT = 0, MTS_WRAP_MIRROR = 1, MTS_WRAP_CLAMP = 2 }; /* grid3d.cpp:52-61 */
typedef struct mts_volume {
    int32_t type;
    float value[3];           /* constvolume "value" (rgb; gray values replicated)               */
    mts_transform to_world;   /* Volume::to_world (render/texture.h:210-279); unit cube -> world */
    /* gridvolume only: the decoded contents of the .vol file (volume_data.h:42-102) */
    const float *data;        /* nz*ny*nx*channels floats, x fastest (grid3d.cpp:30-34)         */
    int32_t nx, ny, nz, channels;   /* channels: 1 or 3                                          */
    int32_t filter_type;      /* "filter_type", default trilinear                                */
    int32_t wrap_mode;        /* "wrap_mode", default clamp                                      */
    int32_t use_grid_bbox;    /* "use_grid_bbox" (grid3d.cpp:152-155)                            */
    float file_bbox_min[3], file_bbox_max[3]; /* bbox stored in the .vol header                  */
    int32_t has_max_value;    /* "max_value" override (grid3d.cpp:157-160)                       */
    float max_value;
    /* spectral variant */
    int32_t value_spectrum;   /* constvolume: index into mts_scene_desc.spectra of its colour (-1 in rgb / mono scenes)  */
    float lambda_min, lambda_max;   /* gridvolume_spectral: the interval its `channels` nodes cover (:186-190); any channel count >= 2 */
} mts_volume;

/* ---- Phase functions (src/phase/{isotropic,hg,rayleigh,blendphase,tabphase}.cpp) ---- */
enum { MTS_PHASE_ISOTROPIC = 0, MTS_PHASE_HG = 1, MTS_PHASE_RAYLEIGH = 2,
       MTS_PHASE_BLEND = 3, MTS_PHASE_TABULATED = 4 };
typedef struct mts_phase {
    int32_t type;
    float g;                  /* hg "g", default 0.8 (hg.cpp:44)                                 */
    int32_t child[2];         /* blendphase nested phase functions, in declaration order         */
    int32_t weight_volume;    /* blendphase "weight" volume index                                */
    const float *tab_values;  /* tabphase "values" on a regular cos(theta) grid over [-1, 1]     */
    int32_t tab_count;
} mts_phase;

/* ---- Media (src/media/{homogeneous,heterogeneous}.cpp, src/librender/medium.cpp) ---- */
enum { MTS_MEDIUM_HOMOGENEOUS = 0, MTS_MEDIUM_HETEROGENEOUS = 1 };
typedef struct mts_medium {
    int32_t type;
    int32_t sigma_t_volume;   /* "sigma_t", default constvolume 1                                */
    int32_t albedo_volume;    /* "albedo", default constvolume 0.75                              */
    float scale;              /* "scale", default 1                                              */
    int32_t phase;            /* nested phase function, default isotropic (medium.cpp:23-27)     */
    int32_t sample_emitters;  /* "sample_emitters", default true (medium.cpp:28)                 */
    int32_t has_spectral_extinction; /* default true (homogeneous.cpp:27, heterogeneous.cpp:27) */
} mts_medium;

/* ---- BSDFs (src/bsdfs/{diffuse,null,rpv,bilambertian}.cpp) ---- */
enum { MTS_BSDF_DIFFUSE = 0, MTS_BSDF_NULL = 1, MTS_BSDF_RPV = 2, MTS_BSDF_BILAMBERTIAN = 3 };
typedef struct mts_bsdf {
    int32_t type;
    float reflectance[3];     /* diffuse "reflectance" (rgb), default 0.5                        */
    float rho_0[3], k[3], g[3], rho_c[3];  /* rpv parameters (rpv.cpp:60-70)                     */
    float transmittance[3];   /* bilambertian "transmittance" (with "reflectance"), default 0.5 (bilambertian.cpp:52-53) */
    int32_t spectrum[6];      /* spectral variant: indices into mts_scene_desc.spectra of reflectance, rho_0, k, g, rho_c,
                                 transmittance, in this order (-1 in rgb / mono scenes)                                   */
} mts_bsdf;

/* ---- Shapes (src/shapes/{rectangle,cube,sphere,disk}.cpp, src/librender/mesh.cpp) ---- */
enum { MTS_SHAPE_RECTANGLE = 0, MTS_SHAPE_CUBE = 1, MTS_SHAPE_SPHERE = 2, MTS_SHAPE_MESH = 3, MTS_SHAPE_DISK = 4 /* src/shapes/disk.cpp */ };
typedef struct mts_shape {
    int32_t type;
    mts_transform to_world;
    int32_t flip_normals;     /* rectangle / sphere "flip_normals"                               */
    float center[3];          /* sphere "center" (default 0) and "radius" (default 1)            */
    float radius;
    /* MTS_SHAPE_MESH: object-space triangle soup (positions mandatory; normals / uvs optional) */
    const float *vertex_positions;   /* 3 * vertex_count */
    const float *vertex_normals;     /* 3 * vertex_count or NULL */
    const float *vertex_texcoords;   /* 2 * vertex_count or NULL */
    const uint32_t *faces;           /* 3 * face_count */
    int32_t vertex_count, face_count;
    int32_t bsdf;             /* index into bsdfs, -1 = default diffuse (shape.cpp:77-79)        */
    int32_t interior_medium;  /* child "interior" (shape.cpp:58-68), -1 = none                   */
    int32_t exterior_medium;  /* child "exterior", -1 = none                                     */
    int32_t emitter;          /* index of the attached area emitter, -1 = none                   */
} mts_shape;

/* ---- Emitters (src/emitters/{directional,area,constant}.cpp) ---- */
enum { MTS_EMITTER_DIRECTIONAL = 0, MTS_EMITTER_AREA = 1, MTS_EMITTER_CONSTANT = 2,
       MTS_EMITTER_POINT = 3 /* src/emitters/point.cpp: position = translation of to_world, `radiance` holds the intensity */ };
typedef struct mts_emitter {
    int32_t type;
    mts_transform to_world;   /* directional: local +z is the direction of propagation           */
    float radiance[3];        /* "irradiance" (directional) / "radiance" (area, constant), rgb   */
    int32_t shape;            /* area: index of the owning shape                                 */
    int32_t radiance_spectrum; /* spectral variant: index into mts_scene_desc.spectra (-1 in rgb / mono scenes)           */
} mts_emitter;

/* ---- Sensor + film + sampler (src/sensors/{perspective,distant}.cpp, src/films/hdrfilm.cpp,
 *      src/samplers/independent.cpp, src/rfilters/{box,gaussian}.cpp) ---- */
enum { MTS_SENSOR_PERSPECTIVE = 0, MTS_SENSOR_DISTANT = 1,
       MTS_SENSOR_MRADIANCEMETER = 2 /* src/sensors/mradiancemeter.cpp */, MTS_SENSOR_MDISTANT = 3 /* src/sensors/mdistant.cpp */,
       MTS_SENSOR_DISTANTFLUX = 4 /* src/sensors/distantflux.cpp: to_world + distant_target_*; "origin" shapes are not supported */ };
enum { MTS_RFILTER_BOX = 0, MTS_RFILTER_GAUSSIAN = 1 };
enum { MTS_DISTANT_TARGET_NONE = 0, MTS_DISTANT_TARGET_POINT = 1, MTS_DISTANT_TARGET_SHAPE = 2 };
typedef struct mts_sensor {
    int32_t type;
    mts_transform to_world;
    float fov_x;              /* perspective: horizontal field of view in degrees (parse_fov,
                                 src/librender/sensor.cpp:113-167, already resolved to the x axis) */
    float near_clip, far_clip;       /* defaults 1e-2 / 1e4 (sensor.cpp:95-96)                   */
    float principal_point_offset[2];
    /* distant (src/sensors/distant.cpp:225-290).  The direction mode follows the film size
     * (1x1: single direction, Nx1: directions in the local xz plane, NxM: hemisphere); a "direction"
     * parameter is folded into to_world by the caller exactly as the constructor does (look_at).
     * Ray origins lie on the scene's bounding sphere ("ray_origin" shapes are not supported). */
    int32_t distant_flip_directions; /* "flip_directions", default false                         */
    int32_t distant_target_type;     /* "ray_target": none / point / shape                       */
    float distant_target_point[3];
    mts_shape distant_target_shape;  /* the nested target shape (rectangle or sphere); it is owned
                                        by the sensor and is NOT part of the scene geometry       */
    /* film */
    int32_t film_width, film_height;
    int32_t crop_offset[2], crop_size[2];
    int32_t rfilter_type;     /* hdrfilm default: gaussian (src/librender/film.cpp:45-49)        */
    float rfilter_radius;     /* box "radius" (default 0.5)                                      */
    float rfilter_stddev;     /* gaussian "stddev" (default 0.5)                                 */
    /* sampler (independent) */
    int32_t sample_count;     /* "sample_count", default 4                                       */
    uint64_t sampler_seed;    /* "seed", default 0                                               */
    int32_t medium;           /* sensor medium, -1 = none                                        */
    /* mradiancemeter / mdistant: one 4x4 matrix (row-major) per sub-sensor, exactly what the constructors store in
     * m_transforms (mradiancemeter.cpp:95-113: look_at(origin, origin + direction, up = coordinate_system(direction).first);
     * mdistant.cpp:160-175: look_at(0, direction, up = coordinate_system(direction).second)).  The film must be
     * multi_count x 1.  mdistant's "target" travels in distant_target_type / _point / _shape. */
    const float *multi_transforms;
    int32_t multi_count;
    float shutter_open_time;  /* "shutter_close" - "shutter_open" (sensor.cpp:20-27).  Nothing on this path is animated, so
                                 the only effect is the reference's: one more sampler draw per sample when it is > 0
                                 (integrator.cpp:248-250) */
    /* distant "ray_origin" / distantflux "origin" (distant.cpp:126-130,280-289,367-383; distantflux.cpp:172-184,244-255):
     * 0 = ray origins on the scene's bounding sphere, 1 = the target point is projected onto this shape against the ray
     * direction (rectangle, disk or sphere; a miss gives the sample a zero weight) */
    int32_t distant_origin_type;
    mts_shape distant_origin_shape;
    /* "srf" of perspective / radiancemeter (perspective.cpp:113-116,173-182; radiancemeter.cpp:62-66,116-124): the spectral response
       function the sample's wavelengths are drawn from (Texture::sample_spectrum of a uniform or discrete spectrum) instead of
       sample_wavelength().  1 + index into mts_scene_desc.spectra, 0 = none (a zeroed record has none).  Spectral variant only. */
    int32_t srf;
    /* Random streams.  0: the scalar variants' -- one PCG32 stream per pixel, seeded from the spiral block id (integrator.cpp:198),
       all samples of the pixel drawn from it in order: fixed-seed parity with scalar_rgb.  1: the wavefront (gpu_*) variants' --
       one stream per (pixel, sample): lane L = pixel * sample_count + sample of the wavefront is seeded with
       (sample_tea_64(seed, L), sample_tea_64(L, seed)) (librender/sampler.cpp:89-92, integrator.cpp:140-172; pixel = y * crop_width + x
       inside the crop window).  Samples are then independent of each other, so small films are spread over the whole GPU (several
       workgroups share a block's samples); needs samples_per_pass = all (one pass). */
    int32_t sampler_wavefront;
} mts_sensor;

/* ---- Integrator (src/integrators/{path,volpath}.cpp, src/librender/integrator.cpp:23-39,302-315) */
enum { MTS_INTEGRATOR_PATH = 0, MTS_INTEGRATOR_VOLPATH = 1, MTS_INTEGRATOR_VOLPATHMIS = 2 /* src/integrators/volpathmis.cpp */ };
typedef struct mts_integrator {
    int32_t type;
    int32_t max_depth;        /* default -1 (infinite)                                           */
    int32_t rr_depth;         /* default 5                                                       */
    int32_t hide_emitters;    /* default false                                                   */
    int32_t block_size;       /* default 0 = heuristic; this backend pins 32 (MTS_BLOCK_SIZE)    */
    int32_t samples_per_pass; /* default -1 = all                                                */
    float timeout;            /* seconds, < 0 = none                                             */
    int32_t use_spectral_mis; /* volpathmis "use_spectral_mis", default true (volpathmis.cpp:29,38)  */
    int32_t monochrome;       /* 1: the semantics of the *_mono variants (is_monochromatic_v): no colour-channel draw
                                 (volpath.cpp:64-67), film X = Y = Z = L (integrator.cpp:270-271); the caller passes every
                                 colour as its luminance in all three channels.  0: *_rgb.                                */
    int32_t spectral;         /* 1: the semantics of scalar_spectral (is_spectral_v): Spectrum<Float, 4>, four wavelengths per sample drawn
                                 as include/mitsuba/core/spectrum.h:305-314 prescribes, colours given as spectra (mts_scene_desc.spectra),
                                 film = spectrum_to_xyz (:210-217).  Integrators: path, volpath.                          */
    /* Eradiate's wavelength-bin integrators wrapped around `type` (spectral variant only): src/integrators/nbins.cpp (bin_mode 1:
       bin i collects the wavelengths within bin_hi[i] = "tolerance" of bin_lo[i] = the i-th of "wavelengths") and
       src/integrators/bins.cpp (bin_mode 2: bin i = the interval [bin_lo[i], bin_hi[i]] of "bins").  Each bin adds two AOV
       channels to the film behind X, Y, Z, A, W: the summed radiance of the sample's wavelengths that fall into the bin and their
       number (nbins.cpp:107-121, bins.cpp:99-107); the film then holds 5 + 2 bin_count floats per pixel.                       */
    int32_t bin_mode, bin_count;
    const float *bin_lo, *bin_hi;
} mts_integrator;

typedef struct mts_scene_desc {
    uint32_t abi_version;     /* MTS_ABI_VERSION */
    const mts_volume *volumes;   int32_t volume_count;
    const mts_phase *phases;     int32_t phase_count;
    const mts_medium *media;     int32_t medium_count;
    const mts_bsdf *bsdfs;       int32_t bsdf_count;
    const mts_shape *shapes;     int32_t shape_count;
    const mts_emitter *emitters; int32_t emitter_count;   /* in scene declaration order */
    mts_sensor sensor;
    mts_integrator integrator;
    const mts_spectrum *spectra; int32_t spectrum_count;  /* spectral variant only */
} mts_scene_desc;

typedef struct mts_scene mts_scene;   /* opaque */

/* Render statistics; the three counters feed the algorithmic-bytes model of SURVEY.md 8(d). */
typedef struct mts_stats {
    uint64_t samples;         /* camera samples rendered by this call                            */
    uint64_t n_iter;          /* iterations of the main integrator loop (volpath.cpp:72)         */
    uint64_t n_lookup;        /* get_scattering_coefficients calls in heterogeneous media        */
    uint64_t n_nee_step;      /* iterations of the two NEE tracking loops (volpath.cpp:282,385)  */
    double kernel_ms;         /* device time of the render launches, HIP events on the stream (the calibration launch is not in it) */
    double wall_ms;           /* host wall time of the call                                      */
    int32_t kernel_launches;  /* render launches behind kernel_ms                                */
    int32_t cancelled;        /* 1 if mts_cancel stopped the render (render() == false, integrator.cpp:178) */
    int32_t timed_out;        /* 1 if the "timeout" of the integrator stopped it (should_stop(), integrator.h:143-146;
                                 like the reference, render() still returns true then)            */
    int32_t kernel_variant;   /* kernel formulation of the last launch: 0 = nested per-lane loops, 1 = per-lane state machine,
                                 10000 + P = regrouping machine on LDS rings with P paths per workgroup, 20000 + P = lane-affine driver;
                                 + 100000 U when the kernel came from lean translation unit U (1 a, 2 b, 3 s, 4 p, 5 ps, 6 h, 7 c: the same kernel
                                 compiled without what this scene cannot contain -- same film; MTSAMD_LEAN=0 turns them off) */
    int32_t calibration_launches; /* 0 or 1: the short launch that measures the blocks' costs before a render with more blocks than CUs
                                     (expensive blocks first; its samples are discarded)          */
    int32_t reserved_;
    double calibration_ms;    /* device time of that launch                                      */
} mts_stats;

typedef struct mts_render_opts {
    int32_t shard_index;      /* this rank renders blocks with block_id % shard_count == index   */
    int32_t shard_count;      /* 1 = whole image                                                 */
    int32_t device;           /* HIP device ordinal                                              */
    void *stream;             /* hipStream_t to launch on, NULL = default stream                 */
    int32_t film_on_device;   /* 0: `film` is host memory; 1: `film` is a device pointer         */
    int32_t collect_counters; /* 1: fill n_iter / n_lookup / n_nee_step (slower kernel variant)  */
    int64_t film_capacity;    /* floats the caller's `film` buffer holds; mts_render refuses a buffer smaller than
                                 crop_width x crop_height x (5 + 2 x spectral bins).  0: unchecked                */
} mts_render_opts;

/* Library / device queries */
int  mts_abi_version(void);
const char *mts_build_id(void);                 /* 16 hex digits: hash of the sources, headers and compiler flags this library was built from
                                                   (eradiate-kernel_amd/_buildid.py); the binding refuses a library that is not the tree's */
const char *mts_last_error(void);
int  mts_device_count(int *count);
int  mts_abi_sizeof(const char *struct_name);   /* sizeof(<struct_name>) as compiled, -1 if unknown (binding self-check) */

/* Scene(const Properties&) + plugin constructors: validates the description, runs the reference's
 * constructor-time precomputation, builds the acceleration structure and uploads everything to HBM. */
int  mts_scene_create(const mts_scene_desc *desc, int device, mts_scene **out);
int  mts_scene_destroy(mts_scene *scene);

/* Integrator::render + Film::bitmap(raw=True): renders `sensor.sample_count` samples per pixel and
 * ADDS nothing to previous content: `film` receives crop_height*crop_width*5 floats (X,Y,Z,A,W),
 * row-major, exactly the reference's raw film storage.  Returns 0 also when cancelled
 * (stats->cancelled = 1, like render() returning false). */
int  mts_render(mts_scene *scene, const mts_render_opts *opts, float *film, mts_stats *stats);

/* Integrator::cancel */
int  mts_cancel(mts_scene *scene);

/* The SIGINT scope of the reference's Python binding (src/librender/python/integrator_v.cpp:129-151): between enter and exit a
 * SIGINT cancels `scene`'s running render from an async-signal-safe C handler (two stores: the stop flag and the word the kernels
 * poll), puts the previous handler back and re-raises the signal for it -- so Ctrl-C winds the render down within milliseconds,
 * the finished samples stay on the film, and the caller's own handler (Python: KeyboardInterrupt) still sees the signal.
 * A binding calls enter before mts_render and exit after it, on the thread that owns signal handling (Python: the main thread).
 * One scope per process at a time: enter fails while another scope is open. */
int  mts_sigint_scope_enter(mts_scene *scene);
int  mts_sigint_scope_exit(void);

/* SamplingIntegrator::sample for n caller-supplied rays (python binding integrator_v.cpp:62-78):
 * lane i uses a PCG32 seeded as `sampler.seed(seed_offset + i)`; inputs/outputs are host SoA arrays. */
int  mts_sample(mts_scene *scene, int32_t n, uint64_t seed_offset,
                const float *ox, const float *oy, const float *oz,
                const float *dx, const float *dy, const float *dz,
                float *out_rgb /* 3*n */, uint8_t *out_valid /* n */);

/* The same for a scene of the spectral variant: the rays carry their four wavelengths (Ray::wavelengths, include/mitsuba/core/ray.h;
 * python: RayDifferential3f(..., wavelengths=...)), the result is the integrator's Spectrum at those wavelengths -- for `nbins` /
 * `bins` the wrapped integrator's (nbins.cpp:127-134).  Nothing is converted to XYZ: that is render_sample's job (integrator.cpp:266). */
int  mts_sample_spectral(mts_scene *scene, int32_t n, uint64_t seed_offset,
                         const float *ox, const float *oy, const float *oz,
                         const float *dx, const float *dy, const float *dz,
                         const float *wavelengths /* 4*n, nm */,
                         float *out_spec /* 4*n */, uint8_t *out_valid /* n */);

/* Closest-hit query used by the traversal parity tests (Scene::ray_intersect, scene.cpp:117-125). */
int  mts_ray_intersect(mts_scene *scene, int32_t n,
                       const float *o /* 3*n */, const float *d /* 3*n */,
                       const float *mint, const float *maxt,
                       float *out_t /* n, inf = miss */, int32_t *out_shape /* n */,
                       int32_t *out_prim /* n */, float *out_p /* 3*n */, float *out_n /* 3*n */);

/* sample_tea_32 / sample_tea_64 / sample_tea_float32 (include/mitsuba/core/random.h:75-85,106-116,137-140) for n (v0, v1) pairs,
 * computed on the device: the hash the reference's wavefront variants seed their per-lane PCG32 streams with. */
int  mts_sample_tea(int device, int32_t n, const uint32_t *v0, const uint32_t *v1, int32_t rounds,
                    uint32_t *out32 /* n */, uint64_t *out64 /* n */, float *out_float32 /* n */);

/* PCG32Sampler::seed of the reference's wavefront (gpu_*) variants (src/librender/sampler.cpp:83-92): lane i is seeded with
 * (sample_tea_64(seed_value, i), sample_tea_64(i, seed_value)); writes the first `count` next_1d() of every lane, lane-major.
 * The render path of this backend uses the scalar seeding instead (one stream per pixel, integrator.cpp:198), which is what
 * makes its films equal scalar_rgb's; this entry exposes the other scheme for callers that want the reference's gpu_rgb streams. */
int  mts_wavefront_sampler(int device, int32_t lanes, uint64_t seed_value, int32_t count, float *out /* lanes * count */);

#ifdef __cplusplus
}
#endif
#endif /* MTSAMD_H */
