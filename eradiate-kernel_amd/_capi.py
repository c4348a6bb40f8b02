"""ctypes mirror of include/mtsamd.h (the C ABI of libmtsamd.so).

The structures below are field-for-field copies of the C declarations; tests/test_abi.py checks
their sizes against the compiled library (`mts_abi_sizeof`).  The library is loaded lazily and the
loader fails loudly: there is no CPU fallback for the product path.
"""
import ctypes as C
import os

from . import _buildid

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MTSAMD_LIB") or os.path.join(HERE, "libmtsamd.so")

MTS_ABI_VERSION = 9

# enums (include/mtsamd.h)
VOLUME_CONST, VOLUME_GRID, VOLUME_GRID_SPECTRAL = 0, 1, 2
SPECTRUM_UNIFORM, SPECTRUM_REGULAR, SPECTRUM_IRREGULAR, SPECTRUM_DISCRETE = 0, 1, 2, 3
FILTER_NEAREST, FILTER_TRILINEAR = 0, 1
WRAP_REPEAT, WRAP_MIRROR, WRAP_CLAMP = 0, 1, 2
PHASE_ISOTROPIC, PHASE_HG, PHASE_RAYLEIGH, PHASE_BLEND, PHASE_TABULATED = 0, 1, 2, 3, 4
MEDIUM_HOMOGENEOUS, MEDIUM_HETEROGENEOUS = 0, 1
BSDF_DIFFUSE, BSDF_NULL, BSDF_RPV, BSDF_BILAMBERTIAN = 0, 1, 2, 3
SHAPE_RECTANGLE, SHAPE_CUBE, SHAPE_SPHERE, SHAPE_MESH, SHAPE_DISK = 0, 1, 2, 3, 4
EMITTER_DIRECTIONAL, EMITTER_AREA, EMITTER_CONSTANT, EMITTER_POINT = 0, 1, 2, 3
SENSOR_PERSPECTIVE, SENSOR_DISTANT, SENSOR_MRADIANCEMETER, SENSOR_MDISTANT, SENSOR_DISTANTFLUX = 0, 1, 2, 3, 4
RFILTER_BOX, RFILTER_GAUSSIAN = 0, 1
DISTANT_TARGET_NONE, DISTANT_TARGET_POINT, DISTANT_TARGET_SHAPE = 0, 1, 2
INTEGRATOR_PATH, INTEGRATOR_VOLPATH, INTEGRATOR_VOLPATHMIS = 0, 1, 2

f32 = C.c_float
i32 = C.c_int32
fp = C.POINTER(C.c_float)


class Transform(C.Structure):
    _fields_ = [("matrix", f32 * 16), ("inverse_transpose", f32 * 16)]


class Spectrum(C.Structure):
    _fields_ = [("type", i32), ("value", f32), ("lambda_min", f32), ("lambda_max", f32), ("values", fp), ("count", i32),
                ("wavelengths", fp), ("pmf", fp)]


class Volume(C.Structure):
    _fields_ = [("type", i32), ("value", f32 * 3), ("to_world", Transform), ("data", fp),
                ("nx", i32), ("ny", i32), ("nz", i32), ("channels", i32),
                ("filter_type", i32), ("wrap_mode", i32), ("use_grid_bbox", i32),
                ("file_bbox_min", f32 * 3), ("file_bbox_max", f32 * 3),
                ("has_max_value", i32), ("max_value", f32),
                ("value_spectrum", i32), ("lambda_min", f32), ("lambda_max", f32)]


class Phase(C.Structure):
    _fields_ = [("type", i32), ("g", f32), ("child", i32 * 2), ("weight_volume", i32),
                ("tab_values", fp), ("tab_count", i32)]


class Medium(C.Structure):
    _fields_ = [("type", i32), ("sigma_t_volume", i32), ("albedo_volume", i32), ("scale", f32),
                ("phase", i32), ("sample_emitters", i32), ("has_spectral_extinction", i32)]


class Bsdf(C.Structure):
    _fields_ = [("type", i32), ("reflectance", f32 * 3), ("rho_0", f32 * 3), ("k", f32 * 3),
                ("g", f32 * 3), ("rho_c", f32 * 3), ("transmittance", f32 * 3), ("spectrum", i32 * 6)]


class Shape(C.Structure):
    _fields_ = [("type", i32), ("to_world", Transform), ("flip_normals", i32),
                ("center", f32 * 3), ("radius", f32),
                ("vertex_positions", fp), ("vertex_normals", fp), ("vertex_texcoords", fp),
                ("faces", C.POINTER(C.c_uint32)), ("vertex_count", i32), ("face_count", i32),
                ("bsdf", i32), ("interior_medium", i32), ("exterior_medium", i32), ("emitter", i32)]


class Emitter(C.Structure):
    _fields_ = [("type", i32), ("to_world", Transform), ("radiance", f32 * 3), ("shape", i32), ("radiance_spectrum", i32)]


class Sensor(C.Structure):
    _fields_ = [("type", i32), ("to_world", Transform), ("fov_x", f32),
                ("near_clip", f32), ("far_clip", f32), ("principal_point_offset", f32 * 2),
                ("distant_flip_directions", i32), ("distant_target_type", i32),
                ("distant_target_point", f32 * 3), ("distant_target_shape", Shape),
                ("film_width", i32), ("film_height", i32), ("crop_offset", i32 * 2), ("crop_size", i32 * 2),
                ("rfilter_type", i32), ("rfilter_radius", f32), ("rfilter_stddev", f32),
                ("sample_count", i32), ("sampler_seed", C.c_uint64), ("medium", i32),
                ("multi_transforms", C.POINTER(C.c_float)), ("multi_count", i32),
                ("shutter_open_time", f32), ("distant_origin_type", i32), ("distant_origin_shape", Shape), ("srf", i32), ("sampler_wavefront", i32)]


class Integrator(C.Structure):
    _fields_ = [("type", i32), ("max_depth", i32), ("rr_depth", i32), ("hide_emitters", i32),
                ("block_size", i32), ("samples_per_pass", i32), ("timeout", f32), ("use_spectral_mis", i32), ("monochrome", i32),
                ("spectral", i32), ("bin_mode", i32), ("bin_count", i32), ("bin_lo", fp), ("bin_hi", fp)]


class SceneDesc(C.Structure):
    _fields_ = [("abi_version", C.c_uint32),
                ("volumes", C.POINTER(Volume)), ("volume_count", i32),
                ("phases", C.POINTER(Phase)), ("phase_count", i32),
                ("media", C.POINTER(Medium)), ("medium_count", i32),
                ("bsdfs", C.POINTER(Bsdf)), ("bsdf_count", i32),
                ("shapes", C.POINTER(Shape)), ("shape_count", i32),
                ("emitters", C.POINTER(Emitter)), ("emitter_count", i32),
                ("sensor", Sensor), ("integrator", Integrator),
                ("spectra", C.POINTER(Spectrum)), ("spectrum_count", i32)]


class Stats(C.Structure):
    _fields_ = [("samples", C.c_uint64), ("n_iter", C.c_uint64), ("n_lookup", C.c_uint64),
                ("n_nee_step", C.c_uint64), ("kernel_ms", C.c_double), ("wall_ms", C.c_double),
                ("kernel_launches", i32), ("cancelled", i32), ("timed_out", i32), ("kernel_variant", i32),
                ("calibration_launches", i32), ("reserved_", i32), ("calibration_ms", C.c_double)]


class RenderOpts(C.Structure):
    _fields_ = [("shard_index", i32), ("shard_count", i32), ("device", i32), ("stream", C.c_void_p),
                ("film_on_device", i32), ("collect_counters", i32), ("film_capacity", C.c_int64)]


ABI_STRUCTS = {"mts_spectrum": Spectrum, "mts_transform": Transform, "mts_volume": Volume, "mts_phase": Phase, "mts_medium": Medium,
               "mts_bsdf": Bsdf, "mts_shape": Shape, "mts_emitter": Emitter, "mts_sensor": Sensor,
               "mts_integrator": Integrator, "mts_scene_desc": SceneDesc, "mts_stats": Stats,
               "mts_render_opts": RenderOpts}

# every symbol include/mtsamd.h declares
ABI_SYMBOLS = ["mts_abi_version", "mts_build_id", "mts_last_error", "mts_device_count", "mts_scene_create", "mts_scene_destroy",
               "mts_render", "mts_cancel", "mts_sigint_scope_enter", "mts_sigint_scope_exit", "mts_sample", "mts_sample_spectral", "mts_ray_intersect", "mts_abi_sizeof", "mts_sample_tea", "mts_wavefront_sampler"]

_lib = None


class BackendError(RuntimeError):
    pass


def lib():
    """Load libmtsamd.so (built in-tree by build.py). No fallback: a missing library is an error."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BackendError(
            "libmtsamd.so not found at %s -- run `python __graft_entry__.py build` (hipcc, gfx950). "
            "The gpu_rgb backend has no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    L.mts_abi_version.restype = C.c_int
    L.mts_last_error.restype = C.c_char_p
    L.mts_device_count.argtypes = [C.POINTER(C.c_int)]
    L.mts_scene_create.argtypes = [C.POINTER(SceneDesc), C.c_int, C.POINTER(C.c_void_p)]
    L.mts_scene_destroy.argtypes = [C.c_void_p]
    L.mts_render.argtypes = [C.c_void_p, C.POINTER(RenderOpts), C.c_void_p, C.POINTER(Stats)]
    L.mts_cancel.argtypes = [C.c_void_p]
    L.mts_sigint_scope_enter.argtypes = [C.c_void_p]
    L.mts_sigint_scope_exit.argtypes = []
    L.mts_sample.argtypes = [C.c_void_p, i32, C.c_uint64] + [fp] * 6 + [fp, C.POINTER(C.c_uint8)]
    L.mts_sample_spectral.argtypes = [C.c_void_p, i32, C.c_uint64] + [fp] * 7 + [fp, C.POINTER(C.c_uint8)]
    L.mts_ray_intersect.argtypes = [C.c_void_p, i32, fp, fp, fp, fp, fp, C.POINTER(i32), C.POINTER(i32), fp, fp]
    L.mts_sample_tea.argtypes = [C.c_int, i32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), i32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), fp]
    L.mts_wavefront_sampler.argtypes = [C.c_int, i32, C.c_uint64, i32, fp]
    L.mts_abi_sizeof.argtypes = [C.c_char_p]
    L.mts_abi_sizeof.restype = C.c_int
    if L.mts_abi_version() != MTS_ABI_VERSION:
        raise BackendError("libmtsamd.so ABI version %d != %d" % (L.mts_abi_version(), MTS_ABI_VERSION))
    # the binary must be the one this tree builds (a stale or foreign library would render with other kernels); an explicit MTSAMD_LIB
    # (side-by-side measurement builds with extra flags) is taken as it is
    L.mts_build_id.restype = C.c_char_p
    if not os.environ.get("MTSAMD_LIB") and os.path.isdir(_buildid.CSRC):       # a deployed package without csrc/ has nothing to compare with
        try:
            built, tree = L.mts_build_id().decode(), _buildid.tree_build_id(_buildid.effective_flags())
        except OSError as e:
            raise BackendError("cannot read the sources of libmtsamd.so to verify the build (%s)" % e)
        if built != tree:
            raise BackendError("%s was built from other sources or flags (build id %s, this tree: %s) -- run `python eradiate-kernel_amd/build.py`"
                               % (LIB_PATH, built, tree))
    _lib = L
    return L


def check(status):
    if status != 0:
        msg = lib().mts_last_error()
        raise RuntimeError(msg.decode() if msg else "libmtsamd error %d" % status)
