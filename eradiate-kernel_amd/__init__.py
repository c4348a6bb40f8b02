"""eradiate-kernel_amd: host-side mirror of the reference's Python surface for the path / volpath hot path.

Call sequence kept from the reference (/root/reference/src/python/__init__.py:124-188,
src/librender/python/integrator_v.cpp:124-156, src/films/hdrfilm.cpp:251-259,
docs/src/python_interface/rendering_scene.rst:11-60):

    import mitsuba_amd as mitsuba            # (root-level shim around this package)
    mitsuba.set_variant('gpu_rgb')
    from mitsuba.core import ScalarTransform4f
    from mitsuba.core.xml import load_dict
    scene = load_dict({...})
    scene.integrator().render(scene, scene.sensors()[0])
    img = np.array(scene.sensors()[0].film().bitmap(raw=True))      # (H, W, 5) XYZAW

Everything below the Python layer is libmtsamd.so (C ABI in include/mtsamd.h, HIP kernels for gfx950).
Only the `gpu_rgb` variant exists: there is no CPU fallback in the product path.
"""
import ctypes as C
import threading
import types

import os
import numpy as np

from . import _capi as A
from .transform import ScalarTransform4f
from .scene_dict import build_scene_desc
from . import volume_io
from .fresolver import FileResolver, file_resolver, set_file_resolver        # Thread.thread().file_resolver() of the reference

__version__ = "0.1.0"
ERADIATE_KERNEL = True          # src/python/__init__.py:191-193
# gpu_mono: the semantics of scalar_mono (one channel, luminance of every colour); gpu_spectral: those of scalar_spectral
# (Spectrum<Float, 4>: four wavelengths per sample, colours given as spectra; integrators path and volpath)
_VARIANTS = ["gpu_rgb", "gpu_mono", "gpu_spectral"]
_tls = threading.local()


def variants():
    """mitsuba.variants() (src/python/__init__.py:185-188)."""
    return list(_VARIANTS)


def variant():
    return getattr(_tls, "variant", None)


def set_variant(name):
    """mitsuba.set_variant() (src/python/__init__.py:124-177): per-thread; unknown variants raise ImportError
    with the list of compiled variants, like the reference."""
    if name not in _VARIANTS:
        raise ImportError("Requested an unsupported variant \"%s\". The following variants are available: %s."
                          % (name, ", ".join(_VARIANTS)))
    A.lib()                      # fail loudly if the HIP backend is missing
    _tls.variant = name


def _require_variant():
    if variant() is None:
        raise ImportError("Before importing any packages, you must specify the desired variant of Mitsuba "
                          "using \"mitsuba.set_variant(..)\".\nThe following variants are available: %s."
                          % ", ".join(_VARIANTS))


class PixelFormat:
    Y, YA, RGB, RGBA, XYZ, XYZA, XYZAW, MultiChannel = range(8)


class StructType:
    Float32 = "float32"


class Struct:
    Type = StructType


class Bitmap:
    """Minimal Bitmap: an (H, W, C) float32 image with a pixel format (src/libcore/bitmap.cpp)."""
    PixelFormat = PixelFormat

    def __init__(self, array, pixel_format):
        self._a = np.ascontiguousarray(array, dtype=np.float32)
        self._fmt = pixel_format

    def pixel_format(self):
        return self._fmt

    def width(self):
        return self._a.shape[1]

    def height(self):
        return self._a.shape[0]

    def channel_count(self):
        return self._a.shape[2]

    def __array__(self, dtype=None, copy=None):
        return self._a if dtype is None else self._a.astype(dtype)

    def convert(self, pixel_format, component_format=StructType.Float32, srgb_gamma=False):
        """XYZAW -> other formats: divide by the weight channel, then XYZ -> linear sRGB with the matrix of
        src/films/hdrfilm.cpp:277-297 / include/mitsuba/core/spectrum.h:229-239."""
        if srgb_gamma:
            raise RuntimeError("sRGB gamma encoding is not supported")
        a = self._a
        if self._fmt in (PixelFormat.RGB, PixelFormat.RGBA) and pixel_format in (PixelFormat.RGB, self._fmt):
            return Bitmap(a[..., :3] if pixel_format == PixelFormat.RGB else a, pixel_format)
        if self._fmt != PixelFormat.XYZAW:
            raise RuntimeError("Bitmap.convert(): only XYZAW sources are supported")
        w = a[..., 4:5]
        inv = np.where(w != 0, 1.0 / np.where(w != 0, w, 1), 0).astype(np.float32)
        xyz = a[..., :3] * inv
        alpha = a[..., 3:4] * inv
        m = np.array([[3.240479, -1.537150, -0.498535],
                      [-0.969256, 1.875991, 0.041556],
                      [0.055648, -0.204043, 1.057311]], dtype=np.float32)
        rgb = xyz @ m.T
        if pixel_format == PixelFormat.RGB:
            return Bitmap(rgb, pixel_format)
        if pixel_format == PixelFormat.RGBA:
            return Bitmap(np.concatenate([rgb, alpha], -1), pixel_format)
        if pixel_format == PixelFormat.XYZ:
            return Bitmap(xyz, pixel_format)
        if pixel_format == PixelFormat.XYZA:
            return Bitmap(np.concatenate([xyz, alpha], -1), pixel_format)
        if pixel_format == PixelFormat.Y:
            return Bitmap(xyz[..., 1:2], pixel_format)
        if pixel_format == PixelFormat.YA:
            return Bitmap(np.concatenate([xyz[..., 1:2], alpha], -1), pixel_format)
        raise RuntimeError("Bitmap.convert(): unsupported target pixel format")


class Film:
    """hdrfilm (src/films/hdrfilm.cpp): owns the XYZAW storage the integrator writes."""

    _FORMATS = {"luminance": PixelFormat.Y, "luminance_alpha": PixelFormat.YA, "rgb": PixelFormat.RGB, "rgba": PixelFormat.RGBA,
                "xyz": PixelFormat.XYZ, "xyza": PixelFormat.XYZA}

    def __init__(self, sensor_rec, pixel_format="rgba"):
        self._rec = sensor_rec
        self._storage = None
        self._pixel_format = self._FORMATS[pixel_format]      # hdrfilm.cpp:122-151 (monochrome variants force 'luminance')

    def size(self):
        return (self._rec.film_width, self._rec.film_height)

    def crop_size(self):
        return tuple(self._rec.crop_size)

    def crop_offset(self):
        return tuple(self._rec.crop_offset)

    def bitmap(self, raw=False):
        if self._storage is None:
            raise RuntimeError("Film.bitmap(): nothing has been rendered yet")
        if self._storage.shape[2] != 5:
            # AOV channels behind X, Y, Z, A, W (hdrfilm.cpp:262-320): the raw storage as a multichannel bitmap; developed: R, G, B, A
            # and every AOV channel, all divided by the weight channel, which is dropped
            src = Bitmap(self._storage, PixelFormat.MultiChannel)
            if raw:
                return src
            rgba = np.asarray(Bitmap(self._storage[..., :5], PixelFormat.XYZAW).convert(PixelFormat.RGBA))
            w = self._storage[..., 4:5]
            inv = np.where(w != 0, 1.0 / np.where(w != 0, w, 1), 0).astype(np.float32)
            return Bitmap(np.concatenate([rgba, self._storage[..., 5:] * inv], -1), PixelFormat.MultiChannel)
        src = Bitmap(self._storage, PixelFormat.XYZAW)
        if raw:
            return src
        return src.convert(self._pixel_format)                # hdrfilm.cpp:277-297


class Sampler:
    def __init__(self, rec):
        self._rec = rec

    def sample_count(self):
        return self._rec.sample_count


class Sensor:
    def __init__(self, rec, pixel_format="rgba"):
        self._rec = rec
        self._film = Film(rec, pixel_format)
        self._sampler = Sampler(rec)

    def film(self):
        return self._film

    def sampler(self):
        return self._sampler

    def needs_aperture_sample(self):
        return self._rec.type in (A.SENSOR_DISTANT, A.SENSOR_MDISTANT, A.SENSOR_DISTANTFLUX)


class Integrator:
    """SamplingIntegrator (include/mitsuba/render/integrator.h:42-51,114-119)."""

    def __init__(self, scene):
        self._scene = scene
        self.last_stats = None

    def render(self, scene, sensor, shard_index=0, shard_count=1, device_film=None, stream=None,
               collect_counters=False, device_film_floats=None):
        """Integrator.render(scene, sensor) -> bool: `not m_stop` (integrator.cpp:178), i.e. False iff cancel() stopped it; a render
        cut short by the integrator's "timeout" returns True like the reference's (last_stats["timed_out"] tells).

        The reference binding releases the GIL and turns SIGINT into cancel() from a C signal handler
        (integrator_v.cpp:124-156); ctypes releases the GIL for the duration of mts_render, and the
        SIGINT scope of the C ABI (mts_sigint_scope_enter / _exit) does the same here.  Extra keyword arguments are extensions used by the
        multi-GPU path: `shard_*` selects the blocks this rank renders, `device_film` is a device
        pointer (e.g. torch tensor data_ptr) receiving the film instead of host memory: crop_height x crop_width x (5 + 2 x bins)
        floats -- X, Y, Z, A, W and, under `nbins` / `bins`, two AOV channels per spectral bin.  `device_film_floats` is the size of
        that buffer in floats (default: height x width x 5, the plain XYZAW film); mts_render refuses a buffer that is too small.
        """
        if scene is not self._scene:
            raise RuntimeError("Integrator.render(): the integrator belongs to another scene")
        rec = scene._desc.sensor
        h, w = rec.crop_size[1], rec.crop_size[0]
        opts = A.RenderOpts()
        opts.shard_index, opts.shard_count = shard_index, shard_count
        opts.device = scene._device
        opts.stream = stream
        opts.collect_counters = int(bool(collect_counters))
        stats = A.Stats()
        # SIGINT -> cancel() as the reference's binding does it (integrator_v.cpp:129-151): a C-level handler, because a Python-level
        # one only runs between bytecodes, i.e. after mts_render has returned.  The handler cancels the render, puts the previous
        # handler back and re-raises, so KeyboardInterrupt still reaches the caller -- once the render has wound down and its
        # finished samples are on the film (which is why the film storage is attached BEFORE the call).
        scoped = threading.current_thread() is threading.main_thread() and A.lib().mts_sigint_scope_enter(scene._handle) == 0
        try:
            if device_film is not None:
                opts.film_on_device = 1
                opts.film_capacity = int(device_film_floats) if device_film_floats is not None else h * w * 5
                sensor._film._storage = None
                A.check(A.lib().mts_render(scene._handle, C.byref(opts), C.c_void_p(int(device_film)), C.byref(stats)))
            else:
                out = np.zeros((h, w, 5 + 2 * scene._desc.integrator.bin_count), dtype=np.float32)    # X, Y, Z, A, W + aov_names()
                opts.film_on_device = 0
                opts.film_capacity = out.size
                sensor._film._storage = out
                A.check(A.lib().mts_render(scene._handle, C.byref(opts), out.ctypes.data_as(C.c_void_p), C.byref(stats)))
        finally:
            # stats first: a KeyboardInterrupt re-raised by the handler surfaces at the next bytecode
            self.last_stats = {k: getattr(stats, k) for k, _ in A.Stats._fields_ if k != "reserved_"}
            if scoped:
                A.lib().mts_sigint_scope_exit()
        return not bool(stats.cancelled)

    def cancel(self):
        A.lib().mts_cancel(self._scene._handle)

    def aov_names(self):
        """SamplingIntegrator::aov_names (integrator.cpp:47-49; nbins.cpp:127-134, bins.cpp:112-119)."""
        return list(getattr(self._scene._keep, "aov_names", []))

    def sample(self, scene, origins, directions, seed_offset=0, wavelengths=None):
        """SamplingIntegrator.sample for a batch of rays (integrator_v.cpp:62-78): returns (rgb (n,3), valid (n,)).  Scenes of the
        spectral variant take the rays' `wavelengths` ((4,) for all rays or (n, 4), nm: Ray.wavelengths) and return the (n, 4)
        spectrum at those wavelengths."""
        o = np.ascontiguousarray(origins, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(directions, dtype=np.float32).reshape(-1, 3)
        n = o.shape[0]
        cols = [np.ascontiguousarray(o[:, i]) for i in range(3)] + [np.ascontiguousarray(d[:, i]) for i in range(3)]
        valid = np.zeros(n, dtype=np.uint8)
        if wavelengths is not None:
            w = np.ascontiguousarray(np.broadcast_to(np.asarray(wavelengths, dtype=np.float32).reshape(-1, 4), (n, 4)))
            spec = np.zeros((n, 4), dtype=np.float32)
            A.check(A.lib().mts_sample_spectral(scene._handle, n, seed_offset, *[c.ctypes.data_as(A.fp) for c in cols], w.ctypes.data_as(A.fp),
                                                spec.ctypes.data_as(A.fp), valid.ctypes.data_as(C.POINTER(C.c_uint8))))
            return spec, valid.astype(bool)
        rgb = np.zeros((n, 3), dtype=np.float32)
        A.check(A.lib().mts_sample(scene._handle, n, seed_offset, *[c.ctypes.data_as(A.fp) for c in cols],
                                   rgb.ctypes.data_as(A.fp), valid.ctypes.data_as(C.POINTER(C.c_uint8))))
        return rgb, valid.astype(bool)


class Scene:
    """Scene (src/librender/scene.cpp:22-104): owns the device-resident flattened scene."""

    def __init__(self, desc, keep, device=0):
        self._desc, self._keep, self._device = desc, keep, device
        h = C.c_void_p()
        A.check(A.lib().mts_scene_create(C.byref(desc), device, C.byref(h)))
        self._handle = h
        self._sensor = Sensor(desc.sensor, getattr(keep, "film_pixel_format", "rgba"))
        self._integrator = Integrator(self)

    def sensors(self):
        return [self._sensor]

    def integrator(self):
        return self._integrator

    def ray_intersect(self, o, d, mint=None, maxt=None):
        """Scene::ray_intersect (scene.cpp:117-125) for a batch of rays; returns a dict of arrays."""
        o = np.ascontiguousarray(o, dtype=np.float32).reshape(-1, 3)
        d = np.ascontiguousarray(d, dtype=np.float32).reshape(-1, 3)
        n = o.shape[0]
        mint = np.full(n, 1500 * 2.0 ** -24, dtype=np.float32) if mint is None else np.ascontiguousarray(mint, dtype=np.float32)
        maxt = np.full(n, np.inf, dtype=np.float32) if maxt is None else np.ascontiguousarray(maxt, dtype=np.float32)
        t = np.zeros(n, np.float32); shape = np.zeros(n, np.int32); prim = np.zeros(n, np.int32)
        p = np.zeros((n, 3), np.float32); nn = np.zeros((n, 3), np.float32)
        A.check(A.lib().mts_ray_intersect(self._handle, n, o.ctypes.data_as(A.fp), d.ctypes.data_as(A.fp),
                                          mint.ctypes.data_as(A.fp), maxt.ctypes.data_as(A.fp), t.ctypes.data_as(A.fp),
                                          shape.ctypes.data_as(C.POINTER(C.c_int32)), prim.ctypes.data_as(C.POINTER(C.c_int32)),
                                          p.ctypes.data_as(A.fp), nn.ctypes.data_as(A.fp)))
        return {"t": t, "shape": shape, "prim_index": prim, "p": p, "n": nn}

    def __del__(self):
        try:
            if getattr(self, "_handle", None):
                A.lib().mts_scene_destroy(self._handle)
                self._handle = None
        except Exception:
            pass


def sample_tea(v0, v1, rounds=4, device=0):
    """mitsuba.core.sample_tea_32 / sample_tea_64 / sample_tea_float32 (core/random.h:75-140) on the device: returns the three arrays."""
    _require_variant()
    a = np.ascontiguousarray(np.broadcast_arrays(np.asarray(v0, np.uint32), np.asarray(v1, np.uint32))[0].ravel())
    b = np.ascontiguousarray(np.broadcast_arrays(np.asarray(v0, np.uint32), np.asarray(v1, np.uint32))[1].ravel())
    n = a.size
    o32, o64, of = np.zeros(n, np.uint32), np.zeros(n, np.uint64), np.zeros(n, np.float32)
    u32p = C.POINTER(C.c_uint32)
    A.check(A.lib().mts_sample_tea(device, n, a.ctypes.data_as(u32p), b.ctypes.data_as(u32p), int(rounds), o32.ctypes.data_as(u32p),
                                   o64.ctypes.data_as(C.POINTER(C.c_uint64)), of.ctypes.data_as(A.fp)))
    return o32, o64, of


def wavefront_sampler(lanes, seed_value, count, device=0):
    """The per-lane PCG32 streams of the reference's wavefront (gpu_*) variants (PCG32Sampler::seed, sampler.cpp:83-92): (lanes, count) floats."""
    _require_variant()
    out = np.zeros((int(lanes), int(count)), np.float32)
    A.check(A.lib().mts_wavefront_sampler(device, int(lanes), int(seed_value), int(count), out.ctypes.data_as(A.fp)))
    return out


def load_dict(d, device=0):
    """mitsuba.core.xml.load_dict (src/libcore/python/xml_v.cpp:23-68,100-272)."""
    _require_variant()
    if isinstance(d, dict) and d.get("type") in ("nbins", "bins"):
        # the reference's tests construct these integrators on their own (src/integrators/tests/test_bins.py:10-53): validated like
        # inside a scene, and good for aov_names()
        from . import scene_dict as SD
        b = SD.SceneBuilder(); b.spectra = []
        SD._SPECTRAL = b if variant() == "gpu_spectral" else None
        try:
            b.set_integrator(d, "integrator")
        finally:
            SD._SPECTRAL = None
        return _DetachedIntegrator(b.aov_names)
    desc, keep = build_scene_desc(d, mono=(variant() == "gpu_mono"), spectral=(variant() == "gpu_spectral"))
    return Scene(desc, keep, device)


class _DetachedIntegrator:
    def __init__(self, names):
        self._names = list(names)

    def aov_names(self):
        return list(self._names)


def load_string(string, device=0, **kwargs):
    """mitsuba.core.xml.load_string(string, variant, **parameters) (src/libcore/python/xml_v.cpp:76-98): the XML is turned into
    a scene dictionary (xml_io.py), keyword arguments fill `$parameters`."""
    from .xml_io import xml_to_dict
    from .fresolver import FileResolver, file_resolver, set_file_resolver
    kwargs.pop("variant", None)
    # as load_file: the parser (a <path> tag prepends to it) works on a copy of the FileResolver, the caller's comes back afterwards
    # (xml.cpp:1238-1240, 1275 do so for both entry points)
    backup = file_resolver()
    set_file_resolver(FileResolver(list(backup)))
    try:
        return load_dict(xml_to_dict(string, kwargs), device)
    finally:
        set_file_resolver(backup)


def load_file(path, device=0, **kwargs):
    """mitsuba.core.xml.load_file(path, variant, update_scene=False, **parameters) (xml_v.cpp:70-75)."""
    from .xml_io import file_to_dict
    from .fresolver import FileResolver, file_resolver, set_file_resolver
    kwargs.pop("variant", None); kwargs.pop("update_scene", None)
    # The parser works on a copy of the FileResolver and restores the caller's afterwards (xml.cpp:1238-1240,1275); like the
    # `mitsuba` executable (src/mitsuba/mitsuba.cpp:230-235) the copy also searches the scene file's own directory.
    backup = file_resolver()
    fr = FileResolver(list(backup))
    fr.append(os.path.dirname(os.path.abspath(path)))
    set_file_resolver(fr)
    try:
        return load_dict(file_to_dict(path, kwargs), device)
    finally:
        set_file_resolver(backup)


# virtual modules mitsuba.core / mitsuba.core.xml / mitsuba.render (src/python/__init__.py:115-121), registered so
# that `from mitsuba_amd.core.xml import load_dict` works like the reference's import line
def _virtual_module(name, **members):
    import sys
    mods = []
    for prefix in (__name__, "mitsuba_amd"):
        m = types.ModuleType(prefix + "." + name)
        m.__dict__.update(members)
        sys.modules[prefix + "." + name] = m
        mods.append(m)
    return mods[0]


_xml = _virtual_module("core.xml", load_dict=load_dict, load_string=load_string, load_file=load_file)
core = _virtual_module("core", ScalarTransform4f=ScalarTransform4f, Bitmap=Bitmap, Struct=Struct, xml=_xml)
for _p in ("mitsuba_amd",):
    import sys as _sys
    _sys.modules[_p + ".core"].xml = _sys.modules[_p + ".core.xml"]
render = _virtual_module("render", Scene=Scene, Integrator=Integrator, Sensor=Sensor, Film=Film)
