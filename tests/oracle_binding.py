"""ctypes binding of oracle/liboracle.so -- the CPU restatement used as the parity checker.

TEST INFRASTRUCTURE: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg only.
It consumes the very same C-ABI scene description (include/mtsamd.h) as the product library.
"""
import ctypes as C
import importlib
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = importlib.import_module("eradiate-kernel_amd")
A = importlib.import_module("eradiate-kernel_amd._capi")
SD = importlib.import_module("eradiate-kernel_amd.scene_dict")
SUFFIX = os.environ.get("MTSAMD_ORACLE_SUFFIX", "")          # "_asan": the sanitizer builds (tests/test_oracle_sanitizers.py)
LIB_PATH = os.path.join(ROOT, "oracle", "liboracle%s.so" % SUFFIX)
_lib = None
fp = A.fp


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), os.path.basename(LIB_PATH)], stdout=subprocess.DEVNULL)
        L = C.CDLL(LIB_PATH)
        L.oracle_last_error.restype = C.c_char_p
        L.oracle_scene_create.argtypes = [C.POINTER(A.SceneDesc), C.POINTER(C.c_void_p)]
        L.oracle_scene_destroy.argtypes = [C.c_void_p]
        L.oracle_render.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, fp, C.POINTER(A.Stats)]
        L.oracle_sample.argtypes = [C.c_void_p, C.c_int32, C.c_uint64] + [fp] * 6 + [fp, C.POINTER(C.c_uint8)]
        L.oracle_ray_intersect.argtypes = [C.c_void_p, C.c_int32, fp, fp, fp, fp, fp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), fp, fp]
        L.oracle_tea32.argtypes = [C.c_uint32, C.c_uint32, C.c_int]; L.oracle_tea32.restype = C.c_uint32
        L.oracle_tea64.argtypes = [C.c_uint32, C.c_uint32, C.c_int]; L.oracle_tea64.restype = C.c_uint64
        L.oracle_tea_float32.argtypes = [C.c_uint32, C.c_uint32, C.c_int]; L.oracle_tea_float32.restype = C.c_float
        L.oracle_pcg32.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.POINTER(C.c_uint32), fp]
        L.oracle_sampler_stream.argtypes = [C.c_uint64, C.c_uint64, C.c_int, fp]
        L.oracle_warp.argtypes = [C.c_int, C.c_float, C.c_float, fp]
        L.oracle_coordinate_system.argtypes = [fp, fp, fp]
        L.oracle_morton_decode.argtypes = [C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.oracle_spiral.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_int32)]; L.oracle_spiral.restype = C.c_int
        L.oracle_imageblock_put.argtypes = [C.c_int] * 6 + [C.c_float, C.c_float, C.c_int, C.c_int, fp, fp, fp, C.POINTER(C.c_int)]
        L.oracle_rfilter_eval.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float, C.c_int]; L.oracle_rfilter_eval.restype = C.c_float
        L.oracle_phase_eval.argtypes = [C.c_void_p, C.c_int, fp, fp, fp, fp]
        L.oracle_phase_sample.argtypes = [C.c_void_p, C.c_int, fp, fp, C.c_float, C.c_float, C.c_float, fp, fp]
        L.oracle_phase_eval_component.argtypes = [C.c_void_p, C.c_int, C.c_int, fp, fp, fp, fp, C.POINTER(C.c_int)]
        L.oracle_phase_sample_component.argtypes = [C.c_void_p, C.c_int, C.c_int, fp, fp, C.c_float, C.c_float, C.c_float, fp, fp]
        L.oracle_bsdf_eval.argtypes = [C.c_void_p, C.c_int, fp, fp, fp, fp]
        L.oracle_bsdf_sample.argtypes = [C.c_void_p, C.c_int, fp, C.c_float, C.c_float, C.c_float, fp, fp, fp, C.POINTER(C.c_uint32)]
        L.oracle_volume_eval.argtypes = [C.c_void_p, C.c_int, C.c_int, fp, fp]
        L.oracle_sensor_sample_ray.argtypes = [C.c_void_p, C.c_int, fp, fp, fp, fp, fp]
        L.oracle_emitter_sample_direction.argtypes = [C.c_void_p, fp, C.c_float, C.c_float, fp, fp, fp, fp]
        L.oracle_math.argtypes = [C.c_int, C.c_float, C.c_float]; L.oracle_math.restype = C.c_float
        L.oracle_math_n.argtypes = [C.c_int, C.c_int64, fp, fp, fp]; L.oracle_math_n.restype = None
        _lib = L
    return _lib


_lib_libm = None


def lib_libm():
    """oracle/liboracle_libm.so: the restatement built on glibc's libm instead of csrc/pmath.h (sensitivity measurement only)."""
    global _lib_libm
    if _lib_libm is None:
        path = os.path.join(ROOT, "oracle", "liboracle_libm.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle_libm.so"], stdout=subprocess.DEVNULL)
        L = C.CDLL(path)
        L.oracle_last_error.restype = C.c_char_p
        L.oracle_scene_create.argtypes = [C.POINTER(A.SceneDesc), C.POINTER(C.c_void_p)]
        L.oracle_scene_destroy.argtypes = [C.c_void_p]
        L.oracle_render.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, fp, C.POINTER(A.Stats)]
        L.oracle_math_n.argtypes = [C.c_int, C.c_int64, fp, fp, fp]; L.oracle_math_n.restype = None
        _lib_libm = L
    return _lib_libm


_lib_spectral = None


def lib_spectral():
    """oracle/liboracle_spectral.so: the restatement compiled for the spectral variant (Spectrum<Float, 4>, scalar_spectral semantics)."""
    global _lib_spectral
    if _lib_spectral is None:
        path = os.path.join(ROOT, "oracle", "liboracle_spectral%s.so" % SUFFIX)
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), os.path.basename(path)], stdout=subprocess.DEVNULL)
        L = C.CDLL(path)
        L.oracle_last_error.restype = C.c_char_p
        L.oracle_scene_create.argtypes = [C.POINTER(A.SceneDesc), C.POINTER(C.c_void_p)]
        L.oracle_scene_destroy.argtypes = [C.c_void_p]
        L.oracle_render.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, fp, C.POINTER(A.Stats)]
        L.oracle_spectrum_eval.argtypes = [C.c_void_p, C.c_int, fp, fp]
        L.oracle_volume_eval_spectral.argtypes = [C.c_void_p, C.c_int, fp, fp, fp]
        L.oracle_spectrum_to_xyz.argtypes = [fp, fp, fp]
        L.oracle_spectrum_sample.argtypes = [C.c_void_p, C.c_int, fp, C.c_int, fp, fp]
        L.oracle_set_wavelengths.argtypes = [fp]
        L.oracle_sample_spectral.argtypes = [C.c_void_p, C.c_int32, C.c_uint64] + [fp] * 7 + [fp, C.POINTER(C.c_uint8)]
        L.oracle_bsdf_eval.argtypes = [C.c_void_p, C.c_int, fp, fp, fp, fp]
        assert L.oracle_spec_n() == 4
        _lib_spectral = L
    return _lib_spectral


def _check(status, L=None):
    if status != 0:
        raise RuntimeError((L or lib()).oracle_last_error().decode())


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(fp)


class OracleScene:
    """Scene built from a Mitsuba-style dict (or a ready SceneDesc) and rendered by the CPU restatement."""

    def __init__(self, scene_dict=None, desc=None, keep=None, mono=False, libm=False, spectral=False):
        if desc is None:
            desc, keep = SD.build_scene_desc(scene_dict, mono=mono, spectral=spectral)
        self.desc, self.keep = desc, keep
        self.L = lib_spectral() if spectral else (lib_libm() if libm else lib())
        h = C.c_void_p()
        _check(self.L.oracle_scene_create(C.byref(desc), C.byref(h)), self.L)
        self.h = h
        self.last_stats = None

    def __del__(self):
        try:
            if self.h:
                self.L.oracle_scene_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def render(self, threads=None, shard_index=0, shard_count=1):
        s = self.desc.sensor
        h, w = s.crop_size[1], s.crop_size[0]
        out = np.zeros((h, w, 5 + 2 * self.desc.integrator.bin_count), dtype=np.float32)      # X, Y, Z, A, W + the bins' AOV channels
        st = A.Stats()
        threads = threads or os.cpu_count() or 1
        _check(self.L.oracle_render(self.h, threads, shard_index, shard_count, _p(out), C.byref(st)), self.L)
        self.last_stats = {k: getattr(st, k) for k, _ in A.Stats._fields_}
        return out

    # ---- spectral build only
    def spectrum_eval(self, spectrum, wavelengths):
        w = _f(wavelengths); out = np.zeros(4, np.float32)
        _check(self.L.oracle_spectrum_eval(self.h, spectrum, _p(w), _p(out)), self.L)
        return out

    def spectrum_sample(self, spectrum, samples):
        u = _f(samples); wl = np.zeros(u.size, np.float32); wt = np.zeros(u.size, np.float32)
        _check(self.L.oracle_spectrum_sample(self.h, spectrum, _p(u), int(u.size), _p(wl), _p(wt)), self.L)
        return wl, wt

    def volume_eval_spectral(self, volume, p, wavelengths):
        w = _f(wavelengths); q = _f(p); out = np.zeros(4, np.float32)
        _check(self.L.oracle_volume_eval_spectral(self.h, volume, _p(q), _p(w), _p(out)), self.L)
        return out

    def sample(self, origins, directions, seed_offset=0, wavelengths=None):
        o = _f(origins).reshape(-1, 3); d = _f(directions).reshape(-1, 3)
        n = o.shape[0]
        cols = [np.ascontiguousarray(o[:, i]) for i in range(3)] + [np.ascontiguousarray(d[:, i]) for i in range(3)]
        if wavelengths is not None:                             # spectral build: the rays carry their wavelengths
            w = np.ascontiguousarray(np.broadcast_to(_f(wavelengths).reshape(-1, 4), (n, 4)))
            spec = np.zeros((n, 4), np.float32); valid = np.zeros(n, np.uint8)
            _check(self.L.oracle_sample_spectral(self.h, n, seed_offset, *[_p(c) for c in cols], _p(w), _p(spec),
                                                 valid.ctypes.data_as(C.POINTER(C.c_uint8))), self.L)
            return spec, valid.astype(bool)
        rgb = np.zeros((n, 3), np.float32); valid = np.zeros(n, np.uint8)
        _check(lib().oracle_sample(self.h, n, seed_offset, *[_p(c) for c in cols], _p(rgb), valid.ctypes.data_as(C.POINTER(C.c_uint8))))
        return rgb, valid.astype(bool)

    def ray_intersect(self, o, d, mint=None, maxt=None):
        o = _f(o).reshape(-1, 3); d = _f(d).reshape(-1, 3)
        n = o.shape[0]
        mint = np.full(n, 1500 * 2.0 ** -24, np.float32) if mint is None else _f(mint)
        maxt = np.full(n, np.inf, np.float32) if maxt is None else _f(maxt)
        t = np.zeros(n, np.float32); shape = np.zeros(n, np.int32); prim = np.zeros(n, np.int32)
        p = np.zeros((n, 3), np.float32); nn = np.zeros((n, 3), np.float32)
        _check(lib().oracle_ray_intersect(self.h, n, _p(o), _p(d), _p(mint), _p(maxt), _p(t),
                                          shape.ctypes.data_as(C.POINTER(C.c_int32)), prim.ctypes.data_as(C.POINTER(C.c_int32)), _p(p), _p(nn)))
        return {"t": t, "shape": shape, "prim_index": prim, "p": p, "n": nn}

    def phase_eval(self, phase, wi, wo, p=(0, 0, 0)):
        out = C.c_float()
        _check(lib().oracle_phase_eval(self.h, phase, _p(_f(wi)), _p(_f(p)), _p(_f(wo)), C.byref(out)))
        return out.value

    def phase_sample(self, phase, wi, s1, s2, p=(0, 0, 0)):
        wo = np.zeros(3, np.float32); pdf = C.c_float()
        _check(lib().oracle_phase_sample(self.h, phase, _p(_f(wi)), _p(_f(p)), s1, s2[0], s2[1], _p(wo), C.byref(pdf)))
        return wo, pdf.value

    def phase_eval_component(self, phase, component, wi, wo, p=(0, 0, 0)):
        """PhaseFunction::eval with ctx.component = component; returns (value, component_count)."""
        out = C.c_float(); n = C.c_int()
        _check(lib().oracle_phase_eval_component(self.h, phase, component, _p(_f(wi)), _p(_f(p)), _p(_f(wo)), C.byref(out), C.byref(n)))
        return out.value, n.value

    def phase_sample_component(self, phase, component, wi, s1, s2, p=(0, 0, 0)):
        wo = np.zeros(3, np.float32); pdf = C.c_float()
        _check(lib().oracle_phase_sample_component(self.h, phase, component, _p(_f(wi)), _p(_f(p)), s1, s2[0], s2[1], _p(wo), C.byref(pdf)))
        return wo, pdf.value

    def bsdf_eval(self, bsdf, wi, wo):
        v = np.zeros(3, np.float32); pdf = C.c_float()
        _check(lib().oracle_bsdf_eval(self.h, bsdf, _p(_f(wi)), _p(_f(wo)), _p(v), C.byref(pdf)))
        return v, pdf.value

    def bsdf_sample(self, bsdf, wi, s1, s2):
        wo = np.zeros(3, np.float32); w = np.zeros(3, np.float32); pdf = C.c_float(); st = C.c_uint32()
        _check(lib().oracle_bsdf_sample(self.h, bsdf, _p(_f(wi)), s1, s2[0], s2[1], _p(wo), C.byref(pdf), _p(w), C.byref(st)))
        return wo, pdf.value, w, st.value

    def volume_eval(self, volume, points):
        p = _f(points).reshape(-1, 3)
        out = np.zeros_like(p)
        _check(lib().oracle_volume_eval(self.h, volume, p.shape[0], _p(p), _p(out)))
        return out

    def sensor_sample_ray(self, film_samples, aperture_samples):
        f = _f(film_samples).reshape(-1, 2); a = _f(aperture_samples).reshape(-1, 2)
        n = f.shape[0]
        o = np.zeros((n, 3), np.float32); d = np.zeros((n, 3), np.float32); w = np.zeros((n, 3), np.float32)
        _check(lib().oracle_sensor_sample_ray(self.h, n, _p(f), _p(a), _p(o), _p(d), _p(w)))
        return o, d, w

    def emitter_sample_direction(self, ref_p, u, v):
        d = np.zeros(3, np.float32); spec = np.zeros(3, np.float32); dist = C.c_float(); pdf = C.c_float()
        _check(lib().oracle_emitter_sample_direction(self.h, _p(_f(ref_p)), u, v, _p(d), C.byref(dist), C.byref(pdf), _p(spec)))
        return d, dist.value, pdf.value, spec
