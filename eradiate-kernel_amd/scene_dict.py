"""load_dict: Mitsuba-style scene dictionaries -> the flattened C-ABI scene description.

Mirrors the conventions of the reference's C++ `load_dict`
(/root/reference/src/libcore/python/xml_v.cpp:100-272): key "type" selects the plugin, "id" names an
instance, nested dicts recurse, {"type": "rgb" | "spectrum", "value": ...} become constant colours in
rgb variants (src/libcore/xml.cpp:1073-1111), {"type": "ref", "id": ...} references an earlier
instance, media attach to shapes under the keys "interior" / "exterior"
(src/librender/shape.cpp:58-68), and a property nobody queried is an error (xml_v.cpp:268-269).
Only the plugins of the path / volpath hot path are known.
"""
import ctypes as C
import math
import threading

import numpy as np

from . import _capi as A
from .transform import ScalarTransform4f, coordinate_system, normalize32
from .volume_io import read_volume
from .mesh_io import load_mesh
from .fresolver import file_resolver


def _key_less(a, b):
    """Properties' SortKey (src/libcore/properties.cpp:41-63): lexicographic with numeric suffixes compared
    as numbers, so children named shape_2, shape_10 keep their numeric order."""
    i = 0
    while i < len(a) and i < len(b) and a[i] == b[i]:
        i += 1
    while i > 0 and a[i - 1].isdigit():
        i -= 1
    ta, tb = a[i:], b[i:]
    if ta[:1].isdigit() and tb[:1].isdigit() and ta.isdigit() and tb.isdigit():
        return int(ta) < int(tb)
    return ta < tb


def sorted_items(d):
    """Children in the order Properties::objects() yields them (std::map ordered by SortKey)."""
    import functools
    keys = sorted(d.keys(), key=functools.cmp_to_key(lambda a, b: -1 if _key_less(a, b) else (1 if _key_less(b, a) else 0)))
    return [(k, d[k]) for k in keys]


class Props:
    """Properties bag with 'queried' tracking (src/libcore/properties.cpp)."""

    def __init__(self, d, where):
        if not isinstance(d, dict) or "type" not in d:
            raise RuntimeError("Missing key 'type' in dictionary: %s" % where)
        self.d = d
        self.type = d["type"]
        self.where = where
        self.queried = {"type", "id"}

    def has(self, k):
        return k in self.d

    def get(self, k, default=None):
        self.queried.add(k)
        return self.d.get(k, default)

    def finish(self):
        left = [k for k in self.d if k not in self.queried]
        if left:
            raise RuntimeError("Error while loading \"%s\": unreferenced propert%s %s in plugin of type \"%s\""
                               % (self.where, "ies" if len(left) > 1 else "y", left, self.type))


def _xf(t):
    r = A.Transform()
    if t is None:
        t = ScalarTransform4f()
    if not isinstance(t, ScalarTransform4f):
        t = ScalarTransform4f(np.asarray(t, dtype=np.float32))
    r.matrix[:] = t.matrix.reshape(-1).tolist()
    r.inverse_transpose[:] = t.inverse_transpose.reshape(-1).tolist()
    return r


_MONO = False      # set by build_scene_desc(mono=True): colours become their luminance (src/spectra/srgb.cpp in *_mono variants)
_BUILD_LOCK = threading.RLock()
_SPECTRAL = None   # set by build_scene_desc(spectral=True): the SceneBuilder that collects the scene's spectra


WAVELENGTH_MIN, WAVELENGTH_MAX = 280.0, 2400.0       # MTS_WAVELENGTH_MIN / MAX, include/mitsuba/core/spectrum.h:15-21


def _spectrum(v, where, default=None, emitter=False):
    """Spectral variant: a colour parameter -> index of its spectrum record (mts_spectrum).
    float -> `uniform` (Properties::texture, properties.h:275-296); {"type": "uniform" | "regular" | "d65"} -> the plugin of that
    name (src/spectra/*.cpp; d65 expands to regular, d65.cpp:52-71); {"type": "spectrum", "value": c} -> uniform, or d65 scaled by c
    inside an emitter (create_texture_from_spectrum, xml.cpp:1087-1111); a missing emitter spectrum is D65 (directional.cpp:49,
    area.cpp:39, constant.cpp:33, point.cpp:47).  rgb colours need the sRGB upsampling model, whose data (ext/rgb2spec) is absent."""
    b = _SPECTRAL
    if v is None:
        v = {"type": "d65"} if (emitter and default is None) else default
    rec = A.Spectrum()
    rec.lambda_min, rec.lambda_max = WAVELENGTH_MIN, WAVELENGTH_MAX
    if isinstance(v, dict):
        p = Props(v, where)
        t = p.type
        if t == "spectrum":
            val = _spectrum_pairs(v, where)                                 # inline pairs or spectrum_from_file (xml.cpp:823-851)
            if p.has("filename"):
                p.get("filename")
            if val is None:
                val = p.get("value", 1.0)
            elif p.has("value"):
                p.get("value")
            if isinstance(val, (list, tuple)):
                # create_texture_from_spectrum, xml.cpp:1113-1150 (spectral mode): values scaled by MTS_CIE_Y_NORMALIZATION inside an
                # emitter; `regular` when the wavelengths are equidistant (to math::Epsilon), `irregular` otherwise
                pairs = np.asarray(val, np.float64).reshape(-1, 2)
                f = np.float32
                wl = pairs[:, 0].astype(np.float32)
                vals = (pairs[:, 1].astype(np.float32) * (f(1.0 / 106.7502593994140625) if emitter else f(1.0))).astype(np.float32)
                dist = np.diff(wl)
                if (dist < 0).any():
                    raise RuntimeError("Wavelengths must be specified in increasing order!")
                p.finish()
                if wl.size >= 2 and (np.abs(dist - dist[0]) <= 2.0 ** -24).all():
                    return _spectrum({"type": "regular", "lambda_min": float(wl[0]), "lambda_max": float(wl[-1]), "values": vals}, where)
                return _spectrum({"type": "irregular", "wavelengths": wl, "values": vals}, where)
            p.finish()
            return _spectrum({"type": "d65", "scale": float(val)} if emitter else {"type": "uniform", "value": float(val)}, where)
        if t == "uniform":
            rec.type = A.SPECTRUM_UNIFORM
            rec.value = float(p.get("value", 1.0))
            if p.has("lambda_min"):
                rec.lambda_min = max(float(p.get("lambda_min")), WAVELENGTH_MIN)
            if p.has("lambda_max"):
                rec.lambda_max = min(float(p.get("lambda_max")), WAVELENGTH_MAX)
            if not rec.lambda_min < rec.lambda_max:
                raise RuntimeError("UniformSpectrum: 'lambda_min' must be less than 'lambda_max'")
        elif t in ("regular", "d65"):
            rec.type = A.SPECTRUM_REGULAR
            if t == "d65":
                from .spectra_data import D65, D65_NORMALIZATION
                f = np.float32
                scale = f(f(p.get("scale", 1.0)) * f(D65_NORMALIZATION))                  # d65.cpp:56-57: m_scale *= 1.f / 10568.f
                values = (np.asarray(D65, np.float32) * scale).astype(np.float32)         # :64-65
                rec.lambda_min, rec.lambda_max = 360.0, 830.0
            else:
                if not (p.has("lambda_min") and p.has("lambda_max")):
                    raise RuntimeError("Property \"lambda_min\" has not been specified!" if not p.has("lambda_min") else "Property \"lambda_max\" has not been specified!")
                rec.lambda_min, rec.lambda_max = float(p.get("lambda_min")), float(p.get("lambda_max"))
                vals = p.get("values")
                if isinstance(vals, str):
                    vals = [float(x) for x in vals.replace(",", " ").split()]
                values = np.asarray(vals, np.float32).reshape(-1)
            values = np.ascontiguousarray(values, np.float32)
            b.keep.append(values)
            rec.values = values.ctypes.data_as(A.fp)
            rec.count = int(values.size)
        elif t in ("irregular", "discrete"):
            # spectra/irregular.cpp:33-63 (strings of comma-separated numbers, or arrays), spectra/discrete.cpp:45-100
            def numbers(key, default=None):
                x = p.get(key, default)
                if x is None:
                    raise RuntimeError("Property \"%s\" has not been specified!" % key)
                if isinstance(x, str):
                    try:
                        x = [float(tok) for tok in x.replace(",", " ").split()]
                    except ValueError as e:
                        raise RuntimeError("Could not parse floating point value '%s'" % str(e).split("'")[-2])
                return np.ascontiguousarray(np.asarray(x, np.float64).reshape(-1), np.float32)
            wl = numbers("wavelengths")
            if t == "irregular":
                rec.type = A.SPECTRUM_IRREGULAR
                values = numbers("values")
                if values.size != wl.size:
                    raise RuntimeError("IrregularSpectrum: 'wavelengths' and 'values' parameters must have the same size!")
                pmf = None
            else:
                rec.type = A.SPECTRUM_DISCRETE
                values, pmf = numbers("values", "1"), numbers("pmf", "1")
                if values.size == 1:
                    values = np.full(wl.size, values[0], np.float32)
                if pmf.size == 1:
                    pmf = np.full(wl.size, pmf[0], np.float32)
                if values.size != wl.size:
                    raise RuntimeError("DiscreteSpectrum: 'wavelengths' and 'values' parameters must have the same size!")
                if pmf.size != wl.size:
                    raise RuntimeError("DiscreteSpectrum: 'wavelengths' and 'pmf' parameters must have the same size!")
                b.keep.append(pmf)
                rec.pmf = pmf.ctypes.data_as(A.fp)
            b.keep.append(wl); b.keep.append(values)
            rec.wavelengths = wl.ctypes.data_as(A.fp)
            rec.values = values.ctypes.data_as(A.fp)
            rec.count = int(wl.size)
        elif t in ("rgb", "srgb", "srgb_d65"):
            raise RuntimeError("rgb colours cannot be used in the spectral variant: the sRGB upsampling model needs the coefficient data of "
                               "ext/rgb2spec, absent here; give a 'uniform' or 'regular' spectrum instead (%s)" % where)
        else:
            raise RuntimeError("Unsupported spectrum plugin \"%s\" in %s" % (t, where))
        p.finish()
    else:
        c = np.asarray(v, dtype=np.float32).reshape(-1)
        if c.size != 1:
            raise RuntimeError("Cannot interpret %r as a spectrum in %s (spectral variant)" % (v, where))
        rec.type = A.SPECTRUM_UNIFORM
        rec.value = float(c[0])
    b.spectra.append(rec)
    return len(b.spectra) - 1


def spectrum_from_file(filename):
    """src/libcore/spectrum.cpp:9-39: `wavelength value` per line, empty lines and lines starting with '#' skipped, anything after the
    pair is an error; the name goes through the FileResolver."""
    import os
    path = file_resolver().resolve(filename)
    if not os.path.exists(path):
        raise RuntimeError("\"%s\": file does not exist!" % path)
    pairs = []
    with open(path) as f:
        for line in f:
            line = line.rstrip("\r\n")
            if len(line) == 0 or line[0] == "#":
                continue
            tok = line.split()
            if len(tok) > 2:
                raise RuntimeError("\"%s\": excess tokens after wavlengths-value pair in file:\n%s!" % (path, line))
            if len(tok) < 2:                                 # `iss >> value` fails: the reference keeps whatever the variables held; not reproduced
                raise RuntimeError("\"%s\": could not parse wavelength-value pair:\n%s" % (path, line))
            pairs.append((float(np.float32(tok[0])), float(np.float32(tok[1]))))
    return pairs


def _spectrum_pairs(v, where):
    """The wavelength:value pairs of a {"type": "spectrum"} dictionary (inline string / list, or a file), or None for a constant."""
    if "filename" in v:
        if "value" in v:
            raise RuntimeError("'spectrum' tag requires one of \"value\" or \"filename\" attributes")      # xml.cpp:815-816
        return spectrum_from_file(v["filename"])
    val = v.get("value", 1.0)
    if isinstance(val, str) and ":" in val:                                 # "400:0.1, 500:0.2, ..." (xml.cpp:823-846)
        try:
            return [tuple(float(x) for x in tok.split(":")) for tok in val.replace(",", " ").split()]
        except ValueError:
            raise RuntimeError("Could not parse wavelength:value pairs in %s" % where)
    if isinstance(val, (list, tuple, np.ndarray)):
        return [tuple(x) for x in np.asarray(val, np.float64).reshape(-1, 2)]
    return None


def _cie1931_xyz(x):
    """core/spectrum.h:148-178 in float32: linear interpolation of the 5 nm tables over 360 .. 830 nm, zero outside."""
    from .spectra_data import CIE_1931
    f = np.float32
    t = f(f(f(x) - f(360.0)) * f(f(94) / f(f(830.0) - f(360.0))))
    if not (x >= 360.0 and x <= 830.0):
        return np.zeros(3, np.float32)
    i0 = int(min(max(int(t), 0), 93)); i1 = i0 + 1
    w1 = f(t - f(i0)); w0 = f(f(1.0) - w1)
    return np.array([f(f(w0 * f(CIE_1931[c][i0])) + f(w1 * f(CIE_1931[c][i1]))) for c in range(3)], np.float32)       # fmadd(w0, v0, w1 * v1): differs in the last bit at most


def spectrum_to_rgb(wavelengths, values, bounded=True):
    """src/libcore/spectrum.cpp:41-89: pre-integration of a tabulated spectrum against the CIE 1931 curves (1000 steps over
    MTS_WAVELENGTH_MIN .. MAX = 280 .. 2400 nm, this fork's range), XYZ -> linear sRGB, clamped (to [0, 1] when `bounded`)."""
    f = np.float32
    wl = [f(w) for w in wavelengths]; vs = [f(v) for v in values]
    color = np.zeros(3, np.float32)
    steps = 1000
    for i in range(steps):
        x = f(f(WAVELENGTH_MIN) + f(f(i) / f(steps - 1)) * f(f(WAVELENGTH_MAX) - f(WAVELENGTH_MIN)))
        if x < wl[0] or x > wl[-1]:
            continue
        # math::find_interval(size, pred): the last index with pred true, clamped to [0, size - 2]
        index = 0
        for k in range(len(wl)):
            if wl[k] <= x:
                index = k
        index = min(max(index, 0), len(wl) - 2)
        x0, x1, y0, y1 = wl[index], wl[index + 1], vs[index], vs[index + 1]
        y = f(f(f(f(f(x * y0) - f(x1 * y0)) - f(x * y1)) + f(x0 * y1)) / f(x0 - x1))
        color = (color + _cie1931_xyz(x) * y).astype(np.float32)
    color = (color * f(f(f(WAVELENGTH_MAX) - f(WAVELENGTH_MIN)) / f(steps))).astype(np.float32)
    m = np.array([[3.240479, -1.537150, -0.498535], [-0.969256, 1.875991, 0.041556], [0.055648, -0.204043, 1.057311]], np.float32)   # xyz_to_srgb, spectrum.h:229-235
    color = (m @ color).astype(np.float32)
    color = np.clip(color, 0.0, 1.0) if bounded else np.maximum(color, 0.0)
    return tuple(float(c) for c in color)


CIE_Y_NORMALIZATION = 1.0 / 106.856895                     # MTS_CIE_Y_NORMALIZATION, core/spectrum.h:136
_UNBOUNDED_NAMES = ("eta", "k", "int_ior", "ext_ior")      # is_unbounded_spectrum, xml.cpp:142-144


def _color(v, where, default=None, emitter=False):
    """float | [r,g,b] | {"type":"rgb","value":..} | {"type":"spectrum"/"uniform","value":x} -> (r,g,b)."""
    c = _color_rgb(v, where, default, emitter)
    if _MONO and not (c[0] == c[1] == c[2]):
        f = np.float32
        lum = float(f(f(f(c[0]) * f(0.212671) + f(c[1]) * f(0.715160)) + f(c[2]) * f(0.072169)))      # spectrum.h:246-248
        return (lum, lum, lum)
    return c


def _color_rgb(v, where, default=None, emitter=False):
    if v is None:
        v = default
    if isinstance(v, dict):
        t = v.get("type")
        if t == "rgb":
            if len(v) != 2:
                raise RuntimeError("'rgb' dictionary should always contain 2 entries ('type' and 'value'), got %d." % len(v))
            c = np.asarray(v["value"], dtype=np.float32).reshape(-1)
            if c.size == 1:
                c = np.repeat(c, 3)
            return tuple(float(x) for x in c[:3])
        if t == "spectrum":
            pairs = _spectrum_pairs(v, where)
            if pairs is not None:
                # create_texture_from_spectrum in the non-spectral modes (xml.cpp:1113-1170): values x MTS_CIE_Y_NORMALIZATION, wavelengths
                # in increasing order, pre-integrated against the CIE curves -> `srgb` (bounded) / `srgb_d65` (emitters) colour
                f = np.float32
                wl = [p_[0] for p_ in pairs]
                if any(b_ - a_ < 0 for a_, b_ in zip(wl, wl[1:])):
                    raise RuntimeError("Wavelengths must be specified in increasing order!")
                if len(wl) < 2:
                    raise RuntimeError("a tabulated spectrum needs at least two wavelength:value pairs: %s" % where)
                name = where.rsplit(".", 1)[-1]
                return spectrum_to_rgb(wl, [f(f(p_[1]) * f(CIE_Y_NORMALIZATION)) for p_ in pairs], bounded=not (emitter or name in _UNBOUNDED_NAMES))
            return (float(v.get("value", 1.0)),) * 3
        if t == "uniform":
            val = v.get("value", 1.0)
            if isinstance(val, (list, tuple, str)):
                raise RuntimeError("'uniform' takes one value: %s" % where)
            return (float(val),) * 3
        if t in ("srgb", "srgb_d65"):
            c = np.asarray(v.get("color"), dtype=np.float32).reshape(-1)
            return tuple(float(x) for x in c[:3])
        raise RuntimeError("Unsupported texture / spectrum plugin \"%s\" in %s (rgb constants only)" % (t, where))
    c = np.asarray(v, dtype=np.float32).reshape(-1)
    if c.size == 1:
        return (float(c[0]),) * 3
    if c.size == 3:
        return tuple(float(x) for x in c)
    raise RuntimeError("Cannot interpret %r as a colour in %s" % (v, where))


def parse_fov(p, aspect):
    """src/librender/sensor.cpp:113-167."""
    if p.has("fov") and p.has("focal_length"):
        raise RuntimeError("Please specify either a focal length ('focal_length') or a field of view ('fov')!")
    if p.has("fov"):
        fov = float(p.get("fov"))
        axis = str(p.get("fov_axis", "x")).lower()
        if axis == "smaller":
            axis = "y" if aspect > 1 else "x"
        elif axis == "larger":
            axis = "x" if aspect > 1 else "y"
    else:
        f = str(p.get("focal_length", "50mm"))
        if f.endswith("mm"):
            f = f[:-2]
        value = float(f)
        fov = 2.0 * math.degrees(math.atan(math.sqrt(36 * 36 + 24 * 24) / (2.0 * value)))
        axis = "diagonal"
    if axis == "x":
        result = fov
    elif axis == "y":
        result = math.degrees(2.0 * math.atan(math.tan(0.5 * math.radians(fov)) * aspect))
    elif axis == "diagonal":
        diagonal = 2.0 * math.tan(0.5 * math.radians(fov))
        width = diagonal / math.sqrt(1.0 + 1.0 / (aspect * aspect))
        result = math.degrees(2.0 * math.atan(width * 0.5))
    else:
        raise RuntimeError("The 'fov_axis' parameter must be set to one of 'smaller', 'larger', 'diagonal', 'x', or 'y'!")
    if result <= 0.0 or result >= 180.0:
        raise RuntimeError("The horizontal field of view must be in the range [0, 180]!")
    return float(np.float32(result))


SHAPE_PLUGINS = ("rectangle", "cube", "sphere", "mesh", "obj", "ply", "disk")


class SceneBuilder:
    """Accumulates plugin records; keeps every numpy buffer alive for the lifetime of the description."""

    def __init__(self):
        self.volumes, self.phases, self.media, self.bsdfs, self.shapes, self.emitters = [], [], [], [], [], []
        self.sensor = None
        self.sensor_dict = None
        self.integrator = None
        self.keep = []
        self.instances = {}      # id -> (kind, index)
        self.info = {}

    # ------------------------------------------------------------ helpers
    def _register(self, d, kind, index):
        if isinstance(d, dict) and "id" in d:
            self.instances[d["id"]] = (kind, index)

    def _resolve_ref(self, d, kind):
        if isinstance(d, dict) and d.get("type") == "ref":
            for k in d:
                if k not in ("type", "id"):
                    raise RuntimeError("Unexpected key in ref dictionary: %s" % k)
            rid = d.get("id")
            if rid not in self.instances:
                raise RuntimeError("Referenced id \"%s\" not found" % rid)
            k, i = self.instances[rid]
            if k != kind:
                raise RuntimeError("Referenced id \"%s\" is a %s, expected a %s" % (rid, k, kind))
            return i
        return None

    # ------------------------------------------------------------ volumes
    def add_volume(self, v, where, default=None):
        r = self._resolve_ref(v, "volume")
        if r is not None:
            return r
        rec = A.Volume()
        rec.to_world = _xf(None)
        rec.filter_type = A.FILTER_TRILINEAR
        rec.wrap_mode = A.WRAP_CLAMP
        if v is None:
            v = default
        rec.value_spectrum = -1
        if isinstance(v, dict) and v.get("type") in ("gridvolume", "constvolume", "gridvolume_spectral"):
            p = Props(v, where)
            rec.to_world = _xf(p.get("to_world"))
            if p.type == "constvolume":
                rec.type = A.VOLUME_CONST
                # constant3d.cpp: "color" texture; load_dict users pass "value" (xml.cpp spectrum shorthand)
                val = p.get("value", p.get("color", 1.0))
                if _SPECTRAL is not None:
                    rec.value_spectrum = _spectrum(val, where)
                else:
                    rec.value[:] = _color(val, where)
            else:
                rec.type = A.VOLUME_GRID
                if p.type == "gridvolume_spectral":                      # src/textures/gridvolume_spectral.cpp:84-135,186-190
                    if _SPECTRAL is None:
                        raise RuntimeError("This volume data source can only be used with a spectral variant!")
                    rec.type = A.VOLUME_GRID_SPECTRAL
                    st = str(p.get("spectrum_type", "regular"))
                    if st != "regular":
                        raise RuntimeError("Invalid spectrum type \"%s\", must be \"regular\"!" % st)
                    if not p.has("lambda_min"):
                        raise RuntimeError("Property \"lambda_min\" has not been specified!")
                    if not p.has("lambda_max"):
                        raise RuntimeError("Property \"lambda_max\" has not been specified!")
                    rec.lambda_min, rec.lambda_max = float(p.get("lambda_min")), float(p.get("lambda_max"))
                if p.has("filename"):
                    data, meta = read_volume(file_resolver().resolve(p.get("filename")))
                    rec.file_bbox_min[:] = meta["bbox_min"]
                    rec.file_bbox_max[:] = meta["bbox_max"]
                elif p.has("data"):
                    # in-memory grid: array of shape (nz, ny, nx[, channels]) (extension; the reference reads a file)
                    data = np.asarray(p.get("data"), dtype=np.float32)
                    if data.ndim == 3:
                        data = data[..., None]
                    rec.file_bbox_min[:] = (0.0, 0.0, 0.0)
                    rec.file_bbox_max[:] = (1.0, 1.0, 1.0)
                else:
                    raise RuntimeError("gridvolume: property \"filename\" has not been specified!")
                data = np.ascontiguousarray(data, dtype=np.float32)
                if data.ndim != 4:
                    raise RuntimeError("gridvolume: data must have shape (nz, ny, nx, channels)")
                mono_max = None
                if _MONO and data.shape[3] == 3:
                    # grid3d.cpp:178-179: *_mono variants return the luminance of the interpolated colour. Luminance is
                    # linear, so it is applied per voxel here; the majorant stays the file's maximum (volume metadata).
                    mono_max = float(data.max()) if data.size else 0.0
                    f = np.float32
                    data = np.ascontiguousarray(((data[..., 0] * f(0.212671) + data[..., 1] * f(0.715160))
                                                 + data[..., 2] * f(0.072169))[..., None], dtype=np.float32)
                self.keep.append(data)
                rec.data = data.ctypes.data_as(A.fp)
                rec.nz, rec.ny, rec.nx, rec.channels = data.shape
                ft = str(p.get("filter_type", "trilinear"))
                if p.type == "gridvolume_spectral" and ft != "trilinear":
                    raise RuntimeError("Invalid filter type \"%s\", must be \"trilinear\"!" % ft)
                if ft not in ("nearest", "trilinear"):
                    raise RuntimeError("Invalid filter type \"%s\", must be one of: \"nearest\" or \"trilinear\"!" % ft)
                rec.filter_type = A.FILTER_NEAREST if ft == "nearest" else A.FILTER_TRILINEAR
                wm = str(p.get("wrap_mode", "clamp"))
                if wm not in ("repeat", "mirror", "clamp"):
                    raise RuntimeError("Invalid wrap mode \"%s\", must be one of: \"repeat\", \"mirror\", or \"clamp\"!" % wm)
                rec.wrap_mode = {"repeat": A.WRAP_REPEAT, "mirror": A.WRAP_MIRROR, "clamp": A.WRAP_CLAMP}[wm]
                rec.use_grid_bbox = int(bool(p.get("use_grid_bbox", False)))
                p.get("raw", False)
                if p.has("max_value"):
                    rec.has_max_value = 1
                    rec.max_value = float(p.get("max_value"))
                elif mono_max is not None:
                    rec.has_max_value = 1
                    rec.max_value = mono_max
            p.finish()
        else:
            # float / rgb given where a volume is expected -> constvolume (properties.h:319-366)
            rec.type = A.VOLUME_CONST
            if _SPECTRAL is not None:
                rec.value_spectrum = _spectrum(v, where)
            else:
                rec.value[:] = _color(v, where)
        self.volumes.append(rec)
        self._register(v, "volume", len(self.volumes) - 1)
        return len(self.volumes) - 1

    # ------------------------------------------------------------ phase functions
    def add_phase(self, d, where):
        r = self._resolve_ref(d, "phase")
        if r is not None:
            return r
        rec = A.Phase()
        rec.child[:] = (-1, -1)
        rec.weight_volume = -1
        if d is None:
            rec.type = A.PHASE_ISOTROPIC
        else:
            p = Props(d, where)
            if p.type == "isotropic":
                rec.type = A.PHASE_ISOTROPIC
            elif p.type == "hg":
                rec.type = A.PHASE_HG
                rec.g = float(p.get("g", 0.8))
            elif p.type == "rayleigh":
                rec.type = A.PHASE_RAYLEIGH
            elif p.type == "tabphase":
                rec.type = A.PHASE_TABULATED
                vals = p.get("values")
                if not isinstance(vals, str):
                    raise RuntimeError("'values' must be a string")
                arr = np.array([float(s) for s in vals.replace(",", " ").split()], dtype=np.float32)
                self.keep.append(arr)
                rec.tab_values = arr.ctypes.data_as(A.fp)
                rec.tab_count = arr.size
            elif p.type == "blendphase":
                rec.type = A.PHASE_BLEND
                children = []
                for k, v in sorted_items(d):  # nested phase functions in Properties order (blendphase.cpp:33-43)
                    if isinstance(v, dict) and v.get("type") in ("isotropic", "hg", "rayleigh", "tabphase", "blendphase", "ref") and k != "weight":
                        p.queried.add(k)
                        children.append(self.add_phase(v, where + "." + k))
                if len(children) != 2:
                    raise RuntimeError("BlendPhase: Two child phase functions must be specified!")
                rec.child[:] = children
                if not p.has("weight"):
                    raise RuntimeError("Property \"weight\" has not been specified!")
                rec.weight_volume = self.add_volume(p.get("weight"), where + ".weight")
            else:
                raise RuntimeError("Unknown / unsupported phase function plugin \"%s\"" % p.type)
            p.finish()
        self.phases.append(rec)
        self._register(d, "phase", len(self.phases) - 1)
        return len(self.phases) - 1

    # ------------------------------------------------------------ media
    def add_medium(self, d, where):
        r = self._resolve_ref(d, "medium")
        if r is not None:
            return r
        p = Props(d, where)
        rec = A.Medium()
        if p.type == "homogeneous":
            rec.type = A.MEDIUM_HOMOGENEOUS
        elif p.type == "heterogeneous":
            rec.type = A.MEDIUM_HETEROGENEOUS
        else:
            raise RuntimeError("Unknown / unsupported medium plugin \"%s\"" % p.type)
        rec.albedo_volume = self.add_volume(p.get("albedo"), where + ".albedo", default=0.75)
        rec.sigma_t_volume = self.add_volume(p.get("sigma_t"), where + ".sigma_t", default=1.0)
        rec.scale = float(p.get("scale", 1.0))
        rec.has_spectral_extinction = int(bool(p.get("has_spectral_extinction", True)))
        rec.sample_emitters = int(bool(p.get("sample_emitters", True)))
        phase = None
        for k, v in sorted_items(d):         # any nested phase function (medium.cpp:13-22)
            if isinstance(v, dict) and v.get("type") in ("isotropic", "hg", "rayleigh", "tabphase", "blendphase") \
                    or (isinstance(v, dict) and v.get("type") == "ref" and self.instances.get(v.get("id"), ("",))[0] == "phase"):
                if phase is not None:
                    raise RuntimeError("Only a single phase function can be specified per medium")
                p.queried.add(k)
                phase = self.add_phase(v, where + "." + k)
        rec.phase = phase if phase is not None else self.add_phase(None, where + ".phase")
        p.finish()
        self.media.append(rec)
        self._register(d, "medium", len(self.media) - 1)
        return len(self.media) - 1

    # ------------------------------------------------------------ BSDFs
    def add_bsdf(self, d, where):
        r = self._resolve_ref(d, "bsdf")
        if r is not None:
            return r
        p = Props(d, where)
        rec = A.Bsdf()
        rec.spectrum[:] = [-1] * 6
        spectral = _SPECTRAL is not None
        def colour(field, slot, key, default):                           # rgb triple, or (spectral variant) the index of the spectrum
            if spectral:
                rec.spectrum[slot] = _spectrum(p.get(key), where + "." + key, default=default)
            else:
                getattr(rec, field)[:] = _color(p.get(key), where, default=default)
        if p.type == "diffuse":
            rec.type = A.BSDF_DIFFUSE
            colour("reflectance", 0, "reflectance", 0.5)
        elif p.type == "null":
            rec.type = A.BSDF_NULL
        elif p.type == "bilambertian":                                   # src/bsdfs/bilambertian.cpp:51-60
            rec.type = A.BSDF_BILAMBERTIAN
            colour("reflectance", 0, "reflectance", 0.5)
            colour("transmittance", 5, "transmittance", 0.5)
        elif p.type == "rpv":
            rec.type = A.BSDF_RPV
            colour("rho_0", 1, "rho_0", 0.1)
            colour("g", 3, "g", 0.0)
            colour("k", 2, "k", 0.1)
            if p.has("rho_c"):
                colour("rho_c", 4, "rho_c", None)
            elif spectral:
                rec.spectrum[4] = rec.spectrum[1]                        # rpv.cpp:75-79: rho_c defaults to rho_0
            else:
                rec.rho_c[:] = tuple(rec.rho_0)
        else:
            raise RuntimeError("Unknown / unsupported BSDF plugin \"%s\"" % p.type)
        p.finish()
        self.bsdfs.append(rec)
        self._register(d, "bsdf", len(self.bsdfs) - 1)
        return len(self.bsdfs) - 1

    # ------------------------------------------------------------ shapes
    def make_shape(self, d, where, in_scene=True):
        p = Props(d, where)
        rec = A.Shape()
        rec.bsdf = rec.interior_medium = rec.exterior_medium = rec.emitter = -1
        rec.radius = 1.0
        rec.to_world = _xf(p.get("to_world"))
        if p.type == "rectangle":
            rec.type = A.SHAPE_RECTANGLE
            rec.flip_normals = int(bool(p.get("flip_normals", False)))
        elif p.type == "disk":                                           # src/shapes/disk.cpp:74-81
            rec.type = A.SHAPE_DISK
            rec.flip_normals = int(bool(p.get("flip_normals", False)))
        elif p.type == "cube":
            rec.type = A.SHAPE_CUBE
        elif p.type == "sphere":
            rec.type = A.SHAPE_SPHERE
            rec.flip_normals = int(bool(p.get("flip_normals", False)))
            rec.center[:] = tuple(float(x) for x in np.asarray(p.get("center", (0, 0, 0)), dtype=np.float32))
            rec.radius = float(p.get("radius", 1.0))
        elif p.type in ("mesh", "obj", "ply"):
            # "mesh": in-memory triangle mesh (extension); "obj" / "ply": src/shapes/obj.cpp, src/shapes/ply.cpp via mesh_io.py,
            # which applies to_world at load time like the reference's loaders
            rec.type = A.SHAPE_MESH
            if p.type == "mesh":
                arrays = {k: p.get(k) for k in ("vertex_positions", "faces", "vertex_normals", "vertex_texcoords") if p.has(k)}
            else:
                arrays = load_mesh(p.type, file_resolver().resolve(p.get("filename")), p.get("to_world"), bool(p.get("face_normals", False)),
                                   bool(p.get("flip_tex_coords", True)) if p.type == "obj" else True)
                rec.to_world = _xf(None)

            class _Arr:                                                    # the parsed arrays, read like properties below
                def has(self, k): return k in arrays
                def get(self, k): return arrays[k]
            p_arr = _Arr()
            pos = np.ascontiguousarray(arrays["vertex_positions"], dtype=np.float32).reshape(-1, 3)
            faces = np.ascontiguousarray(arrays["faces"], dtype=np.uint32).reshape(-1, 3)
            self.keep += [pos, faces]
            rec.vertex_positions = pos.ctypes.data_as(A.fp)
            rec.faces = faces.ctypes.data_as(C.POINTER(C.c_uint32))
            rec.vertex_count, rec.face_count = pos.shape[0], faces.shape[0]
            if p_arr.has("vertex_normals"):
                nor = np.ascontiguousarray(p_arr.get("vertex_normals"), dtype=np.float32).reshape(-1, 3)
                self.keep.append(nor)
                rec.vertex_normals = nor.ctypes.data_as(A.fp)
            if p_arr.has("vertex_texcoords"):
                uv = np.ascontiguousarray(p_arr.get("vertex_texcoords"), dtype=np.float32).reshape(-1, 2)
                self.keep.append(uv)
                rec.vertex_texcoords = uv.ctypes.data_as(A.fp)
        else:
            raise RuntimeError("Unknown / unsupported shape plugin \"%s\"" % p.type)
        emitter_dict = None
        for k, v in sorted_items(d):
            if k in p.queried or not isinstance(v, dict):
                continue
            t = v.get("type")
            kind = self.instances.get(v.get("id"), ("",))[0] if t == "ref" else None
            if t in ("diffuse", "null", "rpv", "bilambertian") or kind == "bsdf":
                if rec.bsdf >= 0:
                    raise RuntimeError("Only a single BSDF child object can be specified per shape.")
                p.queried.add(k)
                rec.bsdf = self.add_bsdf(v, where + "." + k)
            elif t in ("homogeneous", "heterogeneous") or kind == "medium":
                p.queried.add(k)
                if k == "interior":
                    rec.interior_medium = self.add_medium(v, where + "." + k)
                elif k == "exterior":
                    rec.exterior_medium = self.add_medium(v, where + "." + k)
                else:
                    self.add_medium(v, where + "." + k)   # shape.cpp:58-68: other names are ignored
            elif t == "area":
                if emitter_dict is not None:
                    raise RuntimeError("Only a single Emitter child object can be specified per shape.")
                p.queried.add(k)
                emitter_dict = v
        p.finish()
        if not in_scene:
            return rec, None
        return rec, emitter_dict

    def add_shape(self, d, where):
        rec, emitter_dict = self.make_shape(d, where)
        self.shapes.append(rec)
        idx = len(self.shapes) - 1
        if emitter_dict is not None:
            ep = Props(emitter_dict, where + ".emitter")
            e = A.Emitter()
            e.type = A.EMITTER_AREA
            e.to_world = _xf(None)
            self._emitter_colour(e, ep.get("radiance"), where)
            e.shape = idx
            ep.finish()
            self.emitters.append(e)
            self.shapes[idx].emitter = len(self.emitters) - 1
        self._register(d, "shape", idx)
        return idx

    # ------------------------------------------------------------ emitters
    @staticmethod
    def _emitter_colour(e, v, where):
        e.radiance_spectrum = -1
        if _SPECTRAL is not None:
            e.radiance_spectrum = _spectrum(v, where, emitter=True)
        else:
            e.radiance[:] = _color(v, where, default=1.0, emitter=True)

    def add_emitter(self, d, where):
        p = Props(d, where)
        e = A.Emitter()
        e.shape = -1
        if p.type == "directional":
            e.type = A.EMITTER_DIRECTIONAL
            if p.has("direction"):
                if p.has("to_world"):
                    raise RuntimeError("Only one of the parameters 'direction' and 'to_world' can be specified at the same time!'")
                direction = np.asarray(p.get("direction"), dtype=np.float32)
                direction = normalize32(direction)
                up, _ = coordinate_system(direction)                       # directional.cpp:55-60
                e.to_world = _xf(ScalarTransform4f.look_at([0, 0, 0], direction, up))
            else:
                e.to_world = _xf(p.get("to_world"))
            self._emitter_colour(e, p.get("irradiance"), where)
        elif p.type == "constant":
            e.type = A.EMITTER_CONSTANT
            e.to_world = _xf(None)
            self._emitter_colour(e, p.get("radiance"), where)
        elif p.type == "point":                                             # point.cpp:45-58
            e.type = A.EMITTER_POINT
            if p.has("position"):
                if p.has("to_world"):
                    raise RuntimeError("Only one of the parameters 'position' and 'to_world' can be specified at the same time!'")
                e.to_world = _xf(ScalarTransform4f.translate(np.asarray(p.get("position"), dtype=np.float32)))
            else:
                e.to_world = _xf(p.get("to_world"))
            self._emitter_colour(e, p.get("intensity"), where)
        elif p.type == "area":
            raise RuntimeError("Can't sample from an area emitter without an associated Shape.")
        else:
            raise RuntimeError("Unknown / unsupported emitter plugin \"%s\"" % p.type)
        p.finish()
        self.emitters.append(e)
        return len(self.emitters) - 1

    # ------------------------------------------------------------ sensor
    def set_sensor(self, d, where):
        p = Props(d, where)
        s = A.Sensor()
        s.medium = -1
        s.distant_target_shape.bsdf = s.distant_target_shape.interior_medium = -1
        s.distant_target_shape.exterior_medium = s.distant_target_shape.emitter = -1
        s.distant_origin_shape.bsdf = s.distant_origin_shape.interior_medium = -1
        s.distant_origin_shape.exterior_medium = s.distant_origin_shape.emitter = -1
        # film (src/librender/film.cpp:14-50, src/films/hdrfilm.cpp)
        fd = p.get("film", {"type": "hdrfilm"})
        fp_ = Props(fd, where + ".film")
        if fp_.type != "hdrfilm":
            raise RuntimeError("Unknown / unsupported film plugin \"%s\"" % fp_.type)
        s.film_width = int(fp_.get("width", 768))
        s.film_height = int(fp_.get("height", 576))
        s.crop_offset[:] = (int(fp_.get("crop_offset_x", 0)), int(fp_.get("crop_offset_y", 0)))
        s.crop_size[:] = (int(fp_.get("crop_width", s.film_width)), int(fp_.get("crop_height", s.film_height)))
        if (s.crop_offset[0] < 0 or s.crop_offset[1] < 0 or s.crop_size[0] <= 0 or s.crop_size[1] <= 0 or
                s.crop_offset[0] + s.crop_size[0] > s.film_width or s.crop_offset[1] + s.crop_size[1] > s.film_height):
            raise RuntimeError("Invalid crop window specification!")
        # hdrfilm.cpp:100-151: the format strings are validated even though only Film.bitmap() consumes the pixel format here
        ff = str(fp_.get("file_format", "openexr")).lower()
        if ff not in ("openexr", "exr", "rgbe", "pfm"):
            raise RuntimeError("The \"file_format\" parameter must either be equal to \"openexr\", \"pfm\", or \"rgbe\", found %s instead." % ff)
        pf = str(fp_.get("pixel_format", "rgba")).lower()
        if pf not in ("luminance", "luminance_alpha", "rgb", "rgba", "xyz", "xyza"):
            raise RuntimeError("The \"pixel_format\" parameter must either be equal to \"luminance\", \"luminance_alpha\", \"rgb\", "
                               "\"rgba\",  \"xyz\", \"xyza\". Found %s." % pf)
        cf = str(fp_.get("component_format", "float16")).lower()
        if cf not in ("float16", "float32", "uint32"):
            raise RuntimeError("The \"component_format\" parameter must either be equal to \"float16\", \"float32\", or \"uint32\". "
                               "Found %s instead." % cf)
        if ff in ("rgbe", "pfm") and not (ff == "pfm" and pf == "luminance"):
            pf = "rgb"                                                     # hdrfilm.cpp:153-176
        self.film_pixel_format = "luminance" if _MONO else pf             # hdrfilm.cpp:122-128
        fp_.get("high_quality_edges"); fp_.get("filename")
        rf = fp_.get("rfilter", {"type": "gaussian"})
        rp = Props(rf, where + ".film.rfilter")
        s.rfilter_radius, s.rfilter_stddev = 0.5, 0.5
        if rp.type == "box":
            s.rfilter_type = A.RFILTER_BOX
            s.rfilter_radius = float(rp.get("radius", 0.5))
        elif rp.type == "gaussian":
            s.rfilter_type = A.RFILTER_GAUSSIAN
            s.rfilter_stddev = float(rp.get("stddev", 0.5))
        else:
            raise RuntimeError("Unknown / unsupported reconstruction filter plugin \"%s\"" % rp.type)
        rp.finish()
        fp_.finish()
        # sampler (src/librender/sensor.cpp:44-49, src/librender/sampler.cpp:11-18)
        sd = p.get("sampler", {"type": "independent", "sample_count": 4})
        sp = Props(sd, where + ".sampler")
        if sp.type != "independent":
            raise RuntimeError("Unknown / unsupported sampler plugin \"%s\" (parity target is 'independent')" % sp.type)
        s.sample_count = int(sp.get("sample_count", 4))
        s.sampler_seed = int(sp.get("seed", 0))
        # Extension of this backend (the reference picks the seeding by variant): "wavefront": True gives the streams of the gpu_* variants --
        # one TEA-seeded PCG32 per (pixel, sample), librender/sampler.cpp:89-92 -- instead of scalar_rgb's one stream per pixel
        s.sampler_wavefront = int(bool(sp.get("wavefront", False)))
        sp.finish()
        shutter_open, shutter_close = float(p.get("shutter_open", 0.0)), float(p.get("shutter_close", 0.0))      # sensor.cpp:20-27
        if shutter_close < shutter_open:
            raise RuntimeError("Shutter opening time must be less than or equal to the shutter closing time!")
        s.shutter_open_time = shutter_close - shutter_open
        if p.has("medium"):
            s.medium = self.add_medium(p.get("medium"), where + ".medium")
        if p.type == "perspective":
            s.type = A.SENSOR_PERSPECTIVE
            tw = p.get("to_world")
            tw = ScalarTransform4f() if tw is None else tw
            m = tw.matrix[:3, :3].astype(np.float64)
            if np.any(np.abs(m @ m.T - np.eye(3)) > 1e-3):                 # transform.h:300-312, perspective.cpp:85-86
                raise RuntimeError("Scale factors in the camera-to-world transformation are not allowed!")
            s.to_world = _xf(tw)
            s.near_clip = float(p.get("near_clip", 1e-2))
            s.far_clip = float(p.get("far_clip", 1e4))
            p.get("focus_distance")
            self._srf(p, s, where)                                         # perspective.cpp:113-121
            if s.near_clip <= 0:
                raise RuntimeError("The 'near_clip' parameter must be greater than zero!")
            if s.far_clip <= s.near_clip:
                raise RuntimeError("The 'far_clip' parameter must be greater than 'near_clip'.")
            s.fov_x = parse_fov(p, s.film_width / float(s.film_height))
            s.principal_point_offset[:] = (float(p.get("principal_point_offset_x", 0.0)),
                                           float(p.get("principal_point_offset_y", 0.0)))
        elif p.type == "distant":
            s.type = A.SENSOR_DISTANT
            if p.has("direction"):                                         # distant.cpp:243-259
                if p.has("to_world"):
                    raise RuntimeError("Only one of the parameters 'direction' and 'to_world'can be specified at the same time!'")
                direction = np.asarray(p.get("direction"), dtype=np.float32)
                direction = normalize32(direction)
                if p.has("orientation"):
                    up = np.cross(direction, np.asarray(p.get("orientation"), dtype=np.float32)).astype(np.float32)
                    up = normalize32(up)
                else:
                    _, up = coordinate_system(direction)
                s.to_world = _xf(ScalarTransform4f.look_at([0, 0, 0], direction, up))
            else:
                s.to_world = _xf(p.get("to_world"))
            s.distant_flip_directions = int(bool(p.get("flip_directions", False)))
            s.distant_target_type = A.DISTANT_TARGET_NONE
            if p.has("ray_target"):
                rt = p.get("ray_target")
                if isinstance(rt, dict):
                    s.distant_target_type = A.DISTANT_TARGET_SHAPE
                    rec, _ = self.make_shape(rt, where + ".ray_target", in_scene=False)
                    s.distant_target_shape = rec
                else:
                    s.distant_target_type = A.DISTANT_TARGET_POINT
                    s.distant_target_point[:] = tuple(float(x) for x in np.asarray(rt, dtype=np.float32))
            if p.has("ray_origin"):                                         # distant.cpp:280-289
                ro = p.get("ray_origin")
                if not isinstance(ro, dict) or ro.get("type") not in SHAPE_PLUGINS:
                    raise RuntimeError("Invalid parameter ray_origin, must be a Shape.")
                s.distant_origin_type = 1
                rec, _ = self.make_shape(ro, where + ".ray_origin", in_scene=False)
                s.distant_origin_shape = rec
        elif p.type == "radiancemeter":
            # src/sensors/radiancemeter.cpp:60-98: one ray along +z of to_world (or of look_at(origin, origin + direction)); it is the
            # one-sub-sensor case of mradiancemeter (same ray arithmetic, :124-127 vs mradiancemeter.cpp:150-153)
            s.type = A.SENSOR_MRADIANCEMETER
            self._srf(p, s, where)                                         # radiancemeter.cpp:61-68
            if p.has("to_world"):
                p.get("direction"); p.get("origin")
                m = p.get("to_world").matrix
            else:
                if p.has("direction") != p.has("origin"):
                    raise RuntimeError("If the sensor is specified through origin and direction both values must be set!")
                if p.has("direction"):
                    origin = np.asarray(p.get("origin"), dtype=np.float32)
                    direction = np.asarray(p.get("direction"), dtype=np.float32)
                    up, _ = coordinate_system(direction)
                    m = ScalarTransform4f.look_at(origin, (origin + direction).astype(np.float32), up).matrix
                else:
                    m = ScalarTransform4f().matrix
            if (s.film_width, s.film_height) != (1, 1):
                raise RuntimeError("This sensor only supports films of size 1x1 Pixels!")
            buf = np.ascontiguousarray(np.asarray(m, dtype=np.float32).reshape(-1))
            self.keep.append(buf)
            s.multi_transforms = buf.ctypes.data_as(C.POINTER(C.c_float))
            s.multi_count = 1
            s.to_world = _xf(ScalarTransform4f())
            s.distant_target_type = A.DISTANT_TARGET_NONE
        elif p.type == "distantflux":                                      # src/sensors/distantflux.cpp:60-105,141-187
            s.type = A.SENSOR_DISTANTFLUX
            tw = p.get("to_world")
            s.to_world = _xf(ScalarTransform4f() if tw is None else tw)
            s.distant_target_type = A.DISTANT_TARGET_NONE
            if p.has("target"):
                tg = p.get("target")
                if isinstance(tg, dict):
                    s.distant_target_type = A.DISTANT_TARGET_SHAPE
                    rec, _ = self.make_shape(tg, where + ".target", in_scene=False)
                    s.distant_target_shape = rec
                else:
                    s.distant_target_type = A.DISTANT_TARGET_POINT
                    s.distant_target_point[:] = tuple(float(x) for x in np.asarray(tg, dtype=np.float32))
            if p.has("origin"):                                             # distantflux.cpp:172-184
                ro = p.get("origin")
                if not isinstance(ro, dict) or ro.get("type") not in SHAPE_PLUGINS:
                    raise RuntimeError("Invalid parameter origin, must be a Shape.")
                s.distant_origin_type = 1
                rec, _ = self.make_shape(ro, where + ".origin", in_scene=False)
                s.distant_origin_shape = rec
        elif p.type in ("mradiancemeter", "mdistant"):
            # src/sensors/mradiancemeter.cpp:72-132 / src/sensors/mdistant.cpp:60-98,147-203: N sub-sensors, one per film column
            multi = p.type == "mradiancemeter"
            s.type = A.SENSOR_MRADIANCEMETER if multi else A.SENSOR_MDISTANT
            if multi and p.has("to_world"):
                raise RuntimeError("This sensor is specified through a set of origin and direction values and cannot use the to_world transform.")

            def tokens(key):                                                   # string::tokenize(props.string(key), " ,")
                return [t for t in str(p.get(key)).replace(",", " ").split() if t]
            dirs = tokens("directions")
            if len(dirs) % 3 != 0:
                raise RuntimeError("Invalid specification! Number of parameters %d, is not a multiple of three." % len(dirs))
            count = len(dirs) // 3
            mats = np.zeros((count, 4, 4), dtype=np.float32)
            if multi:
                origs = tokens("origins")
                if len(origs) % 3 != 0:
                    raise RuntimeError("Invalid specification! Number of parameters %d, is not a multiple of three." % len(origs))
                if len(origs) != len(dirs):
                    raise RuntimeError("Invalid specification! Number of parameters for origins and directions (%d, %d) are not equal."
                                       % (len(origs), len(dirs)))
            for i in range(count):
                direction = np.array([np.float32(x) for x in dirs[3 * i:3 * i + 3]], dtype=np.float32)    # std::stof
                if multi:
                    origin = np.array([np.float32(x) for x in origs[3 * i:3 * i + 3]], dtype=np.float32)
                    up, _ = coordinate_system(direction)                                               # mradiancemeter.cpp:109: first vector
                    mats[i] = ScalarTransform4f.look_at(origin, (origin + direction).astype(np.float32), up).matrix
                else:
                    _, up = coordinate_system(direction)                                               # mdistant.cpp:166-167: second vector
                    mats[i] = ScalarTransform4f.look_at([0, 0, 0], direction, up).matrix
            if (s.film_width, s.film_height) != (count, 1):
                raise RuntimeError("Film size must be [sensor_count, 1]. Expected [%d, 1], got [%d, %d]" % (count, s.film_width, s.film_height))
            buf = np.ascontiguousarray(mats.reshape(-1), dtype=np.float32)
            self.keep.append(buf)
            s.multi_transforms = buf.ctypes.data_as(C.POINTER(C.c_float))
            s.multi_count = count
            s.to_world = _xf(ScalarTransform4f())
            s.distant_target_type = A.DISTANT_TARGET_NONE
            if not multi and p.has("target"):                                  # mdistant.cpp:71-88,188-198
                tg = p.get("target")
                if isinstance(tg, dict):
                    s.distant_target_type = A.DISTANT_TARGET_SHAPE
                    rec, _ = self.make_shape(tg, where + ".target", in_scene=False)
                    s.distant_target_shape = rec
                else:
                    s.distant_target_type = A.DISTANT_TARGET_POINT
                    s.distant_target_point[:] = tuple(float(x) for x in np.asarray(tg, dtype=np.float32))
        else:
            raise RuntimeError("Unknown / unsupported sensor plugin \"%s\"" % p.type)
        p.finish()
        self.sensor = s

    def _srf(self, p, s, where):
        """"srf" of perspective / radiancemeter: the spectral response function the wavelengths are sampled from.  Outside the spectral
        variants the plugins warn and ignore it (perspective.cpp:113-121, radiancemeter.cpp:62-68)."""
        if not p.has("srf"):
            return
        v = p.get("srf")
        if _SPECTRAL is None:
            return
        if isinstance(v, dict) and v.get("type") not in ("uniform", "discrete"):
            raise RuntimeError("srf: sample_spectrum is available for 'uniform' and 'discrete' spectra in this backend (%s)" % where)
        s.srf = 1 + _spectrum(v, where + ".srf")

    # ------------------------------------------------------------ integrator
    def set_integrator(self, d, where):
        p = Props(d, where)
        it = A.Integrator()
        self.aov_names = []
        if p.type in ("nbins", "bins"):
            # src/integrators/nbins.cpp:55-98, bins.cpp:23-85: wavelength-bin AOVs around one nested sampling integrator
            if _SPECTRAL is None:
                raise RuntimeError("This integrator can only be used with a spectral variant!")
            nested = [(k, v) for k, v in sorted_items(d) if isinstance(v, dict) and k not in ("type", "id")]
            for k, v in nested:
                if v.get("type") not in ("path", "volpath", "volpathmis", "nbins", "bins"):
                    raise RuntimeError("Child objects must be of type 'SamplingIntegrator'!")
            if len(nested) > 1:
                raise RuntimeError("More than one sub-integrator specified!")
            if not nested:
                raise RuntimeError("Must specify a sub-integrator!")
            if nested[0][1].get("type") in ("nbins", "bins"):
                raise RuntimeError("nested bin integrators are not supported by this backend")
            lo, hi, names = [], [], []
            if p.type == "nbins":
                spec = p.get("wavelengths")
                if spec is None:
                    raise RuntimeError("Property \"wavelengths\" has not been specified!")
                tol = float(p.get("tolerance", 1e-5))
                for tok in str(spec).replace(",", " ").split():
                    try:
                        lo.append(float(tok))
                    except ValueError:
                        raise RuntimeError("Could not parse floating point value '%s'" % tok)
                    hi.append(tol); names += [tok, tok + "_pop"]
                mode = 1
            else:
                spec = p.get("bins")
                if spec is None:
                    raise RuntimeError("Property \"bins\" has not been specified!")
                for tok in str(spec).replace(",", " ").split():
                    item = tok.split(":")
                    if len(item) != 3 or not all(item):
                        continue                                            # bins.cpp:47-51: warn and skip
                    try:
                        lo.append(float(item[1])); hi.append(float(item[2]))
                    except ValueError as e:
                        raise RuntimeError("Could not parse floating point value '%s'" % str(e).split("'")[-2])
                    names += [item[0], item[0] + "_weights"]
                mode = 2
            p.get(nested[0][0])
            # the wrapper is the SamplingIntegrator that renders (nbins.cpp / bins.cpp: Base(props)): ITS block_size, samples_per_pass and
            # timeout drive the render loop (integrator.cpp:29-48); of the nested integrator only sample() is used
            outer = (int(p.get("block_size", 0)), int(p.get("samples_per_pass", -1)), float(p.get("timeout", -1.0)))
            p.finish()
            self.set_integrator(nested[0][1], where + "." + nested[0][0])
            it = self.integrator
            it.block_size, it.samples_per_pass, it.timeout = outer
            if len(lo) > 64:
                raise RuntimeError("this backend supports at most 64 spectral bins")
            it.bin_mode, it.bin_count = mode, len(lo)
            self._bins = (np.ascontiguousarray(lo, np.float32), np.ascontiguousarray(hi, np.float32))
            self.keep += list(self._bins)
            it.bin_lo = self._bins[0].ctypes.data_as(A.fp); it.bin_hi = self._bins[1].ctypes.data_as(A.fp)
            self.aov_names = names
            return
        if p.type == "path":
            it.type = A.INTEGRATOR_PATH
        elif p.type == "volpath":
            it.type = A.INTEGRATOR_VOLPATH
        elif p.type == "volpathmis":
            it.type = A.INTEGRATOR_VOLPATHMIS
        else:
            raise RuntimeError("Unknown / unsupported integrator plugin \"%s\"" % p.type)
        it.use_spectral_mis = int(bool(p.get("use_spectral_mis", True))) if p.type == "volpathmis" else 1
        it.max_depth = int(p.get("max_depth", -1))
        it.rr_depth = int(p.get("rr_depth", 5))
        it.hide_emitters = int(bool(p.get("hide_emitters", False)))
        it.block_size = int(p.get("block_size", 0))
        it.samples_per_pass = int(p.get("samples_per_pass", -1))
        it.timeout = float(p.get("timeout", -1.0))
        if it.rr_depth <= 0:
            raise RuntimeError("\"rr_depth\" must be set to a value greater than zero!")
        if it.max_depth < 0 and it.max_depth != -1:
            raise RuntimeError("\"max_depth\" must be set to -1 (infinite) or a value >= 0")
        p.finish()
        self.integrator = it

    # ------------------------------------------------------------ scene
    def load(self, d):
        if not isinstance(d, dict) or d.get("type") != "scene":
            raise RuntimeError("load_dict(): the top-level dictionary must have type 'scene' in this backend")
        SHAPES = ("rectangle", "cube", "sphere", "mesh", "obj", "ply", "disk")
        for k, v in sorted_items(d):          # scene.cpp:23: props.objects() order
            if k in ("type", "id"):
                continue
            if not isinstance(v, dict):
                raise RuntimeError("Unexpected property \"%s\" at the scene level" % k)
            t = v.get("type")
            if t == "ref":
                raise RuntimeError("Reference found at the scene level: %s" % k)
            if t in SHAPES:
                self.add_shape(v, k)
            elif t in ("directional", "constant", "area", "point"):
                self.add_emitter(v, k)
            elif t in ("perspective", "distant", "mradiancemeter", "mdistant", "distantflux", "radiancemeter"):
                if self.sensor is not None:
                    raise RuntimeError("this backend supports a single sensor per scene")
                self.set_sensor(v, k)
            elif t in ("path", "volpath", "volpathmis", "nbins", "bins"):
                if self.integrator is not None:
                    raise RuntimeError("Only one integrator can be specified per scene.")
                self.set_integrator(v, k)
            elif t in ("diffuse", "null", "rpv", "bilambertian"):
                self.add_bsdf(v, k)
            elif t in ("homogeneous", "heterogeneous"):
                self.add_medium(v, k)
            elif t in ("isotropic", "hg", "rayleigh", "tabphase", "blendphase"):
                self.add_phase(v, k)
            elif t in ("gridvolume", "constvolume"):
                self.add_volume(v, k)
            else:
                raise RuntimeError("Unknown / unsupported plugin \"%s\" (key \"%s\")" % (t, k))
        if self.sensor is None:
            raise RuntimeError("No sensors found! (this backend does not instantiate a default camera)")
        if self.integrator is None:
            self.set_integrator({"type": "path"}, "integrator")          # scene.cpp:88-92
        return self.finish()

    def finish(self):
        desc = A.SceneDesc()
        desc.abi_version = A.MTS_ABI_VERSION

        def arr(cls, items):
            a = (cls * max(len(items), 1))(*items)
            self.keep.append(a)
            return a
        desc.volumes, desc.volume_count = arr(A.Volume, self.volumes), len(self.volumes)
        desc.phases, desc.phase_count = arr(A.Phase, self.phases), len(self.phases)
        desc.media, desc.medium_count = arr(A.Medium, self.media), len(self.media)
        desc.bsdfs, desc.bsdf_count = arr(A.Bsdf, self.bsdfs), len(self.bsdfs)
        desc.shapes, desc.shape_count = arr(A.Shape, self.shapes), len(self.shapes)
        desc.emitters, desc.emitter_count = arr(A.Emitter, self.emitters), len(self.emitters)
        desc.sensor = self.sensor
        desc.integrator = self.integrator
        desc.spectra, desc.spectrum_count = arr(A.Spectrum, self.spectra), len(self.spectra)
        self.keep.append(desc)
        return desc


def build_scene_desc(d, mono=False, spectral=False):
    """Returns (SceneDesc, keepalive). The keepalive object owns every buffer the description points to.
    mono: build the scene with the semantics of the *_mono variants; spectral: with those of scalar_spectral."""
    global _MONO, _SPECTRAL
    b = SceneBuilder()
    b.spectra = []
    with _BUILD_LOCK:                    # the variant of the scene being built is module state: one build at a time (threads may load scenes side by side)
        _MONO = bool(mono)
        _SPECTRAL = b if spectral else None
        try:
            desc = b.load(d)
        finally:
            _MONO = False
            _SPECTRAL = None
    desc.integrator.monochrome = int(bool(mono))
    desc.integrator.spectral = int(bool(spectral))
    return desc, b
