"""AddressSanitizer + UndefinedBehaviorSanitizer over the PRODUCT's host side ("ASAN for
host lib").

csrc/scene_host.cpp (validation and flattening of the caller's records: 790 lines of pointer chasing) and csrc/capi.cpp (the entry
points) are compiled once more with the product's own compiler in host-only mode -- `hipcc --cuda-host-only -DMTSAMD_HOST_ONLY
-fsanitize=address,undefined`: no device code, no upload, no launch; mts_scene_create runs up to the upload, mts_render up to its
first device call -- and a child interpreter with the sanitizer runtime preloaded drives it through the package's own binding:
  * every scene builder of the package (rgb, mono, spectral, nbins / bins, srf, multi-sensors, meshes above the BVH threshold,
    nested blendphase, wavefront streams, crop windows) through mts_scene_create and the validation half of mts_render;
  * ~60 corrupted records -- every index out of range, unknown enum values, null pointers with positive counts, negative counts,
    reversed intervals, invalid crop windows, shard specifications, film buffers that are too small -- each of which must come back
    as a status + message, never as a sanitizer report.
Any report (out-of-bounds read of a caller array, use after free, signed overflow, misaligned or null access) aborts the child.
CPU only: the GPU pool offers no device sanitizer."""
import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CSRC = os.path.join(ROOT, "eradiate-kernel_amd", "csrc")

CHILD = r"""
import ctypes as C, importlib, sys
import numpy as np
sys.path.insert(0, %(root)r)
A = importlib.import_module("eradiate-kernel_amd._capi")
SD = importlib.import_module("eradiate-kernel_amd.scene_dict")
scenes = importlib.import_module("eradiate-kernel_amd.scenes")
T = importlib.import_module("eradiate-kernel_amd.transform").ScalarTransform4f
L = A.lib()


def create(desc):
    h = C.c_void_p()
    rc = L.mts_scene_create(C.byref(desc), 0, C.byref(h))
    msg = (L.mts_last_error() or b"").decode()
    return rc, msg, h


def render_validation(h, desc, **opt):
    # the validation half of mts_render: options, passes, spiral, shard filter, film capacity; ends in "host-only build"
    opts = A.RenderOpts(); opts.shard_count = 1
    for k, v in opt.items():
        setattr(opts, k, v)
    n = desc.sensor.crop_size[0] * desc.sensor.crop_size[1] * (5 + 2 * desc.integrator.bin_count)
    film = np.zeros(max(n, 1), np.float32)
    if "film_capacity" not in opt:
        opts.film_capacity = film.size
    st = A.Stats()
    rc = L.mts_render(h, C.byref(opts), film.ctypes.data_as(C.c_void_p), C.byref(st))
    return rc, (L.mts_last_error() or b"").decode()


# ---- valid scenes
rng = np.random.default_rng(5)
pos = rng.uniform(-1, 1, (90, 3)).astype(np.float32); faces = rng.integers(0, 90, (60, 3)).astype(np.uint32)
mesh_scene = scenes.c1_cornell(20, 16, 2)
mesh_scene["blob"] = {"type": "mesh", "vertex_positions": pos, "faces": faces, "to_world": T.translate([0, 0, 2])}
mesh_scene["sensor"]["film"] = dict(mesh_scene["sensor"]["film"], crop_offset_x=3, crop_offset_y=2, crop_width=12, crop_height=9, rfilter={"type": "gaussian"})
wave = scenes.c1_cornell(16, 16, 8); wave["sensor"]["sampler"]["wavefront"] = True
nb = scenes.c5_atmosphere_spectral(16, 16, 4, layers=8, nodes=5)
nb["integrator"] = {"type": "nbins", "wavelengths": "400, 500, 600", "tolerance": 20.0, "integrator": nb["integrator"]}
passes = scenes.c3_heterogeneous(48, 40, 8, res=8, samples_per_pass=2)
VALID = [("c1", scenes.c1_cornell(16, 16, 4), {}), ("c2", scenes.c2_homogeneous_slab(16, 12, 4), {}), ("c3", scenes.c3_heterogeneous(24, 16, 4, res=8), {}),
         ("c3 passes", passes, {}), ("c4", scenes.c4_atmosphere(16, 16, 2, layers=8), {}), ("c4 one column", scenes.c4_atmosphere(16, 16, 2, layers=8, columns=1), {}),
         ("three species", scenes.c4_three_species(16, 16, 2, layers=8), {}), ("blend chain", scenes.c4_three_species(16, 16, 2, layers=8, chain=True), {}),
         ("c3 mono", scenes.c3_heterogeneous(16, 16, 4, res=8), {"mono": True}), ("c5 spectral", scenes.c5_atmosphere_spectral(16, 16, 2, layers=8, nodes=5), {"spectral": True}),
         ("nbins", nb, {"spectral": True}), ("mesh + crop + gaussian", mesh_scene, {}), ("wavefront", wave, {})]
for integ in ("path", "volpathmis"):
    d = scenes.c3_heterogeneous(16, 16, 4, res=8); d["integrator"]["type"] = integ
    VALID.append(("c3 " + integ, d, {}))
done = 0
kept = []
for name, d, kw in VALID:
    desc, keep = SD.build_scene_desc(d, **kw)
    rc, msg, h = create(desc)
    assert rc == 0, (name, msg)
    rc, msg = render_validation(h, desc)
    assert rc == 1 and "host-only build" in msg, (name, msg)
    # shard specifications and film capacity are checked before anything else
    for bad in ({"shard_count": 0}, {"shard_index": 2, "shard_count": 2}, {"shard_index": -1}):
        rc, msg = render_validation(h, desc, **bad)
        assert rc == 1 and "invalid shard" in msg, (name, bad, msg)
    rc, msg = render_validation(h, desc, film_capacity=4)
    assert rc == 1 and "film buffer holds" in msg, (name, msg)
    rc, msg = render_validation(h, desc, shard_index=1, shard_count=3)
    assert rc == 1 and "host-only build" in msg, (name, msg)
    L.mts_scene_destroy(h)
    kept.append((name, desc, keep))
    done += 1

# ---- corrupted records: (scene, path to the field, bad value, expected part of the message)
def scene(name):
    return next(d for n, d, _ in kept if n == name)


class Restore:
    def __init__(self, obj, field, value):
        self.obj, self.field, self.value = obj, field, value
    def __enter__(self):
        self.old = getattr(self.obj, self.field)
        self.ptr_type = None
        if isinstance(self.old, C.Array):
            self.old = list(self.old)
            for i, v in enumerate(self.value):
                getattr(self.obj, self.field)[i] = v
            return
        if isinstance(self.old, C._Pointer):         # a pointer read from a struct field aliases the field: keep the address instead
            self.ptr_type, self.old = type(self.old), C.cast(self.old, C.c_void_p).value
        setattr(self.obj, self.field, self.value)
    def __exit__(self, *a):
        if isinstance(self.old, list):
            for i, v in enumerate(self.old):
                getattr(self.obj, self.field)[i] = v
        elif self.ptr_type is not None:
            setattr(self.obj, self.field, C.cast(self.old, self.ptr_type))
        else:
            setattr(self.obj, self.field, self.old)


c1, c3, c4, blend, c5, nbd, mesh = scene("c1"), scene("c3"), scene("c4"), scene("three species"), scene("c5 spectral"), scene("nbins"), scene("mesh + crop + gaussian")
NULLF = C.POINTER(C.c_float)()
NULLU = C.POINTER(C.c_uint32)()
blend_i = next(i for i in range(blend.phase_count) if blend.phases[i].type == A.PHASE_BLEND)
tab_i = next(i for i in range(c4.phase_count) if c4.phases[i].type == A.PHASE_TABULATED)
mesh_i = next(i for i in range(mesh.shape_count) if mesh.shapes[i].type == A.SHAPE_MESH and mesh.shapes[i].vertex_count == 90)   # the blob
area_i = next(i for i in range(c1.emitter_count) if c1.emitters[i].type == A.EMITTER_AREA)
bs_i = next(i for i in range(c5.bsdf_count) if c5.bsdfs[i].type != A.BSDF_NULL)
CASES = [
    (c1, c1, "abi_version", 1, "ABI version"),
    (c1, c1.shapes[0], "bsdf", 99, "index out of range"), (c1, c1.shapes[0], "bsdf", -5, None),       # any negative index of an optional reference means "none"
    (c1, c1.shapes[0], "interior_medium", 7, "index out of range"), (c1, c1.shapes[0], "exterior_medium", -3, None),
    (c1, c1.shapes[0], "emitter", 40, "index out of range"), (c1, c1.shapes[0], "type", 17, "unknown shape"),
    (c1, c1.emitters[area_i], "shape", 1000, "index out of range"), (c1, c1.emitters[area_i], "shape", -1, "index out of range"),
    (c1, c1.emitters[0], "type", 9, "unknown emitter"), (c1, c1.bsdfs[0], "type", -1, "unknown BSDF"), (c1, c1.bsdfs[0], "type", 12, "unknown BSDF"),
    (c1, c1.sensor, "type", 11, "unknown sensor"), (c1, c1.sensor, "medium", 3, "index out of range"), (c1, c1.sensor, "sample_count", 0, "sample_count"),
    (c1, c1.sensor, "rfilter_type", 5, "unknown reconstruction filter"), (c1, c1.sensor, "rfilter_radius", 1e4, "radius too large"),
    (c1, c1.sensor, "crop_size", (900, 16), "crop"), (c1, c1.sensor, "crop_offset", (-1, 0), "crop"), (c1, c1.sensor, "film_width", 0, "crop"),
    (c1, c1.sensor, "srf", 1, None),        # read in the spectral variant only (c1, c1.integrator, "type", 3, "unknown integrator"), (c1, c1.integrator, "rr_depth", 0, "rr_depth"),
    (c1, c1.integrator, "max_depth", -2, "max_depth"), (c1, c1.integrator, "max_depth", 40000, "32767"), (c1, c1.integrator, "bin_mode", 1, "spectral variant"),
    (c1, c1.integrator, "spectral", 1, "spectr"), (c1, c1.sensor, "sampler_wavefront", 1, None),       # valid: one pass
    (mesh, mesh.shapes[mesh_i], "vertex_positions", NULLF, "missing vertex"), (mesh, mesh.shapes[mesh_i], "faces", NULLU, "missing vertex"),
    (mesh, mesh.shapes[mesh_i], "vertex_count", 0, "missing vertex"), (mesh, mesh.shapes[mesh_i], "vertex_count", 10, "face index out of range"),
    (mesh, mesh.shapes[mesh_i], "face_count", -4, "missing vertex"),
    (c3, c3.media[0], "sigma_t_volume", 9, "index out of range"), (c3, c3.media[0], "albedo_volume", -1, "index out of range"),
    (c3, c3.media[0], "phase", 5, "index out of range"), (c3, c3.media[0], "type", 2, "unknown medium"),
    (c3, c3.volumes[c3.media[0].sigma_t_volume], "data", NULLF, "missing data"), (c3, c3.volumes[c3.media[0].sigma_t_volume], "nx", 0, "Invalid grid dimensions"),
    (c3, c3.volumes[c3.media[0].sigma_t_volume], "nz", -8, "Invalid grid dimensions"), (c3, c3.volumes[c3.media[0].sigma_t_volume], "channels", 2, "channel count"),
    (c3, c3.volumes[c3.media[0].sigma_t_volume], "type", 7, "unknown volume"), (c3, c3.volumes[c3.media[0].sigma_t_volume], "type", A.VOLUME_CONST, "max() not implemented"),
    (c3, c3.volumes[c3.media[0].sigma_t_volume], "type", A.VOLUME_GRID_SPECTRAL, "spectral variant"),
    (c3, c3.phases[c3.media[0].phase], "g", 1.5, "asymmetry"), (c3, c3.phases[c3.media[0].phase], "type", 9, "unknown phase"),
    (c3, c3.sensor, "sampler_wavefront", 1, "samples_per_pass") if False else (c3, c3.integrator, "samples_per_pass", 3, None),
    (blend, blend.phases[blend_i], "child", (blend_i, 0), "precede"), (blend, blend.phases[blend_i], "child", (50, 0), "index out of range"),
    (blend, blend.phases[blend_i], "weight_volume", 77, "index out of range"), (blend, blend.phases[blend_i], "weight_volume", -1, "index out of range"),
    (c4, c4.phases[tab_i], "tab_values", NULLF, "at least two entries"), (c4, c4.phases[tab_i], "tab_count", 1, "at least two entries"),
    (c5, c5.spectra[0], "type", 8, "unknown spectrum"), (c5, c5.bsdfs[bs_i], "spectrum", (400, 400, 400, 400, 400, 400), "missing spectrum"),
    (c5, c5.emitters[0], "radiance_spectrum", -1, "missing spectrum"), (c5, c5.integrator, "monochrome", 1, "either monochromatic or spectral"),
    (nbd, nbd.integrator, "bin_mode", 3, "unknown bin mode"), (nbd, nbd.integrator, "bin_count", 65, "bins"), (nbd, nbd.integrator, "bin_count", -1, "bins"),
    (nbd, nbd.integrator, "bin_lo", NULLF, "bins"),
]
for desc, obj, field, value, expect in CASES:
    with Restore(obj, field, value):
        rc, msg, h = create(desc)
        if expect is None:
            assert rc == 0, (field, value, msg)
            rc2, msg2 = render_validation(h, desc)
            assert rc2 == 1, (field, value, msg2)          # "host-only build", or the message of a check in mts_render (samples_per_pass)
            L.mts_scene_destroy(h)
        else:
            assert rc == 1 and expect in msg, (field, value, expect, msg)
    done += 1
# a null description and a null output
assert L.mts_scene_create(None, 0, C.byref(C.c_void_p())) == 1 and "NULL" in L.mts_last_error().decode()
assert L.mts_scene_create(C.byref(c1), 0, None) == 1
assert L.mts_render(None, None, None, None) == 1 and L.mts_cancel(None) == 1 and L.mts_scene_destroy(None) == 0
# every record size the binding declares is the size the library compiled
for name, cls in A.ABI_STRUCTS.items():
    assert L.mts_abi_sizeof(name.encode()) == C.sizeof(cls), name
assert L.mts_abi_sizeof(b"nonsense") == -1
# the scenes are intact after all of it
for name, desc, keep in kept:
    rc, msg, h = create(desc)
    assert rc == 0, (name, msg)
    L.mts_scene_destroy(h)
print("host sanitizer child: %%d scenes, %%d corrupted records" %% (len(kept), len(CASES)))
"""


def _runtime():
    hits = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    return hits[-1] if hits else None


def build_host_only(out):
    cmd = [HIPCC, "-x", "hip", "--cuda-host-only", "-std=c++17", "-O1", "-g", "-fPIC", "-mfma", "-ffp-contract=off", "-fno-fast-math",
           "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-shared-libsan", "-DMTSAMD_HOST_ONLY",
           "-Wall", "-Wno-unused-function", "-Wno-unused-result", "-shared",
           os.path.join(CSRC, "scene_host.cpp"), os.path.join(CSRC, "capi.cpp"), "-o", out]
    subprocess.check_call(cmd)


def test_product_host_side_under_address_and_ub_sanitizers(tmp_path):
    asan = _runtime()
    if not asan or not os.path.exists(HIPCC):
        pytest.skip("hipcc or clang's sanitizer runtime not found")
    lib = str(tmp_path / "libmtsamd_host_asan.so")
    build_host_only(lib)
    env = dict(os.environ, LD_PRELOAD=asan, MTSAMD_LIB=lib,
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=900)
    report = r.stdout[-2000:] + r.stderr[-6000:]
    assert r.returncode == 0, report
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, report
    assert "host sanitizer child: 15 scenes" in r.stdout, report
