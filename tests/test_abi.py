"""The C-ABI library loads, exports every symbol include/mtsamd.h declares, and the ctypes mirror has the
compiled struct sizes.  No compute calls (works without a GPU)."""
import ctypes as C
import importlib
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
A = importlib.import_module("eradiate-kernel_amd._capi")


@pytest.fixture(scope="module")
def L():
    importlib.import_module("eradiate-kernel_amd.build").build_backend(verbose=False)
    return A.lib()


def declared_functions():
    src = open(os.path.join(ROOT, "include", "mtsamd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mts_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_are_exported(L):
    names = declared_functions()
    assert {"mts_scene_create", "mts_scene_destroy", "mts_render", "mts_cancel", "mts_sample", "mts_ray_intersect",
            "mts_last_error", "mts_abi_version", "mts_device_count", "mts_abi_sizeof"} <= set(names)
    for n in names:
        assert hasattr(L, n), "libmtsamd.so does not export %s" % n
    assert set(A.ABI_SYMBOLS) == set(names)


def test_abi_version_and_struct_sizes(L):
    assert L.mts_abi_version() == A.MTS_ABI_VERSION
    for name, cls in A.ABI_STRUCTS.items():
        assert L.mts_abi_sizeof(name.encode()) == C.sizeof(cls), name
    assert L.mts_abi_sizeof(b"nonsense") == -1


def test_errors_do_not_cross_the_boundary(L):
    h = C.c_void_p()
    desc = A.SceneDesc()            # abi_version = 0
    assert L.mts_scene_create(C.byref(desc), 0, C.byref(h)) != 0
    assert b"ABI version" in L.mts_last_error()
    assert L.mts_cancel(None) != 0


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(A, "_lib", None)
    monkeypatch.setattr(A, "LIB_PATH", "/nonexistent/libmtsamd.so")
    with pytest.raises(A.BackendError):
        A.lib()


def test_default_kernel_resource_budget(tmp_path):
    """Guards the default render kernel against silent code-generation regressions (a reference to the kernel-argument record
    handed to a real function once put the whole record in scratch memory and cost 2.4x): reads the code object's own metadata."""
    import re
    import shutil
    import subprocess
    llvm = "/opt/rocm/lib/llvm/bin"
    tools = [os.path.join(llvm, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-readelf")]
    if not all(os.path.exists(t) for t in tools):
        pytest.skip("ROCm LLVM tools not found")
    lib = os.environ.get("MTSAMD_LIB") or os.path.join(ROOT, "eradiate-kernel_amd", "libmtsamd.so")
    fat = str(tmp_path / "fat.bin")
    subprocess.run([tools[0], "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
    # the section holds one offload bundle per translation unit with device code (kernels.hip: rgb / mono, kernels_spectral.hip: spectral,
    # kernels_lean_a.hip / _b.hip / _s.hip: the regrouping kernels without what a scene of their traits cannot contain)
    blob, magic = open(fat, "rb").read(), b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    assert len(starts) == 9, starts
    kernels = {}
    for k, o in enumerate(starts):
        part, co = str(tmp_path / ("b%d.bin" % k)), str(tmp_path / ("b%d.co" % k))
        open(part, "wb").write(blob[o:starts[k + 1] if k + 1 < len(starts) else len(blob)])
        subprocess.run([tools[1], "--unbundle", "--type=o", "--input=" + part, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        notes = subprocess.run([tools[2], "--notes", co], check=True, capture_output=True, text=True).stdout
        for block in notes.split("- .agpr_count")[1:]:
            name = re.search(r"\.name:\s+(\S+)", block).group(1)
            kernels[name] = {f: int(re.search(r"\.%s:\s+(\d+)" % f, block).group(1))
                             for f in ("private_segment_fixed_size", "group_segment_fixed_size", "vgpr_count", "vgpr_spill_count", "sgpr_spill_count")}

    def one(pattern):
        found = [v for k, v in kernels.items() if pattern in k]
        assert len(found) == 1, (pattern, sorted(kernels))
        return found[0]
    d = one("5v_rgb17render_kernel_wgaILb0ELi1024ELi1024ELi4ELb0E")
    assert d["vgpr_count"] <= 128                       # 4 waves per SIMD (launch bounds 1024 threads x 4)
    # scratch: the frames of the real (out-of-line) functions -- sphere / BVH intersection, generic volume lookups and, since round 3, the
    # walk of nested blendphase trees (four 8-entry stacks); none of it is touched by the atmosphere scenes
    # spills (round 4: 12): the ray origin of the distant sensor's sample_ray, kept in scratch across its `ray_origin` branch at the
    # four sites begin_sample is inlined at -- once per camera sample, not per tracking step (disassembly: the stores follow the
    # bounding-sphere arithmetic).  More than 16 means a spill has reached a block that runs per step.
    assert d["private_segment_fixed_size"] <= 384 and d["vgpr_spill_count"] <= 16, d
    assert d["sgpr_spill_count"] <= 400, d
    assert d["group_segment_fixed_size"] <= 160 * 1024, d
    # the same machine for the wavefront (gpu_*) streams, an instantiation of its own (wg_block, WF): the same budget
    w = one("5v_rgb17render_kernel_wgaILb0ELi1024ELi1024ELi4ELb1E")
    assert w["vgpr_count"] <= 128 and w["vgpr_spill_count"] <= 24 and w["group_segment_fixed_size"] <= 160 * 1024, w
    # volpathmis on the rings (round 4: two matrix slots, the path's pair parked during walks): 512 paths x 50 state dwords, two waves per SIMD
    m = one("5v_rgb21render_kernel_wga_misILb0ELb1ELi512ELi512E")
    assert m["vgpr_count"] <= 200 and m["vgpr_spill_count"] == 0 and m["group_segment_fixed_size"] <= 112 * 1024, m
    # ... and four wide: 256 paths x 69 dwords, TWO workgroups per CU at two waves per SIMD
    ms = one("10v_spectral21render_kernel_wga_misILb0ELb1ELi256ELi256E")
    assert ms["vgpr_count"] <= 256 and ms["vgpr_spill_count"] == 0 and 2 * ms["group_segment_fixed_size"] <= 160 * 1024, ms
    # `path` as a flat loop with regeneration: 128 VGPRs (4 waves per SIMD) with spills; 5 waves measured 40 % slower
    pk = one("5v_rgb13render_kernelILb0ELb1ELi0EE")
    assert pk["vgpr_count"] <= 128 and pk["vgpr_spill_count"] <= 100, pk
    # ... and over four-wide spectra (round 4): three waves per SIMD (168 VGPRs), fewer spills than the nested loop it replaces (51)
    pks = one("10v_spectral13render_kernelILb0ELb1ELi0EE")
    assert pks["vgpr_count"] <= 168 and pks["vgpr_spill_count"] <= 48, pks
    # the lean units (integrator_dev.h: MTS_TRAITS).  a: every promise kept (the metric scene) -- a kernel without a call: no spilled VGPR,
    # no scratch frame to speak of; b: rpv and blend-weight grids allowed (the layered atmosphere)
    la = one("12v_rgb_lean_a17render_kernel_wgaILb0ELi1024ELi1024ELi4ELb0E")
    assert la["vgpr_count"] <= 128 and la["vgpr_spill_count"] == 0 and la["private_segment_fixed_size"] <= 64 and la["sgpr_spill_count"] <= 160, la
    lb = one("12v_rgb_lean_b17render_kernel_wgaILb0ELi1024ELi1024ELi4ELb0E")
    assert lb["vgpr_count"] <= 128 and lb["vgpr_spill_count"] <= 16 and lb["private_segment_fixed_size"] <= 256 and lb["sgpr_spill_count"] <= 160, lb
    lm = one("12v_rgb_lean_a21render_kernel_wga_misILb0ELb1ELi512ELi768E")            # 512 paths, 768 threads: three waves per SIMD
    assert lm["vgpr_count"] <= 168 and lm["vgpr_spill_count"] == 0, lm
    # `path` without the callees a scene without BVH / spheres / rpv cannot reach (kernels_lean_p.hip, _ps.hip)
    lp = one("12v_rgb_lean_p13render_kernelILb0ELb1ELi0EE")
    assert lp["vgpr_count"] <= 128 and lp["vgpr_spill_count"] <= 48, lp
    lps = one("17v_spectral_lean_p13render_kernelILb0ELb1ELi0EE")
    assert lps["vgpr_count"] <= 168 and lps["vgpr_spill_count"] <= 8, lps
    # the spectral variant's volpath: 256 paths x 42 state dwords, three workgroups per CU
    sp = one("10v_spectral17render_kernel_wgaILb0ELi256ELi256ELi2ELb0E")
    assert sp["vgpr_count"] <= 168 and sp["vgpr_spill_count"] == 0 and 3 * sp["group_segment_fixed_size"] <= 160 * 1024, sp


def test_scene_traits(L):
    """Which lean translation unit mts_render may launch is decided per scene on the host (scene_host.cpp: scene_traits; the promises are
    integrator_dev.h's MT_* bits).  A wrong promise would be a wrong image, so the decision is pinned here, without a GPU, through a
    host-only debug export: the BASELINE scenes qualify for the units they are benchmarked on, and every feature a unit was compiled
    without -- a BVH, a sphere (also as a distant sensor's origin shape), an area emitter, a nested blendphase, rpv, a grid behind
    volume_eval(), a medium of the other kind -- clears its bit."""
    SD = importlib.import_module("eradiate-kernel_amd.scene_dict")
    scenes = importlib.import_module("eradiate-kernel_amd.scenes")
    T = importlib.import_module("eradiate-kernel_amd.transform").ScalarTransform4f
    L.mts_debug_scene_traits.argtypes = [C.POINTER(A.SceneDesc), C.POINTER(C.c_int32)]
    MEDIA, NO_BVH, NO_SPHERE, NO_GRID_EVAL, NO_SHAPE_EMITTER, NO_PHASE_TREE, NO_RPV, HOMOG = 1, 2, 4, 8, 16, 32, 64, 128
    A_UNIT, B_UNIT, H_UNIT, P_UNIT = 127, 1 | 2 | 4 | 16 | 32, 190, 70

    def traits(d, spectral=False):
        desc, keep = SD.build_scene_desc(d, spectral=spectral)
        t = C.c_int32(-1)
        assert L.mts_debug_scene_traits(C.byref(desc), C.byref(t)) == 0, L.mts_last_error()
        return t.value

    grey = {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.5, 0.5, 0.5]}}
    c3 = scenes.c3_heterogeneous(64, 64, 4, res=8)
    assert traits(c3) & A_UNIT == A_UNIT and not traits(c3) & HOMOG                               # the metric scene: unit a
    c4 = scenes.c4_atmosphere(32, 32, 4, layers=8)
    t4 = traits(c4)
    assert t4 & B_UNIT == B_UNIT and not t4 & NO_RPV and t4 & A_UNIT != A_UNIT                     # rpv ground: unit b, not a
    c2 = scenes.c2_homogeneous_slab(32, 32, 4)
    assert traits(c2) & H_UNIT == H_UNIT and not traits(c2) & MEDIA                                # homogeneous slab: unit h
    c1 = scenes.c1_cornell(32, 32, 4)
    assert traits(c1) & P_UNIT == P_UNIT and not traits(c1) & NO_SHAPE_EMITTER                     # cornell box: unit p (its light is an area emitter)
    with_sphere = dict(c3, ball={"type": "sphere", "center": [0.0, 0.0, 30.0], "radius": 0.5, "bsdf": grey})
    assert not traits(with_sphere) & NO_SPHERE and traits(with_sphere) & MEDIA
    many = dict(c3)
    for k in range(45):
        many["leaf%02d" % k] = {"type": "rectangle", "to_world": T.translate([0.1 * k, 0.0, 20.0]) @ T.scale(0.2), "bsdf": grey}
    assert not traits(many) & NO_BVH and traits(many) & 53 == 53                                   # 47 primitives: a BVH is built -- unit c
    mixed = dict(c3, haze={"type": "cube", "to_world": T.translate([0.0, 0.0, 40.0]), "bsdf": {"type": "null"},
                           "interior": {"type": "homogeneous", "sigma_t": 0.05, "albedo": 0.9}})
    assert not traits(mixed) & MEDIA and not traits(mixed) & HOMOG and not traits(mixed) & NO_GRID_EVAL   # its grids are then read through volume_eval()
    distant_sphere = scenes.c2_homogeneous_slab(8, 6, 4)
    distant_sphere["sensor"] = {"type": "distant", "film": dict(distant_sphere["sensor"]["film"]), "sampler": distant_sphere["sensor"]["sampler"],
                                "ray_origin": {"type": "sphere", "center": [0, 0, 1], "radius": 30.0},
                                "ray_target": {"type": "rectangle", "to_world": T.translate([0, 0, 2.0]) @ T.scale(2.0)}}
    assert not traits(distant_sphere) & NO_SPHERE                                                  # the sensor's own origin shape counts
    assert traits(scenes.c5_atmosphere_spectral(16, 16, 4, layers=8), spectral=True) & B_UNIT == B_UNIT   # spectral variant: unit s


def test_binary_identifies_its_sources(L, tmp_path, monkeypatch):
    """mts_build_id(): the library carries a hash of the sources, headers and flags it was built from; the build script rebuilds when
    the tree's hash differs (no mtimes), and the binding refuses a library that is not the tree's -- a header touched without a
    rebuild makes set_variant fail loudly instead of rendering with the old kernels."""
    import shutil
    B = importlib.import_module("eradiate-kernel_amd._buildid")
    lib_path = os.path.join(ROOT, "eradiate-kernel_amd", "libmtsamd.so")
    L.mts_build_id.restype = C.c_char_p
    built = L.mts_build_id().decode()
    assert re.fullmatch(r"[0-9a-f]{16}", built)
    assert built == B.tree_build_id(B.effective_flags()) == B.binary_build_id(lib_path)            # from the symbol, from the tree, from the file's bytes
    # the same sources with one byte appended to a header: another id
    csrc = tmp_path / "csrc"
    shutil.copytree(B.CSRC, csrc)
    with open(csrc / "dmath.h", "a") as f:
        f.write("\n")
    changed = B.tree_build_id(csrc=str(csrc))
    assert changed != built and B.tree_build_id(flags=B.FLAGS + ["-DX"]) != built
    # ... which the binding notices when it loads the (now stale) library
    monkeypatch.setattr(A, "_lib", None)
    monkeypatch.setattr(B, "tree_build_id", lambda *a, **k: changed)
    monkeypatch.delenv("MTSAMD_LIB", raising=False)
    with pytest.raises(A.BackendError, match="built from other sources"):
        A.lib()
    pkg = importlib.import_module("eradiate-kernel_amd")
    with pytest.raises(A.BackendError, match="built from other sources"):
        pkg.set_variant("gpu_rgb")
    assert B.binary_build_id(str(tmp_path / "missing.so")) is None


def test_build_id_follows_the_environment_of_the_build(monkeypatch, tmp_path):
    """build.py hashed FLAGS + MTSAMD_EXTRA_FLAGS (and swapped a flag under MTSAMD_EXP_FASTDIV) while the loader compared
    with the id of the default flags -- a library built with extra flags at the default path was refused.  Both now take the flag list
    from _buildid.effective_flags(); a package deployed without csrc/ skips the comparison instead of raising FileNotFoundError; the
    compiler's version is embedded next to the id (build.py rebuilds when it changes) without being part of it."""
    B = importlib.import_module("eradiate-kernel_amd._buildid")
    base = B.tree_build_id(B.effective_flags({}))
    assert base == B.tree_build_id()                                            # no environment: the default flags
    extra = B.effective_flags({"MTSAMD_EXTRA_FLAGS": "-DEXP_X=1 -DEXP_Y"})
    assert extra[-2:] == ["-DEXP_X=1", "-DEXP_Y"] and B.tree_build_id(extra) != base
    fast = B.effective_flags({"MTSAMD_EXP_FASTDIV": "1"})
    assert "-fno-hip-fp32-correctly-rounded-divide-sqrt" in fast and "-fhip-fp32-correctly-rounded-divide-sqrt" not in fast
    # the loader under the environment of the build accepts what build.py would have produced under it
    lib_path = os.path.join(ROOT, "eradiate-kernel_amd", "libmtsamd.so")
    monkeypatch.setattr(A, "_lib", None)
    monkeypatch.setenv("MTSAMD_EXTRA_FLAGS", "-DEXP_X=1")
    monkeypatch.delenv("MTSAMD_LIB", raising=False)
    with pytest.raises(A.BackendError, match="built from other sources or flags"):
        A.lib()                                                                 # this library was built without the flag
    monkeypatch.delenv("MTSAMD_EXTRA_FLAGS")
    monkeypatch.setattr(A, "_lib", None)
    assert A.lib() is not None
    # toolchain id: 8 hex digits in the file, equal to the hash of this box's `hipcc --version`
    tc = B.binary_toolchain_id(lib_path)
    assert re.fullmatch(r"[0-9a-f]{8}", tc or "")
    if os.path.exists("/opt/rocm/bin/hipcc"):
        assert tc == B.toolchain_id("/opt/rocm/bin/hipcc")
    # a deployed package (no csrc/): nothing to compare with, the library is taken as it is
    monkeypatch.setattr(A, "_lib", None)
    monkeypatch.setattr(B, "CSRC", str(tmp_path / "no_such_dir"))
    assert A.lib() is not None
    monkeypatch.setattr(A, "_lib", None)
