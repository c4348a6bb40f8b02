"""Identity of a libmtsamd.so build: a hash of everything that goes into it (sources, headers, flags).

build.py embeds the id in the library (`mts_build_id()`, and as the byte string MTSAMD_BUILD_ID=<16 hex digits> so that it can be read
without loading the library); build_backend() rebuilds when the id of the tree differs from the id of the binary, and _capi.lib()
refuses a binary whose id is not the tree's -- a stale or foreign libmtsamd.so fails loudly instead of rendering with old kernels."""
import hashlib
import os

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["kernels.hip", "kernels_spectral.hip", "kernels_lean_a.hip", "kernels_lean_b.hip", "kernels_lean_c.hip", "kernels_lean_s.hip", "kernels_lean_h.hip", "kernels_lean_p.hip", "kernels_lean_ps.hip", "scene_host.cpp", "capi.cpp"]
HEADERS = ["pmath.h", "dmath.h", "dscene.h", "integrator_dev.h", "volpath_flat.h", "volpathmis_flat.h", "launch.h", "scene_host.h", "cie_tables.h"]
MARKER = b"MTSAMD_BUILD_ID="
# hipcc flags of the product build (build.py).  Flags that matter for parity with the CPU restatement: see build.py's docstring.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
         "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fgpu-flush-denormals-to-zero",
         "-mfma", "-fno-fast-math",
         # machine LICM hoists the materialisation of constants out of the path loop and pays for it with registers: without it the
         # default kernel fits 128 VGPRs without a spill (four waves per SIMD: +9 % on the metric scene)
         "-mllvm", "-disable-machine-licm", "-Wall", "-Wno-unused-function", "-Wno-unused-result"]


def effective_flags(environ=None):
    """The flag list of the build as the environment shapes it -- the ONE place build.py and the loader (_capi.lib) take it from, so a
    library built with MTSAMD_EXTRA_FLAGS / MTSAMD_EXP_FASTDIV at the default path is recognised by the loader under the same
    environment (and refused under another)."""
    env = os.environ if environ is None else environ
    flags = [("-fno-hip-fp32-correctly-rounded-divide-sqrt" if (env.get("MTSAMD_EXP_FASTDIV") and f == "-fhip-fp32-correctly-rounded-divide-sqrt") else f)
             for f in FLAGS]                                 # MTSAMD_EXP_FASTDIV: measurement only, breaks parity
    return flags + env.get("MTSAMD_EXTRA_FLAGS", "").split()


TOOLCHAIN_MARKER = b"MTSAMD_TOOLCHAIN="


def toolchain_id(hipcc):
    """8 hex digits of `hipcc --version`: embedded next to the build id, so that build.py rebuilds after a compiler update.  Not part
    of the build id itself -- the loader must be able to verify a library on a box without the compiler."""
    import subprocess
    try:
        out = subprocess.run([hipcc, "--version"], capture_output=True, timeout=60).stdout
    except (OSError, subprocess.SubprocessError):
        out = b"unknown"
    return hashlib.sha256(out).hexdigest()[:8]


def binary_toolchain_id(lib_path):
    try:
        with open(lib_path, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    k = blob.find(TOOLCHAIN_MARKER)
    return blob[k + len(TOOLCHAIN_MARKER):k + len(TOOLCHAIN_MARKER) + 8].decode("ascii", "replace") if k >= 0 else None


def tree_build_id(flags=FLAGS, csrc=CSRC, include=None):
    """Hash of the sources as they are in the tree now, plus the compiler flags."""
    h = hashlib.sha256()
    for name in SOURCES + HEADERS:
        h.update(name.encode() + b"\0")
        with open(os.path.join(csrc, name), "rb") as f:
            h.update(f.read())
    with open(os.path.join(include or os.path.join(ROOT, "include"), "mtsamd.h"), "rb") as f:
        h.update(b"mtsamd.h\0" + f.read())
    h.update("\0".join(flags).encode())
    return h.hexdigest()[:16]


def binary_build_id(lib_path):
    """The id embedded in a built library, or None (no library / built before ids existed)."""
    try:
        with open(lib_path, "rb") as f:
            blob = f.read()
    except OSError:
        return None
    k = blob.find(MARKER)
    return blob[k + len(MARKER):k + len(MARKER) + 16].decode("ascii", "replace") if k >= 0 else None
