"""In-tree build of libmtsamd.so (HIP kernels + C ABI) for gfx950, and of the oracle (test infrastructure).

    python eradiate-kernel_amd/build.py [--force]

hipcc cross-compiles without a GPU.  Flags that matter for parity with the CPU restatement:
  -ffp-contract=off                          fused multiply-adds only where the source says pm_fma
  -fhip-fp32-correctly-rounded-divide-sqrt   IEEE fp32 division / sqrt on the device
  -fgpu-flush-denormals-to-zero              reference worker threads run with FTZ (integrator.cpp:117)
  -mfma                                      host-side constructors use hardware fma like the oracle
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.environ.get("MTSAMD_LIB_OUT") or os.path.join(HERE, "libmtsamd.so")   # MTSAMD_LIB_OUT: A/B builds next to the product
SOURCES = ["kernels.hip", "kernels_spectral.hip", "scene_host.cpp", "capi.cpp"]
HEADERS = ["pmath.h", "dmath.h", "dscene.h", "integrator_dev.h", "volpath_flat.h", "volpathmis_flat.h", "launch.h", "scene_host.h", "cie_tables.h"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
         "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fgpu-flush-denormals-to-zero",
         "-mfma", "-fno-fast-math",
         # machine LICM hoists the materialisation of constants out of the path loop and pays for it with registers: without it the
         # default kernel fits 128 VGPRs without a spill (four waves per SIMD: +9 % on the metric scene)
         "-mllvm", "-disable-machine-licm", "-Wall", "-Wno-unused-function", "-Wno-unused-result"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_backend(force=False, verbose=True, extra=()):
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(ROOT, "include", "mtsamd.h"), __file__]
    if not force and not _stale(LIB, deps):
        return LIB
    extra = list(extra) + os.environ.get("MTSAMD_EXTRA_FLAGS", "").split()
    flags = [("-fno-hip-fp32-correctly-rounded-divide-sqrt" if (os.environ.get("MTSAMD_EXP_FASTDIV") and f == "-fhip-fp32-correctly-rounded-divide-sqrt") else f)
             for f in FLAGS]                                 # MTSAMD_EXP_FASTDIV: measurement only, breaks parity
    # one hipcc per translation unit, side by side (the two kernel files take ~2.5 minutes each), then one link
    import tempfile
    cflags = [f for f in flags if f != "-shared"]
    with tempfile.TemporaryDirectory(prefix="mtsamd_build_") as tmp:
        jobs = []
        for f in SOURCES:
            obj = os.path.join(tmp, os.path.splitext(f)[0] + ".o")
            cmd = [HIPCC] + cflags + list(extra) + ["-x", "hip", "-c", os.path.join(CSRC, f), "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            jobs.append((cmd, obj, subprocess.Popen(cmd)))
        failed = [cmd for cmd, _, proc in jobs if proc.wait() != 0]
        if failed:
            raise subprocess.CalledProcessError(1, failed[0])
        link = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + [obj for _, obj, _ in jobs] + ["-o", LIB]
        if verbose:
            print(" ".join(link), flush=True)
        subprocess.check_call(link)
    return LIB


def build_oracle(force=False, verbose=True):
    odir = os.path.join(ROOT, "oracle")
    cmd = ["make", "-C", odir] + (["-B"] if force else [])
    subprocess.check_call(cmd, stdout=None if verbose else subprocess.DEVNULL)
    return os.path.join(odir, "liboracle.so")


if __name__ == "__main__":
    force = "--force" in sys.argv
    build_backend(force)
    build_oracle(force)
