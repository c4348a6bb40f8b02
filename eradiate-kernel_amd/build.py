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
sys.path.insert(0, HERE)
import _buildid                                          # noqa: E402  (plain module: build.py also runs as a script)
SOURCES, HEADERS = _buildid.SOURCES, _buildid.HEADERS
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = _buildid.FLAGS


def build_backend(force=False, verbose=True, extra=()):
    flags = _buildid.effective_flags()                       # FLAGS as MTSAMD_EXP_FASTDIV / MTSAMD_EXTRA_FLAGS shape them: the loader computes the same
    extra = list(extra)
    # The binary says which sources it was built from (mts_build_id): rebuild iff that differs from the tree -- not by mtimes, which
    # a checkout, a copy to another box or a reverted experiment all falsify -- or the compiler has changed.
    build_id = _buildid.tree_build_id(flags + extra)
    toolchain = _buildid.toolchain_id(HIPCC)
    if not force and _buildid.binary_build_id(LIB) == build_id and _buildid.binary_toolchain_id(LIB) == toolchain:
        return LIB
    # one hipcc per translation unit, side by side (the two kernel files take ~2.5 minutes each), then one link
    import shutil
    import tempfile
    cflags = [f for f in flags if f != "-shared"]
    with tempfile.TemporaryDirectory(prefix="mtsamd_build_") as tmp:
        # Compile a SNAPSHOT of the sources: hipcc reads a translation unit twice (host and device pass, minutes apart), so a file
        # edited while the build runs would give a binary that matches neither tree -- and whose build id claims the old one.  The
        # snapshot keeps the tree's relative layout (csrc includes "../../include/mtsamd.h") and is checked against the id.
        snap_csrc = os.path.join(tmp, "pkg", "csrc")
        os.makedirs(os.path.join(tmp, "include"))
        shutil.copytree(CSRC, snap_csrc)
        shutil.copy(os.path.join(ROOT, "include", "mtsamd.h"), os.path.join(tmp, "include", "mtsamd.h"))
        if _buildid.tree_build_id(flags + extra, csrc=snap_csrc, include=os.path.join(tmp, "include")) != build_id:
            raise RuntimeError("the sources changed while they were being copied; run the build again")
        jobs = []
        for f in SOURCES:
            obj = os.path.join(tmp, os.path.splitext(f)[0] + ".o")
            cmd = [HIPCC] + cflags + list(extra) + ['-DMTSAMD_BUILD_ID="%s"' % build_id, '-DMTSAMD_TOOLCHAIN_ID="%s"' % toolchain,
                                                    "-x", "hip", "-c", os.path.join(snap_csrc, f), "-o", obj]
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
