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
        # Object cache (build/obj, git-ignored): a translation unit is recompiled only when ITS inputs changed -- its own source, every
        # header of csrc/ and include/ (any of them may be included), the flags and the compiler.  The build id goes into capi.cpp
        # alone (mts_build_id), so an edit of the host side costs seconds, not the minutes of the two kernel files.
        import hashlib
        cache = os.path.join(ROOT, "build", "obj")
        os.makedirs(cache, exist_ok=True)
        hh = hashlib.sha256()
        for name in sorted(HEADERS):
            with open(os.path.join(snap_csrc, name), "rb") as fh:
                hh.update(name.encode() + b"\0" + fh.read())
        with open(os.path.join(tmp, "include", "mtsamd.h"), "rb") as fh:
            hh.update(b"mtsamd.h\0" + fh.read())
        hh.update(("\0".join(cflags + list(extra)) + "\0" + toolchain).encode())
        jobs = []
        for f in SOURCES:
            ids = ['-DMTSAMD_BUILD_ID="%s"' % build_id, '-DMTSAMD_TOOLCHAIN_ID="%s"' % toolchain] if f == "capi.cpp" else []
            h = hh.copy()
            with open(os.path.join(snap_csrc, f), "rb") as fh:
                text = fh.read()
            h.update(f.encode() + b"\0" + text)
            for other in SOURCES:                               # a translation unit that includes another one (kernels_spectral.hip -> kernels.hip)
                if other != f and ('#include "%s"' % other).encode() in text:
                    with open(os.path.join(snap_csrc, other), "rb") as fh:
                        h.update(other.encode() + b"\0" + fh.read())
            h.update(" ".join(ids).encode())
            obj = os.path.join(cache, "%s.%s.o" % (os.path.splitext(f)[0], h.hexdigest()[:20]))
            if os.path.exists(obj) and not force:
                jobs.append((None, obj, None))
                continue
            cmd = [HIPCC] + cflags + list(extra) + ids + ["-x", "hip", "-c", os.path.join(snap_csrc, f), "-o", obj + ".tmp"]
            if verbose:
                print(" ".join(cmd), flush=True)
            jobs.append((cmd, obj, subprocess.Popen(cmd)))
        failed = [cmd for cmd, _, proc in jobs if proc is not None and proc.wait() != 0]
        if failed:
            raise subprocess.CalledProcessError(1, failed[0])
        for cmd, obj, proc in jobs:
            if proc is not None:
                os.replace(obj + ".tmp", obj)
        # keep the cache small: the newest few objects per translation unit
        for f in SOURCES:
            stem = os.path.splitext(f)[0] + "."
            old = sorted((os.path.join(cache, n) for n in os.listdir(cache) if n.startswith(stem) and n.endswith(".o")), key=os.path.getmtime)
            for path in old[:-4]:
                os.remove(path)
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
