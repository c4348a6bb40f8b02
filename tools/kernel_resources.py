"""Prints VGPRs / spills / scratch / LDS of every kernel in libmtsamd.so (read from the code objects' metadata).

    python tools/kernel_resources.py [pattern] [--lib path]
"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_resources(lib=None):
    lib = lib or os.environ.get("MTSAMD_LIB") or os.path.join(ROOT, "eradiate-kernel_amd", "libmtsamd.so")
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fat.bin")
        subprocess.run([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat], check=True)
        blob, magic = open(fat, "rb").read(), b"__CLANG_OFFLOAD_BUNDLE__"
        starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
        for k, o in enumerate(starts):
            part, co = os.path.join(tmp, "b%d.bin" % k), os.path.join(tmp, "b%d.co" % k)
            open(part, "wb").write(blob[o:starts[k + 1] if k + 1 < len(starts) else len(blob)])
            subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + part,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
            for block in notes.split("- .agpr_count")[1:]:
                name = re.search(r"\.name:\s+(\S+)", block).group(1)
                out[name] = {f: int(re.search(r"\.%s:\s+(\d+)" % f, block).group(1))
                             for f in ("private_segment_fixed_size", "group_segment_fixed_size", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count")}
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = None
    if "--lib" in sys.argv:
        lib = sys.argv[sys.argv.index("--lib") + 1]; args = [a for a in args if a != lib]
    pat = args[0] if args else ""
    for name, d in sorted(kernel_resources(lib).items()):
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("mtsamd::", "")
        if pat in name or pat in dem:
            print("%-90s vgpr %3d sgpr %3d vspill %3d sspill %3d scratch %4d lds %6d" % (dem[:90], d["vgpr_count"], d["sgpr_count"], d["vgpr_spill_count"],
                  d["sgpr_spill_count"], d["private_segment_fixed_size"], d["group_segment_fixed_size"]))
