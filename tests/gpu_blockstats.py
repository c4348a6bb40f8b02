"""Diagnostic: per-block executions / lanes served of the flat volpath kernel (needs a -DMTSAMD_BLOCKSTATS build)."""
import ctypes as C, importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("eradiate-kernel_amd"); A = importlib.import_module("eradiate-kernel_amd._capi")
scenes = importlib.import_module("eradiate-kernel_amd.scenes")
w, h, spp = [int(x) for x in sys.argv[1:4]]
cfg = sys.argv[4] if len(sys.argv) > 4 else "C3"
pkg.set_variant("gpu_spectral" if cfg == "C5S" else "gpu_rgb")
scene = pkg.load_dict({"C3": scenes.c3_heterogeneous, "C4": scenes.c4_atmosphere, "C2": scenes.c2_homogeneous_slab, "C5S": scenes.c5_atmosphere_spectral}[cfg](w, h, spp)); sensor = scene.sensors()[0]
out = (C.c_ulonglong * 48)()
read = A.lib().mts_debug_blockstats_spectral if cfg == "C5S" else A.lib().mts_debug_blockstats
read(out, 1)
scene.integrator().render(scene, sensor, collect_counters=True)
st = scene.integrator().last_stats
read(out, 0)
names = ["INT", "MED", "SCATTER", "WSURF", "SURF", "PHASE", "NEW", "MEDW"]
waves = w * h / 64          # normalisation unit: one 64-pixel group (the launch may use fewer, fuller waves)
print("samples/wave-lane", spp, "kernel ms", st["kernel_ms"])
total = sum(out[24:36]) + out[42] + out[43] + out[44]
for i, n in enumerate(names):
    ex, lanes, cyc = out[i], out[12 + i], out[24 + i]
    print("%-7s executions/wave/sample %7.2f  lanes/execution %6.2f  lane-visits/sample %6.2f  cycles/execution %8.1f  share %5.1f %%" % (
        n, ex / waves / spp, lanes / max(ex, 1), lanes / (w * h * spp), cyc / max(ex, 1), 100.0 * cyc / max(total, 1)))
print("claim / vote %5.1f %%   idle %5.1f %%   push %5.1f %%   total cycles/wave/sample %9.0f" % (
    100.0 * out[44] / max(total, 1), 100.0 * out[42] / max(total, 1), 100.0 * out[43] / max(total, 1), total / waves / spp))
print("claim attempts %d, lost compare-and-swaps %d (%.1f %%)" % (out[10], out[11], 100.0 * out[11] / max(out[10], 1)))
seg = ["lds load", "rng+free-flight sample", "grid lookup", "transmittance/decide", "top", "lds store"]
print("MED segments (cycles / execution):", ", ".join("%s %.0f" % (n, out[36 + i] / max(out[1], 1)) for i, n in enumerate(seg)))
if out[47]:
    print("population at the claims' snapshots: finished paths %.1f, waiting in the rings %.1f of the workgroup's paths (%d snapshots)" % (out[45] / out[47], out[46] / out[47], out[47]))
