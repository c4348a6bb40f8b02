"""Diagnostic: per-block executions / lanes served of the flat volpath kernel (needs a -DMTSAMD_BLOCKSTATS build)."""
import ctypes as C, importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("eradiate-kernel_amd"); A = importlib.import_module("eradiate-kernel_amd._capi")
scenes = importlib.import_module("eradiate-kernel_amd.scenes")
pkg.set_variant("gpu_rgb")
w, h, spp = [int(x) for x in sys.argv[1:4]]
scene = pkg.load_dict(scenes.c3_heterogeneous(w, h, spp)); sensor = scene.sensors()[0]
out = (C.c_ulonglong * 48)()
A.lib().mts_debug_blockstats(out, 1)
scene.integrator().render(scene, sensor, collect_counters=True)
st = scene.integrator().last_stats
A.lib().mts_debug_blockstats(out, 0)
names = ["INT", "MED", "SCATTER", "WSURF", "SURF", "PHASE", "NEW", "MEDW"]
waves = w * h / 64
print("samples/wave-lane", spp, "kernel ms", st["kernel_ms"])
for i, n in enumerate(names):
    ex, lanes = out[2 * i], out[2 * i + 1]
    print("%-6s executions/wave/sample %8.2f  lanes/execution %6.2f  lane-visits/sample %7.2f  cycles/execution %8.1f  cycles/wave/sample %9.0f" % (
        n, ex / waves / spp, lanes / max(ex, 1), lanes / (w * h * spp), out[16 + i] / max(ex, 1), out[16 + i] / waves / spp))
print("claim (sort/vote/barrier) cycles/wave/sample %9.0f   idle %9.0f   push %9.0f   total %9.0f" % (out[32] / waves / spp, out[30] / waves / spp, out[31] / waves / spp, (sum(out[16:24]) + out[30] + out[31] + out[32]) / waves / spp))
seg = ["lds load", "rng+free-flight sample", "grid lookup", "transmittance/decide", "top", "lds store", "-", "-"]
med_exec = max(out[2], 1)
print("MED segments (cycles / execution):", ", ".join("%s %.0f" % (n, out[24 + i] / med_exec) for i, n in enumerate(seg[:6])))
