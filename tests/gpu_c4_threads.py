"""Diagnostic: C4 timing for a given MTSAMD_WG_THREADS (read from the environment) -- usage: python tests/gpu_c4_threads.py W H SPP"""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("eradiate-kernel_amd"); scenes = importlib.import_module("eradiate-kernel_amd.scenes")
pkg.set_variant("gpu_rgb")
w, h, spp = [int(x) for x in sys.argv[1:4]]
cfg = sys.argv[4] if len(sys.argv) > 4 else "C4"
scene = pkg.load_dict({"C4": scenes.c4_atmosphere, "C3": scenes.c3_heterogeneous, "C2": scenes.c2_homogeneous_slab}[cfg](w, h, spp)); sensor = scene.sensors()[0]
best = 1e30
for rep in range(3):
    scene.integrator().render(scene, sensor); st = scene.integrator().last_stats
    best = min(best, st["kernel_ms"])
print("%s threads=%s %dx%dx%d: kernel %.1f ms -> %.1f Msamples/s" % (cfg, os.environ.get("MTSAMD_WG_THREADS", "default"), w, h, spp, best, st["samples"] / best / 1e3), flush=True)
