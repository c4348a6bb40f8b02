"""Diagnostic: C1 (cornell box, path) timing -- usage: python tests/gpu_c1_time.py [spp]"""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("eradiate-kernel_amd"); scenes = importlib.import_module("eradiate-kernel_amd.scenes")
pkg.set_variant("gpu_rgb")
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 256
scene = pkg.load_dict(scenes.c1_cornell(512, 512, spp)); sensor = scene.sensors()[0]
best = 1e30
for rep in range(4):
    scene.integrator().render(scene, sensor); st = scene.integrator().last_stats
    best = min(best, st["kernel_ms"])
print("%s C1 512x512x%d: kernel %.2f ms -> %.1f Msamples/s" % (os.path.basename(os.environ.get("MTSAMD_LIB", "default")), spp, best, st["samples"] / best / 1e3), flush=True)
