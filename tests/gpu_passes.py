"""Diagnostic: the metric scene rendered in 1, 2, 4, 8 passes (samples_per_pass): more, shorter workgroups let the hardware balance blocks
of unequal cost over the CUs.  usage: python tests/gpu_passes.py [C3|C4] W H SPP"""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("eradiate-kernel_amd"); scenes = importlib.import_module("eradiate-kernel_amd.scenes")
pkg.set_variant("gpu_rgb")
cfg, w, h, spp = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
mk = {"C3": scenes.c3_heterogeneous, "C4": scenes.c4_atmosphere}[cfg]
for passes in (1, 2, 4, 8):
    d = mk(w, h, spp); d["integrator"]["samples_per_pass"] = spp // passes
    scene = pkg.load_dict(d); sensor = scene.sensors()[0]
    for rep in range(2):
        scene.integrator().render(scene, sensor); st = scene.integrator().last_stats
    print("%s %dx%dx%d in %d passes: kernel %.1f ms (%d launches) -> %.1f Msamples/s" % (cfg, w, h, spp, passes, st["kernel_ms"], st["kernel_launches"], st["samples"] / st["kernel_ms"] / 1e3), flush=True)
