"""Diagnostic: timings of the nested kernels (path on C1, volpathmis / nested volpath on C3) -- usage: python tests/gpu_nested_time.py"""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("eradiate-kernel_amd"); scenes = importlib.import_module("eradiate-kernel_amd.scenes")
pkg.set_variant("gpu_rgb")


def run(tag, d, env=None):
    for k, v in (env or {}).items():
        os.environ[k] = v
    scene = pkg.load_dict(d); sensor = scene.sensors()[0]
    best = 1e30
    for rep in range(3):
        scene.integrator().render(scene, sensor); st = scene.integrator().last_stats
        best = min(best, st["kernel_ms"])
    print("%-28s kernel %8.2f ms -> %7.1f Msamples/s" % (tag, best, st["samples"] / best / 1e3), flush=True)
    for k in (env or {}):
        os.environ.pop(k, None)


run("C1 path 512x512x64", scenes.c1_cornell(512, 512, 64))
d = scenes.c3_heterogeneous(512, 512, 32)
d["integrator"] = dict(d["integrator"], type="volpathmis")
run("C3 volpathmis 512x512x32", d)
run("C3 volpath nested 512x512x32", scenes.c3_heterogeneous(512, 512, 32), {"MTSAMD_KERNEL": "nested"})
run("C3 volpath default 512x512x32", scenes.c3_heterogeneous(512, 512, 32))
d = scenes.c4_atmosphere(256, 256, 64)
run("C4 volpath default 256x256x64", d)
d = scenes.c4_atmosphere(256, 256, 64); d["integrator"] = dict(d["integrator"], type="volpathmis")
run("C4 volpathmis 256x256x64", d)
