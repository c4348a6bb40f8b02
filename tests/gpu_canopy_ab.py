"""Diagnostic: a canopy of 200 leaves under the C4 atmosphere (a BVH is built: lean unit c), lean unit against the general kernel.  usage: python tests/gpu_canopy_ab.py"""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("eradiate-kernel_amd"); scenes = importlib.import_module("eradiate-kernel_amd.scenes")
T = importlib.import_module("eradiate-kernel_amd.transform").ScalarTransform4f
pkg.set_variant("gpu_rgb")
d = scenes.c4_atmosphere(512, 512, 64)
rng = np.random.default_rng(11)
for k in range(200):
    c = rng.uniform([-20, -20, 0.2], [20, 20, 3.0])
    d["leaf%03d" % k] = {"type": "rectangle", "to_world": T.translate(c) @ T.rotate(rng.normal(size=3), float(rng.uniform(0, 180))) @ T.scale(1.5),
                         "bsdf": {"type": "bilambertian", "reflectance": {"type": "rgb", "value": [0.1, 0.45, 0.08]}, "transmittance": {"type": "rgb", "value": [0.05, 0.4, 0.04]}}}
films = {}
for lean in ("1", "0", "1", "0"):
    os.environ["MTSAMD_LEAN"] = lean
    scene = pkg.load_dict(d); sensor = scene.sensors()[0]
    scene.integrator().render(scene, sensor)
    st = scene.integrator().last_stats
    films[lean] = np.array(sensor.film().bitmap(raw=True))
    print("MTSAMD_LEAN=%s: variant %d, kernel %.1f ms -> %.1f Msamples/s" % (lean, st["kernel_variant"], st["kernel_ms"], st["samples"] / st["kernel_ms"] / 1e3), flush=True)
print("films equal:", np.array_equal(films["0"], films["1"]))
