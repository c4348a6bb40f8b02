"""Diagnostic: path tracing of a cornell box holding a large triangle mesh (BVH traversal speed), parity checked at a small size."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("eradiate-kernel_amd"); scenes = importlib.import_module("eradiate-kernel_amd.scenes")
import tests.oracle_binding as ob
from tests.test_gpu_parity import _uv_sphere
pkg.set_variant("gpu_rgb")
n_lat, n_lon, w, spp = [int(x) for x in sys.argv[1:5]]


def scene(w, spp, n_lat, n_lon):
    d = scenes.c1_cornell(w, w, spp)
    v, f = _uv_sphere(n_lat, n_lon, 1.0, (0.5, 0.3, 2.0))
    d["ball"] = {"type": "mesh", "vertex_positions": v, "faces": f, "bsdf": {"type": "diffuse", "reflectance": 0.7}}
    return d, len(f)


d, nf = scene(24, 2, 40, 80)
s = pkg.load_dict(d); se = s.sensors()[0]; s.integrator().render(s, se)
ok = np.array_equal(np.array(se.film().bitmap(raw=True)), ob.OracleScene(d).render())
d, nf = scene(w, spp, n_lat, n_lon)
s = pkg.load_dict(d); se = s.sensors()[0]
for _ in range(2):
    s.integrator().render(s, se); st = s.integrator().last_stats
print("parity %s   %d triangles, %dx%dx%d: kernel %.1f ms -> %.1f Msamples/s" % ("EXACT" if ok else "MISMATCH", nf, w, w, spp, st["kernel_ms"], st["samples"] / st["kernel_ms"] / 1e3))
