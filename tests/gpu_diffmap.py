"""Diagnostic: which pixels of a one-block render differ from the oracle (as path ids of the 32x32 block)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("eradiate-kernel_amd"); scenes = importlib.import_module("eradiate-kernel_amd.scenes")
import tests.oracle_binding as ob
pkg.set_variant("gpu_rgb")
spp = int(sys.argv[1])
d = scenes.c3_heterogeneous(32, 32, spp, res=16)
s = pkg.load_dict(d); se = s.sensors()[0]
s.integrator().render(s, se)
gpu = np.array(se.film().bitmap(raw=True)); ref = ob.OracleScene(d).render()
bad = np.argwhere(np.any(gpu != ref, axis=2))
def morton(x, y):
    r = 0
    for b in range(5): r |= ((x >> b) & 1) << (2 * b) | ((y >> b) & 1) << (2 * b + 1)
    return r
ids = sorted(morton(int(x), int(y)) for y, x in bad)
print("mismatching pixels:", len(ids), "path ids:", ids[:40], "..." if len(ids) > 40 else "")
print("weights of bad pixels (W channel):", sorted(set(gpu[tuple(bad.T)][:, 4].tolist()))[:10], "expected", spp)
rel = np.abs(gpu - ref) / np.maximum(np.abs(ref), 1e-6)
print("max rel diff", float(rel.max()), "median rel diff of bad values", float(np.median(rel[gpu != ref])), "sum gpu/ref", float(gpu[..., :3].sum() / ref[..., :3].sum()))
