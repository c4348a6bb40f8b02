"""Ad-hoc GPU probe: parity of small scenes against the oracle + a first timing of the metric scene."""
import importlib, sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module("eradiate-kernel_amd")
scenes = importlib.import_module("eradiate-kernel_amd.scenes")
import tests.oracle_binding as ob
import tests.transport_cases as tc
pkg.set_variant("gpu_rgb")

def compare(name, d, counters=True):
    scene = pkg.load_dict(d); sensor = scene.sensors()[0]
    scene.integrator().render(scene, sensor, collect_counters=counters)
    gpu = np.array(sensor.film().bitmap(raw=True)); st = scene.integrator().last_stats
    o = ob.OracleScene(d); ref = o.render(); so = o.last_stats
    rel = np.abs(gpu - ref) / np.maximum(np.abs(ref), 1e-6)
    bad = (rel > 1e-3).any(axis=-1).mean()
    print("%-22s exact %.5f  max rel %.3g  pixels>1e-3 %.5f  counters gpu %s oracle %s  kernel %.1f ms" % (
        name, np.mean(gpu == ref), rel.max(), bad, (st["n_iter"], st["n_lookup"], st["n_nee_step"]),
        (so["n_iter"], so["n_lookup"], so["n_nee_step"]), st["kernel_ms"]), flush=True)

compare("C3 64x64x16 res32", scenes.c3_heterogeneous(64, 64, 16, res=32))
compare("C2 64x64x16", scenes.c2_homogeneous_slab(64, 64, 16))
compare("C1 64x64x16", scenes.c1_cornell(64, 64, 16))
compare("C3 100x70x8 res16", scenes.c3_heterogeneous(100, 70, 8, res=16))
for nm, case in [("absorbing", tc.absorbing_slab(4000)), ("single", tc.single_scattering_slab(4000)), ("furnace_het", tc.white_furnace(500, heterogeneous=True))]:
    compare(nm, case[0])
if len(sys.argv) > 1:
    w, h, spp = [int(x) for x in sys.argv[1:4]]
    d = scenes.c3_heterogeneous(w, h, spp)
    scene = pkg.load_dict(d); sensor = scene.sensors()[0]
    for rep in range(2):
        t = time.time(); scene.integrator().render(scene, sensor); st = scene.integrator().last_stats
        print("C3 %dx%dx%d: kernel %.1f ms wall %.1f ms -> %.1f Msamples/s" % (w, h, spp, st["kernel_ms"], st["wall_ms"], st["samples"] / st["kernel_ms"] / 1e3), flush=True)
