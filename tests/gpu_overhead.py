"""Diagnostic: host overhead of one render call (C1 at its BASELINE size): python wall / library wall / kernel time."""
import importlib, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module("eradiate-kernel_amd"); scenes = importlib.import_module("eradiate-kernel_amd.scenes")
pkg.set_variant("gpu_rgb")
scene = pkg.load_dict(scenes.c1_cornell(256, 256, 64)); sensor = scene.sensors()[0]
film = torch.zeros((256, 256, 5), dtype=torch.float32, device="cuda")
integ = scene.integrator()
for mode in ("device film", "host film"):
    ts = []
    for rep in range(12):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        if mode == "device film":
            integ.render(scene, sensor, device_film=film.data_ptr())
        else:
            integ.render(scene, sensor)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        st = integ.last_stats
        ts.append((t1 - t0) * 1e3)
    print("%-12s python wall %.2f ms (min %.2f)   library wall %.2f ms   kernel %.2f ms" % (mode, sorted(ts)[len(ts) // 2], min(ts), st["wall_ms"], st["kernel_ms"]), flush=True)
