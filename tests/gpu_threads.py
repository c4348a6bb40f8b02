"""Diagnostic: wga kernels with fewer threads than paths (MTSAMD_WG_THREADS)."""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
w, h, spp = sys.argv[1:4]
for cfg in sys.argv[4:]:
    variant, nt = cfg.split(":")
    env = dict(os.environ, MTSAMD_KERNEL=variant, MTSAMD_WG_THREADS=nt)
    print("threads", nt, end=": ", flush=True)
    subprocess.run([sys.executable, os.path.join(here, "gpu_variants.py"), "child", w, h, spp], env=env, timeout=300)
