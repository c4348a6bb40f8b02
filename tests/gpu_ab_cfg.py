"""Diagnostic: A/B builds on one of the bench configurations.  usage: python tests/gpu_ab_cfg.py CFG W H SPP lib1.so lib2.so ..."""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
cfg, w, h, spp = sys.argv[1:5]
for lib in sys.argv[5:]:
    env = dict(os.environ, MTSAMD_LIB=os.path.abspath(lib))
    print(os.path.basename(lib), end=": ", flush=True)
    subprocess.run([sys.executable, os.path.join(here, "gpu_c4_threads.py"), w, h, spp, cfg], env=env, timeout=300)
