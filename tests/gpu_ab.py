"""Diagnostic: A/B several builds of libmtsamd.so (ab/*.so, made with MTSAMD_LIB_OUT=...) on the metric scene.
usage: python tests/gpu_ab.py W H SPP variant lib1.so lib2.so ..."""
import os, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
w, h, spp, variant = sys.argv[1:5]
for lib in sys.argv[5:]:
    env = dict(os.environ, MTSAMD_KERNEL=variant, MTSAMD_LIB=os.path.abspath(lib))
    print(os.path.basename(lib), end=": ", flush=True)
    subprocess.run([sys.executable, os.path.join(here, "gpu_variants.py"), "child", w, h, spp], env=env, timeout=300)
