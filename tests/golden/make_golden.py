"""Generates the golden fixtures tests/golden/*.npz with the oracle (CPU restatement).

    python tests/golden/make_golden.py [case ...]

Fixtures are data only: the scene parameters needed to rebuild the input, the expected XYZAW film and the
loop counters.  They pin (a) the oracle against regressions and (b) the HIP path on the GPU box without
needing the oracle.  The reference itself cannot be run here to produce vectors."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import tests.oracle_binding as ob  # noqa: E402

scenes = importlib.import_module("eradiate-kernel_amd.scenes")
HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    "c1_cornell_32x32x16": ("c1_cornell", dict(width=32, height=32, spp=16)),
    "c2_homogeneous_32x32x16": ("c2_homogeneous_slab", dict(width=32, height=32, spp=16)),
    "c3_heterogeneous_32x32x16_res16": ("c3_heterogeneous", dict(width=32, height=32, spp=16, res=16)),
    "c3_heterogeneous_48x40x8_res32_2pass": ("c3_heterogeneous", dict(width=48, height=40, spp=8, res=32, samples_per_pass=4)),
    "c4_atmosphere_16x16x16": ("c4_atmosphere", dict(width=16, height=16, spp=16, layers=16)),
}

if __name__ == "__main__":
    for name, (fn, kw) in CASES.items():
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        o = ob.OracleScene(getattr(scenes, fn)(**kw))
        film = o.render(threads=1)
        st = o.last_stats
        np.savez_compressed(os.path.join(HERE, name + ".npz"), film=film,
                            counters=np.array([st["n_iter"], st["n_lookup"], st["n_nee_step"], st["samples"]], dtype=np.int64),
                            builder=fn, kwargs=repr(kw))
        print(name, film.shape, float(film[..., :3].mean()), st["n_iter"], st["n_lookup"], st["n_nee_step"])
