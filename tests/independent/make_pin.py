"""Generates the fixtures tests/golden/indep_pin_*.npz: per-pixel mean radiance and variance of that mean from the
independent estimator (tests/independent/walk.py) on the C3 / C4 miniatures.

    python -m tests.independent.make_pin [c3] [c4]

Takes a few minutes on one core; tests/test_independent_pin.py Z-tests the oracle (and, under -m gpu, the HIP path)
against these numbers and re-runs the estimator at a small sample count to show that the fixtures come from this code."""
import os
import sys
import time

import numpy as np

from . import problems, walk

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden")
CASES = {   # name: (problem factory, width, height, walks per pixel, seed)
    "c3": (lambda: problems.c3(16, 16, res=16), 16, 16, 3000, 20261004),
    "c4": (lambda: problems.c4(16, 16), 16, 16, 1600, 20261005),
    "c4x3": (lambda: problems.c4x3(16, 16), 16, 16, 1600, 20261006),          # three species: a blendphase nested in a blendphase (round 3)
    # a chromatic slab, one monochromatic run per colour channel (round 3): what spectral MIS is for
    "chroma_r": (lambda: problems.c2_chroma(0), 16, 16, 2000, 20261007),
    "chroma_g": (lambda: problems.c2_chroma(1), 16, 16, 2000, 20261008),
    "chroma_b": (lambda: problems.c2_chroma(2), 16, 16, 2000, 20261009),
}
STEPS, REFINE = 96, 16


def main(names):
    for name in names:
        mk, w, h, per_pixel, seed = CASES[name]
        _, prob, sensor = mk()
        t0 = time.time()
        def progress(done, total):
            if done % 200000 < 20000:
                print("  %s: %d / %d walks, %.0f s" % (name, done, total, time.time() - t0), flush=True)
        mean, var = walk.render(prob, sensor, w, h, per_pixel, seed=seed, steps=STEPS, refine=REFINE, progress=progress)
        se = np.sqrt(var.sum()) / var.size
        print("%s: image mean %.6f, standard error %.2e (%.3f %%), %.0f s" % (name, mean.mean(), se, 100 * se / mean.mean(), time.time() - t0))
        np.savez_compressed(os.path.join(OUT, "indep_pin_%s_%dx%d.npz" % (name, w, h)), mean=mean, var=var,
                            per_pixel=per_pixel, seed=seed, steps=STEPS, refine=REFINE)


if __name__ == "__main__":
    main(sys.argv[1:] or list(CASES))
