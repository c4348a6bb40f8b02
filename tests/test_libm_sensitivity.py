"""How far is the oracle from a build on glibc's libm?  (round 3, next #1.)

The reference's scalar_rgb calls libm (enoki's scalar fallbacks); the oracle and the HIP kernels share csrc/pmath.h so that GPU
and CPU agree bit for bit.  oracle/liboracle_libm.so is the same restatement with logf / expf / sinf / cosf / cbrtf / powf from
glibc.  A 1-ulp difference matters where it flips a comparison (sampled_t <= maxt, the roulette test, ...) and desynchronises the
pixel's random stream from there on.  Rounds 1-3 measured that distance (pmath.h was a ~1.5-ulp implementation: at the metric's
1024 spp only 99.6 % (C3) / 93 % (C4) of the pixels stayed within 1e-3 of the libm build).  Round 4 removed it: pmath.h restates
glibc's own algorithms (tests/test_pmath.py: the same bits for every argument), so on glibc 2.28 .. 2.40 the two builds render THE
SAME FILM, which is what this file now asserts.  On another libm the one-sided bounds of the correctly rounded routines apply
(measured with -DPM_CORRECTLY_ROUNDED against glibc 2.35: 64-spp miniatures C1 0.951 / 1.0, C2 0.998 / 1.0, C3 0.916 / 1.0, C4
0.688 / 0.9998 bit-identical / within 1e-3; at 1024 spp C3 0.817 / 1.0, C4 0.608 / 0.990).  Still not a claim about the reference
BINARY, whose compiler and fma contraction are unknown."""
import importlib

import numpy as np
import pytest

import tests.oracle_binding as ob
from tests.test_pmath import GLIBC

scenes = importlib.import_module("eradiate-kernel_amd.scenes")

CASES = {
    "C1": lambda: scenes.c1_cornell(64, 64, 64),
    "C2": lambda: scenes.c2_homogeneous_slab(64, 64, 64),
    "C3": lambda: scenes.c3_heterogeneous(64, 64, 64, res=32),
    "C4": lambda: scenes.c4_atmosphere(64, 64, 64),
}
# Lower bounds (one-sided) on the fraction of pixels bit-identical / within 1e-3 relative for a libm whose float functions are not
# the ones pmath.h restates (there the correctly rounded routines' numbers of the module docstring are the expectation).
SAME_LIBM = GLIBC is not None and (2, 28) <= GLIBC <= (2, 40)
AT_LEAST = {"C1": (0.90, 0.999), "C2": (0.98, 0.999), "C3": (0.85, 0.999), "C4": (0.60, 0.998)}


@pytest.mark.parametrize("name", sorted(CASES))
def test_libm_build_stays_within_the_tolerance_of_the_metric(name):
    d = CASES[name]()
    a = ob.OracleScene(d).render()
    b = ob.OracleScene(d, libm=True).render()
    la, lb = a[..., :3] / a[..., 4:5], b[..., :3] / b[..., 4:5]
    identical = float((a[..., :3] == b[..., :3]).all(-1).mean())
    rel = np.abs(la - lb) / np.maximum(np.abs(la), 1e-6)
    within = float((rel.max(-1) < 1e-3).mean())
    mean_diff = abs(float(la.mean()) - float(lb.mean())) / float(la.mean())
    print("%s: pixels bit-identical %.4f, within 1e-3 %.4f, image mean differs by %.1e" % (name, identical, within, mean_diff))
    assert within >= 0.99                       # BASELINE.json: per-pixel radiance within 1e-3 relative
    assert mean_diff < 2e-4                     # desynchronised pixels are independent estimates of the same quantity
    if SAME_LIBM:
        assert identical == 1.0 and (a == b).all()          # the same film, bit for bit
    else:
        assert identical >= AT_LEAST[name][0] and within >= AT_LEAST[name][1]


# ---------------------------------------------------------------------------------------------------------------------------------
# The same measurement at the METRIC's sample count: 1024 spp on a 32 x 32 crop of the full-size C3 film
# (512 x 512) and of the C4 film (1024 x 1024).  Measured here (glibc 2.35, gcc 11): fraction of the crop's pixels bit-identical,
# within 1e-3 / 1e-2 / 3e-2 relative of the libm build, and accepted by the per-pixel Z-test of the reference's render tests
# (test_renders.py:63-137, Sidak-corrected 1 %) against the Monte Carlo noise of a 1024-spp pixel.
def _crop(d, x, y, n=32):
    d["sensor"]["film"].update({"crop_offset_x": x, "crop_offset_y": y, "crop_width": n, "crop_height": n})
    return d


FULL = {"C3": lambda spp: _crop(scenes.c3_heterogeneous(512, 512, spp), 240, 300),
        "C4": lambda spp: _crop(scenes.c4_atmosphere(1024, 1024, spp), 600, 420)}
# one-sided bounds for another libm (the correctly rounded routines against glibc 2.35 reach 0.817 / 1.0 and 0.608 / 0.990)
MEASURED_1024 = {"C3": {"identical": 0.70, 1e-3: 0.995, 1e-2: 0.999, 3e-2: 1.0, "max_z": 0.15},
                 "C4": {"identical": 0.50, 1e-3: 0.98, 1e-2: 0.995, 3e-2: 1.0, "max_z": 0.80}}


@pytest.mark.parametrize("name", sorted(FULL))
def test_libm_build_at_the_sample_count_of_the_metric(name):
    """What "per-pixel radiance within 1e-3 relative of scalar_rgb" (BASELINE.json) means for this backend at 1024 spp: on glibc
    2.28 .. 2.40 the build on libm and the build on pmath.h render the same film bit for bit (rounds 1-3: 99.6 % / 93 % of the
    pixels within 1e-3).  On another libm: one-sided bounds, and every pixel inside the per-pixel Z-test of the reference's render
    tests -- a pixel whose stream desynchronises is an independent estimate of the same radiance."""
    from scipy.stats import norm
    d = FULL[name](1024)
    a = ob.OracleScene(d).render()
    b = ob.OracleScene(d, libm=True).render()
    assert abs(float(a[..., 4].mean()) - 1024) < 1 and abs(float(b[..., 4].mean()) - 1024) < 1
    la, lb = a[..., :3] / a[..., 4:5], b[..., :3] / b[..., 4:5]
    rel = (np.abs(la - lb) / np.maximum(np.abs(la), 1e-6)).max(-1)
    m = MEASURED_1024[name]
    identical = float((a[..., :3] == b[..., :3]).all(-1).mean())
    print("%s 1024 spp: identical %.4f, within 1e-3 %.4f, 1e-2 %.4f, 3e-2 %.4f" % (name, identical, float((rel < 1e-3).mean()), float((rel < 1e-2).mean()), float((rel < 3e-2).mean())))
    if SAME_LIBM:
        assert (a == b).all()
        return
    assert identical >= m["identical"], identical
    for tol in (1e-3, 1e-2, 3e-2):
        frac = float((rel < tol).mean())
        assert frac >= m[tol], (tol, frac)
    assert abs(float(la.mean()) / float(lb.mean()) - 1.0) < 1e-4
    # per-pixel variance of a 1024-spp mean from 16 independent 64-spp renders (other seeds)
    imgs = []
    for seed in range(16):
        dd = FULL[name](64)
        dd["sensor"]["sampler"]["seed"] = 1000 + seed
        r = ob.OracleScene(dd).render()
        imgs.append(r[..., 1] / r[..., 4])
    var = np.var(np.array(imgs, np.float64), axis=0, ddof=1) / 16.0
    z = np.abs(la[..., 1] - lb[..., 1]) / np.sqrt(2.0 * var)
    p = 2.0 * norm.sf(z)
    alpha = 1.0 - (1.0 - 0.01) ** (1.0 / z.size)
    print("%s 1024 spp: identical %.4f, within 1e-3 %.4f, 1e-2 %.4f, 3e-2 %.4f, Z-test accepted %.4f, max |z| %.2f"
          % (name, identical, float((rel < 1e-3).mean()), float((rel < 1e-2).mean()), float((rel < 3e-2).mean()), float((p > alpha).mean()), float(z.max())))
    assert (p > alpha).all() and z.max() < 2.0 * max(m["max_z"], 0.5)
