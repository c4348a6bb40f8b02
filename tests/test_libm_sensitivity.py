"""How far is the oracle from a build on glibc's libm?  

The oracle and the HIP kernels share csrc/pmath.h so that GPU and CPU agree bit for bit; the reference's scalar_rgb calls
libm (enoki's scalar fallbacks).  oracle/liboracle_libm.so is the same restatement with logf / expf / sinf / cosf / cbrtf /
powf from glibc.  A 1-ulp difference matters only where it flips a comparison (sampled_t <= maxt, the roulette test, ...)
and desynchronises the pixel's random stream from there on; everywhere else it moves the result by ~1e-7.  This test
measures both effects per BASELINE configuration (miniatures, 64 spp) and keeps the numbers from drifting; DESIGN.md
section 2 quotes them.  It is a sensitivity measurement, not a parity claim about the reference binary (whose libm,
compiler and fma contraction are unknown)."""
import importlib

import numpy as np
import pytest

import tests.oracle_binding as ob

scenes = importlib.import_module("eradiate-kernel_amd.scenes")

CASES = {
    "C1": lambda: scenes.c1_cornell(64, 64, 64),
    "C2": lambda: scenes.c2_homogeneous_slab(64, 64, 64),
    "C3": lambda: scenes.c3_heterogeneous(64, 64, 64, res=32),
    "C4": lambda: scenes.c4_atmosphere(64, 64, 64),
}
# measured here (glibc 2.35, gcc 11, -mfma -ffp-contract=off): fraction of pixels bit-identical / within 1e-3 relative
MEASURED = {"C1": (0.389, 1.0), "C2": (0.885, 1.0), "C3": (0.144, 0.9995), "C4": (0.111, 0.9958)}


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
    assert abs(identical - MEASURED[name][0]) < 0.1 and within >= MEASURED[name][1] - 5e-3


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
MEASURED_1024 = {"C3": {"identical": 0.049, 1e-3: 0.9961, 1e-2: 0.9990, 3e-2: 1.0, "max_z": 0.15},
                 "C4": {"identical": 0.104, 1e-3: 0.9297, 1e-2: 0.9932, 3e-2: 1.0, "max_z": 0.80}}


@pytest.mark.parametrize("name", sorted(FULL))
def test_libm_build_at_the_sample_count_of_the_metric(name):
    """What "per-pixel radiance within 1e-3 relative of scalar_rgb" (BASELINE.json) means for this backend at 1024 spp.  A pixel whose
    arithmetic never flips a comparison differs from the libm build in the last bits (median 1.4e-7 of the differing pixels); one whose
    stream desynchronises becomes an independent estimate of the same radiance and differs by its Monte Carlo noise (relative standard
    error of a 1024-spp pixel: 5 % on C3, 2 % on C4).  Asserted as measured, not as a looser bound: C3 99.6 % of the pixels within
    1e-3, C4 93 % within 1e-3 and 99.3 % within 1e-2, every pixel within 3e-2 and inside the Z-test with |z| < 1."""
    from scipy.stats import norm
    d = FULL[name](1024)
    a = ob.OracleScene(d).render()
    b = ob.OracleScene(d, libm=True).render()
    assert abs(float(a[..., 4].mean()) - 1024) < 1 and abs(float(b[..., 4].mean()) - 1024) < 1
    la, lb = a[..., :3] / a[..., 4:5], b[..., :3] / b[..., 4:5]
    rel = (np.abs(la - lb) / np.maximum(np.abs(la), 1e-6)).max(-1)
    m = MEASURED_1024[name]
    identical = float((a[..., :3] == b[..., :3]).all(-1).mean())
    assert abs(identical - m["identical"]) < 0.03, identical
    for tol in (1e-3, 1e-2, 3e-2):
        frac = float((rel < tol).mean())
        assert abs(frac - m[tol]) < 0.015, (tol, frac)
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
