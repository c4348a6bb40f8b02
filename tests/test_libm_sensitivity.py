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
