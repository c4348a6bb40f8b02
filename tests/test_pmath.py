"""csrc/pmath.h (the bit-reproducible math shared by host and device) against float64 libm."""
import numpy as np
import tests.oracle_binding as ob

LOG, EXP, SIN, COS, CBRT, POW = range(6)


def ulp_err(got, ref):
    ref32 = np.float32(ref)
    ulp = np.spacing(np.abs(ref32)).astype(np.float64)
    return np.abs(np.float64(got) - ref) / np.maximum(ulp, 1e-45)


def evalf(fn, xs, ys=None):
    L = ob.lib()
    ys = np.zeros_like(xs) if ys is None else ys
    return np.array([L.oracle_math(fn, float(x), float(y)) for x, y in zip(xs, ys)], dtype=np.float32)


def test_log_exp_cbrt_within_1p5_ulp():
    rng = np.random.default_rng(0)
    x = rng.random(20000).astype(np.float32)
    x = x[x > 0]
    assert ulp_err(evalf(LOG, x), np.log(x.astype(np.float64))).max() < 1.5
    big = (rng.random(5000).astype(np.float32) + np.float32(0.5)) * np.float32(2.0) ** rng.integers(-60, 60, 5000).astype(np.float32)
    assert ulp_err(evalf(LOG, big), np.log(big.astype(np.float64))).max() < 1.5
    e = (-87.0 * rng.random(20000)).astype(np.float32)
    assert ulp_err(evalf(EXP, e), np.exp(e.astype(np.float64))).max() < 1.5
    c = ((rng.random(20000) - 0.5) * 20).astype(np.float32)
    assert ulp_err(evalf(CBRT, c), np.cbrt(c.astype(np.float64))).max() < 1.5


def test_sincos_abs_error():
    rng = np.random.default_rng(1)
    x = (rng.random(20000) * 2 * np.pi).astype(np.float32)
    assert np.abs(evalf(SIN, x) - np.sin(x.astype(np.float64))).max() < 2e-7
    assert np.abs(evalf(COS, x) - np.cos(x.astype(np.float64))).max() < 2e-7


def test_pow():
    rng = np.random.default_rng(2)
    x = (rng.random(5000) * 2).astype(np.float32) + np.float32(1e-3)
    y = (rng.random(5000) * 6 - 3).astype(np.float32)
    assert ulp_err(evalf(POW, x, y), np.power(x.astype(np.float64), y.astype(np.float64))).max() < 1.5


def test_special_values():
    L = ob.lib()
    assert L.oracle_math(LOG, 1.0, 0) == 0.0
    assert L.oracle_math(LOG, 0.0, 0) == -np.inf
    assert np.isnan(L.oracle_math(LOG, -1.0, 0))
    assert L.oracle_math(EXP, 0.0, 0) == 1.0
    assert L.oracle_math(EXP, -100.0, 0) == 0.0          # flush-to-zero below FLT_MIN
    assert L.oracle_math(EXP, 100.0, 0) == np.inf
    assert L.oracle_math(SIN, 0.0, 0) == 0.0 and L.oracle_math(COS, 0.0, 0) == 1.0
    assert L.oracle_math(CBRT, 8.0, 0) == 2.0 and L.oracle_math(CBRT, -27.0, 0) == -3.0
    assert L.oracle_math(POW, 2.0, 10.0) == 1024.0 and L.oracle_math(POW, 0.0, 2.0) == 0.0 and L.oracle_math(POW, 3.0, 0.0) == 1.0
