"""Reference-held literals for the small core pieces the oracle's path / volpath stand on (missing #5):
DiscreteDistribution / ContinuousDistribution (tabphase and mesh area sampling), solve_quadratic (sphere), Morton
decoding (pixel order inside a block), BoundingBox3f (scene bounds, bounding sphere of the distant sensor and the
directional emitter).  Expected values are the ones /root/reference/src/libcore/tests/ asserts."""
import ctypes as C

import numpy as np
import pytest

import tests.oracle_binding as ob

fp = ob.fp


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def discrete(pmf, samples):
    L = ob.lib()
    pmf, samples = _f(pmf), _f(samples)
    n, m = len(pmf), len(samples)
    idx = np.zeros(m, np.int32); reuse = np.zeros(m, np.float32); pn = np.zeros(m, np.float32)
    cdf = np.zeros(n, np.float32); sn = np.zeros(2, np.float32)
    L.oracle_discrete_distribution.argtypes = [fp, C.c_int, fp, C.c_int, C.POINTER(C.c_int32), fp, fp, fp, fp]
    if L.oracle_discrete_distribution(ob._p(pmf), n, ob._p(samples), m, idx.ctypes.data_as(C.POINTER(C.c_int32)), ob._p(reuse), ob._p(pn), ob._p(cdf), ob._p(sn)):
        raise RuntimeError(L.oracle_last_error().decode())
    return {"index": idx, "reuse": reuse, "pmf": pn, "cdf": cdf, "sum": sn[0], "normalization": sn[1]}


def continuous(rng, pdf, x):
    L = ob.lib()
    pdf, x = _f(pdf), _f(x)
    m = len(x)
    out = [np.zeros(m, np.float32) for _ in range(4)]; inn = np.zeros(2, np.float32)
    L.oracle_continuous_distribution.argtypes = [C.c_float, C.c_float, fp, C.c_int, fp, C.c_int, fp, fp, fp, fp, fp]
    if L.oracle_continuous_distribution(rng[0], rng[1], ob._p(pdf), len(pdf), ob._p(x), m, *[ob._p(o) for o in out], ob._p(inn)):
        raise RuntimeError(L.oracle_last_error().decode())
    return {"pdf": out[0], "cdf": out[1], "sample": out[2], "sample_pdf": out[3], "integral": inn[0], "normalization": inn[1]}


def test_discrete_distribution_errors():
    """test_distr_1d.py:5-32"""
    with pytest.raises(RuntimeError, match="empty distribution"):
        discrete([], [0.5])
    with pytest.raises(RuntimeError, match="no probability mass found"):
        discrete([0, 0, 0], [0.5])
    with pytest.raises(RuntimeError, match="entries must be non-negative"):
        discrete([1, -1, 1], [0.5])


def test_discrete_distribution_literals():
    """test_distr_1d.py:35-103 (test04_discr_basic, test05_discr_sample)"""
    eps = 1e-7
    d = discrete([1, 3, 2], [-1, 0, 1, 2])
    assert d["sum"] == 6 and np.isclose(d["normalization"], 1.0 / 6.0) and list(d["cdf"]) == [1, 4, 6]
    assert list(d["index"]) == [0, 0, 2, 2] and np.allclose(d["pmf"], np.array([1, 1, 2, 2]) / 6)
    d = discrete([1, 3, 2], [1 / 6.0 - eps, 1 / 6.0 + eps])
    assert list(d["index"]) == [0, 1] and np.allclose(d["pmf"], np.array([1, 3]) / 6)
    d = discrete([1, 3, 2], [4 / 6.0 - eps, 4 / 6.0 + eps])
    assert list(d["index"]) == [1, 2] and np.allclose(d["pmf"], np.array([3, 2]) / 6)
    d = discrete([1, 3, 2], [0, 1 / 12.0, 1 / 6.0 - eps, 1 / 6.0 + eps])
    assert list(d["index"]) == [0, 0, 0, 1] and np.allclose(d["reuse"], [0, .5, 1, 0], atol=3 * eps)
    assert np.allclose(d["pmf"], np.array([1, 1, 1, 3]) / 6)
    assert list(discrete([1, 1, 1], [0.5])["cdf"]) == [1, 2, 3]


def test_discrete_distribution_bruteforce_and_zero_buckets():
    """test_distr_1d.py:106-132 (test06 with numpy densities instead of the PCG32 ones -- the property is what is tested -- and
    test07_discr_leading_trailing_zeros literally)"""
    rng = np.random.default_rng(3)
    for size in range(2, 20):
        for i in range(2, 50, 3):
            density = rng.integers(0, i, size).astype(np.float32)
            if density.sum() == 0:
                continue
            x = np.linspace(0, 1, 20, dtype=np.float32)
            d = discrete(density, x)
            y = d["index"]
            z = np.where(y > 0, d["cdf"][np.maximum(y - 1, 0)], 0.0)
            xs = x * d["sum"]
            assert np.all((xs > z) | ((xs == 0) & (xs >= z)))
    d = discrete([0, 0, 1, 0, 1, 0, 0, 0], [-100, 0, 0.5, 0.5 + 1e-6, 1, 100])
    assert list(d["index"]) == [2, 2, 2, 4, 4, 4] and list(d["pmf"]) == [.5] * 6


def test_continuous_distribution_errors():
    """test_distr_1d.py:135-183"""
    with pytest.raises(RuntimeError, match="needs at least two entries"):
        continuous([1, 2], [1], [0.5])
    with pytest.raises(RuntimeError, match="invalid range"):
        continuous([1, 1], [1, 1], [0.5])
    with pytest.raises(RuntimeError, match="invalid range"):
        continuous([2, 1], [1, 1], [0.5])
    with pytest.raises(RuntimeError, match="no probability mass found"):
        continuous([1, 2], [0, 0, 0], [0.5])
    with pytest.raises(RuntimeError, match="entries must be non-negative"):
        continuous([1, 2], [1, -1, 1], [0.5])


def test_continuous_distribution_literals():
    """test_distr_1d.py:184-221 (test12_cont_eval, test13_cont_func)"""
    eps = 1e-6
    d = continuous([2, 3], [1, 2], [1, 2 - eps, 2, 2.5, 3, 3 + eps, 4])
    assert np.isclose(d["integral"], 1.5) and np.isclose(d["normalization"], 2.0 / 3.0)
    assert np.allclose(d["pdf"], [0, 0, 2.0 / 3.0, 1.0, 4.0 / 3.0, 0, 0])
    d = continuous([2, 3], [1, 2], [1, 2, 2.5, 3, 4])
    assert np.allclose(d["cdf"], [0, 0, 5.0 / 12.0, 1, 1])
    d = continuous([2, 3], [1, 2], [0, 0.5, 1])
    dx = (np.sqrt(10) - 2) / 2
    assert np.allclose(d["sample"], [2, 2 + dx, 3], rtol=1e-6)            # the reference compares x == [...] in fp32
    assert np.allclose(d["sample_pdf"], [2.0 / 3.0, (4 * dx + 2 * (1 - dx)) / 3.0, 4.0 / 3.0])
    import math
    x = np.linspace(-2, 2, 513, dtype=np.float32)
    y = np.exp(-x.astype(np.float64) ** 2).astype(np.float32)
    d = continuous([-2, 2], y, [0, 0.5, 1])
    assert np.isclose(d["integral"], math.sqrt(math.pi) * math.erf(2.0), rtol=1e-5)
    assert np.allclose(d["sample"], [-2, 0, 2], atol=1e-6)
    d1 = continuous([-2, 2], y, [1.0])
    assert np.isclose(d1["pdf"][0] / d1["normalization"], math.exp(-1), rtol=1e-5)


def test_solve_quadratic_literals():
    """test_math.py:32-35"""
    L = ob.lib()
    L.oracle_solve_quadratic.argtypes = [C.c_double] * 3 + [C.POINTER(C.c_double)] * 2
    def sq(a, b, c):
        x0, x1 = C.c_double(), C.c_double()
        ok = L.oracle_solve_quadratic(a, b, c, C.byref(x0), C.byref(x1))
        return ok, x0.value, x1.value
    assert np.allclose(sq(1, 4, -5), (1, -5, 1))
    assert np.allclose(sq(0, 5, -10), (1, 2, 2))
    assert np.allclose(sq(0, -5, 10), (1, 2, 2))
    assert sq(1, 0, 1)[0] == 0 and sq(0, 0, 1)[0] == 0                     # math.h:371-411: no real root / degenerate


def test_morton_round_trip():
    """test_math.py:74-80 (test07_morton2): decode(encode([123, 456])) with the encoder written here"""
    def encode2(x, y):
        out = 0
        for b in range(16):
            out |= ((x >> b) & 1) << (2 * b) | ((y >> b) & 1) << (2 * b + 1)
        return out
    L = ob.lib()
    for x, y in [(123, 456), (0, 0), (31, 31), (65535, 1)]:
        a, b = C.c_uint32(), C.c_uint32()
        L.oracle_morton_decode(encode2(x, y), C.byref(a), C.byref(b))
        assert (a.value, b.value) == (x, y)


def test_bounding_box_literals():
    """test_bbox.py:6-53: the operations the scene bounds are built with (expand by points and boxes == merge, validity)"""
    L = ob.lib()
    L.oracle_bbox_ops.argtypes = [C.c_int, fp, C.c_int, fp, fp, fp, C.POINTER(C.c_int), fp]
    def ops(points=(), boxes=()):
        p = _f(points).reshape(-1, 3); b = _f(boxes).reshape(-1, 6)
        mn, mx, bs = np.zeros(3, np.float32), np.zeros(3, np.float32), np.zeros(4, np.float32); valid = C.c_int()
        L.oracle_bbox_ops(len(p), ob._p(p), len(b), ob._p(b), ob._p(mn), ob._p(mx), C.byref(valid), ob._p(bs))
        return list(mn), list(mx), bool(valid.value), bs
    assert not ops()[2]                                                    # BBox() is invalid
    assert ops(points=[[0, 1, 2]])[:3] == ([0, 1, 2], [0, 1, 2], True)     # collapsed but valid
    assert ops(boxes=[[0, 1, 2, 0, 1, 2], [1, 2, 3, 2, 3, 5]])[:2] == ([0, 1, 2], [2, 3, 5])      # merge(bbox2, bbox3)
    assert ops(points=[[0, 0, 0]])[:2] == ([0, 0, 0], [0, 0, 0])           # reset + expand
    assert ops(points=[[0, 0, 0], [1, 1, 1]])[:2] == ([0, 0, 0], [1, 1, 1])
    assert ops(points=[[0, 0, 0], [1, 1, 1]], boxes=[[-1, -2, -3, 4, 5, 6]])[:2] == ([-1, -2, -3], [4, 5, 6])
    mn, mx, _, bs = ops(boxes=[[1, 2, 3, 2, 3, 5]])
    assert np.allclose(bs[:3], [1.5, 2.5, 4]) and np.isclose(bs[3], np.sqrt(0.25 + 0.25 + 1.0))      # center(), bounding_sphere (bbox.h:327-331)
