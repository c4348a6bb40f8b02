"""Pins the oracle (CPU restatement) with the known answers the reference's own tests hold for the hot path
(SURVEY.md section 8(c)).  Each test names the reference test it restates; literals are copied as data."""
import ctypes as C
import importlib
import math

import numpy as np
import pytest

import tests.oracle_binding as ob

T = importlib.import_module("eradiate-kernel_amd.transform").ScalarTransform4f
coordinate_system = importlib.import_module("eradiate-kernel_amd.transform").coordinate_system


def L():
    return ob.lib()


# ---------------------------------------------------------------- RNG
def test_tea_float32_literals():
    """src/libcore/tests/test_random.py:6-16"""
    lit = {(1, 1): 0.5424730777740479, (1, 2): 0.5079904794692993, (1, 3): 0.4171961545944214, (1, 4): 0.008385419845581055,
           (1, 5): 0.8085528612136841, (2, 1): 0.6939879655838013, (3, 1): 0.6978365182876587, (4, 1): 0.4897364377975464}
    for (a, b), v in lit.items():
        assert L().oracle_tea_float32(a, b, 4) == np.float32(v)


def test_tea_float64_literals():
    """src/libcore/tests/test_random.py:19-29: sample_tea_float64 = bits(tea64 >> 12 | 0x3ff0...) - 1"""
    lit = {(1, 1): 0.5424730799533735, (1, 2): 0.5079905082233922, (1, 3): 0.4171962610608142, (1, 4): 0.008385529523330604,
           (1, 5): 0.80855288317879, (2, 1): 0.6939880404156831, (3, 1): 0.6978365636630994, (4, 1): 0.48973647949223253}
    for (a, b), v in lit.items():
        u = L().oracle_tea64(a, b, 4)
        f = np.array([(u >> 12) | 0x3ff0000000000000], dtype=np.uint64).view(np.float64)[0] - 1.0
        assert f == v
        assert L().oracle_tea32(a, b, 4) == (u >> 32)


def test_pcg32_published_vector():
    """pcg32 demo vector (seed 42, stream 54) -- external fact, the reference holds no PCG32 literals."""
    out = (C.c_uint32 * 6)()
    L().oracle_pcg32(42, 54, 6, out, None)
    assert list(out) == [0xa15c02b7, 0x7b47f409, 0xba1d3330, 0x83d2f293, 0xbfa4784b, 0xcbed606e]


def test_independent_sampler_is_default_pcg32():
    """src/samplers/tests/test_independent.py:28-33: the sampler equals a default-seeded PCG32; next_2d = two next_1d."""
    a = np.zeros(16, np.float32); b = np.zeros(16, np.float32)
    L().oracle_sampler_stream(0, 0x853c49e6748fea9b, 16, a.ctypes.data_as(ob.fp))       # seed(PCG32_DEFAULT_STATE)
    L().oracle_pcg32(0x853c49e6748fea9b, 0xda3e39cb94b95bdb, 16, None, b.ctypes.data_as(ob.fp))
    assert np.array_equal(a, b)
    assert ((a >= 0) & (a < 1)).all()


def test_wavefront_seeding_uses_the_64_bit_instantiation_of_tea():
    """librender/sampler.cpp:89-92 seeds lane idx of a wavefront with rng.seed(sample_tea_64(UInt64(seed), idx), sample_tea_64(idx,
    UInt64(seed))), idx = arange<UInt64>: the template of core/random.h:106-116 instantiated with 64-bit arrays, so sums, shifts and
    xors run in 64 bits (no wrap at 2^32 inside the rounds) and the result is v0 + (v1 << 32) mod 2^64.  Pinned by an evaluation of the
    template written out here in Python integers (by hand from the source text; pcg32 = the published algorithm)."""
    M = (1 << 64) - 1

    def tea64_u64(v0, v1, rounds=4):
        total = 0
        for _ in range(rounds):
            total = (total + 0x9e3779b9) & M
            v0 = (v0 + ((((v1 << 4) & M) + 0xa341316c & M) ^ ((v1 + total) & M) ^ (((v1 >> 5) + 0xc8013ea4) & M))) & M
            v1 = (v1 + ((((v0 << 4) & M) + 0xad90777d & M) ^ ((v0 + total) & M) ^ (((v0 >> 5) + 0x7e95761e) & M))) & M
        return (v0 + ((v1 << 32) & M)) & M

    class Pcg:
        def __init__(self, initstate, initseq):
            self.state, self.inc = 0, ((initseq << 1) | 1) & M
            self.next(); self.state = (self.state + initstate) & M; self.next()

        def next(self):
            old = self.state
            self.state = (old * 0x5851f42d4c957f2d + self.inc) & M
            xorshifted = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
            rot = old >> 59
            return ((xorshifted >> rot) | (xorshifted << ((-rot) & 31))) & 0xFFFFFFFF

        def next_float32(self):
            return float(np.array([(self.next() >> 9) | 0x3f800000], np.uint32).view(np.float32)[0] - np.float32(1.0))

    # the first round already differs from the 32-bit instantiation: (v1 << 4) keeps its upper bits
    assert tea64_u64(1, 1) != L().oracle_tea64(1, 1, 4) and tea64_u64(1, 1) >> 40 != 0
    lanes, count = 40, 6
    for seed in (7, (1 << 40) + 12345):
        out = np.zeros((lanes, count), np.float32)
        L().oracle_wavefront_sampler.argtypes = [C.c_int, C.c_uint64, C.c_int, ob.fp]
        L().oracle_wavefront_sampler(lanes, seed, count, out.ctypes.data_as(ob.fp))
        for i in (0, 1, 2, 17, 39):
            rng = Pcg(tea64_u64(seed, i), tea64_u64(i, seed))
            assert [float(x) for x in out[i]] == [rng.next_float32() for _ in range(count)], (seed, i)


def test_wavefront_streams_as_a_render_mode():
    """Sampler key "wavefront": True = the seeding of the reference's gpu_* variants as a render mode (mts_sensor.sampler_wavefront): one
    stream per (pixel, sample), lane L = pixel * spp + sample (integrator.cpp:143-163).  On the oracle: another realisation of the same
    image, reproducible, independent of the block decomposition (the scalar streams are seeded from the spiral block id and are not),
    and one pass only."""
    import importlib
    scenes = importlib.import_module("eradiate-kernel_amd.scenes")
    d = scenes.c2_homogeneous_slab(40, 24, 64)
    scalar = ob.OracleScene(d).render()
    d["sensor"]["sampler"]["wavefront"] = True
    a = ob.OracleScene(d).render(); b = ob.OracleScene(d).render(threads=1)
    assert np.array_equal(a, b) and not np.array_equal(a, scalar)
    assert abs(a[..., 1].sum() / scalar[..., 1].sum() - 1.0) < 0.02
    d16 = scenes.c2_homogeneous_slab(40, 24, 64); d16["sensor"]["sampler"]["wavefront"] = True; d16["integrator"]["block_size"] = 16
    assert np.array_equal(ob.OracleScene(d16).render(), a)                   # lane numbers follow pixels, not blocks
    s16 = scenes.c2_homogeneous_slab(40, 24, 64); s16["integrator"]["block_size"] = 16
    assert not np.array_equal(ob.OracleScene(s16).render(), scalar)          # ... unlike the scalar seeds (integrator.cpp:198)
    d["integrator"]["samples_per_pass"] = 32
    with pytest.raises(RuntimeError, match="one pass"):
        ob.OracleScene(d)


# ---------------------------------------------------------------- warps / frames
def warp(kind, u, v):
    out = np.zeros(3, np.float32)
    L().oracle_warp(kind, u, v, out.ctypes.data_as(ob.fp))
    return out


def test_warp_corner_cases():
    """src/libcore/tests/test_warp.py:68-77,121-149"""
    s = 1 / math.sqrt(2)
    assert np.allclose(warp(0, 0, 0)[:2], [-s, -s])
    assert np.allclose(warp(0, .5, .5)[:2], [0, 0])
    assert np.allclose(warp(1, 0, 0), [0, 0, 1])
    assert np.allclose(warp(1, 0, 1), [0, 0, -1])
    assert np.allclose(warp(1, .5, .5), [-1, 0, 0], atol=1e-7)
    assert np.allclose(warp(2, .5, .5), [0, 0, 1])
    assert np.allclose(warp(2, 0, .5), [-1, 0, 0])
    assert np.allclose(warp(3, .5, .5), [0, 0, 1])
    assert np.allclose(warp(3, .5, 0), [0, -1, 0], atol=1e-7)


def test_warps_are_unit_vectors_and_uniform():
    rng = np.random.default_rng(3)
    uv = rng.random((2000, 2)).astype(np.float32)
    for kind in (1, 2, 3):
        v = np.array([warp(kind, float(a), float(b)) for a, b in uv])
        assert np.allclose(np.linalg.norm(v, axis=1), 1, atol=2e-6)
        if kind > 1:
            assert (v[:, 2] >= 0).all()
    v = np.array([warp(3, float(a), float(b)) for a, b in uv])
    assert abs(v[:, 2].mean() - 2.0 / 3.0) < 0.02          # E[cos] under the cosine density


def test_coordinate_system_and_directional_matrix():
    """src/emitters/tests/test_directional.py:67-75: direction (0,0,-1) -> [[0,1,0],[1,0,0],[0,0,-1]]"""
    n = np.array([0, 0, -1], np.float32); s = np.zeros(3, np.float32); t = np.zeros(3, np.float32)
    L().oracle_coordinate_system(n.ctypes.data_as(ob.fp), s.ctypes.data_as(ob.fp), t.ctypes.data_as(ob.fp))
    s_py, t_py = coordinate_system(n)
    assert np.array_equal(s, s_py) and np.array_equal(t, t_py)
    m = T.look_at([0, 0, 0], n, s).matrix
    assert np.allclose(m, [[0, 1, 0, 0], [1, 0, 0, 0], [0, 0, -1, 0], [0, 0, 0, 1]])
    rng = np.random.default_rng(4)
    for _ in range(100):
        n = rng.normal(size=3).astype(np.float32); n /= np.linalg.norm(n)
        L().oracle_coordinate_system(n.ctypes.data_as(ob.fp), s.ctypes.data_as(ob.fp), t.ctypes.data_as(ob.fp))
        assert abs(np.dot(s, t)) < 1e-6 and abs(np.dot(s, n)) < 1e-6 and abs(np.dot(t, n)) < 1e-6
        assert np.allclose(np.cross(s, t), n, atol=1e-6)


# ---------------------------------------------------------------- spiral / image block / filters
def spiral(sx, sy, bs, passes=1, max_blocks=4096):
    out = (C.c_int32 * (5 * max_blocks))()
    n = L().oracle_spiral(sx, sy, 0, 0, bs, passes, max_blocks, out)
    return np.array(out[:5 * n]).reshape(n, 5)


def test_spiral_order():
    """src/librender/tests/test_spiral.py:63-80 (318x322 film, 32x32 blocks: 110 blocks, first twelve positions)"""
    b = spiral(318, 322, 32)
    assert len(b) == 110
    c = np.array([160, 160]); w = 32
    exp = [c, c + [w, 0], c + [w, w], c + [0, w], c + [-w, w], c + [-w, 0], c + [-w, -w], c + [0, -w], c + [w, -w],
           c + [2 * w, -w], c + [2 * w, 0], c + [2 * w, w]]
    assert np.array_equal(b[:12, :2], np.array(exp))
    assert (b[:12, 2:4] == 32).all()
    assert np.array_equal(b[:, 4], np.arange(110))
    covered = np.zeros((322, 318), int)
    for ox, oy, sx, sy, _ in b:
        covered[oy:oy + sy, ox:ox + sx] += 1
    assert (covered == 1).all()
    one = spiral(15, 12, 32)                                   # test_spiral.py:52-60: a single block
    assert len(one) == 1 and list(one[0]) == [0, 0, 15, 12, 0]


def test_spiral_pass_ids():
    """src/librender/spiral.cpp:41: block_id = counter + (remaining_passes - 1) * count"""
    b = spiral(64, 64, 32, passes=3)
    assert len(b) == 12
    assert list(b[:, 4]) == [8, 9, 10, 11, 4, 5, 6, 7, 0, 1, 2, 3]


def test_morton_decode():
    x = C.c_uint32(); y = C.c_uint32()
    seen = set()
    for i in range(1024):
        L().oracle_morton_decode(i, C.byref(x), C.byref(y))
        seen.add((x.value, y.value))
        assert x.value < 32 and y.value < 32
    assert len(seen) == 1024
    L().oracle_morton_decode(0b1101, C.byref(x), C.byref(y))
    assert (x.value, y.value) == (0b11, 0b10)


def imageblock_put(w, h, channels, rf, radius, stddev, pos, vals, border=True, ox=0, oy=0):
    pos = np.ascontiguousarray(pos, np.float32); vals = np.ascontiguousarray(vals, np.float32)
    out = np.zeros((h + 16) * (w + 16) * channels, np.float32); b = C.c_int()
    assert L().oracle_imageblock_put(w, h, ox, oy, channels, rf, radius, stddev, int(border), len(pos),
                                     pos.ctypes.data_as(ob.fp), vals.ctypes.data_as(ob.fp), out.ctypes.data_as(ob.fp), C.byref(b)) == 0
    bs = b.value
    return out[:(h + 2 * bs) * (w + 2 * bs) * channels].reshape(h + 2 * bs, w + 2 * bs, channels), bs


def test_imageblock_put_box():
    """src/librender/tests/test_imageblock.py:51-98: one sample in the centre of each pixel lands in that pixel"""
    w, h = 10, 8
    rng = np.random.default_rng(5)
    vals = rng.random((h * w, 5)).astype(np.float32)
    pos = np.array([[j + 0.5, i + 0.5] for i in range(h) for j in range(w)], np.float32)
    blk, border = imageblock_put(w, h, 5, 0, 0.4, 0.5, pos, vals)
    assert border == 0
    assert np.array_equal(blk.reshape(-1, 5), vals)
    blk, border = imageblock_put(w, h, 5, 0, 0.5, 0.5, pos, vals)          # default box radius
    assert border == 0 and np.array_equal(blk.reshape(-1, 5), vals)


def test_imageblock_put_gaussian_matches_numpy():
    """src/librender/tests/test_imageblock.py:100-144 (numpy re-implementation of the filtered splat)"""
    w, h = 9, 7
    rng = np.random.default_rng(6)
    pos = (rng.random((40, 2)) * [w, h]).astype(np.float32)
    vals = rng.random((40, 5)).astype(np.float32)
    blk, border = imageblock_put(w, h, 5, 1, 0.5, 0.5, pos, vals)
    assert border == 2
    radius, res = 2.0, 31
    alpha = -1.0 / (2 * 0.5 * 0.5); bias = math.exp(alpha * radius * radius)
    table = np.array([max(0.0, math.exp(alpha * (radius * i / res) ** 2) - bias) for i in range(res)] + [0.0])
    ref = np.zeros((h + 2 * border, w + 2 * border, 5))
    for p, v in zip(pos.astype(np.float64), vals.astype(np.float64)):
        q = p - (-border + 0.5)
        lo = np.maximum(np.ceil(q - radius).astype(int), 0)
        hi = np.minimum(np.floor(q + radius).astype(int), [w + 2 * border - 1, h + 2 * border - 1])
        for y in range(lo[1], hi[1] + 1):
            for x in range(lo[0], hi[0] + 1):
                wx = table[min(int(abs((x - q[0]) * res / radius)), res)]
                wy = table[min(int(abs((y - q[1]) * res / radius)), res)]
                ref[y, x] += v * wx * wy
    assert np.allclose(blk, ref, rtol=1e-4, atol=1e-6)


def test_rfilter_tables():
    """src/libcore/rfilter.cpp:9-20, src/rfilters/{box,gaussian}.cpp"""
    f = L().oracle_rfilter_eval
    assert f(0, 0.5, 0.5, 0.0, 0) == 1.0 and f(0, 0.5, 0.5, 0.5, 0) == 1.0 and f(0, 0.5, 0.5, 0.51, 0) == 0.0
    assert abs(f(1, 0.5, 0.5, 0.0, 0) - (1 - math.exp(-8))) < 1e-6
    assert f(1, 0.5, 0.5, 2.0, 0) == 0.0 and f(1, 0.5, 0.5, 2.0, 1) == 0.0
    assert abs(f(1, 0.5, 0.5, 1.0, 1) - f(1, 0.5, 0.5, 2.0 * 15 / 31, 0)) < 1e-7     # discretised: bin floor(|x| * 31 / r)
    # the reference's spot checks, src/rfilters/tests/test_rfilter.py:8-22
    assert f(0, 0.5, 0.5, 0.49, 0) == 1 and f(0, 0.5, 0.5, 0.51, 0) == 0 and f(0, 0.5, 0.5, 0.49, 1) == 1 and f(0, 0.5, 0.5, 0.51, 1) == 0
    assert abs(f(1, 2.0, 0.5, 0.2, 0) - 0.9227) < 8e-3 and abs(f(1, 2.0, 0.5, 0.2, 1) - 0.9227) < 8e-3
    assert f(1, 2.0, 0.5, 2.1, 0) == 0 and f(1, 2.0, 0.5, 2.1, 1) == 0


# ---------------------------------------------------------------- scene-level known answers
@pytest.fixture(scope="module")
def plugin_scene():
    """One scene holding every phase function / BSDF of the hot path, addressed by index."""
    d = {
        "type": "scene",
        "integrator": {"type": "volpath"},
        "sensor": {"type": "perspective", "film": {"type": "hdrfilm", "width": 4, "height": 4, "rfilter": {"type": "box"}}},
        "p0_iso": {"type": "isotropic"},
        "p1_hg": {"type": "hg", "g": 0.3},
        "p2_rayleigh": {"type": "rayleigh"},
        "p3_tab": {"type": "tabphase", "values": "0.5, 1.0, 1.5"},
        "p4_blend": {"type": "blendphase", "phase1": {"type": "isotropic"}, "phase2": {"type": "hg", "g": 0.2}, "weight": 0.2},
        "b0_diffuse": {"type": "diffuse"},
        "b1_null": {"type": "null"},
        "b2_rpv": {"type": "rpv", "rho_0": 0.2, "k": 0.7, "g": -0.1},
        "b3_rpv_lambert": {"type": "rpv", "rho_0": 0.5, "k": 1.0, "g": 0.0, "rho_c": 1.0},
        "shape": {"type": "rectangle"},
    }
    o = ob.OracleScene(d)
    phases = {"iso": 0, "hg": 1, "rayleigh": 2, "tab": 3, "blend_child0": 4, "blend_child1": 5, "blend": 6}
    return o, phases


def test_phase_indices(plugin_scene):
    o, ph = plugin_scene
    types = [o.desc.phases[i].type for i in range(o.desc.phase_count)]
    assert types == [0, 1, 2, 4, 0, 1, 3]


def test_isotropic_phase(plugin_scene):
    """src/phase/tests/test_isotropic.py:11-23: eval = pdf = 1/(4 pi)"""
    o, ph = plugin_scene
    for theta in np.linspace(0, np.pi, 7):
        for phi in np.linspace(0, 2 * np.pi, 5):
            wo = [np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta)]
            assert np.isclose(o.phase_eval(ph["iso"], [0, 0, 1], wo), 1 / (4 * np.pi))
    wo, pdf = o.phase_sample(ph["iso"], [0, 0, 1], 0.3, (0.2, 0.7))
    assert np.isclose(pdf, 1 / (4 * np.pi)) and np.isclose(np.linalg.norm(wo), 1, atol=1e-6)


def chi2_sample_vs_pdf(o, phase, wi, n=60000, bins=20):
    """Reduced form of the chi-square test of src/python/python/chi2.py:98-316 over cos(theta) bins."""
    rng = np.random.default_rng(11)
    u = rng.random((n, 3)).astype(np.float32)
    cos = np.empty(n)
    for i in range(n):
        wo, _ = o.phase_sample(phase, wi, float(u[i, 0]), (float(u[i, 1]), float(u[i, 2])))
        cos[i] = np.dot(wo, wi)
    hist, edges = np.histogram(cos, bins=bins, range=(-1, 1))
    xs = np.linspace(-1, 1, bins * 64 + 1)
    wi = np.asarray(wi, np.float64)
    # pdf as a function of cos(angle between wo and wi), integrated over azimuth
    a = np.array([1.0, 0.0, 0.0]) if abs(wi[2]) > 0.9 else np.array([0.0, 0.0, 1.0])
    s = np.cross(wi, a); s /= np.linalg.norm(s)
    pdf = np.array([o.phase_eval(phase, wi, list(x * wi + math.sqrt(max(0, 1 - x * x)) * s)) for x in xs]) * 2 * np.pi
    cdf = np.concatenate([[0], np.cumsum((pdf[1:] + pdf[:-1]) * 0.5 * np.diff(xs))])
    expected = np.diff(cdf[::64]) * n
    assert abs(cdf[-1] - 1) < 2e-3                                     # normalisation
    chi2 = ((hist - expected) ** 2 / np.maximum(expected, 1e-9)).sum()
    return chi2, bins - 1


@pytest.mark.parametrize("name", ["hg", "rayleigh", "tab", "blend"])
def test_phase_sample_matches_pdf(plugin_scene, name):
    """src/phase/tests/test_hg.py:8-22, test_rayleigh.py:8-22, test_tabphase.py, test_blendphase.py (chi^2)"""
    o, ph = plugin_scene
    chi2, dof = chi2_sample_vs_pdf(o, ph[name], [0.0, 0.6, 0.8], n=20000)
    assert chi2 < dof + 5 * math.sqrt(2 * dof)


def test_hg_convention_and_sampled_pdf(plugin_scene):
    """src/phase/hg.cpp:52-84: eval(wo) = hg(cos(wo, wi)); the sampled pdf equals eval at the sampled direction"""
    o, ph = plugin_scene
    g = 0.3
    wi = np.array([0, 0, 1.0])
    for c in (-1.0, -0.3, 0.5, 1.0):
        wo = [math.sqrt(1 - c * c), 0, c]
        expected = (1 - g * g) / (4 * np.pi * (1 + g * g + 2 * g * c) ** 1.5)
        assert np.isclose(o.phase_eval(ph["hg"], wi, wo), expected, rtol=1e-5)
    wo, pdf = o.phase_sample(ph["hg"], wi, 0.1, (0.35, 0.8))
    assert np.isclose(pdf, o.phase_eval(ph["hg"], wi, wo), rtol=1e-4)


def test_tabphase_eval(plugin_scene):
    """src/phase/tests/test_tabphase.py:24-66: linear interpolation of the table over cos(theta), normalised, / 2 pi"""
    o, ph = plugin_scene
    ref_x = np.array([-1, 0, 1.0]); ref_y = np.array([0.5, 1.0, 1.5])
    integral = np.trapezoid(ref_y, ref_x)
    wi = [0, 0, 1]
    for c in np.linspace(-1, 1, 11):
        wo = [math.sqrt(max(0, 1 - c * c)), 0, c]
        expected = np.interp(-c, ref_x, ref_y) / integral / (2 * np.pi)
        assert np.isclose(o.phase_eval(ph["tab"], wi, wo), expected, rtol=1e-5)


def test_blendphase_eval_and_component_choice(plugin_scene):
    """src/phase/tests/test_blendphase.py:36-108: eval = (1 - w) iso + w hg; sample1 > w picks the first component"""
    o, ph = plugin_scene
    w, g = 0.2, 0.2
    wi = [0, 0, 1]; wo = [0, 0, 1]
    hg = (1 - g * g) / (4 * np.pi * (1 + g * g + 2 * g) ** 1.5)
    assert np.isclose(o.phase_eval(ph["blend"], wi, wo), (1 - w) / (4 * np.pi) + w * hg, rtol=1e-5)
    wo_a, pdf_a = o.phase_sample(ph["blend"], wi, 0.3, (0.4, 0.6))          # > weight: isotropic
    wo_i, pdf_i = o.phase_sample(ph["iso"], wi, 0.0, (0.4, 0.6))
    assert np.array_equal(wo_a, wo_i) and pdf_a == pdf_i
    wo_b, pdf_b = o.phase_sample(ph["blend"], wi, 0.1, (0.4, 0.6))          # <= weight: hg
    wo_h, pdf_h = o.phase_sample(ph["blend_child1"], wi, 0.5, (0.4, 0.6))
    assert np.array_equal(wo_b, wo_h) and pdf_b == pdf_h


def test_blendphase_components_and_nested_trees():
    """src/phase/tests/test_blendphase.py:111-200 (test04 / test05: ctx.component addresses one component, its value and pdf carry the
    blending weight) on the reference's own two-component phase function, and the same rules one level down: a blendphase whose second
    child is a blendphase (blendphase.cpp:42-66 holds arbitrary children, :58-61 concatenates their components, :92-108 hands the
    rescaled sample1 down)."""
    w, g = 0.2, 0.2
    base = {"type": "scene", "integrator": {"type": "volpath"},
            "sensor": {"type": "perspective", "film": {"type": "hdrfilm", "width": 4, "height": 4, "rfilter": {"type": "box"}}}}
    o = ob.OracleScene(dict(base, ph={"type": "blendphase", "phase1": {"type": "isotropic"}, "phase2": {"type": "hg", "g": g}, "weight": w}))
    root = o.desc.phase_count - 1
    wi = [0, 0, 1]; wo = [0, 0, 1]
    inv4pi = 1 / (4 * np.pi)
    hg_fwd = inv4pi * (1 - g) / (1 + g) ** 2                            # eval_hg at cos = 1 (wo = wi: back scattering in the phase convention)
    v0, n = o.phase_eval_component(root, 0, wi, wo)
    v1, _ = o.phase_eval_component(root, 1, wi, wo)
    assert n == 2 and np.isclose(v0, (1 - w) * inv4pi) and np.isclose(v1, w * hg_fwd)                   # test04
    for s1 in (0.3, 0.1):                                                                               # test05: the selected component is always sampled
        assert np.isclose(o.phase_sample_component(root, 0, wi, s1, (0.5, 0.5))[1], (1 - w) * inv4pi)
        assert np.isclose(o.phase_sample_component(root, 1, wi, s1, (0.0, 0.0))[1], w * hg_fwd)
    # ---- nested: blend(iso, blend(hg(g), rayleigh; w2); w1)
    w1, w2 = 0.6, 0.25
    o = ob.OracleScene(dict(base, ph={"type": "blendphase", "a": {"type": "isotropic"},
                                      "b": {"type": "blendphase", "a": {"type": "hg", "g": g}, "b": {"type": "rayleigh"}, "weight": w2}, "weight": w1}))
    root = o.desc.phase_count - 1
    assert [o.desc.phases[i].type for i in range(o.desc.phase_count)] == [0, 1, 2, 3, 3]                # children first
    ray = 3 / (16 * np.pi) * 2.0
    full = o.phase_eval(root, wi, wo)
    parts = [o.phase_eval_component(root, c, wi, wo) for c in range(3)]
    assert all(n == 3 for _, n in parts)
    assert np.allclose([v for v, _ in parts], [(1 - w1) * inv4pi, w1 * (1 - w2) * hg_fwd, w1 * w2 * ray], rtol=1e-6)
    assert np.isclose(sum(v for v, _ in parts), full, rtol=1e-6)
    # sample1 > w1: isotropic; else the inner blend sees sample1 / w1: > w2 -> hg, <= w2 -> rayleigh (blendphase.cpp:92-108)
    hg_i, ray_i = 1, 2
    for s1, leaf in ((0.9, 0), (0.3, hg_i), (0.12, ray_i), (0.15 + 1e-4, hg_i), (0.15 - 1e-4, ray_i)):
        wo_t, pdf_t = o.phase_sample(root, wi, s1, (0.35, 0.8))
        wo_l, pdf_l = o.phase_sample(leaf, wi, 0.5, (0.35, 0.8))
        assert np.array_equal(wo_t, wo_l) and pdf_t == pdf_l, (s1, leaf)
    # a component of the inner blend: pdf carries both weights
    assert np.isclose(o.phase_sample_component(root, 2, wi, 0.9, (0.35, 0.8))[1], w1 * w2 * o.phase_sample(ray_i, wi, 0.5, (0.35, 0.8))[1])


def test_diffuse_bsdf(plugin_scene):
    """src/bsdfs/tests/test_diffuse.py:16-38: eval = rho cos/pi, pdf = cos/pi with rho = 0.5"""
    o, _ = plugin_scene
    wi = [0, 0, 1]
    for theta in np.linspace(0, np.pi / 2, 20):
        wo = [np.sin(theta), 0, np.cos(theta)]
        v, pdf = o.bsdf_eval(0, wi, wo)
        assert np.allclose(v, 0.5 * np.cos(theta) / np.pi, atol=1e-7) and np.isclose(pdf, np.cos(theta) / np.pi, atol=1e-7)
    v, pdf = o.bsdf_eval(0, wi, [0, 0, -1])
    assert np.all(v == 0) and pdf == 0
    wo, pdf, wgt, st = o.bsdf_sample(0, wi, 0.5, (0.2, 0.9))
    assert st == 0x2 and np.allclose(wgt, 0.5) and np.isclose(pdf, wo[2] / np.pi)
    wo, pdf, wgt, st = o.bsdf_sample(0, [0, 0, -1], 0.5, (0.2, 0.9))         # back side: nothing sampled
    assert st == 0 and np.all(wgt == 0) and pdf == 0


def test_null_bsdf(plugin_scene):
    """src/bsdfs/null.cpp:41-72"""
    o, _ = plugin_scene
    wo, pdf, wgt, st = o.bsdf_sample(1, [0.3, 0.1, 0.9], 0.5, (0.2, 0.9))
    assert st == 0x1 and pdf == 1 and np.all(wgt == 1) and np.allclose(wo, [-0.3, -0.1, -0.9])
    v, pdf = o.bsdf_eval(1, [0, 0, 1], [0, 0, 1])
    assert np.all(v == 0) and pdf == 0


def rpv_reference(rho_0, k, g, rho_c, wi, wo):
    """Independent numpy formula of src/bsdfs/tests/test_rpv.py:35-57 (angles from direction vectors)."""
    ti, to = math.acos(wi[2]), math.acos(wo[2])
    pi_, po = math.atan2(wi[1], wi[0]), math.atan2(wo[1], wo[0])
    cos_g = math.cos(ti) * math.cos(to) + math.sin(ti) * math.sin(to) * math.cos(pi_ - po)
    G = math.sqrt(max(0, math.tan(ti) ** 2 + math.tan(to) ** 2 - 2 * math.tan(ti) * math.tan(to) * math.cos(pi_ - po)))
    F = (1 - g * g) / (1 + g * g + 2 * g * cos_g) ** 1.5
    M = (math.cos(ti) * math.cos(to) * (math.cos(ti) + math.cos(to))) ** (k - 1)
    return rho_0 * M * F * (1 + (1 - rho_c) / (1 + G)) / math.pi * abs(math.cos(to))


def test_rpv_bsdf(plugin_scene):
    """src/bsdfs/tests/test_rpv.py:77-106 (closed form, rtol 1e-3) and :109-150 (k=1, g=0, rho_c=1 is Lambertian)"""
    o, _ = plugin_scene
    rng = np.random.default_rng(8)
    for _ in range(50):
        ti, to = rng.random(2) * 1.4
        pi_, po = rng.random(2) * 2 * np.pi
        wi = [math.sin(ti) * math.cos(pi_), math.sin(ti) * math.sin(pi_), math.cos(ti)]
        wo = [math.sin(to) * math.cos(po), math.sin(to) * math.sin(po), math.cos(to)]
        v, pdf = o.bsdf_eval(2, wi, wo)
        assert np.allclose(v, rpv_reference(0.2, 0.7, -0.1, 0.2, wi, wo), rtol=1e-3)
        assert np.isclose(pdf, wo[2] / np.pi, rtol=1e-5)
        v, _ = o.bsdf_eval(3, wi, wo)
        assert np.allclose(v, 0.5 * wo[2] / np.pi, rtol=1e-3)


def test_rectangle_hits():
    """src/shapes/tests/test_rectangle.py:37-63: rectangle scale(2, .5, 1); rays o=(a,a,5), d=-z; hit iff |a| <= 0.5"""
    d = {"type": "scene", "integrator": {"type": "path"},
         "sensor": {"type": "perspective", "film": {"type": "hdrfilm", "width": 4, "height": 4, "rfilter": {"type": "box"}}},
         "foo": {"type": "rectangle", "to_world": T.scale([2.0, 0.5, 1.0])}}
    o = ob.OracleScene(d)
    coords = np.linspace(-1, 1, 15, dtype=np.float32)
    orig = np.stack([coords, coords, np.full(15, 5, np.float32)], 1)
    dirs = np.tile(np.array([0, 0, -1], np.float32), (15, 1))
    r = o.ray_intersect(orig, dirs, mint=np.zeros(15, np.float32))
    valid = np.isfinite(r["t"])
    assert np.array_equal(valid, np.abs(coords) <= 0.5)
    assert valid.sum() == 7
    assert np.allclose(r["t"][valid], 5) and np.allclose(r["n"][valid], [0, 0, 1])


def test_mesh_traversal_matches_brute_force_depth():
    """src/librender/tests/test_kdtrees.py:26-60: staircase mesh, t = 2 - step / 20 along -z"""
    n = 10
    verts, faces = [], []
    for i in range(n):
        h = i / 20.0; x0, x1 = i / n, (i + 1) / n
        b = len(verts)
        verts += [[x0, 0, h], [x1, 0, h], [x1, 1, h], [x0, 1, h]]
        faces += [[b, b + 1, b + 2], [b, b + 2, b + 3]]
    d = {"type": "scene", "integrator": {"type": "path"},
         "sensor": {"type": "perspective", "film": {"type": "hdrfilm", "width": 4, "height": 4, "rfilter": {"type": "box"}}},
         "stairs": {"type": "mesh", "vertex_positions": np.array(verts, np.float32), "faces": np.array(faces, np.uint32)}}
    o = ob.OracleScene(d)
    xs = (np.arange(n) + 0.5) / n
    orig = np.stack([xs, np.full(n, 0.4), np.full(n, 2.0)], 1).astype(np.float32)
    dirs = np.tile(np.array([0, 0, -1], np.float32), (n, 1))
    r = o.ray_intersect(orig, dirs)
    assert np.allclose(r["t"], 2 - np.arange(n) / 20.0, atol=1e-6)
    assert np.array_equal(r["prim_index"] // 2, np.arange(n))


def test_cube_faces_and_normals():
    """src/shapes/cube.cpp:43-67: unit cube, outward face normals"""
    d = {"type": "scene", "integrator": {"type": "path"},
         "sensor": {"type": "perspective", "film": {"type": "hdrfilm", "width": 4, "height": 4, "rfilter": {"type": "box"}}},
         "c": {"type": "cube"}}
    o = ob.OracleScene(d)
    for axis in range(3):
        for sgn in (-1, 1):
            orig = np.array([0.1, 0.2, 0.3], np.float32); orig[axis] = 5 * sgn
            dr = np.zeros(3, np.float32); dr[axis] = -sgn
            r = o.ray_intersect(orig[None], dr[None])
            assert np.isclose(r["t"][0], 4)
            nn = np.zeros(3); nn[axis] = sgn
            assert np.allclose(r["n"][0], nn, atol=1e-6)


def test_sphere_hits():
    """src/shapes/sphere.cpp:272-306: near/far roots, rays starting inside"""
    d = {"type": "scene", "integrator": {"type": "path"},
         "sensor": {"type": "perspective", "film": {"type": "hdrfilm", "width": 4, "height": 4, "rfilter": {"type": "box"}}},
         "s": {"type": "sphere", "center": [1, 2, 3], "radius": 2.0}}
    o = ob.OracleScene(d)
    r = o.ray_intersect([[1, 2, 10]], [[0, 0, -1]])
    assert np.isclose(r["t"][0], 5) and np.allclose(r["p"][0], [1, 2, 5]) and np.allclose(r["n"][0], [0, 0, 1])
    r = o.ray_intersect([[1, 2, 3]], [[0, 0, -1]])                 # from the centre: far root
    assert np.isclose(r["t"][0], 2) and np.allclose(r["n"][0], [0, 0, -1])
    r = o.ray_intersect([[5, 2, 10]], [[0, 0, -1]])                # miss
    assert np.isinf(r["t"][0]) and r["shape"][0] == -1


# ---------------------------------------------------------------- distant sensor / directional emitter
def distant_scene(sensor):
    return {"type": "scene", "integrator": {"type": "path"}, "sensor": sensor,
            "shape": {"type": "rectangle", "bsdf": {"type": "diffuse", "reflectance": 1.0}},
            "emitter": {"type": "directional", "direction": [0, 0, -1], "irradiance": 1.0}}


def test_distant_sensor_rays():
    """src/sensors/tests/test_distant.py:139-297: ray directions / origins of the distant sensor"""
    film1 = {"type": "hdrfilm", "width": 1, "height": 1, "rfilter": {"type": "box"}}
    for direction in ([0, 0, 1], [0, 1, 1], [1, 1, 1]):
        dn = np.array(direction, np.float64) / np.linalg.norm(direction)
        o = ob.OracleScene(distant_scene({"type": "distant", "direction": direction, "ray_target": [0, 0, 0], "film": film1}))
        org, dr, w = o.sensor_sample_ray([[0.5, 0.5]], [[0.3, 0.6]])
        assert np.allclose(dr[0], -dn, atol=1e-6)                              # rays point opposite to `direction`
        radius = np.sqrt(2.0) * (1 + 1500 * 2.0 ** -24)                        # bounding sphere of the unit rectangle
        assert np.allclose(org[0], 2 * radius * dn, rtol=1e-5) and np.allclose(w[0], 1)
    # no target: origins on the bounding-sphere disk, weight 1 / cos (distant.cpp:355-365)
    o = ob.OracleScene(distant_scene({"type": "distant", "direction": [0, 0, 1], "film": film1}))
    rng = np.random.default_rng(9)
    ap = rng.random((200, 2)).astype(np.float32)
    org, dr, w = o.sensor_sample_ray(np.full((200, 2), 0.5, np.float32), ap)
    assert np.allclose(dr, [0, 0, -1], atol=1e-6) and np.allclose(w, 1)
    assert (np.linalg.norm(org[:, :2], axis=1) <= np.sqrt(2) * 1.001).all() and np.allclose(org[:, 2], np.sqrt(2), rtol=1e-4)
    # film-sized hemisphere of directions (SampleAll): direction = -to_world * square_to_uniform_hemisphere(film sample)
    film = {"type": "hdrfilm", "width": 8, "height": 8, "rfilter": {"type": "box"}}
    o = ob.OracleScene(distant_scene({"type": "distant", "ray_target": [0, 0, 0], "film": film}))
    fs = rng.random((50, 2)).astype(np.float32)
    org, dr, w = o.sensor_sample_ray(fs, np.full((50, 2), 0.5, np.float32))
    exp = -np.array([warp(2, float(a), float(b)) for a, b in fs])
    assert np.allclose(dr, exp, atol=1e-6) and (dr[:, 2] <= 0).all()
    # target shape: points on the rectangle, weight 1 / (pdf * area) = 1
    o = ob.OracleScene(distant_scene({"type": "distant", "direction": [0, 0, 1], "film": film1,
                                      "ray_target": {"type": "rectangle", "to_world": T.scale(0.5)}}))
    org, dr, w = o.sensor_sample_ray(np.full((100, 2), 0.5, np.float32), rng.random((100, 2)).astype(np.float32))
    assert np.allclose(w, 1, rtol=1e-5) and (np.abs(org[:, :2]) <= 0.5 + 1e-6).all()


def test_directional_emitter_sample_direction():
    """src/emitters/directional.cpp:109-141: delta direction, pdf 1, dist = 2 R"""
    film1 = {"type": "hdrfilm", "width": 1, "height": 1, "rfilter": {"type": "box"}}
    o = ob.OracleScene(distant_scene({"type": "distant", "direction": [0, 0, 1], "ray_target": [0, 0, 0], "film": film1}))
    d, dist, pdf, spec = o.emitter_sample_direction([0.2, 0.1, 0.0], 0.3, 0.4)
    assert np.allclose(d, [0, 0, 1], atol=1e-7) and pdf == 1 and np.allclose(spec, 1)
    assert np.isclose(dist, 2 * np.sqrt(2) * (1 + 1500 * 2.0 ** -24), rtol=1e-6)


def _distant_render_scene(setup, w_e, w_o, spp):
    sensor = {"type": "distant", "direction": w_o, "sampler": {"type": "independent", "sample_count": spp},
              "film": {"type": "hdrfilm", "height": 1, "width": 1, "rfilter": {"type": "box"}}}
    if setup == "target_point":
        sensor["ray_target"] = [0, 0, 0]
    elif setup == "target_disk":
        sensor["ray_target"] = {"type": "disk", "to_world": T.scale(1.0)}
    elif setup != "default":
        scale = {"target_square": 1.0, "target_square_small": 0.5, "target_square_large": 2.0}[setup]
        sensor["ray_target"] = {"type": "rectangle", "to_world": T.scale(scale)}
    return {"type": "scene", "integrator": {"type": "path"}, "sensor": sensor,
            "shape": {"type": "rectangle", "to_world": T.scale(1.0), "bsdf": {"type": "diffuse", "reflectance": 1.0}},
            "emitter": {"type": "directional", "direction": w_e, "irradiance": 1.0}}


# The reference's assertion (test_distant.py:462-475) holds at ITS sample count (1e5) and ITS tolerances (5e-3; 1e-2 for the large
# target) for every combination except these: emitter direction [0, 1, -1] with a target that makes every sample the same
# constant.  There the red channel comes out +0.5056 % (G -0.3224 %, B -0.0961 %), 0.0056 % outside the 5e-3 bound.  The cause is
# not a last-ulp difference upstream: summing a constant v 1e5 times in fp32, sample by sample as ImageBlock::put does in
# scalar_rgb (imageblock.cpp:163-168), quantises v to the ulp of the running sum (X = 0.2139 is 109.53 ulps of a sum in
# [16384, 32768): every addition rounds to 110), and XYZ -> RGB amplifies the per-channel biases.  test_constant_sum_bias_is_not_an_ulp_effect
# below shows that all 27 one-ulp perturbations of the per-sample XYZ give the identical 0.5056 %, i.e. any implementation that
# accumulates the block in fp32 lands on this number.
DISTANT_MARGINAL = {(s, (0, 1, -1), w) for s in ("target_square", "target_square_small", "target_disk", "target_point") for w in ((0, 0, 1), (0, 1, 1))}


@pytest.mark.parametrize("setup", ["default", "target_square", "target_square_small", "target_square_large", "target_disk", "target_point"])
@pytest.mark.parametrize("w_e", [[0, 0, -1], [0, 1, -1]])
@pytest.mark.parametrize("w_o", [[0, 0, 1], [0, 1, 1]])
def test_distant_sensor_render(setup, w_e, w_o):
    """src/sensors/tests/test_distant.py:300-475: path + directional + diffuse + distant + hdrfilm, closed form
    L = E cos(theta_e) rho / pi (x 2/pi without target, x 0.25 for the large square), at the reference's 1e5 spp and tolerances."""
    key = (setup, tuple(w_e), tuple(w_o))
    w_e = list(np.array(w_e) / np.linalg.norm(w_e)); w_o = list(np.array(w_o) / np.linalg.norm(w_o))
    img = ob.OracleScene(_distant_render_scene(setup, w_e, w_o, 100000)).render(threads=1)
    import tests.transport_cases as tc
    rgb = tc.radiance_rgb(img).reshape(3)
    l_o = abs(w_e[2]) / np.pi
    expected = {"default": l_o * 2.0 / np.pi, "target_square_large": l_o * 0.25}.get(setup, l_o)
    rtol = {"target_square_large": 1e-2}.get(setup, 5e-3)                      # test_distant.py:471-474
    if key in DISTANT_MARGINAL:
        assert np.allclose(rgb / expected - 1.0, [5.056e-3, -3.224e-3, -0.961e-3], atol=2e-5), rgb / expected - 1.0
    else:
        assert np.allclose(rgb, expected, rtol=rtol), rgb / expected - 1.0


def test_constant_sum_bias_is_not_an_ulp_effect():
    """The +0.5056 % of the marginal cases above follows from fp32 accumulation alone and does not move when the per-sample
    value moves by an ulp in any channel."""
    w_e = list(np.array([0, 1, -1]) / np.sqrt(2.0))
    v = ob.OracleScene(_distant_render_scene("target_point", w_e, [0, 0, 1], 1)).render(threads=1).reshape(5)[:3]
    assert np.allclose(v, np.array([0.950456, 1.0, 1.088754]) * abs(w_e[2]) / np.pi, rtol=1e-6)     # srgb_to_xyz of a grey L = cos / pi
    m = np.array([[3.240479, -1.537150, -0.498535], [-0.969256, 1.875991, 0.041556], [0.055648, -0.204043, 1.057311]], np.float32)
    cand = np.array([[np.nextafter(np.float32(c), np.float32(-np.inf)), np.float32(c), np.nextafter(np.float32(c), np.float32(np.inf))] for c in v], np.float32)
    sums = np.zeros_like(cand)
    for _ in range(100000):                                                    # the nine running sums side by side, fp32 throughout
        sums = sums + cand
    assert sums.dtype == np.float32
    out = set()
    for ix in range(3):
        for iy in range(3):
            for iz in range(3):
                mean = np.array([sums[0, ix], sums[1, iy], sums[2, iz]], np.float32) / np.float32(100000)
                out.add(tuple(np.round((m @ mean) / (abs(w_e[2]) / np.pi) - 1.0, 6)))
    assert len(out) == 1 and np.allclose(list(out)[0], [5.056e-3, -3.224e-3, -0.961e-3], atol=2e-5), out


# ---------------------------------------------------------------- mradiancemeter / mdistant
def _mrad_sensor(orig, dirs, pixels):
    return {"type": "mradiancemeter", "origins": orig, "directions": dirs,
            "film": {"type": "hdrfilm", "width": pixels, "height": 1, "rfilter": {"type": "box"}}}


def _with_sensor(sensor, **extra):
    d = {"type": "scene", "integrator": {"type": "path"}, "sensor": sensor}
    d.update(extra)
    return d


def test_mradiancemeter_construct_and_rays():
    """src/sensors/tests/test_mradiancemeter.py:57-105: instantiation errors and the sub-sensor picked by position_sample.x"""
    for origins, directions in [(["0, 0, 0"], ["1, 0, 0"]), (["0, 0, 0"] * 2, ["1, 0, 0", "-1, 0, 0"])]:
        ob.OracleScene(_with_sensor(_mrad_sensor(", ".join(origins), ", ".join(directions), len(origins))))
    for bad in [("0, 0, 0", "1, 0, 0", 2), ("0, 0", "1, 0", 1), ("0, 0, 0", "1, 0", 1)]:
        with pytest.raises(RuntimeError):
            ob.OracleScene(_with_sensor(_mrad_sensor(*bad)))
    o = ob.OracleScene(_with_sensor(_mrad_sensor("0, 0, 0, 1, 0, 1", "1, 0, 0, 1, 1, 1", 2)))
    import random
    random.seed(42)
    for _ in range(10):
        random.random()
        ps = (random.random(), random.random())
        ro, rd, w = o.sensor_sample_ray([ps], [(0, 0)])
        if ps[0] < 0.5:
            assert np.allclose(ro[0], (0, 0, 0)) and np.allclose(rd[0], (1, 0, 0), atol=1e-6)
        else:
            assert np.allclose(ro[0], (1, 0, 1)) and np.allclose(rd[0], np.ones(3) / np.sqrt(3.0), atol=1e-6)
        assert np.allclose(w[0], 1.0)


@pytest.mark.parametrize("radiance", [10.0 ** x for x in range(-3, 4)])
def test_mradiancemeter_render_constant(radiance):
    """test_mradiancemeter.py:150-163: three radiancemeters inside a constant environment see its radiance"""
    sensor = _mrad_sensor("1, 0, 0, 0, 1, 0, 0, 0, 1", "1, 0, 0, 0, -1, 0, 0, 0, -1", 3)
    sensor["sampler"] = {"type": "independent", "sample_count": 1}
    d = _with_sensor(sensor, emitter={"type": "constant", "radiance": {"type": "uniform", "value": radiance}})
    img = ob.OracleScene(d).render(threads=1)
    import tests.transport_cases as tc
    assert np.allclose(tc.radiance_rgb(img), radiance, rtol=1e-5)


def test_mradiancemeter_render_complex():
    """test_mradiancemeter.py:166-240: three sub-sensors looking at three surfaces with different reflectances under a
    constant environment of radiance 1 see a radiance equal to the reflectance"""
    sensor = {"type": "mradiancemeter", "origins": "-2, 0, 1, 0, 0, 1, 2, 0, 1", "directions": "0, 0, -1, 0, 0, -1 0, 0, -1",
              "film": {"type": "hdrfilm", "width": 3, "height": 1, "pixel_format": "luminance", "rfilter": {"type": "box"}},
              "sampler": {"type": "independent", "sample_count": 20000}}
    d = _with_sensor(sensor, emitter={"type": "constant", "radiance": {"type": "uniform", "value": 1.0}})
    for k, (x, rho) in enumerate([(-2, 0.0), (0, 0.5), (2, 1.0)]):
        d["rect%d" % k] = {"type": "rectangle", "to_world": T.translate([x, 0, 0]) @ T.scale(0.5),
                           "bsdf": {"type": "diffuse", "reflectance": {"type": "uniform", "value": rho}}}
    img = ob.OracleScene(d).render(threads=1)
    import tests.transport_cases as tc
    rgb = tc.radiance_rgb(img).reshape(3, 3)
    assert np.allclose(rgb[:, 0], [0.0, 0.5, 1.0], atol=1e-2)


def _mdist_sensor(target=None, directions="0, 0, 1", pixel_count=1, spp=None):
    s = {"type": "mdistant", "directions": directions,
         "film": {"type": "hdrfilm", "width": pixel_count, "height": 1, "rfilter": {"type": "box"}}}
    if target == "point":
        s["target"] = [0, 0, 0]
    elif target == "shape":
        s["target"] = {"type": "rectangle"}
    elif isinstance(target, dict):
        s["target"] = target
    if spp:
        s["sampler"] = {"type": "independent", "sample_count": spp}
    return s


def test_mdistant_construct_and_directions():
    """src/sensors/tests/test_mdistant.py:37-132"""
    directions = ",".join(str(x) for x in [0, 0, 1, 0, 0, -1])
    unit = {"shape": {"type": "rectangle"}}
    ob.OracleScene(_with_sensor(_mdist_sensor(directions=directions, pixel_count=2), **unit))
    with pytest.raises(RuntimeError):
        ob.OracleScene(_with_sensor({"type": "mdistant"}, **unit))
    with pytest.raises(RuntimeError):
        ob.OracleScene(_with_sensor({"type": "mdistant", "directions": directions, "film": {"type": "hdrfilm", "width": 2, "height": 2}}, **unit))
    for target in (None, "point", "shape"):
        ob.OracleScene(_with_sensor(_mdist_sensor(target, directions, 2), **unit))
    with pytest.raises(RuntimeError):
        ob.OracleScene(_with_sensor(_mdist_sensor({"type": "constant"}, directions, 2), **unit))
    o = ob.OracleScene(_with_sensor(_mdist_sensor("point", "0, 0, -1, -1, -1, 0, -2, 0, 0", 3), **unit))
    for s1, s2, expected in [[[0.32, 0.87], [0.16, 0.44], [0, 0, -1]], [[0.17, 0.44], [0.22, 0.81], [0, 0, -1]],
                             [[0.51, 0.82], [0.99, 0.42], [-1, -1, 0]], [[0.72, 0.40], [0.01, 0.61], [-2, 0, 0]]]:
        _, rd, _ = o.sensor_sample_ray([s1], [s2])
        e = np.array(expected, dtype=np.float64)
        assert np.allclose(rd[0], e / np.linalg.norm(e), atol=1e-6)


@pytest.mark.parametrize("setup", ["default", "target_square", "target_square_small", "target_square_large", "target_point"])
@pytest.mark.parametrize("w_e", [[0, 0, -1], [0, 1, -1]])
@pytest.mark.parametrize("w_o", [[0, 0, -1], [0, -1, -1]])
def test_mdistant_render_targets(setup, w_e, w_o):
    """test_mdistant.py:135-296: path + directional + diffuse under mdistant; L = E cos(theta_e) rho / pi, times (2/pi) cos(theta_o)
    without target, times 0.25 for a target twice the surface's size.  20000 spp instead of 1e5."""
    w_e = list(np.array(w_e) / np.linalg.norm(w_e)); w_o = list(np.array(w_o) / np.linalg.norm(w_o))
    cos_e, cos_o = abs(w_e[2]), abs(w_o[2])
    target = {"default": None, "target_point": "point"}.get(setup, setup)
    if isinstance(target, str) and target.startswith("target_square"):
        scale = {"target_square": 1.0, "target_square_small": 0.5, "target_square_large": 2.0}[setup]
        target = {"type": "rectangle", "to_world": T.scale(scale)}
    sensor = _mdist_sensor(target, ",".join(map(str, w_o)), 1, spp=20000)
    d = _with_sensor(sensor, shape={"type": "rectangle", "to_world": T.scale(1.0), "bsdf": {"type": "diffuse", "reflectance": 1.0}},
                     emitter={"type": "directional", "direction": w_e, "irradiance": 1.0})
    img = ob.OracleScene(d).render(threads=1)
    import tests.transport_cases as tc
    rgb = tc.radiance_rgb(img).reshape(3)
    l_o = cos_e / np.pi
    expected = {"default": l_o * (2.0 / np.pi) * cos_o, "target_square_large": l_o * 0.25}.get(setup, l_o)
    assert np.allclose(rgb, expected, rtol={"target_square_large": 2e-2, "default": 2e-2}.get(setup, 5e-3))


# ---------------------------------------------------------------- distantflux
def _flux_sensor(target=None, to_world=None, film="1x1", spp=None):
    w, h = {"1x1": (1, 1), "16x16": (16, 16), "32x32": (32, 32)}[film]
    s = {"type": "distantflux", "film": {"type": "hdrfilm", "width": w, "height": h, "rfilter": {"type": "box"}}}
    if to_world is not None:
        s["to_world"] = to_world
    if target == "point":
        s["target"] = [0, 0, 0]
    elif target == "shape":
        s["target"] = {"type": "rectangle"}
    elif target is not None:
        s["target"] = target
    if spp:
        s["sampler"] = {"type": "independent", "sample_count": spp}
    return s


def test_distantflux_construct_and_rays():
    """src/sensors/tests/test_distantflux.py:58-106,109-132,184-199: construction, origins outside the bounding sphere, the
    literal ray directions of the hemisphere warp"""
    unit = {"shape": {"type": "rectangle"}}
    for sd in (_flux_sensor(), _flux_sensor(to_world=T.look_at([0, 0, 0], [0, 0, 1], [1, 0, 0])), _flux_sensor("point"), _flux_sensor("shape")):
        ob.OracleScene(_with_sensor(sd, **unit))
    with pytest.raises(RuntimeError):
        ob.OracleScene(_with_sensor(_flux_sensor({"type": "constant"}), **unit))
    o = ob.OracleScene(_with_sensor(_flux_sensor(), **unit))
    radius = np.linalg.norm([1.0, 1.0, 0.0])                 # bounding sphere of the unit rectangle
    for s1, s2 in [[[0.32, 0.87], [0.16, 0.44]], [[0.17, 0.44], [0.22, 0.81]], [[0.12, 0.82], [0.99, 0.42]], [[0.72, 0.40], [0.01, 0.61]]]:
        ro, _, _ = o.sensor_sample_ray([s1], [s2])
        assert np.linalg.norm(ro[0]) > radius
    for s1, s2, expected in [[[0.5, 0.5], [0.16, 0.44], [0, 0, -1]], [[0.0, 0.0], [0.23, 0.40], [0.707107, 0.707107, 0]],
                             [[1.0, 0.0], [0.22, 0.81], [-0.707107, 0.707107, 0]], [[0.0, 1.0], [0.99, 0.42], [0.707107, -0.707107, 0]],
                             [[1.0, 1.0], [0.52, 0.31], [-0.707107, -0.707107, 0]]]:
        _, rd, _ = o.sensor_sample_ray([s1], [s2])
        assert np.allclose(rd[0], expected, atol=1e-6)


@pytest.mark.parametrize("setup", ["default", "target_square", "target_square_small", "target_square_large", "target_point"])
def test_distantflux_render_targets(setup):
    """test_distantflux.py:202-369: the film of a distantflux sensor sums to the exitant flux density of a white Lambertian
    square under unit irradiance (x 2/pi without target, x 0.25 for a target twice as large)"""
    target = {"default": None, "target_point": "point"}.get(setup, setup)
    if isinstance(target, str) and target.startswith("target_square"):
        target = {"type": "rectangle", "to_world": T.scale({"target_square": 1.0, "target_square_small": 0.5, "target_square_large": 2.0}[setup])}
    spp = 10000 if setup in ("default", "target_square_large") else 100
    if setup in ("default", "target_square_large"):
        spp = 2000                                           # 1e4 in the reference; tolerance widened accordingly
    d = _with_sensor(_flux_sensor(target, film="16x16", spp=spp),
                     shape={"type": "rectangle", "to_world": T.scale(1.0), "bsdf": {"type": "diffuse", "reflectance": 1.0}},
                     emitter={"type": "directional", "direction": [0, 0, -1], "irradiance": 1.0})
    img = ob.OracleScene(d).render()
    import tests.transport_cases as tc
    total = tc.radiance_rgb(img)[..., 1].sum()
    expected = {"default": 2.0 / np.pi, "target_square_large": 0.25}.get(setup, 1.0)
    assert np.allclose(total, expected, rtol={"default": 1e-2, "target_square_large": 1e-2}.get(setup, 1e-3))


# ---------------------------------------------------------------- bilambertian (Eradiate's leaf BSDF)
@pytest.mark.parametrize("r,t", [(0.2, 0.4), (0.4, 0.2), (0.1, 0.9), (0.9, 0.1), (0.4, 0.6), (0.6, 0.4)])
def test_bilambertian_eval_pdf(r, t):
    """src/bsdfs/tests/test_bilambertian.py:23-63: reflection on the side of wi, transmission on the other, from both sides"""
    d = {"type": "scene", "integrator": {"type": "path"}, "sensor": {"type": "perspective", "film": {"type": "hdrfilm", "width": 1, "height": 1}},
         "leaf": {"type": "rectangle", "bsdf": {"type": "bilambertian", "reflectance": r, "transmittance": t}}}
    o = ob.OracleScene(d)
    albedo = r + t
    for wi in ([0, 0, 1], [0, 0, -1]):
        for i in range(20):
            theta = i / 19.0 * np.pi
            wo = [np.sin(theta), 0, np.cos(theta)]
            v, pdf = o.bsdf_eval(0, wi, wo)
            k = r if np.dot(wi, wo) > 0 else t
            assert np.allclose(v, k * abs(wo[2]) / np.pi, atol=1e-7) and np.isclose(pdf, k / albedo * abs(wo[2]) / np.pi, atol=1e-7)


@pytest.mark.parametrize("r,t", [(0.6, 0.2), (0.2, 0.6), (0.9, 0.1), (1.0, 0.0), (0.0, 1.0), (0.0, 0.0)])
def test_bilambertian_sample_matches_its_pdf(r, t):
    """test_bilambertian.py:66-95 runs a chi^2 test; here: sample() reports the pdf pdf() reports for its direction, the weight is
    value / pdf, and the lobe frequencies follow the sampling weights (both sides)."""
    d = {"type": "scene", "integrator": {"type": "path"}, "sensor": {"type": "perspective", "film": {"type": "hdrfilm", "width": 1, "height": 1}},
         "leaf": {"type": "rectangle", "bsdf": {"type": "bilambertian", "reflectance": r, "transmittance": t}}}
    o = ob.OracleScene(d)
    rng = np.random.default_rng(3)
    for wi in ([0.3, -0.2, 0.93], [0.1, 0.4, -0.91]):
        wi = np.array(wi) / np.linalg.norm(wi)
        refl = 0
        n = 400
        for _ in range(n):
            s1, s2 = rng.random(), rng.random(2)
            wo, pdf, wgt, st = o.bsdf_sample(0, wi, s1, s2)
            if r + t == 0:
                assert np.all(wgt == 0)
                continue
            assert st in (0x2, 0x4) and np.isclose(np.linalg.norm(wo), 1.0, atol=1e-5)
            refl += st == 0x2
            assert (st == 0x2) == (wo[2] * wi[2] > 0)
            v, p2 = o.bsdf_eval(0, wi, wo)
            assert np.isclose(p2, pdf, rtol=1e-5, atol=1e-8)
            if pdf > 0:
                assert np.allclose(wgt, v / pdf, rtol=1e-4)
        if r + t > 0:
            assert abs(refl / n - r / (r + t)) < 0.08


# ---------------------------------------------------------------- radiancemeter
def _radiancemeter(origin=None, direction=None, to_world=None, pixels=1):
    d = {"type": "radiancemeter", "film": {"type": "hdrfilm", "width": pixels, "height": pixels, "rfilter": {"type": "box"}}}
    if origin is not None: d["origin"] = origin
    if direction is not None: d["direction"] = direction
    if to_world is not None: d["to_world"] = to_world
    return d


def test_radiancemeter_construct_and_rays():
    """src/sensors/tests/test_radiancemeter.py:32-111"""
    look = T.look_at([0, 0, 0], [0, 1, 0], [0, 0, 1])
    o = ob.OracleScene(_with_sensor(_radiancemeter(to_world=look)))
    ro, rd, _ = o.sensor_sample_ray([[0.32, 0.87]], [[0.16, 0.44]])
    assert np.allclose(ro[0], 0) and np.allclose(rd[0], [0, 1, 0], atol=1e-6)
    o = ob.OracleScene(_with_sensor(_radiancemeter(to_world=look, origin=[1, 0, 0], direction=[4, 1, 0])))      # to_world wins
    ro, rd, _ = o.sensor_sample_ray([[0.5, 0.5]], [[0.5, 0.5]])
    assert np.allclose(ro[0], 0) and np.allclose(rd[0], [0, 1, 0], atol=1e-6)
    for bad in (dict(direction=[0, 1, 0]), dict(origin=[0, 1, 0]), dict(pixels=2)):
        with pytest.raises(RuntimeError):
            ob.OracleScene(_with_sensor(_radiancemeter(**bad)))
    for direction in ([0.0, 0.0, 1.0], [-1.0, -1.0, 0.0], [2.0, 0.0, 0.0]):
        for origin in ([0.0, 0.0, 0.0], [-1.0, -1.0, 0.5], [4.0, 1.0, 0.0]):
            o = ob.OracleScene(_with_sensor(_radiancemeter(origin, direction)))
            ro, rd, w = o.sensor_sample_ray([[0.32, 0.87]], [[0.16, 0.44]])
            assert np.allclose(ro[0], origin) and np.allclose(rd[0], np.array(direction) / np.linalg.norm(direction), atol=1e-6) and np.allclose(w, 1)


@pytest.mark.parametrize("radiance", [10.0 ** x for x in range(-3, 4)])
def test_radiancemeter_render(radiance):
    """test_radiancemeter.py:114-150: a radiancemeter inside a constant environment measures its radiance"""
    s = _radiancemeter([1, 0, 0], [1, 0, 0]); s["sampler"] = {"type": "independent", "sample_count": 1}
    d = _with_sensor(s, emitter={"type": "constant", "radiance": {"type": "uniform", "value": radiance}})
    import tests.transport_cases as tc
    assert np.allclose(tc.radiance_rgb(ob.OracleScene(d).render(threads=1)), radiance, rtol=1e-5)


# ---------------------------------------------------------------- disk shape
def test_disk_hits_and_targets():
    """src/shapes/tests/test_disk.py:37-65: rays along +z through a grid of points hit iff x^2 + y^2 <= r^2; the hit lies on the
    disk's plane with the disk's normal.  Also: a disk as sensor target (test_mdistant.py "target_disk")."""
    for r in (1, 3, 5):
        for translate in ([0.0, 0.0, 0.0], [1.0, -5.0, 0.0]):
            d = {"type": "scene", "integrator": {"type": "path"},
                 "sensor": {"type": "perspective", "film": {"type": "hdrfilm", "width": 4, "height": 4, "rfilter": {"type": "box"}}},
                 "foo": {"type": "disk", "to_world": T.translate(translate) @ T.scale([r, r, 1.0])}}
            o = ob.OracleScene(d)
            g = np.linspace(-1, 1, 10, dtype=np.float32)
            xs = (1.1 * r * (g[:, None] - translate[0]) + 0 * g[None, :]).reshape(-1)
            ys = (1.1 * r * (g[None, :] - translate[1]) + 0 * g[:, None]).reshape(-1)
            orig = np.stack([xs, ys, np.full(xs.size, -10.0)], 1).astype(np.float32)
            dirs = np.tile(np.array([0, 0, 1], np.float32), (xs.size, 1))
            res = o.ray_intersect(orig, dirs, mint=np.zeros(xs.size, np.float32))
            found = np.isfinite(res["t"])
            lx, ly = (xs - translate[0]) / r, (ys - translate[1]) / r
            expect = lx.astype(np.float32) ** 2 + ly.astype(np.float32) ** 2 <= 1
            margin = np.abs(lx ** 2 + ly ** 2 - 1) > 1e-4
            assert np.array_equal(found[margin], expect[margin]) and (~found).any()
            assert found.any() == (translate[1] == 0.0)      # the reference's second grid (test_disk.py:52-53) misses the translated disk entirely
            assert np.allclose(res["t"][found], 10) and np.allclose(res["n"][found], [0, 0, 1])
    # surface area pi sx sy (test_disk.py:7-34) through the area-sampling pdf of an emitter on a disk
    sensor = _mdist_sensor({"type": "disk", "to_world": T.scale(1.0)}, "0, 0, -1", 1, spp=20000)
    d = _with_sensor(sensor, shape={"type": "rectangle", "to_world": T.scale(1.0), "bsdf": {"type": "diffuse", "reflectance": 1.0}},
                     emitter={"type": "directional", "direction": [0, 0, -1], "irradiance": 1.0})
    import tests.transport_cases as tc
    rgb = tc.radiance_rgb(ob.OracleScene(d).render(threads=1)).reshape(3)
    assert np.allclose(rgb, 1.0 / np.pi, rtol=5e-3)        # test_mdistant.py:135-296, "target_disk": every ray lands on the square
