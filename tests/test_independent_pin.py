"""An independent pin for `volpath`.

The reference tree holds no numeric vector for volpath, media or grid volumes (src/media/tests/ is empty, the Z-test
images of src/librender/tests/test_renders.py live in an absent submodule), and GPU <-> oracle bit-equality is common
mode.  tests/independent/walk.py therefore computes the radiance of the C3 / C4 miniatures with a structurally different
estimator (float64 numpy, ray-marched optical depth inverted for the free paths, quadrature transmittance, collision
estimator with the solar beam as explicit source, Philox random numbers; see its docstring).  Its per-pixel means and
variances are committed as tests/golden/indep_pin_*.npz (tests/independent/make_pin.py).

Here: (1) the estimator is itself checked against a closed form and against its fixtures; (2) the oracle is Z-tested
against the fixtures per pixel, with the Sidak correction the reference's own render tests use
(src/librender/tests/test_renders.py:63-137, significance 0.01), and on the image mean, whose combined standard error
is below 0.25 % -- so a misreading of volpath.cpp that moves the image by 1 % is four standard errors away;
(3) the pin has teeth: it rejects the C4 scene as it was built in round 1 (ground just below the medium's boundary),
where a light leak of the reference's own ray-epsilon rule brightened slanted views by up to 20 %."""
import ast
import copy
import importlib
import math
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import tests.oracle_binding as ob
from tests.independent import problems, walk

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
scenes = importlib.import_module("eradiate-kernel_amd.scenes")
SIGNIFICANCE = 0.01


def load_pin(name):
    z = np.load(os.path.join(GOLDEN, "indep_pin_%s_16x16.npz" % name))
    return z["mean"], z["var"], int(z["per_pixel"])


def oracle_estimate(d, seeds=16, spp=256):
    """Per-pixel mean luminance and variance of that mean from `seeds` independent oracle renders."""
    def one(seed):
        dd = copy.deepcopy(d)
        dd["sensor"]["sampler"]["sample_count"] = spp
        dd["sensor"]["sampler"]["seed"] = seed
        img = ob.OracleScene(dd).render(threads=1)
        return img[..., 1] / img[..., 4]                      # Y / W: the luminance of a grey radiance
    with ThreadPoolExecutor(min(8, os.cpu_count() or 1)) as ex:
        imgs = np.array(list(ex.map(one, range(seeds))), np.float64)
    return imgs.mean(0), imgs.var(0, ddof=1) / seeds


def z_test(mean_a, var_a, mean_b, var_b):
    """Two-sample version of test_renders.py:63-80: p-value per pixel; accepted when > the Sidak-corrected level."""
    z = np.abs(mean_a - mean_b) / np.sqrt(var_a + var_b)
    p = 2.0 * (1.0 - 0.5 * (1.0 + np.vectorize(math.erf)(z / math.sqrt(2.0))))
    alpha = 1.0 - (1.0 - SIGNIFICANCE) ** (1.0 / z.size)
    return p, alpha, z


def test_estimator_imports_nothing_from_the_product_or_the_oracle():
    src = open(walk.__file__).read()
    mods = set()
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.Import):
            mods |= {a.name.split(".")[0] for a in node.names}
        elif isinstance(node, ast.ImportFrom):
            mods.add((node.module or "").split(".")[0])
    assert mods == {"math", "numpy"}, mods
    assert "ctypes" not in src and "oracle" not in src.replace("oracle/", "") .replace("the oracle", "")


def test_estimator_reproduces_the_single_scattering_closed_form():
    """Homogeneous slab, black ground, one event: L = albedo p(theta) E mu0 / (mu + mu0) (1 - exp(-tau (1/mu + 1/mu0)))."""
    sig, alb, g, h = 0.7, 0.9, 0.5, 2.0
    mu0 = 0.8
    sun = [math.sqrt(1 - mu0 * mu0), 0.0, -mu0]
    prob = walk.SlabProblem([-1e3, -1e3, 0], [1e3, 1e3, h], np.full((4, 2, 2), sig), alb, ("hg", g), -0.01, 2e3, ("diffuse", 0.0), sun)
    rng = np.random.Generator(np.random.Philox(3))
    for mu, phi in [(1.0, 0.0), (0.5, 0.3), (0.25, 2.0)]:
        v = np.array([math.sqrt(1 - mu * mu) * math.cos(phi), math.sqrt(1 - mu * mu) * math.sin(phi), mu])
        n = 40000
        o = np.tile(np.array([0.0, 0.0, h]) + 5.0 * v, (n, 1)); d = np.tile(-v, (n, 1))
        val = walk.radiance(prob, rng, o, d, max_events=1)
        cos_sc = float(np.dot(-v, -np.asarray(sun)))          # sunlight travelling along `sun` is scattered into +v
        cos_sc = float(np.dot(np.asarray(sun), v))
        p = (1 - g * g) / (4 * math.pi * (1 + g * g - 2 * g * cos_sc) ** 1.5)
        tau = sig * h
        expected = alb * p * mu0 / (mu + mu0) * (1 - math.exp(-tau * (1 / mu + 1 / mu0)))
        se = val.std() / math.sqrt(n)
        assert abs(val.mean() - expected) < 4 * se + 2e-4 * expected, (mu, val.mean(), expected, se)


@pytest.mark.parametrize("name", ["c3", "c4", "c4x3"])
def test_fixture_comes_from_the_committed_estimator(name):
    """A short live run of the estimator agrees with its fixture (same code, other seed, 40 walks per pixel)."""
    mean, var, _ = load_pin(name)
    _, prob, sensor = getattr(problems, name)()
    m, v = walk.render(prob, sensor, 16, 16, 40, seed=77)
    se = math.sqrt(v.sum() + var.sum()) / m.size
    assert abs(m.mean() - mean.mean()) < 4 * se
    # 40 walks do not give a usable per-pixel variance: compare 4 x 4 tiles of 16 pixels instead
    tm, tv = m.reshape(4, 4, 4, 4).mean((1, 3)), v.reshape(4, 4, 4, 4).sum((1, 3)) / 256
    fm, fv = mean.reshape(4, 4, 4, 4).mean((1, 3)), var.reshape(4, 4, 4, 4).sum((1, 3)) / 256
    p, alpha, _ = z_test(tm, tv, fm, fv)
    assert (p > alpha).all()


@pytest.mark.parametrize("integrator", ["volpath", "volpathmis"])
@pytest.mark.parametrize("name", ["c3", "c4", "c4x3"])
def test_oracle_agrees_with_the_independent_estimator(name, integrator):
    """`volpath`, and `volpathmis` (src/integrators/volpathmis.cpp: another estimator of the same radiance, for which the reference
    holds no vector either), against the same fixtures.  c4x3 (round 3) = the atmosphere with a third species, i.e. a blendphase nested
    in a blendphase (src/phase/blendphase.cpp:42-66, 92-108), restated by the estimator as a three-way mixture."""
    mean, var, _ = load_pin(name)
    d, _, _ = getattr(problems, name)()
    d["integrator"] = dict(d["integrator"], type=integrator)
    om, ov = oracle_estimate(d)
    se_pin = math.sqrt(var.sum()) / var.size / mean.mean()
    se_orc = math.sqrt(ov.sum()) / ov.size / om.mean()
    se = math.hypot(se_pin, se_orc)
    rel = om.mean() / mean.mean() - 1.0
    print("%s: image mean oracle %.6f, independent %.6f, difference %+.3f %% (standard errors %.3f %% / %.3f %%)"
          % (name, om.mean(), mean.mean(), 100 * rel, 100 * se_orc, 100 * se_pin))
    assert se < 2.5e-3                                        # the sensitivity this pin claims
    assert abs(rel) < 4 * se and abs(rel) < 1e-2
    p, alpha, z = z_test(om, ov, mean, var)
    print("%s: min p-value %.2e, Sidak level %.2e, max |z| %.2f" % (name, p.min(), alpha, z.max()))
    assert (p > alpha).mean() >= 0.9975                       # test_renders.py:121


def test_pin_rejects_the_round_1_c4_scene():
    """Round 1 put the C4 ground 0.01 below the medium's bottom face.  A ray leaving a surface point p starts at
    (1 + max|p|) RayEpsilon (interaction.h:58-61): beyond that face for |p| > 130, so the reflected ray never entered the
    medium and the ground was lit by an unattenuated sun.  The oracle (and the HIP kernels, bit for bit) reproduce that rule
    faithfully; the independent estimator, which knows nothing about ray epsilons, does not -- and the Z-test says so."""
    mean, var, _ = load_pin("c4")
    d, _, _ = problems.c4()
    d["ground"]["to_world"] = scenes.T.translate([0, 0, -0.01]) @ scenes.T.scale(1.2e4)
    om, ov = oracle_estimate(d, seeds=8)
    p, alpha, z = z_test(om, ov, mean, var)
    assert (p > alpha).mean() < 0.9 and z.max() > 5.0
    assert om.mean() / mean.mean() - 1.0 < -1.5e-2


@pytest.mark.parametrize("integrator", ["volpath", "volpathmis", "volpathmis_no_spectral_mis"])
def test_spectral_oracle_agrees_with_the_independent_estimator(integrator):
    """The spectral variant on the grey C3 miniature (under `volpath`, and under `volpathmis` with and without spectral MIS: the 4 x 4
    WeightMatrix of volpathmis.cpp:66-69): every sample draws four wavelengths, the film holds hmean(cmf(lambda) L(lambda)) x
    470 nm (core/spectrum.h:210-217, 250-254), so Y / W estimates L x the integral of the piecewise-linear y-bar table over 360 ..
    830 nm (106.857 for the 5 nm table of libcore/spectrum.cpp).  Divided by that constant the spectral render must agree with the
    independent estimator like the rgb one does -- wavelength sampling, CIE weighting and the four-wide transport included."""
    import re
    mean, var, _ = load_pin("c3")
    d, _, _ = problems.c3()

    def grey(node):                                           # grey rgb colours -> uniform spectra (the variant has no sRGB upsampling model)
        if isinstance(node, dict):
            if node.get("type") == "rgb":
                v = np.atleast_1d(np.asarray(node["value"], np.float64))
                assert np.all(v == v[0])
                return float(v[0])
            return {k: grey(x) for k, x in node.items()}
        return node
    d = grey(d)
    if integrator != "volpath":
        d["integrator"] = dict(d["integrator"], type="volpathmis", use_spectral_mis=integrator == "volpathmis")
    tbl = np.array([float(x) for x in re.findall(r"([0-9.eE+-]+)f", open(os.path.join(os.path.dirname(GOLDEN), "..", "eradiate-kernel_amd", "csrc", "cie_tables.h")).read().split("{")[1])], np.float64).reshape(3, 95)
    y_integral = float(((tbl[1][:-1] + tbl[1][1:]) * 0.5 * 5.0).sum())
    assert abs(y_integral - 106.857) < 1e-2

    def one(seed):
        dd = copy.deepcopy(d)
        dd["sensor"]["sampler"]["sample_count"] = 512
        dd["sensor"]["sampler"]["seed"] = seed
        img = ob.OracleScene(dd, spectral=True).render(threads=1)
        return img[..., 1] / img[..., 4] / y_integral
    with ThreadPoolExecutor(min(8, os.cpu_count() or 1)) as ex:
        imgs = np.array(list(ex.map(one, range(16))), np.float64)
    om, ov = imgs.mean(0), imgs.var(0, ddof=1) / len(imgs)
    se = math.hypot(math.sqrt(var.sum()) / var.size / mean.mean(), math.sqrt(ov.sum()) / ov.size / om.mean())
    rel = om.mean() / mean.mean() - 1.0
    print("c3 spectral " + integrator + ": image mean %.6f, independent %.6f, difference %+.3f %% +- %.3f %%" % (om.mean(), mean.mean(), 100 * rel, 100 * se))
    assert se < 6e-3 and abs(rel) < 4 * se and abs(rel) < 1.5e-2
    p, alpha, z = z_test(om, ov, mean, var)
    assert (p > alpha).mean() >= 0.9975


XYZ_TO_RGB = np.array([[3.240479, -1.537150, -0.498535], [-0.969256, 1.875991, 0.041556], [0.055648, -0.204043, 1.057311]])   # core/spectrum.h:229-235


def chroma_estimate(render, seeds=16, spp=256):
    """Per-pixel mean and variance of the mean of the linear RGB radiance from `seeds` renders (render(dict) -> XYZAW film)."""
    def one(seed):
        dd, _, _ = problems.c2_chroma(0)
        dd["sensor"]["sampler"]["sample_count"] = spp
        dd["sensor"]["sampler"]["seed"] = seed
        return dd
    imgs = []
    for film in render([one(s) for s in range(seeds)]):
        imgs.append((film[..., :3] / film[..., 4:5]) @ XYZ_TO_RGB.T)
    imgs = np.array(imgs, np.float64)
    return imgs.mean(0), imgs.var(0, ddof=1) / seeds


def check_chroma(mean_rgb, var_rgb, label, strict):
    worst = 0.0
    for c, ch in enumerate("rgb"):
        mean, var, _ = load_pin("chroma_" + ch)
        se = math.hypot(math.sqrt(var.sum()) / var.size / mean.mean(), math.sqrt(var_rgb[..., c].sum()) / var.size / mean_rgb[..., c].mean())
        rel = mean_rgb[..., c].mean() / mean.mean() - 1.0
        p, alpha, z = z_test(mean_rgb[..., c], var_rgb[..., c], mean, var)
        print("%s channel %s: image mean %.6f, independent %.6f, difference %+.3f %% +- %.3f %%, pixels accepted %.4f, max |z| %.2f"
              % (label, ch, mean_rgb[..., c].mean(), mean.mean(), 100 * rel, 100 * se, (p > alpha).mean(), z.max()))
        worst = max(worst, abs(rel) / se)
        if strict:
            assert se < 4e-3 and abs(rel) < 4 * se and abs(rel) < 1e-2, (label, ch, rel, se)
            assert (p > alpha).mean() >= 0.9975
    return worst


@pytest.mark.parametrize("integrator", ["volpathmis", "volpathmis_no_spectral_mis", "volpath"])
def test_chromatic_medium_against_the_independent_estimator(integrator):
    """A chromatic homogeneous slab (sigma_t 0.4 / 1.0 / 1.6, albedo 0.9 / 0.7 / 0.5 per channel) -- the case spectral MIS exists for.
    The estimator runs once per channel with that channel's coefficients (fixtures indep_pin_chroma_{r,g,b}); the integrators follow a
    hero channel and reweight the others.  `volpathmis` with spectral MIS must agree per channel like a grey scene does.  The
    single-channel estimators (`volpath`, `volpathmis` without spectral MIS) are unbiased too, but their weights e^{(sigma_hero -
    sigma_c) t} are heavy-tailed: the sample mean sits below the expectation with high probability and its empirical standard error
    understates the spread (measured here: red, the thinnest channel, -1.1 % / -1.6 % at 16 x 256 spp; green and blue within 0.7 %) --
    they are held to 3 % on the image mean, not to the Z-test."""
    def render(dicts):
        def one(dd):
            dd["integrator"]["type"] = "volpath" if integrator == "volpath" else "volpathmis"
            if integrator != "volpath":
                dd["integrator"]["use_spectral_mis"] = integrator == "volpathmis"
            return ob.OracleScene(dd).render(threads=1)
        with ThreadPoolExecutor(min(8, os.cpu_count() or 1)) as ex:
            return list(ex.map(one, dicts))
    mean_rgb, var_rgb = chroma_estimate(render)
    strict = integrator == "volpathmis"
    check_chroma(mean_rgb, var_rgb, integrator, strict)
    if not strict:
        for c, ch in enumerate("rgb"):
            mean, _, _ = load_pin("chroma_" + ch)
            assert abs(mean_rgb[..., c].mean() / mean.mean() - 1.0) < 3e-2
