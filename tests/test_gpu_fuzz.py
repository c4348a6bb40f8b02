"""Differential fuzzing of the HIP path against the oracle: random scenes assembled from every supported plugin (shapes, BSDFs,
media, phase functions, emitters, sensors, integrators), small films, bit-for-bit comparison of film and loop counters."""
import importlib

import numpy as np
import pytest

import tests.oracle_binding as ob

pytestmark = pytest.mark.gpu
T = importlib.import_module("eradiate-kernel_amd.transform").ScalarTransform4f
scenes = importlib.import_module("eradiate-kernel_amd.scenes")


def _bsdf(rng):
    k = rng.integers(0, 4)
    if k == 0:
        return {"type": "diffuse", "reflectance": {"type": "rgb", "value": rng.uniform(0.05, 0.95, 3).tolist()}}
    if k == 1:
        return {"type": "bilambertian", "reflectance": float(rng.uniform(0, 0.6)), "transmittance": float(rng.uniform(0, 0.4))}
    if k == 2:
        return {"type": "rpv", "rho_0": float(rng.uniform(0.05, 0.3)), "k": float(rng.uniform(0.4, 0.9)), "g": float(rng.uniform(-0.3, 0.1))}
    return {"type": "diffuse", "reflectance": float(rng.uniform(0.1, 0.9))}


def _phase(rng, depth=0):
    if depth < 2 and rng.random() < 0.25:            # a blendphase holding blendphases (blendphase.cpp:42-66), up to depth 2
        return {"type": "blendphase", "phase_0": _phase(rng, depth + 1), "phase_1": _phase(rng, depth + 1), "weight": float(rng.uniform(0.1, 0.9))}
    k = rng.integers(0, 5)
    if k == 0: return {"type": "isotropic"}
    if k == 1: return {"type": "hg", "g": float(rng.uniform(-0.8, 0.9))}
    if k == 2: return {"type": "rayleigh"}
    if k == 3: return {"type": "tabphase", "values": scenes.hg_table(float(rng.uniform(0.1, 0.8)), 61)}
    return {"type": "blendphase", "phase_0": {"type": "rayleigh"}, "phase_1": {"type": "hg", "g": 0.6}, "weight": float(rng.uniform(0.1, 0.9))}


def _medium(rng):
    if rng.random() < 0.5:
        chroma = rng.random() < 0.4
        st = rng.uniform(0.2, 2.0, 3) if chroma else np.full(3, rng.uniform(0.2, 2.0))
        return {"type": "homogeneous", "sigma_t": {"type": "rgb", "value": st.tolist()}, "albedo": float(rng.uniform(0.3, 0.95)), "phase": _phase(rng)}
    res = int(rng.choice([4, 8, 12]))
    grid_xf = T.translate([-3, -3, 0]) @ T.scale([6, 6, 2])
    sig = rng.uniform(0.1, 2.5, (res, res, res)).astype(np.float32)
    alb = rng.uniform(0.3, 0.95, (res, res, res)).astype(np.float32) if rng.random() < 0.7 else np.full((res, res, res), 0.8, np.float32)
    m = {"type": "heterogeneous", "sigma_t": {"type": "gridvolume", "data": sig, "to_world": grid_xf},
         "albedo": {"type": "gridvolume", "data": alb, "to_world": grid_xf}, "phase": _phase(rng), "scale": float(rng.uniform(0.5, 1.5))}
    if rng.random() < 0.3:
        m["sigma_t"]["filter_type"] = "nearest"; m["albedo"]["filter_type"] = "nearest"
    return m


def _scene(seed):
    rng = np.random.default_rng(seed)
    integrator = str(rng.choice(["volpath", "volpath", "volpathmis", "path"]))
    d = {"type": "scene", "integrator": {"type": integrator, "max_depth": int(rng.choice([-1, 3, 8])), "rr_depth": int(rng.choice([2, 5])), "block_size": 32}}
    if integrator == "volpathmis":
        d["integrator"]["use_spectral_mis"] = bool(rng.random() < 0.6)
    w, h, spp = int(rng.integers(5, 40)), int(rng.integers(4, 36)), int(rng.choice([2, 4, 6]))
    film = {"type": "hdrfilm", "width": w, "height": h, "rfilter": {"type": "box"} if rng.random() < 0.8 else {"type": "gaussian"}}
    st = rng.integers(0, 4)
    if st == 0:
        d["sensor"] = {"type": "perspective", "to_world": T.look_at([rng.uniform(-1, 1), -9, rng.uniform(3, 7)], [0, 0, 1], [0, 0, 1]), "fov": 45, "film": film}
    elif st == 1:
        d["sensor"] = {"type": "distant", "direction": [0.2, -0.1, -1], "film": film}
        if rng.random() < 0.5: d["sensor"]["ray_target"] = {"type": "rectangle", "to_world": T.translate([0, 0, 2.1]) @ T.scale(3.0)}
    elif st == 2:
        n = int(rng.integers(2, 6)); film["width"], film["height"] = n, 1
        dirs = ", ".join("%g, %g, -1" % (rng.uniform(-.5, .5), rng.uniform(-.5, .5)) for _ in range(n))
        d["sensor"] = {"type": "mdistant", "directions": dirs, "film": film, "target": [0.0, 0.0, 1.0]}
    else:
        film["width"], film["height"] = 6, 5
        d["sensor"] = {"type": "distantflux", "film": film, "target": {"type": "disk", "to_world": T.translate([0, 0, 2.1]) @ T.scale(2.5)}}
    d["sensor"]["sampler"] = {"type": "independent", "sample_count": spp, "seed": int(rng.integers(0, 1000))}
    d["ground"] = {"type": "rectangle", "to_world": T.translate([0, 0, -0.01]) @ T.scale(8.0), "bsdf": _bsdf(rng)}
    if integrator != "path":
        d["slab"] = {"type": "cube", "to_world": T.translate([0, 0, 1]) @ T.scale([3, 3, 1]), "bsdf": {"type": "null"}, "interior": _medium(rng)}
    for k in range(int(rng.integers(0, 9))):
        c = rng.uniform([-2.5, -2.5, 2.2], [2.5, 2.5, 4.5])
        xf = T.translate(c) @ T.rotate(rng.normal(size=3), float(rng.uniform(0, 180))) @ T.scale(rng.uniform(0.2, 0.7, 3))
        kind = str(rng.choice(["rectangle", "disk", "cube", "sphere"]))
        shape = {"type": kind, "bsdf": _bsdf(rng)}
        if kind == "sphere":
            shape.update(center=c.tolist(), radius=float(rng.uniform(0.2, 0.6)))
        else:
            shape["to_world"] = xf
        d["obj%d" % k] = shape
    et = rng.integers(0, 4)
    if et == 0:
        d["sun"] = {"type": "directional", "direction": [float(rng.uniform(-.5, .5)), float(rng.uniform(-.5, .5)), -1.0], "irradiance": 2.0}
    elif et == 1:
        d["sky"] = {"type": "constant", "radiance": 0.7}
        d["sun"] = {"type": "directional", "direction": [0.1, 0.3, -1.0], "irradiance": 1.0}
    elif et == 3:
        d["bulb"] = {"type": "point", "position": [float(rng.uniform(-2, 2)), float(rng.uniform(-2, 2)), 5.5], "intensity": {"type": "rgb", "value": [30.0, 25.0, 20.0]}}
        if rng.random() < 0.5: d["sky"] = {"type": "constant", "radiance": 0.2}
    else:
        d["lamp"] = {"type": str(rng.choice(["rectangle", "disk", "cube", "sphere"])), "to_world": T.translate([0.5, 0, 6]) @ T.rotate([1, 0, 0], 180) @ T.scale(1.5),
                     "emitter": {"type": "area", "radiance": {"type": "rgb", "value": [4.0, 3.5, 3.0]}}}
    if rng.random() < 0.2:                               # the streams of the gpu_* variants: one per (pixel, sample) (drawn last: the scenes of round 2 keep their layout)
        d["sensor"]["sampler"]["wavefront"] = True
    return d


@pytest.mark.parametrize("seed", range(48))
@pytest.mark.parametrize("bvh", [False, True])
def test_random_scene(gpu_rgb, monkeypatch, seed, bvh):
    if bvh:
        monkeypatch.setenv("MTSAMD_BVH_THRESHOLD", "0")
    d = _scene(seed)
    scene = gpu_rgb.load_dict(d)
    sensor = scene.sensors()[0]
    assert scene.integrator().render(scene, sensor, collect_counters=True)
    gpu = np.array(sensor.film().bitmap(raw=True)); st = scene.integrator().last_stats
    o = ob.OracleScene(d); ref = o.render(); so = o.last_stats
    if d["sensor"]["film"]["rfilter"]["type"] == "gaussian":          # neighbouring pixels are reached by atomics in arbitrary order
        assert np.allclose(gpu, ref, rtol=2e-4, atol=1e-6)
    elif d["sensor"]["sampler"].get("wavefront"):                      # a small film's samples are spread over several workgroup entries: partial sums meet by atomics
        assert np.allclose(gpu, ref, rtol=2e-5, atol=1e-7) and np.array_equal(gpu[..., 4], ref[..., 4])
    else:
        assert np.array_equal(gpu, ref), (seed, float(np.abs(gpu - ref).max()))
    assert (st["n_iter"], st["n_lookup"], st["n_nee_step"]) == (so["n_iter"], so["n_lookup"], so["n_nee_step"])


def _enlarge(d, w, h, spp):
    """The same random scene on a film of several full blocks (every ring of a 1024-path workgroup in use)."""
    if d["sensor"]["type"] in ("mdistant", "distantflux"):
        return None
    d["sensor"]["film"] = dict(d["sensor"]["film"], width=w, height=h, rfilter={"type": "box"})
    d["sensor"]["sampler"] = dict(d["sensor"]["sampler"], sample_count=spp)
    return d


@pytest.mark.parametrize("seed", range(12))
def test_random_scene_full_blocks(gpu_rgb, seed):
    d = _enlarge(_scene(100 + seed), 96, 72, 12)
    if d is None or d["integrator"]["type"] == "path":
        pytest.skip("sensor with a fixed film size / per-lane integrator")
    d["sensor"]["sampler"].pop("wavefront", None)             # this test is about the regrouping machines (scalar streams)
    gpu, st = None, None
    scene = gpu_rgb.load_dict(d)
    sensor = scene.sensors()[0]
    assert scene.integrator().render(scene, sensor, collect_counters=True)
    gpu = np.array(sensor.film().bitmap(raw=True)); st = scene.integrator().last_stats
    o = ob.OracleScene(d); ref = o.render(); so = o.last_stats
    assert np.array_equal(gpu, ref), (seed, float(np.abs(gpu - ref).max()))
    assert (st["n_iter"], st["n_lookup"], st["n_nee_step"]) == (so["n_iter"], so["n_lookup"], so["n_nee_step"])


def _to_spectral(node, rng):
    """rgb colours -> `regular` spectra over 400 .. 700 nm (the spectral variant has no sRGB upsampling model, see scene_dict._spectrum)."""
    if isinstance(node, dict):
        if node.get("type") == "rgb":
            return {"type": "regular", "lambda_min": 400.0, "lambda_max": 700.0, "values": [float(x) for x in np.atleast_1d(node["value"]).repeat(3)[:3]]}
        return {k: _to_spectral(v, rng) for k, v in node.items()}
    return node


def _scene_spectral(seed):
    rng = np.random.default_rng(1000 + seed)
    d = _scene(seed)                                          # volpathmis stays volpathmis (round 3: 4 x 4 weight matrices in the spectral variant)
    d = _to_spectral(d, rng)
    med = d.get("slab", {}).get("interior")
    if med and med["type"] == "heterogeneous" and rng.random() < 0.6:        # spectral grids for extinction and / or albedo
        xf = med["sigma_t"]["to_world"]
        res, nodes = int(rng.choice([3, 5])), int(rng.choice([2, 4, 7]))
        med["sigma_t"] = {"type": "gridvolume_spectral", "data": rng.uniform(0.1, 2.0, (res, res, res, nodes)).astype(np.float32),
                          "lambda_min": 0.0, "lambda_max": 1000.0, "to_world": xf}
        if rng.random() < 0.5:
            med["albedo"] = {"type": "gridvolume_spectral", "data": rng.uniform(0.3, 0.95, (res, res, res, nodes)).astype(np.float32),
                             "lambda_min": 0.0, "lambda_max": 1000.0, "to_world": xf}
    return d


@pytest.fixture(scope="module")
def gpu_spectral(pkg):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    pkg.set_variant("gpu_spectral")
    yield pkg
    pkg.set_variant("gpu_rgb")


@pytest.mark.parametrize("seed", range(40))
def test_random_scene_spectral(gpu_spectral, seed):
    """The same random scenes in the spectral variant: `volpath` and `volpathmis` on the four-wide ring machines (or per lane without
    media / with wavefront streams), `path` per lane, spectra on every colour parameter, spectral grids -- film and counters against
    liboracle_spectral.so."""
    d = _scene_spectral(seed)
    scene = gpu_spectral.load_dict(d)
    sensor = scene.sensors()[0]
    assert scene.integrator().render(scene, sensor, collect_counters=True)
    gpu = np.array(sensor.film().bitmap(raw=True)); st = scene.integrator().last_stats
    o = ob.OracleScene(d, spectral=True); ref = o.render(); so = o.last_stats
    if d["sensor"]["film"]["rfilter"]["type"] == "gaussian":
        assert np.allclose(gpu, ref, rtol=2e-4, atol=1e-6)
    elif d["sensor"]["sampler"].get("wavefront"):
        assert np.allclose(gpu, ref, rtol=2e-5, atol=1e-7) and np.array_equal(gpu[..., 4], ref[..., 4])
    else:
        assert np.array_equal(gpu, ref), (seed, float(np.abs(gpu - ref).max()))
    assert (st["n_iter"], st["n_lookup"], st["n_nee_step"]) == (so["n_iter"], so["n_lookup"], so["n_nee_step"])
