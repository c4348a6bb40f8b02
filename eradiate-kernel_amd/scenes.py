"""Synthetic benchmark / parity scenes C1-C5 as Mitsuba-style scene dictionaries.

Exact definitions: SURVEY.md section 8(d) ("Concrete synthetic inputs"), BASELINE.json `configs`.
Every scene uses the `independent` sampler with seed 0, block_size 32 and the box filter unless
stated otherwise, so the per-pixel random streams are those of the reference's scalar_rgb variant.
"""
import numpy as np

from .transform import ScalarTransform4f as T


def _film(width, height, rfilter="box"):
    return {"type": "hdrfilm", "width": int(width), "height": int(height), "rfilter": {"type": rfilter}}


def _sampler(spp, seed=0):
    return {"type": "independent", "sample_count": int(spp), "seed": int(seed)}


def c1_cornell(width=256, height=256, spp=64, max_depth=-1):
    """C1: Cornell box (walls + ceiling light as rectangles, teapot omitted), `path` integrator.
    Geometry after /root/reference/src/python/python/test/scenes.py:121-185; rectangle normals face inward."""
    def wall(to_world, rgb):
        return {"type": "rectangle", "to_world": to_world,
                "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": rgb}}}
    white, red, green = [0.8, 0.8, 0.8], [0.8, 0.1, 0.1], [0.1, 0.8, 0.1]
    s = T.scale([5.0, 5.0, 1.0])
    return {
        "type": "scene",
        "integrator": {"type": "path", "max_depth": max_depth, "rr_depth": 5, "block_size": 32},
        "sensor": {"type": "perspective",
                   "to_world": T.look_at([0, -14, 3.5], [0, 0, 3.5], [0, 0, 1]),
                   "fov": 40.0, "near_clip": 1.0, "far_clip": 1000.0,
                   "film": _film(width, height), "sampler": _sampler(spp)},
        "floor": wall(T.translate([0, 0, 0]) @ s, white),
        "ceiling": wall(T.translate([0, 0, 7]) @ T.rotate([1, 0, 0], 180) @ s, white),
        "back": wall(T.translate([0, 5, 3.5]) @ T.rotate([1, 0, 0], 90) @ T.scale([5.0, 3.5, 1.0]), white),
        "left": wall(T.translate([-5, 0, 3.5]) @ T.rotate([0, 1, 0], 90) @ T.scale([3.5, 5.0, 1.0]), red),
        "right": wall(T.translate([5, 0, 3.5]) @ T.rotate([0, 1, 0], -90) @ T.scale([3.5, 5.0, 1.0]), green),
        "light": {"type": "rectangle",
                  "to_world": T.translate([0, 0, 6.99]) @ T.rotate([1, 0, 0], 180) @ T.scale([1.5, 1.5, 1.0]),
                  "emitter": {"type": "area", "radiance": {"type": "rgb", "value": [3.0, 3.0, 3.0]}}},
    }


def _slab_scene(medium, width, height, spp, max_depth, rr_depth, samples_per_pass=-1):
    return {
        "type": "scene",
        "integrator": {"type": "volpath", "max_depth": max_depth, "rr_depth": rr_depth, "block_size": 32,
                       "samples_per_pass": samples_per_pass},
        "sensor": {"type": "perspective",
                   "to_world": T.look_at([0, 0, 20], [0, 0, 0], [0, 1, 0]),
                   "fov": 45.0, "near_clip": 0.1, "far_clip": 100.0,
                   "film": _film(width, height), "sampler": _sampler(spp)},
        "slab": {"type": "cube", "to_world": T.translate([0, 0, 1]) @ T.scale([50, 50, 1]),
                 "bsdf": {"type": "null"}, "interior": medium},
        "ground": {"type": "rectangle", "to_world": T.translate([0, 0, -0.01]) @ T.scale(60.0),
                   "bsdf": {"type": "diffuse", "reflectance": {"type": "rgb", "value": [0.5, 0.5, 0.5]}}},
        "sun": {"type": "directional", "direction": [0.5, 0.0, -0.866], "irradiance": 1.0},
    }


def c2_homogeneous_slab(width=512, height=512, spp=256, max_depth=-1, rr_depth=5, sigma_t=1.0, albedo=0.8, phase=None):
    """C2: homogeneous slab (sigma_t 1, albedo 0.8, isotropic) over a Lambertian ground, directional emitter."""
    medium = {"type": "homogeneous", "sigma_t": sigma_t, "albedo": albedo,
              "phase": phase if phase is not None else {"type": "isotropic"}}
    return _slab_scene(medium, width, height, spp, max_depth, rr_depth)


def c3_sigma_t_grid(res=128, seed=1234):
    """sigma[k,j,i] = 0.1 + 2 exp(-4 (k + 0.5) / res) (0.75 + 0.25 u[k,j,i]), u ~ default_rng(seed) in C order."""
    u = np.random.default_rng(seed).random((res, res, res), dtype=np.float32)
    k = (np.arange(res, dtype=np.float32) + np.float32(0.5)) / np.float32(res)
    prof = (np.float32(2.0) * np.exp(np.float32(-4.0) * k)).astype(np.float32)
    sig = np.float32(0.1) + prof[:, None, None] * (np.float32(0.75) + np.float32(0.25) * u)
    return sig.astype(np.float32)


def c3_heterogeneous(width=512, height=512, spp=1024, res=128, max_depth=-1, rr_depth=5, g=0.8, albedo=0.9,
                     samples_per_pass=-1, grid_seed=1234):
    """C3 (the metric scene): heterogeneous res^3 grid medium (delta tracking) + HG g = 0.8."""
    grid_xf = T.translate([-50, -50, 0]) @ T.scale([100, 100, 2])
    medium = {
        "type": "heterogeneous",
        "sigma_t": {"type": "gridvolume", "data": c3_sigma_t_grid(res, grid_seed), "to_world": grid_xf},
        "albedo": {"type": "gridvolume", "data": np.full((res, res, res), albedo, dtype=np.float32), "to_world": grid_xf},
        "scale": 1.0,
        "phase": {"type": "hg", "g": g},
    }
    return _slab_scene(medium, width, height, spp, max_depth, rr_depth, samples_per_pass)


def hg_table(g=0.7, n=181):
    """Tabulated HG phase function on a regular cos(theta) grid, in the convention of tabphase
    (values indexed by cos(theta) over [-1, 1], /root/reference/src/phase/tabphase.cpp)."""
    mu = np.linspace(-1.0, 1.0, n)
    val = (1.0 / (4.0 * np.pi)) * (1.0 - g * g) / np.power(1.0 + g * g - 2.0 * g * mu, 1.5)
    return " ".join("%.9g" % v for v in val.astype(np.float32))


C4_GROUND_Z = 0.05


def c4_atmosphere(width=1024, height=1024, spp=4096, layers=64, sigma_r0=0.012, sigma_a0=0.1, sza_deg=30.0,
                  rayleigh_scale=1.0, samples_per_pass=-1, columns=2):
    """C4: plane-parallel atmosphere, 50 km thick and 2*10^4 km wide: Rayleigh (scale height 8) + aerosol
    (scale height 2) extinction in `layers` homogeneous layers, blendphase(rayleigh, tabphase(HG 0.7)) weighted
    by the aerosol scattering fraction, RPV ground, directional sun, distant sensor over the hemisphere."""
    top = 50.0
    z = (np.arange(layers, dtype=np.float64) + 0.5) * (top / layers)
    s_r = rayleigh_scale * sigma_r0 * np.exp(-z / 8.0)
    s_a = sigma_a0 * np.exp(-z / 2.0)
    alb_r, alb_a = 1.0, 0.92
    sigma_t = (s_r + s_a).astype(np.float32)
    sigma_s = s_r * alb_r + s_a * alb_a
    albedo = (sigma_s / (s_r + s_a)).astype(np.float32)
    weight = (s_a * alb_a / sigma_s).astype(np.float32)
    # (nz, ny, nx) = (layers, columns, columns); the reference requires >= 8 voxels in all (volume_data.h:67-72), so columns = 1 -- the
    # nz x 1 x 1 grid of a 1-D atmosphere -- is a legal file from 8 layers on
    def grid(v):
        return np.ascontiguousarray(np.broadcast_to(v[:, None, None], (layers, columns, columns)), dtype=np.float32)
    ext = 1.0e4
    grid_xf = T.translate([-ext, -ext, 0]) @ T.scale([2 * ext, 2 * ext, top])
    sun = [np.sin(np.radians(sza_deg)), 0.0, -np.cos(np.radians(sza_deg))]
    return {
        "type": "scene",
        "integrator": {"type": "volpath", "max_depth": -1, "rr_depth": 5, "block_size": 32, "samples_per_pass": samples_per_pass},
        "sensor": {"type": "distant", "direction": [0, 0, 1],
                   "ray_target": {"type": "rectangle", "to_world": T.translate([0, 0, top]) @ T.scale(1.0)},
                   "film": _film(width, height), "sampler": _sampler(spp)},
        "atmosphere": {"type": "cube", "to_world": T.translate([0, 0, top / 2]) @ T.scale([ext, ext, top / 2]),
                       "bsdf": {"type": "null"},
                       "interior": {"type": "heterogeneous",
                                    "sigma_t": {"type": "gridvolume", "data": grid(sigma_t), "to_world": grid_xf},
                                    "albedo": {"type": "gridvolume", "data": grid(albedo), "to_world": grid_xf},
                                    "phase": {"type": "blendphase",
                                              "phase_0": {"type": "rayleigh"},
                                              "phase_1": {"type": "tabphase", "values": hg_table(0.7, 181)},
                                              "weight": {"type": "gridvolume", "data": grid(weight), "to_world": grid_xf}}}},
        # The ground lies INSIDE the medium (Eradiate's own arrangement), 0.05 above the cube's bottom face.  A ground just below the
        # cube, as in the slab scenes, leaks light at this scale: a ray leaving a surface point p starts at (1 + max|p|) RayEpsilon
        # (interaction.h:58-61), i.e. 0.014 .. 1 at |p| = 160 .. 10^4 -- beyond the bottom face, so the medium was never entered and
        # the ground saw an unattenuated sun (found by the independent estimator, tests/independent/).
        "ground": {"type": "rectangle", "to_world": T.translate([0, 0, C4_GROUND_Z]) @ T.scale(1.2 * ext),
                   "bsdf": {"type": "rpv", "rho_0": 0.1, "k": 0.6, "g": -0.2}},
        "sun": {"type": "directional", "direction": sun, "irradiance": 1.0},
    }


def c4_three_species(width=1024, height=1024, spp=4096, layers=64, g_cloud=0.85, chain=False):
    """The C4 atmosphere with a third species: blend(blend(rayleigh, tabulated aerosol; w_a), hg cloud droplets; w_c) -- a blendphase
    inside a blendphase (src/phase/blendphase.cpp:42-66), every weight a grid over the layers.  chain=True nests three levels on the right."""
    d = c4_atmosphere(width, height, spp, layers=layers)
    ph = d["atmosphere"]["interior"]["phase"]
    xf = ph["weight"]["to_world"]
    wc = np.ascontiguousarray(np.broadcast_to((0.15 + 0.6 * np.exp(-np.arange(layers) / (layers / 4.0))).astype(np.float32)[:, None, None], (layers, 2, 2)))
    tree = {"type": "blendphase", "phase_0": ph, "phase_1": {"type": "hg", "g": g_cloud}, "weight": {"type": "gridvolume", "data": wc, "to_world": xf}}
    if chain:
        tree = {"type": "blendphase", "phase_0": {"type": "isotropic"}, "phase_1": {"type": "blendphase", "phase_0": {"type": "hg", "g": -0.3}, "phase_1": tree, "weight": 0.7},
                "weight": {"type": "gridvolume", "data": np.ascontiguousarray(1.0 - 0.5 * wc), "to_world": xf}}
    d["atmosphere"]["interior"]["phase"] = tree
    return d


def c5_atmosphere_spectral(width=1024, height=1024, spp=4096, layers=64, nodes=17, samples_per_pass=-1, columns=2):
    """C5 in the spectral variant (gpu_spectral): the C4 atmosphere with extinction and albedo as `gridvolume_spectral` grids whose
    `nodes` spectral nodes cover 0 .. 1600 nm (Rayleigh ~ lambda^-4 relative to 550 nm, aerosol grey), a D65 sun and an RPV ground with
    a sloped rho_0.  lambda_min = 0 because of the mask gridvolume_spectral applies (tests/test_spectral.py::test_gridvolume_spectral_eval).
    The blend weight stays a plain grid (gridvolume_spectral has no eval_1, gridvolume_spectral.cpp:204-214): its 550 nm value."""
    d = c4_atmosphere(width, height, spp, layers=layers, samples_per_pass=samples_per_pass, columns=columns)
    top = 50.0
    z = (np.arange(layers, dtype=np.float64) + 0.5) * (top / layers)
    lam = np.linspace(0.0, 1600.0, nodes)
    ray = 0.012 * np.exp(-z / 8.0)[:, None] * (550.0 / np.maximum(lam, 350.0)[None, :]) ** 4       # flat below 350 nm (never sampled: 360 .. 830)
    aer = (0.1 * np.exp(-z / 2.0))[:, None] * np.ones_like(lam)[None, :]
    sigma_t = ray + aer
    albedo = (ray * 1.0 + aer * 0.92) / sigma_t
    def grid(v):
        return np.ascontiguousarray(np.broadcast_to(v[:, None, None, :], (layers, columns, columns, nodes)), dtype=np.float32)
    med = d["atmosphere"]["interior"]
    xf = med["sigma_t"]["to_world"]
    med["sigma_t"] = {"type": "gridvolume_spectral", "data": grid(sigma_t), "lambda_min": 0.0, "lambda_max": 1600.0, "to_world": xf}
    med["albedo"] = {"type": "gridvolume_spectral", "data": grid(albedo), "lambda_min": 0.0, "lambda_max": 1600.0, "to_world": xf}
    del d["sun"]["irradiance"]                                                                  # D65 (directional.cpp:49)
    d["ground"]["bsdf"]["rho_0"] = {"type": "regular", "lambda_min": 300.0, "lambda_max": 900.0, "values": [0.05, 0.1, 0.3]}
    return d


CONFIGS = {
    "C1": ("Cornell box, path, 256x256x64spp", c1_cornell),
    "C2": ("volpath homogeneous slab, 512x512x256spp", c2_homogeneous_slab),
    "C3": ("volpath heterogeneous 128^3 grid + HG, 512x512x1024spp", c3_heterogeneous),
    "C4": ("plane-parallel atmosphere, distant sensor, 1024x1024x4096spp", c4_atmosphere),
}
