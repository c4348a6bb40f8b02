"""An estimator of the radiance seen through a scattering slab that shares NOTHING with oracle/ or the product.

TEST INFRASTRUCTURE.  Purpose ("manufacture an independent pin for volpath"): the oracle and the
HIP kernels restate the same reading of /root/reference/src/integrators/volpath.cpp, so their bit-equality says
nothing about that reading.  This file computes the same radiometric quantity in a structurally different way:

  * numpy float64, counter-based Philox random numbers (no PCG32 stream, no sampler draw order);
  * free paths by INVERTING A RAY-MARCHED OPTICAL DEPTH (midpoint quadrature of the trilinearly interpolated
    extinction along the ray, refined in a second pass) -- no majorant, no delta tracking, no null collisions;
  * transmittance towards the sun by the same quadrature -- no ratio tracking;
  * the collision estimator of the transport equation with the attenuated solar beam as the explicit source
    (every real collision / ground hit scores  weight x phase-or-BRDF x E x T_sun): no emitter sampling routine,
    no multiple importance sampling, no Russian roulette (weights, fixed event cap);
  * its own trilinear lookup (the cell-centred convention of src/textures/grid3d.cpp:259-341: x = p n - 1/2, clamped
    indices), its own Henyey-Greenstein / Rayleigh / tabulated phase functions, Lambertian and RPV ground, pinhole
    and distant-hemisphere sensors -- all written from the formulas, none imported.

It must not import anything from eradiate-kernel_amd/ or oracle/ (tests/test_independent_pin.py checks that).
Scenes are passed in as plain numbers and arrays (`SlabProblem`).
"""
import math

import numpy as np


class SlabProblem:
    """A box of scattering medium with a transparent boundary over a flat reflecting ground, lit by a distant sun."""

    def __init__(self, box_min, box_max, sigma_t_grid, albedo, phase, ground_z, ground_half, ground, sun_dir, irradiance=1.0,
                 sigma_scale=1.0, blend_weight_grid=None):
        self.bmin = np.asarray(box_min, np.float64)
        self.bmax = np.asarray(box_max, np.float64)
        self.sig = np.asarray(sigma_t_grid, np.float64) * float(sigma_scale)      # [z, y, x]
        self.albedo = albedo                                                     # scalar or grid [z, y, x]
        self.phase = phase                                                       # ("hg", g) | ("blend", table_values) | ("mix3", table_values, g_third)
        self.wgrid = None if blend_weight_grid is None else np.asarray(blend_weight_grid, np.float64)
        self.wgrid3 = None                                                       # "mix3": share of the third species (set by the caller)
        self.ground_z, self.ground_half = float(ground_z), float(ground_half)
        self.ground = ground                                                     # ("diffuse", rho) | ("rpv", rho0, k, g, rho_c, continuation)
        s = np.asarray(sun_dir, np.float64)
        self.sun = s / np.linalg.norm(s)                                         # direction of propagation of the sunlight
        self.E = float(irradiance)
        if phase[0] in ("blend", "mix3"):
            tab = np.asarray(phase[1], np.float64)
            dx = 2.0 / (len(tab) - 1)
            self.tab = tab / (np.sum(0.5 * (tab[1:] + tab[:-1])) * dx)           # piecewise linear, unit integral over mu in [-1, 1]

    # ---- fields -------------------------------------------------------------------------------------------------
    def _trilinear(self, grid, p):
        """Cell-centred trilinear interpolation with clamped indices; p: (..., 3) world points."""
        nz, ny, nx = grid.shape
        q = (p - self.bmin) / (self.bmax - self.bmin)
        fx, fy, fz = q[..., 0] * nx - 0.5, q[..., 1] * ny - 0.5, q[..., 2] * nz - 0.5
        ix, iy, iz = np.floor(fx).astype(np.int64), np.floor(fy).astype(np.int64), np.floor(fz).astype(np.int64)
        wx, wy, wz = fx - ix, fy - iy, fz - iz
        x0, x1 = np.clip(ix, 0, nx - 1), np.clip(ix + 1, 0, nx - 1)
        y0, y1 = np.clip(iy, 0, ny - 1), np.clip(iy + 1, 0, ny - 1)
        z0, z1 = np.clip(iz, 0, nz - 1), np.clip(iz + 1, 0, nz - 1)
        def row(zz, yy):
            return grid[zz, yy, x0] * (1.0 - wx) + grid[zz, yy, x1] * wx
        lo = row(z0, y0) * (1.0 - wy) + row(z0, y1) * wy
        hi = row(z1, y0) * (1.0 - wy) + row(z1, y1) * wy
        return lo * (1.0 - wz) + hi * wz

    def sigma_t(self, p):
        return self._trilinear(self.sig, p)

    def albedo_at(self, p):
        if np.isscalar(self.albedo):
            return np.full(p.shape[:-1], float(self.albedo))
        return self._trilinear(np.asarray(self.albedo, np.float64), p)

    # ---- phase functions (mu = cosine between the directions of propagation before and after scattering) ------------
    def _tab_eval(self, mu):
        n = len(self.tab)
        x = (np.clip(mu, -1.0, 1.0) + 1.0) * 0.5 * (n - 1)
        i = np.minimum(np.floor(x).astype(np.int64), n - 2)
        w = x - i
        return (self.tab[i] * (1.0 - w) + self.tab[i + 1] * w) / (2.0 * math.pi)

    def phase_eval(self, mu, p):
        if self.phase[0] == "hg":
            g = self.phase[1]
            return (1.0 - g * g) / (4.0 * math.pi * (1.0 + g * g - 2.0 * g * mu) ** 1.5)
        w = self._trilinear(self.wgrid, p)                                       # probability / weight of the tabulated lobe
        ray = 3.0 / (16.0 * math.pi) * (1.0 + mu * mu)
        two = (1.0 - w) * ray + w * self._tab_eval(mu)
        if self.phase[0] == "blend":
            return two
        # "mix3": a third species (Henyey-Greenstein, asymmetry phase[2]) with local share w3 beside the two-species mixture
        w3 = self._trilinear(self.wgrid3, p)
        g = self.phase[2]
        return (1.0 - w3) * two + w3 * (1.0 - g * g) / (4.0 * math.pi * (1.0 + g * g - 2.0 * g * mu) ** 1.5)

    def phase_sample(self, rng, d, p):
        """New direction of the walk and the weight phase / pdf."""
        n = d.shape[0]
        u1, u2 = rng.random(n), rng.random(n)
        if self.phase[0] == "hg":
            g = self.phase[1]
            s = (1.0 - g * g) / (1.0 - g + 2.0 * g * u1)
            mu = (1.0 + g * g - s * s) / (2.0 * g)
            weight = np.ones(n)
        else:
            # proposal: isotropic for the Rayleigh share, HG(0.7) for the tabulated share (and the third species' own HG for its share);
            # weight = true phase / proposal pdf
            w = self._trilinear(self.wgrid, p)
            w3 = self._trilinear(self.wgrid3, p) if self.phase[0] == "mix3" else np.zeros(n)
            g3 = self.phase[2] if self.phase[0] == "mix3" else 0.5
            pick3 = rng.random(n) < w3
            pick_tab = rng.random(n) < w
            g = 0.7
            s = (1.0 - g * g) / (1.0 - g + 2.0 * g * u1)
            s3 = (1.0 - g3 * g3) / (1.0 - g3 + 2.0 * g3 * u1)
            mu = np.where(pick3, (1.0 + g3 * g3 - s3 * s3) / (2.0 * g3), np.where(pick_tab, (1.0 + g * g - s * s) / (2.0 * g), 1.0 - 2.0 * u1))
            q2 = (1.0 - w) / (4.0 * math.pi) + w * (1.0 - g * g) / (4.0 * math.pi * (1.0 + g * g - 2.0 * g * mu) ** 1.5)
            q = (1.0 - w3) * q2 + w3 * (1.0 - g3 * g3) / (4.0 * math.pi * (1.0 + g3 * g3 - 2.0 * g3 * mu) ** 1.5)
            weight = self.phase_eval(mu, p) / q
        mu = np.clip(mu, -1.0, 1.0)
        return _rotate_about(d, mu, 2.0 * math.pi * u2), weight

    # ---- ground ---------------------------------------------------------------------------------------------------
    def ground_brdf(self, wi, wo):
        """BRDF for unit vectors pointing away from the ground (z up)."""
        if self.ground[0] == "diffuse":
            return np.full(wi.shape[0], self.ground[1] / math.pi)
        _, rho0, k, g, rho_c, _ = self.ground
        c1, c2 = wi[:, 2], wo[:, 2]
        s1, s2 = np.sqrt(np.maximum(0.0, 1.0 - c1 * c1)), np.sqrt(np.maximum(0.0, 1.0 - c2 * c2))
        t1, t2 = s1 / c1, s2 / c2
        n1, n2 = np.maximum(s1, 1e-300), np.maximum(s2, 1e-300)
        cos_dphi = np.where((s1 > 0) & (s2 > 0), (wi[:, 0] * wo[:, 0] + wi[:, 1] * wo[:, 1]) / (n1 * n2), 1.0)
        G = np.sqrt(np.maximum(0.0, t1 * t1 + t2 * t2 - 2.0 * t1 * t2 * cos_dphi))
        cos_g = c1 * c2 + s1 * s2 * cos_dphi
        F = (1.0 - g * g) / (1.0 + g * g + 2.0 * g * cos_g) ** 1.5
        return rho0 * (c1 * c2 * (c1 + c2)) ** (k - 1.0) * F * (1.0 + (1.0 - rho_c) / (1.0 + G)) / math.pi

    # ---- geometry -------------------------------------------------------------------------------------------------
    def box_interval(self, o, d):
        """Parametric interval of the ray inside the box (t0 >= 0); empty intervals have t1 <= t0."""
        with np.errstate(divide="ignore", invalid="ignore"):
            inv = 1.0 / d
            ta, tb = (self.bmin - o) * inv, (self.bmax - o) * inv
        lo, hi = np.minimum(ta, tb), np.maximum(ta, tb)
        par = d == 0.0                                                           # parallel to a slab: inside or never
        inside = (o >= self.bmin) & (o <= self.bmax)
        lo = np.where(par, np.where(inside, -np.inf, np.inf), lo)
        hi = np.where(par, np.where(inside, np.inf, -np.inf), hi)
        return np.maximum(lo.max(axis=1), 0.0), hi.min(axis=1)

    def optical_depth(self, o, d, t0, t1, steps):
        """Midpoint quadrature of sigma_t over [t0, t1]; returns per-step optical thickness (n, steps) and the step length."""
        h = np.maximum(t1 - t0, 0.0) / steps
        tm = t0[:, None] + (np.arange(steps) + 0.5)[None, :] * h[:, None]
        pts = o[:, None, :] + tm[:, :, None] * d[:, None, :]
        return self.sigma_t(pts) * h[:, None], h

    def transmittance_to_sun(self, p, steps=96):
        d = np.broadcast_to(-self.sun, p.shape)
        t0, t1 = self.box_interval(p, d)
        hit = t1 > t0
        tau = np.zeros(p.shape[0])
        if hit.any():
            dt, _ = self.optical_depth(p[hit], d[hit], t0[hit], t1[hit], steps)
            tau[hit] = dt.sum(axis=1)
        return np.exp(-tau)

    def sample_collision(self, rng, o, d, t0, t1, steps=96, refine=16):
        """Distance to the next real collision inside [t0, t1] by inverting the marched optical depth; inf = none."""
        n = o.shape[0]
        tau = -np.log1p(-rng.random(n))
        dt, h = self.optical_depth(o, d, t0, t1, steps)
        cum = np.cumsum(dt, axis=1)
        hit = cum[:, -1] >= tau
        t = np.full(n, np.inf)
        if hit.any():
            idx = np.argmax(cum[hit] >= tau[hit, None], axis=1)
            before = np.where(idx > 0, cum[hit, np.maximum(idx - 1, 0)], 0.0)
            rest = tau[hit] - before                                             # optical depth still to go inside coarse step idx
            a = t0[hit] + idx * h[hit]
            b = a + h[hit]
            dt2, h2 = self.optical_depth(o[hit], d[hit], a, b, refine)           # second pass: the coarse step in `refine` pieces
            scale = dt[hit, idx] / np.maximum(dt2.sum(axis=1), 1e-300)           # keep both passes consistent
            cum2 = np.cumsum(dt2 * scale[:, None], axis=1)
            j = np.minimum(np.argmax(cum2 >= rest[:, None], axis=1), refine - 1)
            j = np.where(cum2[:, -1] >= rest, j, refine - 1)
            before2 = np.where(j > 0, cum2[np.arange(len(j)), np.maximum(j - 1, 0)], 0.0)
            step_tau = (dt2 * scale[:, None])[np.arange(len(j)), j]
            frac = np.clip((rest - before2) / np.maximum(step_tau, 1e-300), 0.0, 1.0)
            t[hit] = a + (j + frac) * h2
        return t


def _rotate_about(d, mu, phi):
    """Unit vectors at polar cosine mu and azimuth phi about the unit vectors d."""
    helper = np.where(np.abs(d[:, [0]]) > 0.9, np.array([[0.0, 1.0, 0.0]]), np.array([[1.0, 0.0, 0.0]]))
    a = np.cross(d, helper)
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    b = np.cross(d, a)
    s = np.sqrt(np.maximum(0.0, 1.0 - mu * mu))
    out = (s * np.cos(phi))[:, None] * a + (s * np.sin(phi))[:, None] * b + mu[:, None] * d
    return out / np.linalg.norm(out, axis=1, keepdims=True)


def radiance(problem, rng, o, d, max_events=400, steps=96, refine=16):
    """Radiance arriving at points o from directions d (i.e. carried by light travelling along -d), one walk per ray."""
    P = problem
    n = o.shape[0]
    o, d = o.copy(), d.copy()
    w = np.ones(n)
    total = np.zeros(n)
    alive = np.ones(n, bool)
    up = np.array([0.0, 0.0, 1.0])
    cos_sun = max(0.0, -P.sun[2])
    for _ in range(max_events):
        ids = np.nonzero(alive)[0]
        if ids.size == 0:
            break
        oo, dd = o[ids], d[ids]
        t0, t1 = P.box_interval(oo, dd)
        in_box = t1 > t0
        with np.errstate(divide="ignore", invalid="ignore"):
            tg = np.where(dd[:, 2] < 0.0, (P.ground_z - oo[:, 2]) / dd[:, 2], np.inf)
        pg = oo + np.where(np.isfinite(tg), tg, 0.0)[:, None] * dd
        tg = np.where((np.abs(pg[:, 0]) <= P.ground_half) & (np.abs(pg[:, 1]) <= P.ground_half) & (tg > 0.0), tg, np.inf)
        tc = np.full(ids.size, np.inf)
        go = in_box & (t0 < tg)                                                  # the medium is reached before the ground
        if go.any():
            tend = np.minimum(t1[go], tg[go])
            tc[go] = P.sample_collision(rng, oo[go], dd[go], t0[go], tend, steps, refine)
        coll = np.isfinite(tc)
        grd = ~coll & np.isfinite(tg)
        gone = ~coll & ~grd
        alive[ids[gone]] = False
        if coll.any():
            k = ids[coll]
            x = oo[coll] + tc[coll][:, None] * dd[coll]
            alb = P.albedo_at(x)
            mu_sun = -(dd[coll] @ P.sun)                                         # sunlight (along sun) scattered into -d
            total[k] += w[k] * alb * P.phase_eval(mu_sun, x) * P.E * P.transmittance_to_sun(x, steps)
            nd, pw = P.phase_sample(rng, dd[coll], x)
            w[k] *= alb * pw
            o[k], d[k] = x, nd
        if grd.any():
            k = ids[grd]
            y = oo[grd] + tg[grd][:, None] * dd[grd]
            wi = -dd[grd]
            sunv = np.broadcast_to(-P.sun, wi.shape)
            total[k] += w[k] * P.ground_brdf(wi, sunv) * cos_sun * P.E * P.transmittance_to_sun(y, steps)
            u1, u2 = rng.random(k.size), rng.random(k.size)                      # cosine-weighted direction about +z
            r, ph = np.sqrt(u1), 2.0 * math.pi * u2
            nd = np.stack([r * np.cos(ph), r * np.sin(ph), np.sqrt(np.maximum(0.0, 1.0 - u1))], axis=1)
            if P.ground[0] == "diffuse":
                w[k] *= P.ground[1]
            else:
                # physically the weight of a cosine-sampled direction is pi x BRDF; the reference's RPV::sample returns the BRDF
                # value itself (src/bsdfs/rpv.cpp:99-101), which is what "reference" reproduces
                w[k] *= P.ground_brdf(wi, nd) * (math.pi if P.ground[5] == "physical" else 1.0)
            o[k], d[k] = y + 1e-9 * up, nd
        alive[ids] &= w[ids] > 1e-12
    return total


# ---- sensors ----------------------------------------------------------------------------------------------------------
def pinhole_rays(rng, width, height, fov_x_deg, origin, forward, upv, pix):
    """Pinhole camera in the convention of a film whose x runs to the camera's right seen from behind and y runs down:
    sample (sx, sy) in [0,1]^2 -> direction forward + (1 - 2 sx) tan(fov/2) left + (1 - 2 sy) tan(fov/2) / aspect up'."""
    n = pix.shape[0]
    f = np.asarray(forward, np.float64); f /= np.linalg.norm(f)
    left = np.cross(np.asarray(upv, np.float64), f); left /= np.linalg.norm(left)
    up2 = np.cross(f, left)
    sx = ((pix % width) + rng.random(n)) / width
    sy = ((pix // width) + rng.random(n)) / height
    tan = math.tan(math.radians(fov_x_deg) / 2.0)
    aspect = width / height
    d = f[None, :] + ((1.0 - 2.0 * sx) * tan)[:, None] * left[None, :] + ((1.0 - 2.0 * sy) * tan / aspect)[:, None] * up2[None, :]
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.broadcast_to(np.asarray(origin, np.float64), (n, 3)).copy(), d


def _concentric_disk(u, v):
    x, y = 2.0 * u - 1.0, 2.0 * v - 1.0
    use_y = np.abs(x) < np.abs(y)
    r = np.where(use_y, y, x)
    rp = np.where(use_y, x, y)
    with np.errstate(divide="ignore", invalid="ignore"):
        phi = np.where(r != 0.0, 0.25 * math.pi * rp / r, 0.0)
    phi = np.where(use_y, 0.5 * math.pi - phi, phi)
    return r * np.cos(phi), r * np.sin(phi)


def distant_hemisphere_rays(rng, width, height, target_lo, target_hi, target_z, start_height, pix):
    """Distant sensor looking down: pixel (i, j) covers the directions v = hemisphere((i + u) / W, (j + v) / H) (equal-area map
    of the square onto the upper hemisphere through the concentric disk map); rays travel along -v towards a point drawn
    uniformly on the rectangle target."""
    n = pix.shape[0]
    sx = ((pix % width) + rng.random(n)) / width
    sy = ((pix // width) + rng.random(n)) / height
    px, py = _concentric_disk(sx, sy)
    z = 1.0 - (px * px + py * py)
    sc = np.sqrt(z + 1.0)
    v = np.stack([px * sc, py * sc, z], axis=1)
    d = -v
    tgt = np.stack([target_lo[0] + (target_hi[0] - target_lo[0]) * rng.random(n),
                    target_lo[1] + (target_hi[1] - target_lo[1]) * rng.random(n), np.full(n, float(target_z))], axis=1)
    return tgt - d * start_height, d


def render(problem, sensor, width, height, per_pixel, seed=1, chunk=20000, steps=96, refine=16, progress=None):
    """Per-pixel mean radiance and the variance of that mean, `per_pixel` independent walks per pixel."""
    rng = np.random.Generator(np.random.Philox(seed))
    npx = width * height
    s1, s2 = np.zeros(npx), np.zeros(npx)
    pix_all = np.tile(np.arange(npx), per_pixel)
    for a in range(0, pix_all.size, chunk):
        pix = pix_all[a:a + chunk]
        o, d = sensor(rng, pix)
        val = radiance(problem, rng, o, d, steps=steps, refine=refine)
        np.add.at(s1, pix, val)
        np.add.at(s2, pix, val * val)
        if progress:
            progress(a + pix.size, pix_all.size)
    mean = s1 / per_pixel
    var = np.maximum(s2 / per_pixel - mean * mean, 0.0) / max(per_pixel - 1, 1)
    return mean.reshape(height, width), var.reshape(height, width)
