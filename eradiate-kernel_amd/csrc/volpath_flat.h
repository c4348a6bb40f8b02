// volpath_flat.h -- volpath (integrators/volpath.cpp:38-465) as ONE flat per-lane state machine.
//
// Why: the reference nests three loops (main loop :72, NEE ratio tracking :282, direct-light walk :385).
// Compiled as written, a wave serialises them: while a few lanes run an inner tracking loop to
// completion the other lanes of the wave idle.  Here every lane carries (mode, state) and each trip of
// the single loop below advances every lane by at most one delta-tracking step -- whichever of the
// three reference loops that step belongs to -- so the expensive block (free-flight sample + grid
// gathers) is executed by all lanes together.
//
// Wave-level scheduling: the lanes of a wave sit in different states, and running every block on every
// trip would execute each block with a handful of active lanes (measured: 17 % VALU lane utilisation).
// So each trip first runs the cheap loop-head dispatch, then takes a census of the states with
// __ballot / __popcll and executes only the ONE heavy block most lanes are waiting for; the others wait
// (their state is in registers, nothing is lost) until their block wins the vote.
//
// The random draws happen in exactly the order of the scalar_rgb variant (SURVEY.md 8(a')); results are
// bit-identical to the nested formulation in integrator_dev.h and to the CPU restatement.
// Citations are relative to /root/reference.
#pragma once
#include "integrator_dev.h"

namespace mtsamd {

enum : uint32_t { S_TOP = 0, S_MED = 1, S_SURF = 2, S_PHASE = 3, S_BSDF = 4, S_DIRB = 5, S_NEW = 6, S_DONE = 7 };
enum : uint32_t { M_MAIN = 0, M_NEE = 1, M_DIR = 2 };
enum : uint32_t { FL_ALIVE = 1, FL_VALID_RAY = 2, FL_SPEC_CHAIN = 4, FL_NEEDS_INT = 8, FL_FROM_MEDIUM = 16 };

// Result of one free-flight sample (librender/medium.cpp:34-75); sigma_n is derived by the caller
// (heterogeneous.cpp:46: combined - sigma_t, homogeneous.cpp:44: 0).
struct MedStep { float t, mint; F3 p, sigma_t, sigma_s, combined; uint32_t info; };
enum : uint32_t { MI_HOMOGENEOUS = 1, MI_SPECTRAL = 2, MI_SAMPLE_EMITTERS = 4, MI_GREY = 8, MI_PHASE_SHIFT = 8 + 8 };

#if defined(MTSAMD_BLOCKSTATS)
__device__ unsigned long long g_blockstats[32];    // [2b]: executions, [2b+1]: lanes served, [16+b]: cycles
#endif

// textures/grid3d.cpp:259-341 split in two: cell coordinates / weights (shared by grids with the same
// transform and resolution) and the 8 gathers + trilinear blend of one grid.
struct GridCell { int32_t r00, r10, r01, r11, x0, x1; F3 w0, w1; };
DEV GridCell grid_cell(const DVolume &v, F3 p_world) {
    F3 p = v.affine ? mat_point_affine(v.w2l, p_world) : mat_point(v.w2l, p_world);
    const int nx = v.nx, ny = v.ny, nz = v.nz;
    p = f3(pm_fma(p.x, (float) nx, -.5f), pm_fma(p.y, (float) ny, -.5f), pm_fma(p.z, (float) nz, -.5f));
    int ix = (int) pm_floor(p.x), iy = (int) pm_floor(p.y), iz = (int) pm_floor(p.z);
    GridCell c;
    c.w1 = p - f3((float) ix, (float) iy, (float) iz); c.w0 = f3(1.f - c.w1.x, 1.f - c.w1.y, 1.f - c.w1.z);
    c.x0 = wrap_coord(v.wrap, ix, nx); c.x1 = wrap_coord(v.wrap, ix + 1, nx);
    int y0 = wrap_coord(v.wrap, iy, ny), y1 = wrap_coord(v.wrap, iy + 1, ny), z0 = wrap_coord(v.wrap, iz, nz), z1 = wrap_coord(v.wrap, iz + 1, nz);
    c.r00 = (z0 * ny + y0) * nx; c.r10 = (z0 * ny + y1) * nx; c.r01 = (z1 * ny + y0) * nx; c.r11 = (z1 * ny + y1) * nx;
    return c;
}
DEV float grid_fetch1(const float *__restrict__ D, const GridCell &c) {
#if defined(EXP_NOGATHER)
    return trilerp(0.5f, 0.6f, 0.7f, 0.8f, 0.9f, 1.0f, 1.1f, 1.2f, c.w0, c.w1) + 1e-9f * (float) (c.r00 + c.x0 + c.r11 + c.x1 + c.r01 + c.r10);
#endif
    return trilerp(D[c.r00 + c.x0], D[c.r00 + c.x1], D[c.r10 + c.x0], D[c.r10 + c.x1],
                   D[c.r01 + c.x0], D[c.r01 + c.x1], D[c.r11 + c.x0], D[c.r11 + c.x1], c.w0, c.w1);
}

template <bool COUNT>
DEV MedStep medium_step(const DScene &sc, const DMedium m, const DRay &ray, float sample, uint32_t channel, bool want_albedo, Counters &cnt) {
    MedStep mi;
    bool active = true; float mint = 0.f, maxt = pm_inf();
    if (!m.is_homogeneous) {
        active = bbox_ray_intersect(m.aabb, ray, mint, maxt);
        active = active && (pm_isfinite(mint) || pm_isfinite(maxt));
        if (!active) { mint = 0.f; maxt = pm_inf(); }
    }
    mint = pm_max(ray.mint, mint);
    maxt = pm_min(ray.maxt, maxt);
    F3 combined = m.is_homogeneous ? volume_eval(cload(sc.volumes + m.sigma_t), ray.o) * m.scale : f3s(m.max_density);
    float mext = pick(combined, channel);
    float sampled_t = mint + (-pm_log(1.f - sample) / mext);
    bool valid_mi = active && (sampled_t <= maxt);
    mi.t = valid_mi ? sampled_t : pm_inf();
    mi.p = ray_at(ray, sampled_t);
    mi.mint = mint;
    mi.sigma_t = mi.sigma_s = f3s(0.f);
    if (m.is_homogeneous) {
        F3 st = volume_eval(cload(sc.volumes + m.sigma_t), mi.p) * m.scale;
        mi.sigma_t = st;
        if (want_albedo) mi.sigma_s = st * volume_eval(cload(sc.volumes + m.albedo), mi.p);
    } else if (valid_mi) {
        const DVolume vs = cload(sc.volumes + m.sigma_t), va = cload(sc.volumes + m.albedo);
        if (m.shared_grid && m.grey && vs.filter == MTS_FILTER_TRILINEAR) {
            // both grids share one cell / one set of weights; single channel: one value serves the three channels
            GridCell c = grid_cell(vs, mi.p);
            float st = m.scale * grid_fetch1(vs.data, c);
            mi.sigma_t = f3s(st);
            if (want_albedo) mi.sigma_s = f3s(st * grid_fetch1(va.data, c));       // the tracking walks never read sigma_s
        } else {
            F3 st = m.scale * volume_eval(vs, mi.p);
            mi.sigma_t = st;
            if (want_albedo) mi.sigma_s = st * volume_eval(va, mi.p);
        }
        if (COUNT) cnt.n_lookup++;
    }
    mi.combined = combined;
    mi.info = (m.is_homogeneous ? MI_HOMOGENEOUS : 0u) | (m.has_spectral_extinction ? MI_SPECTRAL : 0u) |
              (m.sample_emitters ? MI_SAMPLE_EMITTERS : 0u) | (m.grey ? MI_GREY : 0u) | ((uint32_t) m.phase << MI_PHASE_SHIFT);
    return mi;
}

// exp(-t * combined) per channel (medium.cpp:84); grey media evaluate it once
DEV F3 transmittance_exp_g(float t, F3 combined, bool grey) {
    if (grey) return f3s(pm_exp(-t * combined.x));
    return transmittance_exp(t, combined);
}

// Film splat of one finished sample: librender/integrator.cpp:265-285 + librender/imageblock.cpp:79-172
DEV void splat_sample(const DScene &sc, const DBlock &blk, uint32_t lx, uint32_t ly, F2 position_sample, F3 L, bool valid,
                      float *__restrict__ film, float acc[5]) {
    const DSensor &se = sc.sensor;
    float v[5];                                                 // srgb_to_xyz, core/spectrum.h:221-227
    v[0] = pm_fma(0.180423f, L.z, pm_fma(0.357580f, L.y, 0.412453f * L.x));
    v[1] = pm_fma(0.072169f, L.z, pm_fma(0.715160f, L.y, 0.212671f * L.x));
    v[2] = pm_fma(0.950227f, L.z, pm_fma(0.119193f, L.y, 0.019334f * L.x));
    v[3] = valid ? 1.f : 0.f;
    v[4] = 1.f;
    bool ok = true;                                             // imageblock.cpp:85-109: invalid samples are dropped
    for (int k = 0; k < 5; ++k) ok = ok && v[k] >= -1e-5f && pm_isfinite(v[k]);
    if (!ok) return;
    const DRFilter &rf = se.rfilter;
    const int border = rf.border_size;
    const int sx = blk.sx + 2 * border, sy = blk.sy + 2 * border;
    float posx = position_sample.x - ((float) (blk.ox - border) + .5f), posy = position_sample.y - ((float) (blk.oy - border) + .5f);
    if (rf.radius > 0.5f + MTS_RAY_EPSILON) {
        int lox = max((int) pm_ceil(posx - rf.radius), 0), loy = max((int) pm_ceil(posy - rf.radius), 0);
        int hix = min((int) pm_floor(posx + rf.radius), sx - 1), hiy = min((int) pm_floor(posy + rf.radius), sy - 1);
        uint32_t n = (uint32_t) pm_ceil((rf.radius - 2.f * MTS_RAY_EPSILON) * 2.f);
        float basex = (float) lox - posx, basey = (float) loy - posy;
        for (uint32_t yr = 0; yr < n; ++yr) {
            int y = loy + (int) yr;
            if (y > hiy) break;
            float wy = rf.values[min((int) pm_abs((basey + (float) yr) * rf.scale_factor), 31)];     // eval_discretized, core/rfilter.h:62-65
            int fy = blk.oy - border + y - se.crop_y;
            for (uint32_t xr = 0; xr < n; ++xr) {
                int x = lox + (int) xr;
                if (x > hix) break;
                float wx = rf.values[min((int) pm_abs((basex + (float) xr) * rf.scale_factor), 31)];
                float weight = wy * wx;
                int fx = blk.ox - border + x - se.crop_x;
                if (fx >= 0 && fy >= 0 && fx < se.crop_w && fy < se.crop_h) {                         // film clipping, imageblock.cpp:49-77
                    float *dst = film + 5 * ((size_t) fy * se.crop_w + fx);
                    for (int k = 0; k < 5; ++k) atomicAdd(dst + k, v[k] * weight);
                }
            }
        }
    } else {
        int lox = (int) pm_ceil(posx - .5f), loy = (int) pm_ceil(posy - .5f);
        if (lox == (int) lx && loy == (int) ly) {
            for (int k = 0; k < 5; ++k) acc[k] += v[k];
        } else if (lox >= 0 && loy >= 0 && lox < sx && loy < sy) {
            float *dst = film + 5 * ((size_t) (blk.oy + loy - se.crop_y) * se.crop_w + (blk.ox + lox - se.crop_x));
            for (int k = 0; k < 5; ++k) atomicAdd(dst + k, v[k]);
        }
    }
}

// Cold per-lane state parked in LDS (struct-of-arrays over the 256 lanes of the workgroup, conflict-free):
// values that are touched once per sample or once per NEE / direct-light walk, so they do not have to
// occupy VGPRs during the thousands of tracking steps in between.
enum { C_ACC = 0,            // 5: film accumulators X, Y, Z, A, W of this pixel
       C_POS = 5,            // 2: film position of the current sample
       C_RAYW = 7,           // 1: sensor ray weight
       C_SO = 8, C_SD = 11,  // 3 + 3: parked main-path origin / direction
       C_SHIT = 14,          // 8: parked main-path hit (t, p, uv, shape, prim)
       C_SMED = 22,          // 1: parked medium id
       C_CW = 23,            // 3: pending NEE weight
       C_EMIT = 26,          // 3: emitter value of the NEE sample
       C_COUNT = 29 };
struct ColdState {
    float *base;             // &lds[0][lane]
    DEV float &f(int k) const { return base[k * 256]; }
    DEV void put3(int k, F3 v) const { f(k) = v.x; f(k + 1) = v.y; f(k + 2) = v.z; }
    DEV F3 get3(int k) const { return f3(f(k), f(k + 1), f(k + 2)); }
    DEV void put_hit(const Hit &h) const {
        f(C_SHIT) = h.t; put3(C_SHIT + 1, h.p); f(C_SHIT + 4) = h.uv.x; f(C_SHIT + 5) = h.uv.y;
        f(C_SHIT + 6) = __int_as_float(h.shape); f(C_SHIT + 7) = __int_as_float(h.prim);
    }
    DEV Hit get_hit() const {
        Hit h; h.t = f(C_SHIT); h.p = get3(C_SHIT + 1); h.uv.x = f(C_SHIT + 4); h.uv.y = f(C_SHIT + 5);
        h.shape = __float_as_int(f(C_SHIT + 6)); h.prim = __float_as_int(f(C_SHIT + 7)); return h;
    }
};

// All samples of one pixel (librender/integrator.cpp:197-209 + :233-288 + integrators/volpath.cpp)
template <bool COUNT>
DEV void volpath_pixel_flat(const DScene &sc, Pcg32 &rng, const DBlock &blk, uint32_t lx, uint32_t ly, uint32_t sample_count,
                            float *__restrict__ film, const ColdState cold, Counters &cnt) {
    const DSensor &se = sc.sensor;
    const uint32_t max_depth = (uint32_t) sc.integrator.max_depth, rr_depth = (uint32_t) sc.integrator.rr_depth;
    const bool hide_emitters = sc.integrator.hide_emitters != 0;

    // ---- hot per-lane state (registers)
    DRay ray;                       // the ray being tracked now (main path, or the NEE / direct-light walk)
    Hit si;                         // cached closest hit of `ray`
    int medium;                     // medium containing ray.o
    F3 thr, res; float eta; uint32_t depth, channel;         // main path (volpath.cpp:54-67)
    F3 trans; float wa, wb;         // walk: transmittance; NEE: wa = total_dist, wb = ds.dist; direct: wb = bs.pdf
    uint32_t st, mode, flags, sample_idx = 0;

    // A freshly spawned ray that cannot reach the scene's bounding box is resolved on the spot (this is the
    // first test of ShapeKDTree::ray_intersect_scalar, kdtree.h:2095-2098); everything else queues for INTERSECT.
    auto queue_intersection = [&]() {
        float bmint, bmaxt;
        bbox_ray_intersect(sc.bbox, ray, bmint, bmaxt);
        si.t = pm_inf(); si.shape = -1;
        if (pm_max(ray.mint, bmint) <= bmaxt) flags |= FL_NEEDS_INT; else flags &= ~FL_NEEDS_INT;
    };
    auto begin_sample = [&]() {                                // integrator.cpp:242-264, volpath.cpp:48-71
        const float px = (float) (lx + (uint32_t) blk.ox), py = (float) (ly + (uint32_t) blk.oy);
        F2 u = rng.next_2d();
        F2 position_sample; position_sample.x = px + u.x; position_sample.y = py + u.y;
        F2 aperture_sample; aperture_sample.x = .5f; aperture_sample.y = .5f;
        if (se.needs_aperture_sample) aperture_sample = rng.next_2d();
        (void) rng.next_1d();                                  // wavelength sample, unused in rgb
        F2 adjusted;
        adjusted.x = (position_sample.x - (float) se.crop_x) / (float) se.crop_w;
        adjusted.y = (position_sample.y - (float) se.crop_y) / (float) se.crop_h;
        F3 rw;
        ray = sensor_sample_ray(sc, adjusted, aperture_sample, rw);
        cold.f(C_POS) = position_sample.x; cold.f(C_POS + 1) = position_sample.y; cold.f(C_RAYW) = rw.x;   // all supported sensors: grey weight
        medium = se.medium;
        thr = f3s(1.f); res = f3s(0.f); eta = 1.f; depth = 0;
        channel = (uint32_t) pm_min(rng.next_1d() * 3.f, 2.f);
        si.p = f3s(0.f); si.uv.x = si.uv.y = 0.f; si.prim = 0;
        flags = FL_ALIVE | ((!hide_emitters && sc.environment >= 0) ? FL_VALID_RAY : 0u) | (!hide_emitters ? FL_SPEC_CHAIN : 0u);
        queue_intersection();
        mode = M_MAIN; st = S_TOP;
    };
    // NEE walk finished (volpath.cpp:366 + :165-166 / :211): add the contribution, resume the main path
    auto end_nee = [&]() {
        F3 emitted = trans * cold.get3(C_EMIT);
        res = res + cold.get3(C_CW) * emitted;
        mode = M_MAIN; medium = __float_as_int(cold.f(C_SMED)); ray.d = cold.get3(C_SD);
        if (flags & FL_FROM_MEDIUM) { ray.o = cold.get3(C_SO); st = S_PHASE; }
        else { si = cold.get_hit(); st = S_BSDF; }
    };
    // direct-light walk finished (volpath.cpp:464 + :246-252): MIS-weighted emitter hit, resume the main path
    auto end_direct = [&](F3 emitter_val, float emitter_pdf) {
        F3 emitted = trans * emitter_val;
        if (emitter_pdf != 0.f) res = res + mis_weight(wb, emitter_pdf) * thr * emitted;
        ray = spawn_ray(cold.get3(C_SO), cold.get3(C_SD));
        si = cold.get_hit(); medium = __float_as_int(cold.f(C_SMED));
        flags = (flags & ~FL_NEEDS_INT) | FL_ALIVE;
        mode = M_MAIN; st = S_TOP;
    };

    for (int k = 0; k < 5; ++k) cold.f(C_ACC + k) = 0.f;
    begin_sample();

    enum { B_INT = 0, B_MED, B_SURF, B_PHASE, B_NEW, B_COUNT };
#if defined(MTSAMD_BLOCKSTATS)
    long long bs_t0 = clock64(); int bs_prev_sel = 7;
    unsigned long long bs_loc[24] = {};
#endif
    while (__ballot(st != S_DONE)) {
#if defined(MTSAMD_BLOCKSTATS)
        if (COUNT) { long long t = clock64(); bs_loc[16 + bs_prev_sel] += (unsigned long long) (t - bs_t0); bs_t0 = t; bs_prev_sel = 6; }
#endif
        // ================================================================= TOP: loop heads (cheap, every trip)
        if (st == S_TOP) {
            if (mode == M_MAIN) {                              // volpath.cpp:79-87
                bool active = (flags & FL_ALIVE) && any_nonzero(thr);
                float q = pm_min(hmax(thr) * (eta * eta), .95f);
                bool perform_rr = depth > rr_depth;
                active = active && (rng.next_1d() < q || !perform_rr);
                if (perform_rr) thr = thr * pm_rcp(q);
                if (!active || depth >= max_depth) st = S_NEW;
                else {
                    if (COUNT) cnt.n_iter++;
                    st = medium >= 0 ? S_MED : S_SURF;
                }
            } else if (mode == M_NEE) {                        // volpath.cpp:283-287
                float remaining_dist = wb * (1.f - MTS_SHADOW_EPSILON) - wa;
                ray.maxt = remaining_dist;
                if (!(remaining_dist > 0.f)) end_nee();
                else { if (COUNT) cnt.n_nee_step++; st = medium >= 0 ? S_MED : S_SURF; }
            } else {                                           // volpath.cpp:385-388
                if (COUNT) cnt.n_nee_step++;
                st = medium >= 0 ? S_MED : S_SURF;
            }
        }
        // ================================================================= census + vote
        const bool want_int = (st == S_MED || st == S_SURF || st == S_DIRB) && (flags & FL_NEEDS_INT);
        int votes[B_COUNT];
        votes[B_INT] = __popcll(__ballot(want_int));
        votes[B_MED] = __popcll(__ballot(st == S_MED && !want_int));
        votes[B_SURF] = __popcll(__ballot((st == S_SURF && !want_int) || st == S_BSDF));
        votes[B_PHASE] = __popcll(__ballot(st == S_PHASE));
        votes[B_NEW] = __popcll(__ballot(st == S_NEW));
        int sel = B_MED, best = votes[B_MED];
        for (int b = 0; b < B_COUNT; ++b) if (votes[b] > best) { best = votes[b]; sel = b; }
        if (best == 0) continue;                               // only S_TOP / S_DONE lanes: next trip dispatches them
#if defined(MTSAMD_BLOCKSTATS)                                  // diagnostic build: executions and lanes served per block
        if (COUNT) {
            bs_loc[2 * sel] += 1ull; bs_loc[2 * sel + 1] += (unsigned long long) best;
            long long t = clock64(); bs_loc[16 + 6] += (unsigned long long) (t - bs_t0); bs_t0 = t; bs_prev_sel = sel;
        }
#endif
        // ================================================================= NEW: finish a sample, start the next (integrator.cpp:265-288)
        if (sel == B_NEW && st == S_NEW) {
            float acc[5];
            for (int k = 0; k < 5; ++k) acc[k] = cold.f(C_ACC + k);
            F2 position_sample; position_sample.x = cold.f(C_POS); position_sample.y = cold.f(C_POS + 1);
            splat_sample(sc, blk, lx, ly, position_sample, f3s(cold.f(C_RAYW)) * res, (flags & FL_VALID_RAY) != 0, film, acc);
            for (int k = 0; k < 5; ++k) cold.f(C_ACC + k) = acc[k];
            if (++sample_idx == sample_count) st = S_DONE;
            else begin_sample();
        }
        // ================================================================= INTERSECT (volpath.cpp:109,182,241,298,339,395,425)
        if (sel == B_INT && want_int) {
            si = ray_intersect(sc, ray);
            flags &= ~FL_NEEDS_INT;
        }
        if (st == S_DIRB && !(flags & FL_NEEDS_INT)) {         // volpath.cpp:239-245: start the direct-light walk on a copy
            cold.put3(C_SO, ray.o); cold.put3(C_SD, ray.d); cold.put_hit(si);
            trans = f3s(1.f);
            mode = M_DIR; st = S_TOP;
        }
        // ================================================================= MEDIUM: one free-flight step
        if (sel == B_MED && st == S_MED && !want_int) {
            const float u = rng.next_1d();                     // volpath.cpp:105 / :294 / :391
            MedStep mi;
#if defined(EXP_NOWF)
            mi = medium_step<COUNT>(sc, cload(sc.media), ray, u, channel, mode == M_MAIN, cnt);
#else
            WATERFALL_BEGIN(medium, mu)
                mi = medium_step<COUNT>(sc, cload(sc.media + mu), ray, u, channel, mode == M_MAIN, cnt);
            WATERFALL_END
#endif
            if (si.t < mi.t) mi.t = pm_inf();                  // volpath.cpp:112 / :300 / :397
            const bool spectral = (mi.info & MI_SPECTRAL) != 0, homogeneous = (mi.info & MI_HOMOGENEOUS) != 0, grey = (mi.info & MI_GREY) != 0;
            const F3 sigma_n = homogeneous ? f3s(0.f) : mi.combined - mi.sigma_t;
            const bool valid = mi.t != pm_inf();
            if (mode == M_MAIN) {
                if (spectral) {                                // medium.cpp:77-89, volpath.cpp:113-117
                    float t = pm_min(mi.t, si.t) - mi.mint;
                    F3 tr = transmittance_exp_g(t, mi.combined, grey);
                    F3 free_flight_pdf = si.t < mi.t ? tr : tr * mi.combined;
                    float tr_pdf = pick(free_flight_pdf, channel);
                    thr = thr * (tr_pdf > 0.f ? tr / tr_pdf : f3s(0.f));
                }
                const float u2 = rng.next_1d();                // volpath.cpp:123 (drawn even when the medium was left)
                if (!valid) st = S_SURF;                       // escaped_medium: surface part of this iteration
                else {
                    bool null_scatter = u2 >= pick(mi.sigma_t, channel) / pick(mi.combined, channel);
                    if (null_scatter) {                        // volpath.cpp:128-131,140-144
                        if (spectral) thr = thr * (sigma_n * pick(mi.combined, channel) / pick(sigma_n, channel));
                        ray.o = mi.p; ray.mint = 0.f; si.t = si.t - mi.t;
                        st = S_TOP;                            // stays alive
                    } else {                                   // real scattering event, volpath.cpp:133-175
                        depth += 1;
                        if (!(depth < max_depth)) { flags &= ~FL_ALIVE; st = S_TOP; }
                        else {
                            if (spectral) thr = thr * (mi.sigma_s * pick(mi.combined, channel) / pick(mi.sigma_t, channel));
                            else thr = thr * (mi.sigma_s / mi.sigma_t);
                            const bool sample_emitters = (mi.info & MI_SAMPLE_EMITTERS) != 0;
                            flags |= FL_VALID_RAY;
                            flags = sample_emitters ? (flags & ~FL_SPEC_CHAIN) : (flags | FL_SPEC_CHAIN);
                            ray.o = mi.p;                      // scattering position; ray.d stays the incident direction
                            st = S_PHASE;
                            if (sample_emitters) {             // volpath.cpp:162-167 -> sample_emitter :261-281
                                F3 emitter_val;
                                DirSample ds = sample_emitter_direction(sc, mi.p, rng.next_2d(), false, emitter_val);
                                if (ds.pdf != 0.f) {
                                    float phase_val = phase_eval(sc, (int) (mi.info >> MI_PHASE_SHIFT), -ray.d, mi.p, ds.d);
                                    cold.put3(C_CW, thr * phase_val); cold.put3(C_EMIT, emitter_val);
                                    cold.put3(C_SO, mi.p); cold.put3(C_SD, ray.d); cold.f(C_SMED) = __int_as_float(medium);
                                    trans = f3s(1.f); wa = 0.f; wb = ds.dist;
                                    ray = spawn_ray(mi.p, ds.d); ray.mint = 0.f;
                                    queue_intersection();
                                    flags |= FL_FROM_MEDIUM;
                                    mode = M_NEE; st = S_TOP;
                                }
                            }
                        }
                    }
                }
            } else if (mode == M_NEE) {                        // volpath.cpp:303-334
                const float remaining_dist = ray.maxt;
                if (spectral) {
                    float t = pm_min(remaining_dist, pm_min(mi.t, si.t)) - mi.mint;
                    F3 tr = transmittance_exp_g(t, mi.combined, grey);
                    F3 free_flight_pdf = (si.t < mi.t || mi.t > remaining_dist) ? tr : tr * mi.combined;
                    float tr_pdf = pick(free_flight_pdf, channel);
                    trans = trans * (tr_pdf > 0.f ? tr / tr_pdf : f3s(0.f));
                }
                if (mi.t > remaining_dist && mi.t != pm_inf()) wa = wb;
                if (mi.t > remaining_dist) mi.t = pm_inf();
                if (mi.t == pm_inf()) st = S_SURF;             // escaped_medium
                else {
                    wa += mi.t;
                    ray.o = mi.p; ray.mint = 0.f; si.t = si.t - mi.t;
                    if (spectral) trans = trans * sigma_n; else trans = trans * (sigma_n / mi.combined);
                    if (any_nonzero(trans)) st = S_TOP; else end_nee();     // volpath.cpp:358
                }
            } else {                                           // direct-light walk, volpath.cpp:399-421
                if (spectral) {
                    float t = pm_min(mi.t, si.t) - mi.mint;
                    F3 tr = transmittance_exp_g(t, mi.combined, grey);
                    F3 free_flight_pdf = si.t < mi.t ? tr : tr * mi.combined;
                    float tr_pdf = pick(free_flight_pdf, channel);
                    trans = trans * (tr_pdf > 0.f ? tr / tr_pdf : f3s(0.f));
                }
                if (!valid) st = S_SURF;
                else {
                    ray.o = mi.p; ray.mint = 0.f; si.t = si.t - mi.t;
                    if (spectral) trans = trans * sigma_n; else trans = trans * (sigma_n / mi.combined);
                    if (any_nonzero(trans)) st = S_TOP; else end_direct(f3s(0.f), 0.f);   // volpath.cpp:456
                }
            }
        }
        // ================================================================= SURFACE step of a walk (cheap; runs in whatever trip produced it)
        if (st == S_SURF && mode != M_MAIN && !(flags & FL_NEEDS_INT) && (sel == B_INT || sel == B_MED || sel == B_SURF)) {
            const bool hit = hit_valid(si);
            if (mode == M_NEE) {                               // volpath.cpp:336-364
                wa += si.t;
                if (hit) {
                    F3 nt = f3s(0.f), n = f3s(0.f); int is_tr = 0, ext = -1, inte = -1;
                    WATERFALL_BEGIN(si.shape, su)
                        const DShape s = cload(sc.shapes + su);
                        nt = cload(sc.bsdfs + s.bsdf).type == MTS_BSDF_NULL ? f3s(1.f) : f3s(0.f);    // null.cpp:70-73, bsdf.cpp:11-14
                        is_tr = s.is_medium_transition; ext = s.exterior; inte = s.interior;
                        if (is_tr) n = hit_geo_normal(sc, s, si);
                    WATERFALL_END
                    trans = trans * nt;
                    ray = spawn_ray(si.p, ray.d);
                    queue_intersection();
                    if (is_tr) medium = dot(ray.d, n) > 0 ? ext : inte;     // interaction.h:178-200
                }
                if (hit && any_nonzero(trans)) st = S_TOP; else end_nee();
            } else {                                           // direct-light walk, volpath.cpp:423-462
                int emitter = sc.environment;
                Surf sf; sf.wi = -ray.d; sf.sh.n = f3s(0.f); sf.n = f3s(0.f);
                F3 nt = f3s(0.f); int is_tr = 0, ext = -1, inte = -1;
                if (hit) {
                    WATERFALL_BEGIN(si.shape, su)
                        const DShape s = cload(sc.shapes + su);
                        emitter = s.emitter;
                        nt = cload(sc.bsdfs + s.bsdf).type == MTS_BSDF_NULL ? f3s(1.f) : f3s(0.f);
                        is_tr = s.is_medium_transition; ext = s.exterior; inte = s.interior;
                        if (emitter >= 0 || is_tr) complete_surface(sc, s, si, ray.d, sf);
                    WATERFALL_END
                }
                if (emitter >= 0) {                            // volpath.cpp:430-440
                    const F3 ref_p = cold.get3(C_SO);
                    DirSample ds;                              // render/records.h:168-174
                    ds.p = si.p; ds.n = sf.sh.n; ds.d = si.p - ref_p; ds.dist = norm(ds.d); ds.d = ds.d / ds.dist;
                    if (!hit) ds.d = -sf.wi;
                    ds.emitter = emitter; ds.pdf = 0.f; ds.delta = false;
                    end_direct(emitter_eval(sc, emitter, sf.wi.z), pdf_emitter_direction(sc, ref_p, ds));
                } else {
                    if (hit) {
                        trans = trans * nt;
                        ray = spawn_ray(si.p, ray.d);
                        queue_intersection();
                        if (is_tr) medium = dot(ray.d, sf.n) > 0 ? ext : inte;
                    }
                    if (hit && any_nonzero(trans)) st = S_TOP; else end_direct(f3s(0.f), 0.f);
                }
            }
        }
        // ================================================================= SURFACE interaction of the main path (volpath.cpp:184-212)
        if (sel == B_SURF && st == S_SURF && mode == M_MAIN && !want_int) {
            const bool hit = hit_valid(si);
            Surf sf; sf.wi = -ray.d; sf.n = f3s(0.f); sf.sh.s = sf.sh.t = sf.sh.n = f3s(0.f);
            int emitter = sc.environment, bsdf_id = 0;
            if (hit) {
                WATERFALL_BEGIN(si.shape, su)
                    const DShape s = cload(sc.shapes + su);
                    complete_surface(sc, s, si, ray.d, sf);
                    emitter = s.emitter; bsdf_id = s.bsdf;
                WATERFALL_END
            }
            if ((flags & FL_SPEC_CHAIN) && emitter >= 0) res = res + thr * emitter_eval(sc, emitter, sf.wi.z);
            if (!hit) { flags &= ~FL_ALIVE; st = S_TOP; }
            else {
                st = S_BSDF;
                const DBsdf &bsdf = sc.bsdfs[bsdf_id];
                bool active_e = (bsdf.flags & F_Smooth) != 0 && (depth + 1 < max_depth);
                if (active_e) {                                // volpath.cpp:200-212 -> sample_emitter :261-281
                    F3 emitter_val;
                    DirSample ds = sample_emitter_direction(sc, si.p, rng.next_2d(), false, emitter_val);
                    if (ds.pdf != 0.f) {
                        F3 wo = to_local(sf.sh, ds.d);
                        F3 bsdf_val = bsdf_eval(bsdf, sf.wi, wo);
                        float bpdf = bsdf_pdf(bsdf, sf.wi, wo);
                        cold.put3(C_CW, thr * bsdf_val * mis_weight(ds.pdf, ds.delta ? 0.f : bpdf)); cold.put3(C_EMIT, emitter_val);
                        cold.put_hit(si); cold.put3(C_SD, ray.d); cold.f(C_SMED) = __int_as_float(medium);
                        trans = f3s(1.f); wa = 0.f; wb = ds.dist;
                        ray = spawn_ray(si.p, ds.d);
                        queue_intersection();
                        flags &= ~FL_FROM_MEDIUM;
                        mode = M_NEE; st = S_TOP;
                    }
                }
            }
        }
        // ================================================================= BSDF sampling (volpath.cpp:214-252)
        if (sel == B_SURF && st == S_BSDF) {
            Surf sf; int bsdf_id = 0, is_tr = 0, ext = -1, inte = -1;
            WATERFALL_BEGIN(si.shape, su)
                const DShape s = cload(sc.shapes + su);
                complete_surface(sc, s, si, ray.d, sf);
                bsdf_id = s.bsdf; is_tr = s.is_medium_transition; ext = s.exterior; inte = s.interior;
            WATERFALL_END
            const float s1 = rng.next_1d(); const F2 s2 = rng.next_2d(); (void) s1;
            BSDFSample bs;
            F3 bsdf_val = bsdf_sample(sc.bsdfs[bsdf_id], sf.wi, s2, bs);
            thr = thr * bsdf_val;
            eta *= bs.eta;
            ray = spawn_ray(si.p, to_world(sf.sh, bs.wo));
            flags |= FL_ALIVE;
            const bool non_null_bsdf = !(bs.sampled_type & F_Null);
            if (non_null_bsdf) { depth += 1; flags |= FL_VALID_RAY; }
            if (non_null_bsdf && (bs.sampled_type & F_Delta)) flags |= FL_SPEC_CHAIN;
            if (bs.sampled_type & F_Smooth) flags &= ~FL_SPEC_CHAIN;
            const bool add_emitter = !(bs.sampled_type & F_Delta) && any_nonzero(thr) && (depth < max_depth);
            const int new_medium = is_tr ? (dot(ray.d, sf.n) > 0 ? ext : inte) : medium;     // volpath.cpp:249-250
            queue_intersection();
            if (add_emitter) { cold.f(C_SMED) = __int_as_float(new_medium); wb = bs.pdf; st = S_DIRB; }   // the walk runs in the old medium
            else { medium = new_medium; st = S_TOP; }
        }
        // ================================================================= PHASE sampling (volpath.cpp:169-175)
        if (st == S_PHASE && (sel == B_INT || sel == B_MED || sel == B_PHASE)) {
            const float s1 = rng.next_1d(); const F2 s2 = rng.next_2d();      // left-to-right (SURVEY.md 8(a'))
            F3 wo;
            WATERFALL_BEGIN(medium, mu)
                wo = phase_sample(sc, cload(sc.media + mu).phase, make_frame(ray.d), ray.o, s1, s2);   // mi.sh_frame = Frame3f(ray.d), medium.cpp:42
            WATERFALL_END
            ray = spawn_ray(ray.o, wo); ray.mint = 0.0f;
            queue_intersection();
            flags |= FL_ALIVE;
            st = S_TOP;
        }
    }
#if defined(MTSAMD_BLOCKSTATS)
    if (COUNT && __builtin_amdgcn_readfirstlane((int) (threadIdx.x & 63)) == (int) (threadIdx.x & 63))
        for (int k = 0; k < 24; ++k) atomicAdd(&g_blockstats[k], bs_loc[k]);
#endif
    // the pixel's own film entry (block accumulation, imageblock.cpp:163-168) -> film (hdrfilm.cpp:207-211)
    float *dst = film + 5 * ((size_t) (blk.oy + (int) ly - se.crop_y) * se.crop_w + (blk.ox + (int) lx - se.crop_x));
    for (int k = 0; k < 5; ++k) atomicAdd(dst + k, cold.f(C_ACC + k));
}

} // namespace mtsamd
