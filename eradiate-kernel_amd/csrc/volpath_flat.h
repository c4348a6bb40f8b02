// volpath_flat.h -- volpath (integrators/volpath.cpp:38-465) as ONE flat state machine over explicit path state.
//
// Why: the reference nests three loops (main loop :72, NEE ratio tracking :282, direct-light walk :385).
// Compiled as written, a wave serialises them: while a few lanes run an inner tracking loop to
// completion the other lanes of the wave idle.  Here every path carries (mode, state); the work is cut
// into small blocks (INTERSECT, MEDIUM step, SCATTER, walk SURFACE step, main SURFACE + BSDF, PHASE, NEW
// sample) and a path advances by one block at a time -- whichever of the three reference loops that block
// belongs to -- so the expensive block (free-flight sample + grid gathers) is shared by all of them.
//
// Two drivers schedule the blocks:
//   1. volpath_pixel_flat: one lane = one pixel, state in registers, per-wave census (__ballot/__popcll)
//      and a vote for the block most lanes wait for (measured: ~40 % of the lanes served per block).
//   2. volpath_workgroup_async: the hot state of a workgroup's paths lives in LDS (struct of arrays), one LDS
//      ring of path ids per block class; a wave claims 64 paths that wait for the same block, runs it with
//      every lane active, and appends the paths to the rings of their next blocks.  No barriers, no sort.
//      (A barrier-synchronised counting-sort driver was measured at half the speed and removed.)
//
// The random draws happen in exactly the order of the scalar_rgb variant (SURVEY.md 8(a')); results are
// bit-identical to the nested formulation in integrator_dev.h and to the CPU restatement.
// Citations are relative to /root/reference.
#pragma once
#include "integrator_dev.h"

namespace mtsamd {
inline namespace MTS_VARIANT_NS {

// The machine is written once for both builds (MTS_SPEC_N = 3: rgb / mono, 4: spectral).  In the rgb build `Spec` is F3, the
// wavelength context is an empty struct and the macros below vanish, so its kernels are compiled from the same expressions as ever.
#if MTS_SPEC_N == 3
#define MTS_CX
#define MTS_CXI(id)
#define MTS_SPEC_DW 3
#else
#define MTS_CX , cx
#define MTS_CXI(id) , cx, (id)
#define MTS_SPEC_DW 4
#endif

#if MTS_SPEC_N == 3
#define MTS_FILM_STRIDE(sc) ((size_t) 5)
#else
#define MTS_FILM_STRIDE(sc) ((size_t) (sc).film_channels)          // X, Y, Z, A, W (+ two AOV channels per spectral bin: nbins / bins)
#endif
// Blocks that run a path's NEXT block in the same visit (bit mask; wg_block).  A visit costs a ring push, a claim and an LDS round trip
// of the state, but what runs in place runs on the lanes that want it only.  Measured one by one (round 3, profiles/r03_ab_experiments.log;
// C3 512 x 512 x 256 / C4 1024 x 1024 x 64, Msamples/s against 547 / 267 without; all bit-identical):
//   1  walk SURFACE twice: a walk that leaves through a boundary face with nothing behind it (queue_intersection: the new ray
//      misses the scene's box, no INTERSECT visit) comes straight back to this class, only to end.  569 / 268: ON.
//   2  ... and then the PHASE sample of the main path the ended walk hands back to: 567 / 258 together with 1.
//   4  the walk's MEDIUM block runs the SURFACE step(s) of the lanes whose walk left the medium: 540 / 229 together with 1 and 2.
//   8  the main path's MEDIUM block runs SCATTER (emitter sample, start of the walk) for its real collisions: 564 / 250 with 1.
//  16  main-path SURFACE twice (a ray that leaves through a boundary face into nothing ends the sample): 564 / 267 with 1.
// Only the first pays: its second round is a handful of instructions, every other candidate runs real work on a thinned wave.
#ifndef MTS_CHAIN
#define MTS_CHAIN 1
#endif
#ifndef MTS_REPEAT_MIN_W
#define MTS_REPEAT_MIN_W 32        // MTS_REPEAT_MIN for the walks' class: 16 and 48 measured 539 / 265 and 536 / 265
#endif
#define MTS_REPEAT_MIN 32          // lanes that must stay in a MEDIUM class for the block to run again in place (wg_block)

enum : uint32_t { S_TOP = 0, S_MED = 1, S_SURF = 2, S_PHASE = 3, S_BSDF = 4, S_DIRB = 5, S_NEW = 6, S_DONE = 7, S_SCATTER = 8,
                  S_ENDNEE = 9, S_ENDDIR0 = 10 };   // transient (workgroup drivers): end_nee / end_direct(0, 0) still to run on the full state
enum : uint32_t { M_MAIN = 0, M_NEE = 1, M_DIR = 2 };
enum : uint32_t { FL_ALIVE = 1, FL_VALID_RAY = 2, FL_SPEC_CHAIN = 4, FL_NEEDS_INT = 8, FL_FROM_MEDIUM = 16 };

// Result of one free-flight sample (librender/medium.cpp:34-75); sigma_n is derived by the caller
// (heterogeneous.cpp:46: combined - sigma_t, homogeneous.cpp:44: 0).
struct MedStep { float t, mint; F3 p; Spec sigma_t, sigma_s, combined; uint32_t info; float inv_combined /* DMedium::inv_max_density */; };

// x / d for a divisor whose correctly rounded reciprocal rd = RN(1 / d) is at hand (the majorant of a heterogeneous medium: a constant
// of the medium record): q = x rd, two Markstein corrections q += RN(x - q d) rd with the remainders exact by fma.  The first makes q
// faithful, the second correctly rounded -- the IEEE quotient, bit for bit (the host excludes divisors with an all-ones significand
// and exponents near the ends of the range, scene_host.cpp; checked against x / d on 4 * 10^10 quotients, tests/test_pmath.py has the
// sampled version).  Five multiply-adds instead of v_div_scale x 2, v_rcp, six fma, v_div_fmas, v_div_fixup.  rd == 0: plain division.
// The sign of a zero quotient is the dividend's (d > 0).
DEV float div_by_invariant(float x, float d, float rd) {
    if (rd == 0.f) return x / d;
    float q = x * rd;
    float r = pm_fma(-q, d, x);
    q = pm_fma(r, rd, q);
    r = pm_fma(-q, d, x);
    q = pm_fma(r, rd, q);
    return pm_from_bits(pm_bits(q) | (pm_bits(x) & 0x80000000u));
}
DEV Spec div_by_invariant(Spec x, Spec d, float rd) {          // every channel of d holds the same value when rd != 0
    if (rd == 0.f) return x / d;
#if MTS_SPEC_N == 3
    return f3(div_by_invariant(x.x, d.x, rd), div_by_invariant(x.y, d.x, rd), div_by_invariant(x.z, d.x, rd));
#else
    return spec4(div_by_invariant(x.x, d.x, rd), div_by_invariant(x.y, d.x, rd), div_by_invariant(x.z, d.x, rd), div_by_invariant(x.w, d.x, rd));
#endif
}
enum : uint32_t { MI_HOMOGENEOUS = 1, MI_SPECTRAL = 2, MI_SAMPLE_EMITTERS = 4, MI_GREY = 8, MI_PHASE_SHIFT = 8 + 8 };

#if defined(MTSAMD_BLOCKSTATS)
__device__ unsigned long long g_blockstats[48];    // class b < 12: [b] executions, [12+b] lanes served, [24+b] cycles; [36..41] MEDIUM segments, [42] idle, [43] push, [44] claim / vote
#endif

// textures/grid3d.cpp:259-341 split in two: cell coordinates / weights (shared by grids with the same
// transform and resolution) and the 8 gathers + trilinear blend of one grid.
struct GridCell { int32_t r00, r10, r01, r11, x0, x1; F3 w0, w1; };
DEV GridCell grid_cell_clamp(const float *w2l, int affine, int nx, int ny, int nz, int sx /* voxels per stored row */, F3 p_world) {      // clamp mode (pair grids)
    F3 p = affine ? mat_point_affine(w2l, p_world) : mat_point(w2l, p_world);
    p = f3(pm_fma(p.x, (float) nx, -.5f), pm_fma(p.y, (float) ny, -.5f), pm_fma(p.z, (float) nz, -.5f));
    int ix = (int) pm_floor(p.x), iy = (int) pm_floor(p.y), iz = (int) pm_floor(p.z);
    GridCell c;
    c.w1 = p - f3((float) ix, (float) iy, (float) iz); c.w0 = f3(1.f - c.w1.x, 1.f - c.w1.y, 1.f - c.w1.z);
    c.x0 = min(max(ix, 0), nx - 1); c.x1 = min(max(ix + 1, 0), nx - 1);                                  // grid3d.cpp:234-250, clamp
    int y0 = min(max(iy, 0), ny - 1), y1 = min(max(iy + 1, 0), ny - 1), z0 = min(max(iz, 0), nz - 1), z1 = min(max(iz + 1, 0), nz - 1);
    c.r00 = (z0 * ny + y0) * sx; c.r10 = (z0 * ny + y1) * sx; c.r01 = (z1 * ny + y0) * sx; c.r11 = (z1 * ny + y1) * sx;
    return c;
}
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
DEV float grid_fetch1(const MTS_GLOBAL_AS float *__restrict__ D, const GridCell &c) {
#if defined(EXP_NOGATHER)
    return trilerp(0.5f, 0.6f, 0.7f, 0.8f, 0.9f, 1.0f, 1.1f, 1.2f, c.w0, c.w1) + 1e-9f * (float) (c.r00 + c.x0 + c.r11 + c.x1 + c.r01 + c.r10);
#endif
    return trilerp(D[c.r00 + c.x0], D[c.r00 + c.x1], D[c.r10 + c.x0], D[c.r10 + c.x1],
                   D[c.r01 + c.x0], D[c.r01 + c.x1], D[c.r11 + c.x0], D[c.r11 + c.x1], c.w0, c.w1);
}

// Both grids of a medium from the interleaved copy (DMedium::pair_grid): one 16-byte gather per (z, y) row covers the two
// x-neighbours of sigma_t and albedo.  Clamp mode only: x1 is x0 + 1 except on the last column, where both are nx - 1.
// `nx` here is the stored row length: a one-column grid (the 1-D atmospheres: nz x 1 x 1) is stored two voxels wide.
typedef float mts_float4 __attribute__((ext_vector_type(4)));
typedef mts_float4 __attribute__((aligned(8))) mts_float4_a8;
// columns_equal (wave-uniform): a profile in z only -- the two rows of a z-level hold the same voxels, one gather serves both.
DEV void grid_fetch_pair(const MTS_GLOBAL_AS float *__restrict__ P, const GridCell &c, int nx, bool columns_equal, float &sigma_t, float &albedo) {
    const int xb = min(c.x0, nx - 2);
    const bool hi0 = c.x0 != xb, hi1 = c.x1 != xb;           // take the second voxel of the pair
#if defined(EXP_GATHER_LOCAL)                                // measurement only (breaks parity): every gather inside the first 4 KiB of the grid, i.e.
#define MTS_GL(i) ((i) & 511)                                // L1-resident -- how much of a tracking step is the L2 latency of its lookups?
    mts_float4 q00 = *(const MTS_GLOBAL_AS mts_float4_a8 *) (P + 2 * MTS_GL(c.r00 + xb)), q01 = *(const MTS_GLOBAL_AS mts_float4_a8 *) (P + 2 * MTS_GL(c.r01 + xb)), q10 = q00, q11 = q01;
    if (!columns_equal) { q10 = *(const MTS_GLOBAL_AS mts_float4_a8 *) (P + 2 * MTS_GL(c.r10 + xb)); q11 = *(const MTS_GLOBAL_AS mts_float4_a8 *) (P + 2 * MTS_GL(c.r11 + xb)); }
#else
    mts_float4 q00 = *(const MTS_GLOBAL_AS mts_float4_a8 *) (P + 2 * (c.r00 + xb)), q01 = *(const MTS_GLOBAL_AS mts_float4_a8 *) (P + 2 * (c.r01 + xb)), q10 = q00, q11 = q01;
    if (!columns_equal) { q10 = *(const MTS_GLOBAL_AS mts_float4_a8 *) (P + 2 * (c.r10 + xb)); q11 = *(const MTS_GLOBAL_AS mts_float4_a8 *) (P + 2 * (c.r11 + xb)); }
#endif
    sigma_t = trilerp(hi0 ? q00.z : q00.x, hi1 ? q00.z : q00.x, hi0 ? q10.z : q10.x, hi1 ? q10.z : q10.x,
                      hi0 ? q01.z : q01.x, hi1 ? q01.z : q01.x, hi0 ? q11.z : q11.x, hi1 ? q11.z : q11.x, c.w0, c.w1);
    albedo = trilerp(hi0 ? q00.w : q00.y, hi1 ? q00.w : q00.y, hi0 ? q10.w : q10.y, hi1 ? q10.w : q10.y,
                     hi0 ? q01.w : q01.y, hi1 ? q01.w : q01.y, hi0 ? q11.w : q11.y, hi1 ? q11.w : q11.y, c.w0, c.w1);
}

template <bool COUNT>
DEV MedStep medium_step(const DScene &sc, const DMedium m, const DRay &ray, float sample, uint32_t channel, bool want_albedo, Counters &cnt,
                        const SpecCtx &cx = SpecCtx()) {
    MedStep mi;
#if (MTS_TRAITS & MT_MEDIA) && MTS_SPEC_N == 3      // promised: every medium heterogeneous, grey, on a pair grid, with spectral extinction
    const bool m_homogeneous = false, m_pair = true, m_grey = true, m_spectral = true;
#elif MTS_TRAITS & MT_MEDIA                         // spectral variant: heterogeneous, two gridvolume_spectral grids on one geometry and interval
    const bool m_homogeneous = false, m_pair = false, m_grey = false, m_spectral = true;
#elif MTS_TRAITS & MT_HOMOG                         // promised: every medium homogeneous
    const bool m_homogeneous = true, m_pair = false, m_grey = m.grey != 0, m_spectral = m.has_spectral_extinction != 0;
#else
    const bool m_homogeneous = m.is_homogeneous != 0, m_pair = m.pair_grid != nullptr, m_grey = m.grey != 0, m_spectral = m.has_spectral_extinction != 0;
#endif
    bool active = true; float mint = 0.f, maxt = pm_inf();
    if (!m_homogeneous) {
        active = bbox_ray_intersect(m.aabb, ray, mint, maxt);
        active = active && (pm_isfinite(mint) || pm_isfinite(maxt));
        if (!active) { mint = 0.f; maxt = pm_inf(); }
    }
    mint = pm_max(ray.mint, mint);
    maxt = pm_min(ray.maxt, maxt);
    Spec combined = m_homogeneous ? volume_eval(cload(sc.volumes + m.sigma_t), ray.o MTS_CXI(m.sigma_t)) * m.scale : spec_s(m.max_density);
    float mext = pick(combined, channel);
    const float inv_mext = m_homogeneous ? 0.f : m.inv_max_density;
    float sampled_t = mint + div_by_invariant(-pm_log(1.f - sample), mext, inv_mext);
    bool valid_mi = active && (sampled_t <= maxt);
    mi.t = valid_mi ? sampled_t : pm_inf();
    mi.p = ray_at(ray, sampled_t);
    mi.mint = mint;
    mi.sigma_t = mi.sigma_s = spec_s(0.f);
    if (m_homogeneous) {
        Spec st = volume_eval(cload(sc.volumes + m.sigma_t), mi.p MTS_CXI(m.sigma_t)) * m.scale;
        mi.sigma_t = st;
        if (want_albedo) mi.sigma_s = st * volume_eval(cload(sc.volumes + m.albedo), mi.p MTS_CXI(m.albedo));
    } else if (valid_mi) {
        if (COUNT) MTS_SEG(cnt, 1);
        if (m_pair) {                          // everything comes from the medium record and the interleaved grid
            const int sx = m.pair_nx < 2 ? 2 : m.pair_nx;
            GridCell c = grid_cell_clamp(m.pair_w2l, m.pair_affine & 1, m.pair_nx, m.pair_ny, m.pair_nz, sx, mi.p);
            float st_raw, al_raw;
            grid_fetch_pair(as_global(m.pair_grid), c, sx, (m.pair_affine & 2) != 0, st_raw, al_raw);
            float st = m.scale * st_raw;
            mi.sigma_t = spec_s(st);
            if (want_albedo) mi.sigma_s = spec_s(st * al_raw);
        } else {
            const DVolume vs = cload(sc.volumes + m.sigma_t), va = cload(sc.volumes + m.albedo);
            if (m.shared_grid && m_grey && vs.filter == MTS_FILTER_TRILINEAR) {
                // both grids share one cell / one set of weights; single channel: one value serves the three channels
                GridCell c = grid_cell(vs, mi.p);
                float st = m.scale * grid_fetch1(as_global(vs.data), c);
                mi.sigma_t = spec_s(st);
                if (want_albedo) mi.sigma_s = spec_s(st * grid_fetch1(as_global(va.data), c));    // the tracking walks never read sigma_s
#if MTS_SPEC_N != 3
            } else if ((MTS_TRAITS & MT_MEDIA) != 0 || m.shared_grid == 2) {
                // gridvolume_spectral for both, same geometry and spectral interval: one cell, one set of weights and spectral nodes
                GridRef g;
                for (int k = 0; k < 16; ++k) g.w2l[k] = vs.w2l[k];
                g.data = vs.data; g.nx = vs.nx; g.ny = vs.ny; g.nz = vs.nz;
                g.channels_affine_filter_wrap = (uint32_t) vs.channels | ((uint32_t) (vs.affine != 0) << 8) | ((uint32_t) vs.filter << 16) | ((uint32_t) vs.wrap << 24) |
                                                ((uint32_t) (vs.columns_equal != 0 && (!want_albedo || va.columns_equal != 0)) << 9);
                const DVolumeSp sp = cload(cx.volume_sp + m.sigma_t);
                if (want_albedo) {
                    const SpecPair r = volume_eval_grid_spectral_pair(g, va.data, mi.p, cx.wl, sp.lambda_min, sp.lambda_max);
                    const Spec st = m.scale * r.a;
                    mi.sigma_t = st; mi.sigma_s = st * r.b;
                } else
                    mi.sigma_t = m.scale * volume_eval_grid_spectral(g, mi.p, cx.wl, sp.lambda_min, sp.lambda_max);
#endif
            } else {
                Spec st = m.scale * volume_eval(vs, mi.p MTS_CXI(m.sigma_t));
                mi.sigma_t = st;
                if (want_albedo) mi.sigma_s = st * volume_eval(va, mi.p MTS_CXI(m.albedo));
            }
        }
        if (COUNT) cnt.n_lookup++;
        if (COUNT) MTS_SEG(cnt, 2);
    }
    mi.combined = combined;
    mi.inv_combined = inv_mext;
    mi.info = (m_homogeneous ? MI_HOMOGENEOUS : 0u) | (m_spectral ? MI_SPECTRAL : 0u) |
              (m.sample_emitters ? MI_SAMPLE_EMITTERS : 0u) | (m_grey ? MI_GREY : 0u) | ((uint32_t) m.phase << MI_PHASE_SHIFT);
    return mi;
}

// exp(-t * combined) per channel (medium.cpp:84); grey media evaluate it once
DEV Spec transmittance_exp_g(float t, Spec combined, bool grey) {
    if (grey) return spec_s(pm_exp(-t * combined.x));
    return transmittance_exp(t, combined);
}

#if MTS_SPEC_N == 3
// Film splat of one finished sample: librender/integrator.cpp:265-285 + librender/imageblock.cpp:79-172
// (The spectral build's splat_values_t in integrator_dev.h is this function's second half; sharing it changed the register allocation of
// the regrouping kernel -- 170 -> 197 SGPR spills -- so the rgb kernels keep their own copy.)
// `own` receives the samples that land in the lane's own pixel: either register accumulators (nested
// formulation) or the pixel's film entry itself, updated with float atomics in sample order.
template <bool OWN_ATOMIC>
DEV void splat_sample_t(const DScene &sc, const DBlock &blk, uint32_t lx, uint32_t ly, F2 position_sample, F3 L, bool valid,
                        MTS_GLOBAL_AS float *film, float *own) {
    const DSensor &se = sc.sensor;
    float v[5];                                                 // srgb_to_xyz, core/spectrum.h:221-227
    v[0] = pm_fma(0.180423f, L.z, pm_fma(0.357580f, L.y, 0.412453f * L.x));
    v[1] = pm_fma(0.072169f, L.z, pm_fma(0.715160f, L.y, 0.212671f * L.x));
    v[2] = pm_fma(0.950227f, L.z, pm_fma(0.119193f, L.y, 0.019334f * L.x));
    if (sc.integrator.monochrome)                               // integrator.cpp:270-271: xyz = spec_u.x()
        v[0] = v[1] = v[2] = L.x;
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
            float wy = as_global(rf.values)[min((int) pm_abs((basey + (float) yr) * rf.scale_factor), 31)];     // eval_discretized, core/rfilter.h:62-65
            int fy = blk.oy - border + y - se.crop_y;
            for (uint32_t xr = 0; xr < n; ++xr) {
                int x = lox + (int) xr;
                if (x > hix) break;
                float wx = as_global(rf.values)[min((int) pm_abs((basex + (float) xr) * rf.scale_factor), 31)];
                float weight = wy * wx;
                int fx = blk.ox - border + x - se.crop_x;
                if (fx >= 0 && fy >= 0 && fx < se.crop_w && fy < se.crop_h) {                         // film clipping, imageblock.cpp:49-77
                    float *dst = (float *) (film + 5 * ((size_t) fy * se.crop_w + fx));
                    for (int k = 0; k < 5; ++k) atomicAdd(dst + k, v[k] * weight);
                }
            }
        }
    } else {
        int lox = (int) pm_ceil(posx - .5f), loy = (int) pm_ceil(posy - .5f);
        if (lox == (int) lx && loy == (int) ly) {
            if (OWN_ATOMIC) { for (int k = 0; k < 5; ++k) atomicAdd(own + k, v[k]); }
            else { for (int k = 0; k < 5; ++k) own[k] += v[k]; }
        } else if (lox >= 0 && loy >= 0 && lox < sx && loy < sy) {
            float *dst = (float *) (film + 5 * ((size_t) (blk.oy + loy - se.crop_y) * se.crop_w + (blk.ox + lox - se.crop_x)));
            for (int k = 0; k < 5; ++k) atomicAdd(dst + k, v[k]);
        }
    }
}

#endif // MTS_SPEC_N == 3 (the spectral build splats through splat_values_t, integrator_dev.h)

// ---------------------------------------------------------------------------------------------------------
// Per-path state.  "Hot" fields are touched by every tracking step; "cold" fields (ColdStore) only once per
// sample or per NEE / direct-light walk.
struct PathState {
    Pcg32 rng;
    DRay ray;                       // the ray being tracked now (main path, or the NEE / direct-light walk)
    Hit si;                         // cached closest hit of `ray`
    int medium;                     // medium containing ray.o
    Spec thr, res; float eta; uint32_t depth, channel;       // main path (volpath.cpp:54-67)
    Spec trans; float wa, wb;       // walk: transmittance; NEE: wa = total_dist, wb = ds.dist; direct: wb = bs.pdf
#if MTS_SPEC_N != 3
    Spec wl;                        // the sample's wavelengths (integrator.cpp:252)
#endif
    uint32_t st, mode, flags;
};
// cold field offsets (floats)
enum { C_POS = 0,            // 2: film position of the current sample
       C_RAYW = 2,           // 1: sensor ray weight
       C_SO = 3, C_SD = 6,   // 3 + 3: parked main-path origin / direction
       C_SHIT = 9,           // 8: parked main-path hit (t, p, uv, shape, prim)
       C_SMED = 17,          // 1: parked medium id
       C_CW = 18,                           // 3 (spectral: 4): pending NEE weight
       C_EMIT = C_CW + MTS_SPEC_DW,         // 3 (4): emitter value of the NEE sample
       C_ACC = C_EMIT + MTS_SPEC_DW,        // 5: this path's film accumulators X, Y, Z, A, W (the reference's per-block ImageBlock entry)
       C_SAMPLE = C_ACC + 5,                // 1: index of the sample in flight (bits of a uint32)
       C_COUNT = C_SAMPLE + 1 };            // 30 (32)
static_assert(C_COUNT <= 32, "the cold record is one 128-byte line");
// Struct-of-arrays store addressed as base[k * stride]: LDS (stride 256, one workgroup) or HBM (stride = paths in flight)
// or as one 128-byte record per path (AOS: the workgroup driver, whose lanes hold arbitrary paths -- a record is written by one
// lane as whole cache lines instead of 30 scattered dwords)
#define MTS_COLD_RECORD 32      // floats per path record (C_COUNT rounded up to a 128-byte line)
template <class Ptr /* float* into LDS, or MTS_GLOBAL_AS float* into HBM */, bool AOS = false>
struct ColdStoreT {
    Ptr base; uint32_t stride;
    DEV auto &f(int k) const { return AOS ? base[k] : base[(size_t) k * stride]; }
    DEV void put3(int k, F3 v) const { f(k) = v.x; f(k + 1) = v.y; f(k + 2) = v.z; }
    DEV F3 get3(int k) const { return f3(f(k), f(k + 1), f(k + 2)); }
#if MTS_SPEC_N == 3
    DEV void put_spec(int k, Spec v) const { put3(k, v); }
    DEV Spec get_spec(int k) const { return get3(k); }
#else
    DEV void put_spec(int k, Spec v) const { f(k) = v.x; f(k + 1) = v.y; f(k + 2) = v.z; f(k + 3) = v.w; }
    DEV Spec get_spec(int k) const { return spec4(f(k), f(k + 1), f(k + 2), f(k + 3)); }
#endif
    DEV void put_hit(const Hit &h) const {
        f(C_SHIT) = h.t; put3(C_SHIT + 1, h.p); f(C_SHIT + 4) = h.uv.x; f(C_SHIT + 5) = h.uv.y;
        f(C_SHIT + 6) = __int_as_float(h.shape); f(C_SHIT + 7) = __int_as_float(h.prim);
    }
    DEV Hit get_hit() const {
        Hit h; h.t = f(C_SHIT); h.p = get3(C_SHIT + 1); h.uv.x = f(C_SHIT + 4); h.uv.y = f(C_SHIT + 5);
        h.shape = __float_as_int(f(C_SHIT + 6)); h.prim = __float_as_int(f(C_SHIT + 7)); return h;
    }
};
// What a path needs from its surroundings
typedef ColdStoreT<float *> ColdStore;                       // generic pointer (the per-lane driver parks cold state in LDS)
#if defined(EXP_COLD_SOA)
typedef ColdStoreT<MTS_GLOBAL_AS float *> ColdStoreHbm;
#else
typedef ColdStoreT<MTS_GLOBAL_AS float *, true> ColdStoreHbm;
#endif
// (workgroup driver: cold state in HBM, addressed with GLOBAL instructions)
template <class Cold>
struct PathEnvT {
    DBlock blk; uint32_t lx, ly, sample_count; MTS_GLOBAL_AS float *film; Cold cold;
    MTS_GLOBAL_AS float *park;     // volpathmis: a second 128-byte record per path -- p_over_f / p_over_f_nee while an emitter-sampling walk runs (volpathmis_flat.h)
    uint32_t index;                // the pixel's Morton index inside its spiral block (seeds its stream, integrator.cpp:198)
};
// Scheduling classes: the block a path is waiting for
enum { B_INT = 0, B_MED /* free-flight step of the main path */, B_SCATTER, B_WSURF, B_SURF, B_PHASE, B_NEW,
       B_MEDW /* free-flight step of an NEE / direct-light walk */, B_DONE, B_COUNT };

template <bool COUNT>
struct VolpathMachine {
    const DScene &sc;
    Counters &cnt;
    DEV VolpathMachine(const DScene &sc_, Counters &cnt_) : sc(sc_), cnt(cnt_) {}
    // wavelength context of a path (empty in the rgb build)
#if MTS_SPEC_N == 3
    DEV SpecCtx ctx(const PathState &) const { return SpecCtx(); }
#else
    DEV SpecCtx ctx(const PathState &p) const { SpecCtx cx = make_ctx(sc); cx.wl = p.wl; return cx; }
#endif

    // A freshly spawned ray that cannot reach the scene's bounding box is resolved on the spot (the first
    // test of ShapeKDTree::ray_intersect_scalar, kdtree.h:2095-2098); everything else queues for INTERSECT.
    DEV void queue_intersection(PathState &p) const {
        float bmint, bmaxt;
        bbox_ray_intersect(sc.bbox, p.ray, bmint, bmaxt);
        p.si.t = pm_inf();                                     // si.shape etc. are only read behind hit_valid()
        if (pm_max(p.ray.mint, bmint) <= bmaxt) p.flags |= FL_NEEDS_INT; else p.flags &= ~FL_NEEDS_INT;
    }
    template <class E> DEV void begin_sample(PathState &p, const E &e) const {      // integrator.cpp:242-264, volpath.cpp:48-71
        const DSensor &se = sc.sensor;
        const float px = (float) (e.lx + (uint32_t) e.blk.ox), py = (float) (e.ly + (uint32_t) e.blk.oy);
        if (se.wavefront) seed_wavefront_sample(p.rng, se, e.blk, e.lx, e.ly, __float_as_uint(e.cold.f(C_SAMPLE)));   // gpu_* streams: one per (pixel, sample)
        F2 u = p.rng.next_2d();
        F2 position_sample; position_sample.x = px + u.x; position_sample.y = py + u.y;
        F2 aperture_sample; aperture_sample.x = .5f; aperture_sample.y = .5f;
        if (se.needs_aperture_sample) aperture_sample = p.rng.next_2d();
        if (se.shutter_open_time > 0.f) (void) p.rng.next_1d();   // time sample (integrator.cpp:248-250)
#if MTS_SPEC_N == 3
        (void) p.rng.next_1d();                                // wavelength sample, unused in rgb
#else
        {                                                      // integrator.cpp:252 -> perspective.cpp:169-182, distant.cpp:311-313; blk_new recomputes the weights
            const float wavelength_sample = p.rng.next_1d();
            float wav_weight; Spec srf_weight;
            p.wl = sc.srf >= 0 ? sample_wavelengths_srf(sc, wavelength_sample, srf_weight) : sample_wavelengths(wavelength_sample, wav_weight);
        }
#endif
        F2 adjusted;
        adjusted.x = (position_sample.x - (float) se.crop_x) / (float) se.crop_w;
        adjusted.y = (position_sample.y - (float) se.crop_y) / (float) se.crop_h;
        F3 rw;
        p.ray = sensor_sample_ray(sc, adjusted, aperture_sample, rw);
        e.cold.f(C_POS) = position_sample.x; e.cold.f(C_POS + 1) = position_sample.y; e.cold.f(C_RAYW) = rw.x;   // grey weight
        p.medium = se.medium;
        p.thr = spec_s(1.f); p.res = spec_s(0.f); p.eta = 1.f; p.depth = 0;
#if MTS_SPEC_N == 3
        p.channel = sc.integrator.monochrome ? 0u : (uint32_t) pm_min(p.rng.next_1d() * 3.f, 2.f);     // volpath.cpp:64-67: rgb variants only
#else
        p.channel = 0u;                                        // volpath.cpp:63-67: no draw outside the rgb variants
#endif
        p.si.p = f3s(0.f); p.si.uv.x = p.si.uv.y = 0.f; p.si.prim = 0;
        const bool hide_emitters = sc.integrator.hide_emitters != 0;
        p.flags = FL_ALIVE | ((!hide_emitters && sc.environment >= 0) ? FL_VALID_RAY : 0u) | (!hide_emitters ? FL_SPEC_CHAIN : 0u);
        p.trans = spec_s(1.f); p.wa = p.wb = 0.f;
        queue_intersection(p);
        p.mode = M_MAIN; p.st = S_TOP;
    }
    // NEE walk finished (volpath.cpp:366 + :165-166 / :211): add the contribution, resume the main path
    template <class E> DEV void end_nee(PathState &p, const E &e) const {
        Spec emitted = p.trans * e.cold.get_spec(C_EMIT);
        p.res = p.res + e.cold.get_spec(C_CW) * emitted;
        p.mode = M_MAIN; p.medium = __float_as_int(e.cold.f(C_SMED));
        F3 d = e.cold.get3(C_SD);
        p.ray.d = d; p.ray.d_rcp = vrcp(d);
        if (p.flags & FL_FROM_MEDIUM) { p.ray.o = e.cold.get3(C_SO); p.st = S_PHASE; }
        else { p.si = e.cold.get_hit(); p.st = S_BSDF; }
    }
    // direct-light walk finished (volpath.cpp:464 + :246-252): MIS-weighted emitter hit, resume the main path
    template <class E> DEV void end_direct(PathState &p, const E &e, Spec emitter_val, float emitter_pdf) const {
        Spec emitted = p.trans * emitter_val;
        if (emitter_pdf != 0.f) p.res = p.res + mis_weight(p.wb, emitter_pdf) * p.thr * emitted;
        p.ray = spawn_ray(e.cold.get3(C_SO), e.cold.get3(C_SD));
        p.si = e.cold.get_hit(); p.medium = __float_as_int(e.cold.f(C_SMED));
        p.flags = (p.flags & ~FL_NEEDS_INT) | FL_ALIVE;
        p.mode = M_MAIN; p.st = S_TOP;
    }

    // Loop heads of the three reference loops (cheap): volpath.cpp:79-87, :283-287, :385-388
    template <bool DEFER = false, class E> DEV void top(PathState &p, const E &e) const {
        if (p.st != S_TOP) return;
        const uint32_t max_depth = (uint32_t) sc.integrator.max_depth, rr_depth = (uint32_t) sc.integrator.rr_depth;
        if (p.mode == M_MAIN) {
            bool active = (p.flags & FL_ALIVE) && any_nonzero(p.thr);
            float q = pm_min(hmax(p.thr) * (p.eta * p.eta), .95f);
            bool perform_rr = p.depth > rr_depth;
            active = active && (p.rng.next_1d() < q || !perform_rr);
            if (perform_rr) p.thr = p.thr * pm_rcp(q);
            if (!active || p.depth >= max_depth) p.st = S_NEW;
            else { if (COUNT) cnt.n_iter++; p.st = p.medium >= 0 ? S_MED : S_SURF; }
        } else if (p.mode == M_NEE) {
            float remaining_dist = p.wb * (1.f - MTS_SHADOW_EPSILON) - p.wa;
            p.ray.maxt = remaining_dist;
            if (!(remaining_dist > 0.f)) { if (DEFER) p.st = S_ENDNEE; else end_nee(p, e); }
            else { if (COUNT) cnt.n_nee_step++; p.st = p.medium >= 0 ? S_MED : S_SURF; }
        } else {
            if (COUNT) cnt.n_nee_step++;
            p.st = p.medium >= 0 ? S_MED : S_SURF;
        }
    }
    DEV static bool wants_int(const PathState &p) { return (p.st == S_MED || p.st == S_SURF || p.st == S_DIRB) && (p.flags & FL_NEEDS_INT); }
    // block this path waits for (valid once top() has run)
    DEV static int classify(const PathState &p) {
        if (p.st == S_DONE) return B_DONE;
        if (wants_int(p)) return B_INT;
        if (p.st == S_MED) return p.mode == M_MAIN ? B_MED : B_MEDW;
        if (p.st == S_SCATTER) return B_SCATTER;
        if (p.st == S_SURF) return p.mode == M_MAIN ? B_SURF : B_WSURF;
        if (p.st == S_BSDF) return B_SURF;
        if (p.st == S_PHASE) return B_PHASE;
        return B_NEW;
    }

    // ================================================================= NEW: finish a sample, start the next (integrator.cpp:265-288)
    template <class E> DEV void blk_new(PathState &p, const E &e) const {
        if (p.st != S_NEW) return;
        const DSensor &se = sc.sensor;
        F2 position_sample; position_sample.x = e.cold.f(C_POS); position_sample.y = e.cold.f(C_POS + 1);
        float acc[5];                                          // summed in sample order like the block entry (imageblock.cpp:163-168)
        for (int k = 0; k < 5; ++k) acc[k] = e.cold.f(C_ACC + k);
#if MTS_SPEC_N == 3
        splat_sample_t<false>(sc, e.blk, e.lx, e.ly, position_sample, f3s(e.cold.f(C_RAYW)) * p.res, (p.flags & FL_VALID_RAY) != 0, e.film, acc);
#else
        {
            float wav_weight; (void) sample_wavelengths(0.f, wav_weight);
            const Spec ww = sc.srf >= 0 ? srf_weights_of(sc, p.wl) : spec_s(wav_weight);
            const Spec L = (ww * e.cold.f(C_RAYW)) * p.res;             // ray_weight = wav_weight (x the sensor's grey weight), integrator.cpp:265
            float xyz[3];
            spectrum_to_xyz(sc.cie, L, p.wl, xyz);                      // integrator.cpp:266-269
            const float v[5] = { xyz[0], xyz[1], xyz[2], (p.flags & FL_VALID_RAY) != 0 ? 1.f : 0.f, 1.f };
            if (sc.bin_count == 0) splat_values_t<false>(sc, e.blk, e.lx, e.ly, position_sample, v, e.film, acc);
            else splat_values_bins(sc, e.blk, e.lx, e.ly, position_sample, v, p.res, p.wl, e.film, acc);
        }
#endif
        const uint32_t sample_idx = __float_as_uint(e.cold.f(C_SAMPLE)) + 1u;
        if (sample_idx == e.sample_count) {                    // block -> film (hdrfilm.cpp:207-211)
            float *own = (float *) (e.film + MTS_FILM_STRIDE(sc) * ((size_t) (e.blk.oy + (int) e.ly - se.crop_y) * se.crop_w + (e.blk.ox + (int) e.lx - se.crop_x)));
            for (int k = 0; k < 5; ++k) atomicAdd(own + k, acc[k]);
            p.st = S_DONE;
        } else {
            for (int k = 0; k < 5; ++k) e.cold.f(C_ACC + k) = acc[k];
            e.cold.f(C_SAMPLE) = __uint_as_float(sample_idx);
            begin_sample(p, e);
        }
    }
    // ================================================================= INTERSECT (volpath.cpp:109,182,241,298,339,395,425)
    template <class E> DEV void blk_int(PathState &p, const E &e) const {
        if (!wants_int(p)) return;
        p.si = ray_intersect(sc, p.ray);
        p.flags &= ~FL_NEEDS_INT;
        start_direct(p, e);
    }
    template <class E> DEV void start_direct(PathState &p, const E &e) const {      // volpath.cpp:239-245: the direct-light walk runs on a copy
        if (p.st != S_DIRB || (p.flags & FL_NEEDS_INT)) return;
        e.cold.put3(C_SO, p.ray.o); e.cold.put3(C_SD, p.ray.d); e.cold.put_hit(p.si);
        p.trans = spec_s(1.f);
        p.mode = M_DIR; p.st = S_TOP;
    }
    // ================================================================= MEDIUM: one free-flight step of any of the three loops
    // MODEK: 0 = lanes of the main path only, 1 = lanes of a walk only (workgroup driver: one class each), -1 = any
    template <bool DEFER = false, int MODEK = -1, class E> DEV void blk_med(PathState &p, const E &e) const {
        if (p.st != S_MED || (p.flags & FL_NEEDS_INT)) return;
        if (MODEK == 0 && p.mode != M_MAIN) return;
        if (MODEK == 1 && p.mode == M_MAIN) return;
        const uint32_t max_depth = (uint32_t) sc.integrator.max_depth;
        const float u = p.rng.next_1d();                       // volpath.cpp:105 / :294 / :391
        MedStep mi;
        const SpecCtx cx = ctx(p); (void) cx;
        WATERFALL_BEGIN(p.medium, mu)
            mi = medium_step<COUNT>(sc, cload(sc.media + mu), p.ray, u, p.channel, MODEK == 0 ? true : (MODEK == 1 ? false : p.mode == M_MAIN), cnt MTS_CX);
        WATERFALL_END
        if (p.si.t < mi.t) mi.t = pm_inf();                    // volpath.cpp:112 / :300 / :397
#if MTS_TRAITS & MT_MEDIA
        const bool spectral = true, homogeneous = false, grey = MTS_SPEC_N == 3;
#elif MTS_TRAITS & MT_HOMOG
        const bool spectral = (mi.info & MI_SPECTRAL) != 0, homogeneous = true, grey = (mi.info & MI_GREY) != 0;
#else
        const bool spectral = (mi.info & MI_SPECTRAL) != 0, homogeneous = (mi.info & MI_HOMOGENEOUS) != 0, grey = (mi.info & MI_GREY) != 0;
#endif
        const Spec sigma_n = homogeneous ? spec_s(0.f) : mi.combined - mi.sigma_t;
        const uint32_t channel = p.channel;
        const bool is_main = MODEK == 0 ? true : (MODEK == 1 ? false : p.mode == M_MAIN), is_nee = MODEK == 0 ? false : p.mode == M_NEE;
        // transmittance / free-flight pdf of this step, one formula for the three loops:
        // medium.cpp:77-89 (volpath.cpp:113-117, :401-405) and the NEE variant bounded by remaining_dist (:305-311)
        const float remaining_dist = is_nee ? p.ray.maxt : pm_inf();
        Spec weight = is_main ? p.thr : p.trans;
        if (spectral) {
            float t = pm_min(mi.t, p.si.t);
            if (is_nee) t = pm_min(remaining_dist, t);
            t = t - mi.mint;
            const bool surface_first = p.si.t < mi.t || mi.t > remaining_dist;
            if (grey || !homogeneous) {                        // the combined extinction is one value for every channel (a grey medium, or a
                float tr = pm_exp(-t * mi.combined.x);          // heterogeneous one: its majorant is a scalar, heterogeneous.cpp:29): one exp, one division
                float tr_pdf = surface_first ? tr : tr * mi.combined.x;
                weight = weight * (tr_pdf > 0.f ? tr * pm_rcp(tr_pdf) : 0.f);     // spectrum / scalar = spectrum * (1 / scalar), dmath.h
            } else {
                Spec tr = transmittance_exp(t, mi.combined);
                Spec free_flight_pdf = surface_first ? tr : tr * mi.combined;
                float tr_pdf = pick(free_flight_pdf, channel);
                weight = weight * (tr_pdf > 0.f ? tr / tr_pdf : spec_s(0.f));
            }
        }
        float u2 = 0.f;
        if (is_main) u2 = p.rng.next_1d();                     // volpath.cpp:123 (drawn even when the medium was left)
        if (is_nee) {                                          // volpath.cpp:313-315
            if (mi.t > remaining_dist && mi.t != pm_inf()) p.wa = p.wb;
            if (mi.t > remaining_dist) mi.t = pm_inf();
        }
        const bool valid = mi.t != pm_inf();
        if (!valid) {                                          // escaped_medium: surface part of this iteration
            if (is_main) p.thr = weight; else p.trans = weight;
            p.st = S_SURF;
            return;
        }
        const bool real_scatter = is_main && !(u2 >= (grey ? div_by_invariant(mi.sigma_t.x, mi.combined.x, mi.inv_combined)
                                                                : div_by_invariant(pick(mi.sigma_t, channel), pick(mi.combined, channel), mi.inv_combined)));
        if (!real_scatter) {
            // null collision of the main path (volpath.cpp:128-131,140-144) or a step of a walk (:322-333, :411-420)
            if (grey) {
                if (is_main) { if (spectral) weight = weight * ((sigma_n.x * mi.combined.x) * pm_rcp(sigma_n.x)); }
                else { if (spectral) weight = weight * sigma_n.x; else weight = weight * div_by_invariant(sigma_n.x, mi.combined.x, mi.inv_combined); }
            } else {
                if (is_main) { if (spectral) weight = weight * (sigma_n * pick(mi.combined, channel) / pick(sigma_n, channel)); }
                else { if (spectral) weight = weight * sigma_n; else weight = weight * div_by_invariant(sigma_n, mi.combined, mi.inv_combined); }
            }
            if (is_nee) p.wa += mi.t;
            p.ray.o = mi.p; p.ray.mint = 0.f; p.si.t = p.si.t - mi.t;
            p.st = S_TOP;
            if (is_main) p.thr = weight;
            else {
                p.trans = weight;
                if (!any_nonzero(weight)) {                        // volpath.cpp:358 / :456
                    if (DEFER) p.st = is_nee ? S_ENDNEE : S_ENDDIR0;
                    else if (is_nee) end_nee(p, e); else end_direct(p, e, spec_s(0.f), 0.f);
                }
            }
            return;
        }
        // real scattering event of the main path, volpath.cpp:133-160
        p.depth += 1;
        if (!(p.depth < max_depth)) { p.thr = weight; p.flags &= ~FL_ALIVE; p.st = S_TOP; return; }
        if (grey) {
            if (spectral) weight = weight * ((mi.sigma_s.x * mi.combined.x) * pm_rcp(mi.sigma_t.x));
            else weight = weight * (mi.sigma_s.x / mi.sigma_t.x);
        } else {
            if (spectral) weight = weight * (mi.sigma_s * pick(mi.combined, channel) / pick(mi.sigma_t, channel));
            else weight = weight * (mi.sigma_s / mi.sigma_t);
        }
        p.thr = weight;
        const bool sample_emitters = (mi.info & MI_SAMPLE_EMITTERS) != 0;
        p.flags |= FL_VALID_RAY;
        p.flags = sample_emitters ? (p.flags & ~FL_SPEC_CHAIN) : (p.flags | FL_SPEC_CHAIN);
        p.ray.o = mi.p;                                        // scattering position; ray.d stays the incident direction
        p.st = sample_emitters ? S_SCATTER : S_PHASE;
    }
    // ================================================================= SCATTER: emitter sampling at a medium interaction
    // (volpath.cpp:162-167 -> sample_emitter :261-281); the walk itself runs as M_NEE steps
    template <class E> DEV void blk_scatter(PathState &p, const E &e) const {
        if (p.st != S_SCATTER) return;
        p.st = S_PHASE;
        Spec emitter_val;
        const SpecCtx cx = ctx(p); (void) cx;
        DirSample ds = sample_emitter_direction(sc, p.ray.o, p.rng.next_2d(), false, emitter_val MTS_CX);
        if (ds.pdf == 0.f) return;
        float phase_val = 0.f;
        WATERFALL_BEGIN(p.medium, mu)
            phase_val = phase_eval<true>(sc, cload(sc.media + mu).phase, -p.ray.d, p.ray.o, ds.d MTS_CX);
        WATERFALL_END
        e.cold.put_spec(C_CW, p.thr * phase_val); e.cold.put_spec(C_EMIT, emitter_val);
        e.cold.put3(C_SO, p.ray.o); e.cold.put3(C_SD, p.ray.d); e.cold.f(C_SMED) = __int_as_float(p.medium);
        p.trans = spec_s(1.f); p.wa = 0.f; p.wb = ds.dist;
        p.ray = spawn_ray(p.ray.o, ds.d); p.ray.mint = 0.f;
        queue_intersection(p);
        p.flags |= FL_FROM_MEDIUM;
        p.mode = M_NEE; p.st = S_TOP;
    }
    // ================================================================= SURFACE step of a walk (volpath.cpp:336-364, :423-462)
    template <class E> DEV void blk_wsurf(PathState &p, const E &e) const {
        if (p.st != S_SURF || p.mode == M_MAIN || (p.flags & FL_NEEDS_INT)) return;
        const bool hit = hit_valid(p.si);
        const bool is_nee = p.mode == M_NEE;
        if (is_nee) p.wa += p.si.t;
        int emitter = is_nee ? -1 : sc.environment;
        Surf sf; sf.wi = -p.ray.d; sf.sh.n = f3s(0.f); sf.n = f3s(0.f);
        Spec nt = spec_s(0.f); int is_tr = 0, ext = -1, inte = -1;
        if (hit) {
            WATERFALL_BEGIN(p.si.shape, su)
                const DShape s = cload(sc.shapes + su);
                if (!is_nee) emitter = s.emitter;
                nt = s.bsdf_type == MTS_BSDF_NULL ? spec_s(1.f) : spec_s(0.f);                // null.cpp:70-73, bsdf.cpp:11-14
                is_tr = s.is_medium_transition; ext = s.exterior; inte = s.interior;
                if (emitter >= 0) complete_surface(sc, s, p.si, p.ray.d, sf);
                else if (is_tr) sf.n = hit_geo_normal(sc, s, p.si);
            WATERFALL_END
        }
        if (emitter >= 0) {                                    // direct-light walk reached an emitter, volpath.cpp:430-440
            const F3 ref_p = e.cold.get3(C_SO);
            DirSample ds;                                      // render/records.h:168-174
            ds.p = p.si.p; ds.n = sf.sh.n; ds.d = p.si.p - ref_p; ds.dist = norm(ds.d); ds.d = ds.d / ds.dist;
            if (!hit) ds.d = -sf.wi;
            ds.emitter = emitter; ds.pdf = 0.f; ds.delta = false;
            const SpecCtx cx = ctx(p); (void) cx;
            end_direct(p, e, emitter_eval(sc, emitter, sf.wi.z MTS_CX), pdf_emitter_direction(sc, ref_p, ds));
            return;
        }
        if (hit) {
            p.trans = p.trans * nt;
            p.ray = spawn_ray(p.si.p, p.ray.d);
            queue_intersection(p);
            if (is_tr) p.medium = dot(p.ray.d, sf.n) > 0 ? ext : inte;         // interaction.h:178-200
        }
        if (hit && any_nonzero(p.trans)) p.st = S_TOP;
        else if (is_nee) end_nee(p, e);
        else end_direct(p, e, spec_s(0.f), 0.f);
    }
    // ================================================================= SURFACE interaction of the main path (volpath.cpp:184-212)
    template <class E> DEV void blk_surf(PathState &p, const E &e) const {
        if (p.st != S_SURF || p.mode != M_MAIN || (p.flags & FL_NEEDS_INT)) return;
        const uint32_t max_depth = (uint32_t) sc.integrator.max_depth;
        const bool hit = hit_valid(p.si);
        Surf sf; sf.wi = -p.ray.d; sf.n = f3s(0.f); sf.sh.s = sf.sh.t = sf.sh.n = f3s(0.f);
        int emitter = sc.environment, bsdf_id = 0;
        if (hit) {
            WATERFALL_BEGIN(p.si.shape, su)
                const DShape s = cload(sc.shapes + su);
                emitter = s.emitter; bsdf_id = s.bsdf;
                // the shading frame is only read by emitter evaluation and by the NEE of a smooth BSDF: a null boundary needs neither
                if (s.emitter >= 0 || (s.bsdf_flags & F_Smooth) != 0) complete_surface(sc, s, p.si, p.ray.d, sf);
            WATERFALL_END
        }
        const SpecCtx cx = ctx(p); (void) cx;
        if ((p.flags & FL_SPEC_CHAIN) && emitter >= 0) p.res = p.res + p.thr * emitter_eval(sc, emitter, sf.wi.z MTS_CX);
        if (!hit) { p.flags &= ~FL_ALIVE; p.st = S_TOP; return; }
        p.st = S_BSDF;
        bool active_e = false;
        WATERFALL_BEGIN(bsdf_id, bu)
            active_e = (cload(sc.bsdfs + bu).flags & F_Smooth) != 0 && (p.depth + 1 < max_depth);
        WATERFALL_END
        if (!active_e) return;
        Spec emitter_val;                                      // volpath.cpp:200-212 -> sample_emitter :261-281
        DirSample ds = sample_emitter_direction(sc, p.si.p, p.rng.next_2d(), false, emitter_val MTS_CX);
        if (ds.pdf == 0.f) return;
        F3 wo = to_local(sf.sh, ds.d);
        Spec bsdf_val; float bpdf;
        WATERFALL_BEGIN(bsdf_id, bu)
            const DBsdf bsdf = cload(sc.bsdfs + bu);
            bsdf_val = bsdf_eval(bsdf, sf.wi, wo MTS_CXI(bu));
            bpdf = bsdf_pdf(bsdf, sf.wi, wo MTS_CXI(bu));
        WATERFALL_END
        e.cold.put_spec(C_CW, p.thr * bsdf_val * mis_weight(ds.pdf, ds.delta ? 0.f : bpdf)); e.cold.put_spec(C_EMIT, emitter_val);
        e.cold.put_hit(p.si); e.cold.put3(C_SD, p.ray.d); e.cold.f(C_SMED) = __int_as_float(p.medium);
        p.trans = spec_s(1.f); p.wa = 0.f; p.wb = ds.dist;
        p.ray = spawn_ray(p.si.p, ds.d);
        queue_intersection(p);
        p.flags &= ~FL_FROM_MEDIUM;
        p.mode = M_NEE; p.st = S_TOP;
    }
    // ================================================================= BSDF sampling (volpath.cpp:214-252)
    template <class E> DEV void blk_bsdf(PathState &p, const E &e) const {
        if (p.st != S_BSDF) return;
        const uint32_t max_depth = (uint32_t) sc.integrator.max_depth;
        Surf sf; int bsdf_id = 0, is_tr = 0, ext = -1, inte = -1;
        WATERFALL_BEGIN(p.si.shape, su)
            const DShape s = cload(sc.shapes + su);
            complete_surface(sc, s, p.si, p.ray.d, sf);
            bsdf_id = s.bsdf; is_tr = s.is_medium_transition; ext = s.exterior; inte = s.interior;
        WATERFALL_END
        const float s1 = p.rng.next_1d(); const F2 s2 = p.rng.next_2d();
        BSDFSample bs; Spec bsdf_val;
        const SpecCtx cx = ctx(p); (void) cx;
        WATERFALL_BEGIN(bsdf_id, bu)
            const DBsdf bsdf = cload(sc.bsdfs + bu);
            bsdf_val = bsdf_sample(bsdf, sf.wi, s1, s2, bs MTS_CXI(bu));
        WATERFALL_END
        p.thr = p.thr * bsdf_val;
        p.eta *= bs.eta;
        p.ray = spawn_ray(p.si.p, to_world(sf.sh, bs.wo));
        p.flags |= FL_ALIVE;
        const bool non_null_bsdf = !(bs.sampled_type & F_Null);
        if (non_null_bsdf) { p.depth += 1; p.flags |= FL_VALID_RAY; }
        if (non_null_bsdf && (bs.sampled_type & F_Delta)) p.flags |= FL_SPEC_CHAIN;
        if (bs.sampled_type & F_Smooth) p.flags &= ~FL_SPEC_CHAIN;
        const bool add_emitter = !(bs.sampled_type & F_Delta) && any_nonzero(p.thr) && (p.depth < max_depth);
        const int new_medium = is_tr ? (dot(p.ray.d, sf.n) > 0 ? ext : inte) : p.medium;     // volpath.cpp:249-250
        queue_intersection(p);
        if (add_emitter) { e.cold.f(C_SMED) = __int_as_float(new_medium); p.wb = bs.pdf; p.st = S_DIRB; start_direct(p, e); }   // the walk runs in the old medium
        else { p.medium = new_medium; p.st = S_TOP; }
    }
    // ================================================================= PHASE sampling (volpath.cpp:169-175)
    template <class E> DEV void blk_phase(PathState &p, const E &e) const {
        if (p.st != S_PHASE) return;
        const float s1 = p.rng.next_1d(); const F2 s2 = p.rng.next_2d();      // left-to-right (SURVEY.md 8(a'))
        F3 wo;
        const SpecCtx cx = ctx(p); (void) cx;
        WATERFALL_BEGIN(p.medium, mu)
            wo = phase_sample<true>(sc, cload(sc.media + mu).phase, make_frame(p.ray.d), p.ray.o, s1, s2 MTS_CX);   // mi.sh_frame = Frame3f(ray.d), medium.cpp:42
        WATERFALL_END
        p.ray = spawn_ray(p.ray.o, wo); p.ray.mint = 0.0f;
        queue_intersection(p);
        p.flags |= FL_ALIVE;
        p.st = S_TOP;
    }

    // Run the block(s) of class `sel` for this lane (a lane whose state does not match falls through)
    // the deferred tail of a block that ran with DEFER (p holds the full state here)
    template <class E> DEV void finish(PathState &p, const E &e) const {
        if (p.st == S_ENDNEE) end_nee(p, e);
        else if (p.st == S_ENDDIR0) end_direct(p, e, spec_s(0.f), 0.f);
    }
    template <bool DEFER = false, bool SPLIT = false, class E> DEV void run(PathState &p, const E &e, int sel) const {
        switch (sel) {
            case B_NEW: blk_new(p, e); break;
            case B_INT: blk_int(p, e); break;
            case B_MED: if (SPLIT) blk_med<DEFER, 0>(p, e); else blk_med<DEFER, -1>(p, e); break;
            case B_MEDW: if (SPLIT) blk_med<DEFER, 1>(p, e); else blk_med<DEFER, -1>(p, e); break;
            case B_SCATTER: blk_scatter(p, e); break;
            case B_WSURF: blk_wsurf(p, e); break;
            case B_SURF: blk_surf(p, e); blk_bsdf(p, e); break;     // no NEE at this surface: the BSDF is sampled in the same visit
            case B_PHASE: blk_phase(p, e); break;
            default: break;
        }
    }
};

#if MTS_SPEC_N == 3
// ---------------------------------------------------------------------------------------------------------
// Driver 1: one lane = one pixel, state in registers, cold state in LDS, blocks chosen by a per-wave vote.
template <bool COUNT>
DEV void volpath_pixel_flat(const DScene &sc, Pcg32 &rng, const DBlock &blk, uint32_t lx, uint32_t ly, uint32_t sample_count,
                            float *__restrict__ film, ColdStore cold, Counters &cnt, const uint32_t *stop_flag) {
    VolpathMachine<COUNT> vm(sc, cnt);
    PathEnvT<ColdStore> e; e.blk = blk; e.lx = lx; e.ly = ly; e.sample_count = sample_count; e.film = as_global(film); e.cold = cold;
    PathState p; p.rng = rng;
    for (int k = 0; k < 5; ++k) e.cold.f(C_ACC + k) = 0.f;
    e.cold.f(C_SAMPLE) = __uint_as_float(0u);
    vm.begin_sample(p, e);
#if defined(MTSAMD_BLOCKSTATS)
    long long bs_t0 = clock64(); int bs_prev_sel = 20;
    unsigned long long bs_loc[45] = {};                      // laid out like g_blockstats
#endif
    for (uint32_t iter = 0; __ballot(p.st != S_DONE); ++iter) {
        if ((iter & 1023u) == 1023u && stop_requested(stop_flag)) break;       // should_stop(), integrator.h:143-146
#if defined(MTSAMD_BLOCKSTATS)
        if (COUNT) { long long t = clock64(); bs_loc[24 + bs_prev_sel] += (unsigned long long) (t - bs_t0); bs_t0 = t; bs_prev_sel = 20; }
#endif
        vm.top(p, e);
        // census + vote: run the ONE block most lanes of this wave are waiting for
        int cls = vm.classify(p);
        if (cls == B_MEDW) cls = B_MED;                        // this driver runs the generic MEDIUM block for both
        int sel = B_MED, best = -1;
        for (int b = 0; b < B_DONE; ++b) { int v = __popcll(__ballot(cls == b)); if (v > best) { best = v; sel = b; } }
        if (best == 0) continue;
#if defined(MTSAMD_BLOCKSTATS)
        if (COUNT) {
            bs_loc[sel] += 1ull; bs_loc[12 + sel] += (unsigned long long) best;
            long long t = clock64(); bs_loc[44] += (unsigned long long) (t - bs_t0); bs_t0 = t; bs_prev_sel = sel;
        }
#endif
        vm.run(p, e, sel);
    }
    if (p.st != S_DONE) {                                     // stopped (should_stop()): the finished samples of this pixel go to the film, integrator.cpp:120-130
        float *own = (float *) (e.film + 5 * ((size_t) (blk.oy + (int) ly - sc.sensor.crop_y) * sc.sensor.crop_w + (blk.ox + (int) lx - sc.sensor.crop_x)));
        for (int k = 0; k < 5; ++k) atomicAdd(own + k, e.cold.f(C_ACC + k));
    }
#if defined(MTSAMD_BLOCKSTATS)
    if (COUNT && __builtin_amdgcn_readfirstlane((int) (threadIdx.x & 63)) == (int) (threadIdx.x & 63))
        for (int k = 0; k < 45; ++k) atomicAdd(&g_blockstats[k], bs_loc[k]);
#endif
}

#endif // MTS_SPEC_N == 3

// ---------------------------------------------------------------------------------------------------------
// Hot path state of the workgroup driver: WG paths (pixels) per workgroup, struct-of-arrays in LDS; cold state lives in HBM.
// Every block class loads / stores only the fields it can touch (ClassFields), which keeps the blocks' register budget small.
// depth shares a dword with the packed small fields (15 bits: a path of more than 32767 scattering events would need to
// survive Russian roulette with probability < 0.95^32000).
// The sixteen dwords every tracking step touches come first: an LDS instruction reaches 64 KB (16 fields of 1024 paths) from its
// address register, later fields cost an address computation each.
enum { H_RNG = 0, H_O = 2, H_D = 5, H_DRCP = 8, H_MINT = 11, H_MAXT = 12, H_SIT = 13, H_MEDIUM = 14,
       H_PACKED = 15 /* st, mode, channel, flags, class, depth */, H_THR = 16, H_TRANS = H_THR + MTS_SPEC_DW, H_WA = H_TRANS + MTS_SPEC_DW,
       H_WB = H_WA + 1, H_ETA = H_WB + 1, H_RES = H_ETA + 1, H_SIX = H_RES + MTS_SPEC_DW /* si.p, si.uv, si.shape, si.prim */,
#if MTS_SPEC_N == 3
       H_COUNT = H_SIX + 7 };                                   // 35
#else
       H_WL = H_SIX + 7 /* the sample's four wavelengths */, H_COUNT = H_WL + 4 };      // 42
#endif

enum : uint32_t { G_RNG = 1, G_O = 2, G_D = 4 /* d and 1/d */, G_MINT = 8, G_MAXT = 16, G_SIT = 32 /* si.t */, G_SIX = 64 /* rest of si */,
                  G_MED = 128, G_THR = 256, G_RES = 512, G_ETA = 1024, G_TRANS = 2048, G_WA = 4096, G_WB = 8192,
#if MTS_SPEC_N == 3
                  G_WL = 0, G_ALL = 16383 };
#else
                  G_WL = 16384 /* read by every block that evaluates a spectrum, written by NEW */, G_ALL = 32767 };
#endif
// Measurement build (-DMTS_FUSE_INT=1, round 3): scenes without a BVH run the intersection of a freshly spawned ray at the end of
// the block that spawned it (PHASE, SCATTER, walk SURFACE, SURFACE + BSDF, NEW) instead of queueing the path for an INTERSECT visit,
// which saves 14 of 64 visits per sample on the metric scene (claim, state load / store, push).  Bit-identical; measured 546 -> 537
// Msamples/s on C3 (512 x 512 x 256): the twelve triangle tests then run at the 45 - 50 lanes of the host blocks instead of the 54
// of a class of their own, which costs what the saved visits gain.  Off by default.
#ifndef MTS_FUSE_INT
#define MTS_FUSE_INT 0
#endif
// What block class C (followed by top()) may read (`load`, a superset of `store`) and write (`store`).  end_nee / end_direct touch
// almost everything; the classes with partial sets run them deferred (VolpathMachine::finish on the full state).
template <int C> struct ClassFields { static constexpr uint32_t load = G_ALL, store = G_ALL; static constexpr bool defer = false; };
template <> struct ClassFields<B_MED> {          // main path: tracks thr
    static constexpr uint32_t load = G_RNG | G_O | G_D | G_MINT | G_MAXT | G_SIT | G_MED | G_THR | G_ETA | G_WL,
                              store = G_RNG | G_O | G_MINT | G_SIT | G_THR;
    static constexpr bool defer = true; };
template <> struct ClassFields<B_MEDW> {         // NEE / direct-light walk: tracks trans, the NEE walk also its distance budget
    static constexpr uint32_t load = G_RNG | G_O | G_D | G_MINT | G_MAXT | G_SIT | G_MED | G_TRANS | G_WA | G_WB | G_WL,
                              store = G_RNG | G_O | G_MINT | G_MAXT | G_SIT | G_TRANS | G_WA;
    static constexpr bool defer = true; };
template <> struct ClassFields<B_INT> {          // every lane of the class wants the intersection: si is written, never read
    static constexpr uint32_t load = G_O | G_D | G_MINT | G_MAXT | G_MED | G_TRANS, store = G_SIT | G_SIX | G_TRANS;
    static constexpr bool defer = true; };
template <> struct ClassFields<B_SCATTER> {
    static constexpr uint32_t load = G_RNG | G_O | G_D | G_MINT | G_MAXT | G_SIT | G_MED | G_THR | G_TRANS | G_WA | G_WB | G_WL,
                              store = G_RNG | G_O | G_D | G_MINT | G_MAXT | G_SIT | G_TRANS | G_WA | G_WB | (MTS_FUSE_INT ? G_SIX : 0u);
    static constexpr bool defer = true; };
template <> struct ClassFields<B_PHASE> {
    static constexpr uint32_t load = G_RNG | G_O | G_D | G_MINT | G_MAXT | G_SIT | G_MED | G_THR | G_ETA | G_WL,
                              store = G_RNG | G_O | G_D | G_MINT | G_MAXT | G_SIT | G_THR | (MTS_FUSE_INT ? G_SIX : 0u);
    static constexpr bool defer = true; };

template <int WG>
struct HotStore {
    uint32_t *base;                                          // &lds[0][path]
    DEV uint32_t &u(int k) const { return base[k * WG]; }
    DEV float f(int k) const { return __uint_as_float(base[k * WG]); }
    DEV void putf(int k, float v) const { base[k * WG] = __float_as_uint(v); }
    DEV void put3(int k, F3 v) const { putf(k, v.x); putf(k + 1, v.y); putf(k + 2, v.z); }
    DEV F3 get3(int k) const { return f3(f(k), f(k + 1), f(k + 2)); }
#if MTS_SPEC_N == 3
    DEV void put_spec(int k, Spec v) const { put3(k, v); }
    DEV Spec get_spec(int k) const { return get3(k); }
#else
    DEV void put_spec(int k, Spec v) const { putf(k, v.x); putf(k + 1, v.y); putf(k + 2, v.z); putf(k + 3, v.w); }
    DEV Spec get_spec(int k) const { return spec4(f(k), f(k + 1), f(k + 2), f(k + 3)); }
#endif
    DEV static uint32_t pack(const PathState &p, int cls) {
        return p.st | (p.mode << 4) | (p.channel << 6) | (p.flags << 8) | ((uint32_t) cls << 13) | ((p.depth < 32767u ? p.depth : 32767u) << 17);
    }
    DEV static int cls_of(uint32_t packed) { return (int) (packed >> 13) & 15; }
    DEV static void unpack(uint32_t pk, PathState &p) {
        p.st = pk & 15u; p.mode = (pk >> 4) & 3u; p.channel = (pk >> 6) & 3u; p.flags = (pk >> 8) & 31u; p.depth = pk >> 17;
    }
    // field groups: a block loads / stores only what it can read / write (ClassFields below)
    template <uint32_t M> DEV void store_m(const PathState &p, int cls) const {
        if (M & G_RNG) { u(H_RNG) = (uint32_t) p.rng.state; u(H_RNG + 1) = (uint32_t) (p.rng.state >> 32); }
        if (M & G_O) put3(H_O, p.ray.o);
        if (M & G_D) { put3(H_D, p.ray.d); put3(H_DRCP, p.ray.d_rcp); }
        if (M & G_MINT) putf(H_MINT, p.ray.mint);
        if (M & G_MAXT) putf(H_MAXT, p.ray.maxt);
        if (M & G_SIT) putf(H_SIT, p.si.t);
        if (M & G_SIX) { put3(H_SIX, p.si.p); putf(H_SIX + 3, p.si.uv.x); putf(H_SIX + 4, p.si.uv.y); u(H_SIX + 5) = (uint32_t) p.si.shape; u(H_SIX + 6) = (uint32_t) p.si.prim; }
        if (M & G_MED) u(H_MEDIUM) = (uint32_t) p.medium;
        if (M & G_THR) put_spec(H_THR, p.thr);
        if (M & G_RES) put_spec(H_RES, p.res);
        if (M & G_ETA) putf(H_ETA, p.eta);
        if (M & G_TRANS) put_spec(H_TRANS, p.trans);
#if MTS_SPEC_N != 3
        if (M & G_WL) put_spec(H_WL, p.wl);
#endif
        if (M & G_WA) putf(H_WA, p.wa);
        if (M & G_WB) putf(H_WB, p.wb);
        u(H_PACKED) = pack(p, cls);
    }
    template <uint32_t M> DEV void load_m(PathState &p) const {
        p.rng.state = 0; p.rng.inc = (PCG32_DEFAULT_STREAM << 1u) | 1u;
        if (M & G_RNG) p.rng.state = (uint64_t) u(H_RNG) | ((uint64_t) u(H_RNG + 1) << 32);
        p.ray.o = (M & G_O) ? get3(H_O) : f3s(0.f);
        p.ray.d = (M & G_D) ? get3(H_D) : f3s(0.f); p.ray.d_rcp = (M & G_D) ? get3(H_DRCP) : f3s(0.f);
        p.ray.mint = (M & G_MINT) ? f(H_MINT) : 0.f; p.ray.maxt = (M & G_MAXT) ? f(H_MAXT) : 0.f;
        p.si.t = (M & G_SIT) ? f(H_SIT) : pm_inf();
        p.si.p = f3s(0.f); p.si.uv.x = p.si.uv.y = 0.f; p.si.shape = -1; p.si.prim = 0;
        if (M & G_SIX) { p.si.p = get3(H_SIX); p.si.uv.x = f(H_SIX + 3); p.si.uv.y = f(H_SIX + 4); p.si.shape = (int) u(H_SIX + 5); p.si.prim = (int) u(H_SIX + 6); }
        p.medium = (M & G_MED) ? (int) u(H_MEDIUM) : -1;
        p.thr = (M & G_THR) ? get_spec(H_THR) : spec_s(0.f); p.res = (M & G_RES) ? get_spec(H_RES) : spec_s(0.f);
        p.eta = (M & G_ETA) ? f(H_ETA) : 1.f;
        p.trans = (M & G_TRANS) ? get_spec(H_TRANS) : spec_s(0.f); p.wa = (M & G_WA) ? f(H_WA) : 0.f; p.wb = (M & G_WB) ? f(H_WB) : 0.f;
#if MTS_SPEC_N != 3
        p.wl = (M & G_WL) ? get_spec(H_WL) : spec_s(0.f);
#endif
        unpack(u(H_PACKED), p);
    }
    // the fields of M on top of a state that already holds the others (st / mode / flags / depth stay as the registers have them)
    template <uint32_t M> DEV void load_add(PathState &p) const {
        if (M & G_RNG) p.rng.state = (uint64_t) u(H_RNG) | ((uint64_t) u(H_RNG + 1) << 32);
        if (M & G_O) p.ray.o = get3(H_O);
        if (M & G_D) { p.ray.d = get3(H_D); p.ray.d_rcp = get3(H_DRCP); }
        if (M & G_MINT) p.ray.mint = f(H_MINT);
        if (M & G_MAXT) p.ray.maxt = f(H_MAXT);
        if (M & G_SIT) p.si.t = f(H_SIT);
        if (M & G_SIX) { p.si.p = get3(H_SIX); p.si.uv.x = f(H_SIX + 3); p.si.uv.y = f(H_SIX + 4); p.si.shape = (int) u(H_SIX + 5); p.si.prim = (int) u(H_SIX + 6); }
        if (M & G_MED) p.medium = (int) u(H_MEDIUM);
        if (M & G_THR) p.thr = get_spec(H_THR);
        if (M & G_RES) p.res = get_spec(H_RES);
        if (M & G_ETA) p.eta = f(H_ETA);
        if (M & G_TRANS) p.trans = get_spec(H_TRANS);
        if (M & G_WA) p.wa = f(H_WA);
        if (M & G_WB) p.wb = f(H_WB);
#if MTS_SPEC_N != 3
        if (M & G_WL) p.wl = get_spec(H_WL);
#endif
    }
    DEV void store(const PathState &p, int cls) const { store_m<G_ALL>(p, cls); }
    DEV void load(PathState &p) const { load_m<G_ALL>(p); }
};

// kernel arguments of render_kernel_wga, re-read by the block functions through the constant address space
struct WgArgs {
    DScene sc; const DBlock *blocks; uint32_t n_blocks, block_size, sample_count; float *film; float *cold_g; uint32_t cold_stride;
    unsigned long long *counters;            // [0..2] loop counters, [MTS_DIAG_BASE ..] ring-stall record
    const uint32_t *stop_flag;               // host-visible word: non-zero = Integrator::cancel() / timeout (integrator.h:143-146)
    // Cost-sorted tiles (round 4).  NULL: workgroup w renders paths w WG .. w WG + WG - 1 of the blocks' concatenated Morton orders (a
    // workgroup sits in ONE spiral block).  Otherwise slot s = (w WG + pid) / 16 holds (block index << 12) | tile: sixteen
    // Morton-consecutive pixels (a 4 x 4 square) of that block; 0xFFFFFFFF = padding.  mts_render sorts the tiles of a launch by the
    // cost its calibration launch measured, so that a workgroup holds pixels of EQUAL cost and its paths finish together -- in a
    // spatial block of an atmosphere seen by a distant sensor the costs differ by a multiple near the horizon, the cheap pixels
    // finished early and four fifths of the paths of such a workgroup were done while the rest kept it (and its 16 waves) resident.
    const uint32_t *tiles; uint32_t n_tiles;
};
#define MTS_TILE_PIXELS 16u

template <int WG>
DEV bool wg_env(const WgArgs &a, uint32_t wg_base, uint32_t pid, PathEnvT<ColdStoreHbm> &e) {     // pixel owned by path `pid`; false: outside the block
    const uint32_t ppb = a.block_size * a.block_size;       // a multiple of WG (checked by the launcher)
    e.sample_count = a.sample_count; e.film = as_global(a.film);
#if defined(EXP_COLD_SOA)
    e.cold.base = as_global(a.cold_g) + wg_base + pid; e.cold.stride = a.cold_stride;
#else
    e.cold.base = as_global(a.cold_g) + (size_t) (wg_base + pid) * MTS_COLD_RECORD; e.cold.stride = 1;
    __builtin_assume(((uintptr_t) e.cold.base & 127u) == 0);       // hipMalloc'ed base, 128-byte records: lets neighbouring fields share one wide access
#endif
    e.park = as_global(a.cold_g) + ((size_t) a.cold_stride + wg_base + pid) * MTS_COLD_RECORD;      // behind the cold records (volpathmis launches allocate it)
    e.lx = e.ly = 0; e.index = 0;
    if (a.tiles != nullptr) {                               // cost-sorted tiles: the block is per lane (vector loads; only NEW and the start need them)
        const uint32_t slot = (wg_base + pid) / MTS_TILE_PIXELS;
        if (slot >= a.n_tiles) return false;
        const uint32_t ent = as_global(a.tiles)[slot];
        if (ent == 0xFFFFFFFFu) return false;
        const MTS_GLOBAL_AS int32_t *bp = (const MTS_GLOBAL_AS int32_t *) as_global(a.blocks + (ent >> 12));
        e.blk.ox = bp[0]; e.blk.oy = bp[1]; e.blk.sx = bp[2]; e.blk.sy = bp[3]; e.blk.id = (uint32_t) bp[4]; e.blk.sample_base = (uint32_t) bp[5];
        e.blk.film_off_lo = (uint32_t) bp[6]; e.blk.film_off_hi = (uint32_t) bp[7];
        e.index = ((ent & 4095u) * MTS_TILE_PIXELS) | (pid & (MTS_TILE_PIXELS - 1u));
    } else {
        // one spiral block per workgroup.  block_size is a power of two (mts_render rounds it up, as integrator.cpp:91-97 does): shift
        // and mask instead of the ~30 instructions of a 32-bit division, on every block visit
        const uint32_t b = wg_base >> (uint32_t) __builtin_ctz(ppb);   // uniform
        e.index = (wg_base & (ppb - 1u)) + pid;
        if (b >= a.n_blocks) return false;
        e.blk = cload(a.blocks + b);
    }
    e.film += ((size_t) e.blk.film_off_hi << 32) | e.blk.film_off_lo;     // the film slot of the entry's pass (mts_render; 0 with a single pass)
    e.lx = compact_bits(e.index); e.ly = compact_bits(e.index >> 1);      // morton_decode, integrator.cpp:200
    return e.lx < (uint32_t) e.blk.sx && e.ly < (uint32_t) e.blk.sy;
}

// One block of class C for the path `pid`: load, run (repeat while the path stays in class C and enough lanes do), store.
// WF: wavefront (gpu_*) streams -- one PCG32 per (pixel, sample), seeded with TEA (sampler.cpp:89-92).  The hot state holds only the
// generator's 64-bit state; its increment, which for the scalar variants' streams is the default stream's constant, is recomputed here
// from (pixel, sample index) on every load: a 64-bit TEA of four rounds, ~60 instructions, and one dword of the cold record -- in an
// instantiation of its own, so that the kernels of the scalar streams do not change by an instruction.
template <bool COUNT, int WG, int C, bool WF = false>
#ifndef WG_BLOCK_ATTR
#define WG_BLOCK_ATTR __forceinline__   // a real call costs 48 callee-saved VGPR spills + reloads per block visit (measured: 5 TB of scratch writes per render)
#endif
static __device__ WG_BLOCK_ATTR int wg_block(const MTS_CONST_AS void *kernarg_, uint32_t *hot_lds, uint32_t wg_base_, uint32_t pid, Counters *cnt) {
    // arguments of a non-kernel function arrive in VGPRs; tell the compiler which ones are wave-uniform
    const uint64_t ka = (uint64_t) (uintptr_t) kernarg_;
    uint32_t ka_lo = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) ka), ka_hi = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) (ka >> 32));
    asm volatile("" : "+s"(ka_lo), "+s"(ka_hi));             // opaque: scene loads stay inside this block (no hoisting when inlined)
    const MTS_CONST_AS void *kernarg = (const MTS_CONST_AS void *) (uintptr_t) ((uint64_t) ka_lo | ((uint64_t) ka_hi << 32));
    const uint32_t wg_base = (uint32_t) __builtin_amdgcn_readfirstlane((int) wg_base_);
    const WgArgs a = cload_k<WgArgs>(kernarg);
    VolpathMachine<COUNT> vm(a.sc, *cnt);
    PathEnvT<ColdStoreHbm> e; wg_env<WG>(a, wg_base, pid, e);
    HotStore<WG> hs; hs.base = hot_lds + pid;
    typedef ClassFields<C> CF;
    PathState p;
    if (COUNT && C == B_MED) MTS_SEG_BEGIN(*cnt);
#if defined(EXP_COLD_TOUCH)
    // A walk's step may end the walk, and the end reads the path's cold record (parked at the start of the walk, ~10^5 cycles ago: by now in
    // the Infinity Cache or in HBM) in a tail that the whole wave waits for.  One dword of the record is requested here, a block's worth of
    // work before it is needed, so that the tail's loads find the line in the cache.  Hand-issued: the compiler would sink a plain load to
    // its use.  The register is only read behind the s_waitcnt below.
    uint32_t cold_touch = 0;
    if (C == B_MEDW || C == B_WSURF)
        asm volatile("global_load_dword %0, %1, off" : "=v"(cold_touch) : "v"((const MTS_GLOBAL_AS float *) (e.cold.base + C_SD)) : "memory");
#endif
    hs.template load_m<CF::load>(p);
    if (WF && C != B_INT) p.rng.inc = wavefront_increment(a.sc.sensor, e.blk, e.lx, e.ly, __float_as_uint(e.cold.f(C_SAMPLE)));
    if (COUNT && C == B_MED) MTS_SEG(*cnt, 0);
    int cls;
    // A tracking step is most often followed by another one (null collisions): while at least half of the wave's lanes stay in this
    // class the block runs again on the registers it holds -- no LDS round trip, no ring push, no claim for those paths; lanes that
    // leave wait at the store below.  Measured (C3 / C4, Msamples/s): never 535 / 224, from 32 lanes 548 / 259, from 44 lanes
    // 546 / 255, from 16 lanes 526 / 255, always 397 / 239.
#pragma nounroll
    for (int rounds = 0;; ++rounds) {
        vm.template run<CF::defer, true>(p, e, C);
        if (COUNT && C == B_MED) MTS_SEG(*cnt, 3);
        vm.template top<CF::defer>(p, e);
        if (MTS_FUSE_INT && (C == B_PHASE || C == B_SCATTER || C == B_WSURF || C == B_SURF || C == B_NEW) && a.sc.bvh_node_count == 0) {
            vm.blk_int(p, e);                                // acts on the lanes whose new ray can reach the scene (wants_int)
            vm.template top<CF::defer>(p, e);                // start_direct() leaves a direct-light walk at its loop head
        }
        if (COUNT && C == B_MED) MTS_SEG(*cnt, 4);
        cls = vm.classify(p);
        if ((MTS_CHAIN & 16) && C == B_SURF) {                 // the same for the main path (its second visit finds no hit: the sample ends)
            if (cls != C || rounds >= 1) break;
            continue;
        }
        if ((MTS_CHAIN & 1) && C == B_WSURF) {                 // a walk that left through a boundary face with nothing behind it (queue_intersection:
            if (cls != C || rounds >= 1) break;                 // the ray misses the scene's box) comes straight back to this class to end
            continue;
        }
        if (!(C == B_MED || C == B_MEDW) || cls != C || rounds >= 16) break;
        if (__popcll(__ballot(true)) < (C == B_MEDW ? MTS_REPEAT_MIN_W : MTS_REPEAT_MIN)) break;
    }
#if defined(EXP_COLD_TOUCH)
    if (C == B_MEDW || C == B_WSURF) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); asm volatile("" :: "v"(cold_touch)); }
#endif
    bool chained = false;
    if ((MTS_CHAIN & 4) && C == B_MEDW && cls == B_WSURF) {    // the walk left the medium: its surface step(s) run here, on the lanes that have one
        hs.template load_add<G_ALL & ~CF::load>(p);
#pragma nounroll
        for (int r = 0; r < 2; ++r) {
            vm.blk_wsurf(p, e);
            vm.top(p, e);
            cls = vm.classify(p);
            if (cls != B_WSURF) break;
        }
        chained = true;
    }
    if ((MTS_CHAIN & 2) && (C == B_WSURF || chained) && cls == B_PHASE) {      // the walk has ended: the main path's phase sample (volpath.cpp:169-175)
        vm.blk_phase(p, e);
        vm.top(p, e);
        cls = vm.classify(p);
    }
    if ((MTS_CHAIN & 8) && C == B_MED && cls == B_SCATTER) {   // a real collision: emitter sample and the start of its walk (volpath.cpp:162-167)
        vm.blk_scatter(p, e);
        vm.template top<true>(p, e);
        cls = vm.classify(p);
        hs.template store_m<CF::store | ClassFields<B_SCATTER>::store>(p, cls);
    } else
    if (chained) hs.template store_m<G_ALL>(p, cls);
    else hs.template store_m<CF::store>(p, cls);
    if (COUNT && C == B_MED) MTS_SEG(*cnt, 5);
    if (CF::defer && (p.st == S_ENDNEE || p.st == S_ENDDIR0)) {     // rare tail on the full state
        PathState q;
        hs.template load_m<G_ALL>(q);
        if (WF) q.rng.inc = wavefront_increment(a.sc.sensor, e.blk, e.lx, e.ly, __float_as_uint(e.cold.f(C_SAMPLE)));     // top() draws the roulette sample
        vm.finish(q, e);
        vm.top(q, e);
        cls = vm.classify(q);
        hs.template store_m<G_ALL>(q, cls);
    }
    return cls;
}

// ---------------------------------------------------------------------------------------------------------
// Driver 2: asynchronous regrouping.  No workgroup barriers and no sort: one LDS ring of path ids per block class.  A wave claims up to 64 ids from the fullest ring
// (compare-and-swap on its head), runs that block with every claimed lane active, and appends each path to
// the ring of the class it waits for next (one LDS atomic add on the tail per lane; the value returned is the slot).  Waves never wait for
// each other; a wave that finds every ring empty naps briefly.
//
// Ring protocol.  q_ctl[2c] / q_ctl[2c + 1] are head / tail of ring c: two 32-bit counters that only ever grow (slot = counter mod
// WG), each touched with 32-bit atomics only; "ring" B_DONE has no slots, its tail counts the finished paths; q_ctl[2 B_COUNT] is the
// workgroup's stop word.  The slots are 16 bit wide and touched with 16-bit loads / stores only; a slot holds (lap, id), lap = the
// ring index's lap number modulo 2^(16 - log2 WG) (RingSlot below; round 3 -- rounds 1 and 2 marked a slot empty / full instead):
//   producer: release fence (state in LDS / HBM is written), tail++ -> index, store (lap of index, id): no look at the slot, no wait;
//   consumer: head: h -> h + n by compare-and-swap (n <= tail - h of a snapshot: those indices are already handed out), the n slots
//             read together with it; a slot is accepted when it carries the lap of its index, acquire fence.  Nothing is written back.
// A consumer can be ahead of its producer (index handed out, id not stored yet): the slot then still carries the previous lap, and
// the lane waits in a wave-uniform loop (a divergent `while` would park the ready lanes behind the reconvergence point).  The wait is
// BOUNDED: after MTS_RING_SPIN_LIMIT polls it writes a diagnostic record (ring, index, head, tail) and stops the workgroup --
// mts_render reports an error instead of hanging.  A slot is overwritten one lap (WG pushes through this ring) after it was written,
// while its consumer reads it within a few instructions of its claim; a lane that ever sat between claim and read for a whole lap
// would find a newer lap, run into that bound and fail the render loudly.  A path that never comes back for any other reason leaves
// the finished count short: the idle wait is bounded too -- by ELAPSED TIME (round 4; a nap count measured the wave's own speed, and
// a debugger, a profiler or a throttled clock stretches another wave's long block visit but not the naps): MTS_IDLE_TICKS of the
// constant 100 MHz clock (s_memrealtime) in a row with every ring empty -> diagnostic code 3.
// Stopping (Integrator::cancel / timeout, or a stall) adds no exit to the claim loop (a second exit measured 4.5 % slower): the
// first lane to raise the stop word adds 2^31 to every head, which makes every ring look empty to every wave (a count above WG is no
// count, see the snapshot) and every pending claim fail; a wave that finds every ring empty looks at the stop word before it naps.
#define MTS_RING_SPIN_LIMIT (1u << 22)
#define MTS_IDLE_TICKS 1000000000u     // ten seconds of the 100 MHz constant clock with nothing waiting anywhere before a wave reports a lost path (never seen)
#define MTS_COST_FLAG 15           // counters[15] != 0: a calibration launch; counters[MTS_COST_BASE + slot]: summed finish times of the paths of tile `slot`
#define MTS_COST_BASE 16
#define MTS_INJECT_SLOT 14         // counters[14] != 0 (set by mts_render from MTSAMD_TEST_INJECT_LOST_PATH, counting kernel variants only):
                                   // the first wave of workgroup 0 drops one hand-over, and the idle bound is that many ticks -- the
                                   // test of the error path (tests/test_gpu_parity.py::test_lost_path_is_reported)
#define MTS_DIAG_BASE 4            // counters[MTS_DIAG_BASE + 0..5]: code (1 consumer / 2 producer), ring, index, head, tail, workgroup
enum : uint32_t { STOP_NONE = 0, STOP_CANCEL = 1, STOP_STALL = 2 };

// One nap of an idle wave: true once the rings have looked empty for `limit` ticks in a row.  The clock is read on the first nap of
// an idle period and on every 1024th after it (32-bit differences: the checks are ~1 ms apart, the counter wraps after 43 s).
DEV bool wga_idle_expired(uint32_t &idle_naps, uint32_t &idle_t0, uint32_t limit) {
    if (idle_naps++ == 0u) { idle_t0 = (uint32_t) __builtin_amdgcn_s_memrealtime(); return false; }
    if ((idle_naps & 1023u) != 0u) return false;
    return (uint32_t) __builtin_amdgcn_s_memrealtime() - idle_t0 > limit;
}

template <int WG>
DEV bool wga_raise_stop(uint32_t *q_ctl, uint32_t why) {
    if (atomicCAS(&q_ctl[2 * B_COUNT], (uint32_t) STOP_NONE, why) != STOP_NONE) return false;          // already stopping
#pragma unroll 1
    for (int c = 0; c < B_DONE; ++c) atomicAdd(&q_ctl[2 * c], 0x80000000u);
    return true;
}
template <int WG>
DEV void wga_stall(uint32_t code, int ring, uint32_t index, uint32_t *q_ctl, unsigned long long *counters) {
    const uint32_t hd = __atomic_load_n(&q_ctl[2 * ring], __ATOMIC_RELAXED), tl = __atomic_load_n(&q_ctl[2 * ring + 1], __ATOMIC_RELAXED);
    if (wga_raise_stop<WG>(q_ctl, STOP_STALL)) {                                                       // first lane of the workgroup to give up
        if (atomicCAS(counters + MTS_DIAG_BASE, 0ull, (unsigned long long) code) == 0ull) {            // first workgroup of the launch
            counters[MTS_DIAG_BASE + 1] = (unsigned long long) ring; counters[MTS_DIAG_BASE + 2] = index;
            counters[MTS_DIAG_BASE + 3] = hd; counters[MTS_DIAG_BASE + 4] = tl; counters[MTS_DIAG_BASE + 5] = blockIdx.x;
        }
    }
}
// Tagged slots: (lap << log2(WG)) | id.  Against the empty / full marking of rounds 1 and 2 (load, store on both sides, and a producer
// that waits for the previous lap's consumer) this is two dependent LDS round trips less per block visit; measured C3 +-0, C4 +2 %
// (profiles/r03_ab_experiments.log).
template <int WG> struct RingSlot {
    static constexpr uint32_t IDBITS = WG == 1024 ? 10 : WG == 512 ? 9 : WG == 256 ? 8 : WG == 128 ? 7 : 6;
    static_assert((1u << IDBITS) == (uint32_t) WG, "paths per workgroup: 64 .. 1024, a power of two");
    static constexpr uint32_t TAGMASK = (1u << (16 - IDBITS)) - 1u;
    DEV static uint32_t tag_of(uint32_t index) { return (index >> IDBITS) & TAGMASK; }
    DEV static uint32_t make(uint32_t index, uint32_t pid) { return (tag_of(index) << IDBITS) | pid; }
    DEV static bool matches(uint32_t v, uint32_t index) { return (v >> IDBITS) == tag_of(index); }
    DEV static uint32_t id(uint32_t v) { return v & (uint32_t) (WG - 1); }
};
// a lane ahead of its producer: wait (wave-uniform loop, bounded) until the slot carries the lane's lap
template <int WG>
DEV uint32_t wga_tag_wait(bool pending, uint16_t *slot, uint32_t *q_ctl, unsigned long long *counters, int ring, uint32_t index) {
    uint32_t out = 0xFFFFu;
#pragma nounroll
    for (uint32_t spins = 0;; ++spins) {
        if (pending) {
            const uint32_t v = __atomic_load_n(slot, __ATOMIC_RELAXED);
            if (RingSlot<WG>::matches(v, index)) { out = RingSlot<WG>::id(v); pending = false; }
        }
        if (!__builtin_amdgcn_ballot_w64(pending)) break;
        if (spins > MTS_RING_SPIN_LIMIT) { if (pending) wga_stall<WG>(1u, ring, index, q_ctl, counters); break; }
        if (__atomic_load_n(&q_ctl[2 * B_COUNT], __ATOMIC_RELAXED) != STOP_NONE) break;
    }
    return out;
}

template <int WG>
DEV void wga_push(int cls, uint32_t pid, bool valid, uint16_t (*q_ids)[WG], uint32_t *q_ctl) {
    // One LDS atomic per lane: the tail value it returns IS the lane's ring index; the LDS unit serialises the lanes that share a ring.
    // (Ranking the lanes first -- nine ballots, per-class counts, one atomic per class -- took 45 to 100 VALU instructions per push
    // and measured 1 to 3 % slower; the order of the ids inside a ring is immaterial.)
    uint32_t ti = 0;
    if (valid) ti = atomicAdd(&q_ctl[2 * cls + 1], 1u);
    if (valid && cls != B_DONE) __atomic_store_n(&q_ids[cls][ti & (uint32_t) (WG - 1)], (uint16_t) RingSlot<WG>::make(ti, pid), __ATOMIC_RELAXED);
}

// A stopped workgroup (Integrator::cancel(), the integrator's timeout, a stall) adds the accumulators of its unfinished pixels to the
// film: the reference puts a partially rendered block on the film as well (integrator.cpp:120-130: render_block returns early on
// should_stop(), film->put(block) follows; :213-216).  The sample in flight is dropped, W counts the finished ones.  Runs behind a
// workgroup barrier, when no wave touches the path state any more; `packed_at` = the hot dword that holds a path's state (S_DONE:
// already on the film).
template <int WG, int NT>
DEV void wg_flush_unfinished(const MTS_CONST_AS void *kernarg, const uint32_t *hot_lds, int packed_at, uint32_t wg_base) {
    const WgArgs a = cload_k<WgArgs>(kernarg);
#pragma unroll 1
    for (uint32_t pid = threadIdx.x; pid < (uint32_t) WG; pid += NT) {
        if ((hot_lds[packed_at * WG + pid] & 15u) == S_DONE) continue;
        PathEnvT<ColdStoreHbm> e;
        if (!wg_env<WG>(a, wg_base, pid, e)) continue;
        float *own = (float *) (e.film + MTS_FILM_STRIDE(a.sc) * ((size_t) (e.blk.oy + (int) e.ly - a.sc.sensor.crop_y) * a.sc.sensor.crop_w + (e.blk.ox + (int) e.lx - a.sc.sensor.crop_x)));
        for (int k = 0; k < 5; ++k) atomicAdd(own + k, e.cold.f(C_ACC + k));
    }
}

template <bool COUNT, int WG /* paths */, int NT /* threads: fewer threads than paths keeps the rings fuller */, bool WF = false /* wavefront streams, wg_block */>
DEV void volpath_workgroup_async(const MTS_CONST_AS void *kernarg, Counters &cnt) {
    constexpr int NQ = B_DONE;
    static_assert((WG & (WG - 1)) == 0 && WG <= 32768, "ring indices wrap with a mask and ids are 16 bit");
    static_assert(NT % 64 == 0 && WG % 64 == 0 && NT <= WG, "whole waves");
    __shared__ uint32_t hot_lds[H_COUNT * WG];
    __shared__ uint16_t q_ids[NQ][WG];
    __shared__ __attribute__((aligned(8))) uint32_t q_ctl[2 * B_COUNT + 2];      // head / tail pairs, then the stop word
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wg_base = blockIdx.x * WG;
#pragma unroll 1
    for (int c = 0; c < NQ; ++c) {
#pragma unroll 1
        for (uint32_t i = tid; i < (uint32_t) WG; i += NT) q_ids[c][i] = 0xFFFFu;      // runs once: not worth 300 unrolled instructions
    }
    if (tid < 2u * B_COUNT + 2u) q_ctl[tid] = 0;
    pm_tables_to_lds(tid);
    __syncthreads();
#pragma unroll 1
    for (uint32_t pid0 = tid; pid0 < (uint32_t) WG; pid0 += NT) {   // ---- initialise the paths (integrator.cpp:198) and queue them
        const WgArgs a = cload_k<WgArgs>(kernarg);
        VolpathMachine<COUNT> vm(a.sc, cnt);
        PathEnvT<ColdStoreHbm> e; PathState p;
        HotStore<WG> hs; hs.base = hot_lds + pid0;
        const bool ok = wg_env<WG>(a, wg_base, pid0, e);
        p.rng.state = 0; p.rng.inc = 0;
        p.ray = make_ray(f3s(0.f), f3(0.f, 0.f, 1.f), 0.f, 0.f); p.si.t = pm_inf(); p.si.p = f3s(0.f); p.si.uv.x = p.si.uv.y = 0.f; p.si.shape = -1; p.si.prim = 0;
        p.medium = -1; p.thr = p.res = p.trans = spec_s(0.f); p.eta = 1.f; p.depth = 0; p.channel = 0; p.mode = M_MAIN; p.flags = 0; p.wa = p.wb = 0.f;
#if MTS_SPEC_N != 3
        p.wl = spec_s(0.f);
#endif
        p.st = S_DONE;
        if (ok) {
            const uint32_t ppb = a.block_size * a.block_size;
            p.rng.seed(a.sc.sensor.seed + (uint64_t) e.blk.id * ppb + e.index, PCG32_DEFAULT_STREAM);     // sampler.cpp:83-96
            for (int k = 0; k < 5; ++k) e.cold.f(C_ACC + k) = 0.f;
            e.cold.f(C_SAMPLE) = __uint_as_float(0u);
            vm.begin_sample(p, e);
            vm.top(p, e);
        }
        const int cls = vm.classify(p);
        hs.store(p, cls);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        wga_push<WG>(cls, pid0, true, q_ids, q_ctl);
    }
#if defined(MTSAMD_BLOCKSTATS)
    long long bs_t0 = clock64(); unsigned long long bs_loc[48] = {};                      // laid out like g_blockstats
#endif
    uint32_t poll_ticks = (tid >> 6) * 2048u, idle_naps = 0, idle_t0 = 0; // per wave, staggered: paces the polls of the host's stop word
    uint32_t idle_limit = MTS_IDLE_TICKS; bool drop_one = false;
    if (COUNT) {                                              // error-path test hook (MTS_INJECT_SLOT): never in the production instantiation
        const uint32_t inj = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) cload_k<WgArgs>(kernarg).counters[MTS_INJECT_SLOT]);
        if (inj != 0u) { idle_limit = inj; drop_one = blockIdx.x == 0u && tid < 64u; }
    }
    // calibration launch of mts_render (counters[MTS_COST_FLAG] != 0): every path adds the time at which it finished its pixel to its tile's cost
    const bool record_cost = __builtin_amdgcn_readfirstlane((int) (uint32_t) cload_k<WgArgs>(kernarg).counters[MTS_COST_FLAG]) != 0;
    const long long cost_t0 = record_cost ? clock64() : 0ll;
#pragma unroll 1
    for (;;) {
      uint32_t n = 0, h = 0, spec_slot = 0xFFFFu; int sel = 0; bool finished = false;
      // ---- the claim: an inner loop of its own (snapshot, vote, compare-and-swap), left with a claim or when the workgroup is done.
      // (As `continue`s of the outer loop the retries dragged eighteen register copies of dead path state through every round.)
#pragma unroll 1
      for (;;) {
        // ---- snapshot of the rings, pick the fullest
        uint32_t hd = 0, avail = 0;
        if (lane < (uint32_t) B_COUNT) {
            hd = __atomic_load_n(&q_ctl[2 * lane], __ATOMIC_RELAXED);
            const uint32_t tl = __atomic_load_n(&q_ctl[2 * lane + 1], __ATOMIC_RELAXED);
            // the two loads are not one atomic snapshot: a head newer than the tail gives a "negative" count, which is no count at all
            // (so does a stopped ring, wga_raise_stop).  Any tail that was ever read is a lower bound of the tail now, so tl - hd
            // entries exist whenever the claim finds head == hd.
            avail = tl - hd;
            if (avail > (uint32_t) WG) avail = 0;
        }
        // argmax over the NQ rings in three DPP steps: lanes 0..7 hold (avail << 4 | 15 - ring), the maximum of a row's first eight
        // lanes ends up in lane 7 (ties go to the lower ring, as a first-maximum scan would have it); one readlane instead of eight
        // and no scalar compare chain
        uint32_t key = lane < (uint32_t) NQ ? ((avail << 4) | (15u - lane)) : 0u;
        key = max(key, (uint32_t) __builtin_amdgcn_update_dpp(0, (int) key, 0x111 /* row_shr:1 */, 0xf, 0xf, true));
        key = max(key, (uint32_t) __builtin_amdgcn_update_dpp(0, (int) key, 0x112 /* row_shr:2 */, 0xf, 0xf, true));
        key = max(key, (uint32_t) __builtin_amdgcn_update_dpp(0, (int) key, 0x114 /* row_shr:4 */, 0xf, 0xf, true));
        const uint32_t top_key = (uint32_t) __builtin_amdgcn_readlane((int) key, 7);
        const uint32_t best = top_key >> 4; sel = 15 - (int) (top_key & 15u);
#if defined(MTSAMD_BLOCKSTATS)
        if (COUNT) {      // population at snapshot time: finished paths [45], paths waiting in the rings [46], snapshots [47]
            uint32_t waiting = 0;
            for (int c = 0; c < NQ; ++c) waiting += (uint32_t) __builtin_amdgcn_readlane((int) avail, c);
            bs_loc[45] += (uint32_t) __builtin_amdgcn_readlane((int) avail, B_DONE); bs_loc[46] += waiting; bs_loc[47] += 1ull;
        }
#endif
        if (best == 0) {
            // every path of the workgroup has finished, or the workgroup was stopped (then every ring looks empty for good)
            if ((uint32_t) __builtin_amdgcn_readlane((int) avail, B_DONE) == (uint32_t) WG || __atomic_load_n(&q_ctl[2 * B_COUNT], __ATOMIC_RELAXED) != STOP_NONE) { finished = true; break; }
            // Integrator::should_stop() (integrator.h:143-146): waves look at the host's stop word (pinned host memory) now and then.  Reads
            // of host memory are a scarce resource -- the whole GPU sustains about 3 * 10^7 per second, and a poll on every nap made the
            // render 4.7 times longer -- so a wave earns a poll with 32768 ticks: one per nap, 256 per execution of the NEW block (below).
            // That is about 10^5 polls per second over all workgroups, and a few milliseconds until a workgroup notices.
            if ((poll_ticks += 1u) >= 32768u) {
                poll_ticks = 0;
                if (lane == 0 && __hip_atomic_load(cload_k<WgArgs>(kernarg).stop_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u)
                    (void) wga_raise_stop<WG>(q_ctl, STOP_CANCEL);
            }
            // a path that never comes back (a lost hand-over, see the protocol notes) would leave the finished count short for ever:
            // after MTS_IDLE_TICKS with every ring empty the wave reports it (diagnostic code 3) instead
            if (wga_idle_expired(idle_naps, idle_t0, idle_limit)) wga_stall<WG>(3u, B_DONE, 0u, q_ctl, cload_k<WgArgs>(kernarg).counters);
            __builtin_amdgcn_s_sleep(2);
#if defined(MTSAMD_BLOCKSTATS)
            if (COUNT) { long long t = clock64(); bs_loc[42] += (unsigned long long) (t - bs_t0); bs_t0 = t; }
#endif
            continue;
        }
        idle_naps = 0;
        // ---- claim up to 64 ids
        n = best < 64u ? best : 64u;
        h = (uint32_t) __builtin_amdgcn_readlane((int) hd, sel);
        uint32_t won = 0;
        // the slots are read together with the compare-and-swap (both depend on the snapshot only); a lost claim discards them
        if (lane < n) spec_slot = __atomic_load_n(&q_ids[sel][(h + lane) & (uint32_t) (WG - 1)], __ATOMIC_RELAXED);
        if (lane == 0) won = atomicCAS(&q_ctl[2 * sel], h, h + n) == h ? 1u : 0u;
#if defined(MTSAMD_BLOCKSTATS)
        if (COUNT) { bs_loc[10] += 1ull; if (!__builtin_amdgcn_readfirstlane((int) won)) bs_loc[11] += 1ull; }      // claim attempts / lost compare-and-swaps
#endif
        if (__builtin_amdgcn_readfirstlane((int) won)) break;
      }
      if (finished) break;
        uint32_t pid = 0xFFFFu;
        bool mine = lane < n;
        {
            uint16_t *slot = &q_ids[sel][(h + lane) & (uint32_t) (WG - 1)];
            const bool ready = mine && RingSlot<WG>::matches(spec_slot, h + lane);
            if (ready) pid = RingSlot<WG>::id(spec_slot);
            if (__builtin_amdgcn_ballot_w64(mine && !ready) != 0ull) {           // a lane ahead of its producer (rare)
                const uint32_t got = wga_tag_wait<WG>(mine && !ready, slot, q_ctl, cload_k<WgArgs>(kernarg).counters, sel, h + lane);
                if (mine && !ready) pid = got;
                mine = mine && pid != 0xFFFFu;                // 0xFFFF: the workgroup is stopping, the lane drops out
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#if defined(MTSAMD_BLOCKSTATS)
        if (COUNT) { bs_loc[sel] += 1ull; bs_loc[12 + sel] += (unsigned long long) n;
                     long long t = clock64(); bs_loc[44] += (unsigned long long) (t - bs_t0); bs_t0 = t; }
#endif
        int cls = B_DONE;
        if (mine) {
            switch (sel) {
                case B_INT: cls = wg_block<COUNT, WG, B_INT, WF>(kernarg, hot_lds, wg_base, pid, &cnt); break;
                case B_MED: cls = wg_block<COUNT, WG, B_MED, WF>(kernarg, hot_lds, wg_base, pid, &cnt); break;
                case B_MEDW: cls = wg_block<COUNT, WG, B_MEDW, WF>(kernarg, hot_lds, wg_base, pid, &cnt); break;
                case B_SCATTER: cls = wg_block<COUNT, WG, B_SCATTER, WF>(kernarg, hot_lds, wg_base, pid, &cnt); break;
                case B_WSURF: cls = wg_block<COUNT, WG, B_WSURF, WF>(kernarg, hot_lds, wg_base, pid, &cnt); break;
                case B_SURF: cls = wg_block<COUNT, WG, B_SURF, WF>(kernarg, hot_lds, wg_base, pid, &cnt); break;
                case B_PHASE: cls = wg_block<COUNT, WG, B_PHASE, WF>(kernarg, hot_lds, wg_base, pid, &cnt); break;
                default: cls = wg_block<COUNT, WG, B_NEW, WF>(kernarg, hot_lds, wg_base, pid, &cnt); break;
            }
        }
        // should_stop() for a busy wave: see the nap above.  Here, in the wake of the NEW block's film atomics, the vector load is cheap; at
        // the head of the claim loop it measured 3.5 % however seldom it ran.
        if (sel == B_NEW && (poll_ticks += 256u) >= 32768u) {
            poll_ticks = 0;
            if (lane == 0 && __hip_atomic_load(cload_k<WgArgs>(kernarg).stop_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u)
                (void) wga_raise_stop<WG>(q_ctl, STOP_CANCEL);
        }
#if defined(MTSAMD_BLOCKSTATS)
        if (COUNT) { long long t = clock64(); bs_loc[24 + sel] += (unsigned long long) (t - bs_t0); bs_t0 = t; }
#endif
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (COUNT && drop_one && cls != B_DONE) { mine = mine && lane != 0u; drop_one = false; }      // the injected lost hand-over (test hook)
        if (record_cost && mine && cls == B_DONE)             // once per path: calibration launches only
            atomicAdd(cload_k<WgArgs>(kernarg).counters + MTS_COST_BASE + (wg_base + pid) / MTS_TILE_PIXELS, (unsigned long long) (clock64() - cost_t0));
        wga_push<WG>(cls, pid, mine, q_ids, q_ctl);
#if defined(MTSAMD_BLOCKSTATS)
        if (COUNT) { long long t = clock64(); bs_loc[43] += (unsigned long long) (t - bs_t0); bs_t0 = t; }
#endif
    }
#if defined(MTSAMD_BLOCKSTATS)
    if (COUNT) {
        long long t = clock64(); bs_loc[44] += (unsigned long long) (t - bs_t0);
        for (int k = 0; k < 6; ++k) bs_loc[36 + k] = cnt.seg[k];
        if (lane == 0) for (int k = 0; k < 48; ++k) atomicAdd(&g_blockstats[k], bs_loc[k]);
    }
#endif
    __syncthreads();                                          // every wave has left the loop: the path state is final
    if (__atomic_load_n(&q_ctl[2 * B_COUNT], __ATOMIC_RELAXED) != STOP_NONE) wg_flush_unfinished<WG, NT>(kernarg, hot_lds, H_PACKED, wg_base);
}

// ---------------------------------------------------------------------------------------------------------
// Driver 3: lane-affine regrouping.  As driver 2 a wave runs ONE block class at a time with the hot state in LDS, but a path is bound
// to a lane: path `pid` is only ever executed by lane (pid mod 64) of whichever wave takes it.  That one rule pays three times:
//   * LDS banks: field k of path p lives at word k * WG + p, i.e. in bank p mod 32 for the dword accesses (MI355X_MICROARCH.md,
//     "LDS").  With the rings a wave's lanes hold arbitrary ids -- 32 random ids on 32 banks are a ~3.5-way conflict, the measured
//     SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.67 of round 2.  Here lane L holds an id = L mod 64: every state access of every
//     block is conflict-free.
//   * No rings: what waits for class c on lane L is a 16-bit mask (bit j = path L + 64 j).  Push = one atomic OR into the lane's own
//     word (64 lanes, 64 addresses, 32 banks x 2 groups: one LDS pass, where the ring tails serialised 64 same-address atomics);
//     claim = the wave reads its 64 mask records, votes for the class that serves most lanes (eight ballots), and every lane clears
//     one bit of its own word with an atomic AND.  No head / tail counters, no compare-and-swap that one lane wins for the wave
//     (61 % of those were lost), no slot hand-over and hence no wait that could stall.
//   * Waves do not queue behind one shared counter: sixteen waves that want the same class take different bits (each wave starts
//     its search at slot (wave id) mod 16), a lost bit is retried from the value the atomic returned.
// The price is that a lane can only be filled from its own <= 16 paths: a class with n waiting paths serves 64 (1 - exp(-n / 64))
// lanes on average instead of min(n, 64).
// q_mask[k][L]: classes 2k (low half) and 2k + 1 (high half) of lane L.  q_ctl[0] counts finished paths, q_ctl[1] is the stop word.
// Stopping needs no second exit: a raised stop word makes the vote come out empty, which is the (rare) path that already looks at
// the finished count.  After a stop the workgroup adds the accumulators of its unfinished pixels to the film, as the reference puts
// a partially rendered block on the film (integrator.cpp:120-130, 213-216).
template <bool COUNT, int WG, int NT, class M /* machine: HOT dwords per path, PACKED_AT, init(), block() */>
DEV void workgroup_lanes(const MTS_CONST_AS void *kernarg, Counters &cnt) {
    constexpr int PPL = WG / 64;                              // paths per lane
    static_assert(WG % 64 == 0 && NT % 64 == 0 && NT <= WG && PPL <= 16 && B_DONE == 8, "whole waves, 16-bit masks, eight classes in four words");
    __shared__ uint32_t hot_lds[M::HOT * WG];
    __shared__ uint32_t q_mask[4][64];
    __shared__ uint32_t q_ctl[4];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wg_base = blockIdx.x * WG;
    if (tid < 256u) (&q_mask[0][0])[tid] = 0u;
    if (tid < 4u) q_ctl[tid] = 0u;
    pm_tables_to_lds(tid);
    __syncthreads();
#pragma unroll 1
    for (uint32_t pid0 = tid; pid0 < (uint32_t) WG; pid0 += NT) {   // ---- initialise the paths (integrator.cpp:198) and queue them
        const int cls = M::init(kernarg, hot_lds, wg_base, pid0, &cnt);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        const uint32_t j = pid0 >> 6;
        if (cls != B_DONE) atomicOr(&q_mask[cls >> 1][lane], 1u << (j + 16u * ((uint32_t) cls & 1u)));
        const unsigned long long dm = __builtin_amdgcn_ballot_w64(cls == B_DONE);
        if (dm != 0ull && lane == (uint32_t) __builtin_ctzll(dm)) atomicAdd(&q_ctl[0], (uint32_t) __popcll(dm));
    }
    const uint32_t rot = (tid >> 6) & 15u;                    // where this wave starts looking in a mask
#if defined(MTSAMD_BLOCKSTATS)
    long long bs_t0 = clock64(); unsigned long long bs_loc[45] = {};                      // laid out like g_blockstats
#endif
    uint32_t poll_ticks = (tid >> 6) * 2048u, idle_naps = 0, idle_t0 = 0;  // poll_ticks paces the reads of the host's stop word (see driver 2)
#pragma unroll 1
    for (;;) {
        int sel = 0; bool finished = false, mine = false; uint32_t j = 0;
#pragma unroll 1
        for (;;) {                                            // ---- the claim: snapshot, vote, one atomic AND per lane
            const uint32_t m0 = __atomic_load_n(&q_mask[0][lane], __ATOMIC_RELAXED), m1 = __atomic_load_n(&q_mask[1][lane], __ATOMIC_RELAXED),
                           m2 = __atomic_load_n(&q_mask[2][lane], __ATOMIC_RELAXED), m3 = __atomic_load_n(&q_mask[3][lane], __ATOMIC_RELAXED);
            const uint32_t stop = (uint32_t) __builtin_amdgcn_readfirstlane((int) __atomic_load_n(&q_ctl[1], __ATOMIC_RELAXED));
            uint32_t best = 0;
#define MTS_VOTE(c, expr) do { const uint32_t v_ = (uint32_t) __popcll(__builtin_amdgcn_ballot_w64((expr) != 0u)); if (v_ > best) { best = v_; sel = (c); } } while (0)
            MTS_VOTE(0, m0 & 0xFFFFu); MTS_VOTE(1, m0 >> 16); MTS_VOTE(2, m1 & 0xFFFFu); MTS_VOTE(3, m1 >> 16);
            MTS_VOTE(4, m2 & 0xFFFFu); MTS_VOTE(5, m2 >> 16); MTS_VOTE(6, m3 & 0xFFFFu); MTS_VOTE(7, m3 >> 16);
#undef MTS_VOTE
            if (stop != STOP_NONE) best = 0;
            if (best == 0) {
                // nothing waits: every path has finished, or is being executed by another wave, or the workgroup was stopped
                if (stop != STOP_NONE || (uint32_t) __builtin_amdgcn_readfirstlane((int) __atomic_load_n(&q_ctl[0], __ATOMIC_RELAXED)) == (uint32_t) WG) { finished = true; break; }
                if ((poll_ticks += 1u) >= 32768u) {           // Integrator::should_stop(): see driver 2 for the pacing
                    poll_ticks = 0;
                    if (lane == 0 && __hip_atomic_load(cload_k<WgArgs>(kernarg).stop_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u)
                        atomicCAS(&q_ctl[1], (uint32_t) STOP_NONE, (uint32_t) STOP_CANCEL);
                }
                if (wga_idle_expired(idle_naps, idle_t0, MTS_IDLE_TICKS)) {   // a path was lost: report instead of waiting for ever
                    if (lane == 0 && atomicCAS(&q_ctl[1], (uint32_t) STOP_NONE, (uint32_t) STOP_STALL) == STOP_NONE) {
                        unsigned long long *counters = cload_k<WgArgs>(kernarg).counters;
                        if (atomicCAS(counters + MTS_DIAG_BASE, 0ull, 3ull) == 0ull) {
                            counters[MTS_DIAG_BASE + 1] = 0ull; counters[MTS_DIAG_BASE + 2] = 0ull; counters[MTS_DIAG_BASE + 3] = 0ull;
                            counters[MTS_DIAG_BASE + 4] = __atomic_load_n(&q_ctl[0], __ATOMIC_RELAXED); counters[MTS_DIAG_BASE + 5] = blockIdx.x;
                        }
                    }
                }
                __builtin_amdgcn_s_sleep(2);
#if defined(MTSAMD_BLOCKSTATS)
                if (COUNT) { long long t = clock64(); bs_loc[42] += (unsigned long long) (t - bs_t0); bs_t0 = t; }
#endif
                continue;
            }
            idle_naps = 0;
#if defined(MTSAMD_BLOCKSTATS)
            if (COUNT) bs_loc[10] += 1ull;                    // claim attempts
#endif
            const uint32_t w = sel < 4 ? (sel < 2 ? m0 : m1) : (sel < 6 ? m2 : m3), sh = 16u * ((uint32_t) sel & 1u);
            uint32_t f = (w >> sh) & 0xFFFFu;
            uint32_t *word = &q_mask[sel >> 1][lane];
#pragma nounroll
            while (__builtin_amdgcn_ballot_w64(f != 0u) != 0ull) {      // one round unless another wave took the same bit
                if (f != 0u) {
                    const uint32_t fr = f & (0xFFFFu << rot), src = fr != 0u ? fr : f;
                    j = (uint32_t) __builtin_ctz(src);
                    const uint32_t bit = 1u << (j + sh);
                    const uint32_t old = atomicAnd(word, ~bit);
                    if (old & bit) { mine = true; f = 0u; }
                    else f = (old >> sh) & 0xFFFFu;
                }
#if defined(MTSAMD_BLOCKSTATS)
                if (COUNT) bs_loc[11] += 1ull;                // rounds of the per-lane take
#endif
            }
            if (__builtin_amdgcn_ballot_w64(mine) != 0ull) break;
        }
        if (finished) break;
        const uint32_t pid = lane + 64u * j;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#if defined(MTSAMD_BLOCKSTATS)
        if (COUNT) { bs_loc[sel] += 1ull; bs_loc[12 + sel] += (unsigned long long) __popcll(__builtin_amdgcn_ballot_w64(mine));
                     long long t = clock64(); bs_loc[44] += (unsigned long long) (t - bs_t0); bs_t0 = t; }
#endif
        int cls = B_DONE;
        if (mine) cls = M::block(sel, kernarg, hot_lds, wg_base, pid, &cnt);
#if defined(MTSAMD_BLOCKSTATS)
        if (COUNT) { long long t = clock64(); bs_loc[24 + sel] += (unsigned long long) (t - bs_t0); bs_t0 = t; }
#endif
        if (sel == B_NEW && (poll_ticks += 256u) >= 32768u) {
            poll_ticks = 0;
            if (lane == 0 && __hip_atomic_load(cload_k<WgArgs>(kernarg).stop_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u)
                atomicCAS(&q_ctl[1], (uint32_t) STOP_NONE, (uint32_t) STOP_CANCEL);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (mine && cls != B_DONE) atomicOr(&q_mask[cls >> 1][lane], 1u << (j + 16u * ((uint32_t) cls & 1u)));
        const unsigned long long dm = __builtin_amdgcn_ballot_w64(mine && cls == B_DONE);
        if (dm != 0ull && lane == (uint32_t) __builtin_ctzll(dm)) atomicAdd(&q_ctl[0], (uint32_t) __popcll(dm));
#if defined(MTSAMD_BLOCKSTATS)
        if (COUNT) { long long t = clock64(); bs_loc[43] += (unsigned long long) (t - bs_t0); bs_t0 = t; }
#endif
    }
#if defined(MTSAMD_BLOCKSTATS)
    if (COUNT) {
        long long t = clock64(); bs_loc[44] += (unsigned long long) (t - bs_t0);
        for (int k = 0; k < 6; ++k) bs_loc[36 + k] = cnt.seg[k];
        if (lane == 0) for (int k = 0; k < 45; ++k) atomicAdd(&g_blockstats[k], bs_loc[k]);
    }
#endif
    // ---- stopped (cancel / timeout / stall): the samples the unfinished pixels have accumulated go to the film
    __syncthreads();
    if (__atomic_load_n(&q_ctl[1], __ATOMIC_RELAXED) != STOP_NONE) wg_flush_unfinished<WG, NT>(kernarg, hot_lds, M::PACKED_AT, wg_base);
}

// The volpath machine on driver 3
template <bool COUNT, int WG>
struct VolpathLanes {
    static constexpr int HOT = H_COUNT, PACKED_AT = H_PACKED;
    DEV static int init(const MTS_CONST_AS void *kernarg, uint32_t *hot_lds, uint32_t wg_base, uint32_t pid0, Counters *cnt) {
        const WgArgs a = cload_k<WgArgs>(kernarg);
        VolpathMachine<COUNT> vm(a.sc, *cnt);
        PathEnvT<ColdStoreHbm> e; PathState p;
        HotStore<WG> hs; hs.base = hot_lds + pid0;
        const bool ok = wg_env<WG>(a, wg_base, pid0, e);
        p.rng.state = 0; p.rng.inc = 0;
        p.ray = make_ray(f3s(0.f), f3(0.f, 0.f, 1.f), 0.f, 0.f); p.si.t = pm_inf(); p.si.p = f3s(0.f); p.si.uv.x = p.si.uv.y = 0.f; p.si.shape = -1; p.si.prim = 0;
        p.medium = -1; p.thr = p.res = p.trans = spec_s(0.f); p.eta = 1.f; p.depth = 0; p.channel = 0; p.mode = M_MAIN; p.flags = 0; p.wa = p.wb = 0.f;
#if MTS_SPEC_N != 3
        p.wl = spec_s(0.f);
#endif
        p.st = S_DONE;
        if (ok) {
            const uint32_t ppb = a.block_size * a.block_size;
            p.rng.seed(a.sc.sensor.seed + (uint64_t) e.blk.id * ppb + e.index, PCG32_DEFAULT_STREAM);     // sampler.cpp:83-96
            for (int k = 0; k < 5; ++k) e.cold.f(C_ACC + k) = 0.f;
            e.cold.f(C_SAMPLE) = __uint_as_float(0u);
            vm.begin_sample(p, e);
            vm.top(p, e);
        }
        const int cls = vm.classify(p);
        hs.store(p, cls);
        return cls;
    }
    DEV static int block(int sel, const MTS_CONST_AS void *kernarg, uint32_t *hot_lds, uint32_t wg_base, uint32_t pid, Counters *cnt) {
        switch (sel) {                                          // wave-uniform
            case B_INT: return wg_block<COUNT, WG, B_INT>(kernarg, hot_lds, wg_base, pid, cnt);
            case B_MED: return wg_block<COUNT, WG, B_MED>(kernarg, hot_lds, wg_base, pid, cnt);
            case B_MEDW: return wg_block<COUNT, WG, B_MEDW>(kernarg, hot_lds, wg_base, pid, cnt);
            case B_SCATTER: return wg_block<COUNT, WG, B_SCATTER>(kernarg, hot_lds, wg_base, pid, cnt);
            case B_WSURF: return wg_block<COUNT, WG, B_WSURF>(kernarg, hot_lds, wg_base, pid, cnt);
            case B_SURF: return wg_block<COUNT, WG, B_SURF>(kernarg, hot_lds, wg_base, pid, cnt);
            case B_PHASE: return wg_block<COUNT, WG, B_PHASE>(kernarg, hot_lds, wg_base, pid, cnt);
            default: return wg_block<COUNT, WG, B_NEW>(kernarg, hot_lds, wg_base, pid, cnt);
        }
    }
};

} // inline namespace
} // namespace mtsamd
