// volpathmis_flat.h -- volpathmis (integrators/volpathmis.cpp:86-445) on the regrouping machinery of volpath_flat.h.
//
// The spectral-MIS volumetric path tracer carries probability-ratio matrices instead of a throughput: p_over_f / p_over_f_nee along
// the path (volpathmis.cpp:117-118), and two more, started from p_over_f, along every emitter-sampling walk (sample_emitter,
// :330-445).  As in volpath_flat.h the three nested loops become one state machine -- a path is (mode, state) and advances one block
// at a time: INTERSECT, MEDIUM step of the path, MEDIUM step of a walk, SCATTER (emitter sampling at a medium interaction), walk
// SURFACE, path SURFACE + BSDF, PHASE, NEW sample -- and the workgroup regroups its paths by the block they wait for through the LDS
// rings of volpath_flat.h (same protocol: wga_push, wga_tag_wait, wga_raise_stop).
// Hot state (round 4): TWO matrix slots per path, not four.  The path's pair (p_over_f, p_over_f_nee) is only read and written
// outside a walk, the walk's pair (p_over_f_uni, p_over_f_nee of sample_emitter) only inside one, and a walk starts from a copy of
// p_over_f (:345-346): start_walk PARKS the path's pair in a second 128-byte cold record (PathEnvT::park, HBM), the two slots then
// hold the walk's pair, and end_nee fetches the path's pair back.  49 hot dwords per path with the 3 x 3 matrices (67 before), 69 with
// the 4 x 4 matrices of the spectral build (MTS_SPEC_N = 4, volpathmis.cpp:66-69; 101 before): two 256-path workgroups per CU
// instead of one, i.e. twice the paths behind the same eight waves.
// The draws happen in the order of the nested formulation (integrator_dev.h, volpathmis_sample) -- results are bit-identical to it
// and to the CPU restatement.  Citations are relative to /root/reference.
#pragma once
#include "volpath_flat.h"

namespace mtsamd {
inline namespace MTS_VARIANT_NS {

// cold record: (delta ? 0 : phase / bsdf pdf) of the pending emitter sample; C_CW holds the phase / bsdf value.  The spectral build's
// record is full (C_COUNT = 32); a walk parks either the scattering position (C_SO) or the surface hit (C_SHIT), never both, so the
// pdf takes the first slot of the one that is free.
#if MTS_SPEC_N == 3
DEV int c_pdfv(bool) { return 30; }
#else
DEV int c_pdfv(bool from_medium) { return from_medium ? C_SHIT : C_SO; }
#endif

template <bool SPEC>
struct MisPathState {
    Pcg32 rng;
    DRay ray; Hit si; int medium;
    MisWeights<SPEC> pf, pn;       // path: p_over_f, p_over_f_nee (volpathmis.cpp:117-118)
    MisWeights<SPEC> wn, wu;       // walk: p_over_f_nee / p_over_f_uni of sample_emitter (:345-346)
    Spec res; F3 lsp;              // result; last_scatter_event.p (:131-132: only the position is read)
#if MTS_SPEC_N != 3
    Spec wl;                       // the sample's wavelengths (integrator.cpp:252)
#endif
    float eta, wa, wb;             // walk: wa = total_dist, wb = ds.dist
    uint32_t depth, channel, st, mode, flags;
};

template <bool COUNT, bool SPEC>
struct VolpathMisMachine {
    typedef MisPathState<SPEC> P;
    typedef MisWeights<SPEC> W;
    const DScene &sc;
    Counters &cnt;
    DEV VolpathMisMachine(const DScene &sc_, Counters &cnt_) : sc(sc_), cnt(cnt_) {}
#if MTS_SPEC_N == 3
    DEV SpecCtx ctx(const P &) const { return SpecCtx(); }
#else
    DEV SpecCtx ctx(const P &p) const { SpecCtx cx = make_ctx(sc); cx.wl = p.wl; return cx; }
#endif

    DEV void queue_intersection(P &p) const {                  // as VolpathMachine::queue_intersection
        float bmint, bmaxt;
        bbox_ray_intersect(sc.bbox, p.ray, bmint, bmaxt);
        p.si.t = pm_inf();
        if (pm_max(p.ray.mint, bmint) <= bmaxt) p.flags |= FL_NEEDS_INT; else p.flags &= ~FL_NEEDS_INT;
    }
    // the parked pair: one 128-byte record per path behind the cold records, written and read by one lane as whole lines
    template <class E> DEV static void park_put(const E &e, const W &a, const W &b) {
        constexpr int NR = SPEC ? MTS_SPEC_N : 1;
        for (int i = 0; i < NR; ++i) {
            MTS_GLOBAL_AS float *pa = e.park + MTS_SPEC_N * i, *pb = e.park + MTS_SPEC_N * (NR + i);
            pa[0] = a.r[i].x; pa[1] = a.r[i].y; pa[2] = a.r[i].z; pb[0] = b.r[i].x; pb[1] = b.r[i].y; pb[2] = b.r[i].z;
#if MTS_SPEC_N != 3
            pa[3] = a.r[i].w; pb[3] = b.r[i].w;
#endif
        }
    }
    template <class E> DEV static void park_get(const E &e, W &a, W &b) {
        constexpr int NR = SPEC ? MTS_SPEC_N : 1;
        for (int i = 0; i < NR; ++i) {
            const MTS_GLOBAL_AS float *pa = e.park + MTS_SPEC_N * i, *pb = e.park + MTS_SPEC_N * (NR + i);
            a.r[i].x = pa[0]; a.r[i].y = pa[1]; a.r[i].z = pa[2]; b.r[i].x = pb[0]; b.r[i].y = pb[1]; b.r[i].z = pb[2];
#if MTS_SPEC_N != 3
            a.r[i].w = pa[3]; b.r[i].w = pb[3];
#endif
        }
    }
    DEV bool walk_goes_on(const P &p) const {                  // volpathmis.cpp:438-441
        if (SPEC) return any_nonzero(mis_weight_w(p.wu));
        return any_nonzero(p.wu.r[0]) || any_nonzero(p.wn.r[0]);
    }
    template <class E> DEV void begin_sample(P &p, const E &e) const {      // integrator.cpp:242-264, volpathmis.cpp:98-132
        const DSensor &se = sc.sensor;
        const float px = (float) (e.lx + (uint32_t) e.blk.ox), py = (float) (e.ly + (uint32_t) e.blk.oy);
        if (se.wavefront) seed_wavefront_sample(p.rng, se, e.blk, e.lx, e.ly, __float_as_uint(e.cold.f(C_SAMPLE)));   // gpu_* streams (volpath_flat.h)
        F2 u = p.rng.next_2d();
        F2 position_sample; position_sample.x = px + u.x; position_sample.y = py + u.y;
        F2 aperture_sample; aperture_sample.x = .5f; aperture_sample.y = .5f;
        if (se.needs_aperture_sample) aperture_sample = p.rng.next_2d();
        if (se.shutter_open_time > 0.f) (void) p.rng.next_1d();
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
        e.cold.f(C_POS) = position_sample.x; e.cold.f(C_POS + 1) = position_sample.y; e.cold.f(C_RAYW) = rw.x;
        p.medium = se.medium;
        p.res = spec_s(0.f); p.lsp = f3s(0.f); p.eta = 1.f; p.depth = 0;
        p.pf = mw_full<SPEC>(1.f); p.pn = mw_full<SPEC>(1.f); p.wn = mw_full<SPEC>(1.f); p.wu = mw_full<SPEC>(1.f);
#if MTS_SPEC_N == 3
        p.channel = sc.integrator.monochrome ? 0u : (uint32_t) pm_min(p.rng.next_1d() * 3.f, 2.f);     // volpathmis.cpp:120-124
#else
        p.channel = 0u;                                        // :120-124: a draw in the rgb variants only
#endif
        p.si.p = f3s(0.f); p.si.uv.x = p.si.uv.y = 0.f; p.si.prim = 0; p.si.shape = -1;
        const bool hide_emitters = sc.integrator.hide_emitters != 0;
        p.flags = FL_ALIVE | ((!hide_emitters && sc.environment >= 0) ? FL_VALID_RAY : 0u) | (!hide_emitters ? FL_SPEC_CHAIN : 0u);
        p.wa = p.wb = 0.f;
        queue_intersection(p);
        p.mode = M_MAIN; p.st = S_TOP;
    }
    // emitter-sampling walk finished (volpathmis.cpp:443-444 + :233-236 / :297-299): MIS-weighted contribution, resume the path
    template <class E> DEV void end_nee(P &p, const E &e) const {
        const Spec fval = e.cold.get_spec(C_CW), emitted = e.cold.get_spec(C_EMIT);
        const float pdfv = e.cold.f(c_pdfv((p.flags & FL_FROM_MEDIUM) != 0));
        update_weights(p.wn, 1.0f, fval, p.channel, true);
        update_weights(p.wu, pdfv, fval, p.channel, true);
        p.res = p.res + mis_weight_w(p.wn, p.wu) * emitted;
        park_get(e, p.pf, p.pn);                               // ... and comes back
        p.mode = M_MAIN; p.medium = __float_as_int(e.cold.f(C_SMED));
        F3 d = e.cold.get3(C_SD);
        p.ray.d = d; p.ray.d_rcp = vrcp(d);
        if (p.flags & FL_FROM_MEDIUM) { p.ray.o = e.cold.get3(C_SO); p.st = S_PHASE; }
        else { p.si = e.cold.get_hit(); p.st = S_BSDF; }
    }
    // A block that runs on a partial state (MisClassFields) leaves the end of a walk to finish(), which runs on the full state
    template <bool DEFER, class E> DEV void nee_done(P &p, const E &e) const { if (DEFER) p.st = S_ENDNEE; else end_nee(p, e); }
    template <class E> DEV void finish(P &p, const E &e) const { if (p.st == S_ENDNEE) end_nee(p, e); }
    // loop heads: volpathmis.cpp:134-151 (path), :358-362 (walk)
    template <bool DEFER = false, class E> DEV void top(P &p, const E &e) const {
        if (p.st != S_TOP) return;
        const uint32_t max_depth = (uint32_t) sc.integrator.max_depth, rr_depth = (uint32_t) sc.integrator.rr_depth;
        if (p.mode == M_MAIN) {
            bool active = (p.flags & FL_ALIVE) != 0;
            Spec mis_throughput = mis_weight_w(p.pf);
            float q = pm_min(hmax(mis_throughput) * (p.eta * p.eta), .95f);
            bool perform_rr = active && (p.depth > rr_depth);                          // last_event_was_null is never set (:146)
            active = active && !(p.rng.next_1d() >= q && perform_rr);
            update_weights(p.pf, q, 1.0f, p.channel, perform_rr);
            active = active && !(p.depth >= max_depth);
            active = active && any_nonzero(mis_weight_w(p.pf));
            if (!active) p.st = S_NEW;
            else { if (COUNT) cnt.n_iter++; p.st = p.medium >= 0 ? S_MED : S_SURF; }
        } else {
            float remaining_dist = p.wb * (1.f - MTS_SHADOW_EPSILON) - p.wa;
            p.ray.maxt = remaining_dist;
            if (!(remaining_dist > 0.f)) nee_done<DEFER>(p, e);
            else { if (COUNT) cnt.n_nee_step++; p.st = p.medium >= 0 ? S_MED : S_SURF; }
        }
    }
    DEV static bool wants_int(const P &p) { return (p.st == S_MED || p.st == S_SURF) && (p.flags & FL_NEEDS_INT); }
    DEV static int classify(const P &p) {
        if (p.st == S_DONE) return B_DONE;
        if (wants_int(p)) return B_INT;
        if (p.st == S_MED) return p.mode == M_MAIN ? B_MED : B_MEDW;
        if (p.st == S_SCATTER) return B_SCATTER;
        if (p.st == S_SURF) return p.mode == M_MAIN ? B_SURF : B_WSURF;
        if (p.st == S_BSDF) return B_SURF;
        if (p.st == S_PHASE) return B_PHASE;
        return B_NEW;
    }

    // ================================================================= NEW (integrator.cpp:265-288)
    template <class E> DEV void blk_new(P &p, const E &e) const {
        if (p.st != S_NEW) return;
        const DSensor &se = sc.sensor;
        F2 position_sample; position_sample.x = e.cold.f(C_POS); position_sample.y = e.cold.f(C_POS + 1);
        float acc[5];
        for (int k = 0; k < 5; ++k) acc[k] = e.cold.f(C_ACC + k);
#if MTS_SPEC_N == 3
        splat_sample_t<false>(sc, e.blk, e.lx, e.ly, position_sample, f3s(e.cold.f(C_RAYW)) * p.res, (p.flags & FL_VALID_RAY) != 0, e.film, acc);
#else
        {
            float wav_weight; (void) sample_wavelengths(0.f, wav_weight);
            const Spec ww = sc.srf >= 0 ? srf_weights_of(sc, p.wl) : spec_s(wav_weight);      // a sensor response function: weights recovered from the wavelengths (volpath_flat.h)
            const Spec L = (ww * e.cold.f(C_RAYW)) * p.res;             // ray_weight = wav_weight (x the sensor's grey weight), integrator.cpp:265
            float xyz[3];
            spectrum_to_xyz(sc.cie, L, p.wl, xyz);                      // integrator.cpp:266-269
            const float v[5] = { xyz[0], xyz[1], xyz[2], (p.flags & FL_VALID_RAY) != 0 ? 1.f : 0.f, 1.f };
            if (sc.bin_count == 0) splat_values_t<false>(sc, e.blk, e.lx, e.ly, position_sample, v, e.film, acc);
            else splat_values_bins(sc, e.blk, e.lx, e.ly, position_sample, v, p.res, p.wl, e.film, acc);      // nbins / bins around volpathmis (round 4)
        }
#endif
        const uint32_t sample_idx = __float_as_uint(e.cold.f(C_SAMPLE)) + 1u;
        if (sample_idx == e.sample_count) {
            float *own = (float *) (e.film + MTS_FILM_STRIDE(sc) * ((size_t) (e.blk.oy + (int) e.ly - se.crop_y) * se.crop_w + (e.blk.ox + (int) e.lx - se.crop_x)));
            for (int k = 0; k < 5; ++k) atomicAdd(own + k, acc[k]);
            p.st = S_DONE;
        } else {
            for (int k = 0; k < 5; ++k) e.cold.f(C_ACC + k) = acc[k];
            e.cold.f(C_SAMPLE) = __uint_as_float(sample_idx);
            begin_sample(p, e);
        }
    }
    // ================================================================= INTERSECT
    template <class E> DEV void blk_int(P &p, const E &) const {
        if (!wants_int(p)) return;
        p.si = ray_intersect(sc, p.ray);
        p.flags &= ~FL_NEEDS_INT;
    }
    // ================================================================= MEDIUM step of the path (volpathmis.cpp:165-245)
    template <class E> DEV void blk_med(P &p, const E &) const {
        if (p.st != S_MED || p.mode != M_MAIN || (p.flags & FL_NEEDS_INT)) return;
        const uint32_t max_depth = (uint32_t) sc.integrator.max_depth, channel = p.channel;
        const float u = p.rng.next_1d();
        MedStep mi;
        const SpecCtx cx = ctx(p); (void) cx;
        WATERFALL_BEGIN(p.medium, mu)
            mi = medium_step<COUNT>(sc, cload(sc.media + mu), p.ray, u, channel, true, cnt MTS_CX);
        WATERFALL_END
        if (p.si.t < mi.t) mi.t = pm_inf();
#if MTS_TRAITS & MT_MEDIA
        const bool spectral = true, homogeneous = false, grey = MTS_SPEC_N == 3;
#elif MTS_TRAITS & MT_HOMOG
        const bool spectral = (mi.info & MI_SPECTRAL) != 0, homogeneous = true, grey = (mi.info & MI_GREY) != 0;
#else
        const bool spectral = (mi.info & MI_SPECTRAL) != 0, homogeneous = (mi.info & MI_HOMOGENEOUS) != 0, grey = (mi.info & MI_GREY) != 0;
#endif
        const Spec sigma_n = homogeneous ? spec_s(0.f) : mi.combined - mi.sigma_t;
        if (spectral) {
            float t = pm_min(mi.t, p.si.t) - mi.mint;                                  // medium.cpp:77-89
            // a heterogeneous medium's combined extinction is its scalar majorant (heterogeneous.cpp:29): one exponential for every channel
            if (homogeneous) {
                Spec tr = transmittance_exp(t, mi.combined);
                Spec free_flight_pdf = p.si.t < mi.t ? tr : tr * mi.combined;
                update_weights(p.pf, free_flight_pdf, tr, channel, true);
                update_weights(p.pn, free_flight_pdf, tr, channel, true);
            } else {
                const float tr = pm_exp(-t * mi.combined.x), free_flight_pdf = p.si.t < mi.t ? tr : tr * mi.combined.x;
                update_weights_uniform(p.pf, free_flight_pdf, tr);
                update_weights_uniform(p.pn, free_flight_pdf, tr);
            }
        }
        if (mi.t == pm_inf()) { p.st = S_SURF; return; }                               // escaped_medium: the surface part of this iteration
        const bool null_scatter = p.rng.next_1d() >= div_by_invariant(pick(mi.sigma_t, channel), pick(mi.combined, channel), mi.inv_combined);
        if (null_scatter) {
            if (spectral && grey) {
                update_weights_uniform(p.pf, div_by_invariant(sigma_n.x, mi.combined.x, mi.inv_combined), sigma_n.x);
                update_weights_uniform(p.pn, 1.0f, sigma_n.x);
            } else if (spectral) {
                update_weights(p.pf, div_by_invariant(sigma_n, mi.combined, mi.inv_combined), sigma_n, channel, true);
                update_weights(p.pn, 1.0f, sigma_n, channel, true);
            } else {
                update_weights(p.pf, sigma_n, sigma_n, channel, true);
                update_weights(p.pn, 1.0f, div_by_invariant(sigma_n, mi.combined, mi.inv_combined), channel, true);
            }
            p.ray.o = mi.p; p.ray.mint = 0.f; p.si.t = p.si.t - mi.t;
            p.st = S_TOP;
            return;
        }
        p.depth += 1; p.lsp = mi.p;
        const bool sample_emitters = (mi.info & MI_SAMPLE_EMITTERS) != 0;
        if (!(p.depth < max_depth)) { p.flags &= ~FL_ALIVE; p.st = S_TOP; return; }    // :197-198: the path ends at the next loop head
        if (sample_emitters) p.flags &= ~FL_SPEC_CHAIN;                                // :199
        if (spectral && grey) update_weights_uniform(p.pf, div_by_invariant(mi.sigma_t.x, mi.combined.x, mi.inv_combined), mi.sigma_s.x);
        else if (spectral) update_weights(p.pf, div_by_invariant(mi.sigma_t, mi.combined, mi.inv_combined), mi.sigma_s, channel, true);
        else update_weights(p.pf, mi.sigma_t, mi.sigma_s, channel, true);
        p.flags |= FL_VALID_RAY;
        p.ray.o = mi.p;                                                                // scattering position; ray.d stays the incident direction
        p.st = sample_emitters ? S_SCATTER : S_PHASE;
    }
    // ================================================================= SCATTER: emitter sampling at a medium interaction (:228-237 -> :330-356)
    template <bool DEFER, class E> DEV void start_walk(P &p, const E &e, F3 ref_p, const DirSample &ds, Spec emitter_sample_weight, Spec fval, float pdfv, bool from_medium) const {
        park_put(e, p.pf, p.pn);                               // the path's pair leaves the hot state for the duration of the walk
        p.wn = p.pf; p.wu = p.pf;
        Spec emitter_val = emitter_sample_weight * ds.pdf;
        if (ds.pdf == 0.f) emitter_val = spec_s(0.f);
        const bool active = ds.pdf != 0.f;
        update_weights(p.wn, ds.pdf, 1.0f, p.channel, active);
        e.cold.put_spec(C_CW, fval); e.cold.put_spec(C_EMIT, emitter_val);
        e.cold.put3(C_SD, p.ray.d); e.cold.f(C_SMED) = __int_as_float(p.medium);
        if (from_medium) { e.cold.put3(C_SO, p.ray.o); p.flags |= FL_FROM_MEDIUM; }
        else { e.cold.put_hit(p.si); p.flags &= ~FL_FROM_MEDIUM; }
        e.cold.f(c_pdfv(from_medium)) = pdfv;
        p.mode = M_NEE;
        if (!active) { nee_done<DEFER>(p, e); return; }
        p.wa = 0.f; p.wb = ds.dist;
        p.ray = spawn_ray(ref_p, ds.d);
        if (from_medium) p.ray.mint = 0.f;
        queue_intersection(p);
        p.st = S_TOP;
    }
    template <bool DEFER = false, class E> DEV void blk_scatter(P &p, const E &e) const {
        if (p.st != S_SCATTER) return;
        Spec esw;
        const SpecCtx cx = ctx(p); (void) cx;
        DirSample ds = sample_emitter_direction(sc, p.ray.o, p.rng.next_2d(), false, esw MTS_CX);
        float phase_val = 0.f;
        WATERFALL_BEGIN(p.medium, mu)
            phase_val = phase_eval<true>(sc, cload(sc.media + mu).phase, -p.ray.d, p.ray.o, ds.d MTS_CX);
        WATERFALL_END
        start_walk<DEFER>(p, e, p.ray.o, ds, esw, spec_s(phase_val), ds.delta ? 0.f : phase_val, true);
    }
    // ================================================================= MEDIUM step of a walk (volpathmis.cpp:364-411)
    template <bool DEFER = false, class E> DEV void blk_medw(P &p, const E &e) const {
        if (p.st != S_MED || p.mode == M_MAIN || (p.flags & FL_NEEDS_INT)) return;
        const uint32_t channel = p.channel;
        const float u = p.rng.next_1d();
        MedStep mi;
        const SpecCtx cx = ctx(p); (void) cx;
        WATERFALL_BEGIN(p.medium, mu)
            mi = medium_step<COUNT>(sc, cload(sc.media + mu), p.ray, u, channel, false, cnt MTS_CX);
        WATERFALL_END
        if (p.si.t < mi.t) mi.t = pm_inf();
#if MTS_TRAITS & MT_MEDIA
        const bool spectral = true, homogeneous = false, grey = MTS_SPEC_N == 3;
#elif MTS_TRAITS & MT_HOMOG
        const bool spectral = (mi.info & MI_SPECTRAL) != 0, homogeneous = true, grey = (mi.info & MI_GREY) != 0;
#else
        const bool spectral = (mi.info & MI_SPECTRAL) != 0, homogeneous = (mi.info & MI_HOMOGENEOUS) != 0, grey = (mi.info & MI_GREY) != 0;
#endif
        const Spec sigma_n = homogeneous ? spec_s(0.f) : mi.combined - mi.sigma_t;
        const float remaining_dist = p.ray.maxt;
        if (spectral) {
            float t = pm_min(remaining_dist, pm_min(mi.t, p.si.t)) - mi.mint;
            // a heterogeneous medium's combined extinction is its scalar majorant (heterogeneous.cpp:29): one exponential for every channel
            const bool no_event = p.si.t < mi.t || mi.t > remaining_dist;
            if (homogeneous) {
                Spec tr = transmittance_exp(t, mi.combined);
                Spec free_flight_pdf = no_event ? tr : tr * mi.combined;
                update_weights(p.wn, free_flight_pdf, tr, channel, true);
                update_weights(p.wu, free_flight_pdf, tr, channel, true);
            } else {
                const float tr = pm_exp(-t * mi.combined.x), free_flight_pdf = no_event ? tr : tr * mi.combined.x;
                update_weights_uniform(p.wn, free_flight_pdf, tr);
                update_weights_uniform(p.wu, free_flight_pdf, tr);
            }
        }
        if (mi.t > remaining_dist && mi.t != pm_inf()) p.wa = p.wb;
        if (mi.t > remaining_dist) mi.t = pm_inf();
        if (mi.t == pm_inf()) { p.st = S_SURF; return; }                               // escaped_medium
        p.wa += mi.t;
        p.ray.o = mi.p; p.ray.mint = 0.f; p.si.t = p.si.t - mi.t;
        if (spectral && grey) {
            update_weights_uniform(p.wn, 1.f, sigma_n.x);
            update_weights_uniform(p.wu, div_by_invariant(sigma_n.x, mi.combined.x, mi.inv_combined), sigma_n.x);
        } else if (spectral) {
            update_weights(p.wn, 1.f, sigma_n, channel, true);
            update_weights(p.wu, div_by_invariant(sigma_n, mi.combined, mi.inv_combined), sigma_n, channel, true);
        } else {
            update_weights(p.wn, 1.f, div_by_invariant(sigma_n, mi.combined, mi.inv_combined), channel, true);
            update_weights(p.wu, sigma_n, sigma_n, channel, true);
        }
        if (walk_goes_on(p)) p.st = S_TOP; else nee_done<DEFER>(p, e);
    }
    // ================================================================= SURFACE step of a walk (volpathmis.cpp:413-441)
    template <class E> DEV void blk_wsurf(P &p, const E &e) const {
        if (p.st != S_SURF || p.mode == M_MAIN || (p.flags & FL_NEEDS_INT)) return;
        const bool hit = hit_valid(p.si);
        p.wa += p.si.t;
        if (!hit) { end_nee(p, e); return; }
        Spec nt = spec_s(0.f); F3 n = f3s(0.f); int is_tr = 0, ext = -1, inte = -1;
        WATERFALL_BEGIN(p.si.shape, su)
            const DShape s = cload(sc.shapes + su);
            nt = s.bsdf_type == MTS_BSDF_NULL ? spec_s(1.f) : spec_s(0.f);
            is_tr = s.is_medium_transition; ext = s.exterior; inte = s.interior;
            if (is_tr) n = hit_geo_normal(sc, s, p.si);
        WATERFALL_END
        update_weights(p.wn, 1.0f, nt, p.channel, true);
        update_weights(p.wu, 1.0f, nt, p.channel, true);
        p.ray = spawn_ray(p.si.p, p.ray.d);
        queue_intersection(p);
        const bool go_on = walk_goes_on(p);
        if (is_tr) p.medium = dot(p.ray.d, n) > 0 ? ext : inte;
        if (go_on) p.st = S_TOP; else end_nee(p, e);
    }
    // ================================================================= SURFACE interaction of the path (volpathmis.cpp:253-300)
    template <class E> DEV void blk_surf(P &p, const E &e) const {
        if (p.st != S_SURF || p.mode != M_MAIN || (p.flags & FL_NEEDS_INT)) return;
        const uint32_t max_depth = (uint32_t) sc.integrator.max_depth;
        const bool hide_emitters = sc.integrator.hide_emitters != 0;
        const bool hit = hit_valid(p.si);
        Surf sf; sf.wi = -p.ray.d; sf.n = f3s(0.f); sf.sh.s = sf.sh.t = sf.sh.n = f3s(0.f);
        int emitter = sc.environment, bsdf_id = 0;
        if (hit) {
            WATERFALL_BEGIN(p.si.shape, su)
                const DShape s = cload(sc.shapes + su);
                emitter = s.emitter; bsdf_id = s.bsdf;
                complete_surface(sc, s, p.si, p.ray.d, sf);
            WATERFALL_END
        }
        const bool count_direct = p.depth == 0 || (p.flags & FL_SPEC_CHAIN);
        const SpecCtx cx = ctx(p); (void) cx;
        if (emitter >= 0 && !(p.depth == 0 && hide_emitters)) {
            if (!count_direct) {
                DirSample ds;                                                          // records.h:168-174
                ds.p = p.si.p; ds.n = sf.sh.n; ds.d = p.si.p - p.lsp; ds.dist = norm(ds.d); ds.d = ds.d / ds.dist;
                if (!hit) ds.d = -sf.wi;
                ds.emitter = emitter; ds.pdf = 0.f; ds.delta = false;
                float emitter_pdf = pdf_emitter_direction(sc, p.lsp, ds);
                update_weights(p.pn, emitter_pdf, 1.f, p.channel, true);
            }
            Spec emitted = emitter_eval(sc, emitter, sf.wi.z MTS_CX);
            p.res = p.res + (count_direct ? mis_weight_w(p.pf) * emitted : mis_weight_w(p.pf, p.pn) * emitted);
        }
        if (!hit) { p.flags &= ~FL_ALIVE; p.st = S_TOP; return; }
        p.st = S_BSDF;
        bool active_e = false;
        WATERFALL_BEGIN(bsdf_id, bu)
            active_e = (cload(sc.bsdfs + bu).flags & F_Smooth) != 0 && (p.depth + 1 < max_depth);
        WATERFALL_END
        if (!active_e) return;
        Spec esw;
        DirSample ds = sample_emitter_direction(sc, p.si.p, p.rng.next_2d(), false, esw MTS_CX);
        F3 wo = to_local(sf.sh, ds.d);
        Spec bsdf_val; float bpdf;
        WATERFALL_BEGIN(bsdf_id, bu)
            const DBsdf bsdf = cload(sc.bsdfs + bu);
            bsdf_val = bsdf_eval(bsdf, sf.wi, wo MTS_CXI(bu));
            bpdf = bsdf_pdf(bsdf, sf.wi, wo MTS_CXI(bu));
        WATERFALL_END
        start_walk<false>(p, e, p.si.p, ds, esw, bsdf_val, ds.delta ? 0.f : bpdf, false);
    }
    // ================================================================= BSDF sampling (volpathmis.cpp:302-328)
    template <class E> DEV void blk_bsdf(P &p, const E &) const {
        if (p.st != S_BSDF) return;
        Surf sf; int bsdf_id = 0, is_tr = 0, ext = -1, inte = -1;
        WATERFALL_BEGIN(p.si.shape, su)
            const DShape s = cload(sc.shapes + su);
            complete_surface(sc, s, p.si, p.ray.d, sf);
            bsdf_id = s.bsdf; is_tr = s.is_medium_transition; ext = s.exterior; inte = s.interior;
        WATERFALL_END
        const float s1 = p.rng.next_1d(); const F2 s2 = p.rng.next_2d();
        BSDFSample bs; Spec bsdf_weight;
        const SpecCtx cx = ctx(p); (void) cx;
        WATERFALL_BEGIN(bsdf_id, bu)
            const DBsdf bsdf = cload(sc.bsdfs + bu);
            bsdf_weight = bsdf_sample(bsdf, sf.wi, s1, s2, bs MTS_CXI(bu));
        WATERFALL_END
        const bool invalid_bsdf_sample = bs.pdf == 0.f;
        const bool active_surface = bs.pdf > 0.f;
        if (active_surface) p.eta *= bs.eta;
        if (active_surface) { p.ray = spawn_ray(p.si.p, to_world(sf.sh, bs.wo)); queue_intersection(p); }
        const bool non_null_bsdf = active_surface && !(bs.sampled_type & F_Null);
        if (non_null_bsdf || invalid_bsdf_sample) p.flags |= FL_VALID_RAY;
        if (non_null_bsdf && (bs.sampled_type & F_Delta)) p.flags |= FL_SPEC_CHAIN;
        if (active_surface && (bs.sampled_type & F_Smooth)) p.flags &= ~FL_SPEC_CHAIN;
        if (non_null_bsdf) { p.depth += 1; p.lsp = p.si.p; p.pn = p.pf; }
        update_weights(p.pf, bs.pdf, bsdf_weight * bs.pdf, p.channel, active_surface);
        update_weights(p.pn, 1.f, bsdf_weight * bs.pdf, p.channel, non_null_bsdf);
        if (active_surface && is_tr) p.medium = dot(p.ray.d, sf.n) > 0 ? ext : inte;
        if (!active_surface) p.flags &= ~FL_ALIVE;                                     // :327: active &= (active_surface | active_medium)
        p.st = S_TOP;
    }
    // ================================================================= PHASE sampling (volpathmis.cpp:239-251)
    template <class E> DEV void blk_phase(P &p, const E &) const {
        if (p.st != S_PHASE) return;
        p.pn = p.pf;                                                                   // :240: a real interaction resets p_over_f_nee
        const float s1 = p.rng.next_1d(); const F2 s2 = p.rng.next_2d();
        F3 wo; float phase_pdf = 0.f;
        const SpecCtx cx = ctx(p); (void) cx;
        WATERFALL_BEGIN(p.medium, mu)
            wo = phase_sample_pdf(sc, cload(sc.media + mu).phase, make_frame(p.ray.d), p.ray.o, s1, s2, phase_pdf MTS_CX);
        WATERFALL_END
        p.ray = spawn_ray(p.ray.o, wo); p.ray.mint = 0.0f;
        queue_intersection(p);
        update_weights(p.pf, phase_pdf, phase_pdf, p.channel, true);
        update_weights(p.pn, 1.f, phase_pdf, p.channel, true);
        p.st = S_TOP;
    }
    template <bool DEFER, class E> DEV void run(P &p, const E &e, int sel) const {
        switch (sel) {
            case B_NEW: blk_new(p, e); break;
            case B_INT: blk_int(p, e); break;
            case B_MED: blk_med(p, e); break;
            case B_MEDW: blk_medw<DEFER>(p, e); break;
            case B_SCATTER: blk_scatter<DEFER>(p, e); break;
            case B_WSURF: blk_wsurf(p, e); break;
            case B_SURF: blk_surf(p, e); blk_bsdf(p, e); break;
            case B_PHASE: blk_phase(p, e); break;
            default: break;
        }
    }
};

// Hot state in LDS, struct of arrays over the workgroup's paths (as HotStore of volpath_flat.h).  A block loads / stores only the field
// groups its class can read / write (MisClassFields); at two waves per SIMD the register budget of 256 VGPRs holds the whole state.
enum : uint32_t { K_RNG = 1, K_O = 2, K_D = 4 /* d and 1/d */, K_MINT = 8, K_MAXT = 16, K_SIT = 32, K_SIX = 64 /* rest of si */, K_MED = 128,
                  K_PF = 256, K_PN = 512, K_WN = 1024, K_WU = 2048, K_RES = 4096, K_LSP = 8192, K_ETA = 16384, K_WA = 32768, K_WB = 65536,
#if MTS_SPEC_N == 3
                  K_WL = 0, K_ALL = 131071 };
#else
                  K_WL = 131072 /* the sample's wavelengths: read by every block that evaluates a spectrum, written by NEW */, K_ALL = 262143 };
#endif
// What block class C (followed by top()) may read (`load`, a superset of `store`) and write (`store`); `defer`: the end of a walk,
// which touches almost everything, runs afterwards on the full state (VolpathMisMachine::finish).
template <int C> struct MisClassFields { static constexpr uint32_t load = K_ALL, store = K_ALL; static constexpr bool defer = false; };
template <> struct MisClassFields<B_INT> {
    static constexpr uint32_t load = K_O | K_D | K_MINT | K_MAXT, store = K_SIT | K_SIX; static constexpr bool defer = true; };
template <> struct MisClassFields<B_MED> {       // the path: p_over_f, p_over_f_nee; the loop head reads eta
    static constexpr uint32_t load = K_RNG | K_O | K_D | K_MINT | K_MAXT | K_SIT | K_MED | K_PF | K_PN | K_ETA | K_LSP | K_WL,
                              store = K_RNG | K_O | K_MINT | K_SIT | K_PF | K_PN | K_LSP;
    static constexpr bool defer = true; };
template <> struct MisClassFields<B_MEDW> {      // a walk: its two matrices and its distance budget
    static constexpr uint32_t load = K_RNG | K_O | K_D | K_MINT | K_MAXT | K_SIT | K_MED | K_WN | K_WU | K_WA | K_WB | K_WL,
                              store = K_RNG | K_O | K_MINT | K_MAXT | K_SIT | K_WN | K_WU | K_WA;
    static constexpr bool defer = true; };
template <> struct MisClassFields<B_SCATTER> {
    static constexpr uint32_t load = K_RNG | K_O | K_D | K_MINT | K_MAXT | K_SIT | K_MED | K_PF | K_PN /* parked by start_walk */ | K_WA | K_WB | K_WL,
                              store = K_RNG | K_O | K_D | K_MINT | K_MAXT | K_SIT | K_WN | K_WU | K_WA | K_WB;
    static constexpr bool defer = true; };
template <> struct MisClassFields<B_PHASE> {
    static constexpr uint32_t load = K_RNG | K_O | K_D | K_MINT | K_MAXT | K_SIT | K_MED | K_PF | K_ETA | K_WL,
                              store = K_RNG | K_O | K_D | K_MINT | K_MAXT | K_SIT | K_PF | K_PN;
    static constexpr bool defer = true; };

template <int WG, bool SPEC>
struct MisHotStore {
    static constexpr int NW = SPEC ? MTS_SPEC_N * MTS_SPEC_N : MTS_SPEC_N;      // floats per weight matrix
    enum { M_RNG = 0, M_O = 2, M_D = 5, M_DRCP = 8, M_MINT = 11, M_MAXT = 12, M_SIT = 13, M_MEDIUM = 14, M_PACKED = 15, M_WA = 16, M_WB = 17,
           M_ETA = 18, M_RES = 19, M_LSP = M_RES + MTS_SPEC_DW, M_SIX = M_LSP + 3,
#if MTS_SPEC_N == 3
           M_W = 32,
#else
           M_WL = M_SIX + 7, M_W = M_WL + 4,                    // 37
#endif
           M_COUNT = M_W + 2 * NW };            // two matrix slots: A = p_over_f or the walk's p_over_f_uni, B = p_over_f_nee of the path or of the walk
    uint32_t *base;
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
    DEV void putw(int k, const MisWeights<SPEC> &w) const { for (int i = 0; i < NW / MTS_SPEC_N; ++i) put_spec(k + MTS_SPEC_N * i, w.r[i]); }
    DEV MisWeights<SPEC> getw(int k) const { MisWeights<SPEC> w; for (int i = 0; i < NW / MTS_SPEC_N; ++i) w.r[i] = get_spec(k + MTS_SPEC_N * i); return w; }
    template <uint32_t M> DEV void store_m(const MisPathState<SPEC> &p, int cls) const {
        if (M & K_RNG) { u(M_RNG) = (uint32_t) p.rng.state; u(M_RNG + 1) = (uint32_t) (p.rng.state >> 32); }
        if (M & K_O) put3(M_O, p.ray.o);
        if (M & K_D) { put3(M_D, p.ray.d); put3(M_DRCP, p.ray.d_rcp); }
        if (M & K_MINT) putf(M_MINT, p.ray.mint);
        if (M & K_MAXT) putf(M_MAXT, p.ray.maxt);
        if (M & K_SIT) putf(M_SIT, p.si.t);
        if (M & K_SIX) { put3(M_SIX, p.si.p); putf(M_SIX + 3, p.si.uv.x); putf(M_SIX + 4, p.si.uv.y); u(M_SIX + 5) = (uint32_t) p.si.shape; u(M_SIX + 6) = (uint32_t) p.si.prim; }
        if (M & K_MED) u(M_MEDIUM) = (uint32_t) p.medium;
        u(M_PACKED) = p.st | (p.mode << 4) | (p.channel << 6) | (p.flags << 8) | ((uint32_t) cls << 13) | ((p.depth < 32767u ? p.depth : 32767u) << 17);
        if (M & K_WA) putf(M_WA, p.wa);
        if (M & K_WB) putf(M_WB, p.wb);
        if (M & K_ETA) putf(M_ETA, p.eta);
        if (M & K_RES) put_spec(M_RES, p.res);
        if (M & K_LSP) put3(M_LSP, p.lsp);
#if MTS_SPEC_N != 3
        if (M & K_WL) put_spec(M_WL, p.wl);
#endif
        // slot A / slot B: the path's pair outside a walk, the walk's pair inside one (a block that can hold either -- K_ALL -- picks by mode)
        constexpr bool path_a = (M & K_PF) != 0, walk_a = (M & K_WU) != 0, path_b = (M & K_PN) != 0, walk_b = (M & K_WN) != 0;
        if (path_a && walk_a) putw(M_W, p.mode == M_MAIN ? p.pf : p.wu); else if (path_a) putw(M_W, p.pf); else if (walk_a) putw(M_W, p.wu);
        if (path_b && walk_b) putw(M_W + NW, p.mode == M_MAIN ? p.pn : p.wn); else if (path_b) putw(M_W + NW, p.pn); else if (walk_b) putw(M_W + NW, p.wn);
    }
    template <uint32_t M> DEV void load_m(MisPathState<SPEC> &p) const {
        p.rng.state = 0; p.rng.inc = (PCG32_DEFAULT_STREAM << 1u) | 1u;
        if (M & K_RNG) p.rng.state = (uint64_t) u(M_RNG) | ((uint64_t) u(M_RNG + 1) << 32);
        p.ray.o = (M & K_O) ? get3(M_O) : f3s(0.f);
        p.ray.d = (M & K_D) ? get3(M_D) : f3s(0.f); p.ray.d_rcp = (M & K_D) ? get3(M_DRCP) : f3s(0.f);
        p.ray.mint = (M & K_MINT) ? f(M_MINT) : 0.f; p.ray.maxt = (M & K_MAXT) ? f(M_MAXT) : 0.f;
        p.si.t = (M & K_SIT) ? f(M_SIT) : pm_inf();
        p.si.p = f3s(0.f); p.si.uv.x = p.si.uv.y = 0.f; p.si.shape = -1; p.si.prim = 0;
        if (M & K_SIX) { p.si.p = get3(M_SIX); p.si.uv.x = f(M_SIX + 3); p.si.uv.y = f(M_SIX + 4); p.si.shape = (int) u(M_SIX + 5); p.si.prim = (int) u(M_SIX + 6); }
        p.medium = (M & K_MED) ? (int) u(M_MEDIUM) : -1;
        const uint32_t pk = u(M_PACKED);
        p.st = pk & 15u; p.mode = (pk >> 4) & 3u; p.channel = (pk >> 6) & 3u; p.flags = (pk >> 8) & 31u; p.depth = pk >> 17;
        p.wa = (M & K_WA) ? f(M_WA) : 0.f; p.wb = (M & K_WB) ? f(M_WB) : 0.f; p.eta = (M & K_ETA) ? f(M_ETA) : 1.f;
        p.res = (M & K_RES) ? get_spec(M_RES) : spec_s(0.f); p.lsp = (M & K_LSP) ? get3(M_LSP) : f3s(0.f);
#if MTS_SPEC_N != 3
        p.wl = (M & K_WL) ? get_spec(M_WL) : spec_s(0.f);
#endif
        // a block that may meet either pair (K_ALL) reads the two slots once; which pair they are follows from the mode, the other is never read
        const MisWeights<SPEC> slot_a = (M & (K_PF | K_WU)) ? getw(M_W) : mw_full<SPEC>(1.f), slot_b = (M & (K_PN | K_WN)) ? getw(M_W + NW) : mw_full<SPEC>(1.f);
        p.pf = slot_a; p.wu = slot_a; p.pn = slot_b; p.wn = slot_b;
    }
    DEV void store(const MisPathState<SPEC> &p, int cls) const { store_m<K_ALL>(p, cls); }
    DEV void load(MisPathState<SPEC> &p) const { load_m<K_ALL>(p); }
};

// One block of class C for the path `pid` (the class is wave-uniform, the blocks are inlined as in volpath_flat.h)
template <bool COUNT, bool SPEC, int WG, int C>
static __device__ __forceinline__ int mis_block(const MTS_CONST_AS void *kernarg_, uint32_t *hot_lds, uint32_t wg_base_, uint32_t pid, Counters *cnt) {
    const uint64_t ka = (uint64_t) (uintptr_t) kernarg_;
    uint32_t ka_lo = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) ka), ka_hi = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) (ka >> 32));
    asm volatile("" : "+s"(ka_lo), "+s"(ka_hi));             // opaque: scene loads stay inside this block
    const MTS_CONST_AS void *kernarg = (const MTS_CONST_AS void *) (uintptr_t) ((uint64_t) ka_lo | ((uint64_t) ka_hi << 32));
    const uint32_t wg_base = (uint32_t) __builtin_amdgcn_readfirstlane((int) wg_base_);
    const WgArgs a = cload_k<WgArgs>(kernarg);
    VolpathMisMachine<COUNT, SPEC> vm(a.sc, *cnt);
    PathEnvT<ColdStoreHbm> e; wg_env<WG>(a, wg_base, pid, e);
    MisHotStore<WG, SPEC> hs; hs.base = hot_lds + pid;
    typedef MisClassFields<C> CF;
    MisPathState<SPEC> p;
    hs.template load_m<CF::load>(p);
    int cls;
#pragma nounroll
    for (int rounds = 0;; ++rounds) {                           // tracking steps repeat in place while half the wave stays (volpath_flat.h, wg_block)
        vm.template run<CF::defer>(p, e, C);
        vm.template top<CF::defer>(p, e);
        cls = vm.classify(p);
        if ((MTS_CHAIN & 1) && C == B_WSURF) {                 // a walk that left the scene's box ends in the same visit (volpath_flat.h, MTS_CHAIN)
            if (cls != C || rounds >= 1) break;
            continue;
        }
        if (!(C == B_MED || C == B_MEDW) || cls != C || rounds >= 16) break;
        if (__popcll(__ballot(true)) < MTS_REPEAT_MIN) break;
    }
    hs.template store_m<CF::store>(p, cls);
    if (CF::defer && p.st == S_ENDNEE) {                        // the end of a walk, on the full state
        MisPathState<SPEC> q;
        hs.template load_m<K_ALL>(q);
        vm.finish(q, e);
        vm.top(q, e);
        cls = vm.classify(q);
        hs.template store_m<K_ALL>(q, cls);
    }
    return cls;
}

// The asynchronous-regrouping driver of volpath_flat.h (volpath_workgroup_async) for the MIS machine: same rings, same claim, same
// hand-over and stop protocol; it is a second copy rather than a shared template so that the tuned volpath kernel keeps its code.
template <bool COUNT, bool SPEC, int WG /* paths */, int NT /* threads */>
DEV void volpathmis_workgroup_async(const MTS_CONST_AS void *kernarg, Counters &cnt) {
    constexpr int NQ = B_DONE;
    typedef MisHotStore<WG, SPEC> Hot;
    static_assert((WG & (WG - 1)) == 0 && NT % 64 == 0 && WG % 64 == 0 && NT <= 2 * WG, "whole waves, power-of-two rings");
    __shared__ uint32_t hot_lds[Hot::M_COUNT * WG];
    __shared__ uint16_t q_ids[NQ][WG];
    __shared__ __attribute__((aligned(8))) uint32_t q_ctl[2 * B_COUNT + 2];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wg_base = blockIdx.x * WG;
#pragma unroll 1
    for (int c = 0; c < NQ; ++c) {
#pragma unroll 1
        for (uint32_t i = tid; i < (uint32_t) WG; i += NT) q_ids[c][i] = 0xFFFFu;
    }
    if (tid < 2u * B_COUNT + 2u) q_ctl[tid] = 0;
    pm_tables_to_lds(tid);
    __syncthreads();
#pragma unroll 1
    for (uint32_t pid0 = tid; pid0 < (uint32_t) WG; pid0 += NT) {   // ---- initialise the paths (integrator.cpp:198) and queue them
        const WgArgs a = cload_k<WgArgs>(kernarg);
        VolpathMisMachine<COUNT, SPEC> vm(a.sc, cnt);
        PathEnvT<ColdStoreHbm> e; MisPathState<SPEC> p;
        Hot hs; hs.base = hot_lds + pid0;
        const bool ok = wg_env<WG>(a, wg_base, pid0, e);
        p.rng.state = 0; p.rng.inc = 0;
        p.ray = make_ray(f3s(0.f), f3(0.f, 0.f, 1.f), 0.f, 0.f); p.si.t = pm_inf(); p.si.p = f3s(0.f); p.si.uv.x = p.si.uv.y = 0.f; p.si.shape = -1; p.si.prim = 0;
        p.medium = -1; p.res = spec_s(0.f); p.lsp = f3s(0.f); p.eta = 1.f; p.depth = 0; p.channel = 0; p.mode = M_MAIN; p.flags = 0; p.wa = p.wb = 0.f;
#if MTS_SPEC_N != 3
        p.wl = spec_s(0.f);
#endif
        p.pf = p.pn = p.wn = p.wu = mw_full<SPEC>(1.f);
        p.st = S_DONE;
        if (ok) {
            const uint32_t ppb = a.block_size * a.block_size;
            p.rng.seed(a.sc.sensor.seed + (uint64_t) e.blk.id * ppb + e.index, PCG32_DEFAULT_STREAM);
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
    uint32_t poll_ticks = (tid >> 6) * 2048u, idle_naps = 0, idle_t0 = 0;
    uint32_t idle_limit = MTS_IDLE_TICKS; bool drop_one = false;
    if (COUNT) {                                              // error-path test hook (volpath_flat.h, MTS_INJECT_SLOT)
        const uint32_t inj = (uint32_t) __builtin_amdgcn_readfirstlane((int) (uint32_t) cload_k<WgArgs>(kernarg).counters[MTS_INJECT_SLOT]);
        if (inj != 0u) { idle_limit = inj; drop_one = blockIdx.x == 0u && tid < 64u; }
    }
    const bool record_cost = __builtin_amdgcn_readfirstlane((int) (uint32_t) cload_k<WgArgs>(kernarg).counters[MTS_COST_FLAG]) != 0;   // calibration launch (volpath_flat.h)
    const long long cost_t0 = record_cost ? clock64() : 0ll;
#pragma unroll 1
    for (;;) {
      uint32_t n = 0, h = 0, spec_slot = 0xFFFFu; int sel = 0; bool finished = false;
#pragma unroll 1
      for (;;) {                                                // the claim: snapshot, vote, compare-and-swap (see volpath_flat.h)
        uint32_t hd = 0, avail = 0;
        if (lane < (uint32_t) B_COUNT) {
            hd = __atomic_load_n(&q_ctl[2 * lane], __ATOMIC_RELAXED);
            const uint32_t tl = __atomic_load_n(&q_ctl[2 * lane + 1], __ATOMIC_RELAXED);
            avail = tl - hd;
            if (avail > (uint32_t) WG) avail = 0;
        }
        uint32_t key = lane < (uint32_t) NQ ? ((avail << 4) | (15u - lane)) : 0u;
        key = max(key, (uint32_t) __builtin_amdgcn_update_dpp(0, (int) key, 0x111 /* row_shr:1 */, 0xf, 0xf, true));
        key = max(key, (uint32_t) __builtin_amdgcn_update_dpp(0, (int) key, 0x112 /* row_shr:2 */, 0xf, 0xf, true));
        key = max(key, (uint32_t) __builtin_amdgcn_update_dpp(0, (int) key, 0x114 /* row_shr:4 */, 0xf, 0xf, true));
        const uint32_t top_key = (uint32_t) __builtin_amdgcn_readlane((int) key, 7);
        const uint32_t best = top_key >> 4; sel = 15 - (int) (top_key & 15u);
        if (best == 0) {
            if ((uint32_t) __builtin_amdgcn_readlane((int) avail, B_DONE) == (uint32_t) WG || __atomic_load_n(&q_ctl[2 * B_COUNT], __ATOMIC_RELAXED) != STOP_NONE) { finished = true; break; }
            if ((poll_ticks += 1u) >= 32768u) {
                poll_ticks = 0;
                if (lane == 0 && __hip_atomic_load(cload_k<WgArgs>(kernarg).stop_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u)
                    (void) wga_raise_stop<WG>(q_ctl, STOP_CANCEL);
            }
            if (wga_idle_expired(idle_naps, idle_t0, idle_limit)) wga_stall<WG>(3u, B_DONE, 0u, q_ctl, cload_k<WgArgs>(kernarg).counters);   // a lost path: report, do not hang
            __builtin_amdgcn_s_sleep(2);
            continue;
        }
        idle_naps = 0;
        n = best < 64u ? best : 64u;
        h = (uint32_t) __builtin_amdgcn_readlane((int) hd, sel);
        uint32_t won = 0;
        if (lane < n) spec_slot = __atomic_load_n(&q_ids[sel][(h + lane) & (uint32_t) (WG - 1)], __ATOMIC_RELAXED);     // tagged slots: read with the claim (volpath_flat.h)
        if (lane == 0) won = atomicCAS(&q_ctl[2 * sel], h, h + n) == h ? 1u : 0u;
        if (__builtin_amdgcn_readfirstlane((int) won)) break;
      }
      if (finished) break;
        uint32_t pid = 0xFFFFu;
        bool mine = lane < n;
        {
            uint16_t *slot = &q_ids[sel][(h + lane) & (uint32_t) (WG - 1)];
            const bool ready = mine && RingSlot<WG>::matches(spec_slot, h + lane);
            if (ready) pid = RingSlot<WG>::id(spec_slot);
            if (__builtin_amdgcn_ballot_w64(mine && !ready) != 0ull) {
                const uint32_t got = wga_tag_wait<WG>(mine && !ready, slot, q_ctl, cload_k<WgArgs>(kernarg).counters, sel, h + lane);
                if (mine && !ready) pid = got;
                mine = mine && pid != 0xFFFFu;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        int cls = B_DONE;
        if (mine) {
            switch (sel) {                                      // wave-uniform
                case B_INT: cls = mis_block<COUNT, SPEC, WG, B_INT>(kernarg, hot_lds, wg_base, pid, &cnt); break;
                case B_MED: cls = mis_block<COUNT, SPEC, WG, B_MED>(kernarg, hot_lds, wg_base, pid, &cnt); break;
                case B_MEDW: cls = mis_block<COUNT, SPEC, WG, B_MEDW>(kernarg, hot_lds, wg_base, pid, &cnt); break;
                case B_SCATTER: cls = mis_block<COUNT, SPEC, WG, B_SCATTER>(kernarg, hot_lds, wg_base, pid, &cnt); break;
                case B_WSURF: cls = mis_block<COUNT, SPEC, WG, B_WSURF>(kernarg, hot_lds, wg_base, pid, &cnt); break;
                case B_SURF: cls = mis_block<COUNT, SPEC, WG, B_SURF>(kernarg, hot_lds, wg_base, pid, &cnt); break;
                case B_PHASE: cls = mis_block<COUNT, SPEC, WG, B_PHASE>(kernarg, hot_lds, wg_base, pid, &cnt); break;
                default: cls = mis_block<COUNT, SPEC, WG, B_NEW>(kernarg, hot_lds, wg_base, pid, &cnt); break;
            }
        }
        if (sel == B_NEW && (poll_ticks += 256u) >= 32768u) {
            poll_ticks = 0;
            if (lane == 0 && __hip_atomic_load(cload_k<WgArgs>(kernarg).stop_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u)
                (void) wga_raise_stop<WG>(q_ctl, STOP_CANCEL);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (COUNT && drop_one && cls != B_DONE) { mine = mine && lane != 0u; drop_one = false; }      // the injected lost hand-over (test hook)
        if (record_cost && mine && cls == B_DONE)
            atomicAdd(cload_k<WgArgs>(kernarg).counters + MTS_COST_BASE + (wg_base + pid) / MTS_TILE_PIXELS, (unsigned long long) (clock64() - cost_t0));
        wga_push<WG>(cls, pid, mine, q_ids, q_ctl);
    }
    __syncthreads();                                          // stopped: the unfinished pixels' samples go to the film (volpath_flat.h)
    if (__atomic_load_n(&q_ctl[2 * B_COUNT], __ATOMIC_RELAXED) != STOP_NONE) wg_flush_unfinished<WG, NT>(kernarg, hot_lds, Hot::M_PACKED, wg_base);
}

} // inline namespace
} // namespace mtsamd
