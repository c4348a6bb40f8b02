// kernels.hip -- HIP kernels of the path / volpath render loop for gfx950 (MI355X) and their launchers.
//
// Execution model: one lane per pixel, one wave = an 8x8 Morton tile, one 256-thread workgroup = a
// 16x16 tile of a 32x32 spiral block.  A lane seeds the reference's per-pixel PCG32 stream
// (librender/integrator.cpp:198) and runs all samples of its pixel; path state never leaves
// registers, the scene is read with scalar loads, the only vector memory traffic is the volume
// gathers and one film update per pixel.  Citations are relative to /root/reference.
// Compiled twice into libmtsamd.so: as it is (MTS_SPEC_N = 3: the rgb / mono variants, all kernels) and through kernels_spectral.hip
// (MTS_SPEC_N = 4: the spectral variant: `volpath` on the regrouping machine with 512-path workgroups, `path` per lane; launchers carry
// the suffix _spectral).
#include <hip/hip_runtime.h>
#if !defined(EXP_PM_TABLES_CONST)     // measurement only: the tables of pm_log / pm_exp read from the constant address space
#define PM_TABLES_IN_LDS 1            // pmath.h: LDS copies of the two hot lookup tables; every kernel below fills them first
#endif
#include "integrator_dev.h"
#include "volpath_flat.h"
#include "volpathmis_flat.h"
#if defined(MTS_LEAN)               // kernels_lean_*.hip: the regrouping kernels once more, for scenes that keep the promises of MTS_TRAITS
#define MTS_LAUNCHER_CAT2(a, b) a##b
#define MTS_LAUNCHER_CAT(a, b) MTS_LAUNCHER_CAT2(a, b)
#define MTS_LAUNCHER(name) MTS_LAUNCHER_CAT(name, MTS_LEAN)
#elif MTS_SPEC_N == 3
#define MTS_LAUNCHER(name) name
#else
#define MTS_LAUNCHER(name) name##_spectral
#endif
#include "launch.h"

namespace mtsamd {
inline namespace MTS_VARIANT_NS {

#if !defined(MTS_LEAN) || defined(MTS_LEAN_PATH)
// librender/integrator.cpp:233-288 + librender/imageblock.cpp:79-172, fused: the sample is splatted
// straight into the film.  With the default box filter a sample lands in its own pixel and is summed
// in registers in sample order (bit-identical to the reference's block accumulation); the rare
// sample that falls on the left/top pixel edge (u == 0) goes to the neighbour through an atomic.
template <bool COUNT, int INTEG>
__device__ __forceinline__ void render_sample(const DScene &sc, Pcg32 &rng, const DBlock &blk, uint32_t lx, uint32_t ly,
                                              float *__restrict__ film, float acc[5], Counters &cnt) {
    const DSensor &se = sc.sensor;
    float px = (float) (lx + (uint32_t) blk.ox), py = (float) (ly + (uint32_t) blk.oy);
    F2 u = rng.next_2d();
    F2 position_sample; position_sample.x = px + u.x; position_sample.y = py + u.y;
    F2 aperture_sample; aperture_sample.x = .5f; aperture_sample.y = .5f;
    if (se.needs_aperture_sample) aperture_sample = rng.next_2d();
    if (se.shutter_open_time > 0.f) (void) rng.next_1d();        // time sample (integrator.cpp:248-250)
#if MTS_SPEC_N == 3
    (void) rng.next_1d();                                       // wavelength sample (integrator.cpp:252), unused in rgb
#else
    SpecCtx cx = make_ctx(sc);
    Spec wav_weight;
    {
        const float wavelength_sample = rng.next_1d();          // integrator.cpp:252 -> Sensor::sample_ray: perspective.cpp:169-182, distant.cpp:311-313
        if (sc.srf >= 0) cx.wl = sample_wavelengths_srf(sc, wavelength_sample, wav_weight);
        else { float w; cx.wl = sample_wavelengths(wavelength_sample, w); wav_weight = spec_s(w); }
    }
#endif
    F2 adjusted;
    adjusted.x = (position_sample.x - (float) se.crop_x) / (float) se.crop_w;
    adjusted.y = (position_sample.y - (float) se.crop_y) / (float) se.crop_h;
    F3 ray_weight;
    DRay ray = sensor_sample_ray(sc, adjusted, aperture_sample, ray_weight);
    bool valid;
#if MTS_SPEC_N == 3
    F3 L = integrator_sample<COUNT, INTEG>(sc, rng, ray, se.medium, valid, cnt);
    L = ray_weight * L;
    splat_sample_t<false>(sc, blk, lx, ly, position_sample, L, valid, as_global(film), acc);
#else
    Spec L = integrator_sample<COUNT, INTEG>(sc, rng, ray, se.medium, valid, cnt, cx);
    float aov[2 * 64]; const int na = 2 * sc.bin_count;        // nbins / bins: the wrapped integrator's own result, before the ray weight
    for (int i = 0; i < sc.bin_count; ++i) bin_aovs(sc, L, cx.wl, i, aov[2 * i], aov[2 * i + 1]);
    L = (wav_weight * ray_weight.x) * L;                        // ray_weight = wav_weight (x the sensor's grey weight), integrator.cpp:265
    float xyz[3];
    spectrum_to_xyz(sc.cie, L, cx.wl, xyz);                     // integrator.cpp:266-269
    const float v[5] = { xyz[0], xyz[1], xyz[2], valid ? 1.f : 0.f, 1.f };
    splat_values_t<false>(sc, blk, lx, ly, position_sample, v, as_global(film), acc, aov, na);
#endif
}

// `path` (integrators/path.cpp:100-211) for one pixel as ONE flat loop over path segments with regeneration: a lane whose path has ended
// splats its sample and starts the pixel's next one at once, instead of idling until the longest path of the wave's 64 pixels has ended
// (path lengths under Russian roulette are roughly geometric: the longest of 64 is several times the mean).  One ray_intersect site
// serves camera rays and BSDF-sampled rays alike.  The lane draws exactly the numbers path_sample (integrator_dev.h) draws, in the
// same order, and sums its samples in the same order: bit-identical to the nested formulation and to the CPU restatement.
// Written over the variant's spectrum type: in the spectral build a sample also draws its four wavelengths (from the sensor's response
// function when there is one), carries the bins' AOV values and reaches the film through spectrum_to_xyz, as render_sample above.
template <bool COUNT>
__device__ __forceinline__ void path_pixel_flat(const DScene &sc, Pcg32 &rng, const DBlock &blk, uint32_t lx, uint32_t ly, uint32_t sample_count,
                                                float *__restrict__ film, float acc[5], Counters &cnt, const uint32_t *stop_flag) {
    const DSensor &se = sc.sensor;
    const int max_depth = sc.integrator.max_depth, rr_depth = sc.integrator.rr_depth;
    const float px = (float) (lx + (uint32_t) blk.ox), py = (float) (ly + (uint32_t) blk.oy);
    F2 position_sample; float ray_weight = 1.f;
    DRay ray;
    Spec throughput = spec_s(1.f), result = spec_s(0.f);
    F3 ref_p = f3s(0.f);
    float eta = 1.f, emission_weight = 1.f, bs_pdf = 0.f;
    uint32_t bs_type = 0;
    bool valid_ray = false;
    int depth = 0;                                              // 0: the camera ray of a fresh sample has not been traced yet
    uint32_t j = 0;
#if MTS_SPEC_N != 3
    SpecCtx cx = make_ctx(sc);
    Spec wav_weight = spec_s(0.f);
#endif
    auto begin_sample = [&]() {                                 // integrator.cpp:242-264, path.cpp:106-119
        if (se.wavefront) seed_wavefront_sample(rng, se, blk, lx, ly, j);     // gpu_* streams: one per (pixel, sample)
        F2 u = rng.next_2d();
        position_sample.x = px + u.x; position_sample.y = py + u.y;
        F2 aperture_sample; aperture_sample.x = .5f; aperture_sample.y = .5f;
        if (se.needs_aperture_sample) aperture_sample = rng.next_2d();
        if (se.shutter_open_time > 0.f) (void) rng.next_1d();
#if MTS_SPEC_N == 3
        (void) rng.next_1d();                                   // wavelength sample, unused in rgb
#else
        {
            const float wavelength_sample = rng.next_1d();      // integrator.cpp:252 -> Sensor::sample_ray: perspective.cpp:169-182, distant.cpp:311-313
            if (sc.srf >= 0) cx.wl = sample_wavelengths_srf(sc, wavelength_sample, wav_weight);
            else { float w; cx.wl = sample_wavelengths(wavelength_sample, w); wav_weight = spec_s(w); }
        }
#endif
        F2 adjusted;
        adjusted.x = (position_sample.x - (float) se.crop_x) / (float) se.crop_w;
        adjusted.y = (position_sample.y - (float) se.crop_y) / (float) se.crop_h;
        F3 rw;
        ray = sensor_sample_ray(sc, adjusted, aperture_sample, rw);
        ray_weight = rw.x;
        throughput = spec_s(1.f); result = spec_s(0.f); eta = 1.f; emission_weight = 1.f; depth = 0;
    };
    begin_sample();
    for (uint32_t it = 0;; ++it) {
        if ((it & 1023u) == 1023u && stop_requested(stop_flag)) break;      // should_stop(), integrator.h:143-146
        const Hit si = ray_intersect(sc, ray);
        const int emitter = hit_emitter(sc, si);
        if (depth == 0) valid_ray = hit_valid(si);               // path.cpp:113-115
        else if (emitter >= 0) {                                 // :193-205: MIS weight of the emitter the BSDF sample found
            Surf sb; sb.wi = -ray.d; sb.sh.n = f3s(0.f);
            if (hit_valid(si)) complete_surface(sc, si, ray.d, sb);
            DirSample ds;                                        // render/records.h:168-174
            ds.p = si.p; ds.n = sb.sh.n; ds.d = si.p - ref_p; ds.dist = norm(ds.d); ds.d = ds.d / ds.dist;
            if (!hit_valid(si)) ds.d = -sb.wi;
            ds.emitter = emitter; ds.pdf = 0.f; ds.delta = false;
            const float emitter_pdf = !(bs_type & F_Delta) ? pdf_emitter_direction(sc, ref_p, ds) : 0.f;
            emission_weight = mis_weight(bs_pdf, emitter_pdf);
        }
        depth += 1;
        // ---- one iteration of the loop of path.cpp:121-207
        if (COUNT) cnt.n_iter++;
        Surf sf; sf.wi = -ray.d;
        if (hit_valid(si)) complete_surface(sc, si, ray.d, sf);
        if (emitter >= 0) result = result + emission_weight * throughput * emitter_eval(sc, emitter, sf.wi.z MTS_CX);
        bool active = hit_valid(si);
        if (depth > rr_depth) {
            float q = pm_min(hmax(throughput) * (eta * eta), .95f);
            active = active && rng.next_1d() < q;
            throughput = throughput * pm_rcp(q);
        }
        bool ended = (uint32_t) depth >= (uint32_t) max_depth || !active;
        if (!ended) {
            const int bsdf_id = sc.shapes[si.shape].bsdf;
            const DBsdf &bsdf = sc.bsdfs[bsdf_id];
            bool active_e = (bsdf.flags & F_Smooth) != 0;
            if (active_e) {
                Spec emitter_val;
                DirSample ds = sample_emitter_direction(sc, si.p, rng.next_2d(), true, emitter_val MTS_CX);
                active_e = active_e && ds.pdf != 0.f;
                F3 wo = to_local(sf.sh, ds.d);
                Spec bsdf_val = bsdf_eval(bsdf, sf.wi, wo MTS_CXI(bsdf_id));
                float bpdf = bsdf_pdf(bsdf, sf.wi, wo MTS_CXI(bsdf_id));
                float mis = ds.delta ? 1.f : mis_weight(ds.pdf, bpdf);
                if (active_e) result = result + mis * throughput * bsdf_val * emitter_val;
            }
            float s1 = rng.next_1d(); F2 s2 = rng.next_2d();
            BSDFSample bs;
            Spec bsdf_val = bsdf_sample(bsdf, sf.wi, s1, s2, bs MTS_CXI(bsdf_id));
            throughput = throughput * bsdf_val;
            ended = !any_nonzero(throughput);
            if (!ended) {
                eta *= bs.eta;
                ref_p = si.p; bs_pdf = bs.pdf; bs_type = bs.sampled_type;
                ray = spawn_ray(si.p, to_world(sf.sh, bs.wo));
            }
        }
        if (ended) {                                             // integrator.cpp:265-288: splat, next sample of this pixel
#if MTS_SPEC_N == 3
            splat_sample_t<false>(sc, blk, lx, ly, position_sample, f3s(ray_weight) * result, valid_ray, as_global(film), acc);
#else
            float aov[2 * 64]; const int na = 2 * sc.bin_count;  // nbins / bins: the wrapped integrator's own result, before the ray weight
            for (int i = 0; i < sc.bin_count; ++i) bin_aovs(sc, result, cx.wl, i, aov[2 * i], aov[2 * i + 1]);
            const Spec L = (wav_weight * ray_weight) * result;   // integrator.cpp:265
            float xyz[3];
            spectrum_to_xyz(sc.cie, L, cx.wl, xyz);              // integrator.cpp:266-269
            const float v[5] = { xyz[0], xyz[1], xyz[2], valid_ray ? 1.f : 0.f, 1.f };
            splat_values_t<false>(sc, blk, lx, ly, position_sample, v, as_global(film), acc, aov, na);
#endif
            if (++j == sample_count) break;
            begin_sample();
        }
    }
}

// librender/integrator.cpp:181-209 (scalar branch) for every block of this launch at once.
// FLAT = true: volpath as the flat state machine of volpath_flat.h (the production kernel of the metric);
// FLAT = false: the nested formulation of integrator_dev.h (path and volpathmis; volpath cross-check, MTSAMD_KERNEL=nested),
// one instantiation per integrator (INTEG = NI_*).
#ifndef MTS_NESTED_WAVES
#define MTS_NESTED_WAVES 1
#endif
#if !defined(MTS_PATH_WAVES) && MTS_SPEC_N != 3
#define MTS_PATH_WAVES 3      // four-wide spectra: the register budget of four waves per SIMD is not met
#endif
#ifndef MTS_PATH_WAVES
#define MTS_PATH_WAVES 4      // measured on the cornell box: 1 -> 1631, 3 -> 1717, 4 -> 2355, 5 -> 1870, 6 -> 1241 Msamples/s
#endif
template <bool COUNT, bool FLAT, int INTEG>
__global__ void __launch_bounds__(256, (FLAT && INTEG != NI_PATH) ? 1 : (INTEG == NI_PATH ? MTS_PATH_WAVES : MTS_NESTED_WAVES)) render_kernel(DScene sc, const DBlock *__restrict__ blocks, uint32_t n_blocks, uint32_t block_size,
                                                     uint32_t sample_count, float *__restrict__ film_base, unsigned long long *__restrict__ counters,
                                                     const uint32_t *__restrict__ stop_flag) {
    // LDS-staged BVH top: the breadth-first top levels of the host-built BVH (dscene.h), shared by the workgroup's traversals
    __shared__ float bvh_top[MTS_BVH_LDS_NODES * 8];
    {
#if defined(EXP_NO_BVH_LDS)
        const int staged = 0;                                 // measurement only
#else
        const int staged = min(sc.bvh_node_count, MTS_BVH_LDS_NODES);
#endif
        for (int k = (int) threadIdx.x; k < staged * 8; k += (int) blockDim.x) bvh_top[k] = sc.bvh_nodes[k];
        pm_tables_to_lds(threadIdx.x);
        __syncthreads();
        sc.bvh_lds = bvh_top; sc.bvh_lds_count = staged;
    }
    const uint32_t ppb = block_size * block_size;
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t b = gid / ppb, i = gid - b * ppb;
    if (b >= n_blocks) return;
    const DBlock blk = blocks[b];
    float *__restrict__ film = film_base + (((size_t) blk.film_off_hi << 32) | blk.film_off_lo);      // the film slot of the entry's pass (mts_render)
    const uint32_t lx = compact_bits(i), ly = compact_bits(i >> 1);                           // morton_decode, integrator.cpp:200
    if (lx >= (uint32_t) blk.sx || ly >= (uint32_t) blk.sy) return;
    Pcg32 rng;
    rng.seed(sc.sensor.seed + (uint64_t) blk.id * ppb + i, PCG32_DEFAULT_STREAM);             // sampler.cpp:83-96, integrator.cpp:198
    Counters cnt = {};
#if MTS_SPEC_N == 3
    if (FLAT && INTEG != NI_PATH) {
        __shared__ float cold_lds[C_COUNT * 256];
        ColdStore cold; cold.base = cold_lds + threadIdx.x; cold.stride = 256;
        volpath_pixel_flat<COUNT>(sc, rng, blk, lx, ly, sample_count, film, cold, cnt, stop_flag);
    } else
#endif
    {
        float acc[5] = { 0.f, 0.f, 0.f, 0.f, 0.f };
        if (INTEG == NI_PATH && FLAT) path_pixel_flat<COUNT>(sc, rng, blk, lx, ly, sample_count, film, acc, cnt, stop_flag);
        else
        for (uint32_t j = 0; j < sample_count; ++j) {
            // should_stop(), integrator.h:143-146: the reference looks at its flag once per sample; here one lane of the wave reads the
            // host-visible word every 64 samples
            if ((j & 63u) == 63u && stop_requested(stop_flag)) break;
            if (sc.sensor.wavefront) seed_wavefront_sample(rng, sc.sensor, blk, lx, ly, j);       // gpu_* streams: one per (pixel, sample)
            render_sample<COUNT, INTEG>(sc, rng, blk, lx, ly, film, acc, cnt);
        }
#if MTS_SPEC_N == 3
        const size_t film_channels = 5;
#else
        const size_t film_channels = (size_t) sc.film_channels;  // X, Y, Z, A, W (+ the bins' AOV channels)
#endif
        float *dst = film + film_channels * ((size_t) (blk.oy + (int) ly - sc.sensor.crop_y) * sc.sensor.crop_w + (blk.ox + (int) lx - sc.sensor.crop_x));
        for (int k = 0; k < 5; ++k) atomicAdd(dst + k, acc[k]);
    }
    if (COUNT) {
        atomicAdd(counters + 0, (unsigned long long) cnt.n_iter);
        atomicAdd(counters + 1, (unsigned long long) cnt.n_lookup);
        atomicAdd(counters + 2, (unsigned long long) cnt.n_nee_step);
    }
}

#endif // !MTS_LEAN || MTS_LEAN_PATH

// Asynchronous-regrouping variant of the volpath render kernel (volpath_flat.h, driver 2).  The parameter list must stay in
// sync with WgArgs: the block functions re-read it from the kernarg segment with scalar loads.  WG paths are served by NT threads;
// WPE = waves per SIMD the register budget is sized for (512 / WPE VGPRs).
template <bool COUNT, int WG, int NT, int WPE, bool WF = false>
__global__ void __launch_bounds__(NT, WPE) render_kernel_wga(DScene sc, const DBlock *blocks, uint32_t n_blocks, uint32_t block_size,
                                                           uint32_t sample_count, float *film, float *cold_g, uint32_t cold_stride,
                                                           unsigned long long *counters, const uint32_t *stop_flag,
                                                           const uint32_t *tiles, uint32_t n_tiles) {
    Counters cnt = {};
    volpath_workgroup_async<COUNT, WG, NT, WF>((const MTS_CONST_AS void *) __builtin_amdgcn_kernarg_segment_ptr(), cnt);
    if (COUNT) {
        atomicAdd(counters + 0, (unsigned long long) cnt.n_iter);
        atomicAdd(counters + 1, (unsigned long long) cnt.n_lookup);
        atomicAdd(counters + 2, (unsigned long long) cnt.n_nee_step);
    }
}
static_assert(sizeof(WgArgs) % 4 == 0, "WgArgs mirrors the kernel parameters");

#if !defined(MTS_LEAN)
// The same machine on the lane-affine driver (volpath_flat.h, driver 3): conflict-free LDS state, mask claims instead of rings.
template <bool COUNT, int WG, int NT, int WPE>
__global__ void __launch_bounds__(NT, WPE) render_kernel_wgl(DScene sc, const DBlock *blocks, uint32_t n_blocks, uint32_t block_size,
                                                           uint32_t sample_count, float *film, float *cold_g, uint32_t cold_stride,
                                                           unsigned long long *counters, const uint32_t *stop_flag,
                                                           const uint32_t *tiles, uint32_t n_tiles) {
    Counters cnt = {};
    workgroup_lanes<COUNT, WG, NT, VolpathLanes<COUNT, WG>>((const MTS_CONST_AS void *) __builtin_amdgcn_kernarg_segment_ptr(), cnt);
    if (COUNT) {
        atomicAdd(counters + 0, (unsigned long long) cnt.n_iter);
        atomicAdd(counters + 1, (unsigned long long) cnt.n_lookup);
        atomicAdd(counters + 2, (unsigned long long) cnt.n_nee_step);
    }
}

#endif // !MTS_LEAN

// The same driver for volpathmis (volpathmis_flat.h): four weight matrices per path, 512 paths per workgroup, two waves per SIMD.
template <bool COUNT, bool SPEC, int WG, int NT>
__global__ void __launch_bounds__(NT, NT <= 256 ? 2 : 1) render_kernel_wga_mis(DScene sc, const DBlock *blocks, uint32_t n_blocks, uint32_t block_size,
                                                               uint32_t sample_count, float *film, float *cold_g, uint32_t cold_stride,
                                                               unsigned long long *counters, const uint32_t *stop_flag,
                                                               const uint32_t *tiles, uint32_t n_tiles) {
    Counters cnt = {};
    volpathmis_workgroup_async<COUNT, SPEC, WG, NT>((const MTS_CONST_AS void *) __builtin_amdgcn_kernarg_segment_ptr(), cnt);
    if (COUNT) {
        atomicAdd(counters + 0, (unsigned long long) cnt.n_iter);
        atomicAdd(counters + 1, (unsigned long long) cnt.n_lookup);
        atomicAdd(counters + 2, (unsigned long long) cnt.n_nee_step);
    }
}

#if MTS_SPEC_N == 3 && !defined(MTS_LEAN)
// SamplingIntegrator::sample for caller-supplied rays (librender/python/integrator_v.cpp:62-78)
__global__ void __launch_bounds__(256) sample_kernel(DScene sc, int32_t n, uint64_t seed_offset, const float *__restrict__ rays /* 6 SoA rows */,
                                                     float *__restrict__ out_rgb, uint8_t *__restrict__ out_valid) {
    pm_tables_to_lds(threadIdx.x);
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Pcg32 rng; rng.seed(sc.sensor.seed + seed_offset + (uint64_t) i, PCG32_DEFAULT_STREAM);
    DRay ray = make_ray(f3(rays[i], rays[n + i], rays[2 * n + i]), f3(rays[3 * n + i], rays[4 * n + i], rays[5 * n + i]), MTS_RAY_EPSILON, pm_inf());
    bool valid; Counters cnt;
    F3 L = integrator_sample<false>(sc, rng, ray, sc.sensor.medium, valid, cnt);
    out_rgb[3 * i] = L.x; out_rgb[3 * i + 1] = L.y; out_rgb[3 * i + 2] = L.z; out_valid[i] = valid ? 1 : 0;
}

// Scene::ray_intersect for caller-supplied rays (librender/scene.cpp:117-125)
__global__ void __launch_bounds__(256) intersect_kernel(DScene sc, int32_t n, const float *__restrict__ o, const float *__restrict__ d,
                                                        const float *__restrict__ mint, const float *__restrict__ maxt,
                                                        float *__restrict__ out_t, int32_t *__restrict__ out_shape, int32_t *__restrict__ out_prim,
                                                        float *__restrict__ out_p, float *__restrict__ out_n) {
    pm_tables_to_lds(threadIdx.x);
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    DRay ray = make_ray(f3(o + 3 * i), f3(d + 3 * i), mint[i], maxt[i]);
    Hit h = ray_intersect(sc, ray);
    F3 nn = f3s(0.f);
    int prim = -1;
    if (hit_valid(h)) {
        Surf sf; complete_surface(sc, h, ray.d, sf); nn = sf.n;
        prim = sc.shapes[h.shape].type == MTS_SHAPE_SPHERE ? 0 : h.prim;
    }
    out_t[i] = h.t; out_shape[i] = h.shape; out_prim[i] = prim;
    out_p[3 * i] = h.p.x; out_p[3 * i + 1] = h.p.y; out_p[3 * i + 2] = h.p.z;
    out_n[3 * i] = nn.x; out_n[3 * i + 1] = nn.y; out_n[3 * i + 2] = nn.z;
}

// sample_tea_32 / sample_tea_64 / sample_tea_float32 for n (v0, v1) pairs (core/random.h:75-140)
__global__ void __launch_bounds__(256) tea_kernel(int32_t n, const uint32_t *__restrict__ v0, const uint32_t *__restrict__ v1, int rounds,
                                                  uint32_t *__restrict__ out32, uint64_t *__restrict__ out64, float *__restrict__ outf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out32[i] = sample_tea_32(v0[i], v1[i], rounds); out64[i] = sample_tea_64(v0[i], v1[i], rounds); outf[i] = sample_tea_float32(v0[i], v1[i], rounds);
}
// PCG32Sampler::seed of the wavefront variants (librender/sampler.cpp:83-92): lane idx of a wavefront gets
// rng.seed(tea64(seed_value, idx), tea64(idx, seed_value)); the first `count` next_1d() of every lane are written lane-major.
__global__ void __launch_bounds__(256) wavefront_sampler_kernel(int32_t lanes, uint64_t seed_value, int32_t count, float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= lanes) return;
    Pcg32 rng;
    rng.seed(sample_tea_64_u64(seed_value, (uint64_t) i), sample_tea_64_u64((uint64_t) i, seed_value));        // UInt64 instantiation, dmath.h
    for (int k = 0; k < count; ++k) out[(size_t) i * count + k] = rng.next_1d();
}

#endif // MTS_SPEC_N == 3
} // inline namespace

// ---------------------------------------------------------------- launchers
#if MTS_SPEC_N == 3 && !defined(MTS_LEAN)
hipError_t launch_tea(int32_t n, const uint32_t *v0, const uint32_t *v1, int rounds, uint32_t *out32, uint64_t *out64, float *outf, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(tea_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, n, v0, v1, rounds, out32, out64, outf);
    return hipGetLastError();
}
hipError_t launch_wavefront_sampler(int32_t lanes, uint64_t seed_value, int32_t count, float *out, hipStream_t stream) {
    if (lanes <= 0 || count <= 0) return hipSuccess;
    hipLaunchKernelGGL(wavefront_sampler_kernel, dim3((lanes + 255) / 256), dim3(256), 0, stream, lanes, seed_value, count, out);
    return hipGetLastError();
}

// film[i] = ((slot 0 [i] + slot 1 [i]) + slot 2 [i]) + ... : the passes of a render added in pass order, as Film::put(block) adds the
// blocks of pass after pass (integrator.cpp:98-107 -> hdrfilm.cpp:205-217 -> imageblock.cpp:59-77)
__global__ void __launch_bounds__(256) film_sum_slots_kernel(float *__restrict__ film, const float *__restrict__ slots, size_t n, uint32_t count) {
    const size_t i = (size_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = slots[i];
    for (uint32_t k = 1; k < count; ++k) v += slots[(size_t) k * n + i];
    film[i] = v;
}
hipError_t launch_film_sum_slots(float *d_film, const float *d_slots, size_t film_floats, uint32_t count, hipStream_t stream) {
    if (film_floats == 0 || count == 0) return hipSuccess;
    hipLaunchKernelGGL(film_sum_slots_kernel, dim3((unsigned) ((film_floats + 255) / 256)), dim3(256), 0, stream, d_film, d_slots, film_floats, count);
    return hipGetLastError();
}

size_t render_workspace_floats(uint64_t threads, int variant) {
    if (variant < 256) return 0;
    if (variant >= 20000) variant -= 20000;
    if (variant >= 10000) variant -= 10000;
    const uint64_t padded = (threads + variant - 1) / variant * variant;
    return (size_t) padded * (variant >= 256 && variant <= 4096 ? MTS_COLD_RECORD : C_COUNT) + 32;      // workgroup drivers: one 128-byte record per path
}

#endif // MTS_SPEC_N == 3

#if defined(MTS_LEAN)
// the 1024-path `volpath` machine and the 512-path `volpathmis` machine, nothing else (mts_render sends everything else to launch_render)
hipError_t MTS_LAUNCHER(launch_render)(const DScene &sc, const DBlock *d_blocks, uint32_t n_blocks, uint32_t block_size, uint32_t sample_count,
                         float *d_film, unsigned long long *d_counters, bool count, int variant, int wg_threads, float *d_workspace,
                         const uint32_t *d_stop_flag, const uint32_t *d_tiles, uint32_t n_tiles, hipStream_t stream) {
    if (n_blocks == 0) return hipSuccess;
    const uint64_t threads = d_tiles != nullptr ? (uint64_t) n_tiles * MTS_TILE_PIXELS : (uint64_t) n_blocks * block_size * block_size;
    if (threads + 1024 >= ((uint64_t) 1 << 32)) return hipErrorInvalidValue;
#if defined(MTS_LEAN_PATH)    // kernels_lean_p.hip / _ps.hip: `path` as the flat loop with regeneration, nothing else
    if (variant == 1 && sc.integrator.type == MTS_INTEGRATOR_PATH) {
        const uint32_t grid = (uint32_t) ((threads + 255) / 256);
        if (count) hipLaunchKernelGGL((render_kernel<true, true, NI_PATH>), dim3(grid), dim3(256), 0, stream, sc, d_blocks, n_blocks, block_size, sample_count, d_film, d_counters, d_stop_flag);
        else hipLaunchKernelGGL((render_kernel<false, true, NI_PATH>), dim3(grid), dim3(256), 0, stream, sc, d_blocks, n_blocks, block_size, sample_count, d_film, d_counters, d_stop_flag);
        return hipGetLastError();
    }
    (void) d_workspace; (void) wg_threads;
    return hipErrorInvalidConfiguration;
#else
    if (sc.sensor.wavefront || wg_threads != 0) return hipErrorInvalidConfiguration;
#if MTS_SPEC_N != 3          // the spectral variant's machines: 256-path workgroups
    if (variant == 10256 && sc.integrator.type == MTS_INTEGRATOR_VOLPATH) {
        const uint32_t grid = (uint32_t) ((threads + 255) / 256), stride = grid * 256;
        // register budget of three waves per SIMD (three 256-path workgroups per CU): with the spectral grid lookups inline the allocator needs the bound
        if (count) hipLaunchKernelGGL((render_kernel_wga<true, 256, 256, 3>), dim3(grid), dim3(256), 0, stream, sc, d_blocks, n_blocks, block_size, sample_count, d_film, d_workspace, stride, d_counters, d_stop_flag, d_tiles, n_tiles);
        else hipLaunchKernelGGL((render_kernel_wga<false, 256, 256, 3>), dim3(grid), dim3(256), 0, stream, sc, d_blocks, n_blocks, block_size, sample_count, d_film, d_workspace, stride, d_counters, d_stop_flag, d_tiles, n_tiles);
        return hipGetLastError();
    }
    if (variant == 10256 && sc.integrator.type == MTS_INTEGRATOR_VOLPATHMIS && sc.integrator.use_spectral_mis) {
        const uint32_t grid = (uint32_t) ((threads + 255) / 256), stride = grid * 256;
        if (count) hipLaunchKernelGGL((render_kernel_wga_mis<true, true, 256, 256>), dim3(grid), dim3(256), 0, stream, sc, d_blocks, n_blocks, block_size, sample_count, d_film, d_workspace, stride, d_counters, d_stop_flag, d_tiles, n_tiles);
        else hipLaunchKernelGGL((render_kernel_wga_mis<false, true, 256, 256>), dim3(grid), dim3(256), 0, stream, sc, d_blocks, n_blocks, block_size, sample_count, d_film, d_workspace, stride, d_counters, d_stop_flag, d_tiles, n_tiles);
        return hipGetLastError();
    }
    return hipErrorInvalidConfiguration;
#else
    if (variant == 11024 && sc.integrator.type == MTS_INTEGRATOR_VOLPATH) {
        const uint32_t grid = (uint32_t) ((threads + 1023) / 1024), stride = grid * 1024;
        if (count) hipLaunchKernelGGL((render_kernel_wga<true, 1024, 1024, 4>), dim3(grid), dim3(1024), 0, stream, sc, d_blocks, n_blocks, block_size, sample_count, d_film, d_workspace, stride, d_counters, d_stop_flag, d_tiles, n_tiles);
        else hipLaunchKernelGGL((render_kernel_wga<false, 1024, 1024, 4>), dim3(grid), dim3(1024), 0, stream, sc, d_blocks, n_blocks, block_size, sample_count, d_film, d_workspace, stride, d_counters, d_stop_flag, d_tiles, n_tiles);
        return hipGetLastError();
    }
    if (variant == 10512 && sc.integrator.type == MTS_INTEGRATOR_VOLPATHMIS && sc.integrator.use_spectral_mis) {
        const uint32_t grid = (uint32_t) ((threads + 511) / 512), stride = grid * 512;
#if defined(MTS_LEAN_MIS_768)   // the 512 paths served by 768 threads: three waves per SIMD want <= 168 VGPRs, which this unit's kernel meets (C3M 375 -> 393)
        if (count) hipLaunchKernelGGL((render_kernel_wga_mis<true, true, 512, 768>), dim3(grid), dim3(768), 0, stream, sc, d_blocks, n_blocks, block_size, sample_count, d_film, d_workspace, stride, d_counters, d_stop_flag, d_tiles, n_tiles);
        else hipLaunchKernelGGL((render_kernel_wga_mis<false, true, 512, 768>), dim3(grid), dim3(768), 0, stream, sc, d_blocks, n_blocks, block_size, sample_count, d_film, d_workspace, stride, d_counters, d_stop_flag, d_tiles, n_tiles);
        return hipGetLastError();
#endif
        if (count) hipLaunchKernelGGL((render_kernel_wga_mis<true, true, 512, 512>), dim3(grid), dim3(512), 0, stream, sc, d_blocks, n_blocks, block_size, sample_count, d_film, d_workspace, stride, d_counters, d_stop_flag, d_tiles, n_tiles);
        else hipLaunchKernelGGL((render_kernel_wga_mis<false, true, 512, 512>), dim3(grid), dim3(512), 0, stream, sc, d_blocks, n_blocks, block_size, sample_count, d_film, d_workspace, stride, d_counters, d_stop_flag, d_tiles, n_tiles);
        return hipGetLastError();
    }
    return hipErrorInvalidConfiguration;
#endif
#endif // MTS_LEAN_PATH
}
#else
hipError_t MTS_LAUNCHER(launch_render)(const DScene &sc, const DBlock *d_blocks, uint32_t n_blocks, uint32_t block_size, uint32_t sample_count,
                         float *d_film, unsigned long long *d_counters, bool count, int variant, int wg_threads, float *d_workspace,
                         const uint32_t *d_stop_flag, const uint32_t *d_tiles, uint32_t n_tiles, hipStream_t stream) {
    if (n_blocks == 0) return hipSuccess;
    // cost-sorted tiles (regrouping kernels only): the launch covers n_tiles slots of 16 paths instead of the blocks' concatenated Morton orders
    const uint64_t threads = d_tiles != nullptr ? (uint64_t) n_tiles * MTS_TILE_PIXELS : (uint64_t) n_blocks * block_size * block_size;
    if (d_tiles != nullptr && variant < 10000) return hipErrorInvalidConfiguration;
    if (threads + 1024 >= ((uint64_t) 1 << 32)) return hipErrorInvalidValue;      // thread and path indices are 32 bit (mts_render launches in chunks)
#if MTS_SPEC_N == 3
    if (variant >= 20000 && sc.integrator.type == MTS_INTEGRATOR_VOLPATH) {        // lane-affine regrouping, variant = 20000 + paths per workgroup
        const uint32_t wg = (uint32_t) (variant - 20000);
        const uint32_t grid = (uint32_t) ((threads + wg - 1) / wg);
        const uint32_t stride = grid * wg;
        const int nt = wg_threads > 0 ? wg_threads : (int) wg;
#define LAUNCH_WGL(W, T, E) do { if (count) hipLaunchKernelGGL((render_kernel_wgl<true, W, T, E>), dim3(grid), dim3(T), 0, stream, sc, d_blocks, n_blocks, block_size, sample_count, d_film, d_workspace, stride, d_counters, d_stop_flag, d_tiles, n_tiles); \
                                 else hipLaunchKernelGGL((render_kernel_wgl<false, W, T, E>), dim3(grid), dim3(T), 0, stream, sc, d_blocks, n_blocks, block_size, sample_count, d_film, d_workspace, stride, d_counters, d_stop_flag, d_tiles, n_tiles); } while (0)
        if (wg == 1024 && nt == 1024) LAUNCH_WGL(1024, 1024, 4);
        else return hipErrorInvalidConfiguration;
#undef LAUNCH_WGL
        return hipGetLastError();
    }
    if (variant >= 10000 && sc.integrator.type == MTS_INTEGRATOR_VOLPATH) {        // asynchronous regrouping, variant = 10000 + paths per workgroup
        const uint32_t wg = (uint32_t) (variant - 10000);
        const uint32_t grid = (uint32_t) ((threads + wg - 1) / wg);
        const uint32_t stride = grid * wg;
        const int nt = wg_threads > 0 ? wg_threads : (int) wg;
#define LAUNCH_WGA(W, T, E) do { if (count) hipLaunchKernelGGL((render_kernel_wga<true, W, T, E>), dim3(grid), dim3(T), 0, stream, sc, d_blocks, n_blocks, block_size, sample_count, d_film, d_workspace, stride, d_counters, d_stop_flag, d_tiles, n_tiles); \
                                 else hipLaunchKernelGGL((render_kernel_wga<false, W, T, E>), dim3(grid), dim3(T), 0, stream, sc, d_blocks, n_blocks, block_size, sample_count, d_film, d_workspace, stride, d_counters, d_stop_flag, d_tiles, n_tiles); } while (0)
        if (sc.sensor.wavefront) {                              // gpu_* streams: the instantiation that recomputes the generator's increment (wg_block, WF)
            if (wg != 1024 || nt != 1024) return hipErrorInvalidConfiguration;
            if (count) hipLaunchKernelGGL((render_kernel_wga<true, 1024, 1024, 4, true>), dim3(grid), dim3(1024), 0, stream, sc, d_blocks, n_blocks, block_size, sample_count, d_film, d_workspace, stride, d_counters, d_stop_flag, d_tiles, n_tiles);
            else hipLaunchKernelGGL((render_kernel_wga<false, 1024, 1024, 4, true>), dim3(grid), dim3(1024), 0, stream, sc, d_blocks, n_blocks, block_size, sample_count, d_film, d_workspace, stride, d_counters, d_stop_flag, d_tiles, n_tiles);
        } else
        if (wg == 256 && nt == 256) LAUNCH_WGA(256, 256, 4);
        else if (wg == 512 && nt == 512) LAUNCH_WGA(512, 512, 4);
        else if (wg == 512 && nt == 256) LAUNCH_WGA(512, 256, 2);
        else if (wg == 1024 && nt == 1024) LAUNCH_WGA(1024, 1024, 4);
        else if (wg == 1024 && nt == 768) LAUNCH_WGA(1024, 768, 3);
        else if (wg == 1024 && nt == 512) LAUNCH_WGA(1024, 512, 2);
        else return hipErrorInvalidConfiguration;
#undef LAUNCH_WGA
        return hipGetLastError();
    }
    if (variant >= 10000 && sc.integrator.type == MTS_INTEGRATOR_VOLPATHMIS) {     // the same machinery, variant = 10000 + paths per workgroup (<= 512)
        const uint32_t wg = (uint32_t) (variant - 10000);
        const uint32_t grid = (uint32_t) ((threads + wg - 1) / wg);
        const uint32_t stride = grid * wg;
        const bool spec = sc.integrator.use_spectral_mis != 0;
#define LAUNCH_MIS(C, S, W) hipLaunchKernelGGL((render_kernel_wga_mis<C, S, W, W>), dim3(grid), dim3(W), 0, stream, sc, d_blocks, n_blocks, block_size, sample_count, d_film, d_workspace, stride, d_counters, d_stop_flag, d_tiles, n_tiles)
#define LAUNCH_MIS_W(W) do { if (count) { if (spec) LAUNCH_MIS(true, true, W); else LAUNCH_MIS(true, false, W); } \
                             else { if (spec) LAUNCH_MIS(false, true, W); else LAUNCH_MIS(false, false, W); } } while (0)
        if (wg == 512) LAUNCH_MIS_W(512);
        else if (wg == 256) LAUNCH_MIS_W(256);
        else return hipErrorInvalidConfiguration;
#undef LAUNCH_MIS_W
#undef LAUNCH_MIS
        return hipGetLastError();
    }
    const bool flat = variant != 0;
#else
    if (variant >= 10000 && sc.integrator.type == MTS_INTEGRATOR_VOLPATH) {        // four-wide state: 42 hot dwords per path, 512 paths fill the LDS
        const uint32_t wg = (uint32_t) (variant - 10000);
        const uint32_t grid = (uint32_t) ((threads + wg - 1) / wg);
        const uint32_t stride = grid * wg;
#define LAUNCH_WGA(W) do { if (count) hipLaunchKernelGGL((render_kernel_wga<true, W, W, 2>), dim3(grid), dim3(W), 0, stream, sc, d_blocks, n_blocks, block_size, sample_count, d_film, d_workspace, stride, d_counters, d_stop_flag, d_tiles, n_tiles); \
                           else hipLaunchKernelGGL((render_kernel_wga<false, W, W, 2>), dim3(grid), dim3(W), 0, stream, sc, d_blocks, n_blocks, block_size, sample_count, d_film, d_workspace, stride, d_counters, d_stop_flag, d_tiles, n_tiles); } while (0)
        if (wg == 512) LAUNCH_WGA(512);
        else if (wg == 256) LAUNCH_WGA(256);
        else return hipErrorInvalidConfiguration;
#undef LAUNCH_WGA
        return hipGetLastError();
    }
    if (variant >= 10000 && sc.integrator.type == MTS_INTEGRATOR_VOLPATHMIS) {     // 4 x 4 weight matrices: 101 hot dwords per path with spectral MIS (256 paths, one workgroup per CU), 53 without
        const uint32_t wg = (uint32_t) (variant - 10000);
        const uint32_t grid = (uint32_t) ((threads + wg - 1) / wg);
        const uint32_t stride = grid * wg;
        const bool spec = sc.integrator.use_spectral_mis != 0;
        if (wg != 256) return hipErrorInvalidConfiguration;
#define LAUNCH_MIS(C, S) hipLaunchKernelGGL((render_kernel_wga_mis<C, S, 256, 256>), dim3(grid), dim3(256), 0, stream, sc, d_blocks, n_blocks, block_size, sample_count, d_film, d_workspace, stride, d_counters, d_stop_flag, d_tiles, n_tiles)
        // 69 hot dwords per path with spectral MIS (round 4: the path's matrices are parked during walks): two 256-path workgroups per CU,
        // 8 waves at <= 256 VGPRs.  (Before: 101 dwords, ONE workgroup per CU -- 256 threads: 38.7 Msamples/s on C5SM, 512 threads: 42.3;
        // MTSAMD_WG_THREADS=512 still gives the latter launch.)
        if (spec && wg_threads == 512) {
            if (count) hipLaunchKernelGGL((render_kernel_wga_mis<true, true, 256, 512>), dim3(grid), dim3(512), 0, stream, sc, d_blocks, n_blocks, block_size, sample_count, d_film, d_workspace, stride, d_counters, d_stop_flag, d_tiles, n_tiles);
            else hipLaunchKernelGGL((render_kernel_wga_mis<false, true, 256, 512>), dim3(grid), dim3(512), 0, stream, sc, d_blocks, n_blocks, block_size, sample_count, d_film, d_workspace, stride, d_counters, d_stop_flag, d_tiles, n_tiles);
        } else
        if (count) { if (spec) LAUNCH_MIS(true, true); else LAUNCH_MIS(true, false); }
        else { if (spec) LAUNCH_MIS(false, true); else LAUNCH_MIS(false, false); }
#undef LAUNCH_MIS
        return hipGetLastError();
    }
    const bool flat = variant != 0; (void) d_workspace;                           // path: the per-lane kernels
#endif
    const uint32_t grid = (uint32_t) ((threads + 255) / 256);
    const bool use_flat = flat && sc.integrator.type == MTS_INTEGRATOR_VOLPATH;
#define LAUNCH(C, F, I) hipLaunchKernelGGL((render_kernel<C, F, I>), dim3(grid), dim3(256), 0, stream, sc, d_blocks, n_blocks, block_size, sample_count, d_film, d_counters, d_stop_flag)
#define LAUNCH_C(F, I) do { if (count) LAUNCH(true, F, I); else LAUNCH(false, F, I); } while (0)
#if MTS_SPEC_N == 3
    if (use_flat) LAUNCH_C(true, NI_VOLPATH);
    else
#endif
    if (sc.integrator.type == MTS_INTEGRATOR_PATH) {
        if (flat) LAUNCH_C(true, NI_PATH);                         // one flat loop over path segments with regeneration (path_pixel_flat)
        else LAUNCH_C(false, NI_PATH);
    }
    else if (sc.integrator.type == MTS_INTEGRATOR_VOLPATH) LAUNCH_C(false, NI_VOLPATH);
    else if (sc.integrator.use_spectral_mis) LAUNCH_C(false, NI_VOLPATHMIS);
    else LAUNCH_C(false, NI_VOLPATHMIS_NOSPEC);
#undef LAUNCH_C
#undef LAUNCH
    (void) use_flat;
    return hipGetLastError();
}
#endif // MTS_LEAN

#if MTS_SPEC_N == 3 && !defined(MTS_LEAN)

hipError_t launch_sample(const DScene &sc, int32_t n, uint64_t seed_offset, const float *d_rays, float *d_rgb, uint8_t *d_valid, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(sample_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, sc, n, seed_offset, d_rays, d_rgb, d_valid);
    return hipGetLastError();
}

hipError_t launch_intersect(const DScene &sc, int32_t n, const float *o, const float *d, const float *mint, const float *maxt,
                            float *t, int32_t *shape, int32_t *prim, float *p, float *nn, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(intersect_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, sc, n, o, d, mint, maxt, t, shape, prim, p, nn);
    return hipGetLastError();
}
#endif // MTS_SPEC_N == 3

#if MTS_SPEC_N != 3 && !defined(MTS_LEAN)
// SamplingIntegrator::sample in the spectral variant (librender/python/integrator_v.cpp:62-78): the caller's rays carry their wavelengths
__global__ void __launch_bounds__(256) sample_spectral_kernel(DScene sc, int32_t n, uint64_t seed_offset, const float *__restrict__ rays /* 6 SoA rows */,
                                                              const float *__restrict__ wavelengths /* 4 per ray */,
                                                              float *__restrict__ out_spec, uint8_t *__restrict__ out_valid) {
    pm_tables_to_lds(threadIdx.x);
    __syncthreads();
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Pcg32 rng; rng.seed(sc.sensor.seed + seed_offset + (uint64_t) i, PCG32_DEFAULT_STREAM);
    DRay ray = make_ray(f3(rays[i], rays[n + i], rays[2 * n + i]), f3(rays[3 * n + i], rays[4 * n + i], rays[5 * n + i]), MTS_RAY_EPSILON, pm_inf());
    SpecCtx cx = make_ctx(sc);
    cx.wl = spec4(wavelengths[4 * i], wavelengths[4 * i + 1], wavelengths[4 * i + 2], wavelengths[4 * i + 3]);
    bool valid; Counters cnt;
    const Spec L = integrator_sample<false>(sc, rng, ray, sc.sensor.medium, valid, cnt, cx);
    out_spec[4 * i] = L.x; out_spec[4 * i + 1] = L.y; out_spec[4 * i + 2] = L.z; out_spec[4 * i + 3] = L.w; out_valid[i] = valid ? 1 : 0;
}

hipError_t launch_sample_spectral(const DScene &sc, int32_t n, uint64_t seed_offset, const float *d_rays, const float *d_wavelengths, float *d_spec, uint8_t *d_valid,
                                  hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(sample_spectral_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, sc, n, seed_offset, d_rays, d_wavelengths, d_spec, d_valid);
    return hipGetLastError();
}
#endif

} // namespace mtsamd

#if defined(MTSAMD_BLOCKSTATS)
// diagnostic build only (python eradiate-kernel_amd/build.py with MTSAMD_EXTRA_FLAGS=-DMTSAMD_BLOCKSTATS); one copy per variant
#if MTS_SPEC_N == 3
extern "C" int mts_debug_blockstats(unsigned long long *out32, int reset) {
#else
extern "C" int mts_debug_blockstats_spectral(unsigned long long *out32, int reset) {
#endif
    if (hipMemcpyFromSymbol(out32, HIP_SYMBOL(mtsamd::g_blockstats), 48 * sizeof(unsigned long long)) != hipSuccess) return 1;
    if (reset) { unsigned long long z[48] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(mtsamd::g_blockstats), z, sizeof(z)) != hipSuccess) return 1; }
    return 0;
}
#endif
