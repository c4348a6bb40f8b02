// launch.h -- host-callable launchers of the kernels in kernels.hip.
#pragma once
#include <hip/hip_runtime_api.h>
#include "dscene.h"

namespace mtsamd {

// variant: 0 = nested formulation, 1 = flat per-lane state machine, 256 / 512 / 1024 = workgroup-regrouping kernel with
// that workgroup size (needs a workspace of render_workspace_floats() floats for the cold path state)
size_t render_workspace_floats(uint64_t paths, int variant);
hipError_t launch_render(const DScene &sc, const DBlock *d_blocks, uint32_t n_blocks, uint32_t block_size, uint32_t sample_count,
                         float *d_film, unsigned long long *d_counters, bool count, int variant, int wg_threads /* 0 = one thread per path */, float *d_workspace,
                         const uint32_t *d_stop_flag /* host-visible word polled by the kernels: non-zero = stop */,
                         const uint32_t *d_tiles /* cost-sorted tiles of the regrouping kernels (volpath_flat.h, WgArgs::tiles) or NULL */, uint32_t n_tiles,
                         hipStream_t stream);
// the same for a scene of the spectral variant (kernels_spectral.hip)
hipError_t launch_render_spectral(const DScene &sc, const DBlock *d_blocks, uint32_t n_blocks, uint32_t block_size, uint32_t sample_count,
                                  float *d_film, unsigned long long *d_counters, bool count, int variant, int wg_threads, float *d_workspace,
                                  const uint32_t *d_stop_flag, const uint32_t *d_tiles, uint32_t n_tiles, hipStream_t stream);
// film = sum over k < count of the film-sized slot k of d_slots, added in slot order (the passes of a render: capi.cpp)
hipError_t launch_film_sum_slots(float *d_film, const float *d_slots, size_t film_floats, uint32_t count, hipStream_t stream);
// the regrouping kernels of rgb / mono `volpath` (variant 11024) and `volpathmis` (10512) for scenes that keep the promises of
// kernels_lean_a.hip / kernels_lean_b.hip (integrator_dev.h: MTS_TRAITS); anything else: hipErrorInvalidConfiguration
hipError_t launch_render_lean_a(const DScene &sc, const DBlock *d_blocks, uint32_t n_blocks, uint32_t block_size, uint32_t sample_count,
                                float *d_film, unsigned long long *d_counters, bool count, int variant, int wg_threads, float *d_workspace,
                                const uint32_t *d_stop_flag, const uint32_t *d_tiles, uint32_t n_tiles, hipStream_t stream);
hipError_t launch_render_lean_b(const DScene &sc, const DBlock *d_blocks, uint32_t n_blocks, uint32_t block_size, uint32_t sample_count,
                                float *d_film, unsigned long long *d_counters, bool count, int variant, int wg_threads, float *d_workspace,
                                const uint32_t *d_stop_flag, const uint32_t *d_tiles, uint32_t n_tiles, hipStream_t stream);
hipError_t launch_render_lean_h(const DScene &sc, const DBlock *d_blocks, uint32_t n_blocks, uint32_t block_size, uint32_t sample_count,
                                float *d_film, unsigned long long *d_counters, bool count, int variant, int wg_threads, float *d_workspace,
                                const uint32_t *d_stop_flag, const uint32_t *d_tiles, uint32_t n_tiles, hipStream_t stream);      // kernels_lean_h.hip: homogeneous media
hipError_t launch_render_lean_c(const DScene &sc, const DBlock *d_blocks, uint32_t n_blocks, uint32_t block_size, uint32_t sample_count,
                                float *d_film, unsigned long long *d_counters, bool count, int variant, int wg_threads, float *d_workspace,
                                const uint32_t *d_stop_flag, const uint32_t *d_tiles, uint32_t n_tiles, hipStream_t stream);      // kernels_lean_c.hip: as b, with a BVH
// ... and of the spectral variant's 256-path machines (variant 10256; kernels_lean_s.hip)
hipError_t launch_render_lean_s(const DScene &sc, const DBlock *d_blocks, uint32_t n_blocks, uint32_t block_size, uint32_t sample_count,
                                float *d_film, unsigned long long *d_counters, bool count, int variant, int wg_threads, float *d_workspace,
                                const uint32_t *d_stop_flag, const uint32_t *d_tiles, uint32_t n_tiles, hipStream_t stream);
// ... and `path` as the flat loop (variant 1) for scenes without a BVH, spheres and rpv: kernels_lean_p.hip (rgb / mono), _ps.hip (spectral)
hipError_t launch_render_lean_p(const DScene &sc, const DBlock *d_blocks, uint32_t n_blocks, uint32_t block_size, uint32_t sample_count,
                                float *d_film, unsigned long long *d_counters, bool count, int variant, int wg_threads, float *d_workspace,
                                const uint32_t *d_stop_flag, const uint32_t *d_tiles, uint32_t n_tiles, hipStream_t stream);
hipError_t launch_render_lean_ps(const DScene &sc, const DBlock *d_blocks, uint32_t n_blocks, uint32_t block_size, uint32_t sample_count,
                                 float *d_film, unsigned long long *d_counters, bool count, int variant, int wg_threads, float *d_workspace,
                                 const uint32_t *d_stop_flag, const uint32_t *d_tiles, uint32_t n_tiles, hipStream_t stream);
hipError_t launch_sample(const DScene &sc, int32_t n, uint64_t seed_offset, const float *d_rays, float *d_rgb, uint8_t *d_valid, hipStream_t stream);
// spectral variant (kernels_spectral.hip): per-ray wavelengths (4 n floats), four-wide result
hipError_t launch_sample_spectral(const DScene &sc, int32_t n, uint64_t seed_offset, const float *d_rays, const float *d_wavelengths, float *d_spec, uint8_t *d_valid,
                                  hipStream_t stream);
hipError_t launch_intersect(const DScene &sc, int32_t n, const float *o, const float *d, const float *mint, const float *maxt,
                            float *t, int32_t *shape, int32_t *prim, float *p, float *nn, hipStream_t stream);

hipError_t launch_tea(int32_t n, const uint32_t *v0, const uint32_t *v1, int rounds, uint32_t *out32, uint64_t *out64, float *outf, hipStream_t stream);
hipError_t launch_wavefront_sampler(int32_t lanes, uint64_t seed_value, int32_t count, float *out, hipStream_t stream);

} // namespace mtsamd
