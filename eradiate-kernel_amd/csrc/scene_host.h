// scene_host.h -- host-side scene object behind the opaque mts_scene handle.
#pragma once
#include <vector>
#include <string>
#include <memory>
#include <atomic>
#include <stdexcept>
#include <hip/hip_runtime_api.h>
#include "dscene.h"
#include "dmath.h"
#include "../../include/mtsamd.h"

namespace mtsamd {

struct HostScene {
    DScene scene;                               // kernel argument (device pointers filled by upload)
    mts_integrator integrator;
    std::vector<DVolume> volumes;
    std::vector<DPhase> phases;
    std::vector<DMedium> media;
    std::vector<DBsdf> bsdfs;
    std::vector<DShape> shapes;
    std::vector<DPrim> prims;
    std::vector<DWalkPrim> walk;
    std::vector<DEmitter> emitters;
    std::vector<float> positions, normals, texcoords;
    std::vector<uint32_t> faces;
    std::vector<float> tri;
    std::vector<float> tri_attr;
    std::vector<float> area_pmf, area_cdf;           // prim order (DScene::area_pmf / area_cdf)
    std::vector<float> bvh_nodes;                    // 8 floats per node (DScene::bvh_nodes); empty = no BVH
    std::vector<int32_t> bvh_prims;
    std::vector<float> rfilter_values;
    std::vector<std::vector<float>> grid_data, tab_pdf, tab_cdf;
    std::vector<float> multi_transforms;            // mradiancemeter / mdistant sub-sensor matrices
    std::vector<std::vector<float>> pair_data;       // per medium: interleaved {sigma_t, albedo} voxels (DMedium::pair_grid), or empty
    // spectral variant (DScene::spectra ...)
    std::vector<DSpectrum> spectra; std::vector<std::vector<float>> spectrum_values, spectrum_wavelengths, spectrum_cdf;
    std::vector<float> bin_lo, bin_hi;
    bool srf_lookup_by_wavelength = true;            // the response function's weights can be recovered from the sampled wavelengths (regrouping kernel)
    std::vector<int32_t> bsdf_sp, emitter_sp; std::vector<DVolumeSp> volume_sp;
    std::vector<void *> device_allocs;
    int traits = 0;                                  // promises of integrator_dev.h's scene traits this scene keeps (MT_* bits; scene_traits())
    int device = 0;
    bool uploaded = false;
    std::atomic<int> stop{0};
};

HostScene *build_host_scene(const mts_scene_desc *desc);   // throws std::runtime_error
void build_bvh(HostScene &hs);                 // fills bvh_nodes / bvh_prims when the scene has many primitives
void upload_host_scene(HostScene &hs, int device);          // throws std::runtime_error
void free_host_scene(HostScene *hs);

} // namespace mtsamd
