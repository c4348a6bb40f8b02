// dscene.h -- flattened, device-resident scene: what the reference's plugins hold after construction,
// as POD records in HBM.  Built on the host by scene_host.cpp (the "plugin constructors"), read by the
// render kernels through wave-uniform (scalar) loads: every record below is indexed by values that
// are uniform across a wavefront except where noted, so the scene lives in SGPRs / the scalar cache
// and costs no vector memory traffic.  Citations are relative to /root/reference.
#pragma once
#include <stdint.h>

namespace mtsamd {

#define MTS_BVH_TOP_DEPTH 8           // levels 0..8 (at most 511 nodes, 16 KB) are laid out breadth-first and staged in LDS
#define MTS_BVH_LDS_NODES 511

struct DXf { float m[16]; float it[16]; };            // Transform4f: matrix + inverse transpose (transform.h:36-50)

struct DBBox { float min[3], max[3]; };

// Volume (render/texture.h:210-279, textures/grid3d.cpp, textures/constant3d.cpp)
struct DVolume {
    int32_t type;                 // MTS_VOLUME_*
    float value[3];
    float w2l[16];                // world_to_local matrix (row-major)
    int32_t affine;               // last row of w2l is exactly (0, 0, 0, 1): the homogeneous divide is x / 1
    DBBox bbox;
    const float *data;            // device pointer, nz*ny*nx*channels, x fastest
    int32_t nx, ny, nz, channels, filter, wrap;
    float max;
    int32_t has_max;
    int32_t columns_equal;        // every (y, x) column holds the same values (a 1-D, plane-parallel profile stored as a 3-D grid): the eight
                                  // corners of a cell are two distinct voxels (z0, z1) -- the lookups gather those two and run the same arithmetic
};

// Phase functions (phase/*.cpp); the tabulated distribution of tabphase (core/distr_1d.h:293-345)
struct DPhase {
    int32_t type;
    float g;
    int32_t child[2];
    int32_t weight_volume;
    const float *pdf, *cdf;       // device pointers
    int32_t size;
    float range_x, range_y, integral, normalization, interval_size, inv_interval_size;
    uint32_t valid_x, valid_y;
};

// Media (media/homogeneous.cpp, media/heterogeneous.cpp)
struct DMedium {
    int32_t type, sigma_t, albedo, phase;
    int32_t shared_grid;          // sigma_t and albedo are grids with identical transform / resolution / filter / wrap
    int32_t grey;                 // every channel of sigma_t / albedo / combined extinction carries the same value
    float scale;
    int32_t sample_emitters, has_spectral_extinction, is_homogeneous;
    float max_density;
    // RN(1 / max_density) when the kernels may divide by the majorant through it (div_by_invariant, volpath_flat.h): 0 otherwise
    // (homogeneous media, or a majorant whose reciprocal is not safely representable)
    float inv_max_density;
    DBBox aabb;
    // MI355X layout: when sigma_t and albedo are single-channel trilinear clamp-mode grids sharing one transform, the host also
    // uploads them interleaved, voxel by voxel {sigma_t, albedo} (padded by one voxel): the two x-neighbours of both grids are
    // 16 contiguous bytes, so a lookup is 4 wide gathers instead of 16 dword gathers.  NULL otherwise.
    const float *pair_grid;
    // ... together with a copy of what a lookup needs from the sigma_t volume record (one scalar load for the whole step
    // instead of the chain medium -> volume): world_to_local, resolution, affine flag.  Valid when pair_grid != NULL.
    float pair_w2l[16];
    int32_t pair_nx, pair_ny, pair_nz, pair_affine;      // pair_affine: bit 0 = affine, bit 1 = both grids have equal columns (DVolume::columns_equal)
};

struct DBsdf { int32_t type; float reflectance[3], rho_0[3], k[3], g[3], rho_c[3]; uint32_t flags; float transmittance[3]; };

// Shapes (shapes/rectangle.cpp, shapes/cube.cpp + librender/mesh.cpp, shapes/sphere.cpp)
struct DShape {
    int32_t type;
    DXf to_world, to_object;
    int32_t bsdf, interior, exterior, emitter;   // bsdf always resolved (default diffuse appended by the host)
    float frame_s[3], frame_t[3], frame_n[3];     // rectangle.cpp:66-74
    float inv_surface_area;
    float surface_area;                           // disk (disk.cpp:108-111)
    float center[3], radius;                      // sphere
    int32_t flip_normals;
    int32_t vertex_offset, face_offset;           // into the scene-wide mesh arrays
    int32_t has_normals, has_texcoords;
    int32_t is_medium_transition;
    int32_t bsdf_type; uint32_t bsdf_flags;       // copies of bsdfs[bsdf].type / .flags: spares the surface blocks a dependent record load
    int32_t prim_offset;                          // index of this shape's first primitive in prim order (DScene::tri_attr)
    int32_t area_lo, area_hi;                     // meshes: first / last face with a non-zero area (distr_1d.h:60-75); -1 = none
};

struct DPrim { int32_t shape, index; };
// Everything the primitive-list walk needs about one primitive in ONE 64-byte record (one scalar load per primitive instead of a
// chain prim -> shape -> geometry): triangle: p0, e1, e2; rectangle: rows 0..2 of to_object; sphere: center, radius.
struct DWalkPrim { int32_t type, shape, index, pad; float f[12]; };

struct DEmitter { int32_t type; DXf to_world; float radiance[3]; int32_t shape; float bsphere_center[3], bsphere_radius; };

struct DRFilter { int32_t type; float radius, stddev, alpha, bias; const float *values /* device, 32 entries (rfilter.cpp:9-20) */; float scale_factor; int32_t border_size; };

struct DSensor {
    int32_t type;
    DXf to_world;
    float s2c[16];                 // sample_to_camera matrix (perspective.cpp:107-111)
    float near_clip, far_clip, ppo[2];
    int32_t direction_type, flip_directions, target_type;
    float target_point[3];
    DShape target_shape;
    float target_area;
    float bsphere_center[3], bsphere_radius;
    int32_t needs_aperture_sample, medium;
    float shutter_open_time;       // > 0: one more draw per sample (integrator.cpp:248-250)
    int32_t origin_type;           // distant / distantflux: 1 = project the target onto origin_shape (distant.cpp:368-375)
    DShape origin_shape;
    int32_t width, height, crop_x, crop_y, crop_w, crop_h;
    DRFilter rfilter;
    int32_t sample_count;
    int32_t wavefront;             // 1: one TEA-seeded stream per (pixel, sample) as the reference's gpu_* variants (seed_wavefront_sample)
    uint64_t seed;
    const float *multi;            // mradiancemeter / mdistant: multi_count 4x4 matrices (device pointer)
    int32_t multi_count;
};

// Spectra (spectral variants): `uniform` (spectra/uniform.cpp:34-62: value inside [lambda_min, lambda_max], 0 outside) and `regular`
// (spectra/regular.cpp: a ContinuousDistribution over [lambda_min, lambda_max], core/distr_1d.h:378-400 eval_pdf)
// `irregular` (spectra/irregular.cpp: values at the nodes `wavelengths`) and `discrete` (spectra/discrete.cpp: a sampling-only sensor
// response: `wavelengths`, `values` and the running sum of the pmf, DiscreteDistribution, core/distr_1d.h:49-83)
struct DSpectrum { int32_t type; float value, lambda_min, lambda_max; const float *values; int32_t count; float inv_interval_size;
                   const float *wavelengths, *cdf; float cdf_sum; uint32_t valid_x, valid_y; };
enum { MTS_BSDF_SP_REFLECTANCE = 0, MTS_BSDF_SP_RHO_0, MTS_BSDF_SP_K, MTS_BSDF_SP_G, MTS_BSDF_SP_RHO_C, MTS_BSDF_SP_TRANSMITTANCE, MTS_BSDF_SP_COUNT };
// constvolume: the spectrum of its value; gridvolume_spectral (textures/gridvolume_spectral.cpp): the spectral interval its channels cover
struct DVolumeSp { int32_t value_sp; int32_t spectral_grid; float lambda_min, lambda_max; };

struct DIntegrator { int32_t type, max_depth, rr_depth, hide_emitters, use_spectral_mis, monochrome; };

// One spiral block (librender/spiral.cpp:27-72) assigned to this launch
// Scene traits (integrator_dev.h: MTS_TRAITS; scene_host.cpp: scene_traits; capi.cpp: the choice of the lean translation unit).
#define MT_MEDIA 1              // every medium: heterogeneous with spectral extinction and -- rgb / mono: grey, on a pair grid (DMedium::pair_grid);
                                // spectral variant: two gridvolume_spectral grids sharing geometry and interval (DMedium::shared_grid == 2)
#define MT_NO_BVH 2             // the primitive list is walked (no BVH)
#define MT_NO_SPHERE 4          // no sphere shapes
#define MT_NO_GRID_EVAL 8       // no grid volume is evaluated through volume_eval() (media go through their pair grids; no grid as a blend weight ...)
#define MT_NO_SHAPE_EMITTER 16  // no area emitters (shape_sample_direction)
#define MT_NO_PHASE_TREE 32     // no nested blendphase
#define MT_NO_RPV 64            // no rpv BSDF
#define MT_HOMOG 128            // every medium homogeneous (excludes MT_MEDIA)
// what each lean translation unit promises (kernels_lean_*.hip) -- mts_render launches the first one in this order whose promises a scene keeps
#define MT_UNIT_A (MT_MEDIA | MT_NO_BVH | MT_NO_SPHERE | MT_NO_GRID_EVAL | MT_NO_SHAPE_EMITTER | MT_NO_PHASE_TREE | MT_NO_RPV)
#define MT_UNIT_B (MT_MEDIA | MT_NO_BVH | MT_NO_SPHERE | MT_NO_SHAPE_EMITTER | MT_NO_PHASE_TREE)      // also unit s (spectral variant)
#define MT_UNIT_C (MT_MEDIA | MT_NO_SPHERE | MT_NO_SHAPE_EMITTER | MT_NO_PHASE_TREE)
#define MT_UNIT_H (MT_HOMOG | MT_NO_BVH | MT_NO_SPHERE | MT_NO_GRID_EVAL | MT_NO_SHAPE_EMITTER | MT_NO_PHASE_TREE)
#define MT_UNIT_P (MT_NO_BVH | MT_NO_SPHERE | MT_NO_GRID_EVAL | MT_NO_PHASE_TREE | MT_NO_RPV)          // compiled with these ...
#define MT_UNIT_P_NEEDS (MT_NO_BVH | MT_NO_SPHERE | MT_NO_RPV)                                          // ... of which `path` can reach these

struct DBlock { int32_t ox, oy, sx, sy; uint32_t id; uint32_t sample_base; /* wavefront streams: first sample index this entry renders */
                uint32_t film_off_lo, film_off_hi; /* floats from the launch's film pointer to the film this entry adds to: the slot of its pass (capi.cpp) */ };
static_assert(sizeof(DBlock) == 32, "DBlock is read as eight dwords (volpath_flat.h: wg_env)");

struct DScene {
    const DVolume *volumes;
    const DPhase *phases;
    const DMedium *media;
    const DBsdf *bsdfs;
    const DShape *shapes;
    const DPrim *prims;
    const DWalkPrim *walk;                          // prim order, one record per primitive
    const DEmitter *emitters;
    const float *positions, *normals, *texcoords;   // world-space mesh data of all meshes
    const uint32_t *faces;
    const float *tri;                               // per primitive (prim order): p0, e1 = p1 - p0, e2 = p2 - p0 (triangles only)
    const float *tri_attr;                          // per primitive, 24 floats: p0 p1 p2, n0 n1 n2, uv0 uv1 uv2 of a triangle -- what hit_point()
                                                    // and complete_surface() need, in one place instead of behind the face indices
    const float *area_pmf, *area_cdf;               // per primitive (prim order): face area and its running sum within the mesh
                                                    // (Mesh::build_pmf, mesh.cpp:285-312); 0 for other primitives
    // Bounding-volume hierarchy over the primitives, built by the host for scenes with many primitives (NULL otherwise: the
    // primitive list is walked).  32-byte nodes: bmin[3], bmax[3] (conservatively enlarged), skip = the node to visit when this
    // subtree is missed or done, link = (first << 3 | count) into bvh_prims for a leaf, minus the index of the left child for an
    // inner node (the right child is left.skip).  The top MTS_BVH_TOP_DEPTH levels come first, in breadth-first order: the
    // per-lane kernels stage them in LDS (bvh_lds / bvh_lds_count, set by the kernel on its copy of this record).
    const float *bvh_nodes;
    const int32_t *bvh_prims;                       // leaf contents: primitive indices (prim order decides ties, kdtree.h:2152-2154)
    int32_t bvh_node_count;
    const float *bvh_lds; int32_t bvh_lds_count;
    int32_t volume_count, phase_count, medium_count, bsdf_count, shape_count, prim_count, emitter_count;
    int32_t environment;
    DBBox bbox;
    DSensor sensor;
    DIntegrator integrator;
    // Spectral variants only (gpu_spectral; NULL / unused otherwise).  They sit BEHIND the records the rgb kernels load, in arrays of
    // their own, so that the rgb records -- and the scalar loads that fetch them -- stay exactly as they are.
    const DSpectrum *spectra;                       // every spectrum of the scene (spectra/uniform.cpp, spectra/regular.cpp; d65 expands to regular)
    const int32_t *bsdf_sp;                         // per BSDF, MTS_BSDF_SP_COUNT indices into spectra: reflectance, rho_0, k, g, rho_c, transmittance
    const int32_t *emitter_sp;                      // per emitter: radiance / irradiance / intensity
    const DVolumeSp *volume_sp;                     // per volume
    const float *cie;                               // CIE 1931 x, y, z, 95 samples each over 360 .. 830 nm (core/spectrum.h:127-133)
    int32_t srf;                                    // the sensor's spectral response function: index into spectra, -1 = none (perspective.cpp:173-182)
    // integrators/nbins.cpp (bin_mode 1: wavelength bin_lo[i] +- bin_hi[i]) / bins.cpp (2: interval [bin_lo[i], bin_hi[i]]): two AOV
    // channels per bin behind X, Y, Z, A, W; film_channels = 5 + 2 bin_count floats per pixel
    int32_t bin_mode, bin_count;
    const float *bin_lo, *bin_hi;
    int32_t film_channels;
};

// bsdf.h:38-124
enum : uint32_t { F_Null = 0x1, F_DiffuseReflection = 0x2, F_DiffuseTransmission = 0x4, F_GlossyReflection = 0x8,
                  F_FrontSide = 0x8000, F_BackSide = 0x10000,
                  F_Smooth = 0x2 | 0x4 | 0x8 | 0x10, F_Delta = 0x1 | 0x20 | 0x40 };

} // namespace mtsamd
