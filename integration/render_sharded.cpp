// integration/render_sharded.cpp -- libmtsamd.so driven from C++ alone: the multi-GPU pattern of INTEGRATION.md section 3 as a program.
//
// One process per GPU.  Every rank builds the same scene through the C ABI (include/mtsamd.h: plain records, no Python, no torch),
// renders the (pass, block) pairs with block_id % nranks == rank into a device-resident film on its own HIP stream, and ONE
// ncclReduce(sum) over RCCL / xGMI leaves the image on rank 0 -- the only exchange of the path.
//
//     hipcc -O2 -I include integration/render_sharded.cpp -L eradiate-kernel_amd -lmtsamd -lrccl -Wl,-rpath,$PWD/eradiate-kernel_amd -o /tmp/render_sharded
//     /tmp/render_sharded <rank> <nranks> <id-file> <out.f32> [samples_per_pass]      (one process per rank; rank r uses device r)
//
// The ncclUniqueId travels through <id-file> (rank 0 writes it, the others wait for it): no launcher needed.  With nranks = 1 the
// reduce is the identity and the film must equal what the Python binding renders of the same scene bit for bit -- that is
// tests/test_gpu_parity.py::test_c_abi_from_cpp_with_an_rccl_film_reduce, which compiles and runs this file on the GPU box.
// The scene: a homogeneous slab (null-BSDF cube, sigma_t 0.5, albedo 0.75, isotropic phase) in front of a diffuse wall, lit and
// seen along +z (identity camera, directional emitter), 64 x 48 x 32 spp, `volpath`.  Every transform is a translation times a
// scaling by powers of two, so the records below hold exactly what ScalarTransform4f computes for the Python twin.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#include "mtsamd.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_NCCL(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) { std::fprintf(stderr, "%s: %s\n", #x, ncclGetErrorString(r_)); return 3; } } while (0)
#define CHECK_MTS(x) do { if ((x) != 0) { std::fprintf(stderr, "%s: %s\n", #x, mts_last_error()); return 4; } } while (0)

// translate(t) * scale(s): matrix and inverse transpose, row-major (Transform4f, include/mitsuba/core/transform.h:36-50)
static mts_transform translate_scale(float tx, float ty, float tz, float sx, float sy, float sz) {
    mts_transform t; std::memset(&t, 0, sizeof(t));
    t.matrix[0] = sx; t.matrix[5] = sy; t.matrix[10] = sz; t.matrix[15] = 1.f; t.matrix[3] = tx; t.matrix[7] = ty; t.matrix[11] = tz;
    t.inverse_transpose[0] = 1.f / sx; t.inverse_transpose[5] = 1.f / sy; t.inverse_transpose[10] = 1.f / sz; t.inverse_transpose[15] = 1.f;
    t.inverse_transpose[12] = -tx * (1.f / sx); t.inverse_transpose[13] = -ty * (1.f / sy); t.inverse_transpose[14] = -tz * (1.f / sz);
    return t;
}

int main(int argc, char **argv) {
    if (argc < 5) { std::fprintf(stderr, "usage: %s rank nranks id-file out.f32 [samples_per_pass]\n", argv[0]); return 1; }
    const int rank = std::atoi(argv[1]), nranks = std::atoi(argv[2]);
    const char *id_file = argv[3], *out_file = argv[4];
    const int samples_per_pass = argc > 5 ? std::atoi(argv[5]) : -1;
    if (mts_abi_version() != MTS_ABI_VERSION) { std::fprintf(stderr, "libmtsamd.so has ABI %d, this program was written against %d\n", mts_abi_version(), MTS_ABI_VERSION); return 1; }
    if (mts_abi_sizeof("mts_scene_desc") != (int) sizeof(mts_scene_desc) || mts_abi_sizeof("mts_sensor") != (int) sizeof(mts_sensor)) { std::fprintf(stderr, "record sizes differ\n"); return 1; }

    // ---- the scene, record by record (each field named after the Properties key of the reference plugin it stands for)
    mts_volume volumes[2]; std::memset(volumes, 0, sizeof(volumes));
    for (int k = 0; k < 2; ++k) {
        volumes[k].type = MTS_VOLUME_CONST; volumes[k].to_world = translate_scale(0, 0, 0, 1, 1, 1); volumes[k].value_spectrum = -1;
        for (int c = 0; c < 3; ++c) volumes[k].value[c] = k == 0 ? 0.5f : 0.75f;                 // sigma_t, albedo
    }
    mts_phase phase; std::memset(&phase, 0, sizeof(phase)); phase.type = MTS_PHASE_ISOTROPIC; phase.child[0] = phase.child[1] = phase.weight_volume = -1;
    mts_medium medium; std::memset(&medium, 0, sizeof(medium));
    medium.type = MTS_MEDIUM_HOMOGENEOUS; medium.sigma_t_volume = 0; medium.albedo_volume = 1; medium.scale = 1.f; medium.phase = 0;
    medium.sample_emitters = 1; medium.has_spectral_extinction = 1;
    mts_bsdf bsdfs[2]; std::memset(bsdfs, 0, sizeof(bsdfs));
    bsdfs[0].type = MTS_BSDF_NULL; bsdfs[1].type = MTS_BSDF_DIFFUSE;
    for (int c = 0; c < 3; ++c) bsdfs[1].reflectance[c] = 0.5f;
    for (int k = 0; k < 2; ++k) for (int j = 0; j < 6; ++j) bsdfs[k].spectrum[j] = -1;
    mts_shape shapes[2]; std::memset(shapes, 0, sizeof(shapes));
    shapes[0].type = MTS_SHAPE_RECTANGLE; shapes[0].to_world = translate_scale(0, 0, 10, 8, 8, 8); shapes[0].flip_normals = 1;       // the wall, facing the camera
    shapes[0].bsdf = 1; shapes[0].interior_medium = shapes[0].exterior_medium = shapes[0].emitter = -1; shapes[0].radius = 1.f;
    shapes[1].type = MTS_SHAPE_CUBE; shapes[1].to_world = translate_scale(0, 0, 8, 4, 4, 1);                                          // the slab: [-4, 4]^2 x [7, 9]
    shapes[1].bsdf = 0; shapes[1].interior_medium = 0; shapes[1].exterior_medium = -1; shapes[1].emitter = -1; shapes[1].radius = 1.f;
    mts_emitter sun; std::memset(&sun, 0, sizeof(sun));
    sun.type = MTS_EMITTER_DIRECTIONAL; sun.to_world = translate_scale(0, 0, 0, 1, 1, 1); sun.shape = -1; sun.radiance_spectrum = -1;  // local +z = direction of propagation
    for (int c = 0; c < 3; ++c) sun.radiance[c] = 1.f;

    mts_scene_desc d; std::memset(&d, 0, sizeof(d));
    d.abi_version = MTS_ABI_VERSION;
    d.volumes = volumes; d.volume_count = 2; d.phases = &phase; d.phase_count = 1; d.media = &medium; d.medium_count = 1;
    d.bsdfs = bsdfs; d.bsdf_count = 2; d.shapes = shapes; d.shape_count = 2; d.emitters = &sun; d.emitter_count = 1;
    mts_sensor &s = d.sensor;
    s.type = MTS_SENSOR_PERSPECTIVE; s.to_world = translate_scale(0, 0, 0, 1, 1, 1); s.fov_x = 45.f; s.near_clip = 0.1f; s.far_clip = 100.f;
    s.film_width = 64; s.film_height = 48; s.crop_size[0] = 64; s.crop_size[1] = 48;
    s.rfilter_type = MTS_RFILTER_BOX; s.rfilter_radius = 0.5f; s.rfilter_stddev = 0.5f;
    s.sample_count = 32; s.sampler_seed = 0; s.medium = -1;
    s.distant_target_shape.bsdf = s.distant_target_shape.interior_medium = s.distant_target_shape.exterior_medium = s.distant_target_shape.emitter = -1;
    s.distant_origin_shape = s.distant_target_shape;
    mts_integrator &it = d.integrator;
    it.type = MTS_INTEGRATOR_VOLPATH; it.max_depth = -1; it.rr_depth = 5; it.block_size = 32; it.samples_per_pass = samples_per_pass; it.timeout = -1.f;
    it.use_spectral_mis = 1;

    // ---- one GPU, one stream, one communicator per rank
    CHECK_HIP(hipSetDevice(rank));
    hipStream_t stream; CHECK_HIP(hipStreamCreate(&stream));
    ncclUniqueId id;
    if (rank == 0) {
        CHECK_NCCL(ncclGetUniqueId(&id));
        std::FILE *f = std::fopen((std::string(id_file) + ".tmp").c_str(), "wb"); if (!f) return 1;
        std::fwrite(&id, sizeof(id), 1, f); std::fclose(f); std::rename((std::string(id_file) + ".tmp").c_str(), id_file);
    } else {
        std::FILE *f = nullptr;
        for (int tries = 0; tries < 6000 && !(f = std::fopen(id_file, "rb")); ++tries) std::this_thread::sleep_for(std::chrono::milliseconds(10));
        if (!f || std::fread(&id, sizeof(id), 1, f) != 1) { std::fprintf(stderr, "rank %d: no id file\n", rank); return 1; }
        std::fclose(f);
    }
    ncclComm_t comm; CHECK_NCCL(ncclCommInitRank(&comm, nranks, id, rank));

    mts_scene *scene = nullptr;
    CHECK_MTS(mts_scene_create(&d, rank, &scene));
    const size_t film_floats = (size_t) s.crop_size[0] * s.crop_size[1] * 5;                     // X, Y, Z, A, W
    float *d_film = nullptr; CHECK_HIP(hipMalloc((void **) &d_film, film_floats * sizeof(float)));
    mts_render_opts opts; std::memset(&opts, 0, sizeof(opts));
    opts.shard_index = rank; opts.shard_count = nranks; opts.device = rank; opts.stream = (void *) stream;
    opts.film_on_device = 1; opts.film_capacity = (int64_t) film_floats;
    mts_stats stats;
    CHECK_MTS(mts_render(scene, &opts, d_film, &stats));
    CHECK_NCCL(ncclReduce(d_film, d_film, film_floats, ncclFloat, ncclSum, /*root*/ 0, comm, stream));      // the film of SURVEY.md 8(e)
    CHECK_HIP(hipStreamSynchronize(stream));
    if (rank == 0) {
        std::vector<float> film(film_floats);
        CHECK_HIP(hipMemcpy(film.data(), d_film, film_floats * sizeof(float), hipMemcpyDeviceToHost));
        std::FILE *f = std::fopen(out_file, "wb"); if (!f) return 1;
        std::fwrite(film.data(), sizeof(float), film_floats, f); std::fclose(f);
        std::printf("rank 0 of %d: %llu samples here, kernel %.2f ms, variant %d, film written to %s\n", nranks, (unsigned long long) stats.samples,
                    stats.kernel_ms, stats.kernel_variant, out_file);
    }
    CHECK_NCCL(ncclCommDestroy(comm));
    mts_scene_destroy(scene);
    (void) hipFree(d_film); (void) hipStreamDestroy(stream);
    return 0;
}
