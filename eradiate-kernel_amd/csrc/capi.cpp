// capi.cpp -- the extern "C" entry points declared in include/mtsamd.h.
//
// -DMTSAMD_HOST_ONLY (tests/test_host_sanitizers.py: the host pass alone, built with AddressSanitizer + UndefinedBehaviorSanitizer):
// everything that validates and flattens caller-owned records runs as in the product -- mts_scene_create up to the upload, mts_render
// up to the first device call (options, passes, spiral, shard filter, film capacity) -- and every entry point that would touch the GPU
// reports "host-only build" instead.
//
// mts_render mirrors SamplingIntegrator::render (/root/reference/src/librender/integrator.cpp:51-179):
// pass / block bookkeeping on the host, one kernel launch per pass over every spiral block this shard
// owns.  No exception crosses the boundary: errors become a non-zero status + mts_last_error().
#include <algorithm>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <mutex>
#include <thread>
#include <atomic>
#include <signal.h>
#include "scene_host.h"
#include "launch.h"

using namespace mtsamd;

static thread_local std::string g_error;

#define API_TRY try {
#define API_CATCH } catch (const std::exception &e) { g_error = e.what(); return 1; } catch (...) { g_error = "unknown error"; return 1; } return 0;
#define HIP_CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) throw std::runtime_error(std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)

// Device buffers and events a render needs besides the scene; they are kept with the handle and only grow, so that a sequence of
// renders of one scene (passes, sensors swept by the caller, benchmark steps) pays for hipMalloc / hipFree -- which synchronise
// the device -- once.  Guarded by render_mutex.
struct RenderCache {
    void *ptr[6] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };   // 0: film (host-film renders), 1: counters, 2: blocks, 3: workspace, 4: tile table, 5: film slots of the passes
    size_t cap[6] = { 0, 0, 0, 0, 0, 0 };
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    void *get(int k, size_t bytes) {
        bytes = std::max<size_t>(bytes, 16);
        if (cap[k] < bytes) {
            if (ptr[k]) { (void) hipFree(ptr[k]); ptr[k] = nullptr; cap[k] = 0; }
            hipError_t e = hipMalloc(&ptr[k], bytes);
            if (e != hipSuccess) throw std::runtime_error(std::string("hipMalloc failed: ") + hipGetErrorString(e));
            cap[k] = bytes;
        }
        return ptr[k];
    }
    void events() {
        if (!ev0) { if (hipEventCreate(&ev0) != hipSuccess || hipEventCreate(&ev1) != hipSuccess) throw std::runtime_error("hipEventCreate failed"); }
    }
    void release() {
        for (int k = 0; k < 6; ++k) if (ptr[k]) { (void) hipFree(ptr[k]); ptr[k] = nullptr; cap[k] = 0; }
        if (ev0) { (void) hipEventDestroy(ev0); ev0 = nullptr; }
        if (ev1) { (void) hipEventDestroy(ev1); ev1 = nullptr; }
    }
};
// stop_word: one word of pinned, device-visible host memory per scene -- the m_stop of Integrator::cancel() (integrator.cpp:43-45)
// as the kernels see it (stop_requested(), the stop word of the ring driver).  mts_cancel stores to it from any thread.
struct mts_scene { HostScene *hs; std::mutex render_mutex; RenderCache cache; volatile uint32_t *stop_word = nullptr; };

// librender/spiral.cpp:11-72
namespace {
struct Spiral {
    int size_x, size_y, off_x, off_y, block_size, blocks_x, blocks_y;
    size_t block_count, block_counter, remaining_passes;
    int dir, pos_x, pos_y, steps_left, steps;
    void init(int sx, int sy, int ox, int oy, int bs, size_t passes) {
        size_x = sx; size_y = sy; off_x = ox; off_y = oy; block_size = bs; remaining_passes = passes;
        blocks_x = (int) std::ceil((float) sx / bs); blocks_y = (int) std::ceil((float) sy / bs);
        block_count = (size_t) blocks_x * blocks_y;
        reset();
    }
    void reset() { block_counter = 0; dir = 0; pos_x = blocks_x / 2; pos_y = blocks_y / 2; steps_left = 1; steps = 1; }
    bool next_block(DBlock &b, size_t &block_id) {
        if (block_count == block_counter) {
            if (remaining_passes > 1) { --remaining_passes; reset(); }
            else return false;
        }
        block_id = block_counter + (remaining_passes - 1) * block_count;
        int offx = pos_x * block_size, offy = pos_y * block_size;
        b.sx = std::min(block_size, size_x - offx); b.sy = std::min(block_size, size_y - offy);
        b.ox = offx + off_x; b.oy = offy + off_y; b.film_off_lo = b.film_off_hi = 0;
        ++block_counter;
        if (block_counter != block_count) {
            do {
                switch (dir) { case 0: ++pos_x; break; case 1: ++pos_y; break; case 2: --pos_x; break; case 3: --pos_y; break; }
                if (--steps_left == 0) { dir = (dir + 1) % 4; if (dir == 2 || dir == 0) ++steps; steps_left = steps; }
            } while (pos_x < 0 || pos_y < 0 || pos_x >= blocks_x || pos_y >= blocks_y);
        }
        return true;
    }
};

template <typename T> struct DeviceBuffer {
    T *p = nullptr;
    explicit DeviceBuffer(size_t n) { HIP_CHECK(hipMalloc((void **) &p, std::max<size_t>(n * sizeof(T), 16))); }
    ~DeviceBuffer() { if (p) (void) hipFree(p); }
    DeviceBuffer(const DeviceBuffer &) = delete;
};
} // namespace

extern "C" {

int mts_abi_version(void) { return MTS_ABI_VERSION; }
#ifndef MTSAMD_BUILD_ID
#define MTSAMD_BUILD_ID "unidentified...."
#endif
static const char g_build_id[] = "MTSAMD_BUILD_ID=" MTSAMD_BUILD_ID;   // also readable from the file without loading it (eradiate-kernel_amd/_buildid.py)
#ifndef MTSAMD_TOOLCHAIN_ID
#define MTSAMD_TOOLCHAIN_ID "unknown."
#endif
// the compiler that built it (hash of `hipcc --version`): build.py rebuilds when it changes; read from the file's bytes only
__attribute__((used)) static const char g_toolchain_id[] = "MTSAMD_TOOLCHAIN=" MTSAMD_TOOLCHAIN_ID;
const char *mts_build_id(void) { return g_build_id + 16; }
const char *mts_last_error(void) { return g_error.c_str(); }

int mts_abi_sizeof(const char *name) {
#define SZ(T) if (!strcmp(name, #T)) return (int) sizeof(T);
    SZ(mts_spectrum) SZ(mts_transform) SZ(mts_volume) SZ(mts_phase) SZ(mts_medium) SZ(mts_bsdf) SZ(mts_shape) SZ(mts_emitter)
    SZ(mts_sensor) SZ(mts_integrator) SZ(mts_scene_desc) SZ(mts_stats) SZ(mts_render_opts)
#undef SZ
    return -1;
}

#if defined(MTSAMD_HOST_ONLY)
#define HOST_ONLY_STOP(what) throw std::runtime_error(std::string(what) + ": host-only build (validated, nothing launched)")
#endif

int mts_device_count(int *count) {
    API_TRY
#if defined(MTSAMD_HOST_ONLY)
    (void) count; HOST_ONLY_STOP("mts_device_count");
#else
    HIP_CHECK(hipGetDeviceCount(count));
#endif
    API_CATCH
}

int mts_scene_create(const mts_scene_desc *desc, int device, mts_scene **out) {
    API_TRY
    if (!out) throw std::runtime_error("mts_scene_create: out is NULL");
    HostScene *hs = build_host_scene(desc);
#if defined(MTSAMD_HOST_ONLY)
    (void) device;
    mts_scene *s = new mts_scene(); s->hs = hs;
    s->stop_word = new uint32_t(0);
#else
    try { upload_host_scene(*hs, device); } catch (...) { free_host_scene(hs); throw; }
    mts_scene *s = new mts_scene(); s->hs = hs;
    void *sw = nullptr;
    if (hipHostMalloc(&sw, 64, hipHostMallocDefault) != hipSuccess) { free_host_scene(hs); delete s; throw std::runtime_error("hipHostMalloc failed (stop word)"); }
    s->stop_word = (volatile uint32_t *) sw; *s->stop_word = 0;
#endif
    *out = s;
    API_CATCH
}

// Not part of the ABI (include/mtsamd.h does not declare it): which promises of integrator_dev.h's scene traits a description keeps
// (scene_host.cpp: scene_traits) -- what decides which lean translation unit mts_render launches.  Host only: no device is touched, so the
// CPU test-suite pins the decision (tests/test_abi.py::test_scene_traits).
int mts_debug_scene_traits(const mts_scene_desc *desc, int32_t *traits) {
    API_TRY
    if (!traits) throw std::runtime_error("mts_debug_scene_traits: traits is NULL");
    HostScene *hs = build_host_scene(desc);
    *traits = hs->traits;
    free_host_scene(hs);
    API_CATCH
}

int mts_scene_destroy(mts_scene *scene) {
    if (scene) {
#if defined(MTSAMD_HOST_ONLY)
        delete (uint32_t *) scene->stop_word;
#else
        (void) hipSetDevice(scene->hs->device); scene->cache.release();
        if (scene->stop_word) (void) hipHostFree((void *) scene->stop_word);
#endif
        free_host_scene(scene->hs); delete scene;
    }
    return 0;
}

int mts_cancel(mts_scene *scene) {
    if (!scene) { g_error = "mts_cancel: scene is NULL"; return 1; }
    scene->hs->stop.store(1);
    if (scene->stop_word) *scene->stop_word = 1;          // seen by the running kernels within a few microseconds
    return 0;
}

// ---- the SIGINT scope (integrator_v.cpp:129-151).  The handler touches lock-free atomics, one word of pinned host memory,
// sigaction and raise: all async-signal-safe.
static std::atomic<mts_scene *> g_sigint_scene{nullptr};
static struct sigaction g_sigint_prev;
static_assert(std::atomic<int>::is_always_lock_free, "the stop flag is stored from a signal handler");
static void mts_sigint_handler(int) {
    mts_scene *s = g_sigint_scene.exchange(nullptr);
    if (!s) return;
    s->hs->stop.store(1);                                  // = mts_cancel
    if (s->stop_word) *s->stop_word = 1;
    (void) sigaction(SIGINT, &g_sigint_prev, nullptr);     // the previous handler sees the signal as well (Python: KeyboardInterrupt once
    (void) raise(SIGINT);                                  // the interpreter runs again, i.e. after the render has wound down)
}

int mts_sigint_scope_enter(mts_scene *scene) {
    if (!scene) { g_error = "mts_sigint_scope_enter: scene is NULL"; return 1; }
    mts_scene *expected = nullptr;
    if (!g_sigint_scene.compare_exchange_strong(expected, scene)) { g_error = "mts_sigint_scope_enter: another scope is open"; return 1; }
    struct sigaction sa; memset(&sa, 0, sizeof(sa));
    sa.sa_handler = mts_sigint_handler; sigemptyset(&sa.sa_mask);
    if (sigaction(SIGINT, &sa, &g_sigint_prev) != 0) { g_sigint_scene.store(nullptr); g_error = "mts_sigint_scope_enter: sigaction failed"; return 1; }
    return 0;
}

int mts_sigint_scope_exit(void) {
    // a handler that ran has already put the previous one back (and emptied the slot); otherwise do it here
    if (g_sigint_scene.exchange(nullptr)) (void) sigaction(SIGINT, &g_sigint_prev, nullptr);
    return 0;
}

int mts_render(mts_scene *scene, const mts_render_opts *opts_, float *film, mts_stats *stats) {
    API_TRY
    if (!scene || !film) throw std::runtime_error("mts_render: NULL argument");
    std::lock_guard<std::mutex> guard(scene->render_mutex);         // one render per handle at a time
    HostScene &hs = *scene->hs;
    mts_render_opts opts; memset(&opts, 0, sizeof(opts)); opts.shard_count = 1;
    if (opts_) opts = *opts_;
    if (opts.shard_count < 1 || opts.shard_index < 0 || opts.shard_index >= opts.shard_count) throw std::runtime_error("mts_render: invalid shard specification");
    auto t0 = std::chrono::steady_clock::now();
    hs.stop.store(0);                                               // integrator.cpp:53
#if !defined(MTSAMD_HOST_ONLY)
    HIP_CHECK(hipSetDevice(hs.device));
#endif
    hipStream_t stream = (hipStream_t) opts.stream;
    const DSensor &se = hs.scene.sensor;
    // integrator.cpp:58-65
    size_t total_spp = (size_t) se.sample_count;
    size_t samples_per_pass = hs.integrator.samples_per_pass < 0 ? total_spp : std::min((size_t) hs.integrator.samples_per_pass, total_spp);
    if (samples_per_pass == 0 || (total_spp % samples_per_pass) != 0)
        throw std::runtime_error("sample_count (" + std::to_string(total_spp) + ") must be a multiple of samples_per_pass (" + std::to_string(samples_per_pass) + ").");
    size_t n_passes = (total_spp + samples_per_pass - 1) / samples_per_pass;
    // integrator.cpp:26-32,89-97: the reference's heuristic depends on the host thread count; this
    // backend pins MTS_BLOCK_SIZE = 32 when the scene leaves block_size at 0
    uint32_t block_size = hs.integrator.block_size > 0 ? (uint32_t) hs.integrator.block_size : 32u;
    { uint32_t p = 1; while (p < block_size) p <<= 1; block_size = p; }
    if (block_size > 1024) throw std::runtime_error("block_size too large");
    // spiral.cpp: enumerate every (pass, block) pair in the reference's order; keep this shard's blocks
    Spiral spiral; spiral.init(se.crop_w, se.crop_h, se.crop_x, se.crop_y, (int) block_size, n_passes);
    // Passes are independent jobs (each block id seeds its own streams) whose results add up in the film, so the (pass, block)
    // pairs of this shard are launched together, MAX_BLOCKS_PER_LAUNCH at a time: the workspace (one 128-byte record per path in
    // flight) stays below 1 GiB however many passes samples_per_pass asks for, and thread indices stay far below 2^32.
    // should_stop() (integrator.h:143-146) is honoured inside a launch: the kernels poll the scene's stop word.
    const size_t MAX_BLOCKS_PER_LAUNCH = std::max<size_t>(1, ((size_t) 8 << 20) / ((size_t) block_size * block_size));
    // Wavefront (gpu_*) streams make the samples of a pixel independent of each other (one stream per (pixel, sample)), so a film with
    // fewer pixels than the GPU has lanes to fill is spread over more workgroups: `split` entries per spiral block, each rendering
    // sample_count / split samples of every pixel of the block (DBlock::sample_base); their sums go to film slots of their own, added in sample order at the end (below).
    size_t split = 1;
    if (se.wavefront) {
        int cus = 256;
#if !defined(MTSAMD_HOST_ONLY)
        HIP_CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, hs.device));
#endif
        const size_t pixels = (size_t) se.crop_w * se.crop_h, target = (size_t) std::max(cus, 1) * 4096;      // four 1024-path workgroups' worth per CU
        if (const char *sv = getenv("MTSAMD_WAVEFRONT_SPLIT")) {
            char *end = nullptr; const long v = strtol(sv, &end, 10);
            if (end == sv || *end != '\0' || v < 1) throw std::runtime_error("MTSAMD_WAVEFRONT_SPLIT must be a positive integer");
            split = (size_t) v;
        }
        else while (pixels * split < target && split * 2 <= total_spp && total_spp % (split * 2) == 0) split *= 2;
        if (total_spp % split != 0) throw std::runtime_error("MTSAMD_WAVEFRONT_SPLIT must divide the sample count");
    }
    const size_t launch_spp = samples_per_pass / split;            // samples per pixel one entry of a launch renders
    const size_t film_floats = (size_t) se.crop_w * se.crop_h * (size_t) hs.scene.film_channels;     // X, Y, Z, A, W (+ two AOV channels per spectral bin)
    // The reference renders pass after pass and Film::put adds every finished block to the film (integrator.cpp:98-107,
    // imageblock.cpp:59-77): film = ((pass 1 + pass 2) + pass 3) + ...  Here the (pass, block) pairs of a shard run concurrently, so every
    // pass adds into a film-sized SLOT of its own (DBlock::film_off_*) and the slots are summed in pass order at the end
    // (launch_film_sum_slots): the same additions in the same order -- for the AOV channels of nbins / bins too, whose samples go
    // straight to the film by atomics.  Slots beyond 2 GiB are not allocated: the passes then meet in the one film in launch order.
    // The `split` entries of a block under wavefront streams (above) get slots of their own as well: their partial sums are added in
    // sample order instead of in the order their atomics happen to land -- the same film run after run.
    const size_t n_slots = n_passes * split;
    bool pass_slots = n_slots > 1 && (uint64_t) n_slots * film_floats * sizeof(float) <= ((uint64_t) 2 << 30);
    if (const char *ps = getenv("MTSAMD_PASS_SLOTS")) if (atoi(ps) == 0) pass_slots = false;
    std::vector<std::vector<DBlock>> pass_blocks(1);
    uint64_t samples = 0;
    for (size_t pass = 0; pass < n_passes; ++pass)
        for (size_t k = 0; k < spiral.block_count; ++k) {
            DBlock b; size_t id;
            if (!spiral.next_block(b, id)) throw std::runtime_error("spiral exhausted early");
            if ((int) (id % (size_t) opts.shard_count) != opts.shard_index) continue;
            if (id >= ((uint64_t) 1 << 32)) throw std::runtime_error("block id overflow");
            b.id = (uint32_t) id; b.sample_base = 0;
            for (size_t sub = 0; sub < split; ++sub) {               // wavefront streams: `split` entries share a block's samples
                b.sample_base = (uint32_t) (sub * launch_spp);
                { const uint64_t off = pass_slots ? (uint64_t) (pass * split + sub) * film_floats : 0; b.film_off_lo = (uint32_t) off; b.film_off_hi = (uint32_t) (off >> 32); }
                if (pass_blocks.back().size() >= MAX_BLOCKS_PER_LAUNCH) pass_blocks.emplace_back();
                pass_blocks.back().push_back(b);
            }
            samples += (uint64_t) b.sx * b.sy * samples_per_pass;
        }
    if (opts.film_capacity > 0 && (uint64_t) opts.film_capacity < (uint64_t) film_floats)
        throw std::runtime_error("mts_render: the film buffer holds " + std::to_string(opts.film_capacity) + " floats, this scene writes " + std::to_string(film_floats) +
                                 " (crop_width x crop_height x " + std::to_string(hs.scene.film_channels) + " channels: X, Y, Z, A, W + two per spectral bin)");
#if defined(MTSAMD_HOST_ONLY)
    (void) stream; (void) samples; (void) launch_spp; (void) t0; (void) stats; (void) pass_slots; (void) n_slots;
    HOST_ONLY_STOP("mts_render");
#else
    RenderCache &rc = scene->cache;
    float *d_film = film;
    if (!opts.film_on_device) d_film = (float *) rc.get(0, film_floats * sizeof(float));
    constexpr int N_COUNTERS = 16;                                   // [0..2] loop counters, [4..9] ring-stall record (volpath_flat.h, MTS_DIAG_BASE), [15] cost-recording flag
    // [16 + s]: cost of tile slot s of a calibration launch (16 pixels per tile: block_size^2 / 16 slots per block of the first chunk)
    unsigned long long *d_counters = (unsigned long long *) rc.get(1, (N_COUNTERS + std::max<size_t>(1, pass_blocks[0].size()) * ((size_t) block_size * block_size / 16u + 1u)) * sizeof(unsigned long long));
    HIP_CHECK(hipMemsetAsync(d_film, 0, film_floats * sizeof(float), stream));               // hdrfilm.cpp:201-203 (storage cleared by prepare())
    float *d_target = d_film;                                        // what the kernels add to: the film, or the slots of the passes
    if (pass_slots) {
        d_target = (float *) rc.get(5, n_slots * film_floats * sizeof(float));
        HIP_CHECK(hipMemsetAsync(d_target, 0, n_slots * film_floats * sizeof(float), stream));
    }
    HIP_CHECK(hipMemsetAsync(d_counters, 0, N_COUNTERS * sizeof(unsigned long long), stream));
    if (const char *inj = getenv("MTSAMD_TEST_INJECT_LOST_PATH")) {   // test hook of the ring drivers' error path (volpath_flat.h, MTS_INJECT_SLOT): idle bound in ticks
        const unsigned long long ticks = strtoull(inj, nullptr, 10);
        if (ticks != 0ull && opts.collect_counters) HIP_CHECK(hipMemcpyAsync(d_counters + 14, &ticks, sizeof(ticks), hipMemcpyHostToDevice, stream));
    }
    rc.events();
    hipEvent_t ev0 = rc.ev0, ev1 = rc.ev1;
    double kernel_ms = 0.0, calibration_ms = 0.0; int launches = 0, calibration_launches = 0, last_variant = 0; bool timed_out = false;
    const float timeout = hs.integrator.timeout;
    *scene->stop_word = 0;
    if (hs.stop.load()) *scene->stop_word = 1;                      // cancel() raced the start of the render
    auto should_stop = [&]() {                                      // integrator.h:143-146
        if (hs.stop.load()) return true;
        if (timeout > 0.f && std::chrono::duration<float>(std::chrono::steady_clock::now() - t0).count() > timeout) { timed_out = true; return true; }
        return false;
    };
    auto throw_on_ring_stall = [](const unsigned long long *c) {    // counters[4..9]: a bounded ring wait gave up (volpath_flat.h): an error, never a hang
        if (c[4] != 0)
            throw std::runtime_error("render kernel: " + std::string(c[4] == 3 ? "lost path (nothing waiting, finished paths = tail" : c[4] == 1 ? "ring stall (consumer" : "ring stall (producer") + ", ring " + std::to_string(c[5]) +
                                     ", index " + std::to_string(c[6]) + ", head " + std::to_string(c[7]) + ", tail " + std::to_string(c[8]) +
                                     ", workgroup " + std::to_string(c[9]) + ")");
    };
    try {
        // ---- kernel variant: MTSAMD_KERNEL = nested | flat | wga256 | wga512 | wga1024 (default) | wgl1024
        int variant = 11024;                                       // asynchronous regrouping, 1024 paths served by 1024 threads
        if (const char *kv = getenv("MTSAMD_KERNEL")) {
            if (!strcmp(kv, "nested")) variant = 0; else if (!strcmp(kv, "flat")) variant = 1;
            else if (!strcmp(kv, "wga256")) variant = 10256; else if (!strcmp(kv, "wga512")) variant = 10512; else if (!strcmp(kv, "wga1024")) variant = 11024;
            else if (!strcmp(kv, "wgl1024")) variant = 21024;
            else throw std::runtime_error("MTSAMD_KERNEL must be one of nested, flat, wga256, wga512, wga1024, wgl1024");
        }
        // without media there are no tracking walks to regroup: the per-lane kernels win (cornell box 512 x 512 x 256, volpath: rings 992,
        // per lane 1242 Msamples/s; `path` per lane: 2342, as one flat loop with regeneration 2910)
        if (!getenv("MTSAMD_KERNEL") && hs.media.empty() && hs.integrator.type != MTS_INTEGRATOR_PATH) variant = 0;
        if (hs.integrator.type == MTS_INTEGRATOR_PATH) variant = variant != 0 ? 1 : 0;   // per lane: flat loop with regeneration (every variant), or nested (MTSAMD_KERNEL=nested)
        // variant = family * 10000 + paths per workgroup (family 1: ring driver, 2: lane-affine driver)
        if (variant >= 10000) {
            int family = variant / 10000, wg = variant % 10000;
            if (hs.integrator.type == MTS_INTEGRATOR_VOLPATHMIS) { family = 1; wg = std::min(wg, 512); }   // four weight matrices per path: 512 paths fill the LDS
            if (hs.integrator.spectral) { family = 1; wg = std::min(wg, 256); }   // four-wide spectra: 42 hot dwords per path; three 256-path workgroups per CU (12 waves) beat one of 512 (8 waves) by 10 %
            // a workgroup of the regrouping kernels sits in ONE spiral block: blocks smaller than its path count get the largest
            // workgroup that divides them (16 x 16 -> 256 paths); only blocks below 256 pixels fall back to the per-lane kernel
            while (wg > 256 && (block_size * block_size) % (uint32_t) wg != 0) wg /= 2;
            if (wg != 1024) family = 1;                           // the lane-affine driver is built for 1024 paths only
            variant = (block_size * block_size) % (uint32_t) wg != 0 ? 1 : family * 10000 + wg;
        }
        if (hs.integrator.spectral && variant == 1 && hs.integrator.type != MTS_INTEGRATOR_PATH) variant = 0;   // the spectral build's per-lane flat kernel is `path`'s
        // AOV channels (nbins / bins) and a sensor response function: `volpath` and (round 4) `volpathmis` carry them on the regrouping
        // machines (their NEW blocks) and `path` in its flat loop (kernels.hip: path_pixel_flat); a discrete response function with repeated
        // wavelengths keeps the volumetric integrators per lane
        if ((hs.scene.bin_count > 0 || hs.scene.srf >= 0) && !(variant == 1 && hs.integrator.type == MTS_INTEGRATOR_PATH) &&
            !(variant >= 10000 && hs.integrator.type != MTS_INTEGRATOR_PATH && hs.srf_lookup_by_wavelength)) variant = 0;
        // Wavefront (gpu_*) streams carry their own PCG32 increment per (pixel, sample).  The regrouping machine of rgb / mono `volpath` keeps
        // only the generator's 64-bit state in LDS and recomputes the increment on every load (round 4: wg_block's WF instantiation, 1024-path
        // workgroups); everything else runs per lane, where the generator lives in registers: `volpath` as the flat state machine, the
        // others nested
        if (se.wavefront && variant >= 10000 &&
            !(variant == 11024 && hs.integrator.type == MTS_INTEGRATOR_VOLPATH && !hs.integrator.spectral && !getenv("MTSAMD_WG_THREADS")))
            variant = (hs.integrator.type == MTS_INTEGRATOR_VOLPATH && !hs.integrator.spectral) ? 1 : 0;
        int wg_threads = 0;                                         // MTSAMD_WG_THREADS: threads per workgroup of the wga kernels (<= paths; default = paths)
        if (const char *tv = getenv("MTSAMD_WG_THREADS")) wg_threads = atoi(tv);
        // Lean kernels (kernels_lean_a.hip / _b.hip: the regrouping machines of rgb / mono `volpath` and `volpathmis` compiled WITHOUT what
        // this scene cannot contain, integrator_dev.h: MTS_TRAITS): the leanest unit whose promises the scene keeps.  MTSAMD_LEAN=0: never.
        // mts_stats.kernel_variant reports it as + 100000 (a) / + 200000 (b) / + 300000 (s: the spectral variant's unit) / + 400000, + 500000 (`path`: p, ps) / + 600000 (h: homogeneous media) / + 700000 (c: as b, with a BVH).
        int lean = 0;
#if !defined(MTSAMD_BLOCKSTATS)
        {
            const char *lv = getenv("MTSAMD_LEAN");
            const bool mis = hs.integrator.type == MTS_INTEGRATOR_VOLPATHMIS && hs.integrator.use_spectral_mis, vol = hs.integrator.type == MTS_INTEGRATOR_VOLPATH;
            const bool machine = hs.integrator.spectral ? (variant == 10256 && (vol || mis)) : ((variant == 11024 && vol) || (variant == 10512 && mis));
            if (!(lv && atoi(lv) == 0) && machine && !se.wavefront && wg_threads == 0) {
                auto keeps = [&](int promises) { return (hs.traits & promises) == promises; };      // dscene.h: MT_UNIT_*
                if (hs.integrator.spectral) { if (keeps(MT_UNIT_B)) lean = 3; }     // kernels_lean_s.hip
                else if (keeps(MT_UNIT_A)) lean = 1;                                 // every promise: no call left
                else if (keeps(MT_UNIT_B)) lean = 2;                                 // rpv and grids behind volume_eval() allowed
                else if (keeps(MT_UNIT_C)) lean = 7;                                 // ... and a BVH
                else if (keeps(MT_UNIT_H)) lean = 6;                                 // homogeneous media
                if (lv && atoi(lv) == 2 && lean == 1) lean = 2;                     // diagnostics: the b unit on a scene that qualifies for a
            }
            // `path` as the flat loop: kernels_lean_p.hip / _ps.hip want a walked primitive list, no spheres, no rpv
            if (!(lv && atoi(lv) == 0) && variant == 1 && hs.integrator.type == MTS_INTEGRATOR_PATH && (hs.traits & MT_UNIT_P_NEEDS) == MT_UNIT_P_NEEDS) lean = hs.integrator.spectral ? 5 : 4;
        }
#endif
        last_variant = variant + 100000 * lean;

        // one launch over `blocks` with `spp` samples per pixel, watched for cancel() / the timeout (which reach the kernel through the stop word).
        // `tiles`: the cost-sorted tile table of the regrouping kernels (volpath_flat.h, WgArgs::tiles), or empty: one workgroup per
        // run of a block's Morton order.
        auto launch = [&](const std::vector<DBlock> &blocks, uint32_t spp, const std::vector<uint32_t> &tiles) {
            DBlock *d_blocks = (DBlock *) rc.get(2, blocks.size() * sizeof(DBlock));
            HIP_CHECK(hipMemcpyAsync(d_blocks, blocks.data(), blocks.size() * sizeof(DBlock), hipMemcpyHostToDevice, stream));
            uint32_t *d_tiles = nullptr;
            if (!tiles.empty()) {
                d_tiles = (uint32_t *) rc.get(4, tiles.size() * sizeof(uint32_t));
                HIP_CHECK(hipMemcpyAsync(d_tiles, tiles.data(), tiles.size() * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
            }
            const uint64_t paths = tiles.empty() ? (uint64_t) blocks.size() * block_size * block_size : (uint64_t) tiles.size() * 16u;
            HIP_CHECK(hipEventRecord(ev0, stream));
            // one 128-byte cold record per path in flight; volpathmis parks the path's two weight matrices in a second one (volpathmis_flat.h)
            const size_t ws_records = hs.integrator.type == MTS_INTEGRATOR_VOLPATHMIS ? 2 : 1;
            float *d_ws = (float *) rc.get(3, render_workspace_floats(paths, variant) * ws_records * sizeof(float));
            auto launcher = hs.integrator.spectral ? launch_render_spectral : launch_render;
#if !defined(MTSAMD_BLOCKSTATS)
            if (lean == 1) launcher = launch_render_lean_a; else if (lean == 2) launcher = launch_render_lean_b; else if (lean == 3) launcher = launch_render_lean_s;
            else if (lean == 4) launcher = launch_render_lean_p; else if (lean == 5) launcher = launch_render_lean_ps; else if (lean == 6) launcher = launch_render_lean_h; else if (lean == 7) launcher = launch_render_lean_c;
#endif
            HIP_CHECK(launcher(
                          hs.scene, d_blocks, (uint32_t) blocks.size(), block_size, spp, d_target, d_counters,
                          opts.collect_counters != 0, variant, wg_threads, d_ws, (const uint32_t *) scene->stop_word, d_tiles, (uint32_t) tiles.size(), stream));
            HIP_CHECK(hipEventRecord(ev1, stream));
            for (;;) {
                hipError_t q = hipEventQuery(ev1);
                if (q == hipSuccess) break;
                if (q != hipErrorNotReady) HIP_CHECK(q);
                if (should_stop()) *scene->stop_word = 1;
                std::this_thread::sleep_for(std::chrono::microseconds(50));
            }
            float ms = 0.f; HIP_CHECK(hipEventElapsedTime(&ms, ev0, ev1));
            kernel_ms += ms; ++launches;
        };

        // ---- Workgroups of equal-cost pixels, expensive ones first.
        // (a) A launch with more workgroups than the GPU holds at once runs them in rounds, in array order, and the blocks of a scene
        //     differ in cost (the horizon of an atmosphere costs a multiple of its zenith): in spiral order the tail of the launch waited
        //     for whichever expensive block happened to start last (round 3: C4 278 -> 412 Msamples/s by starting expensive blocks first).
        // (b) INSIDE a spatial block the costs differ as well -- steeply where they are highest -- and a path renders ONE pixel (that is
        //     what makes the random streams those of scalar_rgb): the cheap pixels of a workgroup finish early, and for the rest of its
        //     life the workgroup's 16 waves share a fraction of its 1024 paths (measured on C4: four fifths of the paths finished at the
        //     snapshots of idle waves; waves idle 21 % of their time at 256 and at 2048 spp alike -- profiles/r04_ab_experiments.log).
        // So the regrouping kernels first render a few samples per pixel with every path adding the time at which it finished to the
        // cost of its TILE (16 Morton-consecutive pixels: a 4 x 4 square; < 0.5 % of the job, results discarded -- the film is cleared
        // again); the tiles of every launch are then sorted by descending cost and cut into workgroups: a workgroup holds pixels of
        // equal cost, whose paths finish together, and the expensive workgroups start first (longest processing time first).  Which
        // pixel receives which samples does not depend on where its path runs (the stream is seeded by block id and Morton index): same film.
        const uint32_t ppb = block_size * block_size, tiles_per_block = ppb / 16u;
        int lpt_mode = -1;
        std::vector<std::pair<uint64_t, uint32_t>> cost_index;   // (block position, index of its first tile in `tile_cost`), sorted by position
        std::vector<uint64_t> tile_cost;
        auto pos = [](const DBlock &b) { return ((uint64_t) (uint32_t) b.ox << 32) | (uint32_t) b.oy; };
        {
            int cu_count = 0;
            HIP_CHECK(hipDeviceGetAttribute(&cu_count, hipDeviceAttributeMultiprocessorCount, hs.device));
            const char *lpt = getenv("MTSAMD_LPT");                // 0: spiral order, one workgroup per run of a block's Morton order
            lpt_mode = lpt ? atoi(lpt) : -1;
            const bool force = lpt && (atoi(lpt) == 2 || atoi(lpt) == 3);   // 2, 3: calibrate whatever the block count (tests, diagnostics with MTSAMD_LPT_DEBUG)
            const uint32_t cal_spp = (uint32_t) std::max<size_t>(std::min<size_t>(4, launch_spp / 128), force && launch_spp >= 2 ? 1 : 0);
            if (variant >= 10000 && variant < 20000 && block_size <= 256 && (!lpt || atoi(lpt) != 0) && cal_spp > 0 &&
                (force || pass_blocks[0].size() > (size_t) std::max(cu_count, 1)) && !should_stop()) {
                std::vector<DBlock> cal(pass_blocks[0]);           // the distinct block positions of the first chunk
                std::sort(cal.begin(), cal.end(), [&](const DBlock &x, const DBlock &y) { return pos(x) < pos(y); });
                cal.erase(std::unique(cal.begin(), cal.end(), [&](const DBlock &x, const DBlock &y) { return pos(x) == pos(y); }), cal.end());
                for (DBlock &c : cal) c.film_off_lo = c.film_off_hi = 0;      // the calibration samples land in the first slot
                const size_t n_cost = cal.size() * tiles_per_block;
                HIP_CHECK(hipMemsetAsync(d_counters + N_COUNTERS, 0, n_cost * sizeof(unsigned long long), stream));
                const unsigned long long flag = 1ull;
                HIP_CHECK(hipMemcpyAsync(d_counters + 15, &flag, sizeof(flag), hipMemcpyHostToDevice, stream));
                launch(cal, cal_spp, {});                          // identity tiles: tile slot = block index * tiles_per_block + tile
                calibration_ms = kernel_ms; calibration_launches = launches; kernel_ms = 0.0; launches = 0;      // timed apart from the render (mts_stats)
                tile_cost.resize(n_cost);
                HIP_CHECK(hipMemcpyAsync(tile_cost.data(), d_counters + N_COUNTERS, n_cost * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
                unsigned long long diag[N_COUNTERS] = {};
                HIP_CHECK(hipMemcpyAsync(diag, d_counters, sizeof(diag), hipMemcpyDeviceToHost, stream));
                HIP_CHECK(hipStreamSynchronize(stream));
                throw_on_ring_stall(diag);                         // a bounded wait that gave up during calibration is an error like any other
                for (size_t k = 0; k < cal.size(); ++k) cost_index.emplace_back(pos(cal[k]), (uint32_t) (k * tiles_per_block));
                // A tile's measurement is 16 pixels x 1-4 samples of a heavy-tailed quantity: too noisy to sort by (a workgroup of tiles
                // with "equal" measurements would still spread by tens of per cent).  The cost of a pixel varies smoothly over the film, so
                // every tile takes the mean over the 7 x 7 tiles around it (28 x 28 pixels), on the film-wide grid of 4 x 4-pixel tiles.
                {
                    const int gw = (se.crop_w + 3) / 4, gh = (se.crop_h + 3) / 4;
                    std::vector<double> grid((size_t) gw * gh, -1.0);
                    std::vector<uint32_t> where(tile_cost.size(), 0xFFFFFFFFu);      // tile slot -> grid cell
                    for (size_t k = 0; k < cal.size(); ++k)
                        for (uint32_t t = 0; t < tiles_per_block; ++t) {
                            uint32_t x0 = 0, y0 = 0;
                            for (uint32_t bit = 0; bit < 16; ++bit) { x0 |= (((16u * t) >> (2 * bit)) & 1u) << bit; y0 |= (((16u * t) >> (2 * bit + 1)) & 1u) << bit; }
                            if ((int) x0 >= cal[k].sx || (int) y0 >= cal[k].sy) continue;
                            const int gx = (cal[k].ox - se.crop_x + (int) x0) / 4, gy = (cal[k].oy - se.crop_y + (int) y0) / 4;
                            if (gx < 0 || gy < 0 || gx >= gw || gy >= gh) continue;
                            grid[(size_t) gy * gw + gx] = (double) tile_cost[k * tiles_per_block + t];
                            where[k * tiles_per_block + t] = (uint32_t) ((size_t) gy * gw + gx);
                        }
                    // summed-area table over the cells that hold a measurement
                    std::vector<double> sat((size_t) (gw + 1) * (gh + 1), 0.0), cnt((size_t) (gw + 1) * (gh + 1), 0.0);
                    for (int y = 0; y < gh; ++y)
                        for (int x = 0; x < gw; ++x) {
                            const double v = grid[(size_t) y * gw + x];
                            const size_t i = (size_t) (y + 1) * (gw + 1) + (x + 1);
                            sat[i] = (v >= 0.0 ? v : 0.0) + sat[i - 1] + sat[i - (gw + 1)] - sat[i - (gw + 1) - 1];
                            cnt[i] = (v >= 0.0 ? 1.0 : 0.0) + cnt[i - 1] + cnt[i - (gw + 1)] - cnt[i - (gw + 1) - 1];
                        }
                    const int R = 3;
                    for (size_t sl = 0; sl < tile_cost.size(); ++sl) {
                        if (where[sl] == 0xFFFFFFFFu) continue;
                        const int x = (int) (where[sl] % (uint32_t) gw), y = (int) (where[sl] / (uint32_t) gw);
                        const int x0 = std::max(0, x - R), x1 = std::min(gw, x + R + 1), y0 = std::max(0, y - R), y1 = std::min(gh, y + R + 1);
                        auto box = [&](const std::vector<double> &a) { return a[(size_t) y1 * (gw + 1) + x1] - a[(size_t) y0 * (gw + 1) + x1] - a[(size_t) y1 * (gw + 1) + x0] + a[(size_t) y0 * (gw + 1) + x0]; };
                        const double n = box(cnt);
                        if (n > 0.0) tile_cost[sl] = (uint64_t) (box(sat) / n);
                    }
                }
                if (getenv("MTSAMD_LPT_DEBUG")) {                  // spread of the costs: between blocks, and between the tiles of a block
                    double sum = 0.0, worst_ratio = 1.0; uint64_t lo = ~0ull, hi = 0;
                    for (size_t k = 0; k < cal.size(); ++k) {
                        uint64_t bsum = 0, tlo = ~0ull, thi = 0;
                        for (uint32_t t = 0; t < tiles_per_block; ++t) { const uint64_t c = tile_cost[k * tiles_per_block + t]; bsum += c; if (c) { tlo = std::min(tlo, c); thi = std::max(thi, c); } }
                        sum += (double) bsum; lo = std::min(lo, bsum); hi = std::max(hi, bsum);
                        if (thi && tlo != ~0ull) worst_ratio = std::max(worst_ratio, (double) thi / (double) tlo);
                    }
                    size_t odd = 0;
                    for (size_t k = 0; k < tile_cost.size(); ++k) if (tile_cost[k] >> 62) { if (odd < 8) fprintf(stderr, "[mtsamd] odd tile cost %llx at tile slot %zu\n", (unsigned long long) tile_cost[k], k); ++odd; }
                    fprintf(stderr, "[mtsamd] %zu of %zu tile costs have their top bits set\n", odd, tile_cost.size());
                    fprintf(stderr, "[mtsamd] tile costs over %zu blocks x %u tiles (%u spp): block sums min %.3g mean %.3g max %.3g, max / mean %.3f; largest max / min tile cost inside one block %.2f\n",
                            cal.size(), tiles_per_block, cal_spp, (double) lo, sum / (double) cal.size(), (double) hi, (double) hi * (double) cal.size() / sum, worst_ratio);
                }
                // A cancel or the timeout that landed during calibration: the samples it rendered are the first of every pixel's stream --
                // they stay as the (partial) film, as a stopped render keeps its finished samples; otherwise they are not part of the image
                if (!should_stop()) {
                    HIP_CHECK(hipMemsetAsync(d_target, 0, film_floats * sizeof(float), stream));
                    HIP_CHECK(hipMemsetAsync(d_counters, 0, N_COUNTERS * sizeof(unsigned long long), stream));
                }
            }
        }
        // Which of the two the measured costs are used for (profiles/r04_ab_experiments.log): the kernels that run few waves per CU -- the
        // spectral variant (12 or 8) and volpathmis (8) -- gain from workgroups of equal-cost pixels (C5S +7 %, C5SM +47 %); the rgb
        // volpath kernel (16 waves per CU) does not (C4 -1.4 %: its idle waves cost nothing it could use) and keeps one workgroup per
        // spatial block, whose block record is a scalar load.  MTSAMD_LPT: 0 none, 1 whole blocks by cost, 3 tiles by cost, 2 = the
        // default policy with the calibration forced.
        const bool use_tiles = lpt_mode == 3 || (lpt_mode != 1 && (hs.integrator.spectral || hs.integrator.type == MTS_INTEGRATOR_VOLPATHMIS));
        for (size_t pass = 0; pass < pass_blocks.size(); ++pass) {
            if (should_stop()) break;
            std::vector<DBlock> &blocks = pass_blocks[pass];
            if (blocks.empty()) continue;
            std::vector<uint32_t> tiles;
            if (!cost_index.empty()) {
                // every tile of this chunk that holds a pixel, by descending cost (ties: spiral order), cut into workgroups of `wg` paths
                std::vector<std::pair<uint64_t, uint32_t>> order;
                order.reserve(blocks.size() * tiles_per_block);
                std::vector<uint64_t> bsum(blocks.size(), 0);
                for (size_t bi = 0; bi < blocks.size(); ++bi) {
                    const DBlock &bk = blocks[bi];
                    auto it = std::lower_bound(cost_index.begin(), cost_index.end(), std::make_pair(pos(bk), (uint32_t) 0));
                    const bool known = it != cost_index.end() && it->first == pos(bk);
                    for (uint32_t t = 0; t < tiles_per_block; ++t) {
                        // the tile's first pixel: Morton index 16 t -> (x, y) by de-interleaving the bits (the other fifteen lie right of / below it)
                        uint32_t x0 = 0, y0 = 0;
                        for (uint32_t bit = 0; bit < 16; ++bit) { x0 |= (((16u * t) >> (2 * bit)) & 1u) << bit; y0 |= (((16u * t) >> (2 * bit + 1)) & 1u) << bit; }
                        if ((int) x0 >= bk.sx || (int) y0 >= bk.sy) continue;                      // a partial block at the image border
                        const uint64_t c = known ? tile_cost[it->second + t] : 0ull;
                        bsum[bi] += c;
                        if (use_tiles) order.emplace_back(c, (uint32_t) ((bi << 12) | t));
                    }
                }
                if (use_tiles && blocks.size() < ((size_t) 1 << 20)) {                             // 20 bits of block index
                    std::stable_sort(order.begin(), order.end(), [](const auto &x, const auto &y) { return x.first > y.first; });
                    const size_t wg_tiles = (size_t) (variant % 10000) / 16u;
                    tiles.reserve((order.size() + wg_tiles - 1) / wg_tiles * wg_tiles);
                    for (const auto &o : order) tiles.push_back(o.second);
                    while (tiles.size() % wg_tiles) tiles.push_back(0xFFFFFFFFu);
                } else {                                                                          // whole blocks, the expensive ones first
                    std::vector<size_t> idx(blocks.size());
                    for (size_t k = 0; k < idx.size(); ++k) idx[k] = k;
                    std::stable_sort(idx.begin(), idx.end(), [&](size_t x, size_t y) { return bsum[x] > bsum[y]; });
                    std::vector<DBlock> sorted(blocks.size());
                    for (size_t k = 0; k < idx.size(); ++k) sorted[k] = blocks[idx[k]];
                    blocks.swap(sorted);
                }
            }
            launch(blocks, (uint32_t) launch_spp, tiles);
        }
        if (pass_slots) HIP_CHECK(launch_film_sum_slots(d_film, d_target, film_floats, (uint32_t) n_slots, stream));
        if (!opts.film_on_device) HIP_CHECK(hipMemcpyAsync(film, d_film, film_floats * sizeof(float), hipMemcpyDeviceToHost, stream));
        unsigned long long h_counters[N_COUNTERS] = {};
        HIP_CHECK(hipMemcpyAsync(h_counters, d_counters, sizeof(h_counters), hipMemcpyDeviceToHost, stream));
        HIP_CHECK(hipStreamSynchronize(stream));
        throw_on_ring_stall(h_counters);
        const bool cancelled = hs.stop.load() != 0;                  // render() returns !m_stop (integrator.cpp:178): a timeout alone is not a cancellation
        if (stats) {
            memset(stats, 0, sizeof(*stats));
            stats->samples = samples; stats->n_iter = h_counters[0]; stats->n_lookup = h_counters[1]; stats->n_nee_step = h_counters[2];
            stats->kernel_ms = kernel_ms; stats->kernel_launches = launches; stats->cancelled = cancelled ? 1 : 0; stats->timed_out = timed_out ? 1 : 0; stats->kernel_variant = last_variant;
            stats->calibration_ms = calibration_ms; stats->calibration_launches = calibration_launches;
            stats->wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        }
    } catch (...) { (void) hipStreamSynchronize(stream); throw; }     // nothing of this render may still be using the cached buffers
#endif
    API_CATCH
}

int mts_sample(mts_scene *scene, int32_t n, uint64_t seed_offset, const float *ox, const float *oy, const float *oz,
               const float *dx, const float *dy, const float *dz, float *out_rgb, uint8_t *out_valid) {
    API_TRY
    if (!scene || n < 0) throw std::runtime_error("mts_sample: invalid argument");
    if (n == 0) return 0;
    HostScene &hs = *scene->hs;
    if (hs.integrator.spectral) throw std::runtime_error("mts_sample: the scene was built for the spectral variant (use mts_sample_spectral: the rays carry wavelengths)");
#if defined(MTSAMD_HOST_ONLY)
    HOST_ONLY_STOP("mts_sample");
#else
    HIP_CHECK(hipSetDevice(hs.device));
    DeviceBuffer<float> d_rays((size_t) 6 * n), d_rgb((size_t) 3 * n);
    DeviceBuffer<uint8_t> d_valid((size_t) n);
    const float *rows[6] = { ox, oy, oz, dx, dy, dz };
    for (int r = 0; r < 6; ++r) HIP_CHECK(hipMemcpy(d_rays.p + (size_t) r * n, rows[r], (size_t) n * sizeof(float), hipMemcpyHostToDevice));
    HIP_CHECK(launch_sample(hs.scene, n, seed_offset, d_rays.p, d_rgb.p, d_valid.p, nullptr));
    HIP_CHECK(hipMemcpy(out_rgb, d_rgb.p, (size_t) 3 * n * sizeof(float), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(out_valid, d_valid.p, (size_t) n, hipMemcpyDeviceToHost));
#endif
    API_CATCH
}

int mts_sample_spectral(mts_scene *scene, int32_t n, uint64_t seed_offset, const float *ox, const float *oy, const float *oz,
                        const float *dx, const float *dy, const float *dz, const float *wavelengths, float *out_spec, uint8_t *out_valid) {
    API_TRY
    if (!scene || n < 0) throw std::runtime_error("mts_sample_spectral: invalid argument");
    if (n == 0) return 0;
    if (!ox || !oy || !oz || !dx || !dy || !dz || !wavelengths || !out_spec || !out_valid) throw std::runtime_error("mts_sample_spectral: null array");
    HostScene &hs = *scene->hs;
    if (!hs.integrator.spectral) throw std::runtime_error("mts_sample_spectral: the scene was built for an rgb / mono variant (use mts_sample)");
#if defined(MTSAMD_HOST_ONLY)
    HOST_ONLY_STOP("mts_sample_spectral");
#else
    HIP_CHECK(hipSetDevice(hs.device));
    DeviceBuffer<float> d_rays((size_t) 6 * n), d_wl((size_t) 4 * n), d_spec((size_t) 4 * n);
    DeviceBuffer<uint8_t> d_valid((size_t) n);
    const float *rows[6] = { ox, oy, oz, dx, dy, dz };
    for (int r = 0; r < 6; ++r) HIP_CHECK(hipMemcpy(d_rays.p + (size_t) r * n, rows[r], (size_t) n * sizeof(float), hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d_wl.p, wavelengths, (size_t) 4 * n * sizeof(float), hipMemcpyHostToDevice));
    HIP_CHECK(launch_sample_spectral(hs.scene, n, seed_offset, d_rays.p, d_wl.p, d_spec.p, d_valid.p, nullptr));
    HIP_CHECK(hipMemcpy(out_spec, d_spec.p, (size_t) 4 * n * sizeof(float), hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(out_valid, d_valid.p, (size_t) n, hipMemcpyDeviceToHost));
#endif
    API_CATCH
}

int mts_sample_tea(int device, int32_t n, const uint32_t *v0, const uint32_t *v1, int32_t rounds, uint32_t *out32, uint64_t *out64, float *out_float32) {
    API_TRY
    if (n < 0 || !v0 || !v1 || !out32 || !out64 || !out_float32) throw std::runtime_error("mts_sample_tea: invalid argument");
    if (n == 0) return 0;
#if defined(MTSAMD_HOST_ONLY)
    HOST_ONLY_STOP("mts_sample_tea");
#else
    HIP_CHECK(hipSetDevice(device));
    DeviceBuffer<uint32_t> d0(n), d1(n), o32(n); DeviceBuffer<uint64_t> o64(n); DeviceBuffer<float> of(n);
    HIP_CHECK(hipMemcpy(d0.p, v0, (size_t) n * 4, hipMemcpyHostToDevice)); HIP_CHECK(hipMemcpy(d1.p, v1, (size_t) n * 4, hipMemcpyHostToDevice));
    HIP_CHECK(launch_tea(n, d0.p, d1.p, rounds, o32.p, o64.p, of.p, nullptr));
    HIP_CHECK(hipMemcpy(out32, o32.p, (size_t) n * 4, hipMemcpyDeviceToHost)); HIP_CHECK(hipMemcpy(out64, o64.p, (size_t) n * 8, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(out_float32, of.p, (size_t) n * 4, hipMemcpyDeviceToHost));
#endif
    API_CATCH
}

int mts_wavefront_sampler(int device, int32_t lanes, uint64_t seed_value, int32_t count, float *out) {
    API_TRY
    if (lanes < 0 || count < 0 || !out) throw std::runtime_error("mts_wavefront_sampler: invalid argument");
    if (lanes == 0 || count == 0) return 0;
#if defined(MTSAMD_HOST_ONLY)
    HOST_ONLY_STOP("mts_wavefront_sampler");
#else
    HIP_CHECK(hipSetDevice(device));
    DeviceBuffer<float> d((size_t) lanes * count);
    HIP_CHECK(launch_wavefront_sampler(lanes, seed_value, count, d.p, nullptr));
    HIP_CHECK(hipMemcpy(out, d.p, (size_t) lanes * count * 4, hipMemcpyDeviceToHost));
#endif
    API_CATCH
}

int mts_ray_intersect(mts_scene *scene, int32_t n, const float *o, const float *d, const float *mint, const float *maxt,
                      float *out_t, int32_t *out_shape, int32_t *out_prim, float *out_p, float *out_n) {
    API_TRY
    if (!scene || n < 0) throw std::runtime_error("mts_ray_intersect: invalid argument");
    if (n == 0) return 0;
    HostScene &hs = *scene->hs; (void) hs;
#if defined(MTSAMD_HOST_ONLY)
    HOST_ONLY_STOP("mts_ray_intersect");
#else
    HIP_CHECK(hipSetDevice(hs.device));
    DeviceBuffer<float> d_o((size_t) 3 * n), d_d((size_t) 3 * n), d_mint(n), d_maxt(n), d_t(n), d_p((size_t) 3 * n), d_n((size_t) 3 * n);
    DeviceBuffer<int32_t> d_shape(n), d_prim(n);
    HIP_CHECK(hipMemcpy(d_o.p, o, (size_t) 3 * n * 4, hipMemcpyHostToDevice)); HIP_CHECK(hipMemcpy(d_d.p, d, (size_t) 3 * n * 4, hipMemcpyHostToDevice));
    HIP_CHECK(hipMemcpy(d_mint.p, mint, (size_t) n * 4, hipMemcpyHostToDevice)); HIP_CHECK(hipMemcpy(d_maxt.p, maxt, (size_t) n * 4, hipMemcpyHostToDevice));
    HIP_CHECK(launch_intersect(hs.scene, n, d_o.p, d_d.p, d_mint.p, d_maxt.p, d_t.p, d_shape.p, d_prim.p, d_p.p, d_n.p, nullptr));
    HIP_CHECK(hipMemcpy(out_t, d_t.p, (size_t) n * 4, hipMemcpyDeviceToHost)); HIP_CHECK(hipMemcpy(out_shape, d_shape.p, (size_t) n * 4, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(out_prim, d_prim.p, (size_t) n * 4, hipMemcpyDeviceToHost));
    HIP_CHECK(hipMemcpy(out_p, d_p.p, (size_t) 3 * n * 4, hipMemcpyDeviceToHost)); HIP_CHECK(hipMemcpy(out_n, d_n.p, (size_t) 3 * n * 4, hipMemcpyDeviceToHost));
#endif
    API_CATCH
}

} // extern "C"
