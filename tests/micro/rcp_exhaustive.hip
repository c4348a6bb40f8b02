// csrc/pmath.h: pm_rcp on the device is one Newton step on v_rcp_f32 followed by v_div_fixup_f32 (four instructions).  Is that 1.0f / x as
// the kernels' build mode expands it (v_div_scale x 2, v_rcp, two denormal-mode switches, five fma, v_div_fmas, v_div_fixup) -- for EVERY
// fp32 x?  All 2^32 bit patterns; exit status 1 on any difference.  Also: the latency of both (dependent chains, s_memtime).
//     hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fgpu-flush-denormals-to-zero tests/micro/rcp_exhaustive.hip -o /tmp/rcp_ex && /tmp/rcp_ex
// Measured (MI355X): 0 differences; 35 against 92 cycles per reciprocal + multiply-add at one and at four waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include "../../eradiate-kernel_amd/csrc/pmath.h"
__device__ __forceinline__ float rcp_one_step(float x) { return pm_rcp(x); }
__device__ __forceinline__ float rcp_two_steps(float x) {            // a second step changes nothing either
    const float y0 = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, y0, 1.0f);
    float y = __builtin_fmaf(e, y0, y0);
    e = __builtin_fmaf(-x, y, 1.0f);
    y = __builtin_fmaf(e, y, y);
    return __builtin_amdgcn_div_fixupf(y, x, 1.0f);
}
__global__ void sweep(unsigned long long *bad, uint32_t *first) {
    const uint64_t i0 = ((uint64_t) blockIdx.x * blockDim.x + threadIdx.x) * 256u;
    for (uint32_t k = 0; k < 256u; ++k) {
        const uint32_t bits = (uint32_t) (i0 + k);
        const float x = __uint_as_float(bits);
        const uint32_t q = __float_as_uint(1.0f / x), a = __float_as_uint(rcp_one_step(x)), b = __float_as_uint(rcp_two_steps(x));
        const bool qn = (q & 0x7fffffffu) > 0x7f800000u;
        if (a != q && !(qn && (a & 0x7fffffffu) > 0x7f800000u)) { const unsigned long long n = atomicAdd(bad + 0, 1ull); if (n < 8) first[n] = bits; }
        if (b != q && !(qn && (b & 0x7fffffffu) > 0x7f800000u)) { const unsigned long long n = atomicAdd(bad + 1, 1ull); if (n < 8) first[8 + n] = bits; }
    }
}
template <int MODE> __global__ void chain(long long *out, float *sink, int iters, float seed) {
    float a[4]; for (int i = 0; i < 4; ++i) a[i] = seed + 0.01f * i + 1e-4f * threadIdx.x;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = MODE == 0 ? 1.0f / a[i] : rcp_one_step(a[i]); a[i] = a[i] * 0.75f + 0.5f; asm volatile("" : "+v"(a[i])); }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a[0] + a[1] + a[2] + a[3];
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}
int main() {
    unsigned long long *bad, h[2]; uint32_t *first, hf[16];
    hipMalloc(&bad, 16); hipMalloc(&first, 64); hipMemset(bad, 0, 16); hipMemset(first, 0, 64);
    hipLaunchKernelGGL(sweep, dim3(65536), dim3(256), 0, 0, bad, first);
    hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost); hipMemcpy(hf, first, 64, hipMemcpyDeviceToHost);
    std::printf("pm_rcp (one Newton step + div_fixup) against 1.0f / x on 2^32 arguments: %llu differences", h[0]);
    for (unsigned long long k = 0; k < (h[0] < 8 ? h[0] : 8); ++k) std::printf(" %08x", hf[k]);
    std::printf("\ntwo Newton steps + div_fixup: %llu differences", h[1]);
    for (unsigned long long k = 0; k < (h[1] < 8 ? h[1] : 8); ++k) std::printf(" %08x", hf[8 + k]);
    std::printf("\n");
    long long *d; float *sink; hipMalloc(&d, 2048 * 8); hipMalloc(&sink, 256 * 1024 * 4);
    for (int mode = 0; mode < 2; ++mode) for (int wps : { 1, 4 }) {
        for (int rep = 0; rep < 2; ++rep) { if (mode == 0) hipLaunchKernelGGL(chain<0>, dim3(256), dim3(256 * wps), 0, 0, d, sink, 2000, 1.0001f); else hipLaunchKernelGGL(chain<1>, dim3(256), dim3(256 * wps), 0, 0, d, sink, 2000, 1.0001f); }
        hipDeviceSynchronize();
        long long hh[256]; hipMemcpy(hh, d, sizeof(hh), hipMemcpyDeviceToHost);
        double sum = 0; for (long long v : hh) sum += (double) v;
        std::printf("%-28s %d wave(s) per SIMD: %6.1f cycles per reciprocal + multiply-add (wave's view)\n", mode ? "pm_rcp" : "1.0f / x", wps, sum / 256 / 8000.0);
    }
    return h[0] != 0;
}
