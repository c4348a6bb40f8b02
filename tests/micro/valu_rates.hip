// Issue cost of the vector instructions csrc/pmath.h's fp64 routines are made of, on gfx950: cycles per wave-instruction for one wave
// alone on a SIMD and for four waves per SIMD (the occupancy of the render kernel).  Eight independent chains per lane, s_memtime around.
//     hipcc --offload-arch=gfx950 -O3 tests/micro/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define REP 64
#define BODY(INSTR) \
    for (int it = 0; it < iters; ++it) { \
        _Pragma("unroll") for (int r = 0; r < REP / 8; ++r) { INSTR(0) INSTR(1) INSTR(2) INSTR(3) INSTR(4) INSTR(5) INSTR(6) INSTR(7) } }
#define K(NAME, DECL, INSTR, SINK) \
__global__ void NAME(long long *out, int iters, double seed) { DECL \
    long long t0 = __builtin_amdgcn_s_memtime(); BODY(INSTR) long long t1 = __builtin_amdgcn_s_memtime(); \
    SINK if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0; }
#define DD double a[8], b = seed, c = seed * 0.5; for (int i = 0; i < 8; ++i) a[i] = seed + i;
#define FD float a[8], b = (float) seed, c = (float) seed * 0.5f; for (int i = 0; i < 8; ++i) a[i] = (float) seed + i;
#define SD double s = 0; for (int i = 0; i < 8; ++i) s += a[i]; if (s == 12345.678) out[1000] = 1;
#define SF float s = 0; for (int i = 0; i < 8; ++i) s += a[i]; if (s == 12345.678f) out[1000] = 1;
#define I_FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
#define I_MUL64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define I_ADD64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
#define I_FMA32(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
K(k_fma64, DD, I_FMA64, SD) K(k_mul64, DD, I_MUL64, SD) K(k_add64, DD, I_ADD64, SD) K(k_fma32, FD, I_FMA32, SF)
// conversions: chains through a float and a double register
#define CD double a[8]; float f[8]; int n[8]; for (int i = 0; i < 8; ++i) { a[i] = seed + i; f[i] = (float) seed + i; n[i] = (int) seed + i; }
#define SC double s = 0; for (int i = 0; i < 8; ++i) s += a[i] + f[i] + n[i]; if (s == 12345.678) out[1000] = 1;
#define I_CVT_F64_F32(i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(f[i]));
#define I_CVT_F32_F64(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(a[i]));
#define I_CVT_F64_I32(i) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a[i]) : "v"(n[i]));
#define I_RCP32(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(f[i]));
#define I_LSHL64(i) asm volatile("v_lshlrev_b64 %0, 3, %0" : "+v"(a[i]));
#define I_MUL_U32(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 7]));
#define I_MULHI_U32(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 7]));
K(k_cvt_f64_f32, CD, I_CVT_F64_F32, SC) K(k_cvt_f32_f64, CD, I_CVT_F32_F64, SC) K(k_cvt_f64_i32, CD, I_CVT_F64_I32, SC) K(k_rcp32, CD, I_RCP32, SC)
K(k_lshl64, CD, I_LSHL64, SC) K(k_mul_u32, CD, I_MUL_U32, SC) K(k_mulhi_u32, CD, I_MULHI_U32, SC)

template <class F> static void run(const char *name, F kern) {
    long long *d; hipMalloc(&d, 2048 * sizeof(long long));
    const int iters = 2000;
    for (int waves_per_simd : { 1, 4 }) {
        const int threads = 64 * 4 * waves_per_simd;                 // one workgroup per CU, 4 SIMDs
        hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, d, iters, 1.0001);
        hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, d, iters, 1.0001);
        hipDeviceSynchronize();
        std::vector<long long> h(256); hipMemcpy(h.data(), d, 256 * sizeof(long long), hipMemcpyDeviceToHost);
        double sum = 0; for (long long v : h) sum += (double) v;
        const double per_wave_instr = sum / 256 / ((double) iters * REP);
        std::printf("%-14s %d wave(s) per SIMD: %6.2f cycles per wave-instruction (wave's view), %6.2f cycles of SIMD time per instruction\n", name, waves_per_simd, per_wave_instr, per_wave_instr / waves_per_simd);
    }
    hipFree(d);
}
int main() {
    run("v_fma_f32", k_fma32); run("v_fma_f64", k_fma64); run("v_mul_f64", k_mul64); run("v_add_f64", k_add64);
    run("v_cvt_f64_f32", k_cvt_f64_f32); run("v_cvt_f32_f64", k_cvt_f32_f64); run("v_cvt_f64_i32", k_cvt_f64_i32); run("v_rcp_f32", k_rcp32);
    run("v_lshlrev_b64", k_lshl64); run("v_mul_lo_u32", k_mul_u32); run("v_mul_hi_u32", k_mulhi_u32);
    return 0;
}
