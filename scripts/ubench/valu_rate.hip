// VALU issue-rate microbenchmark for gfx950: cycles per wave64 instruction per SIMD for fma / mul / exp2 / rcp / cvt_pk.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(seed));
            if (OP == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
            if (OP == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            if (OP == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (OP == 4) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
            if (OP == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*(double*)&a[i & ~1]) : "v"(*(double*)&a[i & ~1]));
            if (OP == 6) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
            if (OP == 7) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
            if (OP == 8) asm volatile("v_exp_f16 %0, %0" : "+v"(a[i]));
            if (OP == 9) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(seed));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
    if (s == 12345.678f) out[0] = s;
}
template <int OP> int run(const char* name, float* d, int waves_per_simd) {
    const int iters = 20000, blocks = 256 * waves_per_simd;     // 256-thread blocks = 4 waves: one per SIMD
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, 100, 1.0001f);
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f);
    CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double inst_per_simd = (double)iters * 8 * waves_per_simd;
    printf("%-22s waves/SIMD %d: %.3f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz)\n", name, waves_per_simd, ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
    return 0;
}
int main() {
    float* d; CHECK(hipMalloc(&d, 1024));
    for (int w : {1, 2, 4}) {
        run<0>("v_fma_f32", d, w); run<1>("v_mul_f32", d, w); run<9>("v_add_f32", d, w); run<6>("v_max_f32", d, w); run<2>("v_exp_f32", d, w); run<3>("v_rcp_f32", d, w);
        run<4>("v_cvt_pk_bf16_f32", d, w); run<5>("v_pk_fma_f32 (2 elem)", d, w); run<7>("v_pk_mul_f16", d, w); run<8>("v_exp_f16", d, w);
    }
    return 0;
}
