// MFMA issue-rate microbenchmark on gfx950: 16x16x32 vs 32x32x16 bf16, 1 / 2 waves per SIMD, independent accumulators.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int SHAPE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (short)(0x3f80 + threadIdx.x + i); b[i] = (short)(0x3f00 + i); }
    if (SHAPE == 16) {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        float s = 0; for (int i = 0; i < 16; ++i) s += acc[i][0];
        if (s == 1.2345f) out[0] = s;
    } else {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
        float s = 0; for (int i = 0; i < 4; ++i) s += acc[i][0];
        if (s == 1.2345f) out[0] = s;
    }
}
template <int SHAPE> int run(float* d, int wps) {
    const int iters = 4000, blocks = 256 * wps;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, d, 50);
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k<SHAPE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double mfma_per_simd = (double)iters * (SHAPE == 16 ? 16 : 8) * wps;
    const double flop = mfma_per_simd * 1024 * (SHAPE == 16 ? 16384.0 : 32768.0);
    printf("%s  waves/SIMD %d: %.2f ns per MFMA per SIMD, %.0f TFLOP/s chip\n", SHAPE == 16 ? "16x16x32" : "32x32x16", wps, ms * 1e6 / mfma_per_simd, flop / (ms * 1e-3) / 1e12);
    return 0;
}
int main() {
    float* d; CHECK(hipMalloc(&d, 1024));
    for (int w : {1, 2, 4}) { run<16>(d, w); run<32>(d, w); }
    return 0;
}
