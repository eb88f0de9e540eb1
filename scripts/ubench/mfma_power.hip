// mfma_power.hip -- what the matrix pipes SUSTAIN on this chip, by operand data (round 6).
//
// One 256-thread workgroup per CU (one wave per SIMD), every wave issues back-to-back v_mfma_f32_32x32x16_bf16 on 12 independent
// accumulators from operands held in registers -- no memory, no LDS, no other instruction in the loop -- for ~100 us.  Operands:
// zeros | a constant | random normal bf16 values that ROTATE through a ring of 8 register sets (new A and B bits every instruction,
// as in a conv's K loop).  Reported per case: TFLOP/s, the shader clock (s_memtime cycles per s_memrealtime tick of 10 ns) and the
// cycles per MFMA: the pipe always issues at 32 cycles per instruction -- what changes with the data is the CLOCK the power
// management allows.  This is the ceiling a 3x3 conv's K loop can reach on the same kind of data.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o scratch/ubench/mfma_power scripts/ubench/mfma_power.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <random>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short bf16x8;

template <int NACC>
__global__ __launch_bounds__(256, 1) void mfma_loop(const bf16x8* __restrict__ ops, float* out, unsigned long long* stamps, int iters, int nsets) {
    const int lane = threadIdx.x & 63;
    bf16x8 a[8], b[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        a[s] = ops[((s % nsets) * 2 + 0) * 64 + lane];
        b[s] = ops[((s % nsets) * 2 + 1) * 64 + lane];
    }
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    unsigned long long t0, r0, t1, r1;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(s + i) & 7], b[(s + 3 * i) & 7], acc[i], 0, 0, 0);
    }
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) sum += acc[i][lane & 15];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
}

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); }

int main(int argc, char** argv) {
    const int cus = argc > 1 ? atoi(argv[1]) : 256;
    const int iters = argc > 2 ? atoi(argv[2]) : 60;
    constexpr int NACC = 12;
    std::vector<unsigned short> h(8 * 2 * 64 * 8);
    bf16x8* d_ops; float* d_out; unsigned long long* d_st;
    hipMalloc(&d_ops, h.size() * 2); hipMalloc(&d_out, 256 * 256 * 4); hipMalloc(&d_st, 256 * 16);
    std::mt19937 rng(1);
    std::normal_distribution<float> nd(0.f, 1.f);
    const char* names[] = {"zeros", "constant 1.0 / 0.01", "random normal, one operand set", "random normal, 8 rotating operand sets", "random normal x small weights, 8 sets"};
    for (int mode = 0; mode < 5; ++mode) {
        for (size_t i = 0; i < h.size(); ++i) {
            const bool is_b = (i / (64 * 8)) & 1;
            float v = 0.f;
            if (mode == 1) v = is_b ? 0.01f : 1.0f;
            if (mode >= 2) v = nd(rng) * (mode == 4 && !is_b ? 0.02f : 1.f);
            h[i] = f2bf(v);
        }
        hipMemcpy(d_ops, h.data(), h.size() * 2, hipMemcpyHostToDevice);
        const int nsets = mode >= 3 ? 8 : 1;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(mfma_loop<NACC>, dim3(cus), dim3(256), 0, 0, d_ops, d_out, d_st, iters, nsets);
        hipEventRecord(e0);
        const int reps = 20;
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(mfma_loop<NACC>, dim3(cus), dim3(256), 0, 0, d_ops, d_out, d_st, iters, nsets);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long st[512]; hipMemcpy(st, d_st, sizeof(st), hipMemcpyDeviceToHost);
        double cyc = 0, rt = 0;
        for (int b = 0; b < cus && b < 256; ++b) { cyc += (double)st[2 * b]; rt += (double)st[2 * b + 1]; }
        const double mfmas = (double)iters * 8 * NACC;
        const double flops = (double)cus * 4 * mfmas * 32 * 32 * 16 * 2 * reps;
        printf("%-44s %4d CUs: %7.1f us per launch  %7.0f TFLOP/s  clock %.3f GHz  %.2f cycles per MFMA\n", names[mode], cus, ms * 1e3 / reps, flops / (ms * 1e-3) / 1e12,
               cyc / rt * 0.1, cyc / cus / mfmas);
    }
    return 0;
}
