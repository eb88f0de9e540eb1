// L2 -> LDS delivery-rate microbenchmark on gfx950: buffer_load_dwordx4 ... lds in the conv kernel's staging pattern
// (8 lanes per 128-byte row piece, rows `rowstride` bytes apart), no consumers.  Varies: loads in flight per wave,
// row stride, footprint per XCD, workgroups per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rs, char* dst, int voff, int soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ u32x4 ld16(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    return __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk(const void* p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000); }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// L loads per wave per stage (each 64 lanes x 16 B = 1 KB), D stages in flight
template <int L, int D, int MODE>
__global__ __launch_bounds__(512) void k(const char* buf, unsigned bytes, int iters, int rowstride, int tile_bytes, int ntiles, int kwrap, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6;
    const int b = blockIdx.x, per = ntiles >> 3;
    const int t = (b & 7) * per + ((b >> 3) % per);
    const int row = tid >> 3, q = tid & 7;
    int voff[L];
#pragma unroll
    for (int j = 0; j < L; ++j) voff[j] = t * tile_bytes + (row + 64 * j) * rowstride + q * 16;
    const __amdgpu_buffer_rsrc_t rs = mk(buf, bytes);
    const int wave_lds = __builtin_amdgcn_readfirstlane(wave * 1024);
    u32x4 sink = {0, 0, 0, 0};
    int kk = 0, tap = 0;
    for (int it = 0; it < iters; ++it) {
        const int soff = kk * 128 + tap * rowstride;
        if (++kk == kwrap) { kk = 0; if (++tap == 9) tap = 0; }
#pragma unroll
        for (int j = 0; j < L; ++j) {
            if (MODE == 0)
                dma16(rs, smem + wave_lds + j * 8192, voff[j], soff);
            else {
                u32x4 v = ld16(rs, voff[j], soff);
                sink ^= v;
            }
        }
        if (MODE == 0) wait_vm<L * (D - 1)>();
    }
    wait_vm<0>();
    if (sink[0] == 0x12345u && sink[1] == 7u) out[0] = 1.f;
}

template <int L, int D, int MODE>
int run(const char* buf, unsigned bytes, float* out, int wgs_per_cu, int rowstride, int tile_rows, int ntiles, int lds) {
    const int iters = 2000, blocks = 256 * wgs_per_cu;
    const int tile_bytes = tile_rows * rowstride, kwrap = rowstride / 128;
    if ((long)ntiles * tile_bytes + (64L * L + 16) * rowstride > (long)bytes) { printf("buffer too small\n"); return 1; }
    auto kern = k<L, D, MODE>;
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, 0, buf, bytes, 50, rowstride, tile_bytes, ntiles, kwrap, out);
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, 0, buf, bytes, iters, rowstride, tile_bytes, ntiles, kwrap, out);
    CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double total = (double)blocks * iters * L * 8 * 1024;
    printf("%s L=%d D=%d wg/CU=%d stride=%4d tiles=%4d (%.1f MB/XCD): %.2f TB/s chip, %.1f GB/s/CU (%.1f B/clk @2.4GHz), in flight/CU %.0f KB\n",
           MODE ? "vgpr" : "lds ", L, D, wgs_per_cu, rowstride, ntiles, ntiles / 8.0 * tile_bytes / 1e6, total / (ms * 1e-3) / 1e12,
           total / (ms * 1e-3) / 256 / 1e9, total / (ms * 1e-3) / 256 / 2.4e9, (double)wgs_per_cu * L * 8 * (MODE ? 1 : D - 1 + 1));
    return 0;
}

int main() {
    const unsigned bytes = 1u << 30;
    char* buf; float* out;
    CHECK(hipMalloc(&buf, bytes)); CHECK(hipMemset(buf, 1, bytes)); CHECK(hipMalloc(&out, 1024));
    const int big = 120 * 1024, small = 60 * 1024;
    // footprint sweep at the conv's shape (7 loads/stage, 2 stages, 1 wg/CU, 512-byte rows, 192-row tiles)
    for (int nt : {8, 64, 256, 512, 2048}) run<7, 2, 0>(buf, bytes, out, 1, 512, 192, nt, big);
    // depth sweep
    run<7, 1, 0>(buf, bytes, out, 1, 512, 192, 256, big);
    run<7, 3, 0>(buf, bytes, out, 1, 512, 192, 256, big);
    run<7, 4, 0>(buf, bytes, out, 1, 512, 192, 256, big);
    run<7, 8, 0>(buf, bytes, out, 1, 512, 192, 256, big);
    run<4, 2, 0>(buf, bytes, out, 2, 512, 192, 256, small);
    run<4, 4, 0>(buf, bytes, out, 2, 512, 192, 256, small);
    run<7, 4, 0>(buf, bytes, out, 2, 512, 192, 256, small);
    // row stride sweep (128 = fully linear)
    for (int rs : {128, 256, 1024, 2048}) run<7, 4, 0>(buf, bytes, out, 1, rs, 192, 256, big);
    // to VGPRs instead of LDS
    run<7, 1, 1>(buf, bytes, out, 1, 512, 192, 256, big);
    run<7, 1, 1>(buf, bytes, out, 2, 512, 192, 256, small);
    run<7, 1, 1>(buf, bytes, out, 1, 128, 192, 256, big);
    return 0;
}
