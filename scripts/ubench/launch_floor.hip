// launch_floor.hip -- what a dependent chain of tiny kernels costs per launch on this GPU / runtime (LABNOTES.md section 4.7):
//   empty:      no arguments read, nothing done
//   args:       a 512-byte by-value struct, one field of its LAST 64 bytes read (the kernarg fetch: one scalar-cache miss per launch)
//   args+load:  the same, then one dependent global load through a pointer from the struct and one store (two serial round trips,
//               like a conv kernel's "arguments -> first tile")
// 256 workgroups of 256 threads each, 2000 launches back to back on one stream, time per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
struct Big { const float* src; float* dst; int pad[120]; int last[4]; };
__global__ void k_empty() {}
__global__ void k_args(const Big b) { if (b.last[3] == 12345 && threadIdx.x == 0) b.dst[0] = 1.f; }
__global__ void k_args_load(const Big b) {
    const float v = b.src[(blockIdx.x * 256 + threadIdx.x) * 16 + (b.last[3] & 1)];
    if (v == 123.f) b.dst[blockIdx.x] = v;
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <class F> static double per_launch_us(F launch, hipStream_t s, int n) {
    for (int i = 0; i < 50; ++i) launch();
    (void)hipStreamSynchronize(s);
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < n; ++i) launch();
    (void)hipStreamSynchronize(s);
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
}
int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    float *src, *dst; CK(hipMalloc(&src, 256 * 256 * 16 * 4 + 64)); CK(hipMalloc(&dst, 4096)); CK(hipMemset(src, 0, 256 * 256 * 16 * 4 + 64));
    Big b{}; b.src = src; b.dst = dst;
    for (int wg : {1, 256}) {
        printf("%3d workgroups: empty %.2f us/launch, 512-byte args %.2f, args + dependent load %.2f\n", wg,
               per_launch_us([&] { hipLaunchKernelGGL(k_empty, dim3(wg), dim3(256), 0, s); }, s, 2000),
               per_launch_us([&] { hipLaunchKernelGGL(k_args, dim3(wg), dim3(256), 0, s, b); }, s, 2000),
               per_launch_us([&] { hipLaunchKernelGGL(k_args_load, dim3(wg), dim3(256), 0, s, b); }, s, 2000));
    }
    return 0;
}
