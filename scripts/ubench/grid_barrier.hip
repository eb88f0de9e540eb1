// grid_barrier.hip -- what ONE persistent kernel pays per layer boundary, against what a launch pays (DESIGN.md section 8, item 3:
// the "persistent multi-layer kernel" for one image).  256 workgroups of 256 threads (one per CU, co-resident) run PHASES phases;
// in every phase a workgroup writes `bytes` bytes of a slab (the "layer output"), all workgroups meet at a grid barrier (agent-scope
// release: the XCD's L2 writes its dirty lines back; one atomic per workgroup on one counter; spin; agent-scope acquire: the L2
// invalidates), and then every workgroup reads the slab of a workgroup of ANOTHER XCD and checks it (the "next layer's input").
//   barrier only     bytes = 0
//   + 4 KB / 16 KB / 64 KB per workgroup   (1 / 4 / 16 MB per phase over the chip: a 19^2 / 38^2 / 76^2 layer of one image is 0.2 - 1.5 MB)
// Next to it: the same phases as separate launches on one stream (the kernel boundary does the release / acquire).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int G = 256, T = 256;

// KIND 0: every workgroup adds to ONE counter and polls it.  KIND 1: the last arriver (the add returns target - 1) publishes the
// phase number in a flag on another cache line; the others poll the flag, not the counter the adds go to.  KIND 2: two levels --
// the 32 workgroups of an XCD (block b -> XCD b % 8) meet on their XCD's counter, the eight last arrivers on a global one.
// KIND 3: KIND 1 WITHOUT the release / acquire fences -- not a usable barrier (the check below counts the stale reads), it prices them.
// KIND 4: only the 32 workgroups of XCD 0 (blocks b % 8 == 0; the others leave at once) -- they share one L2, so the release is
// workgroup scope (wait for the stores: no L2 write-back) and the acquire an agent-scope one (the CU's L1 is invalidated).
// KIND 5: KIND 4's fences with all 256 workgroups: prices the write-back alone (stale reads across XCDs expected).
// KIND 6: KIND 4 without the acquire: the consumer reads the slab with L1-bypassing loads instead (agent-scope relaxed atomic loads).
template <int KIND>
__device__ __forceinline__ void grid_barrier(unsigned* sync, unsigned phase1, unsigned np = G) {       // phase1 = phase + 1, np = participants
    __syncthreads();
    if (threadIdx.x == 0) {
        if (KIND >= 4) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");     // the stores have left the CU (its L1 writes through)
        else if (KIND != 3) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        unsigned* flag = sync + 64;
        if (KIND == 0) {
            __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase1 * G) __builtin_amdgcn_s_sleep(1);
        } else if (KIND == 1 || KIND >= 3) {
            if (__hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == phase1 * np - 1)
                __hip_atomic_store(flag, phase1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else
                while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase1) __builtin_amdgcn_s_sleep(1);
        } else {
            unsigned* mine = sync + 128 + 64 * (blockIdx.x & 7);
            bool last = false;
            if (__hip_atomic_fetch_add(mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == phase1 * (G / 8) - 1)
                last = __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == phase1 * 8 - 1;
            if (last) __hip_atomic_store(flag, phase1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase1) __builtin_amdgcn_s_sleep(1);
        }
        if (KIND != 3 && KIND != 6) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    if (KIND != 3 && KIND != 6) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

__device__ __forceinline__ void phase_write(uint4* slab, int n16, unsigned tag) {
    for (int i = threadIdx.x; i < n16; i += T) slab[i] = make_uint4(tag, (unsigned)i, blockIdx.x, tag ^ 0x5a5a5a5au);
}
__device__ __forceinline__ unsigned phase_check(const uint4* slab, int n16, unsigned tag, unsigned owner) {
    unsigned bad = 0;
    for (int i = threadIdx.x; i < n16; i += T) {
        const uint4 v = slab[i];
        bad += (v.x != tag) | (v.y != (unsigned)i) | (v.z != owner) | (v.w != (tag ^ 0x5a5a5a5au));
    }
    return bad;
}

__device__ __forceinline__ unsigned phase_check_l1_bypass(const uint4* slab, int n16, unsigned tag, unsigned owner);
template <int KIND>
__global__ __launch_bounds__(T) void persistent(uint4* buf, int n16, int phases, unsigned* counter, unsigned* bad_out) {
    unsigned bad = 0;
    constexpr bool ONE_XCD = KIND == 4 || KIND == 6;
    if (ONE_XCD && (blockIdx.x & 7) != 0) return;
    const unsigned step = ONE_XCD ? 8 : 1;
    for (int p = 0; p < phases; ++p) {
        uint4* mine = buf + ((size_t)(p & 1) * G + blockIdx.x) * n16;
        phase_write(mine, n16, (unsigned)p + 1);
        grid_barrier<KIND>(counter, (unsigned)p + 1, G / step);
        const unsigned other = (blockIdx.x + step) % G;                // block b runs on XCD b % 8: the neighbour is another XCD's (KIND 4: another CU's)
        const uint4* theirs = buf + ((size_t)(p & 1) * G + other) * n16;
        bad += KIND == 6 ? phase_check_l1_bypass(theirs, n16, (unsigned)p + 1, other) : phase_check(theirs, n16, (unsigned)p + 1, other);
    }
    if (bad) atomicAdd(bad_out, bad);
}

__device__ __forceinline__ unsigned phase_check_l1_bypass(const uint4* slab, int n16, unsigned tag, unsigned owner) {
    unsigned bad = 0;
    const unsigned* w = (const unsigned*)slab;
    for (int i = threadIdx.x; i < n16; i += T) {
        const unsigned x = __hip_atomic_load(w + 4 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), y = __hip_atomic_load(w + 4 * i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned z = __hip_atomic_load(w + 4 * i + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), q = __hip_atomic_load(w + 4 * i + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bad += (x != tag) | (y != (unsigned)i) | (z != owner) | (q != (tag ^ 0x5a5a5a5au));
    }
    return bad;
}

__global__ __launch_bounds__(T) void one_phase(uint4* buf, int n16, int p, unsigned* bad_out) {
    unsigned bad = 0;
    if (p > 0) {
        const unsigned other = (blockIdx.x + 1) % G;
        bad = phase_check(buf + ((size_t)((p - 1) & 1) * G + other) * n16, n16, (unsigned)p, other);
    }
    phase_write(buf + ((size_t)(p & 1) * G + blockIdx.x) * n16, n16, (unsigned)p + 1);
    if (bad) atomicAdd(bad_out, bad);
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    const int phases = 2000;
    unsigned *counter, *bad; CK(hipMalloc(&counter, 4096)); CK(hipMalloc(&bad, 4));
    uint4* buf; CK(hipMalloc(&buf, (size_t)2 * G * 65536));
    for (int bytes : {0, 4096, 16384, 65536}) {
        const int n16 = bytes / 16;
        double us[8]; unsigned nbad[8];
        for (int mode = 0; mode < 8; ++mode) {
            double best = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                CK(hipMemsetAsync(counter, 0, 4096, s)); CK(hipMemsetAsync(bad, 0, 4, s));
                CK(hipStreamSynchronize(s));
                const auto t0 = std::chrono::steady_clock::now();
                if (mode == 0) hipLaunchKernelGGL(persistent<0>, dim3(G), dim3(T), 0, s, buf, n16, phases, counter, bad);
                else if (mode == 1) hipLaunchKernelGGL(persistent<1>, dim3(G), dim3(T), 0, s, buf, n16, phases, counter, bad);
                else if (mode == 2) hipLaunchKernelGGL(persistent<2>, dim3(G), dim3(T), 0, s, buf, n16, phases, counter, bad);
                else if (mode == 3) hipLaunchKernelGGL(persistent<3>, dim3(G), dim3(T), 0, s, buf, n16, phases, counter, bad);
                else if (mode == 4) hipLaunchKernelGGL(persistent<4>, dim3(G), dim3(T), 0, s, buf, n16, phases, counter, bad);
                else if (mode == 5) hipLaunchKernelGGL(persistent<5>, dim3(G), dim3(T), 0, s, buf, n16, phases, counter, bad);
                else if (mode == 6) hipLaunchKernelGGL(persistent<6>, dim3(G), dim3(T), 0, s, buf, n16, phases, counter, bad);
                else for (int p = 0; p < phases; ++p) hipLaunchKernelGGL(one_phase, dim3(G), dim3(T), 0, s, buf, n16, p, bad);
                CK(hipStreamSynchronize(s));
                const double t = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / phases;
                if (t < best) best = t;
            }
            us[mode] = best;
            CK(hipMemcpy(&nbad[mode], bad, 4, hipMemcpyDeviceToHost));
        }
        printf("%6d bytes per workgroup per phase (%5.2f MB per phase): persistent kernel, us per phase: one counter %.2f, counter + flag "
               "%.2f, per-XCD counters + flag %.2f (%u bad), counter + flag without fences %.2f (%u stale reads), 32 workgroups of ONE XCD without "
               "L2 write-back %.2f (%u bad), the same fences on 256 workgroups %.2f (%u stale reads), ONE XCD with L1-bypassing reads instead of the acquire %.2f (%u bad); one "
               "launch per phase %.2f us (%u bad)\n",
               bytes, bytes * (double)G / 1e6, us[0], us[1], us[2], nbad[0] + nbad[1] + nbad[2], us[3], nbad[3], us[4], nbad[4], us[5], nbad[5],
               us[6], nbad[6], us[7], nbad[7]);
    }
    return 0;
}
