// common.h -- shared host/device helpers of libyolo4hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include <string>

#include "../../include/yolo4hip.h"

namespace y4 {

void set_error(const char* fmt, ...);

#define Y4_CHECK_HIP(expr)                                                                    \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            y4::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return Y4_EHIP;                                                                   \
        }                                                                                     \
    } while (0)

#define Y4_REQUIRE(cond, code, ...)       \
    do {                                  \
        if (!(cond)) {                    \
            y4::set_error(__VA_ARGS__);   \
            return (code);                \
        }                                 \
    } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a per-DEVICE property of a kernel: one flag per (call site, device
// ordinal), so a second engine on another GPU of the same process -- or a second thread -- opts in too.  Setting the
// attribute twice is harmless, so a lost race costs one redundant call.
struct PerDeviceOnce {
    std::atomic<uint64_t> done{0};
    // returns the bit of the current device if the action is still due there, else 0
    uint64_t due() {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return 1;
        const uint64_t bit = 1ull << (dev & 63);
        return (done.load(std::memory_order_acquire) & bit) ? 0 : bit;
    }
    void mark(uint64_t bit) { done.fetch_or(bit, std::memory_order_release); }
};

inline int elem_size(int dtype) { return dtype == Y4_F32 ? 4 : 2; }
inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

constexpr int COUT_PAD = 128;      // packed weight matrices have a multiple of this many rows
// THE K order of every conv kernel (round 5).  The K axis of a k x k conv over Cin channels runs  chunk of KC channels (outer) ->
// tap (ky, kx) -> channel inside the chunk,  KC = min(Cin, K_CHUNK) for 3x3 convs and Cin for 1x1 convs:
//     k = (chunk * k*k + tap) * KC + c,    input channel = chunk * KC + c.
// Packed weights are laid out along it ([cout_pad][Cin/KC][k*k][KC]), every kernel walks it in ascending order, and the MFMAs
// accumulate in that order -- so all kernels of one dtype still agree bit for bit.  Until round 4 the order was tap-major
// ([kh][kw][Cin]); chunk-major is what lets a kernel keep ONE 64-channel halo tile in LDS and run all nine taps from it
// (conv_halo_kernel.h) without changing the sums.  For Cin <= 64 the two orders are the same.
constexpr int K_CHUNK = 64;
inline __host__ __device__ int k_chunk_channels(int cin, int ksize) { return ksize == 1 || cin < K_CHUNK ? cin : K_CHUNK; }
constexpr int ZERO_PAGE_BYTES = 256;

// ---- device-side storage conversions (bf16 is stored as raw uint16_t, fp16 as _Float16).
// Conversions go through the compiler's native __bf16 / _Float16 vector types so that gfx950's packed
// converts (v_cvt_pk_bf16_f32: two round-to-nearest-even results per instruction) are used.
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(8))) float f32x8_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

__device__ __forceinline__ float bf16_to_f32(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }

template <int DT> struct Elem;
template <> struct Elem<Y4_F32> {
    using type = float;
    static constexpr int EPC = 4;                               // elements per 16-byte chunk
    static __device__ __forceinline__ float ld(float v) { return v; }
    static __device__ __forceinline__ float st(float v) { return v; }
    static __device__ __forceinline__ void load_chunk(const void* p, float* v) {
        const f32x4_t x = *(const f32x4_t*)p;
        v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
    }
    static __device__ __forceinline__ void store_chunk(void* p, const float* v) {
        *(f32x4_t*)p = f32x4_t{v[0], v[1], v[2], v[3]};
    }
};
template <> struct Elem<Y4_BF16> {
    using type = uint16_t;
    static constexpr int EPC = 8;
    static __device__ __forceinline__ float ld(uint16_t v) { return bf16_to_f32(v); }
    static __device__ __forceinline__ uint16_t st(float v) { return f32_to_bf16(v); }
    static __device__ __forceinline__ void load_chunk(const void* p, float* v) {
        const u32x4_t x = *(const u32x4_t*)p;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(x[i] << 16);
            v[2 * i + 1] = __uint_as_float(x[i] & 0xffff0000u);
        }
    }
    static __device__ __forceinline__ void store_chunk(void* p, const float* v) {
        const f32x8_t x = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
        *(bf16x8_t*)p = __builtin_convertvector(x, bf16x8_t);
    }
};
template <> struct Elem<Y4_F16> {
    using type = _Float16;
    static constexpr int EPC = 8;
    static __device__ __forceinline__ float ld(_Float16 v) { return (float)v; }
    static __device__ __forceinline__ _Float16 st(float v) { return (_Float16)v; }
    static __device__ __forceinline__ void load_chunk(const void* p, float* v) {
        const f32x8_t x = __builtin_convertvector(*(const f16x8_t*)p, f32x8_t);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = x[i];
    }
    static __device__ __forceinline__ void store_chunk(void* p, const float* v) {
        const f32x8_t x = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
        *(f16x8_t*)p = __builtin_convertvector(x, f16x8_t);
    }
};

// Activations of the reference's conv() unit (custom_layers.py:5-31).
// mish(x) = x*tanh(softplus(x)); with e = exp(x), tanh(log(1+e)) = (e*e+2e)/(e*e+2e+2), which has no
// cancellation for negative x; for x > 20 tanh(softplus(x)) == 1 in fp32 (TF's softplus also returns x
// beyond its threshold).
template <bool FAST>
__device__ __forceinline__ float mish_f(float x) {
    if (FAST) {
        // 16-bit storage paths.  With e = exp(x):  tanh(softplus(x)) = 1 - 2/(e*e + 2e + 2), so
        //     mish(x) = x - x * r,   r = 1 / (e*(e/2 + 1) + 1)   (= 2/(e*e + 2e + 2))
        // = 5 plain VALU ops + v_exp_f32 + v_rcp_f32 (~1 ulp each), every plain op an FMA/MUL that exists in packed form
        // (see bn_act4: two elements per instruction).  No clamp: e = inf gives r = 0 and mish = x; e = 0 gives r = 1 and
        // mish = 0.  The final FMA cancels only for x << 0, where the absolute error |x| * 1.5e-7 stays below the storage
        // type's resolution (|mish(x)| < 3e-3 for x < -8).  bn_act4's packed sequence is this one, op for op.
        const float e = __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
        const float r = __builtin_amdgcn_rcpf(fmaf(e, fmaf(e, 0.5f, 1.f), 1.f));
        return fmaf(-x, r, x);
    }
    const float e = expf(fminf(x, 20.f));
    const float n = e * (e + 2.f);
    return x > 20.f ? x : x * (n / (n + 2.f));
}

// Exact unsigned division by a runtime-constant divisor for dividends < 2^31 (pixel indices): q = n / d via
// one multiply-high and a shift; mul/shr are computed on the host (fastdiv_make).
struct FastDiv {
    uint32_t mul, shr, d;
};
inline FastDiv fastdiv_make(uint32_t d) {
    FastDiv f{0, 0, d};
    if (d <= 1) return f;
    uint32_t lg = 0;
    while ((1ull << lg) < d) ++lg;
    const uint32_t p = 31 + lg;
    f.mul = (uint32_t)(((1ull << p) + d - 1) / d);
    f.shr = p - 32;
    return f;
}
__device__ __forceinline__ uint32_t fastdiv(uint32_t n, const FastDiv f) {
    return f.d <= 1 ? n : (__umulhi(n, f.mul) >> f.shr);
}
__device__ __forceinline__ float leaky_f(float x) { return x > 0.f ? x : 0.1f * x; }
// max(x, 0.1x) == LeakyReLU(0.1) for every finite x; one v_max without the NaN-canonicalising v_max pair
// that fmaxf() costs (16-bit paths only; the fp32 path keeps the select form above)
__device__ __forceinline__ float leaky_fast(float x) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(0.1f * x));
    return r;
}

template <bool FAST>
__device__ __forceinline__ float apply_act(float x, int act) {
    if (act == Y4_ACT_MISH) return mish_f<FAST>(x);
    if (act == Y4_ACT_LEAKY) return FAST ? leaky_fast(x) : leaky_f(x);
    return x;
}
template <bool FAST, int ACT>
__device__ __forceinline__ float apply_act_t(float x) {
    if (ACT == Y4_ACT_MISH) return mish_f<FAST>(x);
    if (ACT == Y4_ACT_LEAKY) return FAST ? leaky_fast(x) : leaky_f(x);
    return x;
}

// out[r] = act(a[r] * sc[r] + sh[r]), r < 4 (one MFMA accumulator fragment of a lane): THE epilogue arithmetic of every
// conv kernel.  Where it is VALU-bound (Mish: the 304^2 / 152^2 stages), the 16-bit paths run it two elements per
// instruction -- v_pk_fma_f32 / v_pk_mul_f32 issue at the rate of their scalar forms (measured on MI355X: v_fma_f32 4.5,
// v_pk_fma_f32 4.9 cycles per wave instruction per SIMD; v_exp_f32 / v_rcp_f32 8.5) and are IEEE per element, so the
// results equal mish_f<true> bit for bit.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
template <bool FAST, int ACT>
__device__ __forceinline__ void bn_act4(const f32x4_t& a, const float* sc, const float* sh, float* out) {
    if constexpr (FAST && ACT == Y4_ACT_MISH) {
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
            const f32x2_t x = __builtin_elementwise_fma(f32x2_t{a[r], a[r + 1]}, f32x2_t{sc[r], sc[r + 1]}, f32x2_t{sh[r], sh[r + 1]});
            const f32x2_t t = x * 1.4426950408889634f;
            const f32x2_t e = {__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
            const f32x2_t dh = __builtin_elementwise_fma(e, __builtin_elementwise_fma(e, f32x2_t{0.5f, 0.5f}, f32x2_t{1.f, 1.f}), f32x2_t{1.f, 1.f});
            const f32x2_t rr = {__builtin_amdgcn_rcpf(dh.x), __builtin_amdgcn_rcpf(dh.y)};
            const f32x2_t y = __builtin_elementwise_fma(-x, rr, x);
            out[r] = y.x; out[r + 1] = y.y;
        }
    } else if constexpr (FAST && ACT == Y4_ACT_LEAKY) {
        // LeakyReLU on the 16-bit paths: the BN affine and 0.1x as packed FMA / MUL (IEEE per element = fmaf / one multiply), then
        // max(x, 0.1x) per element (leaky_fast's v_max): 4 instructions per two values instead of 6, same bits
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
            const f32x2_t x = __builtin_elementwise_fma(f32x2_t{a[r], a[r + 1]}, f32x2_t{sc[r], sc[r + 1]}, f32x2_t{sh[r], sh[r + 1]});
            const f32x2_t s = x * 0.1f;
            asm("v_max_f32 %0, %1, %2" : "=v"(out[r]) : "v"(x.x), "v"(s.x));
            asm("v_max_f32 %0, %1, %2" : "=v"(out[r + 1]) : "v"(x.y), "v"(s.y));
        }
    } else if constexpr (FAST) {
        // linear (the heads) on the 16-bit paths: the BN affine as packed FMAs (IEEE per element = fmaf)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
            const f32x2_t x = __builtin_elementwise_fma(f32x2_t{a[r], a[r + 1]}, f32x2_t{sc[r], sc[r + 1]}, f32x2_t{sh[r], sh[r + 1]});
            out[r] = apply_act_t<true, ACT>(x.x); out[r + 1] = apply_act_t<true, ACT>(x.y);
        }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) out[r] = apply_act_t<FAST, ACT>(fmaf(a[r], sc[r], sh[r]));
    }
}

}  // namespace y4
