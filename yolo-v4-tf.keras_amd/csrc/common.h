// common.h -- shared host/device helpers of libyolo4hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
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

inline int elem_size(int dtype) { return dtype == Y4_F32 ? 4 : 2; }
inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

constexpr int COUT_PAD = 128;      // packed weight matrices have a multiple of this many rows
constexpr int ZERO_PAGE_BYTES = 256;

// ---- device-side storage conversions (bf16 is raw uint16_t, fp16 is _Float16)
__device__ __forceinline__ float bf16_to_f32(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);                                              // round to nearest even
    return (uint16_t)(u >> 16);
}

template <int DT> struct Elem;
template <> struct Elem<Y4_F32> {
    using type = float;
    static __device__ __forceinline__ float ld(float v) { return v; }
    static __device__ __forceinline__ float st(float v) { return v; }
};
template <> struct Elem<Y4_BF16> {
    using type = uint16_t;
    static __device__ __forceinline__ float ld(uint16_t v) { return bf16_to_f32(v); }
    static __device__ __forceinline__ uint16_t st(float v) { return f32_to_bf16(v); }
};
template <> struct Elem<Y4_F16> {
    using type = _Float16;
    static __device__ __forceinline__ float ld(_Float16 v) { return (float)v; }
    static __device__ __forceinline__ _Float16 st(float v) { return (_Float16)v; }
};

// Activations of the reference's conv() unit (custom_layers.py:5-31).
// mish(x) = x*tanh(softplus(x)); with e = exp(x), tanh(log(1+e)) = (e*e+2e)/(e*e+2e+2), which has no
// cancellation for negative x; for x > 20 tanh(softplus(x)) == 1 in fp32 (TF's softplus also returns x
// beyond its threshold).
template <bool FAST>
__device__ __forceinline__ float mish_f(float x) {
    float e = FAST ? __expf(fminf(x, 20.f)) : expf(fminf(x, 20.f));
    float n = e * (e + 2.f);
    float t = FAST ? __fdividef(n, n + 2.f) : n / (n + 2.f);
    return x > 20.f ? x : x * t;
}
__device__ __forceinline__ float leaky_f(float x) { return x > 0.f ? x : 0.1f * x; }

template <bool FAST>
__device__ __forceinline__ float apply_act(float x, int act) {
    if (act == Y4_ACT_MISH) return mish_f<FAST>(x);
    if (act == Y4_ACT_LEAKY) return leaky_f(x);
    return x;
}

}  // namespace y4
