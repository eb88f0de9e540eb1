// stem_common.h -- the MFMA stem's K layout, shared by stem_mfma_kernel (misc_kernels.hip), the fused convs 0+1
// kernel (stem_down.hip) and pack_stem_kernel.
// The 3x3x3 patch of pixel (y, x) is three runs of 9 consecutive floats in the NHWC image, one per row, starting
// at pixel x-1.  K = 27 is padded to the MFMA's 32 as four lane groups of 8:
//   group g < 3 : row y+g-1, floats 0..7 of its run (pixels x-1 and x, channels 0 and 1 of pixel x+1)
//                 -> two 16-byte loads from one address per lane instead of eight scattered dword gathers
//   group 3     : float 8 of the runs of rows y-1, y, y+1 (channel 2 of pixel x+1), then five zeros.
#pragma once
#include "common.h"

namespace y4 {

// K slot (g, e) -> (ky, j = kx*3 + ci), or ky = -1 for the zero padding.  Host and device.
__host__ __device__ inline void stem_k_slot(int g, int e, int& ky, int& j) {
    if (g < 3) { ky = g; j = e; }
    else if (e < 3) { ky = e; j = 8; }
    else { ky = -1; j = 0; }
}

// ---- image element types.  float: the preprocessed tensor of `Yolov4.preprocess_img` (reference models.py:95-98).
// uint8_t: a frame already at network size, BEFORE the `/ 255.`; the stem then applies it on the fly:
//   16-bit dtypes: float(v) * (1/255f), whose bf16 / fp16 rounding equals that of float(double(v)/255.) for all 256 values;
//   float32      : one Newton step on top, q' = fma(fma(-255, q, v), 1/255f, q), which equals float(double(v)/255.) for
//                  all 256 values (both checked exhaustively: tests/test_host.py, tests/test_gpu_api.py).
template <bool EXACT32>
__device__ __forceinline__ float unit_from_u8(uint32_t v) {
    const float f = (float)v, r = 1.0f / 255.0f;
    const float q = f * r;
    if (!EXACT32) return q;
    return fmaf(fmaf(-255.0f, q, f), r, q);
}
template <bool EXACT32> __device__ __forceinline__ float img_elem(const float* p) { return *p; }
template <bool EXACT32> __device__ __forceinline__ float img_elem(const uint8_t* p) { return unit_from_u8<EXACT32>(*p); }
// 8 consecutive elements starting at p (no alignment beyond the element's own)
template <bool EXACT32> __device__ __forceinline__ void img_run8(const float* p, float v[8]) {
    f32x4_t a, b;
    __builtin_memcpy(&a, p, 16);
    __builtin_memcpy(&b, p + 4, 16);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b[e]; }
}
template <bool EXACT32> __device__ __forceinline__ void img_run8(const uint8_t* p, float v[8]) {
    uint32_t w[2];
    __builtin_memcpy(w, p, 8);                       // one (unaligned) global_load_dwordx2
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = unit_from_u8<EXACT32>((w[e >> 2] >> (8 * (e & 3))) & 0xffu);
}

// The same through a raw buffer descriptor over the whole image batch and a 32-bit ELEMENT offset (stem_down.hip's preloaded
// tiles: a 64-bit pointer per load in flight costs two registers each).  Loads need only the element's own alignment.
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
__device__ __forceinline__ void img_buf_run8(__amdgpu_buffer_rsrc_t rs, const float*, uint32_t off, float v[8]) {
    const u32x4_t a = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off * 4u), 0, 0);
    const u32x4_t b = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(off * 4u + 16u), 0, 0);
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[e] = __uint_as_float(a[e]); v[4 + e] = __uint_as_float(b[e]); }
}
__device__ __forceinline__ void img_buf_run8(__amdgpu_buffer_rsrc_t rs, const uint8_t*, uint32_t off, float v[8]) {
    const u32x2_t w = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 0, 0);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = unit_from_u8<false>((w[e >> 2] >> (8 * (e & 3))) & 0xffu);
}
__device__ __forceinline__ float img_buf_elem(__amdgpu_buffer_rsrc_t rs, const float*, uint32_t off) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(off * 4u), 0, 0));
}
__device__ __forceinline__ float img_buf_elem(__amdgpu_buffer_rsrc_t rs, const uint8_t*, uint32_t off) {
    return unit_from_u8<false>((uint32_t)__builtin_amdgcn_raw_buffer_load_b8(rs, (int)off, 0, 0));
}

// This lane's 8 K values for pixel (y, x) of the image at `img` (H x W x 3 elements).
// EDGE = false: rows y-1..y+1 and columns x-1..x+1 are inside the image for every lane of the wave.
template <bool EDGE, bool EXACT32 = false, class IMG = float>
__device__ __forceinline__ void stem_gather(const IMG* __restrict__ img, int y, int x, int H, int W, int g, float v[8]) {
    if (!EDGE) {
        if (g < 3) {
            img_run8<EXACT32>(img + ((y + g - 1) * W + x - 1) * 3, v);
        } else {
            const IMG* p = img + ((y - 1) * W + x + 1) * 3 + 2;
            v[0] = img_elem<EXACT32>(p); v[1] = img_elem<EXACT32>(p + W * 3); v[2] = img_elem<EXACT32>(p + 2 * W * 3);
#pragma unroll
            for (int e = 3; e < 8; ++e) v[e] = 0.f;
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int ky, j;
            stem_k_slot(g, e, ky, j);
            const int kx = j / 3, ci = j - kx * 3;
            const int yy = y + ky - 1, xx = x + kx - 1;
            const bool ok = ky >= 0 && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
            v[e] = ok ? img_elem<EXACT32>(img + (yy * W + xx) * 3 + ci) : 0.f;
        }
    }
}

}  // namespace y4
