// misc_kernels.hip -- the non-GEMM kernels of the forward pass:
//   stem conv (reference custom_layers.py:101: conv(input, 32, 3), Cin = 3 -> K = 27, too thin for MFMA tiles)
//   SPP max-pools + concat (reference custom_layers.py:130-134)
//   view -> dense float32 copies (what Keras returns from yolo_model.predict, reference models.py:514)
#include "kernels.h"
#include "stem_common.h"

namespace y4 {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

// ------------------------------------------------------------------------------------------ stem
// One thread = one output pixel x all COUT(32) channels.  The 27x32 weights are read with wave-uniform
// addresses from a [27][COUT] float table, so they arrive as scalar loads (s_load_dwordx16) and feed
// v_fma as SGPR operands: no LDS, no per-lane weight registers.  HBM-bound by design: reads 12 B/pixel
// (fp32 RGB, L1-shared between neighbours), writes COUT*sizeof(T) per pixel, fully coalesced.
template <int DT, int COUT, class IMG>
__global__ __launch_bounds__(256) void stem_conv_kernel(const IMG* __restrict__ img, const float* __restrict__ wk,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        typename Elem<DT>::type* __restrict__ out, int N, int H, int W,
                                                        int out_cstride, int out_coff, int act) {
    using E = Elem<DT>;
    using T = typename E::type;
    const int64_t total = (int64_t)N * H * W;
    const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (pix >= total) return;
    const int x = (int)(pix % W);
    const int64_t r = pix / W;
    const int y = (int)(r % H), n = (int)(r / H);
    float acc[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) acc[c] = 0.f;
#pragma unroll 1
    for (int ky = 0; ky < 3; ++ky) {
        const int yy = y + ky - 1;
        const bool oky = (unsigned)yy < (unsigned)H;
        const IMG* row = img + (((int64_t)n * H + (oky ? yy : 0)) * W) * 3;
        float v[9];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int xx = x + kx - 1;
            const bool ok = oky && (unsigned)xx < (unsigned)W;
            const IMG* ip = row + (ok ? xx : 0) * 3;
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) v[kx * 3 + ci] = ok ? img_elem<DT == Y4_F32>(ip + ci) : 0.f;
        }
        const float* wrow = wk + ky * 9 * COUT;        // wave-uniform -> scalar loads
#pragma unroll
        for (int j = 0; j < 9; ++j)
#pragma unroll
            for (int c = 0; c < COUT; ++c) acc[c] = fmaf(v[j], wrow[j * COUT + c], acc[c]);
    }
    constexpr int EPC = 16 / (int)sizeof(T);
    T* op = out + pix * out_cstride + out_coff;
    constexpr bool FAST = (DT != Y4_F32);
#pragma unroll
    for (int c = 0; c < COUT; c += EPC) {
        u32x4 raw;
        T* ov = (T*)&raw;
#pragma unroll
        for (int e = 0; e < EPC; ++e) ov[e] = E::st(apply_act<FAST>(acc[c + e] * scale[c + e] + shift[c + e], act));
        *(u32x4*)(op + c) = raw;
    }
}

// MFMA stem for the 16-bit paths.  GEMM view D[ch][px] = sum_k Wt[ch][k] * X[px][k] with K = 27 padded to 32:
// ONE v_mfma_f32_16x16x32 per 16 pixels x 16 channels.  The pixel operand is gathered straight from the float32
// image into registers (lane = (pixel l&15, k-chunk l>>4): 8 taps, converted to the 16-bit type), the weight
// operand is a pre-built fragment table (channel rows permuted so that each lane ends up with 8 CONSECUTIVE
// channels of its pixel -> one 16-byte store per lane, 1 KiB contiguous per wave).  No LDS.
typedef __attribute__((ext_vector_type(4))) float stem_f32x4;
typedef __attribute__((ext_vector_type(8))) short stem_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 stem_f16x8;

template <int DT, class IMG>
__global__ __launch_bounds__(256) void stem_mfma_kernel(const IMG* __restrict__ img, const u32x4* __restrict__ wfrag,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        typename Elem<DT>::type* __restrict__ out, int N, int H, int W,
                                                        int out_cstride, int out_coff, int act, FastDiv div_hw,
                                                        FastDiv div_w, int tiles_per_wave) {
    using E = Elem<DT>;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane & 15, g = lane >> 4;
    const u32x4 wf0 = wfrag[lane], wf1 = wfrag[64 + lane];
    float sc[8], sh[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) { sc[c] = scale[g * 8 + c]; sh[c] = shift[g * 8 + c]; }
    const int64_t total = (int64_t)N * H * W;
    const int HW = H * W;
    const bool tile_rows = (W & 15) == 0;                        // a 16-pixel tile never straddles two rows
    int64_t p0 = ((int64_t)blockIdx.x * 4 + wave) * tiles_per_wave * 16;
    constexpr bool FAST = true;
    for (int t = 0; t < tiles_per_wave; ++t, p0 += 16) {
        if (p0 >= total) break;                                   // wave-uniform
        const int64_t p = p0 + q;
        const bool pv = p < total;
        const uint32_t pp = (uint32_t)(pv ? p : total - 1);
        const int n = (int)fastdiv(pp, div_hw);
        const int rem = (int)pp - n * HW;
        const int y = (int)fastdiv((uint32_t)rem, div_w), x = rem - y * W;
        // wave-uniform: the 16 pixels are one run of a row (W % 16 == 0) away from the image border
        const int y0 = __builtin_amdgcn_readfirstlane(y), x0 = __builtin_amdgcn_readfirstlane(x);
        const bool interior = tile_rows && p0 + 16 <= total && y0 >= 1 && y0 <= H - 2 && x0 >= 1 && x0 + 16 <= W - 1;
        const IMG* imgn = img + (int64_t)n * HW * 3;
        float v[8];
        if (interior) stem_gather<false, false, IMG>(imgn, y, x, H, W, g, v);
        else stem_gather<true, false, IMG>(imgn, y, x, H, W, g, v);
        u32x4 xf;
        E::store_chunk(&xf, v);
        stem_f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
        if (DT == Y4_BF16) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(stem_bf16x8, wf0), __builtin_bit_cast(stem_bf16x8, xf), a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(stem_bf16x8, wf1), __builtin_bit_cast(stem_bf16x8, xf), a1, 0, 0, 0);
        } else {
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(stem_f16x8, wf0), __builtin_bit_cast(stem_f16x8, xf), a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(stem_f16x8, wf1), __builtin_bit_cast(stem_f16x8, xf), a1, 0, 0, 0);
        }
        float o[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            o[r] = apply_act<FAST>(fmaf(a0[r], sc[r], sh[r]), act);
            o[4 + r] = apply_act<FAST>(fmaf(a1[r], sc[4 + r], sh[4 + r]), act);
        }
        if (pv) {
            u32x4 packed;
            E::store_chunk(&packed, o);
            *(u32x4*)(out + p * out_cstride + out_coff + g * 8) = packed;
        }
    }
}

// Darknet (cout,3,3,3) -> (a) [(ky*3+kx)*3+ci][cout] float table for the fp32 stem kernel at wk[0..27*cout),
// (b) at byte 4096 the bf16 and at byte 6144 the fp16 MFMA weight-fragment tables [2 frags][64 lanes][8]:
//     fragment jn, lane l holds row i = l&15 = g'*4 + r'  <->  channel g'*8 + jn*4 + r', K slots 8*(l>>4) .. +7
//     in the order of stem_common.h
__global__ void pack_stem_kernel(const float* __restrict__ w_oihw, float* __restrict__ wk, int cout) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 27 * cout) {
        const int co = i % cout, k = i / cout;
        const int ci = k % 3, tap = k / 3;
        wk[i] = w_oihw[(co * 3 + ci) * 9 + tap];
    }
    if (cout == 32 && i < 2 * 64 * 8) {
        const int e = i & 7, l = (i >> 3) & 63, jn = i >> 9;
        const int row = l & 15, gq = row >> 2, rr = row & 3;
        const int ch = gq * 8 + jn * 4 + rr;
        int ky, j;
        stem_k_slot(l >> 4, e, ky, j);                  // K layout of stem_common.h
        float v = 0.f;
        if (ky >= 0) {
            const int kx = j / 3, ci = j - kx * 3;
            v = w_oihw[(ch * 3 + ci) * 9 + ky * 3 + kx];
        }
        ((uint16_t*)((char*)wk + 4096))[i] = Elem<Y4_BF16>::st(v);
        ((_Float16*)((char*)wk + 6144))[i] = Elem<Y4_F16>::st(v);
    }
}

int pack_stem_weights(const float* w_oihw, float* wk, int cout, hipStream_t stream) {
    Y4_REQUIRE(w_oihw && wk && cout > 0, Y4_EINVAL, "pack_stem_weights: bad argument");
    const int work = 27 * cout > 1024 ? 27 * cout : 1024;
    hipLaunchKernelGGL(pack_stem_kernel, dim3((work + 255) / 256), dim3(256), 0, stream, w_oihw, wk, cout);
    Y4_CHECK_HIP(hipGetLastError());
    return Y4_OK;
}

template <class IMG>
static int stem_conv_launch_t(int dtype, const IMG* imgs, int n, int h, int w, const float* wk, const float* scale,
                              const float* shift, int act, void* out, int out_cstride, int out_coff, hipStream_t stream) {
    const int64_t total = (int64_t)n * h * w;
    if (dtype == Y4_F32) {
        hipLaunchKernelGGL((stem_conv_kernel<Y4_F32, 32, IMG>), dim3((int)((total + 255) / 256)), dim3(256), 0, stream, imgs, wk, scale, shift,
                           (float*)out, n, h, w, out_cstride, out_coff, act);
    } else {
        const int tpw = 8;                                        // 16-pixel tiles per wave
        const int blocks = (int)((total + 4 * tpw * 16 - 1) / (4 * tpw * 16));
        const FastDiv dhw = fastdiv_make((uint32_t)(h * w)), dw = fastdiv_make((uint32_t)w);
        if (dtype == Y4_BF16)
            hipLaunchKernelGGL((stem_mfma_kernel<Y4_BF16, IMG>), dim3(blocks), dim3(256), 0, stream, imgs, (const u32x4*)((const char*)wk + 4096),
                               scale, shift, (uint16_t*)out, n, h, w, out_cstride, out_coff, act, dhw, dw, tpw);
        else if (dtype == Y4_F16)
            hipLaunchKernelGGL((stem_mfma_kernel<Y4_F16, IMG>), dim3(blocks), dim3(256), 0, stream, imgs, (const u32x4*)((const char*)wk + 6144),
                               scale, shift, (_Float16*)out, n, h, w, out_cstride, out_coff, act, dhw, dw, tpw);
        else { set_error("stem_conv: bad dtype %d", dtype); return Y4_EINVAL; }
    }
    Y4_CHECK_HIP(hipGetLastError());
    return Y4_OK;
}

// imgs: float32 [n,h,w,3] in [0,1], or (img_u8) uint8 [n,h,w,3] before the /255 (applied on the fly, stem_common.h)
int stem_conv_launch(int dtype, const void* imgs, int img_u8, int n, int h, int w, const float* wk, const float* scale,
                     const float* shift, int cout, int act, void* out, int out_cstride, int out_coff,
                     hipStream_t stream) {
    Y4_REQUIRE(cout == 32, Y4_EINVAL, "stem_conv: cout %d (the plan's stem has 32 filters)", cout);
    Y4_REQUIRE(imgs && wk && scale && shift && out, Y4_EINVAL, "stem_conv: null pointer");
    const int epc = 16 / elem_size(dtype);
    Y4_REQUIRE(out_cstride % epc == 0 && out_coff % epc == 0, Y4_EINVAL, "stem_conv: output view not 16-byte aligned");
    Y4_REQUIRE((int64_t)n * h * w * 3 < (1ll << 31), Y4_EINVAL, "stem_conv: image batch too large");
    if (img_u8) return stem_conv_launch_t(dtype, (const uint8_t*)imgs, n, h, w, wk, scale, shift, act, out, out_cstride, out_coff, stream);
    return stem_conv_launch_t(dtype, (const float*)imgs, n, h, w, wk, scale, shift, act, out, out_cstride, out_coff, stream);
}

// ------------------------------------------------------------------------------------------- SPP
// buf [n, side, side, 4c]: x lives in channels [3c,4c); writes maxpool13 -> [0,c), maxpool9 -> [c,2c),
// maxpool5 -> [2c,3c).  stride 1, 'same': windows are clipped at the border (Keras pads with -inf).
// The 5/9/13 windows are nested, so one sweep of the 13x13 neighbourhood yields all three maxima.
template <int DT>
__global__ __launch_bounds__(256) void spp_kernel(typename Elem<DT>::type* __restrict__ buf, int N, int S, int C) {
    using E = Elem<DT>;
    using T = typename E::type;
    constexpr int EPC = 16 / (int)sizeof(T);
    const int cchunks = C / EPC;
    const int64_t total = (int64_t)N * S * S * cchunks;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int cc = (int)(i % cchunks);
    int64_t r = i / cchunks;
    const int x = (int)(r % S); r /= S;
    const int y = (int)(r % S);
    const int n = (int)(r / S);
    const int cs = 4 * C;
    float m5[EPC], m9[EPC], m13[EPC];
#pragma unroll
    for (int e = 0; e < EPC; ++e) m5[e] = m9[e] = m13[e] = -INFINITY;
    const T* base = buf + (int64_t)n * S * S * cs + 3 * C + cc * EPC;
    for (int dy = -6; dy <= 6; ++dy) {
        const int yy = y + dy;
        if ((unsigned)yy >= (unsigned)S) continue;
        const int ady = dy < 0 ? -dy : dy;
        for (int dx = -6; dx <= 6; ++dx) {
            const int xx = x + dx;
            if ((unsigned)xx >= (unsigned)S) continue;
            const int adx = dx < 0 ? -dx : dx;
            const int rad = ady > adx ? ady : adx;
            const u32x4 raw = *(const u32x4*)(base + ((int64_t)yy * S + xx) * cs);
            const T* v = (const T*)&raw;
#pragma unroll
            for (int e = 0; e < EPC; ++e) {
                const float f = E::ld(v[e]);
                m13[e] = fmaxf(m13[e], f);
                if (rad <= 4) m9[e] = fmaxf(m9[e], f);
                if (rad <= 2) m5[e] = fmaxf(m5[e], f);
            }
        }
    }
    T* op = buf + (((int64_t)n * S + y) * S + x) * cs + cc * EPC;
    u32x4 o13, o9, o5;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        ((T*)&o13)[e] = E::st(m13[e]);
        ((T*)&o9)[e] = E::st(m9[e]);
        ((T*)&o5)[e] = E::st(m5[e]);
    }
    *(u32x4*)(op) = o13;
    *(u32x4*)(op + C) = o9;
    *(u32x4*)(op + 2 * C) = o5;
}

// LDS-resident separable version (used when the 19x19 / 13x13 plane fits): one workgroup per (image, group of
// 4 channel chunks = 64 bytes per pixel).  The plane is staged in LDS once; pass 1 forms the horizontal running
// maxima of radius 2/4/6 (13 LDS reads -> 3 results, nested windows), pass 2 the vertical ones (5+9+13 reads) and
// writes the three concat slices: 40 LDS reads per output chunk instead of 169 global loads.
// CPG = 16-byte channel chunks per workgroup: 2 (46 KB of LDS at 19x19 -> three workgroups per CU overlap their load / pool /
// store phases: 0.059 -> 0.045 ms; 1 chunk: 0.054) instead of 4 (92 KB, one workgroup per CU).
// Packed max in the storage type: the pooling compares 16-byte chunks without unpacking them to float.  fp16: v_pk_max_f16.
// bf16 has no packed max: a chunk is kept in "sortable" form (x ^ 0x7fff when the sign bit is set -- an involution that keeps
// the sign bit, after which signed 16-bit order is float order) from the staging load to the final store, and the max is
// v_pk_max_i16.  fp32: fmaxf.  max is exact in every form (signed zeros may come out as +0 where fmaxf kept -0).
typedef short i16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
template <int DT> struct PMax;
template <> struct PMax<Y4_BF16> {
    static __device__ __forceinline__ u32x4 enc(const u32x4& r) { return r ^ (((r >> 15) & 0x00010001u) * 0x7fffu); }
    static __device__ __forceinline__ u32x4 dec(const u32x4& r) { return enc(r); }
    static __device__ __forceinline__ u32x4 mx(const u32x4& a, const u32x4& b) {
        return __builtin_bit_cast(u32x4, __builtin_elementwise_max(__builtin_bit_cast(i16x8, a), __builtin_bit_cast(i16x8, b)));
    }
    static __device__ __forceinline__ u32x4 lowest() { return enc(u32x4{0xff80ff80u, 0xff80ff80u, 0xff80ff80u, 0xff80ff80u}); }
};
template <> struct PMax<Y4_F16> {
    static __device__ __forceinline__ u32x4 enc(const u32x4& r) { return r; }
    static __device__ __forceinline__ u32x4 dec(const u32x4& r) { return r; }
    static __device__ __forceinline__ u32x4 mx(const u32x4& a, const u32x4& b) {
        return __builtin_bit_cast(u32x4, __builtin_elementwise_max(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b)));
    }
    static __device__ __forceinline__ u32x4 lowest() { return u32x4{0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u, 0xfc00fc00u}; }
};
template <> struct PMax<Y4_F32> {
    static __device__ __forceinline__ u32x4 enc(const u32x4& r) { return r; }
    static __device__ __forceinline__ u32x4 dec(const u32x4& r) { return r; }
    static __device__ __forceinline__ u32x4 mx(const u32x4& a, const u32x4& b) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        return __builtin_bit_cast(u32x4, __builtin_elementwise_max(__builtin_bit_cast(f4, a), __builtin_bit_cast(f4, b)));
    }
    static __device__ __forceinline__ u32x4 lowest() { return u32x4{0xff800000u, 0xff800000u, 0xff800000u, 0xff800000u}; }
};

template <int DT, int CPG>
__global__ __launch_bounds__(256) void spp_lds_kernel(typename Elem<DT>::type* __restrict__ buf, int N, int S, int C) {
    using E = Elem<DT>;
    using T = typename E::type;
    using M = PMax<DT>;
    constexpr int EPC = E::EPC;
    extern __shared__ __attribute__((aligned(16))) char ssm[];
    const int P = S * S;
    u32x4* X = (u32x4*)ssm;             // [P][CPG]
    u32x4* H2 = X + P * CPG;
    u32x4* H4 = H2 + P * CPG;
    u32x4* H6 = H4 + P * CPG;
    const int groups = C / (CPG * EPC);
    const int n = blockIdx.x / groups, g = blockIdx.x - n * groups;
    const int cs = 4 * C;
    T* const img = buf + (int64_t)n * P * cs;
    const int ch0 = g * CPG * EPC;
    for (int t = threadIdx.x; t < P * CPG; t += 256) {
        const int px = t / CPG, q = t - px * CPG;
        X[t] = M::enc(*(const u32x4*)(img + (int64_t)px * cs + 3 * C + ch0 + q * EPC));
    }
    __syncthreads();
    for (int t = threadIdx.x; t < P * CPG; t += 256) {
        const int px = t / CPG, q = t - px * CPG;
        const int y = px / S, x = px - y * S;
        u32x4 m = M::lowest();
        const u32x4* row = X + (y * S) * CPG + q;
        for (int dx = -2; dx <= 2; ++dx)
            if ((unsigned)(x + dx) < (unsigned)S) m = M::mx(m, row[(x + dx) * CPG]);
        H2[t] = m;
        for (int k = 3; k <= 4; ++k) {
            if (x - k >= 0) m = M::mx(m, row[(x - k) * CPG]);
            if (x + k < S) m = M::mx(m, row[(x + k) * CPG]);
        }
        H4[t] = m;
        for (int k = 5; k <= 6; ++k) {
            if (x - k >= 0) m = M::mx(m, row[(x - k) * CPG]);
            if (x + k < S) m = M::mx(m, row[(x + k) * CPG]);
        }
        H6[t] = m;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < P * CPG; t += 256) {
        const int px = t / CPG, q = t - px * CPG;
        const int y = px / S, x = px - y * S;
        u32x4 m5 = M::lowest(), m9 = m5, m13 = m5;
        for (int dy = -6; dy <= 6; ++dy) {
            const int yy = y + dy;
            if ((unsigned)yy >= (unsigned)S) continue;
            const int o = (yy * S + x) * CPG + q;
            const int ady = dy < 0 ? -dy : dy;
            m13 = M::mx(m13, H6[o]);
            if (ady <= 4) m9 = M::mx(m9, H4[o]);
            if (ady <= 2) m5 = M::mx(m5, H2[o]);
        }
        T* op = img + (int64_t)px * cs + ch0 + q * EPC;
        *(u32x4*)(op) = M::dec(m13);
        *(u32x4*)(op + C) = M::dec(m9);
        *(u32x4*)(op + 2 * C) = M::dec(m5);
    }
}

template <int DT>
static int spp_dispatch(void* buf, int n, int side, int c, hipStream_t stream) {
    using T = typename Elem<DT>::type;
    constexpr int EPC = Elem<DT>::EPC;
    constexpr int CPG = 2;
    const size_t lds = (size_t)side * side * CPG * 16 * 4;
    if (c % (CPG * EPC) == 0 && lds <= 150 * 1024) {
        static PerDeviceOnce once;
        if (const uint64_t bit = once.due()) {
            Y4_CHECK_HIP(hipFuncSetAttribute((const void*)spp_lds_kernel<DT, CPG>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
            once.mark(bit);
        }
        hipLaunchKernelGGL((spp_lds_kernel<DT, CPG>), dim3(n * (c / (CPG * EPC))), dim3(256), lds, stream, (T*)buf, n, side, c);
    } else {
        const int64_t total = (int64_t)n * side * side * (c / EPC);
        hipLaunchKernelGGL(spp_kernel<DT>, dim3((int)((total + 255) / 256)), dim3(256), 0, stream, (T*)buf, n, side, c);
    }
    Y4_CHECK_HIP(hipGetLastError());
    return Y4_OK;
}

int spp_launch(int dtype, void* buf, int n, int side, int c, hipStream_t stream) {
    Y4_REQUIRE(buf && n > 0 && side > 0, Y4_EINVAL, "spp: bad arguments");
    const int epc = 16 / elem_size(dtype);
    Y4_REQUIRE(c % epc == 0, Y4_EINVAL, "spp: channels %d not a multiple of %d", c, epc);
    switch (dtype) {
        case Y4_F32: return spp_dispatch<Y4_F32>(buf, n, side, c, stream);
        case Y4_BF16: return spp_dispatch<Y4_BF16>(buf, n, side, c, stream);
        case Y4_F16: return spp_dispatch<Y4_F16>(buf, n, side, c, stream);
    }
    set_error("spp: bad dtype %d", dtype);
    return Y4_EINVAL;
}

// ------------------------------------------------------------------------------ image pre-processing
// Yolov4.preprocess_img (reference models.py:95-98): cv2.resize(img, (W,H)) [INTER_LINEAR, plain stretch] then
// img / 255.  uint8 RGB [h,w,3] on device -> float32 [H,W,3] in [0,1].  Restates OpenCV's uint8 fixed-point
// bilinear scheme (half-pixel centres, 11-bit coefficients, two rounding shifts) exactly like the host version
// yolo4hip/prepost.py: resize_bilinear, so both paths give identical floats.
__device__ __forceinline__ void lin_coeff(int d, int dst, int src, int& s0, int& s1, int& a0, int& a1) {
    const double scale = (double)src / (double)dst;
    double f = ((double)d + 0.5) * scale - 0.5;
    int s = (int)floor(f);
    float fr = (float)(f - (double)s);
    if (s < 0) { fr = 0.f; s = 0; }
    if (s >= src - 1) { fr = 0.f; s = src - 1; }
    s0 = s;
    s1 = s + 1 < src ? s + 1 : src - 1;
    a1 = (int)rintf(fr * 2048.0f);
    a0 = (int)rintf((1.0f - fr) * 2048.0f);
}

__global__ void preprocess_u8_kernel(const uint8_t* __restrict__ img, int h, int w, float* __restrict__ out, int H, int W) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * W) return;
    const int y = i / W, x = i - y * W;
    int x0, x1, ax0, ax1, y0, y1, ay0, ay1;
    lin_coeff(x, W, w, x0, x1, ax0, ax1);
    lin_coeff(y, H, h, y0, y1, ay0, ay1);
    const bool same = (h == H && w == W);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int v;
        if (same) {
            v = img[(y * w + x) * 3 + c];
        } else {
            const int top = img[(y0 * w + x0) * 3 + c] * ax0 + img[(y0 * w + x1) * 3 + c] * ax1;   // x2048
            const int bot = img[(y1 * w + x0) * 3 + c] * ax0 + img[(y1 * w + x1) * 3 + c] * ax1;
            v = (((ay0 * (top >> 4)) >> 16) + ((ay1 * (bot >> 4)) >> 16) + 2) >> 2;
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
        }
        out[i * 3 + c] = (float)((double)v / 255.0);
    }
}

// The resize half alone, batched: uint8 [n,h,w,3] -> uint8 [n,H,W,3] (cv2.resize's own output type); the `/ 255.` then
// happens inside the stem's operand load (y4_forward_u8), so no float image tensor exists at all.
__global__ void resize_u8_kernel(const uint8_t* __restrict__ img, int n, int h, int w, uint8_t* __restrict__ out, int H, int W) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)n * H * W) return;
    const int b = (int)(i / (H * W)), r = (int)(i - (int64_t)b * H * W);
    const int y = r / W, x = r - y * W;
    const uint8_t* src = img + (int64_t)b * h * w * 3;
    int x0, x1, ax0, ax1, y0, y1, ay0, ay1;
    lin_coeff(x, W, w, x0, x1, ax0, ax1);
    lin_coeff(y, H, h, y0, y1, ay0, ay1);
    const bool same = (h == H && w == W);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        int v;
        if (same) {
            v = src[(y * w + x) * 3 + c];
        } else {
            const int top = src[(y0 * w + x0) * 3 + c] * ax0 + src[(y0 * w + x1) * 3 + c] * ax1;   // x2048
            const int bot = src[(y1 * w + x0) * 3 + c] * ax0 + src[(y1 * w + x1) * 3 + c] * ax1;
            v = (((ay0 * (top >> 4)) >> 16) + ((ay1 * (bot >> 4)) >> 16) + 2) >> 2;
            v = v < 0 ? 0 : (v > 255 ? 255 : v);
        }
        out[i * 3 + c] = (uint8_t)v;
    }
}

int resize_u8_launch(const uint8_t* img, int n, int h, int w, uint8_t* out, int H, int W, hipStream_t stream) {
    Y4_REQUIRE(img && out && n > 0 && h > 0 && w > 0 && H > 0 && W > 0, Y4_EINVAL, "resize_u8: bad argument");
    Y4_REQUIRE((int64_t)n * h * w * 3 < (1ll << 31) && (int64_t)n * H * W * 3 < (1ll << 31), Y4_EINVAL, "resize_u8: batch too large");
    const int64_t total = (int64_t)n * H * W;
    hipLaunchKernelGGL(resize_u8_kernel, dim3((int)((total + 255) / 256)), dim3(256), 0, stream, img, n, h, w, out, H, W);
    Y4_CHECK_HIP(hipGetLastError());
    return Y4_OK;
}

int preprocess_u8_launch(const uint8_t* img, int h, int w, float* out, int H, int W, hipStream_t stream) {
    Y4_REQUIRE(img && out && h > 0 && w > 0 && H > 0 && W > 0, Y4_EINVAL, "preprocess_u8: bad argument");
    hipLaunchKernelGGL(preprocess_u8_kernel, dim3((H * W + 255) / 256), dim3(256), 0, stream, img, h, w, out, H, W);
    Y4_CHECK_HIP(hipGetLastError());
    return Y4_OK;
}

// ----------------------------------------------------------------------- view -> dense float32 copy
template <int DT>
__global__ void view_to_f32_kernel(const typename Elem<DT>::type* __restrict__ src, float* __restrict__ dst,
                                   int64_t pixels, int cstride, int coff, int c) {
    const int64_t total = pixels * c;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t px = i / c;
        const int ch = (int)(i - px * c);
        dst[i] = Elem<DT>::ld(src[px * cstride + coff + ch]);
    }
}

int view_to_f32_launch(int dtype, const void* src, float* dst, int64_t pixels, int cstride, int coff, int c,
                       hipStream_t stream) {
    const int64_t total = pixels * c;
    if (total == 0) return Y4_OK;
    const int blocks = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    switch (dtype) {
        case Y4_F32: hipLaunchKernelGGL(view_to_f32_kernel<Y4_F32>, dim3(blocks), dim3(256), 0, stream, (const float*)src, dst, pixels, cstride, coff, c); break;
        case Y4_BF16: hipLaunchKernelGGL(view_to_f32_kernel<Y4_BF16>, dim3(blocks), dim3(256), 0, stream, (const uint16_t*)src, dst, pixels, cstride, coff, c); break;
        case Y4_F16: hipLaunchKernelGGL(view_to_f32_kernel<Y4_F16>, dim3(blocks), dim3(256), 0, stream, (const _Float16*)src, dst, pixels, cstride, coff, c); break;
        default: set_error("view_to_f32: bad dtype %d", dtype); return Y4_EINVAL;
    }
    Y4_CHECK_HIP(hipGetLastError());
    return Y4_OK;
}

// dense float32 [pixels, c] -> float32 view (pad channels zeroed); used to inject raw heads for decode/NMS tests
__global__ void f32_to_view_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t pixels, int cstride,
                                   int c) {
    const int64_t total = pixels * cstride;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t px = i / cstride;
        const int ch = (int)(i - px * cstride);
        dst[i] = ch < c ? src[px * c + ch] : 0.f;
    }
}

int f32_to_view_launch(const float* src, float* dst, int64_t pixels, int cstride, int c, hipStream_t stream) {
    const int64_t total = pixels * cstride;
    if (total == 0) return Y4_OK;
    const int blocks = (int)((total + 255) / 256 > 16384 ? 16384 : (total + 255) / 256);
    hipLaunchKernelGGL(f32_to_view_kernel, dim3(blocks), dim3(256), 0, stream, src, dst, pixels, cstride, c);
    Y4_CHECK_HIP(hipGetLastError());
    return Y4_OK;
}

// per-channel BN fold: scale = gamma*rsqrt(var+eps), shift = beta - mean*scale  (Keras eps 1e-3), or the
// head convs' (1, bias).  rows of `bn` are Darknet order [beta, gamma, mean, var] (reference utils.py:28-31).
__global__ void fold_bn_kernel(const float* __restrict__ rec, float* __restrict__ scale, float* __restrict__ shift,
                               int cout, int cout_pad, int has_bn) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= cout_pad) return;
    float s = 1.f, h = 0.f;
    if (c < cout) {
        if (has_bn) {
            const float beta = rec[c], gamma = rec[cout + c], mean = rec[2 * cout + c], var = rec[3 * cout + c];
            s = gamma * (1.0f / sqrtf(var + 1e-3f));
            h = beta - mean * s;
        } else {
            h = rec[c];
        }
    } else {
        s = 0.f;
    }
    scale[c] = s;
    shift[c] = h;
}

int fold_bn_launch(const float* rec, float* scale, float* shift, int cout, int cout_pad, int has_bn, hipStream_t stream) {
    hipLaunchKernelGGL(fold_bn_kernel, dim3((cout_pad + 255) / 256), dim3(256), 0, stream, rec, scale, shift, cout,
                       cout_pad, has_bn);
    Y4_CHECK_HIP(hipGetLastError());
    return Y4_OK;
}

// ---- tuner aid: stream a region through the L2s (y4_autotune for latency schedules times every candidate on cold weights, the
// state a layer finds them in inside a real step: in no L2, at best in the Infinity Cache)
__global__ void l2_flush_kernel(const u32x4_t* __restrict__ p, size_t n16, u32x4_t* sink) {
    u32x4_t acc = {0u, 0u, 0u, 0u};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const u32x4_t v = p[i];
        acc[0] ^= v[0]; acc[1] ^= v[1]; acc[2] ^= v[2]; acc[3] ^= v[3];
    }
    if (acc[0] == 0x9e3779b9u && acc[1] == 0x7f4a7c15u && acc[2] == 0x12345678u) *sink = acc;       // (never: keeps the loads alive)
}
int l2_flush_launch(const void* p, size_t bytes, void* sink, hipStream_t stream) {
    hipLaunchKernelGGL(l2_flush_kernel, dim3(2048), dim3(256), 0, stream, (const u32x4_t*)p, bytes / 16, (u32x4_t*)sink);
    Y4_CHECK_HIP(hipGetLastError());
    return Y4_OK;
}

}  // namespace y4
