// conv_common.h -- pieces shared by the implicit-GEMM conv kernels (conv_igemm.hip, conv_halo.hip).
#pragma once
#include <type_traits>

#include "kernels.h"

namespace y4 {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) short bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

// A 1x1 conv chained onto the producing conv's register tile (conv_chain.h)
struct ChainTail {
    const char* w;                 // fragment-ordered weights (pack_tail_weights)
    const float* scale;
    const float* shift;
    const char* src2;              // second 64-channel input slice (the concat partner), or null
    int src2_cstride, src2_coff;
    int cout;                      // 64 or 128
};

struct ConvK {
    const char* in;
    const char* wt;
    const float* scale;
    const float* shift;
    const char* res;
    char* out;
    const char* zero;
    int N, H, W, Cin, Ho, Wo, cout_store, M, K;
    int in_cstride, in_coff, out_cstride, out_coff, res_cstride, res_coff;
    int ksize, stride, pad, act, upsample, out_f32;
    int grid_m, grid_n;
    unsigned in_bytes, wt_bytes;   // buffer-descriptor extents (bounds-checked loads)
    unsigned out_bytes, out2_bytes, res_bytes;         // extents of the output / residual views; fast_epi: all below 2 GiB
    int fast_epi;
    unsigned fin_bytes, fin2_bytes;                    // the same for an LDS pair's tail (fast_tail)
    int fast_tail;
    float* obj;                    // float32 head (out_f32): objectness side array (kernels.h: ConvObjDesc), or null
    int obj_nf, obj_cpi, obj_base;
    FastDiv div_howo, div_wo;      // m -> (n, ho, wo) without integer division
    FastDiv div_gridn;
    char* out2;                    // channels >= split go to this view (fused CSP route + main-in pair)
    int out2_cstride, out2_coff, split;
    // chained 1x1 convs (ntail > 0): `out` is this conv's own output view, written only if store_x; the last
    // tail writes `fin`
    int ntail, store_x;
    ChainTail tail[2];
    char* fin;
    int fin_cstride, fin_coff;
    // LDS pair (pair != 0, conv_igemm.hip): this conv's output tile is also kept in LDS and a following 1x1 conv
    // (tail[0]: ordinary packed weights, `tail_w_bytes` long; output view `fin`) runs from there in the same kernel
    int pair, tail_act;
    unsigned tail_w_bytes;
    char* fin2;                    // the tail is a fused CSP pair: its rows >= tail_split go to this view
    int fin2_cstride, fin2_coff, tail_split;
    int tail_k, tail_panel0;       // the tail reads tail_k input channels starting at LDS panel tail_panel0 (64 channels each)
    int touch;                     // != 0: the workgroups of an XCD touch this channel tile's weights into their L2 at start (weight_touch)
    // split-K (ksplit > 1; plain launches only): split s of an output tile walks K-tiles [s nk / ksplit, (s + 1) nk / ksplit) and
    // leaves its fp32 accumulators in `part`; the split that arrives last at the tile's counter adds them in index order and
    // runs the epilogue (conv_igemm_kernel.h)
    int ksplit;
    float* part;                   // [ksplit][tiles][MREP * NREP][threads] float4
    int* split_cnt;                // [tiles], zero between launches
    // halo tiles (conv_halo_kernel.h): an output tile is a band of h_rows full image rows (h_bands bands per image; grid_m = N * h_bands),
    // its input a halo tile of (h_rows + 2) x h_pitch LDS rows per 64-channel chunk (h_pitch = W + 2 rounded up to 8)
    int h_rows, h_bands, h_pitch;
    int h_xmap;                    // != 0: XCD x runs channel tile x (weights larger than an L2: conv_halo_kernel.h)
    int h_abl;                     // experiments only (HALO_ABL: timing ablations of conv_halo_kernel, wrong results when != 0)
    FastDiv h_div_pitch, h_div_bands;
    // halo2 tiles (conv_halo2_kernel.h): the weights once more in MFMA-fragment order (pack_conv_frag32)
    const char* wfrag;
    unsigned wfrag_bytes;
};

template <int CPR> __device__ __forceinline__ int swz(int row) {
    return CPR == 8 ? (row & 7) : ((row >> 1) & 3);
}

template <int DT> struct Mma;
template <> struct Mma<Y4_F32> {
    static __device__ __forceinline__ void run(f32x4& acc, const u32x4& w, const u32x4& x) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(w[j]), __uint_as_float(x[j]), acc, 0, 0, 0);
    }
};
template <> struct Mma<Y4_BF16> {
    static __device__ __forceinline__ void run(f32x4& acc, const u32x4& w, const u32x4& x) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x),
                                                      acc, 0, 0, 0);
    }
};
template <> struct Mma<Y4_F16> {
    static __device__ __forceinline__ void run(f32x4& acc, const u32x4& w, const u32x4& x) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x),
                                                     acc, 0, 0, 0);
    }
};

// 32x32x16 MFMA (16-bit dtypes): A = 32 weight rows x 16 k (lane: row lane&31, k 8*(lane>>5) ..+7), B = 32 pixels x 16 k,
// D: lane (pixel lane&31, half lane>>5) holds rows 8*(i>>2) + 4*half + (i&3), i = 0..15
template <int DT> struct Mma32;
template <> struct Mma32<Y4_BF16> {
    static __device__ __forceinline__ void run(f32x16& acc, const u32x4& w, const u32x4& x) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, x), acc, 0, 0, 0);
    }
};
template <> struct Mma32<Y4_F16> {
    static __device__ __forceinline__ void run(f32x16& acc, const u32x4& w, const u32x4& x) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), acc, 0, 0, 0);
    }
};
template <> struct Mma32<Y4_F32> {     // (never instantiated for real: the fp32 path has no 32x32 tiles)
    static __device__ __forceinline__ void run(f32x16&, const u32x4&, const u32x4&) {}
};

// buffer_load_dwordx4 ... lds: 16 bytes per lane from (descriptor base + voffset + soffset) to LDS at
// (wave-uniform lds_dst + lane*16); lanes whose voffset is outside the descriptor's extent receive zeros.
// Kept in a NON-template function: inside a template the target builtin is checked at instantiation time
// for the host pass too, which silently drops the kernel's host stub.
__device__ __forceinline__ void buffer_load16_lds(__amdgpu_buffer_rsrc_t rsrc, char* lds_dst, int voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, voffset, soffset, 0, 0);
}
// 16 bytes per lane to / from registers through a descriptor: 32-bit byte offsets, out-of-range lanes read zeros / store nothing
__device__ __forceinline__ u32x4 buffer_load16(__amdgpu_buffer_rsrc_t rsrc, int voffset) {
    return __builtin_amdgcn_raw_buffer_load_b128(rsrc, voffset, 0, 0);
}
// AUX: cache policy bits of the instruction (gfx940+: 1 = sc0, 2 = nt, 16 = sc1)
template <int AUX = 0> __device__ __forceinline__ void buffer_store16(__amdgpu_buffer_rsrc_t rsrc, u32x4 v, int voffset) {
    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, voffset, 0, AUX);
}
// 16 bytes per lane at AGENT scope (relaxed atomics, dword by dword: the compiler emits the sc1 accesses and its own waits) -- coherent
// between the XCDs' L2s without a cache write-back / invalidate (the split-K partial sums, conv_igemm_kernel.h)
__device__ __forceinline__ void store16_agent(f32x4* ptr, f32x4 v) {
#pragma unroll
    for (int k = 0; k < 4; ++k) __hip_atomic_store((float*)ptr + k, v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ f32x4 load16_agent(const f32x4* ptr) {
    f32x4 v;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = __hip_atomic_load((const float*)ptr + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v;
}
// one dword per lane, same addressing: used to TOUCH a cache line (the data lands in an LDS scratch nobody reads)
__device__ __forceinline__ void buffer_load4_lds(__amdgpu_buffer_rsrc_t rsrc, char* lds_dst, int voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 4, voffset, soffset, 0, 0);
}
// Weight touch.  In a step a layer's weights are in no L2 when its kernel starts, and the workgroups of an XCD walk them in
// lock-step: every K-tile's weight rows are a cold miss for all of them at once, K-tile after K-tile, each miss longer than
// the one K-tile the staging ring looks ahead (measured, scripts/touch_probe.sh: a 3x3 256->256 @38^2 takes 60 us in a step
// against 50 in a hot loop, its 1x1 neighbours 19 against 13).  So the workgroups that run together on an XCD (blockIdx.x >> 3
// counts them; at most 32 CUs) share out the `bytes` of the block they are all about to stream and touch it once, one
// dword per 128-byte line, before their first stage: about one load instruction per wave, every miss in flight together,
// and the ring's loads then hit L2.  The dwords land in `scratch`: TOUCH_LDS bytes of LDS that belong to the touch alone
// (round 5; every wave of the workgroup writes the same 256 bytes, nobody ever reads them).  Until round 4 the scratch was the
// wave's own first staging piece, correct only as long as a wave's LDS-DMA loads LAND in issue order; nothing observable
// rests on that ordering any more.  No arithmetic is involved: results cannot change.
constexpr int TOUCH_LDS = 256;
__device__ __forceinline__ void weight_touch(__amdgpu_buffer_rsrc_t rs, char* scratch, int byte0, int bytes, int wave, int nwaves, int lane) {
    const int me = ((int)blockIdx.x >> 3) & 31;
    const int lines = bytes >> 7;
    for (int i = (me * nwaves + wave) * 64 + lane; i < lines; i += 32 * nwaves * 64)
        buffer_load4_lds(rs, scratch, byte0 + i * 128, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, bytes, 0x00020000);
}

// The same, and ALSO this wave's outstanding LDS reads (lgkmcnt): for a barrier behind which another wave's LDS-DMA may
// overwrite a slot this wave has issued (but not yet consumed) ds_reads from -- the barrier then proves "read to the end"
// by construction, not by timing (resblock.hip phase C).
template <int N> __device__ __forceinline__ void wait_vmcnt_lgkm_then_barrier() {
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(N) : "memory");
}
// the same for a count that is a compile-time constant only after unrolling (0 <= n <= MAXN)
template <int MAXN> __device__ __forceinline__ void wait_vmcnt_lgkm_then_barrier_n(int n) {
    if constexpr (MAXN >= 0) {
        if (n == MAXN) wait_vmcnt_lgkm_then_barrier<MAXN>();
        else wait_vmcnt_lgkm_then_barrier_n<MAXN - 1>(n);
    }
}
template <int N> __device__ __forceinline__ void wait_vmcnt_then_barrier() {
    // counted wait for this wave's own LDS-DMA loads, then the workgroup barrier; one asm statement with a
    // "memory" clobber so that neither the compiler's loads/stores nor its own waitcnt logic move across it
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// min(k, MAXK) younger stages of UNIT loads each may stay in flight (k: runtime, wave-uniform)
template <int MAXK, int UNIT> __device__ __forceinline__ void wait_vmcnt_then_barrier_k(int k) {
    if constexpr (MAXK <= 0) wait_vmcnt_then_barrier<0>();
    else {
        if (k >= MAXK) wait_vmcnt_then_barrier<MAXK * UNIT>();
        else wait_vmcnt_then_barrier_k<MAXK - 1, UNIT>(k);
    }
}


// ---- channel <-> lane layout of a wave's accumulator tile ("chunked"): within the wave's block of 16*NREP channels
// starting at `chw`, lane group g = lane>>4 owns, for c = 0 .. NREP/2-1, the 8 channels chw + (4c + g)*8 .. +7:
// fragment 2c holds the first four of them (rows g*4 + r), fragment 2c+1 the other four.  So
//   * every 16-byte NHWC store/load instruction of a wave covers 64 contiguous bytes per pixel (4 lane groups), and
//   * chunk c of a lane IS the B operand of v_mfma_*_16x16x32 for k-step c in the NATURAL K order
//     (channel 32c + 8g + e), which is what lets conv_chain.h feed a following 1x1 conv from registers with
//     bit-identical results.
// The assignment is applied as a row permutation when the weight tile is staged (conv_igemm.hip, b_off).
__device__ __forceinline__ int chunk_channel(int chw, int c, int fg) { return chw + (c * 4 + fg) * 8; }
// The same with GW lane groups per wave: 4 for the 16x16 MFMA tiles (16 pixels per fragment, lane group = lane >> 4), 2 for
// the 32x32x16 tiles (32 pixels per block, lane half = lane >> 5; a block's 16 accumulator values per lane are two chunks).
template <int GW> __device__ __forceinline__ int chunk_channel_g(int chw, int c, int fg) { return chw + (c * GW + fg) * 8; }

// ---- epilogue shared by the conv kernels: y = act(acc*scale + shift) (+ residual) -> NHWC slice store
// (optionally 2x2 replicated, optionally split over two output views).  `mrow` is this lane's output pixel index
// for fragment 0 (fragment i is 16 pixels further; 32 with GW = 2, see chunk_channel_g), pixels >= m_limit are not stored, `chw` is the first channel of
// the wave's block.  FULL (block-uniform): no per-row / per-chunk predicates at all.
// XL: the packed 16-bit tile is ALSO written to LDS at `xl` as Cout/64 consecutive [BM rows][128 B] panels (row =
// pixel `xrow` + 16 i of the block, chunk index XOR-swizzled by the row like a staged activation tile), i.e. in the
// layout the K loop reads its pixel operand from -- the input of a following 1x1 conv (LDS pair).
template <int DT, int MREP, int NREP, int ACT, bool FULL, bool XL = false, int GW = 4>
__device__ __forceinline__ void conv_epilogue_impl(const ConvK& p, f32x4 (&acc)[MREP][NREP], const float* sc,
                                                   const float* sh, int mrow, int m_limit, int chw, int fg,
                                                   char* xl = nullptr, int xrow = 0, int xpanel = 0) {
    using E = Elem<DT>;
    using T = typename E::type;
    constexpr int EPC = E::EPC;
    constexpr int NC = NREP / 2;
    constexpr bool FAST = (DT != Y4_F32);
    static_assert(NREP % 2 == 0, "chunked layout: fragments come in pairs");
    const int HoWo = p.Ho * p.Wo;
    // per chunk: its first channel, and where it goes (a chunk never straddles `split`: both are multiples of 8)
    int ch[NC], ooff[NC];
    bool second[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        ch[c] = chunk_channel_g<GW>(chw, c, fg);
        second[c] = p.split > 0 && ch[c] >= p.split;
        ooff[c] = second[c] ? p.out2_coff + ch[c] - p.split : p.out_coff + ch[c];
    }
#pragma unroll
    for (int i = 0; i < MREP; ++i) {
        const int m = mrow + i * (64 / GW);
        if (!FULL && m >= m_limit) continue;
        float v[NC * 8];
#pragma unroll
        for (int j = 0; j < NREP; ++j)                       // v[j*4 + r] = chunk (j>>1), element (j&1)*4 + r
            bn_act4<FAST, ACT>(acc[i][j], sc + j * 4, sh + j * 4, v + j * 4);
        if (p.res) {
            const T* rp = (const T*)p.res + (int64_t)m * p.res_cstride + p.res_coff;
#pragma unroll
            for (int c = 0; c < NC; ++c)
                if (FULL || ch[c] < p.cout_store) {
#pragma unroll
                    for (int e0 = 0; e0 < 8; e0 += EPC) {
                        float rv[EPC];
                        E::load_chunk(rp + ch[c] + e0, rv);
#pragma unroll
                        for (int e = 0; e < EPC; ++e) v[c * 8 + e0 + e] += rv[e];
                    }
                }
        }
        int64_t pix[4];
        int npix = 1;
        if (p.upsample) {
            const int n = (int)fastdiv((uint32_t)m, p.div_howo), rem = m - n * HoWo;
            const int ho = (int)fastdiv((uint32_t)rem, p.div_wo), wo = rem - ho * p.Wo;
            const int W2 = 2 * p.Wo;
            const int64_t base = ((int64_t)n * 2 * p.Ho + 2 * ho) * W2 + 2 * wo;
            pix[0] = base; pix[1] = base + 1; pix[2] = base + W2; pix[3] = base + W2 + 1;
            npix = 4;
        } else {
            pix[0] = m;
        }
        if (p.out_f32) {
            if (p.obj) {                                     // the cell's objectness logits, as stored, also to the side array
                const int n = (int)fastdiv((uint32_t)m, p.div_howo), rem = m - n * HoWo;
                float* const slot = p.obj + ((int64_t)n * p.obj_cpi + p.obj_base + rem) * 4;
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        const int e = a * p.obj_nf + 4 - ch[c];
                        if (e >= 0 && e < 8) {
                            float val = v[c * 8];
#pragma unroll
                            for (int q = 1; q < 8; ++q) val = e == q ? v[c * 8 + q] : val;
                            slot[a] = val;
                        }
                    }
            }
            for (int u = 0; u < npix; ++u) {
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    if (FULL || ch[c] < p.cout_store) {
                        float* op = (float*)(second[c] ? p.out2 : p.out) + pix[u] * (second[c] ? p.out2_cstride : p.out_cstride) + ooff[c];
                        Elem<Y4_F32>::store_chunk(op, v + c * 8);
                        Elem<Y4_F32>::store_chunk(op + 4, v + c * 8 + 4);
                    }
            }
        } else {
            u32x4 packed[NC * 8 / EPC];
#pragma unroll
            for (int k = 0; k < NC * 8; k += EPC) E::store_chunk(&packed[k / EPC], v + k);
            if constexpr (XL) {
                static_assert(!XL || EPC == 8, "LDS pair: 16-bit dtypes");
                static_assert(!XL || GW == 4, "LDS pair: 16x16 tiles");
                const int row = xrow + i * 16;
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    *(u32x4*)(xl + (ch[c] >> 6) * xpanel + row * 128 + ((((ch[c] & 63) >> 3) ^ (row & 7)) * 16)) = packed[c];
            }
            for (int u = 0; u < (XL ? 0 : npix); ++u) {     // XL: the tile goes out from LDS after the tail (pair_store_tile)
#pragma unroll
                for (int c = 0; c < NC; ++c)
                    if (FULL || ch[c] < p.cout_store) {
                        T* op = (T*)(second[c] ? p.out2 : p.out) + pix[u] * (second[c] ? p.out2_cstride : p.out_cstride) + ooff[c];
#pragma unroll
                        for (int e0 = 0; e0 < 8; e0 += EPC) *(u32x4*)(op + e0) = packed[(c * 8 + e0) / EPC];
                    }
            }
        }
    }
}

#ifndef Y4_FAST_EPI
#define Y4_FAST_EPI 1              // (0: kernel experiments, the general epilogue everywhere)
#endif
// The same for the case nearly every tile of a step is: a FULL tile (no row / channel predicates), 16-byte stores of the compute
// dtype, no 2x2 replication.  Addresses are 32-bit byte offsets into buffer descriptors of the views -- one multiply per pixel row,
// the chunks' offsets immediates -- where the general path above pays 64-bit pointer arithmetic and a view select per store (its
// ISA: ~200 VALU instructions per pixel row beside the ~80 of the activation), and the residual rows are loaded two rows ahead of
// their use instead of in front of it.  Same values, same order of operations on them: bit-identical.
// XL (an LDS pair's head): the packed tile goes to LDS in the K loop's pixel-operand layout instead (conv_epilogue_impl), nothing to
// memory here -- the row's swizzle term is the lane's (rows 16 apart share their low bits), so a chunk's LDS address is a per-lane base
// plus a compile-time row stride.
// LIM: the tile's pixel rows end at `m_limit` (a band of image rows that does not fill the MFMA tile, conv_halo_kernel.h): rows past it
// get an out-of-range offset -- their residual loads read zeros, their stores are dropped by the descriptor's bounds check.
template <int DT, int MREP, int NREP, int ACT, bool RES, int GW, bool XL = false, bool LIM = false>
__device__ __forceinline__ void conv_epilogue_fast(const ConvK& p, f32x4 (&acc)[MREP][NREP], const float* sc, const float* sh, int mrow,
                                                   int chw, int fg, char* xl = nullptr, int xrow = 0, int xpanel = 0, int m_limit = 0) {
    using E = Elem<DT>;
    constexpr int EPC = E::EPC, ES = 16 / EPC, NC = NREP / 2, SPC = 8 / EPC;      // SPC: 16-byte stores per 8-channel chunk
    constexpr bool FAST = (DT != Y4_F32);
    constexpr int ROWS = 64 / GW, CSTEP = GW * 8 * ES;                             // pixels between fragments, bytes between chunks
    const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(p.out, p.out_bytes), rs2 = make_rsrc(p.split > 0 ? p.out2 : p.out, p.split > 0 ? p.out2_bytes : p.out_bytes);
    // chunks [0, nfirst) go to `out`, the others to `out2` (wave-uniform: split and chw are multiples of 32)
    const int chw_u = __builtin_amdgcn_readfirstlane(chw);
    const int nfirst = p.split > 0 ? min(max((p.split - chw_u) / (GW * 8), 0), NC) : NC;
    const int lane1 = (p.out_coff + chw + fg * 8) * ES, lane2 = (p.out2_coff + chw - p.split + fg * 8) * ES;
    const int row1 = p.out_cstride * ES, row2 = p.out2_cstride * ES;
    char* xbase[NC];
    if constexpr (XL) {
        static_assert(!XL || (EPC == 8 && GW == 4), "LDS pair: 16-bit dtypes, 16x16 tiles");
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int ch = chunk_channel_g<GW>(chw, c, fg);
            xbase[c] = xl + (ch >> 6) * xpanel + xrow * 128 + ((((ch & 63) >> 3) ^ (xrow & 7)) * 16);
        }
    }
    u32x4 res[RES ? 3 : 1][NC * SPC];
    const __amdgpu_buffer_rsrc_t rsr = make_rsrc(RES ? p.res : p.out, RES ? p.res_bytes : p.out_bytes);
    const int laner = (p.res_coff + chw + fg * 8) * ES, rowr = p.res_cstride * ES;
    auto load_res = [&](int i) {
        const int o = (LIM && mrow + i * ROWS >= m_limit) ? (int)0x80000000 : (mrow + i * ROWS) * rowr + laner;
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int h = 0; h < SPC; ++h) res[i % 3][c * SPC + h] = buffer_load16(rsr, o + c * CSTEP + h * 16);
    };
    if constexpr (RES) {
        load_res(0);
        if (MREP > 1) load_res(1);
    }
#pragma unroll
    for (int i = 0; i < MREP; ++i) {
        if constexpr (RES) {
            if (i + 2 < MREP) load_res(i + 2);
        }
        float v[NC * 8];
#pragma unroll
        for (int j = 0; j < NREP; ++j) bn_act4<FAST, ACT>(acc[i][j], sc + j * 4, sh + j * 4, v + j * 4);
        if constexpr (RES) {
#pragma unroll
            for (int k = 0; k < NC * SPC; ++k) {
                float rv[EPC];
                E::load_chunk(&res[i % 3][k], rv);
#pragma unroll
                for (int e = 0; e < EPC; ++e) v[k * EPC + e] += rv[e];
            }
        }
        const int m = mrow + i * ROWS;
        const bool dead = LIM && m >= m_limit;
        const int o1 = dead ? (int)0x80000000 : m * row1 + lane1, o2 = dead ? (int)0x80000000 : m * row2 + lane2;
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int h = 0; h < SPC; ++h) {
                u32x4 pk;
                E::store_chunk(&pk, v + c * 8 + h * EPC);
                if constexpr (XL) *(u32x4*)(xbase[c] + i * 16 * 128) = pk;
                else if (c < nfirst) buffer_store16(rs1, pk, o1 + c * CSTEP + h * 16);
                else buffer_store16(rs2, pk, o2 + c * CSTEP + h * 16);
            }
    }
}

// the folded BatchNorm scale / shift of this lane's channels (a wave's block of 16 NREP channels from `chw`)
template <int NREP, int GW = 4>
__device__ __forceinline__ void conv_epilogue_tables(const ConvK& p, int chw, int fg, float (&sc)[NREP * 4], float (&sh)[NREP * 4]) {
    constexpr int NC = NREP / 2;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int ch = chunk_channel_g<GW>(chw, c, fg);
#pragma unroll
        for (int h = 0; h < 8; h += 4) {
            const f32x4 s4 = *(const f32x4*)(p.scale + ch + h);
            const f32x4 h4 = *(const f32x4*)(p.shift + ch + h);
#pragma unroll
            for (int e = 0; e < 4; ++e) { sc[c * 8 + h + e] = s4[e]; sh[c * 8 + h + e] = h4[e]; }
        }
    }
}

template <int DT, int MREP, int NREP, bool XL = false, int GW = 4>
__device__ __forceinline__ void conv_epilogue_with(const ConvK& p, f32x4 (&acc)[MREP][NREP], const float (&sc)[NREP * 4], const float (&sh)[NREP * 4],
                                                   int mrow, int m_limit, int chw, int fg, bool full, char* xl = nullptr, int xrow = 0,
                                                   int xpanel = 0) {
    // the activation and the mask mode are compile-time inside; one uniform switch outside the pixel loop
    if (Y4_FAST_EPI && GW == 4 && full && p.fast_epi && p.act != Y4_ACT_LINEAR && (p.act == Y4_ACT_MISH || !p.res)) {
        if (p.act == Y4_ACT_LEAKY) conv_epilogue_fast<DT, MREP, NREP, Y4_ACT_LEAKY, false, GW, XL>(p, acc, sc, sh, mrow, chw, fg, xl, xrow, xpanel);
        else if (p.res) conv_epilogue_fast<DT, MREP, NREP, Y4_ACT_MISH, true, GW, XL>(p, acc, sc, sh, mrow, chw, fg, xl, xrow, xpanel);
        else conv_epilogue_fast<DT, MREP, NREP, Y4_ACT_MISH, false, GW, XL>(p, acc, sc, sh, mrow, chw, fg, xl, xrow, xpanel);
        return;
    }
    if (p.act == Y4_ACT_MISH) {
        if (full) conv_epilogue_impl<DT, MREP, NREP, Y4_ACT_MISH, true, XL, GW>(p, acc, sc, sh, mrow, m_limit, chw, fg, xl, xrow, xpanel);
        else conv_epilogue_impl<DT, MREP, NREP, Y4_ACT_MISH, false, XL, GW>(p, acc, sc, sh, mrow, m_limit, chw, fg, xl, xrow, xpanel);
    } else if (p.act == Y4_ACT_LEAKY) {
        if (full) conv_epilogue_impl<DT, MREP, NREP, Y4_ACT_LEAKY, true, XL, GW>(p, acc, sc, sh, mrow, m_limit, chw, fg, xl, xrow, xpanel);
        else conv_epilogue_impl<DT, MREP, NREP, Y4_ACT_LEAKY, false, XL, GW>(p, acc, sc, sh, mrow, m_limit, chw, fg, xl, xrow, xpanel);
    } else {
        if (full) conv_epilogue_impl<DT, MREP, NREP, Y4_ACT_LINEAR, true, XL, GW>(p, acc, sc, sh, mrow, m_limit, chw, fg, xl, xrow, xpanel);
        else conv_epilogue_impl<DT, MREP, NREP, Y4_ACT_LINEAR, false, XL, GW>(p, acc, sc, sh, mrow, m_limit, chw, fg, xl, xrow, xpanel);
    }
}

template <int DT, int MREP, int NREP, bool XL = false, int GW = 4>
__device__ __forceinline__ void conv_epilogue(const ConvK& p, f32x4 (&acc)[MREP][NREP], int mrow, int m_limit, int chw,
                                              int fg, bool full, char* xl = nullptr, int xrow = 0, int xpanel = 0) {
    float sc[NREP * 4], sh[NREP * 4];
    conv_epilogue_tables<NREP, GW>(p, chw, fg, sc, sh);
    conv_epilogue_with<DT, MREP, NREP, XL, GW>(p, acc, sc, sh, mrow, m_limit, chw, fg, full, xl, xrow, xpanel);
}

// A tile whose channel block is full but whose pixel rows end at m_limit (halo tiles): the fast epilogue with dropped rows where it
// applies (16-byte stores of the compute dtype, Mish / LeakyReLU), else the general one.  Same arithmetic either way.
// GW = 4: 16 x 16 fragments (lane group = lane >> 4); GW = 2: the 32 x 32 blocks of conv_halo2_kernel.h (lane half = lane >> 5).
template <int DT, int MREP, int NREP, int GW>
__device__ __forceinline__ void conv_epilogue_rows_g(const ConvK& p, f32x4 (&acc)[MREP][NREP], int mrow, int m_limit, int chw, int fg, bool ch_full) {
    float sc[NREP * 4], sh[NREP * 4];
    conv_epilogue_tables<NREP, GW>(p, chw, fg, sc, sh);
    if (Y4_FAST_EPI && ch_full && p.fast_epi && p.act != Y4_ACT_LINEAR && (p.act == Y4_ACT_MISH || !p.res)) {
        if (p.act == Y4_ACT_LEAKY) conv_epilogue_fast<DT, MREP, NREP, Y4_ACT_LEAKY, false, GW, false, true>(p, acc, sc, sh, mrow, chw, fg, nullptr, 0, 0, m_limit);
        else if (p.res) conv_epilogue_fast<DT, MREP, NREP, Y4_ACT_MISH, true, GW, false, true>(p, acc, sc, sh, mrow, chw, fg, nullptr, 0, 0, m_limit);
        else conv_epilogue_fast<DT, MREP, NREP, Y4_ACT_MISH, false, GW, false, true>(p, acc, sc, sh, mrow, chw, fg, nullptr, 0, 0, m_limit);
        return;
    }
    conv_epilogue_with<DT, MREP, NREP, false, GW>(p, acc, sc, sh, mrow, m_limit, chw, fg, false);
}
template <int DT, int MREP, int NREP>
__device__ __forceinline__ void conv_epilogue_rows(const ConvK& p, f32x4 (&acc)[MREP][NREP], int mrow, int m_limit, int chw, int fg, bool ch_full) {
    conv_epilogue_rows_g<DT, MREP, NREP, 4>(p, acc, mrow, m_limit, chw, fg, ch_full);
}

// LDS pair: the head conv's tile, kept in LDS by the XL epilogue, goes to its HBM view(s) (lane re-reads the chunks it
// wrote; a fused CSP pair as head splits them over out / out2 like the ordinary epilogue).
template <int DT, int MREP, int NREP>
__device__ __forceinline__ void pair_store_tile(const ConvK& p, const char* xl, int xrow, int xpanel, int mrow, int m_limit,
                                                int chw, int fg) {
    using T = typename Elem<DT>::type;
    constexpr int NC = NREP / 2;
#pragma unroll
    for (int i = 0; i < MREP; ++i) {
        const int m = mrow + i * 16, row = xrow + i * 16;
        if (m >= m_limit) continue;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int ch = chunk_channel(chw, c, fg);
            const bool second = p.split > 0 && ch >= p.split;
            T* op = second ? (T*)p.out2 + (int64_t)m * p.out2_cstride + p.out2_coff + ch - p.split
                           : (T*)p.out + (int64_t)m * p.out_cstride + p.out_coff + ch;
            *(u32x4*)op = *(const u32x4*)(xl + (ch >> 6) * xpanel + row * 128 + ((((ch & 63) >> 3) ^ (row & 7)) * 16));
        }
    }
}

}  // namespace y4
