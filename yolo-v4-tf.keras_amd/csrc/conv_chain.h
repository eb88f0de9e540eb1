// conv_chain.h -- 1x1 convs chained onto the register tile of the conv that produces their input.
//
// In conv_igemm_kernel the accumulator layout is D[channel][pixel]; with ONE wave column (WN == 1) a wave holds every
// output channel of its pixels, and in the chunked layout (conv_common.h) chunk c of lane (q, g) -- 8 values packed to
// one 16-byte register quad -- holds channels 32c + 8g .. +7 of pixel q: exactly the B operand of
// v_mfma_f32_16x16x32 for k-step c of a following 1x1 conv in the NATURAL K order.  So a run
//     3x3 conv (+ residual Add)  ->  1x1 conv  ->  1x1 conv over Concatenate([that, route])        (CFG 1..3)
// (reference custom_layers.py:34-44 residual_block, :47-69 csp_block and the conv after it, :104/:109) executes in
// one kernel: the intermediate tensors never leave the registers, the concat partner is read from HBM straight into the
// same fragment layout, the tail weights are ordinary A fragments.  Each chained conv issues the same MFMAs on the same
// 16-bit inputs in the same order as its stand-alone kernel would (k-steps ascending over the 16-bit rounded
// activations), so the results are BIT-IDENTICAL to the unfused path (tested) -- a pure scheduling choice.
#pragma once
#include "conv_common.h"

namespace y4 {

// scale/shift of this lane's chunks c0 .. c0+NC-1 of the channel block starting at chw
template <int NC>
__device__ __forceinline__ void chain_load_affine(const float* scale, const float* shift, int chw, int fg, float* sc, float* sh) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int ch = chunk_channel(chw, c, fg);
#pragma unroll
        for (int h = 0; h < 8; h += 4) {
            const f32x4 s4 = *(const f32x4*)(scale + ch + h);
            const f32x4 h4 = *(const f32x4*)(shift + ch + h);
#pragma unroll
            for (int e = 0; e < 4; ++e) { sc[c * 8 + h + e] = s4[e]; sh[c * 8 + h + e] = h4[e]; }
        }
    }
}

// Shape of a chain (compile time, so that each gets its own register allocation):
//   CFG 1: head 64 ch -> tail 64          CFG 2: head 64 -> 64 -> (64 | partner 64) -> 64
//   CFG 3: head 64 -> 64 -> (64 | partner 64) -> 128
//   CFG 4: head 64 -> (64 | partner 64) -> 128      (no 64 -> 64 tail: the head IS the conv before the concat; ConvK::tail[1]
//          describes the conv over the concat, tail[0] is unused) -- the post conv of a CSP stage whose last residual block
//          runs in resblock_kernel, chained to the conv over the concat (custom_layers.py:66-69)
template <int CFG> struct ChainShape {
    static constexpr int HEAD_NREP = 4;
    static constexpr int T0_K = CFG == 4 ? 0 : 64;
    static constexpr int T0_COUT = CFG == 4 ? 0 : 64;
    static constexpr int T1_K = (CFG == 2 || CFG == 3 || CFG == 4) ? 128 : 0;
    static constexpr int T1_COUT = CFG == 2 ? 64 : ((CFG == 3 || CFG == 4) ? 128 : 0);
    static constexpr int T0_BYTES = T0_COUT * T0_K * 2, T1_BYTES = T1_COUT * T1_K * 2;
    static constexpr int LDS_BYTES = T0_BYTES + T1_BYTES;
};

// Tail weights (fragment order, pack_tail_kernel) -> LDS, issued at kernel start with buffer_load ... lds so the copies
// fly under the head conv's K loop; the K loop's last iteration waits for vmcnt(0) and takes a workgroup barrier, after
// which every wave may read them.
template <int CFG, int NWAVES>
__device__ __forceinline__ void chain_stage_weights(const ConvK& p, char* lds, int wave, int lane) {
    using S = ChainShape<CFG>;
    if constexpr (S::T0_BYTES != 0) {
        const __amdgpu_buffer_rsrc_t r0 = make_rsrc(p.tail[0].w, S::T0_BYTES);
        for (int u = wave; u < S::T0_BYTES / 1024; u += NWAVES)
            buffer_load16_lds(r0, lds + __builtin_amdgcn_readfirstlane(u * 1024), u * 1024 + lane * 16, 0);
    }
    if constexpr (S::T1_BYTES != 0) {
        const __amdgpu_buffer_rsrc_t r1 = make_rsrc(p.tail[1].w, S::T1_BYTES);
        for (int u = wave; u < S::T1_BYTES / 1024; u += NWAVES)
            buffer_load16_lds(r1, lds + S::T0_BYTES + __builtin_amdgcn_readfirstlane(u * 1024), u * 1024 + lane * 16, 0);
    }
}

// The chain's HBM inputs besides the head conv's own operands: the residual and the concat partner.  They depend on
// nothing, so the kernel issues these loads BEFORE the head's K loop (registers held across it) and their round trip
// is hidden under it instead of being paid at the epilogue.  Both are this lane's two chunks of its pixels.
template <int MREP> struct ChainPrefetch {
    u32x4 res[MREP][2], x2[MREP][2];
};
template <int DT, int MREP, int CFG>
__device__ __forceinline__ void chain_prefetch(const ConvK& p, ChainPrefetch<MREP>& pf, int mrow, int m_limit, int lane) {
    using T = typename Elem<DT>::type;
    const int fg = lane >> 4;
#pragma unroll
    for (int i = 0; i < MREP; ++i) {
        const int m = mrow + i * 16;
        const int64_t mm = m < m_limit ? m : 0;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int ch = chunk_channel(0, c, fg);
            if (p.res) pf.res[i][c] = *(const u32x4*)((const T*)p.res + mm * p.res_cstride + p.res_coff + ch);
            if constexpr (ChainShape<CFG>::T1_K != 0) {
                const ChainTail& t1 = p.tail[1];
                pf.x2[i][c] = *(const u32x4*)((const T*)t1.src2 + mm * t1.src2_cstride + t1.src2_coff + ch);
            }
        }
    }
}

// One chained 1x1 conv over KS k-steps of register chunks: acc[i][j] += W(s, j) * X(i, s).  `X(i, s)` is supplied by the
// caller's functor so that the sources (head chunks, previous tail's chunks, concat partner) need no copies.
template <int DT, int MREP, int NREP2, int KS, class XF>
__device__ __forceinline__ void chain_gemm(const char* lds_w, int lane, f32x4 (&acc)[MREP][NREP2], XF&& xf) {
#pragma unroll
    for (int i = 0; i < MREP; ++i)
#pragma unroll
        for (int j = 0; j < NREP2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const u32x4* const wf = (const u32x4*)lds_w + lane;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        u32x4 w[NREP2];
#pragma unroll
        for (int j = 0; j < NREP2; ++j) w[j] = wf[(s * NREP2 + j) * 64];
#pragma unroll
        for (int i = 0; i < MREP; ++i)
#pragma unroll
            for (int j = 0; j < NREP2; ++j) Mma<DT>::run(acc[i][j], w[j], xf(i, s));
    }
}

// BN + Mish of one pixel fragment's NREP2*16 channels -> packed chunks
template <int DT, int NREP2>
__device__ __forceinline__ void chain_bn_act_pack(const f32x4 (&acc)[NREP2], const float* sc, const float* sh, u32x4* out) {
    float v[NREP2 * 4];
#pragma unroll
    for (int j = 0; j < NREP2; ++j)
        bn_act4<true, Y4_ACT_MISH>(acc[j], sc + j * 4, sh + j * 4, v + j * 4);
#pragma unroll
    for (int c = 0; c < NREP2 / 2; ++c) Elem<DT>::store_chunk(&out[c], v + c * 8);
}

template <int DT, int NC>
__device__ __forceinline__ void chain_store(char* base, int64_t m, int cstride, int coff, int fg, const u32x4* chunks) {
    using T = typename Elem<DT>::type;
    T* op = (T*)base + m * cstride + coff;
#pragma unroll
    for (int c = 0; c < NC; ++c) *(u32x4*)(op + chunk_channel(0, c, fg)) = chunks[c];
}

// Epilogue of a chain head: BN + Mish (+ residual) into fragment registers, then the tails.
template <int DT, int MREP, int CFG>
__device__ __forceinline__ void chain_epilogue(const ConvK& p, const char* lds_w, f32x4 (&acc)[MREP][ChainShape<CFG>::HEAD_NREP],
                                               const ChainPrefetch<MREP>& pf, int mrow, int m_limit, int lane) {
    using E = Elem<DT>;
    using S = ChainShape<CFG>;
    constexpr int HN = S::HEAD_NREP, HC = HN / 2;           // head fragments / chunks per lane
    const int fg = lane >> 4;
    float sc[HC * 8], sh[HC * 8];
    chain_load_affine<HC>(p.scale, p.shift, 0, fg, sc, sh);
    u32x4 X[MREP][HC];
#pragma unroll
    for (int i = 0; i < MREP; ++i) {
        const int m = mrow + i * 16;
        float v[HC * 8];
#pragma unroll
        for (int j = 0; j < HN; ++j)
            bn_act4<true, Y4_ACT_MISH>(acc[i][j], sc + j * 4, sh + j * 4, v + j * 4);
        if (p.res) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float rv[8];
                E::load_chunk(&pf.res[i][c], rv);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[c * 8 + e] += rv[e];
            }
        }
#pragma unroll
        for (int c = 0; c < HC; ++c) E::store_chunk(&X[i][c], v + c * 8);
        if (p.store_x && m < m_limit) chain_store<DT, 2>(p.out, m, p.out_cstride, p.out_coff, fg, &X[i][0]);
    }
    if constexpr (S::T0_K == 0) {
        // ---- CFG 4: the conv over Concatenate([the head's 64 channels, partner's 64]) straight away
        constexpr int N1 = S::T1_COUT / 16;
        f32x4 a1[MREP][N1];
        chain_gemm<DT, MREP, N1, 4>(lds_w, lane, a1, [&](int i, int s) -> const u32x4& { return s < 2 ? X[i][s] : pf.x2[i][s - 2]; });
        float sc1[N1 * 4], sh1[N1 * 4];
        chain_load_affine<N1 / 2>(p.tail[1].scale, p.tail[1].shift, 0, fg, sc1, sh1);
#pragma unroll
        for (int i = 0; i < MREP; ++i) {
            u32x4 Z[N1 / 2];
            chain_bn_act_pack<DT, N1>(a1[i], sc1, sh1, Z);
            if (mrow + i * 16 < m_limit) chain_store<DT, N1 / 2>(p.fin, mrow + i * 16, p.fin_cstride, p.fin_coff, fg, Z);
        }
        return;
    }
    // ---- tail 0: the head's 64 channels -> 64
    constexpr int N0 = S::T0_COUT ? S::T0_COUT / 16 : 4, XOFF = 0;
    f32x4 a0[MREP][N0];
    chain_gemm<DT, MREP, N0, 2>(lds_w, lane, a0, [&](int i, int s) -> const u32x4& { return X[i][XOFF + s]; });
    float sc0[N0 * 4], sh0[N0 * 4];
    chain_load_affine<N0 / 2>(p.tail[0].scale, p.tail[0].shift, 0, fg, sc0, sh0);
    u32x4 Y[MREP][N0 / 2];
#pragma unroll
    for (int i = 0; i < MREP; ++i) chain_bn_act_pack<DT, N0>(a0[i], sc0, sh0, Y[i]);
    if constexpr (S::T1_K == 0) {
#pragma unroll
        for (int i = 0; i < MREP; ++i)
            if (mrow + i * 16 < m_limit) chain_store<DT, N0 / 2>(p.fin, mrow + i * 16, p.fin_cstride, p.fin_coff, fg, Y[i]);
    } else {
        // ---- tail 1: Concatenate([tail 0's 64 channels, partner's 64]) -> T1_COUT
        constexpr int N1 = S::T1_COUT / 16;
        f32x4 a1[MREP][N1];
        chain_gemm<DT, MREP, N1, 4>(lds_w + S::T0_BYTES, lane, a1,
                                    [&](int i, int s) -> const u32x4& { return s < 2 ? Y[i][s] : pf.x2[i][s - 2]; });
        float sc1[N1 * 4], sh1[N1 * 4];
        chain_load_affine<N1 / 2>(p.tail[1].scale, p.tail[1].shift, 0, fg, sc1, sh1);
#pragma unroll
        for (int i = 0; i < MREP; ++i) {
            u32x4 Z[N1 / 2];
            chain_bn_act_pack<DT, N1>(a1[i], sc1, sh1, Z);
            if (mrow + i * 16 < m_limit) chain_store<DT, N1 / 2>(p.fin, mrow + i * 16, p.fin_cstride, p.fin_coff, fg, Z);
        }
    }
}

}  // namespace y4
