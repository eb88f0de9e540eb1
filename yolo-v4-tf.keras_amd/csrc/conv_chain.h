// conv_chain.h -- 1x1 convs chained onto the register tile of the conv that produces their input.
//
// In conv_igemm_kernel the accumulator layout is D[channel][pixel] and, with ONE wave column (WN == 1, BN == 64 ==
// cout), lane (q = lane & 15, g = lane >> 4) ends the conv holding channels 16g .. 16g+15 of pixels q + 16i.  Packed
// to 16 bits those values ARE the B operand of v_mfma_f32_16x16x32 for a following 1x1 conv, provided its K order is
// permuted to "k-step s, lane group g, element e  <->  input channel 16g + 8s + e" -- a permutation applied once to
// the 1x1 conv's weights when they are packed (pack_tail_kernel).  So a run
//     3x3 conv (+ residual Add)  ->  1x1 conv  ->  1x1 conv over Concatenate([that, route])
// (reference custom_layers.py:41-44 residual_block tail, :66-69 csp_block tail, :104/:109 the conv after it) executes
// in one kernel: the intermediate tensors never leave the registers, the concat partner (`src2`) is read from HBM
// straight into the same fragment layout, and only the last conv's output is stored.  Same fp32-accumulated
// products as the separate kernels, summed in a different order (not bitwise equal to them).
#pragma once
#include "conv_common.h"

namespace y4 {

// scale/shift of this lane's CPL consecutive channels starting at chb
template <int CPL>
__device__ __forceinline__ void chain_load_affine(const float* scale, const float* shift, int chb, float* sc, float* sh) {
#pragma unroll
    for (int c = 0; c < CPL; c += 4) {
        const f32x4 s4 = *(const f32x4*)(scale + chb + c);
        const f32x4 h4 = *(const f32x4*)(shift + chb + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) { sc[c + e] = s4[e]; sh[c + e] = h4[e]; }
    }
}

// One chained 1x1 conv: X (64 channels in registers) [+ X2 (the concat partner's 64 channels, HAS2)] -> NREP2*16 channels.
// LAST: store to p.fin; otherwise the result (64 channels) replaces X.
template <int DT, int MREP, int NREP2, bool LAST, bool HAS2>
__device__ __forceinline__ void chain_step(const ConvK& p, const ChainTail& t, const char* lds_w, u32x4 (&X)[MREP][2],
                                           const u32x4 (&X2)[MREP][2], int mrow, int m_limit, int lane) {
    using E = Elem<DT>;
    using T = typename E::type;
    constexpr int CPL2 = 4 * NREP2;
    const int fg = lane >> 4;
    f32x4 acc[MREP][NREP2];
#pragma unroll
    for (int i = 0; i < MREP; ++i)
#pragma unroll
        for (int j = 0; j < NREP2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const u32x4* const wf = (const u32x4*)lds_w + lane;      // fragment-ordered weights, staged by chain_stage_weights
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        u32x4 w[NREP2];
#pragma unroll
        for (int j = 0; j < NREP2; ++j) w[j] = wf[(s * NREP2 + j) * 64];
#pragma unroll
        for (int i = 0; i < MREP; ++i)
#pragma unroll
            for (int j = 0; j < NREP2; ++j) Mma<DT>::run(acc[i][j], w[j], X[i][s]);
    }
    if constexpr (HAS2) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            u32x4 w[NREP2];
#pragma unroll
            for (int j = 0; j < NREP2; ++j) w[j] = wf[((2 + s) * NREP2 + j) * 64];
#pragma unroll
            for (int i = 0; i < MREP; ++i)
#pragma unroll
                for (int j = 0; j < NREP2; ++j) Mma<DT>::run(acc[i][j], w[j], X2[i][s]);
        }
    }
    const int chb = fg * CPL2;
    float sc[CPL2], sh[CPL2];
    chain_load_affine<CPL2>(t.scale, t.shift, chb, sc, sh);
#pragma unroll
    for (int i = 0; i < MREP; ++i) {
        float v[CPL2];
#pragma unroll
        for (int j = 0; j < NREP2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[j * 4 + r] = apply_act_t<true, Y4_ACT_MISH>(fmaf(acc[i][j][r], sc[j * 4 + r], sh[j * 4 + r]));
        if constexpr (LAST) {
            const int m = mrow + i * 16;
            if (m < m_limit) {
                T* op = (T*)p.fin + (int64_t)m * p.fin_cstride + p.fin_coff + chb;
#pragma unroll
                for (int c = 0; c < CPL2; c += 8) {
                    u32x4 pk;
                    E::store_chunk(&pk, v + c);
                    *(u32x4*)(op + c) = pk;
                }
            }
        } else {
            static_assert(LAST || NREP2 == 4, "an inner chained conv has 64 output channels");
            E::store_chunk(&X[i][0], v);
            E::store_chunk(&X[i][1], v + 8);
        }
    }
}

// Bytes of fragment-ordered tail weights per chain shape (CFG as in chain_epilogue), and their LDS staging: issued
// at kernel start with buffer_load ... lds, so the copies fly under the head conv's K loop; the K loop's last
// iteration waits for vmcnt(0) and takes a workgroup barrier, after which every wave may read them.
template <int CFG> struct ChainLds {
    static constexpr int T0 = 64 * 64 * 2;
    static constexpr int T1 = CFG == 1 ? 0 : (CFG == 2 ? 64 * 128 * 2 : 128 * 128 * 2);
    static constexpr int BYTES = T0 + T1;
};
template <int CFG, int NWAVES>
__device__ __forceinline__ void chain_stage_weights(const ConvK& p, char* lds, int wave, int lane) {
    using L = ChainLds<CFG>;
    const __amdgpu_buffer_rsrc_t r0 = make_rsrc(p.tail[0].w, L::T0);
    for (int u = wave; u < L::T0 / 1024; u += NWAVES)
        buffer_load16_lds(r0, lds + __builtin_amdgcn_readfirstlane(u * 1024), u * 1024 + lane * 16, 0);
    if constexpr (CFG != 1) {
        const __amdgpu_buffer_rsrc_t r1 = make_rsrc(p.tail[1].w, L::T1);
        for (int u = wave; u < L::T1 / 1024; u += NWAVES)
            buffer_load16_lds(r1, lds + L::T0 + __builtin_amdgcn_readfirstlane(u * 1024), u * 1024 + lane * 16, 0);
    }
}

// The chain's HBM inputs besides the head conv's own operands: the residual and the concat partner.  They depend on
// nothing, so the kernel issues these loads BEFORE the head's K loop (32 VGPRs held across it) and their round trip
// is hidden under it instead of being paid at the epilogue.  Both are this lane's 16 channels of its pixels.
template <int MREP> struct ChainPrefetch {
    u32x4 res[MREP][2], x2[MREP][2];
};
template <int DT, int MREP, int CFG>
__device__ __forceinline__ void chain_prefetch(const ConvK& p, ChainPrefetch<MREP>& pf, int mrow, int m_limit, int lane) {
    using T = typename Elem<DT>::type;
    const int chb = (lane >> 4) * 16;
#pragma unroll
    for (int i = 0; i < MREP; ++i) {
        const int m = mrow + i * 16;
        const int64_t mm = m < m_limit ? m : 0;
        if (p.res) {
            const T* rp = (const T*)p.res + mm * p.res_cstride + p.res_coff + chb;
            pf.res[i][0] = *(const u32x4*)rp;
            pf.res[i][1] = *(const u32x4*)(rp + 8);
        }
        if constexpr (CFG != 1) {
            const ChainTail& t1 = p.tail[1];
            const T* sp = (const T*)t1.src2 + mm * t1.src2_cstride + t1.src2_coff + chb;
            pf.x2[i][0] = *(const u32x4*)sp;
            pf.x2[i][1] = *(const u32x4*)(sp + 8);
        }
    }
}

// Epilogue of a chain head (NREP == 4, one wave column): BN + Mish (+ residual) into fragment registers, then the tails.
template <int DT, int MREP, int CFG>
__device__ __forceinline__ void chain_epilogue(const ConvK& p, const char* lds_w, f32x4 (&acc)[MREP][4],
                                               const ChainPrefetch<MREP>& pf, int mrow, int m_limit, int lane) {
    using E = Elem<DT>;
    using T = typename E::type;
    const int fg = lane >> 4, chb = fg * 16;
    float sc[16], sh[16];
    chain_load_affine<16>(p.scale, p.shift, chb, sc, sh);
    u32x4 X[MREP][2];
#pragma unroll
    for (int i = 0; i < MREP; ++i) {
        const int m = mrow + i * 16;
        float v[16];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[j * 4 + r] = apply_act_t<true, Y4_ACT_MISH>(fmaf(acc[i][j][r], sc[j * 4 + r], sh[j * 4 + r]));
        if (p.res) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                float rv[8];
                E::load_chunk(&pf.res[i][c], rv);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[c * 8 + e] += rv[e];
            }
        }
        E::store_chunk(&X[i][0], v);
        E::store_chunk(&X[i][1], v + 8);
        if (p.store_x && m < m_limit) {
            T* op = (T*)p.out + (int64_t)m * p.out_cstride + p.out_coff + chb;
            *(u32x4*)op = X[i][0];
            *(u32x4*)(op + 8) = X[i][1];
        }
    }
    // CFG (compile time, so that each shape gets its own register allocation): 1 = one 64-channel tail,
    // 2 = two 64-channel tails, 3 = a 64- then a 128-channel tail
    if constexpr (CFG == 1) {
        chain_step<DT, MREP, 4, true, false>(p, p.tail[0], lds_w, X, pf.x2, mrow, m_limit, lane);
    } else {
        chain_step<DT, MREP, 4, false, false>(p, p.tail[0], lds_w, X, pf.x2, mrow, m_limit, lane);
        chain_step<DT, MREP, CFG == 2 ? 4 : 8, true, true>(p, p.tail[1], lds_w + ChainLds<CFG>::T0, X, pf.x2, mrow, m_limit, lane);
    }
}

// ---- split-head chain (CFG 4: 32-channel tail, CFG 5: 64-channel tail) ---------------------------------------------
// Head = the fused CSP pair (route conv | main-in conv, one 1x1 GEMM with 128 output rows, custom_layers.py:58-60);
// tail = the 1x1 conv that reads the main-in half (the residual block's first conv, :36, or csp's bottleneck conv).
// One wave column over all 128 rows: lane group g holds fused rows 32g .. 32g+31, i.e. groups 0,1 the route
// channels and groups 2,3 the 64 main-in channels.  The tail's K is laid over all four groups (4 k-steps of 32) with
// ZERO weights in the route groups' slots (pack_tail_split_kernel) -- twice the MFMAs of a dense K = 64, on a GEMM
// that is 3 % of the head's work, and no cross-lane traffic.  Both head halves are stored (they have later readers).
template <int CFG> struct ChainSplitLds {
    static constexpr int NREP2 = CFG == 4 ? 2 : 4;
    static constexpr int BYTES = 4 * NREP2 * 1024;
};
template <int CFG, int NWAVES>
__device__ __forceinline__ void chain_split_stage_weights(const ConvK& p, char* lds, int wave, int lane) {
    const __amdgpu_buffer_rsrc_t r0 = make_rsrc(p.tail[0].w, ChainSplitLds<CFG>::BYTES);
    for (int u = wave; u < ChainSplitLds<CFG>::BYTES / 1024; u += NWAVES)
        buffer_load16_lds(r0, lds + __builtin_amdgcn_readfirstlane(u * 1024), u * 1024 + lane * 16, 0);
}

template <int DT, int MREP, int CFG>
__device__ __forceinline__ void chain_split_epilogue(const ConvK& p, const char* lds_w, f32x4 (&acc)[MREP][8], int mrow,
                                                     int m_limit, int lane) {
    using E = Elem<DT>;
    using T = typename E::type;
    constexpr int NREP2 = ChainSplitLds<CFG>::NREP2, CPL2 = 4 * NREP2;
    const int fg = lane >> 4, chb = fg * 32;
    const bool second = chb >= p.split;                 // this lane's rows belong to the main-in conv -> out2
    T* const obase = (T*)(second ? p.out2 : p.out) + (second ? p.out2_coff + chb - p.split : p.out_coff + chb);
    const int ocs = second ? p.out2_cstride : p.out_cstride;
    float sc[32], sh[32];
    chain_load_affine<32>(p.scale, p.shift, chb, sc, sh);
    u32x4 X[MREP][4];
#pragma unroll
    for (int i = 0; i < MREP; ++i) {
        const int m = mrow + i * 16;
        float v[32];
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[j * 4 + r] = apply_act_t<true, Y4_ACT_MISH>(fmaf(acc[i][j][r], sc[j * 4 + r], sh[j * 4 + r]));
#pragma unroll
        for (int c = 0; c < 4; ++c) E::store_chunk(&X[i][c], v + c * 8);
        if (m < m_limit) {
            T* op = obase + (int64_t)m * ocs;
#pragma unroll
            for (int c = 0; c < 4; ++c) *(u32x4*)(op + c * 8) = X[i][c];
        }
    }
    f32x4 acc2[MREP][NREP2];
#pragma unroll
    for (int i = 0; i < MREP; ++i)
#pragma unroll
        for (int j = 0; j < NREP2; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const u32x4* const wf = (const u32x4*)lds_w + lane;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        u32x4 w[NREP2];
#pragma unroll
        for (int j = 0; j < NREP2; ++j) w[j] = wf[(s * NREP2 + j) * 64];
#pragma unroll
        for (int i = 0; i < MREP; ++i)
#pragma unroll
            for (int j = 0; j < NREP2; ++j) Mma<DT>::run(acc2[i][j], w[j], X[i][s]);
    }
    const ChainTail& t = p.tail[0];
    const int chb2 = fg * CPL2;
    float sc2[CPL2], sh2[CPL2];
    chain_load_affine<CPL2>(t.scale, t.shift, chb2, sc2, sh2);
#pragma unroll
    for (int i = 0; i < MREP; ++i) {
        const int m = mrow + i * 16;
        float v[CPL2];
#pragma unroll
        for (int j = 0; j < NREP2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[j * 4 + r] = apply_act_t<true, Y4_ACT_MISH>(fmaf(acc2[i][j][r], sc2[j * 4 + r], sh2[j * 4 + r]));
        if (m < m_limit) {
            T* op = (T*)p.fin + (int64_t)m * p.fin_cstride + p.fin_coff + chb2;
#pragma unroll
            for (int c = 0; c < CPL2; c += 8) {
                u32x4 pk;
                E::store_chunk(&pk, v + c);
                *(u32x4*)(op + c) = pk;
            }
        }
    }
}

}  // namespace y4
