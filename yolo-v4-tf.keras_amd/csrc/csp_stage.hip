// csp_stage.hip -- the whole first CSP stage of cspdarknet53 as ONE spatially tiled, persistent kernel (16-bit dtypes):
//   route = conv2(x)  main = conv3(x)                    1x1 64->64, Mish      reference custom_layers.py:58-60
//   t     = conv4(main)                                  1x1 64->32, Mish      :36 (residual_block, bottleneck)
//   main  = main + conv5(t)                              3x3 32->64, Mish, Add :37-44
//   main  = conv6(main)                                  1x1 64->64, Mish      :66
//   y     = conv7(Concatenate([main, route]))            1x1 128->64, Mish     :68, :105
// with x = conv 1's output (stem_down.hip) and y = conv 7's output the ONLY tensors that touch HBM.  Unfused, this stage
// moved ~3.8 GB per 32 images at 608x608 for 9 % of the FLOPs (six tensors of 190-760 MB written and read back).
//
// Decomposition.  A workgroup (8 waves) owns a 16x16-pixel output tile at a time and loops over tiles (persistent: the
// 80 KB of weights are staged into LDS once per workgroup).  Per tile:
//   0. the 18x18-pixel halo'd tile of x arrives in LDS by LDS-DMA (issued one tile ahead; out-of-image pixels read as 0)
//   A. per pixel fragment (16 pixels x all channels per wave, so that every 1x1 conv chains from REGISTERS, see
//      conv_chain.h): conv3 -> conv4 -> LDS tile T4 (zeroed outside the image: it is conv5's zero padding), for the 16
//      inner rows also conv2; conv2's and conv3's packed outputs stay in registers (route / residual).
//      The 1-pixel halo ring (row 0, row 17, columns 0 and 17 = 5 more fragments) is recomputed, 1.27x on conv3/conv4.
//   C. conv5 as 9 taps over T4 (+ residual from registers) -> conv6 -> conv7 over [conv6 | route], all from registers,
//      then 16-byte NHWC stores.
// Every conv issues the same MFMAs (v_mfma_f32_16x16x32, k-steps ascending, natural channel order per k-step) on the
// same 16-bit rounded inputs as its stand-alone conv_igemm kernel, and the same fp32 epilogue: outputs are
// BIT-IDENTICAL to the unfused path (tests/test_gpu_forward.py::test_stage_fusion_is_bit_identical).
// What bounds it: VALU, not MFMA or HBM -- Mish costs ~8 VALU ops (2 transcendental) per element and the stage has
// 352 Mish channels per pixel (LABNOTES.md section 4.1b).
#include "conv_chain.h"

// experiment switches (scripts/build_variant.sh): CS_ABL 1 = LeakyReLU instead of Mish (VALU ablation), 2 = no output
// stores, 3 = the input tile is loaded once only; CS_VMCNT = stores that may stay in flight across the tile boundary
#ifndef CS_ABL
#define CS_ABL 0
#endif
#ifndef CS_PRIO
// 1: "leapfrog" priorities -- a wave lowers its s_setprio level as it moves through the segments of a barrier interval, so the wave
// of a SIMD that is BEHIND outranks its partner (the hardware's own tie-break is age: the older wave of a SIMD runs unimpeded and
// the younger one gets the leftover VALU / transcendental slots, then finishes alone at a single wave's issue rate).
// 2: waves 4..7 at priority 1 throughout (the mirror image of the default).
#define CS_PRIO 1
#endif
#ifndef CS_LX
#define CS_LX 1     // 1: the halo tile's per-lane offsets and border flags are computed once per workgroup, not once per tile
#endif
#ifndef CS_EXTRA
#define CS_EXTRA 1  // 1: the five halo-ring fragments on waves 0..3 (wave 0 takes two) instead of waves 0..4
#endif
#ifndef CS_VMCNT
// 0: the tile boundary drains everything.  4 would let the previous tile's 4 output stores stay in flight (measured: 583 ->
// 578 us), but it relies on stores and LDS-DMA loads retiring in issue order RELATIVE TO EACH OTHER on one vmcnt counter,
// which gfx9-family hardware does not promise (LLVM treats mixed pending loads / stores as out of order): not worth 1 %.
#define CS_VMCNT 0
#endif

namespace y4 {

#ifdef CS_TRACE
// In-kernel phase trace (kernel experiments only; scripts/stage_trace.py): workgroup CS_TR_WG records s_memtime per wave at
// fixed points of its tiles CS_TR_T0 .. +3 into cs_trace_buf[tile][wave][point]; y4_cs_trace_read() copies it out.
__device__ unsigned long long cs_trace_buf[4 * 8 * 16];
#define CS_TR_WG 8
#define CS_TR_T0 3
#define CS_POINT(P)                                                                                         \
    do {                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        if (tr_on) asm volatile("s_memtime %0" : "=s"(tr_t[P]));                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
    } while (0)
#else
#define CS_POINT(P)
#endif

constexpr int CS_ACT = CS_ABL == 1 ? Y4_ACT_LEAKY : Y4_ACT_MISH;
#if CS_PRIO == 1
#define CS_SEG(K) do { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(K); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define CS_SEG(K)
#endif

// ---- LDS map (bytes).  Weight fragments are "fragment ordered" (pack_frag_kernel): [(kstep*NREP + j)*64 + lane][8].
constexpr int CS_W3 = 0;                        // 64 x 64      : 2 k-steps x 4 fragments x 1 KB
constexpr int CS_W4 = CS_W3 + 8 * 1024;         // 32 x 64      : 2 x 2
constexpr int CS_W2 = CS_W4 + 4 * 1024;         // 64 x 64
constexpr int CS_W5 = CS_W2 + 8 * 1024;         // 64 x (9*32)  : 9 x 4
constexpr int CS_W6 = CS_W5 + 36 * 1024;        // 64 x 64
constexpr int CS_W7 = CS_W6 + 8 * 1024;         // 64 x 128     : 4 x 4
constexpr int CS_WBYTES = CS_W7 + 16 * 1024;    // 80 KB
// affine (float scale[c], shift[c] per conv), in this order
constexpr int CS_A3 = 0, CS_A4 = 128, CS_A2 = 192, CS_A5 = 320, CS_A6 = 448, CS_A7 = 576, CS_AFLOATS = 704;
constexpr int CS_AFF = CS_WBYTES;               // 2816 B, padded to 3 KB
constexpr int CS_X1 = CS_AFF + 3 * 1024;        // [336 rows][128 B]: halo'd tile of x, row = hy*18 + hx (324 used)
constexpr int CS_HROWS = 336;
constexpr int CS_X4 = CS_X1 + CS_HROWS * 128;   // [336 rows][64 B]: conv4's output on the halo'd tile
constexpr int CS_LDS = CS_X4 + CS_HROWS * 64;   // 149 504 B
constexpr int CS_BLOB_BYTES = CS_WBYTES + 3 * 1024;
static_assert(CS_LDS <= 160 * 1024, "LDS budget");
constexpr int CS_WAVES = 8, CS_T = 16, CS_H = CS_T + 2;

struct CspStageK {
    const char* in;              // x = conv 1's output view
    char* out;                   // y = conv 7's output view
    const char* blob;            // CS_BLOB_BYTES: fragment-ordered weights + affine (pack_csp_stage)
    int in_cstride, in_coff, out_cstride, out_coff;
    unsigned in_bytes;
    int N, S;                    // images, side of the stage's tensors (a multiple of 16)
    int tiles_x, tiles_per_img, ntiles;
};

// this lane's scale / shift of channels (4c + g)*8 .. +7, c < NC, from the LDS affine table of one conv
template <int NC>
__device__ __forceinline__ void cs_affine(const float* tab, int cout, int fg, float* sc, float* sh) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int ch = chunk_channel(0, c, fg);
#pragma unroll
        for (int h = 0; h < 8; h += 4) {
            const f32x4 s4 = *(const f32x4*)(tab + ch + h);
            const f32x4 h4 = *(const f32x4*)(tab + cout + ch + h);
#pragma unroll
            for (int e = 0; e < 4; ++e) { sc[c * 8 + h + e] = s4[e]; sh[c * 8 + h + e] = h4[e]; }
        }
    }
}

// A fragments j < NREP of k-step s
template <int NREP>
__device__ __forceinline__ void cs_wfrag(const char* w, int s, int lane, u32x4 (&wf)[NREP]) {
#pragma unroll
    for (int j = 0; j < NREP; ++j) wf[j] = *(const u32x4*)(w + ((s * NREP + j) * 64 + lane) * 16);
}

// BN + Mish of NREP*4 accumulator values -> NREP/2 packed 16-byte chunks (chunk c = channels (4c+g)*8 .. +7)
template <int DT, int NREP>
__device__ __forceinline__ void cs_act_pack(const f32x4 (&acc)[NREP], const float* sc, const float* sh, u32x4* out) {
    float v[NREP * 4];
#pragma unroll
    for (int j = 0; j < NREP; ++j)
        bn_act4<true, CS_ACT>(acc[j], sc + j * 4, sh + j * 4, v + j * 4);
#pragma unroll
    for (int c = 0; c < NREP / 2; ++c) Elem<DT>::store_chunk(&out[c], v + c * 8);
}

template <int DT>
__global__ __launch_bounds__(64 * CS_WAVES, 1) void csp_stage_kernel(const CspStageK p) {
    using E = Elem<DT>;
    using T = typename E::type;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane & 15, g = lane >> 4;
    const float* const aff = (const float*)(smem + CS_AFF);
    char* const X1 = smem + CS_X1;
    char* const X4 = smem + CS_X4;

    // ---- tile schedule: block b runs on XCD b % 8; each XCD owns a contiguous range of tiles and its blocks walk it
    //      together, so tiles that share halo pixels are read through one L2 at about the same time (speed only)
    const int G = gridDim.x, xcd = blockIdx.x & 7, bi = blockIdx.x >> 3;
    const int nb_x = (G - xcd + 7) >> 3;
    const int t_lo = (int)((int64_t)p.ntiles * xcd / 8), t_hi = (int)((int64_t)p.ntiles * (xcd + 1) / 8);
    int t = t_lo + bi;

    // ---- weights + affine -> LDS, once per workgroup (83 x 1 KB wave-wide LDS-DMA pieces)
    {
        const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.blob, CS_BLOB_BYTES);
        for (int u = wave; u < CS_BLOB_BYTES / 1024; u += CS_WAVES)
            buffer_load16_lds(rb, smem + __builtin_amdgcn_readfirstlane(u * 1024), u * 1024 + lane * 16, 0);
    }
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(p.in, p.in_bytes);
    // halo'd tile of x -> X1: one wave-wide piece = 8 rows x 128 B; LDS-DMA writes lane-linearly, so the XOR swizzle of
    // the 16-byte chunk index is applied to the SOURCE address
#if CS_LX
    // per piece k of this wave (u = wave + 8k): byte offset of the lane's 16 bytes relative to halo pixel (0, 0) of the tile, with
    // the lane's border flags in the four low bits (always zero in an offset): 1 = halo row 0, 2 = halo row 17, 4 = halo column 0,
    // 8 = halo column 17.  Rows past the 324 halo pixels get every flag and are always out of range.
    constexpr int CS_NPIECE = (CS_HROWS / 8 + CS_WAVES - 1) / CS_WAVES;
    int lx_rel[CS_NPIECE];
#pragma unroll
    for (int k = 0; k < CS_NPIECE; ++k) {
        const int hp = (wave + CS_WAVES * k) * 8 + (lane >> 3);
        const int hy = hp / CS_H, hx = hp - hy * CS_H;
        const int flags = hp >= CS_H * CS_H ? 15 : ((hy == 0) | ((hy == CS_H - 1) << 1) | ((hx == 0) << 2) | ((hx == CS_H - 1) << 3));
        lx_rel[k] = (((hy * p.S + hx) * p.in_cstride + (((lane & 7) ^ (hp & 7)) * 8)) * 2) | flags;
    }
    auto load_x = [&](int tile) {
        const int n = tile / p.tiles_per_img, rem = tile - n * p.tiles_per_img;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        const int base = (((n * p.S + ty * CS_T - 1) * p.S + tx * CS_T - 1) * p.in_cstride + p.in_coff) * 2;
        const int tmask = (ty == 0) | ((ty == p.tiles_x - 1) << 1) | ((tx == 0) << 2) | ((tx == p.tiles_x - 1) << 3);
#pragma unroll
        for (int k = 0; k < CS_NPIECE; ++k) {
            const int u = wave + CS_WAVES * k;
            if (u >= CS_HROWS / 8) break;                                  // wave-uniform
            const int off = (lx_rel[k] & tmask) || (lx_rel[k] & 15) == 15 ? (int)0x80000000 : base + (lx_rel[k] & ~15);
            buffer_load16_lds(rs_in, X1 + __builtin_amdgcn_readfirstlane(u * 1024), off, 0);
        }
    };
#else
    auto load_x = [&](int tile) {
        const int n = tile / p.tiles_per_img, rem = tile - n * p.tiles_per_img;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        const int y0 = ty * CS_T - 1, x0 = tx * CS_T - 1;
        for (int u = wave; u < CS_HROWS / 8; u += CS_WAVES) {
            const int hp = u * 8 + (lane >> 3);
            const int hy = hp / CS_H, hx = hp - hy * CS_H;
            const int gy = y0 + hy, gx = x0 + hx;
            const bool ok = hp < CS_H * CS_H && (unsigned)gy < (unsigned)p.S && (unsigned)gx < (unsigned)p.S;
            const int off = (((n * p.S + gy) * p.S + gx) * p.in_cstride + p.in_coff + (((lane & 7) ^ (hp & 7)) * 8)) * 2;
            buffer_load16_lds(rs_in, X1 + __builtin_amdgcn_readfirstlane(u * 1024), ok ? off : (int)0x80000000, 0);
        }
    };
#endif
    if (t < t_hi) load_x(t);

    // ---- per-wave fragment geometry (tile independent)
    // main fragments: inner rows iy = 2*wave + i -> halo'd row iy + 1, halo'd columns 1..16
    // extra fragment (waves 0..4): the halo ring -- row 0, row 17, and the two halo columns as 36 pixels in 3 fragments
    int xr_main[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) xr_main[i] = (2 * wave + i + 1) * CS_H + 1 + q;
    // ring fragment e = 0..4: 0 = halo row 0, 1 = halo row 17 (columns 1..16), 2..4 = the two halo columns (36 pixels, clamped)
    auto ring_geom = [&](int e, int& hy, int& hx) {
        if (e < 2) { hy = e == 0 ? 0 : CS_H - 1; hx = 1 + q; }
        else {
            const int pp = min((e - 2) * 16 + q, 2 * CS_H - 1);
            hy = pp >> 1; hx = (pp & 1) * (CS_H - 1);
        }
    };
#if CS_EXTRA
    // the older half of the workgroup (waves 0..3: they win every VALU arbitration on their SIMD) takes all five; wave 0 takes two
    const int n_extra = wave == 0 ? 2 : wave < 4 ? 1 : 0;
#else
    const int n_extra = wave < 5 ? 1 : 0;
#endif

#if CS_PRIO == 2
    if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
    bool first_tile = true;
#ifdef CS_TRACE
    int tr_i = 0;
    unsigned long long tr_t[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (; t < t_hi; t += nb_x) {
        const int n = t / p.tiles_per_img, rem = t - n * p.tiles_per_img;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
#ifdef CS_TRACE
        const bool tr_on = blockIdx.x == CS_TR_WG && tr_i >= CS_TR_T0 && tr_i < CS_TR_T0 + 4;
#endif
        CS_POINT(0);                       // arrives at the tile barrier
        // X1 (and, the first time, the weights) have landed for every wave; every wave is done with the previous tile's
        // T4.  The previous tile's 4 output stores of this wave may still be in flight (they were issued after the DMA).
        if (first_tile) wait_vmcnt_then_barrier<0>();
        else wait_vmcnt_then_barrier<CS_VMCNT>();
        first_tile = false;
        CS_POINT(1);                       // tile landed, barrier passed
        CS_SEG(3);

        // ================= phase A: conv3 -> conv4 -> T4 on the halo'd tile; conv2 on the inner rows
        u32x4 C2[2][2], R3[2][2];              // route (conv2) and residual (conv3) of the wave's two inner fragments, packed
        {
            float sc3[16], sh3[16], sc4[8], sh4[8];
            cs_affine<2>(aff + CS_A3, 64, g, sc3, sh3);
            cs_affine<1>(aff + CS_A4, 32, g, sc4, sh4);
            // conv3 + conv2 of the two main fragments, sharing the weight fragments
            f32x4 a3[2][4], a2[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) a3[i][j] = a2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                u32x4 xf[2], w3[4], w2[4];
#pragma unroll
                for (int i = 0; i < 2; ++i) xf[i] = *(const u32x4*)(X1 + xr_main[i] * 128 + (((s * 4 + g) ^ (xr_main[i] & 7)) * 16));
                cs_wfrag<4>(smem + CS_W3, s, lane, w3);
                cs_wfrag<4>(smem + CS_W2, s, lane, w2);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        Mma<DT>::run(a3[i][j], w3[j], xf[i]);
                        Mma<DT>::run(a2[i][j], w2[j], xf[i]);
                    }
            }
            CS_POINT(2);                   // conv3 + conv2 MFMAs of the main fragments issued
#pragma unroll
            for (int i = 0; i < 2; ++i) cs_act_pack<DT, 4>(a3[i], sc3, sh3, R3[i]);
            {
                float sc2[16], sh2[16];
                cs_affine<2>(aff + CS_A2, 64, g, sc2, sh2);
#pragma unroll
                for (int i = 0; i < 2; ++i) cs_act_pack<DT, 4>(a2[i], sc2, sh2, C2[i]);
            }
            CS_POINT(3);                   // their Mish
            CS_SEG(2);
            // conv4 (64 -> 32) from registers, masked to zero outside the image, -> T4
            auto conv4_to_t4 = [&](const u32x4* x3, int xr, int hy, int hx) {
                f32x4 a4[2];
                a4[0] = a4[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    u32x4 w4[2];
                    cs_wfrag<2>(smem + CS_W4, s, lane, w4);
#pragma unroll
                    for (int j = 0; j < 2; ++j) Mma<DT>::run(a4[j], w4[j], x3[s]);
                }
                u32x4 pk;
                cs_act_pack<DT, 2>(a4, sc4, sh4, &pk);
                const int gy = ty * CS_T - 1 + hy, gx = tx * CS_T - 1 + hx;
                if (!((unsigned)gy < (unsigned)p.S && (unsigned)gx < (unsigned)p.S)) pk = u32x4{0u, 0u, 0u, 0u};
                *(u32x4*)(X4 + xr * 64 + ((g ^ ((xr >> 1) & 3)) * 16)) = pk;
            };
            // conv3 -> conv4 of the halo-ring fragment(s)
            for (int ex = 0; ex < n_extra; ++ex) {
                int ex_hy, ex_hx;
                ring_geom(CS_EXTRA && ex == 1 ? 4 : wave, ex_hy, ex_hx);
                const int xr_extra = ex_hy * CS_H + ex_hx;
                u32x4 X3e[2];
                f32x4 a3e[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) a3e[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    u32x4 w3[4];
                    const u32x4 xf = *(const u32x4*)(X1 + xr_extra * 128 + (((s * 4 + g) ^ (xr_extra & 7)) * 16));
                    cs_wfrag<4>(smem + CS_W3, s, lane, w3);
#pragma unroll
                    for (int j = 0; j < 4; ++j) Mma<DT>::run(a3e[j], w3[j], xf);
                }
                cs_act_pack<DT, 4>(a3e, sc3, sh3, X3e);
                conv4_to_t4(X3e, xr_extra, ex_hy, ex_hx);
            }
            CS_POINT(4);                   // halo-ring conv3 -> conv4
            CS_SEG(1);
#pragma unroll
            for (int i = 0; i < 2; ++i) conv4_to_t4(R3[i], xr_main[i], 2 * wave + i + 1, 1 + q);
        }
        CS_POINT(5);                       // conv4 + Mish + T4 writes
        // T4 complete for every wave; X1 is free: start the next tile's DMA under phases C..E
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        CS_POINT(6);                       // mid barrier passed
        CS_SEG(3);
        if (CS_ABL != 3 && t + nb_x < t_hi) load_x(t + nb_x);
        CS_POINT(7);                       // next tile's DMA issued

        // ================= phase C: conv5 (3x3 over T4) + Add -> conv6 -> conv7 over [conv6 | route] -> HBM
        {
            f32x4 a5[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) a5[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - ky * 3;
                u32x4 xf[2], w5[4];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int r = (2 * wave + i + ky) * CS_H + q + kx;
                    xf[i] = *(const u32x4*)(X4 + r * 64 + ((g ^ ((r >> 1) & 3)) * 16));
                }
                cs_wfrag<4>(smem + CS_W5, tap, lane, w5);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) Mma<DT>::run(a5[i][j], w5[j], xf[i]);
            }
            CS_POINT(8);                   // conv5 MFMAs issued
            u32x4 X5[2][2];
            {
                float sc5[16], sh5[16];
                cs_affine<2>(aff + CS_A5, 64, g, sc5, sh5);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    float v[16];
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        bn_act4<true, CS_ACT>(a5[i][j], sc5 + j * 4, sh5 + j * 4, v + j * 4);
#pragma unroll
                    for (int c = 0; c < 2; ++c) {                  // residual Add (custom_layers.py:44), after the activation
                        float rv[8];
                        E::load_chunk(&R3[i][c], rv);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[c * 8 + e] += rv[e];
                    }
#pragma unroll
                    for (int c = 0; c < 2; ++c) E::store_chunk(&X5[i][c], v + c * 8);
                }
            }
            CS_POINT(9);                   // conv5 Mish + Add
            CS_SEG(2);
            // conv6: 64 -> 64 from registers
            u32x4 Y6[2][2];
            {
                f32x4 a6[2][4];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) a6[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    u32x4 w6[4];
                    cs_wfrag<4>(smem + CS_W6, s, lane, w6);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) Mma<DT>::run(a6[i][j], w6[j], X5[i][s]);
                }
                float sc6[16], sh6[16];
                cs_affine<2>(aff + CS_A6, 64, g, sc6, sh6);
#pragma unroll
                for (int i = 0; i < 2; ++i) cs_act_pack<DT, 4>(a6[i], sc6, sh6, Y6[i]);
            }
            CS_POINT(10);                  // conv6 + Mish
            CS_SEG(1);
            // conv7: Concatenate([conv6, route]) (128) -> 64, then out
            {
                f32x4 a7[2][4];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) a7[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    u32x4 w7[4];
                    cs_wfrag<4>(smem + CS_W7, s, lane, w7);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j) Mma<DT>::run(a7[i][j], w7[j], s < 2 ? Y6[i][s] : C2[i][s - 2]);
                }
                CS_POINT(11);              // conv7 MFMAs issued
                float sc7[16], sh7[16];
                cs_affine<2>(aff + CS_A7, 64, g, sc7, sh7);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    u32x4 Z[2];
                    cs_act_pack<DT, 4>(a7[i], sc7, sh7, Z);
                    const int64_t pix = ((int64_t)n * p.S + ty * CS_T + 2 * wave + i) * p.S + tx * CS_T + q;
                    T* op = (T*)p.out + pix * p.out_cstride + p.out_coff;
#pragma unroll
                    for (int c = 0; c < 2; ++c)
                        if (CS_ABL != 2 || Z[c][0] == 0x12345678u) *(u32x4*)(op + chunk_channel(0, c, g)) = Z[c];
                }
            }
        }
        CS_POINT(12);                      // conv7 Mish + stores issued
#ifdef CS_TRACE
        if (tr_on && lane == 0) {
#pragma unroll
            for (int k = 0; k < 13; ++k) cs_trace_buf[((tr_i - CS_TR_T0) * 8 + wave) * 16 + k] = tr_t[k];
        }
        ++tr_i;
#endif
    }
}

#ifdef CS_TRACE
}  // namespace y4
extern "C" int y4_cs_trace_read(unsigned long long* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(y4::cs_trace_buf), sizeof(unsigned long long) * 4 * 8 * 16);
}
namespace y4 {
#endif

// ------------------------------------------------------------------------------------------------ launch
bool csp_stage_supported(int dtype, int side) { return dtype != Y4_F32 && side % CS_T == 0 && side >= CS_T; }
size_t csp_stage_blob_bytes() { return CS_BLOB_BYTES; }

int csp_stage_launch(int dtype, const void* in, int n, int side, int in_cstride, int in_coff, const void* blob, void* out,
                     int out_cstride, int out_coff, hipStream_t stream) {
    Y4_REQUIRE(csp_stage_supported(dtype, side), Y4_EINVAL, "csp_stage: dtype %d / side %d not supported", dtype, side);
    Y4_REQUIRE(in && blob && out && n > 0, Y4_EINVAL, "csp_stage: null pointer / empty batch");
    Y4_REQUIRE(in_cstride % 8 == 0 && in_coff % 8 == 0 && out_cstride % 8 == 0 && out_coff % 8 == 0, Y4_EINVAL,
               "csp_stage: views not 16-byte aligned");
    const int64_t in_bytes = (int64_t)n * side * side * in_cstride * 2;
    Y4_REQUIRE(in_bytes < (1ll << 31), Y4_EINVAL, "csp_stage: input (%lld B) exceeds the 2 GiB buffer-descriptor range", (long long)in_bytes);
    CspStageK k{};
    k.in = (const char*)in; k.out = (char*)out; k.blob = (const char*)blob;
    k.in_cstride = in_cstride; k.in_coff = in_coff; k.out_cstride = out_cstride; k.out_coff = out_coff;
    k.in_bytes = (unsigned)in_bytes;
    k.N = n; k.S = side;
    k.tiles_x = side / CS_T; k.tiles_per_img = k.tiles_x * k.tiles_x; k.ntiles = n * k.tiles_per_img;
    static int n_cus[64] = {0};
    int dev = 0;
    Y4_CHECK_HIP(hipGetDevice(&dev));
    if (n_cus[dev & 63] == 0) {
        int v = 0;
        Y4_CHECK_HIP(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
        n_cus[dev & 63] = v > 0 ? v : 256;
    }
    // one workgroup per CU (LDS); a multiple of 8 so that every XCD gets the same number of blocks
    int grid = n_cus[dev & 63] & ~7;
    if (grid < 8) grid = 8;
    if (grid > ((k.ntiles + 7) & ~7)) grid = (k.ntiles + 7) & ~7;
    auto launch = [&](auto kern) -> int {
        static PerDeviceOnce once;
        if (const uint64_t bit = once.due()) {
            Y4_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, CS_LDS));
            once.mark(bit);
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * CS_WAVES), CS_LDS, stream, k);
        Y4_CHECK_HIP(hipGetLastError());
        return Y4_OK;
    };
    return dtype == Y4_BF16 ? launch(csp_stage_kernel<Y4_BF16>) : launch(csp_stage_kernel<Y4_F16>);
}

// ------------------------------------------------------------------------------------------------ weight packing
// Darknet (cout, cin, k, k) float32 -> ready-made MFMA A fragments, natural K order, chunked output-channel layout:
//   out[((s*NREP + j)*64 + lane)*8 + e] = W[ch][ci][tap],  ch = ((j>>1)*4 + (i>>2))*8 + (j&1)*4 + (i&3), i = lane & 15,
//   k-step s = tap*(cin/32) + cb, ci = 32*cb + 8*(lane>>4) + e      (1x1: tap = 0; 3x3: tap = ky*3 + kx)
template <int DT>
__global__ void pack_frag_kernel(const float* __restrict__ w, typename Elem<DT>::type* __restrict__ out, int cout, int cin, int kk) {
    const int nrep = cout / 16, cbs = cin / 32, total = cout * cin * kk;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int e = idx & 7, lane = (idx >> 3) & 63, r = idx >> 9;
        const int j = r % nrep, s = r / nrep;
        const int tap = s / cbs, cb = s - tap * cbs;
        const int i = lane & 15, gg = lane >> 4;
        const int ch = ((j >> 1) * 4 + (i >> 2)) * 8 + (j & 1) * 4 + (i & 3);
        const int ci = 32 * cb + 8 * gg + e;
        out[idx] = Elem<DT>::st(w[((int64_t)ch * cin + ci) * kk + tap]);
    }
}

__global__ void csp_affine_kernel(const float* s0, const float* h0, float* dst, int cout) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cout) { dst[i] = s0[i]; dst[cout + i] = h0[i]; }
}

// `w[6]` = Darknet-order float32 kernels of convs 2..7 (device), `scale/shift[6]` their folded BN (device, fp32)
int pack_csp_stage(int dtype, const float* const* w, const float* const* scale, const float* const* shift, void* blob,
                   hipStream_t stream) {
    Y4_REQUIRE(dtype == Y4_BF16 || dtype == Y4_F16, Y4_EINVAL, "pack_csp_stage: 16-bit dtypes only (got %d)", dtype);
    struct Item { int conv, cout, cin, kk, woff, aoff; };
    static const Item items[6] = {{2, 64, 64, 1, CS_W2, CS_A2},  {3, 64, 64, 1, CS_W3, CS_A3}, {4, 32, 64, 1, CS_W4, CS_A4},
                                  {5, 64, 32, 9, CS_W5, CS_A5},  {6, 64, 64, 1, CS_W6, CS_A6}, {7, 64, 128, 1, CS_W7, CS_A7}};
    for (int k = 0; k < 6; ++k) {
        const Item& it = items[k];
        Y4_REQUIRE(w[k] && scale[k] && shift[k], Y4_EINVAL, "pack_csp_stage: null pointer for conv %d", it.conv);
        const int total = it.cout * it.cin * it.kk, blocks = (total + 255) / 256;
        char* dst = (char*)blob + it.woff;
        if (dtype == Y4_BF16) hipLaunchKernelGGL(pack_frag_kernel<Y4_BF16>, dim3(blocks), dim3(256), 0, stream, w[k], (uint16_t*)dst, it.cout, it.cin, it.kk);
        else hipLaunchKernelGGL(pack_frag_kernel<Y4_F16>, dim3(blocks), dim3(256), 0, stream, w[k], (_Float16*)dst, it.cout, it.cin, it.kk);
        hipLaunchKernelGGL(csp_affine_kernel, dim3(1), dim3(128), 0, stream, scale[k], shift[k],
                           (float*)((char*)blob + CS_AFF) + it.aoff, it.cout);
    }
    Y4_CHECK_HIP(hipGetLastError());
    return Y4_OK;
}

}  // namespace y4
