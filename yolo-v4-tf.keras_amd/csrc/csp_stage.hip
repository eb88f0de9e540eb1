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
// 352 Mish channels per pixel.
#include "conv_chain.h"

// experiment switches (scripts/build_variant.sh): CS_ABL 1 = LeakyReLU instead of Mish (VALU ablation), 2 = no output
// stores, 3 = the input tile is loaded once only; CS_VMCNT = stores that may stay in flight across the tile boundary
#ifndef CS_ABL
#define CS_ABL 0
#endif
#ifndef CS_VMCNT
// 0: the tile boundary drains everything.  4 would let the previous tile's 4 output stores stay in flight (measured: 583 ->
// 578 us), but it relies on stores and LDS-DMA loads retiring in issue order RELATIVE TO EACH OTHER on one vmcnt counter,
// which gfx9-family hardware does not promise (LLVM treats mixed pending loads / stores as out of order): not worth 1 %.
#define CS_VMCNT 0
#endif

namespace y4 {

constexpr int CS_ACT = CS_ABL == 1 ? Y4_ACT_LEAKY : Y4_ACT_MISH;

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
    if (t < t_hi) load_x(t);

    // ---- per-wave fragment geometry (tile independent)
    // main fragments: inner rows iy = 2*wave + i -> halo'd row iy + 1, halo'd columns 1..16
    // extra fragment (waves 0..4): the halo ring -- row 0, row 17, and the two halo columns as 36 pixels in 3 fragments
    int xr_main[2], xr_extra, ex_hy, ex_hx;
#pragma unroll
    for (int i = 0; i < 2; ++i) xr_main[i] = (2 * wave + i + 1) * CS_H + 1 + q;
    {
        if (wave < 2) { ex_hy = wave == 0 ? 0 : CS_H - 1; ex_hx = 1 + q; }
        else {
            const int pp = min((wave - 2) * 16 + q, 2 * CS_H - 1);
            ex_hy = pp >> 1; ex_hx = (pp & 1) * (CS_H - 1);
        }
        xr_extra = ex_hy * CS_H + ex_hx;
    }
    const bool has_extra = wave < 5;

    bool first_tile = true;
    for (; t < t_hi; t += nb_x) {
        const int n = t / p.tiles_per_img, rem = t - n * p.tiles_per_img;
        const int ty = rem / p.tiles_x, tx = rem - ty * p.tiles_x;
        // X1 (and, the first time, the weights) have landed for every wave; every wave is done with the previous tile's
        // T4.  The previous tile's 4 output stores of this wave may still be in flight (they were issued after the DMA).
        if (first_tile) wait_vmcnt_then_barrier<0>();
        else wait_vmcnt_then_barrier<CS_VMCNT>();
        first_tile = false;

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
#pragma unroll
            for (int i = 0; i < 2; ++i) cs_act_pack<DT, 4>(a3[i], sc3, sh3, R3[i]);
            {
                float sc2[16], sh2[16];
                cs_affine<2>(aff + CS_A2, 64, g, sc2, sh2);
#pragma unroll
                for (int i = 0; i < 2; ++i) cs_act_pack<DT, 4>(a2[i], sc2, sh2, C2[i]);
            }
            // conv3 of the halo-ring fragment
            u32x4 X3e[2];
            if (has_extra) {
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
            }
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
#pragma unroll
            for (int i = 0; i < 2; ++i) conv4_to_t4(R3[i], xr_main[i], 2 * wave + i + 1, 1 + q);
            if (has_extra) conv4_to_t4(X3e, xr_extra, ex_hy, ex_hx);
        }
        // T4 complete for every wave; X1 is free: start the next tile's DMA under phases C..E
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (CS_ABL != 3 && t + nb_x < t_hi) load_x(t + nb_x);

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
    }
}

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
