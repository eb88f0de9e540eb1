// conv_halo_kernel.h -- 3x3 stride-1 'same' conv (the reference's conv() unit, custom_layers.py:5-31, with its Add / concat-slice
// epilogue) with the INPUT HALO TILE STAGED ONCE PER 64-CHANNEL CHUNK and all nine taps run from it (round 5; VERDICT r4 item 3,
// north_star's "LDS-staged input tiles").
//
// conv_igemm_kernel re-stages the activation rows of every tap: per K-tile of a 192 x 256 tile 24 KB of pixels + 32 KB of weights go
// L2 -> LDS, and the LDS-DMA stream is what bounds its K loop (LABNOTES.md section 4.1d: 7 one-KB pieces per wave and K-tile at
// 100-185 cycles of issue each, 36 % MFMA-busy).  Here an output tile is a BAND of R full image rows of one image -- R * W <= BM
// consecutive pixels of the NHWC tensor -- and its input for one chunk of 64 channels is the band plus one row above and below:
// (R + 2) rows of P = roundup(W + 2, 8) LDS rows of 128 bytes (column 0 and columns > W are the conv's zero padding: out-of-range
// lanes of the descriptor load, no traffic).  In that image tap (ky, kx) of pixel (y, x) is the row  (y + ky) P + (x + kx): the
// tile's own row plus a constant, so the nine K-tiles of a chunk read the SAME LDS tile at nine shifts and only the weights
// (BN x 128 bytes per tap) stream.  Per nine K-tiles of a 384 x 128 tile at 38 x 38: 60 KB of pixels + 144 KB of weights instead
// of 432 + 144 (same FLOPs as 192 x 256's 216 + 288): 2.5x fewer LDS-DMA pieces per FLOP.
//
// K order: the canonical one of common.h (64-channel chunk -> tap -> channel), i.e. exactly what every other conv kernel sums in
// since round 5 -- same MFMAs (v_mfma_f32_16x16x32), same operands, same order: results are BIT-IDENTICAL to conv_igemm_kernel's.
//
// LDS: [halo tile, chunk c & 1][halo tile, (c + 1) & 1][weights, K-tile kt & 1][weights, (kt + 1) & 1][touch scratch].  One barrier
// per K-tile (tap), as in the 2-stage ring: it proves this K-tile's weights (issued one K-tile ago) and -- at tap 0 -- this chunk's
// halo tile (issued piece by piece during the previous chunk's taps) have landed for every wave, and that every wave is done with
// the buffers the next loads overwrite.  Rows are XOR-swizzled by their low three bits like conv_igemm_kernel's; P % 8 == 0 keeps
// a vertical tap shift from changing them, so a lane keeps three fragment base addresses per pixel fragment (kx = 0, 1, 2).
#pragma once
#include "conv_common.h"
#include "conv_tiles.h"

namespace y4 {

#ifndef HALO_ABLATIONS
// 1 (kernel experiments only, scripts/halo_abl.sh builds a variant library with it): the HALO_ABL environment variable then switches
// off parts of the K loop at run time -- bit 1 the weight loads, 2 the halo-tile loads, 4 the barrier -- for TIMING; results are wrong.
// The regular build contains none of it.
#define HALO_ABLATIONS 0
#endif
constexpr int HALO_KA_MAX = 9;     // halo-tile pieces (1 KB) a wave stages per chunk, at most (one per tap)

// PAIR: an LDS pair (conv_igemm_kernel.h) with this conv as head -- BN = Cout: the finished tile stays in LDS in the K loop's
// pixel-operand layout and the following 1x1 conv (p.tail[0], output view p.fin) runs from it in the same kernel.
template <int DT, int BM, int BN, int WM, int WN, bool PAIR = false>
__global__ __launch_bounds__(64 * WM * WN, 1) void conv_halo_kernel(const ConvK p) {
    static_assert(DT != Y4_F32, "halo tiles: 16-bit dtypes");
    constexpr int NT = 64 * WM * WN, NW = WM * WN;
    constexpr int ES = 2, BKB = 128, CPR = 8, EPC = 8;
    constexpr int RPI = NT / CPR;                       // weight rows staged per block-wide load instruction
    constexpr int B_IT = BN / RPI;
    constexpr int WPX = BM / WM, WCH = BN / WN, MREP = WPX / 16, NREP = WCH / 16;
    constexpr int WSTAGE = BN * BKB;
    static_assert(BN % RPI == 0 && WPX % 16 == 0 && WCH % 32 == 0, "tile shape");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    asm volatile("" ::"s"(p.H), "s"(p.W), "s"(p.Cin), "s"(p.K), "s"(p.in_cstride), "s"(p.in_coff), "s"(p.grid_m), "s"(p.grid_n), "s"(p.in_bytes),
                 "s"(p.wt_bytes), "s"(p.h_rows), "s"(p.h_bands), "s"(p.h_pitch), "s"(p.touch));

    // ---- XCD-aware tile mapping (as conv_igemm_kernel: an XCD's blocks take a contiguous run of tiles, channel tile fastest)
    const int nwg = p.grid_m * p.grid_n;
    int t;
    {
        const int b = blockIdx.x, qq = nwg >> 3, rr = nwg & 7, xcd = b & 7, idx = b >> 3;
        t = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
    }
    int tile_m = (int)fastdiv((uint32_t)t, p.div_gridn), tile_n = t - tile_m * p.grid_n;
    if (p.h_xmap != 0 && (nwg & 7) == 0) {
        // Layers whose weights do not fit an XCD's 4 MB L2 (512 -> 1024: 9.4 MB): an XCD keeps ONE channel tile's weights -- the
        // latency-critical operand, one tap of look-ahead -- and streams every image's activations, which have a whole chunk of
        // look-ahead, instead of the other way round.  L2-cold (as in a step) 512 -> 1024 @19^2 100.2 -> 95.7 us; layers whose
        // weights fit lose with it (256 -> 512 @38^2 98 -> 103: eight times the activation traffic), so the launcher decides.
        const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
        if (p.grid_n >= 8 && (p.grid_n & 7) == 0) { const int g8 = p.grid_n >> 3; tile_n = xcd + 8 * (idx % g8); tile_m = idx / g8; }
        else if (p.grid_n == 4 || p.grid_n == 2 || p.grid_n == 1) { const int per = 8 / p.grid_n; tile_n = xcd % p.grid_n; tile_m = idx * per + xcd / p.grid_n; }
    }
    const int n0 = tile_n * BN;
    const int img = (int)fastdiv((uint32_t)tile_m, p.h_div_bands), band = tile_m - img * p.h_bands;
    const int R = p.h_rows, P = p.h_pitch, W = p.W, H = p.H;
    const int y0 = band * R;
    const int rows_here = H - y0 < R ? H - y0 : R;
    const int npx = rows_here * W;                      // valid pixels of the tile: m_base .. m_base + npx
    const int m_base = (img * H + y0) * W;
    const int AROWS = (R + 2) * P, ABYTES = AROWS * BKB, APIECES = AROWS >> 3;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;
    char* const lds_wt = smem + 2 * ABYTES;

    // ---- staging set-up.  Halo tile: piece u = 8 LDS rows; this wave stages pieces wave, wave + NW, ...; lane (row u*8 + lane/8,
    //      physical chunk lane%8) copies logical chunk (lane%8) ^ (row & 7) of image pixel (y0 - 1 + row / P, row % P - 1).
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(p.in, p.in_bytes);
    const __amdgpu_buffer_rsrc_t rs_wt = make_rsrc(p.wt, p.wt_bytes);
    // A piece's eight LDS rows lie in ONE halo row (P % 8 == 0), so its source offset is a wave-uniform scalar (image row, first
    // column) plus a per-lane constant (row within the piece, swizzled chunk) -- computed when the piece is issued instead of kept:
    // nine registers per wave in a K loop that runs at 254-256 (kept, the LDS-pair variants reloaded them from scratch every tap).
    const int lrow = lane >> 3;
    const int lane_part = (lrow * p.in_cstride + (((lane & 7) ^ lrow) * EPC)) * ES;
    auto piece_off = [&](int k) {
        int wv = wave;
        asm volatile("" : "+s"(wv));                             // (opaque: or the optimiser computes all nine in front of the loop again)
        const int u = wv + k * NW;                               // (wave-uniform from here ...)
        const int r8 = u << 3;
        const int yy = (int)fastdiv((uint32_t)r8, p.h_div_pitch), xx0 = r8 - yy * P;
        const int iy = y0 - 1 + yy;
        const bool row_ok = u < APIECES && (unsigned)iy < (unsigned)H;
        const int s_base = (((img * H + iy) * W + xx0 - 1) * p.in_cstride + p.in_coff) * ES;
        const int ix = xx0 - 1 + lrow;                           // (... per lane)
        return row_ok && (unsigned)ix < (unsigned)W ? s_base + lane_part : (int)0x80000000;
    };
    // Weights: as conv_igemm_kernel (row permutation of the chunked accumulator layout, source-side swizzle)
    const int q = tid % CPR, r0 = tid / CPR;
    int b_off[B_IT];
#pragma unroll
    for (int j = 0; j < B_IT; ++j) {
        const int row = r0 + j * RPI;
        const int wb = row / WCH, pr = row - wb * WCH;
        const int jn = pr >> 4, i = pr & 15, g = i >> 2, r = i & 3;
        const int ch = chunk_channel(n0 + wb * WCH, jn >> 1, g) + (jn & 1) * 4 + r;
        b_off[j] = (ch * p.K + ((q ^ swz<CPR>(row)) * EPC)) * ES;
    }
    const int wave_lds = __builtin_amdgcn_readfirstlane(wave * 1024);
    auto stage_a = [&](int k, int buf, int cbyte) {      // piece k of this wave's share of the chunk at channel byte offset cbyte
        buffer_load16_lds(rs_in, smem + buf * ABYTES + (wave + k * NW) * 1024, piece_off(k), cbyte);
    };
    auto stage_w = [&](int st, int ktb) {
#pragma unroll
        for (int j = 0; j < B_IT; ++j) buffer_load16_lds(rs_wt, lds_wt + st * WSTAGE + wave_lds + j * (NT * 16), b_off[j], ktb);
    };

    if (p.touch != 0) weight_touch(rs_wt, lds_wt + 2 * WSTAGE, n0 * p.K * ES, BN * p.K * ES, wave, NW, lane);

    // ---- fragment addresses.  Pixel fragment i of this lane = tile pixel wm*WPX + 16 i + frow = image position (y, x) of the band;
    //      its halo row for tap (ky, kx) is (y + ky) P + x + kx; the three kx variants are kept (the row's swizzle bits move with kx),
    //      ky adds a multiple of 8 rows.  Pixels past the band (the MFMA tile's padding) read pixel 0's rows; they are never stored.
    const int frow = lane & 15, fg = lane >> 4;
    int abase[3][MREP];
#pragma unroll
    for (int i = 0; i < MREP; ++i) {
        int pix = wm * WPX + i * 16 + frow;
        pix = pix < npx ? pix : 0;
        const int y = (int)fastdiv((uint32_t)pix, p.div_wo), x = pix - y * W;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int r = y * P + x + kx;                     // tap row ky = 0 of this pixel
            abase[kx][i] = r * BKB + ((fg ^ (r & 7)) << 4);
        }
    }
    int xo[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) xo[kk] = frow * BKB + (((kk * 4 + fg) ^ swz<CPR>(frow)) * 16);
    const char* const lds_w = lds_wt + (wn * WCH) * BKB;

    f32x4 acc[MREP][NREP];
#pragma unroll
    for (int i = 0; i < MREP; ++i)
#pragma unroll
        for (int j = 0; j < NREP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nchunks = p.Cin >> 6;
    const int my_pieces = (APIECES - wave + NW - 1) / NW;     // pieces of a halo tile this wave stages (wave-uniform)
    // ---- prologue: chunk 0's halo tile and K-tile 0's weights
#pragma unroll
    for (int k = 0; k < HALO_KA_MAX; ++k)
        if (k < my_pieces) stage_a(k, 0, 0);
    stage_w(0, 0);
    int ktb = BKB;                                      // byte offset in a weight row of the NEXT K-tile to stage
    const int row_shift = P * BKB;                      // one halo row down
    for (int c = 0; c < nchunks; ++c) {
        const int abuf = (c & 1) * ABYTES;
        const bool more_chunks = c + 1 < nchunks;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
            if (HALO_ABLATIONS && (p.h_abl & 4)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else wait_vmcnt_then_barrier<0>();
            const int st = (c + tap) & 1;           // stage parity: K-tile index c*9 + tap; 9 is odd, so the parity alternates with c as well
            const char* const sa = smem + abuf + ky * row_shift;
            const char* const sw = lds_w + st * WSTAGE;
            // the next K-tile's weights and one piece of the next chunk's halo tile: a whole K-tile to land (issued in front of the
            // fragment reads: behind them measured equal on the 96 x 64 wave tiles and 8 % slower on the 96 x 32 one).
            // Also measured and not kept: reading the NEXT tap's k-step-0 pixel fragments under this tap's k-step-1 MFMAs (legal: the
            // halo tile does not change with the tap) -- no gain on the 96 x 32 wave tile, which has the registers for it (105 us
            // against 104-107), and 44-288 bytes of scratch per lane on the 96 x 64 ones, which run at 254-256 registers as they are.
            if ((tap < 8 || more_chunks) && !(HALO_ABLATIONS && (p.h_abl & 1))) { stage_w(st ^ 1, ktb); ktb += BKB; }
            if (more_chunks && tap < my_pieces && !(HALO_ABLATIONS && (p.h_abl & 2))) stage_a(tap, (c + 1) & 1, (c + 1) * BKB);
            u32x4 xf[2][MREP], wf[2][NREP];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int i = 0; i < MREP; ++i) xf[kk][i] = *(const u32x4*)(sa + (abase[kx][i] ^ (kk << 6)));
#pragma unroll
                for (int j = 0; j < NREP; ++j) wf[kk][j] = *(const u32x4*)(sw + j * 16 * BKB + xo[kk]);
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < MREP; ++i)
#pragma unroll
                    for (int j = 0; j < NREP; ++j) Mma<DT>::run(acc[i][j], wf[kk][j], xf[kk][i]);
            __builtin_amdgcn_s_setprio(0);
        }
    }

    const int mrow = m_base + wm * WPX + frow, m_limit = m_base + npx;
    if constexpr (PAIR) {
        // ---- LDS pair (as conv_igemm_kernel's): the head's tile -> LDS panels [BN/64][BM rows][128 B] over the dead halo buffers, the
        // tail's weights through two stages behind them, the same fragment reads and K order as the 1x1 conv's own kernel.  Rows
        // past the band hold whatever their (never stored) accumulators held.
        constexpr int XPANEL = BM * 128, NK2 = BN / 64, XBYTES = NK2 * XPANEL, W2STAGE = BN * 128;
        __syncthreads();                                   // every wave is done with the K loop's buffers
        char* const xl = smem;
        const int xrow = wm * WPX + frow, chw = wn * WCH;
        // (the tail's staging offsets are computed HERE, from values laundered through an empty volatile asm: hoisted in front of the K
        //  loop -- which runs at 254-256 registers -- they cost it scratch accesses)
        int r0b = r0, qb = q;
        asm volatile("" : "+v"(r0b), "+v"(qb));
        int b2_off[B_IT];
#pragma unroll
        for (int j = 0; j < B_IT; ++j) {
            const int row = r0b + j * RPI;
            const int wb = row / WCH, pr = row - wb * WCH;
            const int jn = pr >> 4, i = pr & 15, g = i >> 2, r = i & 3;
            const int ch = chunk_channel(wb * WCH, jn >> 1, g) + (jn & 1) * 4 + r;
            b2_off[j] = (ch * p.tail_k + ((qb ^ swz<CPR>(row)) * EPC)) * ES;
        }
        const __amdgpu_buffer_rsrc_t rs_w2 = make_rsrc(p.tail[0].w, p.tail_w_bytes);
        auto stage_w2 = [&](int buf, int kt) {
#pragma unroll
            for (int j = 0; j < B_IT; ++j)
                buffer_load16_lds(rs_w2, smem + XBYTES + buf * W2STAGE + wave_lds + j * (NT * 16), b2_off[j], kt * BKB);
        };
        stage_w2(0, 0);                                    // its round trip hides under the head's epilogue
        conv_epilogue<DT, MREP, NREP, true>(p, acc, mrow, m_limit, chw, fg, true, xl, xrow, XPANEL);
        __syncthreads();                                   // X complete
        f32x4 acc2[MREP][NREP];
#pragma unroll
        for (int i = 0; i < MREP; ++i)
#pragma unroll
            for (int j = 0; j < NREP; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int nk2 = p.tail_k >> 6;
        const int (&xo2)[2] = xo;
        for (int kt = 0; kt < nk2; ++kt) {
            wait_vmcnt_then_barrier<0>();
            if (kt + 1 < nk2) stage_w2((kt + 1) & 1, kt + 1);
            const char* sx = xl + (p.tail_panel0 + kt) * XPANEL + (wm * WPX) * BKB;
            const char* sw2 = smem + XBYTES + (kt & 1) * W2STAGE + (wn * WCH) * BKB;
            u32x4 xf2[2][MREP], wf2[2][NREP];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int i = 0; i < MREP; ++i) xf2[kk][i] = *(const u32x4*)(sx + i * 16 * BKB + xo2[kk]);
#pragma unroll
                for (int j = 0; j < NREP; ++j) wf2[kk][j] = *(const u32x4*)(sw2 + j * 16 * BKB + xo2[kk]);
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int i = 0; i < MREP; ++i)
#pragma unroll
                    for (int j = 0; j < NREP; ++j) Mma<DT>::run(acc2[i][j], wf2[kk][j], xf2[kk][i]);
        }
        ConvK p2 = p;
        p2.scale = p.tail[0].scale; p2.shift = p.tail[0].shift; p2.act = p.tail_act; p2.res = nullptr;
        p2.out = p.fin; p2.out_cstride = p.fin_cstride; p2.out_coff = p.fin_coff;
        p2.cout_store = p.tail[0].cout; p2.upsample = 0; p2.out_f32 = p.pair >> 1;
        p2.split = p.tail_split; p2.out2 = p.fin2; p2.out2_cstride = p.fin2_cstride; p2.out2_coff = p.fin2_coff;
        p2.fast_epi = p.fast_tail; p2.out_bytes = p.fin_bytes; p2.out2_bytes = p.fin2_bytes;
        conv_epilogue_rows<DT, MREP, NREP>(p2, acc2, mrow, m_limit, chw, fg, BN <= p2.cout_store);
        if (p.store_x) pair_store_tile<DT, MREP, NREP>(p, xl, xrow, XPANEL, mrow, m_limit, chw, fg);
    } else {
        // ---- epilogue (conv_common.h): rows past the band are dropped
        const bool ch_full = n0 + BN <= p.cout_store;
        conv_epilogue_rows<DT, MREP, NREP>(p, acc, mrow, m_limit, n0 + wn * WCH, fg, ch_full);
    }
}

// ------------------------------------------------------------------------------------------- launch
template <int DT, int BM, int BN, int WM, int WN, bool PAIR = false>
static int launch_halo_cfg(const ConvK& k, hipStream_t stream) {
    size_t lds = halo_lds_bytes(k.h_rows, k.h_pitch, BN);
    Y4_REQUIRE(lds <= 160 * 1024, Y4_EINVAL, "conv2d: halo tile needs %zu bytes of LDS", lds);
    Y4_REQUIRE(((k.h_rows + 2) * k.h_pitch / 8 + WM * WN - 1) / (WM * WN) <= HALO_KA_MAX, Y4_EINVAL, "conv2d: halo tile has too many pieces per wave");
    if (PAIR) {
        const size_t lds_pair = (size_t)(BN / 64) * BM * 128 + (size_t)2 * BN * 128;      // tile panels + two weight stages of the tail
        static_assert(!PAIR || (BN / 64) * BM * 128 + 2 * BN * 128 <= 160 * 1024, "LDS budget of the pair phase");
        lds = lds > lds_pair ? lds : lds_pair;
    }
    auto kern = conv_halo_kernel<DT, BM, BN, WM, WN, PAIR>;
    static PerDeviceOnce once;
    if (const uint64_t bit = once.due()) {
        Y4_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        once.mark(bit);
    }
    hipLaunchKernelGGL(kern, dim3(k.grid_m * k.grid_n), dim3(64 * WM * WN), lds, stream, k);
    Y4_CHECK_HIP(hipGetLastError());
    return Y4_OK;
}

template <int DT>
static int launch_halo(int bm, int bn, const ConvK& k, hipStream_t s) {
    if (k.pair) {
        if (bm == 192 && bn == 256) return launch_halo_cfg<DT, 192, 256, 2, 4, true>(k, s);
        if (bm == 384 && bn == 128) return launch_halo_cfg<DT, 384, 128, 4, 2, true>(k, s);
        if (bm == 320 && bn == 128) return launch_halo_cfg<DT, 320, 128, 4, 2, true>(k, s);
        set_error("conv2d: no halo tile %d x %d heads an LDS pair", bm, bn);
        return Y4_EINVAL;
    }
    if (bm == 384 && bn == 128) return launch_halo_cfg<DT, 384, 128, 4, 2>(k, s);
    if (bm == 320 && bn == 128) return launch_halo_cfg<DT, 320, 128, 4, 2>(k, s);
    if (bm == 192 && bn == 256) return launch_halo_cfg<DT, 192, 256, 2, 4>(k, s);
    if (bm == 192 && bn == 128) return launch_halo_cfg<DT, 192, 128, 2, 4>(k, s);
    set_error("conv2d: no halo tile %d x %d", bm, bn);
    return Y4_EINVAL;
}

}  // namespace y4
