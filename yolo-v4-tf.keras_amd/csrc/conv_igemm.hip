// conv_igemm.hip -- the reference's conv() unit (custom_layers.py:5-31: Conv2D -> BatchNormalization ->
// Mish/LeakyReLU) as ONE im2col-free implicit-GEMM MFMA kernel for gfx950, with the graph's glue ops
// folded into its epilogue: residual Add (custom_layers.py:44), Concatenate (:68,:149,... -> a channel
// slice store), UpSampling2D (:147,:159 -> 2x2 replicated store).
//
// GEMM view (NHWC activations):  D[ch][px] = sum_k Wt[ch][k] * X[px][k],  k = (ky*kw + kx)*Cin + ci
//   X rows are gathered on the fly: for K-tile (tap, c0) row px is the BK contiguous channels
//   in[n, ho*s+ky-pad, wo*s+kx-pad, c0:c0+BK]  (zero page when the tap falls in the padding),
//   copied HBM/L2 -> LDS with global_load_lds (16 B per lane, no VGPR round trip).
//   Weights are pre-packed [cout_pad][kh][kw][cin] so their K-tile rows are contiguous too.
// The weight fragment is the MFMA *A* operand and the pixel fragment the *B* operand, so the accumulator
// layout is D[row = channel][col = pixel]: every lane ends up holding runs of 8 CONSECUTIVE channels of one
// pixel (the channel <-> MFMA-row assignment is free; the "chunked" assignment of conv_common.h is applied as a
// row permutation when staging the weight tile), i.e. the NHWC epilogue is 16-byte vector loads/stores with no
// LDS transpose, 64 contiguous bytes per pixel per instruction, and a lane's chunks are ready-made MFMA B
// operands for a chained 1x1 conv (conv_chain.h).
// LDS rows are BKB (64|128) bytes with the 16-byte chunk index XOR-swizzled by the row so that every
// ds_read_b128 lane group hits 16 distinct slots; global_load_lds writes LDS lane-linearly, so the
// swizzle is applied to the per-lane SOURCE address and again on the fragment read (same involution).
// Pipeline: 2 LDS stages, one barrier per K-tile: loads of tile t+1 fly during the MFMAs of tile t.
#include "conv_chain.h"

namespace y4 {

template <int DT, int BM, int BN, int WM, int WN, int BKB, int NST, int CHAIN = 0, bool PAIR = false>
__global__ __launch_bounds__(64 * WM * WN, CHAIN ? 2 : 1) void conv_igemm_kernel(const ConvK p) {
    constexpr int NT = 64 * WM * WN;
    constexpr int ES = (DT == Y4_F32) ? 4 : 2;
    constexpr int BK = BKB / ES;            // K elements per tile
    constexpr int CPR = BKB / 16;           // 16-byte chunks per LDS row
    constexpr int EPC = 16 / ES;            // elements per chunk
    constexpr int RPI = NT / CPR;           // rows staged per block-wide load instruction
    constexpr int A_IT = (BM + RPI - 1) / RPI;          // last iteration is predicated when BM % RPI != 0
    constexpr bool A_PART = BM % RPI != 0;
    constexpr int B_IT = BN >= RPI ? BN / RPI : 1;      // BN < RPI: only the first BN*CPR threads stage weights
    constexpr bool B_PART = BN < RPI;
    constexpr int WPX = BM / WM, WCH = BN / WN;
    constexpr int MREP = WPX / 16, NREP = WCH / 16;
    constexpr int CPL = 4 * NREP;           // consecutive channels a lane owns
    constexpr bool PHASED = (NST == 12);     // 2 LDS stages, two wave groups staggered by one of 4 phases per K-tile
    constexpr int SN = PHASED ? 2 : NST;     // LDS stages
    constexpr bool PREFRAG = true;           // all fragments of a K-tile are read before its first MFMA (measured: never slower)
    constexpr int STAGE = (BM + BN) * BKB;
    constexpr int KSTEPS = BKB / 64;        // MFMA k-steps (4 chunks each) per tile
    static_assert(BM % 16 == 0 && (BN % RPI == 0 || (RPI % BN == 0 && (BN * CPR) % 64 == 0)), "tile rows vs rows-per-iteration");
    static_assert(MREP >= 1 && NREP >= 1, "wave tile");
    constexpr int LPT = A_IT + B_IT;        // LDS-DMA instructions a wave issues per stage
    static_assert(SN >= 2 && SN <= 4 && (SN == 2 || !B_PART), "deep pipelines need uniform weight loads per wave");
    static_assert((SN - 2) * LPT <= 63, "vmcnt field");
    static_assert(!PHASED || (NT == 512 && KSTEPS == 2), "phased schedule: 8 waves, 128-byte K rows");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    // ---- XCD-aware tile mapping: block b runs on XCD b%8; give each XCD a contiguous run of tiles with the
    //      channel tile fastest, so blocks sharing an activation row panel share an L2 (speed only).
    const int nwg = p.grid_m * p.grid_n;
    int t;
    {
        const int b = blockIdx.x, qq = nwg >> 3, rr = nwg & 7, xcd = b & 7, idx = b >> 3;
        t = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
    }
    const int tile_m = (int)fastdiv((uint32_t)t, p.div_gridn), tile_n = t - tile_m * p.grid_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
    ChainPrefetch<CHAIN != 0 ? MREP : 1> chain_pf;
    if constexpr (CHAIN != 0) {
        chain_stage_weights<CHAIN, WM * WN>(p, smem + SN * STAGE, __builtin_amdgcn_readfirstlane(wave), lane);
        chain_prefetch<DT, MREP, CHAIN>(p, chain_pf, m0 + wm * WPX + (lane & 15), p.M, lane);
    }

    // ---- staging set-up: this thread copies physical chunk slot `q` of rows r0 + j*RPI.
    // Loads are buffer_load_dwordx4 ... lds through two raw buffer descriptors (activations, weights): the
    // per-lane byte offset is fixed per (row, tap), the K-tile advance (c0) rides in the scalar soffset, and a
    // tap that falls into the padding (or a row past M) gets an out-of-range voffset, which the hardware
    // bounds check turns into zeros -- no per-tile address arithmetic, no zero page.
    const int q = tid % CPR, r0 = tid / CPR;
    int a_off[A_IT], a_hi[A_IT], a_wi[A_IT];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int j = 0; j < A_IT; ++j) {
        const int row = r0 + j * RPI;
        const int m = m0 + row;
        const int mm = m < p.M ? m : 0;
        const int n = (int)fastdiv((uint32_t)mm, p.div_howo), rem = mm - n * HoWo;
        const int ho = (int)fastdiv((uint32_t)rem, p.div_wo), wo = rem - ho * p.Wo;
        const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
        a_off[j] = (((n * p.H + hi0) * p.W + wi0) * p.in_cstride + p.in_coff + ((q ^ swz<CPR>(row)) * EPC)) * ES;
        a_hi[j] = m < p.M ? hi0 : -100000;     // rows past M never validate -> zeros
        a_wi[j] = wi0;
    }
    int b_off[B_IT];
#pragma unroll
    for (int j = 0; j < B_IT; ++j) {
        const int row = B_PART ? (r0 % BN) : (r0 + j * RPI);
        // LDS row (wave block, fragment jn, MFMA row i = g*4 + r)  <-  channel of the chunked layout (conv_common.h)
        const int wb = row / WCH, pr = row - wb * WCH;
        const int jn = pr >> 4, i = pr & 15, g = i >> 2, r = i & 3;
        const int ch = chunk_channel(n0 + wb * WCH, jn >> 1, g) + (jn & 1) * 4 + r;
        b_off[j] = (ch * p.K + ((q ^ swz<CPR>(row)) * EPC)) * ES;
    }
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(p.in, p.in_bytes);
    const __amdgpu_buffer_rsrc_t rs_wt = make_rsrc(p.wt, p.wt_bytes);
    const int wave_lds = __builtin_amdgcn_readfirstlane(wave * 1024);          // provably uniform -> SALU/M0 path

    // wave-uniform: does this wave skip the (partial) last A iteration?  (rows r0 + (A_IT-1)*RPI >= BM)
    const bool a_skip = A_PART && __builtin_amdgcn_readfirstlane(r0 + (A_IT - 1) * RPI >= BM ? 1 : 0) != 0;
    int ky = 0, kx = 0, c0b = 0, ktb = 0;          // staging cursor: tap, byte offset of c0, byte offset of k in the weights
    int a_vo[A_IT];
    auto set_tap = [&]() {
        const int tap_off = ((ky * p.W + kx) * p.in_cstride) * ES;
#pragma unroll
        for (int j = 0; j < A_IT; ++j) {
            const bool ok = (unsigned)(a_hi[j] + ky) < (unsigned)p.H && (unsigned)(a_wi[j] + kx) < (unsigned)p.W;
            a_vo[j] = ok ? a_off[j] + tap_off : (int)0x80000000;            // >= num_records -> reads as zero
        }
    };
    set_tap();
    auto stage = [&](int buf) {
        const int da = buf * STAGE + wave_lds;
        const int db = da + BM * BKB;
#pragma unroll
        for (int j = 0; j < A_IT; ++j)
            if (!A_PART || r0 + j * RPI < BM)                                // wave-uniform (8 rows per wave, BM % 16 == 0)
                buffer_load16_lds(rs_in, smem + da + j * (NT * 16), a_vo[j], c0b);
#pragma unroll
        for (int j = 0; j < B_IT; ++j)
            if (!B_PART || tid < BN * CPR)                                   // wave-uniform predicate
                buffer_load16_lds(rs_wt, smem + db + j * (NT * 16), b_off[j], ktb);
        ktb += BKB;
        c0b += BKB;
        if (c0b >= p.Cin * ES) {
            c0b = 0;
            if (++kx >= p.ksize) { kx = 0; ++ky; }
            set_tap();
        }
    };

    // ---- fragment read addresses (row & swizzle depend on the lane only)
    const int frow = lane & 15, fg = lane >> 4;
    int xo[KSTEPS];
#pragma unroll
    for (int kk = 0; kk < KSTEPS; ++kk) xo[kk] = frow * BKB + (((kk * 4 + fg) ^ swz<CPR>(frow)) * 16);
    const char* const lds_x = smem + (wm * WPX) * BKB;
    const char* const lds_w = smem + BM * BKB + (wn * WCH) * BKB;

    f32x4 acc[MREP][NREP];
#pragma unroll
    for (int i = 0; i < MREP; ++i)
#pragma unroll
        for (int j = 0; j < NREP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    if constexpr (PHASED) {
        // Staggered 4-phase schedule for one 8-wave workgroup per CU.  A K-tile is READ(k-step 0) | MMA | READ(k-step 1) |
        // MMA with a workgroup barrier after every phase; waves 4-7 run ONE phase behind waves 0-3 (they take one
        // extra barrier first, waves 0-3 one extra at the end), so while one group's 4 waves (one per SIMD) issue
        // MFMAs the other group's 4 waves read their fragments from LDS: the matrix pipe and the LDS port are busy
        // at the same time instead of alternately.  Stage hazards (2 LDS stages):
        //   group 0 issues tile t+1's loads at the start of its tile t   (both groups have finished reading t-1),
        //   group 1 issues tile t+2's loads at the start of its last MMA phase of tile t,
        //   each wave waits for its own loads (vmcnt(0)) before the barrier that precedes the first read of them.
        auto phase_barrier = [] {
            asm volatile("s_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        auto wait_loads = [] { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
        const bool grp1 = __builtin_amdgcn_readfirstlane(wave >= 4 ? 1 : 0) != 0;
        stage(0);
        wait_loads();
        phase_barrier();
        if (grp1) {
            if (nk > 1) stage(1);
            phase_barrier();
        }
        for (int kt = 0; kt < nk; ++kt) {
            const char* sx = lds_x + (kt & 1) * STAGE;
            const char* sw = lds_w + (kt & 1) * STAGE;
            u32x4 xf[MREP], wf[NREP];
            if (!grp1 && kt + 1 < nk) stage((kt + 1) & 1);
#pragma unroll
            for (int i = 0; i < MREP; ++i) xf[i] = *(const u32x4*)(sx + i * 16 * BKB + xo[0]);
#pragma unroll
            for (int j = 0; j < NREP; ++j) wf[j] = *(const u32x4*)(sw + j * 16 * BKB + xo[0]);
            phase_barrier();
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < MREP; ++i)
#pragma unroll
                for (int j = 0; j < NREP; ++j) Mma<DT>::run(acc[i][j], wf[j], xf[i]);
            __builtin_amdgcn_s_setprio(0);
            phase_barrier();
#pragma unroll
            for (int i = 0; i < MREP; ++i) xf[i] = *(const u32x4*)(sx + i * 16 * BKB + xo[1]);
#pragma unroll
            for (int j = 0; j < NREP; ++j) wf[j] = *(const u32x4*)(sw + j * 16 * BKB + xo[1]);
            if (grp1) wait_loads();
            phase_barrier();
            if (grp1 && kt + 2 < nk) stage(kt & 1);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < MREP; ++i)
#pragma unroll
                for (int j = 0; j < NREP; ++j) Mma<DT>::run(acc[i][j], wf[j], xf[i]);
            __builtin_amdgcn_s_setprio(0);
            if (!grp1) wait_loads();
            phase_barrier();
        }
        if (!grp1) phase_barrier();
    } else {
    // NST-stage ring: tiles kt+1 .. kt+NST-2 stay in flight (counted vmcnt, never drained in steady state)
    // while tile kt is consumed; ONE barrier per K-tile: it proves tile kt has landed for every wave and
    // that every wave is done reading the stage (tile kt-1's) that the next stage() call overwrites.
#pragma unroll
    for (int s = 0; s < SN - 1; ++s)
        if (s < nk) stage(s);
    int cur = 0, nxt = SN - 1;
    for (int kt = 0; kt < nk; ++kt) {
        const int ahead = nk - 1 - kt;         // tiles issued after kt so far (capped at NST-2)
        if (SN == 2 || ahead == 0) wait_vmcnt_then_barrier<0>();
        else if (a_skip) {                     // this wave issues one load fewer per stage (partial last A iteration)
            if (SN == 3 || ahead == 1) wait_vmcnt_then_barrier<LPT - 1>();
            else wait_vmcnt_then_barrier<2 * (LPT - 1)>();
        } else if (SN == 3 || ahead == 1) wait_vmcnt_then_barrier<LPT>();
        else wait_vmcnt_then_barrier<2 * LPT>();
        if (kt + SN - 1 < nk) stage(nxt);
        const char* sx = lds_x + cur * STAGE;
        const char* sw = lds_w + cur * STAGE;
        cur = cur + 1 == SN ? 0 : cur + 1;
        nxt = nxt + 1 == SN ? 0 : nxt + 1;
        if constexpr (PREFRAG) {
            // issue every fragment read of the K-tile first: the reads of k-step 1 then fly behind the MFMAs of
            // k-step 0 (the compiler's counted lgkmcnt waits keep the order), at the price of a second fragment set
            u32x4 xf[KSTEPS][MREP], wf[KSTEPS][NREP];
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
#pragma unroll
                for (int i = 0; i < MREP; ++i) xf[kk][i] = *(const u32x4*)(sx + i * 16 * BKB + xo[kk]);
#pragma unroll
                for (int j = 0; j < NREP; ++j) wf[kk][j] = *(const u32x4*)(sw + j * 16 * BKB + xo[kk]);
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk)
#pragma unroll
                for (int i = 0; i < MREP; ++i)
#pragma unroll
                    for (int j = 0; j < NREP; ++j) Mma<DT>::run(acc[i][j], wf[kk][j], xf[kk][i]);
            __builtin_amdgcn_s_setprio(0);
        } else
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            u32x4 xf[MREP], wf[NREP];
#pragma unroll
            for (int i = 0; i < MREP; ++i) xf[i] = *(const u32x4*)(sx + i * 16 * BKB + xo[kk]);
#pragma unroll
            for (int j = 0; j < NREP; ++j) wf[j] = *(const u32x4*)(sw + j * 16 * BKB + xo[kk]);
#pragma unroll
            for (int i = 0; i < MREP; ++i)
#pragma unroll
                for (int j = 0; j < NREP; ++j) Mma<DT>::run(acc[i][j], wf[j], xf[i]);
        }
    }
    }

    // ---- epilogue (conv_common.h): scale/shift, activation, residual, packed converts, slice / upsampled / split store
    if constexpr (CHAIN != 0) {
        // chained 1x1 convs consume the tile straight from the accumulators (conv_chain.h)
        static_assert(WN == 1 && BN == 16 * ChainShape<CHAIN>::HEAD_NREP && DT != Y4_F32,
                      "chain head: one wave column over all its output channels, 16-bit");
        chain_epilogue<DT, MREP, CHAIN>(p, smem + SN * STAGE, acc, chain_pf, m0 + wm * WPX + frow, p.M, lane);
    } else if constexpr (PAIR) {
        // ---- LDS pair: this conv's tile stays in LDS (as BN/64 panels in the K loop's pixel-operand layout) and the
        // following 1x1 conv (BN -> BN channels, ordinary packed weights) runs from it with the same fragment reads and
        // the same K order as its own kernel would -- bit-identical, one launch and one HBM read less.
        static_assert(BKB == 128 && DT != Y4_F32 && SN == 2 && !PHASED && !B_PART, "LDS pair: plain 2-stage, 128-byte rows");
        constexpr int XPANEL = BM * 128, NK2 = BN / 64, XBYTES = NK2 * XPANEL, W2STAGE = BN * 128;
        __syncthreads();                                   // every wave is done with the stage buffers X overwrites
        char* const xl = smem;
        const int xrow = wm * WPX + frow, mrow = m0 + xrow, chw = wn * WCH;
        // tail weights: same row permutation as the head's weight tile, tail_k contiguous elements per channel row
        int b2_off[B_IT];
#pragma unroll
        for (int j = 0; j < B_IT; ++j) {
            const int row = r0 + j * RPI;
            const int wb = row / WCH, pr = row - wb * WCH;
            const int jn = pr >> 4, i = pr & 15, g = i >> 2, r = i & 3;
            const int ch = chunk_channel(wb * WCH, jn >> 1, g) + (jn & 1) * 4 + r;
            b2_off[j] = (ch * p.tail_k + ((q ^ swz<CPR>(row)) * EPC)) * ES;
        }
        const __amdgpu_buffer_rsrc_t rs_w2 = make_rsrc(p.tail[0].w, p.tail_w_bytes);
        auto stage_w2 = [&](int buf, int kt) {
#pragma unroll
            for (int j = 0; j < B_IT; ++j)
                buffer_load16_lds(rs_w2, smem + XBYTES + buf * W2STAGE + wave_lds + j * (NT * 16), b2_off[j], kt * BKB);
        };
        stage_w2(0, 0);                                    // its round trip hides under the head's epilogue
        conv_epilogue<DT, MREP, NREP, true>(p, acc, mrow, p.M, chw, fg, m0 + BM <= p.M, xl, xrow, XPANEL);
        __syncthreads();                                   // X complete (ds_write -> lgkmcnt(0) -> barrier)
        f32x4 acc2[MREP][NREP];
#pragma unroll
        for (int i = 0; i < MREP; ++i)
#pragma unroll
            for (int j = 0; j < NREP; ++j) acc2[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int nk2 = p.tail_k >> 6;                     // K-tiles of the tail: its input is panels tail_panel0 .. +nk2-1
        for (int kt = 0; kt < nk2; ++kt) {
            wait_vmcnt_then_barrier<0>();
            if (kt + 1 < nk2) stage_w2((kt + 1) & 1, kt + 1);
            const char* sx = xl + (p.tail_panel0 + kt) * XPANEL + (wm * WPX) * BKB;
            const char* sw = smem + XBYTES + (kt & 1) * W2STAGE + (wn * WCH) * BKB;
            u32x4 xf[KSTEPS][MREP], wf[KSTEPS][NREP];
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) {
#pragma unroll
                for (int i = 0; i < MREP; ++i) xf[kk][i] = *(const u32x4*)(sx + i * 16 * BKB + xo[kk]);
#pragma unroll
                for (int j = 0; j < NREP; ++j) wf[kk][j] = *(const u32x4*)(sw + j * 16 * BKB + xo[kk]);
            }
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk)
#pragma unroll
                for (int i = 0; i < MREP; ++i)
#pragma unroll
                    for (int j = 0; j < NREP; ++j) Mma<DT>::run(acc2[i][j], wf[kk][j], xf[kk][i]);
        }
        // tail epilogue through the ordinary path: a ConvK that describes the 1x1 conv's output side
        ConvK p2 = p;
        p2.scale = p.tail[0].scale; p2.shift = p.tail[0].shift; p2.act = p.tail_act; p2.res = nullptr;
        p2.out = p.fin; p2.out_cstride = p.fin_cstride; p2.out_coff = p.fin_coff;
        p2.cout_store = p.tail[0].cout; p2.upsample = 0; p2.out_f32 = p.pair >> 1;
        p2.split = p.tail_split; p2.out2 = p.fin2; p2.out2_cstride = p.fin2_cstride; p2.out2_coff = p.fin2_coff;
        conv_epilogue<DT, MREP, NREP>(p2, acc2, mrow, p.M, chw, fg, (m0 + BM <= p.M) && BN <= p2.cout_store);
        if (p.store_x) pair_store_tile<DT, MREP, NREP>(p, xl, xrow, XPANEL, mrow, p.M, chw, fg);
    } else {
        const bool full = (m0 + BM <= p.M) && (n0 + BN <= p.cout_store);
        conv_epilogue<DT, MREP, NREP>(p, acc, m0 + wm * WPX + frow, p.M, n0 + wn * WCH, fg, full);
    }
}

// ------------------------------------------------------------------------------------------- launch
struct TileCfg {
    int bm, bn, wm, wn, bkb, nst;
};
// id, BM (pixels), BN (channels), WM, WN (wave grid), BKB (bytes of K per LDS row), NST (ring stages).
// The table is also what tests and the autotuner sweep through y4_conv_desc.tile.
#define Y4_TILES(X)            \
    X(1, 128, 128, 2, 2, 128, 2)  \
    X(2, 128, 128, 2, 2, 64, 2)   \
    X(3, 128, 64, 4, 1, 128, 2)   \
    X(4, 128, 64, 4, 1, 64, 2)    \
    X(5, 128, 32, 4, 1, 128, 2)   \
    X(6, 128, 32, 4, 1, 64, 2)    \
    X(7, 256, 128, 4, 2, 128, 2)  \
    X(8, 64, 128, 1, 4, 128, 2)   \
    X(9, 64, 128, 1, 4, 64, 2)    \
    X(10, 64, 64, 2, 2, 128, 2)   \
    X(11, 64, 64, 2, 2, 64, 2)    \
    X(12, 64, 256, 1, 4, 128, 2)  \
    X(13, 128, 256, 2, 4, 128, 2) \
    X(14, 128, 128, 2, 2, 128, 3) \
    X(15, 128, 64, 4, 1, 64, 4)   \
    X(16, 32, 128, 1, 4, 128, 2)  \
    X(17, 64, 128, 1, 4, 128, 3)  \
    X(18, 256, 256, 2, 4, 128, 2) \
    X(19, 192, 256, 2, 4, 128, 2) \
    X(20, 96, 128, 2, 2, 128, 2)  \
    X(21, 96, 256, 2, 4, 128, 2)  \
    X(22, 160, 256, 2, 4, 128, 2) \
    X(23, 224, 256, 2, 4, 128, 2) \
    X(24, 160, 128, 2, 2, 128, 2) \
    X(25, 192, 128, 2, 2, 128, 2) \
    X(26, 144, 128, 1, 4, 128, 2) \
    X(27, 80, 128, 1, 4, 128, 2)  \
    X(28, 48, 128, 1, 4, 128, 2)  \
    X(29, 112, 128, 1, 4, 128, 2) \
    X(30, 192, 256, 2, 4, 128, 12) \
    X(31, 256, 256, 2, 4, 128, 12) \
    X(32, 224, 256, 2, 4, 128, 12)

#define Y4_TILE_ROW(id, bm, bn, wm, wn, bkb, nst) {bm, bn, wm, wn, bkb, nst},
static const TileCfg kTiles[] = {Y4_TILES(Y4_TILE_ROW)};
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);

int conv_tile_count() { return kNumTiles; }

template <int DT, int BM, int BN, int WM, int WN, int BKB, int NST, int CHAIN = 0, bool PAIR = false>
static int launch_cfg(const ConvK& k, hipStream_t stream) {
    constexpr int lds_main = (NST == 12 ? 2 : NST) * (BM + BN) * BKB + (CHAIN ? ChainShape<CHAIN ? CHAIN : 1>::LDS_BYTES : 0);
    constexpr int lds_pair = PAIR ? (BN / 64) * BM * 128 + 2 * BN * 128 : 0;     // tile panels + two weight stages of the tail
    constexpr int lds = lds_main > lds_pair ? lds_main : lds_pair;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = conv_igemm_kernel<DT, BM, BN, WM, WN, BKB, NST, CHAIN, PAIR>;
    static bool attr_set = false;
    if (!attr_set && lds > 48 * 1024) {
        Y4_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3(k.grid_m * k.grid_n), dim3(64 * WM * WN), lds, stream, k);
    Y4_CHECK_HIP(hipGetLastError());
    return Y4_OK;
}

constexpr int F32_TILES = 12;

// chain heads: the tiles with one wave column over 64 channels
static bool chain_tile(int tile) { return tile == 3 || tile == 4 || tile == 15; }

// LDS-pair heads: 128-byte K rows, 2 stages, one channel tile over all of Cout (128 or 256), tile + tail stages in LDS
static bool pair_tile(int tile) {
    switch (tile) { case 1: case 8: case 20: case 24: case 25: case 29: case 13: case 19: case 21: case 22: return true; }
    return false;
}

template <int DT>
static int launch_dt(int tile, const ConvK& k, hipStream_t s) {
    if (k.pair) {
        if constexpr (DT != Y4_F32) {
#define Y4_PAIR_CASE(id, bm, bn, wm, wn) case id: return launch_cfg<DT, bm, bn, wm, wn, 128, 2, 0, true>(k, s);
            switch (tile) {
                Y4_PAIR_CASE(1, 128, 128, 2, 2) Y4_PAIR_CASE(8, 64, 128, 1, 4) Y4_PAIR_CASE(20, 96, 128, 2, 2)
                Y4_PAIR_CASE(24, 160, 128, 2, 2) Y4_PAIR_CASE(25, 192, 128, 2, 2) Y4_PAIR_CASE(29, 112, 128, 1, 4)
                Y4_PAIR_CASE(13, 128, 256, 2, 4) Y4_PAIR_CASE(19, 192, 256, 2, 4) Y4_PAIR_CASE(21, 96, 256, 2, 4)
                Y4_PAIR_CASE(22, 160, 256, 2, 4)
            }
#undef Y4_PAIR_CASE
        }
        set_error("conv2d: tile id %d cannot head an LDS pair", tile);
        return Y4_EINVAL;
    }
    if (k.ntail > 0) {
        if constexpr (DT != Y4_F32) {
            const int cfg = k.ntail == 1 ? 1 : (k.tail[1].cout == 64 ? 2 : 3);
#define Y4_CHAIN_CASE(CFG)                                                                   \
    case CFG:                                                                                \
        switch (tile) {                                                                      \
            case 3: return launch_cfg<DT, 128, 64, 4, 1, 128, 2, CFG>(k, s);                 \
            case 4: return launch_cfg<DT, 128, 64, 4, 1, 64, 2, CFG>(k, s);                  \
            case 15: return launch_cfg<DT, 128, 64, 4, 1, 64, 4, CFG>(k, s);                 \
        }                                                                                    \
        break;
            switch (cfg) { Y4_CHAIN_CASE(1) Y4_CHAIN_CASE(2) Y4_CHAIN_CASE(3) }
#undef Y4_CHAIN_CASE
        }
        set_error("conv2d: tile id %d cannot head a chain", tile);
        return Y4_EINVAL;
    }
    // the fp32 (parity) path instantiates only the first F32_TILES configurations (build time)
#define Y4_TILE_CASE(id, bm, bn, wm, wn, bkb, nst)                                            \
    case id:                                                                                  \
        if constexpr (DT == Y4_F32 && (id > F32_TILES)) break;                                \
        else return launch_cfg<DT, bm, bn, wm, wn, bkb, nst>(k, s);
    switch (tile) { Y4_TILES(Y4_TILE_CASE) }
    set_error("conv2d: tile id %d is not available for this dtype", tile);
    return Y4_EINVAL;
}

static bool tile_ok(const TileCfg& tc, int dtype, int cin, int cout_pad) {
    const int bk = tc.bkb / elem_size(dtype);
    return cin % bk == 0 && cout_pad % tc.bn == 0;
}

// Heuristic tile choice (speed only; every valid tile gives the same result up to fp32 summation order
// -- and the K order is identical across tiles, so results are in fact bitwise equal).
int conv_pick_tile(int dtype, int M, int cin, int cout) {
    const int es = elem_size(dtype);
    const bool k128 = cin % (128 / es) == 0;
    if (cout <= 32) return k128 ? 5 : 6;
    if (cout <= 64) return k128 ? 3 : 4;
    return k128 ? 8 : 9;      // 64x128: more, smaller blocks overlap load / MFMA / store phases best (measured)
}

int conv2d_launch(const y4_conv_desc* d, const char* zero_page, hipStream_t stream, const ConvChainDesc* chain,
                  const ConvPairDesc* pair) {
    Y4_REQUIRE(d && d->in && d->wt && d->out && d->scale && d->shift, Y4_EINVAL, "conv2d: null pointer");
    Y4_REQUIRE(d->dtype >= Y4_F32 && d->dtype <= Y4_F16, Y4_EINVAL, "conv2d: bad dtype %d", d->dtype);
    Y4_REQUIRE(d->ksize == 1 || d->ksize == 3, Y4_EINVAL, "conv2d: ksize %d (only 1 or 3)", d->ksize);
    Y4_REQUIRE(d->stride == 1 || (d->stride == 2 && d->ksize == 3), Y4_EINVAL, "conv2d: stride %d", d->stride);
    Y4_REQUIRE(d->stride == 1 || (d->h % 2 == 0 && d->w % 2 == 0), Y4_EINVAL, "conv2d: stride 2 needs even h,w");
    Y4_REQUIRE(d->n > 0 && d->h > 0 && d->w > 0 && d->cin > 0 && d->cout > 0, Y4_EINVAL, "conv2d: empty shape");
    const int es = elem_size(d->dtype);
    const int epc = 16 / es;
    Y4_REQUIRE(d->cin % (64 / es) == 0, Y4_EINVAL, "conv2d: cin %d must be a multiple of %d for dtype %d",
               d->cin, 64 / es, d->dtype);
    Y4_REQUIRE(d->in_cstride % epc == 0 && d->in_coff % epc == 0, Y4_EINVAL, "conv2d: input view not 16-byte aligned");
    const int oepc = d->out_f32 ? 4 : epc;
    Y4_REQUIRE(d->out_cstride % oepc == 0 && d->out_coff % oepc == 0, Y4_EINVAL,
               "conv2d: output view not 16-byte aligned");
    Y4_REQUIRE(!d->res || (d->res_cstride % epc == 0 && d->res_coff % epc == 0), Y4_EINVAL,
               "conv2d: residual view not 16-byte aligned");
    ConvK k{};
    k.in = (const char*)d->in; k.wt = (const char*)d->wt; k.scale = d->scale; k.shift = d->shift;
    k.res = (const char*)d->res; k.out = (char*)d->out; k.zero = zero_page;
    k.N = d->n; k.H = d->h; k.W = d->w; k.Cin = d->cin;
    k.Ho = d->h / d->stride; k.Wo = d->w / d->stride;
    k.cout_store = (int)round_up(d->cout, 8);
    Y4_REQUIRE((int64_t)d->n * k.Ho * k.Wo < (1ll << 31), Y4_EINVAL, "conv2d: too many output pixels");
    k.M = d->n * k.Ho * k.Wo;
    k.K = d->ksize * d->ksize * d->cin;
    const int64_t in_bytes = (int64_t)d->n * d->h * d->w * d->in_cstride * es;
    const int64_t wt_bytes = (int64_t)round_up(d->cout, COUT_PAD) * k.K * es;
    Y4_REQUIRE(in_bytes < (1ll << 31) && wt_bytes < (1ll << 31), Y4_EINVAL,
               "conv2d: input (%lld B) or weights (%lld B) exceed the 2 GiB buffer-descriptor range", (long long)in_bytes,
               (long long)wt_bytes);
    k.in_bytes = (unsigned)in_bytes; k.wt_bytes = (unsigned)wt_bytes;
    k.div_howo = fastdiv_make((uint32_t)(k.Ho * k.Wo)); k.div_wo = fastdiv_make((uint32_t)k.Wo);
    k.in_cstride = d->in_cstride; k.in_coff = d->in_coff;
    k.out_cstride = d->out_cstride; k.out_coff = d->out_coff;
    k.res_cstride = d->res_cstride; k.res_coff = d->res_coff;
    k.ksize = d->ksize; k.stride = d->stride; k.pad = d->ksize == 3 ? 1 : 0;
    k.act = d->act; k.upsample = d->upsample; k.out_f32 = d->out_f32;
    const int cout_pad = (int)round_up(d->cout, COUT_PAD);
    if (pair) {
        Y4_REQUIRE(!chain && d->dtype != Y4_F32 && (d->cout == 128 || d->cout == 256) && pair->cout >= 1 && pair->cout <= d->cout && !d->upsample &&
                       !d->out_f32 && (!d->out2 || (d->split == 64 && d->cout == 128)) && pair->w && pair->scale && pair->shift && pair->fin &&
                       pair->fin_cstride % (pair->out_f32 ? 4 : epc) == 0 && pair->fin_coff % (pair->out_f32 ? 4 : epc) == 0,
                   Y4_EINVAL, "conv2d: bad LDS-pair description");
        k.pair = pair->out_f32 ? 3 : 1; k.tail_act = pair->act; k.store_x = pair->store_x;
        k.tail[0].w = (const char*)pair->w; k.tail[0].scale = pair->scale; k.tail[0].shift = pair->shift; k.tail[0].cout = (int)round_up(pair->cout, 8);
        // a fused CSP pair as head: the tail reads its main-in half (rows >= split), i.e. the LDS panels from split/64 on
        k.tail_k = d->out2 ? d->cout - d->split : d->cout;
        k.tail_panel0 = d->out2 ? d->split / 64 : 0;
        k.tail_w_bytes = (unsigned)(round_up(pair->cout, COUT_PAD) * k.tail_k * es);
        k.fin = (char*)pair->fin; k.fin_cstride = pair->fin_cstride; k.fin_coff = pair->fin_coff;
        Y4_REQUIRE(!pair->fin2 || (pair->split > 0 && pair->split % 32 == 0 && pair->split < pair->cout && !pair->out_f32 &&
                                   pair->fin2_cstride % epc == 0 && pair->fin2_coff % epc == 0),
                   Y4_EINVAL, "conv2d: bad LDS-pair split description");
        k.fin2 = (char*)pair->fin2; k.fin2_cstride = pair->fin2_cstride; k.fin2_coff = pair->fin2_coff;
        k.tail_split = pair->fin2 ? pair->split : 0;
    }
    if (chain && chain->ntail > 0) {
        Y4_REQUIRE(d->dtype != Y4_F32 && d->cout == 64 && d->act == Y4_ACT_MISH && !d->upsample && !d->out_f32 && !d->out2 &&
                       chain->ntail <= 2 && chain->fin && chain->fin_cstride % epc == 0 && chain->fin_coff % epc == 0,
                   Y4_EINVAL, "conv2d: bad chain description");
        k.ntail = chain->ntail; k.store_x = chain->store_x;
        k.fin = (char*)chain->fin; k.fin_cstride = chain->fin_cstride; k.fin_coff = chain->fin_coff;
        for (int t = 0; t < chain->ntail; ++t) {
            const auto& ct = chain->tail[t];
            Y4_REQUIRE(ct.w && ct.scale && ct.shift && (ct.cout == 64 || (ct.cout == 128 && t == chain->ntail - 1)) &&
                           ((t == 1) == (ct.src2 != nullptr)) && (!ct.src2 || (ct.src2_cstride % epc == 0 && ct.src2_coff % epc == 0)),
                       Y4_EINVAL, "conv2d: bad chain tail %d", t);
            k.tail[t].w = (const char*)ct.w; k.tail[t].scale = ct.scale; k.tail[t].shift = ct.shift;
            k.tail[t].src2 = (const char*)ct.src2; k.tail[t].src2_cstride = ct.src2_cstride; k.tail[t].src2_coff = ct.src2_coff;
            k.tail[t].cout = ct.cout;
        }
    }
    int tile = d->tile ? d->tile : pair ? (d->cout == 128 ? 20 : 19) : (k.ntail > 0 ? (d->cin % (128 / es) == 0 ? 3 : 4) : conv_pick_tile(d->dtype, k.M, d->cin, d->cout));
    Y4_REQUIRE(k.ntail == 0 || chain_tile(tile), Y4_EINVAL, "conv2d: tile id %d cannot head this chain", tile);
    Y4_REQUIRE(!pair || (pair_tile(tile) && kTiles[tile - 1].bn == d->cout), Y4_EINVAL, "conv2d: tile id %d cannot head this LDS pair", tile);
    Y4_REQUIRE(tile >= 1 && tile <= kNumTiles, Y4_EINVAL, "conv2d: tile id %d out of range", tile);
    const TileCfg& tc = kTiles[tile - 1];
    Y4_REQUIRE(tile_ok(tc, d->dtype, d->cin, cout_pad), Y4_EINVAL,
               "conv2d: tile %d (bk bytes %d, bn %d) does not fit cin %d / cout_pad %d", tile, tc.bkb, tc.bn,
               d->cin, cout_pad);
    k.grid_m = (k.M + tc.bm - 1) / tc.bm;
    k.grid_n = (int)((round_up(d->cout, 8) + tc.bn - 1) / tc.bn);
    Y4_REQUIRE((int64_t)k.grid_n * tc.bn <= cout_pad, Y4_EINVAL, "conv2d: tile %d overruns the packed weight rows", tile);
    k.div_gridn = fastdiv_make((uint32_t)k.grid_n);
    k.out2 = (char*)d->out2; k.out2_cstride = d->out2_cstride; k.out2_coff = d->out2_coff; k.split = d->out2 ? d->split : 0;
    Y4_REQUIRE(!d->out2 || (d->split > 0 && d->split % 32 == 0 && d->split < d->cout && !d->res && !d->upsample && !d->out_f32 &&
                            d->out2_cstride % epc == 0 && d->out2_coff % epc == 0),
               Y4_EINVAL, "conv2d: bad split-output description");
    switch (d->dtype) {
        case Y4_F32: return launch_dt<Y4_F32>(tile, k, stream);
        case Y4_BF16: return launch_dt<Y4_BF16>(tile, k, stream);
        default: return launch_dt<Y4_F16>(tile, k, stream);
    }
}

// ------------------------------------------------------------------------------ weight re-layout
template <int DT>
__global__ void pack_conv_kernel(const float* __restrict__ w, typename Elem<DT>::type* __restrict__ out, int cout,
                                 int cout_pad, int cin, int kk) {
    const int64_t total = (int64_t)cout_pad * kk * cin;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % cin);
        const int64_t r = i / cin;
        const int tap = (int)(r % kk), co = (int)(r / kk);
        const float v = co < cout ? w[((int64_t)co * cin + ci) * kk + tap] : 0.f;
        out[i] = Elem<DT>::st(v);
    }
}

int pack_conv_weights(int dtype, int cout, int cin, int ksize, const float* oihw, void* packed, hipStream_t stream) {
    const int cout_pad = (int)round_up(cout, COUT_PAD), kk = ksize * ksize;
    const int64_t total = (int64_t)cout_pad * kk * cin;
    const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    switch (dtype) {
        case Y4_F32: hipLaunchKernelGGL(pack_conv_kernel<Y4_F32>, dim3(blocks), dim3(256), 0, stream, oihw, (float*)packed, cout, cout_pad, cin, kk); break;
        case Y4_BF16: hipLaunchKernelGGL(pack_conv_kernel<Y4_BF16>, dim3(blocks), dim3(256), 0, stream, oihw, (uint16_t*)packed, cout, cout_pad, cin, kk); break;
        case Y4_F16: hipLaunchKernelGGL(pack_conv_kernel<Y4_F16>, dim3(blocks), dim3(256), 0, stream, oihw, (_Float16*)packed, cout, cout_pad, cin, kk); break;
        default: set_error("pack_conv_weights: bad dtype %d", dtype); return Y4_EINVAL;
    }
    Y4_CHECK_HIP(hipGetLastError());
    return Y4_OK;
}

// 1x1 conv weights (cout, cin, 1, 1) -> ready-made MFMA A fragments for conv_chain.h (natural K order, chunked
// output-channel layout): out[((s*NREP2 + j2)*64 + lane)][e] = W[ch2][32*s + 8*(lane>>4) + e],
// ch2 = ((j2>>1)*4 + (i>>2))*8 + (j2&1)*4 + (i&3), i = lane & 15, NREP2 = cout/16, s < cin/32.
template <int DT>
__global__ void pack_tail_kernel(const float* __restrict__ w, typename Elem<DT>::type* __restrict__ out, int cout, int cin) {
    const int total = cout * cin, nrep2 = cout / 16;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int e = idx & 7, lane = (idx >> 3) & 63;
        const int r = idx >> 9;
        const int j2 = r % nrep2, s = r / nrep2;
        const int i = lane & 15, g = lane >> 4;
        const int ch2 = ((j2 >> 1) * 4 + (i >> 2)) * 8 + (j2 & 1) * 4 + (i & 3);
        out[idx] = Elem<DT>::st(w[ch2 * cin + 32 * s + 8 * g + e]);
    }
}

int pack_tail_weights(int dtype, int cout, int cin, const float* oihw, void* packed, hipStream_t stream) {
    Y4_REQUIRE((cout == 32 || cout == 64 || cout == 128) && (cin == 64 || cin == 128) && oihw && packed, Y4_EINVAL,
               "pack_tail_weights: cout %d cin %d", cout, cin);
    const int blocks = (cout * cin + 255) / 256;
    switch (dtype) {
        case Y4_BF16: hipLaunchKernelGGL(pack_tail_kernel<Y4_BF16>, dim3(blocks), dim3(256), 0, stream, oihw, (uint16_t*)packed, cout, cin); break;
        case Y4_F16: hipLaunchKernelGGL(pack_tail_kernel<Y4_F16>, dim3(blocks), dim3(256), 0, stream, oihw, (_Float16*)packed, cout, cin); break;
        default: set_error("pack_tail_weights: 16-bit dtypes only (got %d)", dtype); return Y4_EINVAL;
    }
    Y4_CHECK_HIP(hipGetLastError());
    return Y4_OK;
}

}  // namespace y4
