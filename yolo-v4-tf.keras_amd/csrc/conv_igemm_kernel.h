// conv_igemm_kernel.h -- the reference's conv() unit (custom_layers.py:5-31: Conv2D -> BatchNormalization ->
// Mish/LeakyReLU) as ONE im2col-free implicit-GEMM MFMA kernel for gfx950, with the graph's glue ops
// folded into its epilogue: residual Add (custom_layers.py:44), Concatenate (:68,:149,... -> a channel
// slice store), UpSampling2D (:147,:159 -> 2x2 replicated store).
//
// GEMM view (NHWC activations):  D[ch][px] = sum_k Wt[ch][k] * X[px][k],  k in the canonical order of common.h:
//   64-channel chunk -> tap -> channel (for Cin <= 64 and for 1x1 convs: tap -> channel).
//   X rows are gathered on the fly: for K-tile (chunk, tap, c0) row px is the BK contiguous channels
//   in[n, ho*s+ky-pad, wo*s+kx-pad, c0:c0+BK]  (an out-of-range offset -> zeros when the tap falls in the padding),
//   copied HBM/L2 -> LDS with buffer_load ... lds (16 B per lane, no VGPR round trip).
//   Weights are pre-packed along the same order ([cout_pad][Cin/KC][kh*kw][KC]) so their K-tile rows are contiguous too.
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
#pragma once
#include "conv_chain.h"
#include "conv_tiles.h"

namespace y4 {

#ifdef Y4_TRACE
// In-kernel phase trace (kernel experiments only; scripts/trace_read.py, LABNOTES.md section 4.1).  Build one translation
// unit with -DY4_TRACE=1 -DY4_TRACE_BM=<BM> -DY4_TRACE_NST=<NST> (scripts/build_variant.sh): workgroup TR_WG of the plain
// <BM> x 256 tile with <NST> stages records s_memtime (shader clock) per wave at fixed points of K-tiles TR_KT0 .. +3, and
// s_memrealtime (100 MHz) once per K-tile, into y4_trace_buf[k-tile][wave][point]; y4_trace_read() copies it out.
__device__ unsigned long long y4_trace_buf[4 * 8 * 8];
// ... and the workgroup's life outside the K loop: [wave][0 entry, 1 staging set-up done, 2 first K-tile's barrier passed,
// 3 K loop done, 4 epilogue done (stores issued), 5 s_memrealtime at entry, 6 s_memrealtime at the end]
__device__ unsigned long long y4_trace_life[8 * 8];
#define TR_LIFE(P, INSN)                                                                                    \
    do {                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        if (tr_life_on) asm volatile(INSN " %0" : "=s"(tr_life[P]));                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
    } while (0)
#define TR_KT0 10
#define TR_WG 8
#define TR_POINT(P)                                                                                         \
    do {                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        if (tr_on && kt >= TR_KT0 && kt < TR_KT0 + 4) asm volatile("s_memtime %0" : "=s"(tr_t[P]));         \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
    } while (0)
#else
#define TR_POINT(P)
#define TR_LIFE(P, INSN)
#endif

#ifndef Y4_EARLY_ARGS
#define Y4_EARLY_ARGS 1
#endif

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
    constexpr bool M32 = (NST == 32);        // 2 LDS stages, v_mfma_f32_32x32x16 (a different fp32 summation order: these
                                             // tiles agree with each other bit for bit, not with the 16x16x32 ones)
    constexpr int SN = (PHASED || M32) ? 2 : NST;     // LDS stages
    constexpr bool PREFRAG = true;           // all fragments of a K-tile are read before its first MFMA (measured: never slower)
    constexpr int STAGE = (BM + BN) * BKB;
    constexpr int KSTEPS = BKB / 64;        // MFMA k-steps (4 chunks each) per tile
    static_assert(BM % 16 == 0 && (BN % RPI == 0 || (RPI % BN == 0 && (BN * CPR) % 64 == 0)), "tile rows vs rows-per-iteration");
    // LDS: the K loop's stages (+ a chain's weight fragments) | TOUCH_LDS bytes that only weight_touch() writes.  An LDS pair's
    // tail region (launch_cfg: lds_pair) overlays all of it after the K loop, when the touch has long landed.
    constexpr int LDS_MAIN = SN * STAGE + (CHAIN ? ChainShape<CHAIN ? CHAIN : 1>::LDS_BYTES : 0);
    static_assert(MREP >= 1 && NREP >= 1, "wave tile");
    constexpr int LPT = A_IT + B_IT;        // LDS-DMA instructions a wave issues per stage
    static_assert(SN >= 2 && SN <= 7 && (SN == 2 || !B_PART), "deep pipelines need uniform weight loads per wave");
    static_assert((SN - 2) * LPT <= 63, "vmcnt field");
    static_assert(!PHASED || (NT == 512 && KSTEPS == 2), "phased schedule: 8 waves, 128-byte K rows");
    static_assert(!M32 || (DT != Y4_F32 && BKB == 128 && WPX % 32 == 0 && WCH % 32 == 0 && CHAIN == 0 && !PAIR && !B_PART),
                  "32x32x16 tiles: 16-bit, 128-byte K rows, wave tile in 32x32 blocks, plain launches");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    // The prologue's scalar fields in ONE round trip: hipcc fetches a kernel argument where it is first needed, and the shape fields
    // (first used behind the tile-mapping branches) came one scalar-cache miss after the pointers -- a cold miss per workgroup that a
    // short kernel feels (a single image: ~90 of them back to back).  Naming them here puts their loads into the first batch.
#if Y4_EARLY_ARGS
    asm volatile("" ::"s"(p.H), "s"(p.W), "s"(p.Cin), "s"(p.Ho), "s"(p.Wo), "s"(p.M), "s"(p.K), "s"(p.in_cstride), "s"(p.in_coff), "s"(p.ksize),
                 "s"(p.stride), "s"(p.pad), "s"(p.grid_m), "s"(p.grid_n), "s"(p.in_bytes), "s"(p.wt_bytes), "s"(p.ksplit), "s"(p.touch));
#endif

    // ---- XCD-aware tile mapping: block b runs on XCD b%8; give each XCD a contiguous run of tiles with the
    //      channel tile fastest, so blocks sharing an activation row panel share an L2 (speed only).
    // Split-K (p.ksplit = S > 1, plain launches at small batch): the S splits of a tile are S consecutive blocks of ONE XCD (their
    // partial sums then meet in one L2); an XCD with fewer tiles than its neighbours leaves its surplus blocks idle.
    const int nwg = p.grid_m * p.grid_n;
    int t, ks = 0;
    {
        const int b = blockIdx.x, qq = nwg >> 3, rr = nwg & 7, xcd = b & 7;
        int idx = b >> 3;
        if constexpr (CHAIN == 0 && !PAIR && NST != 12 && NST != 32) {
            if (p.ksplit > 1) {
                ks = idx % p.ksplit;
                idx /= p.ksplit;
                if (idx >= qq + (xcd < rr ? 1 : 0)) return;       // (the whole block: uniform)
            }
        }
        t = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
    }
    const int tile_m = (int)fastdiv((uint32_t)t, p.div_gridn), tile_n = t - tile_m * p.grid_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
#ifdef Y4_TRACE
#ifndef Y4_TRACE_PAIR
#define Y4_TRACE_PAIR 0            // 1: the life trace of the LDS-pair head of that tile instead (points 3 head K loop done, 7 tile in LDS, 2' -> [2] unchanged, 4 all stores issued)
#endif
    const bool tr_life_on = blockIdx.x == TR_WG && BM == Y4_TRACE_BM && BN == 256 && NST == Y4_TRACE_NST && CHAIN == 0 && PAIR == (Y4_TRACE_PAIR != 0);
    unsigned long long tr_life[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    TR_LIFE(0, "s_memtime");
    TR_LIFE(5, "s_memrealtime");
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
    // chunk swizzle of an LDS row: by the row's low bits (conv_common.h: swz); the 32x32x16 tiles also flip bit 0 with row
    // bit 4 -- their fragment reads put rows r and r + 16 (not r and r + 8 with the next chunk) into one ds_read_b128 lane group
    auto tswz = [](int row) { return M32 ? (swz<CPR>(row) ^ ((row >> 4) & 1)) : swz<CPR>(row); };
    int a_off[A_IT], a_mask[A_IT];         // a_mask: bit ky*ksize + kx = that tap of the row reads inside the image (else zeros)
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int j = 0; j < A_IT; ++j) {
        const int row = r0 + j * RPI;
        const int m = m0 + row;
        const int mm = m < p.M ? m : 0;
        const int n = (int)fastdiv((uint32_t)mm, p.div_howo), rem = mm - n * HoWo;
        const int ho = (int)fastdiv((uint32_t)rem, p.div_wo), wo = rem - ho * p.Wo;
        const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
        // (relative to a descriptor base moved back by one row + one pixel, `tap_bias`: never negative, so that the tap's offset --
        //  uniform over the lanes -- can ride in the scalar offset and a valid row's vector offset never changes)
        a_off[j] = (((n * p.H + hi0 + 1) * p.W + wi0 + 1) * p.in_cstride + p.in_coff + ((q ^ tswz(row)) * EPC)) * ES;
        // (three row bits x three column bits: a short-K layer -- conv 8 walks nine K-tiles per tile -- feels a prologue of 9 x 2 compares)
        int mask = 1;
        if (p.ksize == 3) {
            const int cols = ((unsigned)wi0 < (unsigned)p.W ? 1 : 0) | ((unsigned)(wi0 + 1) < (unsigned)p.W ? 2 : 0) | ((unsigned)(wi0 + 2) < (unsigned)p.W ? 4 : 0);
            mask = ((unsigned)hi0 < (unsigned)p.H ? cols : 0) | ((unsigned)(hi0 + 1) < (unsigned)p.H ? cols << 3 : 0) |
                   ((unsigned)(hi0 + 2) < (unsigned)p.H ? cols << 6 : 0);
        }
        a_mask[j] = m < p.M ? mask : 0;        // rows past M never validate -> zeros
    }
    int b_off[B_IT];
#pragma unroll
    for (int j = 0; j < B_IT; ++j) {
        const int row = B_PART ? (r0 % BN) : (r0 + j * RPI);
        // LDS row (wave block, fragment jn, MFMA row i = g*4 + r)  <-  channel of the chunked layout (conv_common.h)
        const int wb = row / WCH, pr = row - wb * WCH;
        int ch;
        if constexpr (M32) {
            // MFMA row R = 8g + 4h + j of 32-row block jb is accumulator value 4g + j of lane half h: chunk 2 jb + (g >> 1)
            // of that half (chunk_channel_g<2>), element 4 (g & 1) + j
            const int jb = pr >> 5, R = pr & 31, g = R >> 3, hh = (R >> 2) & 1, jj = R & 3;
            ch = chunk_channel_g<2>(n0 + wb * WCH, 2 * jb + (g >> 1), hh) + (g & 1) * 4 + jj;
        } else {
            const int jn = pr >> 4, i = pr & 15, g = i >> 2, r = i & 3;
            ch = chunk_channel(n0 + wb * WCH, jn >> 1, g) + (jn & 1) * 4 + r;
        }
        b_off[j] = (ch * p.K + ((q ^ tswz(row)) * EPC)) * ES;
    }
    const int tap_bias = (p.W + 1) * p.in_cstride * ES;
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(p.in - tap_bias, p.in_bytes + (unsigned)tap_bias);
    const __amdgpu_buffer_rsrc_t rs_wt = make_rsrc(p.wt, p.wt_bytes);
    const int wave_lds = __builtin_amdgcn_readfirstlane(wave * 1024);          // provably uniform -> SALU/M0 path

    // wave-uniform: does this wave skip the (partial) last A iteration?  (rows r0 + (A_IT-1)*RPI >= BM)
    const bool a_skip = A_PART && __builtin_amdgcn_readfirstlane(r0 + (A_IT - 1) * RPI >= BM ? 1 : 0) != 0;
    // staging cursor over the canonical K order (common.h): chunk of `chb` bytes of channels -> tap -> K-tile inside the chunk
    const int chb = k_chunk_channels(p.Cin, p.ksize) * ES;
    int ky = 0, kx = 0, cbase = 0, c_in = 0, ktb = 0;      // tap, byte offset of the chunk, of the K-tile in it, of k in the weights
    int nk = p.K / BK;                             // K-tiles this block walks
    if constexpr (CHAIN == 0 && !PAIR && NST != 12 && NST != 32) {
        if (p.ksplit > 1) {                        // this split's K range: cursor to its first K-tile
            const int kt0 = (int)((int64_t)ks * nk / p.ksplit), kt1 = (int)((int64_t)(ks + 1) * nk / p.ksplit);
            const int per = chb / BKB;             // K-tiles per (chunk, tap)
            const int idx = kt0 / per, taps = p.ksize * p.ksize;
            const int chunk = idx / taps, tap = idx - chunk * taps;
            ky = tap / p.ksize; kx = tap - ky * p.ksize;
            cbase = chunk * chb;
            c_in = (kt0 - idx * per) * BKB;
            ktb = kt0 * BKB;
            nk = kt1 - kt0;
        }
    }
    // Per tap change (every K-tile of a 3x3 conv with 128-byte rows since the K order is chunk-major; a wave issues one instruction
    // per ~5-8 cycles, so both a single image's latency-bound small tiles and the big tiles feel every instruction here --
    // scripts/conv_b1_ab.py, and +4 % on the whole batch-32 step with the first form of this cursor):
    //   * the tap's offset (ky W + kx) pixels is uniform over the lanes: it rides in the SCALAR offset with the chunk's, so a row's
    //     vector offset is either its own constant a_off or the out-of-range value -- one bit-field extract + one select per row;
    //   * TAPVEC (deep-ring tiles that stage at most two rows per thread): even that is done once -- the nine selects of a row sit in a
    //     16-element register vector and the tap's entry is picked by a uniform index (s_set_gpr_idx).
    constexpr bool TAPVEC = A_IT <= 2 && SN >= 5;   // (the deep rings of a single image's latency-bound tiles; a short-K batch-32 layer on a
                                                    //  small tile would pay the table's set-up in its prologue: convs 11-16 +9 %, measured)
    typedef int i32x16 __attribute__((ext_vector_type(16)));
    const int cs = p.in_cstride * ES;
    const int step_row = (p.W - (p.ksize - 1)) * cs, step_back = -((p.ksize - 1) * p.W + (p.ksize - 1)) * cs;
    int tap = ky * p.ksize + kx;
    int tap_off = (ky * p.W + kx) * cs;            // scalar: this tap's byte offset from the (biased) row offset
    int a_vo[A_IT];
    i32x16 a_tapv[TAPVEC ? A_IT : 1];
    if constexpr (TAPVEC) {
#pragma unroll
        for (int j = 0; j < A_IT; ++j)
#pragma unroll
            for (int t = 0; t < 9; ++t) a_tapv[j][t] = ((a_mask[j] >> t) & 1) ? a_off[j] : (int)0x80000000;
    }
    auto set_tap = [&]() {
#pragma unroll
        for (int j = 0; j < A_IT; ++j) {
            if constexpr (TAPVEC) a_vo[j] = a_tapv[j][tap];
            else a_vo[j] = ((a_mask[j] >> tap) & 1) ? a_off[j] : (int)0x80000000;      // >= num_records -> zeros
        }
    };
    set_tap();
    int c0b = cbase + c_in + tap_off;
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
        // The tap advance is the COMMON case since the K order is chunk-major: marked likely so that the block stays in line.  (As
        // scalar selects instead of branches the compiler moves the cursor into vector registers -- v_cndmask + v_readfirstlane, 300
        // scratch accesses: 6x slower, measured.)
        c_in += BKB;
        if (__builtin_expect(c_in >= chb, 1)) {     // next tap of this chunk; after the last tap the next chunk
            c_in = 0;
            ++tap;
            tap_off += cs;
            if (__builtin_expect(++kx >= p.ksize, 0)) {
                kx = 0;
                tap_off += step_row - cs;
                if (++ky >= p.ksize) { ky = 0; tap = 0; tap_off = 0; cbase += chb; }
            }
            set_tap();
        }
        c0b = cbase + c_in + tap_off;
    };

    // ---- weight touch (conv_common.h): this channel tile's weight block, BN rows x K, is contiguous
    if (p.touch != 0) {
        weight_touch(rs_wt, smem + LDS_MAIN, n0 * p.K * ES, BN * p.K * ES, wave, NT / 64, lane);
        if constexpr (PAIR)                    // ... and the tail conv's, needed only after the whole K loop
            weight_touch(make_rsrc(p.tail[0].w, p.tail_w_bytes), smem + LDS_MAIN, 0, (int)p.tail_w_bytes, wave, NT / 64, lane);
    }

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

    // small tiles (one chunk pair per lane: 16 registers): the BatchNorm tables' loads go out in front of the K loop instead of in
    // front of the epilogue, where their round trip is all a short kernel would be waiting for (a single image: ~90 such kernels)
    constexpr bool EARLY_TABLES = CHAIN == 0 && !PAIR && !M32 && !PHASED && NREP <= 2;
    float sc_early[NREP * 4], sh_early[NREP * 4];
    if constexpr (EARLY_TABLES) conv_epilogue_tables<NREP>(p, n0 + wn * WCH, fg, sc_early, sh_early);

    TR_LIFE(1, "s_memtime");
    if constexpr (M32) {
        // Same staging, same LDS image, same one-barrier 2-stage loop; the K-tile is 4 k-steps of 16 with 32x32 blocks:
        // half the MFMA instructions for the same FLOPs and the same fragment bytes.
        constexpr int MB = WPX / 32, NB = WCH / 32;
        const int frow32 = lane & 31, fh = lane >> 5;
        int xo32[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) xo32[ks] = frow32 * 128 + (((2 * ks + fh) ^ tswz(frow32)) * 16);
        f32x16 acc32[MB][NB];
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc32[i][j][e] = 0.f;
        stage(0);
        for (int kt = 0; kt < nk; ++kt) {
            wait_vmcnt_then_barrier<0>();
            if (kt + 1 < nk) stage((kt + 1) & 1);
            // (LDS byte addresses: the low 32 bits of a __shared__ pointer; the weight rows sit BM*BKB behind the pixel rows,
            // which goes into the instruction offset together with the block index)
            unsigned xa[4], wa[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                xa[ks] = (unsigned)(uintptr_t)(smem + (kt & 1) * STAGE + (wm * WPX) * BKB + xo32[ks]);
                wa[ks] = (unsigned)(uintptr_t)(smem + (kt & 1) * STAGE + (wn * WCH) * BKB + xo32[ks]);
            }
            // The 20 fragment reads and their waits are written out (inline asm): left to the compiler, the first MFMA of
            // the K-tile waits for lgkmcnt(0), i.e. for all 20 reads, instead of for its own k-step's five.
            u32x4 xf[4][MB], wf[4][NB];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int i = 0; i < MB; ++i)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(xf[ks][i]) : "v"(xa[ks]), "n"(i * 32 * 128));
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[ks][j]) : "v"(wa[ks]), "n"(BM * BKB + j * 32 * 128));
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                // k-step ks's MB + NB reads are complete once at most (3 - ks) * (MB + NB) are outstanding; the "+v" operands
                // keep this k-step's MFMAs behind the wait
                constexpr int R = MB + NB;                       // (the count field holds 15 at most: waiting for more is safe)
                if (ks == 0) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(xf[0][0]), "+v"(wf[0][0]) : "n"(3 * R > 15 ? 15 : 3 * R));
                if (ks == 1) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(xf[1][0]), "+v"(wf[1][0]) : "n"(2 * R > 15 ? 15 : 2 * R));
                if (ks == 2) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(xf[2][0]), "+v"(wf[2][0]) : "n"(R > 15 ? 15 : R));
                if (ks == 3) asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(xf[3][0]), "+v"(wf[3][0]) : "n"(0));
#pragma unroll
                for (int j = 0; j < NB; ++j)
#pragma unroll
                    for (int i = 0; i < MB; ++i) Mma32<DT>::run(acc32[i][j], wf[ks][j], xf[ks][i]);
                __builtin_amdgcn_sched_barrier(0);               // (MFMAs that do not read operand 0 would sink below the next wait)
            }
            __builtin_amdgcn_s_setprio(0);
        }
        // the shared epilogue on a view of the blocks as f32x4 fragments: value 4f + r of block (i, j) = fragment 4j + f,
        // i.e. chunk 2j + (f >> 1) of this lane half, element 4 (f & 1) + r
        f32x4 accv[MB][NB * 4];
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int f = 0; f < 4; ++f)
                    accv[i][j * 4 + f] = f32x4{acc32[i][j][4 * f], acc32[i][j][4 * f + 1], acc32[i][j][4 * f + 2], acc32[i][j][4 * f + 3]};
        const bool full32 = (m0 + BM <= p.M) && (n0 + BN <= p.cout_store);
        conv_epilogue<DT, MB, NB * 4, false, 2>(p, accv, m0 + wm * WPX + frow32, p.M, n0 + wn * WCH, fh, full32);
        return;
    } else if constexpr (PHASED) {
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
#ifdef Y4_TRACE
    const bool tr_on = blockIdx.x == TR_WG && BM == Y4_TRACE_BM && BN == 256 && NST == Y4_TRACE_NST && CHAIN == 0 && !PAIR;
    unsigned long long tr_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (int kt = 0; kt < nk; ++kt) {
        TR_POINT(0);                           // arrives at the K-tile barrier
#ifdef Y4_TRACE
        if (tr_on && kt >= TR_KT0 && kt < TR_KT0 + 4) asm volatile("s_memrealtime %0" : "=s"(tr_t[7]));
#endif
        const int ahead = nk - 1 - kt;         // tiles issued after kt so far (capped at NST-2)
#ifdef Y4_TRACE
        if (SN == 2 || ahead == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            TR_POINT(4);                       // this wave's own loads of tile kt have landed
            asm volatile("s_barrier" ::: "memory");
        } else
#endif
        if (a_skip) wait_vmcnt_then_barrier_k<SN - 2, LPT - 1>(ahead);       // (this wave issues one load fewer per stage: partial last A iteration)
        else wait_vmcnt_then_barrier_k<SN - 2, LPT>(ahead);
        TR_POINT(1);                           // past the barrier
        if (kt == 0) TR_LIFE(2, "s_memtime");
        // deep rings (a single image's latency-bound K loops: 4 MFMAs per wave and K-tile) issue the fragment reads FIRST: the next
        // stage's loads have five K-tiles of slack, the reads' round trip is on the K-tile's critical path
        constexpr bool READS_FIRST = SN >= 5;
        if (!READS_FIRST && kt + SN - 1 < nk) stage(nxt);
        TR_POINT(2);                           // next tile's loads issued (and the staging cursor advanced)
        const char* sx = lds_x + cur * STAGE;
        const char* sw = lds_w + cur * STAGE;
        const int nxt_now = nxt;
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
            TR_POINT(3);                       // fragment reads issued
            if (READS_FIRST && kt + SN - 1 < nk) stage(nxt_now);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk)
#pragma unroll
                for (int i = 0; i < MREP; ++i)
#pragma unroll
                    for (int j = 0; j < NREP; ++j) Mma<DT>::run(acc[i][j], wf[kk][j], xf[kk][i]);
            __builtin_amdgcn_s_setprio(0);
            TR_POINT(5);                       // last MFMA issued
#ifdef Y4_TRACE
            if (tr_on && kt >= TR_KT0 && kt < TR_KT0 + 4) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (lane == 0)
                    for (int q2 = 0; q2 < 8; ++q2) y4_trace_buf[((kt - TR_KT0) * 8 + wave) * 8 + q2] = tr_t[q2];
            }
#endif
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
        TR_LIFE(3, "s_memtime");
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
        TR_LIFE(7, "s_memtime");
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
#if defined(Y4_TRACE) && Y4_TRACE_PAIR
        TR_LIFE(0, "s_memtime");                           // (pair trace: slots 0 / 1 are re-used for the tail's K loop / epilogue)
#endif
        // tail epilogue through the ordinary path: a ConvK that describes the 1x1 conv's output side
        ConvK p2 = p;
        p2.scale = p.tail[0].scale; p2.shift = p.tail[0].shift; p2.act = p.tail_act; p2.res = nullptr;
        p2.out = p.fin; p2.out_cstride = p.fin_cstride; p2.out_coff = p.fin_coff;
        p2.cout_store = p.tail[0].cout; p2.upsample = 0; p2.out_f32 = p.pair >> 1;
        p2.split = p.tail_split; p2.out2 = p.fin2; p2.out2_cstride = p.fin2_cstride; p2.out2_coff = p.fin2_coff;
        p2.fast_epi = p.fast_tail; p2.out_bytes = p.fin_bytes; p2.out2_bytes = p.fin2_bytes;
        conv_epilogue<DT, MREP, NREP>(p2, acc2, mrow, p.M, chw, fg, (m0 + BM <= p.M) && BN <= p2.cout_store);
#if defined(Y4_TRACE) && Y4_TRACE_PAIR
        TR_LIFE(1, "s_memtime");
#endif
        if (p.store_x) pair_store_tile<DT, MREP, NREP>(p, xl, xrow, XPANEL, mrow, p.M, chw, fg);
        TR_LIFE(4, "s_memtime");
        TR_LIFE(6, "s_memrealtime");
#ifdef Y4_TRACE
        if (tr_life_on) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0)
                for (int q2 = 0; q2 < 8; ++q2) y4_trace_life[wave * 8 + q2] = tr_life[q2];
        }
#endif
    } else {
        TR_LIFE(3, "s_memtime");
        if constexpr (NST != 12 && NST != 32) {
            if (p.ksplit > 1) {
                // split-K: every split leaves its raw accumulators in `part` (one wave-wide 1 KB row per fragment); the split that
                // arrives LAST at the tile's counter adds all of them in split order -- a fixed fp32 summation order whoever is
                // last, but another one than the unsplit K loop's -- and runs the ordinary epilogue.  The counter is back at zero
                // when the kernel ends (the next split-K launch re-uses it).
                // Round 4: the partial sums travel as AGENT-scope accesses (sc1: coherent between the XCDs' L2s by themselves) and the
                // release is "my stores have completed" (vmcnt(0)) -- where __threadfence() made every wave of every split write back
                // its XCD's whole L2 (buffer_wbl2: megabytes of dirty activations) and invalidate it: +39 .. +97 us on a 19^2 3x3 at fp32.
                constexpr int FR = MREP * NREP;
                f32x4* const mine = (f32x4*)p.part + ((size_t)ks * nwg + t) * FR * NT + tid;
#pragma unroll
                for (int i = 0; i < MREP; ++i)
#pragma unroll
                    for (int j = 0; j < NREP; ++j) store16_agent(mine + (i * NREP + j) * NT, acc[i][j]);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // release: the partial sums before the counter
                __syncthreads();                   // (also: every wave is done with the K loop's LDS)
                int* const flag = (int*)smem;
                if (tid == 0) {
                    const int old = atomicAdd(p.split_cnt + t, 1);
                    if (old == p.ksplit - 1) p.split_cnt[t] = 0;
                    *flag = old == p.ksplit - 1;
                }
                __syncthreads();
                if (*flag == 0) return;
                // acquire: every other split's stores completed before its counter increment; agent-scope loads read them
                const f32x4* src = (const f32x4*)p.part + (size_t)t * FR * NT + tid;
                for (int sp = 0; sp < p.ksplit; ++sp) {
#pragma unroll
                    for (int i = 0; i < MREP; ++i) {               // (a pixel row of fragments at a time: NREP registers x 4, not the tile's)
                        f32x4 part[NREP];
#pragma unroll
                        for (int j = 0; j < NREP; ++j) part[j] = load16_agent(src + (i * NREP + j) * NT);
#pragma unroll
                        for (int j = 0; j < NREP; ++j) acc[i][j] = sp == 0 ? part[j] : acc[i][j] + part[j];
                    }
                    src += (size_t)nwg * FR * NT;
                }
            }
        }
        const bool full = (m0 + BM <= p.M) && (n0 + BN <= p.cout_store);
        if constexpr (EARLY_TABLES) conv_epilogue_with<DT, MREP, NREP>(p, acc, sc_early, sh_early, m0 + wm * WPX + frow, p.M, n0 + wn * WCH, fg, full);
        else conv_epilogue<DT, MREP, NREP>(p, acc, m0 + wm * WPX + frow, p.M, n0 + wn * WCH, fg, full);
        TR_LIFE(4, "s_memtime");
        TR_LIFE(6, "s_memrealtime");
#ifdef Y4_TRACE
        if (tr_life_on) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0)
                for (int q2 = 0; q2 < 8; ++q2) y4_trace_life[wave * 8 + q2] = tr_life[q2];
        }
#endif
    }
}

// ------------------------------------------------------------------------------------------- launch
template <int DT, int BM, int BN, int WM, int WN, int BKB, int NST, int CHAIN = 0, bool PAIR = false>
static int launch_cfg(const ConvK& k, hipStream_t stream) {
    // (the kernel's LDS_MAIN + the weight touch's own scratch)
    constexpr int lds_main = (NST == 12 || NST == 32 ? 2 : NST) * (BM + BN) * BKB + (CHAIN ? ChainShape<CHAIN ? CHAIN : 1>::LDS_BYTES : 0) + TOUCH_LDS;
    constexpr int lds_pair = PAIR ? (BN / 64) * BM * 128 + 2 * BN * 128 : 0;     // tile panels + two weight stages of the tail
    constexpr int lds = lds_main > lds_pair ? lds_main : lds_pair;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = conv_igemm_kernel<DT, BM, BN, WM, WN, BKB, NST, CHAIN, PAIR>;
    if (lds > 48 * 1024) {
        static PerDeviceOnce once;
        if (const uint64_t bit = once.due()) {
            Y4_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            once.mark(bit);
        }
    }
    if constexpr (CHAIN == 0 && !PAIR && NST != 12 && NST != 32) {
        if (k.ksplit > 1) {
            const int nwg = k.grid_m * k.grid_n, per_xcd = (nwg >> 3) + ((nwg & 7) ? 1 : 0);
            hipLaunchKernelGGL(kern, dim3(8 * per_xcd * k.ksplit), dim3(64 * WM * WN), lds, stream, k);
            Y4_CHECK_HIP(hipGetLastError());
            return Y4_OK;
        }
    }
    Y4_REQUIRE(k.ksplit <= 1, Y4_EINVAL, "conv2d: this tile / fusion cannot run split-K");
    hipLaunchKernelGGL(kern, dim3(k.grid_m * k.grid_n), dim3(64 * WM * WN), lds, stream, k);
    Y4_CHECK_HIP(hipGetLastError());
    return Y4_OK;
}

// the phased kernel (conv_p8_kernel.h) lives in its own translation units (conv_p8_<dt>.hip), the halo kernel
// (conv_halo_kernel.h) in conv_halo_<dt>.hip
int conv_p8_launch(int dtype, int bm, int nst, const ConvK& k, hipStream_t s);
int conv_halo_launch(int dtype, int bm, int bn, const ConvK& k, hipStream_t s);
int conv_halo2_launch(int dtype, int tile, const ConvK& k, hipStream_t s);      // conv_halo2_kernel.h, in conv_halo2_<dt>.hip

// plain tiles of one dtype (one translation unit per dtype: conv_igemm_<dt>.hip)
template <int DT>
static int launch_plain(int tile, const ConvK& k, hipStream_t s) {
    // the fp32 (parity) path instantiates only the first F32_TILES configurations (build time)
#define Y4_TILE_CASE(id, bm, bn, wm, wn, bkb, nst)                                            \
    case id:                                                                                  \
        if constexpr (DT == Y4_F32 && !f32_tile(id)) break;                                   \
        else if constexpr (DT == Y4_F32 && nst == 32) break;                                  \
        else if constexpr (nst == 8 || nst == 9 || nst == 10) return conv_p8_launch(DT, bm, nst, k, s); \
        else if constexpr (nst == 20) return conv_halo_launch(DT, bm, bn, k, s);              \
        else if constexpr (nst == 21) return conv_halo2_launch(DT, id, k, s);                 \
        else return launch_cfg<DT, bm, bn, wm, wn, bkb, nst>(k, s);
    switch (tile) { Y4_TILES(Y4_TILE_CASE) }
    set_error("conv2d: tile id %d is not available for this dtype", tile);
    return Y4_EINVAL;
}

// chain heads and LDS-pair heads of one 16-bit dtype (conv_igemm_<dt>_fused.hip)
template <int DT>
static int launch_fused(int tile, const ConvK& k, hipStream_t s) {
    if (k.pair) {
        if constexpr (DT != Y4_F32) {
#define Y4_PAIR_CASE(id, bm, bn, wm, wn) case id: return launch_cfg<DT, bm, bn, wm, wn, 128, 2, 0, true>(k, s);
            switch (tile) {
                Y4_PAIR_CASE(1, 128, 128, 2, 2) Y4_PAIR_CASE(8, 64, 128, 1, 4) Y4_PAIR_CASE(20, 96, 128, 2, 2)
                Y4_PAIR_CASE(24, 160, 128, 2, 2) Y4_PAIR_CASE(25, 192, 128, 2, 2) Y4_PAIR_CASE(29, 112, 128, 1, 4)
                Y4_PAIR_CASE(13, 128, 256, 2, 4) Y4_PAIR_CASE(19, 192, 256, 2, 4) Y4_PAIR_CASE(21, 96, 256, 2, 4)
                Y4_PAIR_CASE(22, 160, 256, 2, 4) Y4_PAIR_CASE(38, 384, 128, 4, 2)
                case 51: return conv_halo_launch(DT, 384, 128, k, s);        // the halo tiles as heads (conv_halo_kernel.h)
                case 52: return conv_halo_launch(DT, 192, 256, k, s);
                case 54: return conv_halo_launch(DT, 320, 128, k, s);
            }
#undef Y4_PAIR_CASE
        }
        set_error("conv2d: tile id %d cannot head an LDS pair", tile);
        return Y4_EINVAL;
    }
    if (k.ntail > 0) {
        if constexpr (DT != Y4_F32) {
            const int cfg = k.ntail == 1 ? 1 : (k.tail[0].w == nullptr ? 4 : (k.tail[1].cout == 64 ? 2 : 3));
#define Y4_CHAIN_CASE(CFG)                                                                   \
    case CFG:                                                                                \
        switch (tile) {                                                                      \
            case 3: return launch_cfg<DT, 128, 64, 4, 1, 128, 2, CFG>(k, s);                 \
            case 4: return launch_cfg<DT, 128, 64, 4, 1, 64, 2, CFG>(k, s);                  \
            case 15: return launch_cfg<DT, 128, 64, 4, 1, 64, 4, CFG>(k, s);                 \
        }                                                                                    \
        break;
            switch (cfg) { Y4_CHAIN_CASE(1) Y4_CHAIN_CASE(2) Y4_CHAIN_CASE(3) Y4_CHAIN_CASE(4) }
#undef Y4_CHAIN_CASE
        }
        set_error("conv2d: tile id %d cannot head a chain", tile);
        return Y4_EINVAL;
    }
    set_error("conv2d: not a fused launch");
    return Y4_EINVAL;
}

}  // namespace y4
