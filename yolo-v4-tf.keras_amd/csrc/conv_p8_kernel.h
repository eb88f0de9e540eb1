// conv_p8_kernel.h -- the reference's conv() unit (custom_layers.py:5-31) as a PHASED implicit-GEMM MFMA kernel for gfx950:
// same GEMM view, operand roles, LDS image, K order and epilogue as conv_igemm_kernel.h (bit-identical results), another
// schedule of the K loop.  Round 2's phase trace of the plain loop (LABNOTES.md section 4.1) showed the two waves of a SIMD
// paying their non-MFMA instruction streams at the same time after every barrier and every load having exactly one K-tile
// period to land.  Here
//   * a K-tile is NP = BM/64 phases; a phase is { read-half: fragment ds_reads + LDS-DMA issue | s_barrier | MFMA-half:
//     16 MFMAs on one 32-pixel x 64-channel block of the wave's tile | s_barrier };
//   * waves 4-7 (pixel half 1) run ONE half-phase behind waves 0-3 (they take one extra barrier first, waves 0-3 one at the
//     end): on every SIMD one wave multiplies while the other one reads and issues loads;
//   * LDS holds two K-tiles, recycled REGION by region: the 64 pixel rows a phase reads (and, with phase 0, the weight rows,
//     whose fragments then stay in registers for the whole K-tile) are refilled two phases after their last read with the
//     data of K-tile t+2, so loads fly 1 to 1 1/3 K-tiles ahead inside the LDS of a 2-stage ring;
//   * nothing ever drains: every wave issues exactly LPT = NP + 4 loads per K-tile in the same order, so ONE counted
//     `s_waitcnt vmcnt(LPT)` per phase retires the region the next phase reads (loads retire in order); past the last
//     K-tile the stream continues with out-of-range (zero-filled, traffic-free) loads so that the count stays uniform.
// Orderings (H = half-phase counter; waves 0-3 read phase P in H = 2P and multiply in 2P+1, waves 4-7 one later):
//   RAW  a region read in phase P+1 is waited for by waves 0-3 at the END of their MFMA-half of P and by waves 4-7 at the
//        end of their read-half of P -- both before the barrier that closes H = 2P+1, the first reader comes after it;
//   WAR  a region last read in phase P (reads complete: lgkmcnt(0) at the start of the readers' MFMA-half, i.e. at the latest
//        at the start of H = 2P+2) is refilled in the read-half of phase P+2 (H >= 2P+4): a full barrier in between.
#pragma once
#include <type_traits>

#include "conv_common.h"
#include "conv_tiles.h"

namespace y4 {

__device__ __forceinline__ void p8_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int N> __device__ __forceinline__ void p8_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// The issue / wait schedule as compile-time tables.  Issue slot of phase ph: region (ph - 2) mod NP -- pixel part r, plus
// the four weight parts with r = 0 -- of K-tile kt + 2 - late (late = the region index wrapped into the next K-tile).
template <int NP> struct P8Sched {
    static constexpr int region(int ph) { return (ph + NP - 2) % NP; }
    static constexpr int late(int ph) { return (region(ph) + 2) / NP; }
    static constexpr int loads(int ph) { return region(ph) == 0 ? 5 : 1; }
    // the wait of phase ph retires the region phase ph + 1 reads: loads issued behind it, up to and including phase ph's
    // own slot, may stay in flight (2 NP - 3 slots: NP = 3 -> always NP + 4, NP = 4 -> 9 or 13)
    static constexpr int wait(int ph) {
        const int s0 = (ph + 3) % NP;
        int n = 0;
        for (int d = 1; d <= 2 * NP - 3; ++d) n += loads((s0 + d) % NP);
        return n;
    }
    // the prologue runs the slots of the two virtual K-tiles before the first; all but K-tile 0's region 0 may stay in flight
    static constexpr int prologue_wait() {
        int n = 0;
        for (int vk = -2; vk < 0; ++vk)
            for (int ph = 0; ph < NP; ++ph)
                if (vk + 2 - late(ph) >= 0) n += loads(ph);
        return n - 5;
    }
};
template <int NP> __device__ __forceinline__ void p8_wait_phase(int ph) {      // ph is a constant after unrolling
    if (ph == 0) p8_wait_vm<P8Sched<NP>::wait(0)>();
    else if (ph == 1) p8_wait_vm<P8Sched<NP>::wait(1)>();
    else if (ph == 2) p8_wait_vm<P8Sched<NP>::wait(2)>();
    else p8_wait_vm<P8Sched<NP>::wait(NP == 4 ? 3 : 0)>();
}

// ---- SCHED = 9: the software-pipelined form.  All eight waves run the same stream (no stagger, ONE barrier per phase); a
// phase's non-MFMA instructions are placed BETWEEN its MFMAs (pinned with sched_barrier): the fragment reads of the NEXT
// phase's pixel part -- each into the register its last user has just released, order (k-step, pixel fragment, channel
// fragment) so that a fragment has a dozen MFMAs to land --, in the last phase of a K-tile the weight fragments of the next
// K-tile the same way, and the LDS-DMA loads.  A wave then never issues MFMAs back to back (one wave alone gets a
// 16x16x32 MFMA out only every ~25 cycles, LABNOTES.md section 4.0) and never stops multiplying to read.
// Region life cycle (G = global phase index; fragments used in phase G are read during phase G-1):
//   pixel part r of K-tile T: loaded in phase (T-2, r+1)  [r = NP-1: (T-1, 0)], retired by the wait at the end of the phase
//   before the one that reads it, read during phase (T, r) - 1, free two phases after that read;
//   weights of K-tile T: loaded in phase (T-2, 1), read during the last phase of K-tile T-1.
// Every region is waited for exactly one K-tile (NP = 3) or 1.25 K-tiles (NP = 4) after its issue; the counts come from a
// compile-time simulation of the issue order (S9Sched).
template <int NP> struct S9Sched {
    // loads issued in phase ph of any K-tile: pixel part (ph-1) mod NP, and with ph == 1 the four weight parts
    static constexpr int loads(int ph) { return ph == 1 ? 5 : 1; }
    // region ids: 0..NP-1 pixel parts, NP weights; (T, region) -> global index of its LAST load in the issue order
    static constexpr int first_load_of_phase(int g) {           // loads issued before global phase g (g >= 0, K-tile = g / NP)
        int n = 0;
        for (int i = 0; i < g; ++i) n += loads(i % NP);
        return n;
    }
    static constexpr int issue_phase(int T, int region) {       // global phase in which the region of K-tile T is issued
        return region == NP ? (T - 2) * NP + 1 : region == NP - 1 ? (T - 1) * NP : (T - 2) * NP + region + 1;
    }
    static constexpr int last_load(int T, int region) {         // index (in issue order) of the region's last load
        const int g = issue_phase(T, region);
        // phase 1 issues pixel part 0 first, then the four weight parts
        return first_load_of_phase(g) + (region == NP ? 4 : 0);
    }
    // wait at the end of phase ph: everything read during the next phase must have landed
    static constexpr int wait(int ph) {
        const int K0 = 4;                                       // a steady-state K-tile
        const int g = K0 * NP + ph, gr = g + 1, gu = g + 2;     // wait phase, reading phase, using phase
        int need = last_load(gu / NP, gu % NP);
        if (gr % NP == NP - 1) {                                // the last phase of a K-tile also reads the next K-tile's weights
            const int w = last_load(gr / NP + 1, NP);
            need = w > need ? w : need;
        }
        return first_load_of_phase(g + 1) - 1 - need;           // loads issued behind it, through phase g's own
    }
    static constexpr int prologue_loads() {                     // the slots of the virtual K-tiles -2 and -1 that target K-tiles >= 0
        return first_load_of_phase(2 * NP) - loads(0);          // all of them except K-tile -2's phase 0 (pixel part NP-1 of K-tile -1)
    }
};
template <int NP, int PH> __device__ __forceinline__ void s9_wait_phase() { p8_wait_vm<S9Sched<NP>::wait(PH)>(); }

// (Round 3 also built both schedules with v_mfma_f32_32x32x16: 115-117 us where the 16x16x32 forms take 96-101 -- like the
// plain kernel's 32x32x16 tiles, LABNOTES.md section 4.0 -- and removed them again.)
template <int DT, int BM, int SCHED>
__global__ __launch_bounds__(512, 1) void conv_p8_kernel(const ConvK p) {
    static_assert(SCHED == 8 || SCHED == 9, "8: staggered wave groups, 9: software-pipelined");
    static_assert(DT != Y4_F32, "16-bit dtypes");
    static_assert(BM == 192 || BM == 256, "pixel tile: 3 or 4 phases of 64 rows");
    constexpr int ES = 2, BKB = 128, BK = 64, EPC = 8;
    constexpr int NP = BM / 64;                    // phases per K-tile = 64-row pixel parts
    constexpr int WPX = BM / 2, WCH = 64;          // wave tile: 2 (pixels) x 4 (channels) waves
    constexpr int MREP = WPX / 16, NREP = 4;
    constexpr int A_TILE = BM * BKB, B_TILE = 256 * BKB, STAGE = A_TILE + B_TILE;
    constexpr int PART = 64 * BKB;                 // one block-wide load instruction = 64 rows
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // ---- XCD-aware tile mapping (as conv_igemm_kernel.h)
    const int nwg = p.grid_m * p.grid_n;
    int t;
    {
        const int b = blockIdx.x, qq = nwg >> 3, rr = nwg & 7, xcd = b & 7, idx = b >> 3;
        t = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
    }
    const int tile_m = (int)fastdiv((uint32_t)t, p.div_gridn), tile_n = t - tile_m * p.grid_n;
    const int m0 = tile_m * BM, n0 = tile_n * 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const bool grp1 = __builtin_amdgcn_readfirstlane(wm) != 0;

    // ---- staging set-up.  This thread copies chunk slot q of LDS row r0 of every 64-row part.  Pixel part j holds the rows
    // phase j reads: LDS row j*64 + w*32 + i  <-  pixel row w*WPX + j*32 + i of the tile (w = pixel half, i < 32).
    const int q = tid & 7, r0 = tid >> 3;
    auto tswz = [](int row) { return row & 7; };
    int a_off[NP], a_mask[NP];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int m = m0 + (r0 >> 5) * WPX + j * 32 + (r0 & 31);
        const int mm = m < p.M ? m : 0;
        const int n = (int)fastdiv((uint32_t)mm, p.div_howo), rem = mm - n * HoWo;
        const int ho = (int)fastdiv((uint32_t)rem, p.div_wo), wo = rem - ho * p.Wo;
        const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
        // (relative to a descriptor base moved back by one row + one pixel -- conv_igemm_kernel.h: the tap's offset rides in the scalar offset)
        a_off[j] = (((n * p.H + hi0 + 1) * p.W + wi0 + 1) * p.in_cstride + p.in_coff + ((q ^ tswz(r0)) * EPC)) * ES;
        int mask = 1;                              // bit ky*ksize + kx: that tap reads inside the image (else zeros)
        if (p.ksize == 3) {
            const int cols = ((unsigned)wi0 < (unsigned)p.W ? 1 : 0) | ((unsigned)(wi0 + 1) < (unsigned)p.W ? 2 : 0) | ((unsigned)(wi0 + 2) < (unsigned)p.W ? 4 : 0);
            mask = ((unsigned)hi0 < (unsigned)p.H ? cols : 0) | ((unsigned)(hi0 + 1) < (unsigned)p.H ? cols << 3 : 0) |
                   ((unsigned)(hi0 + 2) < (unsigned)p.H ? cols << 6 : 0);
        }
        a_mask[j] = m < p.M ? mask : 0;
    }
    int b_vo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = r0 + j * 64;
        const int wb = row >> 6, pr = row & 63;
        const int jn = pr >> 4, i = pr & 15, g = i >> 2, r = i & 3;
        const int ch = chunk_channel(n0 + wb * WCH, jn >> 1, g) + (jn & 1) * 4 + r;
        b_vo[j] = (ch * p.K + ((q ^ tswz(row)) * EPC)) * ES;
    }
    const int tap_bias = (p.W + 1) * p.in_cstride * ES;
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(p.in - tap_bias, p.in_bytes + (unsigned)tap_bias);
    const __amdgpu_buffer_rsrc_t rs_wt = make_rsrc(p.wt, p.wt_bytes);
    const int wave_lds = __builtin_amdgcn_readfirstlane(wave * 1024);
    // (the touch has its own TOUCH_LDS bytes behind the two K-tiles: conv_common.h)
    if (p.touch != 0) weight_touch(rs_wt, smem + 2 * STAGE, n0 * p.K * ES, 256 * p.K * ES, wave, 8, lane);
    const int nk = p.K / BK;
    // staging cursor: the K-tile whose regions are being issued (tap, byte offset of c0, byte offset of k in the weights)
    // (canonical K order, common.h: 64-channel chunk -> tap -> channel; a K-tile of this kernel is one (chunk, tap))
    const int chb = k_chunk_channels(p.Cin, p.ksize) * ES;
    int tap = 0, ky = 0, kx = 0, c0b = 0, cbase = 0, c_in = 0, ktb = 0, lk = 0;
    int a_vo[NP];
    int tap_off = 0;                               // scalar: the tap's byte offset, part of the scalar offset c0b
    auto set_tap = [&]() {
        tap_off = ((ky * p.W + kx) * p.in_cstride) * ES;
#pragma unroll
        for (int j = 0; j < NP; ++j) a_vo[j] = ((a_mask[j] >> tap) & 1) ? a_off[j] : (int)0x80000000;
    };
    set_tap();
    // the cursor's K-tile -> LDS stage st: pixel part r, the four weight parts; `advance` moves the cursor to the next K-tile
#ifdef Y4_ABL_NOLOAD        // timing-only ablation builds (scripts/build_variant.sh; wrong results)
    auto issue_a = [&](int r, int st) {};
    auto issue_b = [&](int j, int st) {};
#else
    auto issue_a = [&](int r, int st) { buffer_load16_lds(rs_in, smem + st * STAGE + wave_lds + r * PART, a_vo[r], c0b); };
    auto issue_b = [&](int j, int st) { buffer_load16_lds(rs_wt, smem + st * STAGE + wave_lds + A_TILE + j * PART, b_vo[j], ktb); };
#endif
    auto advance = [&]() {
        ktb += BKB;
        c_in += BKB;
        ++lk;
        bool moved = false;
        if (c_in >= chb) {                         // next tap of this chunk; after the last tap the next chunk
            c_in = 0;
            ++tap;
            if (++kx >= p.ksize) {
                kx = 0;
                if (++ky >= p.ksize) { ky = 0; tap = 0; cbase += chb; }
            }
            moved = true;
        }
        if (lk >= nk) {                            // past the last K-tile: keep the load count, move no bytes
#pragma unroll
            for (int j = 0; j < NP; ++j) a_mask[j] = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) b_vo[j] = (int)0x80000000;
            tap = 0;
        }
        if (moved || lk >= nk) set_tap();
        c0b = cbase + c_in + tap_off;
    };
    const int rowA = (wm * 32) * BKB, rowB = A_TILE + (wn * WCH) * BKB;
    const bool full = (m0 + BM <= p.M) && (n0 + 256 <= p.cout_store);
    if constexpr (SCHED == 9) {
        using S9 = S9Sched<NP>;
        // issue slot of phase ph of K-tile kt (stage st = kt & 1)
        auto slot_a = [&](int ph, int st) {
            if (ph == 0) { issue_a(NP - 1, st ^ 1); advance(); }       // last pixel part of K-tile kt + 1
            else issue_a(ph - 1, st);                                  // pixel part ph - 1 of K-tile kt + 2
        };
        // ---- prologue: the slots of the virtual K-tiles -2 and -1 (only loads that target K-tiles >= 0)
#pragma unroll
        for (int ph = 1; ph < NP; ++ph) {
            slot_a(ph, 0);
            if (ph == 1)
#pragma unroll
                for (int j = 0; j < 4; ++j) issue_b(j, 0);
        }
#pragma unroll
        for (int ph = 0; ph < NP; ++ph) {
            slot_a(ph, 1);
            if (ph == 1)
#pragma unroll
                for (int j = 0; j < 4; ++j) issue_b(j, 1);
        }
        p8_wait_vm<S9::prologue_loads() - 5>();    // K-tile 0's pixel part 0 and weights have landed
        p8_barrier();
        {
            const int frow = lane & 15, fg = lane >> 4;
            int xo[2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) xo[kk] = frow * BKB + (((kk * 4 + fg) ^ (frow & 7)) * 16);
            f32x4 acc[MREP][NREP];
#pragma unroll
            for (int i = 0; i < MREP; ++i)
#pragma unroll
                for (int j = 0; j < NREP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            u32x4 wf[2][NREP], xf[2][2];
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
                for (int j = 0; j < NREP; ++j) wf[kk][j] = *(const u32x4*)(smem + rowB + j * 16 * BKB + xo[kk]);
#pragma unroll
                for (int h = 0; h < 2; ++h) xf[kk][h] = *(const u32x4*)(smem + rowA + h * 16 * BKB + xo[kk]);
            }
            s9_wait_phase<NP, NP - 1>();           // ... and pixel part 1, which phase 0 reads
            p8_barrier();
            for (int kt = 0; kt < nk; ++kt) {
                const int st = kt & 1;
                auto phase = [&](auto PH) {
                    constexpr int ph = decltype(PH)::value;
                    constexpr bool last = ph == NP - 1;
                    // fragments of the next phase: pixel part ph + 1 of this K-tile, or part 0 (and the weights) of the next one
                    const char* const na = smem + (last ? st ^ 1 : st) * STAGE + rowA + (last ? 0 : ph + 1) * PART;
                    const char* const nb = smem + (st ^ 1) * STAGE + rowB;
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int j = 0; j < NREP; ++j) {
                                const int i = kk * 8 + h * 4 + j;      // MFMA slot of this phase
#ifndef Y4_ABL_NOMFMA
                                Mma<DT>::run(acc[2 * ph + h][j], wf[kk][j], xf[kk][h]);
#else
                                asm volatile("" : "+v"(acc[2 * ph + h][j]) : "v"(wf[kk][j]), "v"(xf[kk][h]));
#endif
                                __builtin_amdgcn_sched_barrier(0);
                                // the side instruction(s) of the slot
                                if (i == 0) slot_a(ph, st);
                                if (ph == 1 && i >= 1 && i <= 4) issue_b(i - 1, st);
#ifndef Y4_ABL_NOREAD
                                if (j == NREP - 1) xf[kk][h] = *(const u32x4*)(na + h * 16 * BKB + xo[kk]);    // its last user has issued
                                if (last && h == 1) wf[kk][j] = *(const u32x4*)(nb + j * 16 * BKB + xo[kk]);
#endif
                                __builtin_amdgcn_sched_barrier(0);
                            }
                    s9_wait_phase<NP, ph>();
#ifndef Y4_ABL_NOBAR
                    p8_barrier();
#endif
                };
                phase(std::integral_constant<int, 0>{});
                phase(std::integral_constant<int, 1>{});
                phase(std::integral_constant<int, 2>{});
                if constexpr (NP == 4) phase(std::integral_constant<int, NP - 1>{});
            }
            conv_epilogue<DT, MREP, NREP>(p, acc, m0 + wm * WPX + frow, p.M, n0 + wn * WCH, fg, full);
        }
        return;
    }
    // ---- SCHED == 8 from here on.  Region r of the cursor's K-tile -> LDS stage st; the cursor advances behind the last region
    auto issue = [&](int r, int st) {
        issue_a(r, st);
        if (r == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) issue_b(j, st);
        }
        if (r == NP - 1) advance();
    };
    using Sched = P8Sched<NP>;
    auto slot_region = [](int ph) { return Sched::region(ph); };
    auto slot_late = [](int ph) { return Sched::late(ph); };

    // ---- prologue: the issue slots of the two virtual K-tiles before the first one
#pragma unroll
    for (int vk = -2; vk < 0; ++vk)
#pragma unroll
        for (int ph = 0; ph < NP; ++ph) {
            const int tk = vk + 2 - slot_late(ph);
            if (tk >= 0) issue(slot_region(ph), tk & 1);
        }
    p8_wait_vm<Sched::prologue_wait()>();          // K-tile 0's region 0 (pixel part 0 + weights) has landed
    p8_barrier();
    if (grp1) p8_barrier();                        // the stagger

    {
        const int frow = lane & 15, fg = lane >> 4;
        int xo[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) xo[kk] = frow * BKB + (((kk * 4 + fg) ^ (frow & 7)) * 16);
        f32x4 acc[MREP][NREP];
#pragma unroll
        for (int i = 0; i < MREP; ++i)
#pragma unroll
            for (int j = 0; j < NREP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        u32x4 wf[2][NREP];
        for (int kt = 0; kt < nk; ++kt) {
            const int st = kt & 1;
            const char* const sa = smem + st * STAGE + rowA;
            const char* const sb = smem + st * STAGE + rowB;
#pragma unroll
            for (int ph = 0; ph < NP; ++ph) {
                // ---- read-half
                u32x4 xf[2][2];
                if (ph == 0) {
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                        for (int j = 0; j < NREP; ++j) wf[kk][j] = *(const u32x4*)(sb + j * 16 * BKB + xo[kk]);
                }
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int h = 0; h < 2; ++h) xf[kk][h] = *(const u32x4*)(sa + ph * PART + h * 16 * BKB + xo[kk]);
                __builtin_amdgcn_sched_barrier(0);
                issue(slot_region(ph), slot_late(ph) ? st ^ 1 : st);
                if (grp1) p8_wait_phase<NP>(ph);
                p8_barrier();
                // ---- MFMA-half
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int j = 0; j < NREP; ++j) Mma<DT>::run(acc[2 * ph + h][j], wf[kk][j], xf[kk][h]);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                if (!grp1) p8_wait_phase<NP>(ph);
                p8_barrier();
            }
        }
        if (!grp1) p8_barrier();
        conv_epilogue<DT, MREP, NREP>(p, acc, m0 + wm * WPX + frow, p.M, n0 + wn * WCH, fg, full);
    }
}

template <int DT, int BM, int SCHED>
static int launch_p8_cfg(const ConvK& k, hipStream_t stream) {
    constexpr int lds = 2 * (BM + 256) * 128 + TOUCH_LDS;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = conv_p8_kernel<DT, BM, SCHED>;
    static PerDeviceOnce once;
    if (const uint64_t bit = once.due()) {
        Y4_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        once.mark(bit);
    }
    hipLaunchKernelGGL(kern, dim3(k.grid_m * k.grid_n), dim3(512), lds, stream, k);
    Y4_CHECK_HIP(hipGetLastError());
    return Y4_OK;
}

// nst: the schedule code of conv_tiles.h (8: staggered wave groups, 9: software-pipelined)
template <int DT>
static int launch_p8(int bm, int nst, const ConvK& k, hipStream_t s) {
    if (bm == 192 && nst == 8) return launch_p8_cfg<DT, 192, 8>(k, s);
    if (bm == 256 && nst == 8) return launch_p8_cfg<DT, 256, 8>(k, s);
    if (bm == 192 && nst == 9) return launch_p8_cfg<DT, 192, 9>(k, s);
    set_error("conv2d: no phased kernel with a %d-pixel tile and schedule %d", bm, nst);
    return Y4_EINVAL;
}

}  // namespace y4
