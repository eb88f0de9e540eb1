// conv_l12_kernel.h -- the reference's conv() unit (custom_layers.py:5-31) as a PRODUCER / CONSUMER implicit-GEMM kernel for
// gfx950: 12 waves per workgroup, one workgroup per CU.  Same GEMM view, operand roles, LDS image, K order and epilogue as
// conv_igemm_kernel.h (bit-identical results); what changes is WHO issues the LDS-DMA loads.
//
// Why (round 3, timing-only ablations of the software-pipelined kernel of conv_p8_kernel.h on a 3x3 256->512 @38^2, 192x256
// tile): complete 112 us; without the LDS-DMA loads 85; without fragment reads 104; without barriers 100; MFMAs alone 77.5;
// everything but the MFMAs 66.  Three different orders of the same instructions in the SAME eight waves (plain loop,
// staggered wave groups, loads and reads slotted between the MFMAs) all take the same time: a `buffer_load ... lds` costs its
// issuing wave 100-185 cycles (the CU's one address unit takes a 1 KB piece per ~16 cycles and the wave waits its turn),
// seven of them per K-tile, and a wave that is stuck in the memory pipe issues no MFMAs.  So here
//   * waves 0-7 (2 x 4 over the 192 x 256 tile, two per SIMD) ONLY multiply and read fragments: per phase 16 MFMAs with the
//     next phase's fragment reads slotted in between (each into the register its last user has just released);
//   * waves 8-11 (one per SIMD) ONLY stage: all 56 1-KB pieces of a K-tile, 14 each, with the counted `vmcnt` waits;
//   * one `s_barrier` per phase (all 12 waves) carries both orderings: a region is read one phase after the producers'
//     wait that retires it, and refilled two phases after the phase that read it (as conv_p8_kernel.h, SCHED 9).
// 3 waves per SIMD leave 168 registers per wave: 96 accumulators + 48 fragment registers + addresses fit.
#pragma once
#include <type_traits>

#include "conv_p8_kernel.h"

namespace y4 {

// Issue order of ONE producer wave, NP = 3 phases per K-tile.  Pixel part r = 64 rows = 8 pieces, 2 per producer; weights =
// 256 rows = 32 pieces, 8 per producer, split over two phases.  Phase ph of K-tile k issues
//   ph 0: pixel part 2 of K-tile k+1 (2)       ph 1: pixel part 0 of k+2 (2), weight pieces 0-3 of k+2 (4)
//   ph 2: pixel part 1 of k+2 (2), weight pieces 4-7 of k+2 (4)
// (regions: pixel part r read during phase (T, r) - 1, weights read during the last phase of K-tile T - 1; free two phases later).
struct L12Sched {
    static constexpr int NP = 3;
    static constexpr int loads(int ph) { return ph == 0 ? 2 : 6; }
    static constexpr int before(int g) {                        // loads issued before global phase g
        int n = 0;
        for (int i = 0; i < g; ++i) n += loads(i % NP);
        return n;
    }
    // index (issue order) of the LAST load of a region of K-tile T: region 0..2 pixel parts, 3 = weights (second half)
    static constexpr int last_load(int T, int region) {
        if (region == 2) return before((T - 1) * NP) + 1;
        if (region == 0) return before((T - 2) * NP + 1) + 1;
        if (region == 1) return before((T - 2) * NP + 2) + 1;
        return before((T - 2) * NP + 2) + 5;                    // weights: pieces 4-7 come behind pixel part 1 in phase 2
    }
    static constexpr int wait(int ph) {                         // in flight allowed at the end of phase ph
        const int g = 4 * NP + ph, gr = g + 1, gu = g + 2;
        int need = last_load(gu / NP, gu % NP);
        if (gr % NP == NP - 1) {
            const int w = last_load(gr / NP + 1, 3);
            need = w > need ? w : need;
        }
        return before(g + 1) - 1 - need;
    }
    static constexpr int prologue_loads() { return before(2 * NP) - loads(0); }
    // the prologue skips global phase 0 of a run whose K-tile 0 is T = 2 here: loads behind K-tile 0's last weight piece
    static constexpr int prologue_first() { return prologue_loads() - (last_load(2, 3) - loads(0) + 1); }
};
static_assert(L12Sched::wait(0) == 14 && L12Sched::wait(1) == 8 && L12Sched::wait(2) == 18 && L12Sched::prologue_loads() == 26 &&
                  L12Sched::prologue_first() == 14, "producer wait counts");

template <int DT>
__global__ __launch_bounds__(768, 1) void conv_l12_kernel(const ConvK p) {
    static_assert(DT != Y4_F32, "16-bit dtypes");
    constexpr int BM = 192, NP = 3, ES = 2, BKB = 128, BK = 64, EPC = 8;
    constexpr int WPX = BM / 2, WCH = 64, MREP = WPX / 16, NREP = 4;
    constexpr int A_TILE = BM * BKB, B_TILE = 256 * BKB, PART = 64 * BKB;
    constexpr int A_BASE = 0, B_BASE = 2 * A_TILE;                 // [A stage 0][A stage 1][B stage 0][B stage 1]
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using S = L12Sched;

    const int nwg = p.grid_m * p.grid_n;
    int t;
    {
        const int b = blockIdx.x, qq = nwg >> 3, rr = nwg & 7, xcd = b & 7, idx = b >> 3;
        t = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
    }
    const int tile_m = (int)fastdiv((uint32_t)t, p.div_gridn), tile_n = t - tile_m * p.grid_n;
    const int m0 = tile_m * BM, n0 = tile_n * 256;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = p.K / BK;

    if (wave >= 8) {
        // =============================================================== producer wave: LDS-DMA only
        const int l = wave - 8;                                    // pieces l and l + 4 of every 8-piece group
        const int q = lane & 7, pr = lane >> 3;                    // chunk slot, row within the piece
        int a_off[NP][2], a_mask[NP][2];
        const int HoWo = p.Ho * p.Wo;
#pragma unroll
        for (int j = 0; j < NP; ++j)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int r0 = (l + 4 * u) * 8 + pr;               // LDS row within the part; pixel row as conv_p8_kernel.h
                const int m = m0 + (r0 >> 5) * WPX + j * 32 + (r0 & 31);
                const int mm = m < p.M ? m : 0;
                const int n = (int)fastdiv((uint32_t)mm, p.div_howo), rem = mm - n * HoWo;
                const int ho = (int)fastdiv((uint32_t)rem, p.div_wo), wo = rem - ho * p.Wo;
                const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
                a_off[j][u] = (((n * p.H + hi0) * p.W + wi0) * p.in_cstride + p.in_coff + ((q ^ (r0 & 7)) * EPC)) * ES;
                int mask = 0;
                for (int ky = 0; ky < p.ksize; ++ky)
                    for (int kx = 0; kx < p.ksize; ++kx)
                        if ((unsigned)(hi0 + ky) < (unsigned)p.H && (unsigned)(wi0 + kx) < (unsigned)p.W) mask |= 1 << (ky * p.ksize + kx);
                a_mask[j][u] = m < p.M ? mask : 0;
            }
        int b_vo[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = (l + 4 * i) * 8 + pr;                  // weight row 0..255 of the tile
            const int wb = row >> 6, prr = row & 63;
            const int jn = prr >> 4, ii = prr & 15, g = ii >> 2, r = ii & 3;
            const int ch = chunk_channel(n0 + wb * WCH, jn >> 1, g) + (jn & 1) * 4 + r;
            b_vo[i] = (ch * p.K + ((q ^ (row & 7)) * EPC)) * ES;
        }
        const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(p.in, p.in_bytes);
        const __amdgpu_buffer_rsrc_t rs_wt = make_rsrc(p.wt, p.wt_bytes);
        // pixel cursor: the K-tile whose parts are being issued (canonical K order, common.h: chunk -> tap -> channel)
        const int chb = k_chunk_channels(p.Cin, p.ksize) * ES;
        int tap = 0, ky = 0, kx = 0, c0b = 0, cbase = 0, c_in = 0, lk = 0;
        int a_vo[NP][2];
        auto set_tap = [&]() {
            const int tap_off = ((ky * p.W + kx) * p.in_cstride) * ES;
#pragma unroll
            for (int j = 0; j < NP; ++j)
#pragma unroll
                for (int u = 0; u < 2; ++u) a_vo[j][u] = ((a_mask[j][u] >> tap) & 1) ? a_off[j][u] + tap_off : (int)0x80000000;
        };
        set_tap();
        auto issue_a = [&](int r, int st) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
                buffer_load16_lds(rs_in, smem + A_BASE + st * A_TILE + r * PART + (l + 4 * u) * 1024, a_vo[r][u], c0b);
        };
        auto advance = [&]() {
            c_in += BKB;
            ++lk;
            bool moved = false;
            if (c_in >= chb) {
                c_in = 0;
                ++tap;
                if (++kx >= p.ksize) {
                    kx = 0;
                    if (++ky >= p.ksize) { ky = 0; tap = 0; cbase += chb; }
                }
                moved = true;
            }
            c0b = cbase + c_in;
            if (lk >= nk) {
#pragma unroll
                for (int j = 0; j < NP; ++j) a_mask[j][0] = a_mask[j][1] = 0;
                tap = 0;
            }
            if (moved || lk >= nk) set_tap();
        };
        // weight pieces i0..i0+3 of K-tile T -> weight stage T & 1 (past the last K-tile: out of range = no traffic)
        auto issue_b = [&](int i0, int T) {
            const bool live = T < nk;
#pragma unroll
            for (int i = i0; i < i0 + 4; ++i)
                buffer_load16_lds(rs_wt, smem + B_BASE + (T & 1) * B_TILE + (l + 4 * i) * 1024, live ? b_vo[i] : (int)0x80000000, T * BKB);
        };
        // the issue slot of phase ph of K-tile kt
        auto slot = [&](int ph, int kt) {
            if (ph == 0) { issue_a(NP - 1, (kt + 1) & 1); advance(); }
            else if (ph == 1) { issue_a(0, kt & 1); issue_b(0, kt + 2); }
            else { issue_a(1, kt & 1); issue_b(4, kt + 2); }
        };
        slot(1, -2); slot(2, -2);                                  // virtual K-tiles -2 and -1: K-tile 0 complete, K-tile 1 begun
        slot(0, -1); slot(1, -1); slot(2, -1);
        p8_wait_vm<S::prologue_first()>();                         // K-tile 0's pixel part 0 and weights have landed
        p8_barrier();
        p8_wait_vm<S::wait(NP - 1)>();                             // ... and its pixel part 1
        p8_barrier();
        for (int kt = 0; kt < nk; ++kt) {
            slot(0, kt); p8_wait_vm<S::wait(0)>(); p8_barrier();
            slot(1, kt); p8_wait_vm<S::wait(1)>(); p8_barrier();
            slot(2, kt); p8_wait_vm<S::wait(2)>(); p8_barrier();
        }
        return;
    }

    // =================================================================== consumer wave: fragment reads + MFMAs
    const int wm = wave >> 2, wn = wave & 3;
    const int frow = lane & 15, fg = lane >> 4;
    const int rowA = A_BASE + (wm * 32) * BKB, rowB = B_BASE + (wn * WCH) * BKB;
    int xo[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) xo[kk] = frow * BKB + (((kk * 4 + fg) ^ (frow & 7)) * 16);
    f32x4 acc[MREP][NREP];
#pragma unroll
    for (int i = 0; i < MREP; ++i)
#pragma unroll
        for (int j = 0; j < NREP; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 wf[2][NREP], xf[2][2];
    p8_barrier();
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int j = 0; j < NREP; ++j) wf[kk][j] = *(const u32x4*)(smem + rowB + j * 16 * BKB + xo[kk]);
#pragma unroll
        for (int h = 0; h < 2; ++h) xf[kk][h] = *(const u32x4*)(smem + rowA + h * 16 * BKB + xo[kk]);
    }
    p8_barrier();
    for (int kt = 0; kt < nk; ++kt) {
        const int st = kt & 1;
        auto phase = [&](auto PH) {
            constexpr int ph = decltype(PH)::value;
            constexpr bool last = ph == NP - 1;
            const char* const na = smem + rowA + (last ? st ^ 1 : st) * A_TILE + (last ? 0 : ph + 1) * PART;
            const char* const nb = smem + rowB + (st ^ 1) * B_TILE;
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int j = 0; j < NREP; ++j) {
                        Mma<DT>::run(acc[2 * ph + h][j], wf[kk][j], xf[kk][h]);
                        __builtin_amdgcn_sched_barrier(0);
                        if (j == NREP - 1) xf[kk][h] = *(const u32x4*)(na + h * 16 * BKB + xo[kk]);      // its last user has issued
                        if (last && h == 1) wf[kk][j] = *(const u32x4*)(nb + j * 16 * BKB + xo[kk]);
                        __builtin_amdgcn_sched_barrier(0);
                    }
            __builtin_amdgcn_s_setprio(0);
            p8_barrier();
        };
        phase(std::integral_constant<int, 0>{});
        phase(std::integral_constant<int, 1>{});
        phase(std::integral_constant<int, 2>{});
    }
    const bool full = (m0 + BM <= p.M) && (n0 + 256 <= p.cout_store);
    conv_epilogue<DT, MREP, NREP>(p, acc, m0 + wm * WPX + frow, p.M, n0 + wn * WCH, fg, full);
}

template <int DT>
static int launch_l12(const ConvK& k, hipStream_t stream) {
    constexpr int lds = 2 * (192 + 256) * 128;
    auto kern = conv_l12_kernel<DT>;
    static PerDeviceOnce once;
    if (const uint64_t bit = once.due()) {
        Y4_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        once.mark(bit);
    }
    hipLaunchKernelGGL(kern, dim3(k.grid_m * k.grid_n), dim3(768), lds, stream, k);
    Y4_CHECK_HIP(hipGetLastError());
    return Y4_OK;
}

}  // namespace y4
