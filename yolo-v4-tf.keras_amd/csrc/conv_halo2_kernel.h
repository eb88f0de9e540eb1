// conv_halo2_kernel.h -- 3x3 stride-1 'same' conv (the reference's conv() unit, custom_layers.py:5-31, with its Add / concat-slice
// epilogue) as a ONE-WAVE-PER-SIMD kernel (round 6).
//
// Every earlier conv kernel of this library is the same machine: >= 512 threads = two waves per SIMD under a 256-register cap,
// v_mfma_f32_16x16x32, both operands through LDS, one block-wide barrier per K-tile -- and none keeps the matrix pipe busier than
// 51 % (profiles/r05/mfma_util.json; issue-stalled 52 % of its wave cycles).  This one is built the other way round:
//
//   * 256 threads = four waves, one per SIMD, up to 512 registers each: a wave tile of 192 x 64 (or 96 x 128 ...) outputs as
//     v_mfma_f32_32x32x16 blocks -- the one MFMA shape a single wave issues at the pipe's full rate (32 cycles per instruction);
//   * the INPUT is the halo tile of conv_halo_kernel.h (a band of R full image rows plus one above and below, one 64-channel chunk at
//     a time, two buffers), staged by LDS-DMA -- but at pitch W + 2 exactly and with a swizzle that follows the PIXEL index, so the
//     nine shifted fragment reads of a 32-pixel block are free of LDS bank conflicts (scripts/bank_sim.py; conv_halo_kernel's row & 7
//     swizzle costs its kx = 1, 2 reads 1.5 extra cycles per lane group);
//   * the WEIGHTS never touch LDS: they are kept once more in MFMA-fragment order (pack_conv_frag32: one k-step's A operand of a
//     32-channel block = 1 KB contiguous), and each wave loads the fragments of its own channel blocks straight into registers, eight
//     k-steps (two taps) ahead, through a nine-slot register ring.  No weight stage, no per-tap barrier: the only workgroup barrier
//     is the halo-buffer swap, once per chunk (36 k-steps, ~14 k cycles);
//   * the pixel fragments of k-step s + 1 are read from LDS BETWEEN the MFMAs of k-step s (two register sets), as are the weight
//     loads and the next chunk's halo pieces: at most ~3 single-issue instructions per 32-cycle MFMA, against the ~5 such a gap hides
//     (MI355X_MICROARCH.md, "one wave per SIMD").
//
// All loads with a register destination are inline asm with hand-counted s_waitcnt (the compiler would drain the whole queue at
// every use beside the LDS-DMA, cdna_hip_programming.md section 5 trap (b)); the order of the K loop's statements is pinned with
// sched_barrier.  Every wave issues exactly the same VMEM instructions per k-step (surplus halo pieces are all-lanes-out-of-range
// dummies into a dead KB of LDS), so the counts are compile-time constants.  Two rules keep such a kernel honest, both learnt the hard
// way this round and both checked on the ISA by scripts/h2_audit.py (tests/test_h2_isa_audit.py): no instruction
// may touch a register whose load has not been waited for -- which includes the registers of loads still in flight when the K loop
// ENDS: they are dead to the compiler, so every ring register is "used" once more behind the drain --, and no LDS-DMA piece may be
// outstanding at the barrier behind which its buffer is read.
//
// Template parameters beside the tile: KC = channels per staged sub-chunk (64: LDS rows of 128 B; 32: rows of 64 B at pitch W + 4, the
// chunk's two halves one after the other -- half the LDS and half the prologue), OCC = workgroups per CU the register budget is cut for.
//
// K order: 64-channel chunk -> tap -> k-step of 16 channels, each v_mfma_32x32x16 summing its 16 products internally: the order of
// conv_igemm_kernel's 32x32x16 tiles (ids 33-37) -- bit-identical to those, NOT to the 16x16x32 kernels.  The shipped schedule is
// part of the numerics, as with split-K; tests hold these tiles to the oracle and to run-to-run determinism.
#pragma once
#include "conv_common.h"
#include "conv_tiles.h"

namespace y4 {

typedef int i32x4 __attribute__((ext_vector_type(4)));

// compile-time loop: f(std::integral_constant<int, I>) for I = 0 .. N-1.  The K loop's k-step index must be a constant expression
// in every statement (wait counts, immediates, register-array indices): `#pragma unroll` over 36 x 12 statements is only a request, and
// when the optimiser declines, every one of them turns into a chain of run-time branches.
template <int I, int N, class F> __device__ __forceinline__ void h2_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        h2_static_for<I + 1, N>(f);
    }
}

template <int N> __device__ __forceinline__ void h2_wait_vm_n() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void h2_ds_read(u32x4& dst, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr)); }
template <int OFF> __device__ __forceinline__ void h2_wload_n(u32x4& dst, int voff, i32x4 rs) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "=v"(dst) : "v"(voff), "s"(rs), "n"(OFF));
}
// "this register now holds loaded data": orders every consumer behind the wait that precedes it (cdna_hip_programming.md 5.7 form ii)
__device__ __forceinline__ void h2_landed(u32x4& v) { asm volatile("" : "+v"(v)); }

#ifdef H2_TRACE
// experiments only (scripts/h2_trace.py; a variant library built with -DH2_TRACE): the four waves of workgroups 0..15 stamp s_memtime at
// fixed points -- [wg][wave][0] start, [1] prologue issued, [2] first barrier passed, [3 + c] end of chunk c, [12] loop done, [13] end
__device__ unsigned long long h2_trace_buf[16][4][16];
__device__ unsigned long long h2_trace_blocks[4096][4];        // per workgroup (wave 0): realtime at start / end (100 MHz), HW_ID, cycles
#define H2_STAMP(slot)                                                                                   \
    do {                                                                                                 \
        if (blockIdx.x < 16) {                                                                           \
            unsigned long long t_;                                                                       \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                   \
            if (lane == 0) h2_trace_buf[blockIdx.x][wave][slot] = t_;                                    \
        }                                                                                                \
    } while (0)
#define H2_STAMP_RT(slot)                                                                                \
    do {                                                                                                 \
        if (blockIdx.x < 16) {                                                                           \
            unsigned long long t_;                                                                       \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
            if (lane == 0) h2_trace_buf[blockIdx.x][wave][slot] = t_;                                    \
        }                                                                                                \
    } while (0)
#else
#define H2_STAMP(slot) do { } while (0)
#define H2_STAMP_RT(slot) do { } while (0)
#endif

#ifndef H2_NOP
#define H2_NOP -1               // >= 0: an `s_nop H2_NOP` behind every MFMA of the K loop (experiments only; see there)
#endif
#ifndef H2_LOOP_PRIO
#define H2_LOOP_PRIO 2
#endif
#ifndef HALO2_ABLATIONS
// 1 (kernel experiments only): the HALO_ABL environment variable then switches parts of the K loop off at run time -- bit 1 the
// weight loads, 2 the halo pieces, 4 the fragment reads, 8 the barrier -- for TIMING; results are wrong.  Not in the regular build.
#define HALO2_ABLATIONS 0
#endif

// KC: channels per staged sub-chunk (64: LDS rows of 128 B, one sub-chunk per 64-channel chunk of the K order; 32: rows of 64 B, the chunk's
// two halves one after the other -- half the LDS, which is what lets two workgroups share a CU).  OCC: workgroups per CU the register
// budget is cut for (1: 512 registers per lane, 2: 256).
template <int DT, int BM, int BN, int WM, int WN, int KC, int OCC>
__global__ __launch_bounds__(256, OCC) void conv_halo2_kernel(const ConvK p) {
    static_assert(DT != Y4_F32, "halo2 tiles: 16-bit dtypes");
    static_assert(WM * WN == 4, "four waves, one per SIMD");
    static_assert(KC == 64 || KC == 32, "sub-chunk");
    constexpr int NW = 4, ES = 2;
    constexpr int RB = KC * ES, CPR = RB / 16, RPP = 1024 / RB, SH = KC == 64 ? 1 : 2;      // row bytes, 16-B columns per row, rows per 1-KB piece
    constexpr int KS = KC / 16, SUB = 9 * KS, NSUB = 36 / SUB;                               // k-steps per tap, per sub-chunk; sub-chunks per chunk
    constexpr int PMAX = halo2_pmax(KC, OCC);
    constexpr int WPX = BM / WM, WCH = BN / WN, MB = WPX / 32, NB = WCH / 32, NMMA = MB * NB;
    static_assert(WPX % 32 == 0 && WCH % 32 == 0, "wave tile = 32 x 32 blocks");
    constexpr int RING = 9, AHEAD = 8;                  // weight-fragment ring: k-steps held / k-steps of look-ahead (36 % RING == 0)
    static_assert(NMMA >= 4 && PMAX <= SUB - AHEAD, "a k-step needs MFMAs to hide its loads behind; the pieces are older than the chunk's last waits");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    asm volatile("" ::"s"(p.H), "s"(p.W), "s"(p.Cin), "s"(p.K), "s"(p.in_cstride), "s"(p.in_coff), "s"(p.grid_m), "s"(p.grid_n), "s"(p.in_bytes),
                 "s"(p.wfrag_bytes), "s"(p.h_rows), "s"(p.h_bands), "s"(p.h_pitch), "s"(p.touch));

    // ---- XCD-aware tile mapping (as conv_halo_kernel)
    const int nwg = p.grid_m * p.grid_n;
    int t;
    {
        const int b = blockIdx.x, qq = nwg >> 3, rr = nwg & 7, xcd = b & 7, idx = b >> 3;
        t = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + idx;
    }
    int tile_m = (int)fastdiv((uint32_t)t, p.div_gridn), tile_n = t - tile_m * p.grid_n;
    if (p.h_xmap != 0 && (nwg & 7) == 0) {
        const int b = blockIdx.x, xcd = b & 7, idx = b >> 3;
        if (p.grid_n >= 8 && (p.grid_n & 7) == 0) { const int g8 = p.grid_n >> 3; tile_n = xcd + 8 * (idx % g8); tile_m = idx / g8; }
        else if (p.grid_n == 4 || p.grid_n == 2 || p.grid_n == 1) { const int per = 8 / p.grid_n; tile_n = xcd % p.grid_n; tile_m = idx * per + xcd / p.grid_n; }
    }
    const int n0 = tile_n * BN;
    const int img = (int)fastdiv((uint32_t)tile_m, p.h_div_bands), band = tile_m - img * p.h_bands;
    const int R = p.h_rows, P = p.h_pitch, W = p.W, H = p.H;
    const int y0 = band * R;
    const int rows_here = H - y0 < R ? H - y0 : R;
    const int npx = rows_here * W;
    const int m_base = (img * H + y0) * W;
    const int AROWS = halo2_rows_alloc(R, P), ABYTES = AROWS * RB, APIECES = ABYTES >> 10;

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave - wm * WN;
    const int l32 = lane & 31, fh = lane >> 5;
    char* const lds_dummy = smem + 2 * ABYTES;            // 1 KB nobody reads: where the surplus pieces land
    char* const lds_touch = lds_dummy + 1024;
    H2_STAMP(0);
    H2_STAMP_RT(14);
#ifdef H2_TRACE
    unsigned long long h2_t0_rt, h2_t0_c;
    asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(h2_t0_rt), "=s"(h2_t0_c)::"memory");
#endif

    // ---- halo staging: piece u = LDS rows RPP u .. of a buffer; this wave stages pieces wave, wave + 4, ...  Lane (row r = RPP u +
    //      lane / CPR, physical 16-byte column lane % CPR) copies LOGICAL column (lane % CPR) ^ f(r) of halo position (yy, xx) = (r / P,
    //      r % P), i.e. of image pixel (y0 - 1 + yy, xx - 1); positions outside the image (the conv's zero padding and the pitch's spare
    //      columns), rows past the halo tile and pieces past the buffer get an out-of-range offset: zeros, no traffic.
    //      f(r) = ((r >> SH) - yy) & (CPR - 1): the pitch is W + 256 / RB, so the row of band pixel q = y W + x under tap (ky, kx) is
    //      q + (256 / RB) y + ky P + kx and (row mod 256 / RB, f) is a function of q + const alone: the 16 lanes of every ds_read_b128
    //      lane group land in 16 distinct 16-byte slots of the 256-byte bank window (scripts/bank_sim.py).
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(p.in, p.in_bytes);
    int poff[PMAX];
#pragma unroll
    for (int k = 0; k < PMAX; ++k) {
        const int u = wave + k * NW;
        const int r = u * RPP + lane / CPR;
        const int yy = (int)fastdiv((uint32_t)r, p.h_div_pitch), xx = r - yy * P;
        const int iy = y0 - 1 + yy, ix = xx - 1;
        const int f = ((r >> SH) - yy) & (CPR - 1);
        const bool ok = u < APIECES && yy < R + 2 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
        poff[k] = ok ? (((img * H + iy) * W + ix) * p.in_cstride + p.in_coff + (((lane % CPR) ^ f) << 3)) * ES : (int)0x80000000;
    }
    auto stage_piece = [&](int k, int buf, int cbyte, bool real) __attribute__((always_inline)) {
        const int u = wave + k * NW;
        char* const dst = u < APIECES ? smem + buf * ABYTES + u * 1024 : lds_dummy;
        buffer_load16_lds(rs_in, dst, real ? poff[k] : (int)0x80000000, cbyte);
    };

    // ---- weights: fragment order [32-channel block][chunk][tap][k-step of 16][lane][16 B] (pack_conv_frag32); a 32-channel sub-chunk
    //      walks k-steps 2h, 2h + 1 of every tap of its chunk
    i32x4 rs_w;
    {
        const uint64_t a = (uint64_t)p.wfrag;
        rs_w = i32x4{(int)(uint32_t)a, (int)(uint32_t)((a >> 32) & 0xffffu), (int)p.wfrag_bytes, 0x00020000};
    }
    const int nchunks = p.Cin >> 6;
    const int blk_stride = nchunks * 9 * 4096;            // bytes of one 32-channel block
    int wv[NB];                                           // this lane's byte offset of chunk c's fragments of channel block j
#pragma unroll
    for (int j = 0; j < NB; ++j) wv[j] = (((n0 + wn * WCH) >> 5) + j) * blk_stride + lane * 16;
    // byte offset of k-step `slot` (0 .. 35 within a chunk, may run into the next chunk) from wv: a multiple of 4096 for the address + an immediate
    auto w_hi = [](int slot) constexpr { const int c = slot / 36, l = slot % 36; return c * 36864 + (KC == 64 ? l >> 2 : (l % 18) >> 1) * 4096; };
    auto w_lo = [](int slot) constexpr { const int l = slot % 36; return (KC == 64 ? l & 3 : 2 * (l / 18) + (l & 1)) * 1024; };

    if (p.touch != 0) weight_touch(make_rsrc(p.wfrag, p.wfrag_bytes), lds_touch, n0 * p.K * ES, BN * p.K * ES, wave, NW, lane);

    // ---- pixel fragments: block i of this wave = band pixels wm*WPX + 32 i + l32 (pixels past the band read pixel 0's rows; never stored)
    int r0[MB], yq[MB];                                   // halo row of tap (0, 0) in buffer 0; the pixel's row in the band
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        int pix = wm * WPX + i * 32 + l32;
        pix = pix < npx ? pix : 0;
        const int y = (int)fastdiv((uint32_t)pix, p.div_wo), x = pix - y * W;
        r0[i] = y * P + x;
        yq[i] = y;
    }
    unsigned abase[MB];                                   // byte address of k-step 0's fragment under the current tap
    const unsigned smem_u = (unsigned)(uintptr_t)smem;
    auto frag_base = [&](int i, int ky, int kx, int brow) __attribute__((always_inline)) {
        // (OCC = 2: r0 laundered -- the nine taps' addresses are invariant over the chunk loop, and hoisted they are 9 MB registers a
        //  256-register K loop does not have: the optimiser then spills them and drains the whole load queue (vmcnt(0)) at every reload.
        //  With 512 registers they ARE hoisted, which is worth 10 % of the loop: 33.5 against 36.8 cycles per MFMA)
        int rr = r0[i];
#if defined(H2_DBG) && H2_DBG == 3
        asm volatile("" : "+v"(rr));
#else
        if constexpr (OCC == 2) asm volatile("" : "+v"(rr));
#endif
        const int r = rr + ky * P + kx;                   // (brow % 16 == 0: the swizzle does not see the buffer)
        const int e = ((r >> SH) - yq[i] - ky) ^ fh;
        return smem_u + (unsigned)((r + brow) * RB + ((e & (CPR - 1)) << 4));
    };

    f32x16 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // pixel fragments: two sets by k-step parity -- or ONE when the wave has a single channel block (NB == 1): fragment i is then dead as
    // soon as MFMA i of the k-step has issued, and the next k-step's read of it (issued behind that MFMA, landing >= 64 cycles later,
    // long after the MFMA has fetched its operands) goes into the same registers
    constexpr int XS = NB == 1 ? 1 : 2;
    u32x4 xf[XS][MB];
    u32x4 wf[RING][NB];                                   // weight fragments: k-step % RING

    // ---- prologue: sub-chunk 0's halo tile, the weights of k-steps 0 .. AHEAD-1
#pragma unroll
    for (int k = 0; k < PMAX; ++k) stage_piece(k, 0, 0, true);
    h2_static_for<0, AHEAD>([&](auto sc) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value;
#pragma unroll
        for (int j = 0; j < NB; ++j) h2_wload_n<w_lo(s)>(wf[s % RING][j], wv[j] + w_hi(s), rs_w);
    });
    H2_STAMP(1);
    h2_wait_vm_n<AHEAD * NB>();                           // the pieces are older than the weight loads
    asm volatile("s_barrier" ::: "memory");
    H2_STAMP(2);
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        abase[i] = frag_base(i, 0, 0, 0);
        h2_ds_read(xf[0][i], abase[i]);
    }

    const int nsteps = nchunks * 36;
    // Two workgroups per CU = two waves per SIMD: the one in its K loop must win the issue arbitration against the one in its
    // prologue / epilogue (plain age order lets an older wave's Mish epilogue starve the younger wave's MFMA stream of VALU slots)
    if constexpr (OCC == 2) __builtin_amdgcn_s_setprio(H2_LOOP_PRIO);
    for (int c = 0; c < nchunks; ++c) {
        h2_static_for<0, 36>([&](auto sgc) __attribute__((always_inline)) {
            constexpr int sg = decltype(sgc)::value;
            constexpr int sb = sg / SUB, ls = sg % SUB, cur = (sg & 1) % XS, nxt = (cur + 1) % XS;            // sub-chunk of the chunk, k-step in it
            // -- this k-step's weights (loaded AHEAD k-steps ago): VMEM instructions issued since = the NB loads of each of the
            //    AHEAD-1 k-steps between, plus the one halo piece of every k-step among them that carries one.  In a k-step the piece goes
            //    out IN FRONT of the weight loads, so the piece of k-step sg - AHEAD is older than the loads waited for here: at the last
            //    k-step of a sub-chunk (SUB - 1) every piece of k-steps 0 .. SUB - 1 - AHEAD -- all of them, PMAX <= SUB - AHEAD -- has landed
            constexpr int younger = [] {
                auto piece = [](int q) { return ((q + 36) % SUB) < PMAX ? 1 : 0; };
                int y = 0;
                for (int q = sg - (AHEAD - 1); q < sg; ++q) y += NB + piece(q);
                return y;
            }();
#if defined(H2_DBG) && H2_DBG == 4
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
            h2_wait_vm_n<younger>();
#pragma unroll
            for (int j = 0; j < NB; ++j) h2_landed(wf[sg % RING][j]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < MB; ++i) h2_landed(xf[cur][i]);
            if constexpr (ls == SUB - 1) {
                // sub-chunk boundary: every wave's pieces of the next one have landed (the wait above left only weight loads in flight) and
                // every wave has read the current buffer to the end -- the next k-step's fragments come from the other buffer, and the
                // pieces issued from now on overwrite this one
                if (!(HALO2_ABLATIONS && (p.h_abl & 8))) asm volatile("s_barrier" ::: "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
            // -- MFMAs of this k-step, with the next k-step's fragment reads, the weight loads of k-step sg + AHEAD and one halo
            //    piece of the next sub-chunk between them
            constexpr int nls = (ls + 1) % SUB, ntap = nls / KS, nks = nls % KS;
            // halo buffer the NEXT k-step reads / the next sub-chunk is staged into: sub-chunk index parity (compile-time for KC = 32)
            const int cc = c * NSUB + sb;
            const int rd_buf = ((ls == SUB - 1 ? cc + 1 : cc) & 1), st_buf = (cc + 1) & 1;
            h2_static_for<0, NMMA>([&](auto mc) __attribute__((always_inline)) {
                constexpr int m = decltype(mc)::value;
                constexpr int j = m / MB, i = m - j * MB;     // channel block outer
                Mma32<DT>::run(acc[i][j], wf[sg % RING][j], xf[cur][i]);
                // (experiment switch, off: an s_nop behind every MFMA.  While the bug fixed behind the K loop -- re-used destinations of
                //  loads still in flight -- was being hunted, such a nop made the failures vanish, which looked like a write-after-read
                //  hazard on the MFMA's source registers; it only moved the register allocation.  With the real fix in, the kernel is
                //  clean with and without it: 0 differing launches of 3 600 / 0 of 600 forwards either way, scripts/h2_stress.py, h2_det.py)
                if constexpr (H2_NOP >= 0) asm volatile("s_nop %0" ::"n"(H2_NOP >= 0 ? H2_NOP : 0) : "memory");
                // filler work of this gap
                if constexpr (m < MB) {
                    if (!(HALO2_ABLATIONS && (p.h_abl & 4))) {
                        if constexpr (nks == 0) {
                            abase[m] = frag_base(m, ntap / 3, ntap % 3, rd_buf * AROWS);         // next tap: fresh base address
                            h2_ds_read(xf[nxt][m], abase[m]);
                        } else {
                            h2_ds_read(xf[nxt][m], abase[m] ^ (unsigned)(nks << 5));
                        }
                    }
                }
                {
                    // the halo piece and then the NB weight loads go into the gaps after the reads (or share the last gaps when there are few)
                    constexpr int first = NMMA - MB >= NB + 1 ? MB : (NMMA - (NB + 1));
                    constexpr int slot = m - first - 1;
                    if constexpr (slot == -1 && ls < PMAX) {
                        if (!(HALO2_ABLATIONS && (p.h_abl & 2))) stage_piece(ls, st_buf, (cc + 1) * RB, cc + 1 < nchunks * NSUB);
                    }
                    if constexpr (slot >= 0 && slot < NB) {
                        if (!(HALO2_ABLATIONS && (p.h_abl & 1))) {
                            // weights of k-step sg + AHEAD into the ring slot k-step sg - 1 has just finished with
                            const int v = c * 36 + sg + AHEAD < nsteps ? wv[slot] + w_hi(sg + AHEAD) : (int)0x80000000;      // (past the end: nothing to load, same count)
                            h2_wload_n<w_lo(sg + AHEAD)>(wf[(sg + AHEAD) % RING][slot], v, rs_w);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        });
#pragma unroll
        for (int j = 0; j < NB; ++j) wv[j] += 36864;
        H2_STAMP(3 + (c < 8 ? c : 8));
    }
    // The last AHEAD k-steps' weight loads (out-of-range dummies) and the last fragment reads are still in flight, and their destination
    // registers are dead as far as the compiler can see: without the statements below it re-used them for the epilogue's first
    // temporaries ABOVE the wait (an asm output counts as written at the statement) and a load that landed late then overwrote an
    // accumulator value on its way through such a temporary -- a few elements of a launch wrong, only when the memory system was slow
    // (tests/test_gpu_determinism.py under a second stream's load).  Every ring register is therefore "used" once more BEHIND the wait.
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int r = 0; r < RING; ++r)
#pragma unroll
        for (int j = 0; j < NB; ++j) h2_landed(wf[r][j]);
#pragma unroll
    for (int x = 0; x < XS; ++x)
#pragma unroll
        for (int i = 0; i < MB; ++i) h2_landed(xf[x][i]);
#if defined(H2_DBG) && H2_DBG == 1
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
#elif defined(H2_DBG) && H2_DBG == 2
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
    if constexpr (OCC == 2) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    H2_STAMP(12);

    // ---- epilogue: the shared one on a view of the 32 x 32 blocks as f32x4 fragments (value 4f + r of block (i, j) = fragment 4j + f:
    //      chunk 2j + (f >> 1) of this lane half, element 4 (f & 1) + r -- conv_igemm_kernel's 32x32x16 path); rows past the band dropped
    f32x4 accv[MB][NB * 4];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int f = 0; f < 4; ++f)
                accv[i][j * 4 + f] = f32x4{acc[i][j][4 * f], acc[i][j][4 * f + 1], acc[i][j][4 * f + 2], acc[i][j][4 * f + 3]};
    const int mrow = m_base + wm * WPX + l32, m_limit = m_base + npx;
    const bool ch_full = n0 + BN <= p.cout_store;
    conv_epilogue_rows_g<DT, MB, NB * 4, 2>(p, accv, mrow, m_limit, n0 + wn * WCH, fh, ch_full);
#ifdef H2_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    H2_STAMP(13);
    H2_STAMP_RT(15);
    if (blockIdx.x < 4096 && wave == 0) {
        unsigned long long t_, c_;
        unsigned hw_, xcc_;
        asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_getreg_b32 %2, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %3, hwreg(HW_REG_XCC_ID)\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "=s"(c_), "=s"(hw_), "=s"(xcc_)::"memory");
        if (lane == 0) {
            h2_trace_blocks[blockIdx.x][0] = h2_t0_rt; h2_trace_blocks[blockIdx.x][1] = t_;
            h2_trace_blocks[blockIdx.x][2] = ((unsigned long long)xcc_ << 32) | hw_; h2_trace_blocks[blockIdx.x][3] = c_ - h2_t0_c;
        }
    }
#endif
}

// ------------------------------------------------------------------------------------------- launch
template <int DT, int BM, int BN, int WM, int WN, int KC, int OCC>
static int launch_halo2_cfg(const ConvK& k, hipStream_t stream) {
    const size_t lds = halo2_lds_bytes(k.h_rows, k.h_pitch, KC);
    Y4_REQUIRE(lds <= (size_t)(160 * 1024 / OCC), Y4_EINVAL, "conv2d: halo2 tile needs %zu bytes of LDS", lds);
    auto kern = conv_halo2_kernel<DT, BM, BN, WM, WN, KC, OCC>;
    static PerDeviceOnce once;
    if (const uint64_t bit = once.due()) {
        Y4_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 / OCC));
        once.mark(bit);
    }
    hipLaunchKernelGGL(kern, dim3(k.grid_m * k.grid_n), dim3(256), lds, stream, k);
    Y4_CHECK_HIP(hipGetLastError());
    return Y4_OK;
}

template <int DT>
static int launch_halo2(int tile, const ConvK& k, hipStream_t s) {
#define Y4_H2_CASE(id, bm, bn, wm, wn, kc, occ) case id: return launch_halo2_cfg<DT, bm, bn, wm, wn, kc, occ>(k, s);
    switch (tile) { Y4_HALO2_TILES(Y4_H2_CASE) }
#undef Y4_H2_CASE
    set_error("conv2d: no halo2 tile with id %d", tile);
    return Y4_EINVAL;
}

}  // namespace y4
