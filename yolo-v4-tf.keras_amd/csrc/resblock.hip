// resblock.hip -- one residual block of a CSP stage as ONE spatially tiled kernel (16-bit dtypes, C = 128 or 64 channels):
//     y = x + Mish(BN(conv3x3(Mish(BN(conv1x1(x))))))            reference custom_layers.py:34-44 (residual_block)
// i.e. the 8 blocks of the 76x76 stage (C = 128) and the 2 blocks of the 152x152 stage (C = 64) at 608x608.
//
// Why.  As implicit GEMMs these 3x3 convs have N = C <= 128 output channels, so the M x N tile that bounds the staging
// traffic per FLOP cannot grow: every K-tile re-stages the activation rows of all 9 taps through L2 -> LDS
// (15-18 bytes per KFLOP against 9 for the 256-channel layers) and they run at 0.65 PFLOP/s.  Here a workgroup owns a
// 16x16-pixel tile: the 18x18 halo'd tile of x (all C channels) is brought into LDS ONCE, the 1x1 conv runs on it IN
// PLACE (x is pixel-local for a 1x1, so each wave overwrites exactly the rows it read), and the 3x3 conv then reads its
// pixel operand for all 9 taps straight from that tile at shifted rows -- nothing but the 3x3 weights (C*C*2 bytes per
// tap, fragment ordered, lane-linear LDS-DMA through a 2-stage ring) is streamed during the K loop, and the 1x1's
// output never exists in HBM.  Cost: the 1-pixel halo ring of the 1x1 conv is recomputed (1.27x of 10 % of the FLOPs),
// and tensors whose side is not a multiple of 16 (76, 52) have partially filled edge tiles.
//
// Numerics: every conv issues the same MFMAs (v_mfma_f32_16x16x32, K ascending in the canonical order of common.h: 64-channel block, tap, channel) on the
// same 16-bit inputs as its conv_igemm kernel and the same fp32 epilogue -> bit-identical to the unfused path.
#include <type_traits>

#include "conv_chain.h"

// experiment switches (scripts/build_variant.sh), a bit mask: 1 no 1x1 phase, 2 LeakyReLU instead of Mish, 4 the 3x3 weights
// are not streamed, 8 the x tile is loaded once, 16 no output stores, 32 no fragment reads in the 3x3 loop, 64 no barriers
// in the 3x3 loop (racy: timing only)
#ifndef RB_ABL
#define RB_ABL 0
#endif
#ifndef RB_RES_LDS
// 1 (2: also C = 64, where it spills at the 128-register cap): the residual is read from the x tile in LDS (a barrier in front of the
// 1x1 phase, which overwrites the tile in place) instead of a second time from memory.  Same bytes; measured NEUTRAL (r4): the loop's
// tail gets 1.4 k cycles shorter per tile, but the next tile's x pieces then block their issuers for 5.4 - 9.8 k cycles instead of
// 2.5 - 4.8 k -- the residual's loads had been evicting the previous epilogue's dirty output lines from L2 under the loop, which the x
// fills now have to wait for (write-through stores, RB_ST_AUX=17, bring the 2.9 - 5.8 k back; tile 46.3 k cycles against 45.9 k).
#define RB_RES_LDS 0
#endif
#ifndef RB_ST_AUX
// cache policy bits of the output stores (1 sc0, 2 nt, 16 sc1); 0: write-back.  nt: tile 49.4 k cycles, write-through 46.3 k (see above)
#define RB_ST_AUX 0
#endif
#ifndef RB_LATE
// 1: the residual's loads are issued four stream steps before the end of the 3x3 loop instead of in front of it (they are needed in
// the epilogue only), and stream step 3 -- whose ring slot is free from the start -- with steps 0..2 at the tile hand-over instead of
// behind the mid barrier.  In-kernel trace: the stretch between the mid barrier and the loop was 1.9 k cycles for the older wave of a
// SIMD and 4.1 k for the younger one (its loads queue behind the partner's), on the tile's critical path.  Bit-identical.
#define RB_LATE 1
#endif

namespace y4 {

#ifdef RB_TRACE
// In-kernel phase trace (kernel experiments only; scripts/res_trace.py): workgroup RB_TR_WG of the FULL-tile launch records s_memtime
// per wave at fixed points of its tiles RB_TR_T0 .. +2 into rb_trace_buf[tile][wave][point]; y4_rb_trace_read() copies it out.
__device__ unsigned long long rb_trace_buf[3 * 8 * 16];
#define RB_TR_WG 8
#define RB_TR_T0 1
#define RB_POINT(P)                                                                                         \
    do {                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        if (tr_on) asm volatile("s_memtime %0" : "=s"(tr_t[P]));                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
    } while (0)
#else
#define RB_POINT(P)
#endif

constexpr int RB_ACT = (RB_ABL & 2) ? Y4_ACT_LEAKY : Y4_ACT_MISH;
constexpr int RB_T = 16, RB_H = RB_T + 2, RB_WAVES = 8;

// TY = output rows of a workgroup's tile: 16 (the full 16x16 tile), or 8 / 4 -- a HALF / QUARTER tile (the same 16 columns): the
// last, partial round of a launch runs as part tiles on the compute units it would leave idle (resblock_dispatch), and so does a
// batch with fewer tiles than compute units.  Same MFMAs per output pixel in the same order: bit-identical to the full tile.
template <int C, int TY = RB_T> struct RbGeom {
    static constexpr int HR = TY + 2;                  // halo'd rows of the tile (halo'd width: RB_H)
    static constexpr int HROWS = HR * RB_H;            // pixel rows of the LDS tile (no fragment reads past the last one)
    static constexpr int NFA = HR + (2 * HR + 15) / 16;                  // pixel fragments of the 1x1 phase: rows + the two halo columns
    static constexpr int CPR = C / 8;                  // 16-byte chunks of data per pixel row of the LDS tile
    // Rows are PADDED by one chunk instead of XOR-swizzled: 16 lanes reading one chunk of 16 consecutive rows then hit 16
    // different 4-bank groups for ANY first row and chunk (row stride 272 B = 68 banks = 4 mod 64; 144 B = 36 banks), and
    // every fragment address of the 3x3 loop is "lane base + compile-time constant" -- no address VALU, one base VGPR.
    static constexpr int ROWB = C * 2 + 16;            // bytes per pixel row incl. the pad chunk
    static constexpr int XT_PIECES = (HROWS * (CPR + 1) + 63) / 64;        // 1 KB LDS-DMA pieces that fill the tile
    static constexpr int NF = C / 16;                  // output-channel fragments of a conv
    static constexpr int KS1 = C / 32;                 // MFMA k-steps of the 1x1 conv
    // The weights of both convs form ONE stream of equal steps of 64 input channels (2 MFMA k-steps x NF fragments x 1 KB):
    // P steps of the 1x1 conv, then 9 * P steps of the 3x3 conv (64-channel block, then tap: common.h).  They go through a 4-slot LDS ring, three
    // steps ahead of their use, with counted vmcnt waits -- an L2 round trip (~1 us under load) is longer than one step.
    static constexpr int STEP_BYTES = 2 * NF * 1024;
    static constexpr int P = KS1 / 2;                  // steps of the 1x1 conv
    static constexpr int NSTEPS = 10 * P;              // whole stream
    static constexpr int RING = 4;
    static constexpr int PPW = STEP_BYTES / 1024 / RB_WAVES;      // LDS-DMA pieces a wave issues per step
    static constexpr int AFF_BYTES = 4 * C * 4;        // scale1, shift1, scale3, shift3
    static constexpr int AFF_PAD = (AFF_BYTES + 1023) / 1024 * 1024;
    static constexpr int BLOB_BYTES = AFF_PAD + NSTEPS * STEP_BYTES;      // [affine | W1 | W3]
    static constexpr int L_AFF = 0, L_RING = AFF_PAD, L_XT = L_RING + RING * STEP_BYTES;
    // the weight touch's own TOUCH_LDS bytes (conv_common.h): the slack behind the tile's last pixel row inside its last 1 KB piece
    // -- written by that piece's surplus lanes and by the touch, read by nobody -- or, where the slack is smaller, one more KB
    static constexpr int XT_SLACK = XT_PIECES * 1024 - HROWS * (CPR + 1) * 16;
    static constexpr int LDS = L_XT + XT_PIECES * 1024 + (XT_SLACK >= TOUCH_LDS ? 0 : 1024);
    static constexpr int L_TOUCH = LDS - TOUCH_LDS;
    static constexpr int WN = NF / 4, WM = RB_WAVES / WN, MREP = TY / WM;         // wave grid of the 3x3 phase; NREP = 4
    static_assert(C != 64 || 2 * LDS <= 160 * 1024, "C = 64: two workgroups per compute unit (resblock_dispatch)");
    static_assert(LDS <= 160 * 1024 && MREP >= 1 && MREP * WM == TY && PPW >= 1 && PPW * RB_WAVES * 1024 == STEP_BYTES && NFA <= 3 * RB_WAVES,
                  "resblock geometry");
};

struct ResBlockK {
    const char* in;              // x view
    char* out;                   // y view
    const char* blob;            // RbGeom<C>::BLOB_BYTES (pack_resblock)
    int in_cstride, in_coff, out_cstride, out_coff;
    unsigned in_bytes, out_bytes;
    int N, S;
    int tiles_x, tiles_per_img, ntiles;
    int t_begin, t_end;          // this launch's range of PART tiles: part tile u = 16x16 tile u / (16 / TY), rows (u % (16 / TY)) * TY .. + TY
    int touch;                   // weight touch (conv_common.h) of the block's blob at kernel start
};

template <int DT, int C, int TY>
__global__ __launch_bounds__(64 * RB_WAVES, C == 64 ? 4 : 2) void resblock_kernel(const ResBlockK p) {
    using E = Elem<DT>;
    using G = RbGeom<C, TY>;
    constexpr int PARTS = RB_T / TY;
    constexpr int ROWB = G::ROWB, CPR = G::CPR, NF = G::NF, MREP = G::MREP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane & 15, g = lane >> 4;
    const int wm = wave / G::WN, wn = wave - wm * G::WN;
    const float* const aff = (const float*)(smem + G::L_AFF);
    char* const XT = smem + G::L_XT;
    char* const ring = smem + G::L_RING;

    const int GR = gridDim.x, xcd = blockIdx.x & 7, bi = blockIdx.x >> 3;
    const int nb_x = (GR - xcd + 7) >> 3;
    const int span = p.t_end - p.t_begin;
    const int t_lo = p.t_begin + (int)((int64_t)span * xcd / 8), t_hi = p.t_begin + (int)((int64_t)span * (xcd + 1) / 8);
    int t = t_lo + bi;
    // part tile -> image, tile row / column of the 16x16 tiling, first output row
    auto locate = [&](int u, int& n, int& y_out0, int& tx) {
        const int full = u / PARTS, part = u - full * PARTS;
        n = full / p.tiles_per_img;
        const int rem = full - n * p.tiles_per_img, ty = rem / p.tiles_x;
        tx = rem - ty * p.tiles_x;
        y_out0 = ty * RB_T + part * TY;
    };

    const __amdgpu_buffer_rsrc_t rb = make_rsrc(p.blob, G::BLOB_BYTES);
    const __amdgpu_buffer_rsrc_t rs_in = make_rsrc(p.in, p.in_bytes), rs_out = make_rsrc(p.out, p.out_bytes);
    // the blob (affine tables + both convs' weight streams: every tile streams all of it) -> this XCD's L2, shared out over the
    // workgroups (conv_common.h: weight_touch); the dwords land in LDS bytes of their own (RbGeom::L_TOUCH)
    if (p.touch != 0) weight_touch(rb, smem + G::L_TOUCH, 0, G::BLOB_BYTES, wave, RB_WAVES, lane);
    // affine -> LDS once per workgroup
    for (int u = wave; u < G::AFF_PAD / 1024; u += RB_WAVES)
        buffer_load16_lds(rb, smem + __builtin_amdgcn_readfirstlane(u * 1024), u * 1024 + lane * 16, 0);
    // step `st` of the weight stream -> ring slot st % RING (fragment ordered in memory: a plain lane-linear copy);
    // every wave issues exactly PPW pieces per step, which is what the counted waits below rely on
    auto stage_w = [&](int st) {
        char* const dst = ring + (st & (G::RING - 1)) * G::STEP_BYTES;
#pragma unroll
        for (int k = 0; k < G::PPW; ++k) {
            const int u = wave + k * RB_WAVES;
            buffer_load16_lds(rb, dst + __builtin_amdgcn_readfirstlane(u * 1024), G::AFF_PAD + u * 1024 + lane * 16, st * G::STEP_BYTES);
        }
    };
    // halo'd tile of x -> XT (out-of-image pixels and the 12 spare rows read as zeros)
    // The tile is written lane-linearly in 16-byte slots (slot = piece * 64 + lane; a pixel row is CPR data slots + the pad slot), a
    // wave takes pieces wave, wave + 8, ...: from one piece to the next every lane's slot advances by the same 512, so its (row,
    // column, chunk) and its byte offset advance by constants plus two carries.  ~15 VALU instructions per piece instead of the ~35
    // of two divisions by constants and the full offset polynomial: the wave issues one VALU instruction per ~9 cycles, and the
    // trace priced the 12 pieces of a C=128 tile at 3.3 k cycles of the epilogue they are issued under.
    auto load_x = [&](int tile) {
        int n, yo, tx;
        locate(tile, n, yo, tx);
        const int y0 = yo - 1, x0 = tx * RB_T - 1;
        constexpr int SPR = CPR + 1, DSLOT = RB_WAVES * 64, DHP = DSLOT / SPR, DCH = DSLOT % SPR, DHY = DHP / RB_H, DHX = DHP % RB_H;
        static_assert(DCH + SPR - 1 < 2 * SPR && DHX + RB_H < 2 * RB_H, "one carry per step");
        int slot0 = wave * 64 + lane;
        asm volatile("" : "+v"(slot0));        // (per tile: hoisted out of the tile loop, the pieces' positions would cost ~3 registers each)
        const int hp0 = slot0 / SPR;
        int ch = slot0 - hp0 * SPR, hy = hp0 / RB_H;
        int hx = hp0 - hy * RB_H;
        const int pix_b = p.in_cstride * 2, row_b = p.S * pix_b;           // bytes per pixel / per image row of the view
        int off = (((n * p.S + y0) * p.S + x0) * p.in_cstride + p.in_coff) * 2 + hy * row_b + hx * pix_b + ch * 16;
        const int d_off = DHY * row_b + DHX * pix_b + DCH * 16, c_ch = pix_b - SPR * 16, c_hx = row_b - RB_H * pix_b;
#pragma unroll
        for (int k = 0; k * RB_WAVES < G::XT_PIECES; ++k) {
            const int u = wave + k * RB_WAVES;
            if (u < G::XT_PIECES) {                                         // (wave-uniform; false for some waves' last piece only)
                const bool ok = ch < CPR && hy < G::HR && (unsigned)(y0 + hy) < (unsigned)p.S && (unsigned)(x0 + hx) < (unsigned)p.S;
                const int off_x = (RB_ABL & 128) ? (off & ~63) + (lane & 3) * 16 : off;      // (timing experiment: 64-byte aligned quads, wrong data)
                buffer_load16_lds(rs_in, XT + __builtin_amdgcn_readfirstlane(u * 1024), ok ? off_x : (int)0x80000000, 0);
            }
            ch += DCH; hx += DHX; hy += DHY; off += d_off;
            if (ch >= SPR) { ch -= SPR; hx += 1; off += c_ch; }
            if (hx >= RB_H) { hx -= RB_H; hy += 1; off += c_hx; }
        }
    };
    if (t < t_hi) { load_x(t); stage_w(0); stage_w(1); stage_w(2); if (RB_LATE) stage_w(3); }

#ifdef RB_TRACE
    int tr_i = 0;
    unsigned long long tr_t[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
    for (; t < t_hi; t += nb_x) {
        int n, y_out0, tx;
        locate(t, n, y_out0, tx);
#ifdef RB_TRACE
        const bool tr_on = TY == RB_T && C == RB_TRACE && blockIdx.x == RB_TR_WG && tr_i >= RB_TR_T0 && tr_i < RB_TR_T0 + 3;
#endif
        RB_POINT(0);                           // arrives at the tile barrier
        wait_vmcnt_then_barrier<0>();          // x tile and stream steps 0..2 landed; all waves left the previous tile
        RB_POINT(1);                           // barrier passed

        u32x4 res[MREP][2];                    // residual = x at this lane's output pixels / channels
        constexpr bool RES_LDS = RB_RES_LDS && (C == 128 || RB_RES_LDS > 1);       // (C = 64: at the 128-register cap it would spill)
        if (RES_LDS) {
#pragma unroll
            for (int i = 0; i < MREP; ++i)
#pragma unroll
                for (int c = 0; c < 2; ++c)
                    res[i][c] = *(const u32x4*)(XT + ((wm * MREP + i + 1) * RB_H + q + 1) * ROWB + chunk_channel(wn * 64, c, g) * 2);
            // every wave has its residual before any wave's 1x1 phase overwrites the tile
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }

        // ================= phase A: t = Mish(BN(conv1x1(x))) on the halo'd tile, IN PLACE; zero outside the image.
        // 21 pixel fragments (18 rows + 3 fragments holding the two halo columns); a wave takes fragments wave, wave+8,
        // wave+16 TOGETHER, so that each weight fragment is read from LDS once per wave and 16-24 MFMAs are independent.
        auto phase_a = [&](auto nfr_c) {
            constexpr int NFR = decltype(nfr_c)::value;
            char* row[NFR];
            bool inside[NFR];
#pragma unroll
            for (int k = 0; k < NFR; ++k) {
                const int f = wave + k * RB_WAVES;
                int hy, hx;
                if (f < G::HR) { hy = f; hx = 1 + q; }
                else { const int pp = min((f - G::HR) * 16 + q, 2 * G::HR - 1); hy = pp >> 1; hx = (pp & 1) * (RB_H - 1); }
                const int r = hy * RB_H + hx;
                row[k] = XT + r * ROWB;
                const int gy = y_out0 - 1 + hy, gx = tx * RB_T - 1 + hx;
                inside[k] = (unsigned)gy < (unsigned)p.S && (unsigned)gx < (unsigned)p.S;
            }
            f32x4 acc[NFR][NF];
#pragma unroll
            for (int k = 0; k < NFR; ++k)
#pragma unroll
                for (int j = 0; j < NF; ++j) acc[k][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < G::KS1; ++s) {
                u32x4 xf[NFR], wf[NF];
#pragma unroll
                for (int k = 0; k < NFR; ++k) xf[k] = *(const u32x4*)(row[k] + (s * 4 + g) * 16);
#pragma unroll
                for (int j = 0; j < NF; ++j) wf[j] = *(const u32x4*)(ring + (s >> 1) * G::STEP_BYTES + (((s & 1) * NF + j) * 64 + lane) * 16);
#pragma unroll
                for (int k = 0; k < NFR; ++k)
#pragma unroll
                    for (int j = 0; j < NF; ++j) Mma<DT>::run(acc[k][j], wf[j], xf[k]);
            }
#pragma unroll
            for (int c = 0; c < NF / 2; ++c) {
                float sc[8], sh[8];
#pragma unroll
                for (int h = 0; h < 8; h += 4) {
                    const int ch = chunk_channel(0, c, g) + h;
                    const f32x4 s4 = *(const f32x4*)(aff + ch), h4 = *(const f32x4*)(aff + C + ch);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { sc[h + e] = s4[e]; sh[h + e] = h4[e]; }
                }
#pragma unroll
                for (int k = 0; k < NFR; ++k) {
                    float v[8];
                    bn_act4<true, RB_ACT>(acc[k][2 * c], sc, sh, v);
                    bn_act4<true, RB_ACT>(acc[k][2 * c + 1], sc + 4, sh + 4, v + 4);
                    u32x4 pk;
                    E::store_chunk(&pk, v);
                    if (!inside[k]) pk = u32x4{0u, 0u, 0u, 0u};
                    *(u32x4*)(row[k] + (4 * c + g) * 16) = pk;
                }
            }
        };
        if (!(RB_ABL & 1)) {                   // NFA fragments over 8 waves: fragments wave, wave + 8, wave + 16 (wave-uniform counts)
            if (wave + 2 * RB_WAVES < G::NFA) phase_a(std::integral_constant<int, 3>{});
            else if (wave + RB_WAVES < G::NFA) phase_a(std::integral_constant<int, 2>{});
            else if (wave < G::NFA) phase_a(std::integral_constant<int, 1>{});
        }
        RB_POINT(2);                           // 1x1 phase done (MFMAs, Mish, in-place stores issued)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        RB_POINT(3);                           // mid barrier passed

        // ================= phase C: y = x + Mish(BN(conv3x3(t))): 9 taps x C/64 stream steps, weights through the ring.
        // Software pipelined at MFMA k-step (32-channel) granularity: the fragments of half-step h+1 are read from LDS
        // while the 4*MREP MFMAs of half-step h issue; the wait + barrier that admits the next stream step sits between
        // the two halves of a step.  Fully unrolled (taps, slots and wait counts are compile-time).
        {
            // residual = x at this lane's output pixels / channels, straight from memory (its round trip hides under the loop)
            // (through the buffer descriptors: one 32-bit offset per access -- a dead pixel's is out of range, its load reads zeros and
            // its store is dropped -- instead of 64-bit pointer arithmetic per access)
            int pix[MREP];
#pragma unroll
            for (int i = 0; i < MREP; ++i) {
                const int gy = y_out0 + wm * MREP + i, gx = tx * RB_T + q;
                pix[i] = gy < p.S && gx < p.S ? (n * p.S + gy) * p.S + gx : -1;
            }
            auto load_res = [&]() {
#pragma unroll
                for (int i = 0; i < MREP; ++i) {
                    const int o = pix[i] < 0 ? (int)0x80000000 : (pix[i] * p.in_cstride + p.in_coff) * 2;
#pragma unroll
                    for (int c = 0; c < 2; ++c) res[i][c] = buffer_load16(rs_in, o + chunk_channel(wn * 64, c, g) * 2);
                }
            };
            if (!RB_LATE && !RES_LDS) load_res();
            // RB_LATE: the 2 MREP residual loads go out at the top of stream step S0 and stay in flight beside the stream's own loads:
            // a wave's loads retire in order, so the counted waits of steps S0 .. S0 + 2 allow that many more outstanding
            constexpr int S0 = G::NSTEPS - 4, RES_LOADS = RES_LDS ? 0 : 2 * MREP;
            f32x4 acc[MREP][4];
#pragma unroll
            for (int i = 0; i < MREP; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            // stream steps up to P+3 (4 slots: slot 3 is unused so far, slots < P held the 1x1 weights)
#pragma unroll
            for (int st = RB_LATE ? 4 : 3; st <= G::P + 3; ++st) stage_w(st);
            const char* const xbase = XT + ((wm * MREP) * RB_H + q) * ROWB + g * 16;       // fragment 0, tap (0, 0), chunk g
            u32x4 xf[2][MREP], wf[2][4];
            auto read_frags = [&](int buf, int st, int kk) {              // all arguments are compile-time after unrolling
                if ((RB_ABL & 32) && st > G::P) return;
                const int ks = st - G::P;
                const int cb = ks / 9, tap = ks - cb * 9;                  // tap `tap` of 64-channel block `cb`: the canonical K order (common.h)
                const int ky = tap / 3, kx = tap - ky * 3;
                const char* const wst = ring + (st & (G::RING - 1)) * G::STEP_BYTES;
#pragma unroll
                for (int i = 0; i < MREP; ++i) {
                    xf[buf][i] = *(const u32x4*)(xbase + ((i + ky) * RB_H + kx) * ROWB + (cb * 8 + kk * 4) * 16);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) wf[buf][j] = *(const u32x4*)(wst + ((kk * NF + wn * 4 + j) * 64 + lane) * 16);
            };
            auto mma = [&](int buf) {
#pragma unroll
                for (int i = 0; i < MREP; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        Mma<DT>::run(acc[i][j], wf[buf][j], xf[buf][i]);
            };
            RB_POINT(4);                           // residual loads + stream steps issued
            read_frags(0, G::P, 0);                                        // step P landed at the top of the tile
#pragma unroll
            for (int st = G::P; st < G::NSTEPS; ++st) {
                if (st == G::P + 3 * G::P) RB_POINT(5);                    // a third of the stream done
                if (st == G::P + 6 * G::P) RB_POINT(6);                    // two thirds
                if (RB_LATE && !RES_LDS && st == S0) load_res();
                // sched_barrier: hipcc otherwise re-serialises the pipeline into "read one fragment, wait for it, 4 MFMAs"
                // (fewer live registers, but every wait exposes the LDS latency); pinned, the 8-12 reads of the next
                // half-step are all in flight while the 4*MREP MFMAs of this one issue
                read_frags(1, st, 1);
                __builtin_amdgcn_sched_barrier(0);
                mma(0);
                __builtin_amdgcn_sched_barrier(0);
                if (st + 1 < G::NSTEPS) {
                    // step st+1 has landed when only the younger steps (issued through st+3) are still in flight; the
                    // barrier also proves every wave has read slot(st) to the end, which the next stage_w overwrites: the
                    // wait includes lgkmcnt(0) -- the kk = 1 reads of slot(st) issued above are not consumed before mma(1),
                    // so without it "read to the end" would hold by timing only (they have had all of mma(0) to land: free)
                    constexpr int LAST = G::NSTEPS - 1;
                    const int younger = (st + 3 < LAST ? st + 3 : LAST) - (st + 1);
                    // (st is a compile-time constant after unrolling: one of the variants survives)
                    const int extra = RB_LATE && st >= S0 && st <= S0 + 2 ? RES_LOADS : 0;
                    static_assert(S0 > G::P && 2 * G::PPW + RES_LOADS < 64, "window of the residual's loads");
                    if (RB_ABL & 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (timing experiment: no barrier)
                    else wait_vmcnt_lgkm_then_barrier_n<2 * G::PPW + RES_LOADS>(younger == 0 ? 0 : (younger >= 2 ? 2 : 1) * G::PPW + extra);
                    if (!(RB_ABL & 4) && st + 4 < G::NSTEPS) stage_w(st + 4);
                    read_frags(0, st + 1, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                mma(1);
                __builtin_amdgcn_sched_barrier(0);
            }
            RB_POINT(7);                           // all taps done
            // every wave is done with the tile and the ring: bring in the next tile under this tile's epilogue
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            RB_POINT(8);                           // end barrier passed
            if constexpr (RB_LATE && !RES_LDS) {
            // The residual has landed (the loop's last counted wait was vmcnt(0)), but the compiler does not know: its own wait in
            // front of the first use would sit behind the next tile's loads issued below and, counted conservatively, hold the
            // epilogue until most of THOSE have landed too.  Using the registers here puts that wait where it costs nothing.
#pragma unroll
                for (int i = 0; i < MREP; ++i)
#pragma unroll
                    for (int c = 0; c < 2; ++c) asm volatile("" : "+v"(res[i][c]));
            }
            const bool more = t + nb_x < t_hi;
            if (more) { if (!(RB_ABL & 8)) load_x(t + nb_x); stage_w(0); stage_w(1); stage_w(2); if (RB_LATE) stage_w(3); }
            RB_POINT(9);                           // next tile's loads issued
            float sc3[16], sh3[16];
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int h = 0; h < 8; h += 4) {
                    const int ch = chunk_channel(wn * 64, c, g) + h;
                    const f32x4 s4 = *(const f32x4*)(aff + 2 * C + ch), h4 = *(const f32x4*)(aff + 3 * C + ch);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { sc3[c * 8 + h + e] = s4[e]; sh3[c * 8 + h + e] = h4[e]; }
                }
#pragma unroll
            for (int i = 0; i < MREP; ++i) {
                float v[16];
#pragma unroll
                for (int j = 0; j < 4; ++j) bn_act4<true, RB_ACT>(acc[i][j], sc3 + j * 4, sh3 + j * 4, v + j * 4);
                const int o = pix[i] < 0 ? (int)0x80000000 : (pix[i] * p.out_cstride + p.out_coff) * 2;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    float rv[8];
                    E::load_chunk(&res[i][c], rv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[c * 8 + e] += rv[e];      // residual Add (custom_layers.py:44), after the activation
                    u32x4 pk;
                    E::store_chunk(&pk, v + c * 8);
                    if (!(RB_ABL & 16) || pk[0] == 0x12345678u) buffer_store16<RB_ST_AUX>(rs_out, pk, o + chunk_channel(wn * 64, c, g) * 2);
                }
            }
        }
        RB_POINT(10);                          // epilogue done, stores issued
#ifdef RB_TRACE
        if (tr_on && lane == 0) {
#pragma unroll
            for (int k = 0; k < 11; ++k) rb_trace_buf[((tr_i - RB_TR_T0) * 8 + wave) * 16 + k] = tr_t[k];
        }
        ++tr_i;
#endif
    }
}

#ifdef RB_TRACE
}  // namespace y4
extern "C" int y4_rb_trace_read(unsigned long long* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(y4::rb_trace_buf), sizeof(unsigned long long) * 3 * 8 * 16);
}
namespace y4 {
#endif

// ------------------------------------------------------------------------------------------------ launch
bool resblock_supported(int dtype, int c) { return dtype != Y4_F32 && (c == 128 || c == 64); }
size_t resblock_blob_bytes(int c) { return c == 128 ? RbGeom<128>::BLOB_BYTES : RbGeom<64>::BLOB_BYTES; }

template <int DT, int C, int TY>
static int resblock_run(const ResBlockK& k, int slots, hipStream_t stream) {
    // `slots` workgroups fit the GPU at once (the same number on every XCD); persistent over this launch's part tiles
    int grid = slots;
    const int span = k.t_end - k.t_begin;
    if (grid > ((span + 7) & ~7)) grid = (span + 7) & ~7;
    if (grid < 8) grid = 8;
    constexpr int lds = RbGeom<C, TY>::LDS;
    static PerDeviceOnce once;
    if (const uint64_t bit = once.due()) {
        Y4_CHECK_HIP(hipFuncSetAttribute((const void*)resblock_kernel<DT, C, TY>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        once.mark(bit);
    }
    hipLaunchKernelGGL((resblock_kernel<DT, C, TY>), dim3(grid), dim3(64 * RB_WAVES), lds, stream, k);
    Y4_CHECK_HIP(hipGetLastError());
    return Y4_OK;
}

// Y4_RB_PARTS=0 runs every tile as a full 16x16 tile (A/B measurements; results are the same either way)
static bool rb_parts_enabled() {
    static const bool on = [] { const char* e = getenv("Y4_RB_PARTS"); return !(e && e[0] == '0'); }();
    return on;
}

template <int DT, int C>
static int resblock_dispatch(ResBlockK& k, hipStream_t stream) {
    static int n_cus[64] = {0};
    int dev = 0;
    Y4_CHECK_HIP(hipGetDevice(&dev));
    if (n_cus[dev & 63] == 0) {
        int v = 0;
        Y4_CHECK_HIP(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
        n_cus[dev & 63] = v > 0 ? v : 256;
    }
    // one workgroup per CU for C = 128 (LDS), two for C = 64 (79 KB each: their phases overlap)
    const int slots = (n_cus[dev & 63] & ~7) * (C == 64 ? 2 : 1);
    // Round quantisation: 800 tiles on 256 slots are 3 full rounds and one round on 32 slots (4 tile times for 3.1 of work).  The
    // tiles of the partial last round run as half / quarter tiles (8 / 4 output rows: the halo'd 1x1 phase is repeated per part,
    // the 3x3 phase splits) when all their parts fit one round; a launch with fewer tiles than slots is split the same way.
    constexpr int MAXP = C == 128 ? 4 : 2;                 // C = 64: eight wave rows need at least 8 output rows
    const int full = k.ntiles / slots * slots, rem = k.ntiles - full;
    int parts = 1;
    if (rb_parts_enabled() && rem > 0)
        for (int pp = 2; pp <= MAXP; pp *= 2)
            if (rem * pp <= slots) parts = pp;
    if (parts == 1) {
        k.t_begin = 0; k.t_end = k.ntiles;
        return resblock_run<DT, C, RB_T>(k, slots, stream);
    }
    if (full > 0) {
        k.t_begin = 0; k.t_end = full;
        if (int r = resblock_run<DT, C, RB_T>(k, slots, stream)) return r;
    }
    k.t_begin = full * parts; k.t_end = k.ntiles * parts;
    if (parts == 2) return resblock_run<DT, C, RB_T / 2>(k, slots, stream);
    if constexpr (MAXP >= 4) return resblock_run<DT, C, RB_T / 4>(k, slots, stream);
    return Y4_EINVAL;
}

int resblock_launch(int dtype, int c, const void* in, int n, int side, int in_cstride, int in_coff, const void* blob, void* out,
                    int out_cstride, int out_coff, hipStream_t stream) {
    Y4_REQUIRE(resblock_supported(dtype, c), Y4_EINVAL, "resblock: dtype %d / %d channels not supported", dtype, c);
    Y4_REQUIRE(in && blob && out && n > 0 && side > 0, Y4_EINVAL, "resblock: null pointer / empty batch");
    Y4_REQUIRE(in_cstride % 8 == 0 && in_coff % 8 == 0 && out_cstride % 8 == 0 && out_coff % 8 == 0, Y4_EINVAL,
               "resblock: views not 16-byte aligned");
    const int64_t in_bytes = (int64_t)n * side * side * in_cstride * 2;
    const int64_t out_bytes = (int64_t)n * side * side * out_cstride * 2;
    Y4_REQUIRE(in_bytes < (1ll << 31) && out_bytes < (1ll << 31), Y4_EINVAL,
               "resblock: input (%lld B) or output (%lld B) exceeds the 2 GiB buffer-descriptor range", (long long)in_bytes, (long long)out_bytes);
    ResBlockK k{};
    k.in = (const char*)in; k.out = (char*)out; k.blob = (const char*)blob;
    k.in_cstride = in_cstride; k.in_coff = in_coff; k.out_cstride = out_cstride; k.out_coff = out_coff;
    k.in_bytes = (unsigned)in_bytes; k.out_bytes = (unsigned)out_bytes;
    k.N = n; k.S = side;
    k.touch = weight_touch_enabled() ? 1 : 0;
    k.tiles_x = (side + RB_T - 1) / RB_T; k.tiles_per_img = k.tiles_x * k.tiles_x; k.ntiles = n * k.tiles_per_img;
    if (c == 128) return dtype == Y4_BF16 ? resblock_dispatch<Y4_BF16, 128>(k, stream) : resblock_dispatch<Y4_F16, 128>(k, stream);
    return dtype == Y4_BF16 ? resblock_dispatch<Y4_BF16, 64>(k, stream) : resblock_dispatch<Y4_F16, 64>(k, stream);
}

// ------------------------------------------------------------------------------------------------ weight packing
// (cout, cin, k, k) float32 -> MFMA A fragments, natural K order, chunked output-channel layout (same as csp_stage.hip):
//   out[((s*NREP + j)*64 + lane)*8 + e] = W[ch][ci][tap], ch = ((j>>1)*4 + (i>>2))*8 + (j&1)*4 + (i&3), i = lane & 15,
//   k-step s = 2*(cb64*k*k + tap) + half (64-channel block cb64 -> tap -> half: common.h), ci = 32*(2*cb64 + half) + 8*(lane>>4) + e
template <int DT>
__global__ void rb_pack_frag_kernel(const float* __restrict__ w, typename Elem<DT>::type* __restrict__ out, int cout, int cin, int kk) {
    const int nrep = cout / 16, cbs = cin / 32, total = cout * cin * kk;
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int e = idx & 7, lane = (idx >> 3) & 63, r = idx >> 9;
        const int j = r % nrep, s = r / nrep;
        // canonical K order (common.h): 64-channel block -> tap -> the block's two 32-channel k-steps
        const int step = s >> 1, cb64 = step / kk, tap = step - cb64 * kk, cb = cb64 * 2 + (s & 1);
        const int i = lane & 15, gg = lane >> 4;
        const int ch = ((j >> 1) * 4 + (i >> 2)) * 8 + (j & 1) * 4 + (i & 3);
        const int ci = 32 * cb + 8 * gg + e;
        out[idx] = Elem<DT>::st(w[((int64_t)ch * cin + ci) * kk + tap]);
    }
}
__global__ void rb_affine_kernel(const float* s1, const float* h1, const float* s3, const float* h3, float* dst, int c) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < c) { dst[i] = s1[i]; dst[c + i] = h1[i]; dst[2 * c + i] = s3[i]; dst[3 * c + i] = h3[i]; }
}

// w1 / w3: Darknet-order float32 kernels of the block's 1x1 and 3x3 convs (device); scale / shift: their folded BN
int pack_resblock(int dtype, int c, const float* w1, const float* scale1, const float* shift1, const float* w3, const float* scale3,
                  const float* shift3, void* blob, hipStream_t stream) {
    Y4_REQUIRE(resblock_supported(dtype, c) && w1 && w3 && scale1 && shift1 && scale3 && shift3 && blob, Y4_EINVAL, "pack_resblock: bad argument");
    const int w1_bytes = c * c * 2, aff_pad = (4 * c * 4 + 1023) / 1024 * 1024;
    char* b = (char*)blob;                           // [affine | W1 fragments | W3 fragments]
    if (dtype == Y4_BF16) {
        hipLaunchKernelGGL(rb_pack_frag_kernel<Y4_BF16>, dim3((c * c + 255) / 256), dim3(256), 0, stream, w1, (uint16_t*)(b + aff_pad), c, c, 1);
        hipLaunchKernelGGL(rb_pack_frag_kernel<Y4_BF16>, dim3((9 * c * c + 255) / 256), dim3(256), 0, stream, w3, (uint16_t*)(b + aff_pad + w1_bytes), c, c, 9);
    } else {
        hipLaunchKernelGGL(rb_pack_frag_kernel<Y4_F16>, dim3((c * c + 255) / 256), dim3(256), 0, stream, w1, (_Float16*)(b + aff_pad), c, c, 1);
        hipLaunchKernelGGL(rb_pack_frag_kernel<Y4_F16>, dim3((9 * c * c + 255) / 256), dim3(256), 0, stream, w3, (_Float16*)(b + aff_pad + w1_bytes), c, c, 9);
    }
    hipLaunchKernelGGL(rb_affine_kernel, dim3(1), dim3(128), 0, stream, scale1, shift1, scale3, shift3, (float*)b, c);
    Y4_CHECK_HIP(hipGetLastError());
    return Y4_OK;
}

}  // namespace y4
