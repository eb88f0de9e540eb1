// stem_down.hip -- convs 0 and 1 of the plan as ONE kernel (16-bit paths):
//   c0 = LeakyReLU(BN(conv3x3 s1 'same', 3 -> 32))            reference custom_layers.py:101
//   c1 = LeakyReLU(BN(ZeroPad((1,0),(1,0)) + conv3x3 s2 'valid', 32 -> 64))   reference custom_layers.py:102, :9-12
// Unfused, c0 is the largest tensor of the network (608^2 x 32 per image): writing it and reading it back is
// ~1.5 GB of HBM traffic per 32 images for 0.5 % of the FLOPs.  Here a 1024-thread workgroup owns one OUTPUT ROW (n, ho) of c1:
//   1. the c1 weights (64 x 288) are staged into LDS once (buffer_load ... lds),
//   2. the three c0 rows 2ho-1 .. 2ho+1 it needs are COMPUTED into LDS with the MFMA stem (image patches loaded from
//      global memory into registers in the K layout of stem_common.h), BN + LeakyReLU applied, stored as 16-bit,
//   3. the stride-2 3x3 conv runs out of LDS: 9 taps x (Wo/16) pixel fragments x 4 channel fragments of MFMAs,
//   4. BN + LeakyReLU, 16-byte NHWC stores.
// c0 never exists in HBM.  A workgroup is PERSISTENT over a band of consecutive output rows (9728 rows / 256 CUs = 38 at
// 608x608 batch 32): the c1 weights are staged once, and the three c0 rows live in a 3-slot ring (c0 row y -> slot
// (y + 1) mod 3), so going from output row ho to ho+1 keeps c0 row 2ho+1 and computes only the two new ones (the band's
// first row and every image's first row compute all three): 1.03x stem work instead of the 1.5x of one workgroup per row.  LDS image of the c0 strip: [3 rows][2 column-parity planes][Wo+1 slots][64 B]; splitting
// even and odd columns into planes turns the stride-2 tap walk into unit-stride slot reads (conflict-free with the
// usual XOR swizzle), and slot 0 of the odd plane is the zero column left of the image.
#include "conv_common.h"
#include "stem_common.h"

namespace y4 {

struct StemDownK {
    const void* img;             // [N, S, S, 3] float32 in [0,1], or uint8 before the /255 (template IMG)
    const u32x4* stem_frag;      // [2][64] 16-byte weight fragments of the stem (see pack_stem_kernel)
    const float* s0_scale;       // [32]
    const float* s0_shift;
    const char* w1;              // packed c1 weights [cout_pad][9][32] (16-bit)
    const float* s1_scale;       // [64..]
    const float* s1_shift;
    char* out;                   // c1 view
    int out_cstride, out_coff;
    int N, S, act0, act1;
    unsigned w1_bytes;
};

#ifndef SD_W8
// Experiment switch, measured SLOWER and off: 1 = EIGHT waves (4 pixel rows x 2 channel halves) with conv 1's weights in REGISTERS
// (72 per wave; a 512-thread workgroup may use 256) instead of sixteen waves that re-read their 18 weight fragments from LDS for
// every output row -- the conv phase is bound by its LDS reads (720 ds_read_b128 per row = 5.8 k of its ~7 k cycles, in-kernel
// trace), and this halves them.  Result (same box, 608^2 batch 32): 321 us against 289-291; 416^2 batch 64 fp16: 317 against 257.
// With two waves per SIMD the stem phase takes 7-8 k cycles instead of 2.3-4.5 k: a wave issues one VALU instruction per ~9
// cycles at most, so two waves cannot fill the vector pipe through their own stalls, and the conv phase (5.4-6.7 k) gains nothing
// because the MFMAs of ten waves' worth of fragments now queue behind two waves' issue.  Bit-identical either way.
#define SD_W8 0
#endif
constexpr int SD_WAVES = SD_W8 ? 8 : 16, SD_WM = SD_W8 ? 4 : 8, SD_WN = 2;      // the block is alone on its CU (LDS), so it
                                                                                // brings its own latency hiding
constexpr int SD_LDS_W = SD_W8 ? 0 : 9 * 64 * 64;       // conv 1's weights in LDS: [9 taps][64 rows][64 B] (sixteen-wave form only)
constexpr int SD_UNROLL = 4;                            // stem tiles whose gathers are in flight together
#ifndef SD_PRELOAD
// 1: on a "regular" output row (two new conv-0 rows, every patch inside the image) a wave's first SD_PRE stem tiles are loaded
// into registers one output row AHEAD -- issued before the previous row's conv phase, which hides their round trip -- and its
// remaining tiles are requested first thing in the stem phase, so that they land while the preloaded ones are converted,
// multiplied and stored.  Before, every wave of the workgroup started a row by waiting for its own loads (the stem phase of a
// row was 40 % idle: in-kernel trace, LABNOTES.md section 4.2a).  Same values through the same MFMAs: bit-identical.
#define SD_PRELOAD 1
#endif
#ifndef SD_TAPBAR
#define SD_TAPBAR 1
#endif
#ifndef SD_PRE_TILES
#define SD_PRE_TILES 1
#endif
// (measured, same box, 608^2 batch 32, us per launch: no preload 302 | 3 tiles 268 | 2 tiles 259-262, 241 | 1 tile 234.5 against 241: what
//  pays is that NO wave starts its stem phase by waiting -- one tile is there, the other four are requested at once and land while
//  it is converted and multiplied -- and more tiles held across the conv phase only cost that phase registers)
constexpr int SD_PRE = SD_PRE_TILES;                    // preloaded stem tiles per wave (8 registers each across the conv phase)
// the generic stem loop then serves only the rows that are NOT regular (the first row of a band or image, the last of an image:
// ~2 % of the rows): two tiles in flight instead of four keep its register peak below what the preloaded tiles leave free
constexpr int SD_GEN_UNROLL = SD_PRELOAD ? 2 : SD_UNROLL;

static __device__ __forceinline__ int sd_swz(int row) { return (row >> 1) & 3; }

#ifdef SD_TRACE
// In-kernel phase trace (kernel experiments only; scripts/stem_trace.py): workgroup SD_TR_WG records s_memtime per wave at fixed
// points of its output rows SD_TR_R0 .. +3 into sd_trace_buf[row][wave][point]; y4_sd_trace_read() copies it out.
__device__ unsigned long long sd_trace_buf[4 * 16 * 8];
#define SD_TR_WG 8
#define SD_TR_R0 5
#define SD_POINT(P)                                                                                         \
    do {                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
        if (tr_on) asm volatile("s_memtime %0" : "=s"(tr_t[P]));                                            \
        __builtin_amdgcn_sched_barrier(0);                                                                  \
    } while (0)
#else
#define SD_POINT(P)
#endif

template <int DT, int MFW, class IMG>     // MFW = max pixel fragments per wave row: ceil((Wo/16) / SD_WM)
__global__ __launch_bounds__(64 * SD_WAVES) void stem_down_kernel(const StemDownK p) {
    using E = Elem<DT>;
    using T = typename E::type;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int S = p.S, Wo = S >> 1, PW = Wo + 1;
    const int MF = Wo >> 4;                                   // pixel fragments per output row
    char* const lds_w = smem;                                 // [9 taps][64 rows][64 B] (SD_W8 = 0)
    char* const lds_s = smem + SD_LDS_W;                      // [3 rows][2 planes][PW slots][64 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane & 15, g = lane >> 4;
    // this workgroup's band of output rows (row index = n * Wo + ho)
    const int rows_total = p.N * Wo, per = (rows_total + (int)gridDim.x - 1) / (int)gridDim.x;
    const int r_begin = blockIdx.x * per, r_end = min(r_begin + per, rows_total);
    if (r_begin >= r_end) return;

    const int wm = wave % SD_WM, wn = wave / SD_WM;
    // ---- 1. c1 weights.  A lane's 16 accumulator values are 16 consecutive channels: MFMA row i = g'*4 + r' of channel fragment
    //         jn holds channel g'*16 + jn*4 + r'.
#if SD_W8
    //         Into registers, once: fragment (tap t, channel fragment wn*2 + j) = 8 input channels 8g.. of this lane's row
    u32x4 wreg[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ch = (q >> 2) * 16 + (wn * 2 + j) * 4 + (q & 3);
            wreg[t][j] = *(const u32x4*)(p.w1 + ((ch * 9 + t) * 32 + g * 8) * 2);
        }
#else
    //         Into LDS: row (tap*64 + pr), pr = jn*16 + i.
    {
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(p.w1, p.w1_bytes);
        for (int u = wave; u < 9 * 4; u += SD_WAVES) {        // unit = 16 rows = 1024 B = one wave-wide load
            const int t = u >> 2, pr = (u & 3) * 16 + (lane >> 2), chunk = lane & 3;
            const int jn = pr >> 4, i = pr & 15;
            const int ch = (i >> 2) * 16 + jn * 4 + (i & 3);
            const int row = t * 64 + pr;
            const int voff = ((ch * 9 + t) * 32 + ((chunk ^ sd_swz(row)) * 8)) * 2;
            buffer_load16_lds(rs, lds_w + __builtin_amdgcn_readfirstlane(u * 1024), voff, 0);
        }
    }
#endif

    if (tid < 3 * 4) {                                        // the zero column left of the image, once for the three slots
        const int row = ((tid >> 2) * 2 + 1) * PW;
        *(u32x4*)(lds_s + row * 64 + (tid & 3) * 16) = u32x4{0u, 0u, 0u, 0u};
    }
    const int64_t img_bytes = (int64_t)p.N * S * S * 3 * (int64_t)sizeof(IMG);
    const __amdgpu_buffer_rsrc_t rs_img = make_rsrc(p.img, img_bytes < 0xffffffffll ? (unsigned)img_bytes : 0xffffffffu);
    const u32x4 wf0 = p.stem_frag[lane], wf1 = p.stem_frag[64 + lane];
    // BN scale / shift of both convs live in LDS ([scale0 32 | shift0 32 | scale1 64 | shift1 64] floats behind the prefetch scratch)
    // and are read where they are used, once per output row: 32 registers that are not held across the other conv's phase
    float* const lds_aff = (float*)(lds_s + 3 * 2 * PW * 64 + 1024);
    if (tid < 32) { lds_aff[tid] = p.s0_scale[tid]; lds_aff[32 + tid] = p.s0_shift[tid]; }
    else if (tid < 96) { lds_aff[64 + tid - 32] = p.s1_scale[tid - 32]; lds_aff[128 + tid - 32] = p.s1_shift[tid - 32]; }
    __syncthreads();
    auto read_aff8 = [&](const float* tab, float (&dst)[8]) {
        const f32x4 a = *(const f32x4*)tab, b = *(const f32x4*)(tab + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { dst[e] = a[e]; dst[4 + e] = b[e]; }
    };

#ifdef SD_TRACE
    unsigned long long tr_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    // ---- the regular-row machinery (SD_PRELOAD): tile k of this wave on a row with two new conv-0 rows is tile wave + 16 k of
    //      2 * (S / 16); a tile past the end repeats the last one (same values to the same slots)
    const int sd_tpr = S >> 4, sd_nt = 2 * sd_tpr;
    const int sd_lane_off = g < 3 ? ((g - 1) * S + q - 1) * 3 : (-S + q + 1) * 3 + 2;
    auto reg_tile = [&](int k, int ho_, int& xt, int& y, int& slot) {
        const int tile = min(wave + SD_WAVES * k, sd_nt - 1);
        const int rr = tile >= sd_tpr ? 1 : 0;
        xt = tile - rr * sd_tpr; y = 2 * ho_ + rr; slot = (2 * ho_ + 1 + rr) % 3;
    };
    // (the patch address as ONE 32-bit element offset from the batch's first image: a 64-bit pointer per load in flight costs two
    //  registers each, and a spilled address is reloaded through scratch, whose wait also waits for every patch still in flight)
    // `lane_lo`: sd_lane_off seen through an empty asm once per output row.  Every tile's column offset (xt * 48 + lane offset) and
    // strip address is the same on every row for a given wave, so the compiler would compute them once, keep ~10 registers for them
    // across the whole loop and -- at this kernel's 128-register cap -- spill them; their reloads then sit between the preloads and
    // wait for them (a scratch load shares the vector-memory counter).  Recomputing them costs one multiply-add per tile.
    int lane_lo = sd_lane_off;
    auto reg_load = [&](int n_, int y, int xt, float (&v)[8]) {
        const uint32_t off = (uint32_t)(((n_ * S + y) * S + xt * 16) * 3 + lane_lo);
        if (g < 3) img_buf_run8(rs_img, (const IMG*)nullptr, off, v);
        else {
            v[0] = img_buf_elem(rs_img, (const IMG*)nullptr, off);
            v[1] = img_buf_elem(rs_img, (const IMG*)nullptr, off + (uint32_t)(S * 3));
            v[2] = img_buf_elem(rs_img, (const IMG*)nullptr, off + (uint32_t)(2 * S * 3));
#pragma unroll
            for (int e = 3; e < 8; ++e) v[e] = 0.f;
        }
    };
    // a patch that touches the image border, element by element (stem_common.h: stem_gather<EDGE = true>), same addressing
    auto edge_load = [&](int n_, int y, int x, float (&v)[8]) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int ky, j;
            stem_k_slot(g, e, ky, j);
            const int kx = j / 3, ci = j - kx * 3;
            const int yy = y + ky - 1, xx = x + kx - 1;
            const bool ok = ky >= 0 && (unsigned)yy < (unsigned)S && (unsigned)xx < (unsigned)S;
            v[e] = ok ? img_buf_elem(rs_img, (const IMG*)nullptr, (uint32_t)(((n_ * S + yy) * S + xx) * 3 + ci)) : 0.f;
        }
    };
    auto reg_row = [&](int row_i) {                   // is output row `row_i` of this band regular?
        if (!SD_PRELOAD || row_i >= r_end || row_i == r_begin) return false;
        const int ho_ = row_i % Wo;
        return ho_ >= 1 && ho_ <= Wo - 2;
    };
    float vpre[SD_PRE][8];
    bool have_pre = false;
    for (int orow_i = r_begin; orow_i < r_end; ++orow_i) {
    const int n = orow_i / Wo, ho = orow_i - n * Wo;
    asm volatile("" : "+v"(lane_lo));
#ifdef SD_TRACE
    const bool tr_on = blockIdx.x == SD_TR_WG && orow_i - r_begin >= SD_TR_R0 && orow_i - r_begin < SD_TR_R0 + 4;
#endif
    SD_POINT(0);                          // row start
    // c0 rows 2ho-1+ry, ry = ry_first .. 2, are new; row 2ho-1 is the previous output row's 2(ho-1)+1, still in its slot
    const int ry_first = (orow_i == r_begin || ho == 0) ? 0 : 1;
    // ---- 2. c0 rows -> LDS ring (MFMA stem: stem_mfma_kernel's arithmetic, stem_common.h's K layout)
    {
        float sc0[8], sh0[8];
        read_aff8(lds_aff + g * 8, sc0);
        read_aff8(lds_aff + 32 + g * 8, sh0);
        const float* sc = sc0; const float* sh = sh0;
        const int tiles_per_row = S >> 4;
        const int ntiles = (3 - ry_first) * tiles_per_row;
        // (this lane's share of a pixel's patch, relative to the tile's first pixel -- stem_common.h: groups 0..2 read floats 0..7
        // of row y+g-1's run, group 3 float 8 of the three runs -- is sd_lane_off above)
        // a regular row's first tiles were loaded into `vpre` before the previous row's conv phase (have_pre was decided there)
        if (have_pre) {
            // a wave's tiles: wave + SD_WAVES k < 2 (S / 16); S / 32 <= 16 MFW * ... / SD_WM fragments per wave row bound them by 2 MFW
            constexpr int NB = 2 * MFW - SD_PRE;        // tiles behind the preloaded ones
            float vb[NB][8];
            int xtb[NB], yb[NB], slb[NB];
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                if (wave + SD_WAVES * (SD_PRE + k) < sd_nt) {         // wave-uniform
                    reg_tile(SD_PRE + k, ho, xtb[k], yb[k], slb[k]);
                    reg_load(n, yb[k], xtb[k], vb[k]);
                }
            }
            int xta[SD_PRE], ya[SD_PRE], sla[SD_PRE];
#pragma unroll
            for (int k = 0; k < SD_PRE; ++k) {
                reg_tile(k, ho, xta[k], ya[k], sla[k]);
            }
            auto finish = [&](float (&v)[8], int xt, int slot) {
                const int x = xt * 16 + q;
                if (xt == 0 || xt == sd_tpr - 1) {      // wave-uniform: the conv's zero padding left / right of the image
                    const bool lo = g < 3 ? x == 0 : x == S - 1, hi = x == S - 1;
#pragma unroll
                    for (int e = 0; e < 3; ++e) v[e] = lo ? 0.f : v[e];
                    v[6] = hi ? 0.f : v[6];
                    v[7] = hi ? 0.f : v[7];
                }
                u32x4 xf;
                E::store_chunk(&xf, v);
                f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
                Mma<DT>::run(a0, wf0, xf);
                Mma<DT>::run(a1, wf1, xf);
                float o[8];
                bn_act4<true, Y4_ACT_LEAKY>(a0, sc, sh, o);
                bn_act4<true, Y4_ACT_LEAKY>(a1, sc + 4, sh + 4, o + 4);
                u32x4 packed;
                E::store_chunk(&packed, o);
                const int plane = x & 1, row = (slot * 2 + plane) * PW + ((x + plane) >> 1);
                *(u32x4*)(lds_s + row * 64 + ((g ^ sd_swz(row)) * 16)) = packed;
            };
#pragma unroll
            for (int k = 0; k < SD_PRE; ++k) finish(vpre[k], xta[k], sla[k]);
#pragma unroll
            for (int k = 0; k < NB; ++k)
                if (wave + SD_WAVES * (SD_PRE + k) < sd_nt) finish(vb[k], xtb[k], slb[k]);
        } else
        for (int t0 = wave; t0 < ntiles; t0 += SD_WAVES * SD_GEN_UNROLL) {
            float v[SD_GEN_UNROLL][8];
            int xs[SD_GEN_UNROLL], rys[SD_GEN_UNROLL];          // rys: ring slot 0..2 of the c0 row | 4 if that row is outside the image
            int ys[SD_GEN_UNROLL], xts[SD_GEN_UNROLL];
            bool fast = true;                           // no tile of the batch is within 2 rows of the top / bottom border
#pragma unroll
            for (int u = 0; u < SD_GEN_UNROLL; ++u) {
                // wave-uniform (scalar registers).  Past the end: redo the last tile (same values to the same slots)
                const int tile = min(t0 + u * SD_WAVES, ntiles - 1);
                const int rr = (tile >= tiles_per_row) + (tile >= 2 * tiles_per_row);
                const int ry = ry_first + rr;
                const int xt = tile - rr * tiles_per_row, y = 2 * ho - 1 + ry;
                const int slot = (2 * ho + ry) % 3;              // ring slot of c0 row y = (y + 1) mod 3
                xs[u] = xt * 16 + q; ys[u] = y; xts[u] = xt;
                rys[u] = (unsigned)y < (unsigned)S ? slot : slot | 4;
                fast = fast && y >= 2 && y <= S - 3;      // rows y-1..y+1 inside, and the runs' overhang stays inside the image
            }
            // c0 value of (tile pixel, 8 channels) -> its slot of the strip.  column x: plane = x & 1; even plane slot
            // x/2, odd plane slot (x+1)/2 (slot 0 = column -1)
            auto strip_store = [&](int ry, int x, const u32x4& packed) {
                const int plane = x & 1, row = (ry * 2 + plane) * PW + ((x + plane) >> 1);
                *(u32x4*)(lds_s + row * 64 + ((g ^ sd_swz(row)) * 16)) = packed;
            };
            auto bn_act_pack = [&](const f32x4& a0, const f32x4& a1) {
                float o[8];
                bn_act4<true, Y4_ACT_LEAKY>(a0, sc, sh, o);              // packed BN + 0.1x, per-element max: same bits as the scalar form
                bn_act4<true, Y4_ACT_LEAKY>(a1, sc + 4, sh + 4, o + 4);
                u32x4 packed;
                E::store_chunk(&packed, o);
                return packed;
            };
            if (fast) {
                // branch-free batch: all loads back to back (one divergent if/else around them), then all MFMAs, then
                // the epilogues, so that neither the load nor the MFMA latency is paid per tile
#pragma unroll
                for (int u = 0; u < SD_GEN_UNROLL; ++u) reg_load(n, ys[u], xts[u], v[u]);
                // left / right image border: the run's pixel x-1 (floats 0..2) or x+1 (floats 6, 7 and group 3's values)
                // was read from the neighbouring row -- it is the conv's zero padding
#pragma unroll
                for (int u = 0; u < SD_GEN_UNROLL; ++u) {
                    if (xts[u] != 0 && xts[u] != tiles_per_row - 1) continue;        // wave-uniform
                    const bool lo = g < 3 ? xs[u] == 0 : xs[u] == S - 1, hi = xs[u] == S - 1;
#pragma unroll
                    for (int e = 0; e < 3; ++e) v[u][e] = lo ? 0.f : v[u][e];
                    v[u][6] = hi ? 0.f : v[u][6];
                    v[u][7] = hi ? 0.f : v[u][7];
                }
                f32x4 a0[SD_GEN_UNROLL], a1[SD_GEN_UNROLL];
#pragma unroll
                for (int u = 0; u < SD_GEN_UNROLL; ++u) {
                    u32x4 xf;
                    E::store_chunk(&xf, v[u]);
                    a0[u] = a1[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                    Mma<DT>::run(a0[u], wf0, xf);
                    Mma<DT>::run(a1[u], wf1, xf);
                }
#pragma unroll
                for (int u = 0; u < SD_GEN_UNROLL; ++u) {
                    strip_store(rys[u], xs[u], bn_act_pack(a0[u], a1[u]));
                }
            } else {
#pragma unroll
                for (int u = 0; u < SD_GEN_UNROLL; ++u) {
                    u32x4 packed = u32x4{0u, 0u, 0u, 0u};     // c0 rows outside the image are the conv's zero padding
                    if (!(rys[u] & 4)) {
                        edge_load(n, ys[u], xs[u], v[u]);
                        u32x4 xf;
                        E::store_chunk(&xf, v[u]);
                        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f};
                        Mma<DT>::run(a0, wf0, xf);
                        Mma<DT>::run(a1, wf1, xf);
                        packed = bn_act_pack(a0, a1);
                    }
                    strip_store(rys[u] & 3, xs[u], packed);
                }
            }
        }
    }
    SD_POINT(1);                          // stem tiles done (strip stores issued)
    if (orow_i == r_begin) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the c1 weights (staged once)
    __syncthreads();
    SD_POINT(2);                          // barrier passed

    // ---- L2 prefetch of the image rows the NEXT output row's stem will read for the first time (two rows, contiguous
    //      in memory; three at an image's first row): one LDS-DMA load per wave into a scratch KB -- no registers, nobody
    //      reads the data; the stem's patch loads, whose round trip is not overlapped with anything, then hit L2.
    if (orow_i + 1 < r_end) {
        const int n2 = (orow_i + 1) / Wo, ho2 = orow_i + 1 - n2 * Wo;
        const int64_t byte0 = (((int64_t)n2 * S + (ho2 == 0 ? 0 : 2 * ho2 + 1)) * S * 3) * (int64_t)sizeof(IMG);
        if (byte0 + 16 * 1024 < (int64_t)0x7fffffff)
#pragma unroll
            for (int k = 0; k < 16 / SD_WAVES; ++k)
                buffer_load16_lds(rs_img, lds_s + 3 * 2 * PW * 64, (int)byte0 + (wave + SD_WAVES * k) * 1024 + lane * 16, 0);
    }

    // ---- the next output row's first stem tiles into registers (their round trip hides under the conv phase below)
    have_pre = reg_row(orow_i + 1);
    if (have_pre) {
        const int ho2 = ho + 1;                           // (regular rows never start an image: same n)
#pragma unroll
        for (int k = 0; k < SD_PRE; ++k) {
            int xt, y, slot;
            reg_tile(k, ho2, xt, y, slot);
            reg_load(n, y, xt, vpre[k]);
        }
    }

    // ---- 3. stride-2 3x3 conv out of LDS.  Wave (wm, wn): pixel fragments wm, wm+8, ... x channel fragments
    //         2wn, 2wn+1; taps outer so that a tap's weight fragments are read once per wave.
    f32x4 acc[MFW][2];
#pragma unroll
    for (int f = 0; f < MFW; ++f) acc[f][0] = acc[f][1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int ky = t / 3, kx = t - ky * 3;
        u32x4 wf[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#if SD_W8
            wf[j] = wreg[t][j];
#else
            const int row = t * 64 + (wn * 2 + j) * 16 + q;
            wf[j] = *(const u32x4*)(lds_w + row * 64 + ((g ^ sd_swz(row)) * 16));
#endif
        }
        // input column 2wo + kx - 1: kx = 0 -> odd plane slot wo, kx = 1 -> even plane slot wo, kx = 2 -> odd plane slot wo+1
        const int row0 = (((2 * ho + ky) % 3) * 2 + (kx == 1 ? 0 : 1)) * PW + (kx == 2 ? 1 : 0) + q;     // ring slot of c0 row 2ho-1+ky
#pragma unroll
        for (int f = 0; f < MFW; ++f) {
            const int frag = wm + SD_WM * f;
            if (frag < MF) {                                  // wave-uniform
                const int row = row0 + frag * 16;
                const u32x4 xf = *(const u32x4*)(lds_s + row * 64 + ((g ^ sd_swz(row)) * 16));
                Mma<DT>::run(acc[f][0], wf[0], xf);
                Mma<DT>::run(acc[f][1], wf[1], xf);
            }
        }
#if SD_TAPBAR
        // the preloaded stem tiles stay in registers across this loop: keep the scheduler from pulling several taps' fragment reads
        // forward (it would spill -- and a spill of a register that a load is still filling waits for that load, here)
        if (t % SD_TAPBAR == SD_TAPBAR - 1) __builtin_amdgcn_sched_barrier(0);
#endif
    }

    SD_POINT(3);                          // conv 1 MFMAs issued
    // ---- 4. BN + activation; the lane holds channels g*16 + wn*8 + (0..7) of its pixel -> one 16-byte store
    float sc1[8], sh1[8];
    read_aff8(lds_aff + 64 + g * 16 + wn * 8, sc1);
    read_aff8(lds_aff + 128 + g * 16 + wn * 8, sh1);
    const float* sc = sc1; const float* sh = sh1;
    T* const orow = (T*)p.out + ((int64_t)(n * Wo + ho) * Wo) * p.out_cstride + p.out_coff + g * 16 + wn * 8;
#pragma unroll
    for (int f = 0; f < MFW; ++f) {
        const int frag = wm + SD_WM * f;
        if (frag < MF) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 2; ++j) bn_act4<true, Y4_ACT_LEAKY>(acc[f][j], sc + j * 4, sh + j * 4, v + j * 4);
            u32x4 pk;
            E::store_chunk(&pk, v);
            *(u32x4*)(orow + (int64_t)(frag * 16 + q) * p.out_cstride) = pk;
        }
    }
    SD_POINT(4);                          // epilogue done, stores issued
    __syncthreads();                  // every wave has read the ring: the next row's stem may overwrite two of its slots
    SD_POINT(5);                          // second barrier passed
#ifdef SD_TRACE
    if (tr_on && lane == 0) {
#pragma unroll
        for (int k = 0; k < 6; ++k) sd_trace_buf[((orow_i - r_begin - SD_TR_R0) * 16 + wave) * 8 + k] = tr_t[k];
    }
#endif
    }                                 // output rows of the band
}

#ifdef SD_TRACE
}  // namespace y4
extern "C" int y4_sd_trace_read(unsigned long long* dst) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(y4::sd_trace_buf), sizeof(unsigned long long) * 4 * 16 * 8);
}
namespace y4 {
#endif

// c1 weights + c0 ring + the prefetch's scratch KB
size_t stem_down_lds_bytes(int S) { return (size_t)SD_LDS_W + (size_t)3 * 2 * (S / 2 + 1) * 64 + 1024 + 1024; }

bool stem_down_supported(int dtype, int S) {
    return dtype != Y4_F32 && S % 32 == 0 && stem_down_lds_bytes(S) <= 160 * 1024 && (S / 32 + SD_WM - 1) / SD_WM <= (SD_W8 ? 5 : 3);
}

template <int DT, class IMG>
static int stem_down_dispatch(const StemDownK& k, hipStream_t stream) {
    const int mfw = (k.S / 32 + SD_WM - 1) / SD_WM;         // ceil((Wo/16) / SD_WM)
    const size_t lds = stem_down_lds_bytes(k.S);
    // persistent: one workgroup per CU (LDS), each a contiguous band of the N * S/2 output rows
    static int n_cus[64] = {0};
    int dev = 0;
    Y4_CHECK_HIP(hipGetDevice(&dev));
    if (n_cus[dev & 63] == 0) {
        int v = 0;
        Y4_CHECK_HIP(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
        n_cus[dev & 63] = v > 0 ? v : 256;
    }
    const int rows = k.N * (k.S / 2);
    const int blocks = rows < n_cus[dev & 63] ? rows : n_cus[dev & 63];
#define Y4_SD_CASE(M)                                                                                        \
    case M: {                                                                                                \
        static PerDeviceOnce once;                                                                           \
        if (const uint64_t bit = once.due()) {                                                               \
            Y4_CHECK_HIP(hipFuncSetAttribute((const void*)stem_down_kernel<DT, M, IMG>,                     \
                                             hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));       \
            once.mark(bit);                                                                                  \
        }                                                                                                    \
        hipLaunchKernelGGL((stem_down_kernel<DT, M, IMG>), dim3(blocks), dim3(64 * SD_WAVES), lds, stream, k);         \
        break;                                                                                               \
    }
    switch (mfw) {
        Y4_SD_CASE(1) Y4_SD_CASE(2) Y4_SD_CASE(3)
#if SD_W8
        Y4_SD_CASE(4) Y4_SD_CASE(5)
#endif
        default: set_error("stem_down: image side %d not supported", k.S); return Y4_EINVAL;
    }
    Y4_CHECK_HIP(hipGetLastError());
    return Y4_OK;
}

int stem_down_launch(int dtype, const void* imgs, int img_u8, int n, int S, const void* stem_wk, const float* s0_scale,
                     const float* s0_shift, int act0, const void* w1_packed, const float* s1_scale, const float* s1_shift,
                     int act1, void* out, int out_cstride, int out_coff, hipStream_t stream) {
    Y4_REQUIRE(stem_down_supported(dtype, S), Y4_EINVAL, "stem_down: dtype %d / image side %d not supported", dtype, S);
    Y4_REQUIRE(imgs && stem_wk && w1_packed && out, Y4_EINVAL, "stem_down: null pointer");
    Y4_REQUIRE(act0 == Y4_ACT_LEAKY && act1 == Y4_ACT_LEAKY, Y4_EINVAL, "stem_down: both convs are LeakyReLU in the plan (got %d, %d)", act0, act1);
    // (the kernel addresses the image batch through one buffer descriptor with 32-bit byte offsets)
    Y4_REQUIRE((int64_t)n * S * S * 3 * (img_u8 ? 1 : 4) < (1ll << 32), Y4_EINVAL, "stem_down: image batch too large (>= 4 GiB)");
    Y4_REQUIRE(out_cstride % 8 == 0 && out_coff % 8 == 0, Y4_EINVAL, "stem_down: output view not 16-byte aligned");
    StemDownK k{};
    k.img = imgs;
    k.stem_frag = (const u32x4*)((const char*)stem_wk + (dtype == Y4_BF16 ? 4096 : 6144));
    k.s0_scale = s0_scale; k.s0_shift = s0_shift; k.act0 = act0;
    k.w1 = (const char*)w1_packed; k.s1_scale = s1_scale; k.s1_shift = s1_shift; k.act1 = act1;
    k.w1_bytes = (unsigned)(round_up(64, COUT_PAD) * 9 * 32 * 2);
    k.out = (char*)out; k.out_cstride = out_cstride; k.out_coff = out_coff;
    k.N = n; k.S = S;
    if (img_u8) return dtype == Y4_BF16 ? stem_down_dispatch<Y4_BF16, uint8_t>(k, stream) : stem_down_dispatch<Y4_F16, uint8_t>(k, stream);
    return dtype == Y4_BF16 ? stem_down_dispatch<Y4_BF16, float>(k, stream) : stem_down_dispatch<Y4_F16, float>(k, stream);
}

}  // namespace y4
