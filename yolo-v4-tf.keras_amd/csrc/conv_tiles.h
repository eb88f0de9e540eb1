// conv_tiles.h -- the tile configurations of conv_igemm_kernel (shared by the per-dtype kernel translation units and the
// launcher in conv_igemm.hip).
#pragma once

namespace y4 {

struct TileCfg {
    int bm, bn, wm, wn, bkb, nst;
};
// id, BM (pixels), BN (channels), WM, WN (wave grid), BKB (bytes of K per LDS row), NST (ring stages, 2..7; 12 = the staggered
// 2-stage schedule, 32 = 2 stages with the 32x32x16 MFMA; conv_p8_kernel.h: 8 = staggered wave groups, 9 = software-pipelined;
// 10 = the producer / consumer kernel of conv_l12_kernel.h; 20 = a HALO tile of conv_halo_kernel.h: 3x3 stride-1 convs, BM pixels = a
// band of full image rows, the input halo tile staged once per 64-channel chunk; 21 = a HALO2 tile of conv_halo2_kernel.h: the same band
// geometry with ONE wave per SIMD, v_mfma_f32_32x32x16 and the weights read straight into registers in MFMA-fragment order).
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
    X(32, 224, 256, 2, 4, 128, 12) \
    X(33, 192, 256, 2, 4, 128, 32) \
    X(34, 256, 256, 2, 4, 128, 32) \
    X(35, 128, 256, 2, 4, 128, 32) \
    X(36, 128, 128, 2, 2, 128, 32) \
    X(37, 256, 128, 4, 2, 128, 32) \
    X(38, 384, 128, 4, 2, 128, 2) \
    X(39, 192, 256, 2, 4, 128, 8) \
    X(40, 256, 256, 2, 4, 128, 8) \
    X(41, 192, 256, 2, 4, 128, 9) \
    X(42, 192, 256, 2, 4, 128, 10) \
    X(43, 64, 128, 1, 4, 128, 6)  \
    X(44, 32, 128, 1, 4, 128, 6)  \
    X(45, 128, 128, 2, 2, 128, 4) \
    X(46, 64, 64, 2, 2, 128, 6)   \
    X(47, 128, 64, 4, 1, 128, 4)  \
    X(48, 32, 64, 2, 2, 128, 7)   \
    X(49, 32, 64, 2, 2, 128, 5)   \
    X(50, 64, 64, 2, 2, 128, 4)   \
    X(51, 384, 128, 4, 2, 128, 20) \
    X(52, 192, 256, 2, 4, 128, 20) \
    X(53, 192, 128, 2, 4, 128, 20) \
    X(54, 320, 128, 4, 2, 128, 20) \
    X(55, 384, 128, 2, 2, 128, 21) \
    X(56, 192, 256, 1, 4, 128, 21) \
    X(57, 192, 128, 2, 2, 128, 21) \
    X(58, 384, 64, 2, 2, 64, 21) \
    X(59, 192, 128, 1, 4, 128, 21) \
    X(60, 384, 64, 2, 2, 128, 21) \
    X(61, 384, 128, 2, 2, 64, 21) \
    X(62, 192, 128, 2, 2, 64, 21)

#define Y4_TILE_ROW(id, bm, bn, wm, wn, bkb, nst) {bm, bn, wm, wn, bkb, nst},
static const TileCfg kTiles[] = {Y4_TILES(Y4_TILE_ROW)};
constexpr int kNumTiles = sizeof(kTiles) / sizeof(kTiles[0]);

constexpr int F32_TILES = 12;
// ... and the deep rings (round 4: what a single image's latency-bound K loops need)
constexpr int DEEP_TILE0 = 43;
constexpr int DEEP_TILE1 = 50;
inline constexpr bool f32_tile(int id) { return id <= F32_TILES || (id >= DEEP_TILE0 && id <= DEEP_TILE1); }

// the tiles that sum in the 32x32x16 MFMA's order (bit-identical among themselves, not with the others)
inline bool mfma32_tile(int tile) { return tile >= 1 && tile <= kNumTiles && (kTiles[tile - 1].nst == 32 || kTiles[tile - 1].nst == 21); }   // (21: the halo2 tiles, same order)

// tiles the autotuner does not offer: the 32x32x16 ones (another summation order) and the producer / consumer kernel (measured
// 30-90 % slower than every other form on every layer: kept selectable by id as the record of that experiment)
inline bool tuner_skips_tile(int tile) { return mfma32_tile(tile) || (tile >= 1 && tile <= kNumTiles && kTiles[tile - 1].nst == 10); }

// Split-K (small batches: fewer tiles than compute units and a long serial K loop): tile id = base + 100 e runs base tile `base`
// with its K loop split 2^e ways over 2^e workgroups per output tile.  A split run adds its partial sums in split order: a fixed
// fp32 summation order, but not the unsplit loop's -- like the 32x32x16 tiles these ids are never offered by the bit-identical
// tuner, only by y4_autotune after y4_set_splitk(h, 1).  Base tiles: the plain ring schedules (2..7 stages).
constexpr int SPLITK_MAX_E = 3;
constexpr int SPLITK_CNT_BYTES = 16 * 1024;          // 4096 tile counters in front of the partial sums
constexpr size_t SPLITK_WS_BYTES = SPLITK_CNT_BYTES + (size_t)32 * 1024 * 1024;      // what an engine's workspace reserves for it
inline bool splitk_tile(int base) { return base >= 1 && base <= kNumTiles && kTiles[base - 1].nst >= 2 && kTiles[base - 1].nst <= 7; }
inline int tile_base(int tile) { return tile >= 100 ? tile % 100 : tile; }
inline int tile_split_e(int tile) { return tile >= 100 ? tile / 100 : 0; }
// a well-formed tile id (0 = heuristic)
inline bool tile_id_ok(int tile) {
    return tile >= 0 && tile_base(tile) <= kNumTiles && tile_split_e(tile) <= SPLITK_MAX_E && (tile < 100 || splitk_tile(tile_base(tile)));
}

// Halo tiles (conv_halo_kernel.h): the band geometry of a BM-pixel tile on an H x W feature map -- rows per band (balanced over the
// image), bands per image, LDS pitch of a halo row -- and the LDS it needs with BN output channels per tile.
inline bool halo_tile(int tile) { return tile >= 1 && tile <= kNumTiles && kTiles[tile - 1].nst == 20; }
struct HaloPlan { int rows, bands, pitch; };
inline size_t halo_lds_bytes(int rows, int pitch, int bn) { return (size_t)2 * (rows + 2) * pitch * 128 + (size_t)2 * bn * 128 + 256; }
inline bool halo_plan(int bm, int bn, int H, int W, HaloPlan* out) {
    if (W < 1 || H < 1 || W > bm) return false;
    int rows = bm / W;
    if (rows > H) rows = H;
    const int bands = (H + rows - 1) / rows;
    rows = (H + bands - 1) / bands;                       // balanced: the last band is at most one row short per band
    const int pitch = (W + 2 + 7) / 8 * 8;
    if (halo_lds_bytes(rows, pitch, bn) > 160 * 1024) return false;
    if ((int64_t)H * W * 4 < (int64_t)bands * bm * 3) return false;      // less than 3/4 of the MFMA tiles' rows would be pixels
    *out = HaloPlan{rows, bands, pitch};
    return true;
}

// Halo2 tiles (conv_halo2_kernel.h): id, BM, BN, WM, WN, KC (channels per staged sub-chunk: LDS rows of 2 KC bytes), OCC (workgroups
// per CU the kernel is built for: 2 = at most 256 registers and half the LDS).  The same bands as the halo tiles, halo rows at pitch
// W + 256 / (2 KC) exactly (the y-dependent swizzle that keeps the nine shifted fragment reads free of LDS bank conflicts wants that),
// two halo buffers of a whole number of 16-row groups, one dummy piece, the touch scratch.  Each of the four waves stages
// ceil(pieces / 4) <= halo2_pmax one-KB pieces per sub-chunk.
#define Y4_HALO2_TILES(X)          \
    X(55, 384, 128, 2, 2, 64, 1)   \
    X(56, 192, 256, 1, 4, 64, 1)   \
    X(57, 192, 128, 2, 2, 64, 1)   \
    X(58, 384, 64, 2, 2, 32, 2)    \
    X(59, 192, 128, 1, 4, 64, 2)   \
    X(60, 384, 64, 2, 2, 64, 1)    \
    X(61, 384, 128, 2, 2, 32, 1)   \
    X(62, 192, 128, 2, 2, 32, 1)
struct Halo2Cfg { int id, kc, occ; };
#define Y4_H2_ROW(id, bm, bn, wm, wn, kc, occ) {id, kc, occ},
static const Halo2Cfg kHalo2[] = {Y4_HALO2_TILES(Y4_H2_ROW)};
inline bool halo2_tile(int tile) { return tile >= 1 && tile <= kNumTiles && kTiles[tile - 1].nst == 21; }
inline const Halo2Cfg* halo2_cfg(int tile) {
    for (const Halo2Cfg& c : kHalo2) if (c.id == tile) return &c;
    return nullptr;
}
inline constexpr __host__ __device__ int halo2_pmax(int kc, int occ) { return (kc == 32 || occ == 2) ? 10 : 20; }
inline __host__ __device__ int halo2_rows_alloc(int rows, int pitch) { return ((rows + 2) * pitch + 15) / 16 * 16; }
inline size_t halo2_lds_bytes(int rows, int pitch, int kc) { return (size_t)2 * halo2_rows_alloc(rows, pitch) * 2 * kc + 1024 + 256; }
inline bool halo2_plan(int tile, int H, int W, HaloPlan* out) {
    const Halo2Cfg* hc = halo2_cfg(tile);
    if (!hc || W < 1 || H < 1) return false;
    const int bm = kTiles[tile - 1].bm;
    if (W > bm) return false;
    int rows = bm / W;
    if (rows > H) rows = H;
    const int bands = (H + rows - 1) / rows;
    rows = (H + bands - 1) / bands;
    const int pitch = W + 128 / hc->kc;
    if (halo2_lds_bytes(rows, pitch, hc->kc) > (size_t)(160 * 1024 / hc->occ)) return false;
    if (((size_t)halo2_rows_alloc(rows, pitch) * 2 * hc->kc / 1024 + 3) / 4 > (size_t)halo2_pmax(hc->kc, hc->occ)) return false;
    if ((int64_t)H * W * 4 < (int64_t)bands * bm * 3) return false;
    *out = HaloPlan{rows, bands, pitch};
    return true;
}

// chain heads: the tiles with one wave column over 64 channels
inline bool chain_tile(int tile) { return tile == 3 || tile == 4 || tile == 15; }

// LDS-pair heads: 128-byte K rows, 2 stages, one channel tile over all of Cout (128 or 256), tile + tail stages in LDS
// (51, 52, 54: the halo tiles of conv_halo_kernel.h -- 384 x 128 and 320 x 128 for Cout = 128, 192 x 256 for Cout = 256)
inline bool pair_tile(int tile) {
    switch (tile) { case 1: case 8: case 20: case 24: case 25: case 29: case 13: case 19: case 21: case 22: case 38: case 51: case 52: case 54: return true; }
    return false;
}

}  // namespace y4
