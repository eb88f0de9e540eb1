// conv_igemm.hip -- launch side of conv_igemm_kernel (conv_igemm_kernel.h): argument checks, tile choice, dispatch to the
// per-dtype translation units, and the weight re-layout kernels.
#include "conv_chain.h"
#include "conv_tiles.h"

namespace y4 {

int conv_launch_f32(int tile, const ConvK& k, hipStream_t s);
int conv_launch_bf16(int tile, const ConvK& k, hipStream_t s);
int conv_launch_f16(int tile, const ConvK& k, hipStream_t s);
int conv_launch_bf16_fused(int tile, const ConvK& k, hipStream_t s);
int conv_launch_f16_fused(int tile, const ConvK& k, hipStream_t s);

int conv_p8_launch_bf16(int bm, int nst, const ConvK& k, hipStream_t s);
int conv_p8_launch_f16(int bm, int nst, const ConvK& k, hipStream_t s);
int conv_p8_launch(int dtype, int bm, int nst, const ConvK& k, hipStream_t s) {
    if (dtype == Y4_BF16) return conv_p8_launch_bf16(bm, nst, k, s);
    if (dtype == Y4_F16) return conv_p8_launch_f16(bm, nst, k, s);
    set_error("conv2d: the phased kernel is 16-bit only");
    return Y4_EINVAL;
}

int conv_halo_launch_bf16(int bm, int bn, const ConvK& k, hipStream_t s);
int conv_halo_launch_f16(int bm, int bn, const ConvK& k, hipStream_t s);
int conv_halo_launch(int dtype, int bm, int bn, const ConvK& k, hipStream_t s) {
    if (dtype == Y4_BF16) return conv_halo_launch_bf16(bm, bn, k, s);
    if (dtype == Y4_F16) return conv_halo_launch_f16(bm, bn, k, s);
    set_error("conv2d: halo tiles are 16-bit only");
    return Y4_EINVAL;
}

int conv_halo2_launch_bf16(int tile, const ConvK& k, hipStream_t s);
int conv_halo2_launch_f16(int tile, const ConvK& k, hipStream_t s);
int conv_halo2_launch(int dtype, int tile, const ConvK& k, hipStream_t s) {
    if (dtype == Y4_BF16) return conv_halo2_launch_bf16(tile, k, s);
    if (dtype == Y4_F16) return conv_halo2_launch_f16(tile, k, s);
    set_error("conv2d: halo2 tiles are 16-bit only");
    return Y4_EINVAL;
}

int conv_tile_count() { return kNumTiles; }

// the weight touch of conv_common.h is on unless Y4_NO_WEIGHT_TOUCH=1 (A/B measurements only: results are the same)
bool weight_touch_enabled() {
    static const bool on = [] { const char* e = getenv("Y4_NO_WEIGHT_TOUCH"); return !(e && e[0] == '1'); }();
    return on;
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
                  const ConvPairDesc* pair, const ConvObjDesc* obj) {
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
    Y4_REQUIRE(d->cin % k_chunk_channels(d->cin, d->ksize) == 0, Y4_EINVAL,
               "conv2d: cin %d is not a whole number of %d-channel K chunks (the canonical K order of a 3x3 conv, common.h)", d->cin,
               k_chunk_channels(d->cin, d->ksize));
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
    // (the implicit-GEMM kernels address the input through a descriptor based one row + one pixel BEFORE it, with row offsets biased by
    //  the same amount and taps up to two rows + two pixels further: all of that must stay below 2^31 too, or an out-of-range padding
    //  offset would wrap into range -- ADVICE r5)
    const int64_t tap_span = d->ksize == 3 ? ((int64_t)3 * d->w + 3) * d->in_cstride * es + (int64_t)d->cin * es : 0;
    Y4_REQUIRE(in_bytes + tap_span < (1ll << 31) && wt_bytes < (1ll << 31), Y4_EINVAL,
               "conv2d: input (%lld B + %lld B of tap reach) or weights (%lld B) exceed the 2 GiB buffer-descriptor range", (long long)in_bytes,
               (long long)tap_span, (long long)wt_bytes);
    k.in_bytes = (unsigned)in_bytes; k.wt_bytes = (unsigned)wt_bytes;
    {   // extents of the views the fast epilogue addresses through buffer descriptors (conv_common.h: conv_epilogue_fast)
        const int64_t ob = (int64_t)k.M * d->out_cstride * es, ob2 = d->out2 ? (int64_t)k.M * d->out2_cstride * es : 0,
                      rb = d->res ? (int64_t)k.M * d->res_cstride * es : 0;
        k.fast_epi = !d->upsample && !d->out_f32 && ob < (1ll << 31) && ob2 < (1ll << 31) && rb < (1ll << 31);
        if (k.fast_epi) { k.out_bytes = (unsigned)ob; k.out2_bytes = (unsigned)ob2; k.res_bytes = (unsigned)rb; }
    }
    k.div_howo = fastdiv_make((uint32_t)(k.Ho * k.Wo)); k.div_wo = fastdiv_make((uint32_t)k.Wo);
    k.in_cstride = d->in_cstride; k.in_coff = d->in_coff;
    k.out_cstride = d->out_cstride; k.out_coff = d->out_coff;
    k.res_cstride = d->res_cstride; k.res_coff = d->res_coff;
    k.ksize = d->ksize; k.stride = d->stride; k.pad = d->ksize == 3 ? 1 : 0;
    k.act = d->act; k.upsample = d->upsample; k.out_f32 = d->out_f32;
    if (obj) {
        Y4_REQUIRE(obj->obj && !d->upsample && ((pair && pair->out_f32) || (!pair && d->out_f32)), Y4_EINVAL,
                   "conv2d: an objectness side array needs a float32 head");
        k.obj = obj->obj; k.obj_nf = obj->nf; k.obj_cpi = obj->cells_per_img; k.obj_base = obj->cell_base;
    }
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
        const int64_t fb = (int64_t)k.M * pair->fin_cstride * es, fb2 = pair->fin2 ? (int64_t)k.M * pair->fin2_cstride * es : 0;
        k.fast_tail = !pair->out_f32 && fb < (1ll << 31) && fb2 < (1ll << 31);
        if (k.fast_tail) { k.fin_bytes = (unsigned)fb; k.fin2_bytes = (unsigned)fb2; }
    }
    if (chain && chain->ntail > 0) {
        Y4_REQUIRE(d->dtype != Y4_F32 && d->cout == 64 && d->act == Y4_ACT_MISH && !d->upsample && !d->out_f32 && !d->out2 &&
                       chain->ntail <= 2 && chain->fin && chain->fin_cstride % epc == 0 && chain->fin_coff % epc == 0,
                   Y4_EINVAL, "conv2d: bad chain description");
        k.ntail = chain->ntail; k.store_x = chain->store_x;
        k.fin = (char*)chain->fin; k.fin_cstride = chain->fin_cstride; k.fin_coff = chain->fin_coff;
        Y4_REQUIRE(!chain->concat_only || (chain->ntail == 2 && chain->tail[1].cout == 128 && !chain->store_x), Y4_EINVAL,
                   "conv2d: a head chained straight to the conv over the concat needs ntail = 2, 128 output channels, no own store");
        for (int t = chain->concat_only ? 1 : 0; t < chain->ntail; ++t) {
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
    // split-K: tile id = base + 100 e runs the base tile with the K loop split 2^e ways (e = 1..3; kernels.h: splitk_*)
    const int split_e = tile >= 100 ? tile / 100 : 0;
    if (split_e) tile %= 100;
    Y4_REQUIRE(split_e <= SPLITK_MAX_E, Y4_EINVAL, "conv2d: tile id %d: at most %d-way split-K", d->tile, 1 << SPLITK_MAX_E);
    Y4_REQUIRE(k.ntail == 0 || chain_tile(tile), Y4_EINVAL, "conv2d: tile id %d cannot head this chain", tile);
    Y4_REQUIRE(!pair || (pair_tile(tile) && kTiles[tile - 1].bn == d->cout), Y4_EINVAL, "conv2d: tile id %d cannot head this LDS pair", tile);
    Y4_REQUIRE(tile >= 1 && tile <= kNumTiles, Y4_EINVAL, "conv2d: tile id %d out of range", tile);
    k.touch = weight_touch_enabled() && !split_e ? 1 : 0;     // (a split walks only its share of the weight block)
    const TileCfg& tc = kTiles[tile - 1];
    Y4_REQUIRE(tile_ok(tc, d->dtype, d->cin, cout_pad), Y4_EINVAL,
               "conv2d: tile %d (bk bytes %d, bn %d) does not fit cin %d / cout_pad %d", tile, tc.bkb, tc.bn,
               d->cin, cout_pad);
    k.grid_m = (k.M + tc.bm - 1) / tc.bm;
    k.grid_n = (int)((round_up(d->cout, 8) + tc.bn - 1) / tc.bn);
    if (tc.nst == 20) {
        // halo tiles (conv_halo_kernel.h): 3x3 stride-1 convs whose band geometry fits the LDS; plain launches with 16-byte stores
        HaloPlan hp{};
        Y4_REQUIRE(d->dtype != Y4_F32 && d->ksize == 3 && d->stride == 1 && d->cin % 64 == 0 && k.ntail == 0 && !split_e && !d->upsample &&
                       !d->out_f32 && !d->out2 && halo_plan(tc.bm, tc.bn, d->h, d->w, &hp),
                   Y4_EINVAL, "conv2d: tile %d (halo, %d x %d) does not fit this conv (3x3 stride 1, 16-bit, cin %% 64 == 0, %d x %d map)", tile,
                   tc.bm, tc.bn, d->h, d->w);
        k.h_rows = hp.rows; k.h_bands = hp.bands; k.h_pitch = hp.pitch;
        k.h_div_pitch = fastdiv_make((uint32_t)hp.pitch); k.h_div_bands = fastdiv_make((uint32_t)hp.bands);
        k.grid_m = d->n * hp.bands;
        k.h_xmap = wt_bytes > ((int64_t)3 << 20) ? 1 : 0;
        { static const int abl = [] { const char* e = getenv("HALO_ABL"); return e ? atoi(e) : 0; }(); k.h_abl = abl; }      // (read only by a -DHALO_ABLATIONS=1 build of the kernel)
    }
    if (tc.nst == 21) {
        // halo2 tiles (conv_halo2_kernel.h): plain launches of 3x3 stride-1 convs, 16-byte stores, weights also in fragment order
        HaloPlan hp{};
        Y4_REQUIRE(d->dtype != Y4_F32 && d->ksize == 3 && d->stride == 1 && d->cin % 64 == 0 && k.ntail == 0 && !pair && !split_e && !d->upsample &&
                       !d->out_f32 && !d->out2 && halo2_plan(tile, d->h, d->w, &hp),
                   Y4_EINVAL, "conv2d: tile %d (halo2, %d x %d) does not fit this conv (3x3 stride 1, 16-bit, cin %% 64 == 0, %d x %d map)", tile,
                   tc.bm, tc.bn, d->h, d->w);
        Y4_REQUIRE(d->wt_frag, Y4_EINVAL, "conv2d: tile %d (halo2) needs the fragment-ordered weights (y4_conv_desc.wt_frag, y4_pack_conv_frag32)", tile);
        k.wfrag = (const char*)d->wt_frag; k.wfrag_bytes = (unsigned)wt_bytes;
        k.h_rows = hp.rows; k.h_bands = hp.bands; k.h_pitch = hp.pitch;
        k.h_div_pitch = fastdiv_make((uint32_t)hp.pitch); k.h_div_bands = fastdiv_make((uint32_t)hp.bands);
        k.grid_m = d->n * hp.bands;
        // (the halo kernel's XCD <-> channel tile map for weights larger than an L2 is OFF here: these kernels look two taps ahead for their
        //  weights, and measured L2-cold -- as in a step -- the map costs 512 -> 1024 @19^2 88 -> 106 us and 256 -> 512 @38^2 95 -> 100;
        //  Y4_H2_XMAP=1 turns it on for experiments)
        { static const int xm = [] { const char* e = getenv("Y4_H2_XMAP"); return e ? atoi(e) : 0; }(); k.h_xmap = xm; }
        { static const int abl = [] { const char* e = getenv("HALO_ABL"); return e ? atoi(e) : 0; }(); k.h_abl = abl; }      // (read only by a -DHALO2_ABLATIONS=1 build)
    }
    Y4_REQUIRE((int64_t)k.grid_n * tc.bn <= cout_pad, Y4_EINVAL, "conv2d: tile %d overruns the packed weight rows", tile);
    k.div_gridn = fastdiv_make((uint32_t)k.grid_n);
    if (split_e) {
        const int S = 1 << split_e;
        const int64_t nwg = (int64_t)k.grid_m * k.grid_n, need = SPLITK_CNT_BYTES + nwg * S * tc.bm * tc.bn * 4;
        Y4_REQUIRE(!pair && !(chain && chain->ntail > 0) && splitk_tile(tile), Y4_EINVAL,
                   "conv2d: tile id %d cannot run split-K (plain launches of the 2..7-stage ring tiles only)", d->tile);
        Y4_REQUIRE(k.K / (tc.bkb / es) >= 2 * S, Y4_EINVAL, "conv2d: tile id %d: %d K-tiles are too few for a %d-way split", d->tile,
                   k.K / (tc.bkb / es), S);
        Y4_REQUIRE(d->splitk_ws && nwg <= SPLITK_CNT_BYTES / 4 && (int64_t)d->splitk_ws_bytes >= need, Y4_EINVAL,
                   "conv2d: tile id %d needs a split-K workspace of %lld bytes for %lld tiles (got %lld; at most %d tiles)", d->tile,
                   (long long)need, (long long)nwg, (long long)d->splitk_ws_bytes, SPLITK_CNT_BYTES / 4);
        k.ksplit = S;
        k.split_cnt = (int*)d->splitk_ws;
        k.part = (float*)((char*)d->splitk_ws + SPLITK_CNT_BYTES);
    }
    k.out2 = (char*)d->out2; k.out2_cstride = d->out2_cstride; k.out2_coff = d->out2_coff; k.split = d->out2 ? d->split : 0;
    Y4_REQUIRE(!d->out2 || (d->split > 0 && d->split % 32 == 0 && d->split < d->cout && !d->res && !d->upsample && !d->out_f32 &&
                            d->out2_cstride % epc == 0 && d->out2_coff % epc == 0),
               Y4_EINVAL, "conv2d: bad split-output description");
    const bool fused = k.pair || k.ntail > 0;
    if (tc.nst == 21) return conv_halo2_launch(d->dtype, tile, k, stream);
    switch (d->dtype) {
        case Y4_F32:
            Y4_REQUIRE(!fused, Y4_EINVAL, "conv2d: chains and LDS pairs are 16-bit only");
            return conv_launch_f32(tile, k, stream);
        case Y4_BF16: return fused ? conv_launch_bf16_fused(tile, k, stream) : conv_launch_bf16(tile, k, stream);
        default: return fused ? conv_launch_f16_fused(tile, k, stream) : conv_launch_f16(tile, k, stream);
    }
}

// ------------------------------------------------------------------------------ weight re-layout
// Darknet (cout, cin, k, k) float32 -> [cout_pad][cin / KC][k*k][KC] in the storage type: the canonical K order of common.h
template <int DT>
__global__ void pack_conv_kernel(const float* __restrict__ w, typename Elem<DT>::type* __restrict__ out, int cout,
                                 int cout_pad, int cin, int kk, int kc) {
    const int64_t total = (int64_t)cout_pad * kk * cin;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % kc);
        int64_t r = i / kc;
        const int tap = (int)(r % kk);
        r /= kk;
        const int nchunk = cin / kc;
        const int chunk = (int)(r % nchunk), co = (int)(r / nchunk);
        const int ci = chunk * kc + c;
        const float v = co < cout ? w[((int64_t)co * cin + ci) * kk + tap] : 0.f;
        out[i] = Elem<DT>::st(v);
    }
}

int pack_conv_weights(int dtype, int cout, int cin, int ksize, const float* oihw, void* packed, hipStream_t stream) {
    const int cout_pad = (int)round_up(cout, COUT_PAD), kk = ksize * ksize;
    const int kc = k_chunk_channels(cin, ksize);
    Y4_REQUIRE(cin % kc == 0, Y4_EINVAL, "pack_conv_weights: cin %d is not a multiple of the K chunk %d", cin, kc);
    const int64_t total = (int64_t)cout_pad * kk * cin;
    const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    switch (dtype) {
        case Y4_F32: hipLaunchKernelGGL(pack_conv_kernel<Y4_F32>, dim3(blocks), dim3(256), 0, stream, oihw, (float*)packed, cout, cout_pad, cin, kk, kc); break;
        case Y4_BF16: hipLaunchKernelGGL(pack_conv_kernel<Y4_BF16>, dim3(blocks), dim3(256), 0, stream, oihw, (uint16_t*)packed, cout, cout_pad, cin, kk, kc); break;
        case Y4_F16: hipLaunchKernelGGL(pack_conv_kernel<Y4_F16>, dim3(blocks), dim3(256), 0, stream, oihw, (_Float16*)packed, cout, cout_pad, cin, kk, kc); break;
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

// packed 3x3 weights [cout_pad][cin/64][9][64] -> MFMA-fragment order of the halo2 tiles (conv_halo2_kernel.h):
// out[((((blk * nch + c) * 9 + tap) * 4 + s) * 64 + lane) * 8 + e] = packed[ch(blk, lane & 31)][c][tap][16 s + 8 (lane >> 5) + e], with
// MFMA row R = 8 g + 4 h + j of block blk <-> channel 32 blk + 16 (g >> 1) + 8 h + 4 (g & 1) + j (accumulator value 4 g + j of lane half h is
// element 4 (g & 1) + j of 8-channel chunk 2 (g >> 1) + h: chunk_channel_g<2>, what the shared epilogue stores 16 bytes at a time)
__global__ void pack_frag32_kernel(const u32x4_t* __restrict__ packed, u32x4_t* __restrict__ out, int cout_pad, int nch) {
    const int64_t total = (int64_t)cout_pad * nch * 9 * 8;           // 16-byte groups
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63);
        int64_t r = i >> 6;
        const int s = (int)(r & 3);
        r >>= 2;
        const int tap = (int)(r % 9);
        r /= 9;
        const int c = (int)(r % nch), blk = (int)(r / nch);
        const int R = lane & 31, hk = lane >> 5, g = R >> 3, h = (R >> 2) & 1, j = R & 3;
        const int ch = 32 * blk + 16 * (g >> 1) + 8 * h + 4 * (g & 1) + j;
        out[i] = packed[(((int64_t)ch * nch + c) * 9 + tap) * 8 + 2 * s + hk];
    }
}

int pack_conv_frag32(int dtype, int cout, int cin, const void* packed, void* frag, hipStream_t stream) {
    Y4_REQUIRE((dtype == Y4_BF16 || dtype == Y4_F16) && cout > 0 && cin > 0 && cin % 64 == 0, Y4_EINVAL,
               "pack_conv_frag32: 16-bit 3x3 weights with cin %% 64 == 0 (dtype %d, cin %d)", dtype, cin);
    const int cout_pad = (int)round_up(cout, COUT_PAD), nch = cin / 64;
    const int64_t total = (int64_t)cout_pad * nch * 9 * 8;
    const int blocks = (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256);
    hipLaunchKernelGGL(pack_frag32_kernel, dim3(blocks), dim3(256), 0, stream, (const u32x4_t*)packed, (u32x4_t*)frag, cout_pad, nch);
    Y4_CHECK_HIP(hipGetLastError());
    return Y4_OK;
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
