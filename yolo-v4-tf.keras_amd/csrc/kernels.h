// kernels.h -- launch-side interface between runtime.hip and the kernel files.
#pragma once
#include "common.h"

namespace y4 {

// conv_igemm.hip
// 1x1 convs chained onto a conv's register tile (conv_chain.h); d->out is then the head conv's own output view,
// written only if store_x, and the last tail writes `fin`.  16-bit dtypes, head cout == 64, Mish everywhere.
struct ConvChainDesc {
    int ntail, store_x;            // (concat_only: ntail = 2 with tail[0] unused -- the head feeds the conv over the concat directly)
    int concat_only;
    struct {
        const void* w;             // pack_tail_weights layout
        const float* scale;
        const float* shift;
        const void* src2;          // concat partner slice (64 channels) or null
        int src2_cstride, src2_coff, cout;
    } tail[2];
    void* fin;
    int fin_cstride, fin_coff;
};
// LDS pair: a following 1x1 conv with at most as many output channels as the head (128 or 256) runs from the head's
// tile kept in LDS.  `w` = that conv's ordinary packed weights; d->out is written only if store_x.
struct ConvPairDesc {
    const void* w;
    const float* scale;
    const float* shift;
    int act, cout, out_f32, store_x;
    void* fin;
    int fin_cstride, fin_coff;
    void* fin2;                    // tail = a fused CSP pair (cout = both convs' rows): rows >= split go here
    int fin2_cstride, fin2_coff, split;
};
// A head conv (float32 raw logits) also leaves its three OBJECTNESS logits per cell -- channels a * nf + 4 -- in a dense side
// array [image][cell of the image, the three scales back to back][4] (one 16-byte slot per cell), which is what decode's screen
// reads instead of three 64-byte sectors of the 1 KB cell (decode_nms.hip).  The written values are the stored logits themselves.
struct ConvObjDesc {
    float* obj;                    // slot of this launch's first image's first cell
    int nf, cells_per_img, cell_base;
};
int conv2d_launch(const y4_conv_desc* d, const char* zero_page, hipStream_t stream, const ConvChainDesc* chain = nullptr,
                  const ConvPairDesc* pair = nullptr, const ConvObjDesc* obj = nullptr);
int pack_tail_weights(int dtype, int cout, int cin, const float* oihw, void* packed, hipStream_t stream);
int conv_tile_count();
bool weight_touch_enabled();          // conv_common.h: weight_touch (off with Y4_NO_WEIGHT_TOUCH=1, for A/B runs)
int conv_pick_tile(int dtype, int M, int cin, int cout);
int pack_conv_weights(int dtype, int cout, int cin, int ksize, const float* oihw, void* packed, hipStream_t stream);
// 3x3 packed weights -> MFMA-fragment order of the halo2 tiles (conv_halo2_kernel.h); same byte count
int pack_conv_frag32(int dtype, int cout, int cin, const void* packed, void* frag, hipStream_t stream);

// misc_kernels.hip
int stem_conv_launch(int dtype, const void* imgs, int img_u8, int n, int h, int w, const float* wk, const float* scale,
                     const float* shift, int cout, int act, void* out, int out_cstride, int out_coff, hipStream_t stream);
int resize_u8_launch(const uint8_t* img, int n, int h, int w, uint8_t* out, int H, int W, hipStream_t stream);
int pack_stem_weights(const float* w_oihw, float* wk, int cout, hipStream_t stream);
int preprocess_u8_launch(const uint8_t* img, int h, int w, float* out, int H, int W, hipStream_t stream);
int spp_launch(int dtype, void* buf, int n, int side, int c, hipStream_t stream);
int view_to_f32_launch(int dtype, const void* src, float* dst, int64_t pixels, int cstride, int coff, int c,
                       hipStream_t stream);
int f32_to_view_launch(const float* src, float* dst, int64_t pixels, int cstride, int c, hipStream_t stream);
int fold_bn_launch(const float* rec, float* scale, float* shift, int cout, int cout_pad, int has_bn, hipStream_t stream);

// stem_down.hip: convs 0+1 fused (16-bit dtypes), c0 stays in LDS
bool stem_down_supported(int dtype, int S);
int stem_down_launch(int dtype, const void* imgs, int img_u8, int n, int S, const void* stem_wk, const float* s0_scale,
                     const float* s0_shift, int act0, const void* w1_packed, const float* s1_scale, const float* s1_shift,
                     int act1, void* out, int out_cstride, int out_coff, hipStream_t stream);

// csp_stage.hip: convs 2..7 (the first CSP stage) as one spatially tiled persistent kernel (16-bit dtypes)
bool csp_stage_supported(int dtype, int side);
size_t csp_stage_blob_bytes();
int pack_csp_stage(int dtype, const float* const* w, const float* const* scale, const float* const* shift, void* blob,
                   hipStream_t stream);
int csp_stage_launch(int dtype, const void* in, int n, int side, int in_cstride, int in_coff, const void* blob, void* out,
                     int out_cstride, int out_coff, hipStream_t stream);

// resblock.hip: "1x1 conv -> 3x3 conv + Add" residual blocks with 64 / 128 channels as one spatially tiled kernel
bool resblock_supported(int dtype, int c);
size_t resblock_blob_bytes(int c);
int pack_resblock(int dtype, int c, const float* w1, const float* scale1, const float* shift1, const float* w3, const float* scale3,
                  const float* shift3, void* blob, hipStream_t stream);
int resblock_launch(int dtype, int c, const void* in, int n, int side, int in_cstride, int in_coff, const void* blob, void* out,
                    int out_cstride, int out_coff, hipStream_t stream);

// decode_nms.hip
// Per-image candidate counters are spaced one per 256 bytes: packed into one cache line, the ~10^3 appends per
// image of a whole batch serialise on a single L2 line (measured: decode 195 us -> see DESIGN.md).
constexpr int COUNT_STRIDE = 64;     // uint32 words
struct DecodeK {
    const float* head[3];
    int g[3], stride[3], box_off[3];
    float xyscale[3], xyoff[3];      // xyoff = float(0.5*(xyscale-1)) computed in double like the reference
    float anchors[18];
    int cells_per_img;               // g0^2 + g1^2 + g2^2
    int N, C, hcs, nbox;
    float img_size, score_thr;
    float* dboxes;                   // [N, nbox, 4] normalised x1,y1,x2,y2
    unsigned long long* keys;        // [N, cap]
    uint32_t* counts;                // [N * COUNT_STRIDE]: one counter per image, each on its own 256-byte line
    uint32_t cap;
    FastDiv div_cells, div_g[3];     // cell id -> image, cell -> row (ids < 2^31: checked at y4_create)
    const float* obj;                // [N * cells_per_img][4]: the cells' objectness logits as the head convs left them (ConvObjDesc), or null
};
struct NmsK {
    const float* dboxes;             // [N, nbox, 4]
    const unsigned long long* keys;  // [N, cap]
    uint32_t* counts;                // [N * COUNT_STRIDE]; a block resets its image's counter once it has read it
    uint32_t cap;
    int N, C, nbox, max_total, max_per_class;
    float iou_thr;
    float* out_boxes;                // [N, max_total, 4]
    float* out_scores;               // [N, max_total]
    float* out_classes;              // [N, max_total]
    int32_t* out_valid;              // [N]
    int32_t* out_idx;                // [N, max_total] or null
    uint32_t* status;                // [1] bit0: candidate list overflowed its capacity
    FastDiv div_c;                   // id -> (box, class) without integer division (ids < 2^31 checked at y4_create)
};
int decode_launch(const DecodeK& k, hipStream_t stream, int clear_images);
// tuner aid (latency schedules): streams `bytes` of `p` through the L2s so that the next launch starts on cold weights
int l2_flush_launch(const void* p, size_t bytes, void* sink, hipStream_t stream);
int nms_launch(const NmsK& k, hipStream_t stream);

}  // namespace y4
