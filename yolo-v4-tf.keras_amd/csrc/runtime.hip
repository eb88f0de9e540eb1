// runtime.hip -- host side of libyolo4hip: the C ABI of include/yolo4hip.h, the 110-conv plan
// (reference custom_layers.py:100-198 cspdarknet53 + yolov4_neck), the buffer plan and the executor.
//
// Plan notes
//  * conv index == call order below == Keras creation order == Darknet blob order (reference utils.py:19-21).
//  * every Concatenate of the reference is realised as ONE NHWC buffer whose channel slices are written in
//    place by the producing convs; UpSampling2D is the 2x2-replicating store of the
//    producing 1x1 conv; each residual Add is the epilogue of its 3x3 conv.
//  * activations for max_batch images are carved out of the caller's workspace once, at bind time.
#include <stdarg.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "conv_tiles.h"
#include "kernels.h"

namespace y4 {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

struct View {
    int buf = -1;          // buffer id
    int side = 0, cstride = 0, coff = 0, c = 0;
};
struct Buffer {
    int side, channels;
    bool f32;              // raw heads are float32 whatever the compute dtype
    size_t offset = 0;     // byte offset in the act workspace (for max_batch images)
    size_t bytes = 0;
};
enum OpKind { OP_STEM, OP_CONV, OP_SPP };
struct Op {
    OpKind kind;
    int conv = -1;
    View in, out, res;
    bool has_res = false, upsample = false, out_f32 = false;
    int tile = 0;          // 0 = heuristic; set by y4_autotune
    int conv2 = -1;        // fused CSP pair: second conv index (main-in), its output view and the channel split
    View out2;
    int split = 0;
    char name[16];
};
struct Layer {
    y4_layer_desc d;
    int cout_pad;
    int fused_with = -1;   // >= 0: this layer's rows live in layer `fused_with`'s packed matrix at row `fused_row`
    int fused_row = 0;
    int extra_rows = 0;    // rows appended by a fused partner
    size_t w_off, scale_off, shift_off;   // byte offsets in the wts workspace
    bool has_tail = false;                // a chain can run this 1x1 conv as a tail: its weights are also kept in
    size_t tail_off = 0;                  // fragment order (pack_tail_weights) at this offset
    bool has_frag = false;                // a 3x3 stride-1 conv the halo2 tiles can run (16-bit, cin % 64 == 0): its weights are also kept
    size_t frag_off = 0;                  // in MFMA-fragment order (pack_conv_frag32) at this offset
};

// A run of ops executed as one kernel (conv_chain.h): head = 3x3 conv with residual Add and 64 output channels,
// then one or two 1x1 convs; the second reads Concatenate([first tail's output, route]).
struct Chain {
    int head, tail[2];      // op indices (tail[1] = -1 for a 2-conv chain)
    bool store_x;           // the head's output has other consumers and is still written
    bool lds_pair = false;  // the tail runs from the head's tile kept in LDS (128/256 channels), not from registers
    bool enabled = true;    // y4_autotune turns a run off when its separate kernels measure faster
    int tile = 0;           // the head's tile when it runs chained (0 = heuristic); Op::tile stays the unfused choice
    // "Alternative" run (conv_chain.h CFG 4): the 64 -> 64 tail of run `alt_of` as a HEAD chained straight to that run's conv
    // over the concat (tail[0] here).  In force only while run `alt_of` cannot exist because its own head executes inside
    // a residual-block kernel.
    int alt_of = -1;
};

// A residual block "1x1 conv -> 3x3 conv + Add(block input)" executed as one spatially tiled kernel (resblock.hip)
struct ResRun {
    int head, tail;         // op indices of the 1x1 and the 3x3 conv
    int c;                  // channels (64 | 128): the tuner decides per channel group
    size_t blob_off = 0;    // fragment-ordered weights + affine in the wts workspace
};

}  // namespace y4

using namespace y4;

struct y4_ctx {
    y4_config cfg;
    int es;                           // element size of the compute dtype
    int S;                            // img side
    int hcs;                          // padded channels of a raw head
    int nbox;
    int64_t flops_per_image = 0;
    int64_t weight_floats = 0;
    std::vector<Layer> layers;
    std::vector<Buffer> bufs;
    std::vector<Op> ops;
    View heads[3];
    // workspace layout
    size_t act_bytes = 0, wts_bytes = 0;
    size_t zero_off = 0, dbox_off = 0, keys_off = 0, counts_off = 0, status_off = 0, scratch_off = 0, splitk_off = 0, obj_off = 0;
    // the cells' objectness logits beside the raw heads (kernels.h: ConvObjDesc): head i's are those of its images [0, obj_n[i])
    // -- written by the head conv's own launch, so they ARE the stored logits; y4_set_heads writes heads only and voids them
    int obj_n[3] = {-1, -1, -1};
    int cells_per_img = 0, cell_base[3] = {0, 0, 0};
    // latency schedules: y4_autotune may pick split-K tile ids (conv_tiles.h); their counters + partial sums live at splitk_off
    bool allow_splitk = false;
    // y4_autotune may pick halo2 tile ids (conv_halo2_kernel.h: v_mfma_32x32x16, another fp32 summation order than the 16x16x32 tiles)
    bool allow_halo2 = false;
    // decode's per-image candidate counters are zero (nms_kernel resets them); false: the next decode clears them itself
    bool counts_clean = false;
    int counts_n = 0;
    uint32_t cand_cap = 0;
    char* act = nullptr;
    char* wts = nullptr;
    bool weights_ready = false;
    // timing session (y4_timing_begin/end): per-op HIP events recorded by y4_predict
    std::vector<hipEvent_t> t_events;
    int t_max_steps = 0, t_steps = 0, t_per_step = 0, t_n = 0;
    std::vector<int> t_slot_op;       // op index credited with each event interval of a step
    std::vector<char> t_rec;          // per launch: record an event after it?
    bool t_coarse = false;            // events only where the op kind changes (stem | conv run | spp | conv run | decode | nms)
    // sub-batching: ops [0, sub_last_op] run over `sub_images` images at a time (keeps the large early
    // activations of one sub-batch resident in the 256 MiB Infinity Cache between producer and consumer)
    int sub_images = 0, sub_last_op = -1;
    // convs 0+1 as one kernel (stem_down.hip): op 0 launches it into op 1's output view, op 1 becomes a no-op
    bool fuse_stem = false;
    // 3x3+Add -> 1x1 (-> 1x1 over the concat) runs found in the plan, executed as one kernel each when fuse_chains
    std::vector<Chain> chains;
    bool fuse_chains = false;
    bool t_recorded_this_call = false;
    // the first CSP stage (convs 2..7) as one spatially tiled kernel (csp_stage.hip): ops [stage_first, stage_last];
    // stage_first < 0 when the plan / dtype does not allow it.  `stage_on` = requested, `stage_enabled` = the tuner's verdict
    int stage_first = -1, stage_last = -1;
    size_t stage_blob_off = 0;
    bool stage_on = false, stage_enabled = true;
    bool stage_active() const { return stage_first >= 0 && stage_on && stage_enabled; }
    // residual blocks of the 128- and 64-channel stages as one kernel each; group 0 = C 128, group 1 = C 64
    std::vector<ResRun> resruns;
    bool res_on = false, res_enabled[2] = {true, true};
    const ResRun* res_of(int oi, bool* is_head) const {
        if (!res_on) return nullptr;
        for (const ResRun& r : resruns)
            if ((r.head == oi || r.tail == oi) && res_enabled[r.c == 128 ? 0 : 1]) { if (is_head) *is_head = r.head == oi; return &r; }
        return nullptr;
    }
    // does this run execute as one kernel under the current settings?  (force_chain: the tuner times an alternative run while
    // the residual-block kernels are switched off)
    int force_chain = -1;
    bool chain_active(const Chain& ch) const {
        if (!fuse_chains || !ch.enabled) return false;
        const int ci = (int)(&ch - chains.data());
        if (ch.alt_of >= 0) return ci == force_chain || res_of(chains[ch.alt_of].head, nullptr) != nullptr;
        if (force_chain >= 0 && chains[force_chain].alt_of == ci) return false;
        return res_of(ch.head, nullptr) == nullptr;
    }
    // images of the current call are uint8 frames at network size (y4_forward_u8 / y4_predict_u8): the stem divides by 255
    bool img_u8 = false;
    // activation buffers share memory when their lifetimes do not overlap (y4_set_workspace_aliasing): a quarter of the
    // workspace, a hotter working set for the Infinity Cache; intermediate tensors are then not retained after a forward
    bool alias_bufs = false;
};

namespace {

struct Builder {
    y4_ctx& c;
    explicit Builder(y4_ctx& ctx) : c(ctx) {}

    View alloc(int side, int ch, bool f32 = false) {
        Buffer b{side, ch, f32};
        c.bufs.push_back(b);
        View v;
        v.buf = (int)c.bufs.size() - 1; v.side = side; v.cstride = ch; v.coff = 0; v.c = ch;
        return v;
    }
    static View slice(View t, int off, int ch) {
        View v = t;
        v.coff = t.coff + off; v.c = ch;
        return v;
    }
    // one conv() unit of the reference; `out` must already have the right shape
    void conv(View in, View out, int filters, int k, bool down, int act, bool bn = true, const View* res = nullptr,
              bool upsample = false, bool out_f32 = false) {
        Layer L{};
        const int idx = (int)c.layers.size();
        L.d.idx = idx; L.d.ksize = k; L.d.stride = down ? 2 : 1; L.d.cin = in.c; L.d.cout = filters;
        L.d.act = act; L.d.has_bn = bn ? 1 : 0; L.d.in_side = in.side; L.d.out_side = down ? in.side / 2 : in.side;
        L.d.weight_offset = c.weight_floats;
        c.weight_floats += (bn ? 4 : 1) * (int64_t)filters + (int64_t)filters * in.c * k * k;
        c.flops_per_image += 2ll * k * k * in.c * filters * L.d.out_side * L.d.out_side;
        L.cout_pad = (int)round_up(filters, COUT_PAD);
        c.layers.push_back(L);
        Op op{};
        op.kind = idx == 0 ? OP_STEM : OP_CONV;
        op.conv = idx; op.in = in; op.out = out; op.upsample = upsample; op.out_f32 = out_f32;
        if (res) { op.res = *res; op.has_res = true; }
        snprintf(op.name, sizeof(op.name), "c%d", idx);
        c.ops.push_back(op);
    }
    View conv_new(View in, int filters, int k, bool down, int act) {
        View out = alloc(down ? in.side / 2 : in.side, filters);
        conv(in, out, filters, k, down, act);
        return out;
    }
    // csp_block (reference custom_layers.py:47-69): returns the [x, route] concat buffer
    View csp(View din, int width, int repeat, bool bottleneck) {
        View cat = alloc(din.side, 2 * width);
        // route conv (created first) and main-in conv read the same tensor: ONE GEMM with 2*width output
        // channels, [0,width) -> the route slice of the concat buffer, [width,2*width) -> x
        conv(din, slice(cat, width, width), width, 1, false, Y4_ACT_MISH);          // route
        View x = conv_new(din, width, 1, false, Y4_ACT_MISH);                        // main-in
        {
            Op second = c.ops.back(); c.ops.pop_back();
            Op& first = c.ops.back();
            first.conv2 = second.conv; first.out2 = second.out; first.split = width;
            snprintf(first.name, sizeof(first.name), "c%d+%d", first.conv, second.conv);
            Layer& L1 = c.layers[first.conv];
            Layer& L2 = c.layers[second.conv];
            L2.fused_with = first.conv; L2.fused_row = width;
            L1.extra_rows = width + COUT_PAD;      // partner rows + slack for its zero padding
        }
        for (int r = 0; r < repeat; ++r) {                                           // residual_block :34-44
            View t = conv_new(x, bottleneck ? width / 2 : width, 1, false, Y4_ACT_MISH);
            View x2 = alloc(din.side, width);
            conv(t, x2, width, 3, false, Y4_ACT_MISH, true, &x);                     // 3x3 + Add
            x = x2;
        }
        conv(x, slice(cat, 0, width), width, 1, false, Y4_ACT_MISH);                 // main-out
        return cat;
    }
    // the 5-conv PANet run (1x1, 3x3, 1x1, 3x3, 1x1); the last 1x1 writes into `last_out`
    View five(View x, int narrow, View last_out) {
        const int L = Y4_ACT_LEAKY;
        x = conv_new(x, narrow, 1, false, L);
        x = conv_new(x, narrow * 2, 3, false, L);
        x = conv_new(x, narrow, 1, false, L);
        x = conv_new(x, narrow * 2, 3, false, L);
        conv(x, last_out, narrow, 1, false, L);
        return last_out;
    }

    void build() {
        const int S = c.S, M = Y4_ACT_MISH, L = Y4_ACT_LEAKY;
        const int nout = 3 * (c.cfg.num_classes + 5);
        View img; img.buf = -1; img.side = S; img.cstride = 3; img.coff = 0; img.c = 3;     // caller's images
        // ---- cspdarknet53, reference custom_layers.py:100-138
        View x = conv_new(img, 32, 3, false, L);                 // c0  (leaky, as the reference)
        x = conv_new(x, 64, 3, true, L);                         // c1  (leaky, as the reference)
        x = csp(x, 64, 1, true);
        x = conv_new(x, 64, 1, false, M);
        x = conv_new(x, 128, 3, true, M);
        x = csp(x, 64, 2, false);
        x = conv_new(x, 128, 1, false, M);
        x = conv_new(x, 256, 3, true, M);
        x = csp(x, 128, 8, false);
        View route0 = x = conv_new(x, 256, 1, false, M);         // c37
        x = conv_new(x, 512, 3, true, M);
        x = csp(x, 256, 8, false);
        View route1 = x = conv_new(x, 512, 1, false, M);         // c58
        x = conv_new(x, 1024, 3, true, M);
        x = csp(x, 512, 4, false);
        x = conv_new(x, 1024, 1, false, M);                      // c71
        x = conv_new(x, 512, 1, false, L);
        x = conv_new(x, 1024, 3, false, L);
        View cat_spp = alloc(x.side, 2048);                      // [mp13 | mp9 | mp5 | x]
        conv(x, slice(cat_spp, 1536, 512), 512, 1, false, L);    // c74
        {
            Op op{};
            op.kind = OP_SPP; op.in = cat_spp; op.out = cat_spp;
            snprintf(op.name, sizeof(op.name), "spp");
            c.ops.push_back(op);
        }
        x = conv_new(cat_spp, 512, 1, false, L);
        x = conv_new(x, 1024, 3, false, L);
        View cat_bu2 = alloc(x.side, 1024);                      // [down(route1'') | route2]
        View route2 = slice(cat_bu2, 512, 512);
        conv(x, route2, 512, 1, false, L);                       // c77
        // ---- yolov4_neck, reference custom_layers.py:141-198
        View cat_td1 = alloc(route1.side, 512);                  // [lateral(route1) | up(route2)]
        conv(route2, slice(cat_td1, 256, 256), 256, 1, false, L, true, nullptr, /*upsample=*/true);   // c78
        conv(route1, slice(cat_td1, 0, 256), 256, 1, false, L);                                       // c79
        View cat_bu1 = alloc(route1.side, 512);                  // [down(route0') | route1']
        View route1p = five(cat_td1, 256, slice(cat_bu1, 256, 256));                                  // c80-c84
        View cat_td0 = alloc(route0.side, 256);                  // [lateral(route0) | up(route1')]
        conv(route1p, slice(cat_td0, 128, 128), 128, 1, false, L, true, nullptr, /*upsample=*/true);  // c85
        conv(route0, slice(cat_td0, 0, 128), 128, 1, false, L);                                       // c86
        View route0p = five(cat_td0, 128, alloc(route0.side, 128));                                   // c87-c91
        x = conv_new(route0p, 256, 3, false, L);                                                      // c92
        c.heads[0] = alloc(route0.side, c.hcs, true); c.heads[0].c = nout;
        conv(x, c.heads[0], nout, 1, false, Y4_ACT_LINEAR, false, nullptr, false, true);              // c93
        conv(route0p, slice(cat_bu1, 0, 256), 256, 3, true, L);                                       // c94
        View route1pp = five(cat_bu1, 256, alloc(route1.side, 256));                                  // c95-c99
        x = conv_new(route1pp, 512, 3, false, L);                                                     // c100
        c.heads[1] = alloc(route1.side, c.hcs, true); c.heads[1].c = nout;
        conv(x, c.heads[1], nout, 1, false, Y4_ACT_LINEAR, false, nullptr, false, true);              // c101
        conv(route1pp, slice(cat_bu2, 0, 512), 512, 3, true, L);                                      // c102
        x = five(cat_bu2, 512, alloc(cat_bu2.side, 512));                                             // c103-c107
        x = conv_new(x, 1024, 3, false, L);                                                           // c108
        c.heads[2] = alloc(cat_bu2.side, c.hcs, true); c.heads[2].c = nout;
        conv(x, c.heads[2], nout, 1, false, Y4_ACT_LINEAR, false, nullptr, false, true);              // c109
    }
};

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

bool same_view(const View& a, const View& b) { return a.buf == b.buf && a.coff == b.coff && a.c == b.c; }

// Find the 3x3+Add -> 1x1 [-> 1x1 over Concatenate([., route])] runs of the plan that conv_chain.h can execute as one
// kernel (16-bit dtypes): head cout 64, Mish everywhere, tails 64 -> 64 and 128 -> 64|128.
void find_chains(y4_ctx& c) {
    if (c.cfg.dtype == Y4_F32) return;
    const int nops = (int)c.ops.size();
    auto readers = [&](int buf, int except_a, int except_b) {     // does any other op read this buffer?
        for (int i = 0; i < nops; ++i) {
            if (i == except_a || i == except_b) continue;
            const Op& o = c.ops[i];
            if (o.in.buf == buf || (o.has_res && o.res.buf == buf)) return true;
        }
        return false;
    };
    auto in_chain = [&](int oi) {
        for (const Chain& ch : c.chains)
            if (ch.head == oi || ch.tail[0] == oi || ch.tail[1] == oi) return true;
        return false;
    };
    // LDS pairs whose tail is a fused CSP pair: the stride-2 3x3 conv that opens a stage -> that stage's route | main-in
    // 1x1 GEMM (custom_layers.py:107-108, :112-113 -> :58-60).  Found first: they save more traffic than the
    // split-head register chain that could start at the same CSP pair.
    for (int i = 0; i + 1 < nops; ++i) {
        const Op& a = c.ops[i];
        const Op& b = c.ops[i + 1];
        if (a.kind != OP_CONV || b.kind != OP_CONV || a.conv2 >= 0 || b.conv2 < 0) continue;
        const Layer& la = c.layers[a.conv];
        const Layer& lb = c.layers[b.conv];
        if ((la.d.cout != 128 && la.d.cout != 256) || a.upsample || a.out_f32 || a.has_res) continue;
        if (!(lb.d.ksize == 1 && lb.d.cin == la.d.cout && 2 * lb.d.cout == la.d.cout && b.split == lb.d.cout && !b.has_res &&
              same_view(b.in, a.out)))
            continue;
        Chain ch{i, {i + 1, -1}, readers(a.out.buf, i + 1, -1)};
        ch.lds_pair = true;
        c.chains.push_back(ch);
    }
    for (int i = 0; i + 1 < nops; ++i) {
        const Op& a = c.ops[i];
        const Op& b = c.ops[i + 1];
        if (in_chain(i) || in_chain(i + 1)) continue;
        if (a.kind == OP_CONV && b.kind == OP_CONV && a.conv2 >= 0 && b.conv2 < 0) {
            // fused CSP pair (64 + 64 rows over a 64-channel input) -> the 1x1 conv on its main-in half
            const Layer& la = c.layers[a.conv];
            const Layer& lb = c.layers[b.conv];
            if (la.d.ksize == 1 && la.d.cin % 64 == 0 && la.d.cout == 64 && a.split == 64 && la.d.act == Y4_ACT_MISH && !a.has_res &&
                lb.d.ksize == 1 && lb.d.cin == 64 && (lb.d.cout == 32 || lb.d.cout == 64) && lb.d.act == Y4_ACT_MISH && !b.has_res &&
                !b.upsample && !b.out_f32 && same_view(b.in, a.out2) && !c.layers[b.conv].has_tail) {
                Chain ch{i, {i + 1, -1}, true};          // both halves of the pair have later readers
                ch.lds_pair = true;
                c.chains.push_back(ch);
            }
            continue;
        }
        if (a.kind != OP_CONV || b.kind != OP_CONV || a.conv2 >= 0 || b.conv2 >= 0) continue;
        const Layer& la = c.layers[a.conv];
        const Layer& lb = c.layers[b.conv];
        if (!(la.d.ksize == 3 && la.d.stride == 1 && a.has_res && la.d.cout == 64 && la.d.act == Y4_ACT_MISH && !a.upsample)) continue;
        if (!(lb.d.ksize == 1 && lb.d.cin == 64 && lb.d.cout == 64 && lb.d.act == Y4_ACT_MISH && !b.has_res && !b.upsample &&
              !b.out_f32 && same_view(b.in, a.out)))
            continue;
        Chain ch{i, {i + 1, -1}, readers(a.out.buf, i + 1, -1)};
        if (i + 2 < nops) {
            const Op& d = c.ops[i + 2];
            if (d.kind == OP_CONV && d.conv2 < 0) {
                const Layer& ld = c.layers[d.conv];
                // d reads the whole concat buffer whose first 64 channels are b's output
                if (ld.d.ksize == 1 && ld.d.cin == 128 && (ld.d.cout == 64 || ld.d.cout == 128) && ld.d.act == Y4_ACT_MISH &&
                    !d.has_res && !d.upsample && !d.out_f32 && d.in.buf == b.out.buf && d.in.coff == 0 && b.out.coff == 0 &&
                    d.in.cstride == 128 && !readers(b.out.buf, i + 2, -1))
                    ch.tail[1] = i + 2;
            }
        }
        c.layers[b.conv].has_tail = true;
        if (ch.tail[1] >= 0) c.layers[c.ops[ch.tail[1]].conv].has_tail = true;
        c.chains.push_back(ch);
    }
    // alternatives (CFG 4): [1x1 64 -> 64] -> [1x1 over the concat, 128 -> 128] of every three-conv run above
    for (size_t ci = 0, nci = c.chains.size(); ci < nci; ++ci) {
        const Chain& pc = c.chains[ci];
        if (pc.lds_pair || pc.tail[1] < 0 || c.layers[c.ops[pc.tail[1]].conv].d.cout != 128) continue;
        Chain alt{pc.tail[0], {pc.tail[1], -1}, false};
        alt.alt_of = (int)ci;
        c.chains.push_back(alt);
    }
    // LDS pairs: conv (128 or 256 output channels) -> 1x1 conv with the same channel count reading exactly that output
    // (the residual blocks of the 76^2 and 38^2 stages: 3x3 + Add -> the next block's 1x1, custom_layers.py:34-44)
    for (int i = 0; i + 1 < nops; ++i) {
        const Op& a = c.ops[i];
        const Op& b = c.ops[i + 1];
        if (a.kind != OP_CONV || b.kind != OP_CONV || a.conv2 >= 0 || b.conv2 >= 0 || in_chain(i) || in_chain(i + 1)) continue;
        const Layer& la = c.layers[a.conv];
        const Layer& lb = c.layers[b.conv];
        if ((la.d.cout != 128 && la.d.cout != 256) || a.upsample || a.out_f32) continue;
        if (!(lb.d.ksize == 1 && lb.d.cin == la.d.cout && lb.d.cout <= la.d.cout && !b.has_res && !b.upsample && same_view(b.in, a.out)))
            continue;
        Chain ch{i, {i + 1, -1}, readers(a.out.buf, i + 1, -1)};
        ch.lds_pair = true;
        c.chains.push_back(ch);
    }
}

// The first CSP stage in the op list: [route|main-in pair (64+64 over 64)] [1x1 64->32] [3x3 32->64 + Add] [1x1 64->64]
// [1x1 128->64 over the concat], all Mish -- reference custom_layers.py:47-69 with residual_bottleneck=True, and :105.
void find_stage(y4_ctx& c) {
    if (c.cfg.dtype == Y4_F32) return;
    const int nops = (int)c.ops.size();
    for (int i = 0; i + 4 < nops; ++i) {
        const Op &a = c.ops[i], &b = c.ops[i + 1], &d = c.ops[i + 2], &e = c.ops[i + 3], &f = c.ops[i + 4];
        if (a.kind != OP_CONV || b.kind != OP_CONV || d.kind != OP_CONV || e.kind != OP_CONV || f.kind != OP_CONV) continue;
        if (a.conv2 < 0 || b.conv2 >= 0 || d.conv2 >= 0 || e.conv2 >= 0 || f.conv2 >= 0) continue;
        const Layer &la = c.layers[a.conv], &la2 = c.layers[a.conv2], &lb = c.layers[b.conv], &ld = c.layers[d.conv],
                    &le = c.layers[e.conv], &lf = c.layers[f.conv];
        auto mish1x1 = [](const Layer& l, int cin, int cout) {
            return l.d.ksize == 1 && l.d.cin == cin && l.d.cout == cout && l.d.act == Y4_ACT_MISH && l.d.has_bn;
        };
        if (!(mish1x1(la, 64, 64) && mish1x1(la2, 64, 64) && a.split == 64 && mish1x1(lb, 64, 32) && mish1x1(le, 64, 64) &&
              mish1x1(lf, 128, 64)))
            continue;
        if (!(ld.d.ksize == 3 && ld.d.stride == 1 && ld.d.cin == 32 && ld.d.cout == 64 && ld.d.act == Y4_ACT_MISH && d.has_res)) continue;
        // dataflow: b reads a's main-in half, d reads b and adds a's main-in half, e reads d, f reads [e | a's route half]
        if (!(same_view(b.in, a.out2) && same_view(d.in, b.out) && same_view(d.res, a.out2) && same_view(e.in, d.out))) continue;
        if (!(f.in.buf == e.out.buf && f.in.buf == a.out.buf && e.out.coff == f.in.coff && a.out.coff == f.in.coff + 64 && f.in.c == 128)) continue;
        if (a.upsample || f.upsample || f.out_f32 || a.in.side % 16 != 0) continue;
        // no other op may read the tensors that stop existing
        bool leak = false;
        for (int k = 0; k < nops; ++k) {
            if (k >= i && k <= i + 4) continue;
            const Op& o = c.ops[k];
            for (int bufid : {a.out.buf, a.out2.buf, b.out.buf, d.out.buf})
                if (o.in.buf == bufid || (o.has_res && o.res.buf == bufid)) leak = true;
        }
        if (leak) continue;
        c.stage_first = i; c.stage_last = i + 4;
        return;
    }
}

// residual_block (reference custom_layers.py:34-44) with filters1 == filters2 in {64, 128}: ops [1x1 C->C][3x3 C->C + Add(x)]
void find_resruns(y4_ctx& c) {
    if (c.cfg.dtype == Y4_F32) return;
    const int nops = (int)c.ops.size();
    for (int i = 0; i + 1 < nops; ++i) {
        const Op &a = c.ops[i], &b = c.ops[i + 1];
        if (a.kind != OP_CONV || b.kind != OP_CONV || a.conv2 >= 0 || b.conv2 >= 0 || a.has_res || !b.has_res) continue;
        const Layer &la = c.layers[a.conv], &lb = c.layers[b.conv];
        const int ch = la.d.cin;
        if (!resblock_supported(c.cfg.dtype, ch)) continue;
        if (!(la.d.ksize == 1 && la.d.cout == ch && la.d.act == Y4_ACT_MISH && la.d.has_bn && lb.d.ksize == 3 && lb.d.stride == 1 &&
              lb.d.cin == ch && lb.d.cout == ch && lb.d.act == Y4_ACT_MISH && lb.d.has_bn))
            continue;
        if (!(same_view(b.in, a.out) && same_view(b.res, a.in)) || a.upsample || b.upsample || a.out_f32 || b.out_f32) continue;
        bool leak = false;                         // the 1x1 conv's output must have no other reader
        for (int k = 0; k < nops; ++k)
            if (k != i + 1 && (c.ops[k].in.buf == a.out.buf || (c.ops[k].has_res && c.ops[k].res.buf == a.out.buf))) leak = true;
        if (leak) continue;
        c.resruns.push_back(ResRun{i, i + 1, ch});
    }
}

// Lifetime of every activation buffer in op-index time, for the aliasing layout.  A buffer lives from the first op that
// writes it to the last op that reads it; every group of ops that CAN run as one kernel (chains and LDS pairs, the stage
// kernel, residual blocks -- whether or not the tuner enables them) counts as one instant for everything it touches, because a
// fused kernel writes its tails' outputs while other workgroups of the same launch still read the head's inputs.  The raw
// heads live to the end (decode reads them; y4_get_heads).
static void buffer_lifetimes(const y4_ctx& c, std::vector<int>& first, std::vector<int>& last) {
    const int nb = (int)c.bufs.size(), nops = (int)c.ops.size();
    first.assign(nb, nops); last.assign(nb, -1);
    std::vector<int> lo(nops), hi(nops);                 // the instant [lo, hi] an op belongs to
    for (int i = 0; i < nops; ++i) lo[i] = hi[i] = i;
    auto merge = [&](int a, int b) {                     // ops a..b are one instant (merged with whatever they already belong to)
        int l = a, h = b;
        for (int i = a; i <= b; ++i) { l = std::min(l, lo[i]); h = std::max(h, hi[i]); }
        for (int i = l; i <= h; ++i) { lo[i] = std::min(lo[i], l); hi[i] = std::max(hi[i], h); }
    };
    for (const Chain& ch : c.chains) merge(ch.head, std::max(ch.tail[0], ch.tail[1]));
    if (c.stage_first >= 0) merge(c.stage_first, c.stage_last);
    for (const ResRun& r : c.resruns) merge(r.head, r.tail);
    if (nops > 1) merge(0, 1);                           // convs 0 + 1 (stem_down)
    for (int pass = 0; pass < 4; ++pass)                 // close the merge transitively (groups that overlap chain together)
        for (int i = 0; i < nops; ++i) merge(lo[i], hi[i]);
    auto touch = [&](const View& v, int i) {
        if (v.buf < 0) return;
        first[v.buf] = std::min(first[v.buf], lo[i]);
        last[v.buf] = std::max(last[v.buf], hi[i]);
    };
    for (int i = 0; i < nops; ++i) {
        const Op& o = c.ops[i];
        touch(o.in, i); touch(o.out, i);
        if (o.has_res) touch(o.res, i);
        if (o.conv2 >= 0) touch(o.out2, i);
    }
    for (int k = 0; k < 3; ++k) last[c.heads[k].buf] = nops;          // the raw heads outlive the forward
}

void layout(y4_ctx& c) {
    // ---- activations
    size_t off = 0;
    c.zero_off = off; off += ZERO_PAGE_BYTES;
    const size_t nb = (size_t)c.cfg.max_batch;
    for (auto& b : c.bufs) b.bytes = nb * b.side * b.side * b.channels * (b.f32 ? 4 : c.es);
    if (!c.alias_bufs) {
        for (auto& b : c.bufs) {
            b.offset = off;
            off = align256(off + b.bytes);
        }
    } else {
        // greedy interval colouring by size: the largest buffer first, each at the lowest offset where it does not overlap (in
        // memory) any already placed buffer whose lifetime intersects its own
        std::vector<int> first, last, order(c.bufs.size());
        buffer_lifetimes(c, first, last);
        for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
        std::sort(order.begin(), order.end(), [&](int a, int b) { return c.bufs[a].bytes != c.bufs[b].bytes ? c.bufs[a].bytes > c.bufs[b].bytes : a < b; });
        const size_t base = align256(off);
        size_t top = base;
        std::vector<int> placed;
        for (int bi : order) {
            Buffer& b = c.bufs[bi];
            size_t cand = base;
            for (bool moved = true; moved;) {
                moved = false;
                for (int pj : placed) {
                    const Buffer& o = c.bufs[pj];
                    const bool live_together = !(last[bi] < first[pj] || last[pj] < first[bi]);
                    if (live_together && cand < o.offset + align256(o.bytes) && o.offset < cand + align256(b.bytes)) {
                        cand = o.offset + align256(o.bytes);
                        moved = true;
                    }
                }
            }
            b.offset = cand;
            placed.push_back(bi);
            top = std::max(top, cand + align256(b.bytes));
        }
        off = top;
    }
    c.dbox_off = off; off = align256(off + nb * c.nbox * 16);
    c.cand_cap = (uint32_t)c.nbox * (uint32_t)c.cfg.num_classes;         // worst case: exact for any input
    c.keys_off = off; off = align256(off + nb * (size_t)c.cand_cap * 8);
    c.counts_off = off; off = align256(off + nb * 4 * COUNT_STRIDE);
    c.status_off = off; off = align256(off + 256);
    c.scratch_off = off; off = align256(off + nb * (size_t)c.cfg.max_total * 28 + nb * 4);
    c.splitk_off = off; off = align256(off + SPLITK_WS_BYTES);
    c.cells_per_img = 0;
    for (int i = 0; i < 3; ++i) { c.cell_base[i] = c.cells_per_img; c.cells_per_img += c.heads[i].side * c.heads[i].side; }
    c.obj_off = off; off = align256(off + nb * (size_t)c.cells_per_img * 16);
    c.act_bytes = off;
    // ---- weights
    off = 0;
    for (auto& L : c.layers) {
        const size_t rows = (size_t)L.cout_pad + L.extra_rows;
        const size_t wbytes = L.d.idx == 0 ? (size_t)8192 : rows * L.d.ksize * L.d.ksize * L.d.cin * c.es;
        L.w_off = off; off = align256(off + wbytes);
        L.scale_off = off; off = align256(off + rows * 4);
        L.shift_off = off; off = align256(off + rows * 4);
        if (L.has_tail) { L.tail_off = off; off = align256(off + (size_t)L.d.cout * L.d.cin * c.es); }
        L.has_frag = c.cfg.dtype != Y4_F32 && L.d.idx > 0 && L.d.ksize == 3 && L.d.stride == 1 && L.d.cin % 64 == 0 && L.fused_with < 0 && L.extra_rows == 0;
        if (L.has_frag) { L.frag_off = off; off = align256(off + wbytes); }
    }
    if (c.stage_first >= 0) { c.stage_blob_off = off; off = align256(off + csp_stage_blob_bytes()); }
    for (ResRun& r : c.resruns) { r.blob_off = off; off = align256(off + resblock_blob_bytes(r.c)); }
    c.wts_bytes = off;
}

int check_handle(y4_handle h) {
    Y4_REQUIRE(h != nullptr, Y4_EINVAL, "null handle");
    return Y4_OK;
}
int check_ready(y4_handle h, int n) {
    if (int r = check_handle(h)) return r;
    Y4_REQUIRE(h->act && h->wts, Y4_ESTATE, "workspace not bound (call y4_bind_workspace first)");
    Y4_REQUIRE(h->weights_ready, Y4_ESTATE, "weights not packed (call y4_pack_weights first)");
    Y4_REQUIRE(n >= 1 && n <= h->cfg.max_batch, Y4_EINVAL, "batch %d outside [1, max_batch=%d]", n, h->cfg.max_batch);
    return Y4_OK;
}

char* buf_ptr(y4_handle h, const View& v, int img0 = 0) {
    const Buffer& b = h->bufs[v.buf];
    return h->act + b.offset + (size_t)img0 * b.side * b.side * b.channels * (b.f32 ? 4 : h->es);
}

// image `img0` of the caller's batch (float32 or uint8 elements, see y4_ctx::img_u8)
const void* img_at(y4_handle h, const void* imgs, int img0) {
    if (!imgs) return imgs;
    return (const char*)imgs + (size_t)img0 * h->S * h->S * 3 * (h->img_u8 ? 1 : 4);
}

struct Launch {
    int op, img0, cnt;
};
// execution schedule for a batch of n images (see y4_set_subbatch)
void build_schedule(y4_handle h, int n, std::vector<Launch>& out) {
    out.clear();
    const int nops = (int)h->ops.size();
    int first_full = 0;
    if (h->sub_images > 0 && h->sub_last_op >= 0 && n > h->sub_images) {
        for (int img0 = 0; img0 < n; img0 += h->sub_images)
            for (int i = 0; i <= h->sub_last_op; ++i) out.push_back({i, img0, n - img0 < h->sub_images ? n - img0 : h->sub_images});
        first_full = h->sub_last_op + 1;
    }
    for (int i = first_full; i < nops; ++i) out.push_back({i, 0, n});
}

int run_op(y4_handle h, const Op& op, const void* imgs, int n, hipStream_t s, int img0 = 0, bool allow_chain = true) {
    if (op.kind == OP_SPP) return spp_launch(h->cfg.dtype, buf_ptr(h, op.in, img0), n, op.in.side, op.in.cstride / 4, s);
    const Layer& L = h->layers[op.conv];
    const float* scale = (const float*)(h->wts + L.scale_off);
    const float* shift = (const float*)(h->wts + L.shift_off);
    if (op.kind == OP_STEM && h->fuse_stem) {
        const Op& o1 = h->ops[1];
        const Layer& L1 = h->layers[1];
        return stem_down_launch(h->cfg.dtype, img_at(h, imgs, img0), h->img_u8 ? 1 : 0, n, h->S,
                                h->wts + L.w_off, scale, shift, L.d.act, h->wts + L1.w_off,
                                (const float*)(h->wts + L1.scale_off), (const float*)(h->wts + L1.shift_off), L1.d.act,
                                buf_ptr(h, o1.out, img0), o1.out.cstride, o1.out.coff, s);
    }
    if (h->fuse_stem && op.kind == OP_CONV && op.conv == 1) return Y4_OK;
    if (allow_chain && op.kind == OP_CONV) {
        bool is_head = false;
        if (const ResRun* rr = h->res_of((int)(&op - h->ops.data()), &is_head)) {
            if (!is_head) return Y4_OK;                          // the 3x3 conv ran inside its block's kernel
            const Op& last = h->ops[rr->tail];
            const int64_t per_img = (int64_t)op.in.side * op.in.side * op.in.cstride * h->es;
            const int max_n = (int)(((1ll << 31) - 1) / per_img);
            Y4_REQUIRE(max_n >= 1, Y4_EINVAL, "residual block: one image of the input view (%lld B) exceeds the 2 GiB buffer range", (long long)per_img);
            for (int i0 = 0; i0 < n; i0 += max_n) {
                const int cnt = n - i0 < max_n ? n - i0 : max_n;
                if (int r = resblock_launch(h->cfg.dtype, rr->c, buf_ptr(h, op.in, img0 + i0), cnt, op.in.side, op.in.cstride, op.in.coff,
                                            h->wts + rr->blob_off, buf_ptr(h, last.out, img0 + i0), last.out.cstride, last.out.coff, s))
                    return r;
            }
            return Y4_OK;
        }
    }
    if (allow_chain && h->stage_active() && op.kind == OP_CONV) {
        const int oi = (int)(&op - h->ops.data());
        if (oi > h->stage_first && oi <= h->stage_last) return Y4_OK;      // ran inside the stage kernel
        if (oi == h->stage_first) {
            const Op& last = h->ops[h->stage_last];
            const int64_t per_img = (int64_t)op.in.side * op.in.side * op.in.cstride * h->es;
            const int max_n = (int)(((1ll << 31) - 1) / per_img);
            Y4_REQUIRE(max_n >= 1, Y4_EINVAL, "stage kernel: one image of the input view (%lld B) exceeds the 2 GiB buffer range", (long long)per_img);
            for (int i0 = 0; i0 < n; i0 += max_n) {           // 2 GiB buffer-descriptor range: image chunks
                const int cnt = n - i0 < max_n ? n - i0 : max_n;
                if (int r = csp_stage_launch(h->cfg.dtype, buf_ptr(h, op.in, img0 + i0), cnt, op.in.side, op.in.cstride, op.in.coff,
                                             h->wts + h->stage_blob_off, buf_ptr(h, last.out, img0 + i0), last.out.cstride,
                                             last.out.coff, s))
                    return r;
            }
            return Y4_OK;
        }
    }
    const Chain* chain = nullptr;
    if (h->fuse_chains && allow_chain && op.kind == OP_CONV)
        for (const Chain& ch : h->chains) {
            if (!h->chain_active(ch)) continue;     // (a head inside a residual-block kernel: its tails run alone or as the alternative run)
            if (&h->ops[ch.head] == &op) chain = &ch;
            else if (&h->ops[ch.tail[0]] == &op || (ch.tail[1] >= 0 && &h->ops[ch.tail[1]] == &op)) return Y4_OK;   // ran with its head
        }
    if (op.kind == OP_STEM)
        return stem_conv_launch(h->cfg.dtype, img_at(h, imgs, img0), h->img_u8 ? 1 : 0, n, h->S, h->S,
                                (const float*)(h->wts + L.w_off), scale, shift, L.d.cout, L.d.act,
                                buf_ptr(h, op.out, img0), op.out.cstride, op.out.coff, s);
    {
        // the conv kernels address their input through a raw buffer descriptor (2 GiB range, conv_igemm.hip): a batch
        // whose input view is larger runs as consecutive image chunks of this same op (images are independent)
        const int64_t per_img = (int64_t)op.in.side * op.in.side * op.in.cstride * h->es;
        // (minus the reach of a 3x3 kernel's biased tap offsets: conv2d_launch's own check)
        const int64_t tap_span = L.d.ksize == 3 ? ((int64_t)3 * op.in.side + 3) * op.in.cstride * h->es + (int64_t)op.in.c * h->es : 0;
        const int max_n = (int)(((1ll << 31) - 1 - tap_span) / per_img);
        Y4_REQUIRE(max_n >= 1, Y4_EINVAL, "conv %d: one image's input (%lld B) exceeds the 2 GiB buffer-descriptor range",
                   op.conv, (long long)per_img);
        if (n > max_n) {
            for (int i0 = 0; i0 < n; i0 += max_n)
                if (int r = run_op(h, op, imgs, n - i0 < max_n ? n - i0 : max_n, s, img0 + i0, allow_chain)) return r;
            return Y4_OK;
        }
    }
    y4_conv_desc d{};
    d.dtype = h->cfg.dtype;
    d.n = n; d.h = op.in.side; d.w = op.in.side; d.cin = op.in.c;
    d.cout = op.conv2 >= 0 ? 2 * L.d.cout : L.d.cout; d.ksize = L.d.ksize; d.stride = L.d.stride; d.act = L.d.act;
    d.upsample = op.upsample ? 1 : 0; d.out_f32 = op.out_f32 ? 1 : 0;
    d.in_cstride = op.in.cstride; d.in_coff = op.in.coff;
    d.out_cstride = op.out.cstride; d.out_coff = op.out.coff;
    d.in = buf_ptr(h, op.in, img0); d.wt = h->wts + L.w_off; d.scale = scale; d.shift = shift;
    d.wt_frag = L.has_frag ? h->wts + L.frag_off : nullptr;
    d.out = buf_ptr(h, op.out, img0);
    if (op.has_res) { d.res = buf_ptr(h, op.res, img0); d.res_cstride = op.res.cstride; d.res_coff = op.res.coff; }
    if (op.conv2 >= 0) { d.out2 = buf_ptr(h, op.out2, img0); d.out2_cstride = op.out2.cstride; d.out2_coff = op.out2.coff; d.split = op.split; }
    d.tile = chain ? chain->tile : op.tile;
    d.splitk_ws = h->act + h->splitk_off; d.splitk_ws_bytes = SPLITK_WS_BYTES;
    // a float32 head (as a plain launch or as the tail of an LDS pair) also fills its cells' slots of the objectness array
    ConvObjDesc od{};
    const ConvObjDesc* odp = nullptr;
    int obj_head = -1;
    {
        const Op* ho = op.out_f32 ? &op : (chain && chain->lds_pair && h->ops[chain->tail[0]].out_f32 ? &h->ops[chain->tail[0]] : nullptr);
        if (ho)
            for (int i = 0; i < 3; ++i)
                if (h->heads[i].buf == ho->out.buf) {
                    od.obj = (float*)(h->act + h->obj_off) + (size_t)img0 * h->cells_per_img * 4;
                    od.nf = 5 + h->cfg.num_classes; od.cells_per_img = h->cells_per_img; od.cell_base = h->cell_base[i];
                    odp = &od;
                    obj_head = i;
                    h->obj_n[i] = -1;                     // (valid again once the launch below has been enqueued)
                }
    }
    auto launched = [&](int rc) {                          // the side array counts only after a successful enqueue
        if (rc == Y4_OK && obj_head >= 0) h->obj_n[obj_head] = img0 + n;
        return rc;
    };
    if (chain && chain->lds_pair) {
        const Op& to = h->ops[chain->tail[0]];
        const Layer& TL = h->layers[to.conv];
        ConvPairDesc pd{};
        pd.w = h->wts + TL.w_off;
        pd.scale = (const float*)(h->wts + TL.scale_off);
        pd.shift = (const float*)(h->wts + TL.shift_off);
        pd.act = TL.d.act; pd.cout = TL.d.cout; pd.out_f32 = to.out_f32 ? 1 : 0; pd.store_x = chain->store_x ? 1 : 0;
        pd.fin = buf_ptr(h, to.out, img0); pd.fin_cstride = to.out.cstride; pd.fin_coff = to.out.coff;
        if (to.conv2 >= 0) {          // the tail is a fused CSP pair: both convs' rows, split over two views
            pd.cout = 2 * TL.d.cout; pd.split = to.split;
            pd.fin2 = buf_ptr(h, to.out2, img0); pd.fin2_cstride = to.out2.cstride; pd.fin2_coff = to.out2.coff;
        }
        return launched(conv2d_launch(&d, h->act + h->zero_off, s, nullptr, &pd, odp));
    }
    if (chain && chain->alt_of >= 0) {
        // this op (64 -> 64) feeds the conv over Concatenate([its output, route]) in registers; its own output is not stored
        const Op& to = h->ops[chain->tail[0]];
        const Layer& TL = h->layers[to.conv];
        ConvChainDesc cd{};
        cd.ntail = 2; cd.concat_only = 1; cd.store_x = 0;
        cd.tail[1].w = h->wts + TL.tail_off;
        cd.tail[1].scale = (const float*)(h->wts + TL.scale_off);
        cd.tail[1].shift = (const float*)(h->wts + TL.shift_off);
        cd.tail[1].cout = TL.d.cout;
        cd.tail[1].src2 = buf_ptr(h, to.in, img0);
        cd.tail[1].src2_cstride = to.in.cstride;
        cd.tail[1].src2_coff = to.in.coff + 64;
        cd.fin = buf_ptr(h, to.out, img0); cd.fin_cstride = to.out.cstride; cd.fin_coff = to.out.coff;
        return conv2d_launch(&d, h->act + h->zero_off, s, &cd);
    }
    if (chain) {
        ConvChainDesc cd{};
        cd.store_x = chain->store_x ? 1 : 0;
        const Op* last = &op;
        for (int t = 0; t < 2 && chain->tail[t] >= 0; ++t) {
            const Op& to = h->ops[chain->tail[t]];
            const Layer& TL = h->layers[to.conv];
            cd.tail[t].w = h->wts + TL.tail_off;
            cd.tail[t].scale = (const float*)(h->wts + TL.scale_off);
            cd.tail[t].shift = (const float*)(h->wts + TL.shift_off);
            cd.tail[t].cout = TL.d.cout;
            if (t == 1) {                       // the concat partner: channels [64, 128) of the buffer the first tail writes into
                cd.tail[t].src2 = buf_ptr(h, to.in, img0);
                cd.tail[t].src2_cstride = to.in.cstride;
                cd.tail[t].src2_coff = to.in.coff + 64;
            }
            cd.ntail = t + 1;
            last = &to;
        }
        cd.fin = buf_ptr(h, last->out, img0); cd.fin_cstride = last->out.cstride; cd.fin_coff = last->out.coff;
        return conv2d_launch(&d, h->act + h->zero_off, s, &cd);
    }
    return launched(conv2d_launch(&d, h->act + h->zero_off, s, nullptr, nullptr, odp));
}

int run_decode_nms(y4_handle h, int n, float iou_thr, float score_thr, float* boxes, float* scores, float* classes,
                   int32_t* valid, int32_t* kept_idx, hipStream_t s, int stage /*0 both, 1 decode, 2 nms*/) {
    const y4_config& cfg = h->cfg;
    if (stage != 2) {
        DecodeK k{};
        int off = 0, cells = 0;
        for (int i = 0; i < 3; ++i) {
            k.head[i] = (const float*)buf_ptr(h, h->heads[i]);
            k.g[i] = h->heads[i].side; k.stride[i] = cfg.strides[i]; k.box_off[i] = off;
            off += 3 * k.g[i] * k.g[i]; cells += k.g[i] * k.g[i];
            k.xyscale[i] = cfg.xyscale[i];
            k.xyoff[i] = (float)(0.5 * ((double)cfg.xyscale[i] - 1.0));
        }
        memcpy(k.anchors, cfg.anchors, sizeof(k.anchors));
        k.cells_per_img = cells; k.N = n; k.C = cfg.num_classes; k.hcs = h->hcs; k.nbox = h->nbox;
        k.div_cells = fastdiv_make((uint32_t)cells);
        for (int i = 0; i < 3; ++i) k.div_g[i] = fastdiv_make((uint32_t)k.g[i]);
        k.img_size = (float)cfg.img_size; k.score_thr = score_thr;
        k.dboxes = (float*)(h->act + h->dbox_off);
        k.keys = (unsigned long long*)(h->act + h->keys_off);
        k.counts = (uint32_t*)(h->act + h->counts_off);
        k.cap = h->cand_cap;
        k.obj = h->obj_n[0] >= n && h->obj_n[1] >= n && h->obj_n[2] >= n ? (const float*)(h->act + h->obj_off) : nullptr;
        if (int r = decode_launch(k, s, h->counts_clean ? 0 : cfg.max_batch)) return r;
        h->counts_clean = false;                          // ... until this decode's NMS has been enqueued behind it
        h->counts_n = n;
    }
    if (stage != 1) {
        NmsK k{};
        k.dboxes = (const float*)(h->act + h->dbox_off);
        k.keys = (const unsigned long long*)(h->act + h->keys_off);
        k.counts = (uint32_t*)(h->act + h->counts_off);
        k.cap = h->cand_cap; k.N = n; k.C = cfg.num_classes; k.nbox = h->nbox;
        k.max_total = cfg.max_total; k.max_per_class = cfg.max_per_class; k.iou_thr = iou_thr;
        k.out_boxes = boxes; k.out_scores = scores; k.out_classes = classes; k.out_valid = valid; k.out_idx = kept_idx;
        k.status = (uint32_t*)(h->act + h->status_off);
        k.div_c = fastdiv_make((uint32_t)cfg.num_classes);
        if (int r = nms_launch(k, s)) return r;
        h->counts_clean = h->counts_n <= n;               // (an NMS over fewer images than were decoded leaves counters behind)
    }
    return Y4_OK;
}

}  // namespace

extern "C" {

const char* y4_last_error(void) { return g_err; }
const char* y4_version(void) { return "yolo4hip 0.3 (gfx950)"; }

int y4_create(const y4_config* cfg, y4_handle* out) {
    Y4_REQUIRE(cfg && out, Y4_EINVAL, "y4_create: null argument");
    // the reference's asserts (models.py:23-24,38); non-square inputs are 'not support yet' there too
    Y4_REQUIRE(cfg->img_size > 0 && cfg->img_size % 32 == 0, Y4_EINVAL,
               "img_size %d must be a positive multiple of the last stride (32)", cfg->img_size);
    Y4_REQUIRE(cfg->num_classes > 0, Y4_EINVAL, "no classes detected!");
    Y4_REQUIRE(cfg->num_classes <= 4096, Y4_EINVAL, "num_classes %d too large", cfg->num_classes);
    Y4_REQUIRE(cfg->max_batch >= 1, Y4_EINVAL, "max_batch %d", cfg->max_batch);
    Y4_REQUIRE(cfg->dtype >= Y4_F32 && cfg->dtype <= Y4_F16, Y4_EINVAL, "dtype %d", cfg->dtype);
    Y4_REQUIRE(cfg->strides[0] == 8 && cfg->strides[1] == 16 && cfg->strides[2] == 32, Y4_EINVAL,
               "strides must be 8,16,32 (they are fixed by the graph)");
    Y4_REQUIRE(cfg->max_total >= 1 && cfg->max_total <= 1024 && cfg->max_per_class >= 1, Y4_EINVAL, "max_total/max_per_class");
    y4_ctx* c = new y4_ctx();
    c->cfg = *cfg;
    c->es = elem_size(cfg->dtype);
    c->S = cfg->img_size;
    c->hcs = (int)round_up(3 * (cfg->num_classes + 5), 8);
    c->nbox = 0;
    for (int i = 0; i < 3; ++i) { const int g = c->S / cfg->strides[i]; c->nbox += 3 * g * g; }
    if ((int64_t)c->nbox * cfg->num_classes >= (1ll << 31)) {
        delete c;
        set_error("num_boxes*num_classes overflows the 32-bit candidate id");
        return Y4_EINVAL;
    }
    Builder(*c).build();
    find_chains(*c);
    find_stage(*c);
    find_resruns(*c);
    layout(*c);
    *out = c;
    return Y4_OK;
}

int y4_destroy(y4_handle h) {
    delete h;
    return Y4_OK;
}

int y4_num_layers(y4_handle h) { return h ? (int)h->layers.size() : Y4_EINVAL; }

int y4_layer_info(y4_handle h, int idx, y4_layer_desc* out) {
    if (int r = check_handle(h)) return r;
    Y4_REQUIRE(out && idx >= 0 && idx < (int)h->layers.size(), Y4_EINVAL, "layer index %d", idx);
    *out = h->layers[idx].d;
    return Y4_OK;
}

int y4_model_info(y4_handle h, int64_t* flops_per_image, int32_t* num_boxes, int32_t* head_cstride, int64_t* weight_floats) {
    if (int r = check_handle(h)) return r;
    if (flops_per_image) *flops_per_image = h->flops_per_image;
    if (num_boxes) *num_boxes = h->nbox;
    if (head_cstride) *head_cstride = h->hcs;
    if (weight_floats) *weight_floats = h->weight_floats;
    return Y4_OK;
}

int y4_set_workspace_aliasing(y4_handle h, int on) {
    if (int r = check_handle(h)) return r;
    Y4_REQUIRE(!h->act, Y4_ESTATE, "y4_set_workspace_aliasing: the workspace is already bound (call it before y4_workspace_bytes / y4_bind_workspace)");
    Y4_REQUIRE(!(on && h->sub_images > 0), Y4_ESTATE, "y4_set_workspace_aliasing: not together with sub-batching");
    h->alias_bufs = on != 0;
    layout(*h);
    return Y4_OK;
}

int y4_workspace_bytes(y4_handle h, size_t* act_bytes, size_t* wts_bytes) {
    if (int r = check_handle(h)) return r;
    if (act_bytes) *act_bytes = h->act_bytes;
    if (wts_bytes) *wts_bytes = h->wts_bytes;
    return Y4_OK;
}

int y4_bind_workspace(y4_handle h, void* act_dev, size_t act_bytes, void* wts_dev, size_t wts_bytes) {
    if (int r = check_handle(h)) return r;
    Y4_REQUIRE(act_dev && wts_dev, Y4_EINVAL, "y4_bind_workspace: null workspace");
    Y4_REQUIRE(((uintptr_t)act_dev & 255) == 0 && ((uintptr_t)wts_dev & 255) == 0, Y4_EINVAL, "workspaces must be 256-byte aligned");
    Y4_REQUIRE(act_bytes >= h->act_bytes, Y4_ENOMEM, "act workspace %zu < %zu bytes", act_bytes, h->act_bytes);
    Y4_REQUIRE(wts_bytes >= h->wts_bytes, Y4_ENOMEM, "wts workspace %zu < %zu bytes", wts_bytes, h->wts_bytes);
    h->act = (char*)act_dev;
    h->counts_clean = false;
    h->wts = (char*)wts_dev;
    h->weights_ready = false;
    // The WHOLE activation workspace starts as zeros (round 5; once per bind, ~1 ms per 3 GB): the zero page, the NMS status word and
    // the split-K tile counters (zero between launches) need it, and no result can then depend on what the caller's memory held
    // before -- a workspace from a caching allocator carries the previous owner's tensors.  The previous owner's WORK too: kernels still
    // pending on some non-blocking stream against the recycled block would land after a null-stream memset, so the
    // device is drained first -- binding is an init-time call.
    Y4_CHECK_HIP(hipDeviceSynchronize());
    Y4_CHECK_HIP(hipMemset(h->act, 0, h->act_bytes));
    // ... and it HAS happened when this returns: the caller may use the handle on any stream next, and a non-blocking stream (every
    // torch.cuda.Stream is one) does not order itself behind the null stream the memset ran on
    Y4_CHECK_HIP(hipStreamSynchronize(nullptr));
    return Y4_OK;
}

int y4_pack_weights(y4_handle h, const float* blob, size_t n_floats, void* stream) {
    if (int r = check_handle(h)) return r;
    Y4_REQUIRE(h->act && h->wts, Y4_ESTATE, "workspace not bound (call y4_bind_workspace first)");
    Y4_REQUIRE(blob, Y4_EINVAL, "y4_pack_weights: null blob");
    Y4_REQUIRE((int64_t)n_floats >= h->weight_floats, Y4_EINVAL, "weight blob has %zu floats, the plan needs %lld",
               n_floats, (long long)h->weight_floats);
    hipStream_t s = (hipStream_t)stream;
    for (const Layer& L : h->layers) {
        const float* rec = blob + L.d.weight_offset;
        const float* w = rec + (L.d.has_bn ? 4 : 1) * (int64_t)L.d.cout;
        // a fused CSP pair shares one packed matrix: the partner's rows (and scale/shift) follow the first conv's
        const Layer& D = L.fused_with >= 0 ? h->layers[L.fused_with] : L;
        const size_t row_bytes = (size_t)L.d.ksize * L.d.ksize * L.d.cin * h->es;
        char* wdst = h->wts + D.w_off + (size_t)L.fused_row * row_bytes;
        if (int r = fold_bn_launch(rec, (float*)(h->wts + D.scale_off) + L.fused_row, (float*)(h->wts + D.shift_off) + L.fused_row,
                                   L.d.cout, L.cout_pad, L.d.has_bn, s))
            return r;
        if (L.d.idx == 0) {
            if (int r = pack_stem_weights(w, (float*)(h->wts + L.w_off), L.d.cout, s)) return r;
        } else {
            if (int r = pack_conv_weights(h->cfg.dtype, L.d.cout, L.d.cin, L.d.ksize, w, wdst, s)) return r;
            if (L.has_tail)
                if (int r = pack_tail_weights(h->cfg.dtype, L.d.cout, L.d.cin, w, h->wts + L.tail_off, s)) return r;
            if (L.has_frag)
                if (int r = pack_conv_frag32(h->cfg.dtype, L.d.cout, L.d.cin, wdst, h->wts + L.frag_off, s)) return r;
        }
    }
    if (h->stage_first >= 0) {
        // convs 2..7 once more, as the fragment-ordered blob of csp_stage_kernel (same rounding of the same floats)
        const Op& a = h->ops[h->stage_first];
        const int convs[6] = {a.conv, a.conv2, h->ops[h->stage_first + 1].conv, h->ops[h->stage_first + 2].conv,
                              h->ops[h->stage_first + 3].conv, h->ops[h->stage_first + 4].conv};
        const float *w[6], *sc[6], *sh[6];
        for (int k = 0; k < 6; ++k) {
            const Layer& L = h->layers[convs[k]];
            const Layer& D = L.fused_with >= 0 ? h->layers[L.fused_with] : L;
            w[k] = blob + L.d.weight_offset + 4 * (int64_t)L.d.cout;
            sc[k] = (const float*)(h->wts + D.scale_off) + L.fused_row;
            sh[k] = (const float*)(h->wts + D.shift_off) + L.fused_row;
        }
        if (int r = pack_csp_stage(h->cfg.dtype, w, sc, sh, h->wts + h->stage_blob_off, s)) return r;
    }
    for (const ResRun& r : h->resruns) {
        const Layer &L1 = h->layers[h->ops[r.head].conv], &L3 = h->layers[h->ops[r.tail].conv];
        if (int rc = pack_resblock(h->cfg.dtype, r.c, blob + L1.d.weight_offset + 4 * (int64_t)L1.d.cout,
                                   (const float*)(h->wts + L1.scale_off), (const float*)(h->wts + L1.shift_off),
                                   blob + L3.d.weight_offset + 4 * (int64_t)L3.d.cout, (const float*)(h->wts + L3.scale_off),
                                   (const float*)(h->wts + L3.shift_off), h->wts + r.blob_off, s))
            return rc;
    }
    h->weights_ready = true;
    return Y4_OK;
}

int y4_adopt_packed_weights(y4_handle h) {
    if (int r = check_handle(h)) return r;
    Y4_REQUIRE(h->act && h->wts, Y4_ESTATE, "workspace not bound (call y4_bind_workspace first)");
    h->weights_ready = true;
    return Y4_OK;
}

static int forward_impl(y4_handle h, const void* imgs, bool u8, int n, void* stream) {
    if (int r = check_ready(h, n)) return r;
    Y4_REQUIRE(imgs, Y4_EINVAL, "y4_forward: null images");
    h->img_u8 = u8;
    std::vector<Launch> sched;
    build_schedule(h, n, sched);
    int rc = Y4_OK;
    for (const Launch& l : sched)
        if ((rc = run_op(h, h->ops[l.op], imgs, l.cnt, (hipStream_t)stream, l.img0))) break;
    h->img_u8 = false;
    return rc;
}

// The forward pass up to and including conv `last_conv` (bench.py's backbone-only mode: last_conv = 71 is CSPDarknet53 proper,
// reference custom_layers.py:100-124).  The conv must end a launch: a run / residual-block kernel that is in force and continues
// past it is refused instead of being cut.
int y4_forward_until(y4_handle h, const float* imgs, int n, int last_conv, void* stream) {
    if (int r = check_ready(h, n)) return r;
    Y4_REQUIRE(imgs, Y4_EINVAL, "y4_forward_until: null images");
    int last_op = -1;
    for (int i = 0; i < (int)h->ops.size(); ++i)
        if (h->ops[i].kind != OP_SPP && (h->ops[i].conv == last_conv || h->ops[i].conv2 == last_conv)) last_op = i;
    Y4_REQUIRE(last_op >= 0, Y4_EINVAL, "y4_forward_until: no conv %d", last_conv);
    if (h->fuse_chains)
        for (const Chain& ch : h->chains)
            Y4_REQUIRE(!(h->chain_active(ch) && ch.head <= last_op && (ch.tail[0] > last_op || ch.tail[1] > last_op)), Y4_EINVAL,
                       "y4_forward_until: conv %d sits inside a fused run that continues behind it", last_conv);
    {
        bool is_head = false;
        Y4_REQUIRE(!(h->res_of(last_op, &is_head) && is_head), Y4_EINVAL,
                   "y4_forward_until: conv %d heads a residual-block kernel that continues behind it", last_conv);
    }
    Y4_REQUIRE(!(h->stage_active() && last_op >= h->stage_first && last_op < h->stage_last), Y4_EINVAL,
               "y4_forward_until: conv %d sits inside the stage kernel", last_conv);
    Y4_REQUIRE(!(h->fuse_stem && last_op == 0), Y4_EINVAL, "y4_forward_until: conv 0 runs fused with conv 1");
    h->img_u8 = false;
    std::vector<Launch> sched;
    build_schedule(h, n, sched);
    int rc = Y4_OK;
    for (const Launch& l : sched)
        if (l.op <= last_op && (rc = run_op(h, h->ops[l.op], imgs, l.cnt, (hipStream_t)stream, l.img0))) break;
    return rc;
}

int y4_forward(y4_handle h, const float* imgs, int n, void* stream) { return forward_impl(h, imgs, false, n, stream); }
int y4_forward_u8(y4_handle h, const uint8_t* imgs, int n, void* stream) { return forward_impl(h, imgs, true, n, stream); }

int y4_get_heads(y4_handle h, int n, float* out_s, float* out_m, float* out_l, void* stream) {
    if (int r = check_ready(h, n)) return r;
    float* outs[3] = {out_s, out_m, out_l};
    for (int i = 0; i < 3; ++i) {
        if (!outs[i]) continue;
        const View& v = h->heads[i];
        if (int r = view_to_f32_launch(Y4_F32, buf_ptr(h, v), outs[i], (int64_t)n * v.side * v.side, v.cstride, v.coff,
                                       v.c, (hipStream_t)stream))
            return r;
    }
    return Y4_OK;
}

int y4_set_heads(y4_handle h, int n, const float* in_s, const float* in_m, const float* in_l, void* stream) {
    if (int r = check_ready(h, n)) return r;
    const float* ins[3] = {in_s, in_m, in_l};
    for (int i = 0; i < 3; ++i) {
        Y4_REQUIRE(ins[i], Y4_EINVAL, "y4_set_heads: null head %d", i);
        h->obj_n[i] = -1;
        const View& v = h->heads[i];
        if (int r = f32_to_view_launch(ins[i], (float*)buf_ptr(h, v), (int64_t)n * v.side * v.side, v.cstride, v.c,
                                       (hipStream_t)stream))
            return r;
    }
    return Y4_OK;
}

int y4_get_conv_output(y4_handle h, int conv_idx, int n, float* out, size_t out_floats, void* stream) {
    if (int r = check_ready(h, n)) return r;
    Y4_REQUIRE(!h->alias_bufs, Y4_ESTATE, "intermediate tensors are not retained with workspace aliasing on");
    for (const Op& op : h->ops) {
        if (op.kind == OP_SPP || (op.conv != conv_idx && op.conv2 != conv_idx)) continue;
        Y4_REQUIRE(!(h->fuse_stem && conv_idx == 0), Y4_ESTATE, "conv 0 is not materialised while stem fusion is on");
        {
            const int oi = (int)(&op - h->ops.data());
            Y4_REQUIRE(!(h->stage_active() && oi >= h->stage_first && oi < h->stage_last), Y4_ESTATE,
                       "conv %d is not materialised while the stage fusion is on", conv_idx);
            bool is_head = false;
            Y4_REQUIRE(!(h->res_of(oi, &is_head) && is_head), Y4_ESTATE,
                       "conv %d is not materialised while the residual-block fusion is on", conv_idx);
        }
        const View& v = op.conv == conv_idx ? op.out : op.out2;
        const int64_t px = (int64_t)n * v.side * v.side;   // for an upsampling conv: the upsampled tensor
        Y4_REQUIRE((int64_t)out_floats >= px * v.c, Y4_EINVAL, "output buffer too small: %zu < %lld", out_floats,
                   (long long)(px * v.c));
        return view_to_f32_launch(op.out_f32 ? Y4_F32 : h->cfg.dtype, buf_ptr(h, v), out, px, v.cstride, v.coff, v.c,
                                  (hipStream_t)stream);
    }
    set_error("no conv with index %d", conv_idx);
    return Y4_EINVAL;
}

int y4_decode_nms(y4_handle h, int n, float iou_threshold, float score_threshold, float* boxes, float* scores,
                  float* classes, int32_t* valid, int32_t* kept_idx, void* stream) {
    if (int r = check_ready(h, n)) return r;
    Y4_REQUIRE(boxes && scores && classes && valid, Y4_EINVAL, "y4_decode_nms: null output");
    const float iou = iou_threshold < 0.f ? h->cfg.iou_threshold : iou_threshold;
    const float sc = score_threshold < 0.f ? h->cfg.score_threshold : score_threshold;
    return run_decode_nms(h, n, iou, sc, boxes, scores, classes, valid, kept_idx, (hipStream_t)stream, 0);
}

// forward + decode + NMS; when `ev` is given, ev[0] is recorded before the first op and ev[i+1] after op i
static int predict_impl(y4_handle h, const void* imgs, int n, float* boxes, float* scores, float* classes,
                        int32_t* valid, int32_t* kept_idx, hipStream_t s, hipEvent_t* ev) {
    if (int r = check_ready(h, n)) return r;
    Y4_REQUIRE(imgs && boxes && scores && classes && valid, Y4_EINVAL, "y4_predict: null argument");
    std::vector<Launch> sched;
    build_schedule(h, n, sched);
    if (ev && (int)sched.size() + 3 > h->t_per_step) ev = nullptr;       // cannot happen: sized for max_batch
    if (ev && h->t_steps == 0) {
        h->t_n = n;
        h->t_slot_op.clear();
        h->t_rec.assign(sched.size(), 1);
        int seg_first = sched.empty() ? 0 : sched[0].op;
        for (size_t k = 0; k < sched.size(); ++k) {
            const bool last = k + 1 == sched.size();
            // coarse segments: runs of same-kind ops, and the conv family additionally cut behind conv 71 -- the end of
            // CSPDarknet53 proper (reference custom_layers.py:124; north_star words its target on that backbone)
            const bool boundary = last || h->ops[sched[k + 1].op].kind != h->ops[sched[k].op].kind ||
                                  (h->ops[sched[k].op].kind == OP_CONV && h->ops[sched[k].op].conv == 71 &&
                                   h->ops[sched[k + 1].op].conv != 71);
            if (!h->t_coarse) {
                h->t_slot_op.push_back(sched[k].op);
            } else if (boundary) {
                h->t_slot_op.push_back(seg_first);           // the whole run of same-kind ops is credited to its first op
                if (!last) seg_first = sched[k + 1].op;
            } else {
                h->t_rec[k] = 0;
            }
        }
        h->t_slot_op.push_back((int)h->ops.size());
        h->t_slot_op.push_back((int)h->ops.size() + 1);
    }
    if (ev && n != h->t_n) ev = nullptr;                                   // a session times steps of one batch size
    h->t_recorded_this_call = ev != nullptr;
    int i = 0;
    if (ev) Y4_CHECK_HIP(hipEventRecord(ev[0], s));
    for (size_t k = 0; k < sched.size(); ++k) {
        const Launch& l = sched[k];
        if (int r = run_op(h, h->ops[l.op], imgs, l.cnt, s, l.img0)) return r;
        if (ev && h->t_rec[k]) Y4_CHECK_HIP(hipEventRecord(ev[++i], s));
    }
    for (int stage = 1; stage <= 2; ++stage) {
        if (int r = run_decode_nms(h, n, h->cfg.iou_threshold, h->cfg.score_threshold, boxes, scores, classes, valid,
                                   kept_idx, s, stage))
            return r;
        if (ev) Y4_CHECK_HIP(hipEventRecord(ev[++i], s));
    }
    return Y4_OK;
}

static int predict_any(y4_handle h, const void* imgs, bool u8, int n, float* boxes, float* scores, float* classes, int32_t* valid,
                       int32_t* kept_idx, void* stream) {
    if (int r = check_handle(h)) return r;
    hipEvent_t* ev = nullptr;
    if (h->t_max_steps > 0 && h->t_steps < h->t_max_steps) ev = h->t_events.data() + (size_t)h->t_steps * h->t_per_step;
    h->img_u8 = u8;
    const int rc = predict_impl(h, imgs, n, boxes, scores, classes, valid, kept_idx, (hipStream_t)stream, ev);
    h->img_u8 = false;
    if (ev && rc == Y4_OK && h->t_recorded_this_call) ++h->t_steps;
    return rc;
}

int y4_predict(y4_handle h, const float* imgs, int n, float* boxes, float* scores, float* classes, int32_t* valid,
               int32_t* kept_idx, void* stream) {
    return predict_any(h, imgs, false, n, boxes, scores, classes, valid, kept_idx, stream);
}
int y4_predict_u8(y4_handle h, const uint8_t* imgs, int n, float* boxes, float* scores, float* classes, int32_t* valid,
                  int32_t* kept_idx, void* stream) {
    return predict_any(h, imgs, true, n, boxes, scores, classes, valid, kept_idx, stream);
}

// Per-layer tile choice by measurement: every tile configuration that fits a conv is timed on the layer's
// real shape (HIP events on `stream`, whatever data is in the workspace) and the fastest is kept.  All tiles
// produce bit-identical results (same K order), so this changes speed only.
// With a second handle `h2` (a sibling: same plan, own workspace; y4_autotune_pair) every timed launch is issued on BOTH
// handles, each on its own stream, and the time until both streams are done is what counts: the objective becomes
// throughput with two batches in flight -- a tile whose last round leaves compute units idle no longer pays for them (the
// other stream fills them), so what wins is the least work, not the shortest solitary launch.  Both handles end up with
// the same choices.
static int autotune_impl(y4_handle h, y4_handle h2, int n, int reps, hipStream_t s, hipStream_t s2, int pair_passes) {
    if (int r = check_ready(h, n)) return r;
    if (h2)
        if (int r = check_ready(h2, n)) return r;
    Y4_REQUIRE(reps >= 1 && reps <= 100, Y4_EINVAL, "y4_autotune: reps %d", reps);
    Y4_REQUIRE(!h2 || (h2 != h && h2->ops.size() == h->ops.size() && h2->chains.size() == h->chains.size() &&
                       h2->cfg.dtype == h->cfg.dtype && h2->S == h->S && s2 != s),
               Y4_EINVAL, "y4_autotune_pair: the second handle must be a sibling of the first (same plan) on another stream");
    hipEvent_t e0, e1, e2;
    Y4_CHECK_HIP(hipEventCreate(&e0));
    Y4_CHECK_HIP(hipEventCreate(&e1));
    Y4_CHECK_HIP(hipEventCreate(&e2));
    int rc = Y4_OK;
    const int ntiles = conv_tile_count();
    // one op on the first handle, and on the second (its own stream) in the passes that use the two-stream objective
    // (pair_passes bit 0: tiles, 1: chains / LDS pairs, 2: stage kernel, 3: residual-block kernels)
    bool two = false;
    auto run_both = [&](int oi, int ne, bool chained) -> int {
        int r = run_op(h, h->ops[oi], nullptr, ne, s, 0, chained);
        if (r == Y4_OK && two) r = run_op(h2, h2->ops[oi], nullptr, ne, s2, 0, chained);
        return r;
    };
    // the timed region: e0 on s (s2 starts behind it), the launches, then s waits for s2's tail and e1 closes it
    auto t_begin = [&]() -> bool {
        if (hipEventRecord(e0, s) != hipSuccess) return false;
        return !two || hipStreamWaitEvent(s2, e0, 0) == hipSuccess;
    };
    auto t_end = [&](float* ms) -> bool {
        if (two && (hipEventRecord(e2, s2) != hipSuccess || hipStreamWaitEvent(s, e2, 0) != hipSuccess)) return false;
        return hipEventRecord(e1, s) == hipSuccess && hipEventSynchronize(e1) == hipSuccess &&
               hipEventElapsedTime(ms, e0, e1) == hipSuccess;
    };
    // time `reps` launches of one op; < 0: this tile does not fit, -2: HIP failure
    // Latency schedules (y4_set_splitk) are timed COLD: in a step of one image a layer finds its weights in no L2 (the other 109
    // layers' weights went through since), and the long serial K loop of a few-tile launch is exactly what that hurts -- back to
    // back the same launch runs on warm weights and the unsplit tile looks as good as the split one.  So every timed launch is
    // preceded by a pass of <= 64 MB of the packed weights through the L2s, outside its event pair.
    // (Y4_TUNE_COLD=1: cold timing for throughput schedules too -- an experiment switch, scripts/sched_ab_headline.sh)
    static const bool force_cold = [] { const char* e = getenv("Y4_TUNE_COLD"); return e && e[0] == '1'; }();
    const bool cold = (h->allow_splitk && !h2) || (force_cold && !h2);
    const size_t flush_bytes = h->wts_bytes < ((size_t)64 << 20) ? h->wts_bytes : ((size_t)64 << 20);
    auto time_op = [&](int oi, int ne, bool chained) -> float {
        if (run_both(oi, ne, chained) != Y4_OK) return -1.f;
        if (cold) {
            float total = 0.f;
            for (int i = 0; i < reps; ++i) {
                if (l2_flush_launch(h->wts, flush_bytes, h->act + h->scratch_off, s) != Y4_OK) return -2.f;
                if (hipEventRecord(e0, s) != hipSuccess) return -2.f;
                run_both(oi, ne, chained);
                float ms = 0.f;
                if (hipEventRecord(e1, s) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess)
                    return -2.f;
                total += ms;
            }
            return total;
        }
        if (!t_begin()) return -2.f;
        for (int i = 0; i < reps; ++i) run_both(oi, ne, chained);
        float ms = 0.f;
        return t_end(&ms) ? ms : -2.f;
    };
    auto set_tile = [&](int oi, int tile) { h->ops[oi].tile = tile; if (h2) h2->ops[oi].tile = tile; };
    auto images_of = [&](int oi) { return (h->sub_images > 0 && oi <= h->sub_last_op && n > h->sub_images) ? h->sub_images : n; };
    auto set_stage = [&](bool on) { h->stage_enabled = on; if (h2) h2->stage_enabled = on; };
    auto set_res = [&](int grp, bool on) { h->res_enabled[grp] = on; if (h2) h2->res_enabled[grp] = on; };
    set_stage(false);                  // passes 1 and 2 tune the stage's convs as separate kernels; pass 3 decides
    set_res(0, false); set_res(1, false);               // likewise the residual-block kernels: pass 4
    // pass 1: every conv as its own kernel
    two = h2 && (pair_passes & 1);
    std::vector<float> best_ms(h->ops.size(), 0.f);
    for (int oi = 0; oi < (int)h->ops.size() && rc == Y4_OK; ++oi) {
        Op& op = h->ops[oi];
        if (op.kind != OP_CONV || (h->fuse_stem && op.conv == 1)) continue;
        float best = 1e30f;
        int best_tile = 0;
        for (int tile = 1; tile <= ntiles; ++tile) {
            // the 32x32x16-MFMA tiles sum in another order than all the others: offering them here would make the outputs
            // depend on the tuner's choice.  They are measured slower anyway (LABNOTES.md section 4.1) and stay explicit-only.
            if (tuner_skips_tile(tile) && !(h->allow_halo2 && halo2_tile(tile))) continue;
            set_tile(oi, tile);
            const float ms = time_op(oi, images_of(oi), false);
            if (ms == -2.f) { rc = Y4_EHIP; break; }
            if (ms >= 0.f && ms < best) { best = ms; best_tile = tile; }
        }
        // latency schedules (y4_set_splitk): the same tiles with their K loop split 2 / 4 / 8 ways; the launcher refuses the ids
        // that do not fit the layer (too few K-tiles, too many output tiles for the scratch), which reads as "does not fit" here
        if (h->allow_splitk && rc == Y4_OK)
            for (int tile = 1; tile <= ntiles && rc == Y4_OK; ++tile) {
                if (tuner_skips_tile(tile) || !splitk_tile(tile)) continue;
                for (int e = 1; e <= SPLITK_MAX_E; ++e) {
                    set_tile(oi, tile + 100 * e);
                    const float ms = time_op(oi, images_of(oi), false);
                    if (ms == -2.f) { rc = Y4_EHIP; break; }
                    if (ms < 0.f) break;                       // a wider split of this tile will not fit either
                    if (ms < best) { best = ms; best_tile = tile + 100 * e; }
                }
            }
        set_tile(oi, best_tile);
        best_ms[oi] = best;
    }
    // pass 2: each chain as one kernel against the sum of its separate kernels
    two = h2 && (pair_passes & 2);
    if (h->fuse_chains)
        for (size_t ci = 0; ci < h->chains.size(); ++ci) {
            Chain& ch = h->chains[ci];
            auto mirror = [&]() { if (h2) { h2->chains[ci].tile = ch.tile; h2->chains[ci].enabled = ch.enabled; } };
            if (rc != Y4_OK) break;
            // an alternative run is timed as if its parent's head sat in a residual-block kernel (those are off until pass 4)
            h->force_chain = ch.alt_of >= 0 ? (int)ci : -1;
            if (h2) h2->force_chain = h->force_chain;
            float separate = best_ms[ch.head] + best_ms[ch.tail[0]] + (ch.tail[1] >= 0 ? best_ms[ch.tail[1]] : 0.f);
            float best = 1e30f;
            int best_tile = 0;
            ch.enabled = true;
            for (int tile = 1; tile <= ntiles; ++tile) {
                ch.tile = tile;
                mirror();
                const float ms = time_op(ch.head, images_of(ch.head), true);
                if (ms == -2.f) { rc = Y4_EHIP; break; }
                if (ms >= 0.f && ms < best) { best = ms; best_tile = tile; }
            }
            ch.tile = best_tile;
            ch.enabled = best_tile > 0;
            mirror();
            if (!ch.enabled) continue;
            // Final decision head to head, each variant as the executor would run it: the separate kernels in their real
            // order (head, tail, tail -- not the same kernel back to back, which runs warmer than it ever does in a step)
            // against the one fused launch, in alternating blocks and with several times the launches of a tile probe.
            // Comparing `best` with the SUM of three per-tile minima instead is biased towards "separate" (a minimum of
            // noisy samples is low, and three of them add up) and flipped the 8 % wins of the 76^2 pairs run to run.
            (void)separate;
            const int rounds = 4, per_round = reps > 3 ? reps : 3;
            auto block = [&](bool fused) -> float {
                if (!t_begin()) return -2.f;
                for (int i = 0; i < per_round; ++i) {
                    int r = Y4_OK;
                    if (fused) {
                        r = run_both(ch.head, images_of(ch.head), true);
                    } else {
                        r = run_both(ch.head, images_of(ch.head), false);
                        if (r == Y4_OK) r = run_both(ch.tail[0], images_of(ch.tail[0]), false);
                        if (r == Y4_OK && ch.tail[1] >= 0) r = run_both(ch.tail[1], images_of(ch.tail[1]), false);
                    }
                    if (r != Y4_OK) return fused ? -3.f : -4.f;      // a failed launch must not "win" with ~0 ms
                }
                float ms = 0.f;
                return t_end(&ms) ? ms : -2.f;
            };
            float t_fused = 0.f, t_sep = 0.f;
            const float w_f = block(true), w_s = block(false);                 // one untimed block each: both start equally warm
            if (w_f == -2.f || w_s == -2.f) { rc = Y4_EHIP; break; }           // (an event failure is not "fused lost")
            bool fused_ok = w_f >= 0.f, sep_ok = w_s >= 0.f;
            for (int r = 0; r < rounds && rc == Y4_OK && fused_ok && sep_ok; ++r) {
                const float a = block(true), b = block(false);
                if (a == -2.f || b == -2.f) { rc = Y4_EHIP; break; }
                if (a < 0.f) fused_ok = false;
                if (b < 0.f) sep_ok = false;
                t_fused += a; t_sep += b;
            }
            if (!sep_ok) { rc = Y4_EINVAL; break; }                   // the plain kernels themselves fail: report it (y4_last_error)
            ch.enabled = fused_ok && t_fused < t_sep;
            mirror();
        }
    h->force_chain = -1;
    if (h2) h2->force_chain = -1;
    // pass 3: the stage kernel (convs 2..7 in one launch) head to head against the same ops as tuned above
    two = h2 && (pair_passes & 4);
    if (rc == Y4_OK && h->stage_first >= 0 && h->stage_on) {
        const int ne = images_of(h->stage_first);
        const int rounds = 4, per_round = reps > 3 ? reps : 3;
        auto block = [&](bool fused) -> float {
            set_stage(fused);
            if (!t_begin()) return -2.f;
            for (int i = 0; i < per_round; ++i)
                for (int oi = h->stage_first; oi <= h->stage_last; ++oi)
                    if (run_both(oi, ne, true) != Y4_OK) return fused ? -3.f : -4.f;
            float ms = 0.f;
            return t_end(&ms) ? ms : -2.f;
        };
        float t_fused = 0.f, t_sep = 0.f;
        const float w_f = block(true), w_s = block(false);
        if (w_f == -2.f || w_s == -2.f) rc = Y4_EHIP;
        bool fused_ok = w_f >= 0.f, sep_ok = w_s >= 0.f;
        for (int r = 0; r < rounds && rc == Y4_OK && fused_ok && sep_ok; ++r) {
            const float a = block(true), b = block(false);
            if (a == -2.f || b == -2.f) { rc = Y4_EHIP; break; }
            if (a < 0.f) fused_ok = false;
            if (b < 0.f) sep_ok = false;
            t_fused += a; t_sep += b;
        }
        if (!sep_ok && rc == Y4_OK) rc = Y4_EINVAL;
        set_stage(rc == Y4_OK && fused_ok && t_fused < t_sep);
    }
    // pass 4: per channel group, the residual blocks as one kernel each against the same op range as tuned above
    two = h2 && (pair_passes & 8);
    for (int grp = 0; grp < 2 && rc == Y4_OK && h->res_on; ++grp) {
        int lo = -1, hi = -1;
        for (const ResRun& r : h->resruns)
            if ((r.c == 128 ? 0 : 1) == grp) { if (lo < 0) lo = r.head; hi = r.tail; }
        if (lo < 0) continue;
        // include the ops up to the end of the stage's residual chain consumers that existing chains may have fused with it
        for (const Chain& ch : h->chains)
            if (ch.head >= lo && ch.head <= hi) { if (ch.tail[0] > hi) hi = ch.tail[0]; if (ch.tail[1] > hi) hi = ch.tail[1]; }
        const int ne = images_of(lo);
        const int rounds = 4, per_round = reps > 3 ? reps : 3;
        auto block = [&](bool fused) -> float {
            set_res(grp, fused);
            if (!t_begin()) return -2.f;
            for (int i = 0; i < per_round; ++i)
                for (int oi = lo; oi <= hi; ++oi)
                    if (run_both(oi, ne, true) != Y4_OK) return fused ? -3.f : -4.f;
            float ms = 0.f;
            return t_end(&ms) ? ms : -2.f;
        };
        float t_fused = 0.f, t_sep = 0.f;
        const float w_f = block(true), w_s = block(false);
        if (w_f == -2.f || w_s == -2.f) rc = Y4_EHIP;
        bool fused_ok = w_f >= 0.f, sep_ok = w_s >= 0.f;
        for (int r = 0; r < rounds && rc == Y4_OK && fused_ok && sep_ok; ++r) {
            const float a = block(true), b = block(false);
            if (a == -2.f || b == -2.f) { rc = Y4_EHIP; break; }
            if (a < 0.f) fused_ok = false;
            if (b < 0.f) sep_ok = false;
            t_fused += a; t_sep += b;
        }
        if (!sep_ok && rc == Y4_OK) rc = Y4_EINVAL;
        set_res(grp, rc == Y4_OK && fused_ok && t_fused < t_sep);
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipEventDestroy(e2);
    if (rc == Y4_EHIP) set_error("y4_autotune: HIP event failure");     // (Y4_EINVAL: the failing launch's own message stays)
    return rc;
}

int y4_autotune(y4_handle h, int n, int reps, void* stream) { return autotune_impl(h, nullptr, n, reps, (hipStream_t)stream, nullptr, 0); }

int y4_autotune_pair(y4_handle h, y4_handle h2, int n, int reps, void* stream, void* stream2, int pair_passes) {
    if (int r = check_handle(h2)) return r;
    Y4_REQUIRE(pair_passes >= 0 && pair_passes <= 15, Y4_EINVAL, "y4_autotune_pair: pair_passes %d (bits 0..3)", pair_passes);
    if (int r = check_handle(h)) return r;
    // the sibling is tuned with the primary's choices: they only apply to it if it runs the same kernels
    Y4_REQUIRE(h2->fuse_stem == h->fuse_stem && h2->fuse_chains == h->fuse_chains && h2->stage_on == h->stage_on &&
               h2->res_on == h->res_on && h2->sub_images == h->sub_images && h2->sub_last_op == h->sub_last_op &&
               h2->alias_bufs == h->alias_bufs, Y4_ESTATE,
               "y4_autotune_pair: the two handles differ in their fusion switches, sub-batching or workspace aliasing "
               "(y4_copy_schedule(h, h2) makes them equal)");
    return autotune_impl(h, h2, n, reps, (hipStream_t)stream, (hipStream_t)stream2, pair_passes);
}

int y4_set_tiles(y4_handle h, const int32_t* tiles, int count) {
    if (int r = check_handle(h)) return r;
    Y4_REQUIRE(tiles && count == (int)h->layers.size(), Y4_EINVAL, "y4_set_tiles: expected %d entries", (int)h->layers.size());
    for (int oi = 0; oi < (int)h->ops.size(); ++oi) {
        Op& op = h->ops[oi];
        if (op.kind != OP_CONV) continue;
        const int v = tiles[op.conv];
        Chain* head_of = nullptr;
        for (Chain& ch : h->chains)
            if (ch.head == oi) head_of = &ch;
        // a run's head carries two choices in one entry: -(run tile + 1000 * stand-alone tile); plain -t leaves the stand-alone one
        const int run_tile = v < 0 ? (-v) % 1000 : 0, own_tile = v < 0 ? (-v) / 1000 : v;
        Y4_REQUIRE(tile_id_ok(own_tile) && run_tile <= conv_tile_count() && (v >= 0 || (head_of && h->fuse_chains)), Y4_EINVAL,
                   "y4_set_tiles: tile id %d for conv %d", v, op.conv);
        if (head_of && h->fuse_chains) {      // < 0: chained with tile run_tile; > 0: separate kernels; 0: chained, heuristic tile
            head_of->enabled = v <= 0;
            head_of->tile = run_tile;
            if (v > 0 || own_tile > 0) op.tile = own_tile;
        } else {
            op.tile = v;
        }
    }
    return Y4_OK;
}

int y4_set_subbatch(y4_handle h, int images, int last_conv) {
    if (int r = check_handle(h)) return r;
    Y4_REQUIRE(h->t_max_steps == 0, Y4_ESTATE, "y4_set_subbatch: a timing session is open");
    Y4_REQUIRE(images <= 0 || !h->alias_bufs, Y4_ESTATE, "y4_set_subbatch: not together with workspace aliasing (a sub-batch's "
               "early tensors would share memory with another sub-batch's later ones)");
    if (images <= 0) { h->sub_images = 0; h->sub_last_op = -1; return Y4_OK; }
    int last_op = -1;
    for (int i = 0; i < (int)h->ops.size(); ++i)
        if (h->ops[i].kind != OP_SPP && (h->ops[i].conv == last_conv || h->ops[i].conv2 == last_conv)) last_op = i;
    Y4_REQUIRE(last_op >= 0, Y4_EINVAL, "y4_set_subbatch: no conv %d", last_conv);
    h->sub_images = images;
    h->sub_last_op = last_op;
    return Y4_OK;
}

int y4_set_stem_fusion(y4_handle h, int on) {
    if (int r = check_handle(h)) return r;
    Y4_REQUIRE(h->t_max_steps == 0, Y4_ESTATE, "y4_set_stem_fusion: a timing session is open");
    if (on) {
        const bool shape_ok = h->ops.size() > 1 && h->ops[0].kind == OP_STEM && h->ops[1].kind == OP_CONV &&
                              h->ops[1].conv == 1 && h->ops[1].conv2 < 0 && !h->ops[1].has_res &&
                              h->layers[0].d.cout == 32 && h->layers[1].d.cout == 64 && h->layers[1].d.ksize == 3 &&
                              h->layers[1].d.stride == 2;
        Y4_REQUIRE(shape_ok && stem_down_supported(h->cfg.dtype, h->S), Y4_EINVAL,
                   "y4_set_stem_fusion: needs a 16-bit dtype and img_size <= 640 (dtype %d, img_size %d)", h->cfg.dtype, h->S);
    }
    h->fuse_stem = on != 0;
    return Y4_OK;
}

int y4_set_chain_fusion(y4_handle h, int on) {
    if (int r = check_handle(h)) return r;
    Y4_REQUIRE(h->t_max_steps == 0, Y4_ESTATE, "y4_set_chain_fusion: a timing session is open");
    Y4_REQUIRE(!on || h->cfg.dtype != Y4_F32, Y4_EINVAL, "y4_set_chain_fusion: 16-bit dtypes only");
    h->fuse_chains = on != 0;
    for (Chain& ch : h->chains) { ch.enabled = true; ch.tile = 0; }
    return on ? (int)h->chains.size() : Y4_OK;
}

int y4_set_stage_fusion(y4_handle h, int on) {
    if (int r = check_handle(h)) return r;
    Y4_REQUIRE(h->t_max_steps == 0, Y4_ESTATE, "y4_set_stage_fusion: a timing session is open");
    Y4_REQUIRE(!on || h->stage_first >= 0, Y4_EINVAL, "y4_set_stage_fusion: needs a 16-bit dtype (dtype %d)", h->cfg.dtype);
    h->stage_on = on != 0;
    h->stage_enabled = true;
    return h->stage_active() ? 1 : 0;
}

int y4_set_res_fusion(y4_handle h, int on) {
    if (int r = check_handle(h)) return r;
    Y4_REQUIRE(h->t_max_steps == 0, Y4_ESTATE, "y4_set_res_fusion: a timing session is open");
    Y4_REQUIRE(!on || h->cfg.dtype != Y4_F32, Y4_EINVAL, "y4_set_res_fusion: 16-bit dtypes only");
    h->res_on = on != 0;
    h->res_enabled[0] = h->res_enabled[1] = true;
    return on ? (int)h->resruns.size() : Y4_OK;
}

int y4_get_res_fusion(y4_handle h) {
    if (int r = check_handle(h)) return r;
    if (!h->res_on) return 0;
    int mask = 0;
    for (const ResRun& r : h->resruns) mask |= h->res_enabled[r.c == 128 ? 0 : 1] ? (r.c == 128 ? 1 : 2) : 0;
    return mask;
}

int y4_set_res_fusion_mask(y4_handle h, int mask) {
    if (int r = check_handle(h)) return r;
    Y4_REQUIRE(h->cfg.dtype != Y4_F32 || mask == 0, Y4_EINVAL, "y4_set_res_fusion_mask: 16-bit dtypes only");
    Y4_REQUIRE(mask >= 0 && mask <= 3, Y4_EINVAL, "y4_set_res_fusion_mask: mask %d (bit 0: 128-channel blocks, bit 1: 64-channel blocks)", mask);
    Y4_REQUIRE(h->t_max_steps == 0, Y4_ESTATE, "y4_set_res_fusion_mask: a timing session is open");
    h->res_on = mask != 0;
    h->res_enabled[0] = (mask & 1) != 0;
    h->res_enabled[1] = (mask & 2) != 0;
    return Y4_OK;
}

int y4_set_splitk(y4_handle h, int on) {
    if (int r = check_handle(h)) return r;
    h->allow_splitk = on != 0;
    return Y4_OK;
}

int y4_set_halo2(y4_handle h, int on) {
    if (int r = check_handle(h)) return r;
    Y4_REQUIRE(!on || h->cfg.dtype != Y4_F32, Y4_EINVAL, "y4_set_halo2: 16-bit dtypes only");
    h->allow_halo2 = on != 0;
    return Y4_OK;
}

int y4_get_stage_fusion(y4_handle h) {
    if (int r = check_handle(h)) return r;
    return h->stage_active() ? 1 : 0;
}

int y4_get_tiles(y4_handle h, int32_t* tiles, int cap) {
    if (int r = check_handle(h)) return r;
    Y4_REQUIRE(tiles && cap >= (int)h->layers.size(), Y4_EINVAL, "y4_get_tiles: need room for %d layers", (int)h->layers.size());
    for (int i = 0; i < (int)h->layers.size(); ++i) tiles[i] = 0;
    for (const Op& op : h->ops)
        if (op.kind == OP_CONV) {
            tiles[op.conv] = op.tile;
            if (op.conv2 >= 0) tiles[op.conv2] = op.tile;
        }
    if (h->fuse_chains)                       // a chained run reports its head as -(run tile + 1000 * the conv's stand-alone tile), so
        for (const Chain& ch : h->chains)     // that a get -> set round trip loses neither (see y4_set_tiles)
            if (ch.enabled) tiles[h->ops[ch.head].conv] = -(ch.tile + 1000 * h->ops[ch.head].tile);
    return Y4_OK;
}

// Every scheduling choice of `src` -> `dst` (a sibling built from the same configuration): the per-op tiles, every run's
// enabled / tile state INCLUDING the stand-alone tile of a conv that currently heads a run (y4_get_tiles reports such a conv as
// -run_tile and cannot carry both), the stage-kernel and residual-block verdicts, the fusion switches and sub-batching.
int y4_copy_schedule(y4_handle src, y4_handle dst) {
    if (int r = check_handle(src)) return r;
    if (int r = check_handle(dst)) return r;
    Y4_REQUIRE(src != dst, Y4_EINVAL, "y4_copy_schedule: source and destination are the same handle");
    Y4_REQUIRE(dst->t_max_steps == 0, Y4_ESTATE, "y4_copy_schedule: a timing session is open on the destination");
    Y4_REQUIRE(src->ops.size() == dst->ops.size() && src->chains.size() == dst->chains.size() &&
               src->resruns.size() == dst->resruns.size() && src->cfg.dtype == dst->cfg.dtype && src->S == dst->S &&
               src->cfg.num_classes == dst->cfg.num_classes && src->cfg.max_batch == dst->cfg.max_batch, Y4_EINVAL,
               "y4_copy_schedule: the handles were not created from the same configuration");
    Y4_REQUIRE(src->sub_images <= 0 || !dst->alias_bufs, Y4_ESTATE,
               "y4_copy_schedule: the source runs sub-batches, the destination's workspace is aliased");
    for (size_t i = 0; i < src->ops.size(); ++i) dst->ops[i].tile = src->ops[i].tile;
    for (size_t i = 0; i < src->chains.size(); ++i) {
        dst->chains[i].enabled = src->chains[i].enabled;
        dst->chains[i].tile = src->chains[i].tile;
    }
    dst->fuse_stem = src->fuse_stem; dst->fuse_chains = src->fuse_chains;
    dst->stage_on = src->stage_on; dst->stage_enabled = src->stage_enabled;
    dst->res_on = src->res_on; dst->res_enabled[0] = src->res_enabled[0]; dst->res_enabled[1] = src->res_enabled[1];
    dst->sub_images = src->sub_images; dst->sub_last_op = src->sub_last_op;
    dst->allow_halo2 = src->allow_halo2;
    return Y4_OK;
}

// does op `oi` launch a kernel of its own under the current fusion settings?  (mirrors run_op's skip logic)
static bool op_launches(y4_handle h, int oi) {
    const Op& op = h->ops[oi];
    if (op.kind != OP_CONV) return true;
    if (h->fuse_stem && op.conv == 1) return false;
    {
        bool is_head = false;
        if (h->res_of(oi, &is_head)) return is_head;
    }
    if (h->stage_active() && oi >= h->stage_first && oi <= h->stage_last) return oi == h->stage_first;
    if (h->fuse_chains)
        for (const Chain& ch : h->chains)
            if (h->chain_active(ch) && (ch.tail[0] == oi || ch.tail[1] == oi)) return false;
    return true;
}

int y4_launch_counts(y4_handle h, int32_t* conv_family, int32_t* total) {
    if (int r = check_handle(h)) return r;
    int convs = 0, all = 2;                                  // decode + nms
    for (int oi = 0; oi < (int)h->ops.size(); ++oi) {
        if (!op_launches(h, oi)) continue;
        ++all;
        if (h->ops[oi].kind == OP_CONV) ++convs;
    }
    if (conv_family) *conv_family = convs;
    if (total) *total = all;
    return Y4_OK;
}

int y4_timing_begin(y4_handle h, int max_steps, int coarse) {
    if (int r = check_handle(h)) return r;
    Y4_REQUIRE(max_steps >= 1 && max_steps <= 4096, Y4_EINVAL, "y4_timing_begin: max_steps %d", max_steps);
    Y4_REQUIRE(h->t_max_steps == 0, Y4_ESTATE, "a timing session is already open");
    h->t_coarse = coarse != 0;
    {
        std::vector<Launch> sched;
        build_schedule(h, h->cfg.max_batch, sched);
        h->t_per_step = (int)sched.size() + 3;
    }
    h->t_events.resize((size_t)max_steps * h->t_per_step);
    for (auto& e : h->t_events) Y4_CHECK_HIP(hipEventCreate(&e));
    h->t_max_steps = max_steps;
    h->t_steps = 0;
    return Y4_OK;
}

int y4_timing_end(y4_handle h, float* op_ms_mean, char* names, int cap, int* n_ops, int* steps, void* stream) {
    if (int r = check_handle(h)) return r;
    Y4_REQUIRE(h->t_max_steps > 0, Y4_ESTATE, "no timing session is open");
    const int nops = (int)h->ops.size() + 2;
    int rc = Y4_OK;
    if (!op_ms_mean || !n_ops || cap < nops) {
        set_error("y4_timing_end: need room for %d ops", nops);
        rc = Y4_EINVAL;
    }
    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess && rc == Y4_OK) {
        set_error("y4_timing_end: stream synchronize failed");
        rc = Y4_EHIP;
    }
    if (rc == Y4_OK) {
        std::vector<double> acc(nops, 0.0);
        for (int st = 0; st < h->t_steps; ++st) {
            hipEvent_t* ev = h->t_events.data() + (size_t)st * h->t_per_step;
            for (int k = 0; k < (int)h->t_slot_op.size(); ++k) {
                float ms = 0.f;
                if (hipEventElapsedTime(&ms, ev[k], ev[k + 1]) != hipSuccess) { rc = Y4_EHIP; set_error("hipEventElapsedTime failed"); }
                acc[h->t_slot_op[k]] += ms;          // a sub-batched op has several slots per step
            }
        }
        for (int j = 0; j < nops; ++j) {
            op_ms_mean[j] = h->t_steps ? (float)(acc[j] / h->t_steps) : 0.f;
            if (names) {
                memset(names + 16 * j, 0, 16);
                const char* nm = j < (int)h->ops.size() ? h->ops[j].name : (j == (int)h->ops.size() ? "decode" : "nms");
                strncpy(names + 16 * j, nm, 15);
            }
        }
        *n_ops = nops;
        if (steps) *steps = h->t_steps;
    }
    for (auto& e : h->t_events) (void)hipEventDestroy(e);
    h->t_events.clear();
    h->t_max_steps = 0;
    h->t_steps = 0;
    return rc;
}

// One forward + decode + NMS with per-op HIP-event timing (synchronises): a one-step timing session.
int y4_profile(y4_handle h, const float* imgs, int n, float* op_ms, char* names, int cap, int* n_ops, void* stream) {
    if (int r = check_ready(h, n)) return r;
    Y4_REQUIRE(imgs && op_ms && n_ops, Y4_EINVAL, "y4_profile: null argument");
    if (int r = y4_timing_begin(h, 1, 0)) return r;
    float* boxes = (float*)(h->act + h->scratch_off);
    float* scores = boxes + (size_t)n * h->cfg.max_total * 4;
    float* classes = scores + (size_t)n * h->cfg.max_total;
    int32_t* kept = (int32_t*)(classes + (size_t)n * h->cfg.max_total);
    int32_t* valid = kept + (size_t)n * h->cfg.max_total;
    const int rc = y4_predict(h, imgs, n, boxes, scores, classes, valid, kept, stream);
    const int rc2 = y4_timing_end(h, op_ms, names, cap, n_ops, nullptr, stream);
    return rc ? rc : rc2;
}

// ---------------------------------------------------------------- standalone operators
int y4_packed_conv_bytes(int dtype, int cout, int cin, int ksize, int32_t* cout_pad, size_t* bytes) {
    Y4_REQUIRE(dtype >= Y4_F32 && dtype <= Y4_F16 && cout > 0 && cin > 0 && (ksize == 1 || ksize == 3), Y4_EINVAL,
               "y4_packed_conv_bytes: bad argument");
    const int cp = (int)round_up(cout, COUT_PAD);
    if (cout_pad) *cout_pad = cp;
    if (bytes) *bytes = (size_t)cp * ksize * ksize * cin * elem_size(dtype);
    return Y4_OK;
}

int y4_pack_conv_weights(int dtype, int cout, int cin, int ksize, const float* oihw_dev, void* packed_dev, void* stream) {
    Y4_REQUIRE(oihw_dev && packed_dev, Y4_EINVAL, "y4_pack_conv_weights: null pointer");
    return pack_conv_weights(dtype, cout, cin, ksize, oihw_dev, packed_dev, (hipStream_t)stream);
}

int y4_pack_conv_frag32(int dtype, int cout, int cin, const void* packed_dev, void* frag_dev, void* stream) {
    Y4_REQUIRE(packed_dev && frag_dev, Y4_EINVAL, "y4_pack_conv_frag32: null pointer");
    return pack_conv_frag32(dtype, cout, cin, packed_dev, frag_dev, (hipStream_t)stream);
}

static char* g_zero_page = nullptr;
int y4_conv2d(const y4_conv_desc* d, void* stream) {
    if (!g_zero_page) {
        Y4_CHECK_HIP(hipMalloc((void**)&g_zero_page, ZERO_PAGE_BYTES));
        Y4_CHECK_HIP(hipMemset(g_zero_page, 0, ZERO_PAGE_BYTES));
    }
    return conv2d_launch(d, g_zero_page, (hipStream_t)stream);
}
int y4_conv_tile_count(void) { return conv_tile_count(); }

int y4_conv_tile_desc(int tile, int32_t cfg[6]) {
    Y4_REQUIRE(cfg && tile >= 1 && tile <= conv_tile_count(), Y4_EINVAL, "y4_conv_tile_desc: tile %d", tile);
    const TileCfg& t = kTiles[tile - 1];
    cfg[0] = t.bm; cfg[1] = t.bn; cfg[2] = t.wm; cfg[3] = t.wn; cfg[4] = t.bkb; cfg[5] = t.nst;
    return Y4_OK;
}

int y4_pack_stem_weights(const float* w_oihw_dev, float* wk_dev, int cout, void* stream) {
    return pack_stem_weights(w_oihw_dev, wk_dev, cout, (hipStream_t)stream);
}

int y4_stem_conv(int dtype, const float* imgs_dev, int n, int h, int w, const float* wk_dev, const float* scale,
                 const float* shift, int cout, int act, void* out_dev, int out_cstride, int out_coff, void* stream) {
    return stem_conv_launch(dtype, imgs_dev, 0, n, h, w, wk_dev, scale, shift, cout, act, out_dev, out_cstride,
                            out_coff, (hipStream_t)stream);
}

int y4_resize_u8(const uint8_t* imgs_dev, int n, int h, int w, uint8_t* out_dev, int out_h, int out_w, void* stream) {
    return resize_u8_launch(imgs_dev, n, h, w, out_dev, out_h, out_w, (hipStream_t)stream);
}

int y4_preprocess_u8(const uint8_t* img_dev, int h, int w, float* out_dev, int out_h, int out_w, void* stream) {
    return preprocess_u8_launch(img_dev, h, w, out_dev, out_h, out_w, (hipStream_t)stream);
}

int y4_spp(int dtype, void* buf_dev, int n, int side, int c, void* stream) {
    return spp_launch(dtype, buf_dev, n, side, c, (hipStream_t)stream);
}

}  // extern "C"
