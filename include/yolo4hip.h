/* yolo4hip.h -- C ABI of libyolo4hip.so: the MI355X (gfx950) YOLOv4 inference hot path.
 *
 * The reference (taipingeric/yolo-v4-tf.keras) has no FFI: its hot path is the tf.keras graph executed by
 * `Model.predict` (reference models.py:113,159,514).  This ABI is the seam a host binds instead of
 * TensorFlow; each entry point cites the reference code it stands in for.  See INTEGRATION.md for the
 * ctypes binding that `yolo-v4-tf.keras_amd/yolo4hip/ext.py` uses.
 *
 * Conventions
 *   - every function returns 0 on success or a negative Y4_E* code; `y4_last_error()` gives the text
 *     (thread-local).  Nothing throws, nothing aborts.
 *   - all `*_dev` / workspace pointers are DEVICE pointers owned by the caller (the Python host owns them
 *     as torch-ROCm tensors).  A handle never allocates device memory: scratch comes out of the bound
 *     workspace.  (Only the standalone y4_conv2d lazily allocates one 256-byte zero page per process.)
 *   - every launch goes on the caller's `stream` (a hipStream_t passed as void*; NULL = default stream).
 *     Calls are asynchronous; the caller synchronises.  All calls on ONE handle must be enqueued in stream order (on one
 *     stream, or on streams the caller orders with events): a handle keeps host-side notes of what its workspace holds
 *     (which head wrote the objectness side array for how many images, whether the decode counters are clean) that are
 *     updated when a call is ENQUEUED, after its launches succeeded -- not when it completes.
 *   - activations are NHWC.  Images are float32 [n, H, W, 3] in [0,1] (what `Yolov4.preprocess_img`
 *     produces, reference models.py:95-98, after Keras' cast to float32).
 *   - one handle per process/GPU; a handle is not re-entrant.
 */
#ifndef YOLO4HIP_H
#define YOLO4HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define Y4_OK 0
#define Y4_EINVAL (-22)      /* bad argument / unsupported shape */
#define Y4_ENOMEM (-12)      /* bound workspace too small */
#define Y4_ESTATE (-1)       /* call order (workspace not bound, weights not packed) */
#define Y4_EHIP (-5)         /* a HIP runtime call failed */

#define Y4_F32 0
#define Y4_BF16 1
#define Y4_F16 2

#define Y4_ACT_LINEAR 0
#define Y4_ACT_LEAKY 1       /* LeakyReLU(alpha=0.1), reference custom_layers.py:29-30 */
#define Y4_ACT_MISH 2        /* x*tanh(softplus(x)),  reference custom_layers.py:6-7  */

typedef struct y4_ctx* y4_handle;

/* Mirrors what `Yolov4.__init__` reads from `yolo_config` (reference models.py:26-37, config.py:1-17). */
typedef struct y4_config {
    int32_t img_size;          /* square input side, multiple of 32 (reference models.py:23-24) */
    int32_t num_classes;       /* len(class_names) (reference models.py:25,27) */
    int32_t max_batch;         /* largest n any call will pass */
    int32_t dtype;             /* Y4_F32 | Y4_BF16 | Y4_F16: storage/MFMA-input type; accumulation is fp32 */
    float anchors[18];         /* 3 scales x 3 anchors x (w,h) px at input resolution (config.py:4, models.py:29) */
    float xyscale[3];          /* config.py:6 */
    int32_t strides[3];        /* config.py:5 */
    float iou_threshold;       /* config.py:15 */
    float score_threshold;     /* config.py:16 */
    int32_t max_per_class;     /* 100: custom_layers.py:293 */
    int32_t max_total;         /* 100: custom_layers.py:294 */
} y4_config;

/* One row of the 110-conv plan, as built by the C++ runtime. */
typedef struct y4_layer_desc {
    int32_t idx, ksize, stride, cin, cout, act, has_bn, in_side, out_side;
    int64_t weight_offset;     /* float offset of this layer's record in the Darknet stream (bn/bias first) */
} y4_layer_desc;

const char* y4_last_error(void);
const char* y4_version(void);

/* Replaces Yolov4.__init__ -> build_model (inference half), reference models.py:18-52,67-73: builds the
 * 110-conv CSPDarknet53+SPP+PANet plan, the decode and the NMS stages for this config.  Host-only. */
int y4_create(const y4_config* cfg, y4_handle* out);
int y4_destroy(y4_handle h);

int y4_num_layers(y4_handle h);
int y4_layer_info(y4_handle h, int idx, y4_layer_desc* out);
/* per image: conv FLOPs (2*k*k*cin*cout*ho*wo summed), decoded boxes, padded channel count of a raw head */
int y4_model_info(y4_handle h, int64_t* flops_per_image, int32_t* num_boxes, int32_t* head_cstride,
                  int64_t* weight_floats);

/* Optional, BEFORE y4_workspace_bytes / y4_bind_workspace: let activation buffers whose lifetimes do not overlap share memory
 * (liveness over the op order, every fusable group of ops counted as one instant).  The activation workspace shrinks to about a
 * quarter and the working set the Infinity Cache sees becomes hotter; the price: intermediate tensors are not retained after a
 * forward (y4_get_conv_output then returns Y4_ESTATE) and sub-batching is refused.  Results are unchanged.  (Stands where
 * TensorFlow's memory planner stands in the reference; not in tree.) */
int y4_set_workspace_aliasing(y4_handle h, int on);
/* Device memory the caller must provide: `act` = activations + decode/NMS scratch for max_batch images,
 * `wts` = packed weights (+ per-channel scale/shift). */
int y4_workspace_bytes(y4_handle h, size_t* act_bytes, size_t* wts_bytes);
int y4_bind_workspace(y4_handle h, void* act_dev, size_t act_bytes, void* wts_dev, size_t wts_bytes);

/* Replaces utils.load_weights (reference utils.py:12-53) + Keras set_weights: `darknet_floats_dev` is the
 * float32 stream of a Darknet .weights file after its 20-byte header, already on the device: for conv
 * 0..109, [beta,gamma,mean,var] x cout (or cout biases for convs 93/101/109), then cout*cin*k*k weights in
 * (out,in,h,w) order.  Computes scale = gamma*rsqrt(var+1e-3), shift = beta-mean*scale (Keras BN eps) and
 * re-lays every kernel along the library's canonical K order -- [cout_pad][cin/KC][kh*kw][KC], KC = min(cin, 64) for 3x3 and cin for 1x1
 * kernels: 64-channel chunk, then tap, then channel -- in the handle's dtype inside the bound `wts` workspace. */
int y4_pack_weights(y4_handle h, const float* darknet_floats_dev, size_t n_floats, void* stream);
/* Multi-GPU: after rank 0 packed and the caller broadcast the whole `wts` workspace (RCCL), the other
 * ranks mark their copy as valid. */
int y4_adopt_packed_weights(y4_handle h);

/* Replaces yolo_model.predict(imgs) (reference models.py:50-52,514; graph custom_layers.py:100-198).
 * Raw heads stay inside the workspace (padded to head_cstride channels). */
int y4_forward(y4_handle h, const float* imgs_nhwc_dev, int n, void* stream);
/* The same on uint8 frames [n, H, W, 3] ALREADY at network size, BEFORE the `/ 255.` of Yolov4.preprocess_img (reference
 * models.py:95-98): the stem applies it inside its operand load, so no float image tensor exists and the frames cross
 * PCIe / HBM at 3 B per pixel.  Bit-identical to y4_forward on float32(double(v) / 255.) for every dtype.
 * Frames of another size go through y4_resize_u8 (cv2.resize's uint8 INTER_LINEAR arithmetic) first. */
int y4_forward_u8(y4_handle h, const uint8_t* imgs_nhwc_u8_dev, int n, void* stream);
/* y4_forward cut behind conv `last_conv` (0-based, the reference's creation order): the ops up to and including the launch
 * that computes it.  last_conv = 71 is `cspdarknet53` proper -- the five CSP stages, reference custom_layers.py:100-124, before the
 * SPP block's convs -- which bench.py times on its own (`backbone` in its line).  Y4_EINVAL when the conv does not end a launch
 * under the current fusion settings (a run that continues behind it is never cut). */
int y4_forward_until(y4_handle h, const float* imgs_nhwc_dev, int n, int last_conv, void* stream);
/* Dense float32 copies of the three raw heads, [n,g,g,3*(C+5)] each, as Keras returns them. */
int y4_get_heads(y4_handle h, int n, float* out_s_dev, float* out_m_dev, float* out_l_dev, void* stream);
/* Inverse of y4_get_heads: load dense float32 raw heads [n,g,g,3*(C+5)] into the workspace, so that
 * y4_decode_nms can be driven with arbitrary logits (decode/NMS known-answer tests; reference
 * predict_nonms feeds yolov4_head/nms with precomputed heads the same way, models.py:521-523). */
int y4_set_heads(y4_handle h, int n, const float* in_s_dev, const float* in_m_dev, const float* in_l_dev,
                 void* stream);
/* Debug/parity tap: dense float32 NHWC copy of conv `idx`'s output tensor (post BN/activation/residual;
 * for convs that write a concat slice, that slice). */
int y4_get_conv_output(y4_handle h, int conv_idx, int n, float* out_dev, size_t out_floats, void* stream);

/* Replaces yolov4_head/get_boxes + nms (reference custom_layers.py:201-298, i.e.
 * tf.image.combined_non_max_suppression) on the heads left by y4_forward.
 * boxes [n,max_total,4] (x1,y1,x2,y2 / img_size, clipped to [0,1], zero padded), scores [n,max_total],
 * classes [n,max_total] (class id as float), valid [n] int32, kept_idx [n,max_total] int32 (box index
 * n = scale_offset + (row*g+col)*3 + anchor, -1 padded; may be NULL).  iou/score thresholds < 0 mean
 * "use the config's" (predict_nonms passes its own, reference models.py:516-523). */
int y4_decode_nms(y4_handle h, int n, float iou_threshold, float score_threshold, float* boxes_dev,
                  float* scores_dev, float* classes_dev, int32_t* valid_dev, int32_t* kept_idx_dev,
                  void* stream);
/* Replaces inference_model.predict(imgs) (reference models.py:69-73,113,159) = forward + decode + NMS. */
int y4_predict(y4_handle h, const float* imgs_nhwc_dev, int n, float* boxes_dev, float* scores_dev,
               float* classes_dev, int32_t* valid_dev, int32_t* kept_idx_dev, void* stream);

int y4_predict_u8(y4_handle h, const uint8_t* imgs_nhwc_u8_dev, int n, float* boxes_dev, float* scores_dev,
                  float* classes_dev, int32_t* valid_dev, int32_t* kept_idx_dev, void* stream);

/* Per-op device time of one forward (+decode+NMS) in ms, measured with HIP events on `stream`
 * (synchronises).  `names` receives op names ('c17', 'spp', 'decode', 'nms', ...), 16 bytes each. */
int y4_profile(y4_handle h, const float* imgs_nhwc_dev, int n, float* op_ms, char* names, int cap,
               int* n_ops, void* stream);

/* Measured per-layer tile selection for batch n (call once after y4_pack_weights; optional).  Every tile
 * configuration gives bit-identical outputs, so this affects speed only.  y4_get_tiles reports the choice
 * per conv index (0 = built-in heuristic). */
int y4_autotune(y4_handle h, int n, int reps, void* stream);
/* The same with THROUGHPUT as objective, for two batches in flight: `h2` is a second handle of the same plan (own activation
 * workspace, may share the packed-weight workspace) that will run on `stream2` beside `h` on `stream`.  Every timed
 * launch is issued on both handles and the time until both streams are done counts, so a tile whose last round leaves
 * compute units idle is not charged for them (the neighbour stream fills them): the least WORK wins, not the shortest solitary
 * launch.  `pair_passes` says which decisions use that objective (the others are taken one launch at a time, as y4_autotune
 * does): bit 0 the tile of every conv, bit 1 chains / LDS pairs fused or separate, bit 2 the stage kernel, bit 3 the
 * residual-block kernels.  Both handles end up with the same choices.  (No reference counterpart: TensorFlow's executor
 * schedules the reference's graph; models.py:113,159.) */
int y4_autotune_pair(y4_handle h, y4_handle h2, int n, int reps, void* stream, void* stream2, int pair_passes);
int y4_get_tiles(y4_handle h, int32_t* tiles, int cap);
/* Restore a tile choice saved from y4_get_tiles (one entry per conv index; an id that does not fit its layer makes
 * the next forward fail with Y4_EINVAL rather than compute anything different).  The head conv of a fused run (see
 * y4_set_chain_fusion) carries two choices in its entry: -(run tile + 1000 * the conv's own tile for when the run is not in
 * force); a plain -t (older files) leaves the own tile as it is, so y4_set_tiles(y4_get_tiles()) restores a handle exactly. */
int y4_set_tiles(y4_handle h, const int32_t* tiles, int count);
/* Every scheduling choice of `src` -> `dst`, a second handle created from the same configuration (the sibling that runs a second
 * batch in flight): tiles, each run's state, stage-kernel and residual-block verdicts, the fusion switches, sub-batching.
 * (No reference counterpart: TensorFlow's executor schedules the reference's graph; models.py:113,159.) */
int y4_copy_schedule(y4_handle src, y4_handle dst);

/* Scheduling knob (results unchanged): run the ops up to and including conv `last_conv` over `images` images at a
 * time instead of the whole batch, so that the large early activations of a sub-batch are still resident in the
 * 256 MiB Infinity Cache when their consumer runs.  images <= 0 turns it off. */
int y4_set_subbatch(y4_handle h, int images, int last_conv);

/* Scheduling knob (16-bit dtypes, img_size <= 640; results unchanged): run convs 0 and 1 (reference
 * custom_layers.py:101-102) as one kernel that keeps conv 0's output -- the largest tensor of the network -- in LDS
 * instead of writing it to HBM and reading it back.  While on, y4_get_conv_output(0) fails with Y4_ESTATE and the
 * profile reports the pair under 'c0' ('c1' reads 0).  Y4_EINVAL if the dtype / size is not supported. */
int y4_set_stem_fusion(y4_handle h, int on);

/* Scheduling knob (16-bit dtypes): run each "3x3 conv + residual Add -> 1x1 conv [-> 1x1 conv over
 * Concatenate([., route])]" run of the CSP stages with 64-channel blocks (reference custom_layers.py:41-44, :66-69
 * and the conv that follows csp_block, :104/:109) as ONE kernel: the intermediate tensors stay in registers instead
 * of going through HBM; likewise "3x3 conv + Add -> the next block's 1x1 conv" of the 128- and 256-channel stages
 * through an LDS-resident tile.  A chained conv issues the same MFMAs on the same inputs in the same order as its own
 * kernel: results are bit-identical.  Returns the number of fusable runs (>= 0) when turned on, Y4_OK when turned off,
 * a negative Y4_E* code on error.  While on, y4_get_conv_output of a conv inside a run (not its last) reads a
 * tensor that is not materialised.  y4_autotune then also decides per run, by measurement, whether it executes as
 * one kernel or as separate ones; y4_get_tiles reports a fused run's head conv as MINUS its tile id (y4_set_tiles
 * accepts the same encoding; > 0 there means separate kernels, 0 fused with the built-in tile).  One run is an ALTERNATIVE:
 * "1x1 conv 64 -> 64 -> 1x1 conv over Concatenate([., route])" (custom_layers.py:66-69) is in force only while the three-conv run
 * it is the tail of cannot exist because that run's 3x3 conv executes inside a residual-block kernel (y4_set_res_fusion). */
int y4_set_chain_fusion(y4_handle h, int on);

/* Scheduling knob (16-bit dtypes; results unchanged): run the whole first CSP stage -- convs 2..7, reference
 * custom_layers.py:47-69 (csp_block with residual_bottleneck=True) and the transition conv :105 -- as ONE spatially tiled
 * kernel (csrc/csp_stage.hip): per 16x16-pixel tile the route / main-in / bottleneck / 3x3+Add / main-out / transition
 * convs run from LDS and registers, so only conv 1's output is read and conv 7's written.  Every conv issues the same
 * MFMAs on the same 16-bit inputs in the same order as its own kernel: bit-identical.  Returns 1 when the stage kernel
 * is active, 0 when off, a negative Y4_E* code on error (Y4_EINVAL for fp32).  y4_autotune afterwards keeps it only if
 * it measures faster than the separately tuned kernels (y4_get_stage_fusion tells).  While active,
 * y4_get_conv_output of convs 2..6 fails with Y4_ESTATE (not materialised). */
int y4_set_stage_fusion(y4_handle h, int on);
int y4_get_stage_fusion(y4_handle h);

/* Scheduling knob (16-bit dtypes; results unchanged): run every residual block "1x1 conv -> 3x3 conv + Add" of the 64- and
 * 128-channel CSP stages (reference custom_layers.py:34-44; the 152^2 and 76^2 stages at 608x608) as ONE spatially tiled
 * kernel (csrc/resblock.hip): the halo'd 18x18-pixel tile of the block input is brought into LDS once, the 1x1 conv runs
 * on it in place and the 3x3 conv reads all nine taps from that tile, streaming only its weights.  Bit-identical to the
 * separate kernels.  Returns the number of such blocks (>= 0) when turned on, Y4_OK when turned off, < 0 on error.
 * y4_autotune afterwards keeps it per channel group only where it measures faster; y4_get_res_fusion reports the groups
 * in use as a bit mask (1: 128 channels, 2: 64 channels) and y4_set_res_fusion_mask restores such a choice.  While a
 * block runs fused, y4_get_conv_output of its 1x1 conv fails with Y4_ESTATE (not materialised). */
int y4_set_res_fusion(y4_handle h, int on);
int y4_get_res_fusion(y4_handle h);
int y4_set_res_fusion_mask(y4_handle h, int mask);

/* Kernel launches of one y4_predict under the current fusion / tile settings (whole batch, no sub-batching):
 * `conv_family` = launches of the conv kernels other than the stem (what bench.py's roofline is quoted on),
 * `total` = all launches including stem, SPP, decode and NMS. */
int y4_launch_counts(y4_handle h, int32_t* conv_family, int32_t* total);

/* Live per-op timing of the calls in between: while a session is open, each y4_predict (up to max_steps of
 * them) records a HIP event on its stream after every op, without synchronising.  y4_timing_end
 * synchronises the stream and returns the mean device time per op in ms ('c1'.., 'spp', 'decode', 'nms').
 * coarse != 0: events only where the op kind changes (stem | run of convs | spp | run of convs | decode | nms);
 * each run's time is reported under its first op, the others read 0 -- 7 events per step instead of 115. */
int y4_timing_begin(y4_handle h, int max_steps, int coarse);
int y4_timing_end(y4_handle h, float* op_ms_mean, char* names, int cap, int* n_ops, int* steps_recorded,
                  void* stream);

/* ---- standalone operators (same kernels as the plan uses; for unit tests and other hosts) ---- */

typedef struct y4_conv_desc {
    int32_t dtype;                 /* Y4_* of in/weights/out/res */
    int32_t n, h, w, cin;          /* input NHWC view: [n,h,w,in_cstride][..., in_coff:in_coff+cin] */
    int32_t cout, ksize, stride;   /* ksize 1|3; stride 1 ('same') | 2 (pad top/left 1, 'valid') */
    int32_t act;                   /* Y4_ACT_* */
    int32_t upsample;              /* 1: write each output pixel to its 2x2 nearest-upsampled block */
    int32_t out_f32;               /* 1: output buffer is float32 regardless of dtype */
    int32_t in_cstride, in_coff;
    int32_t out_cstride, out_coff;
    int32_t res_cstride, res_coff; /* residual view (same spatial dims as the output), used if res != NULL */
    const void* in;
    const void* wt;                /* packed by y4_pack_conv_weights */
    const float* scale;            /* [cout_pad] */
    const float* shift;            /* [cout_pad] */
    const void* res;
    void* out;
    int32_t tile;                  /* 0 = auto; otherwise a tile-config id (see y4_conv_tile_count) */
    /* optional second output view: channels [split, cout) are stored to out2 (channel c -> out2_coff + c - split).
     * Used to run a CSP block's route conv and main-in conv (same input, custom_layers.py:58-60) as ONE GEMM. */
    void* out2;
    int32_t out2_cstride, out2_coff, split;
    /* split-K (tile = base + 100 e: the base tile's K loop split 2^e ways, e = 1..3; for launches with fewer tiles than compute
     * units): device scratch of 16 KiB of tile counters, ZERO before the first use (every launch leaves them zero), followed by
     * 2^e x tiles x BM x BN floats of partial sums.  NULL / 0 when no split tile is used. */
    void* splitk_ws;
    size_t splitk_ws_bytes;
    /* halo2 tiles (schedule code 21: one wave per SIMD, v_mfma_32x32x16, weights read into registers): the same weights once more in
     * MFMA-fragment order, made from `wt` by y4_pack_conv_frag32.  NULL for every other tile. */
    const void* wt_frag;
} y4_conv_desc;

/* cout_pad (rows of the packed matrix) and bytes needed for a packed kernel */
int y4_packed_conv_bytes(int dtype, int cout, int cin, int ksize, int32_t* cout_pad, size_t* bytes);
/* Darknet (cout,cin,k,k) float32 on device -> packed [cout_pad][cin/KC][k*k][KC] dtype (KC = min(cin, 64) for k = 3, cin for k = 1: the
 * K order every conv kernel sums in); rows >= cout are zero */
int y4_pack_conv_weights(int dtype, int cout, int cin, int ksize, const float* oihw_dev, void* packed_dev,
                         void* stream);
/* 3x3 conv weights packed by y4_pack_conv_weights (16-bit dtypes, cin % 64 == 0) -> the layout the halo2 tiles read:
 * [cout_pad / 32 channel blocks][cin / 64 chunks][9 taps][4 k-steps of 16 channels][64 lanes][8 elements], lane l of a k-step = row
 * (l & 31) of the block in the accumulator layout's channel order, input channels 16 s + 8 (l >> 5) .. +7 of the chunk: one k-step's
 * v_mfma_f32_32x32x16 A operand of a block is 1 KB contiguous.  Same size as the packed matrix (y4_packed_conv_bytes). */
int y4_pack_conv_frag32(int dtype, int cout, int cin, const void* packed_dev, void* frag_dev, void* stream);
/* One conv() unit of the reference (custom_layers.py:5-31) + optional Add (custom_layers.py:44) +
 * optional UpSampling2D (custom_layers.py:147,159) + concat-slice store (custom_layers.py:68,...). */
int y4_conv2d(const y4_conv_desc* d, void* stream);
int y4_conv_tile_count(void);
/* Tile configuration `tile` (1 .. y4_conv_tile_count()): cfg = {BM pixels, BN channels, waves over pixels, waves over
 * channels, bytes of K per LDS row, schedule code (2..7 = ring stages, 12 = staggered 2-stage, 32 = 2-stage with the 32x32x16
 * MFMA)}.  All tiles with the 16x16x32 MFMA give bit-identical results; the 32x32x16 tiles agree among themselves. */
int y4_conv_tile_desc(int tile, int32_t cfg[6]);
/* Latency schedules (the reference's own call is one image: Yolov4.predict, models.py:109-127).  With few images the deep layers
 * have fewer output tiles than the GPU has compute units and each tile walks a long K loop alone; `on` lets y4_autotune also offer
 * split-K tile ids (base + 100 e: the K loop of a tile split over 2^e workgroups, the last one to finish adds the partial sums in
 * split order and runs the epilogue).  A split launch sums in another fp32 order than the unsplit one, so with this switch the
 * tuned schedule is part of the numerical result (like the 32x32x16 tiles; tested against the oracle); off by default. */
int y4_set_splitk(y4_handle h, int on);
/* The halo2 tiles (schedule code 21, conv_halo2_kernel.h: the 3x3 stride-1 convs of custom_layers.py:14-24 as a one-wave-per-SIMD kernel on
 * v_mfma_f32_32x32x16 with the weights read into registers in MFMA-fragment order) sum the K axis in k-steps of 16 instead of 32:
 * another fixed fp32 order than the 16x16x32 tiles.  `on` lets y4_autotune offer them too; the tuned schedule then is part of the
 * numerical result, exactly as with y4_set_splitk (tested against the oracle and for run-to-run determinism); off by default. */
int y4_set_halo2(y4_handle h, int on);
/* Stem conv (cin = 3, reference custom_layers.py:101): float32 images -> dtype.  `wk_dev` is an 8192-byte table
 * made by y4_pack_stem_weights from Darknet (cout,3,3,3) order: float32 [(ky*3+kx)*3+ci][cout] for the fp32
 * kernel, then (byte 4096 / 6144) the bf16 / fp16 MFMA weight fragments used by the 16-bit kernels. */
int y4_pack_stem_weights(const float* w_oihw_dev, float* wk_dev, int cout, void* stream);
int y4_stem_conv(int dtype, const float* imgs_dev, int n, int h, int w, const float* wk_dev,
                 const float* scale, const float* shift, int cout, int act, void* out_dev, int out_cstride,
                 int out_coff, void* stream);
/* Device-side Yolov4.preprocess_img (reference models.py:95-98: cv2.resize INTER_LINEAR stretch, then /255.):
 * uint8 RGB image [h,w,3] -> float32 [out_h,out_w,3] in [0,1], one image slot of the batch tensor y4_forward takes.
 * Saves the 4x fatter float32 host->device copy and the host-side float64 tensor. */
int y4_preprocess_u8(const uint8_t* img_dev, int h, int w, float* out_dev, int out_h, int out_w, void* stream);
/* The resize half of preprocess_img alone, batched: uint8 [n,h,w,3] -> uint8 [n,out_h,out_w,3] with cv2.resize's
 * uint8 INTER_LINEAR fixed-point arithmetic (what cv2.resize itself returns for a uint8 image). */
int y4_resize_u8(const uint8_t* imgs_dev, int n, int h, int w, uint8_t* out_dev, int out_h, int out_w, void* stream);
/* SPP (custom_layers.py:130-134): x = buf[..., 3c:4c] -> buf[..., 0:c]=maxpool13, [c:2c]=maxpool9,
 * [2c:3c]=maxpool5 (stride 1, 'same'), buf is [n,side,side,4c] */
int y4_spp(int dtype, void* buf_dev, int n, int side, int c, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* YOLO4HIP_H */
