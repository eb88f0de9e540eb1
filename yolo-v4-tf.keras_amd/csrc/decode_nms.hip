// decode_nms.hip -- the tail of the reference's inference_model (models.py:68-73):
//   yolov4_head/get_boxes (custom_layers.py:201-258)  -> decode_kernel
//   nms() = tf.image.combined_non_max_suppression (custom_layers.py:261-298) -> nms_kernel
//
// decode_kernel: one lane per box reads its objectness logit; boxes whose sigmoid(obj) alone is not above the
// score threshold cannot yield candidates and are dropped at once; the wave then sweeps the class logits of
// the surviving boxes (coalesced), forms score = sigmoid(obj)*sigmoid(cls) and appends the ones with
// score > score_threshold (strict) to the image's candidate list with ONE wave-aggregated atomic per pass
// (ballot + popcount).  Only surviving boxes are decoded to coordinates.  A candidate is a 64-bit key  (score bits << 32) | ~(box_index*C + class): sorting keys
// descending gives (score desc, box index asc, class asc) -- the tie order this build defines.
//
// nms_kernel: one workgroup per image.  Class-aware greedy NMS is done in ONE pass over the globally
// sorted candidates (suppression only against kept boxes of the same class): this visits every class's
// candidates in that class's own descending order, so the per-class kept sets equal TensorFlow's
// per-class greedy runs, and the kept boxes come out already in the final descending-score order; the
// pass stops after max_total kept boxes (a class's 101st kept box can never be in the overall top 100, so
// max_per_class is enforced with a same-class count).  Candidates are taken in chunks in descending key order
// (radix-select of the chunk pivot when an image has more than a chunk holds): the first of at most NMS_THREADS keys
// -- merge-rank sorted, and decided ROUND-PARALLEL, one candidate per thread (round 4; see the greedy pass) --, the
// later ones, which only an image that has not filled max_total from its best 1024 candidates reaches, of at most
// SORT_CAP keys through the register bitonic network and wave 0's pass over 64 candidates at a time.
// IoU is TensorFlow's: corners min/max-normalised, 0 if either area <= 0, suppress iff IoU > threshold.
#include "kernels.h"

#pragma clang fp contract(off)   // keep the reference's float32 op order (no fused multiply-add)

namespace y4 {

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// One lane per (cell, anchor) box.  Because score = sigmoid(obj)*sigmoid(cls) <= sigmoid(obj) in float32
// (sigmoid <= 1), a box whose objectness alone is not > score_threshold cannot produce a candidate: each lane
// reads just its objectness logit first (4 bytes) and the wave then sweeps, cooperatively and coalesced, the
// class logits of the few boxes that pass (ballot loop).  Only those boxes are decoded to coordinates.
__global__ __launch_bounds__(256) void decode_kernel(const DecodeK p) {
    const int lane = threadIdx.x & 63;
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;          // box id over all images
    const int n = (int)(t / p.nbox);
    const int b = (int)(t - (int64_t)n * p.nbox);                        // box index within the image
    const bool in_range = n < p.N;
    const int nf = 5 + p.C;
    int s = 0, a = 0, cell = 0;
    const float* cellp = nullptr;
    float s_obj = 0.f;
    if (in_range) {
        const int c3 = b / 3;
        a = b - c3 * 3;
        const int n0 = p.g[0] * p.g[0], n1 = n0 + p.g[1] * p.g[1];
        s = c3 >= n1 ? 2 : (c3 >= n0 ? 1 : 0);
        cell = c3 - (s == 2 ? n1 : (s == 1 ? n0 : 0));
        const int g = p.g[s];
        cellp = p.head[s] + ((int64_t)n * g * g + cell) * p.hcs + a * nf;
        s_obj = sigmoid_f(cellp[4]);
    }
    unsigned long long todo = __ballot(in_range && s_obj > p.score_thr);
    while (todo) {
        const int j = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        // broadcast lane j's box
        const float* bp = (const float*)__shfl((unsigned long long)(uintptr_t)cellp, j);
        const float so = __shfl(s_obj, j);
        const int bn = __shfl(n, j), bb = __shfl(b, j);
        unsigned long long* keys = p.keys + (int64_t)bn * p.cap;
        for (int c0 = 0; c0 < p.C; c0 += 64) {
            const int c = c0 + lane;
            bool hit = false;
            unsigned long long key = 0;
            if (c < p.C) {
                // cheap screen first (v_exp_f32 + v_rcp_f32, ~1e-6 relative): the exact float32 sigmoid (IEEE
                // expf + division, what the decision is defined on) runs only for scores within 0.2 % of the cut
                const float x = bp[5 + c];
                const float approx = so * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
                const float sc = approx > 0.998f * p.score_thr ? so * sigmoid_f(x) : 0.f;
                if (sc > p.score_thr) {
                    hit = true;
                    const uint32_t id = (uint32_t)bb * (uint32_t)p.C + (uint32_t)c;
                    key = ((unsigned long long)__float_as_uint(sc) << 32) | (unsigned long long)(~id);
                }
            }
            const unsigned long long mask = __ballot(hit);
            if (mask) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(p.counts + (size_t)bn * COUNT_STRIDE, (uint32_t)__popcll(mask));
                base = __shfl(base, 0);
                if (hit) {
                    const uint32_t pos = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
                    if (pos < p.cap) keys[pos] = key;
                }
            }
        }
        if (lane == j) {
            // custom_layers.py:251-256 (only boxes that can be candidates are ever read back by the NMS stage)
            const int g = p.g[s];
            const int row = cell / g, col = cell - row * g;
            const float bx = ((sigmoid_f(cellp[0]) * p.xyscale[s]) - p.xyoff[s] + (float)col) * (float)p.stride[s];
            const float by = ((sigmoid_f(cellp[1]) * p.xyscale[s]) - p.xyoff[s] + (float)row) * (float)p.stride[s];
            const float bw = expf(cellp[2]) * p.anchors[(s * 3 + a) * 2 + 0];
            const float bh = expf(cellp[3]) * p.anchors[(s * 3 + a) * 2 + 1];
            float4 o;
            o.x = (bx - bw / 2.0f) / p.img_size;
            o.y = (by - bh / 2.0f) / p.img_size;
            o.z = (bx + bw / 2.0f) / p.img_size;
            o.w = (by + bh / 2.0f) / p.img_size;
            *(float4*)(p.dboxes + ((int64_t)n * p.nbox + b) * 4) = o;
        }
    }
}

// Variant for cells of at most 256 values (C <= 80): one wavefront per grid cell, the cell's 3*(5+C) logits are
// read fully coalesced straight into registers (<= 4 per lane, one memory round trip), the three objectness
// logits are broadcast by shuffles, and the 73 % of cells where no anchor can pass exit right there.
// (Round 2 also tried 4 cells per wave with all 16 full-cell loads issued up front: 0.123 ms against 0.114 for one cell per
// wave; the objectness pre-screen of decode_cell_kernel below is what removed traffic instead.)

struct DecodeCell { int n, s, rem; const float* src; };
__device__ __forceinline__ DecodeCell decode_locate(const DecodeK& p, int64_t cellid) {
    DecodeCell c;
    c.n = (int)fastdiv((uint32_t)cellid, p.div_cells);
    c.rem = (int)cellid - c.n * p.cells_per_img;
    c.s = 0;
    if (c.rem >= p.g[0] * p.g[0]) { c.rem -= p.g[0] * p.g[0]; c.s = 1; if (c.rem >= p.g[1] * p.g[1]) { c.rem -= p.g[1] * p.g[1]; c.s = 2; } }
    const int g = p.g[c.s];
    c.src = p.head[c.s] + ((int64_t)c.n * g * g + c.rem) * p.hcs;
    return c;
}

// A wave's candidates are staged in its own LDS slice and go out with ONE returning atomic per flush (image change, slice
// nearly full, end of the wave's cells) instead of one per cell: the atomic's round trip -- the wave can do nothing until it
// knows where to write -- was paid 4-5 times per wave, and an image's counter took four times the same-address atomics.
// The order of an image's candidate list is irrelevant (NMS sorts it; keys are unique).
constexpr int DC_STAGE = 512;               // keys per wave slice; a cell yields at most 3 * C <= 240
struct DecodeStage { unsigned long long* slice; uint32_t fill; int img; };
__device__ __forceinline__ void decode_flush(const DecodeK& p, DecodeStage& st, int lane) {
    if (st.fill) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(p.counts + (size_t)st.img * COUNT_STRIDE, st.fill);
        base = __shfl(base, 0);
        unsigned long long* keys = p.keys + (int64_t)st.img * p.cap;
        for (uint32_t i = lane; i < st.fill; i += 64)
            if (base + i < p.cap) keys[base + i] = st.slice[i];
        st.fill = 0;
    }
}

__device__ __forceinline__ void decode_one_cell(const DecodeK& p, const DecodeCell& cl, const float (&v)[4], const float (&so)[3], int lane,
                                                DecodeStage& st) {
    const int n = cl.n, s = cl.s, rem = cl.rem;
    const int g = p.g[s];
    const int nf = 5 + p.C, nval = 3 * nf;
    // so[a] = sigmoid(objectness of anchor a): the screen's value for this cell (the same float32 logit through the same sigmoid_f)
    const int box0 = p.box_off[s] + rem * 3;
    if (n != st.img || st.fill + 3 * p.C > DC_STAGE) { decode_flush(p, st, lane); st.img = n; }     // wave-uniform
    const float cut = 0.998f * p.score_thr;
    unsigned long long key[4];
    unsigned long long mask[4];
    uint32_t total = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int e = lane + 64 * k;
        bool hit = false;
        key[k] = 0;
        if (e < nval) {
            const int a = e >= 2 * nf ? 2 : (e >= nf ? 1 : 0);
            const int f = e - a * nf;
            const float sa = a == 0 ? so[0] : (a == 1 ? so[1] : so[2]);
            if (f >= 5 && sa > p.score_thr) {
                const float approx = sa * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * v[k]));
                if (approx > cut) {
                    const float sc = sa * sigmoid_f(v[k]);
                    if (sc > p.score_thr) {
                        hit = true;
                        const uint32_t id = (uint32_t)(box0 + a) * (uint32_t)p.C + (uint32_t)(f - 5);
                        key[k] = ((unsigned long long)__float_as_uint(sc) << 32) | (unsigned long long)(~id);
                    }
                }
            }
        }
        mask[k] = __ballot(hit);
        total += (uint32_t)__popcll(mask[k]);
    }
    if (total) {                              // into the wave's slice; no atomic here
        uint32_t base = st.fill;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if ((mask[k] >> lane) & 1ull)
                st.slice[base + (uint32_t)__popcll(mask[k] & ((1ull << lane) - 1ull))] = key[k];
            base += (uint32_t)__popcll(mask[k]);
        }
        st.fill = base;
    }
}

// A wave first SCREENS 16 consecutive cells: lane (c, a) = (lane >> 2, lane & 3 < 3) reads the one objectness logit of
// (cell c, anchor a) -- 48 four-byte loads, three 64-byte sectors per cell instead of its 1 KB -- and the ballot of
// sigmoid(obj) > threshold (the same float32 decision decode_one_cell makes) says which cells can have candidates at all
// (27 % on the synthetic heads).  Only those are then read in full, one at a time.  HBM bytes per cell 1024 -> ~470.
// Round 4: DC_SCREEN is a template parameter -- a few images (the reference's own call is ONE) are 1 421 waves of 16 cells on 1 024
// SIMDs, each walking its ~4 flagged cells one after the other with nothing beside it to hide the loads; 4 cells per wave are four
// times the waves with a quarter of the chain (34 -> see LABNOTES.md section 4.4).  The candidate lists' order differs, NMS sorts them.
constexpr int DC_SCREEN = 16, DC_SCREEN_SMALL = 4;
template <int DC_SCREEN>
__global__ __launch_bounds__(256) void decode_cell_kernel(const DecodeK p) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t cell0 = ((int64_t)blockIdx.x * 4 + wave) * DC_SCREEN, ncell = (int64_t)p.N * p.cells_per_img;
    if (cell0 >= ncell) return;
    const int nf = 5 + p.C, nval = 3 * nf;
    const int c = lane >> 2, a = lane & 3;
    bool pass = false;
    float sig = 0.f;                          // sigmoid(objectness) of this lane's (cell, anchor)
    if (a < 3 && c < DC_SCREEN && cell0 + c < ncell) {
        // round 4: from the head convs' dense objectness array when they wrote one (one coalesced 16 bytes per cell: the same
        // float32 logits), else from the cell itself
        if (p.obj) sig = sigmoid_f(p.obj[(cell0 + c) * 4 + a]);
        else {
            const DecodeCell cl = decode_locate(p, cell0 + c);
            sig = sigmoid_f(cl.src[a * nf + 4]);
        }
        pass = sig > p.score_thr;
    }
    // Round 4: the box of every (cell, anchor) that passed -- the condition under which decode_one_cell used to compute it on 3 of its
    // 64 lanes, once per flagged cell -- is computed HERE, one per lane, for all of the wave's cells at once (custom_layers.py:251-256;
    // the four box logits come from the cell: its lines are read by the full-cell load anyway)
    if (pass) {
        const DecodeCell cl = decode_locate(p, cell0 + c);
        const int s = cl.s, g = p.g[s];
        const float* tp = cl.src + a * nf;
        const float t0 = tp[0], t1 = tp[1], t2 = tp[2], t3 = tp[3];
        const int row = (int)fastdiv((uint32_t)cl.rem, p.div_g[s]), col = cl.rem - row * g;
        const float bx = ((sigmoid_f(t0) * p.xyscale[s]) - p.xyoff[s] + (float)col) * (float)p.stride[s];
        const float by = ((sigmoid_f(t1) * p.xyscale[s]) - p.xyoff[s] + (float)row) * (float)p.stride[s];
        const float bw = expf(t2) * p.anchors[(s * 3 + a) * 2 + 0];
        const float bh = expf(t3) * p.anchors[(s * 3 + a) * 2 + 1];
        float4 o;
        o.x = (bx - bw / 2.0f) / p.img_size;
        o.y = (by - bh / 2.0f) / p.img_size;
        o.z = (bx + bw / 2.0f) / p.img_size;
        o.w = (by + bh / 2.0f) / p.img_size;
        *(float4*)(p.dboxes + ((int64_t)cl.n * p.nbox + p.box_off[s] + cl.rem * 3 + a) * 4) = o;
    }
    unsigned long long m = __ballot(pass);
    if (!m) return;
    auto next_cell = [&](DecodeCell& cl, float (&v)[4], float (&so)[3]) {        // pops the lowest flagged cell and issues its loads
        const int cc = (__ffsll((long long)m) - 1) >> 2;
        m &= ~(0xFull << (4 * cc));
        cl = decode_locate(p, cell0 + cc);
#pragma unroll
        for (int k = 0; k < 3; ++k) so[k] = __shfl(sig, cc * 4 + k);
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = (lane + 64 * k < nval) ? cl.src[lane + 64 * k] : 0.f;
    };
    __shared__ unsigned long long stage[4][DC_STAGE];
    DecodeStage st{stage[wave], 0u, -1};
    DecodeCell cur, nxt;
    float vc[4], vn[4], sc[3], sn[3];
    next_cell(cur, vc, sc);
    while (true) {                                                // wave-uniform: the next flagged cell's loads fly under this one
        const bool more = m != 0;
        if (more) next_cell(nxt, vn, sn);
        decode_one_cell(p, cur, vc, sc, lane, st);
        if (!more) break;
        cur = nxt;
#pragma unroll
        for (int k = 0; k < 4; ++k) vc[k] = vn[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) sc[k] = sn[k];
    }
    decode_flush(p, st, lane);
}

// ------------------------------------------------------------------------------------------- NMS
constexpr int NMS_THREADS = 1024;
constexpr int SORT_CAP = 4096;
constexpr int PAR_MAX_C = 1024;      // class count up to which the round-parallel greedy pass has its per-class LDS words
constexpr uint32_t PAR_NONE = 0xffffffffu;

__device__ __forceinline__ float iou_tf(const float4 a, const float4 b) {
    const float ay0 = fminf(a.x, a.z), ax0 = fminf(a.y, a.w), ay1 = fmaxf(a.x, a.z), ax1 = fmaxf(a.y, a.w);
    const float by0 = fminf(b.x, b.z), bx0 = fminf(b.y, b.w), by1 = fmaxf(b.x, b.z), bx1 = fmaxf(b.y, b.w);
    const float area_a = (ay1 - ay0) * (ax1 - ax0);
    const float area_b = (by1 - by0) * (bx1 - bx0);
    if (area_a <= 0.f || area_b <= 0.f) return 0.f;
    const float iy0 = fmaxf(ay0, by0), ix0 = fmaxf(ax0, bx0), iy1 = fminf(ay1, by1), ix1 = fminf(ax1, bx1);
    const float inter = fmaxf(iy1 - iy0, 0.f) * fmaxf(ix1 - ix0, 0.f);
    return inter / (area_a + area_b - inter);
}

// Bitonic sort of skey[0 .. m2) (m2 a power of two >= 64, <= NE * NMS_THREADS), descending, with the keys in registers
// (element tid + NMS_THREADS * e): partners less than 64 apart are exchanged by wave shuffles and partners NMS_THREADS or
// more apart sit in the same thread, so only the stages with 64 <= j < NMS_THREADS go through LDS (`xbuf`, m2 keys) and a
// workgroup barrier -- 14 of the 78 stages at 4096 keys.  (Chunks of at most 1024 keys take the rank sort in nms_kernel.)
template <int NE>
__device__ __forceinline__ void nms_sort_regs(unsigned long long* skey, unsigned long long* xbuf, uint32_t m2, int tid) {
    unsigned long long v[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) v[e] = (uint32_t)(tid + NMS_THREADS * e) < m2 ? skey[tid + NMS_THREADS * e] : 0ull;
    for (uint32_t k = 2; k <= m2; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            if (NE > 1 && j >= (uint32_t)NMS_THREADS) {          // partner = another register of this thread
                const int de = (int)(j / NMS_THREADS);
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    const int pe = e ^ de;
                    if (pe > e && pe < NE) {
                        const uint32_t i = (uint32_t)(tid + NMS_THREADS * e);
                        const bool desc = (i & k) == 0;
                        const unsigned long long a = v[e], b = v[pe];
                        if (desc ? a < b : a > b) { v[e] = b; v[pe] = a; }
                    }
                }
            } else if (j >= 64) {                                // partner in another wave: through LDS
#pragma unroll
                for (int e = 0; e < NE; ++e)
                    if ((uint32_t)(tid + NMS_THREADS * e) < m2) xbuf[tid + NMS_THREADS * e] = v[e];
                __syncthreads();
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    const uint32_t i = (uint32_t)(tid + NMS_THREADS * e);
                    if (i < m2) {
                        const unsigned long long b = xbuf[i ^ j];
                        const bool want_max = ((i & j) == 0) == ((i & k) == 0);
                        v[e] = want_max ? (v[e] > b ? v[e] : b) : (v[e] < b ? v[e] : b);
                    }
                }
                __syncthreads();
            } else {                                             // partner in this wave: shuffle
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    if (NE > 1 && (uint32_t)(NMS_THREADS * e) >= m2) continue;      // (uniform) nothing lives in this register
                    const uint32_t i = (uint32_t)(tid + NMS_THREADS * e);
                    const unsigned long long b = __shfl_xor(v[e], (int)j);
                    const bool want_max = ((i & j) == 0) == ((i & k) == 0);
                    v[e] = want_max ? (v[e] > b ? v[e] : b) : (v[e] < b ? v[e] : b);
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < NE; ++e)
        if ((uint32_t)(tid + NMS_THREADS * e) < m2) skey[tid + NMS_THREADS * e] = v[e];
    __syncthreads();
}

__global__ __launch_bounds__(NMS_THREADS) void nms_kernel(const NmsK p) {
    extern __shared__ __attribute__((aligned(16))) char nsm[];
    unsigned long long* skey = (unsigned long long*)nsm;                  // SORT_CAP
    float4* sbox = (float4*)(nsm + SORT_CAP * 8);                          // SORT_CAP
    float4* kbox = (float4*)(nsm + SORT_CAP * 24);                         // max_total
    int* kcls = (int*)(kbox + p.max_total);                                // max_total
    float* kscore = (float*)(kcls + p.max_total);                          // max_total
    int* kidx = (int*)(kscore + p.max_total);                              // max_total
    uint32_t* hist = (uint32_t*)(kidx + p.max_total);                      // 256
    uint32_t* sh = hist + 256;                                             // scratch words
    // sh[0]=count in chunk, sh[1]=kept, sh[2]=remaining below cutoff, sh[3..4]=pivot lo/hi, sh[5]=need;
    // round-parallel pass: sh[8..9] = lowest undecided position (two alternating buffers), sh[10..11] = kept below it
    uint32_t* pfirst = sh + 16;                                            // [2][C] lowest undecided position of a class
    uint32_t* pcount = pfirst + 2 * (p.C <= PAR_MAX_C ? p.C : 0);          // [C] kept boxes of a class
    uint32_t* pflag = pcount + (p.C <= PAR_MAX_C ? p.C : 0);               // [NMS_THREADS] candidate was kept; then [16] per-wave kept counts

    const int n = blockIdx.x, tid = threadIdx.x;
    uint32_t cnt = p.counts[(size_t)n * COUNT_STRIDE];
    if (cnt > p.cap) {
        if (tid == 0) atomicOr(p.status, 1u);
        cnt = p.cap;
    }
    const unsigned long long* gk = p.keys + (int64_t)n * p.cap;
    const float* gb = p.dboxes + (int64_t)n * p.nbox * 4;
    if (tid == 0) sh[1] = 0;
    unsigned long long cutoff = ~0ull;           // exclusive upper bound of keys still to visit
    __syncthreads();
    // every thread has read the image's candidate count: leave the counter at zero for the next step's decode (which therefore needs
    // no memset launch in front of it; decode_launch clears only when a decode was NOT followed by its NMS)
    if (tid == 0) p.counts[(size_t)n * COUNT_STRIDE] = 0;

    while (true) {
        // ---- how many keys remain below the cutoff?
        if (tid == 0) sh[2] = 0;
        __syncthreads();
        {
            uint32_t local = 0;
            for (uint32_t i = tid; i < cnt; i += NMS_THREADS) local += gk[i] < cutoff ? 1u : 0u;
            for (int o = 32; o > 0; o >>= 1) local += __shfl_down(local, o);
            if ((tid & 63) == 0 && local) atomicAdd(&sh[2], local);
        }
        __syncthreads();
        const uint32_t remaining = sh[2];
        if (remaining == 0) break;
        // ---- pivot = chunk_cap-th largest key below the cutoff (0 when everything fits).  The FIRST chunk is at most NMS_THREADS
        // keys -- what the merge-rank sort and the round-parallel pass take, and nearly always all that max_total kept boxes
        // need; the later ones SORT_CAP
        const uint32_t chunk_cap = cutoff == ~0ull ? (uint32_t)NMS_THREADS : (uint32_t)SORT_CAP;
        unsigned long long pivot = 0;
        bool have_pivot = false;
        if (cutoff == ~0ull && remaining > chunk_cap) {
            // The first chunk need not hold EXACTLY chunk_cap keys: one histogram pass over 1024 score bins (score bits >> 14 from
            // 0.25 up: any monotonic map of the key would do) and a suffix sum say from which bin up at most chunk_cap keys lie;
            // that bin's lowest key is the pivot.  ~2 us where the exact radix select below takes eight passes, each with a
            // serial scan of its 256 bins.  (A top bin that alone overflows the chunk -- a thousand equal scores -- falls through
            // to the exact select.)
            constexpr uint32_t BASE = 0x3e800000u >> 14;
            uint32_t* const hb = (uint32_t*)sbox;                   // (the box slots are not in use yet)
            const int lane = tid & 63, wv = tid >> 6;
            hb[tid] = 0;
            if (tid == 0) sh[3] = 0;                                // 1 + the pivot bin; stays 0 when not even the top bin fits
            __syncthreads();
            for (uint32_t i = tid; i < cnt; i += NMS_THREADS) {
                const int x = (int)((uint32_t)(gk[i] >> 46)) - (int)BASE;
                atomicAdd(&hb[x < 0 ? 0 : (x > 1023 ? 1023 : x)], 1u);
            }
            __syncthreads();
            uint32_t suf = hb[tid];                                 // keys in bins tid .. 1023: the wave's suffix sum, then the later waves'
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t v = __shfl_down(suf, o);
                suf += lane + o < 64 ? v : 0u;
            }
            if (lane == 0) hb[1024 + wv] = suf;
            __syncthreads();
            for (int w = wv + 1; w < NMS_THREADS / 64; ++w) suf += hb[1024 + w];
            // the lowest bin whose suffix fits the chunk (bin 0's suffix is `remaining`: it never does)
            if (tid > 0 && suf <= chunk_cap && suf + hb[tid - 1] > chunk_cap) { sh[3] = (uint32_t)tid + 1u; sh[4] = suf; }
            __syncthreads();
            // (one 2^-9-wide score bin holding more than a chunk while the bins above it are nearly empty would give a first
            //  chunk of a few keys and leave the image to the slow later passes: a suffix under half a chunk falls through to the exact
            //  radix select below, which fills the chunk)
            const uint32_t pb = sh[3] != 0 && 2u * sh[4] >= chunk_cap ? sh[3] : 0u;
            if (pb != 0) {
                pivot = (unsigned long long)((pb - 1u + BASE) << 14) << 32;       // the bin's lowest key
                have_pivot = true;
            }
            __syncthreads();
        }
        if (remaining > chunk_cap && !have_pivot) {
            unsigned long long prefix = 0, pmask = 0;
            uint32_t need = chunk_cap;
            for (int shift = 56; shift >= 0; shift -= 8) {
                for (int i = tid; i < 256; i += NMS_THREADS) hist[i] = 0;
                __syncthreads();
                for (uint32_t i = tid; i < cnt; i += NMS_THREADS) {
                    const unsigned long long k = gk[i];
                    if (k < cutoff && (k & pmask) == prefix) atomicAdd(&hist[(k >> shift) & 255], 1u);
                }
                __syncthreads();
                if (tid == 0) {
                    uint32_t acc = 0;
                    int d = 255;
                    for (; d > 0; --d) {
                        if (acc + hist[d] >= need) break;
                        acc += hist[d];
                    }
                    sh[5] = need - acc;                  // still needed inside digit d
                    sh[3] = (uint32_t)d;
                }
                __syncthreads();
                need = sh[5];
                prefix |= (unsigned long long)sh[3] << shift;
                pmask |= 255ull << shift;
                __syncthreads();
            }
            pivot = prefix;                               // keys are unique: exactly chunk_cap keys in [pivot, cutoff)
        }
        // ---- gather the chunk, sort it descending
        if (tid == 0) sh[0] = 0;
        __syncthreads();
        for (uint32_t i0 = 0; i0 < cnt; i0 += NMS_THREADS) {      // one LDS atomic per wave and pass, not per key: a thousand
            const uint32_t i = i0 + tid;                           // same-address atomics serialise
            const unsigned long long k = i < cnt ? gk[i] : 0ull;
            const bool take = i < cnt && k < cutoff && k >= pivot;
            const unsigned long long mask = __ballot(take);
            if (mask) {                                            // wave-uniform
                uint32_t base = 0;
                if ((tid & 63) == 0) base = atomicAdd(&sh[0], (uint32_t)__popcll(mask));
                base = __shfl(base, 0);
                const uint32_t pos = base + (uint32_t)__popcll(mask & ((1ull << (tid & 63)) - 1ull));
                if (take && pos < chunk_cap) skey[pos] = k;
            }
        }
        __syncthreads();
        const uint32_t m = sh[0] < chunk_cap ? sh[0] : chunk_cap;
        uint32_t m2 = 64;
        while (m2 < m) m2 <<= 1;
        for (uint32_t i = m + tid; i < m2; i += NMS_THREADS) skey[i] = 0;
        __syncthreads();
        // sort the chunk, descending
        if (m2 <= (uint32_t)NMS_THREADS) {
            // Rank sort: a thread's key goes to position (number of larger keys); keys are unique.  No dependent chain, two
            // barriers.  In-kernel timestamps at ~1000 keys: 22 us, the same as the 55 dependent exchange stages of a bitonic
            // network in LDS or in registers with shuffles, and as broadcasting the keys with v_readlane instead of LDS reads
            // (16 waves on one CU: LDS cycles of the broadcast reads here, instruction issue there) -- kept because it is the
            // shortest; the whole kernel went 80 -> 76.5 us with it and the wave-aggregated gather above.
            // Round 4: the ranks come from a merge instead of all pairs -- every wave sorts its 64 keys in registers (21 shuffle
            // stages), the sorted runs go back to LDS, and a key's rank is its place in its own run plus, per other run, the number
            // of larger keys found by a 6-step binary search (the searches of the runs are independent chains): ~600 instructions
            // per thread instead of ~3 500.
            unsigned long long* const xbuf = (unsigned long long*)sbox;
            unsigned long long key = (uint32_t)tid < m2 ? skey[tid] : 0ull;
            const int lane = tid & 63, wv = tid >> 6;
            for (int k = 2; k <= 64; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    const unsigned long long o = __shfl_xor(key, j);
                    const bool want_max = ((lane & j) == 0) == ((lane & k) == 0);      // descending run
                    key = (want_max ? o > key : o < key) ? o : key;
                }
            __syncthreads();
            if ((uint32_t)tid < m2) skey[tid] = key;
            __syncthreads();
            uint32_t rank = (uint32_t)lane;
            const int nruns = (int)(m2 >> 6);
            for (int r = 0; r < nruns; ++r) {
                if (r == wv) continue;                               // (wave-uniform)
                const unsigned long long* run = skey + r * 64;
                uint32_t lo = 0;                                     // number of keys of the run known to be larger
#pragma unroll
                for (int st = 32; st > 0; st >>= 1) lo += run[lo + st - 1] > key ? (uint32_t)st : 0u;
                lo += run[lo] > key ? 1u : 0u;                       // (lo <= 63 here)
                rank += lo;
            }
            __syncthreads();
            if ((uint32_t)tid < m2 && key != 0ull) xbuf[rank] = key;  // (keys are unique and non-zero; the zero padding stays behind position m)
            __syncthreads();
            if ((uint32_t)tid < m) skey[tid] = xbuf[tid];
            __syncthreads();
        } else {
            nms_sort_regs<SORT_CAP / NMS_THREADS>(skey, (unsigned long long*)sbox, m2, tid);
        }
        // ---- gather the chunk's boxes into LDS
        for (uint32_t i = tid; i < m; i += NMS_THREADS) {
            const uint32_t id = ~(uint32_t)(skey[i] & 0xffffffffull);
            sbox[i] = *(const float4*)(gb + (int64_t)fastdiv(id, p.div_c) * 4);
        }
        __syncthreads();
        // ---- greedy pass (wave 0), 64 sorted candidates at a time, one per lane.  The sequential rule "keep a candidate
        // unless an EARLIER KEPT box of its class overlaps it" is evaluated in two steps that give exactly the sequential
        // result: (1) every lane tests its candidate against the boxes kept before this batch (uniform loop over the kept
        // list in LDS; the IoU only for the rare same-class pairs); (2) the batch's survivors are visited in order -- the
        // lowest surviving lane is final, it is kept, its class and box are broadcast, and later same-class lanes test
        // against it.  Per candidate that is a few instructions instead of the ~90 of one-candidate-per-iteration on a wave
        // that retires one instruction per ~10 cycles (in-kernel timestamps: 41 us of this kernel were this pass).
        // Round 4 -- the usual case (all of the image's candidates in one chunk of at most NMS_THREADS) runs ROUND-PARALLEL on every
        // wave instead: one candidate per thread; in a round the lowest undecided candidate of every class (an LDS atomicMin per
        // class) is final -- every earlier kept box of its class was such a head in an earlier round and it survived the test
        // against it then -- so it is kept (class cap permitting) and the class's other undecided candidates test against it.
        // The rounds stop once the candidates in front of the lowest undecided one hold max_total kept boxes; the kept boxes'
        // ranks in sorted order are their output slots.  Same decisions as the sequential rule (same tests on the same pairs),
        // in as many rounds as the busiest class among the leading candidates has kept boxes (bench input: ~7) of two
        // barriers each, where the wave-0 pass below visits candidates 64 at a time at ~10 us per batch.
        const bool par = p.C <= PAR_MAX_C && cutoff == ~0ull;      // (the first chunk: m <= NMS_THREADS; the kept lists it leaves are
                                                                   // what the wave-0 pass of a later chunk continues from)
        if (par) {
            const uint32_t t = (uint32_t)tid;
            const int lane = tid & 63;
            const bool cap_binds = p.max_per_class < p.max_total;
            bool und = t < m, keptf = false;
            const unsigned long long key = und ? skey[t] : 0ull;
            const float4 cb = und ? sbox[t] : make_float4(0.f, 0.f, 0.f, 0.f);
            const uint32_t id = ~(uint32_t)(key & 0xffffffffull);
            const uint32_t bi = fastdiv(id, p.div_c);
            const int cls = und ? (int)(id - bi * (uint32_t)p.C) : 0;
            for (int c = tid; c < p.C; c += NMS_THREADS) { pfirst[c] = PAR_NONE; pfirst[p.C + c] = PAR_NONE; pcount[c] = 0; }
            if (tid == 0) { sh[8] = PAR_NONE; sh[9] = PAR_NONE; sh[10] = 0; sh[11] = 0; }
            __syncthreads();
            for (int r = 0;; ++r) {
                const int b = r & 1;
                uint32_t* const first = pfirst + b * p.C;
                if (und) atomicMin(&first[cls], t);
                {
                    const unsigned long long mk = __ballot(und);
                    if (mk && lane == __ffsll((long long)mk) - 1) atomicMin(&sh[8 + b], t);
                }
                __syncthreads();
                const uint32_t U = sh[8 + b];                       // every candidate in front of U is decided
                {
                    const unsigned long long mk = __ballot(keptf && t < U);
                    if (mk && lane == 0) atomicAdd(&sh[10 + b], (uint32_t)__popcll(mk));
                }
                const uint32_t h = und ? first[cls] : PAR_NONE;
                const bool head = und && h == t;
                if (head) {
                    keptf = !cap_binds || (int)pcount[cls] < p.max_per_class;      // (the class's only head this round)
                    if (keptf) pcount[cls] = pcount[cls] + 1;
                    pflag[t] = keptf ? 1u : 0u;
                    und = false;
                }
                if (tid == 0) { sh[8 + (b ^ 1)] = PAR_NONE; sh[10 + (b ^ 1)] = 0; }     // (last read before this round's first barrier)
                __syncthreads();
                if (U == PAR_NONE || sh[10 + b] >= (uint32_t)p.max_total) break;   // uniform
                if (head) first[cls] = PAR_NONE;                    // (every thread read it before the barrier; next use: round r + 2)
                if (und) {
                    // a head that the class cap turned away suppresses nothing -- and everything later in its class is turned away too
                    if (pflag[h] == 0u) und = false;
                    else if (iou_tf(cb, sbox[h]) > p.iou_thr) und = false;
                }
            }
            // output slot = number of kept candidates in front (sorted order)
            const unsigned long long mk = __ballot(keptf);
            uint32_t* const wtot = pflag + NMS_THREADS;
            if (lane == 0) wtot[tid >> 6] = (uint32_t)__popcll(mk);
            __syncthreads();
            uint32_t slot = (uint32_t)__popcll(mk & ((1ull << lane) - 1ull)), total = 0;
            for (int w = 0; w < NMS_THREADS / 64; ++w) {
                const uint32_t c = wtot[w];
                slot += w < (tid >> 6) ? c : 0u;
                total += c;
            }
            if (keptf && slot < (uint32_t)p.max_total) {
                kbox[slot] = cb;
                kcls[slot] = cls;
                kscore[slot] = __uint_as_float((uint32_t)(key >> 32));
                kidx[slot] = (int)bi;
            }
            if (tid == 0) sh[1] = total < (uint32_t)p.max_total ? total : (uint32_t)p.max_total;
        } else if (tid < 64) {
            int kept = (int)sh[1];
            const bool cap_binds = p.max_per_class < p.max_total;
            for (uint32_t t0 = 0; t0 < m && kept < p.max_total; t0 += 64) {
                const uint32_t t = t0 + (uint32_t)tid;
                const bool valid = t < m;
                const unsigned long long key = valid ? skey[t] : 0ull;
                const float4 cb = valid ? sbox[t] : make_float4(0.f, 0.f, 0.f, 0.f);
                const uint32_t id = ~(uint32_t)(key & 0xffffffffull);
                const uint32_t bi = fastdiv(id, p.div_c);
                const int cls = valid ? (int)(id - bi * (uint32_t)p.C) : -1;
                bool sup = !valid;
                int same = 0;                                   // kept boxes of this lane's class so far
                // (1) against the boxes kept before this batch: their classes come into registers 64 at a time and are
                // broadcast with v_readlane (an LDS read per kept box would put its latency into every iteration)
                for (int j0 = 0; j0 < kept; j0 += 64) {
                    const int kc_reg = j0 + tid < kept ? kcls[j0 + tid] : -2;
                    const int jn = kept - j0 < 64 ? kept - j0 : 64;
                    for (int j = 0; j < jn; ++j) {
                        if (__builtin_amdgcn_readlane(kc_reg, j) == cls) {
                            ++same;
                            sup = sup || (iou_tf(cb, kbox[j0 + j]) > p.iou_thr);
                        }
                    }
                }
                unsigned long long alive = __ballot(!sup);      // (2) survivors in order
                while (alive && kept < p.max_total) {
                    const int c = __ffsll((long long)alive) - 1;           // wave-uniform
                    alive &= alive - 1;
                    const int ccls = __builtin_amdgcn_readlane(cls, c);
                    if (cap_binds && __builtin_amdgcn_readlane(same, c) >= p.max_per_class) continue;   // class is full
                    float4 kb;
                    kb.x = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(cb.x), c));
                    kb.y = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(cb.y), c));
                    kb.z = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(cb.z), c));
                    kb.w = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(cb.w), c));
                    if (tid == c) {
                        kbox[kept] = cb;
                        kcls[kept] = cls;
                        kscore[kept] = __uint_as_float((uint32_t)(key >> 32));
                        kidx[kept] = (int)bi;
                    }
                    ++kept;
                    bool hit = false;
                    if (tid > c && !sup && cls == ccls) {
                        ++same;
                        hit = iou_tf(cb, kb) > p.iou_thr;
                    }
                    sup = sup || hit;
                    alive &= ~__ballot(hit);
                }
            }
            if (tid == 0) sh[1] = (uint32_t)kept;
        }
        __syncthreads();
        if ((int)sh[1] >= p.max_total || pivot == 0) break;
        cutoff = pivot;
        __syncthreads();
    }
    __syncthreads();
    // ---- outputs: clipped boxes, zero padded (clip_boxes=True, pad to max_total)
    const int kept = (int)sh[1];
    for (int i = tid; i < p.max_total; i += NMS_THREADS) {
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        float s = 0.f, c = 0.f;
        int id = -1;
        if (i < kept) {
            b = kbox[i];
            b.x = fmaxf(fminf(b.x, 1.f), 0.f); b.y = fmaxf(fminf(b.y, 1.f), 0.f);
            b.z = fmaxf(fminf(b.z, 1.f), 0.f); b.w = fmaxf(fminf(b.w, 1.f), 0.f);
            s = kscore[i]; c = (float)kcls[i]; id = kidx[i];
        }
        *(float4*)(p.out_boxes + ((int64_t)n * p.max_total + i) * 4) = b;
        p.out_scores[(int64_t)n * p.max_total + i] = s;
        p.out_classes[(int64_t)n * p.max_total + i] = c;
        if (p.out_idx) p.out_idx[(int64_t)n * p.max_total + i] = id;
    }
    if (tid == 0) p.out_valid[n] = kept;
}

size_t nms_lds_bytes(int max_total, int classes) {
    return (size_t)SORT_CAP * 24 + (size_t)max_total * (16 + 12) + 256 * 4 + 64 + (classes <= PAR_MAX_C ? (size_t)classes * 12 : 0) +
           (size_t)(NMS_THREADS + 16) * 4;
}

int decode_launch(const DecodeK& k, hipStream_t stream, int clear_images) {
    // the per-image candidate counters are zero between steps (nms_kernel resets its image's); `clear_images` > 0: the caller
    // cannot vouch for that (first use, or a decode whose NMS never ran) -- clear that many counters first
    if (clear_images > 0) Y4_CHECK_HIP(hipMemsetAsync(k.counts, 0, sizeof(uint32_t) * (size_t)clear_images * COUNT_STRIDE, stream));
    if (3 * (5 + k.C) <= 256) {
        const int64_t cells = (int64_t)k.N * k.cells_per_img;
        Y4_REQUIRE(cells < (1ll << 31), Y4_EINVAL, "decode: %lld cells exceed the 2^31 the cell -> image mapping divides exactly", (long long)cells);
        // fewer than eight 16-cell waves per SIMD of the chip (up to 5 images at 608^2): 4-cell waves
        if (cells < (int64_t)DC_SCREEN * 8192)
            hipLaunchKernelGGL(decode_cell_kernel<DC_SCREEN_SMALL>, dim3((int)((cells + 4 * DC_SCREEN_SMALL - 1) / (4 * DC_SCREEN_SMALL))), dim3(256), 0, stream, k);
        else
            hipLaunchKernelGGL(decode_cell_kernel<DC_SCREEN>, dim3((int)((cells + 4 * DC_SCREEN - 1) / (4 * DC_SCREEN))), dim3(256), 0, stream, k);
    } else {
        const int64_t boxes = (int64_t)k.N * k.nbox;
        hipLaunchKernelGGL(decode_kernel, dim3((int)((boxes + 255) / 256)), dim3(256), 0, stream, k);
    }
    Y4_CHECK_HIP(hipGetLastError());
    return Y4_OK;
}

int nms_launch(const NmsK& k, hipStream_t stream) {
    const size_t lds = nms_lds_bytes(k.max_total, k.C);
    Y4_REQUIRE(lds <= 160 * 1024 && k.max_total <= 1024, Y4_EINVAL, "nms: max_total %d needs %zu bytes of LDS", k.max_total, lds);
    static PerDeviceOnce once;
    if (const uint64_t bit = once.due()) {
        Y4_CHECK_HIP(hipFuncSetAttribute((const void*)nms_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        once.mark(bit);
    }
    hipLaunchKernelGGL(nms_kernel, dim3(k.N), dim3(NMS_THREADS), lds, stream, k);
    Y4_CHECK_HIP(hipGetLastError());
    return Y4_OK;
}

}  // namespace y4
