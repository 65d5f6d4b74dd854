// rcnn_ops.hip -- Mask R-CNN RoI / selection kernels (App. A.4-A.8).
//
//   rpn_sigmoid        : fused RPN head output [N][HW][A*5] -> objectness probabilities [N][HW*A]
//   rpn_decode_nms     : per (image, level): decode(1,1,1,1) + clip + min-size + greedy NMS on the
//                        score-sorted top-k; wavefront-64 ballots resolve each 64-box chunk
//   gather_proposals   : per-image top post_nms over all levels -> proposals
//   roi_align          : LevelMapper + legacy RoIAlign (aligned=False, sampling 2), NHWC gather
//   softmax_rows       : class probabilities
//   box_cls_nms        : per (image, class): score filter (proposal order) -> bitonic sort ->
//                        decode(10,10,5,5) + clip -> greedy NMS
//   finalize_dets      : kth-value cut to detections_per_img, order preserved
//   mask_logits_select : 1x1 conv to the detection's own class + sigmoid
//   paste_masks        : Masker(threshold, padding=1) into the image plane
// All arithmetic follows oracle/ora_ops.c operation for operation (bit-exact contract).
// Reference anchor: COCODemo.run_on_opencv_image (README.md:331) reaches every one of these.
#include "../../include/isegmi.h"
#include "common.h"
#include "rpn_levels.h"
#include "detmath.h"
#include <string.h>

namespace isegmi {

__device__ __forceinline__ unsigned f2ord_(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f_(unsigned o) {
    const unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(u);
}

// ------------------------------------------------------------------ box coder
__device__ __forceinline__ float4 decode_box(const float4 a, const float4 d, float wx, float wy, float ww, float wh) {
    const float clipv = 4.135166556742356f;
    const float widths = a.z - a.x + 1.0f, heights = a.w - a.y + 1.0f;
    const float ctr_x = a.x + 0.5f * widths, ctr_y = a.y + 0.5f * heights;
    const float dx = dm_div(d.x, wx), dy = dm_div(d.y, wy);
    float dw = dm_div(d.z, ww), dh = dm_div(d.w, wh);
    dw = dw < clipv ? dw : clipv;
    dh = dh < clipv ? dh : clipv;
    const float pcx = dx * widths + ctr_x, pcy = dy * heights + ctr_y;
    const float pw = dm_exp(dw) * widths, ph = dm_exp(dh) * heights;
    float4 o;
    o.x = pcx - 0.5f * pw;
    o.y = pcy - 0.5f * ph;
    o.z = pcx + 0.5f * pw - 1.0f;
    o.w = pcy + 0.5f * ph - 1.0f;
    return o;
}
__device__ __forceinline__ float clampf(float v, float hi) { return v < 0.0f ? 0.0f : (v > hi ? hi : v); }
__device__ __forceinline__ float4 clip_box(float4 b, float im_w, float im_h) {
    const float mx = im_w - 1.0f, my = im_h - 1.0f;
    b.x = clampf(b.x, mx); b.y = clampf(b.y, my); b.z = clampf(b.z, mx); b.w = clampf(b.w, my);
    return b;
}
__device__ __forceinline__ float iou_one(const float4 a, const float4 b, float one) {
    const float aa = (a.z - a.x + one) * (a.w - a.y + one);
    const float ab = (b.z - b.x + one) * (b.w - b.y + one);
    const float xx1 = a.x > b.x ? a.x : b.x, yy1 = a.y > b.y ? a.y : b.y;
    const float xx2 = a.z < b.z ? a.z : b.z, yy2 = a.w < b.w ? a.w : b.w;
    float w = xx2 - xx1 + one, h = yy2 - yy1 + one;
    w = w > 0.0f ? w : 0.0f;
    h = h > 0.0f ? h : 0.0f;
    const float inter = w * h;
    return dm_div(inter, aa + ab - inter);
}

// Division-free, EXACT form of `RN(inter / uni) > thr` (ge: `>= thr`).  RN is monotone, so the fp32 quotient
// exceeds thr iff the real quotient lies beyond the midpoint between thr and its fp32 neighbour (ties go to the
// even mantissa).  inter, uni are 24-bit, the midpoint 25-bit: their product is exact in fp64.  ~6 instructions
// instead of an IEEE-correct fp32 division (~40) in the innermost NMS loop; bit-identical to the oracle's division.
struct IouThr {
    double m;       // midpoint
    bool tie_true;  // result when inter == m * uni exactly
};
__device__ __forceinline__ IouThr make_iou_thr(float thr, int ge) {
    IouThr t;
    const unsigned b = __float_as_uint(thr);  // thr > 0
    if (ge) {  // q >= thr  <=>  x >= mid(pred(thr), thr) (tie -> thr iff thr's mantissa is even)
        const float lo = __uint_as_float(b - 1u);
        t.m = 0.5 * ((double)lo + (double)thr);
        t.tie_true = (b & 1u) == 0u;
    } else {   // q > thr   <=>  x >= mid(thr, succ(thr)) (tie -> succ iff succ's mantissa is even)
        const float hi = __uint_as_float(b + 1u);
        t.m = 0.5 * ((double)thr + (double)hi);
        t.tie_true = ((b + 1u) & 1u) == 0u;
    }
    return t;
}
__device__ __forceinline__ bool iou_exceeds(const float4 a, const float4 b, float one, const IouThr t) {
    const float aa = (a.z - a.x + one) * (a.w - a.y + one);
    const float ab = (b.z - b.x + one) * (b.w - b.y + one);
    const float xx1 = a.x > b.x ? a.x : b.x, yy1 = a.y > b.y ? a.y : b.y;
    const float xx2 = a.z < b.z ? a.z : b.z, yy2 = a.w < b.w ? a.w : b.w;
    float w = xx2 - xx1 + one, h = yy2 - yy1 + one;
    w = w > 0.0f ? w : 0.0f;
    h = h > 0.0f ? h : 0.0f;
    const float inter = w * h;
    const float uni = aa + ab - inter;
    if (!(uni > 0.0f)) return false;  // 0/0 or negative union: NaN / non-positive quotient never exceeds thr > 0
    const double lhs = (double)inter, rhs = t.m * (double)uni;
    return lhs > rhs || (lhs == rhs && t.tie_true);
}

// The `ge` / `nms_flags` argument of the engine-level launches below is the OR of the App. A.6 forks (include/isegmi.h): ISEGMI_NMS_GE (1) suppress on
// iou >= thr instead of >; ISEGMI_NMS_NO_PLUS_ONE (2) plain areas instead of the legacy +1; ISEGMI_NMS_INDEX_ORDER (4, box post-processing only) a class's
// kept detections in ascending proposal index (the CPU NMS's nonzero order) instead of score order.
__device__ __forceinline__ float nms_one(int flags) { return (flags & ISEGMI_NMS_NO_PLUS_ONE) ? 0.0f : 1.0f; }

// ------------------------------------------------------------------ greedy NMS core (block = NT threads)
// sb[0..n) boxes in visiting order (score desc, index asc); pre_dead[i] != 0 marks boxes removed beforehand.
// Writes kept positions (indices into sb) to kept[] in visiting order; returns the count (<= max_keep).
// Chunk of 64: (1) 4 waves test the chunk against the kept list; (2) wave 0 resolves the chunk with one
// ballot per surviving box.
constexpr int NMS_CAP = 1024;      // per-level FPN problems, per-class box NMS
constexpr int NMS_CAP_BIG = 6144;  // single-map RPN (PRE_NMS_TOP_N_TEST 6000) and the n = 4819 unit case: 96 KB of boxes in LDS
template <int CAP>
struct NmsSharedT {
    float4 sb[CAP];
    unsigned short kept[CAP];
    unsigned long long supp[16];
    int kc;
};
typedef NmsSharedT<NMS_CAP> NmsShared;
template <int CAP>
__device__ int nms_block(NmsSharedT<CAP>& S, int n, float thr, float one, int ge, int max_keep, const unsigned char* pre_dead) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nw = blockDim.x >> 6;
    const IouThr T = make_iou_thr(thr, ge);
    if (tid == 0) S.kc = 0;
    __syncthreads();
    for (int c0 = 0; c0 < n; c0 += 64) {
        const int kc = S.kc;
        if (max_keep > 0 && kc >= max_keep) break;
        const int i = c0 + lane;
        const bool valid = i < n;
        const float4 mine = S.sb[valid ? i : 0];
        bool sup = false;
        for (int j = wave; j < kc; j += nw) sup = sup || iou_exceeds(S.sb[S.kept[j]], mine, one, T);
        const unsigned long long sm = __ballot(sup);
        if (lane == 0) S.supp[wave] = sm;
        __syncthreads();
        if (wave == 0) {
            unsigned long long alive = __ballot(valid && !(pre_dead && pre_dead[i]));
            unsigned long long sall = 0ull;
            for (int q = 0; q < nw; ++q) sall |= S.supp[q];
            alive &= ~sall;
            unsigned long long keepm = 0ull;
            int cnt = kc;
            for (int b = 0; b < 64; ++b) {
                if (!((alive >> b) & 1ull)) continue;     // wave-uniform
                keepm |= 1ull << b;
                ++cnt;
                if (max_keep > 0 && cnt >= max_keep) break;
                const unsigned long long m = __ballot(iou_exceeds(S.sb[c0 + b], mine, one, T));
                const unsigned long long later = b == 63 ? 0ull : (~0ull << (b + 1));
                alive &= ~(m & later);
            }
            if ((keepm >> lane) & 1ull) {
                const int pos = kc + __popcll(keepm & ((1ull << lane) - 1ull));
                S.kept[pos] = (unsigned short)i;
            }
            if (lane == 0) S.kc = kc + __popcll(keepm);
        }
        __syncthreads();
    }
    return S.kc;
}

// Generic op: one problem per block; boxes pre-sorted by the caller? No: sorts here (n <= 1024).
// keys: (score desc, idx asc) bitonic in LDS.
__device__ void sort_desc_1024(unsigned long long* keys, int npow2) {
    for (int size = 2; size <= npow2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = threadIdx.x; t < npow2 / 2; t += blockDim.x) {
                const int lo = ((t / stride) * stride * 2) + (t % stride), hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const unsigned long long x = keys[lo], y = keys[hi];
                if (desc ? (x < y) : (x > y)) { keys[lo] = y; keys[hi] = x; }
            }
            __syncthreads();
        }
}
__device__ __forceinline__ int next_pow2(int n) { int p = 2; while (p < n) p <<= 1; return p; }

// boxes [P][n][4], scores [P][n] (any order); keep [P][n] original indices in score order; cnt [P]
__global__ __launch_bounds__(256) void nms_kernel(const float* __restrict__ boxes, const float* __restrict__ scores, int n,
                                                   float thr, int plus_one, int ge, int max_keep, int* __restrict__ keep,
                                                   int* __restrict__ cnt) {
    __shared__ NmsShared S;
    __shared__ unsigned long long keys[NMS_CAP];
    const int pb = blockIdx.x;
    const float* b = boxes + (int64_t)pb * n * 4;
    const float* s = scores + (int64_t)pb * n;
    const int np2 = next_pow2(n);
    for (int i = threadIdx.x; i < np2; i += 256)
        keys[i] = i < n ? (((unsigned long long)f2ord_(s[i]) << 32) | (unsigned long long)(0xffffffffu - (unsigned)i)) : 0ull;
    __syncthreads();
    sort_desc_1024(keys, np2);
    for (int i = threadIdx.x; i < n; i += 256) {
        const int src = (int)(0xffffffffu - (unsigned)(keys[i] & 0xffffffffull));
        S.sb[i] = *(const float4*)(b + (int64_t)src * 4);
    }
    __syncthreads();
    const int kc = nms_block(S, n, thr, plus_one ? 1.0f : 0.0f, ge, max_keep, nullptr);
    for (int i = threadIdx.x; i < kc; i += 256)
        keep[(int64_t)pb * n + i] = (int)(0xffffffffu - (unsigned)(keys[S.kept[i]] & 0xffffffffull));
    if (threadIdx.x == 0) cnt[pb] = kc;
}

// Same op for 1024 < n <= NMS_CAP_BIG: the sort keys and the sorted boxes share one LDS region (the original indices are
// parked in `order` in between), 1024 threads.
__global__ __launch_bounds__(1024) void nms_big_kernel(const float* __restrict__ boxes, const float* __restrict__ scores, int n, float thr,
                                                       int plus_one, int ge, int max_keep, int* __restrict__ keep, int* __restrict__ cnt) {
    __shared__ NmsSharedT<NMS_CAP_BIG> S;
    __shared__ unsigned short order[NMS_CAP_BIG];
    unsigned long long* keys = (unsigned long long*)S.sb;  // 8192 keys = 64 KB <= 96 KB of sb
    const int pb = blockIdx.x;
    const float* b = boxes + (int64_t)pb * n * 4;
    const float* sc = scores + (int64_t)pb * n;
    const int np2 = next_pow2(n);
    for (int i = threadIdx.x; i < np2; i += blockDim.x)
        keys[i] = i < n ? (((unsigned long long)f2ord_(sc[i]) << 32) | (unsigned long long)(0xffffffffu - (unsigned)i)) : 0ull;
    __syncthreads();
    sort_desc_1024(keys, np2);
    for (int i = threadIdx.x; i < n; i += blockDim.x) order[i] = (unsigned short)(0xffffffffu - (unsigned)(keys[i] & 0xffffffffull));
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) S.sb[i] = *(const float4*)(b + (int64_t)order[i] * 4);
    __syncthreads();
    const int kc = nms_block(S, n, thr, plus_one ? 1.0f : 0.0f, ge, max_keep, nullptr);
    for (int i = threadIdx.x; i < kc; i += blockDim.x) keep[(int64_t)pb * n + i] = (int)order[S.kept[i]];
    if (threadIdx.x == 0) cnt[pb] = kc;
}

// ------------------------------------------------------------------ RPN
// head [N][HW][CH] with CH = A*5: channel a = objectness logit of anchor a, channel A + a*4 + c = delta c.
__global__ void rpn_sigmoid_kernel(const float* __restrict__ head, int64_t total, int A, int CH, float* __restrict__ prob) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = i / A;
        const int a = (int)(i - pix * A);
        prob[i] = dm_sigmoid(head[pix * CH + a]);
    }
}

// RPN selection batched over (level, image) -- SURVEY 2.1 "level x image as ONE batch dimension" (round 5): the five levels' sigmoid,
// pre-NMS top-k, suppression matrix and greedy scan as FIVE launches instead of five per level.  Row (l, n) = level slot l of the batch, image n; per-level
// constants come from the table, per-row arrays (tk_*, nms workspace) are laid out [l][N][...].  Same kernels, same total order, same ballot ranking as the
// per-level path (which is this one with nl = 1): indices are bit-identical.
struct RpnLevels {
    int nl, N, A, CH, pre_nms, post_nms, post_cap, L;
    float thr, min_size;
    int ge;
    const float* head[RPN_MAX_LEVELS];      // [N][HW_l][CH]: A objectness logits, then A * 4 deltas per pixel
    const float* anchors[RPN_MAX_LEVELS];   // [HWA_l][4]
    int HWA[RPN_MAX_LEVELS];
    int level[RPN_MAX_LEVELS];              // slot of the level in the [N][L][post_cap] candidate lists
    int64_t prob_off[RPN_MAX_LEVELS + 1];   // floats: level l's probabilities [N][HWA_l] start at prob + prob_off[l]
};

__global__ void rpn_sigmoid_levels_kernel(const RpnLevels b, float* __restrict__ prob) {
    const int64_t total = b.prob_off[b.nl];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int l = 0;
        while (l + 1 < b.nl && i >= b.prob_off[l + 1]) ++l;
        const int64_t j = i - b.prob_off[l];
        const int64_t pix = j / b.A;
        const int a = (int)(j - pix * b.A);
        prob[i] = dm_sigmoid(b.head[l][pix * b.CH + a]);
    }
}

// grid (N); one (image, level) per block.  tk_vals/tk_idx [N][pre_nms] sorted; tk_cnt [N].
// out_boxes [N][L][post_cap][4], out_scores [N][L][post_cap] (-1 beyond count), out_cnt [N][L].
template <int CAP>
__global__ __launch_bounds__(1024) void rpn_decode_nms_kernel(const float* __restrict__ head, const float* __restrict__ anchors,
                                                              const float* __restrict__ tk_vals, const int* __restrict__ tk_idx,
                                                              const int* __restrict__ tk_cnt, const int* __restrict__ image_hw,
                                                              int HWA, int A, int CH, int pre_nms, int post_nms, float thr,
                                                              float min_size, int ge, int level, int L, int post_cap,
                                                              float* __restrict__ out_boxes, float* __restrict__ out_scores,
                                                              int* __restrict__ out_cnt) {
    __shared__ NmsSharedT<CAP> S;
    __shared__ unsigned char dead[CAP];
    const int n = blockIdx.x;
    const int cnt = tk_cnt[n];
    const float im_h = (float)image_hw[2 * n], im_w = (float)image_hw[2 * n + 1];
    for (int j = threadIdx.x; j < cnt; j += blockDim.x) {
        const int idx = tk_idx[(int64_t)n * pre_nms + j];
        const int pix = idx / A, a = idx - pix * A;
        const float* hp = head + ((int64_t)n * (HWA / A) + pix) * CH + A + a * 4;
        const float4 d = make_float4(hp[0], hp[1], hp[2], hp[3]);
        const float4 an = *(const float4*)(anchors + (int64_t)idx * 4);
        float4 b = clip_box(decode_box(an, d, 1.f, 1.f, 1.f, 1.f), im_w, im_h);
        S.sb[j] = b;
        const float ws = b.z - b.x + 1.0f, hs = b.w - b.y + 1.0f;
        dead[j] = (ws >= min_size && hs >= min_size) ? 0 : 1;
    }
    __syncthreads();
    const int kc = nms_block(S, cnt, thr, nms_one(ge), ge & ISEGMI_NMS_GE, post_nms, dead);
    const int64_t ob = ((int64_t)n * L + level) * post_cap;
    for (int i = threadIdx.x; i < post_cap; i += blockDim.x) {
        if (i < kc) {
            const int src = S.kept[i];
            *(float4*)(out_boxes + (ob + i) * 4) = S.sb[src];
            out_scores[ob + i] = tk_vals[(int64_t)n * pre_nms + src];
        } else {
            *(float4*)(out_boxes + (ob + i) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
            out_scores[ob + i] = -1.0f;
        }
    }
    if (threadIdx.x == 0) out_cnt[n * L + level] = kc;
}

// Greedy NMS scan over a finished suppression matrix (ONE wave; M in LDS, NMS_CAP/64 words per row: bit j of M[i][w] set iff
// box 64w+j comes after box i and IoU(i, j) exceeds the threshold).  A chunk of 64 is resolved from its diagonal words by a
// 64-step scalar chain (readlane / bitcmp / andn2: no IoU, no memory), then lane (w, q) ORs the kept rows' word w into the
// `removed` words with 16 unconditional LDS reads.  dead[i] != 0 removes box i beforehand; at most max_keep boxes are kept.
// Returns the kept count (uniform); kept[] holds the kept positions in visiting order.
__device__ __forceinline__ int nms_bit_scan(const unsigned long long* M, int cnt, int max_keep, const unsigned char* dead, unsigned short* kept) {
    constexpr int W = NMS_CAP / 64;
    const int lane = threadIdx.x & 63;
    const int nwords = (cnt + 63) >> 6;
    const int post_nms = max_keep;
    int kc_out;
    {
        unsigned long long rem = 0ull;  // lane w < nwords: removed / not-a-candidate bits of word w
        for (int w = 0; w < nwords; ++w) {
            const int i = (w << 6) + lane;
            const unsigned long long m = __ballot(i >= cnt || (dead != nullptr && dead[i < cnt ? i : 0]));
            if (lane == w) rem = m;
        }
        const unsigned long long lt_mask = (1ull << lane) - 1ull;
        const int pw = lane & 15, pq = lane >> 4;
        int kc = 0;
        for (int c = 0; c < nwords; ++c) {
            if (kc >= post_nms) break;
            const int i = (c << 6) + lane;
            const unsigned long long d = i < cnt ? M[i * W + c] : 0ull;
            const int dlo = (int)(unsigned)d, dhi = (int)(unsigned)(d >> 32);
            const unsigned long long rc = __shfl(rem, c, 64);
            unsigned long long alive = ~(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(rc >> 32)) << 32) |
                                         (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)rc));
#pragma unroll
            for (int b = 0; b < 64; ++b) {  // box b survives => it strikes its later chunk-mates; survivors are the kept ones
                const unsigned long long db = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(dhi, b) << 32) |
                                              (unsigned long long)(unsigned)__builtin_amdgcn_readlane(dlo, b);
                alive &= ((alive >> b) & 1ull) ? ~db : ~0ull;
            }
            unsigned long long keepm = alive;
            while (kc + __popcll(keepm) > post_nms) keepm &= ~(1ull << (63 - __builtin_clzll(keepm)));  // uniform; last chunk only
            if ((keepm >> lane) & 1ull) kept[kc + __popcll(keepm & lt_mask)] = (unsigned short)i;
            unsigned long long acc = 0ull;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int b = pq * 16 + u;
                const unsigned long long v = M[((c << 6) + b) * W + pw];
                acc |= ((keepm >> b) & 1ull) ? v : 0ull;
            }
            acc |= __shfl_xor(acc, 16, 64);
            acc |= __shfl_xor(acc, 32, 64);
            if (lane < 16 && lane > c) rem |= acc;
            kc += __popcll(keepm);
        }
        kc_out = kc;
    }
    return kc_out;
}

// The suppression matrix of n <= NMS_CAP boxes built by ONE block (all its waves): wave-task (r, w), r <= w, owns rows
// 64r..64r+63 (lane = row) x the 64 broadcast column boxes of word w.  For problems that have a block to themselves anyway
// (a crowded class in the per-class box NMS: 600 candidates took 250 us in nms_block's chunk-against-kept-list form).
__device__ void nms_matrix_block(const float4* sb, int n, float thr, float one, int ge, unsigned long long* M) {
    constexpr int W = NMS_CAP / 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const IouThr T = make_iou_thr(thr, ge);
    const int nwords = (n + 63) >> 6;
    const int npairs = nwords * (nwords + 1) / 2;
    for (int pair = wave; pair < npairs; pair += nw) {
        int w = 0;
        while ((w + 1) * (w + 2) / 2 <= pair) ++w;
        const int r = pair - w * (w + 1) / 2;
        const int i = (r << 6) + lane;
        const float4 mine = sb[i < n ? i : 0];
        unsigned long long m = 0ull;
#pragma unroll 8
        for (int b = 0; b < 64; ++b) {
            const int j = (w << 6) + b;
            const bool sup = j > i && j < n && iou_exceeds(mine, sb[j < n ? j : 0], one, T);
            m |= sup ? (1ull << b) : 0ull;
        }
        if (i < n) M[i * W + w] = m;
    }
}

// ---- the same per-level RPN NMS split over the chip (pre_nms <= NMS_CAP and a matrix workspace from the caller).
// The single-block kernel above is bound by ONE CU's VALU rate (~n^2/2 exact IoU tests, 256 us per level at n = 1000).  Here
//   rpn_nms_matrix_kernel  (136 waves per image): wave (r, w), r <= w, owns rows 64r..64r+63 x columns 64w..64w+63 of the
//       suppression matrix: bit j of M[i][w] set iff j > i and IoU(i, j) exceeds thr (lane = row, 64 broadcast column boxes);
//   rpn_nms_scan_kernel    (one block per image): copies the matrix into LDS and runs the greedy scan on bits alone -- a
//       chunk of 64 is resolved from its diagonal words by a 64-step scalar chain (readlane / bitcmp / andn2), then lane
//       (w, q) ORs the kept rows' word w into the `removed` words with 16 unconditional LDS reads.
// Same predicate, same visiting order => the kept list is identical to nms_block's.  ws: [N][NMS_CAP][NMS_CAP/64] u64.
__device__ __forceinline__ float4 rpn_candidate_box(const float* __restrict__ head, const float* __restrict__ anchors,
                                                    const int* __restrict__ tk_idx, int n, int j, int pre_nms, int HWA, int A, int CH,
                                                    float im_w, float im_h) {
    const int idx = tk_idx[(int64_t)n * pre_nms + j];
    const int pix = idx / A, a = idx - pix * A;
    const float* hp = head + ((int64_t)n * (HWA / A) + pix) * CH + A + a * 4;
    const float4 d = make_float4(hp[0], hp[1], hp[2], hp[3]);
    const float4 an = *(const float4*)(anchors + (int64_t)idx * 4);
    return clip_box(decode_box(an, d, 1.f, 1.f, 1.f, 1.f), im_w, im_h);
}

__global__ __launch_bounds__(256) void rpn_nms_matrix_kernel(const RpnLevels b, const int* __restrict__ tk_idx, const int* __restrict__ tk_cnt,
                                                             const int* __restrict__ image_hw, unsigned long long* __restrict__ ws) {
    constexpr int W = NMS_CAP / 64;
    __shared__ float4 cols[4][64];
    const int row = blockIdx.y;                      // (l, n)
    const int l = row / b.N, n = row - l * b.N;
    const float* __restrict__ head = b.head[l];
    const float* __restrict__ anchors = b.anchors[l];
    const int HWA = b.HWA[l], A = b.A, CH = b.CH, pre_nms = b.pre_nms;
    tk_idx += (int64_t)l * b.N * pre_nms;            // level l's [N][pre_nms]
    const int cnt = tk_cnt[row];
    const int nwords = (cnt + 63) >> 6;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = blockIdx.x * 4 + wave;  // wave-uniform
    int w = 0;
    while ((w + 1) * (w + 2) / 2 <= pair) ++w;
    const int r = pair - w * (w + 1) / 2;
    if (w >= nwords) return;  // whole wave; no block-level barrier below
    const float im_h = (float)image_hw[2 * n], im_w = (float)image_hw[2 * n + 1];
    const IouThr T = make_iou_thr(b.thr, b.ge & ISEGMI_NMS_GE);
    const float one = nms_one(b.ge);
    const int i = (r << 6) + lane, jc = (w << 6) + lane;
    const float4 mine = rpn_candidate_box(head, anchors, tk_idx, n, i < cnt ? i : 0, pre_nms, HWA, A, CH, im_w, im_h);
    cols[wave][lane] = rpn_candidate_box(head, anchors, tk_idx, n, jc < cnt ? jc : 0, pre_nms, HWA, A, CH, im_w, im_h);
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's own LDS writes have landed
    unsigned long long m = 0ull;
#pragma unroll 8
    for (int bb = 0; bb < 64; ++bb) {
        const int j = (w << 6) + bb;
        const bool sup = j > i && j < cnt && iou_exceeds(mine, cols[wave][bb], one, T);
        m |= sup ? (1ull << bb) : 0ull;
    }
    if (i < cnt) ws[((int64_t)row * NMS_CAP + i) * W + w] = m;
}

__global__ __launch_bounds__(1024) void rpn_nms_scan_kernel(const RpnLevels b, const float* __restrict__ tk_vals, const int* __restrict__ tk_idx,
                                                            const int* __restrict__ tk_cnt, const int* __restrict__ image_hw,
                                                            const unsigned long long* __restrict__ ws, float* __restrict__ out_boxes,
                                                            float* __restrict__ out_scores, int* __restrict__ out_cnt) {
    constexpr int W = NMS_CAP / 64;
    extern __shared__ unsigned long long M[];  // [NMS_CAP][W]
    __shared__ float4 sb[NMS_CAP];
    __shared__ unsigned short kept[NMS_CAP];
    __shared__ unsigned char dead[NMS_CAP];
    __shared__ int kc_sh;
    const int row = blockIdx.x;                      // (l, n)
    const int l = row / b.N, n = row - l * b.N;
    const float* __restrict__ head = b.head[l];
    const float* __restrict__ anchors = b.anchors[l];
    const int HWA = b.HWA[l], A = b.A, CH = b.CH, pre_nms = b.pre_nms, post_nms = b.post_nms, post_cap = b.post_cap, L = b.L, level = b.level[l];
    const float min_size = b.min_size;
    tk_idx += (int64_t)l * b.N * pre_nms;
    tk_vals += (int64_t)l * b.N * pre_nms;
    const int cnt = tk_cnt[row];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float im_h = (float)image_hw[2 * n], im_w = (float)image_hw[2 * n + 1];
    {   // matrix rows [0, cnt) -> LDS, 16 bytes per thread and step (rows past cnt and words left of the diagonal are stale
        // workspace; the scan never selects them)
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const u32x4* src = (const u32x4*)(ws + (int64_t)row * NMS_CAP * W);
        u32x4* dst = (u32x4*)M;
        const int n16 = cnt * W / 2;
        for (int q = tid; q < n16; q += 1024) dst[q] = src[q];
    }
    for (int j = tid; j < cnt; j += 1024) {
        const float4 bx = rpn_candidate_box(head, anchors, tk_idx, n, j, pre_nms, HWA, A, CH, im_w, im_h);
        sb[j] = bx;
        const float ws_ = bx.z - bx.x + 1.0f, hs = bx.w - bx.y + 1.0f;
        dead[j] = (ws_ >= min_size && hs >= min_size) ? 0 : 1;
    }
    __syncthreads();
    if (wave == 0) {
        const int kc = nms_bit_scan(M, cnt, post_nms, dead, kept);
        if (lane == 0) kc_sh = kc;
    }
    __syncthreads();
    const int kc = kc_sh;
    const int64_t ob = ((int64_t)n * L + level) * post_cap;
    for (int i = tid; i < post_cap; i += 1024) {
        if (i < kc) {
            const int src = kept[i];
            *(float4*)(out_boxes + (ob + i) * 4) = sb[src];
            out_scores[ob + i] = tk_vals[(int64_t)n * pre_nms + src];
        } else {
            *(float4*)(out_boxes + (ob + i) * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
            out_scores[ob + i] = -1.0f;
        }
    }
    if (tid == 0) out_cnt[n * L + level] = kc;
}

__global__ void sum_counts_kernel(const int* __restrict__ cnt, int L, int* __restrict__ total) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= (int)gridDim.x * (int)blockDim.x) return;
    int s = 0;
    for (int l = 0; l < L; ++l) s += cnt[n * L + l];
    total[n] = s;
}

// proposals [N][K][4] (+ batch index implied), from level-major candidate boxes via the final top-k indices
__global__ void gather_proposals_kernel(const float* __restrict__ cand_boxes, const float* __restrict__ fin_vals,
                                        const int* __restrict__ fin_idx, const int* __restrict__ fin_cnt, int cand_per_img, int K,
                                        float* __restrict__ props, float* __restrict__ prop_scores, int* __restrict__ prop_cnt) {
    const int n = blockIdx.x;
    const int cnt = fin_cnt[n];
    if (threadIdx.x == 0) prop_cnt[n] = cnt;
    for (int q = threadIdx.x; q < K; q += blockDim.x) {
        const int64_t o = (int64_t)n * K + q;
        if (q < cnt) {
            *(float4*)(props + o * 4) = *(const float4*)(cand_boxes + ((int64_t)n * cand_per_img + fin_idx[o]) * 4);
            prop_scores[o] = fin_vals[o];
        } else {
            *(float4*)(props + o * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
            prop_scores[o] = -1.0f;
        }
    }
}

// ------------------------------------------------------------------ RoIAlign
struct RoiLevels {
    const float* feat[4];
    int H[4], W[4];
    float scale[4];
};
__device__ __forceinline__ int level_of(const float4 b, int k_min, int k_max) {
    const float area = (b.z - b.x + 1.0f) * (b.w - b.y + 1.0f);
    const float s = dm_sqrt(area);
    const float t = floorf(4.0f + dm_log2(dm_div(s, 224.0f) + 1e-6f));
    int l = (int)t;
    return l < k_min ? k_min : (l > k_max ? k_max : l);
}
__device__ __forceinline__ float4 roi_bilinear4(const float* f, int H, int W, int C, float y, float x) {
    float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return z;
    if (y <= 0.0f) y = 0.0f;
    if (x <= 0.0f) x = 0.0f;
    int yl = (int)y, xl = (int)x, yh, xh;
    if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
    if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
    const float ly = y - (float)yl, lx = x - (float)xl, hy = 1.0f - ly, hx = 1.0f - lx;
    const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
    const float4 v1 = *(const float4*)(f + ((int64_t)yl * W + xl) * C), v2 = *(const float4*)(f + ((int64_t)yl * W + xh) * C);
    const float4 v3 = *(const float4*)(f + ((int64_t)yh * W + xl) * C), v4 = *(const float4*)(f + ((int64_t)yh * W + xh) * C);
    float4 v;
    v.x = w1 * v1.x; v.x = v.x + w2 * v2.x; v.x = v.x + w3 * v3.x; v.x = v.x + w4 * v4.x;
    v.y = w1 * v1.y; v.y = v.y + w2 * v2.y; v.y = v.y + w3 * v3.y; v.y = v.y + w4 * v4.y;
    v.z = w1 * v1.z; v.z = v.z + w2 * v2.z; v.z = v.z + w3 * v3.z; v.z = v.z + w4 * v4.z;
    v.w = w1 * v1.w; v.w = v.w + w2 * v2.w; v.w = v.w + w3 * v3.w; v.w = v.w + w4 * v4.w;
    return v;
}
// rois [N][K][4] image coords; counts [N]; out [N*K][PH][PW][C]; rows beyond count are zero-filled.
// fixed_level >= 0 forces that level index (unit tests of a single map); else LevelMapper(k_min..k_max).
// One block per (RoI, slice of its PH*PW*C/4 output vectors): everything that depends on the RoI alone -- level (sqrt, log2,
// division), scaled corners, bin sizes (two IEEE divisions), sampling grid -- is computed once per thread instead of once per
// output vector, and all index arithmetic is 32-bit.  The per-sample arithmetic is unchanged (same rounding sequence).
// GS = 2: the FPN models' fixed 2 x 2 sampling grid with the sample loops unrolled, so that a bin's 16 tap loads are in flight
// together (with run-time loop bounds every sample waited for its own four taps: four memory round trips per output vector).
template <int GS>
__global__ __launch_bounds__(256) void roi_align_kernel(const RoiLevels lv, const float* __restrict__ rois, const int* __restrict__ counts,
                                                         int N, int K, int C, int PH, int PW, int g, int k_min, int k_max,
                                                         int fixed_level, int aligned, float* __restrict__ out, int* __restrict__ out_level) {
    const int roi = blockIdx.x;
    const int n = roi / K, k = roi - n * K;
    const int c4n = C >> 2;
    const int per = PH * PW * c4n;
    const int chunk = (per + gridDim.y - 1) / gridDim.y;
    const int start = blockIdx.y * chunk, end = start + chunk < per ? start + chunk : per;
    float4* o4 = (float4*)out + (int64_t)roi * per;
    if (k >= counts[n]) {
        for (int j = start + threadIdx.x; j < end; j += 256) o4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const float4 b = *(const float4*)(rois + (int64_t)roi * 4);
    const int li = fixed_level >= 0 ? fixed_level : level_of(b, k_min, k_max) - k_min;
    if (out_level && threadIdx.x == 0 && blockIdx.y == 0) out_level[roi] = li + k_min;
    const int H = lv.H[li], W = lv.W[li];
    const float sc = lv.scale[li];
    const float* fbase = lv.feat[li] + (int64_t)n * H * W * C;
    // App. A.7 fork: aligned (ROIAlign aligned=True) = pixel-centre coordinates (scaled corner - 0.5) and no minimum RoI size of one pixel
    const float off = aligned ? 0.5f : 0.0f;
    const float sw = b.x * sc - off, sh = b.y * sc - off, ew = b.z * sc - off, eh = b.w * sc - off;
    float rw = ew - sw, rh = eh - sh;
    if (!aligned) {
        rw = rw > 1.0f ? rw : 1.0f;
        rh = rh > 1.0f ? rh : 1.0f;
    }
    const float bh = dm_div(rh, (float)PH), bw = dm_div(rw, (float)PW);
    // g > 0: fixed sampling grid; g <= 0: adaptive ceil(roi / pooled) (ROIAlign's sampling_ratio = 0, the C4 config)
    const int gh = GS > 0 ? GS : (g > 0 ? g : (int)ceilf(bh)), gw = GS > 0 ? GS : (g > 0 ? g : (int)ceilf(bw));
    const float cnt = (float)(gh * gw);
    for (int j = start + threadIdx.x; j < end; j += 256) {
        const int bin = j / c4n, c4 = j - bin * c4n;
        const int ph = bin / PW, pw = bin - ph * PW;
        const float* f = fbase + c4 * 4;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (GS > 0) {
#pragma unroll
            for (int iy = 0; iy < GS; ++iy) {
                const float y = sh + (float)ph * bh + dm_div(((float)iy + 0.5f) * bh, (float)GS);
#pragma unroll
                for (int ix = 0; ix < GS; ++ix) {
                    const float x = sw + (float)pw * bw + dm_div(((float)ix + 0.5f) * bw, (float)GS);
                    const float4 v = roi_bilinear4(f, H, W, C, y, x);
                    o.x = o.x + v.x; o.y = o.y + v.y; o.z = o.z + v.z; o.w = o.w + v.w;
                }
            }
        } else {
            for (int iy = 0; iy < gh; ++iy) {
                const float y = sh + (float)ph * bh + dm_div(((float)iy + 0.5f) * bh, (float)gh);
                for (int ix = 0; ix < gw; ++ix) {
                    const float x = sw + (float)pw * bw + dm_div(((float)ix + 0.5f) * bw, (float)gw);
                    const float4 v = roi_bilinear4(f, H, W, C, y, x);
                    o.x = o.x + v.x; o.y = o.y + v.y; o.z = o.z + v.z; o.w = o.w + v.w;
                }
            }
        }
        o.x = dm_div(o.x, cnt); o.y = dm_div(o.y, cnt); o.z = dm_div(o.z, cnt); o.w = dm_div(o.w, cnt);
        o4[j] = o;
    }
}

// ---- RoIAlign of the FPN heads (sampling 2, LevelMapper) in two launches: roi_prep_kernel, then roi_align_tab_kernel (fp32) or its fp16 twin in
// spatial_f16.hip.  Counters of the one-workgroup-per-RoI form (profiles/r05_hbm_stage_traffic_*.json, tools/roi_align_fetch.sh): the box head fetched
// 2.8x the P2-P5 maps -- the RoIs' footprints cover the maps 2.5x, every workgroup of the launch is resident at once and which of the eight L2s
// sees a RoI is its index mod 8, so nothing is re-used between RoIs -- and its VALU work (sample coordinates, clamps, bilinear weights, two IEEE
// divisions per coordinate, all repeated by every lane of a bin) was 2.5x the arithmetic on the taps.
//   roi_prep_kernel, one thread per RoI: (a) everything that depends on the RoI alone -- level, and for each of the 2*PH sample rows and 2*PW sample
//   columns {byte offset of the low / high tap row (column), weight of the low / high tap} with the validity test and the clamps of the scalar form
//   applied -- into a table of 2*(PH+PW)+1 16-byte entries per RoI; (b) the launch order: the image's RoIs ranked by (level, Morton code of the
//   centre on that level's map), by counting smaller keys in LDS.
//   roi_align_tab_kernel: workgroup L -- dispatched to XCD L % 8 -- takes one 128-byte channel slice of RoI order[L / 8 ...]: the whole chip works on
//   one window of spatially adjacent RoIs and every L2 holds its own slice of the window's pixels, so a pixel two RoIs share is fetched once (box
//   head, fp32 bs = 2: 874 -> 171 MB).  A lane reads its bin's four table entries from LDS, forms the 16 tap offsets with one add3 each and the
//   16 weights with one multiply each, has all 16 taps in flight, and runs the same multiply / add sequence per sample as roi_bilinear4.
// Per output element the arithmetic is roi_align_kernel<2>'s, operation for operation, with two exceptions that cannot change a bit, whatever the
// features hold (inf and NaN included): a sample outside the map is added as (+0) * (+0) instead of skipped -- its table entry carries weights +0 AND tap
// offsets past the end of the map, which the range-checked buffer load answers with 0, so no feature value ever meets the zero weight -- and the
// divisions by the powers of two 2 (half-bin offset) and 4 (sample count) are exact multiplications.  Results do not depend on `order` (one that is not
// a permutation leaves rows unwritten, entries outside [0, N*K) are skipped).  The last entry of a RoI's table carries {level, H, W, signature of
// (C * elem_bytes, PH, PW)}: a pooling launch given a table made for another layout writes NaN instead of plausible numbers.
__device__ __forceinline__ unsigned morton9(unsigned v) {   // 9 bits -> every second bit
    v &= 0x1ffu;
    v = (v | (v << 8)) & 0x00ff00ffu;
    v = (v | (v << 4)) & 0x0f0f0f0fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}
struct RoiPrepLevels {
    int H[4], W[4];
    float scale[4];
};
constexpr int ROI_ORDER_MAX = 2048;
// invalid:1 | level:2 | morton(cy, cx):18 | k:11
__device__ __forceinline__ unsigned roi_key(const float* __restrict__ rois, int64_t row0, int i, int K, int cnt, const RoiPrepLevels& lv, int k_min, int k_max) {
    if (i >= K) return 0xffffffffu;
    if (i >= cnt) return 0x80000000u | (unsigned)i;
    const float4 b = *(const float4*)(rois + (row0 + i) * 4);
    const int li = level_of(b, k_min, k_max) - k_min;
    const float s = lv.scale[li];
    const unsigned cx = (unsigned)fminf(fmaxf((b.x + b.z) * 0.5f * s, 0.0f), 511.0f);
    const unsigned cy = (unsigned)fminf(fmaxf((b.y + b.w) * 0.5f * s, 0.0f), 511.0f);
    return ((unsigned)li << 29) | (((morton9(cy) << 1) | morton9(cx)) << 11) | (unsigned)i;
}
// one table entry: a sample coordinate -> {low tap, high tap} x {byte offset, weight}; `stride` = bytes between consecutive taps along this axis
constexpr int ROI_TAB_OOR = 0x20000000;   // a tap offset no map reaches (roi_prep_launch: a level's map stays under 512 MiB per image); two of them + a slice offset still fit an int
__host__ __device__ __forceinline__ int roi_tab_sig(int C, int esize, int PH, int PW) { return (C * esize) | (PH << 16) | (PW << 24); }
__device__ __forceinline__ int4 roi_tab_entry(float v, int size, int stride) {
    if (v < -1.0f || v > (float)size) return make_int4(ROI_TAB_OOR, ROI_TAB_OOR, 0, 0);   // the sample contributes nothing: both weights +0, both taps read as 0
    if (v <= 0.0f) v = 0.0f;
    int lo = (int)v, hi;
    if (lo >= size - 1) { hi = lo = size - 1; v = (float)lo; } else hi = lo + 1;
    const float l = v - (float)lo, h = 1.0f - l;
    return make_int4(lo * stride, hi * stride, __float_as_int(h), __float_as_int(l));
}
// grid (N, ceil(K / 32)), 256 threads = 32 RoIs x 8 lanes; esize = bytes per feature element; tab [N*K][2*(PH+PW)+1] int4; order [N][K].
// Every workgroup builds the image's K keys in LDS; a RoI's rank = the number of smaller keys (keys are distinct: k is part of them), counted by
// its eight lanes over an eighth of the keys each; its table entries are spread over the same eight lanes (128-byte stores).
__global__ __launch_bounds__(256) void roi_prep_kernel(const float* __restrict__ rois, const int* __restrict__ counts, int K, RoiPrepLevels lv, int k_min,
                                                        int k_max, int C, int PH, int PW, int esize, int aligned, int* __restrict__ order, int4* __restrict__ tab) {
    __shared__ __attribute__((__aligned__(16))) unsigned key[ROI_ORDER_MAX];   // (read four at a time)
    const int n = blockIdx.x, cnt = counts[n];
    const int64_t row0 = (int64_t)n * K;
    const int K4 = (K + 3) & ~3;
    const int l = threadIdx.x & 7, i = blockIdx.y * 32 + (threadIdx.x >> 3);
    if (order) {
        for (int q = threadIdx.x; q < K4; q += 256) key[q] = roi_key(rois, row0, q, K, cnt, lv, k_min, k_max);
        __syncthreads();
        if (i < K) {
            const unsigned mine = key[i];
            int rank = 0;
            for (int j = l * 4; j < K4; j += 32) {
                const uint4 q = *(const uint4*)(key + j);
                rank += (q.x < mine) + (q.y < mine) + (q.z < mine) + (q.w < mine);
            }
            rank += __shfl_xor(rank, 1);
            rank += __shfl_xor(rank, 2);
            rank += __shfl_xor(rank, 4);
            if (l == 0) order[row0 + rank] = (int)row0 + i;
        }
    }
    if (i >= K || i >= cnt) return;
    const float4 b = *(const float4*)(rois + (row0 + i) * 4);
    const int li = level_of(b, k_min, k_max) - k_min;
    const int H = lv.H[li], W = lv.W[li];
    const float sc = lv.scale[li];
    const float off = aligned ? 0.5f : 0.0f;   // App. A.7 fork, as in roi_align_kernel
    const float sw = b.x * sc - off, sh = b.y * sc - off, ew = b.z * sc - off, eh = b.w * sc - off;
    float rw = ew - sw, rh = eh - sh;
    if (!aligned) {
        rw = rw > 1.0f ? rw : 1.0f;
        rh = rh > 1.0f ? rh : 1.0f;
    }
    const float bh = dm_div(rh, (float)PH), bw = dm_div(rw, (float)PW);
    const int TS = 2 * (PH + PW) + 1;
    int4* t = tab + (row0 + i) * TS;
    for (int s = l; s < TS; s += 8) {
        int4 e;
        if (s < 2 * PH) e = roi_tab_entry(sh + (float)(s >> 1) * bh + dm_div(((float)(s & 1) + 0.5f) * bh, 2.0f), H, W * C * esize);
        else if (s < TS - 1) {
            const int q = s - 2 * PH;
            e = roi_tab_entry(sw + (float)(q >> 1) * bw + dm_div(((float)(q & 1) + 0.5f) * bw, 2.0f), W, C * esize);
        } else e = make_int4(li, H, W, roi_tab_sig(C, esize, PH, PW));
        t[s] = e;
    }
}
// a tap = a 16-byte buffer load range-checked against the image's map: whatever the table holds, no load leaves the level's allocation (out of range reads 0)
typedef unsigned int roi_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 roi_tap4(const __amdgpu_buffer_rsrc_t rs, unsigned off) {
    return __builtin_bit_cast(float4, (roi_u32x4)__builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
}
// one sample of roi_bilinear4 from its table entries: v = w1*v1; v += w2*v2; v += w3*v3; v += w4*v4; o += v
#define ROI_SAMPLE(o, ye, xe, a, b, c, d)                                                                                   \
    {                                                                                                                       \
        const float hy = __int_as_float((ye).z), ly = __int_as_float((ye).w), hx = __int_as_float((xe).z), lx = __int_as_float((xe).w); \
        const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;                                              \
        float4 v;                                                                                                           \
        v.x = w1 * a.x; v.x = v.x + w2 * b.x; v.x = v.x + w3 * c.x; v.x = v.x + w4 * d.x;                                   \
        v.y = w1 * a.y; v.y = v.y + w2 * b.y; v.y = v.y + w3 * c.y; v.y = v.y + w4 * d.y;                                   \
        v.z = w1 * a.z; v.z = v.z + w2 * b.z; v.z = v.z + w3 * c.z; v.z = v.z + w4 * d.z;                                   \
        v.w = w1 * a.w; v.w = v.w + w2 * b.w; v.w = v.w + w3 * c.w; v.w = v.w + w4 * d.w;                                   \
        o.x = o.x + v.x; o.y = o.y + v.y; o.z = o.z + v.z; o.w = o.w + v.w;                                                 \
    }
// C / 4 = 8 * ns float4 columns, ns in {1, 2, 4, 8} slices of 128 B; 8 / ns RoIs per group of eight workgroups
template <int PH, int PW>
__global__ __launch_bounds__(256) void roi_align_tab_kernel(const RoiLevels lv, const int4* __restrict__ tab, const int* __restrict__ counts,
                                                             const int* __restrict__ order, int NK, int K, int C, int ns, float* __restrict__ out) {
    constexpr int TS = 2 * (PH + PW) + 1, NB = PH * PW;
    __shared__ int4 t[TS];
    const int x = blockIdx.x & 7, rpg = 8 / ns;
    const int seq = (blockIdx.x >> 3) * rpg + x / ns, slice = x % ns;
    if (seq >= NK) return;
    const int roi = order ? order[seq] : seq;
    if ((unsigned)roi >= (unsigned)NK) return;
    const int n = roi / K, k = roi - n * K;
    const int c4n = C >> 2;
    float4* o4 = (float4*)out + (int64_t)roi * (NB * c4n) + slice * 8 + (threadIdx.x & 7);
    if (k >= counts[n]) {
        for (int j = threadIdx.x; j < NB * 8; j += 256) o4[(j >> 3) * c4n] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const int4* tr = tab + (int64_t)roi * TS;
    if (threadIdx.x < TS) t[threadIdx.x] = tr[threadIdx.x];
    const int li = tr[TS - 1].x;   // (uniform address: a scalar load); the level's geometry comes from the launch arguments, not from the table
    if (tr[TS - 1].w != roi_tab_sig(C, 4, PH, PW)) {   // a table made for another element size / channel count / bin grid: fail loudly
        const float q = __int_as_float(0x7fc00000);
        for (int j = threadIdx.x; j < NB * 8; j += 256) o4[(j >> 3) * c4n] = make_float4(q, q, q, q);
        return;
    }
    const float* f0 = li == 0 ? lv.feat[0] : li == 1 ? lv.feat[1] : li == 2 ? lv.feat[2] : lv.feat[3];
    const int64_t img = (int64_t)(li == 0 ? lv.H[0] : li == 1 ? lv.H[1] : li == 2 ? lv.H[2] : lv.H[3]) * (li == 0 ? lv.W[0] : li == 1 ? lv.W[1] : li == 2 ? lv.W[2] : lv.W[3]) * C;
    const __amdgpu_buffer_rsrc_t fb = __builtin_amdgcn_make_buffer_rsrc((void*)(f0 + (int64_t)n * img), 0, (unsigned)(img * 4), 0x00020000);
    const unsigned lo = slice * 128 + (threadIdx.x & 7) * 16;
    __syncthreads();
    for (int j = threadIdx.x; j < NB * 8; j += 256) {
        const int bin = j >> 3;
        const int ph = bin / PW, pw = bin - ph * PW;
        const int4 y0 = t[2 * ph], y1 = t[2 * ph + 1], x0 = t[2 * PH + 2 * pw], x1 = t[2 * PH + 2 * pw + 1];
        const float4 a00 = roi_tap4(fb, y0.x + x0.x + lo), b00 = roi_tap4(fb, y0.x + x0.y + lo), c00 = roi_tap4(fb, y0.y + x0.x + lo), d00 = roi_tap4(fb, y0.y + x0.y + lo);
        const float4 a01 = roi_tap4(fb, y0.x + x1.x + lo), b01 = roi_tap4(fb, y0.x + x1.y + lo), c01 = roi_tap4(fb, y0.y + x1.x + lo), d01 = roi_tap4(fb, y0.y + x1.y + lo);
        const float4 a10 = roi_tap4(fb, y1.x + x0.x + lo), b10 = roi_tap4(fb, y1.x + x0.y + lo), c10 = roi_tap4(fb, y1.y + x0.x + lo), d10 = roi_tap4(fb, y1.y + x0.y + lo);
        const float4 a11 = roi_tap4(fb, y1.x + x1.x + lo), b11 = roi_tap4(fb, y1.x + x1.y + lo), c11 = roi_tap4(fb, y1.y + x1.x + lo), d11 = roi_tap4(fb, y1.y + x1.y + lo);
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        ROI_SAMPLE(o, y0, x0, a00, b00, c00, d00)
        ROI_SAMPLE(o, y0, x1, a01, b01, c01, d01)
        ROI_SAMPLE(o, y1, x0, a10, b10, c10, d10)
        ROI_SAMPLE(o, y1, x1, a11, b11, c11, d11)
        o.x = o.x * 0.25f; o.y = o.y * 0.25f; o.z = o.z * 0.25f; o.w = o.w * 0.25f;
        o4[bin * c4n] = o;
    }
}

// AvgPool2d over the whole HW window of every RoI (C4 FastRCNNPredictor): x [R][HW][C] -> out [R][C]; sequential fp32 sum, one division
__global__ void avgpool_full_kernel(const float* __restrict__ x, int64_t R, int HW, int C, float* __restrict__ out) {
    const int c4n = C >> 2;
    const int64_t total = R * c4n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / c4n;
        const int c4 = (int)(i - r * c4n);
        const float* p = x + (r * HW) * C + c4 * 4;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int q = 0; q < HW; ++q) {
            const float4 v = *(const float4*)(p + (int64_t)q * C);
            a.x = a.x + v.x; a.y = a.y + v.y; a.z = a.z + v.z; a.w = a.w + v.w;
        }
        const float n = (float)HW;
        *(float4*)(out + r * C + c4 * 4) = make_float4(dm_div(a.x, n), dm_div(a.y, n), dm_div(a.z, n), dm_div(a.w, n));
    }
}

// ------------------------------------------------------------------ box head post-processing
// One wave per row (C <= 320): the exponentials and the final divisions run across the lanes; the max is order-free; the SUM
// keeps the oracle's sequential order (s = ((e0 + e1) + e2) + ...), every lane adding the row's exponentials from LDS in
// class order.  (One thread per row left 2000-row problems on 32 waves: 60 us.)
constexpr int SOFTMAX_MAXC = 320;
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ x, int64_t rows, int C, int64_t in_stride,
                                                           float* __restrict__ y) {
    __shared__ float ex[4][SOFTMAX_MAXC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t r = (int64_t)blockIdx.x * 4 + wave;
    if (r >= rows) return;  // whole wave; no block barrier below
    const float* a = x + r * in_stride;
    float* o = y + r * C;
    float v[SOFTMAX_MAXC / 64];
    float m = -3.0e38f;
#pragma unroll
    for (int q = 0; q < SOFTMAX_MAXC / 64; ++q) {
        const int c = lane + 64 * q;
        v[q] = c < C ? a[c] : -3.0e38f;
        m = v[q] > m ? v[q] : m;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const float t = __shfl_xor(m, off, 64); m = t > m ? t : m; }
#pragma unroll
    for (int q = 0; q < SOFTMAX_MAXC / 64; ++q) {
        const int c = lane + 64 * q;
        if (c < C) { v[q] = dm_exp(v[q] - m); ex[wave][c] = v[q]; }
    }
    __builtin_amdgcn_wave_barrier();
    float sum = 0.0f;
    for (int c = 0; c < C; ++c) sum = sum + ex[wave][c];
#pragma unroll
    for (int q = 0; q < SOFTMAX_MAXC / 64; ++q) {
        const int c = lane + 64 * q;
        if (c < C) o[c] = dm_div(v[q], sum);
    }
}

// grid (ncls-1, N).  prob [N][R][ncls]; regr [N][R][regr_stride] (class j deltas at 4j..4j+3); props [N][R][4];
// prop_cnt [N].  cand_scores/cand_boxes [N][ncls-1][R] in NMS (score) order, -1 beyond the kept count.
// 16 waves: a crowded class's suppression matrix is ~60 dependent instructions per IoU test, and a wave alone on its SIMD
// issues one every ~6 cycles (11 us per 64x64 matrix block); four waves per SIMD fill the issue slots.
// CROWDED CLASSES ON THE WHOLE CHIP (round 6).  A class with more than 128 candidates needs its suppression matrix: m^2 / 2 exact IoU tests, ~60 dependent
// instructions each, and inside this kernel ONE block -- one CU -- builds it (m = 1000: 136 wave tasks of 64 x 64 tests over 16 waves, ~85 us; the
// bench's calibrated random weights put a few classes there, a crowded street scene would too).  With a `BoxCrowd` workspace the class is handed over
// instead: phase 0 (this kernel) stops after the sort + decode and parks the class's sorted keys and boxes; box_nms_matrix_kernel builds the matrices of
// all parked classes with one WAVE per 64 x 64 block over the whole chip (the RPN's rpn_nms_matrix_kernel, same predicate); phase 1 (this kernel again)
// reads a parked class back, runs the bit scan and emits.  Same predicate, same visiting order: the kept lists are those of the single-block path
// (tests/test_rcnn_ops_gpu.py::test_box_postprocess_matches_oracle runs both).  Uncrowded classes finish in phase 0 as before; for them the two extra
// launches are empty grids.  Mask R-CNN bs = 1: box_cls_nms 103 us -> phase 0 + matrix + phase 1 (profiles/r06_experiments.txt 3).
struct BoxCrowd {
    unsigned long long* matrix;   // [N * nc][NMS_CAP][NMS_CAP / 64]; nullptr: no hand-over (single-block path for every class)
    unsigned long long* keys;     // [N * nc][R] sorted (score desc, index asc) keys of a parked class
    float* boxes;                 // [N * nc][R][4] its decoded, clipped boxes in that order
    int* m;                       // [N * nc] candidates of a parked class, 0 = not parked
};
constexpr int BOX_NMS_THREADS = 1024;
__global__ __launch_bounds__(BOX_NMS_THREADS) void box_cls_nms_kernel(const float* __restrict__ prob, const float* __restrict__ regr,
                                                           int64_t regr_stride, const float* __restrict__ props,
                                                           const int* __restrict__ prop_cnt, const int* __restrict__ image_hw, int R,
                                                           int ncls, float score_thr, float nms_thr, int ge,
                                                           float* __restrict__ cand_scores, float* __restrict__ cand_boxes,
                                                           int* __restrict__ kept_total, const BoxCrowd crowd, const int phase) {
    __shared__ NmsShared S;
    __shared__ unsigned long long keys[NMS_CAP];
    constexpr int NT = BOX_NMS_THREADS, NW = NT / 64;
    __shared__ int wcnt[NW];
    const int j = blockIdx.x + 1, n = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int crow = n * (ncls - 1) + (j - 1);
    int m = 0, kc = 0;
    if (phase == 1) {   // a parked class comes back: keys + boxes into LDS, its finished matrix into the dynamic LDS, then the scan
        m = crowd.m[crow];
        if (m == 0) return;   // (uniform) finished in phase 0
        extern __shared__ unsigned long long nms_matrix[];
        for (int q = tid; q < m; q += NT) {
            keys[q] = crowd.keys[(int64_t)crow * R + q];
            S.sb[q] = *(const float4*)(crowd.boxes + ((int64_t)crow * R + q) * 4);
        }
        typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
        const u32x4* src = (const u32x4*)(crowd.matrix + (int64_t)crow * NMS_CAP * (NMS_CAP / 64));
        u32x4* dst = (u32x4*)nms_matrix;
        for (int q = tid; q < m * (NMS_CAP / 64) / 2; q += NT) dst[q] = src[q];   // rows [0, m): words left of the diagonal are stale workspace the scan never selects
        __syncthreads();
        if (wave == 0) {
            const int k = nms_bit_scan(nms_matrix, m, m, nullptr, S.kept);
            if (lane == 0) S.kc = k;
        }
        __syncthreads();
        kc = S.kc;
    } else {
    const int Rn = prop_cnt[n];
    // ordered compaction of candidates (proposal order): wave w owns a contiguous share
    const int seg = ((Rn + NW - 1) / NW + 63) & ~63;
    const int s0 = wave * seg, s1 = (s0 + seg) < Rn ? (s0 + seg) : Rn;
    int c = 0;
    for (int i = s0 + lane; (i - lane) < s1; i += 64) {
        const bool ok = i < s1 && prob[((int64_t)n * R + i) * ncls + j] > score_thr;
        c += __popcll(__ballot(ok));
    }
    if (lane == 0) wcnt[wave] = c;
    __syncthreads();
    int base = 0;
    for (int w = 0; w < wave; ++w) base += wcnt[w];
    for (int w = 0; w < NW; ++w) m += wcnt[w];
    int run = base;
    for (int i = s0 + lane; (i - lane) < s1; i += 64) {
        float p = 0.0f;
        const bool ok = i < s1 && (p = prob[((int64_t)n * R + i) * ncls + j]) > score_thr;
        const unsigned long long bm = __ballot(ok);
        if (ok) {
            const int pos = run + __popcll(bm & ((1ull << lane) - 1ull));
            keys[pos] = ((unsigned long long)f2ord_(p) << 32) | (unsigned long long)(0xffffffffu - (unsigned)i);
        }
        run += __popcll(bm);
    }
    const int np2 = next_pow2(m > 1 ? m : 2);
    for (int i = m + tid; i < np2; i += NT) keys[i] = 0ull;
    __syncthreads();
    sort_desc_1024(keys, np2);
    const float im_h = (float)image_hw[2 * n], im_w = (float)image_hw[2 * n + 1];
    for (int q = tid; q < m; q += NT) {
        const int i = (int)(0xffffffffu - (unsigned)(keys[q] & 0xffffffffull));
        const float4 pr = *(const float4*)(props + ((int64_t)n * R + i) * 4);
        const float* dp = regr + ((int64_t)n * R + i) * regr_stride + 4 * j;
        const float4 d = make_float4(dp[0], dp[1], dp[2], dp[3]);
        S.sb[q] = clip_box(decode_box(pr, d, 10.f, 10.f, 5.f, 5.f), im_w, im_h);
    }
    __syncthreads();
    if (crowd.matrix != nullptr) {   // (uniform)
        if (m > 128) {   // park the class for the chip-wide matrix
            for (int q = tid; q < m; q += NT) {
                crowd.keys[(int64_t)crow * R + q] = keys[q];
                *(float4*)(crowd.boxes + ((int64_t)crow * R + q) * 4) = S.sb[q];
            }
            if (tid == 0) crowd.m[crow] = m;
            return;
        }
        if (tid == 0) crowd.m[crow] = 0;
    }
    if (m > 128) {  // crowded class: bitmask NMS (identical kept list), the matrix in this block's dynamic LDS
        extern __shared__ unsigned long long nms_matrix[];
        nms_matrix_block(S.sb, m, nms_thr, nms_one(ge), ge & ISEGMI_NMS_GE, nms_matrix);
        __syncthreads();
        if (wave == 0) {
            const int k = nms_bit_scan(nms_matrix, m, m, nullptr, S.kept);
            if (lane == 0) S.kc = k;
        }
        __syncthreads();
        kc = S.kc;
    } else {
        kc = nms_block(S, m, nms_thr, nms_one(ge), ge & ISEGMI_NMS_GE, 0, nullptr);
    }
    }   // phase 0
    if (ge & ISEGMI_NMS_INDEX_ORDER) {
        // App. A.6 fork: the CPU NMS hands back the kept boxes in ascending ORIGINAL index (nonzero of the keep mask), so a class's detections come out in
        // proposal order.  slot[i] = sorted position of kept proposal i (0xffff: not kept), then an ordered compaction over the proposal indices
        // (ballot / popcount ranks: deterministic); the matrix region of the dynamic LDS is free here.
        extern __shared__ unsigned long long nms_matrix[];
        unsigned short* slot = (unsigned short*)nms_matrix;
        __syncthreads();
        for (int i = tid; i < R; i += NT) slot[i] = 0xffffu;
        __syncthreads();
        for (int q = tid; q < kc; q += NT) {
            const int src = S.kept[q];
            slot[(int)(0xffffffffu - (unsigned)(keys[src] & 0xffffffffull))] = (unsigned short)src;
        }
        __syncthreads();
        int run = 0;   // kept proposals before this pass of NT indices
        for (int i0 = 0; i0 < R; i0 += NT) {
            const int i = i0 + tid;
            const unsigned short sv = i < R ? slot[i] : (unsigned short)0xffffu;
            const unsigned long long bm = __ballot(sv != 0xffffu);
            if (lane == 0) wcnt[wave] = __popcll(bm);
            __syncthreads();
            int before = run, all = 0;
            for (int w = 0; w < NW; ++w) { before += w < wave ? wcnt[w] : 0; all += wcnt[w]; }
            if (sv != 0xffffu) S.kept[before + __popcll(bm & ((1ull << lane) - 1ull))] = sv;
            run += all;
            __syncthreads();
        }
    }
    const int64_t ob = ((int64_t)n * (ncls - 1) + (j - 1)) * R;
    for (int q = tid; q < R; q += NT) {
        if (q < kc) {
            const int src = S.kept[q];
            cand_scores[ob + q] = ord2f_((unsigned)(keys[src] >> 32));
            *(float4*)(cand_boxes + (ob + q) * 4) = S.sb[src];
        } else cand_scores[ob + q] = -1.0f;
    }
    if (tid == 0 && kc) atomicAdd(&kept_total[n], kc);
}

// grid (ceil(136 / 4), N * nc), 256 threads: wave (r, w), r <= w, of a parked class builds the 64 x 64 block of its suppression matrix -- rows 64r.., columns
// 64w.. -- as rpn_nms_matrix_kernel does for the RPN levels; classes that were not parked leave at once
__global__ __launch_bounds__(256) void box_nms_matrix_kernel(const BoxCrowd crowd, int R, float nms_thr, int flags) {
    constexpr int W = NMS_CAP / 64;
    __shared__ float4 cols[4][64];
    const int crow = blockIdx.y;
    const int m = crowd.m[crow];
    if (m == 0) return;
    const int nwords = (m + 63) >> 6;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair = blockIdx.x * 4 + wave;  // wave-uniform
    int w = 0;
    while ((w + 1) * (w + 2) / 2 <= pair) ++w;
    const int r = pair - w * (w + 1) / 2;
    if (w >= nwords) return;  // whole wave; no block-level barrier below
    const IouThr T = make_iou_thr(nms_thr, flags & ISEGMI_NMS_GE);
    const float one = nms_one(flags);
    const float4* sb = (const float4*)(crowd.boxes + (int64_t)crow * R * 4);
    const int i = (r << 6) + lane, jc = (w << 6) + lane;
    const float4 mine = sb[i < m ? i : 0];
    cols[wave][lane] = sb[jc < m ? jc : 0];
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's own LDS writes have landed
    unsigned long long bits = 0ull;
#pragma unroll 8
    for (int bb = 0; bb < 64; ++bb) {
        const int jj = (w << 6) + bb;
        const bool sup = jj > i && jj < m && iou_exceeds(mine, cols[wave][bb], one, T);
        bits |= sup ? (1ull << bb) : 0ull;
    }
    if (i < m) crowd.matrix[((int64_t)crow * NMS_CAP + i) * W + w] = bits;
}

// grid (N), block 1024 (16 waves).  Keeps score >= thr (thr = the det_per_img-th largest when more than det_per_img
// survive), class-major order preserved, at most cap rows.  Kept entries of a class are contiguous from rank 0
// (-1 beyond), so wave w walks classes w, w+16, ... 64 ranks at a time and stops at the first -1: work ~ sum of kept
// counts, not nc*R.  Two passes (count, then place) with ballot/popcount ranks keep the output order deterministic.
__global__ __launch_bounds__(1024) void finalize_dets_kernel(const float* __restrict__ cand_scores, const float* __restrict__ cand_boxes,
                                                              const int* __restrict__ kept_total, const float* __restrict__ top_vals,
                                                              int nc, int R, int det_per_img, int cap, int* __restrict__ out_cnt,
                                                              float* __restrict__ out_boxes, float* __restrict__ out_scores,
                                                              int* __restrict__ out_labels) {
    __shared__ int cls_pass[256];
    __shared__ int cls_base[256];
    __shared__ int total_s;
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int tot = kept_total[n];
    const bool cut = det_per_img > 0 && tot > det_per_img;
    const float thr = cut ? top_vals[(int64_t)n * det_per_img + det_per_img - 1] : -1e30f;
    const float* cs = cand_scores + (int64_t)n * nc * R;
    for (int c = wave; c < nc; c += nw) {
        int cnt = 0;
        for (int r0 = 0; r0 < R; r0 += 64) {
            const int r = r0 + lane;
            const float s = r < R ? cs[(int64_t)c * R + r] : -1.0f;
            cnt += __popcll(__ballot(s >= 0.0f && s >= thr));
            if (__ballot(s < 0.0f)) break;  // end of this class's kept list (wave-uniform)
        }
        if (lane == 0) cls_pass[c] = cnt;
    }
    __syncthreads();
    if (tid == 0) {
        int run = 0;
        for (int c = 0; c < nc; ++c) { cls_base[c] = run; run += cls_pass[c]; }
        total_s = run;
    }
    __syncthreads();
    for (int c = wave; c < nc; c += nw) {
        int run = cls_base[c];
        if (cls_pass[c] == 0) continue;
        for (int r0 = 0; r0 < R; r0 += 64) {
            const int r = r0 + lane;
            const float s = r < R ? cs[(int64_t)c * R + r] : -1.0f;
            const bool ok = s >= 0.0f && s >= thr;
            const unsigned long long bm = __ballot(ok);
            const int pos = run + __popcll(bm & ((1ull << lane) - 1ull));
            if (ok && pos < cap) {
                const int64_t o = (int64_t)n * cap + pos;
                out_scores[o] = s;
                out_labels[o] = c + 1;
                *(float4*)(out_boxes + o * 4) = *(const float4*)(cand_boxes + ((int64_t)n * nc * R + (int64_t)c * R + r) * 4);
            }
            run += __popcll(bm);
            if (__ballot(s < 0.0f)) break;
        }
    }
    const int cnt = total_s < cap ? total_s : cap;
    if (tid == 0) out_cnt[n] = cnt;
    for (int q = cnt + tid; q < cap; q += blockDim.x) {
        const int64_t o = (int64_t)n * cap + q;
        out_scores[o] = 0.0f; out_labels[o] = 0;
        *(float4*)(out_boxes + o * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// ------------------------------------------------------------------ mask head tail
// feat [R][HW][C]; w [ncls][C]; b [ncls]; labels [R] (0 = empty row -> zeros); out [R][HW].  grid (R).
__global__ __launch_bounds__(256) void mask_logits_select_kernel(const float* __restrict__ feat, int HW, int C, const float* __restrict__ w,
                                                                  const float* __restrict__ b, const int* __restrict__ labels,
                                                                  float* __restrict__ out) {
    extern __shared__ float wr[];
    const int r = blockIdx.x;
    const int lab = labels[r];
    if (lab <= 0) {
        for (int p = threadIdx.x; p < HW; p += 256) out[(int64_t)r * HW + p] = 0.0f;
        return;
    }
    for (int c = threadIdx.x; c < C; c += 256) wr[c] = w[(int64_t)lab * C + c];
    __syncthreads();
    const float bias = b[lab];
    for (int p = threadIdx.x; p < HW; p += 256) {
        const float4* x = (const float4*)(feat + ((int64_t)r * HW + p) * C);
        float acc = 0.0f;
        for (int c4 = 0; c4 < C / 4; ++c4) {
            const float4 v = x[c4];
            acc = fmaf(v.x, wr[4 * c4], acc); acc = fmaf(v.y, wr[4 * c4 + 1], acc);
            acc = fmaf(v.z, wr[4 * c4 + 2], acc); acc = fmaf(v.w, wr[4 * c4 + 3], acc);
        }
        out[(int64_t)r * HW + p] = dm_sigmoid(fmaf(acc, 1.0f, bias));
    }
}

// masks [N][K][M][M]; boxes [N][K][4] in output-image coordinates; counts [N]; out [N][K][im_h][im_w] u8.
__global__ __launch_bounds__(256) void paste_masks_kernel(const float* __restrict__ masks, const float* __restrict__ boxes,
                                                           const int* __restrict__ counts, int K, int M, int im_h, int im_w, float thr,
                                                           uint8_t* __restrict__ out, int* __restrict__ win) {
    const int n = blockIdx.z, d = blockIdx.y;
    if (d >= counts[n]) return;
    const float4 b = *(const float4*)(boxes + ((int64_t)n * K + d) * 4);
    const int P = M + 2;
    const float scale = dm_div((float)P, (float)M);
    float w_half = (b.z - b.x) * 0.5f, h_half = (b.w - b.y) * 0.5f;
    const float x_c = (b.z + b.x) * 0.5f, y_c = (b.w + b.y) * 0.5f;
    w_half = w_half * scale;
    h_half = h_half * scale;
    const int x1 = (int)(x_c - w_half), x2 = (int)(x_c + w_half);
    const int y1 = (int)(y_c - h_half), y2 = (int)(y_c + h_half);
    int w = x2 - x1 + 1, h = y2 - y1 + 1;
    w = w > 1 ? w : 1;
    h = h > 1 ? h : 1;
    const int x_0 = x1 > 0 ? x1 : 0, x_1 = (x2 + 1) < im_w ? (x2 + 1) : im_w;
    const int y_0 = y1 > 0 ? y1 : 0, y_1 = (y2 + 1) < im_h ? (y2 + 1) : im_h;
    const float* m = masks + ((int64_t)n * K + d) * M * M;
    uint8_t* o = out + ((int64_t)n * K + d) * im_h * im_w;
    // the window every set pixel of this plane lies in: [x_0, x_1) x [y_0, y_1) -- the run-length encoder reads nothing else of the plane
    if (win != nullptr && blockIdx.x == 0 && threadIdx.x == 0) { int* wq = win + ((int64_t)n * K + d) * 4; wq[0] = x_0; wq[1] = y_0; wq[2] = x_1; wq[3] = y_1; }
    // Only the box WINDOW is visited (round 3; round 2 walked the full width of every row the box touches: a box 15 % of the image wide paid for
    // 100 %): item = (window row, 4-pixel word of that row's window); a word is stored whole, its pixels outside the window are zero (they
    // are zero in the plane as well -- cleared beforehand, or never read when the planes are consumed through their windows).  Words that
    // straddle two rows may be written by both rows' items: identical bytes.
    const int64_t total = (int64_t)im_h * im_w;
    const int wpr = (x_1 - x_0 + 3) / 4 + 1;                 // words that can overlap one row's window
    const int items = (y_1 - y_0) * wpr;
    const float sy = dm_div((float)P, (float)h), sx = dm_div((float)P, (float)w);  // dm_bil_coef's scale, hoisted (same value)
    for (int i = blockIdx.x * 256 + threadIdx.x; i < items; i += gridDim.x * 256) {
        const int ry = i / wpr, j = i - ry * wpr;
        const int yrow = y_0 + ry;
        const int64_t rowbeg = (int64_t)yrow * im_w + x_0, rowend = (int64_t)yrow * im_w + x_1;
        const int64_t q = (rowbeg & ~3ll) + 4 * (int64_t)j;
        if (q >= rowend) continue;
        uint32_t word = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t f = q + e;
            if (f >= total) break;
            int y = yrow, x = (int)(f - (int64_t)yrow * im_w);   // the word may reach into the row above or below
            if (x < 0) { x += im_w; --y; } else if (x >= im_w) { x -= im_w; ++y; }
            if (y < y_0 || y >= y_1 || x < x_0 || x >= x_1) continue;
            int sy0, sy1, sx0, sx1; float ly0, ly1, lx0, lx1;
            dm_bil_coef_s(y - y1, P, sy, sy0, sy1, ly0, ly1);
            dm_bil_coef_s(x - x1, P, sx, sx0, sx1, lx0, lx1);
            auto padv = [&](int yy, int xx) -> float { return (yy >= 1 && yy <= M && xx >= 1 && xx <= M) ? m[(yy - 1) * M + (xx - 1)] : 0.0f; };
            float top = lx0 * padv(sy0, sx0); top = fmaf(lx1, padv(sy0, sx1), top);
            float bot = lx0 * padv(sy1, sx0); bot = fmaf(lx1, padv(sy1, sx1), bot);
            float v = ly0 * top; v = fmaf(ly1, bot, v);
            if (v > thr) word |= (1u << (8 * e));
        }
        if (q + 3 < total) *(uint32_t*)(o + q) = word;
        else for (int e = 0; e < 4 && q + e < total; ++e) o[q + e] = (uint8_t)((word >> (8 * e)) & 0xff);
    }
}

__global__ void scale_boxes_kernel(const float* __restrict__ boxes, const float* __restrict__ ratios, int K, float* __restrict__ out) {
    const int n = blockIdx.x;
    const float rw = ratios[2 * n], rh = ratios[2 * n + 1];
    for (int k = threadIdx.x; k < K; k += blockDim.x) {
        const float4 b = *(const float4*)(boxes + ((int64_t)n * K + k) * 4);
        *(float4*)(out + ((int64_t)n * K + k) * 4) = make_float4(b.x * rw, b.y * rh, b.z * rw, b.w * rh);
    }
}

// ------------------------------------------------------------------ launchers
static inline unsigned grid_for(int64_t total, int per = 256) {
    int64_t b = cdiv64(total, per);
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (unsigned)b;
}

int scale_boxes_launch(const float* boxes, const float* ratios, int N, int K, float* out, hipStream_t st) {
    hipLaunchKernelGGL(scale_boxes_kernel, dim3(N), dim3(128), 0, st, boxes, ratios, K, out);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}

int nms_launch(const float* boxes, const float* scores, int problems, int n, float thr, int plus_one, int ge, int max_keep, int* keep,
               int* cnt, hipStream_t st) {
    ARG_CHECK(n > 0 && n <= NMS_CAP_BIG, "nms n must be in 1..6144");
    ARG_CHECK(thr > 0.0f, "nms threshold must be > 0");
    if (problems == 0) return ISEGMI_OK;
    if (n <= NMS_CAP) hipLaunchKernelGGL(nms_kernel, dim3(problems), dim3(256), 0, st, boxes, scores, n, thr, plus_one, ge, max_keep, keep, cnt);
    else hipLaunchKernelGGL(nms_big_kernel, dim3(problems), dim3(1024), 0, st, boxes, scores, n, thr, plus_one, ge, max_keep, keep, cnt);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}

// AnchorGenerator.grid_anchors (A.3) on the device: out[(y * gw + x) * A + a] = base[a] + (x*stride, y*stride, x*stride, y*stride) -- one
// fp32 add per coordinate, exactly the host's numpy arithmetic (the shifts are small integers, exact in fp32).  Lets ONE engine serve any
// padded canvas: the anchors of a level are regenerated when the canvas changes instead of being a per-canvas constant tensor.
__global__ void grid_anchors_kernel(const float* __restrict__ base, int A, int stride, int gw, int64_t total, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t cell = i / A;
        const int a = (int)(i - cell * A);
        const int y = (int)(cell / gw), x = (int)(cell - (int64_t)y * gw);
        const float sx = (float)(x * stride), sy = (float)(y * stride);
        const float4 b = *(const float4*)(base + a * 4);
        *(float4*)(out + i * 4) = make_float4(sx + b.x, sy + b.y, sx + b.z, sy + b.w);
    }
}

int grid_anchors_launch(const float* base, int A, int stride, int gh, int gw, float* out, hipStream_t st) {
    const int64_t total = (int64_t)gh * gw * A;
    if (total <= 0) return ISEGMI_OK;
    hipLaunchKernelGGL(grid_anchors_kernel, dim3(grid_for(total)), dim3(256), 0, st, base, A, stride, gw, total, out);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}

int rpn_sigmoid_launch(const float* head, int64_t total, int A, int CH, float* prob, hipStream_t st) {
    hipLaunchKernelGGL(rpn_sigmoid_kernel, dim3(grid_for(total)), dim3(256), 0, st, head, total, A, CH, prob);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}

// suppression matrix + greedy scan of b.nl x b.N rows (tk_* laid out [l][N][pre_nms], tk_cnt [l][N], nms_ws b.nl * b.N * 131072 bytes): two launches
static int rpn_nms_levels_launch(const RpnLevels& b, const float* tk_vals, const int* tk_idx, const int* tk_cnt, const int* image_hw, void* nms_ws,
                                 float* out_boxes, float* out_scores, int* out_cnt, hipStream_t st) {
    constexpr int W = NMS_CAP / 64;
    constexpr int matrix_bytes = NMS_CAP * W * 8;  // 128 KB next to 19 KB of static LDS
    LDS_LIMIT_ONCE(matrix_bytes, rpn_nms_scan_kernel);
    const int pairs = W * (W + 1) / 2;
    hipLaunchKernelGGL(rpn_nms_matrix_kernel, dim3(cdiv(pairs, 4), b.nl * b.N), dim3(256), 0, st, b, tk_idx, tk_cnt, image_hw, (unsigned long long*)nms_ws);
    hipLaunchKernelGGL(rpn_nms_scan_kernel, dim3(b.nl * b.N), dim3(1024), matrix_bytes, st, b, tk_vals, tk_idx, tk_cnt, image_hw,
                       (const unsigned long long*)nms_ws, out_boxes, out_scores, out_cnt);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}

// Workspace of the batched selection (all in elements of the buffer's type): prob Sum N * HWA_l floats; cand_vals / cand_idx: the level-1 top-k candidates
// (rpn_topk_plan); tk_vals / tk_idx nl * N * pre_nms; tk_cnt nl * N; nms_ws nl * N * 131072 bytes.
int rpn_levels_workspace(int nl, int N, const int* HWA, int pre_nms, int64_t* prob_elems, int64_t* cand_elems) {
    ARG_CHECK(nl >= 1 && nl <= RPN_MAX_LEVELS && N >= 1 && prob_elems && cand_elems, "rpn levels");
    int slices[RPN_MAX_LEVELS];
    int64_t coff[RPN_MAX_LEVELS + 1], p = 0;
    for (int l = 0; l < nl; ++l) p += (int64_t)N * HWA[l];
    rpn_topk_plan(nl, N, HWA, pre_nms, slices, coff);
    *prob_elems = p;
    *cand_elems = coff[nl];
    return ISEGMI_OK;
}

// sigmoid -> top-k (two launches) -> decode + suppression matrix -> greedy scan for nl levels x N images: five launches on `st`.
// heads[l] [N][HW_l][A * 5] fp32, anchors[l] [HW_l * A][4]; level_slot[l] = the level's slot in the [N][L][post_cap] outputs.
int rpn_levels_select_launch(int nl, const float* const* heads, const float* const* anchors, const int* HWA, const int* level_slot, const int* image_hw, int N,
                             int A, int CH, int pre_nms, int post_nms, float thr, float min_size, int ge, int L, int post_cap, float* prob, float* cand_vals,
                             int* cand_idx, float* tk_vals, int* tk_idx, int* tk_cnt, void* nms_ws, float* out_boxes, float* out_scores, int* out_cnt,
                             hipStream_t st) {
    ARG_CHECK(nl >= 1 && nl <= RPN_MAX_LEVELS && N >= 1, "rpn levels");
    ARG_CHECK(pre_nms > 256 && pre_nms <= NMS_CAP && post_nms > 0 && post_nms <= post_cap && nms_ws, "batched RPN selection: 256 < pre_nms <= 1024, a matrix workspace");
    ARG_CHECK(thr > 0.0f, "nms threshold must be > 0");
    RpnLevels b;
    memset(&b, 0, sizeof(b));
    b.nl = nl; b.N = N; b.A = A; b.CH = CH; b.pre_nms = pre_nms; b.post_nms = post_nms; b.post_cap = post_cap; b.L = L;
    b.thr = thr; b.min_size = min_size; b.ge = ge;
    int slices[RPN_MAX_LEVELS];
    int64_t coff[RPN_MAX_LEVELS + 1];
    for (int l = 0; l < nl; ++l) {
        ARG_CHECK(heads[l] && anchors[l] && HWA[l] > 0 && HWA[l] % A == 0 && level_slot[l] >= 0 && level_slot[l] < L, "rpn level table");
        b.head[l] = heads[l]; b.anchors[l] = anchors[l]; b.HWA[l] = HWA[l]; b.level[l] = level_slot[l];
        b.prob_off[l + 1] = b.prob_off[l] + (int64_t)N * HWA[l];
    }
    rpn_topk_plan(nl, N, HWA, pre_nms, slices, coff);
    hipLaunchKernelGGL(rpn_sigmoid_levels_kernel, dim3(grid_for(b.prob_off[nl])), dim3(256), 0, st, b, prob);
    int rc = rpn_topk_levels_launch(nl, N, prob, b.prob_off, HWA, pre_nms, slices, coff, cand_vals, cand_idx, tk_vals, tk_idx, tk_cnt, st);
    if (rc) return rc;
    return rpn_nms_levels_launch(b, tk_vals, tk_idx, tk_cnt, image_hw, nms_ws, out_boxes, out_scores, out_cnt, st);
}

int rpn_decode_nms_launch(const float* head, const float* anchors, const float* tk_vals, const int* tk_idx, const int* tk_cnt,
                          const int* image_hw, int N, int HWA, int A, int CH, int pre_nms, int post_nms, float thr, float min_size,
                          int ge, int level, int L, int post_cap, float* out_boxes, float* out_scores, int* out_cnt, void* nms_ws,
                          hipStream_t st) {
    ARG_CHECK(pre_nms <= NMS_CAP_BIG && post_nms <= post_cap, "rpn sizes (pre_nms <= 6144)");
    ARG_CHECK(thr > 0.0f, "nms threshold must be > 0");
    if (pre_nms <= NMS_CAP && nms_ws != nullptr && post_nms > 0) {
        RpnLevels b;
        memset(&b, 0, sizeof(b));
        b.nl = 1; b.N = N; b.A = A; b.CH = CH; b.pre_nms = pre_nms; b.post_nms = post_nms; b.post_cap = post_cap; b.L = L;
        b.thr = thr; b.min_size = min_size; b.ge = ge;
        b.head[0] = head; b.anchors[0] = anchors; b.HWA[0] = HWA; b.level[0] = level;
        return rpn_nms_levels_launch(b, tk_vals, tk_idx, tk_cnt, image_hw, nms_ws, out_boxes, out_scores, out_cnt, st);
    } else if (pre_nms <= NMS_CAP)
        hipLaunchKernelGGL(rpn_decode_nms_kernel<NMS_CAP>, dim3(N), dim3(1024), 0, st, head, anchors, tk_vals, tk_idx, tk_cnt, image_hw, HWA, A,
                           CH, pre_nms, post_nms, thr, min_size, ge, level, L, post_cap, out_boxes, out_scores, out_cnt);
    else
        hipLaunchKernelGGL(rpn_decode_nms_kernel<NMS_CAP_BIG>, dim3(N), dim3(1024), 0, st, head, anchors, tk_vals, tk_idx, tk_cnt, image_hw, HWA,
                           A, CH, pre_nms, post_nms, thr, min_size, ge, level, L, post_cap, out_boxes, out_scores, out_cnt);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}

int sum_counts_launch(const int* cnt, int N, int L, int* total, hipStream_t st) {
    hipLaunchKernelGGL(sum_counts_kernel, dim3(1), dim3(N), 0, st, cnt, L, total);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}

int gather_proposals_launch(const float* cand_boxes, const float* fin_vals, const int* fin_idx, const int* fin_cnt, int N,
                            int cand_per_img, int K, float* props, float* prop_scores, int* prop_cnt, hipStream_t st) {
    hipLaunchKernelGGL(gather_proposals_kernel, dim3(N), dim3(256), 0, st, cand_boxes, fin_vals, fin_idx, fin_cnt, cand_per_img, K,
                       props, prop_scores, prop_cnt);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}

int roi_align_launch(const float* const* feats, const int* Hs, const int* Ws, const float* scales, int nlevels, const float* rois,
                     const int* counts, int N, int K, int C, int PH, int PW, int g, int k_min, int fixed_level, float* out,
                     int* out_level, hipStream_t st, const int* order = nullptr, const void* tab = nullptr, int aligned = 0) {
    ARG_CHECK(nlevels >= 1 && nlevels <= 4 && C % 4 == 0, "roi_align levels/C");
    RoiLevels lv;
    for (int i = 0; i < 4; ++i) {
        const int s = i < nlevels ? i : nlevels - 1;
        lv.feat[i] = feats[s]; lv.H[i] = Hs[s]; lv.W[i] = Ws[s]; lv.scale[i] = scales[s];
    }
    ARG_CHECK(N > 0 && K > 0 && (int64_t)N * K < (1ll << 31) && (int64_t)PH * PW * (C / 4) < (1ll << 30), "roi_align sizes");
    if (tab) {
        const int ns = C / 32;
        ARG_CHECK(g == 2 && fixed_level < 0 && !out_level && C % 32 == 0 && (ns == 1 || ns == 2 || ns == 4 || ns == 8) && (int64_t)N * K < (1ll << 27) &&
                      ((PH == 7 && PW == 7) || (PH == 14 && PW == 14)),
                  "roi_align from a table: sampling 2, LevelMapper, 7x7 or 14x14 bins, C in {32, 64, 128, 256}");
        for (int i = 0; i < nlevels; ++i) ARG_CHECK((int64_t)Hs[i] * Ws[i] * C * 4 < (int64_t)ROI_TAB_OOR, "roi_align from a table: a level's map must stay under 512 MiB per image");
        const int rpg = 8 / ns, NK = N * K;
        const dim3 grid((unsigned)((NK + rpg - 1) / rpg * 8));
        if (PH == 7)
            hipLaunchKernelGGL((roi_align_tab_kernel<7, 7>), grid, dim3(256), 0, st, lv, (const int4*)tab, counts, order, NK, K, C, ns, out);
        else
            hipLaunchKernelGGL((roi_align_tab_kernel<14, 14>), grid, dim3(256), 0, st, lv, (const int4*)tab, counts, order, NK, K, C, ns, out);
        HIP_TRY(hipGetLastError());
        return ISEGMI_OK;
    }
    // one block per (RoI, slice): enough slices that few-RoI launches (the mask head: N x 100 RoIs of 14 x 14 bins) still fill the chip
    const int per = PH * PW * (C / 4);
    int slices = (2048 + N * K - 1) / (N * K);
    const int max_slices = (per + 255) / 256;
    if (slices > max_slices) slices = max_slices;
    if (slices < 1) slices = 1;
    if (g == 2)
        hipLaunchKernelGGL(roi_align_kernel<2>, dim3((unsigned)(N * K), (unsigned)slices), dim3(256), 0, st, lv, rois, counts, N, K, C, PH, PW, g, k_min,
                           k_min + nlevels - 1, fixed_level, aligned, out, out_level);
    else
        hipLaunchKernelGGL(roi_align_kernel<0>, dim3((unsigned)(N * K), (unsigned)slices), dim3(256), 0, st, lv, rois, counts, N, K, C, PH, PW, g, k_min,
                           k_min + nlevels - 1, fixed_level, aligned, out, out_level);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}

// The FPN heads' RoIAlign, first launch: tab [N*K][2*(PH+PW)+1][4] int32 (sample-row / sample-column entries, then {level index, H, W, 0}) and, when
// `order` is given, order [N][K] = each image's RoI rows (n*K + k) sorted by (level, Morton code of the centre on that level's map), rows beyond
// counts[n] last.  esize = bytes per feature element (4 fp32, 2 fp16): the table holds byte offsets.
int roi_prep_launch(const float* rois, const int* counts, int N, int K, const int* Hs, const int* Ws, const float* scales, int nlevels, int k_min, int C,
                    int PH, int PW, int esize, int* order, void* tab, hipStream_t st, int aligned) {  // (default 0 in engine.h)
    ARG_CHECK(nlevels >= 1 && nlevels <= 4 && N > 0 && K > 0 && K <= ROI_ORDER_MAX && PH > 0 && PW > 0 && PH <= 64 && PW <= 64 && C > 0 &&
                  (esize == 2 || esize == 4) && tab,
              "roi_prep: 1..4 levels, K <= 2048");
    RoiPrepLevels lv;
    for (int i = 0; i < 4; ++i) {
        const int s = i < nlevels ? i : nlevels - 1;
        lv.H[i] = Hs[s]; lv.W[i] = Ws[s]; lv.scale[i] = scales[s];
        ARG_CHECK(Hs[s] > 0 && Ws[s] > 0 && (int64_t)Hs[s] * Ws[s] * C * esize < (int64_t)ROI_TAB_OOR, "roi_prep: a level's map must stay under 512 MiB per image");
        ARG_CHECK((int64_t)C * esize < 65536 && PH < 128 && PW < 128, "roi_prep: C * elem_bytes < 65536");
    }
    hipLaunchKernelGGL(roi_prep_kernel, dim3((unsigned)N, (unsigned)((K + 31) / 32)), dim3(256), 0, st, rois, counts, K, lv, k_min, k_min + nlevels - 1, C,
                       PH, PW, esize, aligned, order, (int4*)tab);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}

int avgpool_full_launch(const float* x, int64_t R, int HW, int C, float* out, hipStream_t st) {
    ARG_CHECK(C % 4 == 0 && HW > 0, "avgpool: C % 4, HW");
    if (R == 0) return ISEGMI_OK;
    hipLaunchKernelGGL(avgpool_full_kernel, dim3(grid_for(R * (C / 4))), dim3(256), 0, st, x, R, HW, C, out);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}

int softmax_rows_launch(const float* x, int64_t rows, int C, int64_t in_stride, float* y, hipStream_t st) {
    ARG_CHECK(C >= 1 && C <= SOFTMAX_MAXC, "softmax width (<= 320 classes)");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, x, rows, C, in_stride, y);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}

int topk_segmented_launch(const float* keys, int64_t row_stride, int rows, int nseg, int seg_len, int seg_take, int k, const int* limit,
                          int rows_per_limit, float* out_vals, int* out_idx, int* out_cnt, hipStream_t st);
int topk_launch(const float* keys, int64_t row_stride, int rows, int n, int k, const int* limit, int rows_per_limit,
                float* out_vals, int* out_idx, int* out_cnt, hipStream_t st);

// box post-processing for N images: logits -> detections.  Workspaces are caller-provided.
int box_postprocess_launch(const isegmi_box_post_args* a, hipStream_t st) {
    ARG_CHECK(a->N > 0 && a->R > 0 && a->R <= NMS_CAP && a->ncls >= 2 && a->ncls <= 257, "box post sizes (R <= 1024, ncls <= 257)");
    ARG_CHECK(a->det_per_img > 0 && a->det_per_img <= 1024 && a->cap >= a->det_per_img, "det_per_img / cap");
    ARG_CHECK(a->nms_thresh > 0.0f, "nms threshold must be > 0");
    const int nc = a->ncls - 1;
    int rc = softmax_rows_launch(a->d_logits, (int64_t)a->N * a->R, a->ncls, a->logits_stride, a->d_ws_prob, st);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(a->d_ws_kept_total, 0, sizeof(int) * (size_t)a->N, st));
    constexpr int matrix_bytes = NMS_CAP * (NMS_CAP / 64) * 8;  // 128 KB next to 26 KB of static LDS: one block per CU
    LDS_LIMIT_ONCE(matrix_bytes, box_cls_nms_kernel);
    BoxCrowd crowd;
    crowd.matrix = (unsigned long long*)a->d_ws_crowd_matrix; crowd.keys = (unsigned long long*)a->d_ws_crowd_keys;
    crowd.boxes = a->d_ws_crowd_boxes; crowd.m = a->d_ws_crowd_m;
    ARG_CHECK((crowd.matrix && crowd.keys && crowd.boxes && crowd.m) || (!crowd.matrix && !crowd.keys && !crowd.boxes && !crowd.m),
              "box post-processing: the four crowd workspaces come together or not at all");
    // phase 0 with a hand-over needs the dynamic LDS only for the index-order fork's slot table (2 KB): two blocks share a CU
    hipLaunchKernelGGL(box_cls_nms_kernel, dim3(nc, a->N), dim3(BOX_NMS_THREADS), crowd.matrix ? 4096 : matrix_bytes, st, a->d_ws_prob, a->d_regr,
                       a->regr_stride, a->d_props, a->d_prop_cnt, a->d_image_hw, a->R, a->ncls, a->score_thresh, a->nms_thresh, a->nms_flags,
                       a->d_ws_cand_scores, a->d_ws_cand_boxes, a->d_ws_kept_total, crowd, 0);
    if (crowd.matrix) {
        constexpr int W = NMS_CAP / 64, pairs = W * (W + 1) / 2;
        hipLaunchKernelGGL(box_nms_matrix_kernel, dim3(cdiv(pairs, 4), nc * a->N), dim3(256), 0, st, crowd, a->R, a->nms_thresh, a->nms_flags);
        hipLaunchKernelGGL(box_cls_nms_kernel, dim3(nc, a->N), dim3(BOX_NMS_THREADS), matrix_bytes, st, a->d_ws_prob, a->d_regr, a->regr_stride,
                           a->d_props, a->d_prop_cnt, a->d_image_hw, a->R, a->ncls, a->score_thresh, a->nms_thresh, a->nms_flags,
                           a->d_ws_cand_scores, a->d_ws_cand_boxes, a->d_ws_kept_total, crowd, 1);
    }
    HIP_TRY(hipGetLastError());
    // the per-class lists are sorted (score order): nothing past a class's first det_per_img entries can make the image's top det_per_img; in index
    // order the lists are not sorted and the general top-k over all nc * R slots finds the kth value
    if (a->det_per_img <= 128 && a->det_per_img <= a->R && !(a->nms_flags & ISEGMI_NMS_INDEX_ORDER))
        rc = topk_segmented_launch(a->d_ws_cand_scores, (int64_t)nc * a->R, a->N, nc, a->R, a->det_per_img, a->det_per_img, a->d_ws_kept_total, 1,
                                   a->d_ws_top_vals, a->d_ws_top_idx, nullptr, st);
    else
        rc = topk_launch(a->d_ws_cand_scores, (int64_t)nc * a->R, a->N, nc * a->R, a->det_per_img, a->d_ws_kept_total, 1, a->d_ws_top_vals,
                         a->d_ws_top_idx, nullptr, st);
    if (rc) return rc;
    hipLaunchKernelGGL(finalize_dets_kernel, dim3(a->N), dim3(1024), 0, st, a->d_ws_cand_scores, a->d_ws_cand_boxes, a->d_ws_kept_total,
                       a->d_ws_top_vals, nc, a->R, a->det_per_img, a->cap, a->d_out_count, a->d_out_boxes, a->d_out_scores,
                       a->d_out_labels);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}

int mask_logits_select_launch(const float* feat, int R, int HW, int C, const float* w, const float* b, const int* labels, float* out,
                              hipStream_t st) {
    ARG_CHECK(C % 4 == 0, "C % 4");
    if (R == 0) return ISEGMI_OK;
    hipLaunchKernelGGL(mask_logits_select_kernel, dim3(R), dim3(256), (size_t)C * sizeof(float), st, feat, HW, C, w, b, labels, out);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}

// clear: zero the planes first (the kernel writes only the rows of each box, and inside them only the words that touch the box); a caller that
// reads the planes through the windows alone (`win` [N][K][4], the run-length encoder) may skip it.
int paste_masks_launch(const float* masks, const float* boxes, const int* counts, int N, int K, int M, int im_h, int im_w, float thr,
                       uint8_t* out, hipStream_t st, int* win, bool clear) {
    if (clear) HIP_TRY(hipMemsetAsync(out, 0, (size_t)N * K * im_h * im_w, st));
    const unsigned bx = grid_for((int64_t)im_h * im_w, 1024) > 64 ? 64 : grid_for((int64_t)im_h * im_w, 1024);
    hipLaunchKernelGGL(paste_masks_kernel, dim3(bx, K, N), dim3(256), 0, st, masks, boxes, counts, K, M, im_h, im_w, thr, out, win);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}

}  // namespace isegmi

using namespace isegmi;

extern "C" int isegmi_op_nms(const float* d_boxes, const float* d_scores, int problems, int n, float thr, int plus_one, int ge,
                             int max_keep, int32_t* d_keep, int32_t* d_cnt, void* stream) {
    return nms_launch(d_boxes, d_scores, problems, n, thr, plus_one, ge, max_keep, d_keep, d_cnt, (hipStream_t)stream);
}

extern "C" int isegmi_op_roi_align(const float* const* d_feats, const int32_t* Hs, const int32_t* Ws, const float* scales, int nlevels,
                                   const float* d_rois, const int32_t* d_counts, int N, int K, int C, int PH, int PW, int sampling,
                                   int aligned, int k_min, int fixed_level, float* d_out, int32_t* d_out_level, void* stream) {
    return roi_align_launch(d_feats, Hs, Ws, scales, nlevels, d_rois, d_counts, N, K, C, PH, PW, sampling, k_min, fixed_level, d_out,
                            d_out_level, (hipStream_t)stream, nullptr, nullptr, aligned);
}
extern "C" int64_t isegmi_op_roi_table_bytes(int N, int K, int PH, int PW) {
    return N > 0 && K > 0 && PH > 0 && PW > 0 ? (int64_t)N * K * (2 * (PH + PW) + 1) * 16 : 0;
}
extern "C" int isegmi_op_roi_prep(const float* d_rois, const int32_t* d_counts, int N, int K, const int32_t* Hs, const int32_t* Ws, const float* scales,
                                  int nlevels, int k_min, int C, int PH, int PW, int elem_bytes, int aligned, int32_t* d_order, void* d_table,
                                  void* stream) {
    ARG_CHECK(d_rois && d_counts && Hs && Ws && scales && d_table, "args");
    return roi_prep_launch(d_rois, d_counts, N, K, Hs, Ws, scales, nlevels, k_min, C, PH, PW, elem_bytes, d_order, d_table, (hipStream_t)stream, aligned);
}
extern "C" int isegmi_op_roi_align_ordered(const float* const* d_feats, const int32_t* Hs, const int32_t* Ws, const float* scales, int nlevels,
                                           const float* d_rois, const int32_t* d_counts, const int32_t* d_order, const void* d_table, int N, int K,
                                           int C, int PH, int PW, int k_min, float* d_out, void* stream) {
    ARG_CHECK(d_feats && Hs && Ws && scales && d_rois && d_counts && d_table && d_out && N > 0 && K > 0 && PH > 0 && PW > 0, "args");
    return roi_align_launch(d_feats, Hs, Ws, scales, nlevels, d_rois, d_counts, N, K, C, PH, PW, 2, k_min, -1, d_out, nullptr, (hipStream_t)stream,
                            d_order, d_table);
}

extern "C" int isegmi_op_avgpool_full(const float* d_x, int64_t R, int HW, int C, float* d_out, void* stream) {
    ARG_CHECK(d_x && d_out && R >= 0, "args");
    return avgpool_full_launch(d_x, R, HW, C, d_out, (hipStream_t)stream);
}

extern "C" int isegmi_op_box_postprocess(const isegmi_box_post_args* a, void* stream) {
    ARG_CHECK(a, "null");
    return box_postprocess_launch(a, (hipStream_t)stream);
}

extern "C" int isegmi_op_mask_logits_select(const float* d_feat, int R, int HW, int C, const float* d_w, const float* d_b,
                                            const int32_t* d_labels, float* d_out, void* stream) {
    return mask_logits_select_launch(d_feat, R, HW, C, d_w, d_b, d_labels, d_out, (hipStream_t)stream);
}

extern "C" int isegmi_op_paste_masks(const float* d_masks, const float* d_boxes, const int32_t* d_counts, int N, int K, int M,
                                     int im_h, int im_w, float thr, uint8_t* d_out, void* stream) {
    return paste_masks_launch(d_masks, d_boxes, d_counts, N, K, M, im_h, im_w, thr, d_out, (hipStream_t)stream, nullptr, true);
}

// One RPN level for N images (parity-test entry; the engine calls the same launchers):
// head [N][HW][A*5] -> boxes/scores [N][post_nms] (+count).  Workspaces: prob [N][HWA], tk_* [N][pre_nms]; d_ws_nms: optional
// N * 128 KiB for the chip-wide NMS (pre_nms <= 1024), NULL = single-block NMS.
extern "C" int isegmi_op_grid_anchors(const float* d_base, int A, int stride, int grid_h, int grid_w, float* d_out, void* stream) {
    ARG_CHECK(d_base && d_out && A > 0 && stride > 0 && grid_h > 0 && grid_w > 0, "grid_anchors args");
    return grid_anchors_launch(d_base, A, stride, grid_h, grid_w, d_out, (hipStream_t)stream);
}

extern "C" int isegmi_op_rpn_level(const float* d_head, const float* d_anchors, const int32_t* d_image_hw, int N, int HW, int A,
                                   int pre_nms, int post_nms, float nms_thr, float min_size, int nms_ge, float* d_ws_prob,
                                   float* d_ws_tk_vals, int32_t* d_ws_tk_idx, int32_t* d_ws_tk_cnt, float* d_out_boxes,
                                   float* d_out_scores, int32_t* d_out_cnt, void* d_ws_nms, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    const int HWA = HW * A, CH = A * 5;
    int rc = rpn_sigmoid_launch(d_head, (int64_t)N * HWA, A, CH, d_ws_prob, st);
    if (rc) return rc;
    const int k = pre_nms < HWA ? pre_nms : HWA;
    rc = topk_launch(d_ws_prob, HWA, N, HWA, pre_nms, nullptr, 1, d_ws_tk_vals, d_ws_tk_idx, d_ws_tk_cnt, st);
    if (rc) return rc;
    (void)k;
    return rpn_decode_nms_launch(d_head, d_anchors, d_ws_tk_vals, d_ws_tk_idx, d_ws_tk_cnt, d_image_hw, N, HWA, A, CH, pre_nms, post_nms,
                                 nms_thr, min_size, nms_ge, 0, 1, post_nms, d_out_boxes, d_out_scores, d_out_cnt, d_ws_nms, st);
}

// All FPN levels of the RPN selection in one call (five launches): see rpn_levels_select_launch.  d_heads / d_anchors: HOST arrays of nl device pointers.
extern "C" int isegmi_op_rpn_levels(int nl, const float* const* d_heads, const float* const* d_anchors, const int32_t* HW, const int32_t* d_image_hw, int N, int A,
                                    int pre_nms, int post_nms, float nms_thr, float min_size, int nms_ge, float* d_ws_prob, float* d_ws_cand_vals,
                                    int32_t* d_ws_cand_idx, float* d_ws_tk_vals, int32_t* d_ws_tk_idx, int32_t* d_ws_tk_cnt, void* d_ws_nms,
                                    float* d_out_boxes, float* d_out_scores, int32_t* d_out_cnt, void* stream) {
    ARG_CHECK(nl >= 1 && nl <= RPN_MAX_LEVELS && d_heads && d_anchors && HW && d_image_hw, "rpn levels: null / 1-5 levels");
    int HWA[RPN_MAX_LEVELS], slot[RPN_MAX_LEVELS];
    for (int l = 0; l < nl; ++l) { HWA[l] = HW[l] * A; slot[l] = l; }
    return rpn_levels_select_launch(nl, d_heads, d_anchors, HWA, slot, d_image_hw, N, A, A * 5, pre_nms, post_nms, nms_thr, min_size, nms_ge, nl, post_nms,
                                    d_ws_prob, d_ws_cand_vals, d_ws_cand_idx, d_ws_tk_vals, d_ws_tk_idx, d_ws_tk_cnt, d_ws_nms, d_out_boxes, d_out_scores,
                                    d_out_cnt, (hipStream_t)stream);
}
extern "C" int isegmi_op_rpn_levels_workspace(int nl, int N, const int32_t* HW, int A, int pre_nms, int64_t* prob_elems, int64_t* cand_elems) {
    ARG_CHECK(nl >= 1 && nl <= RPN_MAX_LEVELS && HW, "rpn levels");
    int HWA[RPN_MAX_LEVELS];
    for (int l = 0; l < nl; ++l) HWA[l] = HW[l] * A;
    return rpn_levels_workspace(nl, N, HWA, pre_nms, prob_elems, cand_elems);
}
