// yolact_detect.hip -- Yolact Detect + postprocess on the GPU (App. A.6, A.9).
//
//   softmax_decode : conf [N][P][C] logits -> probabilities, written TRANSPOSED [N][C-1][P]
//                    (per-class rows are then contiguous for the top-k kernel); priors whose
//                    max foreground prob <= conf_thresh get -1 in every class row (excluded);
//                    loc+priors -> xyxy boxes; per-image kept-prior count.
//   fast_nms       : per (image, class): box j survives iff max_{i<j} IoU(i,j) <= thr over the
//                    class's score-sorted top-k (IoU without +1, NaN drops the box); with
//                    second_threshold also iff its class score > conf_thresh.
//   gather         : final per-image top-100 -> boxes / scores / classes / 32 mask coefficients.
//   proto masks    : m = sigmoid(fmaf-chain_k proto*coeff) cropped to the box (+1 px) in proto space,
//                    then bilinear (align_corners=False) to (h,w), > 0.5 -> uint8.
// Reference anchors: README.md:243-249 (eval.py --score_threshold/--top_k reach Detect+postprocess).
#include "../../include/isegmi.h"
#include "common.h"
#include "detmath.h"

namespace isegmi {

constexpr int SM_ROWS = 128;

// Addressing of the head outputs: separate contiguous buffers (pix_stride == 0) or the fused per-pixel row layout.
struct HeadLayout {
    int A;
    int64_t pix_stride;
    int off_loc, off_conf, off_mask, mask_tanh;
};
__device__ __forceinline__ const float* head_ptr(const float* base, const HeadLayout& L, int n, int P, int p, int width, int off) {
    if (L.pix_stride == 0) return base + ((int64_t)n * P + p) * width;
    const int pix = p / L.A, a = p - pix * L.A;
    return base + ((int64_t)n * (P / L.A) + pix) * L.pix_stride + off + a * width;
}

// SM_PARTS threads per prior row (tid = part * SM_ROWS + row, so a wave still holds 64 consecutive rows: conflict-free LDS
// rows of C floats, coalesced transposed stores): the exponentials, divisions and the order-free max / foreground-max are split
// over the parts; only the SUM keeps the oracle's sequential class order (part 0 adds the row's exponentials from LDS).
// One thread per row left the 19 248-prior softmax of a single image on 302 waves with ~3700 dependent instructions each.
constexpr int SM_PARTS = 4;
__global__ __launch_bounds__(SM_ROWS * SM_PARTS) void yolact_softmax_decode_kernel(
    const float* __restrict__ conf, const float* __restrict__ loc, const float* __restrict__ priors, int P, int C,
    float conf_thresh, const HeadLayout L, float* __restrict__ scoresT, float* __restrict__ boxes, int* __restrict__ keep_count) {
    extern __shared__ float sm[];  // [SM_ROWS][C] rows, then [SM_PARTS][SM_ROWS] partial maxima, then [SM_ROWS] sums
    constexpr int NT = SM_ROWS * SM_PARTS;
    float* pmax = sm + SM_ROWS * C;
    float* rsum = pmax + SM_PARTS * SM_ROWS;
    const int n = blockIdx.y;
    const int p0 = blockIdx.x * SM_ROWS;
    const int rows = (P - p0) < SM_ROWS ? (P - p0) : SM_ROWS;
    if (L.pix_stride == 0) {
        const float* src = conf + ((int64_t)n * P + p0) * C;
        for (int i = threadIdx.x; i < rows * C; i += NT) sm[i] = src[i];
    } else {
        for (int i = threadIdx.x; i < rows * C; i += NT) {
            const int r = i / C, c = i - r * C;
            sm[i] = head_ptr(conf, L, n, P, p0 + r, C, L.off_conf)[c];
        }
    }
    __syncthreads();
    const int t = threadIdx.x % SM_ROWS, part = threadIdx.x / SM_ROWS;
    const int p = p0 + t;
    const bool live = t < rows;
    float* r = sm + t * C;
    {   // max over the row: partial per part, then combined (order-free)
        float m = -3.0e38f;
        if (live) for (int c = part; c < C; c += SM_PARTS) m = r[c] > m ? r[c] : m;
        pmax[part * SM_ROWS + t] = m;
    }
    __syncthreads();
    float m = pmax[t];
#pragma unroll
    for (int q = 1; q < SM_PARTS; ++q) { const float v = pmax[q * SM_ROWS + t]; m = v > m ? v : m; }
    if (live) for (int c = part; c < C; c += SM_PARTS) r[c] = dm_exp(r[c] - m);
    __syncthreads();
    if (live && part == 0) {
        float s = 0.0f;
        for (int c = 0; c < C; ++c) s = s + r[c];
        rsum[t] = s;
    }
    __syncthreads();
    {   // probabilities in place; foreground max per part
        float fg = -1.0f;
        if (live) {
            const float s = rsum[t];
            for (int c = part; c < C; c += SM_PARTS) { const float pr = dm_div(r[c], s); r[c] = pr; if (c >= 1) fg = pr > fg ? pr : fg; }
        }
        pmax[part * SM_ROWS + t] = fg;
    }
    __syncthreads();
    float fg = pmax[t];
#pragma unroll
    for (int q = 1; q < SM_PARTS; ++q) { const float v = pmax[q * SM_ROWS + t]; fg = v > fg ? v : fg; }
    const bool kept = live && fg > conf_thresh;
    if (live && part == 1) {  // decode (one part per row; not the one that did the sum)
        const float* lp = head_ptr(loc, L, n, P, p, 4, L.off_loc);
        const float4 l = make_float4(lp[0], lp[1], lp[2], lp[3]);
        const float4 q = *(const float4*)(priors + (int64_t)p * 4);
        float tx = l.x * 0.1f; tx = tx * q.z;
        float ty = l.y * 0.1f; ty = ty * q.w;
        const float cx = q.x + tx, cy = q.y + ty;
        const float w = q.z * dm_exp(l.z * 0.2f), h = q.w * dm_exp(l.w * 0.2f);
        const float x1 = cx - dm_div(w, 2.0f), y1 = cy - dm_div(h, 2.0f);
        *(float4*)(boxes + ((int64_t)n * P + p) * 4) = make_float4(x1, y1, w + x1, h + y1);
    }
    const int cnt = __syncthreads_count((kept && part == 0) ? 1 : 0);
    if (threadIdx.x == 0 && cnt) atomicAdd(&keep_count[n], cnt);
    if (live) {
        float* dst = scoresT + (int64_t)n * (C - 1) * P + p;
        for (int c = 1 + part; c < C; c += SM_PARTS) dst[(int64_t)(c - 1) * P] = kept ? r[c] : -1.0f;
    }
}

__device__ __forceinline__ float jaccard(const float4 a, const float4 b) {
    const float mx2 = a.z < b.z ? a.z : b.z, mx1 = a.x > b.x ? a.x : b.x;
    const float my2 = a.w < b.w ? a.w : b.w, my1 = a.y > b.y ? a.y : b.y;
    float iw = mx2 - mx1, ih = my2 - my1;
    iw = iw > 0.0f ? iw : 0.0f;
    ih = ih > 0.0f ? ih : 0.0f;
    const float inter = iw * ih;
    const float aa = (a.z - a.x) * (a.w - a.y), ab = (b.z - b.x) * (b.w - b.y);
    const float uni = aa + ab - inter;
    return dm_div(inter, uni);
}

// grid (ncls_fg, N); block 256. topk_idx/vals: [N][nc][top_k]; cnt: [N][nc].
// Four threads per box (tid = part * 256 + j): "no earlier box of the class overlaps box j by more than thr" is an AND over the
// earlier boxes, so each part tests every fourth one and the verdicts are combined in LDS -- a 4x shorter dependent chain than
// one thread walking up to 199 IEEE divisions (31 -> ~15 us on one image's 80 class blocks).
__global__ __launch_bounds__(1024) void yolact_fast_nms_kernel(const float* __restrict__ boxes, const float* __restrict__ tk_vals,
                                                                const int* __restrict__ tk_idx, const int* __restrict__ tk_cnt,
                                                                int P, int nc, int top_k, float thr, int second_threshold, float conf_thresh,
                                                                float* __restrict__ cand, int* __restrict__ kept_count) {
    __shared__ float4 sb[256];
    __shared__ unsigned char okp[4][256];
    const int c = blockIdx.x, n = blockIdx.y;
    const int base = (n * nc + c) * top_k;
    const int cnt = tk_cnt[n * nc + c];
    const int j = threadIdx.x & 255, part = threadIdx.x >> 8;
    if (part == 0 && j < cnt) sb[j] = *(const float4*)(boxes + ((int64_t)n * P + tk_idx[base + j]) * 4);
    __syncthreads();
    bool ok = true;
    if (j < cnt) {
        const float4 bj = sb[j];
        for (int i = part; i < j; i += 4) {
            const float o = jaccard(sb[i], bj);
            if (!(o <= thr)) { ok = false; break; }
        }
    }
    okp[part][j] = ok ? 1 : 0;
    __syncthreads();
    // App. A.6 fork: fast_nms(second_threshold=True) additionally drops a box whose OWN class score is not above conf_thresh (a prior passes the
    // pre-filter on its best class and is then ranked in every class); off in the default detect() call
    bool keep = part == 0 && j < cnt && okp[0][j] && okp[1][j] && okp[2][j] && okp[3][j];
    if (second_threshold && keep) keep = tk_vals[base + j] > conf_thresh;
    if (part == 0 && j < top_k) cand[base + j] = keep ? tk_vals[base + j] : -1.0f;
    const int k = __syncthreads_count(keep ? 1 : 0);
    if (threadIdx.x == 0 && k) atomicAdd(&kept_count[n], k);
}

// grid (N); block 128: one thread per output slot.
__global__ void yolact_gather_kernel(const float* __restrict__ boxes, const float* __restrict__ mask, const int* __restrict__ tk_idx,
                                     const float* __restrict__ fin_vals, const int* __restrict__ fin_idx,
                                     const int* __restrict__ fin_cnt, int P, int nc, int top_k, int mask_dim, int max_det,
                                     const HeadLayout L,
                                     int* __restrict__ out_count, float* __restrict__ out_boxes, float* __restrict__ out_scores,
                                     int* __restrict__ out_classes, float* __restrict__ out_coeffs, int* __restrict__ out_prior) {
    const int n = blockIdx.x;
    const int cnt = fin_cnt[n];
    if (threadIdx.x == 0) out_count[n] = cnt;
    for (int q = threadIdx.x; q < max_det; q += blockDim.x) {
        const int64_t o = (int64_t)n * max_det + q;
        if (q < cnt) {
            const int flat = fin_idx[o];
            const int c = flat / top_k;
            const int prior = tk_idx[(int64_t)n * nc * top_k + flat];
            *(float4*)(out_boxes + o * 4) = *(const float4*)(boxes + ((int64_t)n * P + prior) * 4);
            out_scores[o] = fin_vals[o];
            out_classes[o] = c;
            out_prior[o] = prior;
            const float* mp = head_ptr(mask, L, n, P, prior, mask_dim, L.off_mask);
            for (int k = 0; k < mask_dim; ++k) out_coeffs[o * mask_dim + k] = L.mask_tanh ? dm_tanh(mp[k]) : mp[k];
        } else {
            *(float4*)(out_boxes + o * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
            out_scores[o] = 0.0f;
            out_classes[o] = -1;
            out_prior[o] = -1;
            for (int k = 0; k < mask_dim; ++k) out_coeffs[o * mask_dim + k] = 0.0f;
        }
    }
}

__device__ __forceinline__ void sanitize(float a, float b, int img, float padding, float& o1, float& o2) {
    a = a * (float)img; b = b * (float)img;
    float lo = a < b ? a : b, hi = a > b ? a : b;
    lo = lo - padding; hi = hi + padding;
    lo = lo > 0.0f ? lo : 0.0f;
    hi = hi < (float)img ? hi : (float)img;
    o1 = lo; o2 = hi;
}

// proto [N][PH][PW][32]; coeffs [N][K][32]; boxes [N][K][4]; lo [N][K][PH][PW].
// grid (ceil(PH*PW/256), N); one thread per proto pixel keeps its 32 prototype values in registers
// and loops over the image's detections (coefficients + crop windows staged in LDS).
// DENSE = false (round 5): a pixel outside a detection's crop window is neither computed nor written -- the upsampling kernel below takes such a tap as
// the 0 it is by its own window test -- so `lo` holds the window's pixels only (the rest is stale): 62 MB of zeros per bs = 8 step were written here and
// a dot product + sigmoid spent on each.  A wave none of whose 64 pixels lies in the window skips the detection.  DENSE = true keeps the full
// zero-padded masks (YOLACT++'s MaskIoU head convolves them).
constexpr int MD = 32;
constexpr int PROTO_DG = 25;
template <bool DENSE>
__global__ __launch_bounds__(256) void yolact_proto_masks_kernel(const float* __restrict__ proto, const float* __restrict__ coeffs,
                                                                  const float* __restrict__ boxes, const int* __restrict__ count,
                                                                  int PH, int PW, int K, float* __restrict__ lo) {
    extern __shared__ float sm[];  // [K][MD] coeffs + [K][4] windows
    const int n = blockIdx.y;
    // grid.z splits the image's detections into groups of PROTO_DG: at bs=1 the 75 pixel blocks alone left one wave per SIMD
    // walking 100 detections serially (52 us)
    const int d_lo = blockIdx.z * PROTO_DG;
    const int cnt = count[n] < d_lo + PROTO_DG ? count[n] : d_lo + PROTO_DG;
    if (d_lo >= cnt) return;
    float* sc = sm;
    float* sw = sm + K * MD;
    for (int i = d_lo * MD + threadIdx.x; i < cnt * MD; i += 256) sc[i] = coeffs[(int64_t)n * K * MD + i];
    for (int d = d_lo + threadIdx.x; d < cnt; d += 256) {
        const float4 b = *(const float4*)(boxes + ((int64_t)n * K + d) * 4);
        float x1, x2, y1, y2;
        sanitize(b.x, b.z, PW, 1.0f, x1, x2);
        sanitize(b.y, b.w, PH, 1.0f, y1, y2);
        sw[d * 4 + 0] = x1; sw[d * 4 + 1] = x2; sw[d * 4 + 2] = y1; sw[d * 4 + 3] = y2;
    }
    __syncthreads();
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= PH * PW) return;
    const int y = pix / PW, x = pix - y * PW;
    float pv[MD];
    const float4* ps = (const float4*)(proto + ((int64_t)n * PH * PW + pix) * MD);
#pragma unroll
    for (int k = 0; k < MD / 4; ++k) { const float4 v = ps[k]; pv[4 * k] = v.x; pv[4 * k + 1] = v.y; pv[4 * k + 2] = v.z; pv[4 * k + 3] = v.w; }
    const float fx = (float)x, fy = (float)y;
    for (int d = d_lo; d < cnt; ++d) {
        const bool inside = fx >= sw[d * 4] && fx < sw[d * 4 + 1] && fy >= sw[d * 4 + 2] && fy < sw[d * 4 + 3];
        if (!DENSE && !inside) continue;
        const float* cf = sc + d * MD;
        float acc = 0.0f;
#pragma unroll
        for (int k = 0; k < MD; ++k) acc = fmaf(pv[k], cf[k], acc);
        const float v = dm_sigmoid(acc);
        lo[(((int64_t)n * K + d) * PH * PW) + pix] = inside ? v : 0.0f;
    }
}

// out [N][K][h][w] uint8.  Pixels whose bilinear taps all fall outside the detection's crop window are exactly 0, so the plane is
// cleared by one hipMemsetAsync (pure streaming writes at the chip's fill rate) and this kernel then writes only each detection's
// conservative output-space window: typically < 20 % of the plane.  (A single-pass kernel over the flat plane was measured at 1.3
// TB/s whatever its store width: a wave's 64 lanes span two image rows, so nearly every wave met some window and walked the
// per-pixel path with most lanes idle.)  grid (chunks, K, N); one thread per window pixel, byte stores coalesced by the lanes of
// a row segment; the four taps come from the proto-resolution masks (L2 / Infinity Cache resident).
// image_hw (optional, [N][2]): image n is upsampled to ITS (h_n, w_n) inside the common (h, w) plane (a batch of images of different
// original sizes, each postprocess()ed at its own size as upstream's per-image evalimage does); NULL = every image at (h, w).
__global__ __launch_bounds__(256) void yolact_upsample_masks_kernel(const float* __restrict__ lo, const float* __restrict__ boxes,
                                                                     const int* __restrict__ count, int PH, int PW, int K, int h, int w,
                                                                     const int* __restrict__ image_hw, uint8_t* __restrict__ out, int* __restrict__ win) {
    const int n = blockIdx.z, d = blockIdx.y;
    if (d >= count[n]) return;
    const int plane_w = w;
    const int64_t plane = (int64_t)h * w;
    if (image_hw) { h = image_hw[2 * n]; w = image_hw[2 * n + 1]; }
    const float4 b = *(const float4*)(boxes + ((int64_t)n * K + d) * 4);
    float x1, x2, y1, y2;
    sanitize(b.x, b.z, PW, 1.0f, x1, x2);
    sanitize(b.y, b.w, PH, 1.0f, y1, y2);
    // proto pixels i with x1 <= i < x2 are inside; an output pixel can be non-zero only if one of its two
    // taps (floor(src), floor(src)+1) is inside.  +-2 output pixels of slack keep this a strict superset.
    const float fsx = (float)w / (float)PW, fsy = (float)h / (float)PH;
    int ox0 = (int)floorf((ceilf(x1) - 1.0f + 0.5f) * fsx - 0.5f) - 2, ox1 = (int)ceilf((ceilf(x2) + 0.5f) * fsx - 0.5f) + 2;
    int oy0 = (int)floorf((ceilf(y1) - 1.0f + 0.5f) * fsy - 0.5f) - 2, oy1 = (int)ceilf((ceilf(y2) + 0.5f) * fsy - 0.5f) + 2;
    ox0 = ox0 < 0 ? 0 : ox0; ox1 = ox1 > w ? w : ox1;
    oy0 = oy0 < 0 ? 0 : oy0; oy1 = oy1 > h ? h : oy1;
    const int ww = ox1 - ox0, wh = oy1 - oy0;
    // the window every set pixel of this plane lies in (the run-length encoder reads nothing else of the plane)
    if (win != nullptr && blockIdx.x == 0 && threadIdx.x == 0) { int* wq = win + ((int64_t)n * K + d) * 4; wq[0] = ox0; wq[1] = oy0; wq[2] = ox1; wq[3] = oy1; }
    if (ww <= 0 || wh <= 0) return;
    const int area = ww * wh;
    const float sy = dm_div((float)PH, (float)h), sx = dm_div((float)PW, (float)w);  // dm_bil_coef's scale, hoisted
    const float* m = lo + ((int64_t)n * K + d) * PH * PW;
    uint8_t* o = out + ((int64_t)n * K + d) * plane;
    // a tap outside the crop window is 0 by definition (the crop): read as such, never from `lo` (which holds the window's pixels only)
    auto tap = [&](int yy, int xx) -> float {
        const float fx = (float)xx, fy = (float)yy;
        return (fx >= x1 && fx < x2 && fy >= y1 && fy < y2) ? m[yy * PW + xx] : 0.0f;
    };
    for (int i = blockIdx.x * 256 + threadIdx.x; i < area; i += gridDim.x * 256) {
        const int ry = i / ww;
        const int y = oy0 + ry, x = ox0 + (i - ry * ww);
        int y0, yb, x0, xb; float ly0, ly1, lx0, lx1;
        dm_bil_coef_s(y, PH, sy, y0, yb, ly0, ly1);
        dm_bil_coef_s(x, PW, sx, x0, xb, lx0, lx1);
        float top = lx0 * tap(y0, x0); top = fmaf(lx1, tap(y0, xb), top);
        float bot = lx0 * tap(yb, x0); bot = fmaf(lx1, tap(yb, xb), bot);
        float v = ly0 * top; v = fmaf(ly1, bot, v);
        o[y * plane_w + x] = v > 0.5f ? (uint8_t)1 : (uint8_t)0;
    }
}

// integer boxes (A.9 last line): sanitize(pad 0) against (w,h) then truncate to int64.
__global__ void yolact_int_boxes_kernel(const float* __restrict__ boxes, const int* __restrict__ count, int K, int h, int w,
                                        const int* __restrict__ image_hw, int64_t* __restrict__ out) {
    const int n = blockIdx.x;
    if (image_hw) { h = image_hw[2 * n]; w = image_hw[2 * n + 1]; }
    for (int d = threadIdx.x; d < K; d += blockDim.x) {
        int64_t* o = out + ((int64_t)n * K + d) * 4;
        if (d < count[n]) {
            const float4 b = *(const float4*)(boxes + ((int64_t)n * K + d) * 4);
            float x1, x2, y1, y2;
            sanitize(b.x, b.z, w, 0.0f, x1, x2);
            sanitize(b.y, b.w, h, 0.0f, y1, y2);
            o[0] = (int64_t)x1; o[1] = (int64_t)y1; o[2] = (int64_t)x2; o[3] = (int64_t)y2;
        } else { o[0] = o[1] = o[2] = o[3] = 0; }
    }
}

int topk_launch(const float* keys, int64_t row_stride, int rows, int n, int k, const int* limit, int rows_per_limit,
                float* out_vals, int* out_idx, int* out_cnt, hipStream_t st);

int yolact_detect_launch(const isegmi_yolact_detect_args* a, hipStream_t st) {
    ARG_CHECK(a->N > 0 && a->P > 0 && a->ncls >= 2 && a->ncls <= 256, "detect sizes");
    ARG_CHECK(a->top_k > 0 && a->top_k <= 256 && a->max_det > 0 && a->max_det <= 128, "top_k<=256, max_det<=128");
    ARG_CHECK(a->mask_dim > 0, "mask_dim");
    const int nc = a->ncls - 1;
    HIP_TRY(hipMemsetAsync(a->d_ws_counts, 0, sizeof(int) * 2 * (size_t)a->N, st));
    int* keep_count = a->d_ws_counts;
    int* kept2 = a->d_ws_counts + a->N;
    const size_t lds = ((size_t)SM_ROWS * a->ncls + (size_t)(SM_PARTS + 1) * SM_ROWS) * sizeof(float);
    HeadLayout L;
    L.A = a->A > 0 ? a->A : 1; L.pix_stride = a->pix_stride; L.off_loc = a->off_loc; L.off_conf = a->off_conf; L.off_mask = a->off_mask;
    L.mask_tanh = a->mask_tanh;
    ARG_CHECK(a->pix_stride == 0 || (a->A > 0 && a->P % a->A == 0), "fused head layout needs A > 0 and P % A == 0");
    hipLaunchKernelGGL(yolact_softmax_decode_kernel, dim3(cdiv(a->P, SM_ROWS), a->N), dim3(SM_ROWS * SM_PARTS), lds, st, a->d_conf, a->d_loc,
                       a->d_priors, a->P, a->ncls, a->conf_thresh, L, a->d_ws_scoresT, a->d_ws_boxes, keep_count);
    HIP_TRY(hipGetLastError());
    int rc = topk_launch(a->d_ws_scoresT, a->P, a->N * nc, a->P, a->top_k, keep_count, nc, a->d_ws_tk_vals, a->d_ws_tk_idx,
                         a->d_ws_tk_cnt, st);
    if (rc) return rc;
    hipLaunchKernelGGL(yolact_fast_nms_kernel, dim3(nc, a->N), dim3(1024), 0, st, a->d_ws_boxes, a->d_ws_tk_vals, a->d_ws_tk_idx,
                       a->d_ws_tk_cnt, a->P, nc, a->top_k, a->nms_thresh, a->second_threshold, a->conf_thresh, a->d_ws_cand, kept2);
    HIP_TRY(hipGetLastError());
    rc = topk_launch(a->d_ws_cand, (int64_t)nc * a->top_k, a->N, nc * a->top_k, a->max_det, kept2, 1, a->d_ws_fin_vals,
                     a->d_ws_fin_idx, a->d_ws_fin_cnt, st);
    if (rc) return rc;
    hipLaunchKernelGGL(yolact_gather_kernel, dim3(a->N), dim3(128), 0, st, a->d_ws_boxes, a->d_mask, a->d_ws_tk_idx, a->d_ws_fin_vals,
                       a->d_ws_fin_idx, a->d_ws_fin_cnt, a->P, nc, a->top_k, a->mask_dim, a->max_det, L, a->d_out_count,
                       a->d_out_boxes, a->d_out_scores, a->d_out_classes, a->d_out_coeffs, a->d_out_prior);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}

// ---- YOLACT++ fast mask re-scoring (FastMaskIoUNet) ends: the first layer (one input channel: the cropped proto-resolution
// mask) and the final global-max + class pick; the four 3x3 layers and the 1x1 in between run on the MFMA conv kernels.
// lo [NK][PH][PW]; w [8][9] (KRSC with C = 1), b [8]; out [NK][Ho][Wo][32], channels 8..31 zero (the next conv's Cin is padded
// to 32 with zero weights: exact, the extra products are +-0).  k-ordered fmaf chain over the nine taps, + bias, ReLU.
__global__ __launch_bounds__(256) void maskiou_conv1_kernel(const float* __restrict__ lo, int64_t total, int PH, int PW, int Ho, int Wo,
                                                            const float* __restrict__ w, const float* __restrict__ b,
                                                            float* __restrict__ out) {
    __shared__ float sw[72], sb[8];
    if (threadIdx.x < 72) sw[threadIdx.x] = w[threadIdx.x];
    if (threadIdx.x < 8) sb[threadIdx.x] = b[threadIdx.x];
    __syncthreads();
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int wo = (int)(t % Wo);
        const int64_t q = t / Wo;
        const int ho = (int)(q % Ho);
        const int64_t img = q / Ho;
        const float* src = lo + (img * PH + 2 * ho) * PW + 2 * wo;
        float x[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) x[r * 3 + c] = src[r * PW + c];
        float y[8];
#pragma unroll
        for (int co = 0; co < 8; ++co) {
            float acc = 0.0f;
#pragma unroll
            for (int k = 0; k < 9; ++k) acc = fmaf(x[k], sw[co * 9 + k], acc);
            const float v = fmaf(acc, 1.0f, sb[co]) + 0.0f;
            y[co] = v > 0.0f ? v : 0.0f;
        }
        float4* d = (float4*)(out + t * 32);
        d[0] = make_float4(y[0], y[1], y[2], y[3]);
        d[1] = make_float4(y[4], y[5], y[6], y[7]);
        const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 2; k < 8; ++k) d[k] = z;
    }
}
// feat [N*K][HW][C] (post-ReLU class maps); mask_score[n][d] = score * max over HW of feat[.][.][class] for d < count[n], else 0
__global__ void maskiou_rescore_kernel(const float* __restrict__ feat, int HW, int C, const int* __restrict__ cls,
                                       const float* __restrict__ score, const int* __restrict__ count, int K,
                                       float* __restrict__ out) {
    const int n = blockIdx.x;
    for (int d = threadIdx.x; d < K; d += blockDim.x) {
        float r = 0.0f;
        if (d < count[n]) {
            const float* f = feat + ((int64_t)n * K + d) * HW * C + cls[n * K + d];
            float m = f[0];
            for (int p = 1; p < HW; ++p) { const float v = f[(int64_t)p * C]; m = v > m ? v : m; }
            r = score[n * K + d] * m;
        }
        out[n * K + d] = r;
    }
}
int maskiou_conv1_launch(const float* lo, int NK, int PH, int PW, const float* w, const float* b, float* out, hipStream_t st) {
    ARG_CHECK(PH >= 3 && PW >= 3, "mask-IoU net input smaller than 3x3");
    const int Ho = (PH - 3) / 2 + 1, Wo = (PW - 3) / 2 + 1;
    const int64_t total = (int64_t)NK * Ho * Wo;
    int64_t blocks = cdiv64(total, 256);
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(maskiou_conv1_kernel, dim3((unsigned)blocks), dim3(256), 0, st, lo, total, PH, PW, Ho, Wo, w, b, out);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}
int maskiou_rescore_launch(const float* feat, int N, int K, int HW, int C, const int* cls, const float* score, const int* count,
                           float* out, hipStream_t st) {
    hipLaunchKernelGGL(maskiou_rescore_kernel, dim3(N), dim3(128), 0, st, feat, HW, C, cls, score, count, K, out);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}

int yolact_masks_launch(const float* proto, const float* coeffs, const float* boxes, const int* count, int N, int PH, int PW,
                        int mask_dim, int K, int h, int w, float* ws_lo, uint8_t* out_masks, int64_t* out_boxes, hipStream_t st,
                        const int* image_hw, int* win, bool clear, bool dense_lo) {
    ARG_CHECK(mask_dim == MD, "mask_dim must be 32");
    ARG_CHECK(N > 0 && K > 0 && K <= 128 && h > 0 && w > 0, "mask sizes");
    const size_t lds = (size_t)K * (MD + 4) * sizeof(float);
    // dense_lo: ws_lo receives the complete zero-padded proto-resolution masks (a consumer other than the upsampling below reads them)
    if (dense_lo)
        hipLaunchKernelGGL(yolact_proto_masks_kernel<true>, dim3(cdiv(PH * PW, 256), N, cdiv(K, PROTO_DG)), dim3(256), lds, st, proto, coeffs, boxes, count,
                           PH, PW, K, ws_lo);
    else
        hipLaunchKernelGGL(yolact_proto_masks_kernel<false>, dim3(cdiv(PH * PW, 256), N, cdiv(K, PROTO_DG)), dim3(256), lds, st, proto, coeffs, boxes, count,
                           PH, PW, K, ws_lo);
    HIP_TRY(hipGetLastError());
    ARG_CHECK((int64_t)h * w < (1ll << 31), "h * w must stay below 2^31");
    if (clear) HIP_TRY(hipMemsetAsync(out_masks, 0, (size_t)N * K * h * w, st));  // skipped when the planes are read through their windows only
    // chunks per detection: a full-image window is h*w pixels; 256 threads x ~8 pixels per thread and chunk
    int chunks = (int)cdiv64((int64_t)h * w, 2048);
    if (chunks > 32) chunks = 32;
    hipLaunchKernelGGL(yolact_upsample_masks_kernel, dim3((unsigned)chunks, (unsigned)K, (unsigned)N), dim3(256), 0, st, ws_lo, boxes, count, PH, PW, K, h, w,
                       image_hw, out_masks, win);
    HIP_TRY(hipGetLastError());
    if (out_boxes) {
        hipLaunchKernelGGL(yolact_int_boxes_kernel, dim3(N), dim3(128), 0, st, boxes, count, K, h, w, image_hw, out_boxes);
        HIP_TRY(hipGetLastError());
    }
    return ISEGMI_OK;
}

}  // namespace isegmi

using namespace isegmi;

extern "C" int isegmi_op_yolact_detect(const isegmi_yolact_detect_args* a, void* stream) {
    ARG_CHECK(a, "null args");
    return yolact_detect_launch(a, (hipStream_t)stream);
}
extern "C" int isegmi_op_yolact_masks(const float* d_proto, const float* d_coeffs, const float* d_boxes, const int* d_count,
                                      int N, int PH, int PW, int mask_dim, int K, int h, int w, float* d_ws_lo,
                                      uint8_t* d_out_masks, int64_t* d_out_boxes, void* stream) {
    return yolact_masks_launch(d_proto, d_coeffs, d_boxes, d_count, N, PH, PW, mask_dim, K, h, w, d_ws_lo, d_out_masks, d_out_boxes,
                               (hipStream_t)stream, nullptr, nullptr, true, false);
}
