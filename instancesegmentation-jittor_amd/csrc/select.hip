// select.hip -- batched top-k selection + sort on wavefront-64 primitives.
//
// Stands in for torch.topk / sort as used by RPNPostProcessor (M6: per-level top-1000 of up to
// 201 600 objectness scores), Yolact Detect (Y6: per-class top-200 of 19 248, final top-100) and
// the box post-processor (M9).  Total order = (score descending, index ascending): exactly the
// oracle's ora_topk, so indices are bit-identical.
//
// One block per problem row:
//   A. 3-pass radix select (12+12+8 bits, LDS histogram) of the threshold score T with
//      count(score > T) < k <= count(score >= T);
//   B. ordered compaction: each wave owns a contiguous segment and walks it 64 elements at a time;
//      __ballot + popcount give every survivor its rank in INDEX order, so ties at T are resolved
//      lowest-index-first without atomics (deterministic);
//   C. bitonic sort of <= KCAP 64-bit keys (ordered score << 32 | ~index) in LDS.
#include "../../include/isegmi.h"
#include "common.h"
#include "rpn_levels.h"
#include <mutex>
#include <unordered_map>

namespace isegmi {

__device__ __forceinline__ unsigned f2ord(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) {
    const unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(u);
}

struct TopkArgs {
    const float* keys;      // row r at keys + r*row_stride
    int64_t row_stride;
    int n;                  // elements per row
    int k;                  // requested k (<= KCAP)
    const int* limit;       // optional: k_eff = min(k, n, limit[r / rows_per_limit])
    int rows_per_limit;
    float* out_vals;        // [rows][k]
    int* out_idx;           // [rows][k]
    int* out_cnt;           // [rows] (optional)
    int seg_len, seg_take;  // SEG kernels: the row is nseg segments of seg_len keys, of which only the first seg_take count
    // two-level form for a few very long rows (topk_launch): level 1 runs one block per (row, slice) -- block b = row * slices + slice takes
    // keys [slice * slice_len, ...) of its row, writes its k best to out row b (indices in the full row; the tail past its own count holds
    // the smallest key, never selected); level 2 selects over a row's slices * k candidates and reports remap[row][position]
    int slices, slice_len;
    const int* remap;
    int klim;               // > 0: k_eff = min(k, n, klim) -- a host-side bound (level 2 of a row whose level-1 slices hold fewer than k real keys)
};

// several uniform top-k problems in ONE launch (the RPN's per-level rows, csrc/rcnn_ops.hip rpn_levels_select_launch): group g owns blocks
// [blk0[g], blk0[g + 1]) and runs them exactly as its own launch of `g[g]` would
constexpr int TOPK_MAX_GROUPS = 5;
struct TopkGroups {
    int ng;
    int blk0[TOPK_MAX_GROUPS + 1];
    TopkArgs g[TOPK_MAX_GROUPS];
};

// SEG: the logical row is the concatenation of the first seg_take keys of every seg_len-long segment (element e lives at
// (e / seg_take) * seg_len + e % seg_take); reported indices are positions in the FULL row, and since the mapping is
// monotonic the (score desc, index asc) order is the full row's.  For rows made of per-class lists that are already sorted
// (the final top-100 over 80 x 1000 per-class NMS outputs: nothing past a class's first 100 entries can make the top 100).
template <int NT, int KCAP, bool SEG = false>
__device__ __forceinline__ void topk_block(const TopkArgs& a, const int bx) {   // bx: the block's index inside its launch / group
    constexpr int NW = NT / 64;
    constexpr int BINS = 4096;
    constexpr int UNR = 16;
    __shared__ unsigned hist[BINS];
    __shared__ unsigned long long sbuf[KCAP];
    __shared__ unsigned wsum[NW];
    __shared__ unsigned w_gt[NW], w_eq[NW];
    __shared__ unsigned sel_digit, sel_kk;

    const int row = a.slices > 0 ? bx / a.slices : bx;
    const int orow = bx;                                           // output row
    const int slice_off = a.slices > 0 ? (bx - row * a.slices) * a.slice_len : 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* keys = a.keys + (int64_t)row * a.row_stride + slice_off;
    const int n = a.slices > 0 ? (a.n - slice_off < a.slice_len ? a.n - slice_off : a.slice_len) : a.n;
    auto phys = [&](int e) { return SEG ? (e / a.seg_take) * a.seg_len + (e % a.seg_take) : e; };
    int k_eff = a.k < n ? a.k : n;
    if (a.klim > 0 && a.klim < k_eff) k_eff = a.klim;
    if (a.limit) {
        const int l = a.limit[row / a.rows_per_limit];
        k_eff = k_eff < l ? k_eff : l;
    }
    if (k_eff <= 0) {
        if (a.slices > 0)
            for (int i = tid; i < a.k; i += NT) { a.out_vals[(int64_t)orow * a.k + i] = ord2f(0u); a.out_idx[(int64_t)orow * a.k + i] = -1; }
        if (tid == 0 && a.out_cnt) a.out_cnt[orow] = 0;
        return;
    }

    // ---- A. radix select of threshold T
    unsigned prefix = 0, kk = (unsigned)k_eff;
    for (int pass = 0; pass < 3; ++pass) {
        const int shift = pass == 0 ? 20 : (pass == 1 ? 8 : 0);
        const unsigned dmask = pass == 2 ? 0xffu : 0xfffu;
        const int pshift = pass == 0 ? 32 : (pass == 1 ? 20 : 8);  // bits above this pass's digit
        for (int i = tid; i < BINS; i += NT) hist[i] = 0;
        __syncthreads();
        // eight independent loads per thread in flight: one load per trip left every element waiting a full L2 latency
        // (270 us for a 201 600-key row whatever the key distribution -- the LDS atomics were never the cost)
        for (int i0 = tid; i0 < n; i0 += NT * UNR) {
            float kv[UNR];
#pragma unroll
            for (int q = 0; q < UNR; ++q) { const int i = i0 + q * NT; kv[q] = i < n ? keys[phys(i)] : 0.0f; }
#pragma unroll
            for (int q = 0; q < UNR; ++q) {
                const unsigned u = f2ord(kv[q]);
                const bool match = (i0 + q * NT < n) && (pshift >= 32 ? true : ((u >> pshift) == (prefix >> pshift)));
                if (match) atomicAdd(&hist[(u >> shift) & dmask], 1u);
            }
        }
        __syncthreads();
        // suffix scan over bins: thread owns BINS/NT consecutive bins
        constexpr int PER = BINS / NT;
        unsigned loc[PER], tsum = 0;
#pragma unroll
        for (int j = 0; j < PER; ++j) { loc[j] = hist[tid * PER + j]; tsum += loc[j]; }
        // inclusive suffix scan across threads (higher tid = higher bins)
        unsigned v = tsum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned o = __shfl_down(v, off, 64);
            if (lane + off < 64) v += o;
        }
        if (lane == 0) wsum[wave] = v;
        __syncthreads();
        unsigned above_waves = 0;
        for (int w = wave + 1; w < NW; ++w) above_waves += wsum[w];
        // v = sum over lanes >= lane in this wave (inclusive); exclusive-above for this thread:
        unsigned above = above_waves + (v - tsum);
#pragma unroll
        for (int j = PER - 1; j >= 0; --j) {
            if (above < kk && above + loc[j] >= kk) { sel_digit = (unsigned)(tid * PER + j); sel_kk = kk - above; }
            above += loc[j];
        }
        __syncthreads();
        prefix |= sel_digit << shift;
        kk = sel_kk;
        __syncthreads();
    }
    const unsigned T = prefix;
    const unsigned need_eq = kk;

    // ---- B. ordered compaction
    const int seg = ((n + NW - 1) / NW + 63) & ~63;
    const int s0 = wave * seg, s1 = (s0 + seg) < n ? (s0 + seg) : n;
    unsigned cgt = 0, ceq = 0;
    for (int i0 = s0 + lane; (i0 - lane) < s1; i0 += 64 * UNR) {
        float kv[UNR];
#pragma unroll
        for (int q = 0; q < UNR; ++q) { const int i = i0 + q * 64; kv[q] = i < s1 ? keys[phys(i)] : 0.0f; }
#pragma unroll
        for (int q = 0; q < UNR; ++q) {
            const bool in = i0 + q * 64 < s1;
            const unsigned u = f2ord(kv[q]);
            cgt += __popcll(__ballot(in && u > T));
            ceq += __popcll(__ballot(in && u == T));
        }
    }
    if (lane == 0) { w_gt[wave] = cgt; w_eq[wave] = ceq; }
    __syncthreads();
    unsigned gt_before = 0, eq_before = 0;
    for (int w = 0; w < wave; ++w) { gt_before += w_gt[w]; eq_before += w_eq[w]; }
    unsigned run_sel = gt_before + (eq_before < need_eq ? eq_before : need_eq);
    unsigned run_eq = eq_before;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (int i0 = s0 + lane; (i0 - lane) < s1; i0 += 64 * UNR) {
        float kv[UNR];
#pragma unroll
        for (int q = 0; q < UNR; ++q) { const int i = i0 + q * 64; kv[q] = i < s1 ? keys[phys(i)] : 0.0f; }
#pragma unroll
        for (int q = 0; q < UNR; ++q) {  // index order is kept: q walks the wave's segment 64 elements at a time
            const int i = i0 + q * 64;
            const bool in = i < s1;
            const unsigned u = f2ord(kv[q]);
            const bool gt = in && u > T, eq = in && u == T;
            const unsigned long long beq = __ballot(eq);
            const unsigned eq_rank = run_eq + (unsigned)__popcll(beq & lt_mask);
            const bool sel = gt || (eq && eq_rank < need_eq);
            const unsigned long long bsel = __ballot(sel);
            if (sel) {
                const unsigned pos = run_sel + (unsigned)__popcll(bsel & lt_mask);
                if (pos < (unsigned)KCAP) sbuf[pos] = ((unsigned long long)u << 32) | (unsigned long long)(0xffffffffu - (unsigned)phys(i));
            }
            run_sel += (unsigned)__popcll(bsel);
            run_eq += (unsigned)__popcll(beq);
        }
    }
    for (int i = k_eff + tid; i < KCAP; i += NT) sbuf[i] = 0ull;
    __syncthreads();

    if (a.slices > 0) {  // level 1 of the two-level form: the survivors leave in INDEX order, unsorted -- level 2 only needs equal keys to
                         // appear in index order (they do: the compaction is ordered and the slices follow one another), and sorts itself
        for (int i = tid; i < a.k; i += NT) {
            const unsigned long long kx = sbuf[i];
            const bool real = i < k_eff;
            a.out_vals[(int64_t)orow * a.k + i] = real ? ord2f((unsigned)(kx >> 32)) : ord2f(0u);
            a.out_idx[(int64_t)orow * a.k + i] = real ? (int)(0xffffffffu - (unsigned)(kx & 0xffffffffull)) + slice_off : -1;
        }
        return;
    }
    // ---- C. bitonic sort, descending
    for (int size = 2; size <= KCAP; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int t = tid; t < KCAP / 2; t += NT) {
                const int lo = ((t / stride) * stride * 2) + (t % stride);
                const int hi = lo + stride;
                const bool desc = ((lo & size) == 0);
                const unsigned long long x = sbuf[lo], y = sbuf[hi];
                if (desc ? (x < y) : (x > y)) { sbuf[lo] = y; sbuf[hi] = x; }
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < k_eff; i += NT) {
        const unsigned long long kx = sbuf[i];
        const int pos = (int)(0xffffffffu - (unsigned)(kx & 0xffffffffull));
        a.out_vals[(int64_t)orow * a.k + i] = ord2f((unsigned)(kx >> 32));
        a.out_idx[(int64_t)orow * a.k + i] = a.remap ? a.remap[(int64_t)row * a.n + pos] : pos;
    }
    if (tid == 0 && a.out_cnt) a.out_cnt[orow] = k_eff;
}

template <int NT, int KCAP, bool SEG = false>
__global__ __launch_bounds__(NT) void topk_kernel(const TopkArgs a) { topk_block<NT, KCAP, SEG>(a, (int)blockIdx.x); }

template <int NT, int KCAP>
__global__ __launch_bounds__(NT) void topk_groups_kernel(const TopkGroups t) {
    int g = 0;
    while (g + 1 < t.ng && (int)blockIdx.x >= t.blk0[g + 1]) ++g;   // uniform
    topk_block<NT, KCAP, false>(t.g[g], (int)blockIdx.x - t.blk0[g]);
}

// Candidate buffers of the two-level top-k for callers that bring none (the op-level C ABI): one grow-only pair per stream (calls on
// one stream are ordered).  The engines pass their own named buffers instead (topk_launch_ws), so nothing captured into a hipGraph
// or shared between engines ever points into this pool.
static int topk_scratch(hipStream_t st, size_t elems, float** vals, int** idx) {
    struct S { float* v = nullptr; int* i = nullptr; size_t cap = 0; };
    static std::mutex mu;
    static std::unordered_map<void*, S> pool;
    std::lock_guard<std::mutex> g(mu);
    S& s = pool[(void*)st];
    if (s.cap < elems) {
        if (s.v) { HIP_TRY(hipStreamSynchronize(st)); HIP_TRY(hipFree(s.v)); HIP_TRY(hipFree(s.i)); s.v = nullptr; s.i = nullptr; s.cap = 0; }
        HIP_TRY(hipMalloc((void**)&s.v, elems * sizeof(float)));
        HIP_TRY(hipMalloc((void**)&s.i, elems * sizeof(int)));
        s.cap = elems;
    }
    *vals = s.v; *idx = s.i;
    return ISEGMI_OK;
}

// Number of (value, index) candidate pairs the two-level path needs for this problem; 0 = the one-level path is taken.
static int topk_two_level_slices(int rows, int n, int k, const int* limit) {
    if (limit != nullptr || k <= 256 || k > 1024 || rows > 32 || n < 40000) return 0;  // break-even with one level at ~32 K keys (tools/topk_time.py)
    int slices = (n + 12287) / 12288;
    return slices > 32 ? 32 : slices;
}
int64_t topk_scratch_elems(int rows, int n, int k) { return (int64_t)rows * topk_two_level_slices(rows, n, k, nullptr) * k; }

// ws_vals / ws_idx: caller-owned candidate buffers of topk_scratch_elems() entries each (the engine passes named buffers so that
// captured graphs never hold pointers this file could free); NULL = the per-stream grow-only pool above (op-level calls).
int topk_launch_ws(const float* keys, int64_t row_stride, int rows, int n, int k, const int* limit, int rows_per_limit,
                   float* out_vals, int* out_idx, int* out_cnt, float* ws_vals, int* ws_idx, hipStream_t st) {
    ARG_CHECK(rows >= 0 && n >= 0 && k > 0 && k <= 8192, "topk sizes (k <= 8192)");
    if (rows == 0) return ISEGMI_OK;
    TopkArgs a{keys, row_stride, n, k, limit, rows_per_limit > 0 ? rows_per_limit : 1, out_vals, out_idx, out_cnt, 0, 0, 0, 0, nullptr, 0};
    // A few very long rows (RPN pre-NMS top-k: one row of up to 201 600 anchors per image and level): a row's five latency-bound passes in
    // ONE block took 56-108 us with the rest of the chip idle.  Two levels instead: every (row, slice of ~12 K keys) block keeps its k best
    // (select + ordered compaction, no sort), then the row's slices * k candidates go through the full kernel.  Exact: a global top-k key is
    // in its slice's top-k, and the (key desc, index asc) order survives because equal keys reach level 2 in index order.
    if (const int slices = topk_two_level_slices(rows, n, k, limit)) {
        const int slice_len = (n + slices - 1) / slices;
        float* cv = ws_vals; int* ci = ws_idx;
        if (cv == nullptr || ci == nullptr) { const int rc = topk_scratch(st, (size_t)rows * slices * k, &cv, &ci); if (rc != ISEGMI_OK) return rc; }
        TopkArgs l1 = a;
        l1.slices = slices; l1.slice_len = slice_len; l1.out_vals = cv; l1.out_idx = ci; l1.out_cnt = nullptr;
        hipLaunchKernelGGL((topk_kernel<1024, 1024>), dim3(rows * slices), dim3(1024), 0, st, l1);
        TopkArgs l2{cv, (int64_t)slices * k, slices * k, k, nullptr, 1, out_vals, out_idx, out_cnt, 0, 0, 0, 0, ci, 0};
        hipLaunchKernelGGL((topk_kernel<1024, 1024>), dim3(rows), dim3(1024), 0, st, l2);
        HIP_TRY(hipGetLastError());
        return ISEGMI_OK;
    }
    // wide blocks when the rows alone cannot fill the chip (bs=1 Detect: 80 class rows of 19 248 priors, then ONE row of 16 000
    // candidates): a row's five passes over its keys are latency-bound per thread, so 1024 threads cut them ~3x
    const bool wide = n > 65536 || (n > 8192 && rows < 256);
    if (k <= 128) {
        if (wide) hipLaunchKernelGGL((topk_kernel<1024, 128>), dim3(rows), dim3(1024), 0, st, a);
        else hipLaunchKernelGGL((topk_kernel<256, 128>), dim3(rows), dim3(256), 0, st, a);
    } else if (k <= 256) {
        if (wide) hipLaunchKernelGGL((topk_kernel<1024, 256>), dim3(rows), dim3(1024), 0, st, a);
        else hipLaunchKernelGGL((topk_kernel<256, 256>), dim3(rows), dim3(256), 0, st, a);
    } else if (k <= 1024) {
        if (n > 16384 || rows < 64) hipLaunchKernelGGL((topk_kernel<1024, 1024>), dim3(rows), dim3(1024), 0, st, a);  // few rows: the 1024-key sort wants the threads
        else hipLaunchKernelGGL((topk_kernel<256, 1024>), dim3(rows), dim3(256), 0, st, a);
    } else {  // single-map RPN: PRE_NMS_TOP_N_TEST up to 6000 (64 KB of sort keys in LDS)
        hipLaunchKernelGGL((topk_kernel<1024, 8192>), dim3(rows), dim3(1024), 0, st, a);
    }
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}

// The RPN's pre-NMS top-k of `nl` levels x N images in TWO launches (level x image as one batch dimension): level l's rows hold n[l] keys at
// keys + key_off[l] (row stride n[l]); every row is cut into slices[l] slices of ~12 K keys (rpn_topk_plan), launch 1 keeps each slice's k best in index
// order (cand_* + cand_off[l]: [N][slices[l]][k]), launch 2 selects and sorts a row's candidates into out_* [l][N][k], out_cnt [l][N] = min(k, n[l]).
// Exact for the reasons given at topk_launch_ws -- a one-slice level simply sorts in launch 2 -- so indices equal the per-level launches'.
int rpn_topk_plan(int nl, int N, const int* n, int k, int* slices, int64_t* cand_off) {
    int64_t off = 0;
    for (int l = 0; l < nl; ++l) {
        int s = (n[l] + 12287) / 12288;
        s = s < 1 ? 1 : (s > 32 ? 32 : s);
        slices[l] = s;
        cand_off[l] = off;
        off += (int64_t)N * s * k;
    }
    cand_off[nl] = off;
    return ISEGMI_OK;
}
int rpn_topk_levels_launch(int nl, int N, const float* keys, const int64_t* key_off, const int* n, int k, const int* slices, const int64_t* cand_off,
                           float* cand_vals, int* cand_idx, float* out_vals, int* out_idx, int* out_cnt, hipStream_t st) {
    ARG_CHECK(nl >= 1 && nl <= TOPK_MAX_GROUPS && N >= 1 && k > 256 && k <= 1024, "batched RPN top-k: 1-5 levels, 256 < k <= 1024");
    TopkGroups l1, l2;
    l1.ng = l2.ng = nl;
    l1.blk0[0] = l2.blk0[0] = 0;
    for (int l = 0; l < nl; ++l) {
        ARG_CHECK(n[l] > 0 && slices[l] >= 1 && slices[l] <= 32, "batched RPN top-k: level size / slices");
        const int slice_len = (n[l] + slices[l] - 1) / slices[l];
        l1.g[l] = TopkArgs{keys + key_off[l], (int64_t)n[l], n[l], k, nullptr, 1, cand_vals + cand_off[l], cand_idx + cand_off[l], nullptr, 0, 0, slices[l], slice_len, nullptr, 0};
        l1.blk0[l + 1] = l1.blk0[l] + N * slices[l];
        l2.g[l] = TopkArgs{cand_vals + cand_off[l], (int64_t)slices[l] * k, slices[l] * k, k, nullptr, 1, out_vals + (int64_t)l * N * k, out_idx + (int64_t)l * N * k,
                           out_cnt + (int64_t)l * N, 0, 0, 0, 0, cand_idx + cand_off[l], k < n[l] ? k : n[l]};
        l2.blk0[l + 1] = l2.blk0[l] + N;
    }
    hipLaunchKernelGGL((topk_groups_kernel<1024, 1024>), dim3(l1.blk0[nl]), dim3(1024), 0, st, l1);
    hipLaunchKernelGGL((topk_groups_kernel<1024, 1024>), dim3(l2.blk0[nl]), dim3(1024), 0, st, l2);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}

int topk_launch(const float* keys, int64_t row_stride, int rows, int n, int k, const int* limit, int rows_per_limit,
                float* out_vals, int* out_idx, int* out_cnt, hipStream_t st) {
    return topk_launch_ws(keys, row_stride, rows, n, k, limit, rows_per_limit, out_vals, out_idx, out_cnt, nullptr, nullptr, st);
}

// top-k over the first seg_take keys of each of nseg segments of seg_len keys per row (k <= 128); indices refer to the full row
int topk_segmented_launch(const float* keys, int64_t row_stride, int rows, int nseg, int seg_len, int seg_take, int k, const int* limit,
                          int rows_per_limit, float* out_vals, int* out_idx, int* out_cnt, hipStream_t st) {
    ARG_CHECK(rows >= 0 && nseg > 0 && seg_len > 0 && seg_take > 0 && seg_take <= seg_len && k > 0 && k <= 128, "segmented topk sizes (k <= 128)");
    if (rows == 0) return ISEGMI_OK;
    TopkArgs a{keys, row_stride, nseg * seg_take, k, limit, rows_per_limit > 0 ? rows_per_limit : 1, out_vals, out_idx, out_cnt, seg_len, seg_take, 0, 0, nullptr, 0};
    hipLaunchKernelGGL((topk_kernel<1024, 128, true>), dim3(rows), dim3(1024), 0, st, a);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}

}  // namespace isegmi

using namespace isegmi;

extern "C" int isegmi_op_topk(const float* d_keys, int64_t row_stride, int rows, int n, int k, const int* d_limit,
                              int rows_per_limit, float* d_vals, int* d_idx, int* d_cnt, void* stream) {
    ARG_CHECK(d_keys && d_vals && d_idx, "null device pointer");
    return topk_launch(d_keys, row_stride, rows, n, k, d_limit, rows_per_limit, d_vals, d_idx, d_cnt, (hipStream_t)stream);
}
