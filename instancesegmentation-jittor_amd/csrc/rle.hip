// rle.hip -- COCO run-length encoding of binary masks on the device (README.md:344-347 -> inference() ->
// COCO json, annotation layout README.md:55-66; Yolact eval.py Detections.add_mask -> dump, README.md:243-249).
//
// Restates pycocotools' maskApi.c rleEncode (column-major run lengths, starting with the run of zeros) and rleToString (delta against
// the count two runs back, 5 data bits + a continuation bit per character, + 48) -- the same two functions isegmi/coco.py restates in
// numpy (rle_counts / rle_to_string), bit for bit.  With it a step ships a few hundred KB of strings instead of the N x K x h x w uint8
// planes (242 MB for a Yolact bs=8 step).
//
// Pipeline (7 small launches on the caller's stream; every offset is an exclusive prefix sum, nothing is atomically placed, so the
// output bytes are deterministic):
//   rle_pack      one lane per image column: 64 rows -> one 64-bit word, XOR with the word shifted by one pixel (the pixel before row 0 of
//                 column x is the last pixel of column x-1) = the run STARTS of the column; popcount -> starts per column
//   rle_colscan   per mask: exclusive scan of the per-column start counts (wave shuffles) -> first output slot of every column
//   rle_maskscan  exclusive scan over masks -> run_off[m] (a mask with T run starts has T + 1 runs); overflow flag
//   rle_emit      one thread per column: bit positions -> flat start positions (x * h + y), plus the closing h * w
//   rle_len       per run: count = start[i + 1] - start[i]; x = count - count[i - 2] (i > 2); number of characters; tile sums
//   rle_tilescan  exclusive scan of the tile sums; total characters; offsets of trailing empty masks
//   rle_chars     per tile: local scan + tile offset -> the characters; str_off[m] at the first run of every mask
#include "../../include/isegmi.h"
#include "common.h"

namespace isegmi {

constexpr int RLE_TILE = 1024;  // runs per tile of the character passes (256 threads x 4 consecutive runs)

struct RleK {
    int N, K, plane_h, plane_w, qmax, cap_runs, cap_chars;
    const uint8_t* masks;
    const int* count;      // [N] valid slots per image, or NULL (all valid)
    const int* image_hw;   // [N][2] (h, w) of the window of every plane that is encoded, or NULL (the whole plane)
    const int* windows;    // [M][4] (x0, y0, x1, y1) or NULL: every set pixel of slot m lies inside [x0, x1) x [y0, y1) (the box window the paste /
                           // mask assembly kernel wrote); only that part of the plane is read, the rest is known to be zero
    unsigned long long* trans;  // [M][plane_w][qmax]
    int* col;              // [M][plane_w] starts per column -> exclusive offsets
    int* nruns;            // [M]
    int* tile;             // [cap_runs / RLE_TILE + 1]
    uint8_t* len;          // [cap_runs] characters of run i | 0x80 when i > 2
    int* run_off;          // [M + 1]
    uint32_t* counts;      // [cap_runs] start positions (rle_emit), then the run lengths in place (rle_len)
    int* str_off;          // [M + 1]
    uint8_t* chars;        // [cap_chars]
    int* status;           // [4]: total runs, total chars, overflow (1 runs, 2 chars), reserved
};

__device__ __forceinline__ void rle_dims(const RleK& p, int m, int& hi, int& wi, bool& valid) {
    const int n = m / p.K, k = m - n * p.K;
    valid = p.count == nullptr || k < p.count[n];
    hi = p.image_hw ? p.image_hw[2 * n] : p.plane_h;
    wi = p.image_hw ? p.image_hw[2 * n + 1] : p.plane_w;
}

// Which 64-row chunks of column x can hold a run start when every set pixel of slot m lies in its window [x0, x1) x [y0, y1) (clamped to the
// image; the paste / mask assembly kernel wrote it next to the plane).  A run starts at a set pixel (inside the window) or at the zero that
// follows one: the next row of the same column (rows y0 .. y1), or -- when the window touches the last row -- row 0 of the NEXT column.  So:
// columns x0 .. x1 - 1 (and x1 itself when the window touches the last row), chunks covering rows y0 .. min(y1, h - 1), plus chunk 0 when the
// window touches the last row.  Pixels outside the window are never READ (the planes need not be cleared for this kernel): a chunk's word is
// masked to the window's rows, and the column after the window is all zero by definition.
struct RleSpan {
    bool any;         // the column has chunks to visit
    bool zero_col;    // x == x1: no pixel of this column is set
    bool head;        // visit chunk 0 first although it lies above q_lo (the predecessor of row 0 may be set)
    bool prev_col;    // the last pixel of column x - 1 may be set: read it
    int q_lo, q_hi;   // chunks covering the window's rows (+ the row after it)
    int y0, y1;       // clamped window rows
};
__device__ __forceinline__ RleSpan rle_column_span(const RleK& p, int m, int x, int hi, int wi) {
    RleSpan s;
    s.any = true; s.zero_col = false; s.head = false; s.prev_col = x > 0; s.q_lo = 0; s.q_hi = (hi - 1) >> 6; s.y0 = 0; s.y1 = hi;
    if (p.windows == nullptr) return s;
    const int* w = p.windows + 4 * (int64_t)m;
    int x0 = w[0], y0 = w[1], x1 = w[2], y1 = w[3];
    x0 = x0 < 0 ? 0 : x0; y0 = y0 < 0 ? 0 : y0; x1 = x1 > wi ? wi : x1; y1 = y1 > hi ? hi : y1;
    const bool bottom = y1 >= hi;
    s.any = x1 > x0 && y1 > y0 && x >= x0 && (x < x1 || (bottom && x == x1));
    if (!s.any) return s;
    s.zero_col = x >= x1;
    s.prev_col = bottom && x > x0;                 // column x - 1 is inside the window and its last row is a window row
    s.y0 = y0; s.y1 = y1;
    if (s.zero_col) { s.q_lo = 0; s.q_hi = 0; return s; }
    s.q_lo = y0 >> 6;
    s.q_hi = (y1 < hi ? y1 : hi - 1) >> 6;
    s.head = s.prev_col && s.q_lo > 0;
    return s;
}
// bits of chunk q whose rows lie inside [y0, y1)
__device__ __forceinline__ unsigned long long rle_row_mask(int q, int y0, int y1) {
    const int lo = y0 - 64 * q, hi = y1 - 64 * q;  // window rows relative to the chunk
    if (hi <= 0 || lo >= 64) return 0ull;
    const unsigned long long upto = hi >= 64 ? ~0ull : ((1ull << hi) - 1ull);
    const unsigned long long from = lo <= 0 ? ~0ull : ~((1ull << lo) - 1ull);
    return upto & from;
}

// grid (ceil(plane_w / 256), M), 256 threads; thread = column
__global__ __launch_bounds__(256) void rle_pack_kernel(const RleK p) {
    const int m = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
    int hi, wi; bool valid;
    rle_dims(p, m, hi, wi, valid);
    if (!valid || x >= wi) return;
    const RleSpan s = rle_column_span(p, m, x, hi, wi);
    int cnt = 0;
    if (s.any) {
        const uint8_t* mp = p.masks + (int64_t)m * p.plane_h * p.plane_w;
        unsigned long long* tw = p.trans + ((int64_t)m * p.plane_w + x) * p.qmax;
        // the pixel before row 0 of this column is the last pixel of column x - 1
        const unsigned long long prev0 = (s.prev_col && mp[(int64_t)(hi - 1) * p.plane_w + x - 1] != 0) ? 1ull : 0ull;
        if (s.head) {  // chunk 0 lies above the window: all zero; its bit 0 starts a run exactly when the predecessor is set
            tw[0] = prev0;
            cnt += (int)prev0;
        }
        unsigned long long prev = s.q_lo == 0 ? prev0 : 0ull;
        for (int q = s.q_lo; q <= s.q_hi; ++q) {
            const int rows = (hi - 64 * q) < 64 ? (hi - 64 * q) : 64;
            unsigned long long word = 0;
            const unsigned long long rm = rle_row_mask(q, s.y0, s.y1);
            if (!s.zero_col && rm != 0ull) {
                const uint8_t* cp = mp + (int64_t)(64 * q) * p.plane_w + x;
                if (rm == ~0ull) {
#pragma unroll 16
                    for (int r = 0; r < 64; ++r) word |= (unsigned long long)(cp[(int64_t)r * p.plane_w] != 0) << r;
                } else {
                    const int r0 = __ffsll((long long)rm) - 1, r1 = 64 - __clzll((long long)rm);  // rm is one run of bits: rows r0 .. r1 - 1
                    for (int r = r0; r < r1; ++r) word |= (unsigned long long)(cp[(int64_t)r * p.plane_w] != 0) << r;
                }
            }
            unsigned long long t = word ^ ((word << 1) | prev);
            if (rows < 64) t &= (1ull << rows) - 1ull;
            prev = word >> 63;
            tw[q] = t;
            cnt += __popcll(t);
        }
    }
    p.col[(int64_t)m * p.plane_w + x] = cnt;
}

// block-wide exclusive scan of one int per thread (256 threads); returns the exclusive prefix, *total the block sum
__device__ __forceinline__ int block_excl_scan_256(int v, int* total, int* lds /* >= 4 ints */) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(inc, d, 64);
        if (lane >= d) inc += o;
    }
    if (lane == 63) lds[wv] = inc;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int s = lds[i];
        if (i < wv) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + inc - v;
}

// grid (M), 256 threads: col[m][0 .. wi) -> exclusive offsets; nruns[m] = starts + 1 (0 for an invalid slot)
__global__ __launch_bounds__(256) void rle_colscan_kernel(const RleK p) {
    __shared__ int lds[4];
    const int m = blockIdx.x;
    int hi, wi; bool valid;
    rle_dims(p, m, hi, wi, valid);
    if (!valid) { if (threadIdx.x == 0) p.nruns[m] = 0; return; }
    int* c = p.col + (int64_t)m * p.plane_w;
    int carry = 0;
    for (int x0 = 0; x0 < wi; x0 += 256) {
        const int x = x0 + threadIdx.x;
        const int v = x < wi ? c[x] : 0;
        int tot;
        const int ex = block_excl_scan_256(v, &tot, lds);
        if (x < wi) c[x] = carry + ex;
        carry += tot;
    }
    if (threadIdx.x == 0) p.nruns[m] = carry + 1;
}

// one block of 1024 threads: run_off = exclusive scan of nruns over the M slots
__global__ __launch_bounds__(1024) void rle_maskscan_kernel(const RleK p) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    const int M = p.N * p.K;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int m0 = 0; m0 < M; m0 += 1024) {
        const int m = m0 + threadIdx.x;
        const int v = m < M ? p.nruns[m] : 0;
        int inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        int base = carry_s, tot = 0;
        for (int i = 0; i < 16; ++i) {
            const int s = wsum[i];
            if (i < wv) base += s;
            tot += s;
        }
        if (m < M) p.run_off[m] = base + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) carry_s += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int total = carry_s;
        p.run_off[M] = total;
        p.status[0] = total;
        p.status[1] = 0;
        p.status[2] = total > p.cap_runs ? 1 : 0;
        p.status[3] = 0;
    }
}

// grid (ceil(plane_w / 256), M): start positions of column x -> counts[run_off[m] + col[m][x] ...]; the last column adds h * w
__global__ __launch_bounds__(256) void rle_emit_kernel(const RleK p) {
    const int m = blockIdx.y, x = blockIdx.x * 256 + threadIdx.x;
    int hi, wi; bool valid;
    rle_dims(p, m, hi, wi, valid);
    if (!valid || x >= wi || p.run_off[m + 1] > p.cap_runs) return;  // overflow: status[2] says so, nothing past the capacity is written
    const unsigned long long* tw = p.trans + ((int64_t)m * p.plane_w + x) * p.qmax;
    uint32_t* out = p.counts + p.run_off[m];
    int o = p.col[(int64_t)m * p.plane_w + x];
    const uint32_t colbase = (uint32_t)x * (uint32_t)hi;
    const RleSpan s = rle_column_span(p, m, x, hi, wi);   // the words rle_pack wrote for this column, in its order, and no others
    if (s.any) {
        if (s.head && tw[0]) out[o++] = colbase;
        for (int q = s.q_lo; q <= s.q_hi; ++q) {
            unsigned long long t = tw[q];
            while (t) {
                const int b = __ffsll((long long)t) - 1;
                out[o++] = colbase + (uint32_t)(64 * q + b);
                t &= t - 1ull;
            }
        }
    }
    if (x == wi - 1) out[o] = (uint32_t)hi * (uint32_t)wi;
}

__device__ __forceinline__ int rle_nchars(long long x) {
    int n = 0;
    bool more = true;
    while (more) {
        const int c = (int)(x & 0x1f);
        x >>= 5;
        more = (c & 0x10) ? (x != -1) : (x != 0);
        ++n;
    }
    return n;
}

// largest m with run_off[m] <= g among slots that own at least one run (run_off is non-decreasing; empty slots repeat a value)
__device__ __forceinline__ int rle_find_mask(const int* run_off, int M, int g) {
    int lo = 0, hi = M;  // invariant: run_off[lo] <= g < run_off[hi]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (run_off[mid] <= g) lo = mid; else hi = mid;
    }
    return lo;
}

// grid (cap_runs / RLE_TILE), 256 threads x 4 consecutive runs.  Start positions -> run lengths (in place is impossible: neighbours read
// each other's starts), so the lengths go to a second array handed in as `out_counts`.
__global__ __launch_bounds__(256) void rle_len_kernel(const RleK p, uint32_t* __restrict__ out_counts) {
    __shared__ int lds[4];
    int total = p.status[0];
    if (p.status[2] & 1) total = 0;
    const int g0 = blockIdx.x * RLE_TILE + threadIdx.x * 4;
    if (blockIdx.x * RLE_TILE >= total) return;
    const int M = p.N * p.K;
    int sum = 0;
    int m = -1, mend = 0, mbeg = 0;
    for (int e = 0; e < 4; ++e) {
        const int g = g0 + e;
        if (g >= total) break;
        if (m < 0 || g >= mend) { m = rle_find_mask(p.run_off, M, g); mbeg = p.run_off[m]; mend = p.run_off[m + 1]; }
        const int i = g - mbeg;
        const uint32_t s0 = p.counts[g];
        const uint32_t c = s0 - (i > 0 ? p.counts[g - 1] : 0u);
        long long x = (long long)c;
        if (i > 2) x -= (long long)(p.counts[g - 2] - p.counts[g - 3]);
        const int n = rle_nchars(x);
        out_counts[g] = c;
        p.len[g] = (uint8_t)(n | (i > 2 ? 0x80 : 0));
        sum += n;
    }
    int tot;
    (void)block_excl_scan_256(sum, &tot, lds);
    if (threadIdx.x == 0) p.tile[blockIdx.x] = tot;
}

// one block of 1024 threads: tile sums -> exclusive offsets; total characters; string offsets of the slots behind the last run
__global__ __launch_bounds__(1024) void rle_tilescan_kernel(const RleK p) {
    __shared__ int wsum[16];
    __shared__ int carry_s;
    int total = p.status[0];
    if (p.status[2] & 1) total = 0;
    const int ntiles = (total + RLE_TILE - 1) / RLE_TILE;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int t0 = 0; t0 < ntiles; t0 += 1024) {
        const int t = t0 + threadIdx.x;
        const int v = t < ntiles ? p.tile[t] : 0;
        int inc = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int o = __shfl_up(inc, d, 64);
            if (lane >= d) inc += o;
        }
        if (lane == 63) wsum[wv] = inc;
        __syncthreads();
        int base = carry_s, tot = 0;
        for (int i = 0; i < 16; ++i) {
            const int s = wsum[i];
            if (i < wv) base += s;
            tot += s;
        }
        if (t < ntiles) p.tile[t] = base + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) carry_s += tot;
        __syncthreads();
    }
    const int nchars = carry_s;
    if (threadIdx.x == 0) {
        p.status[1] = nchars;
        if (nchars > p.cap_chars) p.status[2] |= 2;
    }
    const int M = p.N * p.K;
    for (int m = threadIdx.x; m <= M; m += 1024)
        if (p.run_off[m] >= total) p.str_off[m] = nchars;  // slots without runs at the end (and the closing entry); all slots on overflow
}

// grid (cap_runs / RLE_TILE), 256 threads x 4 runs: characters of every run at tile offset + local offset
__global__ __launch_bounds__(256) void rle_chars_kernel(const RleK p, const uint32_t* __restrict__ counts) {
    __shared__ int lds[4];
    if (p.status[2]) return;
    const int total = p.status[0];
    if (blockIdx.x * RLE_TILE >= total) return;
    const int g0 = blockIdx.x * RLE_TILE + threadIdx.x * 4;
    int n4[4] = {0, 0, 0, 0};
    int sum = 0;
    for (int e = 0; e < 4; ++e)
        if (g0 + e < total) { n4[e] = p.len[g0 + e] & 0x7f; sum += n4[e]; }
    int tot;
    int off = p.tile[blockIdx.x] + block_excl_scan_256(sum, &tot, lds);
    const int M = p.N * p.K;
    for (int e = 0; e < 4; ++e) {
        const int g = g0 + e;
        if (g >= total) break;
        long long x = (long long)counts[g];
        if (p.len[g] & 0x80) x -= (long long)counts[g - 2];
        uint8_t* s = p.chars + off;
        bool more = true;
        int k = 0;
        while (more) {
            int c = (int)(x & 0x1f);
            x >>= 5;
            more = (c & 0x10) ? (x != -1) : (x != 0);
            if (more) c |= 0x20;
            s[k++] = (uint8_t)(c + 48);
        }
        // first run of a slot: its string starts here -- and so do the (empty) strings of the run-less slots right before it
        const int m = rle_find_mask(p.run_off, M, g);
        if (p.run_off[m] == g) {
            p.str_off[m] = off;
            for (int mm = m - 1; mm >= 0 && p.run_off[mm] == g; --mm) p.str_off[mm] = off;
        }
        off += n4[e];
    }
}

int rle_encode_launch(const isegmi_rle_args* a, hipStream_t st) {
    ARG_CHECK(a && a->d_masks && a->d_ws_trans && a->d_ws_col && a->d_ws_nruns && a->d_ws_tile && a->d_ws_len && a->d_ws_starts && a->d_out_run_off &&
              a->d_out_counts && a->d_out_str_off && a->d_out_chars && a->d_out_status, "rle: null pointer");
    ARG_CHECK(a->N > 0 && a->K > 0 && a->plane_h > 0 && a->plane_w > 0, "rle sizes");
    ARG_CHECK((int64_t)a->plane_h * a->plane_w < (1ll << 31), "rle: plane must have fewer than 2^31 pixels");
    ARG_CHECK(a->cap_runs >= RLE_TILE && a->cap_runs % RLE_TILE == 0 && a->cap_chars > 0, "rle capacities (cap_runs: a multiple of 1024)");
    RleK p;
    p.N = a->N; p.K = a->K; p.plane_h = a->plane_h; p.plane_w = a->plane_w; p.qmax = (a->plane_h + 63) / 64;
    p.cap_runs = a->cap_runs; p.cap_chars = a->cap_chars;
    p.masks = a->d_masks; p.count = a->d_count; p.image_hw = a->d_image_hw; p.windows = a->d_windows;
    p.trans = (unsigned long long*)a->d_ws_trans; p.col = a->d_ws_col; p.nruns = a->d_ws_nruns; p.tile = a->d_ws_tile; p.len = a->d_ws_len;
    p.run_off = a->d_out_run_off; p.counts = a->d_ws_starts; p.str_off = a->d_out_str_off; p.chars = a->d_out_chars; p.status = a->d_out_status;
    const int M = a->N * a->K;
    const dim3 gcol((unsigned)cdiv(a->plane_w, 256), (unsigned)M);
    hipLaunchKernelGGL(rle_pack_kernel, gcol, dim3(256), 0, st, p);
    hipLaunchKernelGGL(rle_colscan_kernel, dim3(M), dim3(256), 0, st, p);
    hipLaunchKernelGGL(rle_maskscan_kernel, dim3(1), dim3(1024), 0, st, p);
    hipLaunchKernelGGL(rle_emit_kernel, gcol, dim3(256), 0, st, p);
    const int ntiles = a->cap_runs / RLE_TILE;
    hipLaunchKernelGGL(rle_len_kernel, dim3(ntiles), dim3(256), 0, st, p, a->d_out_counts);
    hipLaunchKernelGGL(rle_tilescan_kernel, dim3(1), dim3(1024), 0, st, p);
    hipLaunchKernelGGL(rle_chars_kernel, dim3(ntiles), dim3(256), 0, st, p, (const uint32_t*)a->d_out_counts);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}

}  // namespace isegmi

using namespace isegmi;

extern "C" int isegmi_rle_workspace(int N, int K, int plane_h, int plane_w, int cap_runs, int64_t* trans_bytes, int64_t* col_bytes,
                                    int64_t* nruns_bytes, int64_t* tile_bytes, int64_t* len_bytes, int64_t* starts_bytes) {
    ARG_CHECK(N > 0 && K > 0 && plane_h > 0 && plane_w > 0 && cap_runs > 0, "rle sizes");
    const int64_t M = (int64_t)N * K;
    if (trans_bytes) *trans_bytes = M * plane_w * ((plane_h + 63) / 64) * 8;
    if (col_bytes) *col_bytes = M * plane_w * 4;
    if (nruns_bytes) *nruns_bytes = M * 4;
    if (tile_bytes) *tile_bytes = ((int64_t)cap_runs / RLE_TILE + 1) * 4;
    if (len_bytes) *len_bytes = cap_runs;
    if (starts_bytes) *starts_bytes = (int64_t)cap_runs * 4;
    return ISEGMI_OK;
}

extern "C" int isegmi_op_rle_encode(const isegmi_rle_args* a, void* stream) { return rle_encode_launch(a, (hipStream_t)stream); }
