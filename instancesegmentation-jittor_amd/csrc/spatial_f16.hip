// spatial_f16.hip -- fp16-storage variants of the HBM-bound NHWC ops used by the fp16 Mask R-CNN path
// (BASELINE configs[4]): max-pool (fp32 or fp16 in -> fp16 out), nearest2x + add, LevelMapper + RoIAlign and the
// class-selected mask 1x1.  All arithmetic is fp32 (same operation order as the fp32 kernels / the oracle); only
// loads and stores convert, 4 channels (8 B) per lane.
#include "../../include/isegmi.h"
#include "common.h"
#include "detmath.h"

namespace isegmi {

typedef _Float16 half_t;
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 ld4(const half_t* p) { const h4 v = *(const h4*)p; return make_float4((float)v.x, (float)v.y, (float)v.z, (float)v.w); }
__device__ __forceinline__ float4 ld4(const float* p) { return *(const float4*)p; }
__device__ __forceinline__ void st4(half_t* p, float4 v) { h4 o; o.x = (half_t)v.x; o.y = (half_t)v.y; o.z = (half_t)v.z; o.w = (half_t)v.w; *(h4*)p = o; }

template <typename TI>
__global__ void maxpool_to_f16_kernel(const TI* __restrict__ in, int N, int H, int W, int C, int k, int s, int p, int Ho, int Wo,
                                      half_t* __restrict__ out) {
    const int c4n = C >> 2;
    const int64_t total = (int64_t)N * Ho * Wo * c4n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        int64_t t = i / c4n;
        const int wo = (int)(t % Wo); t /= Wo;
        const int ho = (int)(t % Ho);
        const int n = (int)(t / Ho);
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        for (int r = 0; r < k; ++r) {
            const int hi = ho * s + r - p;
            if ((unsigned)hi >= (unsigned)H) continue;
            for (int q = 0; q < k; ++q) {
                const int wi = wo * s + q - p;
                if ((unsigned)wi >= (unsigned)W) continue;
                const float4 v = ld4(in + (((int64_t)n * H + hi) * W + wi) * C + c4 * 4);
                m.x = v.x > m.x ? v.x : m.x; m.y = v.y > m.y ? v.y : m.y;
                m.z = v.z > m.z ? v.z : m.z; m.w = v.w > m.w ? v.w : m.w;
            }
        }
        st4(out + (((int64_t)n * Ho + ho) * Wo + wo) * C + c4 * 4, m);
    }
}

// fp16 -> fp16 max-pool with 8 channels (16 B) per lane (the stem pool reads 9 taps per output: half the load instructions)
typedef _Float16 h8p __attribute__((ext_vector_type(8)));
__global__ void maxpool_f16_c8_kernel(const half_t* __restrict__ in, int N, int H, int W, int C, int k, int s, int p, int Ho, int Wo,
                                      half_t* __restrict__ out) {
    const int c8n = C >> 3;
    const int64_t total = (int64_t)N * Ho * Wo * c8n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % c8n);
        int64_t t = i / c8n;
        const int wo = (int)(t % Wo); t /= Wo;
        const int ho = (int)(t % Ho);
        const int n = (int)(t / Ho);
        h8p m;
#pragma unroll
        for (int j = 0; j < 8; ++j) m[j] = (half_t)(-INFINITY);
        for (int r = 0; r < k; ++r) {
            const int hi = ho * s + r - p;
            if ((unsigned)hi >= (unsigned)H) continue;
            for (int q = 0; q < k; ++q) {
                const int wi = wo * s + q - p;
                if ((unsigned)wi >= (unsigned)W) continue;
                const h8p v = *(const h8p*)(in + (((int64_t)n * H + hi) * W + wi) * C + c8 * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) m[j] = v[j] > m[j] ? v[j] : m[j];  // exact: a max of fp16 values is one of them
            }
        }
        *(h8p*)(out + (((int64_t)n * Ho + ho) * Wo + wo) * C + c8 * 8) = m;
    }
}

__global__ void nearest2x_add_f16_kernel(const half_t* __restrict__ coarse, int N, int Hc, int Wc, int C, const half_t* __restrict__ lat,
                                         int H, int W, half_t* __restrict__ out) {
    const int c4n = C >> 2;
    const int64_t total = (int64_t)N * H * W * c4n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        int64_t t = i / c4n;
        const int x = (int)(t % W); t /= W;
        const int y = (int)(t % H);
        const int n = (int)(t / H);
        int yc = y >> 1, xc = x >> 1;
        yc = yc > Hc - 1 ? Hc - 1 : yc;
        xc = xc > Wc - 1 ? Wc - 1 : xc;
        const float4 a = ld4(lat + i * 4);
        const float4 b = ld4(coarse + (((int64_t)n * Hc + yc) * Wc + xc) * C + c4 * 4);
        st4(out + i * 4, make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w));
    }
}

struct RoiLevelsH {
    const half_t* feat[4];
    int H[4], W[4];
    float scale[4];
};
__device__ __forceinline__ int level_of_h(const float4 b, int k_min, int k_max) {
    const float area = (b.z - b.x + 1.0f) * (b.w - b.y + 1.0f);
    const float s = dm_sqrt(area);
    const float t = floorf(4.0f + dm_log2(dm_div(s, 224.0f) + 1e-6f));
    int l = (int)t;
    return l < k_min ? k_min : (l > k_max ? k_max : l);
}
__device__ __forceinline__ float4 roi_bilinear4_h(const half_t* f, int H, int W, int C, float y, float x) {
    float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return z;
    if (y <= 0.0f) y = 0.0f;
    if (x <= 0.0f) x = 0.0f;
    int yl = (int)y, xl = (int)x, yh, xh;
    if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
    if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
    const float ly = y - (float)yl, lx = x - (float)xl, hy = 1.0f - ly, hx = 1.0f - lx;
    const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
    const float4 v1 = ld4(f + ((int64_t)yl * W + xl) * C), v2 = ld4(f + ((int64_t)yl * W + xh) * C);
    const float4 v3 = ld4(f + ((int64_t)yh * W + xl) * C), v4 = ld4(f + ((int64_t)yh * W + xh) * C);
    float4 v;
    v.x = w1 * v1.x; v.x = v.x + w2 * v2.x; v.x = v.x + w3 * v3.x; v.x = v.x + w4 * v4.x;
    v.y = w1 * v1.y; v.y = v.y + w2 * v2.y; v.y = v.y + w3 * v3.y; v.y = v.y + w4 * v4.y;
    v.z = w1 * v1.z; v.z = v.z + w2 * v2.z; v.z = v.z + w3 * v3.z; v.z = v.z + w4 * v4.z;
    v.w = w1 * v1.w; v.w = v.w + w2 * v2.w; v.w = v.w + w3 * v3.w; v.w = v.w + w4 * v4.w;
    return v;
}
__global__ __launch_bounds__(256) void roi_align_f16_kernel(const RoiLevelsH lv, const float* __restrict__ rois, const int* __restrict__ counts,
                                                             int N, int K, int C, int PH, int PW, int g, int k_min, int k_max,
                                                             int aligned, half_t* __restrict__ out) {
    const int c4n = C >> 2;
    const int64_t total = (int64_t)N * K * PH * PW * c4n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        int64_t t = i / c4n;
        const int pw = (int)(t % PW); t /= PW;
        const int ph = (int)(t % PH); t /= PH;
        const int k = (int)(t % K);
        const int n = (int)(t / K);
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < counts[n]) {
            const float4 b = *(const float4*)(rois + ((int64_t)n * K + k) * 4);
            const int li = level_of_h(b, k_min, k_max) - k_min;
            const int H = lv.H[li], W = lv.W[li];
            const float sc = lv.scale[li];
            const half_t* f = lv.feat[li] + (int64_t)n * H * W * C + c4 * 4;
            const float off = aligned ? 0.5f : 0.0f;   // App. A.7 fork (see rcnn_ops.hip roi_align_kernel)
            const float sw = b.x * sc - off, sh = b.y * sc - off, ew = b.z * sc - off, eh = b.w * sc - off;
            float rw = ew - sw, rh = eh - sh;
            if (!aligned) {
                rw = rw > 1.0f ? rw : 1.0f;
                rh = rh > 1.0f ? rh : 1.0f;
            }
            const float bh = dm_div(rh, (float)PH), bw = dm_div(rw, (float)PW);
            for (int iy = 0; iy < g; ++iy) {
                const float y = sh + (float)ph * bh + dm_div(((float)iy + 0.5f) * bh, (float)g);
                for (int ix = 0; ix < g; ++ix) {
                    const float x = sw + (float)pw * bw + dm_div(((float)ix + 0.5f) * bw, (float)g);
                    const float4 v = roi_bilinear4_h(f, H, W, C, y, x);
                    o.x = o.x + v.x; o.y = o.y + v.y; o.z = o.z + v.z; o.w = o.w + v.w;
                }
            }
            const float cnt = (float)(g * g);
            o.x = dm_div(o.x, cnt); o.y = dm_div(o.y, cnt); o.z = dm_div(o.z, cnt); o.w = dm_div(o.w, cnt);
        }
        st4(out + i * 4, o);
    }
}

// One 256-thread block per RoI, 8 channels (16 B) per lane: C/8 lanes cover a bin, the block walks the PH*PW bins 256/(C/8) at
// a time, and the per-RoI arithmetic (level, scale, bin size) is done once per thread instead of once per output element.
// Per-channel arithmetic and its order are those of roi_align_f16_kernel (same bits out).
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void roi_align_f16_c8_kernel(const RoiLevelsH lv, const float* __restrict__ rois, const int* __restrict__ counts,
                                                                int K, int C, int PH, int PW, int g, int k_min, int k_max, int aligned,
                                                                half_t* __restrict__ out) {
    const int c8n = C >> 3, bpp = 256 / c8n;         // lanes per bin, bins per pass
    const int c8 = threadIdx.x % c8n, slot = threadIdx.x / c8n;
    const int n = blockIdx.x / K, k = blockIdx.x - n * K;
    const int nb = PH * PW;
    half_t* o = out + ((int64_t)blockIdx.x * nb) * C + c8 * 8;
    if (k >= counts[n]) {
        const h8 z = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
        for (int b = slot; b < nb; b += bpp) *(h8*)(o + (int64_t)b * C) = z;
        return;
    }
    const float4 bx = *(const float4*)(rois + ((int64_t)n * K + k) * 4);
    const int li = level_of_h(bx, k_min, k_max) - k_min;
    const int H = lv.H[li], W = lv.W[li];
    const float sc = lv.scale[li];
    const half_t* f = lv.feat[li] + (int64_t)n * H * W * C + c8 * 8;
    const float off = aligned ? 0.5f : 0.0f;   // App. A.7 fork
    const float sw = bx.x * sc - off, sh = bx.y * sc - off, ew = bx.z * sc - off, eh = bx.w * sc - off;
    float rw = ew - sw, rh = eh - sh;
    if (!aligned) {
        rw = rw > 1.0f ? rw : 1.0f;
        rh = rh > 1.0f ? rh : 1.0f;
    }
    const float bh = dm_div(rh, (float)PH), bw = dm_div(rw, (float)PW);
    const float cnt = (float)(g * g);
    if (g == 2) {
        // sampling_ratio 2 (every FPN head): the bin's four samples are set up first -- tap offsets and weights; a sample outside the map reads
        // clamped taps and its sum is REPLACED by +0 (not multiplied by zero weights: 0 * inf would be NaN), which is what the scalar form's
        // `continue` adds -- then all SIXTEEN 16-byte taps are
        // requested before the first is used (round 2 issued four at a time behind a branch per sample: 200 us per R101 bs=8 call at 2.7 TB/s)
        for (int b = slot; b < nb; b += bpp) {
            const int ph = b / PW, pw = b - ph * PW;
            int off[4][4];
            float wt[4][4];
            bool ins[4];
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) {
                const int iy = sidx >> 1, ix = sidx & 1;
                float y = sh + (float)ph * bh + dm_div(((float)iy + 0.5f) * bh, 2.0f);
                float x = sw + (float)pw * bw + dm_div(((float)ix + 0.5f) * bw, 2.0f);
                const bool inside = !(y < -1.0f || y > (float)H || x < -1.0f || x > (float)W);
                ins[sidx] = inside;
                if (y <= 0.0f) y = 0.0f;
                if (x <= 0.0f) x = 0.0f;
                int yl = (int)y, xl = (int)x, yh, xh;
                if (!inside) { yl = 0; xl = 0; y = 0.0f; x = 0.0f; }
                if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
                if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
                const float ly = y - (float)yl, lx = x - (float)xl, hy = 1.0f - ly, hx = 1.0f - lx;
                wt[sidx][0] = inside ? hy * hx : 0.0f; wt[sidx][1] = inside ? hy * lx : 0.0f;
                wt[sidx][2] = inside ? ly * hx : 0.0f; wt[sidx][3] = inside ? ly * lx : 0.0f;
                off[sidx][0] = (yl * W + xl) * C; off[sidx][1] = (yl * W + xh) * C; off[sidx][2] = (yh * W + xl) * C; off[sidx][3] = (yh * W + xh) * C;
            }
            h8 v[4][4];
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
                for (int t = 0; t < 4; ++t) v[sidx][t] = *(const h8*)(f + off[sidx][t]);
            float acc[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = 0.0f;
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float q = wt[sidx][0] * (float)v[sidx][0][i];
                    q = q + wt[sidx][1] * (float)v[sidx][1][i];
                    q = q + wt[sidx][2] * (float)v[sidx][2][i];
                    q = q + wt[sidx][3] * (float)v[sidx][3][i];
                    acc[i] = acc[i] + (ins[sidx] ? q : 0.0f);
                }
            h8 r;
#pragma unroll
            for (int i = 0; i < 8; ++i) r[i] = (half_t)dm_div(acc[i], cnt);
            *(h8*)(o + (int64_t)b * C) = r;
        }
        return;
    }
    for (int b = slot; b < nb; b += bpp) {
        const int ph = b / PW, pw = b - ph * PW;
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.0f;
        for (int iy = 0; iy < g; ++iy) {
            float y = sh + (float)ph * bh + dm_div(((float)iy + 0.5f) * bh, (float)g);
            for (int ix = 0; ix < g; ++ix) {
                float x = sw + (float)pw * bw + dm_div(((float)ix + 0.5f) * bw, (float)g);
                float yy = y;
                if (yy < -1.0f || yy > (float)H || x < -1.0f || x > (float)W) continue;  // sample contributes +0
                if (yy <= 0.0f) yy = 0.0f;
                if (x <= 0.0f) x = 0.0f;
                int yl = (int)yy, xl = (int)x, yh, xh;
                if (yl >= H - 1) { yh = yl = H - 1; yy = (float)yl; } else yh = yl + 1;
                if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
                const float ly = yy - (float)yl, lx = x - (float)xl, hy = 1.0f - ly, hx = 1.0f - lx;
                const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
                const h8 v1 = *(const h8*)(f + ((int64_t)yl * W + xl) * C), v2 = *(const h8*)(f + ((int64_t)yl * W + xh) * C);
                const h8 v3 = *(const h8*)(f + ((int64_t)yh * W + xl) * C), v4 = *(const h8*)(f + ((int64_t)yh * W + xh) * C);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float v = w1 * (float)v1[i];
                    v = v + w2 * (float)v2[i];
                    v = v + w3 * (float)v3[i];
                    v = v + w4 * (float)v4[i];
                    acc[i] = acc[i] + v;
                }
            }
        }
        h8 r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = (half_t)dm_div(acc[i], cnt);
        *(h8*)(o + (int64_t)b * C) = r;
    }
}

// The FPN heads' RoIAlign from the table of rcnn_ops.hip's roi_prep_kernel (see there): workgroup L (XCD L % 8) takes one 128-byte channel slice --
// eight lanes of 8 channels -- of RoI order[...]; C / 8 = 8 * ns lanes per pixel, ns in {1, 2, 4, 8}, 8 / ns RoIs per group of eight workgroups.
// Per-element arithmetic is roi_align_f16_c8_kernel's g == 2 branch (same bits out for any features, inf / NaN included: an invalid sample's taps point past
// the map and read 0; the division by the sample count 4 is an exact multiplication).
typedef unsigned int roi_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ h8 tap8(const __amdgpu_buffer_rsrc_t rs, unsigned off) {
    return __builtin_bit_cast(h8, (roi_u32x4)__builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
}
template <int PH, int PW>
__global__ __launch_bounds__(256) void roi_align_f16_tab_kernel(const RoiLevelsH lv, const int4* __restrict__ tab, const int* __restrict__ counts,
                                                                 const int* __restrict__ order, int NK, int K, int C, int ns, half_t* __restrict__ out) {
    constexpr int TS = 2 * (PH + PW) + 1, NB = PH * PW;
    __shared__ int4 t[TS];
    const int x = blockIdx.x & 7, rpg = 8 / ns;
    const int seq = (blockIdx.x >> 3) * rpg + x / ns, slice = x % ns;
    if (seq >= NK) return;
    const int roi = order ? order[seq] : seq;
    if ((unsigned)roi >= (unsigned)NK) return;
    const int n = roi / K, k = roi - n * K;
    half_t* o = out + ((int64_t)roi * NB) * C + slice * 64 + (threadIdx.x & 7) * 8;
    if (k >= counts[n]) {
        const h8 z = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
        for (int b = threadIdx.x >> 3; b < NB; b += 32) *(h8*)(o + (int64_t)b * C) = z;
        return;
    }
    const int4* tr = tab + (int64_t)roi * TS;
    if (threadIdx.x < TS) t[threadIdx.x] = tr[threadIdx.x];
    const int li = tr[TS - 1].x;   // (uniform address: a scalar load); the level's geometry comes from the launch arguments, not from the table
    if (tr[TS - 1].w != ((C * 2) | (PH << 16) | (PW << 24))) {   // roi_tab_sig: a table made for another layout -> NaN, not plausible numbers
        const half_t q = __builtin_bit_cast(half_t, (unsigned short)0x7e00);
        const h8 z = {q, q, q, q, q, q, q, q};
        for (int b = threadIdx.x >> 3; b < NB; b += 32) *(h8*)(o + (int64_t)b * C) = z;
        return;
    }
    const half_t* f0 = li == 0 ? lv.feat[0] : li == 1 ? lv.feat[1] : li == 2 ? lv.feat[2] : lv.feat[3];
    const int64_t img = (int64_t)(li == 0 ? lv.H[0] : li == 1 ? lv.H[1] : li == 2 ? lv.H[2] : lv.H[3]) * (li == 0 ? lv.W[0] : li == 1 ? lv.W[1] : li == 2 ? lv.W[2] : lv.W[3]) * C;
    // taps are 16-byte buffer loads range-checked against the image's map: whatever the table holds, no load leaves the level's allocation
    const __amdgpu_buffer_rsrc_t fb = __builtin_amdgcn_make_buffer_rsrc((void*)(f0 + (int64_t)n * img), 0, (unsigned)(img * 2), 0x00020000);
    const unsigned lo = slice * 128 + (threadIdx.x & 7) * 16;
    __syncthreads();
    for (int b = threadIdx.x >> 3; b < NB; b += 32) {
        const int ph = b / PW, pw = b - ph * PW;
        int4 e[4];
        e[0] = t[2 * ph]; e[1] = t[2 * ph + 1]; e[2] = t[2 * PH + 2 * pw]; e[3] = t[2 * PH + 2 * pw + 1];
        h8 v[4][4];
        float wt[4][4];
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
            const int4 ye = e[sidx >> 1], xe = e[2 + (sidx & 1)];
            const float hy = __int_as_float(ye.z), ly = __int_as_float(ye.w), hx = __int_as_float(xe.z), lx = __int_as_float(xe.w);
            wt[sidx][0] = hy * hx; wt[sidx][1] = hy * lx; wt[sidx][2] = ly * hx; wt[sidx][3] = ly * lx;
            v[sidx][0] = tap8(fb, (unsigned)(ye.x + xe.x) + lo); v[sidx][1] = tap8(fb, (unsigned)(ye.x + xe.y) + lo);
            v[sidx][2] = tap8(fb, (unsigned)(ye.y + xe.x) + lo); v[sidx][3] = tap8(fb, (unsigned)(ye.y + xe.y) + lo);
        }
        float acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = 0.0f;
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float q = wt[sidx][0] * (float)v[sidx][0][i];
                q = q + wt[sidx][1] * (float)v[sidx][1][i];
                q = q + wt[sidx][2] * (float)v[sidx][2][i];
                q = q + wt[sidx][3] * (float)v[sidx][3][i];
                acc[i] = acc[i] + q;
            }
        h8 r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r[i] = (half_t)(acc[i] * 0.25f);
        *(h8*)(o + (int64_t)b * C) = r;
    }
}

__global__ __launch_bounds__(256) void mask_logits_select_f16_kernel(const half_t* __restrict__ feat, int HW, int C, const float* __restrict__ w,
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
        const half_t* x = feat + ((int64_t)r * HW + p) * C;
        float acc = 0.0f;
        for (int c4 = 0; c4 < C / 4; ++c4) {
            const float4 v = ld4(x + 4 * c4);
            acc = fmaf(v.x, wr[4 * c4], acc); acc = fmaf(v.y, wr[4 * c4 + 1], acc);
            acc = fmaf(v.z, wr[4 * c4 + 2], acc); acc = fmaf(v.w, wr[4 * c4 + 3], acc);
        }
        out[(int64_t)r * HW + p] = dm_sigmoid(fmaf(acc, 1.0f, bias));
    }
}

// C = 256 form (round 3): the wave reads a pixel's 256 channels as ONE coalesced 512-B run (32 lanes x 16 B; two pixels per wave-instruction)
// instead of every lane walking its own pixel's 512 bytes 8 at a time (64 cache lines per instruction, each line touched 16 times: 118 us per
// R101 bs=8 step against a 64 us HBM floor).  Lane l holds the label's weights for channels 8 (l & 31) .. + 7; the dot product is finished by a
// butterfly over the 32 lanes -- a different fp32 association than the scalar chain (fp16 path: parity by tolerance).
__global__ __launch_bounds__(256) void mask_logits_select_f16_c256_kernel(const half_t* __restrict__ feat, int HW, const float* __restrict__ w,
                                                                           const float* __restrict__ b, const int* __restrict__ labels,
                                                                           float* __restrict__ out) {
    const int r = blockIdx.x;
    const int lab = labels[r];
    if (lab <= 0) {
        for (int p = threadIdx.x; p < HW; p += 256) out[(int64_t)r * HW + p] = 0.0f;
        return;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane >> 5, l32 = lane & 31;
    float wr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) wr[i] = w[(int64_t)lab * 256 + l32 * 8 + i];
    const float bias = b[lab];
    const half_t* base = feat + (int64_t)r * HW * 256 + l32 * 8;
    constexpr int U = 4;  // pixels in flight per lane
    for (int q0 = wave * 2; q0 < HW; q0 += 8 * U) {  // wave-uniform trip count: both halves of the wave reach every shuffle
        const int p0 = q0 + sub;
        h8 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int p = p0 + 8 * u;
            v[u] = *(const h8*)(base + (int64_t)(p < HW ? p : q0) * 256);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float acc = 0.0f;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc = fmaf((float)v[u][i], wr[i], acc);
#pragma unroll
            for (int d = 16; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
            const int p = p0 + 8 * u;
            if (l32 == 0 && p < HW) out[(int64_t)r * HW + p] = dm_sigmoid(fmaf(acc, 1.0f, bias));
        }
    }
}

static inline unsigned gridf(int64_t total) {
    int64_t b = cdiv64(total, 256);
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// fp32 NHWC C=3 image -> fp16 [N][H+6][(W+7)&~1][4] with a 3-pixel zero halo (and zero 4th channel): the fp16 stem's input
__global__ void pad_c3_to_f16_halo_kernel(const float* __restrict__ in, int N, int H, int W, int Hp, int Wp, half_t* __restrict__ out) {
    const int64_t total = (int64_t)N * Hp * Wp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wp) - 3;
        const int64_t t = i / Wp;
        const int y = (int)(t % Hp) - 3, n = (int)(t / Hp);
        h4 v = {(half_t)0.0f, (half_t)0.0f, (half_t)0.0f, (half_t)0.0f};
        if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
            const float* s = in + (((int64_t)n * H + y) * W + x) * 3;
            v[0] = (half_t)s[0]; v[1] = (half_t)s[1]; v[2] = (half_t)s[2];
        }
        *(h4*)(out + i * 4) = v;
    }
}
int pad_c3_to_f16_halo_launch(const float* in, int N, int H, int W, void* out, hipStream_t st) {
    const int Hp = H + 6, Wp = (W + 7) & ~1;
    hipLaunchKernelGGL(pad_c3_to_f16_halo_kernel, dim3(gridf((int64_t)N * Hp * Wp)), dim3(256), 0, st, in, N, H, W, Hp, Wp, (half_t*)out);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}

// F.interpolate(bilinear, align_corners=False) on fp16 storage (+ optional add, + optional ReLU): Yolact's FPN top-down and
// the protonet 2x upsample under fp16.  Arithmetic = resize_bilinear_kernel's (fp32, same order).
__device__ __forceinline__ float bil1_h(float lx0, float lx1, float ly0, float ly1, float v00, float v01, float v10, float v11) {
    float top = lx0 * v00; top = fmaf(lx1, v01, top);
    float bot = lx0 * v10; bot = fmaf(lx1, v11, bot);
    float v = ly0 * top; v = fmaf(ly1, bot, v);
    return v;
}
__global__ void resize_bilinear_f16_kernel(const half_t* __restrict__ in, int N, int H, int W, int C, int Ho, int Wo,
                                           const half_t* __restrict__ add, int relu, half_t* __restrict__ out) {
    const int c4n = C >> 2;
    const int64_t total = (int64_t)N * Ho * Wo * c4n;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % c4n);
        int64_t t = i / c4n;
        const int x = (int)(t % Wo); t /= Wo;
        const int y = (int)(t % Ho);
        const int n = (int)(t / Ho);
        int y0, y1, x0, x1; float ly0, ly1, lx0, lx1;
        dm_bil_coef(y, H, Ho, y0, y1, ly0, ly1);
        dm_bil_coef(x, W, Wo, x0, x1, lx0, lx1);
        const half_t* b = in + (int64_t)n * H * W * C + c4 * 4;
        const float4 v00 = ld4(b + ((int64_t)y0 * W + x0) * C), v01 = ld4(b + ((int64_t)y0 * W + x1) * C);
        const float4 v10 = ld4(b + ((int64_t)y1 * W + x0) * C), v11 = ld4(b + ((int64_t)y1 * W + x1) * C);
        float4 o;
        o.x = bil1_h(lx0, lx1, ly0, ly1, v00.x, v01.x, v10.x, v11.x);
        o.y = bil1_h(lx0, lx1, ly0, ly1, v00.y, v01.y, v10.y, v11.y);
        o.z = bil1_h(lx0, lx1, ly0, ly1, v00.z, v01.z, v10.z, v11.z);
        o.w = bil1_h(lx0, lx1, ly0, ly1, v00.w, v01.w, v10.w, v11.w);
        const int64_t oo = (((int64_t)n * Ho + y) * Wo + x) * C + c4 * 4;
        if (add) {
            const float4 a = ld4(add + oo);
            o.x = o.x + a.x; o.y = o.y + a.y; o.z = o.z + a.z; o.w = o.w + a.w;
        }
        if (relu) {
            o.x = o.x > 0.f ? o.x : 0.f; o.y = o.y > 0.f ? o.y : 0.f;
            o.z = o.z > 0.f ? o.z : 0.f; o.w = o.w > 0.f ? o.w : 0.f;
        }
        st4(out + oo, o);
    }
}

int maxpool_to_f16_launch(const void* in, int in_f16, int N, int H, int W, int C, int k, int s, int p, void* out, hipStream_t st) {
    ARG_CHECK(C % 4 == 0, "C % 4");
    const int Ho = (H + 2 * p - k) / s + 1, Wo = (W + 2 * p - k) / s + 1;
    const unsigned g = gridf((int64_t)N * Ho * Wo * (C / 4));
    if (in_f16 && C % 8 == 0)
        hipLaunchKernelGGL(maxpool_f16_c8_kernel, dim3(gridf((int64_t)N * Ho * Wo * (C / 8))), dim3(256), 0, st, (const half_t*)in, N, H, W, C, k, s, p, Ho,
                           Wo, (half_t*)out);
    else if (in_f16) hipLaunchKernelGGL(maxpool_to_f16_kernel<half_t>, dim3(g), dim3(256), 0, st, (const half_t*)in, N, H, W, C, k, s, p, Ho, Wo, (half_t*)out);
    else hipLaunchKernelGGL(maxpool_to_f16_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)in, N, H, W, C, k, s, p, Ho, Wo, (half_t*)out);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}
int resize_bilinear_f16_launch(const void* in, int N, int H, int W, int C, int Ho, int Wo, const void* add, int relu, void* out, hipStream_t st) {
    ARG_CHECK(C % 4 == 0, "C % 4");
    hipLaunchKernelGGL(resize_bilinear_f16_kernel, dim3(gridf((int64_t)N * Ho * Wo * (C / 4))), dim3(256), 0, st, (const half_t*)in, N, H, W, C, Ho,
                       Wo, (const half_t*)add, relu, (half_t*)out);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}
int nearest2x_add_f16_launch(const void* coarse, int N, int Hc, int Wc, int C, const void* lat, int H, int W, void* out, hipStream_t st) {
    ARG_CHECK(C % 4 == 0, "C % 4");
    hipLaunchKernelGGL(nearest2x_add_f16_kernel, dim3(gridf((int64_t)N * H * W * (C / 4))), dim3(256), 0, st, (const half_t*)coarse, N, Hc, Wc, C,
                       (const half_t*)lat, H, W, (half_t*)out);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}
int roi_align_f16_launch(const void* const* feats, const int* Hs, const int* Ws, const float* scales, int nlevels, const float* rois,
                         const int* counts, int N, int K, int C, int PH, int PW, int g, int k_min, void* out, hipStream_t st,
                         const int* order = nullptr, const void* tab = nullptr, int aligned = 0) {
    ARG_CHECK(nlevels >= 1 && nlevels <= 4 && C % 4 == 0, "roi_align levels/C");
    RoiLevelsH lv;
    for (int i = 0; i < 4; ++i) {
        const int s = i < nlevels ? i : nlevels - 1;
        lv.feat[i] = (const half_t*)feats[s]; lv.H[i] = Hs[s]; lv.W[i] = Ws[s]; lv.scale[i] = scales[s];
    }
    if (tab) {
        const int ns = C / 64;
        ARG_CHECK(g == 2 && C % 64 == 0 && (ns == 1 || ns == 2 || ns == 4 || ns == 8) && N > 0 && K > 0 && (int64_t)N * K < (1ll << 27) &&
                      ((PH == 7 && PW == 7) || (PH == 14 && PW == 14)),
                  "roi_align_f16 from a table: sampling 2, 7x7 or 14x14 bins, C in {64, 128, 256, 512}");
        for (int i = 0; i < nlevels; ++i) ARG_CHECK((int64_t)Hs[i] * Ws[i] * C * 2 < 0x20000000ll, "roi_align_f16 from a table: a level's map must stay under 512 MiB per image");
        const int rpg = 8 / ns, NK = N * K;
        const dim3 grid((unsigned)((NK + rpg - 1) / rpg * 8));
        if (PH == 7)
            hipLaunchKernelGGL((roi_align_f16_tab_kernel<7, 7>), grid, dim3(256), 0, st, lv, (const int4*)tab, counts, order, NK, K, C, ns, (half_t*)out);
        else
            hipLaunchKernelGGL((roi_align_f16_tab_kernel<14, 14>), grid, dim3(256), 0, st, lv, (const int4*)tab, counts, order, NK, K, C, ns, (half_t*)out);
        HIP_TRY(hipGetLastError());
        return ISEGMI_OK;
    }
    if (C % 8 == 0 && 256 % (C / 8) == 0 && (int64_t)N * K < (1ll << 31)) {
        if (N * K > 0)
            hipLaunchKernelGGL(roi_align_f16_c8_kernel, dim3((unsigned)(N * K)), dim3(256), 0, st, lv, rois, counts, K, C, PH, PW, g, k_min,
                               k_min + nlevels - 1, aligned, (half_t*)out);
    } else {
        hipLaunchKernelGGL(roi_align_f16_kernel, dim3(gridf((int64_t)N * K * PH * PW * (C / 4))), dim3(256), 0, st, lv, rois, counts, N, K, C, PH, PW,
                           g, k_min, k_min + nlevels - 1, aligned, (half_t*)out);
    }
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}
int mask_logits_select_f16_launch(const void* feat, int R, int HW, int C, const float* w, const float* b, const int* labels, float* out,
                                  hipStream_t st) {
    ARG_CHECK(C % 4 == 0, "C % 4");
    if (R == 0) return ISEGMI_OK;
    if (C == 256 && ((uintptr_t)feat & 15) == 0)
        hipLaunchKernelGGL(mask_logits_select_f16_c256_kernel, dim3(R), dim3(256), 0, st, (const half_t*)feat, HW, w, b, labels, out);
    else
        hipLaunchKernelGGL(mask_logits_select_f16_kernel, dim3(R), dim3(256), (size_t)C * sizeof(float), st, (const half_t*)feat, HW, C, w, b, labels, out);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}

}  // namespace isegmi
extern "C" int isegmi_op_pad_c3_to_f16_halo(const float* d_in_nhwc3, int N, int H, int W, void* d_out, void* stream) {
    ARG_CHECK(d_in_nhwc3 && d_out && N > 0 && H > 0 && W > 0, "args");
    return isegmi::pad_c3_to_f16_halo_launch(d_in_nhwc3, N, H, W, d_out, (hipStream_t)stream);
}
extern "C" int isegmi_op_roi_align_f16(const void* const* d_feats, const int32_t* Hs, const int32_t* Ws, const float* scales, int nlevels,
                                       const float* d_rois, const int32_t* d_counts, int N, int K, int C, int PH, int PW, int sampling, int aligned,
                                       int k_min, void* d_out, void* stream) {
    ARG_CHECK(d_feats && Hs && Ws && scales && d_rois && d_counts && d_out && N > 0 && K > 0 && PH > 0 && PW > 0 && sampling > 0, "args");
    return isegmi::roi_align_f16_launch(d_feats, Hs, Ws, scales, nlevels, d_rois, d_counts, N, K, C, PH, PW, sampling, k_min, d_out,
                                        (hipStream_t)stream, nullptr, nullptr, aligned);
}
extern "C" int isegmi_op_roi_align_f16_ordered(const void* const* d_feats, const int32_t* Hs, const int32_t* Ws, const float* scales, int nlevels,
                                               const float* d_rois, const int32_t* d_counts, const int32_t* d_order, const void* d_table, int N, int K,
                                               int C, int PH, int PW, int k_min, void* d_out, void* stream) {
    ARG_CHECK(d_feats && Hs && Ws && scales && d_rois && d_counts && d_table && d_out && N > 0 && K > 0 && PH > 0 && PW > 0, "args");
    return isegmi::roi_align_f16_launch(d_feats, Hs, Ws, scales, nlevels, d_rois, d_counts, N, K, C, PH, PW, 2, k_min, d_out, (hipStream_t)stream,
                                        d_order, d_table);
}
