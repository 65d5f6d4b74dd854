// conv_mfma.hip -- NHWC implicit-GEMM convolution on v_mfma_f32_32x32x2_f32 (gfx950).
//
// Stands in for the conv / FC / deconv framework ops the reference's models run on
// (Appendix A.1; reached from README.md:331 and
// README.md:243).  GEMM view: M = N*Ho*Wo output pixels, Ngemm = Cout, K = R*S*Cin.
//
// Numerics contract (shared with oracle/ora_ops.c::ora_conv2d): every output element is ONE
// accumulator chain acc = fmaf(x_k, w_k, acc) over k = (r, s, cin) ascending from +0.  The
// f32 MFMA is bit-for-bit such a chain (lanes 0-31 supply the first k of an instruction, lanes
// 32-63 the second), so there is no split-K and no multi-accumulator partial sum anywhere.
//
// Tiling: 256 threads = 4 waves; block tile BM x BN, K-chunk 32; each wave owns TM x TN tiles of
// 32x32.  A (pixels x k) and B (couts x k) chunks are staged global -> registers -> LDS with the
// issue-early / write-late split (loads for chunk t+1 are issued before the MFMAs of chunk t and
// written to the other LDS stage after them; one barrier per chunk).  LDS rows are 36 floats
// (32 + 4 pad): ds_write_b128 and ds_read_b128 are both conflict-free.  Inside each group of 8
// consecutive k the LDS image is stored as [k0 k2 k4 k6 | k1 k3 k5 k7] so that one ds_read_b128
// per lane half feeds 4 MFMAs in natural k order (weights are pre-permuted at pack time,
// activations by register renaming at LDS-write time).
#include <type_traits>
#include "../../include/isegmi.h"
#include "common.h"
#include "detmath.h"

#ifndef CONV_F32_RES_AUX  // cache policy of the residual loads (0 default, 2 = nt); see conv_mfma_f16.hip
#define CONV_F32_RES_AUX 0
#endif

namespace isegmi {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvK {
    const float* in;
    const float* w;
    const float* scale;
    const float* shift;
    const float* res;
    float* out;
    int N, H, W, Cin, Cout, R, S, stride, pad, Ho, Wo, M;
    int nchunks, cin_chunks;
    int kgroup;  // K order: channel groups of kgroup 32-chunks outermost, then (r, s), then the group's chunks (kgroup == cin_chunks: plain (r, s, cin))
    int64_t wrow;
    unsigned in_bytes, out_bytes, res_bytes, w_bytes;
    int64_t in_touched;  // host only (conv_set_band)
    int act, out_div, contiguous;
    int64_t out_img_stride, out_pix_stride;
    int mtiles, ntiles;
    int nband;  // tile order: Cout tiles in BANDS of nband; a band's tiles are walked pixel rows outermost, the band's Cout tiles innermost, band after
                // band; every XCD takes a contiguous eighth of that walk (nband == ntiles: Cout fastest; nband == 1: M fastest).  See conv_set_band().
    int m_begin;  // first output row of this launch's tile grid (0, or the start of the small-tile tail of a hybrid launch)
};

// logical tile index (contiguous per XCD) -> (pixel-row tile, Cout tile); all scalar
__device__ __forceinline__ void conv_tile_of(const ConvK& p, const int logical, int& mt, int& nt) {
    const int per = p.mtiles * p.nband;  // tiles of a full band
    const int band = logical / per, rem = logical - band * per;
    const int nb0 = band * p.nband;
    const int bw = min(p.nband, p.ntiles - nb0);  // the last band may be narrower
    mt = rem / bw;
    nt = nb0 + (rem - mt * bw);
}

constexpr int LDS_ROW = 36;
constexpr int CONV_KGROUP_CHUNKS = 4;  // 128 input channels per K group (ORA_CONV_CGROUP in the oracle)



// ---- shared epilogue of the 32x32-tile kernels.  D layout: col (cout) = lane&31, row (pixel) = (e&3) + 8*(e>>2) + 4*(lane>>5).
// Vector instructions between MFMAs cost matrix-pipe cycles (tools/microbench/mfma_switch.hip), so the common layout -- a
// dense [M, Cout] destination -- gets its 16 element offsets with ONE add each: row r of the wave's tile is a wave-uniform
// multiple of the row pitch away from the lane's base, rows past M fall behind a descriptor cut at M rows, and a lane whose
// column is past Cout starts from the out-of-range base.  Other destinations (concatenated head buffers, deconv parities) keep
// the general per-row arithmetic.  `row0` = the lane's first pixel row (tile row base + 4*(lane>>5)), `co` = its channel.
struct Epi {
    __amdgpu_buffer_rsrc_t rs_out, rs_res;
    unsigned off[16];   // byte offset of element e in the destination (out of range = dropped)
    float rv[16];       // residual (+0 without one)
};

__device__ __forceinline__ void epi_begin(const ConvK& p, int row0, int co, Epi& ep) {
    constexpr unsigned OOB = 0x80000000u;
    const bool cok = co < p.Cout;
    const bool dense = p.contiguous && p.out_pix_stride == p.Cout;  // wave-uniform
    const unsigned dense_bytes = (unsigned)p.M * (unsigned)p.Cout * 4u;
    ep.rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, dense ? dense_bytes : p.out_bytes, 0x00020000);
    ep.rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res ? p.res : p.out), 0, p.res ? p.res_bytes : 0u, 0x00020000);
    if (dense) {
        const unsigned pitch = (unsigned)p.Cout * 4u;
        const unsigned base = cok ? (unsigned)row0 * pitch + (unsigned)co * 4u : OOB;  // OOB + 27 rows stays out of range
#pragma unroll
        for (int e = 0; e < 16; ++e) ep.off[e] = base + (unsigned)((e & 3) + 8 * (e >> 2)) * pitch;
        if (p.res) {
#pragma unroll
            for (int e = 0; e < 16; ++e) ep.rv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ep.rs_res, ep.off[e], 0, CONV_F32_RES_AUX));
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) ep.rv[e] = 0.0f;
        }
    } else {
        const unsigned cooff = cok ? (unsigned)co * 4u : OOB;  // OOB + anything stays out of range (< 2^32)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = row0 + (e & 3) + 8 * (e >> 2);
            const unsigned resoff = m < p.M ? (unsigned)m * (unsigned)p.Cout * 4u : OOB;
            ep.rv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ep.rs_res, (resoff | cooff) >= OOB ? OOB : resoff + cooff, 0, CONV_F32_RES_AUX));
            unsigned rowoff;
            if (p.contiguous) rowoff = m < p.M ? (unsigned)m * (unsigned)p.out_pix_stride * 4u : OOB;
            else {
                const int ni = m / p.out_div, pi = m - ni * p.out_div;
                rowoff = m < p.M ? (unsigned)(((int64_t)ni * p.out_img_stride + (int64_t)pi * p.out_pix_stride) * 4) : OOB;
            }
            ep.off[e] = (rowoff | cooff) >= OOB ? OOB : rowoff + cooff;
        }
    }
}

// y = fmaf(acc, scale, shift) (+res) -> act -> NHWC store; the activation is a wave-uniform branch around the element loop.
template <typename Acc>
__device__ __forceinline__ void epi_finish(const ConvK& p, const Acc& acc, float sc, float sh, const Epi& ep) {
    float yv[16];
    if (p.act == 1) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float y = fmaf(acc[e], sc, sh) + ep.rv[e];  // rv is +0 without a residual: y + 0 == y for every y we can produce
            yv[e] = y > 0.0f ? y : 0.0f;
        }
    } else if (p.act == 0) {
#pragma unroll
        for (int e = 0; e < 16; ++e) yv[e] = fmaf(acc[e], sc, sh) + ep.rv[e];
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float y = fmaf(acc[e], sc, sh);
            if (p.act == 4) {  // DarkNet block: LeakyReLU(0.1) FIRST, then the shortcut
                y = y > 0.0f ? y : y * 0.1f;
                yv[e] = y + ep.rv[e];
                continue;
            }
            y = y + ep.rv[e];
            yv[e] = p.act == 3 ? (y > 0.0f ? y : y * 0.1f) : y;
        }
        if (p.act == 2) {  // tanh only on the Yolact coefficient head
#pragma unroll
            for (int e = 0; e < 16; ++e) yv[e] = dm_tanh(yv[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, yv[e]), ep.rs_out, ep.off[e], 0, 0);
}

template <int BM, int BN, int WM, int WN, bool STEM>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ConvK p) {
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int APASS = BM / 64, BPASS = BN / 64;
    constexpr int STAGE = (BM + BN) * LDS_ROW;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform, and known to be (scalar address arithmetic)
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware bijective remap: blocks that share an XCD get consecutive logical tiles
    // (same pixel rows / neighbouring rows -> activation re-reads hit that XCD's L2).
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    // an XCD's consecutive tiles walk the operand that is SMALLER in bytes, so its L2 holds all of that one and only an
    // eighth of the larger (weights on small-M layers: res5 / P5-P7 heads / everything at bs=1 were re-streaming the whole
    // filter bank from HBM into every XCD)
    int mt, nt;
    conv_tile_of(p, logical, mt, nt);
    const int m0 = mt * BM, n0 = nt * BN;

    // ---- loader state: thread covers row (tid>>2)+64*j, 8-group g = tid&3 of the 32-chunk
    const int lrow = tid >> 2, g = tid & 3;
    int hi0[APASS], wi0[APASS], nb[APASS];
#pragma unroll
    for (int j = 0; j < APASS; ++j) {
        const int m = m0 + lrow + 64 * j;
        if (m < p.M) {
            const int hw = p.Ho * p.Wo;
            const int n = m / hw, rem = m - n * hw;
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            hi0[j] = ho * p.stride - p.pad;
            wi0[j] = wo * p.stride - p.pad;
            nb[j] = n * p.H;
        } else {
            hi0[j] = -(1 << 28);
            wi0[j] = 0;
            nb[j] = 0;
        }
    }
    const float* wsrc = p.w + (int64_t)(n0 + lrow) * p.wrow + g * 8;

    // A is fetched with raw buffer loads: padding taps / rows past M use an out-of-range offset, for
    // which the hardware returns 0 -- no branch, no select, and the loads stay in flight under the MFMAs.
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    u32x4 ra[APASS][2];
    u32x4 rb[BPASS][2];
    int kr = 0, ks = 0, kc = 0, kg = 0, glen = p.kgroup;  // (r, s, chunk within the channel group, group base) of the NEXT chunk to load

    auto load_chunk = [&](int chunk) {
#pragma unroll
        for (int j = 0; j < APASS; ++j) {
            if (STEM) {
                const int hi = hi0[j] + chunk, wi = wi0[j] + 2 * g;
                const bool okh = (unsigned)hi < (unsigned)p.H;
                const unsigned off = (unsigned)(((nb[j] + hi) * p.W + wi) * 16);
                const unsigned o0 = (okh && (unsigned)wi < (unsigned)p.W) ? off : OOB;
                const unsigned o1 = (okh && g < 3 && (unsigned)(wi + 1) < (unsigned)p.W) ? off + 16u : OOB;
                ra[j][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, o0, 0, 0);
                ra[j][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, o1, 0, 0);
            } else {
                const int hi = hi0[j] + kr, wi = wi0[j] + ks;
                const bool ok = (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
                const unsigned off = ((unsigned)((nb[j] + hi) * p.W + wi) * (unsigned)p.Cin + (unsigned)((kg + kc) * 32 + g * 8)) * 4u;
                const unsigned o0 = ok ? off : OOB;
                ra[j][0] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, o0, 0, 0);
                ra[j][1] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, ok ? off + 16u : OOB, 0, 0);
            }
        }
#pragma unroll
        for (int j = 0; j < BPASS; ++j) {
            const float* src = wsrc + (int64_t)(64 * j) * p.wrow + (STEM ? chunk : (kr * p.S + ks) * p.cin_chunks + kg + kc) * 32;
            rb[j][0] = *(const u32x4*)src;
            rb[j][1] = *(const u32x4*)(src + 4);
        }
        if (!STEM) {
            if (++kc == glen) {
                kc = 0;
                if (++ks == p.S) {
                    ks = 0;
                    if (++kr == p.R) { kr = 0; kg += glen; glen = min(p.kgroup, p.cin_chunks - kg); if (glen <= 0) { glen = 1; kg = 0; } }
                }
            }
        }
    };
    auto store_chunk = [&](int stage) {
        float* As = smem + stage * STAGE;
        float* Bs = As + BM * LDS_ROW;
#pragma unroll
        for (int j = 0; j < APASS; ++j) {
            float* d = As + (lrow + 64 * j) * LDS_ROW + g * 8;
            *(u32x4*)d = u32x4{ra[j][0].x, ra[j][0].z, ra[j][1].x, ra[j][1].z};
            *(u32x4*)(d + 4) = u32x4{ra[j][0].y, ra[j][0].w, ra[j][1].y, ra[j][1].w};
        }
#pragma unroll
        for (int j = 0; j < BPASS; ++j) {
            float* d = Bs + (lrow + 64 * j) * LDS_ROW + g * 8;
            *(u32x4*)d = rb[j][0];
            *(u32x4*)(d + 4) = rb[j][1];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.0f;

    const int lr = lane & 31, lh = lane >> 5;
    const int a_off = (wm * TM * 32 + lr) * LDS_ROW + lh * 4;
    const int b_off = BM * LDS_ROW + (wn * TN * 32 + lr) * LDS_ROW + lh * 4;

    load_chunk(0);
    store_chunk(0);
    __syncthreads();

    auto compute = [&](int stage) {
        const float* sb = smem + stage * STAGE;
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
            float4 fa[TM], fb[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) fa[a] = *(const float4*)(sb + a_off + a * 32 * LDS_ROW + gg * 8);
#pragma unroll
            for (int b = 0; b < TN; ++b) fb[b] = *(const float4*)(sb + b_off + b * 32 * LDS_ROW + gg * 8);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].x, fb[b].x, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].y, fb[b].y, acc[a][b], 0, 0, 0);
                    // stem: a chunk is one filter row, 7 taps x 4 channels; k = 28..31 are zero pads in both operands and
                    // fmaf(0, 0, acc) == acc for every acc a +0 start can reach, so their two MFMAs are left out
                    if (STEM && gg == 3) continue;
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].z, fb[b].z, acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a].w, fb[b].w, acc[a][b], 0, 0, 0);
                }
        }
    };

    // Steady state (last chunk peeled so the loop body is branch-free): issue chunk t+1's loads, run
    // chunk t's 16*TM*TN MFMAs under them, then write t+1 to the other LDS stage; one barrier per chunk.
    int cur = 0;
    for (int t = 0; t + 1 < p.nchunks; ++t) {
        load_chunk(t + 1);
        __builtin_amdgcn_sched_barrier(0);
        compute(cur);
        __builtin_amdgcn_sched_barrier(0);  // keep the LDS write (and its vmcnt wait) BELOW the MFMAs
        store_chunk(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    compute(cur);

    // ---- epilogue (epi_begin / epi_finish above): residual loads and output stores are raw buffer ops whose offset is out of
    // range for rows >= M / couts >= Cout (hardware returns 0 / drops the store), so all 16 loads of a tile are in flight together.
#pragma unroll
    for (int a = 0; a < TM; ++a) {
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int co = n0 + (wn * TN + b) * 32 + lr;
            const bool cok = co < p.Cout;
            const float sc = (cok && p.scale) ? p.scale[co] : 1.0f;
            const float sh = (cok && p.shift) ? p.shift[co] : 0.0f;
            Epi ep;
            epi_begin(p, m0 + (wm * TM + a) * 32 + 4 * lh, co, ep);
            epi_finish(p, acc[a][b], sc, sh, ep);
            __builtin_amdgcn_sched_barrier(0);  // one 32x32 tile's loads in flight at a time (register budget)
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// v2 of the 64x64 / 32x32x2 kernel (tiles 7, 8, 9 = RING 2, 3, 4): same tile, same LDS image, same numerics, other schedule.
//  * Loads run RING chunks ahead in registers, issued by inline asm and awaited by an exact `s_waitcnt vmcnt` (see the ring
//    notes on conv_mfma16_kernel): with one chunk of prefetch a block whose A operand streams from HBM (Cout <= 128: the
//    1x1 reductions of res2 / res3, 8 KB per chunk) had at most 32 KB in flight per CU and sat at ~3 TB/s.
//  * A chunk's registers reach LDS at the START of the iteration that precedes its use (they landed an iteration ago), next to
//    the first fragment reads; the next loads' address arithmetic follows, and only then the 16 dependent MFMAs with the
//    fragment reads of step g+1 issued under the MFMAs of step g.  A wave that is alone on its SIMD (grids of 1-3 blocks per
//    CU) no longer serialises [loads | MFMAs | LDS store | barrier]: everything but the barrier sits under its own MFMAs.
//  * The residual is requested before the LAST chunk's MFMAs instead of after them (one memory round trip per tile, all of a
//    K = 64 layer's tile time besides the loads themselves).
template <int RING, bool EARLY>
__device__ __forceinline__ void conv_v2_body(const ConvK& p, float* smem, const int bid, const int nwg) {
    constexpr int BM = 64, BN = 64;
    constexpr int STAGE = (BM + BN) * LDS_ROW;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform, and known to be (scalar address arithmetic)
    const int wm = wave >> 1, wn = wave & 1;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    int mt, nt;
    conv_tile_of(p, logical, mt, nt);
    const int m0 = p.m_begin + mt * BM, n0 = nt * BN;

    // loader: thread covers row tid>>2 of the A and of the B tile, 8 consecutive k (g = tid&3) of the 32-chunk
    const int lrow = tid >> 2, g = tid & 3;
    int hi0, wi0;
    unsigned abase;  // byte offset of (n, hi0, wi0, 8 g); wraps for padding taps, which the range test rejects
    {
        const int m = m0 + lrow;
        if (m >= p.M) { hi0 = -(1 << 28); wi0 = 0; abase = 0; }
        else if (p.R * p.S == 1 && p.stride == 1 && p.pad == 0) {  // uniform: output pixel m IS input pixel m (no two integer divisions)
            hi0 = 0; wi0 = 0;
            abase = ((unsigned)m * (unsigned)p.Cin + (unsigned)(g * 8)) * 4u;
        } else {
            const int hw = p.Ho * p.Wo;
            const int n = m / hw, rem = m - n * hw;
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            hi0 = ho * p.stride - p.pad; wi0 = wo * p.stride - p.pad;
            abase = ((unsigned)((n * p.H + hi0) * p.W + wi0) * (unsigned)p.Cin + (unsigned)(g * 8)) * 4u;
        }
    }
    const unsigned wbase = ((unsigned)(n0 + lrow) * (unsigned)p.wrow + (unsigned)(g * 8)) * 4u;
    constexpr unsigned OOB = 0x80000000u;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    auto make_rsrc = [](const void* ptr, unsigned bytes) {
        const unsigned long long a = (unsigned long long)ptr;
        return u32x4{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a),
                     (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu)),
                     (unsigned)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
    };
    const u32x4 rs_in = make_rsrc(p.in, p.in_bytes), rs_w = make_rsrc(p.w, p.w_bytes);
    // Every wave64 VALU instruction in this loop costs 2-3 cycles of the SIMD's matrix pipe with four waves per SIMD and 4-6 with
    // one (tools/microbench/mfma_switch.hip: independent v_mov / v_add between dependent MFMAs; SALU and un-awaited LDS reads cost
    // nothing), so the per-chunk vector work is cut to two instructions:
    //  * the tap's byte offset and the past-the-end switch are scalar (SALU): the A load's voffset is `abase + tap` (one v_add),
    //    made out-of-range by ONE v_cndmask under a lane mask that is recomputed only when the tap changes (uniform branch);
    //    past-the-end chunks swap in a zero-length descriptor with scalar selects; the B load's voffset is a loop constant and
    //    its chunk offset rides in the instruction's scalar offset;
    //  * the LDS image's k permutation is done by ds_write2_b32 (two data registers, two offsets) instead of moving the loaded
    //    registers into b128 order (8 v_mov per chunk), and all LDS addresses are per-stage constants with immediate offsets.
    const u32x4 rs_null = u32x4{rs_in.x, rs_in.y, 0u, rs_in.w};
    u32x4 ra[RING][2], rb[RING][2];
    int kr = 0, ks = 0, kc = 0, kg = 0, glen = p.kgroup, chunk = 0;  // position of the next chunk to load (strictly in order)
    const bool taps = p.R * p.S > 1 || p.pad > 0;  // 1x1 / pad 0: every tap of a row < M is inside the image
    // round 2, second pass: the two remaining instructions went too -- the lane's voffset (pixel base + tap offset, or the
    // out-of-range constant for a padding tap / a row past M) is rebuilt only when the tap changes, behind a scalar branch; the
    // cin chunk inside the tap rides in the scalar offset operand (outside the range check; it never leaves the pixel's Cin floats).
    const bool ok0 = taps ? ((unsigned)hi0 < (unsigned)p.H && (unsigned)wi0 < (unsigned)p.W) : hi0 >= 0;
    unsigned avoff = ok0 ? abase : OOB;
    unsigned soffa = 0;
    auto load_chunk = [&](int slot) {
        const bool live = chunk < p.nchunks;
        const u32x4 rsa = live ? rs_in : rs_null, rsb = live ? rs_w : rs_null;  // scalar selects
        const unsigned soffb = (unsigned)((kr * p.S + ks) * p.cin_chunks + kg + kc) * 128u;   // scalar: the chunk's place in the (r, s, cin) packed weight row
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(ra[slot][0]) : "v"(avoff), "s"(rsa), "s"(soffa) : "memory");
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:16" : "=v"(ra[slot][1]) : "v"(avoff), "s"(rsa), "s"(soffa) : "memory");
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(rb[slot][0]) : "v"(wbase), "s"(rsb), "s"(soffb) : "memory");
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:16" : "=v"(rb[slot][1]) : "v"(wbase), "s"(rsb), "s"(soffb) : "memory");
        ++chunk;
        soffa += 128u;
        if (++kc == glen) {  // uniform: next tap (of this channel group, or the first of the next group) -- the only place with per-lane work
            kc = 0;
            if (++ks == p.S) {
                ks = 0;
                if (++kr == p.R) { kr = 0; kg += glen; glen = min(p.kgroup, p.cin_chunks - kg); if (glen <= 0) { glen = 1; kg = 0; } }  // past the end: loads are dead
            }
            soffa = (unsigned)kg * 128u;
            int tr = __builtin_amdgcn_readfirstlane(kr), ts = __builtin_amdgcn_readfirstlane(ks);
            asm volatile("" : "+s"(tr), "+s"(ts));  // keeps the tap change behind its branch: speculated, its VALU work would run every chunk
            // branch-free on purpose (bitwise &, no short circuit): with && hipcc built this from three exec-mask regions (s_and_saveexec + branches),
            // ~70 cycles per tap change; this form is 2 v_add + 2 v_cmp + v_add + v_cndmask
            const unsigned inb = (unsigned)((unsigned)(hi0 + tr) < (unsigned)p.H) & (unsigned)((unsigned)(wi0 + ts) < (unsigned)p.W);  // 1x1 / pad 0: hi0 = wi0 = 0
            const unsigned tapv = abase + (unsigned)((tr * p.W + ts) * p.Cin) * 4u;
            avoff = inb ? tapv : OOB;
        }
    };
    // LDS addresses (floats): per-thread constants; the stage is a compile-time term wherever the loop is unrolled over it
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const unsigned lds_wr_addr = (unsigned)(uintptr_t)(lds_ptr)(smem + lrow * LDS_ROW + g * 8);  // LDS byte address of this thread's 8 floats, stage 0
    auto store_chunk = [&](int slot, int stage, auto pending) {  // pending = loads issued after this slot's
        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(ra[slot][0]), "+v"(ra[slot][1]), "+v"(rb[slot][0]), "+v"(rb[slot][1]) : "n"(decltype(pending)::value) : "memory");
        // [k0 k2 k4 k6 | k1 k3 k5 k7]: four ds_write2_b32, each taking its two dwords from two different loaded registers (written
        // as plain stores hipcc re-vectorises them into ds_write_b128 behind eight v_mov); B keeps its two ds_write_b128.  The
        // block's s_barrier below is preceded by an explicit lgkmcnt(0): the compiler does not count asm LDS operations.
        const unsigned wa = lds_wr_addr + (unsigned)(stage * STAGE * 4);
        asm volatile("ds_write2_b32 %0, %1, %2 offset1:1" :: "v"(wa), "v"(ra[slot][0].x), "v"(ra[slot][0].z) : "memory");
        asm volatile("ds_write2_b32 %0, %1, %2 offset0:2 offset1:3" :: "v"(wa), "v"(ra[slot][1].x), "v"(ra[slot][1].z) : "memory");
        asm volatile("ds_write2_b32 %0, %1, %2 offset0:4 offset1:5" :: "v"(wa), "v"(ra[slot][0].y), "v"(ra[slot][0].w) : "memory");
        asm volatile("ds_write2_b32 %0, %1, %2 offset0:6 offset1:7" :: "v"(wa), "v"(ra[slot][1].y), "v"(ra[slot][1].w) : "memory");
        asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(wa), "v"(rb[slot][0]), "n"(BM * LDS_ROW * 4) : "memory");
        asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(wa), "v"(rb[slot][1]), "n"(BM * LDS_ROW * 4 + 16) : "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.0f;
    const int lr = lane & 31, lh = lane >> 5;
    const float* const lds_a = smem + (wm * 32 + lr) * LDS_ROW + lh * 4;
    const float* const lds_b = smem + BM * LDS_ROW + (wn * 32 + lr) * LDS_ROW + lh * 4;
    float4 fa[2], fb[2];
    auto read_frags = [&](int set, int stage, int gg) {
        fa[set] = *(const float4*)(lds_a + stage * STAGE + gg * 8);
        fb[set] = *(const float4*)(lds_b + stage * STAGE + gg * 8);
    };
    auto mma4 = [&](int set) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set].x, fb[set].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set].y, fb[set].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set].z, fb[set].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set].w, fb[set].w, acc, 0, 0, 0);
    };
    auto compute = [&](int stage) {  // fragment reads of step g+1 under the MFMAs of step g; set 0 of step 0 is already requested
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
            if (gg < 3) read_frags((gg + 1) & 1, stage, gg + 1);
            mma4(gg & 1);
        }
    };
    typedef std::integral_constant<int, 4 * (RING - 1)> Steady;   // store right after the refill of the previous slot
    typedef std::integral_constant<int, 4 * (RING - 2) < 0 ? 0 : 4 * (RING - 2)> Early;  // store BEFORE this iteration's refill
    static_assert(RING % 2 == 0, "the LDS stage of an unrolled position is its parity");

#pragma unroll
    for (int i = 0; i < RING; ++i) load_chunk(i);
    store_chunk(0, 0, Steady());
    __syncthreads();
    const int last = p.nchunks - 1;
    for (int t0 = 0; t0 < last; t0 += RING) {
#pragma unroll
        for (int j = 0; j < RING; ++j) {
            if (t0 + j >= last) break;  // uniform
            const int cur = j & 1;      // compile-time: t0 is a multiple of the (even) ring depth
            read_frags(0, cur, 0);
            if (EARLY) store_chunk((j + 1) % RING, cur ^ 1, Early());   // chunk t+1: requested RING-1 iterations ago
            load_chunk(j);                                               // chunk t+RING into the slot chunk t left
            __builtin_amdgcn_sched_barrier(0);
            compute(cur);
            __builtin_amdgcn_sched_barrier(0);
            // EARLY = false (layers that stream A from HBM): the wait for chunk t+1 sits BELOW chunk t's MFMAs
            if (!EARLY) store_chunk((j + 1) % RING, cur ^ 1, Steady());
            __syncthreads();
        }
    }
    const int cur = last & 1;
    // last chunk (already in LDS): the residual is requested first, so that it travels under the chunk's MFMAs
    const int co = n0 + wn * 32 + lr;
    const bool cok = co < p.Cout;
    Epi ep;
    epi_begin(p, m0 + wm * 32 + 4 * lh, co, ep);
    const float sc = (cok && p.scale) ? p.scale[co] : 1.0f;
    const float sh = (cok && p.shift) ? p.shift[co] : 0.0f;
    __builtin_amdgcn_sched_barrier(0);
    read_frags(0, cur, 0);
    compute(cur);
    __builtin_amdgcn_sched_barrier(0);
    // the past-the-end loads of the ring must have landed before their registers are reused
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < RING; ++i) asm volatile("" :: "v"(ra[i][0]), "v"(ra[i][1]), "v"(rb[i][0]), "v"(rb[i][1]));
    epi_finish(p, acc, sc, sh, ep);
}

template <int RING, bool EARLY>
__global__ __launch_bounds__(256, 4) void conv_mfma_v2_kernel(const ConvK p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    conv_v2_body<RING, EARLY>(p, smem, (int)blockIdx.x, (int)gridDim.x);
}

// ---------------------------------------------------------------------------------------------------------
// Small-M / latency-bound variant: block tile 32x32, four waves, each wave ONE 16x16 tile on
// v_mfma_f32_16x16x4_f32 (lane (i, q) supplies k = 4s + q; the instruction is the same ordered fmaf chain, 4 k deep).
// A wave's accumulator chain advances 4 k per 40 cycles instead of 2 k per 64, so the K-order latency floor that
// bounds tiny layers (res5, P5-P7 heads, everything at bs=1) drops ~2.5x, and a layer yields 4x more blocks than
// with 64x64 tiles.  Same packed weights and buffer-op loads/epilogue as the main kernel; the LDS image differs: rows
// hold the 32 k of a chunk in NATURAL order at a 34-float pitch, so the 32 lanes of a ds_read_b32 group (16 rows x 2
// k-quads) land on 32 different banks (bank = 2*row + k mod 32).  The main kernel's 36-float permuted image made these
// reads 4-way conflicted (512 LDS cycles per chunk and block, the largest single term of a chunk at 1-2 blocks per CU).
// 17.4 KB LDS -> 8 blocks/CU.
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Loads run RING chunks ahead in registers (one 16-B A load and one 16-B B load per thread and chunk): these layers run at
// 1-2 blocks per CU with only 8 short MFMAs per chunk, so a single chunk of prefetch left every chunk waiting ~0.7 us for
// its own loads (53 us per K=2304 layer at bs=1); chunks past the end load zeros through the buffer range check.
//
// LW = true adds four LOADER waves (512 threads): at one block per CU a wave's chunk is bound by its own instruction stream
// (timing-only builds: ~290 cycles of address arithmetic + loop control, ~110 of LDS traffic, ~60 of barrier on top of the
// 256-cycle dependent MFMA chain, nothing of it overlapped because there is no second wave on the SIMD), so the global
// loads, their address arithmetic and the LDS stores move to a partner wave on the same SIMD and the MFMA wave is left
// with 8 ds_read2 + 8 MFMA + 1 barrier per chunk.
// WN = 16x16 tiles per wave along Cout (block 32 x 32*WN): with WN = 2 a wave runs two independent accumulator chains off one
// A fragment -- 0.75 instead of 1 LDS fragment read per MFMA, 3 instead of 4 global loads per 16 MFMAs, and the MFMA pipe sees
// two chains per wave -- for the mid-size grids where the 32x32 block is steady-state LDS/issue-bound.
// TIMING-ONLY build switches (results wrong, time only; both 0 in the product): HALF_B makes the 32x64 block read one B fragment instead of
// two (a third fewer LDS reads), NO_LOADS turns every global load of the 16x16x4 kernels into a zero-length-descriptor load.
// profiles/r02_experiments.txt has what they showed.
#ifndef EXPERIMENT_HALF_B
#define EXPERIMENT_HALF_B 0
#endif
#ifndef EXPERIMENT_NO_LOADS
#define EXPERIMENT_NO_LOADS 0
#endif
#ifndef EXPERIMENT_MODE  // bit 0: no per-chunk barrier, bit 1: no LDS stores (plain 16x16x4 loop)
#define EXPERIMENT_MODE 0
#endif
// KS > 1 (round 6, tile 15: FIXED-TREE SPLIT-K, an opt-in numerics mode -- see conv2d_launch): the block is KS sets of 256 threads; set q walks the K
// chunks [q L, (q + 1) L), L = ceil(nchunks / KS), of the kernel's ordinary K order as its own k-ordered chain from +0 -- every set runs the plain
// variant's loop on its own LDS stages -- and set 0 adds the partial sums in the fixed order ((p0 + p1) + p2) + p3 before the epilogue.  A 16 x 16 output
// tile is then FOUR dependent MFMA chains of a quarter of the length on the four waves a SIMD holds, instead of one chain on one wave with the other
// three slots empty: what a small-M layer at bs = 1 lacks is independent work, not CUs.  The oracle restates exactly this sum (ora_conv2d_split).
template <int RING, bool LW, int WN = 1, int KS = 1>
__device__ __forceinline__ void conv_mfma16_body(const ConvK& p, float* smem16, const int bid, const int nwg) {
    static_assert(WN == 1 || !LW, "the loader-wave variant runs one tile per wave");
    static_assert(KS == 1 || (!LW && WN == 1), "split-K runs KS plain 32 x 32 sets");
    constexpr int BM = 32, BN = 32 * WN;
    constexpr int ROW = 34;  // floats; pitch = 2 mod 32: bank = 2*row + k for the fragment reads
    constexpr int STAGE = (BM + BN) * ROW;
    constexpr int NST = LW ? 3 : 2;  // LDS stages (LW: chunk t computed while t+1 is read into registers and t+2 written)
    float* const smem_block = smem16;
    const int kset = KS > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) : 0;   // wave-uniform
    if (KS > 1) smem16 += kset * NST * STAGE;
    // this set's chunks [c_begin, c_end); every set runs `count` iterations (the block's barriers are shared): chunks past c_end are dead (zeros)
    const int count = KS > 1 ? (p.nchunks + KS - 1) / KS : p.nchunks;
    const int c_begin = KS > 1 ? min(kset * count, p.nchunks) : 0;
    const int c_end = KS > 1 ? min(c_begin + count, p.nchunks) : p.nchunks;

    const int tid = threadIdx.x & 255;
    const bool loader = LW && __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8)) != 0;  // wave-uniform, and known to be
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform, and known to be (scalar address arithmetic)
    const int wm = wave >> 1, wn = wave & 1;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7;
    const int logical = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    // an XCD's consecutive tiles walk the operand that is SMALLER in bytes, so its L2 holds all of that one and only an
    // eighth of the larger
    int mt, nt;
    conv_tile_of(p, logical, mt, nt);
    const int m0 = p.m_begin + mt * BM, n0 = nt * BN;

    // loader: thread covers row tid>>3 (32 rows) and 4 consecutive k (g8 = tid&7) of the 32-chunk
    const int lrow = tid >> 3, g8 = tid & 7;
    int hi0, wi0;
    unsigned abase;  // byte offset of (n, hi0, wi0, 4*g8); wraps for padding taps, which the range test below rejects
    {
        const int m = m0 + lrow;
        if (m >= p.M) { hi0 = -(1 << 28); wi0 = 0; abase = 0; }
        else if (p.R * p.S == 1 && p.stride == 1 && p.pad == 0) {  // uniform: output pixel m IS input pixel m (no two integer divisions)
            hi0 = 0; wi0 = 0;
            abase = ((unsigned)m * (unsigned)p.Cin + (unsigned)(g8 * 4)) * 4u;
        } else {
            const int hw = p.Ho * p.Wo;
            const int n = m / hw, rem = m - n * hw;
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            hi0 = ho * p.stride - p.pad; wi0 = wo * p.stride - p.pad;
            abase = ((unsigned)((n * p.H + hi0) * p.W + wi0) * (unsigned)p.Cin + (unsigned)(g8 * 4)) * 4u;
        }
    }
    const unsigned wbase = ((unsigned)(n0 + lrow) * (unsigned)p.wrow + (unsigned)(g8 * 4)) * 4u;
    constexpr unsigned OOB = 0x80000000u;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    // The ring's loads are issued and awaited through inline asm: with the builtin loads the compiler's own s_waitcnt
    // placement loses count across the unrolled loop's back edge (vmcnt(1), (0) at the head of every RING-th iteration: the
    // whole ring drained there and a full memory latency, ~850 cycles, showed up at the block's barrier every RING chunks).
    // Loads return in order, so "the chunk RING-1 behind the newest has landed" is exactly vmcnt(2 * (RING - 1)); chunks
    // past the end still issue (out of range, zeros) and keep that count exact.
    auto make_rsrc = [](const void* ptr, unsigned bytes) {
        const unsigned long long a = (unsigned long long)ptr;
        return u32x4{(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a),
                     (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu)),
                     (unsigned)__builtin_amdgcn_readfirstlane((int)bytes), 0x00020000u};
    };
    const u32x4 rs_in = make_rsrc(p.in, p.in_bytes), rs_w = make_rsrc(p.w, p.w_bytes);
    const u32x4 rs_null = u32x4{rs_in.x, rs_in.y, 0u, rs_in.w};
    u32x4 ra[RING], rb[RING][WN];
    int kr = 0, ks = 0, kc = 0, kg = 0, glen = p.kgroup, chunk = c_begin;  // position of the next chunk to load (they are loaded strictly in order)
    if (KS > 1 && c_begin > 0) {   // the K position of chunk c_begin: channel group, tap, chunk inside the group (scalar; once per block)
        int rem = c_begin;
        glen = min(p.kgroup, p.cin_chunks);
        while (rem >= p.R * p.S * glen && kg + glen < p.cin_chunks) { rem -= p.R * p.S * glen; kg += glen; glen = min(p.kgroup, p.cin_chunks - kg); }
        const int tap = rem / glen;
        kc = rem - tap * glen;
        kr = tap / p.S;
        ks = tap - kr * p.S;
    }
    // Per-chunk vector work is ZERO instructions (see conv_mfma_v2_kernel: every VALU instruction costs matrix-pipe cycles): the
    // lane's voffset = pixel base + tap offset, or the out-of-range constant for a padding tap / a row past M, is rebuilt only
    // when the tap changes (a scalar branch); the cin chunk inside the tap and B's chunk offset ride in the scalar offset
    // operand (not part of the range check, and it never leaves the pixel's Cin floats); past-the-end chunks swap in a
    // zero-length descriptor (scalar selects).
    const bool taps = p.R * p.S > 1 || p.pad > 0;
    const bool ok0 = taps ? ((unsigned)(hi0 + kr) < (unsigned)p.H && (unsigned)(wi0 + ks) < (unsigned)p.W) : hi0 >= 0;   // (kr = ks = 0 unless a split set starts mid-K)
    unsigned avoff = ok0 ? abase + (unsigned)((kr * p.W + ks) * p.Cin) * 4u : OOB;
    unsigned soffa = (unsigned)(kg + kc) * 128u;
    auto load_chunk = [&](int slot) {
        const bool live = chunk < c_end && !EXPERIMENT_NO_LOADS;
        const u32x4 rsa = live ? rs_in : rs_null, rsb = live ? rs_w : rs_null;
        const unsigned soffb = (unsigned)((kr * p.S + ks) * p.cin_chunks + kg + kc) * 128u;
        asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(ra[slot]) : "v"(avoff), "s"(rsa), "s"(soffa) : "memory");
#pragma unroll
        for (int t = 0; t < WN; ++t) {  // weight rows lrow and lrow + 32
            const unsigned vb = wbase + (unsigned)t * 32u * (unsigned)p.wrow * 4u;  // loop constant
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(rb[slot][t]) : "v"(vb), "s"(rsb), "s"(soffb) : "memory");
        }
        ++chunk;
        soffa += 128u;
        if (++kc == glen) {  // uniform: next tap (of this channel group, or the first of the next group)
            kc = 0;
            if (++ks == p.S) {
                ks = 0;
                if (++kr == p.R) { kr = 0; kg += glen; glen = min(p.kgroup, p.cin_chunks - kg); if (glen <= 0) { glen = 1; kg = 0; } }
            }
            soffa = (unsigned)kg * 128u;
            int tr = __builtin_amdgcn_readfirstlane(kr), ts = __builtin_amdgcn_readfirstlane(ks);
            asm volatile("" : "+s"(tr), "+s"(ts));  // keeps the tap change behind its branch: speculated, its VALU work would run every chunk
            // branch-free on purpose (bitwise &, no short circuit): with && hipcc built this from three exec-mask regions (s_and_saveexec + branches),
            // ~70 cycles per tap change; this form is 2 v_add + 2 v_cmp + v_add + v_cndmask
            const unsigned inb = (unsigned)((unsigned)(hi0 + tr) < (unsigned)p.H) & (unsigned)((unsigned)(wi0 + ts) < (unsigned)p.W);  // 1x1 / pad 0: hi0 = wi0 = 0
            const unsigned tapv = abase + (unsigned)((tr * p.W + ts) * p.Cin) * 4u;
            avoff = inb ? tapv : OOB;
        }
    };
    // A arrives in natural k order (two 8-byte stores: rows are only 8-byte aligned at this pitch); the packed weight row
    // holds [k0 k2 k4 k6 | k1 k3 k5 k7] per 8-group, so thread (grp, half) owns k = 8*grp + 2j + half
    const int grp = g8 >> 1, half = g8 & 1;
    auto store_chunk = [&](int slot, int stage, auto pending) {  // pending = loads issued after this slot's
        asm volatile("s_waitcnt vmcnt(%1)" : "+v"(ra[slot]) : "n"(decltype(pending)::value) : "memory");
#pragma unroll
        for (int t = 0; t < WN; ++t) asm volatile("" : "+v"(rb[slot][t]) :: "memory");
        float* As = smem16 + stage * STAGE;
        float* Bs = As + BM * ROW;
        float* d = As + lrow * ROW + g8 * 4;
        *(u32x2*)d = u32x2{ra[slot].x, ra[slot].y};
        *(u32x2*)(d + 2) = u32x2{ra[slot].z, ra[slot].w};
#pragma unroll
        for (int t = 0; t < WN; ++t) {
            unsigned* b = (unsigned*)(Bs + (lrow + 32 * t) * ROW + grp * 8 + half);
            b[0] = rb[slot][t].x; b[2] = rb[slot][t].y; b[4] = rb[slot][t].z; b[6] = rb[slot][t].w;
        }
    };

    f32x4 acc[WN];
#pragma unroll
    for (int t = 0; t < WN; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const int li = lane & 15, lq = lane >> 4;
    const int a_off = (wm * 16 + li) * ROW + lq;  // MFMA step s reads k = 4s + lq
    const int b_off = BM * ROW + (wn * 16 * WN + li) * ROW + lq;
    float fa[LW ? 2 : 1][8], fb[LW ? 2 : 1][WN][8];
    auto read_frags = [&](int set, int stage) {
        const float* sb = smem16 + stage * STAGE;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            fa[set][s] = sb[a_off + s * 4];
#pragma unroll
            for (int t = 0; t < WN; ++t) fb[set][t][s] = sb[b_off + (EXPERIMENT_HALF_B ? 0 : t) * 16 * ROW + s * 4];
        }
    };
    auto mma = [&](int set) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int t = 0; t < WN; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[set][s], fb[set][t][s], acc[t], 0, 0, 0);
    };
    typedef std::integral_constant<int, (1 + WN) * (RING - 1)> Steady;
    static_assert(RING % 2 == 0 && RING >= 4, "fragment sets alternate with the unrolled slot index");

    if (!LW) {
        // Plain variant (17.4 KB LDS, 8 blocks/CU -- the one for grids of several blocks per CU, where the other blocks hide
        // a chunk's latencies): chunk c waits in register slot c % RING; iteration t refills the slot chunk t left, issues all
        // sixteen fragment reads of chunk t BEFORE its eight dependent MFMAs (one LDS latency per chunk instead of four) and
        // moves chunk t+1 to the other LDS stage.
#pragma unroll
        for (int i = 0; i < RING; ++i) load_chunk(i);
        store_chunk(0, 0, Steady());
        __syncthreads();
        for (int t0 = 0; t0 < count; t0 += RING) {
#pragma unroll
            for (int j = 0; j < RING; ++j) {
                if (t0 + j >= count) break;  // uniform
                const int cur = j & 1;  // compile-time (t0 is a multiple of the even ring depth): LDS addresses are immediates
                load_chunk(j);
                __builtin_amdgcn_sched_barrier(0);
                read_frags(0, cur);
                mma(0);
                __builtin_amdgcn_sched_barrier(0);
                if (!(EXPERIMENT_MODE & 2)) store_chunk((j + 1) % RING, cur ^ 1, Steady());
                if (!(EXPERIMENT_MODE & 1)) __syncthreads();
            }
        }
        // the past-the-end loads still in flight target ring registers the epilogue is about to reuse
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        // Loader-wave variant, software-pipelined over three LDS stages (the K order of every accumulator is untouched):
        // in iteration t the MFMA waves issue the fragment reads of chunk t+1 into the other register set and run the eight
        // dependent MFMAs of chunk t under them, while the loader waves move chunk t+2 from its register slot to the third
        // LDS stage and refill that slot with chunk t+2+RING.  One barrier per chunk, same count in both roles; the stage
        // written at t was last read by the fragment reads issued at t-2, consumed before the barrier of t-1.
        if (loader) {
#pragma unroll
            for (int i = 0; i < RING; ++i) load_chunk(i);
            store_chunk(0, 0, Steady());
            store_chunk(1, 1, std::integral_constant<int, (1 + WN) * (RING - 2)>());
            load_chunk(0);
            load_chunk(1);
        }
        __syncthreads();
        if (loader) {
            int st2 = 2;
            for (int t0 = 0; t0 < p.nchunks; t0 += RING) {
#pragma unroll
                for (int j = 0; j < RING; ++j) {
                    if (t0 + j >= p.nchunks) break;  // uniform
                    store_chunk((j + 2) % RING, st2, Steady());
                    load_chunk((j + 2) % RING);
                    if (!(EXPERIMENT_MODE & 4)) __syncthreads();   // (bit 2, timing only: the loader-wave variant without its per-chunk barrier)
                    st2 = st2 == NST - 1 ? 0 : st2 + 1;
                }
            }
            return;  // the epilogue is not theirs
        }
        read_frags(0, 0);
        int st1 = 1;
        for (int t0 = 0; t0 < p.nchunks; t0 += 2) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (t0 + j >= p.nchunks) break;  // uniform
                read_frags((j + 1) & 1, st1);
                mma(j & 1);
                // issue order (round 5): the chunk's fragment reads (eight ds_read2_b32) two per MFMA in the shadow of the FIRST four dependent MFMAs, so that
                // they have landed when the wave reaches the barrier's lgkmcnt(0) behind the seventh.  One read per MFMA (round 2) issued the last two
                // right in front of that wait: an LDS latency exposed per chunk (0.28 us per 32-k chunk against 0.13 of MFMA chain at one block per CU).
                // CONV_LW_READS_PER_MFMA = 1 restores it for A/B.
#ifndef CONV_LW_READS_PER_MFMA
#define CONV_LW_READS_PER_MFMA 2
#endif
#pragma unroll
                for (int s = 0; s < 8; ++s) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
                    if (s < 8 / CONV_LW_READS_PER_MFMA) __builtin_amdgcn_sched_group_barrier(0x100, CONV_LW_READS_PER_MFMA, 0);  // DS read
                }
                __builtin_amdgcn_sched_barrier(0);   // (register-only MFMAs are not ordered by the barrier's fence: without this hipcc sinks four of them below it)
                if (!(EXPERIMENT_MODE & 4)) __syncthreads();
                st1 = st1 == NST - 1 ? 0 : st1 + 1;
            }
        }
    }

    if (KS > 1) {   // ((p0 + p1) + p2) + ...: sets 1.. park their partial tile in LDS (every stage is free: the loop ended on a barrier), set 0 adds them in order
        float* red = smem_block + (kset > 0 ? (kset - 1) * 1024 : 0) + tid * 4;
        if (kset > 0) *(f32x4*)red = acc[0];
        __syncthreads();
        if (kset > 0) return;
#pragma unroll
        for (int q = 1; q < KS; ++q) {
            const f32x4 v = *(const f32x4*)(smem_block + (q - 1) * 1024 + tid * 4);
            acc[0][0] = acc[0][0] + v[0]; acc[0][1] = acc[0][1] + v[1]; acc[0][2] = acc[0][2] + v[2]; acc[0][3] = acc[0][3] + v[3];
        }
    }

    // epilogue: D col (cout) = lane&15, row (pixel) = (lane>>4)*4 + e.  Same two layouts as epi_begin: a dense [M, Cout]
    // destination takes one add per element (descriptor cut at M rows, out-of-range base for a column past Cout).
    const bool dense = p.contiguous && p.out_pix_stride == p.Cout;  // wave-uniform
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, dense ? (unsigned)p.M * (unsigned)p.Cout * 4u : p.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res ? p.res : p.out), 0, p.res ? p.res_bytes : 0u, 0x00020000);
    const int row0 = m0 + wm * 16 + lq * 4;
    const unsigned pitch = (unsigned)p.Cout * 4u;
    const unsigned rowbase = (unsigned)row0 * pitch;
#pragma unroll
    for (int t = 0; t < WN; ++t) {
        const int co = n0 + (wn * WN + t) * 16 + li;
        const bool cok = co < p.Cout;
        const float sc = (cok && p.scale) ? p.scale[co] : 1.0f;
        const float sh = (cok && p.shift) ? p.shift[co] : 0.0f;
        unsigned ooff[4];
        float rv[4];
        if (dense) {
            const unsigned base = cok ? rowbase + (unsigned)co * 4u : OOB;
#pragma unroll
            for (int e = 0; e < 4; ++e) ooff[e] = base + (unsigned)e * pitch;
            if (p.res) {
#pragma unroll
                for (int e = 0; e < 4; ++e) rv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_res, ooff[e], 0, CONV_F32_RES_AUX));
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) rv[e] = 0.0f;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = row0 + e;
                const bool ok = cok && m < p.M;
                unsigned off;
                if (p.contiguous) off = ((unsigned)m * (unsigned)p.out_pix_stride + (unsigned)co) * 4u;
                else {
                    const int ni = m / p.out_div, pi = m - ni * p.out_div;
                    off = (unsigned)(((int64_t)ni * p.out_img_stride + (int64_t)pi * p.out_pix_stride + co) * 4);
                }
                ooff[e] = ok ? off : OOB;
                rv[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_res, ok ? ((unsigned)m * (unsigned)p.Cout + (unsigned)co) * 4u : OOB, 0, CONV_F32_RES_AUX));
            }
        }
        float yv[4];
        if (p.act == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float y = fmaf(acc[t][e], sc, sh) + rv[e];
                yv[e] = y > 0.0f ? y : 0.0f;
            }
        } else if (p.act == 0) {
#pragma unroll
            for (int e = 0; e < 4; ++e) yv[e] = fmaf(acc[t][e], sc, sh) + rv[e];
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float y = fmaf(acc[t][e], sc, sh);
                if (p.act == 4) {  // DarkNet block: LeakyReLU(0.1) first, then the shortcut
                    y = y > 0.0f ? y : y * 0.1f;
                    yv[e] = y + rv[e];
                    continue;
                }
                y = y + rv[e];
                yv[e] = p.act == 3 ? (y > 0.0f ? y : y * 0.1f) : y;
            }
            if (p.act == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) yv[e] = dm_tanh(yv[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, yv[e]), rs_out, ooff[e], 0, 0);
    }
}

template <int RING, bool LW, int WN = 1>
__global__ __launch_bounds__(LW ? 512 : 256) void conv_mfma16_kernel(const ConvK p) {
    constexpr int STAGE16 = (32 + 32 * WN) * 34;
    __shared__ __attribute__((aligned(16))) float smem16[(LW ? 3 : 2) * STAGE16];
    conv_mfma16_body<RING, LW, WN>(p, smem16, (int)blockIdx.x, (int)gridDim.x);
}

// tile 15: the 32 x 32 block with its K range cut over KS sets of four waves (fixed-tree split-K, see conv_mfma16_body)
template <int KS>
__global__ __launch_bounds__(256 * KS) void conv_mfma16_split_kernel(const ConvK p) {
    extern __shared__ __attribute__((aligned(16))) float smem_split[];   // KS x 2 stages x (32 + 32) rows x 34 floats
    conv_mfma16_body<4, false, 1, KS>(p, smem_split, (int)blockIdx.x, (int)gridDim.x);
}

// ---------------------------------------------------------------------------------------------------------
// HYBRID launch (tiles 13 / 14 = v2 with loads 2 / 4 chunks ahead): a grid of 64x64 tiles that does not divide over the 256 CUs
// (528 tiles: two per CU and sixteen left over; 1050: four per CU and 26 left over) loses up to a third of a layer to the CUs
// that carry one tile more than the others -- the 16x16x4 kernels' finer tiles fix the balance but pay for it on every tile.
// Here the rows that fill the chip evenly run on the 64x64 v2 tile and ONLY the remaining rows on 32x32 blocks, in one launch:
// blocks [0, nmain) are v2 tiles of rows [0, m_begin of the tail), the others 32x32 tiles of the tail rows (four times as many,
// a quarter of the work each, dispatched last so that the launch ends on short blocks).  Same numerics in both parts (one
// k-ordered chain per output).  tools/hybrid_tail_experiment.py measured the idea with two concurrent launches: 9-15 % ahead
// of the best single kernel on 528 / 616 / 1056 / 1192-tile layers.
template <int RING>
__global__ __launch_bounds__(256, 4) void conv_hybrid_kernel(const ConvK pm, const ConvK pt, const int nmain) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if ((int)blockIdx.x < nmain) conv_v2_body<RING, false>(pm, smem, (int)blockIdx.x, nmain);
    else conv_mfma16_body<4, false, 1>(pt, smem, (int)blockIdx.x - nmain, (int)gridDim.x - nmain);
}

// GROUPED launch (round 5): up to CONV_GROUP_MAX independent convolutions in ONE launch -- the shared-weight heads of all FPN levels (Yolact upfeature /
// head_cat over P3..P7, Mask R-CNN's RPN head over P2..P6), the FPN's lateral and output convs.  Part i owns blocks [blk0[i], blk0[i + 1]) and runs them
// exactly as its own launch would (kind 0: 64 x 64 v2 tiles, kind 1: 32 x 32 blocks; the tile -> XCD walk is by the block's index inside its part), so
// every output keeps its single k-ordered chain: bit-identical to separate launches.  What it buys: the small levels (P5..P7: 16-164 tiles of 64 x 64 at
// bs 8, 10-41 TF/s and ~23 us each as launches of their own) run inside the big level's launch on CUs that would idle in its last round, and a forward
// is 12-14 launches shorter.  Big parts first: the launch ends on short blocks.
constexpr int CONV_GROUP_MAX = 10;
struct ConvGroupK {
    int n;
    int blk0[CONV_GROUP_MAX + 1];
    int kind[CONV_GROUP_MAX];
    ConvK k[CONV_GROUP_MAX];
};
static_assert(sizeof(ConvGroupK) <= 4096, "kernel arguments");
template <int RING>
__global__ __launch_bounds__(256, 4) void conv_group_kernel(const ConvGroupK g) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int i = 0;
    while (i + 1 < g.n && (int)blockIdx.x >= g.blk0[i + 1]) ++i;   // uniform
    const int bid = (int)blockIdx.x - g.blk0[i], nwg = g.blk0[i + 1] - g.blk0[i];
    if (g.kind[i] == 0) conv_v2_body<RING, false>(g.k[i], smem, bid, nwg);
    else conv_mfma16_body<4, false, 1>(g.k[i], smem, bid, nwg);
}

static inline int perm8(int e) { return 4 * (e & 1) + (e >> 1); }
static bool is_stem(const isegmi_conv_desc* d) { return d->Cin == 4 && d->R == 7 && d->S == 7; }
static int cout_pad(const isegmi_conv_desc* d) { return cdiv(d->Cout, 128) * 128; }
static int n_chunks(const isegmi_conv_desc* d) { return is_stem(d) ? 7 : d->R * d->S * (d->Cin / 32); }

static int check_desc(const isegmi_conv_desc* d) {
    ARG_CHECK(d != nullptr, "null desc");
    ARG_CHECK(d->N > 0 && d->H > 0 && d->W > 0 && d->Cout > 0 && d->R > 0 && d->S > 0 && d->stride > 0 && d->pad >= 0,
              "non-positive conv geometry");
    ARG_CHECK(is_stem(d) || (d->Cin > 0 && d->Cin % 32 == 0), "Cin must be a multiple of 32 (or the Cin=4 7x7 stem)");
    ARG_CHECK(d->H + 2 * d->pad >= d->R && d->W + 2 * d->pad >= d->S, "kernel larger than padded input");
    ARG_CHECK(d->act >= 0 && d->act <= 4, "act");
    ARG_CHECK(d->tile >= 0 && d->tile <= 15, "tile");
    return ISEGMI_OK;
}

// Tile order of a launch (k.mtiles / k.ntiles set).  Each of the 8 XCDs has its own 4 MB L2 and takes a contiguous eighth of the tile walk, so the
// walk decides how often the two operands cross the fabric.  With nb bands of Cout tiles: nb <= 8 -- a band is shared by 8 / nb XCDs, each on its
// own range of pixel rows: the input is fetched nb times in total, the filter bank 8 / nb times; nb > 8 -- every XCD walks all pixel rows for each of
// its bands: the input 8 times (once per XCD if the XCD's tiles are all resident at once, else once per band), the bank once.  A band whose filters
// (bn * nband * K * 4 bytes) do not stay in L2 next to the streaming input is fetched again by every group of pixel-row tiles the XCD runs one after
// the other.  Rounds 1-2 knew only nb = 1 ("Cout fastest") and nb = ntiles ("M fastest", whenever the bank was larger than the input): res5's 1x1
// layers, the stride-2 3x3, the 351-wide Yolact head and Mask R-CNN's fc6 fetched 4-9x their operands (profiles/r03_conv_traffic_*.txt).
static void conv_set_band(ConvK& k, const int bn, const int blocks_per_cu, const int64_t in_touched) {
    static int legacy = -1;
    if (legacy < 0) { const char* e = getenv("ISEGMI_CONV_BAND"); legacy = (e && atoi(e) == 0) ? 1 : 0; }
    if (legacy) { k.nband = (int64_t)k.w_bytes > (int64_t)k.in_bytes ? 1 : k.ntiles; return; }
    const double A = (double)in_touched, B = (double)k.nchunks * 128.0 * bn * k.ntiles;
    const double l2_keep = 2.5e6;                       // of the 4 MB: what a resident operand may take next to the streaming one
    const double slots = 32.0 * blocks_per_cu;          // resident blocks of one XCD
    const double tiles_xcd = (double)k.mtiles * k.ntiles / 8.0;
    const double rounds = tiles_xcd / slots > 1.0 ? tiles_xcd / slots : 1.0;
    double best = 0;
    int best_bw = k.ntiles;
    for (int nb = 1; nb <= k.ntiles; nb *= 2) {
        const int bw = cdiv(k.ntiles, nb), nbands = cdiv(k.ntiles, bw);
        const double band_b = B / nbands;
        double a, b;
        if (nbands <= 8) { a = A * nbands; b = B * (8.0 / nbands); }
        else { a = A * 8.0 * (rounds > 1.0 ? nbands / 8.0 : 1.0); b = B; }
        if (band_b > l2_keep && rounds > 1.0) b *= rounds;   // not resident: streamed again by every round of pixel-row tiles
        const double cost = a + b;
        if (nb == 1 || cost < best * 0.97) { best = cost; best_bw = bw; }  // ties and near-ties stay with the wider band
        if (bw == 1) break;
    }
    k.nband = best_bw;
}

template <int BM, int BN, int WM, int WN>
static int launch(const isegmi_conv_desc* d, ConvK& k, hipStream_t st) {
    k.mtiles = cdiv(k.M, BM);
    k.ntiles = cdiv(d->Cout, BN);
    conv_set_band(k, BN, 2, k.in_touched);
    const size_t lds = 2 * (size_t)(BM + BN) * LDS_ROW * sizeof(float);
    const dim3 grid((unsigned)(k.mtiles * k.ntiles)), block(256);
    if (is_stem(d)) {
        LDS_LIMIT_ONCE((int)lds, conv_mfma_kernel<BM, BN, WM, WN, true>);
        hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, WM, WN, true>), grid, block, lds, st, k);
    } else {
        LDS_LIMIT_ONCE((int)lds, conv_mfma_kernel<BM, BN, WM, WN, false>);
        hipLaunchKernelGGL((conv_mfma_kernel<BM, BN, WM, WN, false>), grid, block, lds, st, k);
    }
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}

// compute units of the current device (256 on an MI355X in SPX mode; fewer in a partitioned mode): queried once per device
int device_cu_count() {
    static int cached[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        cached[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return cached[dev];
}

// desc + operands -> the kernel argument block (everything but the tile grid: mtiles / ntiles / nband / m_begin belong to the launch form)
static int conv_fill(const isegmi_conv_desc* d, const float* in, const float* w, const float* scale, const float* shift, const float* res, float* out,
                     ConvK& k) {
    int rc = check_desc(d);
    if (rc) return rc;
    k.in = in; k.w = w; k.scale = scale; k.shift = shift; k.res = res; k.out = out;
    k.N = d->N; k.H = d->H; k.W = d->W; k.Cin = d->Cin; k.Cout = d->Cout; k.R = d->R; k.S = d->S;
    k.stride = d->stride; k.pad = d->pad;
    k.Ho = (d->H + 2 * d->pad - d->R) / d->stride + 1;
    k.Wo = (d->W + 2 * d->pad - d->S) / d->stride + 1;
    const int64_t M64 = (int64_t)d->N * k.Ho * k.Wo;
    ARG_CHECK(M64 < (1ll << 31) - 256, "too many output pixels");
    k.M = (int)M64;
    k.nchunks = n_chunks(d);
    k.cin_chunks = is_stem(d) ? 1 : d->Cin / 32;
    // K order (the numerics contract with oracle/ora_ops.c, ora_conv2d): input channels in groups of 128 (= 4 chunks) outermost, then (r, s), then
    // the group's chunks.  With all of Cin inside every tap (rounds 1-2) the concurrently running tiles of an XCD streamed their whole activation
    // window once PER TAP through a 4 MB L2 that also holds the filter bank: 3x3 layers fetched their input 3.7-4.2x (proto_net, FPN / RPN heads,
    // the mask head) and res5's 3x3s 26x; with 128-channel groups the window of one group stays in L2 for its nine taps: 10.2 -> 8.5 GB fetched
    // + written per Yolact step (1.51x -> 1.26x algorithmic), 9.8 -> 8.6 GB Mask R-CNN (profiles/r03_conv_traffic_*.txt).  Groups of 32 or 64
    // channels fetch no less (1.22-1.27x) and pay for twice / four times the tap changes (+0.7 % / +1.8 % conv time; 128: +0.1-0.9 %).
    k.kgroup = (d->R * d->S == 1 || is_stem(d)) ? k.cin_chunks : (k.cin_chunks < CONV_KGROUP_CHUNKS ? k.cin_chunks : CONV_KGROUP_CHUNKS);
    k.wrow = (int64_t)k.nchunks * 32;
    const int64_t in_bytes = (int64_t)d->N * d->H * d->W * d->Cin * 4;
    ARG_CHECK(in_bytes < (1ll << 31), "conv input must be < 2 GiB (32-bit buffer offsets)");
    k.in_bytes = (unsigned)in_bytes;
    k.w_bytes = (unsigned)((int64_t)cout_pad(d) * k.wrow * 4);
    // input bytes the launch really reads: a strided 1x1 skips (stride^2 - 1) / stride^2 of the pixels
    k.in_touched = (d->R == 1 && d->S == 1 && d->stride > 1) ? in_bytes / ((int64_t)d->stride * d->stride) : in_bytes;
    k.m_begin = 0;
    k.act = d->act;
    k.out_div = d->out_div > 0 ? d->out_div : k.Ho * k.Wo;
    k.out_pix_stride = d->out_pix_stride > 0 ? d->out_pix_stride : d->Cout;
    k.out_img_stride = d->out_img_stride > 0 ? d->out_img_stride : (int64_t)k.out_div * k.out_pix_stride;
    k.contiguous = (k.out_img_stride == (int64_t)k.out_div * k.out_pix_stride) ? 1 : 0;
    // extent of the destination reachable by this launch (for the store descriptor's range check)
    const int64_t n_img = (k.M + k.out_div - 1) / k.out_div;
    const int64_t out_extent = ((n_img - 1) * k.out_img_stride + (int64_t)(k.out_div - 1) * k.out_pix_stride + d->Cout) * 4;
    ARG_CHECK(out_extent < (1ll << 31), "conv output span must be < 2 GiB (32-bit buffer offsets)");
    k.out_bytes = (unsigned)out_extent;
    k.res_bytes = (unsigned)((int64_t)k.M * d->Cout * 4);
    ARG_CHECK(res == nullptr || (int64_t)k.M * d->Cout * 4 < (1ll << 31), "residual must be < 2 GiB");
    k.mtiles = 0; k.ntiles = 0; k.nband = 0;
    return ISEGMI_OK;
}

int conv2d_launch(const isegmi_conv_desc* d, const float* in, const float* w, const float* scale, const float* shift,
                  const float* res, float* out, hipStream_t st) {
    ConvK k;
    const int rc_fill = conv_fill(d, in, w, scale, shift, res, out, k);
    if (rc_fill) return rc_fill;
    int tile = d->tile;
    if (tile == 0) {
        // Tile rule, refitted in round 2 on per-layer sweeps of both models at bs 1 / 2 / 8 with every kernel forced in turn
        // (tools/conv_tile_sweep.py -> profiles/r02_conv_tile_sweep.txt; t64 = number of 64x64 tiles of the layer, nck = K chunks):
        //   * grids of many 64x64 rounds: the v2 schedule (tile 10, loads two chunks ahead; tile 12, four chunks ahead, from
        //     K = 2304 on) -- 3-8 % ahead of the round-1 64x64 kernel on every layer; the round-1 kernel (tile 3) keeps the stem;
        //   * the three 16x16x4 kernels take the small grids (second sweep, after the per-chunk vector work was cut to two
        //     instructions, profiles/r02_conv_tile_sweep_v2.txt): the loader-wave variant (tile 5) up to 176 tiles, the 32x32 block
        //     (tile 4) to 480 tiles, the 32x64 block (tile 6) for K >= 1024 where a grid sits just past a whole number of 64x64
        //     rounds (513-640, 1025-1100);
        //   * outputs of at most 32 channels never use a 64-wide tile (padding them to 64 wastes half the MFMA work);
        //   * fourth sweep (profiles/r02_conv_tile_sweep_v4.txt): the hybrid launch (tile 13) wherever a 513-2600-tile grid leaves few tiles over.
        // Round 3 tried to replace this ladder by a fitted cost model (tools/fit_conv_model.py: per kernel f + max(q (nck m + em), ceil(q / occ)
        // (nck l + el)), five parameters each, fitted on 355 layer shapes of both models at bs 1 / 2 / 8 and five further canvases): the model's
        // choices cost +1.4 % over the per-layer best in total, the ladder's +0.85 % (+0.55 % after the change below) -- the ladder stays; the CU
        // count it uses is the device's.
        const int64_t t64 = (int64_t)cdiv(k.M, 64) * cdiv(d->Cout, 64);
        const int nck = k.nchunks;
        const int ncu = device_cu_count();
        const int v2 = nck >= 72 ? 12 : 10;
        if (is_stem(d)) tile = 3;
        else if (t64 <= 176) tile = 5;
        else if (nck <= 2 && d->Cout > 32) tile = v2;  // K = 64: after the one-add epilogue the 64x64 v2 tile leads at every grid size
        else if (d->Cout <= 32 || t64 <= 480) tile = 4;
        // hybrid launch (tile 13: v2 on the rows that fill the CUs a whole number of times, 32x32 blocks on the rest) wherever the
        // left-over 64x64 tiles are few: 3-12 % ahead of the rule below on 526-2400-tile layers in the fourth sweep
        // (profiles/r02_conv_tile_sweep_v4.txt), behind it once the tail passes ~15 % of the layer (616, 1228 tiles)
        // round 3 sweep over 355 layer shapes incl. the canvases COCODemo produces (profiles/r03_conv_tile_sweep.txt): no upper limit any more -- on
        // the 3400-9500-tile 3x3 layers (FPN / RPN at P2, proto_net.8) the hybrid launch is 0.5-2.6 % ahead of plain v2 as well
        else if (t64 >= 513 && t64 % ncu != 0 && (t64 % ncu) * 100 <= t64 * 15) tile = 13;
        else if (t64 <= 512) tile = v2;
        else if (t64 <= 640) tile = nck >= 32 ? 6 : 4;
        else if (t64 <= 1024) tile = v2;
        else if (t64 <= 1100) tile = nck >= 32 ? 6 : 4;
        else tile = v2;
    }
    if (tile >= 4 && is_stem(d)) tile = 3;  // only the original 64x64 kernel has the stem path
    if (tile == 13 || tile == 14) {  // hybrid: whole-CU multiples of 64x64 tiles on v2, the remaining rows on 32x32 blocks, one launch
        const int nt64 = cdiv(d->Cout, 64);
        const int64_t mt64 = cdiv(k.M, 64);
        const int ncu = device_cu_count();
        int64_t main_mt = (mt64 * nt64 / ncu) * ncu / nt64;   // 64-row tile rows whose tiles fill the CUs a whole number of times
        if (main_mt * 64 > k.M) main_mt = k.M / 64;
        if (main_mt <= 0 || main_mt >= mt64) {                 // nothing to split: the plain v2 launch
            tile = tile == 13 ? 10 : 12;
        } else {
            ConvK km = k, kt = k;
            km.mtiles = (int)main_mt; km.ntiles = nt64;
            kt.m_begin = (int)(main_mt * 64);
            kt.mtiles = cdiv(k.M - kt.m_begin, 32); kt.ntiles = cdiv(d->Cout, 32);
            const int64_t in_tail = (int64_t)((double)k.in_touched * (k.M - kt.m_begin) / k.M);
            conv_set_band(km, 64, 4, k.in_touched - in_tail);
            conv_set_band(kt, 32, 4, in_tail);
            const int nmain = km.mtiles * km.ntiles, ntail = kt.mtiles * kt.ntiles;
            const size_t lds = 2 * (size_t)(64 + 64) * LDS_ROW * sizeof(float);
            if (tile == 13) hipLaunchKernelGGL((conv_hybrid_kernel<2>), dim3((unsigned)(nmain + ntail)), dim3(256), lds, st, km, kt, nmain);
            else hipLaunchKernelGGL((conv_hybrid_kernel<4>), dim3((unsigned)(nmain + ntail)), dim3(256), lds, st, km, kt, nmain);
            HIP_TRY(hipGetLastError());
            return ISEGMI_OK;
        }
    }
    if (tile == 15) {
        // FIXED-TREE SPLIT-K (round 6).  NOT the default numerics: the output is ((p0 + p1) + p2) + p3 of four k-ordered partial chains over
        // the chunk ranges [q L, (q + 1) L), L = ceil(nchunks / 4), instead of ONE chain -- another valid fp32 evaluation of the same sum, restated by the
        // oracle as ora_conv2d_split(..., 4) and bit-exact against it.  The engines take it only under `conv_split_k` (default 0) and only for the backbone
        // layers the oracle models split by the same rule (isegmi_conv_split_qualifies); bs = 8 / bs = 2 numerics and every default-mode test are untouched.
        ARG_CHECK(!is_stem(d), "split-K: not for the stem");
        constexpr int KS = 4;
        k.mtiles = cdiv(k.M, 32);
        k.ntiles = cdiv(d->Cout, 32);
        conv_set_band(k, 32, 1, k.in_touched);
        const size_t lds = (size_t)KS * 2 * (32 + 32) * 34 * sizeof(float);
        LDS_LIMIT_ONCE((int)lds, conv_mfma16_split_kernel<KS>);
        hipLaunchKernelGGL((conv_mfma16_split_kernel<KS>), dim3((unsigned)(k.mtiles * k.ntiles)), dim3(256 * KS), lds, st, k);
        HIP_TRY(hipGetLastError());
        return ISEGMI_OK;
    }
    if (tile >= 7) {
        ARG_CHECK(tile != 8 && tile != 11, "tiles 8 / 11 (ring of 3) were dropped: the LDS stage of an unrolled position is its parity");
        k.mtiles = cdiv(k.M, 64);
        k.ntiles = cdiv(d->Cout, 64);
        const size_t lds = 2 * (size_t)(64 + 64) * LDS_ROW * sizeof(float);
        conv_set_band(k, 64, 4, k.in_touched);
        const dim3 g7((unsigned)(k.mtiles * k.ntiles));
        if (tile == 7) hipLaunchKernelGGL((conv_mfma_v2_kernel<2, true>), g7, dim3(256), lds, st, k);
        else if (tile == 9) hipLaunchKernelGGL((conv_mfma_v2_kernel<4, true>), g7, dim3(256), lds, st, k);
        else if (tile == 10) hipLaunchKernelGGL((conv_mfma_v2_kernel<2, false>), g7, dim3(256), lds, st, k);
        else hipLaunchKernelGGL((conv_mfma_v2_kernel<4, false>), g7, dim3(256), lds, st, k);
        HIP_TRY(hipGetLastError());
        return ISEGMI_OK;
    }
    if (tile >= 4) {
        k.mtiles = cdiv(k.M, 32);
        k.ntiles = cdiv(d->Cout, tile == 6 ? 64 : 32);
        conv_set_band(k, tile == 6 ? 64 : 32, tile == 4 ? 8 : 4, k.in_touched);
        const dim3 g16((unsigned)(k.mtiles * k.ntiles));
        if (tile == 5) hipLaunchKernelGGL((conv_mfma16_kernel<8, true>), g16, dim3(512), 0, st, k);
        else if (tile == 6) hipLaunchKernelGGL((conv_mfma16_kernel<4, false, 2>), g16, dim3(256), 0, st, k);
        else hipLaunchKernelGGL((conv_mfma16_kernel<4, false>), g16, dim3(256), 0, st, k);
        HIP_TRY(hipGetLastError());
        return ISEGMI_OK;
    }
    switch (tile) {
        case 1: return launch<128, 128, 2, 2>(d, k, st);
        case 2: return launch<128, 64, 2, 2>(d, k, st);
        default: return launch<64, 64, 2, 2>(d, k, st);
    }
}

// n independent fp32 convolutions as one launch (see conv_group_kernel).  Parts: a conv of at least 256 tiles of 64 x 64 and more than 32 output channels
// runs on v2 tiles when the group as a whole fills the chip twice over (> 480 such tiles), everything else on 32 x 32 blocks.  The tile FORM is chosen
// per group, not per member as a member's own launch would choose it (one LDS ring depth for all v2 parts: 4 as soon as any of them walks >= 72
// chunks; the small / narrow members on 32 x 32 blocks whatever conv2d_launch's ladder would pick): same results -- every fp32 tile form keeps the
// single k-ordered chain per output and is oracle-exact -- not the same launch.
int conv2d_group_launch(int n, const isegmi_conv_desc* const* d, const float* const* in, const float* const* w, const float* const* scale,
                        const float* const* shift, const float* const* res, float* const* out, hipStream_t st) {
    ARG_CHECK(n >= 1 && n <= CONV_GROUP_MAX && d && in && w && out, "conv group: 1..10 members");
    ConvGroupK g;
    ConvK ks[CONV_GROUP_MAX];
    int64_t t64[CONV_GROUP_MAX], total64 = 0;
    for (int i = 0; i < n; ++i) {
        ARG_CHECK(d[i] && !is_stem(d[i]) && d[i]->tile == 0, "conv group: no stem, no forced tile");
        ARG_CHECK(in[i] && w[i] && out[i], "conv group: a member's input, weights or output pointer is NULL");
        const int rc = conv_fill(d[i], in[i], w[i], scale ? scale[i] : nullptr, shift ? shift[i] : nullptr, res ? res[i] : nullptr, out[i], ks[i]);
        if (rc) return rc;
        t64[i] = (int64_t)cdiv(ks[i].M, 64) * cdiv(d[i]->Cout, 64);
        total64 += t64[i];
    }
    int order[CONV_GROUP_MAX], kind[CONV_GROUP_MAX];
    int ring = 2;
    for (int i = 0; i < n; ++i) {
        order[i] = i;
        kind[i] = (total64 > 480 && t64[i] >= 256 && d[i]->Cout > 32) ? 0 : 1;
        if (kind[i] == 0 && ks[i].nchunks >= 72) ring = 4;
    }
    // v2 parts first, each kind by falling size (insertion sort: n <= 10)
    for (int a = 1; a < n; ++a)
        for (int b = a; b > 0; --b) {
            const int x = order[b - 1], y = order[b];
            const bool swap = kind[y] < kind[x] || (kind[y] == kind[x] && t64[y] > t64[x]);
            if (!swap) break;
            order[b - 1] = y; order[b] = x;
        }
    g.n = n;
    g.blk0[0] = 0;
    for (int j = 0; j < n; ++j) {
        const int i = order[j];
        ConvK& k = ks[i];
        const int bt = kind[i] == 0 ? 64 : 32;
        k.mtiles = cdiv(k.M, bt);
        k.ntiles = cdiv(d[i]->Cout, bt);
        conv_set_band(k, bt, kind[i] == 0 ? 4 : 8, k.in_touched);
        const int64_t blocks = (int64_t)k.mtiles * k.ntiles;
        ARG_CHECK(g.blk0[j] + blocks < (1ll << 30), "conv group: too many tiles");
        g.k[j] = k;
        g.kind[j] = kind[i];
        g.blk0[j + 1] = g.blk0[j] + (int)blocks;
    }
    const size_t lds = 2 * (size_t)(64 + 64) * LDS_ROW * sizeof(float);
    if (ring == 4) hipLaunchKernelGGL((conv_group_kernel<4>), dim3((unsigned)g.blk0[n]), dim3(256), lds, st, g);
    else hipLaunchKernelGGL((conv_group_kernel<2>), dim3((unsigned)g.blk0[n]), dim3(256), lds, st, g);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}

}  // namespace isegmi

using namespace isegmi;

extern "C" int isegmi_op_conv2d_group(int n, const isegmi_conv_desc* descs, const float* const* d_in, const float* const* d_w, const float* const* d_scale,
                                      const float* const* d_shift, const float* const* d_res, float* const* d_out, void* stream) {
    ARG_CHECK(n >= 1 && n <= CONV_GROUP_MAX && descs, "conv group: 1..10 members");
    const isegmi_conv_desc* dp[CONV_GROUP_MAX];
    for (int i = 0; i < n; ++i) dp[i] = descs + i;
    return conv2d_group_launch(n, dp, d_in, d_w, d_scale, d_shift, d_res, d_out, (hipStream_t)stream);
}

// the shape rule of the split-K mode (mirrored by oracle/ora.py conv_split_qualifies): a layer of at most 176 tiles of 64 x 64 -- the grids the 16 x 16 x 4
// kernels run at about one block per CU -- with at least 32 K chunks (K >= 1024)
extern "C" int isegmi_conv_split_qualifies(const isegmi_conv_desc* d) {
    if (!d || check_desc(d) != ISEGMI_OK || is_stem(d)) return 0;
    const int Ho = (d->H + 2 * d->pad - d->R) / d->stride + 1, Wo = (d->W + 2 * d->pad - d->S) / d->stride + 1;
    const int64_t M = (int64_t)d->N * Ho * Wo;
    const int64_t t64 = ((M + 63) / 64) * ((d->Cout + 63) / 64);
    return t64 <= 176 && n_chunks(d) >= 32 ? 1 : 0;
}

extern "C" int isegmi_conv_out_hw(const isegmi_conv_desc* d, int32_t* Ho, int32_t* Wo) {
    int rc = check_desc(d);
    if (rc) return rc;
    *Ho = (d->H + 2 * d->pad - d->R) / d->stride + 1;
    *Wo = (d->W + 2 * d->pad - d->S) / d->stride + 1;
    return ISEGMI_OK;
}

extern "C" int isegmi_conv_packed_floats(const isegmi_conv_desc* d, int64_t* n) {
    int rc = check_desc(d);
    if (rc) return rc;
    *n = (int64_t)cout_pad(d) * n_chunks(d) * 32;
    return ISEGMI_OK;
}

extern "C" int isegmi_pack_conv_weights(const isegmi_conv_desc* d, const float* w, float* packed) {
    int rc = check_desc(d);
    if (rc) return rc;
    ARG_CHECK(w && packed, "null pointer");
    const int nch = n_chunks(d);
    const int64_t wrow = (int64_t)nch * 32;
    const int64_t total = (int64_t)cout_pad(d) * wrow;
    for (int64_t i = 0; i < total; ++i) packed[i] = 0.0f;
    const int K = d->R * d->S * d->Cin;
    for (int co = 0; co < d->Cout; ++co) {
        const float* src = w + (int64_t)co * K;
        float* dst = packed + (int64_t)co * wrow;
        if (is_stem(d)) {
            for (int r = 0; r < 7; ++r)
                for (int e = 0; e < 28; ++e) {  // e = s*4 + c
                    const int grp = e >> 3, pos = perm8(e & 7);
                    dst[r * 32 + grp * 8 + pos] = src[r * 28 + e];
                }
        } else {
            for (int k = 0; k < K; ++k) dst[(k & ~7) + perm8(k & 7)] = src[k];
        }
    }
    return ISEGMI_OK;
}

extern "C" int isegmi_op_conv2d(const isegmi_conv_desc* d, const float* d_in, const float* d_w, const float* d_scale,
                                const float* d_shift, const float* d_res, float* d_out, void* stream) {
    ARG_CHECK(d_in && d_w && d_out, "null device pointer");
    return conv2d_launch(d, d_in, d_w, d_scale, d_shift, d_res, d_out, (hipStream_t)stream);
}
