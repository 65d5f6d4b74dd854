// bottleneck_f16.hip -- one ResNet identity bottleneck as ONE kernel (gfx950, fp16 storage, v_mfma_f32_32x32x16_f16):
//     t1 = relu(bn1(conv1x1(x)));  t2 = relu(bn2(conv3x3(t1)));  out = relu(bn3(conv1x1(t2)) + x)
// SURVEY 8a M2 `BottleneckWithFixedBatchNorm` under BASELINE configs[4] ("fp16 MFMA conv").  As three launches the K <= 256 layers of
// res2 / res3 are HBM-bound (round 3: 27 % of an R101 bs=8 step at 15.9 % MfmaUtil): per res2 block 1.1 GB cross HBM, of which the
// 64-channel intermediates t1 / t2 (4 x 69 MB) and the second read of x as the residual (275 MB) are avoidable.  Here a block owns a
// TH x TW spatial tile of output pixels: conv1 runs over the tile plus a one-pixel halo (recomputed per tile), t1 and t2 stay in LDS
// as fp16 -- rounded exactly where the three-launch path rounds them when it stores them, so results are bit-identical to it -- x is
// read once (the residual re-read of the tile's centre hits L2) and out is written once.
//
// Structure: persistent blocks of 8 MFMA waves + 4 LOADER waves (the loader-wave protocol of conv_f16_persist_kernel).  Everything
// that comes from memory through LDS -- conv1's x chunks, and the weight chunks of all three convolutions -- is one stream of
// fixed-size STEPS through an NSTAGE-deep ring that the loader waves fill by LDS-DMA (`buffer_load ... lds`) and that runs ahead across
// phase and tile boundaries; one barrier per step:
//     loader:  s_waitcnt vmcnt((NSTAGE-2) * PP); s_barrier; issue step t + NSTAGE - 1        (PP pieces of 1 KiB per loader wave and step,
//     MFMA:    s_barrier; fragments + MFMAs of step t from stage t % NSTAGE                    padded with dropped pieces so it is uniform)
// Steps of a tile: KC1 conv1 steps (x chunk [halo pixels x 64 ch] + w1 chunk), S2 conv2 steps (G2 chunks of w2 each; the A operand is
// t1 in LDS at a per-tap row offset), one extra barrier (t2 overwrites t1), S3 conv3 steps (w3 chunk of 256 couts x 64 k; the A
// operand is t2 in LDS), each 256-cout slab followed by its epilogue (fp32 strips through a per-wave LDS scratch, residual, ReLU, 16-B stores).
// conv1 and conv2 are computed TRANSPOSED (D[cout][pixel]: the weight fragment is the MFMA's A operand), so that a lane holds four
// consecutive channels of ONE pixel per accumulator register group and t1 / t2 are written to LDS with 8-byte stores in the row-major
// [pixel][channel] image the next convolution reads its k-contiguous fragments from.
#include "../../include/isegmi.h"
#include "common.h"

namespace isegmi {

typedef _Float16 half_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16h __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4h __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2h __attribute__((ext_vector_type(2)));
typedef float f32x4h __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct BtlK {
    const half_t* x;
    const half_t* w1;
    const half_t* w2;
    const half_t* w3;
    const float* s1; const float* b1;
    const float* s2; const float* b2;
    const float* s3; const float* b3;
    const half_t* wd;            // projection shortcut (first block of a stage): 1x1 CIN -> COUT on x, its own folded BN
    const float* sd; const float* bd;
    const half_t* res;
    half_t* out;
    int N, H, W;                 // x is [N][H][W][CIN]; the output has the same H x W (stride 1)
    int tiles_x, tiles_y, total;
    unsigned x_bytes, out_bytes, res_bytes, w1_bytes, w2_bytes, w3_bytes, wd_bytes;
    int exp_flags;  // TIMING-ONLY experiments: 32 the loader waves issue nothing, 64 the MFMA waves only keep the barriers;
                    // 128 / 256 / 512 / 1024 / 2048 / 4096: the MFMA waves skip conv1's K loop / its epilogue / conv2's K loop / its epilogue / conv3's K loop / its epilogue
};

// DS: the block has a PROJECTION shortcut (first block of res2: CIN = 64 -> COUT = 256, stride 1): the residual is
//   r = fp16(bnd(conv1x1_d(x)))  -- rounded to fp16 as the three-launch path rounds the shortcut tensor when it stores it --
// computed in two extra steps between conv2 and conv3 (the tile's centre pixels of x once more through the ring, L2 hits, next to one half
// of the projection's weights each), transposed through the epilogue's scratch into the registers the identity kernel loads the residual into.
template <int CIN_, int CMID_, int TH_, int TW_, bool DS_ = false>
struct BtlCfg {
    static constexpr int CIN = CIN_, CMID = CMID_, COUT = 4 * CMID_, TH = TH_, TW = TW_;
    static constexpr bool DS = DS_;
    static constexpr int SD = DS_ ? 2 : 0;        // projection steps per tile
    static constexpr int NW = 8, LW = 4, NSTAGE = 3;
    static constexpr int HW2 = TW + 2, MH = (TH + 2) * (TW + 2), MHP = (MH + 31) / 32 * 32, MT = TH * TW, MTP = (MT + 31) / 32 * 32;
    static constexpr int KC1 = CIN / 64, KM = CMID / 64;
    static constexpr int P1B = CMID * 2;          // row pitch (bytes) of the t1 / t2 images
    static constexpr int NCOL = CMID / 8;         // 16-B columns per row
    static constexpr int CH2 = CMID * 128;        // one conv2 weight chunk [CMID couts x 64 k]
    static constexpr int G2 = KM == 1 ? 3 : 2;    // conv2 chunks per step
    static constexpr int S1 = KC1, S2 = 9 * KM / G2, S3 = (COUT / 256) * KM;
    // 1-KiB pieces per loader wave and step
    static constexpr int PA1 = MHP / 8 / LW, PW1 = CMID / 8 / LW, PW2 = CMID / 8 / LW, PP2 = G2 * PW2, PP3 = 256 / 8 / LW;
    static constexpr int PP = PA1 + PW1;
    static constexpr int P1_BYTES = MHP * 128 + CMID * 128, P2_BYTES = G2 * CH2, P3_BYTES = 256 * 128;
    static constexpr int max3(int a, int b, int c) { return a > b ? (a > c ? a : c) : (b > c ? b : c); }
    // a step short of PP real pieces is padded with DROPPED pieces (zero-length descriptor: zeros land in LDS) so that the loaders' counted
    // vmcnt wait is uniform; each goes to its own KiB at the unused end of the stage -- hipcc merges LDS-DMA builtins that are identical, and
    // a merged piece makes the count one short (seen: garbage in t1 on cold caches)
    static constexpr int STAGEB = max3(P1_BYTES, P2_BYTES + (PP - PP2) * 1024, P3_BYTES + (PP - PP3) * 1024);
    static_assert(!DS_ || (CIN_ == 64 && CMID_ == 64), "the projection variant is res2's first block");
    static constexpr int RD = CMID == 64 ? 8 : 4; // residual passes (16 B per lane each) in flight: requested before conv2, refilled pass by pass
    static constexpr int EPITCH = 68;             // floats per scratch row (64 + 4)
    static constexpr int SCRATCH = NW * 8 * EPITCH * 4;
    static constexpr int REGION = MHP * P1B > MTP * P1B + SCRATCH ? MHP * P1B : MTP * P1B + SCRATCH;
    static constexpr int TABLE = 4 * CMID * 4;
    static constexpr int LDS = NSTAGE * STAGEB + REGION + TABLE;
    static_assert(CIN % 64 == 0 && (CMID == 64 || CMID == 128), "channel counts");
    static_assert((MHP / 8) % LW == 0 && (CMID / 8) % LW == 0, "whole piece rounds per loader wave");
    static_assert(PP2 <= PP && PP3 <= PP, "conv1 steps carry the most pieces");
    static_assert(MTP == 128, "conv3's wave layout is 2 x 4 waves of 64 x 64 over a 128-pixel tile");
    static_assert((9 * KM) % G2 == 0 && COUT % 256 == 0, "step split");
    static_assert(STAGEB % 1024 == 0 && LDS <= 163840, "LDS budget");
};

// 16-B column swizzle of row `r` of a t1 / t2 image (conflict-free ds_read_b128 over rows at one logical column)
template <int NCOL>
__device__ __forceinline__ int mid_sw(int r) { return NCOL == 8 ? ((r >> 1) & 7) : (r & 15); }

// Pins the interleave of a K-loop body of NG groups, each RPG fragment reads (ds_read_b128) feeding MPG MFMAs, as a software pipeline whose
// reads run LA groups ahead of the MFMAs that consume them.  Left to itself hipcc keeps ONE read in flight per MFMA (s_waitcnt lgkmcnt(1) in
// front of every v_mfma): with two MFMA waves per SIMD that exposes an LDS latency per MFMA and the waves ran at a third of the matrix rate.
template <int NG, int RPG, int MPG, int LA>
__device__ __forceinline__ void pipeline_reads_mfmas() {
    constexpr int L = LA < NG ? LA : NG;
    __builtin_amdgcn_sched_group_barrier(0x100, L * RPG, 0);
#pragma unroll
    for (int i = 0; i < NG - L; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, MPG, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, RPG, 0);
    }
#pragma unroll
    for (int i = 0; i < L; ++i) __builtin_amdgcn_sched_group_barrier(0x008, MPG, 0);
}

template <class CF>
__global__ __launch_bounds__((CF::NW + CF::LW) * 64, 1) void bottleneck_f16_kernel(const BtlK p) {
    constexpr int CIN = CF::CIN, CMID = CF::CMID, COUT = CF::COUT, TH = CF::TH, TW = CF::TW, NW = CF::NW, NSTAGE = CF::NSTAGE;
    constexpr int HW2 = CF::HW2, MH = CF::MH, MHP = CF::MHP, MT = CF::MT, MTP = CF::MTP, KC1 = CF::KC1, KM = CF::KM, P1B = CF::P1B;
    constexpr int NCOL = CF::NCOL, CH2 = CF::CH2, G2 = CF::G2, S1 = CF::S1, S2 = CF::S2, S3 = CF::S3, PP = CF::PP, STAGEB = CF::STAGEB;
    constexpr bool DS = CF::DS;
    constexpr int SD = CF::SD;
    constexpr unsigned OOB = 0x80000000u;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const ring = smem;
    char* const mid = smem + NSTAGE * STAGEB;
    float* const tab = (float*)(mid + CF::REGION);  // [s1 | b1 | s2 | b2], CMID floats each

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int total = p.total, G = (int)gridDim.x, bid = (int)blockIdx.x;
    const int q8 = total >> 3, r8g = total & 7;
    const int tiles_img = p.tiles_x * p.tiles_y;
    auto tile_origin = [&](int v, int& n, int& y0, int& x0) {  // v = bid + i * G keeps v & 7 (G is a multiple of 8 or the whole grid)
        const int xcd = v & 7;
        const int logical = (xcd < r8g ? xcd * (q8 + 1) : r8g * (q8 + 1) + (xcd - r8g) * q8) + (v >> 3);
        n = logical / tiles_img;
        const int t = logical - n * tiles_img;
        const int ty = t / p.tiles_x;
        y0 = ty * TH;
        x0 = (t - ty * p.tiles_x) * TW;
    };
    const int my_tiles = (total - bid + G - 1) / G;

    for (int i = tid; i < 4 * CMID; i += (NW + CF::LW) * 64) {
        const float* src = i < CMID ? p.s1 : i < 2 * CMID ? p.b1 : i < 3 * CMID ? p.s2 : p.b2;
        tab[i] = src[i & (CMID - 1)];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the table is complete at every wave's first barrier

    if (wave >= NW) {
        // ------------------------------------------------------------------------------------------------ loader waves
        constexpr int PA1 = CF::PA1, PW1 = CF::PW1, PW2 = CF::PW2, PP2 = CF::PP2, PP3 = CF::PP3, LW = CF::LW;
        const int lw = wave - NW;
        const int r8 = lane >> 3, cs = lane & 7;
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.x_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_w1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1, 0, p.w1_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_w2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, p.w2_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_w3 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w3, 0, p.w3_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, 0, 0x00020000);  // drops every load
        const __amdgpu_buffer_rsrc_t rs_wd = __builtin_amdgcn_make_buffer_rsrc((void*)(DS ? p.wd : p.w1), 0, DS ? p.wd_bytes : 0u, 0x00020000);
        constexpr int PCD = DS ? MTP / 8 / LW : 1, PWD = 128 / 8 / LW;   // projection step: the tile's centre pixels of x + 128 rows of wd
        static_assert(!DS || PCD + PWD == PP, "projection steps carry PP pieces");
        unsigned w1off[PW1], w2off[PW2], w3off[PP3], avoff[PA1], cvoff[PCD], wdoff[PWD];
#pragma unroll
        for (int i = 0; i < PW1; ++i) {
            const int row = (lw + i * LW) * 8 + r8, c = cs ^ ((row >> 1) & 7);
            w1off[i] = (unsigned)(row * (CIN * 2) + c * 16);
            w2off[i] = (unsigned)(row * (9 * CMID * 2) + c * 16);
        }
#pragma unroll
        for (int i = 0; i < PP3; ++i) {
            const int row = (lw + i * LW) * 8 + r8, c = cs ^ ((row >> 1) & 7);
            w3off[i] = (unsigned)(row * (CMID * 2) + c * 16);
        }
#pragma unroll
        for (int i = 0; i < PWD; ++i) {
            const int row = (lw + i * LW) * 8 + r8, c = cs ^ ((row >> 1) & 7);
            wdoff[i] = (unsigned)(row * (CIN * 2) + c * 16);
        }
        auto setup_tile = [&](int v) {
            int n, y0, x0;
            tile_origin(v, n, y0, x0);
#pragma unroll
            for (int i = 0; i < PA1; ++i) {
                const int j = (lw + i * LW) * 8 + r8, c = cs ^ ((j >> 1) & 7);
                const int hy = j / HW2, hx = j - hy * HW2;
                const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
                const bool ok = (j < MH) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
                avoff[i] = ok ? (unsigned)((((n * p.H + iy) * p.W + ix) * CIN) * 2 + c * 16) : OOB;
            }
            if (DS) {
#pragma unroll
                for (int i = 0; i < PCD; ++i) {
                    const int m = (lw + i * LW) * 8 + r8, c = cs ^ ((m >> 1) & 7);
                    const int y = m / TW, x = m - y * TW;
                    const bool ok = (m < MT) & (y0 + y < p.H) & (x0 + x < p.W);
                    cvoff[i] = ok ? (unsigned)((((n * p.H + y0 + y) * p.W + x0 + x) * CIN) * 2 + c * 16) : OOB;
                }
            }
        };
        int iv = bid, iphase = 0, iidx = 0;
        bool live = true;
        setup_tile(iv);
        auto dummies = [&](char* base, int n) {  // n dropped pieces, each to a KiB of its own at the end of the stage
#pragma unroll
            for (int i = 0; i < n; ++i) {
                const unsigned voff = 0u, soff = 0u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_0, (lds_ptr_t)(base + STAGEB - 1024 * (i + 1)), 16, voff, soff, 0, 0);
            }
        };
        auto issue = [&](int stage) {  // this wave's PP pieces of the next step of the stream, then advance the stream
            if (p.exp_flags & 32) return;
            char* base = ring + stage * STAGEB;
            if (!live) { dummies(base, PP); return; }
            if (iphase == 0) {
                const unsigned soff = (unsigned)iidx * 128u;
#pragma unroll
                for (int i = 0; i < PA1; ++i) {
                    const unsigned voff = avoff[i];
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(base + (lw + i * LW) * 1024), 16, voff, soff, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < PW1; ++i) {
                    const unsigned voff = w1off[i];
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w1, (lds_ptr_t)(base + MHP * 128 + (lw + i * LW) * 1024), 16, voff, soff, 0, 0);
                }
                if (++iidx == S1) { iphase = 1; iidx = 0; }
            } else if (iphase == 1) {
#pragma unroll
                for (int g = 0; g < G2; ++g) {
                    const unsigned soff = (unsigned)(iidx * G2 + g) * 128u;  // packed k = (tap, cin): chunk q = tap * KM + kc starts at byte 128 q
#pragma unroll
                    for (int i = 0; i < PW2; ++i) {
                        const unsigned voff = w2off[i];
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w2, (lds_ptr_t)(base + g * CH2 + (lw + i * LW) * 1024), 16, voff, soff, 0, 0);
                    }
                }
                dummies(base, PP - PP2);
                if (++iidx == S2) { iphase = DS ? 2 : 3; iidx = 0; }
            } else if (DS && iphase == 2) {   // projection step h: centre pixels of x [MTP x 64 ch] + rows 128 h .. of wd
                const unsigned soffw = (unsigned)(iidx * 128 * CIN * 2), soff0 = 0u;
#pragma unroll
                for (int i = 0; i < PCD; ++i) {
                    const unsigned voff = cvoff[i];
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(base + (lw + i * LW) * 1024), 16, voff, soff0, 0, 0);
                }
#pragma unroll
                for (int i = 0; i < PWD; ++i) {
                    const unsigned voff = wdoff[i];
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_wd, (lds_ptr_t)(base + MTP * 128 + (lw + i * LW) * 1024), 16, voff, soffw, 0, 0);
                }
                if (++iidx == SD) { iphase = 3; iidx = 0; }
            } else {
                const int slab = iidx / KM, kc = iidx - slab * KM;
                const unsigned soff = (unsigned)(slab * 256 * CMID * 2 + kc * 128);
#pragma unroll
                for (int i = 0; i < PP3; ++i) {
                    const unsigned voff = w3off[i];
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w3, (lds_ptr_t)(base + (lw + i * LW) * 1024), 16, voff, soff, 0, 0);
                }
                dummies(base, PP - PP3);
                if (++iidx == S3) {
                    iphase = 0; iidx = 0;
                    iv += G;
                    live = iv < total;
                    if (live) setup_tile(iv);
                }
            }
        };
#pragma unroll
        for (int s = 0; s < NSTAGE - 1; ++s) issue(s);
        int wr = NSTAGE - 1;
        for (int i = 0; i < my_tiles; ++i) {
            for (int t = 0; t < S1 + S2; ++t) {
                asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NSTAGE - 2) * PP) : "memory");
                issue(wr);
                wr = wr + 1 == NSTAGE ? 0 : wr + 1;
            }
            asm volatile("s_barrier" ::: "memory");  // X: see the MFMA waves
            for (int t = 0; t < SD + S3; ++t) {
                asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NSTAGE - 2) * PP) : "memory");
                issue(wr);
                wr = wr + 1 == NSTAGE ? 0 : wr + 1;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // trailing dropped pieces have landed
        return;
    }

    // ---------------------------------------------------------------------------------------------------- MFMA waves
    // (the per-lane constants of the tile loop are derived from an OPAQUE copy of the lane id inside the loop: hipcc otherwise hoists every
    // loop-invariant address -- fragment offsets per k-step, epilogue rows, table slots: ~100 registers -- out of the tile loop and spills them)
    const int lr0 = lane & 31, lh0 = lane >> 5;
    // conv1 (transposed): wave w < NRT owns the halo-pixel tile w and all NCT1 cout tiles (one x fragment feeds NCT1 MFMAs); the other
    // waves only keep the barriers (MFMA time is a quarter of a tile's HBM time: balance is not what bounds this kernel)
    constexpr int NCT1 = CMID / 32, NRT = MHP / 32;
    static_assert(NRT <= NW, "one halo-pixel tile per wave");
    const bool act1 = wave < NRT;
    // conv2 (transposed): pixel tiles pt = wp + NWP * i, cout tiles wc * NCW + b
    constexpr int NPT = MTP / 32, NWP = NPT < NW ? NPT : NW, NWC = NW / NWP, NPW = NPT / NWP, NCT2 = CMID / 32, NCW = NCT2 / NWC;
    static_assert(NPT % NWP == 0 && NCT2 % NWC == 0 && NCW >= 1, "conv2 wave layout");
    const int wp2 = wave % NWP, wc2 = wave / NWP;
    int a2off[NPW][9];  // byte offset of the lane's t1 row for tap (r, s), k-step 0; ^ (ks << 5) ^ (kc << 7)
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const int m = (wp2 + NWP * i) * 32 + lr0, mm = m < MT ? m : 0;
        const int y = mm / TW, x = mm - y * TW;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int j = (y + tap / 3) * HW2 + x + tap % 3;
            a2off[i][tap] = j * P1B + ((lh0 ^ mid_sw<NCOL>(j)) << 4);
        }
    }
    // conv3 (standard): 2 x 4 waves of 64 pixels x 64 couts per 256-cout slab
    constexpr int TM3 = NPT / 2;
    const int wn3 = wave & 3, wm3 = wave >> 2;
    int a3off[TM3];
#pragma unroll
    for (int a = 0; a < TM3; ++a) {
        const int row = (wm3 * TM3 + a) * 32 + lr0;
        a3off[a] = row * P1B + ((lh0 ^ mid_sw<NCOL>(row)) << 4);
    }
    float* const ew = (float*)(mid + MTP * P1B) + wave * 8 * CF::EPITCH;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, p.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)p.res, 0, p.res_bytes, 0x00020000);

    int st = 0;
    auto next_stage = [&]() { st = st + 1 == NSTAGE ? 0 : st + 1; };
    if (p.exp_flags & 64) {
        for (int i = 0; i < my_tiles; ++i)
            for (int t = 0; t < S1 + S2 + 1 + SD + S3; ++t) asm volatile("s_barrier" ::: "memory");
        return;
    }
    for (int v = bid; v < total; v += G) {
        int n, y0, x0;
        tile_origin(v, n, y0, x0);
        int lane_t = lane;
        asm volatile("" : "+v"(lane_t));
        const int lr = lane_t & 31, lh = lane_t >> 5;
        const int swz = lr * 128 + ((lh ^ ((lr >> 1) & 7)) << 4);  // fragment of a 128-B-row chunk image; k-step ks: ^ (ks << 5)
        // opaque per tile: left visibly loop-invariant, the 9 x 4 x KM (tap, k-step, chunk) variants of these offsets are all hoisted out of the
        // tile loop and spilled (108 VGPR spills in the 512-channel kernel); one v_xor per fragment read costs nothing here (HBM-bound)
#pragma unroll
        for (int i = 0; i < NPW; ++i)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) asm volatile("" : "+v"(a2off[i][tap]));
#pragma unroll
        for (int a = 0; a < TM3; ++a) asm volatile("" : "+v"(a3off[a]));

        // ---- conv1 over the halo tile: t1[j][c] = relu(bn1(sum_k x[j][k] w1[c][k])), zero where pixel j lies outside the image
        {
            f32x16h acc[NCT1];
#pragma unroll
            for (int b = 0; b < NCT1; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[b][e] = 0.0f;
            for (int kc = 0; kc < KC1; ++kc) {
                asm volatile("s_barrier" ::: "memory");
                if (act1 && !(p.exp_flags & 128)) {  // uniform
                    const char* sb = ring + st * STAGEB;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const f16x8 xf = *(const f16x8*)(sb + wave * 4096 + (swz ^ (ks << 5)));
#pragma unroll
                        for (int b = 0; b < NCT1; ++b) {
                            const f16x8 wf = *(const f16x8*)(sb + MHP * 128 + b * 4096 + (swz ^ (ks << 5)));
                            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, xf, acc[b], 0, 0, 0);
                        }
                    }
                    pipeline_reads_mfmas<4, 1 + NCT1, NCT1, 2>();
                }
                next_stage();
            }
            if (act1 && !(p.exp_flags & 256)) {
                const int j = wave * 32 + lr;
                const int hy = j / HW2, hx = j - hy * HW2;
                const bool in_img = (j < MH) & ((unsigned)(y0 - 1 + hy) < (unsigned)p.H) & ((unsigned)(x0 - 1 + hx) < (unsigned)p.W);
                const int swj = mid_sw<NCOL>(j);
#pragma unroll
                for (int b = 0; b < NCT1; ++b)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int c0 = b * 32 + 8 * g + 4 * lh;
                        const f32x4h sc = *(const f32x4h*)(tab + c0), sh = *(const f32x4h*)(tab + CMID + c0);
                        f16x4 o;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float y = fmaf(acc[b][4 * g + i], sc[i], sh[i]);
                            // the fp32 value is rounded to fp32 FIRST, as in the three-launch path (which parks it in LDS): left to itself hipcc folds
                            // fmaf + the fp16 conversion into v_fma_mixlo_f16, ONE rounding -- 1-ulp differences in 3e-5 of the elements
                            asm("" : "+v"(y));
                            y = y > 0.0f ? y : 0.0f;
                            o[i] = (half_t)(in_img ? y : 0.0f);
                        }
                        *(u32x2h*)(mid + j * P1B + (((c0 >> 3) ^ swj) << 4) + 8 * lh) = __builtin_bit_cast(u32x2h, o);
                    }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // t1 is complete at the next barrier
        }

        // ---- the residual (this tile's pixels of x: L2 hits, the loaders have just streamed them) is REQUESTED here, a whole phase before the
        // epilogue adds it: RD passes of 16 B per lane in flight, refilled as the epilogue consumes them.  (With two passes requested right
        // before conv3's short K loop every pass of the epilogue waited out most of a memory latency.)
        const int er = lane_t >> 3, ec = (lane_t & 7) * 8;
        auto strip_off = [&](int slab, int q) -> unsigned {  // byte offset of this lane's 8 channels in pass q (8 pixels of the wave's tile)
            const int m = (wm3 * TM3) * 32 + q * 8 + er;
            const int y = m / TW, x = m - y * TW;
            const bool ok = (m < MT) & (y0 + y < p.H) & (x0 + x < p.W);
            return ok ? (unsigned)((((n * p.H + y0 + y) * p.W + x0 + x) * COUT + slab * 256 + wn3 * 64 + ec) * 2) : OOB;
        };
        constexpr int NQ = TM3 * 4, RD = CF::RD, NSLAB = COUT / 256;
        static_assert(NQ % RD == 0, "residual window");
        static_assert(!DS || (RD == NQ && NSLAB == 1), "the projection fills the whole residual window at once");
        u32x4h rw[RD];
        if (!DS) {
#pragma unroll
            for (int q = 0; q < RD; ++q) rw[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, strip_off(0, q), 0, 0);
        }

        // ---- conv2: t2[m][c] = relu(bn2(sum_{tap, k} t1[j(m) + tap][k] w2[c][tap][k]))
        {
            f32x16h acc[NPW][NCW];
#pragma unroll
            for (int i = 0; i < NPW; ++i)
#pragma unroll
                for (int b = 0; b < NCW; ++b)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][b][e] = 0.0f;
#pragma unroll
            for (int u = 0; u < S2; ++u) {
                asm volatile("s_barrier" ::: "memory");
                const char* sb = ring + st * STAGEB;
                if (p.exp_flags & 512) { next_stage(); continue; }
#pragma unroll
                for (int g = 0; g < G2; ++g) {
                    const int q = u * G2 + g, tap = q / KM, kc = q % KM;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        f16x8 wf[NCW];
#pragma unroll
                        for (int b = 0; b < NCW; ++b) wf[b] = *(const f16x8*)(sb + g * CH2 + (wc2 * NCW + b) * 4096 + (swz ^ (ks << 5)));
#pragma unroll
                        for (int i = 0; i < NPW; ++i) {
                            const f16x8 xf = *(const f16x8*)(mid + (a2off[i][tap] ^ (ks << 5) ^ (kc << 7)));
#pragma unroll
                            for (int b = 0; b < NCW; ++b) acc[i][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[b], xf, acc[i][b], 0, 0, 0);
                        }
                    }
                }
                pipeline_reads_mfmas<G2 * 4, NCW + NPW, NCW * NPW, 4>();
                next_stage();
            }
            asm volatile("s_barrier" ::: "memory");  // X: every wave has read t1 for the last time; t2 and the scratch overwrite it
#pragma unroll
            for (int i = 0; i < NPW; ++i) {
                if (p.exp_flags & 1024) break;
                const int m = (wp2 + NWP * i) * 32 + lr;
                const int swm = mid_sw<NCOL>(m);
#pragma unroll
                for (int b = 0; b < NCW; ++b)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int c0 = (wc2 * NCW + b) * 32 + 8 * g + 4 * lh;
                        const f32x4h sc = *(const f32x4h*)(tab + 2 * CMID + c0), sh = *(const f32x4h*)(tab + 3 * CMID + c0);
                        f16x4 o;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            float y = fmaf(acc[i][b][4 * g + k], sc[k], sh[k]);
                            asm("" : "+v"(y));  // (see conv1: no v_fma_mixlo_f16)
                            o[k] = (half_t)(y > 0.0f ? y : 0.0f);
                        }
                        *(u32x2h*)(mid + m * P1B + (((c0 >> 3) ^ swm) << 4) + 8 * lh) = __builtin_bit_cast(u32x2h, o);
                    }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // t2 is complete at the next barrier
        }

        // ---- projection shortcut (DS): r[m][c] = fp16(bnd(sum_k x[m][k] wd[c][k])) for the tile's 128 pixels x 256 couts, in the conv3 layout (2 x 4
        // waves of 64 x 64) so that its transposition lands in the residual window pass for pass; step h holds wd rows 128 h ..: the waves of
        // that half of the couts multiply, the others only keep the barrier
        if (DS) {
            f32x16h acc[TM3][2];
#pragma unroll
            for (int a = 0; a < TM3; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.0f;
            for (int h = 0; h < SD; ++h) {
                asm volatile("s_barrier" ::: "memory");
                if ((wn3 >> 1) == h) {  // uniform
                    const char* sb = ring + st * STAGEB;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        f16x8 af[TM3], bf[2];
#pragma unroll
                        for (int a = 0; a < TM3; ++a) af[a] = *(const f16x8*)(sb + (wm3 * TM3 + a) * 4096 + (swz ^ (ks << 5)));
#pragma unroll
                        for (int b = 0; b < 2; ++b) bf[b] = *(const f16x8*)(sb + MTP * 128 + ((wn3 & 1) * 2 + b) * 4096 + (swz ^ (ks << 5)));
#pragma unroll
                        for (int a = 0; a < TM3; ++a)
#pragma unroll
                            for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
                    }
                    pipeline_reads_mfmas<4, TM3 + 2, TM3 * 2, 2>();
                }
                next_stage();
            }
            float sc[2], sh[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int co = (wn3 * 2 + b) * 32 + lr;
                sc[b] = p.sd[co];
                sh[b] = p.bd[co];
            }
#pragma unroll
            for (int a = 0; a < TM3; ++a)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int q = a * 4 + g;
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int k = 0; k < 4; ++k) ew[(k + 4 * lh) * CF::EPITCH + b * 32 + lr] = fmaf(acc[a][b][4 * g + k], sc[b], sh[b]);
                    const f32x4h v0 = *(const f32x4h*)(ew + er * CF::EPITCH + ec);
                    const f32x4h v1 = *(const f32x4h*)(ew + er * CF::EPITCH + ec + 4);
                    f16x8 o;
#pragma unroll
                    for (int k = 0; k < 8; ++k) o[k] = (half_t)(k < 4 ? v0[k] : v1[k - 4]);   // the shortcut tensor as the three-launch path stores it
                    rw[q % RD] = __builtin_bit_cast(u32x4h, o);
                }
        }

        // ---- conv3 + bn3 + residual + ReLU, one 256-cout slab at a time
        for (int slab = 0; slab < NSLAB; ++slab) {
            f32x16h acc[TM3][2];
#pragma unroll
            for (int a = 0; a < TM3; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.0f;
            float sc[2], sh[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int co = slab * 256 + (wn3 * 2 + b) * 32 + lr;
                sc[b] = p.s3[co];
                sh[b] = p.b3[co];
            }
            for (int kc = 0; kc < KM; ++kc) {
                asm volatile("s_barrier" ::: "memory");
                const char* sb = ring + st * STAGEB;
                if (p.exp_flags & 2048) { next_stage(); continue; }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    f16x8 af[TM3], bf[2];
#pragma unroll
                    for (int a = 0; a < TM3; ++a) af[a] = *(const f16x8*)(mid + (a3off[a] ^ (ks << 5) ^ (kc << 7)));
#pragma unroll
                    for (int b = 0; b < 2; ++b) bf[b] = *(const f16x8*)(sb + (wn3 * 2 + b) * 4096 + (swz ^ (ks << 5)));
#pragma unroll
                    for (int a = 0; a < TM3; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[a], bf[b], acc[a][b], 0, 0, 0);
                }
                pipeline_reads_mfmas<4, TM3 + 2, TM3 * 2, 2>();
                next_stage();
            }
#pragma unroll
            for (int a = 0; a < TM3; ++a) {
                if (p.exp_flags & 4096) break;
#pragma unroll
                for (int g = 0; g < 4; ++g) {  // accumulator registers 4g..4g+3 = tile rows 8g + (0..3) + 4 * (lane >> 5)
                    const int q = a * 4 + g;
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int k = 0; k < 4; ++k) ew[(k + 4 * lh) * CF::EPITCH + b * 32 + lr] = fmaf(acc[a][b][4 * g + k], sc[b], sh[b]);
                    // (no waits around the scratch: a wave's LDS instructions execute in issue order, so these reads see the writes above and the next
                    // pass's writes cannot overtake them; the compiler keeps the program order of accesses that may alias and waits only where a
                    // read's RESULT is used -- passes overlap instead of paying two LDS round trips each)
                    const f32x4h v0 = *(const f32x4h*)(ew + er * CF::EPITCH + ec);
                    const f32x4h v1 = *(const f32x4h*)(ew + er * CF::EPITCH + ec + 4);
                    const unsigned ooff = strip_off(slab, q);
                    const f16x8 rh = __builtin_bit_cast(f16x8, rw[q % RD]);
                    // refill the slot with the pass RD further on (this slab's, or the next slab's first ones; NQ % RD == 0 keeps slots static)
                    if (!DS) {
                        if (q + RD < NQ) rw[q % RD] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, strip_off(slab, q + RD), 0, 0);
                        else if (slab + 1 < NSLAB) rw[q % RD] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, strip_off(slab + 1, q + RD - NQ), 0, 0);
                    }
                    f16x8 o;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        float y = (k < 4 ? v0[k] : v1[k - 4]) + (float)rh[k];
                        o[k] = (half_t)(y > 0.0f ? y : 0.0f);
                    }
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4h, o), rs_out, ooff, 0, 0);
                }
            }
        }
    }
}

template <class CF>
static int launch_btl(BtlK& k, hipStream_t st, int few) {
    k.tiles_x = cdiv(k.W, CF::TW);
    k.tiles_y = cdiv(k.H, CF::TH);
    const int64_t total = (int64_t)k.N * k.tiles_x * k.tiles_y;
    ARG_CHECK(total < (1 << 30), "too many tiles");
    k.total = (int)total;
    LDS_LIMIT_ONCE(CF::LDS, bottleneck_f16_kernel<CF>);
    const int ncu = device_cu_count();
    int64_t slots = few ? 8 : (int64_t)(ncu / 8) * 8;  // one block per CU; a multiple of 8 keeps a block's tiles on its XCD
    if (slots < 8) slots = 8;
    const unsigned grid = (unsigned)(total < slots ? total : slots);
    hipLaunchKernelGGL((bottleneck_f16_kernel<CF>), dim3(grid), dim3((CF::NW + CF::LW) * 64), CF::LDS, st, k);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}

bool bottleneck_f16_supported(int Cin, int Cmid) { return (Cin == 256 && Cmid == 64) || (Cin == 512 && Cmid == 128); }
bool bottleneck_f16_ds_supported(int Cin, int Cmid) { return Cin == 64 && Cmid == 64; }

static int64_t pad128(int c) { return (int64_t)cdiv(c, 128) * 128; }

// x [N][H][W][Cin] fp16 -> out [N][H][W][4 Cmid]; w1 / w2 / w3 (/ wd) are the isegmi_pack_conv_weights_f16 images of the layers.  wd == nullptr: identity
// block (Cin == 4 Cmid, the residual is x); else the first block of res2 with its projection shortcut (Cin = Cmid = 64).
int bottleneck_f16_launch(const isegmi_bottleneck_desc* d, const void* x, const void* w1, const float* s1, const float* b1, const void* w2,
                          const float* s2, const float* b2, const void* w3, const float* s3, const float* b3, const void* wd, const float* sd,
                          const float* bd, void* out, hipStream_t st) {
    ARG_CHECK(d && x && w1 && w2 && w3 && out && s1 && b1 && s2 && b2 && s3 && b3, "null");
    const bool ds = wd != nullptr;
    if (ds) ARG_CHECK(sd && bd && bottleneck_f16_ds_supported(d->Cin, d->Cmid), "fused projection bottleneck: (Cin, Cmid) must be (64, 64) with the shortcut's scale / shift");
    else ARG_CHECK(bottleneck_f16_supported(d->Cin, d->Cmid), "fused bottleneck: (Cin, Cmid) must be (256, 64) or (512, 128)");
    ARG_CHECK(d->N > 0 && d->H > 0 && d->W > 0, "shape");
    ARG_CHECK(x != out, "in-place");
    BtlK k;
    k.x = (const half_t*)x; k.w1 = (const half_t*)w1; k.w2 = (const half_t*)w2; k.w3 = (const half_t*)w3;
    k.s1 = s1; k.b1 = b1; k.s2 = s2; k.b2 = b2; k.s3 = s3; k.b3 = b3;
    k.wd = (const half_t*)wd; k.sd = sd; k.bd = bd;
    k.res = (const half_t*)x; k.out = (half_t*)out;
    k.N = d->N; k.H = d->H; k.W = d->W;
    const int64_t xb = (int64_t)d->N * d->H * d->W * d->Cin * 2, ob = (int64_t)d->N * d->H * d->W * d->Cmid * 4 * 2;
    ARG_CHECK(xb < (1ll << 31) && ob < (1ll << 31), "activation must be < 2 GiB");
    k.x_bytes = (unsigned)xb;
    k.out_bytes = (unsigned)ob;
    k.w1_bytes = (unsigned)(pad128(d->Cmid) * d->Cin * 2);
    k.w2_bytes = (unsigned)(pad128(d->Cmid) * 9 * d->Cmid * 2);
    k.w3_bytes = (unsigned)(pad128(4 * d->Cmid) * d->Cmid * 2);
    k.wd_bytes = ds ? (unsigned)(pad128(4 * d->Cmid) * d->Cin * 2) : 0u;
    k.res_bytes = ds ? 0u : k.out_bytes;
    ARG_CHECK(kExperimentFlags || (d->flags & ~1) == 0, "isegmi_bottleneck_desc.flags: only bit 0 exists in a release build (timing-only experiment bits need -DISEGMI_EXPERIMENT_FLAGS)");
    const int few = d->flags & 1;  // TEST HOOK: 8-block grid, so that small shapes exercise the multi-tile stream
    // TIMING-ONLY experiments (results wrong): the range check drops every x load (2) / residual load (4) / output store (8) / weight load (16)
    if (d->flags & 2) k.x_bytes = 0;
    if (d->flags & 4) k.res_bytes = 0;
    if (d->flags & 8) k.out_bytes = 0;
    if (d->flags & 16) k.w1_bytes = k.w2_bytes = k.w3_bytes = k.wd_bytes = 0;
    k.exp_flags = d->flags & (32 | 64 | 128 | 256 | 512 | 1024 | 2048 | 4096);
    if (ds) return launch_btl<BtlCfg<64, 64, 8, 16, true>>(k, st, few);
    if (d->Cin == 256) return launch_btl<BtlCfg<256, 64, 8, 16>>(k, st, few);
    return launch_btl<BtlCfg<512, 128, 8, 14>>(k, st, few);
}

}  // namespace isegmi

using namespace isegmi;

extern "C" int isegmi_op_bottleneck_f16(const isegmi_bottleneck_desc* d, const void* d_x, const void* d_w1, const float* d_s1, const float* d_b1,
                                        const void* d_w2, const float* d_s2, const float* d_b2, const void* d_w3, const float* d_s3,
                                        const float* d_b3, void* d_out, void* stream) {
    return bottleneck_f16_launch(d, d_x, d_w1, d_s1, d_b1, d_w2, d_s2, d_b2, d_w3, d_s3, d_b3, nullptr, nullptr, nullptr, d_out, (hipStream_t)stream);
}

extern "C" int isegmi_op_bottleneck_ds_f16(const isegmi_bottleneck_desc* d, const void* d_x, const void* d_w1, const float* d_s1, const float* d_b1,
                                           const void* d_w2, const float* d_s2, const float* d_b2, const void* d_w3, const float* d_s3,
                                           const float* d_b3, const void* d_wd, const float* d_sd, const float* d_bd, void* d_out, void* stream) {
    ARG_CHECK(d_wd, "null");
    return bottleneck_f16_launch(d, d_x, d_w1, d_s1, d_b1, d_w2, d_s2, d_b2, d_w3, d_s3, d_b3, d_wd, d_sd, d_bd, d_out, (hipStream_t)stream);
}
