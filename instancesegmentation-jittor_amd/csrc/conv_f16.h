// conv_f16.h -- what the fp16 convolution translation units share (csrc/conv_mfma_f16.hip: the v_mfma_f32_32x32x16_f16 kernels and the launcher;
// csrc/conv_mfma_f16_m16.hip: the backbone tiles on v_mfma_f32_16x16x32_f16): build-time cache-policy knobs, vector types, the kernel argument block.
#pragma once
#include "../../include/isegmi.h"
#include "common.h"
#include "detmath.h"
#include <type_traits>

// Cache policy (the buffer instructions' aux field: 0 default, 2 = nt) of the fp16 kernels' three big streams, settable per build for A/B
// (tools/ab_cache_policy.sh, R101 bs=8, same box, two runs each; profiles/r03_experiments.txt 5):
//   activation (A operand) LDS-DMA loads nt: conv 8.47 -> 8.80 ms per step -- tiles of several Cout bands and the taps of a 3x3 re-read A through L2;
//   residual loads nt: 7.99 -> 7.94 ms (read once, written three layers ago: nothing to keep) -- adopted;
//   output stores nt: no change.
#ifndef CONV_F16_A_AUX
#define CONV_F16_A_AUX 0
#endif
#ifndef CONV_F16_RES_AUX
#define CONV_F16_RES_AUX 2
#endif
#ifndef CONV_F16_OUT_AUX
#define CONV_F16_OUT_AUX 0
#endif
// wave priority (s_setprio 0..3) of the loader waves / the MFMA waves of the loader-wave kernels; -1: left alone
#ifndef CONV_F16_LOADER_PRIO
#define CONV_F16_LOADER_PRIO -1
#endif
#ifndef CONV_F16_MFMA_PRIO
#define CONV_F16_MFMA_PRIO -1
#endif

namespace isegmi {

__device__ __forceinline__ void conv_f16_role_prio(const bool loader) {
    if (loader) { if (CONV_F16_LOADER_PRIO >= 0) __builtin_amdgcn_s_setprio(CONV_F16_LOADER_PRIO < 0 ? 0 : CONV_F16_LOADER_PRIO); }
    else if (CONV_F16_MFMA_PRIO >= 0) __builtin_amdgcn_s_setprio(CONV_F16_MFMA_PRIO < 0 ? 0 : CONV_F16_MFMA_PRIO);
}


typedef _Float16 half_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16h __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4h __attribute__((ext_vector_type(4)));
typedef float f32x4h __attribute__((ext_vector_type(4)));

struct ConvKH {
    const half_t* in;
    const half_t* w;
    const float* scale;
    const float* shift;
    const half_t* res;
    void* out;
    int N, H, W, Cin, Cout, R, S, stride, pad, Ho, Wo, M;
    int nchunks, cin_chunks;
    int64_t wrow;  // halfs per packed cout row
    unsigned in_bytes, out_bytes, res_bytes, w_bytes;
    int act, out_div, contiguous, out_f32, vec_epi;
    int64_t out_img_stride, out_pix_stride;
    int mtiles, ntiles;
#ifdef ISEGMI_STRIP_TRACE
    unsigned* trace;
#endif
    int dbg;  // TIMING-ONLY experiments (tile bit 4096): strip kernel loaders stop after the first two groups
    int res_up2x, rHc, rWc;   // residual = a [N][rHc][rWc][Cout] tensor read through nearest-2x upsampling (FPN top-down merge; tile 38 only)
    // fused 1x1 head on the tile's result (192 x 256 row-strip tile only; see conv_f16_epilogue_head): packed [128][256] fp16 weights, scale / shift,
    // fp32 [M][f_cout] output; f_w == nullptr: the ordinary epilogue
    const half_t* f_w;
    const float* f_scale;
    const float* f_shift;
    float* f_out;
    int f_cout;
    unsigned f_out_bytes;
};

// ---------------------------------------------------------------------------------------------------------------------
// The A (pixels x 64 halfs) and B (couts x 64 halfs) chunk images are filled by
// `buffer_load_dwordx4 ... lds` (no VGPR round trip, no ds_write): one wave-instruction moves 8 rows x 128 B = 1 KiB that
// lands lane-linear in LDS, so rows are unpadded and the bank swizzle is applied on the SOURCE side: LDS slot
// (row, cs) holds the row's 16-B column c = cs ^ ((row >> 1) & 7); readers fetch column c of row r at slot
// c ^ ((r >> 1) & 7), which is conflict-free for ds_read_b128's 16-lane groups.  Padding / M tail / Cout tail come
// from the buffer range check (offset 0x80000000 -> zeros written to LDS).  NSTAGE-deep ring, ONE barrier per chunk:
//     s_waitcnt vmcnt((NSTAGE-2)*pieces) ; s_barrier ; issue chunk t+NSTAGE-1 ; MFMAs on chunk t
// (past the last chunk the issue slot sends all-OOB pieces so the counted wait stays uniform).
// Epilogue: each wave transposes its fp32 strip through LDS and stores / loads the residual 16 B per lane.
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// ---------------------------------------------------------------------------------------------------------------------
// The LOADER role of the persistent loader-wave kernels (conv_f16_persist_kernel in BOTH translation units: the v_mfma_f32_32x32x16_f16 tiles of
// conv_mfma_f16.hip and the v_mfma_f32_16x16x32_f16 tiles of conv_mfma_f16_m16.hip).  Nothing in it depends on the MFMA shape -- chunk images, the LDS-DMA
// pieces, the counted vmcnt and the one-barrier-per-chunk protocol are what the two kernels share -- so it lives here once (VERDICT r5 item 8 / ADVICE:
// two copies of the protocol would drift); the MFMA role (fragment reads, instruction, accumulator staging) stays per translation unit.
//   issue chunks 0..NSTAGE-2;  per chunk t: vmcnt((NSTAGE-2)*PP); barrier #t; issue chunk t+NSTAGE-1 (dead past the last tile);  after a tile's last
//   chunk: barrier E.
// A wave without a piece in a partial last round of A pieces (144 rows = 18 pieces over 4 loaders) sends a DROPPED piece to its own KiB at
// smemg + NSTAGE * STAGEB + ESPARE instead, so that every loader's vmcnt count per chunk stays PP.
typedef const ConvKH __attribute__((address_space(4)))* conv_f16_karg_t;
template <int BM, int BN, int NW, int NL, int NSTAGE, int ESPARE>
__device__ __forceinline__ void conv_f16_persist_loader(conv_f16_karg_t kp0, char* smemg, const int wave, const int lane, const int bid, const int G,
                                                        const int total, const int my_tiles, const int p_ntiles) {
    typedef conv_f16_karg_t karg_t;
    constexpr int PA = BM / 8, PB = BN / 8;
    constexpr int PPA = (PA + NL - 1) / NL, PPB = (PB + NL - 1) / NL;
    constexpr int STAGEB = (BM + BN) * 128;
    constexpr int PP = PPA + PPB;
    constexpr unsigned OOB = 0x80000000u;
    const int q8 = total >> 3, r8g = total & 7;
    auto tile_origin = [&](int v, int& m0, int& n0) {  // v = bid + i * G: same XCD as bid (G is a multiple of 8 or the whole grid)
        const int xcd = v & 7;
        const int logical = (xcd < r8g ? xcd * (q8 + 1) : r8g * (q8 + 1) + (xcd - r8g) * q8) + (v >> 3);
        const int nt = logical % p_ntiles, mt = logical / p_ntiles;
        m0 = mt * BM; n0 = nt * BN;
    };
    karg_t kl = kp0;
    asm volatile("" : "+s"(kl));
    const ConvKH& p = *(const ConvKH*)kl;
    const int lw = wave - NW;
    const int r8 = lane >> 3, cs = lane & 7;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_in0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 0, 0x00020000);
    int hi0[PPA], wi0[PPA], abase[PPA];
    unsigned avoff[PPA], bbase[PPB];
    int kr = 0, ks = 0, kc = 0, in_tile = 0, v = bid;
    unsigned soffa = 0;
    bool live = true;
    auto setup_tile = [&](int vv) {  // per-lane row bases of tile vv; the only place with integer divisions
        int m0, n0;
        tile_origin(vv, m0, n0);
#pragma unroll
        for (int j = 0; j < PPA; ++j) {
            const int row = (lw + j * NL) * 8 + r8;
            const int c = cs ^ ((row >> 1) & 7);
            const int m = m0 + row;
            if (m < p.M) {
                const int hw = p.Ho * p.Wo;
                const int n = m / hw, rem = m - n * hw;
                const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
                hi0[j] = ho * p.stride - p.pad;
                wi0[j] = wo * p.stride - p.pad;
                abase[j] = (((n * p.H + hi0[j]) * p.W + wi0[j]) * p.Cin) * 2 + c * 16;
            } else {
                hi0[j] = -(1 << 28); wi0[j] = 0; abase[j] = 0;
            }
            const bool ok = (unsigned)hi0[j] < (unsigned)p.H && (unsigned)wi0[j] < (unsigned)p.W;
            avoff[j] = ok ? (unsigned)abase[j] : OOB;
        }
#pragma unroll
        for (int j = 0; j < PPB; ++j) {
            const int row = (lw + j * NL) * 8 + r8;
            const int c = cs ^ ((row >> 1) & 7);
            bbase[j] = (unsigned)(n0 + row) * (unsigned)(p.wrow * 2) + (unsigned)(c * 16);
        }
        kr = 0; ks = 0; kc = 0; in_tile = 0; soffa = 0;
    };
    setup_tile(v);
    auto issue_chunk = [&](int stage) {  // all of this wave's pieces of the next chunk of the stream, then advance the stream
        char* sA = smemg + stage * STAGEB;
        const unsigned soffb = (unsigned)in_tile * 128u;
#pragma unroll
        for (int i = 0; i < PPA; ++i) {
            // a wave without a piece in the partial last round (144 rows = 18 pieces over 4 loaders) sends a DROPPED one (zero-length descriptor: zeros
            // land in the wave's own dummy KiB behind the ring) so that every wave's vmcnt count per chunk stays PP (wave-uniform branch)
            const bool dummy = PA % NL != 0 && lw + i * NL >= PA;
            // (named operands: hipcc 7.2 silently drops the host stub of the kernel when this builtin takes expressions)
            const __amdgpu_buffer_rsrc_t rs = (live && !dummy) ? rs_in : rs_in0;
            const unsigned voff = avoff[i];
            char* dstp = dummy ? smemg + NSTAGE * STAGEB + ESPARE + lw * 1024 : sA + (lw + i * NL) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)dstp, 16, voff, soffa, 0, CONV_F16_A_AUX);
        }
        static_assert(PB % NL == 0, "whole B piece rounds");
#pragma unroll
        for (int j = 0; j < PPB; ++j) {
            if (PB % NL != 0 && lw + j * NL >= PB) continue;
            const __amdgpu_buffer_rsrc_t rs = live ? rs_w : rs_w0;
            const unsigned voff = bbase[j];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(sA + BM * 128 + (lw + j * NL) * 1024), 16, voff, soffb, 0, 0);
        }
        if (!live) return;
        soffa += 128u;
        if (++in_tile == p.nchunks) {  // uniform: the stream moves on to this block's next tile
            v += G;
            live = v < total;
            if (live) setup_tile(v);
            return;
        }
        if (++kc == p.cin_chunks) {  // uniform: next tap -- the only per-lane work inside a tile
            kc = 0;
            soffa = 0;
            if (++ks == p.S) { ks = 0; ++kr; }
            int tr = __builtin_amdgcn_readfirstlane(kr), ts = __builtin_amdgcn_readfirstlane(ks);
            asm volatile("" : "+s"(tr), "+s"(ts));  // keeps the tap change behind its branch
            const int delta = ((tr * p.W + ts) * p.Cin) * 2;
#pragma unroll
            for (int j = 0; j < PPA; ++j) {
                const bool ok = (unsigned)(hi0[j] + tr) < (unsigned)p.H && (unsigned)(wi0[j] + ts) < (unsigned)p.W;
                avoff[j] = ok ? (unsigned)(abase[j] + delta) : OOB;
            }
        }
    };
#pragma unroll
    for (int s = 0; s < NSTAGE - 1; ++s) issue_chunk(s);
    int wr = NSTAGE - 1;
    for (int i = 0; i < my_tiles; ++i) {
        for (int t = 0; t < p.nchunks; ++t) {
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NSTAGE - 2) * PP) : "memory");
            issue_chunk(wr);
            wr = wr + 1 == NSTAGE ? 0 : wr + 1;
        }
        asm volatile("s_barrier" ::: "memory");  // E: see the MFMA waves
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // trailing dead pieces have landed
}

// the 16 x 16 x 32 tiles (conv_mfma_f16_m16.hip): tile 40 = row strips 192 x 256 (three B buffers), 41 = row strips 144 x 256, 44 / 47 / 49 = persistent
// 256 x 128 / 192 x 256 / 128 x 256, 46 = persistent 144 x 256 (three-deep ring), 48 = 47 with the UP2X residual walk.  k is filled by conv2d_f16_launch_impl exactly as for the 32 x 32 x 16 tiles.
int conv_f16_m16_launch(int tile, ConvKH& k, hipStream_t st, bool few);

}  // namespace isegmi
