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

// the 16 x 16 x 32 tiles (conv_mfma_f16_m16.hip): tile 40 = row strips 192 x 256 (three B buffers), 41 = row strips 144 x 256, 44 / 47 / 49 = persistent
// 256 x 128 / 192 x 256 / 128 x 256, 46 = persistent 144 x 256 (three-deep ring), 48 = 47 with the UP2X residual walk.  k is filled by conv2d_f16_launch_impl exactly as for the 32 x 32 x 16 tiles.
int conv_f16_m16_launch(int tile, ConvKH& k, hipStream_t st, bool few);

}  // namespace isegmi
