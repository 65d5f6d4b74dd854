// stem_pool_f16.hip -- the ResNet stem of the fp16 path as ONE kernel (gfx950, v_mfma_f32_32x32x16_f16):
//     out = maxpool3x3/2/1( relu( bn( conv7x7/2/3(image) ) ) )
// SURVEY 8a M2 `StemWithFixedBatchNorm` + `max_pool2d` under BASELINE configs[4].  As two launches (conv_f16_glds_kernel<64,64,...,STEM> +
// maxpool_f16_c8_kernel) the stem of an R101 bs=8 step writes a 275 MB fp16 tensor and reads it back (197 + 85 us, 1.8 TB/s), and the conv
// stages a 512-byte im2col row per output pixel through the CU's fill path (1.1 GB for 69 MB of distinct input).  Here
//   * a block owns a STRIP of 30 pooled columns (61 conv columns, 128 haloed input pixels) and walks down it in GROUPS of four conv rows;
//   * input rows go ONCE into an LDS ring (one 1-KiB LDS-DMA piece per input row: lane j brings pixels 2j, 2j+1 = 16 B), and the MFMA
//     operand of conv pixel c, filter row r, taps t..t+1 is read straight from the ring at 16 (c + t/2) bytes -- no im2col image at all;
//   * the 64 x 256 packed weights live in REGISTERS (128 VGPRs per wave, loaded once per block): the convolution is computed TRANSPOSED
//     (D[cout][pixel], the weight fragment is the MFMA's A operand; the MFMA is bitwise symmetric under that swap -- tools/microbench/mfma_sym.hip),
//     so a lane holds four consecutive channels of one pixel per register group and the BN + ReLU + fp16 result goes to an LDS ring of conv
//     rows with 8-byte stores;
//   * the 3x3/2 max-pool reads that ring (two pooled rows per group, two conv rows carried over) and stores 16 B per lane: the conv output
//     never leaves the CU.
// Bit-identical to the two launches it replaces: same packed weights and K order (16 k-steps of (row, tap, channel)), the same fp32
// epilogue expression, fp16 rounding at the same place, and a max of fp16 values is one of them (tests/test_stem_pool_f16_gpu.py).
// Work unit = (image, row segment of 2G-1 pooled rows, strip); G groups cost 4G conv rows for 4G-2 useful ones, the host picks G so that the
// units fill the CUs a whole number of times.  Per group: wait for the group's input rows + barrier, prefetch two input rows per wave two
// groups ahead, 2 x 32 MFMAs per wave (one conv row x 32 columns x all 64 couts at a time, then its epilogue), barrier, pool, store.
#include "../../include/isegmi.h"
#include "common.h"

namespace isegmi {

typedef _Float16 half_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16h __attribute__((ext_vector_type(16)));
typedef float f32x4h __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4h __attribute__((ext_vector_type(4)));
typedef short i16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct StemPoolK {
    const half_t* in;    // haloed image [N][Hh][Wh][4] (pad_c3_to_f16_halo: 3 zero pixels on every side, Wh even)
    const half_t* w;     // packed stem weights [64][8 rows][8 taps][4 ch] (isegmi_pack_conv_weights_f16)
    const float* scale;
    const float* shift;
    half_t* out;         // [N][Hp][Wp][64]
    int N, Hh, Wh, Hc, Wc, Hp, Wp;
    int G, strips, segs, total;
    int dbg;             // TIMING-ONLY experiments (flags bits 4 / 8 / 16 / 32: no pooling / no epilogue / no MFMAs / no input loads)
    unsigned in_bytes, out_bytes;
};

constexpr int SP_PW = 30;                 // pooled columns per strip
constexpr int SP_RING_ROWS = 24;          // input ring: rows 8g .. 8g + 23 of the unit
constexpr int SP_CROWS = 6;               // conv-row ring: rows 4g - 2 .. 4g + 3
constexpr int SP_RING = 0;                // 24 KiB
constexpr int SP_CONV = 24 * 1024 + 256;  // 6 rows x 64 columns x 64 channels fp16 = 48 KiB (the 256 B in front: taps of the wasted columns 61..63 of ring row 23 read there)
constexpr int SP_SCALE = SP_CONV + SP_CROWS * 8192;
constexpr int SP_LDS = SP_SCALE + 512;    // 73.5 KiB: two blocks per CU

// Two blocks of four waves per CU rather than one of eight: a block's phases (MFMA / epilogue / pool) are barrier-separated, and only the MFMA phase uses
// the matrix pipe; two independent blocks drift apart, so one's vector work runs under the other's MFMAs.
__global__ __launch_bounds__(256, 2) void stem_pool_f16_kernel(const StemPoolK p) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    constexpr unsigned OOB = 0x80000000u;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rp = wave >> 1, mt = wave & 1;   // conv row pair of the group (rows 2 rp, 2 rp + 1), 32-column half of the strip
    const int p32 = lane & 31, kg = lane >> 5;

    // weights: A operand of the transposed product -- lane (cout = ct * 32 + p32, k-half kg) holds k = 16 ks + 8 kg .. + 7 of its cout
    f16x8 wf[2][16];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) wf[ct][ks] = *(const f16x8*)(p.w + (ct * 32 + p32) * 256 + ks * 16 + kg * 8);
    if (tid < 64) {
        ((float*)(smem + SP_SCALE))[tid] = p.scale[tid];
        ((float*)(smem + SP_SCALE + 256))[tid] = p.shift[tid];
    }
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)p.out, 0, p.out_bytes, 0x00020000);
    const int G = p.G, R = 2 * G - 1;      // groups per unit, pooled rows per unit
    const int kmax = 8 * G + 4;            // last input row (unit-relative) any conv row of the unit reads with a non-zero weight

    for (int unit = blockIdx.x; unit < p.total; unit += gridDim.x) {
        const int strip = unit % p.strips;
        const int t_ = unit / p.strips;
        const int seg = t_ % p.segs, n = t_ / p.segs;
        const int ph0 = seg * R, pw0 = strip * SP_PW;
        const int crow0 = 2 * ph0 - 1, c0 = 2 * pw0 - 1;   // conv row / column of the unit's (0, 0)
        const int irow0 = 2 * crow0;                       // haloed input row of ring row 0
        const int px = 2 * c0 + 2 * lane;                  // haloed input pixel of this lane's 16 B in every ring row
        const bool colok = (unsigned)px < (unsigned)p.Wh;
        const int nbase = n * p.Hh;
        auto issue_row = [&](int k) {  // unit-relative input row k (wave-uniform) -> ring slot k % 24
            const int irow = irow0 + k;
            const bool ok = !(p.dbg & 32) && colok && (unsigned)irow < (unsigned)p.Hh && k <= kmax;
            const unsigned voff = ok ? (unsigned)(((nbase + irow) * p.Wh + px) * 8) : OOB;
            const int slot = k % SP_RING_ROWS;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_ptr_t)(smem + SP_RING + slot * 1024), 16, voff, 0, 0, 0);
        };
        // every wave is done with the previous unit's rings, and nothing of it is still on its way into them
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (int j = 0; j < 4; ++j) issue_row(4 * wave + j);

        for (int g = 0; g < G; ++g) {
            // ---- the group's input rows (<= 8g + 12) are down; everybody is past the previous group's pooling
            if (g == 0) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");   // all but this wave's two pooled-row stores of the previous group
            issue_row(8 * g + 16 + 2 * wave);                                     // rows 8g + 16 .. 8g + 23: the slots group g - 1 read
            issue_row(8 * g + 17 + 2 * wave);

            // ---- 4 conv rows x 64 columns x 64 couts: this wave's rows 4g + 2 rp, + 1, columns 32 mt .. + 31
            const int col = mt * 32 + p32;
            const int lane_off = SP_RING + (col + kg) * 16;
            const int sw = (col >> 1) & 7;
#pragma unroll 1
            for (int j = (p.dbg & 16) ? 2 : 0; j < 2; ++j) {
                const int lrow = 2 * rp + j;
                f32x16h acc[2];
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[ct][e] = 0.0f;
                const int rbase = (8 * g) % SP_RING_ROWS + 2 * lrow;              // ring slot of filter row 0 (before wrapping)
                auto frag = [&](int ks) {  // pixel fragment of k-step ks: filter row ks / 2, taps 4 (ks & 1) + 2 kg, + 1
                    int rr = rbase + (ks >> 1);
                    rr = rr >= SP_RING_ROWS ? rr - SP_RING_ROWS : rr;
                    return *(const f16x8*)(smem + lane_off + rr * 1024 + (ks & 1) * 32);
                };
                f16x8 bf[4];
                bf[0] = frag(0); bf[1] = frag(1); bf[2] = frag(2);
#pragma unroll
                for (int ks = 0; ks < 16; ++ks) {
                    if (ks + 3 < 16) bf[(ks + 3) & 3] = frag(ks + 3);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[0][ks], bf[ks & 3], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[1][ks], bf[ks & 3], acc[1], 0, 0, 0);
                }
                if (p.dbg & 8) { asm volatile("" :: "v"(acc[0]), "v"(acc[1])); continue; }
                // epilogue: y = fmaf(acc, scale, shift); ReLU; fp16 -> conv-row ring slot (4g + lrow) % 6, column col, 8 B per register group
                char* dst = smem + SP_CONV + ((4 * g + lrow) % SP_CROWS) * 8192 + col * 128 + kg * 8;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int c = ct * 32 + 8 * q + 4 * kg;   // first of this lane's four consecutive couts
                        const f32x4h sc = *(const f32x4h*)(smem + SP_SCALE + c * 4), sh = *(const f32x4h*)(smem + SP_SCALE + 256 + c * 4);
                        f16x4 o;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float y = fmaf(acc[ct][q * 4 + i], sc[i], sh[i]);
                            y = y > 0.0f ? y : 0.0f;
                            o[i] = (half_t)y;
                        }
                        *(f16x4*)(dst + (((ct * 4 + q) ^ sw) << 4)) = o;
                    }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

            // ---- pooled rows 2g - 1 and 2g of the unit: conv rows 4g - 2 .. 4g + 2 (two of them carried over from the previous group).  480 items of
            // (pooled row, pooled column, 8 channels), two per thread.  Window taps outside the conv image are CLAMPED onto it: the duplicate cannot change a max.
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int item = tid + pass * 256;
                const int pr = item >= 240 ? 1 : 0, prem = item - pr * 240;
                const int pc = prem >> 3, ch8 = prem & 7;
                const int phr = 2 * g - 1 + pr;
                const int ph = ph0 + phr, pw = pw0 + pc;
                const bool valid = item < 480 && phr >= 0 && phr < R && ph < p.Hp && pw < p.Wp;
                const u32x4h ninf = {0xFC00FC00u, 0xFC00FC00u, 0xFC00FC00u, 0xFC00FC00u};
                i16x8 m = __builtin_bit_cast(i16x8, ninf);   // the max is taken on the bit patterns as int16: monotone on values >= +0, and -inf is negative
                int coff[3];
#pragma unroll
                for (int dc = 0; dc < 3; ++dc) {
                    int ca = c0 + 2 * pc + dc;
                    ca = ca < 0 ? 0 : ca; ca = ca > p.Wc - 1 ? p.Wc - 1 : ca;
                    const int ccol = (ca - c0) & 63;
                    coff[dc] = ccol * 128 + ((ch8 ^ ((ccol >> 1) & 7)) << 4);
                }
#pragma unroll
                for (int dr = 0; dr < 3; ++dr) {
                    if (p.dbg & 4) break;
                    int ra = crow0 + 2 * phr + dr;
                    ra = ra < 0 ? 0 : ra; ra = ra > p.Hc - 1 ? p.Hc - 1 : ra;
                    int crel = ra - crow0;
                    crel = crel < 0 ? 0 : crel;
                    const char* rowp = smem + SP_CONV + (crel % SP_CROWS) * 8192;
#pragma unroll
                    for (int dc = 0; dc < 3; ++dc) m = __builtin_elementwise_max(m, *(const i16x8*)(rowp + coff[dc]));
                }
                const unsigned ooff = valid ? (unsigned)((((n * p.Hp + ph) * p.Wp + pw) * 64 + ch8 * 8) * 2) : OOB;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4h, m), rs_out, ooff, 0, 0);   // exactly two stores per wave and group (the counted wait above)
            }
        }
    }
}

// units that fill the CUs a whole number of times at the least redundant conv rows: cost ~ rounds x (G + fixed cost of a unit in groups)
static int stem_pool_groups(int N, int Hp, int strips, int ncu) {
    int best = 2; double bc = 1e30;
    for (int G = 2; G <= 32; ++G) {
        const int64_t units = (int64_t)N * strips * ((Hp + 2 * G - 2) / (2 * G - 1));
        const double c = (double)((units + ncu - 1) / ncu) * (G + 1.5);
        if (c < bc) { bc = c; best = G; }
    }
    return best;
}

bool stem_pool_f16_supported(int Cout) { return Cout == 64; }

// `halo` = [N][H + 6][(W + 7) & ~1][4] fp16 (pad_c3_to_f16_halo), w = the packed fp16 stem weights, out = [N][Hp][Wp][64] fp16 with
// Hc = (H - 1) / 2 + 1, Hp = (Hc - 1) / 2 + 1 (7x7/2/3 then 3x3/2/1)
int stem_pool_f16_launch(int N, int H, int W, const void* halo, const void* w, const float* scale, const float* shift, void* out, int flags,
                         hipStream_t st) {
    ARG_CHECK(halo && w && scale && shift && out, "null");
    ARG_CHECK(N > 0 && H > 0 && W > 0, "shape");
    ARG_CHECK(kExperimentFlags || (flags & ~3) == 0, "stem flags: only the test hooks (bits 0, 1) exist in a release build (timing-only experiment bits need -DISEGMI_EXPERIMENT_FLAGS)");
    StemPoolK k;
    k.in = (const half_t*)halo; k.w = (const half_t*)w; k.scale = scale; k.shift = shift; k.out = (half_t*)out;
    k.N = N; k.Hh = H + 6; k.Wh = (W + 7) & ~1;
    k.Hc = (H + 6 - 7) / 2 + 1; k.Wc = (W + 6 - 7) / 2 + 1;
    k.Hp = (k.Hc + 2 - 3) / 2 + 1; k.Wp = (k.Wc + 2 - 3) / 2 + 1;
    const int64_t in_bytes = (int64_t)N * k.Hh * k.Wh * 8, out_bytes = (int64_t)N * k.Hp * k.Wp * 128;
    ARG_CHECK(in_bytes < (1ll << 31) && out_bytes < (1ll << 31), "stem tensors must be < 2 GiB");
    k.in_bytes = (unsigned)in_bytes; k.out_bytes = (unsigned)out_bytes;
    k.strips = (k.Wp + SP_PW - 1) / SP_PW;
    const int ncu = 512;   // block slots: two blocks per CU
    k.G = (flags & 2) ? 2 : stem_pool_groups(N, k.Hp, k.strips, ncu);   // flags bit 1 (test hook): the shortest units (every seam between units)
    k.segs = (k.Hp + 2 * k.G - 2) / (2 * k.G - 1);
    const int64_t total = (int64_t)N * k.strips * k.segs;
    ARG_CHECK(total < (1ll << 31), "too many units");
    k.total = (int)total;
    k.dbg = (flags >> 2) & 15 ? (flags & 60) : 0;
    const int grid = (flags & 1) ? (k.total < 8 ? k.total : 8) : (k.total < ncu ? k.total : ncu);   // flags bit 0 (test hook): blocks that walk many units
    LDS_LIMIT_ONCE(SP_LDS, stem_pool_f16_kernel);
    hipLaunchKernelGGL(stem_pool_f16_kernel, dim3((unsigned)grid), dim3(256), SP_LDS, st, k);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}

}  // namespace isegmi

using namespace isegmi;

extern "C" int isegmi_op_stem_pool_f16(int N, int H, int W, const void* d_halo, const void* d_w, const float* d_scale, const float* d_shift,
                                       void* d_out, int flags, void* stream) {
    return stem_pool_f16_launch(N, H, W, d_halo, d_w, d_scale, d_shift, d_out, flags, (hipStream_t)stream);
}
