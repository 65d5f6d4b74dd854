// conv_mfma_f16_m16.hip -- the fp16 backbone tiles on v_mfma_f32_16x16x32_f16 (round 5): the 192 x 256 row-strip kernel (3x3 / 1 / 1) and the
// persistent loader-wave kernel of csrc/conv_mfma_f16.hip with the SAME loaders, LDS chunk images, barrier protocol, tile walk and output tile per wave
// (64 x 64) -- only the MFMA waves' fragment reads, the instruction and the accumulator layout (hence the epilogues' staging) differ.  Its own translation
// unit so that the 32 x 32 x 16 kernels' code generation is untouched (co-compiled template variants share register-allocation context: the first attempt,
// one template with a shape parameter, spilled in the 32 x 32 x 16 instantiations that had not spilled before).  What the two forms share lives ONCE
// (round 6): the persistent kernel's loader role in conv_f16.h (conv_f16_persist_loader), the strip kernel's geometry / loader state / loader loop in
// conv_f16_strip_state.inc and conv_f16_strip_loader_loop.inc; what remains here is what the shape decides -- accumulator layout, fragment reads, the chunk
// of MFMAs, the epilogues' staging.  (The template parameter MS = 1 only marks the form in the kernels' mangled names.)
#include "conv_f16.h"

namespace isegmi {

// ---- the two f16 MFMA shapes (round 5).  A wave tile of TM*32 rows x TN*32 columns is held either as TM x TN blocks of 32 x 32 (sixteen registers each,
// v_mfma_f32_32x32x16_f16: lane (lr = lane & 31, lh = lane >> 5), register e <-> row (e & 3) + 8 (e >> 2) + 4 lh, column lr) or as 2TM x 2TN blocks of
// 16 x 16 (four registers each, v_mfma_f32_16x16x32_f16: lane (l15 = lane & 15, lq = lane >> 4), register e <-> row 4 lq + e, column l15).  Both read their
// operand fragments -- 8 consecutive k per lane -- from the SAME LDS chunk images: row r's 16-B column c sits at slot c ^ ((r >> 1) & 7); the 32 x 32 form
// reads column 2 s + lh of row lr in 16-deep step s, the 16 x 16 form column 4 s + lq of row l15 in 32-deep step s, conflict-free for ds_read_b128 either
// way (every 16-lane service group meets eight distinct (r >> 1) & 7 values x two row parities).  The chip holds a higher clock on the 16 x 16 x 32 form in
// power-bound loops (MI355X_MICROARCH.md "DVFS give-back" 7; tools/microbench/mfma_shape.hip: 1.12-1.13x on random data at equal cycles).  The 32-term sum
// of one 16 x 16 x 32 instruction is bit for bit what two chained 32 x 32 x 16 instructions give (same microbenchmark: 0 of 204 800 elements differ), and the
// kernels here walk K exactly as their twins do: results do not depend on the shape (tests/test_conv_f16_gpu.py::test_mfma_shape_does_not_change_results).
// In this file a wave tile is NA x NC blocks of 16 x 16: WR = 16 NA rows (48 or 64), 16 NC = 32 TN columns.
template <class T> struct acc_traits;
template <int NA_, int NC_> struct acc_traits<f32x4h[NA_][NC_]> { static constexpr int TM = (NA_ + 1) / 2, TN = NC_ / 2, MS = 1, SR = 16, NA = NA_, NC = NC_, WR = 16 * NA_; };

// block row i (rows [16 i, 16 i + 16) of the wave tile) -> ew[row in strip][column], y = fmaf(acc, scale, shift)
template <int NA, int NC>
__device__ __forceinline__ void epi_stage16(const ConvKH& p, f32x4h (&acc)[NA][NC], int i, float* ew, int lane, int wn, int n0) {
    constexpr int PITCH = NC * 16 + 4;
    const int l15 = lane & 15, lq = lane >> 4;
#pragma unroll
    for (int jb = 0; jb < NC; ++jb) {
        const int co = n0 + wn * NC * 16 + jb * 16 + l15;
        const bool cok = co < p.Cout;
        const float sc = (cok && p.scale) ? p.scale[co] : 1.0f;
        const float sh = (cok && p.shift) ? p.shift[co] : 0.0f;
#pragma unroll
        for (int e = 0; e < 4; ++e) ew[(4 * lq + e) * PITCH + jb * 16 + l15] = fmaf(acc[i][jb][e], sc, sh);
    }
}

// Shared epilogue (fp32 math): y = fmaf(acc, scale, shift) + residual -> act -> fp16 (or fp32) NHWC store.  Vector path: each
// wave transposes its 32-row fp32 strips through a private LDS region (all staging LDS is free by now) so that residual
// loads and output stores are 16 B per lane; strided destinations / Cout % 8 != 0 take the per-element path.
// vector path of conv_f16_epilogue (32-row strips through the wave's private LDS region).  RES: a residual is added -- its 16-B loads run
// a rolling window of two passes ahead of their use (see epi8_prefetch: a load awaited on the spot also drains the previous pass's stores);
// without a residual no load is issued at all (round 2 sent a dropped out-of-range load per pass and waited for it).
template <bool RES, class ACC>
__device__ __forceinline__ void conv_f16_epilogue_vec(const ConvKH& p, ACC& acc, char* smemg, int wave, int lane, int wm, int wn, int m0, int n0) {
    constexpr int TN = acc_traits<ACC>::TN, NA = acc_traits<ACC>::NA, WR = acc_traits<ACC>::WR;
    static_assert(acc_traits<ACC>::MS == 1, "this file holds the 16 x 16 x 32 kernels");
    constexpr unsigned OOB = 0x80000000u;
    constexpr int PITCH = TN * 32 + 4;
    constexpr int LPR = TN * 4, RPP = 64 / LPR, NPASS = 16 / RPP, NQ = NA * NPASS, D = 2 < NQ ? 2 : NQ;   // 16-row strips: one block row each
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)(RES ? (const void*)p.res : (const void*)p.out), 0, RES ? p.res_bytes : 0u, 0x00020000);
    const unsigned esz = p.out_f32 ? 4u : 2u;
    float* ew = (float*)smemg + wave * 16 * PITCH;
    const int er = lane / LPR, ec = (lane % LPR) * 8;
    const int co8 = n0 + wn * TN * 32 + ec;
    const bool cok8 = co8 < p.Cout;
    // pass q covers tile rows RPP q ..: a lane's offsets advance by one stride per pass; rows past M are behind the descriptors' ranges
    // (contiguous destinations; strided ones take the general arithmetic)
    const unsigned rstep = (unsigned)p.Cout * (2u * RPP), ostep = (unsigned)p.out_pix_stride * esz * RPP;
    unsigned rnext = cok8 ? ((unsigned)(m0 + wm * WR + er) * (unsigned)p.Cout + (unsigned)co8) * 2u : OOB;
    unsigned onext = cok8 ? ((unsigned)(m0 + wm * WR + er) * (unsigned)p.out_pix_stride + (unsigned)co8) * esz : OOB;
    u32x4h rw[D];
    if (RES) {
#pragma unroll
        for (int q = 0; q < D; ++q) { rw[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, rnext, 0, CONV_F16_RES_AUX); rnext += rstep; }
    }
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        epi_stage16(p, acc, a, ew, lane, wn, n0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int q = a * NPASS + ps;
            const int rr = ps * RPP + er;
            const f32x4h v0 = *(const f32x4h*)(ew + rr * PITCH + ec);
            const f32x4h v1 = *(const f32x4h*)(ew + rr * PITCH + ec + 4);
            unsigned ooff;
            if (p.contiguous) { ooff = onext; onext += ostep; }
            else {
                const int m = m0 + wm * WR + q * RPP + er;
                const int ni = m / p.out_div, pi = m - ni * p.out_div;
                ooff = (m < p.M && cok8) ? (unsigned)(((int64_t)ni * p.out_img_stride + (int64_t)pi * p.out_pix_stride + co8) * esz) : OOB;
            }
            float y[8];
            if (RES) {
                const f16x8 rh = __builtin_bit_cast(f16x8, rw[q % D]);
#pragma unroll
                for (int i = 0; i < 8; ++i) y[i] = (i < 4 ? v0[i] : v1[i - 4]) + (float)rh[i];
                if (q + D < NQ) { rw[q % D] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, rnext, 0, CONV_F16_RES_AUX); rnext += rstep; }
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) y[i] = i < 4 ? v0[i] : v1[i - 4];
            }
            if (p.act == 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) y[i] = y[i] > 0.0f ? y[i] : 0.0f;
            }
            if (p.out_f32) {
                u32x4h o0, o1;
#pragma unroll
                for (int i = 0; i < 4; ++i) { o0[i] = __builtin_bit_cast(unsigned, y[i]); o1[i] = __builtin_bit_cast(unsigned, y[i + 4]); }
                __builtin_amdgcn_raw_buffer_store_b128(o0, rs_out, ooff, 0, CONV_F16_OUT_AUX);
                __builtin_amdgcn_raw_buffer_store_b128(o1, rs_out, ooff >= OOB ? OOB : ooff + 16u, 0, CONV_F16_OUT_AUX);
            } else {
                f16x8 o;
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = (half_t)y[i];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4h, o), rs_out, ooff, 0, CONV_F16_OUT_AUX);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// per-element path (strided destinations with Cout % 8 != 0: the RPN / prediction heads): y = fmaf(acc, scale, shift) + residual -> act; one accumulator
// register at a time -- ROWB x COLB blocks of NE registers, register e of a lane at (row_of(e), its column)
template <class ACC>
__device__ __forceinline__ void conv_f16_epilogue_elem(const ConvKH& p, ACC& acc, int lane, int wm, int wn, int m0, int n0) {
    constexpr int TN = acc_traits<ACC>::TN, MS = acc_traits<ACC>::MS, NA = acc_traits<ACC>::NA, NC = acc_traits<ACC>::NC, WR = acc_traits<ACC>::WR;
    constexpr int RB = MS ? 16 : 32, NE = MS ? 4 : 16;
    constexpr unsigned OOB = 0x80000000u;
    const int lc = MS ? (lane & 15) : (lane & 31);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res ? (const void*)p.res : (const void*)p.out), 0, p.res ? p.res_bytes : 0u, 0x00020000);
    const unsigned esz = p.out_f32 ? 4u : 2u;
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        unsigned rowoff[NE], resoff[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int rin = MS ? 4 * (lane >> 4) + e : (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
            const int m = m0 + wm * WR + a * RB + rin;
            resoff[e] = m < p.M ? (unsigned)m * (unsigned)p.Cout * 2u : OOB;
            if (p.contiguous) rowoff[e] = m < p.M ? (unsigned)m * (unsigned)p.out_pix_stride * esz : OOB;
            else {
                const int ni = m / p.out_div, pi = m - ni * p.out_div;
                rowoff[e] = m < p.M ? (unsigned)(((int64_t)ni * p.out_img_stride + (int64_t)pi * p.out_pix_stride) * esz) : OOB;
            }
        }
#pragma unroll
        for (int b = 0; b < NC; ++b) {
            const int co = n0 + wn * TN * 32 + b * RB + lc;
            const bool cok = co < p.Cout;
            const float sc = (cok && p.scale) ? p.scale[co] : 1.0f;
            const float sh = (cok && p.shift) ? p.shift[co] : 0.0f;
            const unsigned cooff = cok ? (unsigned)co : OOB;
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const unsigned short hb = __builtin_amdgcn_raw_buffer_load_b16(rs_res, (resoff[e] | cooff) >= OOB ? OOB : resoff[e] + cooff * 2u, 0, 0);
                float y = fmaf(acc[a][b][e], sc, sh);
                y = y + (float)__builtin_bit_cast(half_t, hb);
                y = p.act == 1 ? (y > 0.0f ? y : 0.0f) : y;
                const unsigned off = (rowoff[e] | cooff) >= OOB ? OOB : rowoff[e] + cooff * esz;
                if (p.out_f32) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y), rs_out, off, 0, 0);
                else __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (half_t)y), rs_out, off, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <class ACC>
__device__ __forceinline__ void conv_f16_epilogue(const ConvKH& p, ACC& acc, char* smemg, int wave, int lane, int wm, int wn, int m0, int n0) {
    if (p.vec_epi) {  // uniform
        if (p.res) conv_f16_epilogue_vec<true>(p, acc, smemg, wave, lane, wm, wn, m0, n0);
        else conv_f16_epilogue_vec<false>(p, acc, smemg, wave, lane, wm, wn, m0, n0);
        return;
    }
    conv_f16_epilogue_elem(p, acc, lane, wm, wn, m0, n0);
}

// ---------------------------------------------------------------------------------------------------------------------
// PERSISTENT variant of the loader-wave kernel (round 2).  The one-tile-per-block kernel above runs a tile as
// [fill the ring: one memory latency] -> [chunks] -> [epilogue: a residual round trip + the stores], nothing of it overlapped
// with anything at one block per CU; on the short-K layers (R101 res4's 1x1s: four to sixteen chunks, 11-17 % MfmaUtil) those
// two latencies are most of a tile's time.  Here a block walks tiles bid, bid + grid, ... and its LOADER waves never stop:
// the chunk stream of the next tile follows the last chunk of this one through the same ring and the same one-barrier-per-
// chunk protocol, so while the MFMA waves run a tile's epilogue the ring already holds the first NSTAGE-1 chunks of the next.
//   loader:  issue chunks 0..NSTAGE-2;  per chunk t: vmcnt((NSTAGE-2)*PP); barrier #t; issue chunk t+NSTAGE-1 (dead past the
//            last tile);  after a tile's last chunk: barrier E.
//   MFMA:    per chunk t: barrier #t; fragments + MFMAs from stage t % NSTAGE;  after a tile's last chunk: barrier E; epilogue.
// Barrier E tells every MFMA wave that all of them are done reading the last chunk's stage: that stage is the epilogue's
// scratch (the loader refills it only after barrier #t+1, which needs the MFMA waves), so no LDS is set aside for it.  The
// epilogue transposes 8 rows at a time (2.2 KB per wave).  Tile order: virtual block v = bid + i * grid keeps v & 7 = bid & 7,
// so the XCD-aware remap of the one-tile kernel applies to v unchanged.
// Round 3: the residual of a tile is REQUESTED BEFORE its K loop and consumed strip by strip behind a rolling window.  Before, every
// 8-row strip issued its residual load and waited for it on the spot -- and on gfx9 a wave's stores count in the same vmcnt as its loads, so
// the wait also drained the previous strip's stores: eight exposed memory round trips per 64x64 wave tile, on layers whose K loop is four
// 16-MFMA chunks (R101 res4 conv3: 14.7 % MfmaUtil at 3.9 TB/s, on neither roof).  Now the first EPI_D strips' residuals are in flight
// during the K loop (the MFMA waves issue no other vector-memory instruction there), strip q's slot is refilled with strip q + EPI_D's
// right after use, and the counted waits the compiler derives from program order never drain the stores of the strip just written.
// Offsets: a lane's residual / output offset is ONE add per strip -- lane base (its row inside the strip, its 8 channels; the out-of-range
// constant for a column past Cout) plus a wave-uniform row term; rows past M fall behind descriptors cut at M rows (contiguous
// destinations; strided ones keep the general arithmetic).
// One 64-deep chunk on v_mfma_f32_16x16x32_f16: NA x NC blocks of 16 x 16, two 32-deep steps.  oa[i] / ob + j * 2048 = byte offsets (from sa / sb) of the
// lane's step-0 fragment of block i / j (row l15 of the block, 16-B column lq, swizzled); oa1 / ob1 = the same for step 1: the offset ^ 64 (every other
// term is a multiple of 128).  Registers: the wave tile's 64 accumulators (the kernels run four waves per SIMD: 128 registers) leave room for SIX fragment
// sets, not for the sixteen a chunk reads: the B fragments of a step stay (four sets), the A fragments go through TWO sets -- the chunk is 2 NA block rows
// (step 0's, then step 1's), row r multiplies out of set r & 1, and as soon as its MFMAs are issued the set is refilled with row r + 2's fragment: one row
// of MFMAs (64 cycles of this wave's, more with the SIMD's other waves in between) covers the read.  Step 1's B fragments replace step 0's one per MFMA
// inside step 0's last row.  sched_barrier pins this placement (left alone, hipcc sinks every reload to just in front of its first use).
template <int NA, int NC>
__device__ __forceinline__ void mfma16_chunk(f32x4h (&acc)[NA][NC], const char* sa, const int (&oa)[NA], const int (&oa1)[NA], const char* sb, int ob, int ob1) {
    static_assert(NA >= 2, "two A sets alternate over the chunk's 2 NA block rows");
    f16x8 fa[2], fb[NC];
    fa[0] = *(const f16x8*)(sa + oa[0]);
#pragma unroll
    for (int j = 0; j < NC; ++j) fb[j] = *(const f16x8*)(sb + ob + j * 2048);
    fa[1] = *(const f16x8*)(sa + oa[1]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int r = 0; r < 2 * NA; ++r) {
        const int i = r % NA;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[r & 1], fb[j], acc[i][j], 0, 0, 0);
            if (r == NA - 1) fb[j] = *(const f16x8*)(sb + ob1 + j * 2048);
        }
        if (r + 2 < 2 * NA) fa[r & 1] = *(const f16x8*)(sa + (r + 2 < NA ? oa[(r + 2) % NA] : oa1[(r + 2) % NA]));
        __builtin_amdgcn_sched_barrier(0);
    }
}

#ifndef ISEGMI_EPI_D
#define ISEGMI_EPI_D 2
#endif
constexpr int EPI_D = ISEGMI_EPI_D;   // residual passes in flight per lane (tools/build_variant.sh ... -DISEGMI_EPI_D=4 for an A/B)

template <int WR, int TN>   // WR rows x 32 TN columns per wave
struct Epi8 {
    static_assert(TN == 2 || TN == 4, "strips are 64 or 128 channels wide");
    static constexpr int LPR = TN * 4, RPP = 64 / LPR, NQ = WR / RPP;  // passes (b128 per lane) per wave tile
    // residual passes in flight per lane: a 64-row wave tile (64 accumulator registers of the 128 a wave has at 16 waves per CU) leaves room for two.
    // Round 6 tried the other end -- a 96 x 256 tile, 32-row wave tiles, ALL four passes of a wave in flight under the K loop (100 registers, no
    // spill, bit-identical) -- on R101 res4 conv3 (M = 33 600, K = 256, Cout = 1024, with residual): 45 us against 47 (tile 37) / 44 (47) alone, level
    // in the model: the residual round trips are not what bounds these layers, the CU's LDS fill path is (224 KB of A + B per 192 x 256 tile for
    // 25 MFLOP; profiles/r06_experiments.txt 2).  Not kept.
    static constexpr int D = EPI_D < NQ ? EPI_D : NQ;
    u32x4h r[D];
    unsigned rnext;   // residual offset (bytes) of the next pass to request, or >= OOB for a column past Cout; pass q covers tile rows RPP q ..
    int ux, uy, un;   // UP2X: output pixel (x, y, image) of the next pass to request
};

// UP2X (FPN top-down merge, SURVEY 8a M3: `last_inner = inner_lateral + interpolate(last_inner, scale_factor=2, mode="nearest")`): the residual of output pixel
// (n, y, x) is pixel (n, min(y >> 1, Hc - 1), min(x >> 1, Wc - 1)) of the coarser level.  A lane's pass-to-pass step is RPP pixels along the row: the walk is
// incremental (one wrap test per pass), the two divisions that start it are per tile.
template <int WR, int TN>
__device__ __forceinline__ unsigned epi8_up2x_next(const ConvKH& p, Epi8<WR, TN>& E, int co8) {
    constexpr unsigned OOB = 0x80000000u;
    int yc = E.uy >> 1, xc = E.ux >> 1;
    yc = yc > p.rHc - 1 ? p.rHc - 1 : yc;
    xc = xc > p.rWc - 1 ? p.rWc - 1 : xc;
    const unsigned off = co8 < p.Cout ? ((unsigned)((E.un * p.rHc + yc) * p.rWc + xc) * (unsigned)p.Cout + (unsigned)co8) * 2u : OOB;   // images past N are past the range
    E.ux += Epi8<WR, TN>::RPP;
    if (E.ux >= p.Wo) { E.ux -= p.Wo; E.uy += 1; if (E.uy >= p.Ho) { E.uy = 0; E.un += 1; } }
    return off;
}

// before the K loop: the first D residual passes of the wave's tile
template <int WR, int TN, bool UP2X = false>
__device__ __forceinline__ void epi8_prefetch(const ConvKH& p, Epi8<WR, TN>& E, int lane, int wm, int wn, int m0, int n0) {
    constexpr unsigned OOB = 0x80000000u;
    constexpr int LPR = Epi8<WR, TN>::LPR, RPP = Epi8<WR, TN>::RPP;
    // no branch on p.res here: a conditional prefetch makes the window a phi, and the compiler then parks the loaded registers in copies
    // behind an s_waitcnt vmcnt(0) in FRONT of the K loop; without a residual the descriptor is empty and the loads return zeros unused
    const int er = lane / LPR, ec = (lane % LPR) * 8;
    const int co8 = n0 + wn * TN * 32 + ec;
    E.rnext = co8 < p.Cout ? ((unsigned)(m0 + wm * WR + er) * (unsigned)p.Cout + (unsigned)co8) * 2u : OOB;
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res ? (const void*)p.res : (const void*)p.out), 0, p.res ? p.res_bytes : 0u, 0x00020000);
    const unsigned rstep = (unsigned)p.Cout * (2u * RPP);
    if constexpr (UP2X) {
        const int m = m0 + wm * WR + er, hw = p.Ho * p.Wo;
        E.un = m / hw;
        const int rem = m - E.un * hw;
        E.uy = rem / p.Wo;
        E.ux = rem - E.uy * p.Wo;
#pragma unroll
        for (int q = 0; q < Epi8<WR, TN>::D; ++q) E.r[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, epi8_up2x_next<WR, TN>(p, E, co8), 0, CONV_F16_RES_AUX);
        return;
    }
#pragma unroll
    for (int q = 0; q < Epi8<WR, TN>::D; ++q) { E.r[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, E.rnext, 0, CONV_F16_RES_AUX); E.rnext += rstep; }
}

// SR-row strips: 8 rows (registers 4g .. 4g + 3 of the 32 x 32 blocks) or 16 rows (one row of 16 x 16 blocks); ew = the wave's SR x PITCH floats of scratch
template <bool RES, bool UP2X, class ACC>
__device__ __forceinline__ void epi8_finish_impl(const ConvKH& p, ACC& acc, Epi8<acc_traits<ACC>::WR, acc_traits<ACC>::TN>& E, float* ew, int lane, int wm, int wn, int m0, int n0) {
    constexpr int TN = acc_traits<ACC>::TN, MS = acc_traits<ACC>::MS, SR = acc_traits<ACC>::SR, NA = acc_traits<ACC>::NA, WR = acc_traits<ACC>::WR;
    static_assert(MS == 1, "this file holds the 16 x 16 x 32 kernels");
    constexpr unsigned OOB = 0x80000000u;
    constexpr int PITCH = TN * 32 + 4;
    constexpr int LPR = Epi8<WR, TN>::LPR, RPP = Epi8<WR, TN>::RPP, NPASS = SR / RPP, NQ = Epi8<WR, TN>::NQ, D = Epi8<WR, TN>::D;
    constexpr int NCB = 2 * TN, CB = 16;   // column blocks of the wave tile
    static_assert(NA * NPASS == NQ, "passes per wave tile");
    const int lr = MS ? (lane & 15) : (lane & 31), lh = lane >> 5;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)(RES ? (const void*)p.res : (const void*)p.out), 0, RES ? p.res_bytes : 0u, 0x00020000);
    const unsigned esz = p.out_f32 ? 4u : 2u;
    const int er = lane / LPR, ec = (lane % LPR) * 8;
    const int co8 = n0 + wn * TN * 32 + ec;
    const bool cok8 = co8 < p.Cout;
    const unsigned rstep = (unsigned)p.Cout * (2u * RPP);
    // contiguous destination: offset = lane base + pass * step, rows past M are out of the descriptor's range
    const unsigned ostep = (unsigned)p.out_pix_stride * esz * RPP;
    unsigned onext = cok8 ? ((unsigned)(m0 + wm * WR + er) * (unsigned)p.out_pix_stride + (unsigned)co8) * esz : OOB;
    float sc[NCB], sh[NCB];
#pragma unroll
    for (int b = 0; b < NCB; ++b) {
        const int co = n0 + wn * TN * 32 + b * CB + lr;
        const bool cok = co < p.Cout;
        sc[b] = (cok && p.scale) ? p.scale[co] : 1.0f;
        sh[b] = (cok && p.shift) ? p.shift[co] : 0.0f;
    }
#pragma unroll
    for (int a = 0; a < NA; ++a) {   // block row a: register j = tile row 16 a + 4 * (lane >> 4) + j
        {
            (void)lh;
#pragma unroll
            for (int b = 0; b < NCB; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j) ew[(j + 4 * (lane >> 4)) * PITCH + b * 16 + lr] = fmaf(acc[a][b][j], sc[b], sh[b]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const int q = a * NPASS + ps;
                const int rr = ps * RPP + er;
                const f32x4h v0 = *(const f32x4h*)(ew + rr * PITCH + ec);
                const f32x4h v1 = *(const f32x4h*)(ew + rr * PITCH + ec + 4);
                unsigned ooff;
                if (p.contiguous) { ooff = onext; onext += ostep; }
                else {
                    const int m = m0 + wm * WR + q * RPP + er;
                    const int ni = m / p.out_div, pi = m - ni * p.out_div;
                    ooff = (m < p.M && cok8) ? (unsigned)(((int64_t)ni * p.out_img_stride + (int64_t)pi * p.out_pix_stride + co8) * esz) : OOB;
                }
                float y[8];
                if (RES) {
                    const f16x8 rh = __builtin_bit_cast(f16x8, E.r[q % D]);
                    if constexpr (UP2X) {
                        // the lateral conv's result is rounded to fp16 FIRST, as the two-launch path stores it before nearest2x_add_f16 adds the coarser level
#pragma unroll
                        for (int i = 0; i < 8; ++i) y[i] = (float)(half_t)(i < 4 ? v0[i] : v1[i - 4]) + (float)rh[i];
                        if (q + D < NQ) E.r[q % D] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, epi8_up2x_next<WR, TN>(p, E, co8), 0, CONV_F16_RES_AUX);
                    } else {
#pragma unroll
                        for (int i = 0; i < 8; ++i) y[i] = (i < 4 ? v0[i] : v1[i - 4]) + (float)rh[i];
                        if (q + D < NQ) { E.r[q % D] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, E.rnext, 0, CONV_F16_RES_AUX); E.rnext += rstep; }
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) y[i] = i < 4 ? v0[i] : v1[i - 4];
                }
                if (p.act == 1) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) y[i] = y[i] > 0.0f ? y[i] : 0.0f;
                }
                if (p.out_f32) {
                    u32x4h o0, o1;
#pragma unroll
                    for (int i = 0; i < 4; ++i) { o0[i] = __builtin_bit_cast(unsigned, y[i]); o1[i] = __builtin_bit_cast(unsigned, y[i + 4]); }
                    __builtin_amdgcn_raw_buffer_store_b128(o0, rs_out, ooff, 0, CONV_F16_OUT_AUX);
                    __builtin_amdgcn_raw_buffer_store_b128(o1, rs_out, ooff >= OOB ? OOB : ooff + 16u, 0, CONV_F16_OUT_AUX);
                } else {
                    f16x8 o;
#pragma unroll
                    for (int i = 0; i < 8; ++i) o[i] = (half_t)y[i];
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4h, o), rs_out, ooff, 0, CONV_F16_OUT_AUX);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
}

template <bool UP2X, class ACC>
__device__ __forceinline__ void epi8_finish(const ConvKH& p, ACC& acc, Epi8<acc_traits<ACC>::WR, acc_traits<ACC>::TN>& E, float* ew, int lane, int wm, int wn, int m0, int n0) {
    if constexpr (UP2X) { epi8_finish_impl<true, true>(p, acc, E, ew, lane, wm, wn, m0, n0); return; }
    if (p.res) epi8_finish_impl<true, false>(p, acc, E, ew, lane, wm, wn, m0, n0);   // uniform
    else epi8_finish_impl<false, false>(p, acc, E, ew, lane, wm, wn, m0, n0);
}

// bytes of strip scratch the persistent kernel needs BEHIND its ring: the epilogue strips of the waves that do not fit the stage read last (at least the
// 5120 B of round 5's 144 x 256 tile, whose layout tests pin)
template <int BM, int BN, int WM, int WN>
constexpr int p_espare() {
    constexpr int NW = WM * WN, TN = BN / WN / 32, STAGEB = (BM + BN) * 128, EWB = 16 * (TN * 32 + 4) * 4;
    constexpr int EFIT = STAGEB / EWB < NW ? STAGEB / EWB : NW;
    constexpr int need = (NW - EFIT) * EWB;
    return need > 5120 ? (need + 1023) / 1024 * 1024 : 5120;
}
template <int BM, int BN, int WM, int WN, int NSTAGE, int OCC, int LW, bool UP2X = false, int MS = 0>
__global__ __launch_bounds__((WM * WN + LW) * 64, OCC) void conv_f16_persist_kernel(const ConvKH p_arg) {
    static_assert(LW > 0 && NSTAGE >= 2 && NSTAGE <= 3, "loader waves, a 2- or 3-deep ring");
    // Kernel arguments are read where they are used, per role: taken by value the ~45 dwords of ConvKH are all loaded at entry and stay
    // live in SGPRs across both roles' loops (29 SGPR spills in round 2: the epilogue re-read its buffer descriptors with v_readlane before
    // every store group).  Each role reads the fields it needs from the kernarg segment behind an opaque copy of its address (the empty asm
    // keeps the loads from being hoisted back to the entry block); the epilogue's fields are re-read per tile, right in front of it.
    (void)p_arg;
    typedef const ConvKH __attribute__((address_space(4)))* karg_t;
    karg_t kp0 = (karg_t)__builtin_amdgcn_kernarg_segment_ptr();
    const int p_mtiles = kp0->mtiles, p_ntiles = kp0->ntiles, p_nchunks = kp0->nchunks;
    constexpr int NW = WM * WN, NL = LW;
    static_assert(MS == 1 && (BM / WM) % 16 == 0 && BM % WM == 0, "16 x 16 x 32: a wave owns a whole number of 16-row blocks");
    constexpr int WR = BM / WM, TN = BN / WN / 32;   // rows per wave, 32-column units per wave
    constexpr int PA = BM / 8, PB = BN / 8;
    constexpr bool UNEVEN = (PA % NL != 0) || (PB % NL != 0);
    // (a partial piece round issues one dropped piece in its place -- see issue_chunk -- so a wave's vmcnt count stays uniform)
    constexpr int STAGEB = (BM + BN) * 128;
    constexpr int ESR = 16;            // rows per epilogue strip (acc_traits::SR)
    constexpr int EWB = ESR * (TN * 32 + 4) * 4;   // a wave's strip scratch (bytes)
    // the epilogue's strips live in the ring stage the MFMA waves read last; where that stage is too small for all of them (144 x 256: 51 200 B for
    // twelve strips of 4352 B) the waves past EFIT take theirs from the spare LDS behind the ring (launch_p sizes it)
    constexpr int EFIT = STAGEB / EWB < NW ? STAGEB / EWB : NW;
    constexpr int ESPARE = p_espare<BM, BN, WM, WN>();   // bytes behind the ring for those waves' strips (launch_p sizes the same)
    (void)UNEVEN;
    extern __shared__ __attribute__((aligned(1024))) char smemg[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int total = p_mtiles * p_ntiles, G = (int)gridDim.x, bid = (int)blockIdx.x;
    const int q8 = total >> 3, r8g = total & 7;
    auto tile_origin = [&](int v, int& m0, int& n0) {  // v = bid + i * G: same XCD as bid (G is a multiple of 8 or the whole grid)
        const int xcd = v & 7;
        const int logical = (xcd < r8g ? xcd * (q8 + 1) : r8g * (q8 + 1) + (xcd - r8g) * q8) + (v >> 3);
        const int nt = logical % p_ntiles, mt = logical / p_ntiles;
        m0 = mt * BM; n0 = nt * BN;
    };
    const int my_tiles = (total - bid + G - 1) / G;  // >= 1: the launcher never starts more blocks than tiles

    conv_f16_role_prio(wave >= NW);
    if (wave >= NW) {   // ---------------- loader waves (conv_f16.h: shared with the other MFMA shape's translation unit)
        conv_f16_persist_loader<BM, BN, NW, NL, NSTAGE, ESPARE>(kp0, smemg, wave, lane, bid, G, total, my_tiles, p_ntiles);
        return;
    }

    // ---------------- MFMA waves (K loop: tile geometry only; the epilogue reads its arguments per tile)
    const int wm = wave / WN, wn = wave % WN;
    const int a_off = wm * WR * 128;
    const int b_off = BM * 128 + wn * TN * 32 * 128;
    constexpr int NA = WR / 16, NC = 2 * TN, NE = 4;   // accumulator blocks down / across, registers per block
    typedef f32x4h acc_t[NA][NC];
    acc_t acc;
    const int swz16 = (lane & 15) * 128 + (((lane >> 4) ^ (((lane & 15) >> 1) & 7)) << 4);   // 16 x 16 x 32 form: 32-deep step s adds ^ (s << 6)
    auto chunk = [&](int rd) {  // run-time stage: branching over compile-time stages made hipcc copy and spill the accumulators
        const char* sb = smemg + rd * STAGEB;
        int oa[NA], oa1[NA];
#pragma unroll
        for (int i = 0; i < NA; ++i) { oa[i] = a_off + swz16 + i * 2048; oa1[i] = a_off + (swz16 ^ 64) + i * 2048; }
        mfma16_chunk<NA, NC>(acc, sb, oa, oa1, sb, b_off + swz16, b_off + (swz16 ^ 64));
    };
    int st = 0;
    for (int v = bid; v < total; v += G) {
        int m0, n0;
        tile_origin(v, m0, n0);
#pragma unroll
        for (int a = 0; a < NA; ++a)
#pragma unroll
            for (int b = 0; b < NC; ++b)
#pragma unroll
                for (int e = 0; e < NE; ++e) acc[a][b][e] = 0.0f;
        Epi8<WR, TN> E;
        {
            karg_t k1 = kp0;
            asm volatile("" : "+s"(k1));
            const ConvKH& p = *(const ConvKH*)k1;
            if (p.vec_epi) epi8_prefetch<WR, TN, UP2X>(p, E, lane, wm, wn, m0, n0);  // the tile's first residual strips travel under its K loop
        }
        int last = 0;
        for (int t = 0; t < p_nchunks; ++t) {
            asm volatile("s_barrier" ::: "memory");
            chunk(st);
            last = st;
            st = st + 1 == NSTAGE ? 0 : st + 1;
        }
        asm volatile("s_barrier" ::: "memory");  // E: every MFMA wave has read the last chunk; its stage is now scratch
        karg_t k2 = kp0;
        asm volatile("" : "+s"(k2));
        const ConvKH& p = *(const ConvKH*)k2;
        if (p.vec_epi) epi8_finish<UP2X>(p, acc, E, (float*)(wave < EFIT ? smemg + last * STAGEB + wave * EWB : smemg + NSTAGE * STAGEB + (wave - EFIT) * EWB), lane, wm, wn, m0, n0);
        else conv_f16_epilogue(p, acc, smemg, wave, lane, wm, wn, m0, n0);  // per-element path: no LDS
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// FUSED 1x1 HEAD (round 4; RPNHead under configs[4]: `t = relu(conv3x3(x)); logits, deltas = cls_logits(t), bbox_pred(t)`).  On a 192 x 256
// row-strip tile the block holds ALL 256 channels of its 192 pixels, so the 1x1 convolution that follows (256 -> f_cout <= 32 outputs: 3 objectness + 12
// box deltas) can run on the tile before it ever leaves the CU: the BN + ReLU result goes to LDS as fp16 -- rounded exactly where the two-launch path
// rounds it when it stores t -- in a row-major [pixel][256] image (16-B columns XOR-swizzled by the row), the head's packed weights (32 rows) next to
// it, and six waves multiply one 32-pixel tile each on the same 16 k-steps the stand-alone 1x1 launch would walk.  t is never written (275 MB per
// P2 level at R101 bs=8) nor read back.  Bit-identical to the two launches (tests/test_rpn_head_f16_gpu.py).
template <class ACC>
__device__ __forceinline__ void conv_f16_epilogue_head(const ConvKH& p, ACC& acc, char* smemg, int wave, int lane, int wm, int wn, int m0) {
    constexpr int TN = acc_traits<ACC>::TN, MS = acc_traits<ACC>::MS, NA = acc_traits<ACC>::NA, NC = acc_traits<ACC>::NC, WR = acc_traits<ACC>::WR;
    constexpr int RB = MS ? 16 : 32, NE = MS ? 4 : 16;
    constexpr unsigned OOB = 0x80000000u;
    constexpr int BMH = 192, T_BYTES = BMH * 512;   // the tile image; the head's weights follow it (32 rows x 512 B)
    const int lr = lane & 31, lh = lane >> 5;
    const int tid = wave * 64 + lane;                // 12 MFMA waves: 768 threads
    // head weights: rows 0..31 of the packed [128][256] image (zero rows past f_cout), 16 KB = 1024 16-B pieces
    u32x4h wv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int piece = tid + j * 768;
        wv[j] = piece < 1024 ? *(const u32x4h*)((const char*)p.f_w + piece * 16) : u32x4h{0u, 0u, 0u, 0u};
    }
    // 1. y = relu(fmaf(acc, scale, shift)) -> fp16 -> T[row][channel]
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NC; ++b) {
            const int co = wn * TN * 32 + b * RB + (MS ? (lane & 15) : lr);
            const float sc = p.scale ? p.scale[co] : 1.0f, sh = p.shift ? p.shift[co] : 0.0f;
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const int row = wm * WR + a * RB + (MS ? 4 * (lane >> 4) + e : (e & 3) + 8 * (e >> 2) + 4 * lh);
                float y = fmaf(acc[a][b][e], sc, sh);
                y = y > 0.0f ? y : 0.0f;
                *(half_t*)(smemg + row * 512 + (((co >> 3) ^ (row & 15)) << 4) + (co & 7) * 2) = (half_t)y;
            }
        }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int piece = tid + j * 768;
        if (piece < 1024) {
            const int row = piece >> 5, c16 = piece & 31;
            *(u32x4h*)(smemg + T_BYTES + row * 512 + ((c16 ^ (row & 15)) << 4)) = wv[j];
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // the MFMA waves (the loader waves have left)
    if (wave >= BMH / 32) return;
    // 2. wave w: pixels 32 w .. + 31 of the tile x 32 head outputs, K = 256 in the order the stand-alone 1x1 launch walks it
    f32x16h h;
#pragma unroll
    for (int e = 0; e < 16; ++e) h[e] = 0.0f;
    const int prow = wave * 32 + lr;
    const char* pa = smemg + prow * 512;
    const char* pb = smemg + T_BYTES + lr * 512;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        const f16x8 fa = *(const f16x8*)(pa + (((ks * 2 + lh) ^ (prow & 15)) << 4));
        const f16x8 fb = *(const f16x8*)(pb + (((ks * 2 + lh) ^ (lr & 15)) << 4));
        h = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, h, 0, 0, 0);
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.f_out, 0, p.f_out_bytes, 0x00020000);
    const bool cok = lr < p.f_cout;
    const float sc2 = (cok && p.f_scale) ? p.f_scale[lr] : 1.0f, sh2 = (cok && p.f_shift) ? p.f_shift[lr] : 0.0f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int m = m0 + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        float y = fmaf(h[e], sc2, sh2);
        y = y + 0.0f;   // the stand-alone launch's per-element epilogue adds its (absent) residual: -0 becomes +0 there, so here too
        const unsigned off = (cok && m < p.M) ? ((unsigned)m * (unsigned)p.f_cout + (unsigned)lr) * 4u : OOB;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y), rs, off, 0, 0);
    }
}

// 3x3 / stride 1 / pad 1 with ROW-STRIP staging.  The generic kernel above stages a fresh BM-row A image for each of
// the nine taps; here the three taps of one filter row share ONE strip: for filter row r and cin chunk kc the block
// loads, per image-row segment its BM output pixels touch, [left neighbour | the segment's pixels | right neighbour]
// (a zero row where the neighbour is padding), so tap s of output pixel m is strip row j(m) + s - 1 with
// j(m) = (m - m0) + 1 + 2 * (m / W - m0 / W).  A bytes drop ~2.7x (B unchanged: one BN x 64 chunk per tap).  The
// CU's fill path (~40-50 GB/s of LDS-DMA per CU), not MFMA, bounds these layers -- see DESIGN.md.
// Step u = (r, kc, s), s fastest: A strips are double-buffered per (r, kc) group and filled a third per step, B chunks
// double-buffered per step; every step starts with `s_waitcnt vmcnt(0); s_barrier`.  K is walked as (r, kc, s) instead of
// (r, s, kc): fp32 accumulation order differs from the generic kernel within the stated fp16 tolerance.
// NB = 3 (round 4, loader-wave form only): THREE B buffers.  With two, a step's loads are issued after its opening barrier and must ALL have landed at the
// next one; with three, step u issues its third of the next strip FIRST and then the B chunk of step u + 2 into the buffer step u - 1 read, and the wait in
// front of the next barrier leaves exactly those B pieces in flight (`vmcnt(RB)`, in-order completion).  Worth +3 % on the 634-GF layer and no more: with
// the loaders stopped after two groups (tile bit 4096) the kernel is only 8 % faster, and in a loop it holds the package at 1381 of its 1400 W
// (profiles/r04_f16_power.txt): these layers are bound by the power cap, not by a pipe.  LDS: 2 strips + 3 B chunks = 163 840 B, all there is.
#ifdef ISEGMI_STRIP_TRACE
// development build only (-DISEGMI_STRIP_TRACE, tools/strip_trace.py): block 0 keeps four cycle stamps per step and wave in LDS behind the staging buffers and dumps them to p.trace
#define STRIP_TRACE(u, slot)                                                                                      \
    do {                                                                                                          \
        if (bid == 0 && (u) < 64 && !(p.dbg & 16)) {                                                              \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                           \
            *(volatile unsigned*)(smemg + TRACE_OFF + ((wave * 64 + (u)) * 4 + (slot)) * 4) = (unsigned)t_;        \
        }                                                                                                         \
    } while (0)
#else
#define STRIP_TRACE(u, slot) do {} while (0)
#endif
template <int BM, int BN, int WM, int WN, int LW, int NB = 2, int MS = 0>
__global__ __launch_bounds__((WM * WN + LW) * 64, 1) void conv3x3_f16_strip_kernel(const ConvKH p) {
    static_assert(NB == 2 || (NB == 3 && LW > 0 && (BN / 8) % LW == 0), "the three-buffer form needs loader waves and whole B piece rounds");
    constexpr int NW = WM * WN;
    constexpr int NL = LW > 0 ? LW : NW;            // waves that issue loads (LW > 0: dedicated loader waves, see above)
    static_assert(MS == 1 && (BM / WM) % 16 == 0 && BM % WM == 0, "16 x 16 x 32: a wave owns a whole number of 16-row blocks");
    constexpr int WR = BM / WM, TN = BN / WN / 32;   // rows per wave, 32-column units per wave
#include "conv_f16_strip_state.inc"   // tile geometry, loader state, issue_strip / next_group / issue_b / next_b: shared with the other MFMA shape's translation unit

    constexpr int NA = WR / 16, NC = 2 * TN, NE = 4, RBLK = 16;   // accumulator blocks down / across, registers per block, block rows
    typedef f32x4h acc_t[NA][NC];
    acc_t acc;
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int b = 0; b < NC; ++b)
#pragma unroll
            for (int e = 0; e < NE; ++e) acc[a][b][e] = 0.0f;

    // ---- fragment addressing: A rows are strip rows j(m) + s - 1 (per lane, per block row a, per tap s); the lane's row inside a block is lane & 31
    // (32 x 32 x 16 form, 16-B column 2 ks + (lane >> 5)) or lane & 15 (16 x 16 x 32 form, column 4 ks + (lane >> 4))
    const int lr = MS ? (lane & 15) : (lane & 31), lh = MS ? (lane >> 4) : (lane >> 5);
    int abase_s[NA][3];
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        const int m = m0 + wm * WR + a * RBLK + lr;
        const int jm = (m - m0) + 1 + 2 * (m / W - row0);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int row = jm + s - 1;
            abase_s[a][s] = row * 128 + ((lh ^ ((row >> 1) & 7)) << 4);  // k-step ks adds ^ (ks << 5) (16-deep steps) / ^ (ks << 6) (32-deep steps)
        }
    }
    const int swzb = lr * 128 + ((lh ^ ((lr >> 1) & 7)) << 4);
    const int b_off = 2 * ABYTES + wn * TN * 32 * 128;

    // prologue: whole strip of group 0, B of step 0 (and of step 1 with three B buffers)
    if (loads) {
        issue_strip(0, 0); issue_strip(1, 0); issue_strip(2, 0);
        next_group();
        issue_b(0);
        next_b();
        if (NB == 3) { issue_b(1); next_b(); }
    }
#ifdef ISEGMI_STRIP_TRACE
    unsigned long long tr_c0 = 0, tr_r0 = 0;
    if (bid == 0) { tr_c0 = __builtin_amdgcn_s_memtime(); tr_r0 = __builtin_amdgcn_s_memrealtime(); }
#endif
    if (LW > 0) conv_f16_role_prio(wave >= NW);
#include "conv_f16_strip_loader_loop.inc"   // loader waves: the barrier sequence of the chunk loop below, loads only (shared text)
    // unrolled over two groups so that both LDS stages are compile-time terms (A buffer = gi & 1, B buffer = (gi + s) & 1, because
    // u = 3 gi + s): fragment addresses are a per-lane base plus an immediate
    for (int g0 = 0; g0 < ngroups; g0 += 2) {
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
            if (g0 + gg >= ngroups) break;  // uniform
            const char* sa = smemg + gg * ABYTES;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                STRIP_TRACE((g0 + gg) * 3 + s, 0);
                if (LW == 0) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
                else asm volatile("s_barrier" ::: "memory");
                STRIP_TRACE((g0 + gg) * 3 + s, 1);
                const int ub = NB == 3 ? s : (gg + s) & 1;   // step u = 3 gi + s: u % 3 = s
                const char* sb = smemg + b_off + ub * BBYTES;
                static_assert(LW > 0, "the 16 x 16 x 32 form exists with loader waves only");
                int oa0[NA], oa1[NA];
#pragma unroll
                for (int a = 0; a < NA; ++a) { oa0[a] = abase_s[a][s]; oa1[a] = abase_s[a][s] ^ 64; }
                mfma16_chunk<NA, NC>(acc, sa, oa0, oa1, sb, swzb, swzb ^ 64);
            }
            if (LW == 0) next_group();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    STRIP_TRACE(63, 0);
    if constexpr (BM == 192 && BN == 256 && WM == 3 && WN == 4 && NB == 3) {
        if (p.f_w) { conv_f16_epilogue_head(p, acc, smemg, wave, lane, wm, wn, m0); return; }   // uniform
    }
    conv_f16_epilogue(p, acc, smemg, wave, lane, wm, wn, m0, n0);
    STRIP_TRACE(63, 1);
#ifdef ISEGMI_STRIP_TRACE
    if (bid == 0 && p.trace && wave == 0 && lane == 0) {
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        p.trace[4096] = (unsigned)(c1 - tr_c0); p.trace[4097] = (unsigned)(r1 - tr_r0);
    }
    if (bid == 0 && p.trace) for (int i = lane; i < 256; i += 64) p.trace[wave * 256 + i] = *(volatile unsigned*)(smemg + TRACE_OFF + (wave * 256 + i) * 4);
#endif
}

template <int BM, int BN, int WM, int WN, int LW = 0, int NB = 2, int MS = 0>
static int launch_strip(ConvKH& k, hipStream_t st) {
    k.mtiles = cdiv(k.M, BM);
    k.ntiles = cdiv(k.Cout, BN);
    constexpr int NW = WM * WN, TN = BN / WN / 32;
    size_t lds = 2 * (size_t)(BM + 64) * 128 + NB * (size_t)BN * 128;
    static_assert(2 * (BM + 64) * 128 + NB * BN * 128 <= 163840, "LDS");
    const size_t epi = (size_t)NW * 16 * (TN * 32 + 4) * 4;
    if (epi > lds) lds = epi;
    size_t lds_attr = lds;
#ifdef ISEGMI_STRIP_TRACE
    if (NB == 2) lds_attr = lds + 16384;
    static unsigned* trace_buf = nullptr;
    const bool tracing = NB == 2 && getenv("ISEGMI_STRIP_TRACE_DUMP") != nullptr;   // (NB = 3 fills the LDS: no room for the stamps)
    if (tracing) {
        lds = 2 * (size_t)(BM + 64) * 128 + NB * (size_t)BN * 128 + 16384;
        if (!trace_buf) HIP_TRY(hipMalloc((void**)&trace_buf, 16448));
        HIP_TRY(hipMemsetAsync(trace_buf, 0, 16448, st));
    }
    k.trace = tracing ? trace_buf : nullptr;
    if (tracing && getenv("ISEGMI_STRIP_TRACE_LIGHT")) k.dbg |= 16;
#endif
    LDS_LIMIT_ONCE((int)lds_attr, conv3x3_f16_strip_kernel<BM, BN, WM, WN, LW, NB, MS>);
    hipLaunchKernelGGL((conv3x3_f16_strip_kernel<BM, BN, WM, WN, LW, NB, MS>), dim3((unsigned)(k.mtiles * k.ntiles)), dim3((NW + LW) * 64), lds, st, k);
    HIP_TRY(hipGetLastError());
#ifdef ISEGMI_STRIP_TRACE
    if (tracing) {
        static unsigned host[4112];
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(hipMemcpy(host, trace_buf, 16448, hipMemcpyDeviceToHost));
        FILE* f = fopen(getenv("ISEGMI_STRIP_TRACE_DUMP"), "w");
        if (f) {
            fprintf(f, "-1 -1 %u %u 0 0\n", host[4096], host[4097]);
            for (int w = 0; w < NW + LW; ++w)
                for (int u = 0; u < 64; ++u) fprintf(f, "%d %d %u %u %u %u\n", w, u, host[(w * 64 + u) * 4], host[(w * 64 + u) * 4 + 1], host[(w * 64 + u) * 4 + 2], host[(w * 64 + u) * 4 + 3]);
            fclose(f);
        }
    }
#endif
    return ISEGMI_OK;
}

// persistent loader-wave kernel: at most one block per CU slot, each walking tiles bid, bid + grid, ...; `few` (test hook) forces
// an 8-block grid so that small test shapes exercise the multi-tile stream
template <int BM, int BN, int WM, int WN, int NSTAGE, int OCC, int LW, bool UP2X = false, int MS = 0>
static int launch_p(ConvKH& k, hipStream_t st, bool few) {
    k.mtiles = cdiv(k.M, BM);
    k.ntiles = cdiv(k.Cout, BN);
    constexpr int NW = WM * WN, TN = BN / WN / 32;
    size_t lds = (size_t)NSTAGE * (BM + BN) * 128;
    // behind the ring: 5120 B for the strip scratch of the waves that do not fit the last stage, LW KiB of landing zone for dropped pieces
    // (conv_f16_persist_kernel: EFIT, issue_chunk) -- only where the tile needs them
    constexpr int ESPARE = p_espare<BM, BN, WM, WN>();
    if ((BM / 8) % LW != 0 || (size_t)NW * 16 * (TN * 32 + 4) * 4 > (size_t)(BM + BN) * 128) lds += ESPARE + (size_t)LW * 1024;
    static_assert(NSTAGE * (BM + BN) * 128 + ESPARE + LW * 1024 <= 163840 || ((BM / 8) % LW == 0 && NW * 16 * (TN * 32 + 4) * 4 <= (BM + BN) * 128), "LDS");
    LDS_LIMIT_ONCE((int)lds, conv_f16_persist_kernel<BM, BN, WM, WN, NSTAGE, OCC, LW, UP2X, MS>);
    const int ncu = device_cu_count();
    const int64_t total = (int64_t)k.mtiles * k.ntiles;
    int64_t slots = few ? 8 : (int64_t)(ncu / 8) * 8 * OCC;  // a multiple of 8, so that a block's tiles stay on its XCD
    if (slots < 8) slots = 8;
    const unsigned grid = (unsigned)(total < slots ? total : slots);
    hipLaunchKernelGGL((conv_f16_persist_kernel<BM, BN, WM, WN, NSTAGE, OCC, LW, UP2X, MS>), dim3(grid), dim3((NW + LW) * 64), lds, st, k);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}


int conv_f16_m16_launch(int tile, ConvKH& k, hipStream_t st, bool few) {
    switch (tile) {
        case 40: return launch_strip<192, 256, 3, 4, 4, 3, 1>(k, st);
        case 41: return launch_strip<144, 256, 3, 4, 4, 3, 1>(k, st);                 // 12 MFMA waves of 48 x 64: M = 33 600 (res4 at bs 8) is 234 tiles, one round
        case 46: return launch_p<144, 256, 3, 4, 3, 1, 4, false, 1>(k, st, few);      // the same rows for the 1x1 layers, three-deep ring
        case 44: return launch_p<256, 128, 4, 2, 3, 1, 4, false, 1>(k, st, few);
        case 47: return launch_p<192, 256, 3, 4, 2, 1, 4, false, 1>(k, st, few);
        case 48: return launch_p<192, 256, 3, 4, 2, 1, 4, true, 1>(k, st, few);
        case 49: return launch_p<128, 256, 2, 4, 3, 1, 4, false, 1>(k, st, few);
        default: break;
    }
    ARG_CHECK(false, "unknown 16 x 16 x 32 conv tile");
}

}  // namespace isegmi
