// conv_mfma_f16.hip -- NHWC implicit-GEMM convolution on v_mfma_f32_32x32x16_f16 (gfx950): fp16 storage,
// fp32 accumulate.  BASELINE.json configs[4] ("Mask R-CNN R101-FPN ... with fp16 MFMA conv"); SURVEY 8a M2-M11
// "num. type f32 (f16 cfg5)".
//
// One accumulator per output, K walked as (r, s, cin) in 64-half chunks, padding by the buffer range check, fused
// scale/shift/residual/ReLU epilogue in fp32.  The f16 MFMA consumes 8 consecutive k per lane half (lane (r,h) holds
// k = 8h..8h+7 of a 16-deep step), so the NHWC channel run is used as stored.  Numerics: products are exact in fp32, the
// 16-term sum inside one MFMA is not an ordered fmaf chain, so parity with the oracle is TOLERANCE-based here (tests
// state it).
#include "conv_f16.h"

namespace isegmi {


// Shared epilogue (fp32 math): y = fmaf(acc, scale, shift) + residual -> act -> fp16 (or fp32) NHWC store.  Vector path: each
// wave transposes its 32-row fp32 strips through a private LDS region (all staging LDS is free by now) so that residual
// loads and output stores are 16 B per lane; strided destinations / Cout % 8 != 0 take the per-element path.
// vector path of conv_f16_epilogue (32-row strips through the wave's private LDS region).  RES: a residual is added -- its 16-B loads run
// a rolling window of two passes ahead of their use (see epi8_prefetch: a load awaited on the spot also drains the previous pass's stores);
// without a residual no load is issued at all (round 2 sent a dropped out-of-range load per pass and waited for it).
template <int TM, int TN, bool RES>
__device__ __forceinline__ void conv_f16_epilogue_vec(const ConvKH& p, f32x16h (&acc)[TM][TN], char* smemg, int wave, int lane, int wm, int wn, int m0, int n0) {
    constexpr unsigned OOB = 0x80000000u;
    constexpr int PITCH = TN * 32 + 4;
    constexpr int LPR = TN * 4, RPP = 64 / LPR, NPASS = 32 / RPP, NQ = TM * NPASS, D = 2 < NQ ? 2 : NQ;
    const int lr = lane & 31, lh = lane >> 5;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)(RES ? (const void*)p.res : (const void*)p.out), 0, RES ? p.res_bytes : 0u, 0x00020000);
    const unsigned esz = p.out_f32 ? 4u : 2u;
    float* ew = (float*)smemg + wave * 32 * PITCH;
    const int er = lane / LPR, ec = (lane % LPR) * 8;
    const int co8 = n0 + wn * TN * 32 + ec;
    const bool cok8 = co8 < p.Cout;
    // pass q covers tile rows RPP q ..: a lane's offsets advance by one stride per pass; rows past M are behind the descriptors' ranges
    // (contiguous destinations; strided ones take the general arithmetic)
    const unsigned rstep = (unsigned)p.Cout * (2u * RPP), ostep = (unsigned)p.out_pix_stride * esz * RPP;
    unsigned rnext = cok8 ? ((unsigned)(m0 + wm * TM * 32 + er) * (unsigned)p.Cout + (unsigned)co8) * 2u : OOB;
    unsigned onext = cok8 ? ((unsigned)(m0 + wm * TM * 32 + er) * (unsigned)p.out_pix_stride + (unsigned)co8) * esz : OOB;
    u32x4h rw[D];
    if (RES) {
#pragma unroll
        for (int q = 0; q < D; ++q) { rw[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, rnext, 0, CONV_F16_RES_AUX); rnext += rstep; }
    }
#pragma unroll
    for (int a = 0; a < TM; ++a) {
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int co = n0 + (wn * TN + b) * 32 + lr;
            const bool cok = co < p.Cout;
            const float sc = (cok && p.scale) ? p.scale[co] : 1.0f;
            const float sh = (cok && p.shift) ? p.shift[co] : 0.0f;
#pragma unroll
            for (int e = 0; e < 16; ++e) ew[((e & 3) + 8 * (e >> 2) + 4 * lh) * PITCH + b * 32 + lr] = fmaf(acc[a][b][e], sc, sh);
        }
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
                const int m = m0 + wm * TM * 32 + q * RPP + er;
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

template <int TM, int TN>
__device__ __forceinline__ void conv_f16_epilogue(const ConvKH& p, f32x16h (&acc)[TM][TN], char* smemg, int wave, int lane, int wm, int wn,
                                                  int m0, int n0) {
    constexpr unsigned OOB = 0x80000000u;
    const int lr = lane & 31, lh = lane >> 5;
    if (p.vec_epi) {  // uniform
        if (p.res) conv_f16_epilogue_vec<TM, TN, true>(p, acc, smemg, wave, lane, wm, wn, m0, n0);
        else conv_f16_epilogue_vec<TM, TN, false>(p, acc, smemg, wave, lane, wm, wn, m0, n0);
        return;
    }
    // ---- per-element path (strided destinations with Cout % 8 != 0: the RPN / prediction heads): y = fmaf(acc, scale, shift) + residual -> act
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res ? (const void*)p.res : (const void*)p.out), 0, p.res ? p.res_bytes : 0u, 0x00020000);
    const unsigned esz = p.out_f32 ? 4u : 2u;
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        unsigned rowoff[16], resoff[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = m0 + (wm * TM + a) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
            resoff[e] = m < p.M ? (unsigned)m * (unsigned)p.Cout * 2u : OOB;
            if (p.contiguous) rowoff[e] = m < p.M ? (unsigned)m * (unsigned)p.out_pix_stride * esz : OOB;
            else {
                const int ni = m / p.out_div, pi = m - ni * p.out_div;
                rowoff[e] = m < p.M ? (unsigned)(((int64_t)ni * p.out_img_stride + (int64_t)pi * p.out_pix_stride) * esz) : OOB;
            }
        }
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int co = n0 + (wn * TN + b) * 32 + lr;
            const bool cok = co < p.Cout;
            const float sc = (cok && p.scale) ? p.scale[co] : 1.0f;
            const float sh = (cok && p.shift) ? p.shift[co] : 0.0f;
            const unsigned cooff = cok ? (unsigned)co : OOB;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
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

// LW > 0: LW extra LOADER waves issue every LDS-DMA piece and the NW MFMA waves issue none (an LDS-DMA instruction stalls its
// wave for 100-180 cycles while the fill path is busy -- time the MFMA waves then spend on matrix work); LW == 0: every wave
// loads its share between its MFMAs.
template <int BM, int BN, int WM, int WN, int NSTAGE, int OCC, bool STEM, int LW>
__global__ __launch_bounds__((WM * WN + LW) * 64, OCC) void conv_f16_glds_kernel(const ConvKH p) {
    constexpr int NW = WM * WN;
    constexpr int NL = LW > 0 ? LW : NW;  // waves that issue loads
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int PA = BM / 8, PB = BN / 8;                          // 1-KiB pieces per chunk
    constexpr int PPA = (PA + NL - 1) / NL, PPB = (PB + NL - 1) / NL;  // rounds per loading wave (the last one may be partial)
    constexpr bool UNEVEN = (PA % NL != 0) || (PB % NL != 0);
    static_assert(!UNEVEN || NSTAGE == 2, "a partial piece round changes a wave's vmcnt count: only with the vmcnt(0) ring");
    constexpr int STAGEB = (BM + BN) * 128;
    static_assert(PPA >= 1 && PPB >= 1 && TM >= 1 && TN >= 1, "tile/wave split");
    extern __shared__ __attribute__((aligned(1024))) char smemg[];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform, and known to be (scalar address arithmetic)
    const int wm = wave / WN, wn = wave % WN;
    const int lw = LW > 0 ? wave - NW : wave;           // index among the loading waves
    const bool loads = LW == 0 || wave >= NW;

    const int nwg = gridDim.x, bid = blockIdx.x;
    const int q8 = nwg >> 3, r8g = nwg & 7, xcd = bid & 7;
    const int logical = (xcd < r8g ? xcd * (q8 + 1) : r8g * (q8 + 1) + (xcd - r8g) * q8) + (bid >> 3);
    const int nt = logical % p.ntiles, mt = logical / p.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;

    const int r8 = lane >> 3, cs = lane & 7;
    int hi0[PPA], wi0[PPA], abase[PPA];
#pragma unroll
    for (int j = 0; j < PPA; ++j) {
        if (!loads) break;
        const int row = (lw + j * NL) * 8 + r8;
        const int c = cs ^ ((row >> 1) & 7);
        const int m = m0 + row;
        if (m < p.M) {
            const int hw = p.Ho * p.Wo;
            const int n = m / hw, rem = m - n * hw;
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            if (STEM) {  // haloed 4-channel image: chunk j = filter rows 2j, 2j+1; a row's 8 px x 4 ch = 64 B (see conv2d_f16_launch)
                hi0[j] = 0;
                wi0[j] = 0;
                abase[j] = ((n * p.H + 2 * ho + (c >> 2)) * p.W + 2 * wo + 2 * (c & 3)) * 8;
            } else {
                hi0[j] = ho * p.stride - p.pad;
                wi0[j] = wo * p.stride - p.pad;
                abase[j] = (((n * p.H + hi0[j]) * p.W + wi0[j]) * p.Cin) * 2 + c * 16;
            }
        } else {
            hi0[j] = -(1 << 28);
            wi0[j] = 0;
            abase[j] = 0;
        }
    }
    unsigned bbase[PPB];
#pragma unroll
    for (int j = 0; j < PPB; ++j) {
        if (!loads) break;
        const int row = (lw + j * NL) * 8 + r8;
        const int c = cs ^ ((row >> 1) & 7);
        bbase[j] = (unsigned)(n0 + row) * (unsigned)(p.wrow * 2) + (unsigned)(c * 16);
    }
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    // zero-length twins: every piece of a chunk past the last one is dropped by the range check (zeros land in LDS)
    const __amdgpu_buffer_rsrc_t rs_in0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 0, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    int kr = 0, ks = 0, kc = 0, issued = 0;
    // A vector instruction issued by ANY wave of a SIMD takes issue slots from that SIMD's MFMAs (tools/microbench/mfma_switch.hip),
    // and the loader used to spend ~7 of them per A piece and 2 per B piece (93 per chunk and wave on the 192x256 tile).  Now a
    // piece is its LDS-DMA instruction alone: the lane's voffset (pixel base + tap offset, or the out-of-range constant for a
    // padding tap / a row past M) is rebuilt only when the tap changes, behind a scalar branch; the cin chunk inside the tap
    // (stem: the filter-row pair) and B's chunk offset ride in the scalar offset operand, which is outside the range check and
    // never leaves the pixel's Cin halfs / the weight row; past-the-end chunks take the zero-length descriptors (scalar selects).
    unsigned avoff[PPA];
#pragma unroll
    for (int j = 0; j < PPA; ++j) {
        if (!loads) break;
        const bool ok = STEM ? hi0[j] == 0 : ((unsigned)hi0[j] < (unsigned)p.H && (unsigned)wi0[j] < (unsigned)p.W);
        avoff[j] = ok ? (unsigned)abase[j] : OOB;
    }
    unsigned soffa = 0;

    // one 1-KiB piece (i < PPA: A rows, else B rows) of the chunk being issued; `live` = not past the last chunk
    auto piece = [&](int i, int stage, bool live) {
        char* sA = smemg + stage * STAGEB;
        if (i < PPA) {
            if (PA % NL != 0 && lw + i * NL >= PA) return;  // wave-uniform: this wave has no piece in the partial round
            // (named operands: hipcc 7.2 silently drops the host stub of the kernel when this builtin takes expressions)
            const __amdgpu_buffer_rsrc_t rs = live ? rs_in : rs_in0;
            const unsigned voff = avoff[i];
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(sA + (lw + i * NL) * 1024), 16, voff, soffa, 0, CONV_F16_A_AUX);
        } else {
            const int j = i - PPA;
            if (PB % NL != 0 && lw + j * NL >= PB) return;
            const __amdgpu_buffer_rsrc_t rs = live ? rs_w : rs_w0;
            const unsigned voff = bbase[j], soffb = (unsigned)issued * 128u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(sA + BM * 128 + (lw + j * NL) * 1024), 16, voff, soffb, 0, 0);
        }
    };
    auto advance = [&]() {
        ++issued;
        if (STEM) { soffa = (unsigned)(issued * 2 * p.W * 8); return; }
        soffa += 128u;
        if (++kc == p.cin_chunks) {  // uniform: next tap -- the only place with per-lane work
            kc = 0;
            soffa = 0;
            if (++ks == p.S) { ks = 0; ++kr; }
            int tr = __builtin_amdgcn_readfirstlane(kr), ts = __builtin_amdgcn_readfirstlane(ks);
            asm volatile("" : "+s"(tr), "+s"(ts));  // keeps the tap change behind its branch: speculated, its VALU work would run every chunk
            const int delta = ((tr * p.W + ts) * p.Cin) * 2;
#pragma unroll
            for (int j = 0; j < PPA; ++j) {
                const bool ok = (unsigned)(hi0[j] + tr) < (unsigned)p.H && (unsigned)(wi0[j] + ts) < (unsigned)p.W;
                avoff[j] = ok ? (unsigned)(abase[j] + delta) : OOB;
            }
        }
    };
    auto is_live = [&]() -> bool { return issued < p.nchunks; };  // the chunk about to be issued exists

    f32x16h acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.0f;

    const int lr = lane & 31, lh = lane >> 5;
    const int swz = lr * 128 + ((lh ^ ((lr >> 1) & 7)) << 4);  // k-step s adds ^ (s << 5)
    const int a_off = wm * TM * 32 * 128;
    const int b_off = BM * 128 + wn * TN * 32 * 128;
    constexpr int PP = PPA + PPB;
    constexpr int ISSUE_STEPS = 4;  // the next chunk's pieces are spread over this chunk's four MFMA steps

    if (loads) {
#pragma unroll
        for (int s = 0; s < NSTAGE - 1; ++s) {
            const bool live = is_live();
#pragma unroll
            for (int i = 0; i < PP; ++i) piece(i, s, live);
            advance();
        }
    }
    int wr = NSTAGE - 1;
    if (LW > 0) conv_f16_role_prio(wave >= NW);
    if (LW > 0 && wave >= NW) {  // loader wave: same barrier sequence as the MFMA waves, no matrix work
        for (int t = 0; t < p.nchunks; ++t) {
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NSTAGE - 2) * PP) : "memory");
            const bool live = is_live();
#pragma unroll
            for (int i = 0; i < PP; ++i) piece(i, wr, live);
            advance();
            wr = wr + 1 == NSTAGE ? 0 : wr + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        return;
    }
    // the chunk loop is unrolled over the ring so that the stage is a compile-time term: fragment addresses are then a
    // per-lane base plus an immediate (the per-chunk v_add of the stage offset took MFMA issue slots, see above)
    for (int t0 = 0; t0 < p.nchunks; t0 += NSTAGE) {
#pragma unroll
        for (int u = 0; u < NSTAGE; ++u) {
            if (t0 + u >= p.nchunks) break;  // uniform
            const int rd = u, wr = (u + NSTAGE - 1) % NSTAGE;
            if (LW == 0) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((NSTAGE - 2) * PP) : "memory");
            else asm volatile("s_barrier" ::: "memory");
            const bool live = LW == 0 ? is_live() : true;
            const char* sb = smemg + rd * STAGEB;
            f16x8 fa[2][TM], fb[2][TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) fa[0][a] = *(const f16x8*)(sb + a_off + a * 4096 + swz);
#pragma unroll
            for (int b = 0; b < TN; ++b) fb[0][b] = *(const f16x8*)(sb + b_off + b * 4096 + swz);
#pragma unroll
            for (int s = 0; s < 4; ++s) {  // four 16-deep MFMA steps per chunk; fragments of step s+1 and the next chunk's pieces issue under step s
                if (s < 3) {
                    const int so = swz ^ ((s + 1) << 5);
#pragma unroll
                    for (int a = 0; a < TM; ++a) fa[(s + 1) & 1][a] = *(const f16x8*)(sb + a_off + a * 4096 + so);
#pragma unroll
                    for (int b = 0; b < TN; ++b) fb[(s + 1) & 1][b] = *(const f16x8*)(sb + b_off + b * 4096 + so);
                }
#pragma unroll
                for (int i = 0; i < PP; ++i)
                    if (LW == 0 && i * ISSUE_STEPS / PP == s) piece(i, wr, live);
#pragma unroll
                for (int a = 0; a < TM; ++a)
#pragma unroll
                    for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s & 1][a], fb[s & 1][b], acc[a][b], 0, 0, 0);
            }
            if (LW == 0) advance();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // trailing all-OOB pieces have landed; LDS is free

    conv_f16_epilogue<TM, TN>(p, acc, smemg, wave, lane, wm, wn, m0, n0);
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
#ifndef ISEGMI_EPI_D
#define ISEGMI_EPI_D 2
#endif
constexpr int EPI_D = ISEGMI_EPI_D;   // residual passes in flight per lane (tools/build_variant.sh ... -DISEGMI_EPI_D=4 for an A/B)

template <int TM, int TN>
struct Epi8 {
    static_assert(TN == 2 || TN == 4, "8-row strips are 64 or 128 channels wide");
    static constexpr int LPR = TN * 4, RPP = 64 / LPR, NPASS = 8 / RPP, NQ = TM * 4 * NPASS;  // passes (b128 per lane) per wave tile
    static constexpr int D = EPI_D < NQ ? EPI_D : NQ;
    u32x4h r[D];
    unsigned rnext;   // residual offset (bytes) of the next pass to request, or >= OOB for a column past Cout; pass q covers tile rows RPP q ..
    int ux, uy, un;   // UP2X: output pixel (x, y, image) of the next pass to request
};

// UP2X (FPN top-down merge, SURVEY 8a M3: `last_inner = inner_lateral + interpolate(last_inner, scale_factor=2, mode="nearest")`): the residual of output pixel
// (n, y, x) is pixel (n, min(y >> 1, Hc - 1), min(x >> 1, Wc - 1)) of the coarser level.  A lane's pass-to-pass step is RPP pixels along the row: the walk is
// incremental (one wrap test per pass), the two divisions that start it are per tile.
template <int TM, int TN>
__device__ __forceinline__ unsigned epi8_up2x_next(const ConvKH& p, Epi8<TM, TN>& E, int co8) {
    constexpr unsigned OOB = 0x80000000u;
    int yc = E.uy >> 1, xc = E.ux >> 1;
    yc = yc > p.rHc - 1 ? p.rHc - 1 : yc;
    xc = xc > p.rWc - 1 ? p.rWc - 1 : xc;
    const unsigned off = co8 < p.Cout ? ((unsigned)((E.un * p.rHc + yc) * p.rWc + xc) * (unsigned)p.Cout + (unsigned)co8) * 2u : OOB;   // images past N are past the range
    E.ux += Epi8<TM, TN>::RPP;
    if (E.ux >= p.Wo) { E.ux -= p.Wo; E.uy += 1; if (E.uy >= p.Ho) { E.uy = 0; E.un += 1; } }
    return off;
}

// before the K loop: the first D residual passes of the wave's tile
template <int TM, int TN, bool UP2X = false>
__device__ __forceinline__ void epi8_prefetch(const ConvKH& p, Epi8<TM, TN>& E, int lane, int wm, int wn, int m0, int n0) {
    constexpr unsigned OOB = 0x80000000u;
    constexpr int LPR = Epi8<TM, TN>::LPR, RPP = Epi8<TM, TN>::RPP;
    // no branch on p.res here: a conditional prefetch makes the window a phi, and the compiler then parks the loaded registers in copies
    // behind an s_waitcnt vmcnt(0) in FRONT of the K loop; without a residual the descriptor is empty and the loads return zeros unused
    const int er = lane / LPR, ec = (lane % LPR) * 8;
    const int co8 = n0 + wn * TN * 32 + ec;
    E.rnext = co8 < p.Cout ? ((unsigned)(m0 + wm * TM * 32 + er) * (unsigned)p.Cout + (unsigned)co8) * 2u : OOB;
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res ? (const void*)p.res : (const void*)p.out), 0, p.res ? p.res_bytes : 0u, 0x00020000);
    const unsigned rstep = (unsigned)p.Cout * (2u * RPP);
    if constexpr (UP2X) {
        const int m = m0 + wm * TM * 32 + er, hw = p.Ho * p.Wo;
        E.un = m / hw;
        const int rem = m - E.un * hw;
        E.uy = rem / p.Wo;
        E.ux = rem - E.uy * p.Wo;
#pragma unroll
        for (int q = 0; q < Epi8<TM, TN>::D; ++q) E.r[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, epi8_up2x_next<TM, TN>(p, E, co8), 0, CONV_F16_RES_AUX);
        return;
    }
#pragma unroll
    for (int q = 0; q < Epi8<TM, TN>::D; ++q) { E.r[q] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, E.rnext, 0, CONV_F16_RES_AUX); E.rnext += rstep; }
}

template <int TM, int TN, bool RES, bool UP2X = false>
__device__ __forceinline__ void epi8_finish_impl(const ConvKH& p, f32x16h (&acc)[TM][TN], Epi8<TM, TN>& E, float* ew, int lane, int wm, int wn, int m0, int n0) {
    constexpr unsigned OOB = 0x80000000u;
    constexpr int PITCH = TN * 32 + 4;
    constexpr int LPR = Epi8<TM, TN>::LPR, RPP = Epi8<TM, TN>::RPP, NPASS = Epi8<TM, TN>::NPASS, NQ = Epi8<TM, TN>::NQ, D = Epi8<TM, TN>::D;
    const int lr = lane & 31, lh = lane >> 5;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc((void*)(RES ? (const void*)p.res : (const void*)p.out), 0, RES ? p.res_bytes : 0u, 0x00020000);
    const unsigned esz = p.out_f32 ? 4u : 2u;
    const int er = lane / LPR, ec = (lane % LPR) * 8;
    const int co8 = n0 + wn * TN * 32 + ec;
    const bool cok8 = co8 < p.Cout;
    const unsigned rstep = (unsigned)p.Cout * (2u * RPP);
    // contiguous destination: offset = lane base + pass * step, rows past M are out of the descriptor's range
    const unsigned ostep = (unsigned)p.out_pix_stride * esz * RPP;
    unsigned onext = cok8 ? ((unsigned)(m0 + wm * TM * 32 + er) * (unsigned)p.out_pix_stride + (unsigned)co8) * esz : OOB;
    float sc[TN], sh[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int co = n0 + (wn * TN + b) * 32 + lr;
        const bool cok = co < p.Cout;
        sc[b] = (cok && p.scale) ? p.scale[co] : 1.0f;
        sh[b] = (cok && p.shift) ? p.shift[co] : 0.0f;
    }
#pragma unroll
    for (int a = 0; a < TM; ++a) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {  // accumulator registers 4g..4g+3 = tile rows 8g + (0..3) + 4 * (lane >> 5)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j) ew[(j + 4 * lh) * PITCH + b * 32 + lr] = fmaf(acc[a][b][4 * g + j], sc[b], sh[b]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const int q = (a * 4 + g) * NPASS + ps;
                const int rr = ps * RPP + er;
                const f32x4h v0 = *(const f32x4h*)(ew + rr * PITCH + ec);
                const f32x4h v1 = *(const f32x4h*)(ew + rr * PITCH + ec + 4);
                unsigned ooff;
                if (p.contiguous) { ooff = onext; onext += ostep; }
                else {
                    const int m = m0 + wm * TM * 32 + q * RPP + er;
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
                        if (q + D < NQ) E.r[q % D] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, epi8_up2x_next<TM, TN>(p, E, co8), 0, CONV_F16_RES_AUX);
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

template <int TM, int TN, bool UP2X = false>
__device__ __forceinline__ void epi8_finish(const ConvKH& p, f32x16h (&acc)[TM][TN], Epi8<TM, TN>& E, float* ew, int lane, int wm, int wn, int m0, int n0) {
    if constexpr (UP2X) { epi8_finish_impl<TM, TN, true, true>(p, acc, E, ew, lane, wm, wn, m0, n0); return; }
    if (p.res) epi8_finish_impl<TM, TN, true>(p, acc, E, ew, lane, wm, wn, m0, n0);   // uniform
    else epi8_finish_impl<TM, TN, false>(p, acc, E, ew, lane, wm, wn, m0, n0);
}

template <int BM, int BN, int WM, int WN, int NSTAGE, int OCC, int LW, bool UP2X = false>
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
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int PA = BM / 8, PB = BN / 8;
    constexpr bool UNEVEN = (PA % NL != 0) || (PB % NL != 0);
    static_assert(!UNEVEN || NSTAGE == 2, "a partial piece round changes a wave's vmcnt count: only with the vmcnt(0) ring");
    constexpr int STAGEB = (BM + BN) * 128;
    static_assert(NW * 8 * (TN * 32 + 4) * 4 <= STAGEB, "the epilogue's 8-row strips must fit one ring stage");
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
        conv_f16_persist_loader<BM, BN, NW, NL, NSTAGE, 0>(kp0, smemg, wave, lane, bid, G, total, my_tiles, p_ntiles);
        return;
    }

    // ---------------- MFMA waves (K loop: tile geometry only; the epilogue reads its arguments per tile)
    const int wm = wave / WN, wn = wave % WN;
    const int lr = lane & 31, lh = lane >> 5;
    const int swz = lr * 128 + ((lh ^ ((lr >> 1) & 7)) << 4);  // k-step s adds ^ (s << 5)
    const int a_off = wm * TM * 32 * 128;
    const int b_off = BM * 128 + wn * TN * 32 * 128;
    f32x16h acc[TM][TN];
    auto chunk = [&](int rd) {  // run-time stage: branching over compile-time stages made hipcc copy and spill the accumulators
        const char* sb = smemg + rd * STAGEB;
        f16x8 fa[2][TM], fb[2][TN];
#pragma unroll
        for (int a = 0; a < TM; ++a) fa[0][a] = *(const f16x8*)(sb + a_off + a * 4096 + swz);
#pragma unroll
        for (int b = 0; b < TN; ++b) fb[0][b] = *(const f16x8*)(sb + b_off + b * 4096 + swz);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s < 3) {
                const int so = swz ^ ((s + 1) << 5);
#pragma unroll
                for (int a = 0; a < TM; ++a) fa[(s + 1) & 1][a] = *(const f16x8*)(sb + a_off + a * 4096 + so);
#pragma unroll
                for (int b = 0; b < TN; ++b) fb[(s + 1) & 1][b] = *(const f16x8*)(sb + b_off + b * 4096 + so);
            }
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s & 1][a], fb[s & 1][b], acc[a][b], 0, 0, 0);
        }
    };
    int st = 0;
    for (int v = bid; v < total; v += G) {
        int m0, n0;
        tile_origin(v, m0, n0);
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.0f;
        Epi8<TM, TN> E;
        {
            karg_t k1 = kp0;
            asm volatile("" : "+s"(k1));
            const ConvKH& p = *(const ConvKH*)k1;
            if (p.vec_epi) epi8_prefetch<TM, TN, UP2X>(p, E, lane, wm, wn, m0, n0);  // the tile's first residual strips travel under its K loop
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
        if (p.vec_epi) epi8_finish<TM, TN, UP2X>(p, acc, E, (float*)(smemg + last * STAGEB) + wave * 8 * (TN * 32 + 4), lane, wm, wn, m0, n0);
        else conv_f16_epilogue<TM, TN>(p, acc, smemg, wave, lane, wm, wn, m0, n0);  // per-element path: no LDS
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// FUSED 1x1 HEAD (round 4; RPNHead under configs[4]: `t = relu(conv3x3(x)); logits, deltas = cls_logits(t), bbox_pred(t)`).  On a 192 x 256
// row-strip tile the block holds ALL 256 channels of its 192 pixels, so the 1x1 convolution that follows (256 -> f_cout <= 32 outputs: 3 objectness + 12
// box deltas) can run on the tile before it ever leaves the CU: the BN + ReLU result goes to LDS as fp16 -- rounded exactly where the two-launch path
// rounds it when it stores t -- in a row-major [pixel][256] image (16-B columns XOR-swizzled by the row), the head's packed weights (32 rows) next to
// it, and six waves multiply one 32-pixel tile each on the same 16 k-steps the stand-alone 1x1 launch would walk.  t is never written (275 MB per
// P2 level at R101 bs=8) nor read back.  Bit-identical to the two launches (tests/test_rpn_head_f16_gpu.py).
template <int TM, int TN>
__device__ __forceinline__ void conv_f16_epilogue_head(const ConvKH& p, f32x16h (&acc)[TM][TN], char* smemg, int wave, int lane, int wm, int wn, int m0) {
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
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            const int co = (wn * TN + b) * 32 + lr;
            const float sc = p.scale ? p.scale[co] : 1.0f, sh = p.shift ? p.shift[co] : 0.0f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (wm * TM + a) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
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
template <int BM, int BN, int WM, int WN, int LW, int NB = 2>
__global__ __launch_bounds__((WM * WN + LW) * 64, 1) void conv3x3_f16_strip_kernel(const ConvKH p) {
    static_assert(NB == 2 || (NB == 3 && LW > 0 && (BN / 8) % LW == 0), "the three-buffer form needs loader waves and whole B piece rounds");
    constexpr int NW = WM * WN;
    constexpr int NL = LW > 0 ? LW : NW;            // waves that issue loads (LW > 0: dedicated loader waves, see above)
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
#include "conv_f16_strip_state.inc"   // tile geometry, loader state, issue_strip / next_group / issue_b / next_b: shared with the other MFMA shape's translation unit

    f32x16h acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.0f;

    // ---- fragment addressing: A rows are strip rows j(m) + s - 1 (per lane, per 32-row tile a, per tap s)
    const int lr = lane & 31, lh = lane >> 5;
    int abase_s[TM][3];
#pragma unroll
    for (int a = 0; a < TM; ++a) {
        const int m = m0 + (wm * TM + a) * 32 + lr;
        const int jm = (m - m0) + 1 + 2 * (m / W - row0);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int row = jm + s - 1;
            abase_s[a][s] = row * 128 + ((lh ^ ((row >> 1) & 7)) << 4);  // k-step ks adds ^ (ks << 5)
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
                f16x8 fa[2][TM], fb[2][TN];
#pragma unroll
                for (int a = 0; a < TM; ++a) fa[0][a] = *(const f16x8*)(sa + abase_s[a][s]);
#pragma unroll
                for (int b = 0; b < TN; ++b) fb[0][b] = *(const f16x8*)(sb + b * 4096 + swzb);
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    if (ks < 3) {
#pragma unroll
                        for (int a = 0; a < TM; ++a) fa[(ks + 1) & 1][a] = *(const f16x8*)(sa + (abase_s[a][s] ^ ((ks + 1) << 5)));
#pragma unroll
                        for (int b = 0; b < TN; ++b) fb[(ks + 1) & 1][b] = *(const f16x8*)(sb + b * 4096 + (swzb ^ ((ks + 1) << 5)));
                    }
                    if (LW == 0 && ks == 0) { issue_b(ub ^ 1); next_b(); }
                    if (LW == 0 && ks == 1) issue_strip(s, gg ^ 1);
#pragma unroll
                    for (int a = 0; a < TM; ++a)
#pragma unroll
                        for (int b = 0; b < TN; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[ks & 1][a], fb[ks & 1][b], acc[a][b], 0, 0, 0);
                }
            }
            if (LW == 0) next_group();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    STRIP_TRACE(63, 0);
    if constexpr (BM == 192 && BN == 256 && WM == 3 && WN == 4 && NB == 3) {
        if (p.f_w) { conv_f16_epilogue_head<TM, TN>(p, acc, smemg, wave, lane, wm, wn, m0); return; }   // uniform
    }
    conv_f16_epilogue<TM, TN>(p, acc, smemg, wave, lane, wm, wn, m0, n0);
    STRIP_TRACE(63, 1);
#ifdef ISEGMI_STRIP_TRACE
    if (bid == 0 && p.trace && wave == 0 && lane == 0) {
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        p.trace[4096] = (unsigned)(c1 - tr_c0); p.trace[4097] = (unsigned)(r1 - tr_r0);
    }
    if (bid == 0 && p.trace) for (int i = lane; i < 256; i += 64) p.trace[wave * 256 + i] = *(volatile unsigned*)(smemg + TRACE_OFF + (wave * 256 + i) * 4);
#endif
}

template <int BM, int BN, int WM, int WN, int LW = 0, int NB = 2>
static int launch_strip(ConvKH& k, hipStream_t st) {
    k.mtiles = cdiv(k.M, BM);
    k.ntiles = cdiv(k.Cout, BN);
    constexpr int NW = WM * WN, TN = BN / WN / 32;
    size_t lds = 2 * (size_t)(BM + 64) * 128 + NB * (size_t)BN * 128;
    static_assert(2 * (BM + 64) * 128 + NB * BN * 128 <= 163840, "LDS");
    const size_t epi = (size_t)NW * 32 * (TN * 32 + 4) * 4;
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
    LDS_LIMIT_ONCE((int)lds_attr, conv3x3_f16_strip_kernel<BM, BN, WM, WN, LW, NB>);
    hipLaunchKernelGGL((conv3x3_f16_strip_kernel<BM, BN, WM, WN, LW, NB>), dim3((unsigned)(k.mtiles * k.ntiles)), dim3((NW + LW) * 64), lds, st, k);
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

template <int BM, int BN, int WM, int WN, int NSTAGE, int OCC, bool STEM = false, int LW = 0>
static int launch_g(ConvKH& k, hipStream_t st) {
    k.mtiles = cdiv(k.M, BM);
    k.ntiles = cdiv(k.Cout, BN);
    constexpr int NW = WM * WN, TN = BN / WN / 32;
    size_t lds = (size_t)NSTAGE * (BM + BN) * 128;
    const size_t epi = (size_t)NW * 32 * (TN * 32 + 4) * 4;
    if (epi > lds) lds = epi;
    LDS_LIMIT_ONCE((int)lds, conv_f16_glds_kernel<BM, BN, WM, WN, NSTAGE, OCC, STEM, LW>);
    hipLaunchKernelGGL((conv_f16_glds_kernel<BM, BN, WM, WN, NSTAGE, OCC, STEM, LW>), dim3((unsigned)(k.mtiles * k.ntiles)), dim3((NW + LW) * 64), lds, st, k);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}

// persistent loader-wave kernel: at most one block per CU slot, each walking tiles bid, bid + grid, ...; `few` (test hook) forces
// an 8-block grid so that small test shapes exercise the multi-tile stream
template <int BM, int BN, int WM, int WN, int NSTAGE, int OCC, int LW, bool UP2X = false>
static int launch_p(ConvKH& k, hipStream_t st, bool few) {
    k.mtiles = cdiv(k.M, BM);
    k.ntiles = cdiv(k.Cout, BN);
    constexpr int NW = WM * WN, TN = BN / WN / 32;
    size_t lds = (size_t)NSTAGE * (BM + BN) * 128;
    const size_t epi = (size_t)NW * 32 * (TN * 32 + 4) * 4;  // the per-element epilogue path uses none; kept >= the one-tile kernel's request
    (void)epi;
    LDS_LIMIT_ONCE((int)lds, conv_f16_persist_kernel<BM, BN, WM, WN, NSTAGE, OCC, LW, UP2X>);
    const int ncu = device_cu_count();
    const int64_t total = (int64_t)k.mtiles * k.ntiles;
    int64_t slots = few ? 8 : (int64_t)(ncu / 8) * 8 * OCC;  // a multiple of 8, so that a block's tiles stay on its XCD
    if (slots < 8) slots = 8;
    const unsigned grid = (unsigned)(total < slots ? total : slots);
    hipLaunchKernelGGL((conv_f16_persist_kernel<BM, BN, WM, WN, NSTAGE, OCC, LW, UP2X>), dim3(grid), dim3((NW + LW) * 64), lds, st, k);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}

// 0: v_mfma_f32_32x32x16_f16 everywhere; 1: the row-strip tile (30 -> 40) and the fused RPN head on v_mfma_f32_16x16x32_f16; 2: 1 + the persistent tiles
// (34 / 37 / 39 -> 44 / 47 / 49) and the UP2X merge; 3 (default): 1 + the 144-row forms (41 / 46) where 192-row tiles leave CUs idle in a single round
static int g_f16_mfma16 = 3;
int conv_f16_mfma_shape(int set) { if (set >= 0) g_f16_mfma16 = set > 3 ? 3 : set; return g_f16_mfma16; }
static int cout_pad_h(int Cout) { return cdiv(Cout, 128) * 128; }
static bool is_stem_h(const isegmi_conv_desc* d) { return d->Cin == 4 && d->R == 7 && d->S == 7; }

// fp16 conv: in/w/res are fp16; out is fp16, or fp32 when out_f32 (predictor heads feeding fp32 selection kernels)
// `head` != nullptr: the 3x3 conv with the fused 1x1 head of conv_f16_epilogue_head (out is then unused and may be null); *head->fused tells whether the
// launch happened -- from half a round of 192 x 256 row-strip tiles on (otherwise nothing is launched)
struct ConvHeadF16 { const void* w; const float* scale; const float* shift; float* out; int cout; bool* fused; int up_hc, up_wc; };   // up_hc > 0: not a head but the UP2X residual mode (res = the coarser level)
static int conv2d_f16_launch_impl(const isegmi_conv_desc* d, const void* in, const void* w, const float* scale, const float* shift, const void* res,
                                  void* out, int out_f32, hipStream_t st, const ConvHeadF16* head);
int conv2d_f16_launch(const isegmi_conv_desc* d, const void* in, const void* w, const float* scale, const float* shift, const void* res,
                      void* out, int out_f32, hipStream_t st) {
    return conv2d_f16_launch_impl(d, in, w, scale, shift, res, out, out_f32, st, nullptr);
}
// FPN top-down merge in the lateral conv's epilogue (UP2X, see epi8_up2x_next): out = fp16(fp16(conv1x1(x) + bias) + coarse[n, y >> 1, x >> 1]) -- the lateral
// result is rounded to fp16 before the add exactly as the two-launch path (conv, then nearest2x_add_f16) stores it, so the result is bit-identical to it
int conv2d_f16_up2x_launch(const isegmi_conv_desc* d, const void* in, const void* w, const float* scale, const float* shift, const void* coarse, int Hc, int Wc,
                           void* out, hipStream_t st) {
    ARG_CHECK(d && coarse && Hc > 0 && Wc > 0, "up2x: null / shape");
    ARG_CHECK(d->R == 1 && d->S == 1 && d->stride == 1 && d->pad == 0 && d->out_pix_stride == 0 && d->out_img_stride == 0 && d->Cout % 8 == 0 && d->W >= 8 &&
              (d->tile & 255) == 0 && ((uintptr_t)coarse & 15) == 0, "up2x: a contiguous 1x1 / 1 / 0 conv with Cout % 8 == 0, W >= 8, tile 0");
    ARG_CHECK((d->H + 1) / 2 <= Hc + 1 && (d->W + 1) / 2 <= Wc + 1, "up2x: the coarser level is about half the size");
    const ConvHeadF16 h = {nullptr, nullptr, nullptr, nullptr, 0, nullptr, Hc, Wc};
    return conv2d_f16_launch_impl(d, in, w, scale, shift, coarse, out, 0, st, &h);
}
int conv2d_f16_head_launch(const isegmi_conv_desc* d, const void* in, const void* w, const float* scale, const float* shift, const void* w2,
                           const float* scale2, const float* shift2, int cout2, float* out2, bool* fused, hipStream_t st) {
    ARG_CHECK(d && w2 && out2 && fused && cout2 > 0 && cout2 <= 32, "fused head: null / more than 32 outputs");
    *fused = false;
    if (!(d->R == 3 && d->S == 3 && d->stride == 1 && d->pad == 1 && d->Cout == 256 && d->act == 1 && (d->tile & 255) == 0 && d->out_pix_stride == 0 && d->out_img_stride == 0))
        return ISEGMI_OK;
    const ConvHeadF16 h = {w2, scale2, shift2, out2, cout2, fused, 0, 0};
    return conv2d_f16_launch_impl(d, in, w, scale, shift, nullptr, out2 /* never written: a non-null placeholder */, 0, st, &h);
}
static int conv2d_f16_launch_impl(const isegmi_conv_desc* d, const void* in, const void* w, const float* scale, const float* shift, const void* res,
                                  void* out, int out_f32, hipStream_t st, const ConvHeadF16* head) {
    ARG_CHECK(d && in && w && out, "null");
    const bool stem = is_stem_h(d);
    ARG_CHECK(stem || (d->Cin > 0 && d->Cin % 64 == 0), "fp16 conv needs Cin % 64 == 0 (or the 7x7/2 Cin=4 stem)");
    ARG_CHECK(!stem || (d->stride == 2 && d->pad == 3 && res == nullptr), "fp16 stem is 7x7 stride 2 pad 3, no residual");
    ARG_CHECK(d->act == 0 || d->act == 1, "fp16 conv supports act none/relu");
    ConvKH k;
    k.in = (const half_t*)in; k.w = (const half_t*)w; k.scale = scale; k.shift = shift; k.res = (const half_t*)res; k.out = out;
    k.N = d->N; k.H = d->H; k.W = d->W; k.Cin = d->Cin; k.Cout = d->Cout; k.R = d->R; k.S = d->S; k.stride = d->stride; k.pad = d->pad;
    k.Ho = (d->H + 2 * d->pad - d->R) / d->stride + 1;
    k.Wo = (d->W + 2 * d->pad - d->S) / d->stride + 1;
    const int64_t M64 = (int64_t)d->N * k.Ho * k.Wo;
    ARG_CHECK(M64 < (1ll << 31) - 256, "too many output pixels");
    k.M = (int)M64;
    k.cin_chunks = stem ? 1 : d->Cin / 64;
    k.nchunks = stem ? 4 : d->R * d->S * k.cin_chunks;
    k.wrow = (int64_t)k.nchunks * 64;
    // stem: `in` is the haloed image [N][H+6][(W+7)&~1][4] written by pad_c3_to_f16_halo (3 zero pixels on every side), so a
    // filter row's taps wi0..wi0+7 of an output pixel are one aligned 64-B run and need no bounds test; K is laid out as
    // 8 rows x 8 taps x 4 channels = 256 with zero weights at tap 7 / row 7 / channel 3 (4 chunks of two filter rows).
    if (stem) { k.H = d->H + 6; k.W = (d->W + 7) & ~1; }
    const int64_t in_bytes = stem ? (int64_t)d->N * k.H * k.W * 8 : (int64_t)d->N * d->H * d->W * d->Cin * 2;
    ARG_CHECK(in_bytes < (1ll << 31), "conv input must be < 2 GiB");
    k.in_bytes = (unsigned)in_bytes;
    k.act = d->act; k.out_f32 = out_f32;
    k.out_div = d->out_div > 0 ? d->out_div : k.Ho * k.Wo;
    k.out_pix_stride = d->out_pix_stride > 0 ? d->out_pix_stride : d->Cout;
    k.out_img_stride = d->out_img_stride > 0 ? d->out_img_stride : (int64_t)k.out_div * k.out_pix_stride;
    k.contiguous = (k.out_img_stride == (int64_t)k.out_div * k.out_pix_stride) ? 1 : 0;
    const int64_t n_img = (k.M + k.out_div - 1) / k.out_div;
    const int64_t out_extent = ((n_img - 1) * k.out_img_stride + (int64_t)(k.out_div - 1) * k.out_pix_stride + d->Cout) * (out_f32 ? 4 : 2);
    ARG_CHECK(out_extent < (1ll << 31), "conv output span must be < 2 GiB");
    k.out_bytes = (unsigned)out_extent;
    k.res_bytes = (unsigned)((int64_t)k.M * d->Cout * 2);
    k.w_bytes = (unsigned)((int64_t)cout_pad_h(d->Cout) * k.wrow * 2);
    const int64_t align_mask = out_f32 ? 3 : 7;
    k.vec_epi = (d->Cout % 8 == 0 && (k.out_pix_stride & align_mask) == 0 && (k.out_img_stride & align_mask) == 0 && ((uintptr_t)out & 15) == 0 &&
                 (res == nullptr || ((uintptr_t)res & 15) == 0)) ? 1 : 0;
    int tile = d->tile;
    ARG_CHECK(kExperimentFlags || (tile & ~(255 | 2048)) == 0, "conv tile: bits 256 / 512 / 1024 / 4096 are timing-only experiments (-DISEGMI_EXPERIMENT_FLAGS builds only)");
    if (tile & 256) { k.in_bytes = 0; k.w_bytes = 0; }
    if (tile & 512) k.in_bytes = 0;
    if (tile & 1024) k.w_bytes = 0;
    k.dbg = (tile & 4096) ? 1 : 0;
    k.f_w = nullptr; k.f_scale = nullptr; k.f_shift = nullptr; k.f_out = nullptr; k.f_cout = 0; k.f_out_bytes = 0;
    const bool few = (tile & 2048) != 0;  // TEST HOOK: persistent kernels run on an 8-block grid (multi-tile blocks on small shapes)
    tile &= 255;  // (bits 256 / 512 / 1024: TIMING-ONLY experiments, every A / B load dropped by the range check)
    if (stem) {
        if (tile == 8) return launch_g<128, 64, 2, 2, 3, 2, true>(k, st);
        return launch_g<64, 64, 2, 2, 3, 3, true>(k, st);
    }
    if (tile == 0) {
        // Cost model: time ~ (blocks on the busiest CU) x BM x BN x (K / eff + epi) -- eff = the tile's relative MFMA efficiency (a 64x64
        // tile cannot exceed ~1/2 of the MFMA rate: 32 FLOP per L2 byte), epi = its per-tile fixed cost in K units (ring fill + epilogue:
        // small for the persistent kernels, which overlap both with the neighbouring tiles).  Tiles that share a CU (occ > 1) are assumed
        // packed onto as few CUs as the dispatcher may choose.  Round 2: persistent forms of the loader-wave tiles (32-39), and (eff, epi)
        // refitted on an in-model sweep of R101 bs 8 and R50 bs 2 with every generic tile forced in turn (tools/conv_tile_sweep.py ... f16
        // -> profiles/r02_conv_f16_tile_sweep.txt).  Same-box A/B in bench.py, R101 bs 8, two runs each: round-1 table 815 / 813 img/s
        // (conv 8.98 ms per step), persistent ids with the round-1 parameters 855 / 853 (8.53), refit 865 / 864 (8.33 ms = 656 TF/s).
        // The non-persistent big tiles 1 / 2 / 7 / 9 / 11 / 12-14 / 17 / 19 left the candidate list.
        static const struct { int id, bm, bn, occ; double eff, epi; } T[] = {
            {3, 128, 128, 2, 0.613, 176}, {4, 64, 64, 3, 0.642, 106}, {5, 64, 128, 3, 0.751, 79}, {6, 64, 256, 2, 0.452, 266}, {10, 192, 128, 2, 0.772, 91},
            {16, 160, 256, 1, 0.529, 200}, {20, 192, 128, 2, 0.300, 60},
            // persistent, 4 loader waves (32: 8 MFMA waves of 96x64; 37: 12 of 64x64)
            {32, 192, 256, 1, 0.929, 447}, {34, 256, 128, 1, 1.043, 127}, {37, 192, 256, 1, 1.197, 83}, {39, 128, 256, 1, 1.032, 99},
            // row-strip variants (3x3 / stride 1 / pad 1 only: ~2.7x fewer A bytes through the CU's fill path) + 4 loader waves; measured
            // level with tile 37 on every 3x3 layer of the sweep
            {26, 192, 256, 1, 1.18, 83}, {27, 256, 128, 1, 1.03, 127}, {28, 160, 256, 1, 0.5, 200}, {29, 192, 256, 1, 1.21, 83},
            {30, 192, 256, 1, 1.24, 83}};  // 29 with three B buffers (round 4): +3 % on the big layers, never slower (tools/conv_f16_bench.py)
        const bool strip_ok = d->R == 3 && d->S == 3 && d->stride == 1 && d->pad == 1;
        double best = 0.0;
        for (const auto& t : T) {
            if (t.id >= 26 && t.id <= 31 && !(strip_ok && (t.bm - 1) / d->W + 2 <= 32)) continue;
            const int64_t blocks = (int64_t)cdiv(k.M, t.bm) * cdiv(d->Cout, t.bn);
            int64_t per_cu = (blocks + 255) / 256;
            if (t.occ > 1 && blocks <= 256 * t.occ) per_cu = blocks < t.occ ? blocks : t.occ;
            const double c = (double)per_cu * t.bm * t.bn * ((double)k.nchunks * 64.0 / t.eff + t.epi);
            if (tile == 0 || c < best) { best = c; tile = t.id; }
        }
    }
    k.res_up2x = 0; k.rHc = 0; k.rWc = 0;
    if (head && head->up_hc > 0) {
        ARG_CHECK(k.vec_epi, "up2x needs 16-byte aligned tensors");
        const int64_t rb = (int64_t)d->N * head->up_hc * head->up_wc * d->Cout * 2;
        ARG_CHECK(rb < (1ll << 31), "up2x: coarse level must be < 2 GiB");
        k.res_up2x = 1; k.rHc = head->up_hc; k.rWc = head->up_wc; k.res_bytes = (unsigned)rb;
        if (g_f16_mfma16 == 2) return conv_f16_m16_launch(48, k, st, few);
        return launch_p<192, 256, 3, 4, 2, 1, 4, true>(k, st, few);   // "tile 38": tile 37 with the walked residual
    }
    if (head) {
        // fused where the 192 x 256 row-strip tile is usable and the layer has at least half a round of such tiles (the fusion saves the 1x1 launch and t's
        // round trip, worth more than the tile quantisation the cost model might avoid with another tile); smaller levels: the caller runs the two launches
        if (!(191 / d->W + 2 <= 32) || cdiv(k.M, 192) < 128) return ISEGMI_OK;
        const int64_t ob = (int64_t)k.M * head->cout * 4;
        ARG_CHECK(ob < (1ll << 31), "fused head output must be < 2 GiB");
        k.f_w = (const half_t*)head->w; k.f_scale = head->scale; k.f_shift = head->shift; k.f_out = head->out; k.f_cout = head->cout; k.f_out_bytes = (unsigned)ob;
        *head->fused = true;
        tile = g_f16_mfma16 >= 1 ? 40 : 30;
    }
    // the MFMA shape of the tiles that carry the fp16 backbone: 30 / 34 / 37 / 39 run on v_mfma_f32_32x32x16_f16 (this file), 40 / 44 / 47 / 49 are the same tiles
    // on v_mfma_f32_16x16x32_f16 (csrc/conv_mfma_f16_m16.hip).  The cost model chooses among the former; g_f16_mfma16 (isegmi_set_f16_mfma_shape) maps its
    // choice onto the latter: 1 the row-strip tile and the fused head that lives on it -- +8 % on the 634-GF layer (983 -> 1062 TF/s, same box;
    // profiles/r05_experiments.txt 1) -- 2 the persistent tiles and the UP2X merge as well (memory-bound layers: level or 1-3 % slower, kept for A/B), 3 (default)
    // = 1 + the 144-row forms below.
    // Results do not depend on the shape: one 16 x 16 x 32 instruction sums its 32 products exactly as two chained 32 x 32 x 16 instructions do
    // (tools/microbench/mfma_shape.hip: 0 of 204 800 elements differ).
    if (g_f16_mfma16 >= 1 && (d->tile & 255) == 0 && tile == 30) tile = 40;
    if (g_f16_mfma16 == 2 && (d->tile & 255) == 0) tile = tile == 34 ? 44 : tile == 37 ? 47 : tile == 39 ? 49 : tile;
    // 144-row forms (16 x 16 blocks allow 48-row wave tiles): where the 192-row tiles of a Cout <= 256 layer leave CUs idle in their last round and
    // 144-row tiles fill it better -- M = 33 600, res4 / P4 at bs 8: 175 -> 234 tiles on 256 CUs, each 3/4 of the work: 0.046 -> 0.040 ms -- the
    // row-strip tile runs as 41 and the persistent 192 x 256 tile of a 1x1 layer as 46 (three-deep ring: 3 x 51 200 B)
    if (g_f16_mfma16 == 3 && (d->tile & 255) == 0 && !head && d->Cout <= 256 && (tile == 40 || tile == 37)) {
        const int ncu = device_cu_count();
        // rounds x rows: M = 33 600 one round either way (192 -> 144 rows per CU); the mask head's M = 156 800: 817 tiles = 4 rounds of 192 against 1089 = 5
        // rounds of 144 (768 -> 720, measured +3.5 %); P3's M = 134 400 ties (576 = 576) and keeps the larger tile
        const int b192 = cdiv(k.M, 192), b144 = cdiv(k.M, 144);
        if (cdiv(b144, ncu) * 144 < cdiv(b192, ncu) * 192 && (tile != 40 || 143 / d->W + 2 <= 32)) tile = tile == 40 ? 41 : 46;
    }
    if (tile == 40 || tile == 41) ARG_CHECK(d->R == 3 && d->S == 3 && d->stride == 1 && d->pad == 1 && (tile == 40 ? 191 : 143) / d->W + 2 <= 32, "strip tiles are for 3x3 / stride 1 / pad 1, <= 32 image-row segments");
    if (tile == 40 || tile == 41 || tile == 44 || tile == 46 || tile == 47 || tile == 49) return conv_f16_m16_launch(tile, k, st, few);
    if (tile >= 26 && tile <= 31) {
        ARG_CHECK(d->R == 3 && d->S == 3 && d->stride == 1 && d->pad == 1, "strip tiles are for 3x3 / stride 1 / pad 1");
        const int bm = tile == 27 ? 256 : tile == 28 ? 160 : 192;
        ARG_CHECK((bm - 1) / d->W + 2 <= 32, "strip tile: too many image-row segments (W too small)");
        switch (tile) {  // row strips + 4 loader waves
            case 26: return launch_strip<192, 256, 2, 4, 4>(k, st);
            case 27: return launch_strip<256, 128, 4, 2, 4>(k, st);
            case 29: return launch_strip<192, 256, 3, 4, 4>(k, st);  // 12 MFMA waves (64x64 each) + 4 loader waves
            case 30: return launch_strip<192, 256, 3, 4, 4, 3>(k, st);  // the same with three B buffers and a counted wait (round 4)
            case 31: return launch_strip<192, 256, 2, 4, 4, 3>(k, st);  // 8 MFMA waves (96x64 each), three B buffers
            default: return launch_strip<160, 256, 1, 8, 4>(k, st);
        }
    }
    switch (tile) {
        case 1: return launch_g<256, 256, 2, 4, 2, 1>(k, st);  // 8 waves, wave tile 128x64
        case 2: return launch_g<256, 128, 4, 2, 3, 1>(k, st);  // 8 waves, wave tile 64x64, 3-deep ring
        case 3: return launch_g<128, 128, 2, 2, 2, 2>(k, st);
        case 4: return launch_g<64, 64, 2, 2, 3, 3>(k, st);
        case 5: return launch_g<64, 128, 1, 4, 2, 3>(k, st);
        case 6: return launch_g<64, 256, 1, 4, 2, 2>(k, st);
        case 7: return launch_g<128, 256, 2, 4, 3, 1>(k, st);
        case 8: return launch_g<128, 64, 2, 2, 3, 2>(k, st);
        case 9: return launch_g<192, 256, 2, 4, 2, 1>(k, st);   // wave tile 96x64
        case 10: return launch_g<192, 128, 2, 2, 2, 2>(k, st);
        case 11: return launch_g<160, 256, 1, 8, 2, 1>(k, st);
        case 12: return launch_g<192, 256, 2, 4, 2, 1, false, 4>(k, st);  // 8 MFMA waves + 4 loader waves
        case 13: return launch_g<256, 256, 2, 4, 2, 1, false, 4>(k, st);
        case 14: return launch_g<256, 128, 4, 2, 3, 1, false, 4>(k, st);
        case 16: return launch_g<160, 256, 1, 8, 2, 1, false, 4>(k, st);
        case 17: return launch_g<192, 256, 3, 4, 2, 1, false, 4>(k, st);  // 12 MFMA waves (64x64 each) + 4 loader waves
        case 19: return launch_g<128, 256, 2, 4, 3, 1, false, 4>(k, st);  // 128x256, 3-deep ring, 8 MFMA + 4 loader waves
        case 20: return launch_g<192, 128, 3, 2, 2, 2, false, 2>(k, st);  // 6 MFMA + 2 loader waves, 2 blocks/CU  // wave tile 160x32; 20 A pieces over 8 waves (partial round)
        // persistent forms of 12 / 14 / 17 / 19 (loader waves stream the next tile during this tile's epilogue; 13's 256x256 form was dropped in
        // round 3: 128 accumulator registers + the residual window spill, and the cost model never chose it); the two-blocks-
        // per-CU tile 20 has no persistent form: 128 registers per wave do not hold its accumulators next to the tile loop (it spilled)
        case 32: return launch_p<192, 256, 2, 4, 2, 1, 4>(k, st, few);
        case 34: return launch_p<256, 128, 4, 2, 3, 1, 4>(k, st, few);
        case 37: return launch_p<192, 256, 3, 4, 2, 1, 4>(k, st, few);
        case 39: return launch_p<128, 256, 2, 4, 3, 1, 4>(k, st, few);
        default: break;
    }
    ARG_CHECK(false, "unknown fp16 conv tile");
}

}  // namespace isegmi

using namespace isegmi;

extern "C" int isegmi_conv_packed_halfs(const isegmi_conv_desc* d, int64_t* n) {
    ARG_CHECK(d && n && d->Cin > 0 && (d->Cin % 64 == 0 || is_stem_h(d)), "fp16 pack needs Cin % 64 == 0 (or the stem)");
    *n = is_stem_h(d) ? (int64_t)cout_pad_h(d->Cout) * 256 : (int64_t)cout_pad_h(d->Cout) * d->R * d->S * d->Cin;
    return ISEGMI_OK;
}

// host: natural fp32 [Cout][R][S][Cin] -> fp16 [Cout padded to 128][R*S*Cin] (round to nearest even)
extern "C" int isegmi_pack_conv_weights_f16(const isegmi_conv_desc* d, const float* w, uint16_t* packed) {
    ARG_CHECK(d && w && packed && d->Cin > 0 && (d->Cin % 64 == 0 || is_stem_h(d)), "fp16 pack needs Cin % 64 == 0 (or the stem)");
    if (is_stem_h(d)) {  // [Cout][7][7][4] -> [Cout padded][8 rows][8 taps][4 ch], zero at row 7 / tap 7
        const int64_t total = (int64_t)cout_pad_h(d->Cout) * 256;
        for (int64_t i = 0; i < total; ++i) packed[i] = 0;
        for (int co = 0; co < d->Cout; ++co)
            for (int r = 0; r < 7; ++r)
                for (int t = 0; t < 7; ++t)
                    for (int c = 0; c < 4; ++c) {
                        const half_t h = (half_t)w[(((int64_t)co * 7 + r) * 7 + t) * 4 + c];
                        packed[(int64_t)co * 256 + r * 32 + t * 4 + c] = __builtin_bit_cast(uint16_t, h);
                    }
        return ISEGMI_OK;
    }
    const int64_t K = (int64_t)d->R * d->S * d->Cin;
    const int64_t total = (int64_t)cout_pad_h(d->Cout) * K;
    for (int64_t i = 0; i < total; ++i) packed[i] = 0;
    for (int co = 0; co < d->Cout; ++co)
        for (int64_t k = 0; k < K; ++k) {
            const half_t h = (half_t)w[(int64_t)co * K + k];
            packed[(int64_t)co * K + k] = __builtin_bit_cast(uint16_t, h);
        }
    return ISEGMI_OK;
}

extern "C" int isegmi_op_conv1x1_up2x_add_f16(const isegmi_conv_desc* d, const void* d_in, const void* d_wpacked, const float* d_scale, const float* d_shift,
                                              const void* d_coarse, int Hc, int Wc, void* d_out, void* stream) {
    return conv2d_f16_up2x_launch(d, d_in, d_wpacked, d_scale, d_shift, d_coarse, Hc, Wc, d_out, (hipStream_t)stream);
}

extern "C" int isegmi_op_conv3x3_head_f16(const isegmi_conv_desc* d, const void* d_in, const void* d_wpacked, const float* d_scale, const float* d_shift,
                                          const void* d_w2packed, const float* d_scale2, const float* d_shift2, int cout2, float* d_out2, int* fused, void* stream) {
    ARG_CHECK(fused, "null");
    bool f = false;
    const int rc = conv2d_f16_head_launch(d, d_in, d_wpacked, d_scale, d_shift, d_w2packed, d_scale2, d_shift2, cout2, d_out2, &f, (hipStream_t)stream);
    *fused = f ? 1 : 0;
    return rc;
}

extern "C" int isegmi_op_conv2d_f16(const isegmi_conv_desc* d, const void* d_in, const void* d_wpacked, const float* d_scale,
                                    const float* d_shift, const void* d_residual, void* d_out, int out_f32, void* stream) {
    return conv2d_f16_launch(d, d_in, d_wpacked, d_scale, d_shift, d_residual, d_out, out_f32, (hipStream_t)stream);
}

extern "C" int isegmi_set_f16_mfma_shape(int shape) {
    ARG_CHECK(shape >= 0 && shape <= 3, "0: v_mfma_f32_32x32x16_f16 everywhere, 1: row strips on v_mfma_f32_16x16x32_f16, 2: persistent tiles too, 3: 1 + 144-row forms");
    conv_f16_mfma_shape(shape);
    return ISEGMI_OK;
}
extern "C" int isegmi_get_f16_mfma_shape(int* shape) {
    ARG_CHECK(shape, "null");
    *shape = conv_f16_mfma_shape(-1);
    return ISEGMI_OK;
}
