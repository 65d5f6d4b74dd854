/*
 * oracle/ora_ops.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the RoI/mask inference hot path of detectron.jittor's
 * Mask R-CNN and Yolact.jittor, op by op, in fp32 with a fully specified
 * rounding sequence so that a GPU implementation can be compared BIT-EXACTLY.
 *
 * PARITY UNPINNED: /root/reference holds no source for this path (the two
 * submodules are empty directories) and ships no tests or
 * golden vectors.  The only citeable anchors are README lines; the algorithms
 * below follow SURVEY.md Appendix A (a recall of the public lineage the
 * reference names at README.md:353-358: maskrcnn-benchmark and dbolya/yolact).
 * Each function cites the Appendix-A item and the README anchor that reaches it.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product (libisegmi.so) never links or calls it.
 *
 * Arithmetic conventions (shared by contract with the HIP kernels):
 *   - every multiply-accumulate chain is a k-ordered fmaf() chain starting at +0
 *     (this is what v_mfma_f32_32x32x2_f32 computes, bit for bit);
 *   - everything else is individually rounded IEEE fp32 ( + - * / sqrt ),
 *     compiled with -ffp-contract=off;
 *   - exp / tanh / log2 are the Cephes single-precision polynomials written out
 *     with explicit fmaf (ora_expf ...), not libm, so they are reproducible;
 *   - all sorts are total orders: (score descending, index ascending).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef __AVX2__
#include <immintrin.h>
#endif

#define ORA_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ math -- */
static inline float bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* Cephes expf.  Range: x<-87.3 -> 0, x clamped above at 88.376. */
ORA_API float ora_expf(float x) {
    if (x != x) return x;
    if (x > 88.3762626647949f) x = 88.3762626647949f;
    if (x < -87.3f) return 0.0f;
    float fx = floorf(fmaf(x, 1.44269504088896341f, 0.5f));
    float r = fmaf(fx, -0.693359375f, x);
    r = fmaf(fx, 2.12194440e-4f, r);
    float z = r * r;
    float p = 1.9875691500E-4f;
    p = fmaf(p, r, 1.3981999507E-3f);
    p = fmaf(p, r, 8.3334519073E-3f);
    p = fmaf(p, r, 4.1665795894E-2f);
    p = fmaf(p, r, 1.6666665459E-1f);
    p = fmaf(p, r, 5.0000001201E-1f);
    float y = fmaf(p, z, r);
    y = y + 1.0f;
    int n = (int)fx;
    return y * bits2f((uint32_t)(n + 127) << 23);
}

ORA_API float ora_sigmoidf(float x) {
    float e = ora_expf(-x);
    return 1.0f / (1.0f + e);
}

/* Cephes tanhf. */
ORA_API float ora_tanhf(float x) {
    float z = fabsf(x);
    if (z >= 0.625f) {
        float r;
        if (z > 44.0f) r = 1.0f;
        else {
            float e = ora_expf(z + z);
            r = 1.0f - 2.0f / (e + 1.0f);
        }
        return x < 0.0f ? -r : r;
    }
    float s = x * x;
    float p = -5.70498872745E-3f;
    p = fmaf(p, s, 2.06390887954E-2f);
    p = fmaf(p, s, -5.37397155531E-2f);
    p = fmaf(p, s, 1.33314422036E-1f);
    p = fmaf(p, s, -3.33332819422E-1f);
    float t = p * s;
    return fmaf(t, x, x);
}

/* Cephes logf core, returned as log2.  x must be a positive normal float. */
ORA_API float ora_log2f(float x) {
    uint32_t b = f2bits(x);
    int e = (int)((b >> 23) & 255u) - 126;
    float m = bits2f((b & 0x007fffffu) | 0x3f000000u); /* [0.5,1) */
    if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; }
    else m = m - 1.0f;
    float z = m * m;
    float p = 7.0376836292E-2f;
    p = fmaf(p, m, -1.1514610310E-1f);
    p = fmaf(p, m, 1.1676998740E-1f);
    p = fmaf(p, m, -1.2420140846E-1f);
    p = fmaf(p, m, 1.4249322787E-1f);
    p = fmaf(p, m, -1.6668057665E-1f);
    p = fmaf(p, m, 2.0000714765E-1f);
    p = fmaf(p, m, -2.4999993993E-1f);
    p = fmaf(p, m, 3.3333331174E-1f);
    float y = p * m;
    y = y * z;
    y = fmaf(-0.5f, z, y);
    float ln = m + y;
    return fmaf(ln, 1.44269504088896341f, (float)e);
}

ORA_API void ora_map_f32(const float* x, float* y, int64_t n, int fn) {
    for (int64_t i = 0; i < n; ++i) {
        float v = x[i];
        y[i] = fn == 0 ? ora_expf(v) : fn == 1 ? ora_sigmoidf(v) : fn == 2 ? ora_tanhf(v) : ora_log2f(v);
    }
}

/* ------------------------------------------------------------ conv family -- */
/* Appendix A.1 conv2d (cross-correlation, zero padding), NHWC activations,
 * weights [Cout][R][S][Cin].  acc = ONE fmaf chain from +0 over the K = R*S*Cin
 * products (padding taps contribute fmaf(0,w,acc)), walked in this order: input
 * channels in groups of ORA_CONV_CGROUP = 128 outermost, then (r, s), then the
 * group's channels ascending -- for Cin <= 128 and for every 1x1 convolution that
 * is plain (r, s, c).  (Any order is a valid fp32 evaluation of the sum; this one
 * is the contract with the HIP kernels since round 3: walking all of Cin per tap
 * made every tap re-fetch the activation window through a thrashing L2.)  Then
 *   y = fmaf(acc, scale[co], shift[co]);  y += residual;  act(y)
 * act: 0 none, 1 relu, 2 tanh.  scale==NULL means 1, shift==NULL means 0.
 * Output element (n, p=ho*Wo+wo, co) is written at
 *   out[n*out_img_stride + p*out_pix_stride + co].
 * Reached from README.md:331 (run_on_opencv_image) / README.md:243 (eval.py). */
/* Summation-order switch of ora_conv2d (default 0 = THE oracle: one k-ordered fmaf chain per output).  Mode 1 restates the same sum in
 * another, equally valid fp32 association -- the one an f16 MFMA implies: every run of 16 consecutive k is summed on its own from +0 and
 * the partial sum is then added to the accumulator.  It exists only to MEASURE how far two correct fp32 evaluations of the fp16-storage
 * network drift apart (tests/test_maskrcnn_e2e_gpu.py derives the fp16 path's tolerance from it); nothing is ever checked against mode 1
 * as if it were the truth. */
static int g_conv_sum_mode = 0;
#define ORA_CONV_CGROUP 128
ORA_API void ora_set_conv_sum_mode(int mode) { g_conv_sum_mode = mode; }

/* ksplit > 1: the FIXED-TREE SPLIT-K evaluation (round 6; the product's conv tile 15, an opt-in mode of the engines: `conv_split_k`).  The K order above is
 * a sequence of 32-channel chunks t = 0 .. R*S*Cin/32 - 1; with L = ceil(chunks / ksplit) the output is
 *     ((p_0 + p_1) + p_2) + ... ,   p_q = the k-ordered fmaf chain from +0 over the chunks [q L, (q + 1) L)
 * (individually rounded fp32 adds, left to right), then the same epilogue.  Another valid fp32 evaluation of the same sum, one that a GPU can run as
 * ksplit independent dependency chains per output.  Cin % 32 == 0. */
static void conv2d_impl(const float* in, int N, int H, int W, int Cin,
                        const float* w, int Cout, int R, int S, int stride, int pad,
                        const float* scale, const float* shift, const float* residual, int act,
                        float* out, int64_t out_img_stride, int64_t out_pix_stride, int ksplit) {
    const int split_L = (ksplit > 1 && Cin % 32 == 0) ? (R * S * (Cin / 32) + ksplit - 1) / ksplit : 0;
    const int Ho = (H + 2 * pad - R) / stride + 1;
    const int Wo = (W + 2 * pad - S) / stride + 1;
    const int K = R * S * Cin;
    const int CB = 32;
    const int Cp = (Cout + CB - 1) / CB * CB;
    float* wt = (float*)aligned_alloc(64, (size_t)K * Cp * sizeof(float));
    memset(wt, 0, (size_t)K * Cp * sizeof(float));
    for (int co = 0; co < Cout; ++co)
        for (int k = 0; k < K; ++k) wt[(size_t)k * Cp + co] = w[(size_t)co * K + k];
    float* zrow = (float*)calloc((size_t)Cin, sizeof(float));
    enum { PB = 4 };
#pragma omp parallel for collapse(2) schedule(dynamic, 4)
    for (int n = 0; n < N; ++n)
        for (int ho = 0; ho < Ho; ++ho) {
            /* channel block outside the pixel groups: one block's weights (K x 32 floats) serve a whole output row from the core's L2 instead of being
             * streamed again for every four pixels (order of evaluation only: each output's chain is untouched) */
            for (int cb = 0; cb < Cp; cb += CB) {
                for (int wo0 = 0; wo0 < Wo; wo0 += PB) {
                const int np = (Wo - wo0) < PB ? (Wo - wo0) : PB;
#ifdef __AVX2__
                    __m256 acc[PB][4], tot[PB][4];
                    for (int p = 0; p < PB; ++p)
                        for (int j = 0; j < 4; ++j) { acc[p][j] = _mm256_setzero_ps(); tot[p][j] = _mm256_setzero_ps(); }
#else
                    float acc[PB][32], tot[PB][32];
                    for (int p = 0; p < PB; ++p)
                        for (int j = 0; j < 32; ++j) { acc[p][j] = 0.0f; tot[p][j] = 0.0f; }
#endif
                    int t_chunk = 0, nflush = 0;   /* split-K: chunk counter along the K order, partial sums folded so far */
                    for (int cg = 0; cg < Cin; cg += ORA_CONV_CGROUP)
                    for (int r = 0; r < R; ++r)
                        for (int s = 0; s < S; ++s) {
                            const int cge = cg + ORA_CONV_CGROUP < Cin ? cg + ORA_CONV_CGROUP : Cin;
                            const float* rows[PB];
                            const int hi = ho * stride + r - pad;
                            for (int p = 0; p < PB; ++p) {
                                const int wo = wo0 + (p < np ? p : 0);
                                const int wi = wo * stride + s - pad;
                                rows[p] = (hi >= 0 && hi < H && wi >= 0 && wi < W)
                                              ? in + (((size_t)n * H + hi) * W + wi) * Cin
                                              : zrow;
                            }
                            const float* wk = wt + (size_t)((r * S + s) * Cin + cg) * Cp + cb;
                            if (g_conv_sum_mode == 1 && Cin % 16 == 0) {  /* 16-term partial sums (see ora_set_conv_sum_mode) */
                                for (int c0 = cg; c0 < cge; c0 += 16) {
                                    float part[PB][32];
                                    memset(part, 0, sizeof(part));
                                    for (int c = c0; c < c0 + 16; ++c, wk += Cp)
                                        for (int p = 0; p < PB; ++p) {
                                            const float a = rows[p][c];
                                            for (int j = 0; j < 32; ++j) part[p][j] = fmaf(a, wk[j], part[p][j]);
                                        }
                                    for (int p = 0; p < PB; ++p) {
#ifdef __AVX2__
                                        for (int j = 0; j < 4; ++j) acc[p][j] = _mm256_add_ps(acc[p][j], _mm256_loadu_ps(part[p] + 8 * j));
#else
                                        for (int j = 0; j < 32; ++j) acc[p][j] += part[p][j];
#endif
                                    }
                                }
                                continue;
                            }
                            for (int c = cg; c < cge; ++c, wk += Cp) {
                                if (split_L && (c & 31) == 0) {   /* a 32-channel chunk starts: fold the finished partial chain when a split range ends here */
                                    if (t_chunk > 0 && t_chunk % split_L == 0) {
                                        for (int p = 0; p < PB; ++p)
#ifdef __AVX2__
                                            for (int j = 0; j < 4; ++j) { tot[p][j] = nflush ? _mm256_add_ps(tot[p][j], acc[p][j]) : acc[p][j]; acc[p][j] = _mm256_setzero_ps(); }
#else
                                            for (int j = 0; j < 32; ++j) { tot[p][j] = nflush ? tot[p][j] + acc[p][j] : acc[p][j]; acc[p][j] = 0.0f; }
#endif
                                        ++nflush;
                                    }
                                    ++t_chunk;
                                }
#ifdef __AVX2__
                                const __m256 w0 = _mm256_load_ps(wk), w1 = _mm256_load_ps(wk + 8),
                                             w2 = _mm256_load_ps(wk + 16), w3 = _mm256_load_ps(wk + 24);
                                for (int p = 0; p < PB; ++p) {
                                    const __m256 a = _mm256_broadcast_ss(rows[p] + c);
                                    acc[p][0] = _mm256_fmadd_ps(a, w0, acc[p][0]);
                                    acc[p][1] = _mm256_fmadd_ps(a, w1, acc[p][1]);
                                    acc[p][2] = _mm256_fmadd_ps(a, w2, acc[p][2]);
                                    acc[p][3] = _mm256_fmadd_ps(a, w3, acc[p][3]);
                                }
#else
                                for (int p = 0; p < PB; ++p) {
                                    const float a = rows[p][c];
                                    for (int j = 0; j < 32; ++j) acc[p][j] = fmaf(a, wk[j], acc[p][j]);
                                }
#endif
                            }
                        }
                    if (nflush)   /* ((p0 + p1) + ...) + the last partial chain */
                        for (int p = 0; p < PB; ++p)
#ifdef __AVX2__
                            for (int j = 0; j < 4; ++j) acc[p][j] = _mm256_add_ps(tot[p][j], acc[p][j]);
#else
                            for (int j = 0; j < 32; ++j) acc[p][j] = tot[p][j] + acc[p][j];
#endif
                    for (int p = 0; p < np; ++p) {
                        float av[32];
#ifdef __AVX2__
                        for (int j = 0; j < 4; ++j) _mm256_storeu_ps(av + 8 * j, acc[p][j]);
#else
                        memcpy(av, acc[p], sizeof(av));
#endif
                        const int64_t pix = (int64_t)ho * Wo + wo0 + p;
                        float* o = out + (int64_t)n * out_img_stride + pix * out_pix_stride;
                        const float* res = residual ? residual + (((int64_t)n * Ho * Wo) + pix) * Cout : NULL;
                        for (int j = 0; j < 32 && cb + j < Cout; ++j) {
                            const int co = cb + j;
                            float y = fmaf(av[j], scale ? scale[co] : 1.0f, shift ? shift[co] : 0.0f);
                            if (act == 4) {  /* DarkNet block: LeakyReLU(0.1) first, then the shortcut */
                                y = y > 0.0f ? y : y * 0.1f;
                                if (res) y = y + res[co];
                                o[co] = y;
                                continue;
                            }
                            if (res) y = y + res[co];
                            if (act == 1) y = y > 0.0f ? y : 0.0f;
                            else if (act == 2) y = ora_tanhf(y);
                            else if (act == 3) y = y > 0.0f ? y : y * 0.1f;  /* LeakyReLU(0.1) */
                            o[co] = y;
                        }
                    }
                }
            }
        }
    free(zrow);
    free(wt);
}
ORA_API void ora_conv2d(const float* in, int N, int H, int W, int Cin,
                        const float* w, int Cout, int R, int S, int stride, int pad,
                        const float* scale, const float* shift, const float* residual, int act,
                        float* out, int64_t out_img_stride, int64_t out_pix_stride) {
    conv2d_impl(in, N, H, W, Cin, w, Cout, R, S, stride, pad, scale, shift, residual, act, out, out_img_stride, out_pix_stride, 1);
}
ORA_API void ora_conv2d_split(const float* in, int N, int H, int W, int Cin,
                              const float* w, int Cout, int R, int S, int stride, int pad,
                              const float* scale, const float* shift, const float* residual, int act,
                              float* out, int64_t out_img_stride, int64_t out_pix_stride, int ksplit) {
    conv2d_impl(in, N, H, W, Cin, w, Cout, R, S, stride, pad, scale, shift, residual, act, out, out_img_stride, out_pix_stride, ksplit);
}

/* Appendix A.1 ConvTranspose2d(k2,s2,p0): w [Cin][Cout][2][2] (upstream layout),
 * out[n,2i+a,2j+b,co] = relu?( fmaf-chain_ci(in[n,i,j,ci]*w[ci,co,a,b]) + bias[co] ),
 * chain over ci ascending from +0, then y = acc + bias. NHWC in/out. (M11) */
ORA_API void ora_deconv2x2(const float* in, int N, int H, int W, int Cin, const float* w, int Cout,
                           const float* bias, int relu, float* out) {
#pragma omp parallel for collapse(2)
    for (int n = 0; n < N; ++n)
        for (int i = 0; i < H; ++i)
            for (int j = 0; j < W; ++j) {
                const float* x = in + (((size_t)n * H + i) * W + j) * Cin;
                for (int a = 0; a < 2; ++a)
                    for (int b = 0; b < 2; ++b) {
                        float* o = out + (((size_t)n * 2 * H + 2 * i + a) * 2 * W + 2 * j + b) * Cout;
                        for (int co = 0; co < Cout; ++co) {
                            float acc = 0.0f;
                            for (int ci = 0; ci < Cin; ++ci)
                                acc = fmaf(x[ci], w[(((size_t)ci * Cout + co) * 2 + a) * 2 + b], acc);
                            float y = fmaf(acc, 1.0f, bias ? bias[co] : 0.0f);
                            if (relu) y = y > 0.0f ? y : 0.0f;
                            o[co] = y;
                        }
                    }
            }
}

/* Appendix A.1 max_pool2d(k,s,p) with -inf padding, NHWC. (M2 stem, Y2 stem;
 * k=1,s=2,p=0 is LastLevelMaxPool, M3) */
ORA_API void ora_maxpool(const float* in, int N, int H, int W, int C, int k, int s, int p, float* out) {
    const int Ho = (H + 2 * p - k) / s + 1, Wo = (W + 2 * p - k) / s + 1;
#pragma omp parallel for collapse(2)
    for (int n = 0; n < N; ++n)
        for (int ho = 0; ho < Ho; ++ho)
            for (int wo = 0; wo < Wo; ++wo) {
                float* o = out + (((size_t)n * Ho + ho) * Wo + wo) * C;
                for (int c = 0; c < C; ++c) o[c] = -INFINITY;
                for (int r = 0; r < k; ++r)
                    for (int q = 0; q < k; ++q) {
                        const int hi = ho * s + r - p, wi = wo * s + q - p;
                        if (hi < 0 || hi >= H || wi < 0 || wi >= W) continue;
                        const float* x = in + (((size_t)n * H + hi) * W + wi) * C;
                        for (int c = 0; c < C; ++c) o[c] = x[c] > o[c] ? x[c] : o[c];
                    }
            }
}

/* Appendix A.1 bilinear, align_corners=False:
 *   scale = (float)in/(float)out; src = scale*(dst+0.5)-0.5; src=max(src,0);
 *   i0=(int)src; i1=min(i0+1,in-1); l1=src-i0; l0=1-l1
 *   top = fmaf(l1x, v01, l0x*v00); bot = fmaf(l1x, v11, l0x*v10)
 *   val = fmaf(l1y, bot, l0y*top);  out = val (+ add) ; relu optional.
 * (Y3 FPN top-down, Y4 protonet x2, Y7 mask upsample, M12 paste) */
static inline void bil_coef(int dst, int in_sz, int out_sz, int* i0, int* i1, float* l0, float* l1) {
    const float scale = (float)in_sz / (float)out_sz;
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.0f) src = 0.0f;
    int a = (int)src;
    if (a > in_sz - 1) a = in_sz - 1;
    *i0 = a;
    *i1 = a < in_sz - 1 ? a + 1 : a;
    *l1 = src - (float)a;
    *l0 = 1.0f - *l1;
}
/* Modulated deformable im2col -- the sampling stage of DCNv2, the deformable 3x3 convolutions of the YOLACT++ backbones
 * (README.md:216-221 lists the YOLACT++ models; dbolya/yolact's DCNv2 lineage [UPSTREAM-RECALL], PARITY UNPINNED).
 *   x   [N][H][W][C]
 *   om  [N][Ho][Wo][3K], K = R*S taps, the raw output of the block's conv_offset_mask: channel 2k = dy_k, 2k+1 = dx_k,
 *       2K+k = mask logit (DCNv2: offset = cat(o1, o2), mask = sigmoid(o3))
 *   out [N][Ho][Wo][K][C]: tap k = (i, j) sampled at (ho*stride - pad + i*dil + dy_k, wo*stride - pad + j*dil + dx_k) by the
 *       DCNv2 bilinear rule (zero outside the open range (-1, H) x (-1, W); corners outside the image contribute 0), times
 *       sigmoid(mask logit).  The deformable conv itself is then a 1x1 convolution over K*C channels with the block's KRSC
 *       weights (the same k-ordered chain as a plain 3x3 convolution over (r, s, cin)).
 * Rounding sequence: h = (float)(integer tap row) + dy; lh = h - floor(h); hh = 1 - lh; w1..w4 products; value =
 * ((w1*v1 + w2*v2) + w3*v3) + w4*v4, each operation individually rounded; result * sigmoid(m). */
ORA_API void ora_deform_im2col(const float* x, int N, int H, int W, int C, const float* om, int R, int S, int stride, int pad,
                               int dil, float* out) {
    const int K = R * S;
    const int Ho = (H + 2 * pad - dil * (R - 1) - 1) / stride + 1, Wo = (W + 2 * pad - dil * (S - 1) - 1) / stride + 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int ho = 0; ho < Ho; ++ho)
            for (int wo = 0; wo < Wo; ++wo) {
                const float* o = om + (((int64_t)n * Ho + ho) * Wo + wo) * 3 * K;
                float* dst = out + (((int64_t)n * Ho + ho) * Wo + wo) * K * C;
                for (int k = 0; k < K; ++k) {
                    const int i = k / S, j = k - i * S;
                    const float h = (float)(ho * stride - pad + i * dil) + o[2 * k];
                    const float w = (float)(wo * stride - pad + j * dil) + o[2 * k + 1];
                    const float m = ora_sigmoidf(o[2 * K + k]);
                    float* d = dst + (int64_t)k * C;
                    if (!(h > -1.0f && w > -1.0f && h < (float)H && w < (float)W)) {
                        for (int c = 0; c < C; ++c) d[c] = 0.0f;
                        continue;
                    }
                    const float hf = floorf(h), wf = floorf(w);
                    const int hl = (int)hf, wl = (int)wf, hh_ = hl + 1, wh_ = wl + 1;
                    const float lh = h - hf, lw = w - wf, hh = 1.0f - lh, hw = 1.0f - lw;
                    const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
                    const float* p1 = (hl >= 0 && wl >= 0) ? x + (((int64_t)n * H + hl) * W + wl) * C : NULL;
                    const float* p2 = (hl >= 0 && wh_ <= W - 1) ? x + (((int64_t)n * H + hl) * W + wh_) * C : NULL;
                    const float* p3 = (hh_ <= H - 1 && wl >= 0) ? x + (((int64_t)n * H + hh_) * W + wl) * C : NULL;
                    const float* p4 = (hh_ <= H - 1 && wh_ <= W - 1) ? x + (((int64_t)n * H + hh_) * W + wh_) * C : NULL;
                    for (int c = 0; c < C; ++c) {
                        const float v1 = p1 ? p1[c] : 0.0f, v2 = p2 ? p2[c] : 0.0f, v3 = p3 ? p3[c] : 0.0f, v4 = p4 ? p4[c] : 0.0f;
                        float v = w1 * v1 + w2 * v2;
                        v = v + w3 * v3;
                        v = v + w4 * v4;
                        d[c] = v * m;
                    }
                }
            }
}

ORA_API void ora_resize_bilinear(const float* in, int N, int H, int W, int C, int Ho, int Wo,
                                 const float* add, int relu, float* out) {
#pragma omp parallel for collapse(2)
    for (int n = 0; n < N; ++n)
        for (int y = 0; y < Ho; ++y) {
            int y0, y1; float ly0, ly1;
            bil_coef(y, H, Ho, &y0, &y1, &ly0, &ly1);
            for (int x = 0; x < Wo; ++x) {
                int x0, x1; float lx0, lx1;
                bil_coef(x, W, Wo, &x0, &x1, &lx0, &lx1);
                const float* p00 = in + (((size_t)n * H + y0) * W + x0) * C;
                const float* p01 = in + (((size_t)n * H + y0) * W + x1) * C;
                const float* p10 = in + (((size_t)n * H + y1) * W + x0) * C;
                const float* p11 = in + (((size_t)n * H + y1) * W + x1) * C;
                float* o = out + (((size_t)n * Ho + y) * Wo + x) * C;
                const float* ad = add ? add + (((size_t)n * Ho + y) * Wo + x) * C : NULL;
                for (int c = 0; c < C; ++c) {
                    float top = lx0 * p00[c]; top = fmaf(lx1, p01[c], top);
                    float bot = lx0 * p10[c]; bot = fmaf(lx1, p11[c], bot);
                    float v = ly0 * top; v = fmaf(ly1, bot, v);
                    if (ad) v = v + ad[c];
                    if (relu) v = v > 0.0f ? v : 0.0f;
                    o[c] = v;
                }
            }
        }
}

/* Appendix A.2 FPN top-down: out[y,x] = lateral[y,x] + coarse[y/2,x/2] (nearest x2). (M3) */
ORA_API void ora_upsample_nearest2x_add(const float* coarse, int N, int Hc, int Wc, int C,
                                        const float* lateral, int H, int W, float* out) {
#pragma omp parallel for collapse(2)
    for (int n = 0; n < N; ++n)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                int yc = y / 2, xc = x / 2;
                if (yc > Hc - 1) yc = Hc - 1;
                if (xc > Wc - 1) xc = Wc - 1;
                const float* a = lateral + (((size_t)n * H + y) * W + x) * C;
                const float* b = coarse + (((size_t)n * Hc + yc) * Wc + xc) * C;
                float* o = out + (((size_t)n * H + y) * W + x) * C;
                for (int c = 0; c < C; ++c) o[c] = a[c] + b[c];
            }
}

/* Softmax over the last dim (A.1): m=max; e=exp(x-m); s=sum in index order; p=e/s. */
ORA_API void ora_softmax(const float* x, int64_t rows, int C, float* y) {
#pragma omp parallel for
    for (int64_t r = 0; r < rows; ++r) {
        const float* a = x + r * C;
        float* o = y + r * C;
        float m = a[0];
        for (int c = 1; c < C; ++c) m = a[c] > m ? a[c] : m;
        float s = 0.0f;
        for (int c = 0; c < C; ++c) { o[c] = ora_expf(a[c] - m); s = s + o[c]; }
        for (int c = 0; c < C; ++c) o[c] = o[c] / s;
    }
}

/* ------------------------------------------------------------- selection -- */
typedef struct { float s; int32_t i; } ora_si;
static int cmp_si_desc(const void* a, const void* b) {
    const ora_si* x = (const ora_si*)a; const ora_si* y = (const ora_si*)b;
    if (x->s > y->s) return -1;
    if (x->s < y->s) return 1;
    return x->i < y->i ? -1 : (x->i > y->i ? 1 : 0);
}
/* top-k of n scores, sorted (score desc, index asc). Returns count=min(k,n). */
ORA_API int ora_topk(const float* scores, int n, int k, float* out_s, int32_t* out_i) {
    ora_si* v = (ora_si*)malloc(sizeof(ora_si) * (size_t)(n > 0 ? n : 1));
    for (int i = 0; i < n; ++i) { v[i].s = scores[i]; v[i].i = i; }
    qsort(v, (size_t)n, sizeof(ora_si), cmp_si_desc);
    const int m = k < n ? k : n;
    for (int i = 0; i < m; ++i) { out_s[i] = v[i].s; out_i[i] = v[i].i; }
    free(v);
    return m;
}

/* ------------------------------------------------------- Mask R-CNN: RPN -- */
/* Appendix A.4 BoxCoder.decode, legacy +1 widths, dw/dh clamp log(1000/16). (M6, M9) */
static inline void decode_box(const float* a, const float* d, float wx, float wy, float ww, float wh,
                              float* o) {
    const float clipv = 4.135166556742356f; /* log(1000/16) */
    const float widths = a[2] - a[0] + 1.0f, heights = a[3] - a[1] + 1.0f;
    const float ctr_x = a[0] + 0.5f * widths, ctr_y = a[1] + 0.5f * heights;
    const float dx = d[0] / wx, dy = d[1] / wy;
    float dw = d[2] / ww, dh = d[3] / wh;
    dw = dw < clipv ? dw : clipv;
    dh = dh < clipv ? dh : clipv;
    const float pcx = dx * widths + ctr_x, pcy = dy * heights + ctr_y;
    const float pw = ora_expf(dw) * widths, ph = ora_expf(dh) * heights;
    o[0] = pcx - 0.5f * pw;
    o[1] = pcy - 0.5f * ph;
    o[2] = pcx + 0.5f * pw - 1.0f;
    o[3] = pcy + 0.5f * ph - 1.0f;
}
static inline void clip_box(float* b, float im_w, float im_h) {
    const float mx = im_w - 1.0f, my = im_h - 1.0f;
    b[0] = b[0] < 0.0f ? 0.0f : (b[0] > mx ? mx : b[0]);
    b[1] = b[1] < 0.0f ? 0.0f : (b[1] > my ? my : b[1]);
    b[2] = b[2] < 0.0f ? 0.0f : (b[2] > mx ? mx : b[2]);
    b[3] = b[3] < 0.0f ? 0.0f : (b[3] > my ? my : b[3]);
}
ORA_API void ora_decode_boxes(const float* anchors, const float* deltas, int n, float wx, float wy,
                              float ww, float wh, float im_w, float im_h, int clip, float* out) {
    for (int i = 0; i < n; ++i) {
        decode_box(anchors + 4 * i, deltas + 4 * i, wx, wy, ww, wh, out + 4 * i);
        if (clip) clip_box(out + 4 * i, im_w, im_h);
    }
}

/* Appendix A.6 greedy NMS.  Candidates are visited in (score desc, index asc)
 * order.  IoU uses plus_one ? +1 legacy areas : plain areas.
 * ge ? suppress if iou>=thr : suppress if iou>thr.
 * keep[] receives ORIGINAL indices in score order; returns the count
 * (truncated to max_keep if max_keep>0). (M6 thr .7, M9 thr .5) */
static int cmp_i32_asc(const void* a, const void* b) {
    const int32_t x = *(const int32_t*)a, y = *(const int32_t*)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}
static inline float iou_plus(const float* a, const float* b, float one) {
    const float aa = (a[2] - a[0] + one) * (a[3] - a[1] + one);
    const float ab = (b[2] - b[0] + one) * (b[3] - b[1] + one);
    const float xx1 = a[0] > b[0] ? a[0] : b[0], yy1 = a[1] > b[1] ? a[1] : b[1];
    const float xx2 = a[2] < b[2] ? a[2] : b[2], yy2 = a[3] < b[3] ? a[3] : b[3];
    float w = xx2 - xx1 + one, h = yy2 - yy1 + one;
    w = w > 0.0f ? w : 0.0f;
    h = h > 0.0f ? h : 0.0f;
    const float inter = w * h;
    return inter / (aa + ab - inter);
}
ORA_API int ora_nms(const float* boxes, const float* scores, int n, float thr, int plus_one, int ge,
                    int max_keep, int32_t* keep) {
    if (n <= 0) return 0;
    ora_si* v = (ora_si*)malloc(sizeof(ora_si) * (size_t)n);
    for (int i = 0; i < n; ++i) { v[i].s = scores[i]; v[i].i = i; }
    qsort(v, (size_t)n, sizeof(ora_si), cmp_si_desc);
    unsigned char* dead = (unsigned char*)calloc((size_t)n, 1);
    const float one = plus_one ? 1.0f : 0.0f;
    int cnt = 0;
    for (int a = 0; a < n; ++a) {
        if (dead[a]) continue;
        keep[cnt++] = v[a].i;
        if (max_keep > 0 && cnt >= max_keep) break;
        const float* ba = boxes + 4 * (size_t)v[a].i;
        for (int b = a + 1; b < n; ++b) {
            if (dead[b]) continue;
            const float o = iou_plus(ba, boxes + 4 * (size_t)v[b].i, one);
            if (ge ? (o >= thr) : (o > thr)) dead[b] = 1;
        }
    }
    free(dead);
    free(v);
    return cnt;
}

/* Appendix A.4 RPNPostProcessor.forward_for_single_feature_map for ONE image
 * and ONE level: sigmoid -> top-k(pre_nms) sorted -> decode(1,1,1,1) -> clip to
 * the UNPADDED image -> remove_small(min_size) -> NMS(thr) keep <= post_nms.
 * logits [HWA], deltas [HWA][4], anchors [HWA][4].  Returns count. (M6) */
ORA_API int ora_rpn_level(const float* logits, const float* deltas, const float* anchors, int hwa,
                          int pre_nms, int post_nms, float nms_thr, float min_size, float im_w,
                          float im_h, int nms_flags, float* out_boxes, float* out_scores) {
    float* prob = (float*)malloc(sizeof(float) * (size_t)hwa);
    for (int i = 0; i < hwa; ++i) prob[i] = ora_sigmoidf(logits[i]);
    const int k = pre_nms < hwa ? pre_nms : hwa;
    float* ts = (float*)malloc(sizeof(float) * (size_t)k);
    int32_t* ti = (int32_t*)malloc(sizeof(int32_t) * (size_t)k);
    ora_topk(prob, hwa, k, ts, ti);
    float* bx = (float*)malloc(sizeof(float) * 4 * (size_t)k);
    float* sc = (float*)malloc(sizeof(float) * (size_t)k);
    int m = 0;
    for (int j = 0; j < k; ++j) {
        float b[4];
        decode_box(anchors + 4 * (size_t)ti[j], deltas + 4 * (size_t)ti[j], 1.f, 1.f, 1.f, 1.f, b);
        clip_box(b, im_w, im_h);
        const float ws = b[2] - b[0] + 1.0f, hs = b[3] - b[1] + 1.0f;
        if (ws >= min_size && hs >= min_size) { memcpy(bx + 4 * m, b, 16); sc[m] = ts[j]; ++m; }
    }
    int32_t* keep = (int32_t*)malloc(sizeof(int32_t) * (size_t)(m > 0 ? m : 1));
    /* nms_flags (App. A.6 forks): bit 0 suppress on >=, bit 1 plain areas instead of the legacy +1.  (Index order, bit 2, is moot here:
     * boxlist_nms truncates keep[:post_nms] of a score-sorted list, and in a score-sorted list index order IS score order.) */
    const int cnt = ora_nms(bx, sc, m, nms_thr, (nms_flags & 2) ? 0 : 1, nms_flags & 1, post_nms, keep);
    for (int j = 0; j < cnt; ++j) { memcpy(out_boxes + 4 * j, bx + 4 * keep[j], 16); out_scores[j] = sc[keep[j]]; }
    free(keep); free(sc); free(bx); free(ti); free(ts); free(prob);
    return cnt;
}

/* --------------------------------------------------- Mask R-CNN: RoIAlign -- */
/* Appendix A.7 LevelMapper: lvl=floor(4+log2(sqrt(area)/224+1e-6)) clamp[2,5]; area with +1. (M7) */
ORA_API void ora_level_map(const float* boxes, int n, int k_min, int k_max, int32_t* lvl) {
    for (int i = 0; i < n; ++i) {
        const float* b = boxes + 4 * i;
        const float area = (b[2] - b[0] + 1.0f) * (b[3] - b[1] + 1.0f);
        const float s = sqrtf(area);
        float t = floorf(4.0f + ora_log2f(s / 224.0f + 1e-6f));
        int l = (int)t;
        l = l < k_min ? k_min : (l > k_max ? k_max : l);
        lvl[i] = l;
    }
}

static inline float roi_bilinear(const float* f, int H, int W, int C, int c, float y, float x) {
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return 0.0f;
    if (y <= 0.0f) y = 0.0f;
    if (x <= 0.0f) x = 0.0f;
    int yl = (int)y, xl = (int)x, yh, xh;
    if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
    if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
    const float ly = y - (float)yl, lx = x - (float)xl, hy = 1.0f - ly, hx = 1.0f - lx;
    const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
    const float v1 = f[((size_t)yl * W + xl) * C + c], v2 = f[((size_t)yl * W + xh) * C + c];
    const float v3 = f[((size_t)yh * W + xl) * C + c], v4 = f[((size_t)yh * W + xh) * C + c];
    float v = w1 * v1;
    v = v + w2 * v2;
    v = v + w3 * v3;
    v = v + w4 * v4;
    return v;
}
/* Appendix A.7 RoIAlign (sampling g fixed) on ONE level.
 * feat [N][H][W][C] NHWC; rois [R][5] = (batch, x1,y1,x2,y2); out [R][PH][PW][C].
 * Sum over iy then ix, then / (g*g). (M7, M10)
 * aligned = 0: the legacy op (maskrcnn-benchmark ROIAlign_cpu / _cuda): no half-pixel shift, RoI at least 1 x 1.
 * aligned = 1: the other side of the A.7 fork, ROIAlign(aligned=True) as in the later detectron2 / torchvision kernels: scaled corners
 * minus 0.5 and NO minimum size. */
ORA_API void ora_roi_align2(const float* feat, int N, int H, int W, int C, const float* rois, int R,
                            float spatial_scale, int PH, int PW, int g, int aligned, float* out) {
    (void)N;
#pragma omp parallel for
    for (int r = 0; r < R; ++r) {
        const float* roi = rois + 5 * (size_t)r;
        const int b = (int)roi[0];
        const float* f = feat + (size_t)b * H * W * C;
        const float off = aligned ? 0.5f : 0.0f;
        const float sw = roi[1] * spatial_scale - off, sh = roi[2] * spatial_scale - off;
        const float ew = roi[3] * spatial_scale - off, eh = roi[4] * spatial_scale - off;
        float rw = ew - sw, rh = eh - sh;
        if (!aligned) {
            rw = rw > 1.0f ? rw : 1.0f;
            rh = rh > 1.0f ? rh : 1.0f;
        }
        const float bh = rh / (float)PH, bw = rw / (float)PW;
        /* sampling_ratio > 0: fixed g x g grid; <= 0: adaptive ceil(roi_size / pooled_size) (the ROIAlign default used by
         * the R-50-C4 config, whose yaml does not set POOLER_SAMPLING_RATIO) */
        const int gh = g > 0 ? g : (int)ceilf(rh / (float)PH);
        const int gw = g > 0 ? g : (int)ceilf(rw / (float)PW);
        const float count = (float)(gh * gw);
        for (int ph = 0; ph < PH; ++ph)
            for (int pw = 0; pw < PW; ++pw) {
                float* o = out + (((size_t)r * PH + ph) * PW + pw) * C;
                for (int c = 0; c < C; ++c) {
                    float acc = 0.0f;
                    for (int iy = 0; iy < gh; ++iy) {
                        const float y = sh + (float)ph * bh + ((float)iy + 0.5f) * bh / (float)gh;
                        for (int ix = 0; ix < gw; ++ix) {
                            const float x = sw + (float)pw * bw + ((float)ix + 0.5f) * bw / (float)gw;
                            acc = acc + roi_bilinear(f, H, W, C, c, y, x);
                        }
                    }
                    o[c] = acc / count;
                }
            }
    }
}

ORA_API void ora_roi_align(const float* feat, int N, int H, int W, int C, const float* rois, int R,
                           float spatial_scale, int PH, int PW, int g, float* out) {
    ora_roi_align2(feat, N, H, W, C, rois, R, spatial_scale, PH, PW, g, 0, out);
}

/* nn.AvgPool2d(k) on an R x k x k x C NHWC tensor with k == H == W (FastRCNNPredictor of the C4 head): sequential fp32
 * sum over (h, w), one division.  out [R][C]. */
ORA_API void ora_avgpool_full(const float* x, int R, int HW, int C, float* out) {
#pragma omp parallel for
    for (int r = 0; r < R; ++r)
        for (int c = 0; c < C; ++c) {
            float acc = 0.0f;
            for (int i = 0; i < HW; ++i) acc = acc + x[((size_t)r * HW + i) * C + c];
            out[(size_t)r * C + c] = acc / (float)HW;
        }
}

/* ------------------------------------------- Mask R-CNN: box post-process -- */
/* Appendix A.5 PostProcessor.filter_results for ONE image.
 * logits [R][ncls], regr [R][4*ncls], props [R][4].
 * Output rows in class order (1..ncls-1), within class NMS (score) order; if
 * more than det_per_img survive keep score >= kth value (ties kept), order
 * preserved; at most cap rows are written.  Returns count. (M9)
 * nms_flags (App. A.6 forks): bit 0 suppress on iou >= thr; bit 1 plain areas (no +1); bit 2 a class's kept boxes in ascending
 * candidate (= proposal) index, as maskrcnn-benchmark's CPU nms returns them (nonzero of the keep mask), instead of score order. */
ORA_API int ora_box_postprocess(const float* logits, const float* regr, const float* props, int R,
                                int ncls, float im_w, float im_h, float score_thr, float nms_thr,
                                int det_per_img, int nms_flags, int cap, float* out_boxes,
                                float* out_scores, int32_t* out_labels) {
    float* prob = (float*)malloc(sizeof(float) * (size_t)R * ncls);
    ora_softmax(logits, R, ncls, prob);
    const int maxtot = R * (ncls - 1);
    float* ab = (float*)malloc(sizeof(float) * 4 * (size_t)(maxtot > 0 ? maxtot : 1));
    float* as = (float*)malloc(sizeof(float) * (size_t)(maxtot > 0 ? maxtot : 1));
    int32_t* al = (int32_t*)malloc(sizeof(int32_t) * (size_t)(maxtot > 0 ? maxtot : 1));
    float* cb = (float*)malloc(sizeof(float) * 4 * (size_t)(R > 0 ? R : 1));
    float* cs = (float*)malloc(sizeof(float) * (size_t)(R > 0 ? R : 1));
    int32_t* keep = (int32_t*)malloc(sizeof(int32_t) * (size_t)(R > 0 ? R : 1));
    int tot = 0;
    for (int j = 1; j < ncls; ++j) {
        int m = 0;
        for (int i = 0; i < R; ++i) {
            const float p = prob[(size_t)i * ncls + j];
            if (p > score_thr) {
                decode_box(props + 4 * (size_t)i, regr + ((size_t)i * ncls + j) * 4, 10.f, 10.f, 5.f, 5.f, cb + 4 * m);
                clip_box(cb + 4 * m, im_w, im_h);
                cs[m] = p;
                ++m;
            }
        }
        const int cnt = ora_nms(cb, cs, m, nms_thr, (nms_flags & 2) ? 0 : 1, nms_flags & 1, 0, keep);
        if (nms_flags & 4) qsort(keep, (size_t)cnt, sizeof(int32_t), cmp_i32_asc);
        for (int q = 0; q < cnt; ++q) {
            memcpy(ab + 4 * (size_t)tot, cb + 4 * (size_t)keep[q], 16);
            as[tot] = cs[keep[q]];
            al[tot] = j;
            ++tot;
        }
    }
    int outc = 0;
    if (tot > det_per_img && det_per_img > 0) {
        /* kthvalue(k = tot - det + 1) = the det-th largest */
        float* tmp = (float*)malloc(sizeof(float) * (size_t)tot);
        int32_t* ti = (int32_t*)malloc(sizeof(int32_t) * (size_t)tot);
        ora_topk(as, tot, tot, tmp, ti);
        const float thr = tmp[det_per_img - 1];
        free(ti); free(tmp);
        for (int q = 0; q < tot && outc < cap; ++q)
            if (as[q] >= thr) { memcpy(out_boxes + 4 * outc, ab + 4 * (size_t)q, 16); out_scores[outc] = as[q]; out_labels[outc] = al[q]; ++outc; }
    } else {
        for (int q = 0; q < tot && outc < cap; ++q) { memcpy(out_boxes + 4 * outc, ab + 4 * (size_t)q, 16); out_scores[outc] = as[q]; out_labels[outc] = al[q]; ++outc; }
    }
    free(keep); free(cs); free(cb); free(al); free(as); free(ab); free(prob);
    return outc;
}

/* Appendix A.8 mask predictor tail: prob[r,y,x] = sigmoid( fmaf-chain_c(feat[r,y,x,c]*w[label_r][c]) + b[label_r] ).
 * feat [R][HW][C]; w [ncls][C]; out [R][HW]. (M11) */
ORA_API void ora_mask_logits_select(const float* feat, int R, int HW, int C, const float* w,
                                    const float* b, const int32_t* labels, float* out) {
#pragma omp parallel for
    for (int r = 0; r < R; ++r) {
        const float* wr = w + (size_t)labels[r] * C;
        for (int p = 0; p < HW; ++p) {
            const float* x = feat + ((size_t)r * HW + p) * C;
            float acc = 0.0f;
            for (int c = 0; c < C; ++c) acc = fmaf(x[c], wr[c], acc);
            const float y = fmaf(acc, 1.0f, b[labels[r]]);
            out[(size_t)r * HW + p] = ora_sigmoidf(y);
        }
    }
}

/* Appendix A.8 Masker(threshold, padding=1).paste for ONE image.
 * masks [n][M][M] probabilities; boxes [n][4] already in output-image coords.
 * out [n][im_h][im_w] uint8 (0/1). (M12) */
ORA_API void ora_paste_masks(const float* masks, const float* boxes, int n, int M, int im_h, int im_w,
                             float thr, uint8_t* out) {
    const int P = M + 2;
    const float scale = (float)P / (float)M;
#pragma omp parallel for
    for (int i = 0; i < n; ++i) {
        uint8_t* o = out + (size_t)i * im_h * im_w;
        memset(o, 0, (size_t)im_h * im_w);
        const float* b = boxes + 4 * i;
        float w_half = (b[2] - b[0]) * 0.5f, h_half = (b[3] - b[1]) * 0.5f;
        const float x_c = (b[2] + b[0]) * 0.5f, y_c = (b[3] + b[1]) * 0.5f;
        w_half = w_half * scale;
        h_half = h_half * scale;
        const int x1 = (int)(x_c - w_half), x2 = (int)(x_c + w_half);
        const int y1 = (int)(y_c - h_half), y2 = (int)(y_c + h_half);
        int w = x2 - x1 + 1, h = y2 - y1 + 1;
        w = w > 1 ? w : 1;
        h = h > 1 ? h : 1;
        const int x_0 = x1 > 0 ? x1 : 0, x_1 = (x2 + 1) < im_w ? (x2 + 1) : im_w;
        const int y_0 = y1 > 0 ? y1 : 0, y_1 = (y2 + 1) < im_h ? (y2 + 1) : im_h;
        const float* m = masks + (size_t)i * M * M;
        for (int y = y_0; y < y_1; ++y) {
            int sy0, sy1; float ly0, ly1;
            bil_coef(y - y1, P, h, &sy0, &sy1, &ly0, &ly1);
            for (int x = x_0; x < x_1; ++x) {
                int sx0, sx1; float lx0, lx1;
                bil_coef(x - x1, P, w, &sx0, &sx1, &lx0, &lx1);
#define PADV(yy, xx) (((yy) >= 1 && (yy) <= M && (xx) >= 1 && (xx) <= M) ? m[((yy)-1) * M + ((xx)-1)] : 0.0f)
                float top = lx0 * PADV(sy0, sx0); top = fmaf(lx1, PADV(sy0, sx1), top);
                float bot = lx0 * PADV(sy1, sx0); bot = fmaf(lx1, PADV(sy1, sx1), bot);
#undef PADV
                float v = ly0 * top; v = fmaf(ly1, bot, v);
                o[(size_t)y * im_w + x] = v > thr ? 1 : 0;
            }
        }
    }
}

/* ------------------------------------------------------------------ Yolact -- */
/* Appendix A.9 decode: priors (cx,cy,w,h), loc [P][4], variances (.1,.2) -> xyxy. (Y6) */
ORA_API void ora_yolact_decode(const float* loc, const float* priors, int P, float* boxes) {
    for (int i = 0; i < P; ++i) {
        const float* l = loc + 4 * (size_t)i; const float* p = priors + 4 * (size_t)i;
        float tx = l[0] * 0.1f; tx = tx * p[2];
        float ty = l[1] * 0.1f; ty = ty * p[3];
        const float cx = p[0] + tx, cy = p[1] + ty;
        const float w = p[2] * ora_expf(l[2] * 0.2f), h = p[3] * ora_expf(l[3] * 0.2f);
        const float x1 = cx - w / 2.0f, y1 = cy - h / 2.0f;
        float* o = boxes + 4 * (size_t)i;
        o[0] = x1; o[1] = y1; o[2] = w + x1; o[3] = h + y1;
    }
}

static inline float jaccard1(const float* a, const float* b) {
    const float mx2 = a[2] < b[2] ? a[2] : b[2], mx1 = a[0] > b[0] ? a[0] : b[0];
    const float my2 = a[3] < b[3] ? a[3] : b[3], my1 = a[1] > b[1] ? a[1] : b[1];
    float iw = mx2 - mx1, ih = my2 - my1;
    iw = iw > 0.0f ? iw : 0.0f;
    ih = ih > 0.0f ? ih : 0.0f;
    const float inter = iw * ih;
    const float aa = (a[2] - a[0]) * (a[3] - a[1]), ab = (b[2] - b[0]) * (b[3] - b[1]);
    const float uni = aa + ab - inter;
    return inter / uni;
}

/* Appendix A.9/A.6 Detect for ONE image: conf [P][ncls] are SOFTMAX probabilities.
 *  keep prior if max_{c>=1} conf > conf_thresh; per class: stable sort desc, top_k;
 *  fast-NMS: box j of class c survives iff max_{i<j} iou(i,j) <= nms_thr (NaN drops)
 *  [and, with second_threshold (the A.6 fork, fast_nms(second_threshold=True); off in the default detect() call), iff its own class
 *  score > conf_thresh];  gather class-major, stable sort desc, first max_det.
 * Outputs: boxes[max_det][4], scores, classes (0..ncls-2), coeffs [max_det][mask_dim],
 * prior index.  Returns count. (Y6) */
ORA_API int ora_yolact_detect2(const float* conf, const float* boxes, const float* mask, int P, int ncls,
                               int mask_dim, float conf_thresh, float nms_thr, int top_k, int max_det, int second_threshold,
                               float* out_boxes, float* out_scores, int32_t* out_classes,
                               float* out_coeffs, int32_t* out_prior) {
    int32_t* kept = (int32_t*)malloc(sizeof(int32_t) * (size_t)P);
    int nk = 0;
    for (int i = 0; i < P; ++i) {
        float m = conf[(size_t)i * ncls + 1];
        for (int c = 2; c < ncls; ++c) { const float v = conf[(size_t)i * ncls + c]; m = v > m ? v : m; }
        if (m > conf_thresh) kept[nk++] = i;
    }
    if (nk == 0) { free(kept); return 0; }
    const int nc = ncls - 1;
    const int tk = top_k < nk ? top_k : nk;
    float* cs = (float*)malloc(sizeof(float) * (size_t)nk);
    float* ts = (float*)malloc(sizeof(float) * (size_t)tk);
    int32_t* ti = (int32_t*)malloc(sizeof(int32_t) * (size_t)tk);
    float* fs = (float*)calloc((size_t)nc * tk > 0 ? (size_t)nc * tk : 1, sizeof(float));
    int32_t* fp = (int32_t*)malloc(sizeof(int32_t) * (size_t)nc * tk);
    int32_t* fc = (int32_t*)malloc(sizeof(int32_t) * (size_t)nc * tk);
    int tot = 0;
    for (int c = 0; c < nc; ++c) {
        for (int q = 0; q < nk; ++q) cs[q] = conf[(size_t)kept[q] * ncls + c + 1];
        ora_topk(cs, nk, tk, ts, ti);
        for (int j = 0; j < tk; ++j) {
            const float* bj = boxes + 4 * (size_t)kept[ti[j]];
            int ok = 1;
            for (int i = 0; i < j; ++i) {
                const float o = jaccard1(boxes + 4 * (size_t)kept[ti[i]], bj);
                if (!(o <= nms_thr)) { ok = 0; break; }
            }
            if (second_threshold && !(ts[j] > conf_thresh)) ok = 0;
            if (ok) { fs[tot] = ts[j]; fp[tot] = kept[ti[j]]; fc[tot] = c; ++tot; }
        }
    }
    const int m = max_det < tot ? max_det : tot;
    float* os = (float*)malloc(sizeof(float) * (size_t)(tot > 0 ? tot : 1));
    int32_t* oi = (int32_t*)malloc(sizeof(int32_t) * (size_t)(tot > 0 ? tot : 1));
    ora_topk(fs, tot, m, os, oi);
    for (int q = 0; q < m; ++q) {
        const int src = oi[q];
        memcpy(out_boxes + 4 * q, boxes + 4 * (size_t)fp[src], 16);
        out_scores[q] = os[q];
        out_classes[q] = fc[src];
        memcpy(out_coeffs + (size_t)q * mask_dim, mask + (size_t)fp[src] * mask_dim, sizeof(float) * (size_t)mask_dim);
        out_prior[q] = fp[src];
    }
    free(oi); free(os); free(fc); free(fp); free(fs); free(ti); free(ts); free(cs); free(kept);
    return m;
}

ORA_API int ora_yolact_detect(const float* conf, const float* boxes, const float* mask, int P, int ncls,
                              int mask_dim, float conf_thresh, float nms_thr, int top_k, int max_det,
                              float* out_boxes, float* out_scores, int32_t* out_classes,
                              float* out_coeffs, int32_t* out_prior) {
    return ora_yolact_detect2(conf, boxes, mask, P, ncls, mask_dim, conf_thresh, nms_thr, top_k, max_det, 0, out_boxes, out_scores,
                              out_classes, out_coeffs, out_prior);
}

/* Appendix A.9 sanitize_coordinates(cast=False). */
static inline void sanitize(float a, float b, int img, float padding, float* o1, float* o2) {
    a = a * (float)img; b = b * (float)img;
    float lo = a < b ? a : b, hi = a > b ? a : b;
    lo = lo - padding; hi = hi + padding;
    lo = lo > 0.0f ? lo : 0.0f;
    hi = hi < (float)img ? hi : (float)img;
    *o1 = lo; *o2 = hi;
}
/* Appendix A.9 postprocess masks for ONE image:
 *   m[d,y,x] = sigmoid(fmaf-chain_k proto[y,x,k]*coeff[d,k]); crop to the box (+1px pad) in
 *   proto space; bilinear (align_corners=False) to (h,w); >0.5 -> uint8.
 * proto [PH][PW][K]; coeffs [n][K]; boxes [n][4] relative xyxy; out [n][h][w].
 * Also emits the integer boxes (int64, A.9 last line) into out_boxes_i64 [n][4]. (Y7) */
ORA_API void ora_yolact_masks(const float* proto, int PH, int PW, int K, const float* coeffs,
                              const float* boxes, int n, int h, int w, uint8_t* out,
                              int64_t* out_boxes_i64) {
#pragma omp parallel for
    for (int d = 0; d < n; ++d) {
        float* lo = (float*)malloc(sizeof(float) * (size_t)PH * PW);
        const float* cf = coeffs + (size_t)d * K;
        const float* b = boxes + 4 * (size_t)d;
        float x1, x2, y1, y2;
        sanitize(b[0], b[2], PW, 1.0f, &x1, &x2);
        sanitize(b[1], b[3], PH, 1.0f, &y1, &y2);
        for (int y = 0; y < PH; ++y)
            for (int x = 0; x < PW; ++x) {
                const float* p = proto + ((size_t)y * PW + x) * K;
                float acc = 0.0f;
                for (int k = 0; k < K; ++k) acc = fmaf(p[k], cf[k], acc);
                float v = ora_sigmoidf(acc);
                const int inside = ((float)x >= x1) && ((float)x < x2) && ((float)y >= y1) && ((float)y < y2);
                lo[(size_t)y * PW + x] = inside ? v : 0.0f;
            }
        uint8_t* o = out + (size_t)d * h * w;
        for (int y = 0; y < h; ++y) {
            int y0i, y1i; float ly0, ly1;
            bil_coef(y, PH, h, &y0i, &y1i, &ly0, &ly1);
            for (int x = 0; x < w; ++x) {
                int x0i, x1i; float lx0, lx1;
                bil_coef(x, PW, w, &x0i, &x1i, &lx0, &lx1);
                float top = lx0 * lo[(size_t)y0i * PW + x0i]; top = fmaf(lx1, lo[(size_t)y0i * PW + x1i], top);
                float bot = lx0 * lo[(size_t)y1i * PW + x0i]; bot = fmaf(lx1, lo[(size_t)y1i * PW + x1i], bot);
                float v = ly0 * top; v = fmaf(ly1, bot, v);
                o[(size_t)y * w + x] = v > 0.5f ? 1 : 0;
            }
        }
        float bx1, bx2, by1, by2;
        sanitize(b[0], b[2], w, 0.0f, &bx1, &bx2);
        sanitize(b[1], b[3], h, 0.0f, &by1, &by2);
        out_boxes_i64[4 * d + 0] = (int64_t)bx1; out_boxes_i64[4 * d + 1] = (int64_t)by1;
        out_boxes_i64[4 * d + 2] = (int64_t)bx2; out_boxes_i64[4 * d + 3] = (int64_t)by2;
        free(lo);
    }
}

/* The proto-resolution stage of ora_yolact_masks on its own: lo[d] = crop(sigmoid(proto @ coeffs[d])), [n][PH][PW].
 * YOLACT++'s fast mask re-scoring net (FastMaskIoUNet) takes these as its input (output_utils.postprocess: maskiou_net(masks)
 * before the upsampling) [UPSTREAM-RECALL, PARITY UNPINNED]. */
ORA_API void ora_yolact_proto_masks(const float* proto, int PH, int PW, int K, const float* coeffs, const float* boxes, int n,
                                    float* lo_out) {
#pragma omp parallel for
    for (int d = 0; d < n; ++d) {
        float* lo = lo_out + (size_t)d * PH * PW;
        const float* cf = coeffs + (size_t)d * K;
        const float* b = boxes + 4 * (size_t)d;
        float x1, x2, y1, y2;
        sanitize(b[0], b[2], PW, 1.0f, &x1, &x2);
        sanitize(b[1], b[3], PH, 1.0f, &y1, &y2);
        for (int y = 0; y < PH; ++y)
            for (int x = 0; x < PW; ++x) {
                const float* p = proto + ((size_t)y * PW + x) * K;
                float acc = 0.0f;
                for (int k = 0; k < K; ++k) acc = fmaf(p[k], cf[k], acc);
                const float v = ora_sigmoidf(acc);
                const int inside = ((float)x >= x1) && ((float)x < x2) && ((float)y >= y1) && ((float)y < y2);
                lo[(size_t)y * PW + x] = inside ? v : 0.0f;
            }
    }
}

/* ---- COCO mask run-length encoding (the on-disk format behind inference() / tools/test_net.py, README.md:344-347, and Yolact eval.py's
 * Detections.add_mask / dump, README.md:243-249).  The algorithm lives in a third-party dependency that is absent from /root/reference and
 * from this image: pycocotools (cocoapi, common/maskApi.c), un-pinned by the reference (its README installs no specific version).  Restated
 * from the published maskApi.c: rleEncode walks the mask in COLUMN-major order and emits run lengths starting with the run of zeros (so a
 * mask whose first pixel is set starts with a 0 count); rleToString writes every count -- from the fourth on as the difference to the count
 * two runs back -- as little-endian groups of 5 bits, bit 5 = "more groups follow", + 48.  PARITY UNPINNED like the rest of the oracle; pinned
 * by the hand-computed cases of tests/test_coco_cpu.py and tests/test_oracle_cpu.py. */
/* mask: h x w row-major uint8 (non-zero = set), row pitch `pitch`; counts: capacity h*w + 1.  Returns the number of counts. */
ORA_API int64_t ora_rle_encode(const uint8_t* mask, int h, int w, int pitch, uint32_t* counts) {
    int64_t k = 0;
    uint32_t c = 0;
    int p = 0;
    for (int x = 0; x < w; ++x)
        for (int y = 0; y < h; ++y) {
            const int v = mask[(size_t)y * pitch + x] != 0;
            if (v != p) { counts[k++] = c; c = 0; p = v; }
            ++c;
        }
    counts[k++] = c;
    return k;
}

/* s: capacity 7 * m + 1 (a count below 2^32 and its difference need at most 7 groups).  Returns the string length (no terminator counted). */
ORA_API int64_t ora_rle_to_string(const uint32_t* counts, int64_t m, char* s) {
    int64_t p = 0;
    for (int64_t i = 0; i < m; ++i) {
        long long x = (long long)counts[i];
        if (i > 2) x -= (long long)counts[i - 2];
        int more = 1;
        while (more) {
            char c = (char)(x & 0x1f);
            x >>= 5;
            more = (c & 0x10) ? x != -1 : x != 0;
            if (more) c |= 0x20;
            c += 48;
            s[p++] = c;
        }
    }
    s[p] = 0;
    return p;
}

/* ---- Front end: M1 (COCODemo.build_transform + to_image_list) and Y1 (FastBaseTransform) -- SURVEY.md 8a rows M1 / Y1, App. A.0 constants,
 * App. A.1 bilinear rule; README.md:320-331 hands the predictor an HxWx3 uint8 BGR image, README.md:243 an image file read the same way.
 * PARITY UNPINNED like the rest of the oracle.  uint8 in, fp32 NHWC out; the PIL resize of M1 (Resize(min 800, max 1333)) stays on the host in
 * front of this (image decode / resize are outside the hot path).
 *
 * ora_fast_base_transform (Y1): img = F.interpolate(img, (S, S), mode='bilinear', align_corners=False) on the float image, then
 *   (img - mean) / std per BGR channel, then BGR -> RGB (swap_rb); `yolact_darknet53_config` normalises with x / 255 instead, which is
 *   mean 0 / std 255 here.  Rounding sequence (every operation individually rounded fp32; the file is built with -ffp-contract=off):
 *     scale = (float)in / (float)out;  src = ((float)dst + 0.5f) * scale - 0.5f;  src = max(src, 0);  i0 = min((int)src, in - 1);
 *     i1 = min(i0 + 1, in - 1);  l1 = src - (float)i0;  l0 = 1 - l1;
 *     top = v00 * l0x + v01 * l1x;  bot = v10 * l0x + v11 * l1x;  v = top * l0y + bot * l1y;  out = (v - mean) / std   (IEEE division)
 *   With in == out every l1 is exactly 0 and the resize is the identity: out = ((float)u8 - mean) / std. */
ORA_API void ora_fast_base_transform(const uint8_t* img, int N, int H, int W, int S, const float* mean3, const float* std3, int swap_rb,
                                     float* out) {
    const float sch = (float)H / (float)S, scw = (float)W / (float)S;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int y = 0; y < S; ++y) {
            float sy = ((float)y + 0.5f) * sch - 0.5f;
            if (sy < 0.0f) sy = 0.0f;
            int y0 = (int)sy;
            if (y0 > H - 1) y0 = H - 1;
            const int y1 = y0 + 1 < H - 1 ? y0 + 1 : H - 1;
            const float ly1 = sy - (float)y0, ly0 = 1.0f - ly1;
            const uint8_t* im = img + (size_t)n * H * W * 3;
            for (int x = 0; x < S; ++x) {
                float sx = ((float)x + 0.5f) * scw - 0.5f;
                if (sx < 0.0f) sx = 0.0f;
                int x0 = (int)sx;
                if (x0 > W - 1) x0 = W - 1;
                const int x1 = x0 + 1 < W - 1 ? x0 + 1 : W - 1;
                const float lx1 = sx - (float)x0, lx0 = 1.0f - lx1;
                float* o = out + (((size_t)n * S + y) * S + x) * 3;
                for (int c = 0; c < 3; ++c) {
                    const float v00 = (float)im[((size_t)y0 * W + x0) * 3 + c], v01 = (float)im[((size_t)y0 * W + x1) * 3 + c];
                    const float v10 = (float)im[((size_t)y1 * W + x0) * 3 + c], v11 = (float)im[((size_t)y1 * W + x1) * 3 + c];
                    float top = v00 * lx0; { const float t = v01 * lx1; top = top + t; }
                    float bot = v10 * lx0; { const float t = v11 * lx1; bot = bot + t; }
                    float v = top * ly0; { const float t = bot * ly1; v = v + t; }
                    v = v - mean3[c];
                    o[swap_rb ? 2 - c : c] = v / std3[c];
                }
            }
        }
}

/* ora_build_transform (M1, after the host's PIL resize): one already-resized h x w x 3 uint8 BGR image -> its slot of the batch tensor
 *   to_image_list(images, SIZE_DIVISIBILITY) builds: ToTensor()*255 keeps BGR 0..255 (TO_BGR255), Normalize(mean = PIXEL_MEAN, std = (1, 1, 1)):
 *   out[y, x, c] = (float)u8 - mean[c] (a division by 1 is exact and omitted), zero in the padding up to Hpad x Wpad (the batch's
 *   largest (h, w), each rounded up to a multiple of SIZE_DIVISIBILITY = 32 by the caller).  The unpadded (h, w) is what the caller
 *   remembers as image_sizes. */
ORA_API void ora_build_transform(const uint8_t* img, int h, int w, int Hpad, int Wpad, const float* mean3, float* out) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < Hpad; ++y)
        for (int x = 0; x < Wpad; ++x) {
            float* o = out + ((size_t)y * Wpad + x) * 3;
            if (y < h && x < w)
                for (int c = 0; c < 3; ++c) o[c] = (float)img[((size_t)y * w + x) * 3 + c] - mean3[c];
            else
                o[0] = o[1] = o[2] = 0.0f;
        }
}

ORA_API int ora_version(void) { return 1; }
