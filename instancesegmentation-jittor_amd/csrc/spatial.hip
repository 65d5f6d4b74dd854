// spatial.hip -- HBM-bound NHWC spatial ops: max-pool, bilinear resize(+add), nearest2x+add,
// channel pad, deterministic elementwise map.  One thread per 4 channels (16-byte accesses,
// consecutive lanes on consecutive channel quads -> fully coalesced NHWC rows).
// Reference anchors: SURVEY.md 8a M2/M3 (maxpool, nearest FPN), Y3/Y4 (bilinear), App. A.1.
#include "../../include/isegmi.h"
#include "common.h"
#include "detmath.h"

namespace isegmi {

__global__ void maxpool_kernel(const float* __restrict__ in, int N, int H, int W, int C, int k, int s, int p, int Ho,
                               int Wo, float* __restrict__ out) {
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
                const float4 v = *(const float4*)(in + (((int64_t)n * H + hi) * W + wi) * C + c4 * 4);
                m.x = v.x > m.x ? v.x : m.x; m.y = v.y > m.y ? v.y : m.y;
                m.z = v.z > m.z ? v.z : m.z; m.w = v.w > m.w ? v.w : m.w;
            }
        }
        *(float4*)(out + (((int64_t)n * Ho + ho) * Wo + wo) * C + c4 * 4) = m;
    }
}

__device__ __forceinline__ float bil1(float lx0, float lx1, float ly0, float ly1, float v00, float v01, float v10,
                                      float v11) {
    float top = lx0 * v00; top = fmaf(lx1, v01, top);
    float bot = lx0 * v10; bot = fmaf(lx1, v11, bot);
    float v = ly0 * top; v = fmaf(ly1, bot, v);
    return v;
}

__global__ void resize_bilinear_kernel(const float* __restrict__ in, int N, int H, int W, int C, int Ho, int Wo,
                                       const float* __restrict__ add, int relu, float* __restrict__ out) {
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
        const float* b = in + (int64_t)n * H * W * C + c4 * 4;
        const float4 v00 = *(const float4*)(b + ((int64_t)y0 * W + x0) * C);
        const float4 v01 = *(const float4*)(b + ((int64_t)y0 * W + x1) * C);
        const float4 v10 = *(const float4*)(b + ((int64_t)y1 * W + x0) * C);
        const float4 v11 = *(const float4*)(b + ((int64_t)y1 * W + x1) * C);
        float4 o;
        o.x = bil1(lx0, lx1, ly0, ly1, v00.x, v01.x, v10.x, v11.x);
        o.y = bil1(lx0, lx1, ly0, ly1, v00.y, v01.y, v10.y, v11.y);
        o.z = bil1(lx0, lx1, ly0, ly1, v00.z, v01.z, v10.z, v11.z);
        o.w = bil1(lx0, lx1, ly0, ly1, v00.w, v01.w, v10.w, v11.w);
        const int64_t oo = (((int64_t)n * Ho + y) * Wo + x) * C + c4 * 4;
        if (add) {
            const float4 a = *(const float4*)(add + oo);
            o.x = o.x + a.x; o.y = o.y + a.y; o.z = o.z + a.z; o.w = o.w + a.w;
        }
        if (relu) {
            o.x = o.x > 0.f ? o.x : 0.f; o.y = o.y > 0.f ? o.y : 0.f;
            o.z = o.z > 0.f ? o.z : 0.f; o.w = o.w > 0.f ? o.w : 0.f;
        }
        *(float4*)(out + oo) = o;
    }
}

__global__ void nearest2x_add_kernel(const float* __restrict__ coarse, int N, int Hc, int Wc, int C,
                                     const float* __restrict__ lat, int H, int W, float* __restrict__ out) {
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
        const float4 a = *(const float4*)(lat + i * 4);
        const float4 b = *(const float4*)(coarse + (((int64_t)n * Hc + yc) * Wc + xc) * C + c4 * 4);
        *(float4*)(out + i * 4) = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}

__global__ void pad_c3_c4_kernel(const float* __restrict__ in, int64_t npix, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (int64_t)gridDim.x * blockDim.x) {
        const float* s = in + i * 3;
        *(float4*)(out + i * 4) = make_float4(s[0], s[1], s[2], 0.0f);
    }
}

// Device front end: a uint8 HxWx3 image batch -> the engines' fp32 NHWC3 input, bit-identical to the host
// transforms of isegmi/transforms.py + fast_base_transform / prepare_images (numpy fp32 semantics: every multiply and add rounded on
// its own -- this file is built with -ffp-contract=off -- and an IEEE division):
//   v = bilinear(in, align_corners = False)   [F.interpolate; identity when the sizes match: the far taps' weight is exactly 0]
//   out[.., swap ? 2 - c : c] = (v[c] - mean[c]) / std[c]      inside Hout x Wout, 0 in the padding up to Hpad x Wpad (to_image_list)
// Uploading the bytes instead of the floats cuts the PCIe traffic of a batch by 4.
struct PreU8 {
    const uint8_t* in;
    float* out;
    int N, Hin, Win, Hout, Wout, Hpad, Wpad, swap;
    int64_t out_img_stride;  // floats
    float sch, scw;          // np.float32(Hin / Hout), np.float32(Win / Wout)
    float mean[3], stdv[3];
};

__global__ void preprocess_u8_kernel(const PreU8 a) {
    const int64_t total = (int64_t)a.N * a.Hpad * a.Wpad;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % a.Wpad);
        const int64_t r = i / a.Wpad;
        const int y = (int)(r % a.Hpad), n = (int)(r / a.Hpad);
        float* o = a.out + (int64_t)n * a.out_img_stride + ((int64_t)y * a.Wpad + x) * 3;
        if (y >= a.Hout || x >= a.Wout) { o[0] = 0.0f; o[1] = 0.0f; o[2] = 0.0f; continue; }
        const float sy = fmaxf(((float)y + 0.5f) * a.sch - 0.5f, 0.0f), sx = fmaxf(((float)x + 0.5f) * a.scw - 0.5f, 0.0f);
        int y0 = (int)sy, x0 = (int)sx;
        y0 = y0 < a.Hin - 1 ? y0 : a.Hin - 1;
        x0 = x0 < a.Win - 1 ? x0 : a.Win - 1;
        const int y1 = y0 + 1 < a.Hin - 1 ? y0 + 1 : a.Hin - 1, x1 = x0 + 1 < a.Win - 1 ? x0 + 1 : a.Win - 1;
        const float ly1 = sy - (float)y0, lx1 = sx - (float)x0;
        const float ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
        const uint8_t* img = a.in + (int64_t)n * a.Hin * a.Win * 3;
        const uint8_t* p00 = img + ((int64_t)y0 * a.Win + x0) * 3;
        const uint8_t* p01 = img + ((int64_t)y0 * a.Win + x1) * 3;
        const uint8_t* p10 = img + ((int64_t)y1 * a.Win + x0) * 3;
        const uint8_t* p11 = img + ((int64_t)y1 * a.Win + x1) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float top = (float)p00[c] * lx0 + (float)p01[c] * lx1;
            const float bot = (float)p10[c] * lx0 + (float)p11[c] * lx1;
            const float v = top * ly0 + bot * ly1;
            o[a.swap ? 2 - c : c] = (v - a.mean[c]) / a.stdv[c];
        }
    }
}

__global__ void map_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t n, int fn) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        y[i] = fn == 0 ? dm_exp(v) : fn == 1 ? dm_sigmoid(v) : fn == 2 ? dm_tanh(v) : dm_log2(v);
    }
}

static inline unsigned grid_for(int64_t total) {
    int64_t b = cdiv64(total, 256);
    if (b > 2048) b = 2048;
    if (b < 1) b = 1;
    return (unsigned)b;
}

int maxpool_launch(const float* in, int N, int H, int W, int C, int k, int s, int p, float* out, hipStream_t st) {
    ARG_CHECK(C % 4 == 0, "C % 4");
    const int Ho = (H + 2 * p - k) / s + 1, Wo = (W + 2 * p - k) / s + 1;
    hipLaunchKernelGGL(maxpool_kernel, dim3(grid_for((int64_t)N * Ho * Wo * (C / 4))), dim3(256), 0, st, in, N, H, W, C, k,
                       s, p, Ho, Wo, out);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}
int resize_bilinear_launch(const float* in, int N, int H, int W, int C, int Ho, int Wo, const float* add, int relu,
                           float* out, hipStream_t st) {
    ARG_CHECK(C % 4 == 0, "C % 4");
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3(grid_for((int64_t)N * Ho * Wo * (C / 4))), dim3(256), 0, st, in, N, H,
                       W, C, Ho, Wo, add, relu, out);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}
// Modulated deformable im2col (DCNv2 sampling stage of the YOLACT++ backbones; oracle: ora_deform_im2col, same rounding
// sequence).  One thread per (output pixel, tap, 4-channel vector): the three offset/mask values of a (pixel, tap) are
// broadcast reads, the four corner vectors 16-byte gathers, the store a coalesced 16 bytes.  HBM-bound: writes K*C*4 B per
// output pixel, reads <= 4 corner vectors per tap (mostly L2 hits: neighbouring taps and pixels share corners).
__global__ __launch_bounds__(256) void deform_im2col_kernel(const float* __restrict__ x, int N, int H, int W, int C,
                                                            const float* __restrict__ om, int R, int S, int stride, int pad, int dil,
                                                            int Ho, int Wo, float* __restrict__ out) {
    const int K = R * S, C4 = C >> 2;
    const int64_t total = (int64_t)N * Ho * Wo * K * C4;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int c4 = (int)(t % C4);
        int64_t r = t / C4;
        const int k = (int)(r % K);
        const int64_t pix = r / K;
        const int wo = (int)(pix % Wo);
        const int64_t q = pix / Wo;
        const int ho = (int)(q % Ho), n = (int)(q / Ho);
        const float* o = om + pix * 3 * K;
        const int i = k / S, j = k - i * S;
        const float h = (float)(ho * stride - pad + i * dil) + o[2 * k];
        const float w = (float)(wo * stride - pad + j * dil) + o[2 * k + 1];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (h > -1.0f && w > -1.0f && h < (float)H && w < (float)W) {
            const float m = dm_sigmoid(o[2 * K + k]);
            const float hf = floorf(h), wf = floorf(w);
            const int hl = (int)hf, wl = (int)wf, hh_ = hl + 1, wh_ = wl + 1;
            const float lh = h - hf, lw = w - wf, hh = 1.0f - lh, hw = 1.0f - lw;
            const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
            const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
            const float4* base = (const float4*)(x + (int64_t)n * H * W * C) + c4;
            const float4 v1 = (hl >= 0 && wl >= 0) ? base[((int64_t)hl * W + wl) * C4] : z;
            const float4 v2 = (hl >= 0 && wh_ <= W - 1) ? base[((int64_t)hl * W + wh_) * C4] : z;
            const float4 v3 = (hh_ <= H - 1 && wl >= 0) ? base[((int64_t)hh_ * W + wl) * C4] : z;
            const float4 v4 = (hh_ <= H - 1 && wh_ <= W - 1) ? base[((int64_t)hh_ * W + wh_) * C4] : z;
            v.x = (((w1 * v1.x + w2 * v2.x) + w3 * v3.x) + w4 * v4.x) * m;
            v.y = (((w1 * v1.y + w2 * v2.y) + w3 * v3.y) + w4 * v4.y) * m;
            v.z = (((w1 * v1.z + w2 * v2.z) + w3 * v3.z) + w4 * v4.z) * m;
            v.w = (((w1 * v1.w + w2 * v2.w) + w3 * v3.w) + w4 * v4.w) * m;
        }
        ((float4*)out)[t] = v;
    }
}
int deform_im2col_launch(const float* x, int N, int H, int W, int C, const float* om, int R, int S, int stride, int pad, int dil,
                         float* out, hipStream_t st) {
    ARG_CHECK(C % 4 == 0 && R > 0 && S > 0 && stride > 0 && dil > 0 && pad >= 0, "deform_im2col geometry (C % 4 == 0)");
    const int Ho = (H + 2 * pad - dil * (R - 1) - 1) / stride + 1, Wo = (W + 2 * pad - dil * (S - 1) - 1) / stride + 1;
    ARG_CHECK(Ho > 0 && Wo > 0, "deform_im2col output size");
    hipLaunchKernelGGL(deform_im2col_kernel, dim3(grid_for((int64_t)N * Ho * Wo * R * S * (C / 4))), dim3(256), 0, st, x, N, H, W, C, om, R,
                       S, stride, pad, dil, Ho, Wo, out);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}
int nearest2x_add_launch(const float* coarse, int N, int Hc, int Wc, int C, const float* lat, int H, int W, float* out,
                         hipStream_t st) {
    ARG_CHECK(C % 4 == 0, "C % 4");
    hipLaunchKernelGGL(nearest2x_add_kernel, dim3(grid_for((int64_t)N * H * W * (C / 4))), dim3(256), 0, st, coarse, N, Hc,
                       Wc, C, lat, H, W, out);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}
// NHWC C=3 -> C=32 (channels 3..31 zero): the input of a 3-channel 3x3 first layer (Darknet53's _preconv) that runs on the
// ordinary Cin % 32 == 0 conv kernels with zero weights on the padding channels.  One thread per 16 output bytes.
__global__ void pad_c3_c32_kernel(const float* __restrict__ in, int64_t npix, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix * 8; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pix = i >> 3;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if ((i & 7) == 0) { const float* s = in + pix * 3; v = make_float4(s[0], s[1], s[2], 0.0f); }
        ((float4*)out)[i] = v;
    }
}
int pad_c3_c32_launch(const float* in, int64_t npix, float* out, hipStream_t st) {
    hipLaunchKernelGGL(pad_c3_c32_kernel, dim3(grid_for(npix * 8)), dim3(256), 0, st, in, npix, out);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}
int pad_c3_c4_launch(const float* in, int64_t npix, float* out, hipStream_t st) {
    hipLaunchKernelGGL(pad_c3_c4_kernel, dim3(grid_for(npix)), dim3(256), 0, st, in, npix, out);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}

int preprocess_u8_launch(const uint8_t* in, int N, int Hin, int Win, float* out, int Hout, int Wout, int Hpad, int Wpad, int64_t out_img_stride,
                         const float* mean3, const float* std3, int swap_rb, hipStream_t st) {
    ARG_CHECK(in && out && mean3 && std3, "null");
    ARG_CHECK(N > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0 && Hpad >= Hout && Wpad >= Wout, "preprocess geometry");
    ARG_CHECK(out_img_stride >= (int64_t)Hpad * Wpad * 3, "output image stride");
    PreU8 a;
    a.in = in; a.out = out; a.N = N; a.Hin = Hin; a.Win = Win; a.Hout = Hout; a.Wout = Wout; a.Hpad = Hpad; a.Wpad = Wpad; a.swap = swap_rb ? 1 : 0;
    a.out_img_stride = out_img_stride;
    a.sch = (float)((double)Hin / (double)Hout);  // np.float32(i / o)
    a.scw = (float)((double)Win / (double)Wout);
    for (int c = 0; c < 3; ++c) { a.mean[c] = mean3[c]; a.stdv[c] = std3[c]; }
    hipLaunchKernelGGL(preprocess_u8_kernel, dim3(grid_for((int64_t)N * Hpad * Wpad)), dim3(256), 0, st, a);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}

}  // namespace isegmi

using namespace isegmi;

extern "C" int isegmi_op_preprocess_u8(const uint8_t* d_in, int N, int Hin, int Win, float* d_out, int Hout, int Wout, int Hpad, int Wpad,
                                       int64_t out_img_stride, const float* mean3, const float* std3, int swap_rb, void* stream) {
    return preprocess_u8_launch(d_in, N, Hin, Win, d_out, Hout, Wout, Hpad, Wpad, out_img_stride, mean3, std3, swap_rb, (hipStream_t)stream);
}
extern "C" int isegmi_op_maxpool(const float* d_in, int N, int H, int W, int C, int k, int s, int p, float* d_out,
                                 void* stream) {
    return maxpool_launch(d_in, N, H, W, C, k, s, p, d_out, (hipStream_t)stream);
}
extern "C" int isegmi_op_resize_bilinear(const float* d_in, int N, int H, int W, int C, int Ho, int Wo,
                                         const float* d_add, int relu, float* d_out, void* stream) {
    return resize_bilinear_launch(d_in, N, H, W, C, Ho, Wo, d_add, relu, d_out, (hipStream_t)stream);
}
extern "C" int isegmi_op_deform_im2col(const float* d_x, int N, int H, int W, int C, const float* d_offset_mask, int R, int S,
                                       int stride, int pad, int dil, float* d_out, void* stream) {
    ARG_CHECK(d_x && d_offset_mask && d_out, "null device pointer");
    return deform_im2col_launch(d_x, N, H, W, C, d_offset_mask, R, S, stride, pad, dil, d_out, (hipStream_t)stream);
}
extern "C" int isegmi_op_upsample_nearest2x_add(const float* d_coarse, int N, int Hc, int Wc, int C,
                                                const float* d_lateral, int H, int W, float* d_out, void* stream) {
    return nearest2x_add_launch(d_coarse, N, Hc, Wc, C, d_lateral, H, W, d_out, (hipStream_t)stream);
}
extern "C" int isegmi_op_pad_c3_to_c4(const float* d_in, int64_t npix, float* d_out, void* stream) {
    return pad_c3_c4_launch(d_in, npix, d_out, (hipStream_t)stream);
}
extern "C" int isegmi_op_map_f32(const float* d_x, float* d_y, int64_t n, int fn, void* stream) {
    hipLaunchKernelGGL(map_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, d_x, d_y, n, fn);
    HIP_TRY(hipGetLastError());
    return ISEGMI_OK;
}
