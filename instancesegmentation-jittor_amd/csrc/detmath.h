// detmath.h -- deterministic fp32 transcendental functions for device code.
//
// The Cephes single-precision polynomials written with explicit fmaf / IEEE
// mul, add, div so the result is a pure function of the input bits (no libm /
// ocml fast paths).  The oracle (oracle/ora_ops.c) carries its own independent
// restatement of the same published algorithm; tests compare the two bitwise.
// Compile with -ffp-contract=off: only the fmaf() written here may fuse.
#pragma once
#include <hip/hip_runtime.h>

namespace isegmi {

__device__ __forceinline__ float dm_div(float a, float b) { return __fdiv_rn(a, b); }
__device__ __forceinline__ float dm_sqrt(float a) { return __fsqrt_rn(a); }

__device__ __forceinline__ float dm_exp(float x) {
    if (x != x) return x;
    if (x > 88.3762626647949f) x = 88.3762626647949f;
    if (x < -87.3f) return 0.0f;
    const float fx = floorf(fmaf(x, 1.44269504088896341f, 0.5f));
    float r = fmaf(fx, -0.693359375f, x);
    r = fmaf(fx, 2.12194440e-4f, r);
    const float z = r * r;
    float p = 1.9875691500E-4f;
    p = fmaf(p, r, 1.3981999507E-3f);
    p = fmaf(p, r, 8.3334519073E-3f);
    p = fmaf(p, r, 4.1665795894E-2f);
    p = fmaf(p, r, 1.6666665459E-1f);
    p = fmaf(p, r, 5.0000001201E-1f);
    float y = fmaf(p, z, r);
    y = y + 1.0f;
    const int n = (int)fx;
    return y * __uint_as_float((unsigned)(n + 127) << 23);
}

__device__ __forceinline__ float dm_sigmoid(float x) {
    const float e = dm_exp(-x);
    return dm_div(1.0f, 1.0f + e);
}

__device__ __forceinline__ float dm_tanh(float x) {
    const float z = fabsf(x);
    if (z >= 0.625f) {
        float r;
        if (z > 44.0f) r = 1.0f;
        else {
            const float e = dm_exp(z + z);
            r = 1.0f - dm_div(2.0f, e + 1.0f);
        }
        return x < 0.0f ? -r : r;
    }
    const float s = x * x;
    float p = -5.70498872745E-3f;
    p = fmaf(p, s, 2.06390887954E-2f);
    p = fmaf(p, s, -5.37397155531E-2f);
    p = fmaf(p, s, 1.33314422036E-1f);
    p = fmaf(p, s, -3.33332819422E-1f);
    const float t = p * s;
    return fmaf(t, x, x);
}

// log2 of a positive normal float
__device__ __forceinline__ float dm_log2(float x) {
    const unsigned b = __float_as_uint(x);
    int e = (int)((b >> 23) & 255u) - 126;
    float m = __uint_as_float((b & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; }
    else m = m - 1.0f;
    const float z = m * m;
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
    const float ln = m + y;
    return fmaf(ln, 1.44269504088896341f, (float)e);
}

// bilinear source coordinate, align_corners=False (matches oracle bil_coef); `scale` = dm_div(in_sz, out_sz), hoistable
__device__ __forceinline__ void dm_bil_coef_s(int dst, int in_sz, float scale, int& i0, int& i1, float& l0, float& l1) {
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.0f) src = 0.0f;
    int a = (int)src;
    if (a > in_sz - 1) a = in_sz - 1;
    i0 = a;
    i1 = a < in_sz - 1 ? a + 1 : a;
    l1 = src - (float)a;
    l0 = 1.0f - l1;
}
__device__ __forceinline__ void dm_bil_coef(int dst, int in_sz, int out_sz, int& i0, int& i1, float& l0,
                                            float& l1) {
    dm_bil_coef_s(dst, in_sz, dm_div((float)in_sz, (float)out_sz), i0, i1, l0, l1);
}

}  // namespace isegmi
