// mfma_clock.hip -- the clock (and so the MFMA rate) the chip actually sustains under a chip-wide matrix load, as a function of
// the operand data.  Every SIMD runs four waves of dependent MFMA chains (the conv kernels' steady state: no memory traffic, no
// LDS, the matrix pipe never idle), so wall time per MFMA = pipe cycles / clock:  GHz = cycles per MFMA / (ns per MFMA on the pipe).
//   constant : a = 1, b = 1e-6 in every lane (what mfma_chain.hip used: almost no switching activity)
//   random   : a, b uniform in [-1, 1], different in every lane and step
//   convlike : a = max(N(0,1), 0) (post-ReLU activations: half zeros), b = N(0, 0.05) (He-initialised weights)
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_clock mfma_clock.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <bool F16>
__global__ __launch_bounds__(256) void chain(const float* __restrict__ ta, const float* __restrict__ tb, float* out, int n, long long* cyc) {
    f32x16 acc = {};
    long long t0 = 0;
    if (F16) {
        f16x8 a[8], b[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                a[j][e] = (_Float16)ta[(threadIdx.x * 64 + j * 8 + e) & 16383];
                b[j][e] = (_Float16)tb[(threadIdx.x * 64 + j * 8 + e + 4096) & 16383];
            }
        t0 = clock64();
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j], b[j], acc, 0, 0, 0);
        }
    } else {
        float a[8], b[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { a[j] = ta[(threadIdx.x * 8 + j) & 16383]; b[j] = tb[(threadIdx.x * 8 + j + 4096) & 16383]; }
        t0 = clock64();
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
        }
    }
    const float r = acc[0] + acc[15];   // (waits for the last MFMA)
    const long long t1 = clock64();
    if (blockIdx.x == 0 && threadIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

static float gauss() { float u = (rand() + 1.0f) / (RAND_MAX + 2.0f), v = (rand() + 1.0f) / (RAND_MAX + 2.0f); return sqrtf(-2.0f * logf(u)) * cosf(6.2831853f * v); }

int main() {
    static float ha[16384], hb[16384];
    float *ta, *tb, *out; long long* dcyc;
    CK(hipMalloc(&ta, sizeof(ha))); CK(hipMalloc(&tb, sizeof(hb))); CK(hipMalloc(&out, 1024 * 256 * 4)); CK(hipMalloc(&dcyc, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char* names[3] = {"constant", "random", "convlike"};
    for (int f16 = 0; f16 < 2; ++f16)
        for (int mode = 0; mode < 3; ++mode) {
            srand(1);
            for (int i = 0; i < 16384; ++i) {
                if (mode == 0) { ha[i] = 1.0f; hb[i] = 1e-6f; }
                else if (mode == 1) { ha[i] = rand() / (float)RAND_MAX * 2.0f - 1.0f; hb[i] = rand() / (float)RAND_MAX * 2.0f - 1.0f; }
                else { const float g = gauss(); ha[i] = g > 0.0f ? g : 0.0f; hb[i] = gauss() * 0.05f; }
            }
            CK(hipMemcpy(ta, ha, sizeof(ha), hipMemcpyHostToDevice)); CK(hipMemcpy(tb, hb, sizeof(hb), hipMemcpyHostToDevice));
            for (int rep = 0; rep < 2; ++rep) {
                const int n = (rep == 0 ? 2000 : 20000) * (f16 ? 2 : 1);  // ~2 ms and ~20 ms kernels: does the clock sag with duration?
                if (f16) chain<true><<<1024, 256>>>(ta, tb, out, 200, dcyc); else chain<false><<<1024, 256>>>(ta, tb, out, 200, dcyc);
                CK(hipEventRecord(e0));
                if (f16) chain<true><<<1024, 256>>>(ta, tb, out, n, dcyc); else chain<false><<<1024, 256>>>(ta, tb, out, n, dcyc);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double ns_pipe = ms * 1e6 / (n * 8.0) / 4.0;  // four waves share a SIMD's pipe
                const double cyc = f16 ? 32.0 : 64.0;                // v_mfma_f32_32x32x16_f16: 8 passes; v_mfma_f32_32x32x2_f32: 16 passes
                const double flop = f16 ? 32768.0 : 4096.0;
                long long c; CK(hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost));
                printf("%s %-8s operands, %6d x 8 MFMAs per wave: %8.3f ms, %6.1f cycles per MFMA of one wave (s_memtime; %.0f if the four waves of a SIMD take turns) -> %.3f GHz sustained, %7.1f TF/s (nominal %s)\n", f16 ? "f16 32x32x16" : "f32 32x32x2 ",
                       names[mode], n, ms, (double)c / (n * 8.0), 4.0 * cyc, (double)c / (n * 8.0) / (ms * 1e6 / (n * 8.0)), 1024.0 * flop / ns_pipe / 1e3, f16 ? "2516" : "157.3");
            }
        }
    return 0;
}
