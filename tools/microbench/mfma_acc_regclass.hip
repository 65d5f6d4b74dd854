// mfma_acc_regclass.hip -- does a dependent f32 MFMA chain run at the same rate with its accumulator in VGPRs ("+v") and in AGPRs ("+a")?
// (hipcc puts the accumulators of the 16x16x4 conv kernels into AGPRs and those of the 32x32x2 kernels into VGPRs.)
// Reports shader cycles per dependent MFMA (s_memtime) for one wave per SIMD, and wall ns per MFMA on the pipe with four waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_acc_regclass mfma_acc_regclass.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE>  // 0: 32x32x2 acc in VGPR, 1: 32x32x2 acc in AGPR, 2: 16x16x4 VGPR, 3: 16x16x4 AGPR
__global__ __launch_bounds__(256) void chain(float* out, long long* cyc, int n, float a0, float b0) {
    float a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = a0 + j * 1e-3f + threadIdx.x * 1e-7f; b[j] = b0 * (1.0f + j); }
    f32x16 c16 = {};
    f32x4 c4 = {0, 0, 0, 0};
    const long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE == 0) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c16) : "v"(a[j]), "v"(b[j]));
            if (MODE == 1) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c16) : "v"(a[j]), "v"(b[j]));
            if (MODE == 2) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c4) : "v"(a[j]), "v"(b[j]));
            if (MODE == 3) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(c4) : "v"(a[j]), "v"(b[j]));
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = MODE < 2 ? c16[0] + c16[15] : c4[0] + c4[3];
}

template <int MODE>
static void run(const char* name) {
    float* out; long long* cyc;
    CK(hipMalloc(&out, 1024 * 256 * 4)); CK(hipMalloc(&cyc, 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int n = 4000;
    for (int cfg = 0; cfg < 2; ++cfg) {
        const int blocks = cfg == 0 ? 256 : 1024, threads = cfg == 0 ? 64 : 256;  // one wave per CU / four waves per SIMD
        chain<MODE><<<blocks, threads>>>(out, cyc, 100, 1.0f, 1e-6f);
        CK(hipEventRecord(e0));
        chain<MODE><<<blocks, threads>>>(out, cyc, n, 1.0f, 1e-6f);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        long long c; CK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
        const double waves_per_simd = cfg == 0 ? 1.0 : 4.0;
        printf("%-22s %4d blocks x %3d thr: %6.1f cycles per MFMA of one wave (s_memtime), %6.2f ns per MFMA on the pipe\n", name, blocks, threads,
               (double)c / (n * 8.0), ms * 1e6 / (n * 8.0) / waves_per_simd);
    }
}
int main() {
    run<0>("32x32x2 f32, acc VGPR"); run<1>("32x32x2 f32, acc AGPR"); run<2>("16x16x4 f32, acc VGPR"); run<3>("16x16x4 f32, acc AGPR");
    run<0>("32x32x2 f32, acc VGPR"); run<1>("32x32x2 f32, acc AGPR");
    return 0;
}
