// mfma_chain.hip -- latency of a DEPENDENT chain of f32 MFMAs (one accumulator, as the bit-exact conv requires), in shader
// cycles (clock64) and in wall time, for one wave per CU and for a full grid.  build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ void chain(float* out, long long* cyc, int n, float a0, float b0) {
    float a = a0 + threadIdx.x * 1e-9f, b = b0;
    f32x4 acc4 = {0, 0, 0, 0};
    f32x16 acc16 = {};
    const long long t0 = clock64();
    for (int i = 0; i < n; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4, 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc16 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc16, 0, 0, 0);
        }
    }
    const long long t1 = clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = MODE == 0 ? acc4[0] + acc4[3] : acc16[0] + acc16[15];
}

template <int MODE>
static void run(const char* name, int blocks, int threads) {
    float* out; long long* cyc; hipMalloc(&out, (size_t)blocks * threads * 4); hipMalloc(&cyc, 8);
    const int n = 2000;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    chain<MODE><<<blocks, threads>>>(out, cyc, 100, 1.0f, 1e-6f);
    hipEventRecord(a);
    chain<MODE><<<blocks, threads>>>(out, cyc, n, 1.0f, 1e-6f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-14s %4d blocks x %3d thr: %7.1f cycles / dependent MFMA (clock64), %6.1f ns / MFMA wall  -> clock ratio %.2f GHz-equivalent\n", name, blocks,
           threads, (double)c / (n * 8.0), ms * 1e6 / (n * 8.0), (double)c / (ms * 1e6));
    hipFree(out); hipFree(cyc);
}
int main() {
    run<0>("16x16x4 f32", 1, 64); run<0>("16x16x4 f32", 256, 64); run<0>("16x16x4 f32", 1024, 256);
    run<1>("32x32x2 f32", 1, 64); run<1>("32x32x2 f32", 256, 64); run<1>("32x32x2 f32", 1024, 256);
    return 0;
}
