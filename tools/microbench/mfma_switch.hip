// mfma_switch.hip -- does the matrix pipe lose cycles when consecutive MFMAs come from DIFFERENT waves of a SIMD?
// Each wave runs a dependent chain of v_mfma_f32_32x32x2_f32; between two MFMAs it executes D dependent v_add_f32 (its own VALU work,
// independent of the accumulator), so that its next MFMA is not ready the moment the pipe frees and a co-resident wave takes the slot:
// with D = 0 the oldest wave keeps the pipe (age-ordered arbitration), with D large the waves take turns.  If a hand-over were free
// the pipe would stay at 64 cycles per MFMA as long as some wave is ready.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_switch mfma_switch.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <int D, int KIND = 0>
__global__ __launch_bounds__(256) void chain(float* out, int n) {
    __shared__ float lds[1024];
    lds[threadIdx.x] = threadIdx.x; __syncthreads();
    float w[8] = {1, 2, 3, 4, 5, 6, 7, 8}; int sreg = n;
    float a = 1.0f + threadIdx.x * 1e-7f, b = 1e-6f, v = threadIdx.x * 1e-3f;
    f32x16 acc = {};
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#pragma unroll
            for (int d = 0; d < D; ++d) {
                if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v) : "v"(b));                       // dependent VALU chain
                if (KIND == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(w[d & 7]) : "v"(b));                // eight independent VALU chains
                if (KIND == 2) asm volatile("s_add_i32 %0, %0, 1" : "+s"(sreg));                              // SALU
                if (KIND == 3) asm volatile("ds_read_b32 %0, %1" : "=v"(w[d & 7]) : "v"((int)(threadIdx.x * 4)));  // LDS reads, never awaited inside the loop
                if (KIND == 4) asm volatile("v_mov_b32 %0, %1" : "=v"(w[d & 7]) : "v"(b));                   // independent moves
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[15] + v + w[0] + w[1] + w[2] + w[3] + w[4] + w[5] + w[6] + w[7] + sreg;
}
template <int D, int KIND = 0>
static void run(float* out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int n = 2000;
    double ns[3];
    const int blocks[3] = {256, 512, 1024};  // 1, 2, 4 waves per SIMD
    for (int c = 0; c < 3; ++c) {
        chain<D, KIND><<<blocks[c], 256>>>(out, 50);
        CK(hipEventRecord(e0));
        chain<D, KIND><<<blocks[c], 256>>>(out, n);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        ns[c] = ms * 1e6 / (n * 8.0) / (blocks[c] / 256);
    }
    static const char* kn[5] = {"dependent v_add_f32", "independent v_add_f32 (8 chains)", "s_add_i32", "ds_read_b32 (not awaited)", "independent v_mov_b32"};
    printf("D = %3d x %-34s between MFMAs: ns per MFMA on the pipe with 1 / 2 / 4 waves per SIMD: %6.2f / %6.2f / %6.2f\n", D, kn[KIND], ns[0], ns[1], ns[2]);
}
int main() {
    float* out; CK(hipMalloc(&out, 1024 * 256 * 4));
    run<0>(out); run<4>(out); run<8>(out); run<16>(out); run<32>(out);
    run<4, 1>(out); run<8, 1>(out); run<16, 1>(out); run<32, 1>(out);
    run<8, 4>(out); run<16, 4>(out);
    run<8, 2>(out); run<16, 2>(out); run<32, 2>(out);
    run<4, 3>(out); run<8, 3>(out); run<16, 3>(out);
    run<0>(out);
    return 0;
}
