// mfma_wait.hip -- what an s_waitcnt / s_barrier / LDS access costs when it sits inside a dependent f32 MFMA chain (the conv kernels'
// inner loop is exactly that: 16 dependent v_mfma_f32_32x32x2_f32 per 32-k chunk with fragment reads, their waits and one barrier).
// Each variant adds K "extras" per 8 MFMAs to the same loop; reported: ns per MFMA on the pipe and the extra pipe cycles per extra.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_wait mfma_wait.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <int V, bool AG = false, bool M16 = false>
__global__ __launch_bounds__(256) void chain(float* out, int n) {
    __shared__ f32x4 lds[256];
    float a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = 1.0f + j * 1e-3f + threadIdx.x * 1e-7f; b[j] = 1e-6f * (1.0f + j); }
    lds[threadIdx.x] = f32x4{a[0], a[1], a[2], a[3]};
    __syncthreads();
    float pre = 0.0f;
    if (V == 30 || V == 31 || V == 32) {  // the wave HAS used vector memory before the loop: one global load, landed and consumed
        pre = out[(blockIdx.x * blockDim.x + threadIdx.x) & 65535];
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(pre) :: "memory");
        a[0] += pre * 0.0f;
    }
    f32x16 acc = {};
    f32x4 acc4 = {0, 0, 0, 0};
    f32x4 fr = {0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool at = (j == 0 || j == 4);
            if (V == 1 && at) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (V == 2 && at) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (V == 3 && at) asm volatile("s_nop 0" ::: "memory");
            if (V == 4 && j == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (V == 5 && j == 0) asm volatile("s_barrier" ::: "memory");
            if (V == 6 && at) { fr = lds[(threadIdx.x + j) & 255]; asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fr) :: "memory"); a[j] += fr[0] * 0.0f; }
            if (V == 7 && at) { fr = lds[(threadIdx.x + j) & 255]; }                    // LDS read, consumed 4 MFMAs later (counted wait placed by hipcc)
            if (V == 7 && (j == 3 || j == 7)) a[(j + 1) & 7] += fr[0] * 0.0f;
            if (V == 8 && at) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if (V == 9 && at) asm volatile("s_setprio 1\n\ts_setprio 0" ::: "memory");
            if (V == 10 && at) asm volatile("v_mov_b32 %0, %0" : "+v"(fr[1]));
            if (V == 30 && at) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (V == 31 && at) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (V >= 20) {
                if ((V == 21) && at) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if ((V == 22) && at) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if ((V == 23) && at) asm volatile("s_waitcnt vmcnt(0)");
                if (M16) acc4 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc4, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
            } else if (M16) {
                if (AG) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc4) : "v"(a[j]), "v"(b[j]));
                else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc4) : "v"(a[j]), "v"(b[j]));
            } else {
                if (AG) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a[j]), "v"(b[j]));
                else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a[j]), "v"(b[j]));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0] + acc[15] + fr[1] + acc4[0] + acc4[3];
}
static double base_ns[2] = {0, 0};
template <int V, bool AG = false, bool M16 = false>
static void run(const char* name, int extras, float* out) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int n = 4000;
    for (int cfg = 0; cfg < 2; ++cfg) {
        const int blocks = cfg == 0 ? 256 : 1024;  // one wave per SIMD / four waves per SIMD
        chain<V, AG, M16><<<blocks, 256>>>(out, 100);
        CK(hipEventRecord(e0));
        chain<V, AG, M16><<<blocks, 256>>>(out, n);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double ns = ms * 1e6 / (n * 8.0) / (cfg == 0 ? 1.0 : 4.0);
        if (V == 0) base_ns[cfg] = ns;
        printf("%-64s %d wave(s)/SIMD: %6.2f ns per MFMA on the pipe", name, cfg == 0 ? 1 : 4, ns);
        if (extras) printf("  -> +%5.0f pipe cycles per extra (at 2.35 GHz)", (ns - base_ns[cfg]) * 8.0 / extras * 2.35);
        printf("\n");
    }
}
int main() {
    float* out; CK(hipMalloc(&out, 1024 * 256 * 4));
    run<0>("V0  8 dependent MFMAs per iteration, nothing else", 0, out);
    run<3>("V3  + 2 x s_nop 0", 2, out);
    run<10>("V10 + 2 x v_mov_b32 (independent VALU)", 2, out);
    run<1>("V1  + 2 x s_waitcnt vmcnt(0) (nothing outstanding)", 2, out);
    run<4>("V4  + 1 x s_waitcnt vmcnt(0)", 1, out);
    run<2>("V2  + 2 x s_waitcnt lgkmcnt(0) (nothing outstanding)", 2, out);
    run<8>("V8  + 2 x s_waitcnt vmcnt(0) lgkmcnt(0)", 2, out);
    run<9>("V9  + 2 x (s_setprio 1; s_setprio 0)", 2, out);
    run<5>("V5  + 1 x s_barrier", 1, out);
    run<6>("V6  + 2 x (ds_read_b128; s_waitcnt lgkmcnt(0)) right before an MFMA", 2, out);
    run<7>("V7  + 2 x ds_read_b128 consumed four MFMAs later", 2, out);
    run<0>("V0  again", 0, out);
    printf("---- accumulator in AGPRs\n");
    run<0, true>("A0  8 dependent MFMAs, acc in AGPRs", 0, out);
    run<1, true>("A1  + 2 x s_waitcnt vmcnt(0)", 2, out);
    run<2, true>("A2  + 2 x s_waitcnt lgkmcnt(0)", 2, out);
    run<5, true>("A5  + 1 x s_barrier", 1, out);
    run<6, true>("A6  + 2 x (ds_read_b128; s_waitcnt lgkmcnt(0))", 2, out);
    run<3, true>("A3  + 2 x s_nop 0", 2, out);
    printf("---- builtin MFMAs (hipcc allocates the accumulator; AGPRs here)\n");
    run<20>("B20 8 dependent builtin MFMAs", 0, out);
    run<21>("B21 + 2 x s_waitcnt vmcnt(0) with a memory clobber", 2, out);
    run<23>("B23 + 2 x s_waitcnt vmcnt(0) without a memory clobber", 2, out);
    run<22>("B22 + 2 x s_waitcnt lgkmcnt(0) with a memory clobber", 2, out);
    run<32>("B32 builtin MFMAs after one global load before the loop", 0, out);
    run<30>("B30 = B32 + 2 x s_waitcnt vmcnt(0) inside the loop", 2, out);
    run<31>("B31 = B32 + 2 x s_waitcnt lgkmcnt(0) inside the loop", 2, out);
    run<20, false, true>("B20m 8 dependent builtin 16x16x4 MFMAs", 0, out);
    run<22, false, true>("B22m + 2 x s_waitcnt lgkmcnt(0), 16x16x4", 2, out);
    printf("---- v_mfma_f32_16x16x4_f32, acc in VGPRs / AGPRs\n");
    run<0, false, true>("M0  8 dependent 16x16x4 MFMAs, VGPR", 0, out);
    run<2, false, true>("M2  + 2 x s_waitcnt lgkmcnt(0), VGPR", 2, out);
    run<6, false, true>("M6  + 2 x (ds_read_b128; wait), VGPR", 2, out);
    run<0, true, true>("N0  8 dependent 16x16x4 MFMAs, AGPR", 0, out);
    run<2, true, true>("N2  + 2 x s_waitcnt lgkmcnt(0), AGPR", 2, out);
    run<6, true, true>("N6  + 2 x (ds_read_b128; wait), AGPR", 2, out);
    run<5, true, true>("N5  + 1 x s_barrier, AGPR", 1, out);
    return 0;
}
