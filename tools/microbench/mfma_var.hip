// mfma_var.hip -- which property of a chip-wide dependent f32 MFMA chain sets its wall time per MFMA (dev experiment)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <int V>
__global__ __launch_bounds__(256) void chain(const float* __restrict__ ta, const float* __restrict__ tb, float* out, int n, unsigned long long* clk) {
    float a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (V == 0 || V == 1) { a[j] = 1.0f + j * 1e-3f + threadIdx.x * 1e-7f; b[j] = 1e-6f * (1.0f + j); }
        if (V == 2 || V == 7) { a[j] = 1.0f; b[j] = 1e-6f; asm volatile("" : "+v"(a[j]), "+v"(b[j])); }
        if (V >= 3) { a[j] = ta[(threadIdx.x * 8 + j) & 16383]; b[j] = tb[(threadIdx.x * 8 + j + 4096) & 16383]; }
    }
    if (V == 8 || V == 9) {  // the loads have landed BEFORE the loop: no s_waitcnt inside it
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(a[j]), "+v"(b[j]));
    }
    f32x16 acc = {};
    // last-dispatched block: its wave 0 runs for most of the kernel (age-ordered arbitration serves the oldest waves first)
    const unsigned long long c0 = clock64(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (V == 0) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a[j]), "v"(b[j]));
            else if (V == 7) {  // V2's loop with two redundant s_waitcnt in it (what hipcc leaves in V3's loop)
                if (j == 0) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                if (j == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
            }
            else if (V == 5) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[(j + 1) & 7], acc, 0, 0, 0);
            else if (V == 6) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[(j + 2) & 7], acc, 0, 0, 0);
            else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
        }
    }
    const float r = acc[0] + acc[15];
    const unsigned long long c1 = clock64(), r1 = __builtin_amdgcn_s_memrealtime();
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int V>
static void run(const char* name, const float* ta, const float* tb, float* out) {
    static unsigned long long* clk = nullptr; if (!clk) CK(hipMalloc(&clk, 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int n = 4000;
    chain<V><<<1024, 256>>>(ta, tb, out, 100, clk);
    CK(hipEventRecord(e0));
    chain<V><<<1024, 256>>>(ta, tb, out, n, clk);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    float h[4]; CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
    unsigned long long hc[2]; CK(hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost));
    printf("%-58s %7.3f ms  %6.2f ns per MFMA on the pipe; shader clock while the last block ran: %.3f GHz (s_memtime / s_memrealtime at 100 MHz)\n", name, ms,
           ms * 1e6 / (n * 8.0) / 4.0, (double)hc[0] / (double)hc[1] * 0.1);
}
int main() {
    static float ha[16384], hb[16384];
    float *ta, *tb, *out;
    CK(hipMalloc(&ta, sizeof(ha))); CK(hipMalloc(&tb, sizeof(hb))); CK(hipMalloc(&out, 1024 * 256 * 4));
    for (int pass = 0; pass < 2; ++pass) {
        for (int i = 0; i < 16384; ++i) { ha[i] = 1.0f; hb[i] = 1e-6f; }
        CK(hipMemcpy(ta, ha, sizeof(ha), hipMemcpyHostToDevice)); CK(hipMemcpy(tb, hb, sizeof(hb), hipMemcpyHostToDevice));
        run<0>("V0 asm MFMA, a = 1 + eps(lane, j), b = 1e-6 (1 + j)", ta, tb, out);
        run<1>("V1 builtin MFMA, same operands", ta, tb, out);
        run<2>("V2 builtin, a = 1.0, b = 1e-6 exactly in every lane", ta, tb, out);
        run<3>("V3 builtin, the same constants loaded from a table", ta, tb, out);
        run<7>("V7 = V2 + two redundant s_waitcnt vmcnt inside the loop", ta, tb, out);
        run<8>("V8 = V3, loads awaited before the loop (no s_waitcnt in it)", ta, tb, out);
        srand(1);
        for (int i = 0; i < 16384; ++i) { ha[i] = rand() / (float)RAND_MAX * 2.0f - 1.0f; hb[i] = rand() / (float)RAND_MAX * 2.0f - 1.0f; }
        CK(hipMemcpy(ta, ha, sizeof(ha), hipMemcpyHostToDevice)); CK(hipMemcpy(tb, hb, sizeof(hb), hipMemcpyHostToDevice));
        run<4>("V4 builtin, uniform random operands in [-1, 1]", ta, tb, out);
        run<9>("V9 random operands, no s_waitcnt inside the loop", ta, tb, out);
        for (int i = 0; i < 16384; ++i) { ha[i] *= 1e-3f; hb[i] *= 1e-3f; }
        CK(hipMemcpy(ta, ha, sizeof(ha), hipMemcpyHostToDevice)); CK(hipMemcpy(tb, hb, sizeof(hb), hipMemcpyHostToDevice));
        run<4>("V4 builtin, uniform random operands in [-1e-3, 1e-3]", ta, tb, out);
    }
    return 0;
}
