// mfma_shape.hip -- v_mfma_f32_32x32x16_f16 against v_mfma_f32_16x16x32_f16 (round 5).
//  (1) numerics: is a 32-deep product summed by ONE 16x16x32 instruction bitwise what TWO chained 32x32x16 instructions give (k 0..15, then 16..31)?
//      If yes every fused-vs-unfused bit-identity of the fp16 family survives a change of shape; if no, the layers that change shape change association.
//  (2) wall time on RANDOM data at the same 64 x 64 output tile per wave, operands re-read from LDS by ds_read_b128 every step (the conv kernels'
//      steady state: 12 waves per CU = 3 per SIMD): MI355X_MICROARCH.md "DVFS give-back" 7 reports ~1.12-1.15x for the 16x16x32 form at equal cycles.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_shape mfma_shape.hip
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 half_t;
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// ---- (1) numerics.  A [32][K] row-major, BT [32][K] (BT[j][k] = B[k][j]), C [32][32]; K = 32 * steps.
__global__ void eq_kernel(const half_t* A, const half_t* BT, const float* C, float* D32, float* D16, int K) {
    const int lane = threadIdx.x;
    {   // 32x32x16, two instructions per 32 k
        const int r = lane & 31, h = lane >> 5;
        f32x16 c;
        for (int e = 0; e < 16; ++e) c[e] = C[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r];
        for (int k0 = 0; k0 < K; k0 += 16) {
            f16x8 a, b;
            for (int j = 0; j < 8; ++j) { a[j] = A[r * K + k0 + 8 * h + j]; b[j] = BT[r * K + k0 + 8 * h + j]; }
            c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
        }
        for (int e = 0; e < 16; ++e) D32[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r] = c[e];
    }
    {   // 16x16x32, four 16 x 16 tiles, one instruction per 32 k
        const int l15 = lane & 15, q = lane >> 4;
        for (int ti = 0; ti < 2; ++ti)
            for (int tj = 0; tj < 2; ++tj) {
                f32x4 c;
                for (int e = 0; e < 4; ++e) c[e] = C[(16 * ti + 4 * q + e) * 32 + 16 * tj + l15];
                for (int k0 = 0; k0 < K; k0 += 32) {
                    f16x8 a, b;
                    for (int j = 0; j < 8; ++j) { a[j] = A[(16 * ti + l15) * K + k0 + 8 * q + j]; b[j] = BT[(16 * tj + l15) * K + k0 + 8 * q + j]; }
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
                }
                for (int e = 0; e < 4; ++e) D16[(16 * ti + 4 * q + e) * 32 + 16 * tj + l15] = c[e];
            }
    }
}

// ---- (2) wall time.  Each wave owns a 64 x 64 output tile; per 64-deep chunk it reads its 64 A rows and 64 B rows (x 64 halfs) from an LDS image
// (rows of 128 B, 16-B columns XOR-swizzled by (row >> 1) & 7, as the conv kernels stage them) and multiplies.  No global traffic in the loop.
template <int MS>
__global__ __launch_bounds__(768) void tile_kernel(const half_t* src, float* out, int nchunks) {
    extern __shared__ __attribute__((aligned(1024))) char lds[];   // [A 192 rows][B 256 rows] x 128 B, two stages
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / 4, wn = wave % 4;
    for (int i = tid; i < 2 * 448 * 8; i += 768) ((uint4*)lds)[i] = ((const uint4*)src)[(i + blockIdx.x * 64) % (2 * 448 * 8)];
    __syncthreads();
    float sum = 0.f;
    if (MS == 0) {
        const int lr = lane & 31, lh = lane >> 5;
        const int swz = lr * 128 + ((lh ^ ((lr >> 1) & 7)) << 4);
        f32x16 acc[2][2] = {};
        for (int t = 0; t < nchunks; ++t) {
            const char* sb = lds + (t & 1) * 448 * 128;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                f16x8 fa[2], fb[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) fa[a] = *(const f16x8*)(sb + (wm * 64 + a * 32) * 128 + (swz ^ (s << 5)));
#pragma unroll
                for (int b = 0; b < 2; ++b) fb[b] = *(const f16x8*)(sb + (192 + wn * 64 + b * 32) * 128 + (swz ^ (s << 5)));
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[a], fb[b], acc[a][b], 0, 0, 0);
            }
        }
        for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int e = 0; e < 16; ++e) sum += acc[a][b][e];
    } else {
        const int l15 = lane & 15, q = lane >> 4;
        const int swz = l15 * 128 + ((q ^ ((l15 >> 1) & 7)) << 4);   // 32-deep step s adds ^ (s << 6)
        f32x4 acc[4][4] = {};
        for (int t = 0; t < nchunks; ++t) {
            const char* sb = lds + (t & 1) * 448 * 128;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                f16x8 fa[4], fb[4];
#pragma unroll
                for (int a = 0; a < 4; ++a) fa[a] = *(const f16x8*)(sb + (wm * 64 + a * 16) * 128 + (swz ^ (s << 6)));
#pragma unroll
                for (int b = 0; b < 4; ++b) fb[b] = *(const f16x8*)(sb + (192 + wn * 64 + b * 16) * 128 + (swz ^ (s << 6)));
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[a], fb[b], acc[a][b], 0, 0, 0);
            }
        }
        for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) for (int e = 0; e < 4; ++e) sum += acc[a][b][e];
    }
    out[blockIdx.x * 768 + tid] = sum;
}

int main() {
    {   // (1)
        const int K = 256, trials = 200;
        half_t *A, *BT; float *C, *D1, *D2;
        CK(hipMallocManaged(&A, 32 * K * 2)); CK(hipMallocManaged(&BT, 32 * K * 2)); CK(hipMallocManaged(&C, 4096)); CK(hipMallocManaged(&D1, 4096)); CK(hipMallocManaged(&D2, 4096));
        srand(1);
        long diff = 0, total = 0; double maxrel = 0;
        for (int t = 0; t < trials; ++t) {
            for (int i = 0; i < 32 * K; ++i) {
                const float u = (rand() / (float)RAND_MAX) * 2 - 1, v = (rand() / (float)RAND_MAX) * 2 - 1;
                A[i] = (half_t)(t % 2 ? fmaxf(u * 2, 0.f) : u);
                BT[i] = (half_t)(v * 0.1f);
            }
            for (int i = 0; i < 1024; ++i) C[i] = t % 4 == 0 ? 0.f : (rand() / (float)RAND_MAX) - 0.5f;
            eq_kernel<<<1, 64>>>(A, BT, C, D1, D2, K);
            CK(hipDeviceSynchronize());
            for (int i = 0; i < 1024; ++i) {
                ++total;
                if (D1[i] != D2[i]) { ++diff; const double r = fabs(D1[i] - D2[i]) / fmax(fabs(D1[i]), 1e-30); if (r > maxrel) maxrel = r; }
            }
        }
        printf("numerics: 32x32x16 (two per 32 k) vs 16x16x32 (one per 32 k), K = %d, %d trials: %ld of %ld elements differ (max rel %.3g)\n", K, trials, diff, total, maxrel);
    }
    {   // (2)
        const int lds = 2 * 448 * 128, blocks = 256, nchunks = 20000;
        half_t* src; float* out;
        CK(hipMalloc(&src, lds)); CK(hipMalloc(&out, blocks * 768 * 4));
        half_t* h = (half_t*)malloc(lds);
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipFuncSetAttribute((const void*)tile_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        CK(hipFuncSetAttribute((const void*)tile_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        const char* names[3] = {"zeros", "random [-1,1)", "convlike (ReLU'd activations x N(0, 0.05) weights)"};
        for (int mode = 0; mode < 3; ++mode) {
            srand(2);
            for (int i = 0; i < lds / 2; ++i) {
                const float u = (rand() / (float)RAND_MAX) * 2 - 1;
                const bool isA = (i % (448 * 64)) < 192 * 64;
                h[i] = (half_t)(mode == 0 ? 0.f : mode == 1 ? u : isA ? fmaxf(u * 2, 0.f) : u * 0.05f);
            }
            CK(hipMemcpy(src, h, lds, hipMemcpyHostToDevice));
            double ms_of[2][3];
            for (int rep = 0; rep < 3; ++rep)      // interleaved rounds in one process
                for (int ms = 0; ms < 2; ++ms) {
                    if (ms == 0) tile_kernel<0><<<blocks, 768, lds>>>(src, out, 200); else tile_kernel<1><<<blocks, 768, lds>>>(src, out, 200);
                    CK(hipEventRecord(e0));
                    if (ms == 0) tile_kernel<0><<<blocks, 768, lds>>>(src, out, nchunks); else tile_kernel<1><<<blocks, 768, lds>>>(src, out, nchunks);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    float t; CK(hipEventElapsedTime(&t, e0, e1));
                    ms_of[ms][rep] = t;
                }
            const double flop = (double)blocks * 12 * nchunks * 2.0 * 64 * 64 * 64;
            for (int ms = 0; ms < 2; ++ms) {
                double best = ms_of[ms][0], med = ms_of[ms][1];
                for (int r = 0; r < 3; ++r) if (ms_of[ms][r] < best) best = ms_of[ms][r];
                printf("%-52s %s: %.2f / %.2f / %.2f ms  -> %.0f TF/s (best)\n", names[mode], ms ? "16x16x32" : "32x32x16", ms_of[ms][0], ms_of[ms][1], ms_of[ms][2], flop / (best * 1e-3) / 1e12);
                (void)med;
            }
        }
    }
    return 0;
}
