// Is v_mfma_f32_32x32x16_f16 bitwise symmetric under D = A B  <->  D^T = B^T A^T ?  (dev tool, round 4: the fused bottleneck computes
// conv1 / conv2 transposed; bit-equality with the three-launch path needs this symmetry.)  hipcc --offload-arch=gfx950 -O3 mfma_sym.hip -o mfma_sym
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// A [32][16] row-major, B^T [32][16] row-major (B^T[j][k] = B[k][j]); C [32][32]; chains `steps` MFMAs over K = 16 * steps
__global__ void k(const _Float16* A, const _Float16* BT, const float* C, float* D1, float* D2T, int steps) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    f32x16 c1, c2;
    for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
        c1[e] = C[row * 32 + r];       // D1[row][col = r]
        c2[e] = C[r * 32 + row];       // D2T[row' = row][col' = r] = D[r][row]
    }
    for (int s = 0; s < steps; ++s) {
        f16x8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = A[(s * 32 + r) * 16 + 8 * h + j]; b[j] = BT[(s * 32 + r) * 16 + 8 * h + j]; }
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);   // D = A B
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c2, 0, 0, 0);   // D^T = B^T A^T
    }
    for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * h;
        D1[row * 32 + r] = c1[e];
        D2T[row * 32 + r] = c2[e];
    }
}
int main() {
    const int steps = 16, trials = 200;
    _Float16 *A, *BT; float *C, *D1, *D2;
    hipMallocManaged(&A, steps * 32 * 16 * 2); hipMallocManaged(&BT, steps * 32 * 16 * 2); hipMallocManaged(&C, 4096); hipMallocManaged(&D1, 4096); hipMallocManaged(&D2, 4096);
    srand(1);
    long diff = 0, total = 0; double maxrel = 0;
    for (int t = 0; t < trials; ++t) {
        for (int i = 0; i < steps * 512; ++i) {
            float u = (rand() / (float)RAND_MAX) * 2 - 1, v = (rand() / (float)RAND_MAX) * 2 - 1;
            A[i] = (_Float16)(t % 2 ? fmaxf(u * 2, 0.f) : u);   // odd trials: ReLU-like activations
            BT[i] = (_Float16)(v * 0.1f);
        }
        for (int i = 0; i < 1024; ++i) C[i] = t % 4 == 0 ? 0.f : (rand() / (float)RAND_MAX) - 0.5f;
        k<<<1, 64>>>(A, BT, C, D1, D2, steps);
        hipDeviceSynchronize();
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            const float x = D1[i * 32 + j], y = D2[j * 32 + i];
            ++total;
            if (x != y) { ++diff; double r = fabs(x - y) / fmax(fabs(x), 1e-30); if (r > maxrel) maxrel = r; }
        }
    }
    printf("mfma_f32_32x32x16_f16: D = A B vs (B^T A^T)^T over %d trials, K = %d: %ld of %ld elements differ (max rel %.3g)\n", trials, 16 * steps, diff, total, maxrel);
    return 0;
}
