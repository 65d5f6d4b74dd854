// grid_sync.hip -- what a grid-wide barrier costs on this chip against a kernel boundary (round 6: would a CHAIN of small dependent convolutions in one
// cooperative launch beat one launch per layer at bs = 1?).  hipcc --offload-arch=gfx950 -O3 tools/microbench/grid_sync.hip -o tools/microbench/grid_sync
//   A: N empty dependent launches on one stream (the floor of a launch boundary);  B: one cooperative launch with N cooperative_groups grid syncs;
//   C: the same with a hand-written barrier (agent-scope atomic counter + fences), the form a conv chain would use between layers.
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <stdio.h>
#include <chrono>
namespace cg = cooperative_groups;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 1000000) *p = 1; }

__global__ void coop_kernel(int n, float* data) {
    cg::grid_group g = cg::this_grid();
    float v = data[blockIdx.x * blockDim.x + threadIdx.x];
    for (int i = 0; i < n; ++i) { v = v * 1.0001f + 1.0f; data[blockIdx.x * blockDim.x + threadIdx.x] = v; g.sync(); v += data[((blockIdx.x + 1) % gridDim.x) * blockDim.x + threadIdx.x]; }
    data[blockIdx.x * blockDim.x + threadIdx.x] = v;
}

// sense-reversing barrier on one counter: arrive with release, spin with acquire (agent scope: the eight XCDs' L2s are not coherent with each other)
__device__ __forceinline__ void grid_barrier(unsigned* ctr, unsigned& phase, unsigned nblocks) {
    __syncthreads();
    if (threadIdx.x == 0) {
        phase += nblocks;
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < phase) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}
__global__ void manual_kernel(int n, float* data, unsigned* ctr) {
    unsigned phase = 0;
    float v = data[blockIdx.x * blockDim.x + threadIdx.x];
    for (int i = 0; i < n; ++i) {
        v = v * 1.0001f + 1.0f; data[blockIdx.x * blockDim.x + threadIdx.x] = v;
        __threadfence();
        grid_barrier(ctr, phase, gridDim.x);
        v += __builtin_nontemporal_load(&data[((blockIdx.x + 1) % gridDim.x) * blockDim.x + threadIdx.x]);
    }
    data[blockIdx.x * blockDim.x + threadIdx.x] = v;
}

int main() {
    const int N = 200;
    for (int threads : {256, 512}) for (int blocks : {128, 256, 512, 1024}) {
        float* d; unsigned* c; int* p;
        CK(hipMalloc(&d, (size_t)blocks * threads * 4)); CK(hipMemset(d, 0, (size_t)blocks * threads * 4)); CK(hipMalloc(&c, 4)); CK(hipMalloc(&p, 4));
        hipStream_t st; CK(hipStreamCreate(&st));
        for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(empty_kernel, dim3(blocks), dim3(threads), 0, st, p);
        CK(hipStreamSynchronize(st));
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(empty_kernel, dim3(blocks), dim3(threads), 0, st, p);
        CK(hipStreamSynchronize(st));
        const double a = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        int n = N; void* args[] = {&n, &d};
        int maxb = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&maxb, coop_kernel, threads, 0));
        double b = -1;
        if ((long)maxb * 256 >= blocks) {
            CK(hipLaunchCooperativeKernel((void*)coop_kernel, dim3(blocks), dim3(threads), args, 0, st)); CK(hipStreamSynchronize(st));
            t0 = std::chrono::steady_clock::now();
            CK(hipLaunchCooperativeKernel((void*)coop_kernel, dim3(blocks), dim3(threads), args, 0, st)); CK(hipStreamSynchronize(st));
            b = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        }
        double cc = -1;
        if ((long)maxb * 256 >= blocks) {
            CK(hipMemsetAsync(c, 0, 4, st));
            void* args2[] = {&n, &d, &c};
            CK(hipLaunchCooperativeKernel((void*)manual_kernel, dim3(blocks), dim3(threads), args2, 0, st)); CK(hipStreamSynchronize(st));
            CK(hipMemsetAsync(c, 0, 4, st)); CK(hipStreamSynchronize(st));
            t0 = std::chrono::steady_clock::now();
            CK(hipLaunchCooperativeKernel((void*)manual_kernel, dim3(blocks), dim3(threads), args2, 0, st)); CK(hipStreamSynchronize(st));
            cc = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        }
        printf("threads %d blocks %4d: launch boundary %.2f us   cg grid.sync %.2f us   manual barrier %.2f us   (max blocks/CU %d)\n", threads, blocks, a, b, cc, maxb);
        hipFree(d); hipFree(c); hipFree(p); hipStreamDestroy(st);
    }
    return 0;
}
