// fill_path.hip -- how fast can ONE CU pull L2-resident bytes (dev microbenchmark, MI355X)?
//   mode 0: buffer_load_dwordx4 ... lds (LDS-DMA), 1 KiB per wave-instruction
//   mode 1: buffer_load_dwordx4 into VGPRs (16 B per lane), results xor-reduced so that they stay live
//   mode 2: both, alternating 1:1
// One 512-thread block per CU; every block re-reads its own 64 KiB window (L2-resident after the first pass).
// build: hipcc --offload-arch=gfx950 -O3 -o fill_path fill_path.hip ; run: ./fill_path
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int MODE>
__global__ __launch_bounds__(512, 1) void fill_kernel(const char* src, unsigned bytes_per_block, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)blockIdx.x * bytes_per_block), 0, bytes_per_block, 0x00020000);
    const unsigned pieces = bytes_per_block / 1024;  // 1 KiB pieces in the window
    u32x4 acc = {0, 0, 0, 0};
    unsigned pc = wave;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned off = (pc % pieces) * 1024u + lane * 16u;
            pc += nw;
            if (MODE == 0 || (MODE == 2 && (j & 1) == 0)) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + (wave * 8 + j) * 1024), 16, off, 0, 0, 0);
            } else {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
                acc ^= v;
            }
        }
        if (MODE != 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[0] = 1;
}

template <int MODE>
static void run(const char* name, const char* d, unsigned* sink, int threads) {
    const unsigned bpb = 64 * 1024; const int iters = 2000, blocks = 256;
    const size_t lds = 64 * 1024;
    hipFuncSetAttribute((const void*)fill_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    fill_kernel<MODE><<<blocks, threads, lds>>>(d, bpb, 50, sink);
    hipEventRecord(a);
    fill_kernel<MODE><<<blocks, threads, lds>>>(d, bpb, iters, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)blocks * (threads / 64) * 8.0 * 1024.0 * iters;
    printf("%-28s %d waves/CU: %7.1f GB/s per CU  (%6.2f TB/s chip, %.3f ms)\n", name, threads / 64, bytes / ms / 1e6 / blocks, bytes / ms / 1e9, ms);
}

int main() {
    char* d; unsigned* sink;
    hipMalloc(&d, 256u * 64 * 1024); hipMemset(d, 1, 256u * 64 * 1024); hipMalloc(&sink, 4);
    for (int threads : {256, 512}) {
        run<0>("LDS-DMA (b128 ... lds)", d, sink, threads);
        run<1>("VGPR loads (b128)", d, sink, threads);
        run<2>("mixed 1:1", d, sink, threads);
    }
    return 0;
}
