// calib.hip -- box calibration: what THIS MI355X sustains on three bare loops, so that a bench line can tell a slow box from a slow build.
//   mfma_f32 : dependent v_mfma_f32_32x32x2_f32 chains, four waves per SIMD, random operands in registers (the fp32 conv kernels' steady state)
//   mfma_f16 : the same on v_mfma_f32_32x32x16_f16 (the fp16 family; power-bound on random data: MI355X_MICROARCH.md "DVFS give-back")
//   hbm_copy : float4 copy of a buffer well past the 256 MiB Infinity Cache (read + written bytes / time)
// No memory traffic in the MFMA loops, no LDS: wall time per MFMA = pipe cycles / clock, so the figures move with the clock the box holds.
// Nothing here is on the product path; bench.py prints the three numbers next to its throughput lines.
#include "../../include/isegmi.h"
#include "common.h"
using namespace isegmi;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

__device__ __forceinline__ float lcg_unit(uint32_t& s) {  // uniform in [-1, 1)
    s = s * 1664525u + 1013904223u;
    return (float)(int32_t)s * (1.0f / 2147483648.0f);
}

template <bool F16>
__global__ __launch_bounds__(256) void calib_mfma_kernel(float* __restrict__ out, int n) {
    uint32_t s = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    f32x16 acc = {};
    if (F16) {
        f16x8 a[8], b[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) { a[j][e] = (_Float16)lcg_unit(s); b[j][e] = (_Float16)lcg_unit(s); }
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[j], b[j], acc, 0, 0, 0);
        }
    } else {
        float a[8], b[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { a[j] = lcg_unit(s); b[j] = lcg_unit(s); }
        for (int i = 0; i < n; ++i) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[15];
}

// four independent 16-byte loads in flight per lane before the first store; 4.7-4.9 TB/s of read + written bytes on the boxes measured so far (a
// figure to compare boxes with, not the chip's ceiling: MI355X_MICROARCH.md quotes 6.3 TB/s for a tuned copy)
__global__ __launch_bounds__(256) void calib_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int64_t n4) {
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const float4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
        dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
    }
    for (; i < n4; i += stride) dst[i] = src[i];
}

}  // namespace

// Each leg: one short untimed launch, then one timed launch of about ms_per_leg milliseconds (sized from the nominal rate) between two events.
extern "C" int isegmi_box_calibrate(double ms_per_leg, double* mfma_f32_tflops, double* mfma_f16_tflops, double* hbm_copy_gbs) {
    ARG_CHECK(mfma_f32_tflops && mfma_f16_tflops && hbm_copy_gbs, "null output");
    ARG_CHECK(ms_per_leg > 0.0 && ms_per_leg <= 1000.0, "ms_per_leg in (0, 1000]");
    const int blocks = 1024;  // 256 CUs x 4 blocks of 4 waves: four waves per SIMD
    float* out = nullptr;
    HIP_TRY(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    float ms = 0.f;
    int rc = ISEGMI_OK;
#define CAL_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { set_error(std::string(#expr " -> ") + hipGetErrorString(e_)); rc = ISEGMI_ERR_HIP; goto done; } } while (0)
    {
        // 32x32x2 f32: 4096 FLOP per MFMA, 64 cycles per SIMD; at 2.4 GHz a wave among four gets one MFMA per 107 ns
        const int n32 = (int)(ms_per_leg * 1e6 / (8.0 * 4.0 * 64.0 / 2.4)) + 1;
        calib_mfma_kernel<false><<<blocks, 256>>>(out, 64);
        CAL_TRY(hipEventRecord(e0));
        calib_mfma_kernel<false><<<blocks, 256>>>(out, n32);
        CAL_TRY(hipEventRecord(e1));
        CAL_TRY(hipEventSynchronize(e1));
        CAL_TRY(hipEventElapsedTime(&ms, e0, e1));
        *mfma_f32_tflops = (double)blocks * 4.0 * n32 * 8.0 * 4096.0 / (ms * 1e-3) / 1e12;
        // 32x32x16 f16: 32768 FLOP per MFMA, 32 cycles per SIMD
        const int n16 = (int)(ms_per_leg * 1e6 / (8.0 * 4.0 * 32.0 / 2.0)) + 1;
        calib_mfma_kernel<true><<<blocks, 256>>>(out, 64);
        CAL_TRY(hipEventRecord(e0));
        calib_mfma_kernel<true><<<blocks, 256>>>(out, n16);
        CAL_TRY(hipEventRecord(e1));
        CAL_TRY(hipEventSynchronize(e1));
        CAL_TRY(hipEventElapsedTime(&ms, e0, e1));
        *mfma_f16_tflops = (double)blocks * 4.0 * n16 * 8.0 * 32768.0 / (ms * 1e-3) / 1e12;
    }
    {
        const int64_t bytes = (int64_t)1 << 30;  // 1 GiB each way: 8x the Infinity Cache
        float4 *src = nullptr, *dst = nullptr;
        CAL_TRY(hipMalloc(&src, (size_t)bytes));
        if (hipMalloc(&dst, (size_t)bytes) != hipSuccess) { (void)hipFree(src); set_error("box_calibrate: hipMalloc of the copy target failed"); rc = ISEGMI_ERR_HIP; goto done; }
        hipError_t e = hipMemsetAsync(src, 1, (size_t)bytes, 0);
        int reps = (int)(ms_per_leg / 0.35) + 1;   // one 2-GiB pass takes ~0.35 ms at 6 TB/s
        if (e == hipSuccess) {
            calib_copy_kernel<<<2048, 256>>>(src, dst, bytes / 16);
            e = hipEventRecord(e0);
            for (int r = 0; r < reps; ++r) calib_copy_kernel<<<2048, 256>>>(src, dst, bytes / 16);
            if (e == hipSuccess) e = hipEventRecord(e1);
            if (e == hipSuccess) e = hipEventSynchronize(e1);
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        }
        (void)hipFree(src);
        (void)hipFree(dst);
        if (e != hipSuccess) { set_error(std::string("box_calibrate copy leg -> ") + hipGetErrorString(e)); rc = ISEGMI_ERR_HIP; goto done; }
        *hbm_copy_gbs = 2.0 * (double)bytes * reps / (ms * 1e-3) / 1e9;
    }
done:
#undef CAL_TRY
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(out);
    return rc;
}
