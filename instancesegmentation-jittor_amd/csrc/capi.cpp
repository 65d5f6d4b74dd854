// capi.cpp -- device plumbing + error reporting of the C ABI (include/isegmi.h).
#include "../../include/isegmi.h"
#include "common.h"

namespace isegmi {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
const char* get_error() { return g_err.c_str(); }
}  // namespace isegmi
using namespace isegmi;

extern "C" int isegmi_version(void) { return ISEGMI_ABI_VERSION; }
extern "C" const char* isegmi_last_error(void) { return get_error(); }
extern "C" int isegmi_device_count(int* n) {
    ARG_CHECK(n, "null");
    hipError_t e = hipGetDeviceCount(n);
    if (e != hipSuccess) { *n = 0; set_error(std::string("hipGetDeviceCount: ") + hipGetErrorString(e)); return ISEGMI_ERR_HIP; }
    return ISEGMI_OK;
}
extern "C" int isegmi_set_device(int id) { HIP_TRY(hipSetDevice(id)); return ISEGMI_OK; }
extern "C" int isegmi_malloc(void** p, int64_t bytes) {
    ARG_CHECK(p && bytes >= 0, "malloc args");
    HIP_TRY(hipMalloc(p, (size_t)(bytes > 0 ? bytes : 16)));
    return ISEGMI_OK;
}
// pinned (page-locked) host memory: the source of isegmi_engine_upload_async
extern "C" int isegmi_malloc_host(void** p, int64_t bytes) {
    ARG_CHECK(p && bytes >= 0, "malloc_host args");
    HIP_TRY(hipHostMalloc(p, (size_t)(bytes > 0 ? bytes : 16), hipHostMallocDefault));
    return ISEGMI_OK;
}
extern "C" int isegmi_free_host(void* p) { HIP_TRY(hipHostFree(p)); return ISEGMI_OK; }
extern "C" int isegmi_free(void* p) { HIP_TRY(hipFree(p)); return ISEGMI_OK; }
extern "C" int isegmi_h2d(void* d, const void* h, int64_t bytes) { HIP_TRY(hipMemcpy(d, h, (size_t)bytes, hipMemcpyHostToDevice)); return ISEGMI_OK; }
extern "C" int isegmi_d2h(void* h, const void* d, int64_t bytes) { HIP_TRY(hipMemcpy(h, d, (size_t)bytes, hipMemcpyDeviceToHost)); return ISEGMI_OK; }
extern "C" int isegmi_memset(void* d, int v, int64_t bytes) { HIP_TRY(hipMemset(d, v, (size_t)bytes)); return ISEGMI_OK; }
extern "C" int isegmi_sync(void) { HIP_TRY(hipDeviceSynchronize()); return ISEGMI_OK; }
