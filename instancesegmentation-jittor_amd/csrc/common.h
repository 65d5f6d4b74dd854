// common.h -- shared host/device helpers for libisegmi (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <atomic>
#include <string>

namespace isegmi {

// thread-local last-error string, surfaced through isegmi_last_error()
void set_error(const std::string& msg);
const char* get_error();

#define ISEGMI_OK 0
#define ISEGMI_ERR_HIP -1
#define ISEGMI_ERR_ARG -2
#define ISEGMI_ERR_STATE -3
#define ISEGMI_ERR_RCCL -4

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            char _b[512];                                                                     \
            snprintf(_b, sizeof(_b), "%s:%d: %s -> %s", __FILE__, __LINE__, #expr,            \
                     hipGetErrorString(_e));                                                  \
            ::isegmi::set_error(_b);                                                          \
            return ISEGMI_ERR_HIP;                                                            \
        }                                                                                     \
    } while (0)

#define ARG_CHECK(cond, msg)                                                                  \
    do {                                                                                      \
        if (!(cond)) {                                                                        \
            char _b[512];                                                                     \
            snprintf(_b, sizeof(_b), "%s:%d: bad argument: %s (%s)", __FILE__, __LINE__, msg, \
                     #cond);                                                                  \
            ::isegmi::set_error(_b);                                                          \
            return ISEGMI_ERR_ARG;                                                            \
        }                                                                                     \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// hipFuncSetAttribute acts on the CURRENT device: a call site remembers per device whether it has raised its kernel's dynamic-LDS limit
// (a process-wide flag left a second device of the same process at the 64 KB default: ADVICE r4).
// The flag of a device is set only AFTER the attribute call has succeeded (a failed call is retried by the next launch instead of leaving every later
// launch on that device at the default limit), and it is atomic: engines on different host threads share the call site's static instance.
struct PerDeviceOnce {
    std::atomic<bool> done[32];
    PerDeviceOnce() { for (auto& d : done) d.store(false, std::memory_order_relaxed); }
    // -> the device whose flag commit() sets once the call went through; -1: device unknown (set the attribute, remember nothing); -2: already set
    int pending() const {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return -1;
        return done[dev].load(std::memory_order_acquire) ? -2 : dev;
    }
    void commit(int dev) { if (dev >= 0) done[dev].store(true, std::memory_order_release); }
};
// raise a kernel's dynamic-LDS limit once per device; the kernel goes last because template arguments carry commas
#define LDS_LIMIT_ONCE(bytes, ...)                                                                                              \
    do {                                                                                                                        \
        static ::isegmi::PerDeviceOnce _once;                                                                                   \
        const int _dev = _once.pending();                                                                                       \
        if (_dev != -2) {                                                                                                       \
            HIP_TRY(hipFuncSetAttribute((const void*)(__VA_ARGS__), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes))); \
            _once.commit(_dev);                                                                                                 \
        }                                                                                                                       \
    } while (0)
int device_cu_count();   // compute units of the current device (cached per device; csrc/conv_mfma.hip)

// Timing-only experiment switches (they drop loads / stores / whole phases: the results are WRONG) exist only in a build made with
// -DISEGMI_EXPERIMENT_FLAGS (tools/build_variant.sh exp -DISEGMI_EXPERIMENT_FLAGS); a release build rejects their bits at the C ABI.
#ifdef ISEGMI_EXPERIMENT_FLAGS
constexpr bool kExperimentFlags = true;
#else
constexpr bool kExperimentFlags = false;
#endif

}  // namespace isegmi
