// common.h -- shared host/device helpers for libisegmi (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
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

}  // namespace isegmi
