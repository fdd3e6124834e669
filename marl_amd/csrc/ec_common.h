// Shared host-side helpers for libmarl_amd.so (error reporting, HIP call checking).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/marl_amd.h"

namespace ec {

inline char* err_buf() {
    static thread_local char buf[512] = "";
    return buf;
}

inline int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#define EC_HIP(call)                                                                                  \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return ec::fail(EC_EHIP, "%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
    } while (0)

}  // namespace ec
