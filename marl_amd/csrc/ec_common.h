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

// ec_mlp.hip: the Actor's weights in the fragment order of ec_policy_inline.h (one small launch), for the looping
// env kernel of ec_rollout_mlp_policy.  dst holds polinl::staged_floats(n_agents, n_actions <= 32 ? 1 : 2) floats.
int stage_policy(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3,
                 int n_agents, int n_actions, float* dst, hipStream_t stream);

}  // namespace ec
