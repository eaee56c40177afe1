// Host-side helpers shared by the C-ABI entry points: argument checks, the thread-local
// error string behind esme_hip_last_error(), and the post-launch error check.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/esme_hip.h"

namespace esme {

char* error_buffer();                       // thread-local, defined in api.hip
static constexpr int kErrorBufferSize = 512;

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int fail(int code, const char* msg) {
    snprintf(error_buffer(), kErrorBufferSize, "%s", msg);
    return code;
}

inline int check_launch(const char* what) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return ESME_OK;
    snprintf(error_buffer(), kErrorBufferSize, "%s: launch failed: %s", what, hipGetErrorString(e));
    return ESME_ERR_LAUNCH;
}

}  // namespace esme

#define ESME_FAIL(code, msg) return ::esme::fail((code), (msg))
#define ESME_CHECK_ARG(cond, msg) \
    do { if (!(cond)) return ::esme::fail(ESME_ERR_ARG, (msg)); } while (0)
