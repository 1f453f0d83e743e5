// Shared helpers for libysmr_hip.so (host side).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include "ysmr_hip.h"

namespace ysmr {

// thread-local message returned by ysmr_last_error()
char *error_buffer();
int fail(int code, const char *fmt, ...);

#define YSMR_HIP_CHECK(expr)                                                                   \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return ::ysmr::fail(YSMR_ERR_HIP, "%s failed: %s (%s:%d)", #expr,                  \
                                hipGetErrorString(e_), __FILE__, __LINE__);                    \
    } while (0)

#define YSMR_LAUNCH_CHECK() YSMR_HIP_CHECK(hipGetLastError())

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace ysmr
