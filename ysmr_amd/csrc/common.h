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

// cv2.cvtColor(BGR2GRAY) on u8: fixed point, (B*b + G*g + R*r + half) >> shift.  OpenCV 4.x uses 15 bits
// (3735 / 19235 / 9798), OpenCV 3.x 14 bits (1868 / 9617 / 4899); both are the identity for B = G = R.
struct GrayCoef { uint32_t b, g, r, half, shift; };
inline GrayCoef gray_coef(int cv_flavour)
{
    if (cv_flavour & YSMR_CV_GRAY_3X) return GrayCoef{1868u, 9617u, 4899u, 8192u, 14u};
    return GrayCoef{3735u, 19235u, 9798u, 16384u, 15u};
}

}  // namespace ysmr

__device__ __forceinline__ uint32_t bgr2gray(const ysmr::GrayCoef &k, uint32_t b, uint32_t g, uint32_t r)
{
    return (b * k.b + g * k.g + r * k.r + k.half) >> k.shift;   // cv2 COLOR_BGR2GRAY (a1)
}
