// Error reporting shared by the C ABI entry points.
#include "common.h"
#include <cstring>

namespace ysmr {

char *error_buffer()
{
    static thread_local char buf[512] = "";
    return buf;
}

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace ysmr

extern "C" {

int ysmr_abi_version(void) { return YSMR_ABI_VERSION; }

const char *ysmr_last_error(void) { return ysmr::error_buffer(); }

}
