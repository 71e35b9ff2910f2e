// pk_lib.hip - library plumbing: version, thread-local error string, device info.
#include <stdarg.h>
#include <string.h>

#include "pk_common.h"

static thread_local char g_err[1024] = "";

void pk_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int pk_version(void) { return 100; }

extern "C" const char* pk_last_error(void) { return g_err; }

extern "C" int pk_num_cu(void) {
    static int cached = 0;
    if (cached > 0) return cached;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
    cached = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    return cached;
}
