// pk_lib.hip - library plumbing: version, thread-local error string, device info.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include "pk_common.h"

static thread_local char g_err[1024] = "";

void pk_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

const char* pk_experiment(const char* key) {
    // (returns a pointer INTO the environment string: the value ends at the next ',' or at the end - atoi / sscanf / a
    // first-character test, which is all the callers do, stop there by themselves; blanks around an item, its key and its
    // '=' are skipped, as pytorch-kaldi_amd/_lib.py::experiment strips them)
    const char* e = getenv("PK_EXPERIMENT");
    if (e == nullptr) return nullptr;
    const size_t kl = strlen(key);
    for (const char* p = e; *p;) {
        const char* end = strchr(p, ',');
        if (end == nullptr) end = p + strlen(p);
        const char* q = p;
        while (q < end && (*q == ' ' || *q == '\t')) ++q;
        if ((size_t)(end - q) > kl && strncmp(q, key, kl) == 0) {
            const char* r = q + kl;
            while (r < end && (*r == ' ' || *r == '\t')) ++r;
            if (r < end && *r == '=') {
                ++r;
                while (r < end && (*r == ' ' || *r == '\t')) ++r;
                return r;
            }
        }
        p = *end ? end + 1 : end;
    }
    return nullptr;
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
