// Library-level entry points: version and last-error string.
#include "common.h"
#include <stdarg.h>
#include <string.h>

static thread_local char g_err[512] = "";

void rs_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {

int rs_version(void) { return 100; }   // 0.1.0

const char* rs_last_error_string(void) { return g_err; }

// Number of compute units of the current device (for host-side grid sizing decisions).
int rs_device_cu_count(void) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return -1;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
    return n;
}

}  // extern "C"
