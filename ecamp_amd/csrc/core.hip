// Error reporting + ABI version for libecamp_hip.so
#include "common.h"
#include <stdarg.h>

thread_local char g_ecamp_err[512] = {0};

int ecamp_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_ecamp_err, sizeof(g_ecamp_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* ecamp_last_error(void) { return g_ecamp_err; }
extern "C" int ecamp_abi_version(void) { return 1; }
