// cabi_common.hip -- error reporting and version for the C ABI (include/prcnn_pointops.h).
#include "common.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

int prcnn_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

PRCNN_API const char* prcnn_last_error(void) { return g_err; }
PRCNN_API int prcnn_abi_version(void) { return 9; }

#ifndef PRCNN_BUILD_ID
#define PRCNN_BUILD_ID "PRCNN_BUILD_ID=unknown"
#endif
// the macro carries the "PRCNN_BUILD_ID=" tag so that the digest can also be read from the file without loading it
PRCNN_API const char* prcnn_build_id(void) { return &PRCNN_BUILD_ID[sizeof("PRCNN_BUILD_ID=") - 1]; }
