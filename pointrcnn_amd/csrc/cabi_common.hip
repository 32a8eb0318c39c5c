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
PRCNN_API int prcnn_abi_version(void) { return 3; }
