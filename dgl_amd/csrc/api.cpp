// Error plumbing and version of the C ABI (host only).
#include <cstdarg>
#include <cstdio>

#include "../../include/dgl_amd.h"

static thread_local char g_err[512] = "";

int dgl_set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

extern "C" int dgl_version(void) { return DGL_AMD_VERSION; }
extern "C" const char *dgl_last_error_string(void) { return g_err; }
