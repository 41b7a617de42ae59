// gf_abi.hip — library identity + thread-local error channel of the C ABI.
#include "gf_common.h"
#include <cstdarg>
#include <cstdio>

#define GF_ABI_VERSION 8

static thread_local char g_err[512] = "";

void gf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" GF_API const char* gf_version(void) { return "goalforce-hip 0.1.0 gfx950"; }
extern "C" GF_API const char* gf_last_error(void) { return g_err; }
extern "C" GF_API int gf_abi_version(void) { return GF_ABI_VERSION; }
