// api.cpp -- version and thread-local error message of libmridc_amd.
#include <cstdarg>
#include <cstdio>

#include "mrx_common.h"

static thread_local char g_err[512] = "";

void mrx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int mrx_version(void) { return 222; /* 0.2.0: bumped with every kernel change (profiles/rNN_traffic.json is keyed on it) */ }
extern "C" const char* mrx_last_error(void) { return g_err; }
