// api.cpp -- version and thread-local error message of libmridc_amd.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "mrx_common.h"

static thread_local char g_err[512] = "";

void mrx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int mrx_version(void) { return 256; /* 0.2.0: bumped with every kernel change (profiles/rNN_traffic.json is keyed on it) */ }
extern "C" const char* mrx_last_error(void) { return g_err; }
extern "C" int mrx_arith(void) {
    const char* e = getenv("MRIDC_AMD_ARITH");
    if (!e) return MRX_ARITH_F16X2;
    if (!strcmp(e, "bf16x3")) return MRX_ARITH_BF16X3;
    if (!strcmp(e, "fp32")) return MRX_ARITH_FP32;
    return MRX_ARITH_F16X2;
}

// 0 when `stream` is not being captured into a hipGraph, otherwise the (non-zero) id of the capture: callers that cache prepared operands
// key them on it, so an operand prepared eagerly is never baked into a graph and one prepared inside a capture is never used outside it.
extern "C" int64_t mrx_stream_capture_id(void* stream) {
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    unsigned long long id = 0;
    if (hipStreamGetCaptureInfo((hipStream_t)stream, &status, &id) != hipSuccess) return 0;
    if (status != hipStreamCaptureStatusActive) return 0;
    return (int64_t)(id ? id : 1ull);
}
