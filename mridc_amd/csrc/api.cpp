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

extern "C" int mrx_version(void) { return 266; /* 0.2.0: bumped with every kernel change (profiles/rNN_traffic.json is keyed on it) */ }
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

#ifdef MRX_CHECK_BOUNDS
extern "C" int mrx_checks_enabled(void) { return 1; }
__global__ void k_check_max_abs(const float* __restrict__ x, long long n, unsigned* __restrict__ out) {
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = fabsf(x[i]);
        m = (v > m || v != v) ? v : m;                                   // NaN sticks
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(m, off, 64);
        m = (o > m || o != o) ? o : m;
    }
    if ((threadIdx.x & 63) == 0) atomicMax(out, m != m ? 0x7fc00000u : __float_as_uint(m));      // non-negative floats order like their bit patterns
}
int mrx_check_bound(const char* who, const float* x, long long n, const float* bound, hipStream_t stream) {
    if (!bound || !x || n <= 0) return MRX_OK;
    hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &status) == hipSuccess && status != hipStreamCaptureStatusNone) return MRX_OK;    // cannot synchronise a capture
    static thread_local unsigned* d_max = nullptr;
    if (!d_max) MRX_HIP(hipMalloc(&d_max, sizeof(unsigned)));
    MRX_HIP(hipMemsetAsync(d_max, 0, sizeof(unsigned), stream));
    long long nb = (n + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(k_check_max_abs, dim3((unsigned)nb), dim3(256), 0, stream, x, n, d_max);
    MRX_LAUNCH_CHECK();
    float h_max = 0.f, h_bound = 0.f;
    MRX_HIP(hipMemcpyAsync(&h_max, d_max, sizeof(float), hipMemcpyDeviceToHost, stream));
    MRX_HIP(hipMemcpyAsync(&h_bound, bound, sizeof(float), hipMemcpyDeviceToHost, stream));
    MRX_HIP(hipStreamSynchronize(stream));
    if (h_max != h_max) return MRX_OK;                                   // NaN input: the kernel's result is NaN with any bound
    if (!(h_max <= h_bound * 1.000001f)) {
        mrx_set_error("%s: operand bound %.9g does NOT bound its tensor (max |x| = %.9g over %lld values): stale or foreign bound", who, (double)h_bound,
                      (double)h_max, n);
        return MRX_EBOUND;
    }
    if (h_max > 0.f && h_bound > 65536.f * h_max) {
        mrx_set_error("%s: operand bound %.9g is more than 2^16 x max |x| = %.9g (%lld values): the two-term fp16 operands lose fp32 accuracy", who,
                      (double)h_bound, (double)h_max, n);
        return MRX_EBOUND;
    }
    return MRX_OK;
}
#else
extern "C" int mrx_checks_enabled(void) { return 0; }
#endif
