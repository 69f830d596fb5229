// mrx_common.h -- shared host-side helpers of libmridc_amd (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>

#include "../../include/mridc_amd.h"

void mrx_set_error(const char* fmt, ...);

// The ONE arithmetic switch of the library: environment MRIDC_AMD_ARITH = "f16x2" (default: two-term fp16 operands where a kernel has them),
// "bf16x3" (three-term bf16 operands, six exact term products per fp32 multiply) or "fp32" (the fp32-input MFMA kernels).  Every form has
// fp32 results; the switch exists to cross-check the forms against each other (tests) and to price them (bench.py exact_fp32_route).
// Read at every call (tests flip it in-process).
#define MRX_ARITH_F16X2 0
#define MRX_ARITH_BF16X3 1
#define MRX_ARITH_FP32 2
extern "C" int mrx_arith(void);

// Instrumentation switches (cycle stamps, phase ablations) exist in PROBE builds only (MRX_BUILD_DEFS=-DMRX_PROBE python -m mridc_amd._build):
// the product library reads no debug environment.
#ifdef MRX_PROBE
#define MRX_DEBUG_ENV(name) getenv(name)
#else
#define MRX_DEBUG_ENV(name) ((const char*)nullptr)
#endif

#define MRX_REQUIRE(cond, code, ...)  \
    do {                              \
        if (!(cond)) {                \
            mrx_set_error(__VA_ARGS__); \
            return (code);            \
        }                             \
    } while (0)

#define MRX_HIP(call)                                                                     \
    do {                                                                                  \
        hipError_t e__ = (call);                                                          \
        if (e__ != hipSuccess) {                                                          \
            mrx_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, __LINE__); \
            return MRX_EHIP;                                                              \
        }                                                                                 \
    } while (0)

#define MRX_LAUNCH_CHECK()                                                                 \
    do {                                                                                   \
        hipError_t e__ = hipGetLastError();                                                \
        if (e__ != hipSuccess) {                                                           \
            mrx_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e__), __FILE__, __LINE__); \
            return MRX_EHIP;                                                               \
        }                                                                                  \
    } while (0)

// re^2 + im^2 rounded as torch's (data ** 2).sum(-1) rounds it -- two rounded squares, one rounded add -- in EVERY translation unit, whatever its
// contraction mode.  HIP's __fmul_rn / __fadd_rn are plain operators: hipcc contracts `__fadd_rn(__fmul_rn(x, x), __fmul_rn(y, y))` into
// v_mul + v_fmac unless the file is built with -ffp-contract=off (round 4: conv_bwd.hip did so once its packed-fp32 v_pk_mul was gone, the
// backward of the l1 loss no longer found the arg-max pixel by `|p| == max |p|` -- the maximum comes from elementwise.hip -- and dropped that
// gradient path: 85 % error on some weights).  The pragma clears the `contract` flag of these three operations themselves; it survives inlining.
#if defined(__HIPCC__)
__device__ __forceinline__ float mrx_sumsq2(float re, float im) {
#pragma clang fp contract(off)
    const float a = re * re;
    const float b = im * im;
    return a + b;
}
// a * s - t, product rounded before the subtraction (the l1 loss term |p| / max - target: forward and backward must see the same sign)
__device__ __forceinline__ float mrx_mul_sub(float a, float s, float t) {
#pragma clang fp contract(off)
    const float m = a * s;
    return m - t;
}
#endif

// output tile of the convolution kernels (also the granule of the fused InstanceNorm tile statistics)
#define MRX_CONV_TILE_H 8
#define MRX_CONV_TILE_W 32

static inline int mrx_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// mask value as the multiplicative factor torch's type promotion gives (bool/uint8 -> float)
struct MrxMask {
    const void* p;
    int kind;  // MRX_MASK_U8 / MRX_MASK_F32
    long long s[4];
};
// XCD-aware work order: workgroup b of a launch runs on XCD b % 8 (8 XCDs, one L2 each).  Returns the work item of
// workgroup b such that every XCD walks one contiguous band of the n items, for any n (speed only; any order is correct):
// neighbouring items, which share cache lines / halo rows, then meet in the same L2 at about the same time.
__device__ __forceinline__ long long mrx_xcd_band(long long b, long long n) {
    const long long x = b & 7, i = b >> 3, q = n >> 3, r = n & 7;
    return x * q + (x < r ? x : r) + i;
}

__device__ __forceinline__ float mrx_mask_val(const MrxMask& m, long long b, long long c, long long h, long long w) {
    const long long off = b * m.s[0] + c * m.s[1] + h * m.s[2] + w * m.s[3];
    return m.kind == MRX_MASK_U8 ? (float)((const unsigned char*)m.p)[off] : ((const float*)m.p)[off];
}
__device__ __forceinline__ bool mrx_mask_true(const MrxMask& m, long long b, long long c, long long h, long long w) {
    const long long off = b * m.s[0] + c * m.s[1] + h * m.s[2] + w * m.s[3];
    return m.kind == MRX_MASK_U8 ? ((const unsigned char*)m.p)[off] != 0 : ((const float*)m.p)[off] != 0.0f;
}
