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

// CHECK builds (-DMRX_CHECK_BOUNDS): the operand bound a two-term fp16 entry point was handed is compared with the tensor it is about to read
// (api.cpp: one reduction launch, a stream synchronisation, two 4-byte copies); the product build compiles the macro away.
#ifdef MRX_CHECK_BOUNDS
int mrx_check_bound(const char* who, const float* x, long long n, const float* bound, hipStream_t stream);
#define MRX_CHECK_BOUND(who, x, n, bound, stream)                                         \
    do {                                                                                  \
        const int rc__ = mrx_check_bound((who), (x), (long long)(n), (bound), (hipStream_t)(stream)); \
        if (rc__ != MRX_OK) return rc__;                                                  \
    } while (0)
#else
#define MRX_CHECK_BOUND(who, x, n, bound, stream) do { } while (0)
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
// Second stage of every weight gradient: out[i] (= or +=) sum over `nparts` workgroup partials [nparts][n] in a FIXED order, in double
// (bit-reproducible).  Launch with (n + 63) / 64 workgroups of 256 threads.  Thread = four consecutive outputs of one of 16 slot groups: a slot is
// read in 256-byte pieces (the first form read 64-byte pieces, one dword per lane, and ran at 0.4 TB/s -- as long as the gradient kernel in front of
// it); group g sums slots g, g + 16, ... and the groups are added in order 0 .. 15, the same order of additions as before.
__device__ __forceinline__ void mrx_reduce_parts(const float* __restrict__ part, int nparts, long long n, float* __restrict__ dw, int accumulate) {
    __shared__ double sh[16][16][4 + 1];
    const int li = threadIdx.x & 15, lp = threadIdx.x >> 4;
    const long long i = ((long long)blockIdx.x * 16 + li) * 4;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if ((n & 3) == 0) {
        if (i < n) {
            const float* q = part + i;
            int p = lp;
            for (; p + 48 < nparts; p += 64) {          // four slots in flight
                const float4 u0 = *reinterpret_cast<const float4*>(q + (long long)p * n), u1 = *reinterpret_cast<const float4*>(q + (long long)(p + 16) * n);
                const float4 u2 = *reinterpret_cast<const float4*>(q + (long long)(p + 32) * n), u3 = *reinterpret_cast<const float4*>(q + (long long)(p + 48) * n);
                s0 += (double)u0.x, s1 += (double)u0.y, s2 += (double)u0.z, s3 += (double)u0.w;
                s0 += (double)u1.x, s1 += (double)u1.y, s2 += (double)u1.z, s3 += (double)u1.w;
                s0 += (double)u2.x, s1 += (double)u2.y, s2 += (double)u2.z, s3 += (double)u2.w;
                s0 += (double)u3.x, s1 += (double)u3.y, s2 += (double)u3.z, s3 += (double)u3.w;
            }
            for (; p < nparts; p += 16) {
                const float4 u = *reinterpret_cast<const float4*>(q + (long long)p * n);
                s0 += (double)u.x, s1 += (double)u.y, s2 += (double)u.z, s3 += (double)u.w;
            }
        }
    } else {                                            // (no shape of this library: n = Cout * Cin * k * k with Cout or Cin a multiple of 4)
        for (int p = lp; p < nparts; p += 16) {
            if (i < n) s0 += (double)part[(long long)p * n + i];
            if (i + 1 < n) s1 += (double)part[(long long)p * n + i + 1];
            if (i + 2 < n) s2 += (double)part[(long long)p * n + i + 2];
            if (i + 3 < n) s3 += (double)part[(long long)p * n + i + 3];
        }
    }
    sh[lp][li][0] = s0, sh[lp][li][1] = s1, sh[lp][li][2] = s2, sh[lp][li][3] = s3;
    __syncthreads();
    if (threadIdx.x < 64) {
        const int c = threadIdx.x >> 2, j = threadIdx.x & 3;
        const long long o = ((long long)blockIdx.x * 16 + c) * 4 + j;
        if (o < n) {
            double t = 0.0;
#pragma unroll
            for (int k = 0; k < 16; ++k) t += sh[k][c][j];
            dw[o] = accumulate ? dw[o] + (float)t : (float)t;
        }
    }
}

#endif

// output tile of the convolution kernels (also the granule of the fused InstanceNorm tile statistics)
#define MRX_CONV_TILE_H 8
#define MRX_CONV_TILE_W 32

static inline int mrx_cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }
// The ONE split rule of the per-plane statistics work buffers (unet.hip writes [2][planes][nsplit] partial sums / squared deviations, diff_bwd.hip's
// instance-norm backward reads rstd back out of the forward's buffer by the same index): both translation units call this, neither has its own copy.
#define MRX_NORM_CHUNK 8192
static inline int mrx_norm_nsplit(long long n) {
    long long s = (n + MRX_NORM_CHUNK - 1) / MRX_NORM_CHUNK;
    return s < 1 ? 1 : (s > 64 ? 64 : (int)s);
}

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
