// elementwise.hip -- HBM-bound pointwise / small-reduction operators of the hot path (gfx950):
// roll/fftshift, complex arithmetic, coil combination, mask application, soft data consistency, GRU/MGU gates.
// All are grid-stride kernels with contiguous per-lane accesses; reductions over the coil dim keep `inner`
// as the fastest index so a wave reads 64 consecutive elements per coil.
#include "mrx_common.h"

#define EW_NT 256
// correctly rounded fp32 square root (what torch's CPU sqrt returns): the double-precision root rounded once more to
// float is exact because 53 >= 2*24 + 2 bits (double rounding is innocuous for sqrt)
__device__ __forceinline__ float sqrt_rn(float s) { return (float)sqrt((double)s); }
static inline int ew_grid(long long n) {
    long long g = (n + EW_NT - 1) / EW_NT;
    if (g > 256 * 16) g = 256 * 16;  // 256 CUs x 16 blocks, grid-stride for the rest
    return g < 1 ? 1 : (int)g;
}

// ---- roll (fft.py:169-240) ---------------------------------------------------------------------------------
struct RollArgs {
    int ndim;
    long long shape[8];
    long long shift[8];  // normalised to [0, n)
    long long total;
};
template <typename T>
__global__ void k_roll(const T* __restrict__ in, T* __restrict__ out, RollArgs a) {
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < a.total;
         o += (long long)gridDim.x * blockDim.x) {
        long long rem = o, src = 0, mul = 1;
        for (int d = a.ndim - 1; d >= 0; --d) {
            const long long n = a.shape[d];
            const long long i = rem % n;
            rem /= n;
            long long j = i - a.shift[d];  // out[i] = in[(i - shift) mod n]
            if (j < 0) j += n;
            src += j * mul;
            mul *= n;
        }
        out[o] = in[src];
    }
}

extern "C" int mrx_roll(const void* in, void* out, int elem_bytes, int ndim, const int64_t* shape,
                        const int64_t* shifts, void* stream) {
    MRX_REQUIRE(in && out && shape && shifts, MRX_EINVAL, "mrx_roll: null pointer");
    MRX_REQUIRE(ndim >= 1 && ndim <= 8, MRX_EINVAL, "mrx_roll: ndim %d outside [1,8]", ndim);
    MRX_REQUIRE(in != out, MRX_EINVAL, "mrx_roll: in-place roll is not supported");
    RollArgs a;
    a.ndim = ndim;
    a.total = 1;
    for (int d = 0; d < ndim; ++d) {
        MRX_REQUIRE(shape[d] >= 0, MRX_EINVAL, "mrx_roll: negative dim");
        a.shape[d] = shape[d];
        a.total *= shape[d];
        long long s = shape[d] > 0 ? shifts[d] % shape[d] : 0;
        if (s < 0) s += shape[d];
        a.shift[d] = s;
    }
    if (a.total == 0) return MRX_OK;
    hipStream_t st = (hipStream_t)stream;
    const int g = ew_grid(a.total);
    switch (elem_bytes) {
        case 1: hipLaunchKernelGGL(k_roll<unsigned char>, dim3(g), dim3(EW_NT), 0, st, (const unsigned char*)in, (unsigned char*)out, a); break;
        case 2: hipLaunchKernelGGL(k_roll<unsigned short>, dim3(g), dim3(EW_NT), 0, st, (const unsigned short*)in, (unsigned short*)out, a); break;
        case 4: hipLaunchKernelGGL(k_roll<unsigned int>, dim3(g), dim3(EW_NT), 0, st, (const unsigned int*)in, (unsigned int*)out, a); break;
        case 8: hipLaunchKernelGGL(k_roll<uint2>, dim3(g), dim3(EW_NT), 0, st, (const uint2*)in, (uint2*)out, a); break;
        case 16: hipLaunchKernelGGL(k_roll<uint4>, dim3(g), dim3(EW_NT), 0, st, (const uint4*)in, (uint4*)out, a); break;
        default: MRX_REQUIRE(false, MRX_EINVAL, "mrx_roll: unsupported element size %d", elem_bytes);
    }
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- complex_mul with broadcasting (utils.py:96-118) ----------------------------------------------------------
struct BcastArgs {
    int ndim;
    long long shape[6], xs[6], ys[6];
    long long total;
    int conj_y;
};
__global__ void k_complex_mul(const float2* __restrict__ x, const float2* __restrict__ y, float2* __restrict__ out,
                              BcastArgs a) {
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < a.total;
         o += (long long)gridDim.x * blockDim.x) {
        long long rem = o, xo = 0, yo = 0;
        for (int d = a.ndim - 1; d >= 0; --d) {
            const long long n = a.shape[d];
            const long long i = rem % n;
            rem /= n;
            xo += i * a.xs[d];
            yo += i * a.ys[d];
        }
        const float2 p = x[xo];
        float2 q = y[yo];
        if (a.conj_y) q.y = -q.y;
        out[o] = make_float2(p.x * q.x - p.y * q.y, p.x * q.y + p.y * q.x);
    }
}
extern "C" int mrx_complex_mul(const float* x, const float* y, float* out, int ndim, const int64_t* shape,
                               const int64_t* xstride, const int64_t* ystride, int conj_y, void* stream) {
    MRX_REQUIRE(x && y && out && shape && xstride && ystride, MRX_EINVAL, "mrx_complex_mul: null pointer");
    MRX_REQUIRE(ndim >= 1 && ndim <= 6, MRX_EINVAL, "mrx_complex_mul: ndim %d outside [1,6]", ndim);
    BcastArgs a;
    a.ndim = ndim;
    a.total = 1;
    a.conj_y = conj_y;
    for (int d = 0; d < ndim; ++d) {
        MRX_REQUIRE(shape[d] >= 0, MRX_EINVAL, "mrx_complex_mul: negative dim");
        a.shape[d] = shape[d];
        a.xs[d] = xstride[d];
        a.ys[d] = ystride[d];
        a.total *= shape[d];
    }
    if (a.total == 0) return MRX_OK;
    hipLaunchKernelGGL(k_complex_mul, dim3(ew_grid(a.total)), dim3(EW_NT), 0, (hipStream_t)stream, (const float2*)x,
                       (const float2*)y, (float2*)out, a);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- complex_conj / abs / abs_sq (utils.py:121-175) -------------------------------------------------------------
__global__ void k_conj(const float2* __restrict__ x, float2* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float2 v = x[i];
        out[i] = make_float2(v.x, -v.y);
    }
}
template <bool SQ>
__global__ void k_abs(const float2* __restrict__ x, float* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float2 v = x[i];
        const float s = mrx_sumsq2(v.x, v.y);  // (data**2).sum(-1): two rounded squares, one add
        out[i] = SQ ? s : sqrt_rn(s);
    }
}
extern "C" int mrx_complex_conj(const float* x, float* out, int64_t n, void* stream) {
    MRX_REQUIRE(x && out && n >= 0, MRX_EINVAL, "mrx_complex_conj: bad argument");
    if (n == 0) return MRX_OK;
    hipLaunchKernelGGL(k_conj, dim3(ew_grid(n)), dim3(EW_NT), 0, (hipStream_t)stream, (const float2*)x, (float2*)out, (long long)n);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
extern "C" int mrx_complex_abs(const float* x, float* out, int64_t n, int squared, void* stream) {
    MRX_REQUIRE(x && out && n >= 0, MRX_EINVAL, "mrx_complex_abs: bad argument");
    if (n == 0) return MRX_OK;
    if (squared)
        hipLaunchKernelGGL(k_abs<true>, dim3(ew_grid(n)), dim3(EW_NT), 0, (hipStream_t)stream, (const float2*)x, out, (long long)n);
    else
        hipLaunchKernelGGL(k_abs<false>, dim3(ew_grid(n)), dim3(EW_NT), 0, (hipStream_t)stream, (const float2*)x, out, (long long)n);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- per-sample normalisation on the device (reconstruction/parts/transforms.py:286-288,527-617) -------------------------
// max over a whole tensor of |x| (mode 0: every float, the reference's torch.max(torch.abs(real view))) or of the complex modulus
// (mode 1: torch.abs of a complex tensor = sqrt(fl(re^2) + fl(im^2)) in fp32); NaN propagates like torch.max.  Two launches, the
// result stays on the device (no host synchronisation inside the preprocessing chain).
__device__ __forceinline__ float nan_max(float a, float b) { return (a > b || a != a) ? a : b; }
template <int MODE>
__global__ void k_max_abs_partial(const float* __restrict__ x, float* __restrict__ part, long long n) {
    __shared__ float red[EW_NT / 64];
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float v;
        if (MODE == 0) {
            v = fabsf(x[i]);
        } else {
            const float2 c = reinterpret_cast<const float2*>(x)[i];
            v = sqrt_rn(mrx_sumsq2(c.x, c.y));
        }
        m = nan_max(v, m);
    }
    for (int off = 32; off > 0; off >>= 1) m = nan_max(__shfl_xor(m, off, 64), m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < EW_NT / 64; ++w) m = nan_max(red[w], m);
        part[blockIdx.x] = m;
    }
}
__global__ void k_max_final(const float* __restrict__ part, int np, float* __restrict__ out) {
    float m = 0.f;
    for (int i = threadIdx.x; i < np; i += 64) m = nan_max(part[i], m);
    for (int off = 32; off > 0; off >>= 1) m = nan_max(__shfl_xor(m, off, 64), m);
    if (threadIdx.x == 0) out[0] = m;
}
#define MAXABS_BLOCKS 1024
extern "C" int64_t mrx_max_abs_work_floats(void) { return MAXABS_BLOCKS; }
extern "C" int mrx_max_abs(const float* x, int64_t n, int mode, float* out, float* work, void* stream) {
    MRX_REQUIRE(x && out && work && n >= 1, MRX_EINVAL, "mrx_max_abs: bad argument");
    MRX_REQUIRE(mode == 0 || mode == 1, MRX_EINVAL, "mrx_max_abs: bad mode %d", mode);
    int nb = (int)((n + EW_NT - 1) / EW_NT);
    if (nb > MAXABS_BLOCKS) nb = MAXABS_BLOCKS;
    if (mode == 0)
        hipLaunchKernelGGL(k_max_abs_partial<0>, dim3(nb), dim3(EW_NT), 0, (hipStream_t)stream, x, work, (long long)n);
    else
        hipLaunchKernelGGL(k_max_abs_partial<1>, dim3(nb), dim3(EW_NT), 0, (hipStream_t)stream, x, work, (long long)n);
    hipLaunchKernelGGL(k_max_final, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float*)work, nb, out);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
// out = x / d[0] (mode 0, n floats) or out[i] = | x_c[i] / d[0] | (mode 1, n complex -> n floats): the divisor is read on the device
template <int MODE>
__global__ void k_div_dev(const float* __restrict__ x, const float* __restrict__ d, float* __restrict__ out, long long n) {
    const float den = d[0];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        if (MODE == 0) {
            out[i] = x[i] / den;
        } else {
            const float2 c = reinterpret_cast<const float2*>(x)[i];
            const float re = c.x / den, im = c.y / den;
            out[i] = sqrt_rn(mrx_sumsq2(re, im));
        }
    }
}
extern "C" int mrx_div_by_device_scalar(const float* x, const float* d, float* out, int64_t n, int mode, void* stream) {
    MRX_REQUIRE(x && d && out && n >= 0, MRX_EINVAL, "mrx_div_by_device_scalar: bad argument");
    MRX_REQUIRE(mode == 0 || mode == 1, MRX_EINVAL, "mrx_div_by_device_scalar: bad mode %d", mode);
    if (n == 0) return MRX_OK;
    if (mode == 0)
        hipLaunchKernelGGL(k_div_dev<0>, dim3(ew_grid(n)), dim3(EW_NT), 0, (hipStream_t)stream, x, d, out, (long long)n);
    else
        hipLaunchKernelGGL(k_div_dev<1>, dim3(ew_grid(n)), dim3(EW_NT), 0, (hipStream_t)stream, x, d, out, (long long)n);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- per-slice evaluation metrics of the test harness (models/base.py:415-436, common/metrics/reconstruction_metrics.py:11-25) ------
// target, output: the `abs / max` images [n].  out5 = { MSE = mean (t - o)^2,  NMSE = sum (t - o)^2 / sum t^2,
// maxval = max(o) - min(o) (the data range base.py:429-436 hands to SSIM and PSNR),  PSNR = 10 log10(maxval^2 / MSE),  sum t^2 }.
// Differences in fp32 (as numpy does on float32 arrays), sums in double with a fixed order: results are reproducible.
#define METRIC_BLOCKS 256
__global__ void k_metric_partial(const float* __restrict__ t, const float* __restrict__ o, double* __restrict__ part, long long n) {
    __shared__ double red[EW_NT / 64][4];
    double sd = 0.0, st = 0.0;
    float mn = INFINITY, mx = -INFINITY;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float tv = t[i], ov = o[i];
        const float d = __fsub_rn(tv, ov);
        sd += (double)__fmul_rn(d, d);
        st += (double)__fmul_rn(tv, tv);
        mn = fminf(mn, ov);
        mx = fmaxf(mx, ov);
    }
    for (int off = 32; off > 0; off >>= 1) {
        sd += __shfl_xor(sd, off, 64);
        st += __shfl_xor(st, off, 64);
        mn = fminf(mn, __shfl_xor(mn, off, 64));
        mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        double* r = red[threadIdx.x >> 6];
        r[0] = sd, r[1] = st, r[2] = (double)mn, r[3] = (double)mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < EW_NT / 64; ++w) {
            sd += red[w][0], st += red[w][1];
            mn = fminf(mn, (float)red[w][2]), mx = fmaxf(mx, (float)red[w][3]);
        }
        double* p = part + 4ll * blockIdx.x;
        p[0] = sd, p[1] = st, p[2] = (double)mn, p[3] = (double)mx;
    }
}
__global__ void k_metric_final(const double* __restrict__ part, int np, long long n, float* __restrict__ out5) {
    if (threadIdx.x != 0) return;
    double sd = 0.0, st = 0.0, mn = INFINITY, mx = -INFINITY;
    for (int i = 0; i < np; ++i) {
        sd += part[4 * i], st += part[4 * i + 1];
        mn = fmin(mn, part[4 * i + 2]), mx = fmax(mx, part[4 * i + 3]);
    }
    const double mse = sd / (double)n, range = (double)((float)mx - (float)mn);
    out5[0] = (float)mse;
    out5[1] = (float)(sd / st);
    out5[2] = (float)range;
    out5[3] = (float)(10.0 * log10(range * range / mse));
    out5[4] = (float)st;
}
extern "C" int64_t mrx_recon_metrics_work_floats(void) { return 2 * 4 * METRIC_BLOCKS; }
extern "C" int mrx_recon_metrics(const float* target, const float* output, float* out5, float* work, int64_t n, void* stream) {
    MRX_REQUIRE(target && output && out5 && work && n >= 1, MRX_EINVAL, "mrx_recon_metrics: bad argument");
    MRX_REQUIRE(((uintptr_t)work & 7) == 0, MRX_EINVAL, "mrx_recon_metrics: work must be 8-byte aligned");
    int nb = (int)((n + EW_NT - 1) / EW_NT);
    if (nb > METRIC_BLOCKS) nb = METRIC_BLOCKS;
    hipLaunchKernelGGL(k_metric_partial, dim3(nb), dim3(EW_NT), 0, (hipStream_t)stream, target, output, (double*)work, (long long)n);
    hipLaunchKernelGGL(k_metric_final, dim3(1), dim3(64), 0, (hipStream_t)stream, (const double*)work, nb, (long long)n, out5);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- coil combination over a middle dim (utils.py:194-248) ------------------------------------------------------
__global__ void k_rss(const float* __restrict__ x, float* __restrict__ out, long long outer, long long R, long long inner) {
    const long long total = outer * inner;
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long long)gridDim.x * blockDim.x) {
        const long long ob = o / inner, i = o - ob * inner;
        const float* p = x + ob * R * inner + i;
        float s = 0.f;
        for (long long r = 0; r < R; ++r) {
            const float v = p[r * inner];
            s += v * v;
        }
        out[o] = sqrt_rn(s);
    }
}
__global__ void k_rss_complex(const float2* __restrict__ x, float* __restrict__ out, long long outer, long long R, long long inner) {
    const long long total = outer * inner;
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long long)gridDim.x * blockDim.x) {
        const long long ob = o / inner, i = o - ob * inner;
        const float2* p = x + ob * R * inner + i;
        float s = 0.f;
        for (long long r = 0; r < R; ++r) {
            const float2 v = p[r * inner];
            s += v.x * v.x + v.y * v.y;
        }
        out[o] = sqrt_rn(s);
    }
}
// x / rss_complex(x, dim) broadcast back over the reduced dim (models/base.py:824-840, BaseSensitivityModel): same summation order
// as k_rss_complex, then one IEEE division per component
__global__ void k_div_rss_complex(const float2* __restrict__ x, float2* __restrict__ out, long long outer, long long R, long long inner) {
    const long long total = outer * inner;
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long long)gridDim.x * blockDim.x) {
        const long long ob = o / inner, i = o - ob * inner;
        const long long base = ob * R * inner + i;
        float s = 0.f;
        for (long long r = 0; r < R; ++r) {
            const float2 v = x[base + r * inner];
            s += v.x * v.x + v.y * v.y;
        }
        const float den = sqrt_rn(s);
        for (long long r = 0; r < R; ++r) {
            const float2 v = x[base + r * inner];
            out[base + r * inner] = make_float2(v.x / den, v.y / den);
        }
    }
}
extern "C" int mrx_div_rss_complex(const float* x, float* out, int64_t outer, int64_t R, int64_t inner, void* stream) {
    MRX_REQUIRE(outer >= 0 && R >= 1 && inner >= 0, MRX_EINVAL, "mrx_div_rss_complex: bad dims");
    if (outer * inner == 0) return MRX_OK;
    MRX_REQUIRE(x && out, MRX_EINVAL, "mrx_div_rss_complex: null pointer");
    hipLaunchKernelGGL(k_div_rss_complex, dim3(ew_grid(outer * inner)), dim3(EW_NT), 0, (hipStream_t)stream, (const float2*)x, (float2*)out,
                       (long long)outer, (long long)R, (long long)inner);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
__global__ void k_sense(const float2* __restrict__ x, const float2* __restrict__ sm, float2* __restrict__ out, long long outer,
                        long long R, long long inner) {
    const long long total = outer * inner;
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long long)gridDim.x * blockDim.x) {
        const long long ob = o / inner, i = o - ob * inner;
        const long long base = ob * R * inner + i;
        float re = 0.f, im = 0.f;
        for (long long r = 0; r < R; ++r) {
            const float2 v = x[base + r * inner];
            const float2 s = sm[base + r * inner];
            re += v.x * s.x + v.y * s.y;
            im += v.y * s.x - v.x * s.y;
        }
        out[o] = make_float2(re, im);
    }
}
#define REDUCE_ENTRY(name, kern, TIN, TOUT, extra_decl, extra_arg)                                                          \
    extern "C" int name(const float* x, extra_decl float* out, int64_t outer, int64_t R, int64_t inner, void* stream) {      \
        MRX_REQUIRE(x && out && outer >= 0 && R >= 0 && inner >= 0, MRX_EINVAL, #name ": bad argument");                     \
        if (outer * inner == 0) return MRX_OK;                                                                               \
        hipLaunchKernelGGL(kern, dim3(ew_grid(outer* inner)), dim3(EW_NT), 0, (hipStream_t)stream, (const TIN*)x, extra_arg  \
                           (TOUT*)out, (long long)outer, (long long)R, (long long)inner);                                    \
        MRX_LAUNCH_CHECK();                                                                                                  \
        return MRX_OK;                                                                                                       \
    }
REDUCE_ENTRY(mrx_rss, k_rss, float, float, , )
REDUCE_ENTRY(mrx_rss_complex, k_rss_complex, float2, float, , )
extern "C" int mrx_sense(const float* x, const float* s, float* out, int64_t outer, int64_t R, int64_t inner, void* stream) {
    MRX_REQUIRE(x && s && out && outer >= 0 && R >= 0 && inner >= 0, MRX_EINVAL, "mrx_sense: bad argument");
    if (outer * inner == 0) return MRX_OK;
    hipLaunchKernelGGL(k_sense, dim3(ew_grid(outer * inner)), dim3(EW_NT), 0, (hipStream_t)stream, (const float2*)x,
                       (const float2*)s, (float2*)out, (long long)outer, (long long)R, (long long)inner);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- apply_mask arithmetic (utils.py:341) -------------------------------------------------------------------------
struct Dims4 {
    long long B, C, H, W, total;
};
__global__ void k_apply_mask(const float2* __restrict__ d, MrxMask m, float2* __restrict__ out, Dims4 s) {
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < s.total; o += (long long)gridDim.x * blockDim.x) {
        long long r = o;
        const long long w = r % s.W;
        r /= s.W;
        const long long h = r % s.H;
        r /= s.H;
        const long long c = r % s.C;
        const long long b = r / s.C;
        const float mv = mrx_mask_val(m, b, c, h, w);
        const float2 v = d[o];
        out[o] = make_float2(__fadd_rn(__fmul_rn(v.x, mv), 0.0f), __fadd_rn(__fmul_rn(v.y, mv), 0.0f));  // + 0.0 kills -0
    }
}
static int fill_mask(MrxMask* m, const void* mask, int kind, const int64_t* ms, const char* who) {
    MRX_REQUIRE(mask && ms, MRX_EINVAL, "%s: null mask", who);
    MRX_REQUIRE(kind == MRX_MASK_U8 || kind == MRX_MASK_F32, MRX_EINVAL, "%s: bad mask kind %d", who, kind);
    m->p = mask;
    m->kind = kind;
    for (int i = 0; i < 4; ++i) m->s[i] = ms[i];
    return MRX_OK;
}
static int fill_dims(Dims4* s, int B, int C, int H, int W, const char* who) {
    MRX_REQUIRE(B >= 0 && C >= 0 && H >= 0 && W >= 0, MRX_EINVAL, "%s: negative dim", who);
    s->B = B;
    s->C = C;
    s->H = H;
    s->W = W;
    s->total = (long long)B * C * H * W;
    return MRX_OK;
}
extern "C" int mrx_apply_mask(const float* data, const float* mask, float* out, int B, int C, int H, int W,
                              const int64_t* mstride, void* stream) {
    MRX_REQUIRE(data && out, MRX_EINVAL, "mrx_apply_mask: null pointer");
    MrxMask m;
    Dims4 s;
    int rc;
    if ((rc = fill_mask(&m, mask, MRX_MASK_F32, mstride, "mrx_apply_mask"))) return rc;
    if ((rc = fill_dims(&s, B, C, H, W, "mrx_apply_mask"))) return rc;
    if (s.total == 0) return MRX_OK;
    hipLaunchKernelGGL(k_apply_mask, dim3(ew_grid(s.total)), dim3(EW_NT), 0, (hipStream_t)stream, (const float2*)data, m, (float2*)out, s);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- soft data consistency (vn_block.py:109-117, rim_block.py:256-259) ------------------------------------------
// MODE 0: out = where(mask, pred - ref, 0) * w ; MODE 1: out = base - where(mask, pred - ref, 0) * w - eta_k
template <int MODE>
__global__ void k_soft_dc(const float2* __restrict__ base, const float2* __restrict__ pred, const float2* __restrict__ ref,
                          MrxMask m, const float* __restrict__ dcw, const float2* __restrict__ etak,
                          float2* __restrict__ out, Dims4 s) {
    const float w8 = dcw[0];
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < s.total; o += (long long)gridDim.x * blockDim.x) {
        long long r = o;
        const long long w = r % s.W;
        r /= s.W;
        const long long h = r % s.H;
        r /= s.H;
        const long long c = r % s.C;
        const long long b = r / s.C;
        const float2 p = pred[o];
        float2 d = make_float2(0.f, 0.f);
        if (mrx_mask_true(m, b, c, h, w)) {
            const float2 q = ref[o];
            d = make_float2(p.x - q.x, p.y - q.y);
        }
        d.x = __fmul_rn(d.x, w8);
        d.y = __fmul_rn(d.y, w8);
        if (MODE == 0) {
            out[o] = d;
        } else {
            const float2 e = etak[o];
            const float2 a = base[o];
            out[o] = make_float2(__fsub_rn(__fsub_rn(a.x, d.x), e.x), __fsub_rn(__fsub_rn(a.y, d.y), e.y));
        }
    }
}
extern "C" int mrx_soft_dc(const float* pred, const float* ref, const void* mask, int mask_kind, const int64_t* mstride,
                           const float* dc_weight, float* out, int B, int C, int H, int W, void* stream) {
    MRX_REQUIRE(pred && ref && dc_weight && out, MRX_EINVAL, "mrx_soft_dc: null pointer");
    MrxMask m;
    Dims4 s;
    int rc;
    if ((rc = fill_mask(&m, mask, mask_kind, mstride, "mrx_soft_dc"))) return rc;
    if ((rc = fill_dims(&s, B, C, H, W, "mrx_soft_dc"))) return rc;
    if (s.total == 0) return MRX_OK;
    hipLaunchKernelGGL(k_soft_dc<0>, dim3(ew_grid(s.total)), dim3(EW_NT), 0, (hipStream_t)stream, (const float2*)nullptr,
                       (const float2*)pred, (const float2*)ref, m, dc_weight, (const float2*)nullptr, (float2*)out, s);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
extern "C" int mrx_dc_combine(const float* base, const float* pred, const float* ref, const void* mask, int mask_kind,
                              const int64_t* mstride, const float* dc_weight, const float* eta_k, float* out, int B, int C,
                              int H, int W, void* stream) {
    MRX_REQUIRE(base && pred && ref && dc_weight && eta_k && out, MRX_EINVAL, "mrx_dc_combine: null pointer");
    MrxMask m;
    Dims4 s;
    int rc;
    if ((rc = fill_mask(&m, mask, mask_kind, mstride, "mrx_dc_combine"))) return rc;
    if ((rc = fill_dims(&s, B, C, H, W, "mrx_dc_combine"))) return rc;
    if (s.total == 0) return MRX_OK;
    hipLaunchKernelGGL(k_soft_dc<1>, dim3(ew_grid(s.total)), dim3(EW_NT), 0, (hipStream_t)stream, (const float2*)base,
                       (const float2*)pred, (const float2*)ref, m, dc_weight, (const float2*)eta_k, (float2*)out, s);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- VSNet hard data consistency and weighted average (variablesplittingnet/vsnet_block.py:23-25, :35-36) ---------
// Operation order and rounding follow the reference's separate torch ops (no contraction).
__global__ void k_hard_dc(const float2* __restrict__ pred, const float2* __restrict__ ref, MrxMask m, const float* __restrict__ dcw,
                          float2* __restrict__ out, Dims4 s) {
    const float w8 = dcw[0];
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < s.total; o += (long long)gridDim.x * blockDim.x) {
        long long r = o;
        const long long w = r % s.W;
        r /= s.W;
        const long long h = r % s.H;
        r /= s.H;
        const long long c = r % s.C;
        const long long b = r / s.C;
        const float mv = mrx_mask_val(m, b, c, h, w), om = __fsub_rn(1.0f, mv);
        const float2 p = pred[o], q = ref[o];
        out[o] = make_float2(__fmul_rn(__fadd_rn(__fmul_rn(om, p.x), __fmul_rn(mv, q.x)), w8),
                             __fmul_rn(__fadd_rn(__fmul_rn(om, p.y), __fmul_rn(mv, q.y)), w8));
    }
}
// out[b,c] = param * (k[b,c] + pred[b,c]) + (1 - param) * sx[b]   (sx: one image per batch element, broadcast over coils)
__global__ void k_vs_average(const float2* __restrict__ k, const float2* __restrict__ pred, const float2* __restrict__ sx,
                             const float* __restrict__ param, float2* __restrict__ out, long long C, long long HW, long long total) {
    const float pa = param[0], pb = __fsub_rn(1.0f, pa);
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long long)gridDim.x * blockDim.x) {
        const long long bc = o / HW, px = o - bc * HW, b = bc / C;
        const float2 a = k[o], p = pred[o], x = sx[b * HW + px];
        out[o] = make_float2(__fadd_rn(__fmul_rn(pa, __fadd_rn(a.x, p.x)), __fmul_rn(pb, x.x)),
                             __fadd_rn(__fmul_rn(pa, __fadd_rn(a.y, p.y)), __fmul_rn(pb, x.y)));
    }
}
extern "C" int mrx_hard_dc(const float* pred, const float* ref, const void* mask, int mask_kind, const int64_t* mstride,
                           const float* dc_weight, float* out, int B, int C, int H, int W, void* stream) {
    MRX_REQUIRE(pred && ref && dc_weight && out, MRX_EINVAL, "mrx_hard_dc: null pointer");
    MrxMask m;
    Dims4 s;
    int rc;
    if ((rc = fill_mask(&m, mask, mask_kind, mstride, "mrx_hard_dc"))) return rc;
    if ((rc = fill_dims(&s, B, C, H, W, "mrx_hard_dc"))) return rc;
    if (s.total == 0) return MRX_OK;
    hipLaunchKernelGGL(k_hard_dc, dim3(ew_grid(s.total)), dim3(EW_NT), 0, (hipStream_t)stream, (const float2*)pred, (const float2*)ref, m,
                       dc_weight, (float2*)out, s);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
extern "C" int mrx_vs_average(const float* kspace, const float* pred, const float* sx, const float* param, float* out, int B, int C,
                              int H, int W, void* stream) {
    MRX_REQUIRE(kspace && pred && sx && param && out, MRX_EINVAL, "mrx_vs_average: null pointer");
    MRX_REQUIRE(B >= 0 && C >= 0 && H >= 0 && W >= 0, MRX_EINVAL, "mrx_vs_average: negative dim");
    const long long total = (long long)B * C * H * W;
    if (total == 0) return MRX_OK;
    hipLaunchKernelGGL(k_vs_average, dim3(ew_grid(total)), dim3(EW_NT), 0, (hipStream_t)stream, (const float2*)kspace, (const float2*)pred,
                       (const float2*)sx, param, (float2*)out, (long long)C, (long long)H * W, total);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- sigmanet data-consistency layers (sigmanet/dc_layers.py): the pointwise pieces between the FFT / sensitivity kernels ------
// coil sum of (optionally masked) k-space: out[b,h,w] = sum_c k[b,c,h,w] * mask   (dc_layers.py:69-81: `fft2(x S) * mask` summed
// over axis -4 -- the reference really sums k-space over the coils)
__global__ void k_coil_sum(const float2* __restrict__ k, MrxMask m, int use_mask, float2* __restrict__ out, Dims4 s) {
    const long long HW = s.H * s.W, n = s.B * HW;
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < n; o += (long long)gridDim.x * blockDim.x) {
        const long long b = o / HW, px = o - b * HW, h = px / s.W, w = px - h * s.W;
        float ax = 0.f, ay = 0.f;
        for (long long c = 0; c < s.C; ++c) {
            const float2 v = k[(b * s.C + c) * HW + px];
            const float mv = use_mask ? mrx_mask_val(m, b, c, h, w) : 1.0f;
            ax = __fadd_rn(ax, __fmul_rn(v.x, mv));
            ay = __fadd_rn(ay, __fmul_rn(v.y, mv));
        }
        out[o] = make_float2(ax, ay);
    }
}
// MODE 0: out[b,c] = (a[b] - y[b,c]) * mask                                              (gradient-descent layer, dc_layers.py:82-87)
// MODE 1: out[b,c] = (1 - mask) * a + mask * (alpha * a + (1 - alpha) * y[b,c])          (variable splitting :381 / DCLayer :463)
//         a is one image per batch element (a_coils = 0, broadcast over the coils) or one per (b, c) (a_coils = 1)
template <int MODE>
__global__ void k_dc_bcast(const float2* __restrict__ a, int a_coils, const float2* __restrict__ y, MrxMask m,
                           const float* __restrict__ alpha, float2* __restrict__ out, Dims4 s) {
    const long long HW = s.H * s.W;
    const float al = MODE == 1 ? alpha[0] : 0.f, oal = __fsub_rn(1.0f, al);
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < s.total; o += (long long)gridDim.x * blockDim.x) {
        const long long bc = o / HW, px = o - bc * HW, b = bc / s.C, c = bc - b * s.C, h = px / s.W, w = px - h * s.W;
        const float mv = mrx_mask_val(m, b, c, h, w);
        const float2 av = a[(a_coils ? bc : b) * HW + px], yv = y[o];
        if (MODE == 0) {
            out[o] = make_float2(__fmul_rn(__fsub_rn(av.x, yv.x), mv), __fmul_rn(__fsub_rn(av.y, yv.y), mv));
        } else {
            const float om = __fsub_rn(1.0f, mv);
            out[o] = make_float2(
                __fadd_rn(__fmul_rn(om, av.x), __fmul_rn(mv, __fadd_rn(__fmul_rn(al, av.x), __fmul_rn(oal, yv.x)))),
                __fadd_rn(__fmul_rn(om, av.y), __fmul_rn(mv, __fadd_rn(__fmul_rn(al, av.y), __fmul_rn(oal, yv.y)))));
        }
    }
}
// out[i] = x[i % nx] - p * g[i % ng] (mode 0, dc_layers.py:96)  |  p * x[i % nx] + (1 - p) * g[i % ng] (mode 1, :402)
//        | p * g[i % ng] + x[i % nx] (mode 2, :250,:254)
__global__ void k_lincomb(const float* __restrict__ x, long long nx, const float* __restrict__ g, long long ng,
                          const float* __restrict__ p, int mode, float* __restrict__ out, long long n) {
    const float pv = p[0], op = __fsub_rn(1.0f, pv);
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < n; o += (long long)gridDim.x * blockDim.x) {
        const float xv = x[o % nx], gv = g[o % ng];
        out[o] = mode == 0 ? __fsub_rn(xv, __fmul_rn(pv, gv)) : mode == 1 ? __fadd_rn(__fmul_rn(pv, xv), __fmul_rn(op, gv)) : __fadd_rn(__fmul_rn(pv, gv), xv);
    }
}
extern "C" int mrx_coil_sum(const float* k, const void* mask, int mask_kind, const int64_t* mstride, float* out, int B, int C, int H,
                            int W, void* stream) {
    MRX_REQUIRE(k && out, MRX_EINVAL, "mrx_coil_sum: null pointer");
    MrxMask m;
    Dims4 s;
    int rc;
    m.p = nullptr;
    m.kind = MRX_MASK_F32;
    for (int i = 0; i < 4; ++i) m.s[i] = 0;
    if (mask && (rc = fill_mask(&m, mask, mask_kind, mstride, "mrx_coil_sum"))) return rc;
    if ((rc = fill_dims(&s, B, C, H, W, "mrx_coil_sum"))) return rc;
    const long long n = (long long)B * H * W;
    if (n == 0) return MRX_OK;
    hipLaunchKernelGGL(k_coil_sum, dim3(ew_grid(n)), dim3(EW_NT), 0, (hipStream_t)stream, (const float2*)k, m, mask ? 1 : 0, (float2*)out, s);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
extern "C" int mrx_dc_bcast(const float* a, int a_coils, const float* y, const void* mask, int mask_kind, const int64_t* mstride,
                            const float* alpha, int mode, float* out, int B, int C, int H, int W, void* stream) {
    MRX_REQUIRE(a && y && out && (mode == 0 || (mode == 1 && alpha)), MRX_EINVAL, "mrx_dc_bcast: bad argument");
    MrxMask m;
    Dims4 s;
    int rc;
    if ((rc = fill_mask(&m, mask, mask_kind, mstride, "mrx_dc_bcast"))) return rc;
    if ((rc = fill_dims(&s, B, C, H, W, "mrx_dc_bcast"))) return rc;
    if (s.total == 0) return MRX_OK;
    if (mode == 0)
        hipLaunchKernelGGL(k_dc_bcast<0>, dim3(ew_grid(s.total)), dim3(EW_NT), 0, (hipStream_t)stream, (const float2*)a, a_coils,
                           (const float2*)y, m, alpha, (float2*)out, s);
    else
        hipLaunchKernelGGL(k_dc_bcast<1>, dim3(ew_grid(s.total)), dim3(EW_NT), 0, (hipStream_t)stream, (const float2*)a, a_coils,
                           (const float2*)y, m, alpha, (float2*)out, s);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
extern "C" int mrx_lincomb(const float* x, int64_t nx, const float* g, int64_t ng, const float* p, int mode, float* out, int64_t n,
                           void* stream) {
    MRX_REQUIRE(x && g && p && out && nx > 0 && ng > 0 && n >= 0 && mode >= 0 && mode <= 2, MRX_EINVAL, "mrx_lincomb: bad argument");
    if (n == 0) return MRX_OK;
    hipLaunchKernelGGL(k_lincomb, dim3(ew_grid(n)), dim3(EW_NT), 0, (hipStream_t)stream, x, (long long)nx, g, (long long)ng, p, mode, out,
                       (long long)n);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- conjugate-gradient pieces of the proximal data layer (sigmanet/dc_layers.py:156-196) -------------------------------------
// complex dot product per batch element, out[b] = (sum re, sum im) of a * conj(b): fixed-order two-stage reduction (per-workgroup
// partials in fp32, combined in double), so results do not depend on scheduling.
#define CDOT_BLOCKS 256
__global__ void k_cdot_partial(const float2* __restrict__ a, const float2* __restrict__ b, float* __restrict__ work, long long n) {
    // grid (CDOT_BLOCKS, B); work[(batch*CDOT_BLOCKS + block)*2 + {0,1}]
    const long long base = (long long)blockIdx.y * n;
    float sr = 0.f, si = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float2 x = a[base + i], y = b[base + i];
        sr += x.x * y.x + x.y * y.y;
        si += x.y * y.x - x.x * y.y;
    }
    __shared__ float shr[EW_NT], shi[EW_NT];
    shr[threadIdx.x] = sr;
    shi[threadIdx.x] = si;
    __syncthreads();
    for (int st = EW_NT / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            shr[threadIdx.x] += shr[threadIdx.x + st];
            shi[threadIdx.x] += shi[threadIdx.x + st];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        work[((long long)blockIdx.y * gridDim.x + blockIdx.x) * 2 + 0] = shr[0];
        work[((long long)blockIdx.y * gridDim.x + blockIdx.x) * 2 + 1] = shi[0];
    }
}
__global__ void k_cdot_final(const float* __restrict__ work, int nb, float* __restrict__ out) {
    if (threadIdx.x == 0) {
        double r = 0.0, i = 0.0;
        for (int k = 0; k < nb; ++k) {
            r += (double)work[((long long)blockIdx.x * nb + k) * 2];
            i += (double)work[((long long)blockIdx.x * nb + k) * 2 + 1];
        }
        out[blockIdx.x * 2] = (float)r;
        out[blockIdx.x * 2 + 1] = (float)i;
    }
}
// alpha = rr * conj(pq) / |pq|^2 (dc_layers.py:186-189);  x += alpha * p;  r -= alpha * q     (:191-192)
__global__ void k_cg_step(float2* __restrict__ x, float2* __restrict__ r, const float2* __restrict__ p, const float2* __restrict__ q,
                          const float* __restrict__ rr, const float* __restrict__ pq, long long n, long long total) {
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long long)gridDim.x * blockDim.x) {
        const long long b = o / n;
        const float re1 = rr[2 * b], im1 = rr[2 * b + 1], re2 = pq[2 * b], im2 = pq[2 * b + 1];
        const float ab = sqrtf(mrx_sumsq2(re2, im2));
        const float den = __fmul_rn(ab, ab);
        const float ar = __fadd_rn(__fmul_rn(re1, re2), __fmul_rn(im1, im2)) / den;
        const float ai = __fsub_rn(__fmul_rn(im1, re2), __fmul_rn(re1, im2)) / den;
        const float2 pv = p[o], qv = q[o];
        float2 xv = x[o], rv = r[o];
        xv.x = __fadd_rn(xv.x, __fsub_rn(__fmul_rn(ar, pv.x), __fmul_rn(ai, pv.y)));
        xv.y = __fadd_rn(xv.y, __fadd_rn(__fmul_rn(ar, pv.y), __fmul_rn(ai, pv.x)));
        rv.x = __fsub_rn(rv.x, __fsub_rn(__fmul_rn(ar, qv.x), __fmul_rn(ai, qv.y)));
        rv.y = __fsub_rn(rv.y, __fadd_rn(__fmul_rn(ar, qv.y), __fmul_rn(ai, qv.x)));
        x[o] = xv;
        r[o] = rv;
    }
}
// beta = rr_new / rr (real);  p = r + beta * p   (dc_layers.py:193-195)
__global__ void k_cg_dir(float2* __restrict__ p, const float2* __restrict__ r, const float* __restrict__ rr_new,
                         const float* __restrict__ rr, long long n, long long total) {
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long long)gridDim.x * blockDim.x) {
        const long long b = o / n;
        const float beta = rr_new[2 * b] / rr[2 * b];
        const float2 pv = p[o], rv = r[o];
        p[o] = make_float2(__fadd_rn(rv.x, __fmul_rn(beta, pv.x)), __fadd_rn(rv.y, __fmul_rn(beta, pv.y)));
    }
}
extern "C" int64_t mrx_cdot_work_floats(int B) { return (int64_t)2 * CDOT_BLOCKS * (B > 0 ? B : 1); }
extern "C" int mrx_cdot(const float* a, const float* b, float* out, float* work, int B, int64_t n, void* stream) {
    MRX_REQUIRE(a && b && out && work && B >= 1 && B <= 65535 && n >= 1, MRX_EINVAL, "mrx_cdot: bad argument");
    const long long nbl = (n + EW_NT - 1) / EW_NT;
    const int nb = (int)(nbl < CDOT_BLOCKS ? nbl : CDOT_BLOCKS);
    hipLaunchKernelGGL(k_cdot_partial, dim3(nb, B), dim3(EW_NT), 0, (hipStream_t)stream, (const float2*)a, (const float2*)b, work,
                       (long long)n);
    hipLaunchKernelGGL(k_cdot_final, dim3(B), dim3(64), 0, (hipStream_t)stream, (const float*)work, nb, out);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
extern "C" int mrx_cg_step(float* x, float* r, const float* p, const float* q, const float* rr, const float* pq, int B, int64_t n,
                           void* stream) {
    MRX_REQUIRE(x && r && p && q && rr && pq && B >= 0 && n >= 0, MRX_EINVAL, "mrx_cg_step: bad argument");
    const long long total = (long long)B * n;
    if (total == 0) return MRX_OK;
    hipLaunchKernelGGL(k_cg_step, dim3(ew_grid(total)), dim3(EW_NT), 0, (hipStream_t)stream, (float2*)x, (float2*)r, (const float2*)p,
                       (const float2*)q, rr, pq, (long long)n, total);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
extern "C" int mrx_cg_dir(float* p, const float* r, const float* rr_new, const float* rr, int B, int64_t n, void* stream) {
    MRX_REQUIRE(p && r && rr_new && rr && B >= 0 && n >= 0, MRX_EINVAL, "mrx_cg_dir: bad argument");
    const long long total = (long long)B * n;
    if (total == 0) return MRX_OK;
    hipLaunchKernelGGL(k_cg_dir, dim3(ew_grid(total)), dim3(EW_NT), 0, (hipStream_t)stream, (float2*)p, (const float2*)r, rr_new, rr,
                       (long long)n, total);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- GRU / MGU gate math (rnn_cells.py:118-127, :255-261) --------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__global__ void k_gru(const float* __restrict__ ih, const float* __restrict__ hh, const float* h, float* out, int F, long long HW,
                      long long total) {
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long long)gridDim.x * blockDim.x) {
        const long long b = o / (F * HW), r = o - b * F * HW;  // r = f*HW + p
        const long long g = b * 3 * F * HW + r;
        const float rg = sigmoidf_(ih[g] + hh[g]);
        const float z = sigmoidf_(ih[g + F * HW] + hh[g + F * HW]);
        const float n = tanhf(ih[g + 2 * F * HW] + rg * hh[g + 2 * F * HW]);
        const float hv = h[o];
        out[o] = n * (1.0f - z) + z * hv;
    }
}
__global__ void k_mgu(const float* __restrict__ ih, const float* __restrict__ hh, const float* h, float* out, int F, long long HW,
                      long long total) {
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long long)gridDim.x * blockDim.x) {
        const long long b = o / (F * HW), r = o - b * F * HW;
        const long long g = b * 2 * F * HW + r;
        const float f = sigmoidf_(ih[g] + hh[g]);
        const float c = tanhf(ih[g + F * HW] + f * hh[g + F * HW]);
        const float hv = h[o];
        out[o] = c + f * (hv - c);
    }
}
// Conv2dGRU (recurrentvarnet/conv2gru.py:147-157), the pointwise steps of the unfused route:
//   k_mul_sigmoid   out = h * sigmoid(pre)                                   (state * reset, :151)
//   k_gru_blend     o = h * (1 - sigmoid(pu)) + tanh(po) * sigmoid(pu); out = o, out_relu = ReLU(o)   (:154-157)
__global__ void k_mul_sigmoid(const float* __restrict__ h, const float* __restrict__ pre, float* __restrict__ out, long long n) {
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < n; o += (long long)gridDim.x * blockDim.x)
        out[o] = (h ? h[o] : 0.f) * sigmoidf_(pre[o]);
}
__global__ void k_gru_blend(const float* __restrict__ h, const float* __restrict__ pu, const float* __restrict__ po,
                            float* __restrict__ out, float* __restrict__ out_relu, long long n) {
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < n; o += (long long)gridDim.x * blockDim.x) {
        const float u = sigmoidf_(pu[o]);
        const float v = (h ? h[o] : 0.f) * (1.0f - u) + tanhf(po[o]) * u;
        out[o] = v;
        if (out_relu) out_relu[o] = v > 0.f ? v : 0.f;
    }
}
extern "C" int mrx_mul_sigmoid(const float* h, const float* pre, float* out, int64_t n, void* stream) {
    MRX_REQUIRE(pre && out && n >= 0, MRX_EINVAL, "mrx_mul_sigmoid: bad argument");
    if (n == 0) return MRX_OK;
    hipLaunchKernelGGL(k_mul_sigmoid, dim3(ew_grid(n)), dim3(EW_NT), 0, (hipStream_t)stream, h, pre, out, (long long)n);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
extern "C" int mrx_gru_blend(const float* h, const float* pre_update, const float* pre_out, float* out, float* out_relu, int64_t n,
                             void* stream) {
    MRX_REQUIRE(pre_update && pre_out && out && n >= 0, MRX_EINVAL, "mrx_gru_blend: bad argument");
    if (n == 0) return MRX_OK;
    hipLaunchKernelGGL(k_gru_blend, dim3(ew_grid(n)), dim3(EW_NT), 0, (hipStream_t)stream, h, pre_update, pre_out, out, out_relu,
                       (long long)n);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
// Backward of the gate math (the derivative of rnn_cells.py:118-127 / :255-261): the gates are recomputed from ih / hh (three / two exponentials per element
// instead of three saved planes), every gradient leaves in one pass.  d_hh's reset / update (forget) planes equal d_ih's.
//   GRU:  dn = dy (1 - z), dz = dy (h - n), dh = dy z;  dn' = dn (1 - n^2): d i_n = dn', d h_n = dn' r, dr = dn' h_n;  d i_z = d h_z = dz z (1 - z);  d i_r = d h_r = dr r (1 - r)
//   MGU:  dc = dy (1 - f), df = dy (h - c), dh = dy f;  dc' = dc (1 - c^2): d i_c = dc', d h_c = dc' f, df += dc' h_c;  d i_f = d h_f = df f (1 - f)
__global__ void k_gru_bwd(const float* __restrict__ dy, const float* __restrict__ ih, const float* __restrict__ hh, const float* __restrict__ h,
                          float* __restrict__ dih, float* __restrict__ dhh, float* __restrict__ dh, int F, long long HW, long long total) {
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long long)gridDim.x * blockDim.x) {
        const long long b = o / (F * HW), r_ = o - b * F * HW;
        const long long g = b * 3 * F * HW + r_, P = F * HW;
        const float hn = hh[g + 2 * P];
        const float r = sigmoidf_(ih[g] + hh[g]), z = sigmoidf_(ih[g + P] + hh[g + P]), n = tanhf(ih[g + 2 * P] + r * hn);
        const float d = dy[o], hv = h[o];
        const float dnp = d * (1.0f - z) * (1.0f - n * n);
        const float dzp = d * (hv - n) * z * (1.0f - z);
        const float drp = dnp * hn * r * (1.0f - r);
        dih[g] = drp, dih[g + P] = dzp, dih[g + 2 * P] = dnp;
        dhh[g] = drp, dhh[g + P] = dzp, dhh[g + 2 * P] = dnp * r;
        dh[o] = d * z;
    }
}
__global__ void k_mgu_bwd(const float* __restrict__ dy, const float* __restrict__ ih, const float* __restrict__ hh, const float* __restrict__ h,
                          float* __restrict__ dih, float* __restrict__ dhh, float* __restrict__ dh, int F, long long HW, long long total) {
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long long)gridDim.x * blockDim.x) {
        const long long b = o / (F * HW), r_ = o - b * F * HW;
        const long long g = b * 2 * F * HW + r_, P = F * HW;
        const float hc = hh[g + P];
        const float f = sigmoidf_(ih[g] + hh[g]), c = tanhf(ih[g + P] + f * hc);
        const float d = dy[o], hv = h[o];
        const float dcp = d * (1.0f - f) * (1.0f - c * c);
        const float dfp = (d * (hv - c) + dcp * hc) * f * (1.0f - f);
        dih[g] = dfp, dih[g + P] = dcp;
        dhh[g] = dfp, dhh[g + P] = dcp * f;
        dh[o] = d * f;
    }
}
extern "C" int mrx_gru_gates_bwd(const float* dy, const float* ih, const float* hh, const float* h, float* dih, float* dhh, float* dh, int B, int F, int64_t HW,
                                 void* stream) {
    MRX_REQUIRE(dy && ih && hh && h && dih && dhh && dh && B >= 0 && F >= 0 && HW >= 0, MRX_EINVAL, "mrx_gru_gates_bwd: bad argument");
    const long long total = (long long)B * F * HW;
    if (total == 0) return MRX_OK;
    hipLaunchKernelGGL(k_gru_bwd, dim3(ew_grid(total)), dim3(EW_NT), 0, (hipStream_t)stream, dy, ih, hh, h, dih, dhh, dh, F, (long long)HW, total);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
extern "C" int mrx_mgu_gates_bwd(const float* dy, const float* ih, const float* hh, const float* h, float* dih, float* dhh, float* dh, int B, int F, int64_t HW,
                                 void* stream) {
    MRX_REQUIRE(dy && ih && hh && h && dih && dhh && dh && B >= 0 && F >= 0 && HW >= 0, MRX_EINVAL, "mrx_mgu_gates_bwd: bad argument");
    const long long total = (long long)B * F * HW;
    if (total == 0) return MRX_OK;
    hipLaunchKernelGGL(k_mgu_bwd, dim3(ew_grid(total)), dim3(EW_NT), 0, (hipStream_t)stream, dy, ih, hh, h, dih, dhh, dh, F, (long long)HW, total);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
extern "C" int mrx_gru_gates(const float* ih, const float* hh, const float* h, float* out, int B, int F, int64_t HW, void* stream) {
    MRX_REQUIRE(ih && hh && h && out && B >= 0 && F >= 0 && HW >= 0, MRX_EINVAL, "mrx_gru_gates: bad argument");
    const long long total = (long long)B * F * HW;
    if (total == 0) return MRX_OK;
    hipLaunchKernelGGL(k_gru, dim3(ew_grid(total)), dim3(EW_NT), 0, (hipStream_t)stream, ih, hh, h, out, F, (long long)HW, total);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
extern "C" int mrx_mgu_gates(const float* ih, const float* hh, const float* h, float* out, int B, int F, int64_t HW, void* stream) {
    MRX_REQUIRE(ih && hh && h && out && B >= 0 && F >= 0 && HW >= 0, MRX_EINVAL, "mrx_mgu_gates: bad argument");
    const long long total = (long long)B * F * HW;
    if (total == 0) return MRX_OK;
    hipLaunchKernelGGL(k_mgu, dim3(ew_grid(total)), dim3(EW_NT), 0, (hipStream_t)stream, ih, hh, h, out, F, (long long)HW, total);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
