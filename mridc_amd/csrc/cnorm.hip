// cnorm.hip -- complex instance normalisation around a regulariser (reference models/sigmanet/sensitivity_net.py:16-139, "Deep Complex
// Networks" whitening): x -> C^(-1/2) (x - m) clamped to [-6, 6], and back y -> C^(1/2) y + m, with m the mean over every real and
// imaginary entry of the input and C the 2x2 covariance of (re, im) per batch element.
//
//   mrx_cnorm_stats   : three launches -- partial sums of all entries; centred second moments per batch element (each block finishes
//                       the mean from the partials in a fixed order first); one thread per batch element forms C^(1/2) and its inverse
//                       in double.  coef[b] = {m, h_xx, h_xy, h_yx, h_yy, i_xx, i_xy, i_yx, i_yy}.
//   mrx_cnorm_apply   : [B,C,H,W,2] -> the regulariser's [B C, 2, H, W] input (normalise + clamp + the reference's view / permute in one pass)
//   mrx_cnorm_unapply : the regulariser's [B C, 2, H, W] output -> [B,C,H,W,2] (permute back + un-normalise in one pass)
// The reference computes C^(1/2) from the eigen-decomposition of C with hand-normalised eigenvectors (sensitivity_net.py:55-83), which is
// 0 / 0 for c_xy = 0; here the principal square root of the symmetric positive matrix is formed directly,
//     C^(1/2) = (C + sqrt(det C) I) / sqrt(tr C + 2 sqrt(det C)),
// the same matrix wherever the reference's is defined.  Reductions are fixed-order (bit-reproducible).
#include "mrx_common.h"

#define CN_NT 256
#define CN_BLOCKS 256

__device__ __forceinline__ double cn_block_sum(double v, double* red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < CN_NT / 64; ++w) s += red[w];
    __syncthreads();
    return s;
}

// part[blk] = sum of x[i] over the block's grid-stride share of the n floats
__global__ __launch_bounds__(CN_NT) void k_cnorm_sum(const float* __restrict__ x, long long n, double* __restrict__ part) {
    __shared__ double red[CN_NT / 64];
    double s = 0.0;
    for (long long i = (long long)blockIdx.x * CN_NT + threadIdx.x; i < n; i += (long long)gridDim.x * CN_NT) s += (double)x[i];
    s = cn_block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// mom[(b * nblk + blk) * 3 + {0,1,2}] = sums of (re - m)^2, (im - m)^2, (re - m)(im - m) over the block's share of batch element b
// (blockIdx.y = b); m = (sum of part) / n_all, the differences rounded to fp32 as the reference's `input - mean` is
__global__ __launch_bounds__(CN_NT) void k_cnorm_moments(const float2* __restrict__ x, long long per_b, const double* __restrict__ part, int npart,
                                                         long long n_all, int center, double* __restrict__ mom) {
    __shared__ double red[CN_NT / 64];
    double tot = 0.0;
    for (int i = 0; i < npart; ++i) tot += part[i];
    const float m = center ? (float)(tot / (double)n_all) : 0.f;
    const float2* xb = x + (long long)blockIdx.y * per_b;
    double sxx = 0.0, syy = 0.0, sxy = 0.0;
    for (long long i = (long long)blockIdx.x * CN_NT + threadIdx.x; i < per_b; i += (long long)gridDim.x * CN_NT) {
        const float2 v = xb[i];
        const float re = v.x - m, im = v.y - m;
        sxx += (double)re * re, syy += (double)im * im, sxy += (double)re * im;
    }
    sxx = cn_block_sum(sxx, red), syy = cn_block_sum(syy, red), sxy = cn_block_sum(sxy, red);
    if (threadIdx.x == 0) {
        double* o = mom + ((long long)blockIdx.y * gridDim.x + blockIdx.x) * 3;
        o[0] = sxx, o[1] = syy, o[2] = sxy;
    }
}

__global__ void k_cnorm_finalize(const double* __restrict__ part, int npart, long long n_all, int center, const double* __restrict__ mom, int nblk,
                                 double divisor, int B, float* __restrict__ coef) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    double tot = 0.0;
    for (int i = 0; i < npart; ++i) tot += part[i];
    const float m = center ? (float)(tot / (double)n_all) : 0.f;
    double cxx = 0.0, cyy = 0.0, cxy = 0.0;
    for (int k = 0; k < nblk; ++k) {
        const double* o = mom + ((long long)b * nblk + k) * 3;
        cxx += o[0], cyy += o[1], cxy += o[2];
    }
    cxx /= divisor, cyy /= divisor, cxy /= divisor;
    const double sd = sqrt(cxx * cyy - cxy * cxy), t = sqrt(cxx + cyy + 2.0 * sd);
    const double hxx = (cxx + sd) / t, hyy = (cyy + sd) / t, hxy = cxy / t;      // C^(1/2), symmetric
    const double det = hxx * hyy - hxy * hxy;
    float* o = coef + (long long)b * 9;
    o[0] = m;
    o[1] = (float)hxx, o[2] = (float)hxy, o[3] = (float)hxy, o[4] = (float)hyy;
    o[5] = (float)(hyy / det), o[6] = (float)(-hxy / det), o[7] = (float)(-hxy / det), o[8] = (float)(hxx / det);
}

extern "C" int64_t mrx_cnorm_work_doubles(int B) { return (int64_t)CN_BLOCKS + (int64_t)B * CN_BLOCKS * 3; }

// x: B batch elements of per_b complex values each (contiguous); divisor: what the covariance sums are divided by (the reference's
// shape[2] * shape[3] - 1 of whatever rank it is handed); center = 0: the data is taken as mean-free (m = 0); coef: [B][9] floats;
// work: mrx_cnorm_work_doubles(B) doubles.
extern "C" int mrx_cnorm_stats(const float* x, int B, int64_t per_b, double divisor, int center, float* coef, double* work, void* stream) {
    MRX_REQUIRE(x && coef && work && B >= 1 && per_b >= 1 && B <= 65535, MRX_EINVAL, "mrx_cnorm_stats: bad argument");
    hipStream_t st = (hipStream_t)stream;
    const long long n_all = (long long)B * per_b * 2;
    const long long want = (n_all + CN_NT - 1) / CN_NT;
    const int npart = (int)(want < CN_BLOCKS ? want : CN_BLOCKS);
    const long long wantb = (per_b + CN_NT - 1) / CN_NT;
    const int nblk = (int)(wantb < CN_BLOCKS ? wantb : CN_BLOCKS);
    double* part = work;
    double* mom = work + CN_BLOCKS;
    hipLaunchKernelGGL(k_cnorm_sum, dim3(npart), dim3(CN_NT), 0, st, x, n_all, part);
    hipLaunchKernelGGL(k_cnorm_moments, dim3(nblk, B), dim3(CN_NT), 0, st, reinterpret_cast<const float2*>(x), (long long)per_b, part, npart, n_all, center, mom);
    hipLaunchKernelGGL(k_cnorm_finalize, dim3((B + 63) / 64), dim3(64), 0, st, part, npart, n_all, center, mom, nblk, divisor, B, coef);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// x [B][C][plane] complex; coef row of image (b, c) = coef[(cb ? b : 0)] (one coefficient set per batch element, shared by its C images);
// out [(b C + c)][2][plane] = clamp(Hinv (x - m), -6, 6)
__global__ void k_cnorm_apply(const float2* __restrict__ x, const float* __restrict__ coef, float* __restrict__ out, long long plane, int C) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= plane) return;
    const int img = blockIdx.y;
    const float* k = coef + (long long)(img / C) * 9;
    const float m = k[0];
    const float2 v = x[(long long)img * plane + p];
    const float re = v.x - m, im = v.y - m;
    float a = k[5] * re + k[6] * im, b = k[7] * re + k[8] * im;
    a = fminf(fmaxf(a, -6.f), 6.f), b = fminf(fmaxf(b, -6.f), 6.f);
    out[((long long)img * 2) * plane + p] = a;
    out[((long long)img * 2 + 1) * plane + p] = b;
}
__global__ void k_cnorm_unapply(const float* __restrict__ y, const float* __restrict__ coef, float2* __restrict__ out, long long plane, int C) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= plane) return;
    const int img = blockIdx.y;
    const float* k = coef + (long long)(img / C) * 9;
    const float re = y[((long long)img * 2) * plane + p], im = y[((long long)img * 2 + 1) * plane + p];
    out[(long long)img * plane + p] = make_float2(k[1] * re + k[2] * im + k[0], k[3] * re + k[4] * im + k[0]);
}
extern "C" int mrx_cnorm_apply(const float* x, const float* coef, float* out, int B, int C, int64_t plane, void* stream) {
    MRX_REQUIRE(x && coef && out && B >= 1 && C >= 1 && plane >= 1 && (long long)B * C <= 65535, MRX_EINVAL, "mrx_cnorm_apply: bad argument");
    hipLaunchKernelGGL(k_cnorm_apply, dim3((unsigned)((plane + 255) / 256), B * C), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float2*>(x), coef,
                       out, (long long)plane, C);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
extern "C" int mrx_cnorm_unapply(const float* y, const float* coef, float* out, int B, int C, int64_t plane, void* stream) {
    MRX_REQUIRE(y && coef && out && B >= 1 && C >= 1 && plane >= 1 && (long long)B * C <= 65535, MRX_EINVAL, "mrx_cnorm_unapply: bad argument");
    hipLaunchKernelGGL(k_cnorm_unapply, dim3((unsigned)((plane + 255) / 256), B * C), dim3(256), 0, (hipStream_t)stream, y, coef,
                       reinterpret_cast<float2*>(out), (long long)plane, C);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
