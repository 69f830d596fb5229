// qmri.hip -- quantitative-MRI (MEGRE) signal model and analytic log-likelihood gradient for qRIM / qCIRIM
// (reference mridc/collections/quantitative/models/qrim/utils.py:71-121, :250-295; qrim_block.py:198-236).
// Pointwise kernels over parameter maps; the k-space part of the gradient is mrx_dc_residual (fft.hip).
#include "mrx_common.h"

#define QM_NT 256
#define QM_MAX_TE 16
struct TEs {
    int n;
    float te[QM_MAX_TE];
};
static inline int qm_grid(long long n) {
    long long g = (n + QM_NT - 1) / QM_NT;
    if (g > 4096) g = 4096;
    return g < 1 ? 1 : (int)g;
}
__device__ __forceinline__ float nan0(float v) { return v != v ? 0.f : v; }

// maps [N, HW] x4 -> signal [N, E, HW, 2]   (utils.py:95-121; NaN -> 0 at :120)
__global__ void k_qmri_signal(const float* __restrict__ r2, const float* __restrict__ s0, const float* __restrict__ b0,
                              const float* __restrict__ ph, float2* __restrict__ out, long long N, long long HW, TEs t,
                              float scaling) {
    const long long total = N * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / HW, p = i - n * HW;
        const float R = r2[i], S0 = s0[i], B0 = b0[i], PH = ph[i];
        for (int e = 0; e < t.n; ++e) {
            const float ft = expf(-t.te[e] * scaling * R);
            const float c = cosf(B0 * scaling * -t.te[e]);
            const float s = sinf(B0 * scaling * -t.te[e]);
            const float re = S0 * ft * c - PH * ft * s;
            const float im = S0 * ft * s + PH * ft * c;
            out[(n * t.n + e) * HW + p] = make_float2(nan0(re), nan0(im));
        }
    }
}
extern "C" int mrx_qmri_signal(const float* r2, const float* s0, const float* b0, const float* phi, const float* tes, int E,
                               float* out, int64_t N, int64_t HW, float scaling, void* stream) {
    MRX_REQUIRE(r2 && s0 && b0 && phi && tes && out, MRX_EINVAL, "mrx_qmri_signal: null pointer");
    MRX_REQUIRE(E >= 1 && E <= QM_MAX_TE && N >= 0 && HW >= 0, MRX_EINVAL, "mrx_qmri_signal: bad dims (E=%d)", E);
    if (N * HW == 0) return MRX_OK;
    TEs t;
    t.n = E;
    for (int i = 0; i < E; ++i) t.te[i] = tes[i];  // tes is a HOST array (the echo times come from the acquisition header)
    hipLaunchKernelGGL(k_qmri_signal, dim3(qm_grid(N * HW)), dim3(QM_NT), 0, (hipStream_t)stream, r2, s0, b0, phi, (float2*)out,
                       (long long)N, (long long)HW, t, scaling);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// dinv [N,E,HW,2] (coil-combined residual) + maps -> grad [N,4,HW] = mean_e of (R2*_re, S0_re, R2*_im, S0_im), times `post`
// (utils.py:250-295; qrim_block.py:223-224: /100 and NaN -> 0).
__global__ void k_qmri_grad(const float2* __restrict__ dinv, const float* __restrict__ r2, const float* __restrict__ s0,
                            const float* __restrict__ b0, const float* __restrict__ ph, float* __restrict__ out, long long N,
                            long long HW, TEs t, float scaling, float post) {
    const long long total = N * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / HW, p = i - n * HW;
        const float R = r2[i], S0 = s0[i], B0 = b0[i], PH = ph[i];
        float r2re = 0.f, r2im = 0.f, s0re = 0.f, s0im = 0.f;
        for (int e = 0; e < t.n; ++e) {
            const float te = t.te[e];
            const float ft = expf(-te * scaling * R);
            const float c = cosf(B0 * scaling * -te);
            const float s = sinf(B0 * scaling * -te);
            const float2 d = dinv[(n * t.n + e) * HW + p];
            const float s0d0 = ft * c, s0d1 = -ft * s;                                   // utils.py:259-261
            const float r2d0 = -te * scaling * ft * (S0 * c - PH * s);                   // :263-275
            const float r2d1 = -te * scaling * ft * (-S0 * s - PH * c);
            s0re += d.x * s0d0 - d.y * s0d1;                                             // :277-288
            s0im += d.x * s0d1 + d.y * s0d0;
            r2re += d.x * r2d0 - d.y * r2d1;
            r2im += d.x * r2d1 + d.y * r2d0;
        }
        const float inv = post / (float)t.n;                                            // mean over echoes, then /100
        float* o = out + n * 4 * HW + p;
        o[0] = nan0(r2re * inv);
        o[HW] = nan0(s0re * inv);
        o[2 * HW] = nan0(r2im * inv);
        o[3 * HW] = nan0(s0im * inv);
    }
}
extern "C" int mrx_qmri_grad(const float* dinv, const float* r2, const float* s0, const float* b0, const float* phi,
                             const float* tes, int E, float* out, int64_t N, int64_t HW, float scaling, float post, void* stream) {
    MRX_REQUIRE(dinv && r2 && s0 && b0 && phi && tes && out, MRX_EINVAL, "mrx_qmri_grad: null pointer");
    MRX_REQUIRE(E >= 1 && E <= QM_MAX_TE && N >= 0 && HW >= 0, MRX_EINVAL, "mrx_qmri_grad: bad dims (E=%d)", E);
    if (N * HW == 0) return MRX_OK;
    TEs t;
    t.n = E;
    for (int i = 0; i < E; ++i) t.te[i] = tes[i];
    hipLaunchKernelGGL(k_qmri_grad, dim3(qm_grid(N * HW)), dim3(QM_NT), 0, (hipStream_t)stream, (const float2*)dinv, r2, s0, b0, phi,
                       out, (long long)N, (long long)HW, t, scaling, post);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// mode bit 0: take |x| first; bit 1: divide by s instead of multiplying   (qrim_block.py:198-201; qcirim.py:248-251,:334)
__global__ void k_scale(const float* __restrict__ x, float* __restrict__ out, long long n, float s, int mode) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float v = (mode & 1) ? fabsf(x[i]) : x[i];
        out[i] = (mode & 2) ? v / s : v * s;
    }
}
extern "C" int mrx_scale(const float* x, float* out, int64_t n, float s, int mode, void* stream) {
    MRX_REQUIRE(x && out && n >= 0, MRX_EINVAL, "mrx_scale: bad argument");
    if (n == 0) return MRX_OK;
    hipLaunchKernelGGL(k_scale, dim3(qm_grid(n)), dim3(QM_NT), 0, (hipStream_t)stream, x, out, (long long)n, s, mode);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// eta_out[B,4,HW] = eta + delta, channel 0 (R2*) clamped at 0   (qrim_block.py:233-236)
__global__ void k_qrim_update(const float* __restrict__ eta, const float* __restrict__ delta, float* __restrict__ out, long long HW,
                              int Cc, long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long ch = (i / HW) % Cc;
        float v = eta[i] + delta[i];
        if (ch == 0 && v < 0.f) v = 0.f;
        out[i] = v;
    }
}
extern "C" int mrx_qrim_update(const float* eta, const float* delta, float* out, int B, int Cc, int64_t HW, void* stream) {
    MRX_REQUIRE(eta && delta && out && B >= 0 && Cc >= 1 && HW >= 0, MRX_EINVAL, "mrx_qrim_update: bad argument");
    const long long total = (long long)B * Cc * HW;
    if (total == 0) return MRX_OK;
    hipLaunchKernelGGL(k_qrim_update, dim3(qm_grid(total)), dim3(QM_NT), 0, (hipStream_t)stream, eta, delta, out, (long long)HW, Cc, total);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
