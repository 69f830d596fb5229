// diff_bwd.hip -- the pointwise / per-plane backward steps of the E2EVN training path (mridc_amd/diff.py; the reference trains the U-Net of
// unet_block.py:189-299 through torch autograd: LeakyReLU, InstanceNorm2d, avg_pool2d, ConvTranspose2d backward).  Round 4: these were torch device
// ops (torch.where, F.interpolate, F.pixel_unshuffle, means over planes) inside the backward; the convolution / transposed-convolution / FFT
// gradients already ran on this library.
//   mrx_act_bwd             dx = dy * act'(y)            (y = the activation's OUTPUT: ReLU / LeakyReLU keep the sign)
//   mrx_inorm_act_bwd       backward of act(InstanceNorm2d(x)) from the activation's output: with z = the normalised value recovered from the output
//                           (z = y for y > 0, y / slope otherwise), g = dy * act'(y):  dx = rstd * (g - mean(g) - z * mean(g z))  per plane;
//                           rstd from the forward's partial sums (the `work` buffer of mrx_instance_norm_act); two deterministic passes
//   mrx_avgpool2x2_bwd      dx[h][w] = dy[h / 2][w / 2] / 4 inside the pooled region, 0 in an odd last row / column
//   mrx_pixel_unshuffle2    [B,C,2H,2W] -> [B,4C,H,W], channel (c, i, j): the layout in which ConvTranspose2d(k 2, s 2)'s two gradients are 1x1 GEMMs
//   mrx_cmul_bcast          out[b,c] = a[b,c] * v[b] (or conj(v[b])) * scale: the sensitivity-map gradient of sens_reduce (vn_block.py:71-87):
//                           dS_c = conj(dy) * ifft2(k)_c
//   mrx_sens_expand_bwd_pw  the pointwise half of sens_expand's backward (vn_block.py:51-69: out_c = fft2(x S_c)) from G_c = adjoint-fft2(dy_c):
//                           dx = scale * sum_c conj(S_c) G_c,  dS_c = scale * conj(x) G_c  -- one pass over G and S
//   mrx_dc_combine_bwd      backward of base - where(mask, pred - ref, 0) * w - eta_k (vn_block.py:113-119): dpred = -where(mask, dy, 0) * w
//                           (+ dy when base and pred are the same tensor), deta = -dy, per-workgroup partial sums of where(mask, (pred - ref) . dy)
//                           for dw = -sum
#include "mrx_common.h"

#define DB_NT 256
static inline int db_nsplit(long long n) { return mrx_norm_nsplit(n); }      // the forward's rule (mrx_common.h): k_inorm_bwd_apply indexes ITS work buffer
static inline unsigned db_grid(long long n) {
    long long g = (n + DB_NT - 1) / DB_NT;
    return (unsigned)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}
__device__ __forceinline__ float db_actgrad(float dy, float y, int act, float slope) {
    return act == MRX_ACT_NONE ? dy : (y > 0.f ? dy : (act == MRX_ACT_LEAKY ? dy * slope : 0.f));
}
__global__ void k_act_bwd(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx, long long n, int act, float slope) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dx[i] = db_actgrad(dy[i], y[i], act, slope);
}
extern "C" int mrx_act_bwd(const float* dy, const float* y, float* dx, int64_t n, int act, float slope, void* stream) {
    MRX_REQUIRE(dy && y && dx && n >= 0, MRX_EINVAL, "mrx_act_bwd: bad argument");
    MRX_REQUIRE(act == MRX_ACT_NONE || act == MRX_ACT_RELU || act == MRX_ACT_LEAKY, MRX_EINVAL, "mrx_act_bwd: bad activation %d", act);
    if (n == 0) return MRX_OK;
    hipLaunchKernelGGL(k_act_bwd, dim3(db_grid(n)), dim3(DB_NT), 0, (hipStream_t)stream, dy, y, dx, (long long)n, act, slope);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

__device__ __forceinline__ float db_block_sum(float v, float* red) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = red[0];
    for (int k = 1; k < DB_NT / 64; ++k) t += red[k];
    return t;
}
__device__ __forceinline__ void db_range(long long n, int nsplit, int s, long long& a, long long& b) {
    const long long per = (n + nsplit - 1) / nsplit;
    a = (long long)s * per;
    b = a + per < n ? a + per : n;
}
// partial sums of g and g z per (plane, split)
__global__ __launch_bounds__(DB_NT) void k_inorm_bwd_sums(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ part, long long n, int nsplit,
                                                          int act, float slope) {
    __shared__ float red[DB_NT / 64];
    long long a, b;
    db_range(n, nsplit, blockIdx.y, a, b);
    const float* pd = dy + (long long)blockIdx.x * n;
    const float* py = y + (long long)blockIdx.x * n;
    const float inv_slope = act == MRX_ACT_LEAKY ? 1.0f / slope : 1.0f;
    float s0 = 0.f, s1 = 0.f;
    for (long long i = a + threadIdx.x; i < b; i += DB_NT) {
        const float yy = py[i], g = db_actgrad(pd[i], yy, act, slope), z = yy > 0.f ? yy : yy * inv_slope;
        s0 += g;
        s1 += g * z;
    }
    s0 = db_block_sum(s0, red);
    s1 = db_block_sum(s1, red);
    if (threadIdx.x == 0) {
        part[((long long)blockIdx.x * nsplit + blockIdx.y) * 2] = s0;
        part[((long long)blockIdx.x * nsplit + blockIdx.y) * 2 + 1] = s1;
    }
}
__global__ __launch_bounds__(DB_NT) void k_inorm_bwd_apply(const float* __restrict__ dy, const float* __restrict__ y, const float* __restrict__ fwd_work,
                                                           const float* __restrict__ part, float* __restrict__ dx, long long planes, long long n, int nsplit,
                                                           float eps, int act, float slope) {
    long long a, b;
    db_range(n, nsplit, blockIdx.y, a, b);
    const long long p = blockIdx.x;
    // the forward's partial squared deviations (mrx_instance_norm_act: work = [planes][nsplit] sums, then [planes][nsplit] squared deviations)
    double sq = 0.0, g0 = 0.0, g1 = 0.0;
    for (int i = 0; i < nsplit; ++i) {
        sq += (double)fwd_work[(planes + p) * nsplit + i];
        g0 += (double)part[(p * nsplit + i) * 2];
        g1 += (double)part[(p * nsplit + i) * 2 + 1];
    }
    const float rstd = 1.0f / sqrtf((float)sq / (float)n + eps), gm = (float)(g0 / (double)n), gz = (float)(g1 / (double)n);
    const float inv_slope = act == MRX_ACT_LEAKY ? 1.0f / slope : 1.0f;
    const float* pd = dy + p * n;
    const float* py = y + p * n;
    float* px = dx + p * n;
    for (long long i = a + threadIdx.x; i < b; i += DB_NT) {
        const float yy = py[i], g = db_actgrad(pd[i], yy, act, slope), z = yy > 0.f ? yy : yy * inv_slope;
        px[i] = rstd * (g - gm - z * gz);
    }
}
extern "C" int64_t mrx_inorm_act_bwd_work_floats(int64_t planes, int64_t n) { return planes < 0 || n < 1 ? -1 : 2 * planes * db_nsplit(n); }
extern "C" int mrx_inorm_act_bwd(const float* dy, const float* y, const float* fwd_work, float* dx, float* work, int64_t planes, int64_t HW, float eps, int act,
                                 float slope, void* stream) {
    MRX_REQUIRE(dy && y && fwd_work && dx && work && planes >= 0 && HW >= 1, MRX_EINVAL, "mrx_inorm_act_bwd: bad argument");
    MRX_REQUIRE(act == MRX_ACT_NONE || act == MRX_ACT_LEAKY, MRX_EUNSUP, "mrx_inorm_act_bwd: the normalised value is not recoverable from a ReLU's output");
    MRX_REQUIRE(act == MRX_ACT_NONE || slope > 0.f, MRX_EINVAL, "mrx_inorm_act_bwd: LeakyReLU slope must be positive");
    if (planes == 0) return MRX_OK;
    const int ns = db_nsplit(HW);
    dim3 grid((unsigned)planes, ns);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_inorm_bwd_sums, grid, dim3(DB_NT), 0, st, dy, y, work, (long long)HW, ns, act, slope);
    hipLaunchKernelGGL(k_inorm_bwd_apply, grid, dim3(DB_NT), 0, st, dy, y, fwd_work, (const float*)work, dx, (long long)planes, (long long)HW, ns, eps, act, slope);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

__global__ void k_avgpool2x2_bwd(const float* __restrict__ dy, float* __restrict__ dx, long long planes, int H, int W) {
    const int Ho = H / 2, Wo = W / 2;
    const long long n = planes * H * W;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long p = i / ((long long)H * W);
        const int r = (int)(i - p * H * W), h = r / W, w = r - h * W;
        dx[i] = (h < 2 * Ho && w < 2 * Wo) ? 0.25f * dy[(p * Ho + (h >> 1)) * Wo + (w >> 1)] : 0.f;
    }
}
extern "C" int mrx_avgpool2x2_bwd(const float* dy, float* dx, int64_t planes, int H, int W, void* stream) {
    MRX_REQUIRE(dy && dx && planes >= 0 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_avgpool2x2_bwd: bad argument");
    if (planes == 0) return MRX_OK;
    hipLaunchKernelGGL(k_avgpool2x2_bwd, dim3(db_grid(planes * H * W)), dim3(DB_NT), 0, (hipStream_t)stream, dy, dx, (long long)planes, H, W);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// out[b][(c * 2 + i) * 2 + j][h][w] = x[b][c][2 h + i][2 w + j]   (x [B,C,2H,2W])
__global__ void k_pixel_unshuffle2(const float* __restrict__ x, float* __restrict__ out, long long BC, int H, int W) {
    const long long n = BC * 4 * H * W;
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < n; o += (long long)gridDim.x * blockDim.x) {
        const int w = (int)(o % W), h = (int)((o / W) % H), ij = (int)((o / ((long long)W * H)) & 3);
        const long long bc = o / ((long long)W * H * 4);
        out[o] = x[(bc * 2 * H + 2 * h + (ij >> 1)) * 2 * W + 2 * w + (ij & 1)];
    }
}
extern "C" int mrx_pixel_unshuffle2(const float* x, float* out, int64_t BC, int H, int W, void* stream) {
    MRX_REQUIRE(x && out && BC >= 0 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_pixel_unshuffle2: bad argument");
    if (BC == 0) return MRX_OK;
    hipLaunchKernelGGL(k_pixel_unshuffle2, dim3(db_grid(BC * 4 * H * W)), dim3(DB_NT), 0, (hipStream_t)stream, x, out, (long long)BC, H, W);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}


// ---- coil operators and data consistency of the VarNet block (vn_block.py:51-119): the pointwise halves of their backward -------------------------
__global__ void k_cmul_bcast(const float2* __restrict__ a, const float2* __restrict__ v, float2* __restrict__ out, long long C, long long N, long long total,
                             int conj_v, float scale) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / (C * N), n = i % N;
        const float2 x = a[i];
        float2 y = v[b * N + n];
        if (conj_v) y.y = -y.y;
        out[i] = make_float2((x.x * y.x - x.y * y.y) * scale, (x.x * y.y + x.y * y.x) * scale);
    }
}
extern "C" int mrx_cmul_bcast(const float* a, const float* v, float* out, int64_t B, int64_t C, int64_t N, int conj_v, float scale, void* stream) {
    MRX_REQUIRE(a && v && out && B >= 0 && C >= 0 && N >= 0, MRX_EINVAL, "mrx_cmul_bcast: bad argument");
    const long long total = (long long)B * C * N;
    if (total == 0) return MRX_OK;
    hipLaunchKernelGGL(k_cmul_bcast, dim3(db_grid(total)), dim3(DB_NT), 0, (hipStream_t)stream, (const float2*)a, (const float2*)v, (float2*)out, (long long)C,
                       (long long)N, total, conj_v, scale);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

__global__ void k_sens_expand_bwd_pw(const float2* __restrict__ G, const float2* __restrict__ S, const float2* __restrict__ x, float2* __restrict__ dx,
                                     float2* __restrict__ dS, long long C, long long N, long long BN, float scale) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < BN; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / N, n = i - b * N;
        const float2 xv = x ? x[i] : make_float2(0.f, 0.f);
        float sx = 0.f, sy = 0.f;
        for (long long c = 0; c < C; ++c) {
            const long long o = (b * C + c) * N + n;
            const float2 g = G[o], sv = S[o];
            sx += sv.x * g.x + sv.y * g.y;                 // conj(S) G
            sy += sv.x * g.y - sv.y * g.x;
            if (dS) dS[o] = make_float2((xv.x * g.x + xv.y * g.y) * scale, (xv.x * g.y - xv.y * g.x) * scale);      // conj(x) G
        }
        if (dx) dx[i] = make_float2(sx * scale, sy * scale);
    }
}
extern "C" int mrx_sens_expand_bwd_pw(const float* G, const float* S, const float* x, float* dx, float* dS, int64_t B, int64_t C, int64_t N, float scale,
                                      void* stream) {
    MRX_REQUIRE(G && S && (dx || dS) && (x || !dS) && B >= 0 && C >= 0 && N >= 0, MRX_EINVAL, "mrx_sens_expand_bwd_pw: bad argument");
    const long long BN = (long long)B * N;
    if (BN == 0 || C == 0) return MRX_OK;
    hipLaunchKernelGGL(k_sens_expand_bwd_pw, dim3(db_grid(BN)), dim3(DB_NT), 0, (hipStream_t)stream, (const float2*)G, (const float2*)S, (const float2*)x,
                       (float2*)dx, (float2*)dS, (long long)C, (long long)N, BN, scale);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

#define DCB_BLOCKS 1024
__global__ __launch_bounds__(DB_NT) void k_dc_combine_bwd(const float2* __restrict__ dy, const float2* __restrict__ pred, const float2* __restrict__ ref, MrxMask m,
                                                          const float* __restrict__ dcw, float2* __restrict__ dpred, float2* __restrict__ deta,
                                                          double* __restrict__ part, int add_dy, long long C, long long H, long long W, long long total) {
    __shared__ double red[DB_NT / 64];
    const float w8 = dcw[0];
    double acc = 0.0;
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long long)gridDim.x * blockDim.x) {
        long long r = o;
        const long long w = r % W;
        r /= W;
        const long long h = r % H;
        r /= H;
        const long long c = r % C, b = r / C;
        const float2 g = dy[o];
        float2 d = make_float2(0.f, 0.f);
        if (mrx_mask_true(m, b, c, h, w)) {
            d = make_float2(g.x * w8, g.y * w8);
            if (part) {
                const float2 p = pred[o], q = ref[o];
                acc += (double)((p.x - q.x) * g.x) + (double)((p.y - q.y) * g.y);
            }
        }
        if (dpred) dpred[o] = add_dy ? make_float2(g.x - d.x, g.y - d.y) : make_float2(-d.x, -d.y);
        if (deta) deta[o] = make_float2(-g.x, -g.y);
    }
    if (part) {
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = red[0];
            for (int k = 1; k < DB_NT / 64; ++k) t += red[k];
            part[blockIdx.x] = t;
        }
    }
}
// one wave: lane l adds partials l, l + 64, ..., then a fixed butterfly (a single thread walking 1024 dependent loads took 60 us)
__global__ void k_dcw_final(const double* __restrict__ part, int n, float* __restrict__ dw) {
    double t = 0.0;
    for (int i = threadIdx.x; i < n; i += 64) t += part[i];
    for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
    if (threadIdx.x == 0) dw[0] = (float)(-t);
}
extern "C" int64_t mrx_dc_combine_bwd_work_doubles(void) { return DCB_BLOCKS; }
// dpred / deta / dw may be null (not needed); add_dy: base and pred are the same tensor (its gradient = dy - where(mask, dy, 0) * w); dw: one float;
// work: mrx_dc_combine_bwd_work_doubles doubles (only with dw)
extern "C" int mrx_dc_combine_bwd(const float* dy, const float* pred, const float* ref, const void* mask, int mask_kind, const int64_t* mstride,
                                  const float* dc_weight, float* dpred, float* deta, float* dw, double* work, int add_dy, int B, int C, int H, int W,
                                  void* stream) {
    MRX_REQUIRE(dy && dc_weight && mask && mstride && (dpred || deta || dw), MRX_EINVAL, "mrx_dc_combine_bwd: null pointer");
    MRX_REQUIRE(!dw || (pred && ref && work), MRX_EINVAL, "mrx_dc_combine_bwd: dw needs pred, ref and work");
    MRX_REQUIRE(mask_kind == MRX_MASK_U8 || mask_kind == MRX_MASK_F32, MRX_EINVAL, "mrx_dc_combine_bwd: bad mask kind %d", mask_kind);
    MRX_REQUIRE(B >= 0 && C >= 0 && H >= 0 && W >= 0, MRX_EINVAL, "mrx_dc_combine_bwd: negative dim");
    MrxMask m;
    m.p = mask, m.kind = mask_kind;
    for (int i = 0; i < 4; ++i) m.s[i] = mstride[i];
    const long long total = (long long)B * C * H * W;
    hipStream_t st = (hipStream_t)stream;
    if (total == 0) {
        if (dw) MRX_HIP(hipMemsetAsync(dw, 0, sizeof(float), st));
        return MRX_OK;
    }
    long long nb = (total + DB_NT - 1) / DB_NT;
    if (nb > DCB_BLOCKS) nb = DCB_BLOCKS;
    hipLaunchKernelGGL(k_dc_combine_bwd, dim3((unsigned)nb), dim3(DB_NT), 0, st, (const float2*)dy, (const float2*)pred, (const float2*)ref, m, dc_weight,
                       (float2*)dpred, (float2*)deta, dw ? work : (double*)nullptr, add_dy, (long long)C, (long long)H, (long long)W, total);
    if (dw) hipLaunchKernelGGL(k_dcw_final, dim3(1), dim3(64), 0, st, (const double*)work, (int)nb, dw);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
