// unet.hip -- NormUnet support kernels (reference models/unet_base/unet_block.py): group-norm statistics with the
// unbiased std (:71-91), zero / reflect / crop padding (:93-111, :215-222), InstanceNorm2d + LeakyReLU (:252-253),
// 2x2 average pooling (:206), ConvTranspose2d(k=2, s=2) (:293), channel-block copy for the skip concat (:224).
// The 3x3 / 1x1 convolutions themselves go through conv.hip.
#include <cstdint>

#include "mrx_common.h"

#define UN_NT 256

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
// block-wide sum, result broadcast to every thread (red: 4-float LDS scratch + 1)
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < UN_NT / 64; ++i) t += red[i];
    return t;
}

// ---- plane statistics, split over many workgroups (a 640x380 plane is 243k floats; one workgroup per plane would leave
// most of the chip idle).  Three deterministic passes, no atomics: partial sums -> partial squared deviations -> apply.
// Workspace: 2 * planes * nsplit floats owned by the caller (mrx_norm_work_floats).
static inline int un_nsplit(long long n) { return mrx_norm_nsplit(n); }      // (mrx_common.h: shared with the backward in diff_bwd.hip)
extern "C" int64_t mrx_norm_work_floats(int64_t planes, int64_t n) { return planes < 0 || n < 1 ? -1 : 2 * planes * un_nsplit(n); }

__device__ __forceinline__ void split_range(long long n, int nsplit, int s, long long& a, long long& b) {
    const long long per = (n + nsplit - 1) / nsplit;
    a = (long long)s * per;
    b = a + per < n ? a + per : n;
}
__global__ __launch_bounds__(UN_NT) void k_plane_sum(const float* x, float* part, long long n, int nsplit) {
    __shared__ float red[UN_NT / 64];
    long long a, b;
    split_range(n, nsplit, blockIdx.y, a, b);
    const float* p = x + (long long)blockIdx.x * n;
    float s = 0.f;
    for (long long i = a + threadIdx.x; i < b; i += UN_NT) s += p[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) part[(long long)blockIdx.x * nsplit + blockIdx.y] = s;
}
__device__ __forceinline__ float combine(const float* part, int nsplit) {
    double t = 0.0;
    for (int i = 0; i < nsplit; ++i) t += (double)part[i];
    return (float)t;
}
__global__ __launch_bounds__(UN_NT) void k_plane_sqdev(const float* x, const float* psum, float* psq, long long n, int nsplit) {
    __shared__ float red[UN_NT / 64];
    long long a, b;
    split_range(n, nsplit, blockIdx.y, a, b);
    const float* p = x + (long long)blockIdx.x * n;
    const float mean = combine(psum + (long long)blockIdx.x * nsplit, nsplit) / (float)n;
    float v = 0.f;
    for (long long i = a + threadIdx.x; i < b; i += UN_NT) {
        const float d = p[i] - mean;
        v += d * d;
    }
    v = block_sum(v, red);
    if (threadIdx.x == 0) psq[(long long)blockIdx.x * nsplit + blockIdx.y] = v;
}
// InstanceNorm2d (biased variance, eps) + activation   (unet_block.py:252-253, :294-295)
__global__ __launch_bounds__(UN_NT) void k_plane_norm_act(const float* x, float* out, const float* psum, const float* psq, long long n,
                                                          int nsplit, float eps, int act, float slope) {
    long long a, b;
    split_range(n, nsplit, blockIdx.y, a, b);
    const float* p = x + (long long)blockIdx.x * n;
    float* q = out + (long long)blockIdx.x * n;
    const float mean = combine(psum + (long long)blockIdx.x * nsplit, nsplit) / (float)n;
    const float var = combine(psq + (long long)blockIdx.x * nsplit, nsplit) / (float)n;
    const float inv = 1.0f / sqrtf(var + eps);
    for (long long i = a + threadIdx.x; i < b; i += UN_NT) {
        float y = (p[i] - mean) * inv;
        if (act == MRX_ACT_RELU)
            y = y > 0.f ? y : 0.f;
        else if (act == MRX_ACT_LEAKY)
            y = y > 0.f ? y : y * slope;
        q[i] = y;
    }
}
extern "C" int mrx_instance_norm_act(const float* x, float* out, float* work, int64_t planes, int64_t HW, float eps, int act,
                                     float slope, void* stream) {
    MRX_REQUIRE(x && out && work && planes >= 0 && HW >= 1, MRX_EINVAL, "mrx_instance_norm_act: bad argument");
    if (planes == 0) return MRX_OK;
    MRX_REQUIRE(planes < (1LL << 31), MRX_EUNSUP, "mrx_instance_norm_act: too many planes");
    const int ns = un_nsplit(HW);
    float* psum = work;
    float* psq = work + planes * ns;
    dim3 grid((unsigned)planes, ns);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_plane_sum, grid, dim3(UN_NT), 0, st, x, psum, (long long)HW, ns);
    hipLaunchKernelGGL(k_plane_sqdev, grid, dim3(UN_NT), 0, st, x, (const float*)psum, psq, (long long)HW, ns);
    hipLaunchKernelGGL(k_plane_norm_act, grid, dim3(UN_NT), 0, st, x, out, (const float*)psum, (const float*)psq, (long long)HW, ns,
                       eps, act, slope);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// InstanceNorm2d + activation from ready per-plane statistics (mean, sum of squared deviations) -- the second half of
// mrx_instance_norm_act for producers that computed the statistics themselves (mrx_conv2d_stats)
__device__ __forceinline__ float un_act(float y, int act, float slope) {
    if (act == MRX_ACT_RELU) return y > 0.f ? y : 0.f;
    if (act == MRX_ACT_LEAKY) return y > 0.f ? y : y * slope;
    return y;
}
// VEC: planes are whole float4s and 16-byte aligned (n % 4 == 0, aligned bases): one 16-byte access per lane
template <bool VEC>
__global__ __launch_bounds__(UN_NT) void k_plane_norm_apply(const float* x, float* out, const float* stats, long long n, int nsplit, float eps,
                                                            int act, float slope) {
    const float* p = x + (long long)blockIdx.x * n;
    float* q = out + (long long)blockIdx.x * n;
    const float mean = stats[(long long)blockIdx.x * 2];
    const float var = stats[(long long)blockIdx.x * 2 + 1] / (float)n;
    const float inv = 1.0f / sqrtf(var + eps);
    if (VEC) {
        long long a, b;
        split_range(n >> 2, nsplit, blockIdx.y, a, b);
        const float4* p4 = reinterpret_cast<const float4*>(p);
        float4* q4 = reinterpret_cast<float4*>(q);
        for (long long i = a + threadIdx.x; i < b; i += UN_NT) {
            float4 v = p4[i];
            v.x = un_act((v.x - mean) * inv, act, slope);
            v.y = un_act((v.y - mean) * inv, act, slope);
            v.z = un_act((v.z - mean) * inv, act, slope);
            v.w = un_act((v.w - mean) * inv, act, slope);
            q4[i] = v;
        }
    } else {
        long long a, b;
        split_range(n, nsplit, blockIdx.y, a, b);
        for (long long i = a + threadIdx.x; i < b; i += UN_NT) q[i] = un_act((p[i] - mean) * inv, act, slope);
    }
}
// the same pass fed by the per-tile statistics of mrx_conv2d_stats directly: every workgroup first merges the tiles of its plane
// (one wave, parallel-variance update in double -- the arithmetic of k_conv_stats_finalize), which costs ~1 us per workgroup and
// saves a dependent launch per convolution
template <bool VEC>
__global__ __launch_bounds__(UN_NT) void k_plane_norm_apply_tiles(const float* x, float* out, const float* tstats, int Cc, int H, int W,
                                                                  int nsplit, float eps, int act, float slope) {
    __shared__ float s_stat[2];
    const long long n = (long long)H * W;
    const int plane_id = blockIdx.x, b = plane_id / Cc, co = plane_id - b * Cc;
    if (threadIdx.x < 64) {
        const int tiles_x = (W + MRX_CONV_TILE_W - 1) / MRX_CONV_TILE_W, ntiles = tiles_x * ((H + MRX_CONV_TILE_H - 1) / MRX_CONV_TILE_H);
        double cnt = 0.0, mean = 0.0, m2 = 0.0;
        for (int t = threadIdx.x; t < ntiles; t += 64) {
            const int ty = t / tiles_x, tx = t - ty * tiles_x;
            const int nr = H - ty * MRX_CONV_TILE_H < MRX_CONV_TILE_H ? H - ty * MRX_CONV_TILE_H : MRX_CONV_TILE_H;
            const int nc = W - tx * MRX_CONV_TILE_W < MRX_CONV_TILE_W ? W - tx * MRX_CONV_TILE_W : MRX_CONV_TILE_W;
            const double nb = (double)(nr * nc);
            const float* p = tstats + (((long long)b * ntiles + t) * Cc + co) * 2;
            const double mb = (double)p[0], qb = (double)p[1];
            const double tot = cnt + nb, delta = mb - mean;
            mean += delta * nb / tot;
            m2 += qb + delta * delta * cnt * nb / tot;
            cnt = tot;
        }
        for (int off = 32; off > 0; off >>= 1) {
            const double nb = __shfl_xor(cnt, off, 64), mb = __shfl_xor(mean, off, 64), qb = __shfl_xor(m2, off, 64);
            const double tot = cnt + nb;
            if (tot > 0.0) {
                const double delta = mb - mean;
                mean += delta * nb / tot;
                m2 += qb + delta * delta * cnt * nb / tot;
            }
            cnt = tot;
        }
        if (threadIdx.x == 0) {
            s_stat[0] = (float)mean;
            s_stat[1] = (float)m2;
        }
    }
    __syncthreads();
    const float mean = s_stat[0];
    const float inv = 1.0f / sqrtf(s_stat[1] / (float)n + eps);
    const float* p = x + (long long)plane_id * n;
    float* q = out + (long long)plane_id * n;
    if (VEC) {
        long long a, e;
        split_range(n >> 2, nsplit, blockIdx.y, a, e);
        const float4* p4 = reinterpret_cast<const float4*>(p);
        float4* q4 = reinterpret_cast<float4*>(q);
        for (long long i = a + threadIdx.x; i < e; i += UN_NT) {
            float4 v = p4[i];
            v.x = un_act((v.x - mean) * inv, act, slope);
            v.y = un_act((v.y - mean) * inv, act, slope);
            v.z = un_act((v.z - mean) * inv, act, slope);
            v.w = un_act((v.w - mean) * inv, act, slope);
            q4[i] = v;
        }
    } else {
        long long a, e;
        split_range(n, nsplit, blockIdx.y, a, e);
        for (long long i = a + threadIdx.x; i < e; i += UN_NT) q[i] = un_act((p[i] - mean) * inv, act, slope);
    }
}
extern "C" int mrx_instance_norm_apply_tiles(const float* x, float* out, const float* tile_stats, int B, int Cc, int H, int W, float eps,
                                             int act, float slope, void* stream) {
    MRX_REQUIRE(x && out && tile_stats && B >= 0 && Cc >= 1 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_instance_norm_apply_tiles: bad argument");
    if (B == 0) return MRX_OK;
    const long long HW = (long long)H * W, planes = (long long)B * Cc;
    MRX_REQUIRE(planes < (1LL << 31), MRX_EUNSUP, "mrx_instance_norm_apply_tiles: too many planes");
    const int ns = un_nsplit(HW);
    if ((HW & 3) == 0 && ((((uintptr_t)x) | ((uintptr_t)out)) & 15) == 0)
        hipLaunchKernelGGL(k_plane_norm_apply_tiles<true>, dim3((unsigned)planes, ns), dim3(UN_NT), 0, (hipStream_t)stream, x, out, tile_stats,
                           Cc, H, W, ns, eps, act, slope);
    else
        hipLaunchKernelGGL(k_plane_norm_apply_tiles<false>, dim3((unsigned)planes, ns), dim3(UN_NT), 0, (hipStream_t)stream, x, out, tile_stats,
                           Cc, H, W, ns, eps, act, slope);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
extern "C" int mrx_instance_norm_apply(const float* x, float* out, const float* stats, int64_t planes, int64_t HW, float eps, int act,
                                       float slope, void* stream) {
    MRX_REQUIRE(x && out && stats && planes >= 0 && HW >= 1, MRX_EINVAL, "mrx_instance_norm_apply: bad argument");
    if (planes == 0) return MRX_OK;
    MRX_REQUIRE(planes < (1LL << 31), MRX_EUNSUP, "mrx_instance_norm_apply: too many planes");
    const int ns = un_nsplit(HW);
    if ((HW & 3) == 0 && ((((uintptr_t)x) | ((uintptr_t)out)) & 15) == 0)
        hipLaunchKernelGGL(k_plane_norm_apply<true>, dim3((unsigned)planes, ns), dim3(UN_NT), 0, (hipStream_t)stream, x, out, stats,
                           (long long)HW, ns, eps, act, slope);
    else
        hipLaunchKernelGGL(k_plane_norm_apply<false>, dim3((unsigned)planes, ns), dim3(UN_NT), 0, (hipStream_t)stream, x, out, stats,
                           (long long)HW, ns, eps, act, slope);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- group norm statistics: mean and UNBIASED std per group (unet_block.py:78-79) ---------------------------------------
__global__ void k_group_finalize(const float* psum, const float* psq, float* mean_o, float* std_o, long long groups, long long n,
                                 int nsplit) {
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= groups) return;
    mean_o[g] = combine(psum + g * nsplit, nsplit) / (float)n;
    std_o[g] = sqrtf(combine(psq + g * nsplit, nsplit) / (float)(n - 1));
}
extern "C" int mrx_group_norm_stats(const float* x, float* mean, float* std_, float* work, int64_t groups, int64_t n, void* stream) {
    MRX_REQUIRE(x && mean && std_ && work && groups >= 0 && n >= 1, MRX_EINVAL, "mrx_group_norm_stats: bad argument");
    if (groups == 0) return MRX_OK;
    const int ns = un_nsplit(n);
    float* psum = work;
    float* psq = work + groups * ns;
    dim3 grid((unsigned)groups, ns);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_plane_sum, grid, dim3(UN_NT), 0, st, x, psum, (long long)n, ns);
    hipLaunchKernelGGL(k_plane_sqdev, grid, dim3(UN_NT), 0, st, x, (const float*)psum, psq, (long long)n, ns);
    hipLaunchKernelGGL(k_group_finalize, dim3((unsigned)((groups + 63) / 64)), dim3(64), 0, st, (const float*)psum, (const float*)psq,
                       mean, std_, (long long)groups, (long long)n, ns);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
// inverse == 0: (x - mean) / std  (:81) ; inverse == 1: x * std + mean  (:91)
__global__ void k_group_apply(const float* x, const float* mean, const float* std_, float* out, long long n, long long total,
                              int inverse) {
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long long)gridDim.x * blockDim.x) {
        const long long g = o / n;
        const float m = mean[g], s = std_[g];
        out[o] = inverse ? x[o] * s + m : (x[o] - m) / s;
    }
}
static inline int un_grid(long long n) {
    long long g = (n + UN_NT - 1) / UN_NT;
    if (g > 4096) g = 4096;
    return g < 1 ? 1 : (int)g;
}
extern "C" int mrx_group_norm_apply(const float* x, const float* mean, const float* std_, float* out, int64_t groups, int64_t n,
                                    int inverse, void* stream) {
    MRX_REQUIRE(x && mean && std_ && out && groups >= 0 && n >= 1, MRX_EINVAL, "mrx_group_norm_apply: bad argument");
    const long long total = groups * n;
    if (total == 0) return MRX_OK;
    hipLaunchKernelGGL(k_group_apply, dim3(un_grid(total)), dim3(UN_NT), 0, (hipStream_t)stream, x, mean, std_, out, (long long)n,
                       total, inverse);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- group norm backward (training E2EVN: the derivative of unet_block.py:71-91, which the reference leaves to autograd) ----------------------------
// With xhat = (x - mean) / std (unbiased std over the group's n values) and the gradients dy (of xhat), dmean, dstd (of the two statistics, which the
// un-normalisation at the end of NormUnet uses):
//   dx = (dy - S1 / n) / std - xhat S2 / (std (n - 1)) + dmean / n + dstd xhat / (n - 1),     S1 = sum dy, S2 = sum dy xhat.
// The un-normalisation y = x std + mean:   dx = dy std,   dstd = sum dy x,   dmean = sum dy -- the same two sums with x in place of xhat.
// Two deterministic passes like the forward: partial sums per (group, split), then the apply pass (which combines them in double).
__global__ __launch_bounds__(UN_NT) void k_group_bwd_sums(const float* __restrict__ dy, const float* __restrict__ v, float* __restrict__ p1, float* __restrict__ p2,
                                                          long long n, int nsplit) {
    __shared__ float red[UN_NT / 64];
    long long a, b;
    split_range(n, nsplit, blockIdx.y, a, b);
    const float* d = dy + (long long)blockIdx.x * n;
    const float* q = v + (long long)blockIdx.x * n;
    float s1 = 0.f, s2 = 0.f;
    for (long long i = a + threadIdx.x; i < b; i += UN_NT) {
        const float di = d[i];
        s1 += di;
        s2 += di * q[i];
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) {
        p1[(long long)blockIdx.x * nsplit + blockIdx.y] = s1;
        p2[(long long)blockIdx.x * nsplit + blockIdx.y] = s2;
    }
}
// inverse == 0: dx of the normalisation (v = xhat; dmean / dstd may be null = 0); inverse == 1: dx = dy std, and (split 0) dmean_o = S1, dstd_o = S2
__global__ __launch_bounds__(UN_NT) void k_group_bwd_apply(const float* __restrict__ dy, const float* __restrict__ v, const float* __restrict__ std_,
                                                           const float* __restrict__ p1, const float* __restrict__ p2, const float* __restrict__ dmean,
                                                           const float* __restrict__ dstd, float* __restrict__ dx, float* __restrict__ dmean_o,
                                                           float* __restrict__ dstd_o, long long n, int nsplit, int inverse) {
    long long a, b;
    split_range(n, nsplit, blockIdx.y, a, b);
    const long long g = blockIdx.x;
    const float S1 = combine(p1 + g * nsplit, nsplit), S2 = combine(p2 + g * nsplit, nsplit), sd = std_[g];
    const float* d = dy + g * n;
    float* o = dx + g * n;
    if (inverse) {
        for (long long i = a + threadIdx.x; i < b; i += UN_NT) o[i] = d[i] * sd;
        if (blockIdx.y == 0 && threadIdx.x == 0) dmean_o[g] = S1, dstd_o[g] = S2;
        return;
    }
    const float* q = v + g * n;
    const float inv = 1.0f / sd, c0 = ((dmean ? dmean[g] : 0.f) - S1 * inv) / (float)n, c1 = ((dstd ? dstd[g] : 0.f) - S2 * inv) / (float)(n - 1);
    for (long long i = a + threadIdx.x; i < b; i += UN_NT) o[i] = d[i] * inv + c0 + c1 * q[i];
}
// work: mrx_norm_work_floats(groups, n) floats.  inverse = 0: dx [groups, n] from dy, xhat, std and (optional) dmean, dstd [groups]; dmean_o / dstd_o unused.
// inverse = 1: dx = dy std and dmean_o, dstd_o [groups] from dy and x (the un-normalisation's input); dmean / dstd unused.
extern "C" int mrx_group_norm_bwd(const float* dy, const float* v, const float* std_, const float* dmean, const float* dstd, float* dx, float* dmean_o,
                                  float* dstd_o, float* work, int64_t groups, int64_t n, int inverse, void* stream) {
    MRX_REQUIRE(dy && v && std_ && dx && work && groups >= 0 && n >= 2, MRX_EINVAL, "mrx_group_norm_bwd: bad argument");
    MRX_REQUIRE(!inverse || (dmean_o && dstd_o), MRX_EINVAL, "mrx_group_norm_bwd: the un-normalisation's backward returns dmean and dstd");
    if (groups == 0) return MRX_OK;
    MRX_REQUIRE(groups < (1LL << 31), MRX_EUNSUP, "mrx_group_norm_bwd: too many groups");
    const int ns = un_nsplit(n);
    float* p1 = work;
    float* p2 = work + groups * ns;
    dim3 grid((unsigned)groups, ns);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_group_bwd_sums, grid, dim3(UN_NT), 0, st, dy, v, p1, p2, (long long)n, ns);
    hipLaunchKernelGGL(k_group_bwd_apply, grid, dim3(UN_NT), 0, st, dy, v, std_, (const float*)p1, (const float*)p2, dmean, dstd, dx, dmean_o, dstd_o,
                       (long long)n, ns, inverse);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- pad / crop: out[y][x] = in[y - top][x - left]; outside: 0 (mode 0) or reflect (mode 1).  Negative pads crop. --------
__global__ void k_pad2d(const float* in, float* out, long long planes, int H, int W, int top, int left, int OH, int OW, int mode) {
    const long long total = planes * OH * OW;
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(o % OW);
        const long long r = o / OW;
        const int oy = (int)(r % OH);
        const long long p = r / OH;
        int iy = oy - top, ix = ox - left;
        float v = 0.f;
        if (mode == 1) {
            iy = iy < 0 ? -iy : (iy >= H ? 2 * (H - 1) - iy : iy);
            ix = ix < 0 ? -ix : (ix >= W ? 2 * (W - 1) - ix : ix);
            v = in[(p * H + iy) * W + ix];
        } else if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
            v = in[(p * H + iy) * W + ix];
        }
        out[o] = v;
    }
}
extern "C" int mrx_pad2d(const float* in, float* out, int64_t planes, int H, int W, int top, int bottom, int left, int right,
                         int mode, void* stream) {
    MRX_REQUIRE(in && out && planes >= 0 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_pad2d: bad argument");
    const int OH = H + top + bottom, OW = W + left + right;
    MRX_REQUIRE(OH >= 1 && OW >= 1, MRX_EINVAL, "mrx_pad2d: empty output");
    MRX_REQUIRE(mode == 0 || (top < H && bottom < H && left < W && right < W), MRX_EINVAL, "mrx_pad2d: reflect pad too large");
    if (planes == 0) return MRX_OK;
    hipLaunchKernelGGL(k_pad2d, dim3(un_grid(planes * OH * OW)), dim3(UN_NT), 0, (hipStream_t)stream, in, out, (long long)planes, H,
                       W, top, left, OH, OW, mode);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- avg_pool2d(kernel 2, stride 2, no padding): floor sizes (unet_block.py:206) ------------------------------------------
__global__ void k_avgpool(const float* in, float* out, long long planes, int H, int W, int OH, int OW) {
    const long long total = planes * OH * OW;
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(o % OW);
        const long long r = o / OW;
        const int oy = (int)(r % OH);
        const long long p = r / OH;
        const float* q = in + (p * H + 2 * oy) * W + 2 * ox;
        out[o] = (q[0] + q[1] + q[W] + q[W + 1]) * 0.25f;
    }
}
extern "C" int mrx_avg_pool2x2(const float* in, float* out, int64_t planes, int H, int W, void* stream) {
    MRX_REQUIRE(in && out && planes >= 0 && H >= 2 && W >= 2, MRX_EINVAL, "mrx_avg_pool2x2: bad argument");
    if (planes == 0) return MRX_OK;
    const int OH = H / 2, OW = W / 2;
    hipLaunchKernelGGL(k_avgpool, dim3(un_grid(planes * OH * OW)), dim3(UN_NT), 0, (hipStream_t)stream, in, out, (long long)planes,
                       H, W, OH, OW);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- ConvTranspose2d(kernel 2, stride 2, no bias): out[b,co,2y+dy,2x+dx] = sum_ci x[b,ci,y,x] * w[ci,co,dy,dx] --------------
__global__ void k_convT2x2(const float* x, const float* w, float* out, int B, int Cin, int Cout, int H, int W) {
    const int OH = 2 * H, OW = 2 * W;
    const long long total = (long long)B * Cout * OH * OW;
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(o % OW);
        long long r = o / OW;
        const int oy = (int)(r % OH);
        r /= OH;
        const int co = (int)(r % Cout);
        const int b = (int)(r / Cout);
        const int y = oy >> 1, dy = oy & 1, xx = ox >> 1, dx = ox & 1;
        const float* xp = x + ((long long)b * Cin * H + y) * W + xx;
        const float* wp = w + ((long long)co * 2 + dy) * 2 + dx;
        float acc = 0.f;
        for (int ci = 0; ci < Cin; ++ci) acc += xp[(long long)ci * H * W] * wp[(long long)ci * Cout * 4];
        out[o] = acc;
    }
}
// Tuned form: one thread per INPUT pixel and group of COG output channels (4 COG accumulators); the input plane is read once per
// channel group (coalesced), the weights sit in LDS as float4 (dy, dx) quads read by broadcast, every output row gets float2 stores
// that are contiguous across the wave.  The generic kernel above re-reads every input Cout*4 times.
typedef float ct_f2 __attribute__((ext_vector_type(2)));
typedef float ct_f4 __attribute__((ext_vector_type(4)));
// STATS: the InstanceNorm statistics of the output come out of the same accumulators -- per workgroup (256 input pixels = 1024 outputs of
// one plane) and cout the mean, then the sum of squared deviations from it (two fixed-order block reductions) into
// tstats[b][tile][cout][2]; k_tile_stats_finalize merges the tiles (unet_block.py:296-299: ConvTranspose2d -> InstanceNorm2d -> LeakyReLU).
template <int COG, bool STATS>
__global__ __launch_bounds__(UN_NT) void k_convT2x2_t(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ out, int Cin,
                                                      int Cout, int H, int W, float* __restrict__ tstats) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];  // [Cin][COG] quads of this channel group
    const int g0 = blockIdx.y * COG, b = blockIdx.z;
    for (int i = threadIdx.x; i < Cin * COG * 4; i += UN_NT) {
        const int q = i & 3, co = (i >> 2) % COG, ci = (i >> 2) / COG;
        wsm[i] = w[((long long)ci * Cout + g0 + co) * 4 + q];
    }
    __syncthreads();
    const long long HW = (long long)H * W;
    const long long pix_raw = (long long)blockIdx.x * UN_NT + threadIdx.x;
    const bool live = pix_raw < HW;
    if (!STATS && !live) return;
    const long long pix = live ? pix_raw : HW - 1;         // (STATS: idle threads stay for the block reductions)
    const int y = (int)(pix / W), xx = (int)(pix - (long long)y * W);
    const float* xp = x + (long long)b * Cin * HW + pix;
    ct_f4 acc[COG];
#pragma unroll
    for (int co = 0; co < COG; ++co) acc[co] = (ct_f4){0.f, 0.f, 0.f, 0.f};
    int ci = 0;
    for (; ci + 4 <= Cin; ci += 4) {                    // four input planes in flight per step (the loads are the latency of this kernel)
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = xp[(long long)(ci + u) * HW];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const ct_f4* wq = reinterpret_cast<const ct_f4*>(wsm) + (ci + u) * COG;
#pragma unroll
            for (int co = 0; co < COG; ++co) acc[co] += v[u] * wq[co];  // same accumulation order over ci as the generic kernel
        }
    }
    for (; ci < Cin; ++ci) {
        const float v = xp[(long long)ci * HW];
        const ct_f4* wq = reinterpret_cast<const ct_f4*>(wsm) + ci * COG;
#pragma unroll
        for (int co = 0; co < COG; ++co) acc[co] += v * wq[co];
    }
    const int OW = 2 * W;
    float* op = out + (((long long)b * Cout + g0) * 2 * H + 2 * y) * OW + 2 * xx;
    if (live) {
#pragma unroll
        for (int co = 0; co < COG; ++co) {
            float* o = op + (long long)co * 4 * HW;
            *reinterpret_cast<ct_f2*>(o) = (ct_f2){acc[co][0], acc[co][1]};
            *reinterpret_cast<ct_f2*>(o + OW) = (ct_f2){acc[co][2], acc[co][3]};
        }
    }
    if (STATS) {
        __shared__ float red[UN_NT / 64];
        const long long rem = HW - (long long)blockIdx.x * UN_NT;
        const float inv_n = 1.0f / (4.0f * (float)(rem < UN_NT ? rem : UN_NT));
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
        for (int co = 0; co < COG; ++co) {
            float mean = 0.f;
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                float t = 0.f;
                if (live) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) t += pass == 0 ? acc[co][q] : (acc[co][q] - mean) * (acc[co][q] - mean);
                }
                for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
                __syncthreads();                   // the previous round's readers are done with `red`
                if (lane == 0) red[wv] = t;
                __syncthreads();
                const float tot = (red[0] + red[1]) + (red[2] + red[3]);
                if (pass == 0)
                    mean = tot * inv_n;
                else if (threadIdx.x == 0) {
                    float* ts = tstats + (((long long)b * gridDim.x + blockIdx.x) * Cout + g0 + co) * 2;
                    ts[0] = mean;
                    ts[1] = tot;
                }
            }
        }
    }
}
// merge per-tile (mean, M2) into per-plane (mean, M2): one wave per (b, cout), Chan et al. pairwise updates in double; every tile holds
// n_tile values except the last (n_last)
__global__ __launch_bounds__(64) void k_tile_stats_finalize(const float* __restrict__ tstats, float* __restrict__ stats, int ntiles, int Cout,
                                                           double n_tile, double n_last) {
    const int plane_id = blockIdx.x, b = plane_id / Cout, co = plane_id - b * Cout;
    double n = 0.0, mean = 0.0, m2 = 0.0;
    for (int t = threadIdx.x; t < ntiles; t += 64) {
        const double nb = t == ntiles - 1 ? n_last : n_tile;
        const float* p = tstats + (((long long)b * ntiles + t) * Cout + co) * 2;
        const double mb = (double)p[0], qb = (double)p[1];
        const double tot = n + nb, delta = mb - mean;
        mean += delta * nb / tot;
        m2 += qb + delta * delta * n * nb / tot;
        n = tot;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const double nb = __shfl_xor(n, off, 64), mb = __shfl_xor(mean, off, 64), qb = __shfl_xor(m2, off, 64);
        const double tot = n + nb;
        if (tot > 0.0) {
            const double delta = mb - mean;
            mean += delta * nb / tot;
            m2 += qb + delta * delta * n * nb / tot;
        }
        n = tot;
    }
    if (threadIdx.x == 0) {
        stats[(long long)plane_id * 2] = (float)mean;
        stats[(long long)plane_id * 2 + 1] = (float)m2;
    }
}
template <int COG>
static int launch_convT2x2_t(const float* x, const float* w, float* out, int B, int Cin, int Cout, int H, int W, hipStream_t st,
                             float* tstats = nullptr) {
    const size_t lds = sizeof(float) * (size_t)Cin * COG * 4;
    const dim3 grid((unsigned)(((long long)H * W + UN_NT - 1) / UN_NT), Cout / COG, B);
    if (tstats)
        hipLaunchKernelGGL((k_convT2x2_t<COG, true>), grid, dim3(UN_NT), lds, st, x, w, out, Cin, Cout, H, W, tstats);
    else
        hipLaunchKernelGGL((k_convT2x2_t<COG, false>), grid, dim3(UN_NT), lds, st, x, w, out, Cin, Cout, H, W, tstats);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
static bool convT_tuned(const float* out, int B, int Cin, int Cout) {
    return B <= 65535 && Cout <= 65535 * 8 && (size_t)Cin * 14 * 16 <= 48 * 1024 && (((uintptr_t)out) & 7) == 0;
}
// the transposed convolution + the InstanceNorm statistics (mean, sum of squared deviations) [B, Cout, 2] of its output in one pass;
// tuned shapes only (even Cout), MRX_EUNSUP otherwise so the caller takes mrx_conv_transpose2x2 + mrx_instance_norm_act
extern "C" int64_t mrx_conv_transpose2x2_stats_work_floats(int B, int Cout, int H, int W) {
    if (B < 0 || Cout < 1 || H < 1 || W < 1) return -1;
    return (int64_t)B * (((long long)H * W + UN_NT - 1) / UN_NT) * Cout * 2;
}
extern "C" int mrx_conv_transpose2x2_stats(const float* x, const float* w, float* out, float* stats, float* work, int B, int Cin, int Cout,
                                           int H, int W, void* stream) {
    MRX_REQUIRE(x && w && out && stats && work && B >= 0 && Cin >= 1 && Cout >= 1 && H >= 1 && W >= 1, MRX_EINVAL,
                "mrx_conv_transpose2x2_stats: bad argument");
    MRX_REQUIRE(convT_tuned(out, B, Cin, Cout) && Cout % 2 == 0, MRX_EUNSUP, "mrx_conv_transpose2x2_stats: Cout=%d Cin=%d", Cout, Cin);
    if (B == 0) return MRX_OK;
    hipStream_t st = (hipStream_t)stream;
    const long long HW = (long long)H * W, ntiles = (HW + UN_NT - 1) / UN_NT;
    int rc;
    if (Cout % 14 == 0 && ntiles * B * (Cout / 14) >= 2048) rc = launch_convT2x2_t<14>(x, w, out, B, Cin, Cout, H, W, st, work);
    else if (Cout % 8 == 0 && ntiles * B * (Cout / 8) >= 2048) rc = launch_convT2x2_t<8>(x, w, out, B, Cin, Cout, H, W, st, work);
    else rc = launch_convT2x2_t<2>(x, w, out, B, Cin, Cout, H, W, st, work);
    if (rc) return rc;
    hipLaunchKernelGGL(k_tile_stats_finalize, dim3(B * Cout), dim3(64), 0, st, (const float*)work, stats, (int)ntiles, Cout, 4.0 * UN_NT,
                       4.0 * (double)(HW - (ntiles - 1) * UN_NT));
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
extern "C" int mrx_conv_transpose2x2(const float* x, const float* w, float* out, int B, int Cin, int Cout, int H, int W,
                                     void* stream) {
    MRX_REQUIRE(x && w && out && B >= 0 && Cin >= 1 && Cout >= 1 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_conv_transpose2x2: bad argument");
    if (B == 0) return MRX_OK;
    const bool tuned = convT_tuned(out, B, Cin, Cout);
    // few pixels per channel group would leave most CUs idle (28 -> 14 channels at 320 x 192: 240 workgroups): narrower groups then
    const long long wgs = ((long long)H * W + UN_NT - 1) / UN_NT * B;
    if (tuned && Cout % 14 == 0 && wgs * (Cout / 14) >= 2048) return launch_convT2x2_t<14>(x, w, out, B, Cin, Cout, H, W, (hipStream_t)stream);
    if (tuned && Cout % 8 == 0 && wgs * (Cout / 8) >= 2048) return launch_convT2x2_t<8>(x, w, out, B, Cin, Cout, H, W, (hipStream_t)stream);
    if (tuned && Cout % 2 == 0) return launch_convT2x2_t<2>(x, w, out, B, Cin, Cout, H, W, (hipStream_t)stream);
    if (tuned && Cout % 14 == 0) return launch_convT2x2_t<14>(x, w, out, B, Cin, Cout, H, W, (hipStream_t)stream);
    if (tuned && Cout % 8 == 0) return launch_convT2x2_t<8>(x, w, out, B, Cin, Cout, H, W, (hipStream_t)stream);
    hipLaunchKernelGGL(k_convT2x2, dim3(un_grid((long long)B * Cout * 4 * H * W)), dim3(UN_NT), 0, (hipStream_t)stream, x, w, out, B,
                       Cin, Cout, H, W);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- copy a [B,C,HW] tensor into channels [c0, c0+C) of a [B,Ctot,HW] tensor (skip concat, unet_block.py:224) -------------
__global__ void k_copy_channels(const float* src, float* dst, int C, long long HW, int Ctot, int c0, long long total) {
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long long)gridDim.x * blockDim.x) {
        const long long per_b = (long long)C * HW;
        const long long b = o / per_b, r = o - b * per_b;
        dst[(b * Ctot + c0) * HW + r] = src[o];
    }
}
extern "C" int mrx_copy_channels(const float* src, float* dst, int B, int C, int64_t HW, int Ctot, int c0, void* stream) {
    MRX_REQUIRE(src && dst && B >= 0 && C >= 0 && HW >= 0 && c0 >= 0 && c0 + C <= Ctot, MRX_EINVAL, "mrx_copy_channels: bad argument");
    const long long total = (long long)B * C * HW;
    if (total == 0) return MRX_OK;
    hipLaunchKernelGGL(k_copy_channels, dim3(un_grid(total)), dim3(UN_NT), 0, (hipStream_t)stream, src, dst, C, (long long)HW, Ctot, c0,
                       total);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// torch.cat([a, b], dim=1) in one launch (unet_block.py:224): out[b] = (a[b] (Ca planes), b[b] (Cb planes))
__global__ void k_concat2(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long long na, long long nb,
                          long long total) {
    const long long per = na + nb;  // floats per batch element of the result
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long long)gridDim.x * blockDim.x) {
        const long long bi = o / per, r = o - bi * per;
        out[o] = r < na ? a[bi * na + r] : b[bi * nb + (r - na)];
    }
}
extern "C" int mrx_concat_channels(const float* a, const float* b, float* out, int B, int Ca, int Cb, int64_t HW, void* stream) {
    MRX_REQUIRE(a && b && out && B >= 0 && Ca >= 0 && Cb >= 0 && HW >= 0, MRX_EINVAL, "mrx_concat_channels: bad argument");
    const long long total = (long long)B * (Ca + Cb) * HW;
    if (total == 0) return MRX_OK;
    hipLaunchKernelGGL(k_concat2, dim3(un_grid(total)), dim3(UN_NT), 0, (hipStream_t)stream, a, b, out, (long long)Ca * HW,
                       (long long)Cb * HW, total);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- SSIM (reference common/losses/ssim.py:46-61): 7x7 uniform window (valid), cov_norm = NP/(NP-1), 1 - mean(S) ----------
// One thread per output pixel of a 16x16 tile, both images staged in LDS with their halo; per-workgroup partial sums of S
// are reduced by a second single-workgroup kernel in double (deterministic, no atomics).
#define SS_T 16
__global__ __launch_bounds__(SS_T* SS_T) void k_ssim_partial(const float* __restrict__ X, const float* __restrict__ Y,
                                                            const float* __restrict__ data_range, float* __restrict__ part,
                                                            int h, int w, int win, float k1, float k2, int tiles_x) {
    extern __shared__ float ssm[];
    const int P = SS_T + win - 1;
    float* xs = ssm;
    float* ys = ssm + P * P;
    __shared__ float red[UN_NT / 64];
    const int b = blockIdx.y;
    const int ty0 = (blockIdx.x / tiles_x) * SS_T, tx0 = (blockIdx.x % tiles_x) * SS_T;
    const float* xb = X + (long long)b * h * w;
    const float* yb = Y + (long long)b * h * w;
    for (int i = threadIdx.x; i < P * P; i += SS_T * SS_T) {
        const int py = i / P, px = i - py * P;
        const int gy = ty0 + py, gx = tx0 + px;
        const bool ok = gy < h && gx < w;
        xs[i] = ok ? xb[(long long)gy * w + gx] : 0.f;
        ys[i] = ok ? yb[(long long)gy * w + gx] : 0.f;
    }
    __syncthreads();
    const int ly = threadIdx.x / SS_T, lx = threadIdx.x % SS_T;
    const int oy = ty0 + ly, ox = tx0 + lx;
    const int oh = h - win + 1, ow = w - win + 1;
    float S = 0.f;
    if (oy < oh && ox < ow) {
        float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
        for (int dy = 0; dy < win; ++dy)
            for (int dx = 0; dx < win; ++dx) {
                const float xv = xs[(ly + dy) * P + lx + dx], yv = ys[(ly + dy) * P + lx + dx];
                sx += xv;
                sy += yv;
                sxx += xv * xv;
                syy += yv * yv;
                sxy += xv * yv;
            }
        const float np = (float)(win * win), wgt = 1.0f / np, cov = np / (np - 1.0f);
        const float ux = sx * wgt, uy = sy * wgt, uxx = sxx * wgt, uyy = syy * wgt, uxy = sxy * wgt;
        const float dr = data_range[b];
        const float C1 = (k1 * dr) * (k1 * dr), C2 = (k2 * dr) * (k2 * dr);
        const float vx = cov * (uxx - ux * ux), vy = cov * (uyy - uy * uy), vxy = cov * (uxy - ux * uy);
        const float A1 = 2 * ux * uy + C1, A2 = 2 * vxy + C2, B1 = ux * ux + uy * uy + C1, B2 = vx + vy + C2;
        S = (A1 * A2) / (B1 * B2);
    }
    S = block_sum(S, red);
    if (threadIdx.x == 0) part[(long long)b * gridDim.x + blockIdx.x] = S;
}
__global__ __launch_bounds__(UN_NT) void k_ssim_final(const float* part, float* out, long long nparts, double count) {
    __shared__ double dred[UN_NT];
    double t = 0.0;
    for (long long i = threadIdx.x; i < nparts; i += UN_NT) t += (double)part[i];
    dred[threadIdx.x] = t;
    __syncthreads();
    for (int o = UN_NT / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) dred[threadIdx.x] += dred[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(1.0 - dred[0] / count);
}
extern "C" int64_t mrx_ssim_work_floats(int B, int h, int w) {
    if (B < 0 || h < 1 || w < 1) return -1;
    return (int64_t)B * ((h + SS_T - 1) / SS_T) * ((w + SS_T - 1) / SS_T);
}
extern "C" int mrx_ssim_loss(const float* X, const float* Y, const float* data_range, float* out, float* work, int B, int h, int w,
                             int win, float k1, float k2, void* stream) {
    MRX_REQUIRE(X && Y && data_range && out && work, MRX_EINVAL, "mrx_ssim_loss: null pointer");
    MRX_REQUIRE(B >= 1 && win >= 1 && h >= win && w >= win && win <= 31, MRX_EINVAL, "mrx_ssim_loss: bad dims B=%d h=%d w=%d win=%d", B, h, w, win);
    const int tiles_x = (w + SS_T - 1) / SS_T, tiles_y = (h + SS_T - 1) / SS_T;
    const int P = SS_T + win - 1;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_ssim_partial, dim3(tiles_x * tiles_y, B), dim3(SS_T * SS_T), sizeof(float) * 2 * P * P, st, X, Y, data_range,
                       work, h, w, win, k1, k2, tiles_x);
    hipLaunchKernelGGL(k_ssim_final, dim3(1), dim3(UN_NT), 0, st, (const float*)work, out, (long long)B * tiles_x * tiles_y,
                       (double)B * (h - win + 1) * (w - win + 1));
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
