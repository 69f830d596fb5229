// unet_fused.hip -- the U-Net of the E2EVN / UNet models (reference unet_base/unet_block.py:139-308) without its normalisation passes.
//
// Every 3x3 convolution and every transposed convolution of that network is followed by InstanceNorm2d + LeakyReLU(0.2)
// (unet_block.py:251-258, 296-299).  The kernels here keep such a tensor as the pair (raw convolution output, per-plane (mean, 1/std)):
// the statistics come out of the producer's accumulators (per-tile mean and squared deviations, merged in double by a one-wave-per-plane
// finalize) and the normalisation + activation is applied by whoever READS the tensor, while it stages its tile:
//   * mrx_unet_conv3x3      3x3 convolution (zero padding, no bias) over the channels of up to TWO sources -- so the skip concatenation
//                           (unet_block.py:224) is never materialised -- each plain or (raw, norm); raw output + its norm
//   * mrx_unet_conv_transpose2x2     ConvTranspose2d(k 2, s 2, no bias) of a (raw, norm) input; raw output + its norm
//   * mrx_unet_avgpool      avg_pool2d(2) of a (raw, norm) input -> plain tensor
//   * mrx_unet_conv1x1      the closing 1x1 convolution (+ bias) of a (raw, norm) input -> plain tensor
//   * mrx_unet_apply        materialises a (raw, norm) tensor (fallback for the odd-size reflect pad, unet_block.py:215-222)
// Against the conv + apply formulation this removes one full read + write of every activation and one launch per layer; the convolution
// itself stages eight input channels per step through registers (loads of step q + 1 in flight under the matrix work of step q, one
// barrier per step) instead of four through a serialised LDS-DMA round trip.
// Arithmetic: fp32 MFMA (v_mfma_f32_16x16x4_f32: exact fp32 FMA chains, channel groups of four in ascending order, taps inside -- the
// order of the kernel it replaces), normalisation as (x - mean) * (1 / sqrtf(M2 / n + eps)) like mrx_instance_norm_apply.
#include <cstdint>
#include <cstdlib>

#include "mrx_common.h"

typedef float uc_f4 __attribute__((ext_vector_type(4)));
typedef float uc_f2 __attribute__((ext_vector_type(2)));

#define UC_NT 256
#define UC_TH 8
#define UC_TW 32
#define UC_PW 34                 // halo'd tile: 10 rows x 34 columns
#define UC_PIX (10 * UC_PW)      // 340
#define UC_PLANE 368             // 368 = 48 mod 64: the four channel planes of an MFMA step land on different bank groups
#define UC_CK 8                  // input channels per step
#define UC_XL 11                 // tile elements per lane and step: a wave stages two channel planes (680 of 704 slots)

struct UConvArgs {
    const float* xa;   // [B,Ca,H,W]
    const float* na;   // [B,Ca,2] (mean, 1/std) or null: source A is a plain tensor
    const float* xb;   // [B,Cb,H,W] or null
    const float* nb;
    const float* w;    // [Cout, Ca + Cb, 3, 3]
    float* y;          // [B,Cout,H,W] raw
    float* tstats;     // [B][ntiles][Cout][2] (mean, M2) per tile
    int Ca, Cb, B, Cout, H, W, tiles_x;
    float slope;
    int abl;           // debug (env MRX_UCONV_ABLATE): 1 no matrix work, 2 no stores, 4 no statistics
};

__device__ __forceinline__ float uc_leaky(float v, float slope) { return v > 0.f ? v : v * slope; }

template <int NCOT>
__global__ __launch_bounds__(UC_NT) void k_uconv(UConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem_uc[];
    float* Xs = smem_uc;                                  // [2][UC_CK][UC_PLANE]
    float* Ws = smem_uc + 2 * UC_CK * UC_PLANE;           // [2][9][NCOT][2][64]: MFMA A operand per lane
    constexpr int WBUF = 9 * NCOT * 2 * 64;
    constexpr int NWL = (WBUF + UC_NT - 1) / UC_NT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int tile = blockIdx.x, ty0 = tile / a.tiles_x;
    const int h0 = ty0 * UC_TH, w0 = (tile - ty0 * a.tiles_x) * UC_TW;
    const int b = blockIdx.z, co0 = blockIdx.y * (NCOT * 16);
    const long long plane = (long long)a.H * a.W;
    const int Ctot = a.Ca + a.Cb;
    const int nchunks = (Ctot + UC_CK - 1) / UC_CK;

    for (int i = tid; i < 2 * UC_CK * UC_PLANE; i += UC_NT) Xs[i] = 0.f;   // zero padding = the slots no step ever writes

    // this lane's tile slots: wave w stages channels 2w and 2w + 1 of a step, slot s = lane + 64 j (s < 340: first channel)
    unsigned goff[UC_XL];
    unsigned okm = 0;
#pragma unroll
    for (int j = 0; j < UC_XL; ++j) {
        const int s = lane + 64 * j, chl = s >= UC_PIX, pos = s - UC_PIX * chl;
        const int ry = pos / UC_PW, rx = pos - ry * UC_PW;
        const int gy = h0 + ry - 1, gx = w0 + rx - 1;
        const bool ok = s < 2 * UC_PIX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        goff[j] = ok ? (unsigned)(gy * a.W + gx) : 0u;
        okm |= (ok ? 1u : 0u) << j;
    }
    float xv[UC_XL];
    float nm0 = 0.f, ni0 = 1.f, nm1 = 0.f, ni1 = 1.f;      // (mean, 1/std) of the two channels in flight
    bool lz0 = false, lz1 = false, cv0 = false, cv1 = false;
    auto chan = [&](int c, const float*& p, float& m, float& iv, bool& lazy, bool& valid) {
        valid = c < Ctot;
        const int cc = valid ? c : 0;
        const bool inb = cc >= a.Ca;
        const int cl = inb ? cc - a.Ca : cc, Cs = inb ? a.Cb : a.Ca;
        p = (inb ? a.xb : a.xa) + ((long long)b * Cs + cl) * plane;
        const float* nrm = inb ? a.nb : a.na;
        lazy = nrm != nullptr;
        if (lazy) {
            m = nrm[((long long)b * Cs + cl) * 2];
            iv = nrm[((long long)b * Cs + cl) * 2 + 1];
        }
    };
    auto issue_x = [&](int q) {
        const float *p0, *p1;
        chan(UC_CK * q + 2 * wave, p0, nm0, ni0, lz0, cv0);
        chan(UC_CK * q + 2 * wave + 1, p1, nm1, ni1, lz1, cv1);
#pragma unroll
        for (int j = 0; j < UC_XL; ++j) {
            const int s = lane + 64 * j;
            const float* p = s >= UC_PIX ? p1 : p0;
            xv[j] = ((okm >> j) & 1u) ? p[goff[j]] : 0.f;
        }
    };
    auto commit_x = [&](int q) {
        float* dst = Xs + (q & 1) * (UC_CK * UC_PLANE) + 2 * wave * UC_PLANE;
#pragma unroll
        for (int j = 0; j < UC_XL; ++j) {
            const int s = lane + 64 * j;
            const bool second = s >= UC_PIX;
            float v = xv[j];
            if (second ? lz1 : lz0) v = uc_leaky((v - (second ? nm1 : nm0)) * (second ? ni1 : ni0), a.slope);
            if (!(second ? cv1 : cv0)) v = 0.f;                         // channels past the last one (their weights are zero too)
            if ((okm >> j) & 1u) dst[s + (UC_PLANE - UC_PIX) * (second ? 1 : 0)] = v;
        }
    };
    // weights of a step: element i = ((tap * NCOT + ct) * 2 + g) * 64 + lane' -> w[co0 + 16 ct + (lane' & 15)][8 q + 4 g + (lane' >> 4)][tap]
    float wr[NWL];
    auto issue_w = [&](int q) {
#pragma unroll
        for (int j = 0; j < NWL; ++j) {
            const int i = tid + j * UC_NT;
            const int ln = i & 63, g = (i >> 6) & 1, ct = (i >> 7) % NCOT, tap = (i >> 7) / NCOT;
            const int co = co0 + 16 * ct + (ln & 15), ci = UC_CK * q + 4 * g + (ln >> 4);
            wr[j] = (i < WBUF && co < a.Cout && ci < Ctot) ? a.w[((long long)co * Ctot + ci) * 9 + tap] : 0.f;
        }
    };
    auto commit_w = [&](int q) {
#pragma unroll
        for (int j = 0; j < NWL; ++j) {
            const int i = tid + j * UC_NT;
            if (i < WBUF) Ws[(q & 1) * WBUF + i] = wr[j];
        }
    };

    uc_f4 acc[4][NCOT];
#pragma unroll
    for (int sg = 0; sg < 4; ++sg)
#pragma unroll
        for (int ct = 0; ct < NCOT; ++ct) acc[sg][ct] = (uc_f4){0.f, 0.f, 0.f, 0.f};
    issue_x(0);
    issue_w(0);
    __syncthreads();      // the tile is zeroed before any value lands
    commit_x(0);
    commit_w(0);
    for (int q = 0; q < nchunks; ++q) {
        __syncthreads();  // step q staged; the buffers of step q - 1 are free
        if (q + 1 < nchunks) {
            issue_x(q + 1);
            issue_w(q + 1);
        }
        const float* xq = Xs + (q & 1) * (UC_CK * UC_PLANE) + lg * UC_PLANE + l15;
        const float* wq = Ws + (q & 1) * WBUF + lane;
        if (!(a.abl & 1))
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap % 3;
                float av[NCOT];
#pragma unroll
                for (int ct = 0; ct < NCOT; ++ct) av[ct] = wq[((tap * NCOT + ct) * 2 + g) * 64];
#pragma unroll
                for (int sg = 0; sg < 4; ++sg) {
                    const float bv = xq[4 * g * UC_PLANE + (2 * wave + (sg >> 1) + ky) * UC_PW + (sg & 1) * 16 + kx];
#pragma unroll
                    for (int ct = 0; ct < NCOT; ++ct) acc[sg][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ct], bv, acc[sg][ct], 0, 0, 0);
                }
            }
        if (q + 1 < nchunks) {
            commit_x(q + 1);
            commit_w(q + 1);
        }
    }

    bool ok[4];
#pragma unroll
    for (int sg = 0; sg < 4; ++sg) {
        const int oy = h0 + 2 * wave + (sg >> 1), ox = w0 + (sg & 1) * 16 + l15;
        ok[sg] = oy < a.H && ox < a.W;
        if (ok[sg] && !(a.abl & 2)) {
#pragma unroll
            for (int ct = 0; ct < NCOT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + 16 * ct + 4 * lg + r;
                    if (co < a.Cout) a.y[((long long)b * a.Cout + co) * plane + (long long)oy * a.W + ox] = acc[sg][ct][r];
                }
        }
    }
    if (a.abl & 4) return;
    // InstanceNorm statistics of this tile, per cout: mean over its valid pixels, then the squared deviations from that mean (two
    // fixed-order block reductions); k_unorm_finalize merges the tiles with the parallel-variance formula in double
    __syncthreads();
    float* red = smem_uc;   // [4 waves][NCOT * 16]
    const int nrow = a.H - h0 < UC_TH ? a.H - h0 : UC_TH, ncol = a.W - w0 < UC_TW ? a.W - w0 : UC_TW;
    const float inv_n = 1.0f / (float)(nrow * ncol);
    float mean[NCOT][4];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int ct = 0; ct < NCOT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float t = 0.f;
#pragma unroll
                for (int sg = 0; sg < 4; ++sg) {
                    const float v = acc[sg][ct][r];
                    const float d = pass == 0 ? v : (v - mean[ct][r]) * (v - mean[ct][r]);
                    t += ok[sg] ? d : 0.f;
                }
                for (int off = 8; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
                if (l15 == 0) red[wave * (NCOT * 16) + 16 * ct + 4 * lg + r] = t;
            }
        __syncthreads();
#pragma unroll
        for (int ct = 0; ct < NCOT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = 16 * ct + 4 * lg + r;
                const float t = (red[c] + red[NCOT * 16 + c]) + (red[2 * NCOT * 16 + c] + red[3 * NCOT * 16 + c]);
                if (pass == 0) {
                    mean[ct][r] = t * inv_n;
                } else if (wave == 0 && l15 == 0 && co0 + c < a.Cout) {
                    float* ts = a.tstats + (((long long)b * gridDim.x + tile) * a.Cout + co0 + c) * 2;
                    ts[0] = mean[ct][r];
                    ts[1] = t;
                }
            }
        __syncthreads();
    }
}

// per-tile (mean, M2) -> per-plane (mean, 1 / sqrt(M2 / n + eps)): one workgroup per (b, cout).  Two fixed-order sums in double over the tiles:
//   mean = sum_i n_i mean_i / N,   M2 = sum_i (M2_i + n_i (mean_i - mean)^2)      (the parallel-variance identity around the global mean)
// -- no division per tile (a pairwise Chan merge has one, in double, on the dependent chain of every thread: 4.6 us per launch, ten launches per
// cascade).  TILED: tile sizes follow from the tile index (8 x 32 tiles of an H x W plane); else every tile holds n_tile values but the last.
__device__ __forceinline__ double uc_block_sum(double v, double* sm) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    const int wv = threadIdx.x >> 6;
    __syncthreads();                                   // (the previous sum's readers are done with sm)
    if ((threadIdx.x & 63) == 0) sm[wv] = v;
    __syncthreads();
    double t = sm[0];
    for (int k = 1; k < UC_NT / 64; ++k) t += sm[k];   // the waves in order
    return t;
}
template <bool TILED>
__global__ __launch_bounds__(UC_NT) void k_unorm_finalize(const float* __restrict__ tstats, float* __restrict__ norm, int ntiles, int tiles_x,
                                                         int Cout, int H, int W, double n_tile, double n_last, float eps) {
    __shared__ double sm[UC_NT / 64];
    const int plane_id = blockIdx.x, b = plane_id / Cout, co = plane_id - b * Cout;
    auto count = [&](int t) {
        if (TILED) {
            const int ty = t / tiles_x, tx = t - ty * tiles_x;
            const int nr = H - ty * UC_TH < UC_TH ? H - ty * UC_TH : UC_TH, nc = W - tx * UC_TW < UC_TW ? W - tx * UC_TW : UC_TW;
            return (double)(nr * nc);
        }
        return t == ntiles - 1 ? n_last : n_tile;
    };
    const float* ts = tstats + ((long long)b * ntiles * Cout + co) * 2;
    double sn = 0.0, smean = 0.0;
    for (int t = threadIdx.x; t < ntiles; t += UC_NT) {
        const double nb = count(t);
        sn += nb;
        smean += nb * (double)ts[(long long)t * Cout * 2];
    }
    const double N = uc_block_sum(sn, sm);
    const double mean = uc_block_sum(smean, sm) / N;
    double q = 0.0;
    for (int t = threadIdx.x; t < ntiles; t += UC_NT) {
        const float2 p = *reinterpret_cast<const float2*>(ts + (long long)t * Cout * 2);
        const double d = (double)p.x - mean;
        q += (double)p.y + count(t) * d * d;
    }
    const double m2 = uc_block_sum(q, sm);
    if (threadIdx.x == 0) {
        norm[(long long)plane_id * 2] = (float)mean;
        norm[(long long)plane_id * 2 + 1] = 1.0f / sqrtf((float)m2 / (float)N + eps);
    }
}

// (shared with the two-term fp16 convolution, unet_f16.hip)
int mrx_unorm_finalize_tiled(const float* tstats, float* norm, int B, int ntiles, int tiles_x, int Cout, int H, int W, float eps, hipStream_t st) {
    hipLaunchKernelGGL(k_unorm_finalize<true>, dim3(B * Cout), dim3(UC_NT), 0, st, tstats, norm, ntiles, tiles_x, Cout, H, W, 0.0, 0.0, eps);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

template <int NCOT>
static void launch_uconv(const UConvArgs& a, int ntiles, hipStream_t st) {
    constexpr size_t lds = sizeof(float) * (2 * UC_CK * UC_PLANE + 2 * 9 * NCOT * 2 * 64);
    static bool attr_done = false;   // once: keeps launches legal under hipGraph capture
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)k_uconv<NCOT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    hipLaunchKernelGGL((k_uconv<NCOT>), dim3(ntiles, mrx_cdiv(a.Cout, NCOT * 16), a.B), dim3(UC_NT), lds, st, a);
}

extern "C" int64_t mrx_unet_conv3x3_work_floats(int B, int Cout, int H, int W) {
    if (B < 0 || Cout < 1 || H < 1 || W < 1) return -1;
    return (int64_t)B * mrx_cdiv(W, UC_TW) * mrx_cdiv(H, UC_TH) * Cout * 2;
}

extern "C" int mrx_unet_conv3x3(const float* xa, const float* na, int Ca, const float* xb, const float* nb, int Cb, const float* w, float* y,
                                float* norm, float* work, int B, int Cout, int H, int W, float eps, float slope, void* stream) {
    MRX_REQUIRE(xa && w && y && norm && work && Ca >= 1 && Cb >= 0 && (Cb == 0 || xb), MRX_EINVAL, "mrx_unet_conv3x3: bad argument");
    MRX_REQUIRE(B >= 0 && Cout >= 1 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_unet_conv3x3: bad dims");
    MRX_REQUIRE(B <= 65535 && Cout <= 16 * 65535 && (long long)H * W < (1ll << 30), MRX_EUNSUP, "mrx_unet_conv3x3: size");
    if (B == 0) return MRX_OK;
    UConvArgs a;
    a.xa = xa, a.na = na, a.xb = Cb ? xb : nullptr, a.nb = Cb ? nb : nullptr, a.w = w, a.y = y, a.tstats = work;
    a.Ca = Ca, a.Cb = Cb, a.B = B, a.Cout = Cout, a.H = H, a.W = W, a.tiles_x = mrx_cdiv(W, UC_TW), a.slope = slope;
    static const int abl = MRX_DEBUG_ENV("MRX_UCONV_ABLATE") ? atoi(MRX_DEBUG_ENV("MRX_UCONV_ABLATE")) : 0;
    a.abl = abl;
    const int ntiles = a.tiles_x * mrx_cdiv(H, UC_TH);
    const int ncot = (Cout + 15) / 16;
    hipStream_t st = (hipStream_t)stream;
    // few tiles (the pooled levels): one cout block per workgroup, cout blocks across grid.y, to fill the chip
    if (ncot == 1 || (long long)ntiles * B < 512) launch_uconv<1>(a, ntiles, st);
    else if (ncot == 2) launch_uconv<2>(a, ntiles, st);
    else if (ncot == 3) launch_uconv<3>(a, ntiles, st);
    else launch_uconv<4>(a, ntiles, st);
    hipLaunchKernelGGL(k_unorm_finalize<true>, dim3(B * Cout), dim3(UC_NT), 0, st, (const float*)work, norm, ntiles, a.tiles_x, Cout, H, W, 0.0, 0.0,
                       eps);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- ConvTranspose2d(k 2, s 2, no bias) of a (raw, norm) input: one thread per INPUT pixel and group of COG output channels ------------
// (the formulation of k_convT2x2_t, unet.hip; the input value is normalised + activated as it is loaded)
template <int COG>
__global__ __launch_bounds__(UC_NT) void k_uconvT(const float* __restrict__ x, const float* __restrict__ nrm, const float* __restrict__ w,
                                                  float* __restrict__ out, int Cin, int Cout, int H, int W, float slope,
                                                  float* __restrict__ tstats) {
    extern __shared__ __attribute__((aligned(16))) float wsm_u[];  // [Cin][COG] quads of this channel group, then [Cin] (mean, 1/std)
    float* nsm = wsm_u + Cin * COG * 4;
    const int g0 = blockIdx.y * COG, b = blockIdx.z;
    for (int i = threadIdx.x; i < Cin * COG * 4; i += UC_NT) {
        const int q = i & 3, co = (i >> 2) % COG, ci = (i >> 2) / COG;
        wsm_u[i] = w[((long long)ci * Cout + g0 + co) * 4 + q];
    }
    for (int i = threadIdx.x; i < 2 * Cin; i += UC_NT) nsm[i] = nrm ? nrm[(long long)b * Cin * 2 + i] : ((i & 1) ? 1.f : 0.f);
    __syncthreads();
    const long long HW = (long long)H * W;
    const long long pix_raw = (long long)blockIdx.x * UC_NT + threadIdx.x;
    const bool live = pix_raw < HW;
    const long long pix = live ? pix_raw : HW - 1;         // idle threads stay for the block reductions
    const int y = (int)(pix / W), xx = (int)(pix - (long long)y * W);
    const float* xp = x + (long long)b * Cin * HW + pix;
    const bool lazy = nrm != nullptr;
    uc_f4 acc[COG];
#pragma unroll
    for (int co = 0; co < COG; ++co) acc[co] = (uc_f4){0.f, 0.f, 0.f, 0.f};
    int ci = 0;
    // four input planes per round, the NEXT round's four requested before this round's 4 x 4 COG multiply-adds (round 5: as a plain loop every round waited
    // for its own loads -- seven memory round trips one after the other for a 28-channel layer, at two workgroups per CU)
    float vn[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) vn[u] = xp[(long long)(u < Cin ? u : 0) * HW];
    for (; ci + 4 <= Cin; ci += 4) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = vn[u];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int cn = ci + 4 + u;
            vn[u] = xp[(long long)(cn < Cin ? cn : 0) * HW];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (lazy) v[u] = uc_leaky((v[u] - nsm[2 * (ci + u)]) * nsm[2 * (ci + u) + 1], slope);
            const uc_f4* wq = reinterpret_cast<const uc_f4*>(wsm_u) + (ci + u) * COG;
#pragma unroll
            for (int co = 0; co < COG; ++co) acc[co] += v[u] * wq[co];
        }
    }
    for (; ci < Cin; ++ci) {
        float v = vn[ci & 3];                      // (the rest of the last round, already requested)
        if (lazy) v = uc_leaky((v - nsm[2 * ci]) * nsm[2 * ci + 1], slope);
        const uc_f4* wq = reinterpret_cast<const uc_f4*>(wsm_u) + ci * COG;
#pragma unroll
        for (int co = 0; co < COG; ++co) acc[co] += v * wq[co];
    }
    const int OW = 2 * W;
    float* op = out + (((long long)b * Cout + g0) * 2 * H + 2 * y) * OW + 2 * xx;
    if (live) {
#pragma unroll
        for (int co = 0; co < COG; ++co) {
            float* o = op + (long long)co * 4 * HW;
            *reinterpret_cast<uc_f2*>(o) = (uc_f2){acc[co][0], acc[co][1]};
            *reinterpret_cast<uc_f2*>(o + OW) = (uc_f2){acc[co][2], acc[co][3]};
        }
    }
    // tile statistics of the COG planes: two block reductions in all (one per pass, every plane at once; a wave's sum on the vector ALU:
    // DPP row sums + four readlanes) -- one pair of barriers per plane made the 14-plane form 56 barriers long
    __shared__ float red[2][UC_NT / 64][COG];
    const long long rem = HW - (long long)blockIdx.x * UC_NT;
    const float inv_n = 1.0f / (4.0f * (float)(rem < UC_NT ? rem : UC_NT));
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    auto wave_sum = [](float t) {
#define UC_DPP(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true))
        t += UC_DPP(t, 0xB1);
        t += UC_DPP(t, 0x4E);
        t += UC_DPP(t, 0x141);
        t += UC_DPP(t, 0x140);      // every lane of a row holds the row's sum
#undef UC_DPP
        const int ti = __builtin_bit_cast(int, t);
        return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(ti, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(ti, 16))) +
               (__builtin_bit_cast(float, __builtin_amdgcn_readlane(ti, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(ti, 48)));
    };
    float mean[COG];
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
        for (int co = 0; co < COG; ++co) {
            float t = 0.f;
            if (live) {
#pragma unroll
                for (int q = 0; q < 4; ++q) t += pass == 0 ? acc[co][q] : (acc[co][q] - mean[co]) * (acc[co][q] - mean[co]);
            }
            t = wave_sum(t);
            if (lane == 0) red[pass][wv][co] = t;
        }
        __syncthreads();
#pragma unroll
        for (int co = 0; co < COG; ++co) {
            const float tot = (red[pass][0][co] + red[pass][1][co]) + (red[pass][2][co] + red[pass][3][co]);
            if (pass == 0)
                mean[co] = tot * inv_n;
            else if (threadIdx.x == 0) {
                float* ts = tstats + (((long long)b * gridDim.x + blockIdx.x) * Cout + g0 + co) * 2;
                ts[0] = mean[co];
                ts[1] = tot;
            }
        }
    }
}

template <int COG>
static void launch_uconvT(const float* x, const float* nrm, const float* w, float* out, int B, int Cin, int Cout, int H, int W, float slope,
                          float* tstats, hipStream_t st) {
    const size_t lds = sizeof(float) * ((size_t)Cin * COG * 4 + 2 * (size_t)Cin);
    const dim3 grid((unsigned)(((long long)H * W + UC_NT - 1) / UC_NT), Cout / COG, B);
    hipLaunchKernelGGL((k_uconvT<COG>), grid, dim3(UC_NT), lds, st, x, nrm, w, out, Cin, Cout, H, W, slope, tstats);
}

extern "C" int64_t mrx_unet_conv_transpose2x2_work_floats(int B, int Cout, int H, int W) {
    if (B < 0 || Cout < 1 || H < 1 || W < 1) return -1;
    return (int64_t)B * (((long long)H * W + UC_NT - 1) / UC_NT) * Cout * 2;
}

// x [B,Cin,H,W] (+ nrm [B,Cin,2] or NULL), w [Cin,Cout,2,2] -> out [B,Cout,2H,2W] raw, norm [B,Cout,2]; even Cout, Cin <= 1228
extern "C" int mrx_unet_conv_transpose2x2(const float* x, const float* nrm, const float* w, float* out, float* norm, float* work, int B, int Cin,
                                 int Cout, int H, int W, float eps, float slope, void* stream) {
    MRX_REQUIRE(x && w && out && norm && work && B >= 0 && Cin >= 1 && Cout >= 1 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_unet_conv_transpose2x2: bad argument");
    MRX_REQUIRE(B <= 65535 && Cout % 2 == 0 && Cout <= 65535 * 2 && (size_t)Cin * (2 * 16 + 8) <= 48 * 1024 && (((uintptr_t)out) & 7) == 0,
                MRX_EUNSUP, "mrx_unet_conv_transpose2x2: Cout=%d Cin=%d", Cout, Cin);
    if (B == 0) return MRX_OK;
    hipStream_t st = (hipStream_t)stream;
    const long long HW = (long long)H * W, ntiles = (HW + UC_NT - 1) / UC_NT;
    auto fits = [&](int cog) { return (size_t)Cin * (cog * 16 + 8) <= 48 * 1024; };     // the group's weight quads + the input norm in LDS
    // the largest group of output channels per workgroup (every group re-reads and re-normalises the input) that still leaves a workgroup per CU
    auto use = [&](int cog) { return Cout % cog == 0 && fits(cog) && ntiles * B * (Cout / cog) >= 256; };
    if (use(14)) launch_uconvT<14>(x, nrm, w, out, B, Cin, Cout, H, W, slope, work, st);
    else if (use(12)) launch_uconvT<12>(x, nrm, w, out, B, Cin, Cout, H, W, slope, work, st);
    else if (use(8)) launch_uconvT<8>(x, nrm, w, out, B, Cin, Cout, H, W, slope, work, st);
    else if (use(6)) launch_uconvT<6>(x, nrm, w, out, B, Cin, Cout, H, W, slope, work, st);
    else launch_uconvT<2>(x, nrm, w, out, B, Cin, Cout, H, W, slope, work, st);
    hipLaunchKernelGGL(k_unorm_finalize<false>, dim3(B * Cout), dim3(UC_NT), 0, st, (const float*)work, norm, (int)ntiles, 0, Cout, 0, 0,
                       4.0 * UC_NT, 4.0 * (double)(HW - (ntiles - 1) * UC_NT), eps);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- avg_pool2d(2) of a (raw, norm) tensor -> plain (unet_block.py:206) ---------------------------------------------------------------
__global__ __launch_bounds__(UC_NT) void k_uavgpool(const float* __restrict__ x, const float* __restrict__ nrm, float* __restrict__ out,
                                                    int H, int W, float slope) {
    const int OH = H / 2, OW = W / 2;
    const long long p = blockIdx.y;                        // plane
    const float m = nrm ? nrm[2 * p] : 0.f, iv = nrm ? nrm[2 * p + 1] : 1.f;
    const bool lazy = nrm != nullptr;
    const float* xp = x + p * (long long)H * W;
    float* op = out + p * (long long)OH * OW;
    for (long long o = (long long)blockIdx.x * UC_NT + threadIdx.x; o < (long long)OH * OW; o += (long long)gridDim.x * UC_NT) {
        const int oy = (int)(o / OW), ox = (int)(o - (long long)oy * OW);
        const float* r0 = xp + (long long)(2 * oy) * W + 2 * ox;
        float v00 = r0[0], v01 = r0[1], v10 = r0[W], v11 = r0[W + 1];
        if (lazy) {
            v00 = uc_leaky((v00 - m) * iv, slope), v01 = uc_leaky((v01 - m) * iv, slope);
            v10 = uc_leaky((v10 - m) * iv, slope), v11 = uc_leaky((v11 - m) * iv, slope);
        }
        op[o] = ((v00 + v01) + (v10 + v11)) * 0.25f;
    }
}
extern "C" int mrx_unet_avgpool(const float* x, const float* nrm, float* out, int64_t planes, int H, int W, float slope, void* stream) {
    MRX_REQUIRE(x && out && planes >= 0 && H >= 2 && W >= 2, MRX_EINVAL, "mrx_unet_avgpool: bad argument");
    MRX_REQUIRE(planes <= 65535, MRX_EUNSUP, "mrx_unet_avgpool: too many planes");
    if (planes == 0) return MRX_OK;
    const long long n = (long long)(H / 2) * (W / 2);
    const int gx = (int)((n + 4 * UC_NT - 1) / (4 * UC_NT));
    hipLaunchKernelGGL(k_uavgpool, dim3(gx > 0 ? gx : 1, (unsigned)planes), dim3(UC_NT), 0, (hipStream_t)stream, x, nrm, out, H, W, slope);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- leaky((x - mean) / std) written out (fallback paths, tests) ------------------------------------------------------------------------
__global__ __launch_bounds__(UC_NT) void k_uapply(const float* __restrict__ x, const float* __restrict__ nrm, float* __restrict__ out,
                                                  long long HW, float slope) {
    const long long p = blockIdx.y;
    const float m = nrm[2 * p], iv = nrm[2 * p + 1];
    for (long long o = (long long)blockIdx.x * UC_NT + threadIdx.x; o < HW; o += (long long)gridDim.x * UC_NT)
        out[p * HW + o] = uc_leaky((x[p * HW + o] - m) * iv, slope);
}
extern "C" int mrx_unet_apply(const float* x, const float* nrm, float* out, int64_t planes, int64_t HW, float slope, void* stream) {
    MRX_REQUIRE(x && nrm && out && planes >= 0 && HW >= 1, MRX_EINVAL, "mrx_unet_apply: bad argument");
    MRX_REQUIRE(planes <= 65535, MRX_EUNSUP, "mrx_unet_apply: too many planes");
    if (planes == 0) return MRX_OK;
    const int gx = (int)((HW + 4 * UC_NT - 1) / (4 * UC_NT));
    hipLaunchKernelGGL(k_uapply, dim3(gx > 0 ? gx : 1, (unsigned)planes), dim3(UC_NT), 0, (hipStream_t)stream, x, nrm, out, (long long)HW, slope);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- the closing 1x1 convolution (+ bias) of a (raw, norm) tensor, Cout <= 4 (unet_block.py:186-189) -----------------------------------
__global__ __launch_bounds__(UC_NT) void k_uconv1x1(const float* __restrict__ x, const float* __restrict__ nrm, const float* __restrict__ w,
                                                    const float* __restrict__ bias, float* __restrict__ out, int Cin, int Cout, long long HW,
                                                    float slope) {
    extern __shared__ float sm_u1[];           // [Cin] (mean, 1/std), then [Cout][Cin] weights
    float* wsm = sm_u1 + 2 * Cin;
    const int b = blockIdx.y;
    for (int i = threadIdx.x; i < 2 * Cin; i += UC_NT) sm_u1[i] = nrm ? nrm[(long long)b * Cin * 2 + i] : ((i & 1) ? 1.f : 0.f);
    for (int i = threadIdx.x; i < Cout * Cin; i += UC_NT) wsm[i] = w[i];
    __syncthreads();
    const bool lazy = nrm != nullptr;
    for (long long o = (long long)blockIdx.x * UC_NT + threadIdx.x; o < HW; o += (long long)gridDim.x * UC_NT) {
        float acc[4];
#pragma unroll
        for (int co = 0; co < 4; ++co) acc[co] = (bias && co < Cout) ? bias[co] : 0.f;
        const float* xp = x + (long long)b * Cin * HW + o;
        for (int c = 0; c < Cin; ++c) {
            float v = xp[(long long)c * HW];
            if (lazy) v = uc_leaky((v - sm_u1[2 * c]) * sm_u1[2 * c + 1], slope);
#pragma unroll
            for (int co = 0; co < 4; ++co)
                if (co < Cout) acc[co] += v * wsm[co * Cin + c];
        }
#pragma unroll
        for (int co = 0; co < 4; ++co)
            if (co < Cout) out[((long long)b * Cout + co) * HW + o] = acc[co];
    }
}
extern "C" int mrx_unet_conv1x1(const float* x, const float* nrm, const float* w, const float* bias, float* out, int B, int Cin, int Cout,
                                int64_t HW, float slope, void* stream) {
    MRX_REQUIRE(x && w && out && B >= 0 && Cin >= 1 && Cout >= 1 && HW >= 1, MRX_EINVAL, "mrx_unet_conv1x1: bad argument");
    MRX_REQUIRE(Cout <= 4 && B <= 65535 && (size_t)Cin * (2 + Cout) * 4 <= 48 * 1024, MRX_EUNSUP, "mrx_unet_conv1x1: Cout=%d Cin=%d", Cout, Cin);
    if (B == 0) return MRX_OK;
    const int gx = (int)((HW + 2 * UC_NT - 1) / (2 * UC_NT));
    hipLaunchKernelGGL(k_uconv1x1, dim3(gx > 0 ? gx : 1, B), dim3(UC_NT), sizeof(float) * (size_t)Cin * (2 + Cout), (hipStream_t)stream, x, nrm, w,
                       bias, out, Cin, Cout, (long long)HW, slope);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- NormUnet head and tail on complex-last tensors (unet_block.py:46-136 with norm_groups = 2) -----------------------------------------
// forward(): complex_to_chan_dim (a permute copy), norm (mean / unbiased std per (batch, real | imaginary) group: three passes + apply),
// pad (a copy) before the U-Net; unpad, unnorm, chan_complex_to_last_dim (two copies + apply) after it.  Here: two statistics passes over
// the complex-last input (both components at once), ONE pass that normalises, permutes and pads; and the closing 1x1 convolution writes the
// cropped, un-normalised, complex-last result itself.
#define UC_NSPLIT_MAX 128
static int uc_nsplit(long long n) {
    long long s = (n + 2047) / 2048;
    return (int)(s < 1 ? 1 : (s > UC_NSPLIT_MAX ? UC_NSPLIT_MAX : s));
}
// fixed-order sum of `count` (<= 128) float2 partials by one wave; every lane returns the total
__device__ __forceinline__ float2 uc_combine2(const float2* p, int count, int lane) {
    float2 a = lane < count ? p[lane] : make_float2(0.f, 0.f);
    if (lane + 64 < count) {
        const float2 b = p[lane + 64];
        a.x += b.x, a.y += b.y;
    }
    for (int off = 32; off > 0; off >>= 1) a.x += __shfl_xor(a.x, off, 64), a.y += __shfl_xor(a.y, off, 64);
    return a;
}
// PASS 0: partial sums; PASS 1: partial sums of squared deviations from the mean (recombined from the partial sums)
template <int PASS>
__global__ __launch_bounds__(UC_NT) void k_ucnorm_part(const float2* __restrict__ x, const float2* __restrict__ psum, float2* __restrict__ part,
                                                       long long n, int nsplit) {
    __shared__ float2 red[UC_NT / 64];
    __shared__ float2 s_mean;
    const int b = blockIdx.y, sp = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float2 mean = make_float2(0.f, 0.f);
    if (PASS == 1) {
        if (wv == 0) {
            const float2 t = uc_combine2(psum + (long long)b * nsplit, nsplit, lane);
            if (lane == 0) s_mean = make_float2(t.x / (float)n, t.y / (float)n);
        }
        __syncthreads();
        mean = s_mean;
    }
    const long long per = (n + nsplit - 1) / nsplit, i0 = sp * per, i1 = i0 + per < n ? i0 + per : n;
    const float2* xb = x + (long long)b * n;
    float2 acc = make_float2(0.f, 0.f);
    for (long long i = i0 + threadIdx.x; i < i1; i += UC_NT) {
        const float2 v = xb[i];
        if (PASS == 0)
            acc.x += v.x, acc.y += v.y;
        else
            acc.x += (v.x - mean.x) * (v.x - mean.x), acc.y += (v.y - mean.y) * (v.y - mean.y);
    }
    for (int off = 32; off > 0; off >>= 1) acc.x += __shfl_xor(acc.x, off, 64), acc.y += __shfl_xor(acc.y, off, 64);
    if (lane == 0) red[wv] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
        part[(long long)b * nsplit + sp] = make_float2((red[0].x + red[1].x) + (red[2].x + red[3].x), (red[0].y + red[1].y) + (red[2].y + red[3].y));
}
// out[b][comp * c + coil][y + top][x + left] = (x[b][coil][y][x][comp] - mean[b][comp]) / std[b][comp], zero in the padding; mean / std [B,2]
__global__ __launch_bounds__(UC_NT) void k_ucnorm_apply(const float2* __restrict__ x, const float2* __restrict__ psum, const float2* __restrict__ psq,
                                                        float* __restrict__ out, float* __restrict__ mean_o, float* __restrict__ std_o, int c, int H,
                                                        int W, int top, int left, int OH, int OW, int nsplit) {
    __shared__ float2 s_ms[2];
    const int b = blockIdx.z, coil = blockIdx.y, lane = threadIdx.x & 63;
    const long long n = (long long)c * H * W;
    if (threadIdx.x < 64) {
        const float2 s = uc_combine2(psum + (long long)b * nsplit, nsplit, lane), q = uc_combine2(psq + (long long)b * nsplit, nsplit, lane);
        if (lane == 0) {
            s_ms[0] = make_float2(s.x / (float)n, s.y / (float)n);
            s_ms[1] = make_float2(sqrtf(q.x / (float)(n - 1)), sqrtf(q.y / (float)(n - 1)));     // unbiased (unet_block.py:79)
            if (blockIdx.x == 0 && coil == 0) {
                mean_o[2 * b] = s_ms[0].x, mean_o[2 * b + 1] = s_ms[0].y;
                std_o[2 * b] = s_ms[1].x, std_o[2 * b + 1] = s_ms[1].y;
            }
        }
    }
    __syncthreads();
    const float2 m = s_ms[0], sd = s_ms[1];
    const float2* xb = x + ((long long)b * c + coil) * H * W;
    float* o_re = out + ((long long)b * 2 * c + coil) * OH * OW;
    float* o_im = out + ((long long)b * 2 * c + c + coil) * OH * OW;
    for (long long o = (long long)blockIdx.x * UC_NT + threadIdx.x; o < (long long)OH * OW; o += (long long)gridDim.x * UC_NT) {
        const int oy = (int)(o / OW), ox = (int)(o - (long long)oy * OW);
        const int iy = oy - top, ix = ox - left;
        float2 v = make_float2(0.f, 0.f);
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
            const float2 t = xb[(long long)iy * W + ix];
            v = make_float2((t.x - m.x) / sd.x, (t.y - m.y) / sd.y);
        }
        o_re[o] = v.x;
        o_im[o] = v.y;
    }
}
extern "C" int64_t mrx_unet_cnorm_work_floats(int B) { return B < 0 ? -1 : (int64_t)B * UC_NSPLIT_MAX * 4; }
// x [B,c,H,W,2] -> out [B,2c,H+top+bottom,W+left+right] (channel = component * c + coil: complex_to_chan_dim), normalised per (batch, component)
// with the unbiased std, zero padded; mean, std [B,2] (norm_groups = 2).  work: mrx_unet_cnorm_work_floats(B) floats, 8-byte aligned.
extern "C" int mrx_unet_cnorm_pad(const float* x, float* out, float* mean, float* std_, float* work, int B, int c, int H, int W, int top,
                                  int bottom, int left, int right, void* stream) {
    MRX_REQUIRE(x && out && mean && std_ && work && B >= 0 && c >= 1 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_unet_cnorm_pad: bad argument");
    MRX_REQUIRE(top >= 0 && bottom >= 0 && left >= 0 && right >= 0 && B <= 65535 && c <= 65535 && (((uintptr_t)x | (uintptr_t)work) & 7) == 0,
                MRX_EUNSUP, "mrx_unet_cnorm_pad: unsupported shape / alignment");
    if (B == 0) return MRX_OK;
    hipStream_t st = (hipStream_t)stream;
    const long long n = (long long)c * H * W;
    const int ns = uc_nsplit(n), OH = H + top + bottom, OW = W + left + right;
    float2* psum = reinterpret_cast<float2*>(work);
    float2* psq = psum + (size_t)B * UC_NSPLIT_MAX;
    const float2* x2 = reinterpret_cast<const float2*>(x);
    hipLaunchKernelGGL(k_ucnorm_part<0>, dim3(ns, B), dim3(UC_NT), 0, st, x2, (const float2*)nullptr, psum, n, ns);
    hipLaunchKernelGGL(k_ucnorm_part<1>, dim3(ns, B), dim3(UC_NT), 0, st, x2, (const float2*)psum, psq, n, ns);
    const int gx = (int)(((long long)OH * OW + 4 * UC_NT - 1) / (4 * UC_NT));
    hipLaunchKernelGGL(k_ucnorm_apply, dim3(gx > 0 ? gx : 1, c, B), dim3(UC_NT), 0, st, x2, (const float2*)psum, (const float2*)psq, out, mean, std_, c, H,
                       W, top, left, OH, OW, ns);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// the closing 1x1 convolution (+ bias) of a (raw, norm) tensor [B,Cin,OH,OW] into 2c channels, written as the cropped, un-normalised complex-last
// result: out[b][coil][y][x][comp] = conv[b][comp * c + coil][y + top][x + left] * std[b][comp] + mean[b][comp]   (unet_block.py:186-189, 91, 56-61)
__global__ __launch_bounds__(UC_NT) void k_uconv1x1_c(const float* __restrict__ x, const float* __restrict__ nrm, const float* __restrict__ w,
                                                      const float* __restrict__ bias, const float* __restrict__ mean, const float* __restrict__ std_,
                                                      float2* __restrict__ out, int Cin, int c, int OH, int OW, int top, int left, int H, int W,
                                                      float slope) {
    extern __shared__ float sm_u1c[];          // [Cin] (mean, 1/std), then [2c][Cin] weights
    float* wsm = sm_u1c + 2 * Cin;
    const int b = blockIdx.y, Cout = 2 * c;
    for (int i = threadIdx.x; i < 2 * Cin; i += UC_NT) sm_u1c[i] = nrm ? nrm[(long long)b * Cin * 2 + i] : ((i & 1) ? 1.f : 0.f);
    for (int i = threadIdx.x; i < Cout * Cin; i += UC_NT) wsm[i] = w[i];
    __syncthreads();
    const bool lazy = nrm != nullptr;
    const float m_re = mean[2 * b], m_im = mean[2 * b + 1], s_re = std_[2 * b], s_im = std_[2 * b + 1];
    const long long HWo = (long long)OH * OW;
    for (long long o = (long long)blockIdx.x * UC_NT + threadIdx.x; o < (long long)H * W; o += (long long)gridDim.x * UC_NT) {
        const int y = (int)(o / W), xx = (int)(o - (long long)y * W);
        float acc[4];
#pragma unroll
        for (int co = 0; co < 4; ++co) acc[co] = (bias && co < Cout) ? bias[co] : 0.f;
        const float* xp = x + (long long)b * Cin * HWo + (long long)(y + top) * OW + xx + left;
        int ci = 0;
        for (; ci + 7 <= Cin; ci += 7) {          // seven planes in flight per thread (a one-load-at-a-time loop left the pass latency-bound)
            float v[7];
#pragma unroll
            for (int u = 0; u < 7; ++u) v[u] = xp[(long long)(ci + u) * HWo];
#pragma unroll
            for (int u = 0; u < 7; ++u) {
                if (lazy) v[u] = uc_leaky((v[u] - sm_u1c[2 * (ci + u)]) * sm_u1c[2 * (ci + u) + 1], slope);
#pragma unroll
                for (int co = 0; co < 4; ++co)
                    if (co < Cout) acc[co] += v[u] * wsm[co * Cin + ci + u];
            }
        }
        for (; ci < Cin; ++ci) {
            float v = xp[(long long)ci * HWo];
            if (lazy) v = uc_leaky((v - sm_u1c[2 * ci]) * sm_u1c[2 * ci + 1], slope);
#pragma unroll
            for (int co = 0; co < 4; ++co)
                if (co < Cout) acc[co] += v * wsm[co * Cin + ci];
        }
        for (int coil = 0; coil < c; ++coil)
            out[((long long)b * c + coil) * H * W + o] = make_float2(acc[coil] * s_re + m_re, acc[c + coil] * s_im + m_im);
    }
}
extern "C" int mrx_unet_conv1x1_cunnorm(const float* x, const float* nrm, const float* w, const float* bias, const float* mean, const float* std_,
                                        float* out, int B, int Cin, int c, int OH, int OW, int top, int left, int H, int W, float slope,
                                        void* stream) {
    MRX_REQUIRE(x && w && mean && std_ && out && B >= 0 && Cin >= 1 && c >= 1 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_unet_conv1x1_cunnorm: bad argument");
    MRX_REQUIRE(top >= 0 && left >= 0 && top + H <= OH && left + W <= OW, MRX_EINVAL, "mrx_unet_conv1x1_cunnorm: crop outside the input");
    MRX_REQUIRE(c <= 2 && B <= 65535 && (size_t)Cin * (2 + 2 * c) * 4 <= 48 * 1024 && (((uintptr_t)out) & 7) == 0, MRX_EUNSUP,
                "mrx_unet_conv1x1_cunnorm: c=%d Cin=%d", c, Cin);
    if (B == 0) return MRX_OK;
    const int gx = (int)(((long long)H * W + 2 * UC_NT - 1) / (2 * UC_NT));
    hipLaunchKernelGGL(k_uconv1x1_c, dim3(gx > 0 ? gx : 1, B), dim3(UC_NT), sizeof(float) * (size_t)Cin * (2 + 2 * c), (hipStream_t)stream, x, nrm, w, bias,
                       mean, std_, reinterpret_cast<float2*>(out), Cin, c, OH, OW, top, left, H, W, slope);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
