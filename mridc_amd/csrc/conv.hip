// conv.hip -- convolutional regulariser kernels for gfx950 (NCHW fp32, stride 1, "same" size):
//   mrx_conv2d            generic implicit-GEMM conv on the fp32-input matrix cores (v_mfma_f32_32x32x2_f32: exact
//                         fp32 fma chains), replicate or zero border handled in the LDS tile loader
//                         (reference models/rim/conv_layers.py:72-85, rnn_cells.py:23-38, unet_block.py:251,255)
//   mrx_rim_layer_indrnn  ConvNonlinear + IndRNNCell fused: the conv accumulators (D[cout][pixel]) are re-used
//                         *in registers* as the B operand of the 1x1 `ih` GEMM -- the contraction index of the second
//                         GEMM is enumerated in the order the first one's C/D layout delivers it, so the 64-channel
//                         activation never leaves the register file (conv_layers.py:121-123 + rnn_cells.py:384-391)
//   mrx_indrnn_cell       stand-alone IndRNN cell (zero-padded ih conv + hh*h + ReLU epilogue)
//   mrx_rim_final         F->2 conv + eta update on the vector ALUs (rim_block.py:239-248)
//
// GEMM roles: D[M = 32 couts][N = 32 consecutive pixels of one image row] += A[cout][k] * B[k][pixel] with
// k = (cin, tap); one MFMA consumes two cins at one tap (lanes 0-31: cin c, lanes 32-63: cin c+1).
// A workgroup (4 waves) owns an 8-row x 32-column output tile for up to 64 couts; each wave owns 2 rows x 2 cout
// tiles = 4 independent accumulators, which is what the 64-cycle MFMA needs to issue back to back.
#include <cstdlib>
#include <mutex>
#include <unordered_map>

#include "mrx_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CV_NT 256
#define CV_TW MRX_CONV_TILE_W
#define CV_TH MRX_CONV_TILE_H

struct ConvArgs {
    const float* x;
    const float* w;
    const float* bias;
    float* y;
    const float* hh;     // optional per-channel recurrent weight (IndRNN epilogue), plain mode
    const float* hprev;  // optional previous hidden state [B,Cout,H,W]
    const float* w_ih;   // fused mode: 1x1 weights [F][F]
    const float* b_ih;
    float* tstats;       // tuned 3x3 with fused InstanceNorm statistics: per-tile (mean, M2) per cout, [B][ntiles][Cout][2]
    int B, Cin, Cout, H, W, k, dil, pad, pad_mode, act;
    float slope;
    int tiles_x, CK, PH, PW;
};

__device__ __forceinline__ float act_apply(float v, int act, float slope) {
    if (act == MRX_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == MRX_ACT_LEAKY) return v > 0.f ? v : v * slope;
    return v;
}

// Stage CK input channels of the halo'd tile into LDS (border: clamp = ReplicationPad2d, or zeros).
__device__ __forceinline__ void stage_x(float* Xs, const ConvArgs& a, const float* xb, int c0, int h0, int w0) {
    const int plane = a.PH * a.PW;
    const int total = a.CK * plane;
    for (int idx = threadIdx.x; idx < total; idx += CV_NT) {
        const int ci = idx / plane;
        const int rem = idx - ci * plane;
        const int ty = rem / a.PW, tx = rem - ty * a.PW;
        const int gc = c0 + ci;
        int gy = h0 + ty - a.pad, gx = w0 + tx - a.pad;
        float v = 0.f;
        if (gc < a.Cin) {
            if (a.pad_mode == MRX_PAD_REPLICATE) {
                gy = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy);
                gx = gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx);
                v = xb[((long long)gc * a.H + gy) * a.W + gx];
            } else if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
                v = xb[((long long)gc * a.H + gy) * a.W + gx];
            }
        }
        Xs[idx] = v;
    }
}

template <int NCT, bool FUSE>
__global__ __launch_bounds__(CV_NT, 2) void k_conv_mfma(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    constexpr int WP = NCT * 32 + 1;  // cout pitch (+1: conflict-free transposing writes)
    const int taps = a.k * a.k;
    float* Xs = smem_f;
    float* Ws = smem_f + a.CK * a.PH * a.PW;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int tile = blockIdx.x;
    const int ty0 = tile / a.tiles_x;
    const int h0 = ty0 * CV_TH, w0 = (tile - ty0 * a.tiles_x) * CV_TW;
    const int co0 = blockIdx.y * (NCT * 32);
    const int b = blockIdx.z;
    const float* xb = a.x + (long long)b * a.Cin * a.H * a.W;

    f32x16 acc[NCT][2];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][pt][r] = 0.f;

    for (int c0 = 0; c0 < a.Cin; c0 += a.CK) {
        __syncthreads();
        stage_x(Xs, a, xb, c0, h0, w0);
        {
            const int per_co = a.CK * taps;
            const int total = NCT * 32 * per_co;
            for (int idx = threadIdx.x; idx < total; idx += CV_NT) {
                const int col = idx / per_co;
                const int rem = idx - col * per_co;  // = ci*taps + tap
                const int ci = rem / taps;
                const int gco = co0 + col, gc = c0 + ci;
                float v = 0.f;
                if (gco < a.Cout && gc < a.Cin) v = a.w[((long long)gco * a.Cin + c0) * taps + rem];
                Ws[rem * WP + col] = v;
            }
        }
        __syncthreads();
        for (int tap = 0; tap < taps; ++tap) {
            const int ky = tap / a.k, kx = tap - ky * a.k;
            const float* xrow = Xs + (wave * 2 + ky * a.dil) * a.PW + kx * a.dil + l31;
            for (int cp = 0; cp < a.CK; cp += 2) {
                const int ci = cp + lhi;
                const float* wp = Ws + (ci * taps + tap) * WP + l31;
                const float* xp = xrow + ci * a.PH * a.PW;
                float av[NCT], bv[2];
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct) av[ct] = wp[ct * 32];
                bv[0] = xp[0];
                bv[1] = xp[a.PW];
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int pt = 0; pt < 2; ++pt)
                        acc[ct][pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[ct], bv[pt], acc[ct][pt], 0, 0, 0);
            }
        }
    }

    const int ox = w0 + l31;
    const long long plane = (long long)a.H * a.W;

    if (!FUSE) {
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                const int oy = h0 + wave * 2 + pt;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = co0 + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    if (co < a.Cout && oy < a.H && ox < a.W) {
                        const long long o = ((long long)b * a.Cout + co) * plane + (long long)oy * a.W + ox;
                        float v = acc[ct][pt][r];
                        if (a.bias) v += a.bias[co];
                        if (a.hh) v += a.hh[co] * (a.hprev ? a.hprev[o] : 0.f);
                        a.y[o] = act_apply(v, a.act, a.slope);
                    }
                }
            }
        return;
    }

    // ---- fused IndRNN: g = ReLU(conv + b_conv) stays in registers and feeds the 1x1 GEMM as the B operand ----------
    constexpr int F = NCT * 32;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            const float bc = a.bias ? a.bias[co] : 0.f;
#pragma unroll
            for (int pt = 0; pt < 2; ++pt) {
                const float v = acc[ct][pt][r] + bc;
                acc[ct][pt][r] = v > 0.f ? v : 0.f;
            }
        }
    __syncthreads();  // all waves are done reading Xs / Ws
    constexpr int WP2 = F + 1;
    float* Wi = smem_f;  // [c][co2], pitch WP2
    for (int idx = threadIdx.x; idx < F * F; idx += CV_NT) {
        const int co2 = idx / F, c = idx - co2 * F;
        Wi[c * WP2 + co2] = a.w_ih[idx];
    }
    __syncthreads();
    f32x16 acc2[NCT][2];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[ct][pt][r] = 0.f;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            // the channel this lane's register r of tile ct holds (C/D layout of the 32x32 MFMA)
            const int c = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            float a2[NCT];
#pragma unroll
            for (int ct2 = 0; ct2 < NCT; ++ct2) a2[ct2] = Wi[c * WP2 + ct2 * 32 + l31];
#pragma unroll
            for (int ct2 = 0; ct2 < NCT; ++ct2)
#pragma unroll
                for (int pt = 0; pt < 2; ++pt)
                    acc2[ct2][pt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[ct2], acc[ct][pt][r], acc2[ct2][pt], 0, 0, 0);
        }
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int pt = 0; pt < 2; ++pt) {
            const int oy = h0 + wave * 2 + pt;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (oy < a.H && ox < a.W) {
                    const long long o = ((long long)b * F + co) * plane + (long long)oy * a.W + ox;
                    float v = acc2[ct][pt][r];
                    if (a.b_ih) v += a.b_ih[co];
                    if (a.hprev) v += a.hh[co] * a.hprev[o];
                    a.y[o] = v > 0.f ? v : 0.f;
                }
            }
        }
}

// ---- small-Cout conv on the vector ALUs (final RIM layer, U-Net 1x1 head) ----------------------------------------
struct SmallArgs {
    const float* x;
    const float* w;
    const float* bias;
    const float* eta;  // MODE 1: [B,H,W,CO] added to the conv output
    float* y;
    int B, Cin, H, W, k, dil, pad, pad_mode;
    int tiles_x, CK, PH, PW;
};
// MODE 0: y[B,CO,H,W] = conv + bias ; MODE 1: y[B,H,W,CO] = eta + conv (+ bias)   (rim_block.py:239-248)
template <int CO, int MODE>
__global__ __launch_bounds__(CV_NT) void k_conv_small(SmallArgs s) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    float* Xs = smem_f;
    const int taps = s.k * s.k;
    const int tile = blockIdx.x;
    const int ty0 = tile / s.tiles_x;
    const int h0 = ty0 * CV_TH, w0 = (tile - ty0 * s.tiles_x) * CV_TW;
    const int b = blockIdx.z;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    ConvArgs a;  // reuse the tile loader
    a.Cin = s.Cin;
    a.H = s.H;
    a.W = s.W;
    a.pad = s.pad;
    a.pad_mode = s.pad_mode;
    a.CK = s.CK;
    a.PH = s.PH;
    a.PW = s.PW;
    const float* xb = s.x + (long long)b * s.Cin * s.H * s.W;
    float acc[CO];
#pragma unroll
    for (int o = 0; o < CO; ++o) acc[o] = 0.f;
    for (int c0 = 0; c0 < s.Cin; c0 += s.CK) {
        __syncthreads();
        stage_x(Xs, a, xb, c0, h0, w0);
        __syncthreads();
        const int nc = min(s.CK, s.Cin - c0);
        for (int ci = 0; ci < nc; ++ci) {
            const float* xp = Xs + (ci * s.PH + ty) * s.PW + tx;
            const float* wp = s.w + (long long)(c0 + ci) * taps;  // w[o][cin][tap]: + o*Cin*taps (wave-uniform -> scalar loads)
            for (int ky = 0; ky < s.k; ++ky)
                for (int kx = 0; kx < s.k; ++kx) {
                    const float xv = xp[ky * s.dil * s.PW + kx * s.dil];
#pragma unroll
                    for (int o = 0; o < CO; ++o) acc[o] += wp[(long long)o * s.Cin * taps + ky * s.k + kx] * xv;
                }
        }
    }
    const int oy = h0 + ty, ox = w0 + tx;
    if (oy >= s.H || ox >= s.W) return;
    if (MODE == 0) {
#pragma unroll
        for (int o = 0; o < CO; ++o)
            s.y[(((long long)b * CO + o) * s.H + oy) * s.W + ox] = acc[o] + (s.bias ? s.bias[o] : 0.f);
    } else {
        const long long base = (((long long)b * s.H + oy) * s.W + ox) * CO;
#pragma unroll
        for (int o = 0; o < CO; ++o) s.y[base + o] = s.eta[base + o] + (acc[o] + (s.bias ? s.bias[o] : 0.f));
    }
}

// ---- small-Cout conv, four pixels per thread (K = 3 or 5, dilation 1; the data gradient of the first RIM layer: 64 -> 4, 5x5) ----
// Workgroup = 8 waves on an 8 x 32 pixel tile.  A lane owns 4 consecutive pixels of a row and all CO outputs; the eight waves
// split the input channels of a chunk (partial sums meet in LDS at the end), so the grid keeps 8 waves per 256 pixels (small images stay latency-tolerant).  Per
// channel a lane reads its K x 12 patch window from LDS as float4s (15 reads for K = 5) and issues 4 * K * K * CO FMAs with the
// weights in scalar registers (wave-uniform loads): 1 LDS access per ~27 FMAs instead of 1 per CO.
#define P4_NT 512
#define P4_TH 8
#define P4_TW 32
#define P4_CK 32
template <int CO, int K>
__global__ __launch_bounds__(P4_NT) void k_conv_small_px4(SmallArgs s) {
    constexpr int PAD = (K - 1) / 2, PH = P4_TH + 2 * PAD, XS = P4_TW + 8;  // tile columns [w0 - 4, w0 + 36): float4-aligned rows
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    float* Xs = smem_f;  // [P4_CK][PH][XS]; afterwards the partial-sum exchange [7][64][CO * 4]
    const int tile = blockIdx.x, ty0 = tile / s.tiles_x;
    const int h0 = ty0 * P4_TH, w0 = (tile - ty0 * s.tiles_x) * P4_TW;
    const int b = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tx = lane & 7, ty = lane >> 3;
    const long long plane = (long long)s.H * s.W;
    const float* xb = s.x + (long long)b * s.Cin * plane;
    const bool rep = s.pad_mode == MRX_PAD_REPLICATE;
    float acc[CO][4];
#pragma unroll
    for (int o = 0; o < CO; ++o)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[o][i] = 0.f;
    for (int c0 = 0; c0 < s.Cin; c0 += P4_CK) {
        __syncthreads();
        for (int idx = threadIdx.x; idx < P4_CK * PH * XS; idx += P4_NT) {
            const int ci = idx / (PH * XS), rem = idx - ci * (PH * XS), r = rem / XS, c = rem - r * XS;
            int gy = h0 + r - PAD, gx = w0 + c - 4;
            float v = 0.f;
            if (c0 + ci < s.Cin) {
                if (rep) {
                    gy = gy < 0 ? 0 : (gy >= s.H ? s.H - 1 : gy);
                    gx = gx < 0 ? 0 : (gx >= s.W ? s.W - 1 : gx);
                    v = xb[(long long)(c0 + ci) * plane + (long long)gy * s.W + gx];
                } else if (gy >= 0 && gy < s.H && gx >= 0 && gx < s.W) {
                    v = xb[(long long)(c0 + ci) * plane + (long long)gy * s.W + gx];
                }
            }
            Xs[idx] = v;
        }
        __syncthreads();
        for (int cj = wave; cj < P4_CK && c0 + cj < s.Cin; cj += P4_NT / 64) {  // this wave's channels of the chunk
            const float* xp = Xs + (cj * PH + ty) * XS + 4 * tx;  // tile column 4 tx = image column w0 + 4 tx - 4
            const float* wp = s.w + (long long)(c0 + cj) * (K * K);  // w[o][cin][tap]: + o * Cin * K * K (wave-uniform -> scalar loads)
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                const float4 v0 = *reinterpret_cast<const float4*>(xp + ky * XS);
                const float4 v1 = *reinterpret_cast<const float4*>(xp + ky * XS + 4);
                const float4 v2 = *reinterpret_cast<const float4*>(xp + ky * XS + 8);
                const float v[12] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w};
#pragma unroll
                for (int kx = 0; kx < K; ++kx)
#pragma unroll
                    for (int o = 0; o < CO; ++o) {
                        const float wv = wp[(long long)o * s.Cin * (K * K) + ky * K + kx];
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[o][i] += wv * v[4 - PAD + kx + i];  // output pixel i, tap kx: column 4 + i + kx - PAD
                    }
            }
        }
    }
    __syncthreads();  // tiles consumed: LDS becomes the partial-sum exchange
    float* Rx = smem_f;
    if (wave > 0) {
#pragma unroll
        for (int o = 0; o < CO; ++o)
#pragma unroll
            for (int i = 0; i < 4; ++i) Rx[(((wave - 1) * 64 + lane) * CO + o) * 4 + i] = acc[o][i];
    }
    __syncthreads();
    if (wave > 0) return;
    const int oy = h0 + ty, ox = w0 + 4 * tx;
    if (oy >= s.H) return;
#pragma unroll
    for (int o = 0; o < CO; ++o) {
        const float bv = s.bias ? s.bias[o] : 0.f;
        float* yo = s.y + (((long long)b * CO + o) * s.H + oy) * s.W + ox;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v = acc[o][i];
#pragma unroll
            for (int wv = 0; wv < P4_NT / 64 - 1; ++wv) v += Rx[((wv * 64 + lane) * CO + o) * 4 + i];
            if (ox + i < s.W) yo[i] = v + bv;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------
static int conv_geometry(int Cin, int k, int dil, int wp, int* CK, int* PH, int* PW, int* pad, size_t* lds) {
    MRX_REQUIRE(k >= 1 && (k & 1) == 1, MRX_EUNSUP, "conv: kernel size %d must be odd", k);
    MRX_REQUIRE(dil >= 1, MRX_EINVAL, "conv: dilation %d", dil);
    *pad = dil * (k - 1) / 2;
    *PH = CV_TH + 2 * *pad;
    *PW = CV_TW + 2 * *pad;
    int ck = 16;
    const int cin_even = (Cin + 1) & ~1;
    if (ck > cin_even) ck = cin_even;
    for (;;) {
        const size_t bytes = sizeof(float) * ((size_t)ck * *PH * *PW + (size_t)ck * k * k * wp);
        if (bytes <= 40 * 1024 || ck <= 2) {
            *lds = bytes;
            break;
        }
        ck = (ck / 2 + 1) & ~1;
        if (ck < 2) ck = 2;
    }
    *CK = ck;
    MRX_REQUIRE(*lds <= 160 * 1024, MRX_EUNSUP, "conv: tile needs %zu bytes of LDS (k=%d dil=%d)", *lds, k, dil);
    return MRX_OK;
}

template <typename K>
static int conv_set_lds(K kern, size_t bytes) {
    static std::mutex mu;
    static std::unordered_map<const void*, size_t> done;
    if (bytes > 48 * 1024) {
        std::lock_guard<std::mutex> lk(mu);
        size_t& cur = done[(const void*)kern];
        if (cur < bytes) {
            MRX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
            cur = bytes;
        }
    }
    return MRX_OK;
}

static int launch_conv(const ConvArgs& a0, int fuse, hipStream_t st) {
    ConvArgs a = a0;
    const int nct = (fuse || a.Cout > 32) ? 2 : 1;
    const int wp = nct * 32 + 1;
    size_t lds;
    int rc = conv_geometry(a.Cin, a.k, a.dil, wp, &a.CK, &a.PH, &a.PW, &a.pad, &lds);
    if (rc) return rc;
    if (fuse) {
        const size_t l2 = sizeof(float) * (size_t)a.Cout * (a.Cout + 1);
        if (l2 > lds) lds = l2;
    }
    a.tiles_x = mrx_cdiv(a.W, CV_TW);
    const int tiles_y = mrx_cdiv(a.H, CV_TH);
    MRX_REQUIRE(a.B <= 65535, MRX_EUNSUP, "conv: batch %d too large", a.B);
    dim3 grid(a.tiles_x * tiles_y, mrx_cdiv(a.Cout, nct * 32), a.B);
    if (fuse) {
        if (a.Cout == 64) {
            if ((rc = conv_set_lds(k_conv_mfma<2, true>, lds))) return rc;
            hipLaunchKernelGGL((k_conv_mfma<2, true>), grid, dim3(CV_NT), lds, st, a);
        } else {
            if ((rc = conv_set_lds(k_conv_mfma<1, true>, lds))) return rc;
            hipLaunchKernelGGL((k_conv_mfma<1, true>), grid, dim3(CV_NT), lds, st, a);
        }
    } else if (nct == 2) {
        if ((rc = conv_set_lds(k_conv_mfma<2, false>, lds))) return rc;
        hipLaunchKernelGGL((k_conv_mfma<2, false>), grid, dim3(CV_NT), lds, st, a);
    } else {
        if ((rc = conv_set_lds(k_conv_mfma<1, false>, lds))) return rc;
        hipLaunchKernelGGL((k_conv_mfma<1, false>), grid, dim3(CV_NT), lds, st, a);
    }
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

static int conv_common_checks(const char* who, int B, int Cin, int Cout, int H, int W) {
    MRX_REQUIRE(B >= 0 && Cin >= 1 && Cout >= 1 && H >= 1 && W >= 1, MRX_EINVAL, "%s: bad dims B=%d Cin=%d Cout=%d H=%d W=%d",
                who, B, Cin, Cout, H, W);
    return MRX_OK;
}

static int launch_small(const float* x, const float* w, const float* bias, const float* eta, float* y, int B, int Cin,
                        int Cout, int H, int W, int k, int dil, int pad_mode, int mode, hipStream_t st) {
    SmallArgs s;
    s.x = x;
    s.w = w;
    s.bias = bias;
    s.eta = eta;
    s.y = y;
    s.B = B;
    s.Cin = Cin;
    s.H = H;
    s.W = W;
    s.k = k;
    s.dil = dil;
    s.pad_mode = pad_mode;
    size_t lds;
    int rc = conv_geometry(Cin, k, dil, 0, &s.CK, &s.PH, &s.PW, &s.pad, &lds);
    if (rc) return rc;
    if (mode == 0 && dil == 1 && (k == 3 || k == 5) && (Cout == 2 || Cout == 4) && Cin >= 8) {
        s.tiles_x = mrx_cdiv(W, P4_TW);
        dim3 g4(s.tiles_x * mrx_cdiv(H, P4_TH), 1, B);
        size_t lds4 = sizeof(float) * P4_CK * (P4_TH + k - 1) * (P4_TW + 8);
        const size_t xch = sizeof(float) * (P4_NT / 64 - 1) * 64 * Cout * 4;  // partial-sum exchange of the waves
        if (lds4 < xch) lds4 = xch;
#define PX4_CASE(CO, KK)                                                                   \
    do {                                                                                   \
        if ((rc = conv_set_lds(k_conv_small_px4<CO, KK>, lds4))) return rc;                \
        hipLaunchKernelGGL((k_conv_small_px4<CO, KK>), g4, dim3(P4_NT), lds4, st, s);      \
    } while (0)
        if (Cout == 2 && k == 3) PX4_CASE(2, 3);
        else if (Cout == 2) PX4_CASE(2, 5);
        else if (k == 3) PX4_CASE(4, 3);
        else PX4_CASE(4, 5);
#undef PX4_CASE
        MRX_LAUNCH_CHECK();
        return MRX_OK;
    }
    s.tiles_x = mrx_cdiv(W, CV_TW);
    dim3 grid(s.tiles_x * mrx_cdiv(H, CV_TH), 1, B);
#define SMALL_CASE(CO, MODE)                                                             \
    do {                                                                                 \
        if ((rc = conv_set_lds(k_conv_small<CO, MODE>, lds))) return rc;                 \
        hipLaunchKernelGGL((k_conv_small<CO, MODE>), grid, dim3(CV_NT), lds, st, s);     \
    } while (0)
    if (mode == 1) {
        MRX_REQUIRE(Cout == 2, MRX_EUNSUP, "rim_final: Cout must be 2 (got %d)", Cout);
        SMALL_CASE(2, 1);
    } else if (Cout == 1) {
        SMALL_CASE(1, 0);
    } else if (Cout == 2) {
        SMALL_CASE(2, 0);
    } else if (Cout == 3) {
        SMALL_CASE(3, 0);
    } else {
        SMALL_CASE(4, 0);
    }
#undef SMALL_CASE
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- tuned 3x3, dilation 1 (NormUnet: unet_block.py:251,255) --------------------------------------------------------------
// Small channel counts (2 / 14 / 28 / 56) make these convolutions memory- and latency-bound, not MFMA-bound: the tile is built
// from 16-wide MFMA blocks (v_mfma_f32_16x16x4_f32: 16 couts x 16 pixels x 4 input channels = one tap of one channel group), so a
// 14-channel layer wastes 1/8 of the block instead of 9/16 of a 32-wide one; the raw halo'd tile arrives by LDS-DMA (one dword per
// lane; lanes outside the image are masked off and keep the zeros -- or, for replicate padding, fetch the clamped element) into
// double-buffered planes; the weights of the next channel group are fetched into registers while the matrix pipe works on the
// current one.  256 threads = 4 waves, 8 x 32 output pixels; wave w: rows 2w, 2w+1 (4 pixel blocks) x NCOT cout blocks.
typedef float c3_f4 __attribute__((ext_vector_type(4)));
#define C3_PW 34
#define C3_PLANE 368   // 10 rows x 34 used (340); 368 = 16 mod 32: the four channel planes of an MFMA land on different banks
#define C3_XBUF (4 * C3_PLANE)
template <int NCOT, bool STATS>
__global__ __launch_bounds__(CV_NT, NCOT >= 4 ? 3 : 4) void k_conv3x3_t(ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    float* Xs = smem_f;                   // [2][4][C3_PLANE]
    float* Ws = smem_f + 2 * C3_XBUF;     // [2][9][NCOT][64]: MFMA A operand per lane
    constexpr int WBUF = 9 * NCOT * 64;
    constexpr int NWL = (WBUF + CV_NT - 1) / CV_NT;  // weight values staged per thread and chunk
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int tile = blockIdx.x;
    const int ty0 = tile / a.tiles_x;
    const int h0 = ty0 * CV_TH, w0 = (tile - ty0 * a.tiles_x) * CV_TW;
    const int b = blockIdx.z;
    const int co0 = blockIdx.y * (NCOT * 16);  // small images: the cout blocks are spread over workgroups (grid.y) to fill the chip
    const long long plane = (long long)a.H * a.W;
    const float* xb = a.x + (long long)b * a.Cin * plane;
    const int nchunks = (a.Cin + 3) / 4;
    for (int i = tid; i < 2 * C3_XBUF; i += CV_NT) Xs[i] = 0.f;  // zero padding = the slots no copy ever writes
    // raw-tile slots of this wave: channel `wave` of the chunk, 6 copies of 64 elements
    unsigned xoff[6];
    unsigned xok = 0;
#pragma unroll
    for (int m = 0; m < 6; ++m) {
        const int sl = m * 64 + lane;
        const int ry = sl / C3_PW, rx = sl - ry * C3_PW;
        int gy = h0 + ry - 1, gx = w0 + rx - 1;
        bool ok = sl < 10 * C3_PW;
        if (a.pad_mode == MRX_PAD_REPLICATE) {
            gy = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy);
            gx = gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx);
        } else {
            ok = ok && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        }
        xoff[m] = ok ? (unsigned)(gy * a.W + gx) * 4u : 0u;
        xok |= (ok ? 1u : 0u) << m;
    }
    auto dma_x = [&](int q) {
        int gc = 4 * q + wave;
        gc = gc < a.Cin ? gc : a.Cin - 1;  // channels past Cin repeat the last one; their weights are staged as zeros
        const char* src = reinterpret_cast<const char*>(xb + (long long)gc * plane);
        float* dst = Xs + (q & 1) * C3_XBUF + wave * C3_PLANE;
#pragma unroll
        for (int m = 0; m < 6; ++m)
            if ((xok >> m) & 1u)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + xoff[m]),
                                                 (__attribute__((address_space(3))) void*)(dst + m * 64), 4, 0, 0);
    };
    // weight staging: element i of the chunk image = (tap, ct, lane') -> w[16 ct + (lane' & 15)][4 q + (lane' >> 4)][tap]
    float wr[NWL];
    auto load_w = [&](int q) {
#pragma unroll
        for (int j = 0; j < NWL; ++j) {
            const int i = tid + j * CV_NT;
            const int ln = i & 63, ct = (i >> 6) % NCOT, tap = (i >> 6) / NCOT;
            const int co = co0 + 16 * ct + (ln & 15), ci = 4 * q + (ln >> 4);
            wr[j] = (i < WBUF && co < a.Cout && ci < a.Cin) ? a.w[((long long)co * a.Cin + ci) * 9 + tap] : 0.f;
        }
    };
    auto store_w = [&](int q) {
#pragma unroll
        for (int j = 0; j < NWL; ++j) {
            const int i = tid + j * CV_NT;
            if (i < WBUF) Ws[(q & 1) * WBUF + i] = wr[j];
        }
    };
    c3_f4 acc[4][NCOT];
#pragma unroll
    for (int sg = 0; sg < 4; ++sg)
#pragma unroll
        for (int ct = 0; ct < NCOT; ++ct) acc[sg][ct] = (c3_f4){0.f, 0.f, 0.f, 0.f};
    load_w(0);
    __syncthreads();  // tile zeroed before any copy lands
    dma_x(0);
    store_w(0);
    for (int q = 0; q < nchunks; ++q) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // chunk q staged (tile + weights); buffers of chunk q - 1 free
        if (q + 1 < nchunks) {
            dma_x(q + 1);
            load_w(q + 1);
        }
        const float* xq = Xs + (q & 1) * C3_XBUF + lg * C3_PLANE + l15;
        const float* wq = Ws + (q & 1) * WBUF + lane;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap % 3;
            float av[NCOT];
#pragma unroll
            for (int ct = 0; ct < NCOT; ++ct) av[ct] = wq[(tap * NCOT + ct) * 64];
#pragma unroll
            for (int sg = 0; sg < 4; ++sg) {
                const float bv = xq[(2 * wave + (sg >> 1) + ky) * C3_PW + (sg & 1) * 16 + kx];
#pragma unroll
                for (int ct = 0; ct < NCOT; ++ct) acc[sg][ct] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ct], bv, acc[sg][ct], 0, 0, 0);
            }
        }
        if (q + 1 < nchunks) store_w(q + 1);
    }
    bool ok[4];
#pragma unroll
    for (int sg = 0; sg < 4; ++sg) {
        const int oy = h0 + 2 * wave + (sg >> 1), ox = w0 + (sg & 1) * 16 + l15;
        ok[sg] = oy < a.H && ox < a.W;
        if (ok[sg]) {
#pragma unroll
            for (int ct = 0; ct < NCOT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + 16 * ct + 4 * lg + r;
                    if (co < a.Cout) {
                        float v = acc[sg][ct][r];
                        if (a.bias) v += a.bias[co];
                        a.y[((long long)b * a.Cout + co) * plane + (long long)oy * a.W + ox] = act_apply(v, a.act, a.slope);
                    }
                }
        }
    }
    if (STATS) {
        // InstanceNorm statistics of this tile, per cout: mean over its valid pixels, then the sum of squared deviations from that
        // mean (two exact-order block reductions); k_conv_stats_finalize merges the tiles with the parallel-variance formula.
        __syncthreads();  // the staging buffers become the reduction scratch
        float* red = smem_f;  // [4 waves][NCOT * 16]
        const int nrow = a.H - h0 < CV_TH ? a.H - h0 : CV_TH, ncol = a.W - w0 < CV_TW ? a.W - w0 : CV_TW;
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
                        const float v = acc[sg][ct][r] + ((a.bias && co0 + 16 * ct + 4 * lg + r < a.Cout) ? a.bias[co0 + 16 * ct + 4 * lg + r] : 0.f);
                        const float d = pass == 0 ? v : (v - mean[ct][r]) * (v - mean[ct][r]);
                        t += ok[sg] ? d : 0.f;
                    }
                    for (int off = 8; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);  // over the 16 pixels lanes of this channel group
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
                        float* o = a.tstats + ((((long long)b * gridDim.x + tile) * a.Cout) + co0 + c) * 2;
                        o[0] = mean[ct][r];
                        o[1] = t;
                    }
                }
            __syncthreads();
        }
    }
}

// merge the per-tile (mean, M2) of k_conv3x3_t<., true> into per-plane (mean, M2): one wave per (b, cout), Chan et al. pairwise
// updates in double; tile sizes follow from the tile index
__global__ __launch_bounds__(64) void k_conv_stats_finalize(const float* __restrict__ tstats, float* __restrict__ stats, int ntiles,
                                                           int tiles_x, int Cout, int H, int W) {
    const int plane_id = blockIdx.x;  // b * Cout + co
    const int b = plane_id / Cout, co = plane_id - b * Cout;
    double n = 0.0, mean = 0.0, m2 = 0.0;
    for (int t = threadIdx.x; t < ntiles; t += 64) {
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        const int nr = H - ty * CV_TH < CV_TH ? H - ty * CV_TH : CV_TH, nc = W - tx * CV_TW < CV_TW ? W - tx * CV_TW : CV_TW;
        const double nb = (double)(nr * nc);
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

template <int NCOT>
static int launch_conv3x3_t(ConvArgs a, hipStream_t st, float* stats = nullptr) {
    a.tiles_x = mrx_cdiv(a.W, CV_TW);
    const int ntiles = a.tiles_x * mrx_cdiv(a.H, CV_TH);
    const int gy = mrx_cdiv(a.Cout, NCOT * 16);
    constexpr size_t lds = sizeof(float) * (2 * C3_XBUF + 2 * 9 * NCOT * 64);
    static_assert(lds <= 48 * 1024, "fits the default dynamic LDS limit");
    if (a.tstats) {
        hipLaunchKernelGGL((k_conv3x3_t<NCOT, true>), dim3(ntiles, gy, a.B), dim3(CV_NT), lds, st, a);
        if (stats)  // (stats == nullptr: the caller merges the tile statistics itself -- mrx_instance_norm_apply_tiles)
            hipLaunchKernelGGL(k_conv_stats_finalize, dim3(a.B * a.Cout), dim3(64), 0, st, (const float*)a.tstats, stats, ntiles,
                               a.tiles_x, a.Cout, a.H, a.W);
    } else {
        hipLaunchKernelGGL((k_conv3x3_t<NCOT, false>), dim3(ntiles, gy, a.B), dim3(CV_NT), lds, st, a);
    }
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
static bool conv3x3_tuned_ok(int B, int Cout, int H, int W, int k, int dil) {
    // any Cout: cout blocks of 16 * NCOT channels are spread over grid.y (the (18, 4) U-Net of the reference yaml reaches 288 channels)
    return k == 3 && dil == 1 && Cout <= 16 * 65535 && (long long)H * W < (1ll << 30) && B <= 65535;
}
static int dispatch_conv3x3_t(const ConvArgs& a, hipStream_t st, float* stats) {
    const int ncot = (a.Cout + 15) / 16;
    // few tiles (the pooled levels of the U-Net): one cout block per workgroup, cout blocks across grid.y, so ~4x the workgroups
    const long long tiles = (long long)mrx_cdiv(a.W, CV_TW) * mrx_cdiv(a.H, CV_TH) * a.B;
    if (ncot == 1 || tiles < 1024) return launch_conv3x3_t<1>(a, st, stats);
    if (ncot == 2) return launch_conv3x3_t<2>(a, st, stats);
    if (ncot == 3) return launch_conv3x3_t<3>(a, st, stats);
    return launch_conv3x3_t<4>(a, st, stats);
}

extern "C" int mrx_conv2d(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, int H,
                          int W, int k, int dil, int pad_mode, int act, float slope, void* stream) {
    MRX_REQUIRE(x && w && y, MRX_EINVAL, "mrx_conv2d: null pointer");
    int rc = conv_common_checks("mrx_conv2d", B, Cin, Cout, H, W);
    if (rc) return rc;
    MRX_REQUIRE(pad_mode == MRX_PAD_ZERO || pad_mode == MRX_PAD_REPLICATE, MRX_EINVAL, "mrx_conv2d: bad pad mode %d", pad_mode);
    MRX_REQUIRE(act >= 0 && act <= 2, MRX_EINVAL, "mrx_conv2d: bad activation %d", act);
    if (B == 0) return MRX_OK;
    if (Cout <= 4 && act == MRX_ACT_NONE)
        return launch_small(x, w, bias, nullptr, y, B, Cin, Cout, H, W, k, dil, pad_mode, 0, (hipStream_t)stream);
    ConvArgs a = {};
    a.x = x;
    a.w = w;
    a.bias = bias;
    a.y = y;
    a.B = B;
    a.Cin = Cin;
    a.Cout = Cout;
    a.H = H;
    a.W = W;
    a.k = k;
    a.dil = dil;
    a.pad_mode = pad_mode;
    a.act = act;
    a.slope = slope;
    if (conv3x3_tuned_ok(B, Cout, H, W, k, dil)) return dispatch_conv3x3_t(a, (hipStream_t)stream, nullptr);
    return launch_conv(a, 0, (hipStream_t)stream);
}

// conv (no activation) + the InstanceNorm statistics of its output in one pass over the accumulators (unet_block.py:251-253):
// stats[b][co] = (mean, sum of squared deviations) over the H x W plane.  work: mrx_conv2d_stats_work_floats() floats.
// Only the tuned 3x3 shapes; returns MRX_EUNSUP (no error text) otherwise so the caller takes mrx_conv2d + mrx_instance_norm_act.
extern "C" int64_t mrx_conv2d_stats_work_floats(int B, int Cout, int H, int W) {
    if (B < 0 || Cout < 1 || H < 1 || W < 1) return -1;
    return (int64_t)B * mrx_cdiv(W, CV_TW) * mrx_cdiv(H, CV_TH) * Cout * 2;
}
extern "C" int mrx_conv2d_stats_supported(int B, int Cout, int H, int W, int k, int dil) { return conv3x3_tuned_ok(B, Cout, H, W, k, dil) ? 1 : 0; }
extern "C" int mrx_conv2d_stats(const float* x, const float* w, const float* bias, float* y, float* stats, float* work, int B, int Cin,
                                int Cout, int H, int W, int k, int dil, int pad_mode, void* stream) {
    MRX_REQUIRE(x && w && y && work, MRX_EINVAL, "mrx_conv2d_stats: null pointer");
    int rc = conv_common_checks("mrx_conv2d_stats", B, Cin, Cout, H, W);
    if (rc) return rc;
    MRX_REQUIRE(pad_mode == MRX_PAD_ZERO || pad_mode == MRX_PAD_REPLICATE, MRX_EINVAL, "mrx_conv2d_stats: bad pad mode %d", pad_mode);
    MRX_REQUIRE(conv3x3_tuned_ok(B, Cout, H, W, k, dil), MRX_EUNSUP, "mrx_conv2d_stats: only 3x3, dilation 1 (got k=%d dil=%d Cout=%d)",
                k, dil, Cout);
    if (B == 0) return MRX_OK;
    ConvArgs a = {};
    a.x = x;
    a.w = w;
    a.bias = bias;
    a.y = y;
    a.tstats = work;
    a.B = B;
    a.Cin = Cin;
    a.Cout = Cout;
    a.H = H;
    a.W = W;
    a.k = k;
    a.dil = dil;
    a.pad_mode = pad_mode;
    a.act = MRX_ACT_NONE;
    return dispatch_conv3x3_t(a, (hipStream_t)stream, stats);
}

extern "C" int mrx_indrnn_cell(const float* x, const float* w_ih, const float* b_ih, const float* hh, const float* h_prev,
                               float* h_new, int B, int Cin, int F, int H, int W, int k, int dil, void* stream) {
    MRX_REQUIRE(x && w_ih && hh && h_new, MRX_EINVAL, "mrx_indrnn_cell: null pointer");
    int rc = conv_common_checks("mrx_indrnn_cell", B, Cin, F, H, W);
    if (rc) return rc;
    if (B == 0) return MRX_OK;
    ConvArgs a = {};
    a.x = x;
    a.w = w_ih;
    a.bias = b_ih;
    a.y = h_new;
    a.hh = hh;
    a.hprev = h_prev;
    a.B = B;
    a.Cin = Cin;
    a.Cout = F;
    a.H = H;
    a.W = W;
    a.k = k;
    a.dil = dil;
    a.pad_mode = MRX_PAD_ZERO;  // rnn_cells.py:299: the ih conv zero-pads
    a.act = MRX_ACT_RELU;
    return launch_conv(a, 0, (hipStream_t)stream);
}

extern "C" int mrx_rim_layer_indrnn(const float* x, const float* w_conv, const float* b_conv, const float* w_ih,
                                    const float* b_ih, const float* hh, const float* h_prev, float* h_new, int B, int Cin,
                                    int F, int H, int W, int k, int dil, void* stream) {
    MRX_REQUIRE(x && w_conv && w_ih && hh && h_new, MRX_EINVAL, "mrx_rim_layer_indrnn: null pointer");
    int rc = conv_common_checks("mrx_rim_layer_indrnn", B, Cin, F, H, W);
    if (rc) return rc;
    MRX_REQUIRE(F == 32 || F == 64, MRX_EUNSUP, "mrx_rim_layer_indrnn: hidden size %d not in {32, 64}", F);
    if (B == 0) return MRX_OK;
    ConvArgs a = {};
    a.x = x;
    a.w = w_conv;
    a.bias = b_conv;
    a.y = h_new;
    a.hh = hh;
    a.hprev = h_prev;
    a.w_ih = w_ih;
    a.b_ih = b_ih;
    a.B = B;
    a.Cin = Cin;
    a.Cout = F;
    a.H = H;
    a.W = W;
    a.k = k;
    a.dil = dil;
    a.pad_mode = MRX_PAD_REPLICATE;
    a.act = MRX_ACT_RELU;
    return launch_conv(a, 1, (hipStream_t)stream);
}

int mrx_rim_final_tuned(const float* h, const float* w, const float* bias, const float* eta, float* eta_out, int B, int F,
                        int H, int W, int k, int dil, hipStream_t st, int* handled);  // rim_layer.hip

extern "C" int mrx_rim_final(const float* h, const float* w, const float* bias, const float* eta, float* eta_out, int B,
                             int F, int H, int W, int k, int dil, void* stream) {
    MRX_REQUIRE(h && w && eta && eta_out, MRX_EINVAL, "mrx_rim_final: null pointer");
    int rc = conv_common_checks("mrx_rim_final", B, F, 2, H, W);
    if (rc) return rc;
    if (B == 0) return MRX_OK;
    MRX_REQUIRE(B <= 65535, MRX_EUNSUP, "mrx_rim_final: batch %d too large", B);
    int handled = 0;
    rc = mrx_rim_final_tuned(h, w, bias, eta, eta_out, B, F, H, W, k, dil, (hipStream_t)stream, &handled);
    if (handled || rc) return rc;
    return launch_small(h, w, bias, eta, eta_out, B, F, 2, H, W, k, dil, MRX_PAD_REPLICATE, 1, (hipStream_t)stream);
}
