// conv_bwd.hip -- backward kernels of the convolutional regulariser for the training path (the reference
// trains through torch autograd -- cuDNN / MIOpen backward of Conv2d, ReLU and the IndRNN cell, rim_block.py:217-249).
//
//   mrx_conv_wgrad        dW[co][ci][tap] = sum over (b, pixel) of dy[b,co,pixel] * xpad[b,ci,pixel + tap*dil]    (weight gradient)
//                         Cout = 64: fp32 matrix cores (v_mfma_f32_32x32x2_f32), one pass over dy and x;
//                         other Cout: a reduction kernel on the vector ALUs (final RIM layer: Cout = 2).
//   mrx_reppad_fold       adjoint of replicate padding: gradient on the padded domain [H+2p, W+2p] -> [H, W]
//                         (the data gradient itself is a zero-padded 'same' convolution of the zero-extended dy with the flipped,
//                         transposed weights -- the forward kernels, incl. the Winograd one)
//   mrx_relu_bwd          dpre = dy * (y > 0), per-channel sums of dpre (bias gradient) and, for the IndRNN cell, dpre * hh -> dh_prev
//                         and per-channel sums of dpre * h_prev (gradient of hh)
// All reductions are two-stage with a fixed order (per-workgroup partials, then one workgroup per output), so gradients do not
// depend on scheduling.
#include <cstdint>
#include <cstdlib>

#include "mrx_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- weight gradient, Cout = 64, matrix cores --------------------------------------------------------------------------------
// GEMM view: D[co][n] += A[co][pixel] * B[pixel][n], n = ci * taps + tap (the natural [co][ci][ky][kx] order of the weights).
// Workgroup: 512 threads, a 4 x 32 pixel tile per step of a persistent loop.  LDS: the dy tile [64][128 (+1)] and the halo'd x tile
// [Cin][4 + 2 pad][32 + 2 pad (+pad to odd)].  Wave w owns the 32-column blocks w, w + 8, ... of n and both halves of co; one MFMA
// takes two pixels: A = dy (lane = channel, lane half = pixel parity), B = x gathered at the lane's (ci, tap) offset.
#define WG_NT 512
#define WG_TH 4
#define WG_TW 32
#define WG_PX (WG_TH * WG_TW)
#define WG_DYS (WG_PX + 1)
#define WG_MAXNB 3  // 32-column blocks per wave: N <= 8 * 3 * 32 = 768 (64 channels x 9 taps = 576)

struct WgradArgs {
    const float* x;   // [B,Cin,H,W]
    const float* dy;  // [B,64,H,W]
    float* part;      // [gridDim.x][64][N]
    int B, Cin, H, W, k, dil, pad, pad_mode, N, tiles_x, ntiles, XS, PH;
    int vec;  // W % 4 == 0 and 16-byte aligned x / dy: float4 tile loads
};

template <int PAD>
__global__ __launch_bounds__(WG_NT, 2) void k_conv_wgrad64(WgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    constexpr int XS = (WG_TW + 2 * PAD) | 1, PH = WG_TH + 2 * PAD, PLANE = PH * XS, PW = WG_TW + 2 * PAD;  // odd row stride
    float* Dy = smem_f;                 // [64][WG_DYS]
    float* Xs = smem_f + 64 * WG_DYS;   // [Cin][PH][XS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int taps = a.k * a.k;
    const int nblocks = (a.N + 31) / 32;
    // Few column blocks (1x1 layers: 2, the 5x5 layer on 4 channels: 4): the waves that would idle split the pixels instead.
    // groups = 8 / nblocks (a power of two) wave groups each take 64 / groups of the 64 two-pixel steps of a tile and write
    // their own partial; with >= 8 blocks every wave owns blocks w, w + 8, ... and all steps.
    const int groups = nblocks >= 8 ? 1 : (nblocks > 4 ? 1 : (nblocks > 2 ? 2 : (nblocks > 1 ? 4 : 8)));
    const int wblk = groups == 1 ? wave : wave % (8 / groups);          // first column block of this wave
    const int grp = groups == 1 ? 0 : wave / (8 / groups);              // its pixel group
    const int gshift = groups == 1 ? 6 : (groups == 2 ? 5 : (groups == 4 ? 4 : 3));  // steps per group = 1 << gshift
    const int nbw = wblk < nblocks ? (groups == 1 ? (nblocks - wblk + 7) / 8 : 1) : 0;  // column blocks of this wave (wave-uniform)
    // per-lane gather base of every n-block of this wave: (ci, tap) -> ci * PLANE + ky * dil * XS + kx * dil, plus the pixel parity
    int bbase[WG_MAXNB];
#pragma unroll
    for (int j = 0; j < WG_MAXNB; ++j) {
        int n = (wblk + 8 * j) * 32 + l31;
        n = n < a.N ? n : a.N - 1;  // columns past N are computed on a valid address and never stored
        const int ci = n / taps, tap = n - ci * taps, ky = tap / a.k, kx = tap - ky * a.k;
        bbase[j] = ci * PLANE + ky * a.dil * XS + kx * a.dil + lhi;
    }
    f32x16 acc[WG_MAXNB][2];
#pragma unroll
    for (int j = 0; j < WG_MAXNB; ++j)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][ct][r] = 0.f;
    const long long plane = (long long)a.H * a.W;
    const long long nt_total = (long long)a.ntiles * a.B;
    // Tile order: workgroup g runs on XCD g % 8 (one L2 each).  With a grid that is a multiple of 8 every XCD walks its own contiguous
    // band of tiles, so the halo rows two neighbouring tiles share are fetched into one L2 (PMC: FETCH_SIZE was 2.4x the input).
    const bool banded = (gridDim.x & 7) == 0;
    const long long per_xcd = (nt_total + 7) >> 3;
    const long long i_step = banded ? (gridDim.x >> 3) : gridDim.x;
    for (long long i = banded ? (blockIdx.x >> 3) : blockIdx.x; i < (banded ? per_xcd : nt_total); i += i_step) {
        const long long v = banded ? (long long)(blockIdx.x & 7) * per_xcd + i : i;
        if (v >= nt_total) break;
        const int b = (int)(v / a.ntiles), t = (int)(v - (long long)b * a.ntiles);
        const int ty0 = t / a.tiles_x, h0 = ty0 * WG_TH, w0 = (t - ty0 * a.tiles_x) * WG_TW;
        __syncthreads();  // the previous tile's operands are consumed
        // dy tile: zeros outside the image (ragged tiles contribute nothing).  W % 4 == 0 (a.vec): aligned float4 loads
        const float* dyb = a.dy + (long long)b * 64 * plane;
        if (a.vec) {
            for (int i = tid; i < 64 * WG_PX / 4; i += WG_NT) {
                const int co = i >> 5, q = i & 31, r = q >> 3, c = (q & 7) * 4;
                const int gy = h0 + r, gx = w0 + c;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (gy < a.H && gx < a.W) v = *reinterpret_cast<const float4*>(dyb + (long long)co * plane + (long long)gy * a.W + gx);
                float* d = Dy + co * WG_DYS + r * 32 + c;
                d[0] = v.x;
                d[1] = v.y;
                d[2] = v.z;
                d[3] = v.w;
            }
        } else {
            for (int i = tid; i < 64 * WG_PX; i += WG_NT) {
                const int co = i / WG_PX, px = i - co * WG_PX, r = px >> 5, c = px & 31;
                const int gy = h0 + r, gx = w0 + c;
                Dy[co * WG_DYS + px] = (gy < a.H && gx < a.W) ? dyb[(long long)co * plane + (long long)gy * a.W + gx] : 0.f;
            }
        }
        // x tile with its halo: replicate (clamped) or zero border
        const float* xb = a.x + (long long)b * a.Cin * plane;
        const bool rep = a.pad_mode == MRX_PAD_REPLICATE;
        if (a.vec) {
            // interior columns [w0, w0 + 32): 8 aligned float4 per row (rows clamped or zeroed); a group straddling the right image
            // edge cannot exist (W % 4 == 0), a group past it is the replicated last column or zero
            for (int i = tid; i < a.Cin * PH * 8; i += WG_NT) {
                const int ci = i / (PH * 8), rem = i - ci * (PH * 8), r = rem >> 3, c = (rem & 7) * 4;
                int gy = h0 + r - PAD;
                const int gx = w0 + c;
                const bool rok = gy >= 0 && gy < a.H;
                gy = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy);
                const float* row = xb + (long long)ci * plane + (long long)gy * a.W;
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (rep || rok) {
                    if (gx < a.W) v = *reinterpret_cast<const float4*>(row + gx);
                    else if (rep) v.x = v.y = v.z = v.w = row[a.W - 1];
                }
                float* d = Xs + ci * PLANE + r * XS + PAD + c;
                d[0] = v.x;
                d[1] = v.y;
                d[2] = v.z;
                d[3] = v.w;
            }
            if (PAD > 0) {  // the 2 * PAD halo columns of every row
                for (int i = tid; i < a.Cin * PH * 2 * PAD; i += WG_NT) {
                    const int ci = i / (PH * 2 * PAD), rem = i - ci * (PH * 2 * PAD), r = rem / (2 * PAD), j = rem - r * (2 * PAD);
                    const int c = j < PAD ? j : WG_TW + j;  // tile column
                    int gy = h0 + r - PAD, gx = w0 + c - PAD;
                    float vv = 0.f;
                    if (rep) {
                        gy = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy);
                        gx = gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx);
                        vv = xb[(long long)ci * plane + (long long)gy * a.W + gx];
                    } else if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
                        vv = xb[(long long)ci * plane + (long long)gy * a.W + gx];
                    }
                    Xs[ci * PLANE + r * XS + c] = vv;
                }
            }
        } else {
            for (int i = tid; i < a.Cin * PH * PW; i += WG_NT) {
                const int ci = i / (PH * PW), rem = i - ci * (PH * PW), r = rem / PW, c = rem - r * PW;
                int gy = h0 + r - PAD, gx = w0 + c - PAD;
                float vv = 0.f;
                if (rep) {
                    gy = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy);
                    gx = gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx);
                    vv = xb[(long long)ci * plane + (long long)gy * a.W + gx];
                } else if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
                    vv = xb[(long long)ci * plane + (long long)gy * a.W + gx];
                }
                Xs[ci * PLANE + r * XS + c] = vv;
            }
        }
        __syncthreads();
        // 64 steps of two pixels (row s >> 4, columns 2 (s & 15) + parity): the dy operands are shared by the wave's column blocks;
        // all LDS offsets are immediates and the reads run WG_PF steps ahead of the MFMAs that consume them
        const float* ap0 = Dy + l31 * WG_DYS + lhi;
        const float* ap1 = Dy + (32 + l31) * WG_DYS + lhi;
        const float* bp0 = Xs + bbase[0];
        const float* bp1 = Xs + bbase[1];
        const float* bp2 = Xs + bbase[2];
        constexpr int PF = 3;
        float ra0[PF + 1], ra1[PF + 1], rb0[PF + 1], rb1[PF + 1], rb2[PF + 1];
        if (nbw > 0) {
#pragma unroll
            for (int s = 0; s < WG_PX / 2 + PF; ++s) {
                if (s < WG_PX / 2 && (s >> gshift) == grp) {
                    const int w = s % (PF + 1), xo = (s >> 4) * XS + 2 * (s & 15);
                    ra0[w] = ap0[2 * s];
                    ra1[w] = ap1[2 * s];
                    rb0[w] = bp0[xo];
                    if (nbw > 1) rb1[w] = bp1[xo];
                    if (nbw > 2) rb2[w] = bp2[xo];
                }
                if (s >= PF && ((s - PF) >> gshift) == grp) {
                    const int c = (s - PF) % (PF + 1);
                    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra0[c], rb0[c], acc[0][0], 0, 0, 0);
                    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra1[c], rb0[c], acc[0][1], 0, 0, 0);
                    if (nbw > 1) {
                        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra0[c], rb1[c], acc[1][0], 0, 0, 0);
                        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra1[c], rb1[c], acc[1][1], 0, 0, 0);
                    }
                    if (nbw > 2) {
                        acc[2][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra0[c], rb2[c], acc[2][0], 0, 0, 0);
                        acc[2][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra1[c], rb2[c], acc[2][1], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float* pb = a.part + ((long long)blockIdx.x * groups + grp) * 64 * a.N;
#pragma unroll
    for (int j = 0; j < WG_MAXNB; ++j) {
        if (j >= nbw) break;
        const int n = (wblk + 8 * j) * 32 + l31;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (n < a.N) pb[(long long)co * a.N + n] = acc[j][ct][r];
            }
    }
}

// dW[i] (= or +=) sum over the workgroup partials in a fixed order, in double: mrx_reduce_parts (mrx_common.h)
__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ part, int nparts, long long n, float* __restrict__ dw,
                                                      int accumulate) {
    mrx_reduce_parts(part, nparts, n, dw, accumulate);
}

// ---- weight gradient, small Cout (final RIM layer: 2): one workgroup per (input channel, slab of rows), all taps and all Cout at
// once -- every x row is read once per tap row (L1/L2 hits) and dy once per input channel instead of once per (channel, tap).
#define WS_NT 256
#define WS_MAXCO 4
struct WsmallArgs {
    const float* x;
    const float* dy;
    float* part;  // [nslab][Cout][Cin*taps]
    int B, Cin, Cout, H, W, k, dil, pad, pad_mode, rows_per_slab;
};
template <int K>
__global__ __launch_bounds__(WS_NT) void k_conv_wgrad_small(WsmallArgs a) {
    constexpr int TAPS = K * K;
    const int ci = blockIdx.x, slab = blockIdx.y, r0 = slab * a.rows_per_slab;
    const int r1 = r0 + a.rows_per_slab < a.B * a.H ? r0 + a.rows_per_slab : a.B * a.H;  // rows enumerate (b, h)
    const long long plane = (long long)a.H * a.W;
    const bool rep = a.pad_mode == MRX_PAD_REPLICATE;
    float acc[WS_MAXCO][TAPS];
#pragma unroll
    for (int o = 0; o < WS_MAXCO; ++o)
#pragma unroll
        for (int t = 0; t < TAPS; ++t) acc[o][t] = 0.f;
    for (int row = r0; row < r1; ++row) {
        const int b = row / a.H, h = row - b * a.H;
        const float* xc = a.x + ((long long)b * a.Cin + ci) * plane;
        const float* dr = a.dy + (long long)b * a.Cout * plane + (long long)h * a.W;
        for (int w = threadIdx.x; w < a.W; w += WS_NT) {
            float d[WS_MAXCO];
#pragma unroll
            for (int o = 0; o < WS_MAXCO; ++o) d[o] = o < a.Cout ? dr[(long long)o * plane + w] : 0.f;
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                int gy = h + ky * a.dil - a.pad;
                const bool rok = rep || (gy >= 0 && gy < a.H);
                gy = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy);
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    int gx = w + kx * a.dil - a.pad;
                    const bool ok = rok && (rep || (gx >= 0 && gx < a.W));
                    gx = gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx);
                    const float xv = ok ? xc[(long long)gy * a.W + gx] : 0.f;
#pragma unroll
                    for (int o = 0; o < WS_MAXCO; ++o) acc[o][ky * K + kx] += d[o] * xv;
                }
            }
        }
    }
    __shared__ float sh[WS_NT];
    for (int o = 0; o < a.Cout; ++o)
        for (int t = 0; t < TAPS; ++t) {
            float v = 0.f;
#pragma unroll
            for (int oo = 0; oo < WS_MAXCO; ++oo)
#pragma unroll
                for (int tt = 0; tt < TAPS; ++tt)
                    if (oo == o && tt == t) v = acc[oo][tt];  // compile-time indexed registers
            sh[threadIdx.x] = v;
            __syncthreads();
            for (int st = WS_NT / 2; st > 0; st >>= 1) {
                if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
                __syncthreads();
            }
            if (threadIdx.x == 0) a.part[((long long)slab * a.Cout + o) * (a.Cin * TAPS) + ci * TAPS + t] = sh[0];
            __syncthreads();
        }
}

// ---- weight gradient, any channel counts (NormUnet 14 / 28 / 56 ..., gate convolutions into 3 F channels): k = 1, 3 or 5, any dilation ------
// dW[co][ci, tap] = sum over pixels of dy[co][p] * pad(x)[ci][p + tap] as a GEMM over pixels on v_mfma_f32_32x32x2_f32: A = dy (32 couts x
// 2 pixels), B = the shifted x (2 pixels x 32 columns; a column block is floor(32 / taps) input channels x taps: 3 x 9 for 3x3 -- 27 of 32
// columns used --, 32 channels for 1x1, one channel for 5x5).  A workgroup owns a group of (cout block, column block) pairs (<= 64: 8 waves x 8 accumulators) and walks the 8 x 32
// pixel tiles of its slab with the x tile of the group's channels and the dy tile of the group's couts in LDS; partial sums per slab are
// summed in fixed order by k_wgrad_reduce (bit-reproducible).
#define WGN_NT 512
#define WGN_TH 8
#define WGN_TW 32
#define WGN_DS 257
typedef float wgn_f32x16 __attribute__((ext_vector_type(16)));
struct WgenArgs {
    const float* x;
    const float* dy;
    float* part;            // [slab][Cout][Cin * taps]
    int B, Cin, Cout, H, W, k, dil, pad, pad_mode, tiles_x, ntiles;
    int cpb;                // input channels per column block
    int co_grp, nb_grp;     // couts / column blocks per workgroup group
    int n_nbgrp, nb_total;
};
template <int SLOTS>
__global__ __launch_bounds__(WGN_NT, 1) void k_conv_wgrad_gen(WgenArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem_wgn[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int cg = blockIdx.y / a.n_nbgrp, ng = blockIdx.y - cg * a.n_nbgrp;
    const int co0 = cg * a.co_grp, co_cnt = a.Cout - co0 < a.co_grp ? a.Cout - co0 : a.co_grp, CB = (co_cnt + 31) / 32;
    const int nb0 = ng * a.nb_grp, nb_cnt = a.nb_total - nb0 < a.nb_grp ? a.nb_total - nb0 : a.nb_grp;
    const int ci0 = nb0 * a.cpb, ci_cnt = a.Cin - ci0 < nb_cnt * a.cpb ? a.Cin - ci0 : nb_cnt * a.cpb;
    const int P = CB * nb_cnt, taps = a.k * a.k;
    const int PH = WGN_TH + 2 * a.pad, PS = (WGN_TW + 2 * a.pad) | 1, PW = WGN_TW + 2 * a.pad;
    float* xs = smem_wgn;                                   // [ci_cnt][PH][PS]
    float* dys = smem_wgn + ((ci_cnt * PH * PS + 3) & ~3);  // [CB * 32][WGN_DS]
    const long long plane = (long long)a.H * a.W;
    const bool rep = a.pad_mode == MRX_PAD_REPLICATE;

    // per slot: the pair's A / B lane offsets and whether this lane's column exists
    int aoff[SLOTS], boff[SLOTS], gcol[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int p = wave + 8 * s;
        const int cb = p % CB, nb = p / CB;
        aoff[s] = (cb * 32 + l31) * WGN_DS;
        const int cil = l31 / taps, tap = l31 - cil * taps;
        const int cl = nb * a.cpb + cil;                    // channel within the group
        const bool ok = p < P && cl < ci_cnt && cil < a.cpb;
        boff[s] = ok ? (cl * PH + (tap / a.k) * a.dil) * PS + (tap % a.k) * a.dil : 0;
        gcol[s] = ok ? (ci0 + cl) * taps + tap : -1;
    }
    wgn_f32x16 acc[SLOTS];
#pragma unroll
    for (int s = 0; s < SLOTS; ++s)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;

    const int total_tiles = a.ntiles * a.B;
    for (int t = blockIdx.x; t < total_tiles; t += gridDim.x) {
        const int b = t / a.ntiles, tile = t - b * a.ntiles, ty0 = tile / a.tiles_x;
        const int h0 = ty0 * WGN_TH, w0 = (tile - ty0 * a.tiles_x) * WGN_TW;
        __syncthreads();                                    // the previous tile is consumed
        for (int i = tid; i < ci_cnt * PH * PW; i += WGN_NT) {
            const int c = i / (PH * PW), e = i - c * (PH * PW), ty = e / PW, tx = e - ty * PW;
            int gy = h0 + ty - a.pad, gx = w0 + tx - a.pad;
            const bool ok = rep || (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W);
            gy = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy);
            gx = gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx);
            xs[(c * PH + ty) * PS + tx] = ok ? a.x[((long long)b * a.Cin + ci0 + c) * plane + (long long)gy * a.W + gx] : 0.f;
        }
        for (int i = tid; i < CB * 32 * WGN_TH * WGN_TW; i += WGN_NT) {
            const int co = i >> 8, e = i & 255, gy = h0 + (e >> 5), gx = w0 + (e & 31);
            const bool ok = co < co_cnt && gy < a.H && gx < a.W;
            dys[co * WGN_DS + e] = ok ? a.dy[((long long)b * a.Cout + co0 + co) * plane + (long long)gy * a.W + gx] : 0.f;
        }
        __syncthreads();
        for (int row = 0; row < WGN_TH; ++row)
#pragma unroll 4
            for (int st = 0; st < WGN_TW / 2; ++st) {
                const int px = 2 * st + lhi;
#pragma unroll
                for (int s = 0; s < SLOTS; ++s) {
                    if (wave + 8 * s < P) {                 // wave-uniform
                        const float av = dys[aoff[s] + row * WGN_TW + px];
                        const float bv = xs[boff[s] + row * PS + px];
                        acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[s], 0, 0, 0);
                    }
                }
            }
    }
    const long long N = (long long)a.Cin * taps;
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
        const int p = wave + 8 * s;
        if (p < P && gcol[s] >= 0) {
            const int cb = p % CB;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = cb * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (co < co_cnt) a.part[((long long)blockIdx.x * a.Cout + co0 + co) * N + gcol[s]] = acc[s][r];
            }
        }
    }
}

struct WgenPlan {
    int cpb, co_grp, nb_grp, n_cogrp, n_nbgrp, nb_total, slots, slabs;
    size_t lds;
};
static int query_cus();
static WgenPlan wgen_plan(int B, int Cin, int Cout, int H, int W, int k, int dil) {
    WgenPlan p;
    const int pad = dil * (k - 1) / 2;
    p.cpb = 32 / (k * k);
    p.nb_total = (Cin + p.cpb - 1) / p.cpb;
    p.co_grp = Cout < 64 ? Cout : 64;
    p.n_cogrp = (Cout + p.co_grp - 1) / p.co_grp;
    const int CBg = (p.co_grp + 31) / 32;
    const size_t dy_bytes = (size_t)CBg * 32 * WGN_DS * 4, xch = (size_t)(WGN_TH + 2 * pad) * ((WGN_TW + 2 * pad) | 1) * 4;
    int nb = (int)((150 * 1024 - dy_bytes) / (xch * p.cpb));
    if (nb > 64 / CBg) nb = 64 / CBg;
    if (nb > p.nb_total) nb = p.nb_total;
    if (nb < 1) nb = 1;
    p.nb_grp = nb;
    p.n_nbgrp = (p.nb_total + nb - 1) / nb;
    const int pairs = CBg * nb;
    p.slots = pairs <= 8 ? 1 : (pairs <= 16 ? 2 : (pairs <= 32 ? 4 : 8));
    int chans = nb * p.cpb < Cin ? nb * p.cpb : Cin;
    p.lds = (((size_t)chans * xch / 4 + 3) & ~(size_t)3) * 4 + dy_bytes;
    const long long tiles = (long long)mrx_cdiv(W, WGN_TW) * mrx_cdiv(H, WGN_TH) * B;
    const int cus = query_cus();
    p.slabs = (int)(tiles < cus ? tiles : cus);
    return p;
}

static int wgrad_nparts64(int n_cu, long long nt_total) { return (int)(nt_total < n_cu ? nt_total : n_cu); }
static int g_n_cu = 0;
static int query_cus() {
    if (!g_n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        g_n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return g_n_cu;
}
#define WS_SLABS 64

static int wgrad_nparts64(int n_cu, long long nt_total);
static int query_cus();
// partial planes the matrix-core kernel writes: persistent workgroups (min(CUs, tiles)) x pixel groups (few column blocks) -- the same
// formula the launcher below uses, so the caller allocates what is written (38 MB for the 64 -> 64 3x3 layer, not a 1.2 GB bound)
static int wgrad_parts64(int B, int H, int W, long long N) {
    const long long ntiles = (long long)mrx_cdiv(W, WG_TW) * mrx_cdiv(H, WG_TH) * B;
    const int cus = query_cus();
    int nparts = wgrad_nparts64(cus < 1024 ? cus : 1024, ntiles);
    const int nbl = (int)((N + 31) / 32);
    return nparts * (nbl >= 5 ? 1 : (nbl > 2 ? 2 : (nbl > 1 ? 4 : 8)));
}
extern "C" int64_t mrx_conv_wgrad_work_floats(int B, int Cin, int Cout, int H, int W, int k) {
    const long long N = (long long)Cin * k * k, n = (long long)Cout * N;
    if (B < 1 || H < 1 || W < 1) return -1;
    if (Cout == 64 && N <= 8 * WG_MAXNB * 32) return (int64_t)wgrad_parts64(B, H, W, N) * n;
    if (Cout > WS_MAXCO && (k == 1 || k == 3 || k == 5)) return (int64_t)wgen_plan(B, Cin, Cout, H, W, k, 1).slabs * n;   // slabs do not depend on the dilation
    return (int64_t)WS_SLABS * n;
}

extern "C" int mrx_conv_wgrad(const float* x, const float* dy, float* dw, float* work, int B, int Cin, int Cout, int H, int W, int k,
                              int dil, int pad_mode, int accumulate, void* stream) {
    MRX_REQUIRE(x && dy && dw && work, MRX_EINVAL, "mrx_conv_wgrad: null pointer");
    MRX_REQUIRE(B >= 1 && Cin >= 1 && Cout >= 1 && H >= 1 && W >= 1 && k >= 1 && (k & 1) && dil >= 1, MRX_EINVAL, "mrx_conv_wgrad: bad dims");
    MRX_REQUIRE(pad_mode == MRX_PAD_ZERO || pad_mode == MRX_PAD_REPLICATE, MRX_EINVAL, "mrx_conv_wgrad: pad mode %d", pad_mode);
    hipStream_t st = (hipStream_t)stream;
    const int pad = dil * (k - 1) / 2;
    const long long N = (long long)Cin * k * k, total = (long long)Cout * N;
    int nparts;
    if (Cout == 64 && N <= 8 * WG_MAXNB * 32) {
        WgradArgs a;
        a.x = x;
        a.dy = dy;
        a.part = work;
        a.B = B;
        a.Cin = Cin;
        a.H = H;
        a.W = W;
        a.k = k;
        a.dil = dil;
        a.pad = pad;
        a.pad_mode = pad_mode;
        a.N = (int)N;
        a.tiles_x = mrx_cdiv(W, WG_TW);
        a.ntiles = a.tiles_x * mrx_cdiv(H, WG_TH);
        a.vec = (W & 3) == 0 && (((uintptr_t)x | (uintptr_t)dy) & 15) == 0;
        MRX_REQUIRE(pad <= 2, MRX_EUNSUP, "mrx_conv_wgrad: padding %d (k=%d dil=%d; up to 2)", pad, k, dil);
        a.PH = WG_TH + 2 * pad;
        a.XS = (WG_TW + 2 * pad) | 1;  // odd row stride
        const size_t lds = sizeof(float) * ((size_t)64 * WG_DYS + (size_t)Cin * a.PH * a.XS);
        MRX_REQUIRE(lds <= 160 * 1024, MRX_EUNSUP, "mrx_conv_wgrad: %zu bytes of LDS (Cin=%d k=%d dil=%d)", lds, Cin, k, dil);
        static size_t attr_bytes[3] = {0, 0, 0};
        const void* kern = pad == 0 ? (const void*)k_conv_wgrad64<0> : pad == 1 ? (const void*)k_conv_wgrad64<1> : (const void*)k_conv_wgrad64<2>;
        if (lds > 48 * 1024 && attr_bytes[pad] < lds) {
            MRX_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_bytes[pad] = lds;
        }
        nparts = wgrad_nparts64(query_cus() < 1024 ? query_cus() : 1024, (long long)a.ntiles * B);
        const int nblk64 = nparts, nbl = (int)((N + 31) / 32);
        nparts *= nbl >= 5 ? 1 : (nbl > 2 ? 2 : (nbl > 1 ? 4 : 8));   // pixel groups of the kernel (few column blocks)
        if (pad == 0) hipLaunchKernelGGL(k_conv_wgrad64<0>, dim3(nblk64), dim3(WG_NT), lds, st, a);
        else if (pad == 1) hipLaunchKernelGGL(k_conv_wgrad64<1>, dim3(nblk64), dim3(WG_NT), lds, st, a);
        else hipLaunchKernelGGL(k_conv_wgrad64<2>, dim3(nblk64), dim3(WG_NT), lds, st, a);
    } else if (Cout > WS_MAXCO) {
        MRX_REQUIRE(k == 1 || k == 3 || k == 5, MRX_EUNSUP, "mrx_conv_wgrad: Cout=%d with k=%d (generic kernel: k 1, 3 or 5)", Cout, k);
        const WgenPlan pl = wgen_plan(B, Cin, Cout, H, W, k, dil);
        MRX_REQUIRE(pl.lds <= 160 * 1024, MRX_EUNSUP, "mrx_conv_wgrad: %zu bytes of LDS (k=%d dil=%d)", pl.lds, k, dil);
        WgenArgs a;
        a.x = x, a.dy = dy, a.part = work;
        a.B = B, a.Cin = Cin, a.Cout = Cout, a.H = H, a.W = W, a.k = k, a.dil = dil, a.pad = pad, a.pad_mode = pad_mode;
        a.tiles_x = mrx_cdiv(W, WGN_TW), a.ntiles = a.tiles_x * mrx_cdiv(H, WGN_TH);
        a.cpb = pl.cpb, a.co_grp = pl.co_grp, a.nb_grp = pl.nb_grp, a.n_nbgrp = pl.n_nbgrp, a.nb_total = pl.nb_total;
        const void* kern = pl.slots == 1 ? (const void*)k_conv_wgrad_gen<1> : pl.slots == 2 ? (const void*)k_conv_wgrad_gen<2>
                         : pl.slots == 4 ? (const void*)k_conv_wgrad_gen<4> : (const void*)k_conv_wgrad_gen<8>;
        static size_t attr_bytes[9] = {0};
        if (attr_bytes[pl.slots] < pl.lds) {
            MRX_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds));
            attr_bytes[pl.slots] = pl.lds;
        }
        nparts = pl.slabs;
        const dim3 grid(pl.slabs, pl.n_cogrp * pl.n_nbgrp);
        if (pl.slots == 1) hipLaunchKernelGGL(k_conv_wgrad_gen<1>, grid, dim3(WGN_NT), pl.lds, st, a);
        else if (pl.slots == 2) hipLaunchKernelGGL(k_conv_wgrad_gen<2>, grid, dim3(WGN_NT), pl.lds, st, a);
        else if (pl.slots == 4) hipLaunchKernelGGL(k_conv_wgrad_gen<4>, grid, dim3(WGN_NT), pl.lds, st, a);
        else hipLaunchKernelGGL(k_conv_wgrad_gen<8>, grid, dim3(WGN_NT), pl.lds, st, a);
    } else {
        MRX_REQUIRE(Cout <= WS_MAXCO, MRX_EUNSUP, "mrx_conv_wgrad: Cout=%d (64 or <= %d)", Cout, WS_MAXCO);
        WsmallArgs a;
        a.x = x;
        a.dy = dy;
        a.part = work;
        a.B = B;
        a.Cin = Cin;
        a.Cout = Cout;
        a.H = H;
        a.W = W;
        a.k = k;
        a.dil = dil;
        a.pad = pad;
        a.pad_mode = pad_mode;
        const int rows = B * H;
        a.rows_per_slab = (rows + WS_SLABS - 1) / WS_SLABS;
        nparts = (rows + a.rows_per_slab - 1) / a.rows_per_slab;
        MRX_REQUIRE(k == 1 || k == 3 || k == 5, MRX_EUNSUP, "mrx_conv_wgrad: kernel size %d with Cout=%d (1, 3, 5)", k, Cout);
        if (k == 1) hipLaunchKernelGGL(k_conv_wgrad_small<1>, dim3(Cin, nparts), dim3(WS_NT), 0, st, a);
        else if (k == 3) hipLaunchKernelGGL(k_conv_wgrad_small<3>, dim3(Cin, nparts), dim3(WS_NT), 0, st, a);
        else hipLaunchKernelGGL(k_conv_wgrad_small<5>, dim3(Cin, nparts), dim3(WS_NT), 0, st, a);
    }
    MRX_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_wgrad_reduce, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, st, (const float*)work, nparts, total, dw,
                       accumulate);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- adjoint of replicate padding ---------------------------------------------------------------------------------------------
// out[p][h][w] = sum of g[p][i][j] over the padded positions (i, j) that clamp to (h, w); planes = B*C.
__global__ void k_reppad_fold(const float* __restrict__ g, float* __restrict__ out, long long planes, int H, int W, int pad) {
    const long long total = planes * H * W;
    const int PW = W + 2 * pad, PH = H + 2 * pad;
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long long)gridDim.x * blockDim.x) {
        const long long p = o / ((long long)H * W);
        const int rem = (int)(o - p * (long long)H * W), h = rem / W, w = rem - h * W;
        const int i0 = h == 0 ? 0 : h + pad, i1 = h == H - 1 ? PH - 1 : h + pad;
        const int j0 = w == 0 ? 0 : w + pad, j1 = w == W - 1 ? PW - 1 : w + pad;
        const float* gp = g + p * (long long)PH * PW;
        float s = 0.f;
        for (int i = i0; i <= i1; ++i)
            for (int j = j0; j <= j1; ++j) s += gp[(long long)i * PW + j];
        out[o] = s;
    }
}
// The same fold when the interior of the gradient is in `out` already (mrx_conv2d_bf16_dgrad_rep): only the edge pixels change, each
// receives the frame positions of g that clamp to it (the interior position itself excluded).  2 (H + W) - 4 pixels per plane.
__global__ void k_reppad_fold_edges(const float* __restrict__ g, float* __restrict__ out, long long planes, int H, int W, int pad) {
    const int per = 2 * W + 2 * (H - 2 > 0 ? H - 2 : 0);
    const long long total = planes * per;
    const int PW = W + 2 * pad, PH = H + 2 * pad;
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long long)gridDim.x * blockDim.x) {
        const long long p = o / per;
        int e = (int)(o - p * per), h, w;
        if (e < W) {
            h = 0, w = e;
        } else if (e < 2 * W) {
            h = H - 1, w = e - W;
        } else {
            e -= 2 * W;
            h = 1 + (e >> 1), w = (e & 1) ? W - 1 : 0;
        }
        if (H == 1 && e >= W && e < 2 * W) continue;          // a single row is listed once
        if (W == 1 && e >= 2 * W && (e & 1)) continue;        // a single column is listed once
        const int i0 = h == 0 ? 0 : h + pad, i1 = h == H - 1 ? PH - 1 : h + pad;
        const int j0 = w == 0 ? 0 : w + pad, j1 = w == W - 1 ? PW - 1 : w + pad;
        const float* gp = g + p * (long long)PH * PW;
        float s = 0.f;
        for (int i = i0; i <= i1; ++i)
            for (int j = j0; j <= j1; ++j)
                if (i != h + pad || j != w + pad) s += gp[(long long)i * PW + j];
        out[p * (long long)H * W + (long long)h * W + w] += s;
    }
}
extern "C" int mrx_reppad_fold_edges(const float* g, float* out, int64_t planes, int H, int W, int pad, void* stream) {
    MRX_REQUIRE(g && out && planes >= 0 && H >= 1 && W >= 1 && pad >= 0, MRX_EINVAL, "mrx_reppad_fold_edges: bad argument");
    const long long total = planes * (2ll * W + 2ll * (H - 2 > 0 ? H - 2 : 0));
    if (total == 0 || pad == 0) return MRX_OK;
    const long long nb = (total + 255) / 256;
    hipLaunchKernelGGL(k_reppad_fold_edges, dim3((unsigned)(nb < 65535 ? nb : 65535)), dim3(256), 0, (hipStream_t)stream, g, out,
                       (long long)planes, H, W, pad);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
extern "C" int mrx_reppad_fold(const float* g, float* out, int64_t planes, int H, int W, int pad, void* stream) {
    MRX_REQUIRE(g && out && planes >= 0 && H >= 1 && W >= 1 && pad >= 0, MRX_EINVAL, "mrx_reppad_fold: bad argument");
    const long long total = planes * H * W;
    if (total == 0) return MRX_OK;
    const long long nb = (total + 255) / 256;
    hipLaunchKernelGGL(k_reppad_fold, dim3((unsigned)(nb < 65535 ? nb : 65535)), dim3(256), 0, (hipStream_t)stream, g, out, (long long)planes,
                       H, W, pad);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- ReLU backward (+ IndRNN cell pieces) with per-channel sums ------------------------------------------------------------------
// dpre = dy * (y > 0).  sums[c][0] = sum dpre; with h_prev: dh_prev = dpre * hh[c] and sums[c][1] = sum dpre * h_prev.
// One workgroup per (channel, slab); partials work[(c*RB_SLABS + slab)*2 + {0,1}], combined in order in double.
#define RB_NT 256
#define RB_SLABS 32
__global__ __launch_bounds__(RB_NT) void k_relu_bwd(const float* __restrict__ dy, const float* __restrict__ dy2, const float* __restrict__ y,
                                                    const float* __restrict__ hprev, const float* __restrict__ hh, float* __restrict__ dpre,
                                                    float* __restrict__ dhprev, float* __restrict__ work, int B, int C, long long HW) {
    const int c = blockIdx.x, slab = blockIdx.y;
    const long long per = (HW + RB_SLABS - 1) / RB_SLABS, p0 = slab * per, p1 = p0 + per < HW ? p0 + per : HW;
    const float hw = hh ? hh[c] : 0.f;
    float s0 = 0.f, s1 = 0.f;
    for (int b = 0; b < B; ++b) {
        const long long base = ((long long)b * C + c) * HW;
        for (long long p = p0 + threadIdx.x; p < p1; p += RB_NT) {
            const float up = dy2 ? dy[base + p] + dy2[base + p] : dy[base + p];   // two gradient paths into y (next layer, next time-step)
            const float d = y[base + p] > 0.f ? up : 0.f;
            dpre[base + p] = d;
            s0 += d;
            if (hprev) {
                s1 += d * hprev[base + p];
                dhprev[base + p] = d * hw;
            }
        }
    }
    __shared__ float sh0[RB_NT], sh1[RB_NT];
    sh0[threadIdx.x] = s0;
    sh1[threadIdx.x] = s1;
    __syncthreads();
    for (int st = RB_NT / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            sh0[threadIdx.x] += sh0[threadIdx.x + st];
            sh1[threadIdx.x] += sh1[threadIdx.x + st];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        work[((long long)c * RB_SLABS + slab) * 2] = sh0[0];
        work[((long long)c * RB_SLABS + slab) * 2 + 1] = sh1[0];
    }
}
__global__ void k_relu_bwd_final(const float* __restrict__ work, float* __restrict__ sums, float* __restrict__ acc0, float* __restrict__ acc1,
                                 int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double a = 0.0, b = 0.0;
    for (int s = 0; s < RB_SLABS; ++s) {
        a += (double)work[((long long)c * RB_SLABS + s) * 2];
        b += (double)work[((long long)c * RB_SLABS + s) * 2 + 1];
    }
    if (sums) {
        sums[2 * c] = (float)a;
        sums[2 * c + 1] = (float)b;
    }
    if (acc0) acc0[c] += (float)a;     // bias gradient accumulated in place (one writer per element: deterministic)
    if (acc1) acc1[c] += (float)b;     // hh gradient
}
extern "C" int64_t mrx_relu_bwd_work_floats(int C) { return (int64_t)2 * RB_SLABS * (C > 0 ? C : 1); }
extern "C" int mrx_relu_bwd(const float* dy, const float* y, const float* h_prev, const float* hh, float* dpre, float* dh_prev,
                            float* sums, float* work, int B, int C, int64_t HW, void* stream) {
    MRX_REQUIRE(dy && y && dpre && sums && work && B >= 1 && C >= 1 && C <= 65535 && HW >= 1, MRX_EINVAL, "mrx_relu_bwd: bad argument");
    MRX_REQUIRE(!h_prev || (hh && dh_prev), MRX_EINVAL, "mrx_relu_bwd: h_prev needs hh and dh_prev");
    hipLaunchKernelGGL(k_relu_bwd, dim3(C, RB_SLABS), dim3(RB_NT), 0, (hipStream_t)stream, dy, (const float*)nullptr, y, h_prev, hh, dpre,
                       dh_prev, work, B, C, (long long)HW);
    hipLaunchKernelGGL(k_relu_bwd_final, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, (const float*)work, sums, (float*)nullptr,
                       (float*)nullptr, C);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
// The same step for the explicit training tape (mridc_amd/training.py): the upstream gradient may arrive as two addends (dy + dy2: the
// paths through the next layer and through the next time-step), and the per-channel sums are ADDED into the bias / hh gradient buffers
// (acc_bias[c] += sum dpre, acc_hh[c] += sum dpre * h_prev; either may be null) -- no separate accumulation launches.
extern "C" int mrx_relu_bwd_acc(const float* dy, const float* dy2, const float* y, const float* h_prev, const float* hh, float* dpre,
                                float* dh_prev, float* acc_bias, float* acc_hh, float* work, int B, int C, int64_t HW, void* stream) {
    MRX_REQUIRE(dy && y && dpre && work && B >= 1 && C >= 1 && C <= 65535 && HW >= 1, MRX_EINVAL, "mrx_relu_bwd_acc: bad argument");
    MRX_REQUIRE(!h_prev || (hh && dh_prev), MRX_EINVAL, "mrx_relu_bwd_acc: h_prev needs hh and dh_prev");
    MRX_REQUIRE(!acc_hh || h_prev, MRX_EINVAL, "mrx_relu_bwd_acc: acc_hh without h_prev");
    hipLaunchKernelGGL(k_relu_bwd, dim3(C, RB_SLABS), dim3(RB_NT), 0, (hipStream_t)stream, dy, dy2, y, h_prev, hh, dpre, dh_prev, work, B, C,
                       (long long)HW);
    hipLaunchKernelGGL(k_relu_bwd_final, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, (const float*)work, (float*)nullptr, acc_bias,
                       acc_hh, C);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- eta-side glue of the explicit tape (rim_block.py:239-248 and rim_utils.py:67 backwards) --------------------------------------------
// mrx_eta_grad_in:   tot = carry (or 0) + gl;  d2[b, c, h, w] = tot[b, h, w, c]   (gradient entering the final conv, NCHW with 2 channels)
// mrx_g4_to_complex: dz[b, h, w, :] = (g4[b, 2], g4[b, 3])                        (the gradient w.r.t. the log-likelihood-gradient channels)
// mrx_eta_grad_out:  out[b, h, w, :] = tot + (g4[b, 0], g4[b, 1]) + (t4[b, 2], t4[b, 3])   (identity path + eta channels + adjoint gradient)
__global__ void k_eta_grad_in(const float2* __restrict__ carry, const float2* __restrict__ gl, float2* __restrict__ tot, float* __restrict__ d2,
                              long long B, long long plane) {
    const long long total = B * plane;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        float2 v = gl[i];
        if (carry) {
            const float2 c = carry[i];
            v.x += c.x, v.y += c.y;
        }
        tot[i] = v;
        const long long b = i / plane, p = i - b * plane;
        d2[(b * 2) * plane + p] = v.x;
        d2[(b * 2 + 1) * plane + p] = v.y;
    }
}
__global__ void k_g4_to_complex(const float* __restrict__ g4, float2* __restrict__ dz, long long B, long long plane) {
    const long long total = B * plane;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / plane, p = i - b * plane;
        dz[i] = make_float2(g4[(b * 4 + 2) * plane + p], g4[(b * 4 + 3) * plane + p]);
    }
}
__global__ void k_eta_grad_out(const float2* __restrict__ tot, const float* __restrict__ g4, const float* __restrict__ t4, float2* __restrict__ out,
                               long long B, long long plane) {
    const long long total = B * plane;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / plane, p = i - b * plane;
        const float2 v = tot[i];
        out[i] = make_float2(v.x + g4[(b * 4) * plane + p] + t4[(b * 4 + 2) * plane + p],
                             v.y + g4[(b * 4 + 1) * plane + p] + t4[(b * 4 + 3) * plane + p]);
    }
}
// the same with the adjoint gradient still in its coil-group partial planes (mrx_llg372 with nparts): t4's channels 2, 3 = post * sum_k part_k, formed
// here in mrx_llg372's own order -- its combine launch and the [B,4,H,W] tensor saved
__global__ void k_eta_grad_out_parts(const float2* __restrict__ tot, const float* __restrict__ g4, const float2* __restrict__ part, int nparts, float post,
                                     float2* __restrict__ out, long long B, long long plane) {
    const long long total = B * plane;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / plane, p = i - b * plane;
        float2 s = part[i];
        for (int k = 1; k < nparts; ++k) {
            const float2 v = part[(long long)k * total + i];
            s.x += v.x;
            s.y += v.y;
        }
        const float2 v = tot[i];
        out[i] = make_float2(v.x + g4[(b * 4) * plane + p] + s.x * post, v.y + g4[(b * 4 + 1) * plane + p] + s.y * post);
    }
}
static inline unsigned tape_grid(long long n) {
    long long g = (n + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 4096 ? 4096 : g));
}
extern "C" int mrx_eta_grad_in(const float* carry, const float* gl, float* tot, float* d2, int B, int64_t plane, void* stream) {
    MRX_REQUIRE(gl && tot && d2 && B >= 1 && plane >= 1, MRX_EINVAL, "mrx_eta_grad_in: bad argument");
    hipLaunchKernelGGL(k_eta_grad_in, dim3(tape_grid((long long)B * plane)), dim3(256), 0, (hipStream_t)stream, (const float2*)carry,
                       (const float2*)gl, (float2*)tot, d2, (long long)B, (long long)plane);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
extern "C" int mrx_g4_to_complex(const float* g4, float* dz, int B, int64_t plane, void* stream) {
    MRX_REQUIRE(g4 && dz && B >= 1 && plane >= 1, MRX_EINVAL, "mrx_g4_to_complex: bad argument");
    hipLaunchKernelGGL(k_g4_to_complex, dim3(tape_grid((long long)B * plane)), dim3(256), 0, (hipStream_t)stream, g4, (float2*)dz, (long long)B,
                       (long long)plane);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
extern "C" int mrx_eta_grad_out(const float* tot, const float* g4, const float* t4, float* out, int B, int64_t plane, void* stream) {
    MRX_REQUIRE(tot && g4 && t4 && out && B >= 1 && plane >= 1, MRX_EINVAL, "mrx_eta_grad_out: bad argument");
    hipLaunchKernelGGL(k_eta_grad_out, dim3(tape_grid((long long)B * plane)), dim3(256), 0, (hipStream_t)stream, (const float2*)tot, g4, t4,
                       (float2*)out, (long long)B, (long long)plane);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

extern "C" int mrx_eta_grad_out_parts(const float* tot, const float* g4, const float* parts, int nparts, float post, float* out, int B, int64_t plane,
                                      void* stream) {
    MRX_REQUIRE(tot && g4 && parts && nparts >= 1 && out && B >= 1 && plane >= 1, MRX_EINVAL, "mrx_eta_grad_out_parts: bad argument");
    hipLaunchKernelGGL(k_eta_grad_out_parts, dim3(tape_grid((long long)B * plane)), dim3(256), 0, (hipStream_t)stream, (const float2*)tot, g4,
                       (const float2*)parts, nparts, post, (float2*)out, (long long)B, (long long)plane);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- training loss of the CIRIM (cirim.py:199-247, l1): L = mean | target - |p| / max|p| | over one complex image ---------------
// forward:  parts[0] = sum |target - |p|/M|, parts[1] = sum sign(|p|/M - target) * |p|   (M = max |p|, a device scalar from mrx_max_abs)
// backward: dp = gscale/n * sign * (1/M) * p/|p|, and the element(s) with |p| = M also receive -(gscale/n) * parts[1] / M^2 * p/|p|
//           (the gradient of the max), exactly what torch autograd produces for abs(t / max(abs(t))).
#define LS_NT 256
#define LS_BLOCKS 128
// |p| exactly as mrx_max_abs forms it (elementwise.hip: two rounded squares, one rounded add, the correctly rounded root), whatever the
// contraction mode of this file: the backward pass finds the arg-max pixel by `a == M`, so both moduli must agree bit for bit
__device__ __forceinline__ float absl1_modulus(float2 v) {
    return (float)sqrt((double)mrx_sumsq2(v.x, v.y));
}
// MP: M is not given but formed here from the per-workgroup maxima of the producer (mrx_tl_final_gather_max: `M` = those np partials), by every
// workgroup alike (NaN propagates as in mrx_max_abs); workgroup 0 leaves it in Mout for the backward -- the k_max_final launch of mrx_max_abs saved
template <bool MP>
__global__ __launch_bounds__(LS_NT) void k_absl1_partial(const float2* __restrict__ p, const float* __restrict__ target,
                                                         const float* __restrict__ M, int np, float* __restrict__ Mout, float* __restrict__ work,
                                                         long long n) {
    __shared__ float mred[LS_NT / 64];
    float mval;
    if (MP) {
        float m = 0.f;
        for (int i = threadIdx.x; i < np; i += LS_NT) {
            const float v = M[i];
            m = (v > m || v != v) ? v : m;
        }
        for (int off = 32; off > 0; off >>= 1) {
            const float o = __shfl_xor(m, off, 64);
            m = (o > m || o != o) ? o : m;
        }
        if ((threadIdx.x & 63) == 0) mred[threadIdx.x >> 6] = m;
        __syncthreads();
        m = mred[0];
        for (int w = 1; w < LS_NT / 64; ++w) m = (mred[w] > m || mred[w] != mred[w]) ? mred[w] : m;
        mval = m;
        if (blockIdx.x == 0 && threadIdx.x == 0) Mout[0] = m;
        __syncthreads();
    } else {
        mval = M[0];
    }
    const float inv = 1.0f / mval;
    float s0 = 0.f, s1 = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float2 v = p[i];
        const float a = absl1_modulus(v), d = mrx_mul_sub(a, inv, target[i]);
        s0 += fabsf(d);
        s1 += (d > 0.f ? a : (d < 0.f ? -a : 0.f));
    }
    __shared__ float sh0[LS_NT], sh1[LS_NT];
    sh0[threadIdx.x] = s0;
    sh1[threadIdx.x] = s1;
    __syncthreads();
    for (int st = LS_NT / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) {
            sh0[threadIdx.x] += sh0[threadIdx.x + st];
            sh1[threadIdx.x] += sh1[threadIdx.x + st];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        work[2 * blockIdx.x] = sh0[0];
        work[2 * blockIdx.x + 1] = sh1[0];
    }
}
// one wave: lane l sums blocks l, l + 64, ... in double, then a fixed butterfly (one thread walking 2 x 128 dependent loads took 9 us)
__global__ void k_absl1_final(const float* __restrict__ work, int nb, long long n, float* __restrict__ out) {
    double a = 0.0, b = 0.0;
    for (int k = threadIdx.x; k < nb; k += 64) {
        a += (double)work[2 * k];
        b += (double)work[2 * k + 1];
    }
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_xor(a, off, 64);
        b += __shfl_xor(b, off, 64);
    }
    if (threadIdx.x == 0) {
        out[0] = (float)(a / (double)n);  // the loss
        out[1] = (float)b;                // sum sign * |p|
    }
}
// ETA: the explicit tape's next glue step in the same pass (mrx_eta_grad_in): tot = carry (or 0) + dp, d2[b, c] = tot[..., c]; dp itself is not stored
template <bool ETA>
__global__ void k_absl1_bwd(const float2* __restrict__ p, const float* __restrict__ target, const float* __restrict__ M,
                            const float* __restrict__ fw, const float* __restrict__ gout, float gscale, float2* __restrict__ dp,
                            long long n, const float2* __restrict__ carry, float* __restrict__ d2, long long plane) {
    const float m = M[0], inv = 1.0f / m, g = (gout ? gout[0] : 1.0f) * gscale / (float)n;
    const float corr = g * fw[1] * inv * inv;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float2 v = p[i];
        const float a = absl1_modulus(v), d = mrx_mul_sub(a, inv, target[i]);
        const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
        float c = g * sg * inv;
        if (a == m) c -= corr;
        const float ia = a > 0.f ? 1.0f / a : 0.f;
        float2 r = make_float2(c * v.x * ia, c * v.y * ia);
        if (ETA) {
            if (carry) {
                const float2 cv = carry[i];
                r = make_float2(r.x + cv.x, r.y + cv.y);       // (= gl + carry: the same two addends as k_eta_grad_in)
            }
            const long long b = i / plane, px = i - b * plane;
            d2[(b * 2) * plane + px] = r.x;
            d2[(b * 2 + 1) * plane + px] = r.y;
        }
        dp[i] = r;
    }
}
extern "C" int64_t mrx_absl1_work_floats(void) { return 2 * LS_BLOCKS; }
extern "C" int mrx_absl1_loss(const float* p, const float* target, const float* maxabs, float* out2, float* work, int64_t n,
                              void* stream) {
    MRX_REQUIRE(p && target && maxabs && out2 && work && n >= 1, MRX_EINVAL, "mrx_absl1_loss: bad argument");
    const long long nbl = (n + LS_NT - 1) / LS_NT;
    const int nb = (int)(nbl < LS_BLOCKS ? nbl : LS_BLOCKS);
    hipLaunchKernelGGL(k_absl1_partial<false>, dim3(nb), dim3(LS_NT), 0, (hipStream_t)stream, (const float2*)p, target, maxabs, 0, (float*)nullptr, work,
                       (long long)n);
    hipLaunchKernelGGL(k_absl1_final, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float*)work, nb, (long long)n, out2);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
// mrx_absl1_loss with the maximum given as `np` per-workgroup partial maxima of |p| (mrx_tl_final_gather_max); the maximum itself is left in maxabs_out
extern "C" int mrx_absl1_loss_mp(const float* p, const float* target, const float* max_partials, int np, float* maxabs_out, float* out2, float* work,
                                 int64_t n, void* stream) {
    MRX_REQUIRE(p && target && max_partials && np >= 1 && maxabs_out && out2 && work && n >= 1, MRX_EINVAL, "mrx_absl1_loss_mp: bad argument");
    const long long nbl = (n + LS_NT - 1) / LS_NT;
    const int nb = (int)(nbl < LS_BLOCKS ? nbl : LS_BLOCKS);
    hipLaunchKernelGGL(k_absl1_partial<true>, dim3(nb), dim3(LS_NT), 0, (hipStream_t)stream, (const float2*)p, target, max_partials, np, maxabs_out, work,
                       (long long)n);
    hipLaunchKernelGGL(k_absl1_final, dim3(1), dim3(64), 0, (hipStream_t)stream, (const float*)work, nb, (long long)n, out2);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
extern "C" int mrx_absl1_loss_bwd(const float* p, const float* target, const float* maxabs, const float* fwd_out2, const float* gout,
                                  float gscale, float* dp, int64_t n, void* stream) {
    MRX_REQUIRE(p && target && maxabs && fwd_out2 && dp && n >= 1, MRX_EINVAL, "mrx_absl1_loss_bwd: bad argument");
    const long long nbl = (n + 255) / 256;
    hipLaunchKernelGGL(k_absl1_bwd<false>, dim3((unsigned)(nbl < 4096 ? nbl : 4096)), dim3(256), 0, (hipStream_t)stream, (const float2*)p, target,
                       maxabs, fwd_out2, gout, gscale, (float2*)dp, (long long)n, (const float2*)nullptr, (float*)nullptr, 1ll);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
// mrx_absl1_loss_bwd + mrx_eta_grad_in in one pass: tot [B,plane,2] = carry (may be NULL) + d(loss)/dp, d2 [B,2,plane] = its two channels as planes
extern "C" int mrx_absl1_loss_bwd_eta(const float* p, const float* target, const float* maxabs, const float* fwd_out2, const float* gout, float gscale,
                                      const float* carry, float* tot, float* d2, int B, int64_t plane, void* stream) {
    MRX_REQUIRE(p && target && maxabs && fwd_out2 && tot && d2 && B >= 1 && plane >= 1, MRX_EINVAL, "mrx_absl1_loss_bwd_eta: bad argument");
    const long long n = (long long)B * plane, nbl = (n + 255) / 256;
    hipLaunchKernelGGL(k_absl1_bwd<true>, dim3((unsigned)(nbl < 4096 ? nbl : 4096)), dim3(256), 0, (hipStream_t)stream, (const float2*)p, target,
                       maxabs, fwd_out2, gout, gscale, (float2*)tot, n, (const float2*)carry, d2, (long long)plane);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- Adam on a flat parameter buffer (torch.optim.Adam semantics, weight_decay = 0, amsgrad = False; base_cirim_train.yaml) -----
__global__ void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long long n,
                       float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt, float gscale) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float gi = g[i] * gscale;
        const float mi = b1 * m[i] + (1.0f - b1) * gi;
        const float vi = b2 * v[i] + (1.0f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = p[i] - (lr / bc1) * mi / (sqrtf(vi) / bc2_sqrt + eps);
    }
}
extern "C" int mrx_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                             float beta2, float eps, int step, float grad_scale, void* stream) {
    MRX_REQUIRE(param && grad && exp_avg && exp_avg_sq && n >= 0 && step >= 1, MRX_EINVAL, "mrx_adam_step: bad argument");
    if (n == 0) return MRX_OK;
    const float bc1 = 1.0f - powf(beta1, (float)step), bc2 = 1.0f - powf(beta2, (float)step);
    const long long nbl = (n + 255) / 256;
    hipLaunchKernelGGL(k_adam, dim3((unsigned)(nbl < 4096 ? nbl : 4096)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq,
                       (long long)n, lr, beta1, beta2, eps, bc1, sqrtf(bc2), grad_scale);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
