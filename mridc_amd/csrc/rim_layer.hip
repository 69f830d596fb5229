// rim_layer.hip -- tuned fused RIM layer for gfx950: ConvNonlinear(ReLU, replicate pad) + IndRNNCell(1x1)
// (reference models/rim/conv_layers.py:121-123 + rnn_cells.py:384-391), fp32 in / fp32 out on v_mfma_f32_32x32x2_f32.
//
// Differences from the generic kernel in conv.hip (kept as the fallback and as a cross-check):
//   * kernel size / dilation / channel chunk are template parameters: the (tap, channel-pair) loop is fully unrolled
//     with immediate LDS offsets, no integer division anywhere in the hot loop or in the staging code;
//   * 8 waves per workgroup, one image row x 64 couts (2 accumulators) each, on an 8x32 tile: the weight chunk in LDS
//     is shared by 8 waves, 2 workgroups per CU give 4 waves per SIMD to cover LDS/global latency;
//   * weights are pre-packed once (mrx_rim_layer_pack) into the exact order the MFMA A-operand is read:
//     [chunk][tap][channel pair][lane half][cout] -- staging is a linear float4 copy, reads are conflict-free;
//   * the next channel chunk (input tile and weights) is prefetched into registers while the current one feeds the
//     matrix cores (issue-early / write-late staging);
//   * the 1x1 `ih` GEMM consumes the conv accumulators in registers (contraction index enumerated in C/D-layout order),
//     its packed weights sit in their own LDS region from kernel start.
#include <cstdint>
#include <cstdlib>
#include <algorithm>
#include <map>
#include <vector>

#include "mrx_common.h"
#include "rim_layer1_sb.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define RL_NT 512
#define RL_TH 8
#define RL_TW 32
#define RL_F 64
#define RL_PF 3  // operand prefetch distance (MFMA steps)

struct RimLayerArgs {
    const float* x;        // [B,Cin,H,W]
    const float* packed;   // conv chunks then ih block (mrx_rim_layer_pack)
    const float* b_conv;   // [F] or null
    const float* b_ih;     // [F] or null
    const float* hh;       // [F]
    const float* hprev;    // [B,F,H,W] or null
    float* hnew;           // [B,F,H,W]
    int B, Cin, H, W, tiles_x, ntiles;
    // input given as (eta, partial gradients) instead of x (Cin = 4): x = (eta.re, eta.im, post * sum_k part_k.re, post * sum_k part_k.im),
    // the last step of log_likelihood_gradient (rim_utils.py:61-67) done by the tile loader (mrx_rim_layer_indrnn_packed_llg)
    const float2* eta2;   // [B,H,W] complex or null
    const float2* part;   // [nparts][B][H][W] complex
    long long part_stride;
    int nparts;
    float post;
    int stagger;  // s_sleep argument for odd dispatch rounds (0 = off)
    unsigned long long* trace;  // debug only (env MRX_TRACE): 6 s_memtime stamps per workgroup
    int ablate;  // debug only (env MRX_ABLATE): 1 no h_prev loads, 2 no stores, 4 no main-loop MFMA, 8 no chunk staging, 16 no 1x1 GEMM
};

__host__ __device__ constexpr int rl_pad(int K, int DIL) { return DIL * (K - 1) / 2; }

// ---- weight packing ------------------------------------------------------------------------------------------------
// conv part : index (((q*TAPS + tap)*(CK/2) + pair)*2 + half)*F + o  <-  w[o][q*CK + 2*pair + half][tap]   (0 beyond Cin)
// ih part   : index ((ct*16 + r)*2 + half)*F + o                     <-  w_ih[o][32*ct + (r&3) + 8*(r>>2) + 4*half]
__global__ void k_rim_pack(const float* __restrict__ w, const float* __restrict__ w_ih, float* __restrict__ out, int Cin,
                           int taps, int CK, int nchunks) {
    const int conv_elems = nchunks * taps * CK * RL_F;
    const int total = conv_elems + RL_F * RL_F;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        float v = 0.f;
        if (i < conv_elems) {
            const int o = i % RL_F;
            int r = i / RL_F;
            const int half = r & 1;
            r >>= 1;
            const int pair = r % (CK / 2);
            r /= (CK / 2);
            const int tap = r % taps;
            const int q = r / taps;
            const int ci = q * CK + 2 * pair + half;
            if (ci < Cin) v = w[((long long)o * Cin + ci) * taps + tap];
        } else {
            const int j = i - conv_elems;
            const int o = j % RL_F;
            int r = j / RL_F;
            const int half = r & 1;
            r >>= 1;  // r = ct*16 + reg
            const int ct = r >> 4, reg = r & 15;
            const int c = 32 * ct + (reg & 3) + 8 * (reg >> 2) + 4 * half;
            v = w_ih[o * RL_F + c];
        }
        out[i] = v;
    }
}

static int rl_ck(int Cin) { return Cin <= 4 ? 4 : 8; }
// the first RIM layer's shape (5x5, <= 4 input channels): its pack carries a second section for k_rim_layer1_sb (rim_layer1_sb.hip)
static bool rl_sb_shape(int Cin, int k) { return Cin <= 4 && k == 5; }
static int64_t rl_fp32_pack_floats(int Cin, int k) {
    const int CK = rl_ck(Cin);
    return (int64_t)((Cin + CK - 1) / CK) * k * k * CK * RL_F + RL_F * RL_F;
}

extern "C" int64_t mrx_rim_layer_pack_floats(int Cin, int F, int k) {
    if (F != RL_F || Cin < 1 || k < 1) return -1;
    return rl_fp32_pack_floats(Cin, k) + (rl_sb_shape(Cin, k) ? MRX_L1SB_PACK_FLOATS : 0);
}

extern "C" int mrx_rim_layer_pack(const float* w_conv, const float* w_ih, float* packed, int Cin, int F, int k, void* stream) {
    MRX_REQUIRE(w_conv && w_ih && packed, MRX_EINVAL, "mrx_rim_layer_pack: null pointer");
    MRX_REQUIRE(F == RL_F, MRX_EUNSUP, "mrx_rim_layer_pack: hidden size %d (only %d is packed)", F, RL_F);
    MRX_REQUIRE(Cin >= 1 && k >= 1 && (k & 1), MRX_EINVAL, "mrx_rim_layer_pack: bad Cin=%d k=%d", Cin, k);
    const int CK = rl_ck(Cin);
    const int nchunks = (Cin + CK - 1) / CK;
    const int total = nchunks * k * k * CK * RL_F + RL_F * RL_F;
    hipLaunchKernelGGL(k_rim_pack, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_conv, w_ih, packed, Cin, k * k,
                       CK, nchunks);
    MRX_LAUNCH_CHECK();
    if (rl_sb_shape(Cin, k)) return mrx_l1sb_pack(w_conv, w_ih, packed + rl_fp32_pack_floats(Cin, k), Cin, (hipStream_t)stream);
    return MRX_OK;
}

// ---- the fused layer -----------------------------------------------------------------------------------------------
// LDS: two buffers, each [CK][PLANE] input tile + [TAPS][CK/2][2][F] weights.  Iteration q runs the matrix cores on buffer
// q&1 while the global loads of chunk q+1 are in flight; afterwards every wave writes its prefetched registers into the
// other buffer and the workgroup meets at ONE barrier per chunk.  After the last chunk the free buffer receives the packed
// 1x1 `ih` weights (prefetched through registers the same way).
template <int K, int DIL, int CK>
__global__ __launch_bounds__(RL_NT, 4) void k_rim_layer(RimLayerArgs a) {
    constexpr int PAD = rl_pad(K, DIL);
    constexpr int PH = RL_TH + 2 * PAD, PW = RL_TW + 2 * PAD, PLANE = PH * PW;
    constexpr int TAPS = K * K;
    constexpr int WCHUNK = TAPS * CK * RL_F;               // floats of packed weights per chunk
    constexpr int XSLOTS = (PLANE + RL_NT - 1) / RL_NT;    // input-tile elements per thread per channel
    constexpr int WVEC = (WCHUNK / 4 + RL_NT - 1) / RL_NT;  // float4 per thread per chunk
    constexpr int BUF = (CK * PLANE + WCHUNK) > RL_F * RL_F ? (CK * PLANE + WCHUNK) : RL_F * RL_F;
    constexpr int WIVEC = RL_F * RL_F / 4 / RL_NT;         // float4 of ih weights per thread (= 2)
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;

    // XCD-aware tile order: workgroup b runs on XCD b % 8; give each XCD a contiguous band of tiles so halo rows of
    // neighbouring tiles are served by the same L2 (speed only; any order is correct)
    const int tile = (int)mrx_xcd_band(blockIdx.x, a.ntiles);
    const int ty0 = tile / a.tiles_x;
    const int h0 = ty0 * RL_TH, w0 = (tile - ty0 * a.tiles_x) * RL_TW;
    const int b = blockIdx.y;
    const long long plane = (long long)a.H * a.W;
    const float* xb = a.x + (long long)b * a.Cin * plane;

    // per-thread staging slots (replicate border = clamp, conv_layers.py:72-76)
    int goff[XSLOTS];
#pragma unroll
    for (int s = 0; s < XSLOTS; ++s) {
        const int e = tid + s * RL_NT;
        const int ty = e / PW, tx = e - ty * PW;           // PW is a compile-time constant
        int gy = h0 + ty - PAD, gx = w0 + tx - PAD;
        gy = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy);
        gx = gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx);
        goff[s] = gy * a.W + gx;
    }
    const int nchunks = (a.Cin + CK - 1) / CK;

    float xr[CK][XSLOTS];
    float4 wr[WVEC];
    static_assert(WIVEC == 2, "ih weights: two float4 per thread");
    typedef float wi4_t __attribute__((ext_vector_type(4)));
    wi4_t wir0 = {0.f, 0.f, 0.f, 0.f}, wir1 = wir0;  // scalars, not an array: the array form stayed in scratch memory
    auto prefetch = [&](int q) {
        if (CK == 4 && a.eta2) {  // Cin = 4, one chunk: channels from eta and the partial coil sums (same order of additions as k_llg_combine)
#pragma unroll
            for (int s = 0; s < XSLOTS; ++s) {
                const int e = tid + s * RL_NT;
                float2 ev = make_float2(0.f, 0.f), sv = make_float2(0.f, 0.f);
                if (e < PLANE) {
                    const long long o = (long long)b * plane + goff[s];
                    ev = a.eta2[o];
                    sv = a.part[o];
                    for (int k = 1; k < a.nparts; ++k) {
                        const float2 v = a.part[(long long)k * a.part_stride + o];
                        sv.x += v.x;
                        sv.y += v.y;
                    }
                    sv.x *= a.post;
                    sv.y *= a.post;
                }
                xr[0][s] = ev.x;
                xr[1 % CK][s] = ev.y;
                xr[2 % CK][s] = sv.x;
                xr[3 % CK][s] = sv.y;
            }
        } else
#pragma unroll
        for (int ci = 0; ci < CK; ++ci) {
            const int gc = q * CK + ci;
#pragma unroll
            for (int s = 0; s < XSLOTS; ++s) {
                const int e = tid + s * RL_NT;
                xr[ci][s] = (e < PLANE && gc < a.Cin) ? xb[(long long)gc * plane + goff[s]] : 0.f;
            }
        }
        const float4* wsrc = reinterpret_cast<const float4*>(a.packed + (long long)q * WCHUNK);
#pragma unroll
        for (int v = 0; v < WVEC; ++v) {
            const int i = tid + v * RL_NT;
            wr[v] = (i < WCHUNK / 4) ? wsrc[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto commit = [&](float* buf) {  // prefetched registers -> LDS buffer
#pragma unroll
        for (int ci = 0; ci < CK; ++ci)
#pragma unroll
            for (int s = 0; s < XSLOTS; ++s) {
                const int e = tid + s * RL_NT;
                if (e < PLANE) buf[ci * PLANE + e] = xr[ci][s];
            }
        float4* wdst = reinterpret_cast<float4*>(buf + CK * PLANE);
#pragma unroll
        for (int v = 0; v < WVEC; ++v) {
            const int i = tid + v * RL_NT;
            if (i < WCHUNK / 4) wdst[i] = wr[v];
        }
    };
    auto prefetch_wi = [&]() {
        const wi4_t* src = reinterpret_cast<const wi4_t*>(a.packed + (long long)nchunks * WCHUNK);
        wir0 = src[tid];
        wir1 = src[tid + RL_NT];
    };

    // The two workgroups sharing a CU start together and would stay in lockstep (both staging, then both on the matrix
    // cores at half rate).  A one-time stagger of every other dispatch round lets one stage while the other computes;
    // the out-of-phase schedule is self-sustaining because the computing block gets the whole MFMA pipe meanwhile.
#define RL_STAMP(i)                                                                          \
    if (a.trace && tid == 0) a.trace[(long long)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = __builtin_readcyclecounter();
    RL_STAMP(0)
    if (a.stagger && ((blockIdx.x >> 8) & 1))
        for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(8);  // 512 cycles each
    prefetch(0);
    commit(smem_f);
    if (nchunks > 1)
        prefetch(1);
    else
        prefetch_wi();
    __syncthreads();
    RL_STAMP(1)

    // the accumulators start at the conv bias (loaded here, behind the staging latency) instead of adding it in the tail,
    // where 32 more live registers would not fit
    f32x16 acc[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = a.b_conv ? a.b_conv[ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi] : 0.f;

    for (int q = 0; q < nchunks; ++q) {
        const float* cur = smem_f + (q & 1) * BUF;
        float* oth = smem_f + ((q + 1) & 1) * BUF;
        const float* xw = cur + lhi * PLANE + wave * PW + l31;
        const float* ww = cur + CK * PLANE + lhi * RL_F + l31;
        if (!(a.ablate & 4)) {
            // Operand reads run RL_PF steps ahead of the MFMAs that consume them (register ring, all indices constant after
            // unrolling): a wave no longer serialises "ds_read -> wait -> 2 MFMAs" per step, so it can keep its SIMD's
            // matrix pipe busy on its own and the LDS round trip is covered by RL_PF * 128 cycles of MFMA work.
            constexpr int NS = TAPS * (CK / 2);
            float ra0[RL_PF + 1], ra1[RL_PF + 1], rb[RL_PF + 1];
#pragma unroll
            for (int s = 0; s < NS + RL_PF; ++s) {
                if (s < NS) {
                    const int tap = s / (CK / 2), pair = s % (CK / 2);
                    const int ky = tap / K, kx = tap % K;
                    rb[s % (RL_PF + 1)] = xw[(2 * pair) * PLANE + ky * DIL * PW + kx * DIL];
                    ra0[s % (RL_PF + 1)] = ww[(s * 2) * RL_F];
                    ra1[s % (RL_PF + 1)] = ww[(s * 2) * RL_F + 32];
                }
                if (s >= RL_PF) {
                    const int c = (s - RL_PF) % (RL_PF + 1);
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra0[c], rb[c], acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra1[c], rb[c], acc[1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);  // keep the reads RL_PF steps ahead (the scheduler would sink them)
            }
        }
        // fill the other buffer (its readers all passed the previous barrier), then one barrier per chunk
        if (q + 1 < nchunks) {
            if (!(a.ablate & 8)) commit(oth);
        } else {
            wi4_t* dst = reinterpret_cast<wi4_t*>(oth);
            dst[tid] = wir0;
            dst[tid + RL_NT] = wir1;
        }
        __syncthreads();
        if (q + 2 < nchunks) {
            if (!(a.ablate & 8)) prefetch(q + 2);
        } else if (q + 2 == nchunks)
            prefetch_wi();
    }
    const float* Wi = smem_f + (nchunks & 1) * BUF;  // [32][2][F], written during the last iteration
    RL_STAMP(2)

    // ---- g = ReLU(conv + b_conv), kept in registers; h = ReLU(Wih g + b_ih + hh * h_prev) -----------------------------
    const int oy = h0 + wave, ox = w0 + l31;
    const bool inside = oy < a.H && ox < a.W;
    const long long obase = (long long)b * RL_F * plane + (long long)oy * a.W + ox;
    // Wide epilogue (W % 4 == 0): the 64x32 result tile of this wave is transposed through LDS so that every lane owns
    // 4 consecutive pixels of one channel: 8 float4 loads of h_prev and 8 float4 stores per lane instead of 32 + 32 scalar
    // accesses (the scalar store tail is instruction-issue bound).  Lane L, iteration i: channel i*8 + L/8, pixels 4*(L%8)..+3.
    const bool wide = (a.W & 3) == 0;
    const int wch = lane >> 3, wpx = (lane & 7) * 4;
    const long long wbase = (long long)b * RL_F * plane + (long long)oy * a.W + w0 + wpx;
    const bool winside = oy < a.H && (w0 + wpx) < a.W;
    // h_prev arrives in two halves: channels 0-31 before the ih GEMM, channels 32-63 halfway through it, when the first
    // half of g is consumed and its 16 registers are free (all 32 up front overflowed the 128-register budget of 4 waves
    // per SIMD: 80 B/lane of scratch, ~40 MB of spill traffic per launch at 640x372)
    float4 hp4[8];
    const bool hp_live = wide && a.hprev && winside && !(a.ablate & 1);
    auto load_hp = [&](int i0) {
#pragma unroll
        for (int i = i0; i < i0 + 4; ++i) {
            hp4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (hp_live) hp4[i] = *reinterpret_cast<const float4*>(a.hprev + wbase + (long long)(i * 8 + wch) * plane);
        }
    };
    load_hp(0);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = acc[ct][r];
            acc[ct][r] = v > 0.f ? v : 0.f;
        }
    f32x16 acc2[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[ct][r] = 0.f;
    const float* wi = Wi + lhi * RL_F + l31;
    if (!(a.ablate & 16)) {
        float qa0[RL_PF + 1], qa1[RL_PF + 1];
#pragma unroll
        for (int s = 0; s < 32 + RL_PF; ++s) {
            if (s < 32) {
                qa0[s % (RL_PF + 1)] = wi[(s * 2) * RL_F];
                qa1[s % (RL_PF + 1)] = wi[(s * 2) * RL_F + 32];
            }
            if (s >= RL_PF) {
                const int t = s - RL_PF, c = t % (RL_PF + 1);
                acc2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(qa0[c], acc[t >> 4][t & 15], acc2[0], 0, 0, 0);
                acc2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(qa1[c], acc[t >> 4][t & 15], acc2[1], 0, 0, 0);
            }
            if (s == 16 + RL_PF) load_hp(4);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else
        load_hp(4);
    RL_STAMP(3)
    if (wide) {
        __syncthreads();  // every wave is done with Wi and the chunk buffers: LDS becomes 8 x [64][32] transpose tiles
        float* T = smem_f + wave * (RL_F * RL_TW);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                T[co * RL_TW + l31] = acc2[ct][r];
            }
        // same-wave LDS traffic is ordered; no workgroup barrier needed for a wave-private tile
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ch = i * 8 + wch;
            float4 v = *reinterpret_cast<const float4*>(T + ch * RL_TW + wpx);
            const float bi = a.b_ih ? a.b_ih[ch] : 0.f;
            const float hw = a.hh[ch];
            v.x = v.x + bi + hw * hp4[i].x;
            v.y = v.y + bi + hw * hp4[i].y;
            v.z = v.z + bi + hw * hp4[i].z;
            v.w = v.w + bi + hw * hp4[i].w;
            v.x = v.x > 0.f ? v.x : 0.f;
            v.y = v.y > 0.f ? v.y : 0.f;
            v.z = v.z > 0.f ? v.z : 0.f;
            v.w = v.w > 0.f ? v.w : 0.f;
            if (winside && (!(a.ablate & 2) || v.x == 12345.678f))
                *reinterpret_cast<float4*>(a.hnew + wbase + (long long)ch * plane) = v;
        }
    } else if (inside) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                float v = acc2[ct][r];
                if (a.b_ih) v += a.b_ih[co];
                if (a.hprev && !(a.ablate & 1)) v += a.hh[co] * a.hprev[obase + (long long)co * plane];
                v = v > 0.f ? v : 0.f;
                if (!(a.ablate & 2) || v == 12345.678f) a.hnew[obase + (long long)co * plane] = v;
            }
    }
    RL_STAMP(4)
    if (a.trace && tid == 0) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        a.trace[(long long)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + 5] = hwid;
        a.trace[(long long)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + 6] = xcc & 0xf;
    }
}

template <int K, int DIL, int CK>
static int launch_rim_layer(const RimLayerArgs& a_in, hipStream_t st) {
    constexpr int PAD = rl_pad(K, DIL);
    constexpr int PLANE = (RL_TH + 2 * PAD) * (RL_TW + 2 * PAD);
    constexpr size_t per_buf = ((size_t)CK * PLANE + (size_t)K * K * CK * RL_F) > (size_t)RL_F * RL_F
                                   ? ((size_t)CK * PLANE + (size_t)K * K * CK * RL_F)
                                   : (size_t)RL_F * RL_F;
    // two staging buffers; the wide epilogue re-uses the space as 8 wave-private [64][32] transpose tiles
    constexpr size_t lds_floats = 2 * per_buf > (size_t)(RL_NT / 64) * RL_F * RL_TW ? 2 * per_buf : (size_t)(RL_NT / 64) * RL_F * RL_TW;
    constexpr size_t lds = sizeof(float) * lds_floats;
    static_assert((CK * PLANE) % 4 == 0 && per_buf % 4 == 0, "weight regions must stay 16-byte aligned");
    static bool attr_done = false;  // once per instantiation: keeps launches legal under hipGraph capture
    if (lds > 48 * 1024 && !attr_done) {
        MRX_HIP(hipFuncSetAttribute((const void*)k_rim_layer<K, DIL, CK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    static unsigned long long* d_trace = nullptr;
    RimLayerArgs a = a_in;
    a.trace = nullptr;
    if (MRX_DEBUG_ENV("MRX_TRACE")) {
        if (!d_trace) (void)hipMalloc((void**)&d_trace, sizeof(unsigned long long) * 8 * 65536);
        (void)hipMemsetAsync(d_trace, 0, sizeof(unsigned long long) * 8 * 65536, st);
        a.trace = d_trace;
    }
    if ((MRX_DEBUG_ENV("MRX_TRACE") && atoi(MRX_DEBUG_ENV("MRX_TRACE")) >= 3)) {
        int nb = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_rim_layer<K, DIL, CK>, RL_NT, lds);
        fprintf(stderr, "[mrx] k_rim_layer<%d,%d,%d>: lds %zu B, occupancy API: %d blocks/CU\n", K, DIL, CK, lds, nb);
    }
    hipLaunchKernelGGL((k_rim_layer<K, DIL, CK>), dim3(a.ntiles, a.B), dim3(RL_NT), lds, st, a);
    MRX_LAUNCH_CHECK();
    if (a.trace && (MRX_DEBUG_ENV("MRX_TRACE") && atoi(MRX_DEBUG_ENV("MRX_TRACE")) >= 2)) {
        (void)hipStreamSynchronize(st);
        const int nb = a.ntiles * a.B;
        std::vector<unsigned long long> h((size_t)nb * 8);
        (void)hipMemcpy(h.data(), d_trace, sizeof(unsigned long long) * 8 * nb, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull, t1 = 0;
        double ph[4] = {0, 0, 0, 0};
        for (int i = 0; i < nb; ++i) {
            const unsigned long long* r = &h[(size_t)i * 8];
            if (r[0] < t0) t0 = r[0];
            if (r[4] > t1) t1 = r[4];
            for (int k = 0; k < 4; ++k) ph[k] += (double)(r[k + 1] - r[k]);
        }
        fprintf(stderr, "[mrx-trace] k_rim_layer<%d,%d,%d> %d blocks: span %llu cyc; mean per block: prologue %.0f main %.0f "
                        "1x1 %.0f epilogue %.0f (sum %.0f)\n", K, DIL, CK, nb, t1 - t0, ph[0] / nb, ph[1] / nb, ph[2] / nb, ph[3] / nb,
                (ph[0] + ph[1] + ph[2] + ph[3]) / nb);
        // start-time histogram: how many blocks start in each 10% of the span, and CU sharing
        int hist[10] = {0};
        for (int i = 0; i < nb; ++i) hist[(int)((double)(h[(size_t)i * 8] - t0) * 10 / (double)(t1 - t0 + 1))]++;
        fprintf(stderr, "[mrx-trace] start histogram:");
        for (int k = 0; k < 10; ++k) fprintf(stderr, " %d", hist[k]);
        fprintf(stderr, "\n");
        // co-residency: group blocks by (xcc, se, sh, cu) and list them in start order (per-XCC clocks are comparable)
        struct Rec { int blk; unsigned long long st, en; };
        std::map<unsigned, std::vector<Rec>> cus;
        for (int i = 0; i < nb; ++i) {
            const unsigned hw = (unsigned)h[(size_t)i * 8 + 5];
            const unsigned key = ((unsigned)h[(size_t)i * 8 + 6] << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15);
            cus[key].push_back({i, h[(size_t)i * 8], h[(size_t)i * 8 + 4]});
        }
        fprintf(stderr, "[mrx-trace] %zu distinct CUs\n", cus.size());
        int shown = 0;
        for (auto& kv : cus) {
            if (shown++ >= 4) break;
            auto& v = kv.second;
            std::sort(v.begin(), v.end(), [](const Rec& x, const Rec& y) { return x.st < y.st; });
            fprintf(stderr, "[mrx-trace] cu %06x:", kv.first);
            for (auto& r : v) fprintf(stderr, " blk%d[%llu..%llu]", r.blk, (r.st - v[0].st) / 1000, (r.en - v[0].st) / 1000);
            fprintf(stderr, "  (kcycles)\n");
        }
    }
    return MRX_OK;
}

// returns 1 when (F, k, dil) has a tuned instantiation
extern "C" int mrx_rim_layer_supported(int Cin, int F, int k, int dil) {
    if (F != RL_F || Cin < 1) return 0;
    return (k == 5 && dil == 1) || (k == 3 && dil == 2) || (k == 3 && dil == 1) || (k == 1 && dil == 1);
}

struct LlgSrc {
    const float2* eta;
    const float2* part;
    int nparts;
    float post;
};
static thread_local LlgSrc g_llg_src = {nullptr, nullptr, 0, 0.f};
static thread_local float* g_l1_xmax = nullptr;   // set by the _xmax entry points around the call
static thread_local int g_l1_cb8 = 0;             // set by mrx_rim_layer1_cb8 around the call: channel-blocked h_prev / h_new

extern "C" int mrx_rim_layer_indrnn_packed(const float* x, const float* packed, const float* b_conv, const float* b_ih,
                                           const float* hh, const float* h_prev, float* h_new, int B, int Cin, int F, int H,
                                           int W, int k, int dil, void* stream) {
    MRX_REQUIRE(x && packed && hh && h_new, MRX_EINVAL, "mrx_rim_layer_indrnn_packed: null pointer");
    MRX_REQUIRE(B >= 0 && Cin >= 1 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_rim_layer_indrnn_packed: bad dims");
    MRX_REQUIRE(mrx_rim_layer_supported(Cin, F, k, dil), MRX_EUNSUP,
                "mrx_rim_layer_indrnn_packed: no tuned kernel for F=%d k=%d dil=%d (use mrx_rim_layer_indrnn)", F, k, dil);
    MRX_REQUIRE(B <= 65535, MRX_EUNSUP, "mrx_rim_layer_indrnn_packed: batch %d too large", B);
    if (B == 0) return MRX_OK;
    RimLayerArgs a;
    a.eta2 = nullptr;
    a.part = nullptr;
    a.part_stride = 0;
    a.nparts = 0;
    a.post = 0.f;
    if (g_llg_src.eta) {  // set by mrx_rim_layer_indrnn_packed_llg around this call
        a.eta2 = g_llg_src.eta;
        a.part = g_llg_src.part;
        a.part_stride = (long long)B * H * W;
        a.nparts = g_llg_src.nparts;
        a.post = g_llg_src.post;
    }
    a.x = x;
    a.packed = packed;
    a.b_conv = b_conv;
    a.b_ih = b_ih;
    a.hh = hh;
    a.hprev = h_prev;
    a.hnew = h_new;
    a.B = B;
    a.Cin = Cin;
    a.H = H;
    a.W = W;
    a.tiles_x = mrx_cdiv(W, RL_TW);
    a.ntiles = a.tiles_x * mrx_cdiv(H, RL_TH);
    static const int ablate = MRX_DEBUG_ENV("MRX_ABLATE") ? atoi(MRX_DEBUG_ENV("MRX_ABLATE")) : 0;
    a.ablate = ablate;
    a.stagger = 0;
    hipStream_t st = (hipStream_t)stream;
    const bool small = rl_ck(Cin) == 4;
    if (rl_sb_shape(Cin, k) && dil == 1 && !ablate && !MRX_DEBUG_ENV("MRX_TRACE")) {
        if (mrx_arith() != MRX_ARITH_FP32) {         // (fp32: the fp32-MFMA kernel below, the cross-check of the split-operand ones)
            MrxL1sbArgs s;
            s.x = x, s.packed = packed + rl_fp32_pack_floats(Cin, k), s.b_conv = b_conv, s.b_ih = b_ih, s.hh = hh, s.hprev = h_prev, s.hnew = h_new;
            s.B = B, s.Cin = Cin, s.H = H, s.W = W, s.tiles_x = a.tiles_x, s.ntiles = a.tiles_x * mrx_cdiv(H, MRX_L1SB_TH);
            s.eta2 = a.eta2, s.part = a.part, s.part_stride = a.part_stride, s.nparts = a.nparts, s.post = a.post;
            s.xmax = reinterpret_cast<unsigned*>(g_l1_xmax);
            const int l1_f16 = mrx_arith() == MRX_ARITH_F16X2 ? 1 : 0;   // else the three-term bf16 form
            s.f16 = l1_f16;
            s.cb8 = g_l1_cb8;
            return mrx_l1sb_launch(s, st);
        }
    }
    MRX_REQUIRE(!g_l1_xmax, MRX_EUNSUP, "mrx_rim_layer_indrnn_packed_xmax: only the split-bf16 first-layer kernel keeps the output bound");
    MRX_REQUIRE(!g_l1_cb8, MRX_EUNSUP, "mrx_rim_layer1_cb8: only the two-term fp16 first-layer kernel writes channel-blocked states");
#define RL_CASE(KK, DD)                                                       \
    if (k == KK && dil == DD)                                                 \
        return small ? launch_rim_layer<KK, DD, 4>(a, st) : launch_rim_layer<KK, DD, 8>(a, st);
    RL_CASE(5, 1)
    RL_CASE(3, 2)
    RL_CASE(3, 1)
    RL_CASE(1, 1)
#undef RL_CASE
    MRX_REQUIRE(false, MRX_EUNSUP, "unreachable");
}

// ---- tuned final layer: F -> 2 conv (replicate pad) + eta update on the vector ALUs (rim_block.py:239-248) ---------------
// 16x32 pixel tile, one pixel per thread; both output channels share every input read; weights live in LDS as (w0, w1)
// pairs and are read with wave-uniform (broadcast) addresses.
#define RF_NT 512
#define RF_TH 16
#define RF_CK 16
struct RimFinalArgs {
    const float* h;     // [B,F,H,W]
    const float* w;     // [2,F,K,K]
    const float* bias;  // [2] or null
    const float* eta;   // [B,H,W,2] (k_rim_final4: null = nothing added)
    float* out;         // [B,H,W,2]
    int B, F, H, W, tiles_x, ntiles;
    int pad_zero;       // k_rim_final4 only: zero padding instead of the RIM's replicate padding
    unsigned long long* trace;  // debug only (env MRX_TRACE)
};
template <int K, int DIL>
__global__ __launch_bounds__(RF_NT) void k_rim_final(RimFinalArgs a) {
    constexpr int PAD = rl_pad(K, DIL);
    constexpr int PH = RF_TH + 2 * PAD, PW = RL_TW + 2 * PAD, PLANE = PH * PW;
    constexpr int TAPS = K * K;
    constexpr int XSLOTS = (PLANE + RF_NT - 1) / RF_NT;
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    float* Xs = smem_f;                                       // [RF_CK][PLANE]
    float2* W2 = reinterpret_cast<float2*>(Xs + RF_CK * PLANE);  // [F][TAPS] (w0, w1)
    const int tid = threadIdx.x;
    const int tile = (int)mrx_xcd_band(blockIdx.x, a.ntiles);
    const int ty0 = tile / a.tiles_x;
    const int h0 = ty0 * RF_TH, w0 = (tile - ty0 * a.tiles_x) * RL_TW;
    const int b = blockIdx.y;
    const long long plane = (long long)a.H * a.W;
    const float* hb = a.h + (long long)b * a.F * plane;
    const int Fpad = (a.F + RF_CK - 1) / RF_CK * RF_CK;
    for (int i = tid; i < Fpad * TAPS; i += RF_NT)
        W2[i] = i < a.F * TAPS ? make_float2(a.w[i], a.w[a.F * TAPS + i]) : make_float2(0.f, 0.f);
    int goff[XSLOTS];
#pragma unroll
    for (int s = 0; s < XSLOTS; ++s) {
        const int e = tid + s * RF_NT;
        const int ty = e / PW, tx = e - ty * PW;
        int gy = h0 + ty - PAD, gx = w0 + tx - PAD;
        gy = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy);
        gx = gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx);
        goff[s] = gy * a.W + gx;
    }
    float xr[RF_CK][XSLOTS];
    auto prefetch = [&](int c0) {
#pragma unroll
        for (int ci = 0; ci < RF_CK; ++ci)
#pragma unroll
            for (int s = 0; s < XSLOTS; ++s) {
                const int e = tid + s * RF_NT;
                xr[ci][s] = (e < PLANE && c0 + ci < a.F) ? hb[(long long)(c0 + ci) * plane + goff[s]] : 0.f;
            }
    };
    prefetch(0);
    const int tx = tid & 31, ty = tid >> 5;
    float acc0 = 0.f, acc1 = 0.f;
    for (int c0 = 0; c0 < a.F; c0 += RF_CK) {
        __syncthreads();
#pragma unroll
        for (int ci = 0; ci < RF_CK; ++ci)
#pragma unroll
            for (int s = 0; s < XSLOTS; ++s) {
                const int e = tid + s * RF_NT;
                if (e < PLANE) Xs[ci * PLANE + e] = xr[ci][s];
            }
        __syncthreads();
        if (c0 + RF_CK < a.F) prefetch(c0 + RF_CK);
        const float* xp = Xs + ty * PW + tx;
        const float2* wp = W2 + c0 * TAPS;
#pragma unroll 4
        for (int ci = 0; ci < RF_CK; ++ci) {
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                const int ky = tap / K, kx = tap % K;
                const float xv = xp[ci * PLANE + ky * DIL * PW + kx * DIL];
                const float2 wv = wp[ci * TAPS + tap];
                acc0 += wv.x * xv;
                acc1 += wv.y * xv;
            }
        }
    }
    const int oy = h0 + ty, ox = w0 + tx;
    if (oy >= a.H || ox >= a.W) return;
    const long long o = ((long long)b * a.H + oy) * a.W + ox;
    const float2 e = reinterpret_cast<const float2*>(a.eta)[o];
    if (a.bias) {
        acc0 += a.bias[0];
        acc1 += a.bias[1];
    }
    reinterpret_cast<float2*>(a.out)[o] = make_float2(e.x + acc0, e.y + acc1);
}

template <int K, int DIL>
static int launch_rim_final(const RimFinalArgs& a, hipStream_t st) {
    constexpr int PAD = rl_pad(K, DIL);
    constexpr int PLANE = (RF_TH + 2 * PAD) * (RL_TW + 2 * PAD);
    const size_t lds = sizeof(float) * ((size_t)RF_CK * PLANE + 2 * (size_t)((a.F + RF_CK - 1) / RF_CK * RF_CK) * K * K);
    MRX_REQUIRE(lds <= 160 * 1024, MRX_EUNSUP, "rim_final: %zu bytes of LDS", lds);
    static size_t attr_bytes = 0;
    if (lds > 48 * 1024 && attr_bytes < lds) {
        MRX_HIP(hipFuncSetAttribute((const void*)k_rim_final<K, DIL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_bytes = lds;
    }
    hipLaunchKernelGGL((k_rim_final<K, DIL>), dim3(a.ntiles, a.B), dim3(RF_NT), lds, st, a);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- final 3x3 conv, W % 4 == 0: four pixels per thread on the vector ALUs --------------------------------------------------
// Cout = 2 leaves the matrix cores idle (2 of 16 rows), so the FMAs run on the VALU, whose cost is the instruction count:
// 18 FMAs per (pixel, channel) are irreducible, everything else is overhead.  A thread owns 4 adjacent pixels (8 accumulators) so
// one patch row is three LDS reads for 24 FMAs; the four waves of a workgroup share the 8x32 tile and split the channels
// (wave w takes channels 2w, 2w+1 of every chunk of 8), their partial sums meet in LDS at the end.  The raw tile arrives by
// LDS-DMA (global_load_lds_dwordx4, aligned float4 groups starting 4 columns left of the tile; rows clamped at the source, the
// column border fixed up after the read in the two border tile columns only); weights are (w0, w1) pairs broadcast from LDS.
#define RF4_NT 256
#define RF4_TH 8
#define RF4_TW 32
#define RF4_XS 40                       // tile row: w0-4 .. w0+35
#define RF4_ROWS (RF4_TH + 2)
#define RF4_PLANE4 128                  // float4 slots per channel plane (10 rows x 10 = 100 used): 2 DMA instructions
#define RF4_RING 4                      // channel planes in flight per wave
#define RF4_BUF (4 * RF4_RING * RF4_PLANE4 * 4 / 2)  // floats of half the staging area (4 waves x ring)
typedef float rf4_f2 __attribute__((ext_vector_type(2)));
typedef float rf4_f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(RF4_NT, 4) void k_rim_final4(RimFinalArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    float* Wp = smem_f + 2 * RF4_BUF;  // [F][9] pairs (w[0][ci][tap], w[1][ci][tap])
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile = (int)mrx_xcd_band(blockIdx.x, a.ntiles);
    const int ty0 = tile / a.tiles_x;
    const int h0 = ty0 * RF4_TH, w0 = (tile - ty0 * a.tiles_x) * RF4_TW;
    const int b = blockIdx.y;
    const long long plane = (long long)a.H * a.W;
    const float* hb = a.h + (long long)b * a.F * plane;
    unsigned off[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        int sl = m * 64 + lane;
        sl = sl < RF4_ROWS * (RF4_XS / 4) ? sl : RF4_ROWS * (RF4_XS / 4) - 1;
        const int ry = sl / (RF4_XS / 4), c4 = sl - ry * (RF4_XS / 4);
        int gy = h0 + ry - 1, gx = w0 - 4 + 4 * c4;
        gy = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy);
        gx = gx < 0 ? 0 : (gx > a.W - 4 ? a.W - 4 : gx);
        off[m] = (unsigned)(gy * a.W + gx) * 4u;
    }
    // wave-private ring of RF4_RING channel planes: wave w streams channels w, w+4, w+8, ... three channels ahead of the one it
    // is consuming, waiting on its own vmcnt only -- no workgroup barrier inside the channel loop, so the waves of the (up to four)
    // co-resident workgroups drift apart and cover each other's memory latency
    float* ring = smem_f + wave * (RF4_RING * RF4_PLANE4 * 4);
    auto dma = [&](int k) {  // k-th channel of this wave -> ring slot k % RF4_RING
        const char* src = reinterpret_cast<const char*>(hb + (long long)(wave + 4 * k) * plane);
        float* dst = ring + (k % RF4_RING) * RF4_PLANE4 * 4;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off[0]),
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
        if (lane < RF4_ROWS * (RF4_XS / 4) - 64)  // the plane has 100 float4: the second copy moves 36 (lanes write base + 16 lane)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + off[1]),
                                             (__attribute__((address_space(3))) void*)(dst + 256), 16, 0, 0);
    };
    const int nch = a.F / 4;  // channels per wave
    unsigned long long t_s = 0, t_w = 0, t_c = 0, t_x = 0;
    if (a.trace) t_s = __builtin_readcyclecounter();
#pragma unroll
    for (int k = 0; k < RF4_RING - 1; ++k)
        if (k < nch) dma(k);
    for (int i = tid; i < a.F * 9; i += RF4_NT) {
        Wp[2 * i] = a.w[i];
        Wp[2 * i + 1] = a.w[a.F * 9 + i];
    }
    const int tx = lane & 7, ty = lane >> 3;
    const int x0 = w0 + 4 * tx;                         // first of this thread's 4 pixels
    const bool fix_l = x0 == 0, fix_r = x0 + 4 >= a.W;  // replicate border columns (conv_layers.py:72-76)
    const bool border = w0 == 0 || w0 + RF4_TW >= a.W;
    // zero padding: rows of the 3-row patch outside the image (the DMA fetched the clamped row instead) contribute nothing
    const bool zrows = a.pad_zero && (h0 == 0 || h0 + RF4_TH >= a.H);
    unsigned rowok = 7u;
    if (zrows) {
        rowok = 0;
#pragma unroll
        for (int r = 0; r < 3; ++r) rowok |= (h0 + ty - 1 + r >= 0 && h0 + ty - 1 + r < a.H) ? (1u << r) : 0u;
    }
    float acc[2][4];
#pragma unroll
    for (int co = 0; co < 2; ++co)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[co][i] = 0.f;
    __syncthreads();  // weights staged
    for (int k = 0; k < nch; ++k) {
        // the ring slot refilled below was read by iteration k - 1, whose LDS reads have returned (their FMAs are issued in order)
        if (a.trace) t_x = __builtin_readcyclecounter();
        if (k + RF4_RING - 1 < nch) {
            dma(k + RF4_RING - 1);
            if (a.trace) t_c -= __builtin_readcyclecounter() - t_x;  // (issue cost goes to t_w below; t_c accumulates the rest)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (RF4_RING - 1)) : "memory");  // all but the newest 3 channels landed
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (a.trace) {
            const unsigned long long t = __builtin_readcyclecounter();
            t_w += t - t_x;
        }
        const float* xp = ring + (k % RF4_RING) * RF4_PLANE4 * 4 + ty * RF4_XS + 4 * tx;
        const float* wp = Wp + (wave + 4 * k) * 18;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const float* row = xp + r * RF4_XS;
            const rf4_f4 v1 = *reinterpret_cast<const rf4_f4*>(row + 4);
            float v[6] = {row[3], v1[0], v1[1], v1[2], v1[3], row[8]};  // image columns x0-1 .. x0+4
            if (border) {
                if (fix_l) v[0] = a.pad_zero ? 0.f : v[1];
                if (fix_r) v[5] = a.pad_zero ? 0.f : v[4];  // W % 4 == 0: the last image column is this thread's 4th pixel
            }
            if (zrows && !((rowok >> r) & 1u)) {
#pragma unroll
                for (int i = 0; i < 6; ++i) v[i] = 0.f;
            }
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const rf4_f2 wv = *reinterpret_cast<const rf4_f2*>(wp + (r * 3 + kx) * 2);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[0][i] += wv[0] * v[i + kx];
                    acc[1][i] += wv[1] * v[i + kx];
                }
            }
        }
    }
    if (a.trace && lane == 0) {
        unsigned long long* tr = a.trace + ((long long)(blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave) * 4;
        tr[0] = __builtin_readcyclecounter() - t_s;  // whole channel loop incl. prologue
        tr[1] = t_w;                                  // DMA issue + wait for the channel to land
        tr[2] = (unsigned long long)(-(long long)t_c);  // of which DMA issue
    }
    __syncthreads();  // all tiles consumed: LDS becomes the partial-sum exchange [wave 1..3][lane][8]
    float* Rx = smem_f;
    if (wave > 0) {
        rf4_f4* dst = reinterpret_cast<rf4_f4*>(Rx + ((wave - 1) * 64 + lane) * 8);
        dst[0] = (rf4_f4){acc[0][0], acc[1][0], acc[0][1], acc[1][1]};
        dst[1] = (rf4_f4){acc[0][2], acc[1][2], acc[0][3], acc[1][3]};
    }
    __syncthreads();
    if (wave > 0) return;
    const int oy = h0 + ty;
    if (oy >= a.H || x0 >= a.W) return;
    rf4_f4 s0 = (rf4_f4){acc[0][0], acc[1][0], acc[0][1], acc[1][1]}, s1 = (rf4_f4){acc[0][2], acc[1][2], acc[0][3], acc[1][3]};
#pragma unroll
    for (int wv = 0; wv < 3; ++wv) {
        const rf4_f4* src = reinterpret_cast<const rf4_f4*>(Rx + (wv * 64 + lane) * 8);
        s0 += src[0];
        s1 += src[1];
    }
    const long long o = (((long long)b * a.H + oy) * a.W + x0) * 2;
    const float b0 = a.bias ? a.bias[0] : 0.f, b1 = a.bias ? a.bias[1] : 0.f;
    const rf4_f4 bb = (rf4_f4){b0, b1, b0, b1};
    if (a.eta) {
        const rf4_f4 e0 = *reinterpret_cast<const rf4_f4*>(a.eta + o), e1 = *reinterpret_cast<const rf4_f4*>(a.eta + o + 4);
        *reinterpret_cast<rf4_f4*>(a.out + o) = e0 + (s0 + bb);
        *reinterpret_cast<rf4_f4*>(a.out + o + 4) = e1 + (s1 + bb);
    } else {
        *reinterpret_cast<rf4_f4*>(a.out + o) = s0 + bb;
        *reinterpret_cast<rf4_f4*>(a.out + o + 4) = s1 + bb;
    }
}

static int launch_rim_final4(RimFinalArgs a, hipStream_t st) {
    a.tiles_x = mrx_cdiv(a.W, RF4_TW);
    a.ntiles = a.tiles_x * mrx_cdiv(a.H, RF4_TH);
    const size_t lds = sizeof(float) * (2 * RF4_BUF + (size_t)a.F * 18);
    MRX_REQUIRE(lds <= 64 * 1024, MRX_EUNSUP, "rim_final: %zu bytes of LDS", lds);
    static size_t attr_bytes = 0;
    if (lds > 48 * 1024 && attr_bytes < lds) {
        MRX_HIP(hipFuncSetAttribute((const void*)k_rim_final4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_bytes = lds;
    }
    static unsigned long long* d_trace = nullptr;
    a.trace = nullptr;
    if (MRX_DEBUG_ENV("MRX_TRACE")) {
        if (!d_trace) (void)hipMalloc((void**)&d_trace, sizeof(unsigned long long) * 16 * 65536);
        a.trace = d_trace;
    }
    hipLaunchKernelGGL(k_rim_final4, dim3(a.ntiles, a.B), dim3(RF4_NT), lds, st, a);
    MRX_LAUNCH_CHECK();
    if (a.trace && (MRX_DEBUG_ENV("MRX_TRACE") && atoi(MRX_DEBUG_ENV("MRX_TRACE")) >= 2)) {
        (void)hipStreamSynchronize(st);
        const int nw = a.ntiles * a.B * 4;
        std::vector<unsigned long long> h((size_t)nw * 4);
        (void)hipMemcpy(h.data(), d_trace, sizeof(unsigned long long) * 4 * nw, hipMemcpyDeviceToHost);
        double tot = 0, wait = 0, issue = 0;
        for (int i = 0; i < nw; ++i) {
            tot += (double)h[(size_t)i * 4];
            wait += (double)h[(size_t)i * 4 + 1];
            issue += (double)h[(size_t)i * 4 + 2];
        }
        fprintf(stderr, "[mrx-trace] k_rim_final4 %d waves: mean channel loop %.0f cyc, of which DMA issue + landing wait %.0f (issue %.0f)\n", nw,
                tot / nw, wait / nw, issue / nw);
    }
    return MRX_OK;
}

// returns MRX_EUNSUP (without setting an error the caller must report) when no tuned instantiation exists
int mrx_rim_final_tuned(const float* h, const float* w, const float* bias, const float* eta, float* eta_out, int B, int F,
                        int H, int W, int k, int dil, hipStream_t st, int* handled) {
    *handled = 0;
    if (!((k == 3 && dil == 1) || (k == 1 && dil == 1) || (k == 3 && dil == 2) || (k == 5 && dil == 1))) return MRX_OK;
    if ((RF_CK * (RF_TH + 2 * rl_pad(k, dil)) * (RL_TW + 2 * rl_pad(k, dil))) % 2) return MRX_OK;
    RimFinalArgs a;
    a.trace = nullptr;
    a.pad_zero = 0;
    a.h = h;
    a.w = w;
    a.bias = bias;
    a.eta = eta;
    a.out = eta_out;
    a.B = B;
    a.F = F;
    a.H = H;
    a.W = W;
    a.tiles_x = mrx_cdiv(W, RL_TW);
    a.ntiles = a.tiles_x * mrx_cdiv(H, RF_TH);
    *handled = 1;
    if (k == 3 && dil == 1 && (W & 3) == 0 && W >= 8 && F % 4 == 0 && (long long)H * W < (1ll << 30) &&
        (((uintptr_t)h | (uintptr_t)eta | (uintptr_t)eta_out) & 15) == 0)
        return launch_rim_final4(a, st);
    if (k == 3 && dil == 1) return launch_rim_final<3, 1>(a, st);
    if (k == 1 && dil == 1) return launch_rim_final<1, 1>(a, st);
    if (k == 3 && dil == 2) return launch_rim_final<3, 2>(a, st);
    return launch_rim_final<5, 1>(a, st);
}

// permute(conv3x3(h), (0, 2, 3, 1)) for a convolution into 2 channels, zero or replicate padding, as [B,H,W,2] (one complex image):
// the tail of the CascadeNet / VSNet / Recurrent VarNet regularisers (conv/conv2d.py:36-43 + ccnn_block.py:133; conv2gru.py:158-162
// + recurrentvarnet.py:221) on the 4-pixels-per-thread kernel of the RIM's final layer.  Returns MRX_EUNSUP for other shapes.
extern "C" int mrx_conv_to_complex(const float* h, const float* w, const float* bias, float* out, int B, int F, int H, int W, int k,
                                   int dil, int pad_mode, void* stream) {
    MRX_REQUIRE(h && w && out, MRX_EINVAL, "mrx_conv_to_complex: null pointer");
    MRX_REQUIRE(B >= 0 && F >= 1 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_conv_to_complex: bad dims");
    MRX_REQUIRE(pad_mode == MRX_PAD_ZERO || pad_mode == MRX_PAD_REPLICATE, MRX_EINVAL, "mrx_conv_to_complex: pad mode %d", pad_mode);
    MRX_REQUIRE(k == 3 && dil == 1 && (W & 3) == 0 && W >= 8 && F % 4 == 0 && (long long)H * W < (1ll << 30) && B <= 65535 &&
                    (((uintptr_t)h | (uintptr_t)out) & 15) == 0,
                MRX_EUNSUP, "mrx_conv_to_complex: shape not covered (3x3, dilation 1, W %% 4 == 0, F %% 4 == 0, 16-byte aligned)");
    if (B == 0) return MRX_OK;
    RimFinalArgs a;
    a.trace = nullptr;
    a.pad_zero = pad_mode == MRX_PAD_ZERO;
    a.h = h;
    a.w = w;
    a.bias = bias;
    a.eta = nullptr;
    a.out = out;
    a.B = B;
    a.F = F;
    a.H = H;
    a.W = W;
    return launch_rim_final4(a, (hipStream_t)stream);
}

// The fused layer on the output of log_likelihood_gradient without materialising it: input channels (eta.re, eta.im, grad.re, grad.im)
// with grad = inv_sigma2 * sum of the `nparts` coil-chunk partials mrx_llg_hinv_parts left in `part` (rim_utils.py:61-67 +
// conv_layers.py:121-123 + rnn_cells.py:384-391).  Cin is 4 by construction.
extern "C" int mrx_rim_layer_indrnn_packed_llg(const float* eta, const float* part, int nparts, float inv_sigma2, const float* packed,
                                               const float* b_conv, const float* b_ih, const float* hh, const float* h_prev,
                                               float* h_new, int B, int F, int H, int W, int k, int dil, void* stream) {
    MRX_REQUIRE(eta && part && nparts >= 1, MRX_EINVAL, "mrx_rim_layer_indrnn_packed_llg: bad argument");
    g_llg_src = {(const float2*)eta, (const float2*)part, nparts, inv_sigma2};
    const int rc = mrx_rim_layer_indrnn_packed(eta /* non-null placeholder, not read */, packed, b_conv, b_ih, hh, h_prev, h_new, B, 4, F, H,
                                               W, k, dil, stream);
    g_llg_src = {nullptr, nullptr, 0, 0.f};
    return rc;
}

// The two calls above that also fold the maximum of their (non-negative) outputs into *xmax with an atomic max -- never reset here: a running
// upper bound of max |h_new|, which mrx_rim_layer2_f16 takes its operand scale from.  Only the split-bf16 kernel (Cin <= 4, 5x5, 64 features)
// does it: mrx_rim_layer1_xmax_supported.
extern "C" int mrx_rim_layer1_xmax_supported(int Cin, int F, int k, int dil) {
    return (F == 64 && rl_sb_shape(Cin, k) && dil == 1 && mrx_arith() != MRX_ARITH_FP32 && !MRX_DEBUG_ENV("MRX_ABLATE") && !MRX_DEBUG_ENV("MRX_TRACE")) ? 1 : 0;
}
extern "C" int mrx_rim_layer_indrnn_packed_xmax(const float* x, const float* packed, const float* b_conv, const float* b_ih, const float* hh,
                                                const float* h_prev, float* h_new, float* xmax, int B, int Cin, int F, int H, int W, int k, int dil,
                                                void* stream) {
    MRX_REQUIRE(xmax, MRX_EINVAL, "mrx_rim_layer_indrnn_packed_xmax: null pointer");
    g_l1_xmax = xmax;
    const int rc = mrx_rim_layer_indrnn_packed(x, packed, b_conv, b_ih, hh, h_prev, h_new, B, Cin, F, H, W, k, dil, stream);
    g_l1_xmax = nullptr;
    return rc;
}
extern "C" int mrx_rim_layer_indrnn_packed_llg_xmax(const float* eta, const float* part, int nparts, float inv_sigma2, const float* packed,
                                                    const float* b_conv, const float* b_ih, const float* hh, const float* h_prev, float* h_new,
                                                    float* xmax, int B, int F, int H, int W, int k, int dil, void* stream) {
    MRX_REQUIRE(xmax, MRX_EINVAL, "mrx_rim_layer_indrnn_packed_llg_xmax: null pointer");
    g_l1_xmax = xmax;
    const int rc = mrx_rim_layer_indrnn_packed_llg(eta, part, nparts, inv_sigma2, packed, b_conv, b_ih, hh, h_prev, h_new, B, F, H, W, k, dil, stream);
    g_l1_xmax = nullptr;
    return rc;
}

// The first layer of a RIM step on CHANNEL-BLOCKED hidden states (h_prev, h_new: [B][8][H][W][8]; mrx_cb8_convert): input either x [B,Cin<=4,H,W]
// (eta NULL) or (eta, coil-group partial sums) as in mrx_rim_layer_indrnn_packed_llg; 5x5 convolution into 64 features + 1x1 IndRNN cell;
// the maximum of the outputs is folded into *xmax (never reset here) for mrx_rim_layer2_f16_cb8's operand scale.  Bit-identical to the NCHW form.
extern "C" int mrx_rim_layer1_cb8(const float* x, int Cin, const float* eta, const float* part, int nparts, float inv_sigma2, const float* packed,
                                  const float* b_conv, const float* b_ih, const float* hh, const float* h_prev, float* h_new, float* xmax, int B,
                                  int H, int W, void* stream) {
    MRX_REQUIRE(xmax && (x || eta), MRX_EINVAL, "mrx_rim_layer1_cb8: null pointer");
    MRX_REQUIRE(mrx_rim_layer1_xmax_supported(eta ? 4 : Cin, 64, 5, 1) && mrx_arith() == MRX_ARITH_F16X2, MRX_EUNSUP,
                "mrx_rim_layer1_cb8: needs the two-term fp16 first-layer kernel (MRIDC_AMD_ARITH=f16x2)");
    g_l1_cb8 = 1;
    const int rc = eta ? mrx_rim_layer_indrnn_packed_llg_xmax(eta, part, nparts, inv_sigma2, packed, b_conv, b_ih, hh, h_prev, h_new, xmax, B, 64, H, W, 5, 1, stream)
                       : mrx_rim_layer_indrnn_packed_xmax(x, packed, b_conv, b_ih, hh, h_prev, h_new, xmax, B, Cin, 64, H, W, 5, 1, stream);
    g_l1_cb8 = 0;
    return rc;
}
