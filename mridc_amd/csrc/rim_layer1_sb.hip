// rim_layer1_sb.hip -- the first RIM layer (ConvNonlinear 5x5, <= 4 -> 64, replicate padding, ReLU + IndRNNCell 1x1 64 -> 64; reference
// models/rim/conv_layers.py:121-123 + rnn_cells.py:384-391) with fp32 results on the bf16 matrix pipe.
//
// On gfx950 the fp32-input MFMA shares the vector ALU (finding 1) and runs at 1/16 of the bf16 rate, so the fp32 kernel
// (k_rim_layer<5,1,4>) spends 38 k of its 56 k cycles per workgroup inside MFMAs while its HBM phase (h_prev in, h out: 122 MB) waits.  Here
// every fp32 operand is written as the exact sum of three bf16 terms,
//     x = x1 + x2 + x3,   x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)        (8 + 8 + 8 significand bits; both differences exact),
// and a product keeps the six term pairs of weight 2^-16 and above:
//     w x = w1 x1 + (w1 x2 + w2 x1) + (w1 x3 + w2 x2 + w3 x1) + O(2^-24 |w x|).
// Each pair is a v_mfma_f32_32x32x16_bf16 (bf16 products are exact in fp32; accumulation in fp32), so the result carries fp32 round-off --
// the parity tolerances of the fp32 kernel apply unchanged -- at 6/16 of the fp32 matrix time, and the bf16 MFMA co-issues with the vector ALU.
// The layer becomes what its bytes say it is: HBM-bound.
//
//   * one persistent workgroup per CU, 16 waves (4 per SIMD, 128 registers each); the weights -- split once per weight version
//     (mrx_rim_layer_pack) into the exact A-operand lane order, 66 KB -- are copied into LDS once per workgroup, followed by the only barrier;
//   * a wave's unit of work is one image row x 32 pixels x 64 couts (2 accumulators); workgroups walk 16 x 32 tiles (XCD-banded), every wave
//     runs its units start to finish on its own: patch -> LDS (wave-private), conv, 1x1, epilogue.  With four independent waves per SIMD at
//     different points of that sequence, one wave's memory phase runs under the others' MFMAs;
//   * conv GEMM: k = (tap, channel), 16 per MFMA: lower half-wave taps 4s, 4s+1, upper half-wave taps 4s+2, 4s+3, four channels each (taps
//     25..27 carry zero weights); the wave's 5 x 36 patch sits in LDS as three bf16 term planes [term][pixel][4 channels] (8 B per pixel: one
//     ds_read_b64 per tap), split once by the loader (which also finishes log_likelihood_gradient when handed (eta, partial sums));
//   * ReLU(conv + b) stays in registers; its three terms are formed eight channels at a time, right before the 1x1 GEMM step that consumes
//     them (the contraction index follows the C/D register order, as in the fp32 kernel);
//   * h_prev (lane = pixel: 128-byte rows per wave instruction) is requested in four groups of eight registers as the 1x1 steps free them,
//     and consumed by the epilogue; no LDS transpose.
// Measured at 640 x 372 (tools/probe/layer1_sb.py): 44-47 us against 60 us for the fp32-MFMA kernel, error against float64 1.8e-7 (fp32
// kernel: 1.7e-7).  Of that time the stores cost 14 us (29-30 us without them): the 61 MB write-through stream is what the layer waits for.
#include <cstdlib>

#include "mrx_common.h"
#include "rim_layer1_sb.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define SB_NT 1024
#define SB_TH 16
#define SB_TW 32
#define SB_F 64
#define SB_K 5
#define SB_PAD 2
#define SB_PH (SB_TH + 2 * SB_PAD)
#define SB_PW (SB_TW + 2 * SB_PAD)
#define SB_NPIX (SB_PH * SB_PW)
#define SB_KS 7                             // conv MFMA steps: 28 taps x 4 channels / 16
#define SB_KS2 4                            // 1x1 MFMA steps: 64 channels / 16
#define SB_WCONV (SB_KS * 3 * 2 * 64)       // 16-byte A operands of the conv part
#define SB_WIH (SB_KS2 * 3 * 2 * 64)
static_assert((SB_WCONV + SB_WIH) * 4 == MRX_L1SB_PACK_FLOATS, "pack size");

__device__ __forceinline__ unsigned sb_pk(float lo, float hi) {   // two fp32 -> packed bf16 pair (lo in bits 0-15), round to nearest even
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// (a, b) -> the three bf16 terms of each, packed pairwise
__device__ __forceinline__ void sb_split2(float a, float b, unsigned& p1, unsigned& p2, unsigned& p3) {
    p1 = sb_pk(a, b);
    float ra = a - __uint_as_float(p1 << 16), rb = b - __uint_as_float(p1 & 0xffff0000u);
    p2 = sb_pk(ra, rb);
    ra -= __uint_as_float(p2 << 16);
    rb -= __uint_as_float(p2 & 0xffff0000u);
    p3 = sb_pk(ra, rb);
}

// channel of conv accumulator register R = 16 ct + r in lane half `half` (v_mfma_f32_32x32 C/D layout)
__host__ __device__ constexpr int sb_chan(int R, int half) { return 32 * (R >> 4) + (R & 3) + 8 * ((R & 15) >> 2) + 4 * half; }

// ---- weight packing --------------------------------------------------------------------------------------------------------------------
// conv: out[((s*3 + t)*2 + blk)*64 + lane][j] = term_t( w[32 blk + lane%32][ci = j & 3][tap = 4 s + 2 (lane/32) + (j >> 2)] )   (0 for tap >= 25, ci >= Cin)
// ih  : out[SB_WCONV + ((s*3 + t)*2 + blk)*64 + lane][j] = term_t( w_ih[32 blk + lane%32][sb_chan(8 s + j, lane/32)] )
__global__ void k_l1sb_pack(const float* __restrict__ w, const float* __restrict__ w_ih, u32x4* __restrict__ out, int Cin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= SB_WCONV + SB_WIH) return;
    const bool conv = i < SB_WCONV;
    int r = conv ? i : i - SB_WCONV;
    const int lane = r & 63;
    r >>= 6;
    const int blk = r & 1;
    r >>= 1;
    const int t = r % 3, s = r / 3;
    const int o = 32 * blk + (lane & 31), half = lane >> 5;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (conv) {
            const int ci = j & 3, tap = 4 * s + 2 * half + (j >> 2);
            v[j] = (tap < SB_K * SB_K && ci < Cin) ? w[((long long)o * Cin + ci) * (SB_K * SB_K) + tap] : 0.f;
        } else
            v[j] = w_ih[o * SB_F + sb_chan(8 * s + j, half)];
    }
    unsigned p[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned p1, p2, p3;
        sb_split2(v[2 * q], v[2 * q + 1], p1, p2, p3);
        p[q] = t == 0 ? p1 : (t == 1 ? p2 : p3);
    }
    out[i] = u32x4{p[0], p[1], p[2], p[3]};
}

int mrx_l1sb_pack(const float* w_conv, const float* w_ih, float* packed, int Cin, hipStream_t st) {
    hipLaunchKernelGGL(k_l1sb_pack, dim3((SB_WCONV + SB_WIH + 255) / 256), dim3(256), 0, st, w_conv, w_ih, reinterpret_cast<u32x4*>(packed), Cin);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// the six term pairs of weight >= 2^-16, smallest first
#define SB_MFMA6(ACC, A1, A2, A3, B1, B2, B3)                                          \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A3, B1, ACC, 0, 0, 0);               \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B3, ACC, 0, 0, 0);               \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2, B2, ACC, 0, 0, 0);               \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A2, B1, ACC, 0, 0, 0);               \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B2, ACC, 0, 0, 0);               \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A1, B1, ACC, 0, 0, 0);
// the same for both cout blocks, smallest first, the two accumulators alternating
#define SB_MFMA12(ACC, A, B1, B2, B3)                                                             \
    ACC[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0][2], B1, ACC[0], 0, 0, 0);               \
    ACC[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1][2], B1, ACC[1], 0, 0, 0);               \
    ACC[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0][0], B3, ACC[0], 0, 0, 0);               \
    ACC[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1][0], B3, ACC[1], 0, 0, 0);               \
    ACC[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0][1], B2, ACC[0], 0, 0, 0);               \
    ACC[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1][1], B2, ACC[1], 0, 0, 0);               \
    ACC[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0][1], B1, ACC[0], 0, 0, 0);               \
    ACC[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1][1], B1, ACC[1], 0, 0, 0);               \
    ACC[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0][0], B2, ACC[0], 0, 0, 0);               \
    ACC[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1][0], B2, ACC[1], 0, 0, 0);               \
    ACC[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0][0], B1, ACC[0], 0, 0, 0);               \
    ACC[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1][0], B1, ACC[1], 0, 0, 0);

#define SB_PROWS SB_K                       // a wave's patch: 5 rows x 36 pixels
#define SB_PPIX (SB_PROWS * SB_PW)          // 180
#define SB_PSTR 184                         // term-plane stride in the wave's LDS patch
#define SB_PSLOT 3                          // patch pixels per lane

// One persistent workgroup per CU, 16 waves (4 per SIMD), weights staged once.  After the single barrier every wave walks its own units
// (image row x 32 pixels x 64 channels) start to finish: patch -> LDS (wave-private), conv, 1x1, epilogue.  Nothing is prefetched across
// units: with four independent waves per SIMD at different points of that sequence, one wave's memory phase runs under the others' MFMAs.
__global__ __launch_bounds__(SB_NT, 1) void k_rim_layer1_sb(MrxL1sbArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_sb[];
    u32x4* Wl = reinterpret_cast<u32x4*>(smem_sb);                                        // [SB_WCONV + SB_WIH] A operands
    float* tabl = reinterpret_cast<float*>(smem_sb + (SB_WCONV + SB_WIH) * 16);           // hh, b_conv, b_ih in register order [R][half]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    u32x2* Xw = reinterpret_cast<u32x2*>(smem_sb + (SB_WCONV + SB_WIH) * 16 + 256 * 4) + wave * (3 * SB_PSTR);   // this wave's patch
    const long long plane = (long long)a.H * a.W;
    const int total = a.ntiles * a.B;

    // ---- once per workgroup: weights and tables into LDS ----------------------------------------------------------------------------------
    {
        const u32x4* src = reinterpret_cast<const u32x4*>(a.packed);
        constexpr int WIT = (SB_WCONV + SB_WIH + SB_NT - 1) / SB_NT;
        u32x4 wreg[WIT];
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const int i = tid + it * SB_NT;
            wreg[it] = src[i < SB_WCONV + SB_WIH ? i : SB_WCONV + SB_WIH - 1];
        }
        if (tid < 64) {
            const int tc = sb_chan(tid >> 1, tid & 1);
            tabl[tid] = a.hh[tc];
            tabl[64 + tid] = a.b_conv ? a.b_conv[tc] : 0.f;
            tabl[128 + tid] = a.b_ih ? a.b_ih[tc] : 0.f;
            tabl[192 + tid] = a.hh[tid];          // channel order (wide epilogue)
        }
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const int i = tid + it * SB_NT;
            if (i < SB_WCONV + SB_WIH) Wl[i] = wreg[it];
        }
    }
    __syncthreads();      // the only workgroup barrier

    // request a unit's 5 x 36 input patch (replicate border = clamp, conv_layers.py:72-76): raw values only -- (eta, four partial planes) as
    // 5 complex values, or the 4 channels of x
    auto load_patch = [&](int b, int oy, int w0, float (&raw)[SB_PSLOT][10], unsigned (&off)[SB_PSLOT]) {
#pragma unroll
        for (int q = 0; q < SB_PSLOT; ++q) {
            int p = lane + 64 * q;
            p = p < SB_PPIX ? p : SB_PPIX - 1;
            const int ty = p / SB_PW, tx = p - ty * SB_PW;
            int gy = oy + ty - SB_PAD, gx = w0 + tx - SB_PAD;
            gy = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy);
            gx = gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx);
            off[q] = (unsigned)(gy * a.W + gx);
            if (a.eta2) {                // the first four partial planes in flight together
                const float2* e2 = a.eta2 + (long long)b * plane;
                const float2* pp = a.part + (long long)b * plane;
                const float2 e = e2[off[q]];
                raw[q][0] = e.x, raw[q][1] = e.y;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float2 v = pp[(long long)(k < a.nparts ? k : 0) * a.part_stride + off[q]];
                    raw[q][2 + 2 * k] = v.x, raw[q][3 + 2 * k] = v.y;
                }
            } else {
                const float* xb = a.x + (long long)b * a.Cin * plane;
#pragma unroll
                for (int c = 0; c < 4; ++c) raw[q][c] = xb[(c < a.Cin ? c * plane : 0) + off[q]];
            }
        }
    };
    // finish the patch, split it into its three bf16 terms and write the wave's LDS planes
    auto commit_patch = [&](int b, const float (&raw)[SB_PSLOT][10], const unsigned (&off)[SB_PSLOT]) {
#pragma unroll
        for (int q = 0; q < SB_PSLOT; ++q) {
            float c0, c1, c2, c3;
            if (a.eta2) {   // the last step of log_likelihood_gradient (rim_utils.py:61-67), same order of additions as k_rim_layer
                float sx = raw[q][2], sy = raw[q][3];
#pragma unroll
                for (int k = 1; k < 4; ++k)
                    if (k < a.nparts) sx += raw[q][2 + 2 * k], sy += raw[q][3 + 2 * k];
                for (int k = 4; k < a.nparts; ++k) {
                    const float2 v = a.part[(long long)k * a.part_stride + (long long)b * plane + off[q]];
                    sx += v.x;
                    sy += v.y;
                }
                c0 = raw[q][0], c1 = raw[q][1], c2 = sx * a.post, c3 = sy * a.post;
            } else {
                c0 = raw[q][0];
                c1 = a.Cin > 1 ? raw[q][1] : 0.f;
                c2 = a.Cin > 2 ? raw[q][2] : 0.f;
                c3 = a.Cin > 3 ? raw[q][3] : 0.f;
            }
            unsigned p1, p2, p3, q1, q2, q3;
            sb_split2(c0, c1, p1, p2, p3);
            sb_split2(c2, c3, q1, q2, q3);
            const int p = lane + 64 * q;
            if (p < SB_PPIX) {
                Xw[p] = u32x2{p1, q1};
                Xw[SB_PSTR + p] = u32x2{p2, q2};
                Xw[2 * SB_PSTR + p] = u32x2{p3, q3};
            }
        }
    };
    auto unit_of = [&](int t, int& b, int& oy, int& w0) {
        const int tt = (int)mrx_xcd_band(t, total);
        b = tt / a.ntiles;
        const int tile = tt - b * a.ntiles, ty0 = tile / a.tiles_x;
        oy = ty0 * SB_TH + wave;
        w0 = (tile - ty0 * a.tiles_x) * SB_TW;
    };

    float wmax = 0.f;                        // maximum of this lane's outputs (a.xmax)
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        int b, oy, w0;
        unit_of(t, b, oy, w0);
        if (oy >= a.H) continue;             // wave-uniform: rows past the image (H % 16 != 0)
        {
            float raw[SB_PSLOT][10];
            unsigned roff[SB_PSLOT];
            load_patch(b, oy, w0, raw, roff);
            commit_patch(b, raw, roff);
        }
        // wave-private LDS: program order is enough, no barrier

        // ---- conv 5x5: accumulators start at the bias ----------------------------------------------------------------------------------------
        f32x16 acc[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][r] = tabl[64 + 2 * (ct * 16 + r) + lhi];
        {
            const u32x2* xw = Xw + l31;
            const u32x4* wl = Wl + lane;
#pragma unroll
            for (int s = 0; s < SB_KS; ++s) {
                auto toff = [](int tp) { return tp < SB_K * SB_K ? (tp / SB_K) * SB_PW + (tp % SB_K) : 0; };   // zero-weight taps read pixel 0
                const int offA = lhi ? toff(4 * s + 2) : toff(4 * s), offB = lhi ? toff(4 * s + 3) : toff(4 * s + 1);
                bf16x8 bt[3], at[2][3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const u32x2 lo = xw[k * SB_PSTR + offA], hi = xw[k * SB_PSTR + offB];
                    bt[k] = __builtin_bit_cast(bf16x8, (u32x4{lo.x, lo.y, hi.x, hi.y}));
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) at[ct][k] = __builtin_bit_cast(bf16x8, wl[((s * 3 + k) * 2 + ct) * 64]);
                }
                SB_MFMA12(acc, at, bt[0], bt[1], bt[2])
            }
        }

        // ---- g = ReLU(conv + b) in registers; h = ReLU(W_ih g + b_ih + hh * h_prev) ------------------------------------------------------------
        // h_prev is requested in four groups of eight, each after the 1x1 step that frees eight registers of g; lanes outside the image read a
        // valid element (clamped) and store nothing; without h_prev the loads go to h_new and are ignored
        const int ox = w0 + l31, cx = ox < a.W ? ox : a.W - 1;
        const float* hb = (a.hprev ? a.hprev : a.hnew) + (long long)b * SB_F * plane + (long long)oy * a.W + cx + 4ll * lhi * plane;
        float hp[32];
        f32x16 acc2[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[ct][r] = tabl[128 + 2 * (ct * 16 + r) + lhi];
        {
            const u32x4* wl = Wl + SB_WCONV + lane;
#pragma unroll
            for (int s = 0; s < SB_KS2; ++s) {
                unsigned g1[4], g2[4], g3[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int R0 = 8 * s + 2 * q, R1 = R0 + 1;
                    float v0 = acc[R0 >> 4][R0 & 15], v1 = acc[R1 >> 4][R1 & 15];
                    v0 = v0 > 0.f ? v0 : 0.f;
                    v1 = v1 > 0.f ? v1 : 0.f;
                    sb_split2(v0, v1, g1[q], g2[q], g3[q]);
                }
#pragma unroll
                for (int R = 8 * s; R < 8 * s + 8; ++R) hp[R] = hb[(long long)sb_chan(R, 0) * plane];
                const bf16x8 b1 = __builtin_bit_cast(bf16x8, (u32x4{g1[0], g1[1], g1[2], g1[3]}));
                const bf16x8 b2 = __builtin_bit_cast(bf16x8, (u32x4{g2[0], g2[1], g2[2], g2[3]}));
                const bf16x8 b3 = __builtin_bit_cast(bf16x8, (u32x4{g3[0], g3[1], g3[2], g3[3]}));
                bf16x8 at[2][3];
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) at[ct][k] = __builtin_bit_cast(bf16x8, wl[((s * 3 + k) * 2 + ct) * 64]);
                SB_MFMA12(acc2, at, b1, b2, b3)
            }
        }

        // ---- epilogue: 128-byte rows per wave instruction ---------------------------------------------------------------------------------------
        if (ox < a.W) {
            float* ob = a.hnew + (long long)b * SB_F * plane + (long long)oy * a.W + ox + 4ll * lhi * plane;
            const bool first = a.hprev == nullptr;
#pragma unroll
            for (int R = 0; R < 32; ++R) {
                float v = acc2[R >> 4][R & 15] + tabl[2 * R + lhi] * (first ? 0.f : hp[R]);
                v = v > 0.f ? v : 0.f;
                ob[(long long)sb_chan(R, 0) * plane] = v;
                wmax = fmaxf(wmax, v);
            }
        }
    }
    if (a.xmax) {
        // bound of max |h_new| for the next layer's fp16 operand scale: one atomic per wave and launch, and only when it would raise the bound
        // (same-address atomics from every CU serialise: one per 32-pixel unit cost 60 us per launch)
        for (int off = 32; off > 0; off >>= 1) wmax = fmaxf(wmax, __shfl_xor(wmax, off, 64));
        if (lane == 0 && __float_as_uint(wmax) > __hip_atomic_load(a.xmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
            atomicMax(a.xmax, __float_as_uint(wmax));
    }
}

int mrx_l1sb_launch(const MrxL1sbArgs& a, hipStream_t st) {
    constexpr size_t lds = (size_t)(SB_WCONV + SB_WIH) * 16 + 256 * sizeof(float) + (size_t)(SB_NT / 64) * 3 * SB_PSTR * 8;
    static bool attr_done = false;   // once: keeps launches legal under hipGraph capture
    static int ncu = 0;
    if (!attr_done) {
        MRX_HIP(hipFuncSetAttribute((const void*)k_rim_layer1_sb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int dev = 0;
        hipDeviceProp_t prop;
        MRX_HIP(hipGetDevice(&dev));
        MRX_HIP(hipGetDeviceProperties(&prop, dev));
        ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        attr_done = true;
    }
    const long long total = (long long)a.ntiles * a.B;
    const int grid = (int)(total < ncu ? total : ncu);     // one persistent workgroup per CU (a multiple of 8: the XCD band map keeps its meaning)
    hipLaunchKernelGGL(k_rim_layer1_sb, dim3(grid), dim3(SB_NT), lds, st, a);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
