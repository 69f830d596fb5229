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
// cache policy of the channel-blocked state streams (nt: streaming) -- round 6 A/B builds, see rim_layer2_sb.hip: h_prev loads streaming (read once), h_new
// stores default (the second layer reads them back in the very next launch)
#ifndef MRX_L1_NT_ST
#define MRX_L1_NT_ST 0
#endif
#ifndef MRX_L1_NT_LD
#define MRX_L1_NT_LD 1
#endif
typedef float sb_f32x4 __attribute__((ext_vector_type(4)));
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
// two-term fp16 form (F16): operands [(s*2 + t)*2 + blk)*64 + lane], header = (conv exponent, 1x1 exponent)
#define SBH_WCONV (SB_KS * 2 * 2 * 64)
#define SBH_WIH (SB_KS2 * 2 * 2 * 64)
#define SBH_OFF (SB_WCONV + SB_WIH)         // where the fp16 section starts in the pack (16-byte units)
static_assert((SB_WCONV + SB_WIH + SBH_WCONV + SBH_WIH + 1) * 4 == MRX_L1SB_PACK_FLOATS, "pack size");
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
// two fp16 terms of a pair of values already scaled into the fp16 range
__device__ __forceinline__ void sb_split2h(float a, float b, unsigned& p1, unsigned& p2) {
    const f16x2 h = {(_Float16)a, (_Float16)b};
    const float ra = a - (float)h.x, rb = b - (float)h.y;     // exact
    const f16x2 l = {(_Float16)ra, (_Float16)rb};
    p1 = __builtin_bit_cast(unsigned, h);
    p2 = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ float sb_pow2(int e) {
    e = e < -120 ? -120 : (e > 120 ? 120 : e);
    return __uint_as_float((unsigned)(127 + e) << 23);
}
// exponent k with m * 2^k in [2^14, 2^15) (0 for zero / non-finite m)
__device__ __forceinline__ int sb_scale_exp(float m) {
    const int ex = (int)((__float_as_uint(m) >> 23) & 0xffu);
    return (ex == 0 || ex == 255) ? 0 : 14 - (ex - 127);
}

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

// fp16 section: the scale exponents (max |w| 2^k in [2^14, 2^15)) into the header, then the operands
__global__ void k_l1f16_wscale(const float* __restrict__ w, const float* __restrict__ w_ih, u32x4* __restrict__ out, int Cin) {
    __shared__ float red[256];
    unsigned ex[2] = {0u, 0u};
    for (int which = 0; which < 2; ++which) {
        const float* p = which == 0 ? w : w_ih;
        const int n = which == 0 ? SB_F * Cin * SB_K * SB_K : SB_F * SB_F;
        float m = 0.f;
        for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(p[i]));
        red[threadIdx.x] = m;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
            __syncthreads();
        }
        ex[which] = (unsigned)sb_scale_exp(red[0]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[SBH_OFF + SBH_WCONV + SBH_WIH] = u32x4{ex[0], ex[1], 0u, 0u};
}
__global__ void k_l1f16_pack(const float* __restrict__ w, const float* __restrict__ w_ih, u32x4* __restrict__ out, int Cin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= SBH_WCONV + SBH_WIH) return;
    const bool conv = i < SBH_WCONV;
    const u32x4 hd = out[SBH_OFF + SBH_WCONV + SBH_WIH];
    const float sw = sb_pow2((int)(conv ? hd[0] : hd[1]));
    int r = conv ? i : i - SBH_WCONV;
    const int lane = r & 63;
    r >>= 6;
    const int blk = r & 1;
    r >>= 1;
    const int t = r & 1, s = r >> 1;
    const int o = 32 * blk + (lane & 31), half = lane >> 5;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (conv) {
            const int ci = j & 3, tap = 4 * s + 2 * half + (j >> 2);
            v[j] = (tap < SB_K * SB_K && ci < Cin) ? w[((long long)o * Cin + ci) * (SB_K * SB_K) + tap] * sw : 0.f;
        } else
            v[j] = w_ih[o * SB_F + sb_chan(8 * s + j, half)] * sw;
    }
    unsigned p[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned p1, p2;
        sb_split2h(v[2 * q], v[2 * q + 1], p1, p2);
        p[q] = t == 0 ? p1 : p2;
    }
    out[SBH_OFF + i] = u32x4{p[0], p[1], p[2], p[3]};
}

int mrx_l1sb_pack(const float* w_conv, const float* w_ih, float* packed, int Cin, hipStream_t st) {
    hipLaunchKernelGGL(k_l1f16_wscale, dim3(1), dim3(256), 0, st, w_conv, w_ih, reinterpret_cast<u32x4*>(packed), Cin);
    hipLaunchKernelGGL(k_l1f16_pack, dim3((SBH_WCONV + SBH_WIH + 255) / 256), dim3(256), 0, st, w_conv, w_ih, reinterpret_cast<u32x4*>(packed), Cin);
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

// two fp16 terms per operand: the three products of weight >= 2^-11, smallest first, both cout blocks
#define SB_MFMA6H(ACC, A, B1, B2)                                                                \
    ACC[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0][1], B1, ACC[0], 0, 0, 0);                \
    ACC[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[1][1], B1, ACC[1], 0, 0, 0);                \
    ACC[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0][0], B2, ACC[0], 0, 0, 0);                \
    ACC[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[1][0], B2, ACC[1], 0, 0, 0);                \
    ACC[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0][0], B1, ACC[0], 0, 0, 0);                \
    ACC[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[1][0], B1, ACC[1], 0, 0, 0);

#define SB_PROWS SB_K                       // a wave's patch: 5 rows x 36 pixels
#define SB_PPIX (SB_PROWS * SB_PW)          // 180
#define SB_PSTR 184                         // term-plane stride in the wave's LDS patch
#define SB_PSLOT 3                          // patch pixels per lane

// One persistent workgroup per CU, 16 waves (4 per SIMD), weights staged once.  After the single barrier every wave walks its own units
// (image row x 32 pixels x 64 channels) start to finish: patch -> LDS (wave-private), conv, 1x1, epilogue.  Nothing is prefetched across
// units: with four independent waves per SIMD at different points of that sequence, one wave's memory phase runs under the others' MFMAs.
// F16: the operands as two fp16 terms (three term products per multiply instead of six) -- the patch scaled by the power of two that puts the
// UNIT's largest input into [2^14, 2^15) (every input of the unit's outputs is in the wave's patch), ReLU(conv + b) scaled per PIXEL for the 1x1
// stage (contraction over the pixel's channels only), the weights at pack time; the accumulators are scaled back exactly before the biases.
// CB8: h_prev / h_new channel-blocked, [b][c / 8][y][x][c % 8] -- registers 16 ct + 4 j .. + 3 of a lane are four consecutive channels of block
// 4 ct + j: 8 + 8 16-byte state accesses per unit instead of 32 + 32 4-byte ones, every unit a 1 KB-aligned 1 KB block per channel block
// MORE: more than four partial planes of the log-likelihood gradient may arrive (a run-time loop over the rest; the headline has four: three coil groups
// and the constant data term).  Without it the patch code is straight-line -- with it `raw` ended up in scratch memory, written behind `s_waitcnt vmcnt(0)`.
// LLGT: 1 = the input is (eta, partial planes) -- known at compile time (the run-time test of a.eta2 inside the unrolled slot loops left part of `raw`
// in scratch memory); -1 = decided at run time.
// NW (round 5, A/B): waves per workgroup = image rows per tile.  16: four waves per SIMD at <= 128 registers -- h_prev can only be requested behind the 1x1
// stage (its 32 registers are g's until then), so every unit waits a full memory latency for it.  12: three waves per SIMD at <= 168 registers,
// h_prev requested TOGETHER with the unit's patch and landed long before the epilogue needs it.
template <bool F16, bool CB8 = false, bool MORE = true, int LLGT = -1, int NW = 16>
__global__ __launch_bounds__(NW * 64, 1) void k_rim_layer1_sb(MrxL1sbArgs a) {
    constexpr int NTHR = NW * 64;
    constexpr bool HP_EARLY = NW == 12 && CB8;
    constexpr int NT = F16 ? 2 : 3, WCONV = F16 ? SBH_WCONV : SB_WCONV, WIH = F16 ? SBH_WIH : SB_WIH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_sb[];
    u32x4* Wl = reinterpret_cast<u32x4*>(smem_sb);                                        // [WCONV + WIH] A operands
    float* tabl = reinterpret_cast<float*>(smem_sb + (WCONV + WIH) * 16);                 // hh, b_conv, b_ih in register order [half][R]: a lane's 32 values are contiguous (the compiler merges them into 16-byte LDS reads)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    u32x2* Xw = reinterpret_cast<u32x2*>(smem_sb + (WCONV + WIH) * 16 + 256 * 4) + wave * (NT * SB_PSTR);   // this wave's patch
    const long long plane = (long long)a.H * a.W;
    const int total = a.ntiles * a.B;

    // ---- once per workgroup: weights and tables into LDS ----------------------------------------------------------------------------------
    {
        const u32x4* src = reinterpret_cast<const u32x4*>(a.packed) + (F16 ? SBH_OFF : 0);
        constexpr int WIT = (WCONV + WIH + NTHR - 1) / NTHR;
        u32x4 wreg[WIT];
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const int i = tid + it * NTHR;
            wreg[it] = src[i < WCONV + WIH ? i : WCONV + WIH - 1];
        }
        if (tid < 64) {
            const int tc = sb_chan(tid >> 1, tid & 1);
            const int ti = (tid & 1) * 32 + (tid >> 1);
            tabl[ti] = a.hh[tc];
            tabl[64 + ti] = a.b_conv ? a.b_conv[tc] : 0.f;
            tabl[128 + ti] = a.b_ih ? a.b_ih[tc] : 0.f;
            tabl[192 + tid] = a.hh[tid];          // channel order (wide epilogue)
        }
#pragma unroll
        for (int it = 0; it < WIT; ++it) {
            const int i = tid + it * NTHR;
            if (i < WCONV + WIH) Wl[i] = wreg[it];
        }
    }
    float unwc = 1.f, unwi = 1.f;            // F16: 2^-k of the weight scales
    if constexpr (F16) {
        const u32x4 hd = reinterpret_cast<const u32x4*>(a.packed)[SBH_OFF + SBH_WCONV + SBH_WIH];
        unwc = sb_pow2(-(int)hd[0]), unwi = sb_pow2(-(int)hd[1]);
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
            if (LLGT > 0 || (LLGT < 0 && a.eta2)) {                // the first four partial planes in flight together
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
    // finish the patch, split it into its bf16 / fp16 terms and write the wave's LDS planes; returns 2^-k of the unit's scale (F16)
    auto commit_patch = [&](int b, const float (&raw)[SB_PSLOT][10], const unsigned (&off)[SB_PSLOT]) -> float {
        float cc[SB_PSLOT][4];
        // the last step of log_likelihood_gradient (rim_utils.py:61-67), same order of additions as k_rim_layer.  The run-time loop over the planes past the
        // fourth stands OUTSIDE the unrolled slot loop: inside it, it kept the slot loop from unrolling, `raw` was indexed dynamically and lived in scratch
        // memory -- the patch loads were written there one by one behind `s_waitcnt vmcnt(0)`
        float sxq[SB_PSLOT], syq[SB_PSLOT];
        const bool llg = LLGT > 0 || (LLGT < 0 && a.eta2);
        if (llg) {
#pragma unroll
            for (int q = 0; q < SB_PSLOT; ++q) {
                float sx = raw[q][2], sy = raw[q][3];
#pragma unroll
                for (int k = 1; k < 4; ++k)
                    if (k < a.nparts) sx += raw[q][2 + 2 * k], sy += raw[q][3 + 2 * k];
                sxq[q] = sx, syq[q] = sy;
            }
            if constexpr (MORE)
            for (int k = 4; k < a.nparts; ++k) {
                const float2* pk = a.part + (long long)k * a.part_stride + (long long)b * plane;
#pragma unroll
                for (int q = 0; q < SB_PSLOT; ++q) {
                    const float2 v = pk[off[q]];
                    sxq[q] += v.x;
                    syq[q] += v.y;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < SB_PSLOT; ++q) {
            float c0, c1, c2, c3;
            if (llg) {
                c0 = raw[q][0], c1 = raw[q][1], c2 = sxq[q] * a.post, c3 = syq[q] * a.post;
            } else {
                c0 = raw[q][0];
                c1 = a.Cin > 1 ? raw[q][1] : 0.f;
                c2 = a.Cin > 2 ? raw[q][2] : 0.f;
                c3 = a.Cin > 3 ? raw[q][3] : 0.f;
            }
            cc[q][0] = c0, cc[q][1] = c1, cc[q][2] = c2, cc[q][3] = c3;
        }
        float sxu = 1.f, unx = 1.f;
        if constexpr (F16) {
            float m = 0.f;
#pragma unroll
            for (int q = 0; q < SB_PSLOT; ++q)
#pragma unroll
                for (int c = 0; c < 4; ++c) m = fmaxf(m, fabsf(cc[q][c]));
            for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
            const int kx = sb_scale_exp(m);
            sxu = sb_pow2(kx), unx = sb_pow2(-kx);
        }
#pragma unroll
        for (int q = 0; q < SB_PSLOT; ++q) {
            const int p = lane + 64 * q;
            if constexpr (F16) {
                unsigned p1, p2, q1, q2;
                sb_split2h(cc[q][0] * sxu, cc[q][1] * sxu, p1, p2);
                sb_split2h(cc[q][2] * sxu, cc[q][3] * sxu, q1, q2);
                if (p < SB_PPIX) {
                    Xw[p] = u32x2{p1, q1};
                    Xw[SB_PSTR + p] = u32x2{p2, q2};
                }
            } else {
                unsigned p1, p2, p3, q1, q2, q3;
                sb_split2(cc[q][0], cc[q][1], p1, p2, p3);
                sb_split2(cc[q][2], cc[q][3], q1, q2, q3);
                if (p < SB_PPIX) {
                    Xw[p] = u32x2{p1, q1};
                    Xw[SB_PSTR + p] = u32x2{p2, q2};
                    Xw[2 * SB_PSTR + p] = u32x2{p3, q3};
                }
            }
        }
        return unx;
    };
    auto unit_of = [&](int t, int& b, int& oy, int& w0) {
        const int tt = (int)mrx_xcd_band(t, total);
        b = tt / a.ntiles;
        const int tile = tt - b * a.ntiles, ty0 = tile / a.tiles_x;
        oy = ty0 * NW + wave;
        w0 = (tile - ty0 * a.tiles_x) * SB_TW;
    };

    float wmax = 0.f;                        // maximum of this lane's outputs (a.xmax)
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        int b, oy, w0;
        unit_of(t, b, oy, w0);
        if (oy >= a.H) continue;             // wave-uniform: rows past the image (H % 16 != 0)
        float unx = 1.f;
        float hp[32];
        {
            float raw[SB_PSLOT][10];
            unsigned roff[SB_PSLOT];
            load_patch(b, oy, w0, raw, roff);
            if constexpr (HP_EARLY) {              // the unit's h_prev, requested behind its patch: both latencies run together, under the other waves' units
                const int ox_ = w0 + l31, cx_ = ox_ < a.W ? ox_ : a.W - 1;
                const float* hb_ = (a.hprev ? a.hprev : a.hnew) + (long long)b * SB_F * plane + ((long long)oy * a.W + cx_) * 8 + 4 * lhi;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float4 u = *reinterpret_cast<const float4*>(hb_ + (long long)q * plane * 8);
                    hp[4 * q] = u.x, hp[4 * q + 1] = u.y, hp[4 * q + 2] = u.z, hp[4 * q + 3] = u.w;
                }
                __builtin_amdgcn_sched_barrier(0);     // (the scheduler would sink the requests behind the patch's s_waitcnt: one latency after the other)
            }
            unx = commit_patch(b, raw, roff);
        }
        // wave-private LDS: program order is enough, no barrier

        // ---- conv 5x5: accumulators start at the bias ----------------------------------------------------------------------------------------
        f32x16 acc[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][r] = F16 ? 0.f : tabl[64 + lhi * 32 + ct * 16 + r];
        {
            const u32x2* xw = Xw + l31;
            const u32x4* wl = Wl + lane;
#pragma unroll
            for (int s = 0; s < SB_KS; ++s) {
                auto toff = [](int tp) { return tp < SB_K * SB_K ? (tp / SB_K) * SB_PW + (tp % SB_K) : 0; };   // zero-weight taps read pixel 0
                const int offA = lhi ? toff(4 * s + 2) : toff(4 * s), offB = lhi ? toff(4 * s + 3) : toff(4 * s + 1);
                if constexpr (F16) {
                    f16x8 bt[2], at[2][2];
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const u32x2 lo = xw[k * SB_PSTR + offA], hi = xw[k * SB_PSTR + offB];
                        bt[k] = __builtin_bit_cast(f16x8, (u32x4{lo.x, lo.y, hi.x, hi.y}));
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct) at[ct][k] = __builtin_bit_cast(f16x8, wl[((s * 2 + k) * 2 + ct) * 64]);
                    }
                    SB_MFMA6H(acc, at, bt[0], bt[1])
                } else {
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
        }
        if constexpr (F16) {                 // back to the scale of conv + b (exact), then the bias
            const float un = unx * unwc;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ct][r] = acc[ct][r] * un + tabl[64 + lhi * 32 + ct * 16 + r];
        }

        // ---- g = ReLU(conv + b) in registers; h = ReLU(W_ih g + b_ih + hh * h_prev) ------------------------------------------------------------
        // h_prev is requested in four groups of eight, each after the 1x1 step that frees eight registers of g; lanes outside the image read a
        // valid element (clamped) and store nothing; without h_prev the loads go to h_new and are ignored
        const int ox = w0 + l31, cx = ox < a.W ? ox : a.W - 1;
        const float* hb = (a.hprev ? a.hprev : a.hnew) + (long long)b * SB_F * plane +
                          (CB8 ? ((long long)oy * a.W + cx) * 8 + 4 * lhi : (long long)oy * a.W + cx + 4ll * lhi * plane);
        auto load_hp8 = [&](int s) {          // registers 8 s .. 8 s + 7
            if constexpr (HP_EARLY) {
                return;
            } else if constexpr (CB8) {
                // (all eight 16-byte loads go out together behind the 1x1 stage, when g's registers are free: requested group by group
                // beside it they need aligned register quads the allocator does not have -- 44 spilled registers, 15 us)
                if (s == SB_KS2 - 1) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const sb_f32x4 u = MRX_L1_NT_LD ? __builtin_nontemporal_load(reinterpret_cast<const sb_f32x4*>(hb + (long long)q * plane * 8))
                                                        : *reinterpret_cast<const sb_f32x4*>(hb + (long long)q * plane * 8);
                        hp[4 * q] = u.x, hp[4 * q + 1] = u.y, hp[4 * q + 2] = u.z, hp[4 * q + 3] = u.w;
                    }
                }
            } else {
#pragma unroll
                for (int R = 8 * s; R < 8 * s + 8; ++R) hp[R] = hb[(long long)sb_chan(R, 0) * plane];
            }
        };
        f32x16 acc2[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[ct][r] = F16 ? 0.f : tabl[128 + lhi * 32 + ct * 16 + r];
        if constexpr (F16) {
            // per-pixel scale: the pixel's 64 channels sit in this lane and in lane ^ 32; a lane's accumulators belong to its own pixel
            float gm = 0.f;
#pragma unroll
            for (int R = 0; R < 32; ++R) gm = fmaxf(gm, acc[R >> 4][R & 15]);
            gm = fmaxf(gm, __shfl_xor(gm, 32, 64));
            const int kg = sb_scale_exp(gm);
            const float sg = sb_pow2(kg), ung = sb_pow2(-kg) * unwi;
            const u32x4* wl = Wl + WCONV + lane;
#pragma unroll
            for (int s = 0; s < SB_KS2; ++s) {
                unsigned g1[4], g2[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int R0 = 8 * s + 2 * q, R1 = R0 + 1;
                    float v0 = acc[R0 >> 4][R0 & 15], v1 = acc[R1 >> 4][R1 & 15];
                    v0 = v0 > 0.f ? v0 : 0.f;
                    v1 = v1 > 0.f ? v1 : 0.f;
                    sb_split2h(v0 * sg, v1 * sg, g1[q], g2[q]);
                }
                load_hp8(s);
                const f16x8 b1 = __builtin_bit_cast(f16x8, (u32x4{g1[0], g1[1], g1[2], g1[3]}));
                const f16x8 b2 = __builtin_bit_cast(f16x8, (u32x4{g2[0], g2[1], g2[2], g2[3]}));
                f16x8 at[2][2];
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) at[ct][k] = __builtin_bit_cast(f16x8, wl[((s * 2 + k) * 2 + ct) * 64]);
                SB_MFMA6H(acc2, at, b1, b2)
            }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[ct][r] = acc2[ct][r] * ung + tabl[128 + lhi * 32 + ct * 16 + r];
        } else {
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
                load_hp8(s);
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
            const bool first = a.hprev == nullptr;
            if constexpr (CB8) {
                float* ob = a.hnew + (long long)b * SB_F * plane + ((long long)oy * a.W + ox) * 8 + 4 * lhi;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int R = 4 * q + i;
                        v[i] = acc2[R >> 4][R & 15] + tabl[lhi * 32 + R] * (first ? 0.f : hp[R]);
                        v[i] = v[i] > 0.f ? v[i] : 0.f;
                        wmax = fmaxf(wmax, v[i]);
                    }
                    if (MRX_L1_NT_ST) __builtin_nontemporal_store((sb_f32x4{v[0], v[1], v[2], v[3]}), reinterpret_cast<sb_f32x4*>(ob + (long long)q * plane * 8));
                    else *reinterpret_cast<float4*>(ob + (long long)q * plane * 8) = make_float4(v[0], v[1], v[2], v[3]);
                }
            } else {
                float* ob = a.hnew + (long long)b * SB_F * plane + (long long)oy * a.W + ox + 4ll * lhi * plane;
#pragma unroll
                for (int R = 0; R < 32; ++R) {
                    float v = acc2[R >> 4][R & 15] + tabl[lhi * 32 + R] * (first ? 0.f : hp[R]);
                    v = v > 0.f ? v : 0.f;
                    ob[(long long)sb_chan(R, 0) * plane] = v;
                    wmax = fmaxf(wmax, v);
                }
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

// One launch of the (F16, CB8) kernel in the form the arguments allow: the input form known at compile time keeps the patch in registers
// (LLGT 1 with at most four partial planes: the headline loop; LLGT 0: a four-channel x); anything else takes the run-time form.
template <bool F16, bool CB8>
static void l1sb_launch_form(const MrxL1sbArgs& a, int grid, size_t lds, hipStream_t st) {
    static bool attr_done = false;   // once: keeps launches legal under hipGraph capture
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)k_rim_layer1_sb<F16, CB8, false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)k_rim_layer1_sb<F16, CB8, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)k_rim_layer1_sb<F16, CB8, true, -1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    if (a.eta2 && a.nparts <= 4)
        hipLaunchKernelGGL((k_rim_layer1_sb<F16, CB8, false, 1>), dim3(grid), dim3(SB_NT), lds, st, a);
    else if (!a.eta2)
        hipLaunchKernelGGL((k_rim_layer1_sb<F16, CB8, false, 0>), dim3(grid), dim3(SB_NT), lds, st, a);
    else
        hipLaunchKernelGGL((k_rim_layer1_sb<F16, CB8, true, -1>), dim3(grid), dim3(SB_NT), lds, st, a);
}

int mrx_l1sb_launch(const MrxL1sbArgs& a, hipStream_t st) {
    constexpr size_t lds = (size_t)(SB_WCONV + SB_WIH) * 16 + 256 * sizeof(float) + (size_t)(SB_NT / 64) * 3 * SB_PSTR * 8;   // (the fp16 form needs less)
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t prop;
        MRX_HIP(hipGetDevice(&dev));
        MRX_HIP(hipGetDeviceProperties(&prop, dev));
        ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const long long total = (long long)a.ntiles * a.B;
    const int grid = (int)(total < ncu ? total : ncu);     // one persistent workgroup per CU (a multiple of 8: the XCD band map keeps its meaning)
    if (a.cb8) {
        if (!a.f16) {
            mrx_set_error("mrx_rim_layer1_cb8: the channel-blocked state layout exists for the two-term fp16 kernel only (MRIDC_AMD_ARITH=f16x2)");
            return MRX_EUNSUP;
        }
        l1sb_launch_form<true, true>(a, grid, lds, st);
    } else if (a.f16)
        l1sb_launch_form<true, false>(a, grid, lds, st);
    else
        l1sb_launch_form<false, false>(a, grid, lds, st);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
