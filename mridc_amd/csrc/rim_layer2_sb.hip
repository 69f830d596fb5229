// rim_layer2_sb.hip -- the second RIM layer (ConvNonlinear 3x3 dilation 2, 64 -> 64, replicate padding, ReLU + IndRNNCell 1x1 64 -> 64;
// reference models/rim/conv_layers.py:121-123 + rnn_cells.py:384-391) with fp32 results on the bf16 matrix pipe: the three-term operand
// split of rim_layer1_sb.hip (x = x1 + x2 + x3 in bf16, six term products per multiply, error O(2^-24)) applied to the DIRECT form of the
// convolution.  19.5 GFLOP x 6 products on v_mfma_f32_32x32x16_bf16 is 44 % of the fp32-MFMA time of the direct form and 87 % of the
// Winograd form's -- and, unlike fp32 MFMA, it co-issues with the vector ALU, which here only splits operands.
//
//   * one persistent workgroup per CU, 8 waves = a 16 x 32 pixel tile, wave = two image rows x 64 couts (4 accumulators: every A operand
//     read from LDS feeds two MFMAs, every B operand two -- 0.5 LDS reads per MFMA); operands of step s + 1 are read before the MFMAs of step s;
//   * the contraction runs over 8 chunks of 8 input channels; per chunk five MFMA steps of 16 = 2 taps x 8 channels (lower half-wave tap
//     2s, upper half-wave tap 2s + 1; the ninth taps of two consecutive chunks share one step); the chunk's halo'd tile sits in LDS as three bf16 term
//     planes [term][pixel][8 channels] (16 B per pixel: one conflict-free ds_read_b128 per B operand), its split weights (30 KB) beside it;
//   * both are double-buffered: the fp32 values of chunk q + 1 and its weights are requested before the MFMAs of chunk q, split / written
//     after them; one barrier per chunk;
//   * ReLU(conv + b) stays in registers and feeds the 1x1 GEMM eight channels at a time (weights resident in LDS), epilogue with
//     lane = pixel as in rim_layer1_sb.hip.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mrx_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define S2_NT 512
// cache policy of the hidden-state streams of the FAST / CB8 form (aux = 2: nt): every state byte is written once and read once, a whole time-step (~2.9 GB at
// 8 slices per launch) later, through a 256 MB memory-side cache.  Round 6, A/B builds on one box (profiles/r06_nt_policy_and_hp_position.txt): this kernel 66.3 ->
// 65.2 us per slice in isolation with both on; the headline 148.4 (default policy, two runs) / 148.7 (all four state streams of both layers) / 149.4 (loads only) /
// 149.9 (everything but layer 1's stores, whose lines this kernel re-reads right away) -- at most 1 %: the 26 us of this launch that do not scale with the clock
// are NOT a cache-policy effect.  Kept on in the best-measured combination.
#ifndef MRX_L2_NT_ST
#define MRX_L2_NT_ST 1
#endif
#ifndef MRX_L2_NT_LD
#define MRX_L2_NT_LD 1
#endif
#define S2_BAR_STEP 4      // even chunks: the step before which the paired ninth-tap operands become visible (barrier); 3 = one step earlier, operands prefetched (measured equal)
#define S2_COMMIT_EVEN 1   // even chunks: the step after which the next chunk is split and written (0 measured equal)
#define S2_TH 16
#define S2_TW 32
#define S2_F 64
#define S2_NPIX_MAX ((S2_TH + 4) * (S2_TW + 4))   // halo'd tile of the dilation-2 layer: 20 x 36 = 720 pixels (dilation 1: 18 x 34)
#define S2_NCH 8                            // channel chunks
#define S2_KS 5                             // MFMA steps per chunk (10 tap slots, 9 used)
#define S2_WFULL (4 * 3 * 2 * 64)           // 16-byte A operands of the four full steps of a chunk
#define S2_WCH (S2_WFULL + 3 * 2 * 32)      // + the fifth step, whose upper half-wave (the padding tap slot) is not stored: 27 KB per chunk
#define S2_WIH (4 * 3 * 2 * 64)             // 1x1 stage (24 KB)
#define S2_WP (4 * 3 * 64)                  // final 64 -> 2 convolution as per-pixel tap products: 18 of 32 rows used (12 KB)
#define S2_PACK_U4 (S2_NCH * S2_WCH + S2_WIH + S2_WP)
// F16 variant (mrx_rim_layer2_f16_*): the convolution's operands as TWO fp16 terms (11 + 11 significand bits, operands pre-scaled by powers of two
// into the fp16 range) and three term products per multiply; the 1x1 and tap stages keep the three-term bf16 form
#define S2F_WFULL (4 * 2 * 2 * 64)
#define S2F_WCH (S2F_WFULL + 2 * 2 * 32)    // 18 KB per chunk
#define S2F_PACK_U4 (S2_NCH * S2F_WCH + S2_WIH + S2_WP + 1)   // + one header element: the weight scale exponent
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

struct L2sbArgs {
    const float* x;        // [B,64,H,W]
    const u32x4* packed;   // mrx_rim_layer2_sb_pack
    const float* b_conv;   // [64] or null
    const float* b_ih;     // [64] or null
    const float* hh;       // [64]
    const float* hprev;    // [B,64,H,W] or null
    float* hnew;           // [B,64,H,W]
    float* P;              // not null: also P[b][tap * 2 + co][y][x] = sum_c w_final[co][c][tap] * h_new[c][y][x]  (mrx_rim_final_gather adds the taps up)
    float* Q = nullptr;    // FAST form, with E: the tap products pre-summed along x inside the tile -- Q[b][dy][y][x][co] = P[dy,0](x - 1) + P[dy,1](x) + P[dy,2](x + 1)
    float* E = nullptr;    // with the terms a NEIGHBOURING 32-pixel tile owes left out, which that tile leaves here: E[b][y][tile column][16]
                           // (its first column's dx = 2 products, for the pixel to its left; its last column's dx = 0 products, for the pixel to its right)
    int B, H, W, tiles_x, ntiles;
    int act;               // TAIL = false: MRX_ACT_* applied to conv + bias
    float slope;
    unsigned long long* trace;   // debug (env MRX_L2SB_TRACE): cycle stamps [workgroup][wave][tile 0..1][4]
    const unsigned* xmax;  // F16: bits of an upper bound of max |x| (>= 0), kept by the producer of x (mrx_rim_layer_indrnn_packed*_xmax)
    unsigned* xmax_out;    // TAIL = false, or null: bits of max |y| are folded in (atomic max, never reset here): the bound the NEXT convolution of a
                           // chain scales its fp16 operands by (mrx_conv3x3_sb_chain)
};

// two fp16 terms of a pair of values already scaled into the fp16 range: a = h1 + h2 + O(2^-22 |a|)
__device__ __forceinline__ void s2_split2h(float a, float b, unsigned& p1, unsigned& p2) {
    const f16x2 h = {(_Float16)a, (_Float16)b};
    const float ra = a - (float)h.x, rb = b - (float)h.y;     // exact
    const f16x2 l = {(_Float16)ra, (_Float16)rb};
    p1 = __builtin_bit_cast(unsigned, h);
    p2 = __builtin_bit_cast(unsigned, l);
}
// the same two terms of (a s, b s) for a power-of-two scale s, in FOUR instructions: the fp32 multiply, the conversion and the subtraction of each
// term are one v_fma_mix{lo,hi}_f16 (a s and a s - h are exact in fp32, so the single rounding of the fused form is the rounding of s2_split2h:
// bit-identical).  hipcc forms these from s2_split2h(a * s, b * s) when s is wave-uniform; with a per-lane scale it emits v_mul, v_cvt_pk,
// v_fma_mix_f32, v_cvt_pk -- six per pair (round 5: the row tails are issue-bound, ~7 cycles per instruction that is not under an MFMA).
__device__ __forceinline__ void s2_split2h_scaled(float a, float b, float s, unsigned& p1, unsigned& p2) {
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=&v"(p1) : "v"(a), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(p1) : "v"(b), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=&v"(p2) : "v"(a), "v"(s), "v"(p1));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(p2) : "v"(b), "v"(s), "v"(p1));
}
// 2^e as a float, e clamped to the normal range
__device__ __forceinline__ float s2_pow2(int e) {
    e = e < -120 ? -120 : (e > 120 ? 120 : e);
    return __uint_as_float((unsigned)(127 + e) << 23);
}
// exponent k with bound * 2^k in [2^14, 2^15) (0 for a zero / non-finite bound)
__device__ __forceinline__ int s2_scale_exp(unsigned bits) {
    const int ex = (int)((bits >> 23) & 0xffu);
    return (ex == 0 || ex == 255) ? 0 : 14 - (ex - 127);
}

__device__ __forceinline__ unsigned s2_pk(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ void s2_split2(float a, float b, unsigned& p1, unsigned& p2, unsigned& p3) {
    p1 = s2_pk(a, b);
    float ra = a - __uint_as_float(p1 << 16), rb = b - __uint_as_float(p1 & 0xffff0000u);
    p2 = s2_pk(ra, rb);
    ra -= __uint_as_float(p2 << 16);
    rb -= __uint_as_float(p2 & 0xffff0000u);
    p3 = s2_pk(ra, rb);
}
__host__ __device__ constexpr int s2_chan(int R, int half) { return 32 * (R >> 4) + (R & 3) + 8 * ((R & 15) >> 2) + 4 * half; }

// conv : out[q * S2_WCH + ((s*3 + t)*2 + blk)*64 + lane][j] = term_t( w[32 blk + lane%32][8 q + j][tap = 2 s + lane/32] ) for s < 4;
//        out[q * S2_WCH + S2_WFULL + (t*2 + blk)*32 + l][j]   = term_t( w[32 blk + l][8 q + j][tap 8] )   (s = 4, lower half-wave only)
// ih   : out[S2_NCH * S2_WCH + ((s*3 + t)*2 + blk)*64 + lane][j] = term_t( w_ih[32 blk + lane%32][s2_chan(8 s + j, lane/32)] )
// final: out[S2_NCH * S2_WCH + S2_WIH + (s*3 + t)*64 + lane][j] = term_t( w_final[m & 1][s2_chan(8 s + j, lane/32)][tap = m >> 1] ), m = lane%32 < 18
__global__ void k_l2sb_pack(const float* __restrict__ w, const float* __restrict__ w_ih, const float* __restrict__ w_final, u32x4* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S2_PACK_U4) return;
    float v[8];
    int t;
    if (i < S2_NCH * S2_WCH) {
        const int q = i / S2_WCH;
        int r = i - q * S2_WCH, s, blk, l, half;
        if (r < S2_WFULL) {
            const int lane = r & 63;
            r >>= 6;
            blk = r & 1;
            r >>= 1;
            t = r % 3, s = r / 3, l = lane & 31, half = lane >> 5;
        } else {
            r -= S2_WFULL;
            l = r & 31, blk = (r >> 5) & 1, t = r >> 6, s = 4, half = 0;
        }
        const int o = 32 * blk + l, tap = 2 * s + half;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = w[((long long)o * S2_F + 8 * q + j) * 9 + tap];
    } else if (i < S2_NCH * S2_WCH + S2_WIH) {
        int r = i - S2_NCH * S2_WCH;
        const int lane = r & 63;
        r >>= 6;
        const int blk = r & 1;
        r >>= 1;
        t = r % 3;
        const int s = r / 3, o = 32 * blk + (lane & 31);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = w_ih ? w_ih[o * S2_F + s2_chan(8 * s + j, lane >> 5)] : 0.f;
    } else {
        int r = i - S2_NCH * S2_WCH - S2_WIH;
        const int lane = r & 63;
        r >>= 6;
        t = r % 3;
        const int s = r / 3, m = lane & 31;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (w_final && m < 18) ? w_final[((long long)(m & 1) * S2_F + s2_chan(8 * s + j, lane >> 5)) * 9 + (m >> 1)] : 0.f;
    }
    unsigned p[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        unsigned p1, p2, p3;
        s2_split2(v[2 * k], v[2 * k + 1], p1, p2, p3);
        p[k] = t == 0 ? p1 : (t == 1 ? p2 : p3);
    }
    out[i] = u32x4{p[0], p[1], p[2], p[3]};
}

// F16 pack.  Step 1: the weight scale exponent kw (max |w| * 2^kw in [2^14, 2^15)) into the header element.
// (header element: [0] the convolution's exponent, [1] the 1x1 weights', [2] the final convolution's)
__global__ void k_l2f16_wscale(const float* __restrict__ w, const float* __restrict__ w_ih, const float* __restrict__ w_final, u32x4* __restrict__ out) {
    __shared__ float red[256];
    unsigned ex[3] = {0u, 0u, 0u};
    for (int which = 0; which < 3; ++which) {
        const float* p = which == 0 ? w : (which == 1 ? w_ih : w_final);
        const int n = which == 0 ? S2_F * S2_F * 9 : (which == 1 ? S2_F * S2_F : 2 * S2_F * 9);
        float m = 0.f;
        if (p)
            for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(p[i]));
        red[threadIdx.x] = m;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
            __syncthreads();
        }
        ex[which] = (unsigned)s2_scale_exp(__float_as_uint(red[0]));
        __syncthreads();
    }
    if (threadIdx.x == 0) out[S2F_PACK_U4 - 1] = u32x4{ex[0], ex[1], ex[2], 0u};
}
// Step 2: conv operands as two fp16 terms of w * 2^kw, in the layout of k_l2sb_pack with two terms:
//   out[q * S2F_WCH + ((s*2 + t)*2 + blk)*64 + lane][j] (s < 4), out[q * S2F_WCH + S2F_WFULL + (t*2 + blk)*32 + l][j] (tap 8);
// the 1x1 / final-conv operands follow at the offsets of k_l2sb_pack, as two fp16 terms too (scaled by their own exponents):
//   ih   : [((s*2 + t)*2 + blk)*64 + lane] (1024 of the S2_WIH elements), final: [(s*2 + t)*64 + lane] (512 of the S2_WP elements); the rest is zero.
__global__ void k_l2f16_pack(const float* __restrict__ w, const float* __restrict__ w_ih, const float* __restrict__ w_final, u32x4* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S2F_PACK_U4 - 1) return;
    float v[8];
    int t;
    if (i < S2_NCH * S2F_WCH) {
        const float sw = s2_pow2((int)out[S2F_PACK_U4 - 1][0]);
        const int q = i / S2F_WCH;
        int r = i - q * S2F_WCH, s, blk, l, half;
        if (r < S2F_WFULL) {
            const int lane = r & 63;
            r >>= 6;
            blk = r & 1;
            r >>= 1;
            t = r & 1, s = r >> 1, l = lane & 31, half = lane >> 5;
        } else {
            r -= S2F_WFULL;
            l = r & 31, blk = (r >> 5) & 1, t = r >> 6, s = 4, half = 0;
        }
        const int o = 32 * blk + l, tap = 2 * s + half;
        unsigned p[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned p1, p2;
            s2_split2h(w[((long long)o * S2_F + 8 * q + 2 * k) * 9 + tap] * sw, w[((long long)o * S2_F + 8 * q + 2 * k + 1) * 9 + tap] * sw, p1, p2);
            p[k] = t == 0 ? p1 : p2;
        }
        out[i] = u32x4{p[0], p[1], p[2], p[3]};
        return;
    } else if (i < S2_NCH * S2F_WCH + S2_WIH) {
        int r = i - S2_NCH * S2F_WCH;
        if (r >= 4 * 2 * 2 * 64) {
            out[i] = u32x4{0u, 0u, 0u, 0u};
            return;
        }
        const float sw = s2_pow2((int)out[S2F_PACK_U4 - 1][1]);
        const int lane = r & 63;
        r >>= 6;
        const int blk = r & 1;
        r >>= 1;
        t = r & 1;
        const int s = r >> 1, o = 32 * blk + (lane & 31);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = w_ih ? w_ih[o * S2_F + s2_chan(8 * s + j, lane >> 5)] * sw : 0.f;
    } else {
        int r = i - S2_NCH * S2F_WCH - S2_WIH;
        if (r >= 4 * 2 * 64) {
            out[i] = u32x4{0u, 0u, 0u, 0u};
            return;
        }
        const float sw = s2_pow2((int)out[S2F_PACK_U4 - 1][2]);
        const int lane = r & 63;
        r >>= 6;
        t = r & 1;
        const int s = r >> 1, m = lane & 31;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            v[j] = (w_final && m < 18) ? w_final[((long long)(m & 1) * S2_F + s2_chan(8 * s + j, lane >> 5)) * 9 + (m >> 1)] * sw : 0.f;
    }
    unsigned p[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        unsigned p1, p2;
        s2_split2h(v[2 * k], v[2 * k + 1], p1, p2);
        p[k] = t == 0 ? p1 : p2;
    }
    out[i] = u32x4{p[0], p[1], p[2], p[3]};
}

// one pixel's scale for a contraction over its channels only (1x1 / tap stages): the lane's 32 values and the 32 of lane ^ 32 are the 64
// channels of the pixel; returns k with max * 2^k in [2^14, 2^15)
__device__ __forceinline__ int s2_pixel_exp(float m) {
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    return s2_scale_exp(__float_as_uint(m));
}
#define S2_MFMA6H(ACC, A, B1, B2)                                                                \
    ACC[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0][1], B1, ACC[0], 0, 0, 0);                \
    ACC[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[1][1], B1, ACC[1], 0, 0, 0);                \
    ACC[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0][0], B2, ACC[0], 0, 0, 0);                \
    ACC[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[1][0], B2, ACC[1], 0, 0, 0);                \
    ACC[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0][0], B1, ACC[0], 0, 0, 0);                \
    ACC[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[1][0], B1, ACC[1], 0, 0, 0);

#define S2_MFMA12(ACC, A, B1, B2, B3)                                                             \
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

// LDS (bytes): ih weights 24576 | final-conv weights 12288 | tables + a zero operand 1024 | 2 x conv weights 27648 | 2 x planes 34560 -> 158.5 KB
// DIL: dilation (= halo) of the 3x3 kernel; TAIL: the IndRNN 1x1 stage (and, with a.P, the final convolution's tap products) -- without it the
// kernel is a plain 3x3 convolution 64 -> 64 + bias + activation (mrx_conv3x3_sb); ZP: zero instead of replicate padding
#define S2_OFF_WP (S2_WIH * 16)
#define S2_OFF_TAB (S2_OFF_WP + S2_WP * 16)
#define S2_OFF_ZERO (S2_OFF_TAB + 1008)
#define S2_OFF_W (S2_OFF_TAB + 1024)
#define S2_OFF_X (S2_OFF_W + 2 * S2_WCH * 16)
#define S2_LDS (S2_OFF_X + 2 * 3 * S2_NPIX_MAX * 16)

// ABL (probe builds only, -DMRX_PROBE + env MRX_L2_ABL): phases switched off to price them -- 1 no x loads, 2 no operand split, 4 no LDS
// staging writes, 8 no convolution MFMAs, 16 no tail, 32 no LDS operand reads, 64 no barriers, 128 no h_prev loads, 256 no stores, 512 no tap
// stage.  Results are garbage; only the time is read.
// CB8: x, h_prev and h_new are channel-blocked, [b][c / 8][y][x][c % 8] (mrx_cb8_convert): a pixel's eight channels of a chunk are 32 contiguous
// bytes for the loader (2 x 16-byte loads instead of 8 x 4 from eight planes), registers 4 q .. 4 q + 3 of a lane are four consecutive channels of
// block q (8 + 8 16-byte state accesses per row instead of 32 + 32 4-byte ones).  The arithmetic is unchanged: results are bit-identical.
// W4 (round 4): FOUR waves per workgroup on an 8 x 32 tile, the 1x1 / final-convolution operands read from L2 instead of LDS (they feed 36 MFMAs per
// 32 pixels, once per row tail) -- 65 KB of LDS and <= 256 registers, so TWO workgroups share a CU with independent barriers: while one is in its row
// tails or waits for a chunk, the other one's convolution owns the matrix pipe of the same SIMDs (the eight waves of the 16 x 32 form sit in the same
// phase between barriers: 44.9 k cycles of chunk loop for 27.6 k of MFMA issue, tails with the pipe idle).  Same arithmetic, same pack: bit-identical.
// FAST (round 5, the product route of mrx_rim_layer2_f16_cb8 whenever a sample's state fits 32-bit byte offsets): the NEWST staging + both rows of a wave
// through the 1x1 / tap stages together (below).  Measured against the round-4 form on the same boxes (tools/probe/l2_time.py, 8 slices per launch):
// 65.1 -> 62.5 us per slice, matrix pipe 0.565 -> 0.62 busy, 28 % fewer vector instructions (profiles/r05_layer2_wave_states_pmc.txt); bit-identical to
// the form without it.
// W16 (round 5, A/B): SIXTEEN waves per workgroup, one image row each (four waves per SIMD instead of two: an in-order wave parked at a waitcnt or a barrier
// leaves three candidates for the matrix pipe instead of one) -- 128 registers per wave, one accumulator pair, every A fragment serves one row.
template <int DIL, bool TAIL, bool ZP, bool F16 = false, int ABL = 0, bool CB8 = false, bool W4 = false, bool W16 = false, bool FAST = false>
__global__ __launch_bounds__(W16 ? 1024 : (W4 ? 256 : S2_NT), W4 ? 2 : 1) void k_rim_layer2_sb(L2sbArgs a) {
    constexpr int NTHR = W16 ? 1024 : (W4 ? 256 : S2_NT), TH = W4 ? 8 : S2_TH, RPW = W16 ? 1 : 2;
    constexpr int S2_PAD = DIL, S2_PH = TH + 2 * DIL, S2_PW = S2_TW + 2 * DIL, S2_NPIX = S2_PH * S2_PW;
    constexpr int NT = F16 ? 2 : 3;                                        // operand terms of the convolution stage
    constexpr int WFULL = F16 ? S2F_WFULL : S2_WFULL, WCH = F16 ? S2F_WCH : S2_WCH;
    constexpr int PK_TAIL = S2_NCH * WCH;                                  // where the 1x1 / final-conv operands start in the pack
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_s2[];
    constexpr int OFF_TAB = W4 ? 0 : S2_OFF_TAB, OFF_ZERO = OFF_TAB + 1008, OFF_W = OFF_TAB + 1024;
    constexpr int OFF_X = W4 ? OFF_W + 2 * WCH * 16 : S2_OFF_X;
    const u32x4* Wih = W4 ? a.packed + PK_TAIL : reinterpret_cast<const u32x4*>(smem_s2);   // (1x1 operands, then the final convolution's)
    float* tabl = reinterpret_cast<float*>(smem_s2 + OFF_TAB);      // hh, b_conv, b_ih in register order [half][R] (a lane's 32 values contiguous: 16-byte LDS reads)
    u32x4* Wc = reinterpret_cast<u32x4*>(smem_s2 + OFF_W);             // [2][S2_WCH]
    u32x4* Xp = reinterpret_cast<u32x4*>(smem_s2 + OFF_X);             // [2][NT terms][S2_NPIX]
    // The wave index lives in an SGPR; the lane index is re-derived (mbcnt, behind an opaque asm so that it is not hoisted) wherever it is needed: kept live
    // across the tile loop, the thread index and the per-lane offsets built from it were what the register allocator spilled (13 dwords at the 256-register
    // cap), and every reload in the tail drained the stores / prefetches in flight -- vmcnt retires in order and a scratch access is a VMEM access.
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    auto lane_now = [&]() {
        int l = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        asm volatile("" : "+v"(l));
        return l;
    };
    const int tid = wave * 64 + lane_now(), lane = tid & 63;
    const long long plane = (long long)a.H * a.W;
    const int total = a.ntiles * a.B;

    // once per workgroup: 1x1 weights and tables
    if (TAIL && !W4)
        for (int i = tid; i < S2_WIH + S2_WP; i += NTHR) reinterpret_cast<u32x4*>(smem_s2)[i] = a.packed[PK_TAIL + i];  // (the final-conv operands follow the 1x1 ones)
    // F16: x is multiplied by 2^kx (the producer's bound of max |x| lands in [2^14, 2^15)), the weights were by 2^kw; the accumulators
    // are scaled back (exactly) before the bias
    float sx = 1.f, unx = 1.f, unw = 1.f;
    if constexpr (F16) {
        const int kx = s2_scale_exp(a.xmax[0]), kw = (int)a.packed[S2F_PACK_U4 - 1][0];
        sx = s2_pow2(kx), unx = s2_pow2(-kx), unw = s2_pow2(-kw);
    }
    const float unwi = F16 ? s2_pow2(-(int)a.packed[S2F_PACK_U4 - 1][1]) : 1.f, unwp = F16 ? s2_pow2(-(int)a.packed[S2F_PACK_U4 - 1][2]) : 1.f;
    if (tid == 0) *reinterpret_cast<u32x4*>(smem_s2 + OFF_ZERO) = u32x4{0u, 0u, 0u, 0u};
    if (tid < 64) {
        const int tc = s2_chan(tid >> 1, tid & 1);
        const int ti = (tid & 1) * 32 + (tid >> 1);
        tabl[ti] = a.hh ? a.hh[tc] : 0.f;
        // F16 + TAIL: the convolution's accumulators START from the bias in their own (scaled) domain, b 2^kx 2^kw -- exact -- and are never scaled
        // back: the 1x1 stage's per-pixel exponent is taken from them as they are (round 5: 128 multiply-adds and 8 LDS reads per wave and tile less)
        tabl[64 + ti] = a.b_conv ? a.b_conv[tc] * ((F16 && TAIL) ? sx * s2_pow2((int)a.packed[S2F_PACK_U4 - 1][0]) : 1.f) : 0.f;
        tabl[128 + ti] = a.b_ih ? a.b_ih[tc] : 0.f;
    }

    // NEWST (round 5): the staging without a branch and without per-chunk address arithmetic.  The fp32 tile and the weight operands arrive through
    // buffer descriptors (per-thread byte offsets made once per tile, the chunk in the scalar offset), threads whose second pixel / third weight
    // operand does not exist write a dummy LDS slot instead of skipping the write, and the split + LDS writes of chunk q + 1 and the requests of
    // chunk q + 2 are cut into slices that sit BETWEEN the MFMAs of steps 2 and 3 of chunk q (pinned with sched_barrier): the chunk loop is one
    // basic block per step, and the ~120 vector / memory instructions per chunk issue in the shadow of 24 MFMAs instead of in a clump that both
    // waves of a SIMD reach at the same time.
    constexpr bool NEWST = FAST && F16 && CB8 && TAIL && !W4 && ABL == 0 && !ZP;
    // LDS strides in 16-byte slots: a term plane and a weight buffer carry ONE extra slot at their end in the NEWST layout -- the dummy that threads
    // without a second pixel / third weight operand write (both term writes of such a thread land in their plane's dummy: same immediate offsets)
    constexpr int PSTR = NEWST ? S2_NPIX + 1 : S2_NPIX, XBUF = NT * PSTR, WBUF = NEWST ? WCH + 1 : WCH;
    // staging roles: thread i owns pixels i and i + 512 of the halo'd tile (8 channels of the chunk) and copies <= 4 weight operands
    constexpr int XV = (S2_NPIX + NTHR - 1) / NTHR;                 // 2
    constexpr int WV = (WCH + NTHR - 1) / NTHR;                     // 4 (F16: 3)

    // The staging pipeline runs two chunks ahead of the MFMAs and across tile boundaries: while chunk q of a tile is multiplied, chunk q + 1
    // is split and written (mid-chunk: the vector ALU work rides under the other wave's MFMAs) and chunk q + 2 is requested; the first
    // chunks of the next tile are staged during the last chunks of this one, so the 1x1 stage and the epilogue hide that latency.
    int st_t = blockIdx.x, st_q = 0;                 // the next chunk to request: tile index, chunk
    const float* st_xb = a.x;
    long long goff[XV];
    unsigned goff32[XV];                             // NEWST: byte offset of this thread's pixels inside the sample's channel-blocked tensor (chunk 0)
    __amdgpu_buffer_rsrc_t st_rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (unsigned)(plane * (S2_F * 4)), 0x00020000);
    const __amdgpu_buffer_rsrc_t st_rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(a.packed), 0, (unsigned)(PK_TAIL * 16), 0x00020000);
    unsigned st_tid16 = 0, st_px1 = 0, st_w2 = 0;    // NEWST: LDS byte offsets inside a buffer: first pixel / first weight operand; the LAST pixel and the LAST weight operand of the thread (or the dummy slots)
    if constexpr (NEWST) {
        const unsigned t_ = (unsigned)(wave * 64 + lane_now());
        st_tid16 = t_ * 16u;
        st_px1 = (t_ + (XV - 1) * NTHR < (unsigned)S2_NPIX ? t_ + (XV - 1) * NTHR : (unsigned)S2_NPIX) * 16u;
        st_w2 = (t_ + (WV - 1) * NTHR < (unsigned)WCH ? t_ + (WV - 1) * NTHR : (unsigned)WCH) * 16u;
    }
    unsigned zmask = 0, zpend = 0;                   // ZP: this thread's pixels outside the image (of the next request / of the pending chunk)
    auto st_coords = [&]() {
        if constexpr (NEWST) {
            // (beyond the last tile the pipeline keeps requesting -- the last tile again: the loads stay in range and nobody reads what they bring)
            const int tq = st_t < total ? st_t : total - 1;
            const int tt = (int)mrx_xcd_band(tq, total);
            const int b = tt / a.ntiles, tile = tt - b * a.ntiles, ty0 = tile / a.tiles_x;
            const int h0 = ty0 * TH, w0 = (tile - ty0 * a.tiles_x) * S2_TW;
            st_rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x) + (long long)b * S2_F * plane, 0, (unsigned)(plane * (S2_F * 4)), 0x00020000);
#pragma unroll
            for (int v = 0; v < XV; ++v) {
                int p = (wave * 64 + lane_now()) + v * NTHR;
                p = p < S2_NPIX ? p : S2_NPIX - 1;
                const int ty = p / S2_PW, tx = p - ty * S2_PW;
                int gy = h0 + ty - S2_PAD, gx = w0 + tx - S2_PAD;
                gy = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy);
                gx = gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx);
                goff32[v] = (unsigned)(gy * a.W + gx) * 32u;
            }
            return;
        }
        if (st_t >= total) return;
        const int tt = (int)mrx_xcd_band(st_t, total);
        const int b = tt / a.ntiles, tile = tt - b * a.ntiles, ty0 = tile / a.tiles_x;
        const int h0 = ty0 * TH, w0 = (tile - ty0 * a.tiles_x) * S2_TW;
        st_xb = a.x + (long long)b * S2_F * plane;
#pragma unroll
        for (int v = 0; v < XV; ++v) {
            int p = (wave * 64 + lane_now()) + v * NTHR;
            p = p < S2_NPIX ? p : S2_NPIX - 1;
            const int ty = p / S2_PW, tx = p - ty * S2_PW;
            int gy = h0 + ty - S2_PAD, gx = w0 + tx - S2_PAD;            // replicate border = clamp (conv_layers.py:72-76)
            if (ZP) {
                const unsigned out = (gy < 0 || gy >= a.H || gx < 0 || gx >= a.W) ? 1u : 0u;
                zmask = v == 0 ? out : (zmask | (out << v));
            }
            gy = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy);
            gx = gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx);
            goff[v] = (long long)gy * a.W + gx;
        }
    };
    float xr[XV][8];
    u32x4 wr[WV];
    bool pending = false;
    auto request_next = [&]() {                      // fp32 values of this thread's pixels and its weight operands of the next chunk
        pending = st_t < total;
        if (!pending) return;
        zpend = zmask;
#pragma unroll
        for (int v = 0; v < XV; ++v)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if constexpr ((ABL & 1) != 0) asm volatile("v_mov_b32 %0, 1.0" : "=v"(xr[v][j]));
                else if constexpr (CB8) {
                    if ((j & 3) == 0) {
                        const float4 u = *reinterpret_cast<const float4*>(st_xb + ((long long)st_q * plane + goff[v]) * 8 + j);
                        xr[v][j] = u.x, xr[v][j + 1] = u.y, xr[v][j + 2] = u.z, xr[v][j + 3] = u.w;
                    }
                } else xr[v][j] = st_xb[(long long)(8 * st_q + j) * plane + goff[v]];
            }
#pragma unroll
        for (int v = 0; v < WV; ++v) {
            const int i = (wave * 64 + lane_now()) + v * NTHR;
            wr[v] = a.packed[(long long)st_q * WCH + (i < WCH ? i : WCH - 1)];
        }
        if (++st_q == S2_NCH) {
            st_q = 0;
            st_t += gridDim.x;
            st_coords();
        }
    };
    auto commit_next = [&](int buf) {                // split into the three bf16 terms, write the planes and the weights
        if (!pending) return;
#pragma unroll
        for (int v = 0; v < XV; ++v) {
            const int p = (wave * 64 + lane_now()) + v * NTHR;
            unsigned p1[4], p2[4], p3[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if constexpr ((ABL & 2) != 0) {
                    p1[k] = __float_as_uint(xr[v][2 * k]), p2[k] = __float_as_uint(xr[v][2 * k + 1]), p3[k] = 0u;
                } else if constexpr (F16) {
                    s2_split2h(xr[v][2 * k] * sx, xr[v][2 * k + 1] * sx, p1[k], p2[k]);
                    p3[k] = 0u;
                } else {
                    s2_split2(xr[v][2 * k], xr[v][2 * k + 1], p1[k], p2[k], p3[k]);
                }
            }
            if (ZP && ((zpend >> v) & 1u)) {
#pragma unroll
                for (int k = 0; k < 4; ++k) p1[k] = p2[k] = p3[k] = 0u;
            }
            if constexpr ((ABL & 4) != 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k) asm volatile("" ::"v"(p1[k]), "v"(p2[k]));
            } else if (p < S2_NPIX) {
                u32x4* dst = Xp + buf * (NT * S2_NPIX) + p;
                dst[0] = u32x4{p1[0], p1[1], p1[2], p1[3]};
                dst[S2_NPIX] = u32x4{p2[0], p2[1], p2[2], p2[3]};
                if constexpr (!F16) dst[2 * S2_NPIX] = u32x4{p3[0], p3[1], p3[2], p3[3]};
            }
        }
#pragma unroll
        for (int v = 0; v < WV; ++v) {
            const int i = (wave * 64 + lane_now()) + v * NTHR;
            if constexpr ((ABL & 4) != 0) asm volatile("" ::"v"(wr[v]));
            else if (i < WCH) Wc[buf * WCH + i] = wr[v];
        }
    };
    // NEWST pieces: the loads of one chunk (x: two 16-byte halves per pixel; weights: three operands), the split of one channel pair group, the LDS writes
    auto ns_load_x = [&](int v, int half) {
        const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(st_rx, goff32[v] + 16u * half, (unsigned)st_q * (unsigned)(plane * 32), 0);
        xr[v][4 * half] = __uint_as_float(u[0]), xr[v][4 * half + 1] = __uint_as_float(u[1]);
        xr[v][4 * half + 2] = __uint_as_float(u[2]), xr[v][4 * half + 3] = __uint_as_float(u[3]);
    };
    auto ns_load_w = [&](int v) { wr[v] = __builtin_amdgcn_raw_buffer_load_b128(st_rw, st_tid16, (unsigned)(st_q * WCH + v * NTHR) * 16u, 0); };
    auto ns_advance = [&]() {                        // after the last load of a chunk
        if (++st_q == S2_NCH) {
            st_q = 0;
            st_t += gridDim.x;
            st_coords();
        }
    };
    unsigned ns_p1[XV][4], ns_p2[XV][4];
    auto ns_split = [&](int v, int k) { s2_split2h_scaled(xr[v][2 * k], xr[v][2 * k + 1], sx, ns_p1[v][k], ns_p2[v][k]); };
    auto ns_write_x = [&](int buf, int v) {
        unsigned char* base = smem_s2 + OFF_X + buf * (XBUF * 16) + (v == XV - 1 ? st_px1 : st_tid16 + (unsigned)(v * NTHR * 16));
        *reinterpret_cast<u32x4*>(base) = u32x4{ns_p1[v][0], ns_p1[v][1], ns_p1[v][2], ns_p1[v][3]};
        *reinterpret_cast<u32x4*>(base + PSTR * 16) = u32x4{ns_p2[v][0], ns_p2[v][1], ns_p2[v][2], ns_p2[v][3]};
    };
    auto ns_write_w = [&](int buf, int v) {
        *reinterpret_cast<u32x4*>(smem_s2 + OFF_W + buf * (WBUF * 16) + (v < WV - 1 ? st_tid16 + (unsigned)(v * NTHR * 16) : st_w2)) = wr[v];
    };
    st_coords();
    if constexpr (NEWST) {
        static_assert(!NEWST || (XV <= 2 && WV <= 3 && 2 * XV + WV < RPW * 6), "the NEWST slices: at most two pixels and three weight operands per thread");
        for (int pre = 0; pre < 2; ++pre) {          // chunk 0 requested, split, written; chunk 1 requested
#pragma unroll
            for (int v = 0; v < XV; ++v) ns_load_x(v, 0), ns_load_x(v, 1);
#pragma unroll
            for (int v = 0; v < WV; ++v) ns_load_w(v);
            ns_advance();
            if (pre == 0) {
#pragma unroll
                for (int v = 0; v < XV; ++v) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) ns_split(v, k);
                    ns_write_x(0, v);
                }
#pragma unroll
                for (int v = 0; v < WV; ++v) ns_write_w(0, v);
            }
        }
    } else {
        request_next();
        commit_next(0);
        request_next();
    }
    __syncthreads();                                 // chunk 0 of the first tile, the 1x1 weights and the tables are in place

    float vmax = 0.f;                                // TAIL = false: maximum |output| of this lane (a.xmax_out)
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        // (lane-derived values per tile, see lane_now)
        const int lane = lane_now(), l31 = lane & 31, lhi = lane >> 5;
        const int tt = (int)mrx_xcd_band(t, total);
        const int b = tt / a.ntiles, tile = tt - b * a.ntiles, ty0 = tile / a.tiles_x;
        const int h0 = ty0 * TH, w0 = (tile - ty0 * a.tiles_x) * S2_TW;

        // (cycle stamps exist in PROBE builds only: even when skipped at run time, each one is a branch that cuts the row tails into separate basic
        // blocks -- the scheduler cannot then place row 1's vector work under row 0's MFMAs)
#ifdef MRX_PROBE
#define S2_STAMP(i) if (a.trace && lane == 0 && (t - (int)blockIdx.x) / (int)gridDim.x < 2) a.trace[(((long long)blockIdx.x * 8 + wave) * 2 + (t - blockIdx.x) / gridDim.x) * 4 + (i)] = __builtin_readcyclecounter();
#else
#define S2_STAMP(i)
#endif
        S2_STAMP(0)
        // acc[row][ct]: rows 2 wave and 2 wave + 1 of the tile
        f32x16 acc[RPW][2];
#pragma unroll
        for (int rw = 0; rw < RPW; ++rw)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rw][ct][r] = (F16 && !TAIL) ? 0.f : tabl[64 + lhi * 32 + ct * 16 + r];

        float hp[RPW][32];
        auto load_hp = [&](int rw) {              // lanes outside the image read a valid element (clamped) and store nothing
            const int oy = h0 + RPW * wave + rw, ox = w0 + l31;
            const int cy = oy < a.H ? oy : a.H - 1, cx = ox < a.W ? ox : a.W - 1;
            if (!a.hprev) {                       // the zero state: nothing to load (and nothing uninitialised to multiply by zero)
#pragma unroll
                for (int R = 0; R < 32; ++R) hp[rw][R] = 0.f;
                return;
            }
            if constexpr (CB8) {
                const float* hb = a.hprev + (long long)b * S2_F * plane + ((long long)cy * a.W + cx) * 8 + 4 * lhi;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    typedef float f32x4_ __attribute__((ext_vector_type(4)));
                    const f32x4_ u = MRX_L2_NT_LD ? __builtin_nontemporal_load(reinterpret_cast<const f32x4_*>(hb + (long long)q * plane * 8))
                                                  : *reinterpret_cast<const f32x4_*>(hb + (long long)q * plane * 8);
                    hp[rw][4 * q] = u.x, hp[rw][4 * q + 1] = u.y, hp[rw][4 * q + 2] = u.z, hp[rw][4 * q + 3] = u.w;
                }
                return;
            }
            const float* hb = a.hprev + (long long)b * S2_F * plane + (long long)cy * a.W + cx + 4ll * lhi * plane;
#pragma unroll
            for (int R = 0; R < 32; ++R) {
                if constexpr ((ABL & 128) != 0) asm volatile("v_mov_b32 %0, 1.0" : "=v"(hp[rw][R]));
                else hp[rw][R] = hb[(long long)s2_chan(R, 0) * plane];
            }
        };
        auto toff = [](int tp) { return tp < 9 ? (tp / 3) * DIL * S2_PW + (tp % 3) * DIL : 0; };  // the zero-weight slot reads pixel 0
        for (int q = 0; q < S2_NCH; ++q) {
            const u32x4* xw = Xp + (q & 1) * XBUF + (RPW * wave) * S2_PW + l31;
            const u32x4* wl = Wc + (q & 1) * WBUF + lane;
            // Ninth tap: chunks are paired.  The fifth step of an EVEN chunk multiplies tap 8 of this chunk (lower half-wave) and tap 8 of the
            // NEXT chunk (upper half-wave: its planes and weights were committed at step 1 of this chunk -- hence the extra barrier before
            // they are fetched); an odd chunk has four steps.  36 instead of 40 MFMA steps per tile, no padding slot.
            const bool even = !(q & 1);
            const u32x4* xw8 = Xp + ((q + lhi) & 1) * XBUF + (RPW * wave) * S2_PW + l31 + toff(8);
            const u32x4* w8 = Wc + ((q + lhi) & 1) * WBUF + WFULL + l31;
            u32x4 bt[2][RPW][NT], at[2][2][NT];     // [buffer][row | ct][term]
            auto fetch = [&](int s, int bf) {
                if constexpr ((ABL & 32) != 0) {
#pragma unroll
                    for (int k = 0; k < NT; ++k)
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            bt[bf][i][k] = u32x4{0x3c003c00u + (unsigned)s, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
                            at[bf][i][k] = u32x4{0x3c003c00u, 0x3c003c00u + (unsigned)q, 0x3c003c00u, 0x3c003c00u};
                            asm volatile("" : "+v"(bt[bf][i][k]), "+v"(at[bf][i][k]));
                        }
                } else if (s < 4) {
                    const int off = lhi ? toff(2 * s + 1) : toff(2 * s);
#pragma unroll
                    for (int k = 0; k < NT; ++k) {
                        bt[bf][0][k] = xw[k * PSTR + off];
                        if constexpr (RPW == 2) bt[bf][1][k] = xw[k * PSTR + off + S2_PW];
                        at[bf][0][k] = wl[((s * NT + k) * 2 + 0) * 64];
                        at[bf][1][k] = wl[((s * NT + k) * 2 + 1) * 64];
                    }
                } else {
#pragma unroll
                    for (int k = 0; k < NT; ++k) {
                        bt[bf][0][k] = xw8[k * PSTR];
                        if constexpr (RPW == 2) bt[bf][1][k] = xw8[k * PSTR + S2_PW];
                        at[bf][0][k] = w8[(k * 2 + 0) * 32];
                        at[bf][1][k] = w8[(k * 2 + 1) * 32];
                    }
                }
            };
            fetch(0, 0);
#pragma unroll
            for (int s = 0; s < S2_KS; ++s) {
                const int bf = s & 1;
                if (s == 4 && !even) break;
                if (s == S2_BAR_STEP && even) {        // every thread's commit of chunk q + 1 is visible
                    if constexpr (!(ABL & 64)) __syncthreads();
                    if (S2_BAR_STEP == 4) fetch(4, bf);
                }
                if (s + 1 < 4 || (s + 1 == 4 && even && S2_BAR_STEP == 3)) fetch(s + 1, bf ^ 1);
                // the six term pairs of weight >= 2^-16, smallest first; the four accumulators alternate
#define S2_P(TA, TB)                                                                                                                       \
    _Pragma("unroll") for (int rw = 0; rw < RPW; ++rw) _Pragma("unroll") for (int ct = 0; ct < 2; ++ct) acc[rw][ct] =                          \
        __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, at[bf][ct][TA]), __builtin_bit_cast(bf16x8, bt[bf][rw][TB]), acc[rw][ct], 0, 0, 0);
                if constexpr ((ABL & 8) != 0) {
#pragma unroll
                    for (int k = 0; k < NT; ++k)
#pragma unroll
                        for (int i = 0; i < 2; ++i) asm volatile("" ::"v"(bt[bf][i][k]), "v"(at[bf][i][k]));
                } else if constexpr (NEWST) {
                    // the same twelve MFMAs (three products x two rows x two cout blocks, smallest product first), one slice of the side work behind each
                    constexpr int TA_[3] = {1, 0, 0}, TB_[3] = {0, 1, 0};
                    constexpr int NM = 6 * RPW;                // MFMAs per step
                    const int nb = (q + 1) & 1;
#pragma unroll
                    for (int m = 0; m < NM; ++m) {
                        const int rw = RPW == 2 ? (m >> 1) & 1 : 0, ct = m & 1, pr = m / (2 * RPW);
                        acc[rw][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, at[bf][ct][TA_[pr]]), __builtin_bit_cast(f16x8, bt[bf][rw][TB_[pr]]),
                                                                            acc[rw][ct], 0, 0, 0);
                        if (s == 2) {                            // chunk q + 1: split (fp32 values requested one chunk ago) and written
                            if constexpr (RPW == 2) {
                                if (m < 4) ns_split(0, m);
                                else if (m == 4) ns_write_x(nb, 0);
                                else if (m < 9 && XV == 2) ns_split(1, m - 5);
                                else if (m == 9 && XV == 2) ns_write_x(nb, 1);
                                else if (m == 10) ns_write_w(nb, 0), ns_write_w(nb, 1);
                                else if (m == 11 && WV == 3) ns_write_w(nb, 2);
                            } else {                             // six MFMAs, one pixel, two weight operands
                                if (m < 4) ns_split(0, m);
                                else if (m == 4) ns_write_x(nb, 0);
                                else {
#pragma unroll
                                    for (int v = 0; v < WV; ++v) ns_write_w(nb, v);
                                }
                            }
                        } else if (s == 3) {                     // chunk q + 2 requested
                            if (m < 2 * XV) ns_load_x(m >> 1, m & 1);
                            else if (m < 2 * XV + WV) ns_load_w(m - 2 * XV);
                            else if (m == 2 * XV + WV) ns_advance();
                        }
                        if (s == 2 || s == 3) __builtin_amdgcn_sched_barrier(0);
                    }
                } else if constexpr (F16) {
                    // two fp16 terms per operand: the three products of weight >= 2^-11, smallest first
#define S2_PH16(TA, TB)                                                                                                                    \
    _Pragma("unroll") for (int rw = 0; rw < RPW; ++rw) _Pragma("unroll") for (int ct = 0; ct < 2; ++ct) acc[rw][ct] =                          \
        __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, at[bf][ct][TA]), __builtin_bit_cast(f16x8, bt[bf][rw][TB]), acc[rw][ct], 0, 0, 0);
                    S2_PH16(1, 0) S2_PH16(0, 1) S2_PH16(0, 0)
#undef S2_PH16
                } else {
                    S2_P(NT - 1, 0) S2_P(0, NT - 1) S2_P(1, 1) S2_P(1, 0) S2_P(0, 1) S2_P(0, 0)
                }
#undef S2_P
                // (committing one step later in the second wave of each SIMD -- so that one wave's vector-ALU block meets the other's MFMAs --
                // measured equal: 69.0 vs 67.3 us, lib 226)
                if (!NEWST && s == (even ? S2_COMMIT_EVEN : 1)) {   // the readers of the other buffer passed the previous barrier
                    commit_next((q + 1) & 1);
                    request_next();
                }
                if (TAIL && !(ABL & 16) && RPW == 2 && s == 2 && q == S2_NCH - 1) load_hp(0);     // (one row per wave: requested behind the 1x1 stage -- 128 registers)
            }
            if constexpr (!(ABL & 64)) __syncthreads();
        }

        if constexpr (F16 && !TAIL) {
            const float unxw = unx * unw;
#pragma unroll
            for (int rw = 0; rw < RPW; ++rw)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[rw][ct][r] = acc[rw][ct][r] * unxw + tabl[64 + lhi * 32 + ct * 16 + r];
        }
        S2_STAMP(1)
        if constexpr ((ABL & 16) != 0) {
#pragma unroll
            for (int rw = 0; rw < RPW; ++rw)
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) asm volatile("" ::"v"(acc[rw][ct]));
        } else if constexpr (TAIL && F16 && CB8 && FAST && RPW == 2) {
        // ---- both rows of the wave through the 1x1 stage TOGETHER (round 5) ----------------------------------------------------------------------
        // The row tails are issue-bound, not matrix-bound: 72 MFMAs in ~1 500 instructions, the two waves of a SIMD in the same phase.  Here a
        // weight fragment read from LDS serves both rows (half the A reads), every MFMA step has four independent accumulators, row 1's operand
        // split sits next to row 0's MFMAs in ONE basic block (stores go through buffer descriptors with an out-of-range offset for pixels outside
        // the image instead of a branch), and the same for the tap stage.  Per accumulator the order of the term products is the row-by-row
        // form's: bit-identical results.
        const int oy0 = h0 + 2 * wave, ox = w0 + l31;
        float sg[2], ung[2];
#pragma unroll
        for (int rw = 0; rw < 2; ++rw) {
            float gm = 0.f;
#pragma unroll
            for (int R = 0; R < 32; ++R) gm = fmaxf(gm, acc[rw][R >> 4][R & 15]);
            const int kg = s2_pixel_exp(gm);
            sg[rw] = s2_pow2(kg), ung[rw] = s2_pow2(-kg) * (unx * unw * unwi);
        }
        f32x16 acc2[2][2];
#pragma unroll
        for (int rw = 0; rw < 2; ++rw)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[rw][ct][r] = 0.f;
        {
            const u32x4* wl = Wih + lane;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                f16x8 b1[2], b2[2];
#pragma unroll
                for (int rw = 0; rw < 2; ++rw) {
                    unsigned g1[4], g2[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int R0 = 8 * s + 2 * k, R1 = R0 + 1;
                        float v0 = acc[rw][R0 >> 4][R0 & 15], v1 = acc[rw][R1 >> 4][R1 & 15];
                        v0 = v0 > 0.f ? v0 : 0.f;
                        v1 = v1 > 0.f ? v1 : 0.f;
                        s2_split2h_scaled(v0, v1, sg[rw], g1[k], g2[k]);
                    }
                    b1[rw] = __builtin_bit_cast(f16x8, (u32x4{g1[0], g1[1], g1[2], g1[3]}));
                    b2[rw] = __builtin_bit_cast(f16x8, (u32x4{g2[0], g2[1], g2[2], g2[3]}));
                }
                f16x8 at[2][2];
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) at[ct][k] = __builtin_bit_cast(f16x8, wl[((s * 2 + k) * 2 + ct) * 64]);
#pragma unroll
                for (int rw = 0; rw < 2; ++rw)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) acc2[rw][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(at[ct][1], b1[rw], acc2[rw][ct], 0, 0, 0);
#pragma unroll
                for (int rw = 0; rw < 2; ++rw)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) acc2[rw][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(at[ct][0], b2[rw], acc2[rw][ct], 0, 0, 0);
#pragma unroll
                for (int rw = 0; rw < 2; ++rw)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) acc2[rw][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(at[ct][0], b1[rw], acc2[rw][ct], 0, 0, 0);
                if (s == 1) load_hp(1);          // (half of the convolution's accumulators are dead by now: row 1's state can land in their registers)
            }
        }
        S2_STAMP(2)        // (probe builds: [1, 2] = both rows' 1x1 stage, [2, 3] = epilogues, stores and the tap stage)
        const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(a.hnew + (long long)b * S2_F * plane, 0, (unsigned)(plane * (S2_F * 4)), 0x00020000);
        unsigned offh[2];
#pragma unroll
        for (int rw = 0; rw < 2; ++rw) {
            const int oy = oy0 + rw;
            offh[rw] = (oy < a.H && ox < a.W) ? (unsigned)((((long long)oy * a.W + ox) * 8 + 4 * lhi) * 4) : 0x80000000u;
#pragma unroll
            for (int R = 0; R < 32; ++R) {
                float v = acc2[rw][R >> 4][R & 15] * ung[rw] + tabl[128 + lhi * 32 + R];
                v += tabl[lhi * 32 + R] * hp[rw][R];
                hp[rw][R] = v > 0.f ? v : 0.f;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(hp[rw][4 * q]), __float_as_uint(hp[rw][4 * q + 1]), __float_as_uint(hp[rw][4 * q + 2]),
                                                             __float_as_uint(hp[rw][4 * q + 3])},
                                                       rh, offh[rw] + (unsigned)q * (unsigned)(plane * 32), 0, MRX_L2_NT_ST ? 2 : 0);
        }
        if (a.P || a.Q) {
            f32x16 accp[2];
            float sh[2], unh[2];
#pragma unroll
            for (int rw = 0; rw < 2; ++rw) {
#pragma unroll
                for (int r = 0; r < 16; ++r) accp[rw][r] = 0.f;
                float hm = 0.f;
#pragma unroll
                for (int R = 0; R < 32; ++R) hm = fmaxf(hm, hp[rw][R]);
                const int kh = s2_pixel_exp(hm);
                sh[rw] = s2_pow2(kh), unh[rw] = s2_pow2(-kh) * unwp;
            }
            const u32x4* wp = Wih + S2_WIH + lane;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                f16x8 b1[2], b2[2];
#pragma unroll
                for (int rw = 0; rw < 2; ++rw) {
                    unsigned g1[4], g2[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) s2_split2h_scaled(hp[rw][8 * s + 2 * k], hp[rw][8 * s + 2 * k + 1], sh[rw], g1[k], g2[k]);
                    b1[rw] = __builtin_bit_cast(f16x8, (u32x4{g1[0], g1[1], g1[2], g1[3]}));
                    b2[rw] = __builtin_bit_cast(f16x8, (u32x4{g2[0], g2[1], g2[2], g2[3]}));
                }
                const f16x8 a1 = __builtin_bit_cast(f16x8, wp[(s * 2 + 0) * 64]);
                const f16x8 a2 = __builtin_bit_cast(f16x8, wp[(s * 2 + 1) * 64]);
#pragma unroll
                for (int rw = 0; rw < 2; ++rw) accp[rw] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b1[rw], accp[rw], 0, 0, 0);
#pragma unroll
                for (int rw = 0; rw < 2; ++rw) accp[rw] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b2[rw], accp[rw], 0, 0, 0);
#pragma unroll
                for (int rw = 0; rw < 2; ++rw) accp[rw] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1[rw], accp[rw], 0, 0, 0);
            }
            if (a.Q) {
                // Eighteen tap planes become six: the three products of a kernel row meet inside the wave's image row (lane = pixel), so the gather that
                // follows (k_llg372<GAT>, k_l2sb_gather_q) reads 6 + 2 instead of 18 + 2 values per pixel and this kernel stores 3 instead of 10 per lane.
                // Lanes of the lower half-wave hold products m = 0..3, 8..11, 16, 17 (m = (dy * 3 + dx) * 2 + co), those of the upper half 4..7, 12..15.
                // Column 0 / 31 of the tile miss their left / right neighbour unless that is the image border (replicate padding: the pixel itself); what
                // they miss the neighbouring tile leaves in E.
                const int tcol = (w0 / S2_TW);
                const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(a.Q + (long long)b * 6 * plane, 0, (unsigned)(plane * (6 * 4)), 0x00020000);
                const long long eslots = (long long)a.H * a.tiles_x;       // 16 floats per (row, tile column)
                const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc(a.E + (long long)b * 16 * eslots, 0, (unsigned)(eslots * 64), 0x00020000);
                const bool hasL = l31 > 0, hasR = l31 < 31 && ox + 1 < a.W;
                const bool useL = hasL || ox == 0, useR = l31 < 31 || ox == a.W - 1;
                // x - 1 / x + 1 are wave shifts by one lane (DPP; the lanes where the shift crosses the half-wave are columns 0 / 31: masked, or the replicated border)
                auto fromL = [&](float v_) {                         // the value of the pixel to the left (the pixel itself on the image border, 0 across a tile border)
                    const float t = __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v_), 0x138, 0xf, 0xf, false));   // wave_shr:1
                    return useL ? (hasL ? t : v_) : 0.f;
                };
                auto fromR = [&](float v_) {
                    const float t = __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v_), 0x130, 0xf, 0xf, false));   // wave_shl:1
                    return useR ? (hasR ? t : v_) : 0.f;
                };
#pragma unroll
                for (int rw = 0; rw < 2; ++rw) {
                    const int oy = oy0 + rw;
                    const bool inside = oy < a.H && ox < a.W;
                    float v[10];
#pragma unroll
                    for (int r = 0; r < 10; ++r) v[r] = accp[rw][r] * unh[rw];
                    // plane dy * 2 + co = A(dx 0, x - 1) + B(dx 1, x) + C(dx 2, x + 1); the products sit in (register, half-wave):
                    //   plane 0, 1: A (co, lower)      B (2 + co, lower)   C (co, upper)        -> stored by the lower half-wave
                    //   plane 2, 3: A (2 + co, upper)  B (4 + co, lower)   C (6 + co, lower)    -> plane 2 by the lower, plane 3 by the upper half-wave
                    //   plane 4, 5: A (4 + co, upper)  B (6 + co, upper)   C (8 + co, lower)    -> stored by the upper half-wave
                    // Three half-wave swaps carry what the other half holds (each serves a lower-stored and an upper-stored plane).
                    const float sw0 = __shfl_xor(lhi ? fromR(v[0]) : v[5] + fromR(v[7]), 32, 64);      // lower gets C of plane 0; upper gets B + C of plane 3
                    const float sw1 = __shfl_xor(lhi ? fromR(v[1]) : fromR(v[8]), 32, 64);             // lower gets C of plane 1; upper gets C of plane 4
                    const float sw2 = __shfl_xor(lhi ? fromL(v[2]) : fromR(v[9]), 32, 64);             // lower gets A of plane 2; upper gets C of plane 5
                    float q[3];
                    q[0] = lhi ? fromL(v[3]) + sw0 : (fromL(v[0]) + v[2]) + sw0;                       // plane 3 | plane 0
                    q[1] = lhi ? (fromL(v[4]) + v[6]) + sw1 : (fromL(v[1]) + v[3]) + sw1;              // plane 4 | plane 1
                    q[2] = lhi ? (fromL(v[5]) + v[7]) + sw2 : sw2 + (v[4] + fromR(v[6]));              // plane 5 | plane 2
                    // Q[b][dy][y][x][co]: the lower half-wave stores the pair of dy 0 and the first half of dy 1's, the upper the second half of dy 1's and the pair
                    // of dy 2 -- one 8-byte and one 4-byte store per lane (the gathers read one float2 per kernel row)
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    const unsigned offq = inside ? (unsigned)(((long long)oy * a.W + ox) * 8) : 0x80000000u;
                    const u32x2 pair = lhi ? u32x2{__float_as_uint(q[1]), __float_as_uint(q[2])} : u32x2{__float_as_uint(q[0]), __float_as_uint(q[1])};
                    __builtin_amdgcn_raw_buffer_store_b64(pair, rq, offq + (lhi ? 2u : 0u) * (unsigned)(plane * 8), 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(lhi ? q[0] : q[2]), rq, offq + (unsigned)(plane * 8) + (lhi ? 4u : 0u), 0, 0);
                    // what the neighbouring tiles miss, ONE 16-byte store per row by the four lanes that hold it -- E[b][y][tile column][16]:
                    //   [0..3]  column 0, lower lane: dx = 2 products of dy 1, 2 (registers 6, 7, 8, 9)      [4..5]   column 0, upper lane: dx = 2 products of dy 0 (registers 0, 1)
                    //   [8..9]  column 31, lower lane: dx = 0 products of dy 0 (registers 0, 1)              [12..15] column 31, upper lane: dx = 0 products of dy 1, 2 (registers 2..5)
                    const bool col0 = l31 == 0 && inside && tcol > 0, col31 = l31 == 31 && inside && tcol + 1 < a.tiles_x;
                    const u32x4 ev = l31 == 0 ? (lhi ? u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), 0u, 0u}
                                                     : u32x4{__float_as_uint(v[6]), __float_as_uint(v[7]), __float_as_uint(v[8]), __float_as_uint(v[9])})
                                              : (lhi ? u32x4{__float_as_uint(v[2]), __float_as_uint(v[3]), __float_as_uint(v[4]), __float_as_uint(v[5])}
                                                     : u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), 0u, 0u});
                    const unsigned eoff = (col0 || col31) ? (unsigned)(((long long)oy * a.tiles_x + tcol) * 64 + (l31 == 0 ? (lhi ? 16 : 0) : (lhi ? 48 : 32))) : 0x80000000u;
                    __builtin_amdgcn_raw_buffer_store_b128(ev, re, eoff, 0, 0);
                }
            } else {
            const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(a.P + (long long)b * 18 * plane, 0, (unsigned)(plane * (18 * 4)), 0x00020000);
#pragma unroll
            for (int rw = 0; rw < 2; ++rw) {
                const int oy = oy0 + rw;
                const bool inside = oy < a.H && ox < a.W;
                const unsigned offp = inside ? (unsigned)((((long long)oy * a.W + ox) + 4ll * lhi * plane) * 4) : 0x80000000u;
                const unsigned offp16 = (inside && !lhi) ? offp : 0x80000000u;       // rows 16, 17: the upper half-wave's 20, 21 do not exist
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(accp[rw][r] * unh[rw]), rp, offp + (unsigned)((r & 3) + 8 * (r >> 2)) * (unsigned)(plane * 4), 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(accp[rw][8] * unh[rw]), rp, offp16 + 16u * (unsigned)(plane * 4), 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(accp[rw][9] * unh[rw]), rp, offp16 + 17u * (unsigned)(plane * 4), 0, 0);
            }
        }
            }
        S2_STAMP(3)
        } else if constexpr (TAIL) {
        // ---- g = ReLU(conv + b) in registers; h = ReLU(W_ih g + b_ih + hh * h_prev), one of the wave's two rows at a time ------------------
        // (h_prev of row 0 was requested inside the last chunk; row 1's request goes out now and hides under row 0's tail)
        if constexpr (RPW == 2) load_hp(1);
#pragma unroll
        for (int rw = 0; rw < RPW; ++rw) {
            const int oy = h0 + RPW * wave + rw, ox = w0 + l31;
            f32x16 acc2[2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[ct][r] = F16 ? 0.f : tabl[128 + lhi * 32 + ct * 16 + r];
            const u32x4* wl = Wih + lane;
            if constexpr (F16) {
                // two fp16 terms, scaled per PIXEL: the contraction runs over the pixel's 64 channels only (this lane's 32 and lane ^ 32's).  The
                // accumulators hold (conv + b) 2^kx 2^kw; the pixel's exponent is taken from them as they are, and ONE factor per lane takes the 1x1
                // stage's sums back: 2^-kg 2^-kx 2^-kw 2^-kwi (all exact)
                float gm = 0.f;
#pragma unroll
                for (int R = 0; R < 32; ++R) gm = fmaxf(gm, acc[rw][R >> 4][R & 15]);
                const int kg = s2_pixel_exp(gm);
                const float sg = s2_pow2(kg), ung = s2_pow2(-kg) * (unx * unw * unwi);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    unsigned g1[4], g2[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int R0 = 8 * s + 2 * k, R1 = R0 + 1;
                        float v0 = acc[rw][R0 >> 4][R0 & 15], v1 = acc[rw][R1 >> 4][R1 & 15];
                        v0 = v0 > 0.f ? v0 : 0.f;
                        v1 = v1 > 0.f ? v1 : 0.f;
                        s2_split2h_scaled(v0, v1, sg, g1[k], g2[k]);
                    }
                    const f16x8 b1 = __builtin_bit_cast(f16x8, (u32x4{g1[0], g1[1], g1[2], g1[3]}));
                    const f16x8 b2 = __builtin_bit_cast(f16x8, (u32x4{g2[0], g2[1], g2[2], g2[3]}));
                    f16x8 at[2][2];
#pragma unroll
                    for (int k = 0; k < 2; ++k)
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct) at[ct][k] = __builtin_bit_cast(f16x8, wl[((s * 2 + k) * 2 + ct) * 64]);
                    S2_MFMA6H(acc2, at, b1, b2)
                }
                if constexpr (RPW == 1) load_hp(0);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc2[ct][r] = acc2[ct][r] * ung + tabl[128 + lhi * 32 + ct * 16 + r];
            } else
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                unsigned g1[4], g2[4], g3[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int R0 = 8 * s + 2 * k, R1 = R0 + 1;
                    float v0 = acc[rw][R0 >> 4][R0 & 15], v1 = acc[rw][R1 >> 4][R1 & 15];
                    v0 = v0 > 0.f ? v0 : 0.f;
                    v1 = v1 > 0.f ? v1 : 0.f;
                    s2_split2(v0, v1, g1[k], g2[k], g3[k]);
                }
                const bf16x8 b1 = __builtin_bit_cast(bf16x8, (u32x4{g1[0], g1[1], g1[2], g1[3]}));
                const bf16x8 b2 = __builtin_bit_cast(bf16x8, (u32x4{g2[0], g2[1], g2[2], g2[3]}));
                const bf16x8 b3 = __builtin_bit_cast(bf16x8, (u32x4{g3[0], g3[1], g3[2], g3[3]}));
                bf16x8 at[2][3];
#pragma unroll
                for (int k = 0; k < 3; ++k)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) at[ct][k] = __builtin_bit_cast(bf16x8, wl[((s * 3 + k) * 2 + ct) * 64]);
                S2_MFMA12(acc2, at, b1, b2, b3)
            }
            const bool inside = oy < a.H && ox < a.W;
            {
                // (the arithmetic first, then ONE predicated block of stores: a branch around every store splits the epilogue into 32 basic
                // blocks, each waiting on its own LDS table read -- 10 k of the 23 k tail cycles per tile)
                float* ob = a.hnew + (long long)b * S2_F * plane + (long long)oy * a.W + ox + 4ll * lhi * plane;
#pragma unroll
                for (int R = 0; R < 32; ++R) {
                    float v = acc2[R >> 4][R & 15] + tabl[lhi * 32 + R] * hp[rw][R];
                    hp[rw][R] = v > 0.f ? v : 0.f;
                }
                if constexpr ((ABL & 256) != 0) {
#pragma unroll
                    for (int R = 0; R < 32; ++R) asm volatile("" ::"v"(hp[rw][R]));
                } else if constexpr ((ABL & 1024) != 0) {      // the form before lib 224: a branch around every store
#pragma unroll
                    for (int R = 0; R < 32; ++R) {
                        asm volatile("" : "+v"(hp[rw][R]));
                        if (inside) ob[(long long)s2_chan(R, 0) * plane] = hp[rw][R];
                    }
                } else if (inside) {
                    if constexpr (CB8) {
                        float* oc = a.hnew + (long long)b * S2_F * plane + ((long long)oy * a.W + ox) * 8 + 4 * lhi;
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                            *reinterpret_cast<float4*>(oc + (long long)q * plane * 8) = make_float4(hp[rw][4 * q], hp[rw][4 * q + 1], hp[rw][4 * q + 2], hp[rw][4 * q + 3]);
                    } else {
#pragma unroll
                        for (int R = 0; R < 32; ++R) ob[(long long)s2_chan(R, 0) * plane] = hp[rw][R];
                    }
                }
            }
            if (a.P && !(ABL & 512)) {
                // the final 64 -> 2 convolution's channel contraction on the new state while it is in registers: D[tap * 2 + co][pixel]
                // (18 of the 32 rows), the k order of the B operand is the register order of h_new as in the 1x1 stage above
                f32x16 accp;
#pragma unroll
                for (int r = 0; r < 16; ++r) accp[r] = 0.f;
                const u32x4* wp = Wih + S2_WIH + lane;
                if constexpr (F16) {
                    float hm = 0.f;
#pragma unroll
                    for (int R = 0; R < 32; ++R) hm = fmaxf(hm, hp[rw][R]);
                    const int kh = s2_pixel_exp(hm);
                    const float sh = s2_pow2(kh), unh = s2_pow2(-kh) * unwp;
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        unsigned g1[4], g2[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) s2_split2h_scaled(hp[rw][8 * s + 2 * k], hp[rw][8 * s + 2 * k + 1], sh, g1[k], g2[k]);
                        const f16x8 b1 = __builtin_bit_cast(f16x8, (u32x4{g1[0], g1[1], g1[2], g1[3]}));
                        const f16x8 b2 = __builtin_bit_cast(f16x8, (u32x4{g2[0], g2[1], g2[2], g2[3]}));
                        const f16x8 a1 = __builtin_bit_cast(f16x8, wp[(s * 2 + 0) * 64]);
                        const f16x8 a2 = __builtin_bit_cast(f16x8, wp[(s * 2 + 1) * 64]);
                        accp = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b1, accp, 0, 0, 0);
                        accp = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b2, accp, 0, 0, 0);
                        accp = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, accp, 0, 0, 0);
                    }
#pragma unroll
                    for (int r = 0; r < 10; ++r) accp[r] = accp[r] * unh;          // (rows 0 .. 9 of the lane's 16 are the ones stored)
                } else
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    unsigned g1[4], g2[4], g3[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) s2_split2(hp[rw][8 * s + 2 * k], hp[rw][8 * s + 2 * k + 1], g1[k], g2[k], g3[k]);
                    const bf16x8 b1 = __builtin_bit_cast(bf16x8, (u32x4{g1[0], g1[1], g1[2], g1[3]}));
                    const bf16x8 b2 = __builtin_bit_cast(bf16x8, (u32x4{g2[0], g2[1], g2[2], g2[3]}));
                    const bf16x8 b3 = __builtin_bit_cast(bf16x8, (u32x4{g3[0], g3[1], g3[2], g3[3]}));
                    const bf16x8 a1 = __builtin_bit_cast(bf16x8, wp[(s * 3 + 0) * 64]);
                    const bf16x8 a2 = __builtin_bit_cast(bf16x8, wp[(s * 3 + 1) * 64]);
                    const bf16x8 a3 = __builtin_bit_cast(bf16x8, wp[(s * 3 + 2) * 64]);
                    accp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, accp, 0, 0, 0);
                    accp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, accp, 0, 0, 0);
                    accp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, accp, 0, 0, 0);
                    accp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, accp, 0, 0, 0);
                    accp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, accp, 0, 0, 0);
                    accp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, accp, 0, 0, 0);
                }
                if constexpr ((ABL & 256) != 0) {
#pragma unroll
                    for (int r = 0; r < 10; ++r) asm volatile("" ::"v"(accp[r]));
                } else if (inside) {
                    float* pb = a.P + (long long)b * 18 * plane + (long long)oy * a.W + ox + 4ll * lhi * plane;
#pragma unroll
                    for (int r = 0; r < 8; ++r) pb[(long long)((r & 3) + 8 * (r >> 2)) * plane] = accp[r];   // rows m0 (lower half-wave) / m0 + 4 (upper)
                    if (!lhi) {                                                                              // rows 16, 17: the upper half-wave's 20, 21 do not exist
                        pb[16ll * plane] = accp[8];
                        pb[17ll * plane] = accp[9];
                    }
                }
            }
            if (rw == 0) { S2_STAMP(2) } else { S2_STAMP(3) }
        }
        } else {
            // ---- plain convolution: act(conv + b), the wave's two rows with lane = pixel (128-byte rows per wave instruction) ------------------
#pragma unroll
            for (int rw = 0; rw < RPW; ++rw) {
                const int oy = h0 + RPW * wave + rw, ox = w0 + l31;
                if (oy < a.H && ox < a.W) {
                    float* ob = a.hnew + (long long)b * S2_F * plane + (long long)oy * a.W + ox + 4ll * lhi * plane;
#pragma unroll
                    for (int R = 0; R < 32; ++R) {
                        float v = acc[rw][R >> 4][R & 15];
                        if (a.act == MRX_ACT_RELU) v = v > 0.f ? v : 0.f;
                        else if (a.act == MRX_ACT_LEAKY) v = v > 0.f ? v : v * a.slope;
                        ob[(long long)s2_chan(R, 0) * plane] = v;
                        vmax = fmaxf(vmax, fabsf(v));
                    }
                }
            }
        }
    }
    if constexpr (!TAIL) {
        if (a.xmax_out) {           // one conditional atomic per workgroup (bit patterns of non-negative floats order like unsigned integers)
            for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
            __syncthreads();        // (every wave is past its last operand read: the start of the x planes is free)
            float* red = reinterpret_cast<float*>(smem_s2 + OFF_X);
            if (lane == 0) red[wave] = vmax;
            __syncthreads();
            if (wave == 0 && lane_now() == 0) {
                for (int w = 1; w < NTHR / 64; ++w) vmax = fmaxf(vmax, red[w]);
                if (__float_as_uint(vmax) > __hip_atomic_load(a.xmax_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a.xmax_out, __float_as_uint(vmax));
            }
        }
    }
}

extern "C" int64_t mrx_rim_layer2_sb_pack_floats(void) { return (int64_t)S2_PACK_U4 * 4; }

// w_conv [64,64,3,3] (dilation 2, replicate padding), w_ih [64,64,1,1] -> the split-bf16 operand pack of mrx_rim_layer2_sb;
// w_final [2,64,3,3] (or null): the operands of the final convolution's channel contraction (mrx_rim_layer2_sb_final)
extern "C" int mrx_rim_layer2_sb_pack(const float* w_conv, const float* w_ih, const float* w_final, float* packed, void* stream) {
    MRX_REQUIRE(w_conv && packed, MRX_EINVAL, "mrx_rim_layer2_sb_pack: null pointer");
    hipLaunchKernelGGL(k_l2sb_pack, dim3((S2_PACK_U4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_conv, w_ih, w_final, reinterpret_cast<u32x4*>(packed));
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// h_new = ReLU(W_ih ReLU(conv3x3_d2(replicate_pad(x)) + b_conv) + b_ih + hh * h_prev), F = 64 (rim_block.py:233-238 for the second layer)
static int l2sb_ncu() {
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return ncu;
}
template <int DIL, bool TAIL, bool ZP, bool F16 = false, int ABL = 0, bool CB8 = false, bool W4 = false, bool W16 = false, bool FAST = false>
static int l2sb_launch_t(L2sbArgs a, hipStream_t st) {
    constexpr int WCH_ = F16 ? S2F_WCH : S2_WCH, NT_ = F16 ? 2 : 3;
    constexpr int lds = W4 ? 1024 + 2 * WCH_ * 16 + 2 * NT_ * (8 + 2 * DIL) * (S2_TW + 2 * DIL) * 16 : (int)S2_LDS;
    static bool attr_done = false;   // once per instantiation: keeps launches legal under hipGraph capture
    if (!attr_done) {
        MRX_HIP(hipFuncSetAttribute((const void*)k_rim_layer2_sb<DIL, TAIL, ZP, F16, ABL, CB8, W4, W16, FAST>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        attr_done = true;
    }
    if (W4) a.ntiles = a.tiles_x * mrx_cdiv(a.H, 8);                   // 8 x 32 tiles
    const int ncu = (W4 ? 2 : 1) * l2sb_ncu();                          // two 4-wave workgroups per CU
    const long long total = (long long)a.ntiles * a.B;
    const int grid = (int)(total < ncu ? total : ncu);
    a.trace = nullptr;
    static unsigned long long* d_trace = nullptr;
    if (TAIL && MRX_DEBUG_ENV("MRX_L2SB_TRACE")) {
        if (!d_trace) (void)hipMalloc((void**)&d_trace, sizeof(unsigned long long) * 512 * 8 * 8);
        (void)hipMemsetAsync(d_trace, 0, sizeof(unsigned long long) * 512 * 8 * 8, st);
        a.trace = d_trace;
    }
    hipLaunchKernelGGL((k_rim_layer2_sb<DIL, TAIL, ZP, F16, ABL, CB8, W4, W16, FAST>), dim3(grid), dim3(W16 ? 1024 : (W4 ? 256 : S2_NT)), lds, st, a);
    MRX_LAUNCH_CHECK();
    if (a.trace) {
        (void)hipStreamSynchronize(st);
        std::vector<unsigned long long> h((size_t)grid * 64);
        (void)hipMemcpy(h.data(), d_trace, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost);
        double ph[3] = {0, 0, 0}, gap = 0;
        long n = 0, ng = 0;
        for (int i = 0; i < grid * 8; ++i)
            for (int k = 0; k < 2; ++k) {
                const unsigned long long* r = &h[((size_t)i * 2 + k) * 4];
                if (!r[0] || !r[3]) continue;
                ph[0] += (double)(r[1] - r[0]), ph[1] += (double)(r[2] - r[1]), ph[2] += (double)(r[3] - r[2]);
                ++n;
                if (k == 1 && h[(size_t)i * 8 + 3]) gap += (double)(r[0] - h[(size_t)i * 8 + 3]), ++ng;
            }
        fprintf(stderr, "[l2sb-trace] %ld wave-tiles: chunk loop %.0f, row 0 tail %.0f, row 1 tail %.0f cycles; gap between tiles %.0f\n", n, ph[0] / n,
                ph[1] / n, ph[2] / n, ng ? gap / ng : 0.0);
    }
    return MRX_OK;
}
static int l2sb_launch(const float* x, const float* packed, const float* b_conv, const float* b_ih, const float* hh, const float* h_prev,
                       float* h_new, float* P, int B, int H, int W, void* stream, const float* xmax = nullptr, bool cb8 = false, float* Q = nullptr,
                       float* E = nullptr) {
    MRX_REQUIRE(x && packed && hh && h_new, MRX_EINVAL, "mrx_rim_layer2_sb: null pointer");
    MRX_REQUIRE(B >= 0 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_rim_layer2_sb: bad dims");
    if (B == 0) return MRX_OK;
    L2sbArgs a;
    a.x = x, a.packed = reinterpret_cast<const u32x4*>(packed), a.b_conv = b_conv, a.b_ih = b_ih, a.hh = hh, a.hprev = h_prev, a.hnew = h_new;
    a.B = B, a.H = H, a.W = W, a.tiles_x = mrx_cdiv(W, S2_TW), a.ntiles = a.tiles_x * mrx_cdiv(H, S2_TH);
    a.P = P, a.act = MRX_ACT_NONE, a.slope = 0.f;
    a.xmax = reinterpret_cast<const unsigned*>(xmax);
    a.xmax_out = nullptr;
    if (Q) {                                            // the row-pre-summed tap planes: the FAST form only
        MRX_REQUIRE(E && !P && xmax && cb8, MRX_EINVAL, "mrx_rim_layer2_f16_cb8_q: bad arguments");
        MRX_REQUIRE((long long)H * W * (S2_F * 4) < (1ll << 31), MRX_EUNSUP, "mrx_rim_layer2_f16_cb8_q: %d x %d: a sample's state exceeds 32-bit byte offsets", H, W);
        a.Q = Q, a.E = E;
        return l2sb_launch_t<2, true, false, true, 0, true, false, false, true>(a, (hipStream_t)stream);
    }
#ifdef MRX_PROBE
    if (xmax && cb8 && getenv("MRX_L2_ABL")) {
        switch (atoi(getenv("MRX_L2_ABL"))) {
#define L2_ABL_CASE(N) case N: return l2sb_launch_t<2, true, false, true, N, true>(a, (hipStream_t)stream);
            L2_ABL_CASE(1) L2_ABL_CASE(3) L2_ABL_CASE(7) L2_ABL_CASE(8) L2_ABL_CASE(32) L2_ABL_CASE(64)
            L2_ABL_CASE(128) L2_ABL_CASE(256) L2_ABL_CASE(384) L2_ABL_CASE(512) L2_ABL_CASE(896)
#undef L2_ABL_CASE
            default: break;
        }
    }
    if (xmax && !cb8 && getenv("MRX_L2_ABL")) {
        switch (atoi(getenv("MRX_L2_ABL"))) {
#define L2_ABL_CASE(N) case N: return l2sb_launch_t<2, true, false, true, N>(a, (hipStream_t)stream);
            L2_ABL_CASE(1) L2_ABL_CASE(2) L2_ABL_CASE(3) L2_ABL_CASE(4) L2_ABL_CASE(7) L2_ABL_CASE(8)
            L2_ABL_CASE(32) L2_ABL_CASE(40) L2_ABL_CASE(64) L2_ABL_CASE(71)
            L2_ABL_CASE(128) L2_ABL_CASE(256) L2_ABL_CASE(384) L2_ABL_CASE(512) L2_ABL_CASE(896) L2_ABL_CASE(1024)
#undef L2_ABL_CASE
            default: break;
        }
    }
#endif
    // (the FAST form addresses a sample's state and its tap planes with 32-bit byte offsets through buffer descriptors)
    if (xmax && cb8 && (long long)H * W * (S2_F * 4) < (1ll << 31)) return l2sb_launch_t<2, true, false, true, 0, true, false, false, true>(a, (hipStream_t)stream);
    if (xmax && cb8) return l2sb_launch_t<2, true, false, true, 0, true>(a, (hipStream_t)stream);
    MRX_REQUIRE(!cb8, MRX_EUNSUP, "mrx_rim_layer2_f16_cb8: the channel-blocked layout exists for the two-term fp16 form only");
    if (xmax) return l2sb_launch_t<2, true, false, true>(a, (hipStream_t)stream);
    return l2sb_launch_t<2, true, false>(a, (hipStream_t)stream);
}

// y = act(conv3x3(x, dilation 1 | 2, zero | replicate padding) + bias), 64 -> 64 channels, on the bf16 matrix pipe with fp32 results (the
// convolution stage of mrx_rim_layer2_sb on its own: conv_layers.py:121-123 for the 64-channel layers of CascadeNet / VSNet / the Recurrent
// VarNet).  packed = mrx_rim_layer2_sb_pack(w, NULL, NULL).
extern "C" int mrx_conv3x3_sb_supported(int Cin, int Cout, int k, int dil) { return Cin == 64 && Cout == 64 && k == 3 && (dil == 1 || dil == 2); }
extern "C" int mrx_conv3x3_sb(const float* x, const float* packed, const float* bias, float* y, int B, int H, int W, int dil, int pad_mode, int act,
                              float slope, void* stream) {
    MRX_REQUIRE(x && packed && y, MRX_EINVAL, "mrx_conv3x3_sb: null pointer");
    MRX_REQUIRE(B >= 0 && H >= 1 && W >= 1 && (dil == 1 || dil == 2), MRX_EINVAL, "mrx_conv3x3_sb: bad dims");
    MRX_REQUIRE(pad_mode == MRX_PAD_ZERO || pad_mode == MRX_PAD_REPLICATE, MRX_EINVAL, "mrx_conv3x3_sb: bad pad mode %d", pad_mode);
    MRX_REQUIRE(act >= 0 && act <= 2, MRX_EINVAL, "mrx_conv3x3_sb: bad activation %d", act);
    if (B == 0) return MRX_OK;
    L2sbArgs a;
    a.x = x, a.packed = reinterpret_cast<const u32x4*>(packed), a.b_conv = bias, a.b_ih = nullptr, a.hh = nullptr, a.hprev = nullptr, a.hnew = y;
    a.B = B, a.H = H, a.W = W, a.tiles_x = mrx_cdiv(W, S2_TW), a.ntiles = a.tiles_x * mrx_cdiv(H, S2_TH);
    a.P = nullptr, a.act = act, a.slope = slope, a.xmax = nullptr, a.xmax_out = nullptr;
    hipStream_t st = (hipStream_t)stream;
    const bool zp = pad_mode == MRX_PAD_ZERO;
    if (dil == 1) return zp ? l2sb_launch_t<1, false, true>(a, st) : l2sb_launch_t<1, false, false>(a, st);
    return zp ? l2sb_launch_t<2, false, true>(a, st) : l2sb_launch_t<2, false, false>(a, st);
}

// mrx_conv3x3_sb for CHAINS of 64-channel convolutions (CascadeNet, VSNet, the Recurrent VarNet, the gated RIMs: conv_layers.py:121-123): every
// call folds max |y| into *xmax_out (device scalar, zeroed by the caller; may be NULL), and a call that is handed the bound of ITS input
// (xmax_in: the previous call's xmax_out) multiplies two-term fp16 operands (packed_f16 = mrx_rim_layer2_f16_pack(w, NULL, NULL): three term
// products per multiply) instead of three-term bf16 ones (packed_bf16 = mrx_rim_layer2_sb_pack(w, NULL, NULL): six).  Same fp32-level results.
extern "C" int mrx_conv3x3_sb_chain(const float* x, const float* packed_bf16, const float* packed_f16, const float* bias, float* y, const float* xmax_in,
                                    float* xmax_out, int B, int H, int W, int dil, int pad_mode, int act, float slope, void* stream) {
    MRX_REQUIRE(x && y && (xmax_in ? packed_f16 != nullptr : packed_bf16 != nullptr), MRX_EINVAL, "mrx_conv3x3_sb_chain: null pointer");
    MRX_REQUIRE(B >= 0 && H >= 1 && W >= 1 && (dil == 1 || dil == 2), MRX_EINVAL, "mrx_conv3x3_sb_chain: bad dims");
    MRX_REQUIRE(pad_mode == MRX_PAD_ZERO || pad_mode == MRX_PAD_REPLICATE, MRX_EINVAL, "mrx_conv3x3_sb_chain: bad pad mode %d", pad_mode);
    MRX_REQUIRE(act >= 0 && act <= 2, MRX_EINVAL, "mrx_conv3x3_sb_chain: bad activation %d", act);
    MRX_REQUIRE(!xmax_in || mrx_arith() == MRX_ARITH_F16X2, MRX_EUNSUP, "mrx_conv3x3_sb_chain: the two-term fp16 form is off (MRIDC_AMD_ARITH)");
    if (B == 0) return MRX_OK;
    MRX_CHECK_BOUND("mrx_conv3x3_sb_chain", x, (long long)B * 64 * H * W, xmax_in, stream);
    L2sbArgs a;
    a.x = x, a.packed = reinterpret_cast<const u32x4*>(xmax_in ? packed_f16 : packed_bf16), a.b_conv = bias, a.b_ih = nullptr, a.hh = nullptr;
    a.hprev = nullptr, a.hnew = y;
    a.B = B, a.H = H, a.W = W, a.tiles_x = mrx_cdiv(W, S2_TW), a.ntiles = a.tiles_x * mrx_cdiv(H, S2_TH);
    a.P = nullptr, a.act = act, a.slope = slope, a.xmax = reinterpret_cast<const unsigned*>(xmax_in), a.xmax_out = reinterpret_cast<unsigned*>(xmax_out);
    hipStream_t st = (hipStream_t)stream;
    const bool zp = pad_mode == MRX_PAD_ZERO;
    if (xmax_in) {
        if (dil == 1) return zp ? l2sb_launch_t<1, false, true, true>(a, st) : l2sb_launch_t<1, false, false, true>(a, st);
        return zp ? l2sb_launch_t<2, false, true, true>(a, st) : l2sb_launch_t<2, false, false, true>(a, st);
    }
    if (dil == 1) return zp ? l2sb_launch_t<1, false, true>(a, st) : l2sb_launch_t<1, false, false>(a, st);
    return zp ? l2sb_launch_t<2, false, true>(a, st) : l2sb_launch_t<2, false, false>(a, st);
}

extern "C" int mrx_rim_layer2_sb(const float* x, const float* packed, const float* b_conv, const float* b_ih, const float* hh,
                                 const float* h_prev, float* h_new, int B, int H, int W, void* stream) {
    return l2sb_launch(x, packed, b_conv, b_ih, hh, h_prev, h_new, nullptr, B, H, W, stream);
}

// eta_out[b][y][x][co] = eta[b][y][x][co] + bias[co] + sum_tap P[b][tap * 2 + co][clamp(y + dy)][clamp(x + dx)]: what is left of the final
// 3x3 convolution (replicate padding, conv_layers.py:72-76; rim_block.py:240-246) once its channel contraction has been done per pixel
__global__ __launch_bounds__(256) void k_l2sb_gather(const float* __restrict__ P, const float* __restrict__ bias, const float* __restrict__ eta,
                                                     float* __restrict__ out, int H, int W) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
    if (x >= W || y >= H) return;
    const long long plane = (long long)H * W;
    const float* pb = P + (long long)b * 18 * plane;
    float s0 = bias ? bias[0] : 0.f, s1 = bias ? bias[1] : 0.f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        int yy = y + dy - 1;
        yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            int xx = x + dx - 1;
            xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
            const long long o = (long long)yy * W + xx;
            s0 += pb[(long long)((dy * 3 + dx) * 2) * plane + o];
            s1 += pb[(long long)((dy * 3 + dx) * 2 + 1) * plane + o];
        }
    }
    const long long e = ((long long)b * plane + (long long)y * W + x) * 2;
    float2 v = eta ? *reinterpret_cast<const float2*>(eta + e) : make_float2(0.f, 0.f);
    v.x += s0, v.y += s1;
    *reinterpret_cast<float2*>(out + e) = v;
}

// The second layer of a RIM step (rim_block.py:233-238) that also leaves the final convolution's per-pixel tap products in `taps`
// ([B][18][H][W]: taps[b][tap * 2 + co] = sum_c w_final[co][c][tap] * h_new[b][c]); packed must hold w_final (mrx_rim_layer2_sb_pack).
extern "C" int mrx_rim_layer2_sb_taps(const float* x, const float* packed, const float* b_conv, const float* b_ih, const float* hh,
                                      const float* h_prev, float* h_new, float* taps, int B, int H, int W, void* stream) {
    MRX_REQUIRE(taps, MRX_EINVAL, "mrx_rim_layer2_sb_taps: null pointer");
    return l2sb_launch(x, packed, b_conv, b_ih, hh, h_prev, h_new, taps, B, H, W, stream);
}

// eta_out [B,H,W,2] = eta + permute(conv3x3_reppad(h_new, w_final) + b_final) from the tap products (rim_block.py:240-246)
// ---- the same layer with the convolution's operands as TWO fp16 terms (mrx_rim_layer2_f16_*) --------------------------------------------
// x = (h1 + h2) 2^-kx with h1 = fp16(x 2^kx), h2 = fp16(x 2^kx - h1) (11 + 11 significand bits), the weights likewise with 2^kw, and the three
// term products of weight >= 2^-11 on v_mfma_f32_32x32x16_f16 with fp32 accumulation: half the MFMAs of the three-term bf16 form of the
// convolution (432 instead of 864 per wave and tile), error per product <= ~3 x 2^-22.  fp16 has a narrow exponent range, so the operands are
// scaled by exact powers of two: kw from max |w| at pack time, kx per launch from `xmax`, a device float holding an upper bound of max |x|
// that the producer of x maintains (mrx_rim_layer_indrnn_packed_xmax / _llg_xmax: atomic max over its outputs).  A stale, larger bound is fine
// (it costs nothing until it is 2^16 times too large); x must be non-negative-or-not, finite, and |x| <= *xmax.  The 1x1 and tap stages keep
// the three-term bf16 form (their inputs are made inside the kernel: no bound is known for them).
extern "C" int64_t mrx_rim_layer2_f16_pack_floats(void) { return (int64_t)S2F_PACK_U4 * 4; }
extern "C" int mrx_rim_layer2_f16_pack(const float* w_conv, const float* w_ih, const float* w_final, float* packed, void* stream) {
    MRX_REQUIRE(w_conv && packed, MRX_EINVAL, "mrx_rim_layer2_f16_pack: null pointer");
    hipLaunchKernelGGL(k_l2f16_wscale, dim3(1), dim3(256), 0, (hipStream_t)stream, w_conv, w_ih, w_final, reinterpret_cast<u32x4*>(packed));
    hipLaunchKernelGGL(k_l2f16_pack, dim3((S2F_PACK_U4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_conv, w_ih, w_final,
                       reinterpret_cast<u32x4*>(packed));
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
// taps may be NULL (no final-convolution tap products)
extern "C" int mrx_rim_layer2_f16(const float* x, const float* packed, const float* b_conv, const float* b_ih, const float* hh,
                                  const float* h_prev, float* h_new, float* taps, const float* xmax, int B, int H, int W, void* stream) {
    MRX_REQUIRE(xmax, MRX_EINVAL, "mrx_rim_layer2_f16: null pointer");
    if (x && B > 0 && H > 0 && W > 0) MRX_CHECK_BOUND("mrx_rim_layer2_f16", x, (long long)B * 64 * H * W, xmax, stream);
    return l2sb_launch(x, packed, b_conv, b_ih, hh, h_prev, h_new, taps, B, H, W, stream, xmax);
}

// mrx_rim_layer2_f16 on channel-blocked tensors: x, h_prev, h_new are [B][8][H][W][8] (mrx_cb8_convert); taps stays [B][18][H][W]
extern "C" int mrx_rim_layer2_f16_cb8(const float* x, const float* packed, const float* b_conv, const float* b_ih, const float* hh,
                                      const float* h_prev, float* h_new, float* taps, const float* xmax, int B, int H, int W, void* stream) {
    MRX_REQUIRE(xmax, MRX_EINVAL, "mrx_rim_layer2_f16_cb8: null pointer");
    if (x && B > 0 && H > 0 && W > 0) MRX_CHECK_BOUND("mrx_rim_layer2_f16_cb8", x, (long long)B * 64 * H * W, xmax, stream);
    return l2sb_launch(x, packed, b_conv, b_ih, hh, h_prev, h_new, taps, B, H, W, stream, xmax, true);
}

// NCHW <-> channel-blocked: one thread per (pixel, block of 8 channels); to_cb8 = 1: y[b][q][p][j] = x[b][8 q + j][p], else the inverse
__global__ void k_cb8_convert(const float* __restrict__ x, float* __restrict__ y, long long plane, int nblk, int to_cb8) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int q = blockIdx.y, b = blockIdx.z;
    if (p >= plane) return;
    const long long base = ((long long)b * nblk + q) * 8 * plane;
    if (to_cb8) {
        float4 lo, hi;
        lo.x = x[base + p], lo.y = x[base + plane + p], lo.z = x[base + 2 * plane + p], lo.w = x[base + 3 * plane + p];
        hi.x = x[base + 4 * plane + p], hi.y = x[base + 5 * plane + p], hi.z = x[base + 6 * plane + p], hi.w = x[base + 7 * plane + p];
        float4* d = reinterpret_cast<float4*>(y + base + p * 8);
        d[0] = lo, d[1] = hi;
    } else {
        const float4* sp = reinterpret_cast<const float4*>(x + base + p * 8);
        const float4 lo = sp[0], hi = sp[1];
        y[base + p] = lo.x, y[base + plane + p] = lo.y, y[base + 2 * plane + p] = lo.z, y[base + 3 * plane + p] = lo.w;
        y[base + 4 * plane + p] = hi.x, y[base + 5 * plane + p] = hi.y, y[base + 6 * plane + p] = hi.z, y[base + 7 * plane + p] = hi.w;
    }
}
extern "C" int mrx_cb8_convert(const float* x, float* y, int B, int C, int H, int W, int to_cb8, void* stream) {
    MRX_REQUIRE(x && y && x != y, MRX_EINVAL, "mrx_cb8_convert: null or aliased pointers");
    MRX_REQUIRE(B >= 0 && C >= 8 && C % 8 == 0 && H >= 1 && W >= 1 && B <= 65535, MRX_EINVAL, "mrx_cb8_convert: bad dims");
    if (B == 0) return MRX_OK;
    const long long plane = (long long)H * W;
    hipLaunchKernelGGL(k_cb8_convert, dim3((unsigned)((plane + 255) / 256), C / 8, B), dim3(256), 0, (hipStream_t)stream, x, y, plane, C / 8, to_cb8 ? 1 : 0);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

extern "C" int mrx_rim_final_gather(const float* taps, const float* b_final, const float* eta, float* eta_out, int B, int H, int W, void* stream) {
    MRX_REQUIRE(taps && eta_out, MRX_EINVAL, "mrx_rim_final_gather: null pointer");
    MRX_REQUIRE(B >= 0 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_rim_final_gather: bad dims");
    if (B == 0) return MRX_OK;
    hipLaunchKernelGGL(k_l2sb_gather, dim3(mrx_cdiv(W, 64), mrx_cdiv(H, 4), B), dim3(256), 0, (hipStream_t)stream, taps, b_final, eta, eta_out, H, W);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// The same gather from the row-pre-summed tap planes of mrx_rim_layer2_f16_cb8_q: eta_out = eta + bias + sum_dy Q[dy][co][clamp(y + dy - 1)][x] + the term the
// neighbouring 32-pixel tile owes column 31 (E side 0 of the tile to the right) or column 0 (E side 1 of the tile to the left).  8 + 2 instead of 18 + 2 loads per
// pixel.  The order of the additions is fixed and shared with the gather inside mrx_llg372_gather_q (bit-identical to it; the nine-tap form differs in rounding).
__device__ __forceinline__ void l2sb_gather_q_px(const float* __restrict__ qb, const float* __restrict__ ebp, long long plane, int tiles_x, int H, int W, int y, int x,
                                                 float b0, float b1, float& s0, float& s1) {
    float q[3][2], e[3][2];
    const int xt = x >> 5, xl = x & 31;
    const bool fromR = xl == 31 && x + 1 < W, fromL = xl == 0 && x > 0;
    // E[y][tile][16]: side 0 (dx = 2 products of the tile's column 0): dy 0 at [4, 5], dy 1 at [0, 1], dy 2 at [2, 3]; side 1 (dx = 0 products of column 31): dy 0 at
    // [8, 9], dy 1 at [12, 13], dy 2 at [14, 15]
    const int et = fromR ? xt + 1 : (fromL ? xt - 1 : xt);
    const int o0 = fromR ? 4 : 8, o1 = fromR ? 0 : 12, o2 = fromR ? 2 : 14;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        int yy = y + dy - 1;
        yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
        const float2 qq = *reinterpret_cast<const float2*>(qb + ((long long)dy * plane + (long long)yy * W + x) * 2);
        q[dy][0] = qq.x, q[dy][1] = qq.y;
        const float2 ee = *reinterpret_cast<const float2*>(ebp + ((long long)yy * tiles_x + et) * 16 + (dy == 0 ? o0 : (dy == 1 ? o1 : o2)));
        e[dy][0] = (fromR || fromL) ? ee.x : 0.f, e[dy][1] = (fromR || fromL) ? ee.y : 0.f;
    }
    s0 = b0, s1 = b1;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) s0 += q[dy][0], s1 += q[dy][1];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) s0 += e[dy][0], s1 += e[dy][1];
}
__global__ __launch_bounds__(256) void k_l2sb_gather_q(const float* __restrict__ Q, const float* __restrict__ E, const float* __restrict__ bias,
                                                       const float* __restrict__ eta, float* __restrict__ out, int H, int W, int tiles_x) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
    if (x >= W || y >= H) return;
    const long long plane = (long long)H * W;
    float s0, s1;
    l2sb_gather_q_px(Q + (long long)b * 6 * plane, E + (long long)b * 16 * H * tiles_x, plane, tiles_x, H, W, y, x, bias ? bias[0] : 0.f, bias ? bias[1] : 0.f, s0, s1);
    const long long e = ((long long)b * plane + (long long)y * W + x) * 2;
    float2 v = eta ? *reinterpret_cast<const float2*>(eta + e) : make_float2(0.f, 0.f);
    v.x += s0, v.y += s1;
    *reinterpret_cast<float2*>(out + e) = v;
}
extern "C" int64_t mrx_rim_taps_q_edge_floats(int B, int H, int W) { return B < 0 || H < 1 || W < 1 ? -1 : (int64_t)B * 16 * H * mrx_cdiv(W, S2_TW); }
// mrx_rim_layer2_f16_cb8 whose tap products leave pre-summed along x: taps_q [B][3][H][W][2] (kernel row dy, pair (co 0, co 1)) and edges [mrx_rim_taps_q_edge_floats] -- consumed by
// mrx_rim_final_gather_q and mrx_llg372_gather_q.  A sample's state must fit 32-bit byte offsets (MRX_EUNSUP otherwise: use the 18-plane form).
extern "C" int mrx_rim_layer2_f16_cb8_q(const float* x, const float* packed, const float* b_conv, const float* b_ih, const float* hh, const float* h_prev,
                                        float* h_new, float* taps_q, float* edges, const float* xmax, int B, int H, int W, void* stream) {
    MRX_REQUIRE(xmax && taps_q && edges, MRX_EINVAL, "mrx_rim_layer2_f16_cb8_q: null pointer");
    if (x && B > 0 && H > 0 && W > 0) MRX_CHECK_BOUND("mrx_rim_layer2_f16_cb8_q", x, (long long)B * 64 * H * W, xmax, stream);
    return l2sb_launch(x, packed, b_conv, b_ih, hh, h_prev, h_new, nullptr, B, H, W, stream, xmax, true, taps_q, edges);
}
extern "C" int mrx_rim_final_gather_q(const float* taps_q, const float* edges, const float* b_final, const float* eta, float* eta_out, int B, int H, int W,
                                      void* stream) {
    MRX_REQUIRE(taps_q && edges && eta_out, MRX_EINVAL, "mrx_rim_final_gather_q: null pointer");
    MRX_REQUIRE(B >= 0 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_rim_final_gather_q: bad dims");
    if (B == 0) return MRX_OK;
    hipLaunchKernelGGL(k_l2sb_gather_q, dim3(mrx_cdiv(W, 64), mrx_cdiv(H, 4), B), dim3(256), 0, (hipStream_t)stream, taps_q, edges, b_final, eta, eta_out, H, W,
                       mrx_cdiv(W, S2_TW));
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- a 3x3 convolution into a few channels as a per-pixel channel contraction + a nine-tap gather ------------------------------------------
// out[b][co][y][x] = bias[co] + sum_tap taps[b][tap * Cout + co][y + dy][x + dx], Cout <= 4, taps [B][Ct >= 9 Cout][H][W]; replicate padding = clamped coordinates, zero
// padding = taps outside the image dropped.  The contraction (a 1x1 convolution Cin -> 9 Cout with weights w[co][c][tap] -> [tap * Cout + co][c])
// runs on the matrix cores (mrx_conv2d); the direct form on the vector ALUs costs 18 Cout FMAs per (pixel, channel).
// CO4: Cout == 4 -- all 36 loads issued unconditionally from clamped coordinates (a dropped tap is multiplied by zero), no branch per load
template <bool CO4>
__global__ __launch_bounds__(256) void k_taps_gather(const float* __restrict__ taps, const float* __restrict__ bias, float* __restrict__ out, int Ct,
                                                     int Cout, int H, int W, int zero_pad) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
    if (x >= W || y >= H) return;
    const long long plane = (long long)H * W;
    const float* pb = taps + (long long)b * Ct * plane;
    float s[4];
#pragma unroll
    for (int co = 0; co < 4; ++co) s[co] = (bias && co < Cout) ? bias[co] : 0.f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        int yy = y + dy - 1;
        const bool oky = yy >= 0 && yy < H;
        yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            int xx = x + dx - 1;
            const bool ok = oky && xx >= 0 && xx < W;
            xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
            const float* p = pb + (long long)((dy * 3 + dx) * Cout) * plane + (long long)yy * W + xx;
            if constexpr (CO4) {
                const float keep = (zero_pad && !ok) ? 0.f : 1.f;
#pragma unroll
                for (int co = 0; co < 4; ++co) s[co] += keep * p[(long long)co * plane];
            } else {
                if (zero_pad && !ok) continue;
#pragma unroll
                for (int co = 0; co < 4; ++co)
                    if (co < Cout) s[co] += p[(long long)co * plane];
            }
        }
    }
#pragma unroll
    for (int co = 0; co < 4; ++co)
        if (co < Cout) out[((long long)b * Cout + co) * plane + (long long)y * W + x] = s[co];
}
extern "C" int mrx_taps_gather(const float* taps, const float* bias, float* out, int B, int Ct, int Cout, int H, int W, int pad_mode, void* stream) {
    MRX_REQUIRE(taps && out && B >= 0 && Cout >= 1 && Cout <= 4 && Ct >= 9 * Cout && H >= 1 && W >= 1, MRX_EINVAL, "mrx_taps_gather: bad argument");
    MRX_REQUIRE(pad_mode == MRX_PAD_ZERO || pad_mode == MRX_PAD_REPLICATE, MRX_EINVAL, "mrx_taps_gather: bad pad mode %d", pad_mode);
    if (B == 0) return MRX_OK;
    if (Cout == 4)
        hipLaunchKernelGGL(k_taps_gather<true>, dim3(mrx_cdiv(W, 64), mrx_cdiv(H, 4), B), dim3(256), 0, (hipStream_t)stream, taps, bias, out, Ct, Cout, H,
                           W, pad_mode == MRX_PAD_ZERO ? 1 : 0);
    else
        hipLaunchKernelGGL(k_taps_gather<false>, dim3(mrx_cdiv(W, 64), mrx_cdiv(H, 4), B), dim3(256), 0, (hipStream_t)stream, taps, bias, out, Ct, Cout, H,
                           W, pad_mode == MRX_PAD_ZERO ? 1 : 0);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
