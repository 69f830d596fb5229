// conv_sbs.hip -- convolutions of FEW input channels (Cin <= 8; 3x3 or 5x5, dilation 1) into up to 128 channels with fp32 results on the bf16
// matrix pipe: the first layers of the cascades (qRIM 5x5 8 -> 128, qrim_block.py:226-236 via conv_layers.py:121-123; CascadeNet / VSNet
// 3x3 2 -> 64), which ran on the generic fp32-MFMA kernel (one MFMA per two input channels of ONE tap: 77 us for the qRIM layer at 256 x 256).
// Every fp32 operand is the exact sum of three bf16 terms, six term products per multiply (error O(2^-24): rim_layer1_sb.hip); with eight
// channels per pixel one v_mfma_f32_32x32x16_bf16 consumes two taps.  The formulation of conv_bf16.hip (whole halo'd tile in LDS as
// [pixel][8 channels], one 16-byte LDS read per B operand, A operands straight from L2, wave = one image row x 32 pixels x all couts),
// times three term planes.
#include <cstdlib>

#include "mrx_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CS_NT 512
#define CS_TH 8
#define CS_TW 32

__device__ __forceinline__ unsigned cs_pk(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ void cs_split2(float a, float b, unsigned& p1, unsigned& p2, unsigned& p3) {
    p1 = cs_pk(a, b);
    float ra = a - __uint_as_float(p1 << 16), rb = b - __uint_as_float(p1 & 0xffff0000u);
    p2 = cs_pk(ra, rb);
    ra -= __uint_as_float(p2 << 16);
    rb -= __uint_as_float(p2 & 0xffff0000u);
    p3 = cs_pk(ra, rb);
}

// ---- two-term fp16 form (F16, the default; MRIDC_AMD_ARITH=bf16x3 selects the three bf16 terms): the whole halo'd tile of a workgroup is staged at
// once, so it is scaled by the power of two that puts the TILE's largest |x| into [2^14, 2^15) (one workgroup reduction: every input of the tile's
// outputs is in it), the weights at pack time; three term products per multiply, accumulators scaled back exactly before the bias
// (rim_layer2_sb.hip for the error analysis: ~3 x 2^-22 per product).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void cs_split2h(float a, float b, unsigned& p1, unsigned& p2) {
    const f16x2 h = {(_Float16)a, (_Float16)b};
    const float ra = a - (float)h.x, rb = b - (float)h.y;     // exact
    const f16x2 l = {(_Float16)ra, (_Float16)rb};
    p1 = __builtin_bit_cast(unsigned, h);
    p2 = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ float cs_pow2(int e) {
    e = e < -120 ? -120 : (e > 120 ? 120 : e);
    return __uint_as_float((unsigned)(127 + e) << 23);
}
__device__ __forceinline__ int cs_scale_exp(float m) {     // k with m 2^k in [2^14, 2^15); 0 for zero / non-finite m
    const int ex = (int)((__float_as_uint(m) >> 23) & 0xffu);
    return (ex == 0 || ex == 255) ? 0 : 14 - (ex - 127);
}

struct ConvSbsArgs {
    const float* x;        // [B,Cin,H,W], Cin <= 8
    const u32x4* packed;   // [3 terms][NSTEP][NCT][64 lanes] x 8 bf16
    const float* bias;     // [Cout] or null
    float* out;            // [B,Cout,H,W]
    int B, Cin, Cout, H, W, tiles_x, ntiles, pad_mode, act;
    float slope;
};
__host__ __device__ constexpr int cs_nstep(int K) { return (K * K + 1) / 2; }

// ONE (round 6; F16 only): the reference's `precision: 16` inference arithmetic (torch.autocast(float16) around forward, conv_layers.py:121-123 under it): the first
// fp16 term of each operand only -- x and W rounded to fp16 once (behind the same exact power-of-two scales), one product per multiply, fp32 sums.
template <int K, int NCT, bool F16, bool ONE = false>
__global__ __launch_bounds__(CS_NT, 2) void k_conv_sbs(ConvSbsArgs a) {
    static_assert(F16 || !ONE, "one term: the fp16 form");
    constexpr int PAD = (K - 1) / 2, PH = CS_TH + 2 * PAD, PW = CS_TW + 2 * PAD, NPIX = PH * PW, NG = K * K, NSTEP = cs_nstep(K);
    static_assert(!F16 || NPIX <= CS_NT, "F16: one staging pass (the tile maximum is taken over the values of that pass)");
    __shared__ float wmaxs[CS_NT / 64];
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_cs[];      // [3 terms][NPIX] x 16 B
    u32x4* Xs = reinterpret_cast<u32x4*>(smem_cs);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int tile = (int)mrx_xcd_band(blockIdx.x, a.ntiles);
    const int ty0 = tile / a.tiles_x;
    const int h0 = ty0 * CS_TH, w0 = (tile - ty0 * a.tiles_x) * CS_TW;
    const int b = blockIdx.y;
    const long long plane = (long long)a.H * a.W;
    const float* xb = a.x + (long long)b * a.Cin * plane;

    // ---- stage the halo'd tile: the 8 channels of a pixel, split into their three bf16 terms, one 16-byte LDS write per term ----------------
    constexpr int ITERS = (NPIX + CS_NT - 1) / CS_NT;
    float xv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // F16: this thread's pixel of the tile (threads past the tile hold zeros)
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int e = tid + it * CS_NT;
        if (e < NPIX) {
            const int ty = e / PW, tx = e - ty * PW;
            int gy = h0 + ty - PAD, gx = w0 + tx - PAD;
            bool inb = true;
            if (a.pad_mode == MRX_PAD_REPLICATE) {
                gy = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy);
                gx = gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx);
            } else {
                inb = gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
                gy = inb ? gy : 0;
                gx = inb ? gx : 0;
            }
            const float* src = xb + (long long)gy * a.W + gx;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = src[(long long)(j < a.Cin ? j : 0) * plane];   // every lane loads a valid element (channel clamped)
            // (the opaque uses keep the eight loads where they are: with the zeroing select as their only use the compiler moved each load behind
            // a branch on `j < Cin`, wait included -- eight memory round trips one after the other at the head of every workgroup)
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(v[j]));
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (inb && j < a.Cin) ? v[j] : 0.f;
            if constexpr (F16) {
#pragma unroll
                for (int j = 0; j < 8; ++j) xv[j] = v[j];
            } else {
                unsigned p1[4], p2[4], p3[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) cs_split2(v[2 * k], v[2 * k + 1], p1[k], p2[k], p3[k]);
                Xs[e] = u32x4{p1[0], p1[1], p1[2], p1[3]};
                Xs[NPIX + e] = u32x4{p2[0], p2[1], p2[2], p2[3]};
                Xs[2 * NPIX + e] = u32x4{p3[0], p3[1], p3[2], p3[3]};
            }
        }
    }
    float un = 1.f;
    if constexpr (F16) {
        // the tile's maximum -> one scale for the workgroup
        float m = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(xv[j]));
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (lane == 0) wmaxs[wave] = m;
        __syncthreads();
        m = wmaxs[0];
#pragma unroll
        for (int w = 1; w < CS_NT / 64; ++w) m = fmaxf(m, wmaxs[w]);
        const int kx = cs_scale_exp(m);
        const float sx = cs_pow2(kx);
        un = cs_pow2(-kx) * cs_pow2(-(int)a.packed[5 * NSTEP * NCT * 64][0]);
        if (tid < NPIX) {
            unsigned p1[4], p2[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) cs_split2h(xv[2 * k] * sx, xv[2 * k + 1] * sx, p1[k], p2[k]);
            Xs[tid] = u32x4{p1[0], p1[1], p1[2], p1[3]};
            if constexpr (!ONE) Xs[NPIX + tid] = u32x4{p2[0], p2[1], p2[2], p2[3]};
        }
    }
    __syncthreads();

    // ---- the matrix loop: per step (two taps) three B reads from LDS, per cout block three A reads from L2 and six term products --------------
    f32x16 acc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
    const u32x4* bx = Xs + wave * PW + l31;
    const u32x4* wp = a.packed + lane;
    constexpr int WT = NSTEP * NCT * 64;       // operands per term
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        const int t0 = 2 * s, t1 = (2 * s + 1 < NG) ? 2 * s + 1 : NG - 1;   // the upper half-wave takes the next tap (zero weights past the last)
        const int o0 = (t0 / K) * PW + (t0 % K), o1 = (t1 / K) * PW + (t1 % K);
        const int off = lhi ? o1 : o0;
        if constexpr (F16) {
            const f16x8 b1 = __builtin_bit_cast(f16x8, bx[off]);
            if constexpr (ONE) {
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
                    acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wp[3 * WT + (s * NCT + ct) * 64]), b1, acc[ct], 0, 0, 0);
                continue;
            }
            const f16x8 b2 = __builtin_bit_cast(f16x8, bx[NPIX + off]);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct) {
                const f16x8 a1 = __builtin_bit_cast(f16x8, wp[3 * WT + (s * NCT + ct) * 64]);
                const f16x8 a2 = __builtin_bit_cast(f16x8, wp[4 * WT + (s * NCT + ct) * 64]);
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b1, acc[ct], 0, 0, 0);
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b2, acc[ct], 0, 0, 0);
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[ct], 0, 0, 0);
            }
            continue;
        }
        const bf16x8 b1 = __builtin_bit_cast(bf16x8, bx[off]);
        const bf16x8 b2 = __builtin_bit_cast(bf16x8, bx[NPIX + off]);
        const bf16x8 b3 = __builtin_bit_cast(bf16x8, bx[2 * NPIX + off]);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const bf16x8 a1 = __builtin_bit_cast(bf16x8, wp[(s * NCT + ct) * 64]);
            const bf16x8 a2 = __builtin_bit_cast(bf16x8, wp[WT + (s * NCT + ct) * 64]);
            const bf16x8 a3 = __builtin_bit_cast(bf16x8, wp[2 * WT + (s * NCT + ct) * 64]);
            // the six term pairs of weight >= 2^-16, smallest first
            acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc[ct], 0, 0, 0);
            acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc[ct], 0, 0, 0);
            acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc[ct], 0, 0, 0);
            acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc[ct], 0, 0, 0);
            acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc[ct], 0, 0, 0);
            acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[ct], 0, 0, 0);
        }
    }

    // ---- epilogue: bias, activation; lane = pixel (128-byte rows per wave instruction) ----------------------------------------------------
    // (the arithmetic first, then the stores under ONE predicate per block of values: a branch around every store -- bias? activation? channel
    // in range? -- made seven branches per store, each basic block waiting on its own loads)
    const int oy = h0 + wave, ox = w0 + l31;
    const float neg = a.act == MRX_ACT_RELU ? 0.f : (a.act == MRX_ACT_LEAKY ? a.slope : 1.f);
    const int clast = a.Cout - 1;
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            float v = F16 ? acc[ct][r] * un : acc[ct][r];
            v += a.bias ? a.bias[co < clast ? co : clast] : 0.f;
            acc[ct][r] = v > 0.f ? v : v * neg;
        }
    if (oy < a.H && ox < a.W) {
        float* ob = a.out + (long long)b * a.Cout * plane + (long long)oy * a.W + ox;
        if (a.Cout == NCT * 32) {            // every channel of every block exists (64, 128 ... features)
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) ob[(long long)(ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * plane] = acc[ct][r];
        } else {
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    if (co < a.Cout) ob[(long long)co * plane] = acc[ct][r];
                }
        }
    }
}

// packed[t * WT + (s * NCT + ct) * 64 + lane][j] = term_t( w[cout = 32 ct + lane % 32][channel j][tap = 2 s + lane / 32] )   (0 past Cin / Cout / taps)
__global__ void k_conv_sbs_pack(const float* __restrict__ w, u32x4* __restrict__ out, int Cin, int Cout, int K, int NCT) {
    const int TAPS = K * K, NSTEP = (TAPS + 1) / 2, WT = NSTEP * NCT * 64;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 3 * WT; i += gridDim.x * blockDim.x) {
        const int t = i / WT, r = i - t * WT;
        const int lane = r & 63, ct = (r >> 6) % NCT, s = (r >> 6) / NCT;
        const int tap = 2 * s + (lane >> 5), co = ct * 32 + (lane & 31);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (tap < TAPS && j < Cin && co < Cout) ? w[((long long)co * Cin + j) * TAPS + tap] : 0.f;
        unsigned p[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned p1, p2, p3;
            cs_split2(v[2 * k], v[2 * k + 1], p1, p2, p3);
            p[k] = t == 0 ? p1 : (t == 1 ? p2 : p3);
        }
        out[i] = u32x4{p[0], p[1], p[2], p[3]};
    }
}

// fp16 section of the pack: [3 WT .. 5 WT) two terms of w 2^kw in the same operand order, [5 WT] the header (kw)
__global__ void k_conv_sbs_wscale(const float* __restrict__ w, u32x4* __restrict__ out, int n, int WT) {
    __shared__ float red[256];
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(w[i]));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[5 * WT] = u32x4{(unsigned)cs_scale_exp(red[0]), 0u, 0u, 0u};
}
__global__ void k_conv_sbs_pack_f16(const float* __restrict__ w, u32x4* __restrict__ out, int Cin, int Cout, int K, int NCT) {
    const int TAPS = K * K, NSTEP = (TAPS + 1) / 2, WT = NSTEP * NCT * 64;
    const float sw = cs_pow2((int)out[5 * WT][0]);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 2 * WT; i += gridDim.x * blockDim.x) {
        const int t = i / WT, r = i - t * WT;
        const int lane = r & 63, ct = (r >> 6) % NCT, s = (r >> 6) / NCT;
        const int tap = 2 * s + (lane >> 5), co = ct * 32 + (lane & 31);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (tap < TAPS && j < Cin && co < Cout) ? w[((long long)co * Cin + j) * TAPS + tap] * sw : 0.f;
        unsigned p[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned p1, p2;
            cs_split2h(v[2 * k], v[2 * k + 1], p1, p2);
            p[k] = t == 0 ? p1 : p2;
        }
        out[3 * WT + i] = u32x4{p[0], p[1], p[2], p[3]};
    }
}

static int cs_nct(int Cout) { return Cout <= 32 ? 1 : (Cout <= 64 ? 2 : 4); }
extern "C" int mrx_conv_sbs_supported(int Cin, int Cout, int k, int dil) {
    return (Cin >= 1 && Cin <= 8 && Cout >= 1 && Cout <= 128 && (k == 3 || k == 5) && dil == 1) ? 1 : 0;
}
extern "C" int64_t mrx_conv_sbs_pack_floats(int Cout, int k) {
    if (Cout < 1 || Cout > 128 || (k != 3 && k != 5)) return -1;
    return ((int64_t)5 * cs_nstep(k) * cs_nct(Cout) * 64 + 1) * 4;      // three bf16 terms, two fp16 terms, header
}
extern "C" int mrx_conv_sbs_pack(const float* w, float* packed, int Cin, int Cout, int k, void* stream) {
    MRX_REQUIRE(w && packed, MRX_EINVAL, "mrx_conv_sbs_pack: null pointer");
    MRX_REQUIRE(mrx_conv_sbs_supported(Cin, Cout, k, 1), MRX_EUNSUP, "mrx_conv_sbs_pack: Cin=%d Cout=%d k=%d", Cin, Cout, k);
    const int total = 3 * cs_nstep(k) * cs_nct(Cout) * 64, WT = cs_nstep(k) * cs_nct(Cout) * 64;
    hipLaunchKernelGGL(k_conv_sbs_wscale, dim3(1), dim3(256), 0, (hipStream_t)stream, w, reinterpret_cast<u32x4*>(packed), Cout * Cin * k * k, WT);
    hipLaunchKernelGGL(k_conv_sbs_pack_f16, dim3((2 * WT + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, reinterpret_cast<u32x4*>(packed), Cin, Cout,
                       k, cs_nct(Cout));
    hipLaunchKernelGGL(k_conv_sbs_pack, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, reinterpret_cast<u32x4*>(packed), Cin, Cout, k,
                       cs_nct(Cout));
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

template <int K, int NCT>
static int cs_launch(const ConvSbsArgs& a, hipStream_t st, int one) {
    constexpr size_t lds = (size_t)3 * (CS_TH + K - 1) * (CS_TW + K - 1) * 16;
    static_assert(lds <= 48 * 1024, "fits the default dynamic LDS limit");
    const int f16 = mrx_arith() == MRX_ARITH_F16X2 ? 1 : 0;   // 0: the three-term bf16 form
    if (one) {
        MRX_REQUIRE(f16, MRX_EUNSUP, "mrx_conv_sbs_p16: the fp16 operand form is off (MRIDC_AMD_ARITH)");
        hipLaunchKernelGGL((k_conv_sbs<K, NCT, true, true>), dim3(a.ntiles, a.B), dim3(CS_NT), lds, st, a);
    } else if (f16)
        hipLaunchKernelGGL((k_conv_sbs<K, NCT, true>), dim3(a.ntiles, a.B), dim3(CS_NT), lds, st, a);
    else
        hipLaunchKernelGGL((k_conv_sbs<K, NCT, false>), dim3(a.ntiles, a.B), dim3(CS_NT), lds, st, a);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
template <int K>
static int cs_launch_nct(const ConvSbsArgs& a, hipStream_t st, int one) {
    const int nct = cs_nct(a.Cout);
    return nct == 1 ? cs_launch<K, 1>(a, st, one) : (nct == 2 ? cs_launch<K, 2>(a, st, one) : cs_launch<K, 4>(a, st, one));
}
// y = act(conv_kxk(x, zero | replicate padding) + bias), Cin <= 8 -> Cout <= 128, k = 3 | 5, dilation 1; packed from mrx_conv_sbs_pack
static int conv_sbs_impl(const float* x, const float* packed, const float* bias, float* y, int B, int Cin, int Cout, int H, int W, int k, int pad_mode, int act,
                         float slope, void* stream, int one) {
    MRX_REQUIRE(x && packed && y, MRX_EINVAL, "mrx_conv_sbs: null pointer");
    MRX_REQUIRE(B >= 0 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_conv_sbs: bad dims");
    MRX_REQUIRE(mrx_conv_sbs_supported(Cin, Cout, k, 1), MRX_EUNSUP, "mrx_conv_sbs: Cin=%d Cout=%d k=%d", Cin, Cout, k);
    MRX_REQUIRE(pad_mode == MRX_PAD_ZERO || pad_mode == MRX_PAD_REPLICATE, MRX_EINVAL, "mrx_conv_sbs: bad pad mode %d", pad_mode);
    MRX_REQUIRE(act >= 0 && act <= 2, MRX_EINVAL, "mrx_conv_sbs: bad activation %d", act);
    MRX_REQUIRE(B <= 65535, MRX_EUNSUP, "mrx_conv_sbs: batch %d too large", B);
    if (B == 0) return MRX_OK;
    ConvSbsArgs a;
    a.x = x, a.packed = reinterpret_cast<const u32x4*>(packed), a.bias = bias, a.out = y;
    a.B = B, a.Cin = Cin, a.Cout = Cout, a.H = H, a.W = W, a.tiles_x = mrx_cdiv(W, CS_TW), a.ntiles = a.tiles_x * mrx_cdiv(H, CS_TH);
    a.pad_mode = pad_mode, a.act = act, a.slope = slope;
    return k == 3 ? cs_launch_nct<3>(a, (hipStream_t)stream, one) : cs_launch_nct<5>(a, (hipStream_t)stream, one);
}
extern "C" int mrx_conv_sbs(const float* x, const float* packed, const float* bias, float* y, int B, int Cin, int Cout, int H, int W, int k,
                            int pad_mode, int act, float slope, void* stream) {
    return conv_sbs_impl(x, packed, bias, y, B, Cin, Cout, H, W, k, pad_mode, act, slope, stream, 0);
}
// ... in the reference's `precision: 16` inference arithmetic (base_qcirim_run.yaml:204 ...): x and W rounded to fp16 once, fp32 sums.  MRIDC_AMD_ARITH = f16x2.
extern "C" int mrx_conv_sbs_p16(const float* x, const float* packed, const float* bias, float* y, int B, int Cin, int Cout, int H, int W, int k,
                                int pad_mode, int act, float slope, void* stream) {
    return conv_sbs_impl(x, packed, bias, y, B, Cin, Cout, H, W, k, pad_mode, act, slope, stream, 1);
}
