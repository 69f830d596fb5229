// gated_cell_sb.hip -- ConvGRUCell / ConvMGUCell with 1x1 kernels on 64 features (reference models/rim/rnn_cells.py:112-127, :249-261) with
// fp32 results on the bf16 matrix pipe.  The formulation of gated_cell.hip (a wave owns 32 consecutive pixels, x and h_prev straight from
// HBM into the MFMA B-operand layout, all 2 * GATES weight matrices resident in LDS, ih + hh parts of the plain gates summed in one
// accumulator, the candidate's parts apart, gate math on the accumulators), with every fp32 operand written as the exact sum of three bf16
// terms and six term products per multiply (error O(2^-24): rim_layer1_sb.hip): 288 v_mfma_f32_32x32x16_bf16 of 32 cycles per 32 pixels for
// the GRU instead of 384 v_mfma_f32_32x32x2_f32 of 64 -- and the bf16 MFMA co-issues with the vector ALU, which evaluates the gates.
#include <cstdint>
#include <cstdlib>

#include "mrx_common.h"
#include "gated_cell_sb.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define GS_NT 512
#define GS_F 64

__device__ __forceinline__ unsigned gs_pk(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ void gs_split2(float a, float b, unsigned& p1, unsigned& p2, unsigned& p3) {
    p1 = gs_pk(a, b);
    float ra = a - __uint_as_float(p1 << 16), rb = b - __uint_as_float(p1 & 0xffff0000u);
    p2 = gs_pk(ra, rb);
    ra -= __uint_as_float(p2 << 16);
    rb -= __uint_as_float(p2 & 0xffff0000u);
    p3 = gs_pk(ra, rb);
}
__device__ __forceinline__ float gs_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float gs_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x)); }

// packed[(((mat * 2 + mb) * 4 + t) * 3 + term) * 64 + lane][j] = term( W_mat[mb * 32 + lane % 32][16 t + 8 (lane / 32) + j] ),
// mat = gate (ih) | GATES + gate (hh)
__global__ void k_gated_pack_sb(const float* __restrict__ w_ih, const float* __restrict__ w_hh, u32x4* __restrict__ out, int gates) {
    const int total = 2 * gates * 2 * 4 * 3 * 64;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int lane = i & 63;
        int r = i >> 6;
        const int term = r % 3;
        r /= 3;
        const int t = r & 3, mb = (r >> 2) & 1, mat = r >> 3;
        const float* w = mat < gates ? w_ih : w_hh;
        const int g = mat < gates ? mat : mat - gates;
        const int row = g * GS_F + mb * 32 + (lane & 31), col0 = 16 * t + 8 * (lane >> 5);
        unsigned p[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned p1, p2, p3;
            gs_split2(w[(long long)row * GS_F + col0 + 2 * k], w[(long long)row * GS_F + col0 + 2 * k + 1], p1, p2, p3);
            p[k] = term == 0 ? p1 : (term == 1 ? p2 : p3);
        }
        out[i] = u32x4{p[0], p[1], p[2], p[3]};
    }
}
// ---- two-term fp16 form (F16): every contraction of the cell runs over a pixel's channels only, so x and h_prev are scaled per PIXEL by the power
// of two that puts the pixel's largest |x|, |h| into [2^14, 2^15) (its 64 + 64 channels sit in this lane and lane ^ 32), all matrices by one power
// of two at pack time, and a lane's accumulators -- its own pixel's -- are scaled back exactly before the biases.  Three term products per
// multiply instead of six, error per product ~3 x 2^-22 (rim_layer2_sb.hip).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void gs_split2h(float a, float b, unsigned& p1, unsigned& p2) {
    const f16x2 h = {(_Float16)a, (_Float16)b};
    const float ra = a - (float)h.x, rb = b - (float)h.y;     // exact
    const f16x2 l = {(_Float16)ra, (_Float16)rb};
    p1 = __builtin_bit_cast(unsigned, h);
    p2 = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ float gs_pow2(int e) {
    e = e < -120 ? -120 : (e > 120 ? 120 : e);
    return __uint_as_float((unsigned)(127 + e) << 23);
}
__device__ __forceinline__ int gs_scale_exp(float m) {     // k with m 2^k in [2^14, 2^15); 0 for zero / non-finite m
    const int ex = (int)((__float_as_uint(m) >> 23) & 0xffu);
    return (ex == 0 || ex == 255) ? 0 : 14 - (ex - 127);
}
// fp16 section of the pack: [3 terms section][(((mat * 2 + mb) * 4 + t) * 2 + term) * 64 + lane][header]
__global__ void k_gated_wscale(const float* __restrict__ w_ih, const float* __restrict__ w_hh, u32x4* __restrict__ out, int gates) {
    __shared__ float red[256];
    const int n = gates * GS_F * GS_F;
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, fmaxf(fabsf(w_ih[i]), fabsf(w_hh[i])));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[2 * gates * 2 * 4 * 5 * 64] = u32x4{(unsigned)gs_scale_exp(red[0]), 0u, 0u, 0u};
}
__global__ void k_gated_pack_f16(const float* __restrict__ w_ih, const float* __restrict__ w_hh, u32x4* __restrict__ out, int gates) {
    const int total = 2 * gates * 2 * 4 * 2 * 64, off = 2 * gates * 2 * 4 * 3 * 64;
    const float sw = gs_pow2((int)out[2 * gates * 2 * 4 * 5 * 64][0]);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int lane = i & 63;
        int r = i >> 6;
        const int term = r & 1;
        r >>= 1;
        const int t = r & 3, mb = (r >> 2) & 1, mat = r >> 3;
        const float* w = mat < gates ? w_ih : w_hh;
        const int g = mat < gates ? mat : mat - gates;
        const int row = g * GS_F + mb * 32 + (lane & 31), col0 = 16 * t + 8 * (lane >> 5);
        unsigned p[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned p1, p2;
            gs_split2h(w[(long long)row * GS_F + col0 + 2 * k] * sw, w[(long long)row * GS_F + col0 + 2 * k + 1] * sw, p1, p2);
            p[k] = term == 0 ? p1 : p2;
        }
        out[off + i] = u32x4{p[0], p[1], p[2], p[3]};
    }
}
int mrx_gated_sb_pack(const float* w_ih, const float* w_hh, float* packed, int gates, hipStream_t st) {
    const int total = 2 * gates * 2 * 4 * 3 * 64;
    hipLaunchKernelGGL(k_gated_wscale, dim3(1), dim3(256), 0, st, w_ih, w_hh, reinterpret_cast<u32x4*>(packed), gates);
    hipLaunchKernelGGL(k_gated_pack_f16, dim3((2 * gates * 2 * 4 * 2 * 64 + 255) / 256), dim3(256), 0, st, w_ih, w_hh, reinterpret_cast<u32x4*>(packed), gates);
    hipLaunchKernelGGL(k_gated_pack_sb, dim3((total + 255) / 256), dim3(256), 0, st, w_ih, w_hh, reinterpret_cast<u32x4*>(packed), gates);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

template <int GATES, bool F16>  // 3 = GRU, 2 = MGU
__global__ __launch_bounds__(GS_NT, 1) void k_gated_cell_sb(MrxGatedSbArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_gs[];
    constexpr int NMAT = 2 * GATES, NTM = F16 ? 2 : 3, NW = NMAT * 2 * 4 * NTM * 64;       // 16-byte A operands
    u32x4* Wl = reinterpret_cast<u32x4*>(smem_gs);
    float* Bs = reinterpret_cast<float*>(smem_gs + (size_t)NW * 16);   // ih bias [GATES][64]
    const int tid = threadIdx.x;
    float unw = 1.f;
    {
        const u32x4* src = reinterpret_cast<const u32x4*>(a.packed) + (F16 ? NMAT * 2 * 4 * 3 * 64 : 0);
        for (int i = tid; i < NW; i += GS_NT) Wl[i] = src[i];
        if (tid < GATES * GS_F) Bs[tid] = a.b_ih ? a.b_ih[tid] : 0.f;
        if constexpr (F16) unw = gs_pow2(-(int)reinterpret_cast<const u32x4*>(a.packed)[NMAT * 2 * 4 * 5 * 64][0]);
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const u32x4* wl = Wl + lane;

    const long long stride = (long long)gridDim.x * (GS_NT / 64);
    const unsigned P32 = (unsigned)a.P;
    // operand registers: value (t, j) of a lane is channel 16 t + 8 lhi + j of its pixel (the k order of the 32x32x16 B operand)
    float xg[4][8], hg[4][8];
    const float* hb = nullptr;
    long long base = 0;
    unsigned pxo = 0;
    bool valid = false;
    int lhi = 0;
    float vmax = 0.f;                              // maximum |output| of this lane (a.xmax)
    auto load = [&](long long sg) {
        int l31 = lane & 31;
        lhi = lane >> 5;
        asm volatile("" : "+v"(l31), "+v"(lhi));      // (keeps the 64 channel offsets from being hoisted out of the segment loop and spilled)
        const long long b = sg / a.nsegb;
        const long long px = (sg - b * a.nsegb) * 32 + l31;
        valid = px < a.P;
        base = __builtin_amdgcn_readfirstlane((int)b) * (long long)GS_F * a.P;
        pxo = valid ? (unsigned)px : 0u;               // lanes past the end read pixel 0 and store nothing
        const float* xb = a.x + base;
        hb = a.h ? a.h + base : nullptr;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) xg[t][j] = xb[(unsigned)(16 * t + 8 * lhi + j) * P32 + pxo];
    };
    auto load_h = [&]() {
        if (hb) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 8; ++j) hg[t][j] = hb[(unsigned)(16 * t + 8 * lhi + j) * P32 + pxo];
        }
    };
    // one contraction step (16 channels) of `nmat` matrices starting at mat0 over the operand v[8]; dst(g) = accumulator of gate g
    long long sg = (long long)blockIdx.x * (GS_NT / 64) + wave;
    if (sg < a.nseg) load(sg);
    while (sg < a.nseg) {
        if constexpr (!F16) load_h();                      // (F16: requested after the x part is issued -- 32 registers fewer alive through it)
        // accumulator d: 0 .. GATES-2 = gates whose ih and hh parts add up; GATES-1 = candidate ih part; GATES = candidate hh part
        f32x16 acc[GATES + 1][2];
#pragma unroll
        for (int d = 0; d < GATES + 1; ++d)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[d][ct][r] = (d < GATES && !F16) ? Bs[d * GS_F + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi] : 0.f;

        float spx = 1.f;                                   // F16: this pixel's operand scale 2^kp
        int kp = 0;
        if constexpr (F16) {
            float m = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(xg[t][j]));
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            kp = gs_scale_exp(m);
            spx = gs_pow2(kp);
        }
#pragma unroll
        for (int part = 0; part < 2; ++part) {             // 0: ih matrices over x, 1: hh matrices over h_prev
            if (part == 1 && !hb) break;                  // (no previous state: the hh parts are zero)
            if constexpr (F16) {
                if (part == 1) {
                    load_h();
                    // h_prev's own maximum: if it needs a smaller scale than x did, the accumulators (the x parts so far) move to it -- exactly
                    float m = 0.f;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(hg[t][j]));
                    m = fmaxf(m, __shfl_xor(m, 32, 64));
                    const int kh = gs_scale_exp(m);
                    const bool lower = m > 0.f && kh < kp;
                    const float f = lower ? gs_pow2(kh - kp) : 1.f;
                    kp = lower ? kh : kp;
                    spx = gs_pow2(kp);
#pragma unroll
                    for (int d = 0; d < GATES; ++d)
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[d][ct][r] *= f;
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if constexpr (F16) {
                    unsigned p1[4], p2[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (part == 0)
                            gs_split2h(xg[t][2 * k] * spx, xg[t][2 * k + 1] * spx, p1[k], p2[k]);
                        else
                            gs_split2h(hg[t][2 * k] * spx, hg[t][2 * k + 1] * spx, p1[k], p2[k]);
                    }
                    const f16x8 b1 = __builtin_bit_cast(f16x8, (u32x4{p1[0], p1[1], p1[2], p1[3]}));
                    const f16x8 b2 = __builtin_bit_cast(f16x8, (u32x4{p2[0], p2[1], p2[2], p2[3]}));
#pragma unroll
                    for (int g = 0; g < GATES; ++g) {
                        const int mat = part * GATES + g;
                        const int d = (part == 1 && g == GATES - 1) ? GATES : g;
#pragma unroll
                        for (int mb = 0; mb < 2; ++mb) {
                            const u32x4* q = wl + (((mat * 2 + mb) * 4 + t) * 2) * 64;
                            const f16x8 a1 = __builtin_bit_cast(f16x8, q[0]);
                            const f16x8 a2 = __builtin_bit_cast(f16x8, q[64]);
                            acc[d][mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b1, acc[d][mb], 0, 0, 0);
                            acc[d][mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b2, acc[d][mb], 0, 0, 0);
                            acc[d][mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[d][mb], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);   // (keeps the next matrices' operand reads from being hoisted over this one: spills)
                    }
                    continue;
                }
                unsigned p1[4], p2[4], p3[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (part == 0)
                        gs_split2(xg[t][2 * k], xg[t][2 * k + 1], p1[k], p2[k], p3[k]);
                    else
                        gs_split2(hg[t][2 * k], hg[t][2 * k + 1], p1[k], p2[k], p3[k]);
                }
                const bf16x8 b1 = __builtin_bit_cast(bf16x8, (u32x4{p1[0], p1[1], p1[2], p1[3]}));
                const bf16x8 b2 = __builtin_bit_cast(bf16x8, (u32x4{p2[0], p2[1], p2[2], p2[3]}));
                const bf16x8 b3 = __builtin_bit_cast(bf16x8, (u32x4{p3[0], p3[1], p3[2], p3[3]}));
#pragma unroll
                for (int g = 0; g < GATES; ++g) {
                    const int mat = part * GATES + g;
                    const int d = (part == 1 && g == GATES - 1) ? GATES : g;   // the candidate's hh part stays separate
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) {
                        const u32x4* q = wl + (((mat * 2 + mb) * 4 + t) * 3) * 64;
                        const bf16x8 a1 = __builtin_bit_cast(bf16x8, q[0]);
                        const bf16x8 a2 = __builtin_bit_cast(bf16x8, q[64]);
                        const bf16x8 a3 = __builtin_bit_cast(bf16x8, q[128]);
                        // the six term pairs of weight >= 2^-16, smallest first
                        acc[d][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc[d][mb], 0, 0, 0);
                        acc[d][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc[d][mb], 0, 0, 0);
                        acc[d][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc[d][mb], 0, 0, 0);
                        acc[d][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc[d][mb], 0, 0, 0);
                        acc[d][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc[d][mb], 0, 0, 0);
                        acc[d][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[d][mb], 0, 0, 0);
                    }
                }
            }
        }
        // F16: back to the scale of the gate pre-activations (exact) and the ih biases, element by element where the gates are evaluated
        const float unpx = F16 ? gs_pow2(-kp) * unw : 1.f;
        // ---- h_prev again, in accumulator layout (row (r, lane half) = channel, column = pixel): an L2 hit ---------------
        float hv[2][16];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                hv[ct][r] = hb ? hb[(unsigned)(ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * P32 + pxo] : 0.f;
        float* ob = a.out + base;
        const unsigned o_pxo = pxo;
        const int o_lhi = lhi;
        const bool o_valid = valid;
        // the x operand registers are free: the next segment's x loads fly while this one's gates are evaluated
        sg += stride;
        if (sg < a.nseg) load(sg);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * o_lhi;
                auto pre = [&](int d) {
                    if constexpr (F16) return acc[d][ct][r] * unpx + (d < GATES ? Bs[d * GS_F + co] : 0.f);
                    else return acc[d][ct][r];
                };
                float o;
                if constexpr (GATES == 3) {  // rnn_cells.py:118-127
                    const float rg = gs_sigmoid(pre(0));
                    const float z = gs_sigmoid(pre(1));
                    const float n = gs_tanh(pre(2) + rg * pre(3));
                    o = n * (1.0f - z) + z * hv[ct][r];
                } else {  // rnn_cells.py:255-261
                    const float f = gs_sigmoid(pre(0));
                    const float c = gs_tanh(pre(1) + f * pre(2));
                    o = c + f * (hv[ct][r] - c);
                }
                if (o_valid) ob[(unsigned)co * P32 + o_pxo] = o;
                vmax = fmaxf(vmax, fabsf(o));          // (lanes past the last pixel repeat pixel 0: harmless for the maximum)
            }
    }
    if (a.xmax) {               // one conditional atomic per workgroup (bit patterns of non-negative floats order like unsigned integers)
        for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
        __syncthreads();        // (every wave is done with the weights: the first floats of the LDS are free)
        float* red = reinterpret_cast<float*>(smem_gs);
        if (lane == 0) red[wave] = vmax;
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < GS_NT / 64; ++w) vmax = fmaxf(vmax, red[w]);
            if (__float_as_uint(vmax) > __hip_atomic_load(reinterpret_cast<unsigned*>(a.xmax), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                atomicMax(reinterpret_cast<unsigned*>(a.xmax), __float_as_uint(vmax));
        }
    }
}

template <int GATES, bool F16>
static int launch_gated_sb(const MrxGatedSbArgs& a, hipStream_t st) {
    constexpr size_t lds = (size_t)(2 * GATES * 2 * 4 * 3 * 64) * 16 + sizeof(float) * GATES * GS_F;
    static_assert(lds <= 160 * 1024, "all weight matrices resident in LDS");
    static bool attr_done = false;  // once per instantiation: keeps launches legal under hipGraph capture
    static int n_cu = 0;
    if (!attr_done) {
        MRX_HIP(hipFuncSetAttribute((const void*)k_gated_cell_sb<GATES, F16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int dev = 0;
        hipDeviceProp_t prop;
        MRX_HIP(hipGetDevice(&dev));
        MRX_HIP(hipGetDeviceProperties(&prop, dev));
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        attr_done = true;
    }
    const long long nblk_need = (a.nseg + GS_NT / 64 - 1) / (GS_NT / 64);
    const unsigned nblk = (unsigned)(nblk_need < n_cu ? nblk_need : n_cu);  // persistent: the weights are staged once per workgroup
    hipLaunchKernelGGL((k_gated_cell_sb<GATES, F16>), dim3(nblk), dim3(GS_NT), lds, st, a);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
int mrx_gated_sb_launch(const MrxGatedSbArgs& a, int gates, hipStream_t st) {
    // the two-term fp16 form (per-pixel scales) is the default: GRU 88.7 -> 70.5 us, MGU 66.9 -> 54.9 us at 640 x 372, error against float64 1.1e-7 as
    // the three-term bf16 form, which MRIDC_AMD_ARITH=bf16x3 selects.  (Its first version spilled -- 356 bytes of scratch per lane, 184 us: h_prev is now
    // requested after the x part is issued and the accumulators are scaled back where the gates are evaluated, not in a pass of their own.)
    const int f16 = mrx_arith() == MRX_ARITH_F16X2 ? 1 : 0;
    if (f16) return gates == 3 ? launch_gated_sb<3, true>(a, st) : launch_gated_sb<2, true>(a, st);
    return gates == 3 ? launch_gated_sb<3, false>(a, st) : launch_gated_sb<2, false>(a, st);
}

// ---- Conv2dGRU layer of the Recurrent Variational Network (recurrentvarnet/conv2gru.py:139-157), 1x1 gates on 64 features -----------------
//   update = sigmoid(Wu [x; h] + bu)   reset = sigmoid(Wr [x; h] + br)   delta = tanh(Wo [x; h * reset] + bo)
//   h_new  = h * (1 - update) + delta * update          (also written as ReLU(h_new): the next layer's input, :157)
// The formulation of k_conv2dgru_cell (gated_cell.hip) with the three-term operand split: h * reset is formed in the accumulator layout and
// fed to the last GEMM as a B operand, eight accumulator registers per contraction step, the candidate's hh weights packed with their
// contraction index in that order (register R = 8 t + j of lane half l holds channel 32 (R >> 4) + (R & 3) + 8 ((R & 15) >> 2) + 4 l).
__host__ __device__ constexpr int gs_chan(int R, int half) { return 32 * (R >> 4) + (R & 3) + 8 * ((R & 15) >> 2) + 4 * half; }

// mats 0..2: Wu, Wr, Wo columns 0..63 (x part); 3, 4: Wu, Wr columns 64..127 (h part); 5: Wo columns 64..127 in accumulator order
__global__ void k_conv2dgru_pack_sb(const float* __restrict__ wu, const float* __restrict__ wr, const float* __restrict__ wo,
                                    u32x4* __restrict__ out) {
    const int total = 6 * 2 * 4 * 3 * 64;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int lane = i & 63;
        int r = i >> 6;
        const int term = r % 3;
        r /= 3;
        const int t = r & 3, mb = (r >> 2) & 1, mat = r >> 3, half = lane >> 5;
        const float* w = (mat == 0 || mat == 3) ? wu : (mat == 1 || mat == 4) ? wr : wo;
        const int row = mb * 32 + (lane & 31);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int col = mat < 3 ? 16 * t + 8 * half + j : (mat < 5 ? GS_F + 16 * t + 8 * half + j : GS_F + gs_chan(8 * t + j, half));
            v[j] = w[(long long)row * (2 * GS_F) + col];
        }
        unsigned p[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned p1, p2, p3;
            gs_split2(v[2 * k], v[2 * k + 1], p1, p2, p3);
            p[k] = term == 0 ? p1 : (term == 1 ? p2 : p3);
        }
        out[i] = u32x4{p[0], p[1], p[2], p[3]};
    }
}
// fp16 section: [6*2*4*3*64 ..) two terms of w 2^kw, [(mat * 2 + mb) * 4 + t) * 2 + term) * 64 + lane], then the header (kw)
__global__ void k_conv2dgru_wscale(const float* __restrict__ wu, const float* __restrict__ wr, const float* __restrict__ wo, u32x4* __restrict__ out) {
    __shared__ float red[256];
    float m = 0.f;
    for (int i = threadIdx.x; i < GS_F * 2 * GS_F; i += 256) m = fmaxf(m, fmaxf(fabsf(wu[i]), fmaxf(fabsf(wr[i]), fabsf(wo[i]))));
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) out[6 * 2 * 4 * 5 * 64] = u32x4{(unsigned)gs_scale_exp(red[0]), 0u, 0u, 0u};
}
__global__ void k_conv2dgru_pack_f16(const float* __restrict__ wu, const float* __restrict__ wr, const float* __restrict__ wo,
                                     u32x4* __restrict__ out) {
    const int total = 6 * 2 * 4 * 2 * 64, off = 6 * 2 * 4 * 3 * 64;
    const float sw = gs_pow2((int)out[6 * 2 * 4 * 5 * 64][0]);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int lane = i & 63;
        int r = i >> 6;
        const int term = r & 1;
        r >>= 1;
        const int t = r & 3, mb = (r >> 2) & 1, mat = r >> 3, half = lane >> 5;
        const float* w = (mat == 0 || mat == 3) ? wu : (mat == 1 || mat == 4) ? wr : wo;
        const int row = mb * 32 + (lane & 31);
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int col = mat < 3 ? 16 * t + 8 * half + j : (mat < 5 ? GS_F + 16 * t + 8 * half + j : GS_F + gs_chan(8 * t + j, half));
            v[j] = w[(long long)row * (2 * GS_F) + col] * sw;
        }
        unsigned p[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned p1, p2;
            gs_split2h(v[2 * k], v[2 * k + 1], p1, p2);
            p[k] = term == 0 ? p1 : p2;
        }
        out[off + i] = u32x4{p[0], p[1], p[2], p[3]};
    }
}
int mrx_conv2dgru_sb_pack(const float* wu, const float* wr, const float* wo, float* packed, hipStream_t st) {
    const int total = 6 * 2 * 4 * 3 * 64;
    hipLaunchKernelGGL(k_conv2dgru_wscale, dim3(1), dim3(256), 0, st, wu, wr, wo, reinterpret_cast<u32x4*>(packed));
    hipLaunchKernelGGL(k_conv2dgru_pack_f16, dim3((6 * 2 * 4 * 2 * 64 + 255) / 256), dim3(256), 0, st, wu, wr, wo, reinterpret_cast<u32x4*>(packed));
    hipLaunchKernelGGL(k_conv2dgru_pack_sb, dim3((total + 255) / 256), dim3(256), 0, st, wu, wr, wo, reinterpret_cast<u32x4*>(packed));
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// six term products of one (matrix, cout block, step) into an accumulator
__device__ __forceinline__ void gs_mma6(f32x16& acc, const u32x4* q, bf16x8 b1, bf16x8 b2, bf16x8 b3) {
    const bf16x8 a1 = __builtin_bit_cast(bf16x8, q[0]);
    const bf16x8 a2 = __builtin_bit_cast(bf16x8, q[64]);
    const bf16x8 a3 = __builtin_bit_cast(bf16x8, q[128]);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc, 0, 0, 0);
}

// three term products (two fp16 terms per operand) of one (matrix, cout block, step)
__device__ __forceinline__ void gs_mma3h(f32x16& acc, const u32x4* q, f16x8 b1, f16x8 b2) {
    const f16x8 a1 = __builtin_bit_cast(f16x8, q[0]);
    const f16x8 a2 = __builtin_bit_cast(f16x8, q[64]);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc, 0, 0, 0);
}

// F16: two fp16 terms per operand with ONE scale per pixel for x, h_prev and h * reset (|h reset| <= |h|): x is scaled first, the accumulators move
// exactly to h_prev's scale when that is the smaller one (k_gated_cell_sb), the biases are added where the gates are evaluated.
template <bool F16>
__global__ __launch_bounds__(GS_NT, 1) void k_conv2dgru_cell_sb(MrxConv2dGruSbArgs a) {
    float vmax2 = 0.f;                             // maximum of ReLU(new state) of this lane (a.xmax)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_gs[];
    constexpr int NTM = F16 ? 2 : 3, NW = 6 * 2 * 4 * NTM * 64;
    u32x4* Wl = reinterpret_cast<u32x4*>(smem_gs);
    float* Bs = reinterpret_cast<float*>(smem_gs + (size_t)NW * 16);   // biases [3][64]: update, reset, out
    const int tid = threadIdx.x;
    float unw = 1.f;
    {
        const u32x4* src = reinterpret_cast<const u32x4*>(a.packed) + (F16 ? 6 * 2 * 4 * 3 * 64 : 0);
        for (int i = tid; i < NW; i += GS_NT) Wl[i] = src[i];
        if (tid < 3 * GS_F) Bs[tid] = a.bias ? a.bias[tid] : 0.f;
        if constexpr (F16) unw = gs_pow2(-(int)reinterpret_cast<const u32x4*>(a.packed)[6 * 2 * 4 * 5 * 64][0]);
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const u32x4* wl = Wl + lane;
    const long long stride = (long long)gridDim.x * (GS_NT / 64);
    const unsigned P32 = (unsigned)a.P;
    float xg[4][8], hg[4][8];
    const float* hb = nullptr;
    long long base = 0;
    unsigned pxo = 0;
    bool valid = false;
    int lhi = 0;
    auto load = [&](long long sg) {  // see k_gated_cell_sb
        int l31 = lane & 31;
        lhi = lane >> 5;
        asm volatile("" : "+v"(l31), "+v"(lhi));
        const long long b = sg / a.nsegb;
        const long long px = (sg - b * a.nsegb) * 32 + l31;
        valid = px < a.P;
        base = __builtin_amdgcn_readfirstlane((int)b) * (long long)GS_F * a.P;
        pxo = valid ? (unsigned)px : 0u;
        const float* xb = a.x + base;
        hb = a.h ? a.h + base : nullptr;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) xg[t][j] = xb[(unsigned)(16 * t + 8 * lhi + j) * P32 + pxo];
    };
    auto load_h = [&]() {
        if (hb) {
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 8; ++j) hg[t][j] = hb[(unsigned)(16 * t + 8 * lhi + j) * P32 + pxo];
        }
    };
    long long sg = (long long)blockIdx.x * (GS_NT / 64) + wave;
    if (sg < a.nseg) load(sg);
    while (sg < a.nseg) {
        if constexpr (!F16) load_h();
        f32x16 acc[3][2];  // 0 update, 1 reset (then h * reset), 2 candidate
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[d][ct][r] = F16 ? 0.f : Bs[d * GS_F + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi];
        float spx = 1.f;
        int kp = 0;
        if constexpr (F16) {
            float m = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(xg[t][j]));
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            kp = gs_scale_exp(m);
            spx = gs_pow2(kp);
        }
        // (Wu, Wr, Wo) x   and   (Wu, Wr) h
#pragma unroll
        for (int part = 0; part < 2; ++part) {
            if (part == 1 && !hb) break;
            if constexpr (F16) {
                if (part == 1) {
                    load_h();
                    float m = 0.f;
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(hg[t][j]));
                    m = fmaxf(m, __shfl_xor(m, 32, 64));
                    const int kh = gs_scale_exp(m);
                    const bool lower = m > 0.f && kh < kp;
                    const float f = lower ? gs_pow2(kh - kp) : 1.f;
                    kp = lower ? kh : kp;
                    spx = gs_pow2(kp);
#pragma unroll
                    for (int d = 0; d < 3; ++d)
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[d][ct][r] *= f;
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if constexpr (F16) {
                    unsigned p1[4], p2[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if (part == 0)
                            gs_split2h(xg[t][2 * k] * spx, xg[t][2 * k + 1] * spx, p1[k], p2[k]);
                        else
                            gs_split2h(hg[t][2 * k] * spx, hg[t][2 * k + 1] * spx, p1[k], p2[k]);
                    }
                    const f16x8 b1 = __builtin_bit_cast(f16x8, (u32x4{p1[0], p1[1], p1[2], p1[3]}));
                    const f16x8 b2 = __builtin_bit_cast(f16x8, (u32x4{p2[0], p2[1], p2[2], p2[3]}));
#pragma unroll
                    for (int g = 0; g < (part == 0 ? 3 : 2); ++g)
#pragma unroll
                        for (int mb = 0; mb < 2; ++mb) gs_mma3h(acc[g][mb], wl + ((((part * 3 + g) * 2 + mb) * 4 + t) * 2) * 64, b1, b2);
                    continue;
                }
                unsigned p1[4], p2[4], p3[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (part == 0)
                        gs_split2(xg[t][2 * k], xg[t][2 * k + 1], p1[k], p2[k], p3[k]);
                    else
                        gs_split2(hg[t][2 * k], hg[t][2 * k + 1], p1[k], p2[k], p3[k]);
                }
                const bf16x8 b1 = __builtin_bit_cast(bf16x8, (u32x4{p1[0], p1[1], p1[2], p1[3]}));
                const bf16x8 b2 = __builtin_bit_cast(bf16x8, (u32x4{p2[0], p2[1], p2[2], p2[3]}));
                const bf16x8 b3 = __builtin_bit_cast(bf16x8, (u32x4{p3[0], p3[1], p3[2], p3[3]}));
#pragma unroll
                for (int g = 0; g < (part == 0 ? 3 : 2); ++g)
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) gs_mma6(acc[g][mb], wl + ((((part * 3 + g) * 2 + mb) * 4 + t) * 3) * 64, b1, b2, b3);
            }
        }
        float hv[2][16];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                hv[ct][r] = hb ? hb[(unsigned)(ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * P32 + pxo] : 0.f;
        float* ob = a.out + base;
        float* orl = a.out_relu ? a.out_relu + base : nullptr;
        const unsigned o_pxo = pxo;
        const int o_lhi = lhi;
        const bool o_valid = valid, have_h = hb != nullptr;
        sg += stride;
        if (sg < a.nseg) load(sg);  // next segment's x loads fly during the rest of this one
        const float unpx = F16 ? gs_pow2(-kp) * unw : 1.f;     // F16: back to the scale of the pre-activations (exact)
        const float spx_o = spx;
        if (have_h) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pre = F16 ? acc[1][ct][r] * unpx + Bs[GS_F + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * o_lhi] : acc[1][ct][r];
                    acc[1][ct][r] = hv[ct][r] * gs_sigmoid(pre);
                }
            // Wo_h (h * reset): the B operand of step t is accumulator registers R = 8 t .. 8 t + 7
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if constexpr (F16) {   // |h reset| <= |h|: the pixel's scale covers it
                    unsigned p1[4], p2[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int R0 = 8 * t + 2 * k, R1 = R0 + 1;
                        gs_split2h(acc[1][R0 >> 4][R0 & 15] * spx_o, acc[1][R1 >> 4][R1 & 15] * spx_o, p1[k], p2[k]);
                    }
                    const f16x8 b1 = __builtin_bit_cast(f16x8, (u32x4{p1[0], p1[1], p1[2], p1[3]}));
                    const f16x8 b2 = __builtin_bit_cast(f16x8, (u32x4{p2[0], p2[1], p2[2], p2[3]}));
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) gs_mma3h(acc[2][mb], wl + (((5 * 2 + mb) * 4 + t) * 2) * 64, b1, b2);
                    continue;
                }
                unsigned p1[4], p2[4], p3[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int R0 = 8 * t + 2 * k, R1 = R0 + 1;
                    gs_split2(acc[1][R0 >> 4][R0 & 15], acc[1][R1 >> 4][R1 & 15], p1[k], p2[k], p3[k]);
                }
                const bf16x8 b1 = __builtin_bit_cast(bf16x8, (u32x4{p1[0], p1[1], p1[2], p1[3]}));
                const bf16x8 b2 = __builtin_bit_cast(bf16x8, (u32x4{p2[0], p2[1], p2[2], p2[3]}));
                const bf16x8 b3 = __builtin_bit_cast(bf16x8, (u32x4{p3[0], p3[1], p3[2], p3[3]}));
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) gs_mma6(acc[2][mb], wl + (((5 * 2 + mb) * 4 + t) * 3) * 64, b1, b2, b3);
            }
        }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * o_lhi;
                const float u = gs_sigmoid(F16 ? acc[0][ct][r] * unpx + Bs[co] : acc[0][ct][r]);
                const float dl = gs_tanh(F16 ? acc[2][ct][r] * unpx + Bs[2 * GS_F + co] : acc[2][ct][r]);
                const float o = hv[ct][r] * (1.0f - u) + dl * u;
                if (o_valid) {
                    ob[(unsigned)co * P32 + o_pxo] = o;
                    if (orl) orl[(unsigned)co * P32 + o_pxo] = o > 0.f ? o : 0.f;
                }
                vmax2 = fmaxf(vmax2, o);                 // (max of ReLU(o): starts at 0; lanes past the last pixel repeat pixel 0)
            }
    }
    if (a.xmax) {               // one conditional atomic per workgroup (bit patterns of non-negative floats order like unsigned integers)
        for (int off = 32; off > 0; off >>= 1) vmax2 = fmaxf(vmax2, __shfl_xor(vmax2, off, 64));
        __syncthreads();        // (every wave is done with the weights: the first floats of the LDS are free)
        float* red = reinterpret_cast<float*>(smem_gs);
        if ((tid & 63) == 0) red[tid >> 6] = vmax2;
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < GS_NT / 64; ++w) vmax2 = fmaxf(vmax2, red[w]);
            if (__float_as_uint(vmax2) > __hip_atomic_load(reinterpret_cast<unsigned*>(a.xmax), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                atomicMax(reinterpret_cast<unsigned*>(a.xmax), __float_as_uint(vmax2));
        }
    }
}
int mrx_conv2dgru_sb_launch(const MrxConv2dGruSbArgs& a, hipStream_t st) {
    constexpr size_t lds = (size_t)(6 * 2 * 4 * 3 * 64) * 16 + sizeof(float) * 3 * GS_F;
    static bool attr_done = false;
    static int n_cu = 0;
    if (!attr_done) {
        MRX_HIP(hipFuncSetAttribute((const void*)k_conv2dgru_cell_sb<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        MRX_HIP(hipFuncSetAttribute((const void*)k_conv2dgru_cell_sb<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int dev = 0;
        hipDeviceProp_t prop;
        MRX_HIP(hipGetDevice(&dev));
        MRX_HIP(hipGetDeviceProperties(&prop, dev));
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        attr_done = true;
    }
    const long long nblk_need = (a.nseg + GS_NT / 64 - 1) / (GS_NT / 64);
    const unsigned nblk = (unsigned)(nblk_need < n_cu ? nblk_need : n_cu);
    const int f16 = mrx_arith() == MRX_ARITH_F16X2 ? 1 : 0;   // 0: the three-term bf16 form (RecurrentVarNet 164 -> 182 slices/s with fp16)
    if (f16)
        hipLaunchKernelGGL(k_conv2dgru_cell_sb<true>, dim3(nblk), dim3(GS_NT), lds, st, a);
    else
        hipLaunchKernelGGL(k_conv2dgru_cell_sb<false>, dim3(nblk), dim3(GS_NT), lds, st, a);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- 1x1 convolution 128 -> 128 with the IndRNN cell as an optional epilogue: out = act(W x + bias [+ hh * h_prev]) ------------------------
// (rnn_cells.py:384-391 for the 128-feature cells of the qRIM; the channel contraction of thin 3x3 convolutions, ops.conv3x3_taps).  The
// fp32-MFMA kernel (k_conv1x1_sq<2>) spends half its time on the matrix pipe at this width; with the three-term split the layer is bound by
// its three 33 MB tensors.  Weights (98 KB as split terms) resident in LDS, x loaded 64 channels at a time in the B-operand k order.
// packed[((((ob * 2 + ib) * 2 + mb) * 4 + t) * 3 + term) * 64 + lane][j] = term( W[ob * 64 + mb * 32 + lane % 32][ib * 64 + 16 t + 8 (lane / 32) + j] )
__global__ void k_conv1x1_pack_sb128(const float* __restrict__ w, u32x4* __restrict__ out) {
    const int total = 2 * 2 * 2 * 4 * 3 * 64;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int lane = i & 63;
        int r = i >> 6;
        const int term = r % 3;
        r /= 3;
        const int t = r & 3, mb = (r >> 2) & 1, ib = (r >> 3) & 1, ob = r >> 4;
        const int row = ob * 64 + mb * 32 + (lane & 31), col0 = ib * 64 + 16 * t + 8 * (lane >> 5);
        unsigned p[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned p1, p2, p3;
            gs_split2(w[(long long)row * 128 + col0 + 2 * k], w[(long long)row * 128 + col0 + 2 * k + 1], p1, p2, p3);
            p[k] = term == 0 ? p1 : (term == 1 ? p2 : p3);
        }
        out[i] = u32x4{p[0], p[1], p[2], p[3]};
    }
}
// the precision-16 section: packed16[(((ob * 2 + ib) * 2 + mb) * 4 + t) * 64 + lane][j] = fp16( W[the same element] ), round to nearest even
__global__ void k_conv1x1_pack_sb128_p16(const float* __restrict__ w, u32x4* __restrict__ out) {
    const int total = 2 * 2 * 2 * 4 * 64;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int lane = i & 63, r = i >> 6;
        const int t = r & 3, mb = (r >> 2) & 1, ib = (r >> 3) & 1, ob = r >> 4;
        const int row = ob * 64 + mb * 32 + (lane & 31), col0 = ib * 64 + 16 * t + 8 * (lane >> 5);
        unsigned p[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f16x2 h = {(_Float16)w[(long long)row * 128 + col0 + 2 * k], (_Float16)w[(long long)row * 128 + col0 + 2 * k + 1]};
            p[k] = __builtin_bit_cast(unsigned, h);
        }
        out[i] = u32x4{p[0], p[1], p[2], p[3]};
    }
}
int mrx_conv1x1_sb128_pack(const float* w, float* packed, hipStream_t st) {
    const int total = 2 * 2 * 2 * 4 * 3 * 64;
    hipLaunchKernelGGL(k_conv1x1_pack_sb128, dim3((total + 255) / 256), dim3(256), 0, st, w, reinterpret_cast<u32x4*>(packed));
    MRX_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_conv1x1_pack_sb128_p16, dim3((total / 3 + 255) / 256), dim3(256), 0, st, w, reinterpret_cast<u32x4*>(packed) + total);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// NOB: output blocks of 64 channels computed (1: the first 64 rows of W only, out [B,64,P] -- the contraction of a thin 3x3 convolution)
// P16 (round 6): the reference's `precision: 16` inference arithmetic (base_qcirim_run.yaml:204: torch.autocast(float16) -- rnn_cells.py:384-391 under it): x and
// W rounded to fp16 once (a plain cast, as autocast's), ONE product per multiply on v_mfma_f32_32x32x16_f16, fp32 sums; hh * h_prev, bias, activation in fp32.
template <int NOB, bool P16 = false>
__global__ __launch_bounds__(GS_NT, 1) void k_conv1x1_sb128(MrxConv1x1SbArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_gs[];
    constexpr int C = 128, NW3 = 2 * 2 * 2 * 4 * 3 * 64, NW = P16 ? NW3 / 3 : NW3;
    u32x4* Wl = reinterpret_cast<u32x4*>(smem_gs);
    float* Bs = reinterpret_cast<float*>(smem_gs + (size_t)NW * 16);   // bias [C], hh [C]
    const int tid = threadIdx.x;
    {
        const u32x4* src = reinterpret_cast<const u32x4*>(a.packed) + (P16 ? NW3 : 0);
        for (int i = tid; i < NW; i += GS_NT) Wl[i] = src[i];
        if (tid < C) {
            Bs[tid] = a.bias ? a.bias[tid] : 0.f;
            Bs[C + tid] = a.hh ? a.hh[tid] : 0.f;
        }
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const u32x4* wl = Wl + lane;
    const long long stride = (long long)gridDim.x * (GS_NT / 64);
    const unsigned P32 = (unsigned)a.P;
    const float neg = a.act == MRX_ACT_RELU ? 0.f : (a.act == MRX_ACT_LEAKY ? a.slope : 1.f);
    float vmax = 0.f;
    for (long long sg = (long long)blockIdx.x * (GS_NT / 64) + wave; sg < a.nseg; sg += stride) {
        int l31 = lane & 31, lhi = lane >> 5;
        asm volatile("" : "+v"(l31), "+v"(lhi));  // keep the channel offsets out of loop-invariant hoisting (spills)
        const long long b = sg / a.nsegb;
        const long long px = (sg - b * a.nsegb) * 32 + l31;
        const bool valid = px < a.P;
        const long long base = __builtin_amdgcn_readfirstlane((int)b) * (long long)C * a.P;
        const unsigned pxo = valid ? (unsigned)px : 0u;
        const float* xb = a.x + base;
        float xg[2][4][8];
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int j = 0; j < 8; ++j) xg[ib][t][j] = xb[(unsigned)(ib * 64 + 16 * t + 8 * lhi + j) * P32 + pxo];
        // (the pixel's 128 channels are requested here, all 64 loads of the lane: left to itself the scheduler moved each group of eight down to
        // its own matrix step and waited for it there -- eight memory round trips per 32 pixels with the matrix pipe idle under each)
        __builtin_amdgcn_sched_barrier(0);
        f32x16 acc[NOB][2];
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ob][ct][r] = Bs[ob * 64 + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi];
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if constexpr (P16) {
                    f16x8 bx;
#pragma unroll
                    for (int j = 0; j < 8; ++j) bx[j] = (_Float16)xg[ib][t][j];
#pragma unroll
                    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
                        for (int mb = 0; mb < 2; ++mb)
                            acc[ob][mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wl[((((ob * 2 + ib) * 2 + mb) * 4 + t)) * 64]), bx, acc[ob][mb], 0, 0, 0);
                    continue;
                }
                unsigned p1[4], p2[4], p3[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) gs_split2(xg[ib][t][2 * k], xg[ib][t][2 * k + 1], p1[k], p2[k], p3[k]);
                const bf16x8 b1 = __builtin_bit_cast(bf16x8, (u32x4{p1[0], p1[1], p1[2], p1[3]}));
                const bf16x8 b2 = __builtin_bit_cast(bf16x8, (u32x4{p2[0], p2[1], p2[2], p2[3]}));
                const bf16x8 b3 = __builtin_bit_cast(bf16x8, (u32x4{p3[0], p3[1], p3[2], p3[3]}));
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) gs_mma6(acc[ob][mb], wl + (((((ob * 2 + ib) * 2 + mb) * 4 + t) * 3)) * 64, b1, b2, b3);
            }
        float* ob_ = a.out + (NOB == 2 ? base : __builtin_amdgcn_readfirstlane((int)b) * (long long)64 * a.P);
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
            float hv[2][16];
            if (a.hprev) {
                const float* hb = a.hprev + base;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int r = 0; r < 16; ++r) hv[ct][r] = hb[(unsigned)(ob * 64 + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * P32 + pxo];
            }
            // the arithmetic first, then ONE predicated block of stores (a branch around every store would split this into 32 basic blocks);
            // lanes past the last pixel hold pixel 0's values again: harmless for the maximum
            if (a.hprev) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[ob][ct][r] += Bs[C + ob * 64 + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi] * hv[ct][r];
            }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[ob][ct][r];
                    acc[ob][ct][r] = v > 0.f ? v : v * neg;
                    vmax = fmaxf(vmax, fabsf(acc[ob][ct][r]));
                }
            if (valid) {
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int r = 0; r < 16; ++r) ob_[(unsigned)(ob * 64 + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * P32 + pxo] = acc[ob][ct][r];
            }
        }
    }
    if (a.xmax) {               // one atomic per WORKGROUP, and only if it raises the bound (2048 same-address atomics cost the launch 16 us)
        for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
        __syncthreads();        // (every wave is done with the weights: the first floats of the LDS are free)
        float* red = reinterpret_cast<float*>(smem_gs);
        if (lane == 0) red[wave] = vmax;
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < GS_NT / 64; ++w) vmax = fmaxf(vmax, red[w]);
            // (bit patterns of non-negative floats order like unsigned integers)
            if (__float_as_uint(vmax) > __hip_atomic_load(reinterpret_cast<unsigned*>(a.xmax), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                atomicMax(reinterpret_cast<unsigned*>(a.xmax), __float_as_uint(vmax));
        }
    }
}
int mrx_conv1x1_sb128_launch(const MrxConv1x1SbArgs& a, hipStream_t st) {
    constexpr size_t lds = (size_t)(2 * 2 * 2 * 4 * 3 * 64) * 16 + sizeof(float) * 2 * 128;
    constexpr size_t lds16 = (size_t)(2 * 2 * 2 * 4 * 64) * 16 + sizeof(float) * 2 * 128;
    static bool attr_done = false;
    static int n_cu = 0;
    if (!attr_done) {
        MRX_HIP(hipFuncSetAttribute((const void*)k_conv1x1_sb128<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        MRX_HIP(hipFuncSetAttribute((const void*)k_conv1x1_sb128<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int dev = 0;
        hipDeviceProp_t prop;
        MRX_HIP(hipGetDevice(&dev));
        MRX_HIP(hipGetDeviceProperties(&prop, dev));
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        attr_done = true;
    }
    const long long nblk_need = (a.nseg + GS_NT / 64 - 1) / (GS_NT / 64);
    const unsigned nblk = (unsigned)(nblk_need < n_cu ? nblk_need : n_cu);
    if (a.p16) {
        if (a.head)
            hipLaunchKernelGGL((k_conv1x1_sb128<1, true>), dim3(nblk), dim3(GS_NT), lds16, st, a);
        else
            hipLaunchKernelGGL((k_conv1x1_sb128<2, true>), dim3(nblk), dim3(GS_NT), lds16, st, a);
    } else if (a.head)
        hipLaunchKernelGGL(k_conv1x1_sb128<1>, dim3(nblk), dim3(GS_NT), lds, st, a);
    else
        hipLaunchKernelGGL(k_conv1x1_sb128<2>, dim3(nblk), dim3(GS_NT), lds, st, a);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
