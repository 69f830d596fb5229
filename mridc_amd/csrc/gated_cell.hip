// gated_cell.hip -- ConvGRUCell / ConvMGUCell with 1x1 kernels for gfx950 (reference models/rim/rnn_cells.py:112-127 and
// :249-261; the RIM / CIRIM model-zoo configs use recurrent_kernels [1, 1, 0]), fp32 in / fp32 out on v_mfma_f32_32x32x2_f32.
//
// The reference runs two convolutions (`ih`: Cin -> GATES*F, `hh`: F -> GATES*F), chunks them and applies the gate math.
// With 1x1 kernels both are per-pixel GEMMs, so the whole cell is ONE launch here: nothing of the GATES*F-channel
// intermediates (183 MB each at 640x372) is ever written.
//
//   * a wave owns 32 consecutive pixels (the 1x1 cell has no spatial structure: the image is a flat run of H*W pixels);
//     x and h_prev are read straight from HBM into the MFMA B-operand layout (lane = pixel, register = channel pair):
//     32 coalesced 128-byte rows per operand, no LDS hop;
//   * the packed weights of all 2*GATES matrices (96 KB for GRU) stay in LDS for the life of the persistent workgroup:
//     one ds_read_b32 per lane per MFMA, conflict-free, read RL-style a few steps ahead of the MFMA that consumes them;
//   * accumulators: the reset/update (GRU) or forget (MGU) pre-activations sum their ih and hh parts in one accumulator;
//     the candidate keeps ih and hh apart (the gate multiplies only the hh part).  They start at the ih bias;
//   * the gate math runs on the accumulators in registers; h_prev is re-read in accumulator layout (an L2 hit).
#include <cstdint>
#include <cstdlib>

#include "mrx_common.h"
#include "gated_cell_sb.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define GC_NT 512
#define GC_F 64
#define GC_PF 3  // operand prefetch distance (MFMA steps)

struct GatedArgs {
    const float* x;       // [B,64,P]
    const float* h;       // [B,64,P] or null (= zeros)
    const float* packed;  // mrx_gated_cell_pack
    const float* b_ih;    // [GATES*64] or null
    float* out;           // [B,64,P]
    long long P, nsegb, nseg;  // pixels per image, 32-pixel segments per image, segments in total
};

// packed index (((mat*2 + mb)*32 + s)*2 + half)*32 + m  <-  W_mat[mb*32 + m][2*s + half],  mat = gate (ih) | GATES + gate (hh)
__global__ void k_gated_pack(const float* __restrict__ w_ih, const float* __restrict__ w_hh, float* __restrict__ out, int gates) {
    const int total = 2 * gates * GC_F * GC_F;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int m = i & 31, half = (i >> 5) & 1, s = (i >> 6) & 31, mb = (i >> 11) & 1, mat = i >> 12;
        const int row = mb * 32 + m, col = 2 * s + half;
        const float* w = mat < gates ? w_ih : w_hh;
        const int g = mat < gates ? mat : mat - gates;
        out[i] = w[(long long)(g * GC_F + row) * GC_F + col];
    }
}

extern "C" int64_t mrx_gated_cell_pack_floats(int Cin, int F, int gates) {
    if (Cin != GC_F || F != GC_F || (gates != 2 && gates != 3)) return -1;
    return (int64_t)2 * gates * GC_F * GC_F + MRX_GATED_SB_PACK_FLOATS(gates);     // fp32 operands, then the split-bf16 ones (gated_cell_sb.hip)
}

extern "C" int mrx_gated_cell_supported(int Cin, int F, int k, int gates) {
    return Cin == GC_F && F == GC_F && k == 1 && (gates == 2 || gates == 3);
}

extern "C" int mrx_gated_cell_pack(const float* w_ih, const float* w_hh, float* packed, int Cin, int F, int gates, void* stream) {
    MRX_REQUIRE(w_ih && w_hh && packed, MRX_EINVAL, "mrx_gated_cell_pack: null pointer");
    MRX_REQUIRE(mrx_gated_cell_supported(Cin, F, 1, gates), MRX_EUNSUP, "mrx_gated_cell_pack: Cin=%d F=%d gates=%d (64/64, 2|3 only)",
                Cin, F, gates);
    const int total = 2 * gates * GC_F * GC_F;
    hipLaunchKernelGGL(k_gated_pack, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_ih, w_hh, packed, gates);
    MRX_LAUNCH_CHECK();
    return mrx_gated_sb_pack(w_ih, w_hh, packed + total, gates, (hipStream_t)stream);
}

// 1/(1+e^-x) and 1 - 2/(e^2x + 1) on the hardware exp2 / reciprocal: both saturate correctly (exp2 -> inf or 0), absolute error
// ~1e-7, which is what the accumulators carry anyway
__device__ __forceinline__ float gc_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float gc_tanh(float x) {
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

template <int GATES>  // 3 = GRU, 2 = MGU
__global__ __launch_bounds__(GC_NT, 2) void k_gated_cell(GatedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float Ws[];  // [2*GATES][2][32][2][32], then the ih bias [GATES][64]
    constexpr int NMAT = 2 * GATES, MATF = GC_F * GC_F;
    const int tid = threadIdx.x;
    float* Bs = Ws + NMAT * MATF;
    {
        const float4* src = reinterpret_cast<const float4*>(a.packed);
        float4* dst = reinterpret_cast<float4*>(Ws);
        for (int i = tid; i < NMAT * MATF / 4; i += GC_NT) dst[i] = src[i];
        if (tid < GATES * GC_F) Bs[tid] = a.b_ih ? a.b_ih[tid] : 0.f;
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const float* wl = Ws + lane;  // lane = half*32 + m

    // accumulator d: 0 .. GATES-2 = gates whose ih and hh parts add up; GATES-1 = candidate ih part; GATES = candidate hh part
    const long long stride = (long long)gridDim.x * (GC_NT / 64);
    const unsigned P32 = (unsigned)a.P;
    float xg[32], hg[32];
    const float* xb = nullptr;
    const float* hb = nullptr;
    long long base = 0;
    unsigned pxo = 0;
    bool valid = false;
    int lhi = 0;
    // operand loads of one segment: wave-uniform image base + 32-bit per-lane element offsets (64 * P < 2^30 is checked by the
    // host).  The per-lane constants pass through an empty asm every time: the compiler would otherwise hoist the 64 channel
    // offsets out of the segment loop and spill them.
    auto load = [&](long long sg) {
        int l31 = lane & 31;
        lhi = lane >> 5;
        asm volatile("" : "+v"(l31), "+v"(lhi));
        const long long b = sg / a.nsegb;
        const long long px = (sg - b * a.nsegb) * 32 + l31;
        valid = px < a.P;
        base = __builtin_amdgcn_readfirstlane((int)b) * (long long)GC_F * a.P;
        pxo = valid ? (unsigned)px : 0u;  // lanes past the end read pixel 0 and store nothing
        xb = a.x + base;
        hb = a.h ? a.h + base : nullptr;
#pragma unroll
        for (int s = 0; s < 32; ++s) xg[s] = xb[(unsigned)(2 * s + lhi) * P32 + pxo];
    };
    // h_prev operand of the current segment: issued right before the ih GEMMs, which cover its latency
    auto load_h = [&]() {
        if (hb) {
#pragma unroll
            for (int s = 0; s < 32; ++s) hg[s] = hb[(unsigned)(2 * s + lhi) * P32 + pxo];
        }
    };
    long long sg = (long long)blockIdx.x * (GC_NT / 64) + wave;
    if (sg < a.nseg) load(sg);
    while (sg < a.nseg) {
        load_h();
        f32x16 acc[GATES + 1][2];
#pragma unroll
        for (int d = 0; d < GATES + 1; ++d)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[d][ct][r] = d < GATES ? Bs[d * GC_F + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi] : 0.f;

        // ---- ih matrices over x ----------------------------------------------------------------------------------
        {
            constexpr int NS = GATES * 32;
            float ra0[GC_PF + 1], ra1[GC_PF + 1];
#pragma unroll
            for (int t = 0; t < NS + GC_PF; ++t) {
                if (t < NS) {
                    const int mat = t >> 5, s = t & 31;
                    ra0[t % (GC_PF + 1)] = wl[((mat * 2 + 0) * 32 + s) * 64];
                    ra1[t % (GC_PF + 1)] = wl[((mat * 2 + 1) * 32 + s) * 64];
                }
                if (t >= GC_PF) {
                    const int u = t - GC_PF, c = u % (GC_PF + 1);
                    const int g = u >> 5, s = u & 31;  // destination: gate g (the candidate's ih part is accumulator GATES-1)
                    acc[g][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra0[c], xg[s], acc[g][0], 0, 0, 0);
                    acc[g][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra1[c], xg[s], acc[g][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- hh matrices over h_prev (all zero when there is no previous state) ----------------------------------
        if (hb) {
            constexpr int NS = GATES * 32;
            float ra0[GC_PF + 1], ra1[GC_PF + 1];
#pragma unroll
            for (int t = 0; t < NS + GC_PF; ++t) {
                if (t < NS) {
                    const int mat = GATES + (t >> 5), s = t & 31;
                    ra0[t % (GC_PF + 1)] = wl[((mat * 2 + 0) * 32 + s) * 64];
                    ra1[t % (GC_PF + 1)] = wl[((mat * 2 + 1) * 32 + s) * 64];
                }
                if (t >= GC_PF) {
                    const int u = t - GC_PF, c = u % (GC_PF + 1);
                    const int g = u >> 5, s = u & 31;
                    constexpr int LAST = GATES - 1;
                    const int d = g < LAST ? g : GATES;  // the candidate's hh part stays separate
                    acc[d][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra0[c], hg[s], acc[d][0], 0, 0, 0);
                    acc[d][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra1[c], hg[s], acc[d][1], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
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
        // ---- gate math on the accumulators.  Sigmoid and tanh through v_exp_f32 / v_rcp_f32 (1 ulp each): the accurate libm
        // forms cost ~100 VALU instructions per element, on the pipe the fp32 MFMAs run on -- as much time as the GEMMs
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * o_lhi;
                float o;
                if constexpr (GATES == 3) {  // rnn_cells.py:118-127
                    const float rg = gc_sigmoid(acc[0][ct][r]);
                    const float z = gc_sigmoid(acc[1][ct][r]);
                    const float n = gc_tanh(acc[2][ct][r] + rg * acc[3][ct][r]);
                    o = n * (1.0f - z) + z * hv[ct][r];
                } else {  // rnn_cells.py:255-261
                    const float f = gc_sigmoid(acc[0][ct][r]);
                    const float c = gc_tanh(acc[1][ct][r] + f * acc[2][ct][r]);
                    o = c + f * (hv[ct][r] - c);
                }
                if (o_valid) ob[(unsigned)co * P32 + o_pxo] = o;
            }
    }
}

template <int GATES>
static int launch_gated(const GatedArgs& a, hipStream_t st) {
    constexpr size_t lds = sizeof(float) * (2 * GATES * GC_F * GC_F + GATES * GC_F);
    static bool attr_done = false;  // once per instantiation: keeps launches legal under hipGraph capture
    if (!attr_done) {
        MRX_HIP(hipFuncSetAttribute((const void*)k_gated_cell<GATES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        MRX_HIP(hipGetDevice(&dev));
        MRX_HIP(hipGetDeviceProperties(&prop, dev));
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const long long nblk_need = (a.nseg + GC_NT / 64 - 1) / (GC_NT / 64);
    const unsigned nblk = (unsigned)(nblk_need < n_cu ? nblk_need : n_cu);  // persistent: the weights are staged once per workgroup
    hipLaunchKernelGGL((k_gated_cell<GATES>), dim3(nblk), dim3(GC_NT), lds, st, a);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

static thread_local float* g_gated_xmax = nullptr;   // set by mrx_gated_cell_1x1_xmax around the call
extern "C" int mrx_gated_cell_1x1(const float* x, const float* h, const float* packed, const float* b_ih, float* out, int B,
                                  int Cin, int F, int64_t HW, int gates, void* stream) {
    MRX_REQUIRE(x && packed && out, MRX_EINVAL, "mrx_gated_cell_1x1: null pointer");
    MRX_REQUIRE(B >= 0 && HW >= 0 && HW < (1ll << 24) && B < (1 << 30), MRX_EINVAL, "mrx_gated_cell_1x1: bad dims");
    MRX_REQUIRE(mrx_gated_cell_supported(Cin, F, 1, gates), MRX_EUNSUP,
                "mrx_gated_cell_1x1: Cin=%d F=%d gates=%d (64/64 and 2|3 gates only; use conv2d + mrx_gru_gates)", Cin, F, gates);
    MRX_REQUIRE(out != x && out != h, MRX_EINVAL, "mrx_gated_cell_1x1: out must not alias an input");
    if (B == 0 || HW == 0) return MRX_OK;
    GatedArgs a;
    a.x = x;
    a.h = h;
    a.packed = packed;
    a.b_ih = b_ih;
    a.out = out;
    a.P = HW;
    a.nsegb = (HW + 31) / 32;
    a.nseg = a.nsegb * B;
    const int fp32 = mrx_arith() == MRX_ARITH_FP32 ? 1 : 0;   // 1: the fp32-MFMA kernel (cross-check)
    if (!fp32) {                                  // default: the bf16 matrix pipe with fp32 results (gated_cell_sb.hip)
        MrxGatedSbArgs s;
        s.x = x, s.h = h, s.packed = packed + (size_t)2 * gates * GC_F * GC_F, s.b_ih = b_ih, s.out = out;
        s.P = a.P, s.nsegb = a.nsegb, s.nseg = a.nseg, s.xmax = g_gated_xmax;
        return mrx_gated_sb_launch(s, gates, (hipStream_t)stream);
    }
    MRX_REQUIRE(!g_gated_xmax, MRX_EUNSUP, "mrx_gated_cell_1x1_xmax: only the matrix-pipe kernel keeps the bound of its outputs");
    return gates == 3 ? launch_gated<3>(a, (hipStream_t)stream) : launch_gated<2>(a, (hipStream_t)stream);
}

// mrx_gated_cell_1x1 that also folds max |out| into the device scalar *xmax (atomic max; the caller zeroes it): the operand bound of a following
// 64-channel convolution on two-term fp16 operands (mrx_conv3x3_sb_chain).  MRIDC_AMD_ARITH != fp32.
extern "C" int mrx_gated_cell_1x1_xmax(const float* x, const float* h, const float* packed, const float* b_ih, float* out, float* xmax, int B,
                                       int Cin, int F, int64_t HW, int gates, void* stream) {
    MRX_REQUIRE(xmax, MRX_EINVAL, "mrx_gated_cell_1x1_xmax: null pointer");
    MRX_REQUIRE(mrx_arith() != MRX_ARITH_FP32, MRX_EUNSUP, "mrx_gated_cell_1x1_xmax: only the matrix-pipe kernel keeps the bound of its outputs");
    g_gated_xmax = xmax;
    const int rc = mrx_gated_cell_1x1(x, h, packed, b_ih, out, B, Cin, F, HW, gates, stream);
    g_gated_xmax = nullptr;
    return rc;
}

// ---- Conv2dGRU layer of the Recurrent Variational Network (recurrentvarnet/conv2gru.py:139-157), 1x1 gates on 64 features ------
//   update = sigmoid(Wu [x; h] + bu)   reset = sigmoid(Wr [x; h] + br)   delta = tanh(Wo [x; h * reset] + bo)
//   h_new  = h * (1 - update) + delta * update          (also written as ReLU(h_new): the next layer's input, :157)
// Unlike the RIM's ConvGRUCell the reset gate multiplies the state BEFORE the candidate's GEMM.  h * reset is formed in the
// accumulator layout (reset lives there) and fed to that GEMM as it is: an accumulator register of the 32x32 MFMA holds one
// channel per lane half, which is exactly a B operand, so the candidate's hh weights are packed with their contraction index
// enumerated in accumulator order (channel of step t, lane half l: 32(t>>4) + (t&3) + 8((t&15)>>2) + 4l) -- the same trick the
// fused RIM layer uses for its ih GEMM.  Six 64x64 GEMMs per 32 pixels, all operands from HBM/LDS once.
struct Conv2dGruArgs {
    const float* x;       // [B,64,P] layer input (after its conv + ReLU)
    const float* h;       // [B,64,P] previous state of this layer or null (= zeros)
    const float* packed;  // mrx_conv2dgru_pack
    const float* bias;    // [3][64]: update, reset, out
    float* out;           // [B,64,P] new state
    float* out_relu;      // [B,64,P] ReLU(new state) or null
    long long P, nsegb, nseg;
};

// mats 0..2: Wu, Wr, Wo columns 0..63 (x part); 3, 4: Wu, Wr columns 64..127 (h part); 5: Wo columns 64..127 in accumulator order
__global__ void k_conv2dgru_pack(const float* __restrict__ wu, const float* __restrict__ wr, const float* __restrict__ wo,
                                 float* __restrict__ out) {
    const int total = 6 * GC_F * GC_F;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int m = i & 31, half = (i >> 5) & 1, s = (i >> 6) & 31, mb = (i >> 11) & 1, mat = i >> 12;
        const int row = mb * 32 + m;
        const float* w = (mat == 0 || mat == 3) ? wu : (mat == 1 || mat == 4) ? wr : wo;
        int col;
        if (mat < 3)
            col = 2 * s + half;
        else if (mat < 5)
            col = GC_F + 2 * s + half;
        else
            col = GC_F + 32 * (s >> 4) + (s & 3) + 8 * ((s & 15) >> 2) + 4 * half;
        out[i] = w[(long long)row * (2 * GC_F) + col];
    }
}

__global__ __launch_bounds__(GC_NT, 2) void k_conv2dgru_cell(Conv2dGruArgs a) {
    extern __shared__ __attribute__((aligned(16))) float Ws[];  // [6][2][32][2][32], then the biases [3][64]
    constexpr int MATF = GC_F * GC_F;
    const int tid = threadIdx.x;
    float* Bs = Ws + 6 * MATF;
    {
        const float4* src = reinterpret_cast<const float4*>(a.packed);
        float4* dst = reinterpret_cast<float4*>(Ws);
        for (int i = tid; i < 6 * MATF / 4; i += GC_NT) dst[i] = src[i];
        if (tid < 3 * GC_F) Bs[tid] = a.bias ? a.bias[tid] : 0.f;
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const float* wl = Ws + lane;
    const long long stride = (long long)gridDim.x * (GC_NT / 64);
    const unsigned P32 = (unsigned)a.P;
    float xg[32], hg[32];
    const float* xb = nullptr;
    const float* hb = nullptr;
    long long base = 0;
    unsigned pxo = 0;
    bool valid = false;
    int lhi = 0;
    auto load = [&](long long sg) {  // see k_gated_cell
        int l31 = lane & 31;
        lhi = lane >> 5;
        asm volatile("" : "+v"(l31), "+v"(lhi));
        const long long b = sg / a.nsegb;
        const long long px = (sg - b * a.nsegb) * 32 + l31;
        valid = px < a.P;
        base = __builtin_amdgcn_readfirstlane((int)b) * (long long)GC_F * a.P;
        pxo = valid ? (unsigned)px : 0u;
        xb = a.x + base;
        hb = a.h ? a.h + base : nullptr;
#pragma unroll
        for (int s = 0; s < 32; ++s) xg[s] = xb[(unsigned)(2 * s + lhi) * P32 + pxo];
    };
    auto load_h = [&]() {
        if (hb) {
#pragma unroll
            for (int s = 0; s < 32; ++s) hg[s] = hb[(unsigned)(2 * s + lhi) * P32 + pxo];
        }
    };
    // one GEMM group: NM matrices starting at MAT0, B operand SRC[s], destinations acc[D0 + mat]
#define GC_GEMM(NM, MAT0, SRCEXPR, D0)                                                                   \
    {                                                                                                    \
        constexpr int NS = (NM) * 32;                                                                    \
        float ra0[GC_PF + 1], ra1[GC_PF + 1];                                                            \
        _Pragma("unroll") for (int t = 0; t < NS + GC_PF; ++t) {                                         \
            if (t < NS) {                                                                                \
                const int mat = (MAT0) + (t >> 5), s = t & 31;                                           \
                ra0[t % (GC_PF + 1)] = wl[((mat * 2 + 0) * 32 + s) * 64];                                \
                ra1[t % (GC_PF + 1)] = wl[((mat * 2 + 1) * 32 + s) * 64];                                \
            }                                                                                            \
            if (t >= GC_PF) {                                                                            \
                const int u = t - GC_PF, c = u % (GC_PF + 1);                                            \
                const int g = (D0) + (u >> 5), s = u & 31;                                               \
                acc[g][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra0[c], SRCEXPR, acc[g][0], 0, 0, 0);   \
                acc[g][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra1[c], SRCEXPR, acc[g][1], 0, 0, 0);   \
            }                                                                                            \
            __builtin_amdgcn_sched_barrier(0);                                                           \
        }                                                                                                \
    }
    long long sg = (long long)blockIdx.x * (GC_NT / 64) + wave;
    if (sg < a.nseg) load(sg);
    while (sg < a.nseg) {
        load_h();
        f32x16 acc[3][2];  // 0 update, 1 reset (then h * reset), 2 candidate
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    acc[d][ct][r] = Bs[d * GC_F + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi];
        GC_GEMM(3, 0, xg[s], 0)
        if (hb) GC_GEMM(2, 3, hg[s], 0)
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
        if (have_h) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[1][ct][r] = hv[ct][r] * gc_sigmoid(acc[1][ct][r]);
            GC_GEMM(1, 5, acc[1][s >> 4][s & 15], 2)
        }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * o_lhi;
                const float u = gc_sigmoid(acc[0][ct][r]);
                const float dl = gc_tanh(acc[2][ct][r]);
                const float o = hv[ct][r] * (1.0f - u) + dl * u;
                if (o_valid) {
                    ob[(unsigned)co * P32 + o_pxo] = o;
                    if (orl) orl[(unsigned)co * P32 + o_pxo] = o > 0.f ? o : 0.f;
                }
            }
    }
#undef GC_GEMM
}

extern "C" int64_t mrx_conv2dgru_pack_floats(int F) {      // fp32 operands, then the split-bf16 ones (gated_cell_sb.hip)
    return F == GC_F ? (int64_t)6 * GC_F * GC_F + MRX_CONV2DGRU_SB_PACK_FLOATS : -1;
}
extern "C" int mrx_conv2dgru_supported(int Cin, int F, int k) { return Cin == GC_F && F == GC_F && k == 1; }

extern "C" int mrx_conv2dgru_pack(const float* w_update, const float* w_reset, const float* w_out, float* packed, int F, void* stream) {
    MRX_REQUIRE(w_update && w_reset && w_out && packed, MRX_EINVAL, "mrx_conv2dgru_pack: null pointer");
    MRX_REQUIRE(F == GC_F, MRX_EUNSUP, "mrx_conv2dgru_pack: hidden size %d (only %d)", F, GC_F);
    hipLaunchKernelGGL(k_conv2dgru_pack, dim3((6 * GC_F * GC_F + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_update, w_reset, w_out,
                       packed);
    MRX_LAUNCH_CHECK();
    return mrx_conv2dgru_sb_pack(w_update, w_reset, w_out, packed + 6 * GC_F * GC_F, (hipStream_t)stream);
}

static thread_local float* g_gru2d_xmax = nullptr;   // set by mrx_conv2dgru_cell_1x1_xmax around the call
extern "C" int mrx_conv2dgru_cell_1x1(const float* x, const float* h, const float* packed, const float* bias, float* out,
                                      float* out_relu, int B, int F, int64_t HW, void* stream) {
    MRX_REQUIRE(x && packed && out, MRX_EINVAL, "mrx_conv2dgru_cell_1x1: null pointer");
    MRX_REQUIRE(B >= 0 && HW >= 0 && HW < (1ll << 24) && B < (1 << 30), MRX_EINVAL, "mrx_conv2dgru_cell_1x1: bad dims");
    MRX_REQUIRE(F == GC_F, MRX_EUNSUP, "mrx_conv2dgru_cell_1x1: hidden size %d (only %d; use the unfused route)", F, GC_F);
    MRX_REQUIRE(out != x && out != h && out_relu != x && out_relu != h && out != out_relu, MRX_EINVAL,
                "mrx_conv2dgru_cell_1x1: outputs must not alias inputs or each other");
    if (B == 0 || HW == 0) return MRX_OK;
    Conv2dGruArgs a;
    a.x = x;
    a.h = h;
    a.packed = packed;
    a.bias = bias;
    a.out = out;
    a.out_relu = out_relu;
    a.P = HW;
    a.nsegb = (HW + 31) / 32;
    a.nseg = a.nsegb * B;
    const int fp32 = mrx_arith() == MRX_ARITH_FP32 ? 1 : 0;   // 1: the fp32-MFMA kernel (cross-check)
    if (!fp32) {                                  // default: the bf16 matrix pipe with fp32 results (gated_cell_sb.hip)
        MrxConv2dGruSbArgs s;
        s.x = x, s.h = h, s.packed = packed + 6 * GC_F * GC_F, s.bias = bias, s.out = out, s.out_relu = out_relu;
        s.P = a.P, s.nsegb = a.nsegb, s.nseg = a.nseg, s.xmax = g_gru2d_xmax;
        return mrx_conv2dgru_sb_launch(s, (hipStream_t)stream);
    }
    constexpr size_t lds = sizeof(float) * (6 * GC_F * GC_F + 3 * GC_F);
    static bool attr_done = false;
    if (!attr_done) {
        MRX_HIP(hipFuncSetAttribute((const void*)k_conv2dgru_cell, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        MRX_HIP(hipGetDevice(&dev));
        MRX_HIP(hipGetDeviceProperties(&prop, dev));
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const long long nblk_need = (a.nseg + GC_NT / 64 - 1) / (GC_NT / 64);
    const unsigned nblk = (unsigned)(nblk_need < n_cu ? nblk_need : n_cu);
    hipLaunchKernelGGL(k_conv2dgru_cell, dim3(nblk), dim3(GC_NT), lds, (hipStream_t)stream, a);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- 1x1 convolution C -> C, C = 64 or 128 (a per-pixel GEMM) with the IndRNN cell as an optional epilogue ------------------------------
//   out = act(W x + bias [+ hh * h_prev])      act: MRX_ACT_NONE / RELU / LEAKY
// The ih stage of the IndRNN cell run on its own (training forward: rnn_cells.py:384-391; the 128-feature cells of the qCIRIM), the
// data gradient of 1x1 layers (W^T as weights), RecurrentInit's 1x1 heads (recurrentvarnet.py:73-76).  Same skeleton as the gated
// cells: x straight from HBM into the B-operand layout (64 channels at a time), the packed weights (16 / 64 KB) in LDS for the life
// of the persistent workgroup, 64 MFMAs per 32 pixels and 64x64 block -- HBM-bound at C = 64 (two or three 61 MB tensors at 640x372).
struct Conv1x1Args {
    const float* x;       // [B,C,P]
    const float* packed;  // mrx_conv1x1_sq_pack
    const float* bias;    // [C] or null
    const float* hh;      // [C] or null (IndRNN: + hh * h_prev)
    const float* hprev;   // [B,C,P] or null
    float* out;           // [B,C,P]
    long long P, nsegb, nseg;
    int act;
    float slope;
};
// packed index ((((ob*CB + ib)*2 + mb)*32 + s)*2 + half)*32 + m  <-  W[ob*64 + mb*32 + m][ib*64 + 2*s + half]
__global__ void k_conv1x1_pack(const float* __restrict__ w, float* __restrict__ out, int C) {
    const int CB = C / GC_F;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < C * C; i += gridDim.x * blockDim.x) {
        const int m = i & 31, half = (i >> 5) & 1, s = (i >> 6) & 31, mb = (i >> 11) & 1, blk = i >> 12;
        const int ob = blk / CB, ib = blk - ob * CB;
        out[i] = w[(long long)(ob * GC_F + mb * 32 + m) * C + ib * GC_F + 2 * s + half];
    }
}
template <int CB>
__global__ __launch_bounds__(GC_NT, (CB == 1 ? 4 : 2)) void k_conv1x1_sq(Conv1x1Args a) {
    extern __shared__ __attribute__((aligned(16))) float Ws[];  // [CB][CB][2][32][2][32], then bias [C], hh [C]
    constexpr int C = CB * GC_F;
    const int tid = threadIdx.x;
    float* Bs = Ws + C * C;
    {
        const float4* src = reinterpret_cast<const float4*>(a.packed);
        float4* dst = reinterpret_cast<float4*>(Ws);
        for (int i = tid; i < C * C / 4; i += GC_NT) dst[i] = src[i];
        if (tid < C) {
            Bs[tid] = a.bias ? a.bias[tid] : 0.f;
            Bs[C + tid] = a.hh ? a.hh[tid] : 0.f;
        }
    }
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    const float* wl = Ws + lane;
    const long long stride = (long long)gridDim.x * (GC_NT / 64);
    const unsigned P32 = (unsigned)a.P;
    const float neg = a.act == MRX_ACT_RELU ? 0.f : (a.act == MRX_ACT_LEAKY ? a.slope : 1.f);
    for (long long sg = (long long)blockIdx.x * (GC_NT / 64) + wave; sg < a.nseg; sg += stride) {
        int l31 = lane & 31, lhi = lane >> 5;
        asm volatile("" : "+v"(l31), "+v"(lhi));  // keep the channel offsets out of loop-invariant hoisting (spills)
        const long long b = sg / a.nsegb;
        const long long px = (sg - b * a.nsegb) * 32 + l31;
        const bool valid = px < a.P;
        const long long base = __builtin_amdgcn_readfirstlane((int)b) * (long long)C * a.P;
        const unsigned pxo = valid ? (unsigned)px : 0u;
        const float* xb = a.x + base;
        f32x16 acc[CB][2];
#pragma unroll
        for (int ob = 0; ob < CB; ++ob)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ob][ct][r] = Bs[ob * GC_F + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi];
#pragma unroll
        for (int ib = 0; ib < CB; ++ib) {
            float xg[32];
#pragma unroll
            for (int s = 0; s < 32; ++s) xg[s] = xb[(unsigned)(ib * GC_F + 2 * s + lhi) * P32 + pxo];
#pragma unroll
            for (int ob = 0; ob < CB; ++ob) {
                const float* wb = wl + (ob * CB + ib) * (GC_F * GC_F);
                float ra0[GC_PF + 1], ra1[GC_PF + 1];
#pragma unroll
                for (int t = 0; t < 32 + GC_PF; ++t) {
                    if (t < 32) {
                        ra0[t % (GC_PF + 1)] = wb[(0 * 32 + t) * 64];
                        ra1[t % (GC_PF + 1)] = wb[(1 * 32 + t) * 64];
                    }
                    if (t >= GC_PF) {
                        const int u = t - GC_PF, c = u % (GC_PF + 1);
                        acc[ob][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra0[c], xg[u], acc[ob][0], 0, 0, 0);
                        acc[ob][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra1[c], xg[u], acc[ob][1], 0, 0, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        float* ob_ = a.out + base;
#pragma unroll
        for (int ob = 0; ob < CB; ++ob) {
            float hv[2][16];  // h_prev in accumulator layout, loaded after the GEMM (register budget)
            if (a.hprev) {
                const float* hb = a.hprev + base;
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        hv[ct][r] = hb[(unsigned)(ob * GC_F + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi) * P32 + pxo];
            }
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = ob * GC_F + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    float v = acc[ob][ct][r];
                    if (a.hprev) v += Bs[C + co] * hv[ct][r];
                    v = v > 0.f ? v : v * neg;
                    if (valid) ob_[(unsigned)co * P32 + pxo] = v;
                }
        }
    }
}
// mrx_conv2dgru_cell_1x1 that also folds the maximum of out_relu into the device scalar *xmax (atomic max; the caller zeroes it): the bound the next
// layer's 64-channel convolution scales its two-term fp16 operands by (mrx_conv3x3_sb_chain).  MRIDC_AMD_ARITH != fp32.
extern "C" int mrx_conv2dgru_cell_1x1_xmax(const float* x, const float* h, const float* packed, const float* bias, float* out, float* out_relu,
                                           float* xmax, int B, int F, int64_t HW, void* stream) {
    MRX_REQUIRE(xmax && out_relu, MRX_EINVAL, "mrx_conv2dgru_cell_1x1_xmax: null pointer");
    MRX_REQUIRE(mrx_arith() != MRX_ARITH_FP32, MRX_EUNSUP, "mrx_conv2dgru_cell_1x1_xmax: only the matrix-pipe kernel keeps the bound of its outputs");
    g_gru2d_xmax = xmax;
    const int rc = mrx_conv2dgru_cell_1x1(x, h, packed, bias, out, out_relu, B, F, HW, stream);
    g_gru2d_xmax = nullptr;
    return rc;
}

extern "C" int mrx_conv1x1_sq_supported(int Cin, int Cout) { return Cin == Cout && (Cin == 64 || Cin == 128); }
// floats of the operand pack: the fp32 section and, at C = 128, the split-bf16 one (gated_cell_sb.hip: the default kernel at that width)
extern "C" int64_t mrx_conv1x1_sq_pack_floats(int C) {
    if (!mrx_conv1x1_sq_supported(C, C)) return -1;
    return (int64_t)C * C + (C == 128 ? MRX_CONV1X1_SB128_PACK_FLOATS : 0);
}
extern "C" int mrx_conv1x1_sq_pack(const float* w, float* packed, int C, void* stream) {
    MRX_REQUIRE(w && packed, MRX_EINVAL, "mrx_conv1x1_sq_pack: null pointer");
    MRX_REQUIRE(mrx_conv1x1_sq_supported(C, C), MRX_EUNSUP, "mrx_conv1x1_sq_pack: C=%d (64 or 128)", C);
    hipLaunchKernelGGL(k_conv1x1_pack, dim3(C * C / 256), dim3(256), 0, (hipStream_t)stream, w, packed, C);
    MRX_LAUNCH_CHECK();
    if (C == 128) return mrx_conv1x1_sb128_pack(w, packed + C * C, (hipStream_t)stream);
    return MRX_OK;
}
static thread_local float* g_sq_xmax = nullptr;   // set by mrx_conv1x1_sq_xmax around the call
static thread_local int g_sq_p16 = 0;             // set by mrx_conv1x1_sq_p16 around the call
extern "C" int mrx_conv1x1_sq(const float* x, const float* packed, const float* bias, const float* hh, const float* h_prev, float* out,
                              int B, int C, int64_t HW, int act, float slope, void* stream) {
    MRX_REQUIRE(x && packed && out, MRX_EINVAL, "mrx_conv1x1_sq: null pointer");
    MRX_REQUIRE(mrx_conv1x1_sq_supported(C, C), MRX_EUNSUP, "mrx_conv1x1_sq: C=%d (64 or 128)", C);
    MRX_REQUIRE(B >= 0 && HW >= 0 && HW < (1ll << 24) && B < (1 << 30), MRX_EINVAL, "mrx_conv1x1_sq: bad dims");
    MRX_REQUIRE(!h_prev || hh, MRX_EINVAL, "mrx_conv1x1_sq: h_prev needs hh");
    MRX_REQUIRE(act == MRX_ACT_NONE || act == MRX_ACT_RELU || act == MRX_ACT_LEAKY, MRX_EINVAL, "mrx_conv1x1_sq: activation %d", act);
    MRX_REQUIRE(out != x && out != h_prev, MRX_EINVAL, "mrx_conv1x1_sq: out must not alias an input");
    if (B == 0 || HW == 0) return MRX_OK;
    Conv1x1Args a;
    a.x = x;
    a.packed = packed;
    a.bias = bias;
    a.hh = hh;
    a.hprev = h_prev;
    a.out = out;
    a.P = HW;
    a.nsegb = (HW + 31) / 32;
    a.nseg = a.nsegb * B;
    a.act = act;
    a.slope = slope;
    const int fp32 = mrx_arith() == MRX_ARITH_FP32 ? 1 : 0;   // 1: the fp32-MFMA kernel (cross-check)
    if (C == 128 && !fp32) {                      // default at 128 features: the bf16 matrix pipe with fp32 results
        MrxConv1x1SbArgs s;
        s.x = x, s.packed = packed + (size_t)C * C, s.bias = bias, s.hh = hh, s.hprev = h_prev, s.out = out;
        s.P = a.P, s.nsegb = a.nsegb, s.nseg = a.nseg, s.act = act, s.slope = slope, s.head = 0, s.xmax = g_sq_xmax, s.p16 = g_sq_p16;
        return mrx_conv1x1_sb128_launch(s, (hipStream_t)stream);
    }
    MRX_REQUIRE(!g_sq_p16, MRX_EUNSUP, "mrx_conv1x1_sq_p16: C = %d (the 128-channel matrix-pipe kernel only)", C);
    MRX_REQUIRE(!g_sq_xmax, MRX_EUNSUP, "mrx_conv1x1_sq_xmax: only the 128-channel matrix-pipe kernel keeps the bound of its outputs");
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        MRX_HIP(hipGetDevice(&dev));
        MRX_HIP(hipGetDeviceProperties(&prop, dev));
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const long long nblk_need = (a.nseg + GC_NT / 64 - 1) / (GC_NT / 64);
    const size_t lds = sizeof(float) * ((size_t)C * C + 2 * C);
    if (C == 64) {
        const long long cap = 2ll * n_cu;  // two persistent workgroups per CU (16 KB of LDS, <= 128 registers)
        hipLaunchKernelGGL(k_conv1x1_sq<1>, dim3((unsigned)(nblk_need < cap ? nblk_need : cap)), dim3(GC_NT), lds, (hipStream_t)stream, a);
    } else {
        static bool attr_done = false;
        if (!attr_done) {
            MRX_HIP(hipFuncSetAttribute((const void*)k_conv1x1_sq<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_done = true;
        }
        const long long cap = n_cu;
        hipLaunchKernelGGL(k_conv1x1_sq<2>, dim3((unsigned)(nblk_need < cap ? nblk_need : cap)), dim3(GC_NT), lds, (hipStream_t)stream, a);
    }
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// mrx_conv1x1_sq that also folds max |out| into the device scalar *xmax (atomic max, never reset here): the operand bound of a two-term fp16
// consumer (mrx_conv3x3_h) without a pass over the tensor.  C = 128, MRIDC_AMD_ARITH != fp32.
extern "C" int mrx_conv1x1_sq_xmax_supported(int C) { return C == 128 && mrx_arith() != MRX_ARITH_FP32; }
extern "C" int mrx_conv1x1_sq_xmax(const float* x, const float* packed, const float* bias, const float* hh, const float* h_prev, float* out,
                                   float* xmax, int B, int C, int64_t HW, int act, float slope, void* stream) {
    MRX_REQUIRE(xmax, MRX_EINVAL, "mrx_conv1x1_sq_xmax: null pointer");
    MRX_REQUIRE(mrx_conv1x1_sq_xmax_supported(C), MRX_EUNSUP, "mrx_conv1x1_sq_xmax: C = %d (128 on the matrix-pipe kernel only)", C);
    g_sq_xmax = xmax;
    const int rc = mrx_conv1x1_sq(x, packed, bias, hh, h_prev, out, B, C, HW, act, slope, stream);
    g_sq_xmax = nullptr;
    return rc;
}

// mrx_conv1x1_sq[_xmax] at C = 128 in the reference's `precision: 16` inference arithmetic (base_qcirim_run.yaml:204; rnn_cells.py:384-391 under torch.autocast(float16)):
// x and W rounded to fp16 once, fp32 sums; hh * h_prev, bias and activation in fp32.  xmax may be null.  MRIDC_AMD_ARITH != fp32.
extern "C" int mrx_conv1x1_sq_p16(const float* x, const float* packed, const float* bias, const float* hh, const float* h_prev, float* out,
                                  float* xmax, int B, int C, int64_t HW, int act, float slope, void* stream) {
    MRX_REQUIRE(mrx_conv1x1_sq_xmax_supported(C), MRX_EUNSUP, "mrx_conv1x1_sq_p16: C = %d (128 on the matrix-pipe kernel only)", C);
    g_sq_xmax = xmax, g_sq_p16 = 1;
    const int rc = mrx_conv1x1_sq(x, packed, bias, hh, h_prev, out, B, C, HW, act, slope, stream);
    g_sq_xmax = nullptr, g_sq_p16 = 0;
    return rc;
}

// out [B,64,P] = the first 64 rows of W (128 x 128) times x [B,128,P]: the channel contraction of a thin 3x3 convolution of 128 channels
// (9 Cout <= 64 tap rows, ops.conv3x3_taps) without the unused half of the outputs.  packed from mrx_conv1x1_sq_pack(w, ., 128).
extern "C" int mrx_conv1x1_sq_head128(const float* x, const float* packed, float* out, int B, int64_t HW, void* stream) {
    MRX_REQUIRE(x && packed && out, MRX_EINVAL, "mrx_conv1x1_sq_head128: null pointer");
    MRX_REQUIRE(B >= 0 && HW >= 0 && HW < (1ll << 24) && B < (1 << 30), MRX_EINVAL, "mrx_conv1x1_sq_head128: bad dims");
    if (B == 0 || HW == 0) return MRX_OK;
    MrxConv1x1SbArgs s;
    s.x = x, s.packed = packed + (size_t)128 * 128, s.bias = nullptr, s.hh = nullptr, s.hprev = nullptr, s.out = out;
    s.P = HW, s.nsegb = (HW + 31) / 32, s.nseg = s.nsegb * B, s.act = MRX_ACT_NONE, s.slope = 0.f, s.head = 1, s.xmax = nullptr, s.p16 = 0;
    return mrx_conv1x1_sb128_launch(s, (hipStream_t)stream);
}
// 64-channel forms kept as named entry points
extern "C" int mrx_conv1x1_64_pack(const float* w, float* packed, void* stream) { return mrx_conv1x1_sq_pack(w, packed, 64, stream); }
extern "C" int mrx_conv1x1_64(const float* x, const float* packed, const float* bias, const float* hh, const float* h_prev, float* out,
                              int B, int64_t HW, int act, float slope, void* stream) {
    return mrx_conv1x1_sq(x, packed, bias, hh, h_prev, out, B, 64, HW, act, slope, stream);
}
