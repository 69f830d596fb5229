// rim_amp16.hip -- the two RIM layers of a time-step (reference models/rim/rim_block.py:217-249: ConvNonlinear 5x5 4 -> 64 + IndRNNCell 1x1, ConvNonlinear
// 3x3 dilation 2 64 -> 64 + IndRNNCell 1x1, then the final 3x3 64 -> 2) in the arithmetic the reference's own INFERENCE configuration runs:
// `precision: 16` (projects/reconstruction/model_zoo/conf/base_cirim_run.yaml:132) = torch.autocast -- every convolution multiplies HALF-precision
// operands and accumulates wide, everything autocast does not list (FFT, the complex products of log_likelihood_gradient, eta) stays fp32.
//
// This is the reduced-precision route of the library (RIMBlock.precision = 16 / MRIDC_AMD_PRECISION=16), never the default: ONE fp16 term per operand (one MFMA per product where the
// fp32-class route issues three), fp32 accumulation, and the hidden states kept in fp16, channel-blocked [B][4][H][W][16] halves (32 bytes per pixel and
// channel block = two B operands of v_mfma_f32_32x32x16_f16; 1 KB of one plane per wave instruction).  With a third of the matrix work and half the state bytes both layers are HBM-bound:
// the kernels are built around bytes in flight, not around MFMA issue.
//
//   k_amp_layer1_t: 8 waves per CU on a 16 x 32 tile, wave = two rows x 64 couts: the halo'd 20 x 36 input patch (eta + the coil-group partial sums of the
//                 gradient launch, or a 4-channel x) of the NEXT tile and its h_prev are in flight during the whole of the current tile; 7 + 4 MFMA steps x 2 cout
//                 blocks per row, 8-byte state accesses through buffer descriptors.
//   k_amp_layer2: 8 waves per CU on a 16 x 32 tile, wave = two rows x 64 couts.  ALL weights of the layer stay in LDS for the life of the workgroup
//                 (conv 72 KB + 1x1 8 KB + final conv 4 KB); the halo'd fp16 tile arrives two channel chunks (16 channels, 23 KB) at a time through a
//                 two-deep register stage + two LDS buffers -- no conversion, no split: the global bytes ARE the B operands.  Nine MFMA steps per chunk
//                 pair (taps (0,1) .. (6,7) of each chunk, the two ninth taps together).  ReLU(conv + b) -> fp16 -> 1x1 from registers; the new state is
//                 rounded to fp16 once, stored, and the same packed registers are the operands of the final convolution's channel contraction, which
//                 leaves the row-pre-summed tap planes of mrx_rim_layer2_f16_cb8_q (fp32; gathered by mrx_llg372_gather_q / mrx_rim_final_gather_q).
#include <cstdlib>

#include "mrx_common.h"

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define AM_F 64
#define AM_TW 32
#define AM_TH 16

// PROBE builds (-DMRX_PROBE, env MRX_AMP_ABL): phases switched off to price them -- 1 no input (patch / x) loads, 2 no h_prev loads, 4 no state stores,
// 8 no tap stage, 16 no convolution MFMAs, 32 tap stage without its stores (64 / 128 / 256: without the 8-byte / the 4-byte / the edge store only).  Results are garbage; only the time is read.  The product build compiles every test away.
#ifdef MRX_PROBE
#define AM_ABL(a, bit) (((a).abl & (bit)) != 0)
#else
#define AM_ABL(a, bit) false
#endif
// cache policy of the state streams (aux bits of the buffer / global instructions; 2 = nt): every state byte is written once and read once, a whole time-step
// (~1 GB at 8 slices per launch) later -- MRX_AMP_NT_ST / MRX_AMP_NT_LD select the streaming policy for the h_new stores / the h_prev loads (A/B builds)
// (measured, 8 slices per launch, profiles/r06_amp16_layer_times_v1.txt: default policy 19.9 / 28.1 us per slice for layer 1 / 2, nt stores 19.1 / 27.8, nt stores
// and nt h_prev loads 18.4 / 26.8: both on)
#ifndef MRX_AMP_NT_ST
#define MRX_AMP_NT_ST 1
#endif
#ifndef MRX_AMP_NT_LD
#define MRX_AMP_NT_LD 1
#endif
// ... layer 1's h_new stores on their own (layer 2 of the same step reads them back in the next launch: 244 MB at 8 slices per launch, about the size of the cache)
#ifndef MRX_AMP_NT_ST1
#define MRX_AMP_NT_ST1 MRX_AMP_NT_ST
#endif
// layer 2: the chunk pair at whose start a tile's h_prev is requested (consumed by the epilogue behind pair 3; no h_prev loads at all measured 6 us per slice faster
// than requesting them at pair 3: their latency was exposed)
#ifndef MRX_AMP_HP_AT
#define MRX_AMP_HP_AT 1
#endif

// channel of accumulator register R = 16 ct + r in lane half `half` (v_mfma_f32_32x32 C/D layout)
__host__ __device__ constexpr int am_chan(int R, int half) { return 32 * (R >> 4) + (R & 3) + 8 * ((R & 15) >> 2) + 4 * half; }
__device__ __forceinline__ unsigned am_pk(float lo, float hi) {          // two fp32 -> packed fp16 pair (lo in bits 0-15), round to nearest even
    const f16x2 h = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(unsigned, h);
}
__device__ __forceinline__ float am_lo(unsigned p) { return (float)__builtin_bit_cast(f16x2, p).x; }
__device__ __forceinline__ float am_hi(unsigned p) { return (float)__builtin_bit_cast(f16x2, p).y; }
__device__ __forceinline__ float am_pow2(int e) {
    e = e < -120 ? -120 : (e > 120 ? 120 : e);
    return __uint_as_float((unsigned)(127 + e) << 23);
}
// exponent k with m * 2^k in [2^14, 2^15) (0 for zero / non-finite m)
__device__ __forceinline__ int am_scale_exp(float m) {
    const int ex = (int)((__float_as_uint(m) >> 23) & 0xffu);
    return (ex == 0 || ex == 255) ? 0 : 14 - (ex - 127);
}

// ---- the hidden-state layout: h[b][c / 16][y][x][c % 16] halves ("CB16", 32 bytes per pixel and block) ---------------------------------------------------------
// A lane of the accumulator layout (pixel n = lane % 32, half = lane / 32) owns channels 8 j + 4 half .. + 3 of every 8-channel group j (one u32x2 of packed halves
// per j).  In memory a pixel's 16 channels of a block are contiguous, so that ONE wave instruction moves 1 KB (32 pixels x 32 bytes) of one plane: the two lanes of a
// pixel trade halves first -- v_permlane32_swap on (group 2 m, group 2 m + 1): afterwards the lower lane holds channels 16 m .. 16 m + 7 and the upper lane 16 m + 8 ..
// 16 m + 15, 16 bytes each.  The swap is its own inverse (loads: swap after the data has arrived).  Measured on a pure read-modify-write stream of the state
// (tools/probe/state_stream_probe.hip, 8 slices, in place): 6.38 TB/s with these 1 KB pieces against 5.84 with the 512-byte pieces of [b][c / 8][y][x][c % 8].
__device__ __forceinline__ void am_swap32(unsigned& a, unsigned& b) {
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);       // the upper 32 lanes of `a` and the lower 32 lanes of `b` change places
    a = r[0], b = r[1];
}
// registers (h[j]: group j of this lane) <-> pieces (p[m]: this lane's 16 bytes of block m)
__device__ __forceinline__ void am_to_pieces(const u32x2 (&h)[8], u32x4 (&p)[4]) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        unsigned ax = h[2 * m].x, ay = h[2 * m].y, bx = h[2 * m + 1].x, by = h[2 * m + 1].y;
        am_swap32(ax, bx);
        am_swap32(ay, by);
        p[m] = u32x4{ax, ay, bx, by};
    }
}
__device__ __forceinline__ void am_from_pieces(const u32x4 (&p)[4], u32x2 (&h)[8]) {
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        unsigned ax = p[m].x, ay = p[m].y, bx = p[m].z, by = p[m].w;
        am_swap32(ax, bx);
        am_swap32(ay, by);
        h[2 * m] = u32x2{ax, ay}, h[2 * m + 1] = u32x2{bx, by};
    }
}
// byte offset of this lane's piece of block 0 at pixel index `pix` (block m: + m * plane * 32)
__device__ __forceinline__ unsigned am_piece_off(long long pix, int lhi) { return (unsigned)(pix * 32 + 16 * lhi); }

// ---- layer 1 ---------------------------------------------------------------------------------------------------------------------------------------
#define A1_K 5
#define A1_PAD 2
#define A1_PW (AM_TW + 2 * A1_PAD)
#define A1_KS 7                             // conv MFMA steps: 28 taps x 4 channels / 16
#define A1_KS2 4
#define A1_WCONV (A1_KS * 2 * 64)
#define A1_WIH (A1_KS2 * 2 * 64)
#define A1_PACK_U4 (A1_WCONV + A1_WIH)      // 1408 16-byte operands
#define A1_NW 16

// conv: out[(s*2 + blk)*64 + lane][j] = fp16( w[32 blk + lane%32][ci = j & 3][tap = 4 s + 2 (lane/32) + (j >> 2)] )   (0 for tap >= 25, ci >= Cin)
// ih  : out[A1_WCONV + (s*2 + blk)*64 + lane][j] = fp16( w_ih[32 blk + lane%32][am_chan(8 s + j, lane/32)] )
__global__ void k_amp1_pack(const float* __restrict__ w, const float* __restrict__ w_ih, u32x4* __restrict__ out, int Cin) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A1_PACK_U4) return;
    const bool conv = i < A1_WCONV;
    int r = conv ? i : i - A1_WCONV;
    const int lane = r & 63;
    r >>= 6;
    const int blk = r & 1, s = r >> 1;
    const int o = 32 * blk + (lane & 31), half = lane >> 5;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (conv) {
            const int ci = j & 3, tap = 4 * s + 2 * half + (j >> 2);
            v[j] = (tap < A1_K * A1_K && ci < Cin) ? w[((long long)o * Cin + ci) * (A1_K * A1_K) + tap] : 0.f;
        } else
            v[j] = w_ih[o * AM_F + am_chan(8 * s + j, half)];
    }
    out[i] = u32x4{am_pk(v[0], v[1]), am_pk(v[2], v[3]), am_pk(v[4], v[5]), am_pk(v[6], v[7])};
}

struct Amp1Args {
    const float* x;        // [B,Cin,H,W], Cin <= 4 (LLGT = 0)
    const float2* eta2;    // [B,H,W] complex (LLGT = 1): input = (eta, post * sum_k part_k) -- the gradient's last pass done by the patch loader
    const float2* part;    // [nparts <= 4][B][H][W] complex
    long long part_stride;
    int nparts;
    float post;
    const u32x4* packed;   // k_amp1_pack
    const float* b_conv;   // [64] or null
    const float* b_ih;     // [64] or null
    const float* hh;       // [64]
    const _Float16* hprev; // [B][4][H][W][16] or null (the zero state)
    _Float16* hnew;        // [B][4][H][W][16]
    int B, Cin, H, W, tiles_x, ntiles;
    int abl;               // probe builds only
};

// ---- layer 1: workgroup-tile form ------------------------------------------------------------------------------------------------------------------
// The first form of this layer was rim_layer1_sb.hip's: sixteen waves per CU, each walking its own image-row units start to finish.  Its phases ran one after the
// other in every wave (patch loads, arithmetic, stores: the phase ablation was additive -- 6 + 5 + 4.5 + 7.6 us per slice for arithmetic / patch / h_prev / stores,
// profiles/r06_amp16_layer_times_v1.txt -- the sixteen waves of a CU fall into step): 18.6 us per slice against 16.6 for this one on the same box
// (profiles/r06_amp16_layer1_tile_form.txt), whose own ablation is what a bandwidth-bound kernel's looks like (time falls with the bytes removed).  Here a workgroup of
// eight waves walks 16 x 32 tiles like layer 2: the halo'd 20 x 36 input patch of the NEXT tile and its h_prev are in flight (registers) during the whole of the
// current tile, the stores of the current tile leave behind them -- the memory system always has a tile's worth of requests queued.  The patch sits in LDS as fp32
// [pixel][4 channels] (read amplification 1.4 x instead of 5.6 x); the tile's power-of-two input scale comes from the waves' maxima left beside it, and the fp16
// operands are formed by the wave that multiplies them.  One barrier per tile.
#define A1T_NT 512
#define A1T_PH (AM_TH + 2 * A1_PAD)         // 20
#define A1T_NPIX (A1T_PH * A1_PW)           // 720
#define A1T_XV 2                            // patch pixels per thread
#define A1T_OFF_TAB (A1_PACK_U4 * 16)
#define A1T_OFF_MAX (A1T_OFF_TAB + 768)     // [2][8] floats
#define A1T_OFF_X (A1T_OFF_MAX + 64)        // [2][A1T_NPIX] float4
#define A1T_LDS (A1T_OFF_X + 2 * A1T_NPIX * 16)

template <int LLGT>
__global__ __launch_bounds__(A1T_NT, 1) void k_amp_layer1_t(Amp1Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_a1t[];
    u32x4* Wl = reinterpret_cast<u32x4*>(smem_a1t);
    float* tabl = reinterpret_cast<float*>(smem_a1t + A1T_OFF_TAB);
    float* tmax = reinterpret_cast<float*>(smem_a1t + A1T_OFF_MAX);
    float4* Xf = reinterpret_cast<float4*>(smem_a1t + A1T_OFF_X);
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const long long plane = (long long)a.H * a.W;
    const int total = a.ntiles * a.B;

    for (int i = tid; i < A1_PACK_U4; i += A1T_NT) Wl[i] = a.packed[i];
    if (tid < 64) {
        const int tc = am_chan(tid >> 1, tid & 1);
        const int ti = (tid & 1) * 32 + (tid >> 1);
        tabl[ti] = a.hh[tc];
        tabl[64 + ti] = a.b_conv ? a.b_conv[tc] : 0.f;
        tabl[128 + ti] = a.b_ih ? a.b_ih[tc] : 0.f;
    }

    constexpr int NRAW = LLGT > 0 ? 10 : 4;
    auto tile_of = [&](int t, int& b, int& h0, int& w0) {
        const int tq = t < total ? t : total - 1;           // (beyond the last tile the pipeline keeps requesting the last tile: in range, never used)
        const int tt = (int)mrx_xcd_band(tq, total);
        b = tt / a.ntiles;
        const int tile = tt - b * a.ntiles, ty0 = tile / a.tiles_x;
        h0 = ty0 * AM_TH, w0 = (tile - ty0 * a.tiles_x) * AM_TW;
    };
    auto request_patch = [&](int t, float (&raw)[A1T_XV][NRAW]) {
        int b, h0, w0;
        tile_of(t, b, h0, w0);
#pragma unroll
        for (int v = 0; v < A1T_XV; ++v) {
            int p = tid + v * A1T_NT;
            p = p < A1T_NPIX ? p : A1T_NPIX - 1;
            const int ty = p / A1_PW, tx = p - ty * A1_PW;
            int gy = h0 + ty - A1_PAD, gx = w0 + tx - A1_PAD;            // replicate border = clamp (conv_layers.py:72-76)
            gy = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy);
            gx = gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx);
            const unsigned off = (unsigned)(gy * a.W + gx);
            if (AM_ABL(a, 1)) {
#pragma unroll
                for (int k = 0; k < NRAW; ++k) raw[v][k] = (float)(off & 7u);
            } else if constexpr (LLGT > 0) {
                const float2 e = a.eta2[(long long)b * plane + off];
                raw[v][0] = e.x, raw[v][1] = e.y;
                const float2* pp = a.part + (long long)b * plane;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float2 u = pp[(long long)(k < a.nparts ? k : 0) * a.part_stride + off];
                    raw[v][2 + 2 * k] = u.x, raw[v][3 + 2 * k] = u.y;
                }
            } else {
                const float* xb = a.x + (long long)b * a.Cin * plane;
#pragma unroll
                for (int c = 0; c < 4; ++c) raw[v][c] = xb[(c < a.Cin ? c * plane : 0) + off];
            }
        }
    };
    auto request_hp = [&](int t, u32x4 (&hp)[2][4]) {          // this lane's four 16-byte pieces per row (am_from_pieces when they are consumed)
        int b, h0, w0;
        tile_of(t, b, h0, w0);
        const int ox = w0 + l31, cx = ox < a.W ? ox : a.W - 1;
        // (buffer descriptor + 32-bit offsets: sixteen 64-bit addresses per lane were what the register allocator spilled)
        const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.hprev ? a.hprev : a.hnew) + (long long)b * AM_F * plane, 0,
                                                                            (unsigned)(plane * (AM_F * 2)), 0x00020000);
#pragma unroll
        for (int rw = 0; rw < 2; ++rw) {
            const int oy = h0 + 2 * wave + rw, cy = oy < a.H ? oy : a.H - 1;
            const unsigned off = (a.hprev && !AM_ABL(a, 2)) ? am_piece_off((long long)cy * a.W + cx, lhi) : 0x80000000u;   // (the zero state: out of range reads 0)
#pragma unroll
            for (int m = 0; m < 4; ++m)
                hp[rw][m] = __builtin_amdgcn_raw_buffer_load_b128(rp, off, (unsigned)m * (unsigned)(plane * 32), MRX_AMP_NT_LD ? 2 : 0);
        }
    };
    // finish the patch (the last step of log_likelihood_gradient, rim_utils.py:61-67: same order of additions as the wave-private form), leave it in LDS as
    // fp32 and this wave's largest |input| beside it
    auto commit = [&](const float (&raw)[A1T_XV][NRAW], int buf) {
        float m = 0.f;
#pragma unroll
        for (int v = 0; v < A1T_XV; ++v) {
            float c0, c1, c2, c3;
            if constexpr (LLGT > 0) {
                float sx = raw[v][2], sy = raw[v][3];
#pragma unroll
                for (int k = 1; k < 4; ++k)
                    if (k < a.nparts) sx += raw[v][2 + 2 * k], sy += raw[v][3 + 2 * k];
                c0 = raw[v][0], c1 = raw[v][1], c2 = sx * a.post, c3 = sy * a.post;
            } else {
                c0 = raw[v][0];
                c1 = a.Cin > 1 ? raw[v][1] : 0.f;
                c2 = a.Cin > 2 ? raw[v][2] : 0.f;
                c3 = a.Cin > 3 ? raw[v][3] : 0.f;
            }
            const int p = tid + v * A1T_NT;
            if (p < A1T_NPIX) {
                Xf[buf * A1T_NPIX + p] = make_float4(c0, c1, c2, c3);
                m = fmaxf(m, fmaxf(fmaxf(fabsf(c0), fabsf(c1)), fmaxf(fabsf(c2), fabsf(c3))));
            }
        }
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (lane == 0) tmax[buf * 8 + wave] = m;
    };

    float raw[A1T_XV][NRAW];
    u32x4 hpA[2][4], hpB[2][4];
    request_patch(blockIdx.x, raw);
    request_hp(blockIdx.x, hpA);
    commit(raw, 0);
    request_patch(blockIdx.x + gridDim.x, raw);
    __syncthreads();

    auto body = [&](int it, int t, u32x4 (&hp_cur)[2][4], u32x4 (&hp_nxt)[2][4]) {
        int b, h0, w0;
        tile_of(t, b, h0, w0);
        const int buf = it & 1;
        float m = tmax[buf * 8];
#pragma unroll
        for (int k = 1; k < 8; ++k) m = fmaxf(m, tmax[buf * 8 + k]);
        const int kx = am_scale_exp(m);
        const float sxu = am_pow2(kx), unx = am_pow2(-kx);
        request_hp(t + gridDim.x, hp_nxt);               // the next tile's state: in flight during the whole of this tile
        __builtin_amdgcn_sched_barrier(0);

        // ---- conv 5x5, both rows of the wave ----------------------------------------------------------------------------------------------------------
        f32x16 acc[2][2];
#pragma unroll
        for (int rw = 0; rw < 2; ++rw)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rw][ct][r] = 0.f;
        {
            const float4* xw = Xf + buf * A1T_NPIX + (2 * wave) * A1_PW + l31;
            const u32x4* wl = Wl + lane;
#pragma unroll
            for (int s = 0; s < A1_KS; ++s) {
                auto toff = [](int tp) { return tp < A1_K * A1_K ? (tp / A1_K) * A1_PW + (tp % A1_K) : 0; };   // zero-weight taps read pixel 0 of the wave's rows
                const int offA = lhi ? toff(4 * s + 2) : toff(4 * s), offB = lhi ? toff(4 * s + 3) : toff(4 * s + 1);
                const f16x8 a0 = __builtin_bit_cast(f16x8, wl[(s * 2 + 0) * 64]), a1 = __builtin_bit_cast(f16x8, wl[(s * 2 + 1) * 64]);
#pragma unroll
                for (int rw = 0; rw < 2; ++rw) {
                    const float4 lo = xw[rw * A1_PW + offA], hi = xw[rw * A1_PW + offB];
                    const f16x8 bt = __builtin_bit_cast(f16x8, (u32x4{am_pk(lo.x * sxu, lo.y * sxu), am_pk(lo.z * sxu, lo.w * sxu),
                                                                      am_pk(hi.x * sxu, hi.y * sxu), am_pk(hi.z * sxu, hi.w * sxu)}));
                    acc[rw][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bt, acc[rw][0], 0, 0, 0);
                    acc[rw][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bt, acc[rw][1], 0, 0, 0);
                }
            }
        }
        // ---- per row: g = ReLU(conv + b) -> fp16, 1x1 from registers, h = ReLU(W_ih g + b_ih + hh * h_prev) -> fp16 ------------------------------------------
        const int ox = w0 + l31;
        const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(a.hnew + (long long)b * AM_F * plane, 0, (unsigned)(plane * (AM_F * 2)), 0x00020000);
#pragma unroll
        for (int rw = 0; rw < 2; ++rw) {
            f32x16 acc2[2];
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[ct][r] = tabl[128 + lhi * 32 + ct * 16 + r];
            const u32x4* wl = Wl + A1_WCONV + lane;
#pragma unroll
            for (int s = 0; s < A1_KS2; ++s) {
                unsigned g[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int R0 = 8 * s + 2 * q, R1 = R0 + 1;
                    float v0 = acc[rw][R0 >> 4][R0 & 15] * unx + tabl[64 + lhi * 32 + R0], v1 = acc[rw][R1 >> 4][R1 & 15] * unx + tabl[64 + lhi * 32 + R1];
                    v0 = v0 > 0.f ? v0 : 0.f;
                    v1 = v1 > 0.f ? v1 : 0.f;
                    g[q] = am_pk(v0, v1);
                }
                const f16x8 bt = __builtin_bit_cast(f16x8, (u32x4{g[0], g[1], g[2], g[3]}));
#pragma unroll
                for (int ct = 0; ct < 2; ++ct)
                    acc2[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wl[(s * 2 + ct) * 64]), bt, acc2[ct], 0, 0, 0);
            }
            const int oy = h0 + 2 * wave + rw;
            const unsigned offh = (oy < a.H && ox < a.W && !AM_ABL(a, 4)) ? am_piece_off((long long)oy * a.W + ox, lhi) : 0x80000000u;
            u32x2 hprev[8], hnew[8];
            am_from_pieces(hp_cur[rw], hprev);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float hv[4] = {am_lo(hprev[q].x), am_hi(hprev[q].x), am_lo(hprev[q].y), am_hi(hprev[q].y)};
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int R = 4 * q + i;
                    v[i] = acc2[R >> 4][R & 15] + tabl[lhi * 32 + R] * hv[i];
                    v[i] = v[i] > 0.f ? v[i] : 0.f;
                }
                hnew[q] = u32x2{am_pk(v[0], v[1]), am_pk(v[2], v[3])};
            }
            u32x4 pc[4];
            am_to_pieces(hnew, pc);
#pragma unroll
            for (int m = 0; m < 4; ++m) __builtin_amdgcn_raw_buffer_store_b128(pc[m], rh, offh + (unsigned)m * (unsigned)(plane * 32), 0, MRX_AMP_NT_ST1 ? 2 : 0);
        }
        // the next tile's patch (requested a tile ago) into the other buffer, the one after it requested
        commit(raw, buf ^ 1);
        request_patch(t + 2 * gridDim.x, raw);
        __syncthreads();
    };
    int it = 0;
    for (int t = blockIdx.x; t < total; t += 2 * gridDim.x, it += 2) {
        body(it, t, hpA, hpB);
        if (t + (int)gridDim.x < total) body(it + 1, t + gridDim.x, hpB, hpA);
    }
}

// ---- layer 2 ---------------------------------------------------------------------------------------------------------------------------------------
#define A2_NT 512
#define A2_DIL 2
#define A2_PH (AM_TH + 2 * A2_DIL)          // 20
#define A2_PW (AM_TW + 2 * A2_DIL)          // 36
#define A2_NPIX (A2_PH * A2_PW)             // 720
#define A2_JOB (2 * A2_NPIX)                // 16-byte elements of a chunk pair: [chunk in pair][pixel]
#define A2_XBUF (A2_JOB + 1)                // + the dummy slot threads without a third element write
#define A2_XV 3                             // elements per thread and job
#define A2_WCONV (4 * 9 * 2 * 64)           // [pair][step][cout block][lane]
#define A2_WIH (4 * 2 * 64)
#define A2_WP (4 * 64)
#define A2_PACK_U4 (A2_WCONV + A2_WIH + A2_WP)   // 5376 16-byte operands (84 KB)
#define A2_OFF_TAB (A2_PACK_U4 * 16)
#define A2_OFF_X (A2_OFF_TAB + 1024)
#define A2_LDS (A2_OFF_X + 2 * A2_XBUF * 16)

// conv : out[((p*9 + st)*2 + blk)*64 + lane][j] = fp16( w[32 blk + lane%32][8 q + j][tap] ),  st < 8: q = 2 p + st/4, tap = 2 (st%4) + lane/32;
//                                                                                             st = 8: q = 2 p + lane/32, tap = 8
// ih   : out[A2_WCONV + (s*2 + blk)*64 + lane][j] = fp16( w_ih[32 blk + lane%32][am_chan(8 s + j, lane/32)] )
// final: out[A2_WCONV + A2_WIH + s*64 + lane][j]  = fp16( w_final[m & 1][am_chan(8 s + j, lane/32)][tap = m >> 1] ), m = lane%32 < 18 (0 otherwise)
__global__ void k_amp2_pack(const float* __restrict__ w, const float* __restrict__ w_ih, const float* __restrict__ w_final, u32x4* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= A2_PACK_U4) return;
    float v[8];
    if (i < A2_WCONV) {
        int r = i;
        const int lane = r & 63;
        r >>= 6;
        const int blk = r & 1;
        r >>= 1;
        const int st = r % 9, p = r / 9, half = lane >> 5;
        const int q = st < 8 ? 2 * p + (st >> 2) : 2 * p + half, tap = st < 8 ? 2 * (st & 3) + half : 8;
        const int o = 32 * blk + (lane & 31);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = w[((long long)o * AM_F + 8 * q + j) * 9 + tap];
    } else if (i < A2_WCONV + A2_WIH) {
        int r = i - A2_WCONV;
        const int lane = r & 63;
        r >>= 6;
        const int blk = r & 1, s = r >> 1, o = 32 * blk + (lane & 31);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = w_ih ? w_ih[o * AM_F + am_chan(8 * s + j, lane >> 5)] : 0.f;
    } else {
        int r = i - A2_WCONV - A2_WIH;
        const int lane = r & 63, s = r >> 6, m = lane & 31;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (w_final && m < 18) ? w_final[((long long)(m & 1) * AM_F + am_chan(8 * s + j, lane >> 5)) * 9 + (m >> 1)] : 0.f;
    }
    out[i] = u32x4{am_pk(v[0], v[1]), am_pk(v[2], v[3]), am_pk(v[4], v[5]), am_pk(v[6], v[7])};
}

struct Amp2Args {
    const _Float16* x;     // [B][4][H][W][16]
    const u32x4* packed;   // k_amp2_pack
    const float* b_conv;   // [64] or null
    const float* b_ih;     // [64] or null
    const float* hh;       // [64]
    const _Float16* hprev; // [B][4][H][W][16] or null
    _Float16* hnew;        // [B][4][H][W][16]
    float* Q;              // [B][3][H][W][2]: the final convolution's tap products pre-summed along x inside the tile (rim_layer2_sb.hip, FAST form)
    float* E;              // [B][H][tile column][16]: what the neighbouring tiles owe columns 0 / 31
    int B, H, W, tiles_x, ntiles;
    int abl;               // probe builds only
};

__global__ __launch_bounds__(A2_NT, 1) void k_amp_layer2(Amp2Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_a2[];
    const u32x4* Wc = reinterpret_cast<const u32x4*>(smem_a2);
    const u32x4* Wih = Wc + A2_WCONV;
    const u32x4* Wp = Wih + A2_WIH;
    float* tabl = reinterpret_cast<float*>(smem_a2 + A2_OFF_TAB);
    const u32x4* Xp = reinterpret_cast<const u32x4*>(smem_a2 + A2_OFF_X);       // [2][A2_XBUF]
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const long long plane = (long long)a.H * a.W;
    const int total = a.ntiles * a.B;

    // once per workgroup: every weight of the layer and the tables
    for (int i = tid; i < A2_PACK_U4; i += A2_NT) reinterpret_cast<u32x4*>(smem_a2)[i] = a.packed[i];
    if (tid < 64) {
        const int tc = am_chan(tid >> 1, tid & 1);
        const int ti = (tid & 1) * 32 + (tid >> 1);
        tabl[ti] = a.hh ? a.hh[tc] : 0.f;
        tabl[64 + ti] = a.b_conv ? a.b_conv[tc] : 0.f;
        tabl[128 + ti] = a.b_ih ? a.b_ih[tc] : 0.f;
    }

    // ---- staging: thread i owns elements i, i + 512, i + 1024 of a job's [chunk in pair][pixel] array (the third exists for i < 416: the others
    // request an out-of-range buffer offset -- zeros, no traffic -- and write the dummy slot).  The pipeline runs two jobs ahead of the MFMAs and
    // across tile boundaries. -------------------------------------------------------------------------------------------------------------------------
    int st_t = blockIdx.x, st_p = 0;
    __amdgpu_buffer_rsrc_t st_rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.x), 0, (unsigned)(plane * (AM_F * 2)), 0x00020000);
    unsigned goff[A2_XV], loff[A2_XV];
#pragma unroll
    for (int v = 0; v < A2_XV; ++v) {
        const int e = tid + v * A2_NT;
        loff[v] = (unsigned)(e < A2_JOB ? e : A2_JOB) * 16u;
    }
    auto st_coords = [&]() {
        const int tq = st_t < total ? st_t : total - 1;      // (beyond the last tile the pipeline keeps requesting the last tile: in range, never read)
        const int tt = (int)mrx_xcd_band(tq, total);
        const int b = tt / a.ntiles, tile = tt - b * a.ntiles, ty0 = tile / a.tiles_x;
        const int h0 = ty0 * AM_TH, w0 = (tile - ty0 * a.tiles_x) * AM_TW;
        st_rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.x) + (long long)b * AM_F * plane, 0, (unsigned)(plane * (AM_F * 2)), 0x00020000);
#pragma unroll
        for (int v = 0; v < A2_XV; ++v) {
            const int e = tid + v * A2_NT;
            const int c = e >= A2_NPIX ? 1 : 0, p = e - c * A2_NPIX;
            const int ty = p / A2_PW, tx = p - ty * A2_PW;
            int gy = h0 + ty - A2_DIL, gx = w0 + tx - A2_DIL;            // replicate border = clamp (conv_layers.py:72-76)
            gy = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy);
            gx = gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx);
            goff[v] = e < A2_JOB ? (unsigned)(((long long)gy * a.W + gx) * 32 + 16 * c) : 0x80000000u;      // a job = one 16-channel block: chunk c is the pixel's second 16 bytes
        }
    };
    u32x4 xr[2][A2_XV];
    auto request = [&](int slot) {
        const unsigned so = (unsigned)st_p * (unsigned)(plane * 32);
#pragma unroll
        for (int v = 0; v < A2_XV; ++v)
            xr[slot][v] = AM_ABL(a, 1) ? u32x4{0x3c003c00u, 0x3c003c00u, so, goff[v]} : __builtin_amdgcn_raw_buffer_load_b128(st_rx, goff[v], so, 0);
        if (++st_p == 4) {
            st_p = 0;
            st_t += gridDim.x;
            st_coords();
        }
    };
    auto commit = [&](int slot, int buf) {
#pragma unroll
        for (int v = 0; v < A2_XV; ++v) *reinterpret_cast<u32x4*>(smem_a2 + A2_OFF_X + buf * (A2_XBUF * 16) + loff[v]) = xr[slot][v];
    };
    st_coords();
    request(0);
    request(1);
    commit(0, 0);
    __syncthreads();                                 // job 0, the weights and the tables are in place

    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        const int tt = (int)mrx_xcd_band(t, total);
        const int b = tt / a.ntiles, tile = tt - b * a.ntiles, ty0 = tile / a.tiles_x;
        const int h0 = ty0 * AM_TH, w0 = (tile - ty0 * a.tiles_x) * AM_TW;
        const int oy0 = h0 + 2 * wave, ox = w0 + l31;

        f32x16 acc[2][2];                            // [row][cout block]: rows 2 wave, 2 wave + 1 of the tile; start at the bias
#pragma unroll
        for (int rw = 0; rw < 2; ++rw)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rw][ct][r] = tabl[64 + lhi * 32 + ct * 16 + r];
        u32x4 hp[2][4];                              // h_prev of the two rows: this lane's four 16-byte pieces each (am_from_pieces in the epilogue)
        const __amdgpu_buffer_rsrc_t rhp = __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16*>(a.hprev ? a.hprev : a.hnew) + (long long)b * AM_F * plane, 0,
                                                                             (unsigned)(plane * (AM_F * 2)), 0x00020000);

        auto toff = [](int tp) { return (tp / 3) * A2_DIL * A2_PW + (tp % 3) * A2_DIL; };
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            request(p & 1);                          // job + 2 (the slot's previous content was committed during the last job)
            __builtin_amdgcn_sched_barrier(0);       // (requests first: the scheduler would sink them behind the commit's s_waitcnt)
            if (p == MRX_AMP_HP_AT) {                // the tile's h_prev, behind it: consumed by the epilogue
                const int cx = ox < a.W ? ox : a.W - 1;
#pragma unroll
                for (int rw = 0; rw < 2; ++rw) {
                    const int cy = oy0 + rw < a.H ? oy0 + rw : a.H - 1;
                    const unsigned off = (a.hprev && !AM_ABL(a, 2)) ? am_piece_off((long long)cy * a.W + cx, lhi) : 0x80000000u;      // (the zero state: out of range reads 0)
#pragma unroll
                    for (int m = 0; m < 4; ++m) hp[rw][m] = __builtin_amdgcn_raw_buffer_load_b128(rhp, off, (unsigned)m * (unsigned)(plane * 32), MRX_AMP_NT_LD ? 2 : 0);
                }
            }
            const u32x4* xw = Xp + (p & 1) * A2_XBUF + (2 * wave) * A2_PW + l31;
            const u32x4* wl = Wc + (p * 9) * 128 + lane;
            u32x4 bt[2][2], at[2][2];                // [buffer][row | cout block]
            auto fetch = [&](int st, int bf) {
                const int off = st < 8 ? (st >> 2) * A2_NPIX + (lhi ? toff(2 * (st & 3) + 1) : toff(2 * (st & 3))) : lhi * A2_NPIX + toff(8);
                bt[bf][0] = xw[off];
                bt[bf][1] = xw[off + A2_PW];
                at[bf][0] = wl[(st * 2 + 0) * 64];
                at[bf][1] = wl[(st * 2 + 1) * 64];
            };
            fetch(0, 0);
#pragma unroll
            for (int st = 0; st < 9; ++st) {
                const int bf = st & 1;
                if (st + 1 < 9) fetch(st + 1, bf ^ 1);
                if (AM_ABL(a, 16)) {
                    asm volatile("" ::"v"(at[bf][0]), "v"(at[bf][1]), "v"(bt[bf][0]), "v"(bt[bf][1]));
                } else {
#pragma unroll
                    for (int rw = 0; rw < 2; ++rw)
#pragma unroll
                        for (int ct = 0; ct < 2; ++ct)
                            acc[rw][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, at[bf][ct]), __builtin_bit_cast(f16x8, bt[bf][rw]), acc[rw][ct], 0, 0, 0);
                }
                if (st == 4) commit((p + 1) & 1, (p + 1) & 1);      // job + 1 (requested one job ago) into the buffer job - 1 was read from
            }
            __syncthreads();
        }

        // ---- g = ReLU(conv + b) rounded to fp16; 1x1 IndRNN stage from registers, both rows together -------------------------------------------------------
        f32x16 acc2[2][2];
#pragma unroll
        for (int rw = 0; rw < 2; ++rw)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[rw][ct][r] = tabl[128 + lhi * 32 + ct * 16 + r];
        {
            const u32x4* wl = Wih + lane;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                f16x8 bg[2];
#pragma unroll
                for (int rw = 0; rw < 2; ++rw) {
                    unsigned g[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int R0 = 8 * s + 2 * k, R1 = R0 + 1;
                        float v0 = acc[rw][R0 >> 4][R0 & 15], v1 = acc[rw][R1 >> 4][R1 & 15];
                        v0 = v0 > 0.f ? v0 : 0.f;
                        v1 = v1 > 0.f ? v1 : 0.f;
                        g[k] = am_pk(v0, v1);
                    }
                    bg[rw] = __builtin_bit_cast(f16x8, (u32x4{g[0], g[1], g[2], g[3]}));
                }
                const f16x8 a0 = __builtin_bit_cast(f16x8, wl[(s * 2 + 0) * 64]), a1 = __builtin_bit_cast(f16x8, wl[(s * 2 + 1) * 64]);
#pragma unroll
                for (int rw = 0; rw < 2; ++rw) {
                    acc2[rw][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bg[rw], acc2[rw][0], 0, 0, 0);
                    acc2[rw][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bg[rw], acc2[rw][1], 0, 0, 0);
                }
            }
        }
        // ---- h = ReLU(W_ih g + b_ih + hh * h_prev) rounded to fp16 once: stored, and the operand of the final convolution's contraction -------------------
        const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(a.hnew + (long long)b * AM_F * plane, 0, (unsigned)(plane * (AM_F * 2)), 0x00020000);
        u32x2 hq[2][8];
#pragma unroll
        for (int rw = 0; rw < 2; ++rw) {
            const int oy = oy0 + rw;
            const unsigned offh = (oy < a.H && ox < a.W && !AM_ABL(a, 4)) ? am_piece_off((long long)oy * a.W + ox, lhi) : 0x80000000u;
            u32x2 hprev[8];
            am_from_pieces(hp[rw], hprev);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float hv[4] = {am_lo(hprev[q].x), am_hi(hprev[q].x), am_lo(hprev[q].y), am_hi(hprev[q].y)};
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int R = 4 * q + i;
                    v[i] = acc2[rw][R >> 4][R & 15] + tabl[lhi * 32 + R] * hv[i];
                    v[i] = v[i] > 0.f ? v[i] : 0.f;
                }
                hq[rw][q] = u32x2{am_pk(v[0], v[1]), am_pk(v[2], v[3])};
            }
            u32x4 pc[4];
            am_to_pieces(hq[rw], pc);
#pragma unroll
            for (int m = 0; m < 4; ++m) __builtin_amdgcn_raw_buffer_store_b128(pc[m], rh, offh + (unsigned)m * (unsigned)(plane * 32), 0, MRX_AMP_NT_ST ? 2 : 0);
        }
        if (a.Q && !AM_ABL(a, 8)) {
            f32x16 accp[2];
#pragma unroll
            for (int rw = 0; rw < 2; ++rw)
#pragma unroll
                for (int r = 0; r < 16; ++r) accp[rw][r] = 0.f;
            const u32x4* wp = Wp + lane;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const f16x8 a1 = __builtin_bit_cast(f16x8, wp[s * 64]);
#pragma unroll
                for (int rw = 0; rw < 2; ++rw) {
                    const f16x8 bh = __builtin_bit_cast(f16x8, (u32x4{hq[rw][2 * s].x, hq[rw][2 * s].y, hq[rw][2 * s + 1].x, hq[rw][2 * s + 1].y}));
                    accp[rw] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bh, accp[rw], 0, 0, 0);
                }
            }
            // Eighteen tap planes become six: the three products of a kernel row meet inside the wave's image row (lane = pixel); what columns 0 / 31 of the
            // tile miss the neighbouring tile leaves in E -- the layout of rim_layer2_sb.hip's FAST form, consumed by k_llg372<GAT> / k_l2sb_gather_q.
            // Lanes of the lower half-wave hold products m = 0..3, 8..11, 16, 17 (m = (dy * 3 + dx) * 2 + co), those of the upper half 4..7, 12..15.
            const int tcol = (w0 / AM_TW);
            const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc(a.Q + (long long)b * 6 * plane, 0, (unsigned)(plane * (6 * 4)), 0x00020000);
            const long long eslots = (long long)a.H * a.tiles_x;       // 16 floats per (row, tile column)
            const __amdgpu_buffer_rsrc_t re = __builtin_amdgcn_make_buffer_rsrc(a.E + (long long)b * 16 * eslots, 0, (unsigned)(eslots * 64), 0x00020000);
            const bool hasL = l31 > 0, hasR = l31 < 31 && ox + 1 < a.W;
            const bool useL = hasL || ox == 0, useR = l31 < 31 || ox == a.W - 1;
            auto fromL = [&](float v_) {                         // the value of the pixel to the left (the pixel itself on the image border, 0 across a tile border)
                const float t_ = __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v_), 0x138, 0xf, 0xf, false));   // wave_shr:1
                return useL ? (hasL ? t_ : v_) : 0.f;
            };
            auto fromR = [&](float v_) {
                const float t_ = __uint_as_float((unsigned)__builtin_amdgcn_update_dpp(0, (int)__float_as_uint(v_), 0x130, 0xf, 0xf, false));   // wave_shl:1
                return useR ? (hasR ? t_ : v_) : 0.f;
            };
#pragma unroll
            for (int rw = 0; rw < 2; ++rw) {
                const int oy = oy0 + rw;
                const bool inside = oy < a.H && ox < a.W;
                float v[10];
#pragma unroll
                for (int r = 0; r < 10; ++r) v[r] = accp[rw][r];
                // plane dy * 2 + co = A(dx 0, x - 1) + B(dx 1, x) + C(dx 2, x + 1); the products sit in (register, half-wave):
                //   plane 0, 1: A (co, lower)      B (2 + co, lower)   C (co, upper)        -> stored by the lower half-wave
                //   plane 2, 3: A (2 + co, upper)  B (4 + co, lower)   C (6 + co, lower)    -> plane 2 by the lower, plane 3 by the upper half-wave
                //   plane 4, 5: A (4 + co, upper)  B (6 + co, upper)   C (8 + co, lower)    -> stored by the upper half-wave
                const float sw0 = __shfl_xor(lhi ? fromR(v[0]) : v[5] + fromR(v[7]), 32, 64);
                const float sw1 = __shfl_xor(lhi ? fromR(v[1]) : fromR(v[8]), 32, 64);
                const float sw2 = __shfl_xor(lhi ? fromL(v[2]) : fromR(v[9]), 32, 64);
                float q[3];
                q[0] = lhi ? fromL(v[3]) + sw0 : (fromL(v[0]) + v[2]) + sw0;
                q[1] = lhi ? (fromL(v[4]) + v[6]) + sw1 : (fromL(v[1]) + v[3]) + sw1;
                q[2] = lhi ? (fromL(v[5]) + v[7]) + sw2 : sw2 + (v[4] + fromR(v[6]));
                const unsigned offq = (inside && !AM_ABL(a, 32)) ? (unsigned)(((long long)oy * a.W + ox) * 8) : 0x80000000u;      // (probe bit 32: tap stage computed, not stored)
                const u32x2 pair = lhi ? u32x2{__float_as_uint(q[1]), __float_as_uint(q[2])} : u32x2{__float_as_uint(q[0]), __float_as_uint(q[1])};
                __builtin_amdgcn_raw_buffer_store_b64(pair, rq, (AM_ABL(a, 64) ? 0x80000000u : offq) + (lhi ? 2u : 0u) * (unsigned)(plane * 8), 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(lhi ? q[0] : q[2]), rq, (AM_ABL(a, 128) ? 0x80000000u : offq) + (unsigned)(plane * 8) + (lhi ? 4u : 0u), 0, 0);
                const bool col0 = l31 == 0 && inside && tcol > 0, col31 = l31 == 31 && inside && tcol + 1 < a.tiles_x;
                const u32x4 ev = l31 == 0 ? (lhi ? u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), 0u, 0u}
                                                 : u32x4{__float_as_uint(v[6]), __float_as_uint(v[7]), __float_as_uint(v[8]), __float_as_uint(v[9])})
                                          : (lhi ? u32x4{__float_as_uint(v[2]), __float_as_uint(v[3]), __float_as_uint(v[4]), __float_as_uint(v[5])}
                                                 : u32x4{__float_as_uint(v[0]), __float_as_uint(v[1]), 0u, 0u});
                const unsigned eoff = ((col0 || col31) && !AM_ABL(a, 32) && !AM_ABL(a, 256)) ? (unsigned)(((long long)oy * a.tiles_x + tcol) * 64 + (l31 == 0 ? (lhi ? 16 : 0) : (lhi ? 48 : 32))) : 0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b128(ev, re, eoff, 0, 0);
            }
        }
    }
}

// ---- C entry points ----------------------------------------------------------------------------------------------------------------------------------
static int am_ncu() {
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return ncu;
}

extern "C" int64_t mrx_amp16_pack_floats(int layer) { return layer == 1 ? (int64_t)A1_PACK_U4 * 4 : (layer == 2 ? (int64_t)A2_PACK_U4 * 4 : -1); }

extern "C" int mrx_amp16_layer1_pack(const float* w_conv, const float* w_ih, float* packed, int Cin, void* stream) {
    MRX_REQUIRE(w_conv && w_ih && packed, MRX_EINVAL, "mrx_amp16_layer1_pack: null pointer");
    MRX_REQUIRE(Cin >= 1 && Cin <= 4, MRX_EUNSUP, "mrx_amp16_layer1_pack: Cin = %d (1 .. 4)", Cin);
    hipLaunchKernelGGL(k_amp1_pack, dim3((A1_PACK_U4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_conv, w_ih, reinterpret_cast<u32x4*>(packed), Cin);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

extern "C" int mrx_amp16_layer2_pack(const float* w_conv, const float* w_ih, const float* w_final, float* packed, void* stream) {
    MRX_REQUIRE(w_conv && w_ih && packed, MRX_EINVAL, "mrx_amp16_layer2_pack: null pointer");
    hipLaunchKernelGGL(k_amp2_pack, dim3((A2_PACK_U4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_conv, w_ih, w_final, reinterpret_cast<u32x4*>(packed));
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

extern "C" int mrx_amp16_layer1(const float* x, int Cin, const float* eta, const float* part, int nparts, float inv_sigma2, const float* packed,
                                const float* b_conv, const float* b_ih, const float* hh, const void* h_prev, void* h_new, int B, int H, int W, void* stream) {
    MRX_REQUIRE((x || eta) && packed && hh && h_new, MRX_EINVAL, "mrx_amp16_layer1: null pointer");
    MRX_REQUIRE(B >= 0 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_amp16_layer1: bad dims");
    MRX_REQUIRE(eta ? (part && nparts >= 1 && nparts <= 4) : (Cin >= 1 && Cin <= 4), MRX_EUNSUP,
                "mrx_amp16_layer1: input is (eta, 1 .. 4 partial planes) or x with 1 .. 4 channels (nparts %d, Cin %d)", nparts, Cin);
    MRX_REQUIRE((long long)H * W < (1ll << 31), MRX_EUNSUP, "mrx_amp16_layer1: image too large");
    if (B == 0) return MRX_OK;
    Amp1Args a;
    a.x = x, a.eta2 = reinterpret_cast<const float2*>(eta), a.part = reinterpret_cast<const float2*>(part), a.part_stride = (long long)B * H * W;
    a.nparts = nparts, a.post = inv_sigma2, a.packed = reinterpret_cast<const u32x4*>(packed), a.b_conv = b_conv, a.b_ih = b_ih, a.hh = hh;
    a.hprev = reinterpret_cast<const _Float16*>(h_prev), a.hnew = reinterpret_cast<_Float16*>(h_new);
    a.B = B, a.Cin = eta ? 4 : Cin, a.H = H, a.W = W, a.tiles_x = mrx_cdiv(W, AM_TW), a.ntiles = a.tiles_x * mrx_cdiv(H, A1_NW);
    a.abl = MRX_DEBUG_ENV("MRX_AMP_ABL") ? atoi(MRX_DEBUG_ENV("MRX_AMP_ABL")) : 0;
    static bool attr_done = false;   // once: keeps launches legal under hipGraph capture
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)k_amp_layer1_t<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)A1T_LDS);
        (void)hipFuncSetAttribute((const void*)k_amp_layer1_t<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)A1T_LDS);
        attr_done = true;
    }
    const long long total = (long long)a.ntiles * B;
    const int ncu = am_ncu(), grid = (int)(total < ncu ? total : ncu);
    if (eta) hipLaunchKernelGGL(k_amp_layer1_t<1>, dim3(grid), dim3(A1T_NT), A1T_LDS, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(k_amp_layer1_t<0>, dim3(grid), dim3(A1T_NT), A1T_LDS, (hipStream_t)stream, a);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

extern "C" int mrx_amp16_layer2(const void* x, const float* packed, const float* b_conv, const float* b_ih, const float* hh, const void* h_prev, void* h_new,
                                float* taps_q, float* edges, int B, int H, int W, void* stream) {
    MRX_REQUIRE(x && packed && hh && h_new, MRX_EINVAL, "mrx_amp16_layer2: null pointer");
    MRX_REQUIRE((taps_q == nullptr) == (edges == nullptr), MRX_EINVAL, "mrx_amp16_layer2: taps_q and edges come together");
    MRX_REQUIRE(B >= 0 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_amp16_layer2: bad dims");
    MRX_REQUIRE((long long)H * W * (AM_F * 4) < (1ll << 31), MRX_EUNSUP, "mrx_amp16_layer2: %d x %d: a sample's state exceeds 32-bit byte offsets", H, W);
    if (B == 0) return MRX_OK;
    Amp2Args a;
    a.x = reinterpret_cast<const _Float16*>(x), a.packed = reinterpret_cast<const u32x4*>(packed), a.b_conv = b_conv, a.b_ih = b_ih, a.hh = hh;
    a.hprev = reinterpret_cast<const _Float16*>(h_prev), a.hnew = reinterpret_cast<_Float16*>(h_new), a.Q = taps_q, a.E = edges;
    a.B = B, a.H = H, a.W = W, a.tiles_x = mrx_cdiv(W, AM_TW), a.ntiles = a.tiles_x * mrx_cdiv(H, AM_TH);
    a.abl = MRX_DEBUG_ENV("MRX_AMP_ABL") ? atoi(MRX_DEBUG_ENV("MRX_AMP_ABL")) : 0;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute((const void*)k_amp_layer2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)A2_LDS);
        attr_done = true;
    }
    const long long total = (long long)a.ntiles * B;
    const int ncu = am_ncu(), grid = (int)(total < ncu ? total : ncu);
    hipLaunchKernelGGL(k_amp_layer2, dim3(grid), dim3(A2_NT), A2_LDS, (hipStream_t)stream, a);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
