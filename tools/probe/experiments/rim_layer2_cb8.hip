// rim_layer2_cb8.hip -- the second RIM layer (ConvNonlinear 3x3 dilation 2, 64 -> 64, replicate padding, ReLU + IndRNNCell 1x1 64 -> 64) and the
// channel contraction of the final 3x3 64 -> 2 convolution (reference models/rim/conv_layers.py:121-123, rnn_cells.py:384-391, rim_block.py:233-246)
// on two-term fp16 operands with fp32 results -- the arithmetic of k_rim_layer2_sb<F16> (rim_layer2_sb.hip), re-organised after its phase
// ablation (profiles/r03_layer2_ablation_*.txt): of 72 us, 25 were the row tails (1x1 stage, epilogue, tap stage) running AFTER the convolution
// with the matrix pipe idle, 10 the staging of 32-bit loads from 64 channel planes.
//
//   * hidden states are CHANNEL-BLOCKED between the kernels of a RIM step: h[b][c / 8][y][x][c % 8] (fp32, "CB8").  A lane of the accumulator
//     layout owns four consecutive channels of each block: every state access is a 16-byte instruction (8 + 8 per image row of a wave instead
//     of 32 + 32), a pixel's eight channels of a block are 32 contiguous bytes for the loader (2 x 16 bytes instead of 8 x 4 from 8 planes),
//     and a 32-pixel unit is a 1 KB-aligned 1 KB block (whole cache lines);
//   * one persistent workgroup per CU, FOUR waves -- one per SIMD, up to 512 registers -- on 8 x 32 pixel tiles, a wave = two image rows x 64
//     couts.  The tail of tile t runs INSIDE the chunk loop of tile t + 1 of the same wave: ReLU(conv + b) is parked in registers at the end of
//     a tile and the eight chunks of the next one each carry one eighth of its tail (row 0: chunks 0-3, row 1: chunks 4-7), so the vector ALU
//     work, the 36 + 36 tail MFMAs and the state traffic sit in the shadow of the convolution's MFMAs instead of behind them;
//   * no branch in the steady-state loop: state loads / stores are buffer instructions whose out-of-range offsets the hardware drops (lanes
//     beyond the image, the empty tail of a workgroup's first tile), the loader re-reads a valid tile when it runs out of tiles;
//   * three LDS buffers for the term planes and the chunk weights: the ninth taps of a chunk pair share an MFMA step (rim_layer2_sb.hip) and
//     chunk c + 2 is written while chunk c is multiplied, with ONE barrier per chunk;
//   * the final convolution's tap products leave as [b][tap][y][x][co] (float2 per lane: 5 stores instead of 10).
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "mrx_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

#define C8_NT 512
#define C8_TH 8
#define C8_TW 32
#define C8_F 64
#define C8_PH (C8_TH + 4)
#define C8_PW (C8_TW + 4)
#define C8_NPIX (C8_PH * C8_PW)               // 432 halo'd pixels
#define C8_NCH 8
// the operand pack is mrx_rim_layer2_f16_pack's (rim_layer2_sb.hip), 16-byte units:
#define C8_WFULL (4 * 2 * 2 * 64)             // four full steps of a chunk
#define C8_WCH (C8_WFULL + 2 * 2 * 32)        // + tap 8 (lower half-wave only): 1152 = 18 KB per chunk
#define C8_PK_TAIL (C8_NCH * C8_WCH)          // 1x1 operands (1024 used of 1536), then the final convolution's (512 used of 768), then the header
#define C8_PK_WIH_SLOTS 1536
#define C8_PK_WP_SLOTS 768
#define C8_PK_HEADER (C8_PK_TAIL + C8_PK_WIH_SLOTS + C8_PK_WP_SLOTS)
#define C8_WIH 1024
#define C8_WP 512
// LDS (bytes)
#define C8_OFF_WP (C8_WIH * 16)
#define C8_OFF_TAB (C8_OFF_WP + C8_WP * 16)
#define C8_OFF_W (C8_OFF_TAB + 1024)
#define C8_XBUF (2 * C8_NPIX * 16)            // one buffer of two term planes: 13824
#define C8_OFF_X (C8_OFF_W + 3 * C8_WCH * 16)
#define C8_LDS (C8_OFF_X + 3 * C8_XBUF)       // 122368

struct L2c8Args {
    const float* x;        // [B][8][H][W][8]  (CB8)
    const u32x4* packed;   // mrx_rim_layer2_f16_pack
    const float* b_conv;   // [64] or null
    const float* b_ih;     // [64] or null
    const float* hh;       // [64]
    const float* hprev;    // CB8 or null
    float* hnew;           // CB8 (may be hprev)
    float* P;              // [B][9][H][W][2] or null
    const unsigned* xmax;  // bits of an upper bound of max |x|
    int B, H, W, tiles_x, ntiles;
    unsigned long long* trace;   // debug (env MRX_L2C8_TRACE): cycle stamps [workgroup][wave][8]
};

__device__ __forceinline__ void c8_split2h(float a, float b, unsigned& p1, unsigned& p2) {
    const f16x2 h = {(_Float16)a, (_Float16)b};
    const float ra = a - (float)h.x, rb = b - (float)h.y;     // exact
    const f16x2 l = {(_Float16)ra, (_Float16)rb};
    p1 = __builtin_bit_cast(unsigned, h);
    p2 = __builtin_bit_cast(unsigned, l);
}
// the same for a * s, b * s with the scale folded into the conversions (two v_fma_mix per value: f16(a s), f16(a s - hi))
__device__ __forceinline__ void c8_split2hs(float a, float b, float s, unsigned& p1, unsigned& p2) {
    const f16x2 h = {(_Float16)(a * s), (_Float16)(b * s)};
    const f16x2 l = {(_Float16)(a * s - (float)h.x), (_Float16)(b * s - (float)h.y)};
    p1 = __builtin_bit_cast(unsigned, h);
    p2 = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ float c8_pow2(int e) {
    e = e < -120 ? -120 : (e > 120 ? 120 : e);
    return __uint_as_float((unsigned)(127 + e) << 23);
}
__device__ __forceinline__ int c8_scale_exp(unsigned bits) {
    const int ex = (int)((bits >> 23) & 0xffu);
    return (ex == 0 || ex == 255) ? 0 : 14 - (ex - 127);
}
__host__ __device__ constexpr int c8_chan(int R, int half) { return 32 * (R >> 4) + (R & 3) + 8 * ((R & 15) >> 2) + 4 * half; }
__device__ __forceinline__ int c8_pixel_exp(float m) {
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    return c8_scale_exp(__float_as_uint(m));
}
#define C8_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, A), __builtin_bit_cast(f16x8, B), C, 0, 0, 0)
#define C8_OOR 0x80000000u                     // a buffer offset beyond every tensor here (also with a block offset added): the load returns 0, the store is dropped

// TAPS: also the final convolution's tap products (a.P).  ABL (probe builds, -DMRX_PROBE + env MRX_L2C8_ABL): 1 no tail slices, 2 no staging
// slices, 4 no convolution MFMAs, 8 no operand fetches, 16 upper wave half in the late order, 32 no barriers
// -- garbage results, only the time is read (1 and 3 let the compiler drop the convolution as dead code: use 4 / 7 / 11 instead).
template <bool TAPS, int ABL = 0>
__global__ __launch_bounds__(C8_NT, 2) void k_rim_layer2_cb8(L2c8Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_c8[];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // = the wave's row of the tile
    const long long plane = (long long)a.H * a.W;
    const int total = a.ntiles * a.B;
    const unsigned sample_bytes = (unsigned)(plane * C8_F * 4);

    // ---- once per workgroup: 1x1 / tap operands and the tables (register order [table][half][R]) ------------------------------------------------
    {
        u32x4* Wih = reinterpret_cast<u32x4*>(smem_c8);
        for (int i = tid; i < C8_WIH; i += C8_NT) Wih[i] = a.packed[C8_PK_TAIL + i];
        u32x4* Wp = reinterpret_cast<u32x4*>(smem_c8 + C8_OFF_WP);
        for (int i = tid; i < C8_WP; i += C8_NT) Wp[i] = a.packed[C8_PK_TAIL + C8_PK_WIH_SLOTS + i];
        float* tabl = reinterpret_cast<float*>(smem_c8 + C8_OFF_TAB);
        if (tid < 192) {
            const int k = tid >> 6, half = (tid >> 5) & 1, R = tid & 31, c = c8_chan(R, half);
            const float* src = k == 0 ? a.hh : (k == 1 ? a.b_conv : a.b_ih);
            tabl[tid] = src ? src[c] : 0.f;
        }
    }
    const u32x4 hd = a.packed[C8_PK_HEADER];
    const int kx = c8_scale_exp(a.xmax[0]);
    const float sx = c8_pow2(kx), un_conv = c8_pow2(-kx) * c8_pow2(-(int)hd[0]);
    const float unwi = c8_pow2(-(int)hd[1]), unwp = c8_pow2(-(int)hd[2]);
    const f32x4* tab4 = reinterpret_cast<const f32x4*>(smem_c8 + C8_OFF_TAB) + lhi * 8;     // + k * 16 + j: table k, registers 4 j .. 4 j + 3

    // ---- staging stream: (tile, chunk) pairs of this workgroup in order; chunk c lives in LDS buffer c % 3 ---------------------------------------
    // items e = tid + 512 v (v < XV) of the 864 (pixel, channel half) pairs of a halo'd tile; e >= 864 repeats item 863 (same value, same address)
    constexpr int XV = (2 * C8_NPIX + C8_NT - 1) / C8_NT;      // 2
    constexpr int WV = (C8_WCH + C8_NT - 1) / C8_NT;           // 3
    int st_t = blockIdx.x, st_q = 0;
    unsigned xoff[XV], xdst[XV];
#pragma unroll
    for (int v = 0; v < XV; ++v) {
        int e = tid + C8_NT * v;
        e = e < 2 * C8_NPIX ? e : 2 * C8_NPIX - 1;
        xdst[v] = (unsigned)((e >> 1) * 16 + (e & 1) * 8);
    }
    __amdgpu_buffer_rsrc_t st_rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, sample_bytes, 0x00020000);
    auto st_coords = [&]() {
        const int t = st_t < total ? st_t : total - 1;                       // out of tiles: stage the last one again (never used)
        const int tt = (int)mrx_xcd_band(t, total);
        const int b = tt / a.ntiles, tile = tt - b * a.ntiles, ty0 = tile / a.tiles_x;
        const int h0 = ty0 * C8_TH, w0 = (tile - ty0 * a.tiles_x) * C8_TW;
        st_rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x) + (long long)b * C8_F * plane, 0, sample_bytes, 0x00020000);
#pragma unroll
        for (int v = 0; v < XV; ++v) {
            int e = tid + C8_NT * v;
            e = e < 2 * C8_NPIX ? e : 2 * C8_NPIX - 1;
            const int p = e >> 1, ty = p / C8_PW, tx = p - ty * C8_PW;
            int gy = h0 + ty - 2, gx = w0 + tx - 2;                           // replicate border = clamp (conv_layers.py:72-76)
            gy = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy);
            gx = gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx);
            xoff[v] = (unsigned)((gy * a.W + gx) * 32 + (e & 1) * 16);
        }
    };
    const __amdgpu_buffer_rsrc_t rw_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(a.packed), 0, (unsigned)(C8_PK_TAIL * 16), 0x00020000);
    unsigned woff[WV];
#pragma unroll
    for (int v = 0; v < WV; ++v) {
        const int i = tid + C8_NT * v;
        woff[v] = (unsigned)((i < C8_WCH ? i : C8_WCH - 1) * 16);
    }
    u32x4 xr[XV], wr[WV];
    unsigned rq_so = 0, rq_wo = 0;                   // scalar offsets of the request in progress (block plane of x, chunk of the weight pack)
    __amdgpu_buffer_rsrc_t rq_rx = st_rx;
    auto request_begin = [&]() {
        rq_so = (unsigned)st_q * (unsigned)(plane * 32);
        rq_wo = (unsigned)st_q * (C8_WCH * 16);
        rq_rx = st_rx;
    };
    auto request_x = [&](int v) { xr[v] = __builtin_amdgcn_raw_buffer_load_b128(rq_rx, xoff[v], rq_so, 0); };
    auto request_w = [&](int v) { wr[v] = __builtin_amdgcn_raw_buffer_load_b128(rw_, woff[v], rq_wo, 0); };
    auto request_end = [&]() {
        if (++st_q == C8_NCH) {
            st_q = 0;
            st_t += gridDim.x;
            st_coords();
        }
    };
    auto commit_x = [&](int v, int buf) {           // two fp16 terms of x 2^kx into the term planes
        unsigned char* xb = smem_c8 + C8_OFF_X + buf * C8_XBUF;
        unsigned h0_, l0_, h1_, l1_;
        c8_split2hs(__uint_as_float(xr[v][0]), __uint_as_float(xr[v][1]), sx, h0_, l0_);
        c8_split2hs(__uint_as_float(xr[v][2]), __uint_as_float(xr[v][3]), sx, h1_, l1_);
        *reinterpret_cast<u32x2*>(xb + xdst[v]) = u32x2{h0_, h1_};
        *reinterpret_cast<u32x2*>(xb + C8_NPIX * 16 + xdst[v]) = u32x2{l0_, l1_};
    };
    auto commit_w = [&](int v, int buf) {
        u32x4* wb = reinterpret_cast<u32x4*>(smem_c8 + C8_OFF_W) + buf * C8_WCH;
        const int i = tid + C8_NT * v;
        wb[i < C8_WCH ? i : C8_WCH - 1] = wr[v];
    };
    auto request_next = [&]() {
        request_begin();
#pragma unroll
        for (int v = 0; v < XV; ++v) request_x(v);
#pragma unroll
        for (int v = 0; v < WV; ++v) request_w(v);
        request_end();
    };
    auto commit_next = [&](int buf) {
#pragma unroll
        for (int v = 0; v < XV; ++v) commit_x(v, buf);
#pragma unroll
        for (int v = 0; v < WV; ++v) commit_w(v, buf);
    };
    // "commit chunk c + 2, request chunk c + 3" in slices, one per MFMA step: slice s writes the items it owns (loaded a chunk ago) to LDS and
    // asks for the same items of the chunk after next
    auto stage_slice = [&](int s, int buf) {
        if (s == 0) request_begin();
        if (s < XV) {
            commit_x(s, buf);
            request_x(s);
        }
        if (s < WV) {
            commit_w(s, buf);
            request_w(s);
        }
        if (s == 3) request_end();
    };
    st_coords();
    request_next();
    commit_next(0);
    request_next();
    commit_next(1);
    request_next();                                  // chunk 2 in flight
    __syncthreads();

    // ---- the parked tile: g = ReLU(conv + b) of the wave's row of the previous tile and where it goes ----------------------------------------------
    float g[32];
#pragma unroll
    for (int R = 0; R < 32; ++R) g[R] = 0.f;
    unsigned pd_off = C8_OOR;                        // byte offset of (the row, this lane's pixel, channel half) in a CB8 block plane, or out of range
    unsigned pd_poff = C8_OOR;                       // the same in a [tap][H][W][2] plane (the upper half-wave's two taps further)
    int pd_b = 0;
    f32x16 acc2[2];
    f32x16 accp;
    float hp[32];
    float sg = 1.f, ung = 1.f, sh = 1.f, unh = 1.f;
    const float* hsrc = a.hprev ? a.hprev : a.hnew;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // The parked row's tail in slices: chunk j of the next tile carries the pieces below at its steps 0 and 2 -- the two waves of a SIMD run the
    // same code half a step apart at most, so what hides a wave's vector-ALU stretches is its partner's MFMAs and the other way round.
    auto tail_1x1_step = [&](int s) {
        // one of the four contraction steps of the 1x1 stage: two fp16 terms of g 2^kg, three term products
        const u32x4* wl = reinterpret_cast<const u32x4*>(smem_c8) + lane;
        unsigned g1[4], g2[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) c8_split2hs(g[8 * s + 2 * k], g[8 * s + 2 * k + 1], sg, g1[k], g2[k]);
        const u32x4 b1 = u32x4{g1[0], g1[1], g1[2], g1[3]}, b2 = u32x4{g2[0], g2[1], g2[2], g2[3]};
        u32x4 at_[2][2];
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) at_[ct][k] = wl[((s * 2 + k) * 2 + ct) * 64];
        acc2[0] = C8_MFMA(at_[0][1], b1, s == 0 ? zero16 : acc2[0]);
        acc2[1] = C8_MFMA(at_[1][1], b1, s == 0 ? zero16 : acc2[1]);
        acc2[0] = C8_MFMA(at_[0][0], b2, acc2[0]);
        acc2[1] = C8_MFMA(at_[1][0], b2, acc2[1]);
        acc2[0] = C8_MFMA(at_[0][0], b1, acc2[0]);
        acc2[1] = C8_MFMA(at_[1][0], b1, acc2[1]);
    };
    auto tail_epilogue = [&](int half) {
        // h = ReLU(W_ih g + b_ih + hh * h_prev), half of the row's registers; 16-byte stores; the new state stays in hp for the tap stage
        const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(a.hnew + (long long)pd_b * C8_F * plane, 0, sample_bytes, 0x00020000);
#pragma unroll
        for (int q = 4 * half; q < 4 * half + 4; ++q) {
            const f32x4 hhv = tab4[q], biv = tab4[32 + q];
            u32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int R = 4 * q + i;
                float v = acc2[R >> 4][R & 15] * ung + biv[i];
                v += hhv[i] * hp[R];
                v = v > 0.f ? v : 0.f;
                hp[R] = v;
                o[i] = __float_as_uint(v);
            }
            // (block offset in the VECTOR offset, scalar offset 0: with a 16-byte store whose scalar offset is a register hipcc pads no wait state
            // before a vector-ALU write of the store's data registers -- on gfx950 lanes 12-15 / 28-31 of the last dword then carry the NEW value)
            __builtin_amdgcn_raw_buffer_store_b128(o, rh, pd_off + (unsigned)q * (unsigned)(plane * 32), 0, 0);
        }
    };
    auto tail_tap_step = [&](int s) {
        // the final 64 -> 2 convolution's channel contraction on the new state, one of its four steps: D[tap * 2 + co][pixel], 18 of 32 rows
        const u32x4* wp = reinterpret_cast<const u32x4*>(smem_c8 + C8_OFF_WP) + lane;
        unsigned g1[4], g2[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) c8_split2hs(hp[8 * s + 2 * k], hp[8 * s + 2 * k + 1], sh, g1[k], g2[k]);
        const u32x4 b1 = u32x4{g1[0], g1[1], g1[2], g1[3]}, b2 = u32x4{g2[0], g2[1], g2[2], g2[3]};
        const u32x4 a1 = wp[(s * 2 + 0) * 64], a2 = wp[(s * 2 + 1) * 64];
        accp = C8_MFMA(a2, b1, s == 0 ? zero16 : accp);
        accp = C8_MFMA(a1, b2, accp);
        accp = C8_MFMA(a1, b1, accp);
    };
    auto tail_slice = [&](int j, int s) {
        if (s != 0 && s != 2) return;
        const int k = 2 * j + (s >> 1);              // piece 0 .. 15
        if (k == 0) {
            // h_prev of the row: eight 16-byte loads (block q holds registers 4 q .. 4 q + 3); the zero state = out-of-range offsets = zeros
            const __amdgpu_buffer_rsrc_t rh =
                __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hsrc) + (long long)pd_b * C8_F * plane, 0, sample_bytes, 0x00020000);
            const unsigned lo = a.hprev ? pd_off : C8_OOR;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rh, lo, (unsigned)q * (unsigned)(plane * 32), 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) hp[4 * q + i] = __uint_as_float(v[i]);
            }
        }
        if (k == 1) {
            // one scale per PIXEL for the 1x1 stage: its contraction runs over the pixel's 64 channels only (this lane's 32 and lane ^ 32's)
            float gm = 0.f;
#pragma unroll
            for (int R = 0; R < 32; ++R) gm = fmaxf(gm, g[R]);
            const int kg = c8_pixel_exp(gm);
            sg = c8_pow2(kg);
            ung = c8_pow2(-kg) * unwi;
        }
        if (k >= 2 && k < 6) tail_1x1_step(k - 2);
        if (k == 6 || k == 7) tail_epilogue(k - 6);
        if (TAPS && k == 8) {
            float hm = 0.f;
#pragma unroll
            for (int R = 0; R < 32; ++R) hm = fmaxf(hm, hp[R]);
            const int kh = c8_pixel_exp(hm);
            sh = c8_pow2(kh);
            unh = c8_pow2(-kh) * unwp;
        }
        if (TAPS && k >= 9 && k < 13) tail_tap_step(k - 9);
        if (TAPS && k == 13) {
            // rows m = (r & 3) + 8 (r >> 2) + 4 half = tap * 2 + co: registers 2 i, 2 i + 1 are the (co 0, co 1) pair of tap (i & 1) + 4 (i >> 1) + 2 half
            const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(a.P + (long long)pd_b * 18 * plane, 0, (unsigned)(plane * 18 * 4), 0x00020000);
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int tap0 = (i & 1) + 4 * (i >> 1);                       // lower half-wave; the upper one holds tap0 + 2 (i < 4 only)
                const u32x2 o = u32x2{__float_as_uint(accp[2 * i] * unh), __float_as_uint(accp[2 * i + 1] * unh)};
                const unsigned off = (i == 4 && lhi) ? C8_OOR : pd_poff;          // (pd_poff already carries the upper half-wave's two taps)
                __builtin_amdgcn_raw_buffer_store_b64(o, rp, off, (unsigned)tap0 * (unsigned)(plane * 8), 0);
            }
        }
    };

    // MFMA operands of a step: [ct | -][term], fetched one side-work slice ahead of the MFMAs that consume them
    u32x4 bt[2], at[2][2];
    auto toff = [](int tp) { return (tp / 3) * 2 * C8_PW + (tp % 3) * 2; };
    auto fetch = [&](int s, int cbuf, int cnext) {
        if (s < 4) {
            const u32x4* xw = reinterpret_cast<const u32x4*>(smem_c8 + C8_OFF_X + cbuf * C8_XBUF) + wave * C8_PW + l31;
            const u32x4* wl = reinterpret_cast<const u32x4*>(smem_c8 + C8_OFF_W) + cbuf * C8_WCH + lane;
            const int off = lhi ? toff(2 * s + 1) : toff(2 * s);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                bt[k] = xw[k * C8_NPIX + off];
                at[0][k] = wl[((s * 2 + k) * 2 + 0) * 64];
                at[1][k] = wl[((s * 2 + k) * 2 + 1) * 64];
            }
        } else {
            // tap 8 of this chunk (lower half-wave) and of the next one (upper half-wave: its buffer was completed before the last barrier)
            const int cbx = lhi ? cnext : cbuf;
            const u32x4* xw8 = reinterpret_cast<const u32x4*>(smem_c8 + C8_OFF_X + cbx * C8_XBUF) + wave * C8_PW + l31 + toff(8);
            const u32x4* w8 = reinterpret_cast<const u32x4*>(smem_c8 + C8_OFF_W) + cbx * C8_WCH + C8_WFULL + l31;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                bt[k] = xw8[k * C8_NPIX];
                at[0][k] = w8[(k * 2 + 0) * 32];
                at[1][k] = w8[(k * 2 + 1) * 32];
            }
        }
    };

    int cb = 0;                                      // LDS buffer of the current chunk
    if (wave < 4 || !(ABL & 16)) fetch(0, 0, 1);
    int tcount = 0;
    for (int t = blockIdx.x; t < total; t += gridDim.x, ++tcount) {
        const int tt = (int)mrx_xcd_band(t, total);
        const int b = tt / a.ntiles, tile = tt - b * a.ntiles, ty0 = tile / a.tiles_x;
        const int h0 = ty0 * C8_TH, w0 = (tile - ty0 * a.tiles_x) * C8_TW;
        if (a.trace && lane == 0 && tcount < 4) a.trace[((long long)blockIdx.x * 8 + wave) * 8 + tcount] = __builtin_readcyclecounter();
        f32x16 acc[2];

        // A step = its six MFMAs, then the operand fetch of the next step and this step's slices of the side work (staging, parked tail), whose
        // time covers the fetch's LDS latency.  (LATE: side work first -- an experiment that runs the two waves of a SIMD half a step apart;
        // measured no faster than lockstep: the kernel is bound by the sum of its issue slots and LDS cycles, not by their order.)
        auto chunk = [&](auto qc, auto lc) {
            constexpr int q = decltype(qc)::value;
            constexpr bool LATE = decltype(lc)::value != 0;
            const int cb1 = cb == 2 ? 0 : cb + 1, cb2 = cb1 == 2 ? 0 : cb1 + 1;
            constexpr bool even = !(q & 1);
            constexpr int nsteps = even ? 5 : 4;
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                if (s >= nsteps) break;
                if constexpr (LATE) {
                    if constexpr (!(ABL & 8)) fetch(s, cb, cb1);
                    if constexpr (!(ABL & 2))
                        if (s < 4) stage_slice(s, cb2);  // chunk c + 2 (requested during chunk c - 1) into the buffer chunk c - 1 was read from; request chunk c + 3
                    if constexpr (!(ABL & 1)) tail_slice(q, s);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // the three term products of weight >= 2^-11, smallest first; the two accumulators alternate
                if constexpr ((ABL & 4) != 0) {
                    asm volatile("" ::"v"(bt[0]), "v"(bt[1]), "v"(at[0][0]), "v"(at[0][1]), "v"(at[1][0]), "v"(at[1][1]));
                    if (q == 0 && s == 0) acc[0] = zero16, acc[1] = zero16;
                } else {
                acc[0] = C8_MFMA(at[0][1], bt[0], (q == 0 && s == 0) ? zero16 : acc[0]);
                acc[1] = C8_MFMA(at[1][1], bt[0], (q == 0 && s == 0) ? zero16 : acc[1]);
                acc[0] = C8_MFMA(at[0][0], bt[1], acc[0]);
                acc[1] = C8_MFMA(at[1][0], bt[1], acc[1]);
                acc[0] = C8_MFMA(at[0][0], bt[0], acc[0]);
                acc[1] = C8_MFMA(at[1][0], bt[0], acc[1]);
                }
                if constexpr (!LATE) {
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (!(ABL & 8)) {
                        if (s + 1 < nsteps) fetch(s + 1, cb, cb1);
                        else fetch(0, cb1, cb2);         // the next chunk's first step (its buffer was complete before the previous barrier)
                    }
                    if constexpr (!(ABL & 2))
                        if (s < 4) stage_slice(s, cb2);
                    if constexpr (!(ABL & 1)) tail_slice(q, s);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            cb = cb1;
            if constexpr (!(ABL & 32)) __syncthreads();
        };
        if (wave >= 4 && (ABL & 16)) {     // (ABL 16: the upper four waves run the steps in the opposite order -- measured no faster, ablation only)
            constexpr std::integral_constant<int, 1> late{};
            chunk(std::integral_constant<int, 0>{}, late);
            chunk(std::integral_constant<int, 1>{}, late);
            chunk(std::integral_constant<int, 2>{}, late);
            chunk(std::integral_constant<int, 3>{}, late);
            chunk(std::integral_constant<int, 4>{}, late);
            chunk(std::integral_constant<int, 5>{}, late);
            chunk(std::integral_constant<int, 6>{}, late);
            chunk(std::integral_constant<int, 7>{}, late);
        } else {
            constexpr std::integral_constant<int, 0> early{};
            chunk(std::integral_constant<int, 0>{}, early);
            chunk(std::integral_constant<int, 1>{}, early);
            chunk(std::integral_constant<int, 2>{}, early);
            chunk(std::integral_constant<int, 3>{}, early);
            chunk(std::integral_constant<int, 4>{}, early);
            chunk(std::integral_constant<int, 5>{}, early);
            chunk(std::integral_constant<int, 6>{}, early);
            chunk(std::integral_constant<int, 7>{}, early);
        }

        // park this tile's row: g = ReLU(conv 2^-kx 2^-kw + b)
        {
            const int oy = h0 + wave, ox = w0 + l31;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const f32x4 bc = tab4[16 + q];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int R = 4 * q + i;
                    const float v = acc[R >> 4][R & 15] * un_conv + bc[i];
                    g[R] = v > 0.f ? v : 0.f;
                }
            }
            const bool inside = oy < a.H && ox < a.W;
            pd_off = inside ? (unsigned)((oy * a.W + ox) * 32 + lhi * 16) : C8_OOR;
            pd_poff = inside ? (unsigned)((oy * a.W + ox) * 8) + (unsigned)lhi * (unsigned)(plane * 16) : C8_OOR;
            pd_b = b;
        }
    }
    if (a.trace && lane == 0) a.trace[((long long)blockIdx.x * 8 + wave) * 8 + (tcount < 4 ? tcount : 4)] = __builtin_readcyclecounter();
    // the last tile's tail has no convolution to hide under
    if constexpr (!(ABL & 1))
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        tail_slice(j, 0);
        tail_slice(j, 2);
    }
    if (a.trace && lane == 0) a.trace[((long long)blockIdx.x * 8 + wave) * 8 + 5] = __builtin_readcyclecounter();
}

// ---- [B][9][H][W][2] tap products -> eta + permute(conv3x3_reppad(h) + b_final) (rim_block.py:240-246) -------------------------------------------
__global__ __launch_bounds__(256) void k_l2c8_gather(const float2* __restrict__ P, const float* __restrict__ bias, const float* __restrict__ eta,
                                                     float* __restrict__ out, int H, int W) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
    if (x >= W || y >= H) return;
    const long long plane = (long long)H * W;
    const float2* pb = P + (long long)b * 9 * plane;
    float s0 = bias ? bias[0] : 0.f, s1 = bias ? bias[1] : 0.f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        int yy = y + dy - 1;
        yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            int xx = x + dx - 1;
            xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
            const float2 v = pb[(long long)(dy * 3 + dx) * plane + (long long)yy * W + xx];
            s0 += v.x;
            s1 += v.y;
        }
    }
    const long long e = ((long long)b * plane + (long long)y * W + x) * 2;
    float2 v = eta ? *reinterpret_cast<const float2*>(eta + e) : make_float2(0.f, 0.f);
    v.x += s0, v.y += s1;
    *reinterpret_cast<float2*>(out + e) = v;
}

// ---- NCHW <-> CB8 ------------------------------------------------------------------------------------------------------------------------------
// one thread per (pixel, block of 8 channels); to_cb8 = 1: y[b][q][p][j] = x[b][8 q + j][p], else the inverse
__global__ void k_cb8_convert(const float* __restrict__ x, float* __restrict__ y, long long plane, int nblk, int to_cb8) {
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int q = blockIdx.y, b = blockIdx.z;
    if (p >= plane) return;
    const long long base = ((long long)b * nblk + q) * 8 * plane;
    if (to_cb8) {
        f32x4 lo, hi;
#pragma unroll
        for (int j = 0; j < 4; ++j) lo[j] = x[base + j * plane + p], hi[j] = x[base + (4 + j) * plane + p];
        f32x4* d = reinterpret_cast<f32x4*>(y + base + p * 8);
        d[0] = lo, d[1] = hi;
    } else {
        const f32x4* s = reinterpret_cast<const f32x4*>(x + base + p * 8);
        const f32x4 lo = s[0], hi = s[1];
#pragma unroll
        for (int j = 0; j < 4; ++j) y[base + j * plane + p] = lo[j], y[base + (4 + j) * plane + p] = hi[j];
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

static int l2c8_ncu() {
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 256;
        ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    return ncu;
}

// h_new = ReLU(W_ih ReLU(conv3x3_d2(replicate_pad(x)) + b_conv) + b_ih + hh * h_prev) on channel-blocked states (x, h_prev, h_new: [B][8][H][W][8];
// h_prev NULL = the zero state, h_new may be h_prev), and -- taps9 not NULL -- taps9[b][tap][y][x][co] = sum_c w_final[co][c][tap] h_new[b][c][y][x].
// packed = mrx_rim_layer2_f16_pack(w_conv, w_ih, w_final); xmax: device float >= max |x| (kept by mrx_rim_layer1_cb8).
extern "C" int mrx_rim_layer2_cb8(const float* x, const float* packed, const float* b_conv, const float* b_ih, const float* hh, const float* h_prev,
                                  float* h_new, float* taps9, const float* xmax, int B, int H, int W, void* stream) {
    MRX_REQUIRE(x && packed && hh && h_new && xmax, MRX_EINVAL, "mrx_rim_layer2_cb8: null pointer");
    MRX_REQUIRE(B >= 0 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_rim_layer2_cb8: bad dims");
    MRX_REQUIRE((long long)H * W * C8_F * 4 < 0x7fffffffLL, MRX_EUNSUP, "mrx_rim_layer2_cb8: %d x %d images exceed the 2 GB buffer range of one sample", H, W);
    if (B == 0) return MRX_OK;
    L2c8Args a;
    a.x = x, a.packed = reinterpret_cast<const u32x4*>(packed), a.b_conv = b_conv, a.b_ih = b_ih, a.hh = hh, a.hprev = h_prev, a.hnew = h_new;
    a.P = taps9, a.xmax = reinterpret_cast<const unsigned*>(xmax);
    a.B = B, a.H = H, a.W = W, a.tiles_x = mrx_cdiv(W, C8_TW), a.ntiles = a.tiles_x * mrx_cdiv(H, C8_TH);
    static bool attr_done = false;   // once: keeps launches legal under hipGraph capture
    if (!attr_done) {
        MRX_HIP(hipFuncSetAttribute((const void*)k_rim_layer2_cb8<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C8_LDS));
        MRX_HIP(hipFuncSetAttribute((const void*)k_rim_layer2_cb8<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C8_LDS));
        attr_done = true;
    }
    const long long total = (long long)a.ntiles * a.B;
    const int ncu = l2c8_ncu();
    const int grid = (int)(total < ncu ? total : ncu);
    a.trace = nullptr;
    static unsigned long long* d_trace = nullptr;
    if (getenv("MRX_L2C8_TRACE")) {
        if (!d_trace) (void)hipMalloc((void**)&d_trace, sizeof(unsigned long long) * 512 * 8 * 8);
        (void)hipMemsetAsync(d_trace, 0, sizeof(unsigned long long) * 512 * 8 * 8, (hipStream_t)stream);
        a.trace = d_trace;
    }
#ifdef MRX_PROBE
    if (taps9 && getenv("MRX_L2C8_ABL")) {
        switch (atoi(getenv("MRX_L2C8_ABL"))) {
#define C8_ABL_CASE(N)                                                                                                                   \
    case N:                                                                                                                              \
        (void)hipFuncSetAttribute((const void*)k_rim_layer2_cb8<true, N>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)C8_LDS);       \
        hipLaunchKernelGGL((k_rim_layer2_cb8<true, N>), dim3(grid), dim3(C8_NT), C8_LDS, (hipStream_t)stream, a);                         \
        return MRX_OK;
            C8_ABL_CASE(1) C8_ABL_CASE(2) C8_ABL_CASE(3) C8_ABL_CASE(4) C8_ABL_CASE(7) C8_ABL_CASE(11) C8_ABL_CASE(16) C8_ABL_CASE(19) C8_ABL_CASE(32) C8_ABL_CASE(35)
#undef C8_ABL_CASE
            default: break;
        }
    }
#endif
    if (!taps9)
        hipLaunchKernelGGL(k_rim_layer2_cb8<false>, dim3(grid), dim3(C8_NT), C8_LDS, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(k_rim_layer2_cb8<true>, dim3(grid), dim3(C8_NT), C8_LDS, (hipStream_t)stream, a);
    if (a.trace) {
        (void)hipStreamSynchronize((hipStream_t)stream);
        static unsigned long long h[512 * 8 * 8];
        (void)hipMemcpy(h, d_trace, sizeof(h), hipMemcpyDeviceToHost);
        double tile = 0, last = 0;
        long nt = 0, nl = 0;
        for (int i = 0; i < grid * 8; ++i) {
            const unsigned long long* r = &h[(size_t)i * 8];
            for (int k = 0; k < 4; ++k)
                if (r[k] && r[k + 1] && k + 1 <= 4) tile += (double)(r[k + 1] - r[k]), ++nt;
            if (r[5]) {
                int k = 4;
                while (k > 0 && !r[k]) --k;
                if (r[k]) last += (double)(r[5] - r[k]), ++nl;
            }
        }
        fprintf(stderr, "[l2c8-trace] %ld wave-tiles, %.0f cycles per tile (chunk loop with the previous tile's tail inside); the last tail alone %.0f\n",
                nt, nt ? tile / nt : 0.0, nl ? last / nl : 0.0);
    }
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// eta_out [B,H,W,2] = eta + permute(conv3x3_reppad(h_new, w_final) + b_final) from mrx_rim_layer2_cb8's tap products
extern "C" int mrx_rim_final_gather9(const float* taps9, const float* b_final, const float* eta, float* eta_out, int B, int H, int W, void* stream) {
    MRX_REQUIRE(taps9 && eta_out, MRX_EINVAL, "mrx_rim_final_gather9: null pointer");
    MRX_REQUIRE(B >= 0 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_rim_final_gather9: bad dims");
    if (B == 0) return MRX_OK;
    hipLaunchKernelGGL(k_l2c8_gather, dim3(mrx_cdiv(W, 64), mrx_cdiv(H, 4), B), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float2*>(taps9),
                       b_final, eta, eta_out, H, W);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
