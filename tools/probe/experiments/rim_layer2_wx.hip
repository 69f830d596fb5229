// rim_layer2_wx.hip -- the second RIM layer (ConvNonlinear 3x3 dilation 2, 64 -> 64, replicate padding, ReLU + IndRNNCell 1x1 64 -> 64 + the channel
// contraction of the final 3x3 64 -> 2 convolution: reference models/rim/conv_layers.py:121-123, rnn_cells.py:384-391, rim_block.py:233-246) with the
// dilated convolution in a WINOGRAD F(2, 3) form ALONG X on two-term fp16 operands -- the arithmetic of k_rim_layer2_sb<F16, CB8, FAST>
// (rim_layer2_sb.hip) with 288 instead of 432 convolution MFMAs per wave and tile.  Round 5: the headline loop runs at the chip's power limit
// (DESIGN.md 4.2), so fewer matrix FLOPs per pixel is the lever; the two-dimensional F(2x2, 3x3) form does not fit the LDS.
//
// Along x the dilation-2 taps of an output pixel touch pixels of its own parity only: a 32-pixel row is two interleaved unit-dilation rows of 16
// ("classes" c = x & 1), each cut into 8 Winograd tiles t of two outputs u = 2t, 2t + 1 from four inputs d0..d3 = in[2t - 1 .. 2t + 2] (class
// coordinates):      V0 = d0 - d2, V1 = d1 + d2, V2 = d2 - d1, V3 = d1 - d3          (fp32, BEFORE the two-term split; |V| <= 2 max |x|)
//                    U0 = g0, U1 = (g0 + g1 + g2) / 2, U2 = (g0 - g1 + g2) / 2, U3 = g2   (per cout, cin, ky; float64 at pack time; |U| <= 1.5 max |w|)
//                    M_p[co][tile] = sum over ci, ky of U_p[ky][co][ci] V_p[ci][row + 2 (ky - 1)][tile]        (y stays direct, dilated)
//                    out[2t] = M0 + M1 + M2,   out[2t + 1] = M1 - M2 - M3.
// 12 (position, ky) slots instead of 9 taps, two outputs per slot: 4 x 12 K-steps x 6 MFMAs = 288 per wave and 16 x 32 tile.  A wave owns two image
// rows = 32 Winograd tiles (one MFMA column block) x 64 couts x 4 positions = 8 accumulators; the odd count of three ky per position pairs across chunks
// like the ninth tap of the direct form (the ky = 2 slot of an even chunk shares its MFMA step with the next chunk's).  After the output transform the
// accumulators are two pixels per lane -- (row 2 wave + (l31 >> 4), x = 4 (l31 & 7) + ((l31 >> 3) & 1) + 2 j), j = 0, 1 -- and the row tails of the direct
// kernel (1x1 stage, epilogue, tap stage: both "rows" together) run on them unchanged but for that pixel map.
// Round-off against float64 (tools/probe/wino_f16x2_error.py, numpy emulation): 2.3e-7 against 1.3e-7 for the direct two-term form and 1.2e-7 for fp32.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "mrx_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

#define WX_NT 512
#define WX_TH 16
#define WX_TW 32
#define WX_F 64
#define WX_NCH 8
#define WX_ROWS (WX_TH + 4)                    // halo'd rows of a tile
#define WX_ITEMS (WX_ROWS * 16)                // (row, class, Winograd tile) items of a chunk: 320
#define WX_PSTR (WX_ITEMS + 1)                 // 16-byte slots of one (position, term) plane + the dummy that idle threads write
#define WX_XBUF (8 * WX_PSTR)                  // 4 positions x 2 terms
#define WX_WFULL (4 * 2 * 2 * 64)              // A operands of the (ky 0 | ky 1) steps: [position][term][cout block][lane]
#define WX_WCH (WX_WFULL + 4 * 2 * 2 * 32)     // + the ky = 2 steps (lower half-wave only): 1536 slots = 24 KB per chunk
#define WX_WIH 1536                            // pack slots of the 1x1 section (1024 used; the layout of mrx_rim_layer2_f16_pack)
#define WX_WP 768                              // pack slots of the final-convolution section (512 used)
#define WX_PK_TAIL (WX_NCH * WX_WCH)
#define WX_PACK_U4 (WX_PK_TAIL + WX_WIH + WX_WP + 1)   // + the header: scale exponents (conv, 1x1, final)
// LDS (bytes): 1x1 operands 16384 | final-conv operands 8192 | tables 1024 | 2 x chunk weights 24576 | 2 x transformed planes 41088  = 156 928
#define WX_LWIH 1024
#define WX_LWP 512
#define WX_OFF_TAB ((WX_LWIH + WX_LWP) * 16)
#define WX_OFF_W (WX_OFF_TAB + 1024)
#define WX_OFF_X (WX_OFF_W + 2 * WX_WCH * 16)
#define WX_LDS (WX_OFF_X + 2 * WX_XBUF * 16)

struct L2wxArgs {
    const float* x;        // [B][8][H][W][8]
    const u32x4* packed;   // mrx_rim_layer2_wx_pack
    const float* b_conv;   // [64] or null
    const float* b_ih;     // [64] or null
    const float* hh;       // [64]
    const float* hprev;    // [B][8][H][W][8] or null
    float* hnew;           // [B][8][H][W][8]
    float* P;              // [B][18][H][W] or null
    const unsigned* xmax;  // bits of an upper bound of max |x|
    int B, H, W, tiles_x, ntiles;
    unsigned long long* trace;   // probe builds (MRX_WX_TRACE): cycle stamps [workgroup][wave][4] of the workgroup's second tile
};

__device__ __forceinline__ void wx_split2h(float a, float b, unsigned& p1, unsigned& p2) {
    const f16x2 h = {(_Float16)a, (_Float16)b};
    const float ra = a - (float)h.x, rb = b - (float)h.y;     // exact
    const f16x2 l = {(_Float16)ra, (_Float16)rb};
    p1 = __builtin_bit_cast(unsigned, h);
    p2 = __builtin_bit_cast(unsigned, l);
}
// the two terms of (a s, b s), s a power of two, in four instructions (rim_layer2_sb.hip: s2_split2h_scaled)
__device__ __forceinline__ void wx_split2h_scaled(float a, float b, float s, unsigned& p1, unsigned& p2) {
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=&v"(p1) : "v"(a), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(p1) : "v"(b), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=&v"(p2) : "v"(a), "v"(s), "v"(p1));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(p2) : "v"(b), "v"(s), "v"(p1));
}
__device__ __forceinline__ float wx_pow2(int e) {
    e = e < -120 ? -120 : (e > 120 ? 120 : e);
    return __uint_as_float((unsigned)(127 + e) << 23);
}
__host__ __device__ __forceinline__ int wx_scale_exp(unsigned bits) {     // k with bound * 2^k in [2^14, 2^15) (0 for a zero / non-finite bound)
    const int ex = (int)((bits >> 23) & 0xffu);
    return (ex == 0 || ex == 255) ? 0 : 14 - (ex - 127);
}
__host__ __device__ constexpr int wx_chan(int R, int half) { return 32 * (R >> 4) + (R & 3) + 8 * ((R & 15) >> 2) + 4 * half; }
__device__ __forceinline__ int wx_pixel_exp(float m) {
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    return wx_scale_exp(__float_as_uint(m));
}

// ---- pack -------------------------------------------------------------------------------------------------------------------------------------
// header: [0] exponent of the TRANSFORMED conv weights (max |w| 2^(k + 1) in [2^14, 2^15): |U| <= 1.5 max |w| stays below 2^15), [1] 1x1, [2] final conv
__global__ void k_l2wx_wscale(const float* __restrict__ w, const float* __restrict__ w_ih, const float* __restrict__ w_final, u32x4* __restrict__ out) {
    __shared__ float red[256];
    int ex[3] = {0, 0, 0};
    for (int which = 0; which < 3; ++which) {
        const float* p = which == 0 ? w : (which == 1 ? w_ih : w_final);
        const int n = which == 0 ? WX_F * WX_F * 9 : (which == 1 ? WX_F * WX_F : 2 * WX_F * 9);
        float m = 0.f;
        if (p)
            for (int i = threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(p[i]));
        red[threadIdx.x] = m;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
            __syncthreads();
        }
        ex[which] = wx_scale_exp(__float_as_uint(red[0]));
        __syncthreads();
    }
    if (threadIdx.x == 0) out[WX_PACK_U4 - 1] = u32x4{(unsigned)(ex[0] - 1), (unsigned)ex[1], (unsigned)ex[2], 0u};
}
// conv : out[q * WX_WCH + ((p*2 + t)*2 + blk)*64 + lane][j]        = term_t( U_p[ky = lane/32][32 blk + lane%32][8 q + j] 2^k )
//        out[q * WX_WCH + WX_WFULL + ((p*2 + t)*2 + blk)*32 + l][j] = term_t( U_p[ky = 2][32 blk + l][8 q + j] 2^k )
// 1x1 / final-conv sections: the layout of k_l2f16_pack (rim_layer2_sb.hip)
__global__ void k_l2wx_pack(const float* __restrict__ w, const float* __restrict__ w_ih, const float* __restrict__ w_final, u32x4* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= WX_PACK_U4 - 1) return;
    float v[8];
    int t;
    if (i < WX_PK_TAIL) {
        const double sw = (double)wx_pow2((int)out[WX_PACK_U4 - 1][0]);
        const int q = i / WX_WCH;
        int r = i - q * WX_WCH, l, blk, pos, ky;
        if (r < WX_WFULL) {
            const int lane = r & 63;
            r >>= 6;
            blk = r & 1, r >>= 1;
            t = r & 1, pos = r >> 1, l = lane & 31, ky = lane >> 5;
        } else {
            r -= WX_WFULL;
            l = r & 31, blk = (r >> 5) & 1, t = (r >> 6) & 1, pos = r >> 7, ky = 2;
        }
        const int o = 32 * blk + l;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float* g = w + ((long long)o * WX_F + 8 * q + j) * 9 + 3 * ky;
            const double g0 = g[0], g1 = g[1], g2 = g[2];
            const double u = pos == 0 ? g0 : (pos == 1 ? 0.5 * (g0 + g1 + g2) : (pos == 2 ? 0.5 * (g0 - g1 + g2) : g2));
            v[j] = (float)(u * sw);
        }
    } else if (i < WX_PK_TAIL + WX_WIH) {
        int r = i - WX_PK_TAIL;
        if (r >= 4 * 2 * 2 * 64) {
            out[i] = u32x4{0u, 0u, 0u, 0u};
            return;
        }
        const float sw = wx_pow2((int)out[WX_PACK_U4 - 1][1]);
        const int lane = r & 63;
        r >>= 6;
        const int blk = r & 1;
        r >>= 1;
        t = r & 1;
        const int s = r >> 1, o = 32 * blk + (lane & 31);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = w_ih ? w_ih[o * WX_F + wx_chan(8 * s + j, lane >> 5)] * sw : 0.f;
    } else {
        int r = i - WX_PK_TAIL - WX_WIH;
        if (r >= 4 * 2 * 64) {
            out[i] = u32x4{0u, 0u, 0u, 0u};
            return;
        }
        const float sw = wx_pow2((int)out[WX_PACK_U4 - 1][2]);
        const int lane = r & 63;
        r >>= 6;
        t = r & 1;
        const int s = r >> 1, m = lane & 31;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            v[j] = (w_final && m < 18) ? w_final[((long long)(m & 1) * WX_F + wx_chan(8 * s + j, lane >> 5)) * 9 + (m >> 1)] * sw : 0.f;
    }
    unsigned p[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        unsigned p1, p2;
        wx_split2h(v[2 * k], v[2 * k + 1], p1, p2);
        p[k] = t == 0 ? p1 : p2;
    }
    out[i] = u32x4{p[0], p[1], p[2], p[3]};
}

// ---- the layer ----------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WX_NT, 1) void k_rim_layer2_wx8(L2wxArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_wx[];
    const u32x4* Wih = reinterpret_cast<const u32x4*>(smem_wx);                       // 1x1 operands, then the final convolution's
    float* tabl = reinterpret_cast<float*>(smem_wx + WX_OFF_TAB);                      // hh, b_conv (scaled), b_ih in register order [half][R]
    u32x4* Wc = reinterpret_cast<u32x4*>(smem_wx + WX_OFF_W);                          // [2][WX_WCH]
    u32x4* Xp = reinterpret_cast<u32x4*>(smem_wx + WX_OFF_X);                          // [2][4 positions][2 terms][WX_PSTR]
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int tid = (int)threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const long long plane = (long long)a.H * a.W;
    const int total = a.ntiles * a.B;

    // once per workgroup: 1x1 / final-conv operands (compacted: 1024 + 512 of the pack's 1536 + 768 slots) and the tables
    for (int i = tid; i < WX_LWIH + WX_LWP; i += WX_NT)
        reinterpret_cast<u32x4*>(smem_wx)[i] = a.packed[WX_PK_TAIL + (i < WX_LWIH ? i : WX_WIH + (i - WX_LWIH))];
    // x is multiplied by 2^(kx - 1) (the transformed values are sums of two: |V| <= 2 max |x|), the transformed weights were by 2^kw
    const int kx = wx_scale_exp(a.xmax[0]) - 1, kw = (int)a.packed[WX_PACK_U4 - 1][0];
    const float sx = wx_pow2(kx), unx = wx_pow2(-kx), unw = wx_pow2(-kw);
    const float unwi = wx_pow2(-(int)a.packed[WX_PACK_U4 - 1][1]), unwp = wx_pow2(-(int)a.packed[WX_PACK_U4 - 1][2]);
    if (tid < 64) {
        const int tc = wx_chan(tid >> 1, tid & 1);
        const int ti = (tid & 1) * 32 + (tid >> 1);
        tabl[ti] = a.hh ? a.hh[tc] : 0.f;
        tabl[64 + ti] = a.b_conv ? a.b_conv[tc] * (sx * wx_pow2(kw)) : 0.f;   // the accumulators start from the bias in their own scaled domain (exact)
        tabl[128 + ti] = a.b_ih ? a.b_ih[tc] : 0.f;
    }

    // ---- staging: thread i < 320 owns item (row i >> 4, class (i >> 3) & 1, tile i & 7) of every chunk -- four fp32 pixels of 8 channels in, the four
    // transformed positions out (two fp16 terms each); every thread copies three weight operands.  Two chunks ahead of the MFMAs, across tile boundaries.
    int st_t = blockIdx.x, st_q = 0;
    unsigned goff32[4];                              // byte offsets of the item's four pixels inside the sample's channel-blocked tensor (chunk 0)
    __amdgpu_buffer_rsrc_t st_rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (unsigned)(plane * (WX_F * 4)), 0x00020000);
    const __amdgpu_buffer_rsrc_t st_rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(a.packed), 0, (unsigned)(WX_PK_TAIL * 16), 0x00020000);
    const bool item = tid < WX_ITEMS;
    const unsigned st_slot16 = (unsigned)(item ? tid : WX_ITEMS) * 16u, st_tid16 = (unsigned)tid * 16u;
    auto st_coords = [&]() {
        const int tq = st_t < total ? st_t : total - 1;      // (beyond the last tile the pipeline keeps requesting the last tile: in range, nobody reads it)
        const int tt = (int)mrx_xcd_band(tq, total);
        const int b = tt / a.ntiles, tile = tt - b * a.ntiles, ty0 = tile / a.tiles_x;
        const int h0 = ty0 * WX_TH, w0 = (tile - ty0 * a.tiles_x) * WX_TW;
        st_rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x) + (long long)b * WX_F * plane, 0, (unsigned)(plane * (WX_F * 4)), 0x00020000);
        const int i = item ? tid : 0, r = i >> 4, c = (i >> 3) & 1, t = i & 7;
        int gy = h0 + r - 2;
        gy = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy);                        // replicate border = clamp (conv_layers.py:72-76)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int gx = w0 + 4 * t + 2 * k + c - 2;
            gx = gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx);
            goff32[k] = item ? (unsigned)(gy * a.W + gx) * 32u : 0x80000000u;   // (idle threads: out of range, the loads return zeros)
        }
    };
    float xr[4][8];
    u32x4 wr[3];
    auto ns_load_x = [&](int k, int half) {
        const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(st_rx, goff32[k] + 16u * half, (unsigned)st_q * (unsigned)(plane * 32), 0);
        xr[k][4 * half] = __uint_as_float(u[0]), xr[k][4 * half + 1] = __uint_as_float(u[1]);
        xr[k][4 * half + 2] = __uint_as_float(u[2]), xr[k][4 * half + 3] = __uint_as_float(u[3]);
    };
    auto ns_load_w = [&](int v) { wr[v] = __builtin_amdgcn_raw_buffer_load_b128(st_rw, st_tid16, (unsigned)(st_q * WX_WCH + v * WX_NT) * 16u, 0); };
    auto ns_advance = [&]() {
        if (++st_q == WX_NCH) {
            st_q = 0;
            st_t += gridDim.x;
            st_coords();
        }
    };
    // position p of the item: transform (fp32), split, two LDS writes
    auto ns_commit = [&](int buf, int p) {
        unsigned p1[4], p2[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v0, v1;
            const int j0 = 2 * k, j1 = 2 * k + 1;
            if (p == 0) v0 = xr[0][j0] - xr[2][j0], v1 = xr[0][j1] - xr[2][j1];
            else if (p == 1) v0 = xr[1][j0] + xr[2][j0], v1 = xr[1][j1] + xr[2][j1];
            else if (p == 2) v0 = xr[2][j0] - xr[1][j0], v1 = xr[2][j1] - xr[1][j1];
            else v0 = xr[1][j0] - xr[3][j0], v1 = xr[1][j1] - xr[3][j1];
            wx_split2h_scaled(v0, v1, sx, p1[k], p2[k]);
        }
        unsigned char* base = smem_wx + WX_OFF_X + (buf * WX_XBUF + (p * 2) * WX_PSTR) * 16 + st_slot16;
        *reinterpret_cast<u32x4*>(base) = u32x4{p1[0], p1[1], p1[2], p1[3]};
        *reinterpret_cast<u32x4*>(base + WX_PSTR * 16) = u32x4{p2[0], p2[1], p2[2], p2[3]};
    };
    auto ns_write_w = [&](int buf, int v) { *reinterpret_cast<u32x4*>(smem_wx + WX_OFF_W + (buf * WX_WCH + v * WX_NT) * 16 + st_tid16) = wr[v]; };
    st_coords();
    // At the top of every tile chunks 0 AND 1 are in LDS and nothing is in flight: the registers of the staging pipeline (32 + 12) are dead across the row
    // tails, which need every register (chunk 1 of the next tile is requested behind the last MFMAs of this one and committed at the head of its tails).
    for (int pre = 0; pre < 2; ++pre) {
#pragma unroll
        for (int k = 0; k < 4; ++k) ns_load_x(k, 0), ns_load_x(k, 1);
#pragma unroll
        for (int v = 0; v < 3; ++v) ns_load_w(v);
        ns_advance();
#pragma unroll
        for (int p = 0; p < 4; ++p) ns_commit(pre, p);
#pragma unroll
        for (int v = 0; v < 3; ++v) ns_write_w(pre, v);
    }
    __syncthreads();

    // the lane's Winograd tile inside the wave's column block: row 2 wave + (l31 >> 4), item index l31 & 15 of that row
    const int brow = 2 * wave + (l31 >> 4), bidx = l31 & 15;
    for (int tI = blockIdx.x; tI < total; tI += gridDim.x) {
        const int tt = (int)mrx_xcd_band(tI, total);
        const int b = tt / a.ntiles, tile = tt - b * a.ntiles, ty0 = tile / a.tiles_x;
        const int h0 = ty0 * WX_TH, w0 = (tile - ty0 * a.tiles_x) * WX_TW;
        f32x16 acc[4][2];                            // [position][cout block]
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[p][ct][r] = p == 1 ? tabl[64 + lhi * 32 + ct * 16 + r] : 0.f;   // M1 enters both outputs with +1: it carries the bias

        auto chunk = [&](const int q, auto FIRST) {
            constexpr bool first = decltype(FIRST)::value;   // chunk 0: chunk 1 is in LDS already (nothing to commit)
            const int cb = q & 1, nb = cb ^ 1;
            const bool even = !(q & 1);
            u32x4 bt[2], at[2][2];                   // [term] / [cout block][term]
            auto fetch = [&](int p, int second) {    // second = 0: the (ky 0 | ky 1) step of position p; 1: the paired ky = 2 step (this chunk | the next)
                if (!second) {
                    const u32x4* xw = Xp + cb * WX_XBUF + (p * 2) * WX_PSTR + (brow + 2 * lhi) * 16 + bidx;
                    const u32x4* wl = Wc + cb * WX_WCH + (p * 2) * 2 * 64 + lane;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        bt[k] = xw[k * WX_PSTR];
                        at[0][k] = wl[(k * 2 + 0) * 64];
                        at[1][k] = wl[(k * 2 + 1) * 64];
                    }
                } else {
                    const int bf = (q + lhi) & 1;
                    const u32x4* xw = Xp + bf * WX_XBUF + (p * 2) * WX_PSTR + (brow + 4) * 16 + bidx;
                    const u32x4* w8 = Wc + bf * WX_WCH + WX_WFULL + (p * 2) * 2 * 32 + l31;
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        bt[k] = xw[k * WX_PSTR];
                        at[0][k] = w8[(k * 2 + 0) * 32];
                        at[1][k] = w8[(k * 2 + 1) * 32];
                    }
                }
            };
            auto mfma6 = [&](int p, auto&& side) {   // the three term products (smallest first) x two cout blocks; side(m) runs behind MFMA m
                constexpr int TA_[3] = {1, 0, 0}, TB_[3] = {0, 1, 0};
#pragma unroll
                for (int m = 0; m < 6; ++m) {
                    const int ct = m & 1, pr = m >> 1;
                    acc[p][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, at[ct][TA_[pr]]), __builtin_bit_cast(f16x8, bt[TB_[pr]]), acc[p][ct], 0, 0, 0);
                    side(m);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                fetch(p, 0);
                // chunk q + 1 is transformed and written behind the MFMAs of positions 0 and 1, chunk q + 2 requested behind those of positions 2 and 3
                mfma6(p, [&](int m) {
#ifdef MRX_WX_ABL_NOSTAGE                           // (timing variant: no staging inside the chunk loop -- garbage results)
                    return;
#endif
                    if (p == 0) {
                        if constexpr (!first) {
                            if (m == 0) ns_commit(nb, 0);
                            else if (m == 2) ns_commit(nb, 1);
                            else if (m == 4) ns_commit(nb, 2);
                        }
                    } else if (p == 1) {
                        if constexpr (!first) {
                            if (m == 0) ns_commit(nb, 3);
                            else if (m >= 2 && m < 5) ns_write_w(nb, m - 2);
                        }
                    } else if (p == 2) {
                        if (m < 4) ns_load_x(m, 0);
                        else if (m == 4) ns_load_x(0, 1), ns_load_x(1, 1);
                        else ns_load_x(2, 1), ns_load_x(3, 1);
                    } else {
                        if (m < 3) ns_load_w(m);
                        else if (m == 3) ns_advance();
                    }
                });
            }
            if (even) {
                __syncthreads();                     // every thread's commit of chunk q + 1 is visible: its ky = 2 rows are read below
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    fetch(p, 1);
                    mfma6(p, [&](int) {});
                }
            }
            __syncthreads();
        };
        chunk(0, std::true_type{});
        for (int q = 1; q < WX_NCH; ++q) chunk(q, std::false_type{});

        // ---- output transform: the lane's two pixels (j = 0, 1) of 64 channels, still in the scaled domain ---------------------------------------
        f32x16 out[2][2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float m0 = acc[0][ct][r], m1 = acc[1][ct][r], m2 = acc[2][ct][r], m3 = acc[3][ct][r];
                out[0][ct][r] = (m0 + m1) + m2;
                out[1][ct][r] = (m1 - m2) - m3;
            }
        // the next tile's chunk 1 (requested behind the last MFMAs above) into buffer 1 -- free since the barrier that closed chunk 7
#pragma unroll
        for (int p = 0; p < 4; ++p) ns_commit(1, p);
#pragma unroll
        for (int v = 0; v < 3; ++v) ns_write_w(1, v);
        // pixel (j, lane): row h0 + brow, column w0 + 4 (l31 & 7) + ((l31 >> 3) & 1) + 2 j
        const int oy = h0 + brow, ox0 = w0 + 4 * (l31 & 7) + ((l31 >> 3) & 1);
        float hp[2][32];
        auto load_hp = [&](int j) {
            if (!a.hprev) {
#pragma unroll
                for (int R = 0; R < 32; ++R) hp[j][R] = 0.f;
                return;
            }
            const int ox = ox0 + 2 * j;
            const int cy = oy < a.H ? oy : a.H - 1, cx = ox < a.W ? ox : a.W - 1;
            const float* hb = a.hprev + (long long)b * WX_F * plane + ((long long)cy * a.W + cx) * 8 + 4 * lhi;
#pragma unroll
            for (int qq = 0; qq < 8; ++qq) {
                const float4 u = *reinterpret_cast<const float4*>(hb + (long long)qq * plane * 8);
                hp[j][4 * qq] = u.x, hp[j][4 * qq + 1] = u.y, hp[j][4 * qq + 2] = u.z, hp[j][4 * qq + 3] = u.w;
            }
        };
        load_hp(0);
        // ---- 1x1 stage, both pixels together (rim_layer2_sb.hip, the FAST row tails) ------------------------------------------------------------
        float sg[2], ung[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float gm = 0.f;
#pragma unroll
            for (int R = 0; R < 32; ++R) gm = fmaxf(gm, out[j][R >> 4][R & 15]);
            const int kg = wx_pixel_exp(gm);
            sg[j] = wx_pow2(kg), ung[j] = wx_pow2(-kg) * (unx * unw * unwi);
        }
        f32x16 acc2[2][2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[j][ct][r] = 0.f;
        {
            const u32x4* wl = Wih + lane;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                f16x8 b1[2], b2[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    unsigned g1[4], g2[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int R0 = 8 * s + 2 * k, R1 = R0 + 1;
                        float v0 = out[j][R0 >> 4][R0 & 15], v1 = out[j][R1 >> 4][R1 & 15];
                        v0 = v0 > 0.f ? v0 : 0.f;
                        v1 = v1 > 0.f ? v1 : 0.f;
                        wx_split2h_scaled(v0, v1, sg[j], g1[k], g2[k]);
                    }
                    b1[j] = __builtin_bit_cast(f16x8, (u32x4{g1[0], g1[1], g1[2], g1[3]}));
                    b2[j] = __builtin_bit_cast(f16x8, (u32x4{g2[0], g2[1], g2[2], g2[3]}));
                }
                f16x8 a_[2][2];
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) a_[ct][k] = __builtin_bit_cast(f16x8, wl[((s * 2 + k) * 2 + ct) * 64]);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) acc2[j][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_[ct][1], b1[j], acc2[j][ct], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) acc2[j][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_[ct][0], b2[j], acc2[j][ct], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) acc2[j][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_[ct][0], b1[j], acc2[j][ct], 0, 0, 0);
                if (s == 1) load_hp(1);
            }
        }
        const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(a.hnew + (long long)b * WX_F * plane, 0, (unsigned)(plane * (WX_F * 4)), 0x00020000);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ox = ox0 + 2 * j;
            const unsigned offh = (oy < a.H && ox < a.W) ? (unsigned)((((long long)oy * a.W + ox) * 8 + 4 * lhi) * 4) : 0x80000000u;
#pragma unroll
            for (int R = 0; R < 32; ++R) {
                float v = acc2[j][R >> 4][R & 15] * ung[j] + tabl[128 + lhi * 32 + R];
                v += tabl[lhi * 32 + R] * hp[j][R];
                hp[j][R] = v > 0.f ? v : 0.f;
            }
#pragma unroll
            for (int qq = 0; qq < 8; ++qq)
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(hp[j][4 * qq]), __float_as_uint(hp[j][4 * qq + 1]), __float_as_uint(hp[j][4 * qq + 2]),
                                                             __float_as_uint(hp[j][4 * qq + 3])},
                                                       rh, offh + (unsigned)qq * (unsigned)(plane * 32), 0, 0);
        }
        if (a.P) {
            f32x16 accp[2];
            float sh[2], unh[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r) accp[j][r] = 0.f;
                float hm = 0.f;
#pragma unroll
                for (int R = 0; R < 32; ++R) hm = fmaxf(hm, hp[j][R]);
                const int kh = wx_pixel_exp(hm);
                sh[j] = wx_pow2(kh), unh[j] = wx_pow2(-kh) * unwp;
            }
            const u32x4* wp = Wih + WX_LWIH + lane;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                f16x8 b1[2], b2[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    unsigned g1[4], g2[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) wx_split2h_scaled(hp[j][8 * s + 2 * k], hp[j][8 * s + 2 * k + 1], sh[j], g1[k], g2[k]);
                    b1[j] = __builtin_bit_cast(f16x8, (u32x4{g1[0], g1[1], g1[2], g1[3]}));
                    b2[j] = __builtin_bit_cast(f16x8, (u32x4{g2[0], g2[1], g2[2], g2[3]}));
                }
                const f16x8 a1 = __builtin_bit_cast(f16x8, wp[(s * 2 + 0) * 64]);
                const f16x8 a2 = __builtin_bit_cast(f16x8, wp[(s * 2 + 1) * 64]);
#pragma unroll
                for (int j = 0; j < 2; ++j) accp[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b1[j], accp[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 2; ++j) accp[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b2[j], accp[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 2; ++j) accp[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1[j], accp[j], 0, 0, 0);
            }
            const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(a.P + (long long)b * 18 * plane, 0, (unsigned)(plane * (18 * 4)), 0x00020000);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ox = ox0 + 2 * j;
                const bool inside = oy < a.H && ox < a.W;
                const unsigned offp = inside ? (unsigned)((((long long)oy * a.W + ox) + 4ll * lhi * plane) * 4) : 0x80000000u;
                const unsigned offp16 = (inside && !lhi) ? offp : 0x80000000u;       // rows 16, 17: the upper half-wave's 20, 21 do not exist
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(accp[j][r] * unh[j]), rp, offp + (unsigned)((r & 3) + 8 * (r >> 2)) * (unsigned)(plane * 4), 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(accp[j][8] * unh[j]), rp, offp16 + 16u * (unsigned)(plane * 4), 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(accp[j][9] * unh[j]), rp, offp16 + 17u * (unsigned)(plane * 4), 0, 0);
            }
        }
    }
}

// ---- the same layer with FOUR waves of 512 registers (one per SIMD) ---------------------------------------------------------------------------------
// The eight-wave form above reads 4 weight + 2 pixel fragments from LDS for 6 MFMAs (one column block per wave: 8 accumulators = 128 registers) -- 1 KB per MFMA,
// exactly the LDS pipe's 128 bytes per cycle when four SIMDs issue an MFMA every 32 cycles: measured 61.3 us per slice for the MFMA loop alone.  Here a wave owns
// FOUR image rows = two column blocks: 4 positions x 2 cout blocks x 2 column blocks = 16 accumulators (256 registers, in AGPRs), 4 + 4 fragments per 12 MFMAs
// (0.67 KB per MFMA, the direct form's ratio).  One wave per SIMD has nobody to hide behind: every step's fragments are requested one step ahead (two register
// sets), the barriers sit where the next step's operands are already in registers (start of the fourth step, and of the eighth in even chunks), and the
// staging is cut into 46 micro-operations that ride behind individual MFMAs.  Staging roles: thread i owns item i (rows 0..15 of the halo'd tile) and one
// QUARTER (two channels) of an item of rows 16..19 -- 1.25 items per thread for every thread -- and six weight operands.
#define WX4_NT 256
__global__ __launch_bounds__(WX4_NT, 1) void k_rim_layer2_wx4(L2wxArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_wx[];
    const u32x4* Wih = reinterpret_cast<const u32x4*>(smem_wx);
    float* tabl = reinterpret_cast<float*>(smem_wx + WX_OFF_TAB);
    u32x4* Wc = reinterpret_cast<u32x4*>(smem_wx + WX_OFF_W);
    u32x4* Xp = reinterpret_cast<u32x4*>(smem_wx + WX_OFF_X);
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int tid = (int)threadIdx.x, lane = tid & 63, l31 = lane & 31, lhi = lane >> 5;
    const long long plane = (long long)a.H * a.W;
    const int total = a.ntiles * a.B;

    for (int i = tid; i < WX_LWIH + WX_LWP; i += WX4_NT)
        reinterpret_cast<u32x4*>(smem_wx)[i] = a.packed[WX_PK_TAIL + (i < WX_LWIH ? i : WX_WIH + (i - WX_LWIH))];
    const int kx = wx_scale_exp(a.xmax[0]) - 1, kw = (int)a.packed[WX_PACK_U4 - 1][0];
    const float sx = wx_pow2(kx), unx = wx_pow2(-kx), unw = wx_pow2(-kw);
    const float unwi = wx_pow2(-(int)a.packed[WX_PACK_U4 - 1][1]), unwp = wx_pow2(-(int)a.packed[WX_PACK_U4 - 1][2]);
    if (tid < 64) {
        const int tc = wx_chan(tid >> 1, tid & 1);
        const int ti = (tid & 1) * 32 + (tid >> 1);
        tabl[ti] = a.hh ? a.hh[tc] : 0.f;
        tabl[64 + ti] = a.b_conv ? a.b_conv[tc] * (sx * wx_pow2(kw)) : 0.f;
        tabl[128 + ti] = a.b_ih ? a.b_ih[tc] : 0.f;
    }

    // ---- staging ------------------------------------------------------------------------------------------------------------------------------------
    int st_t = blockIdx.x, st_q = 0;
    unsigned goffA[4], goffQ[4];                     // byte offsets (chunk 0) of the four pixels of the thread's item / of its quarter item (+ 8 bytes per quarter)
    __amdgpu_buffer_rsrc_t st_rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (unsigned)(plane * (WX_F * 4)), 0x00020000);
    const __amdgpu_buffer_rsrc_t st_rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(a.packed), 0, (unsigned)(WX_PK_TAIL * 16), 0x00020000);
    const unsigned st_tid16 = (unsigned)tid * 16u;
    const unsigned st_q4 = (unsigned)(256 + (tid >> 2)) * 16u + (unsigned)(tid & 3) * 4u;    // LDS byte offset of the quarter item's two halves inside a plane
    auto st_coords = [&]() {
        const int tq = st_t < total ? st_t : total - 1;
        const int tt = (int)mrx_xcd_band(tq, total);
        const int b = tt / a.ntiles, tile = tt - b * a.ntiles, ty0 = tile / a.tiles_x;
        const int h0 = ty0 * WX_TH, w0 = (tile - ty0 * a.tiles_x) * WX_TW;
        st_rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x) + (long long)b * WX_F * plane, 0, (unsigned)(plane * (WX_F * 4)), 0x00020000);
#pragma unroll
        for (int which = 0; which < 2; ++which) {
            const int i = which ? 256 + (tid >> 2) : tid, r = i >> 4, c = (i >> 3) & 1, t = i & 7;
            int gy = h0 + r - 2;
            gy = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int gx = w0 + 4 * t + 2 * k + c - 2;
                gx = gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx);
                const unsigned o = (unsigned)(gy * a.W + gx) * 32u;
                if (which) goffQ[k] = o + (unsigned)(tid & 3) * 8u;
                else goffA[k] = o;
            }
        }
    };
    float xr[4][8], xq[4][2];
    u32x4 wr[6];
    unsigned c1[4], c2[4];                           // the item's packed terms of the position being transformed
    // micro-operations of the staging pipeline.  Commit side (chunk q + 1 -> LDS buffer `buf`): 0..19 item (per position: four pair splits, then the two
    // writes), 20..23 quarter item (per position), 24..26 weights (two writes each).  Request side (chunk q + 2): 0..7 item loads, 8..11 quarter loads,
    // 12..17 weight loads, 18 advance.
    auto comb = [&](int p, const float x0, const float x1, const float x2, const float x3) {
        return p == 0 ? x0 - x2 : (p == 1 ? x1 + x2 : (p == 2 ? x2 - x1 : x1 - x3));
    };
    auto commit_op = [&](int buf, int i) {
        if (i < 20) {
            const int p = i / 5, k = i - 5 * p;
            if (k < 4) {
                wx_split2h_scaled(comb(p, xr[0][2 * k], xr[1][2 * k], xr[2][2 * k], xr[3][2 * k]),
                                  comb(p, xr[0][2 * k + 1], xr[1][2 * k + 1], xr[2][2 * k + 1], xr[3][2 * k + 1]), sx, c1[k], c2[k]);
            } else {
                unsigned char* base = smem_wx + WX_OFF_X + (buf * WX_XBUF + (p * 2) * WX_PSTR) * 16 + st_tid16;
                *reinterpret_cast<u32x4*>(base) = u32x4{c1[0], c1[1], c1[2], c1[3]};
                *reinterpret_cast<u32x4*>(base + WX_PSTR * 16) = u32x4{c2[0], c2[1], c2[2], c2[3]};
            }
        } else if (i < 24) {
            const int p = i - 20;
            unsigned q1, q2;
            wx_split2h_scaled(comb(p, xq[0][0], xq[1][0], xq[2][0], xq[3][0]), comb(p, xq[0][1], xq[1][1], xq[2][1], xq[3][1]), sx, q1, q2);
            unsigned char* base = smem_wx + WX_OFF_X + (buf * WX_XBUF + (p * 2) * WX_PSTR) * 16 + st_q4;
            *reinterpret_cast<unsigned*>(base) = q1;
            *reinterpret_cast<unsigned*>(base + WX_PSTR * 16) = q2;
        } else if (i < 27) {
            const int v = 2 * (i - 24);
            *reinterpret_cast<u32x4*>(smem_wx + WX_OFF_W + (buf * WX_WCH + v * WX4_NT) * 16 + st_tid16) = wr[v];
            *reinterpret_cast<u32x4*>(smem_wx + WX_OFF_W + (buf * WX_WCH + (v + 1) * WX4_NT) * 16 + st_tid16) = wr[v + 1];
        }
    };
    auto request_op = [&](int i) {
        const unsigned so = (unsigned)st_q * (unsigned)(plane * 32);
        if (i < 8) {
            const int k = i >> 1, half = i & 1;
            const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(st_rx, goffA[k] + 16u * half, so, 0);
            xr[k][4 * half] = __uint_as_float(u[0]), xr[k][4 * half + 1] = __uint_as_float(u[1]);
            xr[k][4 * half + 2] = __uint_as_float(u[2]), xr[k][4 * half + 3] = __uint_as_float(u[3]);
        } else if (i < 12) {
            const int k = i - 8;
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 u = __builtin_amdgcn_raw_buffer_load_b64(st_rx, goffQ[k], so, 0);
            xq[k][0] = __uint_as_float(u[0]), xq[k][1] = __uint_as_float(u[1]);
        } else if (i < 18) {
            const int v = i - 12;
            wr[v] = __builtin_amdgcn_raw_buffer_load_b128(st_rw, st_tid16, (unsigned)(st_q * WX_WCH + v * WX4_NT) * 16u, 0);
        } else if (i == 18) {
            if (++st_q == WX_NCH) {
                st_q = 0;
                st_t += gridDim.x;
                st_coords();
            }
        }
    };
    st_coords();
    for (int pre = 0; pre < 2; ++pre) {              // chunks 0 and 1 of the first tile into LDS
#pragma unroll
        for (int i = 0; i < 19; ++i) request_op(i);
#pragma unroll
        for (int i = 0; i < 27; ++i) commit_op(pre, i);
    }
    __syncthreads();

    const int bidx = l31 & 15, rsub = l31 >> 4;
    for (int tI = blockIdx.x; tI < total; tI += gridDim.x) {
        const int tt = (int)mrx_xcd_band(tI, total);
        const int b = tt / a.ntiles, tile = tt - b * a.ntiles, ty0 = tile / a.tiles_x;
        const int h0 = ty0 * WX_TH, w0 = (tile - ty0 * a.tiles_x) * WX_TW;
#define WX_STAMP(i) if (a.trace && lane == 0 && tI == (int)blockIdx.x + (int)gridDim.x) a.trace[((long long)blockIdx.x * 4 + wave) * 4 + (i)] = __builtin_readcyclecounter();
        WX_STAMP(0)
        f32x16 acc[4][2][2];                         // [position][cout block][column block]
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int nb2 = 0; nb2 < 2; ++nb2)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[p][ct][nb2][r] = p == 1 ? tabl[64 + lhi * 32 + ct * 16 + r] : 0.f;
        u32x4 bt[2][2][2], at[2][2][2];              // [register set][column block | cout block][term]
        // operands of step `st` (0..3: the (ky 0 | ky 1) step of position st; 4..7: the paired ky = 2 step of position st - 4) of the chunk in buffer qb
        auto fetch = [&](int qb, int st, int set) {
            if (st < 4) {
                const int p = st;
                const u32x4* xw = Xp + qb * WX_XBUF + (p * 2) * WX_PSTR + (4 * wave + rsub + 2 * lhi) * 16 + bidx;
                const u32x4* wl = Wc + qb * WX_WCH + (p * 2) * 2 * 64 + lane;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    bt[set][0][k] = xw[k * WX_PSTR];
                    bt[set][1][k] = xw[k * WX_PSTR + 2 * 16];
                    at[set][0][k] = wl[(k * 2 + 0) * 64];
                    at[set][1][k] = wl[(k * 2 + 1) * 64];
                }
            } else {
                const int p = st - 4, bf = qb ^ lhi;
                const u32x4* xw = Xp + bf * WX_XBUF + (p * 2) * WX_PSTR + (4 * wave + rsub + 4) * 16 + bidx;
                const u32x4* w8 = Wc + bf * WX_WCH + WX_WFULL + (p * 2) * 2 * 32 + l31;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    bt[set][0][k] = xw[k * WX_PSTR];
                    bt[set][1][k] = xw[k * WX_PSTR + 2 * 16];
                    at[set][0][k] = w8[(k * 2 + 0) * 32];
                    at[set][1][k] = w8[(k * 2 + 1) * 32];
                }
            }
        };
        fetch(0, 0, 0);
        auto chunk = [&](const int q, auto FIRST, auto EVEN, auto LAST) {
            constexpr bool first = decltype(FIRST)::value, even = decltype(EVEN)::value, last = decltype(LAST)::value;
            const int cb = q & 1, nbuf = cb ^ 1;
            constexpr int NST = even ? 8 : 4;
#pragma unroll
            for (int st = 0; st < NST; ++st) {
                const int set = st & 1, p = st & 3;
                // the barriers: start of the fourth step (the commits of chunk q + 1 -- operations 0..26, behind the first 27 MFMAs -- are visible; in an odd chunk
                // nobody reads this chunk's buffer any more) and of the eighth (even chunks: the paired ky = 2 reads of this chunk's buffer have returned)
                if (st == 3 || st == 7) __syncthreads();
                if (st + 1 < NST) fetch(cb, st + 1, set ^ 1);
                else if (!last) fetch(nbuf, 0, set ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                constexpr int TA_[3] = {1, 0, 0}, TB_[3] = {0, 1, 0};
#pragma unroll
                for (int m = 0; m < 12; ++m) {
                    const int ct = m & 1, nb2 = (m >> 1) & 1, pr = m >> 2;
                    acc[p][ct][nb2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, at[set][ct][TA_[pr]]), __builtin_bit_cast(f16x8, bt[set][nb2][TB_[pr]]),
                                                                             acc[p][ct][nb2], 0, 0, 0);
                    const int g = 12 * st + m;       // MFMA index inside the chunk
#ifdef MRX_WX_ABL_NOSTAGE
                    if (false) {
#else
                    if (g < 27) {
#endif
                        if constexpr (!first) commit_op(nbuf, g);
                    } else if (g >= 28 && g < 47) {
#ifndef MRX_WX_ABL_NOSTAGE
                        request_op(g - 28);
#endif
                    }
                    if (g < 47) __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        chunk(0, std::true_type{}, std::true_type{}, std::false_type{});
        for (int q2 = 1; q2 < WX_NCH - 1; q2 += 2) {
            chunk(q2, std::false_type{}, std::false_type{}, std::false_type{});
            chunk(q2 + 1, std::false_type{}, std::true_type{}, std::false_type{});
        }
        chunk(WX_NCH - 1, std::false_type{}, std::false_type{}, std::true_type{});
        WX_STAMP(1)
        __syncthreads();                             // (every wave is past its last operand read of buffer 1: the next tile's chunk 1 goes there below)

        // ---- output transform: four pixel sets per lane, set (nb2, j) = (row h0 + 4 wave + 2 nb2 + rsub, column w0 + 4 (l31 & 7) + ((l31 >> 3) & 1) + 2 j) ------
        f32x16 out[4][2];                            // [set = 2 nb2 + j][cout block]
#pragma unroll
        for (int nb2 = 0; nb2 < 2; ++nb2)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float m0 = acc[0][ct][nb2][r], m1 = acc[1][ct][nb2][r], m2 = acc[2][ct][nb2][r], m3 = acc[3][ct][nb2][r];
                    out[2 * nb2][ct][r] = (m0 + m1) + m2;
                    out[2 * nb2 + 1][ct][r] = (m1 - m2) - m3;
                }
#pragma unroll
        for (int i = 0; i < 27; ++i) commit_op(1, i);     // the next tile's chunk 1 (requested behind the last MFMAs above) into buffer 1
        const int ox0 = w0 + 4 * (l31 & 7) + ((l31 >> 3) & 1);
        float hp[4][32];
        auto load_hp = [&](int sI) {
            if (!a.hprev) {
#pragma unroll
                for (int R = 0; R < 32; ++R) hp[sI][R] = 0.f;
                return;
            }
            const int oy = h0 + 4 * wave + 2 * (sI >> 1) + rsub, ox = ox0 + 2 * (sI & 1);
            const int cy = oy < a.H ? oy : a.H - 1, cx = ox < a.W ? ox : a.W - 1;
            const float* hb = a.hprev + (long long)b * WX_F * plane + ((long long)cy * a.W + cx) * 8 + 4 * lhi;
#pragma unroll
            for (int qq = 0; qq < 8; ++qq) {
                const float4 u = *reinterpret_cast<const float4*>(hb + (long long)qq * plane * 8);
                hp[sI][4 * qq] = u.x, hp[sI][4 * qq + 1] = u.y, hp[sI][4 * qq + 2] = u.z, hp[sI][4 * qq + 3] = u.w;
            }
        };
        load_hp(0), load_hp(1);
        float sg[4], ung[4];
#pragma unroll
        for (int sI = 0; sI < 4; ++sI) {
            float gm = 0.f;
#pragma unroll
            for (int R = 0; R < 32; ++R) gm = fmaxf(gm, out[sI][R >> 4][R & 15]);
            const int kg = wx_pixel_exp(gm);
            sg[sI] = wx_pow2(kg), ung[sI] = wx_pow2(-kg) * (unx * unw * unwi);
        }
        f32x16 acc2[4][2];
#pragma unroll
        for (int sI = 0; sI < 4; ++sI)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[sI][ct][r] = 0.f;
        {
            const u32x4* wl = Wih + lane;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                f16x8 b1[4], b2[4];
#pragma unroll
                for (int sI = 0; sI < 4; ++sI) {
                    unsigned g1[4], g2[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int R0 = 8 * s + 2 * k, R1 = R0 + 1;
                        float v0 = out[sI][R0 >> 4][R0 & 15], v1 = out[sI][R1 >> 4][R1 & 15];
                        v0 = v0 > 0.f ? v0 : 0.f;
                        v1 = v1 > 0.f ? v1 : 0.f;
                        wx_split2h_scaled(v0, v1, sg[sI], g1[k], g2[k]);
                    }
                    b1[sI] = __builtin_bit_cast(f16x8, (u32x4{g1[0], g1[1], g1[2], g1[3]}));
                    b2[sI] = __builtin_bit_cast(f16x8, (u32x4{g2[0], g2[1], g2[2], g2[3]}));
                }
                f16x8 a_[2][2];
#pragma unroll
                for (int k = 0; k < 2; ++k)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) a_[ct][k] = __builtin_bit_cast(f16x8, wl[((s * 2 + k) * 2 + ct) * 64]);
#pragma unroll
                for (int sI = 0; sI < 4; ++sI)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) acc2[sI][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_[ct][1], b1[sI], acc2[sI][ct], 0, 0, 0);
#pragma unroll
                for (int sI = 0; sI < 4; ++sI)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) acc2[sI][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_[ct][0], b2[sI], acc2[sI][ct], 0, 0, 0);
#pragma unroll
                for (int sI = 0; sI < 4; ++sI)
#pragma unroll
                    for (int ct = 0; ct < 2; ++ct) acc2[sI][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_[ct][0], b1[sI], acc2[sI][ct], 0, 0, 0);
                if (s == 1) load_hp(2), load_hp(3);
            }
        }
        WX_STAMP(2)
        const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc(a.hnew + (long long)b * WX_F * plane, 0, (unsigned)(plane * (WX_F * 4)), 0x00020000);
#pragma unroll
        for (int sI = 0; sI < 4; ++sI) {
            const int oy = h0 + 4 * wave + 2 * (sI >> 1) + rsub, ox = ox0 + 2 * (sI & 1);
            const unsigned offh = (oy < a.H && ox < a.W) ? (unsigned)((((long long)oy * a.W + ox) * 8 + 4 * lhi) * 4) : 0x80000000u;
#pragma unroll
            for (int R = 0; R < 32; ++R) {
                float v = acc2[sI][R >> 4][R & 15] * ung[sI] + tabl[128 + lhi * 32 + R];
                v += tabl[lhi * 32 + R] * hp[sI][R];
                hp[sI][R] = v > 0.f ? v : 0.f;
            }
#pragma unroll
            for (int qq = 0; qq < 8; ++qq)
                __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(hp[sI][4 * qq]), __float_as_uint(hp[sI][4 * qq + 1]), __float_as_uint(hp[sI][4 * qq + 2]),
                                                             __float_as_uint(hp[sI][4 * qq + 3])},
                                                       rh, offh + (unsigned)qq * (unsigned)(plane * 32), 0, 0);
        }
        if (a.P) {
            f32x16 accp[4];
            float sh[4], unh[4];
#pragma unroll
            for (int sI = 0; sI < 4; ++sI) {
#pragma unroll
                for (int r = 0; r < 16; ++r) accp[sI][r] = 0.f;
                float hm = 0.f;
#pragma unroll
                for (int R = 0; R < 32; ++R) hm = fmaxf(hm, hp[sI][R]);
                const int kh = wx_pixel_exp(hm);
                sh[sI] = wx_pow2(kh), unh[sI] = wx_pow2(-kh) * unwp;
            }
            const u32x4* wp = Wih + WX_LWIH + lane;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                f16x8 b1[4], b2[4];
#pragma unroll
                for (int sI = 0; sI < 4; ++sI) {
                    unsigned g1[4], g2[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) wx_split2h_scaled(hp[sI][8 * s + 2 * k], hp[sI][8 * s + 2 * k + 1], sh[sI], g1[k], g2[k]);
                    b1[sI] = __builtin_bit_cast(f16x8, (u32x4{g1[0], g1[1], g1[2], g1[3]}));
                    b2[sI] = __builtin_bit_cast(f16x8, (u32x4{g2[0], g2[1], g2[2], g2[3]}));
                }
                const f16x8 a1 = __builtin_bit_cast(f16x8, wp[(s * 2 + 0) * 64]);
                const f16x8 a2 = __builtin_bit_cast(f16x8, wp[(s * 2 + 1) * 64]);
#pragma unroll
                for (int sI = 0; sI < 4; ++sI) accp[sI] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b1[sI], accp[sI], 0, 0, 0);
#pragma unroll
                for (int sI = 0; sI < 4; ++sI) accp[sI] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b2[sI], accp[sI], 0, 0, 0);
#pragma unroll
                for (int sI = 0; sI < 4; ++sI) accp[sI] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1[sI], accp[sI], 0, 0, 0);
            }
            const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(a.P + (long long)b * 18 * plane, 0, (unsigned)(plane * (18 * 4)), 0x00020000);
#pragma unroll
            for (int sI = 0; sI < 4; ++sI) {
                const int oy = h0 + 4 * wave + 2 * (sI >> 1) + rsub, ox = ox0 + 2 * (sI & 1);
                const bool inside = oy < a.H && ox < a.W;
                const unsigned offp = inside ? (unsigned)((((long long)oy * a.W + ox) + 4ll * lhi * plane) * 4) : 0x80000000u;
                const unsigned offp16 = (inside && !lhi) ? offp : 0x80000000u;
#pragma unroll
                for (int r = 0; r < 8; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(accp[sI][r] * unh[sI]), rp, offp + (unsigned)((r & 3) + 8 * (r >> 2)) * (unsigned)(plane * 4), 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(accp[sI][8] * unh[sI]), rp, offp16 + 16u * (unsigned)(plane * 4), 0, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(accp[sI][9] * unh[sI]), rp, offp16 + 17u * (unsigned)(plane * 4), 0, 0);
            }
        }
        WX_STAMP(3)
        __syncthreads();                             // (the commit of the next tile's chunk 1 above is visible before its first paired step reads it; also orders the tails)
    }
}


extern "C" int64_t mrx_rim_layer2_wx_pack_floats(void) { return (int64_t)WX_PACK_U4 * 4; }
// w_conv [64,64,3,3] (dilation 2), w_ih [64,64,1,1], w_final [2,64,3,3] or null -> the operand pack of mrx_rim_layer2_wx_cb8
extern "C" int mrx_rim_layer2_wx_pack(const float* w_conv, const float* w_ih, const float* w_final, float* packed, void* stream) {
    MRX_REQUIRE(w_conv && packed, MRX_EINVAL, "mrx_rim_layer2_wx_pack: null pointer");
    hipLaunchKernelGGL(k_l2wx_wscale, dim3(1), dim3(256), 0, (hipStream_t)stream, w_conv, w_ih, w_final, reinterpret_cast<u32x4*>(packed));
    hipLaunchKernelGGL(k_l2wx_pack, dim3((WX_PACK_U4 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_conv, w_ih, w_final, reinterpret_cast<u32x4*>(packed));
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
// mrx_rim_layer2_f16_cb8 (same arguments, same results to the round-off stated above) with the convolution in the Winograd F(2, 3) form along x.
// x, h_prev, h_new [B][8][H][W][8]; taps [B][18][H][W] or NULL; xmax: a device scalar >= max |x|.  A sample's state must fit 32-bit byte offsets.
extern "C" int mrx_rim_layer2_wx_cb8(const float* x, const float* packed, const float* b_conv, const float* b_ih, const float* hh, const float* h_prev,
                                     float* h_new, float* taps, const float* xmax, int B, int H, int W, void* stream) {
    MRX_REQUIRE(x && packed && hh && h_new && xmax, MRX_EINVAL, "mrx_rim_layer2_wx_cb8: null pointer");
    MRX_REQUIRE(B >= 0 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_rim_layer2_wx_cb8: bad dims");
    if (B == 0) return MRX_OK;
    MRX_REQUIRE((long long)H * W * 256 < (1ll << 31), MRX_EUNSUP, "mrx_rim_layer2_wx_cb8: %d x %d: a sample's state exceeds 32-bit byte offsets", H, W);
    MRX_CHECK_BOUND("mrx_rim_layer2_wx_cb8", x, (long long)B * 64 * H * W, xmax, stream);
    L2wxArgs a;
    a.x = x, a.packed = reinterpret_cast<const u32x4*>(packed), a.b_conv = b_conv, a.b_ih = b_ih, a.hh = hh, a.hprev = h_prev, a.hnew = h_new, a.P = taps;
    a.xmax = reinterpret_cast<const unsigned*>(xmax), a.B = B, a.H = H, a.W = W;
    a.tiles_x = mrx_cdiv(W, WX_TW), a.ntiles = a.tiles_x * mrx_cdiv(H, WX_TH);
    static bool attr_done = false;
    if (!attr_done) {
        MRX_HIP(hipFuncSetAttribute((const void*)k_rim_layer2_wx8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WX_LDS));
        MRX_HIP(hipFuncSetAttribute((const void*)k_rim_layer2_wx4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WX_LDS));
        attr_done = true;
    }
    // (round 5, same box, 8 slices per launch: eight waves 75.9 us per slice, four waves 85.3 -- the direct kernel 62.0; MRX_WX4=1 selects the four-wave form)
    static const bool eight = !(getenv("MRX_WX4") && atoi(getenv("MRX_WX4")) != 0);
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t prop;
        ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    const long long total = (long long)a.ntiles * B;
    a.trace = nullptr;
    static unsigned long long* d_trace = nullptr;
    if (getenv("MRX_WX_TRACE")) {
        if (!d_trace) (void)hipMalloc((void**)&d_trace, sizeof(unsigned long long) * 256 * 16);
        (void)hipMemsetAsync(d_trace, 0, sizeof(unsigned long long) * 256 * 16, (hipStream_t)stream);
        a.trace = d_trace;
    }
    if (eight) hipLaunchKernelGGL(k_rim_layer2_wx8, dim3((unsigned)(total < ncu ? total : ncu)), dim3(WX_NT), WX_LDS, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(k_rim_layer2_wx4, dim3((unsigned)(total < ncu ? total : ncu)), dim3(WX4_NT), WX_LDS, (hipStream_t)stream, a);
    MRX_LAUNCH_CHECK();
    if (a.trace && !eight) {
        (void)hipStreamSynchronize((hipStream_t)stream);
        static unsigned long long h[256 * 16];
        (void)hipMemcpy(h, d_trace, sizeof(h), hipMemcpyDeviceToHost);
        double ph[3] = {0, 0, 0};
        long n = 0;
        for (int w = 0; w < 256 * 4; ++w) {
            const unsigned long long* t = h + w * 4;
            if (t[0] && t[3] > t[0]) ph[0] += (double)(t[1] - t[0]), ph[1] += (double)(t[2] - t[1]), ph[2] += (double)(t[3] - t[2]), ++n;
        }
        if (n) fprintf(stderr, "[wx4-trace] %ld waves: chunk loop %.0f, tail head + 1x1 stage %.0f, epilogue + stores + tap stage %.0f cycles\n", n, ph[0] / n, ph[1] / n, ph[2] / n);
    }
    return MRX_OK;
}
