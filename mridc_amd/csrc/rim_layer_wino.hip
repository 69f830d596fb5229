// rim_layer_wino.hip -- Winograd F(2x2, 3x3) version of the fused RIM layer for the dilated 3x3 convolution
// (conv_layers.py:121-123 with k = 3, dilation = 2, 64 -> 64 channels) + IndRNNCell(1x1) (rnn_cells.py:384-391).
//
// A dilation-2 3x3 convolution is four independent plain 3x3 convolutions on the four (row parity, column parity)
// sub-lattices of the image.  On each sub-lattice F(2x2,3x3) produces a 2x2 output tile from a 4x4 input patch with 16
// multiplies instead of 36: per input-channel chunk the kernel forms V = B^T d B for every (tile, channel) on the vector ALUs,
// runs 16 independent [64 couts x channels] x [channels x tiles] products on the fp32 matrix cores
// (v_mfma_f32_16x16x4_f32, exact fp32 fma chains) -- 2.25x fewer MFMA cycles than the direct form -- and finishes with
// Y = A^T M A per lane in registers (all 16 transform positions of one (cout, tile) element live in the same lane).
// Replicate padding is resolved when the raw halo'd tile is gathered (clamped coordinates), exactly as in the direct kernel;
// the sub-lattices only index that tile.  The 1x1 `ih` GEMM and the wide epilogue are the direct kernel's, fed from LDS.
//
// Workgroup: 512 threads, 8x32 output pixels = 64 Winograd tiles (16 per parity).  Wave w: couts 32*(w&1)..+31, parity w>>1.
// Pipeline (one barrier per 8-channel chunk, everything double-buffered in 156.5 KB of LDS, one workgroup per CU):
//   iteration q:  LDS-DMA  packed U(q+1) and raw X(q+2) global -> LDS (global_load_lds, no staging registers),
//                 transform X(q+1) -> V(q+1),   matrix cores on U(q), V(q).
// The two waves that share a SIMD (w, w+4) run {transform, MFMA} in opposite orders, so the matrix pipe of every SIMD has
// MFMAs to issue while the other wave is on the vector ALUs / LDS.
// LDS images: U and V are [xi][k pair j][column ^ 16*(j&1)][2 k] so one ds_read_b64 gives a lane its operand for two
// MFMA k-steps, conflict-free (64 banks per 32 lanes); the raw tile rows have stride 38 so the 4x4 patch gathers are too.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <type_traits>
#include <vector>

#include "mrx_common.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define WN_NT 512
#define WN_F 64
#define WN_CK 8
#define WN_PH 12
// Raw tile geometry: one 512-float plane per channel (whole DMA instructions, one channel each).  X4 (W % 4 == 0, 16-byte
// aligned x): rows of 10 aligned float4 starting 4 columns left of the tile, two dwordx4 DMA instructions per channel; the
// column part of the replicate border is applied when the patch is gathered.  Otherwise: rows of 36 (+2 pad, conflict-free
// gathers) fetched element-wise with eight dword DMA instructions per channel, clamped at the source.
#define WN_XS(X4) ((X4) ? 40 : 38)
#define WN_OX(X4) ((X4) ? 4 : 2)
#define WN_XPLANE 512
#define WN_XBUF (WN_CK * WN_XPLANE)    // 4096 floats
#define WN_UCHUNK (16 * 4 * WN_F * 2)  // 8192 floats = 32 DMA wave-instructions of 1 KB
#define WN_VBUF WN_UCHUNK
#define WN_LDS_FLOATS (2 * WN_XBUF + 2 * WN_UCHUNK + 2 * WN_VBUF)  // 40960 = the CU's whole 160 KB
#define WN_PF 3

struct WinoArgs {
    const float* x;       // [B,Cin,H,W]
    const float* packed;  // per chunk the LDS image of G g G^T, then the ih block [32][2][F] (rim_layer.hip order)
    const float* b_conv;
    const float* b_ih;
    const float* hh;
    const float* hprev;
    float* hnew;
    int B, Cin, H, W, tiles_x, ntiles;
    unsigned long long* trace;  // debug only (env MRX_TRACE): cycle stamps per workgroup
    int act;      // plain-convolution form (TAIL = false) only: MRX_ACT_* applied to conv + bias
    float slope;  // leaky slope of that activation
    long long out_bstride;  // elements between the batch entries of hnew / hprev (64 * H * W unless a wider tensor is written in 64-channel blocks)
};

// U = G g G^T per (cout, cin); per chunk q the image [xi = 4i+j'][k pair j][p = co ^ 16*(j&1)][kb], cin = 8q + 2j + kb
// (G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]])
__global__ void k_wino_pack(const float* __restrict__ w, const float* __restrict__ w_ih, float* __restrict__ out, int Cin, int nchunks) {
    const int conv_elems = nchunks * WN_UCHUNK;
    const int total = conv_elems + WN_F * WN_F;
    const float G[4][3] = {{1.f, 0.f, 0.f}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0.f, 0.f, 1.f}};
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        float v = 0.f;
        if (idx < conv_elems) {
            const int kb = idx & 1;
            int r = idx >> 1;
            const int p = r % WN_F;
            r /= WN_F;
            const int j = r & 3;
            r >>= 2;
            const int xi = r % 16;
            const int q = r / 16;
            const int co = p ^ (16 * (j & 1));
            const int ci = q * WN_CK + 2 * j + kb;
            if (ci < Cin) {
                const float* g = w + ((long long)co * Cin + ci) * 9;
                const int i = xi >> 2, jj = xi & 3;
                float acc = 0.f;
                for (int k = 0; k < 3; ++k)
                    for (int l = 0; l < 3; ++l) acc += G[i][k] * g[k * 3 + l] * G[jj][l];
                v = acc;
            }
        } else {
            const int jdx = idx - conv_elems;
            const int o = jdx % WN_F;
            int r = jdx / WN_F;
            const int half = r & 1;
            r >>= 1;
            const int ct = r >> 4, reg = r & 15;
            const int c = 32 * ct + (reg & 3) + 8 * (reg >> 2) + 4 * half;
            v = w_ih ? w_ih[o * WN_F + c] : 0.f;
        }
        out[idx] = v;
    }
}

extern "C" int64_t mrx_rim_layer_wino_pack_floats(int Cin, int F) {
    if (F != WN_F || Cin < 1) return -1;
    return (int64_t)((Cin + WN_CK - 1) / WN_CK) * WN_UCHUNK + WN_F * WN_F;
}
extern "C" int mrx_rim_layer_wino_pack(const float* w_conv, const float* w_ih, float* packed, int Cin, int F, void* stream) {
    MRX_REQUIRE(w_conv && packed, MRX_EINVAL, "mrx_rim_layer_wino_pack: null pointer");  // w_ih may be null (plain convolution)
    MRX_REQUIRE(F == WN_F && Cin >= 1, MRX_EUNSUP, "mrx_rim_layer_wino_pack: F=%d Cin=%d", F, Cin);
    const int nchunks = (Cin + WN_CK - 1) / WN_CK;
    const int total = nchunks * WN_UCHUNK + WN_F * WN_F;
    hipLaunchKernelGGL(k_wino_pack, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_conv, w_ih, packed, Cin, nchunks);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

#define WN_GLOBAL(p) ((const __attribute__((address_space(1))) void*)(p))
#define WN_SHARED(p) ((__attribute__((address_space(3))) void*)(p))

// DIL = 2: the four parity sub-lattices (above).  DIL = 1: the same 64 tiles are the 4 x 16 plain 2x2 tiles of the 8 x 32 pixel
// block (tile row 2 tby + py, tile column 2 tbx + px): only the patch gather and the output scatter differ.
// TAIL = false: plain convolution + bias + activation, no 1x1 stage (rows go from Y straight to HBM).
// ZP: zero padding instead of replicate: patch elements outside the image are zeroed when gathered (border tiles only).
template <int ABL, bool X4, int DIL = 2, bool TAIL = true, bool ZP = false>
__global__ __launch_bounds__(WN_NT, 2) void k_rim_layer_wino(WinoArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem_f[];
    float* Xs = smem_f;                          // [2][CK][512] raw halo'd tiles (12 rows of XS)
    constexpr int XS = WN_XS(X4), OX = WN_OX(X4);
    float* Us = Xs + 2 * WN_XBUF;                // [2][16][4][64][2]
    float* Vs = Us + 2 * WN_UCHUNK;              // [2][16][4][64][2]
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cot = wave & 1, par = wave >> 1;
    const int py = par >> 1, px = par & 1;
    // Everything that depends on the lane only is re-derived per tile from a laundered copy of the lane id (refresh_lane): left to
    // the compiler it is all hoisted out of the persistent tile loop, and the ~90 extra live registers spill.
    int lane = tid & 63;
    int l15, lg;  // 16x16x4 operand lane split: column / k index
    unsigned uoff;
    int t_pos, t_dst, m_a0, m_a1, m_b;
    auto refresh_lane = [&]() {
        asm volatile("" : "+v"(lane));
        l15 = lane & 15;
        lg = lane >> 4;
        uoff = (unsigned)lane * 16u;
        // input transform: lane -> tile (tbx = lane&7, px = lane>>3 &1, py = lane>>4 &1, tby = lane>>5), wave -> channel
        t_pos = (lane & 7) + 8 * (((lane >> 3) & 1) ^ (lane >> 5)) + 16 * ((lane >> 4) & 1) + 32 * (lane >> 5);
        t_dst = ((wave >> 1) * WN_F + (t_pos ^ (16 * ((wave >> 1) & 1)))) * 2 + (wave & 1);
        const int m_sw = 16 * (lg & 1);
        m_a0 = (lg * WN_F + ((32 * cot + l15) ^ m_sw)) * 2;
        m_a1 = (lg * WN_F + ((32 * cot + 16 + l15) ^ m_sw)) * 2;
        m_b = (lg * WN_F + (((l15 & 7) + 8 * (px ^ (l15 >> 3)) + 16 * py + 32 * (l15 >> 3)) ^ m_sw)) * 2;
    };
    refresh_lane();

    // Persistent: gridDim.x workgroups (one per CU: the kernel owns all 160 KB of LDS) walk the tiles; virtual index v -> tile keeps
    // every XCD (v & 7 = blockIdx.x & 7 when gridDim.x % 8 == 0) on its own contiguous band of tiles (halo reuse in its L2).
    const long long plane = (long long)a.H * a.W;
    const int nchunks = (a.Cin + WN_CK - 1) / WN_CK;
    const int nt_total = a.ntiles * a.B;
    int h0 = 0, w0 = 0, b = 0;
    const float* xb = a.x;
    bool first = true;

#define WN_STAMP(i) \
    if (a.trace && tid == 0 && first) a.trace[(long long)blockIdx.x * 8 + (i)] = __builtin_readcyclecounter();
    WN_STAMP(0)

    // ---- LDS-DMA work list of this wave: 4 copies of 1 KB of the packed U image (n = wave + 8 m) and the raw tile of channel
    // `wave` of the chunk (replicate border = clamped coordinates, conv_layers.py:72-76).  Sources are a wave-uniform base
    // (scalar registers, advanced on the scalar unit) plus a fixed per-lane byte offset: no vector instructions per copy.
    constexpr int NXM = X4 ? 2 : 8;  // raw-tile DMA instructions per wave and chunk
    unsigned xoff[NXM];
    int t_col[4];  // LDS offsets of the four patch columns in the first patch row (X4: column replicate border applied here)
    unsigned zp_ok = 0xffffu;  // ZP: bit 4 i + j set = patch element (i, j) of this lane's tile lies inside the image
    bool zp_border = false;    // ZP: this tile touches the image border (wave-uniform)
    auto set_tile = [&](int v) {  // tile coordinates and the per-lane DMA / gather offsets that depend on them
        // workgroup g runs on XCD g % 8 and v % 8 = g % 8 when the grid is a multiple of 8: every XCD walks one contiguous band of
        // the (batch x tile) list, for any tile count
        const int vb = (gridDim.x & 7) == 0 ? (int)mrx_xcd_band(v, nt_total) : v;
        b = vb / a.ntiles;
        const int t = vb - b * a.ntiles;
        const int ty0 = t / a.tiles_x;
        h0 = ty0 * 8;
        w0 = (t - ty0 * a.tiles_x) * 32;
        xb = a.x + (long long)b * a.Cin * plane;
#pragma unroll
        for (int m = 0; m < NXM; ++m) {
            int sl = m * 64 + lane, gy, gx;  // float4 (X4) or element slot of the plane; slots past the tile repeat its last one
            if (X4) {
                sl = sl < 12 * (XS / 4) ? sl : 12 * (XS / 4) - 1;
                const int ry = sl / (XS / 4), c4 = sl - ry * (XS / 4);
                gy = h0 + ry - 2;
                gx = w0 - OX + 4 * c4;  // whole groups outside the image fetch the nearest inside one; the gather never reads them
                gx = gx < 0 ? 0 : (gx > a.W - 4 ? a.W - 4 : gx);
            } else {
                sl = sl < 12 * XS ? sl : 12 * XS - 1;
                const int ry = sl / XS, rx = sl - ry * XS;
                gy = h0 + ry - 2;
                gx = w0 + rx - OX;
                gx = gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx);
            }
            gy = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy);
            xoff[m] = (unsigned)(gy * a.W + gx) * 4u;
        }
        // first patch row / column of this lane's tile in the raw tile (origin: image (h0 - 2, w0 - OX))
        const int prow0 = DIL == 2 ? 4 * (lane >> 5) + ((lane >> 4) & 1) : 4 * (lane >> 5) + 2 * ((lane >> 4) & 1) + 1;
        const int pcol0 = (DIL == 2 ? 4 * (lane & 7) + ((lane >> 3) & 1) : 4 * (lane & 7) + 2 * ((lane >> 3) & 1) + 1) + OX - 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int c = pcol0 + DIL * j;  // tile column of image column w0 - OX + c
            if (X4) {
                const int lo = OX - w0, hi = a.W - 1 - w0 + OX;
                c = c < lo ? lo : (c > hi ? hi : c);
            }
            t_col[j] = wave * WN_XPLANE + prow0 * XS + c;
        }
        if constexpr (ZP) {
            zp_border = h0 < 2 || h0 + 10 > a.H || w0 < 2 || w0 + 34 > a.W;
            zp_ok = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int gy = h0 - 2 + prow0 + DIL * i, gx = w0 - OX + pcol0 + DIL * j;
                    zp_ok |= (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) ? (1u << (4 * i + j)) : 0u;
                }
        }
    };
    auto dma_u = [&](int q, int m) {  // m in 0..3
        const int n = wave + 8 * m;
        const char* src = reinterpret_cast<const char*>(a.packed + (long long)q * WN_UCHUNK + n * 256);
        __builtin_amdgcn_global_load_lds(WN_GLOBAL(src + uoff), WN_SHARED(Us + (q & 1) * WN_UCHUNK + n * 256), 16, 0, 0);
    };
    auto dma_x = [&](int q, int m) {  // channels past Cin re-read the last one (their packed weights are zero)
        int gc = q * WN_CK + wave;
        gc = gc < a.Cin ? gc : a.Cin - 1;
        const char* src = reinterpret_cast<const char*>(xb + (long long)gc * plane);
        float* dst = Xs + (q & 1) * WN_XBUF + wave * WN_XPLANE;
        if (X4) __builtin_amdgcn_global_load_lds(WN_GLOBAL(src + xoff[m]), WN_SHARED(dst + m * 256), 16, 0, 0);
        else __builtin_amdgcn_global_load_lds(WN_GLOBAL(src + xoff[m]), WN_SHARED(dst + m * 64), 4, 0, 0);
    };

    // ---- input transform: lane -> tile (tbx = lane&7, px = lane>>3 &1, py = lane>>4 &1, tby = lane>>5), wave -> channel.
    // V = B^T d B with d[i][j] = raw[4 tby + py + 2i][4 tbx + px + 2j]; written to column pos(tile) of row (xi, j = ci>>1).
    auto transform = [&](int q) {  // stand-alone form (prologue only; later chunks ride inside the MFMA loop)
        const float* src = Xs + (q & 1) * WN_XBUF;
        float d[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                d[i][j] = src[t_col[j] + (DIL * i) * XS];
                if (ZP && zp_border) d[i][j] = (zp_ok >> (4 * i + j)) & 1u ? d[i][j] : 0.f;
            }
        float e[4][4];  // B^T d : rows (d0-d2, d1+d2, d2-d1, d1-d3)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            e[0][j] = d[0][j] - d[2][j];
            e[1][j] = d[1][j] + d[2][j];
            e[2][j] = d[2][j] - d[1][j];
            e[3][j] = d[1][j] - d[3][j];
        }
        float* dst = Vs + (q & 1) * WN_VBUF + t_dst;
#pragma unroll
        for (int i = 0; i < 4; ++i) {  // (B^T d) B : columns the same way
            dst[(4 * i + 0) * 512] = e[i][0] - e[i][2];
            dst[(4 * i + 1) * 512] = e[i][1] + e[i][2];
            dst[(4 * i + 2) * 512] = e[i][2] - e[i][1];
            dst[(4 * i + 3) * 512] = e[i][1] - e[i][3];
        }
    };

    f32x4 acc[16][2];

    // ---- one chunk: 16 steps of {4 MFMAs, operand reads for 3 steps ahead}.  The fp32 matrix pipe is the SIMD's fp32 vector
    // pipe (157 TFLOP/s either way): every other VALU instruction costs it 3-4 cycles (measured: tools/probe/filler_probe*),
    // LDS and scalar instructions almost nothing.  So the DMA for the following chunks and the input transform of the next one
    // ride along inside the loop with almost no address arithmetic (immediate offsets, scalar bases).
    auto chunk = [&](int q) {
        const int PAR = q & 1;
        const float* ua0 = Us + PAR * WN_UCHUNK + m_a0;
        const float* ua1 = Us + PAR * WN_UCHUNK + m_a1;
        const float* vb = Vs + PAR * WN_VBUF + m_b;
        const bool more_u = q + 1 < nchunks, more_x = q + 2 < nchunks;
        const bool tr = q + 1 < nchunks && !(ABL & 1);
        const float* tsrc = Xs + (PAR ^ 1) * WN_XBUF;
        float* tdst = Vs + (PAR ^ 1) * WN_VBUF + t_dst;
        float d[4][4], e[4][4];
        constexpr int PF = WN_PF;
        f32x2 ra0[PF + 1], ra1[PF + 1], rb[PF + 1];
        if (ABL & 2)
            for (int i = 0; i <= PF; ++i) ra0[i] = ra1[i] = rb[i] = (f32x2){(float)tid, (float)i};
#pragma unroll
        for (int s = 0; s < 16 + PF; ++s) {
            const int xi = s - PF, c = (xi < 0 ? 0 : xi) % (PF + 1), w = s % (PF + 1);
            // ---- sub-step 0
            if (s >= PF && !(ABL & 8)) acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra0[c][0], rb[c][0], acc[xi][0], 0, 0, 0);
            if (tr && s < 2) {  // patch column 2s
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    d[i][2 * s] = tsrc[t_col[2 * s] + (DIL * i) * XS];
                    if (ZP && zp_border) d[i][2 * s] = (zp_ok >> (4 * i + 2 * s)) & 1u ? d[i][2 * s] : 0.f;
                }
            }
            if (tr && s >= 3 && s < 7) {  // B^T d, one column per step
                const int j = s - 3;
                e[0][j] = d[0][j] - d[2][j];
                e[1][j] = d[1][j] + d[2][j];
            }
            if (tr && s >= 7 && s < 15) {  // (B^T d) B, half a row per step
                const int i = (s - 7) >> 1;
                if (((s - 7) & 1) == 0) tdst[(4 * i + 0) * 512] = e[i][0] - e[i][2];
                else tdst[(4 * i + 2) * 512] = e[i][2] - e[i][1];
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- sub-step 1
            if (s >= PF && !(ABL & 8)) acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra1[c][0], rb[c][0], acc[xi][1], 0, 0, 0);
            if (!(ABL & 4)) {
                if (s < 4) {
                    if (more_u) dma_u(q + 1, s);
                } else if (s < 4 + NXM) {
                    if (more_x) dma_x(q + 2, s - 4);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- sub-step 2
            if (s >= PF && !(ABL & 8)) acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra0[c][1], rb[c][1], acc[xi][0], 0, 0, 0);
            if (tr && s < 2) {  // patch column 2s+1
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    d[i][2 * s + 1] = tsrc[t_col[2 * s + 1] + (DIL * i) * XS];
                    if (ZP && zp_border) d[i][2 * s + 1] = (zp_ok >> (4 * i + 2 * s + 1)) & 1u ? d[i][2 * s + 1] : 0.f;
                }
            }
            if (tr && s >= 3 && s < 7) {
                const int j = s - 3;
                e[2][j] = d[2][j] - d[1][j];
                e[3][j] = d[1][j] - d[3][j];
            }
            if (tr && s >= 7 && s < 15) {
                const int i = (s - 7) >> 1;
                if (((s - 7) & 1) == 0) tdst[(4 * i + 1) * 512] = e[i][1] + e[i][2];
                else tdst[(4 * i + 3) * 512] = e[i][1] - e[i][3];
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- sub-step 3: operands of step s (consumed PF steps later; slot w was last read by the MFMAs of step s - 1)
            if (s >= PF && !(ABL & 8)) acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ra1[c][1], rb[c][1], acc[xi][1], 0, 0, 0);
            if (s < 16 && !(ABL & 2)) {
                rb[w] = *reinterpret_cast<const f32x2*>(vb + s * 512);
                ra0[w] = *reinterpret_cast<const f32x2*>(ua0 + s * 512);
                ra1[w] = *reinterpret_cast<const f32x2*>(ua1 + s * 512);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto sync = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA has landed
        __syncthreads();                                   // ... everyone's, and V(q) is written; the buffers of q - 1 are free
    };

    auto prologue = [&]() {  // U(0), X(0), X(1) of the tile set by set_tile -> LDS
#pragma unroll
        for (int m = 0; m < 4; ++m) dma_u(0, m);
#pragma unroll
        for (int m = 0; m < NXM; ++m) dma_x(0, m);
        if (nchunks > 1) {
#pragma unroll
            for (int m = 0; m < NXM; ++m) dma_x(1, m);
        }
    };
    int v = blockIdx.x;
    if (v < nt_total) {
        set_tile(v);
        prologue();
    }
#pragma nounroll
    for (; v < nt_total; v += gridDim.x) {
    refresh_lane();
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[xi][h][r] = 0.f;
    sync();  // prologue data landed; the previous tile's epilogue is out of LDS
    transform(0);
#pragma nounroll
    for (int q = 0; q < nchunks; ++q) {  // one body for all chunks: the accumulators stay in place
        sync();
        chunk(q);
    }
    __syncthreads();  // all waves done with U / V
    WN_STAMP(1)
    // The tail of this tile lives in the upper LDS (ih weights in U[1], Y and the transpose tiles in V[0..1]); the lower 64 KB
    // (raw tiles, U[0]) take the next tile's prologue DMA now, so its latency hides behind the tail.
    const int ch0 = h0, cw0 = w0, cb = b;
    float* Ys = Vs;
    float* Wi = Us + WN_UCHUNK;
    if constexpr (TAIL) {
        const char* src = reinterpret_cast<const char*>(a.packed + (long long)nchunks * WN_UCHUNK + wave * 512);
        __builtin_amdgcn_global_load_lds(WN_GLOBAL(src + uoff), WN_SHARED(Wi + wave * 512), 16, 0, 0);
        __builtin_amdgcn_global_load_lds(WN_GLOBAL(src + 1024 + uoff), WN_SHARED(Wi + wave * 512 + 256), 16, 0, 0);
    }
    if (v + (int)gridDim.x < nt_total) {
        set_tile(v + gridDim.x);
        prologue();
    }

    // ---- output transform Y = A^T M A per lane, bias, ReLU -> Ys[cout][8 rows x 32 cols] ------------------------------
    {
        const int tby = l15 >> 3, tbx = l15 & 7;
        const int prow = DIL == 2 ? 4 * tby + py : 4 * tby + 2 * py, pcol = DIL == 2 ? 4 * tbx + px : 4 * tbx + 2 * px;
        const int act = TAIL ? MRX_ACT_RELU : a.act;
        const float neg = act == MRX_ACT_RELU ? 0.f : (act == MRX_ACT_LEAKY ? a.slope : 1.f);  // factor applied to negative values
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float s0[4], s1[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s0[j] = acc[0 + j][h][r] + acc[4 + j][h][r] + acc[8 + j][h][r];
                    s1[j] = acc[4 + j][h][r] - acc[8 + j][h][r] - acc[12 + j][h][r];
                }
                const int co = 32 * cot + 16 * h + 4 * lg + r;
                const float bc = a.b_conv ? a.b_conv[co] : 0.f;
                float y00 = s0[0] + s0[1] + s0[2] + bc, y01 = s0[1] - s0[2] - s0[3] + bc;
                float y10 = s1[0] + s1[1] + s1[2] + bc, y11 = s1[1] - s1[2] - s1[3] + bc;
                if constexpr (TAIL) {
                    y00 = y00 > 0.f ? y00 : 0.f;
                    y01 = y01 > 0.f ? y01 : 0.f;
                    y10 = y10 > 0.f ? y10 : 0.f;
                    y11 = y11 > 0.f ? y11 : 0.f;
                } else {
                    y00 = y00 > 0.f ? y00 : y00 * neg;
                    y01 = y01 > 0.f ? y01 : y01 * neg;
                    y10 = y10 > 0.f ? y10 : y10 * neg;
                    y11 = y11 > 0.f ? y11 : y11 * neg;
                }
                float* yo = Ys + co * 256 + prow * 32 + pcol;
                yo[0] = y00;
                yo[DIL] = y01;
                yo[32 * DIL] = y10;
                yo[33 * DIL] = y11;
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // ih weights landed (and, long before they are needed, the next prologue)
    __syncthreads();
    WN_STAMP(2)

    // ---- 1x1 ih GEMM (32x32x2 MFMA, wave = image row) + wide epilogue: same as k_rim_layer, B operand read from Ys ------
    const int l31 = lane & 31, lhi = lane >> 5;
    const int oy = ch0 + wave;
    const bool wide = (a.W & 3) == 0;
    const int wch = lane >> 3, wpx = (lane & 7) * 4;
    const long long wbase = (long long)cb * a.out_bstride + (long long)oy * a.W + cw0 + wpx;
    const bool winside = oy < a.H && (cw0 + wpx) < a.W;
    if constexpr (!TAIL) {
        // plain convolution: wave = image row `wave` of the tile, Ys already holds [cout][row][32 columns]
        const float* yr = Ys + wave * 32;
        if (wide) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int ch = i * 8 + wch;
                const float4 v = *reinterpret_cast<const float4*>(yr + ch * 256 + wpx);
                if (winside) *reinterpret_cast<float4*>(a.hnew + wbase + (long long)ch * plane) = v;
            }
        } else if (oy < a.H && cw0 + l31 < a.W) {
            const long long obase = (long long)cb * a.out_bstride + (long long)oy * a.W + cw0 + l31;
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const int co = 2 * i + lhi;
                a.hnew[obase + (long long)co * plane] = yr[co * 256 + l31];
            }
        }
        WN_STAMP(4)
        first = false;
        continue;  // the barrier at the top of the next tile separates these reads of Ys from its new contents
    }
    float4 hp4[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        hp4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (wide && a.hprev && winside) hp4[i] = *reinterpret_cast<const float4*>(a.hprev + wbase + (long long)(i * 8 + wch) * plane);
    }
    f32x16 acc2[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc2[ct][r] = 0.f;
    {
        const float* wi = Wi + lhi * WN_F + l31;
        const float* yb = Ys + wave * 32 + l31;
        float qa0[WN_PF + 1], qa1[WN_PF + 1], qb[WN_PF + 1];
#pragma unroll
        for (int s = 0; s < 32 + WN_PF; ++s) {
            if (s < 32) {
                const int ct = s >> 4, r = s & 15;
                const int c_lo = 32 * ct + (r & 3) + 8 * (r >> 2);  // k enumeration of the packed ih block (C/D order)
                qb[s % (WN_PF + 1)] = yb[(c_lo + 4 * lhi) * 256];
                qa0[s % (WN_PF + 1)] = wi[(s * 2) * WN_F];
                qa1[s % (WN_PF + 1)] = wi[(s * 2) * WN_F + 32];
            }
            if (s >= WN_PF) {
                const int c = (s - WN_PF) % (WN_PF + 1);
                acc2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(qa0[c], qb[c], acc2[0], 0, 0, 0);
                acc2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(qa1[c], qb[c], acc2[1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();  // Ys / Wi consumed by every wave: LDS becomes 8 wave-private [64][32] transpose tiles
    WN_STAMP(3)
    const int ox = cw0 + l31;
    if (wide) {
        float* T = Ys + wave * (WN_F * 32);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                T[co * 32 + l31] = acc2[ct][r];
            }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int ch = i * 8 + wch;
            float4 v = *reinterpret_cast<const float4*>(T + ch * 32 + wpx);
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
            if (winside) *reinterpret_cast<float4*>(a.hnew + wbase + (long long)ch * plane) = v;
        }
    } else if (oy < a.H && ox < a.W) {
        const long long obase = (long long)cb * a.out_bstride + (long long)oy * a.W + ox;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                float v = acc2[ct][r];
                if (a.b_ih) v += a.b_ih[co];
                if (a.hprev) v += a.hh[co] * a.hprev[obase + (long long)co * plane];
                a.hnew[obase + (long long)co * plane] = v > 0.f ? v : 0.f;
            }
    }
    WN_STAMP(4)
    if (a.trace && tid == 0 && first) {
        unsigned hwid, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        a.trace[(long long)blockIdx.x * 8 + 6] =
            ((unsigned long long)(xcc & 0xf) << 16) | (((hwid >> 13) & 7) << 8) | (((hwid >> 12) & 1) << 4) | ((hwid >> 8) & 15);
    }
    first = false;
    }  // tile loop
}

extern "C" int mrx_rim_layer_indrnn_wino(const float* x, const float* packed, const float* b_conv, const float* b_ih,
                                         const float* hh, const float* h_prev, float* h_new, int B, int Cin, int F, int H, int W,
                                         void* stream) {
    MRX_REQUIRE(x && packed && hh && h_new, MRX_EINVAL, "mrx_rim_layer_indrnn_wino: null pointer");
    MRX_REQUIRE(B >= 0 && Cin >= 1 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_rim_layer_indrnn_wino: bad dims");
    MRX_REQUIRE(F == WN_F, MRX_EUNSUP, "mrx_rim_layer_indrnn_wino: hidden size %d (only %d)", F, WN_F);
    if (B == 0) return MRX_OK;
    WinoArgs a;
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
    a.act = MRX_ACT_RELU;
    a.slope = 0.f;
    a.out_bstride = (long long)WN_F * H * W;
    a.tiles_x = mrx_cdiv(W, 32);
    a.ntiles = a.tiles_x * mrx_cdiv(H, 8);
    MRX_REQUIRE((long long)H * W < (1ll << 28), MRX_EUNSUP, "mrx_rim_layer_indrnn_wino: plane %d x %d too large", H, W);
    static_assert(WN_LDS_FLOATS >= WN_F * 256 + WN_F * WN_F, "Y + ih weights must fit the staging region");
    static_assert(WN_LDS_FLOATS * sizeof(float) <= 160 * 1024, "one workgroup owns the CU's LDS");
    static_assert(12 * WN_XS(true) <= WN_XPLANE && 12 * WN_XS(false) <= WN_XPLANE, "raw tile fits its plane");
    const bool x4 = (W & 3) == 0 && W >= 4 && ((uintptr_t)x & 15) == 0;
    const size_t lds = sizeof(float) * WN_LDS_FLOATS;
    static const int abl = MRX_DEBUG_ENV("MRX_ABLATE") ? atoi(MRX_DEBUG_ENV("MRX_ABLATE")) : 0;  // debug: skip 1 transform, 2 operand reads, 4 DMA, 8 MFMA
    auto kern = x4 ? (abl == 1 ? k_rim_layer_wino<1, true> : abl == 2 ? k_rim_layer_wino<2, true> : abl == 4 ? k_rim_layer_wino<4, true>
                      : abl == 8 ? k_rim_layer_wino<8, true> : k_rim_layer_wino<0, true>)
                   : k_rim_layer_wino<0, false>;
    static bool attr_done[2] = {false, false};  // once per variant: keeps launches legal under hipGraph capture
    if (!attr_done[x4]) {
        MRX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done[x4] = true;
    }
    static unsigned long long* d_trace = nullptr;
    a.trace = nullptr;
    if (MRX_DEBUG_ENV("MRX_TRACE")) {
        if (!d_trace) (void)hipMalloc((void**)&d_trace, sizeof(unsigned long long) * 40 * 65536);
        (void)hipMemsetAsync(d_trace, 0, sizeof(unsigned long long) * 8 * 65536, (hipStream_t)stream);
        a.trace = d_trace;
    }
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        MRX_HIP(hipGetDevice(&dev));
        MRX_HIP(hipGetDeviceProperties(&prop, dev));
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const long long nt_total = (long long)a.ntiles * a.B;
    MRX_REQUIRE(nt_total < (1ll << 31), MRX_EUNSUP, "mrx_rim_layer_indrnn_wino: %lld tiles", nt_total);
    const unsigned nblk = (unsigned)(nt_total < n_cu ? nt_total : n_cu);  // persistent: one workgroup per CU walks the tiles
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(WN_NT), lds, (hipStream_t)stream, a);
    MRX_LAUNCH_CHECK();
    if (a.trace && (MRX_DEBUG_ENV("MRX_TRACE") && atoi(MRX_DEBUG_ENV("MRX_TRACE")) >= 2)) {
        (void)hipStreamSynchronize((hipStream_t)stream);
        const int nb = (int)nblk;
        std::vector<unsigned long long> h((size_t)nb * 8);
        (void)hipMemcpy(h.data(), d_trace, sizeof(unsigned long long) * 8 * nb, hipMemcpyDeviceToHost);
        double ph[4] = {0, 0, 0, 0};
        {
            std::vector<unsigned long long> wv((size_t)32 * 4);
            (void)hipMemcpy(wv.data(), d_trace + 8 * 65536 + 300 * 32, sizeof(unsigned long long) * 32 * 4, hipMemcpyDeviceToHost);
            for (int bk = 0; bk < 4; ++bk) {
                fprintf(stderr, "[mrx-trace] blk %d chunk 3, per wave stamps rel. first barrier exit (phase1 end, mma end, all landed):", 300 + bk);
                unsigned long long base = ~0ull;
                for (int w = 0; w < 8; ++w) base = wv[(bk * 8 + w) * 4] < base ? wv[(bk * 8 + w) * 4] : base;
                for (int w = 0; w < 8; ++w) {
                    const unsigned long long* r = &wv[(bk * 8 + w) * 4];
                    fprintf(stderr, " w%d(%lld: %lld %lld %lld +%lld)", w, (long long)(r[0] - base), (long long)(r[1] - base), (long long)(r[2] - base), (long long)((r[3] >> 20) - (base & 0xfffffffffffull)), (long long)(r[3] & 0xfffff));
                }
                fprintf(stderr, "\n");
            }
        }
        std::map<unsigned long long, std::pair<unsigned long long, unsigned long long>> cu;  // per-CU busy span (one XCC clock each)
        for (int i = 0; i < nb; ++i) {
            const unsigned long long* r = &h[(size_t)i * 8];
            for (int k = 0; k < 4; ++k) ph[k] += (double)(r[k + 1] - r[k]);
            auto it = cu.find(r[6]);
            if (it == cu.end()) cu[r[6]] = {r[0], r[4]};
            else {
                if (r[0] < it->second.first) it->second.first = r[0];
                if (r[4] > it->second.second) it->second.second = r[4];
            }
        }
        double spn = 0;
        for (auto& kv : cu) spn += (double)(kv.second.second - kv.second.first);
        fprintf(stderr, "[mrx-trace] k_rim_layer_wino %d blocks on %zu CUs: mean CU span %.0f cyc; mean per block: main %.0f Y %.0f 1x1 %.0f "
                        "epilogue %.0f (sum %.0f)\n", nb, cu.size(), spn / cu.size(), ph[0] / nb, ph[1] / nb, ph[2] / nb, ph[3] / nb,
                (ph[0] + ph[1] + ph[2] + ph[3]) / nb);
    }
    return MRX_OK;
}

// ---- plain 3x3 convolution into 64 channels on the same kernel (TAIL = false): dilation 1 or 2, zero or replicate padding,
// bias + activation in the output transform.  `packed` = mrx_rim_layer_wino_pack(w, NULL, ...).
extern "C" int mrx_conv3x3_wino_supported(int Cin, int Cout, int k, int dil) {
    return Cout >= WN_F && Cout % WN_F == 0 && Cin >= 1 && k == 3 && (dil == 1 || dil == 2);  // wider outputs: one launch per 64 channels
}

template <bool X4, int DIL, bool ZP>
static int launch_conv_wino(const WinoArgs& a, hipStream_t st, unsigned nblk) {
    constexpr size_t lds = sizeof(float) * WN_LDS_FLOATS;
    static bool attr_done = false;  // once per instantiation: keeps launches legal under hipGraph capture
    if (!attr_done) {
        MRX_HIP(hipFuncSetAttribute((const void*)k_rim_layer_wino<0, X4, DIL, false, ZP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    hipLaunchKernelGGL((k_rim_layer_wino<0, X4, DIL, false, ZP>), dim3(nblk), dim3(WN_NT), lds, st, a);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

extern "C" int mrx_conv3x3_wino(const float* x, const float* packed, const float* bias, float* out, int B, int Cin, int Cout, int H,
                                int W, int dil, int pad_mode, int act, float slope, void* stream) {
    MRX_REQUIRE(x && packed && out, MRX_EINVAL, "mrx_conv3x3_wino: null pointer");
    MRX_REQUIRE(B >= 0 && Cin >= 1 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_conv3x3_wino: bad dims");
    MRX_REQUIRE(mrx_conv3x3_wino_supported(Cin, Cout, 3, dil), MRX_EUNSUP, "mrx_conv3x3_wino: Cout=%d dil=%d (multiples of 64, 1|2)", Cout, dil);
    MRX_REQUIRE(pad_mode == MRX_PAD_ZERO || pad_mode == MRX_PAD_REPLICATE, MRX_EINVAL, "mrx_conv3x3_wino: pad mode %d", pad_mode);
    MRX_REQUIRE(act == MRX_ACT_NONE || act == MRX_ACT_RELU || act == MRX_ACT_LEAKY, MRX_EINVAL, "mrx_conv3x3_wino: activation %d", act);
    MRX_REQUIRE((long long)H * W < (1ll << 28), MRX_EUNSUP, "mrx_conv3x3_wino: plane %d x %d too large", H, W);
    MRX_REQUIRE(out != x, MRX_EINVAL, "mrx_conv3x3_wino: out must not alias x");
    if (B == 0) return MRX_OK;
    WinoArgs a;
    a.x = x;
    a.packed = packed;
    a.b_conv = bias;
    a.b_ih = nullptr;
    a.hh = nullptr;
    a.hprev = nullptr;
    a.hnew = out;
    a.B = B;
    a.Cin = Cin;
    a.H = H;
    a.W = W;
    a.act = act;
    a.slope = slope;
    a.trace = nullptr;
    a.tiles_x = mrx_cdiv(W, 32);
    a.ntiles = a.tiles_x * mrx_cdiv(H, 8);
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        MRX_HIP(hipGetDevice(&dev));
        MRX_HIP(hipGetDeviceProperties(&prop, dev));
        n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const long long nt_total = (long long)a.ntiles * a.B;
    MRX_REQUIRE(nt_total < (1ll << 31), MRX_EUNSUP, "mrx_conv3x3_wino: %lld tiles", nt_total);
    const unsigned nblk = (unsigned)(nt_total < n_cu ? nt_total : n_cu);
    const bool x4 = (W & 3) == 0 && W >= 4 && ((uintptr_t)x & 15) == 0;
    const bool zp = pad_mode == MRX_PAD_ZERO;
    hipStream_t st = (hipStream_t)stream;
    // one launch per block of 64 output channels: `packed` holds the blocks back to back (each mrx_rim_layer_wino_pack_floats(Cin, 64))
    const long long blk_floats = mrx_rim_layer_wino_pack_floats(Cin, WN_F), plane = (long long)H * W;
    a.out_bstride = (long long)Cout * plane;
    for (int ob = 0; ob < Cout / WN_F; ++ob) {
        a.packed = packed + ob * blk_floats;
        a.b_conv = bias ? bias + ob * WN_F : nullptr;
        a.hnew = out + (long long)ob * WN_F * plane;
        int rc;
        if (x4) {
            if (dil == 2) rc = zp ? launch_conv_wino<true, 2, true>(a, st, nblk) : launch_conv_wino<true, 2, false>(a, st, nblk);
            else rc = zp ? launch_conv_wino<true, 1, true>(a, st, nblk) : launch_conv_wino<true, 1, false>(a, st, nblk);
        } else {
            if (dil == 2) rc = zp ? launch_conv_wino<false, 2, true>(a, st, nblk) : launch_conv_wino<false, 2, false>(a, st, nblk);
            else rc = zp ? launch_conv_wino<false, 1, true>(a, st, nblk) : launch_conv_wino<false, 1, false>(a, st, nblk);
        }
        if (rc) return rc;
    }
    return MRX_OK;
}
