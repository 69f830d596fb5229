// train_bf16.hip -- the backward half of one RIM layer in the mixed-precision training arithmetic (BASELINE config 4), and the small kernels
// around the pair tensors of that path.  Reference: rim_block.py:230-238 (conv -> ReLU -> IndRNN cell), rnn_cells.py:384-391, trained under
// pytorch-lightning AMP (base_cirim_train.yaml:180): convolution results -- and therefore the gradients flowing into them -- are half-precision
// tensors, hidden states, `hh * h_prev`, FFTs and the loss are fp32.  Here "half" is bf16 and such tensors live in HBM as PAIR tensors
// u32 [B][32][H][W] = (bf16 channel 2p, bf16 channel 2p + 1): two adjacent rows of the MFMA accumulator layout are two adjacent channels, so a
// lane stores / loads one dword per row pair (128-byte segments per half-wave) and a quad of dwords IS a B operand of the next GEMM.
//
// mrx_tl_cell_bwd replaces five launches of the fp32-storage tape (ReLU backward of the cell, data gradient and weight gradient of the 1x1
// GEMM with their partial reduction, ReLU backward of the convolution) by ONE pass over the tensors:
//     g    = (dh_above + dH) * (h > 0)                       fp32  (dh_above: bf16 pairs, dH: the gradient carried from the next time-step)
//     dh_prev = g * hh;   d_hh += sum g h_prev;   gb = bf16(g);   d_b_ih += sum gb
//     da   = bf16(W_ih^T gb)                                 MFMA, contraction over the 64 output channels, B operands straight from registers
//     ga   = da * (a > 0)  -> pairs;   d_b_conv += sum ga
//     dW_ih += gb a^T                                        MFMA, contraction over the pixels: both operands transposed through LDS
// One persistent workgroup per CU (8 waves = the 8 rows of an 8 x 32 tile); per-lane partial sums stay in registers over all tiles; every workgroup
// ACCUMULATES its partial results in its own slot of `part` across the time-steps of a cascade (a workgroup always walks the same tiles: fixed
// order), and mrx_tl_cell_reduce adds the slots up once per cascade in double -- gradients are bit-reproducible.
#include "mrx_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CL_NT 512
#define CL_RS 80                  // bytes per channel row of a transposed tile: 32 pixels bf16 + 16 (16 lanes x 80 B cover the 64 banks once)
#define CL_TILE (64 * CL_RS)      // one [64 channels][32 pixels] tile
#define CL_PART (64 * 64 + 3 * 64)  // floats per workgroup slot: dW_ih [co][ci], d_b_ih, d_hh, d_b_conv

__device__ __forceinline__ unsigned tl_pk(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
__device__ __forceinline__ float tl_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float tl_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

// Sum of 32 per-lane values over the 32 lanes of a half-wave as a reduce-scatter: at the step of lane bit m a lane hands the half of its values
// that belong to the other side to lane ^ m and adds what it receives -- 31 exchanges instead of 32 x 5; lane l ends with the total of value l % 32.
// Fixed order: bit-reproducible.
template <int N>
__device__ __forceinline__ void tl_rs_step(float (&x)[32], int lane) {
    const bool up = (lane & N) != 0;
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const float send = up ? x[i] : x[i + N], keep = up ? x[i + N] : x[i];
        x[i] = keep + __shfl_xor(send, N, 64);
    }
}
__device__ __forceinline__ float tl_reduce_scatter32(float (&x)[32], int lane) {
    tl_rs_step<16>(x, lane);
    tl_rs_step<8>(x, lane);
    tl_rs_step<4>(x, lane);
    tl_rs_step<2>(x, lane);
    tl_rs_step<1>(x, lane);
    return x[0];
}

struct CellBwdArgs {
    const unsigned* dhP;   // [B,32,H,W] pairs or null: gradient from the layer above
    const float* dH;       // [B,8,H,W,8] (channel-blocked: c = 8 q + j) or null: gradient carried from the next time-step -- written as dh_prev by the previous call
    const float* h;        // [B,8,H,W,8] channel-blocked: this step's state (the cell's ReLU mask) -- not read when hmask is given
    const unsigned* hmask; // [B,H,W,2] or null: (h > 0) as 64 bits per pixel, written by the forward (mrx_tl_layer_fwd): word w, bit 16 c2 + 4 k + m = channel
                           // 32 c2 + 8 k + 4 w + m -- 8 bytes per pixel instead of 256 (the state is needed here as a mask only)
    const float* hprev;    // [B,8,H,W,8] or null (first time-step: zero state)
    const unsigned* aP;    // [B,32,H,W] pairs: a = ReLU(conv)  (the convolution's ReLU mask and the 1x1 weight gradient's operand)
    const u32x4* wT;       // mrx_tl_pack's ihT block: [4 steps][2 blocks][64 lanes]
    const float* hh;       // [64]
    float* dhp;            // [B,8,H,W,8] channel-blocked (written when hprev is given).  dH / dhp travel only between consecutive calls of this kernel, so their
                           // layout is free: a pixel's eight channels of a block are 32 contiguous bytes -- two 16-byte accesses per lane and chunk, 1 KB
                           // contiguous per half-wave, instead of eight 4-byte ones from eight planes 952 KB apart
    unsigned* gaP;         // [B,32,H,W] pairs
    float* part;           // [gridDim.x][CL_PART]
    int B, H, W, tiles_x, ntiles, first;
};

// HAS_DH / HAS_PREV: wave-uniform facts of the call as template parameters, loads unconditional from clamped coordinates (a conditional load is a
// basic block of its own: the first version of this kernel had 360 of them and spilled 92 registers), stores of a chunk under one predicate.
template <bool HAS_DH, bool HAS_PREV, bool HM>
__global__ __launch_bounds__(CL_NT, 2) void k_tl_cell_bwd(CellBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_c[];   // [8 units][gb tile, a tile]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const long long plane = (long long)a.H * a.W;
    f32x16 acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
    float s_hh = 0.f, s_bih = 0.f, s_b = 0.f;   // lane (l31, lhi) owns channel 32 lhi + l31 (s_hh, s_bih) / accumulator row l31 of its half (s_b)
    unsigned char* Tg = smem_c + wave * 2 * CL_TILE;
    unsigned char* Ta = Tg + CL_TILE;
    // hh [64] in the unused tail (bytes 64 .. 79) of the rows of the first tile: the tiles fill 80 KB exactly, two workgroups share a CU's 160 KB
    auto HH = [&](int c) -> float& { return *reinterpret_cast<float*>(smem_c + c * CL_RS + 64); };
    if (HAS_PREV && tid < 64) HH(tid) = a.hh[tid];
    // W_ih^T (the first GEMM's A fragments, 8 KB) behind the tiles: read from LDS, their waits are lgkmcnt -- as global loads issued after the next tile's
    // prefetch their vmcnt waits (in order) waited for the prefetch too
    u32x4* WT = reinterpret_cast<u32x4*>(smem_c + 8 * 2 * CL_TILE);
    for (int i = tid; i < 512; i += CL_NT) WT[i] = a.wT[i];
    __syncthreads();
    // every tensor is addressed as (wave-uniform base) + (32-bit byte offset of the lane): one address register per access instead of a 64-bit pair
    // (the first form precomputed ~60 pointers and spilled them)
    auto ldu = [](const unsigned* p, unsigned o) { return *reinterpret_cast<const unsigned*>(reinterpret_cast<const char*>(p) + o); };
    const unsigned plane4 = (unsigned)plane * 4u;
    const int total = a.ntiles * a.B;
    // Chunks 0 and 1 of tile t + grid are requested at the END of tile t's cell stage, before its two GEMM phases: the memory pipe used to sit idle from
    // there to the next tile's first request (a third of the tile's time).  The loop starts one round early -- that round only issues -- so that this is
    // the one place they are requested from (two copies make hipcc wait for the loads at the loop head, see k_conv_wgrad_bf16).
    unsigned d2[2][4];
    float dHv[2][8], hv[2][8], hpv[2][8];
    struct TileOff { unsigned pix4, pb, ab, cbb; int b; bool valid; };
    auto tile_off = [&](int t) {
        TileOff o;
        const int b = t / a.ntiles, tt = t - b * a.ntiles;
        const int ty0 = tt / a.tiles_x, oy = ty0 * 8 + wave, ox = (tt - ty0 * a.tiles_x) * 32 + l31;
        o.b = b, o.valid = oy < a.H && ox < a.W;
        o.pix4 = (unsigned)((oy < a.H ? oy : a.H - 1) * a.W + (ox < a.W ? ox : a.W - 1)) * 4u;
        o.pb = (unsigned)b * 32u * plane4 + o.pix4 + 16u * lhi * plane4;
        o.ab = (unsigned)b * 32u * plane4 + o.pix4 + 2u * lhi * plane4;
        o.cbb = (unsigned)b * 64u * plane4 + o.pix4 * 8u + 4u * lhi * 8u * plane4;      // channel-blocked: block 4 lhi + ch, 32 bytes per pixel
        return o;
    };
    // eight channels at a time, the NEXT chunk's 28 loads in flight while this one is processed
    auto request = [&](unsigned pb, unsigned cbb, int ch, int bf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) d2[bf][q] = ldu(a.dhP, pb + (unsigned)(4 * ch + q) * plane4);
        if (!HM) {
            const float4* q4 = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a.h) + cbb + (unsigned)ch * 8u * plane4);
            const float4 u0 = q4[0], u1 = q4[1];
            hv[bf][0] = u0.x, hv[bf][1] = u0.y, hv[bf][2] = u0.z, hv[bf][3] = u0.w, hv[bf][4] = u1.x, hv[bf][5] = u1.y, hv[bf][6] = u1.z, hv[bf][7] = u1.w;
        }
        if (HAS_PREV) {
            const float4* q4 = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a.hprev) + cbb + (unsigned)ch * 8u * plane4);
            const float4 u0 = q4[0], u1 = q4[1];
            hpv[bf][0] = u0.x, hpv[bf][1] = u0.y, hpv[bf][2] = u0.z, hpv[bf][3] = u0.w, hpv[bf][4] = u1.x, hpv[bf][5] = u1.y, hpv[bf][6] = u1.z, hpv[bf][7] = u1.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) hpv[bf][j] = 0.f;
        }
        if (HAS_DH) {
            const float4* q4 = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a.dH) + cbb + (unsigned)ch * 8u * plane4);
            const float4 u0 = q4[0], u1 = q4[1];
            dHv[bf][0] = u0.x, dHv[bf][1] = u0.y, dHv[bf][2] = u0.z, dHv[bf][3] = u0.w, dHv[bf][4] = u1.x, dHv[bf][5] = u1.y, dHv[bf][6] = u1.z, dHv[bf][7] = u1.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) dHv[bf][j] = 0.f;
        }
    };
    for (int t = (int)blockIdx.x - (int)gridDim.x; t < total; t += gridDim.x) {
        const bool cur = t >= 0;
        const TileOff o_ = tile_off(cur ? t : 0);
        const int b = o_.b;
        const bool valid = o_.valid;
        const unsigned pix4 = o_.pix4, ab = o_.ab;
        const unsigned pb = o_.pb, cbb = o_.cbb;
        unsigned gbP[16];
        unsigned awv[16];
        float t_hh[32];
        if (cur) {
        // ---- cell stage: lane = pixel, channels 32 lhi + i; eight channels at a time -----------------------------------------------------------
        // the pair tensor `a` is needed after the first GEMM only: requested first, it arrives under the cell stage
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int q = 0; q < 8; ++q) awv[ct * 8 + q] = ldu(a.aP, ab + (unsigned)(ct * 16 + (q & 1) + 4 * (q >> 1)) * plane4);
        uint2 hm = make_uint2(0u, 0u);
        if (HM) hm = *reinterpret_cast<const uint2*>(reinterpret_cast<const char*>(a.hmask) + (unsigned)b * 2u * plane4 + pix4 * 2u);
        // (chunks 0 and 1 of this tile were requested by the previous round; chunk c + 2 goes into the buffer chunk c has just left)
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) {
            const int bf = ch & 1;
            float g[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float up = ((j & 1) ? tl_hi(d2[bf][j >> 1]) : tl_lo(d2[bf][j >> 1])) + dHv[bf][j];
                // HM: channel 32 lhi + 8 ch + j = word (j >> 2), bit 16 lhi + 4 ch + (j & 3) of the tile's mask words
                const bool on = HM ? ((((j & 4) ? hm.y : hm.x) >> (16 * lhi + 4 * ch + (j & 3))) & 1u) != 0u : hv[bf][j] > 0.f;
                g[j] = (valid && on) ? up : 0.f;
                t_hh[8 * ch + j] = g[j] * hpv[bf][j];
            }
            if (ch < 2) request(pb, cbb, ch + 2, bf);
            if (HAS_PREV && valid) {
                float4* q4 = reinterpret_cast<float4*>(reinterpret_cast<char*>(a.dhp) + cbb + (unsigned)ch * 8u * plane4);
                const int c0 = 32 * lhi + 8 * ch;
                q4[0] = make_float4(g[0] * HH(c0), g[1] * HH(c0 + 1), g[2] * HH(c0 + 2), g[3] * HH(c0 + 3));
                q4[1] = make_float4(g[4] * HH(c0 + 4), g[5] * HH(c0 + 5), g[6] * HH(c0 + 6), g[7] * HH(c0 + 7));
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) gbP[4 * ch + q] = tl_pk(g[2 * q], g[2 * q + 1]);
            __builtin_amdgcn_sched_barrier(0);      // keeps the order above: two chunks of loads in flight, not four (hoisting everything spills)
        }
        }       // cur
        {       // the one place the first two chunks of a tile are requested: the NEXT tile's, ahead of this tile's GEMM phases
            const int tn = t + (int)gridDim.x;
            const TileOff on_ = tile_off(tn < total ? tn : total - 1);
            request(on_.pb, on_.cbb, 0, 0);
            request(on_.pb, on_.cbb, 1, 1);
        }
        if (!cur) continue;
        if (HAS_PREV) s_hh += tl_reduce_scatter32(t_hh, lane);
        {
            float t[32];
#pragma unroll
            for (int i = 0; i < 32; ++i) t[i] = (i & 1) ? tl_hi(gbP[i >> 1]) : tl_lo(gbP[i >> 1]);
            s_bih += tl_reduce_scatter32(t, lane);
        }
        // ---- da = W_ih^T gb: contraction over the output channels, B operands = the packed gradient as it sits in the registers ------------------
        f32x16 acc[2];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
        const u32x4* wp = WT + lane;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const u32x4 bw = {gbP[4 * s], gbP[4 * s + 1], gbP[4 * s + 2], gbP[4 * s + 3]};
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
                acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wp[(s * 2 + ct) * 64]), __builtin_bit_cast(bf16x8, bw), acc[ct], 0,
                                                                 0, 0);
        }
        // gb transposed into LDS: [channel][pixel]
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const unsigned p = gbP[i >> 1];
            *reinterpret_cast<unsigned short*>(Tg + (32 * lhi + i) * CL_RS + l31 * 2) = (unsigned short)((i & 1) ? (p >> 16) : (p & 0xffffu));
        }
        // ---- ga = da * (a > 0), a transposed into LDS -------------------------------------------------------------------------------------------
        float t_b[32];
        unsigned gpv[16];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int r = 2 * q, ci = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                const unsigned aw = valid ? awv[ct * 8 + q] : 0u;
                const float v0 = (aw & 0x7fffu) ? acc[ct][r] : 0.f, v1 = (aw & 0x7fff0000u) ? acc[ct][r + 1] : 0.f;
                const unsigned gp = tl_pk(v0, v1);
                gpv[ct * 8 + q] = gp;
                t_b[ct * 16 + r] = tl_lo(gp);
                t_b[ct * 16 + r + 1] = tl_hi(gp);
                *reinterpret_cast<unsigned short*>(Ta + ci * CL_RS + l31 * 2) = (unsigned short)(aw & 0xffffu);
                *reinterpret_cast<unsigned short*>(Ta + (ci + 1) * CL_RS + l31 * 2) = (unsigned short)(aw >> 16);
            }
        if (valid) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    *reinterpret_cast<unsigned*>(reinterpret_cast<char*>(a.gaP) + ab + (unsigned)(ct * 16 + (q & 1) + 4 * (q >> 1)) * plane4) = gpv[ct * 8 + q];
                }
        }
        s_b += tl_reduce_scatter32(t_b, lane);
        __syncthreads();
        // ---- dW_ih += gb a^T over the 8 x 32 pixels of the tile: wave = (cout block, cin block, half of the rows) ---------------------------------
        {
            const int bi = wave & 1, bj = (wave >> 1) & 1, uh = wave >> 2;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const unsigned char* base = smem_c + (uh * 4 + u) * 2 * CL_TILE;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const bf16x8 av = *reinterpret_cast<const bf16x8*>(base + (bi * 32 + l31) * CL_RS + (kk * 16 + lhi * 8) * 2);
                    const bf16x8 bv = *reinterpret_cast<const bf16x8*>(base + CL_TILE + (bj * 32 + l31) * CL_RS + (kk * 16 + lhi * 8) * 2);
                    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc2, 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }
    // ---- workgroup results -> this workgroup's slot (fixed order everywhere) ----------------------------------------------------------------------
    float* R = reinterpret_cast<float*>(smem_c);          // [8 waves][64 lanes x 16] for the weight block, then [8 waves][3][64] for the sums
    __syncthreads();
    if (wave >= 4) {
#pragma unroll
        for (int r = 0; r < 16; ++r) R[((wave - 4) * 16 + r) * 64 + lane] = acc2[r];
    }
    __syncthreads();
    float* slot = a.part + (long long)blockIdx.x * CL_PART;
    if (wave < 4) {
        const int bi = wave & 1, bj = (wave >> 1) & 1;
        // (the slot's sixteen old values are requested in one block: as `first ? v : slot + v` per element every load sat behind its own branch and
        // wait -- sixteen round trips at the very end of every workgroup)
        float old[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) old[r] = 0.f;
        if (!a.first) {
#pragma unroll
            for (int r = 0; r < 16; ++r) old[r] = slot[(32 * bi + (r & 3) + 8 * (r >> 2) + 4 * lhi) * 64 + 32 * bj + l31];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = acc2[r] + R[(wave * 16 + r) * 64 + lane];
            const int co = 32 * bi + (r & 3) + 8 * (r >> 2) + 4 * lhi, ci = 32 * bj + l31;
            slot[co * 64 + ci] = a.first ? v : old[r] + v;
        }
    }
    __syncthreads();
    // per-channel sums: every lane holds the wave's total of its own channel; across the waves in wave order
    float* S = reinterpret_cast<float*>(smem_c);          // [8 waves][3][64]
    S[(wave * 3 + 0) * 64 + 32 * lhi + l31] = s_bih;
    S[(wave * 3 + 1) * 64 + 32 * lhi + l31] = s_hh;
    {
        const int ct = l31 >> 4, r = l31 & 15;
        S[(wave * 3 + 2) * 64 + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi] = s_b;
    }
    __syncthreads();
    if (tid < 192) {
        const int qn = tid >> 6, c = tid & 63;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += S[(w * 3 + qn) * 64 + c];
        float* dst = slot + 4096 + qn * 64 + c;
        *dst = a.first ? v : *dst + v;
    }
}

static int tl_nwg(long long tiles) {
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return (int)(tiles < n_cu ? tiles : n_cu);
}
static int tl_cell_nwg(long long tiles) {       // one workgroup per CU (two -- 80 KB of LDS, 128 registers with 14 spilled -- measured 123 us against 99)
    return tl_nwg(tiles);
}
extern "C" int64_t mrx_tl_cell_part_floats(int B, int H, int W) {
    if (B < 1 || H < 1 || W < 1) return -1;
    return (int64_t)tl_cell_nwg((long long)B * mrx_cdiv(W, 32) * mrx_cdiv(H, 8)) * CL_PART;
}
// The backward pass of one IndRNN layer's cell + convolution ReLU (see the header).  dH (fp32, CHANNEL-BLOCKED [B,8,H,W,8] -- it is the dh_prev a
// previous call wrote) may be null (last time-step); hprev null =
// first time-step (no dh_prev, no hh gradient).  `part` [mrx_tl_cell_part_floats]: the workgroup slots; first != 0 overwrites them (first call of a
// cascade), otherwise the call adds to them.  tl_packed from mrx_tl_pack.
extern "C" int mrx_tl_cell_bwd(const void* dh_above, const float* dH, const float* h, const void* hmask, const float* hprev, const void* a_pairs,
                               const void* tl_packed, const float* hh, float* dh_prev, void* ga_pairs, float* part, int first, int B, int H, int W,
                               void* stream) {
    MRX_REQUIRE((h || hmask) && a_pairs && tl_packed && ga_pairs && part, MRX_EINVAL, "mrx_tl_cell_bwd: null pointer");
    MRX_REQUIRE(!hprev || (hh && dh_prev), MRX_EINVAL, "mrx_tl_cell_bwd: hprev without hh / dh_prev");
    MRX_REQUIRE(B >= 1 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_tl_cell_bwd: bad dims");
    MRX_REQUIRE((long long)B * 64 * H * W * 4 < (1ll << 32), MRX_EUNSUP, "mrx_tl_cell_bwd: tensors of 4 GB and more (32-bit byte offsets)");
    CellBwdArgs a;
    a.dhP = (const unsigned*)dh_above, a.dH = dH, a.h = h, a.hmask = (const unsigned*)hmask, a.hprev = hprev, a.aP = (const unsigned*)a_pairs;
    a.wT = (const u32x4*)tl_packed + 768, a.hh = hh, a.dhp = dh_prev, a.gaP = (unsigned*)ga_pairs, a.part = part;
    a.B = B, a.H = H, a.W = W, a.tiles_x = mrx_cdiv(W, 32), a.ntiles = a.tiles_x * mrx_cdiv(H, 8), a.first = first;
    MRX_REQUIRE(dh_above, MRX_EINVAL, "mrx_tl_cell_bwd: the gradient from the layer above is required");
    constexpr int lds = 8 * 2 * CL_TILE + 512 * 16;
    const dim3 grid(tl_cell_nwg((long long)B * a.ntiles));
    hipStream_t st = (hipStream_t)stream;
    const int which = (dH ? 4 : 0) | (hprev ? 2 : 0) | (hmask ? 1 : 0);
#define CL_CASE(I, D, P, M)                                                                                                              \
    case I: {                                                                                                                            \
        static bool attr_done = false;                                                                                                   \
        if (!attr_done) {                                                                                                                \
            MRX_HIP(hipFuncSetAttribute((const void*)k_tl_cell_bwd<D, P, M>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));          \
            attr_done = true;                                                                                                            \
        }                                                                                                                                \
        hipLaunchKernelGGL((k_tl_cell_bwd<D, P, M>), grid, dim3(CL_NT), lds, st, a);                                                     \
        break;                                                                                                                           \
    }
    switch (which) {
        CL_CASE(0, false, false, false)
        CL_CASE(1, false, false, true)
        CL_CASE(2, false, true, false)
        CL_CASE(3, false, true, true)
        CL_CASE(4, true, false, false)
        CL_CASE(5, true, false, true)
        CL_CASE(6, true, true, false)
        CL_CASE(7, true, true, true)
    }
#undef CL_CASE
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// gradients += the workgroup slots, added in slot order in double: dW_ih [64,64], d_b_ih [64], d_hh [64], d_b_conv [64] (null = not wanted)
__global__ __launch_bounds__(256) void k_tl_cell_reduce(const float* __restrict__ part, int nparts, float* dw, float* dbih, float* dhh, float* db) {
    __shared__ double sh[16][17];
    const int li = threadIdx.x & 15, lp = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + li;
    double s = 0.0;
    if (i < CL_PART)
        for (int p = lp; p < nparts; p += 16) s += (double)part[(long long)p * CL_PART + i];
    sh[lp][li] = s;
    __syncthreads();
    if (lp == 0 && i < CL_PART) {
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sh[k][li];
        float* dst = i < 4096 ? dw + i : (i < 4160 ? (dbih ? dbih + (i - 4096) : nullptr) : (i < 4224 ? (dhh ? dhh + (i - 4160) : nullptr) : (db ? db + (i - 4224) : nullptr)));
        if (dst) *dst += (float)t;
    }
}
extern "C" int mrx_tl_cell_reduce(const float* part, int B, int H, int W, float* dw_ih, float* db_ih, float* dhh, float* db_conv, void* stream) {
    MRX_REQUIRE(part && dw_ih, MRX_EINVAL, "mrx_tl_cell_reduce: null pointer");
    const int n = tl_cell_nwg((long long)B * mrx_cdiv(W, 32) * mrx_cdiv(H, 8));
    hipLaunchKernelGGL(k_tl_cell_reduce, dim3((CL_PART + 15) / 16), dim3(256), 0, (hipStream_t)stream, part, n, dw_ih, db_ih, dhh, db_conv);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- the edge pixels of a replicate-padded data gradient whose interior is in `dx` already (mrx_tl_dgrad): every edge pixel is the sum of the frame
// positions that clamp to it -- its own unrounded value included, which mrx_tl_dgrad left in the frame tensor -- rounded to bf16 ONCE (dx is a bf16
// tensor: pairs, or fp32 holding bf16 values) ----------------------------------------------------------------------------------------------------
__global__ void k_tl_fold_edges(const float* __restrict__ g, void* __restrict__ dx, int pairs, int B, int C, int H, int W, int pad) {
    const int per = 2 * W + 2 * (H - 2 > 0 ? H - 2 : 0);
    const int nch = pairs ? C / 2 : C;
    const long long total = (long long)B * nch * per;
    const int PW = W + 2 * pad, PH = H + 2 * pad;
    for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (long long)gridDim.x * blockDim.x) {
        const long long p = o / per;
        int e = (int)(o - p * per), h, w;
        if (e < W) {
            h = 0, w = e;
        } else if (e < 2 * W) {
            h = H - 1, w = e - W;
            if (H == 1) continue;
        } else {
            e -= 2 * W;
            h = 1 + (e >> 1), w = (e & 1) ? W - 1 : 0;
            if (W == 1 && (e & 1)) continue;
        }
        const int i0 = h == 0 ? 0 : h + pad, i1 = h == H - 1 ? PH - 1 : h + pad;
        const int j0 = w == 0 ? 0 : w + pad, j1 = w == W - 1 ? PW - 1 : w + pad;
        const int bb = (int)(p / nch), cc = (int)(p - (long long)bb * nch);
        float s[2] = {0.f, 0.f};
        for (int k = 0; k < (pairs ? 2 : 1); ++k) {
            const float* gp = g + ((long long)bb * C + (pairs ? 2 * cc + k : cc)) * PH * PW;
            for (int i = i0; i <= i1; ++i)
                for (int j = j0; j <= j1; ++j) s[k] += gp[(long long)i * PW + j];
        }
        const long long at = p * (long long)H * W + (long long)h * W + w;
        if (pairs)
            reinterpret_cast<unsigned*>(dx)[at] = tl_pk(s[0], s[1]);
        else
            reinterpret_cast<float*>(dx)[at] = tl_lo(tl_pk(s[0], 0.f));
    }
}
extern "C" int mrx_tl_fold_edges(const float* frame, void* dx, int dx_pairs, int B, int C, int H, int W, int pad, void* stream) {
    MRX_REQUIRE(frame && dx && B >= 1 && C >= 1 && H >= 1 && W >= 1 && pad >= 0 && (!dx_pairs || C % 2 == 0), MRX_EINVAL, "mrx_tl_fold_edges: bad argument");
    const long long total = (long long)B * (dx_pairs ? C / 2 : C) * (2ll * W + 2ll * (H - 2 > 0 ? H - 2 : 0));
    if (total == 0 || pad == 0) return MRX_OK;
    const long long nb = (total + 255) / 256;
    hipLaunchKernelGGL(k_tl_fold_edges, dim3((unsigned)(nb < 65535 ? nb : 65535)), dim3(256), 0, (hipStream_t)stream, frame, dx, dx_pairs, B, C, H, W, pad);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- eta_new = eta + bf16(sum of the nine shifted tap planes): the final convolution of a RIM step from mrx_tl_layer_fwd's tap products
// (rim_block.py:239-248; replicate padding = clamped coordinates) ------------------------------------------------------------------------------------
// MX: also the workgroup's maximum of |eta_new| (complex modulus formed as mrx_max_abs forms it) -> maxpart[workgroup]: the training loss's maximum
// without a pass of its own (mrx_absl1_loss_mp reduces the partials)
template <bool MX>
__global__ __launch_bounds__(256) void k_tl_final_gather(const float* __restrict__ taps, const float* __restrict__ eta, float* __restrict__ out, int H, int W,
                                                         float* __restrict__ maxpart) {
    __shared__ float red[4];
    const int x0 = blockIdx.x * 64 + (threadIdx.x & 63), y0 = blockIdx.y * 4 + (threadIdx.x >> 6), b = blockIdx.z;
    const bool inside = x0 < W && y0 < H;
    if (!MX && !inside) return;
    const int x = x0 < W ? x0 : W - 1, y = y0 < H ? y0 : H - 1;       // (MX: every thread reaches the reduction; an outside thread repeats an inside pixel)
    const long long plane = (long long)H * W;
    const float* pb = taps + (long long)b * 18 * plane;
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        int yy = y + dy - 1;
        yy = yy < 0 ? 0 : (yy >= H ? H - 1 : yy);
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            int xx = x + dx - 1;
            xx = xx < 0 ? 0 : (xx >= W ? W - 1 : xx);
            const float* p = pb + (long long)((dy * 3 + dx) * 2) * plane + (long long)yy * W + xx;
            s0 += p[0];
            s1 += p[plane];
        }
    }
    const unsigned r = tl_pk(s0, s1);
    const long long o = ((long long)b * plane + (long long)y * W + x) * 2;
    const float vx = eta[o] + tl_lo(r), vy = eta[o + 1] + tl_hi(r);
    if (inside) out[o] = vx, out[o + 1] = vy;
    if (MX) {
        float m = (float)sqrt((double)mrx_sumsq2(vx, vy));
        for (int off = 32; off > 0; off >>= 1) {
            const float t = __shfl_xor(m, off, 64);
            m = (t > m || t != t) ? t : m;
        }
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; ++w) m = (red[w] > m || red[w] != red[w]) ? red[w] : m;
            maxpart[((long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = m;
        }
    }
}
extern "C" int mrx_tl_final_gather(const float* taps, const float* eta, float* eta_out, int B, int H, int W, void* stream) {
    MRX_REQUIRE(taps && eta && eta_out && B >= 1 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_tl_final_gather: bad argument");
    hipLaunchKernelGGL(k_tl_final_gather<false>, dim3(mrx_cdiv(W, 64), mrx_cdiv(H, 4), B), dim3(256), 0, (hipStream_t)stream, taps, eta, eta_out, H, W,
                       (float*)nullptr);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
// ... and the per-workgroup maxima of |eta_out| (complex modulus): max_partials [mrx_tl_final_gather_max_count(B, H, W)]
extern "C" int64_t mrx_tl_final_gather_max_count(int B, int H, int W) {
    if (B < 1 || H < 1 || W < 1) return -1;
    return (int64_t)mrx_cdiv(W, 64) * mrx_cdiv(H, 4) * B;
}
extern "C" int mrx_tl_final_gather_max(const float* taps, const float* eta, float* eta_out, float* max_partials, int B, int H, int W, void* stream) {
    MRX_REQUIRE(taps && eta && eta_out && max_partials && B >= 1 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_tl_final_gather_max: bad argument");
    hipLaunchKernelGGL(k_tl_final_gather<true>, dim3(mrx_cdiv(W, 64), mrx_cdiv(H, 4), B), dim3(256), 0, (hipStream_t)stream, taps, eta, eta_out, H, W,
                       max_partials);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- pair tensors <-> fp32 NCHW (tests, and the first gradient of a step: d_eta [B,H,W,2] -> bf16 values as fp32 [B,2,H,W]) ---------------------
__global__ void k_tl_pairs_to_f32(const unsigned* __restrict__ p, float* __restrict__ out, long long planes, long long plane) {
    const long long n = planes * plane;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long pl = i / plane, px = i - pl * plane;
        const unsigned v = p[i];
        out[(2 * pl) * plane + px] = tl_lo(v);
        out[(2 * pl + 1) * plane + px] = tl_hi(v);
    }
}
__global__ void k_tl_f32_to_pairs(const float* __restrict__ x, unsigned* __restrict__ out, long long planes, long long plane) {
    const long long n = planes * plane;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long pl = i / plane, px = i - pl * plane;
        out[i] = tl_pk(x[(2 * pl) * plane + px], x[(2 * pl + 1) * plane + px]);
    }
}
extern "C" int mrx_tl_pairs_to_f32(const void* pairs, float* out, int64_t pair_planes, int64_t plane, void* stream) {
    MRX_REQUIRE(pairs && out && pair_planes >= 0 && plane >= 0, MRX_EINVAL, "mrx_tl_pairs_to_f32: bad argument");
    const long long n = pair_planes * plane, nb = (n + 255) / 256;
    if (n == 0) return MRX_OK;
    hipLaunchKernelGGL(k_tl_pairs_to_f32, dim3((unsigned)(nb < 65535 ? nb : 65535)), dim3(256), 0, (hipStream_t)stream, (const unsigned*)pairs, out, (long long)pair_planes,
                       (long long)plane);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
extern "C" int mrx_tl_f32_to_pairs(const float* x, void* pairs, int64_t pair_planes, int64_t plane, void* stream) {
    MRX_REQUIRE(pairs && x && pair_planes >= 0 && plane >= 0, MRX_EINVAL, "mrx_tl_f32_to_pairs: bad argument");
    const long long n = pair_planes * plane, nb = (n + 255) / 256;
    if (n == 0) return MRX_OK;
    hipLaunchKernelGGL(k_tl_f32_to_pairs, dim3((unsigned)(nb < 65535 ? nb : 65535)), dim3(256), 0, (hipStream_t)stream, x, (unsigned*)pairs, (long long)pair_planes,
                       (long long)plane);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- weight gradient of the FIRST RIM layer (5x5, Cin <= 5 -> 64; conv_layers.py:121-123 backwards) from a pair tensor ------------------------------
// dW[co][ci][tap] = sum_px dy[co][px] x[ci][px + tap]: a GEMM D[co][n] += A[co][px] B[px][n] with n = ci * 25 + tap (<= 125: four 32-column blocks)
// and the contraction over the pixels.  The generic thin kernel (conv_bf16.hip, one tap per wave, the input channels across the lanes: 28 of 32
// lanes multiply zeros) needed 800 MFMAs per 8 x 32 tile, spilled 84 registers and took 131 us; here a wave owns one column block: lane n gathers
// its eight consecutive pixels of x[ci] at the (ky, kx) shift of its own column from the halo'd bf16 tile in LDS (five aligned dwords + a
// per-lane funnel shift for odd kx), 64 MFMAs per 4 x 32 tile, 20 KB of LDS: several workgroups per CU cover each other's loads.
#define WI_TH 4
#define WI_NT 256
#define WI_DYS (WI_TH * 32 * 2 + 16)      // bytes per dy channel row: 128 pixels bf16 + 16
template <int K>
__global__ __launch_bounds__(WI_NT, 2) void k_tl_wgrad_in(const float* __restrict__ x, const unsigned* __restrict__ dyP, float* __restrict__ part, int B, int Cin,
                                                         int H, int W, int tiles_x, int ntiles) {
    constexpr int PAD = (K - 1) / 2, PH = WI_TH + 2 * PAD, PW = 32 + 2 * PAD + 4, TAPS = K * K;   // PW: row stride in elements (+4: the 5-dword window of the last lane group)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_w[];
    unsigned char* Dy = smem_w;                                   // [64][WI_DYS]
    unsigned short* Xh = reinterpret_cast<unsigned short*>(smem_w + 64 * WI_DYS);   // [Cin][PH][PW] bf16, then one zero row of PH * PW
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const long long plane = (long long)H * W;
    const int N = Cin * TAPS;
    // this lane's column: n = 32 wave + l31 -> (ci, ky, kx); columns past N read the zero plane
    const int n = 32 * wave + l31;
    const int ci = n < N ? n / TAPS : Cin, tp = n < N ? n - ci * TAPS : 0, ky = tp / K, kx = tp - ky * K;
    const int xoff = (ci * PH + ky) * PW + (kx & ~1);             // even element offset of the lane's window; an odd kx shifts by 16 bits
    const unsigned sh = (kx & 1) ? 16u : 0u;
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int i = tid; i < PH * PW; i += WI_NT) Xh[Cin * PH * PW + i] = 0;
    const int total = ntiles * B;
    const bool vec = (W % 4 == 0) && ((reinterpret_cast<uintptr_t>(dyP) & 15u) == 0);
    for (int t_ = blockIdx.x; t_ < total; t_ += gridDim.x) {
        const int t = (gridDim.x & 7u) == 0u ? (int)mrx_xcd_band(t_, total) : t_;      // (XCD band order: a tile's halo rows meet their neighbours' in one L2)
        const int b = t / ntiles, tt = t - b * ntiles, ty0 = tt / tiles_x, h0 = ty0 * WI_TH, w0 = (tt - ty0 * tiles_x) * 32;
        __syncthreads();
        // dy tile from the pair tensor: item = (pair, row, 8-pixel group), exactly two per thread; halo'd x tile: (channel, row, pixel pair), up to four per
        // thread.  Every load of the tile is requested before the first LDS write (as run-time loops each iteration's loads were waited for before the next
        // iteration's went out: five memory round trips per tile, one after the other).
        const unsigned* dyb = dyP + (long long)b * 32 * plane;
        constexpr int DIT = 32 * WI_TH * 4 / WI_NT, XIT = (5 * PH * (PW / 2) + WI_NT - 1) / WI_NT;
        static_assert(DIT * WI_NT == 32 * WI_TH * 4, "two dy items per thread");
        // (x: lanes past the last item repeat the last item -- same address, same value, same LDS word: the write below stays unconditional, so the
        // compiler cannot sink a load into the block of its predicated use)
        const float* xb = x + (long long)b * Cin * plane;
        const int nxi = Cin * PH * (PW / 2);
        float xa0[XIT], xa1[XIT];
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int i = min(tid + it * WI_NT, nxi - 1);
            const int c2 = i % (PW / 2), r = (i / (PW / 2)) % PH, c = i / ((PW / 2) * PH);
            int gy = h0 + r - PAD;
            gy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy);
            int gx0 = w0 + 2 * c2 - PAD, gx1 = gx0 + 1;
            gx0 = gx0 < 0 ? 0 : (gx0 >= W ? W - 1 : gx0);
            gx1 = gx1 < 0 ? 0 : (gx1 >= W ? W - 1 : gx1);
            const float* row = xb + (long long)c * plane + (long long)gy * W;
            xa0[it] = row[gx0], xa1[it] = row[gx1];
        }
        auto dy_commit = [&](int i, const unsigned (&d)[8]) {
            const int pg = i & 3, r = (i >> 2) % WI_TH, pp = i / (4 * WI_TH);
            u32x4 lo, hi;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                lo[q] = (d[2 * q] & 0xffffu) | (d[2 * q + 1] << 16);
                hi[q] = (d[2 * q] >> 16) | (d[2 * q + 1] & 0xffff0000u);
            }
            *reinterpret_cast<u32x4*>(Dy + (2 * pp) * WI_DYS + (r * 32 + pg * 8) * 2) = lo;
            *reinterpret_cast<u32x4*>(Dy + (2 * pp + 1) * WI_DYS + (r * 32 + pg * 8) * 2) = hi;
        };
        if (vec && W >= 8) {
            // W % 4 == 0: a group of eight starts at a multiple of 8, so it lies inside the row, or has exactly its first four pixels inside, or none.
            // Both 16-byte loads always come from the clamped group start min(gx, W - 8) -- no branch around a load; the three cases are selects.
            uint4 q0[DIT], q1[DIT];
#pragma unroll
            for (int it = 0; it < DIT; ++it) {
                const int i = tid + it * WI_NT;
                const int pg = i & 3, r = (i >> 2) % WI_TH, pp = i / (4 * WI_TH);
                const int gy = h0 + r, gx = w0 + pg * 8, gxc = gx < W - 8 ? gx : W - 8;
                const unsigned* src = dyb + (long long)pp * plane + (long long)(gy < H ? gy : H - 1) * W + gxc;
                q0[it] = *reinterpret_cast<const uint4*>(src), q1[it] = *reinterpret_cast<const uint4*>(src + 4);
            }
#pragma unroll
            for (int it = 0; it < DIT; ++it) {
                const int i = tid + it * WI_NT;
                const int pg = i & 3, r = (i >> 2) % WI_TH;
                const int gy = h0 + r, gx = w0 + pg * 8;
                const bool row_in = gy < H, full = row_in && gx + 8 <= W, part = row_in && !full && gx < W;     // part: pixels gx .. gx + 3 = the second load
                const uint4 lo4 = full ? q0[it] : (part ? q1[it] : make_uint4(0u, 0u, 0u, 0u)), hi4 = full ? q1[it] : make_uint4(0u, 0u, 0u, 0u);
                const unsigned d[8] = {lo4.x, lo4.y, lo4.z, lo4.w, hi4.x, hi4.y, hi4.z, hi4.w};
                dy_commit(i, d);
            }
        } else {
            for (int i = tid; i < 32 * WI_TH * 4; i += WI_NT) {
                const int pg = i & 3, r = (i >> 2) % WI_TH, pp = i / (4 * WI_TH);
                const int gy = h0 + r, gx = w0 + pg * 8;
                const unsigned* src = dyb + (long long)pp * plane + (long long)(gy < H ? gy : H - 1) * W;
                unsigned d[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int xx = gx + j;
                    const unsigned v = src[xx < W ? xx : W - 1];
                    d[j] = (gy < H && xx < W) ? v : 0u;
                }
                dy_commit(i, d);
            }
        }
        // halo'd x tile, replicate padding, fp32 -> bf16 (two pixels per thread and step)
#pragma unroll
        for (int it = 0; it < XIT; ++it) {
            const int i = min(tid + it * WI_NT, nxi - 1);
            const int c2 = i % (PW / 2), r = (i / (PW / 2)) % PH, c = i / ((PW / 2) * PH);
            *reinterpret_cast<unsigned*>(Xh + (c * PH + r) * PW + 2 * c2) = tl_pk(xa0[it], xa1[it]);
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < WI_TH; ++r)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int px = kk * 16 + lhi * 8;                  // first of the lane's eight pixels in row r
                const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(Dy + l31 * WI_DYS + (r * 32 + px) * 2);
                const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(Dy + (32 + l31) * WI_DYS + (r * 32 + px) * 2);
                const unsigned* q = reinterpret_cast<const unsigned*>(Xh + xoff + r * PW + px);
                const unsigned d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];
                const u32x4 bw = {__builtin_amdgcn_alignbit(d1, d0, sh), __builtin_amdgcn_alignbit(d2, d1, sh), __builtin_amdgcn_alignbit(d3, d2, sh),
                                  __builtin_amdgcn_alignbit(d4, d3, sh)};
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, __builtin_bit_cast(bf16x8, bw), acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, __builtin_bit_cast(bf16x8, bw), acc[1], 0, 0, 0);
            }
    }
    float* po = part + (long long)blockIdx.x * 64 * N;
    if (n < N) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) po[(32 * i + (r & 3) + 8 * (r >> 2) + 4 * lhi) * N + n] = acc[i][r];
    }
}
__global__ __launch_bounds__(256) void k_tl_part_reduce(const float* __restrict__ part, int nparts, long long n, float* __restrict__ dw, int accumulate) {
    mrx_reduce_parts(part, nparts, n, dw, accumulate);
}
static int wi_nwg(long long tiles) {
    const long long cap = 2ll * tl_nwg(1ll << 40);            // (three per CU measured 29.7 us + 17.3 us for the 768-partial reduction)
    return (int)(tiles < cap ? tiles : cap);
}
extern "C" int64_t mrx_tl_wgrad_in_work_floats(int B, int Cin, int H, int W) {
    if (B < 1 || Cin < 1 || Cin > 5 || H < 1 || W < 1) return -1;
    return (int64_t)wi_nwg((long long)B * mrx_cdiv(W, 32) * mrx_cdiv(H, WI_TH)) * 64 * Cin * 25;
}
// dw [64,Cin,5,5] (= or +=) the weight gradient of the replicate-padded 5x5 convolution Cin <= 5 -> 64 with x fp32 [B,Cin,H,W] (rounded to bf16 by the
// loader) and dy a pair tensor [B,32,H,W]; bf16 products, fp32 sums per workgroup, fixed-order double sum of the workgroup partials.
extern "C" int mrx_tl_wgrad_in(const float* x, const void* dy_pairs, float* dw, float* work, int B, int Cin, int H, int W, int accumulate, void* stream) {
    MRX_REQUIRE(x && dy_pairs && dw && work, MRX_EINVAL, "mrx_tl_wgrad_in: null pointer");
    MRX_REQUIRE(B >= 1 && Cin >= 1 && Cin <= 5 && H >= 1 && W >= 1, MRX_EUNSUP, "mrx_tl_wgrad_in: Cin=%d (1 .. 5: four column blocks)", Cin);
    constexpr int PH = WI_TH + 4, PW = 32 + 4 + 4;
    const int lds = 64 * WI_DYS + (Cin + 1) * PH * PW * 2 + 64;
    const int tiles_x = mrx_cdiv(W, 32), ntiles = tiles_x * mrx_cdiv(H, WI_TH), nwg = wi_nwg((long long)B * ntiles);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_tl_wgrad_in<5>, dim3(nwg), dim3(WI_NT), lds, st, x, (const unsigned*)dy_pairs, work, B, Cin, H, W, tiles_x, ntiles);
    MRX_LAUNCH_CHECK();
    const long long total = 64ll * Cin * 25;
    hipLaunchKernelGGL(k_tl_part_reduce, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, st, (const float*)work, nwg, total, dw, accumulate);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
