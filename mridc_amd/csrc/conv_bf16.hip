// conv_bf16.hip -- convolutions with bf16 operands and fp32 accumulation on v_mfma_f32_32x32x16_bf16: the mixed-precision mode of the
// training path (BASELINE config 4, "CIRIM bf16 training"; the reference trains under pytorch-lightning AMP, `precision: 16`,
// base_cirim_train.yaml:180 / ptl_overrides.py:10-15: autocast runs every convolution on half-precision operands with fp32
// accumulation and leaves FFTs, data consistency and the eta accumulation in fp32).
//
// Tensors stay fp32 NCHW in HBM (the same buffers, autograd tape and fp32 kernels on either side); the tile loader rounds the
// activations to bf16 (round-to-nearest-even, v_cvt_pk_bf16_f32) on their way into LDS and the weights are packed to bf16 once per
// weight version.  One kernel covers the forward convolutions, the IndRNN 1x1 `ih` GEMM with its epilogue and -- with flipped,
// transposed weights -- every data gradient:
//   * GEMM roles  D[32 couts][32 pixels of a row] += A[cout][k] B[k][pixel],  k = (tap, channel);  one MFMA consumes 16 k-values: the
//     lower half-wave eight channels of one (tap, channel-group), the upper half-wave the next group (the next eight channels of the same
//     tap for Cin >= 16, the next tap for Cin <= 8);
//   * input tile in LDS as [pixel][channel] bf16 (pixel stride 144 B for 64 channels: every 16-lane group of a ds_read_b128 covers all
//     64 banks exactly once), so a B operand is ONE 16-byte LDS read per lane;
//   * A operands (packed [step][cout block][lane][8 bf16], 1 KiB per wave instruction) come straight from L2 into registers: at 16x the
//     fp32 matrix rate the kernel is bound by HBM and by operand delivery, not by the matrix pipe, so LDS is spent on pixels, not weights.
//
// Round 4 (bf16 STORAGE of what autocast keeps in half precision, train_bf16.hip): the same kernel reads / writes "pair" tensors
// u32 [B][C / 2][H][W] = (bf16 channel 2p, bf16 channel 2p + 1) -- XP: the input is a pair tensor (no conversion in the loader), OUT 1: the
// result is rounded to bf16 and stored as pairs (data gradients), OUT 2: the whole training-mode RIM layer (conv + bias -> bf16 -> ReLU = a,
// stored as pairs and fed from registers into the IndRNN 1x1 GEMM -> bf16 -> + hh * h_prev -> ReLU = h, optionally the tap products of the
// final convolution): the rounding points of torch.autocast (conv results are bf16 tensors, hidden states fp32).
#include <type_traits>

#include "mrx_common.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CB_NT 512
#define CB_TH 8
#define CB_TW 32

__device__ __forceinline__ unsigned cb_pk(float lo, float hi) {   // two fp32 -> packed bf16 pair, round to nearest even
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

__device__ __forceinline__ float cb_lo(unsigned p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float cb_hi(unsigned p) { return __builtin_bit_cast(float, p & 0xffff0000u); }
__device__ __forceinline__ float cb_round(float v) { return cb_lo(cb_pk(v, 0.f)); }   // fp32 -> nearest bf16, as fp32

struct ConvBfArgs {
    const float* x;        // [B,Cin,H,W]  (XP: u32 pairs [B,Cin/2,H,W])
    const u32x4* packed;   // [NSTEP][NCT][64 lanes] x 8 bf16
    const float* bias;     // [Cout] or null
    const float* hh;       // [Cout] or null: IndRNN epilogue  act(acc + bias + hh * hprev)
    const float* hprev;    // [B,Cout,H,W] or null  (OUT 2: channel-blocked [B,8,H,W,8], like `out`)
    unsigned* hmask;       // OUT 2: [B,H,W,2] or null -- (h > 0) as bits (word = lane half, bit 16 c2 + r = accumulator row r of block c2)
    float* out;            // [B,Cout,H,W]
    int B, Cin, Cout, H, W, tiles_x, ntiles, pad_mode, act;
    float slope;
    int ext, Hin, Win;     // the input is [Hin, Win] = [H - 2 ext, W - 2 ext], read as if zero-extended by `ext` on every side (data gradients)
    float* interior;       // not null: outputs inside the original image go to interior [B,Cout,Hin,Win] (the replicate-padding fold leaves
                           // them unchanged), only the frame of width ext goes to `out` [B,Cout,H,W] for mrx_reppad_fold_edges
                           // (OUT 1: `interior` / without it `out` is a pair tensor; the frame stays fp32)
    int round_out;            // OUT 0: fp32 result rounded to the nearest bf16 value (the gradient w.r.t. an fp32 tensor that autocast cast to bf16)
    // OUT 2 (training-mode RIM layer):
    const u32x4* ih_packed;   // [4 steps][2 cout blocks][64 lanes] x 8 bf16: W_ih in the k order of the accumulator layout (mrx_tl_pack)
    const float* ih_bias;     // [64] or null
    unsigned* a_pairs;        // [B,32,H,W] pairs: a = ReLU(bf16(conv + bias))
    const u32x4* fin_packed;  // null or [4 steps][64 lanes] x 8 bf16: the final convolution's (tap, cout) rows
    float* taps;              // [B,18,H,W]: taps[tap * 2 + co] = sum_c bf16(w_final[co][c][tap]) bf16(h[c])   (mrx_tl_final_gather adds them up)
};

__host__ __device__ constexpr int cb_ps(int CPAD) { return CPAD == 8 ? 16 : CPAD * 2 + 16; }   // bytes per pixel in the LDS tile
__host__ __device__ constexpr int cb_nstep(int K, int CPAD) { return (K * K * (CPAD / 8) + 1) / 2; }

template <int K, int DIL, int CPAD, int NCT, int XP = 0, int OUT = 0>
__global__ __launch_bounds__(CB_NT, OUT == 2 ? 2 : 4) void k_conv_bf16(ConvBfArgs a) {
    static_assert(OUT != 2 || NCT == 2, "the fused layer has 64 features");
    constexpr int PAD = DIL * (K - 1) / 2;
    constexpr int PH = CB_TH + 2 * PAD, PW = CB_TW + 2 * PAD, NPIX = PH * PW;
    constexpr int NC8 = CPAD / 8, PS = cb_ps(CPAD), NG = K * K * NC8, NSTEP = cb_nstep(K, CPAD);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int tile = (int)mrx_xcd_band(blockIdx.x, a.ntiles);
    const int ty0 = tile / a.tiles_x;
    const int h0 = ty0 * CB_TH, w0 = (tile - ty0 * a.tiles_x) * CB_TW;
    const int b = blockIdx.y;
    const long long plane = (long long)a.H * a.W, iplane = (long long)a.Hin * a.Win;
    const float* xb = a.x + (long long)b * a.Cin * iplane;

    // ---- stage the halo'd tile: 8 channels of one pixel per item, fp32 -> bf16, one 16-byte LDS write ----------------------------------
    constexpr int ITEMS = NPIX * NC8, ITERS = (ITEMS + CB_NT - 1) / CB_NT;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int i = tid + it * CB_NT;
        if (i < ITEMS) {
            const int cg = i / NPIX, e = i - cg * NPIX;
            const int ty = e / PW, tx = e - ty * PW;
            int gy = h0 + ty - PAD - a.ext, gx = w0 + tx - PAD - a.ext;
            bool inb = true;
            if (a.pad_mode == MRX_PAD_REPLICATE) {
                gy = gy < 0 ? 0 : (gy >= a.Hin ? a.Hin - 1 : gy);
                gx = gx < 0 ? 0 : (gx >= a.Win ? a.Win - 1 : gx);
            } else {
                inb = gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
                gy = inb ? gy : 0;
                gx = inb ? gx : 0;
            }
            u32x4 p;
            if (XP == 2) {       // channel-blocked fp32 [B][Cin/8][H][W][8] (the training tape's hidden states): a pixel's eight channels are 32 contiguous bytes
                const float4* src = reinterpret_cast<const float4*>(a.x + (((long long)b * (a.Cin >> 3) + cg) * iplane + (long long)gy * a.Win + gx) * 8);
                float4 u0 = make_float4(0.f, 0.f, 0.f, 0.f), u1 = u0;
                if (inb) u0 = src[0], u1 = src[1];
                p = (u32x4){cb_pk(u0.x, u0.y), cb_pk(u0.z, u0.w), cb_pk(u1.x, u1.y), cb_pk(u1.z, u1.w)};
            } else if (XP) {     // pair tensor: four dwords = eight channels of this pixel, already bf16
                const unsigned* src = reinterpret_cast<const unsigned*>(a.x) + ((long long)b * (a.Cin >> 1) + cg * 4) * iplane + (long long)gy * a.Win + gx;
#pragma unroll
                for (int q = 0; q < 4; ++q) p[q] = (inb && cg * 8 + 2 * q < a.Cin) ? src[(long long)q * iplane] : 0u;
            } else {
                const float* src = xb + (long long)gy * a.Win + gx;
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = cg * 8 + j;
                    v[j] = (inb && c < a.Cin) ? src[(long long)c * iplane] : 0.f;
                }
                p = (u32x4){cb_pk(v[0], v[1]), cb_pk(v[2], v[3]), cb_pk(v[4], v[5]), cb_pk(v[6], v[7])};
            }
            *reinterpret_cast<u32x4*>(smem_b + e * PS + cg * 16) = p;
        }
    }
    // OUT 2: the layer's bias / hh tables into LDS (rounded where autocast rounds them) and this lane's 32 h_prev values requested NOW -- they
    // arrive under the matrix loop (requested in the epilogue they cost the launch 55 us: two workgroups per CU cannot cover that latency)
    float* tb = reinterpret_cast<float*>(smem_b + (size_t)NPIX * PS);
    float hpv[OUT == 2 ? 2 : 1][OUT == 2 ? 16 : 1];
    if (OUT == 2) {
        if (tid < 64) {
            tb[tid] = a.bias ? cb_round(a.bias[tid]) : 0.f;
            tb[64 + tid] = a.ih_bias ? cb_round(a.ih_bias[tid]) : 0.f;
            tb[128 + tid] = a.hprev ? a.hh[tid] : 0.f;
        }
        // hidden states are channel-blocked [B][8][H][W][8]: rows 4 k .. 4 k + 3 of a lane's accumulator are four consecutive channels of block
        // 4 c2 + k -> one 16-byte access per block (rim_layer2_sb.hip's CB8 form)
        const int py = h0 + wave < a.H ? h0 + wave : a.H - 1, px = w0 + l31 < a.W ? w0 + l31 : a.W - 1;
        const unsigned plane32 = (unsigned)plane * 32u, o0 = (unsigned)b * 8u * plane32 + (unsigned)(py * a.W + px) * 32u + 16u * lhi;
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float4 u = make_float4(0.f, 0.f, 0.f, 0.f);
                if (a.hprev) u = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(a.hprev) + o0 + (unsigned)(c2 * 4 + k) * plane32);
                hpv[c2][4 * k] = u.x, hpv[c2][4 * k + 1] = u.y, hpv[c2][4 * k + 2] = u.z, hpv[c2][4 * k + 3] = u.w;
            }
    }
    __syncthreads();

    // ---- the matrix loop: one 16-byte LDS read (B) and NCT 16-byte L2 reads (A) per MFMA step ---------------------------------------------
    f32x16 acc[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
    const unsigned char* bx = smem_b + (wave * PW + l31) * PS + (NC8 >= 2 ? lhi * 16 : 0);
    const u32x4* wp = a.packed + lane;
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        int off;
        if (NC8 >= 2) {                       // both halves: the same tap, neighbouring channel groups (the +16 B is in bx)
            constexpr int dummy = 0;
            (void)dummy;
            const int G0 = 2 * s, tap = G0 / NC8, cg0 = G0 % NC8;
            off = ((tap / K) * DIL * PW + (tap % K) * DIL) * PS + cg0 * 16;
        } else {                              // one channel group per tap: the upper half-wave takes the next tap
            const int t0 = 2 * s, t1 = (2 * s + 1 < NG) ? 2 * s + 1 : NG - 1;
            const int o0 = ((t0 / K) * DIL * PW + (t0 % K) * DIL) * PS, o1 = ((t1 / K) * DIL * PW + (t1 % K) * DIL) * PS;
            off = lhi ? o1 : o0;
        }
        const bf16x8 bv = *reinterpret_cast<const bf16x8*>(bx + off);
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
            const u32x4 aw = wp[(s * NCT + ct) * 64];
            acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aw), bv, acc[ct], 0, 0, 0);
        }
    }

    // ---- epilogue: bias, optional IndRNN term, activation ---------------------------------------------------------------------------------
    const int oy = h0 + wave, ox = w0 + l31;
    const bool inside = oy < a.H && ox < a.W;
    if (OUT == 2) {
        // a = ReLU(bf16(conv + bias)): stored as pairs, and -- two adjacent accumulator rows are two adjacent channels -- at the same time the
        // B operands of the 1x1 GEMM (step (ct, hf) takes rows 8 hf .. 8 hf + 7 of block ct: the pack orders W_ih's columns to match)
        const long long pix = (long long)oy * a.W + ox;
        unsigned ap[NCT][8];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int r = 2 * q, co = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                float v0 = acc[ct][r] + tb[co], v1 = acc[ct][r + 1] + tb[co + 1];
                v0 = v0 > 0.f ? v0 : 0.f, v1 = v1 > 0.f ? v1 : 0.f;
                ap[ct][q] = cb_pk(v0, v1);
                if (inside) a.a_pairs[((long long)b * 32 + (co >> 1)) * plane + pix] = ap[ct][q];
            }
        f32x16 acc2[2];
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[c2][r] = 0.f;
        const u32x4* ip = a.ih_packed + lane;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const u32x4 bw = {ap[ct][4 * hf], ap[ct][4 * hf + 1], ap[ct][4 * hf + 2], ap[ct][4 * hf + 3]};
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2)
                    acc2[c2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ip[((ct * 2 + hf) * 2 + c2) * 64]),
                                                                      __builtin_bit_cast(bf16x8, bw), acc2[c2], 0, 0, 0);
            }
        // h = ReLU(bf16(W_ih a + b_ih) + hh * h_prev)  (fp32 state)
        unsigned hp[2][8];
#pragma unroll
        for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = c2 * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                float v = cb_round(acc2[c2][r] + tb[64 + co]);
                v += tb[128 + co] * hpv[c2][r];
                acc2[c2][r] = v > 0.f ? v : 0.f;
            }
        if (inside && a.hmask) {      // (h > 0) of this lane's 32 channels as one word: all the cell's backward needs of h (train_bf16.hip: CellBwdArgs::hmask)
            unsigned mk = 0u;
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int r = 0; r < 16; ++r) mk |= (acc2[c2][r] > 0.f ? 1u : 0u) << (16 * c2 + r);
            a.hmask[((long long)b * plane + pix) * 2 + lhi] = mk;
        }
        if (inside) {
            float* ob = a.out + ((long long)b * 8 * plane + pix) * 8 + 4 * lhi;
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    *reinterpret_cast<float4*>(ob + (long long)(c2 * 4 + k) * plane * 8) = make_float4(acc2[c2][4 * k], acc2[c2][4 * k + 1], acc2[c2][4 * k + 2], acc2[c2][4 * k + 3]);
        }
        if (a.fin_packed) {                      // (tap, cout) rows of the final convolution times bf16(h): 18 of 32 output rows are real
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int q = 0; q < 8; ++q) hp[c2][q] = cb_pk(acc2[c2][2 * q], acc2[c2][2 * q + 1]);
            f32x16 acc3;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc3[r] = 0.f;
            const u32x4* fp = a.fin_packed + lane;
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const u32x4 bw = {hp[c2][4 * hf], hp[c2][4 * hf + 1], hp[c2][4 * hf + 2], hp[c2][4 * hf + 3]};
                    acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fp[(c2 * 2 + hf) * 64]), __builtin_bit_cast(bf16x8, bw), acc3, 0,
                                                                  0, 0);
                }
            if (inside) {
#pragma unroll
                for (int r = 0; r < 10; ++r) {
                    const int m = (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    if (m < 18) a.taps[((long long)b * 18 + m) * plane + pix] = acc3[r];
                }
            }
        }
        return;
    }
    if (inside) {
        long long obase = (long long)b * a.Cout * plane + (long long)oy * a.W + ox;
        long long cstride = plane;
        float* dst = a.out;
        bool to_interior = false, edge = false;
        const long long fbase = obase;            // this pixel in the frame tensor
        if (a.interior) {
            const int iy = oy - a.ext, ix = ox - a.ext;
            if (iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win) {
                dst = a.interior;
                cstride = iplane;
                obase = (long long)b * a.Cout * iplane + (long long)iy * a.Win + ix;
                to_interior = true;
                // bf16 results: an edge pixel of the image also receives the frame positions that clamp to it (mrx_tl_fold_edges); it leaves its
                // UNROUNDED value in the frame tensor too, so that the folded sum is rounded once, like every other element of the gradient
                edge = (OUT == 1 || a.round_out) && (iy == 0 || iy == a.Hin - 1 || ix == 0 || ix == a.Win - 1);
            }
        }
        if (OUT == 1 && (to_interior || !a.interior)) {      // bf16 result as pairs (the frame of a replicate-padded data gradient stays fp32)
            unsigned* dp = reinterpret_cast<unsigned*>(dst) + (long long)b * (a.Cout >> 1) * cstride + (obase - (long long)b * a.Cout * cstride);
#pragma unroll
            for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int r = 2 * q, co = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                    if (co < a.Cout) {
                        float v0 = acc[ct][r], v1 = acc[ct][r + 1];
                        if (a.bias) v0 += cb_round(a.bias[co]), v1 += cb_round(a.bias[co + 1]);
                        if (a.act == MRX_ACT_RELU) v0 = v0 > 0.f ? v0 : 0.f, v1 = v1 > 0.f ? v1 : 0.f;
                        dp[(long long)(co >> 1) * cstride] = cb_pk(v0, v1);
                        if (edge) a.out[fbase + (long long)co * plane] = v0, a.out[fbase + (long long)(co + 1) * plane] = v1;
                    }
                }
            return;
        }
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (co < a.Cout) {
                    float v = acc[ct][r];
                    if (a.bias) v += a.bias[co];
                    if (a.hprev) v += a.hh[co] * a.hprev[obase + (long long)co * plane];
                    if (a.act == MRX_ACT_RELU)
                        v = v > 0.f ? v : 0.f;
                    else if (a.act == MRX_ACT_LEAKY)
                        v = v > 0.f ? v : v * a.slope;
                    if (edge) a.out[fbase + (long long)co * plane] = v;
                    if ((OUT == 1 || a.round_out) && (to_interior || !a.interior)) v = cb_round(v);   // (frame positions stay fp32: they are summed, then rounded once)
                    dst[obase + (long long)co * cstride] = v;
                }
            }
    }
}

__device__ __forceinline__ int wb_tile(int j, int total, unsigned grid);
// ---- the tape's 3x3 dilation-2 64 -> 64 data gradient with the WEIGHTS IN LDS (mrx_tl_dgrad, pairs -> pairs) ----------------------------------------
// k_conv_bf16 gives a wave one image row and streams every weight fragment from L2 for a single MFMA: 576 KB of L2 reads per 8 x 32 tile, 536 MB per
// launch for 8 us of matrix work -- 48 us.  Here the 72 KB of packed weights are copied into LDS once per (persistent) workgroup and every step reads its
// A fragment from there (the inference kernel's arrangement); the arithmetic -- steps, operands, accumulation order -- is k_conv_bf16<3, 2, 64, 2, 1, 1>'s, so
// the results are bit-identical to it.  LDS: 73 728 B of weights + 62 208 B of tile = 133 KB, one workgroup of eight waves per CU.
// THIN (the 5x5 64 -> Cout <= 4 gradient of the first layer, k_conv_bf16<5, 1, 64, 1, 1, 0>'s arithmetic): only four of a fragment's 32 rows are real -- the
// table keeps those ([step][half][4 rows]: 12.8 KB instead of 100 KB, two workgroups per CU), the other lanes feed zeros; results fp32 rounded to bf16 values.
#define DG_NT 512
template <int K, int DIL, bool THIN>
__global__ __launch_bounds__(DG_NT, THIN ? 2 : 1) void k_tl_dgrad64(ConvBfArgs a) {
    constexpr int CPAD = 64, NCT = THIN ? 1 : 2;
    constexpr int PAD = DIL * (K - 1) / 2, PH = CB_TH + 2 * PAD, PW = CB_TW + 2 * PAD, NPIX = PH * PW;
    constexpr int NC8 = CPAD / 8, PS = cb_ps(CPAD), NSTEP = cb_nstep(K, CPAD), NWORDS = THIN ? NSTEP * 8 : NSTEP * NCT * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    u32x4* Wl = reinterpret_cast<u32x4*>(smem_b);                         // [NSTEP][NCT][64 lanes]   (THIN: [NSTEP][2 halves][4 rows])
    unsigned char* Xl = smem_b + (size_t)NWORDS * 16;                     // [NPIX][PS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    if (THIN) {
        for (int i = tid; i < NWORDS; i += DG_NT) Wl[i] = a.packed[(i >> 3) * 64 + ((i >> 2) & 1) * 32 + (i & 3)];
    } else {
        for (int i = tid; i < NWORDS; i += DG_NT) Wl[i] = a.packed[i];
    }
    const long long plane = (long long)a.H * a.W, iplane = (long long)a.Hin * a.Win;
    const int total = a.ntiles * a.B;
    constexpr int ITEMS = NPIX * NC8, ITERS = (ITEMS + DG_NT - 1) / DG_NT;
    // The tile of round r + 1 is requested (all loads of the thread, clamped coordinates, masked at the LDS write) right before the matrix loop of round r and
    // committed after it.  The loop starts one round early -- that round only issues -- so that there is one issue site (k_conv_wgrad_bf16's lesson: two
    // copies of it made hipcc wait for the loads at the head of the matrix block).
    unsigned xr[ITERS][4];
    unsigned inm = 0u;
    auto dg_tid = [&]() { int v = tid; asm volatile("" : "+v"(v)); return v; };      // (item coordinates recomputed per use: hoisted as loop invariants they spill)
    auto issue = [&](int t_) {
        const int tidv = dg_tid();
        const int tc = wb_tile(t_, total, gridDim.x);
        const int b = tc / a.ntiles, tile = tc - b * a.ntiles;
        const int ty0 = tile / a.tiles_x, h0 = ty0 * CB_TH, w0 = (tile - ty0 * a.tiles_x) * CB_TW;
        inm = 0u;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int i = min(tidv + it * DG_NT, ITEMS - 1);
            const int cg = i / NPIX, e = i - cg * NPIX;
            const int ty = e / PW, tx = e - ty * PW;
            int gy = h0 + ty - PAD - a.ext, gx = w0 + tx - PAD - a.ext;
            const bool inb = gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
            gy = gy < 0 ? 0 : (gy >= a.Hin ? a.Hin - 1 : gy);
            gx = gx < 0 ? 0 : (gx >= a.Win ? a.Win - 1 : gx);
            const unsigned* src = reinterpret_cast<const unsigned*>(a.x) + ((long long)b * 32 + cg * 4) * iplane + (long long)gy * a.Win + gx;
#pragma unroll
            for (int q = 0; q < 4; ++q) xr[it][q] = src[(long long)q * iplane];
            inm |= (inb ? 1u : 0u) << it;
        }
    };
    auto commit = [&]() {
        const int tidv = dg_tid();
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int i = tidv + it * DG_NT;
            if (i < ITEMS) {
                const int cg = i / NPIX, e = i - cg * NPIX;
                const bool inb = (inm >> it) & 1u;
                *reinterpret_cast<u32x4*>(Xl + e * PS + cg * 16) = (u32x4){inb ? xr[it][0] : 0u, inb ? xr[it][1] : 0u, inb ? xr[it][2] : 0u, inb ? xr[it][3] : 0u};
            }
        }
    };
    for (int t = (int)blockIdx.x - (int)gridDim.x; t < total; t += gridDim.x) {
        const bool cur = t >= 0;
        const int tc = wb_tile(cur ? t : 0, total, gridDim.x);
        const int b = tc / a.ntiles, tile = tc - b * a.ntiles;
        const int ty0 = tile / a.tiles_x, h0 = ty0 * CB_TH, w0 = (tile - ty0 * a.tiles_x) * CB_TW;
        __syncthreads();                 // the previous tile's readers are done (first round: the weights are in place)
        if (cur) {
            commit();
            __syncthreads();
        }
        {
            const int tn = t + (int)gridDim.x;
            issue(tn < total ? tn : total - 1);         // (past the end: a valid tile, unused)
        }
        if (!cur) continue;
        f32x16 acc[NCT];
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
        const unsigned char* bx = Xl + (wave * PW + l31) * PS + lhi * 16;
        const u32x4* wp = THIN ? Wl + lhi * 4 + (l31 & 3) : Wl + lane;
#pragma unroll
        for (int s_ = 0; s_ < NSTEP; ++s_) {
            const int G0 = 2 * s_, tap = G0 / NC8, cg0 = G0 % NC8;
            const int off = ((tap / K) * DIL * PW + (tap % K) * DIL) * PS + cg0 * 16;
            const bf16x8 bv = *reinterpret_cast<const bf16x8*>(bx + off);
            if (THIN) {
                u32x4 aw = wp[s_ * 8];
                if (l31 >= 4) aw = (u32x4){0u, 0u, 0u, 0u};
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, aw), bv, acc[0], 0, 0, 0);
            } else {
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
                    acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wp[(s_ * NCT + ct) * 64]), bv, acc[ct], 0, 0, 0);
            }
        }
        // ---- epilogue (k_conv_bf16's OUT 1 with `interior`): interior pixels -> dx as pairs, the frame -> fp32, edge pixels leave their unrounded value too
        const int oy = h0 + wave, ox = w0 + l31;
        if (THIN) {             // rows 0 .. 3 of the lower half-wave's accumulator = the output channels: fp32 holding bf16 values inside, unrounded in the frame
            if (oy < a.H && ox < a.W && lhi == 0) {
                const long long fbase = (long long)b * a.Cout * plane + (long long)oy * a.W + ox;
                const int iy = oy - a.ext, ix = ox - a.ext;
                const bool to_interior = iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
                const bool edge = to_interior && (iy == 0 || iy == a.Hin - 1 || ix == 0 || ix == a.Win - 1);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (r < a.Cout) {
                        const float v = acc[0][r];
                        if (edge || !to_interior) a.out[fbase + (long long)r * plane] = v;
                        if (to_interior) a.interior[((long long)b * a.Cout + r) * iplane + (long long)iy * a.Win + ix] = cb_round(v);
                    }
            }
        } else
        if (oy < a.H && ox < a.W) {
            const long long fbase = (long long)b * 64 * plane + (long long)oy * a.W + ox;
            const int iy = oy - a.ext, ix = ox - a.ext;
            const bool to_interior = iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
            const bool edge = to_interior && (iy == 0 || iy == a.Hin - 1 || ix == 0 || ix == a.Win - 1);
            if (to_interior) {
                unsigned* dp = reinterpret_cast<unsigned*>(a.interior) + (long long)b * 32 * iplane + (long long)iy * a.Win + ix;
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int r = 2 * q, co = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                        const float v0 = acc[ct][r], v1 = acc[ct][r + 1];
                        dp[(long long)(co >> 1) * iplane] = cb_pk(v0, v1);
                        if (edge) a.out[fbase + (long long)co * plane] = v0, a.out[fbase + (long long)(co + 1) * plane] = v1;
                    }
            } else {
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                        a.out[fbase + (long long)co * plane] = acc[ct][r];
                    }
            }
        }
    }
}
template <int K, int DIL, bool THIN>
static int dg64_launch(const ConvBfArgs& a, hipStream_t st) {
    constexpr int PAD = DIL * (K - 1) / 2;
    constexpr size_t lds = (size_t)cb_nstep(K, 64) * (THIN ? 8 : 2 * 64) * 16 + (size_t)(CB_TH + 2 * PAD) * (CB_TW + 2 * PAD) * cb_ps(64);
    static bool attr_done = false;
    static int n_cu = 0;
    if (!attr_done) {
        MRX_HIP(hipFuncSetAttribute((const void*)k_tl_dgrad64<K, DIL, THIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int dev = 0;
        hipDeviceProp_t prop;
        n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        attr_done = true;
    }
    const long long total = (long long)a.ntiles * a.B, cap = (long long)n_cu * (THIN ? 2 : 1);
    hipLaunchKernelGGL((k_tl_dgrad64<K, DIL, THIN>), dim3((unsigned)(total < cap ? total : cap)), dim3(DG_NT), lds, st, a);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// packed[(s * NCT + ct) * 64 + lane][j] = bf16(w_fwd[cout = 32 ct + lane % 32][channel = 8 (G % NC8) + j][tap = G / NC8]),  G = 2 s + lane / 32;
// transposed (data gradient): w_fwd[co][c][tap] = w[c][co][TAPS - 1 - tap] (flipped taps, in/out channels swapped)
__global__ void k_conv_bf16_pack(const float* __restrict__ w, u32x4* __restrict__ out, int Cin, int Cout, int K, int CPAD, int NCT,
                                 int transposed) {
    const int NC8 = CPAD / 8, TAPS = K * K, NG = TAPS * NC8, NSTEP = (NG + 1) / 2;
    const int total = NSTEP * NCT * 64;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int lane = i & 63, ct = (i >> 6) % NCT, s = (i >> 6) / NCT;
        const int G = 2 * s + (lane >> 5), co = ct * 32 + (lane & 31);
        const int tap = G / NC8, cg = G - tap * NC8;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = cg * 8 + j;
            v[j] = 0.f;
            if (G < NG && c < Cin && co < Cout)
                v[j] = transposed ? w[((long long)c * Cout + co) * TAPS + (TAPS - 1 - tap)] : w[((long long)co * Cin + c) * TAPS + tap];
        }
        out[i] = (u32x4){cb_pk(v[0], v[1]), cb_pk(v[2], v[3]), cb_pk(v[4], v[5]), cb_pk(v[6], v[7])};
    }
}

static int cb_cpad(int Cin) { return Cin <= 8 ? 8 : (Cin <= 64 ? 64 : -1); }
static bool cb_shape_ok(int Cin, int Cout, int K, int DIL) {
    if (cb_cpad(Cin) < 0 || Cout < 1 || Cout > 64) return false;
    const int cp = cb_cpad(Cin);
    if (K == 1 && DIL == 1) return cp == 64;
    if (K == 3 && DIL == 1) return true;
    if (K == 3 && DIL == 2) return cp == 64;
    if (K == 5 && DIL == 1) return true;
    return false;
}
extern "C" int mrx_conv_bf16_supported(int Cin, int Cout, int k, int dil) { return cb_shape_ok(Cin, Cout, k, dil) ? 1 : 0; }
extern "C" int64_t mrx_conv_bf16_pack_bytes(int Cin, int Cout, int k) {
    const int cp = cb_cpad(Cin);
    if (cp < 0 || Cout < 1 || Cout > 64 || k < 1) return -1;
    return (int64_t)cb_nstep(k, cp) * ((Cout + 31) / 32) * 64 * 16;
}
extern "C" int mrx_conv_bf16_pack(const float* w, void* packed, int Cin, int Cout, int k, int transposed, void* stream) {
    MRX_REQUIRE(w && packed, MRX_EINVAL, "mrx_conv_bf16_pack: null pointer");
    const int cp = cb_cpad(Cin);
    MRX_REQUIRE(cp > 0 && Cout >= 1 && Cout <= 64 && k >= 1 && (k & 1), MRX_EUNSUP, "mrx_conv_bf16_pack: Cin=%d Cout=%d k=%d", Cin, Cout, k);
    const int nct = (Cout + 31) / 32, total = cb_nstep(k, cp) * nct * 64;
    hipLaunchKernelGGL(k_conv_bf16_pack, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, (u32x4*)packed, Cin, Cout, k, cp,
                       nct, transposed);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

template <int K, int DIL, int CPAD, int NCT, int XP = 0, int OUT = 0>
static int cb_launch(const ConvBfArgs& a, hipStream_t st) {
    constexpr int PAD = DIL * (K - 1) / 2;
    constexpr size_t lds = (size_t)(CB_TH + 2 * PAD) * (CB_TW + 2 * PAD) * cb_ps(CPAD) + (OUT == 2 ? 768 : 0);
    static bool attr_done = false;
    if (lds > 48 * 1024 && !attr_done) {
        MRX_HIP(hipFuncSetAttribute((const void*)k_conv_bf16<K, DIL, CPAD, NCT, XP, OUT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    hipLaunchKernelGGL((k_conv_bf16<K, DIL, CPAD, NCT, XP, OUT>), dim3(a.ntiles, a.B), dim3(CB_NT), lds, st, a);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
template <int K, int DIL, int CPAD>
static int cb_launch_nct(const ConvBfArgs& a, hipStream_t st) {
    return a.Cout <= 32 ? cb_launch<K, DIL, CPAD, 1>(a, st) : cb_launch<K, DIL, CPAD, 2>(a, st);
}

// act(conv(x; w) + bias [+ hh * hprev]) with bf16 operands; 'same' size, stride 1; packed from mrx_conv_bf16_pack
static int conv2d_bf16_impl(const float* x, const void* packed, const float* bias, const float* hh, const float* hprev, float* out, int B,
                           int Cin, int Cout, int H, int W, int k, int dil, int pad_mode, int act, float slope, int ext, void* stream,
                           float* interior = nullptr);
extern "C" int mrx_conv2d_bf16(const float* x, const void* packed, const float* bias, const float* hh, const float* hprev, float* out, int B,
                               int Cin, int Cout, int H, int W, int k, int dil, int pad_mode, int act, float slope, void* stream) {
    return conv2d_bf16_impl(x, packed, bias, hh, hprev, out, B, Cin, Cout, H, W, k, dil, pad_mode, act, slope, 0, stream);
}
// The zero-padded convolution of x [B,Cin,H,W] read as if zero-extended by `ext` pixels on every side: out [B,Cout,H + 2 ext,W + 2 ext].
// With a transposed pack and ext = dilation (k - 1) / 2 this is the data gradient on the padded domain that mrx_reppad_fold folds back --
// without materialising the extended gradient.
extern "C" int mrx_conv2d_bf16_ext(const float* x, const void* packed, float* out, int B, int Cin, int Cout, int H, int W, int k, int dil,
                                   int ext, void* stream) {
    MRX_REQUIRE(ext >= 0, MRX_EINVAL, "mrx_conv2d_bf16_ext: bad extension %d", ext);
    return conv2d_bf16_impl(x, packed, nullptr, nullptr, nullptr, out, B, Cin, Cout, H + 2 * ext, W + 2 * ext, k, dil, MRX_PAD_ZERO, MRX_ACT_NONE,
                            0.f, ext, stream);
}
// Data gradient of a replicate-padded convolution in two launches: this one writes the gradient's interior straight into dx [B,Cout,H,W] and
// its frame of width ext into `frame` [B,Cout,H + 2 ext,W + 2 ext] (interior of `frame` untouched); mrx_reppad_fold_edges then adds the
// frame onto the edge pixels.  Replaces conv_ext + mrx_reppad_fold (which re-reads and re-writes the whole gradient).
extern "C" int mrx_conv2d_bf16_dgrad_rep(const float* dy, const void* packed, float* dx, float* frame, int B, int Cin, int Cout, int H, int W,
                                         int k, int dil, void* stream) {
    const int ext = dil * (k - 1) / 2;
    MRX_REQUIRE(dx && frame, MRX_EINVAL, "mrx_conv2d_bf16_dgrad_rep: null pointer");
    return conv2d_bf16_impl(dy, packed, nullptr, nullptr, nullptr, frame, B, Cin, Cout, H + 2 * ext, W + 2 * ext, k, dil, MRX_PAD_ZERO,
                            MRX_ACT_NONE, 0.f, ext, stream, dx);
}
static int conv2d_bf16_impl(const float* x, const void* packed, const float* bias, const float* hh, const float* hprev, float* out, int B,
                           int Cin, int Cout, int H, int W, int k, int dil, int pad_mode, int act, float slope, int ext, void* stream,
                           float* interior) {
    MRX_REQUIRE(x && packed && out, MRX_EINVAL, "mrx_conv2d_bf16: null pointer");
    MRX_REQUIRE(B >= 0 && H >= 1 && W >= 1 && H > 2 * ext && W > 2 * ext, MRX_EINVAL, "mrx_conv2d_bf16: bad dims");
    MRX_REQUIRE(cb_shape_ok(Cin, Cout, k, dil), MRX_EUNSUP, "mrx_conv2d_bf16: Cin=%d Cout=%d k=%d dilation=%d not instantiated", Cin, Cout, k, dil);
    MRX_REQUIRE(pad_mode == MRX_PAD_ZERO || pad_mode == MRX_PAD_REPLICATE, MRX_EINVAL, "mrx_conv2d_bf16: bad pad mode %d", pad_mode);
    MRX_REQUIRE(!hprev || hh, MRX_EINVAL, "mrx_conv2d_bf16: hprev without hh");
    MRX_REQUIRE(B <= 65535, MRX_EUNSUP, "mrx_conv2d_bf16: batch %d too large", B);
    if (B == 0) return MRX_OK;
    ConvBfArgs a = {};
    a.x = x, a.packed = (const u32x4*)packed, a.bias = bias, a.hh = hh, a.hprev = hprev, a.out = out;
    a.B = B, a.Cin = Cin, a.Cout = Cout, a.H = H, a.W = W;
    a.tiles_x = mrx_cdiv(W, CB_TW);
    a.ntiles = a.tiles_x * mrx_cdiv(H, CB_TH);
    a.pad_mode = pad_mode, a.act = act, a.slope = slope;
    a.ext = ext, a.Hin = H - 2 * ext, a.Win = W - 2 * ext, a.interior = interior;
    hipStream_t st = (hipStream_t)stream;
    const int cp = cb_cpad(Cin);
    if (k == 1) return cb_launch_nct<1, 1, 64>(a, st);
    if (k == 3 && dil == 2) return cb_launch_nct<3, 2, 64>(a, st);
    if (k == 3 && cp == 64) return cb_launch_nct<3, 1, 64>(a, st);
    if (k == 3) return cb_launch_nct<3, 1, 8>(a, st);
    if (cp == 64) return cb_launch_nct<5, 1, 64>(a, st);
    return cb_launch_nct<5, 1, 8>(a, st);
}

// ---- training-mode RIM layer and data gradients on pair tensors (round 4; callers: mridc_amd/training.py, bf16 tape) ---------------------
// W_ih [64,64] and the final convolution's weights [2,64,3,3] in the k order in which the accumulator layout of the preceding GEMM delivers the
// channels: step (ct, hf), lane half lh, element j  <->  channel 32 ct + (r & 3) + 8 (r >> 2) + 4 lh with r = 8 hf + j.
//   ih : out[((ct * 2 + hf) * 2 + c2) * 64 + lane][j] = bf16(w_ih[32 c2 + lane % 32][channel])
//   fin: out[512 + (ct * 2 + hf) * 64 + lane][j]      = bf16(w_fin[m & 1][channel][m >> 1]),  m = lane % 32 < 18 (else 0)
//   ihT: out[768 + (s * 2 + ct) * 64 + lane][j]       = bf16(w_ih[32 (lane / 32) + 8 s + j][32 ct + lane % 32])      (mrx_tl_cell_bwd's data gradient)
__global__ void k_tl_pack(const float* __restrict__ w_ih, const float* __restrict__ w_fin, u32x4* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 768 + 512) return;
    const int lane = i & 63, l31 = lane & 31, lh = lane >> 5;
    float v[8];
    if (i < 512) {
        const int c2 = (i >> 6) & 1, hf = (i >> 7) & 1, ct = i >> 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = 8 * hf + j, ch = 32 * ct + (r & 3) + 8 * (r >> 2) + 4 * lh;
            v[j] = w_ih[(32 * c2 + l31) * 64 + ch];
        }
    } else if (i < 768) {
        const int s = (i - 512) >> 6, hf = s & 1, ct = s >> 1, m = l31;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = 8 * hf + j, ch = 32 * ct + (r & 3) + 8 * (r >> 2) + 4 * lh;
            v[j] = (w_fin && m < 18) ? w_fin[((m & 1) * 64 + ch) * 9 + (m >> 1)] : 0.f;
        }
    } else {
        const int q = (i - 768) >> 6, ct = q & 1, s = q >> 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = w_ih[(32 * lh + 8 * s + j) * 64 + 32 * ct + l31];
    }
    out[i] = (u32x4){cb_pk(v[0], v[1]), cb_pk(v[2], v[3]), cb_pk(v[4], v[5]), cb_pk(v[6], v[7])};
}
extern "C" int64_t mrx_tl_pack_bytes(void) { return (768 + 512) * 16; }
extern "C" int mrx_tl_pack(const float* w_ih, const float* w_fin, void* packed, void* stream) {
    MRX_REQUIRE(w_ih && packed, MRX_EINVAL, "mrx_tl_pack: null pointer");
    hipLaunchKernelGGL(k_tl_pack, dim3(5), dim3(256), 0, (hipStream_t)stream, w_ih, w_fin, (u32x4*)packed);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// One RIM layer in training arithmetic (rim_block.py:230-238 under autocast): a = ReLU(bf16(conv_reppad(x) + b)) -> a_pairs [B,32,H,W],
// h = ReLU(bf16(W_ih a + b_ih) + hh * h_prev) -> h fp32, CHANNEL-BLOCKED [B,8,H,W,8] (c = 8 q + j) like h_prev and -- for the 64-channel layer -- x
// (the hidden states of the tape: 16-byte accesses, 1 KB contiguous per half-wave instead of 128-byte pieces of 64 planes); with `taps`: also the (tap, cout) products of the final 3x3 convolution with
// bf16(h) [B,18,H,W].  conv_packed from mrx_conv_bf16_pack (forward), tl_packed from mrx_tl_pack.  k x k = 5x5 (Cin <= 8) or 3x3 dilation 2 (Cin 64).
extern "C" int mrx_tl_layer_fwd(const float* x, const void* conv_packed, const float* conv_bias, const void* tl_packed, const float* ih_bias,
                                const float* hh, const float* hprev, void* a_pairs, float* h, void* hmask, float* taps, int B, int Cin, int H, int W,
                                int k, int dil, void* stream) {
    MRX_REQUIRE(x && conv_packed && tl_packed && a_pairs && h, MRX_EINVAL, "mrx_tl_layer_fwd: null pointer");
    MRX_REQUIRE(!hprev || hh, MRX_EINVAL, "mrx_tl_layer_fwd: hprev without hh");
    MRX_REQUIRE(B >= 1 && B <= 65535 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_tl_layer_fwd: bad dims");
    MRX_REQUIRE((k == 5 && dil == 1 && Cin >= 1 && Cin <= 8) || (k == 3 && dil == 2 && Cin == 64), MRX_EUNSUP,
                "mrx_tl_layer_fwd: Cin=%d k=%d dilation=%d not instantiated", Cin, k, dil);
    ConvBfArgs a = {};
    a.x = x, a.packed = (const u32x4*)conv_packed, a.bias = conv_bias, a.hh = hh, a.hprev = hprev, a.out = h;
    a.B = B, a.Cin = Cin, a.Cout = 64, a.H = H, a.W = W;
    a.tiles_x = mrx_cdiv(W, CB_TW), a.ntiles = a.tiles_x * mrx_cdiv(H, CB_TH);
    a.pad_mode = MRX_PAD_REPLICATE, a.act = MRX_ACT_RELU, a.Hin = H, a.Win = W;
    a.ih_packed = (const u32x4*)tl_packed, a.ih_bias = ih_bias, a.a_pairs = (unsigned*)a_pairs;
    a.fin_packed = taps ? (const u32x4*)tl_packed + 512 : nullptr, a.taps = taps, a.hmask = (unsigned*)hmask;
    return k == 5 ? cb_launch<5, 1, 8, 2, 0, 2>(a, (hipStream_t)stream) : cb_launch<3, 2, 64, 2, 2, 2>(a, (hipStream_t)stream);
}

// Data gradient of a replicate-padded convolution with bf16 results (what autocast's convolution backward returns): dy [B,Cdy,H,W] fp32
// (dy_pairs 0) or pairs [B,Cdy/2,H,W] (1); the interior goes to dx -- pairs [B,Cdx/2,H,W] (dx_pairs 1) or fp32 [B,Cdx,H,W] holding bf16 values --
// the frame of width ext = dil (k - 1) / 2 to `frame` [B,Cdx,H + 2 ext,W + 2 ext] fp32 for mrx_tl_fold_edges.  packed: mrx_conv_bf16_pack(transposed).
static int tl_dgrad_impl(const void* dy, int dy_pairs, const void* packed, void* dx, int dx_pairs, float* frame, int B, int Cdy, int Cdx, int H, int W, int k,
                         int dil, bool weights_in_lds, void* stream);
extern "C" int mrx_tl_dgrad(const void* dy, int dy_pairs, const void* packed, void* dx, int dx_pairs, float* frame, int B, int Cdy, int Cdx, int H,
                            int W, int k, int dil, void* stream) {
    return tl_dgrad_impl(dy, dy_pairs, packed, dx, dx_pairs, frame, B, Cdy, Cdx, H, W, k, dil, true, stream);
}
// the same through k_conv_bf16 for every shape (weights streamed from L2): the form mrx_tl_dgrad's 64 -> 64 kernel is pinned against, bit for bit
extern "C" int mrx_tl_dgrad_l2w(const void* dy, int dy_pairs, const void* packed, void* dx, int dx_pairs, float* frame, int B, int Cdy, int Cdx, int H,
                                int W, int k, int dil, void* stream) {
    return tl_dgrad_impl(dy, dy_pairs, packed, dx, dx_pairs, frame, B, Cdy, Cdx, H, W, k, dil, false, stream);
}
static int tl_dgrad_impl(const void* dy, int dy_pairs, const void* packed, void* dx, int dx_pairs, float* frame, int B, int Cdy, int Cdx, int H, int W, int k,
                         int dil, bool weights_in_lds, void* stream) {
    MRX_REQUIRE(dy && packed && dx && frame, MRX_EINVAL, "mrx_tl_dgrad: null pointer");
    MRX_REQUIRE(B >= 1 && B <= 65535 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_tl_dgrad: bad dims");
    const int ext = dil * (k - 1) / 2;
    ConvBfArgs a = {};
    a.x = (const float*)dy, a.packed = (const u32x4*)packed, a.out = frame, a.interior = (float*)dx;
    a.B = B, a.Cin = Cdy, a.Cout = Cdx, a.H = H + 2 * ext, a.W = W + 2 * ext;
    a.tiles_x = mrx_cdiv(a.W, CB_TW), a.ntiles = a.tiles_x * mrx_cdiv(a.H, CB_TH);
    a.pad_mode = MRX_PAD_ZERO, a.act = MRX_ACT_NONE, a.ext = ext, a.Hin = H, a.Win = W, a.round_out = 1;
    hipStream_t st = (hipStream_t)stream;
    if (k == 3 && dil == 2 && Cdy == 64 && Cdx == 64 && dy_pairs && dx_pairs)
        return weights_in_lds ? dg64_launch<3, 2, false>(a, st) : cb_launch<3, 2, 64, 2, 1, 1>(a, st);
    if (k == 3 && dil == 1 && Cdy <= 8 && Cdx == 64 && !dy_pairs && dx_pairs) return cb_launch<3, 1, 8, 2, 0, 1>(a, st);
    if (k == 5 && dil == 1 && Cdy == 64 && Cdx <= 32 && dy_pairs && !dx_pairs)
        return (weights_in_lds && Cdx <= 4) ? dg64_launch<5, 1, true>(a, st) : cb_launch<5, 1, 64, 1, 1, 0>(a, st);
    MRX_REQUIRE(false, MRX_EUNSUP, "mrx_tl_dgrad: Cdy=%d Cdx=%d k=%d dilation=%d pairs %d -> %d not instantiated", Cdy, Cdx, k, dil, dy_pairs, dx_pairs);
}

// ---- weight gradient, 64 -> 64 channels, bf16 operands ------------------------------------------------------------------------------
// dW[co][ci][tap] = sum over (b, pixel) dy[b,co,pixel] * xpad[b,ci,pixel + tap * dil]   as the GEMM  D[co][ci] += A[co][pixel] B[pixel][ci]
// per tap, contraction over pixels: one MFMA takes 16 consecutive pixels of a tile row (lower half-wave 8, upper half-wave the next 8).
//   * persistent workgroups (one per CU) walk the 8 x 32 pixel tiles; per tile the dy tile [64][8][32] and the halo'd x tile
//     [64][8 + 2 pad][32 + 2 pad] are rounded to bf16 on their way into LDS, pixel-fastest -- the natural NCHW order;
//   * 3x3: nine waves, wave t owns tap t and keeps its whole [64 x 64] block (4 accumulators) in registers over all tiles; the tap shift is
//     a constant offset of the B read.  Even dilations keep the shifted 16-byte reads 4-byte aligned (dword LDS reads); the rows of
//     both tiles are padded so that the 32 channel-lanes of a read fall on distinct banks;
//   * 1x1: four waves split the rows of the tile and add their blocks in LDS (fixed order) at the end;
//   * every workgroup leaves one partial [64][64][taps]; a second launch adds the partials in a fixed order in double, so the gradient
//     does not depend on scheduling (the same second stage as the fp32 kernel in conv_bwd.hip).
#define WB_TW 32
// tile rows: 8 for the 3x3 (halo 1.5x, one workgroup of nine waves per CU), 4 for the 1x1 (two 4-wave workgroups per CU cover each other's loads)
__host__ __device__ constexpr int wb_th(int K) { return K == 1 ? 4 : 8; }
__host__ __device__ constexpr int wb_dys(int K) { return wb_th(K) * WB_TW * 2 + 16; }   // bytes per dy channel in LDS (+16: = 4 mod 64 dwords)

struct WgradBfArgs {
    const float* x;    // [B,64,H,W]
    const float* dy;   // [B,64,H,W]
    float* part;       // [gridDim.x][64][64][taps]
    int B, H, W, tiles_x, tiles_y, ntiles, pad_mode;
    int vec;   // W % 4 == 0 and 16-byte aligned x / dy: vector tile loads
};

__host__ __device__ constexpr int wb_xs(int K, int DIL) {   // bytes per x channel in LDS, = 16 mod 256 (4 mod 64 dwords)
    const int pad = DIL * (K - 1) / 2, raw = (wb_th(K) + 2 * pad) * (WB_TW + 2 * pad) * 2;
    return ((raw + 255 - 16) / 256) * 256 + 16;
}

// eight dwords of a pair tensor (eight pixels of channels 2p, 2p + 1) -> the eight pixels of channel 2p, of channel 2p + 1
__device__ __forceinline__ void wb_unzip(const unsigned (&d)[8], u32x4& lo, u32x4& hi) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        lo[i] = (d[2 * i] & 0xffffu) | (d[2 * i + 1] << 16);
        hi[i] = (d[2 * i] >> 16) | (d[2 * i + 1] & 0xffff0000u);
    }
}
// PF (the training tape's form: dy a pair tensor, x channel-blocked, 3x3): the global loads of tile t + 1 are issued -- all of them, unconditionally, from
// clamped coordinates -- before the matrix phase of tile t and committed to LDS after it.  The first form walked its items in loops of dependent
// load -> convert -> LDS-write round trips (five per tile) with nothing else in flight: 25 us per tile, 6 of them matrix work.
// item j of the persistent loops (j = round * grid + workgroup) -> tile: the XCD band order when the grid is a multiple of the eight XCDs
__device__ __forceinline__ int wb_tile(int j, int total, unsigned grid) {
    return (grid & 7u) == 0u ? (int)mrx_xcd_band(j, total) : j;
}
template <int K, int DIL, int DYP = 0, int XCB = 0>
__global__ __launch_bounds__(K == 1 ? 256 : 576, (K == 3 && DYP && XCB) ? 1 : 2) void k_conv_wgrad_bf16(WgradBfArgs a) {
    constexpr int WB_TH = wb_th(K), WB_DYS = wb_dys(K);
    constexpr int PAD = DIL * (K - 1) / 2, PH = WB_TH + 2 * PAD, PW = WB_TW + 2 * PAD, TAPS = K * K;
    constexpr int NT = K == 1 ? 64 * WB_TH : 576, XS = wb_xs(K, DIL);
    static_assert((DIL & 1) == 0 || K == 1, "shifted reads must stay 4-byte aligned");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    unsigned char* Dy = smem_b;                 // [64][WB_DYS]
    unsigned char* Xs = smem_b + 64 * WB_DYS;   // [64][XS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const long long plane = (long long)a.H * a.W;
    const int tap = K == 1 ? 0 : wave, ky = tap / K, kx = tap % K;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int total_tiles = a.ntiles * a.B;
    constexpr bool PF = (K == 3 && DYP && XCB);
    // ---- PF: tile prefetch in registers ------------------------------------------------------------------------------------------------------
    constexpr int PF_XG2 = PW / 2, PF_NX = (8 * PH * PF_XG2 + NT - 1) / NT, PF_ND = (32 * WB_TH * 4 + NT - 1) / NT;
    float4 pfx[PF ? PF_NX : 1][4];
    uint4 pfd[PF ? PF_ND : 1][2];
    unsigned pfm = 0u;          // bits 2 k, 2 k + 1: the two pixels of x item k are inside (zero padding only); bits 8 + 8 k ..: the eight pixels of dy item k are inside
    static_assert(!PF || (2 * PF_NX <= 8 && 8 + 8 * PF_ND <= 32), "flag word");
    // (x -- three quarters of the bytes -- is prefetched across the matrix phase; dy is requested at the top of its own tile and committed after x: holding
    // it too spilled nine registers, and a spill inside the matrix loop waits on vmcnt, i.e. on the prefetch itself)
    constexpr int PF_NXP = 3;       // x items held across the matrix phase (the third one is requested with dy: all three spilled seven registers)
    // (item coordinates are recomputed from an opaque copy of the thread index at every use: hoisted out of the tile loop as invariants they were the seven
    // registers that spilled -- and a scratch access in the loop waits on vmcnt, i.e. on the prefetch)
    auto pf_tid = [&]() { int v = tid; asm volatile("" : "+v"(v)); return v; };
    auto pf_issue_x = [&](int t_, auto lo, auto hi) {
        constexpr int K0 = decltype(lo)::value, K1 = decltype(hi)::value;
        const int tidv = pf_tid();
        const int t = wb_tile(t_, total_tiles, gridDim.x);
        const int b = t / a.ntiles, tt = t - b * a.ntiles;
        const int ty0 = tt / a.tiles_x, h0 = ty0 * WB_TH, w0 = (tt - ty0 * a.tiles_x) * WB_TW;
#pragma unroll
        for (int k = K0; k < K1; ++k) {
            pfm &= ~(3u << (2 * k));
            const int i = min(tidv + k * NT, 8 * PH * PF_XG2 - 1);
            const int g2 = i % PF_XG2, r = (i / PF_XG2) % PH, q = i / (PF_XG2 * PH);
            int gy = h0 + r - PAD, gx0 = w0 + g2 * 2 - PAD, gx1 = gx0 + 1;
            const bool rowin = gy >= 0 && gy < a.H;
            gy = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy);
            const bool in0 = a.pad_mode == MRX_PAD_REPLICATE || (rowin && gx0 >= 0 && gx0 < a.W), in1 = a.pad_mode == MRX_PAD_REPLICATE || (rowin && gx1 >= 0 && gx1 < a.W);
            gx0 = gx0 < 0 ? 0 : (gx0 >= a.W ? a.W - 1 : gx0);
            gx1 = gx1 < 0 ? 0 : (gx1 >= a.W ? a.W - 1 : gx1);
            const float* bq = a.x + (((long long)b * 8 + q) * plane + (long long)gy * a.W) * 8;
            pfx[k][0] = *reinterpret_cast<const float4*>(bq + (long long)gx0 * 8), pfx[k][1] = *reinterpret_cast<const float4*>(bq + (long long)gx0 * 8 + 4);
            pfx[k][2] = *reinterpret_cast<const float4*>(bq + (long long)gx1 * 8), pfx[k][3] = *reinterpret_cast<const float4*>(bq + (long long)gx1 * 8 + 4);
            pfm |= (in0 ? 1u : 0u) << (2 * k) | (in1 ? 1u : 0u) << (2 * k + 1);
        }
    };
    auto pf_issue_dy = [&](int t_) {
        const int tidv = pf_tid();
        const int t = wb_tile(t_, total_tiles, gridDim.x);
        const int b = t / a.ntiles, tt = t - b * a.ntiles;
        const int ty0 = tt / a.tiles_x, h0 = ty0 * WB_TH, w0 = (tt - ty0 * a.tiles_x) * WB_TW;
        pfm &= 0xffu;
        const unsigned* dyp = reinterpret_cast<const unsigned*>(a.dy) + (long long)b * 32 * plane;
#pragma unroll
        for (int k = 0; k < PF_ND; ++k) {
            const int i = min(tidv + k * NT, 32 * WB_TH * 4 - 1);
            const int pg = i & 3, r = (i >> 2) % WB_TH, pp = i / (4 * WB_TH);
            const int gy = h0 + r, gx = w0 + pg * 8;
            const int gyc = gy < a.H ? gy : a.H - 1;
            const unsigned* row = dyp + (long long)pp * plane + (long long)gyc * a.W;
            if (a.vec && gx + 8 <= a.W) {
                pfd[k][0] = *reinterpret_cast<const uint4*>(row + gx), pfd[k][1] = *reinterpret_cast<const uint4*>(row + gx + 4);
                pfm |= (gy < a.H ? 0xffu : 0u) << (8 + 8 * k);
            } else {               // the ragged right edge (or an unaligned tensor): eight dwords from clamped columns, masked at commit
                unsigned d[8], mk = 0u;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int gxj = gx + j;
                    d[j] = row[gxj < a.W ? gxj : a.W - 1];
                    mk |= (gy < a.H && gxj < a.W ? 1u : 0u) << j;
                }
                pfd[k][0] = make_uint4(d[0], d[1], d[2], d[3]), pfd[k][1] = make_uint4(d[4], d[5], d[6], d[7]);
                pfm |= mk << (8 + 8 * k);
            }
        }
    };
    auto pf_commit_dy = [&]() {
        const int tidv = pf_tid();
#pragma unroll
        for (int k = 0; k < PF_ND; ++k) {
            const int i = tidv + k * NT;
            if (i < 32 * WB_TH * 4) {
                const int pg = i & 3, r = (i >> 2) % WB_TH, pp = i / (4 * WB_TH);
                const unsigned mk = (pfm >> (8 + 8 * k)) & 0xffu;
                unsigned d[8] = {pfd[k][0].x, pfd[k][0].y, pfd[k][0].z, pfd[k][0].w, pfd[k][1].x, pfd[k][1].y, pfd[k][1].z, pfd[k][1].w};
#pragma unroll
                for (int j = 0; j < 8; ++j) d[j] = (mk >> j) & 1u ? d[j] : 0u;
                u32x4 lo, hi;
                wb_unzip(d, lo, hi);
                *reinterpret_cast<u32x4*>(Dy + (2 * pp) * WB_DYS + (r * WB_TW + pg * 8) * 2) = lo;
                *reinterpret_cast<u32x4*>(Dy + (2 * pp + 1) * WB_DYS + (r * WB_TW + pg * 8) * 2) = hi;
            }
        }
    };
    auto pf_commit_x = [&](auto lo, auto hi) {
        constexpr int K0 = decltype(lo)::value, K1 = decltype(hi)::value;
        const int tidv = pf_tid();
#pragma unroll
        for (int k = K0; k < K1; ++k) {
            const int i = tidv + k * NT;
            if (i < 8 * PH * PF_XG2) {
                const int g2 = i % PF_XG2, r = (i / PF_XG2) % PH, q = i / (PF_XG2 * PH);
                const float m0 = (pfm >> (2 * k)) & 1u ? 1.f : 0.f, m1 = (pfm >> (2 * k + 1)) & 1u ? 1.f : 0.f;
                const float va[8] = {pfx[k][0].x, pfx[k][0].y, pfx[k][0].z, pfx[k][0].w, pfx[k][1].x, pfx[k][1].y, pfx[k][1].z, pfx[k][1].w};
                const float vb[8] = {pfx[k][2].x, pfx[k][2].y, pfx[k][2].z, pfx[k][2].w, pfx[k][3].x, pfx[k][3].y, pfx[k][3].z, pfx[k][3].w};
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    *reinterpret_cast<unsigned*>(Xs + (8 * q + j) * XS + (r * PW + g2 * 2) * 2) = cb_pk(m0 != 0.f ? va[j] : 0.f, m1 != 0.f ? vb[j] : 0.f);
            }
        }
    };
    using pf_i0 = std::integral_constant<int, 0>;
    using pf_ip = std::integral_constant<int, (PF_NXP < PF_NX ? PF_NXP : PF_NX)>;
    using pf_in = std::integral_constant<int, PF_NX>;
    // PF: the loop starts one round early -- that round only issues the first tile's x loads -- so that there is ONE issue site, directly in front of the
    // matrix phase (with a second copy ahead of the loop hipcc merged the two paths into a `s_waitcnt vmcnt(0)` at the head of the matrix block)
    for (int t = PF ? (int)blockIdx.x - (int)gridDim.x : (int)blockIdx.x; t < total_tiles; t += gridDim.x) {
        const bool cur = !PF || t >= 0;
        // every XCD walks one contiguous band of tiles (workgroup j runs on XCD j % 8 and the grid is a multiple of 8): the halo rows two vertically
        // adjacent tiles share then meet in ONE L2 instead of being fetched by two
        const int tc = wb_tile(t >= 0 ? t : 0, total_tiles, gridDim.x);
        const int b = tc / a.ntiles, tt = tc - b * a.ntiles;
        const int ty0 = tt / a.tiles_x, h0 = ty0 * WB_TH, w0 = (tt - ty0 * a.tiles_x) * WB_TW;
        const float* dyb = a.dy + (long long)b * 64 * plane;
        const float* xb = a.x + (long long)b * 64 * plane;
        if (cur) __syncthreads();   // the previous tile's readers are done
        if (PF) {
            if (cur) {
                pf_issue_x(t, pf_ip{}, pf_in{});
                pf_issue_dy(t);
#if !(defined(MRX_WB_ABL) && (MRX_WB_ABL & 2))
                pf_commit_x(pf_i0{}, pf_ip{});
                pf_commit_x(pf_ip{}, pf_in{});
                pf_commit_dy();
#endif
                __syncthreads();
            }
            {
                const int tn = t + (int)gridDim.x;
                pf_issue_x(tn < total_tiles ? tn : total_tiles - 1, pf_i0{}, pf_ip{});      // in flight under this tile's matrix phase (past the end: a valid tile, unused)
            }
            if (!cur) continue;
        } else {
        // dy tile: item = (co, row, 8-pixel group); pixels outside the image contribute zero
        if (DYP) {        // dy is a pair tensor [B,32,H,W]: item = (channel pair, row, 8-pixel group)
            const unsigned* dyp = reinterpret_cast<const unsigned*>(a.dy) + (long long)b * 32 * plane;
            for (int i = tid; i < 32 * WB_TH * 4; i += NT) {
                const int pg = i & 3, r = (i >> 2) % WB_TH, pp = i / (4 * WB_TH);
                const int gy = h0 + r, gx = w0 + pg * 8;
                const unsigned* src = dyp + (long long)pp * plane + (long long)gy * a.W + gx;
                unsigned d[8];
                if (a.vec && gy < a.H && gx + 8 <= a.W) {
                    const uint4 q0 = *reinterpret_cast<const uint4*>(src), q1 = *reinterpret_cast<const uint4*>(src + 4);
                    d[0] = q0.x, d[1] = q0.y, d[2] = q0.z, d[3] = q0.w, d[4] = q1.x, d[5] = q1.y, d[6] = q1.z, d[7] = q1.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) d[j] = (gy < a.H && gx + j < a.W) ? src[j] : 0u;
                }
                u32x4 lo, hi;
                wb_unzip(d, lo, hi);
                *reinterpret_cast<u32x4*>(Dy + (2 * pp) * WB_DYS + (r * WB_TW + pg * 8) * 2) = lo;
                *reinterpret_cast<u32x4*>(Dy + (2 * pp + 1) * WB_DYS + (r * WB_TW + pg * 8) * 2) = hi;
            }
        } else
        for (int i = tid; i < 64 * WB_TH * 4; i += NT) {
            const int pg = i & 3, r = (i >> 2) % WB_TH, co = i / (4 * WB_TH);
            const int gy = h0 + r, gx = w0 + pg * 8;
            float v[8];
            const float* src = dyb + (long long)co * plane + (long long)gy * a.W + gx;
            if (a.vec && gy < a.H && gx + 8 <= a.W) {
                const float4 q0 = *reinterpret_cast<const float4*>(src), q1 = *reinterpret_cast<const float4*>(src + 4);
                v[0] = q0.x, v[1] = q0.y, v[2] = q0.z, v[3] = q0.w, v[4] = q1.x, v[5] = q1.y, v[6] = q1.z, v[7] = q1.w;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (gy < a.H && gx + j < a.W) ? src[j] : 0.f;
            }
            *reinterpret_cast<u32x4*>(Dy + co * WB_DYS + (r * WB_TW + pg * 8) * 2) =
                (u32x4){cb_pk(v[0], v[1]), cb_pk(v[2], v[3]), cb_pk(v[4], v[5]), cb_pk(v[6], v[7])};
        }
        // x tile with halo: item = (ci, row, 4-pixel group)
        constexpr int XG = PW / 4;
        if (XCB) {        // x channel-blocked [B][8][H][W][8]: item = (block, row, pixel pair) -> eight channels x two pixels = four 16-byte loads, eight dword writes
            constexpr int XG2 = PW / 2;
            for (int i = tid; i < 8 * PH * XG2; i += NT) {
                const int g2 = i % XG2, r = (i / XG2) % PH, q = i / (XG2 * PH);
                int gy = h0 + r - PAD, gx0 = w0 + g2 * 2 - PAD, gx1 = gx0 + 1;
                const bool rowin = gy >= 0 && gy < a.H;
                gy = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy);
                const bool in0 = a.pad_mode == MRX_PAD_REPLICATE || (rowin && gx0 >= 0 && gx0 < a.W), in1 = a.pad_mode == MRX_PAD_REPLICATE || (rowin && gx1 >= 0 && gx1 < a.W);
                gx0 = gx0 < 0 ? 0 : (gx0 >= a.W ? a.W - 1 : gx0);
                gx1 = gx1 < 0 ? 0 : (gx1 >= a.W ? a.W - 1 : gx1);
                const float* bq = a.x + (((long long)b * 8 + q) * plane + (long long)gy * a.W) * 8;
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 a0 = in0 ? *reinterpret_cast<const float4*>(bq + (long long)gx0 * 8) : z, a1 = in0 ? *reinterpret_cast<const float4*>(bq + (long long)gx0 * 8 + 4) : z;
                const float4 b0 = in1 ? *reinterpret_cast<const float4*>(bq + (long long)gx1 * 8) : z, b1 = in1 ? *reinterpret_cast<const float4*>(bq + (long long)gx1 * 8 + 4) : z;
                const float va[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, vb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) *reinterpret_cast<unsigned*>(Xs + (8 * q + j) * XS + (r * PW + g2 * 2) * 2) = cb_pk(va[j], vb[j]);
            }
        } else
        for (int i = tid; i < 64 * PH * XG; i += NT) {
            const int g4 = i % XG, r = (i / XG) % PH, ci = i / (XG * PH);
            const int gy0 = h0 + r - PAD, gx0 = w0 + g4 * 4 - PAD;
            float v[4];
            int gyc = gy0 < 0 ? 0 : (gy0 >= a.H ? a.H - 1 : gy0);
            const bool rowin = gy0 >= 0 && gy0 < a.H;
            const float* rowp = xb + (long long)ci * plane + (long long)gyc * a.W;
            if (a.vec && (PAD & 1) == 0 && gx0 >= 0 && gx0 + 4 <= a.W && (rowin || a.pad_mode == MRX_PAD_REPLICATE)) {
                // interior: two 8-byte loads (w0 and the even padding keep the pair aligned)
                const float2 q0 = *reinterpret_cast<const float2*>(rowp + gx0), q1 = *reinterpret_cast<const float2*>(rowp + gx0 + 2);
                v[0] = q0.x, v[1] = q0.y, v[2] = q1.x, v[3] = q1.y;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int gx = gx0 + j;
                    bool inb = true;
                    if (a.pad_mode == MRX_PAD_REPLICATE) {
                        gx = gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx);
                    } else {
                        inb = rowin && gx >= 0 && gx < a.W;
                        gx = inb ? gx : 0;
                    }
                    v[j] = inb ? rowp[gx] : 0.f;
                }
            }
            *reinterpret_cast<uint2*>(Xs + ci * XS + (r * PW + g4 * 4) * 2) = make_uint2(cb_pk(v[0], v[1]), cb_pk(v[2], v[3]));
        }
        __syncthreads();
        }
        const unsigned char* ap = Dy + l31 * WB_DYS + lhi * 16;
        const unsigned char* bp = Xs + l31 * XS + ((ky * DIL) * PW + kx * DIL + lhi * 8) * 2;
#if defined(MRX_WB_ABL) && (MRX_WB_ABL & 1)
        if (PF) continue;
#endif
#pragma unroll
        for (int r = 0; r < WB_TH; ++r) {
            if (K == 1 && r != wave) continue;   // 1x1: the waves split the rows of the tile
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(ap + (r * WB_TW + kk * 16) * 2);
                const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(ap + 32 * WB_DYS + (r * WB_TW + kk * 16) * 2);
                const unsigned* q0 = reinterpret_cast<const unsigned*>(bp + (r * PW + kk * 16) * 2);
                const unsigned* q1 = reinterpret_cast<const unsigned*>(bp + 32 * XS + (r * PW + kk * 16) * 2);
                const u32x4 b0 = {q0[0], q0[1], q0[2], q0[3]}, b1 = {q1[0], q1[1], q1[2], q1[3]};
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, __builtin_bit_cast(bf16x8, b0), acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, __builtin_bit_cast(bf16x8, b1), acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, __builtin_bit_cast(bf16x8, b0), acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, __builtin_bit_cast(bf16x8, b1), acc[1][1], 0, 0, 0);
                if (PF) __builtin_amdgcn_sched_barrier(0);      // operand reads stay next to their step: hoisted across the unrolled loop they push the prefetch registers out
            }
        }
    }
    float* po = a.part + (long long)blockIdx.x * 64 * 64 * TAPS;
    if (K == 1) {
        // add the waves' blocks in LDS, halving the number of live blocks each round (fixed order), wave 0 stores
        __syncthreads();
        float* R = reinterpret_cast<float*>(smem_b);   // [WB_TH / 2 waves][4096]
        for (int half = WB_TH / 2; half >= 1; half >>= 1) {
            if (wave >= half && wave < 2 * half) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) R[(wave - half) * 4096 + ((i * 2 + j) * 16 + r) * 64 + lane] = acc[i][j][r];
            }
            __syncthreads();
            if (wave < half) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] += R[wave * 4096 + ((i * 2 + j) * 16 + r) * 64 + lane];
            }
            __syncthreads();
        }
        if (wave != 0) return;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lhi, ci = 32 * j + l31;
                po[((long long)co * 64 + ci) * TAPS + tap] = acc[i][j][r];
            }
}

// ---- the same contraction for the two thin layers of the RIM (dilation 1): 3x3 64 -> 2 (final conv) and 5x5 4 -> 64 (first conv) -----------
// Only Cout (resp. Cin) rows of dy (x) exist in LDS plus one row of zeros that the idle channel-lanes read; a wave owns TPW taps
// (nine waves x 1 tap, thirteen waves x 2 taps).  Dilation 1 makes every other tap shift odd: those B operands are five dword reads
// funnel-shifted by 16 bits (v_alignbit_b32) instead of four.
struct WgradBfGArgs {
    const float* x;    // [B,Cin,H,W]
    const float* dy;   // [B,Cout,H,W]
    float* part;       // [gridDim.x][Cout][Cin][taps]
    int B, Cin, Cout, H, W, tiles_x, ntiles, pad_mode, vec;
};
template <int K, int NCO, int NCI, int TPW, int DYP = 0, int XCB = 0>
__global__ __launch_bounds__(64 * ((K * K + TPW - 1) / TPW), 1) void k_conv_wgrad_bf16_g(WgradBfGArgs a) {
    constexpr int TH = 8, DYS = TH * WB_TW * 2 + 16, PAD = (K - 1) / 2, PH = TH + 2 * PAD, PW = WB_TW + 2 * PAD, TAPS = K * K;
    constexpr int NW = (TAPS + TPW - 1) / TPW, NT = 64 * NW, XS = ((PH * PW * 2 + 255 - 16) / 256) * 256 + 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b[];
    unsigned char* Dy = smem_b;                                // [Cout + 1][DYS], the last row zeros
    unsigned char* Xs = smem_b + (size_t)(a.Cout + 1) * DYS;   // [Cin + 1][XS], the last row zeros
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const long long plane = (long long)a.H * a.W;
    f32x16 acc[TPW][NCO][NCI];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int i = 0; i < NCO; ++i)
#pragma unroll
            for (int j = 0; j < NCI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][i][j][r] = 0.f;
    for (int i = tid; i < DYS / 4; i += NT) reinterpret_cast<unsigned*>(Dy + (size_t)a.Cout * DYS)[i] = 0u;
    for (int i = tid; i < XS / 4; i += NT) reinterpret_cast<unsigned*>(Xs + (size_t)a.Cin * XS)[i] = 0u;
    // per-lane operand rows: channel-lanes beyond the real channel count read the zero row
    int arow[NCO], brow[NCI];
#pragma unroll
    for (int i = 0; i < NCO; ++i) arow[i] = (32 * i + l31 < a.Cout ? 32 * i + l31 : a.Cout) * DYS + lhi * 16;
#pragma unroll
    for (int j = 0; j < NCI; ++j) brow[j] = (32 * j + l31 < a.Cin ? 32 * j + l31 : a.Cin) * XS;

    const int total_tiles = a.ntiles * a.B;
    for (int t_ = blockIdx.x; t_ < total_tiles; t_ += gridDim.x) {
        const int t = wb_tile(t_, total_tiles, gridDim.x);     // (the XCD band order of the other persistent loops: a tile's halo rows meet their neighbours' in one L2)
        const int b = t / a.ntiles, tt = t - b * a.ntiles;
        const int ty0 = tt / a.tiles_x, h0 = ty0 * TH, w0 = (tt - ty0 * a.tiles_x) * WB_TW;
        const float* dyb = a.dy + (long long)b * a.Cout * plane;
        const float* xb = a.x + (long long)b * a.Cin * plane;
        __syncthreads();
        if (DYP) {        // dy is a pair tensor [B,Cout/2,H,W]
            const unsigned* dyp = reinterpret_cast<const unsigned*>(a.dy) + (long long)b * (a.Cout >> 1) * plane;
            for (int i = tid; i < (a.Cout >> 1) * TH * 4; i += NT) {
                const int pg = i & 3, r = (i >> 2) % TH, pp = i / (4 * TH);
                const int gy = h0 + r, gx = w0 + pg * 8;
                const unsigned* src = dyp + (long long)pp * plane + (long long)gy * a.W + gx;
                unsigned d[8];
                if (a.vec && gy < a.H && gx + 8 <= a.W) {
                    const uint4 q0 = *reinterpret_cast<const uint4*>(src), q1 = *reinterpret_cast<const uint4*>(src + 4);
                    d[0] = q0.x, d[1] = q0.y, d[2] = q0.z, d[3] = q0.w, d[4] = q1.x, d[5] = q1.y, d[6] = q1.z, d[7] = q1.w;
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) d[j] = (gy < a.H && gx + j < a.W) ? src[j] : 0u;
                }
                u32x4 lo, hi;
                wb_unzip(d, lo, hi);
                *reinterpret_cast<u32x4*>(Dy + (2 * pp) * DYS + (r * WB_TW + pg * 8) * 2) = lo;
                *reinterpret_cast<u32x4*>(Dy + (2 * pp + 1) * DYS + (r * WB_TW + pg * 8) * 2) = hi;
            }
        } else
        for (int i = tid; i < a.Cout * TH * 4; i += NT) {
            const int pg = i & 3, r = (i >> 2) % TH, co = i / (4 * TH);
            const int gy = h0 + r, gx = w0 + pg * 8;
            float v[8];
            const float* src = dyb + (long long)co * plane + (long long)gy * a.W + gx;
            if (a.vec && gy < a.H && gx + 8 <= a.W) {
                const float4 q0 = *reinterpret_cast<const float4*>(src), q1 = *reinterpret_cast<const float4*>(src + 4);
                v[0] = q0.x, v[1] = q0.y, v[2] = q0.z, v[3] = q0.w, v[4] = q1.x, v[5] = q1.y, v[6] = q1.z, v[7] = q1.w;
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (gy < a.H && gx + j < a.W) ? src[j] : 0.f;
            }
            *reinterpret_cast<u32x4*>(Dy + co * DYS + (r * WB_TW + pg * 8) * 2) =
                (u32x4){cb_pk(v[0], v[1]), cb_pk(v[2], v[3]), cb_pk(v[4], v[5]), cb_pk(v[6], v[7])};
        }
        constexpr int XG = PW / 2;
        if (XCB) {        // x channel-blocked [B][8][H][W][8] (Cin = 64): item = (block, row, pixel pair).  All the loads of the tile first (clamped
                          // coordinates, no branch), then the conversions and LDS writes: the item loop with its load -> write round trip per pass took three
            constexpr int NXI = (8 * PH * XG + NT - 1) / NT;
            float4 xr[NXI][4];
            unsigned xin = 0u;
#pragma unroll
            for (int k = 0; k < NXI; ++k) {
                const int i = min(tid + k * NT, 8 * PH * XG - 1);
                const int g2 = i % XG, r = (i / XG) % PH, q = i / (XG * PH);
                int gy = h0 + r - PAD, gx0 = w0 + g2 * 2 - PAD, gx1 = gx0 + 1;
                const bool rowin = gy >= 0 && gy < a.H;
                gy = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy);
                const bool in0 = a.pad_mode == MRX_PAD_REPLICATE || (rowin && gx0 >= 0 && gx0 < a.W), in1 = a.pad_mode == MRX_PAD_REPLICATE || (rowin && gx1 >= 0 && gx1 < a.W);
                gx0 = gx0 < 0 ? 0 : (gx0 >= a.W ? a.W - 1 : gx0);
                gx1 = gx1 < 0 ? 0 : (gx1 >= a.W ? a.W - 1 : gx1);
                const float* bq = a.x + (((long long)b * 8 + q) * plane + (long long)gy * a.W) * 8;
                xr[k][0] = *reinterpret_cast<const float4*>(bq + (long long)gx0 * 8), xr[k][1] = *reinterpret_cast<const float4*>(bq + (long long)gx0 * 8 + 4);
                xr[k][2] = *reinterpret_cast<const float4*>(bq + (long long)gx1 * 8), xr[k][3] = *reinterpret_cast<const float4*>(bq + (long long)gx1 * 8 + 4);
                xin |= (in0 ? 1u : 0u) << (2 * k) | (in1 ? 1u : 0u) << (2 * k + 1);
            }
#pragma unroll
            for (int k = 0; k < NXI; ++k) {
                const int i = tid + k * NT;
                if (i < 8 * PH * XG) {
                    const int g2 = i % XG, r = (i / XG) % PH, q = i / (XG * PH);
                    const bool in0 = (xin >> (2 * k)) & 1u, in1 = (xin >> (2 * k + 1)) & 1u;
                    const float va[8] = {xr[k][0].x, xr[k][0].y, xr[k][0].z, xr[k][0].w, xr[k][1].x, xr[k][1].y, xr[k][1].z, xr[k][1].w};
                    const float vb[8] = {xr[k][2].x, xr[k][2].y, xr[k][2].z, xr[k][2].w, xr[k][3].x, xr[k][3].y, xr[k][3].z, xr[k][3].w};
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        *reinterpret_cast<unsigned*>(Xs + (8 * q + j) * XS + (r * PW + g2 * 2) * 2) = cb_pk(in0 ? va[j] : 0.f, in1 ? vb[j] : 0.f);
                }
            }
        } else
        for (int i = tid; i < a.Cin * PH * XG; i += NT) {
            const int g2 = i % XG, r = (i / XG) % PH, ci = i / (XG * PH);
            const int gy0 = h0 + r - PAD, gx0 = w0 + g2 * 2 - PAD;
            const bool rowin = gy0 >= 0 && gy0 < a.H;
            const int gyc = gy0 < 0 ? 0 : (gy0 >= a.H ? a.H - 1 : gy0);
            const float* rowp = xb + (long long)ci * plane + (long long)gyc * a.W;
            float v[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                int gx = gx0 + j;
                bool inb = true;
                if (a.pad_mode == MRX_PAD_REPLICATE) {
                    gx = gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx);
                } else {
                    inb = rowin && gx >= 0 && gx < a.W;
                    gx = inb ? gx : 0;
                }
                v[j] = inb ? rowp[gx] : 0.f;
            }
            *reinterpret_cast<unsigned*>(Xs + ci * XS + (r * PW + g2 * 2) * 2) = cb_pk(v[0], v[1]);
        }
        __syncthreads();
#ifdef MRX_WBG_ABL_NOMFMA                           // (timing variant: loads, conversions and LDS writes only)
        if (a.B > 0) continue;
#endif
#pragma unroll
        for (int tw = 0; tw < TPW; ++tw) {
            const int tap = wave * TPW + tw;
            if (tap >= TAPS) break;
            const int ky = tap / K, kx = tap - ky * K;
            const int boff = (ky * PW + (kx & ~1) + lhi * 8) * 2;      // even part of the shift; an odd kx adds one element below
#pragma unroll
            for (int r = 0; r < TH; ++r) {
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    bf16x8 av[NCO], bv[NCI];
#pragma unroll
                    for (int i = 0; i < NCO; ++i) av[i] = *reinterpret_cast<const bf16x8*>(Dy + arow[i] + (r * WB_TW + kk * 16) * 2);
#pragma unroll
                    for (int j = 0; j < NCI; ++j) {
                        const unsigned* q = reinterpret_cast<const unsigned*>(Xs + brow[j] + boff + (r * PW + kk * 16) * 2);
                        u32x4 w4;
                        if (kx & 1) {
                            const unsigned d0 = q[0], d1 = q[1], d2 = q[2], d3 = q[3], d4 = q[4];
                            w4 = (u32x4){__builtin_amdgcn_alignbit(d1, d0, 16), __builtin_amdgcn_alignbit(d2, d1, 16),
                                         __builtin_amdgcn_alignbit(d3, d2, 16), __builtin_amdgcn_alignbit(d4, d3, 16)};
                        } else {
                            w4 = (u32x4){q[0], q[1], q[2], q[3]};
                        }
                        bv[j] = __builtin_bit_cast(bf16x8, w4);
                    }
#pragma unroll
                    for (int i = 0; i < NCO; ++i)
#pragma unroll
                        for (int j = 0; j < NCI; ++j)
                            acc[tw][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[i], bv[j], acc[tw][i][j], 0, 0, 0);
                }
            }
        }
    }
    float* po = a.part + (long long)blockIdx.x * a.Cout * a.Cin * TAPS;
#pragma unroll
    for (int tw = 0; tw < TPW; ++tw) {
        const int tap = wave * TPW + tw;
        if (tap >= TAPS) break;
#pragma unroll
        for (int i = 0; i < NCO; ++i)
#pragma unroll
            for (int j = 0; j < NCI; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lhi, ci = 32 * j + l31;
                    if (co < a.Cout && ci < a.Cin) po[((long long)co * a.Cin + ci) * TAPS + tap] = acc[tw][i][j][r];
                }
    }
}

// dW[i] (= or +=) sum of the workgroup partials in a fixed order, in double: mrx_reduce_parts (mrx_common.h)
__global__ __launch_bounds__(256) void k_wgrad_bf16_reduce(const float* __restrict__ part, int nparts, long long n, float* __restrict__ dw,
                                                           int accumulate) {
    mrx_reduce_parts(part, nparts, n, dw, accumulate);
}

static int wb_nwg(int B, int H, int W, int k) {
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
            n_cu = prop.multiProcessorCount;
        else
            n_cu = 256;
    }
    const int per_cu = k == 1 ? 2 : 1;
    const long long tiles = (long long)mrx_cdiv(W, WB_TW) * mrx_cdiv(H, wb_th(k)) * B;
    return (int)(tiles < (long long)per_cu * n_cu ? tiles : (long long)per_cu * n_cu);
}
// workgroups of the thin 3x3 layer (64 -> <= 32): 63 registers and 52 KB of LDS -- two per CU cover each other's tile loads (one per CU: 76 us at
// 15 x 640 x 372 for 61 MB)
static int wb_nwg_any(int B, int Cin, int Cout, int H, int W, int k) {
    const int n1 = wb_nwg(B, H, W, k);
    if (k == 3 && Cin == 64 && Cout <= 32) {
        const long long tiles = (long long)mrx_cdiv(W, WB_TW) * mrx_cdiv(H, 8) * B;
        return (int)(tiles < 2ll * n1 ? tiles : 2ll * n1);
    }
    return n1;
}
static bool wb_thin(int Cin, int Cout, int k, int dil) {   // the two thin dilation-1 layers of the RIM
    return dil == 1 && ((k == 3 && Cin == 64 && Cout >= 1 && Cout <= 32) || (k == 5 && Cout == 64 && Cin >= 1 && Cin <= 32));
}
extern "C" int mrx_conv_wgrad_bf16_supported(int Cin, int Cout, int k, int dil) {
    return (Cin == 64 && Cout == 64 && ((k == 1 && dil == 1) || (k == 3 && dil == 2))) || wb_thin(Cin, Cout, k, dil);
}
extern "C" int64_t mrx_conv_wgrad_bf16_work_floats(int B, int H, int W, int k) {
    if (B < 1 || H < 1 || W < 1 || (k != 1 && k != 3 && k != 5)) return -1;
    return (int64_t)wb_nwg(B, H, W, k) * 64 * 64 * k * k;      // an upper bound for the thin layers
}
template <int K, int DIL, int DYP = 0, int XCB = 0>
static int wb_launch(const WgradBfArgs& a, int nwg, hipStream_t st) {
    constexpr size_t lds = (size_t)64 * wb_dys(K) + (size_t)64 * wb_xs(K, DIL);
    static_assert(lds >= (wb_th(K) / 2) * 4096 * sizeof(float) || K != 1, "1x1: LDS also holds the wave reduction");
    static bool attr_done = false;
    if (lds > 48 * 1024 && !attr_done) {
        MRX_HIP(hipFuncSetAttribute((const void*)k_conv_wgrad_bf16<K, DIL, DYP, XCB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    hipLaunchKernelGGL((k_conv_wgrad_bf16<K, DIL, DYP, XCB>), dim3(nwg), dim3(K == 1 ? 64 * wb_th(K) : 576), lds, st, a);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
extern "C" int64_t mrx_conv_wgrad_bf16_any_work_floats(int B, int Cin, int Cout, int H, int W, int k) {
    if (B < 1 || H < 1 || W < 1 || Cin < 1 || Cout < 1 || (k != 1 && k != 3 && k != 5)) return -1;
    return (int64_t)wb_nwg_any(B, Cin, Cout, H, W, k) * Cout * Cin * k * k;
}
template <int K, int NCO, int NCI, int TPW, int DYP = 0, int XCB = 0>
static int wbg_launch(const WgradBfGArgs& a, int nwg, hipStream_t st) {
    constexpr int PAD = (K - 1) / 2, PH = 8 + 2 * PAD, PW = WB_TW + 2 * PAD, XS = ((PH * PW * 2 + 255 - 16) / 256) * 256 + 16;
    constexpr int NW = (K * K + TPW - 1) / TPW;
    const size_t lds = (size_t)(a.Cout + 1) * (8 * WB_TW * 2 + 16) + (size_t)(a.Cin + 1) * XS;
    static size_t attr = 0;
    if (lds > 48 * 1024 && attr < lds) {
        MRX_HIP(hipFuncSetAttribute((const void*)k_conv_wgrad_bf16_g<K, NCO, NCI, TPW, DYP, XCB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = lds;
    }
    hipLaunchKernelGGL((k_conv_wgrad_bf16_g<K, NCO, NCI, TPW, DYP, XCB>), dim3(nwg), dim3(64 * NW), lds, st, a);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
// dw [Cout,Cin,k,k] for the shapes of mrx_conv_wgrad_bf16_supported (the 64 -> 64 layers and the two thin dilation-1 layers of the RIM)
extern "C" int mrx_conv_wgrad_bf16_any(const float* x, const float* dy, float* dw, float* work, int B, int Cin, int Cout, int H, int W, int k,
                                       int dil, int pad_mode, int accumulate, void* stream) {
    MRX_REQUIRE(x && dy && dw && work, MRX_EINVAL, "mrx_conv_wgrad_bf16_any: null pointer");
    MRX_REQUIRE(B >= 1 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_conv_wgrad_bf16_any: bad dims");
    MRX_REQUIRE(mrx_conv_wgrad_bf16_supported(Cin, Cout, k, dil), MRX_EUNSUP, "mrx_conv_wgrad_bf16_any: Cin=%d Cout=%d k=%d dilation=%d", Cin, Cout,
                k, dil);
    if (!wb_thin(Cin, Cout, k, dil)) return mrx_conv_wgrad_bf16(x, dy, dw, work, B, H, W, k, dil, pad_mode, accumulate, stream);
    WgradBfGArgs a;
    a.x = x, a.dy = dy, a.part = work, a.B = B, a.Cin = Cin, a.Cout = Cout, a.H = H, a.W = W;
    a.tiles_x = mrx_cdiv(W, WB_TW), a.ntiles = a.tiles_x * mrx_cdiv(H, 8), a.pad_mode = pad_mode;
    a.vec = (W % 4 == 0) && (((uintptr_t)x | (uintptr_t)dy) % 16 == 0);
    const int nwg = wb_nwg_any(B, Cin, Cout, H, W, k);
    hipStream_t st = (hipStream_t)stream;
    int rc = k == 3 ? wbg_launch<3, 1, 2, 1>(a, nwg, st) : wbg_launch<5, 2, 1, 2>(a, nwg, st);
    if (rc) return rc;
    const long long total = (long long)Cout * Cin * k * k;
    hipLaunchKernelGGL(k_wgrad_bf16_reduce, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, st, (const float*)work, nwg, total, dw, accumulate);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
extern "C" int mrx_conv_wgrad_bf16(const float* x, const float* dy, float* dw, float* work, int B, int H, int W, int k, int dil, int pad_mode,
                                   int accumulate, void* stream) {
    MRX_REQUIRE(x && dy && dw && work, MRX_EINVAL, "mrx_conv_wgrad_bf16: null pointer");
    MRX_REQUIRE(B >= 1 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_conv_wgrad_bf16: bad dims");
    MRX_REQUIRE((k == 1 && dil == 1) || (k == 3 && dil == 2), MRX_EUNSUP, "mrx_conv_wgrad_bf16: k=%d dilation=%d not instantiated", k, dil);
    WgradBfArgs a;
    a.x = x, a.dy = dy, a.part = work, a.B = B, a.H = H, a.W = W;
    a.tiles_x = mrx_cdiv(W, WB_TW), a.tiles_y = mrx_cdiv(H, wb_th(k)), a.ntiles = a.tiles_x * a.tiles_y, a.pad_mode = pad_mode;
    a.vec = (W % 4 == 0) && (((uintptr_t)x | (uintptr_t)dy) % 16 == 0);
    const int nwg = wb_nwg(B, H, W, k);
    hipStream_t st = (hipStream_t)stream;
    int rc = k == 1 ? wb_launch<1, 1>(a, nwg, st) : wb_launch<3, 2>(a, nwg, st);
    if (rc) return rc;
    const long long total = 64ll * 64 * k * k;
    hipLaunchKernelGGL(k_wgrad_bf16_reduce, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, st, (const float*)work, nwg, total, dw, accumulate);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// The same weight gradients with dy given as a PAIR tensor [B,32,H,W] (the bf16 gradient mrx_tl_cell_bwd leaves): 3x3 dilation 2 64 -> 64 and the thin
// 5x5 Cin <= 32 -> 64 layer.  x stays fp32 NCHW (rounded by the tile loader).
// Weight gradient of the final 3x3 convolution (64 -> Cout <= 32, dilation 1) with x CHANNEL-BLOCKED [B,8,H,W,8] (the training tape's hidden state) and dy fp32
// [B,Cout,H,W]: mrx_conv_wgrad_bf16_any's thin kernel with the blocked tile loader.  work: mrx_conv_wgrad_bf16_any_work_floats(B, 64, Cout, H, W, 3).
extern "C" int mrx_conv_wgrad_bf16_xcb(const float* x_cb8, const float* dy, float* dw, float* work, int B, int Cout, int H, int W, int pad_mode, int accumulate,
                                       void* stream) {
    MRX_REQUIRE(x_cb8 && dy && dw && work, MRX_EINVAL, "mrx_conv_wgrad_bf16_xcb: null pointer");
    MRX_REQUIRE(B >= 1 && H >= 1 && W >= 1 && Cout >= 1 && Cout <= 32, MRX_EUNSUP, "mrx_conv_wgrad_bf16_xcb: Cout=%d (1 .. 32)", Cout);
    WgradBfGArgs a;
    a.x = x_cb8, a.dy = dy, a.part = work, a.B = B, a.Cin = 64, a.Cout = Cout, a.H = H, a.W = W;
    a.tiles_x = mrx_cdiv(W, WB_TW), a.ntiles = a.tiles_x * mrx_cdiv(H, 8), a.pad_mode = pad_mode;
    a.vec = (W % 4 == 0) && (((uintptr_t)x_cb8 | (uintptr_t)dy) % 16 == 0);
    const int nwg = wb_nwg_any(B, 64, Cout, H, W, 3);
    hipStream_t st = (hipStream_t)stream;
    int rc = wbg_launch<3, 1, 2, 1, 0, 1>(a, nwg, st);
    if (rc) return rc;
    const long long total = (long long)Cout * 64 * 9;
    hipLaunchKernelGGL(k_wgrad_bf16_reduce, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, st, (const float*)work, nwg, total, dw, accumulate);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
// (x_blocked != 0: x is channel-blocked [B,8,H,W,8] -- the 64-channel layer's input in the training tape)
extern "C" int mrx_conv_wgrad_bf16_pairs(const float* x, const void* dy_pairs, float* dw, float* work, int B, int Cin, int H, int W, int k, int dil,
                                         int pad_mode, int accumulate, int x_blocked, void* stream) {
    MRX_REQUIRE(x && dy_pairs && dw && work, MRX_EINVAL, "mrx_conv_wgrad_bf16_pairs: null pointer");
    MRX_REQUIRE(B >= 1 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_conv_wgrad_bf16_pairs: bad dims");
    hipStream_t st = (hipStream_t)stream;
    const int nwg = wb_nwg(B, H, W, k);
    const int vec = (W % 4 == 0) && (((uintptr_t)x | (uintptr_t)dy_pairs) % 16 == 0);
    long long total;
    if (k == 3 && dil == 2 && Cin == 64) {
        WgradBfArgs a;
        a.x = x, a.dy = (const float*)dy_pairs, a.part = work, a.B = B, a.H = H, a.W = W;
        a.tiles_x = mrx_cdiv(W, WB_TW), a.tiles_y = mrx_cdiv(H, wb_th(k)), a.ntiles = a.tiles_x * a.tiles_y, a.pad_mode = pad_mode, a.vec = vec;
        int rc = x_blocked ? wb_launch<3, 2, 1, 1>(a, nwg, st) : wb_launch<3, 2, 1, 0>(a, nwg, st);
        if (rc) return rc;
        total = 64ll * 64 * 9;
    } else if (k == 5 && dil == 1 && Cin >= 1 && Cin <= 32 && !x_blocked) {
        WgradBfGArgs a;
        a.x = x, a.dy = (const float*)dy_pairs, a.part = work, a.B = B, a.Cin = Cin, a.Cout = 64, a.H = H, a.W = W;
        a.tiles_x = mrx_cdiv(W, WB_TW), a.ntiles = a.tiles_x * mrx_cdiv(H, 8), a.pad_mode = pad_mode, a.vec = vec;
        int rc = wbg_launch<5, 2, 1, 2, 1>(a, nwg, st);
        if (rc) return rc;
        total = 64ll * Cin * 25;
    } else {
        MRX_REQUIRE(false, MRX_EUNSUP, "mrx_conv_wgrad_bf16_pairs: Cin=%d k=%d dilation=%d not instantiated", Cin, k, dil);
    }
    hipLaunchKernelGGL(k_wgrad_bf16_reduce, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, st, (const float*)work, nwg, total, dw, accumulate);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
