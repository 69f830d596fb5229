// unet_f16.hip -- the 3x3 convolution of the U-Net (unet_block.py:251-258: Conv2d(3x3, zero pad, no bias) -> InstanceNorm2d -> LeakyReLU(0.2)) on the
// fp16 matrix pipe with fp32 results: mrx_unet_conv3x3_h, the two-term form of mrx_unet_conv3x3 (unet_fused.hip), same contract -- up to two
// sources, each plain or (raw, norm), raw output + the InstanceNorm statistics of its planes from the accumulators.
//
// Arithmetic: every fp32 operand is two fp16 terms, x = (h1 + h2) 2^-k (22 significant bits relative to a block scale that is a power of two), and
// a multiply is the three term products h1 w1 + h1 w2 + h2 w1 on v_mfma_f32_16x16x32_f16, accumulated in fp32 -- the arithmetic of the RIM layer
// kernels (rim_layer2_sb.hip).  The block scales:
//   * x: ONE exponent per launch from a bound of max |x| over all sources.  A (raw, norm) source is bounded analytically: an element of an
//     instance-normalised plane of n values has |z| <= sqrt(n - 1) (LeakyReLU only shrinks it), so sqrt(H W) needs no pass over the data
//     (the (mean, 1/std) pair MUST be the statistics of the raw planes, as this library's kernels produce them); a plain
//     source comes with a device scalar holding a bound of its maximum (mrx_max_abs, or the analytic bound of the normalised tensor it was pooled /
//     padded from).  Values far below the bound keep 22 bits down to 2^-17 of it and an ABSOLUTE error of 2^-39 of the bound below that.
//   * w: one exponent per weight tensor, found by the pack kernel.
// The fp32-MFMA kernel it replaces spends 16 x the matrix cycles per multiply-add; this one is bound by its tile loads and stores.
//
// Work split: one workgroup = one 8 x 32 output tile x NCOT blocks of 16 output channels; a step contracts 16 input channels (two halves of 8) x 9
// taps = 18 slots of 8 k-values, four slots per MFMA (five MFMAs per step and accumulator, the last one half empty: its weights are zero).  A wave
// owns two rows of the tile (four 16-pixel accumulator tiles per output-channel block).  Staging: a thread owns a pixel of the halo'd tile and one
// half (8 channels: eight coalesced plane loads), normalises + activates (the producer's statistics), splits and writes one 16-byte word per term.
#include <cstdint>
#include <cstdlib>

#include "mrx_common.h"

typedef _Float16 uh_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 uh_f16x2 __attribute__((ext_vector_type(2)));
typedef float uh_f4 __attribute__((ext_vector_type(4)));
typedef unsigned uh_u4 __attribute__((ext_vector_type(4)));

#define UH_NT 256
#define UH_TH 8
#ifndef MRX_UH_TH16
#define MRX_UH_TH16 1
#endif
#define UH_TW 32
#define UH_SC 16                   // input channels per step
#define UH_MS 5                    // MFMAs per step and accumulator (18 of 20 slots used)
#define UH_TICKET_STRIDE 32        // ints between two planes' tickets: one 128-byte line each (read-modify-writes of one line serialise: 112 adjacent tickets
                                   // = four lines took 107 k atomics one after the other, +200 us per launch)
// halo'd tile for dilation D: (8 + 2 D) rows x (32 + 2 D) columns -- 10 x 34 = 340 pixels (D = 1), 12 x 36 = 432 (D = 2)
__host__ __device__ constexpr int uh_pw(int D) { return UH_TW + 2 * D; }
__host__ __device__ constexpr int uh_pix(int D, int TH = UH_TH) { return (TH + 2 * D) * uh_pw(D); }
__host__ __device__ constexpr int uh_plane(int D, int TH = UH_TH) { return (uh_pix(D, TH) + 7) / 8 * 8; }      // pixels per (term, half) plane in LDS (16-byte words)
__host__ __device__ constexpr int uh_xbuf(int D, int TH = UH_TH, int TERMS = 2) { return TERMS * 2 * uh_plane(D, TH); }          // 16-byte words of the x buffer: [term][half][pixel]

struct UConvHArgs {
    const float* xa;     // [B,Ca,H,W]
    const float* na;     // [B,Ca,2] (mean, 1/std) or null: source A is a plain tensor
    const float* xb;     // [B,Cb,H,W] or null
    const float* nb;
    const float* bound_a;  // device scalar >= max |xa| (plain sources only)
    const float* bound_b;
    const uh_u4* packed; // mrx_unet_conv3x3_pack
    float* y;            // [B,Cout,H,W] raw
    float* tstats;       // [B][ntiles][Cout][2] (mean, M2) per tile
    int Ca, Cb, B, Cout, H, W, tiles_x, ntiles, nct, nsteps, nitems;
    int ntiles8;         // tiles of the 8-row tiling the tile statistics are indexed by (= ntiles unless the work items are 16-row tiles: TH = 16)
    float slope;
    const float* bias;   // plain convolution (UNET = false): [Cout] or null; act MRX_ACT_*; pad_mode MRX_PAD_ZERO | MRX_PAD_REPLICATE
    int act, pad_mode;
    int abl;             // probe builds (env MRX_UCONVH_ABLATE): 1 no matrix work, 2 no stores, 4 no statistics, 8 no tile loads, 16 no split / LDS writes
    int* counters;       // UNET, or null: [B][Cout] tickets, zero on entry and on exit -- the LAST tile of a plane merges the plane's tile statistics into
                         // `norm` itself (round 5: E2EVN ran 300 k_unorm_finalize launches of 5 us per step behind these kernels)
    float* norm;         // [B][Cout][2] (mean, 1/std): written by the last tile of each plane when `counters` is given
    float eps;
};

__host__ __device__ constexpr long long uh_pack_words(int Cout, int Ctot) {
    return (long long)((Ctot + UH_SC - 1) / UH_SC) * UH_MS * ((Cout + 15) / 16) * 2 * 64 + 1;   // + the header word (the weights' exponent)
}

__device__ __forceinline__ float uh_pow2(int e) {
    e = e < -120 ? -120 : (e > 120 ? 120 : e);
    return __uint_as_float((unsigned)(127 + e) << 23);
}
// exponent k with bound * 2^k in [2^14, 2^15) (0 for a zero / non-finite bound)
__device__ __forceinline__ int uh_scale_exp(float bound) {
    const int ex = (int)((__float_as_uint(bound) >> 23) & 0xffu);
    return (ex == 0 || ex == 255) ? 0 : 14 - (ex - 127);
}
// two fp16 terms of a pair of values already scaled into the fp16 range: a = h1 + h2 + O(2^-22 |a|)
__device__ __forceinline__ void uh_split2(float a, float b, unsigned& p1, unsigned& p2) {
    const uh_f16x2 h = {(_Float16)a, (_Float16)b};
    const float ra = a - (float)h.x, rb = b - (float)h.y;     // exact
    const uh_f16x2 l = {(_Float16)ra, (_Float16)rb};
    p1 = __builtin_bit_cast(unsigned, h);
    p2 = __builtin_bit_cast(unsigned, l);
}

// ---- weight pack -------------------------------------------------------------------------------------------------------------------------
// word (((S * 5 + m) * NCT + ct) * 2 + term) * 64 + lane, element j: term( w[16 ct + lane % 16][16 S + 8 (g & 1) + j][tap = 2 m + (g >> 1)] * 2^kw ),
// g = lane / 16; zero for tap 9, channels >= Ctot, output channels >= Cout.  The last word holds kw.
// one workgroup of 1024 threads, four independent loads in flight per thread (the first form -- 256 threads, one dependent load chain each -- took
// 13-47 us per weight tensor; a captured training step packs every weight on every replay)
__global__ __launch_bounds__(1024) void k_uh_wscale(const float* __restrict__ w, long long n, uh_u4* __restrict__ out, long long header) {
    __shared__ float red[16];
    float m0 = 0.f, m1 = 0.f, m2 = 0.f, m3 = 0.f;
    long long i = threadIdx.x;
    for (; i + 3072 < n; i += 4096) {
        const float a0 = w[i], a1 = w[i + 1024], a2 = w[i + 2048], a3 = w[i + 3072];
        m0 = fmaxf(m0, fabsf(a0)), m1 = fmaxf(m1, fabsf(a1)), m2 = fmaxf(m2, fabsf(a2)), m3 = fmaxf(m3, fabsf(a3));
    }
    for (; i < n; i += 1024) m0 = fmaxf(m0, fabsf(w[i]));
    float m = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 16; ++k) m = fmaxf(m, red[k]);
        out[header] = uh_u4{(unsigned)uh_scale_exp(m), 0u, 0u, 0u};
    }
}
__global__ void k_uh_pack(const float* __restrict__ w, uh_u4* __restrict__ out, int Cout, int Ctot, int nct, long long words) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= words - 1) return;
    const float sw = uh_pow2((int)out[words - 1][0]);
    const int lane = (int)(i & 63), term = (int)((i >> 6) & 1);
    long long r = i >> 7;
    const int ct = (int)(r % nct);
    r /= nct;
    const int m = (int)(r % UH_MS), S = (int)(r / UH_MS);
    const int g = lane >> 4, co = 16 * ct + (lane & 15), tap = 2 * m + (g >> 1);
    unsigned p[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float v[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int c = UH_SC * S + 8 * (g & 1) + 2 * k + e;
            v[e] = (tap < 9 && c < Ctot && co < Cout) ? w[((long long)co * Ctot + c) * 9 + tap] * sw : 0.f;
        }
        unsigned p1, p2;
        uh_split2(v[0], v[1], p1, p2);
        p[k] = term ? p2 : p1;
    }
    out[i] = uh_u4{p[0], p[1], p[2], p[3]};
}

extern "C" int64_t mrx_unet_conv3x3_pack_floats(int Cout, int Ctot) {
    if (Cout < 1 || Ctot < 1) return -1;
    return (int64_t)uh_pack_words(Cout, Ctot) * 4;
}

// w [Cout,Ctot,3,3] -> the two-term fp16 operand pack of mrx_unet_conv3x3_h (mrx_unet_conv3x3_pack_floats(Cout, Ctot) floats, 16-byte aligned)
extern "C" int mrx_unet_conv3x3_pack(const float* w, int Cout, int Ctot, float* packed, void* stream) {
    MRX_REQUIRE(w && packed && Cout >= 1 && Ctot >= 1, MRX_EINVAL, "mrx_unet_conv3x3_pack: bad argument");
    MRX_REQUIRE(((uintptr_t)packed & 15u) == 0, MRX_EINVAL, "mrx_unet_conv3x3_pack: packed must be 16-byte aligned");
    const long long words = uh_pack_words(Cout, Ctot);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_uh_wscale, dim3(1), dim3(1024), 0, st, w, (long long)Cout * Ctot * 9, reinterpret_cast<uh_u4*>(packed), words - 1);
    hipLaunchKernelGGL(k_uh_pack, dim3((unsigned)((words + 254) / 256)), dim3(256), 0, st, w, reinterpret_cast<uh_u4*>(packed), Cout, Ctot,
                       (Cout + 15) / 16, words);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- the convolution -----------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float uh_leaky(float v, float slope) { return v > 0.f ? v : v * slope; }

// sum over the 16 lanes of a row (the 16 pixels of an accumulator tile) on the vector ALU: quad swaps, then the two mirror patterns
__device__ __forceinline__ float uh_row_sum(float t) {
#define UH_DPP(v, ctrl) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xF, 0xF, true))
    t += UH_DPP(t, 0xB1);      // quad_perm [1,0,3,2]
    t += UH_DPP(t, 0x4E);      // quad_perm [2,3,0,1]
    t += UH_DPP(t, 0x141);     // row_half_mirror: quads 0 <-> 1, 2 <-> 3
    t += UH_DPP(t, 0x140);     // row_mirror: halves of the row
#undef UH_DPP
    return t;
}

// The merge of a plane's tile statistics (k_unorm_finalize's arithmetic: counts, mean, then M2 with the parallel-variance term, in double) by ONE wave:
// tiles strided over the lanes, three butterfly sums.  Called by the workgroup that took the plane's last ticket.  Visibility across the XCDs' L2s without
// a fence: the tile statistics are WRITTEN THROUGH (agent-scope relaxed atomic stores: `sc1`), acknowledged (s_waitcnt) before the ticket is taken, and read
// here with agent-scope loads that miss this XCD's L2 -- an agent-scope release fence per tile instead writes back the whole L2 every time: measured
// 7 x slower than the launch it replaces (E2EVN 1 100 -> 148 slices/s).
__device__ __forceinline__ double uh_wave_sum(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ void uh_finalize_plane(const float* __restrict__ tstats, float* __restrict__ norm, int b, int co, int Cout, int ntiles, int tiles_x, int H,
                                                  int W, float eps, int lane) {
    auto count = [&](int t) {
        const int ty = t / tiles_x, tx = t - ty * tiles_x;
        const int nr = H - ty * UH_TH < UH_TH ? H - ty * UH_TH : UH_TH, nc = W - tx * UH_TW < UH_TW ? W - tx * UH_TW : UH_TW;
        return (double)(nr * nc);
    };
    const float* ts = tstats + ((long long)b * ntiles * Cout + co) * 2;
    // every load of a pass is requested before the first one is used (an agent-scope load misses this XCD's L2: a loop of load -> use pays a memory
    // round trip per tile -- 30 per plane, 300 us behind the last tile of an image: measured); up to 1024 tiles stay in registers for the second pass
    double sn = 0.0, smean = 0.0, q = 0.0;
    if (ntiles <= 1024) {
        float mv[16], qv[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int t = lane + 64 * i, tc = t < ntiles ? t : ntiles - 1;
            mv[i] = __hip_atomic_load(ts + (long long)tc * Cout * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            qv[i] = __hip_atomic_load(ts + (long long)tc * Cout * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int t = lane + 64 * i;
            const double nb = t < ntiles ? count(t) : 0.0;
            sn += nb;
            smean += nb * (double)mv[i];
        }
        const double N = uh_wave_sum(sn), mean = uh_wave_sum(smean) / N;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int t = lane + 64 * i;
            const double d = (double)mv[i] - mean;
            q += t < ntiles ? (double)qv[i] + count(t) * d * d : 0.0;
        }
        const double m2 = uh_wave_sum(q);
        if (lane == 0) {
            norm[((long long)b * Cout + co) * 2] = (float)mean;
            norm[((long long)b * Cout + co) * 2 + 1] = 1.0f / sqrtf((float)m2 / (float)N + eps);
        }
        return;
    }
    for (int t = lane; t < ntiles; t += 64) {
        const double nb = count(t);
        sn += nb;
        smean += nb * (double)__hip_atomic_load(ts + (long long)t * Cout * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const double N = uh_wave_sum(sn), mean = uh_wave_sum(smean) / N;
    for (int t = lane; t < ntiles; t += 64) {
        const double d = (double)__hip_atomic_load(ts + (long long)t * Cout * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - mean;
        q += (double)__hip_atomic_load(ts + (long long)t * Cout * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + count(t) * d * d;
    }
    const double m2 = uh_wave_sum(q);
    if (lane == 0) {
        norm[((long long)b * Cout + co) * 2] = (float)mean;
        norm[((long long)b * Cout + co) * 2 + 1] = 1.0f / sqrtf((float)m2 / (float)N + eps);
    }
}

// One workgroup per work item (batch, cout block, tile); the tile loads of the NEXT step are in flight (registers) while the current step is
// multiplied.  (A persistent form that also prefetched the next ITEM's tile under the epilogue measured slower -- 166 instead of 128 registers, three
// instead of four workgroups per CU, uneven item counts: 14 -> 14 at 640 x 384 x 4 64.6 instead of 61.7 us, 56 -> 56 at 160 x 96 61.8 instead of 42.3.)
// UNET: the U-Net contract (lazy / plain sources, zero padding, raw output + tile statistics); else a plain convolution of ONE plain source with
// dilation DIL, zero or replicate padding and a bias + activation epilogue (conv_layers.py:121-123 for layers wider than the RIM's 64 channels).
// TK: the ticket form (a.counters given) is its own instantiation -- its merge keeps 32 tile statistics per lane in registers, which the 128-register
// budget of the four-workgroups-per-CU form does not have (84 bytes of scratch per lane when it was a run-time branch of the one kernel)
// TH (round 6): image rows per work item.  16: a wave owns FOUR rows (eight accumulator tiles per output-channel block), half as many workgroups per launch -- 40 % of the
// 8-row launch is per-workgroup cost that no phase ablation removes (profiles/r06_uconv_h_phase_ablation.txt) -- and a halo of 1.19 x instead of 1.33 x; the tile statistics
// stay on the 8-row granule (the two wave pairs of a workgroup each write their half's record: mrx_unorm_finalize_tiled is unchanged); three workgroups per CU.
// TERMS (round 6): 2 = fp32-class results (two fp16 terms per operand, three term products); 1 = the reference's `precision: 16` inference arithmetic
// (base_vn_run.yaml:98: torch.autocast(float16) around the forward pass -- fp16-rounded operands, fp32 sums; the raw outputs and their statistics STAY fp32
// here, autocast rounds them to fp16): the first term only -- a third of the matrix work, half the LDS image and fragment reads.
template <int NCOT, int DIL, bool UNET, bool TK = false, int TH = UH_TH, int TERMS = 2>
#define UH_WGS(NCOT, TK) ((NCOT) == 1 ? ((TK) ? 3 : 4) : ((NCOT) == 2 ? 3 : 2))
__global__ __launch_bounds__(UH_NT, TH == 16 ? 3 : UH_WGS(NCOT, TK)) void k_uconv_h(UConvHArgs a) {
    static_assert(TH == 8 || TH == 16, "8 or 16 rows per work item");
    static_assert(TH == 8 || !TK, "the ticket form walks 8-row tiles");
    static_assert(TERMS == 2 || (TERMS == 1 && !TK), "one term: the precision-16 route (no ticket form)");
    constexpr int RPW = TH / 4, NSG = 2 * RPW;                    // image rows per wave, 16-pixel accumulator tiles per wave and output-channel block
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_uh[];
    constexpr int UH_PW = uh_pw(DIL), UH_PIX = uh_pix(DIL, TH), UH_PLANE = uh_plane(DIL, TH), UH_XBUF = uh_xbuf(DIL, TH, TERMS);
    constexpr int NSLOT = (UH_PIX + 127) / 128;                 // tile pixels per staging thread: 3 (dilation 1), 4 (dilation 2)
    constexpr int WBUF = UH_MS * NCOT * TERMS * 64;             // 16-byte words of the weight buffer: [m][ct][term][lane]
    constexpr int NWL = (WBUF + UH_NT - 1) / UH_NT;
    // ONE operand buffer: the next step's tile waits in registers while this step is multiplied (32 / 42 KB per workgroup: several workgroups
    // share a CU and fill each other's barriers)
    uh_u4* Xh = reinterpret_cast<uh_u4*>(smem_uh);              // [term][half][UH_PLANE]
    uh_u4* Wh = Xh + UH_XBUF;                                   // [WBUF]
    float* red = reinterpret_cast<float*>(Wh + WBUF);           // [2 passes][4 waves][NCOT * 16]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const long long plane = (long long)a.H * a.W;
    const int Ctot = a.Ca + a.Cb;
    const int ncob = (a.nct + NCOT - 1) / NCOT;

    // the launch's operand scale: lazy sources are bounded by sqrt(n) of their planes, plain ones by the caller's device scalar
    float bound = 0.f;
    {
        const float lazy_bound = 4.f * sqrtf((float)plane);     // (two bits of headroom: statistics that are a rounding error off still cannot overflow)
        bound = (UNET && a.na) ? lazy_bound : a.bound_a[0];
        if (UNET && a.Cb) bound = fmaxf(bound, a.nb ? lazy_bound : a.bound_b[0]);
    }
    const int kx = uh_scale_exp(bound), kw = (int)a.packed[uh_pack_words(a.Cout, Ctot) - 1][0];
    const float sx = uh_pow2(kx), unscale = uh_pow2(-kx) * uh_pow2(-kw);
    const float slope = a.slope;

    // staging roles: waves 0, 1 own the first half (8 channels) of a step, waves 2, 3 the second; a thread owns pixels p, p + 128, p + 256 (, p + 384)
    const int half = wave >> 1, p0 = tid & 127;
    int pry[NSLOT], prx[NSLOT];
#pragma unroll
    for (int v = 0; v < NSLOT; ++v) {
        const int p = p0 + 128 * v;
        pry[v] = p / UH_PW, prx[v] = p - pry[v] * UH_PW;
    }
    float xv[NSLOT][8];
    float nm[8], ni[8];
    unsigned lzm = 0, okm = 0;                                     // of the data in flight: lazy channels / pixels inside the image
    auto issue_x = [&](int item, int q) {
        const int tile = item % a.ntiles, bc = item / a.ntiles, b = bc / ncob;
        const int ty0 = tile / a.tiles_x, h0 = ty0 * TH, w0 = (tile - ty0 * a.tiles_x) * UH_TW;
        unsigned goff[NSLOT];
        okm = 0;
#pragma unroll
        for (int v = 0; v < NSLOT; ++v) {
            const int gy = h0 + pry[v] - DIL, gx = w0 + prx[v] - DIL;
            // (replicate padding IS the clamped load: nothing to zero)
            const bool ok = p0 + 128 * v < UH_PIX && ((!UNET && a.pad_mode == MRX_PAD_REPLICATE) || (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W));
            // every lane loads a valid element (clamped); what lies outside the image is zeroed when it is written to LDS: no branch per load
            const int cy = gy < 0 ? 0 : (gy >= a.H ? a.H - 1 : gy), cx = gx < 0 ? 0 : (gx >= a.W ? a.W - 1 : gx);
            goff[v] = (unsigned)(cy * a.W + cx);
            okm |= (ok ? 1u : 0u) << v;
        }
        lzm = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = UH_SC * q + 8 * half + j;
            const bool valid = c < Ctot;
            const int cc = valid ? c : 0;
            const bool inb = UNET && cc >= a.Ca;
            const int cl = inb ? cc - a.Ca : cc, Cs = inb ? a.Cb : a.Ca;
            const float* p = (inb ? a.xb : a.xa) + ((long long)b * Cs + cl) * plane;
            const float* nrm = UNET ? (inb ? a.nb : a.na) : nullptr;
            // (x - mean) * (1/std * 2^kx): the operand scale rides on the normalisation (exact: a power of two); a plain source has mean 0, 1/std 1
            nm[j] = 0.f, ni[j] = valid ? sx : 0.f;
            if (nrm) {
                nm[j] = nrm[((long long)b * Cs + cl) * 2];
                ni[j] = valid ? nrm[((long long)b * Cs + cl) * 2 + 1] * sx : 0.f;
                lzm |= 1u << j;
            }
#pragma unroll
            for (int v = 0; v < NSLOT; ++v) xv[v][j] = (a.abl & 8) ? 0.f : p[goff[v]];   // (a channel past the last one reads channel 0 and is multiplied by 0)
        }
    };
    auto commit_x = [&]() {
        uh_u4* dst = Xh + half * UH_PLANE;
        if (a.abl & 16) return;
#pragma unroll
        for (int v = 0; v < NSLOT; ++v) {
            const int p = p0 + 128 * v;
            const bool ok = (okm >> v) & 1u;
            unsigned p1[4], p2[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float t[2];
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int j = 2 * k + e;
                    const float z = (xv[v][j] - nm[j]) * ni[j];
                    t[e] = ((lzm >> j) & 1u) ? (slope <= 1.f ? fmaxf(z, z * slope) : fminf(z, z * slope)) : z;      // LeakyReLU of a scaled value
                }
                uh_split2(t[0], t[1], p1[k], p2[k]);
                p1[k] = ok ? p1[k] : 0u;                                     // zero padding applies to the normalised tensor
                p2[k] = ok ? p2[k] : 0u;
            }
            if (p < UH_PIX) {
                dst[p] = uh_u4{p1[0], p1[1], p1[2], p1[3]};
                if constexpr (TERMS == 2) dst[2 * UH_PLANE + p] = uh_u4{p2[0], p2[1], p2[2], p2[3]};
            }
        }
    };
    uh_u4 wr[NWL];
    auto issue_w = [&](int item, int q) {
        const int ct0 = ((item / a.ntiles) % ncob) * NCOT;
#pragma unroll
        for (int j = 0; j < NWL; ++j) {
            const int i = tid + j * UH_NT;
            const int ln = i & 63, term = TERMS == 2 ? (i >> 6) & 1 : 0, r = TERMS == 2 ? i >> 7 : i >> 6, ct = r % NCOT, m = r / NCOT;
            const bool okw = i < WBUF && ct0 + ct < a.nct;
            wr[j] = okw ? a.packed[((((long long)q * UH_MS + m) * a.nct + ct0 + ct) * 2 + term) * 64 + ln] : uh_u4{0u, 0u, 0u, 0u};
        }
    };
    auto commit_w = [&]() {
#pragma unroll
        for (int j = 0; j < NWL; ++j) {
            const int i = tid + j * UH_NT;
            if (i < WBUF) Wh[i] = wr[j];
        }
    };

    // this lane's k-group: slot 4 m + lg of step m -> tap 2 m + (lg >> 1), half lg & 1 (the slots of tap 9 carry zero weights: any valid address)
    int toff[UH_MS];
#pragma unroll
    for (int m = 0; m < UH_MS; ++m) {
        int tap = 2 * m + (lg >> 1);
        tap = tap > 8 ? 8 : tap;
        toff[m] = ((tap / 3) * UH_PW + (tap % 3)) * DIL;
    }
    const uh_u4* xq = Xh + (lg & 1) * UH_PLANE + (RPW * wave) * UH_PW + l15;
    const uh_u4* wq = Wh + lane;

    // workgroup -> work item: the runtime deals consecutive workgroups round-robin over the 8 XCDs (8 L2 caches); with item = blockIdx.x the two horizontal
    // neighbours of a tile -- which re-read 2 of its 34 halo'd columns, and the tiles above / below 2 of its 10 rows -- always sat behind OTHER L2s.  The band
    // map gives every XCD a contiguous range of items (MRX_UH_NO_BAND: A/B).
    // MRX_UH_PERSIST (A/B, round 5): the grid is the number of RESIDENT workgroups and each walks its items -- no workgroup launch (registers, LDS, the
    // scale / offset set-up above) per 8 x 32 tile; with MRX_UH_PERSIST = 2 the next item's first tile is requested as soon as this item's last one is
    // committed to LDS (its registers are free from there on): the load latency of item i + 1 runs under the matrix work, stores and statistics of item i.
#define UH_ITEM(i) ((int)mrx_xcd_band((i), a.nitems))
    {                                                // one item per workgroup (the product form)
    const int item = UH_ITEM((int)blockIdx.x);
    issue_x(item, 0);
    issue_w(item, 0);
    {
        const int tile = item % a.ntiles, bc = item / a.ntiles, b = bc / ncob, co0 = (bc - b * ncob) * NCOT * 16;
        const int ty0 = tile / a.tiles_x, h0 = ty0 * TH, w0 = (tile - ty0 * a.tiles_x) * UH_TW;
        uh_f4 acc[NSG][NCOT];
#pragma unroll
        for (int sg = 0; sg < NSG; ++sg)
#pragma unroll
            for (int ct = 0; ct < NCOT; ++ct) acc[sg][ct] = (uh_f4){0.f, 0.f, 0.f, 0.f};
        for (int q = 0; q < a.nsteps; ++q) {
            if (q) __syncthreads();   // every wave is done with the operands of step q - 1
            commit_x();
            commit_w();
            __syncthreads();          // step q staged
            if (q + 1 < a.nsteps) {   // the next step's loads fly under this step's matrix work
                issue_x(item, q + 1);
                issue_w(item, q + 1);
            }
            if (!(a.abl & 1))
#pragma unroll
            for (int m = 0; m < UH_MS; ++m) {
                uh_f16x8 a1[NCOT], a2[NCOT];
#pragma unroll
                for (int ct = 0; ct < NCOT; ++ct) {
                    a1[ct] = __builtin_bit_cast(uh_f16x8, wq[((m * NCOT + ct) * TERMS + 0) * 64]);
                    if constexpr (TERMS == 2) a2[ct] = __builtin_bit_cast(uh_f16x8, wq[((m * NCOT + ct) * 2 + 1) * 64]);
                }
#pragma unroll
                for (int sg = 0; sg < NSG; ++sg) {
                    const int pix = (sg >> 1) * UH_PW + (sg & 1) * 16 + toff[m];
                    const uh_f16x8 b1 = __builtin_bit_cast(uh_f16x8, xq[pix]);
                    if constexpr (TERMS == 2) {
                        const uh_f16x8 b2 = __builtin_bit_cast(uh_f16x8, xq[2 * UH_PLANE + pix]);
#pragma unroll
                        for (int ct = 0; ct < NCOT; ++ct) {
                            acc[sg][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2[ct], b1, acc[sg][ct], 0, 0, 0);
                            acc[sg][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[ct], b2, acc[sg][ct], 0, 0, 0);
                            acc[sg][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[ct], b1, acc[sg][ct], 0, 0, 0);
                        }
                    } else {
#pragma unroll
                        for (int ct = 0; ct < NCOT; ++ct) acc[sg][ct] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1[ct], b1, acc[sg][ct], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);     // (keeps the scheduler from hoisting every step's operand reads to the top: registers)
            }
        }

        // plain convolution: this lane's 4 NCOT bias values, requested in one block behind the matrix loop (held across it they cost 16 registers the 4-block form does not have; read inside the epilogue's loops, behind `if (a.bias)`, every
        // one of the 16 NCOT loads was waited for on its own: 64 round trips to the cache per workgroup at NCOT = 4)
        float bv[NCOT][4];
#pragma unroll
        for (int ct = 0; ct < NCOT; ++ct)
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[ct][r] = 0.f;
        if constexpr (!UNET) {
            if (a.bias) {
#pragma unroll
                for (int ct = 0; ct < NCOT; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int co = co0 + 16 * ct + 4 * lg + r;
                        bv[ct][r] = a.bias[co < a.Cout ? co : 0];
                    }
            }
        }
        bool ok[NSG];
#pragma unroll
        for (int sg = 0; sg < NSG; ++sg) {
            const int oy = h0 + RPW * wave + (sg >> 1), ox = w0 + (sg & 1) * 16 + l15;
            ok[sg] = oy < a.H && ox < a.W;
#pragma unroll
            for (int ct = 0; ct < NCOT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = acc[sg][ct][r] * unscale;                            // exact: powers of two
                    if constexpr (!UNET) {
                        v += bv[ct][r];
                        if (a.act == MRX_ACT_RELU) v = v > 0.f ? v : 0.f;
                        else if (a.act == MRX_ACT_LEAKY) v = v > 0.f ? v : v * slope;
                    }
                    acc[sg][ct][r] = v;
                }
        }
        // stores: lane = pixel of a 16-pixel accumulator tile, four output channels per lane (64-byte segments; trading rows between the two
        // tiles of an image row with v_permlane16_swap, for 128-byte segments, measured no faster)
        auto store_y = [&]() {
            if (a.abl & 2) return;
#pragma unroll
            for (int sg = 0; sg < NSG; ++sg) {
                const int oy = h0 + RPW * wave + (sg >> 1), ox = w0 + (sg & 1) * 16 + l15;
                if (oy < a.H && ox < a.W) {
#pragma unroll
                    for (int ct = 0; ct < NCOT; ++ct)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int co = co0 + 16 * ct + 4 * lg + r;
                            if (co < a.Cout) a.y[((long long)b * a.Cout + co) * plane + (long long)oy * a.W + ox] = acc[sg][ct][r];
                        }
                }
            }
        };
        constexpr bool tickets = UNET && TK;
        if (!tickets) store_y();          // (with tickets the tile is stored BEHIND them: the stores cover the tickets' round trip)
        if (UNET && !(a.abl & 4)) {
        // InstanceNorm statistics of this tile, per cout (the scheme of k_uconv, unet_fused.hip: mean over the tile's valid pixels, then the squared
        // deviations from that mean; k_unorm_finalize merges the tiles in double)
        // (TH = 16: the workgroup's two wave pairs are the two 8-row granules of the statistics -- rows h0 .. h0 + 7 and h0 + 8 .. h0 + 15; the second may lie below the image)
        constexpr int NG = TH / UH_TH, WPG = 4 / NG;
        const int grp = wave / WPG, hg = h0 + UH_TH * grp;
        const int nrow = a.H - hg < UH_TH ? a.H - hg : UH_TH, ncol = a.W - w0 < UH_TW ? a.W - w0 : UH_TW;
        const float inv_n = nrow > 0 ? 1.0f / (float)(nrow * ncol) : 0.f;
        float mean[NCOT][4];
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            float* rp = red + pass * (4 * NCOT * 16);
#pragma unroll
            for (int ct = 0; ct < NCOT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float t = 0.f;
#pragma unroll
                    for (int sg = 0; sg < NSG; ++sg) {
                        const float v = acc[sg][ct][r];
                        const float d = pass == 0 ? v : (v - mean[ct][r]) * (v - mean[ct][r]);
                        t += ok[sg] ? d : 0.f;
                    }
                    t = uh_row_sum(t);
                    if (l15 == 0) rp[wave * (NCOT * 16) + 16 * ct + 4 * lg + r] = t;
                }
            __syncthreads();
#pragma unroll
            for (int ct = 0; ct < NCOT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = 16 * ct + 4 * lg + r;
                    const float t = NG == 1 ? (rp[c] + rp[NCOT * 16 + c]) + (rp[2 * NCOT * 16 + c] + rp[3 * NCOT * 16 + c])
                                            : rp[(2 * grp) * (NCOT * 16) + c] + rp[(2 * grp + 1) * (NCOT * 16) + c];
                    if (pass == 0) {
                        mean[ct][r] = t * inv_n;
                    } else if (wave == grp * WPG && l15 == 0 && co0 + c < a.Cout && nrow > 0) {
                        const int tile8 = NG == 1 ? tile : (ty0 * NG + grp) * a.tiles_x + (tile - ty0 * a.tiles_x);
                        float* ts = a.tstats + (((long long)b * a.ntiles8 + tile8) * a.Cout + co0 + c) * 2;
                        if constexpr (TK) {                          // written through to the coherence point (read by whichever workgroup merges the plane)
                            __hip_atomic_store(ts, mean[ct][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            __hip_atomic_store(ts + 1, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        } else {
                            ts[0] = mean[ct][r];
                            ts[1] = t;
                        }
                    }
                }
        }
        if constexpr (TK) {
            // tickets: the thread that wrote a channel's tile statistics publishes them (agent-scope release) and counts the tile; whoever counts the
            // plane's LAST tile marks the channel, and the four waves merge the marked planes (usually none; all of a tile's channels for the last
            // tile of an image) -- no launch behind this one, nothing serial across the chip: the other images' tiles keep the CUs busy meanwhile
            int* flag = reinterpret_cast<int*>(red);                 // (the reductions above are done with `red`: every wave passed their last barrier)
            // The statistics stores above are write-through (sc1) but asynchronous: a workgroup-scope release fence emits NO vmcnt wait on gfx950 (non-tgsplit
            // mode), and the ticket below goes to another address and channel -- it could become visible first, and the workgroup of another XCD that takes the
            // plane's last ticket would merge stale statistics.  The storing thread waits for its stores to be acknowledged, explicitly, before the barrier
            // and the ticket (round-5 advisor finding; far cheaper than an agent-scope release, which would also write back the L2).
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            int tk[NCOT][4];
            if (wave == 0 && l15 == 0) {
#pragma unroll
                for (int ct = 0; ct < NCOT; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int c = 16 * ct + 4 * lg + r;
                        tk[ct][r] = co0 + c < a.Cout ? __hip_atomic_fetch_add(a.counters + ((long long)b * a.Cout + co0 + c) * UH_TICKET_STRIDE, 1, __ATOMIC_RELAXED,
                                                                             __HIP_MEMORY_SCOPE_AGENT) : -1;
                    }
            }
            store_y();                                              // the tile's 4 x 4 NCOT stores per lane go out while the tickets travel
            if (wave == 0 && l15 == 0) {
#pragma unroll
                for (int ct = 0; ct < NCOT; ++ct)
#pragma unroll
                    for (int r = 0; r < 4; ++r) flag[16 * ct + 4 * lg + r] = tk[ct][r] == a.ntiles - 1;
            }
            __syncthreads();
            for (int c = wave; c < NCOT * 16; c += UH_NT / 64) {
                if (!flag[c]) continue;                             // (wave-uniform)
                uh_finalize_plane(a.tstats, a.norm, b, co0 + c, a.Cout, a.ntiles, a.tiles_x, a.H, a.W, a.eps, lane);
                if (lane == 0) __hip_atomic_store(a.counters + ((long long)b * a.Cout + co0 + c) * UH_TICKET_STRIDE, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // left as it was found
            }
        }
        }
    }
    }
}

int mrx_unorm_finalize_tiled(const float* tstats, float* norm, int B, int ntiles, int tiles_x, int Cout, int H, int W, float eps, hipStream_t st);

template <int NCOT, int DIL, bool UNET, bool TK = false, int TH = UH_TH, int TERMS = 2>
static int launch_uconv_h(const UConvHArgs& a, hipStream_t st) {
    constexpr size_t lds = 16 * (size_t)(uh_xbuf(DIL, TH, TERMS) + UH_MS * NCOT * TERMS * 64) + sizeof(float) * 2 * 4 * NCOT * 16;
    static bool attr_done = false;   // once per instantiation: keeps launches legal under hipGraph capture
    if (lds > 48 * 1024 && !attr_done) {
        MRX_HIP(hipFuncSetAttribute((const void*)k_uconv_h<NCOT, DIL, UNET, TK, TH, TERMS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_done = true;
    }
    const long long nitems = (long long)a.ntiles * mrx_cdiv(a.nct, NCOT) * a.B;
    MRX_REQUIRE(nitems < (1ll << 31), MRX_EUNSUP, "two-term fp16 convolution: %lld work items", nitems);
    UConvHArgs a2 = a;
    a2.nitems = (int)nitems;
    long long grid = nitems;
    hipLaunchKernelGGL((k_uconv_h<NCOT, DIL, UNET, TK, TH, TERMS>), dim3((unsigned)grid), dim3(UH_NT), lds, st, a2);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
// output-channel blocks per work item: as many as still leave a work item per CU (every block re-stages the tile: fewer blocks, less staging)
static int uh_pick_ncot(int nct, long long tiles_b) {
    if (nct >= 4 && tiles_b * mrx_cdiv(nct, 4) >= 256) return 4;
    if (nct >= 2 && tiles_b * mrx_cdiv(nct, 2) >= 256) return 2;
    return nct >= 2 && tiles_b >= 512 ? 2 : 1;
}

// mrx_unet_conv3x3 with two-term fp16 operands (see the head of this file).  packed: mrx_unet_conv3x3_pack of w [Cout, Ca + Cb, 3, 3];
// bound_a / bound_b: device scalars >= max |x| of a PLAIN source (ignored -- may be NULL -- for a (raw, norm) source, whose bound is sqrt(H W)).
// work: mrx_unet_conv3x3_work_floats(B, Cout, H, W) floats.
static int unet_conv3x3_h_impl(const float* xa, const float* na, const float* bound_a, int Ca, const float* xb, const float* nb, const float* bound_b,
                               int Cb, const float* packed, float* y, float* norm, float* work, int* counters, int B, int Cout, int H, int W, float eps,
                               float slope, void* stream, int terms = 2) {
    MRX_REQUIRE(xa && packed && y && norm && work && Ca >= 1 && Cb >= 0 && (Cb == 0 || xb), MRX_EINVAL, "mrx_unet_conv3x3_h: bad argument");
    MRX_REQUIRE((na || bound_a) && (Cb == 0 || nb || bound_b), MRX_EINVAL, "mrx_unet_conv3x3_h: a plain source needs the bound of its maximum");
    MRX_REQUIRE(B >= 0 && Cout >= 1 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_unet_conv3x3_h: bad dims");
    MRX_REQUIRE(B <= 65535 && Cout <= 16 * 65535 && (long long)H * W < (1ll << 30), MRX_EUNSUP, "mrx_unet_conv3x3_h: size");
    MRX_REQUIRE(mrx_arith() == MRX_ARITH_F16X2, MRX_EUNSUP, "mrx_unet_conv3x3_h: the two-term fp16 form is off (MRIDC_AMD_ARITH)");
    if (B == 0) return MRX_OK;
    if (!na) MRX_CHECK_BOUND("mrx_unet_conv3x3_h (source a)", xa, (long long)B * Ca * H * W, bound_a, stream);
    if (Cb && !nb) MRX_CHECK_BOUND("mrx_unet_conv3x3_h (source b)", xb, (long long)B * Cb * H * W, bound_b, stream);
    UConvHArgs a;
    a.xa = xa, a.na = na, a.bound_a = bound_a, a.xb = Cb ? xb : nullptr, a.nb = Cb ? nb : nullptr, a.bound_b = bound_b;
    a.packed = reinterpret_cast<const uh_u4*>(packed), a.y = y, a.tstats = work;
    a.Ca = Ca, a.Cb = Cb, a.B = B, a.Cout = Cout, a.H = H, a.W = W, a.tiles_x = mrx_cdiv(W, UH_TW), a.slope = slope;
    a.nct = (Cout + 15) / 16, a.nsteps = (Ca + Cb + UH_SC - 1) / UH_SC;
    a.abl = MRX_DEBUG_ENV("MRX_UCONVH_ABLATE") ? atoi(MRX_DEBUG_ENV("MRX_UCONVH_ABLATE")) : 0;
    const int ntiles = a.tiles_x * mrx_cdiv(H, UH_TH);
    a.ntiles = ntiles, a.ntiles8 = ntiles;
    a.bias = nullptr, a.act = MRX_ACT_NONE, a.pad_mode = MRX_PAD_ZERO;
    a.counters = counters, a.norm = norm, a.eps = eps;
    hipStream_t st = (hipStream_t)stream;
    const int ncot = uh_pick_ncot(a.nct, (long long)ntiles * B);
    if (counters) return ncot == 4 ? launch_uconv_h<4, 1, true, true>(a, st) : (ncot == 2 ? launch_uconv_h<2, 1, true, true>(a, st) : launch_uconv_h<1, 1, true, true>(a, st));
    // 16-row work items where a launch has workgroups to spare (MRX_UH_TH16 = 0: the 8-row form everywhere, A/B)
    int rc;
    const long long items16 = (long long)a.tiles_x * mrx_cdiv(H, 16) * B;
    const bool th16 = MRX_UH_TH16 && ncot == 1 && items16 >= 2048;   // (E2EVN's 14 -> 14 layers at 8 x 640 x 372: 1 248 against 1 227 slices/s; no launch of this library with two output-channel blocks has that many items)
    if (th16) a.ntiles = a.tiles_x * mrx_cdiv(H, 16);
    if (terms == 1)
        rc = th16 ? launch_uconv_h<1, 1, true, false, 16, 1>(a, st)
                  : (ncot == 4 ? launch_uconv_h<4, 1, true, false, UH_TH, 1>(a, st)
                               : (ncot == 2 ? launch_uconv_h<2, 1, true, false, UH_TH, 1>(a, st) : launch_uconv_h<1, 1, true, false, UH_TH, 1>(a, st)));
    else
        rc = th16 ? launch_uconv_h<1, 1, true, false, 16>(a, st)
                  : (ncot == 4 ? launch_uconv_h<4, 1, true>(a, st) : (ncot == 2 ? launch_uconv_h<2, 1, true>(a, st) : launch_uconv_h<1, 1, true>(a, st)));
    if (rc) return rc;
    return mrx_unorm_finalize_tiled(work, norm, B, ntiles, a.tiles_x, Cout, H, W, eps, st);
}
extern "C" int mrx_unet_conv3x3_h(const float* xa, const float* na, const float* bound_a, int Ca, const float* xb, const float* nb, const float* bound_b,
                                  int Cb, const float* packed, float* y, float* norm, float* work, int B, int Cout, int H, int W, float eps,
                                  float slope, void* stream) {
    return unet_conv3x3_h_impl(xa, na, bound_a, Ca, xb, nb, bound_b, Cb, packed, y, norm, work, nullptr, B, Cout, H, W, eps, slope, stream);
}
// The same convolution in the reference's `precision: 16` inference arithmetic (base_vn_run.yaml:98, base_unet_run.yaml:96: native AMP = torch.autocast(float16)
// around the forward pass): operands rounded to fp16 ONCE (the first term of the pack; a power-of-two block scale keeps small inputs out of the fp16
// subnormals), exact products, fp32 sums; raw output and statistics fp32 (autocast rounds the output to fp16: this form is the closer one to fp32).
extern "C" int mrx_unet_conv3x3_p16(const float* xa, const float* na, const float* bound_a, int Ca, const float* xb, const float* nb, const float* bound_b,
                                    int Cb, const float* packed, float* y, float* norm, float* work, int B, int Cout, int H, int W, float eps, float slope,
                                    void* stream) {
    return unet_conv3x3_h_impl(xa, na, bound_a, Ca, xb, nb, bound_b, Cb, packed, y, norm, work, nullptr, B, Cout, H, W, eps, slope, stream, 1);
}
// ... with the merge of the tile statistics inside the convolution launch: `counters` = B * Cout ints that are ZERO on entry and zero again on exit
// (one buffer serves every call of a stream; two streams need two buffers); ticket of plane p at counters[32 p]: mrx_unet_conv3x3_hc_ticket_ints(B, Cout) ints.  Same `norm` up to the order of three double-precision sums.
extern "C" int64_t mrx_unet_conv3x3_hc_ticket_ints(int B, int Cout) { return B < 0 || Cout < 1 ? -1 : (int64_t)B * Cout * UH_TICKET_STRIDE; }
extern "C" int mrx_unet_conv3x3_hc(const float* xa, const float* na, const float* bound_a, int Ca, const float* xb, const float* nb, const float* bound_b,
                                   int Cb, const float* packed, float* y, float* norm, float* work, int* counters, int B, int Cout, int H, int W,
                                   float eps, float slope, void* stream) {
    MRX_REQUIRE(counters, MRX_EINVAL, "mrx_unet_conv3x3_hc: null counters");
    return unet_conv3x3_h_impl(xa, na, bound_a, Ca, xb, nb, bound_b, Cb, packed, y, norm, work, counters, B, Cout, H, W, eps, slope, stream);
}

// y = act(conv3x3(x, dilation 1 | 2, zero | replicate padding) + bias) for any channel counts, on two-term fp16 operands (conv_layers.py:121-123; the
// layers the 64-channel kernels do not cover: qRIM's 128 -> 128, DIDN, ...).  bound: device scalar >= max |x| (mrx_max_abs, or what the producer
// of x knows); packed: mrx_unet_conv3x3_pack of w [Cout, Cin, 3, 3].
extern "C" int mrx_conv3x3_h_supported(int Cin, int Cout, int k, int dil) {
    return Cin >= 1 && Cout >= 1 && k == 3 && (dil == 1 || dil == 2) && mrx_arith() == MRX_ARITH_F16X2;
}
static int conv3x3_h_impl(const float* x, const float* bound, const float* packed, const float* bias, float* y, int B, int Cin, int Cout, int H, int W, int dil,
                          int pad_mode, int act, float slope, void* stream, int terms) {
    MRX_REQUIRE(x && bound && packed && y && x != y, MRX_EINVAL, "mrx_conv3x3_h: null or aliased pointer");
    MRX_REQUIRE(B >= 0 && Cin >= 1 && Cout >= 1 && H >= 1 && W >= 1 && (dil == 1 || dil == 2), MRX_EINVAL, "mrx_conv3x3_h: bad dims");
    MRX_REQUIRE(pad_mode == MRX_PAD_ZERO || pad_mode == MRX_PAD_REPLICATE, MRX_EINVAL, "mrx_conv3x3_h: bad pad mode %d", pad_mode);
    MRX_REQUIRE(act >= 0 && act <= 2, MRX_EINVAL, "mrx_conv3x3_h: bad activation %d", act);
    MRX_REQUIRE((long long)H * W < (1ll << 30), MRX_EUNSUP, "mrx_conv3x3_h: size");
    MRX_REQUIRE(mrx_arith() == MRX_ARITH_F16X2, MRX_EUNSUP, "mrx_conv3x3_h: the two-term fp16 form is off (MRIDC_AMD_ARITH)");
    if (B == 0) return MRX_OK;
    MRX_CHECK_BOUND("mrx_conv3x3_h", x, (long long)B * Cin * H * W, bound, stream);
    UConvHArgs a;
    a.xa = x, a.na = nullptr, a.bound_a = bound, a.xb = nullptr, a.nb = nullptr, a.bound_b = nullptr;
    a.packed = reinterpret_cast<const uh_u4*>(packed), a.y = y, a.tstats = nullptr;
    a.Ca = Cin, a.Cb = 0, a.B = B, a.Cout = Cout, a.H = H, a.W = W, a.tiles_x = mrx_cdiv(W, UH_TW), a.slope = slope;
    a.ntiles = a.tiles_x * mrx_cdiv(H, UH_TH), a.nct = (Cout + 15) / 16, a.nsteps = (Cin + UH_SC - 1) / UH_SC;
    a.ntiles8 = a.ntiles;
    a.abl = 0, a.bias = bias, a.act = act, a.pad_mode = pad_mode;
    a.counters = nullptr, a.norm = nullptr, a.eps = 0.f;
    hipStream_t st = (hipStream_t)stream;
    const int ncot = uh_pick_ncot(a.nct, (long long)a.ntiles * B);
    if (terms == 1) {
        if (dil == 1)
            return ncot == 4 ? launch_uconv_h<4, 1, false, false, UH_TH, 1>(a, st)
                             : (ncot == 2 ? launch_uconv_h<2, 1, false, false, UH_TH, 1>(a, st) : launch_uconv_h<1, 1, false, false, UH_TH, 1>(a, st));
        return ncot == 4 ? launch_uconv_h<4, 2, false, false, UH_TH, 1>(a, st)
                         : (ncot == 2 ? launch_uconv_h<2, 2, false, false, UH_TH, 1>(a, st) : launch_uconv_h<1, 2, false, false, UH_TH, 1>(a, st));
    }
    if (dil == 1) return ncot == 4 ? launch_uconv_h<4, 1, false>(a, st) : (ncot == 2 ? launch_uconv_h<2, 1, false>(a, st) : launch_uconv_h<1, 1, false>(a, st));
    return ncot == 4 ? launch_uconv_h<4, 2, false>(a, st) : (ncot == 2 ? launch_uconv_h<2, 2, false>(a, st) : launch_uconv_h<1, 2, false>(a, st));
}
extern "C" int mrx_conv3x3_h(const float* x, const float* bound, const float* packed, const float* bias, float* y, int B, int Cin, int Cout, int H,
                             int W, int dil, int pad_mode, int act, float slope, void* stream) {
    return conv3x3_h_impl(x, bound, packed, bias, y, B, Cin, Cout, H, W, dil, pad_mode, act, slope, stream, 2);
}
// ... in the reference's `precision: 16` inference arithmetic (base_qcirim_run.yaml:204 and every other *_run.yaml: torch.autocast(float16)): operands rounded to
// fp16 once (the first term of the same pack), fp32 sums, fp32 result (see mrx_unet_conv3x3_p16)
extern "C" int mrx_conv3x3_p16(const float* x, const float* bound, const float* packed, const float* bias, float* y, int B, int Cin, int Cout, int H,
                               int W, int dil, int pad_mode, int act, float slope, void* stream) {
    return conv3x3_h_impl(x, bound, packed, bias, y, B, Cin, Cout, H, W, dil, pad_mode, act, slope, stream, 1);
}
