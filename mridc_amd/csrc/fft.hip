// fft.hip -- LDS-resident mixed-radix 2-D FFT kernels for gfx950 and the fused MRI operators built on them:
//   mrx_fft2          fft2 / ifft2                      (reference common/parts/fft.py:13-166)
//   mrx_sens_expand   fft2(x * S)                       (vn_block.py:51-69, rim_utils.py:44-51)
//   mrx_sens_reduce   sum_c ifft2(k) * conj(S)          (vn_block.py:71-87, rim_block.py:199-210)
//   mrx_llg           log_likelihood_gradient           (rim_utils.py:11-67)
//
// A 2-D transform is a row pass (contiguous rows staged in LDS, several rows per workgroup) and a column pass
// (a tile of adjacent columns staged in LDS with the column index fastest, so global accesses stay contiguous and
// all lanes of a wave run the same butterfly on neighbouring columns).  Centred transforms fold ifftshift/fftshift
// into the load/store index: LDS position p <-> global index (p + n/2) mod n on both sides (fft.py:279,:320).
// The data-consistency step of log_likelihood_gradient happens between the forward and the inverse column
// transform while the column tile is still in LDS, so the coil stack makes one HBM round trip fewer per pass.
#include <mutex>
#include <unordered_map>
#include <vector>
#include <cmath>

#include "fft_core.h"
#include "fft_ct.h"
#include "mrx_common.h"

#define MRX_FFT_NT 256
#define MRX_FFT_MAX_LEN 4096
#ifndef MRX_CT_UNROLL_COLS
#define MRX_CT_UNROLL_COLS 1
#endif
#define MRX_FFT_TILE_ELEMS 2048  // target complex elements staged per workgroup
#define MRX_T4_MAX_H 2048          // column-tiled DC pass: 9 H complex values of LDS per workgroup (144 KB at the limit)

// ------------------------------------------------------------------------------------------------------------
// plan cache (host)
// ------------------------------------------------------------------------------------------------------------
struct MrxFftEntry {
    MrxFftPlan plan;
    float2* d_tw;
};
static std::mutex g_plan_mu;
static std::unordered_map<long long, MrxFftEntry> g_plans;  // key: device * 2^32 + n

static int mrx_get_plan(int n, MrxFftEntry* out) {
    MRX_REQUIRE(n >= 1 && n <= MRX_FFT_MAX_LEN, MRX_EUNSUP, "FFT length %d outside [1, %d]", n, MRX_FFT_MAX_LEN);
    int dev = 0;
    MRX_HIP(hipGetDevice(&dev));
    const long long key = ((long long)dev << 32) | (long long)n;
    std::lock_guard<std::mutex> lk(g_plan_mu);
    auto it = g_plans.find(key);
    if (it != g_plans.end()) {
        *out = it->second;
        return MRX_OK;
    }
    MrxFftEntry e;
    MRX_REQUIRE(mrx_make_plan(n, &e.plan) == 0, MRX_EUNSUP, "cannot factor FFT length %d", n);
    std::vector<float2> tw(n);
    for (int m = 0; m < n; ++m) {
        const double a = -2.0 * M_PI * (double)m / (double)n;
        tw[m] = make_float2((float)cos(a), (float)sin(a));
    }
    MRX_HIP(hipMalloc((void**)&e.d_tw, sizeof(float2) * n));
    MRX_HIP(hipMemcpy(e.d_tw, tw.data(), sizeof(float2) * n, hipMemcpyHostToDevice));
    g_plans[key] = e;
    *out = e;
    return MRX_OK;
}

static inline float mrx_scale(int n, int inverse, int norm) {
    // fft.py:77-81,155-159: "backward" scales the inverse by 1/n, "forward" the forward, "ortho" both by 1/sqrt(n)
    if (norm == MRX_NORM_ORTHO) return (float)(1.0 / sqrt((double)n));
    if (norm == MRX_NORM_FORWARD) return inverse ? 1.0f : (float)(1.0 / (double)n);
    return inverse ? (float)(1.0 / (double)n) : 1.0f;  // backward / none
}

// ------------------------------------------------------------------------------------------------------------
// device: run all stages over the sequences held in LDS.  Returns the buffer holding the result.
// Two policies: PlanRT (runtime plan, any length) and PlanCT<N, radices...> (compile-time plan for the hot lengths).
// ------------------------------------------------------------------------------------------------------------
template <bool INV>
__device__ __forceinline__ float2* fft_lds_run(float2* a, float2* b, const float2* tw, const MrxFftPlan& p, int nseq,
                                               int seq_stride, int es, bool seq_fastest) {
    const float inv_nseq = 1.0f / (float)nseq;
    for (int s = 0; s < p.nstages; ++s) {
        const MrxFftStage& S = p.st[s];
        const int total = S.ips * nseq;
        for (int w = threadIdx.x; w < total; w += MRX_FFT_NT) {
            int seq, item;
            if (seq_fastest) {
                item = mrx_fdiv(w, inv_nseq);
                seq = w - item * nseq;
            } else {
                seq = mrx_fdiv(w, S.inv_ips);
                item = w - seq * S.ips;
            }
            mrx_fft_stage_item<INV>(a + seq * seq_stride, b + seq * seq_stride, tw, p.n, S, item, es);
        }
        __syncthreads();
        float2* t = a;
        a = b;
        b = t;
    }
    return a;
}

struct PlanRT {
    static constexpr bool kCT = false;
    static constexpr int N = 0;
};
template <int N_, int... Rs>
struct PlanCT {
    static constexpr bool kCT = true;
    static constexpr int N = N_;
};

// compile-time stage recursion.  COLS: sequences are columns (sequence index fastest, element stride NSEQ);
// otherwise rows (element stride 1, sequence stride N).
template <bool INV, int N, int NSEQ, bool COLS, int NS, int... Rs>
struct RunCT;
template <bool INV, int N, int NSEQ, bool COLS, int NS>
struct RunCT<INV, N, NSEQ, COLS, NS> {
    static __device__ __forceinline__ float2* run(float2* a, float2*, const float2*) { return a; }
};
template <bool INV, int N, int NSEQ, bool COLS, int NS, int R, int... Rest>
struct RunCT<INV, N, NSEQ, COLS, NS, R, Rest...> {
    static __device__ __forceinline__ float2* run(float2* a, float2* b, const float2* tw) {
        constexpr int SEQ_STRIDE = COLS ? 1 : N;
        constexpr int ES = COLS ? NSEQ : 1;
        constexpr int M = N / R;
        if constexpr (!mrx_ct_small(R) && NS == 1) {
            // in-register prime butterflies: output-pair subset `part` is wave-uniform, butterflies spread over lanes
            constexpr int QS = mrx_ct_qsplit(R);
            constexpr int NWAVES = MRX_FFT_NT / 64;
            static_assert(NWAVES % QS == 0, "q-split must divide the wave count");
            const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
            const int part = wave % QS;
            for (int bf = (threadIdx.x & 63) + 64 * (wave / QS); bf < M * NSEQ; bf += 64 * (NWAVES / QS)) {
                int seq, j;
                if (COLS) {
                    j = bf / NSEQ;
                    seq = bf - j * NSEQ;
                } else {
                    seq = bf / M;
                    j = bf - seq * M;
                }
                const float2* in = a + seq * SEQ_STRIDE;
                float2* out = b + seq * SEQ_STRIDE;
                if (part == 0) mrx_ct_prime_first<INV, N, R, 0>(in, out, j, ES);
                if constexpr (QS > 1) {
                    if (part == 1) mrx_ct_prime_first<INV, N, R, 1>(in, out, j, ES);
                }
                if constexpr (QS > 2) {
                    if (part == 2) mrx_ct_prime_first<INV, N, R, 2>(in, out, j, ES);
                    if (part == 3) mrx_ct_prime_first<INV, N, R, 3>(in, out, j, ES);
                }
            }
        } else {
            constexpr int IPS = mrx_ct_ips(N, R, NS);
            constexpr int TOTAL = IPS * NSEQ;
            if constexpr (COLS && MRX_CT_UNROLL_COLS) {
                // column tiles: the two or three items of a thread in flight together (their LDS reads overlap)
#pragma unroll
                for (int w0 = 0; w0 < TOTAL; w0 += MRX_FFT_NT) {
                    const int w = w0 + threadIdx.x;
                    if (w0 + MRX_FFT_NT <= TOTAL || w < TOTAL) {
                        const int item = w / NSEQ, seq = w - item * NSEQ;
                        mrx_ct_item<INV, N, R, NS>(a + seq * SEQ_STRIDE, b + seq * SEQ_STRIDE, tw, item, ES);
                    }
                }
            } else {
#pragma unroll 1
            for (int w = threadIdx.x; w < TOTAL; w += MRX_FFT_NT) {
                int seq, item;
                if (COLS) {
                    item = w / NSEQ;
                    seq = w - item * NSEQ;
                } else {
                    seq = w / IPS;
                    item = w - seq * IPS;
                }
                mrx_ct_item<INV, N, R, NS>(a + seq * SEQ_STRIDE, b + seq * SEQ_STRIDE, tw, item, ES);
            }
            }
        }
        __syncthreads();
        return RunCT<INV, N, NSEQ, COLS, NS * R, Rest...>::run(b, a, tw);
    }
};
template <bool INV, int NSEQ, bool COLS, class P>
struct RunPlan;
template <bool INV, int NSEQ, bool COLS, int N, int... Rs>
struct RunPlan<INV, NSEQ, COLS, PlanCT<N, Rs...>> {
    static __device__ __forceinline__ float2* run(float2* a, float2* b, const float2* tw) {
        return RunCT<INV, N, NSEQ, COLS, 1, Rs...>::run(a, b, tw);
    }
};

__device__ __forceinline__ int shifted(int p, int half, int n) {
    int g = p + half;
    return g >= n ? g - n : g;
}

struct RowArgs {
    MrxFftPlan plan;
    const float2* tw;
    int W, rpb, halfW;
    float scale;
    int C, H;  // images are [.., C][H rows]; expand mode reads the image x[b] for every coil c of batch element b
    int sdiv;  // expand mode: batch element b uses the maps of element b / sdiv (qMRI: echoes share one set of maps)
};

// soft data consistency fused into the expand pass (MODE 2): out = pred - where(mask, pred - ref, 0) * w - FFT(x S)   (vn_block.py:109-117)
struct RowDc {
    const float2* pred;
    const float2* ref;
    const float* w;
    MrxMask m;
};
// MODE 0: out[row] = FFT(in[row]);  MODE 1: out[b,c,h] = FFT(x[b,h] * S[b,c,h]);  MODE 2: MODE 1 + the data-consistency combination.
// NSEQ rows per workgroup (CT plans).
template <bool INV, int MODE, class P, int NSEQ>
__global__ __launch_bounds__(MRX_FFT_NT) void k_fft_rows(const float2* in, const float2* __restrict__ S, float2* out,
                                                         RowArgs a, RowDc dc) {
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    const int W = P::kCT ? P::N : a.W;
    const int RPB = P::kCT ? NSEQ : a.rpb;
    float2* tw = smem;
    float2* A = smem + W;
    float2* B = A + RPB * W;
    const long long img = blockIdx.x;  // image index (b*C + c in expand mode)
    const int hrow0 = blockIdx.y * RPB;
    const int nrows = min(RPB, a.H - hrow0);
    const long long row0 = img * a.H + hrow0;
    const long long bimg = MODE >= 1 ? img / a.C : 0;  // one division per workgroup
    const long long srow0 = MODE >= 1 ? ((bimg / a.sdiv) * a.C + (img - bimg * a.C)) * a.H + hrow0 : 0;
    const float invW = 1.0f / (float)W;
    for (int i = threadIdx.x; i < W; i += MRX_FFT_NT) tw[i] = a.tw[i];
    for (int idx = threadIdx.x; idx < RPB * W; idx += MRX_FFT_NT) {
        const int r = P::kCT ? idx / W : mrx_fdiv(idx, invW), x = idx - r * W;
        const int g = shifted(x, a.halfW, W);
        float2 v = make_float2(0.f, 0.f);
        if (r < nrows) {
            const long long ro = row0 + r;
            if (MODE == 0) {
                v = in[ro * W + g];
            } else {
                const float2 e = in[(bimg * a.H + hrow0 + r) * W + g];
                const float2 s = S[(srow0 + r) * W + g];
                v = make_float2(e.x * s.x - e.y * s.y, e.x * s.y + e.y * s.x);  // utils.py:115-116
            }
        }
        A[idx] = v;
    }
    __syncthreads();
    float2* res;
    if constexpr (P::kCT)
        res = RunPlan<INV, NSEQ, false, P>::run(A, B, tw);
    else
        res = fft_lds_run<INV>(A, B, tw, a.plan, RPB, W, 1, false);
    for (int idx = threadIdx.x; idx < nrows * W; idx += MRX_FFT_NT) {
        const int r = P::kCT ? idx / W : mrx_fdiv(idx, invW), x = idx - r * W;
        const int g = shifted(x, a.halfW, W);
        float2 v = res[idx];
        if (MODE == 2) {  // same operations as k_soft_dc<1> on the separately written transform
            const long long o = (row0 + r) * W + g;
            const float ex = __fmul_rn(v.x, a.scale), ey = __fmul_rn(v.y, a.scale);
            const float2 p = dc.pred[o];
            float dx = 0.f, dy = 0.f;
            if (mrx_mask_true(dc.m, bimg, img - bimg * a.C, hrow0 + r, g)) {
                const float2 q = dc.ref[o];
                dx = __fsub_rn(p.x, q.x);
                dy = __fsub_rn(p.y, q.y);
            }
            const float w8 = dc.w[0];
            dx = __fmul_rn(dx, w8);
            dy = __fmul_rn(dy, w8);
            out[o] = make_float2(__fsub_rn(__fsub_rn(p.x, dx), ex), __fsub_rn(__fsub_rn(p.y, dy), ey));
        } else {
            out[(row0 + r) * W + g] = make_float2(v.x * a.scale, v.y * a.scale);
        }
    }
}

struct ColArgs {
    MrxFftPlan plan;
    const float2* tw;
    int H, W, ct, halfH;
    float scale;   // applied after the (first) transform
    float scale2;  // DC kernel: applied after the inverse transform
    int C;         // DC kernel: images are [B][C]
    int ntx;       // column tiles per image
    long long nblocks;
};

// workgroup id -> (image, column tile).  Workgroup b runs on XCD b % 8: hand every XCD a contiguous band of tiles so
// the 128-byte lines shared by neighbouring column tiles are fetched into one L2 only (speed, not correctness).
__device__ __forceinline__ void col_tile(const ColArgs& a, long long& img, int& w0) {
    // 4-column tiles use 32 B of every 64-B request: the neighbour tile must hit the same L2 (PMC: FETCH_SIZE was 2x the input)
    const long long t = mrx_xcd_band(blockIdx.x, a.nblocks);
    img = t / a.ntx;
    w0 = (int)(t - img * a.ntx) * a.ct;
}

template <bool INV, class P, int NSEQ>
__global__ __launch_bounds__(MRX_FFT_NT) void k_fft_cols(const float2* in, float2* out, ColArgs a) {
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    const int H = P::kCT ? P::N : a.H, W = a.W, CT = P::kCT ? NSEQ : a.ct;
    const float invCT = 1.0f / (float)CT;
    float2* tw = smem;
    float2* A = smem + H;
    float2* B = A + H * CT;
    long long im;
    int w0;
    col_tile(a, im, w0);
    const long long img = im * H * W;
    for (int i = threadIdx.x; i < H; i += MRX_FFT_NT) tw[i] = a.tw[i];
    for (int idx = threadIdx.x; idx < H * CT; idx += MRX_FFT_NT) {
        const int p = P::kCT ? idx / CT : mrx_fdiv(idx, invCT), c = idx - p * CT;
        const int w = w0 + c;
        A[idx] = w < W ? in[img + (long long)shifted(p, a.halfH, H) * W + w] : make_float2(0.f, 0.f);
    }
    __syncthreads();
    float2* res;
    if constexpr (P::kCT)
        res = RunPlan<INV, NSEQ, true, P>::run(A, B, tw);
    else
        res = fft_lds_run<INV>(A, B, tw, a.plan, CT, 1, CT, true);
    for (int idx = threadIdx.x; idx < H * CT; idx += MRX_FFT_NT) {
        const int p = P::kCT ? idx / CT : mrx_fdiv(idx, invCT), c = idx - p * CT;
        const int w = w0 + c;
        if (w < W) {
            float2 v = res[idx];
            out[img + (long long)shifted(p, a.halfH, H) * W + w] = make_float2(v.x * a.scale, v.y * a.scale);
        }
    }
}

// forward column FFT -> mask * (k - y) -> inverse column FFT, all in LDS (rim_utils.py:51-58)
template <class P, int NSEQ>
__global__ __launch_bounds__(MRX_FFT_NT) void k_cols_dc(const float2* in, const float2* __restrict__ y, MrxMask mask,
                                                        float2* out, ColArgs a) {
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    const int H = P::kCT ? P::N : a.H, W = a.W, CT = P::kCT ? NSEQ : a.ct;
    const float invCT = 1.0f / (float)CT;
    float2* tw = smem;
    float2* A = smem + H;
    float2* B = A + H * CT;
    long long bc;
    int w0;
    col_tile(a, bc, w0);
    const long long b = bc / a.C, c_ = bc - b * a.C;
    const long long img = bc * H * W;
    for (int i = threadIdx.x; i < H; i += MRX_FFT_NT) tw[i] = a.tw[i];
    for (int idx = threadIdx.x; idx < H * CT; idx += MRX_FFT_NT) {
        const int p = P::kCT ? idx / CT : mrx_fdiv(idx, invCT), c = idx - p * CT;
        const int w = w0 + c;
        A[idx] = w < W ? in[img + (long long)shifted(p, a.halfH, H) * W + w] : make_float2(0.f, 0.f);
    }
    __syncthreads();
    float2* res;
    if constexpr (P::kCT)
        res = RunPlan<false, NSEQ, true, P>::run(A, B, tw);
    else
        res = fft_lds_run<false>(A, B, tw, a.plan, CT, 1, CT, true);
    float2* oth = (res == A) ? B : A;
    for (int idx = threadIdx.x; idx < H * CT; idx += MRX_FFT_NT) {
        const int p = P::kCT ? idx / CT : mrx_fdiv(idx, invCT), c = idx - p * CT;
        const int w = w0 + c;
        float2 r = make_float2(0.f, 0.f);
        if (w < W) {
            const int hk = shifted(p, a.halfH, H);
            const float2 k = res[idx];
            const float2 yv = y[img + (long long)hk * W + w];
            const float m = mrx_mask_val(mask, b, c_, hk, w);
            r = make_float2(m * (k.x * a.scale - yv.x), m * (k.y * a.scale - yv.y));  // rim_utils.py:54
        }
        res[idx] = r;
    }
    __syncthreads();
    float2* res2;
    if constexpr (P::kCT)
        res2 = RunPlan<true, NSEQ, true, P>::run(res, oth, tw);
    else
        res2 = fft_lds_run<true>(res, oth, tw, a.plan, CT, 1, CT, true);
    for (int idx = threadIdx.x; idx < H * CT; idx += MRX_FFT_NT) {
        const int p = P::kCT ? idx / CT : mrx_fdiv(idx, invCT), c = idx - p * CT;
        const int w = w0 + c;
        if (w < W) {
            float2 v = res2[idx];
            out[img + (long long)shifted(p, a.halfH, H) * W + w] = make_float2(v.x * a.scale2, v.y * a.scale2);
        }
    }
}

// ---- column-tiled ("t4") coil stack: [B*C][W/4][H][4] complex -----------------------------------------------------------------------
// Every 4-column tile of an image is one contiguous block of H * 32 bytes.  The W = 372 prime-factor row kernels write / read this layout
// (llg372.hip: mrx_pfa372_expand_t4 / mrx_pfa372_reduce_t4) and the measured data is laid out once per slice (mrx_tile4_cols), so the
// column pass of the general-mask gradient moves whole contiguous blocks with 16-byte accesses instead of 32-byte pieces of 2976-byte
// rows.  Arithmetic and operation order are those of k_cols_dc (bit-identical results).
// LDS: A | B (H x 4 complex each, 16-byte aligned) | twiddles.
// (Measured and not kept: two tiles per workgroup with the next tile's input and the measured data prefetched into registers -- 34.7 us
// against 29.4 us for this form at 15 x 640 x 372: the pass is bound by the LDS butterflies of its eight stages, not by its loads, and
// 1395 short workgroups fill the stage bubbles of one another better than 698 long ones.)
// NOY: the measured data left out -- IFFT_H(m FFT_H(x)).  By linearity the gradient is A^H M A eta - A^H M y, and the second term is one constant
// image per slice that the caller keeps as one more partial plane for the first RIM layer's loader (ops.llg, parts form): the pass reads
// 28.6 MB less per step at 15 x 640 x 372.
template <class P, bool NOY = false>
__global__ __launch_bounds__(MRX_FFT_NT) void k_cols_dc_t4(const float4* in, const float4* __restrict__ y4, MrxMask mask, float4* out,
                                                           ColArgs a) {
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    const int H = P::kCT ? P::N : a.H;
    float2* A = smem;
    float2* B = A + 4 * H;
    float2* tw = B + 4 * H;
    const long long tile = blockIdx.x;                 // (image, column tile): private to this workgroup
    const long long bc = tile / a.ntx;
    const int w0 = (int)(tile - bc * a.ntx) * 4;
    const long long b = bc / a.C, c_ = bc - b * a.C;
    const float4* src = in + tile * (2ll * H);         // two float4 per tile row
    const float4* ysrc = y4 + tile * (2ll * H);
    float4* dst = out + tile * (2ll * H);
    // Compile-time H: the tile, the twiddles and the mask bits of this thread are requested up front (clamped addresses, no branch around a load) --
    // as run-time loops every load sat in its own basic block with its wait: five memory round trips for the tile, three for the twiddles and
    // ten for the mask (two per row, in the middle of the pass) one after the other.
    constexpr int NIT = P::kCT ? (2 * P::N + MRX_FFT_NT - 1) / MRX_FFT_NT : 1, NTW = P::kCT ? (P::N + MRX_FFT_NT - 1) / MRX_FFT_NT : 1;
    unsigned mraw[NIT][2];
    if constexpr (P::kCT) {
        float4 tv[NIT];
        float2 twv[NTW];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = min((int)threadIdx.x + it * MRX_FFT_NT, 2 * H - 1), p = idx >> 1, hf = idx & 1;
            tv[it] = src[shifted(p, a.halfH, H) * 2 + hf];
        }
#pragma unroll
        for (int it = 0; it < NTW; ++it) twv[it] = a.tw[min((int)threadIdx.x + it * MRX_FFT_NT, H - 1)];
        {
            const bool u8 = mask.kind == MRX_MASK_U8;            // (uniform: the two forms of the loop differ in the width of the load only)
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                const int idx = min((int)threadIdx.x + it * MRX_FFT_NT, 2 * H - 1), p = idx >> 1, hf = idx & 1;
                const long long off = b * mask.s[0] + c_ * mask.s[1] + (long long)shifted(p, a.halfH, H) * mask.s[2] + (long long)(w0 + 2 * hf) * mask.s[3];
                if (u8) {
                    mraw[it][0] = ((const unsigned char*)mask.p)[off];
                    mraw[it][1] = ((const unsigned char*)mask.p)[off + mask.s[3]];
                } else {
                    mraw[it][0] = ((const unsigned*)mask.p)[off];
                    mraw[it][1] = ((const unsigned*)mask.p)[off + mask.s[3]];
                }
            }
        }
        // (unconditional LDS writes at the clamped index -- lanes past the end repeat the last element's own value: a load whose only use sits
        // behind `if (i < H)` is sunk into that block by the compiler, wait included)
#pragma unroll
        for (int it = 0; it < NIT; ++it) reinterpret_cast<float4*>(A)[min((int)threadIdx.x + it * MRX_FFT_NT, 2 * H - 1)] = tv[it];
#pragma unroll
        for (int it = 0; it < NTW; ++it) tw[min((int)threadIdx.x + it * MRX_FFT_NT, H - 1)] = twv[it];
    } else {
        for (int idx = threadIdx.x; idx < 2 * H; idx += MRX_FFT_NT) {
            const int p = idx >> 1, hf = idx & 1;
            reinterpret_cast<float4*>(A)[idx] = src[shifted(p, a.halfH, H) * 2 + hf];
        }
        for (int i = threadIdx.x; i < H; i += MRX_FFT_NT) tw[i] = a.tw[i];
    }
    __syncthreads();
    float2* res;
    if constexpr (P::kCT)
        res = RunPlan<false, 4, true, P>::run(A, B, tw);
    else
        res = fft_lds_run<false>(A, B, tw, a.plan, 4, 1, 4, true);
    float2* oth = (res == A) ? B : A;
    auto dc_row = [&](int idx, float m0, float m1) {
        const int p = idx >> 1, hf = idx & 1;
        const int hk = shifted(p, a.halfH, H);
        const float4 k = reinterpret_cast<float4*>(res)[idx];
        const float4 yv = NOY ? make_float4(0.f, 0.f, 0.f, 0.f) : ysrc[hk * 2 + hf];
        reinterpret_cast<float4*>(res)[idx] = make_float4(m0 * (k.x * a.scale - yv.x), m0 * (k.y * a.scale - yv.y),   // rim_utils.py:54
                                                          m1 * (k.z * a.scale - yv.z), m1 * (k.w * a.scale - yv.w));
    };
    if constexpr (P::kCT) {
        const bool u8 = mask.kind == MRX_MASK_U8;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int idx = (int)threadIdx.x + it * MRX_FFT_NT;
            if (idx < 2 * H) dc_row(idx, u8 ? (float)mraw[it][0] : __uint_as_float(mraw[it][0]), u8 ? (float)mraw[it][1] : __uint_as_float(mraw[it][1]));
        }
    } else {
        for (int idx = threadIdx.x; idx < 2 * H; idx += MRX_FFT_NT) {
            const int hk = shifted(idx >> 1, a.halfH, H), w = w0 + 2 * (idx & 1);
            dc_row(idx, mrx_mask_val(mask, b, c_, hk, w), mrx_mask_val(mask, b, c_, hk, w + 1));
        }
    }
    __syncthreads();
    float2* res2;
    if constexpr (P::kCT)
        res2 = RunPlan<true, 4, true, P>::run(res, oth, tw);
    else
        res2 = fft_lds_run<true>(res, oth, tw, a.plan, 4, 1, 4, true);
    for (int idx = threadIdx.x; idx < 2 * H; idx += MRX_FFT_NT) {
        const int p = idx >> 1, hf = idx & 1;
        const float4 v = reinterpret_cast<float4*>(res2)[idx];
        dst[shifted(p, a.halfH, H) * 2 + hf] = make_float4(v.x * a.scale2, v.y * a.scale2, v.z * a.scale2, v.w * a.scale2);
    }
}

// ---- H = 640 on the column-tiled coil stack: one WAVEFRONT per 4-column tile, register transforms (640 = 10 x 8 x 8) -------------------
// n = 64 n1 + 8 n2 + n3, k = k1 + 10 k2 + 80 k3:  W640^(nk) = W10^(n1 k1) W640^((8 n2 + n3) k1) W8^(n2 k2) W64^(n3 k2) W8^(n3 k3).
//   A : lane m = 8 n2 + n3 holds rows 64 n1 + m of all four columns (ten contiguous 2-KB loads per wave): four 10-point DFTs (2 x 5
//       prime-factor form), twiddles W640^(m k1);
//   B : 320 (k1, n3, column) 8-point DFTs over n2, five per lane, twiddles W64^(n3 k2);
//   C : 320 (k1, k2, column) 8-point DFTs over n3, five per lane -> rows k1 + 10 k2 + 80 k3, where the data-consistency step and the
//       inverse 8-point DFT over k3 happen in the same registers; B', A' mirror B and A with conjugate twiddles.
// The four exchanges go through 23 KB of wave-private LDS (every lane reads all its inputs of a stage before anything is written back); three
// stages per direction instead of four, no workgroup-wide staging, the measured data requested up front.  Same results as k_cols_dc_t4 to
// fp32 round-off (different operation order).  Selectable (MRX_COLS640=1), not the default: see mrx_llg_cols_dc_t4.
#define C640_LDS_C2 (80 * 36)
template <bool INV>
__device__ __forceinline__ void c640_dft5(mrx_c32 (&a)[5]) {
    const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
    const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
    const mrx_c32 p1 = mrx_add(a[1], a[4]), m1 = mrx_sub(a[1], a[4]), p2 = mrx_add(a[2], a[3]), m2 = mrx_sub(a[2], a[3]);
    const mrx_c32 R1 = mrx_mk(a[0].x + c1 * p1.x + c2 * p2.x, a[0].y + c1 * p1.y + c2 * p2.y);
    const mrx_c32 R2 = mrx_mk(a[0].x + c2 * p1.x + c1 * p2.x, a[0].y + c2 * p1.y + c1 * p2.y);
    const mrx_c32 I1 = mrx_rot<INV>(mrx_mk(s1 * m1.x + s2 * m2.x, s1 * m1.y + s2 * m2.y));
    const mrx_c32 I2 = mrx_rot<INV>(mrx_mk(s2 * m1.x - s1 * m2.x, s2 * m1.y - s1 * m2.y));
    a[0] = mrx_add(a[0], mrx_add(p1, p2));
    a[1] = mrx_add(R1, I1);
    a[2] = mrx_add(R2, I2);
    a[3] = mrx_sub(R2, I2);
    a[4] = mrx_sub(R1, I1);
}
// 10-point DFT, twiddle-free 2 x 5 form: n1 = (5 a + 2 b) mod 10, k1 = (5 ka + 6 kb) mod 10
template <bool INV>
__device__ __forceinline__ void c640_dft10(mrx_c32 (&v)[10]) {
    mrx_c32 e[5], o[5];
#pragma unroll
    for (int b = 0; b < 5; ++b) {
        e[b] = mrx_add(v[(2 * b) % 10], v[(2 * b + 5) % 10]);
        o[b] = mrx_sub(v[(2 * b) % 10], v[(2 * b + 5) % 10]);
    }
    c640_dft5<INV>(e);
    c640_dft5<INV>(o);
#pragma unroll
    for (int kb = 0; kb < 5; ++kb) {
        v[(6 * kb) % 10] = e[kb];
        v[(5 + 6 * kb) % 10] = o[kb];
    }
}
template <bool INV>
__device__ __forceinline__ void c640_dft8(mrx_c32 (&a)[8]) {
    mrx_c32 e0, e1, e2, e3, o0, o1, o2, o3;
    mrx_dft4<INV>(a[0], a[2], a[4], a[6], e0, e1, e2, e3);
    mrx_dft4<INV>(a[1], a[3], a[5], a[7], o0, o1, o2, o3);
    const float h = 0.70710678118654752440f;
    const mrx_c32 r1 = mrx_rot<INV>(o1);
    const mrx_c32 t1 = mrx_mk(h * (o1.x + r1.x), h * (o1.y + r1.y));
    const mrx_c32 t2 = mrx_rot<INV>(o2);
    const mrx_c32 r3 = mrx_rot<INV>(o3);
    const mrx_c32 t3 = mrx_mk(h * (r3.x - o3.x), h * (r3.y - o3.y));
    a[0] = mrx_add(e0, o0), a[4] = mrx_sub(e0, o0);
    a[1] = mrx_add(e1, t1), a[5] = mrx_sub(e1, t1);
    a[2] = mrx_add(e2, t2), a[6] = mrx_sub(e2, t2);
    a[3] = mrx_add(e3, t3), a[7] = mrx_sub(e3, t3);
}
__device__ __forceinline__ mrx_c32 c640_cmulc(mrx_c32 a, mrx_c32 w) { return mrx_mk(a.x * w.x + a.y * w.y, a.y * w.x - a.x * w.y); }   // a * conj(w)

__global__ __launch_bounds__(64) void k_cols640_dc_t4(const float4* in, const float4* __restrict__ y4, MrxMask mask, float4* out, ColArgs a) {
    __shared__ __attribute__((aligned(16))) float2 E[C640_LDS_C2];
    const int l = threadIdx.x;
    const long long tile = blockIdx.x;
    const long long bc = tile / a.ntx;
    const int w0 = (int)(tile - bc * a.ntx) * 4;
    const long long b = bc / a.C, c_ = bc - b * a.C;
    const float4* src = in + tile * 1280;
    const float2* ysrc = reinterpret_cast<const float2*>(y4 + tile * 1280);
    float4* dst = out + tile * 1280;
    const int n3B = (l >> 2) & 7, cB = l & 3, hiB = l >> 5;     // stages B / B': tasks (k1 = 2 s + hiB, n3B, cB)
    const int kq = l >> 2;                                      // stages C / C': tasks (kk = 16 s + kq = k1 + 10 k2, cB)
    // stage A inputs, the measured data of stage C and the twiddles: everything requested up front
    mrx_c32 v[4][10];
#pragma unroll
    for (int n1 = 0; n1 < 10; ++n1) {
        const int g = shifted(64 * n1 + l, a.halfH, 640);
        const float4 q0 = src[g * 2], q1 = src[g * 2 + 1];
        v[0][n1] = mrx_mk(q0.x, q0.y), v[1][n1] = mrx_mk(q0.z, q0.w), v[2][n1] = mrx_mk(q1.x, q1.y), v[3][n1] = mrx_mk(q1.z, q1.w);
    }
    mrx_c32 yv[5][8];
#pragma unroll
    for (int s = 0; s < 5; ++s)
#pragma unroll
        for (int k3 = 0; k3 < 8; ++k3) yv[s][k3] = ysrc[shifted(16 * s + kq + 80 * k3, a.halfH, 640) * 4 + cB];
    mrx_c32 t1[10], t2[8];
#pragma unroll
    for (int k1 = 1; k1 < 10; ++k1) t1[k1] = a.tw[l * k1];
#pragma unroll
    for (int k2 = 1; k2 < 8; ++k2) t2[k2] = a.tw[10 * n3B * k2];
    // ---- A ----
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        c640_dft10<false>(v[c]);
#pragma unroll
        for (int k1 = 1; k1 < 10; ++k1) v[c][k1] = mrx_cmul(v[c][k1], t1[k1]);
    }
#pragma unroll
    for (int k1 = 0; k1 < 10; ++k1) {
        reinterpret_cast<float4*>(E)[(k1 * 64 + l) * 2] = make_float4(v[0][k1].x, v[0][k1].y, v[1][k1].x, v[1][k1].y);
        reinterpret_cast<float4*>(E)[(k1 * 64 + l) * 2 + 1] = make_float4(v[2][k1].x, v[2][k1].y, v[3][k1].x, v[3][k1].y);
    }
    __syncthreads();
    // ---- B ----
    mrx_c32 u[5][8];
#pragma unroll
    for (int s = 0; s < 5; ++s)
#pragma unroll
        for (int n2 = 0; n2 < 8; ++n2) u[s][n2] = E[((2 * s + hiB) * 64 + 8 * n2 + n3B) * 4 + cB];
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        c640_dft8<false>(u[s]);
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) E[((2 * s + hiB) + 10 * k2) * 36 + n3B * 4 + cB] = k2 ? mrx_cmul(u[s][k2], t2[k2]) : u[s][k2];
    }
    __syncthreads();
    // ---- C, data consistency, C' ----
    // (the mask values of this lane's 40 outputs, requested together: the kind is decided once, not per element)
    float mk[5][8];
    {
        const long long mb = b * mask.s[0] + c_ * mask.s[1] + (long long)(w0 + cB) * mask.s[3];
        if (mask.kind == MRX_MASK_U8) {
#pragma unroll
            for (int s = 0; s < 5; ++s)
#pragma unroll
                for (int k3 = 0; k3 < 8; ++k3)
                    mk[s][k3] = (float)((const unsigned char*)mask.p)[mb + (long long)shifted(16 * s + kq + 80 * k3, a.halfH, 640) * mask.s[2]];
        } else {
#pragma unroll
            for (int s = 0; s < 5; ++s)
#pragma unroll
                for (int k3 = 0; k3 < 8; ++k3)
                    mk[s][k3] = ((const float*)mask.p)[mb + (long long)shifted(16 * s + kq + 80 * k3, a.halfH, 640) * mask.s[2]];
        }
    }
#pragma unroll
    for (int s = 0; s < 5; ++s)
#pragma unroll
        for (int n3 = 0; n3 < 8; ++n3) u[s][n3] = E[(16 * s + kq) * 36 + n3 * 4 + cB];
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        c640_dft8<false>(u[s]);
#pragma unroll
        for (int k3 = 0; k3 < 8; ++k3) {
            const float m = mk[s][k3];
            u[s][k3] = mrx_mk(m * (u[s][k3].x * a.scale - yv[s][k3].x), m * (u[s][k3].y * a.scale - yv[s][k3].y));   // rim_utils.py:54
        }
        c640_dft8<true>(u[s]);
#pragma unroll
        for (int n3 = 0; n3 < 8; ++n3) E[(16 * s + kq) * 36 + n3 * 4 + cB] = u[s][n3];
    }
    __syncthreads();
    // ---- B' ----
#pragma unroll
    for (int s = 0; s < 5; ++s)
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) {
            const mrx_c32 q = E[((2 * s + hiB) + 10 * k2) * 36 + n3B * 4 + cB];
            u[s][k2] = k2 ? c640_cmulc(q, t2[k2]) : q;
        }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        c640_dft8<true>(u[s]);
#pragma unroll
        for (int n2 = 0; n2 < 8; ++n2) E[((2 * s + hiB) * 64 + 8 * n2 + n3B) * 4 + cB] = u[s][n2];
    }
    __syncthreads();
    // ---- A' ----
#pragma unroll
    for (int k1 = 0; k1 < 10; ++k1) {
        const float4 q0 = reinterpret_cast<float4*>(E)[(k1 * 64 + l) * 2], q1 = reinterpret_cast<float4*>(E)[(k1 * 64 + l) * 2 + 1];
        v[0][k1] = mrx_mk(q0.x, q0.y), v[1][k1] = mrx_mk(q0.z, q0.w), v[2][k1] = mrx_mk(q1.x, q1.y), v[3][k1] = mrx_mk(q1.z, q1.w);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int k1 = 1; k1 < 10; ++k1) v[c][k1] = c640_cmulc(v[c][k1], t1[k1]);
        c640_dft10<true>(v[c]);
    }
#pragma unroll
    for (int n1 = 0; n1 < 10; ++n1) {
        const int g = shifted(64 * n1 + l, a.halfH, 640);
        dst[g * 2] = make_float4(v[0][n1].x * a.scale2, v[0][n1].y * a.scale2, v[1][n1].x * a.scale2, v[1][n1].y * a.scale2);
        dst[g * 2 + 1] = make_float4(v[2][n1].x * a.scale2, v[2][n1].y * a.scale2, v[3][n1].x * a.scale2, v[3][n1].y * a.scale2);
    }
}

// row-major [nimg][H][W] complex -> column-tiled [nimg][W/4][H][4]  (W % 4 == 0; once per slice for the measured data)
__global__ void k_tile4_cols(const float2* __restrict__ in, float2* __restrict__ out, long long nimg, int H, int W) {
    const long long total = nimg * H * W;
    const int ntx = W >> 2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long row = i / W;                   // (image, h)
        const int w = (int)(i - row * W);
        const long long img = row / H;
        const int h = (int)(row - img * H);
        out[((img * ntx + (w >> 2)) * H + h) * 4 + (w & 3)] = in[i];
    }
}

struct ReduceArgs {
    MrxFftPlan plan;
    const float2* tw;
    int sdiv;  // batch element b uses the maps of element b / sdiv
    int C, H, W, g, halfW;
    float scale;
    float post;  // llg: 1/sigma^2
};

// inverse row FFT of every coil row (b, :, h), times conj(S), summed over coils (rim_utils.py:59-62, vn_block.py:87).
// OUT 0: out[b,h,w] complex.  OUT 1: out4[b,0:4,h,w] = (eta_re, eta_im, g_re, g_im) (rim_utils.py:67).
template <int OUT, class P, int NSEQ>
__global__ __launch_bounds__(MRX_FFT_NT) void k_rows_reduce(const float2* __restrict__ k, const float2* __restrict__ S,
                                                            const float2* __restrict__ eta, float* __restrict__ out,
                                                            ReduceArgs a) {
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    const int W = P::kCT ? P::N : a.W, H = a.H, C = a.C, G = P::kCT ? NSEQ : a.g;
    float2* tw = smem;
    float2* acc = smem + W;
    float2* A = acc + W;
    float2* B = A + G * W;
    const int h = blockIdx.x;
    const long long b = blockIdx.y;
    const float invW = 1.0f / (float)W;
    for (int i = threadIdx.x; i < W; i += MRX_FFT_NT) {
        tw[i] = a.tw[i];
        acc[i] = make_float2(0.f, 0.f);
    }
    for (int c0 = 0; c0 < C; c0 += G) {
        const int nrows = min(G, C - c0);
        __syncthreads();
        for (int idx = threadIdx.x; idx < G * W; idx += MRX_FFT_NT) {
            const int r = P::kCT ? idx / W : mrx_fdiv(idx, invW), x = idx - r * W;
            A[idx] = r < nrows ? k[(((b * C + c0 + r) * H) + h) * W + shifted(x, a.halfW, W)] : make_float2(0.f, 0.f);
        }
        __syncthreads();
        float2* res;
        if constexpr (P::kCT)
            res = RunPlan<true, NSEQ, false, P>::run(A, B, tw);
        else
            res = fft_lds_run<true>(A, B, tw, a.plan, G, W, 1, false);
        for (int x = threadIdx.x; x < W; x += MRX_FFT_NT) {
            const int g = shifted(x, a.halfW, W);
            float2 s_acc = acc[x];
            for (int r = 0; r < nrows; ++r) {
                float2 v = res[r * W + x];
                v.x *= a.scale;
                v.y *= a.scale;
                const float2 s = S[((((b / a.sdiv) * C + c0 + r) * H) + h) * W + g];
                s_acc.x += v.x * s.x + v.y * s.y;  // re: rim_utils.py:61
                s_acc.y += v.y * s.x - v.x * s.y;  // im: rim_utils.py:62
            }
            acc[x] = s_acc;
        }
    }
    __syncthreads();
    for (int x = threadIdx.x; x < W; x += MRX_FFT_NT) {
        const int g = shifted(x, a.halfW, W);
        const float2 v = acc[x];
        if (OUT == 0) {
            ((float2*)out)[(b * H + h) * W + g] = v;
        } else {
            const float2 e = eta[(b * H + h) * W + g];
            const long long plane = (long long)H * W;
            float* o = out + b * 4 * plane + (long long)h * W + g;
            o[0] = e.x;
            o[plane] = e.y;
            o[2 * plane] = v.x * a.post;
            o[3 * plane] = v.y * a.post;
        }
    }
}

// log_likelihood_gradient for masks that do not depend on the row index h (mstride_h == 0: 1-D column masks).  Then
//   IFFT_H( m(w) * (FFT_H(FFT_W(eta S)) - y) ) = m(w) * (FFT_W(eta S) - IFFT_H(y))
// (the H transforms cancel for every normalisation, centred or not), so with yt = IFFT_H(y) precomputed once the whole
// step is row transforms only and fuses into ONE kernel per (b, h) row: eta*S -> FFT_W -> m*(. - yt) -> IFFT_W ->
// sum_c conj(S) -> /sigma^2.  HBM traffic = the algorithmic (25 + 16 C) H W bytes; no work buffer.
// PART = true: one workgroup per (row, coil chunk) -- blockIdx.z is the chunk -- writing the un-scaled partial coil sum to
// part[chunk][b][h][w]; k_llg_combine adds the chunks.  Gives 3x the workgroups at C = 15 (1920 instead of 640 at B = 1), which
// is what a latency-bound kernel needs on 256 CUs.
template <class P, int NSEQ, bool PART>
__global__ __launch_bounds__(MRX_FFT_NT) void k_llg_rows_hinv(const float2* __restrict__ eta, const float2* __restrict__ yt,
                                                              const float2* __restrict__ S, MrxMask mask,
                                                              float* __restrict__ out, ReduceArgs a, float scale_f) {
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    const int W = P::kCT ? P::N : a.W, H = a.H, C = a.C, G = P::kCT ? NSEQ : a.g;
    float2* tw = smem;
    float2* acc = smem + W;
    float2* E = acc + W;
    float2* Sb = E + W;
    float2* A = Sb + G * W;
    float2* B = A + G * W;
    const int h = blockIdx.x;
    const long long b = blockIdx.y;
    const float invW = 1.0f / (float)W;
    for (int i = threadIdx.x; i < W; i += MRX_FFT_NT) {
        tw[i] = a.tw[i];
        acc[i] = make_float2(0.f, 0.f);
        E[i] = eta[(b * H + h) * W + shifted(i, a.halfW, W)];
    }
    const int c_begin = PART ? (int)blockIdx.z * G : 0;
    const int c_end = PART ? min(C, c_begin + G) : C;
    for (int c0 = c_begin; c0 < c_end; c0 += G) {
        const int nrows = min(G, C - c0);
        __syncthreads();
        for (int idx = threadIdx.x; idx < G * W; idx += MRX_FFT_NT) {
            const int r = P::kCT ? idx / W : mrx_fdiv(idx, invW), x = idx - r * W;
            float2 s = make_float2(0.f, 0.f), v = make_float2(0.f, 0.f);
            if (r < nrows) {
                s = S[((((b / a.sdiv) * C + c0 + r) * H) + h) * W + shifted(x, a.halfW, W)];
                const float2 e = E[x];
                v = make_float2(e.x * s.x - e.y * s.y, e.x * s.y + e.y * s.x);  // rim_utils.py:47-48
            }
            Sb[idx] = s;
            A[idx] = v;
        }
        __syncthreads();
        float2* res;
        if constexpr (P::kCT)
            res = RunPlan<false, NSEQ, false, P>::run(A, B, tw);
        else
            res = fft_lds_run<false>(A, B, tw, a.plan, G, W, 1, false);
        float2* oth = (res == A) ? B : A;
        for (int idx = threadIdx.x; idx < G * W; idx += MRX_FFT_NT) {
            const int r = P::kCT ? idx / W : mrx_fdiv(idx, invW), x = idx - r * W;
            float2 d = make_float2(0.f, 0.f);
            if (r < nrows) {
                const int g = shifted(x, a.halfW, W);
                const float2 k = res[idx];
                const float2 yv = yt[(((b * C + c0 + r) * H) + h) * W + g];
                const float m = mrx_mask_val(mask, b, c0 + r, 0, g);
                d = make_float2(m * (k.x * scale_f - yv.x), m * (k.y * scale_f - yv.y));  // rim_utils.py:54 in hybrid space
            }
            res[idx] = d;
        }
        __syncthreads();
        float2* res2;
        if constexpr (P::kCT)
            res2 = RunPlan<true, NSEQ, false, P>::run(res, oth, tw);
        else
            res2 = fft_lds_run<true>(res, oth, tw, a.plan, G, W, 1, false);
        for (int x = threadIdx.x; x < W; x += MRX_FFT_NT) {
            float2 s_acc = acc[x];
            for (int r = 0; r < nrows; ++r) {
                float2 v = res2[r * W + x];
                v.x *= a.scale;
                v.y *= a.scale;
                const float2 s = Sb[r * W + x];
                s_acc.x += v.x * s.x + v.y * s.y;  // rim_utils.py:61
                s_acc.y += v.y * s.x - v.x * s.y;  // rim_utils.py:62
            }
            acc[x] = s_acc;
        }
    }
    __syncthreads();
    const long long plane = (long long)H * W;
    if (PART) {
        float2* po = reinterpret_cast<float2*>(out) + ((long long)blockIdx.z * gridDim.y + b) * plane + (long long)h * W;
        for (int x = threadIdx.x; x < W; x += MRX_FFT_NT) po[shifted(x, a.halfW, W)] = acc[x];
        return;
    }
    for (int x = threadIdx.x; x < W; x += MRX_FFT_NT) {
        const int g = shifted(x, a.halfW, W);
        const float2 v = acc[x];
        const float2 e = E[x];
        float* o = out + b * 4 * plane + (long long)h * W + g;
        o[0] = e.x;
        o[plane] = e.y;
        o[2 * plane] = v.x * a.post;
        o[3 * plane] = v.y * a.post;
    }
}

// Compile-time-plan, one-chunk-per-workgroup variant of the PART kernel: both global operands of every element (maps S and
// hybrid-space data yt) are fetched into registers with all loads issued back to back at kernel start (one memory round
// trip per workgroup instead of one per element and phase), S never goes through LDS, the mask row is staged in LDS.
// LDS = (2.5 + 2 G) W float2 (36 KB at W = 372, G = 5): 4 workgroups per CU.
template <class P, int NSEQ>
__global__ __launch_bounds__(MRX_FFT_NT, 4) void k_llg_rows_hinv_part(const float2* __restrict__ eta, const float2* __restrict__ yt,
                                                                      const float2* __restrict__ S, MrxMask mask,
                                                                      float2* __restrict__ part, ReduceArgs a, float scale_f) {
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    constexpr int W = P::N, G = NSEQ;
    constexpr int NE = (G * W + MRX_FFT_NT - 1) / MRX_FFT_NT;
    const int H = a.H, C = a.C;
    float2* tw = smem;
    float2* E = smem + W;
    float* Mk = reinterpret_cast<float*>(E + W);
    float2* A = E + W + (W + 1) / 2;
    float2* B = A + G * W;
    const int h = blockIdx.x;
    const long long b = blockIdx.y;
    const int c0 = (int)blockIdx.z * G;
    const bool mask_lds = mask.s[1] == 0;
    float2 sv[NE], yv[NE];
    {
        const long long base = (((b / a.sdiv) * C + c0) * H + h) * (long long)W;
        const long long ybase = ((b * C + c0) * H + h) * (long long)W;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int idx = threadIdx.x + e * MRX_FFT_NT;
            const int r = idx / W, x = idx - r * W;
            const bool ok = idx < G * W && c0 + r < C;
            const long long off = (long long)r * H * W + shifted(x, a.halfW, W);
            sv[e] = ok ? S[base + off] : make_float2(0.f, 0.f);
            yv[e] = ok ? yt[ybase + off] : make_float2(0.f, 0.f);
        }
    }
    for (int i = threadIdx.x; i < W; i += MRX_FFT_NT) {
        const int g = shifted(i, a.halfW, W);
        tw[i] = a.tw[i];
        E[i] = eta[(b * H + h) * W + g];
        if (mask_lds) Mk[i] = mrx_mask_val(mask, b, 0, 0, g);
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int idx = threadIdx.x + e * MRX_FFT_NT;
        if (idx < G * W) {
            const int r = idx / W, x = idx - r * W;
            const float2 ev = E[x];
            A[idx] = make_float2(ev.x * sv[e].x - ev.y * sv[e].y, ev.x * sv[e].y + ev.y * sv[e].x);  // rim_utils.py:47-48
        }
    }
    __syncthreads();
    float2* res = RunPlan<false, NSEQ, false, P>::run(A, B, tw);
    float2* oth = (res == A) ? B : A;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int idx = threadIdx.x + e * MRX_FFT_NT;
        if (idx < G * W) {
            const int r = idx / W, x = idx - r * W;
            const float m = mask_lds ? Mk[x] : mrx_mask_val(mask, b, c0 + r, 0, shifted(x, a.halfW, W));
            const float2 k = res[idx];
            res[idx] = make_float2(m * (k.x * scale_f - yv[e].x), m * (k.y * scale_f - yv[e].y));  // rim_utils.py:54
        }
    }
    __syncthreads();
    float2* res2 = RunPlan<true, NSEQ, false, P>::run(res, oth, tw);
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int idx = threadIdx.x + e * MRX_FFT_NT;
        if (idx < G * W) {
            float2 v = res2[idx];
            v.x *= a.scale;
            v.y *= a.scale;
            res2[idx] = make_float2(v.x * sv[e].x + v.y * sv[e].y, v.y * sv[e].x - v.x * sv[e].y);  // rim_utils.py:61-62
        }
    }
    __syncthreads();
    float2* po = part + ((long long)blockIdx.z * gridDim.y + b) * (long long)H * W + (long long)h * W;
    for (int x = threadIdx.x; x < W; x += MRX_FFT_NT) {
        float2 sum = make_float2(0.f, 0.f);
#pragma unroll
        for (int r = 0; r < G; ++r) {
            const float2 v = res2[r * W + x];   // rows beyond C are zero
            sum.x += v.x;
            sum.y += v.y;
        }
        po[shifted(x, a.halfW, W)] = sum;
    }
}

// ------------------------------------------------------------------------------------------------------------
// W = 372 on the matrix pipe.  The butterflies of the Stockham kernels above are vector-ALU work (16 fp32 FMA / clk / SIMD,
// packed ops take two passes); the fp32-input MFMA runs at 32.  372 = 31 x 12 (Cooley-Tukey, n = 12 n1 + n2, k = k1 + 31 k2):
//   A) 12 x 31-point DFTs per sequence as MFMA products with the real cos / sin matrices of the symmetric form
//        X_k = x0 + A_k -+ i B_k,  A_k = sum_t cos(2 pi k t / 31) (x_t + x_{31-t}),  B_k = sum_t sin(.) (x_t - x_{31-t}),  t = 1..15
//      (a 16th matrix row of ones delivers X_0 - x0 = sum_t a_t); the MFMA columns are (coil, n2, re|im), so the two halves of a
//      complex number sit in neighbouring lanes and meet through DPP quad swaps;
//   B) twiddle w^(n2 k1) on the way to LDS;
//   C) 31 x 12-point DFTs per sequence as one real [24 x 24] matrix (rows (k2, re|im), contraction (n2, re|im)).
// 46 MFMA 16x16x4 per sequence and direction; the VALU is left with the pre-adds, the sign/twiddle fix-ups and addresses.
// Same contract, grid and partial-sum layout as k_llg_rows_hinv_part<P372, 5>.
// ------------------------------------------------------------------------------------------------------------
typedef float mfx4 __attribute__((ext_vector_type(4)));
#define M372_G 5
#define M372_TS 49                     // stride of the (n2, re|im) rows of the intermediate array (conflict-free both ways)
#define M372_TC (24 * M372_TS)         // floats per coil in the intermediate array
struct M372Tables {
    const float* ca;   // [4][64]  step A cos operand per lane
    const float* sa;   // [4][64]  step A sin operand per lane
    const float* m24;  // [2 inverse][2 tiles][6][64]  step C operand per lane
};

__device__ __forceinline__ float m372_swap(float v) {  // value of the neighbouring lane (lane ^ 1): quad_perm [1,0,3,2]
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

// twT = the step-B twiddles as a [12][33] table (33: rows land on different banks): twT[n2 * 33 + k1] = w^(n2 k1), so a lane's
// eight factors sit at immediate offsets from two bases.
template <bool INV>
__device__ __forceinline__ void m372_step_a(const float* __restrict__ X, float* __restrict__ T, const float2* __restrict__ twT,
                                            const float (&cA)[4], const float (&sA)[4], int wave, int lane) {
    const int l15 = lane & 15, lg = lane >> 4;
    // both tiles of the wave side by side: two independent dependency chains (loads -> MFMA -> DPP fix-ups -> stores)
    bool valid[2];
    int c[2];
    const float2* wlo[2];  // &twT[n2][4 lg + 1]: factors of rows k = 4 lg + 1 + r
    const float2* whi[2];  // &twT[n2][30 - 4 lg]: factors of rows 31 - k
    float* Tc[2];
    float xa[2][4], xb[2][4], x0[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int col_raw = (wave + 4 * tt) * 16 + l15;
        valid[tt] = col_raw < 2 * 12 * M372_G;
        const int col = valid[tt] ? col_raw : 2 * 12 * M372_G - 1;  // idle columns of the last tile repeat column 119 (not stored)
        const int cc = col >> 1;
        c[tt] = col & 1;
        const int coil = cc / 12, n2 = cc - 12 * coil;
        const float* Xc = X + (coil * 372 + n2) * 2 + c[tt] + 24 * (lg + 1);
        const float* Xd = X + (coil * 372 + n2) * 2 + c[tt] + 24 * (30 - lg);
        Tc[tt] = T + coil * M372_TC + (2 * n2 + c[tt]) * M372_TS + 4 * lg + 1;
        wlo[tt] = twT + n2 * 33 + 4 * lg + 1;
        whi[tt] = twT + n2 * 33 + 30 - 4 * lg;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {  // t = 4 ks + lg + 1; t = 16 (ks = lg = 3) is a real element whose matrix column is zero
            xa[tt][ks] = Xc[96 * ks];
            xb[tt][ks] = Xd[-96 * ks];
        }
        x0[tt] = Xc[-24 * (lg + 1)];
    }
    mfx4 accA[2], accB[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        accA[tt] = (mfx4){0.f, 0.f, 0.f, 0.f};
        accB[tt] = (mfx4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            accA[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(cA[ks], xa[tt][ks] + xb[tt][ks], accA[tt], 0, 0, 0);
            accB[tt] = __builtin_amdgcn_mfma_f32_16x16x4f32(sA[ks], xa[tt][ks] - xb[tt][ks], accB[tt], 0, 0, 0);
        }
    const bool ones_row = lg == 3;  // this lane's r = 3 is matrix row 16 (the row of ones): X_0 = x0 + sum_t a_t, twiddle 1
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        // forward: X_k = x0 + A - i B (re: + B_im, im: - B_re), X_{31-k} = x0 + A + i B; inverse: signs of B exchanged
        const float sgn = ((c[tt] == 0) != INV) ? 1.f : -1.f;
        const float tsg = ((c[tt] == 0) != INV) ? -1.f : 1.f;  // y * w (w conjugated for the inverse)
        float o1[4], o2[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float2 w1 = wlo[tt][r], w2 = whi[tt][-r];
            const float bs = sgn * m372_swap(accB[tt][r]);
            const float t0 = x0[tt] + accA[tt][r];
            const float y1 = t0 + bs, y2 = t0 - bs;
            o1[r] = y1 * w1.x + tsg * (m372_swap(y1) * w1.y);
            o2[r] = y2 * w2.x + tsg * (m372_swap(y2) * w2.y);
            if (r == 3 && ones_row) o1[r] = t0;
        }
        if (valid[tt]) {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                Tc[tt][r] = o1[r];
                Tc[tt][29 - 8 * lg - r] = o2[r];  // 31 - k relative to the base 4 lg + 1
            }
            // r = 3: rows k = 4 lg + 4 and 31 - k, except in the lg = 3 lanes, which hold X_0 (slot 0) and nothing else (slot 31: pad)
            Tc[tt][ones_row ? -13 : 3] = o1[3];
            Tc[tt][ones_row ? 18 : 26 - 8 * lg] = o2[3];
        }
    }
}

template <bool INV>
__device__ __forceinline__ void m372_step_c(const float* __restrict__ T, float2* __restrict__ X, const float* __restrict__ m24l, int wave, int lane) {
    const int l15 = lane & 15, lg = lane >> 4;
    float m24[2][6];  // per-lane operand constants from LDS (lane-linear, conflict-free)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) m24[t][ks] = m24l[(((INV ? 2 : 0) + t) * 6 + ks) * 64 + lane];
    for (int tile = wave; tile < 10; tile += 4) {
        const int col_raw = tile * 16 + l15;
        const bool valid = col_raw < 31 * M372_G;
        const int col = valid ? col_raw : 31 * M372_G - 1;
        const int coil = col / 31, k1 = col - 31 * coil;
        const float* Tc = T + coil * M372_TC + k1;
        float v[6];
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) v[ks] = Tc[(4 * ks + lg) * M372_TS];
        mfx4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 6; ++ks) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(m24[0][ks], v[ks], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(m24[1][ks], v[ks], acc1, 0, 0, 0);
        }
        if (valid) {
            float2* Xc = X + coil * 372 + k1;
            Xc[31 * (2 * lg)] = make_float2(acc0[0], acc0[1]);
            Xc[31 * (2 * lg + 1)] = make_float2(acc0[2], acc0[3]);
            if (lg < 2) {
                Xc[31 * (8 + 2 * lg)] = make_float2(acc1[0], acc1[1]);
                Xc[31 * (9 + 2 * lg)] = make_float2(acc1[2], acc1[3]);
            }
        }
    }
}

__device__ unsigned long long* g_m372_trace = nullptr;  // debug only (env MRX_TRACE): 8 cycle stamps per workgroup (first item)
#define M372_STAMP(i) \
    if (trc && threadIdx.x == 0 && first) trc[(long long)blockIdx.x * 8 + (i)] = __builtin_readcyclecounter();
// Persistent: gridDim.x workgroups (3 per CU) walk the (row, batch) pairs and, inside a row, its coil chunks, keeping the coil sum
// in registers (no partial sums in memory, no combine launch); the maps of the NEXT chunk and the hybrid-space data of the
// current one are fetched into registers while the matrix pipe is busy.  Everything that depends on
// the thread only (element -> (coil row, column, byte offset)) is computed once per workgroup; per item the global operands are
// addressed as scalar base + 32-bit lane offset, so the element-wise steps are loads, 4-6 FMAs and an LDS access each.
__global__ __launch_bounds__(MRX_FFT_NT, 3) void k_llg_rows_hinv_mfma372(const float2* __restrict__ eta, const float2* __restrict__ yt,
                                                                          const float2* __restrict__ S, MrxMask mask,
                                                                          float* __restrict__ out4, ReduceArgs a, float scale_f,
                                                                          M372Tables tb, int nB, int nchunks, int abl) {
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    constexpr int W = 372, G = M372_G;
    constexpr int NE = (G * W + MRX_FFT_NT - 1) / MRX_FFT_NT;  // 8; elements tid + 256 e, all < G W except e = 7 for tid >= 68
    const int H = a.H, C = a.C;
    unsigned long long* trc = g_m372_trace;
    float2* tw = smem;  // [12][33] step-B twiddles
    float* Mk = reinterpret_cast<float*>(tw + 12 * 33);
    float2* E = tw + 12 * 33 + W / 2;
    float2* Xa = E + W;
    float* T = reinterpret_cast<float*>(Xa + G * W);
    float* M24 = T + G * M372_TC;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool mask_lds = mask.s[1] == 0;
    const int nitems = H * nB * nchunks;  // item = (row pair index) * nchunks + chunk; a workgroup owns whole rows
    float cA[4], sA[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        cA[ks] = tb.ca[ks * 64 + lane];
        sA[ks] = tb.sa[ks * 64 + lane];
    }
    for (int i = tid; i < 12 * 33; i += MRX_FFT_NT) tw[i] = a.tw[((i / 33) * (i % 33)) % W];  // twT[n2][k1] = w^(n2 k1)
    for (int i = tid; i < 4 * 6 * 64; i += MRX_FFT_NT) M24[i] = tb.m24[i];
    // per-thread element table
    unsigned goff[NE];   // byte offset of element e inside the (b, c0, h) operand block: (r H W + shifted(x)) * 8
    unsigned short xe[NE];  // column x of element e
    unsigned rpack = 0;  // coil row r of element e (3 bits each)
    const bool last_ok = tid + (NE - 1) * MRX_FFT_NT < G * W;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        int idx = tid + e * MRX_FFT_NT;
        idx = idx < G * W ? idx : G * W - 1;
        const int r = idx / W, x = idx - r * W;
        goff[e] = (unsigned)((long long)r * H * W + shifted(x, a.halfW, W)) * 8u;
        xe[e] = (unsigned short)x;
        rpack |= (unsigned)r << (3 * e);
    }
    const unsigned eoff0 = (unsigned)shifted(tid, a.halfW, W) * 8u;
    const unsigned eoff1 = (unsigned)shifted(tid + MRX_FFT_NT < W ? tid + MRX_FFT_NT : W - 1, a.halfW, W) * 8u;
    auto decode = [&](int item, int& h, long long& b, int& z) {
        z = item % nchunks;
        const int row = item / nchunks;
        h = row % H;
        b = row / H;
    };
    // sequence of this workgroup: rows blockIdx.x, blockIdx.x + gridDim.x, ...; all chunks of a row back to back
    auto next_item = [&](int item) { return (item % nchunks) + 1 < nchunks ? item + 1 : (item / nchunks + (int)gridDim.x) * nchunks; };
    // maps sv (live for the whole item), hybrid-space data yv (fetched after the expansion, consumed by the mask step), the next
    // item's maps sn + eta row en (fetched after the mask step, when yv is dead): two sets of NE registers at any time
    float2 sv[NE], ev[2];
    auto fetch_s = [&](int item, float2 (&s_)[NE], float2 (&e_)[2]) {
        int h, z;
        long long b;
        decode(item, h, b, z);
        const int c0 = z * G;
        const char* sb = reinterpret_cast<const char*>(S + (((b / a.sdiv) * C + c0) * H + h) * (long long)W);
        const char* eb = reinterpret_cast<const char*>(eta + (b * H + h) * (long long)W);
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const bool ok = c0 + (int)((rpack >> (3 * e)) & 7u) < C && (e < NE - 1 || last_ok);
            s_[e] = ok ? *reinterpret_cast<const float2*>(sb + goff[e]) : make_float2(0.f, 0.f);
        }
        e_[0] = *reinterpret_cast<const float2*>(eb + eoff0);
        e_[1] = *reinterpret_cast<const float2*>(eb + eoff1);
    };
    bool first = true;
    long long b_mask = -1;
    int item = blockIdx.x * nchunks;
    float2 gsum[2] = {make_float2(0.f, 0.f), make_float2(0.f, 0.f)};  // coil sum of columns tid, tid + 256 over the chunks of the row
    M372_STAMP(0)
    if (item < nitems) fetch_s(item, sv, ev);
    for (; item < nitems; item = next_item(item)) {
        int h, z;
        long long b;
        decode(item, h, b, z);
        const int c0 = z * G;
        // the previous item's readers of E / Mk / Xa are past the barrier that ended it
        E[tid] = ev[0];
        if (tid + MRX_FFT_NT < W) E[tid + MRX_FFT_NT] = ev[1];
        if (mask_lds && b != b_mask) {
            for (int i = tid; i < W; i += MRX_FFT_NT) Mk[i] = mrx_mask_val(mask, b, 0, 0, shifted(i, a.halfW, W));
            b_mask = b;
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            if (e < NE - 1 || last_ok) {
                const float2 e2 = E[xe[e]];
                Xa[tid + e * MRX_FFT_NT] = make_float2(e2.x * sv[e].x - e2.y * sv[e].y, e2.x * sv[e].y + e2.y * sv[e].x);  // rim_utils.py:47-48
            }
        }
        float2 yv[NE];  // in flight during the forward transform
        {
            const char* yb = reinterpret_cast<const char*>(yt + ((b * C + c0) * H + h) * (long long)W);
#pragma unroll
            for (int e = 0; e < NE; ++e) {
                const bool ok = c0 + (int)((rpack >> (3 * e)) & 7u) < C && (e < NE - 1 || last_ok);
                yv[e] = ok ? *reinterpret_cast<const float2*>(yb + goff[e]) : make_float2(0.f, 0.f);
            }
        }
        __syncthreads();
        M372_STAMP(1)
        int lane_o = lane;  // opaque per call: the per-lane LDS addresses of the steps are recomputed, not kept across items
        asm volatile("" : "+v"(lane_o));
        if (!(abl & 1)) m372_step_a<false>(reinterpret_cast<const float*>(Xa), T, tw, cA, sA, wave, lane_o);
        __syncthreads();
        M372_STAMP(2)
        asm volatile("" : "+v"(lane_o));
        if (!(abl & 2)) m372_step_c<false>(T, Xa, M24, wave, lane_o);
        __syncthreads();
        M372_STAMP(3)
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            if (e < NE - 1 || last_ok) {
                const float m = mask_lds ? Mk[xe[e]]
                                         : mrx_mask_val(mask, b, c0 + (int)((rpack >> (3 * e)) & 7u), 0, shifted(xe[e], a.halfW, W));
                const float2 k = Xa[tid + e * MRX_FFT_NT];
                Xa[tid + e * MRX_FFT_NT] = make_float2(m * (k.x * scale_f - yv[e].x), m * (k.y * scale_f - yv[e].y));  // rim_utils.py:54
            }
        }
        float2 sn[NE], en[2];  // next item's maps and eta row: in flight during the inverse transform
        const bool more = next_item(item) < nitems;
        if (more) fetch_s(next_item(item), sn, en);
        __syncthreads();
        M372_STAMP(4)
        asm volatile("" : "+v"(lane_o));
        if (!(abl & 1)) m372_step_a<true>(reinterpret_cast<const float*>(Xa), T, tw, cA, sA, wave, lane_o);
        __syncthreads();
        M372_STAMP(5)
        asm volatile("" : "+v"(lane_o));
        if (!(abl & 2)) m372_step_c<true>(T, Xa, M24, wave, lane_o);
        __syncthreads();
        M372_STAMP(6)
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            if (e < NE - 1 || last_ok) {
                float2 v = Xa[tid + e * MRX_FFT_NT];
                v.x *= a.scale;
                v.y *= a.scale;
                Xa[tid + e * MRX_FFT_NT] = make_float2(v.x * sv[e].x + v.y * sv[e].y, v.y * sv[e].x - v.x * sv[e].y);  // rim_utils.py:61-62
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int x = tid + e * MRX_FFT_NT;
            if (x < W) {
                float2 sum = make_float2(0.f, 0.f);
#pragma unroll
                for (int r = 0; r < G; ++r) {
                    const float2 v = Xa[r * W + x];   // rows beyond C are zero
                    sum.x += v.x;
                    sum.y += v.y;
                }
                gsum[e].x = z == 0 ? sum.x : gsum[e].x + sum.x;
                gsum[e].y = z == 0 ? sum.y : gsum[e].y + sum.y;
                if (z == nchunks - 1) {  // (eta_re, eta_im, g_re / sigma^2, g_im / sigma^2)   rim_utils.py:61-67
                    const float2 e2 = E[x];
                    const long long plane = (long long)H * W;
                    float* o = out4 + b * 4 * plane + (long long)h * W + shifted(x, a.halfW, W);
                    o[0] = e2.x;
                    o[plane] = e2.y;
                    o[2 * plane] = gsum[e].x * a.post;
                    o[3 * plane] = gsum[e].y * a.post;
                }
            }
        }
        M372_STAMP(7)
        first = false;
        if (more) {
#pragma unroll
            for (int e = 0; e < NE; ++e) sv[e] = sn[e];
            ev[0] = en[0];
            ev[1] = en[1];
        }
        __syncthreads();  // Xa / E free for the next item
    }
}

// per-lane MFMA operand tables of the kernel above (built once, device-resident)
static int m372_tables(M372Tables* out) {
    static M372Tables t = {nullptr, nullptr, nullptr};
    if (!t.ca) {
        std::vector<float> ca(4 * 64), sa(4 * 64), m24(2 * 2 * 6 * 64);
        for (int ks = 0; ks < 4; ++ks)
            for (int lane = 0; lane < 64; ++lane) {
                const int k = (lane & 15) + 1, tt = 4 * ks + (lane >> 4) + 1;
                const double th = 2.0 * M_PI * (double)((k * tt) % 31) / 31.0;
                ca[ks * 64 + lane] = tt > 15 ? 0.f : (k == 16 ? 1.f : (float)cos(th));
                sa[ks * 64 + lane] = (tt > 15 || k == 16) ? 0.f : (float)sin(th);
            }
        for (int inv = 0; inv < 2; ++inv)
            for (int tile = 0; tile < 2; ++tile)
                for (int ks = 0; ks < 6; ++ks)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int m = tile * 16 + (lane & 15), rho = 4 * ks + (lane >> 4);
                        float v = 0.f;
                        if (m < 24) {
                            const int k2 = m >> 1, co = m & 1, n2 = rho >> 1, ci = rho & 1;
                            const double ph = 2.0 * M_PI * (double)((k2 * n2) % 12) / 12.0;
                            const double fr = cos(ph), fi = inv ? sin(ph) : -sin(ph);
                            v = (float)(co == ci ? fr : (co == 0 ? -fi : fi));
                        }
                        m24[((inv * 2 + tile) * 6 + ks) * 64 + lane] = v;
                    }
        float *d_ca, *d_sa, *d_m;
        MRX_HIP(hipMalloc((void**)&d_ca, sizeof(float) * ca.size()));
        MRX_HIP(hipMalloc((void**)&d_sa, sizeof(float) * sa.size()));
        MRX_HIP(hipMalloc((void**)&d_m, sizeof(float) * m24.size()));
        MRX_HIP(hipMemcpy(d_ca, ca.data(), sizeof(float) * ca.size(), hipMemcpyHostToDevice));
        MRX_HIP(hipMemcpy(d_sa, sa.data(), sizeof(float) * sa.size(), hipMemcpyHostToDevice));
        MRX_HIP(hipMemcpy(d_m, m24.data(), sizeof(float) * m24.size(), hipMemcpyHostToDevice));
        t.ca = d_ca;
        t.sa = d_sa;
        t.m24 = d_m;
    }
    *out = t;
    return MRX_OK;
}

// out4[b] = (eta_re, eta_im, post * sum_k part_k.re, post * sum_k part_k.im)   (rim_utils.py:61-67)
__global__ void k_llg_combine(const float2* __restrict__ eta, const float2* __restrict__ part, float* __restrict__ out, int nparts,
                              long long B, long long plane, float post) {
    const long long total = B * plane;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / plane, p = i - b * plane;
        float2 s = part[i];
        for (int k = 1; k < nparts; ++k) {
            const float2 v = part[(long long)k * total + i];
            s.x += v.x;
            s.y += v.y;
        }
        const float2 e = eta[i];
        float* o = out + b * 4 * plane + p;
        o[0] = e.x;
        o[plane] = e.y;
        o[2 * plane] = s.x * post;
        o[3 * plane] = s.y * post;
    }
}

// ------------------------------------------------------------------------------------------------------------
// host launch helpers.  Compile-time plans for the lengths of the named configs (640x372 knee, 320x320, 256x256);
// every other length runs the runtime-plan kernels.
// ------------------------------------------------------------------------------------------------------------
typedef PlanCT<372, 31, 3, 4> P372;
typedef PlanCT<640, 5, 8, 4, 4> P640;
typedef PlanCT<320, 5, 8, 8> P320;
typedef PlanCT<256, 8, 8, 4> P256;
// rows per workgroup (row kernels) / columns per workgroup (column kernels) for the compile-time plans
#define NSEQ_ROW_372 4
#define NSEQ_HINV_372 4  // one-launch gradient: 4 coils per workgroup = 4 chunks at C = 15, 5 workgroups per CU, exactly two dispatch rounds (measured 40.6 vs 44.0 us)
#define NSEQ_ROW_320 6
#define NSEQ_ROW_256 8
#define NSEQ_COL_640 4
#define NSEQ_COL_320 8
#define NSEQ_COL_256 8

static inline int pick_rows(int W) {
    if (W == 372) return NSEQ_ROW_372;
    if (W == 320) return NSEQ_ROW_320;
    if (W == 256) return NSEQ_ROW_256;
    int r = MRX_FFT_TILE_ELEMS / W;
    return r < 1 ? 1 : (r > 16 ? 16 : r);
}
static inline int pick_cols(int H) {
    if (H == 640) return NSEQ_COL_640;
    if (H == 320) return NSEQ_COL_320;
    if (H == 256) return NSEQ_COL_256;
    int ct = 16;
    while (ct > 1 && ct * H > MRX_FFT_TILE_ELEMS + 512) ct >>= 1;
    return ct;
}

// raises the dynamic-LDS limit of a kernel once (not on every launch: keeps launches legal under hipGraph capture)
template <typename K>
static int set_lds(K kern, size_t bytes) {
    MRX_REQUIRE(bytes <= 160 * 1024, MRX_EUNSUP, "FFT tile needs %zu bytes of LDS", bytes);
    static std::mutex mu;
    static std::unordered_map<const void*, size_t> done;
    if (bytes > 48 * 1024) {
        std::lock_guard<std::mutex> lk(mu);
        size_t& cur = done[(const void*)kern];
        if (cur < bytes) {
            MRX_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
            cur = bytes;
        }
    }
    return MRX_OK;
}

template <bool INV, int MODE, class P, int NSEQ>
static int launch_rows_t(const float2* in, const float2* S, float2* out, dim3 grid, size_t lds, const RowArgs& a, hipStream_t st,
                         const RowDc* dcp = nullptr) {
    int rc = set_lds(k_fft_rows<INV, MODE, P, NSEQ>, lds);
    if (rc) return rc;
    RowDc dc;
    if (dcp) dc = *dcp;
    else {
        dc.pred = dc.ref = nullptr;
        dc.w = nullptr;
        dc.m.p = nullptr;
        dc.m.kind = MRX_MASK_F32;
        for (int i = 0; i < 4; ++i) dc.m.s[i] = 0;
    }
    hipLaunchKernelGGL((k_fft_rows<INV, MODE, P, NSEQ>), grid, dim3(MRX_FFT_NT), lds, st, in, S, out, a, dc);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
template <class P, int NSEQ>
static int launch_rows_p(const float2* in, const float2* S, float2* out, dim3 grid, size_t lds, const RowArgs& a, int inverse,
                         int expand, hipStream_t st, const RowDc* dc = nullptr) {
    if (expand && dc) return launch_rows_t<false, 2, P, NSEQ>(in, S, out, grid, lds, a, st, dc);
    if (expand) return launch_rows_t<false, 1, P, NSEQ>(in, S, out, grid, lds, a, st);
    if (inverse) return launch_rows_t<true, 0, P, NSEQ>(in, S, out, grid, lds, a, st);
    return launch_rows_t<false, 0, P, NSEQ>(in, S, out, grid, lds, a, st);
}

static int launch_rows(const float2* in, const float2* S, float2* out, long long nimg, int W, int C, int H,
                       int inverse, int norm, int centered, int expand, hipStream_t st, int sdiv = 1, const RowDc* dc = nullptr) {
    MrxFftEntry e;
    int rc = mrx_get_plan(W, &e);
    if (rc) return rc;
    RowArgs a;
    a.plan = e.plan;
    a.tw = e.d_tw;
    a.W = W;
    a.rpb = pick_rows(W);
    a.halfW = centered ? W / 2 : 0;
    a.scale = mrx_scale(W, inverse, norm);
    a.C = C;
    a.H = H;
    a.sdiv = sdiv < 1 ? 1 : sdiv;
    const size_t lds = sizeof(float2) * ((size_t)W + 2 * (size_t)a.rpb * W);
    MRX_REQUIRE(nimg < (1LL << 31), MRX_EUNSUP, "too many images (%lld)", nimg);
    const dim3 grid((unsigned)nimg, mrx_cdiv(H, a.rpb));
    if (W == 372) return launch_rows_p<P372, NSEQ_ROW_372>(in, S, out, grid, lds, a, inverse, expand, st, dc);
    if (W == 320) return launch_rows_p<P320, NSEQ_ROW_320>(in, S, out, grid, lds, a, inverse, expand, st, dc);
    if (W == 256) return launch_rows_p<P256, NSEQ_ROW_256>(in, S, out, grid, lds, a, inverse, expand, st, dc);
    return launch_rows_p<PlanRT, 1>(in, S, out, grid, lds, a, inverse, expand, st, dc);
}

static int make_col_args(ColArgs* a, long long nimg, int H, int W, int inverse, int norm, int centered) {
    MrxFftEntry e;
    int rc = mrx_get_plan(H, &e);
    if (rc) return rc;
    a->plan = e.plan;
    a->tw = e.d_tw;
    a->H = H;
    a->W = W;
    a->ct = pick_cols(H);
    a->halfH = centered ? H / 2 : 0;
    a->scale = mrx_scale(H, inverse, norm);
    a->scale2 = 1.0f;
    a->C = 1;
    a->ntx = mrx_cdiv(W, a->ct);
    a->nblocks = nimg * a->ntx;
    MRX_REQUIRE(a->nblocks < (1LL << 31), MRX_EUNSUP, "too many column tiles (%lld)", a->nblocks);
    return MRX_OK;
}

template <class P, int NSEQ>
static int launch_cols_p(const float2* in, float2* out, const ColArgs& a, size_t lds, int inverse, hipStream_t st) {
    int rc;
    if (inverse) {
        if ((rc = set_lds(k_fft_cols<true, P, NSEQ>, lds))) return rc;
        hipLaunchKernelGGL((k_fft_cols<true, P, NSEQ>), dim3((unsigned)a.nblocks), dim3(MRX_FFT_NT), lds, st, in, out, a);
    } else {
        if ((rc = set_lds(k_fft_cols<false, P, NSEQ>, lds))) return rc;
        hipLaunchKernelGGL((k_fft_cols<false, P, NSEQ>), dim3((unsigned)a.nblocks), dim3(MRX_FFT_NT), lds, st, in, out, a);
    }
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

static int launch_cols(const float2* in, float2* out, long long nimg, int H, int W, int inverse, int norm, int centered,
                       hipStream_t st) {
    ColArgs a;
    int rc = make_col_args(&a, nimg, H, W, inverse, norm, centered);
    if (rc) return rc;
    const size_t lds = sizeof(float2) * ((size_t)H + 2 * (size_t)a.ct * H);
    if (H == 640) return launch_cols_p<P640, NSEQ_COL_640>(in, out, a, lds, inverse, st);
    if (H == 320) return launch_cols_p<P320, NSEQ_COL_320>(in, out, a, lds, inverse, st);
    if (H == 256) return launch_cols_p<P256, NSEQ_COL_256>(in, out, a, lds, inverse, st);
    return launch_cols_p<PlanRT, 1>(in, out, a, lds, inverse, st);
}

template <class P, int NSEQ>
static int launch_dc_p(const float2* in, const float2* y, const MrxMask& m, float2* out, const ColArgs& a, size_t lds,
                       hipStream_t st) {
    int rc = set_lds(k_cols_dc<P, NSEQ>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((k_cols_dc<P, NSEQ>), dim3((unsigned)a.nblocks), dim3(MRX_FFT_NT), lds, st, in, y, m, out, a);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

template <class P, int NSEQ>
static int launch_reduce_p(const float2* k, const float2* S, const float2* eta, float* out, dim3 grid, size_t lds,
                           const ReduceArgs& a, int out_mode, hipStream_t st) {
    int rc;
    if (out_mode == 0) {
        if ((rc = set_lds(k_rows_reduce<0, P, NSEQ>, lds))) return rc;
        hipLaunchKernelGGL((k_rows_reduce<0, P, NSEQ>), grid, dim3(MRX_FFT_NT), lds, st, k, S, eta, out, a);
    } else {
        if ((rc = set_lds(k_rows_reduce<1, P, NSEQ>, lds))) return rc;
        hipLaunchKernelGGL((k_rows_reduce<1, P, NSEQ>), grid, dim3(MRX_FFT_NT), lds, st, k, S, eta, out, a);
    }
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

static int launch_reduce(const float2* k, const float2* S, const float2* eta, float* out, int B, int C, int H, int W,
                         int norm, int centered, float post, int out_mode, hipStream_t st, int sdiv = 1) {
    MrxFftEntry e;
    int rc = mrx_get_plan(W, &e);
    if (rc) return rc;
    ReduceArgs a;
    a.plan = e.plan;
    a.tw = e.d_tw;
    a.sdiv = sdiv < 1 ? 1 : sdiv;
    a.C = C;
    a.H = H;
    a.W = W;
    a.g = pick_rows(W);
    const bool ct = (W == 372 || W == 320 || W == 256);
    if (!ct && a.g > C) a.g = C;
    a.halfW = centered ? W / 2 : 0;
    a.scale = mrx_scale(W, 1, norm);
    a.post = post;
    MRX_REQUIRE(B <= 65535, MRX_EUNSUP, "batch %d too large", B);
    const size_t lds = sizeof(float2) * (2 * (size_t)W + 2 * (size_t)a.g * W);
    dim3 grid(H, B);
    if (W == 372) return launch_reduce_p<P372, NSEQ_ROW_372>(k, S, eta, out, grid, lds, a, out_mode, st);
    if (W == 320) return launch_reduce_p<P320, NSEQ_ROW_320>(k, S, eta, out, grid, lds, a, out_mode, st);
    if (W == 256) return launch_reduce_p<P256, NSEQ_ROW_256>(k, S, eta, out, grid, lds, a, out_mode, st);
    return launch_reduce_p<PlanRT, 1>(k, S, eta, out, grid, lds, a, out_mode, st);
}

// ------------------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------------------
extern "C" int mrx_fft_max_len(void) { return MRX_FFT_MAX_LEN; }

extern "C" int mrx_fft_prepare(int h, int w) {
    MrxFftEntry e;
    int rc = mrx_get_plan(h, &e);
    if (rc) return rc;
    if (w == 372) {
        M372Tables tb;
        if ((rc = m372_tables(&tb))) return rc;
    }
    return mrx_get_plan(w, &e);
}

static int norm_valid(int norm) { return norm >= 0 && norm <= 3; }

extern "C" int mrx_fft2(const float* in, float* out, int64_t batch, int H, int W, int inverse, int norm, int centered,
                        void* stream) {
    MRX_REQUIRE(batch >= 0 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_fft2: bad dims batch=%lld H=%d W=%d", (long long)batch, H, W);
    MRX_REQUIRE(norm_valid(norm), MRX_EINVAL, "mrx_fft2: bad normalization %d", norm);
    if (batch == 0) return MRX_OK;
    MRX_REQUIRE(in && out, MRX_EINVAL, "mrx_fft2: null pointer");
    hipStream_t st = (hipStream_t)stream;
    int rc = launch_rows((const float2*)in, nullptr, (float2*)out, batch, W, 1, H, inverse, norm, centered, 0, st);
    if (rc) return rc;
    return launch_cols((float2*)out, (float2*)out, batch, H, W, inverse, norm, centered, st);
}

extern "C" int mrx_sens_expand(const float* x, const float* S, float* out, int B, int C, int H, int W, int norm,
                               int centered, void* stream) {
    MRX_REQUIRE(x && S && out, MRX_EINVAL, "mrx_sens_expand: null pointer");
    MRX_REQUIRE(B >= 0 && C >= 1 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_sens_expand: bad dims");
    MRX_REQUIRE(norm_valid(norm), MRX_EINVAL, "mrx_sens_expand: bad normalization %d", norm);
    if (B == 0) return MRX_OK;
    hipStream_t st = (hipStream_t)stream;
    int rc = launch_rows((const float2*)x, (const float2*)S, (float2*)out, (long long)B * C, W, C, H, 0, norm,
                         centered, 1, st);
    if (rc) return rc;
    return launch_cols((float2*)out, (float2*)out, (long long)B * C, H, W, 0, norm, centered, st);
}

// Hybrid-space forms for row-invariant (1-D column) masks: with k-space kept as kh = IFFT_H(k) (mrx_fft_cols, once per slice), every
// masked data-consistency step commutes with the H transform, so the cascades need row transforms only:
//   mrx_sens_expand_rows   out = FFT_W(x * S)            (= IFFT_H of mrx_sens_expand)
//   mrx_sens_reduce_rows   out = sum_c IFFT_W(kh) conj(S) (= mrx_sens_reduce of the k-space kh stands for)
extern "C" int mrx_sens_expand_rows(const float* x, const float* S, float* out, int B, int C, int H, int W, int norm, int centered,
                                    void* stream) {
    MRX_REQUIRE(x && S && out, MRX_EINVAL, "mrx_sens_expand_rows: null pointer");
    MRX_REQUIRE(B >= 0 && C >= 1 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_sens_expand_rows: bad dims");
    MRX_REQUIRE(norm_valid(norm), MRX_EINVAL, "mrx_sens_expand_rows: bad normalization %d", norm);
    if (B == 0) return MRX_OK;
    return launch_rows((const float2*)x, (const float2*)S, (float2*)out, (long long)B * C, W, C, H, 0, norm, centered, 1,
                       (hipStream_t)stream);
}
// mrx_sens_expand_rows + mrx_dc_combine in one pass: out = pred - where(mask, pred - ref, 0) * dc_weight - FFT_W(x * S), all k-space
// arguments in hybrid space (vn_block.py:109-117 / ccnn_block.py:127-138 for row-invariant masks)
extern "C" int mrx_sens_expand_rows_dc(const float* x, const float* S, const float* pred, const float* ref, const void* mask,
                                       int mask_kind, const int64_t* mstride, const float* dc_weight, float* out, int B, int C, int H,
                                       int W, int norm, int centered, void* stream) {
    MRX_REQUIRE(x && S && pred && ref && mask && mstride && dc_weight && out, MRX_EINVAL, "mrx_sens_expand_rows_dc: null pointer");
    MRX_REQUIRE(B >= 0 && C >= 1 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_sens_expand_rows_dc: bad dims");
    MRX_REQUIRE(norm_valid(norm), MRX_EINVAL, "mrx_sens_expand_rows_dc: bad normalization %d", norm);
    MRX_REQUIRE(mask_kind == MRX_MASK_U8 || mask_kind == MRX_MASK_F32, MRX_EINVAL, "mrx_sens_expand_rows_dc: bad mask kind %d", mask_kind);
    MRX_REQUIRE(out != (float*)x, MRX_EINVAL, "mrx_sens_expand_rows_dc: out must not alias x");
    if (B == 0) return MRX_OK;
    RowDc dc;
    dc.pred = (const float2*)pred;
    dc.ref = (const float2*)ref;
    dc.w = dc_weight;
    dc.m.p = mask;
    dc.m.kind = mask_kind;
    for (int i = 0; i < 4; ++i) dc.m.s[i] = mstride[i];
    return launch_rows((const float2*)x, (const float2*)S, (float2*)out, (long long)B * C, W, C, H, 0, norm, centered, 1,
                       (hipStream_t)stream, 1, &dc);
}
extern "C" int mrx_sens_reduce_rows(const float* kh, const float* S, float* out, int B, int C, int H, int W, int norm, int centered,
                                    void* stream) {
    MRX_REQUIRE(kh && S && out, MRX_EINVAL, "mrx_sens_reduce_rows: null pointer");
    MRX_REQUIRE(B >= 0 && C >= 1 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_sens_reduce_rows: bad dims");
    MRX_REQUIRE(norm_valid(norm), MRX_EINVAL, "mrx_sens_reduce_rows: bad normalization %d", norm);
    if (B == 0) return MRX_OK;
    return launch_reduce((const float2*)kh, (const float2*)S, nullptr, out, B, C, H, W, norm, centered, 1.0f, 0, (hipStream_t)stream);
}
extern "C" int mrx_sens_reduce(const float* k, const float* S, float* out, float* work, int B, int C, int H, int W,
                               int norm, int centered, void* stream) {
    MRX_REQUIRE(k && S && out && work, MRX_EINVAL, "mrx_sens_reduce: null pointer");
    MRX_REQUIRE(B >= 0 && C >= 1 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_sens_reduce: bad dims");
    MRX_REQUIRE(norm_valid(norm), MRX_EINVAL, "mrx_sens_reduce: bad normalization %d", norm);
    if (B == 0) return MRX_OK;
    hipStream_t st = (hipStream_t)stream;
    int rc = launch_cols((const float2*)k, (float2*)work, (long long)B * C, H, W, 1, norm, centered, st);
    if (rc) return rc;
    return launch_reduce((const float2*)work, (const float2*)S, nullptr, out, B, C, H, W, norm, centered, 1.0f, 0, st);
}

extern "C" int mrx_llg(const float* eta, const float* y, const float* S, const void* mask, int mask_kind,
                       const int64_t* mstride, float* out4, float* work, int B, int C, int H, int W, float inv_sigma2,
                       int norm, int centered, void* stream) {
    MRX_REQUIRE(eta && y && S && mask && mstride && out4 && work, MRX_EINVAL, "mrx_llg: null pointer");
    MRX_REQUIRE(B >= 0 && C >= 1 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_llg: bad dims");
    MRX_REQUIRE(norm_valid(norm), MRX_EINVAL, "mrx_llg: bad normalization %d", norm);
    MRX_REQUIRE(mask_kind == MRX_MASK_U8 || mask_kind == MRX_MASK_F32, MRX_EINVAL, "mrx_llg: bad mask kind %d", mask_kind);
    if (B == 0) return MRX_OK;
    hipStream_t st = (hipStream_t)stream;
    // 1) rows: eta * S -> FFT_W                                   (rim_utils.py:44-51)
    int rc = launch_rows((const float2*)eta, (const float2*)S, (float2*)work, (long long)B * C, W, C, H, 0, norm,
                         centered, 1, st);
    if (rc) return rc;
    // 2) cols: FFT_H -> mask*(k - y) -> IFFT_H                    (rim_utils.py:51-58)
    ColArgs a;
    if ((rc = make_col_args(&a, (long long)B * C, H, W, 0, norm, centered))) return rc;
    a.scale2 = mrx_scale(H, 1, norm);
    a.C = C;
    MrxMask m;
    m.p = mask;
    m.kind = mask_kind;
    for (int i = 0; i < 4; ++i) m.s[i] = mstride[i];
    const size_t lds = sizeof(float2) * ((size_t)H + 2 * (size_t)a.ct * H);
    if (H == 640)
        rc = launch_dc_p<P640, NSEQ_COL_640>((const float2*)work, (const float2*)y, m, (float2*)work, a, lds, st);
    else if (H == 320)
        rc = launch_dc_p<P320, NSEQ_COL_320>((const float2*)work, (const float2*)y, m, (float2*)work, a, lds, st);
    else if (H == 256)
        rc = launch_dc_p<P256, NSEQ_COL_256>((const float2*)work, (const float2*)y, m, (float2*)work, a, lds, st);
    else
        rc = launch_dc_p<PlanRT, 1>((const float2*)work, (const float2*)y, m, (float2*)work, a, lds, st);
    if (rc) return rc;
    // 3) rows: IFFT_W -> sum_c conj(S) -> /sigma^2 -> [B,4,H,W]   (rim_utils.py:59-67)
    return launch_reduce((const float2*)work, (const float2*)S, (const float2*)eta, out4, B, C, H, W, norm, centered,
                         inv_sigma2, 1, st);
}

// The middle pass of mrx_llg on its own (in place on `work` = FFT_W(eta * S)): FFT_H -> mask * (k - y) -> IFFT_H.  Lets the W = 372 row passes
// run on the prime-factor kernels (mrx_pfa372_expand / mrx_pfa372_reduce) around it.
extern "C" int mrx_llg_cols_dc(float* work, const float* y, const void* mask, int mask_kind, const int64_t* mstride, int B, int C, int H, int W,
                               int norm, int centered, void* stream) {
    MRX_REQUIRE(work && y && mask && mstride, MRX_EINVAL, "mrx_llg_cols_dc: null pointer");
    MRX_REQUIRE(B >= 0 && C >= 1 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_llg_cols_dc: bad dims");
    MRX_REQUIRE(norm_valid(norm), MRX_EINVAL, "mrx_llg_cols_dc: bad normalization %d", norm);
    MRX_REQUIRE(mask_kind == MRX_MASK_U8 || mask_kind == MRX_MASK_F32, MRX_EINVAL, "mrx_llg_cols_dc: bad mask kind %d", mask_kind);
    if (B == 0) return MRX_OK;
    hipStream_t st = (hipStream_t)stream;
    ColArgs a;
    int rc;
    if ((rc = make_col_args(&a, (long long)B * C, H, W, 0, norm, centered))) return rc;
    a.scale2 = mrx_scale(H, 1, norm);
    a.C = C;
    MrxMask m;
    m.p = mask;
    m.kind = mask_kind;
    for (int i = 0; i < 4; ++i) m.s[i] = mstride[i];
    const size_t lds = sizeof(float2) * ((size_t)H + 2 * (size_t)a.ct * H);
    if (H == 640) return launch_dc_p<P640, NSEQ_COL_640>((const float2*)work, (const float2*)y, m, (float2*)work, a, lds, st);
    if (H == 320) return launch_dc_p<P320, NSEQ_COL_320>((const float2*)work, (const float2*)y, m, (float2*)work, a, lds, st);
    if (H == 256) return launch_dc_p<P256, NSEQ_COL_256>((const float2*)work, (const float2*)y, m, (float2*)work, a, lds, st);
    return launch_dc_p<PlanRT, 1>((const float2*)work, (const float2*)y, m, (float2*)work, a, lds, st);
}

// mrx_llg_cols_dc on the column-tiled coil stack (see k_cols_dc_t4): work_t4 and y_t4 are [B*C][W/4][H][4] complex, in place on work_t4.
extern "C" int mrx_llg_cols_dc_t4(float* work_t4, const float* y_t4, const void* mask, int mask_kind, const int64_t* mstride, int B, int C,
                                  int H, int W, int norm, int centered, void* stream) {
    MRX_REQUIRE(work_t4 && mask && mstride, MRX_EINVAL, "mrx_llg_cols_dc_t4: null pointer");   // y_t4 null: the pass without the measured data
    MRX_REQUIRE(B >= 0 && C >= 1 && H >= 1 && W >= 4 && W % 4 == 0, MRX_EINVAL, "mrx_llg_cols_dc_t4: bad dims (W must be a multiple of 4)");
    MRX_REQUIRE(norm_valid(norm), MRX_EINVAL, "mrx_llg_cols_dc_t4: bad normalization %d", norm);
    MRX_REQUIRE(mask_kind == MRX_MASK_U8 || mask_kind == MRX_MASK_F32, MRX_EINVAL, "mrx_llg_cols_dc_t4: bad mask kind %d", mask_kind);
    MRX_REQUIRE(H <= MRX_T4_MAX_H, MRX_EUNSUP, "mrx_llg_cols_dc_t4: H = %d above %d (the tile does not fit LDS)", H, MRX_T4_MAX_H);
    if (B == 0) return MRX_OK;
    hipStream_t st = (hipStream_t)stream;
    ColArgs a;
    int rc;
    if ((rc = make_col_args(&a, (long long)B * C, H, W, 0, norm, centered))) return rc;
    a.scale2 = mrx_scale(H, 1, norm);
    a.C = C;
    a.ct = 4;
    a.ntx = W / 4;
    a.nblocks = (long long)B * C * a.ntx;
    MRX_REQUIRE(a.nblocks < (1LL << 31), MRX_EUNSUP, "mrx_llg_cols_dc_t4: too many column tiles (%lld)", a.nblocks);
    MrxMask m;
    m.p = mask;
    m.kind = mask_kind;
    for (int i = 0; i < 4; ++i) m.s[i] = mstride[i];
    const size_t lds = sizeof(float2) * 9 * (size_t)H;
    const dim3 grid((unsigned)a.nblocks), blk(MRX_FFT_NT);
    const float4* in = (const float4*)work_t4;
    const float4* y4 = (const float4*)y_t4;
    float4* out = (float4*)work_t4;
    // MRX_COLS640=1: the wave-private register form (k_cols640_dc_t4) -- measured 33.8 us against 28.5 us for the workgroup (Stockham) form at
    // 15 x 640 x 372 (1395 single-wave tasks are 1.4 waves per SIMD: every exchange and load latency is exposed), so not the default
    static const int wave640 = (MRX_DEBUG_ENV("MRX_COLS640") && atoi(MRX_DEBUG_ENV("MRX_COLS640")) == 1) ? 1 : 0;
    if (!y4) {
        if (H == 640) {
            if ((rc = set_lds(k_cols_dc_t4<P640, true>, lds))) return rc;
            hipLaunchKernelGGL((k_cols_dc_t4<P640, true>), grid, blk, lds, st, in, y4, m, out, a);
        } else if (H == 320) {
            if ((rc = set_lds(k_cols_dc_t4<P320, true>, lds))) return rc;
            hipLaunchKernelGGL((k_cols_dc_t4<P320, true>), grid, blk, lds, st, in, y4, m, out, a);
        } else if (H == 256) {
            if ((rc = set_lds(k_cols_dc_t4<P256, true>, lds))) return rc;
            hipLaunchKernelGGL((k_cols_dc_t4<P256, true>), grid, blk, lds, st, in, y4, m, out, a);
        } else {
            if ((rc = set_lds(k_cols_dc_t4<PlanRT, true>, lds))) return rc;
            hipLaunchKernelGGL((k_cols_dc_t4<PlanRT, true>), grid, blk, lds, st, in, y4, m, out, a);
        }
    } else if (H == 640 && wave640) {
        hipLaunchKernelGGL(k_cols640_dc_t4, grid, dim3(64), 0, st, in, y4, m, out, a);
    } else if (H == 640) {
        if ((rc = set_lds(k_cols_dc_t4<P640>, lds))) return rc;
        hipLaunchKernelGGL((k_cols_dc_t4<P640>), grid, blk, lds, st, in, y4, m, out, a);
    } else if (H == 320) {
        if ((rc = set_lds(k_cols_dc_t4<P320>, lds))) return rc;
        hipLaunchKernelGGL((k_cols_dc_t4<P320>), grid, blk, lds, st, in, y4, m, out, a);
    } else if (H == 256) {
        if ((rc = set_lds(k_cols_dc_t4<P256>, lds))) return rc;
        hipLaunchKernelGGL((k_cols_dc_t4<P256>), grid, blk, lds, st, in, y4, m, out, a);
    } else {
        if ((rc = set_lds(k_cols_dc_t4<PlanRT>, lds))) return rc;
        hipLaunchKernelGGL((k_cols_dc_t4<PlanRT>), grid, blk, lds, st, in, y4, m, out, a);
    }
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
extern "C" int mrx_llg_cols_dc_t4_supported(int H, int W) { return H >= 1 && H <= MRX_T4_MAX_H && W >= 4 && W % 4 == 0; }
// x [nimg,H,W,2] row-major -> out [nimg][W/4][H][4] complex
extern "C" int mrx_tile4_cols(const float* x, float* out, int64_t nimg, int H, int W, void* stream) {
    MRX_REQUIRE(x && out, MRX_EINVAL, "mrx_tile4_cols: null pointer");
    MRX_REQUIRE(nimg >= 0 && H >= 1 && W >= 4 && W % 4 == 0, MRX_EINVAL, "mrx_tile4_cols: bad dims (W must be a multiple of 4)");
    if (nimg == 0) return MRX_OK;
    const long long total = (long long)nimg * H * W;
    long long nb = (total + 255) / 256;
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(k_tile4_cols, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const float2*)x, (float2*)out, (long long)nimg, H, W);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

template <class P, int NSEQ>
// defer != nullptr: the caller adds the coil-chunk partials itself (the fused layer-1 kernel does it in its tile loader);
// *defer = number of partial planes left in `part` (0: `out` is complete)
static int launch_hinv_p(const float2* eta, const float2* yt, const float2* S, const MrxMask& m, float* out, float* part,
                         dim3 grid, size_t lds, const ReduceArgs& a, float scale_f, hipStream_t st, int* defer = nullptr) {
    const int nchunks = mrx_cdiv(a.C, a.g);
    if (defer) *defer = 0;
    // experimental (MRX_LLG_MFMA=1): measured 46.8 us vs 44.2 us for the vector-ALU kernels below at 15 x 640 x 372 -- kept selectable
    if (P::kCT && P::N == 372 && NSEQ == M372_G && MRX_DEBUG_ENV("MRX_LLG_MFMA")) {
            M372Tables tb;
            int rc = m372_tables(&tb);  // (first call allocates: mrx_fft_prepare does it eagerly, before any graph capture)
            if (rc) return rc;
            const size_t lds_m = sizeof(float2) * (12 * 33 + 372 + 372 / 2 + (size_t)M372_G * 372) + sizeof(float) * (M372_G * M372_TC + 4 * 6 * 64);
            rc = set_lds(k_llg_rows_hinv_mfma372, lds_m);
            if (rc) return rc;
            static int n_cu = 0;
            if (!n_cu) {
                int dev = 0;
                hipDeviceProp_t prop;
                MRX_HIP(hipGetDevice(&dev));
                MRX_HIP(hipGetDeviceProperties(&prop, dev));
                n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
            }
            const long long nrows = (long long)grid.x * grid.y;
            const unsigned nblk = (unsigned)(nrows < 3ll * n_cu ? nrows : 3ll * n_cu);
            static unsigned long long* d_trace = nullptr;
            if (MRX_DEBUG_ENV("MRX_TRACE") && !d_trace) {
                (void)hipMalloc((void**)&d_trace, sizeof(unsigned long long) * 8 * 65536);
                (void)hipMemcpyToSymbol(HIP_SYMBOL(g_m372_trace), &d_trace, sizeof(d_trace));
            }
            hipLaunchKernelGGL(k_llg_rows_hinv_mfma372, dim3(nblk), dim3(MRX_FFT_NT), lds_m, st, eta, yt, S, m, out, a, scale_f, tb,
                               (int)grid.y, nchunks, MRX_DEBUG_ENV("MRX_ABLATE") ? atoi(MRX_DEBUG_ENV("MRX_ABLATE")) : 0);
            if (d_trace && (MRX_DEBUG_ENV("MRX_TRACE") && atoi(MRX_DEBUG_ENV("MRX_TRACE")) >= 2)) {
                (void)hipStreamSynchronize(st);
                const int nb = (int)nblk;
                std::vector<unsigned long long> hh((size_t)nb * 8);
                (void)hipMemcpy(hh.data(), d_trace, sizeof(unsigned long long) * 8 * nb, hipMemcpyDeviceToHost);
                double ph[7] = {0, 0, 0, 0, 0, 0, 0};
                for (int i = 0; i < nb; ++i)
                    for (int k = 0; k < 7; ++k) ph[k] += (double)(hh[(size_t)i * 8 + k + 1] - hh[(size_t)i * 8 + k]);
                fprintf(stderr, "[mrx-trace] k_llg_rows_hinv_mfma372 %d workgroups (first item), mean cycles: load+expand %.0f | A %.0f | C %.0f | "
                                "mask %.0f | A^-1 %.0f | C^-1 %.0f | conj(S) sum + store %.0f\n", nb, ph[0] / nb, ph[1] / nb, ph[2] / nb, ph[3] / nb,
                        ph[4] / nb, ph[5] / nb, ph[6] / nb);
            }
        MRX_LAUNCH_CHECK();
        return MRX_OK;
    }
    if (part && nchunks > 1) {
        dim3 g3(grid.x, grid.y, nchunks);
        if constexpr (P::kCT) {
            const size_t lds_p = sizeof(float2) * (2 * (size_t)P::N + (P::N + 1) / 2 + 2 * (size_t)NSEQ * P::N);
            int rc = set_lds(k_llg_rows_hinv_part<P, NSEQ>, lds_p);
            if (rc) return rc;
            hipLaunchKernelGGL((k_llg_rows_hinv_part<P, NSEQ>), g3, dim3(MRX_FFT_NT), lds_p, st, eta, yt, S, m, (float2*)part, a, scale_f);
        } else {
            int rc = set_lds(k_llg_rows_hinv<P, NSEQ, true>, lds);
            if (rc) return rc;
            hipLaunchKernelGGL((k_llg_rows_hinv<P, NSEQ, true>), g3, dim3(MRX_FFT_NT), lds, st, eta, yt, S, m, part, a, scale_f);
        }
        if (defer) {
            *defer = nchunks;
            MRX_LAUNCH_CHECK();
            return MRX_OK;
        }
        const long long plane = (long long)a.H * a.W, total = plane * grid.y;
        long long nb = (total + 255) / 256;
        if (nb > 2048) nb = 2048;
        hipLaunchKernelGGL(k_llg_combine, dim3((unsigned)nb), dim3(256), 0, st, eta, (const float2*)part, out, nchunks,
                           (long long)grid.y, plane, a.post);
        MRX_LAUNCH_CHECK();
        return MRX_OK;
    }
    int rc = set_lds(k_llg_rows_hinv<P, NSEQ, false>, lds);
    if (rc) return rc;
    hipLaunchKernelGGL((k_llg_rows_hinv<P, NSEQ, false>), grid, dim3(MRX_FFT_NT), lds, st, eta, yt, S, m, out, a, scale_f);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

extern "C" int mrx_fft_cols(const float* in, float* out, int64_t nimg, int H, int W, int inverse, int norm, int centered,
                            void* stream) {
    MRX_REQUIRE(nimg >= 0 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_fft_cols: bad dims");
    MRX_REQUIRE(norm_valid(norm), MRX_EINVAL, "mrx_fft_cols: bad normalization %d", norm);
    if (nimg == 0) return MRX_OK;
    MRX_REQUIRE(in && out, MRX_EINVAL, "mrx_fft_cols: null pointer");
    return launch_cols((const float2*)in, (float2*)out, nimg, H, W, inverse, norm, centered, (hipStream_t)stream);
}

extern "C" int64_t mrx_llg_hinv_work_floats(int B, int C, int H, int W) {
    if (B < 0 || C < 1 || H < 1 || W < 1) return -1;
    int g = W == 372 ? NSEQ_HINV_372 : pick_rows(W);
    const bool ct = (W == 372 || W == 320 || W == 256);
    if (!ct && g > C) g = C;
    return (int64_t)mrx_cdiv(C, g) * B * H * W * 2;
}

static int llg_hinv_impl(const float* eta, const float* yt, const float* S, const void* mask, int mask_kind,
                         const int64_t* mstride, float* out4, float* work, int B, int C, int H, int W, float inv_sigma2,
                         int norm, int centered, void* stream, int* defer) {
    MRX_REQUIRE(eta && yt && S && mask && mstride && out4, MRX_EINVAL, "mrx_llg_hinv: null pointer");
    MRX_REQUIRE(B >= 0 && C >= 1 && H >= 1 && W >= 1, MRX_EINVAL, "mrx_llg_hinv: bad dims");
    MRX_REQUIRE(norm_valid(norm), MRX_EINVAL, "mrx_llg_hinv: bad normalization %d", norm);
    MRX_REQUIRE(mask_kind == MRX_MASK_U8 || mask_kind == MRX_MASK_F32, MRX_EINVAL, "mrx_llg_hinv: bad mask kind %d", mask_kind);
    MRX_REQUIRE(mstride[2] == 0, MRX_EINVAL, "mrx_llg_hinv: the mask must not depend on the row index (mstride[2] = %lld)",
                (long long)mstride[2]);
    MRX_REQUIRE(B <= 65535, MRX_EUNSUP, "mrx_llg_hinv: batch %d too large", B);
    if (B == 0) return MRX_OK;
    MrxFftEntry e;
    int rc = mrx_get_plan(W, &e);
    if (rc) return rc;
    ReduceArgs a;
    a.plan = e.plan;
    a.tw = e.d_tw;
    a.sdiv = 1;
    a.C = C;
    a.H = H;
    a.W = W;
    const bool mfma372 = W == 372 && MRX_DEBUG_ENV("MRX_LLG_MFMA");  // experimental matrix-pipe kernel: fixed at 5 coils per workgroup
    a.g = W == 372 ? (mfma372 ? M372_G : NSEQ_HINV_372) : pick_rows(W);
    const bool ct = (W == 372 || W == 320 || W == 256);
    if (!ct && a.g > C) a.g = C;
    a.halfW = centered ? W / 2 : 0;
    a.scale = mrx_scale(W, 1, norm);
    a.post = inv_sigma2;
    const float scale_f = mrx_scale(W, 0, norm);
    MrxMask m;
    m.p = mask;
    m.kind = mask_kind;
    for (int i = 0; i < 4; ++i) m.s[i] = mstride[i];
    const size_t lds = sizeof(float2) * (3 * (size_t)W + 3 * (size_t)a.g * W);
    dim3 grid(H, B);
    hipStream_t st = (hipStream_t)stream;
    const float2 *pe = (const float2*)eta, *py = (const float2*)yt, *ps = (const float2*)S;
    // split the coil sum over workgroups only while the grid is small (rows x batch below ~4 workgroups per CU)
    float* part = (work && (long long)H * B < 1024) ? work : nullptr;
    if (W == 372 && mfma372) return launch_hinv_p<P372, M372_G>(pe, py, ps, m, out4, part, grid, lds, a, scale_f, st, nullptr);
    if (W == 372) return launch_hinv_p<P372, NSEQ_HINV_372>(pe, py, ps, m, out4, part, grid, lds, a, scale_f, st, defer);
    if (W == 320) return launch_hinv_p<P320, NSEQ_ROW_320>(pe, py, ps, m, out4, part, grid, lds, a, scale_f, st, defer);
    if (W == 256) return launch_hinv_p<P256, NSEQ_ROW_256>(pe, py, ps, m, out4, part, grid, lds, a, scale_f, st, defer);
    return launch_hinv_p<PlanRT, 1>(pe, py, ps, m, out4, part, grid, lds, a, scale_f, st, defer);
}

extern "C" int mrx_llg_hinv(const float* eta, const float* yt, const float* S, const void* mask, int mask_kind,
                            const int64_t* mstride, float* out4, float* work, int B, int C, int H, int W, float inv_sigma2,
                            int norm, int centered, void* stream) {
    return llg_hinv_impl(eta, yt, S, mask, mask_kind, mstride, out4, work, B, C, H, W, inv_sigma2, norm, centered, stream, nullptr);
}
// Same, but the sum over the coil-chunk partials is left to the consumer: *nparts (host) = number of partial planes
// work[k][B][H][W][2] still to be added and scaled by inv_sigma2 (0: out4 is complete).  mrx_rim_layer_indrnn_packed_llg consumes them.
extern "C" int mrx_llg_hinv_parts(const float* eta, const float* yt, const float* S, const void* mask, int mask_kind,
                                  const int64_t* mstride, float* out4, float* work, int* nparts, int B, int C, int H, int W,
                                  float inv_sigma2, int norm, int centered, void* stream) {
    MRX_REQUIRE(nparts, MRX_EINVAL, "mrx_llg_hinv_parts: null pointer");
    *nparts = 0;
    if (MRX_DEBUG_ENV("MRX_LLG_MFMA")) return llg_hinv_impl(eta, yt, S, mask, mask_kind, mstride, out4, work, B, C, H, W, inv_sigma2, norm, centered, stream, nullptr);
    return llg_hinv_impl(eta, yt, S, mask, mask_kind, mstride, out4, work, B, C, H, W, inv_sigma2, norm, centered, stream, nparts);
}

// Data-consistency residual in image space with shared maps:  out[b] = sum_c conj(S[b/sdiv,c]) * ifft2( mask * (fft2(x[b] * S[b/sdiv,c]) - y[b,c]) )
// (the linear-operator core of quantitative/models/qrim/utils.py:235-248; b runs over batch x echoes, sdiv = echoes).
extern "C" int mrx_dc_residual(const float* x, const float* y, const float* S, const void* mask, int mask_kind,
                               const int64_t* mstride, float* out, float* work, int B, int C, int H, int W, int sdiv, int norm,
                               int centered, void* stream) {
    MRX_REQUIRE(x && y && S && mask && mstride && out && work, MRX_EINVAL, "mrx_dc_residual: null pointer");
    MRX_REQUIRE(B >= 0 && C >= 1 && H >= 1 && W >= 1 && sdiv >= 1, MRX_EINVAL, "mrx_dc_residual: bad dims");
    MRX_REQUIRE(norm_valid(norm), MRX_EINVAL, "mrx_dc_residual: bad normalization %d", norm);
    MRX_REQUIRE(mask_kind == MRX_MASK_U8 || mask_kind == MRX_MASK_F32, MRX_EINVAL, "mrx_dc_residual: bad mask kind %d", mask_kind);
    if (B == 0) return MRX_OK;
    hipStream_t st = (hipStream_t)stream;
    int rc = launch_rows((const float2*)x, (const float2*)S, (float2*)work, (long long)B * C, W, C, H, 0, norm, centered, 1, st, sdiv);
    if (rc) return rc;
    ColArgs a;
    if ((rc = make_col_args(&a, (long long)B * C, H, W, 0, norm, centered))) return rc;
    a.scale2 = mrx_scale(H, 1, norm);
    a.C = C;
    MrxMask m;
    m.p = mask;
    m.kind = mask_kind;
    for (int i = 0; i < 4; ++i) m.s[i] = mstride[i];
    const size_t lds = sizeof(float2) * ((size_t)H + 2 * (size_t)a.ct * H);
    if (H == 640)
        rc = launch_dc_p<P640, NSEQ_COL_640>((const float2*)work, (const float2*)y, m, (float2*)work, a, lds, st);
    else if (H == 320)
        rc = launch_dc_p<P320, NSEQ_COL_320>((const float2*)work, (const float2*)y, m, (float2*)work, a, lds, st);
    else if (H == 256)
        rc = launch_dc_p<P256, NSEQ_COL_256>((const float2*)work, (const float2*)y, m, (float2*)work, a, lds, st);
    else
        rc = launch_dc_p<PlanRT, 1>((const float2*)work, (const float2*)y, m, (float2*)work, a, lds, st);
    if (rc) return rc;
    return launch_reduce((const float2*)work, (const float2*)S, nullptr, out, B, C, H, W, norm, centered, 1.0f, 0, st, sdiv);
}
