// llg372.hip -- log_likelihood_gradient (reference models/rim/rim_utils.py:11-67) for row-invariant masks at the fastMRI knee width
// W = 372, one launch per RIM step, wave-private prime-factor transforms (pfa372.h).
//
// With yt = IFFT_H(y) the gradient is row transforms only (fft.hip: the H transforms cancel for a mask that does not depend on the
// row):   g[h, :] = sum_c conj(S_c) * IFFT_W( m * (FFT_W(eta * S_c) - yt_c) ).
// The Stockham kernel this replaces (k_llg_rows_hinv_part<P372>) spent its time in workgroup barriers and in a radix-31 stage that
// ran 48 butterflies on 256 threads by giving every wave the same pre-additions (0.22 of the HBM roofline).  Here one wavefront owns
// five coil rows end to end:
//   * 60 lanes each run ONE whole 31-point DFT on registers (1050 VALU operations, no redundancy, 94 % of the lanes busy);
//   * the 155 12-point DFTs, the data-consistency step and the inverse 12-point DFTs run in registers, 64 at a time;
//   * the two transposes between them go through 17 KB of wave-private LDS -- no __syncthreads between waves, no twiddle table;
//   * S, yt and the mask are read in the order the lanes consume them (laid out once per slice by mrx_llg372_prepare), so every
//     global access is one contiguous row per wave instruction and S stays in registers from the expand to the reduce.
// Eight single-wave workgroups per CU; the 1920 tasks of a 15-coil 640-row slice are one dispatch round.
#include <cstdlib>

#include "mrx_common.h"
#include "pfa372.h"
// cache policy of the sensitivity-map stream (nt: streaming): 28.6 MB per slice re-read by every RIM step, each time behind > 1 GB of other traffic at 8 slices per
// launch -- it never hits the 256 MB memory-side cache and only evicts what the next launch is about to read.  Round 6, A/B builds alternating on one box
// (tools/runs/r06h.sh): fp32-class headline 154.8 / 154.0 without, 154.3 / 154.2 with (neutral); precision-16 line 297.2 / 295.3 without, 300.0 / 298.7 with (+1 %): on.
#ifndef MRX_LLG_NT_MAPS
#define MRX_LLG_NT_MAPS 1
#endif

#define L372_TASK_C2 (PFA_N * PFA_G)   // 1860 float2 per task in Sp and in ytp
#define L372_LDS_BYTES (sizeof(float2) * PFA_LDS_C2 + sizeof(float) * PFA_N)

// diagnostics (MRX_LLG372_ABLATE=3): s_memtime stamps of every wave's phases
__device__ unsigned long long* g_l372_trace = nullptr;
#define L372_STAMP(i) \
    if (ABL == 3 && l == 0) g_l372_trace[(size_t)blockIdx.x * 8 + (i)] = __builtin_readcyclecounter();

struct L372Args {
    int B, C, H, T;       // batch, coils, rows, tasks (groups of 5 coils) per row
    int halfW;            // 186 for centred transforms (ifftshift / fftshift folded into the index maps), else 0
    int mask_bstride;     // floats between the batch entries of maskp (0: one mask for the whole batch)
    long long ntasks;     // B * H * T
    float scale_f, scale_i;
    int tiled;            // row kernels of the general-mask gradient: the coil stack is column-tiled (fft.hip: [B*C][93][H][4]), not row-major
};
// element (image bc, row h, column w) of the coil stack between the row and the column pass
// = l372_kbase(image bc, row h) + l372_kcol(column w)
__device__ __forceinline__ long long l372_kbase(const L372Args& a, long long bc, int h) {
    return a.tiled ? bc * ((long long)a.H * PFA_N) + (long long)h * 4 : (bc * a.H + h) * PFA_N;
}
__device__ __forceinline__ int l372_kcol(const L372Args& a, int w) { return a.tiled ? (w >> 2) * (a.H * 4) + (w & 3) : w; }

// ---- once per slice: operands in lane order ------------------------------------------------------------------------------------------
__global__ void k_llg372_prep(const float2* __restrict__ yt, const float2* __restrict__ S, float2* __restrict__ ytp,
                              float2* __restrict__ Sp, L372Args a) {
    const long long total = a.ntasks * L372_TASK_C2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long task = i / L372_TASK_C2;
        const int e = (int)(i - task * L372_TASK_C2);
        const long long row = task / a.T;
        const int z = (int)(task - row * a.T);
        const long long b = row / a.H, h = row - b * a.H;
        int g, w;
        {
            const int n2 = e / PFA_L1, lane = e - n2 * PFA_L1;
            pfa372_sp_src(n2, lane, a.halfW, &g, &w);
            const int c = z * PFA_G + g;
            Sp[i] = c < a.C ? S[((b * a.C + c) * a.H + h) * PFA_N + w] : make_float2(0.f, 0.f);
        }
        {
            // element (k1, d) of the task sits at ((k1 / 2) * 155 + d) * 2 + k1 % 2: a lane's values for k1 = 2 j, 2 j + 1 are one 16-byte load
            const int k1 = e / PFA_D, d = e - k1 * PFA_D;
            pfa372_yt_src(k1, d, a.halfW, &g, &w);
            const int c = z * PFA_G + g;
            ytp[task * L372_TASK_C2 + ((k1 >> 1) * PFA_D + d) * 2 + (k1 & 1)] =
                c < a.C ? yt[((b * a.C + c) * a.H + h) * PFA_N + w] : make_float2(0.f, 0.f);
        }
    }
}
// The same permutation with the task's rows through LDS: one workgroup per task reads the five coil rows of S and of yt as whole 2976-byte rows (the
// element-wise form above gathers every value from a line of its own: 1.2 TB/s for a kernel that runs once per slice -- 96 us of a 6.9-ms slice at 15 x 640 x 372)
// and writes the lane-ordered operands from there.  Pure data movement: bit-identical.
__global__ __launch_bounds__(256) void k_llg372_prep_rows(const float2* __restrict__ yt, const float2* __restrict__ S, float2* __restrict__ ytp, float2* __restrict__ Sp,
                                                         L372Args a) {
    __shared__ float2 rs[PFA_G][PFA_N], ry[PFA_G][PFA_N];
    const long long task = blockIdx.x;
    const long long row = task / a.T;
    const int z = (int)(task - row * a.T);
    const long long b = row / a.H, h = row - b * a.H;
    for (int i = threadIdx.x; i < PFA_G * PFA_N; i += 256) {
        const int g = i / PFA_N, w = i - g * PFA_N, c = z * PFA_G + g;
        const long long src = ((b * a.C + (c < a.C ? c : a.C - 1)) * a.H + h) * PFA_N + w;
        const float2 vs = S[src], vy = yt[src];
        rs[g][w] = c < a.C ? vs : make_float2(0.f, 0.f);
        ry[g][w] = c < a.C ? vy : make_float2(0.f, 0.f);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < L372_TASK_C2; e += 256) {
        int g, w;
        {
            const int n2 = e / PFA_L1, lane = e - n2 * PFA_L1;
            pfa372_sp_src(n2, lane, a.halfW, &g, &w);
            Sp[task * L372_TASK_C2 + e] = rs[g][w];
        }
        {
            // (output-major here: e is the position in ytp, element (k1, d) sits at ((k1 / 2) * 155 + d) * 2 + k1 % 2)
            const int k1 = 2 * (e / (2 * PFA_D)) + (e & 1), d = (e % (2 * PFA_D)) >> 1;
            pfa372_yt_src(k1, d, a.halfW, &g, &w);
            ytp[task * L372_TASK_C2 + e] = ry[g][w];
        }
    }
}
__global__ void k_llg372_prep_mask(MrxMask mask, float* __restrict__ maskp, int nb, int halfW) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb * PFA_N) return;
    const int b = i / PFA_N, e = i - b * PFA_N;
    const int k1 = e / PFA_N2, k2 = e - k1 * PFA_N2;
    maskp[i] = mrx_mask_val(mask, b, 0, 0, pfa372_mask_src(k1, k2, halfW));
}

// ---- the per-step kernel: one wavefront = one task (row, five coils) ----------------------------------------------------------------
// ABL (diagnostics, MRX_LLG372_ABLATE): 0 the kernel; 1 memory only (loads, LDS staging, reduce, stores: no transforms); 2 compute only
// (operands not loaded from HBM); 3 the kernel with s_memtime stamps per phase (tools/probe/llg372_trace.py).  Measured at 15 x 640 x 372:
// memory only 11.1 us, compute only 12.3 us, the kernel 17.3 us (rocprofv3) -- every wave waits ~8 us for its first operands (the whole
// launch requests its 63 MB at once) and the two waves of a SIMD then share the vector ALU for ~8.7 us.
// NOY: the measured data is not read -- the gradient is affine in eta, g = A^H M A eta - A^H M y, and the second term is ONE constant plane per slice
// (mrx_llg372_const_plane: this kernel on eta = 0, its partial planes summed) that the consumer of the partial planes adds like one more coil group:
// 34.5 MB per launch instead of 63.1 MB at 15 coils.  The same kernel with NOY is the ADJOINT of the linear part (training).
// GAT: eta is not read but MADE here -- the previous step's eta plus the nine-tap gather of the final convolution's tap products (the whole of
// k_l2sb_gather, same order of additions: bit-identical) -- and written out by the row's first task; one launch per step less in the RIM loop.
struct L372Gather {
    const float* taps;      // [B][18][H][372]: taps[b][tap * 2 + co] (rim_layer2_sb.hip) -- or, with `edges`, the row-pre-summed planes [B][3][H][372][2] (kernel row dy, pair (co 0, co 1))
    const float* bias;      // [2] or null
    float2* eta_out;        // [B][H][372]
    const float* edges = nullptr;   // not null: taps are mrx_rim_layer2_f16_cb8_q's; edges [B][H][12][16] what the neighbouring 32-pixel tiles owe columns 0 / 31
};
template <int ABL, bool NOY = false, bool GAT = false>
__global__ __launch_bounds__(64, 2) void k_llg372(const float2* __restrict__ eta_, const float2* __restrict__ ytp_,
                                                   const float2* __restrict__ Sp_, const float* __restrict__ maskp,
                                                   float2* __restrict__ part_, L372Args a, L372Gather ga) {
    // complex values travel as packed register pairs (pfa_c); the float2 of the interface is the same 8 bytes
    const pfa_c* __restrict__ eta = reinterpret_cast<const pfa_c*>(eta_);
    const pfa_c* __restrict__ ytp = reinterpret_cast<const pfa_c*>(ytp_);
    const pfa_c* __restrict__ Sp = reinterpret_cast<const pfa_c*>(Sp_);
    pfa_c* __restrict__ part = reinterpret_cast<pfa_c*>(part_);
    extern __shared__ __attribute__((aligned(16))) float2 X_[];
    pfa_c* X = reinterpret_cast<pfa_c*>(X_);
    float* Mk = reinterpret_cast<float*>(X + PFA_LDS_C2);
    const int l = threadIdx.x;
    L372_STAMP(0)
    // every XCD walks one contiguous band of tasks: the tasks of an image row (which share the eta row) meet in one L2
    const unsigned task = (unsigned)mrx_xcd_band(blockIdx.x, a.ntasks);      // ntasks < 2^31 (checked by the launcher)
    const unsigned row = task / (unsigned)a.T;
    const int z = (int)(task - row * (unsigned)a.T);
    const unsigned b = row / (unsigned)a.H;
    const int Cg = min(PFA_G, a.C - z * PFA_G);
    const bool laneA = l < PFA_L1;
    const int g1 = l / PFA_N1, n1 = l - g1 * PFA_N1;

    // all global operands of the task requested up front, unconditionally (idle lanes repeat a neighbour's address): eta row + mask
    // (6 + 6 loads), S (31), first yt pass (12)
    pfa_c ev[6];
    float mv[6];
    const pfa_c* erow = eta + (long long)row * PFA_N;
    const float* mrow = maskp + (long long)b * a.mask_bstride;
    Pfa372Lane L;
    auto load_maps = [&]() {
        const pfa_c* sp = Sp + (long long)task * L372_TASK_C2 + min(l, PFA_L1 - 1);
#pragma unroll
        for (int n2 = 0; n2 < 31; ++n2)
            if (ABL == 2) L.s[n2] = pfa_mk((float)(l + n2), 0.25f);
            else if (MRX_LLG_NT_MAPS) {
                typedef float llg_f2 __attribute__((ext_vector_type(2)));
                const llg_f2 u = __builtin_nontemporal_load(reinterpret_cast<const llg_f2*>(sp + n2 * PFA_L1));
                L.s[n2] = pfa_mk(u.x, u.y);
            } else L.s[n2] = sp[n2 * PFA_L1];
    };
    if (GAT) load_maps();       // the maps are requested BEFORE the gather's 114 dependent loads: they arrive under it
    if (GAT) {
        // All 6 x 19 loads of the row go out before the first sum (GAT_GROUP = 6; 3: two rounds), and eta is written after the last one: with the
        // store of value i between the loads of i and i + 1 (it may alias them for all the compiler knows) the six groups of 19 loads ran one
        // memory round trip after the other -- 8 of the task's 23 us at 8 slices per launch.  Same order of additions per pixel: bit-identical.
        const int h = (int)(row - b * (unsigned)a.H);
        const long long plane = (long long)a.H * PFA_N;
        const float* __restrict__ pb = ga.taps + (long long)b * 18 * plane;
        const int y0 = h > 0 ? h - 1 : 0, y2 = h + 1 < a.H ? h + 1 : a.H - 1;
        const float b0 = ga.bias ? ga.bias[0] : 0.f, b1 = ga.bias ? ga.bias[1] : 0.f;
        constexpr int GAT_GROUP = 6;
        float2 vout[6];
        if (ga.edges) {
            // the row-pre-summed form (mrx_rim_layer2_f16_cb8_q): 6 plane values + 3 edge pairs + eta per pixel instead of 18 + eta; the additions in the order of
            // l2sb_gather_q_px (rim_layer2_sb.hip): bias, the three rows' plane values, the three rows' edge terms -- bit-identical to mrx_rim_final_gather_q
            const float* __restrict__ qb = ga.taps + (long long)b * 6 * plane;
            constexpr int TX = (PFA_N + 31) / 32;
            const float* __restrict__ eb = ga.edges + (long long)b * 16 * a.H * TX;
            float q[6][3][2];
            float2 ee[6][3], e[6];
            bool edge[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int n = min(l + 64 * i, PFA_N - 1);
                const int w = pfa372_shift(n, a.halfW);
                const int xt = w >> 5, xl = w & 31;
                const bool fromR = xl == 31 && w + 1 < PFA_N, fromL = xl == 0 && w > 0;
                const int et = fromR ? xt + 1 : (fromL ? xt - 1 : xt);
                const int oo[3] = {fromR ? 4 : 8, fromR ? 0 : 12, fromR ? 2 : 14};
                const int yy[3] = {y0, h, y2};
                edge[i] = fromR || fromL;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const float2 qq = *reinterpret_cast<const float2*>(qb + ((long long)dy * plane + (long long)yy[dy] * PFA_N + w) * 2);
                    q[i][dy][0] = qq.x, q[i][dy][1] = qq.y;
                    ee[i][dy] = *reinterpret_cast<const float2*>(eb + ((long long)yy[dy] * TX + et) * 16 + oo[dy]);
                }
                e[i] = eta_[(long long)row * PFA_N + w];
                mv[i] = mrow[n];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                float s0 = b0, s1 = b1;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) s0 += q[i][dy][0], s1 += q[i][dy][1];
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) s0 += edge[i] ? ee[i][dy].x : 0.f, s1 += edge[i] ? ee[i][dy].y : 0.f;
                vout[i] = make_float2(e[i].x + s0, e[i].y + s1);
                ev[i] = pfa_mk(vout[i].x, vout[i].y);
            }
        } else
#pragma unroll
        for (int i0 = 0; i0 < 6; i0 += GAT_GROUP) {
            float t[GAT_GROUP][18];
            float2 e[GAT_GROUP];
#pragma unroll
            for (int ii = 0; ii < GAT_GROUP; ++ii) {
                const int i = i0 + ii;
                const int n = min(l + 64 * i, PFA_N - 1);
                const int w = pfa372_shift(n, a.halfW);
                const int x0 = w > 0 ? w - 1 : 0, x2 = w + 1 < PFA_N ? w + 1 : PFA_N - 1;
                const int ro[3] = {y0 * PFA_N, h * PFA_N, y2 * PFA_N}, co[3] = {x0, w, x2};
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const float* q = pb + (long long)((dy * 3 + dx) * 2) * plane + ro[dy] + co[dx];
                        t[ii][(dy * 3 + dx) * 2] = q[0];
                        t[ii][(dy * 3 + dx) * 2 + 1] = q[plane];
                    }
                e[ii] = eta_[(long long)row * PFA_N + w];
                mv[i] = mrow[n];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ii = 0; ii < GAT_GROUP; ++ii) {
                float s0 = b0, s1 = b1;
#pragma unroll
                for (int k = 0; k < 9; ++k) s0 += t[ii][2 * k], s1 += t[ii][2 * k + 1];
                vout[i0 + ii] = make_float2(e[ii].x + s0, e[ii].y + s1);
                ev[i0 + ii] = pfa_mk(vout[i0 + ii].x, vout[i0 + ii].y);
            }
        }
        if (z == 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i)
                if (l + 64 * i < PFA_N) ga.eta_out[(long long)row * PFA_N + pfa372_shift(l + 64 * i, a.halfW)] = vout[i];
        }
    } else {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int n = min(l + 64 * i, PFA_N - 1);
        ev[i] = ABL == 2 ? pfa_mk(0.5f, (float)n) : erow[pfa372_shift(n, a.halfW)];
        mv[i] = ABL == 2 ? 1.f : mrow[n];
    }
    }
    if (!GAT) load_maps();
    // yt of the first two passes of stage B is requested before stage A, the third as soon as stage A has freed its registers: the
    // whole HBM stream of the task is in flight while the 31-point DFTs run.  (Requesting yt only after the maps have arrived, or
    // delaying it by a fixed sleep, changes nothing: measured 18.6 vs 18.7 us -- the memory system is not first-come-first-served.)
    const pfa_c* ytask = ytp + (long long)task * L372_TASK_C2;
    pfa_c yv0[12], yv1[12], yv2[12];
    const int d2 = min(l + 128, PFA_D - 1);
    const int yoff = l;
    auto ldy = [&](pfa_c (&v)[12], int d) {           // six 16-byte loads: (k1 = 2 j, 2 j + 1) are adjacent in ytp
        const float4* q = reinterpret_cast<const float4*>(ytask) + d;
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float4 t = q[j * PFA_D];
            v[2 * j] = pfa_mk(t.x, t.y);
            v[2 * j + 1] = pfa_mk(t.z, t.w);
        }
    };
    if (NOY) {
#pragma unroll
        for (int k1 = 0; k1 < 12; ++k1) yv0[k1] = yv1[k1] = yv2[k1] = pfa_mk(0.f, 0.f);
    } else if (ABL == 2) {
#pragma unroll
        for (int k1 = 0; k1 < 12; ++k1) yv0[k1] = pfa_mk((float)k1, 1.f), yv1[k1] = pfa_mk((float)k1, 2.f);
    } else {
        ldy(yv0, yoff);
        ldy(yv1, yoff + 64);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int n = l + 64 * i;
        if (n < PFA_N) {
            X[n] = ev[i];
            if (n < PFA_ETA_C2 - PFA_N) X[n + PFA_N] = ev[i];     // second copy: the expand reads unwrapped indices
            Mk[n] = mv[i];
        }
    }
    __syncthreads();
    L372_STAMP(1)
    if (ABL == 1) {   // memory only: every loaded value still reaches the output
        pfa_c acc = pfa_mk(0.f, 0.f);
#pragma unroll
        for (int n2 = 0; n2 < 31; ++n2) acc = pfa_add(acc, L.s[n2]);
#pragma unroll
        for (int k1 = 0; k1 < 12; ++k1) acc = pfa_add(acc, pfa_add(yv0[k1], yv1[k1]));
        {
            pfa_c t2[12];
            ldy(t2, d2);
#pragma unroll
            for (int k1 = 0; k1 < 12; ++k1) acc = pfa_add(acc, t2[k1]);
        }
        __syncthreads();
        X[PFA_RS + l] = acc;
        __syncthreads();
    } else {
    if (laneA) pfa372_expand(L, X, n1);
    __syncthreads();
    L372_STAMP(2)
    if (laneA) pfa372_stage_a(L, X, g1, n1);
    L372_STAMP(3)
    if (NOY) {
    } else if (ABL == 2) {
#pragma unroll
        for (int k1 = 0; k1 < 12; ++k1) yv2[k1] = pfa_mk((float)k1, 3.f);
    } else {
        ldy(yv2, d2);
    }
    __syncthreads();
    {
        const int nd = Cg * PFA_N2;
        const int ga = l / PFA_N2, gb = (l + 64) / PFA_N2, gc = (l + 128) / PFA_N2;
        if (l < nd) pfa372_stage_b(X, Mk, yv0, ga, l - ga * PFA_N2, a.scale_f);
        if (l + 64 < nd) pfa372_stage_b(X, Mk, yv1, gb, l + 64 - gb * PFA_N2, a.scale_f);
        if (l + 128 < nd) pfa372_stage_b(X, Mk, yv2, gc, l + 128 - gc * PFA_N2, a.scale_f);
    }
    __syncthreads();
    L372_STAMP(4)
    if (laneA) pfa372_gather_a(L, X, g1, n1);
    __syncthreads();
    if (laneA) pfa372_stage_a_inv(L, X, g1, n1, a.scale_i);
    __syncthreads();
    L372_STAMP(5)
    }
    pfa_c* po = part + (((long long)z * a.B * a.H) + row) * PFA_N;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int n = l + 64 * i;
        if (n < PFA_N) {
            pfa_c s = X[n];
#pragma unroll
            for (int g = 1; g < PFA_G; ++g) s = pfa_add(s, X[g * PFA_RS + n]);
            po[pfa372_shift(n, a.halfW)] = s;
        }
    }
    L372_STAMP(6)
}

// ---- the two halves of the pipeline as stand-alone row operators (natural k-space layout on the far side) ----------------------------------
// k_pfa372_expand:  out[b, c, h, :] = FFT_W(x[b, h, :] * S[b, c, h, :]) * scale_f            (sens_expand restricted to the W transform:
//                   vn_block.py:51-69 in the hybrid space of fft.hip, and the first pass of the general log_likelihood_gradient)
//                   DC epilogue (a.dc): out = pred - where(mask, pred - ref, 0) * w - that      (vn_block.py:109-119)
// k_pfa372_reduce:  part[z][b, h, :] = sum over the task's coils of conj(S) * IFFT_W(k[b, c, h, :]) * scale_i      (vn_block.py:71-87, the
//                   last pass of log_likelihood_gradient); the T partial planes are added by k_pfa372_sum / k_llg372_combine
// Same wave-private transforms and lane-ordered maps (Sp) as k_llg372; the k-space rows are read / written contiguously through the
// wave's LDS buffer, in transform order [coil][k].
struct L372Dc {
    const float2* pred;
    const float2* ref;
    const float* w;     // dc_weight (device scalar)
    MrxMask mask;
    int on;
};

// RED: the new k-space rows stay in the wave's buffer and go straight through the inverse pipeline of k_pfa372_reduce (S is still in
// registers): part[z][b, h, :] = the task's share of sum_c conj(S) IFFT_W(out) -- the next cascade's sens_reduce (vn_block.py:71-87) without
// reading the coil stack and the maps again.
// DC: the data-consistency epilogue compiled in (E2EVN's cascades); without it the pass is the first pass of the general-mask gradient and of sens_expand -- a
// run-time switch cost that form 1.8 us per launch (registers of the operand prefetch, the unrolled coil loop's exits)
// GAT (round 5; general-mask gradient): x is not read but MADE here -- the previous step's eta plus the nine-tap gather of the final convolution's tap
// products (the block of k_llg372<.., GAT>, same order of additions: bit-identical to k_l2sb_gather) -- and written out by the row's first task.
template <bool RED, bool DC, bool GAT = false>
__global__ __launch_bounds__(64, 2) void k_pfa372_expand(const float2* __restrict__ x_, const float2* __restrict__ Sp_,
                                                          float2* __restrict__ out_, L372Args a, L372Dc dc, float2* __restrict__ part_,
                                                          L372Gather ga = L372Gather{nullptr, nullptr, nullptr}) {
    const pfa_c* __restrict__ xin = reinterpret_cast<const pfa_c*>(x_);
    const pfa_c* __restrict__ Sp = reinterpret_cast<const pfa_c*>(Sp_);
    extern __shared__ __attribute__((aligned(16))) float2 X_[];
    pfa_c* X = reinterpret_cast<pfa_c*>(X_);
    const int l = threadIdx.x;
    const unsigned task = (unsigned)mrx_xcd_band(blockIdx.x, a.ntasks);
    const unsigned row = task / (unsigned)a.T;
    const int z = (int)(task - row * (unsigned)a.T);
    const unsigned b = row / (unsigned)a.H, h = row - b * (unsigned)a.H;
    const int Cg = min(PFA_G, a.C - z * PFA_G);
    const bool laneA = l < PFA_L1;
    const int g1 = l / PFA_N1, n1 = l - g1 * PFA_N1;
    pfa_c ev[6];
    const pfa_c* erow = xin + (long long)row * PFA_N;
    Pfa372Lane L;
    auto load_maps = [&]() {
        const pfa_c* sp = Sp + (long long)task * L372_TASK_C2 + min(l, PFA_L1 - 1);
#pragma unroll
        for (int n2 = 0; n2 < 31; ++n2) L.s[n2] = sp[n2 * PFA_L1];
    };
    if constexpr (GAT) {
        load_maps();            // requested BEFORE the gather's 114 loads: they arrive under it
        const long long plane = (long long)a.H * PFA_N;
        const float* __restrict__ pb = ga.taps + (long long)b * 18 * plane;
        const int hh = (int)h, y0 = hh > 0 ? hh - 1 : 0, y2 = hh + 1 < a.H ? hh + 1 : a.H - 1;
        const float b0 = ga.bias ? ga.bias[0] : 0.f, b1 = ga.bias ? ga.bias[1] : 0.f;
        float t[6][18];
        float2 e[6], vout[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {           // all 6 x 19 loads of the row before the first sum, eta written after the last one (k_llg372's note)
            const int n = min(l + 64 * i, PFA_N - 1);
            const int w = pfa372_shift(n, a.halfW);
            const int x0 = w > 0 ? w - 1 : 0, x2 = w + 1 < PFA_N ? w + 1 : PFA_N - 1;
            const int ro[3] = {y0 * PFA_N, hh * PFA_N, y2 * PFA_N}, co[3] = {x0, w, x2};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float* q = pb + (long long)((dy * 3 + dx) * 2) * plane + ro[dy] + co[dx];
                    t[i][(dy * 3 + dx) * 2] = q[0];
                    t[i][(dy * 3 + dx) * 2 + 1] = q[plane];
                }
            e[i] = x_[(long long)row * PFA_N + w];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            float s0 = b0, s1 = b1;
#pragma unroll
            for (int k = 0; k < 9; ++k) s0 += t[i][2 * k], s1 += t[i][2 * k + 1];
            vout[i] = make_float2(e[i].x + s0, e[i].y + s1);
            ev[i] = pfa_mk(vout[i].x, vout[i].y);
        }
        if (z == 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i)
                if (l + 64 * i < PFA_N) ga.eta_out[(long long)row * PFA_N + pfa372_shift(l + 64 * i, a.halfW)] = vout[i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 6; ++i) ev[i] = erow[pfa372_shift(min(l + 64 * i, PFA_N - 1), a.halfW)];
        load_maps();
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int n = l + 64 * i;
        if (n < PFA_N) {
            X[n] = ev[i];
            if (n < PFA_ETA_C2 - PFA_N) X[n + PFA_N] = ev[i];
        }
    }
    __syncthreads();
    if (laneA) pfa372_expand(L, X, n1);
    __syncthreads();
    if (laneA) pfa372_stage_a(L, X, g1, n1);
    __syncthreads();
    // forward 12-point DFTs: all three passes read their inputs first (the natural-order result aliases the exchange buffer)
    pfa_c v[3][12];
    int kb[3];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const int d = min(l + 64 * p, PFA_D - 1), g2 = d / PFA_N2, k2 = d - g2 * PFA_N2;
        const pfa_c* q = X + g2 * PFA_GS + k2 * PFA_KS;
#pragma unroll
        for (int i = 0; i < 12; ++i) v[p][i] = q[i];
        kb[p] = g2 * PFA_RS + (156 * k2) % PFA_N;           // natural-order base of this lane's outputs: k = (217 k1 + 156 k2) mod 372
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        if (l + 64 * p < Cg * PFA_N2) {
            pfa_dft12<false>(v[p]);
            const int g2 = (l + 64 * p) / PFA_N2, base = kb[p] - g2 * PFA_RS;
#pragma unroll
            for (int k1 = 0; k1 < 12; ++k1) {
                int k = base + (217 * k1) % PFA_N;
                k = k >= PFA_N ? k - PFA_N : k;
                X[g2 * PFA_RS + k] = pfa_scale(v[p][k1], a.scale_f);
            }
        }
    }
    __syncthreads();
    const float w = DC ? dc.w[0] : 0.f;
    // data-consistency operands of coil g + 1 (prediction, reference, mask: 18 loads) are requested before coil g is combined and stored: with the loads
    // of a coil behind the stores of the one before, the task paid one memory round trip per coil at its very end
    // (the mask as raw bits, its kind tested once per coil: mrx_mask_val's conversion inside the branch on the kind made every mask load wait for
    // everything in flight -- six more round trips per coil)
    float2 pp[2][6], rr[2][6];
    unsigned mm[2][6];
    const bool mask_u8 = dc.mask.kind == MRX_MASK_U8;
    auto dc_request = [&](int g, int buf) {
        const int c = z * PFA_G + g;
        const long long base = (((long long)b * a.C + c) * a.H + h) * PFA_N;
        const long long mbase = (long long)b * dc.mask.s[0] + (long long)c * dc.mask.s[1] + (long long)h * dc.mask.s[2];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int wcol = pfa372_shift(min(l + 64 * i, PFA_N - 1), a.halfW);
            pp[buf][i] = dc.pred[base + wcol];
            rr[buf][i] = dc.ref[base + wcol];
        }
        if (mask_u8) {
#pragma unroll
            for (int i = 0; i < 6; ++i)
                mm[buf][i] = ((const unsigned char*)dc.mask.p)[mbase + (long long)pfa372_shift(min(l + 64 * i, PFA_N - 1), a.halfW) * dc.mask.s[3]];
        } else {
#pragma unroll
            for (int i = 0; i < 6; ++i)
                mm[buf][i] = ((const unsigned*)dc.mask.p)[mbase + (long long)pfa372_shift(min(l + 64 * i, PFA_N - 1), a.halfW) * dc.mask.s[3]];
        }
    };
    if constexpr (DC) dc_request(0, 0);
#pragma unroll
    for (int g = 0; g < PFA_G; ++g) {
        if (g >= Cg) break;
        const int c = z * PFA_G + g;
        if constexpr (DC) {
            if (g + 1 < Cg) dc_request(g + 1, (g + 1) & 1);
        }
        pfa_c* orow = reinterpret_cast<pfa_c*>(out_) + l372_kbase(a, (long long)b * a.C + c, (int)h);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int n = l + 64 * i;
            if (n < PFA_N) {
                const int wcol = pfa372_shift(n, a.halfW);
                pfa_c r = X[g * PFA_RS + n];
                if constexpr (DC) {
                    const float2 p_ = pp[g & 1][i], rf = rr[g & 1][i];
                    const bool m = (mm[g & 1][i] & 0x7fffffffu) != 0u;      // u8: any bit; fp32: != +-0 (what `!= 0.f` says, NaN included)
                    const float sx = m ? (p_.x - rf.x) * w : 0.f, sy = m ? (p_.y - rf.y) * w : 0.f;   // vn_block.py:109-110
                    r = pfa_mk(p_.x - sx - r[0], p_.y - sy - r[1]);                                      // vn_block.py:119
                }
                orow[l372_kcol(a, wcol)] = r;
                if (RED) X[g * PFA_RS + n] = r;
            }
        }
    }
    if (RED) {
        for (int g = Cg; g < PFA_G; ++g)
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int n = l + 64 * i;
                if (n < PFA_N) X[g * PFA_RS + n] = pfa_mk(0.f, 0.f);
            }
        __syncthreads();
        // ---- the body of k_pfa372_reduce on the rows just written ------------------------------------------------------------------------
        pfa_c u[3][12];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int d = min(l + 64 * p, PFA_D - 1), g2 = d / PFA_N2, k2 = d - g2 * PFA_N2;
            const int base = (156 * k2) % PFA_N;
#pragma unroll
            for (int k1 = 0; k1 < 12; ++k1) {
                int k = base + (217 * k1) % PFA_N;
                k = k >= PFA_N ? k - PFA_N : k;
                u[p][k1] = X[g2 * PFA_RS + k];
            }
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int d = l + 64 * p;
            if (d < PFA_D) {
                pfa_dft12<true>(u[p]);
                const int g2 = d / PFA_N2, k2 = d - g2 * PFA_N2;
                pfa_c* q = X + g2 * PFA_GS + k2 * PFA_KS;
#pragma unroll
                for (int i = 0; i < 12; ++i) q[i] = u[p][i];
            }
        }
        __syncthreads();
        if (laneA) pfa372_gather_a(L, X, g1, n1);
        __syncthreads();
        if (laneA) pfa372_stage_a_inv(L, X, g1, n1, a.scale_i);
        __syncthreads();
        pfa_c* po = reinterpret_cast<pfa_c*>(part_) + (((long long)z * a.B * a.H) + row) * PFA_N;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int n = l + 64 * i;
            if (n < PFA_N) {
                pfa_c sm = X[n];
#pragma unroll
                for (int g = 1; g < PFA_G; ++g) sm = pfa_add(sm, X[g * PFA_RS + n]);
                po[pfa372_shift(n, a.halfW)] = sm;
            }
        }
    }
}

__global__ __launch_bounds__(64, 2) void k_pfa372_reduce(const float2* __restrict__ k_, const float2* __restrict__ Sp_,
                                                          float2* __restrict__ part_, L372Args a) {
    const pfa_c* __restrict__ kin = reinterpret_cast<const pfa_c*>(k_);
    const pfa_c* __restrict__ Sp = reinterpret_cast<const pfa_c*>(Sp_);
    pfa_c* __restrict__ part = reinterpret_cast<pfa_c*>(part_);
    extern __shared__ __attribute__((aligned(16))) float2 X_[];
    pfa_c* X = reinterpret_cast<pfa_c*>(X_);
    const int l = threadIdx.x;
    const unsigned task = (unsigned)mrx_xcd_band(blockIdx.x, a.ntasks);
    const unsigned row = task / (unsigned)a.T;
    const int z = (int)(task - row * (unsigned)a.T);
    const unsigned b = row / (unsigned)a.H, h = row - b * (unsigned)a.H;
    const int Cg = min(PFA_G, a.C - z * PFA_G);
    const bool laneA = l < PFA_L1;
    const int g1 = l / PFA_N1, n1 = l - g1 * PFA_N1;
    // the task's k-space rows, contiguous per coil -> LDS in transform order.  All 30 row loads and the 31 map loads are requested before the first
    // LDS write, unconditionally (idle lanes and missing coils repeat a valid address): behind `if (n < 372)` every load sat in its own basic block
    // and was waited for there -- thirty memory round trips one after the other at the head of every task.
    Pfa372Lane L;
    {
        pfa_c kv[PFA_G][6];
#pragma unroll
        for (int g = 0; g < PFA_G; ++g) {
            const int c = min(z * PFA_G + g, a.C - 1);
            const pfa_c* krow = kin + l372_kbase(a, (long long)b * a.C + c, (int)h);
#pragma unroll
            for (int i = 0; i < 6; ++i) kv[g][i] = krow[l372_kcol(a, pfa372_shift(min(l + 64 * i, PFA_N - 1), a.halfW))];
        }
        const pfa_c* sp = Sp + (long long)task * L372_TASK_C2 + min(l, PFA_L1 - 1);
#pragma unroll
        for (int n2 = 0; n2 < 31; ++n2) L.s[n2] = sp[n2 * PFA_L1];
#pragma unroll
        for (int g = 0; g < PFA_G; ++g)
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int n = l + 64 * i;
                if (n < PFA_N) X[g * PFA_RS + n] = g < Cg ? kv[g][i] : pfa_mk(0.f, 0.f);
            }
    }
    __syncthreads();
    pfa_c v[3][12];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const int d = min(l + 64 * p, PFA_D - 1), g2 = d / PFA_N2, k2 = d - g2 * PFA_N2;
        const int base = (156 * k2) % PFA_N;
#pragma unroll
        for (int k1 = 0; k1 < 12; ++k1) {
            int k = base + (217 * k1) % PFA_N;
            k = k >= PFA_N ? k - PFA_N : k;
            v[p][k1] = X[g2 * PFA_RS + k];
        }
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const int d = l + 64 * p;
        if (d < PFA_D) {
            pfa_dft12<true>(v[p]);
            const int g2 = d / PFA_N2, k2 = d - g2 * PFA_N2;
            pfa_c* q = X + g2 * PFA_GS + k2 * PFA_KS;
#pragma unroll
            for (int i = 0; i < 12; ++i) q[i] = v[p][i];
        }
    }
    __syncthreads();
    if (laneA) pfa372_gather_a(L, X, g1, n1);
    __syncthreads();
    if (laneA) pfa372_stage_a_inv(L, X, g1, n1, a.scale_i);
    __syncthreads();
    pfa_c* po = part + (((long long)z * a.B * a.H) + row) * PFA_N;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int n = l + 64 * i;
        if (n < PFA_N) {
            pfa_c s = X[n];
#pragma unroll
            for (int g = 1; g < PFA_G; ++g) s = pfa_add(s, X[g * PFA_RS + n]);
            po[pfa372_shift(n, a.halfW)] = s;
        }
    }
}

__global__ void k_l372_zero(float2* __restrict__ p, long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) p[i] = make_float2(0.f, 0.f);
}
// out[i] = sum_k part_k[i]   (complex image [B,H,372])
__global__ void k_pfa372_sum(const float2* __restrict__ part, float2* __restrict__ out, int nparts, long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        float2 s = part[i];
        for (int k = 1; k < nparts; ++k) {
            const float2 v = part[(long long)k * total + i];
            s.x += v.x;
            s.y += v.y;
        }
        out[i] = s;
    }
}

// out4[b, 0:4] = (eta_re, eta_im, post * sum_k part_k re, im)   (rim_utils.py:61-67)
__global__ void k_llg372_combine(const float2* __restrict__ eta, const float2* __restrict__ part, float* __restrict__ out, int nparts,
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

static inline float l372_scale(int inverse, int norm) {   // fft.py:77-81,155-159 (same rule as fft.hip)
    if (norm == MRX_NORM_ORTHO) return (float)(1.0 / sqrt((double)PFA_N));
    if (norm == MRX_NORM_FORWARD) return inverse ? 1.0f : (float)(1.0 / (double)PFA_N);
    return inverse ? (float)(1.0 / (double)PFA_N) : 1.0f;
}
static int l372_args(L372Args* a, int B, int C, int H, int norm, int centered, int mask_batched) {
    MRX_REQUIRE(B >= 0 && C >= 1 && H >= 1, MRX_EINVAL, "mrx_llg372: bad dims B=%d C=%d H=%d", B, C, H);
    MRX_REQUIRE(norm >= 0 && norm <= 3, MRX_EINVAL, "mrx_llg372: bad normalization %d", norm);
    a->B = B, a->C = C, a->H = H, a->T = pfa372_tasks(C);
    a->halfW = centered ? PFA_N / 2 : 0;
    a->mask_bstride = mask_batched ? PFA_N : 0;
    a->ntasks = (long long)B * H * a->T;
    a->scale_f = l372_scale(0, norm);
    a->scale_i = l372_scale(1, norm);
    a->tiled = 0;
    return MRX_OK;
}

extern "C" int mrx_llg372_supported(int W) { return W == PFA_N; }
extern "C" int64_t mrx_llg372_operand_floats(int B, int C, int H) {
    if (B < 0 || C < 1 || H < 1) return -1;
    return (int64_t)B * H * pfa372_tasks(C) * L372_TASK_C2 * 2;
}
extern "C" int64_t mrx_llg372_work_floats(int B, int C, int H) {
    if (B < 0 || C < 1 || H < 1) return -1;
    return (int64_t)pfa372_tasks(C) * B * H * PFA_N * 2;
}
extern "C" int mrx_llg372_prepare(const float* yt, const float* S, const void* mask, int mask_kind, const int64_t* mstride, float* ytp,
                                  float* Sp, float* maskp, int B, int C, int H, int centered, void* stream) {
    MRX_REQUIRE(yt && S && mask && mstride && ytp && Sp && maskp, MRX_EINVAL, "mrx_llg372_prepare: null pointer");
    MRX_REQUIRE(mask_kind == MRX_MASK_U8 || mask_kind == MRX_MASK_F32, MRX_EINVAL, "mrx_llg372_prepare: bad mask kind %d", mask_kind);
    MRX_REQUIRE(mstride[1] == 0 && mstride[2] == 0, MRX_EUNSUP,
                "mrx_llg372_prepare: the mask must depend on the column (and batch) index only (strides %lld %lld)", (long long)mstride[1],
                (long long)mstride[2]);
    L372Args a;
    int rc = l372_args(&a, B, C, H, 0, centered, mstride[0] != 0);
    if (rc) return rc;
    if (B == 0) return MRX_OK;
    hipStream_t st = (hipStream_t)stream;
    const long long total = a.ntasks * L372_TASK_C2;
    long long nb = (total + 255) / 256;
    if (nb > 8192) nb = 8192;
    if (a.ntasks < (1ll << 31) && !MRX_DEBUG_ENV("MRX_LLG372_PREP_ELEMENTWISE"))
        hipLaunchKernelGGL(k_llg372_prep_rows, dim3((unsigned)a.ntasks), dim3(256), 0, st, (const float2*)yt, (const float2*)S, (float2*)ytp, (float2*)Sp, a);
    else
        hipLaunchKernelGGL(k_llg372_prep, dim3((unsigned)nb), dim3(256), 0, st, (const float2*)yt, (const float2*)S, (float2*)ytp, (float2*)Sp, a);
    MrxMask m;
    m.p = mask;
    m.kind = mask_kind;
    for (int i = 0; i < 4; ++i) m.s[i] = mstride[i];
    const int nbm = mstride[0] != 0 ? B : 1;
    hipLaunchKernelGGL(k_llg372_prep_mask, dim3(mrx_cdiv((long long)nbm * PFA_N, 256)), dim3(256), 0, st, m, maskp, nbm, a.halfW);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
extern "C" int mrx_llg372(const float* eta, const float* ytp, const float* Sp, const float* maskp, int mask_batched, float* out4,
                          float* work, int* nparts, int B, int C, int H, float inv_sigma2, int norm, int centered, void* stream) {
    MRX_REQUIRE(eta && Sp && maskp && work && (out4 || nparts), MRX_EINVAL, "mrx_llg372: null pointer");
    const bool noy = ytp == nullptr;      // plane T of `work` holds the constant term (mrx_llg372_const_plane; zeros: the linear part alone = its own adjoint)
    L372Args a;
    int rc = l372_args(&a, B, C, H, norm, centered, mask_batched);
    if (rc) return rc;
    if (nparts) *nparts = 0;
    if (B == 0) return MRX_OK;
    MRX_REQUIRE(a.ntasks < (1ll << 31), MRX_EUNSUP, "mrx_llg372: too many tasks");
    hipStream_t st = (hipStream_t)stream;
    static const int ablate = MRX_DEBUG_ENV("MRX_LLG372_ABLATE") ? atoi(MRX_DEBUG_ENV("MRX_LLG372_ABLATE")) : 0;
    const dim3 grid((unsigned)a.ntasks), blk(64);
    const float2 *pe = (const float2*)eta, *py = (const float2*)ytp, *ps = (const float2*)Sp;
    if (noy)
        hipLaunchKernelGGL((k_llg372<0, true>), grid, blk, L372_LDS_BYTES, st, pe, py, ps, maskp, (float2*)work, a, L372Gather{nullptr, nullptr, nullptr});
    else if (ablate == 1)
        hipLaunchKernelGGL((k_llg372<1>), grid, blk, L372_LDS_BYTES, st, pe, py, ps, maskp, (float2*)work, a, L372Gather{nullptr, nullptr, nullptr});
    else if (ablate == 2)
        hipLaunchKernelGGL((k_llg372<2>), grid, blk, L372_LDS_BYTES, st, pe, py, ps, maskp, (float2*)work, a, L372Gather{nullptr, nullptr, nullptr});
    else if (ablate == 3) {
        static unsigned long long* d_trace = nullptr;
        if (!d_trace) {
            (void)hipMalloc((void**)&d_trace, sizeof(unsigned long long) * 8 * (size_t)a.ntasks);
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_l372_trace), &d_trace, sizeof(d_trace));
        }
        hipLaunchKernelGGL((k_llg372<3>), grid, blk, L372_LDS_BYTES, st, pe, py, ps, maskp, (float2*)work, a, L372Gather{nullptr, nullptr, nullptr});
        if ((MRX_DEBUG_ENV("MRX_TRACE") && atoi(MRX_DEBUG_ENV("MRX_TRACE")) >= 2)) {
            (void)hipStreamSynchronize(st);
            const size_t nt = (size_t)a.ntasks;
            unsigned long long* h = (unsigned long long*)malloc(sizeof(unsigned long long) * 8 * nt);
            (void)hipMemcpy(h, d_trace, sizeof(unsigned long long) * 8 * nt, hipMemcpyDeviceToHost);
            const char* names[6] = {"operands requested -> eta staged in LDS", "expand (waits for S)", "stage A (31-point DFTs)",
                                    "stage B (waits for yt; 12-point DFTs + DC)", "stage A' (inverse 31-point DFTs, conj(S))",
                                    "coil-group sum + store"};
            for (int k = 0; k < 6; ++k) {
                double mn = 1e30, mx = 0, sum = 0;
                for (size_t i = 0; i < nt; ++i) {
                    const double v = (double)(h[i * 8 + k + 1] - h[i * 8 + k]);   // s_memtime is per XCD: differences within a wave only
                    mn = v < mn ? v : mn, mx = v > mx ? v : mx, sum += v;
                }
                fprintf(stderr, "[llg372-trace] %-48s min %7.0f  mean %7.0f  max %7.0f cycles\n", names[k], mn, sum / nt, mx);
            }
            {
                double sum = 0, mx = 0;
                for (size_t i = 0; i < nt; ++i) {
                    const double v = (double)(h[i * 8 + 6] - h[i * 8]);
                    sum += v, mx = v > mx ? v : mx;
                }
                fprintf(stderr, "[llg372-trace] whole wave: mean %.0f  max %.0f cycles over %zu waves\n", sum / nt, mx, nt);
            }
            free(h);
        }
    } else
        hipLaunchKernelGGL((k_llg372<0>), grid, blk, L372_LDS_BYTES, st, pe, py, ps, maskp, (float2*)work, a, L372Gather{nullptr, nullptr, nullptr});
    const int np = a.T + (noy ? 1 : 0);
    if (nparts) {
        *nparts = np;
        MRX_LAUNCH_CHECK();
        return MRX_OK;
    }
    const long long plane = (long long)H * PFA_N, total = plane * B;
    long long nb = (total + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_llg372_combine, dim3((unsigned)nb), dim3(256), 0, st, (const float2*)eta, (const float2*)work, out4, np,
                       (long long)B, plane, inv_sigma2);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
// mrx_llg372 (ytp = NULL form) on eta_out = eta + the nine-tap gather of `taps` [B,18,H,372] (+ b_final): the final convolution's gather
// (mrx_rim_final_gather) folded into the next step's gradient.  eta_out is written (bit-identical to mrx_rim_final_gather's result); `work` as for
// mrx_llg372 with ytp = NULL (its constant plane prepared by mrx_llg372_const_plane).
extern "C" int mrx_llg372_gather(const float* eta, const float* taps, const float* b_final, float* eta_out, const float* Sp, const float* maskp,
                                 int mask_batched, float* out4, float* work, int* nparts, int B, int C, int H, float inv_sigma2, int norm, int centered,
                                 void* stream) {
    MRX_REQUIRE(eta && taps && eta_out && Sp && maskp && work && (out4 || nparts), MRX_EINVAL, "mrx_llg372_gather: null pointer");
    L372Args a;
    int rc = l372_args(&a, B, C, H, norm, centered, mask_batched);
    if (rc) return rc;
    if (nparts) *nparts = 0;
    if (B == 0) return MRX_OK;
    MRX_REQUIRE(a.ntasks < (1ll << 31), MRX_EUNSUP, "mrx_llg372_gather: too many tasks");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL((k_llg372<0, true, true>), dim3((unsigned)a.ntasks), dim3(64), L372_LDS_BYTES, st, (const float2*)eta, (const float2*)nullptr,
                       (const float2*)Sp, maskp, (float2*)work, a, L372Gather{taps, b_final, (float2*)eta_out});
    const int np = a.T + 1;
    if (nparts) {
        *nparts = np;
        MRX_LAUNCH_CHECK();
        return MRX_OK;
    }
    const long long plane = (long long)H * PFA_N, total = plane * B;
    long long nb = (total + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_llg372_combine, dim3((unsigned)nb), dim3(256), 0, st, (const float2*)eta_out, (const float2*)work, out4, np, (long long)B, plane,
                       inv_sigma2);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
// mrx_llg372_gather on the row-pre-summed tap planes of mrx_rim_layer2_f16_cb8_q (taps_q [B][3][H][372][2], edges: mrx_rim_taps_q_edge_floats): eta_out is bit-identical to
// mrx_rim_final_gather_q's result.
extern "C" int mrx_llg372_gather_q(const float* eta, const float* taps_q, const float* edges, const float* b_final, float* eta_out, const float* Sp,
                                   const float* maskp, int mask_batched, float* out4, float* work, int* nparts, int B, int C, int H, float inv_sigma2, int norm,
                                   int centered, void* stream) {
    MRX_REQUIRE(eta && taps_q && edges && eta_out && Sp && maskp && work && (out4 || nparts), MRX_EINVAL, "mrx_llg372_gather_q: null pointer");
    L372Args a;
    int rc = l372_args(&a, B, C, H, norm, centered, mask_batched);
    if (rc) return rc;
    if (nparts) *nparts = 0;
    if (B == 0) return MRX_OK;
    MRX_REQUIRE(a.ntasks < (1ll << 31), MRX_EUNSUP, "mrx_llg372_gather_q: too many tasks");
    hipStream_t st = (hipStream_t)stream;
    L372Gather ga{taps_q, b_final, (float2*)eta_out};
    ga.edges = edges;
    hipLaunchKernelGGL((k_llg372<0, true, true>), dim3((unsigned)a.ntasks), dim3(64), L372_LDS_BYTES, st, (const float2*)eta, (const float2*)nullptr,
                       (const float2*)Sp, maskp, (float2*)work, a, ga);
    const int np = a.T + 1;
    if (nparts) {
        *nparts = np;
        MRX_LAUNCH_CHECK();
        return MRX_OK;
    }
    const long long plane = (long long)H * PFA_N, total = plane * B;
    long long nb = (total + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_llg372_combine, dim3((unsigned)nb), dim3(256), 0, st, (const float2*)eta_out, (const float2*)work, out4, np, (long long)B, plane,
                       inv_sigma2);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
// The constant term of the gradient, -A^H M y = -sum_c conj(S_c) IFFT_W(m yt_c), into plane T (T = ceil(C / 5)) of `work`
// ((T + 1) planes [B,H,372,2] = mrx_llg372_work_floats + B * H * 372 * 2 floats): the full kernel on eta = 0 and the sum of its T partial planes, once
// per slice.  Afterwards mrx_llg372(ytp = NULL) leaves planes 0 .. T - 1 = the linear part and reports T + 1 partial planes.
extern "C" int mrx_llg372_const_plane(const float* ytp, const float* Sp, const float* maskp, int mask_batched, float* work, int B, int C, int H,
                                      int norm, int centered, void* stream) {
    MRX_REQUIRE(ytp && Sp && maskp && work, MRX_EINVAL, "mrx_llg372_const_plane: null pointer");
    L372Args a;
    int rc = l372_args(&a, B, C, H, norm, centered, mask_batched);
    if (rc) return rc;
    if (B == 0) return MRX_OK;
    MRX_REQUIRE(a.ntasks < (1ll << 31), MRX_EUNSUP, "mrx_llg372_const_plane: too many tasks");
    hipStream_t st = (hipStream_t)stream;
    const long long total = (long long)B * H * PFA_N;
    float2* cpl = (float2*)work + (long long)a.T * total;
    // eta = 0, read from the plane that receives the result afterwards.  (A kernel, not hipMemsetAsync: captured into a hipGraph the memset node was not
    // ordered against the kernels around it -- replays of tests/test_gpu_graph.py differed from run to run.)
    long long nz = (total + 255) / 256;
    if (nz > 2048) nz = 2048;
    hipLaunchKernelGGL(k_l372_zero, dim3((unsigned)nz), dim3(256), 0, st, cpl, total);
    hipLaunchKernelGGL((k_llg372<0>), dim3((unsigned)a.ntasks), dim3(64), L372_LDS_BYTES, st, (const float2*)cpl, (const float2*)ytp, (const float2*)Sp,
                       maskp, (float2*)work, a, L372Gather{nullptr, nullptr, nullptr});
    long long nb = (total + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_pfa372_sum, dim3((unsigned)nb), dim3(256), 0, st, (const float2*)work, cpl, a.T, total);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// maps only (E2EVN / general-mask paths): Sp as in mrx_llg372_prepare
__global__ void k_pfa372_prep_maps(const float2* __restrict__ S, float2* __restrict__ Sp, L372Args a) {
    const long long total = a.ntasks * L372_TASK_C2;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long task = i / L372_TASK_C2;
        const int e = (int)(i - task * L372_TASK_C2);
        const long long row = task / a.T;
        const int z = (int)(task - row * a.T);
        const long long b = row / a.H, h = row - b * a.H;
        int g, w;
        const int n2 = e / PFA_L1, lane = e - n2 * PFA_L1;
        pfa372_sp_src(n2, lane, a.halfW, &g, &w);
        const int c = z * PFA_G + g;
        Sp[i] = c < a.C ? S[((b * a.C + c) * a.H + h) * PFA_N + w] : make_float2(0.f, 0.f);
    }
}
// (the rows through LDS, as k_llg372_prep_rows: one workgroup per task)
__global__ __launch_bounds__(256) void k_pfa372_prep_maps_rows(const float2* __restrict__ S, float2* __restrict__ Sp, L372Args a) {
    __shared__ float2 rs[PFA_G][PFA_N];
    const long long task = blockIdx.x;
    const long long row = task / a.T;
    const int z = (int)(task - row * a.T);
    const long long b = row / a.H, h = row - b * a.H;
    for (int i = threadIdx.x; i < PFA_G * PFA_N; i += 256) {
        const int g = i / PFA_N, w = i - g * PFA_N, c = z * PFA_G + g;
        const float2 v = S[((b * a.C + (c < a.C ? c : a.C - 1)) * a.H + h) * PFA_N + w];
        rs[g][w] = c < a.C ? v : make_float2(0.f, 0.f);
    }
    __syncthreads();
    for (int e = threadIdx.x; e < L372_TASK_C2; e += 256) {
        int g, w;
        const int n2 = e / PFA_L1, lane = e - n2 * PFA_L1;
        pfa372_sp_src(n2, lane, a.halfW, &g, &w);
        Sp[task * L372_TASK_C2 + e] = rs[g][w];
    }
}
extern "C" int mrx_pfa372_prepare_maps(const float* S, float* Sp, int B, int C, int H, int centered, void* stream) {
    MRX_REQUIRE(S && Sp, MRX_EINVAL, "mrx_pfa372_prepare_maps: null pointer");
    L372Args a;
    int rc = l372_args(&a, B, C, H, 0, centered, 0);
    if (rc) return rc;
    if (B == 0) return MRX_OK;
    const long long total = a.ntasks * L372_TASK_C2;
    long long nb = (total + 255) / 256;
    if (nb > 8192) nb = 8192;
    if (a.ntasks < (1ll << 31))
        hipLaunchKernelGGL(k_pfa372_prep_maps_rows, dim3((unsigned)a.ntasks), dim3(256), 0, (hipStream_t)stream, (const float2*)S, (float2*)Sp, a);
    else
        hipLaunchKernelGGL(k_pfa372_prep_maps, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, (const float2*)S, (float2*)Sp, a);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
// out [B,C,H,372,2] = FFT_W(x * S) (W transform only), optionally with the soft data-consistency combination as the epilogue:
// out = pred - where(mask, pred - ref, 0) * dc_weight[0] - FFT_W(x * S)   (pred may alias out)
extern "C" int mrx_pfa372_expand(const float* x, const float* Sp, float* out, const float* pred, const float* ref, const void* mask,
                                 int mask_kind, const int64_t* mstride, const float* dc_weight, int B, int C, int H, int norm, int centered,
                                 void* stream) {
    MRX_REQUIRE(x && Sp && out, MRX_EINVAL, "mrx_pfa372_expand: null pointer");
    MRX_REQUIRE(!pred || (ref && mask && mstride && dc_weight), MRX_EINVAL, "mrx_pfa372_expand: the DC epilogue needs pred, ref, mask and dc_weight");
    L372Args a;
    int rc = l372_args(&a, B, C, H, norm, centered, 0);
    if (rc) return rc;
    if (B == 0) return MRX_OK;
    MRX_REQUIRE(a.ntasks < (1ll << 31), MRX_EUNSUP, "mrx_pfa372_expand: too many tasks");
    L372Dc dc;
    dc.on = pred != nullptr;
    dc.pred = (const float2*)pred, dc.ref = (const float2*)ref, dc.w = dc_weight;
    dc.mask.p = mask, dc.mask.kind = mask_kind;
    for (int i = 0; i < 4; ++i) dc.mask.s[i] = (dc.on ? mstride[i] : 0);
    if (dc.on)
        hipLaunchKernelGGL((k_pfa372_expand<false, true>), dim3((unsigned)a.ntasks), dim3(64), sizeof(float2) * PFA_LDS_C2, (hipStream_t)stream, (const float2*)x,
                           (const float2*)Sp, (float2*)out, a, dc, (float2*)nullptr);
    else
        hipLaunchKernelGGL((k_pfa372_expand<false, false>), dim3((unsigned)a.ntasks), dim3(64), sizeof(float2) * PFA_LDS_C2, (hipStream_t)stream, (const float2*)x,
                           (const float2*)Sp, (float2*)out, a, dc, (float2*)nullptr);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
// mrx_pfa372_expand and, in the same pass, red [B,H,372,2] = sum_c conj(S) IFFT_W(out): the sens_reduce the next cascade starts with
// (vn_block.py:71-87 after :109-119).  work: mrx_llg372_work_floats(B,C,H) floats.
extern "C" int mrx_pfa372_expand_reduce(const float* x, const float* Sp, float* out, const float* pred, const float* ref, const void* mask,
                                        int mask_kind, const int64_t* mstride, const float* dc_weight, float* red, float* work, int B, int C,
                                        int H, int norm, int centered, void* stream) {
    MRX_REQUIRE(x && Sp && out && red && work, MRX_EINVAL, "mrx_pfa372_expand_reduce: null pointer");
    MRX_REQUIRE(!pred || (ref && mask && mstride && dc_weight), MRX_EINVAL, "mrx_pfa372_expand_reduce: the DC epilogue needs pred, ref, mask and dc_weight");
    L372Args a;
    int rc = l372_args(&a, B, C, H, norm, centered, 0);
    if (rc) return rc;
    if (B == 0) return MRX_OK;
    MRX_REQUIRE(a.ntasks < (1ll << 31), MRX_EUNSUP, "mrx_pfa372_expand_reduce: too many tasks");
    L372Dc dc;
    dc.on = pred != nullptr;
    dc.pred = (const float2*)pred, dc.ref = (const float2*)ref, dc.w = dc_weight;
    dc.mask.p = mask, dc.mask.kind = mask_kind;
    for (int i = 0; i < 4; ++i) dc.mask.s[i] = (dc.on ? mstride[i] : 0);
    hipStream_t st = (hipStream_t)stream;
    if (dc.on)
        hipLaunchKernelGGL((k_pfa372_expand<true, true>), dim3((unsigned)a.ntasks), dim3(64), sizeof(float2) * PFA_LDS_C2, st, (const float2*)x,
                           (const float2*)Sp, (float2*)out, a, dc, (float2*)work);
    else
        hipLaunchKernelGGL((k_pfa372_expand<true, false>), dim3((unsigned)a.ntasks), dim3(64), sizeof(float2) * PFA_LDS_C2, st, (const float2*)x,
                           (const float2*)Sp, (float2*)out, a, dc, (float2*)work);
    const long long total = (long long)H * PFA_N * B;
    long long nb = (total + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_pfa372_sum, dim3((unsigned)nb), dim3(256), 0, st, (const float2*)work, (float2*)red, a.T, total);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
// sum_c conj(S) * IFFT_W(k) (W transform only).  out4 != NULL: [B,4,H,372] = (eta, post * sum) as log_likelihood_gradient returns it (eta
// required); else out [B,H,372,2] = the sum.  work: mrx_llg372_work_floats(B,C,H) floats.
extern "C" int mrx_pfa372_reduce(const float* k, const float* Sp, const float* eta, float* out, float* out4, float* work, int B, int C, int H,
                                 float post, int norm, int centered, void* stream) {
    MRX_REQUIRE(k && Sp && work && (out || (out4 && eta)), MRX_EINVAL, "mrx_pfa372_reduce: null pointer");
    L372Args a;
    int rc = l372_args(&a, B, C, H, norm, centered, 0);
    if (rc) return rc;
    if (B == 0) return MRX_OK;
    MRX_REQUIRE(a.ntasks < (1ll << 31), MRX_EUNSUP, "mrx_pfa372_reduce: too many tasks");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_pfa372_reduce, dim3((unsigned)a.ntasks), dim3(64), sizeof(float2) * PFA_LDS_C2, st, (const float2*)k, (const float2*)Sp,
                       (float2*)work, a);
    const long long plane = (long long)H * PFA_N, total = plane * B;
    long long nb = (total + 255) / 256;
    if (nb > 2048) nb = 2048;
    if (out4)
        hipLaunchKernelGGL(k_llg372_combine, dim3((unsigned)nb), dim3(256), 0, st, (const float2*)eta, (const float2*)work, out4, a.T,
                           (long long)B, plane, post);
    else
        hipLaunchKernelGGL(k_pfa372_sum, dim3((unsigned)nb), dim3(256), 0, st, (const float2*)work, (float2*)out, a.T, total);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}

// ---- general (row-dependent) masks: log_likelihood_gradient = row pass, column pass + DC (fft.hip: mrx_llg_cols_dc_t4), row pass -----------
// The coil stack between the passes is column-tiled, [B*C][93][H][4] complex (every 4-column tile of an image is one contiguous block of
// H * 32 bytes): the row kernels write / read 32-byte pieces, which neighbouring rows complete to whole lines in L2, and the column
// pass moves contiguous blocks.  Same arithmetic as mrx_pfa372_expand / mrx_pfa372_reduce (only the addresses differ).
extern "C" int mrx_pfa372_expand_t4(const float* x, const float* Sp, float* out_t4, int B, int C, int H, int norm, int centered,
                                    void* stream) {
    MRX_REQUIRE(x && Sp && out_t4, MRX_EINVAL, "mrx_pfa372_expand_t4: null pointer");
    L372Args a;
    int rc = l372_args(&a, B, C, H, norm, centered, 0);
    if (rc) return rc;
    if (B == 0) return MRX_OK;
    MRX_REQUIRE(a.ntasks < (1ll << 31), MRX_EUNSUP, "mrx_pfa372_expand_t4: too many tasks");
    MRX_REQUIRE((long long)H * PFA_N < (1ll << 31), MRX_EUNSUP, "mrx_pfa372_expand_t4: image too tall");
    a.tiled = 1;
    L372Dc dc;
    dc.on = 0;
    dc.pred = dc.ref = nullptr, dc.w = nullptr;
    dc.mask.p = nullptr, dc.mask.kind = MRX_MASK_U8;
    for (int i = 0; i < 4; ++i) dc.mask.s[i] = 0;
    hipLaunchKernelGGL((k_pfa372_expand<false, false>), dim3((unsigned)a.ntasks), dim3(64), sizeof(float2) * PFA_LDS_C2, (hipStream_t)stream, (const float2*)x,
                       (const float2*)Sp, (float2*)out_t4, a, dc, (float2*)nullptr);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
// mrx_pfa372_expand_t4 on eta_out = eta + the nine-tap gather of `taps` [B,18,H,372] (+ b_final): the final convolution's gather (mrx_rim_final_gather)
// folded into the first pass of the NEXT step's general-mask gradient; eta_out is written, bit-identical to mrx_rim_final_gather's result.
extern "C" int mrx_pfa372_expand_t4_gather(const float* eta, const float* taps, const float* b_final, float* eta_out, const float* Sp, float* out_t4,
                                           int B, int C, int H, int norm, int centered, void* stream) {
    MRX_REQUIRE(eta && taps && eta_out && Sp && out_t4, MRX_EINVAL, "mrx_pfa372_expand_t4_gather: null pointer");
    L372Args a;
    int rc = l372_args(&a, B, C, H, norm, centered, 0);
    if (rc) return rc;
    if (B == 0) return MRX_OK;
    MRX_REQUIRE(a.ntasks < (1ll << 31), MRX_EUNSUP, "mrx_pfa372_expand_t4_gather: too many tasks");
    MRX_REQUIRE((long long)H * PFA_N < (1ll << 31), MRX_EUNSUP, "mrx_pfa372_expand_t4_gather: image too tall");
    a.tiled = 1;
    L372Dc dc;
    dc.on = 0;
    dc.pred = dc.ref = nullptr, dc.w = nullptr;
    dc.mask.p = nullptr, dc.mask.kind = MRX_MASK_U8;
    for (int i = 0; i < 4; ++i) dc.mask.s[i] = 0;
    L372Gather ga;
    ga.taps = taps, ga.bias = b_final, ga.eta_out = reinterpret_cast<float2*>(eta_out);
    hipLaunchKernelGGL((k_pfa372_expand<false, false, true>), dim3((unsigned)a.ntasks), dim3(64), sizeof(float2) * PFA_LDS_C2, (hipStream_t)stream, (const float2*)eta,
                       (const float2*)Sp, (float2*)out_t4, a, dc, (float2*)nullptr, ga);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
// post * sum_c conj(S) IFFT_W(k_t4) with eta -> out4 [B,4,H,372] (rim_utils.py:59-67), or, with nparts != NULL, the coil-group partial sums
// left in `work` ([*nparts][B,H,372,2]) for a consumer that adds them itself (mrx_rim_layer_indrnn_packed_llg).
// work: mrx_llg372_work_floats(B,C,H) floats.
extern "C" int mrx_pfa372_reduce_t4(const float* k_t4, const float* Sp, const float* eta, float* out4, float* work, int* nparts, int B, int C,
                                    int H, float post, int norm, int centered, void* stream) {
    MRX_REQUIRE(k_t4 && Sp && work && (nparts || (out4 && eta)), MRX_EINVAL, "mrx_pfa372_reduce_t4: null pointer");
    L372Args a;
    int rc = l372_args(&a, B, C, H, norm, centered, 0);
    if (rc) return rc;
    if (nparts) *nparts = 0;
    if (B == 0) return MRX_OK;
    MRX_REQUIRE(a.ntasks < (1ll << 31), MRX_EUNSUP, "mrx_pfa372_reduce_t4: too many tasks");
    MRX_REQUIRE((long long)H * PFA_N < (1ll << 31), MRX_EUNSUP, "mrx_pfa372_reduce_t4: image too tall");
    a.tiled = 1;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(k_pfa372_reduce, dim3((unsigned)a.ntasks), dim3(64), sizeof(float2) * PFA_LDS_C2, st, (const float2*)k_t4, (const float2*)Sp,
                       (float2*)work, a);
    if (nparts) {
        *nparts = a.T;
        MRX_LAUNCH_CHECK();
        return MRX_OK;
    }
    const long long plane = (long long)H * PFA_N, total = plane * B;
    long long nb = (total + 255) / 256;
    if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(k_llg372_combine, dim3((unsigned)nb), dim3(256), 0, st, (const float2*)eta, (const float2*)work, out4, a.T, (long long)B,
                       plane, post);
    MRX_LAUNCH_CHECK();
    return MRX_OK;
}
