// pfa372.h -- the 372-point row transform of the fastMRI knee width as a Good-Thomas (prime-factor) 12 x 31 transform, and the
// whole log-likelihood-gradient row pipeline built on it (rim_utils.py:11-67 restricted to row-invariant masks, see fft.hip).
//
//   372 = 12 * 31, gcd(12, 31) = 1:   n = (31 n1 + 12 n2) mod 372,   k = (217 k1 + 156 k2) mod 372   (217 = 31 * (31^-1 mod 12),
//   156 = 12 * (12^-1 mod 31))  =>  W372^(n k) = W12^(n1 k1) * W31^(n2 k2): a 2-D 12 x 31 DFT with NO twiddle factors between the
//   stages.  The 12-point DFT is itself twiddle-free (3 x 4); the 31-point DFT is the symmetric dense form (15 cosine + 15 sine
//   coefficients, compile-time constants, 900 FMAs) on inputs held in registers.
//
// One wavefront owns G = 5 coil rows ("a task"): 60 lanes = (coil, n1) each run ONE whole 31-point DFT in registers (stage A);
// the 5 * 31 = 155 (coil, k2) 12-point DFTs run 64 at a time (stage B), where the data-consistency step m (s X - yt) and the inverse
// 12-point DFT happen in the same registers; stage A' is the inverse 31-point DFT, followed by conj(S) and the sum over the wave's
// coils.  The two exchanges go through a wave-private LDS buffer; there is no workgroup barrier and no twiddle table.
//
// The loop-invariant operands are laid out ONCE per slice in the order the lanes consume them (pfa372_prep_*), so every global
// access of the hot kernel is a contiguous 64-lane row: S as [task][n2][60 lanes], yt = IFFT_H(y) as [task][k1][155], the mask as
// [k1][k2].  Same code on the host (tests/emu) with loops over the lanes instead of a wavefront.
#pragma once
#include "fft_ct.h"

#define PFA_N 372
#define PFA_N1 12
#define PFA_N2 31
#define PFA_G 5                 // coils per wavefront
#define PFA_L1 (PFA_G * PFA_N1)  // 60 lanes carry a 31-point DFT each
#define PFA_D (PFA_G * PFA_N2)   // 155 12-point DFTs per task
#define PFA_KS 13               // k2 stride (float2) of the exchange buffer: 26 dwords, conflict-free 12-element reads
#define PFA_GS 428              // coil stride (float2) of the exchange buffer (= 12 mod 32: the next coil continues the bank walk)
#define PFA_RS 388              // coil stride (float2) of the reduction buffer
#define PFA_LDS_C2 (PFA_G * PFA_GS)  // float2 elements of the wave-private buffer (exchange / eta staging / reduction alias)

MRX_HD int pfa372_n(int n1, int n2) {
    const int n = 31 * n1 + 12 * n2;  // < 702
    return n >= PFA_N ? n - PFA_N : n;
}
MRX_HD int pfa372_k(int k1, int k2) { return (217 * k1 + 156 * k2) % PFA_N; }

// 31-point DFT of x (destroyed) with every output handed to st(q, value) as soon as it exists (q = 0, then the pairs (q, 31 - q)).
template <bool INV, class Store>
MRX_HD void pfa_dft31(mrx_c32 (&x)[31], Store&& st) {
    constexpr MrxPrimeTable<31> T = mrx_make_prime_table<31>();
    mrx_c32 sum = x[0];
#pragma unroll
    for (int t = 1; t <= 15; ++t) {
        const mrx_c32 a = mrx_add(x[t], x[31 - t]), b = mrx_sub(x[t], x[31 - t]);
        x[t] = a;
        x[31 - t] = b;
        sum = mrx_add(sum, a);
    }
    st(0, sum);
#pragma unroll
    for (int q = 1; q <= 15; ++q) {
        mrx_c32 accR = x[0], accI = mrx_mk(0.f, 0.f);
#pragma unroll
        for (int t = 1; t <= 15; ++t) {
            const int m = (t * q) % 31;
            accR.x += x[t].x * T.c[m];
            accR.y += x[t].y * T.c[m];
            accI.x += x[31 - t].x * T.s[m];
            accI.y += x[31 - t].y * T.s[m];
        }
        const mrx_c32 ri = mrx_rot<INV>(accI);
        st(q, mrx_add(accR, ri));
        st(31 - q, mrx_sub(accR, ri));
    }
}

// 12-point DFT in place, twiddle-free 3 x 4 prime-factor form: n1 = (4 a + 3 b) mod 12, k1 = (4 ka + 9 kb) mod 12.
template <bool INV>
MRX_HD void pfa_dft12(mrx_c32 (&v)[12]) {
    const float s3 = 0.86602540378443864676f;
    mrx_c32 u[3][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const mrx_c32 a0 = v[(3 * b) % 12], a1 = v[(4 + 3 * b) % 12], a2 = v[(8 + 3 * b) % 12];
        const mrx_c32 t1 = mrx_add(a1, a2);
        const mrx_c32 t2 = mrx_mk(a0.x - 0.5f * t1.x, a0.y - 0.5f * t1.y);
        const mrx_c32 d = mrx_sub(a1, a2);
        const mrx_c32 t3 = mrx_rot<INV>(mrx_mk(s3 * d.x, s3 * d.y));
        u[0][b] = mrx_add(a0, t1);
        u[1][b] = mrx_add(t2, t3);
        u[2][b] = mrx_sub(t2, t3);
    }
#pragma unroll
    for (int ka = 0; ka < 3; ++ka) {
        mrx_c32 y0, y1, y2, y3;
        mrx_dft4<INV>(u[ka][0], u[ka][1], u[ka][2], u[ka][3], y0, y1, y2, y3);
        v[(4 * ka) % 12] = y0;
        v[(4 * ka + 9) % 12] = y1;
        v[(4 * ka + 18) % 12] = y2;
        v[(4 * ka + 27) % 12] = y3;
    }
}

// ---- operand layouts of the hot kernel (built once per slice) -----------------------------------------------------------------
// tasks per image row
MRX_HD int pfa372_tasks(int C) { return (C + PFA_G - 1) / PFA_G; }
// Sp [rows * T][31][60]: element (n2, lane) of a task = S[coil 5 z + lane / 12][w = shift(n(lane % 12, n2))]
// ytp [rows * T][12][155]: element (k1, d) = yt[coil 5 z + d / 31][w = shift(k(k1, d % 31))]
// maskp [12][31]: element (k1, k2) = mask[w = shift(k(k1, k2))]
MRX_HD int pfa372_shift(int p, int half) {
    const int g = p + half;
    return g >= PFA_N ? g - PFA_N : g;
}

// ---- the per-lane phases of the gradient pipeline -----------------------------------------------------------------------------------
// All LDS indices are in float2 units inside the wave-private buffer `X`; Mk is the [12][31] mask table.
struct Pfa372Lane {
    mrx_c32 s[31];   // this lane's 31 sensitivity values (stage A layout), kept for the whole pipeline
    mrx_c32 x[31];   // working registers of the 31-point DFTs
};

// phase 1: x = eta * S at the lane's 31 pixels (eta row staged at X[0..372) in transform order)
MRX_HD void pfa372_expand(Pfa372Lane& L, const mrx_c32* X, int n1) {
#pragma unroll
    for (int n2 = 0; n2 < 31; ++n2) L.x[n2] = mrx_cmul(X[pfa372_n(n1, n2)], L.s[n2]);  // rim_utils.py:47-48
}
// phase 2: forward 31-point DFT -> exchange buffer [g][k2][n1]
MRX_HD void pfa372_stage_a(Pfa372Lane& L, mrx_c32* X, int g, int n1) {
    mrx_c32* o = X + g * PFA_GS + n1;
    pfa_dft31<false>(L.x, [&](int q, mrx_c32 v) { o[q * PFA_KS] = v; });
}
// phase 3 (one of the three passes): 12-point DFT, m (s X - yt), inverse 12-point DFT, in place in the exchange buffer
MRX_HD void pfa372_stage_b(mrx_c32* X, const float* Mk, const mrx_c32 (&yv)[12], int g, int k2, float scale_f) {
    mrx_c32* p = X + g * PFA_GS + k2 * PFA_KS;
    mrx_c32 v[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) v[i] = p[i];
    pfa_dft12<false>(v);
#pragma unroll
    for (int k1 = 0; k1 < 12; ++k1) {
        const float m = Mk[k1 * 31 + k2];
        v[k1] = mrx_mk(m * (v[k1].x * scale_f - yv[k1].x), m * (v[k1].y * scale_f - yv[k1].y));  // rim_utils.py:54
    }
    pfa_dft12<true>(v);
#pragma unroll
    for (int i = 0; i < 12; ++i) p[i] = v[i];
}
// phase 4a: fetch the inverse 31-point DFT's inputs
MRX_HD void pfa372_gather_a(Pfa372Lane& L, const mrx_c32* X, int g, int n1) {
    const mrx_c32* p = X + g * PFA_GS + n1;
#pragma unroll
    for (int k2 = 0; k2 < 31; ++k2) L.x[k2] = p[k2 * PFA_KS];
}
// phase 4b: inverse 31-point DFT, conj(S), into the reduction buffer [g][n]
MRX_HD void pfa372_stage_a_inv(Pfa372Lane& L, mrx_c32* X, int g, int n1, float scale_i) {
    mrx_c32* o = X + g * PFA_RS;
    const mrx_c32* s = L.s;
    pfa_dft31<true>(L.x, [&](int q, mrx_c32 v) {
        v.x *= scale_i;
        v.y *= scale_i;
        o[pfa372_n(n1, q)] = mrx_mk(v.x * s[q].x + v.y * s[q].y, v.y * s[q].x - v.x * s[q].y);  // rim_utils.py:61-62
    });
}

// ---- source positions of the permuted operands (shared by the prep kernel and the host emulation) -----------------------------------
// Sp element (n2, lane < 60): coil-in-task lane / 12, image column shift(n(lane % 12, n2))
MRX_HD void pfa372_sp_src(int n2, int lane, int half, int* g, int* w) {
    *g = lane / PFA_N1;
    *w = pfa372_shift(pfa372_n(lane - *g * PFA_N1, n2), half);
}
// ytp element (k1, d < 155): coil-in-task d / 31, k-space column shift(k(k1, d % 31))
MRX_HD void pfa372_yt_src(int k1, int d, int half, int* g, int* w) {
    *g = d / PFA_N2;
    *w = pfa372_shift(pfa372_k(k1, d - *g * PFA_N2), half);
}
MRX_HD int pfa372_mask_src(int k1, int k2, int half) { return pfa372_shift(pfa372_k(k1, k2), half); }
