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

// ---- complex arithmetic layer ---------------------------------------------------------------------------------------------------------
// The library is built with MRX_NO_PACKED_FP32 (mridc_amd/_build.py): the scalar layer at the bottom.  The packed layer is kept for reference --
// on MI355X a wave executing packed-fp32 instructions returns wrong results while a wave of another kernel on the same SIMD issues XDL MFMAs
// ("Concurrent streams").
// Packed layer: a complex number is one aligned VGPR pair (ext_vector_type(2)) and every operation below is ONE packed-fp32 instruction
// (v_pk_add / v_pk_mul / v_pk_fma; the swaps and sign flips of "times +-i", complex and conjugate multiplication ride in the
// op_sel / neg modifiers).  Measured on gfx950 (tools/probe/valu_probe.hip): a packed op issues in ~4.7 cycles for two lanes' worth
// of work with any operand kind, a scalar-fp32 VALU op in ~2.6 cycles -- but ~4.3 when one source is an SGPR, which is where dense
// DFT coefficients live; so the constant-coefficient FMAs must be packed.  Host (tests/emu): the same functions in plain C++.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MRX_NO_PACKED_FP32)
typedef float pfa_c __attribute__((ext_vector_type(2)));
#define PFA_FN __device__ __forceinline__
PFA_FN pfa_c pfa_mk(float x, float y) { return (pfa_c){x, y}; }
PFA_FN pfa_c pfa_add(pfa_c a, pfa_c b) { return a + b; }
PFA_FN pfa_c pfa_sub(pfa_c a, pfa_c b) { return a - b; }
PFA_FN pfa_c pfa_scale(pfa_c a, float s) { return a * (pfa_c){s, s}; }
PFA_FN pfa_c pfa_fma_r(pfa_c x, float c, pfa_c acc) { return __builtin_elementwise_fma(x, (pfa_c){c, c}, acc); }   // x * c + acc, c real
// a + (-i b) [INV false] / a + (+i b) [INV true]; pfa_sub_rot: a - (...)
template <bool INV>
PFA_FN pfa_c pfa_add_rot(pfa_c a, pfa_c b) {
    pfa_c d;
    if (INV)
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    else
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
template <bool INV>
PFA_FN pfa_c pfa_sub_rot(pfa_c a, pfa_c b) { return pfa_add_rot<!INV>(a, b); }
PFA_FN pfa_c pfa_cmul(pfa_c e, pfa_c s) {        // e * s
    pfa_c t, x;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(t) : "v"(e), "v"(s));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_lo:[1,0,0]" : "=v"(x) : "v"(e), "v"(s), "v"(t));
    return x;
}
PFA_FN pfa_c pfa_cmul_conj(pfa_c v, pfa_c s) {   // v * conj(s)
    pfa_c t, x;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(t) : "v"(v), "v"(s));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(x) : "v"(v), "v"(s), "v"(t));
    return x;
}
#else
typedef mrx_c32 pfa_c;
#define PFA_FN MRX_HD
PFA_FN pfa_c pfa_mk(float x, float y) { return mrx_mk(x, y); }
PFA_FN pfa_c pfa_add(pfa_c a, pfa_c b) { return mrx_add(a, b); }
PFA_FN pfa_c pfa_sub(pfa_c a, pfa_c b) { return mrx_sub(a, b); }
PFA_FN pfa_c pfa_scale(pfa_c a, float s) { return mrx_mk(a.x * s, a.y * s); }
PFA_FN pfa_c pfa_fma_r(pfa_c x, float c, pfa_c acc) { return mrx_mk(x.x * c + acc.x, x.y * c + acc.y); }
template <bool INV>
PFA_FN pfa_c pfa_add_rot(pfa_c a, pfa_c b) { return mrx_add(a, mrx_rot<INV>(b)); }
template <bool INV>
PFA_FN pfa_c pfa_sub_rot(pfa_c a, pfa_c b) { return mrx_sub(a, mrx_rot<INV>(b)); }
PFA_FN pfa_c pfa_cmul(pfa_c e, pfa_c s) { return mrx_cmul(e, s); }
PFA_FN pfa_c pfa_cmul_conj(pfa_c v, pfa_c s) { return mrx_mk(v.x * s.x + v.y * s.y, v.y * s.x - v.x * s.y); }
#endif

// 31-point DFT of x (destroyed) with every output handed to st(q, value) as soon as it exists (q = 0, then the pairs (q, 31 - q)).
//   X_q = x_0 + sum_{t=1..15} (x_t + x_{31-t}) cos(2 pi t q / 31)  -/+ i  sum_t (x_t - x_{31-t}) sin(2 pi t q / 31)
// 45 packed additions, then per output pair 30 packed FMAs with compile-time coefficients and 2 packed "a +- i b" additions.
template <bool INV, class Store>
PFA_FN void pfa_dft31(pfa_c (&x)[31], Store&& st) {
    constexpr MrxPrimeTable<31> T = mrx_make_prime_table<31>();
    pfa_c sum = x[0];
#pragma unroll
    for (int t = 1; t <= 15; ++t) {
        const pfa_c a = pfa_add(x[t], x[31 - t]), b = pfa_sub(x[t], x[31 - t]);
        x[t] = a;
        x[31 - t] = b;
        sum = pfa_add(sum, a);
    }
    st(0, sum);
#pragma unroll
    for (int q = 1; q <= 15; ++q) {
        pfa_c accR = x[0];
        pfa_c accI = pfa_scale(x[30], T.s[q % 31]);          // t = 1
        accR = pfa_fma_r(x[1], T.c[q % 31], accR);
#pragma unroll
        for (int t = 2; t <= 15; ++t) {
            const int m = (t * q) % 31;
            accR = pfa_fma_r(x[t], T.c[m], accR);
            accI = pfa_fma_r(x[31 - t], T.s[m], accI);
        }
        st(q, pfa_add_rot<INV>(accR, accI));
        st(31 - q, pfa_sub_rot<INV>(accR, accI));
    }
}

// 12-point DFT in place, twiddle-free 3 x 4 prime-factor form: n1 = (4 a + 3 b) mod 12, k1 = (4 ka + 9 kb) mod 12.
template <bool INV>
PFA_FN void pfa_dft12(pfa_c (&v)[12]) {
    const float s3 = 0.86602540378443864676f;
    pfa_c u[3][4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const pfa_c a0 = v[(3 * b) % 12], a1 = v[(4 + 3 * b) % 12], a2 = v[(8 + 3 * b) % 12];
        const pfa_c t1 = pfa_add(a1, a2);
        const pfa_c t2 = pfa_fma_r(t1, -0.5f, a0);
        const pfa_c d = pfa_scale(pfa_sub(a1, a2), s3);
        u[0][b] = pfa_add(a0, t1);
        u[1][b] = pfa_add_rot<INV>(t2, d);
        u[2][b] = pfa_sub_rot<INV>(t2, d);
    }
#pragma unroll
    for (int ka = 0; ka < 3; ++ka) {
        const pfa_c b0 = pfa_add(u[ka][0], u[ka][2]), b1 = pfa_sub(u[ka][0], u[ka][2]);
        const pfa_c b2 = pfa_add(u[ka][1], u[ka][3]), d = pfa_sub(u[ka][1], u[ka][3]);
        v[(4 * ka) % 12] = pfa_add(b0, b2);
        v[(4 * ka + 9) % 12] = pfa_add_rot<INV>(b1, d);
        v[(4 * ka + 18) % 12] = pfa_sub(b0, b2);
        v[(4 * ka + 27) % 12] = pfa_sub_rot<INV>(b1, d);
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
// All LDS indices are in complex units inside the wave-private buffer `X`; Mk is the [12][31] mask table.
struct Pfa372Lane {
    pfa_c s[31];   // this lane's 31 sensitivity values (stage A layout), kept for the whole pipeline
    pfa_c x[31];   // working registers of the 31-point DFTs
};

// phase 1: x = eta * S at the lane's 31 pixels.  The eta row is staged at X[0..702) in transform order AND once more shifted by 372,
// so pixel (31 n1 + 12 n2) mod 372 is read at the unwrapped index: one base address per lane, immediate offsets per element.
#define PFA_ETA_C2 (31 * 11 + 12 * 30 + 1)   // 702
PFA_FN void pfa372_expand(Pfa372Lane& L, const pfa_c* X, int n1) {
    const pfa_c* e = X + 31 * n1;
#pragma unroll
    for (int n2 = 0; n2 < 31; ++n2) L.x[n2] = pfa_cmul(e[12 * n2], L.s[n2]);  // rim_utils.py:47-48
}
// phase 2: forward 31-point DFT -> exchange buffer [g][k2][n1]
PFA_FN void pfa372_stage_a(Pfa372Lane& L, pfa_c* X, int g, int n1) {
    pfa_c* o = X + g * PFA_GS + n1;
    pfa_dft31<false>(L.x, [&](int q, pfa_c v) { o[q * PFA_KS] = v; });
}
// phase 3 (one of the three passes): 12-point DFT, m (s X - yt), inverse 12-point DFT, in place in the exchange buffer
PFA_FN void pfa372_stage_b(pfa_c* X, const float* Mk, const pfa_c (&yv)[12], int g, int k2, float scale_f) {
    pfa_c* p = X + g * PFA_GS + k2 * PFA_KS;
    pfa_c v[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) v[i] = p[i];
    pfa_dft12<false>(v);
#pragma unroll
    for (int k1 = 0; k1 < 12; ++k1)
        v[k1] = pfa_scale(pfa_sub(pfa_scale(v[k1], scale_f), yv[k1]), Mk[k1 * 31 + k2]);  // rim_utils.py:54: m (s X - yt)
    pfa_dft12<true>(v);
#pragma unroll
    for (int i = 0; i < 12; ++i) p[i] = v[i];
}
// phase 4a: fetch the inverse 31-point DFT's inputs
PFA_FN void pfa372_gather_a(Pfa372Lane& L, const pfa_c* X, int g, int n1) {
    const pfa_c* p = X + g * PFA_GS + n1;
#pragma unroll
    for (int k2 = 0; k2 < 31; ++k2) L.x[k2] = p[k2 * PFA_KS];
}
// phase 4b: inverse 31-point DFT, scale, conj(S), into the reduction buffer [g][n]
PFA_FN void pfa372_stage_a_inv(Pfa372Lane& L, pfa_c* X, int g, int n1, float scale_i) {
    pfa_c* o = X + g * PFA_RS;
    const pfa_c* s = L.s;
    pfa_dft31<true>(L.x, [&](int q, pfa_c v) {
        o[pfa372_n(n1, q)] = pfa_cmul_conj(pfa_scale(v, scale_i), s[q]);  // rim_utils.py:61-62
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
