// fft_ct.h -- compile-time-plan version of the Stockham stage in fft_core.h.
//
// Same algorithm and data flow (ping-pong between two buffers, twiddles from one exp(-2*pi*i*m/N) table), but the
// length N, the radix R of the stage and Ns (product of the previous radices) are template parameters, so every stride,
// divisor and loop bound is a constant: index math becomes multiply/shift, butterflies are straight-line code, and a
// prime radix that runs first (Ns == 1) is one whole P-point DFT held in registers with constant cos/sin coefficients
// (for 372 = 31*3*4 that is 12 butterflies of 31 points per row).  Radices: 2, 3, 4, 5, 8 and any odd prime.
// Host+device so tests/emu can check it without a GPU.
#pragma once
#include "fft_core.h"

template <int P>
struct MrxPrimeTable {
    float c[P], s[P];  // cos / sin of 2*pi*m/P
};
// constexpr sine / cosine (Taylor series after range reduction), accurate to double rounding for |x| <= pi
constexpr double mrx_csin_core(double x) {
    double term = x, sum = x;
    for (int n = 1; n < 14; ++n) {
        term *= -x * x / ((2.0 * n) * (2.0 * n + 1.0));
        sum += term;
    }
    return sum;
}
constexpr double mrx_ccos_core(double x) {
    double term = 1.0, sum = 1.0;
    for (int n = 1; n < 14; ++n) {
        term *= -x * x / ((2.0 * n - 1.0) * (2.0 * n));
        sum += term;
    }
    return sum;
}
template <int P>
constexpr MrxPrimeTable<P> mrx_make_prime_table() {
    MrxPrimeTable<P> t{};
    const double pi = 3.14159265358979323846264338327950288;
    for (int m = 0; m < P; ++m) {
        double a = 2.0 * pi * (double)m / (double)P;
        if (a > pi) a -= 2.0 * pi;
        t.c[m] = (float)mrx_ccos_core(a);
        t.s[m] = (float)mrx_csin_core(a);
    }
    return t;
}

constexpr bool mrx_ct_small(int r) { return r == 2 || r == 3 || r == 4 || r == 5 || r == 8; }
// a first-stage prime butterfly is split over QS work items, each producing every QS-th output pair
constexpr int mrx_ct_qsplit(int r) { return r >= 16 ? 4 : (r >= 7 ? 2 : 1); }
// work items per sequence of a stage
constexpr int mrx_ct_ips(int n, int r, int ns) {
    return mrx_ct_small(r) ? n / r : (ns == 1 ? (n / r) * mrx_ct_qsplit(r) : (n / r) * ((r + 1) / 2));
}

template <bool INV>
MRX_HD void mrx_dft4(mrx_c32 a0, mrx_c32 a1, mrx_c32 a2, mrx_c32 a3, mrx_c32& y0, mrx_c32& y1, mrx_c32& y2, mrx_c32& y3) {
    const mrx_c32 b0 = mrx_add(a0, a2), b1 = mrx_sub(a0, a2), b2 = mrx_add(a1, a3);
    const mrx_c32 b3 = mrx_rot<INV>(mrx_sub(a1, a3));
    y0 = mrx_add(b0, b2);
    y1 = mrx_add(b1, b3);
    y2 = mrx_sub(b0, b2);
    y3 = mrx_sub(b1, b3);
}

// First-stage prime butterfly, part PART of QS: reads the R inputs of butterfly j (stride N/R), writes output 0 (PART 0)
// and the output pairs (q, R-q) for q = PART+1, PART+1+QS, ...  All cos/sin coefficients are compile-time constants.
template <bool INV, int N, int R, int PART>
MRX_HD void mrx_ct_prime_first(const mrx_c32* in, mrx_c32* out, int j, int es) {
    constexpr MrxPrimeTable<R> T = mrx_make_prime_table<R>();
    constexpr int HALF = (R - 1) / 2;
    constexpr int QS = mrx_ct_qsplit(R);
    constexpr int M = N / R;
    const mrx_c32 x0 = in[j * es];
    mrx_c32 a[HALF], b[HALF];
    mrx_c32 sum = x0;
#pragma unroll
    for (int t = 1; t <= HALF; ++t) {
        const mrx_c32 xa = in[(j + t * M) * es], xb = in[(j + (R - t) * M) * es];
        a[t - 1] = mrx_add(xa, xb);
        b[t - 1] = mrx_sub(xa, xb);
        sum = mrx_add(sum, a[t - 1]);
    }
    if (PART == 0) out[(j * R) * es] = sum;
#pragma unroll
    for (int q = PART + 1; q <= HALF; q += QS) {
        mrx_c32 accR = x0, accI = mrx_mk(0.f, 0.f);
#pragma unroll
        for (int t = 1; t <= HALF; ++t) {
            const int m = (t * q) % R;
            accR.x += a[t - 1].x * T.c[m];
            accR.y += a[t - 1].y * T.c[m];
            accI.x += b[t - 1].x * T.s[m];
            accI.y += b[t - 1].y * T.s[m];
        }
        const mrx_c32 ri = mrx_rot<INV>(accI);
        out[(j * R + q) * es] = mrx_add(accR, ri);
        out[(j * R + (R - q)) * es] = mrx_sub(accR, ri);
    }
}

template <bool INV, int N, int R, int NS>
MRX_HD void mrx_ct_item(const mrx_c32* in, mrx_c32* out, const mrx_c32* tw, int item, int es) {
    constexpr int M = N / R;
    constexpr int TMUL = N / (NS * R);
    if constexpr (R == 2 || R == 3 || R == 4 || R == 5 || R == 8) {
        const int j = item;
        const int k = j % NS;  // constant divisor
        const int ob = (j - k) * R + k;
        const int tstep = k * TMUL;
        mrx_c32 a[R];
#pragma unroll
        for (int t = 0; t < R; ++t) {
            a[t] = in[(j + t * M) * es];
            if (NS > 1 && t > 0) a[t] = mrx_cmul(a[t], mrx_tw<INV>(tw, t * tstep));
        }
        mrx_c32 y[R];
        if constexpr (R == 2) {
            y[0] = mrx_add(a[0], a[1]);
            y[1] = mrx_sub(a[0], a[1]);
        } else if constexpr (R == 4) {
            mrx_dft4<INV>(a[0], a[1], a[2], a[3], y[0], y[1], y[2], y[3]);
        } else if constexpr (R == 3) {
            const float s3 = 0.86602540378443864676f;
            const mrx_c32 t1 = mrx_add(a[1], a[2]);
            const mrx_c32 t2 = mrx_mk(a[0].x - 0.5f * t1.x, a[0].y - 0.5f * t1.y);
            const mrx_c32 d = mrx_sub(a[1], a[2]);
            const mrx_c32 t3 = mrx_rot<INV>(mrx_mk(s3 * d.x, s3 * d.y));
            y[0] = mrx_add(a[0], t1);
            y[1] = mrx_add(t2, t3);
            y[2] = mrx_sub(t2, t3);
        } else if constexpr (R == 5) {
            const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
            const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
            const mrx_c32 p1 = mrx_add(a[1], a[4]), m1 = mrx_sub(a[1], a[4]), p2 = mrx_add(a[2], a[3]), m2 = mrx_sub(a[2], a[3]);
            const mrx_c32 R1 = mrx_mk(a[0].x + c1 * p1.x + c2 * p2.x, a[0].y + c1 * p1.y + c2 * p2.y);
            const mrx_c32 R2 = mrx_mk(a[0].x + c2 * p1.x + c1 * p2.x, a[0].y + c2 * p1.y + c1 * p2.y);
            const mrx_c32 I1 = mrx_rot<INV>(mrx_mk(s1 * m1.x + s2 * m2.x, s1 * m1.y + s2 * m2.y));
            const mrx_c32 I2 = mrx_rot<INV>(mrx_mk(s2 * m1.x - s1 * m2.x, s2 * m1.y - s1 * m2.y));
            y[0] = mrx_add(a[0], mrx_add(p1, p2));
            y[1] = mrx_add(R1, I1);
            y[2] = mrx_add(R2, I2);
            y[3] = mrx_sub(R2, I2);
            y[4] = mrx_sub(R1, I1);
        } else {  // R == 8: two 4-point DFTs (even / odd inputs) combined with w8^q
            mrx_c32 e0, e1, e2, e3, o0, o1, o2, o3;
            mrx_dft4<INV>(a[0], a[2], a[4], a[6], e0, e1, e2, e3);
            mrx_dft4<INV>(a[1], a[3], a[5], a[7], o0, o1, o2, o3);
            const float h = 0.70710678118654752440f;
            // w8^1 = (1 -/+ i)/sqrt2, w8^2 = -/+ i, w8^3 = (-1 -/+ i)/sqrt2   (upper sign: forward)
            const mrx_c32 r1 = mrx_rot<INV>(o1);
            const mrx_c32 t1 = mrx_mk(h * (o1.x + r1.x), h * (o1.y + r1.y));
            const mrx_c32 t2 = mrx_rot<INV>(o2);
            const mrx_c32 r3 = mrx_rot<INV>(o3);
            const mrx_c32 t3 = mrx_mk(h * (r3.x - o3.x), h * (r3.y - o3.y));
            y[0] = mrx_add(e0, o0);
            y[4] = mrx_sub(e0, o0);
            y[1] = mrx_add(e1, t1);
            y[5] = mrx_sub(e1, t1);
            y[2] = mrx_add(e2, t2);
            y[6] = mrx_sub(e2, t2);
            y[3] = mrx_add(e3, t3);
            y[7] = mrx_sub(e3, t3);
        }
#pragma unroll
        for (int q = 0; q < R; ++q) out[(ob + q * NS) * es] = y[q];
    } else if constexpr (NS == 1) {
        // whole R-point DFT (R an odd prime) in registers; item = j * QS + part (the device runner makes `part`
        // wave-uniform and calls mrx_ct_prime_first<PART> directly, so each wave executes one branch only)
        constexpr int QS = mrx_ct_qsplit(R);
        const int j = item / QS;
        const int part = item - j * QS;
        if (part == 0) mrx_ct_prime_first<INV, N, R, 0>(in, out, j, es);
        if constexpr (QS > 1) {
            if (part == 1) mrx_ct_prime_first<INV, N, R, 1>(in, out, j, es);
        }
        if constexpr (QS > 2) {
            if (part == 2) mrx_ct_prime_first<INV, N, R, 2>(in, out, j, es);
            if (part == 3) mrx_ct_prime_first<INV, N, R, 3>(in, out, j, es);
        }
    } else {
        // odd prime radix in a later stage: one work item per (j, q), coefficients from the twiddle table
        constexpr int HALFP = (R + 1) / 2;
        const int j = item / HALFP;
        const int q = item - j * HALFP;
        const int k = j % NS;
        const int ob = (j - k) * R + k;
        const int tstep = k * TMUL;
        const mrx_c32 x0 = in[j * es];
        if (q == 0) {
            mrx_c32 acc = x0;
            for (int t = 1; t < R; ++t) acc = mrx_add(acc, mrx_cmul(in[(j + t * M) * es], mrx_tw<INV>(tw, t * tstep)));
            out[ob * es] = acc;
            return;
        }
        mrx_c32 accR = x0, accI = mrx_mk(0.f, 0.f);
        int m = 0;
        for (int t = 1; t < HALFP; ++t) {
            m += q;
            if (m >= R) m -= R;
            const mrx_c32 xa = mrx_cmul(in[(j + t * M) * es], mrx_tw<INV>(tw, t * tstep));
            const mrx_c32 xb = mrx_cmul(in[(j + (R - t) * M) * es], mrx_tw<INV>(tw, (R - t) * tstep));
            const mrx_c32 w = tw[m * M];
            accR.x += (xa.x + xb.x) * w.x;
            accR.y += (xa.y + xb.y) * w.x;
            accI.x += (xa.x - xb.x) * (-w.y);
            accI.y += (xa.y - xb.y) * (-w.y);
        }
        const mrx_c32 ri = mrx_rot<INV>(accI);
        out[(ob + q * NS) * es] = mrx_add(accR, ri);
        out[(ob + (R - q) * NS) * es] = mrx_sub(accR, ri);
    }
}

// A compile-time plan: N and its radices in execution order.
template <int N_, int... Rs>
struct MrxPlanCT {
    static constexpr int N = N_;
    static constexpr int S = sizeof...(Rs);
};
