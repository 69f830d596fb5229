// fft_core.h -- mixed-radix Stockham autosort FFT stage, one work item at a time.
//
// The same code runs inside the HIP kernels (data in LDS) and, compiled for the host, inside the CPU
// unit test of the index math (tests/emu/).  A transform of length N = r0*r1*...*r{S-1} is S stages that
// ping-pong between two buffers; stage s with Ns = r0*...*r{s-1} maps, for j in [0, N/r):
//     k = j mod Ns;  x_t = in[j + t*N/r] * w^(t*k),  w = exp(-/+ 2*pi*i / (Ns*r))
//     (y_0..y_{r-1}) = DFT_r(x_0..x_{r-1});  out[(j - k)*r + k + q*Ns] = y_q
// Radices 2,3,4,5 are straight-line butterflies (one work item per j).  Any other prime p uses the
// symmetric odd-radix form with one work item per (j, q), q in [0,(p-1)/2]: item q >= 1 produces y_q and
// y_{p-q} from (x_t + x_{p-t}) cos - / + i (x_t - x_{p-t}) sin; item 0 produces y_0.
// Twiddles come from one table tw[m] = exp(-2*pi*i*m/N), m in [0,N), computed in double on the host.
#pragma once

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MRX_HD __host__ __device__ __forceinline__
typedef float2 mrx_c32;
#else
#define MRX_HD inline
struct mrx_c32 {
    float x, y;
};
#endif

#define MRX_FFT_MAX_STAGES 16

// Per-stage constants are precomputed on the host so the device code has no integer division:
// q = floor(j / d) is computed as (int)((j + 0.5f) * (1.0f / d)), exact for j < 2^22.
struct MrxFftStage {
    int r;        // radix
    int Ns;       // product of the previous radices
    int M;        // n / r  (butterflies per sequence)
    int tmul;     // n / (Ns * r): twiddle index step per unit of k
    int ips;      // work items per sequence
    float inv_Ns, inv_half, inv_ips;
};
struct MrxFftPlan {
    int n;
    int nstages;
    MrxFftStage st[MRX_FFT_MAX_STAGES];
};
MRX_HD int mrx_fdiv(int a, float inv_d) { return (int)(((float)a + 0.5f) * inv_d); }

MRX_HD mrx_c32 mrx_mk(float x, float y) {
    mrx_c32 r;
    r.x = x;
    r.y = y;
    return r;
}
MRX_HD mrx_c32 mrx_add(mrx_c32 a, mrx_c32 b) { return mrx_mk(a.x + b.x, a.y + b.y); }
MRX_HD mrx_c32 mrx_sub(mrx_c32 a, mrx_c32 b) { return mrx_mk(a.x - b.x, a.y - b.y); }
MRX_HD mrx_c32 mrx_cmul(mrx_c32 a, mrx_c32 b) { return mrx_mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
// multiply by -i (forward) or +i (inverse)
template <bool INV>
MRX_HD mrx_c32 mrx_rot(mrx_c32 a) {
    return INV ? mrx_mk(-a.y, a.x) : mrx_mk(a.y, -a.x);
}
template <bool INV>
MRX_HD mrx_c32 mrx_tw(const mrx_c32* tw, int idx) {
    mrx_c32 w = tw[idx];
    if (INV) w.y = -w.y;
    return w;
}

MRX_HD bool mrx_is_small_radix(int r) { return r == 2 || r == 3 || r == 4 || r == 5; }
// work items per sequence in a stage of radix r
MRX_HD int mrx_stage_items(int n, int r) { return mrx_is_small_radix(r) ? n / r : (n / r) * ((r + 1) / 2); }

// One work item of one stage.  `es` = element stride (in complex elements) inside `in`/`out`.
template <bool INV>
MRX_HD void mrx_fft_stage_item(const mrx_c32* in, mrx_c32* out, const mrx_c32* tw, int N, const MrxFftStage& S, int item,
                               int es) {
    const int M = S.M, Ns = S.Ns, r = S.r;
    if (mrx_is_small_radix(r)) {
        const int j = item;
        const int k = j - Ns * mrx_fdiv(j, S.inv_Ns);
        const int ob = (j - k) * r + k;
        const int tstep = k * S.tmul;  // twiddle index step: w^(t*k) = tw[t*tstep]
        mrx_c32 a0 = in[(j)*es];
        if (r == 2) {
            mrx_c32 a1 = in[(j + M) * es];
            if (Ns > 1) a1 = mrx_cmul(a1, mrx_tw<INV>(tw, tstep));
            out[(ob)*es] = mrx_add(a0, a1);
            out[(ob + Ns) * es] = mrx_sub(a0, a1);
        } else if (r == 4) {
            mrx_c32 a1 = in[(j + M) * es], a2 = in[(j + 2 * M) * es], a3 = in[(j + 3 * M) * es];
            if (Ns > 1) {
                a1 = mrx_cmul(a1, mrx_tw<INV>(tw, tstep));
                a2 = mrx_cmul(a2, mrx_tw<INV>(tw, 2 * tstep));
                a3 = mrx_cmul(a3, mrx_tw<INV>(tw, 3 * tstep));
            }
            mrx_c32 b0 = mrx_add(a0, a2), b1 = mrx_sub(a0, a2), b2 = mrx_add(a1, a3);
            mrx_c32 b3 = mrx_rot<INV>(mrx_sub(a1, a3));
            out[(ob)*es] = mrx_add(b0, b2);
            out[(ob + Ns) * es] = mrx_add(b1, b3);
            out[(ob + 2 * Ns) * es] = mrx_sub(b0, b2);
            out[(ob + 3 * Ns) * es] = mrx_sub(b1, b3);
        } else if (r == 3) {
            mrx_c32 a1 = in[(j + M) * es], a2 = in[(j + 2 * M) * es];
            if (Ns > 1) {
                a1 = mrx_cmul(a1, mrx_tw<INV>(tw, tstep));
                a2 = mrx_cmul(a2, mrx_tw<INV>(tw, 2 * tstep));
            }
            const float s3 = 0.86602540378443864676f;
            mrx_c32 t1 = mrx_add(a1, a2);
            mrx_c32 t2 = mrx_mk(a0.x - 0.5f * t1.x, a0.y - 0.5f * t1.y);
            mrx_c32 d = mrx_sub(a1, a2);
            mrx_c32 t3 = mrx_rot<INV>(mrx_mk(s3 * d.x, s3 * d.y));
            out[(ob)*es] = mrx_add(a0, t1);
            out[(ob + Ns) * es] = mrx_add(t2, t3);
            out[(ob + 2 * Ns) * es] = mrx_sub(t2, t3);
        } else {  // r == 5
            mrx_c32 a1 = in[(j + M) * es], a2 = in[(j + 2 * M) * es], a3 = in[(j + 3 * M) * es],
                    a4 = in[(j + 4 * M) * es];
            if (Ns > 1) {
                a1 = mrx_cmul(a1, mrx_tw<INV>(tw, tstep));
                a2 = mrx_cmul(a2, mrx_tw<INV>(tw, 2 * tstep));
                a3 = mrx_cmul(a3, mrx_tw<INV>(tw, 3 * tstep));
                a4 = mrx_cmul(a4, mrx_tw<INV>(tw, 4 * tstep));
            }
            const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
            const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
            mrx_c32 p1 = mrx_add(a1, a4), m1 = mrx_sub(a1, a4), p2 = mrx_add(a2, a3), m2 = mrx_sub(a2, a3);
            mrx_c32 R1 = mrx_mk(a0.x + c1 * p1.x + c2 * p2.x, a0.y + c1 * p1.y + c2 * p2.y);
            mrx_c32 R2 = mrx_mk(a0.x + c2 * p1.x + c1 * p2.x, a0.y + c2 * p1.y + c1 * p2.y);
            mrx_c32 I1 = mrx_rot<INV>(mrx_mk(s1 * m1.x + s2 * m2.x, s1 * m1.y + s2 * m2.y));
            mrx_c32 I2 = mrx_rot<INV>(mrx_mk(s2 * m1.x - s1 * m2.x, s2 * m1.y - s1 * m2.y));
            out[(ob)*es] = mrx_add(a0, mrx_add(p1, p2));
            out[(ob + Ns) * es] = mrx_add(R1, I1);
            out[(ob + 2 * Ns) * es] = mrx_add(R2, I2);
            out[(ob + 3 * Ns) * es] = mrx_sub(R2, I2);
            out[(ob + 4 * Ns) * es] = mrx_sub(R1, I1);
        }
        return;
    }
    // generic odd prime radix p = r
    const int half = (r + 1) / 2;
    const int j = mrx_fdiv(item, S.inv_half);
    const int q = item - j * half;
    const int k = j - Ns * mrx_fdiv(j, S.inv_Ns);
    const int ob = (j - k) * r + k;
    const int tstep = k * S.tmul;
    const int rstep = M;  // n / r: tw[m*rstep] = exp(-2 pi i m / r)
    mrx_c32 x0 = in[j * es];
    if (q == 0) {
        mrx_c32 acc = x0;
        for (int t = 1; t < r; ++t) {
            mrx_c32 xt = in[(j + t * M) * es];
            if (Ns > 1) xt = mrx_cmul(xt, mrx_tw<INV>(tw, t * tstep));
            acc = mrx_add(acc, xt);
        }
        out[ob * es] = acc;
        return;
    }
    mrx_c32 accR = x0, accI = mrx_mk(0.f, 0.f);
    int m = 0;  // (t*q) mod r
    for (int t = 1; t < half; ++t) {
        m += q;
        if (m >= r) m -= r;
        mrx_c32 xa = in[(j + t * M) * es];
        mrx_c32 xb = in[(j + (r - t) * M) * es];
        if (Ns > 1) {
            xa = mrx_cmul(xa, mrx_tw<INV>(tw, t * tstep));
            xb = mrx_cmul(xb, mrx_tw<INV>(tw, (r - t) * tstep));
        }
        const mrx_c32 w = tw[m * rstep];  // (cos, -sin)
        const float c = w.x, s = -w.y;
        accR.x += (xa.x + xb.x) * c;
        accR.y += (xa.y + xb.y) * c;
        accI.x += (xa.x - xb.x) * s;
        accI.y += (xa.y - xb.y) * s;
    }
    // forward: y_q = accR - i*accI, y_{p-q} = accR + i*accI ; inverse: the opposite
    mrx_c32 ri = mrx_rot<INV>(accI);
    out[(ob + q * Ns) * es] = mrx_add(accR, ri);
    out[(ob + (r - q) * Ns) * es] = mrx_sub(accR, ri);
}

// Factorisation used for every plan: generic primes first (they then run with Ns == 1 and need no input
// twiddles), then 5s, 3s, 4s and at most one 2.  Returns 0 on success.
inline int mrx_make_plan(int n, MrxFftPlan* p) {
    if (n < 1) return -1;
    p->n = n;
    p->nstages = 0;
    int rem = n;
    int small[MRX_FFT_MAX_STAGES * 4];
    int ns = 0;
    int c2 = 0;
    while (rem % 2 == 0) {
        rem /= 2;
        ++c2;
    }
    int c3 = 0, c5 = 0;
    while (rem % 3 == 0) {
        rem /= 3;
        ++c3;
    }
    while (rem % 5 == 0) {
        rem /= 5;
        ++c5;
    }
    // remaining generic primes, largest last found -> emit in descending order
    int gp[32];
    int ng = 0;
    for (int f = 7; (long long)f * f <= rem; f += 2) {
        while (rem % f == 0) {
            if (ng >= 32) return -1;
            gp[ng++] = f;
            rem /= f;
        }
    }
    if (rem > 1) {
        if (ng >= 32) return -1;
        gp[ng++] = rem;
    }
    for (int i = ng - 1; i >= 0; --i) small[ns++] = gp[i];
    for (int i = 0; i < c5; ++i) small[ns++] = 5;
    for (int i = 0; i < c3; ++i) small[ns++] = 3;
    for (int i = 0; i < c2 / 2; ++i) small[ns++] = 4;
    if (c2 % 2) small[ns++] = 2;
    if (ns > MRX_FFT_MAX_STAGES) return -1;
    int Ns = 1;
    for (int i = 0; i < ns; ++i) {
        MrxFftStage& S = p->st[i];
        S.r = small[i];
        S.Ns = Ns;
        S.M = n / S.r;
        S.tmul = n / (Ns * S.r);
        S.ips = mrx_stage_items(n, S.r);
        S.inv_Ns = 1.0f / (float)Ns;
        S.inv_half = 1.0f / (float)((S.r + 1) / 2);
        S.inv_ips = 1.0f / (float)S.ips;
        Ns *= S.r;
    }
    p->nstages = ns;
    return 0;
}
