// Host emulation of the LDS Stockham FFT used by the HIP kernels: the SAME fft_core.h stage code,
// driven by plain loops instead of threads.  Test infrastructure only (built by tests/test_fft_core_emu.py
// with g++); it checks the plan/stage index math without a GPU.  Never part of the product library.
#include <cmath>
#include <cstring>
#include <vector>

#include "../../mridc_amd/csrc/fft_core.h"

extern "C" int emu_plan(int n, int* radices) {
    MrxFftPlan p;
    if (mrx_make_plan(n, &p) != 0) return -1;
    for (int i = 0; i < p.nstages; ++i) radices[i] = p.st[i].r;
    return p.nstages;
}

// data: nseq sequences; element (seq, i) at data[(seq*seq_stride + i*es)] (complex units)
extern "C" int emu_fft(float* data, int n, int nseq, int seq_stride, int es, int inverse, long total) {
    MrxFftPlan p;
    if (mrx_make_plan(n, &p) != 0) return -1;
    std::vector<mrx_c32> tw(n);
    for (int m = 0; m < n; ++m) {
        double a = -2.0 * M_PI * (double)m / (double)n;
        tw[m].x = (float)cos(a);
        tw[m].y = (float)sin(a);
    }
    std::vector<mrx_c32> A(total), B(total);
    memcpy(A.data(), data, sizeof(mrx_c32) * total);
    mrx_c32 *a = A.data(), *b = B.data();
    for (int s = 0; s < p.nstages; ++s) {
        const MrxFftStage& S = p.st[s];
        for (int seq = 0; seq < nseq; ++seq)
            for (int it = 0; it < S.ips; ++it) {
                if (inverse)
                    mrx_fft_stage_item<true>(a + seq * seq_stride, b + seq * seq_stride, tw.data(), n, S, it, es);
                else
                    mrx_fft_stage_item<false>(a + seq * seq_stride, b + seq * seq_stride, tw.data(), n, S, it, es);
            }
        mrx_c32* t = a;
        a = b;
        b = t;
    }
    memcpy(data, a, sizeof(mrx_c32) * total);
    return 0;
}

// ---- compile-time plans (fft_ct.h), the ones instantiated by the HIP kernels plus a few awkward ones -------------------
#include "../../mridc_amd/csrc/fft_ct.h"

template <bool INV, int N, int NS, int... Rs>
struct EmuRunCT;
template <bool INV, int N, int NS>
struct EmuRunCT<INV, N, NS> {
    static mrx_c32* run(mrx_c32* a, mrx_c32*, const mrx_c32*, int, int, int) { return a; }
};
template <bool INV, int N, int NS, int R, int... Rest>
struct EmuRunCT<INV, N, NS, R, Rest...> {
    static mrx_c32* run(mrx_c32* a, mrx_c32* b, const mrx_c32* tw, int nseq, int seq_stride, int es) {
        constexpr int ips = mrx_ct_ips(N, R, NS);
        for (int seq = 0; seq < nseq; ++seq)
            for (int it = 0; it < ips; ++it) mrx_ct_item<INV, N, R, NS>(a + seq * seq_stride, b + seq * seq_stride, tw, it, es);
        return EmuRunCT<INV, N, NS * R, Rest...>::run(b, a, tw, nseq, seq_stride, es);
    }
};

template <int N, int... Rs>
static int emu_ct(float* data, int nseq, int seq_stride, int es, int inverse, long total) {
    std::vector<mrx_c32> tw(N);
    for (int m = 0; m < N; ++m) {
        double a = -2.0 * M_PI * (double)m / (double)N;
        tw[m].x = (float)cos(a);
        tw[m].y = (float)sin(a);
    }
    std::vector<mrx_c32> A(total), B(total);
    memcpy(A.data(), data, sizeof(mrx_c32) * total);
    mrx_c32* res = inverse ? EmuRunCT<true, N, 1, Rs...>::run(A.data(), B.data(), tw.data(), nseq, seq_stride, es)
                           : EmuRunCT<false, N, 1, Rs...>::run(A.data(), B.data(), tw.data(), nseq, seq_stride, es);
    memcpy(data, res, sizeof(mrx_c32) * total);
    return 0;
}

extern "C" int emu_fft_ct(float* data, int n, int nseq, int seq_stride, int es, int inverse, long total) {
    switch (n) {
        case 372: return emu_ct<372, 31, 3, 4>(data, nseq, seq_stride, es, inverse, total);
        case 640: return emu_ct<640, 5, 8, 4, 4>(data, nseq, seq_stride, es, inverse, total);
        case 320: return emu_ct<320, 5, 8, 8>(data, nseq, seq_stride, es, inverse, total);
        case 256: return emu_ct<256, 8, 8, 4>(data, nseq, seq_stride, es, inverse, total);
        case 512: return emu_ct<512, 8, 8, 8>(data, nseq, seq_stride, es, inverse, total);
        case 384: return emu_ct<384, 3, 8, 4, 4>(data, nseq, seq_stride, es, inverse, total);
        case 368: return emu_ct<368, 23, 4, 4>(data, nseq, seq_stride, es, inverse, total);
        case 77: return emu_ct<77, 11, 7>(data, nseq, seq_stride, es, inverse, total);
        case 30: return emu_ct<30, 5, 3, 2>(data, nseq, seq_stride, es, inverse, total);
        case 16: return emu_ct<16, 8, 2>(data, nseq, seq_stride, es, inverse, total);
        case 13: return emu_ct<13, 13>(data, nseq, seq_stride, es, inverse, total);
        default: return -1;
    }
}

// ---- the 372-point prime-factor pipeline (pfa372.h): one task = up to 5 coil rows, the 64 lanes emulated by loops ----------------------
#include "../../mridc_amd/csrc/pfa372.h"

// plain 372-point transform through the 12 x 31 maps (checks the index maps and the two small DFTs)
extern "C" void emu_pfa372_fft(const float* in, float* out, int inverse) {
    const mrx_c32* x = reinterpret_cast<const mrx_c32*>(in);
    mrx_c32* y = reinterpret_cast<mrx_c32*>(out);
    std::vector<mrx_c32> Y(12 * 31);
    for (int n1 = 0; n1 < 12; ++n1) {
        mrx_c32 v[31];
        for (int n2 = 0; n2 < 31; ++n2) v[n2] = x[pfa372_n(n1, n2)];
        auto st = [&](int q, mrx_c32 val) { Y[q * 12 + n1] = val; };
        if (inverse) pfa_dft31<true>(v, st); else pfa_dft31<false>(v, st);
    }
    for (int k2 = 0; k2 < 31; ++k2) {
        mrx_c32 v[12];
        for (int i = 0; i < 12; ++i) v[i] = Y[k2 * 12 + i];
        if (inverse) pfa_dft12<true>(v); else pfa_dft12<false>(v);
        for (int k1 = 0; k1 < 12; ++k1) y[pfa372_k(k1, k2)] = v[k1];
    }
}

// one task of the gradient pipeline exactly as the kernel runs it (same phases, same LDS buffer and index maps):
// eta [372], S / yt [Cg][372] (natural column order), mask [372] -> out [372] = sum_c conj(S_c) IFFT(m (s FFT(eta S_c) - yt_c)) * scale_i
extern "C" void emu_pfa372_task(const float* eta_, const float* S_, const float* yt_, const float* mask, float* out_, int Cg, int half,
                                float scale_f, float scale_i) {
    const mrx_c32 *eta = reinterpret_cast<const mrx_c32*>(eta_), *S = reinterpret_cast<const mrx_c32*>(S_),
                  *yt = reinterpret_cast<const mrx_c32*>(yt_);
    mrx_c32* out = reinterpret_cast<mrx_c32*>(out_);
    // once-per-slice operand layouts
    std::vector<mrx_c32> Sp(31 * PFA_L1), ytp(12 * PFA_D);
    std::vector<float> Mk(12 * 31);
    for (int n2 = 0; n2 < 31; ++n2)
        for (int l = 0; l < PFA_L1; ++l) {
            int g, w;
            pfa372_sp_src(n2, l, half, &g, &w);
            Sp[n2 * PFA_L1 + l] = g < Cg ? S[g * PFA_N + w] : mrx_mk(0.f, 0.f);
        }
    for (int k1 = 0; k1 < 12; ++k1)
        for (int d = 0; d < PFA_D; ++d) {
            int g, w;
            pfa372_yt_src(k1, d, half, &g, &w);
            ytp[k1 * PFA_D + d] = g < Cg ? yt[g * PFA_N + w] : mrx_mk(0.f, 0.f);
        }
    for (int k1 = 0; k1 < 12; ++k1)
        for (int k2 = 0; k2 < 31; ++k2) Mk[k1 * 31 + k2] = mask[pfa372_mask_src(k1, k2, half)];
    std::vector<mrx_c32> X(PFA_LDS_C2);
    std::vector<Pfa372Lane> L(64);
    for (int n = 0; n < PFA_ETA_C2; ++n) X[n] = eta[pfa372_shift(n % PFA_N, half)];
    for (int l = 0; l < PFA_L1; ++l) {
        for (int n2 = 0; n2 < 31; ++n2) L[l].s[n2] = Sp[n2 * PFA_L1 + l];
        pfa372_expand(L[l], X.data(), l % 12);
    }
    for (int l = 0; l < PFA_L1; ++l) pfa372_stage_a(L[l], X.data(), l / 12, l % 12);
    for (int pass = 0; pass < 3; ++pass)
        for (int l = 0; l < 64; ++l) {
            const int d = pass * 64 + l;
            if (d >= PFA_D) continue;
            mrx_c32 yv[12];
            for (int k1 = 0; k1 < 12; ++k1) yv[k1] = ytp[k1 * PFA_D + d];
            pfa372_stage_b(X.data(), Mk.data(), yv, d / 31, d % 31, scale_f);
        }
    for (int l = 0; l < PFA_L1; ++l) pfa372_gather_a(L[l], X.data(), l / 12, l % 12);
    for (int l = 0; l < PFA_L1; ++l) pfa372_stage_a_inv(L[l], X.data(), l / 12, l % 12, scale_i);
    for (int n = 0; n < PFA_N; ++n) {
        mrx_c32 s = mrx_mk(0.f, 0.f);
        for (int g = 0; g < PFA_G; ++g) s = mrx_add(s, X[g * PFA_RS + n]);
        out[pfa372_shift(n, half)] = s;
    }
}
