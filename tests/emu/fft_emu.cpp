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
