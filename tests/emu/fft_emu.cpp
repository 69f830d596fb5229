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
