// dft31_mfma_probe.hip -- round 5: is the 31-point DFT of the W = 372 gradient kernel cheaper on the matrix pipe?  ("or the
// radix-31 stage on the matrix pipe".)  Stand-alone: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMRX_NO_PACKED_FP32 -I mridc_amd/csrc tools/probe/dft31_mfma_probe.hip
//
// A wave of k_llg372 runs 60 whole 31-point DFTs, one per lane, in registers: the symmetric dense form, 900 FMAs per lane (pfa372.h: pfa_dft31).
// Form V below is that code.  Form M is the same transform as a matrix product on v_mfma_f32_16x16x4_f32 (exact fp32 FMA chains):
//   a_t = x_t + x_{31-t}, b_t = x_t - x_{31-t};  accR_q = sum_{t<16} cos(2 pi q t / 31) a_t,  accI_q = sum_{t<16} sin(2 pi q t / 31) b_t;
//   X_q = accR_q - i accI_q, X_{31-q} = accR_q + i accI_q.
// Lane l = (kk = l / 16, n = l % 16) holds, for the complex columns c = 16 blk + n (blk < 4), the inputs t in {4 kk .. 4 kk + 3} and their mirrors
// -- the MFMA's k index is (step j, kk) -> t = 4 kk + j -- and receives the outputs q in {4 kk .. 4 kk + 3} and their mirrors: the SAME index set, so
// a pipeline built on it needs no exchange between the DFT's input and output roles.  64 MFMAs per wave and direction (4 blocks x re/im x cos/sin x 4).
// Both forms iterate REP transforms in registers (scaled by 1 / sqrt(31) per round: unitary); outputs of the last round are compared.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "pfa372.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define COLS 60

__global__ __launch_bounds__(512) void k_dft31_valu(const float2* __restrict__ in, float2* __restrict__ out, int reps) {
    const int lane = threadIdx.x & 63, wv = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int c = lane < COLS ? lane : COLS - 1;
    pfa_c x[31], y[31];
    const float2* p = in + ((long long)wv * COLS + c) * 31;
#pragma unroll
    for (int t = 0; t < 31; ++t) x[t] = pfa_mk(p[t].x, p[t].y);
    for (int r = 0; r < reps; ++r) {
        pfa_dft31<false>(x, [&](int q, pfa_c v) { y[q] = v; });
#pragma unroll
        for (int t = 0; t < 31; ++t) x[t] = pfa_scale(y[t], 0.17960530202677491f);      // 1 / sqrt(31): unitary, the values keep their size
    }
    if (lane < COLS) {
        float2* o = out + ((long long)wv * COLS + c) * 31;
#pragma unroll
        for (int t = 0; t < 31; ++t) o[t] = make_float2(x[t].x, x[t].y);
    }
}

__global__ __launch_bounds__(512) void k_dft31_mfma(const float2* __restrict__ in, float2* __restrict__ out, const float* __restrict__ ctab,
                                                    const float* __restrict__ stab, int reps) {
    const int lane = threadIdx.x & 63, wv = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int kk = lane >> 4, n = lane & 15;
    // A operands: lane (m = q = lane % 16, k = lane / 16) of step j holds C[q][t = 4 k + j]
    float ac[4], as[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int t = 4 * kk + j, m = (n * t) % 31;
        ac[j] = ctab[m];
        as[j] = stab[m];
    }
    // this lane's inputs: blocks x (t_j, mirror) -- xa = x_t, xb = x_{31 - t}
    float2 xa[4][4], xb[4][4];
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
        const int c = 16 * blk + n, cc = c < COLS ? c : COLS - 1;
        const float2* p = in + ((long long)wv * COLS + cc) * 31;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t = 4 * kk + j;
            xa[blk][j] = p[t];
            xb[blk][j] = t ? p[31 - t] : make_float2(0.f, 0.f);
        }
    }
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            f32x4 rr = {0.f, 0.f, 0.f, 0.f}, ri = rr, ir = rr, ii = rr;      // accR.re, accR.im, accI.re, accI.im
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool t0 = (4 * kk + j) == 0;
                const float are = t0 ? xa[blk][j].x : xa[blk][j].x + xb[blk][j].x, aim = t0 ? xa[blk][j].y : xa[blk][j].y + xb[blk][j].y;
                const float bre = xa[blk][j].x - xb[blk][j].x, bim = xa[blk][j].y - xb[blk][j].y;     // (t = 0: the sine row is zero)
                rr = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[j], are, rr, 0, 0, 0);
                ri = __builtin_amdgcn_mfma_f32_16x16x4f32(ac[j], aim, ri, 0, 0, 0);
                ir = __builtin_amdgcn_mfma_f32_16x16x4f32(as[j], bre, ir, 0, 0, 0);
                ii = __builtin_amdgcn_mfma_f32_16x16x4f32(as[j], bim, ii, 0, 0, 0);
            }
            // outputs q = 4 kk + r (row 4 kk + r of D) and 31 - q, scaled, back into the input roles
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float s = 0.17960530202677491f;
                xa[blk][q4] = make_float2((rr[q4] + ii[q4]) * s, (ri[q4] - ir[q4]) * s);          // X_q = accR - i accI
                xb[blk][q4] = make_float2((rr[q4] - ii[q4]) * s, (ri[q4] + ir[q4]) * s);          // X_{31 - q}
            }
        }
    }
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
        const int c = 16 * blk + n;
        if (c < COLS) {
            float2* o = out + ((long long)wv * COLS + c) * 31;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int t = 4 * kk + j;
                o[t] = xa[blk][j];
                if (t) o[31 - t] = xb[blk][j];
            }
        }
    }
}

int main() {
    constexpr MrxPrimeTable<31> T = mrx_make_prime_table<31>();
    const int nblk = 256, wpb = 8, nw = nblk * wpb, reps = 400;
    const size_t n = (size_t)nw * COLS * 31;
    std::vector<float2> h(n);
    unsigned s = 12345u;
    for (auto& v : h) {
        s = s * 1664525u + 1013904223u;
        v.x = (float)(s >> 8) / 16777216.f - 0.5f;
        s = s * 1664525u + 1013904223u;
        v.y = (float)(s >> 8) / 16777216.f - 0.5f;
    }
    float2 *din, *do1, *do2;
    float *dc, *ds;
    hipMalloc(&din, n * 8), hipMalloc(&do1, n * 8), hipMalloc(&do2, n * 8), hipMalloc(&dc, 31 * 4), hipMalloc(&ds, 31 * 4);
    hipMemcpy(din, h.data(), n * 8, hipMemcpyHostToDevice);
    hipMemcpy(dc, T.c, 31 * 4, hipMemcpyHostToDevice), hipMemcpy(ds, T.s, 31 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int pass = 0; pass < 3; ++pass) {
        float tv = 0, tm = 0;
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_dft31_valu, dim3(nblk), dim3(64 * wpb), 0, 0, din, do1, reps);
        hipEventRecord(e1), hipEventSynchronize(e1), hipEventElapsedTime(&tv, e0, e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_dft31_mfma, dim3(nblk), dim3(64 * wpb), 0, 0, din, do2, dc, ds, reps);
        hipEventRecord(e1), hipEventSynchronize(e1), hipEventElapsedTime(&tm, e0, e1);
        std::vector<float2> a(n), b(n);
        hipMemcpy(a.data(), do1, n * 8, hipMemcpyDeviceToHost), hipMemcpy(b.data(), do2, n * 8, hipMemcpyDeviceToHost);
        double num = 0, den = 0;
        for (size_t i = 0; i < n; ++i) {
            num += (double)(a[i].x - b[i].x) * (a[i].x - b[i].x) + (double)(a[i].y - b[i].y) * (a[i].y - b[i].y);
            den += (double)a[i].x * a[i].x + (double)a[i].y * a[i].y;
        }
        // 8 waves per CU on 256 CUs, `reps` transforms of 60 columns per wave: time per wave-transform
        printf("pass %d: vector form %.3f ms (%.0f ns per 60-column DFT set and wave), matrix form %.3f ms (%.0f ns) -> %.2f x;  rel-L2 between them %.2e\n", pass,
               tv, 1e6 * tv / reps, tm, 1e6 * tm / reps, tv / tm, sqrt(num / (den > 0 ? den : 1)));
    }
    return 0;
}
