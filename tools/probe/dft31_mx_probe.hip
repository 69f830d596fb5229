// dft31_mx_probe.hip -- round 6: the 31-point DFT of the W = 372 gradient kernel on v_mfma_f32_32x32x16_f16 with two-term fp16 operands
// (pfa_dft31_mx below: a drop-in, wave-wide form of pfa372.h's pfa_dft31) against the fp32 vector-ALU form and against float64: error and time.
// MEASURED (profiles/r06_dft31_matrix_pipe_probe.txt): correct (8.3e-8 against float64 where the fp32 form has 9.8e-8) and 1.17 x SLOWER in the kernel's own
// occupancy (two waves per SIMD): 660 instructions instead of 1 240, but the operand split is four-instruction dependent chains, a v_permlane32_swap costs 3.5
// fused multiply-adds and the 24 MFMAs do not overlap anything when both waves of a SIMD are in the same phase.  NOT in the product.  hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMRX_NO_PACKED_FP32 -I mridc_amd/csrc
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "pfa372.h"
#define COLS 60

// ---- the 31-point DFT on the matrix pipe (round 6) ---------------------------------------------------------------------------------------------------
// A WAVE-wide form of pfa_dft31: every lane hands in the 31 inputs of its own column and receives the 31 outputs of its own column, the dense part runs on
// v_mfma_f32_32x32x16_f16 with two-term fp16 operands (x = (h1 + h2) 2^-e, e per column; three term products; fp32 accumulation -- the arithmetic of the RIM
// layer kernels, error O(2^-22) of the column's largest input against pfa_dft31's O(2^-24) fp32 chains).
//   a_t = x_t + x_{31-t}, b_t = x_t - x_{31-t} (t = 1..15), a_0 = x_0, b_0 = 0:   X_q = A_q - i B_q, X_{31-q} = A_q + i B_q,   A_q = sum_t a_t cos(2 pi t q / 31), B_q = sum_t b_t sin(..)
//   => with the ONE real 32 x 32 matrix  M = [[C, S], [C, -S]]  (C, S: 16 x 16, rows q, columns t) and the two real operands  B1 = [a_re; b_im],  B2 = [a_im; b_re]:
//        M B1 = [X_q.re; X_{31-q}.re],   M B2 = [X_{31-q}.im; X_q.im]      (the inverse transform exchanges the roles of q and 31 - q: same matrix)
//   8 MFMA products x 3 term pairs = 24 MFMAs per wave (64 columns) instead of 900 fused multiply-adds per lane; the matrix is 16 registers per lane, made once per kernel.
// Operand layout: lane (n = lane % 32, kg = lane / 32) of a B fragment holds 8 consecutive k of column n of its 32-column block.  A lane splits ITS column into the
// kg = 0 and kg = 1 fragments of every K-step; one v_permlane32_swap per register then turns the pair (kg 0 fragment, kg 1 fragment) into the operands of column
// block 0 and column block 1.  The same swap on the accumulators brings every column's 32 output rows back to its own lane.
typedef unsigned pfa_u4 __attribute__((ext_vector_type(4)));
typedef _Float16 pfa_h8 __attribute__((ext_vector_type(8)));
typedef _Float16 pfa_h2 __attribute__((ext_vector_type(2)));
typedef float pfa_f16r __attribute__((ext_vector_type(16)));
struct PfaDft31Tab {
    pfa_u4 a[2][2];   // [K-step][term]: this lane's A fragments of M 2^14
};
__device__ __forceinline__ void pfa_split2h(float a, float b, unsigned& p1, unsigned& p2) {   // two fp16 terms of a pair already scaled into the fp16 range
    const pfa_h2 h = {(_Float16)a, (_Float16)b};
    const float ra = a - (float)h.x, rb = b - (float)h.y;     // exact
    const pfa_h2 l = {(_Float16)ra, (_Float16)rb};
    p1 = __builtin_bit_cast(unsigned, h);
    p2 = __builtin_bit_cast(unsigned, l);
}
// the same two terms of (a s, b s) for a power-of-two scale s in FOUR instructions: the fp32 multiply, the conversion and the subtraction of each term are one
// v_fma_mix{lo,hi}_f16 (a s and a s - h are exact in fp32, so the single rounding of the fused form is the rounding of pfa_split2h: bit-identical; rim_layer2_sb.hip)
__device__ __forceinline__ void pfa_split2h_scaled(float a, float b, float s, unsigned& p1, unsigned& p2) {
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=&v"(p1) : "v"(a), "v"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(p1) : "v"(b), "v"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=&v"(p2) : "v"(a), "v"(s), "v"(p1));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(p2) : "v"(b), "v"(s), "v"(p1));
}
__device__ __forceinline__ PfaDft31Tab pfa_dft31_tab(int lane) {
    constexpr MrxPrimeTable<31> T = mrx_make_prime_table<31>();
    const int m = lane & 31, kg = lane >> 5, q = m & 15;
    const float sg = m < 16 ? 16384.f : -16384.f;
    PfaDft31Tab tab;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        unsigned hi[4], lo[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int t0 = 8 * kg + 2 * j, i0 = (t0 * q) % 31, i1 = ((t0 + 1) * q) % 31;
            const float v0 = ks == 0 ? T.c[i0] * 16384.f : T.s[i0] * sg, v1 = ks == 0 ? T.c[i1] * 16384.f : T.s[i1] * sg;
            pfa_split2h(v0, v1, hi[j], lo[j]);
        }
        tab.a[ks][0] = pfa_u4{hi[0], hi[1], hi[2], hi[3]};
        tab.a[ks][1] = pfa_u4{lo[0], lo[1], lo[2], lo[3]};
    }
    return tab;
}
// (v_permlane32_swap: the upper 32 lanes of `a` and the lower 32 lanes of `b` change places)
#ifndef PFA_MX_ABL
#define PFA_MX_ABL 0        // probe builds (tools/probe/dft31_mx_probe.hip): 1 no lane swaps, 2 no MFMAs, 4 no operand split -- garbage results, time only
#endif
__device__ __forceinline__ void pfa_swap32(unsigned& a, unsigned& b) {
    if (PFA_MX_ABL & 1) return;
    const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);       // (the builtin, not inline assembly: the compiler pads the VALU -> permlane hazard itself)
    a = r[0], b = r[1];
}
// EVERY lane of the wave must call this (lanes without a column hand in zeros and ignore what they receive).
template <bool INV, class Store>
__device__ __forceinline__ void pfa_dft31_mx(pfa_c (&x)[31], const PfaDft31Tab& tab, Store&& st) {
    float are[16], aim[16], bre[16], bim[16];
    are[0] = x[0].x, aim[0] = x[0].y, bre[0] = 0.f, bim[0] = 0.f;
    float mx = fmaxf(fabsf(are[0]), fabsf(aim[0]));
#pragma unroll
    for (int t = 1; t <= 15; ++t) {
        are[t] = x[t].x + x[31 - t].x, aim[t] = x[t].y + x[31 - t].y;
        bre[t] = x[t].x - x[31 - t].x, bim[t] = x[t].y - x[31 - t].y;
        mx = fmaxf(mx, fmaxf(fmaxf(fabsf(are[t]), fabsf(aim[t])), fmaxf(fabsf(bre[t]), fabsf(bim[t]))));
    }
    // the column's power-of-two scale: its largest operand into [2^14, 2^15)
    const int ex = (int)((__float_as_uint(mx) >> 23) & 0xffu);
    const int e = (ex == 0 || ex == 255) ? 0 : 14 - (ex - 127);
    const int ec = e < -100 ? -100 : (e > 100 ? 100 : e);
    const float sc = __uint_as_float((unsigned)(127 + ec) << 23), un = __uint_as_float((unsigned)(127 - ec - 14) << 23);
    // fragments [operand B1 / B2][K-step][kg][term][4 registers]
    unsigned f[2][2][2][2][4];
    auto split8 = [&](const float* v, unsigned (&hi)[4], unsigned (&lo)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (PFA_MX_ABL & 4) hi[j] = __float_as_uint(v[2 * j]), lo[j] = __float_as_uint(v[2 * j + 1]);
            else pfa_split2h_scaled(v[2 * j], v[2 * j + 1], sc, hi[j], lo[j]);
        }
    };
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
        split8(are + 8 * kg, f[0][0][kg][0], f[0][0][kg][1]);
        split8(bim + 8 * kg, f[0][1][kg][0], f[0][1][kg][1]);
        split8(aim + 8 * kg, f[1][0][kg][0], f[1][0][kg][1]);
        split8(bre + 8 * kg, f[1][1][kg][0], f[1][1][kg][1]);
    }
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int tm = 0; tm < 2; ++tm)
#pragma unroll
                for (int c = 0; c < 4; ++c) pfa_swap32(f[o][ks][0][tm][c], f[o][ks][1][tm][c]);      // [..][0]: column block 0, [..][1]: column block 1
    pfa_f16r acc[2][2];
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[o][nb][r] = 0.f;
    // the three term products of both K-steps, smallest first; the four accumulators take turns (a dependent MFMA waits for its predecessor's last pass)
#pragma unroll
    for (int pr = 0; pr < 3; ++pr)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const pfa_h8 at = __builtin_bit_cast(pfa_h8, tab.a[ks][pr == 0 ? 1 : 0]);
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    const int tb = pr == 1 ? 1 : 0;
                    const pfa_h8 bt = __builtin_bit_cast(pfa_h8, (pfa_u4{f[o][ks][nb][tb][0], f[o][ks][nb][tb][1], f[o][ks][nb][tb][2], f[o][ks][nb][tb][3]}));
                    if (PFA_MX_ABL & 2) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[o][nb][r] += (float)bt[r] + (float)at[r];
                    } else
                        acc[o][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(at, bt, acc[o][nb], 0, 0, 0);
                }
        }
    // every column's rows back to its own lane: afterwards acc[o][hs][r] is row (r & 3) + 8 (r >> 2) + 4 hs of THIS lane's column
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            unsigned u0 = __float_as_uint(acc[o][0][r]), u1 = __float_as_uint(acc[o][1][r]);
            pfa_swap32(u0, u1);
            acc[o][0][r] = __uint_as_float(u0), acc[o][1][r] = __uint_as_float(u1);
        }
    auto row = [&](int o, int m) { return acc[o][(m >> 2) & 1][(m & 3) + 4 * (m >> 3)] * un; };
    st(0, pfa_mk(row(0, 0), row(1, 0)));
#pragma unroll
    for (int q = 1; q <= 15; ++q) {
        // forward: X_q = (B1 row q, B2 row 16 + q), X_{31-q} = (B1 row 16 + q, B2 row q); the inverse transform is the forward one with q <-> 31 - q
        const pfa_c lo_ = pfa_mk(row(0, q), row(1, 16 + q)), hi_ = pfa_mk(row(0, 16 + q), row(1, q));
        st(q, INV ? hi_ : lo_);
        st(31 - q, INV ? lo_ : hi_);
    }
}


template <int FORM, bool INV>
__global__ __launch_bounds__(64, 2) void k_dft(const float2* __restrict__ in, float2* __restrict__ out, int reps) {
    const int lane = threadIdx.x, wv = blockIdx.x;
    const int c = lane < COLS ? lane : COLS - 1;
    pfa_c x[31], y[31];
    const float2* p = in + ((long long)wv * COLS + c) * 31;
#pragma unroll
    for (int t = 0; t < 31; ++t) x[t] = lane < COLS ? pfa_mk(p[t].x, p[t].y) : pfa_mk(0.f, 0.f);
    PfaDft31Tab tab;
    if (FORM == 1) tab = pfa_dft31_tab(lane);
    for (int r = 0; r < reps; ++r) {
        if (FORM == 0) pfa_dft31<INV>(x, [&](int q, pfa_c v) { y[q] = v; });
        else pfa_dft31_mx<INV>(x, tab, [&](int q, pfa_c v) { y[q] = v; });
#pragma unroll
        for (int t = 0; t < 31; ++t) x[t] = r + 1 < reps ? pfa_scale(y[t], 0.17960530202677491f) : y[t];      // 1 / sqrt(31): the values keep their size
    }
    if (lane < COLS) {
        float2* o = out + ((long long)wv * COLS + c) * 31;
#pragma unroll
        for (int t = 0; t < 31; ++t) o[t] = make_float2(x[t].x, x[t].y);
    }
}

int main() {
    const int NW = 256 * 8 * 4, N = NW * COLS * 31;
    std::vector<float2> h(N);
    unsigned s = 12345u;
    for (int i = 0; i < N; ++i) {
        s = s * 1664525u + 1013904223u;
        const float a = ((s >> 8) & 0xffff) / 65536.f - 0.5f;
        s = s * 1664525u + 1013904223u;
        const float b = ((s >> 8) & 0xffff) / 65536.f - 0.5f;
        const float amp = std::exp(-6.f * ((i / 31) % 7) / 7.f) * (1.f + 100.f * ((i % 31) == 3));      // columns of different sizes, one dominant sample
        h[i] = make_float2(a * amp, b * amp);
    }
    float2 *din, *d0, *d1;
    hipMalloc(&din, N * sizeof(float2)), hipMalloc(&d0, N * sizeof(float2)), hipMalloc(&d1, N * sizeof(float2));
    hipMemcpy(din, h.data(), N * sizeof(float2), hipMemcpyHostToDevice);
    for (int inv = 0; inv < 2; ++inv) {
        if (inv) {
            hipLaunchKernelGGL((k_dft<0, true>), dim3(NW), dim3(64), 0, 0, din, d0, 1);
            hipLaunchKernelGGL((k_dft<1, true>), dim3(NW), dim3(64), 0, 0, din, d1, 1);
        } else {
            hipLaunchKernelGGL((k_dft<0, false>), dim3(NW), dim3(64), 0, 0, din, d0, 1);
            hipLaunchKernelGGL((k_dft<1, false>), dim3(NW), dim3(64), 0, 0, din, d1, 1);
        }
        std::vector<float2> o0(N), o1(N);
        hipMemcpy(o0.data(), d0, N * sizeof(float2), hipMemcpyDeviceToHost), hipMemcpy(o1.data(), d1, N * sizeof(float2), hipMemcpyDeviceToHost);
        double e0 = 0, e1 = 0, nn = 0, worst0 = 0, worst1 = 0;
        const int check = 2000;
        for (int c = 0; c < check; ++c) {
            const long long col = (long long)c * 977 % ((long long)NW * COLS);
            double cn = 0, c0 = 0, c1 = 0;
            for (int q = 0; q < 31; ++q) {
                double re = 0, im = 0;
                for (int t = 0; t < 31; ++t) {
                    const double th = (inv ? 1.0 : -1.0) * 2.0 * M_PI * ((t * q) % 31) / 31.0, cs = std::cos(th), sn = std::sin(th);
                    const double xr = h[col * 31 + t].x, xi = h[col * 31 + t].y;
                    re += xr * cs - xi * sn, im += xr * sn + xi * cs;
                }
                cn += re * re + im * im;
                c0 += (o0[col * 31 + q].x - re) * (o0[col * 31 + q].x - re) + (o0[col * 31 + q].y - im) * (o0[col * 31 + q].y - im);
                c1 += (o1[col * 31 + q].x - re) * (o1[col * 31 + q].x - re) + (o1[col * 31 + q].y - im) * (o1[col * 31 + q].y - im);
            }
            e0 += c0, e1 += c1, nn += cn;
            worst0 = std::max(worst0, std::sqrt(c0 / cn)), worst1 = std::max(worst1, std::sqrt(c1 / cn));
        }
        printf("%s: rel-L2 against float64 over %d columns: vector-ALU form %.3e (worst column %.3e), matrix-pipe form %.3e (worst column %.3e)\n", inv ? "inverse" : "forward",
               check, std::sqrt(e0 / nn), worst0, std::sqrt(e1 / nn), worst1);
    }
    hipEvent_t a, b;
    hipEventCreate(&a), hipEventCreate(&b);
    const int reps = 64;
    for (int form = 0; form < 2; ++form) {
        for (int w = 0; w < 2; ++w) {
            hipEventRecord(a);
            if (form == 0) hipLaunchKernelGGL((k_dft<0, false>), dim3(NW), dim3(64), 0, 0, din, d0, reps);
            else hipLaunchKernelGGL((k_dft<1, false>), dim3(NW), dim3(64), 0, 0, din, d1, reps);
            hipEventRecord(b);
            hipEventSynchronize(b);
        }
        float ms = 0;
        hipEventElapsedTime(&ms, a, b);
        printf("%s form: %.1f us for %d waves x %d transforms = %.2f ns per wave-transform per CU-slot (%.3f us per 1920-task slice-pass)\n", form ? "matrix-pipe" : "vector-ALU", ms * 1e3, NW, reps,
               ms * 1e6 / ((double)NW * reps) * 256 * 8, ms * 1e3 / ((double)NW * reps) * 1920);
    }
    return 0;
}
