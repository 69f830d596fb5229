// Stand-alone reproducer for the round-3 finding ("Concurrent streams"): do packed-fp32 vector instructions (v_pk_fma_f32 ...)
// of one kernel return wrong results while waves of ANOTHER kernel on the same SIMD issue XDL MFMAs?  No library dependency.
//   build:  hipcc --offload-arch=gfx950 -O3 -o pk_mfma_repro pk_mfma_repro.hip          run:  ./pk_mfma_repro [reps]
// Victims (one wave per workgroup, <= 128 registers so that foreign waves fit on the SIMD; every lane runs a long dependent recurrence on its own
// data and stores the end values):   0 scalar v_fma_f32 (control)            1 v_pk_fma_f32, inline asm, plain
//                                    2 v_pk_fma_f32 with op_sel / neg modifiers (the forms of pfa372.h)   3 compiler-formed packed ops (ext_vector_type(2))
//                                    4 v_pk_mul_f32 + v_pk_add_f32            5 variant 1 with an LDS exchange between the blocks (the FFT kernels' shape)
//                                    6 / 7 / 8 v_pk_fma_f32 with op_sel only / neg only / one op_sel_hi bit cleared   9 the v_pk_mul / v_pk_add forms of pfa372.h
// Aggressors (own stream, launched first, run several times as long as the victim): 0 none   1 v_mfma_f32_32x32x16_f16   2 v_mfma_f32_16x16x32_f16
//                                    3 v_mfma_f32_32x32x16_bf16   4 fp32-input MFMA 32x32x2 (not XDL-paced: vector pipe)   5 VALU only   6 v_mfma_f32_16x16x32_bf16   7 v_mfma_f32_16x16x16_f16
// Every (victim, aggressor, LDS size, s_nop padding) cell: the victim alone gives the reference bits; then `reps` concurrent runs are compared
// with it bit for bit.  Variants 0 and 1 compute the same recurrence, so their solo results are also compared with each other.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

template <int MODE, int PAD>
__device__ __forceinline__ void step(f2& x, f2& y, const f2 c, const f2 d) {
    // x <- x * c + y ; y <- y * d + x     (|c|, |d| < 1: bounded)
    if (MODE == 0) {
        asm volatile("v_fma_f32 %0, %0, %2, %1" : "+v"(x.x) : "v"(y.x), "v"(c.x));
        asm volatile("v_fma_f32 %0, %0, %2, %1" : "+v"(x.y) : "v"(y.y), "v"(c.y));
        asm volatile("v_fma_f32 %0, %0, %2, %1" : "+v"(y.x) : "v"(x.x), "v"(d.x));
        asm volatile("v_fma_f32 %0, %0, %2, %1" : "+v"(y.y) : "v"(x.y), "v"(d.y));
    } else if (MODE == 1 || MODE == 5) {
        if (PAD) asm volatile("s_nop 3");
        asm volatile("v_pk_fma_f32 %0, %0, %2, %1" : "+v"(x) : "v"(y), "v"(c));
        if (PAD) asm volatile("s_nop 3");
        asm volatile("v_pk_fma_f32 %0, %0, %2, %1" : "+v"(y) : "v"(x), "v"(d));
    } else if (MODE == 2) {  // swapped halves + a negated half: x <- (x.x c.y - y.y, x.y c.x + y.x) and back (the "times +-i" forms)
        if (PAD) asm volatile("s_nop 3");
        asm volatile("v_pk_fma_f32 %0, %0, %2, %1 op_sel:[0,1,1] op_sel_hi:[1,0,0] neg_lo:[0,0,1]" : "+v"(x) : "v"(y), "v"(c));
        if (PAD) asm volatile("s_nop 3");
        asm volatile("v_pk_fma_f32 %0, %0, %2, %1 op_sel:[0,1,1] op_sel_hi:[1,0,0] neg_hi:[0,0,1]" : "+v"(y) : "v"(x), "v"(d));
    } else if (MODE == 6) {  // op_sel / op_sel_hi only (halves swapped, no negation)
        asm volatile("v_pk_fma_f32 %0, %0, %2, %1 op_sel:[0,1,1] op_sel_hi:[1,0,0]" : "+v"(x) : "v"(y), "v"(c));
        asm volatile("v_pk_fma_f32 %0, %0, %2, %1 op_sel:[0,1,1] op_sel_hi:[1,0,0]" : "+v"(y) : "v"(x), "v"(d));
    } else if (MODE == 7) {  // neg only
        asm volatile("v_pk_fma_f32 %0, %0, %2, %1 neg_lo:[0,0,1]" : "+v"(x) : "v"(y), "v"(c));
        asm volatile("v_pk_fma_f32 %0, %0, %2, %1 neg_hi:[0,0,1]" : "+v"(y) : "v"(x), "v"(d));
    } else if (MODE == 8) {  // one operand broadcast from its low half (op_sel_hi bit cleared), nothing else
        asm volatile("v_pk_fma_f32 %0, %0, %2, %1 op_sel_hi:[1,1,0]" : "+v"(x) : "v"(y), "v"(c));
        asm volatile("v_pk_fma_f32 %0, %0, %2, %1 op_sel_hi:[1,1,0]" : "+v"(y) : "v"(x), "v"(d));
    } else if (MODE == 9) {  // the add / mul forms of pfa372.h: a + (-+ i b), (a.re, a.im) * s
        f2 t;
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(x), "v"(c));
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(x) : "v"(t), "v"(y));
        asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "v"(y), "v"(d));
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(y) : "v"(t), "v"(x));
    } else if (MODE == 3) {
        x = __builtin_elementwise_fma(x, c, y);
        y = __builtin_elementwise_fma(y, d, x);
    } else {  // 4
        f2 t;
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(x), "v"(c));
        if (PAD) asm volatile("s_nop 3");
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(x) : "v"(t), "v"(y));
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(y), "v"(d));
        if (PAD) asm volatile("s_nop 3");
        asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(y) : "v"(t), "v"(x));
    }
}

template <int MODE, int PAD, int REGS>
__global__ __launch_bounds__(64) void k_victim(const float* __restrict__ in, float* __restrict__ out, int iters) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x;
    const size_t base = (size_t)blockIdx.x * 64 * 16;
    f2 x[4], y[4], c, d;
    for (int i = 0; i < 4; ++i) {
        x[i] = (f2){in[base + lane * 16 + 4 * i], in[base + lane * 16 + 4 * i + 1]};
        y[i] = (f2){in[base + lane * 16 + 4 * i + 2], in[base + lane * 16 + 4 * i + 3]};
    }
    c = (f2){0.61803399f + 1e-3f * lane, -0.70710678f};
    d = (f2){-0.5f, 0.33333334f - 1e-3f * lane};
    if (REGS > 64) asm volatile("v_mov_b32 v%c0, 0" ::"i"(REGS - 1) : "v127");  // raise the allocation to 128 registers (two waves per SIMD left)
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) step<MODE, PAD>(x[i], y[i], c, d);
        if (MODE == 5) {  // exchange through the wave's LDS (stride 65: conflict-free), as the transposes of the transform kernels do
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                lds[(2 * i) * 65 + lane] = x[i].x;
                lds[(2 * i + 1) * 65 + lane] = x[i].y;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                x[i].x = lds[(2 * i) * 65 + (lane ^ 1)] * 0.5f + 0.25f * x[i].x;
                x[i].y = lds[(2 * i + 1) * 65 + (lane ^ 33)] * 0.5f + 0.25f * x[i].y;
            }
        }
    }
    for (int i = 0; i < 4; ++i) {
        out[base + lane * 16 + 4 * i] = x[i].x;
        out[base + lane * 16 + 4 * i + 1] = x[i].y;
        out[base + lane * 16 + 4 * i + 2] = y[i].x;
        out[base + lane * 16 + 4 * i + 3] = y[i].y;
    }
}

template <int KIND>
__global__ __launch_bounds__(64) void k_aggressor(const float* __restrict__ in, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x;
    float s = 0.f;
    if (KIND == 1 || KIND == 3) {
        f32x16 acc[2];
        for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;
        h8 a, b;
        b8 ab, bb;
        for (int j = 0; j < 8; ++j) {
            a[j] = (_Float16)(in[lane * 8 + j] * 0.01f);
            b[j] = (_Float16)(in[512 + lane * 8 + j] * 0.01f);
            ab[j] = (__bf16)(in[lane * 8 + j] * 0.01f);
            bb[j] = (__bf16)(in[512 + lane * 8 + j] * 0.01f);
        }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (KIND == 1) {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc[1], 0, 0, 0);
                } else {
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bb, ab, acc[1], 0, 0, 0);
                }
            }
        }
        for (int r = 0; r < 16; ++r) s += acc[0][r] + acc[1][r];
    } else if (KIND == 2) {
        f32x4 acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0, 0, 0, 0};
        h8 a, b;
        for (int j = 0; j < 8; ++j) {
            a[j] = (_Float16)(in[lane * 8 + j] * 0.01f);
            b[j] = (_Float16)(in[512 + lane * 8 + j] * 0.01f);
        }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u & 3], 0, 0, 0);
                acc[(u + 2) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, acc[(u + 2) & 3], 0, 0, 0);
            }
        }
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else if (KIND == 6 || KIND == 7) {
        f32x4 acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0, 0, 0, 0};
        b8 ab, bb;
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        h4 a4, b4;
        for (int j = 0; j < 8; ++j) {
            ab[j] = (__bf16)(in[lane * 8 + j] * 0.01f);
            bb[j] = (__bf16)(in[512 + lane * 8 + j] * 0.01f);
        }
        for (int j = 0; j < 4; ++j) {
            a4[j] = (_Float16)(in[lane * 4 + j] * 0.01f);
            b4[j] = (_Float16)(in[512 + lane * 4 + j] * 0.01f);
        }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (KIND == 6) {
                    acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[u & 3], 0, 0, 0);
                    acc[(u + 2) & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bb, ab, acc[(u + 2) & 3], 0, 0, 0);
                } else {
                    acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, acc[u & 3], 0, 0, 0);
                    acc[(u + 2) & 3] = __builtin_amdgcn_mfma_f32_16x16x16f16(b4, a4, acc[(u + 2) & 3], 0, 0, 0);
                }
            }
        }
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else if (KIND == 4) {
        f32x16 acc[2];
        for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;
        const float a = in[lane] * 0.01f, b = in[64 + lane] * 0.01f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[1], 0, 0, 0);
            }
        }
        for (int r = 0; r < 16; ++r) s += acc[0][r] + acc[1][r];
    } else {  // 5: vector ALU only
        float x0 = in[lane], x1 = in[64 + lane], x2 = in[128 + lane], x3 = in[192 + lane];
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x0) : "v"(0.5f), "v"(x1));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x1) : "v"(-0.5f), "v"(x2));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x2) : "v"(0.25f), "v"(x3));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x3) : "v"(-0.25f), "v"(x0));
            }
        }
        s = x0 + x1 + x2 + x3;
    }
    out[(size_t)blockIdx.x * 64 + lane] = s;
}

typedef void (*vkern)(const float*, float*, int);
struct Victim { const char* name; vkern fn; };
struct Aggr { const char* name; vkern fn; int iters_scale; };

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 6;
    const int NWG = 256 * 8, VIT = 2000;
    const size_t nv = (size_t)NWG * 64 * 16;
    std::vector<float> h(nv);
    unsigned s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f - 0.5f; }
    float *d_in, *d_out, *d_aout;
    CK(hipMalloc(&d_in, nv * 4)); CK(hipMalloc(&d_out, nv * 4)); CK(hipMalloc(&d_aout, (size_t)NWG * 4 * 64 * 4));
    CK(hipMemcpy(d_in, h.data(), nv * 4, hipMemcpyHostToDevice));
    hipStream_t sv, sa;
    CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    Victim victims[] = {{"0 scalar v_fma_f32", k_victim<0, 0, 128>}, {"1 v_pk_fma_f32 asm", k_victim<1, 0, 128>}, {"1p v_pk_fma_f32 asm + s_nop 3", k_victim<1, 1, 128>},
                        {"2 v_pk_fma_f32 op_sel/neg", k_victim<2, 0, 128>}, {"3 compiler-formed packed", k_victim<3, 0, 128>}, {"4 v_pk_mul + v_pk_add", k_victim<4, 0, 128>},
                        {"4p v_pk_mul + s_nop 3 + v_pk_add", k_victim<4, 1, 128>}, {"5 v_pk_fma + LDS exchange", k_victim<5, 0, 128>}, {"1s v_pk_fma_f32, 64 registers", k_victim<1, 0, 64>},
                        {"2p v_pk_fma op_sel/neg + s_nop 3", k_victim<2, 1, 128>}, {"2s v_pk_fma op_sel/neg, 64 registers", k_victim<2, 0, 64>}, {"6 v_pk_fma op_sel only", k_victim<6, 0, 128>},
                        {"7 v_pk_fma neg only", k_victim<7, 0, 128>}, {"8 v_pk_fma op_sel_hi broadcast", k_victim<8, 0, 128>}, {"9 v_pk_mul/add op_sel+neg (pfa372)", k_victim<9, 0, 128>}};
    Aggr aggrs[] = {{"none", nullptr, 0}, {"mfma_f32_32x32x16_f16", k_aggressor<1>, 1}, {"mfma_f32_16x16x32_f16", k_aggressor<2>, 2}, {"mfma_f32_32x32x16_bf16", k_aggressor<3>, 1},
                    {"mfma_f32_32x32x2_f32", k_aggressor<4>, 1}, {"valu only", k_aggressor<5>, 4}, {"mfma_f32_16x16x32_bf16", k_aggressor<6>, 2}, {"mfma_f32_16x16x16_f16", k_aggressor<7>, 2}};
    const int lds_sizes[] = {2080, 17 * 1024, 40 * 1024};
    std::vector<float> ref(nv), got(nv), ref0;
    int total_bad_cells = 0;
    for (auto& V : victims) {
        for (int lds : lds_sizes) {
            if (lds != 2080 && V.name[0] != '1' && V.name[0] != '5') continue;
            CK(hipMemsetAsync(d_out, 0, nv * 4, sv));
            hipLaunchKernelGGL(V.fn, dim3(NWG), dim3(64), lds, sv, d_in, d_out, VIT);
            CK(hipStreamSynchronize(sv));
            CK(hipMemcpy(ref.data(), d_out, nv * 4, hipMemcpyDeviceToHost));
            if (V.name[0] == '0') ref0 = ref;
            if (V.name[0] == '1' && V.name[1] == ' ' && lds == 2080)
                printf("    solo scalar == solo packed (same recurrence): %s\n", memcmp(ref.data(), ref0.data(), nv * 4) == 0 ? "bit-identical" : "DIFFERENT");
            for (auto& A : aggrs) {
                size_t bad = 0, bad_wg = 0;
                float ms_v = 0;
                for (int r = 0; r < reps; ++r) {
                    CK(hipMemsetAsync(d_out, 0, nv * 4, sv));
                    CK(hipStreamSynchronize(sv));
                    hipEvent_t e0, e1;
                    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                    if (A.fn) hipLaunchKernelGGL(A.fn, dim3(NWG * 4), dim3(64), 0, sa, d_in, d_aout, 6000 * A.iters_scale);
                    CK(hipEventRecord(e0, sv));
                    hipLaunchKernelGGL(V.fn, dim3(NWG), dim3(64), lds, sv, d_in, d_out, VIT);
                    CK(hipEventRecord(e1, sv));
                    CK(hipStreamSynchronize(sv)); CK(hipStreamSynchronize(sa));
                    CK(hipEventElapsedTime(&ms_v, e0, e1));
                    CK(hipMemcpy(got.data(), d_out, nv * 4, hipMemcpyDeviceToHost));
                    for (int w = 0; w < NWG; ++w) {
                        size_t b = 0;
                        for (int i = 0; i < 1024; ++i) b += memcmp(&got[(size_t)w * 1024 + i], &ref[(size_t)w * 1024 + i], 4) != 0;
                        bad += b; bad_wg += b != 0;
                    }
                    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
                }
                printf("victim [%-34s] lds %5d  aggressor [%-24s]  victim %.2f ms  wrong values %zu in %zu workgroup-runs of %d  %s\n", V.name, lds, A.name, ms_v, bad, bad_wg,
                       NWG * reps, bad ? "<-- MISMATCH" : "ok");
                fflush(stdout);
                total_bad_cells += bad != 0;
            }
        }
    }
    printf("cells with mismatches: %d\n", total_bad_cells);
    return 0;
}
