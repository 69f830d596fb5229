// Probe: issue rate of the bf16 MFMA shapes on gfx950 (32x32x16, the older 32x32x8 "_1k", 16x16x32, 16x16x16 "_1k").
// Build: hipcc --offload-arch=gfx950 -O3 mfma_bf16_probe.hip -o mfma_bf16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 1) void k_probe(const float* in, float* out, int iters) {
    const int tid = threadIdx.x;
    bf16x8 a8, b8;
    s16x4 a4, b4;
    for (int i = 0; i < 8; ++i) a8[i] = (__bf16)in[tid + i], b8[i] = (__bf16)in[tid + 8 + i];
    for (int i = 0; i < 4; ++i) a4[i] = (short)in[tid + i], b4[i] = (short)in[tid + 4 + i];
    float s = 0;
    if (MODE == 0 || MODE == 1) {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) acc[i][r] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    acc[i] = MODE == 0 ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[i], 0, 0, 0)
                                       : __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(a4, b4, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) s += acc[i][r];
    } else {
        f32x4 acc[16];
        for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                acc[i] = MODE == 2 ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[i], 0, 0, 0)
                                   : __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    }
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
void run(const char* name, double flop_per_mfma, float* d_in, float* d_out) {
    const int blocks = 256, iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_probe<MODE>, dim3(blocks), dim3(256), 0, 0, d_in, d_out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double n = 16.0 * iters;                       // MFMAs per wave (one wave per SIMD)
    const double tf = flop_per_mfma * n * blocks * 4 / (ms * 1e-3) / 1e12;
    printf("%-28s %8.1f us  %7.1f TFLOP/s  %5.1f ns per MFMA per SIMD (= %4.1f cycles at 2.4 GHz)\n", name, ms * 1e3, tf, ms * 1e6 / n, ms * 1e6 / n * 2.4);
}

int main() {
    float *d_in, *d_out;
    hipMalloc(&d_in, 4096 * 4);
    hipMemset(d_in, 0, 4096 * 4);
    hipMalloc(&d_out, 256 * 256 * 4);
    run<0>("v_mfma_f32_32x32x16_bf16", 2.0 * 32 * 32 * 16, d_in, d_out);
    run<1>("v_mfma_f32_32x32x8_bf16_1k", 2.0 * 32 * 32 * 8, d_in, d_out);
    run<2>("v_mfma_f32_16x16x32_bf16", 2.0 * 16 * 16 * 32, d_in, d_out);
    run<3>("v_mfma_f32_16x16x16_bf16_1k", 2.0 * 16 * 16 * 16, d_in, d_out);
    return 0;
}
