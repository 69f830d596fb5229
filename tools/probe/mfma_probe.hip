// Probe: fp32-input MFMA issue rate and s_memtime tick rate on gfx950.  Build: hipcc --offload-arch=gfx950 -O3 mfma_probe.hip -o mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(512, 2) void k_probe(const float* in, float* out, unsigned long long* ticks, int iters) {
    const int tid = threadIdx.x;
    float a = in[tid], b = in[tid + 512];
    unsigned long long t0 = __builtin_readcyclecounter();
    if (MODE == 0) {  // 16x16x4, 32 independent accumulators
        f32x4 acc[32];
        for (int i = 0; i < 32; ++i) acc[i] = (f32x4){0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
        float s = 0;
        for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        out[blockIdx.x * 512 + tid] = s;
    } else if (MODE == 1) {  // 32x32x2, 8 independent accumulators
        f32x16 acc[8];
        for (int i = 0; i < 8; ++i)
            for (int r = 0; r < 16; ++r) acc[i][r] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[i], 0, 0, 0);
        }
        float s = 0;
        for (int i = 0; i < 8; ++i)
            for (int r = 0; r < 16; ++r) s += acc[i][r];
        out[blockIdx.x * 512 + tid] = s;
    } else {  // 16x16x4, pairs revisiting the same accumulator after 2 MFMAs (the Winograd loop's pattern)
        f32x4 acc[32];
        for (int i = 0; i < 32; ++i) acc[i] = (f32x4){0, 0, 0, 0};
        for (int it = 0; it < iters / 2; ++it) {
#pragma unroll
            for (int i = 0; i < 32; i += 2) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
                acc[i + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, b, acc[i + 1], 0, 0, 0);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc[i], 0, 0, 0);
                acc[i + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, a, acc[i + 1], 0, 0, 0);
            }
        }
        float s = 0;
        for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        out[blockIdx.x * 512 + tid] = s;
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int blocks, int threads, int iters, float* d_in, float* d_out, unsigned long long* d_t, double flop_per_iter_wave) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_probe<MODE>, dim3(blocks), dim3(threads), 0, 0, d_in, d_out, d_t, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), d_t, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double mt = 0;
    for (auto v : h) mt += (double)v;
    mt /= blocks;
    const double waves = (double)blocks * threads / 64;
    const double tf = flop_per_iter_wave * iters * waves / (ms * 1e-3) / 1e12;
    printf("%-44s blocks %4d x %3d thr: %8.1f us  %6.1f TFLOP/s  ticks/block %.0f  tick rate %.3f GHz (if the block spans the kernel)\n", name, blocks,
           threads, ms * 1e3, tf, mt, mt / (ms * 1e-3) / 1e9);
}

int main(int argc, char** argv) {
    float *d_in, *d_out;
    unsigned long long* d_t;
    hipMalloc(&d_in, 4096 * 4);
    hipMalloc(&d_out, 4096 * 512 * 4);
    hipMalloc(&d_t, 4096 * 8);
    std::vector<float> h(4096);
    const bool zeros = argc > 1 && atoi(argv[1]) == 0;
    for (int i = 0; i < 4096; ++i) h[i] = zeros ? 0.f : (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(d_in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    printf("inputs: %s\n", zeros ? "zeros" : "pseudo-random");
    // per wave per iter: MODE0 32 MFMA x 2048 flop; MODE1 16 x 4096; MODE2 (iters/2) x 64 x 2048 -> 32 x 2048 per iter
    run<0>("16x16x4 32 independent acc, 2 waves/SIMD", 256, 512, 2000, d_in, d_out, d_t, 32 * 2048.0);
    run<0>("16x16x4 32 independent acc, 1 wave/SIMD", 256, 256, 2000, d_in, d_out, d_t, 32 * 2048.0);
    run<1>("32x32x2 8 independent acc, 2 waves/SIMD", 256, 512, 2000, d_in, d_out, d_t, 16 * 4096.0);
    run<1>("32x32x2 8 independent acc, 1 wave/SIMD", 256, 256, 2000, d_in, d_out, d_t, 16 * 4096.0);
    run<2>("16x16x4 acc revisited after 2, 2 waves/SIMD", 256, 512, 2000, d_in, d_out, d_t, 32 * 2048.0);
    run<2>("16x16x4 acc revisited after 2, 1 wave/SIMD", 256, 256, 2000, d_in, d_out, d_t, 32 * 2048.0);
    return 0;
}
