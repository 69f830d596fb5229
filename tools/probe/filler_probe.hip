// Probe: cost of non-MFMA instructions issued beside fp32 MFMAs on gfx950 (1 or 2 waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// KIND 0: v_add_f32 fillers, 1: ds_read_b32 fillers, 2: s_nop fillers, 3: v_add fillers clustered after 4 MFMAs
template <int K, int KIND, int BIG>
__global__ __launch_bounds__(512, 2) void k_probe(const float* in, float* out, unsigned long long* ticks, int iters) {
    __shared__ float lds[4096];
    const int tid = threadIdx.x;
    lds[tid] = in[tid];
    lds[tid + 512] = in[tid + 512];
    __syncthreads();
    float a = in[tid], b = in[tid + 512];
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = in[tid + i];
    f32x4 acc[16];
    f32x16 accb[4];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) accb[i][r] = 0;
    const float* lp = lds + (tid & 63);
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (BIG) accb[i & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, accb[i & 3], 0, 0, 0);
            else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            if (KIND != 3 || (i & 3) == 3) {
#pragma unroll
                for (int k = 0; k < (KIND == 3 ? 4 * K : K); ++k) {
                    if (KIND == 0 || KIND == 3) f[k & 7] = f[k & 7] + a;
                    if (KIND == 1) f[k & 7] += lp[(i * 8 + k) * 64 % 3072];
                    if (KIND == 2) asm volatile("s_nop 0");
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += accb[i][r];
    for (int i = 0; i < 8; ++i) s += f[i];
    out[blockIdx.x * 512 + tid] = s;
    __syncthreads();
    unsigned long long t1 = __builtin_readcyclecounter();
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int K, int KIND, int BIG>
void run(int threads, float* d_in, float* d_out, unsigned long long* d_t) {
    const int blocks = 256, iters = 500;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k_probe<K, KIND, BIG>), dim3(blocks), dim3(threads), 0, 0, d_in, d_out, d_t, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), d_t, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double mt = 0;
    for (auto v : h) mt += (double)v;
    mt /= blocks;
    const double n_mfma_simd = (double)iters * 16 * (threads / 256);  // MFMAs issued per SIMD
    static const char* kinds[] = {"v_add", "ds_read_b32", "s_nop", "v_add clustered x4"};
    printf("%s K=%d fillers/MFMA (%s), %d wave(s)/SIMD: %.1f cycles per MFMA per SIMD (MFMA alone %d)\n", BIG ? "32x32x2" : "16x16x4", K, kinds[KIND],
           threads / 256, mt / n_mfma_simd, BIG ? 64 : 32);
}

int main() {
    float *d_in, *d_out;
    unsigned long long* d_t;
    hipMalloc(&d_in, 4096 * 4);
    hipMalloc(&d_out, 256 * 512 * 4);
    hipMalloc(&d_t, 4096 * 8);
    std::vector<float> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.5f;
    hipMemcpy(d_in, h.data(), 4096 * 4, hipMemcpyHostToDevice);
#define RUNK(K, KIND, BIG) run<K, KIND, BIG>(256, d_in, d_out, d_t); run<K, KIND, BIG>(512, d_in, d_out, d_t);
    RUNK(0, 0, 0) RUNK(1, 0, 0) RUNK(2, 0, 0) RUNK(3, 0, 0) RUNK(4, 0, 0) RUNK(6, 0, 0)
    RUNK(1, 1, 0) RUNK(2, 1, 0) RUNK(3, 1, 0)
    RUNK(2, 2, 0) RUNK(4, 2, 0)
    RUNK(2, 3, 0) RUNK(3, 3, 0)
    RUNK(0, 0, 1) RUNK(2, 0, 1) RUNK(4, 0, 1) RUNK(8, 0, 1) RUNK(3, 1, 1)
    return 0;
}
